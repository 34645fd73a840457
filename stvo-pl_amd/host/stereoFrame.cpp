// stereoFrame.cpp — stereo association of one frame: the host glue of
// /root/reference/src/stereoFrame.cpp:120-173 (points) and :309-415 (lines) around the GPU grid
// matchers.  Arithmetic types follow the reference: key-point coordinates are FLOAT, the grid scale
// is double, products are truncated to int (:47-48,129-139,318-338).
#include "stereoFrame.h"

#include <future>

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>

#include "../csrc/pose_math.h"

namespace StVO {

namespace {
void check(int rc, const char* what) {
    if (rc != STVO_OK) throw std::runtime_error(std::string("[StVO-HIP] ") + what + ": " + stvo_error_string(rc));
}
}  // namespace

StereoFrame::StereoFrame(const FrameFeatures& f, const int idx_, PinholeStereoCamera* cam_, stvo_ctx* ctx_, stvo_ctx* ctx_lines_)
    : frame_idx(idx_), points_l(f.points_l), points_r(f.points_r), lines_l(f.lines_l), lines_r(f.lines_r),
      pdesc_l(f.pdesc_l), pdesc_r(f.pdesc_r), ldesc_l(f.ldesc_l), ldesc_r(f.ldesc_r), cam(cam_), ctx(ctx_), ctx_l(ctx_lines_ ? ctx_lines_ : ctx_) {
    if (f.img_cols <= 0 || f.img_rows <= 0) throw std::runtime_error("[StereoFrame] invalid image size");
    if ((int)points_l.size() != pdesc_l.rows || (int)points_r.size() != pdesc_r.rows ||
        (int)lines_l.size() != ldesc_l.rows || (int)lines_r.size() != ldesc_r.rows)
        throw std::runtime_error("[StereoFrame] each feature needs a corresponding descriptor");
    inv_width = GRID_COLS / static_cast<double>(f.img_cols);
    inv_height = GRID_ROWS / static_cast<double>(f.img_rows);
    Tfw = Matrix4d::Identity();
    DT = Matrix4d::Identity();
    Tfw_cov = Matrix6d::Zero();
    DT_cov = Matrix6d::Zero();
    for (int i = 0; i < 6; ++i) DT_cov_eig(i) = 0.0;
    err_norm = -1.0;
}

StereoFrame::~StereoFrame() {
    for (auto pt : stereo_pt) delete pt;
    for (auto ls : stereo_ls) delete ls;
}

void StereoFrame::extractStereoFeatures(double /*llength_th*/, int /*fast_th*/) {
    // detection is done upstream of this library; the two associations are independent.  Like the reference
    // (:64-72: two std::async tasks when plInParallel), points and lines run concurrently when a second context
    // exists: each task owns its context (stream + staging arena), so the GPU overlaps the two kernel chains.
    if (Config::plInParallel() && Config::hasPoints() && Config::hasLines() && ctx_l != ctx && !lines_l.empty() && !lines_r.empty()) {
        auto lines = std::async(std::launch::async, [&] { matchStereoLines(lines_l, lines_r, ldesc_l, ldesc_r, (frame_idx == 0)); });
        matchStereoPoints(points_l, points_r, pdesc_l, pdesc_r, (frame_idx == 0));
        lines.get();  // rethrows
        return;
    }
    if (Config::hasPoints()) matchStereoPoints(points_l, points_r, pdesc_l, pdesc_r, (frame_idx == 0));
    if (Config::hasLines()) matchStereoLines(lines_l, lines_r, ldesc_l, ldesc_r, (frame_idx == 0));
}

namespace {
// GridStructure filled with grid.at(x,y).push_back(idx) (src/gridStructure.cpp:56-63) as CSR
void build_csr(const std::vector<int32_t>& xy, const std::vector<int32_t>& owner, std::vector<int32_t>& start,
               std::vector<int32_t>& items) {
    const int n = (int)owner.size();
    start.assign(GRID_COLS * GRID_ROWS + 1, 0);
    auto inb = [&](int k) { return xy[2 * k] >= 0 && xy[2 * k] < GRID_COLS && xy[2 * k + 1] >= 0 && xy[2 * k + 1] < GRID_ROWS; };
    for (int k = 0; k < n; ++k)
        if (inb(k)) start[xy[2 * k + 1] * GRID_COLS + xy[2 * k] + 1]++;
    for (int c = 0; c < GRID_COLS * GRID_ROWS; ++c) start[c + 1] += start[c];
    items.assign((size_t)start.back() + 1, 0);
    std::vector<int32_t> fill(GRID_COLS * GRID_ROWS, 0);
    for (int k = 0; k < n; ++k)
        if (inb(k)) {
            const int c = xy[2 * k + 1] * GRID_COLS + xy[2 * k];
            items[start[c] + fill[c]++] = owner[k];
        }
}

// LineIterator / getLineCoords (src/lineIterator.cpp:34-77, src/gridStructure.cpp:33-41)
void line_coords(double x1, double y1, double x2, double y2, std::vector<int32_t>& out_xy) {
    const bool steep = std::abs(y2 - y1) > std::abs(x2 - x1);
    if (steep) {
        std::swap(x1, y1);
        std::swap(x2, y2);
    }
    if (x1 > x2) {
        std::swap(x1, x2);
        std::swap(y1, y2);
    }
    const double dx = x2 - x1, dy = std::abs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int y = static_cast<int>(y1);
    const int maxX = static_cast<int>(x2);
    for (int x = static_cast<int>(x1); x <= maxX; ++x) {
        out_xy.push_back(steep ? y : x);
        out_xy.push_back(steep ? x : y);
        error -= dy;
        if (error < 0) {
            y += ystep;
            error += dx;
        }
    }
}
}  // namespace

void StereoFrame::matchStereoPoints(std::vector<KeyPoint> points_l_, std::vector<KeyPoint> points_r_, DescMat& pdesc_l_,
                                    DescMat pdesc_r_, bool initial) {
    for (auto pt : stereo_pt) delete pt;
    stereo_pt.clear();
    if (!Config::hasPoints() || points_l_.empty() || points_r_.empty()) return;

    std::vector<int32_t> coords(2 * points_l_.size());
    for (size_t i = 0; i < points_l_.size(); ++i) {
        coords[2 * i] = (int)(points_l_[i].x * inv_width);  // float * double -> int truncation (:132)
        coords[2 * i + 1] = (int)(points_l_[i].y * inv_height);
    }
    std::vector<int32_t> rxy(2 * points_r_.size()), owner(points_r_.size()), start, items;
    for (size_t i = 0; i < points_r_.size(); ++i) {
        rxy[2 * i] = (int)(points_r_[i].x * inv_width);
        rxy[2 * i + 1] = (int)(points_r_[i].y * inv_height);
        owner[i] = (int)i;
    }
    build_csr(rxy, owner, start, items);

    stvo_grid_window w{Config::matchingSWs(), 0, 0, 0};  // :141-143
    std::vector<int32_t> matches_12(points_l_.size());
    check(stvo_match_grid_points(ctx, coords.data(), pdesc_l_.ptr(), (int)points_l_.size(), start.data(), items.data(),
                                 pdesc_r_.ptr(), (int)points_r_.size(), &w, Config::minRatio12P(),
                                 Config::bestLRMatches() ? 1 : 0, matches_12.data(), nullptr),
          "stvo_match_grid_points");
    buildStereoPoints(points_l_, points_r_, pdesc_l_, matches_12.data(), initial);
}

// :149-172 — epipolar / disparity filters, back-projection and the filtered descriptor matrix, given matches_12
void StereoFrame::buildStereoPoints(const std::vector<KeyPoint>& points_l_, const std::vector<KeyPoint>& points_r_,
                                    DescMat& pdesc_l_, const int32_t* matches_12_, bool initial) {
    for (auto pt : stereo_pt) delete pt;
    stereo_pt.clear();
    if (!Config::hasPoints() || points_l_.empty() || points_r_.empty()) return;
    struct View { const int32_t* p; size_t n; size_t size() const { return n; } int operator[](size_t i) const { return p[i]; } };
    const View matches_12{matches_12_, points_l_.size()};
    DescMat pdesc_l_aux;
    int pt_idx = 0;
    for (size_t i1 = 0; i1 < matches_12.size(); ++i1) {
        const int i2 = matches_12[i1];
        if (i2 < 0) continue;
        if (std::abs(points_l_[i1].y - points_r_[i2].y) <= Config::maxDistEpip()) {  // float difference (:157)
            const double disp_ = points_l_[i1].x - points_r_[i2].x;                  // float difference (:159)
            if (disp_ >= Config::minDisp()) {
                pdesc_l_aux.push_back_row(pdesc_l_.ptr((int)i1));
                Vector2d pl_;
                pl_(0) = points_l_[i1].x;
                pl_(1) = points_l_[i1].y;
                Vector3d P_ = cam->backProjection(pl_(0), pl_(1), disp_);
                stereo_pt.push_back(new PointFeature(pl_, disp_, P_, initial ? pt_idx++ : -1, points_l_[i1].octave));
            }
        }
    }
    pdesc_l_ = pdesc_l_aux;
}

void StereoFrame::matchStereoLines(std::vector<KeyLine> lines_l_, std::vector<KeyLine> lines_r_, DescMat& ldesc_l_,
                                   DescMat ldesc_r_, bool initial) {
    for (auto ls : stereo_ls) delete ls;
    stereo_ls.clear();
    if (!Config::hasLines() || lines_l_.empty() || lines_r_.empty()) return;

    std::vector<int32_t> coords(4 * lines_l_.size());
    for (size_t i = 0; i < lines_l_.size(); ++i) {
        const KeyLine& kl = lines_l_[i];
        coords[4 * i] = (int)(kl.startPointX * inv_width);
        coords[4 * i + 1] = (int)(kl.startPointY * inv_height);
        coords[4 * i + 2] = (int)(kl.endPointX * inv_width);
        coords[4 * i + 3] = (int)(kl.endPointY * inv_height);
    }
    // rasterise the right lines into the grid, unit directions in scaled space (:325-338)
    std::vector<int32_t> ent_xy, ent_owner, start, items;
    std::vector<double> directions(2 * lines_r_.size());
    for (size_t idx = 0; idx < lines_r_.size(); ++idx) {
        const KeyLine& kl = lines_r_[idx];
        double vx = (kl.endPointX - kl.startPointX) * inv_width;  // float difference, then * double
        double vy = (kl.endPointY - kl.startPointY) * inv_height;
        const double mag = std::sqrt(vx * vx + vy * vy);
        directions[2 * idx] = vx / mag;
        directions[2 * idx + 1] = vy / mag;
        const size_t before = ent_xy.size() / 2;
        line_coords(kl.startPointX * inv_width, kl.startPointY * inv_height, kl.endPointX * inv_width,
                    kl.endPointY * inv_height, ent_xy);
        ent_owner.insert(ent_owner.end(), ent_xy.size() / 2 - before, (int)idx);
    }
    build_csr(ent_xy, ent_owner, start, items);

    stvo_grid_window w{Config::matchingSWs(), 0, 0, 0};
    std::vector<int32_t> matches_12(lines_l_.size());
    check(stvo_match_grid_lines(ctx_l, coords.data(), ldesc_l_.ptr(), (int)lines_l_.size(), start.data(), items.data(),
                                ldesc_r_.ptr(), (int)lines_r_.size(), directions.data(), &w, Config::minRatio12P(),
                                Config::lineSimTh(), Config::bestLRMatches() ? 1 : 0, matches_12.data(), nullptr),
          "stvo_match_grid_lines");
    buildStereoLines(lines_l_, lines_r_, ldesc_l_, matches_12.data(), initial);
}

// :348-397 — overlap / disparity / horizontality filters, end-point re-intersection, back-projection, given matches_12
void StereoFrame::buildStereoLines(const std::vector<KeyLine>& lines_l_, const std::vector<KeyLine>& lines_r_,
                                   DescMat& ldesc_l_, const int32_t* matches_12_, bool initial) {
    for (auto ls : stereo_ls) delete ls;
    stereo_ls.clear();
    if (!Config::hasLines() || lines_l_.empty() || lines_r_.empty()) return;
    struct View { const int32_t* p; size_t n; size_t size() const { return n; } int operator[](size_t i) const { return p[i]; } };
    const View matches_12{matches_12_, lines_l_.size()};
    DescMat ldesc_l_aux;
    int ls_idx = 0;
    for (size_t i1 = 0; i1 < matches_12.size(); ++i1) {
        const int i2 = matches_12[i1];
        if (i2 < 0) continue;
        // :353-358
        double sp_l[3] = {lines_l_[i1].startPointX, lines_l_[i1].startPointY, 1.0};
        double ep_l[3] = {lines_l_[i1].endPointX, lines_l_[i1].endPointY, 1.0};
        Vector3d le_l;
        le_l(0) = sp_l[1] * ep_l[2] - sp_l[2] * ep_l[1];
        le_l(1) = sp_l[2] * ep_l[0] - sp_l[0] * ep_l[2];
        le_l(2) = sp_l[0] * ep_l[1] - sp_l[1] * ep_l[0];
        const double nrm = std::sqrt(le_l(0) * le_l(0) + le_l(1) * le_l(1));
        for (int k = 0; k < 3; ++k) le_l(k) = le_l(k) / nrm;
        double sp_r[2] = {lines_r_[i2].startPointX, lines_r_[i2].startPointY};
        double ep_r[2] = {lines_r_[i2].endPointX, lines_r_[i2].endPointY};

        const double overlap = lineSegmentOverlapStereo(sp_l[1], ep_l[1], sp_r[1], ep_r[1]);

        // :363-364 — the second line reads the ALREADY overwritten sp_r (reference quirk, kept)
        sp_r[0] = (sp_r[0] * (sp_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - sp_l[1])) / (sp_r[1] - ep_r[1]);
        sp_r[1] = sp_l[1];
        ep_r[0] = (sp_r[0] * (ep_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - ep_l[1])) / (sp_r[1] - ep_r[1]);
        ep_r[1] = ep_l[1];
        double disp_s, disp_e;
        Vector2d a, b, c, d;
        a(0) = sp_l[0]; a(1) = sp_l[1]; b(0) = ep_l[0]; b(1) = ep_l[1];
        c(0) = sp_r[0]; c(1) = sp_r[1]; d(0) = ep_r[0]; d(1) = ep_r[1];
        filterLineSegmentDisparity(a, b, c, d, disp_s, disp_e);

        if (disp_s >= Config::minDisp() && disp_e >= Config::minDisp() &&
            std::abs(sp_l[1] - ep_l[1]) > Config::lineHorizTh() && std::abs(sp_r[1] - ep_r[1]) > Config::lineHorizTh() &&
            overlap > Config::stereoOverlapTh()) {
            Vector3d sP_ = cam->backProjection(sp_l[0], sp_l[1], disp_s);
            Vector3d eP_ = cam->backProjection(ep_l[0], ep_l[1], disp_e);
            const double angle_l = lines_l_[i1].angle;
            ldesc_l_aux.push_back_row(ldesc_l_.ptr((int)i1));
            stereo_ls.push_back(new LineFeature(a, disp_s, sP_, b, disp_e, eP_, le_l, angle_l, initial ? ls_idx : -1,
                                                lines_l_[i1].octave));
            if (initial) ls_idx++;
        }
    }
    ldesc_l_ = ldesc_l_aux;
}

// Device-pipeline mode of the handler: the grid matching of this frame has already run on the GPU (stvo_seq_*);
// only the host-side feature lists remain to be built, exactly as after matchGrid in the two functions above.
void StereoFrame::adoptStereoMatches(const int32_t* m12_points, const int32_t* m12_lines) {
    if (Config::hasPoints()) buildStereoPoints(points_l, points_r, pdesc_l, m12_points, (frame_idx == 0));
    if (Config::hasLines()) buildStereoLines(lines_l, lines_r, ldesc_l, m12_lines, (frame_idx == 0));
}

// :405-415 and :473-508 — the same building blocks line_tail_kernel uses (csrc/pose_math.h)
void StereoFrame::filterLineSegmentDisparity(Vector2d spl, Vector2d epl, Vector2d spr, Vector2d epr, double& disp_s,
                                             double& disp_e) {
    pm::stereo_line_disparities(spl(0), epl(0), spr(0), epr(0), Config::lsMinDispRatio(), &disp_s, &disp_e);
}

double StereoFrame::lineSegmentOverlapStereo(double spl_obs, double epl_obs, double spl_proj, double epl_proj) {
    return pm::stereo_row_overlap(spl_obs, epl_obs, spl_proj, epl_proj, Config::lineHorizTh());
}

// :510-616 — same building block the device kernel uses
double StereoFrame::lineSegmentOverlap(Vector2d so, Vector2d eo, Vector2d sp, Vector2d ep) {
    return pm::line_overlap(so(0), so(1), eo(0), eo(1), sp(0), sp(1), ep(0), ep(1));
}

}  // namespace StVO
