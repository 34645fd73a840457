/*
 * stvo_oracle.c — TEST INFRASTRUCTURE ONLY (see stvo_oracle.h for the pinning status).
 *
 * CPU restatement of PL-StVO's per-frame hot path.  Every function cites the reference lines
 * (relative to /root/reference) it follows.  Written from the described behaviour; no reference
 * source is copied.  Scalar C99, one thread, FP64 exactly where the reference uses double and
 * FP32 exactly where it uses float.
 */
#include "stvo_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ============================================================================================
 *  Matching half
 * ========================================================================================== */

/* StVO::distance, src/matching.cpp:93-109: 8 x 32-bit XOR + SWAR popcount. */
int orc_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t wa, wb;
        memcpy(&wa, a + 4 * i, 4);
        memcpy(&wb, b + 4 * i, 4);
        uint32_t v = wa ^ wb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24);
    }
    return dist;
}

static inline int hamming256(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) +
           __builtin_popcountll(x[3] ^ y[3]);
}

/* cv::BFMatcher(NORM_HAMMING,false)::knnMatch(q, t, ., 2) as called at src/matching.cpp:47-48.
 * OpenCV 3.x is NOT under /root/reference (CMakeLists.txt:4 only says "OpenCV 3"); restated from
 * its published behaviour: per query the two smallest Hamming distances in ascending order, the
 * candidate scan runs over ascending train index and inserts on strict '<', so among equal
 * distances the lowest train index comes first.  idx0 = -1 / d = INT_MAX when fewer than 1 / 2
 * train rows exist. */
void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx0, int32_t* d0, int32_t* d1) {
    for (int i = 0; i < nq; ++i) {
        int b0 = INT_MAX, b1 = INT_MAX, bi = -1;
        const uint8_t* qi = q + (size_t)i * STVO_DESC_BYTES;
        for (int j = 0; j < nt; ++j) {
            int d = hamming256(qi, t + (size_t)j * STVO_DESC_BYTES);
            if (d < b0) {
                b1 = b0;
                b0 = d;
                bi = j;
            } else if (d < b1) {
                b1 = d;
            }
        }
        idx0[i] = bi;
        d0[i] = b0;
        d1[i] = b1;
    }
}

/* StVO::matchNNR, src/matching.cpp:41-61.  The ratio test is evaluated in FLOAT
 * (DMatch::distance is float, nnr is float): (float)d0 < (float)d1 * nnr  (:54).
 * Deviation from UB: with fewer than 2 train rows the reference indexes matches_[idx][1] out
 * of range (:54); here such a query simply has no match. */
int orc_match_nnr(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int32_t* m12) {
    int matches = 0;
    for (int i = 0; i < n1; ++i) m12[i] = -1; /* :44 */
    if (n1 <= 0 || n2 < 2) return 0;
    int32_t* idx0 = (int32_t*)malloc(sizeof(int32_t) * (size_t)n1 * 3);
    int32_t* b0 = idx0 + n1;
    int32_t* b1 = b0 + n1;
    orc_knn2(d1, n1, d2, n2, idx0, b0, b1);
    for (int i = 0; i < n1; ++i) {
        volatile float lhs = (float)b0[i];
        volatile float rhs = (float)b1[i] * nnr; /* float multiply, no FMA possible */
        if (lhs < rhs) {
            m12[i] = idx0[i];
            matches++;
        }
    }
    free(idx0);
    return matches;
}

/* StVO::match, src/matching.cpp:63-91: 12 and 21 ratio-tested passes + mutual check (:80-86).
 * The two std::async threads (:69-74) are a fork/join with no shared writes; run serially. */
int orc_match(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int best_lr, int32_t* m12) {
    if (!best_lr) return orc_match_nnr(d1, n1, d2, n2, nnr, m12); /* :89-90 */
    int matches = orc_match_nnr(d1, n1, d2, n2, nnr, m12);
    int32_t* m21 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    orc_match_nnr(d2, n2, d1, n1, nnr, m21);
    for (int i1 = 0; i1 < n1; ++i1) {
        int i2 = m12[i1];
        if (i2 >= 0 && m21[i2] != i1) {
            m12[i1] = -1;
            matches--;
        }
    }
    free(m21);
    return matches;
}

/* GridStructure (src/gridStructure.cpp:43-63): `cols x rows` buckets of std::list<int>, filled by
 * grid.at(x,y).push_back(idx); out-of-range cells go to a sink that get() never returns (:56-63).
 * Restated as CSR with cell = y*cols + x and items in insertion order (stable counting sort). */
void orc_grid_build(const int32_t* cell_xy, const int32_t* owner, int n_entries, int32_t* cell_start,
                    int32_t* cell_items) {
    memset(cell_start, 0, sizeof(int32_t) * (STVO_GRID_CELLS + 1));
    for (int k = 0; k < n_entries; ++k) {
        int x = cell_xy[2 * k], y = cell_xy[2 * k + 1];
        if (x >= 0 && x < STVO_GRID_COLS && y >= 0 && y < STVO_GRID_ROWS) cell_start[y * STVO_GRID_COLS + x + 1]++;
    }
    for (int c = 0; c < STVO_GRID_CELLS; ++c) cell_start[c + 1] += cell_start[c];
    int32_t* fill = (int32_t*)calloc(STVO_GRID_CELLS, sizeof(int32_t));
    for (int k = 0; k < n_entries; ++k) {
        int x = cell_xy[2 * k], y = cell_xy[2 * k + 1];
        if (x >= 0 && x < STVO_GRID_COLS && y >= 0 && y < STVO_GRID_ROWS) {
            int c = y * STVO_GRID_COLS + x;
            cell_items[cell_start[c] + fill[c]++] = owner ? owner[k] : k;
        }
    }
    free(fill);
}

/* LineIterator + getLineCoords, src/lineIterator.cpp:34-77, src/gridStructure.cpp:33-41:
 * Bresenham over double endpoints.  Returns the number of cells written (x,y pairs). */
int orc_line_coords(double x1, double y1, double x2, double y2, int32_t* out_xy, int max_cells) {
    int steep = fabs(y2 - y1) > fabs(x2 - x1); /* :35 */
    double t;
    if (steep) {
        t = x1; x1 = y1; y1 = t;
        t = x2; x2 = y2; y2 = t;
    }
    if (x1 > x2) {
        t = x1; x1 = x2; x2 = t;
        t = y1; y1 = y2; y2 = t;
    }
    double dx = x2 - x1, dy = fabs(y2 - y1);
    double error = dx / 2.0;
    int ystep = (y1 < y2) ? 1 : -1;
    int x = (int)x1, y = (int)y1, maxX = (int)x2;
    int n = 0;
    while (x <= maxX) { /* getNext :58-77 */
        if (n < max_cells) {
            out_xy[2 * n] = steep ? y : x;
            out_xy[2 * n + 1] = steep ? x : y;
        }
        n++;
        error -= dy;
        if (error < 0) {
            y += ystep;
            error += dx;
        }
        x++;
    }
    return n;
}

/* GridStructure::get, src/gridStructure.cpp:65-76: union (std::unordered_set) of the cells of the
 * clamped window.  `seen` de-duplicates; entries touched are reset before returning. */
int orc_grid_window_gather(const int32_t* cell_start, const int32_t* cell_items, int x, int y, const stvo_grid_window* w,
                           int32_t* out, uint8_t* seen) {
    int min_x = x - w->w_lo > 0 ? x - w->w_lo : 0;
    int max_x = x + w->w_hi + 1 < STVO_GRID_COLS ? x + w->w_hi + 1 : STVO_GRID_COLS;
    int min_y = y - w->h_lo > 0 ? y - w->h_lo : 0;
    int max_y = y + w->h_hi + 1 < STVO_GRID_ROWS ? y + w->h_hi + 1 : STVO_GRID_ROWS;
    int n = 0;
    for (int x_ = min_x; x_ < max_x; ++x_)
        for (int y_ = min_y; y_ < max_y; ++y_) {
            int c = y_ * STVO_GRID_COLS + x_;
            for (int k = cell_start[c]; k < cell_start[c + 1]; ++k) {
                int id = cell_items[k];
                if (!seen[id]) {
                    seen[id] = 1;
                    out[n++] = id;
                }
            }
        }
    return n;
}

/* Shared inner loop of both matchGrid overloads (src/matching.cpp:140-163, :217-244). */
static int grid_scan_one(const uint8_t* desc, const int32_t* cand, int ncand, const uint8_t* d2, int n2, const double* v,
                         const double* dir2, double line_sim_th, int best_lr, int32_t* distances, int32_t* m21, int i1,
                         double ratio) {
    int best_d = INT_MAX, best_d2 = INT_MAX, best_idx = -1;
    for (int c = 0; c < ncand; ++c) {
        int i2 = cand[c];
        if (i2 < 0 || i2 >= n2) continue;
        if (v) { /* lines only: direction-cosine gate :221-222; NaN (v = 0/0) never skips */
            double dot = v[0] * dir2[2 * i2] + v[1] * dir2[2 * i2 + 1];
            if (fabs(dot) < line_sim_th) continue;
        }
        int d = orc_distance(desc, d2 + (size_t)i2 * STVO_DESC_BYTES);
        if (best_lr) { /* :145-150 running strict minimum per train row */
            if (d < distances[i2]) {
                distances[i2] = d;
                m21[i2] = i1;
            } else
                continue;
        }
        if (d < best_d) {
            best_d2 = best_d;
            best_d = d;
            best_idx = i2;
        } else if (d < best_d2)
            best_d2 = d;
    }
    /* :160 — DOUBLE ratio test; best_d2 == INT_MAX with a single eligible candidate */
    if ((double)best_d < (double)best_d2 * ratio) return best_idx;
    return -1;
}

/* StVO::matchGrid (points), src/matching.cpp:111-177. */
int orc_match_grid_points(const int32_t* cell_xy1, const uint8_t* d1, int n1, const int32_t* cell_start,
                          const int32_t* cell_items, const uint8_t* d2, int n2, const stvo_grid_window* w, double ratio,
                          int best_lr, int32_t* m12) {
    int matches = 0;
    int32_t* distances = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * n2 + 1));
    int32_t* m21 = distances + n2;
    for (int j = 0; j < n2; ++j) {
        distances[j] = INT_MAX;
        m21[j] = -1;
    }
    int ncells_items = cell_start[STVO_GRID_CELLS];
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ncells_items + 1));
    uint8_t* seen = (uint8_t*)calloc((size_t)(n2 + 1), 1);
    for (int i1 = 0; i1 < n1; ++i1) {
        m12[i1] = -1;
        int nc = orc_grid_window_gather(cell_start, cell_items, cell_xy1[2 * i1], cell_xy1[2 * i1 + 1], w, cand, seen);
        for (int c = 0; c < nc; ++c) seen[cand[c]] = 0;
        if (nc == 0) continue; /* :139 */
        int r = grid_scan_one(d1 + (size_t)i1 * STVO_DESC_BYTES, cand, nc, d2, n2, NULL, NULL, 0.0, best_lr, distances,
                              m21, i1, ratio);
        if (r >= 0) {
            m12[i1] = r;
            matches++;
        }
    }
    if (best_lr) /* :166-174 */
        for (int i1 = 0; i1 < n1; ++i1) {
            int i2 = m12[i1];
            if (i2 >= 0 && m21[i2] != i1) {
                m12[i1] = -1;
                matches--;
            }
        }
    free(seen);
    free(cand);
    free(distances);
    return matches;
}

/* StVO::matchGrid (lines), src/matching.cpp:179-258.  Candidates = union of the windows at both
 * end-point cells (:213-215); direction from INTEGER cell differences (:207-211).  The ratio is
 * Config::minRatio12P() here too (:241) — the caller passes it. */
int orc_match_grid_lines(const int32_t* cell_xy1, const uint8_t* d1, int n1, const int32_t* cell_start,
                         const int32_t* cell_items, const uint8_t* d2, int n2, const double* dir2,
                         const stvo_grid_window* w, double ratio, double line_sim_th, int best_lr, int32_t* m12) {
    int matches = 0;
    int32_t* distances = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * n2 + 1));
    int32_t* m21 = distances + n2;
    for (int j = 0; j < n2; ++j) {
        distances[j] = INT_MAX;
        m21[j] = -1;
    }
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    uint8_t* seen = (uint8_t*)calloc((size_t)(n2 + 1), 1);
    for (int i1 = 0; i1 < n1; ++i1) {
        m12[i1] = -1;
        int sx = cell_xy1[4 * i1], sy = cell_xy1[4 * i1 + 1], ex = cell_xy1[4 * i1 + 2], ey = cell_xy1[4 * i1 + 3];
        double v[2] = {(double)(ex - sx), (double)(ey - sy)};
        double mag = sqrt(v[0] * v[0] + v[1] * v[1]); /* include/matching.h:48-53 */
        v[0] /= mag;
        v[1] /= mag;
        int nc = orc_grid_window_gather(cell_start, cell_items, sx, sy, w, cand, seen);
        nc += orc_grid_window_gather(cell_start, cell_items, ex, ey, w, cand + nc, seen);
        for (int c = 0; c < nc; ++c) seen[cand[c]] = 0;
        if (nc == 0) continue;
        int r = grid_scan_one(d1 + (size_t)i1 * STVO_DESC_BYTES, cand, nc, d2, n2, v, dir2, line_sim_th, best_lr,
                              distances, m21, i1, ratio);
        if (r >= 0) {
            m12[i1] = r;
            matches++;
        }
    }
    if (best_lr)
        for (int i1 = 0; i1 < n1; ++i1) {
            int i2 = m12[i1];
            if (i2 >= 0 && m21[i2] != i1) {
                m12[i1] = -1;
                matches--;
            }
        }
    free(seen);
    free(cand);
    free(distances);
    return matches;
}

/* PointFeature ctor, src/stereoFeatures.cpp:41-47: sigma2 = 1 / (scale^level)^2. */
static double level_sigma2(double sigma2_in, double scale, int level) {
    double s = sigma2_in;
    for (int i = 0; i < level; ++i) s *= scale;
    return 1.0 / (s * s);
}

/* StereoFrame::matchStereoPoints, src/stereoFrame.cpp:120-173.  Returns the number of stereo
 * points kept; row k of the outputs corresponds to left key-point src_idx[k] (the reference keeps
 * pdesc_l row k <-> stereo_pt[k], :161,172). */
int orc_stereo_points(const float* kp_l, const int32_t* oct_l, const uint8_t* desc_l, int nl, const float* kp_r,
                      const uint8_t* desc_r, int nr, int img_cols, int img_rows, const stvo_cam* cam,
                      const stvo_match_params* mp, int32_t* src_idx, double* pl, double* disp, double* P, double* sigma2,
                      int32_t* m12_raw) {
    if (nl <= 0 || nr <= 0) return 0; /* :126-127 */
    double inv_width = STVO_GRID_COLS / (double)img_cols; /* :47-48 */
    double inv_height = STVO_GRID_ROWS / (double)img_rows;
    int32_t* coords = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(nl + nr));
    int32_t* rcells = coords + 2 * nl;
    for (int i = 0; i < nl; ++i) { /* :129-132 float*double -> int truncation */
        coords[2 * i] = (int)((double)kp_l[2 * i] * inv_width);
        coords[2 * i + 1] = (int)((double)kp_l[2 * i + 1] * inv_height);
    }
    for (int i = 0; i < nr; ++i) { /* :135-139 */
        rcells[2 * i] = (int)((double)kp_r[2 * i] * inv_width);
        rcells[2 * i + 1] = (int)((double)kp_r[2 * i + 1] * inv_height);
    }
    int32_t* cell_start = (int32_t*)malloc(sizeof(int32_t) * (STVO_GRID_CELLS + 1 + (size_t)nr));
    int32_t* cell_items = cell_start + STVO_GRID_CELLS + 1;
    orc_grid_build(rcells, NULL, nr, cell_start, cell_items);
    stvo_grid_window w = {mp->matching_s_ws, 0, 0, 0}; /* :141-143 */
    int32_t* m12 = (int32_t*)malloc(sizeof(int32_t) * (size_t)nl);
    orc_match_grid_points(coords, desc_l, nl, cell_start, cell_items, desc_r, nr, &w, mp->min_ratio_12_p_d > 0.0 ? mp->min_ratio_12_p_d : (double)mp->min_ratio_12_p /* Config::minRatio12P(), a double */,
                          mp->best_lr_matches, m12);
    if (m12_raw) memcpy(m12_raw, m12, sizeof(int32_t) * (size_t)nl);
    int k = 0;
    for (int i1 = 0; i1 < nl; ++i1) {
        int i2 = m12[i1];
        if (i2 < 0) continue;
        float dy = kp_l[2 * i1 + 1] - kp_r[2 * i2 + 1]; /* float subtraction :157 */
        if ((double)fabsf(dy) <= mp->max_dist_epip) {
            double disp_ = (double)(kp_l[2 * i1] - kp_r[2 * i2]); /* float subtraction :159 */
            if (disp_ >= mp->min_disp) {
                double u = (double)kp_l[2 * i1], v = (double)kp_l[2 * i1 + 1];
                double bd = cam->b / disp_; /* backProjection, src/pinholeStereoCamera.cpp:221-229 */
                src_idx[k] = i1;
                pl[2 * k] = u;
                pl[2 * k + 1] = v;
                disp[k] = disp_;
                P[3 * k] = bd * (u - cam->cx);
                P[3 * k + 1] = bd * (v - cam->cy);
                P[3 * k + 2] = bd * cam->fx;
                sigma2[k] = level_sigma2(1.0, mp->orb_scale_factor, oct_l[i1]);
                k++;
            }
        }
    }
    free(m12);
    free(cell_start);
    free(coords);
    return k;
}

/* StereoFrame::lineSegmentOverlapStereo, src/stereoFrame.cpp:473-508 (note length = eln - spn). */
double orc_line_overlap_stereo(double spl_obs, double epl_obs, double spl_proj, double epl_proj, double line_horiz_th) {
    double overlap = 1.0;
    if (fabs(epl_obs - spl_obs) > line_horiz_th) {
        double sln = fmin(spl_obs, epl_obs), eln = fmax(spl_obs, epl_obs);
        double spn = fmin(spl_proj, epl_proj), epn = fmax(spl_proj, epl_proj);
        double length = eln - spn;
        if (epn < sln || spn > eln)
            overlap = 0.0;
        else if (epn > eln && spn < sln)
            overlap = eln - sln;
        else
            overlap = fmin(eln, epn) - fmax(sln, spn);
        if (length > (double)0.01f)
            overlap = overlap / length;
        else
            overlap = 0.0;
        if (overlap > 1.0) overlap = 1.0;
    }
    return overlap;
}

/* StereoFrame::matchStereoLines, src/stereoFrame.cpp:309-398 (+ filterLineSegmentDisparity :405-415). */
int orc_stereo_lines(const float* kl_l, const float* angle_l, const int32_t* oct_l, const uint8_t* desc_l, int nl,
                     const float* kl_r, const uint8_t* desc_r, int nr, int img_cols, int img_rows, const stvo_cam* cam,
                     const stvo_match_params* mp, int32_t* src_idx, double* spl, double* epl, double* sdisp,
                     double* edisp, double* sP, double* eP, double* le, double* sigma2, int32_t* m12_raw) {
    (void)angle_l;
    if (nl <= 0 || nr <= 0) return 0; /* :315-316 */
    double inv_width = STVO_GRID_COLS / (double)img_cols;
    double inv_height = STVO_GRID_ROWS / (double)img_rows;
    int32_t* coords = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)nl);
    for (int i = 0; i < nl; ++i) { /* :318-322 */
        coords[4 * i] = (int)((double)kl_l[4 * i] * inv_width);
        coords[4 * i + 1] = (int)((double)kl_l[4 * i + 1] * inv_height);
        coords[4 * i + 2] = (int)((double)kl_l[4 * i + 2] * inv_width);
        coords[4 * i + 3] = (int)((double)kl_l[4 * i + 3] * inv_height);
    }
    /* :325-338 rasterise every right line into the grid, unit directions in scaled space */
    const int max_per_line = STVO_GRID_COLS + STVO_GRID_ROWS + 8;
    int cap = nr * max_per_line;
    int32_t* ent_xy = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)cap);
    int32_t* ent_owner = ent_xy + 2 * (size_t)cap;
    double* dir2 = (double*)malloc(sizeof(double) * 2 * (size_t)nr);
    int ne = 0;
    for (int idx = 0; idx < nr; ++idx) {
        const float* kl = kl_r + 4 * idx;
        double vx = (double)(kl[2] - kl[0]) * inv_width; /* float subtraction, then * double (:331) */
        double vy = (double)(kl[3] - kl[1]) * inv_height;
        double mag = sqrt(vx * vx + vy * vy);
        dir2[2 * idx] = vx / mag;
        dir2[2 * idx + 1] = vy / mag;
        int32_t tmp[2 * (STVO_GRID_COLS + STVO_GRID_ROWS + 8)];
        int n = orc_line_coords((double)kl[0] * inv_width, (double)kl[1] * inv_height, (double)kl[2] * inv_width,
                                (double)kl[3] * inv_height, tmp, max_per_line);
        if (n > max_per_line) n = max_per_line; /* cannot happen for in-image lines */
        for (int c = 0; c < n; ++c) {
            ent_xy[2 * ne] = tmp[2 * c];
            ent_xy[2 * ne + 1] = tmp[2 * c + 1];
            ent_owner[ne] = idx;
            ne++;
        }
    }
    int32_t* cell_start = (int32_t*)malloc(sizeof(int32_t) * (STVO_GRID_CELLS + 1 + (size_t)ne + 1));
    int32_t* cell_items = cell_start + STVO_GRID_CELLS + 1;
    orc_grid_build(ent_xy, ent_owner, ne, cell_start, cell_items);
    stvo_grid_window w = {mp->matching_s_ws, 0, 0, 0}; /* :340-342 */
    int32_t* m12 = (int32_t*)malloc(sizeof(int32_t) * (size_t)nl);
    orc_match_grid_lines(coords, desc_l, nl, cell_start, cell_items, desc_r, nr, dir2, &w, mp->min_ratio_12_p_d > 0.0 ? mp->min_ratio_12_p_d : (double)mp->min_ratio_12_p /* Config::minRatio12P(), a double */,
                         mp->line_sim_th, mp->best_lr_matches, m12);
    if (m12_raw) memcpy(m12_raw, m12, sizeof(int32_t) * (size_t)nl);
    int k = 0;
    for (int i1 = 0; i1 < nl; ++i1) {
        int i2 = m12[i1];
        if (i2 < 0) continue;
        double sp_l[3] = {(double)kl_l[4 * i1], (double)kl_l[4 * i1 + 1], 1.0}; /* :353-354 */
        double ep_l[3] = {(double)kl_l[4 * i1 + 2], (double)kl_l[4 * i1 + 3], 1.0};
        double le_l[3] = {sp_l[1] * ep_l[2] - sp_l[2] * ep_l[1], sp_l[2] * ep_l[0] - sp_l[0] * ep_l[2],
                          sp_l[0] * ep_l[1] - sp_l[1] * ep_l[0]};
        double nrm = sqrt(le_l[0] * le_l[0] + le_l[1] * le_l[1]); /* :355 */
        le_l[0] /= nrm;
        le_l[1] /= nrm;
        le_l[2] /= nrm;
        double sp_r[2] = {(double)kl_r[4 * i2], (double)kl_r[4 * i2 + 1]};
        double ep_r[2] = {(double)kl_r[4 * i2 + 2], (double)kl_r[4 * i2 + 3]};
        double overlap = orc_line_overlap_stereo(sp_l[1], ep_l[1], sp_r[1], ep_r[1], mp->line_horiz_th); /* :360 */
        /* :363 — sp_r re-intersected at the left start row */
        double spx = (sp_r[0] * (sp_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - sp_l[1])) / (sp_r[1] - ep_r[1]);
        sp_r[0] = spx;
        sp_r[1] = sp_l[1];
        /* :364 — uses the ALREADY OVERWRITTEN sp_r (quirk) */
        double epx = (sp_r[0] * (ep_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - ep_l[1])) / (sp_r[1] - ep_r[1]);
        ep_r[0] = epx;
        ep_r[1] = ep_l[1];
        double disp_s = sp_l[0] - sp_r[0], disp_e = ep_l[0] - ep_r[0]; /* :407-408 */
        if (fmin(disp_s, disp_e) / fmax(disp_s, disp_e) < mp->ls_min_disp_ratio) {
            disp_s = -1.0;
            disp_e = -1.0;
        }
        if (disp_s >= mp->min_disp && disp_e >= mp->min_disp && fabs(sp_l[1] - ep_l[1]) > mp->line_horiz_th &&
            fabs(sp_r[1] - ep_r[1]) > mp->line_horiz_th && overlap > mp->stereo_overlap_th) { /* :368-371 */
            double bds = cam->b / disp_s, bde = cam->b / disp_e;
            src_idx[k] = i1;
            spl[2 * k] = sp_l[0];
            spl[2 * k + 1] = sp_l[1];
            epl[2 * k] = ep_l[0];
            epl[2 * k + 1] = ep_l[1];
            sdisp[k] = disp_s;
            edisp[k] = disp_e;
            sP[3 * k] = bds * (sp_l[0] - cam->cx);
            sP[3 * k + 1] = bds * (sp_l[1] - cam->cy);
            sP[3 * k + 2] = bds * cam->fx;
            eP[3 * k] = bde * (ep_l[0] - cam->cx);
            eP[3 * k + 1] = bde * (ep_l[1] - cam->cy);
            eP[3 * k + 2] = bde * cam->fx;
            le[3 * k] = le_l[0];
            le[3 * k + 1] = le_l[1];
            le[3 * k + 2] = le_l[2];
            sigma2[k] = level_sigma2(1.0, mp->lsd_scale, oct_l[i1]); /* src/stereoFeatures.cpp:107-115 */
            k++;
        }
    }
    free(m12);
    free(cell_start);
    free(dir2);
    free(ent_xy);
    free(coords);
    return k;
}

/* ============================================================================================
 *  Optimizer half (FP64)
 * ========================================================================================== */

#define M4(T, r, c) (T)[(r)*4 + (c)]
#define M6(A, r, c) (A)[(r)*6 + (c)]

static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

/* skew, src/auxiliar.cpp:29-44 */
static void skew3(const double v[3], double S[9]) {
    S[0] = 0; S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2]; S[4] = 0; S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}

/* inverse_se3, src/auxiliar.cpp:113-122 */
void orc_inverse_se3(const double T[16], double Ti[16]) {
    double out[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M4(out, i, j) = M4(T, j, i);
    for (int i = 0; i < 3; ++i)
        M4(out, i, 3) = -(M4(T, 0, i) * M4(T, 0, 3) + M4(T, 1, i) * M4(T, 1, 3) + M4(T, 2, i) * M4(T, 2, 3));
    memcpy(Ti, out, sizeof(out));
}

/* expmap_se3, src/auxiliar.cpp:124-141: theta < 1e-6 leaves R = I and t un-multiplied. */
void orc_expmap_se3(const double x[6], double T[16]) {
    double t[3] = {x[0], x[1], x[2]}, w[3] = {x[3], x[4], x[5]};
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (!(theta < 0.000001)) {
        double s[9], s2[9], V[9], sk[9];
        skew3(w, sk);
        for (int i = 0; i < 9; ++i) s[i] = sk[i] / theta;
        mat3_mul(s, s, s2);
        double sn = sin(theta), cs = cos(theta);
        for (int i = 0; i < 9; ++i) {
            double I = (i % 4 == 0) ? 1.0 : 0.0;
            R[i] = I + s[i] * sn + s2[i] * (1.0 - cs);
            V[i] = I + s[i] * (1.0 - cs) / theta + s2[i] * (theta - sn) / theta;
        }
        double tn[3];
        for (int i = 0; i < 3; ++i) tn[i] = V[i * 3] * t[0] + V[i * 3 + 1] * t[1] + V[i * 3 + 2] * t[2];
        t[0] = tn[0]; t[1] = tn[1]; t[2] = tn[2];
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) M4(T, i, j) = R[i * 3 + j];
        M4(T, i, 3) = t[i];
    }
    M4(T, 3, 0) = 0; M4(T, 3, 1) = 0; M4(T, 3, 2) = 0; M4(T, 3, 3) = 1;
}

/* Matrix3d::inverse(): Eigen 3 (absent) uses the cofactor formula for fixed 3x3. */
static void inverse3(const double A[9], double Ai[9]) {
    double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    double id = 1.0 / det;
    Ai[0] = c00 * id;
    Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id;
    Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = c01 * id;
    Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id;
    Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = c02 * id;
    Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

/* logmap_se3, src/auxiliar.cpp:143-173 */
void orc_logmap_se3(const double T[16], double x[6]) {
    double R[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, w[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = M4(T, i, j);
    double Vt[3] = {M4(T, 0, 3), M4(T, 1, 3), M4(T, 2, 3)};
    double cosine = (R[0] + R[4] + R[8] - 1.0) / 2.0;
    if (cosine > 1.0) cosine = 1.0;
    else if (cosine < -1.0) cosine = -1.0;
    double sine = sqrt(1.0 - cosine * cosine);
    if (sine > 1.0) sine = 1.0;
    else if (sine < -1.0) sine = -1.0;
    double theta = acos(cosine);
    if (theta > 0.000001) {
        double what[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) what[i * 3 + j] = theta * (R[i * 3 + j] - R[j * 3 + i]) / (2.0 * sine);
        w[0] = what[7]; /* skewcoords: M(2,1), M(0,2), M(1,0)  (src/auxiliar.cpp:58-62) */
        w[1] = what[2];
        w[2] = what[3];
        double sk[9], s[9], s2[9];
        skew3(w, sk);
        for (int i = 0; i < 9; ++i) s[i] = sk[i] / theta;
        mat3_mul(s, s, s2);
        for (int i = 0; i < 9; ++i) {
            double I = (i % 4 == 0) ? 1.0 : 0.0;
            V[i] = I + s[i] * (1.0 - cosine) / theta + s2[i] * (theta - sine) / theta;
        }
    }
    double Vi[9];
    inverse3(V, Vi);
    for (int i = 0; i < 3; ++i) x[i] = Vi[i * 3] * Vt[0] + Vi[i * 3 + 1] * Vt[1] + Vi[i * 3 + 2] * Vt[2];
    x[3] = w[0]; x[4] = w[1]; x[5] = w[2];
}

/* adjoint_se3, src/auxiliar.cpp:175-182: [R, [t]x R; 0, R] */
void orc_adjoint_se3(const double T[16], double A[36]) {
    double R[9], t[3] = {M4(T, 0, 3), M4(T, 1, 3), M4(T, 2, 3)}, sk[9], skR[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = M4(T, i, j);
    skew3(t, sk);
    mat3_mul(sk, R, skR);
    memset(A, 0, sizeof(double) * 36);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            M6(A, i, j) = R[i * 3 + j];
            M6(A, i, j + 3) = skR[i * 3 + j];
            M6(A, i + 3, j + 3) = R[i * 3 + j];
        }
}

/* unccomp_se3, src/auxiliar.cpp:192-197: cov1 + Ad(T1) covinc Ad(T1)^T */
void orc_unccomp_se3(const double T1[16], const double cov1[36], const double covinc[36], double out[36]) {
    double A[36], tmp[36];
    orc_adjoint_se3(T1, A);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += M6(A, i, k) * M6(covinc, k, j);
            M6(tmp, i, j) = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += M6(tmp, i, k) * M6(A, j, k);
            M6(out, i, j) = M6(cov1, i, j) + s;
        }
}

/* Eigen::ColPivHouseholderQR<Matrix6d>(H).solve(g) and logAbsDeterminant()
 * (call sites src/stereoFrameHandler.cpp:417-418,453-455,507-508,526-527).  Eigen 3 is absent;
 * restated from the published algorithm: Householder QR with column pivoting on the largest
 * remaining column norm, rank = number of pivots before the remaining norms vanish at machine
 * precision, minimum-basic solution with the free components set to zero.  Returns the rank. */
int orc_solve6(const double H[36], const double g[6], double x[6], double* log_abs_det) {
    const int n = 6;
    double A[36], c[6], tau[6];
    int perm[6];
    memcpy(A, H, sizeof(A));
    memcpy(c, g, sizeof(c));
    for (int j = 0; j < n; ++j) perm[j] = j;
    double maxnorm = 0;
    for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int i = 0; i < n; ++i) s += M6(A, i, j) * M6(A, i, j);
        if (sqrt(s) > maxnorm) maxnorm = sqrt(s);
    }
    const double eps = 2.220446049250313e-16;
    double thr_helper = (maxnorm * eps) * (maxnorm * eps) / (double)n;
    int rank = n;
    for (int k = 0; k < n; ++k) {
        int big = k;
        double bigsq = -1;
        for (int j = k; j < n; ++j) {
            double s = 0;
            for (int i = k; i < n; ++i) s += M6(A, i, j) * M6(A, i, j);
            if (s > bigsq) {
                bigsq = s;
                big = j;
            }
        }
        if (rank == n && bigsq < thr_helper * (double)(n - k)) rank = k;
        if (big != k) {
            for (int i = 0; i < n; ++i) {
                double t = M6(A, i, k);
                M6(A, i, k) = M6(A, i, big);
                M6(A, i, big) = t;
            }
            int tp = perm[k];
            perm[k] = perm[big];
            perm[big] = tp;
        }
        /* Householder reflector for column k, rows k.. */
        double c0 = M6(A, k, k), tail = 0;
        for (int i = k + 1; i < n; ++i) tail += M6(A, i, k) * M6(A, i, k);
        double beta;
        if (tail <= 2.2250738585072014e-308) {
            tau[k] = 0;
            beta = c0;
            for (int i = k + 1; i < n; ++i) M6(A, i, k) = 0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0) beta = -beta;
            for (int i = k + 1; i < n; ++i) M6(A, i, k) /= (c0 - beta);
            tau[k] = (beta - c0) / beta;
        }
        M6(A, k, k) = beta;
        /* apply (I - tau v v^T) to trailing columns and to the rhs */
        for (int j = k + 1; j < n; ++j) {
            double s = M6(A, k, j);
            for (int i = k + 1; i < n; ++i) s += M6(A, i, k) * M6(A, i, j);
            s *= tau[k];
            M6(A, k, j) -= s;
            for (int i = k + 1; i < n; ++i) M6(A, i, j) -= s * M6(A, i, k);
        }
        {
            double s = c[k];
            for (int i = k + 1; i < n; ++i) s += M6(A, i, k) * c[i];
            s *= tau[k];
            c[k] -= s;
            for (int i = k + 1; i < n; ++i) c[i] -= s * M6(A, i, k);
        }
    }
    if (log_abs_det) {
        double l = 0;
        for (int i = 0; i < n; ++i) l += log(fabs(M6(A, i, i)));
        *log_abs_det = l;
    }
    double y[6] = {0, 0, 0, 0, 0, 0};
    for (int i = rank - 1; i >= 0; --i) {
        double s = c[i];
        for (int j = i + 1; j < rank; ++j) s -= M6(A, i, j) * y[j];
        y[i] = s / M6(A, i, i);
    }
    for (int i = 0; i < n; ++i) x[perm[i]] = (i < rank) ? y[i] : 0.0;
    return rank;
}

/* Matrix6d::inverse() (src/stereoFrameHandler.cpp:429,470,545): Eigen 3 (absent) uses
 * PartialPivLU for fixed sizes > 4; restated as LU with partial pivoting + 6 solves. */
void orc_inverse6(const double Ain[36], double Ai[36]) {
    const int n = 6;
    double A[36];
    int piv[6];
    memcpy(A, Ain, sizeof(A));
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(M6(A, k, k));
        for (int i = k + 1; i < n; ++i)
            if (fabs(M6(A, i, k)) > best) {
                best = fabs(M6(A, i, k));
                p = i;
            }
        if (p != k) {
            for (int j = 0; j < n; ++j) {
                double t = M6(A, k, j);
                M6(A, k, j) = M6(A, p, j);
                M6(A, p, j) = t;
            }
            int t = piv[k];
            piv[k] = piv[p];
            piv[p] = t;
        }
        for (int i = k + 1; i < n; ++i) {
            M6(A, i, k) /= M6(A, k, k);
            for (int j = k + 1; j < n; ++j) M6(A, i, j) -= M6(A, i, k) * M6(A, k, j);
        }
    }
    for (int col = 0; col < n; ++col) {
        double y[6];
        for (int i = 0; i < n; ++i) {
            double s = (piv[i] == col) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) s -= M6(A, i, j) * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < n; ++j) s -= M6(A, i, j) * y[j];
            y[i] = s / M6(A, i, i);
        }
        for (int i = 0; i < n; ++i) M6(Ai, i, col) = y[i];
    }
}

/* SelfAdjointEigenSolver<Matrix6d>(A).eigenvalues() (src/stereoFrameHandler.cpp:294-295,379-380):
 * Eigen 3 (absent) reads the LOWER triangle and returns ascending eigenvalues; restated with
 * cyclic Jacobi rotations on the symmetrised lower triangle. */
void orc_eig6(const double Ain[36], double w[6]) {
    const int n = 6;
    double A[36];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) M6(A, i, j) = (i >= j) ? M6(Ain, i, j) : M6(Ain, j, i);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                if (i != j) off += M6(A, i, j) * M6(A, i, j);
                else diag += M6(A, i, j) * M6(A, i, j);
            }
        if (!(off > 1e-32 * diag) || !(off > 0)) break; /* also exits on NaN */
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = M6(A, p, q);
                if (apq == 0.0) continue;
                double theta = (M6(A, q, q) - M6(A, p, p)) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) {
                    double akp = M6(A, k, p), akq = M6(A, k, q);
                    M6(A, k, p) = cs * akp - sn * akq;
                    M6(A, k, q) = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = M6(A, p, k), aqk = M6(A, q, k);
                    M6(A, p, k) = cs * apk - sn * aqk;
                    M6(A, q, k) = sn * apk + cs * aqk;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = M6(A, i, i);
    for (int i = 1; i < n; ++i) { /* ascending insertion sort; NaNs stay where they are */
        double v = w[i];
        int j = i - 1;
        while (j >= 0 && w[j] > v) {
            w[j + 1] = w[j];
            --j;
        }
        w[j + 1] = v;
    }
}

static int cmp_double(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

/* vector_mean_stdv_mad, src/auxiliar.cpp:387-430 */
void orc_mean_stdv_mad(const double* r, int n, double* mean, double* stdv) {
    *mean = 0.0;
    *stdv = 0.0;
    if (n == 0) return;
    double* s = (double*)malloc(sizeof(double) * (size_t)n);
    memcpy(s, r, sizeof(double) * (size_t)n);
    qsort(s, (size_t)n, sizeof(double), cmp_double);
    double median = s[n / 2];
    for (int i = 0; i < n; ++i) s[i] = (double)fabsf((float)(s[i] - median)); /* fabsf: FLOAT truncation (:401) */
    qsort(s, (size_t)n, sizeof(double), cmp_double);
    *stdv = 1.4826 * s[n / 2];
    free(s);
    int k = 0;
    double m = 0.0;
    for (int i = 0; i < n; ++i)
        if (r[i] < 2.0 * (*stdv)) {
            m += r[i];
            k++;
        }
    if (k >= (int)(0.2 * (double)n))
        m /= (double)k;
    else {
        k = 0;
        m = 0.0;
        for (int i = 0; i < n; ++i) {
            m += r[i];
            k++;
        }
        m /= (double)k;
    }
    *mean = m;
}

/* vector_stdv_mad(vector<double>), src/auxiliar.cpp:444-460 */
double orc_stdv_mad(const double* r, int n) {
    if (n == 0) return 0.0;
    double* s = (double*)malloc(sizeof(double) * (size_t)n);
    memcpy(s, r, sizeof(double) * (size_t)n);
    qsort(s, (size_t)n, sizeof(double), cmp_double);
    double median = s[n / 2];
    for (int i = 0; i < n; ++i) s[i] = (double)fabsf((float)(s[i] - median));
    qsort(s, (size_t)n, sizeof(double), cmp_double);
    double mad = s[n / 2];
    free(s);
    return 1.4826 * mad;
}

static double overlap_from_lambdas(double ls, double le) {
    double lmin = fmin(ls, le), lmax = fmax(ls, le);
    if (lmin < 0.0 && lmax > 1.0) return 1.0;
    if (lmax < 0.0 || lmin > 1.0) return 0.0;
    if (lmin < 0.0) return lmax;
    if (lmax > 1.0) return 1.0 - lmin;
    return lmax - lmin;
}

/* StereoFrame::lineSegmentOverlap, src/stereoFrame.cpp:510-616 */
double orc_line_overlap(const double so[2], const double eo[2], const double sp[2], const double ep[2]) {
    double lx = eo[0] - so[0], ly = eo[1] - so[1];
    if (fabs(so[0] - eo[0]) < 1.0) { /* vertical :515-544 */
        return overlap_from_lambdas((sp[1] - so[1]) / ly, (ep[1] - so[1]) / ly);
    } else if (fabs(so[1] - eo[1]) < 1.0) { /* horizontal :545-574 */
        return overlap_from_lambdas((sp[0] - so[0]) / lx, (ep[0] - so[0]) / lx);
    } else { /* general :575-612: foot points of sp/ep on the observed line */
        double a = so[1] - eo[1], b = eo[0] - so[0], c = so[0] * eo[1] - eo[0] * so[1];
        double lxy = 1.0 / (a * a + b * b);
        double spx = (b * (b * sp[0] - a * sp[1]) - a * c) * lxy;
        double epx = (b * (b * ep[0] - a * ep[1]) - a * c) * lxy;
        return overlap_from_lambdas((spx - so[0]) / lx, (epx - so[0]) / lx);
    }
}

static int all_finite16(const double* T) {
    for (int i = 0; i < 16; ++i)
        if (!isfinite(T[i])) return 0;
    return 1;
}

/* isGoodSolution, src/stereoFrameHandler.cpp:292-305 */
int orc_is_good_solution(const double DT[16], const double cov[36], double err) {
    double w[6];
    orc_eig6(cov, w);
    if (w[0] < 0.0 || w[5] > 1.0 || err < 0.0 || err > 1.0 || !all_finite16(DT)) return 0;
    return 1;
}

/* The 1x6 gradient of the scalar residual shared by points and line end points
 * (src/stereoFrameHandler.cpp:577-586, 630-653): translation first, rotation last; only fx. */
static void grad6(const double P[3], double dx, double dy, double fx, double homog_th, double J[6]) {
    double gx = P[0], gy = P[1], gz = P[2];
    double gz2 = gz * gz;
    double fgz2 = fx / fmax(homog_th, gz2);
    J[0] = +fgz2 * dx * gz;
    J[1] = +fgz2 * dy * gz;
    J[2] = -fgz2 * (gx * dx + gy * dy);
    J[3] = -fgz2 * (gx * gy * dx + gy * gy * dy + gz * gz * dy);
    J[4] = +fgz2 * (gx * gx * dx + gz * gz * dx + gx * gy * dy);
    J[5] = +fgz2 * (gx * gz * dy - gy * gz * dx);
}

static void transform_project(const double DT[16], const double P[3], const stvo_cam* cam, double Pc[3], double uv[2]) {
    for (int i = 0; i < 3; ++i) Pc[i] = M4(DT, i, 0) * P[0] + M4(DT, i, 1) * P[1] + M4(DT, i, 2) * P[2] + M4(DT, i, 3);
    uv[0] = cam->cx + cam->fx * Pc[0] / Pc[2]; /* projection, src/pinholeStereoCamera.cpp:231-237 */
    uv[1] = cam->cy + cam->fy * Pc[1] / Pc[2];
}

static void accumulate(double H[36], double g[6], double* e, const double J[6], double r, double w) {
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) M6(H, i, j) += J[i] * J[j] * w;
        g[i] += J[i] * r * w;
    }
    *e += r * r * w;
}

static double clamp_scale(double s) {
    const double th_min = 0.0001, th_max = sqrt(7.815); /* :744-745 */
    if (s < th_min) s = th_min;
    if (s > th_max) s = th_max;
    return s;
}

/* optimizeFunctions (robust = 0, src/stereoFrameHandler.cpp:549-694) and
 * optimizeFunctionsRobust (robust = 1, :696-962). */
void orc_optimize_functions(const double DT[16], const stvo_cam* cam, const stvo_opt_params* p, const orc_matched* m,
                            int robust, double H[36], double g[6], double* e, int32_t* n_used) {
    double Hp[36] = {0}, Hl[36] = {0}, gp[6] = {0}, gl[6] = {0}, ep = 0.0, el = 0.0;
    double s_p = 1.0, s_l = 1.0;
    if (robust) { /* pre-pass :710-781 */
        double* res = (double*)malloc(sizeof(double) * (size_t)(m->np + m->nl + 1));
        int k = 0;
        for (int i = 0; i < m->np; ++i)
            if (m->inlier_p[i]) {
                double Pc[3], uv[2];
                transform_project(DT, m->P + 3 * i, cam, Pc, uv);
                double dx = uv[0] - m->pl_obs[2 * i], dy = uv[1] - m->pl_obs[2 * i + 1];
                res[k++] = sqrt(dx * dx + dy * dy);
            }
        s_p = clamp_scale(orc_stdv_mad(res, k));
        k = 0;
        for (int i = 0; i < m->nl; ++i)
            if (m->inlier_l[i]) {
                double Pc[3], s[2], t[2];
                transform_project(DT, m->sP + 3 * i, cam, Pc, s);
                transform_project(DT, m->eP + 3 * i, cam, Pc, t);
                const double* l = m->le_obs + 3 * i;
                double ds = l[0] * s[0] + l[1] * s[1] + l[2], de = l[0] * t[0] + l[1] * t[1] + l[2];
                res[k++] = sqrt(ds * ds + de * de);
            }
        s_l = clamp_scale(orc_stdv_mad(res, k));
        free(res);
    }
    int N_p = 0, N_l = 0;
    for (int i = 0; i < m->np; ++i) { /* :563-606 / :785-874 */
        if (!m->inlier_p[i]) continue;
        double Pc[3], uv[2], J[6];
        transform_project(DT, m->P + 3 * i, cam, Pc, uv);
        double dx = uv[0] - m->pl_obs[2 * i], dy = uv[1] - m->pl_obs[2 * i + 1];
        double nrm = sqrt(dx * dx + dy * dy);
        grad6(Pc, dx, dy, cam->fx, p->homog_th, J);
        double den = fmax(p->homog_th, nrm);
        for (int k = 0; k < 6; ++k) J[k] = J[k] / den;
        double r, w;
        if (!robust) {
            r = nrm * sqrt(m->sigma2p[i]); /* :590-591 */
            w = 1.0 / (1.0 + r * r);       /* robustWeightCauchy, src/auxiliar.cpp:556-560 */
        } else {
            r = nrm; /* :813 */
            double x = r / s_p;
            w = 1.0 / (1.0 + x * x);
        }
        accumulate(Hp, gp, &ep, J, r, w);
        N_p++;
    }
    for (int i = 0; i < m->nl; ++i) { /* :610-684 / :878-952 */
        if (!m->inlier_l[i]) continue;
        double sPc[3], ePc[3], s[2], t[2], Js[6], Je[6], J[6];
        transform_project(DT, m->sP + 3 * i, cam, sPc, s);
        transform_project(DT, m->eP + 3 * i, cam, ePc, t);
        const double* l = m->le_obs + 3 * i;
        double ds = l[0] * s[0] + l[1] * s[1] + l[2], de = l[0] * t[0] + l[1] * t[1] + l[2];
        double nrm = sqrt(ds * ds + de * de);
        grad6(sPc, l[0], l[1], cam->fx, p->homog_th, Js);
        grad6(ePc, l[0], l[1], cam->fx, p->homog_th, Je);
        double den = fmax(p->homog_th, nrm);
        for (int k = 0; k < 6; ++k) J[k] = (Js[k] * ds + Je[k] * de) / den;
        double r, w;
        if (!robust) {
            r = nrm * sqrt(m->sigma2l[i]);
            w = 1.0 / (1.0 + r * r);
        } else {
            r = nrm;
            double x = r / s_l;
            w = 1.0 / (1.0 + x * x);
        }
        w *= orc_line_overlap(m->spl + 2 * i, m->epl + 2 * i, s, t); /* :668 prev-frame end points */
        accumulate(Hl, gl, &el, J, r, w);
        N_l++;
    }
    for (int i = 0; i < 36; ++i) H[i] = Hp[i] + Hl[i];
    for (int i = 0; i < 6; ++i) g[i] = gp[i] + gl[i];
    *e = (ep + el) / (double)(N_l + N_p); /* :692 — 0/0 = NaN when nothing is an inlier */
    if (n_used) *n_used = N_p + N_l;
}

static void step_pose(double DT[16], const double inc[6]) {
    /* DT = DT * inverse_se3(expmap_se3(inc))   (:419) */
    double E[16], Ei[16], out[16];
    orc_expmap_se3(inc, E);
    orc_inverse_se3(E, Ei);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += M4(DT, i, k) * M4(Ei, k, j);
            M4(out, i, j) = s;
        }
    memcpy(DT, out, sizeof(out));
}

static double norm3(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* gaussNewtonOptimization, src/stereoFrameHandler.cpp:394-431.  Returns #evaluations. */
static int gauss_newton(double DT[16], double cov[36], double* err_, int max_iters, const stvo_cam* cam,
                        const stvo_opt_params* p, const orc_matched* m) {
    double H[36] = {0}, g[6], inc[6], err = 0.0, err_prev = 999999999.9;
    int evals = 0;
    for (int iters = 0; iters < max_iters; ++iters) {
        orc_optimize_functions(DT, cam, p, m, 0, H, g, &err, NULL);
        evals++;
        if (err > err_prev) {
            if (iters > 0) break;
            *err_ = -1.0;
            return evals;
        }
        if ((err < p->min_error) || fabs(err - err_prev) < p->min_error_change) break;
        orc_solve6(H, g, inc, NULL);
        step_pose(DT, inc);
        if (norm3(inc) < p->min_error_change && norm3(inc + 3) < p->min_error_change) break;
        err_prev = err;
    }
    orc_inverse6(H, cov); /* :429 — H of the last evaluation (zeros if max_iters <= 0) */
    *err_ = err;
    return evals;
}

/* gaussNewtonOptimizationRobust, src/stereoFrameHandler.cpp:433-480 */
static int gauss_newton_robust(double DT[16], double cov[36], double* err_, int max_iters, const stvo_cam* cam,
                               const stvo_opt_params* p, const orc_matched* m) {
    double DT0[16], H[36] = {0}, g[6], inc[6], err = 0.0, err_prev = 999999999.9;
    int good = 1, evals = 0;
    memcpy(DT0, DT, sizeof(DT0));
    for (int iters = 0; iters < max_iters; ++iters) {
        orc_optimize_functions(DT, cam, p, m, 1, H, g, &err, NULL);
        evals++;
        if (fabs(err - err_prev) < p->min_error_change || err < p->min_error) break;
        double lad;
        orc_solve6(H, g, inc, &lad);
        if (lad < 0.0) { /* :455 (info() is always Success for this decomposition) */
            good = 0;
            break;
        }
        step_pose(DT, inc);
        double n6 = 0.0;
        for (int i = 0; i < 6; ++i) n6 += inc[i] * inc[i];
        if (sqrt(n6) < p->min_error_change) break; /* DT_inc.norm() :462 */
        err_prev = err;
    }
    if (good) {
        orc_inverse6(H, cov);
        *err_ = err;
    } else {
        memcpy(DT, DT0, sizeof(DT0));
        *err_ = -1.0;
        memset(cov, 0, sizeof(double) * 36);
        for (int i = 0; i < 6; ++i) M6(cov, i, i) = 1.0;
    }
    return evals;
}

/* levenbergMarquardtOptimization, src/stereoFrameHandler.cpp:482-547 (inverted lambda schedule
 * replicated: err > err_prev => lambda /= 4 and NO step, else lambda *= 4 and step). */
static int levenberg_marquardt(double DT[16], double cov[36], double* err_, int max_iters, const stvo_cam* cam,
                               const stvo_opt_params* p, const orc_matched* m) {
    double H[36], g[6], inc[6], err, err_prev;
    double lambda = 0.000000001, lambda_k = 4.0;
    int evals = 1;
    orc_optimize_functions(DT, cam, p, m, 0, H, g, &err, NULL);
    double Hmax = 0.0;
    for (int i = 0; i < 6; ++i)
        if (M6(H, i, i) > Hmax || M6(H, i, i) < -Hmax) Hmax = fabs(M6(H, i, i));
    lambda *= Hmax;
    for (int i = 0; i < 6; ++i) M6(H, i, i) += lambda;
    orc_solve6(H, g, inc, NULL);
    step_pose(DT, inc);
    err_prev = err;
    for (int iters = 1; iters < max_iters; ++iters) {
        orc_optimize_functions(DT, cam, p, m, 0, H, g, &err, NULL);
        evals++;
        if (fabs(err - err_prev) < p->min_error_change || err < p->min_error) break;
        for (int i = 0; i < 6; ++i) M6(H, i, i) += lambda;
        orc_solve6(H, g, inc, NULL);
        if (err > err_prev)
            lambda /= lambda_k;
        else {
            lambda *= lambda_k;
            step_pose(DT, inc);
        }
        if (norm3(inc) < p->min_error_change && norm3(inc + 3) < p->min_error_change) break;
        err_prev = err;
    }
    orc_inverse6(H, cov);
    *err_ = err;
    return evals;
}

static int run_mode(int mode, double DT[16], double cov[36], double* err, int iters, const stvo_cam* cam,
                    const stvo_opt_params* p, const orc_matched* m) {
    if (mode == 1) return gauss_newton_robust(DT, cov, err, iters, cam, p, m);
    if (mode == 2) return levenberg_marquardt(DT, cov, err, iters, cam, p, m);
    return gauss_newton(DT, cov, err, iters, cam, p, m);
}

/* removeOutliers, src/stereoFrameHandler.cpp:988-1067 */
void orc_remove_outliers(const double DT[16], const stvo_cam* cam, const stvo_opt_params* p, orc_matched* m,
                         int32_t* n_inl_pt, int32_t* n_inl_ls) {
    if (p->has_points) {
        double* res = (double*)malloc(sizeof(double) * (size_t)(m->np + 1));
        for (int i = 0; i < m->np; ++i) { /* ALL matches, including current outliers (:998-1005) */
            double Pc[3], uv[2];
            transform_project(DT, m->P + 3 * i, cam, Pc, uv);
            double dx = uv[0] - m->pl_obs[2 * i], dy = uv[1] - m->pl_obs[2 * i + 1];
            res[i] = sqrt(dx * dx + dy * dy) * sqrt(m->sigma2p[i]);
        }
        double mean, stdv;
        orc_mean_stdv_mad(res, m->np, &mean, &stdv);
        double th = p->inlier_k * stdv;
        for (int i = 0; i < m->np; ++i)
            if (m->inlier_p[i] && fabs(res[i] - mean) > th) {
                m->inlier_p[i] = 0;
                (*n_inl_pt)--;
            }
        free(res);
    }
    if (p->has_lines) {
        double* res = (double*)malloc(sizeof(double) * (size_t)(m->nl + 1));
        for (int i = 0; i < m->nl; ++i) {
            double Pc[3], s[2], t[2];
            transform_project(DT, m->sP + 3 * i, cam, Pc, s);
            transform_project(DT, m->eP + 3 * i, cam, Pc, t);
            const double* l = m->le_obs + 3 * i;
            double ds = l[0] * s[0] + l[1] * s[1] + l[2], de = l[0] * t[0] + l[1] * t[1] + l[2];
            res[i] = sqrt(ds * ds + de * de) * sqrt(m->sigma2l[i]);
        }
        double mean, stdv;
        orc_mean_stdv_mad(res, m->nl, &mean, &stdv);
        double th = p->inlier_k * stdv;
        for (int i = 0; i < m->nl; ++i)
            if (fabs(res[i] - mean) > th && m->inlier_l[i]) {
                m->inlier_l[i] = 0;
                (*n_inl_ls)--;
            }
        free(res);
    }
}

static int is_identity16(const double* T) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (!(M4(T, i, j) == ((i == j) ? 1.0 : 0.0))) return 0;
    return 1;
}

/* optimizePose, src/stereoFrameHandler.cpp:307-392.  `init_T` is the DT chosen at :317-326 (the
 * caller applies the motion-model rule); n_inliers = number of matched records whose inlier flag
 * is set on entry (f2fTracking sets all of them, :126-128,148,176).  The Tfw composition
 * (:377-378) is done by the caller. */
void orc_optimize_pose(const double init_T[16], const stvo_cam* cam, const stvo_opt_params* p, orc_matched* m,
                       stvo_pose_result* out) {
    double DT[16], DT_[16], cov[36] = {0}, err = -1.0;
    memcpy(DT, init_T, sizeof(DT));
    memset(out, 0, sizeof(*out));
    int32_t n_pt = 0, n_ls = 0;
    for (int i = 0; i < m->np; ++i) n_pt += m->inlier_p[i] ? 1 : 0;
    for (int i = 0; i < m->nl; ++i) n_ls += m->inlier_l[i] ? 1 : 0;
    out->n_matched_pt = m->np;
    out->n_matched_ls = m->nl;
    int status = STVO_POSE_OK, path = 0;
    if (n_pt + n_ls >= p->min_features) {
        memcpy(DT_, DT, sizeof(DT));
        out->iters[0] = run_mode(p->mode, DT_, cov, &err, p->max_iters, cam, p, m); /* :335-338 */
        if (orc_is_good_solution(DT_, cov, err)) {                                  /* :341 */
            path |= STVO_PATH_STAGE1_GOOD;
            orc_remove_outliers(DT_, cam, p, m, &n_pt, &n_ls);
            if (n_pt + n_ls >= p->min_features) { /* :345 — restarts from the INITIAL DT */
                path |= STVO_PATH_REFINED;
                out->iters[1] = run_mode(p->mode, DT, cov, &err, p->max_iters_ref, cam, p, m);
            } else {
                for (int i = 0; i < 16; ++i) DT[i] = (i % 5 == 0) ? 1.0 : 0.0;
                status = STVO_POSE_FEW_INLIERS_AFTER;
            }
        } else { /* :357-362 */
            path |= STVO_PATH_ROBUST_FALLBACK;
            out->iters[1] = gauss_newton_robust(DT, cov, &err, p->max_iters_ref, cam, p, m);
        }
    } else {
        for (int i = 0; i < 16; ++i) DT[i] = (i % 5 == 0) ? 1.0 : 0.0;
        status = STVO_POSE_FEW_INLIERS_BEFORE;
    }
    memcpy(out->T_opt, DT, sizeof(DT));
    out->err_opt = err;
    if (orc_is_good_solution(DT, cov, err) && !is_identity16(DT)) { /* :372-381 */
        double Ti[16], x[6];
        orc_inverse_se3(DT, Ti);
        orc_logmap_se3(Ti, x);
        orc_expmap_se3(x, out->T);
        memcpy(out->cov, cov, sizeof(cov));
        out->err = err;
        orc_eig6(cov, out->cov_eig);
    } else { /* :382-391 */
        for (int i = 0; i < 16; ++i) out->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
        memset(out->cov, 0, sizeof(out->cov));
        memset(out->cov_eig, 0, sizeof(out->cov_eig));
        out->err = -1.0;
        if (status == STVO_POSE_OK) status = STVO_POSE_REJECTED;
    }
    out->status = status;
    out->path = path;
    out->n_inliers_pt = n_pt;
    out->n_inliers_ls = n_ls;
}


/* ---- key-frame decision: StereoFrameHandler::needNewKF / currFrameIsKF (src/stereoFrameHandler.cpp:1136-1218), the
 * "slam functions" PL-SLAM drives on top of PL-StVO.  Eigen's Matrix6d::determinant() (PartialPivLU) is restated as
 * Gaussian elimination with partial pivoting; uncTinv_se3 follows src/auxiliar.cpp:184-190. ---- */
static double orc_det6(const double* Ain) {
    double A[36];
    memcpy(A, Ain, sizeof(A));
    double det = 1.0;
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double big = fabs(A[k * 6 + k]);
        for (int i = k + 1; i < 6; ++i)
            if (fabs(A[i * 6 + k]) > big) {
                big = fabs(A[i * 6 + k]);
                p = i;
            }
        if (big == 0.0) return 0.0;
        if (p != k) {
            for (int j = 0; j < 6; ++j) {
                const double t = A[k * 6 + j];
                A[k * 6 + j] = A[p * 6 + j];
                A[p * 6 + j] = t;
            }
            det = -det;
        }
        det *= A[k * 6 + k];
        for (int i = k + 1; i < 6; ++i) {
            const double f = A[i * 6 + k] / A[k * 6 + k];
            for (int j = k; j < 6; ++j) A[i * 6 + j] -= f * A[k * 6 + j];
        }
    }
    return det;
}

static void orc_sandwich6(const double* A, const double* C, double* out) { /* out = A C A^T */
    double T[36];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += A[i * 6 + k] * C[k * 6 + j];
            T[i * 6 + j] = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += T[i * 6 + k] * A[j * 6 + k];
            out[i * 6 + j] = s;
        }
}

/* state = {prev_f_iskf, entropy_first_prevKF, N_prevKF_currF, T_prevKF[16], cov_prevKF_currF[36]} = 55 doubles
 * (src/stereoFrameHandler.cpp:48-51).  Returns 1 when a new key-frame is needed (:1173-1178), else counts the frame. */
int orc_need_new_kf(double* st, const double* Tfw, const double* DT, const double* DT_cov, double min_entropy_ratio,
                    double max_kf_t_dist, double max_kf_r_dist) {
    double* T_prevKF = st + 3;
    double* cov_acc = st + 19;
    const double c0 = 3.0 * (1.0 + log(2.0 * acos(-1.0)));
    if (st[0] != 0.0) { /* :1140-1153 */
        const double det = orc_det6(DT_cov);
        st[1] = det != 0.0 ? c0 + 0.5 * log(det) : -999999999.99;
        st[0] = 0.0;
    }
    double Ti[16], D[16], dX[6];
    orc_inverse_se3(Tfw, Ti);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += Ti[i * 4 + k] * T_prevKF[k * 4 + j];
            D[i * 4 + j] = s;
        }
    orc_logmap_se3(D, dX);
    const double t = sqrt(dX[0] * dX[0] + dX[1] * dX[1] + dX[2] * dX[2]);
    const double r = sqrt(dX[3] * dX[3] + dX[4] * dX[4] + dX[5] * dX[5]) * 180.f / 3.1415926535897932384626433832795;
    double A[36], DTi[16], Ai[36], cinv[36], add[36];
    orc_adjoint_se3(T_prevKF, A);
    orc_inverse_se3(DT, DTi);
    orc_adjoint_se3(DTi, Ai);
    orc_sandwich6(Ai, DT_cov, cinv); /* uncTinv_se3 */
    orc_sandwich6(A, cinv, add);
    for (int i = 0; i < 36; ++i) cov_acc[i] += add[i];
    const double entropy_curr = c0 + 0.5 * log(orc_det6(cov_acc));
    const double ratio = entropy_curr / st[1];
    int zero_cov = 1, ident = 1;
    for (int i = 0; i < 36; ++i) zero_cov = zero_cov && DT_cov[i] == 0.0;
    for (int i = 0; i < 16; ++i) ident = ident && DT[i] == ((i % 5 == 0) ? 1.0 : 0.0);
    if (ratio < min_entropy_ratio || isnan(ratio) || isinf(ratio) || (zero_cov && ident) || t > max_kf_t_dist ||
        r > max_kf_r_dist || st[2] > 10.0)
        return 1;
    st[2] += 1.0;
    return 0;
}

/* the state part of currFrameIsKF (:1205-1216): Tfw = I, then T_prevKF = Tfw, covariance and counter reset */
void orc_curr_frame_is_kf(double* st) {
    st[0] = 1.0;
    st[2] = 0.0;
    for (int i = 0; i < 16; ++i) st[3 + i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 36; ++i) st[19 + i] = 0.0;
}
