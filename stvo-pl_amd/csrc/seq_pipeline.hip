// seq_pipeline.hip — device-resident per-frame pipeline for B independent stereo sequences (SURVEY.md §8f
// rank 1): everything between "features extracted" and "pose committed" stays in HBM.
//
//   stvo_seq_push(features of frame k for B sequences)
//     1  point_cells_kernel   cell coordinates of the left key-points, CSR grid of the right key-points
//                             (/root/reference/src/stereoFrame.cpp:129-139; GridStructure as CSR)
//     2  K3 grid matcher      StVO::matchGrid (points)               (src/matching.cpp:111-177)
//     3  point_tail_kernel    epipolar / disparity filters, back-projection, sigma2, ordered compaction of
//                             the stereo points and their descriptor rows (stereoFrame.cpp:149-172)
//     4  line_cells_kernel    end-point cells, Bresenham rasterisation of the right lines into the CSR grid,
//                             unit directions (stereoFrame.cpp:318-338, src/lineIterator.cpp:34-77)
//     5  K3 grid matcher      StVO::matchGrid (lines)                (src/matching.cpp:179-258)
//     6  line_tail_kernel     overlap, end-point re-intersection, disparity filters, back-projection,
//                             ordered compaction (stereoFrame.cpp:348-397, :405-415, :473-508)
//     7  K1/K2                f2f mutual matching against the previous frame's stereo sets
//                             (src/stereoFrameHandler.cpp:131-180 -> matching.cpp:63-91)
//     8  pose kernel          optimizePose                           (src/stereoFrameHandler.cpp:307-392)
//   then the two stereo-set buffers swap roles (updateFrame, :89-100).  One upload, one small download
//   (B pose results + counters) and one synchronisation per frame.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#include "ctx_internal.h"
#include "point_cells.h"
#include "pose_math.h"

// The geometry filters of the tail kernels decide on thresholds (min_disp, ls_min_disp_ratio, stereo_overlap_th): no
// fused multiply-adds, so that they round exactly like the reference's separate operations (and like the host mirror,
// which is built with -ffp-contract=off).
#pragma clang fp contract(off)

namespace stvo {
namespace {

constexpr int LENT = STVO_GRID_COLS + STVO_GRID_ROWS + 4;  // upper bound of Bresenham cells of one in-image line

struct SeqDev {
    int B, K, M;
    const stvo_cam* cams;   // [B] one calibration per sequence (config/dataset_params/kitti00-02 / 03 / 04-10.yaml differ)
    const double* inv_wh;   // [B][2] 64 / cols, 48 / rows of the sequence's images (stereoFrame.cpp:47-48)
    stvo_match_params mp;
    // raw features of the current frame
    const float* kp_l;      // [B][K][2]
    const int32_t* oct_l;   // [B][K]
    const uint8_t* desc_l;  // [B][K][32]
    const int32_t* n_kp_l;  // [B]
    const float* kp_r;
    const uint8_t* desc_r;
    const int32_t* n_kp_r;
    const float* kl_l;      // [B][M][4]
    const int32_t* oct_ll;  // [B][M]
    const uint8_t* ldesc_l;
    const int32_t* n_kl_l;
    const float* kl_r;
    const uint8_t* ldesc_r;
    const int32_t* n_kl_r;
    // grid scratch
    int32_t* pxy_l;    // [B][K][2]
    int32_t* pstart;   // [B][3073]
    int32_t* pitems;   // [B][K]
    int32_t* prank;    // [B][K]
    int32_t* pperm;    // [B][K]
    int32_t* lxy_l;    // [B][M][4]
    int32_t* lstart;   // [B][3073]
    int32_t* litems;   // [B][M*LENT]
    int32_t* lrank;    // [B][M]
    int32_t* lperm;    // [B][M]
    double* ldir;      // [B][M][2]
    const int32_t* m12s_p;  // [B][K] stereo matches of the left points
    const int32_t* m12s_l;  // [B][M]
    // stereo set being built (curr)
    float4* rc;     // [B][K] {u, v, disparity, level}: the stereo point (point_tail.h)
    uint8_t* desc;  // [B][K][32]
    int32_t* n;     // [B]
    double* spl;    // [B][M][2]
    double* epl;
    double* sP;     // [B][M][3]
    double* eP;
    double* le;     // [B][M][3]
    double* s2l;    // [B][M] stereo sigma2
    double* s2lm;   // [B][M] sigma2 after LineFeature::safeCopy's re-scaling (what matched_ls carries)
    uint8_t* ldesc;
    int32_t* nl;
    // optional mirrors of n / nl in pinned HOST memory (zero-copy read-back for small batches), or nullptr
    int32_t* host_n;
    int32_t* host_nl;
    int zero_nl;  // line stage skipped for this frame: the point tail clears nl[b] (saves a memset launch)
    // scratch of the point grid matcher that point_cells_kernel initialises (no candidate bit-matrix kernel runs for points)
    unsigned long long* top2_p;  // [B][K]
    int32_t* govf_p;             // [B]
    uint32_t* plstart;           // [B][GRID_LSTART_STRIDE] | left key-points counting-sorted by cell for the one-workgroup-per-frame
    int32_t* plperm;             // [B][K]                  | matcher (GridBatch); nullptr when capacity or window exceed what it handles
    int32_t* pcell;              // [B][K] grid cell (y * 64 + x) of the right key-point at each CSR position, -1: outside the grid
    int32_t* prange;             // [B][K][2] candidate range of every left key-point in CSR positions (GridStructure::get, one-row window)
};

__device__ __forceinline__ bool in_grid(int x, int y) {
    return x >= 0 && x < STVO_GRID_COLS && y >= 0 && y < STVO_GRID_ROWS;
}

// exclusive scan of hist[0..256 * PER) in LDS by 256 threads (PER cells each); writes start[0..256 * PER]
template <int PER, typename T>
__device__ __forceinline__ void scan_cells_n(T* hist, int* s_wave, int32_t* start_out) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int local[PER];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        local[k] = hist[tid * PER + k];
        sum += local[k];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wv) base += s_wave[w];
    int run = base + incl - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        hist[tid * PER + k] = (T)run;  // hist becomes the exclusive start
        start_out[tid * PER + k] = run;
        run += local[k];
    }
    if (tid == 255) start_out[256 * PER] = run;
    __syncthreads();
}
template <typename T>
__device__ __forceinline__ void scan_cells(T* hist, int* s_wave, int32_t* start_out) {
    scan_cells_n<STVO_GRID_CELLS / 256>(hist, s_wave, start_out);
}

// ---- 1: points — cells + CSR of the right key-points (point_cells.h) -------------------------------
__host__ __device__ __forceinline__ PointCells point_cells_args(const SeqDev& s) {
    PointCells c;
    c.K = s.K; c.ws = s.mp.matching_s_ws;
    c.kp_l = s.kp_l; c.kp_r = s.kp_r; c.n_kp_l = s.n_kp_l; c.n_kp_r = s.n_kp_r; c.inv_wh = s.inv_wh;
    c.pstart = s.pstart; c.plstart = s.plstart; c.plperm = s.plperm; c.pperm = s.pperm; c.pcell = s.pcell;
    c.pxy_l = s.pxy_l; c.top2_p = s.top2_p; c.govf_p = s.govf_p; c.prange = s.prange; c.pitems = s.pitems; c.prank = s.prank;
    return c;
}
// (16-bit counters — half the LDS, so that line workgroups fit beside this kernel's — gained 0.7 % on 1024 KITTI-shaped streams and
// lost 10 % on 512 EuRoC-shaped ones, where the line stream is the longer one and every speed-up of the point stream takes CUs from it)
template <bool LEAN>
__global__ __launch_bounds__(256) void point_cells_kernel(SeqDev s) {
    __shared__ PointCellsLds<256> lds;
    point_cells_frame<256, LEAN>(point_cells_args(s), blockIdx.x, &lds);
}

// ---- 3: points — filters, back-projection, ordered compaction ---------------------------------------
// 1024 threads per frame: the compaction offset is carried from chunk to chunk (three barriers each), so fewer, larger
// chunks shorten the chain — 2 chunks instead of 8 for ~2000 key-points.
constexpr int TAIL_BLOCK = 1024;
__host__ __device__ __forceinline__ PointTail point_tail_args(const SeqDev& s) {
    PointTail t;
    t.kp_l = s.kp_l; t.kp_r = s.kp_r; t.oct_l = s.oct_l; t.desc_l = s.desc_l; t.cams = s.cams;
    t.max_dist_epip = s.mp.max_dist_epip; t.min_disp = s.mp.min_disp;
    t.rc = s.rc; t.desc = s.desc; t.n = s.n; t.host_n = s.host_n; t.nl = s.nl; t.zero_nl = s.zero_nl;
    return t;
}
__global__ __launch_bounds__(TAIL_BLOCK) void point_tail_kernel(SeqDev s) {
    __shared__ int s_wave[TAIL_BLOCK / 64];
    __shared__ int s_run;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nl = s.n_kp_l[b];
    const size_t off = (size_t)b * s.K;
    const PointTail t = point_tail_args(s);
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nl; base += TAIL_BLOCK) {
        const int i = base + tid;
        bool ok = false;
        double disp = 0.0;
        if (i < nl) {
            const int i2 = s.m12s_p[off + i];
            if (i2 >= 0) ok = point_tail_filter(t, off, i, i2, disp);
        }
        // ordered compaction: stereo_pt / pdesc_l keep ascending left index (:161-172)
        const unsigned long long bal = __ballot(ok);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(bal);
        __syncthreads();
        int wbase = s_run;
        for (int w = 0; w < wv; ++w) wbase += s_wave[w];
        if (ok) point_tail_write(t, off, i, off + (size_t)(wbase + before), disp);
        __syncthreads();
        if (tid == 0) {
            int q = 0;
            for (int w = 0; w < TAIL_BLOCK / 64; ++w) q += s_wave[w];
            s_run += q;
        }
        __syncthreads();
    }
    if (tid == 0) {
        s.n[b] = s_run;
        if (s.host_n) s.host_n[b] = s_run;
        if (s.zero_nl) s.nl[b] = 0;
    }
}

// LineIterator (src/lineIterator.cpp:34-77): calls f(x, y) for every Bresenham cell
template <typename F>
__device__ __forceinline__ void bresenham(double x1, double y1, double x2, double y2, F f) {
    const bool steep = fabs(y2 - y1) > fabs(x2 - x1);
    double t;
    if (steep) {
        t = x1; x1 = y1; y1 = t;
        t = x2; x2 = y2; y2 = t;
    }
    if (x1 > x2) {
        t = x1; x1 = x2; x2 = t;
        t = y1; y1 = y2; y2 = t;
    }
    const double dx = x2 - x1, dy = fabs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int y = (int)y1;
    const int maxX = (int)x2;
    int guard = 0;
    for (int x = (int)x1; x <= maxX && guard < LENT; ++x, ++guard) {
        f(steep ? y : x, steep ? x : y);
        error -= dy;
        if (error < 0) {
            y += ystep;
            error += dx;
        }
    }
}

// ---- 4: lines — end-point cells, rasterised CSR of the right lines, directions ----------------------
__global__ __launch_bounds__(256) void line_cells_kernel(SeqDev s) {
    __shared__ int hist[STVO_GRID_CELLS];
    __shared__ int fill[STVO_GRID_CELLS];
    __shared__ int s_wave[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nl = s.n_kl_l[b], nr = s.n_kl_r[b];
    const size_t off = (size_t)b * s.M;
    const double inv_w = s.inv_wh[2 * b], inv_h = s.inv_wh[2 * b + 1];
    for (int i = tid; i < nl; i += 256) {  // :318-322
        const float* kl = s.kl_l + (off + i) * 4;
        s.lxy_l[(off + i) * 4 + 0] = (int)((double)kl[0] * inv_w);
        s.lxy_l[(off + i) * 4 + 1] = (int)((double)kl[1] * inv_h);
        s.lxy_l[(off + i) * 4 + 2] = (int)((double)kl[2] * inv_w);
        s.lxy_l[(off + i) * 4 + 3] = (int)((double)kl[3] * inv_h);
    }
    for (int c = tid; c < STVO_GRID_CELLS; c += 256) {
        hist[c] = 0;
        fill[c] = 0;
    }
    __syncthreads();
    for (int j = tid; j < nr; j += 256) {  // :325-338
        const float* kl = s.kl_r + (off + j) * 4;
        const double vx = (double)(kl[2] - kl[0]) * inv_w;  // float difference, then * double (:331)
        const double vy = (double)(kl[3] - kl[1]) * inv_h;
        const double mag = sqrt(vx * vx + vy * vy);
        s.ldir[(off + j) * 2 + 0] = vx / mag;
        s.ldir[(off + j) * 2 + 1] = vy / mag;
        s.lperm[off + j] = j;  // few lines: identity scan order
        s.lrank[off + j] = j;
        bresenham((double)kl[0] * inv_w, (double)kl[1] * inv_h, (double)kl[2] * inv_w, (double)kl[3] * inv_h,
                  [&](int x, int y) {
                      if (in_grid(x, y)) atomicAdd(&hist[y * STVO_GRID_COLS + x], 1);
                  });
    }
    __syncthreads();
    scan_cells(hist, s_wave, s.lstart + (size_t)b * (STVO_GRID_CELLS + 1));
    int32_t* items = s.litems + (size_t)b * s.M * LENT;
    for (int j = tid; j < nr; j += 256) {
        const float* kl = s.kl_r + (off + j) * 4;
        bresenham((double)kl[0] * inv_w, (double)kl[1] * inv_h, (double)kl[2] * inv_w, (double)kl[3] * inv_h,
                  [&](int x, int y) {
                      if (in_grid(x, y)) {
                          const int c = y * STVO_GRID_COLS + x;
                          items[hist[c] + atomicAdd(&fill[c], 1)] = j;
                      }
                  });
    }
}

// ---- 6: lines — geometry filters, back-projection, ordered compaction --------------------------------
// the tail of frame b by a workgroup of T threads; match_of(i) = stereo match of left line i (the caller has put a barrier
// after s_run = 0 and after whatever match_of reads)
template <int T, typename MatchOf>
__device__ __forceinline__ void line_tail_frame(const SeqDev& s, const int b, MatchOf match_of, int* s_wave, int* s_run) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nl = s.n_kl_l[b];
    const size_t off = (size_t)b * s.M;
    const stvo_cam cam = s.cams[b];
    for (int base = 0; base < nl; base += T) {
        const int i = base + tid;
        bool ok = false;
        double sp_l[2] = {0, 0}, ep_l[2] = {0, 0}, le_l[3] = {0, 0, 0}, disp_s = 0.0, disp_e = 0.0;
        if (i < nl) {
            const int i2 = match_of(i);
            if (i2 >= 0) {
                const float* l = s.kl_l + (off + i) * 4;
                const float* r = s.kl_r + (off + i2) * 4;
                sp_l[0] = (double)l[0]; sp_l[1] = (double)l[1];
                ep_l[0] = (double)l[2]; ep_l[1] = (double)l[3];
                // le_l = sp_l x ep_l (homogeneous, w = 1), normalised by |(l0, l1)|   (:353-355)
                le_l[0] = sp_l[1] - ep_l[1];
                le_l[1] = ep_l[0] - sp_l[0];
                le_l[2] = sp_l[0] * ep_l[1] - sp_l[1] * ep_l[0];
                const double nrm = sqrt(le_l[0] * le_l[0] + le_l[1] * le_l[1]);
                le_l[0] /= nrm; le_l[1] /= nrm; le_l[2] /= nrm;
                double sp_r[2] = {(double)r[0], (double)r[1]}, ep_r[2] = {(double)r[2], (double)r[3]};
                const double overlap = pm::stereo_row_overlap(sp_l[1], ep_l[1], sp_r[1], ep_r[1], s.mp.line_horiz_th);
                // :363-364 — the second line reads the ALREADY overwritten sp_r (reference quirk, kept)
                sp_r[0] = (sp_r[0] * (sp_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - sp_l[1])) / (sp_r[1] - ep_r[1]);
                sp_r[1] = sp_l[1];
                ep_r[0] = (sp_r[0] * (ep_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - ep_l[1])) / (sp_r[1] - ep_r[1]);
                ep_r[1] = ep_l[1];
                pm::stereo_line_disparities(sp_l[0], ep_l[0], sp_r[0], ep_r[0], s.mp.ls_min_disp_ratio, &disp_s, &disp_e);  // :405-415
                ok = disp_s >= s.mp.min_disp && disp_e >= s.mp.min_disp && fabs(sp_l[1] - ep_l[1]) > s.mp.line_horiz_th &&
                     fabs(sp_r[1] - ep_r[1]) > s.mp.line_horiz_th && overlap > s.mp.stereo_overlap_th;
            }
        }
        const unsigned long long bal = __ballot(ok);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(bal);
        __syncthreads();
        int wbase = (*s_run);
        for (int w = 0; w < wv; ++w) wbase += s_wave[w];
        if (ok) {
            const size_t k = off + (size_t)(wbase + before);
            const double bds = cam.b / disp_s, bde = cam.b / disp_e;
            s.spl[k * 2 + 0] = sp_l[0]; s.spl[k * 2 + 1] = sp_l[1];
            s.epl[k * 2 + 0] = ep_l[0]; s.epl[k * 2 + 1] = ep_l[1];
            s.sP[k * 3 + 0] = bds * (sp_l[0] - cam.cx);
            s.sP[k * 3 + 1] = bds * (sp_l[1] - cam.cy);
            s.sP[k * 3 + 2] = bds * cam.fx;
            s.eP[k * 3 + 0] = bde * (ep_l[0] - cam.cx);
            s.eP[k * 3 + 1] = bde * (ep_l[1] - cam.cy);
            s.eP[k * 3 + 2] = bde * cam.fx;
            s.le[k * 3 + 0] = le_l[0]; s.le[k * 3 + 1] = le_l[1]; s.le[k * 3 + 2] = le_l[2];
            const int level = s.oct_ll[off + i];
            double sg = 1.0;  // LineFeature ctor (src/stereoFeatures.cpp:107-115)
            for (int t = 0; t < level; ++t) sg *= s.mp.lsd_scale;
            const double s2 = 1.0 / (sg * sg);
            s.s2l[k] = s2;
            double sm = s2;  // LineFeature::safeCopy re-applies the level scaling (:117-135)
            for (int t = 0; t < level; ++t) sm *= s.mp.lsd_scale;
            s.s2lm[k] = 1.0 / (sm * sm);
            const uint4* src = reinterpret_cast<const uint4*>(s.ldesc_l + (off + i) * STVO_DESC_BYTES);
            uint4* dst = reinterpret_cast<uint4*>(s.ldesc + k * STVO_DESC_BYTES);
            dst[0] = src[0];
            dst[1] = src[1];
        }
        __syncthreads();
        if (tid == 0)
            for (int w = 0; w < T / 64; ++w) (*s_run) += s_wave[w];
        __syncthreads();
    }
    if (tid == 0) {
        s.nl[b] = *s_run;
        if (s.host_nl) s.host_nl[b] = *s_run;
    }
}

__global__ __launch_bounds__(256) void line_tail_kernel(SeqDev s) {
    __shared__ int s_wave[4];
    __shared__ int s_run;
    const int b = blockIdx.x;
    const size_t off = (size_t)b * s.M;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    line_tail_frame<256>(s, b, [&](int i) { return s.m12s_l[off + i]; }, s_wave, &s_run);
}

// ---- 4 + 5 + 6 in one workgroup per frame: the whole stereo association of the key-lines --------------------------------------
// A frame holds ~100 key-lines (lsd_nfeatures 100 / 300).  Through the general grid machinery (CSR of the rasterised right lines,
// candidate bit-matrix, scan / elig / scan / finalize, tail: seven launches, a thread or a wave per line) the stage took ~290 us
// per 1024 frames on an idle GPU and, sharing the CUs with the key-point stage, stretched that stage and the key-point scan by
// ~0.1 ms.  Here thread j owns RIGHT line j: it rasterises the line once into per-grid-row column intervals in LDS (a Bresenham
// line covers one contiguous run of columns in every row it crosses), and the candidates of left line i1 — the right lines with a
// cell in the window of either END-POINT cell of i1 (src/matching.cpp:213-215; window = matching_s_ws columns to the left, same
// row, stereoFrame.cpp:340-342) — become two interval tests.  matchGrid's order dependence (:145-150: a pair takes part only if it
// strictly improves on every earlier left line that met the same right line) is local to thread j, which walks i1 in ascending
// order; the best of a left line is an LDS atomic min over (distance << 16 | j), the ratio test (:241, DOUBLE) a second walk, the
// mutual check (:247-255) one thread per left line.  Left lines, directions and descriptors are read through wave-uniform LDS
// addresses (broadcasts).  Ties: the best of two equal distances fails the ratio test for ratios <= 1 whichever is "first".
// The columns [cmin, cmax] of a right line in a grid row are kept as the bytes cmin | (255 - cmax) << 8 and read into two 16-bit
// lanes (one v_perm_b32); the window [lo, hi] of a left end-point is hi | (255 - lo) << 16: the intervals meet iff both lanes of
// (window - run) are >= 0 — a permute, one packed subtraction and a mask per (end-point, right line) instead of two byte extractions
// and four comparisons (the candidate matrix is about two thirds of this kernel's instructions).  (The same with 32-bit entries, no
// permute: SLOWER — 0.845 -> 0.853 ms per 1024 KITTI-shaped streams, 0.470 -> 0.598 ms per 512 EuRoC-shaped ones: the LDS the kernel
// asks for, i.e. how many of its workgroups a CU holds, weighs more than its instruction count.)
constexpr uint32_t LSF_ROW_EMPTY = 0xFFFFu;      // cmin = 255 > cmax = 0
constexpr uint32_t LSF_NO_WINDOW = 0x0000FFFFu;  // hi = -1: below every cmin
constexpr int LSF_MAX_LINES = 512, LSF_BYTES_PER_LINE = 32 + 16 + 16 + 4 + 2 * STVO_GRID_ROWS + 2 + 2 + 1;
// Mk (<= s.M): lines per image the LDS arrays are sized for — the host knows that no image of the batch holds more.
// T: threads per frame (256; a single wave per frame was tried to hold fewer wave slots, and was slower).
template <int T>
__global__ __launch_bounds__(T) void line_stereo_fused_kernel(SeqDev s, const int Mk, const int mutual, const double ratio) {
    extern __shared__ uint4 s_dyn[];
    __shared__ int s_wave[4];
    __shared__ int s_run;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int M = Mk, b = blockIdx.x, tid = threadIdx.x;
    u32x4* dl = reinterpret_cast<u32x4*>(s_dyn);                                      // [M][2] left descriptor rows
    int4* cxy = reinterpret_cast<int4*>(dl + 2 * M);                                  // [M] windows of the two end-point cells of the left lines: (grid row * M, packed window) twice
    double2* vdir = reinterpret_cast<double2*>(cxy + M);                              // [M] their unit directions (integer cell differences)
    uint32_t* best = reinterpret_cast<uint32_t*>(vdir + M);                           // [M] min (d << 16 | j) over the eligible pairs
    unsigned short* rows = reinterpret_cast<unsigned short*>(best + M);               // [48][M] cmin | (255 - cmax) << 8 of right line j in grid row y
    unsigned short* owner = rows + (size_t)STVO_GRID_ROWS * M;                        // [M] matches_21
    short* mm = reinterpret_cast<short*>(owner + M);                                  // [M] the stereo match of left line i
    unsigned char* blocked = reinterpret_cast<unsigned char*>(mm + M);                // [M] ratio test failed
    uint32_t* cand = reinterpret_cast<uint32_t*>(blocked + M + ((4 - (M & 3)) & 3));  // [M / 32][M] bit k of word (w, j): left line 32 w + k has right line j as a candidate
    const int nl = min(s.n_kl_l[b], M), nr = min(s.n_kl_r[b], M);
    const size_t off = (size_t)b * s.M;
    const double inv_w = s.inv_wh[2 * b], inv_h = s.inv_wh[2 * b + 1];
    const int ws = s.mp.matching_s_ws;
    const double sim_th = s.mp.line_sim_th;
    if (tid == 0) s_run = 0;
    for (int i = tid; i < nl; i += T) {  // :318-322, include/matching.h:48-53
        const float* kl = s.kl_l + (off + i) * 4;
        int4 c;
        c.x = (int)((double)kl[0] * inv_w);
        c.y = (int)((double)kl[1] * inv_h);
        c.z = (int)((double)kl[2] * inv_w);
        c.w = (int)((double)kl[3] * inv_h);
        // GridStructure::get(x, y, {ws, 0, 0, 0}): columns max(x - ws, 0) .. min(x, 63) of row y (nothing outside the grid's rows)
        auto window = [&](int x, int y, int& row_off, int& q) {
            const bool row_ok = y >= 0 && y < STVO_GRID_ROWS;
            const int lo = max(x - ws, 0), hi = min(x, STVO_GRID_COLS - 1);
            row_off = (row_ok ? y : 0) * M;
            q = (row_ok && lo <= hi) ? (int)((uint32_t)hi | ((uint32_t)(255 - lo) << 16)) : (int)LSF_NO_WINDOW;
        };
        int4 wq;
        window(c.x, c.y, wq.x, wq.y);
        window(c.z, c.w, wq.z, wq.w);
        cxy[i] = wq;
        const double vx = (double)(c.z - c.x), vy = (double)(c.w - c.y);
        const double mag = sqrt(vx * vx + vy * vy);
        vdir[i] = make_double2(vx / mag, vy / mag);  // 0 / 0 = NaN never skips a candidate (:207-222)
        const u32x4* g = reinterpret_cast<const u32x4*>(s.ldesc_l + (off + i) * STVO_DESC_BYTES);
        dl[2 * i] = g[0];
        dl[2 * i + 1] = g[1];
        best[i] = 0xFFFFFFFFu;
        blocked[i] = 0;
    }
    for (int j = tid; j < nr; j += T) {  // :325-338
        for (int y = 0; y < STVO_GRID_ROWS; ++y) rows[y * M + j] = (unsigned short)LSF_ROW_EMPTY;
        const float* kl = s.kl_r + (off + j) * 4;
        // the cells come row by row (the minor coordinate of a Bresenham walk is monotone): the run of the current row stays in
        // registers and is written once — no read-modify-write chain through LDS
        int cy = -1, cmn = 255, cmx = 0;
        auto flush = [&]() {
            if (cy >= 0 && cy < STVO_GRID_ROWS && cmn <= cmx) rows[cy * M + j] = (unsigned short)(cmn | ((255 - cmx) << 8));
        };
        bresenham((double)kl[0] * inv_w, (double)kl[1] * inv_h, (double)kl[2] * inv_w, (double)kl[3] * inv_h, [&](int x, int y) {
            if (y != cy) {
                flush();
                cy = y;
                cmn = 255;
                cmx = 0;
            }
            if (x >= 0 && x < STVO_GRID_COLS) {
                cmn = min(cmn, x);
                cmx = max(cmx, x);
            }
        });
        flush();
    }
    __syncthreads();
    // has right line j a cell in the window?  Both 16-bit lanes of (window - run) non-negative: cmin <= hi and lo <= cmax
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    auto in_window = [&](int j, int row_off, int q) -> bool {
        const uint32_t run = __builtin_amdgcn_perm(0u, (uint32_t)rows[row_off + j], 0x0c010c00u);  // bytes (cmin, 255 - cmax) -> 16-bit lanes
        const i16x2 d = __builtin_bit_cast(i16x2, q) - __builtin_bit_cast(i16x2, run);
        return (__builtin_bit_cast(uint32_t, d) & 0x80008000u) == 0u;
    };
    // candidate bit-matrix, all threads: word (w, j) = left lines 32 w .. 32 w + 31 that have right line j as a candidate.  The
    // pairs are independent (the loads of one trip do not wait for the trip before), and the walks below then visit the few
    // candidates of a right line instead of testing every left line in a chain of dependent LDS reads.
    const int words = (nl + 31) >> 5;
    for (int e = tid; e < words * M; e += T) {
        const int w = e / M, j = e - w * M;
        uint32_t bits = 0u;
        if (j < nr) {
            const int i_end = min(32, nl - 32 * w);
#pragma unroll 4
            for (int k = 0; k < i_end; ++k) {
                const int4 c = cxy[32 * w + k];
                if (in_window(j, c.x, c.y) || in_window(j, c.z, c.w)) bits |= 1u << k;  // (row offset, window) of either end point
            }
        }
        cand[e] = bits;
    }
    __syncthreads();
    // the walk of thread j over its candidates in ascending left index; f(i1, d) is called for every eligible pair
    auto walk = [&](int j, const u32x4 t0, const u32x4 t1, const double dirx, const double diry, auto f) {
        int run_min = 0x7FFFFFFF;
        for (int w = 0; w < words; ++w) {
            for (uint32_t bits = cand[w * M + j]; bits; bits &= bits - 1u) {
                const int i1 = 32 * w + __builtin_ctz(bits);
                const double2 v = vdir[i1];
                const double dot = v.x * dirx + v.y * diry;
                if (fabs(dot) < sim_th) continue;  // :221-222
                const u32x4 q0 = dl[2 * i1], q1 = dl[2 * i1 + 1];
                const int d = __builtin_popcount(q0.x ^ t0.x) + __builtin_popcount(q0.y ^ t0.y) + __builtin_popcount(q0.z ^ t0.z) +
                              __builtin_popcount(q0.w ^ t0.w) + __builtin_popcount(q1.x ^ t1.x) + __builtin_popcount(q1.y ^ t1.y) +
                              __builtin_popcount(q1.z ^ t1.z) + __builtin_popcount(q1.w ^ t1.w);
                if (mutual) {  // :145-150 running strict minimum per right line
                    if (d >= run_min) continue;
                    run_min = d;
                }
                f(i1, d);
            }
        }
    };
    auto right_line = [&](int j, u32x4& t0, u32x4& t1, double& dirx, double& diry) {
        const float* kl = s.kl_r + (off + j) * 4;
        const double vx = (double)(kl[2] - kl[0]) * inv_w;  // float difference, then * double (:331)
        const double vy = (double)(kl[3] - kl[1]) * inv_h;
        const double mag = sqrt(vx * vx + vy * vy);
        dirx = vx / mag;
        diry = vy / mag;
        const u32x4* g = reinterpret_cast<const u32x4*>(s.ldesc_r + (off + j) * STVO_DESC_BYTES);
        t0 = g[0];
        t1 = g[1];
    };
    for (int j = tid; j < nr; j += T) {
        u32x4 t0, t1;
        double dirx, diry;
        right_line(j, t0, t1, dirx, diry);
        int own = 0xFFFF;
        walk(j, t0, t1, dirx, diry, [&](int i1, int d) {
            own = i1;
            atomicMin(&best[i1], ((uint32_t)d << 16) | (uint32_t)j);
        });
        owner[j] = (unsigned short)own;
    }
    __syncthreads();
    for (int j = tid; j < nr; j += T) {  // :241 for every eligible pair that is not its left line's best
        u32x4 t0, t1;
        double dirx, diry;
        right_line(j, t0, t1, dirx, diry);
        walk(j, t0, t1, dirx, diry, [&](int i1, int d) {
            const uint32_t bk = best[i1];
            if (bk != (((uint32_t)d << 16) | (uint32_t)j)) {
                const double best_d = (double)(int)(bk >> 16), d2 = (double)d;
                if (!(best_d < d2 * ratio)) blocked[i1] = 1;
            }
        });
    }
    __syncthreads();
    for (int i1 = tid; i1 < s.M; i1 += T) {  // accept, mutual check (:247-255)
        int m = -1;
        if (i1 < nl) {
            const uint32_t bk = best[i1];
            if (bk != 0xFFFFFFFFu && !blocked[i1] && (double)(int)(bk >> 16) < 2147483647.0 * ratio) {
                const int j = (int)(bk & 0xFFFFu);
                if (!mutual || (int)owner[j] == i1) m = j;
            }
        }
        if (i1 < M) mm[i1] = (short)m;
        const_cast<int32_t*>(s.m12s_l)[off + i1] = m;
    }
    __syncthreads();
    // (left lines beyond the LDS arrays — only possible if the launch was sized for fewer lines than the slot holds — are unmatched,
    //  never read out of bounds)
    line_tail_frame<T>(s, b, [&](int i) { return i < M ? (int)mm[i] : -1; }, s_wave, &s_run);
}

}  // namespace
}  // namespace stvo

// ======================================================================================================
struct stvo_seq {
    stvo_ctx* ctx = nullptr;
    int B = 0, K = 0, M = 0, frame_idx = 0;
    stvo_match_params mp{};
    stvo_opt_params op{};
    double ratio_grid = 0;         // Config::minRatio12P() as the DOUBLE matchGrid compares with (matching.cpp:160,241)
    stvo_cam* d_cams = nullptr;    // [B] per-sequence calibration (device)
    double* d_inv_wh = nullptr;    // [B][2] per-sequence grid scale (device)
    double* d_qtab = nullptr;      // [STVO_POSE_QTAB] sqrt(sigma2) of pyramid level l (kernels.h: PoseArgs::q_tab)
    double* d_motion_T = nullptr;  // [B][16] use_motion_model: the next step's initial DT per sequence, written by the pose kernel's commit (stvo_seq_set_motion_model)
    long long* d_prof = nullptr;   // STVO_POSE_PROF (developer aid): [B][16] phase ticks of the last pose launch, printed by stvo_seq_read
    char* dev = nullptr;     // one allocation, carved below
    size_t dev_bytes = 0;
    // pinned mirrors of the raw-feature block: two, used alternately — packing frame k + 1 on the host overlaps the device work
    // of frame k.  A block may be written again once the copy that last read it has completed.  The usual caller synchronises the
    // stream once per frame anyway (stvo_seq_read), so the uploads are numbered and the read remembers the last one it has
    // waited for: no event behind the copy (on this runtime a recorded event delays the next kernel of the stream by several
    // microseconds).  Only a caller that uploads a third frame without a read in between pays for an event, at that upload.
    char* raw_host[2] = {nullptr, nullptr};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    unsigned long long stage_upload[2] = {0, 0};  // number of the upload that last read the block (0: none)
    unsigned long long uploads = 0, uploads_done = 0;
    int stage_next = 0;
    // The key-line kernels of a step run on their own stream, which must see the frame's key-line arrays.  When the point stream
    // has been idle since the last read (single-stream operation: upload, step, read per frame) stvo_seq_upload copies the key-line
    // part of the block ON the line stream and the step forks without an event; st_dirty = something the line stream would have to
    // wait for has been enqueued on the point stream since the last synchronisation.
    bool st_dirty = false;
    // ordering inside the pose kernel instead of events (kernels.h: PoseArgs::wait_flag / fetch_*), small batches only
    unsigned* d_join_flag = nullptr;     // device word the line stream's last launch of a step is followed by a signal on
    unsigned join_epoch = 0;
    unsigned fetch_epoch = 0;            // value the pose kernel publishes in the pinned flag once the match indices are in fetch_host
    bool fetch_by_pose = false;          // the last step's by-products come that way (stvo_seq_fetch_matches polls the flag)
    std::vector<char> raw_split;  // per slot: its key-line arrays were copied on the line stream
    size_t raw_bytes = 0;
    stvo::SeqDev d{};          // pointers into `dev` (set = current)
    // carve results
    // raw frame slots: 0 / 1 are carved from `dev` (push alternates between them); stvo_seq_set_slots adds more so that a
    // throughput caller can keep several frames of every sequence resident in HBM and rotate through them
    std::vector<char*> raw_dev;
    char* extra_raw = nullptr;
    struct Set {
        float4* rc;
        uint8_t* desc;
        int32_t* n;
        double *spl, *epl, *sP, *eP, *le, *s2l, *s2lm;
        uint8_t* ldesc;
        int32_t* nl;
    } set[3];  // THREE stereo sets (round 6): step k writes set k mod 3 and reads set (k - 1) mod 3, so that the association of step k + 1
               // (which writes set (k + 1) mod 3) may run while the pose kernel of step k still reads sets k - 1 and k (pipelined steps below)
    int cur = 0;
    int prev_set() const { return (cur + 2) % 3; }
    unsigned long long *cover, *top2;
    uint32_t *elig = nullptr, *elig_l = nullptr;         // eligible pairs of the grid scans (points / lines), see GridBatch
    int32_t *elig_cnt = nullptr, *elig_cnt_l = nullptr, *govf = nullptr, *govf_l = nullptr;
    int32_t *owner2, *m12s_p, *m12s_l, *m12p, *m12l, *inlp, *inll, *counts;
    // ---- pipelined steps (batches; STVO_SEQ_PIPE=0 switches them off): optimizePose(k) runs on the context's aux stream, and the point
    // stream goes straight on to the stereo association of step k + 1, whose workgroups take the CUs the pose kernel's one residency round
    // frees as its frame pairs finish (the pairs with the most evaluations are a ~20 % tail: profiles/r05_sq_counters.txt).  What makes the
    // two independent: the third stereo set (above), a second copy of the f2f match indices (odd steps write m12p_alt / m12l_alt), and the
    // events ev_match (point stream -> pose: the indices are complete) / ev_pose[k & 1] (pose -> the step that overwrites what it read).
    // The ORDER in which the two become ready matters for speed only: the pose kernel's workgroups must be dispatched before the matcher's
    // (one per CU, most of its LDS), so the point stream passes a one-thread gate kernel that leaves when the pose kernel
    // has begun (PoseArgs::start_flag; kernels.h says why not "has been dispatched completely") — bounded, a hint: every data dependence is carried by the events.
    int32_t *m12p_alt = nullptr, *m12l_alt = nullptr;
    hipEvent_t ev_match = nullptr, ev_pose[2] = {nullptr, nullptr};
    unsigned* d_pose_flag = nullptr;
    unsigned pose_epoch = 0;
    int32_t *d_dyn_ctr = nullptr, *d_dyn_owner = nullptr;  // GridBatch::dyn_ctr [2] / dyn_owner [B]: frame tickets of the persistent point matcher
    long long dyn_last_frame = -2;                         // the last step whose matcher took tickets (it reset the other counter)
    bool pose_pending[2] = {false, false};  // ev_pose[i] has been recorded and not yet waited for by the point stream
    bool piped_last = false;                // the last step put its pose kernel on the aux stream (stvo_seq_read waits for it)
    // ---- key-line stage AHEAD (the default for batches, round 6; STVO_LINES_AHEAD=0 switches it off): the key-line kernels of step k + 1
    // (stereo association + f2f of ~100 rows per image: 1024 small workgroups each) do not depend on anything the point stream does in
    // step k + 1, so the line stream no longer waits for that step's fork event.  It waits for the fork event of step k — recorded behind
    // optimizePose(k - 1), the last reader of the line set and of the match-index copy step k + 1 overwrites (three sets, two copies) —
    // and passes a gate that opens when optimizePose(k) has been dispatched: the small workgroups then fill the slots that kernel's one
    // residency round frees as its frame pairs finish, instead of sharing the issue ports with the forward scan K1m(k + 1), which they
    // stretched by ~40 us per step.
    long long fork_rec_frame = -2;   // the last step that recorded ev_fork on the point stream
    long long pose_flag_frame = -2;  // the last step whose pose kernel publishes its start (d_pose_flag == pose_epoch)
    stvo_pose_result* results;
    char* out_host = nullptr;  // pinned: results + counts
    // second stream: the line stage (stereo association + f2f of the key-lines) is independent of the point stage
    // until optimizePose (the reference runs the two in parallel threads, stereoFrame.cpp:67-72, stereoFrameHandler.cpp:
    // 115-118) and its kernels are far too small to fill the GPU, so it runs concurrently on its own scratch
    hipStream_t line_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Batches: the grid of frame k + 1 (point_cells_kernel) is enqueued on the LINE stream, which is idle from ~0.2 ms into step k on, and
    // the point stream only waits for it in front of its matcher — 38 us less in the chain of a step (see seq_enqueue_step).  The kernel's
    // outputs are double-buffered by the parity of the step (cells_buf[cur]): the grid of frame k + 1 may be built while the matcher
    // of frame k still reads its own.
    hipEvent_t ev_cells = nullptr, ev_upload = nullptr;
    unsigned long long upload_seq = 0, upload_seen = 0;  // uploads enqueued on the point stream / the last one the line stream has been made to wait for
    long long sl_forked_frame = -2;                      // the last step in which the line stream waited for an event of the point stream
    struct CellsBuf {
        int32_t *pstart, *pperm, *pcell, *plperm;
        uint32_t* plstart;
    } cells_buf[2] = {};
    unsigned long long *cover_l = nullptr, *top2_l = nullptr;
    int32_t* owner2_l = nullptr;
    stvo::LazyScratch lazy_l{};
    // optional by-products for callers that mirror the reference's host-side feature lists (StereoFrameHandler):
    // the four match-index arrays are copied to pinned memory right after the f2f stage (ev_fetch), i.e. while the
    // pose kernel still runs; the inlier masks follow the pose kernel
    bool fetch = false;
    char* fetch_host = nullptr;
    hipEvent_t ev_fetch = nullptr;
    size_t m12_span = 0, inl_span = 0;  // bytes of the contiguous [m12s_p | m12s_l | m12p | m12l] and [inlp | inll] blocks
    hipEvent_t pev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // STVO_SEQ_PROF stage markers (developer aid)
    // captured step chains, keyed by (slot, buffer parity, line-stage flags); small batches only (launch-bound)
    bool graph_mode = false;
    std::unordered_map<unsigned, hipGraphExec_t> graphs;
    // optional live stage timing (bench.py): event pairs around the kernels of every step, on the stream they run on
    int timing = 0;  // 1: every stage; 2: "light" — only the three big kernels of the point stream, the step otherwise as untimed
    std::vector<hipEvent_t> tev;  // STVO_SEQ_NSTAGE start/stop pairs per step, grown on demand
    size_t tev_used = 0;
    bool zero_copy = false;    // small batches: kernels write results / counts straight into out_host
    std::vector<char> raw_lines;  // slot holds at least one left and one right key-line
    std::vector<int> raw_max_lines;  // most key-lines of one image in the slot, as far as the host knows (device ingests: M)
    int set_lines_cap[3] = {0, 0, 0};   // the same bound for the stereo line sets
    bool set_lines[3] = {false, false, false};  // stereo set was built from a frame with key-lines
    bool last_lines = false;             // the last step ran the line stage
    int last_slot = 0;                   // raw slot of the last step
    stvo::GridBatch last_point_grid{};   // arguments of the last point grid match (test hook)
    stvo::GridBatch last_line_grid{};    // the general matcher's arguments for the key-lines of the last step (test hook)
    bool last_line_fused = false;        // the last step ran line_stereo_fused_kernel: the hook rebuilds the grid arrays
    size_t off_kp_l, off_oct_l, off_desc_l, off_nkl, off_kp_r, off_desc_r, off_nkr, off_kl_l, off_oct_ll, off_ldesc_l,
        off_nll, off_kl_r, off_ldesc_r, off_nlr;
};

namespace {

// ---- 0: device-resident features (e.g. the output of stvo_orb_detect_dev) into a raw frame slot ------------------------------
struct IngestArgs {
    int B, K, M, has_points, has_lines;
    stvo_frame_features f;  // DEVICE pointers, counts included
    float* kp_l; int32_t* oct_l; uint8_t* desc_l; int32_t* n_kp_l;
    float* kp_r; uint8_t* desc_r; int32_t* n_kp_r;
    float* kl_l; int32_t* oct_ll; uint8_t* ldesc_l; int32_t* n_kl_l;
    float* kl_r; uint8_t* ldesc_r; int32_t* n_kl_r;
};
__global__ __launch_bounds__(256) void seq_ingest_kernel(IngestArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    auto count = [&](const int32_t* n, int cap, int stride) { return n ? min(max(n[b], 0), min(cap, stride)) : 0; };
    const int nl = a.has_points ? count(a.f.n_kp_l, a.K, a.f.stride_kp) : 0, nr = a.has_points ? count(a.f.n_kp_r, a.K, a.f.stride_kp) : 0;
    const int ml = a.has_lines ? count(a.f.n_kl_l, a.M, a.f.stride_kl) : 0, mr = a.has_lines ? count(a.f.n_kl_r, a.M, a.f.stride_kl) : 0;
    if (tid == 0) {
        a.n_kp_l[b] = nl;
        a.n_kp_r[b] = nr;
        a.n_kl_l[b] = ml;
        a.n_kl_r[b] = mr;
    }
    const size_t sp = (size_t)b * a.f.stride_kp, dp = (size_t)b * a.K, sl = (size_t)b * a.f.stride_kl, dl = (size_t)b * a.M;
    auto rows32 = [&](const uint8_t* src, uint8_t* dst, int n) {  // 32-byte rows as two 16-byte words
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (int i = tid; i < 2 * n; i += 256) d4[i] = s4[i];
    };
    for (int i = tid; i < 2 * nl; i += 256) a.kp_l[dp * 2 + i] = a.f.kp_l[sp * 2 + i];
    for (int i = tid; i < nl; i += 256) a.oct_l[dp + i] = a.f.oct_l ? a.f.oct_l[sp + i] : 0;  // no octaves: one pyramid level
    rows32(a.f.desc_l + sp * 32, a.desc_l + dp * 32, nl);
    for (int i = tid; i < 2 * nr; i += 256) a.kp_r[dp * 2 + i] = a.f.kp_r[sp * 2 + i];
    rows32(a.f.desc_r + sp * 32, a.desc_r + dp * 32, nr);
    for (int i = tid; i < 4 * ml; i += 256) a.kl_l[dl * 4 + i] = a.f.kl_l[sl * 4 + i];
    for (int i = tid; i < ml; i += 256) a.oct_ll[dl + i] = a.f.oct_ll ? a.f.oct_ll[sl + i] : 0;
    if (ml) rows32(a.f.ldesc_l + sl * 32, a.ldesc_l + dl * 32, ml);
    for (int i = tid; i < 4 * mr; i += 256) a.kl_r[dl * 4 + i] = a.f.kl_r[sl * 4 + i];
    if (mr) rows32(a.f.ldesc_r + sl * 32, a.ldesc_r + dl * 32, mr);
}

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~size_t(255);
        return o;
    }
};

}  // namespace

extern "C" {

int stvo_seq_create_multi(stvo_ctx* ctx, int B, int max_keypoints, int max_keylines, const int32_t* img_cols,
                          const int32_t* img_rows, const stvo_cam* cams, const stvo_match_params* mp,
                          const stvo_opt_params* op, stvo_seq** out) {
    if (!ctx || !out || B <= 0 || max_keypoints <= 0 || max_keylines < 0 || !img_cols || !img_rows || !cams || !mp || !op)
        return STVO_ERR_INVALID_ARG;
    for (int b = 0; b < B; ++b)
        if (img_cols[b] <= 0 || img_rows[b] <= 0) return STVO_ERR_INVALID_ARG;
    if (max_keypoints > STVO_POSE_MAX_POINTS || max_keylines > STVO_POSE_MAX_LINES) return STVO_ERR_CAPACITY;
    // the pipeline runs the context's matching scratch (cand / need / qsel: max_rows x max_batch ints, nsel: [5][max_batch])
    // with K = max_keypoints rounded up to 64 as row stride: validate the ROUNDED figures
    const int K = (max_keypoints + 63) & ~63;
    const int M = max_keylines > 0 ? ((max_keylines + 63) & ~63) : 64;
    if (B > ctx->max_batch || K > ctx->max_rows || (size_t)B * (size_t)K > (size_t)ctx->max_batch * (size_t)ctx->max_rows)
        return STVO_ERR_CAPACITY;
    if ((size_t)2 * B * K > ctx->knn_capacity) return STVO_ERR_CAPACITY;  // forward top-2 + reverse-check scratch (reverse_plan)
    if (!(mp->min_ratio_12_p <= 1.0f) || !(mp->min_ratio_12_l <= 1.0f)) return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stvo_seq* s = new (std::nothrow) stvo_seq();
    if (!s) return STVO_ERR_HIP;
    s->ctx = ctx;
    s->B = B;
    s->K = K;
    s->M = M;
    s->mp = *mp;
    s->op = *op;
    s->ratio_grid = mp->min_ratio_12_p_d > 0.0 ? mp->min_ratio_12_p_d : (double)mp->min_ratio_12_p;
    if (!(s->ratio_grid <= 1.0)) {
        delete s;
        return STVO_ERR_INVALID_ARG;
    }
    const size_t nb = (size_t)B;
    // ---- raw block (mirrored in pinned host memory, one H2D per frame)
    Carver rc;
    s->off_kp_l = rc.take(nb * K * 2 * 4);
    s->off_oct_l = rc.take(nb * K * 4);
    s->off_desc_l = rc.take(nb * K * 32);
    s->off_nkl = rc.take(nb * 4);
    s->off_kp_r = rc.take(nb * K * 2 * 4);
    s->off_desc_r = rc.take(nb * K * 32);
    s->off_nkr = rc.take(nb * 4);
    s->off_kl_l = rc.take(nb * M * 4 * 4);
    s->off_oct_ll = rc.take(nb * M * 4);
    s->off_ldesc_l = rc.take(nb * M * 32);
    s->off_nll = rc.take(nb * 4);
    s->off_kl_r = rc.take(nb * M * 4 * 4);
    s->off_ldesc_r = rc.take(nb * M * 32);
    s->off_nlr = rc.take(nb * 4);
    s->raw_bytes = rc.off;
    // ---- everything else
    Carver c;
    const size_t o_raw = c.take(s->raw_bytes), o_raw1 = c.take(s->raw_bytes);
    const size_t o_cams = c.take(nb * sizeof(stvo_cam)), o_invwh = c.take(nb * 2 * 8), o_qtab = c.take(stvo::STVO_POSE_QTAB * 8),
                 o_joinflag = c.take(64), o_poseflag = c.take(64);
    const size_t o_pxy = c.take(nb * K * 2 * 4), o_pstart = c.take(nb * (STVO_GRID_CELLS + 1) * 4), o_pitems = c.take(nb * K * 4),
                 o_prank = c.take(nb * K * 4), o_pperm = c.take(nb * K * 4), o_prange = c.take(nb * K * 2 * 4), o_pcell = c.take(nb * K * 4);
    const bool lsort = K <= 2048 && mp->matching_s_ws >= 0 && mp->matching_s_ws <= stvo::GRID_LW - STVO_GRID_COLS;
    const size_t o_plstart = c.take(lsort ? nb * stvo::GRID_LSTART_STRIDE * 4 : 0), o_plperm = c.take(lsort ? nb * K * 4 : 0);
    // second copy of what the lean point_cells_kernel writes (batches only: stvo_seq::cells_buf)
    const bool cells2 = lsort && B >= 64;
    const size_t o_pstart2 = c.take(cells2 ? nb * (STVO_GRID_CELLS + 1) * 4 : 0), o_pperm2 = c.take(cells2 ? nb * K * 4 : 0),
                 o_pcell2 = c.take(cells2 ? nb * K * 4 : 0), o_plstart2 = c.take(cells2 ? nb * stvo::GRID_LSTART_STRIDE * 4 : 0),
                 o_plperm2 = c.take(cells2 ? nb * K * 4 : 0);
    const size_t o_lxy = c.take(nb * M * 4 * 4), o_lstart = c.take(nb * (STVO_GRID_CELLS + 1) * 4),
                 o_litems = c.take(nb * M * stvo::LENT * 4), o_lrank = c.take(nb * M * 4), o_lperm = c.take(nb * M * 4),
                 o_ldir = c.take(nb * M * 2 * 8);
    const int R = K > M ? K : M;
    const size_t o_cover = c.take(nb * (size_t)(R / 64) * R * 8), o_top2 = c.take(nb * R * 8), o_owner = c.take(nb * R * 4);
    const size_t o_elig = c.take(nb * stvo::GRID_ELIG * (size_t)K * 4), o_eligc = c.take(nb * K * 4), o_govf = c.take(nb * 8);  // ovf[nb] + misfit[nb]
    const size_t o_elig_l = c.take(nb * stvo::GRID_ELIG * (size_t)M * 4), o_eligc_l = c.take(nb * M * 4), o_govf_l = c.take(nb * 4);
    const size_t o_m12sp = c.take(nb * K * 4), o_m12sl = c.take(nb * M * 4), o_m12p = c.take(nb * K * 4), o_m12l = c.take(nb * M * 4),
                 o_inlp = c.take(nb * K * 4), o_inll = c.take(nb * M * 4), o_res = c.take(nb * sizeof(stvo_pose_result)),
                 o_counts = c.take(nb * 4 * 4);
    const bool pipe_ok = B >= 64;  // (pipelined steps: batches only)
    const size_t o_m12p_alt = c.take(pipe_ok ? nb * K * 4 : 0), o_m12l_alt = c.take(pipe_ok ? nb * M * 4 : 0);
    const size_t o_dyn_ctr = c.take(64), o_dyn_owner = c.take(nb * 4);
    const size_t cap_l = nb * (size_t)M * 4;  // line f2f: up to 4 train segments
    const size_t o_cover_l = c.take(nb * (size_t)(M / 64) * M * 8), o_top2_l = c.take(nb * M * 8), o_owner_l = c.take(nb * M * 4),
                 o_knn12_l = c.take(cap_l * 8), o_knn21_l = c.take(cap_l * 8), o_cand_l = c.take(nb * M * 4),
                 o_need_l = c.take(nb * M * 4), o_qsel_l = c.take(nb * M * 4), o_nsel_l = c.take(nb * 4 * 5);
    size_t o_set[3][14];
    for (int t = 0; t < 3; ++t) {
        o_set[t][0] = c.take(nb * K * 16);
        o_set[t][1] = o_set[t][2] = 0;
        o_set[t][3] = c.take(nb * K * 32);
        o_set[t][4] = c.take(nb * 4);
        o_set[t][5] = c.take(nb * M * 2 * 8);
        o_set[t][6] = c.take(nb * M * 2 * 8);
        o_set[t][7] = c.take(nb * M * 3 * 8);
        o_set[t][8] = c.take(nb * M * 3 * 8);
        o_set[t][9] = c.take(nb * M * 3 * 8);
        o_set[t][10] = c.take(nb * M * 8);
        o_set[t][11] = c.take(nb * M * 8);
        o_set[t][12] = c.take(nb * M * 32);
        o_set[t][13] = c.take(nb * 4);
    }
    s->dev_bytes = c.off;
    bool ok = hip_ok(ctx, hipMalloc((void**)&s->dev, s->dev_bytes), "hipMalloc seq") &&
              zero_device(ctx, s->dev, s->dev_bytes, "hipMemset seq") &&
              hip_ok(ctx, hipHostMalloc((void**)&s->raw_host[0], s->raw_bytes, hipHostMallocDefault), "hipHostMalloc seq") &&
              hip_ok(ctx, hipHostMalloc((void**)&s->raw_host[1], s->raw_bytes, hipHostMallocDefault), "hipHostMalloc seq") &&
              hip_ok(ctx, hipHostMalloc((void**)&s->out_host, nb * (sizeof(stvo_pose_result) + 16), hipHostMallocDefault),
                     "hipHostMalloc seq out") &&
              hip_ok(ctx, hipStreamCreateWithFlags(&s->line_stream, hipStreamNonBlocking), "hipStreamCreate seq") &&
              hip_ok(ctx, hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming), "hipEventCreate seq") &&
              hip_ok(ctx, hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming), "hipEventCreate seq") &&
              hip_ok(ctx, hipEventCreateWithFlags(&s->ev_cells, hipEventDisableTiming), "hipEventCreate seq") &&
              hip_ok(ctx, hipEventCreateWithFlags(&s->ev_upload, hipEventDisableTiming), "hipEventCreate seq") &&
              hip_ok(ctx, hipEventCreateWithFlags(&s->ev_match, hipEventDisableTiming), "hipEventCreate seq") &&
              hip_ok(ctx, hipEventCreateWithFlags(&s->ev_pose[0], hipEventDisableTiming), "hipEventCreate seq") &&
              hip_ok(ctx, hipEventCreateWithFlags(&s->ev_pose[1], hipEventDisableTiming), "hipEventCreate seq") &&
              hip_ok(ctx, hipEventCreateWithFlags(&s->ev_stage[0], hipEventDisableTiming), "hipEventCreate seq") &&
              hip_ok(ctx, hipEventCreateWithFlags(&s->ev_stage[1], hipEventDisableTiming), "hipEventCreate seq");
    s->zero_copy = B <= 16;
    if (ok && pipe_ok && !ctx->aux_stream) {  // pipelined steps: optimizePose on the context's aux stream (with its record arena)
        ok = hip_ok(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking), "hipStreamCreate aux");
        if (ok) stvo::pose_retain_stream(ctx->aux_stream);
    }
    {   // graph replay of the step chain: opt-in (STVO_SEQ_GRAPH=1).  Measured on ROCm 7.2 / MI355X for one sequence: 0.290 vs
        // 0.282 ms per frame points-only and 0.477 vs 0.310 ms with the line stage on its second stream — the graph executor
        // adds more per-node latency than the host-side launches cost (profiles/r02_single_stream_latency.txt)
        s->graph_mode = stvo::dbg().seq_graph != stvo::DBG_UNSET && stvo::dbg().seq_graph != 0;
    }
    if (!ok) {
        stvo_seq_destroy(s);
        return STVO_ERR_HIP;
    }
    std::memset(s->raw_host[0], 0, s->raw_bytes);
    std::memset(s->raw_host[1], 0, s->raw_bytes);
    char* D = s->dev;
    s->raw_dev = {D + o_raw, D + o_raw1};
    s->raw_lines.assign(2, 0);
    s->raw_max_lines.assign(2, 0);
    s->raw_split.assign(2, 0);
    s->d_cams = (stvo_cam*)(D + o_cams);
    s->d_inv_wh = (double*)(D + o_invwh);
    s->d_qtab = (double*)(D + o_qtab);
    s->d_join_flag = (unsigned*)(D + o_joinflag);
    s->d_pose_flag = (unsigned*)(D + o_poseflag);
    s->d_dyn_ctr = (int32_t*)(D + o_dyn_ctr);
    s->d_dyn_owner = (int32_t*)(D + o_dyn_owner);
    if (pipe_ok) {
        s->m12p_alt = (int32_t*)(D + o_m12p_alt);
        s->m12l_alt = (int32_t*)(D + o_m12l_alt);
    }
    {
        std::vector<double> iw(2 * nb);
        for (int b = 0; b < B; ++b) {
            iw[2 * b + 0] = STVO_GRID_COLS / (double)img_cols[b];  // stereoFrame.cpp:47-48
            iw[2 * b + 1] = STVO_GRID_ROWS / (double)img_rows[b];
        }
        double qtab[stvo::STVO_POSE_QTAB];  // sqrt(sigma2) per pyramid level: the same operations as the kernels' own computation
        for (int l = 0; l < stvo::STVO_POSE_QTAB; ++l) qtab[l] = std::sqrt(pm::level_sigma2(l, s->mp.orb_scale_factor));
        ok = upload_now(ctx, s->d_qtab, qtab, sizeof(qtab), "hipMemcpy qtab") &&
             upload_now(ctx, s->d_cams, cams, nb * sizeof(stvo_cam), "hipMemcpy cams") &&
             upload_now(ctx, s->d_inv_wh, iw.data(), iw.size() * sizeof(double), "hipMemcpy inv_wh");
        if (!ok) {
            stvo_seq_destroy(s);
            return STVO_ERR_HIP;
        }
    }
    stvo::SeqDev& d = s->d;
    d.B = B; d.K = K; d.M = M;
    d.cams = s->d_cams; d.inv_wh = s->d_inv_wh;
    d.mp = s->mp;
    d.pxy_l = (int32_t*)(D + o_pxy); d.pstart = (int32_t*)(D + o_pstart); d.pitems = (int32_t*)(D + o_pitems);
    d.prank = (int32_t*)(D + o_prank); d.pperm = (int32_t*)(D + o_pperm);
    d.lxy_l = (int32_t*)(D + o_lxy); d.lstart = (int32_t*)(D + o_lstart); d.litems = (int32_t*)(D + o_litems);
    d.lrank = (int32_t*)(D + o_lrank); d.lperm = (int32_t*)(D + o_lperm); d.ldir = (double*)(D + o_ldir);
    s->cover = (unsigned long long*)(D + o_cover); s->top2 = (unsigned long long*)(D + o_top2);
    s->owner2 = (int32_t*)(D + o_owner);
    s->elig = (uint32_t*)(D + o_elig); s->elig_cnt = (int32_t*)(D + o_eligc); s->govf = (int32_t*)(D + o_govf);
    s->elig_l = (uint32_t*)(D + o_elig_l); s->elig_cnt_l = (int32_t*)(D + o_eligc_l); s->govf_l = (int32_t*)(D + o_govf_l);
    s->cover_l = (unsigned long long*)(D + o_cover_l); s->top2_l = (unsigned long long*)(D + o_top2_l);
    s->owner2_l = (int32_t*)(D + o_owner_l);
    s->lazy_l = stvo::LazyScratch{(uint2*)(D + o_knn12_l), (uint2*)(D + o_knn21_l), (int32_t*)(D + o_cand_l), (int32_t*)(D + o_need_l),
                                  (int32_t*)(D + o_qsel_l), (int32_t*)(D + o_nsel_l), cap_l};
    s->m12s_p = (int32_t*)(D + o_m12sp); s->m12s_l = (int32_t*)(D + o_m12sl);
    s->m12p = (int32_t*)(D + o_m12p); s->m12l = (int32_t*)(D + o_m12l);
    s->inlp = (int32_t*)(D + o_inlp); s->inll = (int32_t*)(D + o_inll);
    s->results = (stvo_pose_result*)(D + o_res); s->counts = (int32_t*)(D + o_counts);
    s->m12_span = o_inlp - o_m12sp;  // m12s_p, m12s_l, m12p, m12l are carved back to back
    s->inl_span = o_res - o_inlp;    // inlp, inll likewise
    d.m12s_p = s->m12s_p; d.m12s_l = s->m12s_l;
    d.top2_p = s->top2; d.govf_p = s->govf; d.prange = (int32_t*)(D + o_prange); d.pcell = (int32_t*)(D + o_pcell);
    if (lsort) {
        d.plstart = (uint32_t*)(D + o_plstart); d.plperm = (int32_t*)(D + o_plperm);
    }
    s->cells_buf[0] = {d.pstart, d.pperm, d.pcell, d.plperm, d.plstart};
    s->cells_buf[1] = s->cells_buf[0];
    if (cells2)
        s->cells_buf[1] = {(int32_t*)(D + o_pstart2), (int32_t*)(D + o_pperm2), (int32_t*)(D + o_pcell2), (int32_t*)(D + o_plperm2),
                           (uint32_t*)(D + o_plstart2)};
    for (int t = 0; t < 3; ++t) {
        stvo_seq::Set& q = s->set[t];
        q.rc = (float4*)(D + o_set[t][0]);
        q.desc = (uint8_t*)(D + o_set[t][3]); q.n = (int32_t*)(D + o_set[t][4]);
        q.spl = (double*)(D + o_set[t][5]); q.epl = (double*)(D + o_set[t][6]); q.sP = (double*)(D + o_set[t][7]);
        q.eP = (double*)(D + o_set[t][8]); q.le = (double*)(D + o_set[t][9]); q.s2l = (double*)(D + o_set[t][10]);
        q.s2lm = (double*)(D + o_set[t][11]); q.ldesc = (uint8_t*)(D + o_set[t][12]); q.nl = (int32_t*)(D + o_set[t][13]);
    }
    *out = s;
    return STVO_OK;
}

int stvo_seq_create(stvo_ctx* ctx, int B, int max_keypoints, int max_keylines, int img_cols, int img_rows,
                    const stvo_cam* cam, const stvo_match_params* mp, const stvo_opt_params* op, stvo_seq** out) {
    if (B <= 0 || !cam) return STVO_ERR_INVALID_ARG;
    const std::vector<int32_t> cols((size_t)B, img_cols), rows((size_t)B, img_rows);
    const std::vector<stvo_cam> cams((size_t)B, *cam);
    return stvo_seq_create_multi(ctx, B, max_keypoints, max_keylines, cols.data(), rows.data(), cams.data(), mp, op, out);
}

// More raw frame slots than the two of stvo_seq_push: a throughput caller uploads n_slots consecutive frames of every
// sequence once and rotates stvo_seq_step_dev through them, so that the per-step working set is not the same two frames
// over and over.  Slots 0 / 1 keep their contents.
int stvo_seq_set_slots(stvo_seq* s, int n_slots) {
    if (!s || n_slots < 2 || n_slots > STVO_SEQ_MAX_SLOTS) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->aux_stream));
    // captured step graphs (STVO_SEQ_GRAPH=1) hold the raw-slot addresses of the slots they were captured for
    for (auto& g : s->graphs) (void)hipGraphExecDestroy(g.second);
    s->graphs.clear();
    if (s->extra_raw) {
        HIP_TRY(ctx, hipFree(s->extra_raw));
        s->extra_raw = nullptr;
    }
    s->raw_dev.resize(2);
    s->raw_lines.resize(2);
    s->raw_max_lines.resize(2);
    s->raw_split.assign(2, 0);
    if (n_slots > 2) {
        HIP_TRY(ctx, hipMalloc((void**)&s->extra_raw, (size_t)(n_slots - 2) * s->raw_bytes));
        if (!zero_device(ctx, s->extra_raw, (size_t)(n_slots - 2) * s->raw_bytes, "hipMemset seq raw")) return STVO_ERR_HIP;
        for (int k = 2; k < n_slots; ++k) {
            s->raw_dev.push_back(s->extra_raw + (size_t)(k - 2) * s->raw_bytes);
            s->raw_lines.push_back(0);
            s->raw_max_lines.push_back(0);
            s->raw_split.push_back(0);
        }
    }
    return STVO_OK;
}

int stvo_seq_destroy(stvo_seq* s) {
    if (!s) return STVO_OK;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->ctx->aux_stream) (void)hipStreamSynchronize(s->ctx->aux_stream);
    if (s->d_prof) (void)hipFree(s->d_prof);
    if (s->d_motion_T) (void)hipFree(s->d_motion_T);
    if (s->line_stream) {
        (void)hipStreamSynchronize(s->line_stream);
        (void)hipStreamDestroy(s->line_stream);
    }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->ev_cells) (void)hipEventDestroy(s->ev_cells);
    if (s->ev_match) (void)hipEventDestroy(s->ev_match);
    for (auto e : s->ev_pose)
        if (e) (void)hipEventDestroy(e);
    if (s->ev_upload) (void)hipEventDestroy(s->ev_upload);
    for (auto e : s->ev_stage)
        if (e) (void)hipEventDestroy(e);
    for (auto e : s->pev)
        if (e) (void)hipEventDestroy(e);
    for (auto e : s->tev) (void)hipEventDestroy(e);
    for (auto& g : s->graphs) (void)hipGraphExecDestroy(g.second);
    if (s->ev_fetch) (void)hipEventDestroy(s->ev_fetch);
    if (s->fetch_host) (void)hipHostFree(s->fetch_host);
    if (s->dev) (void)hipFree(s->dev);
    if (s->extra_raw) (void)hipFree(s->extra_raw);
    for (auto h : s->raw_host)
        if (h) (void)hipHostFree(h);
    if (s->out_host) (void)hipHostFree(s->out_host);
    delete s;
    return STVO_OK;
}

namespace {

// pipelined steps: pose kernels still in flight on the aux stream (everything they depend on was enqueued before them)
int seq_wait_pose_stream(stvo_seq* s) {
    stvo_ctx* ctx = s->ctx;
    if ((s->pose_pending[0] || s->pose_pending[1] || s->piped_last) && ctx->aux_stream) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->aux_stream));
        s->pose_pending[0] = s->pose_pending[1] = false;
    }
    return STVO_OK;
}

void bind_raw(stvo_seq* s, stvo::SeqDev& d, int slot) {
    char* Rw = s->raw_dev[slot];
    d.kp_l = (const float*)(Rw + s->off_kp_l); d.oct_l = (const int32_t*)(Rw + s->off_oct_l);
    d.desc_l = (const uint8_t*)(Rw + s->off_desc_l); d.n_kp_l = (const int32_t*)(Rw + s->off_nkl);
    d.kp_r = (const float*)(Rw + s->off_kp_r); d.desc_r = (const uint8_t*)(Rw + s->off_desc_r);
    d.n_kp_r = (const int32_t*)(Rw + s->off_nkr);
    d.kl_l = (const float*)(Rw + s->off_kl_l); d.oct_ll = (const int32_t*)(Rw + s->off_oct_ll);
    d.ldesc_l = (const uint8_t*)(Rw + s->off_ldesc_l); d.n_kl_l = (const int32_t*)(Rw + s->off_nll);
    d.kl_r = (const float*)(Rw + s->off_kl_r); d.ldesc_r = (const uint8_t*)(Rw + s->off_ldesc_r);
    d.n_kl_r = (const int32_t*)(Rw + s->off_nlr);
}

}  // namespace

// Copies one frame's features of all B sequences into device slot 0 / 1 (pinned gather + ONE H2D, asynchronous
// on the context's stream).
int stvo_seq_upload(stvo_seq* s, int slot, const stvo_frame_features* f) {
    if (!s || !f || slot < 0 || slot >= (int)s->raw_dev.size()) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int B = s->B, K = s->K, M = s->M;
    // contract: a positive count needs its arrays (checked before anything is touched)
    for (int b = 0; b < B; ++b) {
        const bool kl = f->n_kp_l && f->n_kp_l[b] > 0, kr = f->n_kp_r && f->n_kp_r[b] > 0;
        const bool ll = f->n_kl_l && f->n_kl_l[b] > 0 && s->op.has_lines, lr = f->n_kl_r && f->n_kl_r[b] > 0 && s->op.has_lines;
        if ((kl && (!f->kp_l || !f->oct_l || !f->desc_l)) || (kr && (!f->kp_r || !f->desc_r)) ||
            (ll && (!f->kl_l || !f->oct_ll || !f->ldesc_l)) || (lr && (!f->kl_r || !f->ldesc_r)))
            return STVO_ERR_INVALID_ARG;
    }
    // two pinned staging blocks used alternately: only wait for the copy that last read THIS block (two uploads ago)
    const int sb = s->stage_next;
    s->stage_next ^= 1;
    if (s->stage_upload[sb] > s->uploads_done) {  // everything enqueued so far covers that copy
        HIP_TRY(ctx, hipEventRecord(s->ev_stage[sb], ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(s->ev_stage[sb]));
        if (s->line_stream) HIP_TRY(ctx, hipStreamSynchronize(s->line_stream));  // (its share of a split copy reads the same block)
        s->uploads_done = s->uploads;
    }
    char* H = s->raw_host[sb];
    int32_t* nkl = (int32_t*)(H + s->off_nkl);
    int32_t* nkr = (int32_t*)(H + s->off_nkr);
    int32_t* nll = (int32_t*)(H + s->off_nll);
    int32_t* nlr = (int32_t*)(H + s->off_nlr);
    int max_lines = 0;
    for (int b = 0; b < B; ++b) {
        const int a = f->n_kp_l ? f->n_kp_l[b] : 0, r = f->n_kp_r ? f->n_kp_r[b] : 0;
        const int la = (f->n_kl_l && s->op.has_lines) ? f->n_kl_l[b] : 0, lr = (f->n_kl_r && s->op.has_lines) ? f->n_kl_r[b] : 0;
        if (a < 0 || r < 0 || la < 0 || lr < 0 || a > K || r > K || la > M || lr > M || a > f->stride_kp || r > f->stride_kp ||
            la > f->stride_kl || lr > f->stride_kl)
            return STVO_ERR_CAPACITY;
        nkl[b] = s->op.has_points ? a : 0;
        nkr[b] = s->op.has_points ? r : 0;
        nll[b] = la;
        nlr[b] = lr;
        max_lines = std::max(max_lines, std::max(la, lr));
        const size_t so = (size_t)b * f->stride_kp, dof = (size_t)b * K;
        if (nkl[b]) {
            std::memcpy(H + s->off_kp_l + dof * 8, f->kp_l + so * 2, (size_t)a * 8);
            std::memcpy(H + s->off_oct_l + dof * 4, f->oct_l + so, (size_t)a * 4);
            std::memcpy(H + s->off_desc_l + dof * 32, f->desc_l + so * 32, (size_t)a * 32);
        }
        if (nkr[b]) {
            std::memcpy(H + s->off_kp_r + dof * 8, f->kp_r + so * 2, (size_t)r * 8);
            std::memcpy(H + s->off_desc_r + dof * 32, f->desc_r + so * 32, (size_t)r * 32);
        }
        const size_t sl = (size_t)b * f->stride_kl, dl = (size_t)b * M;
        if (la) {
            std::memcpy(H + s->off_kl_l + dl * 16, f->kl_l + sl * 4, (size_t)la * 16);
            std::memcpy(H + s->off_oct_ll + dl * 4, f->oct_ll + sl, (size_t)la * 4);
            std::memcpy(H + s->off_ldesc_l + dl * 32, f->ldesc_l + sl * 32, (size_t)la * 32);
        }
        if (lr) {
            std::memcpy(H + s->off_kl_r + dl * 16, f->kl_r + sl * 4, (size_t)lr * 16);
            std::memcpy(H + s->off_ldesc_r + dl * 32, f->ldesc_r + sl * 32, (size_t)lr * 32);
        }
    }
    bool any_lines = false;
    for (int b = 0; b < B; ++b) any_lines = any_lines || (nll[b] > 0 && nlr[b] > 0);
    s->raw_lines[slot] = any_lines;
    s->raw_max_lines[slot] = max_lines;
    s->raw_split[slot] = 0;
    if (s->raw_bytes <= (size_t)4 << 20) {
        // copy kernel instead of the DMA engine: ~5 us less latency for the ~200 KB of one frame
        if (any_lines && s->op.has_points && !s->st_dirty && !s->graph_mode && s->line_stream) {
            stvo::launch_copy16(ctx->stream, H, s->raw_dev[slot], s->off_kl_l);
            stvo::launch_copy16(s->line_stream, H + s->off_kl_l, s->raw_dev[slot] + s->off_kl_l, s->raw_bytes - s->off_kl_l);
            s->raw_split[slot] = 1;
        } else {
            stvo::launch_copy16(ctx->stream, H, s->raw_dev[slot], s->raw_bytes);
        }
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(s->raw_dev[slot], H, s->raw_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    s->stage_upload[sb] = ++s->uploads;
    if (s->B >= 64 && s->line_stream) {  // batches: the line stream may read a slot first (stvo_seq::ev_cells); no event in single-stream operation
        HIP_TRY(ctx, hipEventRecord(s->ev_upload, ctx->stream));
        ++s->upload_seq;
    }
    return STVO_OK;
}

// The same for features that are already in device memory (every pointer of `f`, the count arrays included, is a DEVICE
// pointer; oct_l / oct_ll may be NULL = octave 0).  Enqueued on the context's stream, no host transfer, no synchronisation:
// counts are clamped to the capacities on the device instead of being validated on the host.
int stvo_seq_upload_dev(stvo_seq* s, int slot, const stvo_frame_features* f) {
    if (!s || !f || slot < 0 || slot >= (int)s->raw_dev.size()) return STVO_ERR_INVALID_ARG;
    if (f->stride_kp < 0 || f->stride_kl < 0) return STVO_ERR_INVALID_ARG;
    const bool pts = s->op.has_points && f->n_kp_l && f->n_kp_r, lns = s->op.has_lines && f->n_kl_l && f->n_kl_r;
    if ((pts && (!f->kp_l || !f->desc_l || !f->kp_r || !f->desc_r)) || (lns && (!f->kl_l || !f->ldesc_l || !f->kl_r || !f->ldesc_r)))
        return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    char* Rw = s->raw_dev[slot];
    IngestArgs a{};
    a.B = s->B; a.K = s->K; a.M = s->M; a.has_points = pts; a.has_lines = lns; a.f = *f;
    a.kp_l = (float*)(Rw + s->off_kp_l); a.oct_l = (int32_t*)(Rw + s->off_oct_l); a.desc_l = (uint8_t*)(Rw + s->off_desc_l);
    a.n_kp_l = (int32_t*)(Rw + s->off_nkl); a.kp_r = (float*)(Rw + s->off_kp_r); a.desc_r = (uint8_t*)(Rw + s->off_desc_r);
    a.n_kp_r = (int32_t*)(Rw + s->off_nkr); a.kl_l = (float*)(Rw + s->off_kl_l); a.oct_ll = (int32_t*)(Rw + s->off_oct_ll);
    a.ldesc_l = (uint8_t*)(Rw + s->off_ldesc_l); a.n_kl_l = (int32_t*)(Rw + s->off_nll); a.kl_r = (float*)(Rw + s->off_kl_r);
    a.ldesc_r = (uint8_t*)(Rw + s->off_ldesc_r); a.n_kl_r = (int32_t*)(Rw + s->off_nlr);
    hipLaunchKernelGGL(seq_ingest_kernel, dim3(s->B), dim3(256), 0, ctx->stream, a);
    if (s->B >= 64 && s->line_stream) {
        HIP_TRY(ctx, hipEventRecord(s->ev_upload, ctx->stream));
        ++s->upload_seq;
    }
    s->st_dirty = true;
    s->raw_split[slot] = 0;
    s->raw_max_lines[slot] = s->M;
    s->raw_lines[slot] = lns;  // the line stage runs whenever line arrays were given (empty sets cost two small launches)
    return check_launch(ctx);
}

// Runs the whole per-frame pipeline on the features resident in `slot` (asynchronous; no host transfer).
namespace {

struct StepFlags {
    bool lines_now, lines_prev, track;
};

// Enqueues the kernel chain of one step on the context's stream (and the line stream).  No state of `s` changes here, so
// the same code serves direct launches and stream capture into a hipGraph.
int seq_enqueue_step(stvo_seq* s, int slot, const StepFlags& fl, bool* forked_by_event = nullptr) {
    stvo_ctx* ctx = s->ctx;
    const int B = s->B, K = s->K, M = s->M;
    hipStream_t st = ctx->stream;
    // live stage timing: STVO_SEQ_NSTAGE event pairs per step (see stvo_seq_get_stage_timing)
    hipEvent_t* tev = nullptr;
    if (s->timing) {
        while (s->tev.size() < s->tev_used + 2 * STVO_SEQ_NSTAGE) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) break;
            s->tev.push_back(e);
        }
        if (s->tev.size() >= s->tev_used + 2 * STVO_SEQ_NSTAGE) {
            tev = s->tev.data() + s->tev_used;
            s->tev_used += 2 * STVO_SEQ_NSTAGE;
        }
    }
    // light timing (stvo_seq_set_stage_timing(.., 2)): event pairs around the grid matcher, the forward scan and the pose kernel only, on
    // the point stream; the key-line stream runs unmarked and the next frame's grid is still built ahead on it, so the three kernels
    // keep the neighbours they have in an untimed step (bench.py: `roofline` must come from the region that produced `value`)
    const bool light = tev && s->timing == 2;
    auto mark = [&](int k, hipStream_t q) {
        if (tev && !(light && k < 2)) (void)hipEventRecord(tev[k], q);
    };
    // ---- pipelined step (stvo_seq: "pipelined steps"): this step's pose kernel goes to the aux stream and the NEXT step's association may
    // start beside it.  Batches on the compact-record batch kernel only; not with the by-product fetch (its copies follow the pose kernel in
    // the point stream), the full stage timers, the host-side profile markers or graph replay.
    // (round 6: built, parity-green, measured — no gain: the matcher's workgroups need whole CUs, which the pose kernel frees only at its very
    //  end; profiles/r06_pipelined_steps.txt.  Opt-in: STVO_SEQ_PIPE=1, 2 = without the gate.)
    const int pipe_sw = stvo::dbg().seq_pipe;
    const bool piped = s->m12p_alt != nullptr && ctx->aux_stream != nullptr && !ctx->overlap && !s->fetch && !s->graph_mode && !s->pev[0] &&
                       !(tev && !light) && !s->zero_copy && (pipe_sw == 1 || pipe_sw == 2);
    // what a pose kernel on the aux stream still reads — the stereo set this step overwrites (it was `prev` two steps ago) and the f2f
    // match indices of this step's parity — is free once the pose kernel of two steps ago has finished: long ago, in the steady state
    {
        const int par2 = s->frame_idx & 1;
        if (s->pose_pending[par2]) {
            HIP_TRY(ctx, hipStreamWaitEvent(st, s->ev_pose[par2], 0));
            s->pose_pending[par2] = false;
        }
        if (!piped && s->pose_pending[par2 ^ 1]) {  // leaving the pipelined mode: this step's pose kernel follows the last one in stream order
            HIP_TRY(ctx, hipStreamWaitEvent(st, s->ev_pose[par2 ^ 1], 0));
            s->pose_pending[par2 ^ 1] = false;
        }
    }
    // two copies of the f2f match indices, by the parity of the step (batches without the by-product fetch, whose copies read the first):
    // the matches of step k + 1 may be written while optimizePose(k) reads those of step k
    const bool alt_ok = s->m12p_alt != nullptr && !s->fetch;
    int32_t* const m12p_use = (alt_ok && (s->frame_idx & 1)) ? s->m12p_alt : s->m12p;
    int32_t* const m12l_use = (alt_ok && (s->frame_idx & 1)) ? s->m12l_alt : s->m12l;
    // ---- stereo association of the new frame into set[cur]
    stvo_seq::Set& cs = s->set[s->cur];
    stvo_seq::Set& ps = s->set[s->prev_set()];
    stvo::SeqDev d = s->d;
    bind_raw(s, d, slot);
    d.rc = cs.rc; d.desc = cs.desc; d.n = cs.n;
    d.spl = cs.spl; d.epl = cs.epl; d.sP = cs.sP; d.eP = cs.eP; d.le = cs.le; d.s2l = cs.s2l; d.s2lm = cs.s2lm;
    d.ldesc = cs.ldesc; d.nl = cs.nl;
    const size_t res_bytes = (size_t)B * sizeof(stvo_pose_result);
    d.host_n = s->zero_copy ? reinterpret_cast<int32_t*>(s->out_host + res_bytes) : nullptr;
    d.host_nl = s->zero_copy ? reinterpret_cast<int32_t*>(s->out_host + res_bytes + (size_t)B * 4) : nullptr;
    // a frame without key-lines skips the whole line stage (7 launches) and, below, the f2f line matching (6)
    const bool lines_now = fl.lines_now, lines_prev = fl.lines_prev;
    // fork: everything enqueued so far (ingest, the previous step) happens-before the line stream's work
    const bool par = lines_now && s->op.has_points;
    hipStream_t sl = par ? s->line_stream : st;
    // Where the line stream forks off.  Every big kernel of the point stream fills the register file of the CUs it runs on, so work
    // of the line stream never runs BESIDE it, only instead of it.  Forked at the start of the step (default) the line kernels share
    // the GPU with point_cells_kernel and delay the start of some of the persistent point matcher's workgroups (0.158 -> 0.198 ms
    // per 1024 frames) but leave the key-point scan alone (0.467 ms); forked after the point stage (STVO_LINE_FORK=late) they
    // stretch the scan instead (0.510 ms).  Measured: 1024 KITTI-shaped streams 929 k (early) vs 935 k (late) frame pairs/s, 512
    // EuRoC-shaped streams 857 k vs 777 k, one stream 0.252 vs 0.280 ms per frame.
    // Round 5: a third point — behind the cells kernel (STVO_LINE_FORK=mid), the default for batches: the line kernels then become
    // ready together with the persistent point matcher, whose workgroups (already queued) take their CUs first, instead of finding four
    // line workgroups per CU in their way.  1024 KITTI-shaped streams 0.872 -> 0.853 ms per step, 512 EuRoC-shaped 1.036 -> 1.105 M
    // frame pairs/s; one stream is SLOWER that way (0.205 -> 0.211 ms: its line kernels lose their head start), so small batches keep
    // the fork at the start.  STVO_LINE_FORK=start / mid / late forces one of the three.
    const int fork_sw = stvo::dbg().line_fork_late;  // DBG_UNSET: by batch size
    const bool late_fork = par && fork_sw == 1;
    const bool mid_fork = par && s->op.has_points && (fork_sw == 2 || (fork_sw == stvo::DBG_UNSET && B >= 64));
    const bool fork_free = par && s->raw_split[slot] && !s->st_dirty && !s->graph_mode;  // see stvo_seq::st_dirty
    if (forked_by_event) *forked_by_event = par && (late_fork || mid_fork || !fork_free);
    if (par && !late_fork && !mid_fork && !fork_free) {
        HIP_TRY(ctx, hipEventRecord(s->ev_fork, st));
        HIP_TRY(ctx, hipStreamWaitEvent(sl, s->ev_fork, 0));
    }
    d.zero_nl = (!lines_now && s->op.has_points) ? 1 : 0;
    if (s->pev[0]) (void)hipEventRecord(s->pev[1], st);  // pev[0] was recorded before the ingest
    int stage_rc = STVO_OK;
    auto point_stage = [&]() -> int {
        if (s->op.has_points) {
            mark(0, st);
            stvo::GridBatch g;
            std::memset(&g, 0, sizeof(g));
            g.B = B; g.stride1 = K; g.stride2 = K; g.xy_width = 2; g.items_stride = K; g.words64 = K / 64; g.n1p = K;
            g.cell_xy1 = d.pxy_l; g.d1 = d.desc_l; g.n1 = d.n_kp_l; g.cell_start = d.pstart; g.cell_items = d.pitems;
            g.d2 = d.desc_r; g.n2 = d.n_kp_r; g.dir2 = nullptr;
            g.w = stvo_grid_window{s->mp.matching_s_ws, 0, 0, 0};  // stereoFrame.cpp:141-143
            g.ratio = s->ratio_grid; g.line_sim_th = 0.0; g.mutual = s->mp.best_lr_matches;
            g.cover = s->cover; g.rank = d.prank; g.perm = d.pperm; g.top2 = s->top2; g.owner2 = s->owner2; g.m12 = s->m12s_p;
            if (g.mutual) { g.elig = s->elig; g.elig_cnt = s->elig_cnt; g.ovf = s->govf; }
            g.misfit = s->govf + s->B;
            g.range_points = 1;  // device CSR: right key-points are numbered in cell order, one grid row per window
            g.range1 = d.prange;
            g.cell2 = d.pcell;
            g.lstart = d.plstart; g.lperm = d.plperm;
            // the one-workgroup matcher: lean cells kernel in front, the tail of the association as its last phase
            g.lean_cells = stvo::grid_points_fused_ok(g) ? 1 : 0;
            g.has_tail = g.lean_cells;
            if (stvo::dbg().grid_tail == 0) g.has_tail = 0;  // developer: point_tail_kernel as its own launch
            if (g.has_tail) g.tail = stvo::point_tail_args(d);
            // one frame per workgroup of the matcher (single-stream operation, small batches): the grid of the frame is the matcher's
            // first phase — one dependent launch less in the chain of a frame
            g.fused_cells = g.lean_cells && B <= stvo::device_cu_count() && stvo::dbg().grid_cells != 0;
            // Batches: the grid of this frame on the LINE stream, which has been idle since ~0.2 ms into the previous step — the kernel (38 us
            // per 1024 frames; few instructions, mostly waiting) then runs beside the previous step's forward scan or pose kernel and
            // the point stream meets it with one awaited event in front of the matcher.  Safe because (a) its outputs are double-buffered
            // by the parity of the step (the matcher of the previous frame may still read the other copy; the copy written here was
            // last read two steps ago, and the line stream's work of the previous step waited for an event the point stream recorded
            // after that: sl_forked_frame), (b) the line stream has been made to wait for every upload enqueued on the point stream
            // (stvo_seq_step_dev), (c) everything else the kernel reads is the resident slot.  STVO_CELLS_AHEAD=0: in the point stream.
            const bool cells_ahead = par && mid_fork && g.lean_cells && !g.fused_cells && (!tev || light) && !s->pev[0] && !s->graph_mode &&
                                     s->cells_buf[0].pstart != s->cells_buf[1].pstart && s->sl_forked_frame == (long long)s->frame_idx - 1 &&
                                     stvo::dbg().cells_ahead != 0;
            // key-line stage ahead (stvo_seq: fork_rec_frame): the line stream waits for the PREVIOUS step's fork event — in front of the
            // cells kernel, whose output copy the matcher of two steps ago read — and not for this step's
            // Only where it was measured to pay: ~100 key-lines per image (their kernels are a few per cent of the step) behind a pose kernel
            // that publishes its start, i.e. the two-waves-per-pair batch kernel.  With hundreds of key-lines per image the line kernels are
            // long enough to hold the pose kernel's freed slots against the next matcher: 512 EuRoC-shaped streams 0.442 -> 0.520 ms per step
            // (round 6), so those keep the fork of their own step.  STVO_LINES_AHEAD=1 forces it wherever it is safe.
            const int la_sw = stvo::dbg().lines_ahead;
            const bool la_pays = s->pose_flag_frame == (long long)s->frame_idx - 1 && B > 2 * stvo::device_cu_count() &&
                                 std::max(s->raw_max_lines[slot], s->set_lines_cap[s->prev_set()]) <= 128;
            const bool lines_ahead = cells_ahead && !piped && alt_ok && s->fork_rec_frame == (long long)s->frame_idx - 1 && la_sw != 0 &&
                                     (la_sw == 1 || la_sw == 2 || la_pays);
            // (STVO_LINES_AHEAD=2: the gate in front of the cells kernel too — experiment)
            const bool gate_cells = lines_ahead && stvo::dbg().lines_ahead == 2 && s->pose_flag_frame == (long long)s->frame_idx - 1;
            if (lines_ahead) HIP_TRY(ctx, hipStreamWaitEvent(sl, s->ev_fork, 0));
            if (gate_cells) stvo::launch_stream_gate(sl, s->d_pose_flag, s->pose_epoch);
            if (g.fused_cells)
                g.cells = stvo::point_cells_args(d);
            else if (g.lean_cells)
                hipLaunchKernelGGL(stvo::point_cells_kernel<true>, dim3(B), dim3(256), 0, cells_ahead ? sl : st, d);
            else
                hipLaunchKernelGGL(stvo::point_cells_kernel<false>, dim3(B), dim3(256), 0, st, d);
            if (cells_ahead) {
                HIP_TRY(ctx, hipEventRecord(s->ev_cells, sl));
                HIP_TRY(ctx, hipStreamWaitEvent(st, s->ev_cells, 0));
            }
            // light timing: the matcher's start stamp in FRONT of the fork — a barrier packet between the fork and the matcher would give the
            // key-line kernels a head start on the CUs, which the untimed step does not give them (first GPU call of round 6: 0.226 ms
            // instead of the timed region's 0.139)
            hipEvent_t gev_light[2] = {nullptr, tev ? tev[3] : nullptr};
            if (light) (void)hipEventRecord(tev[2], st);
            if (mid_fork) {
                HIP_TRY(ctx, hipEventRecord(s->ev_fork, st));
                if (!s->graph_mode) s->fork_rec_frame = s->frame_idx;
                if (!lines_ahead)
                    HIP_TRY(ctx, hipStreamWaitEvent(sl, s->ev_fork, 0));
                else if (!gate_cells && s->pose_flag_frame == (long long)s->frame_idx - 1)  // the key-line kernels behind the dispatch of optimizePose(k - 1)
                    stvo::launch_stream_gate(sl, s->d_pose_flag, s->pose_epoch);
            }
            // pipelined steps: the persistent matcher starts beside the previous step's pose kernel — frames by ticket (GridBatch::dyn_ctr)
            if (piped && fl.track && g.lean_cells && !g.fused_cells && stvo::dbg().grid_dyn != 0) {
                g.dyn_ctr = s->d_dyn_ctr;
                g.dyn_owner = s->d_dyn_owner;
                g.dyn_par = s->frame_idx & 1;
                if (s->dyn_last_frame != (long long)s->frame_idx - 1)  // nobody reset this launch's counter
                    HIP_TRY(ctx, hipMemsetAsync(s->d_dyn_ctr, 0, 2 * sizeof(int32_t), st));
                s->dyn_last_frame = s->frame_idx;
            }
            s->last_point_grid = g;
            stvo::launch_grid_batch(st, g, false, tev ? (light ? gev_light : tev + 2) : nullptr);
            if (!g.has_tail) hipLaunchKernelGGL(stvo::point_tail_kernel, dim3(B), dim3(stvo::TAIL_BLOCK), 0, st, d);
            mark(1, st);
        } else {
            HIP_TRY(ctx, hipMemsetAsync(cs.n, 0, (size_t)B * 4, st));
        }
        return STVO_OK;
    };
    auto line_stage = [&]() -> int {
        if (lines_now) {
            stvo::GridBatch g;
            std::memset(&g, 0, sizeof(g));
            g.B = B; g.stride1 = M; g.stride2 = M; g.xy_width = 4; g.items_stride = M * stvo::LENT; g.words64 = M / 64; g.n1p = M;
            g.cell_xy1 = d.lxy_l; g.d1 = d.ldesc_l; g.n1 = d.n_kl_l; g.cell_start = d.lstart; g.cell_items = d.litems;
            g.d2 = d.ldesc_r; g.n2 = d.n_kl_r; g.dir2 = d.ldir;
            g.w = stvo_grid_window{s->mp.matching_s_ws, 0, 0, 0};  // :340-342
            g.ratio = s->ratio_grid /* sic, minRatio12P: matching.cpp:241 */; g.line_sim_th = s->mp.line_sim_th;
            g.mutual = s->mp.best_lr_matches;
            g.cover = s->cover_l; g.rank = d.lrank; g.perm = d.lperm; g.top2 = s->top2_l; g.owner2 = s->owner2_l; g.m12 = s->m12s_l;
            if (g.mutual) { g.elig = s->elig_l; g.elig_cnt = s->elig_cnt_l; g.ovf = s->govf_l; }
            s->last_line_grid = g;
            // few key-lines per frame: the whole association in one workgroup per frame (STVO_LINE_FUSED=0: the general grid matcher)
            // LDS for the lines the slot holds, not for the capacity — except under graph replay: a captured step is replayed for
            // later uploads into the slot, whose line counts the capture cannot know, so it is sized for the capacity (and so are
            // the caps derived from set_lines_cap below)
            const int Mk = s->graph_mode ? M : std::min(M, std::max(64, (s->raw_max_lines[slot] + 63) & ~63));
            s->set_lines_cap[s->cur] = Mk;
            const size_t lds = (size_t)Mk * stvo::LSF_BYTES_PER_LINE + 4 + (size_t)Mk * (Mk / 32) * 4;
            const int ef = stvo::dbg().line_fused;
            // (a single stream with hundreds of key-lines is better off with the general matcher's many small workgroups: EuRoC-shaped,
            // 300 key-lines, one stream 0.310 vs 0.360 ms per frame; 102 key-lines 0.257 vs 0.252)
            s->last_line_fused = M <= stvo::LSF_MAX_LINES && (ef != stvo::DBG_UNSET ? ef != 0 : (B >= 16 || Mk <= 128)) &&
                                 (lds <= (48u << 10) || stvo::lds_opt_in(reinterpret_cast<const void*>(stvo::line_stereo_fused_kernel<256>), (int)lds));
            if (s->last_line_fused) {
                // (one wave per frame, <64>: 185 instead of 150 us beside the key-point scan, which it stretched by 15 us more)
                hipLaunchKernelGGL(stvo::line_stereo_fused_kernel<256>, dim3(B), dim3(256), lds, sl, d, Mk, (int)s->mp.best_lr_matches, s->ratio_grid);
            } else {
                hipLaunchKernelGGL(stvo::line_cells_kernel, dim3(B), dim3(256), 0, sl, d);
                stvo::launch_grid_batch(sl, g, true);
                hipLaunchKernelGGL(stvo::line_tail_kernel, dim3(B), dim3(256), 0, sl, d);
            }
        } else if (!d.zero_nl) {
            HIP_TRY(ctx, hipMemsetAsync(cs.nl, 0, (size_t)B * 4, st));
        }
        return STVO_OK;
    };
    // enqueue order: STVO_LINE_FIRST=1 puts the line kernel in front of the point stage (experiment: the point matcher then starts on
    // free CUs, the cells kernel shares them — same step time on 1024 KITTI-shaped streams, slower with hundreds of lines per image)
    const bool line_first = par && !late_fork && stvo::dbg().line_first == 1;
    if (line_first && (stage_rc = line_stage()) != STVO_OK) return stage_rc;
    if ((stage_rc = point_stage()) != STVO_OK) return stage_rc;
    if (late_fork) {
        HIP_TRY(ctx, hipEventRecord(s->ev_fork, st));
        HIP_TRY(ctx, hipStreamWaitEvent(sl, s->ev_fork, 0));
    }
    if (!line_first && (stage_rc = line_stage()) != STVO_OK) return stage_rc;
    if (s->pev[0]) (void)hipEventRecord(s->pev[2], st);
    const bool track = fl.track;
    if (track) {
        // ---- f2fTracking: prev stereo sets vs curr stereo sets
        const stvo::LazyScratch w{ctx->knn12, ctx->knn21, ctx->cand, ctx->need, ctx->qsel, ctx->nsel, ctx->knn_capacity};
        const int esm = stvo::dbg().match_small;  // developer: 0 = the general machinery for the key-line sets too
        const int lines_cap = std::max(s->set_lines_cap[s->cur], s->set_lines_cap[s->prev_set()]);
        // one workgroup per frame pair (match_small_kernel) up to 128 key-lines per image; beyond that its row-by-row scan is the
        // longest thing on the key-line stream and the general machinery (K1m + planned reverse check, five launches) wins for every
        // batch size: 512 EuRoC-shaped streams with ~250 key-lines 844 k -> 913 k frame pairs/s (both directions in one K1m launch +
        // one ratio / mutual kernel: 895 k), one such stream 0.360 -> 0.310 ms (round 3)
        const bool small_sets = esm != stvo::DBG_UNSET ? esm != 0 : lines_cap <= 128;
        const bool elz = stvo::dbg().match_lazy == 1;  // developer: the lazy formulation for small batches too
        int small_cap = 0;  // rows per set the small-set kernel sizes its LDS for (0: the stride)
        auto match_set = [&](hipStream_t q, const stvo::LazyScratch& ws, int stride, const uint8_t* da, const int32_t* na,
                             const uint8_t* db, const int32_t* nb, float nnr, int32_t* m12, hipEvent_t* mev) {
            if (stvo::match_small_ok(stride) && small_sets && !mev) {  // (mev: the stage timers want the general launches)
                stvo::launch_match_small(q, B, stride, da, na, db, nb, nnr, s->mp.best_lr_matches, m12, small_cap);
            } else if (s->mp.best_lr_matches && B <= 4 && !mev && !elz) {
                // a few frame pairs leave most of the GPU idle: the reverse direction as a full scan in the SAME launch and one
                // ratio / mutual kernel, instead of the plan + two selective reverse scans + final check of the lazy formulation
                // (five dependent launches: 42 -> ~18 us of a single stream's 230 us)
                // (four train segments, not the 16 a single direction gets: both directions already double the workgroups, and the
                // ratio / mutual kernel merges 2 x nseg partial keys per row — one stream 0.2465 -> 0.2357 ms)
                const int nseg = std::min(4, stvo::knn_pick_nseg(B, stride, ws.knn_capacity));
                stvo::launch_hamming_knn2(q, B, stride, stride, da, na, db, nb, ws.knn12, ws.knn21, 1, 0, 0, nullptr, nullptr, nseg);
                stvo::launch_nnr_mutual(q, B, stride, ws.knn12, ws.knn21, na, nb, nnr, 1, m12, nseg);
            } else if (s->mp.best_lr_matches) {
                stvo::launch_match_mutual_lazy(q, B, stride, da, na, db, nb, nnr, ws, m12, 0, nullptr, mev);
            } else {
                const int nseg = stvo::knn_pick_nseg(B, stride, ws.knn_capacity);
                stvo::launch_hamming_knn2(q, B, stride, stride, da, na, db, nb, ws.knn12, ws.knn21, 0, 0, 0, nullptr, nullptr, nseg);
                stvo::launch_nnr_mutual(q, B, stride, ws.knn12, ws.knn21, na, nb, nnr, 0, m12, nseg);
            }
        };
        hipEvent_t mev_light[4] = {tev ? tev[4] : nullptr, tev ? tev[5] : nullptr, nullptr, nullptr};  // (light: no pair around plan + reverse scans)
        if (s->op.has_points) match_set(st, w, K, ps.desc, ps.n, cs.desc, cs.n, s->mp.min_ratio_12_p, m12p_use, tev ? (light ? mev_light : tev + 4) : nullptr);
        small_cap = lines_cap;
        if (lines_prev && lines_now)
            match_set(sl, s->lazy_l, M, ps.ldesc, ps.nl, cs.ldesc, cs.nl, s->mp.min_ratio_12_l, m12l_use, nullptr);
        else if (lines_prev)  // nothing to match against: every prev line is unmatched
            HIP_TRY(ctx, hipMemsetAsync(m12l_use, 0xFF, (size_t)B * M * sizeof(int32_t), st));
        // Small batches (single-stream operation): the pose kernel itself waits for the line stream and hands the match indices to
        // the host — an event awaited or recorded in front of it delays its start by ~6 us each on this runtime.
        const bool inline_sync = stvo::pose_inline_sync_ok(B) && stvo::dbg().seq_inline != 0 && !s->graph_mode && !tev && !s->pev[0] &&
                                 (!s->fetch || (B == 1 && s->zero_copy));
        hipStream_t sp = piped ? ctx->aux_stream : st;  // the stream of optimizePose
        if (piped) {  // the pose kernel waits for both streams' matches; the point stream does not wait for the key-line stream at all
            HIP_TRY(ctx, hipEventRecord(s->ev_match, st));
            HIP_TRY(ctx, hipStreamWaitEvent(sp, s->ev_match, 0));
        }
        if (par && inline_sync) {
            stvo::launch_stream_signal(sl, s->d_join_flag, ++s->join_epoch);
        } else if (par) {  // join before optimizePose
            HIP_TRY(ctx, hipEventRecord(s->ev_join, sl));
            HIP_TRY(ctx, hipStreamWaitEvent(sp, s->ev_join, 0));
        }
        if (s->pev[0]) (void)hipEventRecord(s->pev[3], st);
        s->fetch_by_pose = s->fetch && inline_sync;
        if (s->fetch && !inline_sync) {
            stvo::launch_copy16(st, s->m12s_p, s->fetch_host, s->m12_span);
            HIP_TRY(ctx, hipEventRecord(s->ev_fetch, st));
        }
        // ---- optimizePose
        stvo::PoseArgs a{};
        std::memset(&a, 0, sizeof(a));
        a.B = B; a.max_pts = K; a.max_lines = M;
        a.n_prev_pts = s->op.has_points ? ps.n : nullptr;
        a.prev_rc = ps.rc; a.curr_rc = cs.rc; a.q_tab = s->d_qtab; a.level_scale = s->mp.orb_scale_factor; a.m12p = m12p_use;
        a.n_prev_lines = s->op.has_lines ? ps.nl : nullptr;
        a.prev_sP = ps.sP; a.prev_eP = ps.eP; a.prev_spl = ps.spl; a.prev_epl = ps.epl; a.prev_s2l = ps.s2lm;
        a.curr_le = cs.le; a.m12l = m12l_use;
        a.cams = s->d_cams; a.prm = s->op;
        a.init_T = s->d_motion_T; a.next_T = s->d_motion_T;  // (nullptr: DT = I, use_motion_model = false)
        a.results = s->zero_copy ? reinterpret_cast<stvo_pose_result*>(s->out_host) : s->results;
        a.inl_p_out = s->inlp; a.inl_l_out = s->inll;
        // small batches with the by-product fetch on (the StereoFrameHandler mirror): the pose kernel writes the inlier masks straight
        // into the pinned block — like its result — instead of a copy kernel behind it (one launch less on the single-stream chain)
        const bool inl_zero_copy = s->fetch && s->zero_copy;
        if (inl_zero_copy) {
            a.inl_p_out = reinterpret_cast<int32_t*>(s->fetch_host + s->m12_span);
            a.inl_l_out = reinterpret_cast<int32_t*>(s->fetch_host + s->m12_span + (reinterpret_cast<const char*>(s->inll) - reinterpret_cast<const char*>(s->inlp)));
        }
        if (stvo::dbg().pose_prof != stvo::DBG_UNSET) {
            if (!s->d_prof) HIP_TRY(ctx, hipMalloc((void**)&s->d_prof, (size_t)B * 16 * sizeof(long long)));
            a.prof_out = s->d_prof;
        }
        if (par && inline_sync) {
            a.wait_flag = s->d_join_flag;
            a.wait_value = s->join_epoch;
        }
        // single-stream operation: the eigenvalues of the committed covariance (an output only) are computed by stvo_seq_read
        a.lazy_eig = (inline_sync && s->zero_copy && stvo::dbg().pose_kernel != 4) ? 1 : 0;
        if (s->fetch_by_pose) {
            a.fetch_src = reinterpret_cast<const uint4*>(s->m12s_p);
            a.fetch_dst = reinterpret_cast<uint4*>(s->fetch_host);
            a.fetch_n16 = (unsigned)(s->m12_span / 16);
            a.fetch_flag = reinterpret_cast<unsigned*>(s->fetch_host + s->m12_span + s->inl_span);
            a.fetch_value = ++s->fetch_epoch;
        }
        const bool flagged = s->m12p_alt != nullptr && !s->graph_mode && stvo::pose_start_flag_ok(a);  // (batches on the batch kernel)
        const bool gated = piped && pipe_sw != 2 && flagged;
        if (flagged) {
            a.start_flag = s->d_pose_flag;
            a.start_value = ++s->pose_epoch;
            s->pose_flag_frame = s->frame_idx;
        }
        mark(8, sp);
        TRY(stvo::launch_pose(sp, a));
        mark(9, sp);
        if (piped) {
            const int par2 = s->frame_idx & 1;
            HIP_TRY(ctx, hipEventRecord(s->ev_pose[par2], sp));
            s->pose_pending[par2] = true;
            // the point stream's next kernels (the next step's matcher) behind the start of this pose kernel
            if (gated) stvo::launch_stream_gate(st, s->d_pose_flag, s->pose_epoch);
        }
        s->piped_last = piped;
        if (s->pev[0]) (void)hipEventRecord(s->pev[4], st);
        if (s->fetch && !inl_zero_copy) stvo::launch_copy16(st, s->inlp, s->fetch_host + s->m12_span, s->inl_span);
    } else {
        if (par) {  // first frame: nothing to track, but the main stream must still see the line stage's results
            HIP_TRY(ctx, hipEventRecord(s->ev_join, sl));
            HIP_TRY(ctx, hipStreamWaitEvent(st, s->ev_join, 0));
        }
        s->fetch_by_pose = false;
        if (s->fetch) {
            stvo::launch_copy16(st, s->m12s_p, s->fetch_host, s->m12_span);
            HIP_TRY(ctx, hipEventRecord(s->ev_fetch, st));
        }
    }
    return check_launch(ctx);
}

}  // namespace

// Runs the whole per-frame pipeline on the features resident in `slot` (asynchronous; no host transfer).  Optionally
// (STVO_SEQ_GRAPH=1) the chain of ~25 short kernels on two streams is captured ONCE per (slot, buffer parity, line-stage
// flags) into a hipGraph and replayed with a single launch; measured slower than direct launches, see stvo_seq_create_multi.
int stvo_seq_step_dev(stvo_seq* s, int slot) {
    if (!s || slot < 0 || slot >= (int)s->raw_dev.size()) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    StepFlags fl;
    fl.lines_now = s->op.has_lines && s->raw_lines[slot];
    fl.lines_prev = s->op.has_lines && s->set_lines[s->prev_set()];
    fl.track = s->frame_idx > 0;
    {   // the grid buffers of this step (stvo_seq::cells_buf), and the line stream behind every upload the point stream holds
        const stvo_seq::CellsBuf& cb = s->cells_buf[s->frame_idx & 1];
        s->d.pstart = cb.pstart; s->d.pperm = cb.pperm; s->d.pcell = cb.pcell; s->d.plperm = cb.plperm; s->d.plstart = cb.plstart;
        if (s->line_stream && s->upload_seen != s->upload_seq) {
            HIP_TRY(ctx, hipStreamWaitEvent(s->line_stream, s->ev_upload, 0));
            s->upload_seen = s->upload_seq;
        }
    }
    bool forked = false;
    // the first steps run directly (lazy one-time initialisations must not happen inside a capture); timing / fetch /
    // profiling modes record events the host waits on, which a captured graph cannot provide
    const bool use_graph = s->graph_mode && s->frame_idx >= 2 && !s->timing && !s->fetch && !s->pev[0] && !ctx->overlap;
    if (use_graph) {
        const unsigned key = (unsigned)slot | ((unsigned)s->cur << 8) | ((unsigned)fl.lines_now << 10) | ((unsigned)fl.lines_prev << 11) |
                             ((unsigned)fl.track << 12) | ((unsigned)(s->frame_idx & 1) << 13);  // (set index 0..2, grid-buffer parity)
        auto it = s->graphs.find(key);
        if (it == s->graphs.end()) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            bool ok = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            int rc = STVO_ERR_HIP;
            if (ok) {
                rc = seq_enqueue_step(s, slot, fl, &forked);
                ok = hipStreamEndCapture(ctx->stream, &graph) == hipSuccess && rc == STVO_OK && graph != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (!ok) {  // capture unsupported for some node: fall back to direct launches for good
                (void)hipGetLastError();
                s->graph_mode = false;
                TRY(seq_enqueue_step(s, slot, fl, &forked));
            } else {
                it = s->graphs.emplace(key, exec).first;
            }
        }
        if (s->graph_mode) HIP_TRY(ctx, hipGraphLaunch(it->second, ctx->stream));
    } else {
        TRY(seq_enqueue_step(s, slot, fl, &forked));
    }
    if (forked && !s->graph_mode) s->sl_forked_frame = s->frame_idx;
    s->st_dirty = true;
    s->set_lines[s->cur] = fl.lines_now;
    s->last_lines = fl.lines_now;
    s->last_slot = slot;
    s->cur = (s->cur + 1) % 3;  // updateFrame: curr becomes prev
    s->frame_idx++;
    return STVO_OK;
}

// Results of the LAST step (synchronises).  counts (optional, [B][4]): stereo points, stereo lines, matched
// points, matched lines of that frame; results are zeroed after the first frame (nothing to track against yet).
int stvo_seq_read(stvo_seq* s, stvo_pose_result* results, int32_t* counts) {
    if (!s) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int B = s->B;
    hipStream_t st = ctx->stream;
    const stvo_seq::Set& ls = s->set[s->prev_set()];  // the set built by the last step (cur has moved on)
    TRY(seq_wait_pose_stream(s));  // pipelined steps: the last pose kernels run on the aux stream
    if (s->d_prof && s->frame_idx > 1) {  // STVO_POSE_PROF: mean phase ticks of the last pose launch (as stvo_time_stage_dev prints them)
        HIP_TRY(ctx, hipStreamSynchronize(st));
        std::vector<long long> h((size_t)B * 16);
        HIP_TRY(ctx, hipMemcpy(h.data(), s->d_prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double m[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int f = 0; f < B; ++f)
            for (int i = 0; i < 16; ++i) m[i] += (double)h[(size_t)f * 16 + i] / B;
        std::fprintf(stderr, "[pose prof] mean ticks/pair: evaluate %.0f  iter-algebra %.0f  cov+isgood+commit %.0f  remove_outliers %.0f  total %.0f | "
                             "eval-compute %.0f  fold %.0f  barrier+sum %.0f | prologue %.0f | wave busy %.0f %.0f | removeOutliers: residuals %.0f  statistics %.0f  re-deal %.0f\n",
                     m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[14], m[8], m[9], m[10], m[11], m[12]);
    }
    const bool track = s->frame_idx > 1;
    char* OH = s->out_host;
    const size_t res_bytes = (size_t)B * sizeof(stvo_pose_result);
    if (!s->zero_copy) {  // small batches: the kernels have written results and counts straight into OH
        if (track) HIP_TRY(ctx, hipMemcpyAsync(OH, s->results, res_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(OH + res_bytes, ls.n, (size_t)B * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(OH + res_bytes + (size_t)B * 4, ls.nl, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    s->uploads_done = s->uploads;  // every staging copy enqueued so far has read its block
    s->st_dirty = false;           // (the line stream's work of every step was joined into this stream before its pose kernel)
    const stvo_pose_result* hr = reinterpret_cast<const stvo_pose_result*>(OH);
    const int32_t* hn = reinterpret_cast<const int32_t*>(OH + res_bytes);
    for (int b = 0; b < B; ++b) {
        if (results) {
            if (track) {
                results[b] = hr[b];
                if (results[b].path & stvo::PATH_EIG_PENDING) {  // PoseArgs::lazy_eig: SelfAdjointEigenSolver(DT_cov).eigenvalues(), :379-380
                    pm::eig6_ql(results[b].cov, results[b].cov_eig);
                    results[b].path &= ~stvo::PATH_EIG_PENDING;
                }
            }
            else
                std::memset(&results[b], 0, sizeof(stvo_pose_result));
        }
        if (counts) {
            counts[4 * b + 0] = s->op.has_points ? hn[b] : 0;
            counts[4 * b + 1] = s->last_lines ? hn[B + b] : 0;
            counts[4 * b + 2] = track ? hr[b].n_matched_pt : 0;
            counts[4 * b + 3] = track ? hr[b].n_matched_ls : 0;
        }
    }
    return STVO_OK;
}

int stvo_seq_strides(const stvo_seq* s, int32_t* stride_pts, int32_t* stride_lines) {
    if (!s) return STVO_ERR_INVALID_ARG;
    if (stride_pts) *stride_pts = s->K;
    if (stride_lines) *stride_lines = s->M;
    return STVO_OK;
}

int stvo_seq_enable_fetch(stvo_seq* s, int enable) {
    if (!s) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (enable && !s->fetch_host) {
        HIP_TRY(ctx, hipHostMalloc((void**)&s->fetch_host, s->m12_span + s->inl_span + 64, hipHostMallocDefault));
        *reinterpret_cast<volatile unsigned*>(s->fetch_host + s->m12_span + s->inl_span) = 0u;  // the pose kernel's flag (fetch_by_pose)
        HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_fetch, hipEventDisableTiming));
    }
    s->fetch = enable != 0;
    return STVO_OK;
}

int stvo_seq_fetch_matches(stvo_seq* s, const int32_t** m12_stereo_pts, const int32_t** m12_stereo_lines,
                           const int32_t** m12_pts, const int32_t** m12_lines) {
    if (!s || !s->fetch || s->frame_idx == 0) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (s->fetch_by_pose) {
        // the pose kernel of the last step copies the indices and then publishes its number; it may still run.  Bounded poll of the
        // pinned word, then the stream (a launch that failed never publishes)
        const volatile unsigned* flag = reinterpret_cast<const volatile unsigned*>(s->fetch_host + s->m12_span + s->inl_span);
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (unsigned spin = 0; !(seen = (*flag == s->fetch_epoch)); ++spin)
            if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!seen) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (*flag != s->fetch_epoch) {  // the kernel ended without publishing: it gave up waiting for the key-line stream (STVO_POSE_INTERNAL)
                std::snprintf(ctx->last_error, sizeof(ctx->last_error), "%s", "the pose kernel did not publish the match indices of the last step (no signal from the key-line stream)");
                return STVO_ERR_HIP;
            }
        }
    } else {
        HIP_TRY(ctx, hipEventSynchronize(s->ev_fetch));  // the f2f stage of the last step; its pose kernel may still run
    }
    const char* H = s->fetch_host;
    const char* D0 = reinterpret_cast<const char*>(s->m12s_p);
    if (m12_stereo_pts) *m12_stereo_pts = reinterpret_cast<const int32_t*>(H);
    if (m12_stereo_lines) *m12_stereo_lines = reinterpret_cast<const int32_t*>(H + (reinterpret_cast<const char*>(s->m12s_l) - D0));
    if (m12_pts) *m12_pts = reinterpret_cast<const int32_t*>(H + (reinterpret_cast<const char*>(s->m12p) - D0));
    if (m12_lines) *m12_lines = reinterpret_cast<const int32_t*>(H + (reinterpret_cast<const char*>(s->m12l) - D0));
    return STVO_OK;
}

int stvo_seq_fetch_inliers(stvo_seq* s, const int32_t** inl_pts, const int32_t** inl_lines) {
    if (!s || !s->fetch || s->frame_idx < 2) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const char* H = s->fetch_host + s->m12_span;
    if (inl_pts) *inl_pts = reinterpret_cast<const int32_t*>(H);
    if (inl_lines) *inl_lines = reinterpret_cast<const int32_t*>(H + (reinterpret_cast<const char*>(s->inll) - reinterpret_cast<const char*>(s->inlp)));
    return STVO_OK;
}

// Config::useMotionModel() (src/stereoFrameHandler.cpp:317-324) for every sequence of the pipeline: the rule is applied ON THE DEVICE
// by the commit of the previous step (pose_block.h: t0_commit), the first tracked frame starts from prev_frame->DT = I (:45).
int stvo_seq_set_motion_model(stvo_seq* s, int enable) {
    if (!s) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // (every stream a pose kernel or a captured step may still hold init_T / next_T on — the set stvo_seq_destroy waits for)
    if (ctx->aux_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->aux_stream));
    if (s->line_stream) HIP_TRY(ctx, hipStreamSynchronize(s->line_stream));
    if (!enable) {
        if (s->d_motion_T) (void)hipFree(s->d_motion_T);
        s->d_motion_T = nullptr;
    } else {
        if (!s->d_motion_T) HIP_TRY(ctx, hipMalloc((void**)&s->d_motion_T, (size_t)s->B * 16 * sizeof(double)));
        std::vector<double> I((size_t)s->B * 16, 0.0);
        for (int b = 0; b < s->B; ++b)
            for (int i = 0; i < 4; ++i) I[(size_t)b * 16 + i * 5] = 1.0;
        if (!upload_now(ctx, s->d_motion_T, I.data(), I.size() * sizeof(double), "hipMemcpy motion")) return STVO_ERR_HIP;
    }
    for (auto& g : s->graphs) (void)hipGraphExecDestroy(g.second);  // captured steps hold the old init_T pointer
    s->graphs.clear();
    return STVO_OK;
}

int stvo_seq_set_stage_timing(stvo_seq* s, int enable) {
    if (!s) return STVO_ERR_INVALID_ARG;
    s->timing = enable == 2 ? 2 : (enable != 0);
    s->tev_used = 0;
    return STVO_OK;
}

int stvo_seq_get_stage_timing(stvo_seq* s, float avg_ms[STVO_SEQ_NSTAGE], int32_t* n_steps) {
    if (!s || !avg_ms || !n_steps) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    TRY(seq_wait_pose_stream(s));  // (pipelined steps: the pose kernel's pair is on the aux stream)
    double acc[STVO_SEQ_NSTAGE] = {0};
    int cnt[STVO_SEQ_NSTAGE] = {0};
    for (size_t k = 0; k + 2 * STVO_SEQ_NSTAGE <= s->tev_used; k += 2 * STVO_SEQ_NSTAGE)
        for (int st = 0; st < STVO_SEQ_NSTAGE; ++st) {
            if (s->timing == 2 && (st == 0 || st == 3)) continue;  // light: these pairs were not recorded (their events may hold older stamps)
            float ms = 0.f;  // a stage that did not run in this step (first frame: no f2f, no pose) left its events unrecorded
            if (hipEventElapsedTime(&ms, s->tev[k + 2 * st], s->tev[k + 2 * st + 1]) == hipSuccess) {
                acc[st] += ms;
                ++cnt[st];
            }
        }
    (void)hipGetLastError();  // unrecorded events report an error that is not one
    int n = 0;
    for (int st = 0; st < STVO_SEQ_NSTAGE; ++st) {
        avg_ms[st] = cnt[st] ? (float)(acc[st] / cnt[st]) : 0.f;
        if (cnt[st] > n) n = cnt[st];
    }
    *n_steps = n;
    s->tev_used = 0;
    return STVO_OK;
}

// TEST HOOK.  The grid structures the LAST step built on the device for sequence b: the CSR bucket grid of the right
// features (point_cells_kernel / line_cells_kernel: GridStructure + LineIterator of the reference), the integer cells of
// the left features, and — decoded from grid_cover's bit matrix — the candidate set GridStructure::get returned for every
// left feature.  tests/test_gpu_grid_ref.py compares them cell for cell with outputs of the reference's own
// gridStructure.cpp / lineIterator.cpp (tests/golden/ref_device_grid_goldens.npz).
int stvo_seq_debug_grid(stvo_seq* s, int b, int lines, int32_t* cell_start, int32_t* cell_items, int32_t cap_items,
                        int32_t* cells_left, int32_t* cand_off, int32_t* cand, int32_t cap_cand, int32_t* n_left_out) {
    if (!s || b < 0 || b >= s->B || !cell_start || !cell_items || !cells_left || !cand_off || !cand || !n_left_out || s->frame_idx == 0)
        return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(s->line_stream));
    const stvo::SeqDev& d = s->d;
    if (!lines && s->last_point_grid.lean_cells) {  // the step ran the lean cells kernel: produce the full set of grid arrays for the hook
        stvo::SeqDev full = d;
        bind_raw(s, full, s->last_slot);
        hipLaunchKernelGGL(stvo::point_cells_kernel<false>, dim3(s->B), dim3(256), 0, ctx->stream, full);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (lines && s->last_line_fused) {  // the step ran the fused line kernel: the general matcher's kernels produce the grid arrays
        stvo::SeqDev full = d;
        bind_raw(s, full, s->last_slot);
        hipLaunchKernelGGL(stvo::line_cells_kernel, dim3(s->B), dim3(256), 0, ctx->stream, full);
        stvo::launch_grid_batch(ctx->stream, s->last_line_grid, true);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    const int R = lines ? s->M : s->K, xyw = lines ? 4 : 2;
    const size_t items_stride = lines ? (size_t)s->M * stvo::LENT : (size_t)s->K;
    // counts of the slot the last step ran on are not tracked per slot: read them through the last bound raw block
    const char* Rw = s->raw_dev[s->last_slot];
    int32_t nl = 0, nr = 0;
    HIP_TRY(ctx, hipMemcpy(&nl, Rw + (lines ? s->off_nll : s->off_nkl) + (size_t)b * 4, 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(&nr, Rw + (lines ? s->off_nlr : s->off_nkr) + (size_t)b * 4, 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(cell_start, (lines ? d.lstart : d.pstart) + (size_t)b * (STVO_GRID_CELLS + 1),
                           (STVO_GRID_CELLS + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int n_items = cell_start[STVO_GRID_CELLS];
    if (n_items < 0 || (size_t)n_items > items_stride) return STVO_ERR_HIP;
    if (n_items > cap_items) return STVO_ERR_CAPACITY;
    if (n_items) HIP_TRY(ctx, hipMemcpy(cell_items, (lines ? d.litems : d.pitems) + (size_t)b * items_stride, (size_t)n_items * 4, hipMemcpyDeviceToHost));
    if (nl) HIP_TRY(ctx, hipMemcpy(cells_left, (lines ? d.lxy_l : d.pxy_l) + (size_t)b * R * xyw, (size_t)nl * xyw * 4, hipMemcpyDeviceToHost));
    // candidate sets: bit p % 64 of cover[p / 64][i1] <=> right feature perm[p] is a candidate of left feature i1
    if (!lines && s->last_point_grid.range_points) {  // the point scan derives its masks from ranges: materialise exactly those
        stvo::launch_grid_range_debug(ctx->stream, s->last_point_grid);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    const int words64 = R / 64;
    std::vector<unsigned long long> cover((size_t)words64 * R);
    std::vector<int32_t> perm((size_t)R);
    HIP_TRY(ctx, hipMemcpy(cover.data(), (lines ? s->cover_l : s->cover) + (size_t)b * words64 * R, cover.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(perm.data(), (lines ? d.lperm : d.pperm) + (size_t)b * R, perm.size() * 4, hipMemcpyDeviceToHost));
    int tot = 0;
    std::vector<int32_t> row;
    for (int i1 = 0; i1 < nl; ++i1) {
        row.clear();
        for (int w = 0; w < words64; ++w) {
            unsigned long long m = cover[(size_t)w * R + i1];
            while (m) {
                const int p = w * 64 + __builtin_ctzll(m);
                m &= m - 1ull;
                if (p < nr) row.push_back(perm[p]);
            }
        }
        std::sort(row.begin(), row.end());
        cand_off[i1] = tot;
        for (int id : row) {
            if (tot < cap_cand) cand[tot] = id;
            ++tot;
        }
    }
    cand_off[nl] = tot;
    *n_left_out = nl;
    return tot > cap_cand ? STVO_ERR_CAPACITY : STVO_OK;
}

int stvo_seq_push(stvo_seq* s, const stvo_frame_features* f, stvo_pose_result* results, int32_t* counts) {
    if (!s || !f) return STVO_ERR_INVALID_ARG;
    const int slot = s->frame_idx & 1;  // (slots beyond the first two belong to callers of upload / step_dev)
    const bool prof = stvo::dbg().seq_prof != stvo::DBG_UNSET;  // developer aid: host-side phase times
    if (!prof) {
        TRY(stvo_seq_upload(s, slot, f));
        TRY(stvo_seq_step_dev(s, slot));
        return stvo_seq_read(s, results, counts);
    }
    using clk = std::chrono::steady_clock;
    static double acc[4] = {0, 0, 0, 0}, gacc[4] = {0, 0, 0, 0};
    static int n = 0, gn = 0;
    if (!s->pev[0])
        for (auto& e : s->pev) (void)hipEventCreate(&e);
    const auto t0 = clk::now();
    (void)hipEventRecord(s->pev[0], s->ctx->stream);
    TRY(stvo_seq_upload(s, slot, f));
    const auto t1 = clk::now();
    TRY(stvo_seq_step_dev(s, slot));
    const auto t2 = clk::now();
    (void)hipStreamSynchronize(s->ctx->stream);
    const auto t3 = clk::now();
    const int rc = stvo_seq_read(s, results, counts);
    const auto t4 = clk::now();
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    if (s->frame_idx > 10) {  // skip the warm-up frames
        acc[0] += us(t0, t1); acc[1] += us(t1, t2); acc[2] += us(t2, t3); acc[3] += us(t3, t4);
        ++n;
        float ms[4] = {0, 0, 0, 0};
        bool okev = true;
        for (int k = 0; k < 4; ++k) okev = okev && hipEventElapsedTime(&ms[k], s->pev[k], s->pev[k + 1]) == hipSuccess;
        if (okev) {
            for (int k = 0; k < 4; ++k) gacc[k] += ms[k] * 1e3;
            ++gn;
        }
        if (n % 20 == 0)
            std::fprintf(stderr, "[seq prof] host, mean us over %d frames: pack + ingest enqueue %.1f | kernel launches %.1f | wait for GPU %.1f | "
                                 "read-back %.1f || GPU (events): ingest %.1f | stereo stage %.1f | f2f stage %.1f | pose %.1f\n",
                         n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, gacc[0] / (gn ? gn : 1), gacc[1] / (gn ? gn : 1),
                         gacc[2] / (gn ? gn : 1), gacc[3] / (gn ? gn : 1));
    }
    return rc;
}

}  // extern "C"
