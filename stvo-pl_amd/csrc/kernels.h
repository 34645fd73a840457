// kernels.h — internal launch interface between the C-ABI translation unit and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "../../include/stvo_hip.h"
#include "debug_switches.h"
#include "point_tail.h"

namespace stvo {

// More than 64 KB of dynamic LDS per workgroup needs an explicit opt-in, and hipFuncSetAttribute applies to the CURRENT device
// only: the largest size granted so far is remembered per (kernel, device) — a later, larger request raises the attribute again
// (a kernel whose dynamic LDS depends on the problem, e.g. the fused line matcher, must not run on a stale smaller opt-in).
inline bool lds_opt_in(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> granted;  // bytes granted, -1: the runtime refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(kernel, dev);
    const auto it = granted.find(key);
    if (it != granted.end() && (it->second >= bytes || it->second < 0)) return it->second >= bytes;
    const bool ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    if (ok) granted[key] = bytes;
    else if (it == granted.end()) granted[key] = -1;
    return ok;
}

constexpr int KNN_MIN_NSEG = 2, KNN_MAX_NSEG = 16;  // train-range segments per query tile in K1
// Segments per query tile: 1 for very big batches, 2 for big ones (short dispatch rounds), up to 16 for a single frame pair so
// that one 2000 x 2000 problem still spreads over a few hundred workgroups (single-stream latency).
// `capacity` = elements of the context's knn scratch arrays: nseg * B * max_n must fit.
inline int knn_pick_nseg(int B, int max_n, size_t capacity) {
    const int tiles = (max_n + 255) / 256;
    int nseg = (2048 + B * tiles - 1) / (B * tiles > 0 ? B * tiles : 1);
    nseg = nseg < KNN_MIN_NSEG ? KNN_MIN_NSEG : (nseg > KNN_MAX_NSEG ? KNN_MAX_NSEG : nseg);
    if ((long long)B * tiles >= 4096) nseg = 1;  // >= 5 dispatch rounds even unsegmented: a workgroup's fixed cost is paid once per query tile
                                                 // (1024 frames: K1m 0.562 -> 0.551 ms, reverse check 0.054 -> 0.050 ms)
    if (dbg().knn_nseg != DBG_UNSET && dbg().knn_nseg > 0) nseg = dbg().knn_nseg;  // developer override
    const size_t per_seg = (size_t)(B > 0 ? B : 1) * (size_t)(max_n > 0 ? max_n : 1);
    while (nseg > 1 && (size_t)nseg * per_seg > capacity) --nseg;
    return nseg;
}

// CUs of the current device (persistent-workgroup launches size their grids with it), cached per device
inline int device_cu_count() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lk(mu);
    const auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
    return n;
}

// ---- K1 / K2: brute-force Hamming 2-NN, ratio test, mutual check ----------------------------
// knn: [nseg][B][row_stride] packed (best_key, second_key) per train segment,
// key = (distance << 16) | train_index; consumers merge the segments.
void launch_hamming_knn2(hipStream_t s, int B, int row_stride, int max_n, const uint8_t* d1, const int32_t* n1,
                         const uint8_t* d2, const int32_t* n2, uint2* knn12, uint2* knn21, int both_directions,
                         int lds_pad_bytes = 0, int dir0 = 0, const int32_t* qsel = nullptr,
                         const int32_t* nsel = nullptr, int nseg = KNN_MIN_NSEG, uint32_t* claim_init = nullptr);
// K1m: the same scan with the distances taken from the matrix cores (match_mfma.hip); qb = query blocks of 32 rows per
// wave (1, 2 or 4; a workgroup covers 128 * qb query rows).  Train indices must fit 13 bits (max_n <= 8192).
// the reverse check of the claimed columns in one launch (match_mfma.hip): light columns against S, heavy columns against all rows,
// the train side of a unit resident in LDS; `slots` workgroups per frame pair share the frame's units; m12[claimant] = -1 for every
// listed column whose claim another row blocks
constexpr int REV_MAX_SEG = 8;
__host__ __device__ inline int rev_segments(int train_rows) {  // train segments of a column list with this many train rows
    const int s = (train_rows + 255) / 256;
    return s < 1 ? 1 : (s > REV_MAX_SEG ? REV_MAX_SEG : s);
}
void launch_hamming_knn2_mfma_reverse(hipStream_t s, int B, int row_stride, const uint8_t* d1, const int32_t* n1, const uint8_t* d2, const int32_t* qsel,
                                      const int32_t* nsel, const int32_t* tsel, const uint32_t* claim, float nnr, int32_t* m12, int slots);
void launch_hamming_knn2_mfma(hipStream_t s, int B, int row_stride, int max_n, const uint8_t* d1, const int32_t* n1,
                              const uint8_t* d2, const int32_t* n2, uint2* knn12, uint2* knn21, int both_directions,
                              int dir0, const int32_t* qsel, const int32_t* nsel, int nseg, uint32_t* claim_init, int qb,
                              int qsel_from_back = 0, const int32_t* tsel = nullptr, const int32_t* ntsel = nullptr);
int knn_mfma_qb(int max_n);  // query blocks per wave of K1m for this problem size, 0 = VALU kernels
// mutual matching with a lazy reverse pass: only the columns claimed by an accepted forward match are examined
// (match_kernels.hip) — on the matrix-core path by a per-frame plan + one launch of sparse reverse scans, on the VALU path
// (> 8192 rows, STVO_KNN_MFMA=0) by a range query with early exit
struct LazyScratch {
    uint2* knn12;
    uint2* knn21;   // VALU path: reused as int32 blocked[B][row_stride].  Matrix-core path: tsel[] (B * row_stride int32) in the last
                    // B * row_stride / 2 entries
    int32_t* cand;  // [B][row_stride] forward ratio-tested best (VALU path)
    int32_t* need;  // [B][row_stride] per-column claim (d0 << 16 | claimant), 0xFFFFFFFF = unclaimed
    int32_t* qsel;  // [B][row_stride] compacted flagged columns
    int32_t* nsel;  // [5][B]: claimed columns; light, heavy, |S|, tau of the matrix-core reverse check (reverse_plan_kernel)
    size_t knn_capacity;  // elements of knn12 / knn21
};
void launch_match_mutual_lazy(hipStream_t s, int B, int row_stride, const uint8_t* d1, const int32_t* n1,
                              const uint8_t* d2, const int32_t* n2, float nnr, const LazyScratch& w, int32_t* m12,
                              int lds_pad_bytes, hipEvent_t wait_before_m12_write, hipEvent_t* timing_events = nullptr);
// the verification pass alone, on the claims left by the last launch_match_mutual_lazy (timing tools)
void launch_hamming_verify(hipStream_t s, int B, int row_stride, const uint8_t* d1, const int32_t* n1, const uint8_t* d2,
                           float nnr, const LazyScratch& w, int lds_pad_bytes, int nseg, int32_t* m12);
// StVO::match of B frame pairs with at most 512 rows per set (key-lines): one workgroup per frame pair, both sets in LDS
bool match_small_ok(int row_stride);
void launch_match_small(hipStream_t s, int B, int row_stride, const uint8_t* d1, const int32_t* n1, const uint8_t* d2,
                        const int32_t* n2, float nnr, int mutual, int32_t* m12, int cap = 0 /* rows per set the LDS is sized for; 0: row_stride */);
void launch_nnr_mutual(hipStream_t s, int B, int row_stride, const uint2* knn12, const uint2* knn21, const int32_t* n1,
                       const int32_t* n2, float nnr, int mutual, int32_t* m12, int nseg = KNN_MIN_NSEG);
void launch_valu_probe(hipStream_t s, int blocks, int iters, uint32_t* sink);
// 16-byte-granular copy kernel; either side may be pinned host memory.  For the few hundred KB a single-stream call
// moves it has ~5 us less latency than the DMA engine behind hipMemcpyAsync (6 us vs 11 us on top of an empty launch,
// and ~0 vs 3 us for a result block written straight into pinned memory).
void launch_copy16(hipStream_t s, const void* src, void* dst, size_t bytes);
extern const double kValuProbeOpsPerThreadIter;

// ---- K4-K6: pose optimisation, one workgroup per frame pair ------------------------------------
struct PoseArgs {
    int B, max_pts, max_lines;
    const int32_t* n_prev_pts;
    const double* prev_P;
    const double* prev_s2p;
    const double* curr_pl;
    const int32_t* m12p;        // nullptr: identity association (records mode)
    const int32_t* init_inl_p;  // nullptr: every matched record starts as inlier
    const int32_t* n_prev_lines;
    const double* prev_sP;
    const double* prev_eP;
    const double* prev_spl;
    const double* prev_epl;
    const double* prev_s2l;
    const double* curr_le;
    const int32_t* m12l;
    const int32_t* init_inl_l;
    const double* init_T;  // [B][16] or nullptr
    double* next_T;        // [B][16] or nullptr: the next frame pair's init_T under use_motion_model (pose_block.h: t0_commit); may alias init_T
    stvo_cam cam;
    const stvo_cam* cams;  // [B] per-frame-pair calibration (device) or nullptr: `cam` for every pair
    // set by launch_pose: prev points / lines with an index below these live in the workgroup's LDS record cache
    int lds_cap_pts, lds_cap_lines;
    stvo_opt_params prm;
    stvo_pose_result* results;
    int32_t* inl_p_out;  // [B][max_pts] or nullptr
    int32_t* inl_l_out;
    int eval_only;       // 1: a single optimizeFunctions evaluation at init_T
    int eval_robust;
    double* eval_out;    // [B][44]: H(36) g(6) e n
    long long* prof_out; // optional [B][16] phase ticks (tools only)
    // compact records of the device-resident pipeline (seq_pipeline.hip), or nullptr: a stereo point is {u, v, disparity as
    // floats, pyramid level} — exactly what the reference's PointFeature is built from (src/stereoFrame.cpp:152-167: float
    // key-point coordinates, a float difference, backProjection of the three) — so P, sigma2 and the observation are recomputed
    // on chip instead of being stored and re-read as seven doubles.  prev_rc[i] gives P and sigma2 of prev point i, curr_rc[j]
    // the observation; prev_P / prev_s2p / curr_pl are ignored then.
    const float4* prev_rc;  // [B][max_pts]
    const float4* curr_rc;  // [B][max_pts]
    const double* q_tab;    // [STVO_POSE_QTAB] sqrt(sigma2) of pyramid level l (device memory; levels beyond the table are computed)
    double level_scale;     // orbScaleFactor: sigma2 = 1 / scale^(2 level) (src/stereoFeatures.cpp:41-47)
    // In-kernel ordering for single-stream operation (pose_inline_sync_ok; the latency kernel only), instead of events between
    // streams — on this runtime every recorded / awaited event delays the next kernel of its stream by ~6 us:
    //   wait_flag   the kernel starts by waiting until *wait_flag has reached wait_value (device memory, set by
    //               launch_stream_signal behind the last kernel of another stream whose results this launch reads);
    //   fetch_*     the solver wave of workgroup 0 copies fetch_n16 16-byte words (by-products the host wants while this kernel
    //               runs) to pinned host memory and then publishes fetch_value in *fetch_flag (pinned, system scope).
    const unsigned* wait_flag;
    unsigned wait_value;
    // lazy_eig != 0 (latency kernel only; the caller reads the results through stvo_seq_read): cov_eig is left to the reader, flagged in
    // stvo_pose_result::path (pose_block.h: PATH_EIG_PENDING)
    int lazy_eig;
    // lines_on_solver != 0 (set by launch_pose for the latency kernel; STVO_POSE_LOS=0 clears it): frame pairs with at most 64 LPT
    // key-lines have them evaluated by the solver wave (pose_kernel.hip)
    int lines_on_solver;
    const uint4* fetch_src;
    uint4* fetch_dst;
    unsigned fetch_n16;
    unsigned* fetch_flag;
    unsigned fetch_value;
    // start_flag (batch kernel pose2c only; nullptr otherwise): the FIRST workgroup of the launch publishes start_value there when it
    // starts — the launch is taking its CUs (1024 workgroups are placed within a few microseconds).  seq_pipeline.hip holds the next
    // step's key-line kernels behind it (launch_stream_gate), so that their workgroups do not take the CUs first.  Not the LAST
    // workgroup: 1024 frame pairs fill every VGPR of the chip (2 waves x 256 registers per SIMD), so the SIMD that hosts the waiting
    // gate wave has room for one pose wave only — the last workgroup could not start before the gate left, and the gate waited for the
    // last workgroup: a circular wait that resolved only when the first frame pair of the launch finished, 160 us later (round 6: every
    // second pipeline of a process ran 0.864 instead of 0.793 ms per step that way, depending on where the gate wave had landed).
    unsigned* start_flag;
    unsigned start_value;
};
// the batch kernel honours PoseArgs::start_flag for this launch (same selection as launch_pose)
bool pose_start_flag_ok(const PoseArgs& a);
// one thread that leaves when *flag has reached value, or after ~0.5 ms (a scheduling hint, never a dependence)
void launch_stream_gate(hipStream_t s, const unsigned* flag, unsigned value);
constexpr int STVO_POSE_QTAB = 16;
// internal flag in stvo_pose_result::path (never leaves the library): the eigenvalues of `cov` are still to be computed by the reader
constexpr int PATH_EIG_PENDING = 1 << 30;
// may launch_pose honour wait_flag / fetch_* for a batch of B frame pairs?  (few workgroups: the kernels they wait for always find CUs)
bool pose_inline_sync_ok(int B);
// *flag = value (release, device scope) once everything enqueued on `s` so far has completed
void launch_stream_signal(hipStream_t s, unsigned* flag, unsigned value);
// dispatch: pose_kernel.hip's latency variant up to 256 frame pairs (and for single evaluations), pose_kernel2p.hip beyond
int launch_pose(hipStream_t s, const PoseArgs& a);
int launch_pose2p(hipStream_t s, const PoseArgs& a);  // pose_kernel2p.hip: thread-private records, four frame pairs per CU
// the per-(device, stream) record arena of the batch kernel: every context that holds a stream retains it, the last release frees
// it (the releasing caller has synchronised the stream and made its device current)
void pose2p_retain_stream(hipStream_t s);
void pose2p_release_stream(hipStream_t s);
inline void pose_retain_stream(hipStream_t s) { pose2p_retain_stream(s); }
inline void pose_release_stream(hipStream_t s) { pose2p_release_stream(s); }

// ---- 8-bit image operations of the ORB front-end that the line detector reuses (orb_kernels.hip) ----
// cv::GaussianBlur(7 x 7) in 8-bit fixed point with the integer kernel k7 (weights x 2^8), BORDER_REFLECT_101
void launch_blur7_u8(hipStream_t s, int B, int cols, int rows, const uint8_t* src, uint8_t* dst, const int* k7);
// cv::resize(INTER_LINEAR) for 8-bit images, 11-bit fixed point; fx = fy = 0: resize to a given dsize (sample step ssize / dsize),
// fx, fy > 0: resize(src, dst, Size(), fx, fy) with dsize = cvRound(ssize f) — the sample step stays 1 / f (LSD's scaled image)
void launch_resize_linear_u8(hipStream_t s, int B, int scols, int srows, int dcols, int drows, const uint8_t* src, uint8_t* dst, double fx = 0.0,
                             double fy = 0.0);

// ---- K3: grid-windowed stereo matchers, batched over frame pairs (blockIdx.y) ----------------------
constexpr int GRID_LW = STVO_GRID_COLS + 16, GRID_LCELLS = STVO_GRID_ROWS * GRID_LW, GRID_LSTART_STRIDE = GRID_LCELLS + 4;
// the cells phase of the key-points of a frame (point_cells.h): what it reads and writes
struct PointCells {
    int K, ws;               // rows per frame of every array below; matching_s_ws (stereoFrame.cpp:141-143)
    const float* kp_l;       // [B][K][2]
    const float* kp_r;       // [B][K][2]
    const int32_t* n_kp_l;   // [B]
    const int32_t* n_kp_r;   // [B]
    const double* inv_wh;    // [B][2] 64 / cols, 48 / rows (stereoFrame.cpp:47-48)
    int32_t* pstart;         // [B][3073] out: exclusive cell starts of the right key-points
    uint32_t* plstart;       // [B][GRID_LSTART_STRIDE] out: cell starts of the left key-points (nullptr: not sorted)
    int32_t* plperm;         // [B][K] out: position -> left key-point
    int32_t* pperm;          // [B][K] out: scan position -> right key-point
    int32_t* pcell;          // [B][K] out: cell (y * 64 + x) of the right key-point at each scan position, -1: outside the grid
    // inputs of the scan formulation only (LEAN == false)
    int32_t* pxy_l;              // [B][K][2]
    unsigned long long* top2_p;  // [B][K]
    int32_t* govf_p;             // [B]
    int32_t* prange;             // [B][K][2]
    int32_t* pitems;             // [B][K]
    int32_t* prank;              // [B][K]
};

struct GridBatch {
    int B, stride1, stride2;   // frame pairs; rows per frame of the left / right feature arrays
    int xy_width;              // 2 (points: cx, cy) or 4 (lines: sx, sy, ex, ey)
    int items_stride;          // CSR items per frame
    int words64, n1p;          // ceil(stride2 / 64); stride1 rounded up to 64
    const int32_t* cell_xy1;   // [B][stride1][xy_width]
    const uint8_t* d1;         // [B][stride1][32]
    const int32_t* n1;         // [B]
    const int32_t* cell_start; // [B][3073]
    const int32_t* cell_items; // [B][items_stride]
    const uint8_t* d2;         // [B][stride2][32]
    const int32_t* n2;         // [B]
    const double* dir2;        // [B][stride2][2] (lines) or nullptr
    stvo_grid_window w;
    double ratio, line_sim_th;
    int mutual;
    unsigned long long* cover; // [B][words64][n1p] scratch
    const int32_t* rank;       // [B][stride2] right feature id -> scan position
    const int32_t* perm;       // [B][stride2] scan position -> right feature id
    unsigned long long* top2;  // [B][stride1] scratch
    int32_t* owner2;           // [B][stride2] scratch
    int32_t* m12;              // [B][stride1] out
    // optional scratch of the single-scan formulation (mutual != 0): the ELIGIBLE (left feature, distance) pairs every right
    // feature met in scan pass 1 — nullptr: two full scan passes
    // range_points != 0 (points, window of one grid row, right features numbered in CSR order — the stvo_seq pipeline): the
    // candidates of a left feature are ONE contiguous range of scan positions, cell_start[row][x - w_lo] .. cell_start[row][x +
    // w_hi + 1); the scan derives its masks from cell_start and no candidate bit-matrix is built (cover may be nullptr);
    // top2 / ovf must then be initialised by the caller (kTop2Empty = 0x00000000FFFFFFFF, 0)
    int range_points;
    const int32_t* range1;     // [B][stride1][2] (lo, hi) of every left feature when range_points != 0
    const int32_t* cell2;      // [B][stride2] or nullptr: grid cell (y * 64 + x) of the right feature at each scan position, -1 outside the grid
    // left key-points counting-sorted by cell for the one-workgroup-per-frame point matcher (both or none), on a grid GRID_LW
    // columns wide (a left key-point up to w_lo columns right of the grid still has candidates)
    const uint32_t* lstart;    // [B][GRID_LSTART_STRIDE] exclusive cell starts, [GRID_LCELLS] = number of placed left key-points
    const int32_t* lperm;      // [B][stride1] position -> left feature
    uint32_t* elig;            // [B][GRID_ELIG][stride2]  (i1 << 16 | d), slot-major
    int32_t* elig_cnt;         // [B][stride2]
    int32_t* ovf;              // [B] set when some right feature of the frame met more than GRID_ELIG eligible pairs
    int32_t* misfit;           // [B] or nullptr: written by the one-workgroup-per-frame point matcher (1: frame left to the scan formulation)
    // != 0: the caller left range1 / top2 / ovf uninitialised because grid_points_fused_ok() holds; the one-workgroup matcher
    // derives them from cell_start / lstart / lperm for the frames it hands to the scan formulation
    int lean_cells;
    // has_tail != 0 (only with grid_points_fused_ok): the one-workgroup matcher also runs the tail of the stereo association on the
    // frame it just matched (filters, back-projection, ordered compaction, point_tail.h) — no point_tail_kernel launch, no second
    // pass over the matches
    int has_tail;
    PointTail tail;
    // fused_cells != 0 (only with grid_points_fused_ok and B <= the number of workgroups of the launch, i.e. one frame per workgroup):
    // the one-workgroup matcher builds the grid of its frame itself as its first phase — no point_cells_kernel launch
    int fused_cells;
    PointCells cells;
    // dyn_ctr != nullptr (persistent point matcher, pipelined steps of seq_pipeline.hip): the workgroups take their frames from the counter
    // dyn_ctr[dyn_par] instead of by a static stride — the launch then starts beside the previous step's pose kernel, its workgroups come
    // to their CUs one by one as that kernel's frame pairs finish, and the ones that start early take more frames.  The launch resets
    // dyn_ctr[dyn_par ^ 1] for the next one; dyn_owner [B] remembers who took a frame (for the workgroup's own pass over misfits).
    int32_t* dyn_ctr;
    int32_t* dyn_owner;
    int dyn_par;
};
constexpr int GRID_ELIG = 16;
// scan_events (optional): [0] / [1] are recorded on `s` before / after the two grid_scan passes
void launch_grid_batch(hipStream_t s, const GridBatch& g, bool lines, hipEvent_t* scan_events = nullptr);
// true when launch_grid_batch(.., lines = false) will run the one-workgroup-per-frame point matcher for this batch
bool grid_points_fused_ok(const GridBatch& g);
void launch_grid_range_debug(hipStream_t s, const GridBatch& g);  // test hook, see stvo_seq_debug_grid

}  // namespace stvo
