// grid_kernels.hip — K3: the grid-windowed stereo matchers for gfx950.
//
// Replaces both overloads of StVO::matchGrid (/root/reference/src/matching.cpp:111-177 points,
// :179-258 lines) including GridStructure::get (src/gridStructure.cpp:65-76).
//
// The reference loop is sequential over the left features i1 with a running strict minimum per
// right feature i2 (`distances[i2]`, :145-150): a candidate only takes part in i1's best / second
// if it strictly improves on every EARLIER i1 that had i2 in its window.  Restated without the
// loop-carried dependence:
//     eligible(i1,i2)  <=>  D(i1,i2) < min over covering i1' < i1 of D(i1',i2)
//     owner(i2)        =   lowest i1 attaining the global minimum of D(.,i2)   (= final matches_21)
// which maps onto three kernels:
//   A  `grid_cover`   lane = left feature: walks its window cells in the CSR grid and sets the bit of
//                     every right feature found there in the candidate bit-matrix (the
//                     std::unordered_set de-duplication of gridStructure.cpp:75 is the idempotence of
//                     OR; lines OR both end-point windows).  The matrix is stored TRANSPOSED,
//                     cover[wave of right features][left feature], and right features are numbered in
//                     CSR (= spatial) order, so the masks one scanning wave needs are contiguous and
//                     almost all zero.
//   B  `grid_scan`    lane = right feature, descriptor resident in 8 VGPRs; the left features stream
//                     by in ASCENDING i1 as wave-uniform rows (scalar loads, the next row requested while the
//                     current one is compared) — the same popcount inner loop as K1, predicated by the
//                     candidate bit and (lines) the direction cosine gate (:221-222); 64 masks per coalesced
//                     load, non-zero ones found by ballot; each lane carries its running minimum, so
//                     eligibility and ownership fall out in order.  The scan runs TWICE so that no atomic
//                     ever has to return a value (a returning global atomic costs ~1 us and one wave issues
//                     ~150 of them back to back; that chain was the whole latency of this stage):
//                       pass 1: best_key[i1] = min over eligible pairs of (d << 16 | i2)   (atomic min, no return)
//                       pass 2: blocked[i1] = 1 if another eligible pair fails  best_d < d * ratio  (plain store)
//                     With ratio <= 1 (enforced) "best_d < second_d * ratio" holds iff it holds for every
//                     eligible candidate other than the best one, so the second-best distance itself is never
//                     needed; one eligible candidate => nothing can block (the reference's best_d2 = INT_MAX).
//   C  `grid_finalize` lane = left feature: accepts best unless blocked (DOUBLE ratio test of :160, evaluated
//                     pairwise in pass 2), mutual check against owner (:166-174).
#include <cstdlib>
#include <vector>

#include "ctx_internal.h"
#include "point_cells.h"

namespace stvo {
namespace {

#ifndef GRID_ROWS_PER_TRIP
#define GRID_ROWS_PER_TRIP 1
#endif
constexpr unsigned long long kTop2Empty = 0x00000000FFFFFFFFull;  // no eligible candidate yet, not blocked

struct GridArgs {
    const int32_t* cell_xy1;  // [n1][2] points  |  [n1][4] lines (sx, sy, ex, ey)
    const uint8_t* d1;
    int n1;
    const int32_t* cell_start;
    const int32_t* cell_items;
    const uint8_t* d2;
    int n2;
    const double* dir2;  // lines only
    stvo_grid_window w;
    double ratio, line_sim_th;
    int mutual;
    int words64;                 // number of 64-lane waves of right features = ceil(n2 / 64)
    int n1p;                     // n1 rounded up to the 64-mask scan block
    unsigned long long* cover;   // [words64][n1p]: bit p%64 of cover[p/64][i1] <=> right feature perm[p] is a candidate of i1
    const int32_t* rank;         // [n2] right feature id -> scan position p (spatial = CSR order)
    const int32_t* perm;         // [n2] scan position p -> right feature id
    unsigned long long* top2;    // [n1] low word: best eligible key (d << 16 | i2), high word: blocked flag
    int32_t* owner2;             // [n2]
    int32_t* m12;                // [n1]
    uint32_t* elig;              // [GRID_ELIG][elig_stride] or nullptr
    int32_t* elig_cnt;           // [n2]
    int32_t* ovf;                // [1]
    int elig_stride;
    const int32_t* range1;       // [n1][2] candidate range (lo, hi) in scan positions, range formulation only
};

// per-frame view of a batch (blockIdx.y = frame pair)
__device__ __forceinline__ GridArgs frame_view(const GridBatch& g, int b) {
    GridArgs a;
    a.cell_xy1 = g.cell_xy1 + (size_t)b * g.stride1 * g.xy_width;
    a.d1 = g.d1 + (size_t)b * g.stride1 * STVO_DESC_BYTES;
    a.n1 = g.n1[b];
    a.cell_start = g.cell_start + (size_t)b * (STVO_GRID_CELLS + 1);
    a.cell_items = g.cell_items + (size_t)b * g.items_stride;
    a.d2 = g.d2 + (size_t)b * g.stride2 * STVO_DESC_BYTES;
    a.n2 = g.n2[b];
    a.dir2 = g.dir2 ? g.dir2 + (size_t)b * g.stride2 * 2 : nullptr;
    a.w = g.w;
    a.ratio = g.ratio;
    a.line_sim_th = g.line_sim_th;
    a.mutual = g.mutual;
    a.words64 = g.words64;
    a.n1p = g.n1p;
    a.cover = g.cover + (size_t)b * g.words64 * g.n1p;
    a.rank = g.rank + (size_t)b * g.stride2;
    a.perm = g.perm + (size_t)b * g.stride2;
    a.top2 = g.top2 + (size_t)b * g.stride1;
    a.owner2 = g.owner2 + (size_t)b * g.stride2;
    a.m12 = g.m12 + (size_t)b * g.stride1;
    a.elig = g.elig ? g.elig + (size_t)b * GRID_ELIG * g.stride2 : nullptr;
    a.elig_cnt = g.elig ? g.elig_cnt + (size_t)b * g.stride2 : nullptr;
    a.ovf = g.elig ? g.ovf + b : nullptr;
    a.elig_stride = g.stride2;
    a.range1 = g.range1 ? g.range1 + (size_t)b * g.stride1 * 2 : nullptr;
    return a;
}

// ORs the bit of every right feature found in the window of cell (x, y) into column i1.  `col` / `stride`: where the
// column's words live — an LDS staging column of the calling thread (stride 256) or the global bit-matrix (stride n1p).
template <typename P>
__device__ __forceinline__ void cover_window(const GridArgs& a, int x, int y, P col, size_t stride) {
    // GridStructure::get clamping (src/gridStructure.cpp:67-71)
    const int min_x = max(0, x - a.w.w_lo), max_x = min(STVO_GRID_COLS, x + a.w.w_hi + 1);
    const int min_y = max(0, y - a.w.h_lo), max_y = min(STVO_GRID_ROWS, y + a.w.h_hi + 1);
    for (int yy = min_y; yy < max_y; ++yy) {
        if (min_x >= max_x) break;
        // cells of one grid row are contiguous in the CSR (cell = y*64 + x)
        const int lo = a.cell_start[yy * STVO_GRID_COLS + min_x], hi = a.cell_start[yy * STVO_GRID_COLS + max_x];
        for (int k = lo; k < hi; ++k) {
            const int id = a.cell_items[k];
            if (id >= 0 && id < a.n2) {  // :141 skips out-of-range ids
                const int p = a.rank[id];
                col[(size_t)(p >> 6) * stride] |= 1ull << (p & 63);  // the column is owned by this thread
            }
        }
    }
}

// Builds column i1 of the transposed candidate bit-matrix.  USE_LDS: the column (words64 words) is assembled in an LDS
// staging column of the thread and then written out once with plain stores that are coalesced across the wave — no
// read-modify-write on global memory (each of the ~20 (points) to several hundred (lines) candidate bits of a left
// feature used to cost a dependent L2 round trip) and no separate memset of the matrix.
template <bool LINES, bool USE_LDS>
__global__ __launch_bounds__(256) void grid_cover_kernel(GridBatch g) {
    extern __shared__ unsigned long long s_col[];  // [words64][256] when USE_LDS
    const GridArgs a = frame_view(g, blockIdx.y);
    const int i1 = blockIdx.x * 256 + threadIdx.x;
    if (i1 == 0 && a.ovf) a.ovf[0] = 0;
    if (i1 >= a.n1p) return;
    unsigned long long* gcol = a.cover + i1;
    if (USE_LDS) {
        for (int w = 0; w < a.words64; ++w) s_col[w * 256 + threadIdx.x] = 0ull;
    } else {
        for (int w = 0; w < a.words64; ++w) gcol[(size_t)w * a.n1p] = 0ull;
    }
    if (i1 < a.n1) {
        if (LINES) {
            const int4 c = reinterpret_cast<const int4*>(a.cell_xy1)[i1];
            if (USE_LDS) {
                cover_window(a, c.x, c.y, s_col + threadIdx.x, (size_t)256);
                cover_window(a, c.z, c.w, s_col + threadIdx.x, (size_t)256);
            } else {
                cover_window(a, c.x, c.y, gcol, (size_t)a.n1p);
                cover_window(a, c.z, c.w, gcol, (size_t)a.n1p);
            }
        } else {
            const int2 c = reinterpret_cast<const int2*>(a.cell_xy1)[i1];
            if (USE_LDS)
                cover_window(a, c.x, c.y, s_col + threadIdx.x, (size_t)256);
            else
                cover_window(a, c.x, c.y, gcol, (size_t)a.n1p);
        }
        a.top2[i1] = kTop2Empty;
    }
    if (USE_LDS)
        for (int w = 0; w < a.words64; ++w) gcol[(size_t)w * a.n1p] = s_col[w * 256 + threadIdx.x];  // padding columns: zeros
}

__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// GridStructure::get for a one-row window (src/gridStructure.cpp:65-76) when the right features are numbered in CSR order:
// the cells x - w_lo .. x + w_hi of row y are contiguous in the CSR, so the candidates of left feature i1 are the scan
// positions [lo, hi) and the share of the wave that owns positions wbase .. wbase + 63 is one run of mask bits.
// (lo, hi) come from the kernel that built the CSR (point_cells_kernel, seq_pipeline.hip): one coalesced 8-byte load per lane.
__device__ __forceinline__ unsigned long long range_mask(const GridArgs& a, int i1, int wbase) {
    const int2 r = reinterpret_cast<const int2*>(a.range1)[i1 < a.n1 ? i1 : 0];
    const int b0 = max(r.x, wbase) - wbase, b1 = min(r.y, wbase + 64) - wbase;
    const bool any = i1 < a.n1 && b1 > b0;
    const unsigned long long run = (b1 - b0 >= 64) ? ~0ull : (((1ull << (b1 - b0)) - 1ull) << b0);
    return any ? run : 0ull;
}

// TEST HOOK (stvo_seq_debug_grid): materialises the masks the range formulation feeds the scan into the bit-matrix layout
__global__ __launch_bounds__(256) void grid_range_debug_kernel(GridBatch g) {
    const GridArgs a = frame_view(g, blockIdx.y);
    const int i1 = blockIdx.x * 256 + threadIdx.x;
    if (i1 >= a.n1p) return;
    for (int w = 0; w < a.words64; ++w) a.cover[(size_t)w * a.n1p + i1] = range_mask(a, i1, w * 64);
}

template <bool LINES, int PASS, bool RANGE>
__device__ __forceinline__ void grid_scan_body(const GridArgs& a, const int p) {
    // lane = scan position p; positions follow the CSR (cell) order of the right features, so the 64
    // features of a wave are spatial neighbours and only the few left features whose window touches
    // that neighbourhood have a non-zero mask: the scan skips 64 left features per vector load.
    const int lane = threadIdx.x & 63;
    if ((p & ~63) >= a.n2) return;  // whole wave past the last right feature (wave-uniform)
    // single-scan formulation: pass 1 records the eligible pairs, grid_elig_kernel judges them; this full second scan only
    // runs for a frame in which some right feature met more than GRID_ELIG of them
    if (PASS == 2 && a.elig && a.ovf[0] == 0) return;
    int n_elig = 0;
    const bool live = p < a.n2;
    const int i2 = a.perm[live ? p : a.n2 - 1];
    const uint4 t0 = reinterpret_cast<const uint4*>(a.d2)[2 * i2];
    const uint4 t1 = reinterpret_cast<const uint4*>(a.d2)[2 * i2 + 1];
    double dirx = 0.0, diry = 0.0;
    if (LINES) {
        dirx = a.dir2[2 * i2];
        diry = a.dir2[2 * i2 + 1];
    }
    int run_min = 0x7FFFFFFF, owner = -1;
    const int widx = __builtin_amdgcn_readfirstlane(p >> 6);
    const unsigned long long* __restrict__ col = a.cover + (size_t)widx * a.n1p;  // wave-uniform row of masks
    uint32_t* __restrict__ best_key = reinterpret_cast<uint32_t*>(a.top2);  // [2 * i1] key, [2 * i1 + 1] blocked
    // 64 masks per coalesced vector load; the ballot of the non-zero ones is walked with scalar bit tricks, so an
    // (almost always) all-zero block of 64 left features costs one load + one ballot.
    // Software pipeline, two blocks deep: lane u of the wave fetches the mask of left feature base + u, and — one
    // block later, once the mask is known to be non-zero — that feature's descriptor row (and, per variant, its
    // cell coordinates / best key) with ordinary vector loads.  The row loop then broadcasts row u out of lane u's
    // registers with v_readlane: no memory access, hence no memory latency, inside the per-row loop.  (Fetching
    // each row with wave-uniform scalar loads cost one scalar-cache miss, ~0.3 us, per row even with the next row
    // prefetched.)
    const uint4* __restrict__ Q4 = reinterpret_cast<const uint4*>(a.d1);
    const int4* __restrict__ XY4 = reinterpret_cast<const int4*>(a.cell_xy1);
    (void)XY4;
    // branch-free loads (clamped index, value discarded) so that the compiler can count the loads in flight and
    // wait for exactly the ones it needs instead of draining the queue in every block
    const int last_blk = a.n1p - 64;
    const int wbase = widx * 64;
    auto mask_at = [&](int blk) -> unsigned long long {
        if (RANGE) return range_mask(a, blk + lane, wbase);
        const unsigned long long m = col[(blk < a.n1p ? blk : last_blk) + lane];
        return blk < a.n1p ? m : 0ull;
    };
    unsigned long long m_next = mask_at(0), m_next2 = mask_at(64);
    uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
    double rvx = 0.0, rvy = 0.0;  // lines: unit direction of left line base + lane, computed once per line by its lane
    uint32_t rk = 0u;
    auto rows_at = [&](int blk, unsigned long long m) {
        // lanes whose left feature is nobody's candidate in this wave read row 0 instead (one shared cache line)
        const size_t i1 = (m != 0ull) ? (size_t)(blk + lane) : (size_t)0;
        ra = Q4[2 * i1];
        rb = Q4[2 * i1 + 1];
        if (LINES) {
            // direction of the LEFT line from INTEGER cell differences; 0/0 = NaN never skips (:207-222)
            const int4 c = XY4[i1];
            const double vx = (double)(c.z - c.x), vy = (double)(c.w - c.y);
            const double mag = sqrt(vx * vx + vy * vy);
            rvx = vx / mag;
            rvy = vy / mag;
        }
        if (PASS == 2) rk = best_key[2 * i1];
    };
    rows_at(0, m_next);
    for (int base = 0; base < a.n1p; base += 64) {
        const unsigned long long cur = m_next;
        const uint4 qa = ra, qb = rb;
        const double qvx = rvx, qvy = rvy;
        const uint32_t qk = rk;
        m_next = m_next2;
        m_next2 = mask_at(base + 128);
        rows_at(base + 64, m_next);
        unsigned long long nz = __ballot(cur != 0ull);
        // R rows per trip: their masks / descriptors are broadcast and their distances computed as independent
        // chains (with one wave per SIMD in single-stream operation every dependent instruction pays its full
        // latency), then eligibility is applied in ascending i1.  A short last group repeats its last row, masked off.
        constexpr int R = GRID_ROWS_PER_TRIP;
        while (nz) {
            int us[R];
            bool valid[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                valid[k] = nz != 0ull;
                us[k] = valid[k] ? __builtin_ctzll(nz) : us[k > 0 ? k - 1 : 0];
                if (valid[k]) nz &= nz - 1ull;
            }
            bool on[R];
            uint32_t d[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int u = us[k];
                const unsigned long long mask =
                    ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(cur >> 32), u) << 32) |
                    (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(cur & 0xFFFFFFFFull), u);
                on[k] = valid[k] && live && ((mask >> lane) & 1ull);
                if (LINES) {
                    auto bcast = [&](double v) -> double {
                        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xFFFFFFFFull), u);
                        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), u);
                        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                    };
                    const double dot = bcast(qvx) * dirx + bcast(qvy) * diry;
                    if (fabs(dot) < a.line_sim_th) on[k] = false;  // NaN direction: comparison false, candidate kept
                }
                uint32_t q[8];
                q[0] = (uint32_t)__builtin_amdgcn_readlane((int)qa.x, u);
                q[1] = (uint32_t)__builtin_amdgcn_readlane((int)qa.y, u);
                q[2] = (uint32_t)__builtin_amdgcn_readlane((int)qa.z, u);
                q[3] = (uint32_t)__builtin_amdgcn_readlane((int)qa.w, u);
                q[4] = (uint32_t)__builtin_amdgcn_readlane((int)qb.x, u);
                q[5] = (uint32_t)__builtin_amdgcn_readlane((int)qb.y, u);
                q[6] = (uint32_t)__builtin_amdgcn_readlane((int)qb.z, u);
                q[7] = (uint32_t)__builtin_amdgcn_readlane((int)qb.w, u);
                uint32_t dd = __builtin_popcount(t0.x ^ q[0]);
                dd = bcnt_acc(t0.y ^ q[1], dd);
                dd = bcnt_acc(t0.z ^ q[2], dd);
                dd = bcnt_acc(t0.w ^ q[3], dd);
                dd = bcnt_acc(t1.x ^ q[4], dd);
                dd = bcnt_acc(t1.y ^ q[5], dd);
                dd = bcnt_acc(t1.z ^ q[6], dd);
                dd = bcnt_acc(t1.w ^ q[7], dd);
                d[k] = dd;
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (!on[k]) continue;  // per lane
                const int i1 = base + us[k];
                bool eligible = true;
                if (a.mutual) {  // bestLRMatches: running strict minimum per right feature (:145-150)
                    eligible = (int)d[k] < run_min;
                    if (eligible) {
                        run_min = (int)d[k];
                        owner = i1;
                    }
                }
                if (eligible) {
                    const uint32_t key = (d[k] << 16) | (uint32_t)i2;
                    if (PASS == 1) {
                        if (a.elig) {  // (mutual only: the eligible pairs of a lane are its strict running minima, ~ln(rows) of them)
                            if (n_elig < GRID_ELIG) a.elig[(size_t)n_elig * a.elig_stride + p] = ((uint32_t)i1 << 16) | d[k];
                            ++n_elig;
                        }
                        atomicMin(best_key + 2 * (size_t)i1, key);  // result unused: no-return atomic
                    } else {
                        const uint32_t bk = (uint32_t)__builtin_amdgcn_readlane((int)qk, us[k]);
                        if (key != bk) {
                            // :160 for the pair (best, this candidate): best_d < best_d2 * minRatio12P in DOUBLE
                            const double best_d = (double)(int)(bk >> 16), d2 = (double)(int)d[k];
                            if (!(best_d < d2 * a.ratio)) best_key[2 * (size_t)i1 + 1] = 1u;
                        }
                    }
                }
            }
        }
    }
    if (PASS == 1 && live) a.owner2[i2] = owner;
    if (PASS == 1 && a.elig && live) {
        a.elig_cnt[p] = n_elig < GRID_ELIG ? n_elig : GRID_ELIG;
        if (n_elig > GRID_ELIG) a.ovf[0] = 1;  // benign race: every writer stores 1
    }
}

template <bool LINES, int PASS, bool RANGE>
__global__ __launch_bounds__(256) void grid_scan_kernel(GridBatch g) {
    grid_scan_body<LINES, PASS, RANGE>(frame_view(g, blockIdx.y), blockIdx.x * 256 + threadIdx.x);
}

// Pass 2 of the single-scan formulation: lane = right feature (scan position p); for each eligible pair (i1, d) it met in
// pass 1 that is not i1's best: :160 for the pair (best, this candidate), best_d < d * minRatio12P in DOUBLE, else block i1.
__device__ __forceinline__ void grid_elig_body(const GridArgs& a, const int p) {
    if (p >= a.n2 || a.ovf[0] != 0) return;
    const int i2 = a.perm[p];
    const int cnt = a.elig_cnt[p];
    uint32_t* __restrict__ best_key = reinterpret_cast<uint32_t*>(a.top2);
    for (int s = 0; s < cnt; ++s) {
        const uint32_t e = a.elig[(size_t)s * a.elig_stride + p];
        const uint32_t i1 = e >> 16, d = e & 0xFFFFu;
        const uint32_t bk = best_key[2 * (size_t)i1];
        if (((d << 16) | (uint32_t)i2) != bk) {
            const double best_d = (double)(int)(bk >> 16), d2 = (double)(int)d;
            if (!(best_d < d2 * a.ratio)) best_key[2 * (size_t)i1 + 1] = 1u;
        }
    }
}

__global__ __launch_bounds__(256) void grid_elig_kernel(GridBatch g) {
    grid_elig_body(frame_view(g, blockIdx.y), blockIdx.x * 256 + threadIdx.x);
}

__device__ __forceinline__ void grid_finalize_body(const GridArgs& a, const int i1, const int stride1) {
    if (i1 >= stride1) return;
    if (i1 >= a.n1) {
        a.m12[i1] = -1;
        return;
    }
    const unsigned long long t = a.top2[i1];
    const uint32_t b = (uint32_t)t, blocked = (uint32_t)(t >> 32);
    int m = -1;
    // :160 — decided pairwise by scan pass 2; with a single eligible candidate best_d2 stays INT_MAX
    if (b != 0xFFFFFFFFu && !blocked && (double)(int)(b >> 16) < 2147483647.0 * a.ratio) m = (int)(b & 0xFFFFu);
    if (a.mutual && m >= 0 && a.owner2[m] != i1) m = -1;  // :166-174
    a.m12[i1] = m;
}

__global__ __launch_bounds__(256) void grid_finalize_kernel(GridBatch g) {
    grid_finalize_body(frame_view(g, blockIdx.y), blockIdx.x * 256 + threadIdx.x, g.stride1);
}

// ---- points, one-row windows, at most 2048 x 2048 key-points: the whole matcher of one frame in ONE workgroup -------------------------
// The scan above gives every left row a whole wave although only ~6 of its 64 lanes hold a candidate.  Here a THREAD owns a
// RIGHT feature (two per thread, in scan = cell order) and walks its own candidates: the left key-points of the cells
// x .. x + w_lo of its grid row, which are one contiguous range of a cell-ordered copy of the left descriptors in LDS
// (point_cells_kernel counting-sorts the left indices; neighbouring lanes read neighbouring rows).  Every lane computes a
// distance in every trip.  The order dependence of :145-150 (a pair takes part only if it strictly improves on every earlier
// left row that met the same right feature) is then local to the thread: it sorts its (left row << 9 | distance) keys in
// registers (bitonic network, 16 slots) and takes the strict prefix minima — the eligible pairs; the last one is matches_21
// (:148).  Eligible pairs meet their left rows through LDS: atomic min of (d << 16 | position) = best (:151-154), then one more
// look at every eligible pair that is not the best for the ratio test (:160, pairwise as in the scan formulation), then one
// thread per left row applies the mutual check (:166-174).  A right feature with more than 16 candidates (rare, ~8 per frame)
// is queued in LDS and taken by a whole wave after the barrier, lane = candidate, keys in LDS (handled inline by its own wave
// it made that wave the straggler of the phase: every other wave waited 13 k of a frame's 78 k cycles at the barrier).  Of a
// thread's sorted keys only the eligible ones outlive the phase (four in registers, more in LDS): with the 16 keys of both
// features live across the barrier the prefetched rows of the next frame were spilled — and the spill waited for them.
// Frames with more wide features or keys than the LDS holds are flagged and run the scan formulation above by the same
// workgroup after its last frame (fused_misfit_frame).  Round 3: 0.172 -> 0.133 ms per 1024 frames.
constexpr int FUSED_T = 1024, FUSED_ROWS = 2048, FUSED_REG = 16, FUSED_R = FUSED_ROWS / FUSED_T;
constexpr int FUSED_LW = GRID_LW;  // left key-points up to 16 columns right of the grid still have candidates
constexpr int FUSED_PADDED = FUSED_ROWS + FUSED_REG;          // the unrolled walks read up to 15 rows past a range
constexpr int FUSED_KEY_CAP = 8192;                           // keys of the right features with more than 16 candidates, whole frame
constexpr int FUSED_WQ = 256, FUSED_WQ_WORDS = 12;              // queue of the right features with more than 16 candidates: p, la, cnt, xoff, row
constexpr size_t FUSED_LDS = (size_t)FUSED_PADDED * (16 + 16 + 2) + (size_t)FUSED_ROWS * (4 + 2 + 2 + 1) + (size_t)FUSED_KEY_CAP * 4 +
                             (size_t)FUSED_WQ * FUSED_WQ_WORDS * 4;

// the scan formulation for a frame the fused kernel flagged (run by the same workgroup after its last frame): 16 waves over the
// blocks of 64 scan positions
__device__ __forceinline__ void fused_misfit_frame(const GridBatch& g, const int f) {
    const GridArgs a = frame_view(g, f);
    const int stride1 = g.stride1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blocks = (a.n2 + 63) >> 6;
    if (g.lean_cells) {  // what the lean point_cells_kernel leaves out: empty top-2 records, the candidate range of every left row
        const int32_t* cs = g.cell_start + (size_t)f * (STVO_GRID_CELLS + 1);
        const uint32_t* ls = g.lstart + (size_t)f * GRID_LSTART_STRIDE;
        const int32_t* lp = g.lperm + (size_t)f * g.stride1;
        int32_t* rg = const_cast<int32_t*>(g.range1) + (size_t)f * g.stride1 * 2;
        for (int i = tid; i < stride1; i += FUSED_T) {
            a.top2[i] = kTop2Empty;
            rg[2 * i] = 0;  // not placed on the extended grid: no candidates
            rg[2 * i + 1] = 0;
        }
        if (tid == 0 && a.ovf) *a.ovf = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int n_in = cs[STVO_GRID_CELLS];
        for (int c = tid; c < GRID_LCELLS; c += FUSED_T) {  // GridStructure::get of a left key-point of extended cell c (src/gridStructure.cpp:65-76)
            const int y = c / GRID_LW, x = c - y * GRID_LW;
            const int min_x = min(max(0, x - g.w.w_lo), STVO_GRID_COLS), max_x = max(min(STVO_GRID_COLS, x + 1), min_x);
            const int c0 = y * STVO_GRID_COLS + min_x, c1 = y * STVO_GRID_COLS + max_x;
            const int lo = c0 < STVO_GRID_CELLS ? cs[c0] : n_in, hi = c1 < STVO_GRID_CELLS ? cs[c1] : n_in;
            for (uint32_t pos = ls[c]; pos < ls[c + 1]; ++pos) {
                const int i1 = lp[pos];
                rg[2 * i1] = lo;
                rg[2 * i1 + 1] = hi;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    for (int w = wave; w < blocks; w += FUSED_T / 64) grid_scan_body<false, 1, true>(a, w * 64 + lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (a.elig)
        for (int w = wave; w < blocks; w += FUSED_T / 64) grid_elig_body(a, w * 64 + lane);
    for (int w = wave; w < blocks; w += FUSED_T / 64) grid_scan_body<false, 2, true>(a, w * 64 + lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int i1 = tid; i1 < stride1; i1 += FUSED_T) grid_finalize_body(a, i1, stride1);
}

// native 16-byte vector: an LDS load of it is one ds_read_b128 (HIP's uint4 is a struct, loaded member by member)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 lds_row(const uint4* p) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t hamming256(const uint4& t0, const uint4& t1, const uint4& q0, const uint4& q1) {
    uint32_t dd = __builtin_popcount(t0.x ^ q0.x);
    dd = bcnt_acc(t0.y ^ q0.y, dd);
    dd = bcnt_acc(t0.z ^ q0.z, dd);
    dd = bcnt_acc(t0.w ^ q0.w, dd);
    dd = bcnt_acc(t1.x ^ q1.x, dd);
    dd = bcnt_acc(t1.y ^ q1.y, dd);
    dd = bcnt_acc(t1.z ^ q1.z, dd);
    dd = bcnt_acc(t1.w ^ q1.w, dd);
    return dd;
}

// ascending bitonic network over 16 registers
__device__ __forceinline__ void sort16(uint32_t (&k)[FUSED_REG]) {
#pragma unroll
    for (int size = 2; size <= FUSED_REG; size <<= 1) {
#pragma unroll
        for (int j = size >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < FUSED_REG; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t lo = min(k[i], k[l]), hi = max(k[i], k[l]);
                    const bool asc = (i & size) == 0;
                    k[i] = asc ? lo : hi;
                    k[l] = asc ? hi : lo;
                }
            }
        }
    }
}

constexpr int FUSED_EK = 4;
constexpr uint32_t FUSED_NOKEY = 0xFFFFFFFFu, FUSED_FLAG = 1u << 30;  // key = left row << 9 | distance (0..256)

// what a workgroup fetches ahead for its next frame while it works on the current one (all 256 workgroups of a launch run
// in step, so without this the HBM sits idle during the compute phases and every frame start waits for a burst)
struct FusedNext {
    int i1[FUSED_R], i2[FUSED_R], rc[FUSED_R];   // level 1: left row at this thread's two cell-order positions, right feature and its cell at its two scan positions
    u32x4 l0[FUSED_R], l1[FUSED_R];        // level 2: the left descriptor rows (native vectors: a uint4 member is copied with memcpy
    u32x4 q0[FUSED_R], q1[FUSED_R];        //          and one of them stayed in scratch), the right descriptor rows,
    int la[FUSED_R], lb[FUSED_R];          //          the candidate range [la, lb) of the right features in cell-order positions
};

// level 0: the counts of a frame, fetched one frame earlier still (fused_fetch1 used to load them itself and wait a round trip)
struct FusedHdr {
    int n_placed, n2, n1;
};
__device__ __forceinline__ FusedHdr fused_hdr(const GridBatch& g, const int f) {
    FusedHdr h;
    h.n_placed = (int)g.lstart[(size_t)f * GRID_LSTART_STRIDE + GRID_LCELLS];
    h.n2 = g.n2[f];
    h.n1 = g.n1[f];
    return h;
}

// an opaque copy of a value: address arithmetic that starts from it is redone where it is used instead of being hoisted out of the
// frame loop (hoisted, the allocator spilled it — and every scratch reload waits for ALL memory operations in flight)
__device__ __forceinline__ int fresh(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// Both levels load unpredicated from clamped positions and keep the RAW values; validity (position < the frame's counts, which are
// scalars) is applied where a value is consumed.  Straight-line loads keep the compiler's count of the loads in flight exact
// (behind a predicated load it waits for everything), and a select right after a load would wait for it.
__device__ __forceinline__ void fused_fetch1(const GridBatch& g, const int f, FusedNext& n) {
    const int tid = fresh(threadIdx.x);
    const int32_t* __restrict__ cell2 = g.cell2 + (size_t)f * g.stride2;
    const int32_t* __restrict__ lperm = g.lperm + (size_t)f * g.stride1;
    const int32_t* __restrict__ perm = g.perm + (size_t)f * g.stride2;
#pragma unroll
    for (int r = 0; r < FUSED_R; ++r) {
        const int pos = tid + r * FUSED_T;
        const int p1 = pos < g.stride1 ? pos : g.stride1 - 1, p2 = pos < g.stride2 ? pos : g.stride2 - 1;
        n.i1[r] = lperm[p1];
        n.i2[r] = perm[p2];
        n.rc[r] = cell2[p2];
    }
}

__device__ __forceinline__ void fused_fetch2(const GridBatch& g, const int f, const FusedHdr& h, FusedNext& n) {
    const int tid = fresh(threadIdx.x);
    const u32x4* __restrict__ D1 = reinterpret_cast<const u32x4*>(g.d1) + (size_t)f * g.stride1 * 2;
    const u32x4* __restrict__ D2 = reinterpret_cast<const u32x4*>(g.d2) + (size_t)f * g.stride2 * 2;
    const uint32_t* __restrict__ lstart = g.lstart + (size_t)f * GRID_LSTART_STRIDE;
#pragma unroll
    for (int r = 0; r < FUSED_R; ++r) {
        const int pos = tid + r * FUSED_T;
        // invalid positions read row 0 / cell 0 of the frame and nobody uses the result
        const int i1 = pos < h.n_placed ? n.i1[r] : 0, i2 = pos < h.n2 ? n.i2[r] : 0, rc = pos < h.n2 ? n.rc[r] : 0;
        n.l0[r] = D1[2 * i1];
        n.l1[r] = D1[2 * i1 + 1];
        n.q0[r] = D2[2 * i2];
        n.q1[r] = D2[2 * i2 + 1];
        // GridStructure::get seen from the right feature: left cells x .. x + w_lo of its row (the window is clamped, :67-71)
        const int y = rc >> 6, x = rc & (STVO_GRID_COLS - 1);
        n.la[r] = (int)lstart[y * FUSED_LW + x];
        n.lb[r] = (int)lstart[y * FUSED_LW + x + g.w.w_lo + 1];
    }
}

// The tail of the stereo association for frame f, one thread per left row (rows tid, tid + 1024): mm[r] = its stereo match.  Kept
// rows keep ascending left index (stereoFrame.cpp:161-172): ballot ranks within a wave, wave counts through LDS.  The caller
// separates two calls by a barrier (s_cnt).
__device__ __forceinline__ void fused_tail(const GridBatch& g, const int f, const int n1, const int (&mm)[FUSED_R], int* s_cnt) {
    const PointTail& t = g.tail;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // the addresses below are computed here, per frame: hoisted out of the frame loop they were spilled
                                   // (and pushed a prefetched row into scratch, whose store waits for the row)
    const int lane = tid & 63, wv = tid >> 6;
    const size_t off = (size_t)f * g.stride1;
    bool ok[FUSED_R];
    double disp[FUSED_R];
    int before[FUSED_R];
#pragma unroll
    for (int r = 0; r < FUSED_R; ++r) {
        const int i = tid + r * FUSED_T;
        ok[r] = false;
        disp[r] = 0.0;
        if (i < n1 && mm[r] >= 0) ok[r] = point_tail_filter(t, off, i, mm[r], disp[r]);
        const unsigned long long bal = __ballot(ok[r]);
        before[r] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[r * (FUSED_T / 64) + wv] = __popcll(bal);
    }
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int r = 0; r < FUSED_R; ++r) {
        int wbase = base;
        for (int w = 0; w < FUSED_T / 64; ++w) {
            const int c = s_cnt[r * (FUSED_T / 64) + w];
            wbase += w < wv ? c : 0;
            base += c;
        }
        if (ok[r]) point_tail_write(t, off, tid + r * FUSED_T, off + (size_t)(wbase + before[r]), disp[r]);
    }
    if (tid == 0) {
        t.n[f] = base;
        if (t.host_n) t.host_n[f] = base;
        if (t.zero_nl) t.nl[f] = 0;
    }
}

// Persistent: workgroup w takes frames w, w + gridDim.x, ...
// CELLS (host: one frame per workgroup, g.fused_cells): the workgroup first builds the grid of its frame (point_cells.h) in the LDS
// the matcher uses afterwards; the matcher then reads the cell tables back through L2 like those of a point_cells_kernel launch.
template <bool CELLS>
__global__ __launch_bounds__(FUSED_T) void grid_points_fused_kernel(GridBatch g, const int key_cap) {
    extern __shared__ uint4 s_fused[];
    static_assert(sizeof(PointCellsLds<FUSED_T>) <= FUSED_LDS, "the cells phase borrows the matcher's LDS");
    if (CELLS) {
        if ((int)blockIdx.x >= g.B) return;
        point_cells_frame<FUSED_T, true>(g.cells, blockIdx.x, reinterpret_cast<PointCellsLds<FUSED_T>*>(s_fused));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __shared__ int s_cnt[FUSED_R * (FUSED_T / 64)];  // fused_tail: kept rows per wave
    __shared__ int s_ctl[3];  // [0] bump allocator of the key slots, [1] the frame misfits, [2] entries in the queue of wide right features
    uint4* s_llo = s_fused;                                                              // [pos] first / second half of the left rows,
    uint4* s_lhi = s_llo + FUSED_PADDED;                                                 //       cell order
    uint32_t* s_best = reinterpret_cast<uint32_t*>(s_lhi + FUSED_PADDED);                // [left row] min (d << 16 | scan position)
    uint32_t* s_keys = s_best + FUSED_ROWS;                                              // keys of the wide right features
    unsigned short* s_lperm = reinterpret_cast<unsigned short*>(s_keys + FUSED_KEY_CAP); // [pos] -> left row
    unsigned short* s_owner = s_lperm + FUSED_PADDED;                                    // [scan position] matches_21 (:148)
    unsigned short* s_rperm = s_owner + FUSED_ROWS;                                      // [scan position] -> right row
    unsigned char* s_blocked = reinterpret_cast<unsigned char*>(s_rperm + FUSED_ROWS);   // [left row] ratio test failed
    uint32_t* s_wq = reinterpret_cast<uint32_t*>(s_blocked + FUSED_ROWS);                // [entry][12] wide right features
    const int tid = threadIdx.x;
    FusedNext nx;
    // frames of this workgroup: a static stride, or (GridBatch::dyn_ctr) tickets from a counter — three drawn at the start (the frame
    // worked on, the one being fetched, the one whose header is requested), one more per frame
    const bool dyn = !CELLS && g.dyn_ctr != nullptr;
    __shared__ int s_ticket;
    int f = blockIdx.x, f_n1 = f + (int)gridDim.x, f_n2 = f + 2 * (int)gridDim.x;
    if (dyn) {
        int32_t* ctr = g.dyn_ctr + g.dyn_par;
        if (tid == 0) {
            if (blockIdx.x == 0) g.dyn_ctr[g.dyn_par ^ 1] = 0;  // the next launch's counter (the previous launch is complete: stream order)
            s_ticket = atomicAdd(ctr, 3);
        }
        __syncthreads();
        f = __builtin_amdgcn_readfirstlane(s_ticket);
        f_n1 = f + 1;
        f_n2 = f + 2;
        __syncthreads();  // (s_ticket is written again inside the loop)
    }
    if (f >= g.B) return;
    unsigned long long misfit_mask = 0ull;  // bit i: the i-th frame of this workgroup is left to the scan formulation (block-uniform)
    int trip = 0;
    FusedHdr hn = fused_hdr(g, f);
    fused_fetch1(g, f, nx);
    FusedHdr hc;  // the header of the frame being worked on, in scalar registers
    hc.n_placed = __builtin_amdgcn_readfirstlane(hn.n_placed);
    hc.n2 = __builtin_amdgcn_readfirstlane(hn.n2);
    hc.n1 = __builtin_amdgcn_readfirstlane(hn.n1);
    fused_fetch2(g, f, hc, nx);
    if (f_n1 < g.B) hn = fused_hdr(g, f_n1);
    const bool mutual = g.mutual != 0;
    const double ratio = g.ratio;
    // Left rows of the fetched frame into LDS (cell order), and the header of the frame after it into scalar registers: ONE
    // unconditional wait for everything in flight.  Inside the loop this sits between the ratio tests and the last phase of the
    // frame before — nothing reads the rows then, the fetched data has long arrived, and no store is in flight yet: behind the
    // stores of the last phase (loads and stores return out of order with respect to each other, so the compiler can only wait
    // for ALL of them) the same wait cost the write latency of the whole tail at every frame start.
    FusedHdr hs = hc;
    auto commit_rows = [&](const FusedHdr& h) {
        const int t0 = fresh(tid);
#pragma unroll
        for (int r = 0; r < FUSED_R; ++r) {
            const int pos = t0 + r * FUSED_T;
            if (pos < h.n_placed) {
                *reinterpret_cast<u32x4*>(s_llo + pos) = nx.l0[r];
                *reinterpret_cast<u32x4*>(s_lhi + pos) = nx.l1[r];
                s_lperm[pos] = (unsigned short)nx.i1[r];
            }
        }
        hs.n_placed = __builtin_amdgcn_readfirstlane(hn.n_placed);
        hs.n2 = __builtin_amdgcn_readfirstlane(hn.n2);
        hs.n1 = __builtin_amdgcn_readfirstlane(hn.n1);
    };
    commit_rows(hc);  // afterwards hs = header of the workgroup's next frame (if any)
    bool any_misfit = false;
    int tk = 0;  // the ticket drawn during this iteration (read behind its first barrier; the next iteration's draw is two barriers later)
    for (; f < g.B; f = f_n1, f_n1 = f_n2, f_n2 = dyn ? tk : f_n2 + (int)gridDim.x) {
        // ---- commit the fetched frame, right side: this thread's two right rows stay in registers
        uint4 q0[FUSED_R], q1[FUSED_R];
        int la[FUSED_R], cnt[FUSED_R];
        const int t1 = fresh(tid);
#pragma unroll
        for (int r = 0; r < FUSED_R; ++r) {
            const int pos = t1 + r * FUSED_T;
            if (pos < hc.n2) s_rperm[pos] = (unsigned short)nx.i2[r];
            s_best[pos] = 0xFFFFFFFFu;
            s_blocked[pos] = 0;
            q0[r] = make_uint4(nx.q0[r].x, nx.q0[r].y, nx.q0[r].z, nx.q0[r].w);
            q1[r] = make_uint4(nx.q1[r].x, nx.q1[r].y, nx.q1[r].z, nx.q1[r].w);
            la[r] = pos < hc.n2 ? nx.la[r] : 0;
            cnt[r] = pos < hc.n2 ? nx.lb[r] - nx.la[r] : 0;
        }
        if (tid == 0) {
            s_ctl[0] = 0;
            s_ctl[1] = key_cap < 0;
            s_ctl[2] = 0;
            if (dyn) {  // (read at the loop's step expression, behind this iteration's barriers)
                s_ticket = atomicAdd(g.dyn_ctr + g.dyn_par, 1);
                g.dyn_owner[f] = (int)blockIdx.x;
            }
        }
        const int fn = f_n1;
        const bool more = fn < g.B;  // block-uniform
        if (more) {
            fused_fetch1(g, fn, nx);
            if (f_n2 < g.B) hn = fused_hdr(g, f_n2);
        }
        __syncthreads();
        if (dyn) tk = __builtin_amdgcn_readfirstlane(s_ticket);
        // ---- distances, eligible chains, best per left row
        // what the ratio tests after the barrier need of a right feature: its eligible keys — the first FUSED_EK stay in registers, further
        // ones (4 % of the features) go to the key slots of LDS; the 16 sorted keys themselves live for one r only
        uint32_t ek[FUSED_R][FUSED_EK];
        // its keys in the key slots of LDS, one word: count | offset << 12 | wide << 31 (wide: all its keys, flagged where eligible;
        // otherwise: the eligible keys beyond the FUSED_EK in registers)
        uint32_t kinfo[FUSED_R] = {};
#pragma unroll
        for (int r = 0; r < FUSED_R; ++r) {
            const int p = tid + r * FUSED_T;
            uint32_t key[FUSED_REG];
            uint32_t elig = 0u;
            bool wide = cnt[r] > FUSED_REG;
            if (wide) {  // slots for its keys; a frame with more of them than the LDS holds is left to the scan formulation
                const int xoff = atomicAdd(&s_ctl[0], cnt[r]);
                const int slot = atomicAdd(&s_ctl[2], 1);
                if (xoff + cnt[r] > key_cap || slot >= FUSED_WQ) {
                    s_ctl[1] = 1;
                    wide = false;
                    cnt[r] = 0;
                } else {  // taken after the barrier, spread over the waves (inline they made their wave the straggler of the phase)
                    kinfo[r] = (uint32_t)cnt[r] | ((uint32_t)xoff << 12) | 0x80000000u;
                    uint32_t* e = s_wq + slot * FUSED_WQ_WORDS;
                    e[0] = (uint32_t)p; e[1] = (uint32_t)la[r]; e[2] = (uint32_t)cnt[r]; e[3] = (uint32_t)xoff;
                    e[4] = q0[r].x; e[5] = q0[r].y; e[6] = q0[r].z; e[7] = q0[r].w;
                    e[8] = q1[r].x; e[9] = q1[r].y; e[10] = q1[r].z; e[11] = q1[r].w;
                }
            }
            const int c16 = wide ? 0 : cnt[r];
            const uint4* lo_row = s_llo + la[r];  // constant offsets from one base: LDS immediates, no address registers
            const uint4* hi_row = s_lhi + la[r];
            const unsigned short* perm_row = s_lperm + la[r];
#pragma unroll
            for (int k4 = 0; k4 < FUSED_REG; k4 += 4) {
                if (!__any(c16 > k4)) {  // wave-uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j) key[k4 + j] = FUSED_NOKEY;
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // past the range: a neighbour's row or padding, result masked
                    const uint32_t dd = hamming256(lds_row(lo_row + k4 + j), lds_row(hi_row + k4 + j), q0[r], q1[r]);
                    const uint32_t i1 = perm_row[k4 + j];
                    key[k4 + j] = k4 + j < c16 ? ((i1 << 9) | dd) : FUSED_NOKEY;
                }
            }
            uint32_t owner = 0xFFFFu;
            if (mutual) {  // :145-150 — ascending left row, strict running minimum
                sort16(key);
                uint32_t thr = 512u;
#pragma unroll
                for (int k = 0; k < FUSED_REG; ++k) {
                    const uint32_t dd = key[k] & 511u;
                    if (key[k] != FUSED_NOKEY && dd < thr) {
                        thr = dd;
                        owner = key[k] >> 9;
                        elig |= 1u << k;
                    }
                }
            } else {
                elig = (1u << c16) - 1u;
            }
            int extra = __builtin_popcount(elig) - FUSED_EK, eoff = 0;
            if (extra > 0) {
                eoff = atomicAdd(&s_ctl[0], extra);
                if (eoff + extra > key_cap) {
                    s_ctl[1] = 1;
                    extra = 0;
                } else {
                    kinfo[r] = (uint32_t)extra | ((uint32_t)eoff << 12);
                }
            }
#pragma unroll
            for (int j = 0; j < FUSED_EK; ++j) ek[r][j] = FUSED_NOKEY;
            int ne = 0;
#pragma unroll
            for (int k = 0; k < FUSED_REG; ++k)
                if (elig & (1u << k)) {
                    atomicMin(&s_best[key[k] >> 9], ((key[k] & 511u) << 16) | (uint32_t)p);
#pragma unroll
                    for (int j = 0; j < FUSED_EK; ++j) ek[r][j] = ne == j ? key[k] : ek[r][j];
                    if (k >= FUSED_EK && ne >= FUSED_EK && ne - FUSED_EK < extra) s_keys[eoff + ne - FUSED_EK] = key[k];
                    ++ne;
                }
            if (!wide) s_owner[p] = (unsigned short)owner;
            if (wide && !mutual) s_owner[p] = 0xFFFFu;
        }
        if (more) fused_fetch2(g, fn, hs, nx);  // lands during the two phases below
        __syncthreads();
        const bool misfit = s_ctl[1] != 0;  // block-uniform
        // right features with more than 16 candidates (rare, ~8 per frame): one wave each — keys to LDS (lane = candidate), then
        // every lane decides its candidates against all the others: eligible unless an earlier left row is at least as close
        // (:145-150), matches_21 = the eligible one no later row beats
        const int n_wide = s_ctl[2];  // block-uniform
        if (!misfit && n_wide > 0) {
            for (int ent = tid >> 6; ent < n_wide; ent += FUSED_T / 64) {
                const uint32_t* e = s_wq + ent * FUSED_WQ_WORDS;
                const uint32_t w_p = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[0]);
                const int w_la = __builtin_amdgcn_readfirstlane((int)e[1]), w_cnt = __builtin_amdgcn_readfirstlane((int)e[2]);
                const int w_xoff = __builtin_amdgcn_readfirstlane((int)e[3]);
                const uint4 wq0 = make_uint4(e[4], e[5], e[6], e[7]), wq1 = make_uint4(e[8], e[9], e[10], e[11]);
                uint32_t* kk = s_keys + w_xoff;
                for (int k = tid & 63; k < w_cnt; k += 64) {
                    const int pos = w_la + k;
                    kk[k] = ((uint32_t)s_lperm[pos] << 9) | hamming256(lds_row(s_llo + pos), lds_row(s_lhi + pos), wq0, wq1);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int k = tid & 63; k < w_cnt; k += 64) {
                    const uint32_t mine = kk[k] & ~FUSED_FLAG, i1 = mine >> 9, dd = mine & 511u;
                    bool dominated = false, later_better = false;
                    for (int j = 0; j < w_cnt; ++j) {
                        const uint32_t v = kk[j], vi = (v >> 9) & 2047u, vd = v & 511u;
                        dominated |= vi < i1 && vd <= dd;
                        later_better |= vi > i1 && vd < dd;
                    }
                    if (!mutual || !dominated) {
                        kk[k] = mine | FUSED_FLAG;
                        atomicMin(&s_best[i1], (dd << 16) | w_p);
                        if (mutual && !later_better) s_owner[w_p] = (unsigned short)i1;
                    }
                }
            }
            __syncthreads();
        }
        if (tid == 0) g.misfit[f] = misfit;
        if (misfit && trip < 64) misfit_mask |= 1ull << trip;
        any_misfit |= misfit;
        ++trip;
        if (!misfit) {
            // ---- :160 for every eligible pair that is not its left row's best: best_d < d * minRatio12P in DOUBLE, else the row is out
#pragma unroll
            for (int r = 0; r < FUSED_R; ++r) {
                const uint32_t p = (uint32_t)(tid + r * FUSED_T);
                auto judge = [&](uint32_t k, uint32_t pos) {
                    const uint32_t i1 = (k >> 9) & 2047u, dd = k & 511u;
                    const uint32_t bk = s_best[i1];
                    if (bk != ((dd << 16) | pos)) {
                        const double best_d = (double)(int)(bk >> 16), d2 = (double)(int)dd;
                        if (!(best_d < d2 * ratio)) s_blocked[i1] = 1;
                    }
                };
#pragma unroll
                for (int j = 0; j < FUSED_EK; ++j)
                    if (ek[r][j] != FUSED_NOKEY) judge(ek[r][j], p);
                if (!(kinfo[r] >> 31))
                    for (uint32_t j = 0; j < (kinfo[r] & 4095u); ++j) judge(s_keys[((kinfo[r] >> 12) & 0x3FFFFu) + j], p);
                for (unsigned long long wm = __ballot((kinfo[r] >> 31) != 0u); wm; wm &= wm - 1ull) {  // the wide ones, lane = candidate
                    const int L = __builtin_ctzll(wm);
                    const uint32_t w_info = (uint32_t)__builtin_amdgcn_readlane((int)kinfo[r], L);
                    const int w_cnt = (int)(w_info & 4095u), w_xoff = (int)((w_info >> 12) & 0x3FFFFu);
                    const uint32_t w_p = (uint32_t)((tid & ~63) + L + r * FUSED_T);
                    for (int k = tid & 63; k < w_cnt; k += 64) {
                        const uint32_t v = s_keys[w_xoff + k];
                        if (v & FUSED_FLAG) judge(v & ~FUSED_FLAG, w_p);
                    }
                }
            }
        }
        const FusedHdr hfn = hs;  // header of frame fn: fused_fetch2 above used it, the next iteration works on it
        if (more) commit_rows(hfn);  // hs becomes the header of frame fn + gridDim.x
        __syncthreads();
        // ---- one thread per left row: accept unless blocked, mutual check (:166-174); then the tail of the association
        if (!misfit) {
            int mm[FUSED_R];
#pragma unroll
            for (int r = 0; r < FUSED_R; ++r) {
                const int i1 = tid + r * FUSED_T;
                mm[r] = -1;
                if (i1 >= g.stride1) continue;
                const uint32_t bk = s_best[i1];
                if (i1 < hc.n1 && bk != 0xFFFFFFFFu && !s_blocked[i1] && (double)(int)(bk >> 16) < 2147483647.0 * ratio) {
                    const int pb = (int)(bk & 0xFFFFu);
                    if (!mutual || (int)s_owner[pb] == i1) mm[r] = (int)s_rperm[pb];
                }
                g.m12[(size_t)f * g.stride1 + i1] = mm[r];
            }
            if (g.has_tail) fused_tail(g, f, hc.n1, mm, s_cnt);
        }
        if (more) __syncthreads();  // the next frame's commit overwrites what the phase above reads
        hc = hfn;
    }
    // frames that did not fit (rare): the scan formulation, by the workgroup that flagged them — no second launch
    if (dyn ? any_misfit : (misfit_mask != 0ull || trip > 64)) {
        int i = 0;
        if (dyn) {  // the frames this workgroup took are the ones it signed (its own stores, made visible to all its threads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        for (int fm = dyn ? 0 : (int)blockIdx.x; fm < g.B; fm += dyn ? 1 : (int)gridDim.x, ++i) {
            const bool todo = dyn ? (g.dyn_owner[fm] == (int)blockIdx.x && g.misfit[fm] != 0)
                                  : (i < 64 ? ((misfit_mask >> i) & 1ull) != 0ull : g.misfit[fm] != 0);  // (beyond 64 frames per workgroup: the flag it stored)
            if (todo) {
                __syncthreads();
                fused_misfit_frame(g, fm);
                if (g.has_tail) {  // its matches come back through global memory
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __syncthreads();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    int mm[FUSED_R];
#pragma unroll
                    for (int r = 0; r < FUSED_R; ++r) {
                        const int i1 = threadIdx.x + r * FUSED_T;
                        mm[r] = i1 < g.stride1 ? g.m12[(size_t)fm * g.stride1 + i1] : -1;
                    }
                    fused_tail(g, fm, g.n1[fm], mm, s_cnt);
                }
            }
        }
    }
}

}  // namespace

bool grid_points_fused_ok(const GridBatch& g) {
    const bool range = g.range_points && g.range1 != nullptr;
    // STVO_GRID_FUSED=0: the scan formulation for every batch
    const bool fused = range && g.misfit && g.cell2 && g.lperm && g.lstart && g.stride1 <= FUSED_ROWS && g.stride2 <= FUSED_ROWS && g.w.w_lo >= 0 &&
                       g.w.w_lo <= FUSED_LW - STVO_GRID_COLS && g.w.w_hi == 0 && g.w.h_lo == 0 && g.w.h_hi == 0 && dbg().grid_fused != 0;
    return fused && lds_opt_in(reinterpret_cast<const void*>(grid_points_fused_kernel<false>), (int)FUSED_LDS) &&
           lds_opt_in(reinterpret_cast<const void*>(grid_points_fused_kernel<true>), (int)FUSED_LDS);
}

// cover (zeroed here) -> scan -> finalize for a batch of frame pairs
void launch_grid_batch(hipStream_t s, const GridBatch& g, bool lines, hipEvent_t* scan_events) {
    if (g.B <= 0 || g.stride1 <= 0 || g.stride2 <= 0) return;
    const dim3 g1((g.stride1 + 255) / 256, g.B), g2((g.stride2 + 255) / 256, g.B), gc((g.n1p + 255) / 256, g.B), blk(256);
    // LDS staging column per thread: words64 x 256 x 8 B per workgroup (64 KB at 2048 right features)
    const size_t lds = (size_t)g.words64 * 256 * sizeof(unsigned long long);
    const bool use_lds = lds <= ((size_t)64 << 10);
    if (lines) {
        if (use_lds)
            hipLaunchKernelGGL((grid_cover_kernel<true, true>), gc, blk, lds, s, g);
        else
            hipLaunchKernelGGL((grid_cover_kernel<true, false>), gc, blk, 0, s, g);
        if (scan_events && scan_events[0]) (void)hipEventRecord(scan_events[0], s);
        hipLaunchKernelGGL((grid_scan_kernel<true, 1, false>), g2, blk, 0, s, g);
        if (g.elig) hipLaunchKernelGGL(grid_elig_kernel, g2, blk, 0, s, g);
        hipLaunchKernelGGL((grid_scan_kernel<true, 2, false>), g2, blk, 0, s, g);
        if (scan_events) (void)hipEventRecord(scan_events[1], s);
    } else {
        const bool range = g.range_points && g.range1 != nullptr;
        if (grid_points_fused_ok(g)) {
            // STVO_GRID_FUSED_CAP: capacity for the keys of right features with more than 16 candidates (tests force the misfit path with -1)
            int cap = dbg().grid_fused_cap != DBG_UNSET ? dbg().grid_fused_cap : FUSED_KEY_CAP;
            if (cap > FUSED_KEY_CAP) cap = FUSED_KEY_CAP;  // negative: every frame misfits
            const int fused_wgs = device_cu_count();  // one persistent workgroup per CU (its LDS and registers fill one)
            if (scan_events && scan_events[0]) (void)hipEventRecord(scan_events[0], s);
            if (g.fused_cells && g.B <= fused_wgs)
                hipLaunchKernelGGL(grid_points_fused_kernel<true>, dim3(g.B), dim3(FUSED_T), FUSED_LDS, s, g, cap);
            else
                hipLaunchKernelGGL(grid_points_fused_kernel<false>, dim3(g.B < fused_wgs ? g.B : fused_wgs), dim3(FUSED_T), FUSED_LDS, s, g, cap);
            if (scan_events) (void)hipEventRecord(scan_events[1], s);
            return;
        }
        if (range) {
            if (scan_events && scan_events[0]) (void)hipEventRecord(scan_events[0], s);
            hipLaunchKernelGGL((grid_scan_kernel<false, 1, true>), g2, blk, 0, s, g);
            if (g.elig) hipLaunchKernelGGL(grid_elig_kernel, g2, blk, 0, s, g);
            hipLaunchKernelGGL((grid_scan_kernel<false, 2, true>), g2, blk, 0, s, g);
        } else {
            if (use_lds)
                hipLaunchKernelGGL((grid_cover_kernel<false, true>), gc, blk, lds, s, g);
            else
                hipLaunchKernelGGL((grid_cover_kernel<false, false>), gc, blk, 0, s, g);
            if (scan_events && scan_events[0]) (void)hipEventRecord(scan_events[0], s);
            hipLaunchKernelGGL((grid_scan_kernel<false, 1, false>), g2, blk, 0, s, g);
            if (g.elig) hipLaunchKernelGGL(grid_elig_kernel, g2, blk, 0, s, g);
            hipLaunchKernelGGL((grid_scan_kernel<false, 2, false>), g2, blk, 0, s, g);
        }
        if (scan_events) (void)hipEventRecord(scan_events[1], s);
    }
    hipLaunchKernelGGL(grid_finalize_kernel, g1, blk, 0, s, g);
}

// test hook: the candidate masks of the range formulation, written into g.cover in the bit-matrix layout
void launch_grid_range_debug(hipStream_t s, const GridBatch& g) {
    hipLaunchKernelGGL(grid_range_debug_kernel, dim3((g.n1p + 255) / 256, g.B), dim3(256), 0, s, g);
}

namespace {

template <bool LINES>
int run_grid(stvo_ctx* ctx, const int32_t* cell_xy1, const uint8_t* d1, int n1, const int32_t* cell_start,
             const int32_t* cell_items, const uint8_t* d2, int n2, const double* dir2, const stvo_grid_window* w,
             double ratio, double line_sim_th, int mutual, int32_t* m12, int32_t* n_matches) {
    if (!ctx || n1 < 0 || n2 < 0 || !cell_start || !w || (n1 > 0 && (!cell_xy1 || !d1 || !m12)) ||
        (n2 > 0 && (!d2 || (LINES && !dir2))))
        return STVO_ERR_INVALID_ARG;
    if (!(ratio <= 1.0)) return STVO_ERR_INVALID_ARG;  // candidate-order independence needs ratio <= 1
    if (n1 > ctx->max_rows || n2 > ctx->max_rows || n2 > STVO_MAX_ROWS_LIMIT) return STVO_ERR_CAPACITY;
    if (n_matches) *n_matches = 0;
    if (n1 == 0) return STVO_OK;
    if (n2 == 0) {  // no candidates at all (stereoFrame.cpp:126-127 guards this upstream)
        for (int i = 0; i < n1; ++i) m12[i] = -1;
        return STVO_OK;
    }
    const int n_items = cell_start[STVO_GRID_CELLS];
    if (n_items < 0) return STVO_ERR_INVALID_ARG;
    if (n_items > 0 && !cell_items) return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->arena_off = 0;
    ctx->upload_hi = 0;
    GridBatch a;
    std::memset(&a, 0, sizeof(a));
    int32_t *dxy, *dstart, *ditems, *downer, *dm12, *dn1, *dn2;
    uint8_t *dd1, *dd2;
    double* ddir = nullptr;
    TRY(upload(ctx, &dxy, cell_xy1, (size_t)n1 * (LINES ? 4 : 2)));
    TRY(upload(ctx, &dd1, d1, (size_t)n1 * STVO_DESC_BYTES));
    TRY(upload(ctx, &dstart, cell_start, (size_t)STVO_GRID_CELLS + 1));
    TRY(upload(ctx, &ditems, cell_items, (size_t)n_items));
    TRY(upload(ctx, &dd2, d2, (size_t)n2 * STVO_DESC_BYTES));
    if (LINES) TRY(upload(ctx, &ddir, dir2, (size_t)n2 * 2));
    TRY(upload(ctx, &dn1, &n1, 1));
    TRY(upload(ctx, &dn2, &n2, 1));
    a.B = 1;
    a.stride1 = n1;
    a.stride2 = n2;
    a.xy_width = LINES ? 4 : 2;
    a.items_stride = n_items;
    a.words64 = (n2 + 63) / 64;
    a.n1p = (n1 + 63) & ~63;
    // scan order of the right features = order of first appearance in the CSR grid (cell-major, i.e.
    // spatial); features that are in no cell come last (they are nobody's candidate anyway)
    std::vector<int32_t> rank((size_t)n2, -1), perm((size_t)n2);
    int np_ = 0;
    for (int k = 0; k < n_items; ++k) {
        const int id = cell_items[k];
        if (id >= 0 && id < n2 && rank[id] < 0) {
            rank[id] = np_;
            perm[np_++] = id;
        }
    }
    for (int id = 0; id < n2; ++id)
        if (rank[id] < 0) {
            rank[id] = np_;
            perm[np_++] = id;
        }
    int32_t *drank, *dperm;
    TRY(upload(ctx, &drank, rank.data(), (size_t)n2));
    TRY(upload(ctx, &dperm, perm.data(), (size_t)n2));
    a.rank = drank;
    a.perm = dperm;
    TRY(flush_uploads(ctx));
    if (mutual) {
        a.elig = arena_alloc<uint32_t>(ctx, (size_t)GRID_ELIG * n2);
        a.elig_cnt = arena_alloc<int32_t>(ctx, (size_t)n2);
        a.ovf = arena_alloc<int32_t>(ctx, 1);
        if (!a.elig || !a.elig_cnt || !a.ovf) return STVO_ERR_CAPACITY;
    }
    a.cover = arena_alloc<unsigned long long>(ctx, (size_t)a.n1p * a.words64);
    a.top2 = arena_alloc<unsigned long long>(ctx, (size_t)n1);
    downer = arena_alloc<int32_t>(ctx, (size_t)n2);
    dm12 = arena_alloc<int32_t>(ctx, (size_t)n1);
    if (!a.cover || !a.top2 || !downer || !dm12) return STVO_ERR_CAPACITY;
    a.cell_xy1 = dxy;
    a.d1 = dd1;
    a.n1 = dn1;
    a.cell_start = dstart;
    a.cell_items = ditems;
    a.d2 = dd2;
    a.n2 = dn2;
    a.dir2 = ddir;
    a.w = *w;
    a.ratio = ratio;
    a.line_sim_th = line_sim_th;
    a.mutual = mutual;
    a.owner2 = downer;
    a.m12 = dm12;
    launch_grid_batch(ctx->stream, a, LINES);
    TRY(check_launch(ctx));
    TRY(download_begin(ctx, dm12, (size_t)n1 * sizeof(int32_t)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(m12, host_mirror(ctx, dm12), (size_t)n1 * sizeof(int32_t));
    if (n_matches) {
        int c = 0;
        for (int i = 0; i < n1; ++i) c += m12[i] >= 0;
        *n_matches = c;
    }
    return STVO_OK;
}

}  // namespace
}  // namespace stvo

extern "C" {

int stvo_match_grid_points(stvo_ctx* ctx, const int32_t* cell_xy1, const uint8_t* d1, int n1, const int32_t* cell_start,
                           const int32_t* cell_items, const uint8_t* d2, int n2, const stvo_grid_window* w, double ratio,
                           int mutual, int32_t* m12, int32_t* n_matches) {
    return stvo::run_grid<false>(ctx, cell_xy1, d1, n1, cell_start, cell_items, d2, n2, nullptr, w, ratio, 0.0, mutual,
                                 m12, n_matches);
}

int stvo_match_grid_lines(stvo_ctx* ctx, const int32_t* cell_xy1, const uint8_t* d1, int n1, const int32_t* cell_start,
                          const int32_t* cell_items, const uint8_t* d2, int n2, const double* dir2,
                          const stvo_grid_window* w, double ratio, double line_sim_th, int mutual, int32_t* m12,
                          int32_t* n_matches) {
    return stvo::run_grid<true>(ctx, cell_xy1, d1, n1, cell_start, cell_items, d2, n2, dir2, w, ratio, line_sim_th,
                                mutual, m12, n_matches);
}

}  // extern "C"
