// lsd_kernels.hip — the LSD key-line detector on gfx950: what the reference obtains from
//     lsd->detect(img, lines, Config::lsdScale(), 1, opts)  + the top-N cut by response   (/root/reference/src/stereoFrame.cpp:219-240)
// through LSDDetectorC::detectImpl (3rdparty/line_descriptor/src/LSDDetector_custom.cpp:227-325) and cv::LineSegmentDetector.
// The detector core is third-party code that is not under /root/reference: the semantics are those of oracle/stvo_lsd_oracle.c
// (restated from the published algorithm, parity unpinned), against which these kernels are bit-exact (tests/test_gpu_lsd.py).
//
//   blur + resize        the scaled image (cv::GaussianBlur 7 x 7 + cv::resize on 8-bit data): orb_kernels.hip's kernels
//   lsd_gradient_kernel  ll_angle: 2 x 2 gradient, norm, level-line angle (fastAtan2, degrees), the float cos / sin a pixel
//                        contributes to a region angle, the largest squared gradient of the image (integer atomic max)
//   lsd_hist / lsd_scan / lsd_scatter_kernel   the pseudo-ordering: the defined pixels by bin of the gradient norm, highest bin first,
//                        row-major inside a bin (oracle note (1)) — a stable counting sort written here (until round 5: rocPRIM's radix sort)
//   lsd_grow_kernel      the search: region_grow + region2rect for every unused seed in that order.  Inherently sequential
//                        per image — whether a pixel joins depends on the running region angle, which changes with every pixel
//                        added, and on what all earlier regions took — so ONE wavefront walks an image and the batch supplies
//                        the parallelism.  Inside the wave the work of one step is spread over the lanes: the 9 neighbours of
//                        7 consecutive region points are fetched by 63 lanes at once (one memory round trip per 7 points) and
//                        then resolved in order with ballots; the sums of region2rect are accumulated strictly in region
//                        order (lane-parallel products, serial additions through v_readlane), so every rounding is the oracle's
//   lsd_grow_xcd_kernel  the same search for batches of <= 128 images as an exact speculate / commit protocol over the CUs of one XCD per
//                        image: a committing wave (LDS bitmap), a dispatcher, a feeder, speculating workgroups (9 ms instead of 63 per image)
//   lsd_grow_refine_kernel  lsd_refine = 1 (LSD_REFINE_STD): the plain search with refine / reduce_region_radius — a sparse region gives its
//                        pixels back, is grown again under a tolerance from the angles near its seed, then cut back by radius; one wave per
//                        image for every batch size (flags that turn off break the strike-off and the speculation of the other forms)
//   lsd_keylines_kernel  the wrapper: checkLineExtremes, length, min_length, KeyLine fields (numOfPixels = the CLIPPED cv::LineIterator
//                        count), top-N by response (stable)
// Byte / integer work except where the source computes in floating point; no fused multiply-adds outside the two of the sine /
// cosine reduction, which the oracle has too.
#include <cmath>
#include <new>
#include <vector>

#include "ctx_internal.h"
#include "fast_atan2.h"

#pragma clang fp contract(off)

namespace stvo {
namespace {

constexpr double LSD_PI = 3.14159265358979323846;
constexpr double LSD_DEG2RAD = LSD_PI / 180;
constexpr double LSD_3_2_PI = (3 * LSD_PI) / 2;
constexpr double LSD_2_PI = 2 * LSD_PI;
constexpr uint32_t LSD_NOKEY = 0xFFFFFFFFu;
constexpr int LSD_IDX_BITS = 20;  // pixel index inside a sort key: scaled images up to 2^20 pixels
constexpr int LSD_ROWS = 16;      // image rows per workgroup of the per-pixel kernels

// orc_sincos_det (oracle/stvo_lsd_oracle.c), operation for operation
__device__ __forceinline__ void sincos_det(double x, double& s, double& c) {
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17, TWO_OVER_PI = 6.36619772367581382433e-01;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double k = __builtin_rint(x * TWO_OVER_PI);
    double r = __builtin_fma(-k, PIO2_HI, x);
    r = __builtin_fma(-k, PIO2_LO, r);
    const double z = r * r;
    const double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double sn = r + (z * r) * (S1 + z * ps);
    const double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double cs = 1.0 - (0.5 * z - z * pc);
    const int q = (int)((long long)k & 3);
    s = q == 0 ? sn : (q == 1 ? cs : (q == 2 ? -sn : -cs));
    c = q == 0 ? cs : (q == 1 ? -sn : (q == 2 ? -cs : sn));
}

// A pixel as the region growing sees it.  Three arrays (angle, (cos, sin), flag) were three divergent 64-lane gathers per sub-group
// and round, and the batches are bound by exactly that — the address path of a CU shared by its sixteen waves (SQ counters, round 6:
// the vector port of a SIMD is 29 % taken at four images per SIMD, a round takes 3.3 x as long as alone).
struct LsdPx {
    float ang;     // level-line angle in degrees, < 0: undefined
    float c, s;    // cos / sin of float(angle in radians) as floats: what the pixel adds to a region's direction sums
    int32_t used;  // 0 / 1: taken by a region (the only field written after lsd_gradient_kernel)
};
typedef float lsd_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ LsdPx px_ld(const LsdPx* p) {  // one 16-byte load
    const lsd_f4 v = *reinterpret_cast<const lsd_f4*>(p);
    LsdPx r;
    r.ang = v.x; r.c = v.y; r.s = v.z; r.used = __float_as_int(v.w);
    return r;
}

struct LsdDev {
    int B, w, h;               // the scaled image
    int cols, rows;            // the input image
    int n_bins, min_reg_size, seg_cap, K, nfeatures;
    double rho, prec, scale, min_length;
    const uint8_t* scaled;     // [B][h][w]
    LsdPx* px;                 // [B][w h] what the search reads of a pixel, ONE 16-byte record (one gather per neighbour instead of three)
    int32_t* k32;              // [B][w h] gx^2 + gy^2 of the defined pixels (the gradient norm is sqrt(k / 4): recomputed where it is used), -1: undefined
    uint32_t* cnt;             // [B][units][n_bins] the pseudo-ordering's counters: pixels per bin and unit, then the first rank of every bin in every unit
    uint32_t* order;           // [B][w h] (highest bin first) << 20 | pixel index of the defined pixels in the pseudo-ordering, LSD_NOKEY behind them
    int32_t* reg;              // [B][w h] the region being grown: x | y << 16
    int32_t* kmax;             // [B] largest gx^2 + gy^2 among the defined pixels, -1: none
    float4* seg;               // [B][seg_cap] (x1, y1, x2, y2) in detection order
    int32_t* n_seg;            // [B]
    int32_t* n_pass;           // [B] segments longer than min_length, before the top-N / capacity cut (stvo_lsd_counts)
    double* dbg;               // developer aid (stvo_lsd_debug): [B][seg_cap][8] cx, cy, Ixx, Iyy, Ixy, theta, l_min, l_max, or nullptr
    // outputs of the wrapper
    stvo_keyline* lines;       // [B][K]
    float* response;           // [B][K] or nullptr
    int32_t* n_lines;          // [B]
};

// ll_angle: one thread per pixel of the scaled image
__global__ __launch_bounds__(256) void lsd_gradient_kernel(LsdDev d) {
    const int x = blockIdx.x * 256 + threadIdx.x, b = blockIdx.z;
    const bool in_row = x < d.w;  // (no early return: the whole wave takes part in the reduction below)
    const size_t base = (size_t)b * d.w * d.h;
    int kw = -1;
    // LSD_ROWS rows per workgroup: one row each made 2.8 M workgroups per 1024 images and the launch dispatch-bound (25 ms)
    for (int y = blockIdx.y * LSD_ROWS; y < min((int)(blockIdx.y + 1) * LSD_ROWS, d.h); ++y) {
    const size_t q = base + (size_t)y * d.w + (in_row ? x : 0);
    float ang = -1.f;
    float2 cs = make_float2(0.f, 0.f);
    double norm = 0.0;
    int kdef = -1;
    if (x < d.w - 1 && y < d.h - 1) {
        const uint8_t* r0 = d.scaled + base + (size_t)y * d.w + x;
        const uint8_t* r1 = r0 + d.w;
        const int DA = (int)r1[1] - (int)r0[0], BC = (int)r0[1] - (int)r1[0];
        const int gx = DA + BC, gy = DA - BC;
        const int k = gx * gx + gy * gy;
        norm = sqrt((double)k / 4.0);
        if (!(norm <= d.rho)) {
            ang = fast_atan2_deg((float)gx, (float)-gy);
            const double a = (double)ang * LSD_DEG2RAD;
            double s, c;
            sincos_det((double)(float)a, s, c);
            cs = make_float2((float)c, (float)s);
            kdef = k;
        }
    }
    kw = max(kw, kdef);
    if (in_row) {
        lsd_f4 rec;
        rec.x = ang; rec.y = cs.x; rec.z = cs.y; rec.w = 0.f;  // (used = 0)
        *reinterpret_cast<lsd_f4*>(d.px + q) = rec;
        d.k32[q] = kdef;
    }
    }
    // the image's largest squared gradient: one atomic per wave, and only when it can raise the value
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kw = max(kw, __shfl_xor(kw, off, 64));
    if ((threadIdx.x & 63) == 0 && kw >= 0 && kw > __hip_atomic_load(&d.kmax[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&d.kmax[b], kw);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The pseudo-ordering — pixels by bin of their gradient norm, highest bin first, row-major inside a bin, undefined pixels not at all
// (src of the semantics: oracle/stvo_lsd_oracle.c note (1)) — as a stable counting sort of our own (round 6; until then a key array +
// rocPRIM's segmented radix sort: 36 of the 188 ms of 4096 images).  A wave owns a UNIT of 64 rows of 64 consecutive pixels:
//   lsd_hist_kernel     the pixels of every bin in the unit (LDS counters, wave-private);
//   lsd_scan_kernel     per image: first rank of every (bin, unit) = pixels of higher bins + pixels of the bin in earlier units; LSD_NOKEY
//                       behind the last defined pixel (the searches stop at the first batch that begins with it);
//   lsd_scatter_kernel  the unit again, row by row: a pixel's rank = the bin's running position + the lower lanes of its row with the same
//                       bin (the lanes with equal bins matched bit by bit with ballots) — stable by construction, no atomics on the way.
// The bin is (int)(norm (n_bins - 1) / max norm), norm = sqrt(k / 4) in double precision from the integer k both times.
constexpr int LSD_UNIT_ROWS = 64;
constexpr int LSD_UNIT = 64 * LSD_UNIT_ROWS;
constexpr int LSD_MAX_BINS = 2048;
__device__ __forceinline__ double lsd_bin_coef(const LsdDev& d, int b) {
    const int km = d.kmax[b];
    const double max_grad = km >= 0 ? sqrt((double)km / 4.0) : -1.0;
    return max_grad > 0 ? (double)(d.n_bins - 1) / max_grad : 0.0;
}
__device__ __forceinline__ int lsd_keybin(int k, double bin_coef, int n_bins) {
    int bin = (int)(sqrt((double)k / 4.0) * bin_coef);
    bin = bin < 0 ? 0 : (bin >= n_bins ? n_bins - 1 : bin);
    return n_bins - 1 - bin;
}

__global__ __launch_bounds__(256) void lsd_hist_kernel(LsdDev d) {
    extern __shared__ unsigned s_h[];  // [4][n_bins]
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.y;
    const int npx = d.w * d.h, nunits = (npx + LSD_UNIT - 1) / LSD_UNIT, unit = blockIdx.x * 4 + wv;
    if (unit >= nunits) return;
    unsigned* h = s_h + wv * d.n_bins;
    for (int t = lane; t < d.n_bins; t += 64) h[t] = 0u;
    const double bin_coef = lsd_bin_coef(d, b);
    const int32_t* k32 = d.k32 + (size_t)b * npx;
#pragma unroll 8
    for (int r = 0; r < LSD_UNIT_ROWS; ++r) {
        const int i = unit * LSD_UNIT + r * 64 + lane;
        const int k = i < npx ? k32[i] : -1;
        if (k >= 0) atomicAdd(&h[lsd_keybin(k, bin_coef, d.n_bins)], 1u);
    }
    unsigned* out = d.cnt + ((size_t)b * nunits + unit) * d.n_bins;
    for (int t = lane; t < d.n_bins; t += 64) out[t] = h[t];
}

__global__ __launch_bounds__(1024) void lsd_scan_kernel(LsdDev d) {
    __shared__ unsigned s_w[16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int npx = d.w * d.h, nunits = (npx + LSD_UNIT - 1) / LSD_UNIT, nb = d.n_bins;
    unsigned* cnt = d.cnt + (size_t)b * nunits * nb;
    unsigned carry = 0u;  // pixels of all the bins in front
    for (int t0 = 0; t0 < nb; t0 += 1024) {  // (n_bins <= 2048: one or two turns)
        const int bin = t0 + t;
        unsigned tot = 0u;
        if (bin < nb)
            for (int u = 0; u < nunits; ++u) tot += cnt[(size_t)u * nb + bin];
        // exclusive scan over the 1024 threads
        unsigned inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned v = __shfl_up(inc, off, 64);
            if (lane >= off) inc += v;
        }
        __syncthreads();  // (s_w of the previous turn has been read)
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        unsigned before = 0u, all = 0u;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned x = s_w[v];
            before += v < wv ? x : 0u;
            all += x;
        }
        unsigned run = carry + before + inc - tot;
        if (bin < nb)
            for (int u = 0; u < nunits; ++u) {
                const unsigned c = cnt[(size_t)u * nb + bin];
                cnt[(size_t)u * nb + bin] = run;
                run += c;
            }
        carry += all;
    }
    // the end mark: every reader walks the ranks in batches of 64 and stops at the first batch whose first rank holds LSD_NOKEY
    if (t < 128 && carry + (unsigned)t < (unsigned)npx) d.order[(size_t)b * npx + carry + t] = LSD_NOKEY;
}

__global__ __launch_bounds__(256) void lsd_scatter_kernel(LsdDev d) {
    extern __shared__ unsigned s_h[];  // [4][n_bins] the next rank of every bin, for this wave's unit
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.y;
    const int npx = d.w * d.h, nunits = (npx + LSD_UNIT - 1) / LSD_UNIT, unit = blockIdx.x * 4 + wv;
    if (unit >= nunits) return;
    unsigned* h = s_h + wv * d.n_bins;
    const unsigned* first = d.cnt + ((size_t)b * nunits + unit) * d.n_bins;
    for (int t = lane; t < d.n_bins; t += 64) h[t] = first[t];
    const double bin_coef = lsd_bin_coef(d, b);
    const int32_t* k32 = d.k32 + (size_t)b * npx;
    uint32_t* order = d.order + (size_t)b * npx;
    const unsigned long long lt = lane == 0 ? 0ull : ~0ull >> (64 - lane);
#pragma unroll 4
    for (int r = 0; r < LSD_UNIT_ROWS; ++r) {
        const int i = unit * LSD_UNIT + r * 64 + lane;
        const int k = i < npx ? k32[i] : -1;
        const bool valid = k >= 0;
        const int kb = valid ? lsd_keybin(k, bin_coef, d.n_bins) : 0;
        unsigned long long m = __ballot(valid);  // ... the lanes of the row with this lane's bin
        if (!m) continue;  // uniform
#pragma unroll
        for (int bit = 0; bit < 11; ++bit) {
            const bool one = (kb >> bit) & 1;
            const unsigned long long bb = __ballot(one);
            m &= one ? bb : ~bb;
        }
        if (valid) {
            const int rank = __builtin_popcountll(m & lt);
            const unsigned pos = h[kb] + (unsigned)rank;
            order[pos] = ((uint32_t)kb << LSD_IDX_BITS) | (uint32_t)i;
            if (rank == 0) h[kb] = pos + (unsigned)__builtin_popcountll(m);  // (the lowest lane of every bin; the read above comes first: one wave, LDS in order)
        }
    }
}

constexpr int LSD_RING = 1024;  // the most recent region points, in LDS (4 KB: the LDS must not limit the images in flight per CU)
constexpr int LSD_GR = 2;       // sub-groups of 7 region points (63 lanes) fetched per round of the region growing

// The flags / the region list are written by lane 0 and read by all lanes of the SAME wave later: workgroup-scope ordering is
// what is needed — the vector L1 is write-through and shared by the CU, so such accesses are ordinary loads / stores with a wait
// for the stores in between.  (Device-scope atomics were the first version: on this part they bypass the XCD's L2, ~2 us each,
// and 512 images took 0.5 s.)
__device__ __forceinline__ int ld_coherent(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_coherent(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float readlane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// stores of this wave are complete / later loads of this wave see them
__device__ __forceinline__ void wave_publish() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// FAST (the default; STVO_LSD_GROW=0 selects the plain form, debug_switches.h): a sub-group's candidates are resolved by guess +
// verification instead of one after the other, region2rect's ordered sums read their terms from LDS instead of through v_readlane,
// the loads of all sub-groups of a round are issued before the first is used.  Same results: the plain form is the direct statement
// of the oracle's loops and stays as the in-library cross-check (tests/test_gpu_lsd.py runs both).  Round 4, one KITTI-size image:
// 233 M cycles plain, 155 M fast (NOTES.md has the table of the steps in between).
template <bool FAST>
__global__ __launch_bounds__(64) void lsd_grow_kernel(LsdDev d) {
    __shared__ int s_ring[LSD_RING];
    __shared__ double s_term[3][64];  // region2rect (FAST): the terms of one chunk, lane l < 3 adds row l in order
    const int b = blockIdx.x, lane = threadIdx.x;
    const int w = d.w, h = d.h, npx = w * h;
    const size_t base = (size_t)b * npx;
    LsdPx* px = d.px + base;
    const int32_t* __restrict__ k32 = d.k32 + base;
    const uint32_t* __restrict__ order = d.order + base;
    int32_t* reg = d.reg + base;
    const double prec = d.prec;
    int n_seg = 0;
    const bool prof = d.dbg != nullptr;
    long long t_grow = 0, t_res = 0, t_rect = 0, n_rounds = 0, n_add = 0, n_regions = 0, n_batches = 0;
    long long t_pub = 0, t_issue = 0, t_wait = 0, t_seed = 0;  // phases of a round (profiling only: explicit waits make them attributable)
    auto tick = [&]() -> long long { return prof ? (long long)__builtin_readcyclecounter() : 0ll; };
    const long long t_begin = tick();
    for (int o0 = 0; o0 < npx; o0 += 64) {
        ++n_batches;
        const uint32_t key = o0 + lane < npx ? order[o0 + lane] : LSD_NOKEY;
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)key) == LSD_NOKEY) break;  // sorted: only undefined pixels from here on
        const int q_l = (int)(key & ((1u << LSD_IDX_BITS) - 1u));
        const bool key_ok = key != LSD_NOKEY;
        const LsdPx seed_l = px_ld(px + (key_ok ? q_l : 0));  // the seeds' angles, fetched with the batch
        const float ang_l = key_ok ? seed_l.ang : -1.f;
        unsigned long long todo = __ballot(key_ok && seed_l.used == 0);
        // seeds of this batch that a region grown from an earlier seed of the batch takes are struck off as they are taken (one
        // compare + ballot per pixel added) — re-reading the flags after every region cost a memory round trip per region
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int seed = __builtin_amdgcn_readlane(q_l, j);
            // ---------------- region_grow ----------------
            const long long tg0 = tick();
            ++n_regions;
            const int sx0 = seed % w, sy0 = seed / w;
            double reg_angle = (double)readlane_f32(ang_l, j) * LSD_DEG2RAD;
            double sn0, cs0;
            sincos_det(reg_angle, sn0, cs0);
            float sumdx = (float)cs0, sumdy = (float)sn0;
            int n_reg = 1;
            if (lane == 0) {
                st_coherent(&px[seed].used, 1);
                st_coherent(reg, sx0 | (sy0 << 16));
                s_ring[0] = sx0 | (sy0 << 16);
            }
            t_seed += tick() - tg0;
            for (int i = 0; i < n_reg;) {
                const long long tp0 = tick();
                wave_publish();  // lane 0's stores of the earlier rounds before the flag / list loads below
                if (prof) __builtin_amdgcn_s_waitcnt(0);
                const long long tp1 = tick();
                t_pub += tp1 - tp0;
                // one round: the next (up to) 7 LSD_GR region points — sub-group r holds points i + 7 r .. i + 7 r + 6, lane = 9 x point
                // + neighbour — fetched together, then resolved sub-group after sub-group, lane after lane: the oracle's order
                const int cnt = n_reg - i < 7 * LSD_GR ? n_reg - i : 7 * LSD_GR;  // uniform
                const int slot0 = lane / 9, nb = lane - slot0 * 9;
                int qq[LSD_GR], xy[LSD_GR];
                float2 cs[LSD_GR];
                double ad[LSD_GR];
                bool cand[LSD_GR];
                if constexpr (FAST) {
                    // every sub-group's region point first, then every load, then the uses: the flag / angle / (cos, sin) loads of
                    // the sub-groups overlap (one after the other they were two memory round trips per round).  Lanes without a
                    // neighbour load pixel 0 and drop the values
                    int pxyv[LSD_GR], u[LSD_GR];
                    float a[LSD_GR];
                    bool val[LSD_GR];
                    const bool in_ring = n_reg - i <= LSD_RING;  // uniform: the round's oldest point is still in the LDS ring
#pragma unroll
                    for (int r = 0; r < LSD_GR; ++r) {
                        const int slot = 7 * r + slot0;
                        val[r] = lane < 63 && slot < cnt && nb != 4;
                        const int at = val[r] ? i + slot : i;
                        pxyv[r] = in_ring ? s_ring[at & (LSD_RING - 1)] : ld_coherent(reg + at);
                    }
#pragma unroll
                    for (int r = 0; r < LSD_GR; ++r) {
                        const int xx = (pxyv[r] & 0xFFFF) + (nb % 3) - 1, yy = (pxyv[r] >> 16) + nb / 3 - 1;
                        val[r] = val[r] && xx >= 0 && xx < w && yy >= 0 && yy < h;
                        qq[r] = val[r] ? yy * w + xx : 0;
                        xy[r] = xx | (yy << 16);
                        const LsdPx t = px_ld(px + qq[r]);
                        u[r] = t.used;
                        a[r] = t.ang;
                        cs[r] = make_float2(t.c, t.s);
                    }
#pragma unroll
                    for (int r = 0; r < LSD_GR; ++r) {
                        cand[r] = val[r] && u[r] == 0 && a[r] >= 0.f;
                        ad[r] = (double)a[r] * LSD_DEG2RAD;
                    }
                } else {
#pragma unroll
                for (int r = 0; r < LSD_GR; ++r) {
                    const int slot = 7 * r + slot0;
                    bool valid = lane < 63 && slot < cnt && nb != 4;  // (the centre is the region point itself)
                    int pxy = 0;
                    if (valid) pxy = (n_reg - (i + slot) <= LSD_RING) ? s_ring[(i + slot) & (LSD_RING - 1)] : ld_coherent(reg + i + slot);
                    const int xx = (pxy & 0xFFFF) + (nb % 3) - 1, yy = (pxy >> 16) + nb / 3 - 1;  // neighbours row by row
                    valid = valid && xx >= 0 && xx < w && yy >= 0 && yy < h;
                    qq[r] = valid ? yy * w + xx : 0;
                    xy[r] = xx | (yy << 16);
                    int u = 1;
                    float a = -1.f;
                    cs[r] = make_float2(0.f, 0.f);
                    if (valid) {
                        const LsdPx t = px_ld(px + qq[r]);
                        u = t.used;
                        a = t.ang;
                        cs[r] = make_float2(t.c, t.s);
                    }
                    cand[r] = valid && u == 0 && a >= 0.f;
                    ad[r] = (double)a * LSD_DEG2RAD;
                }
                }
                const long long ti1 = tick();  // the loads are issued ...
                if (prof) __builtin_amdgcn_s_waitcnt(0);
                const long long tr0 = tick();  // ... and have arrived
                t_issue += ti1 - tp1;
                t_wait += tr0 - ti1;
                ++n_rounds;
                if constexpr (!FAST) {
#pragma unroll
                for (int r = 0; r < LSD_GR; ++r) {
                    if (7 * r >= cnt) break;  // uniform
                    int next = 0;  // lanes below `next` have had their turn
                    for (;;) {
                        double n_theta = reg_angle - ad[r];  // isAligned
                        if (n_theta < 0) n_theta = -n_theta;
                        if (n_theta > LSD_3_2_PI) {
                            n_theta -= LSD_2_PI;
                            if (n_theta < 0) n_theta = -n_theta;
                        }
                        const unsigned long long m = __ballot(cand[r] && lane >= next && n_theta <= prec);
                        if (!m || n_reg >= npx) break;  // (the bound can only bind if a flag were lost: never write past the list)
                        const int L = __builtin_ctzll(m);
                        const int qL = __builtin_amdgcn_readlane(qq[r], L), xyL = __builtin_amdgcn_readlane(xy[r], L);
                        const float cL = readlane_f32(cs[r].x, L), sL = readlane_f32(cs[r].y, L);
                        if (lane == 0) {
                            st_coherent(&px[qL].used, 1);
                            st_coherent(reg + n_reg, xyL);
                            s_ring[n_reg & (LSD_RING - 1)] = xyL;
                        }
                        ++n_reg;
#pragma unroll
                        for (int r2 = 0; r2 < LSD_GR; ++r2) cand[r2] = cand[r2] && qq[r2] != qL;  // the same pixel seen from another point of the round
                        todo &= ~__ballot(key_ok && q_l == qL);
                        sumdx += cL;
                        sumdy += sL;
                        reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * LSD_DEG2RAD;
                        next = L + 1;
                    }
                }
                } else {
                // Guess + verification (tools/experiments/lsd_resolve_model.c has the CPU model and the argument).  Only the two float
                // additions per pixel depend on the order; the ~30 operations of the angle update run for all lanes at once:
                //   A   = the lanes aligned with the angle at the start of the sub-group
                //   A'  = A walked from the lowest lane: a lane taken removes the later lanes that hold the same pixel
                //   lane k: sums_k = start sums + (cos, sin) of A's lanes below k, added in lane order; angle_k = fastAtan2(sums_k), or
                //           the start angle if A' has no lane below k;  dup_k = a lane of A' below k holds the same pixel
                //   D   = the lanes aligned with their angle_k and not dup.  D == A': A' is what the sequential loop accepts
                //         (induction over the lanes); else the lanes below the first difference m are final, lane m's decision is
                //         D[m]: again with A = A' below m | D from m on (2 % of the sub-groups of a KITTI-size scene).
                // Every per-lane flag is kept as a 64-bit lane mask in scalar registers (they are compare results to begin with).
                unsigned long long cand_m[LSD_GR];
#pragma unroll
                for (int r = 0; r < LSD_GR; ++r) cand_m[r] = __ballot(cand[r]);
                const unsigned long long key_m = __ballot(key_ok);
#pragma unroll
                for (int r = 0; r < LSD_GR; ++r) {
                    if (7 * r >= cnt) break;  // uniform
                    unsigned long long A;
                    {
                        double n_theta = reg_angle - ad[r];  // isAligned
                        if (n_theta < 0) n_theta = -n_theta;
                        if (n_theta > LSD_3_2_PI) {
                            n_theta -= LSD_2_PI;
                            if (n_theta < 0) n_theta = -n_theta;
                        }
                        A = __ballot(n_theta <= prec) & cand_m[r];
                    }
                    if (!A || n_reg + 64 > npx) continue;  // (the bound can only bind if a flag were lost: never write past the list)
                    unsigned long long acc, kill, hit_m[LSD_GR];
                    float sx, sy;
                    double th;
                    for (;;) {
                        unsigned long long rem = A, dup_m = 0ull;
                        acc = 0ull;
                        kill = 0ull;
#pragma unroll
                        for (int r2 = 0; r2 < LSD_GR; ++r2) hit_m[r2] = 0ull;
                        sx = sumdx;
                        sy = sumdy;
                        while (rem) {  // uniform: the lanes of the guess, lowest first
                            const int j = __builtin_ctzll(rem);
                            const int qj = __builtin_amdgcn_readlane(qq[r], j);
                            const float cj = readlane_f32(cs[r].x, j), sj = readlane_f32(cs[r].y, j);
                            const unsigned long long later = j == 63 ? 0ull : ~0ull << (j + 1);
                            const unsigned long long same = __ballot(qq[r] == qj);
                            acc |= 1ull << j;
                            rem &= ~same;  // lane j itself and the later lanes that hold the same pixel
                            dup_m |= same & later;
                            const float nx = sx + cj, ny = sy + sj;  // (a select, not a masked addition of zero: x + 0 is not x for x = -0)
                            const bool is_later = lane > j;
                            sx = is_later ? nx : sx;
                            sy = is_later ? ny : sy;
                            kill |= __ballot(q_l == qj);
#pragma unroll
                            for (int r2 = 0; r2 < LSD_GR; ++r2)
                                if (r2 > r) hit_m[r2] |= __ballot(qq[r2] == qj);
                        }
                        const bool any = lane > __builtin_ctzll(acc);
                        th = any ? (double)fast_atan2_deg(sy, sx) * LSD_DEG2RAD : reg_angle;
                        double n_theta = th - ad[r];
                        if (n_theta < 0) n_theta = -n_theta;
                        if (n_theta > LSD_3_2_PI) {
                            n_theta -= LSD_2_PI;
                            if (n_theta < 0) n_theta = -n_theta;
                        }
                        const unsigned long long D = __ballot(n_theta <= prec) & cand_m[r] & ~dup_m;
                        if (D == acc) break;
                        const unsigned long long below = (1ull << __builtin_ctzll(D ^ acc)) - 1ull;
                        A = (acc & below) | (D & ~below);  // (never empty: the lowest lane of a guess sees the start angle, as the guess did)
                    }
                    // acc is the sequential result: every accepted lane stores its own pixel, in lane order
                    if ((acc >> lane) & 1ull) {
                        const int idx = n_reg + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(acc >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)acc, 0u));
                        st_coherent(&px[qq[r]].used, 1);
                        st_coherent(reg + idx, xy[r]);
                        s_ring[idx & (LSD_RING - 1)] = xy[r];
                    }
                    n_reg += __builtin_popcountll(acc);
                    sumdx = readlane_f32(sx, 63);  // lane 63 is never a candidate: its sums are the sums over all of acc
                    sumdy = readlane_f32(sy, 63);
                    reg_angle = readlane_f64(th, 63);
                    todo &= ~(kill & key_m);
#pragma unroll
                    for (int r2 = 0; r2 < LSD_GR; ++r2)
                        if (r2 > r) cand_m[r2] &= ~hit_m[r2];
                }
                }
                t_res += tick() - tr0;
                i += cnt;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // s_ring: lane 0's writes before the next round's reads
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            t_grow += tick() - tg0;
            n_add += n_reg;
            if (n_reg < d.min_reg_size) continue;
            // ---------------- region2rect ----------------
            const long long tq0 = tick();
            wave_publish();
            double X = 0.0, Y = 0.0, S = 0.0;
            // FAST: the three ordered sums of a pass are three independent chains — lane l < 3 adds row l of s_term term by term
            // (one LDS read + one addition per region point instead of six v_readlane + three additions)
            double chain = 0.0;
            auto add_ordered = [&](double t0, double t1, double t2, int cn) {
                s_term[0][lane] = t0;
                s_term[1][lane] = t1;
                s_term[2][lane] = t2;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < 3) {
                    const double* row = s_term[lane];
#pragma unroll 8
                    for (int k = 0; k < cn; ++k) chain += row[k];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the reads before the next chunk's writes
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            };
            for (int c0 = 0; c0 < n_reg; c0 += 64) {
                const int t = c0 + lane, cn = n_reg - c0 < 64 ? n_reg - c0 : 64;
                int pxy = 0;
                double wgt = 0.0;
                if (t < n_reg) {
                    pxy = ld_coherent(reg + t);
                    wgt = sqrt((double)k32[(pxy >> 16) * w + (pxy & 0xFFFF)] / 4.0);  // the gradient norm (ll_angle's expression)
                }
                const double px = (double)(pxy & 0xFFFF) * wgt, py = (double)(pxy >> 16) * wgt;
                if constexpr (FAST) {
                    add_ordered(px, py, wgt, cn);
                } else {
                    for (int k = 0; k < cn; ++k) {  // strictly in region order: every addition rounds as the oracle's
                        X += readlane_f64(px, k);
                        Y += readlane_f64(py, k);
                        S += readlane_f64(wgt, k);
                    }
                }
            }
            if constexpr (FAST) {
                X = readlane_f64(chain, 0);
                Y = readlane_f64(chain, 1);
                S = readlane_f64(chain, 2);
                chain = 0.0;
            }
            const double cx = X / S, cy = Y / S;
            double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
            for (int c0 = 0; c0 < n_reg; c0 += 64) {
                const int t = c0 + lane, cn = n_reg - c0 < 64 ? n_reg - c0 : 64;
                double t_xx = 0.0, t_yy = 0.0, t_xy = 0.0;
                if (t < n_reg) {
                    const int pxy = ld_coherent(reg + t);
                    const double wgt = sqrt((double)k32[(pxy >> 16) * w + (pxy & 0xFFFF)] / 4.0);  // the gradient norm (ll_angle's expression)
                    const double ddx = (double)(pxy & 0xFFFF) - cx, ddy = (double)(pxy >> 16) - cy;
                    t_xx = ddy * ddy * wgt;
                    t_yy = ddx * ddx * wgt;
                    t_xy = ddx * ddy * wgt;
                }
                if constexpr (FAST) {
                    add_ordered(t_xx, t_yy, -t_xy, cn);  // (a - b and a + (-b) round alike)
                } else {
                    for (int k = 0; k < cn; ++k) {
                        Ixx += readlane_f64(t_xx, k);
                        Iyy += readlane_f64(t_yy, k);
                        Ixy -= readlane_f64(t_xy, k);
                    }
                }
            }
            if constexpr (FAST) {
                Ixx = readlane_f64(chain, 0);
                Iyy = readlane_f64(chain, 1);
                Ixy = readlane_f64(chain, 2);
            }
            const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
            double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)fast_atan2_deg((float)(lambda - Ixx), (float)Ixy)
                                                   : (double)fast_atan2_deg((float)Ixy, (float)(lambda - Iyy));
            theta *= LSD_DEG2RAD;
            {
                double diff = theta - reg_angle;  // angle_diff
                while (diff <= -LSD_PI) diff += LSD_2_PI;
                while (diff > LSD_PI) diff -= LSD_2_PI;
                if (fabs(diff) > prec) theta += LSD_PI;
            }
            double dx, dy;
            sincos_det(theta, dy, dx);
            double l_min = 0.0, l_max = 0.0;  // (the width of the rectangle is not part of the segment)
            for (int t = lane; t < n_reg; t += 64) {
                const int pxy = ld_coherent(reg + t);
                const double rdx = (double)(pxy & 0xFFFF) - cx, rdy = (double)(pxy >> 16) - cy;
                const double l = rdx * dx + rdy * dy;
                l_max = l > l_max ? l : l_max;
                l_min = l < l_min ? l : l_min;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {  // maxima / minima: any order
                const double om = __shfl_xor(l_max, off, 64), on = __shfl_xor(l_min, off, 64);
                l_max = om > l_max ? om : l_max;
                l_min = on < l_min ? on : l_min;
            }
            double x1 = cx + l_min * dx, y1 = cy + l_min * dy, x2 = cx + l_max * dx, y2 = cy + l_max * dy;
            x1 += 0.5; y1 += 0.5; x2 += 0.5; y2 += 0.5;
            if (d.scale != 1) {
                x1 /= d.scale; y1 /= d.scale; x2 /= d.scale; y2 /= d.scale;
            }
            if (lane == 0 && n_seg < d.seg_cap) d.seg[(size_t)b * d.seg_cap + n_seg] = make_float4((float)x1, (float)y1, (float)x2, (float)y2);
            if (d.dbg && lane == 0 && n_seg < d.seg_cap) {
                double* q = d.dbg + ((size_t)b * d.seg_cap + n_seg) * 8;
                q[0] = cx; q[1] = cy; q[2] = Ixx; q[3] = Iyy; q[4] = Ixy; q[5] = theta; q[6] = l_min; q[7] = l_max;
            }
            ++n_seg;
            t_rect += tick() - tq0;
        }
    }
    if (lane == 0) d.n_seg[b] = n_seg;
    if (prof && lane == 0 && b == 0) {  // tools/lsd_probe.py: the last 16 doubles of image 0's block
        double* q = d.dbg + ((size_t)d.seg_cap - 2) * 8;
        q[0] = (double)(tick() - t_begin); q[1] = (double)t_grow; q[2] = (double)t_res; q[3] = (double)t_rect; q[4] = (double)n_rounds;
        q[5] = (double)n_add; q[6] = (double)n_regions; q[7] = (double)n_batches;
        q[8] = (double)t_pub; q[9] = (double)t_issue; q[10] = (double)t_wait; q[11] = (double)t_seed;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// lsd_refine = 1 (LSD_REFINE_STD; oracle/stvo_lsd_oracle.c: refine_region / reduce_region_radius): a region too sparse for its rectangle
// is given back (its pixels are FREE again), grown once more from the same seed under a tolerance of twice the standard deviation of
// the angles near the seed and, if still too sparse, cut back to 75 % of its radius until it is dense enough or gone.  Flags that turn
// OFF break what the other forms of the search rely on (a batch's seeds struck off for good, speculation against flags that only ever
// turn on), so this mode has its own kernel: one wave per image, the direct statement of the oracle's loops (the plain form above)
// with the growth and the rectangle as routines, for every batch size.  No shipped configuration uses the mode (config/*.yaml:
// lsd_refine : 0); it is there so that a caller who sets it gets the detector it asked for rather than an error.
struct LsdRect {
    double x1, y1, x2, y2, width;
};
__global__ __launch_bounds__(64) void lsd_grow_refine_kernel(LsdDev d, const double density_th) {
    __shared__ int s_ring[LSD_RING];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int w = d.w, h = d.h, npx = w * h;
    const size_t base = (size_t)b * npx;
    LsdPx* px = d.px + base;
    const int32_t* __restrict__ k32 = d.k32 + base;
    const uint32_t* __restrict__ order = d.order + base;
    int32_t* reg = d.reg + base;
    const double prec = d.prec;
    int n_seg = 0;
    int q_l = 0;
    bool key_ok = false;
    unsigned long long todo = 0ull;

    // region_grow from `seed` (angle ang_deg) under the tolerance tol: the list in reg (and the ring), the size returned
    auto grow = [&](const int seed, const float ang_deg, const double tol, double& reg_angle) -> int {
        const int sx0 = seed % w, sy0 = seed / w;
        reg_angle = (double)ang_deg * LSD_DEG2RAD;
        double sn0, cs0;
        sincos_det(reg_angle, sn0, cs0);
        float sumdx = (float)cs0, sumdy = (float)sn0;
        int n_reg = 1;
        if (lane == 0) {
            st_coherent(&px[seed].used, 1);
            st_coherent(reg, sx0 | (sy0 << 16));
            s_ring[0] = sx0 | (sy0 << 16);
        }
        for (int i = 0; i < n_reg;) {
            wave_publish();
            const int cnt = n_reg - i < 7 * LSD_GR ? n_reg - i : 7 * LSD_GR;  // uniform
            const int slot0 = lane / 9, nb = lane - slot0 * 9;
            int qq[LSD_GR], xy[LSD_GR];
            float2 cs[LSD_GR];
            double ad[LSD_GR];
            bool cand[LSD_GR];
#pragma unroll
            for (int r = 0; r < LSD_GR; ++r) {
                const int slot = 7 * r + slot0;
                bool valid = lane < 63 && slot < cnt && nb != 4;  // (the centre is the region point itself)
                int pxy = 0;
                if (valid) pxy = (n_reg - (i + slot) <= LSD_RING) ? s_ring[(i + slot) & (LSD_RING - 1)] : ld_coherent(reg + i + slot);
                const int xx = (pxy & 0xFFFF) + (nb % 3) - 1, yy = (pxy >> 16) + nb / 3 - 1;  // neighbours row by row
                valid = valid && xx >= 0 && xx < w && yy >= 0 && yy < h;
                qq[r] = valid ? yy * w + xx : 0;
                xy[r] = xx | (yy << 16);
                int u = 1;
                float a = -1.f;
                cs[r] = make_float2(0.f, 0.f);
                if (valid) {
                    const LsdPx t = px_ld(px + qq[r]);
                    u = t.used;
                    a = t.ang;
                    cs[r] = make_float2(t.c, t.s);
                }
                cand[r] = valid && u == 0 && a >= 0.f;
                ad[r] = (double)a * LSD_DEG2RAD;
            }
#pragma unroll
            for (int r = 0; r < LSD_GR; ++r) {
                if (7 * r >= cnt) break;  // uniform
                int next = 0;  // lanes below `next` have had their turn
                for (;;) {
                    double n_theta = reg_angle - ad[r];  // isAligned
                    if (n_theta < 0) n_theta = -n_theta;
                    if (n_theta > LSD_3_2_PI) {
                        n_theta -= LSD_2_PI;
                        if (n_theta < 0) n_theta = -n_theta;
                    }
                    const unsigned long long m = __ballot(cand[r] && lane >= next && n_theta <= tol);
                    if (!m || n_reg >= npx) break;
                    const int L = __builtin_ctzll(m);
                    const int qL = __builtin_amdgcn_readlane(qq[r], L), xyL = __builtin_amdgcn_readlane(xy[r], L);
                    const float cL = readlane_f32(cs[r].x, L), sL = readlane_f32(cs[r].y, L);
                    if (lane == 0) {
                        st_coherent(&px[qL].used, 1);
                        st_coherent(reg + n_reg, xyL);
                        s_ring[n_reg & (LSD_RING - 1)] = xyL;
                    }
                    ++n_reg;
#pragma unroll
                    for (int r2 = 0; r2 < LSD_GR; ++r2) cand[r2] = cand[r2] && qq[r2] != qL;  // the same pixel seen from another point of the round
                    todo &= ~__ballot(key_ok && q_l == qL);
                    sumdx += cL;
                    sumdy += sL;
                    reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * LSD_DEG2RAD;
                    next = L + 1;
                }
            }
            i += cnt;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // s_ring: lane 0's writes before the next round's reads
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        return n_reg;
    };

    // region2rect of reg[0 .. n_reg): every sum strictly in region order
    auto rect = [&](const int n_reg, const double reg_angle) -> LsdRect {
        wave_publish();
        double X = 0.0, Y = 0.0, S = 0.0;
        for (int c0 = 0; c0 < n_reg; c0 += 64) {
            const int t = c0 + lane, cn = n_reg - c0 < 64 ? n_reg - c0 : 64;
            int pxy = 0;
            double wgt = 0.0;
            if (t < n_reg) {
                pxy = ld_coherent(reg + t);
                wgt = sqrt((double)k32[(pxy >> 16) * w + (pxy & 0xFFFF)] / 4.0);
            }
            const double pxw = (double)(pxy & 0xFFFF) * wgt, pyw = (double)(pxy >> 16) * wgt;
            for (int k = 0; k < cn; ++k) {
                X += readlane_f64(pxw, k);
                Y += readlane_f64(pyw, k);
                S += readlane_f64(wgt, k);
            }
        }
        const double cx = X / S, cy = Y / S;
        double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
        for (int c0 = 0; c0 < n_reg; c0 += 64) {
            const int t = c0 + lane, cn = n_reg - c0 < 64 ? n_reg - c0 : 64;
            double t_xx = 0.0, t_yy = 0.0, t_xy = 0.0;
            if (t < n_reg) {
                const int pxy = ld_coherent(reg + t);
                const double wgt = sqrt((double)k32[(pxy >> 16) * w + (pxy & 0xFFFF)] / 4.0);
                const double ddx = (double)(pxy & 0xFFFF) - cx, ddy = (double)(pxy >> 16) - cy;
                t_xx = ddy * ddy * wgt;
                t_yy = ddx * ddx * wgt;
                t_xy = ddx * ddy * wgt;
            }
            for (int k = 0; k < cn; ++k) {
                Ixx += readlane_f64(t_xx, k);
                Iyy += readlane_f64(t_yy, k);
                Ixy -= readlane_f64(t_xy, k);
            }
        }
        const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
        double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)fast_atan2_deg((float)(lambda - Ixx), (float)Ixy)
                                               : (double)fast_atan2_deg((float)Ixy, (float)(lambda - Iyy));
        theta *= LSD_DEG2RAD;
        {
            double diff = theta - reg_angle;  // angle_diff
            while (diff <= -LSD_PI) diff += LSD_2_PI;
            while (diff > LSD_PI) diff -= LSD_2_PI;
            if (fabs(diff) > prec) theta += LSD_PI;
        }
        double dx, dy;
        sincos_det(theta, dy, dx);
        double l_min = 0.0, l_max = 0.0, w_min = 0.0, w_max = 0.0;
        for (int t = lane; t < n_reg; t += 64) {
            const int pxy = ld_coherent(reg + t);
            const double rdx = (double)(pxy & 0xFFFF) - cx, rdy = (double)(pxy >> 16) - cy;
            const double l = rdx * dx + rdy * dy, ww = -rdx * dy + rdy * dx;
            l_max = l > l_max ? l : l_max;
            l_min = l < l_min ? l : l_min;
            w_max = ww > w_max ? ww : w_max;
            w_min = ww < w_min ? ww : w_min;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {  // maxima / minima: any order
            const double a0 = __shfl_xor(l_max, off, 64), a1 = __shfl_xor(l_min, off, 64), a2 = __shfl_xor(w_max, off, 64), a3 = __shfl_xor(w_min, off, 64);
            l_max = a0 > l_max ? a0 : l_max;
            l_min = a1 < l_min ? a1 : l_min;
            w_max = a2 > w_max ? a2 : w_max;
            w_min = a3 < w_min ? a3 : w_min;
        }
        LsdRect rc;
        rc.x1 = cx + l_min * dx; rc.y1 = cy + l_min * dy; rc.x2 = cx + l_max * dx; rc.y2 = cy + l_max * dy;
        rc.width = w_max - w_min;
        if (rc.width < 1.0) rc.width = 1.0;
        return rc;
    };
    auto dist_sq = [](double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); };
    auto density_of = [&](const LsdRect& rc, int n_reg) { return (double)n_reg / (sqrt(dist_sq(rc.x1, rc.y1, rc.x2, rc.y2)) * rc.width); };

    for (int o0 = 0; o0 < npx; o0 += 64) {
        const uint32_t key = o0 + lane < npx ? order[o0 + lane] : LSD_NOKEY;
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)key) == LSD_NOKEY) break;  // sorted: only undefined pixels from here on
        q_l = (int)(key & ((1u << LSD_IDX_BITS) - 1u));
        key_ok = key != LSD_NOKEY;
        wave_publish();
        const LsdPx seed_l = px_ld(px + (key_ok ? q_l : 0));
        const float ang_l = key_ok ? seed_l.ang : -1.f;
        todo = __ballot(key_ok && seed_l.used == 0);
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int seed = __builtin_amdgcn_readlane(q_l, j);
            const float seed_ang = readlane_f32(ang_l, j);
            double reg_angle;
            int n_reg = grow(seed, seed_ang, prec, reg_angle);
            if (n_reg < d.min_reg_size) continue;
            LsdRect rc = rect(n_reg, reg_angle);
            double density = density_of(rc, n_reg);
            if (!(density >= density_th)) {
                // ---- refine: the angles near the seed (within the rectangle's width), every pixel of the region free again ----
                const int xy0 = ld_coherent(reg);
                const double xc = (double)(xy0 & 0xFFFF), yc = (double)(xy0 >> 16);
                const double ang_c = (double)seed_ang * LSD_DEG2RAD;
                double sum = 0.0, s_sum = 0.0;
                int n = 0;
                for (int c0 = 0; c0 < n_reg; c0 += 64) {
                    const int t = c0 + lane, cn = n_reg - c0 < 64 ? n_reg - c0 : 64;
                    bool in = false;
                    double ang_d = 0.0;
                    if (t < n_reg) {
                        const int pxy = ld_coherent(reg + t);
                        const int q = (pxy >> 16) * w + (pxy & 0xFFFF);
                        const LsdPx tq = px_ld(px + q);
                        st_coherent(&px[q].used, 0);
                        in = sqrt(dist_sq(xc, yc, (double)(pxy & 0xFFFF), (double)(pxy >> 16))) < rc.width;
                        double diff = (double)tq.ang * LSD_DEG2RAD - ang_c;  // angle_diff_signed
                        while (diff <= -LSD_PI) diff += LSD_2_PI;
                        while (diff > LSD_PI) diff -= LSD_2_PI;
                        ang_d = diff;
                    }
                    const double sq = ang_d * ang_d;
                    const unsigned long long in_m = __ballot(in);
                    for (int k = 0; k < cn; ++k)
                        if ((in_m >> k) & 1ull) {  // uniform
                            sum += readlane_f64(ang_d, k);
                            s_sum += readlane_f64(sq, k);
                            ++n;
                        }
                }
                const double mean_angle = sum / (double)n;
                const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
                n_reg = grow(seed, seed_ang, tau, reg_angle);
                bool gone = n_reg < 2;
                if (!gone) {
                    rc = rect(n_reg, reg_angle);
                    density = density_of(rc, n_reg);
                    if (density < density_th) {
                        // ---- reduce_region_radius: points farther than the (shrinking) radius from the seed leave, the LAST point of the
                        // list takes a leaving point's place (and is looked at next) — one point after the other, as the oracle does
                        const double r1 = dist_sq(xc, yc, rc.x1, rc.y1), r2 = dist_sq(xc, yc, rc.x2, rc.y2);
                        double rad_sq = r1 > r2 ? r1 : r2;
                        while (density < density_th) {
                            rad_sq *= 0.75 * 0.75;
                            wave_publish();
                            for (int c0 = 0; c0 < n_reg; c0 += 64) {
                                const int t = c0 + lane;
                                const int pxy = t < n_reg ? ld_coherent(reg + t) : 0;
                                const bool far = t < n_reg && dist_sq(xc, yc, (double)(pxy & 0xFFFF), (double)(pxy >> 16)) > rad_sq;
                                unsigned long long far_m = __ballot(far);
                                while (far_m) {
                                    const int k = __builtin_ctzll(far_m);
                                    far_m &= far_m - 1ull;
                                    const int i = c0 + k;
                                    if (i >= n_reg) break;  // (the list has shrunk below this position: its point left as a LAST point)
                                    int out_xy = __builtin_amdgcn_readlane(pxy, k);
                                    for (;;) {  // the point at position i leaves; the last point moves in and is tested in turn
                                        if (lane == 0) st_coherent(&px[(out_xy >> 16) * w + (out_xy & 0xFFFF)].used, 0);
                                        --n_reg;
                                        if (i >= n_reg) break;  // (it was the last point itself)
                                        const int last_xy = __builtin_amdgcn_readfirstlane(ld_coherent(reg + n_reg));
                                        if (!(dist_sq(xc, yc, (double)(last_xy & 0xFFFF), (double)(last_xy >> 16)) > rad_sq)) {
                                            if (lane == 0) st_coherent(reg + i, last_xy);
                                            break;
                                        }
                                        out_xy = last_xy;  // the point that moved in is too far as well: it leaves from position i
                                    }
                                }
                                wave_publish();  // (a later chunk may hold positions this one has just filled)
                            }
                            if (n_reg < 2) {
                                gone = true;
                                break;
                            }
                            rc = rect(n_reg, reg_angle);
                            density = density_of(rc, n_reg);
                        }
                    }
                }
                // flags have been turned off: the batch's later seeds as they stand now
                wave_publish();
                const int u_now = key_ok ? ld_coherent(&px[q_l].used) : 1;
                todo = __ballot(key_ok && lane > j && u_now == 0);
                if (gone) continue;
            }
            double x1 = rc.x1 + 0.5, y1 = rc.y1 + 0.5, x2 = rc.x2 + 0.5, y2 = rc.y2 + 0.5;
            if (d.scale != 1) {
                x1 /= d.scale; y1 /= d.scale; x2 /= d.scale; y2 /= d.scale;
            }
            if (lane == 0 && n_seg < d.seg_cap) d.seg[(size_t)b * d.seg_cap + n_seg] = make_float4((float)x1, (float)y1, (float)x2, (float)y2);
            ++n_seg;
        }
    }
    if (lane == 0) d.n_seg[b] = n_seg;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ONE image by many waves — an EXACT asynchronous form of the search, the default for batches of <= LSD_WAVES_MAX_B images
// (STVO_LSD_WAVES=0: one wave per image there too).  The scheme is replayed on the CPU by tools/experiments/lsd_waves_sim.c.
//   the COMMITTER walks the seeds in order.  A seed with a finished PENDING region takes it if every pixel of it is still free (the
//   segment was computed by the wave that grew it), a seed another wave is growing right now is waited for (bounded), any other
//   seed — and any pending region that lost a pixel — is grown by the committer itself, flags set as it goes.
//   the SPECULATING waves grow regions of free seeds in front of the committer against the flags committed so far; a wave marks ITS
//   pixels in a stamp array of its own and leaves the flags alone.  Seeds are handed out in rank order, >= LSD_SEP px (Chebyshev)
//   from every seed in flight, at most LSD_AHEAD ranks ahead of the committer.
// Exactness: flags only turn on.  A region grown against an older state of the flags made the sequential decisions at every pixel it
// examined unless it ACCEPTED a pixel that was taken before its turn — then the validation at its turn fails and the seed is grown again.
// Output order = commit order = seed order.
// Round 5 ran this inside ONE workgroup of sixteen waves (22.8 ms per KITTI-size image: the waves share a CU's four SIMDs and a
// growth round is ~1500 cycles of dependent instructions — fifteen speculators delivered ~2.7 waves' worth of growth); round 6
// spreads it over the CUs of one XCD: lsd_grow_xcd_kernel below (9.9 ms; the oracle takes 17 ms on one host core).
constexpr int LSD_WRING = 256;       // per wave: the most recent region points in LDS
constexpr int LSD_SEP = 8;            // (4 … 16 give the same ~130 failed validations per KITTI-size image, 0: 390; 24 … 32: more seeds reach the committer unclaimed)
constexpr int LSD_LOOK = 2048;       // ranks the dispatcher examines per pass, from the front
constexpr int LSD_AHEAD = 16384;     // the front's lead over the committer (ranks; ~200 seeds of a KITTI-size scene)
// Table entry of a seed: 0 = nobody's; LSD_CLAIM | wave = a speculating wave is growing it; else a finished region (lsd_entry_x)
constexpr long long LSD_CLAIM = (long long)0x8000000000000000ull;
__device__ __forceinline__ long long ld_coherent64(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_coherent64(long long* p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ long long readlane_i64(long long v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), l);
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long readfirstlane64(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
constexpr int LSD_WAVES_MAX_B = 128; // up to 16 images per XCD, each with a committer's workgroup + at least one speculating workgroup (~40 B of scratch per pixel
                                     // and speculating workgroup); beyond that one wave per image and the batch as the parallelism

// loads that bypass the vector L1 (sc1: served by the XCD's L2) — data another CU of the same XCD stores during the launch
__device__ __forceinline__ int ld_l2(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long ld_l2_64(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int lds_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// region_grow from `seed`, the fast form of lsd_grow_kernel.  MARK (the committer): a pixel is taken by setting its flag.  !MARK (a
// speculating wave): the flags are only read; the wave's own pixels carry `id` in `stamp`.  The list goes to `list` (at most `cap`
// entries: -1 if it does not fit), the final region angle to `angle_out`.
// BITS (with MARK): the flags this wave decides on are a bitmap in LDS (`bits`); `used` is only written — for the waves of other CUs.
__device__ __forceinline__ unsigned bit_ld(const unsigned* bits, int q) { return (__hip_atomic_load(bits + (q >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (q & 31)) & 1u; }
__device__ __forceinline__ void bit_set(unsigned* bits, int q) { (void)__hip_atomic_fetch_or(bits + (q >> 5), 1u << (q & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <bool MARK, bool FAR = false, bool BITS = false>
__device__ __forceinline__ int grow_region_w(LsdPx* px, int32_t* stamp, int id,
                                             int32_t* list, int cap, int* ring, int seed, float seed_ang, int w, int h, double prec,
                                             double& angle_out, unsigned* bits = nullptr) {
    const int lane = threadIdx.x & 63;
    const int sx0 = seed % w, sy0 = seed / w;
    double reg_angle = (double)seed_ang * LSD_DEG2RAD;
    double sn0, cs0;
    sincos_det(reg_angle, sn0, cs0);
    float sumdx = (float)cs0, sumdy = (float)sn0;
    int n_reg = 1;
    if (lane == 0) {
        if (MARK) st_coherent(&px[seed].used, 1);
        else st_coherent(stamp + seed, id);
        if (BITS) bit_set(bits, seed);
        st_coherent(list, sx0 | (sy0 << 16));
        ring[0] = sx0 | (sy0 << 16);
    }
    for (int i = 0; i < n_reg;) {
        const int cnt = n_reg - i < 7 * LSD_GR ? n_reg - i : 7 * LSD_GR;  // uniform
        const int slot0 = lane / 9, nb = lane - slot0 * 9;
        int qq[LSD_GR], xy[LSD_GR], pxyv[LSD_GR], u[LSD_GR], own[LSD_GR];
        float2 cs[LSD_GR];
        float a[LSD_GR];
        double ad[LSD_GR];
        bool val[LSD_GR];
        wave_publish();
        const bool in_ring = n_reg - i <= LSD_WRING;  // uniform
#pragma unroll
        for (int r = 0; r < LSD_GR; ++r) {
            const int slot = 7 * r + slot0;
            val[r] = lane < 63 && slot < cnt && nb != 4;
            const int at = val[r] ? i + slot : i;
            pxyv[r] = in_ring ? ring[at & (LSD_WRING - 1)] : ld_coherent(list + at);
        }
#pragma unroll
        for (int r = 0; r < LSD_GR; ++r) {
            const int xx = (pxyv[r] & 0xFFFF) + (nb % 3) - 1, yy = (pxyv[r] >> 16) + nb / 3 - 1;
            val[r] = val[r] && xx >= 0 && xx < w && yy >= 0 && yy < h;
            qq[r] = val[r] ? yy * w + xx : 0;
            xy[r] = xx | (yy << 16);
            own[r] = MARK ? 0 : ld_coherent(stamp + qq[r]);
        }
        {   // the records: one 16-byte gather per sub-group.  FAR: the flag in it is stored by ANOTHER CU — the load bypasses the vector L1
            LsdPx t[LSD_GR];
            if constexpr (FAR) {
                static_assert(LSD_GR == 2, "two loads in one statement");
                lsd_f4 v0, v1;
                asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(v0), "=&v"(v1)
                             : "v"(px + qq[0]), "v"(px + qq[1])
                             : "memory");
                t[0].ang = v0.x; t[0].c = v0.y; t[0].s = v0.z; t[0].used = __float_as_int(v0.w);
                t[1].ang = v1.x; t[1].c = v1.y; t[1].s = v1.z; t[1].used = __float_as_int(v1.w);
            } else {
#pragma unroll
                for (int r = 0; r < LSD_GR; ++r) t[r] = px_ld(px + qq[r]);
            }
#pragma unroll
            for (int r = 0; r < LSD_GR; ++r) {
                u[r] = BITS ? (int)bit_ld(bits, qq[r]) : t[r].used;
                a[r] = t[r].ang;
                cs[r] = make_float2(t[r].c, t[r].s);
            }
        }
        unsigned long long cand_m[LSD_GR];
#pragma unroll
        for (int r = 0; r < LSD_GR; ++r) {
            cand_m[r] = __ballot(val[r] && u[r] == 0 && (MARK || own[r] != id) && a[r] >= 0.f);
            ad[r] = (double)a[r] * LSD_DEG2RAD;
        }
        // guess + verification of a sub-group: lsd_grow_kernel<true> has the explanation
#pragma unroll
        for (int r = 0; r < LSD_GR; ++r) {
            if (7 * r >= cnt) break;  // uniform
            unsigned long long A;
            {
                double n_theta = reg_angle - ad[r];
                if (n_theta < 0) n_theta = -n_theta;
                if (n_theta > LSD_3_2_PI) {
                    n_theta -= LSD_2_PI;
                    if (n_theta < 0) n_theta = -n_theta;
                }
                A = __ballot(n_theta <= prec) & cand_m[r];
            }
            if (!A) continue;
            if (n_reg + 64 > cap) return -1;  // uniform
            unsigned long long acc, hit_m[LSD_GR];
            float sx, sy;
            double th;
            for (;;) {
                unsigned long long rem = A, dup_m = 0ull;
                acc = 0ull;
#pragma unroll
                for (int r2 = 0; r2 < LSD_GR; ++r2) hit_m[r2] = 0ull;
                sx = sumdx;
                sy = sumdy;
                while (rem) {
                    const int j = __builtin_ctzll(rem);
                    const int qj = __builtin_amdgcn_readlane(qq[r], j);
                    const float cj = readlane_f32(cs[r].x, j), sj = readlane_f32(cs[r].y, j);
                    const unsigned long long later = j == 63 ? 0ull : ~0ull << (j + 1);
                    const unsigned long long same = __ballot(qq[r] == qj);
                    acc |= 1ull << j;
                    rem &= ~same;
                    dup_m |= same & later;
                    const float nx = sx + cj, ny = sy + sj;
                    const bool is_later = lane > j;
                    sx = is_later ? nx : sx;
                    sy = is_later ? ny : sy;
#pragma unroll
                    for (int r2 = 0; r2 < LSD_GR; ++r2)
                        if (r2 > r) hit_m[r2] |= __ballot(qq[r2] == qj);
                }
                const bool any = lane > __builtin_ctzll(acc);
                th = any ? (double)fast_atan2_deg(sy, sx) * LSD_DEG2RAD : reg_angle;
                double n_theta = th - ad[r];
                if (n_theta < 0) n_theta = -n_theta;
                if (n_theta > LSD_3_2_PI) {
                    n_theta -= LSD_2_PI;
                    if (n_theta < 0) n_theta = -n_theta;
                }
                const unsigned long long D = __ballot(n_theta <= prec) & cand_m[r] & ~dup_m;
                if (D == acc) break;
                const unsigned long long below = (1ull << __builtin_ctzll(D ^ acc)) - 1ull;
                A = (acc & below) | (D & ~below);
            }
            if ((acc >> lane) & 1ull) {
                const int idx = n_reg + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(acc >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)acc, 0u));
                if (MARK) st_coherent(&px[qq[r]].used, 1);
                else st_coherent(stamp + qq[r], id);
                if (BITS) bit_set(bits, qq[r]);
                st_coherent(list + idx, xy[r]);
                ring[idx & (LSD_WRING - 1)] = xy[r];
            }
            n_reg += __builtin_popcountll(acc);
            sumdx = readlane_f32(sx, 63);
            sumdy = readlane_f32(sy, 63);
            reg_angle = readlane_f64(th, 63);
#pragma unroll
            for (int r2 = 0; r2 < LSD_GR; ++r2)
                if (r2 > r) cand_m[r2] &= ~hit_m[r2];
        }
        i += cnt;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the ring: this round's writes before the next round's reads
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    angle_out = reg_angle;
    return n_reg;
}

// region2rect + the segment of a region of n >= min_reg_size pixels (the fast form of lsd_grow_kernel); term: [3][64] doubles of LDS
__device__ __forceinline__ float4 region_segment_w(const LsdDev& d, const int32_t* list, int n_reg, const int32_t* __restrict__ k32, double reg_angle,
                                                   double (*term)[64]) {
    const int lane = threadIdx.x & 63, w = d.w;
    wave_publish();
    double chain = 0.0;
    auto add_ordered = [&](double t0, double t1, double t2, int cn) {
        term[0][lane] = t0;
        term[1][lane] = t1;
        term[2][lane] = t2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 3) {
            const double* row = term[lane];
#pragma unroll 8
            for (int k = 0; k < cn; ++k) chain += row[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int c0 = 0; c0 < n_reg; c0 += 64) {
        const int t = c0 + lane, cn = n_reg - c0 < 64 ? n_reg - c0 : 64;
        int pxy = 0;
        double wgt = 0.0;
        if (t < n_reg) {
            pxy = ld_coherent(list + t);
            wgt = sqrt((double)k32[(pxy >> 16) * w + (pxy & 0xFFFF)] / 4.0);  // the gradient norm (ll_angle's expression)
        }
        add_ordered((double)(pxy & 0xFFFF) * wgt, (double)(pxy >> 16) * wgt, wgt, cn);
    }
    const double X = readlane_f64(chain, 0), Y = readlane_f64(chain, 1), S = readlane_f64(chain, 2);
    chain = 0.0;
    const double cx = X / S, cy = Y / S;
    for (int c0 = 0; c0 < n_reg; c0 += 64) {
        const int t = c0 + lane, cn = n_reg - c0 < 64 ? n_reg - c0 : 64;
        double t_xx = 0.0, t_yy = 0.0, t_xy = 0.0;
        if (t < n_reg) {
            const int pxy = ld_coherent(list + t);
            const double wgt = sqrt((double)k32[(pxy >> 16) * w + (pxy & 0xFFFF)] / 4.0);  // the gradient norm (ll_angle's expression)
            const double ddx = (double)(pxy & 0xFFFF) - cx, ddy = (double)(pxy >> 16) - cy;
            t_xx = ddy * ddy * wgt;
            t_yy = ddx * ddx * wgt;
            t_xy = ddx * ddy * wgt;
        }
        add_ordered(t_xx, t_yy, -t_xy, cn);
    }
    const double Ixx = readlane_f64(chain, 0), Iyy = readlane_f64(chain, 1), Ixy = readlane_f64(chain, 2);
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)fast_atan2_deg((float)(lambda - Ixx), (float)Ixy)
                                           : (double)fast_atan2_deg((float)Ixy, (float)(lambda - Iyy));
    theta *= LSD_DEG2RAD;
    {
        double diff = theta - reg_angle;
        while (diff <= -LSD_PI) diff += LSD_2_PI;
        while (diff > LSD_PI) diff -= LSD_2_PI;
        if (fabs(diff) > d.prec) theta += LSD_PI;
    }
    double dx, dy;
    sincos_det(theta, dy, dx);
    double l_min = 0.0, l_max = 0.0;
    for (int t = lane; t < n_reg; t += 64) {
        const int pxy = ld_coherent(list + t);
        const double rdx = (double)(pxy & 0xFFFF) - cx, rdy = (double)(pxy >> 16) - cy;
        const double l = rdx * dx + rdy * dy;
        l_max = l > l_max ? l : l_max;
        l_min = l < l_min ? l : l_min;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double om = __shfl_xor(l_max, off, 64), on = __shfl_xor(l_min, off, 64);
        l_max = om > l_max ? om : l_max;
        l_min = on < l_min ? on : l_min;
    }
    double x1 = cx + l_min * dx, y1 = cy + l_min * dy, x2 = cx + l_max * dx, y2 = cy + l_max * dy;
    x1 += 0.5; y1 += 0.5; x2 += 0.5; y2 += 0.5;
    if (d.scale != 1) {
        x1 /= d.scale; y1 /= d.scale; x2 /= d.scale; y2 /= d.scale;
    }
    return make_float4((float)x1, (float)y1, (float)x2, (float)y2);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The protocol above across the CUs of ONE XCD.  A workgroup is four waves, one per SIMD (84 KB of LDS at least: a CU holds one).
// Image b is served by the workgroups b, b + 8, b + 16, ... — the dispatcher of the hardware places those on one XCD:
//   workgroup b       wave 0 the COMMITTER.  One wave on its SIMD pays its issue latency for every instruction, so it executes as
//                            little as possible per region: its flags are a bitmap in LDS (the global flags are written behind, for the
//                            speculating waves), the finished regions reach it through LDS records (the feeder);
//                     wave 1 the DISPATCHER: scans the ranks in front of the committer and hands the free seeds to idle speculating
//                            waves through their mail boxes, in rank order.  The only wave that claims — with every wave scanning for
//                            itself they all find the same first candidate and beyond ~8 waves the claims serialise (measured: 33
//                            waves 38 ms, 125 waves 57 ms per image; with the dispatcher 13 ms before the two items above);
//                     wave 2 the FEEDER: for the free seeds within LSD_FEED_AHEAD ranks of the committer whose table entry shows a
//                            finished region, segment + pixel list from the L2 into a record ring in LDS, in rank order;
//   workgroups b + 8 k  four SPECULATING waves each: mail box -> grow_region_w against the global flags (loads that bypass the L1) ->
//                            region_segment_w -> list + segment stored, waited for (vmcnt), table entry, mail box "idle".
// What the workgroups exchange lives in HBM and meets in that XCD's L2: stores are ordinary (write-through L1, the line stays in the
// L2), loads of anything another CU stores during the launch bypass the vector L1 (sc1).  The flags and the mail boxes are ADVISORY
// for a speculating wave — a stale flag costs a failed validation, never a wrong region; what the committer (or the feeder) takes on
// trust are the pixel list + segment of a finished region: stored, waited for, THEN the table entry (a workgroup-scope release does
// not wait for the stores — the first version lost a segment that way in one of two images).
// Placement is verified, never assumed: the committer publishes the id of ITS XCD (HW_REG_XCC_ID, an agent-scope store), a
// speculating workgroup that finds itself on another one leaves without announcing itself — the committer alone is the sequential
// search, so any placement gives the same segments, only later.
constexpr int LSD_XW = 4;          // waves per workgroup: one per SIMD
constexpr int LSD_X_MAXW = 125;    // waves per image at most (7 bits of a table entry; 31 speculating workgroups of an XCD's 32 CUs)
constexpr int LSD_CTL = 512;       // control words per image, zeroed per call
constexpr size_t LSD_XCD_MAX_PX = 3u << 18;  // the committer's bitmap (96 KB) + the feeder's ring + the per-wave scratch fit the 160 KB of a CU
constexpr int LSD_FEED_PX = 8192;   // words of the feeder's record ring (LDS)
constexpr int LSD_FEED_Q = 256;     // records in it at most
constexpr int LSD_FEED_MAXW = 2048; // a record's pixels at most (a quarter of the ring; larger regions go through the table)
constexpr int LSD_FEED_AHEAD = 192;  // ranks the feeder runs ahead of the committer at most (further ahead most regions are still being grown: 2048: a quarter of the regions found finished, 128 - 256: 95 %)
constexpr int XC_ALIVE = 0;        // the committer's XCC id + 1 (0: it has not started)
constexpr int XC_DONE = 1;
constexpr int XC_MBOX = 64;        // [128] 64-bit mail boxes: 0 = wave v is not there, 1 = idle, else bit 62 | rank << 21 | seed pixel: grow this one
struct LsdXcd {
    int32_t* stamp;  // [B][nw][w h] (slot 0 unused)
    int32_t* wlist;  // [B][nw][w h] per wave: region after region, 4 words of segment + the pixel list
    long long* pend; // [B][w h] by rank: 0 free, LSD_CLAIM | wave, or lsd_entry_x
    int32_t* ctl;    // [B][LSD_CTL]
    int nsb;         // speculating workgroups per image; nw = 1 + nsb LSD_XW
    int feed_ahead, sep, ahead;  // LSD_FEED_AHEAD, LSD_SEP, LSD_AHEAD or their developer overrides (STVO_LSD_FEED_AHEAD / _SEP / _AHEAD)
    int multi;       // 0: the committer takes its seeds one by one (STVO_LSD_MULTI=0)
};
__device__ __forceinline__ long long lsd_entry_x(int wave, int n, int off) { return (1ll << 62) | ((long long)wave << 42) | ((long long)n << 21) | (long long)off; }

__global__ __launch_bounds__(LSD_XW * 64) void lsd_grow_xcd_kernel(LsdDev d, LsdXcd x) {
    __shared__ int s_ring[LSD_XW][LSD_WRING];
    __shared__ double s_term[LSD_XW][3][64];
    __shared__ int s_scan, s_done, s_qhead, s_qtail, s_cwords;
    // image b = XCD (b mod 8), place (b div 8) there; its 1 + nsb workgroups are neighbours in the XCD's share of the grid
    const int slot = blockIdx.x >> 3, role = slot % (1 + x.nsb), b = (blockIdx.x & 7) + 8 * (slot / (1 + x.nsb));
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (b >= d.B) return;
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    const int w = d.w, h = d.h, npx = w * h;
    const size_t base = (size_t)b * npx;
    LsdPx* px = d.px + base;
    const int32_t* __restrict__ k32 = d.k32 + base;
    const uint32_t* __restrict__ order = d.order + base;
    long long* pend = x.pend + base;
    int32_t* ctl = x.ctl + (size_t)b * LSD_CTL;
    long long* mbox = reinterpret_cast<long long*>(ctl + XC_MBOX);
    const int nw = 1 + x.nsb * LSD_XW;
    // dynamic LDS of the committer's workgroup: its flags, one bit per pixel | the feeder's record ring | the records' descriptors
    extern __shared__ unsigned s_bits[];
    const int nbw = (((npx + 31) / 32) + 3) & ~3;  // (whole 16-byte units: the descriptors behind are int4)
    int* const s_px = reinterpret_cast<int*>(s_bits + nbw);  // [LSD_FEED_PX] per record: its pixel indices, the record's number (mod 256) in the top byte
    int4* const s_desc = reinterpret_cast<int4*>(s_px + LSD_FEED_PX);  // [LSD_FEED_Q] {rank, size, position in s_px, words fed up to its end}
    float4* const s_dseg = reinterpret_cast<float4*>(s_desc + LSD_FEED_Q);  // [LSD_FEED_Q] the records' segments
    if (role == 0) {
        if (threadIdx.x == 0) {
            s_scan = -1;
            s_done = 0;
            s_qhead = 0;
            s_qtail = 0;
            s_cwords = 0;
        }
        for (int i = threadIdx.x; i < nbw; i += LSD_XW * 64) s_bits[i] = 0u;
        __syncthreads();
    }
    // keys and table entries of LSD_SB batches per round trip, the next such span requested before this one is worked on (the
    // committer and the feeder walk the ranks alike)
    constexpr int LSD_SB = 4;
    uint32_t kn[LSD_SB], kc[LSD_SB];
    long long en[LSD_SB], ec[LSD_SB];
    auto request = [&](int s0, bool entries) {
#pragma unroll
        for (int i = 0; i < LSD_SB; ++i) {
            const int r = s0 + 64 * i + lane;
            kn[i] = r < npx ? order[r] : LSD_NOKEY;
            en[i] = entries && r < npx ? ld_l2_64(pend + r) : 0ll;
        }
    };
    auto next_span = [&](int s0, bool entries) {
#pragma unroll
        for (int i = 0; i < LSD_SB; ++i) {
            kc[i] = kn[i];
            ec[i] = en[i];
        }
        request(s0 + 64 * LSD_SB, entries);
    };
    auto batch_regs = [&](int bi, uint32_t& key, long long& ent) {
        key = kc[0];
        ent = ec[0];
#pragma unroll
        for (int i = 1; i < LSD_SB; ++i) {
            key = bi == i ? kc[i] : key;
            ent = bi == i ? ec[i] : ent;
        }
    };
    if (role == 0 && wv == 0) {
        // ---------------- the committer ----------------
        // One wave on its SIMD: every instruction costs its issue latency, so the wave executes as little as possible per region.
        // Its flags are the bitmap in LDS (the global flags are written behind, for the speculating waves, and never read here); the
        // finished regions of the seeds in front of it come through LDS too — the feeder wave (below) has read table entry, segment
        // and pixel list from the L2 and left a record in rank order.  A seed without a record (its region was not finished when the
        // feeder passed, or the feeder is behind) takes the path through the table.
        __builtin_amdgcn_s_setprio(3);
        int32_t* wlist = x.wlist + (size_t)b * nw * npx;
        if (lane == 0) __hip_atomic_store(ctl + XC_ALIVE, xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int n_seg = 0;
        const bool prof = d.dbg != nullptr && b == 0;  // tools/lsd_probe.py: where the committer's time goes
        auto tick = [&]() -> long long { return prof ? (long long)__builtin_readcyclecounter() : 0ll; };
        const long long t_begin = tick();
        long long t_self = 0, t_wait = 0, t_take = 0, t_multi = 0, t_peek = 0, t_refresh = 0;
        int n_took = 0, n_self = 0, n_bad = 0, n_waited = 0, n_fast = 0, n_stale = 0, n_batch = 0, n_multi = 0, n_multi_undo = 0, n_multi_try = 0;
        int qt = 0, qh = 0;  // records consumed / known to be there
        request(0, false);
        bool at_end = false;
        for (int s0 = 0; s0 < npx && !at_end; s0 += 64 * LSD_SB) {
        next_span(s0, false);
        for (int bi = 0; bi < LSD_SB; ++bi) {
            const int o0 = s0 + 64 * bi;
            uint32_t key;
            long long ent_unused;
            batch_regs(bi, key, ent_unused);
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)key) == LSD_NOKEY) {
                at_end = true;
                break;
            }
            ++n_batch;
            const int q_l = (int)(key & ((1u << LSD_IDX_BITS) - 1u));
            const bool key_ok = key != LSD_NOKEY;
            unsigned long long todo = __ballot(key_ok && bit_ld(s_bits, key_ok ? q_l : 0) == 0u);
            bool one_record = false;  // after a pass whose regions shared a pixel: the next pass takes the first record alone
            while (todo) {
                // ---- the records at the front of the feeder's queue against the free seeds of the batch, up to eight at once.  A lane
                // below 8 holds a descriptor, a lane of the batch that is a free seed finds "its" descriptor by its ordinal among the
                // free seeds; the pass covers the leading seeds whose records are there, one behind the other in the ring and together
                // at most 128 pixels.  All their pixels are tested in one LDS round trip and taken together as far as the sequential
                // search would take them — up to the first region with a used pixel — unless two of them share a pixel (the returning
                // ds_or of the take shows it: then the bits are cleared again and the first record goes alone).
                const long long tm0 = tick();
                ++n_multi_try;
                if (qh - qt < 8) qh = __hip_atomic_load(&s_qhead, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int have = qh - qt < 8 ? qh - qt : 8;
                const int4 dl = s_desc[(qt + (lane & 7)) & (LSD_FEED_Q - 1)];  // {rank, pixels, position, words fed up to its end}
                const bool is_seed = (todo >> lane) & 1ull;
                const int ord = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(todo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)todo, 0u));
                const int first_rank = o0 + __builtin_ctzll(todo);
                int k = 0;
                {
                    const int drank = __shfl(dl.x, ord & 7, 64);  // the rank in the descriptor with this seed's ordinal
                    const unsigned long long hit = __ballot(is_seed && ord < have && drank == o0 + lane);
                    const unsigned long long miss = todo & ~hit;
                    k = __builtin_popcountll(miss ? todo & ((1ull << __builtin_ctzll(miss)) - 1ull) : todo);
                    // one behind the other in the ring (a wrap shows as a jump of the running word count), 128 pixels together
                    const int base0 = __builtin_amdgcn_readfirstlane(dl.w - dl.y), pos0c = __builtin_amdgcn_readfirstlane(dl.z);
                    const int startw = dl.w - dl.y - base0;
                    const unsigned inrow = (unsigned)__ballot(lane < 8 && dl.z == pos0c + startw && startw + dl.y <= 128) & 0xFFu;
                    const int kc = __builtin_ctz(~inrow);
                    k = k < kc ? k : kc;
                    if ((one_record || !x.multi) && k > 1) k = 1;
                    one_record = false;
                }
                bool regrow_first = false;
                if (k >= 1) {
                    const int pos0 = __builtin_amdgcn_readfirstlane(dl.z);
                    const int total = __builtin_amdgcn_readlane(dl.w, k - 1) - __builtin_amdgcn_readfirstlane(dl.w - dl.y);
                    const bool a0 = lane < total, a1 = 64 + lane < total;
                    const int w0 = a0 ? s_px[pos0 + lane] : 0, w1 = a1 ? s_px[pos0 + 64 + lane] : 0;
                    const int q0 = w0 & 0xFFFFFF, q1 = w1 & 0xFFFFFF;
                    const int r0 = (((unsigned)w0 >> 24) - (unsigned)qt) & 0xFFu, r1 = (((unsigned)w1 >> 24) - (unsigned)qt) & 0xFFu;  // the record of a word, from the front
                    const unsigned long long b0 = __ballot(a0 && bit_ld(s_bits, q0) != 0u), b1 = __ballot(a1 && bit_ld(s_bits, q1) != 0u);
                    int kk = k;  // the records in front of the first one with a used pixel
                    if (b0) kk = __builtin_amdgcn_readlane(r0, __builtin_ctzll(b0));
                    else if (b1) kk = __builtin_amdgcn_readlane(r1, __builtin_ctzll(b1));
                    if (kk == 0) {  // the first record has lost a pixel: its region is grown again below
                        regrow_first = true;
                        ++qt;
                        ++n_bad;
                        if (lane == 0) {
                            lds_st(&s_cwords, __builtin_amdgcn_readfirstlane(dl.w));
                            lds_st(&s_qtail, qt);
                        }
                    } else {
                        const bool t0 = a0 && r0 < kk, t1 = a1 && r1 < kk;
                        unsigned o0w = 0u, o1w = 0u;
                        if (t0) o0w = __hip_atomic_fetch_or(s_bits + (q0 >> 5), 1u << (q0 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (t1) o1w = __hip_atomic_fetch_or(s_bits + (q1 >> 5), 1u << (q1 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const bool c0 = t0 && ((o0w >> (q0 & 31)) & 1u), c1 = t1 && ((o1w >> (q1 & 31)) & 1u);
                        if (__ballot(c0 || c1)) {  // two of the regions share a pixel: undo (the bits were clear before), the first record alone next
                            if (t0 && !c0) (void)__hip_atomic_fetch_and(s_bits + (q0 >> 5), ~(1u << (q0 & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (t1 && !c1) (void)__hip_atomic_fetch_and(s_bits + (q1 >> 5), ~(1u << (q1 & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            ++n_multi_undo;
                            one_record = true;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            t_multi += tick() - tm0;
                            continue;
                        }
                        if (t0) st_coherent(&px[q0].used, 1);
                        if (t1) st_coherent(&px[q1].used, 1);
                        // the segments of the regions large enough, in order: a descriptor's lane each
                        const unsigned long long big = __ballot(lane < kk && dl.y >= d.min_reg_size);
                        if ((big >> lane) & 1ull) {
                            const int idx = n_seg + __builtin_popcountll(big & ((1ull << lane) - 1ull));
                            if (idx < d.seg_cap) d.seg[(size_t)b * d.seg_cap + idx] = s_dseg[(qt + lane) & (LSD_FEED_Q - 1)];
                        }
                        n_seg += __builtin_popcountll(big);
                        todo &= ~__ballot(is_seed && ord < kk);
                        const int last_rank = __builtin_amdgcn_readlane(dl.x, kk - 1), last_end = __builtin_amdgcn_readlane(dl.w, kk - 1);
                        qt += kk;
                        n_took += kk;
                        n_fast += kk;
                        ++n_multi;
                        if (lane == 0) {
                            lds_st(&s_scan, last_rank);
                            lds_st(&s_cwords, last_end);
                            lds_st(&s_qtail, qt);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        todo &= __ballot(key_ok && bit_ld(s_bits, key_ok ? q_l : 0) == 0u);
                        t_multi += tick() - tm0;
                        continue;
                    }
                } else if (have >= 1 && __builtin_amdgcn_readfirstlane(dl.x) == first_rank) {
                    // the first seed's record is too large for a pass (more than 128 pixels: 3 % of the regions): alone, chunk by chunk
                    const int n = __builtin_amdgcn_readfirstlane(dl.y), pos = __builtin_amdgcn_readfirstlane(dl.z);
                    bool bad = false;
                    for (int t = lane; t < n; t += 64) bad = bad || bit_ld(s_bits, s_px[pos + t] & 0xFFFFFF) != 0u;
                    const bool ok = !__ballot(bad);
                    if (ok) {
                        for (int t = lane; t < n; t += 64) {
                            const int q = s_px[pos + t] & 0xFFFFFF;
                            bit_set(s_bits, q);
                            st_coherent(&px[q].used, 1);
                        }
                        if (n >= d.min_reg_size) {
                            if (lane == 0 && n_seg < d.seg_cap) d.seg[(size_t)b * d.seg_cap + n_seg] = s_dseg[qt & (LSD_FEED_Q - 1)];
                            ++n_seg;
                        }
                        todo &= todo - 1ull;
                        ++n_took;
                        ++n_fast;
                    } else {
                        regrow_first = true;
                        ++n_bad;
                    }
                    ++qt;
                    if (lane == 0) {
                        if (ok) lds_st(&s_scan, first_rank);
                        lds_st(&s_cwords, __builtin_amdgcn_readfirstlane(dl.w));
                        lds_st(&s_qtail, qt);
                    }
                    if (ok) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        todo &= __ballot(key_ok && bit_ld(s_bits, key_ok ? q_l : 0) == 0u);
                        t_multi += tick() - tm0;
                        continue;
                    }
                } else {
                    // no record for the first free seed at the front of the queue: records of seeds taken meanwhile are passed over ...
                    const unsigned stale = (unsigned)__ballot(lane < have && dl.x < first_rank) & 0xFFu;
                    const int ns = __builtin_ctz(~stale);
                    if (ns > 0) {
                        const int cend = __builtin_amdgcn_readlane(dl.w, ns - 1);
                        qt += ns;
                        n_stale += ns;
                        if (lane == 0) {
                            lds_st(&s_cwords, cend);
                            lds_st(&s_qtail, qt);
                        }
                        t_multi += tick() - tm0;
                        continue;
                    }
                }
                // ---- ... and ONE seed goes through the table (its region was not finished when the feeder passed, is larger than a
                // record, or the feeder is behind), or — the record's region has lost a pixel — is grown again
                const long long tp0 = tick();
                t_multi += tp0 - tm0;
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const int seed = __builtin_amdgcn_readlane(q_l, j), rank = o0 + j;
                if (lane == 0) lds_st(&s_scan, rank);
                bool took = false;
                long long p = 1;
                const long long tk = tick();
                t_peek += tk - tp0;
                if (!regrow_first) {
                    auto table = [&]() { return readfirstlane64(ld_l2_64(pend + rank)); };  // (one address: a scalar result)
                    p = table();
                    if (p < 0) {  // a speculating wave is growing this very seed: its work is the work this wave would do (bounded wait)
                        const long long tw = tick();
                        for (int spin = 0; spin < (1 << 20) && (p = table()) < 0; ++spin) __builtin_amdgcn_s_sleep(2);
                        if (p < 0) p = 0;  // (gave up: this wave grows the seed itself, whatever the other one publishes later is never read)
                        t_wait += tick() - tw;
                        ++n_waited;
                    }
                    if (p) {
                        const int off = (int)(p & 0x1FFFFF), n = (int)((p >> 21) & 0x1FFFFF), pw = (int)((p >> 42) & 0x7F);
                        if (n > 0) {
                            const int32_t* pl = x.wlist + ((size_t)b * nw + pw) * npx + off;  // 4 words of segment, then the pixels
                            const int w0 = lane < n + 4 ? ld_l2(pl + lane) : 0;
                            const bool has = lane >= 4 && lane < n + 4;
                            const int q0 = (w0 >> 16) * w + (w0 & 0xFFFF);
                            bool bad = has && bit_ld(s_bits, q0) != 0u;
                            for (int t = 64 + lane; t < n + 4; t += 64) {  // (a long list: the rest in further round trips)
                                const int pxy = ld_l2(pl + t);
                                bad = bad || bit_ld(s_bits, (pxy >> 16) * w + (pxy & 0xFFFF)) != 0u;
                            }
                            if (!__ballot(bad)) {  // every pixel still free: this IS the region of the sequential search
                                if (has) {
                                    bit_set(s_bits, q0);
                                    st_coherent(&px[q0].used, 1);
                                }
                                for (int t = 64 + lane; t < n + 4; t += 64) {
                                    const int pxy = ld_l2(pl + t);
                                    const int q = (pxy >> 16) * w + (pxy & 0xFFFF);
                                    bit_set(s_bits, q);
                                    st_coherent(&px[q].used, 1);
                                }
                                took = true;
                                if (n >= d.min_reg_size) {
                                    if (lane < 4 && n_seg < d.seg_cap) reinterpret_cast<int*>(d.seg + (size_t)b * d.seg_cap + n_seg)[lane] = w0;
                                    ++n_seg;
                                }
                            }
                        }
                    }
                    if (took) ++n_took;
                    else if (p) ++n_bad;
                    else ++n_self;
                }
                const long long ts = tick();
                t_take += ts - tk;
                if (!took) {
                    double reg_angle;
                    const int n = grow_region_w<true, false, true>(px, nullptr, 0, wlist, npx, s_ring[0], seed, px[seed].ang, w, h, d.prec,
                                                                   reg_angle, s_bits);
                    if (n >= d.min_reg_size) {
                        const float4 sg = region_segment_w(d, wlist, n, k32, reg_angle, s_term[0]);
                        if (lane == 0 && n_seg < d.seg_cap) d.seg[(size_t)b * d.seg_cap + n_seg] = sg;
                        ++n_seg;
                    }
                    t_self += tick() - ts;
                }
                // seeds of the batch taken meanwhile (an LDS round trip)
                const long long tr0 = tick();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                todo &= __ballot(key_ok && bit_ld(s_bits, key_ok ? q_l : 0) == 0u);
                t_refresh += tick() - tr0;
            }
        }
        }
        if (lane == 0) {
            d.n_seg[b] = n_seg;
            lds_st(&s_done, 1);
            st_coherent(ctl + XC_DONE, 1);
        }
        if (prof && lane == 0) {  // the last 16 doubles of image 0's block (row seg_cap - 1: marker -1 = a committer's counters)
            double* q = d.dbg + ((size_t)d.seg_cap - 2) * 8;
            q[0] = (double)(tick() - t_begin); q[1] = (double)t_self; q[2] = (double)t_wait; q[3] = (double)t_take;
            q[4] = (double)n_took; q[5] = (double)n_self; q[6] = (double)n_bad; q[7] = (double)n_waited;
            q[8] = -1.0;
            q[9] = (double)n_fast; q[10] = (double)n_stale; q[11] = (double)n_batch; q[12] = (double)n_multi; q[13] = (double)n_multi_undo;
            d.dbg[((size_t)d.seg_cap - 5) * 8 + 0] = (double)t_multi; d.dbg[((size_t)d.seg_cap - 5) * 8 + 1] = (double)t_peek;
            d.dbg[((size_t)d.seg_cap - 5) * 8 + 2] = (double)t_refresh; d.dbg[((size_t)d.seg_cap - 5) * 8 + 3] = (double)n_multi_try;
        }
    } else if (role == 0 && wv == 2) {
        // ---------------- the feeder: the finished regions in front of the committer, from the L2 into LDS, in rank order ----------------
        // For every seed that is free now and whose table entry shows a finished region: segment + pixel list (as pixel INDICES) into
        // the record ring, a descriptor behind it.  Never waited for by the committer; waits itself for room in the ring and keeps
        // within LSD_FEED_AHEAD ranks of the committer (entries further ahead are mostly still being grown).
        constexpr int GRP = 4;
        int wpos = 0, qhd = 0;  // words / records fed so far
        request(0, true);
        bool at_end = false;
        auto gone = [&]() { return lds_ld(&s_done) != 0; };
        for (int s0 = 0; s0 < npx && !at_end; s0 += 64 * LSD_SB) {
        next_span(s0, true);
        for (int bi = 0; bi < LSD_SB && !at_end; ++bi) {
            const int o0 = s0 + 64 * bi;
            uint32_t key;
            long long ent;
            batch_regs(bi, key, ent);
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)key) == LSD_NOKEY) {
                at_end = true;
                break;
            }
            for (int spin = 0; o0 > lds_ld(&s_scan) + x.feed_ahead; ++spin) {
                if (gone() || spin > (1 << 22)) return;
                __builtin_amdgcn_s_sleep(8);
            }
            const bool key_ok = key != LSD_NOKEY;
            const int q_l = (int)(key & ((1u << LSD_IDX_BITS) - 1u));
            // (entries read a span ago: those of free seeds that were not finished then are read again now that their batch is near)
            if (__ballot(key_ok && (ent >> 62) != 1ll && bit_ld(s_bits, key_ok ? q_l : 0) == 0u)) {
                const long long e2 = ld_l2_64(pend + (o0 + lane < npx ? o0 + lane : 0));
                ent = o0 + lane < npx ? e2 : ent;
            }
            const int en_n = (int)((ent >> 21) & 0x1FFFFF);
            unsigned long long g = __ballot(key_ok && (ent >> 62) == 1ll && en_n != 0 && en_n <= LSD_FEED_MAXW && bit_ld(s_bits, key_ok ? q_l : 0) == 0u);
            while (g) {
                int J[GRP], W[GRP];
                const int32_t* PL[GRP];
#pragma unroll
                for (int k = 0; k < GRP; ++k) {  // the loads of a group together
                    J[k] = g ? __builtin_ctzll(g) : -1;
                    g &= g - 1ull;
                    W[k] = 0;
                    PL[k] = nullptr;
                    if (J[k] >= 0) {
                        const long long p = readlane_i64(ent, J[k]);
                        const int off = (int)(p & 0x1FFFFF), n = (int)((p >> 21) & 0x1FFFFF), pw = (int)((p >> 42) & 0x7F);
                        PL[k] = x.wlist + ((size_t)b * nw + pw) * npx + off;
                        if (lane < n + 4) W[k] = ld_l2(PL[k] + lane);
                    }
                }
#pragma unroll
                for (int k = 0; k < GRP; ++k) {
                    if (J[k] < 0) continue;
                    const long long p = readlane_i64(ent, J[k]);
                    const int n = (int)((p >> 21) & 0x1FFFFF), L = n;
                    int pos = wpos & (LSD_FEED_PX - 1);
                    if (pos + L > LSD_FEED_PX) {  // (records are contiguous: the rest of the ring's end stays unused)
                        wpos += LSD_FEED_PX - pos;
                        pos = 0;
                    }
                    for (int spin = 0; wpos + L - lds_ld(&s_cwords) > LSD_FEED_PX || qhd - lds_ld(&s_qtail) >= LSD_FEED_Q; ++spin) {
                        if (gone() || spin > (1 << 22)) return;
                        __builtin_amdgcn_s_sleep(4);
                    }
                    // the list's words: the segment (4) to the descriptor's side, the pixels as INDICES tagged with the record's number
                    const int tag = (qhd & 0xFF) << 24;
                    if (lane < 4) reinterpret_cast<int*>(s_dseg + (qhd & (LSD_FEED_Q - 1)))[lane] = W[k];
                    else if (lane < n + 4) s_px[pos + lane - 4] = ((W[k] >> 16) * w + (W[k] & 0xFFFF)) | tag;
                    for (int t = 64 + lane; t < n + 4; t += 64) {
                        const int pxy = ld_l2(PL[k] + t);
                        s_px[pos + t - 4] = ((pxy >> 16) * w + (pxy & 0xFFFF)) | tag;
                    }
                    wpos += L;
                    if (lane == 0) s_desc[qhd & (LSD_FEED_Q - 1)] = make_int4(o0 + J[k], n, pos, wpos);
                    ++qhd;
                    if (lane == 0) __hip_atomic_store(&s_qhead, qhd, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // (the record before its count)
                }
            }
        }
        }
    } else if (role == 0 && wv == 1) {
        // ---------------- the dispatcher: hands the free seeds in front of the committer to idle speculating waves, in rank order ----------------
        // The ONLY wave that claims: no compare-and-swap, no two waves after the same seed (with every wave scanning for itself — the
        // one-workgroup form — all of them find the same first candidate: beyond ~8 waves the claims serialise).  Lane u stands for the
        // speculating waves u + 1 and u + 65: idle or not (its mail box), where its seed in flight lies.
        const int ns = nw - 1;
        int sxy0 = -1, sxy1 = -1;  // x | y << 16 of the seed in flight, -1: none
        int front = 0;             // every rank below is used, pending or claimed — for good
        const bool dprof = d.dbg != nullptr && b == 0;
        long long dp_iter = 0, dp_chunks = 0, dp_given = 0, dp_noidle = 0, dp_blocked = 0;
        for (int idle = 0; idle < (1 << 24); ++idle) {  // (bounded)
            if (lds_ld(&s_done)) break;
            const bool idle0 = lane < ns && ld_l2_64(mbox + lane) == 1, idle1 = lane + 64 < ns && ld_l2_64(mbox + 64 + lane) == 1;  // (1: the wave is there and idle)
            if (idle0) sxy0 = -1;
            if (idle1) sxy1 = -1;
            unsigned long long I0 = __ballot(idle0), I1 = __ballot(idle1);
            ++dp_iter;
            if (!(I0 | I1)) {
                ++dp_noidle;
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
            const int scan = lds_ld(&s_scan);
            const int start = front > scan + 1 ? front : scan + 1;
            if (start - scan > x.ahead) {  // far enough ahead of the committer: regions grown beyond see flags too stale to survive
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            bool contiguous = true, end = false;
            for (int c = 0; c < LSD_LOOK && (I0 | I1) && !end; c += 64) {
                const int r = start + c + lane;
                const uint32_t key = r < npx ? order[r] : LSD_NOKEY;
                end = (uint32_t)__builtin_amdgcn_readfirstlane((int)key) == LSD_NOKEY;
                if (end) break;
                const bool ok = key != LSD_NOKEY;
                const int q = (int)(key & ((1u << LSD_IDX_BITS) - 1u));
                const bool open = ok && bit_ld(s_bits, ok ? q : 0) == 0u && ld_coherent64(pend + (ok ? r : 0)) == 0;  // (the committer's bitmap; this wave's own claims: one CU)
                const int qx = q % w, qy = q / w;
                const unsigned long long m_open = __ballot(open);
                unsigned long long mm = m_open, given = 0ull;
                ++dp_chunks;
                while (mm && (I0 | I1)) {
                    const int L = __builtin_ctzll(mm);
                    mm &= mm - 1ull;
                    const int cx = __builtin_amdgcn_readlane(qx, L), cy = __builtin_amdgcn_readlane(qy, L);
                    auto near = [&](int sxy) {
                        const int ddx = cx - (sxy & 0xFFFF), ddy = cy - (sxy >> 16);
                        const int adx = ddx < 0 ? -ddx : ddx, ady = ddy < 0 ? -ddy : ddy;
                        return sxy >= 0 && (adx > ady ? adx : ady) < x.sep;
                    };
                    if (__ballot(near(sxy0) || near(sxy1))) {  // too close to a growth in flight: later (the front stays in front of it)
                        ++dp_blocked;
                        continue;
                    }
                    const int u = I0 ? __builtin_ctzll(I0) : 64 + __builtin_ctzll(I1);
                    if (u < 64) I0 &= I0 - 1ull;
                    else I1 &= I1 - 1ull;
                    if (lane == (u & 63)) {
                        if (u < 64) sxy0 = cx | (cy << 16);
                        else sxy1 = cx | (cy << 16);
                    }
                    if (lane == L) {
                        st_coherent64(pend + r, LSD_CLAIM | (long long)(u + 1));
                        st_coherent64(mbox + u, (1ll << 62) | ((long long)r << 21) | (long long)q);
                    }
                    given |= 1ull << L;
                    ++dp_given;
                    idle = 0;
                }
                if (contiguous) {  // the ranks of this chunk in front of the first one still open join the solid part
                    const unsigned long long rem = m_open & ~given;
                    front = start + c + (rem ? __builtin_ctzll(rem) : 64);
                    contiguous = rem == 0ull;
                }
            }
        }
        if (dprof && lane == 0) {
            double* q = d.dbg + ((size_t)d.seg_cap - 4) * 8;
            q[0] = (double)dp_iter; q[1] = (double)dp_chunks; q[2] = (double)dp_given; q[3] = (double)dp_noidle; q[4] = (double)dp_blocked;
        }
    } else if (role != 0) {
        // ---------------- a speculating wave ----------------
        {  // on the committer's XCD, or not at all
            int alive = 0;
            for (int spin = 0; spin < (1 << 18) && (alive = __builtin_amdgcn_readfirstlane(ld_l2(ctl + XC_ALIVE))) == 0; ++spin) __builtin_amdgcn_s_sleep(16);
            if (alive != xcc + 1) return;
        }
        const int v = 1 + (role - 1) * LSD_XW + wv;  // this wave among the image's waves
        int32_t* stamp = x.stamp + ((size_t)b * nw + v) * npx;
        int32_t* wl = x.wlist + ((size_t)b * nw + v) * npx;
        int id = 0, off = 0;
        const bool sprof = d.dbg != nullptr && b == 0;
        long long sp_busy = 0, sp_n = 0, sp_grow = 0;
        const long long sp_t0 = sprof ? (long long)__builtin_readcyclecounter() : 0ll;
        if (lane == 0) st_coherent64(mbox + (v - 1), 1ll);  // here, and idle
        for (int idle = 0; idle < (1 << 24); ++idle) {  // (bounded: a wave that is handed nothing for this long gives up)
            const long long m = readfirstlane64(ld_l2_64(mbox + (v - 1)));
            if (m <= 1) {
                if (__builtin_amdgcn_readfirstlane(ld_l2(ctl + XC_DONE))) break;
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            idle = 0;
            const long long sp_a = sprof ? (long long)__builtin_readcyclecounter() : 0ll;
            const int pick_r = (int)((m >> 21) & 0x1FFFFF), pick_q = (int)(m & 0x1FFFFF);
            int n = 0;
            if (npx - off >= 4096) {  // (out of room: empty records from here on — the committer grows those seeds itself)
                ++id;
                double reg_angle = 0.0;
                n = grow_region_w<false, true>(px, stamp, id, wl + off + 4, npx - off - 4, s_ring[wv], pick_q, px[pick_q].ang, w, h, d.prec, reg_angle);
                if (n >= d.min_reg_size) {
                    const float4 sg = region_segment_w(d, wl + off + 4, n, k32, reg_angle, s_term[wv]);
                    if (lane == 0) {
                        st_coherent(wl + off, __float_as_int(sg.x));
                        st_coherent(wl + off + 1, __float_as_int(sg.y));
                        st_coherent(wl + off + 2, __float_as_int(sg.z));
                        st_coherent(wl + off + 3, __float_as_int(sg.w));
                    }
                }
            }
            // the list and the segment have reached the L2 before the table entry leaves (a workgroup-scope release does not wait: the
            // entry is read on ANOTHER CU)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_coherent64(pend + pick_r, lsd_entry_x(v, n > 0 ? n : 0, off));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_coherent64(mbox + (v - 1), 1ll);  // idle again
            if (n > 0) off += 4 + n;
            if (sprof) {
                sp_busy += (long long)__builtin_readcyclecounter() - sp_a;
                ++sp_n;
            }
        }
        if (sprof && lane == 0) {  // tools/lsd_probe.py: the speculating waves together (row seg_cap - 3 of image 0's block, as integers)
            unsigned long long* q = reinterpret_cast<unsigned long long*>(d.dbg + ((size_t)d.seg_cap - 3) * 8);
            atomicAdd(q + 0, (unsigned long long)sp_busy);
            atomicAdd(q + 1, (unsigned long long)((long long)__builtin_readcyclecounter() - sp_t0));
            atomicAdd(q + 2, (unsigned long long)sp_n);
            atomicAdd(q + 3, 1ull);
            (void)sp_grow;
        }
    }
}

// cv::LineIterator(img, Point2f(sx, sy), Point2f(ex, ey)).count (LSDDetector_custom.cpp:286-287): Point2f -> Point by cvRound, then
// cv::clipLine on the image's rectangle (64-bit integers, the intersections through a double quotient truncated towards zero — the
// statement of oracle/stvo_lsd_oracle.c: clip_line), then max(|dx|, |dy|) + 1 for the 8-connected raster; 0 when nothing is left.
__device__ __forceinline__ int line_iterator_count(int cols, int rows, float sx, float sy, float ex, float ey) {
    long long x1 = __float2int_rn(sx), y1 = __float2int_rn(sy), x2 = __float2int_rn(ex), y2 = __float2int_rn(ey);
    const long long right = cols - 1, bottom = rows - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    if ((c1 | c2) != 0) return 0;
    const long long dx = x2 > x1 ? x2 - x1 : x1 - x2, dy = y2 > y1 ? y2 - y1 : y1 - y2;
    return (int)((dx > dy ? dx : dy) + 1);
}

// LSDDetectorC::detectImpl's loop over the segments of the (single) octave (:254-303) and the cut of stereoFrame.cpp:231-240
constexpr int KL_T = 256;
__global__ __launch_bounds__(KL_T) void lsd_keylines_kernel(LsdDev d) {
    extern __shared__ float s_resp[];  // [seg_cap] responses of the kept lines, detection order
    __shared__ int s_wave[KL_T / 64];
    __shared__ int s_run;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(d.n_seg[b], d.seg_cap);
    const float4* seg = d.seg + (size_t)b * d.seg_cap;
    int* s_src = reinterpret_cast<int*>(s_resp + d.seg_cap);  // [seg_cap] segment of kept line k
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += KL_T) {  // the lines that pass min_length, in detection order
        const int k = c0 + tid;
        bool keep = false;
        float len = 0.f;
        if (k < n) {
            float4 e = seg[k];
            if (e.x < 0) e.x = 0;  // checkLineExtremes
            if (e.x >= d.cols) e.x = (float)d.cols - 1.0f;
            if (e.z < 0) e.z = 0;
            if (e.z >= d.cols) e.z = (float)d.cols - 1.0f;
            if (e.y < 0) e.y = 0;
            if (e.y >= d.rows) e.y = (float)d.rows - 1.0f;
            if (e.w < 0) e.w = 0;
            if (e.w >= d.rows) e.w = (float)d.rows - 1.0f;
            const double d0 = (double)(e.x - e.z), d1 = (double)(e.y - e.w);
            const double length = (double)(float)sqrt(d0 * d0 + d1 * d1);
            keep = length > d.min_length;
            len = (float)length;
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wv] = __popcll(bal);
        __syncthreads();
        int pos = s_run + __popcll(bal & ((1ull << lane) - 1ull));
        for (int v = 0; v < wv; ++v) pos += s_wave[v];
        if (keep) {
            s_resp[pos] = __fdiv_rn(len, (float)(d.cols > d.rows ? d.cols : d.rows));
            s_src[pos] = k;
        }
        __syncthreads();
        if (tid == 0) {
            int q = 0;
            for (int v = 0; v < KL_T / 64; ++v) q += s_wave[v];
            s_run += q;
        }
        __syncthreads();
    }
    const int m = s_run;
    // the top-N cut of stereoFrame.cpp:231-240 — and when the CAPACITY binds (lsd_nfeatures = 0 or above max_keylines) the same
    // selection for K lines: the K strongest, not the first K in detection order (the uncut count goes to n_pass: stvo_lsd_counts)
    const int n_out = min(d.nfeatures != 0 ? min(d.nfeatures, m) : m, d.K);
    const bool cut = m > n_out;
    for (int i = tid; i < m; i += KL_T) {
        int rank = i;
        if (cut) {  // position in the stable descending order of the responses
            const float r = s_resp[i];
            rank = 0;
            for (int j = 0; j < m; ++j) {
                const float q = s_resp[j];
                rank += (q > r || (q == r && j < i)) ? 1 : 0;
            }
        }
        if (rank >= n_out) continue;
        float4 e = seg[s_src[i]];
        if (e.x < 0) e.x = 0;
        if (e.x >= d.cols) e.x = (float)d.cols - 1.0f;
        if (e.z < 0) e.z = 0;
        if (e.z >= d.cols) e.z = (float)d.cols - 1.0f;
        if (e.y < 0) e.y = 0;
        if (e.y >= d.rows) e.y = (float)d.rows - 1.0f;
        if (e.w < 0) e.w = 0;
        if (e.w >= d.rows) e.w = (float)d.rows - 1.0f;
        stvo_keyline kl;
        kl.sx = e.x; kl.sy = e.y; kl.ex = e.z; kl.ey = e.w;
        kl.angle = (float)atan2((double)(e.w - e.y), (double)(e.z - e.x));
        // cv::LineIterator count: end points rounded to pixels, the line CLIPPED to the image (checkLineExtremes leaves coordinates in
        // [cols - 0.5, cols) / [rows - 0.5, rows) as they are, and those round to the first position outside), 8-connected raster
        kl.num_pixels = line_iterator_count(d.cols, d.rows, e.x, e.y, e.z, e.w);
        d.lines[(size_t)b * d.K + rank] = kl;
        if (d.response) d.response[(size_t)b * d.K + rank] = s_resp[i];
    }
    if (tid == 0) {
        d.n_lines[b] = n_out;
        d.n_pass[b] = m;
    }
}

// the end points of key-line records as the rows stvo_frame_features::kl_l / kl_r take (unused rows zeroed)
__global__ __launch_bounds__(256) void keylines_xy_kernel(int stride, const stvo_keyline* lines, const int32_t* n_lines, float4* xy) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= stride) return;
    const stvo_keyline q = lines[(size_t)b * stride + i];
    xy[(size_t)b * stride + i] = i < n_lines[b] ? make_float4(q.sx, q.sy, q.ex, q.ey) : make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace
}  // namespace stvo

struct stvo_lsd {
    stvo_ctx* ctx = nullptr;
    stvo::LsdDev d{};
    stvo_lsd_params prm{};
    int k7[7] = {0, 0, 0, 0, 0, 0, 0};
    char* dev = nullptr;
    uint8_t *img = nullptr, *blur = nullptr, *scaled = nullptr;
    stvo_keyline* lines = nullptr;  // device buffers of the host-pointer entry points
    float* response = nullptr;
    int32_t* n_lines = nullptr;
    double* dbg = nullptr;
    char* wdev = nullptr;
    size_t stamp_bytes = 0, pend_bytes = 0;
    stvo::LsdXcd xx{};        // scratch of lsd_grow_xcd_kernel (small batches, the default), or null
    int xcd_lds = 0;          // its dynamic LDS: the committer's bitmap — and at least 84 KB, so that a CU holds ONE of its workgroups (a wave per SIMD)
};

namespace {

int lsd_enqueue(stvo_lsd* o, const uint8_t* images, stvo_keyline* lines, float* response, int32_t* n_lines) {
    stvo_ctx* ctx = o->ctx;
    hipStream_t s = ctx->stream;
    stvo::LsdDev d = o->d;
    const uint8_t* src = images;
    if (d.scale != 1) {
        stvo::launch_blur7_u8(s, d.B, d.cols, d.rows, images, o->blur, o->k7);
        stvo::launch_resize_linear_u8(s, d.B, d.cols, d.rows, d.w, d.h, o->blur, o->scaled, d.scale, d.scale);
        src = o->scaled;
    }
    d.scaled = src;
    d.lines = lines; d.response = response; d.n_lines = n_lines;
    HIP_TRY(ctx, hipMemsetAsync(d.kmax, 0xFF, (size_t)d.B * 4, s));
    const dim3 grid((d.w + 255) / 256, (d.h + stvo::LSD_ROWS - 1) / stvo::LSD_ROWS, d.B);
    hipLaunchKernelGGL(stvo::lsd_gradient_kernel, grid, dim3(256), 0, s, d);
    {   // the pseudo-ordering: counting sort by bin (histogram per unit, scan per image, stable scatter)
        const int npx = d.w * d.h, nunits = (npx + stvo::LSD_UNIT - 1) / stvo::LSD_UNIT;
        const dim3 ug((nunits + 3) / 4, d.B);
        const size_t lds_h = (size_t)4 * d.n_bins * 4;
        hipLaunchKernelGGL(stvo::lsd_hist_kernel, ug, dim3(256), lds_h, s, d);
        hipLaunchKernelGGL(stvo::lsd_scan_kernel, dim3(d.B), dim3(1024), 0, s, d);
        hipLaunchKernelGGL(stvo::lsd_scatter_kernel, ug, dim3(256), lds_h, s, d);
    }
    if (o->prm.refine == 1) {  // LSD_REFINE_STD: regions may give their pixels back — its own one-wave kernel, for every batch size
        hipLaunchKernelGGL(stvo::lsd_grow_refine_kernel, dim3(d.B), dim3(64), 0, s, d, o->prm.density_th);
    } else if (o->wdev) {  // small batches: a committing wave + speculating workgroups on the CUs of one XCD per image (lsd_grow_xcd_kernel)
        HIP_TRY(ctx, hipMemsetAsync(o->xx.stamp, 0, o->stamp_bytes, s));
        HIP_TRY(ctx, hipMemsetAsync(o->xx.pend, 0, o->pend_bytes + (size_t)d.B * stvo::LSD_CTL * 4, s));  // (the control words lie behind the table)
        hipLaunchKernelGGL(stvo::lsd_grow_xcd_kernel, dim3(8 * ((d.B + 7) / 8) * (1 + o->xx.nsb)), dim3(stvo::LSD_XW * 64), o->xcd_lds, s, d, o->xx);
    } else if (stvo::dbg().lsd_grow == 0) {
        hipLaunchKernelGGL(stvo::lsd_grow_kernel<false>, dim3(d.B), dim3(64), 0, s, d);
    } else {
        hipLaunchKernelGGL(stvo::lsd_grow_kernel<true>, dim3(d.B), dim3(64), 0, s, d);
    }
    const size_t lds = (size_t)d.seg_cap * 8;
    hipLaunchKernelGGL(stvo::lsd_keylines_kernel, dim3(d.B), dim3(stvo::KL_T), lds, s, d);
    return check_launch(ctx);
}

}  // namespace

extern "C" {

int stvo_lsd_create(stvo_ctx* ctx, int B, int cols, int rows, int max_keylines, const stvo_lsd_params* prm, stvo_lsd** out) {
    if (!ctx || !prm || !out || B < 1 || cols < 8 || rows < 8 || max_keylines < 1) return STVO_ERR_INVALID_ARG;
    if (prm->refine != 0 && prm->refine != 1) return STVO_ERR_UNSUPPORTED;  // LSD_REFINE_ADV (rect_improve + the NFA test) is not built; no shipped configuration uses it
    if (prm->refine == 1 && !(prm->density_th >= 0)) return STVO_ERR_INVALID_ARG;
    if (!(prm->scale > 0) || !(prm->ang_th > 0 && prm->ang_th < 180) || prm->n_bins < 1 || prm->n_bins > 4096 || prm->nfeatures < 0)
        return STVO_ERR_INVALID_ARG;
    // sort key = (n_bins - 1 - bin) << LSD_IDX_BITS | index, LSD_NOKEY = all ones: with 4096 bins the keys of bin 0 would share their
    // top 12 bits with the undefined pixels' key and the bin-bits-only sort would interleave them
    if (prm->n_bins > stvo::LSD_MAX_BINS) return STVO_ERR_UNSUPPORTED;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stvo_lsd* o = new (std::nothrow) stvo_lsd();
    if (!o) return STVO_ERR_HIP;
    o->ctx = ctx;
    o->prm = *prm;
    stvo::LsdDev& d = o->d;
    d.B = B; d.cols = cols; d.rows = rows; d.w = cols; d.h = rows;
    d.scale = prm->scale;
    if (prm->scale != 1) {
        const double sigma = prm->scale < 1 ? prm->sigma_scale / prm->scale : prm->sigma_scale;
        const unsigned hk = (unsigned)std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0)));
        if (hk != 3) {  // only the 7 x 7 blur is built (sigma between 0.54 and 0.80: scale 1.2 / sigma_scale 0.6 and OpenCV's own 0.8 / 0.6)
            delete o;
            return STVO_ERR_UNSUPPORTED;
        }
        double k[7], sum = 0.0;  // orc_lsd_kernel7
        for (int i = 0; i < 7; ++i) {
            const double x = i - 3;
            k[i] = std::exp(-x * x / (2.0 * sigma * sigma));
            sum += k[i];
        }
        for (int i = 0; i < 7; ++i) o->k7[i] = (int)std::lrint((float)(k[i] / sum) * 256.0);
        d.w = (int)std::lrint((double)cols * prm->scale);  // resize(..., Size(), scale, scale): cvRound
        d.h = (int)std::lrint((double)rows * prm->scale);
    }
    if ((long long)d.w * d.h > (1ll << stvo::LSD_IDX_BITS) || d.w >= 65536 || d.h >= 32768) {
        delete o;
        return STVO_ERR_CAPACITY;
    }
    d.n_bins = prm->n_bins;
    d.rho = prm->quant / std::sin(stvo::LSD_PI * prm->ang_th / 180);
    d.prec = stvo::LSD_PI * prm->ang_th / 180;
    {
        const double p = prm->ang_th / 180;
        const double log_nt = 5 * (std::log10((double)d.w) + std::log10((double)d.h)) / 2 + std::log10(11.0);
        d.min_reg_size = (int)(size_t)(-log_nt / std::log10(p));
    }
    d.min_length = prm->min_length;
    d.nfeatures = prm->nfeatures;
    d.K = max_keylines;
    d.seg_cap = 8192;  // segments per image the wrapper ranks (its LDS: 8 bytes each); more are counted, not stored
    const size_t npx = (size_t)d.w * d.h, nb = (size_t)B;
    struct {
        size_t off = 0;
        size_t take(size_t bytes) {
            const size_t o = off;
            off += (bytes + 255) & ~size_t(255);
            return o;
        }
    } c;
    const size_t o_blur = c.take(nb * cols * rows), o_scaled = c.take(nb * npx), o_img = c.take(nb * cols * rows), o_px = c.take(nb * npx * sizeof(stvo::LsdPx)),
                 o_k32 = c.take(nb * npx * 4), o_cnt = c.take(nb * ((npx + stvo::LSD_UNIT - 1) / stvo::LSD_UNIT) * (size_t)d.n_bins * 4),
                 o_order = c.take(nb * npx * 4), o_reg = c.take(nb * npx * 4), o_kmax = c.take(nb * 4), o_seg = c.take(nb * d.seg_cap * 16),
                 o_nseg = c.take(nb * 4), o_lines = c.take(nb * d.K * sizeof(stvo_keyline)),
                 o_resp = c.take(nb * d.K * 4), o_nl = c.take(nb * 4), o_np = c.take(nb * 4);
    bool ok = hip_ok(ctx, hipMalloc((void**)&o->dev, c.off), "hipMalloc lsd") && zero_device(ctx, o->dev, c.off, "hipMemset lsd");
    if (ok) {
        char* D = o->dev;
        o->blur = (uint8_t*)(D + o_blur); o->scaled = (uint8_t*)(D + o_scaled); o->img = (uint8_t*)(D + o_img);
        d.px = (stvo::LsdPx*)(D + o_px); d.k32 = (int32_t*)(D + o_k32); d.cnt = (uint32_t*)(D + o_cnt);
        d.order = (uint32_t*)(D + o_order); d.reg = (int32_t*)(D + o_reg); d.kmax = (int32_t*)(D + o_kmax);
        d.seg = (float4*)(D + o_seg); d.n_seg = (int32_t*)(D + o_nseg); d.n_pass = (int32_t*)(D + o_np);
        o->lines = (stvo_keyline*)(D + o_lines); o->response = (float*)(D + o_resp); o->n_lines = (int32_t*)(D + o_nl);
    }
    if (ok && (size_t)4 * d.n_bins * 4 > 48 * 1024) {
        ok = stvo::lds_opt_in(reinterpret_cast<const void*>(stvo::lsd_hist_kernel), 4 * d.n_bins * 4) &&
             stvo::lds_opt_in(reinterpret_cast<const void*>(stvo::lsd_scatter_kernel), 4 * d.n_bins * 4);
    }
    if (ok && (size_t)d.seg_cap * 8 > 48 * 1024)
        ok = stvo::lds_opt_in(reinterpret_cast<const void*>(stvo::lsd_keylines_kernel), d.seg_cap * 8);
    if (ok && prm->refine == 0 && stvo::dbg().lsd_waves != 0 && B <= stvo::LSD_WAVES_MAX_B && npx <= stvo::LSD_XCD_MAX_PX) {  // small batches (STVO_LSD_WAVES=0: one wave per image there too)
        // an XCD has 32 CUs and a CU holds one of the kernel's workgroups: images per XCD x (1 + speculating workgroups) <= 32
        const int per_xcd = (B + 7) / 8;
        int nsb = stvo::dbg().lsd_xcd_blocks == stvo::DBG_UNSET ? (32 / per_xcd - 1 < 16 ? 32 / per_xcd - 1 : 16) : stvo::dbg().lsd_xcd_blocks;  // STVO_LSD_XCD_BLOCKS
        nsb = nsb < 0 ? 0 : (nsb > (stvo::LSD_X_MAXW - 1) / stvo::LSD_XW ? (stvo::LSD_X_MAXW - 1) / stvo::LSD_XW : nsb);
        const size_t nw = 1 + (size_t)nsb * stvo::LSD_XW;
        decltype(c) cw;
        const size_t w_stamp = cw.take(nb * nw * npx * 4), w_list = cw.take(nb * nw * npx * 4), w_pend = cw.take(nb * npx * 8 + nb * stvo::LSD_CTL * 4);
        o->xcd_lds = (int)(((((npx + 31) / 32) + 3) & ~size_t(3)) * 4) + stvo::LSD_FEED_PX * 4 + stvo::LSD_FEED_Q * 32;
        if (o->xcd_lds < 84 * 1024) o->xcd_lds = 84 * 1024;
        ok = stvo::lds_opt_in(reinterpret_cast<const void*>(stvo::lsd_grow_xcd_kernel), o->xcd_lds) &&
             hip_ok(ctx, hipMalloc((void**)&o->wdev, cw.off), "hipMalloc lsd xcd");
        if (ok) {
            o->xx.stamp = (int32_t*)(o->wdev + w_stamp); o->xx.wlist = (int32_t*)(o->wdev + w_list);
            o->xx.pend = (long long*)(o->wdev + w_pend); o->xx.ctl = (int32_t*)(o->wdev + w_pend + nb * npx * 8);
            o->xx.nsb = nsb;
            const auto& g = stvo::dbg();
            o->xx.feed_ahead = g.lsd_feed_ahead == stvo::DBG_UNSET ? stvo::LSD_FEED_AHEAD : g.lsd_feed_ahead;
            o->xx.sep = g.lsd_sep == stvo::DBG_UNSET ? stvo::LSD_SEP : g.lsd_sep;
            o->xx.ahead = g.lsd_ahead == stvo::DBG_UNSET ? stvo::LSD_AHEAD : g.lsd_ahead;
            o->xx.multi = g.lsd_multi == 0 ? 0 : 1;
            o->stamp_bytes = nb * nw * npx * 4;
            o->pend_bytes = nb * npx * 8;
        }
    }
    if (!ok) {
        stvo_lsd_destroy(o);
        return STVO_ERR_HIP;
    }
    *out = o;
    return STVO_OK;
}

int stvo_lsd_destroy(stvo_lsd* o) {
    if (!o) return STVO_OK;
    if (o->ctx) {
        (void)hipSetDevice(o->ctx->device);
        (void)hipStreamSynchronize(o->ctx->stream);
    }
    if (o->dbg) (void)hipFree(o->dbg);
    if (o->wdev) (void)hipFree(o->wdev);
    if (o->dev) (void)hipFree(o->dev);
    delete o;
    return STVO_OK;
}

int stvo_lsd_detect_dev(stvo_lsd* o, const uint8_t* images, stvo_keyline* lines, float* response, int32_t* n_lines) {
    if (!o || !images || !lines || !n_lines) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lsd_enqueue(o, images, lines, response, n_lines);
}

int stvo_lsd_detect(stvo_lsd* o, const uint8_t* images, stvo_keyline* lines, float* response, int32_t* n_lines) {
    if (!o || !images || !lines || !n_lines) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const stvo::LsdDev& d = o->d;
    const size_t img_bytes = (size_t)d.B * d.cols * d.rows;
    HIP_TRY(ctx, hipMemcpyAsync(o->img, images, img_bytes, hipMemcpyHostToDevice, ctx->stream));
    TRY(lsd_enqueue(o, o->img, o->lines, o->response, o->n_lines));
    HIP_TRY(ctx, hipMemcpyAsync(lines, o->lines, (size_t)d.B * d.K * sizeof(stvo_keyline), hipMemcpyDeviceToHost, ctx->stream));
    if (response) HIP_TRY(ctx, hipMemcpyAsync(response, o->response, (size_t)d.B * d.K * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(n_lines, o->n_lines, (size_t)d.B * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return STVO_OK;
}

// developer aid, not part of include/: the rectangle intermediates of the last stvo_lsd_segments call's first image
extern "C" int stvo_lsd_debug(stvo_lsd* o, int enable, double* out /* [seg_cap][8] or NULL */) {
    if (!o) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (enable && !o->dbg) {
        HIP_TRY(ctx, hipMalloc((void**)&o->dbg, (size_t)o->d.B * o->d.seg_cap * 64));
        if (!zero_device(ctx, o->dbg, (size_t)o->d.B * o->d.seg_cap * 64, "hipMemset lsd dbg")) return STVO_ERR_HIP;
    }
    o->d.dbg = enable ? o->dbg : nullptr;
    if (out && o->dbg) HIP_TRY(ctx, hipMemcpy(out, o->dbg, (size_t)o->d.seg_cap * 64, hipMemcpyDeviceToHost));
    return STVO_OK;
}

int stvo_keylines_xy_dev(stvo_ctx* ctx, int B, int stride, const stvo_keyline* lines, const int32_t* n_lines, float* kl_xy) {
    if (!ctx || B < 1 || stride < 1 || !lines || !n_lines || !kl_xy) return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(stvo::keylines_xy_kernel, dim3((stride + 255) / 256, B), dim3(256), 0, ctx->stream, stride, lines, n_lines,
                       reinterpret_cast<float4*>(kl_xy));
    return check_launch(ctx);
}

// What the last detection found BEFORE its cuts, per image: n_segments = segments of the detector core (at most 8192 are ranked),
// n_passing = those longer than min_length (n_lines = min(n_passing, nfeatures if set, max_keylines)).  A caller that asked for
// "keep all" (lsd_nfeatures = 0) learns here whether the capacity cut its lines.  Host arrays [B]; synchronises.
int stvo_lsd_counts(stvo_lsd* o, int32_t* n_segments, int32_t* n_passing) {
    if (!o || (!n_segments && !n_passing)) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const stvo::LsdDev& d = o->d;
    if (n_segments) HIP_TRY(ctx, hipMemcpyAsync(n_segments, d.n_seg, (size_t)d.B * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (n_passing) HIP_TRY(ctx, hipMemcpyAsync(n_passing, d.n_pass, (size_t)d.B * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return STVO_OK;
}

int stvo_lsd_segments(stvo_lsd* o, const uint8_t* images, float* segments, int cap, int32_t* n_segments) {
    if (!o || !images || !segments || !n_segments || cap < 1) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const stvo::LsdDev& d = o->d;
    HIP_TRY(ctx, hipMemcpyAsync(o->img, images, (size_t)d.B * d.cols * d.rows, hipMemcpyHostToDevice, ctx->stream));
    TRY(lsd_enqueue(o, o->img, o->lines, o->response, o->n_lines));
    HIP_TRY(ctx, hipMemcpyAsync(n_segments, d.n_seg, (size_t)d.B * 4, hipMemcpyDeviceToHost, ctx->stream));
    const int take = cap < d.seg_cap ? cap : d.seg_cap;
    HIP_TRY(ctx, hipMemcpy2DAsync(segments, (size_t)cap * 16, d.seg, (size_t)d.seg_cap * 16, (size_t)take * 16, d.B, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return STVO_OK;
}

}  // extern "C"
