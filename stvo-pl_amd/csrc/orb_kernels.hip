// orb_kernels.hip — the ORB point front-end on gfx950: what the reference obtains from
//     cv::ORB::create(...)->detectAndCompute(img, Mat(), points, pdesc, false)     (/root/reference/src/stereoFrame.cpp:104-118)
// for orb_nlevels pyramid levels (config_kitti.yaml: 1; config_euroc.yaml / src/config.cpp:96-97: 4 at scale 1.2), FAST_SCORE
// ranking (orb_score 1), WTA_K 2, patch 31.  Per level:
//   orb_fast_nms_kernel  FAST-9/16 score (cornerScore<16>) + 3x3 non-maximum suppression + border filter per 64 x 64 tile, all in
//                        LDS: compass-point rejection, candidates compacted so that the full score runs on dense lanes;
//                        survivors go to a per-image list + response histogram (no score map in global memory)
//   orb_order_kernel     KeyPointsFilter::retainBest as a histogram cut (ties kept) + row-major ordering (bitonic sort in LDS)
//   orb_blur_kernel      GaussianBlur 7x7, sigma 2, 8-bit fixed point, BORDER_REFLECT_101
//   orb_describe_kernel  intensity-centroid angle (ICAngles, fastAtan2) + rotated BRIEF, one wave per key-point
// around them (more than one level): orb_resize_kernel (level l from level l - 1, OpenCV's 8-bit bilinear resize in 11-bit fixed
// point) and orb_concat_kernel (levels in order, coordinates x scale, octave = level).
// OpenCV is third-party code that is not under /root/reference: the semantics are those of oracle/stvo_orb_oracle.c (a
// restatement of OpenCV's algorithm, parity unpinned), against which these kernels are bit-exact (tests/test_gpu_orb.py).
// Integer / byte work throughout; the only floating point is the angle polynomial and the rotation of the test pattern, in
// FP32 exactly as OpenCV evaluates them (no fused multiply-adds).
#include <cmath>
#include <cstring>
#include <new>

#include "ctx_internal.h"
#include "fast_atan2.h"

#pragma clang fp contract(off)

namespace stvo {
namespace {

constexpr int ORB_HP = 15;  // half patch

struct OrbDev {
    int B, cols, rows, K, nfeatures, fast_th, edge_th;
    int cand_cap;         // rows * cols / 4 + 64: a 3x3 strict maximum every 2 x 2 pixels at most, so the list cannot overflow
    int32_t* n_total;     // [B] or nullptr: key-points that qualified before the cap K
    const uint8_t* img;   // [B][rows][cols]
    uint8_t* blur;        // [B][rows][cols]
    int32_t* hist;        // [B][256] responses of the key-points that survive NMS + border
    uint32_t* cand;       // [B][cand_cap] the survivors, unordered: (y << 20 | x << 8 | response)
    int32_t* n_cand;      // [B]
    float* kp;            // [B][K][2]
    float* resp;          // [B][K]
    float* angle;         // [B][K]
    uint8_t* desc;        // [B][K][32]
    int32_t* n_kp;        // [B]
    const int8_t* pattern;  // [256][4]
};

// circle offsets in OpenCV's order (dx, dy)
__device__ __constant__ int8_t c_circle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                                  {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;  // a word at any byte address (gfx950 global memory allows it)
constexpr int FT_W = 64, FT_H = 64;               // output tile of the fused FAST + NMS kernel (a workgroup's life is a chain of
                                                  // memory round trips and barriers: 64 x 32 halves their share against 64 x 16)
constexpr int SC_W = FT_W + 2, SC_H = FT_H + 2;   // scores are needed one pixel beyond the tile (3x3 maximum test)
constexpr int IM_H = FT_H + 8;                    // image tile rows: + 1 (score halo) + 3 (circle radius) on both sides
constexpr int IM_DW = 19, IM_PITCH = 20;          // image tile row: 19 words = pixels x0 - 5 .. x0 + 70 (score position sx <-> byte sx + 4: word aligned)
constexpr int SC_PITCH = 72;                      // score row, bytes (18 words)

// cornerScore<16> of the pixel at byte (lx, ly) of the LDS image tile: max over the 16 arcs of 9 contiguous circle pixels of
// the smallest (signed) difference in the arc, bright and dark, minus 1 — min / max over every window of 9 by doubling
__device__ __forceinline__ int fast_score_lds(const uint8_t* tile, int lx, int ly) {
    const uint8_t* c = tile + ly * (IM_PITCH * 4) + lx;
    const int v = c[0];
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = (int)c[c_circle[k][1] * (IM_PITCH * 4) + c_circle[k][0]] - v;
    int mn2[16], mx2[16], mn4[16], mx4[16], mn8[16], mx8[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn2[k] = min(d[k], d[(k + 1) & 15]);
        mx2[k] = max(d[k], d[(k + 1) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn4[k] = min(mn2[k], mn2[(k + 2) & 15]);
        mx4[k] = max(mx2[k], mx2[(k + 2) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn8[k] = min(mn4[k], mn4[(k + 4) & 15]);
        mx8[k] = max(mx4[k], mx4[(k + 4) & 15]);
    }
    int sb = -255, sd = -255;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        sb = max(sb, min(mn8[k], d[(k + 8) & 15]));
        sd = max(sd, -max(mx8[k], d[(k + 8) & 15]));
    }
    return max(sb, sd) - 1;
}

// One side of cornerScore<16> (round 6): with the differences taken as sgn (p - v) — sgn = +1 for the bright test, -1 for the dark one —
// both are "max over the 16 arcs of the smallest of the arc's 9 differences": half of fast_score_lds' work.  A pixel's score is the larger
// of its two sides minus 1; a side the signed compass test ruled out cannot reach t (a 9-arc holds two compass points), so only the
// sides that passed are evaluated (orb_fast_nms_kernel: one list entry per pixel and side).
__device__ __forceinline__ int fast_score_side_lds(const uint8_t* tile, int lx, int ly, int sgn) {
    const uint8_t* c = tile + ly * (IM_PITCH * 4) + lx;
    const int v = c[0];
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = ((int)c[c_circle[k][1] * (IM_PITCH * 4) + c_circle[k][0]] - v) * sgn;
    int mn2[16], mn4[16], mn8[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) mn2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) mn4[k] = min(mn2[k], mn2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) mn8[k] = min(mn4[k], mn4[(k + 4) & 15]);
    int sb = -255;
#pragma unroll
    for (int k = 0; k < 16; ++k) sb = max(sb, min(mn8[k], d[(k + 8) & 15]));
    return sb - 1;
}

// wave-aggregated append of up to four flagged items per lane: ONE LDS atomic per wave (lanes hammering one counter serialise)
__device__ __forceinline__ void append4(const bool (&flag)[4], int* counter, int (&slot)[4]) {
    const int lane = threadIdx.x & 63;
    unsigned long long bal[4];
    int total = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bal[j] = __ballot(flag[j]);
        total += __popcll(bal[j]);
    }
    int base = 0;
    if (lane == 0 && total) base = atomicAdd(counter, total);
    base = __shfl(base, 0, 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        slot[j] = base + __popcll(bal[j] & ((1ull << lane) - 1ull));
        base += __popcll(bal[j]);
    }
}

// FAST-9/16 + non-maximum suppression + border filter of one 64 x 32 tile, entirely in LDS:
//   1. the image tile (+ halo) is staged once, word by word (unaligned global words; byte by byte only at the image border);
//   2. every score position (tile + halo 1) takes a cheap NECESSARY test, four positions per thread on words: an arc of 9
//      contiguous circle pixels holds at least two of the four compass points, so a corner has sum |compass - centre| >= 2 (t + 1).
//      The 4 x 4 bytes (north, south, east, west of four pixels; east / west by byte alignment) are transposed with v_perm_b32
//      and each pixel's sum is ONE v_sad_u8 — 8 instructions per pixel instead of the 26 of the signed compass test;
//   3. the survivors (~15 % of the positions) are COMPACTED into a list, so that the signed compass test (at least two compass
//      points brighter, or two darker, than the centre by more than t) and the 150-instruction corner score run on dense lanes;
//      the scores go to an LDS tile and the corners to a second list;
//   4. non-maximum suppression walks that corner list (~3 % of the positions), not the tile; a corner that beats its 8
//      neighbours and lies inside the border is appended to the image's candidate list and counted in the response histogram —
//      no score / keep maps ever reach global memory.
__global__ __launch_bounds__(256) void orb_fast_nms_kernel(OrbDev o) {
    __shared__ uint32_t tile[IM_H][IM_PITCH];
    __shared__ uint32_t sc_w[SC_H][SC_PITCH / 4];
    __shared__ uint16_t s_list[SC_H * SC_W + 8];
    __shared__ uint16_t s_corner[SC_H * SC_W + 8];
    __shared__ uint32_t s_kp[FT_W * FT_H / 4];  // key-points of this tile (a 3x3 maximum every 4 pixels at most)
    __shared__ int s_n, s_nc, s_nkp, s_base;
    const int b = blockIdx.z, x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H, tid = threadIdx.x, lane = tid & 63;
    const uint8_t* img = o.img + (size_t)b * o.rows * o.cols;
    const uint8_t* tile8 = reinterpret_cast<const uint8_t*>(&tile[0][0]);
    uint8_t* sc = reinterpret_cast<uint8_t*>(&sc_w[0][0]);
    {
        // every word of the tile is requested before the first is used (as a loop of load -> wait -> LDS store the workgroup began
        // its life with six memory round trips one after the other).  Words that cross the image border are re-read byte by byte
        // afterwards; the first pass reads them from a clamped address so that it needs no branch
        constexpr int ST_TRIPS = (IM_H * IM_DW + 255) / 256;
        uint32_t wv[ST_TRIPS];
#pragma unroll
        for (int k = 0; k < ST_TRIPS; ++k) {
            const int i = min(tid + 256 * k, IM_H * IM_DW - 1);
            const int ty = i / IM_DW, td = i - ty * IM_DW;
            const int gx = x0 - 5 + 4 * td, gy = min(max(y0 + ty - 4, 0), o.rows - 1);
            wv[k] = *reinterpret_cast<const u32_unaligned*>(img + (size_t)gy * o.cols + min(max(gx, 0), o.cols - 4));
        }
#pragma unroll
        for (int k = 0; k < ST_TRIPS; ++k) {
            const int i = tid + 256 * k;
            if (i < IM_H * IM_DW) {
                const int ty = i / IM_DW, td = i - ty * IM_DW;
                const int gx = x0 - 5 + 4 * td, gy = min(max(y0 + ty - 4, 0), o.rows - 1);
                uint32_t w = wv[k];
                if (gx < 0 || gx + 4 > o.cols) {  // image border: replicate (those pixels never pass the border test below, their values only have to exist)
                    const uint8_t* row = img + (size_t)gy * o.cols;
                    w = 0u;
#pragma unroll
                    for (int c = 0; c < 4; ++c) w |= (uint32_t)row[min(max(gx + c, 0), o.cols - 1)] << (8 * c);
                }
                tile[ty][td] = w;
            }
        }
    }
    for (int i = tid; i < SC_H * (SC_PITCH / 4); i += 256) (&sc_w[0][0])[i] = 0u;
    if (tid == 0) {
        s_n = 0;
        s_nc = 0;
        s_nkp = 0;
    }
    __syncthreads();
    const int t = o.fast_th;
    // (all lanes of a wave run the same number of trips: the ballots of append4 need the whole wave)
    constexpr int GROUPS = (SC_W + 3) / 4;  // 17 groups of four score positions per row
    const uint32_t sad_min = 2u * (uint32_t)(t + 1);
    for (int i0 = 0; i0 < SC_H * GROUPS; i0 += 256) {
        const int i = i0 + tid;
        const bool item = i < SC_H * GROUPS;
        const int sy = item ? i / GROUPS : 0, g = item ? i - sy * GROUPS : 0;
        const int r = sy + 3, y = y0 + sy - 1;
        const uint32_t C = tile[r][g + 1], L = tile[r][g], R = tile[r][g + 2], N = tile[r - 3][g + 1], S = tile[r + 3][g + 1];
        const uint32_t W4 = __builtin_amdgcn_alignbyte(C, L, 1), E4 = __builtin_amdgcn_alignbyte(R, C, 3);
        // 4 x 4 byte transpose: Q[j] = (N, S, E, W) of pixel j
        const uint32_t ns01 = __builtin_amdgcn_perm(S, N, 0x05010400u), ns23 = __builtin_amdgcn_perm(S, N, 0x07030602u);
        const uint32_t ew01 = __builtin_amdgcn_perm(W4, E4, 0x05010400u), ew23 = __builtin_amdgcn_perm(W4, E4, 0x07030602u);
        uint32_t Q[4];
        Q[0] = __builtin_amdgcn_perm(ew01, ns01, 0x05040100u);
        Q[1] = __builtin_amdgcn_perm(ew01, ns01, 0x07060302u);
        Q[2] = __builtin_amdgcn_perm(ew23, ns23, 0x05040100u);
        Q[3] = __builtin_amdgcn_perm(ew23, ns23, 0x07060302u);
        const bool row_ok = item && y >= 3 && y < o.rows - 3;
        const int xb = x0 + 4 * g - 1;  // image column of pixel 0
        bool cand[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t V = __builtin_amdgcn_perm(C, C, 0x01010101u * (uint32_t)j);  // the centre in all four bytes
            const uint32_t sad = __builtin_amdgcn_sad_u8(Q[j], V, 0u);
            cand[j] = row_ok && 4 * g + j < SC_W && xb + j >= 3 && xb + j < o.cols - 3 && sad >= sad_min;
        }
        int slot[4];
        append4(cand, &s_n, slot);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (cand[j]) s_list[slot[j]] = (uint16_t)(sy * SC_W + 4 * g + j);
    }
    __syncthreads();
    const int n = s_n;
    // 3a. the signed compass test on the dense list (circle points 0, 4, 8, 12 = south, east, north, west): at least two of them brighter, or two
    //     darker, than the centre by more than t — what passes (~40 % of the list) is compacted ONCE MORE, into s_corner, so that the ~200
    //     instructions of the corner score run on dense lanes too (round 6: a wave of the first list ran the score if any of its lanes passed)
    for (int j = tid; j < n; j += 256) {  // lanes of a wave carry consecutive j: lane 0 is active whenever any lane is
        const int i = s_list[j], sy = i / SC_W, sx = i - sy * SC_W;
        const uint8_t* c = tile8 + (sy + 3) * (IM_PITCH * 4) + (sx + 4);
        const int v = c[0];
        const int c0 = (int)c[3 * (IM_PITCH * 4)] - v, c4 = (int)c[3] - v, c8 = (int)c[-3 * (IM_PITCH * 4)] - v, c12 = (int)c[-3] - v;
        const int nb = (c0 > t) + (c4 > t) + (c8 > t) + (c12 > t), nd = (c0 < -t) + (c4 < -t) + (c8 < -t) + (c12 < -t);
        // one entry per pixel with the side(s) that passed: bit 15 = the dark side only, bit 14 = both (a saddle: rare)
        const bool pb = nb >= 2, pd = nd >= 2, pass = pb || pd;
        const unsigned long long bal = __ballot(pass);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_nc, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (pass) s_corner[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)(i | (pb && pd ? 0x4000 : (pd ? 0x8000 : 0)));
    }
    __syncthreads();
    const int n_pass = s_nc;
    __syncthreads();  // (s_nc is counted anew below; s_list is free: the corners go there)
    if (tid == 0) {
        s_nc = 0;
        s_n = 0;
    }
    __syncthreads();
    // 3b. the score of the side that passed, on the dense list (both sides for the rare entries that ask for it: a uniform branch most
    //     waves skip); a pixel whose score reaches t is a corner: score tile + corner list (s_list)
    for (int j = tid; j < n_pass; j += 256) {
        const int e = s_corner[j], i = e & 0x3FFF, sy = i / SC_W, sx = i - sy * SC_W;
        int s = fast_score_side_lds(tile8, sx + 4, sy + 3, (e & 0x8000) ? -1 : 1);
        if (__any((e & 0x4000) != 0)) {
            const int s2 = fast_score_side_lds(tile8, sx + 4, sy + 3, -1);
            if (e & 0x4000) s = max(s, s2);
        }
        const bool corner = s >= t;  // t >= 1, so a corner's score is positive
        if (corner) sc[sy * SC_PITCH + sx] = (uint8_t)s;
        const unsigned long long bal = __ballot(corner);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_nc, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (corner) s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
    }
    __syncthreads();
    const int nc = s_nc;
    for (int j = tid; j < nc; j += 256) {
        const int i = s_list[j], sy = i / SC_W, sx = i - sy * SC_W;
        const int x = x0 + sx - 1, y = y0 + sy - 1;
        bool kp = false;
        uint32_t s = 0u;
        if (sy >= 1 && sy <= FT_H && sx >= 1 && sx <= FT_W && x < o.cols && y < o.rows) {  // inside the tile proper (the halo only serves as neighbours)
            const uint8_t* q = sc + sy * SC_PITCH + sx;
            s = q[0];
            const uint32_t m8 = max(max(max((uint32_t)q[-SC_PITCH - 1], (uint32_t)q[-SC_PITCH]), max((uint32_t)q[-SC_PITCH + 1], (uint32_t)q[-1])),
                                    max(max((uint32_t)q[1], (uint32_t)q[SC_PITCH - 1]), max((uint32_t)q[SC_PITCH], (uint32_t)q[SC_PITCH + 1])));
            // 3x3 strict maximum, then KeyPointsFilter::runByImageBorder
            kp = s > m8 && x >= o.edge_th && x < o.cols - o.edge_th && y >= o.edge_th && y < o.rows - o.edge_th;
        }
        const unsigned long long bal = __ballot(kp);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_nkp, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (kp) s_kp[base + __popcll(bal & ((1ull << lane) - 1ull))] = ((uint32_t)y << 20) | ((uint32_t)x << 8) | s;
    }
    __syncthreads();
    const int nkp = s_nkp;
    if (nkp == 0) return;
    if (tid == 0) s_base = atomicAdd(&o.n_cand[b], nkp);  // ONE global reservation per tile
    __syncthreads();
    for (int k = tid; k < nkp; k += 256) {
        const uint32_t c = s_kp[k];
        const int slot = s_base + k;
        if (slot < o.cand_cap) o.cand[(size_t)b * o.cand_cap + slot] = c;
        atomicAdd(&o.hist[(size_t)b * 256 + (c & 255u)], 1);
    }
}

// retainBest(nfeatures) as a histogram cut (the smallest response such that at least nfeatures key-points are >= it; ties at
// the cut are all kept) + ROW-MAJOR ordering of the survivors: one workgroup per image, bitonic sort of (y, x, response) words
// in LDS.  Writes key-points and responses, n_kp (capped at K), n_total, and resets the image's counters for the next frame.
// When more key-points qualify than the LDS holds (ORD_CAP; masses of equal responses at the cut), the FIRST K of the row-major
// order are selected exactly — histogram of the qualifying key-points over image rows, then over the columns of the boundary
// row — so the output never depends on the order in which tiles appended their candidates.
constexpr int ORD_CAP = 4096;
__device__ __forceinline__ void order_block_scan(int* bins /* [4096], in place -> inclusive prefix */, int* s_part /* [1024] */) {
    const int tid = threadIdx.x;
    int v0 = bins[4 * tid], v1 = bins[4 * tid + 1], v2 = bins[4 * tid + 2], v3 = bins[4 * tid + 3];
    v1 += v0; v2 += v1; v3 += v2;
    s_part[tid] = v3;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += add;
        __syncthreads();
    }
    const int base = tid ? s_part[tid - 1] : 0;
    bins[4 * tid] = base + v0; bins[4 * tid + 1] = base + v1; bins[4 * tid + 2] = base + v2; bins[4 * tid + 3] = base + v3;
    __syncthreads();
}
__global__ __launch_bounds__(1024) void orb_order_kernel(OrbDev o) {
    __shared__ uint32_t s_key[ORD_CAP];
    __shared__ int s_part[1024];
    __shared__ int s_cut, s_n, s_acc, s_ystar, s_xstar;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int cut = 1, acc = 0;
        for (int s = 255; s >= 1; --s) {
            acc += o.hist[(size_t)b * 256 + s];
            if (acc >= o.nfeatures) {
                cut = s;
                break;
            }
        }
        s_cut = cut;
        s_acc = acc;  // key-points with response >= cut (all of them when fewer than nfeatures exist: cut stays 1)
        s_n = 0;
    }
    __syncthreads();
    const int nc = min(o.n_cand[b], o.cand_cap), cut = s_cut, n_acc = s_acc;
    const uint32_t* cand = o.cand + (size_t)b * o.cand_cap;
    uint32_t bound = 0xFFFFFFFFu;  // accept keys (y << 12 | x) <= bound
    if (n_acc > ORD_CAP) {  // block-uniform, rare: select the first K of the row-major order exactly
        const int want = min(o.K, ORD_CAP);
        int* bins = reinterpret_cast<int*>(s_key);
        for (int i = tid; i < ORD_CAP; i += 1024) bins[i] = 0;
        __syncthreads();
        for (int i = tid; i < nc; i += 1024) {
            const uint32_t c = cand[i];
            if ((int)(c & 255u) >= cut) atomicAdd(&bins[c >> 20], 1);
        }
        __syncthreads();
        order_block_scan(bins, s_part);
        for (int i = tid; i < ORD_CAP; i += 1024)  // the row in which the cumulative count reaches `want`
            if (bins[i] >= want && (i == 0 || bins[i - 1] < want)) s_ystar = i;
        __syncthreads();
        const int ystar = s_ystar;
        const int before = ystar ? bins[ystar - 1] : 0;  // key-points in the rows above
        __syncthreads();
        for (int i = tid; i < ORD_CAP; i += 1024) bins[i] = 0;
        __syncthreads();
        for (int i = tid; i < nc; i += 1024) {
            const uint32_t c = cand[i];
            if ((int)(c & 255u) >= cut && (int)(c >> 20) == ystar) atomicAdd(&bins[(c >> 8) & 0xFFFu], 1);
        }
        __syncthreads();
        order_block_scan(bins, s_part);
        const int need = want - before;  // >= 1 key-points of the boundary row, lowest columns first
        for (int i = tid; i < ORD_CAP; i += 1024)
            if (bins[i] >= need && (i == 0 || bins[i - 1] < need)) s_xstar = i;
        __syncthreads();
        bound = ((uint32_t)ystar << 12) | (uint32_t)s_xstar;
        __syncthreads();
    }
    for (int i = tid; i < ORD_CAP; i += 1024) s_key[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int i = tid; i < nc; i += 1024) {
        const uint32_t c = cand[i];
        if ((int)(c & 255u) >= cut && (c >> 8) <= bound) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < ORD_CAP) s_key[slot] = c;
        }
    }
    __syncthreads();
    const int n = min(s_n, ORD_CAP);
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = s_key[i], c = s_key[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) {
                        s_key[i] = c;
                        s_key[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    const int n_out = min(n, o.K);
    for (int i = tid; i < n_out; i += 1024) {
        const uint32_t c = s_key[i];
        const size_t k = (size_t)b * o.K + i;
        o.kp[2 * k] = (float)((c >> 8) & 0xFFFu);
        o.kp[2 * k + 1] = (float)(c >> 20);
        o.resp[k] = (float)(c & 255u);
    }
    __syncthreads();
    if (tid < 256) o.hist[(size_t)b * 256 + tid] = 0;  // ready for the next frame
    if (tid == 0) {
        o.n_kp[b] = n_out;
        if (o.n_total) o.n_total[b] = n_acc;
        o.n_cand[b] = 0;
    }
}

// Level l of the pyramid from level l - 1: OpenCV's resize(src, dst, dsize, 0, 0, INTER_LINEAR) for 8-bit images — source
// position (d + 0.5) scale - 0.5 in double -> float, weights cvRound((1 - f) 2048) / cvRound(f 2048) as shorts, horizontal pass
// in integers, rows combined as (((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2 (oracle/stvo_orb_oracle.c:
// orc_resize_linear).  One thread per output pixel; the coefficient arithmetic is a few FP ops next to four byte loads.
struct ResizeArgs {
    int B, scols, srows, dcols, drows;
    double scale_x, scale_y;  // 1 / inv_scale as OpenCV forms them: inv_scale = dsize / ssize for a given dsize, = fx, fy for Size()
    const uint8_t* src;
    uint8_t* dst;
};
__global__ __launch_bounds__(256) void orb_resize_kernel(ResizeArgs r) {
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y, b = blockIdx.z;
    if (dx >= r.dcols) return;
    float fx = (float)(((double)dx + 0.5) * r.scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) {
        fx = 0.f;
        sx = 0;
    }
    if (sx >= r.scols - 1) {
        fx = 0.f;
        sx = r.scols - 1;
    }
    const int a0 = (short)__float2int_rn((1.f - fx) * 2048.f), a1 = (short)__float2int_rn(fx * 2048.f);
    float fy = (float)(((double)dy + 0.5) * r.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int b0 = (short)__float2int_rn((1.f - fy) * 2048.f), b1 = (short)__float2int_rn(fy * 2048.f);
    const uint8_t* img = r.src + (size_t)b * r.srows * r.scols;
    const int y0 = min(max(sy, 0), r.srows - 1), y1 = min(max(sy + 1, 0), r.srows - 1);
    const uint8_t* r0 = img + (size_t)y0 * r.scols;
    const uint8_t* r1 = img + (size_t)y1 * r.scols;
    int S0, S1;
    if (sx < r.scols - 1) {
        S0 = (int)r0[sx] * a0 + (int)r0[sx + 1] * a1;
        S1 = (int)r1[sx] * a0 + (int)r1[sx + 1] * a1;
    } else {
        S0 = (int)r0[sx] * 2048;
        S1 = (int)r1[sx] * 2048;
    }
    const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    r.dst[((size_t)b * r.drows + dy) * r.dcols + dx] = (uint8_t)min(max(v, 0), 255);
}

// Key-points of all levels in level order (ORB_Impl::computeKeyPoints appends level after level): coordinates multiplied by the
// level's scale in float (pt *= scale), octave = level; one workgroup per image, capped at K.
struct ConcatArgs {
    int B, K, nlevels, Kl;  // Kl: stride of the per-level arrays
    float scale[STVO_ORB_MAX_LEVELS];
    const float* kp[STVO_ORB_MAX_LEVELS];
    const float* resp[STVO_ORB_MAX_LEVELS];
    const float* ang[STVO_ORB_MAX_LEVELS];
    const uint8_t* desc[STVO_ORB_MAX_LEVELS];
    const int32_t* n[STVO_ORB_MAX_LEVELS];
    const int32_t* n_tot[STVO_ORB_MAX_LEVELS];
    float* kp_o;
    float* resp_o;
    float* ang_o;
    int32_t* oct_o;  // may be nullptr
    uint8_t* desc_o;
    int32_t* n_o;
    int32_t* n_total_o;  // may be nullptr
};
__global__ __launch_bounds__(256) void orb_concat_kernel(ConcatArgs c) {
    const int b = blockIdx.x, tid = threadIdx.x;
    int off = 0, total = 0;
    for (int l = 0; l < c.nlevels; ++l) {
        const int nl = c.n[l][b];
        total += c.n_tot[l][b];
        const float sc = c.scale[l];
        for (int i = tid; i < nl && off + i < c.K; i += 256) {
            const size_t s = (size_t)b * c.Kl + i, d = (size_t)b * c.K + off + i;
            const float x = c.kp[l][2 * s], y = c.kp[l][2 * s + 1];
            c.kp_o[2 * d] = l ? x * sc : x;
            c.kp_o[2 * d + 1] = l ? y * sc : y;
            c.resp_o[d] = c.resp[l][s];
            c.ang_o[d] = c.ang[l][s];
            if (c.oct_o) c.oct_o[d] = l;
            const uint4* src = reinterpret_cast<const uint4*>(c.desc[l] + s * 32);
            uint4* dst = reinterpret_cast<uint4*>(c.desc_o + d * 32);
            dst[0] = src[0];
            dst[1] = src[1];
        }
        off += nl;
    }
    if (tid == 0) {
        c.n_o[b] = min(off, c.K);
        if (c.n_total_o) c.n_total_o[b] = total;
    }
}

__device__ __forceinline__ int reflect101(int p, int n) {
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

struct BlurK {
    int k[7];
};

// Separable 7 x 7 blur in registers, no LDS: a thread produces 4 adjacent pixels (one packed word) of BL_R consecutive rows.
// Per input row it loads the 12 bytes x - 4 .. x + 7 as three (unaligned) words, forms the four horizontal sums with byte
// alignment + 4-way byte dot products (v_alignbyte_b32 / v_dot4_u32_u8: 16 instructions instead of 12 extractions + 28
// multiply-adds), keeps the last seven rows of sums in a register ring and emits one output row per input row.  The arithmetic
// is the integer one of the tile version (weights * 2^8 per pass, (s + 2^15) >> 16), so the sums may be taken in any order.
constexpr int BL_R = 16, BL_T = 64;

__global__ __launch_bounds__(BL_T) void orb_blur_kernel(OrbDev o, BlurK kk) {
    const int b = blockIdx.z, x = (blockIdx.x * BL_T + threadIdx.x) * 4, y0 = blockIdx.y * BL_R;
    if (x >= o.cols) return;
    const uint8_t* img = o.img + (size_t)b * o.rows * o.cols;
    uint8_t* out = o.blur + (size_t)b * o.rows * o.cols;
    const uint32_t k03 = (uint32_t)kk.k[0] | ((uint32_t)kk.k[1] << 8) | ((uint32_t)kk.k[2] << 16) | ((uint32_t)kk.k[3] << 24);
    const uint32_t k46 = (uint32_t)kk.k[4] | ((uint32_t)kk.k[5] << 8) | ((uint32_t)kk.k[6] << 16);
    // The three words of a row — bytes x - 4 .. x - 1, x .. x + 3, x + 4 .. x + 7 under BORDER_REFLECT_101 — as ONE word load + ONE v_perm each,
    // for every thread alike: four consecutive positions reflect into a window of at most four bytes, so each word is a word of the row (at
    // a per-thread offset) with its bytes permuted (a per-thread selector; the identity away from the border).  Offsets and selectors are
    // computed once, not per row.  (Round 6: the border threads used to take a byte-by-byte path — 12 byte loads and 12 reflections per
    // row, inlined 22 times — and dragged their whole wave through it: two of the five waves of an image row.)
    int woff[3];
    uint32_t wsel[3];
#pragma unroll
    for (int wi = 0; wi < 3; ++wi) {
        int q[4], lo = 0x7FFFFFFF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q[i] = reflect101(x - 4 + 4 * wi + i, o.cols);
            lo = min(lo, q[i]);
        }
        lo = max(min(lo, o.cols - 4), 0);
        woff[wi] = lo;
        wsel[wi] = (uint32_t)(q[0] - lo) | ((uint32_t)(q[1] - lo) << 8) | ((uint32_t)(q[2] - lo) << 16) | ((uint32_t)(q[3] - lo) << 24);
    }
    uint32_t h[7][4];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) h[j][i] = 0u;
#pragma unroll
    for (int r = 0; r < BL_R + 6; ++r) {
        const int yo = y0 + r - 6;                  // output row completed by this input row
        if (r >= 6 && yo >= o.rows) break;          // uniform over the workgroup
        const int gy = reflect101(y0 + r - 3, o.rows);
        const uint8_t* row = img + (size_t)gy * o.cols;
        const uint32_t w0 = __builtin_amdgcn_perm(0u, *reinterpret_cast<const u32_unaligned*>(row + woff[0]), wsel[0]);
        const uint32_t w1 = __builtin_amdgcn_perm(0u, *reinterpret_cast<const u32_unaligned*>(row + woff[1]), wsel[1]);
        const uint32_t w2 = __builtin_amdgcn_perm(0u, *reinterpret_cast<const u32_unaligned*>(row + woff[2]), wsel[2]);
        // output i: taps x + i - 3 .. x + i + 3 = bytes i + 1 .. i + 7 of (w0, w1, w2)
        uint32_t hs[4];
        hs[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), k03, 0u, false);
        hs[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), k46, hs[0], false);
        hs[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), k03, 0u, false);
        hs[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), k46, hs[1], false);
        hs[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), k03, 0u, false);
        hs[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), k46, hs[2], false);
        hs[3] = __builtin_amdgcn_udot4(w1, k03, 0u, false);
        hs[3] = __builtin_amdgcn_udot4(w2, k46, hs[3], false);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) h[j][i] = h[j + 1][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) h[6][i] = hs[i];
        if (r >= 6) {
            uint32_t px = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t sv = 1u << 15;
#pragma unroll
                for (int j = 0; j < 7; ++j) sv += __umul24((uint32_t)kk.k[j], h[j][i]);  // (weights <= 2^8, row sums <= 2^16: the 24-bit multiply-add is exact and full rate)
                px |= min(sv >> 16, 255u) << (8 * i);
            }
            uint8_t* dst = out + (size_t)yo * o.cols + x;
            if (x + 4 <= o.cols) {
                *reinterpret_cast<u32_unaligned*>(dst) = px;
            } else {
                for (int i = 0; x + i < o.cols; ++i) dst[i] = (uint8_t)(px >> (8 * i));
            }
        }
    }
}

// sine and cosine of a float angle in [0, 2 pi] in double precision (the routine of lsd_kernels.hip / oracle/stvo_lsd_oracle.c)
__device__ __forceinline__ void orb_sincos(float xf, double& s, double& c) {
    const double x = (double)xf;
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17, TWO_OVER_PI = 6.36619772367581382433e-01;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double k = __builtin_rint(x * TWO_OVER_PI);
    double r = __builtin_fma(-k, PIO2_HI, x);
    r = __builtin_fma(-k, PIO2_LO, r);
    const double z = r * r;
    const double sn = r + (z * r) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
    const double cs = 1.0 - (0.5 * z - z * (z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))));
    const int q = (int)k & 3;
    s = q == 0 ? sn : (q == 1 ? cs : (q == 2 ? -sn : -cs));
    c = q == 0 ? cs : (q == 1 ? -sn : (q == 2 ? -cs : sn));
}

struct Umax {
    int u[ORB_HP + 2];
};

// 16 lanes per key-point (4 key-points per wave, 16 per workgroup): intensity-centroid angle on the image, rotated BRIEF on the
// blurred image.  Both patches are first copied into LDS with word loads (a row of a patch is 32 / 40 contiguous bytes, one
// cache line per key-point and load instruction); the 512 rotated test points of a key-point are then gathered from LDS.
// Gathered from global memory, every test load of a wave touched up to 64 different cache lines — the texture-address path,
// not arithmetic, set the pace of the first two versions (675 / 450 us per 512 k key-points).  The sine / cosine of the angle
// (double precision, like the oracle's libm call) runs once per wave: four key-points share it.
constexpr int DESC_KP_PER_WG = 16;
constexpr int DESC_R = 19;                        // |rotated pattern coordinate| <= 13 sqrt 2 < 19
constexpr int DESC_PW = 10, DESC_PH = 2 * DESC_R + 1;  // blurred patch: 39 rows of 10 words = bytes x - 20 .. x + 19
constexpr int IC_PW = 8, IC_PH = 2 * ORB_HP + 1;       // image patch: 31 rows of 8 words = bytes x - 16 .. x + 15
__global__ __launch_bounds__(256) void orb_describe_kernel(OrbDev o, Umax um) {
    constexpr int PATCH_W = ((DESC_PH * DESC_PW + 15) / 16) * 16;  // rounded up to the 16 lanes of a group: every lane stores every word it loaded
    __shared__ uint32_t s_patch[DESC_KP_PER_WG][PATCH_W];  // the image patch first (31 x 8 words), then the blurred one
    // XCD-aware 1-D grid: workgroup L runs on XCD L mod 8 (round-robin dispatch), so image = 8 (L / 8 / blocks per image) + L mod 8
    // keeps every workgroup of an image on ONE XCD — the patches of neighbouring key-points overlap (2000 key-points read 11 x the
    // image), and with the image's workgroups dealt over all eight private L2s each of them fetched the image from HBM again
    // (FETCH_SIZE 6.6 x the image bytes in round 3)
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int bpi = (o.K + DESC_KP_PER_WG - 1) / DESC_KP_PER_WG;  // workgroups per image
    const int b = (kq / bpi) * 8 + xcd, kb = kq % bpi;
    if (b >= o.B) return;
    const int lane = threadIdx.x & 63, l = lane & 15;
    // the circular patch as byte masks of the row words: row |dy| keeps columns |u| <= umax[|dy|] (byte k of word w is column 4 w + k - 16)
    __shared__ uint32_t s_mask[ORB_HP + 1][IC_PW];
    if (threadIdx.x < (ORB_HP + 1) * IC_PW) {
        const int v = threadIdx.x >> 3, w = threadIdx.x & 7;
        uint32_t m = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = 4 * w + k - 16;
            if ((u < 0 ? -u : u) <= um.u[v]) m |= 0xFFu << (8 * k);
        }
        s_mask[v][w] = m;
    }
    __syncthreads();
    const int n = o.n_kp[b];
    const int k_first = kb * DESC_KP_PER_WG + (threadIdx.x >> 6) * 4;
    if (k_first >= n) return;  // wave-uniform
    const int slot = (threadIdx.x >> 6) * 4 + (lane >> 4);
    const int k_own = k_first + (lane >> 4);
    const bool valid = k_own < n;
    const int k = valid ? k_own : n - 1;  // idle groups shadow the last key-point (the shuffles below want every lane)
    const size_t kk = (size_t)b * o.K + k;
    const int x = (int)o.kp[2 * kk], y = (int)o.kp[2 * kk + 1];
    const size_t base = (size_t)b * o.rows * o.cols;
    uint32_t* patch = s_patch[slot];
    const uint8_t* patch8 = reinterpret_cast<const uint8_t*>(patch);
    // the patches lie inside the image (edge threshold >= 19 + 1 word of slack is NOT guaranteed on the left / right: clamp the
    // word address into the image buffer; the clamped words hold columns the disc / the pattern never reads)
    // (round 6: 32-bit offsets into the image, clamped with one v_med3 — the clamped POINTERS cost two 64-bit compares and four selects per word,
    //  a quarter of the kernel's vector instructions)
    const uint8_t* img_b = o.img + base;
    const int off_hi = o.rows * o.cols - 4;
    {
        // all 16 words of a lane's share are requested before the first is stored (as a loop the compiler made four groups of four
        // loads with a wait after each: four memory round trips one after the other at the start of every wave)
        constexpr int IW_N = (IC_PH * IC_PW + 15) / 16;
        uint32_t iw[IW_N];
        const int org = (y - ORB_HP) * o.cols + (x - 16);
#pragma unroll
        for (int j = 0; j < IW_N; ++j) {
            const int i = l + 16 * j;
            const int r = i >> 3, w = i & 7;
            const int off = min(max(org + (int)__umul24((unsigned)r, (unsigned)o.cols) + 4 * w, 0), off_hi);  // (i beyond the patch: a clamped, unused word)
            iw[j] = *reinterpret_cast<const u32_unaligned*>(img_b + off);
        }
#pragma unroll
        for (int j = 0; j < IW_N; ++j) patch[l + 16 * j] = iw[j];  // (unconditional: a store under `i < 248` took its load along, into a round trip of its own)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the blurred patch is requested NOW, into registers (25 words per lane), and stored to LDS behind the moments: its memory round
    // trip runs beside the computation instead of after it
    constexpr int BW_N = (DESC_PH * DESC_PW + 15) / 16;
    uint32_t bw[BW_N];
    {
        const uint8_t* blur_b = o.blur + base;
        const int org = (y - DESC_R) * o.cols + (x - 20);
#pragma unroll
        for (int j = 0; j < BW_N; ++j) {
            const int i = l + 16 * j;
            const int r = i / DESC_PW, w = i - r * DESC_PW;
            const int off = min(max(org + (int)__umul24((unsigned)r, (unsigned)o.cols) + 4 * w, 0), off_hi);  // (i beyond the patch: a clamped, unused word)
            bw[j] = *reinterpret_cast<const u32_unaligned*>(blur_b + off);
        }
    }
    // ICAngles: lane l owns ROWS l and l + 16 of the patch (dy = row - 15).  Integer moments, so the summation order is free: a row is 8
    // words — masked to the disc, then m10 += sum u I and the row sum for m01 come from byte dot products (v_dot4_u32_u8: weights u + 16
    // and 1) instead of 31 byte loads per column (~70 instead of ~300 instructions per lane)
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = l + 16 * h;
        const bool row_ok = r < IC_PH;
        const int dy = r - ORB_HP, ady = dy < 0 ? -dy : dy;
        const uint32_t* row = patch + (row_ok ? r : 0) * IC_PW;
        const uint32_t* msk = s_mask[row_ok ? ady : 0];
        uint32_t s1 = 0u, su = 0u;
#pragma unroll
        for (int w = 0; w < IC_PW; ++w) {
            const uint32_t W = row[w] & msk[w];
            s1 = __builtin_amdgcn_udot4(W, 0x01010101u, s1, false);
            su = __builtin_amdgcn_udot4(W, 0x03020100u + 0x04040404u * (uint32_t)w, su, false);
        }
        if (row_ok) {
            m10 += (int)su - 16 * (int)s1;
            m01 += dy * (int)s1;
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        m10 += __shfl_xor(m10, off, 64);
        m01 += __shfl_xor(m01, off, 64);
    }
    const float ang = fast_atan2_deg((float)m01, (float)m10);
    if (valid && l == 0) o.angle[kk] = ang;
    // the blurred patch replaces the image patch (every lane of the group is past its reads: the shuffles above synchronise)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < BW_N; ++j) patch[l + 16 * j] = bw[j];  // (unconditional: the row is padded to 16 x BW_N words)
    // computeOrbDescriptors, WTA_K = 2: lane l evaluates tests 16 l .. 16 l + 15 = bytes 2 l, 2 l + 1 of the descriptor
    const float rad = ang * (float)(3.14159265358979323846 / 180.0);
    // (float)cos((double)rad), (float)sin((double)rad): one Cody-Waite + kernel-polynomial evaluation (within an ulp of the library's
    // double results, i.e. the same floats except on rounding ties) instead of two library calls of ~100 instructions each
    double sd, cd;
    orb_sincos(rad, sd, cd);
    const float a = (float)cd, sb = (float)sd;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint8_t* bl = patch8 + DESC_R * (DESC_PW * 4) + 20;  // the key-point's own pixel
    const int* pat = reinterpret_cast<const int*>(o.pattern) + 16 * l;  // (x0, y0, x1, y1) of a test as one word
    uint32_t bits = 0u;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int w = pat[t];
        const float px0 = (float)(int8_t)(w & 0xFF), py0 = (float)(int8_t)((w >> 8) & 0xFF);
        const float px1 = (float)(int8_t)((w >> 16) & 0xFF), py1 = (float)(int8_t)((w >> 24) & 0xFF);
        const int ix0 = __float2int_rn(px0 * a - py0 * sb), iy0 = __float2int_rn(px0 * sb + py0 * a);
        const int ix1 = __float2int_rn(px1 * a - py1 * sb), iy1 = __float2int_rn(px1 * sb + py1 * a);
        const int t0 = bl[iy0 * (DESC_PW * 4) + ix0], t1 = bl[iy1 * (DESC_PW * 4) + ix1];
        bits |= (uint32_t)(t0 < t1) << t;
    }
    if (valid) *reinterpret_cast<uint16_t*>(o.desc + kk * 32 + 2 * l) = (uint16_t)bits;
}

}  // namespace

// the two 8-bit image operations other front-ends share (lsd_kernels.hip: the scaled image of the line detector)
void launch_blur7_u8(hipStream_t s, int B, int cols, int rows, const uint8_t* src, uint8_t* dst, const int* k7) {
    OrbDev o{};
    o.B = B; o.cols = cols; o.rows = rows; o.img = src; o.blur = dst;
    BlurK kk{};
    for (int i = 0; i < 7; ++i) kk.k[i] = k7[i];
    const dim3 tiles((cols + 4 * BL_T - 1) / (4 * BL_T), (rows + BL_R - 1) / BL_R, B);
    hipLaunchKernelGGL(orb_blur_kernel, tiles, dim3(BL_T), 0, s, o, kk);
}
void launch_resize_linear_u8(hipStream_t s, int B, int scols, int srows, int dcols, int drows, const uint8_t* src, uint8_t* dst, double fx, double fy) {
    ResizeArgs r{};
    r.B = B; r.scols = scols; r.srows = srows; r.dcols = dcols; r.drows = drows;
    // fx, fy > 0: resize(src, dst, Size(), fx, fy) — cv::resize keeps inv_scale = fx and rounds only the output size; 0: a given dsize
    r.scale_x = 1.0 / (fx > 0 ? fx : (double)dcols / scols);
    r.scale_y = 1.0 / (fy > 0 ? fy : (double)drows / srows);
    r.src = src; r.dst = dst;
    hipLaunchKernelGGL(orb_resize_kernel, dim3((dcols + 255) / 256, drows, B), dim3(256), 0, s, r);
}
}  // namespace stvo

struct stvo_orb {
    stvo_ctx* ctx = nullptr;
    int B = 0, K = 0, nlevels = 1;
    stvo_orb_params prm{};
    struct Level {
        stvo::OrbDev d{};      // geometry, budget and scratch of the level (img / outputs are set per call)
        float scale = 1.f;
        uint8_t* img = nullptr;  // the resized image of the level (levels > 0)
        // per-level outputs when there is more than one level (level coordinates; orb_concat_kernel assembles the result)
        float *kp = nullptr, *resp = nullptr, *ang = nullptr;
        uint8_t* desc = nullptr;
        int32_t *n = nullptr, *n_tot = nullptr;
        bool active = true;    // a level without budget, or smaller than the border, yields nothing
    } lev[STVO_ORB_MAX_LEVELS];
    char* dev = nullptr;  // every device array of the object, carved from one allocation
    char* io = nullptr;   // staging for the host-buffer entry point: images in, results out
    size_t io_bytes = 0;
    int8_t pattern[1024];
    int8_t* d_pattern = nullptr;
    stvo::BlurK blur_k{};
    stvo::Umax umax{};
};

namespace {

void default_pattern(int8_t* pattern) {  // seeded stand-in for OpenCV's learned table (see stvo_orb_set_pattern)
    uint32_t s = 31u;
    for (int i = 0; i < 256; ++i) {
        int v[4];
        do {
            for (int j = 0; j < 4; ++j) {
                s = s * 1664525u + 1013904223u;
                v[j] = (int)((s >> 16) % 27u) - 13;
            }
        } while (v[0] == v[2] && v[1] == v[3]);
        for (int j = 0; j < 4; ++j) pattern[4 * i + j] = (int8_t)v[j];
    }
}

int cv_round_f(float v) { return (int)std::lrintf(v); }  // cvRound: half to even

}  // namespace

extern "C" {

int stvo_orb_create(stvo_ctx* ctx, int B, int cols, int rows, int max_keypoints, const stvo_orb_params* prm, stvo_orb** out) {
    if (!ctx || !out || !prm || B <= 0 || cols < 64 || rows < 64 || max_keypoints <= 0) return STVO_ERR_INVALID_ARG;
    // the patch (radius 15) and the rotated pattern (|coordinate| <= 13 sqrt 2) must stay inside the image: edge >= 19
    if (prm->nfeatures <= 0 || prm->fast_threshold < 1 || prm->fast_threshold > 254 || prm->edge_threshold < 19 ||
        2 * prm->edge_threshold >= cols || 2 * prm->edge_threshold >= rows)
        return STVO_ERR_INVALID_ARG;
    const int nlevels = prm->nlevels <= 0 ? 1 : prm->nlevels;
    if (nlevels > STVO_ORB_MAX_LEVELS || (nlevels > 1 && !(prm->scale_factor > 1.0 && prm->scale_factor <= 2.0))) return STVO_ERR_INVALID_ARG;
    if (rows >= 4096 || cols >= 4096) return STVO_ERR_CAPACITY;  // candidates pack (y, x) in 12 bits each
    // the ordering kernel emits at most ORD_CAP key-points per level (its LDS sort): an output capacity beyond it cannot be honoured.
    // (nfeatures may be anything: when more key-points qualify than max_keypoints, the first max_keypoints of the row-major order
    // are selected exactly and n_total reports how many qualified)
    if (max_keypoints > stvo::ORD_CAP) return STVO_ERR_CAPACITY;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stvo_orb* o = new (std::nothrow) stvo_orb();
    if (!o) return STVO_ERR_HIP;
    o->ctx = ctx;
    o->prm = *prm;
    o->B = B; o->K = max_keypoints; o->nlevels = nlevels;
    // level geometry and feature budgets exactly as ORB_Impl forms them (oracle/stvo_orb_oracle.c: orc_orb_levels)
    int nf[STVO_ORB_MAX_LEVELS];
    {
        const float factor = (float)(1.0 / (nlevels > 1 ? prm->scale_factor : 1.2));
        float nd = (float)prm->nfeatures * (1.f - factor) / (1.f - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int l = 0; l < nlevels - 1; ++l) {
            nf[l] = cv_round_f(nd);
            sum += nf[l];
            nd *= factor;
        }
        nf[nlevels - 1] = prm->nfeatures - sum > 0 ? prm->nfeatures - sum : 0;
        if (nlevels == 1) nf[0] = prm->nfeatures;
    }
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    size_t total = 1024;  // the pattern
    size_t off_blur[STVO_ORB_MAX_LEVELS], off_hist[STVO_ORB_MAX_LEVELS], off_cand[STVO_ORB_MAX_LEVELS], off_nc[STVO_ORB_MAX_LEVELS],
        off_img[STVO_ORB_MAX_LEVELS], off_out[STVO_ORB_MAX_LEVELS];
    for (int l = 0; l < nlevels; ++l) {
        stvo_orb::Level& L = o->lev[l];
        L.scale = l ? (float)std::pow(prm->scale_factor, (double)l) : 1.f;
        stvo::OrbDev& d = L.d;
        d.B = B; d.K = max_keypoints;
        d.cols = l ? cv_round_f((float)cols / L.scale) : cols;
        d.rows = l ? cv_round_f((float)rows / L.scale) : rows;
        d.nfeatures = nf[l]; d.fast_th = prm->fast_threshold; d.edge_th = prm->edge_threshold;
        d.cand_cap = d.rows * d.cols / 4 + 64;
        L.active = nf[l] > 0 && 2 * prm->edge_threshold < d.cols && 2 * prm->edge_threshold < d.rows && d.cols >= 16 && d.rows >= 16;
        const size_t px = (size_t)B * d.rows * d.cols, nk = (size_t)B * max_keypoints;
        off_blur[l] = total; total += al(px);
        off_hist[l] = total; total += al((size_t)B * 256 * 4);
        off_cand[l] = total; total += al((size_t)B * d.cand_cap * 4);
        off_nc[l] = total; total += al((size_t)B * 4);
        off_img[l] = total; if (l) total += al(px);
        off_out[l] = total; if (nlevels > 1) total += al(nk * 8) + 2 * al(nk * 4) + al(nk * 32) + 2 * al((size_t)B * 4);
    }
    if (!hip_ok(ctx, hipMalloc((void**)&o->dev, total), "hipMalloc orb") || !zero_device(ctx, o->dev, total, "hipMemset orb")) {
        if (o->dev) (void)hipFree(o->dev);
        delete o;
        return STVO_ERR_HIP;
    }
    o->d_pattern = (int8_t*)o->dev;
    for (int l = 0; l < nlevels; ++l) {
        stvo_orb::Level& L = o->lev[l];
        stvo::OrbDev& d = L.d;
        const size_t nk = (size_t)B * max_keypoints;
        d.blur = (uint8_t*)(o->dev + off_blur[l]);
        d.hist = (int32_t*)(o->dev + off_hist[l]); d.cand = (uint32_t*)(o->dev + off_cand[l]); d.n_cand = (int32_t*)(o->dev + off_nc[l]);
        d.pattern = o->d_pattern;
        if (l) L.img = (uint8_t*)(o->dev + off_img[l]);
        if (nlevels > 1) {
            char* q = o->dev + off_out[l];
            L.kp = (float*)q; q += al(nk * 8);
            L.resp = (float*)q; q += al(nk * 4);
            L.ang = (float*)q; q += al(nk * 4);
            L.desc = (uint8_t*)q; q += al(nk * 32);
            L.n = (int32_t*)q; q += al((size_t)B * 4);
            L.n_tot = (int32_t*)q;
        }
    }
    default_pattern(o->pattern);
    if (!upload_now(ctx, o->d_pattern, o->pattern, 1024, "hipMemcpy pattern")) {
        (void)hipFree(o->dev);
        delete o;
        return STVO_ERR_HIP;
    }
    {   // getGaussianKernel(7, 2) in 8-bit fixed point; umax of ICAngles
        double k[7], sum = 0.0;
        for (int i = 0; i < 7; ++i) {
            const double x = i - 3;
            k[i] = std::exp(-x * x / (2.0 * 2.0 * 2.0));
            sum += k[i];
        }
        for (int i = 0; i < 7; ++i) o->blur_k.k[i] = (int)std::lrint((float)(k[i] / sum) * 256.0);
        const int hp = stvo::ORB_HP;
        const int vmax = (int)std::floor(hp * std::sqrt(2.0) / 2 + 1), vmin = (int)std::ceil(hp * std::sqrt(2.0) / 2);
        for (int v = 0; v <= vmax; ++v) o->umax.u[v] = (int)std::lrint(std::sqrt((double)hp * hp - (double)v * v));
        for (int v = hp, v0 = 0; v >= vmin; --v) {
            while (o->umax.u[v0] == o->umax.u[v0 + 1]) ++v0;
            o->umax.u[v] = v0;
            ++v0;
        }
    }
    *out = o;
    return STVO_OK;
}

int stvo_orb_destroy(stvo_orb* o) {
    if (!o) return STVO_OK;
    (void)hipSetDevice(o->ctx->device);
    (void)hipStreamSynchronize(o->ctx->stream);
    if (o->dev) (void)hipFree(o->dev);
    if (o->io) (void)hipFree(o->io);
    delete o;
    return STVO_OK;
}

int stvo_orb_set_pattern(stvo_orb* o, const int8_t* pattern) {
    if (!o || !pattern) return STVO_ERR_INVALID_ARG;
    for (int i = 0; i < 1024; ++i)
        if (pattern[i] < -13 || pattern[i] > 13) return STVO_ERR_INVALID_ARG;  // rotated points must stay within the border
    HIP_TRY(o->ctx, hipSetDevice(o->ctx->device));
    HIP_TRY(o->ctx, hipStreamSynchronize(o->ctx->stream));
    std::memcpy(o->pattern, pattern, 1024);
    if (!upload_now(o->ctx, o->d_pattern, o->pattern, 1024, "hipMemcpy pattern")) return STVO_ERR_HIP;
    return STVO_OK;
}

int stvo_orb_set_fast_threshold(stvo_orb* o, int fast_threshold) {
    if (!o || fast_threshold < 1 || fast_threshold > 254) return STVO_ERR_INVALID_ARG;
    o->prm.fast_threshold = fast_threshold;
    for (int l = 0; l < o->nlevels; ++l) o->lev[l].d.fast_th = fast_threshold;  // kernel argument of the next launches
    return STVO_OK;
}

int stvo_orb_get_pattern(const stvo_orb* o, int8_t* pattern) {
    if (!o || !pattern) return STVO_ERR_INVALID_ARG;
    std::memcpy(pattern, o->pattern, 1024);
    return STVO_OK;
}

int stvo_orb_detect_levels_dev(stvo_orb* o, const uint8_t* images, float* kp_xy, float* response, float* angle, int32_t* octave,
                               uint8_t* desc, int32_t* n_kp, int32_t* n_total) {
    if (!o || !images || !kp_xy || !response || !angle || !desc || !n_kp) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const bool multi = o->nlevels > 1;
    const uint8_t* prev = images;
    for (int l = 0; l < o->nlevels; ++l) {
        stvo_orb::Level& L = o->lev[l];
        stvo::OrbDev d = L.d;
        if (l) {  // resize(level l - 1): needed by the next level even when this one yields nothing
            const stvo::OrbDev& p = o->lev[l - 1].d;
            stvo::ResizeArgs r{d.B, p.cols, p.rows, d.cols, d.rows, 1.0 / ((double)d.cols / p.cols), 1.0 / ((double)d.rows / p.rows), prev, L.img};
            hipLaunchKernelGGL(stvo::orb_resize_kernel, dim3((d.cols + 255) / 256, d.rows, d.B), dim3(256), 0, s, r);
            d.img = L.img;
        } else {
            d.img = images;
        }
        prev = d.img;
        if (multi) {
            d.kp = L.kp; d.resp = L.resp; d.angle = L.ang; d.desc = L.desc; d.n_kp = L.n; d.n_total = L.n_tot;
        } else {
            d.kp = kp_xy; d.resp = response; d.angle = angle; d.desc = desc; d.n_kp = n_kp; d.n_total = n_total;
        }
        if (!L.active) {
            HIP_TRY(ctx, hipMemsetAsync(d.n_kp, 0, (size_t)d.B * 4, s));
            if (d.n_total) HIP_TRY(ctx, hipMemsetAsync(d.n_total, 0, (size_t)d.B * 4, s));
            continue;
        }
        const dim3 tiles((d.cols + 4 * stvo::BL_T - 1) / (4 * stvo::BL_T), (d.rows + stvo::BL_R - 1) / stvo::BL_R, d.B), tb(stvo::BL_T);
        hipLaunchKernelGGL(stvo::orb_fast_nms_kernel, dim3((d.cols + stvo::FT_W - 1) / stvo::FT_W, (d.rows + stvo::FT_H - 1) / stvo::FT_H, d.B),
                           dim3(256), 0, s, d);
        hipLaunchKernelGGL(stvo::orb_blur_kernel, tiles, tb, 0, s, d, o->blur_k);
        hipLaunchKernelGGL(stvo::orb_order_kernel, dim3(d.B), dim3(1024), 0, s, d);
        hipLaunchKernelGGL(stvo::orb_describe_kernel, dim3((unsigned)(((d.K + stvo::DESC_KP_PER_WG - 1) / stvo::DESC_KP_PER_WG) * ((d.B + 7) / 8) * 8)), dim3(256),
                           0, s, d, o->umax);
    }
    if (multi) {
        stvo::ConcatArgs c{};
        c.B = o->B; c.K = o->K; c.nlevels = o->nlevels; c.Kl = o->K;
        for (int l = 0; l < o->nlevels; ++l) {
            const stvo_orb::Level& L = o->lev[l];
            c.scale[l] = L.scale; c.kp[l] = L.kp; c.resp[l] = L.resp; c.ang[l] = L.ang; c.desc[l] = L.desc; c.n[l] = L.n; c.n_tot[l] = L.n_tot;
        }
        c.kp_o = kp_xy; c.resp_o = response; c.ang_o = angle; c.oct_o = octave; c.desc_o = desc; c.n_o = n_kp; c.n_total_o = n_total;
        hipLaunchKernelGGL(stvo::orb_concat_kernel, dim3(o->B), dim3(256), 0, s, c);
    } else if (octave) {
        HIP_TRY(ctx, hipMemsetAsync(octave, 0, (size_t)o->B * o->K * 4, s));
    }
    return check_launch(ctx);
}

int stvo_orb_detect_dev(stvo_orb* o, const uint8_t* images, float* kp_xy, float* response, float* angle, uint8_t* desc,
                        int32_t* n_kp) {
    return stvo_orb_detect_levels_dev(o, images, kp_xy, response, angle, nullptr, desc, n_kp, nullptr);
}

int stvo_orb_detect_levels(stvo_orb* o, const uint8_t* images, float* kp_xy, float* response, float* angle, int32_t* octave, uint8_t* desc,
                           int32_t* n_kp, int32_t* n_total) {
    if (!o || !images || !kp_xy || !response || !angle || !desc || !n_kp) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const stvo::OrbDev& d = o->lev[0].d;
    const size_t px = (size_t)d.B * d.rows * d.cols, nk = (size_t)d.B * d.K;
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_img = 0, o_kp = al(px), o_resp = o_kp + al(nk * 8), o_ang = o_resp + al(nk * 4), o_oct = o_ang + al(nk * 4),
                 o_desc = o_oct + al(nk * 4), o_n = o_desc + al(nk * 32), o_nt = o_n + al((size_t)d.B * 4), total = o_nt + al((size_t)d.B * 4);
    if (o->io_bytes < total) {
        if (o->io) (void)hipFree(o->io);
        o->io = nullptr;
        o->io_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&o->io, total));
        o->io_bytes = total;
    }
    char* D = o->io;
    HIP_TRY(ctx, hipMemcpyAsync(D + o_img, images, px, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(D + o_kp, 0, total - o_kp, ctx->stream));
    TRY(stvo_orb_detect_levels_dev(o, (const uint8_t*)(D + o_img), (float*)(D + o_kp), (float*)(D + o_resp), (float*)(D + o_ang),
                                   (int32_t*)(D + o_oct), (uint8_t*)(D + o_desc), (int32_t*)(D + o_n), (int32_t*)(D + o_nt)));
    HIP_TRY(ctx, hipMemcpyAsync(kp_xy, D + o_kp, nk * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(response, D + o_resp, nk * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(angle, D + o_ang, nk * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (octave) HIP_TRY(ctx, hipMemcpyAsync(octave, D + o_oct, nk * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(desc, D + o_desc, nk * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(n_kp, D + o_n, (size_t)d.B * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (n_total) HIP_TRY(ctx, hipMemcpyAsync(n_total, D + o_nt, (size_t)d.B * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return STVO_OK;
}

int stvo_orb_detect(stvo_orb* o, const uint8_t* images, float* kp_xy, float* response, float* angle, uint8_t* desc, int32_t* n_kp) {
    return stvo_orb_detect_levels(o, images, kp_xy, response, angle, nullptr, desc, n_kp, nullptr);
}

}  // extern "C"
