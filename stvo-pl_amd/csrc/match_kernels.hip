// match_kernels.hip — K1 `hamming_knn2` and K2 `nnr_mutual` for gfx950 (CDNA4, wave64).
//
// K1 replaces cv::BFMatcher(NORM_HAMMING).knnMatch(desc1, desc2, ., 2) as called from
// StVO::matchNNR (/root/reference/src/matching.cpp:47-48), in both directions (:69-74).
// K2 replaces the float ratio test (:53-58) and the mutual check of StVO::match (:80-86).
//
// K1 mapping (integer-VALU bound):
//   * lane  = one QUERY row: its 256-bit descriptor lives in 8 VGPRs for the whole scan
//             (two coalesced 16-byte loads per lane, 2 KiB contiguous per wave);
//   * train rows are wave-uniform: fetched through the SCALAR cache (s_load_dwordx8, 32 B/row),
//     so the inner loop issues no vector-memory or LDS instructions at all;
//   * per (query, train) pair: 8 x v_xor_b32 + 8 x v_bcnt_u32_b32 (popcount-accumulate),
//     then 3 VALU ops of top-2 bookkeeping on a packed key (distance << 16 | train index):
//     v_lshl_or_b32, v_med3_u32 (new second = median(best, second, key)), v_min_u32.
//     The packed key makes `min` pick the LOWEST train index among equal distances, which is
//     knnMatch's tie order (strict '<' insertion over ascending train index).
//   * grid = (query tiles of 256 rows, 2 directions, B frame pairs); no inter-workgroup traffic.
#include "kernels.h"

#include <stdlib.h>

namespace stvo {

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// popcount-accumulate: D = popcount(x) + acc in ONE VALU op.  Written as inline asm because the
// compiler otherwise re-associates the 8-term sum into bcnt(x,0) + v_add3 trees (22 instead of
// 19 VALU ops per pair).
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
// key = (d << 16) | j with the wave-uniform train index taken from an SGPR (one VALU op; the
// compiler's own lowering of (d << 16) | (j + u) is shift + v_or3 = two).
__device__ __forceinline__ uint32_t pack_key(uint32_t d, uint32_t j_uniform) {
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(r) : "v"(d), "s"(j_uniform));
    return r;
}
__device__ __forceinline__ uint32_t bcnt0(uint32_t x) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}

// t is wave-uniform -> 8 SGPRs feeding v_xor_b32 directly.
__device__ __forceinline__ uint32_t hamming256(const uint4& q0, const uint4& q1, const uint32_t* __restrict__ t) {
    uint32_t d = bcnt0(q0.x ^ t[0]);
    d = bcnt_acc(q0.y ^ t[1], d);
    d = bcnt_acc(q0.z ^ t[2], d);
    d = bcnt_acc(q0.w ^ t[3], d);
    d = bcnt_acc(q1.x ^ t[4], d);
    d = bcnt_acc(q1.y ^ t[5], d);
    d = bcnt_acc(q1.z ^ t[6], d);
    d = bcnt_acc(q1.w ^ t[7], d);
    return d;
}

constexpr int KNN_BLOCK = 256;

// merged top-2 of the nseg per-segment partial results of one query row
__device__ __forceinline__ uint2 merged_knn(const uint2* __restrict__ knn, size_t seg_stride, size_t idx, int nseg) {
    uint2 r = knn[idx];
    for (int sgm = 1; sgm < nseg; ++sgm) {
        const uint2 o = knn[(size_t)sgm * seg_stride + idx];
        const uint32_t hi = r.x > o.x ? r.x : o.x;
        uint32_t sec = r.y < o.y ? r.y : o.y;
        sec = sec < hi ? sec : hi;
        r.x = r.x < o.x ? r.x : o.x;
        r.y = sec;
    }
    return r;
}

// qsel / nsel (optional): scan only the listed query rows of this direction (not used by the product path any more:
// the mutual check is hamming_verify below); qsel = nullptr scans every row.
// nseg: the train range of every query tile is split into nseg segments scanned by different workgroups
// (partial top-2 per segment, merged by the consumers).  One wave then works for ~N/nseg rows instead of N,
// which keeps the dispatch rounds short: with 2000 rows a whole-range wave lasts ~0.5 ms and any grid that
// is not a multiple of the 8192 resident waves wastes up to one such round.
__global__ __launch_bounds__(KNN_BLOCK) void hamming_knn2_kernel(int B, int tiles, int ndir, int dir0, int nseg, int row_stride,
                                                                 const uint8_t* __restrict__ d1,
                                                                 const int32_t* __restrict__ n1,
                                                                 const uint8_t* __restrict__ d2,
                                                                 const int32_t* __restrict__ n2,
                                                                 uint2* __restrict__ knn12, uint2* __restrict__ knn21,
                                                                 const int32_t* __restrict__ qsel,
                                                                 const int32_t* __restrict__ nsel,
                                                                 uint32_t* __restrict__ claim_init) {
    // XCD-aware block -> (frame pair, direction, tile) mapping.  Workgroups are dispatched round-robin
    // over the 8 XCDs (workgroup L runs on XCD L % 8, each XCD has a private 4 MiB L2).  With the
    // natural (tile, dir, frame) order the 2*tiles workgroups of one frame pair land on all 8 XCDs
    // and every L2 fetches the pair's descriptors again (measured: FETCH_SIZE = 7.8x the compulsory
    // bytes).  Here all workgroups of a frame pair share one XCD: frame = (k / per_frame) * 8 + xcd.
    const int per_frame = tiles * ndir * nseg;
    const int L = blockIdx.x;
    const int xcd = L & 7, k = L >> 3;
    const int b = (k / per_frame) * 8 + xcd;
    if (b >= B) return;
    const int local = k % per_frame;
    const int seg = local % nseg;
    const int dir = dir0 + (local / nseg) / tiles;
    const int tile = (local / nseg) % tiles;
    const int na = n1[b], nb = n2[b];
    const int nq = qsel ? nsel[b] : (dir == 0 ? na : nb);
    const int nt_all = dir == 0 ? nb : na;
    // this workgroup's train rows [j0, nt): equal segments rounded up to the 4-row trip
    const int seg_len = (((nt_all + nseg - 1) / nseg) + 3) & ~3;
    const int j0 = min(seg * seg_len, nt_all);
    const int nt = min(j0 + seg_len, nt_all);
    const int q_base = tile * KNN_BLOCK;
    const size_t frame_off = (size_t)b * row_stride;
    // lazy mutual matching: the per-column claims consumed by nnr_forward_kernel are reset here (segment 0 covers
    // every column index once) instead of by a separate memset launch
    if (claim_init && seg == 0 && dir == dir0 && q_base + (int)threadIdx.x < row_stride)
        claim_init[frame_off + q_base + threadIdx.x] = 0xFFFFFFFFu;
    if (q_base + (int)(threadIdx.x & ~63u) >= nq) return;  // wave-uniform: no barriers in this kernel
    const uint8_t* Q = (dir == 0 ? d1 : d2) + frame_off * STVO_DESC_BYTES;
    const uint32_t* __restrict__ T = reinterpret_cast<const uint32_t*>((dir == 0 ? d2 : d1) + frame_off * STVO_DESC_BYTES);
    uint2* __restrict__ out = (dir == 0 ? knn12 : knn21) + (size_t)seg * B * row_stride + frame_off;

    const int q = q_base + threadIdx.x;
    const int qc = q < nq ? q : nq - 1;  // tail lanes scan a valid row and discard the result
    const int qi = qsel ? qsel[frame_off + qc] : qc;
    const uint4 q0 = reinterpret_cast<const uint4*>(Q)[2 * qi];
    const uint4 q1 = reinterpret_cast<const uint4*>(Q)[2 * qi + 1];

    uint32_t best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
    uint32_t second_d = 0xFFFFu;  // == second >> 16, refreshed only when the top-2 changes
    int j = j0;
    // 4 train rows (128 B = two s_load_dwordx16) per trip: 4 independent popcount chains give the
    // VALU ILP, 8 waves/SIMD hide the scalar-cache latency of the next trip's loads.
    // Top-2 bookkeeping (3 half-rate VALU ops) is skipped wave-uniformly when no lane can change:
    // key > second for every lane  <=>  d >= second_d for every lane (train indices only grow, so an
    // equal distance can never displace the current second).  After the first few hundred rows most
    // rows take the skip: one v_cmp + s_cbranch instead of lshl_or + med3 + min.
    auto update = [&](uint32_t d, uint32_t jj) {
        if (__builtin_amdgcn_ballot_w64(d < second_d) != 0ull) {
            const uint32_t key = pack_key(d, jj);
            second = med3_u32(best, second, key);
            best = best < key ? best : key;
            second_d = second >> 16;
        }
    };
    for (; j + 4 <= nt; j += 4) {
        uint32_t t[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) t[k] = T[8 * j + k];
        uint32_t d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = hamming256(q0, q1, t + 8 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) update(d[u], (uint32_t)(j + u));
    }
    for (; j < nt; ++j) update(hamming256(q0, q1, T + 8 * j), (uint32_t)j);
    if (q < nq) out[qi] = make_uint2(best, second);
}

// Query blocks per wave of the matrix-core scan (match_mfma.hip), 0 = use the VALU scan: train indices must fit the
// 13 index bits of its keys.  STVO_KNN_MFMA=0 selects the VALU kernels (K1 + K1v) for comparison runs.
int knn_mfma_qb(int max_n) {
    const int qb = dbg().knn_mfma != DBG_UNSET ? dbg().knn_mfma : 2;
    return max_n <= 8192 ? qb : 0;
}

void launch_hamming_knn2(hipStream_t s, int B, int row_stride, int max_n, const uint8_t* d1, const int32_t* n1,
                         const uint8_t* d2, const int32_t* n2, uint2* knn12, uint2* knn21, int both_directions,
                         int lds_pad_bytes, int dir0, const int32_t* qsel, const int32_t* nsel, int nseg, uint32_t* claim_init) {
    if (B <= 0 || max_n <= 0) return;
    const int mfma_qb = knn_mfma_qb(max_n);
    if (mfma_qb > 0) {
        // (lds_pad_bytes is not applied: co-residency with the pose kernel does not pay for this kernel)
        launch_hamming_knn2_mfma(s, B, row_stride, max_n, d1, n1, d2, n2, knn12, knn21, both_directions, dir0, qsel, nsel, nseg,
                                 claim_init, mfma_qb);
        return;
    }
    const int tiles = (max_n + KNN_BLOCK - 1) / KNN_BLOCK, ndir = both_directions ? 2 : 1;
    const int groups = (B + 7) / 8;  // frame pairs are dealt to the 8 XCDs in groups of 8
    dim3 grid((unsigned)(groups * 8 * tiles * ndir * nseg));
    // lds_pad_bytes > 0 only caps the number of resident workgroups per CU (the kernel uses no LDS), leaving
    // wave slots and VGPRs for a concurrently running pose kernel (stvo_ctx_set_overlap)
    hipLaunchKernelGGL(hamming_knn2_kernel, grid, dim3(KNN_BLOCK), (size_t)lds_pad_bytes, s, B, tiles, ndir, dir0,
                       nseg, row_stride, d1, n1, d2, n2, knn12, knn21, qsel, nsel, claim_init);
}

// K2: m12[i] = j  iff  float(d0) < float(d1) * nnr  (12 direction)  and, when `mutual`, the 21
// direction's own ratio-tested best of j is i.  Fewer than two train rows => no match (the
// reference is undefined there, src/matching.cpp:54).
__global__ __launch_bounds__(256) void nnr_mutual_kernel(int nseg, int row_stride, const uint2* __restrict__ knn12,
                                                         const uint2* __restrict__ knn21,
                                                         const int32_t* __restrict__ n1, const int32_t* __restrict__ n2,
                                                         float nnr, int mutual, int32_t* __restrict__ m12) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= row_stride) return;
    const int na = n1[b], nb = n2[b];
    const size_t off = (size_t)b * row_stride;
    int m = -1;
    if (i < na && nb >= 2) {
        const uint2 k = merged_knn(knn12, (size_t)gridDim.y * row_stride, off + i, nseg);
        const float f0 = (float)(k.x >> 16), f1 = (float)(k.y >> 16);
        if (f0 < f1 * nnr) m = (int)(k.x & 0xFFFFu);
        if (mutual && m >= 0) {
            bool keep = false;
            if (na >= 2) {
                const uint2 r = merged_knn(knn21, (size_t)gridDim.y * row_stride, off + m, nseg);
                const float r0 = (float)(r.x >> 16), r1 = (float)(r.y >> 16);
                keep = (r0 < r1 * nnr) && ((int)(r.x & 0xFFFFu) == i);
            }
            if (!keep) m = -1;
        }
    }
    m12[off + i] = m;
}

void launch_nnr_mutual(hipStream_t s, int B, int row_stride, const uint2* knn12, const uint2* knn21, const int32_t* n1,
                       const int32_t* n2, float nnr, int mutual, int32_t* m12, int nseg) {
    if (B <= 0 || row_stride <= 0) return;
    dim3 grid((row_stride + 255) / 256, B);
    hipLaunchKernelGGL(nnr_mutual_kernel, grid, dim3(256), 0, s, nseg, row_stride, knn12, knn21, n1, n2, nnr, mutual, m12);
}

// ---- lazy mutual matching -----------------------------------------------------------------------
// StVO::match needs matches_21[j] only for columns j that are some row's accepted forward match
// (src/matching.cpp:80-86 reads matches_21[matches_12[i1]] and nothing else), and of matches_21[j] it only
// needs to know whether it EQUALS i1.  With nnr <= 1 (enforced at the ABI) that is a RANGE question, not a
// top-2 question.  Let i claim column j with forward distance d0 = D(i, j).  The reverse pass of the reference
// gives matches_21[j] = i  iff  i is the column's nearest row (lowest index among equals) and
// float(d0) < float(second) * nnr, second = the smallest distance of any other row.  x -> float(x) * nnr is
// non-decreasing, and for nnr <= 1 "float(d0) < float(d') * nnr" already implies d' > d0, so
//       matches_21[j] == i   <=>   no row i' != i has D(i', j) <= T,
//       T = the largest d' for which float(d0) < float(d') * nnr is FALSE            (T >= d0).
// Among several rows claiming the same column only the one with the smallest (d0, i) can survive (it blocks
// all the others), so one claim per column is kept (atomicMin on the packed key).
//
// The verification scan therefore needs no top-2 bookkeeping, and it can stop a row early: the popcount of
// the first 128 bits is a lower bound of the distance.  T is ~d0 / nnr (27 for a 20-bit match at 0.75) while
// the first half of an unrelated row differs in 64 +- 5.7 bits, so practically every (wave, row) step takes
// the wave-uniform short cut: 4 x (v_xor + v_bcnt) + one compare instead of 8 x (v_xor + v_bcnt) + the top-2
// update.  Exactness does not depend on the data: whenever any lane's lower bound is within its T (its own
// claimant excepted) the wave finishes the row at full length.
__global__ __launch_bounds__(256) void nnr_forward_kernel(int nseg, int row_stride, const uint2* __restrict__ knn12,
                                                          const int32_t* __restrict__ n1, const int32_t* __restrict__ n2,
                                                          float nnr, int32_t* __restrict__ cand,
                                                          uint32_t* __restrict__ claim /* preset to 0xFFFFFFFF by K1 */) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= row_stride) return;
    const int na = n1[b], nb = n2[b];
    const size_t off = (size_t)b * row_stride;
    int m = -1;
    if (i < na && nb >= 2) {
        const uint2 k = merged_knn(knn12, (size_t)gridDim.y * row_stride, off + i, nseg);
        const float f0 = (float)(k.x >> 16), f1 = (float)(k.y >> 16);
        if (f0 < f1 * nnr) {
            m = (int)(k.x & 0xFFFFu);
            atomicMin(&claim[off + m], (k.x & 0xFFFF0000u) | (uint32_t)i);  // (d0 << 16) | claimant
        }
    }
    cand[off + i] = m;
}

// one workgroup per frame pair: ascending list of the claimed columns + their count; clears their verdicts.
// (Fusing this with the forward test into one workgroup per frame pair was tried and is slower: a single workgroup
// then merges 2000 x nseg partial keys serially — 18 us instead of 6 + 5 for one frame, 68 us instead of 36 for 512.)
__global__ __launch_bounds__(256) void compact_need_kernel(int row_stride, const uint32_t* __restrict__ claim,
                                                           const int32_t* __restrict__ n2, int32_t* __restrict__ qsel,
                                                           int32_t* __restrict__ nsel, int32_t* __restrict__ blocked) {
    __shared__ int s_wave[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t off = (size_t)b * row_stride;
    const int nb = n2[b];
    const int per = (row_stride + 255) / 256;
    const int lo = tid * per, hi = min(lo + per, nb);
    int cnt = 0;
    for (int j = lo; j < hi; ++j) cnt += claim[off + j] != 0xFFFFFFFFu;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wv) base += s_wave[w];
        tot += s_wave[w];
    }
    int pos = base + incl - cnt;
    for (int j = lo; j < hi; ++j)
        if (claim[off + j] != 0xFFFFFFFFu) {
            qsel[off + pos++] = j;
            blocked[off + j] = 0;
        }
    if (tid == 0) nsel[b] = tot;
}

__device__ __forceinline__ uint32_t hamming128_lo(const uint4& q0, const uint32_t* __restrict__ t) {
    uint32_t d = bcnt0(q0.x ^ t[0]);
    d = bcnt_acc(q0.y ^ t[1], d);
    d = bcnt_acc(q0.z ^ t[2], d);
    d = bcnt_acc(q0.w ^ t[3], d);
    return d;
}
__device__ __forceinline__ uint32_t hamming128_hi(const uint4& q1, const uint32_t* __restrict__ t, uint32_t d) {
    d = bcnt_acc(q1.x ^ t[4], d);
    d = bcnt_acc(q1.y ^ t[5], d);
    d = bcnt_acc(q1.z ^ t[6], d);
    d = bcnt_acc(q1.w ^ t[7], d);
    return d;
}

// K1v: lane = one claimed column j of the curr frame (descriptor in 8 VGPRs, threshold T and claimant in two
// more); the prev rows are wave-uniform scalar-cache loads exactly as in K1; same XCD-aware mapping and the same
// row segments.  blocked[j] = 1 iff some prev row other than the claimant lies within T.
__global__ __launch_bounds__(KNN_BLOCK) void hamming_verify_kernel(int B, int tiles, int nseg, int row_stride,
                                                                   const uint8_t* __restrict__ d1,
                                                                   const int32_t* __restrict__ n1,
                                                                   const uint8_t* __restrict__ d2,
                                                                   const uint32_t* __restrict__ claim,
                                                                   const int32_t* __restrict__ qsel,
                                                                   const int32_t* __restrict__ nsel, float nnr,
                                                                   int32_t* __restrict__ blocked) {
    const int per_frame = tiles * nseg;
    const int L = blockIdx.x;
    const int xcd = L & 7, k = L >> 3;
    const int b = (k / per_frame) * 8 + xcd;
    if (b >= B) return;
    const int local = k % per_frame;
    const int seg = local % nseg;
    const int tile = local / nseg;
    const int nq = nsel[b];
    const int nt_all = n1[b];
    const int seg_len = (((nt_all + nseg - 1) / nseg) + 3) & ~3;
    const int j0 = min(seg * seg_len, nt_all);
    const int nt = min(j0 + seg_len, nt_all);
    const int q_base = tile * KNN_BLOCK;
    if (q_base + (int)(threadIdx.x & ~63u) >= nq) return;  // wave-uniform
    const size_t frame_off = (size_t)b * row_stride;
    const uint32_t* __restrict__ T = reinterpret_cast<const uint32_t*>(d1 + frame_off * STVO_DESC_BYTES);

    const int q = q_base + threadIdx.x;
    const int qc = q < nq ? q : nq - 1;  // tail lanes repeat a valid column and discard the verdict
    const int qi = qsel[frame_off + qc];
    const uint4 q0 = reinterpret_cast<const uint4*>(d2 + frame_off * STVO_DESC_BYTES)[2 * qi];
    const uint4 q1 = reinterpret_cast<const uint4*>(d2 + frame_off * STVO_DESC_BYTES)[2 * qi + 1];
    const uint32_t c = claim[frame_off + qi];
    const uint32_t istar = c & 0xFFFFu;
    uint32_t thr = c >> 16;  // d0
    {
        const float f0 = (float)thr;
        while (thr < 256u && !(f0 < (float)(thr + 1u) * nnr)) ++thr;  // largest distance that still blocks
    }
    bool blk = false;
    // 4 prev rows (128 B, two s_load_dwordx16) per step
    auto step4 = [&](const uint32_t* __restrict__ t, int jb) {
        uint32_t p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = hamming128_lo(q0, t + 8 * u);
        unsigned long long any = 0ull;
#pragma unroll
        for (int u = 0; u < 4; ++u) any |= __builtin_amdgcn_ballot_w64(p[u] <= thr);
        if (any != 0ull) {  // wave-uniform, rare: first discount every lane's own claimant, then go the full length
            unsigned long long other = 0ull;
#pragma unroll
            for (int u = 0; u < 4; ++u) other |= __builtin_amdgcn_ballot_w64(p[u] <= thr && (uint32_t)(jb + u) != istar);
            if (other != 0ull) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t d = hamming128_hi(q1, t + 8 * u, p[u]);
                    blk = blk || (d <= thr && (uint32_t)(jb + u) != istar);
                }
            }
        }
    };
    int j = j0;
    // two register sets of 32 SGPRs: the scalar loads of the next 4 rows are in flight while the current 4 are
    // compared (a step is only ~36 VALU ops, too short to hide the scalar-cache latency behind other waves alone)
    if (j + 4 <= nt) {
        uint32_t ta[32], tb[32];
#pragma unroll
        for (int w = 0; w < 32; ++w) ta[w] = T[8 * j + w];
        for (; j + 8 <= nt; j += 8) {
#pragma unroll
            for (int w = 0; w < 32; ++w) tb[w] = T[8 * (j + 4) + w];
            step4(ta, j);
            const int jn = (j + 12 <= nt) ? j + 8 : j;  // last round: reload something valid, result unused
#pragma unroll
            for (int w = 0; w < 32; ++w) ta[w] = T[8 * jn + w];
            step4(tb, j + 4);
        }
        if (j + 4 <= nt) {
            step4(ta, j);  // ta holds rows j..j+3 here (loaded by the prologue or by the last round)
            j += 4;
        }
    }
    for (; j < nt; ++j) {
        const uint32_t d = hamming256(q0, q1, T + 8 * j);
        blk = blk || (d <= thr && (uint32_t)j != istar);
    }
    if (q < nq && blk) blocked[frame_off + qi] = 1;  // benign race between segments: every writer stores 1
}

// m12[i] = cand[i] iff i holds the claim on that column and the verification found no blocking row.
// Fewer than two prev rows => no match (the reference's knnMatch(k = 2) row would have one entry; :54 is UB).
// Column by column: the claimant i of column j has cand[i] = j (that is how it came to claim it), a row claims at most one column, and
// every other row ends without a match — so the pass reads claim[] and blocked[] in order and scatters the survivors, instead of two
// dependent gathers per row (round 5: 18.7 -> us per 1024 frames, 99 -> MB of calibrated fetch).  One workgroup per frame pair.
__global__ __launch_bounds__(1024) void nnr_reverse_check_kernel(int row_stride, const uint32_t* __restrict__ claim,
                                                                 const int32_t* __restrict__ blocked, const int32_t* __restrict__ n1,
                                                                 int32_t* __restrict__ m12) {
    const int b = blockIdx.x;
    const size_t off = (size_t)b * row_stride;
    for (int i = threadIdx.x; i < row_stride; i += 1024) m12[off + i] = -1;
    __syncthreads();  // (the survivors below overwrite entries other threads just cleared)
    if (n1[b] < 2) return;
    for (int j = threadIdx.x; j < row_stride; j += 1024) {
        const uint32_t c = claim[off + j];
        if (c != 0xFFFFFFFFu && blocked[off + j] == 0) m12[off + (c & 0xFFFFu)] = j;
    }
}

// ---- reverse check on the matrix cores: which (column, row) pairs still have to be looked at ------------------
// Row i* holds the claim on column j with d0 = D(i*, j); the match survives iff no other row i' has D(i', j) <= T_j,
// T_j = block_threshold(d0) (see the derivation above).  Let (b, s) be the forward top-2 of such a row i' (already
// computed by the forward scan).  Either j is one of its two entries — then D(i', j) is KNOWN and the row can be judged
// from knn12 alone — or j is not, and then both entries sort before (D(i', j), j), so  second_distance(i') <= D(i', j) <= T_j.
// Hence the only rows whose distance to column j must be evaluated are  S_j = { i' : second_distance(i') <= T_j }.
// For descriptors that discriminate at all S_j is tiny (a second-best distance is that of an unrelated row, T_j that of
// a good match), and the |claimed| x N1 reverse scan collapses to |claimed| x |S|.  Exactness never depends on that:
// a per-frame cut tau splits the claimed columns into LIGHT (T_j <= tau: scanned against S = { second <= tau }, a
// superset of every S_j) and HEAVY (T_j > tau: scanned against all rows); tau minimises the number of distance
// evaluations (C - L(tau)) * N1 + L(tau) * |S(tau)| from two 257-bin histograms, tau = -1 being "everything heavy".
__device__ __forceinline__ uint32_t block_threshold(uint32_t d0, float nnr) {
    const float f0 = (float)d0;
    uint32_t thr = d0;
    while (thr < 256u && !(f0 < (float)(thr + 1u) * nnr)) ++thr;  // largest distance that still blocks
    return thr;
}

// One workgroup per frame pair does everything between the forward scan and the reverse scans:
//   1. merges the per-segment top-2 of every row, applies the forward ratio test (StVO::matchNNR, matching.cpp:53-58) and
//      resolves the column claims with LDS atomics (one claim per column: the smallest (d0, row));
//   2. judges from the forward top-2 alone every (row, column) pair where the column is one of the row's two entries;
//   3. picks the cut tau and writes the LIGHT / HEAVY column lists and the row list S.
// out: m12[i] (the claimants whose claim the forward top-2 does not already block; the reverse scans clear more), claim[j], qsel
// (LIGHT columns from the front, HEAVY from the back of the frame's slot), tsel,
// nsel[0][b] = claimed columns, [1] = light, [2] = heavy, [3] = |S|, [4] = tau.  knn12 segment 0 is left holding the merged top-2.
constexpr int PLAN_BLOCK = 512;  // (1024: two rounds of two workgroups per CU for 1024 frames, 34 us; 512: one round, 29 us; 320: 32 us)
__global__ __launch_bounds__(PLAN_BLOCK) void forward_plan_kernel(int B, int nseg, int row_stride, uint2* __restrict__ knn12,
                                                                  const int32_t* __restrict__ n1,
                                                                  const int32_t* __restrict__ n2, float nnr,
                                                                  int32_t* __restrict__ m12, uint32_t* __restrict__ claim_g,
                                                                  int32_t* __restrict__ qsel, int32_t* __restrict__ tsel,
                                                                  int32_t* __restrict__ nsel) {
    extern __shared__ uint32_t plan_lds[];  // claim u32[stride] | T u16[stride] | verdict u8[stride]
    uint32_t* claim = plan_lds;
    uint16_t* Tc = reinterpret_cast<uint16_t*>(claim + row_stride);
    uint8_t* blk = reinterpret_cast<uint8_t*>(Tc + row_stride);
    __shared__ int hT[257], hS[257], s_cnt[3];
    __shared__ unsigned long long s_best;
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t off = (size_t)b * row_stride;
    const int na = n1[b], nb = n2[b];
    for (int j = tid; j < row_stride; j += PLAN_BLOCK) {
        claim[j] = 0xFFFFFFFFu;
        blk[j] = 0;
    }
    if (tid < 257) hT[tid] = hS[tid] = 0;
    if (tid < 3) s_cnt[tid] = 0;
    if (tid == 0) s_best = ~0ull;
    __syncthreads();
    const bool active = nb >= 2 && na >= 1;
    for (int i = tid; i < row_stride; i += PLAN_BLOCK) {  // forward ratio test + claims
        if (active && i < na) {
            const uint2 k = merged_knn(knn12, (size_t)B * row_stride, off + i, nseg);
            if (nseg > 1) knn12[off + i] = k;
            const float f0 = (float)(k.x >> 16), f1 = (float)(k.y >> 16);
            if (f0 < f1 * nnr) atomicMin(&claim[k.x & 0xFFFFu], (k.x & 0xFFFF0000u) | (uint32_t)i);  // (d0 << 16) | claimant
        }
        m12[off + i] = -1;  // (the claimants that survive are written by the last loop, barriers later)
    }
    __syncthreads();
    for (int j = tid; j < row_stride; j += PLAN_BLOCK) {  // thresholds of the claimed columns
        const uint32_t c = claim[j];
        claim_g[off + j] = c;
        if (c != 0xFFFFFFFFu) {
            const uint32_t T = block_threshold(c >> 16, nnr);
            Tc[j] = (uint16_t)T;
            atomicAdd(&hT[T], 1);
        }
    }
    __syncthreads();
    if (active)
        for (int i = tid; i < na; i += PLAN_BLOCK) {  // verdicts that need no distance evaluation; histogram of the second-best distances
            const uint2 k = knn12[off + i];           // written by this very thread above
            const uint32_t key[2] = {k.x, k.y};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (key[e] == 0xFFFFFFFFu) continue;
                const uint32_t j = key[e] & 0xFFFFu, c = claim[j];
                if (c != 0xFFFFFFFFu && (c & 0xFFFFu) != (uint32_t)i && (key[e] >> 16) <= (uint32_t)Tc[j]) blk[j] = 1;
            }
            if (k.y != 0xFFFFFFFFu) atomicAdd(&hS[min(k.y >> 16, 256u)], 1);
        }
    __syncthreads();
    // candidate cut tau = t - 1 for t = 0 .. 255 (thresholds above 254 are always heavy: they would need S = every row): exclusive
    // prefix sums of the two histograms (in place, by the first wave: five values per lane + a wave scan), then one cost per thread
    if (tid < 64) {
        int vT[5], vS[5], sT = 0, sS = 0;
#pragma unroll
        for (int e = 0; e < 5; ++e) {
            const int t = tid * 5 + e;
            vT[e] = t < 257 ? hT[t] : 0;
            vS[e] = t < 257 ? hS[t] : 0;
            sT += vT[e];
            sS += vS[e];
        }
        int iT = sT, iS = sS;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(iT, o, 64), c = __shfl_up(iS, o, 64);
            if (tid >= o) {
                iT += a;
                iS += c;
            }
        }
        int eT = iT - sT, eS = iS - sS;  // exclusive
#pragma unroll
        for (int e = 0; e < 5; ++e) {
            const int t = tid * 5 + e;
            if (t < 257) {
                hT[t] = eT;
                hS[t] = eS;
            }
            eT += vT[e];
            eS += vS[e];
        }
        if (tid == 63) s_cnt[0] = iT;  // C = all claimed columns (s_cnt[0] is reset below)
    }
    __syncthreads();
    if (tid < 256) {
        const int C = s_cnt[0], cT = hT[tid], cS = hS[tid];
        const unsigned long long cost = (unsigned long long)(C - cT) * (unsigned)na + (unsigned long long)cT * (unsigned)cS;
        atomicMin(&s_best, (cost << 9) | (unsigned)tid);
    }
    __syncthreads();
    if (tid == 0) s_cnt[0] = 0;
    __syncthreads();
    const int tau = (int)(s_best & 511ull) - 1;
    for (int j = tid; j < row_stride; j += PLAN_BLOCK) {
        const uint32_t c = claim[j];
        if (c == 0xFFFFFFFFu) continue;
        // the match as far as the forward top-2 can tell; the reverse scans clear the entries whose claim a scanned row blocks.
        // Fewer than two prev rows => no match (the reference's knnMatch(k = 2) row would have one entry; :54 is UB).
        if (!blk[j] && na >= 2) m12[off + (c & 0xFFFFu)] = j;
        if ((int)Tc[j] <= tau)
            qsel[off + atomicAdd(&s_cnt[0], 1)] = j;
        else
            qsel[off + row_stride - 1 - atomicAdd(&s_cnt[1], 1)] = j;
    }
    if (active)
        for (int i = tid; i < na; i += PLAN_BLOCK) {
            const uint2 k = knn12[off + i];
            if (k.y != 0xFFFFFFFFu && (int)(k.y >> 16) <= tau) tsel[off + atomicAdd(&s_cnt[2], 1)] = i;
        }
    __syncthreads();
    if (tid == 0) {
        nsel[b] = s_cnt[0] + s_cnt[1];
        nsel[(size_t)B + b] = s_cnt[0];
        nsel[2 * (size_t)B + b] = s_cnt[1];
        nsel[3 * (size_t)B + b] = s_cnt[2];
        nsel[4 * (size_t)B + b] = tau;
    }
}

// scratch of the matrix-core reverse check: the last B * row_stride int32 of knn21 hold tsel[] (the reverse top-2 array of the non-lazy
// path is not needed here)
struct ReversePlan {
    int32_t* tsel;
};
constexpr int KNN_REV_SLOTS = 4;  // workgroups per frame pair sharing the frame's reverse-check units (match_mfma.hip)
static ReversePlan reverse_plan(const LazyScratch& w, int B, int row_stride) {
    const size_t per = (size_t)B * row_stride;
    ReversePlan p;
    p.tsel = reinterpret_cast<int32_t*>(w.knn21 + (w.knn_capacity - per)) + per;  // (the place it had beside the flags of round 5's first form)
    return p;
}

// light columns against the rows of S (positions in tsel), heavy columns against every row — one launch; m12 loses the claimants whose
// column some scanned row blocks
static void launch_reverse_scans(hipStream_t s, int B, int row_stride, const uint8_t* d1, const int32_t* n1, const uint8_t* d2, float nnr,
                                 const LazyScratch& w, const ReversePlan& rp, int32_t* m12) {
    launch_hamming_knn2_mfma_reverse(s, B, row_stride, d1, n1, d2, w.qsel, w.nsel, rp.tsel, reinterpret_cast<const uint32_t*>(w.need), nnr, m12,
                                     KNN_REV_SLOTS);
}

void launch_hamming_verify(hipStream_t s, int B, int row_stride, const uint8_t* d1, const int32_t* n1, const uint8_t* d2,
                           float nnr, const LazyScratch& w, int lds_pad_bytes, int nseg, int32_t* m12) {
    if (B <= 0 || row_stride <= 0) return;
    const int mfma_qb = knn_mfma_qb(row_stride);
    if (mfma_qb > 0) {  // the reverse scans on the lists left by the last forward_plan_kernel (they clear the same entries of m12 again)
        launch_reverse_scans(s, B, row_stride, d1, n1, d2, nnr, w, reverse_plan(w, B, row_stride), m12);
        return;
    }
    const int tiles = (row_stride + KNN_BLOCK - 1) / KNN_BLOCK, groups = (B + 7) / 8;
    hipLaunchKernelGGL(hamming_verify_kernel, dim3((unsigned)(groups * 8 * tiles * nseg)), dim3(KNN_BLOCK),
                       (size_t)lds_pad_bytes, s, B, tiles, nseg, row_stride, d1, n1, d2,
                       reinterpret_cast<const uint32_t*>(w.need), w.qsel, w.nsel, nnr, reinterpret_cast<int32_t*>(w.knn21));
}

void launch_match_mutual_lazy(hipStream_t s, int B, int row_stride, const uint8_t* d1, const int32_t* n1,
                              const uint8_t* d2, const int32_t* n2, float nnr, const LazyScratch& w, int32_t* m12,
                              int lds_pad_bytes, hipEvent_t wait_before_m12_write, hipEvent_t* tev) {
    if (B <= 0 || row_stride <= 0) return;
    const dim3 grid2((row_stride + 255) / 256, B);
    const int nseg = knn_pick_nseg(B, row_stride, w.knn_capacity);
    uint32_t* claim = reinterpret_cast<uint32_t*>(w.need);
    int32_t* blocked = reinterpret_cast<int32_t*>(w.knn21);  // the reverse top-2 array is not needed any more
    if (tev) (void)hipEventRecord(tev[0], s);
    // (the VALU path resets the column claims inside K1; forward_plan_kernel of the matrix-core path builds them in LDS)
    launch_hamming_knn2(s, B, row_stride, row_stride, d1, n1, d2, n2, w.knn12, w.knn21, 0, lds_pad_bytes, 0, nullptr,
                        nullptr, nseg, knn_mfma_qb(row_stride) > 0 ? nullptr : claim);
    if (tev) (void)hipEventRecord(tev[1], s);
    const int mfma_qb = knn_mfma_qb(row_stride);
    if (mfma_qb > 0) {
        // Two launches behind the forward scan (three until the end of round 5: a last pass turned claim[] and per-column flags into
        // m12): the plan writes m12 as far as the forward top-2 decides it, the reverse scans clear the claimants they find blocked.
        const ReversePlan rp = reverse_plan(w, B, row_stride);
        if (wait_before_m12_write) (void)hipStreamWaitEvent(s, wait_before_m12_write, 0);
        if (tev && tev[2]) (void)hipEventRecord(tev[2], s);
        hipLaunchKernelGGL(forward_plan_kernel, dim3(B), dim3(PLAN_BLOCK), (size_t)row_stride * 7, s, B, nseg, row_stride, w.knn12,
                           n1, n2, nnr, m12, claim, w.qsel, rp.tsel, w.nsel);
        launch_reverse_scans(s, B, row_stride, d1, n1, d2, nnr, w, rp, m12);
        if (tev && tev[3]) (void)hipEventRecord(tev[3], s);
        return;
    }
    hipLaunchKernelGGL(nnr_forward_kernel, grid2, dim3(256), 0, s, nseg, row_stride, w.knn12, n1, n2, nnr, w.cand, claim);
    hipLaunchKernelGGL(compact_need_kernel, dim3(B), dim3(256), 0, s, row_stride, claim, n2, w.qsel, w.nsel, blocked);
    if (tev && tev[2]) (void)hipEventRecord(tev[2], s);
    launch_hamming_verify(s, B, row_stride, d1, n1, d2, nnr, w, lds_pad_bytes, nseg, m12);
    if (tev && tev[3]) (void)hipEventRecord(tev[3], s);
    if (wait_before_m12_write) (void)hipStreamWaitEvent(s, wait_before_m12_write, 0);
    hipLaunchKernelGGL(nnr_reverse_check_kernel, dim3(B), dim3(1024), 0, s, row_stride, claim, blocked, n1, m12);
}

// Integer-VALU roof probe: the same instruction mix as K1's inner loop (xor, bcnt-accumulate,
// lshl_or, med3, min) on register-only operands; 19 lane-ops per "pair", 8 pairs per iteration.
const double kValuProbeOpsPerThreadIter = 19.0 * 8.0;

__global__ __launch_bounds__(256) void valu_probe_kernel(int iters, uint32_t* sink) {
    uint4 q0 = make_uint4(threadIdx.x * 2654435761u, blockIdx.x * 40503u + 1u, 0x9E3779B9u ^ threadIdx.x, 0x85EBCA6Bu);
    uint4 q1 = make_uint4(q0.y * 3u, q0.z * 5u, q0.w * 7u, q0.x * 11u);
    uint32_t best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
    uint32_t t[8];
    const uint32_t seed = __builtin_amdgcn_readfirstlane(blockIdx.x * 7919u + 13u);
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = seed * (2 * k + 3);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = __builtin_amdgcn_readfirstlane(t[k] + 0x01010101u * (k + 1));  // SALU
            const uint32_t d = hamming256(q0, q1, t);
            const uint32_t key = pack_key(d, (uint32_t)(it * 8 + u));
            second = med3_u32(best, second, key);
            best = best < key ? best : key;
        }
    }
    if ((best ^ second) == 0x12345u) sink[0] = best;  // keep the chain alive
}

__global__ __launch_bounds__(256) void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

void launch_copy16(hipStream_t s, const void* src, void* dst, size_t bytes) {
    const size_t n16 = (bytes + 15) / 16;
    if (n16 == 0) return;
    const size_t blocks = (n16 + 255) / 256;
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16);
}

void launch_valu_probe(hipStream_t s, int blocks, int iters, uint32_t* sink) {
    hipLaunchKernelGGL(valu_probe_kernel, dim3(blocks), dim3(256), 0, s, iters, sink);
}

// ---- small sets (key-lines): StVO::match of one frame pair in ONE workgroup -------------------------------------------
// The key-line sets of a frame hold ~100 rows (config: lsd_nfeatures = 100 / 300).  Through the general machinery (matrix-core scan,
// forward plan, two reverse scans, final check: five launches whose workgroups spend their lives in prologues) the f2f line match
// of 1024 frames took ~70 us alone and, sharing the CUs with the key-point scan on the other stream, stretched that scan by ~15 %.
// Here a WAVE takes 64 query rows of one direction and scans the other set exactly like K1 above (lane = query row in registers,
// train rows wave-uniform through the scalar cache: no vector-memory or LDS instruction in the loop), keeping the two smallest
// packed keys (distance << 16 | index: lowest index among equal distances = knnMatch's tie order); the top-2 of both directions
// meet in LDS for the float ratio test (src/matching.cpp:53-58) and the mutual check (:80-86) of nnr_mutual_kernel, verbatim.
// `cap` (<= row_stride, <= 512): rows per set the LDS arrays are sized for — the caller knows that no set of the batch holds more.
__global__ __launch_bounds__(256) void match_small_kernel(int row_stride, int cap, const uint8_t* __restrict__ d1,
                                                          const int32_t* __restrict__ n1, const uint8_t* __restrict__ d2,
                                                          const int32_t* __restrict__ n2, float nnr, int mutual,
                                                          int32_t* __restrict__ m12) {
    extern __shared__ uint2 s_knn[];  // knn12[cap], knn21[cap]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int na = min(n1[b], cap), nb = min(n2[b], cap);
    const size_t off = (size_t)b * row_stride;
    uint2* k12 = s_knn;
    uint2* k21 = s_knn + cap;
    const int tiles_a = (na + 63) >> 6, tiles_b = mutual ? (nb + 63) >> 6 : 0;
    for (int item = tid >> 6; item < tiles_a + tiles_b; item += 4) {  // wave-uniform
        const int it = __builtin_amdgcn_readfirstlane(item);
        const bool fwd = it < tiles_a;
        const int q = (fwd ? it : it - tiles_a) * 64 + lane;
        const int nq = fwd ? na : nb, nt = fwd ? nb : na;
        const uint8_t* Q = (fwd ? d1 : d2) + off * STVO_DESC_BYTES;
        const uint32_t* __restrict__ T = reinterpret_cast<const uint32_t*>((fwd ? d2 : d1) + off * STVO_DESC_BYTES);
        const int qc = q < nq ? q : nq - 1;  // tail lanes scan a valid row and discard the result
        const uint4 q0 = reinterpret_cast<const uint4*>(Q)[2 * qc];
        const uint4 q1 = reinterpret_cast<const uint4*>(Q)[2 * qc + 1];
        uint32_t best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
        auto update = [&](uint32_t d, uint32_t jj) {
            const uint32_t key = pack_key(d, jj);
            second = med3_u32(best, second, key);
            best = best < key ? best : key;
        };
        int j = 0;
        for (; j + 4 <= nt; j += 4) {
            uint32_t t[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) t[k] = T[8 * j + k];
            uint32_t d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = hamming256(q0, q1, t + 8 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u) update(d[u], (uint32_t)(j + u));
        }
        for (; j < nt; ++j) update(hamming256(q0, q1, T + 8 * j), (uint32_t)j);
        if (q < nq) (fwd ? k12 : k21)[q] = make_uint2(best, second);
    }
    __syncthreads();
    for (int i = tid; i < row_stride; i += 256) {
        int m = -1;
        if (i < na && nb >= 2) {
            const uint2 fk = k12[i];
            const float f0 = (float)(fk.x >> 16), f1 = (float)(fk.y >> 16);
            if (f0 < f1 * nnr) m = (int)(fk.x & 0xFFFFu);
            if (mutual && m >= 0) {
                bool keep = false;
                if (na >= 2) {
                    const uint2 rk = k21[m];
                    const float r0 = (float)(rk.x >> 16), r1 = (float)(rk.y >> 16);
                    keep = (r0 < r1 * nnr) && ((int)(rk.x & 0xFFFFu) == i);
                }
                if (!keep) m = -1;
            }
        }
        m12[off + i] = m;
    }
}

bool match_small_ok(int row_stride) { return row_stride > 0 && row_stride <= 512; }

void launch_match_small(hipStream_t s, int B, int row_stride, const uint8_t* d1, const int32_t* n1, const uint8_t* d2,
                        const int32_t* n2, float nnr, int mutual, int32_t* m12, int cap) {
    if (B <= 0 || !match_small_ok(row_stride)) return;
    if (cap <= 0 || cap > row_stride) cap = row_stride;
    const size_t lds = (size_t)cap * 2 * sizeof(uint2);
    hipLaunchKernelGGL(match_small_kernel, dim3(B), dim3(256), lds, s, row_stride, cap, d1, n1, d2, n2, nnr, mutual, m12);
}

}  // namespace stvo
