// match_mfma.hip — K1m `hamming_knn2_mfma`: the brute-force Hamming 2-NN scan of K1 (match_kernels.hip) with the
// 256-bit distances taken from the gfx950 matrix cores instead of 8 x (v_xor + v_bcnt) per pair.
//
// Same contract as hamming_knn2_kernel — cv::BFMatcher(NORM_HAMMING).knnMatch(desc1, desc2, ., 2) as called from
// StVO::matchNNR (/root/reference/src/matching.cpp:47-48): per query row the two smallest distances in ascending
// order, lowest train index first among equal distances — and the same output ([nseg][B][row_stride] packed keys).
//
// Why this is a matrix product and not a reshaping trick: the all-pairs Hamming distance of bit rows IS a Gram
// matrix.  With bits mapped to x = -+a (query: -a for a set bit) and y = +-a (train: +a for a set bit),
//       sum_k x_k * y_k = a^2 * (#differing - #equal) = 2 a^2 * h - 256 a^2.
// Operands are FP4 (e2m1) on gfx950's block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (round 4; rounds 1-3 used i8 on
// v_mfma_i32_32x32x32_i8): +-4 is an e2m1 value (0b0110 / 0b1110), so a bit becomes a NIBBLE and a 256-bit row four K = 64
// steps, each as long as one of the eight i8 steps it replaces — the matrix pipe needs half the time per tile
// (tools/fp4_probe.hip on MI355X: 151 cycles per 32 x 32 tile of 256-bit distances against 265 - 330 for i8, 16 - 17 T pairs/s
// register-only against 9 - 10; the probe also checks the operand convention used here against a CPU product, entry for
// entry).  With both block scales at 2^4, a = 64 and the f32 accumulator holds 8192 h - 2^20 exactly; the C operand of a
// tile's first matrix instruction carries the (tile-relative) train index, so the accumulator leaves the matrix pipe already
// as the packed key 8192 h + j - 2^20 (|key| < 2^22 after every rebasing: exact in f32) whose order is the (distance, train
// index) order knnMatch uses.  The VALU work per (query, train) pair drops from 16 + ~2 ops (K1) to 2: v_med3_f32 + v_min_f32.
//
// Mapping (wave64, 32x32 tiles):
//   * MFMA "A" operand = 32 TRAIN rows, expanded from bits to nibbles once per workgroup into LDS (4 KB per tile, 2 MF_TPB
//     buffers, one barrier per MF_TPB tiles) and read back as ready-made fragments (one contiguous ds_read_b128 per K step);
//   * MFMA "B" operand = 32 QUERY rows per block, expanded once and kept in VGPRs for the whole scan (16 per block;
//     QB blocks per wave -> every train fragment read from LDS feeds QB matrix instructions);
//   * D[train][query]: a lane owns ONE query column and 16 train rows of the tile, so the running (best, second)
//     of a query is two VGPRs per lane and the final cross-lane merge is a single swap of the wave halves.
//   * bit -> K element mapping: K step kk covers descriptor words 2 kk (lower wave half) and 2 kk + 1 (upper); dword d of a
//     lane's fragment holds bits d, d + 4, ..., d + 28 of its word as nibbles: ((w << (3 - d)) & 0x88888888) | 0x66666666 —
//     shift + and-or per eight elements.  Any mapping works as long as both operands use the same one (a sum over k).
#include <type_traits>

#include "kernels.h"

namespace stvo {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

#ifndef MF_TPB
#define MF_TPB 2  // train tiles per workgroup barrier
#endif
#ifndef MF_FOLD_SLOTS
#define MF_FOLD_SLOTS 5  // vector instructions scheduled behind each matrix instruction of the forward scan
#endif
constexpr int MF_BLOCK = 256;        // 4 waves
constexpr int MF_TILE = 32;          // train rows per tile
constexpr int MF_KEY_SHIFT = 13;     // key = (h << 13) + j - MF_KEY_BIAS, j < 8192
constexpr int MF_KEY_BIAS = 1 << 20;
constexpr int MF_NO_KEY = 0x7FFFFFFF;
constexpr int MF_NO_KEY_MIN = 0x40000000;  // the sentinel after any number of per-tile rebasings (<= 256 x 32)

// FP4: the 32 bits of a descriptor word as 32 e2m1 nibbles, 0b0110 (+4) for a clear bit, 0b1110 (-4) for a set one
__device__ __forceinline__ v4i expand_fp4(uint32_t w) {
    v4i r;
    r.x = (int)(((w << 3) & 0x88888888u) | 0x66666666u);
    r.y = (int)(((w << 2) & 0x88888888u) | 0x66666666u);
    r.z = (int)(((w << 1) & 0x88888888u) | 0x66666666u);
    r.w = (int)((w & 0x88888888u) | 0x66666666u);
    return r;
}
__device__ __forceinline__ float med3_f32(float a, float b, float c) {
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min_f32(float a, float b) {  // (no canonicalisation of the operands: they are exact integers)
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float min3_f32(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
constexpr float MF_NO_KEY_F = 1.0e30f;      // the float form's sentinel: rebasing by -32 never changes it
constexpr float MF_NO_KEY_MIN_F = 1.0e29f;
constexpr int MF_SCALE_2_4 = 127 + 4;       // E8M0 block scale 2^4 on both operands: (+-4 * 16)^2 = 4096

// tile-relative key -> K1's (distance << 16) | train index; base = first train row of the tile the key is relative to
__device__ __forceinline__ uint32_t key_to_knn(int key, int base) {
    if (key >= MF_NO_KEY_MIN) return 0xFFFFFFFFu;
    const uint32_t u = (uint32_t)(key + base + MF_KEY_BIAS);
    return ((u >> MF_KEY_SHIFT) << 16) | (u & ((1u << MF_KEY_SHIFT) - 1u));
}

// MODE 0: every row of the direction's query side against every row of its train side (the forward scan).
// MODE 1: the query rows listed in qsel (front of the per-frame list, or its back when qsel_from_back) against every train row.
// MODE 2: as 1, and the train rows are the ones listed in tsel; the indices in the keys are then positions in tsel.
// (the reverse check of the claimed columns, match_kernels.hip; separate instantiations also keep the uses apart in traces)
template <int QB, int MODE>
__global__ __launch_bounds__(MF_BLOCK, (QB == 1 ? 4 : QB == 2 ? 3 : 1)) void hamming_knn2_mfma_kernel(int B, int tiles, int ndir, int dir0, int nseg, int row_stride,
                                                                      const uint8_t* __restrict__ d1,
                                                                      const int32_t* __restrict__ n1,
                                                                      const uint8_t* __restrict__ d2,
                                                                      const int32_t* __restrict__ n2,
                                                                      uint2* __restrict__ knn12, uint2* __restrict__ knn21,
                                                                      const int32_t* __restrict__ qsel,
                                                                      const int32_t* __restrict__ nsel,
                                                                      uint32_t* __restrict__ claim_init, int qsel_from_back,
                                                                      const int32_t* __restrict__ tsel,
                                                                      const int32_t* __restrict__ ntsel) {
    constexpr bool GATHER = MODE >= 1, TGATHER = MODE == 2;
    constexpr int ROWS = 4 * QB * 32;  // query rows per workgroup
    constexpr int KSTEPS = 4;  // matrix instructions (K = 64) per 32 x 32 tile of 256-bit distances
    using key_t = float;
    using acc_t = v16f;
    // one 16-byte fragment per (K step, wave half, train row): [tile parity][(kk * 2 + hf) * 32 + train row]
    constexpr int NBUF = 2 * MF_TPB;
    __shared__ v4i s_tile[NBUF][KSTEPS * 2 * 32];
    // XCD-aware block -> (frame pair, direction, tile, segment) mapping: as hamming_knn2_kernel
    const int per_frame = tiles * ndir * nseg;
    const int L = blockIdx.x;
    const int xcd = L & 7, k = L >> 3;
    int fb = k / per_frame, local = k % per_frame;
    if (MODE == 0 && ndir == 1 && tiles > 1) {
        // forward scan: the LAST query tile of a frame is the partial one (its workgroups are the shortest) — all of them go to the
        // end of the launch, so that the final, partly filled dispatch round consists of short workgroups (1024 frames of ~1650
        // rows: 6144 full workgroups = 8.0 rounds of 768 slots, then the 1024 partial ones; 0.551 -> 0.499 ms)
        const int groups = (int)(gridDim.x >> 3) / per_frame;  // frames per XCD
        const int full_pf = (tiles - 1) * nseg, full = full_pf * groups;
        if (k < full) {
            fb = k / full_pf;
            local = k % full_pf;
        } else {
            fb = (k - full) / nseg;
            local = full_pf + (k - full) % nseg;
        }
    }
    const int b = fb * 8 + xcd;
    if (b >= B) return;
    const int seg = local % nseg;
    const int dir = dir0 + (local / nseg) / tiles;
    const int tile = (local / nseg) % tiles;
    const size_t frame_off = (size_t)b * row_stride;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, hf = lane >> 5;
    const int q_base = tile * ROWS;
    const uint32_t* __restrict__ Q = reinterpret_cast<const uint32_t*>((dir == 0 ? d1 : d2) + frame_off * STVO_DESC_BYTES);
    // The query rows of the forward scan do not depend on the frame's row counts: fetch them while n1 / n2 are still on
    // their way (rows past the count are inside the frame's slot, are scanned like any other row and never stored).
    uint4 qw[QB][2];
    if (!GATHER) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int qi = min(q_base + (wv * QB + qb) * 32 + col, row_stride - 1);
            qw[qb][0] = reinterpret_cast<const uint4*>(Q)[2 * qi];
            qw[qb][1] = reinterpret_cast<const uint4*>(Q)[2 * qi + 1];
        }
    }
    const int na = n1[b], nb = n2[b];
    const int nq = GATHER ? nsel[b] : (dir == 0 ? na : nb);
    const int nt_all = TGATHER ? ntsel[b] : (dir == 0 ? nb : na);
    if (TGATHER && nt_all == 0) return;  // an empty row list: the consumer does not read this frame's results (nsel[3][b] == 0)
    const int seg_len = (((nt_all + nseg - 1) / nseg) + MF_TILE - 1) & ~(MF_TILE - 1);
    const int j0 = min(seg * seg_len, nt_all);
    const int nt = min(j0 + seg_len, nt_all);
    if (claim_init && seg == 0 && dir == dir0)
        for (int c = q_base + tid; c < min(q_base + ROWS, row_stride); c += MF_BLOCK) claim_init[frame_off + c] = 0xFFFFFFFFu;
    if (q_base >= nq) return;  // workgroup-uniform
    const uint32_t* __restrict__ T = reinterpret_cast<const uint32_t*>((dir == 0 ? d2 : d1) + frame_off * STVO_DESC_BYTES);
    uint2* __restrict__ out = (dir == 0 ? knn12 : knn21) + (size_t)seg * B * row_stride + frame_off;

    // query fragments: block qb of this wave = rows q_base + (wv * QB + qb) * 32 + col, kept for the whole scan
    const bool wave_active = q_base + wv * QB * 32 < nq;
    v4i qf[QB][KSTEPS];
    auto query_row = [&](int qb) {  // recomputed for the final store rather than kept live across the scan
        const int q = q_base + (wv * QB + qb) * 32 + col;
        const int qc = q < nq ? q : nq - 1;  // tail lanes scan a valid row and discard the result
        return GATHER ? qsel[frame_off + (qsel_from_back ? row_stride - 1 - qc : qc)] : qc;
    };
    if (GATHER) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int qi = query_row(qb);
            qw[qb][0] = reinterpret_cast<const uint4*>(Q)[2 * qi];
            qw[qb][1] = reinterpret_cast<const uint4*>(Q)[2 * qi + 1];
        }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const uint32_t w[8] = {qw[qb][0].x, qw[qb][0].y, qw[qb][0].z, qw[qb][0].w, qw[qb][1].x, qw[qb][1].y, qw[qb][1].z, qw[qb][1].w};
        // K step kk covers words 2 kk and 2 kk + 1: the lower wave half takes the first, the upper the second
        const bool up = hf != 0;  // (selects, not an indexed array: that one went through scratch)
        qf[qb][0] = expand_fp4(~(up ? w[1] : w[0]));
        qf[qb][1] = expand_fp4(~(up ? w[3] : w[2]));
        qf[qb][2] = expand_fp4(~(up ? w[5] : w[4]));
        qf[qb][3] = expand_fp4(~(up ? w[7] : w[6]));
    }
    // The train index enters through the C operand of the first matrix instruction of a tile: accumulator register r of
    // this lane belongs to tile row (r & 3) + 8 (r >> 2) + 4 hf.  Keys are therefore TILE-RELATIVE (index - first row of
    // the tile) and the running (best, second) are rebased by -32 per tile: 2 VALU ops per tile and query block instead
    // of a ninth matrix instruction per block.
    acc_t cidx;
#pragma unroll
    for (int r = 0; r < 16; ++r) cidx[r] = (key_t)((r & 3) + 8 * (r >> 2) + 4 * hf);

    const key_t no_key = MF_NO_KEY_F;
    key_t best[QB], second[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) best[qb] = second[qb] = no_key;

    // staging role of this thread: word xk of train row xr of the tile
    const int xr = tid & 31, xk = tid >> 5;
    const int ntiles = (nt - j0 + MF_TILE - 1) / MF_TILE;
    auto fetch = [&](int t) -> uint32_t {
        const int j = j0 + t * MF_TILE + xr;
        if (j >= nt) return 0u;
        const int row = TGATHER ? tsel[frame_off + j] : j;
        return T[(size_t)row * 8 + xk];
    };
    auto stage = [&](int t, uint32_t w) {
        s_tile[t & (NBUF - 1)][xk * 32 + xr] = expand_fp4(w);  // word xk = K step xk / 2, wave half xk & 1
    };
#pragma unroll
    for (int i = 0; i < MF_TPB; ++i)
        if (i < ntiles) stage(i, fetch(i));
    __syncthreads();
    // Software pipeline, depth one tile: step t issues the matrix instructions of tile t and, in their shadow, folds the
    // accumulators of tile t - 1 into (best, second) — a v_mfma occupies the matrix pipe for 8 passes while the wave's
    // VALU slots stay free.  Only the last tile of a segment can be ragged and it is folded after the loop (masked).
    auto rebase = [&]() {  // (best, second) relative to the next tile; the no-key sentinel stays far above any key
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            best[qb] -= (key_t)MF_TILE;
            second[qb] -= (key_t)MF_TILE;
        }
    };
    auto fold = [&](int qb, key_t key) {
        second[qb] = med3_f32(best[qb], second[qb], key);
        best[qb] = min_f32(best[qb], key);
    };
    auto mma = [&](const v4i& tf, const v4i& q, const acc_t& c) -> acc_t {
        const v8i a8 = {tf.x, tf.y, tf.z, tf.w, 0, 0, 0, 0}, b8 = {q.x, q.y, q.z, q.w, 0, 0, 0, 0};  // (FP4 reads four dwords)
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, MF_SCALE_2_4, 0, MF_SCALE_2_4);
    };
    auto mma_fold = [&](acc_t (&cur)[QB], acc_t (&prev)[QB], int t) {
        if (wave_active) {
            const v4i* frag = s_tile[t & (NBUF - 1)];
            const bool fold_prev = t > 0;  // tile t - 1 is a full tile here
            if (fold_prev) rebase();
            key_t node[QB][5];
            v4i tf = frag[lane];
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                v4i tf_ahead = tf;
                if (kk < KSTEPS - 1) tf_ahead = frag[(kk + 1) * 64 + lane];  // one K step ahead of its use
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) cur[qb] = mma(tf, qf[qb][kk], kk == 0 ? cidx : cur[qb]);
                if (fold_prev) {
                    // The fold as a TOURNAMENT of three-input nodes (round 5, last change).  The 16 keys of the previous tile and the
                    // running best are the 17 leaves of a tree of eight nodes; a node forms the smallest of its three inputs (v_min3,
                    // passed up) and their median (v_med3).  The root's minimum is the new best, and the new second is the smallest of
                    // the old second and the eight medians: a median is the minimum of a subtree that does not hold the overall
                    // minimum, so none is below the true second; and the true second — the smallest key that lost DIRECTLY to the
                    // winner, or the old second — is among them.  16 + 4 operations per block and tile instead of the 27 of merging
                    // three keys at a time into the pair (5 per 3 keys); the kernel is bound by vector + matrix ISSUE (rocprofv3:
                    // SQ_ACTIVE_INST_ANY of the three waves of a SIMD = 1.08 of a wave's life) and the fold is most of its vector work.
                    // The previous tile's 16 accumulator registers per block are complete, so their order is free: five operations
                    // behind each of the four matrix instructions of a block.
                    // HAZARD (tools/mfma_hazard_check.py, run by tests/test_isa_mfma_hazard.py on every build): these reads
                    // are inline asm, so the compiler inserts none of the 12 wait states an 8-pass XDL write needs before a vector read
                    // of its destination.  With QB >= 2 the distance is structural — behind the last matrix instruction of a block's
                    // previous tile come its share of the fold, the other blocks' matrix instructions and folds, the rebasing and this
                    // tile's first matrix instruction (>= 14 counted conservatively; the ISA check fails the CPU suite below 12).  With one
                    // block per wave (developer switch) only five fold operations and the rebasing lie between: an explicit fence, tied to
                    // the registers, as in the reverse kernel.
                    if (QB == 1 && kk == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" : "+v"(prev[0]));
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        const acc_t& p = prev[qb];
                        key_t(&a)[5] = node[qb];
                        if (kk == 0) {
                            a[0] = min3_f32(p[0], p[1], p[2]);
                            a[1] = min3_f32(p[3], p[4], p[5]);
                            second[qb] = min3_f32(second[qb], med3_f32(p[0], p[1], p[2]), med3_f32(p[3], p[4], p[5]));
                        } else if (kk == 1) {
                            a[2] = min3_f32(p[6], p[7], p[8]);
                            a[3] = min3_f32(p[9], p[10], p[11]);
                            second[qb] = min3_f32(second[qb], med3_f32(p[6], p[7], p[8]), med3_f32(p[9], p[10], p[11]));
                        } else if (kk == 2) {
                            a[4] = min3_f32(p[12], p[13], p[14]);
                            const key_t b0 = min3_f32(a[0], a[1], a[2]);
                            second[qb] = min3_f32(second[qb], med3_f32(p[12], p[13], p[14]), med3_f32(a[0], a[1], a[2]));
                            a[0] = b0;
                        } else {
                            const key_t b1 = min3_f32(a[3], a[4], p[15]), n1 = med3_f32(a[3], a[4], p[15]);
                            second[qb] = min3_f32(second[qb], n1, med3_f32(a[0], b1, best[qb]));
                            best[qb] = min3_f32(a[0], b1, best[qb]);
                        }
                    }
                }
                if (kk < KSTEPS - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // 1 LDS read (the next step's fragment)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {  // (one instruction, then its share of the fold: with FP4 operands a matrix instruction
                                                   //  lasts about as long as the 8 fold operations of one block take to issue — 0.418 ->
                                                   //  0.401 ms per 1024 frames against QB instructions followed by all of the fold)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         // one matrix instruction
                    __builtin_amdgcn_sched_group_barrier(0x002, MF_FOLD_SLOTS, 0);             // its shadow: a block's share of the fold (20 ops + rebasing)
                }
                tf = tf_ahead;
            }
        }
    };
    acc_t accA[QB], accB[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[qb][r] = accB[qb][r] = 0;
    // MF_TPB tiles per workgroup barrier: while tiles t .. t + MF_TPB - 1 are multiplied (2 MF_TPB LDS buffers), the next MF_TPB are
    // fetched and staged.  With FP4 operands a tile is short (4 matrix instructions per block) and the four waves of a workgroup
    // met at a barrier every ~400 cycles: two tiles per barrier 0.402 -> 0.364 ms per 1024 frames.
    static_assert(MF_TPB % 2 == 0, "the accumulator sets alternate tile by tile");
    int t = 0;
    // the train words travel TWO barrier groups ahead of their tiles (registers only: the LDS buffers are staged as before; round 5:
    // 0.3256 -> 0.3227 ms per 1024 frames).  What else round 5 measured on this loop, all without effect or worse: LDS reads truly one K
    // step ahead (lgkmcnt(1) instead of (0) in front of every pair of matrix instructions): no change; no barrier: no change; no fetch /
    // stage / barrier at all: 0.289; (almost) no fold: 0.273 — the matrix pipe's own 0.17 is not within reach of this skeleton; workgroups
    // persistent over the query tiles of a frame, next tile's rows requested under the drain: 0.40 (168 VGPRs, address reloads in the loop),
    // and as slow with ONE tile per workgroup, i.e. the per-workgroup prologue is not what is missing either.
    uint32_t w_near[MF_TPB];
#pragma unroll
    for (int i = 0; i < MF_TPB; ++i) {
        w_near[i] = 0;
        if (MF_TPB + i < ntiles) w_near[i] = fetch(MF_TPB + i);
    }
    for (; t + MF_TPB <= ntiles; t += MF_TPB) {
        uint32_t w_far[MF_TPB];
#pragma unroll
        for (int i = 0; i < MF_TPB; ++i) {
            w_far[i] = 0;
            if (t + 2 * MF_TPB + i < ntiles) w_far[i] = fetch(t + 2 * MF_TPB + i);
        }
#pragma unroll
        for (int i = 0; i < MF_TPB; ++i) {
            if ((i & 1) == 0) mma_fold(accA, accB, t + i);
            else mma_fold(accB, accA, t + i);
        }
#pragma unroll
        for (int i = 0; i < MF_TPB; ++i) {
            if (t + MF_TPB + i < ntiles) stage(t + MF_TPB + i, w_near[i]);
            w_near[i] = w_far[i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MF_TPB - 1; ++i)  // the last, partly filled group: its tiles are staged already (t is even here)
        if (t + i < ntiles) {
            if ((i & 1) == 0) mma_fold(accA, accB, t + i);
            else mma_fold(accB, accA, t + i);
        }
    if (wave_active && ntiles > 0) {  // drain: the last tile, rows past the segment end carry no key
        const int jt = j0 + (ntiles - 1) * MF_TILE;
        const acc_t(&last)[QB] = (ntiles & 1) ? accA : accB;
        rebase();
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jr = jt + (r & 3) + 8 * (r >> 2) + 4 * hf;
                fold(qb, jr < nt ? last[qb][r] : no_key);
            }
    }
    if (!wave_active) return;
    // the two wave halves hold the same query columns over different train rows: merge, lower half writes
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        // (the f32 keys are exact integers below 2^23: from here on as int)
        const int bi = best[qb] >= MF_NO_KEY_MIN_F ? MF_NO_KEY : (int)best[qb];
        const int si = second[qb] >= MF_NO_KEY_MIN_F ? MF_NO_KEY : (int)second[qb];
        const int ob = __shfl_xor(bi, 32), os = __shfl_xor(si, 32);
        const int hi = max(bi, ob);
        const int sec = min(min(si, os), hi);
        const int bst = min(bi, ob);
        const int q = q_base + (wv * QB + qb) * 32 + col;
        const int base = j0 + (ntiles - 1) * MF_TILE;  // the frame of the last rebasing
        if (hf == 0 && q < nq) out[query_row(qb)] = make_uint2(key_to_knn(bst, base), key_to_knn(sec, base));
    }
}

// ---- the reverse check of the claimed columns (match_kernels.hip: forward_plan_kernel) in ONE launch -------------------------------
// Two lists per frame pair: the LIGHT columns, to be compared with the rows of S (tsel), and the HEAVY columns (back of the list), to
// be compared with ALL prev rows.  A list's train side is cut into rev_segments(rows) segments of at most ~256 rows; a UNIT of work is
// (list, segment): its train rows are made RESIDENT — requested in one go (eight words per thread in flight behind their list entries),
// expanded into LDS once, ONE barrier — and then every wave runs ITS column blocks (32 columns each) against the resident tiles without
// meeting the others again: the next block's rows are requested (list entries two blocks ahead, rows one block ahead) while the
// current block is multiplied.  The general scan above, used for this in rounds 3 - 4 (one launch per list), pays one global fetch per
// barrier and per two tiles and ran at the latency of that fetch (round 5, 512 clustered frames: 97 heavy columns per frame took
// 113 us as one 2000-row scan per frame, the 1340 light ones 48 us against |S| = 178 rows; 0.186 ms for the whole check).
// `slots` workgroups per frame pair share the units: column group g (4 blocks, one per wave) of segment seg belongs to slot
// (g + seg) mod slots — the groups of a long light list spread over the slots, and so do the segments of a short heavy list.  A frame
// whose plan left nothing to scan (i.i.d. rows: S is empty) costs `slots` workgroups that read four counters and leave.
// A segment longer than a chunk (more than 2048 train rows) takes several chunks, one after the other.
// Result: m12[claimant] = -1 for every listed column whose claim some other row blocks (each train range decides for itself: the
// question is a union over ranges; forward_plan_kernel wrote m12[claimant] = column before) — no top-2 array leaves the kernel.
constexpr int REV_CHUNK_TILES = 8, REV_CHUNK = REV_CHUNK_TILES * MF_TILE;

__global__ __launch_bounds__(MF_BLOCK, 4) void hamming_knn2_mfma_reverse_kernel(int B, int slots, int row_stride, const uint8_t* __restrict__ d1,
                                                                                const int32_t* __restrict__ n1, const uint8_t* __restrict__ d2,
                                                                                const int32_t* __restrict__ qsel, const int32_t* __restrict__ nsel,
                                                                                const int32_t* __restrict__ tsel, const uint32_t* __restrict__ claim,
                                                                                float nnr, int32_t* __restrict__ m12) {
    constexpr int KSTEPS = 4;
    using key_t = float;
    using acc_t = v16f;
    __shared__ v4i s_tile[REV_CHUNK_TILES][KSTEPS * 2 * 32];  // 32 KB: every tile of the chunk
    __shared__ uint16_t s_thr[257];                           // block_threshold(d0), d0 = 0 .. 256: the largest distance that still blocks a claim at d0
    const int L = blockIdx.x;
    const int xcd = L & 7, k = L >> 3;
    const int b = (k / slots) * 8 + xcd, slot = k % slots;
    if (b >= B) return;
    const int na = n1[b];
    const int nl = nsel[(size_t)B + b], nh = nsel[2 * (size_t)B + b], ns = nsel[3 * (size_t)B + b];
    if ((ns == 0 || nl == 0) && nh == 0) return;  // (an empty S: nothing can block a light column any more)
    {   // match_kernels.hip: block_threshold — the largest d' for which float(d0) < float(d') * nnr is false
        const float f0 = (float)threadIdx.x;
        uint32_t thr = threadIdx.x;
        while (thr < 256u && !(f0 < (float)(thr + 1u) * nnr)) ++thr;
        s_thr[threadIdx.x] = (uint16_t)thr;
        if (threadIdx.x == 0) s_thr[256] = 256;  // (a claim at distance 256 — possible with nnr > 1 only — is blocked by any other row: nothing is farther)
    }   // (read after the first barrier below)
    const size_t frame_off = (size_t)b * row_stride;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, hf = lane >> 5;
    const int xr = tid & 31, xk = tid >> 5;  // staging role: word xk of train row xr of a tile
    const uint32_t* __restrict__ Q = reinterpret_cast<const uint32_t*>(d2 + frame_off * STVO_DESC_BYTES);  // the columns: curr rows
    const uint32_t* __restrict__ T = reinterpret_cast<const uint32_t*>(d1 + frame_off * STVO_DESC_BYTES);  // against prev rows
    acc_t cidx;
#pragma unroll
    for (int r = 0; r < 16; ++r) cidx[r] = (key_t)((r & 3) + 8 * (r >> 2) + 4 * hf);
    auto mma = [&](const v4i& tf, const v4i& q, const acc_t& c) -> acc_t {
        const v8i a8 = {tf.x, tf.y, tf.z, tf.w, 0, 0, 0, 0}, b8 = {q.x, q.y, q.z, q.w, 0, 0, 0, 0};
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, MF_SCALE_2_4, 0, MF_SCALE_2_4);
    };

    for (int list = 0; list < 2; ++list) {
        const bool light = list == 0;
        const int nq = light ? (ns > 0 ? nl : 0) : nh, nt_all = light ? ns : na;
        if (nq == 0) continue;
        const int nseg = rev_segments(nt_all);
        const int n_groups = (nq + 127) / 128;  // 4 blocks of 32 columns, one per wave
        const int seg_len = (((nt_all + nseg - 1) / nseg) + MF_TILE - 1) & ~(MF_TILE - 1);
        auto column_of = [&](int g) -> int {  // this lane's column (a position in the list) of group g; tail lanes take a valid one
            const int q = (g * 4 + wv) * 32 + col;
            return q < nq ? q : nq - 1;
        };
        auto list_entry = [&](int g) -> int { return qsel[frame_off + (light ? column_of(g) : row_stride - 1 - column_of(g))]; };
        for (int seg = 0; seg < nseg; ++seg) {
            // this workgroup's groups of the unit: g = g_first, g_first + slots, ...
            const int g_first = ((slot - seg) % slots + slots) % slots;
            if (g_first >= n_groups) continue;  // (workgroup-uniform)
            const int j0 = min(seg * seg_len, nt_all), nt = min(j0 + seg_len, nt_all);
            for (int c0 = j0; c0 < nt || c0 == j0; c0 += REV_CHUNK) {  // (an empty segment still writes "no key" for its columns)
                // ---- the chunk's train rows -> LDS
                uint32_t w[REV_CHUNK_TILES];
#pragma unroll
                for (int t = 0; t < REV_CHUNK_TILES; ++t) {
                    const int j = c0 + t * MF_TILE + xr;
                    uint32_t v = 0u;
                    if (j < nt) v = T[(size_t)(light ? tsel[frame_off + j] : j) * 8 + xk];
                    w[t] = v;
                }
                // (the first two groups' list entries travel with them)
                int e_cur = list_entry(g_first);
                int e_next = g_first + slots < n_groups ? list_entry(g_first + slots) : 0;
                const int ntc = min((nt - c0 + MF_TILE - 1) / MF_TILE, REV_CHUNK_TILES);
#pragma unroll
                for (int t = 0; t < REV_CHUNK_TILES; ++t)
                    if (t < ntc) s_tile[t][xk * 32 + xr] = expand_fp4(w[t]);
                uint4 qw0 = reinterpret_cast<const uint4*>(Q)[2 * e_cur], qw1 = reinterpret_cast<const uint4*>(Q)[2 * e_cur + 1];
                uint32_t cw = claim[frame_off + e_cur];  // (d0 << 16) | claimant of the lane's column
                __syncthreads();
                // ---- every wave: its blocks against the resident tiles
                for (int g = g_first; g < n_groups; g += slots) {
                    const bool up = hf != 0;
                    v4i qf[KSTEPS];
                    qf[0] = expand_fp4(~(up ? qw0.y : qw0.x));
                    qf[1] = expand_fp4(~(up ? qw0.w : qw0.z));
                    qf[2] = expand_fp4(~(up ? qw1.y : qw1.x));
                    qf[3] = expand_fp4(~(up ? qw1.w : qw1.z));
                    const uint32_t c_claim = cw;
                    const bool block_active = (g * 4 + wv) * 32 < nq;
                    // requests one block ahead (rows) and two ahead (list entry): under way while this block is multiplied
                    if (g + slots < n_groups) {
                        qw0 = reinterpret_cast<const uint4*>(Q)[2 * e_next];
                        qw1 = reinterpret_cast<const uint4*>(Q)[2 * e_next + 1];
                        cw = claim[frame_off + e_next];
                        e_cur = e_next;
                        if (g + 2 * slots < n_groups) e_next = list_entry(g + 2 * slots);
                    }
                    if (!block_active) continue;
                    key_t best = MF_NO_KEY_F, second = MF_NO_KEY_F;
                    auto fold = [&](key_t key) {
                        second = med3_f32(best, second, key);
                        best = min_f32(best, key);
                    };
                    int last_base = c0;
                    if (ntc > 0) {
                        // software pipeline, depth one tile (as the general scan): tile t's matrix instructions are issued, tile t - 1's
                        // keys are folded in their shadow; the chunk's last tile is folded after the loop
                        auto do_tile = [&](acc_t& cur, acc_t& prev, int t) {
                            key_t nd[5];
                            const v4i* frag = s_tile[t];
                            const bool fold_prev = t > 0;
                            if (fold_prev) {
                                best -= (key_t)MF_TILE;
                                second -= (key_t)MF_TILE;
                            }
                            v4i tf = frag[lane];
#pragma unroll
                            for (int kk = 0; kk < KSTEPS; ++kk) {
                                v4i tf_ahead = tf;
                                if (kk < KSTEPS - 1) tf_ahead = frag[(kk + 1) * 64 + lane];
                                cur = mma(tf, qf[kk], kk == 0 ? cidx : cur);
                                if (fold_prev) {
                                    if (kk == 0) {
                                        // The folds read tile t - 1's accumulators through inline asm (v_min3 / v_med3 without
                                        // canonicalisation), which the compiler's hazard recogniser does not cover: nothing stops it from
                                        // scheduling them right behind the matrix instruction that writes them, and the hardware does not
                                        // interlock that read (round 5: one wrong top-2 in the adversarial test with a single instruction
                                        // between the two).  The wait states of a 16-pass XDL write -> VALU read (19), tied to the
                                        // registers, in the shadow of this tile's first matrix instruction.
                                        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" : "+v"(prev));
                                        // (the fold as a tournament of v_min3 / v_med3 nodes, as in the forward scan above)
                                        nd[0] = min3_f32(prev[0], prev[1], prev[2]);
                                        nd[1] = min3_f32(prev[3], prev[4], prev[5]);
                                        second = min3_f32(second, med3_f32(prev[0], prev[1], prev[2]), med3_f32(prev[3], prev[4], prev[5]));
                                    } else if (kk == 1) {
                                        nd[2] = min3_f32(prev[6], prev[7], prev[8]);
                                        nd[3] = min3_f32(prev[9], prev[10], prev[11]);
                                        second = min3_f32(second, med3_f32(prev[6], prev[7], prev[8]), med3_f32(prev[9], prev[10], prev[11]));
                                    } else if (kk == 2) {
                                        nd[4] = min3_f32(prev[12], prev[13], prev[14]);
                                        const key_t b0 = min3_f32(nd[0], nd[1], nd[2]);
                                        second = min3_f32(second, med3_f32(prev[12], prev[13], prev[14]), med3_f32(nd[0], nd[1], nd[2]));
                                        nd[0] = b0;
                                    } else {
                                        const key_t b1 = min3_f32(nd[3], nd[4], prev[15]), n1 = med3_f32(nd[3], nd[4], prev[15]);
                                        second = min3_f32(second, n1, med3_f32(nd[0], b1, best));
                                        best = min3_f32(nd[0], b1, best);
                                    }
                                }
                                if (kk < KSTEPS - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, MF_FOLD_SLOTS, 0);
                                tf = tf_ahead;
                            }
                        };
                        acc_t accA, accB;
#pragma unroll
                        for (int r = 0; r < 16; ++r) accA[r] = accB[r] = 0;
                        int t = 0;
                        for (; t + 2 <= ntc; t += 2) {
                            do_tile(accA, accB, t);
                            do_tile(accB, accA, t + 1);
                        }
                        if (t < ntc) do_tile(accA, accB, t);
                        acc_t& last = (ntc & 1) ? accA : accB;
                        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" : "+v"(last));  // (as above: read by asm folds)
                        best -= (key_t)MF_TILE;
                        second -= (key_t)MF_TILE;
                        last_base = c0 + (ntc - 1) * MF_TILE;  // first train row of the tile the keys are relative to
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int jr = last_base + (r & 3) + 8 * (r >> 2) + 4 * hf;
                            fold(jr < nt ? last[r] : MF_NO_KEY_F);
                        }
                    }
                    // the two wave halves hold the same columns over different train rows: merge, lower half writes
                    const int bi = best >= MF_NO_KEY_MIN_F ? MF_NO_KEY : (int)best;
                    const int si = second >= MF_NO_KEY_MIN_F ? MF_NO_KEY : (int)second;
                    const int ob = __shfl_xor(bi, 32), os = __shfl_xor(si, 32);
                    const int hi = max(bi, ob);
                    const int sec = min(min(si, os), hi);
                    const int bst = min(bi, ob);
                    // The verdict of this train range: the claim (row i*, distance d0) on the column is BLOCKED iff some other row lies within
                    // T = block_threshold(d0) — a union over segments and chunks, so every range decides for itself from its own top-2
                    // (the nearest row if it is not the claimant, else the second) and the claimant of a blocked column loses its match; writers only store -1.
                    if (hf == 0 && (g * 4 + wv) * 32 + col < nq) {
                        const uint32_t x = key_to_knn(bst, last_base), y = key_to_knn(sec, last_base);
                        if (x != 0xFFFFFFFFu) {
                            const uint32_t T = s_thr[c_claim >> 16], istar = c_claim & 0xFFFFu;
                            bool blk = false;
                            if ((x >> 16) <= T) {  // (else nothing of the range is within T)
                                const uint32_t pos = x & 0xFFFFu;
                                const uint32_t row = light ? (uint32_t)tsel[frame_off + pos] : pos;
                                blk = row != istar || (y != 0xFFFFFFFFu && (y >> 16) <= T);
                            }
                            if (blk) m12[frame_off + istar] = -1;
                        }
                    }
                }
                __syncthreads();  // the tile buffers are staged again (next chunk, next unit)
            }
        }
    }
}

void launch_hamming_knn2_mfma_reverse(hipStream_t s, int B, int row_stride, const uint8_t* d1, const int32_t* n1, const uint8_t* d2, const int32_t* qsel,
                                      const int32_t* nsel, const int32_t* tsel, const uint32_t* claim, float nnr, int32_t* m12, int slots) {
    if (B <= 0 || row_stride <= 0) return;
    const dim3 grid((unsigned)(((B + 7) / 8) * 8 * slots));
    hipLaunchKernelGGL(hamming_knn2_mfma_reverse_kernel, grid, dim3(MF_BLOCK), 0, s, B, slots, row_stride, d1, n1, d2, qsel, nsel, tsel, claim, nnr, m12);
}

int mfma_rows_per_block(int qb) { return 4 * qb * 32; }

template <int QB, int MODE>
static void launch_mf(dim3 grid, hipStream_t s, int B, int tiles, int ndir, int dir0, int nseg, int row_stride, const uint8_t* d1,
                      const int32_t* n1, const uint8_t* d2, const int32_t* n2, uint2* knn12, uint2* knn21, const int32_t* qsel,
                      const int32_t* nsel, uint32_t* claim_init, int qsel_from_back, const int32_t* tsel, const int32_t* ntsel) {
    hipLaunchKernelGGL((hamming_knn2_mfma_kernel<QB, MODE>), grid, dim3(MF_BLOCK), 0, s, B, tiles, ndir, dir0, nseg, row_stride, d1, n1, d2, n2,
                       knn12, knn21, qsel, nsel, claim_init, qsel_from_back, tsel, ntsel);
}

void launch_hamming_knn2_mfma(hipStream_t s, int B, int row_stride, int max_n, const uint8_t* d1, const int32_t* n1,
                              const uint8_t* d2, const int32_t* n2, uint2* knn12, uint2* knn21, int both_directions,
                              int dir0, const int32_t* qsel, const int32_t* nsel, int nseg, uint32_t* claim_init, int qb,
                              int qsel_from_back, const int32_t* tsel, const int32_t* ntsel) {
    if (B <= 0 || max_n <= 0) return;
    const int rows = mfma_rows_per_block(qb);
    const int tiles = (max_n + rows - 1) / rows, ndir = both_directions ? 2 : 1;
    const int groups = (B + 7) / 8;
    const dim3 grid((unsigned)(groups * 8 * tiles * ndir * nseg));
#define STVO_MF_ARGS grid, s, B, tiles, ndir, dir0, nseg, row_stride, d1, n1, d2, n2, knn12, knn21, qsel, nsel, claim_init, qsel_from_back, tsel, ntsel
    // (MODE 1 / 2 — listed query rows / listed train rows — served the reverse check until round 5, which has its own kernel above;
    //  no caller passes a list any more and only the forward form is instantiated)
    if (qsel || tsel) return;
    if (qb == 1) launch_mf<1, 0>(STVO_MF_ARGS);
    else if (qb == 2) launch_mf<2, 0>(STVO_MF_ARGS);
    else launch_mf<4, 0>(STVO_MF_ARGS);
#undef STVO_MF_ARGS
}

}  // namespace stvo
