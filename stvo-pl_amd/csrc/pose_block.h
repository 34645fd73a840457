// pose_block.h — building blocks shared by the two pose kernels (pose_kernel.hip: worker waves + a solver wave, 256 VGPRs;
// pose_kernel2.hip: every wave a worker, 128 VGPRs, row-distributed 6x6 algebra): the per-problem state in LDS, the
// workgroup-wide reductions / selections, and the serial sections of the optimizePose state machine
// (/root/reference/src/stereoFrameHandler.cpp:307-547).
#pragma once
#include "kernels.h"
#include "pose_math.h"

namespace stvo {

namespace {

constexpr int ACT_CONTINUE = 0, ACT_BREAK = 1, ACT_FAIL = 2;

struct PoseSh {
    double tot[28];
    double DT[16];   // optimiser variable
    double DT0[16];  // initial DT of optimizePose (:317-326)
    double DT1[16];  // stage-1 result DT_ (:335)
    double DTr[16];  // robust GN's saved entry pose (:441)
    double cov[36];
    double eig[6];
    double H[36];
    double g[6];
    double err, err_prev, err_out, lambda;
    double stat[4];  // scratch scalars broadcast by thread 0 (median, stdv, mean, ...)
    double s_p, s_l;
    unsigned long long xchg;  // select_kth's one-word mailbox
    int action, good, n_inl_p, n_inl_l, n_m_p, n_m_l, evals, itmp;
};

// Block-wide primitives.  The workgroup has NWORK worker waves (threads 0 .. 64*NWORK-1, they own
// the feature records) plus ONE solver wave (the last 64 threads: 6x6 algebra, SE(3), every
// data-dependent decision).  Both roles run the SAME source with W = true / false, so they execute
// identical barrier sequences by construction; with W == false a primitive only synchronises and
// reads the result.  Keeping the serial algebra in its own wave keeps it out of the register budget
// of the record-holding waves (no call-clobber spills: the first version of this kernel moved
// ~1 MB of scratch per frame pair through HBM, see profiles/r01_a_hbm_counters.txt).
template <int NWORK>
struct BlockOps {
    static constexpr int NW = NWORK;
    static constexpr int WTHREADS = NWORK * 64;

    // Wave64 sum with DPP register moves (no LDS crossbar): inclusive scan inside each 16-lane row
    // (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 carry the row totals across rows.  The
    // total lands in LANE 63.  Fixed association order => bit-reproducible.  A 64-bit value moves as
    // two 32-bit DPP movs; lanes with no source read 0 (bound_ctrl), which is neutral for a sum.
    template <int CTRL, int ROW_MASK>
    static __device__ __forceinline__ double dpp_add(double v) {
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, ROW_MASK, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
        return v + __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
    }
    static __device__ __forceinline__ double wave_sum_lane63(double v) {
        v = dpp_add<0x111, 0xf>(v);  // row_shr:1
        v = dpp_add<0x112, 0xf>(v);  // row_shr:2
        v = dpp_add<0x114, 0xf>(v);  // row_shr:4
        v = dpp_add<0x118, 0xf>(v);  // row_shr:8
        v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
        v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
        return v;
    }
    // xor-butterfly (LDS crossbar): every lane ends with the wave total; used for the small reductions
    static __device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    }
    static __device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    }

    // Reduce-scatter steps on gfx950's lane-swap instructions.  v_permlane32_swap exchanges the upper half of one
    // register with the lower half of another, so ONE swap per 32-bit word plus one add leaves, in lanes 0..31, the
    // pair sums a[L] + a[L+32] and, in lanes 32..63, b[L-32] + b[L]: two values are folded for the price of one.
    // v_permlane16_swap does the same between the odd and even 16-lane rows.
    static __device__ __forceinline__ double fold32(double a, double b) {
        const unsigned long long ab = (unsigned long long)__double_as_longlong(a), bb = (unsigned long long)__double_as_longlong(b);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ab, (unsigned)bb, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
        return __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) +
               __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
    }
    static __device__ __forceinline__ double fold16(double a, double b) {
        const unsigned long long ab = (unsigned long long)__double_as_longlong(a), bb = (unsigned long long)__double_as_longlong(b);
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ab, (unsigned)bb, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
        return __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) +
               __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
    }

    // 28-vector block sum -> sh->tot[0..27] (wave partials summed in wave order by the solver wave).
    // Inside a wave: 28 values -> 14 (fold32) -> 7 (fold16) per lane, then a 16-lane row scan of those 7; row r
    // ends up with the wave totals of values 7r .. 7r+6 in its last lane.  147 VALU ops instead of the 504 of 28
    // independent 64-lane scans; fixed association order => bit-reproducible.
    template <bool W>
    static __device__ __forceinline__ void sum28_fold(double* acc, double (*red)[28]) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (W) {
            double s14[14], s7[7];
#pragma unroll
            for (int k = 0; k < 14; ++k) s14[k] = fold32(acc[k], acc[14 + k]);
#pragma unroll
            for (int k = 0; k < 7; ++k) s7[k] = fold16(s14[k], s14[7 + k]);
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                double v = s7[k];
                v = dpp_add<0x111, 0xf>(v);  // row_shr:1
                v = dpp_add<0x112, 0xf>(v);  // row_shr:2
                v = dpp_add<0x114, 0xf>(v);  // row_shr:4
                v = dpp_add<0x118, 0xf>(v);  // row_shr:8
                if ((lane & 15) == 15) red[wv][(lane >> 4) * 7 + k] = v;
            }
        }
    }
    template <bool W>
    static __device__ __forceinline__ void sum28_finish(double (*red)[28], PoseSh* sh) {
        const int lane = threadIdx.x & 63;
        __syncthreads();
        if (!W) {
            if (lane < 28) {
                double s = red[0][lane];
#pragma unroll
                for (int w = 1; w < NW; ++w) s += red[w][lane];
                sh->tot[lane] = s;
            }
            // tot[] is consumed by lane 0 of THIS wave only (t0_unpack), so a wave-level fence is enough; the
            // worker waves run ahead to the barrier that follows the solver's algebra, which also keeps them from
            // overwriting `red` before it has been read here.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }

    template <int N, bool W>
    static __device__ __forceinline__ void sum_small(const double* v, double (*red)[28], double* out) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (W) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double s = wave_sum(v[k]);
                if (lane == 0) red[wv][k] = s;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double s = red[0][k];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w][k];
            out[k] = s;
        }
        __syncthreads();
    }

    template <bool W>
    static __device__ __forceinline__ int sum_int(int v, int* ired) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (W) {
            const int s = wave_sum_i(v);
            if (lane == 0) ired[wv] = s;
        }
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += ired[w];
        __syncthreads();
        return t;
    }

    // exclusive scan of per-thread counts (worker thread order); returns this thread's offset
    template <bool W>
    static __device__ __forceinline__ int excl_scan(int count, int* ired, int* total) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int incl = count;
        if (W) {
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            if (lane == 63) ired[wv] = incl;
        }
        __syncthreads();
        int base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int c = ired[w];
            if (w < wv) base += c;
            tot += c;
        }
        __syncthreads();
        *total = tot;
        return base + incl - count;
    }

    // workgroup-wide totals of three per-wave counts (wave-uniform c[0..2]); double-buffered partials => ONE barrier per call
    template <bool W>
    static __device__ __forceinline__ void count3_db(int* c, int (*ibuf)[3 * NW], int parity) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (W && lane == 0) {
            ibuf[parity][wv] = c[0];
            ibuf[parity][NW + wv] = c[1];
            ibuf[parity][2 * NW + wv] = c[2];
        }
        __syncthreads();
        int t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            t0 += ibuf[parity][w];
            t1 += ibuf[parity][NW + w];
            t2 += ibuf[parity][2 * NW + w];
        }
        c[0] = t0;
        c[1] = t1;
        c[2] = t2;
    }

    // k-th smallest (0-based) of the n keys {key[k] : bit k of mask} held in REGISTERS across the workgroup:
    // most-significant-first radix search on two bits per round — three thresholds, three workgroup-wide counts (per-wave
    // counts from ballots + s_bcnt1, no cross-lane data movement), one barrier.  lo / hi bracket the number of keys
    // below the current prefix and below its upper end; as soon as exactly one key is left in the bracket it IS the
    // answer and is fetched directly — with n ~ 1500 distinct values that happens after ~13 of the 32 (or ~9 of the 16)
    // rounds.  Equal keys simply keep the search going to the last bit.  Same result as sorting.
    template <int N, bool W, typename K, int BITS>
    static __device__ __forceinline__ K select_kth(const K* key, unsigned mask, int n, int kth, int (*ibuf)[3 * NW], K* xchg) {
        static_assert(BITS % 2 == 0, "two bits per round");
        K res = 0;
        int lo = 0, hi = n, parity = 0;
        for (int bit = BITS - 2; bit >= 0; bit -= 2) {
            const K t1 = res | ((K)1 << bit), t2 = res | ((K)2 << bit), t3 = res | ((K)3 << bit);
            int c[3] = {0, 0, 0};
            if (W) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const bool in = (mask >> k) & 1u;
                    c[0] += __popcll(__builtin_amdgcn_ballot_w64(in && key[k] < t1));
                    c[1] += __popcll(__builtin_amdgcn_ballot_w64(in && key[k] < t2));
                    c[2] += __popcll(__builtin_amdgcn_ballot_w64(in && key[k] < t3));
                }
            }
            count3_db<W>(c, ibuf, parity);
            parity ^= 1;
            // the largest threshold with at most kth keys below it becomes the new prefix
            if (c[2] <= kth) {
                res = t3;
                lo = c[2];
            } else if (c[1] <= kth) {
                res = t2;
                lo = c[1];
                hi = c[2];
            } else if (c[0] <= kth) {
                res = t1;
                lo = c[0];
                hi = c[1];
            } else {
                hi = c[0];
            }
            if (hi - lo == 1 && bit > 0) {  // block-uniform: the single key in [res, res + 2^bit)
                const K top = res + (((K)1 << bit) - 1);
                if (W) {
#pragma unroll
                    for (int k = 0; k < N; ++k)
                        if (((mask >> k) & 1u) && key[k] >= res && key[k] <= top) *xchg = key[k];
                }
                __syncthreads();
                res = *xchg;
                break;
            }
        }
        __syncthreads();  // ibuf / xchg are reused by the caller
        return res;
    }

    // The same selection on EIGHT bits per round: the keys that still carry the current prefix are counted into a 256-bin LDS
    // histogram (no-return ds_add), wave 0 scans the bins (4 per lane) for the one that holds the kth key while the other
    // waves clear the histogram of the next round, everybody adopts bin and rank.  ~1300 doubles take 3 rounds + the fetch of
    // the last key instead of ~11 two-bit rounds of 24 ballots each (the outlier removal of a frame pair: 70 k -> ~35 k
    // cycles).  Integer counts only: the result is the one a sort would give.  hist: [2][HIST_W] ints (HIST_W - 256 mailbox
    // words); zeroed here, so callers need not preserve anything.
    static constexpr int HIST_W = 260;
    template <int N, bool W, typename K, int BITS>
    static __device__ __forceinline__ K select_kth_hist(const K* key, unsigned mask, int kth, int (*hist)[HIST_W], K* xchg) {
        static_assert(BITS % 8 == 0, "eight bits per round");
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        if (W)
            for (int i = tid; i < 256; i += WTHREADS) hist[0][i] = 0;
        __syncthreads();
        K prefix = 0;
        int kk = kth, parity = 0;
        for (int shift = BITS - 8; shift >= 0; shift -= 8) {
            int* h = hist[parity];
            if (W) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const K diff = key[k] ^ prefix;
                    const bool same = shift + 8 >= BITS ? true : (diff >> (shift + 8 >= BITS ? 0 : shift + 8)) == 0;
                    if (((mask >> k) & 1u) && same) atomicAdd(&h[(int)((key[k] >> shift) & 255)], 1);
                }
            }
            __syncthreads();
            if (W && wv == 0) {
                const int c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
                const int sum = c0 + c1 + c2 + c3;
                int incl = sum;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += o;
                }
                const int excl = incl - sum;
                if (excl <= kk && kk < incl) {  // exactly one lane: 0 <= kk < number of keys under the prefix
                    int bin = 4 * lane, below = excl, cnt = c0;
                    if (kk >= below + cnt) { below += cnt; cnt = c1; bin += 1;
                        if (kk >= below + cnt) { below += cnt; cnt = c2; bin += 1;
                            if (kk >= below + cnt) { below += cnt; cnt = c3; bin += 1; } } }
                    h[256] = bin;
                    h[257] = below;
                    h[258] = cnt;
                }
            }
            if (W && (NWORK == 1 || wv != 0)) {  // the next round's histogram
                const int first = NWORK == 1 ? 0 : 64, step = NWORK == 1 ? 64 : WTHREADS - 64;
                for (int i = tid - first; i < 256; i += step) hist[parity ^ 1][i] = 0;
            }
            __syncthreads();
            const int bin = h[256], below = h[257], cnt = h[258];
            prefix |= (K)bin << shift;
            kk -= below;
            parity ^= 1;
            if (cnt == 1 && shift > 0) {  // block-uniform: the single key under the prefix
                if (W) {
#pragma unroll
                    for (int k = 0; k < N; ++k)
                        if (((mask >> k) & 1u) && ((key[k] ^ prefix) >> shift) == 0) *xchg = key[k];
                }
                __syncthreads();
                prefix = *xchg;
                break;
            }
        }
        __syncthreads();  // hist / xchg are reused by the caller
        return prefix;
    }

    // 1.4826 * MAD of the n values {v[k] : bit k of mask}.  Follows vector_stdv_mad / the first half of
    // vector_mean_stdv_mad (src/auxiliar.cpp:395-404, 447-457): median = sorted[n/2]; dev = fabsf(x - median)
    // (FLOAT truncation); MAD = sorted dev[n/2].  The two std::sort calls are replaced by exact k-th-element
    // selection on the order-preserving integer images of the values.  n == 0 -> 0.
    template <int N, bool W>
    static __device__ __forceinline__ double mad_sigma(const double* v, unsigned mask, int n, int (*hist)[HIST_W],
                                                       unsigned long long* xchg) {
        if (n == 0) return 0.0;  // block-uniform
        const int kth = n / 2;
        unsigned long long key[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(v[k]);
            key[k] = (b >> 63) ? ~b : (b | 0x8000000000000000ull);  // total order of IEEE doubles
        }
        const unsigned long long res = select_kth_hist<N, W, unsigned long long, 64>(key, mask, kth, hist, xchg);
        const unsigned long long mb = (res >> 63) ? (res & 0x7FFFFFFFFFFFFFFFull) : ~res;
        const double median = __longlong_as_double((long long)mb);
        unsigned fkey[N];
#pragma unroll
        for (int k = 0; k < N; ++k) fkey[k] = __float_as_uint(fabsf((float)(v[k] - median)));  // >= 0 (or NaN)
        const unsigned fres = select_kth_hist<N, W, unsigned, 32>(fkey, mask, kth, hist, reinterpret_cast<unsigned*>(xchg));
        return 1.4826 * (double)__uint_as_float(fres);
    }
};

__device__ __forceinline__ double norm3(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// ---- row-distributed 6x6 algebra for ONE wave (pose_kernel2.hip) -------------------------------------------------------
// Lane i (0..5) of the calling wave holds row i of the matrix in 6 registers; a pivot row is broadcast with v_readlane (the
// lane index is a compile-time constant, so the values land in SGPRs and feed the other lanes' FMAs as scalar operands).
// Every lane of the wave executes the code (lanes >= 6 carry a copy of row 0 and are never read).  Same certification as
// pm::ldl6: unpivoted elimination is accepted only while every pivot is > 1e-10 of the largest diagonal entry — callers
// then fall back to the pivoted serial routines, so degenerate systems behave exactly as in pose_kernel.hip.
template <int L>
__device__ __forceinline__ double rl(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), L), hi = __builtin_amdgcn_readlane((int)(b >> 32), L);
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double own_diag(const double* h, int lane) {  // h[lane] without a dynamic register index
    double d = h[0];
#pragma unroll
    for (int j = 1; j < 6; ++j) d = (lane == j) ? h[j] : d;
    return d;
}
template <int K>
struct RowStep {  // elimination step K on rows i != K (below only when !JORDAN)
    template <bool JORDAN, int NE>
    static __device__ __forceinline__ void run(double* h, double* e, int lane, double tol, bool& ok, double* piv, double* rinv) {
        double pk[6], ek[NE > 0 ? NE : 1];
#pragma unroll
        for (int j = K; j < 6; ++j) pk[j] = rl<K>(h[j]);
#pragma unroll
        for (int j = 0; j < NE; ++j) ek[j] = rl<K>(e[j]);
        ok = ok && (pk[K] > tol);
        const double ri = 1.0 / pk[K];
        piv[K] = pk[K];
        rinv[K] = ri;
        const bool upd = JORDAN ? (lane != K) : (lane > K);
        const double fcol = h[K];
        const double f = upd ? fcol * ri : 0.0;
#pragma unroll
        for (int j = K + 1; j < 6; ++j) h[j] -= f * pk[j];
#pragma unroll
        for (int j = 0; j < NE; ++j) e[j] -= f * ek[j];
        if (JORDAN) {  // the pivot row itself is normalised at the end (by rinv), keep it untouched here
        }
    }
};

// H x = b.  h: row `lane` of H (lanes 0..5), b: its right-hand side.  x (uniform) and log|det H|; false => not certified.
__device__ __forceinline__ bool row_solve_spd(const double* h_in, double b_in, double* x, double* log_abs_det) {
    const int lane = threadIdx.x & 63;
    double h[6], e[1] = {b_in}, piv[6], rinv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) h[j] = h_in[j];
    const double dg = fabs(own_diag(h, lane));
    double maxd = rl<0>(dg);
    maxd = fmax(maxd, rl<1>(dg)); maxd = fmax(maxd, rl<2>(dg)); maxd = fmax(maxd, rl<3>(dg));
    maxd = fmax(maxd, rl<4>(dg)); maxd = fmax(maxd, rl<5>(dg));
    const double d0 = rl<0>(dg), d1 = rl<1>(dg), d2 = rl<2>(dg), d3 = rl<3>(dg), d4 = rl<4>(dg), d5 = rl<5>(dg);
    const bool finite = (d0 == d0) && (d1 == d1) && (d2 == d2) && (d3 == d3) && (d4 == d4) && (d5 == d5);  // fmax drops NaNs
    const double tol = 1e-10 * maxd;
    bool ok = finite && maxd > 0.0 && maxd < 1.0e300;
    RowStep<0>::run<false, 1>(h, e, lane, tol, ok, piv, rinv);
    RowStep<1>::run<false, 1>(h, e, lane, tol, ok, piv, rinv);
    RowStep<2>::run<false, 1>(h, e, lane, tol, ok, piv, rinv);
    RowStep<3>::run<false, 1>(h, e, lane, tol, ok, piv, rinv);
    RowStep<4>::run<false, 1>(h, e, lane, tol, ok, piv, rinv);
    RowStep<5>::run<false, 1>(h, e, lane, tol, ok, piv, rinv);
    // back substitution on the upper-triangular rows: lane k owns row k
    double t;
    x[5] = rl<5>(e[0]) * rinv[5];
    t = e[0] - h[5] * x[5];
    x[4] = rl<4>(t) * rinv[4];
    t = e[0] - h[5] * x[5] - h[4] * x[4];
    x[3] = rl<3>(t) * rinv[3];
    t = e[0] - h[5] * x[5] - h[4] * x[4] - h[3] * x[3];
    x[2] = rl<2>(t) * rinv[2];
    t = e[0] - h[5] * x[5] - h[4] * x[4] - h[3] * x[3] - h[2] * x[2];
    x[1] = rl<1>(t) * rinv[1];
    t = e[0] - h[5] * x[5] - h[4] * x[4] - h[3] * x[3] - h[2] * x[2] - h[1] * x[1];
    x[0] = rl<0>(t) * rinv[0];
    if (log_abs_det) *log_abs_det = log(piv[0] * piv[1] * piv[2]) + log(piv[3] * piv[4] * piv[5]);
    return ok;
}

// Row `lane` of A^-1 (lower triangle mirrored, i.e. exactly symmetric) by Gauss-Jordan on [A | I]; false => not certified.
__device__ __forceinline__ bool row_inverse_spd(const double* h_in, double* inv_row) {
    const int lane = threadIdx.x & 63;
    double h[6], e[6], piv[6], rinv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        h[j] = h_in[j];
        e[j] = (lane == j) ? 1.0 : 0.0;
    }
    const double dg = fabs(own_diag(h, lane));
    const double d0 = rl<0>(dg), d1 = rl<1>(dg), d2 = rl<2>(dg), d3 = rl<3>(dg), d4 = rl<4>(dg), d5 = rl<5>(dg);
    const double maxd = fmax(fmax(fmax(d0, d1), fmax(d2, d3)), fmax(d4, d5));
    const bool finite = (d0 == d0) && (d1 == d1) && (d2 == d2) && (d3 == d3) && (d4 == d4) && (d5 == d5);
    const double tol = 1e-10 * maxd;
    bool ok = finite && maxd > 0.0 && maxd < 1.0e300;
    RowStep<0>::run<true, 6>(h, e, lane, tol, ok, piv, rinv);
    RowStep<1>::run<true, 6>(h, e, lane, tol, ok, piv, rinv);
    RowStep<2>::run<true, 6>(h, e, lane, tol, ok, piv, rinv);
    RowStep<3>::run<true, 6>(h, e, lane, tol, ok, piv, rinv);
    RowStep<4>::run<true, 6>(h, e, lane, tol, ok, piv, rinv);
    RowStep<5>::run<true, 6>(h, e, lane, tol, ok, piv, rinv);
    double ri = rinv[0];  // the row's own pivot reciprocal
#pragma unroll
    for (int j = 1; j < 6; ++j) ri = (lane == j) ? rinv[j] : ri;
#pragma unroll
    for (int j = 0; j < 6; ++j) e[j] *= ri;
    // mirror the lower triangle: element (i, j), j > i, := element (j, i) held by lane j
    const double m01 = rl<1>(e[0]), m02 = rl<2>(e[0]), m03 = rl<3>(e[0]), m04 = rl<4>(e[0]), m05 = rl<5>(e[0]);
    const double m12 = rl<2>(e[1]), m13 = rl<3>(e[1]), m14 = rl<4>(e[1]), m15 = rl<5>(e[1]);
    const double m23 = rl<3>(e[2]), m24 = rl<4>(e[2]), m25 = rl<5>(e[2]);
    const double m34 = rl<4>(e[3]), m35 = rl<5>(e[3]);
    const double m45 = rl<5>(e[4]);
    if (lane == 0) { e[1] = m01; e[2] = m02; e[3] = m03; e[4] = m04; e[5] = m05; }
    if (lane == 1) { e[2] = m12; e[3] = m13; e[4] = m14; e[5] = m15; }
    if (lane == 2) { e[3] = m23; e[4] = m24; e[5] = m25; }
    if (lane == 3) { e[4] = m34; e[5] = m35; }
    if (lane == 4) { e[5] = m45; }
#pragma unroll
    for (int j = 0; j < 6; ++j) inv_row[j] = e[j];
    return ok;
}

// pm::spd_unit_certificate on rows: s = row `lane` of the symmetric matrix given by the LOWER triangle of C.
__device__ __forceinline__ int row_spd_unit_certificate(const double* s_in) {
    const int lane = threadIdx.x & 63;
    double h[6], e[1] = {0.0}, piv[6], rinv[6];
    double rs = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        h[j] = s_in[j];
        rs += fabs(h[j]);
    }
    const double dgs = own_diag(h, lane);
    double rmax = 0.0, dmax_ = 0.0;
    {
        const double r0 = rl<0>(rs), r1 = rl<1>(rs), r2 = rl<2>(rs), r3 = rl<3>(rs), r4 = rl<4>(rs), r5 = rl<5>(rs);
        const double g0 = rl<0>(dgs), g1 = rl<1>(dgs), g2 = rl<2>(dgs), g3 = rl<3>(dgs), g4 = rl<4>(dgs), g5 = rl<5>(dgs);
        const double rr[6] = {r0, r1, r2, r3, r4, r5}, gg[6] = {g0, g1, g2, g3, g4, g5};
#pragma unroll
        for (int i = 0; i < 6; ++i) {  // the comparison chain of pm::spd_unit_certificate (NaN stays out of the maxima)
            rmax = rr[i] > rmax ? rr[i] : rmax;
            dmax_ = gg[i] > dmax_ ? gg[i] : dmax_;
        }
        bool nan = false;
#pragma unroll
        for (int i = 0; i < 6; ++i) nan = nan || !(rr[i] == rr[i]);
        if (nan) return 0;
    }
    if (!(rmax <= 1.0)) return 0;
    const double tol = 1e-10 * dmax_;
    bool ok = dmax_ > 0.0;
    RowStep<0>::run<false, 0>(h, e, lane, tol, ok, piv, rinv);
    RowStep<1>::run<false, 0>(h, e, lane, tol, ok, piv, rinv);
    RowStep<2>::run<false, 0>(h, e, lane, tol, ok, piv, rinv);
    RowStep<3>::run<false, 0>(h, e, lane, tol, ok, piv, rinv);
    RowStep<4>::run<false, 0>(h, e, lane, tol, ok, piv, rinv);
    RowStep<5>::run<false, 0>(h, e, lane, tol, ok, piv, rinv);
    return ok ? 1 : 0;
}

// ---- solver sections.  ROW == false: executed by ONE lane (the solver lane of pose_kernel.hip).  ROW == true: executed by
// EVERY lane of one wave with identical data (pose_kernel2.hip): the scalar logic runs redundantly, the 6x6 systems are
// solved on rows; LDS stores of identical values from all lanes are benign. ----

__device__ __forceinline__ void t0_unpack(PoseSh* sh) {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) {
            const double v = sh->tot[k++];
            sh->H[i * 6 + j] = v;
            sh->H[j * 6 + i] = v;
        }
#pragma unroll
    for (int i = 0; i < 6; ++i) sh->g[i] = sh->tot[21 + i];
    sh->err = sh->tot[27] / (double)(sh->n_inl_l + sh->n_inl_p);  // :692  (0/0 -> NaN)
}

// H inc = g (ColPivHouseholderQR::solve at :417-418 and siblings): LDL^T when H is certified positive definite
// (the normal case), the pivoted QR otherwise — see pose_math.h.
template <bool ROW>
__device__ __forceinline__ void solve_normal_eq(PoseSh* sh, double* inc, double* log_abs_det) {
    if (ROW) {
        const int lane = threadIdx.x & 63, r = lane < 6 ? lane : 0;
        double h[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) h[j] = sh->H[r * 6 + j];
        if (row_solve_spd(h, sh->g[r], inc, log_abs_det)) return;  // wave-uniform verdict
    }
    double H[36], g[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) H[i] = sh->H[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = sh->g[i];
    if (ROW || !pm::solve6_spd(H, g, inc, log_abs_det)) pm::solve6(H, g, inc, log_abs_det);
}

// body of gaussNewtonOptimization after optimizeFunctions (:405-428)
template <bool ROW>
__device__ __forceinline__ void t0_gn_iter(PoseSh* sh, double min_error, double min_error_change, int it) {
    t0_unpack(sh);
    const double err = sh->err;
    if (err > sh->err_prev) {
        sh->action = it > 0 ? ACT_BREAK : ACT_FAIL;
        return;
    }
    if ((err < min_error) || fabs(err - sh->err_prev) < min_error_change) {
        sh->action = ACT_BREAK;
        return;
    }
    double inc[6], DT[16];
    solve_normal_eq<ROW>(sh, inc, nullptr);
#pragma unroll
    for (int i = 0; i < 16; ++i) DT[i] = sh->DT[i];
    pm::step_pose(DT, inc);
#pragma unroll
    for (int i = 0; i < 16; ++i) sh->DT[i] = DT[i];
    if (norm3(inc) < min_error_change && norm3(inc + 3) < min_error_change) {
        sh->action = ACT_BREAK;
        return;
    }
    sh->err_prev = err;
    sh->action = ACT_CONTINUE;
}

// body of gaussNewtonOptimizationRobust after optimizeFunctionsRobust (:449-467)
template <bool ROW>
__device__ __forceinline__ void t0_gnr_iter(PoseSh* sh, double min_error, double min_error_change) {
    t0_unpack(sh);
    const double err = sh->err;
    if (fabs(err - sh->err_prev) < min_error_change || err < min_error) {
        sh->action = ACT_BREAK;
        return;
    }
    double inc[6], DT[16], lad;
    solve_normal_eq<ROW>(sh, inc, &lad);
    if (lad < 0.0) {
        sh->good = 0;
        sh->action = ACT_BREAK;
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) DT[i] = sh->DT[i];
    pm::step_pose(DT, inc);
#pragma unroll
    for (int i = 0; i < 16; ++i) sh->DT[i] = DT[i];
    double n6 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) n6 += inc[i] * inc[i];
    if (sqrt(n6) < min_error_change) {
        sh->action = ACT_BREAK;
        return;
    }
    sh->err_prev = err;
    sh->action = ACT_CONTINUE;
}

// LM first iteration (:486-510) and loop body (:518-542)
template <bool ROW>
__device__ __forceinline__ void t0_lm_iter(PoseSh* sh, double min_error, double min_error_change, int first) {
    t0_unpack(sh);
    const double err = sh->err;
    double inc[6], DT[16];
    if (first) {
        double Hmax = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double h = sh->H[i * 7];
            if (h > Hmax || h < -Hmax) Hmax = fabs(h);
        }
        sh->lambda = 0.000000001 * Hmax;
#pragma unroll
        for (int i = 0; i < 6; ++i) sh->H[i * 7] += sh->lambda;
        solve_normal_eq<ROW>(sh, inc, nullptr);
#pragma unroll
        for (int i = 0; i < 16; ++i) DT[i] = sh->DT[i];
        pm::step_pose(DT, inc);
#pragma unroll
        for (int i = 0; i < 16; ++i) sh->DT[i] = DT[i];
        sh->err_prev = err;
        sh->action = ACT_CONTINUE;
        return;
    }
    if (fabs(err - sh->err_prev) < min_error_change || err < min_error) {
        sh->action = ACT_BREAK;
        return;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) sh->H[i * 7] += sh->lambda;
    solve_normal_eq<ROW>(sh, inc, nullptr);
    if (err > sh->err_prev)
        sh->lambda /= 4.0;
    else {
        sh->lambda *= 4.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) DT[i] = sh->DT[i];
        pm::step_pose(DT, inc);
#pragma unroll
        for (int i = 0; i < 16; ++i) sh->DT[i] = DT[i];
    }
    if (norm3(inc) < min_error_change && norm3(inc + 3) < min_error_change) {
        sh->action = ACT_BREAK;
        return;
    }
    sh->err_prev = err;
    sh->action = ACT_CONTINUE;
}

template <bool ROW>
__device__ __forceinline__ void t0_cov_from_H(PoseSh* sh) {
    if (ROW) {
        const int lane = threadIdx.x & 63, r = lane < 6 ? lane : 0;
        double h[6], ir[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) h[j] = sh->H[r * 6 + j];
        if (row_inverse_spd(h, ir)) {  // wave-uniform verdict
            if (lane < 6) {
#pragma unroll
                for (int j = 0; j < 6; ++j) sh->cov[lane * 6 + j] = ir[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            return;
        }
    }
    double H[36], Hi[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) H[i] = sh->H[i];
    if (ROW || !pm::inverse6_spd(H, Hi)) pm::inverse6(H, Hi);  // Matrix6d::inverse(), :429 / :470 / :545
#pragma unroll
    for (int i = 0; i < 36; ++i) sh->cov[i] = Hi[i];
}

// isGoodSolution(DT, cov, err) -> sh->good; eigenvalues left in sh->eig
__device__ __forceinline__ void t0_is_good(PoseSh* sh, const double* DT, double err) {
    double C[36], w[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) C[i] = sh->cov[i];
    pm::eig6_ql(C, w);  // SelfAdjointEigenSolver::eigenvalues(), :294-295
#pragma unroll
    for (int i = 0; i < 6; ++i) sh->eig[i] = w[i];
    sh->good = pm::is_good_solution(DT, w, err) ? 1 : 0;
}

// Same decision without the eigenvalues (the stage-1 test of :341 only needs the verdict): positive
// definiteness by LDL^T pivots and lambda_max <= ||.||_inf <= 1 certify the two eigenvalue conditions; anything
// not certified falls back to the eigen-decomposition the reference performs.
template <bool ROW>
__device__ __forceinline__ void t0_is_good_fast(PoseSh* sh, const double* DT, double err) {
    if (err < 0.0 || err > 1.0 || !pm::all_finite16(DT)) {
        sh->good = 0;
        return;
    }
    if (ROW) {
        const int lane = threadIdx.x & 63, r = lane < 6 ? lane : 0;
        double srow[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) srow[j] = (j <= r) ? sh->cov[r * 6 + j] : sh->cov[j * 6 + r];
        if (row_spd_unit_certificate(srow) == 1) {
            sh->good = 1;
            return;
        }
        t0_is_good(sh, DT, err);
        return;
    }
    double C[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) C[i] = sh->cov[i];
    if (pm::spd_unit_certificate(C) == 1) {
        sh->good = 1;
        return;
    }
    t0_is_good(sh, DT, err);
}

__device__ __forceinline__ void t0_commit(PoseSh* sh, stvo_pose_result* out, int status, int path, int it0, int it1) {
    // :372-391
    t0_is_good(sh, sh->DT, sh->err_out);
    double DT[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        DT[i] = sh->DT[i];
        out->T_opt[i] = DT[i];
    }
    out->err_opt = sh->err_out;
    if (sh->good && !pm::is_identity16(DT)) {
        double Ti[16], x[6], T[16];
        pm::inverse_se3(DT, Ti);
        pm::logmap_se3(Ti, x);
        pm::expmap_se3(x, T);
#pragma unroll
        for (int i = 0; i < 16; ++i) out->T[i] = T[i];
#pragma unroll
        for (int i = 0; i < 36; ++i) out->cov[i] = sh->cov[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) out->cov_eig[i] = sh->eig[i];
        out->err = sh->err_out;
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) out->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < 36; ++i) out->cov[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) out->cov_eig[i] = 0.0;
        out->err = -1.0;
        if (status == STVO_POSE_OK) status = STVO_POSE_REJECTED;
    }
    out->status = status;
    out->path = path;
    out->iters[0] = it0;
    out->iters[1] = it1;
    out->n_matched_pt = sh->n_m_p;
    out->n_matched_ls = sh->n_m_l;
    out->n_inliers_pt = sh->n_inl_p;
    out->n_inliers_ls = sh->n_inl_l;
}

}  // namespace
}  // namespace stvo
