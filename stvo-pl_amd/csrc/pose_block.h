// pose_block.h — building blocks shared by the two pose kernels (pose_kernel.hip: worker waves + a solver wave, the latency
// formulation; pose_kernel2p.hip: every wave a worker, thread-private records, the batch formulation): the per-problem state
// in LDS, the workgroup-wide reductions / selections, and the serial sections of the optimizePose state machine
// (/root/reference/src/stereoFrameHandler.cpp:307-547).
#pragma once
#include <type_traits>

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

    // k-th smallest (0-based) of the keys {key[k] : bit k of mask} held in REGISTERS across the workgroup, most significant
    // byte first: the keys that still carry the current prefix are counted into a 256-bin LDS
    // histogram (no-return ds_add), wave 0 scans the bins (4 per lane) for the one that holds the kth key while the other
    // waves clear the histogram of the next round, everybody adopts bin and rank.  ~1300 doubles take 3 rounds + the fetch of
    // the last key instead of ~11 two-bit rounds of 24 ballots each (the outlier removal of a frame pair: 70 k -> ~35 k
    // cycles).  Integer counts only: the result is the one a sort would give.  hist: [2][HIST_W] ints (HIST_W - 256 mailbox
    // words); zeroed here, so callers need not preserve anything.
    static constexpr int HIST_W = 260;
    // Round 4: the search starts at the most significant bit in which the keys DIFFER (block-wide AND / OR of the keys, one extra
    // exchange through the second histogram's words before the first barrier).  The keys of a selection are residual norms: doubles
    // between ~0.01 and ~100 share sign and the top exponent bits, so the first 8-bit round of a search from bit 63 put ~1500
    // ds_add on two or three bins (they serialise in the LDS unit) and decided nothing; from the first differing bit the first
    // round already spreads the keys over most of the 256 bins and the second usually isolates the answer.
    template <int N, bool W, typename K, int BITS>
    static __device__ __forceinline__ K select_kth_hist(const K* key, unsigned mask, int kth, int (*hist)[HIST_W], K* xchg) {
        static_assert(BITS == 32 || BITS == 64, "32- or 64-bit keys");
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        // (only where it pays: the 64-bit keys of a thread with many of them — the key-points of the batch kernels.  The float keys
        //  of the MAD spread over ~15 top bytes by themselves, and with a handful of keys per thread the exchange costs more than the
        //  degenerate round: the robust mode of the latency kernel, which selects at every evaluation, lost 8 % to it.)
        constexpr bool PREFIX = BITS == 64 && N >= 8;
        K k_and = 0, k_or = ~(K)0;  // !PREFIX: every bit counts as differing
        if (PREFIX) {
            k_and = ~(K)0;
            k_or = 0;
        }
        if (W) {
            if (PREFIX) {
#pragma unroll
                for (int k = 0; k < N; ++k)
                    if ((mask >> k) & 1u) {
                        k_and &= key[k];
                        k_or |= key[k];
                    }
                // wave AND / OR on DPP moves (lanes without a source keep their own value: the identity of both), result in lane 63
                auto dpp_step = [&](auto ctrl_c, auto rmask_c) {
                    constexpr int CTRL = decltype(ctrl_c)::value, RM = decltype(rmask_c)::value;
                    const unsigned alo = (unsigned)k_and, ahi = (unsigned)((unsigned long long)k_and >> 32);
                    const unsigned olo = (unsigned)k_or, ohi = (unsigned)((unsigned long long)k_or >> 32);
                    const unsigned a0 = (unsigned)__builtin_amdgcn_update_dpp((int)alo, (int)alo, CTRL, RM, 0xf, false);
                    const unsigned a1 = (unsigned)__builtin_amdgcn_update_dpp((int)ahi, (int)ahi, CTRL, RM, 0xf, false);
                    const unsigned o0 = (unsigned)__builtin_amdgcn_update_dpp((int)olo, (int)olo, CTRL, RM, 0xf, false);
                    const unsigned o1 = (unsigned)__builtin_amdgcn_update_dpp((int)ohi, (int)ohi, CTRL, RM, 0xf, false);
                    k_and &= (K)(((unsigned long long)a1 << 32) | a0);
                    k_or |= (K)(((unsigned long long)o1 << 32) | o0);
                };
                using std::integral_constant;
                dpp_step(integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});  // row_shr:1
                dpp_step(integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});  // row_shr:2
                dpp_step(integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});  // row_shr:4
                dpp_step(integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});  // row_shr:8
                dpp_step(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});  // row_bcast:15 -> rows 1 and 3
                dpp_step(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});  // row_bcast:31 -> rows 2 and 3
                unsigned long long* part = reinterpret_cast<unsigned long long*>(&hist[1][0]);  // [NWORK][2]
                if (lane == 63) {
                    part[2 * wv] = (unsigned long long)k_and;
                    part[2 * wv + 1] = (unsigned long long)k_or;
                }
            }
            for (int i = tid; i < 256; i += WTHREADS) hist[0][i] = 0;
        }
        __syncthreads();
        if (PREFIX) {
            const unsigned long long* part = reinterpret_cast<const unsigned long long*>(&hist[1][0]);
            unsigned long long a = ~0ull, o = 0ull;
#pragma unroll
            for (int w = 0; w < NWORK; ++w) {
                a &= part[2 * w];
                o |= part[2 * w + 1];
            }
            k_and = (K)a;
            k_or = (K)o;
        }
        const K differ = k_and ^ k_or;
        K prefix = k_and;  // the common high bits (and zeros below them, as far as the search is concerned)
        int kk = kth, parity = 0;
        // hi = the most significant undecided bit; a round decides bits hi .. lo = max(0, hi - 7)
        int hi = differ == 0 ? -1 : (BITS - 1) - (BITS == 64 ? __clzll((unsigned long long)differ) : __clz((unsigned)differ));
        while (hi >= 0) {
            const int lo = hi >= 7 ? hi - 7 : 0;
            int* h = hist[parity];
            if (W) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const K diff = key[k] ^ prefix;
                    const bool same = hi + 1 >= BITS ? true : (diff >> (hi + 1 >= BITS ? 0 : hi + 1)) == 0;
                    if (((mask >> k) & 1u) && same) atomicAdd(&h[(int)((key[k] >> lo) & 255)], 1);
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
            prefix = (prefix & ~((K)255 << lo)) | ((K)bin << lo);  // (the window may reach into decided bits: they are the same)
            kk -= below;
            parity ^= 1;
            if (cnt == 1 && lo > 0) {  // block-uniform: the single key under the prefix
                if (W) {
#pragma unroll
                    for (int k = 0; k < N; ++k)
                        if (((mask >> k) & 1u) && ((key[k] ^ prefix) >> lo) == 0) *xchg = key[k];
                }
                __syncthreads();
                prefix = *xchg;
                break;
            }
            hi = lo - 1;
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

// ---- solver sections: the serial 6x6 routines of pose_math.h.  Executed by ONE lane (the solver lane of pose_kernel.hip) or
// redundantly by every lane of one wave with identical data (pose_kernel2p.hip: same cost as one lane, every lane then holds the
// decisions; LDS stores of identical values from all lanes are benign).  A row-distributed variant (lane i = row i, pivot rows
// through v_readlane) was built for 128-VGPR kernels in round 2 and measured slower (88 k vs 80 k cycles of solver time per
// frame pair); it went with those kernels in round 4. ----

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
__device__ __forceinline__ void solve_normal_eq(PoseSh* sh, double* inc, double* log_abs_det) {
    double H[36], g[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) H[i] = sh->H[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = sh->g[i];
    if (!pm::solve6_spd(H, g, inc, log_abs_det)) pm::solve6(H, g, inc, log_abs_det);
}

// body of gaussNewtonOptimization after optimizeFunctions (:405-428)
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
    solve_normal_eq(sh, inc, nullptr);
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
__device__ __forceinline__ void t0_gnr_iter(PoseSh* sh, double min_error, double min_error_change) {
    t0_unpack(sh);
    const double err = sh->err;
    if (fabs(err - sh->err_prev) < min_error_change || err < min_error) {
        sh->action = ACT_BREAK;
        return;
    }
    double inc[6], DT[16], lad;
    solve_normal_eq(sh, inc, &lad);
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
        solve_normal_eq(sh, inc, nullptr);
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
    solve_normal_eq(sh, inc, nullptr);
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

__device__ __forceinline__ void t0_cov_from_H(PoseSh* sh) {
    double H[36], Hi[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) H[i] = sh->H[i];
    if (!pm::inverse6_spd(H, Hi)) pm::inverse6(H, Hi);  // Matrix6d::inverse(), :429 / :470 / :545
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
__device__ __forceinline__ void t0_is_good_fast(PoseSh* sh, const double* DT, double err) {
    if (err < 0.0 || err > 1.0 || !pm::all_finite16(DT)) {
        sh->good = 0;
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
