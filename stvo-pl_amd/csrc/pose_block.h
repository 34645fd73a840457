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
    double DT[16];   // optimiser variable
    double DT0[16];  // initial DT of optimizePose (:317-326)
    double DT1[16];  // stage-1 result DT_ (:335)
    double DTr[16];  // robust GN's saved entry pose (:441)
    double cov[36];
    double eig[6];
    double H[36];
    double g[6];
    double err, err_prev, err_out, lambda;
    double s_p, s_l;
    unsigned long long xchg[2];  // select2's mailboxes
    int action, good, n_inl_p, n_inl_l, n_m_p, n_m_l, evals, itmp;
};

// Total k of the 28-vector block sum (21 upper-triangular H entries row by row, 6 g, 1 e) goes straight to where the serial routines
// read it — H (both triangles), g, err = e / inliers (:692; 0 / 0 -> NaN) — by the lane that holds it: 28 lanes, two stores each,
// instead of one lane (or sixty-four redundant ones) copying 28 values through a staging array at every iteration.
__device__ __forceinline__ void store_total(PoseSh* sh, int k, double s) {
    if (k < 21) {
        const int i = (k >= 6) + (k >= 11) + (k >= 15) + (k >= 18) + (k >= 20), j = k - (i * (13 - i)) / 2 + i;
        sh->H[i * 6 + j] = s;
        sh->H[j * 6 + i] = s;
    } else if (k < 27) {
        sh->g[k - 21] = s;
    } else {
        sh->err = s / (double)(sh->n_inl_l + sh->n_inl_p);
    }
}

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

    // 28-vector block sum -> sh->H / g / err (store_total; wave partials summed in wave order by the solver wave).
    // Inside a wave: 28 values -> 14 (fold32) -> 7 (fold16) per lane, then a 16-lane row scan of those 7; row r
    // ends up with the wave totals of values 7r .. 7r+6 in its last lane.  147 VALU ops instead of the 504 of 28
    // independent 64-lane scans; fixed association order => bit-reproducible.
    // sc ("the solver contributes", block-uniform; pose_kernel.hip only): the solver wave owns features too — the key-lines of a
    // frame pair with few of them — and leaves its partial results in row NW of `red` / `ired`, which then have NW + 1 rows.
    template <bool W>
    static __device__ __forceinline__ void sum28_fold(double* acc, double (*red)[28], bool sc = false) {
        const int lane = threadIdx.x & 63, wv = W ? (int)(threadIdx.x >> 6) : NW;
        if (W || sc) {
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
    static __device__ __forceinline__ void sum28_finish(double (*red)[28], PoseSh* sh, bool sc = false) {
        const int lane = threadIdx.x & 63;
        __syncthreads();
        if (!W) {
            if (lane < 28) {
                double s = red[0][lane];
#pragma unroll
                for (int w = 1; w < NW; ++w) s += red[w][lane];
                if (sc) s += red[NW][lane];
                store_total(sh, lane, s);
            }
            // the totals are consumed by lane 0 of THIS wave only (the serial routines), so a wave-level fence is enough; the
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
    static __device__ __forceinline__ int sum_int(int v, int* ired, bool sc = false) {
        const int lane = threadIdx.x & 63, wv = W ? (int)(threadIdx.x >> 6) : NW;
        if (W || sc) {
            const int s = wave_sum_i(v);
            if (lane == 0) ired[wv] = s;
        }
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += ired[w];
        if (sc) t += ired[NW];
        __syncthreads();
        return t;
    }

    // four integer block sums at once through the (idle) rows of `red`: ONE barrier; the caller keeps a barrier before `red` is used again
    template <bool W>
    static __device__ __forceinline__ void sum_int4(const int* v, double (*red)[28], int* out, bool sc = false) {
        const int lane = threadIdx.x & 63, wv = W ? (int)(threadIdx.x >> 6) : NW;
        if (W || sc) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = wave_sum_i(v[k]);
                if (lane == 0) reinterpret_cast<int*>(&red[wv][0])[k] = t;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int t = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += reinterpret_cast<const int*>(&red[w][0])[k];
            if (sc) t += reinterpret_cast<const int*>(&red[NW][0])[k];
            out[k] = t;
        }
    }

    // ---- exact k-th element selection on register-resident keys, TWO independent key sets in lockstep -------------------------
    // (the key-points and the key-lines of removeOutliers / of the robust pre-pass: the same barriers serve both.)
    // Radix search from the most significant bit in which the keys of a set DIFFER (block-wide AND / OR; 64-bit keys of threads
    // with many keys only — residual norms share sign and top exponent bits, so a first round from bit 63 decided nothing while
    // ~1500 ds_add hit two or three bins), 7 bits per round: the keys that still carry the prefix are counted into a 128-bin
    // histogram of 16-bit counters (64 words; a set has at most 2048 keys), EVERY wave scans it (one word per lane, DPP prefix
    // sum, the lane that holds rank kk found by ballot) and so knows bin and rank without a second barrier.  Three histograms per
    // set are used in rotation — round r counts into r mod 3 and, after its barrier, clears (r + 2) mod 3, which every wave
    // finished scanning before it arrived at that barrier — so a round costs ONE barrier (round 3: clear / count / barrier / scan
    // by wave 0 / barrier / read back = two).  The rotation carries over from call to call (`rot`, block-uniform): the scratch
    // must be zero before the first call and whenever somebody else has used it.  Integer counts only: the result is the one a
    // sort would give.
    // Scratch S (HIST_W * 2 words): [set][rot][64] histograms, then the AND / OR words of the waves.
    static constexpr int HIST_W = 260;
    static constexpr int SEL_HW = 64, SEL_PART = 6 * SEL_HW;  // words per histogram; word offset of the AND / OR exchange
    static_assert(SEL_PART * 4 % 8 == 0 && SEL_PART + 8 * (NWORK + 1) <= 2 * HIST_W, "selection scratch");

    static __device__ __forceinline__ int wave_incl_scan_i(int v) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1 and 3
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2 and 3
        return v;
    }
    // which bin of histogram h holds the key of rank kk (0 <= kk < keys counted), how many keys lie below that bin, how many in it
    static __device__ __forceinline__ void sel_scan(const unsigned* h, int kk, int& bin, int& below, int& cnt) {
        const int lane = threadIdx.x & 63;
        const unsigned w = h[lane];
        const int c0 = (int)(w & 0xFFFFu), c1 = (int)(w >> 16), sum = c0 + c1;
        const int incl = wave_incl_scan_i(sum), excl = incl - sum;
        const unsigned long long hit = __ballot(excl <= kk && kk < incl);  // exactly one lane
        const int L = hit ? __builtin_ctzll(hit) : 0;
        const bool up = kk >= excl + c0;
        bin = __builtin_amdgcn_readlane(2 * lane + (up ? 1 : 0), L);
        below = __builtin_amdgcn_readlane(excl + (up ? c0 : 0), L);
        cnt = __builtin_amdgcn_readlane(up ? c1 : c0, L);
    }
    // wave AND / OR of the keys in `mask` on DPP moves (lanes without a source keep their own value: the identity of both); lane 63
    // of worker wave wv leaves them in part[0], part[1]
    template <int N, typename K, typename KeyFn>
    static __device__ __forceinline__ void sel_and_or(KeyFn&& key, unsigned mask, unsigned long long* part) {
        K k_and = ~(K)0, k_or = 0;
#pragma unroll
        for (int k = 0; k < N; ++k)
            if ((mask >> k) & 1u) {
                const K kv = key(k);
                k_and &= kv;
                k_or |= kv;
            }
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
        if ((threadIdx.x & 63) == 63) {
            part[0] = (unsigned long long)k_and;
            part[1] = (unsigned long long)k_or;
        }
    }

    // outa / outb = the ktha-th / kthb-th smallest (0-based) of {ka(k) : bit k of ma} / {kb(k) : bit k of mb}; acta / actb
    // (block-uniform): the set is not empty — an inactive set costs nothing and leaves its output alone.  xchg: two LDS words.
    // The keys are FUNCTIONS of the slot (the order-preserving image of a value the caller holds anyway, two or three operations):
    // as arrays they were 39 more live registers per lane through every round, in kernels that spill at 256.
    template <int NA, int NB, bool W, typename K, int BITS, typename KeyA, typename KeyB>
    static __device__ __forceinline__ void select2(KeyA&& ka, unsigned ma, int ktha, bool acta, KeyB&& kb, unsigned mb, int kthb, bool actb,
                                                   unsigned* S, unsigned long long* xchg, int& rot, K& outa, K& outb, bool sc = false) {
        static_assert(BITS == 32 || BITS == 64, "32- or 64-bit keys");
        const int tid = threadIdx.x, wv = tid >> 6;
        const bool own_b = W ? !sc : sc;  // who holds the keys of set B (sc: the solver wave, see sum28_fold)
        // (the AND / OR exchange only where it pays: the float keys of the MAD spread over ~15 top bytes by themselves, and with a
        //  handful of keys per thread it costs more than the degenerate round — the robust mode of the latency kernel, which
        //  selects at every evaluation, lost 8 % to it)
        constexpr bool PFA = BITS == 64 && NA >= 8, PFB = BITS == 64 && NB >= 8;
        K anda = 0, ora = ~(K)0, andb = 0, orb = ~(K)0;  // no exchange: every bit counts as differing
        if (PFA || PFB) {
            unsigned long long* part = reinterpret_cast<unsigned long long*>(S + SEL_PART);  // [NWORK][4]
            if (W) {
                if (PFA) sel_and_or<NA, K>(ka, ma, part + 4 * wv);
                if (PFB && !sc) sel_and_or<NB, K>(kb, mb, part + 4 * wv + 2);
            }
            __syncthreads();
            unsigned long long a0 = ~0ull, o0 = 0ull, a1 = ~0ull, o1 = 0ull;
#pragma unroll
            for (int w = 0; w < NWORK; ++w) {
                if (PFA) { a0 &= part[4 * w]; o0 |= part[4 * w + 1]; }
                if (PFB) { a1 &= part[4 * w + 2]; o1 |= part[4 * w + 3]; }
            }
            if (PFA) { anda = (K)a0; ora = (K)o0; }
            if (PFB && !sc) { andb = (K)a1; orb = (K)o1; }  // (sc: the solver's few keys search from the top bit)
        }
        auto top_bit = [](K differ) -> int {
            return differ == 0 ? -1 : (BITS - 1) - (BITS == 64 ? __clzll((unsigned long long)differ) : __clz((unsigned)differ));
        };
        K pa = anda, pb = andb;  // the common high bits (and zeros below them, as far as the search is concerned)
        int kka = ktha, kkb = kthb;
        // hi = the most significant undecided bit of the set (-1: done); a round decides bits hi .. lo = max(0, hi - 6)
        int hia = acta ? top_bit(anda ^ ora) : -1, hib = actb ? top_bit(andb ^ orb) : -1;
        bool penda = false, pendb = false;  // the result comes from the single key left under the prefix, through xchg
        auto count = [&](auto n_c, auto&& key, unsigned mask, K prefix, int hi, int lo, unsigned* h) {
            constexpr int N = decltype(n_c)::value;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const K kv = key(k);
                const K diff = kv ^ prefix;
                const bool same = hi + 1 >= BITS ? true : (diff >> (hi + 1 >= BITS ? 0 : hi + 1)) == 0;
                const unsigned bin = (unsigned)((kv >> lo) & 127);
                if (((mask >> k) & 1u) && same) atomicAdd(&h[bin >> 1], 1u << (16 * (bin & 1u)));
            }
        };
        auto decide = [&](auto n_c, bool own, auto&& key, unsigned mask, const unsigned* h, K& prefix, int& kk, int& hi, int lo, bool& pend,
                          unsigned long long* mail) {
            constexpr int N = decltype(n_c)::value;
            int bin, below, cnt;
            sel_scan(h, kk, bin, below, cnt);
            prefix = (prefix & ~((K)127 << lo)) | ((K)bin << lo);  // (the window may reach into decided bits: they are the same)
            kk -= below;
            hi = lo - 1;
            if (cnt == 1 && lo > 0) {  // block-uniform: the single key under the prefix; visible after the next barrier
                if (own) {
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const K kv = key(k);
                        if (((mask >> k) & 1u) && ((kv ^ prefix) >> lo) == 0) *mail = (unsigned long long)kv;
                    }
                }
                pend = true;
                hi = -1;
            }
        };
        using std::integral_constant;
        while (hia >= 0 || hib >= 0) {
            const int loa = hia >= 6 ? hia - 6 : 0, lob = hib >= 6 ? hib - 6 : 0;
            unsigned* const ha = S + rot * SEL_HW;
            unsigned* const hb = S + (3 + rot) * SEL_HW;
            if (W && hia >= 0) count(integral_constant<int, NA>{}, ka, ma, pa, hia, loa, ha);
            if (own_b && hib >= 0) count(integral_constant<int, NB>{}, kb, mb, pb, hib, lob, hb);
            __syncthreads();
            if (W && tid < 2 * SEL_HW) {  // the histograms of the round after the next (worker threads 0 .. 127)
                const int rc = rot == 0 ? 2 : rot - 1;
                S[((tid >> 6) * 3 + rc) * SEL_HW + (tid & 63)] = 0u;
            }
            if (hia >= 0) decide(integral_constant<int, NA>{}, W, ka, ma, ha, pa, kka, hia, loa, penda, xchg);
            if (hib >= 0) decide(integral_constant<int, NB>{}, own_b, kb, mb, hb, pb, kkb, hib, lob, pendb, xchg + 1);
            rot = rot == 2 ? 0 : rot + 1;
        }
        __syncthreads();  // the mailboxes; and the AND / OR words are free for the next call
        if (penda) pa = (K)xchg[0];
        if (pendb) pb = (K)xchg[1];
        if (acta) outa = pa;
        if (actb) outb = pb;
    }

    // 1.4826 * MAD of the na values {va[k] : bit k of ma} and of the nb values {vb[k] : bit k of mb}.  Follows vector_stdv_mad /
    // the first half of vector_mean_stdv_mad (src/auxiliar.cpp:395-404, 447-457): median = sorted[n/2]; dev = fabsf(x - median)
    // (FLOAT truncation); MAD = sorted dev[n/2].  The two std::sort calls are replaced by exact k-th-element selection on the
    // order-preserving integer images of the values.  n == 0 -> 0.
    template <int NA, int NB, bool W>
    static __device__ __forceinline__ void mad_sigma2(const double* va, unsigned ma, int na, const double* vb, unsigned mb, int nb, unsigned* S,
                                                      unsigned long long* xchg, int& rot, double& sa, double& sb, bool sc = false) {
        sa = 0.0;
        sb = 0.0;
        const bool acta = na != 0, actb = nb != 0;  // block-uniform
        if (!acta && !actb) return;
        auto image = [](double v) -> unsigned long long {  // total order of IEEE doubles
            const unsigned long long b = (unsigned long long)__double_as_longlong(v);
            return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
        };
        auto value = [](unsigned long long k) -> double {
            return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
        };
        unsigned long long ra = 0ull, rb = 0ull;
        select2<NA, NB, W, unsigned long long, 64>([&](int k) { return image(va[k]); }, ma, na / 2, acta, [&](int k) { return image(vb[k]); }, mb,
                                                   nb / 2, actb, S, xchg, rot, ra, rb, sc);
        const double meda = value(ra), medb = value(rb);
        unsigned qa = 0u, qb = 0u;  // |x - median| as a FLOAT (fabsf: >= 0 or NaN), whose bits order like the values
        select2<NA, NB, W, unsigned, 32>([&](int k) { return __float_as_uint(fabsf((float)(va[k] - meda))); }, ma, na / 2, acta,
                                         [&](int k) { return __float_as_uint(fabsf((float)(vb[k] - medb))); }, mb, nb / 2, actb, S, xchg, rot, qa, qb, sc);
        if (acta) sa = 1.4826 * (double)__uint_as_float(qa);
        if (actb) sb = 1.4826 * (double)__uint_as_float(qb);
    }

    // block sums of N doubles to every thread through red[.][slot0 .. slot0 + N): ONE barrier.  The caller keeps a barrier between
    // two uses of the same slots (and of sum28_fold, which uses all of them).
    template <int N, bool W>
    static __device__ __forceinline__ void sum_at(const double* v, double (*red)[28], int slot0, double* out, bool sc = false) {
        const int lane = threadIdx.x & 63, wv = W ? (int)(threadIdx.x >> 6) : NW;
        if (W || sc) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double s = wave_sum_lane63(v[k]);
                if (lane == 63) red[wv][slot0 + k] = s;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double s = red[0][slot0 + k];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w][slot0 + k];
            if (sc) s += red[NW][slot0 + k];
            out[k] = s;
        }
    }

    // removeOutliers (:988-1067) for both kinds of features at once, given the weighted residual norms of ALL matches (ra / rb by
    // slot, matched masks ma / mb, na / nb matches in the block): robust scale (mad_sigma2), the mean of the samples below two
    // sigma or of all samples (src/auxiliar.cpp:405-427), then every inlier further than inlier_k sigma from the mean goes.
    // do_a / do_b = has_points / has_lines (:991, :1026).  inla / inlb are updated; cnt[0], cnt[1] = the new inlier counts of the
    // block (meaningful for the kinds that were processed).  Ends WITHOUT a barrier: the caller has one before red is used again.
    template <int NA, int NB, bool W>
    static __device__ __forceinline__ void outlier_cut(const double* ra, unsigned ma, int na, bool do_a, const double* rb, unsigned mb, int nb,
                                                       bool do_b, double inlier_k, unsigned& inla, unsigned& inlb, unsigned* S,
                                                       unsigned long long* xchg, int& rot, double (*red)[28], int* cnt, bool sc = false) {
        const bool own_b = W ? !sc : sc;
        double sa, sb;
        mad_sigma2<NA, NB, W>(ra, do_a ? ma : 0u, do_a ? na : 0, rb, do_b ? mb : 0u, do_b ? nb : 0, S, xchg, rot, sa, sb, sc);
        double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (W) {
#pragma unroll
            for (int k = 0; k < NA; ++k)
                if (do_a && ((ma >> k) & 1u)) {
                    if (ra[k] < 2.0 * sa) {
                        v[0] += ra[k];
                        v[1] += 1.0;
                    }
                    v[2] += ra[k];
                }
        }
        if (own_b) {
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (do_b && ((mb >> k) & 1u)) {
                    if (rb[k] < 2.0 * sb) {
                        v[3] += rb[k];
                        v[4] += 1.0;
                    }
                    v[5] += rb[k];
                }
        }
        double t[6];
        sum_at<6, W>(v, red, 0, t, sc);
        auto mean_of = [](int tot, const double* t3) -> double {
            if (tot == 0) return 0.0;
            const int ksel = (int)t3[1];
            return (ksel >= (int)(0.2 * (double)tot)) ? t3[0] / (double)ksel : t3[2] / (double)tot;
        };
        const double mean_a = mean_of(na, t), mean_b = mean_of(nb, t + 3);
        const double tha = inlier_k * sa, thb = inlier_k * sb;
        double c[2] = {0.0, 0.0};
        if (W) {
            if (do_a) {
#pragma unroll
                for (int k = 0; k < NA; ++k)
                    if (((inla >> k) & 1u) && fabs(ra[k] - mean_a) > tha) inla &= ~(1u << k);
            }
            c[0] = (double)__popc(inla);
        }
        if (own_b) {
            if (do_b) {
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    if (((inlb >> k) & 1u) && fabs(rb[k] - mean_b) > thb) inlb &= ~(1u << k);
            }
            c[1] = (double)__popc(inlb);
        }
        double ct[2];
        sum_at<2, W>(c, red, 8, ct, sc);
        cnt[0] = (int)ct[0];
        cnt[1] = (int)ct[1];
    }
};

__device__ __forceinline__ double norm3(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// ---- solver sections: the serial 6x6 routines of pose_math.h.  Executed by ONE lane (the solver lane of pose_kernel.hip) or
// redundantly by every lane of one wave with identical data (pose_kernel2p.hip: same cost as one lane, every lane then holds the
// decisions; LDS stores of identical values from all lanes are benign).  A row-distributed variant (lane i = row i, pivot rows
// through v_readlane) was built for 128-VGPR kernels in round 2 and measured slower (88 k vs 80 k cycles of solver time per
// frame pair); it went with those kernels in round 4. ----

// H inc = g (ColPivHouseholderQR::solve at :417-418 and siblings): LDL^T when H is certified positive definite
// (the normal case), the pivoted QR otherwise — see pose_math.h.
// ws: >= 51 doubles of LDS nobody else touches during the serial section (the callers pass the partial-sum rows: the other waves are
// parked at the barrier behind it).  The pivoted fallback runs there, on memory operands with run-time loops (pm::solve6_mem: the same
// operations in the same order) — inlined with its matrices in registers it was, together with inverse6 below, what the batch kernel
// spilled 86 registers for (round 6: with both on memory operands pose2c_kernel<2> has no scratch).
__device__ __forceinline__ void solve_normal_eq(PoseSh* sh, double* inc, double* log_abs_det, double* ws) {
    double H[36], g[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) H[i] = sh->H[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = sh->g[i];
    if (!pm::solve6_spd(H, g, inc, log_abs_det)) {
        double* A = ws;
        double* c = ws + 36;  // right-hand side, then the solution
        double* y = ws + 42;
        int* perm = reinterpret_cast<int*>(ws + 48);
#pragma unroll 1
        for (int i = 0; i < 36; ++i) A[i] = sh->H[i];
#pragma unroll 1
        for (int i = 0; i < 6; ++i) c[i] = sh->g[i];
        pm::solve6_mem(A, c, y, perm, c, log_abs_det);
#pragma unroll
        for (int i = 0; i < 6; ++i) inc[i] = c[i];
    }
}

// body of gaussNewtonOptimization after optimizeFunctions (:405-428)
__device__ __forceinline__ void t0_gn_iter(PoseSh* sh, double min_error, double min_error_change, int it, double* ws) {
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
    solve_normal_eq(sh, inc, nullptr, ws);
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
__device__ __forceinline__ void t0_gnr_iter(PoseSh* sh, double min_error, double min_error_change, double* ws) {
    const double err = sh->err;
    if (fabs(err - sh->err_prev) < min_error_change || err < min_error) {
        sh->action = ACT_BREAK;
        return;
    }
    double inc[6], DT[16], lad;
    solve_normal_eq(sh, inc, &lad, ws);
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
__device__ __forceinline__ void t0_lm_iter(PoseSh* sh, double min_error, double min_error_change, int first, double* ws) {
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
        solve_normal_eq(sh, inc, nullptr, ws);
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
    solve_normal_eq(sh, inc, nullptr, ws);
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

__device__ __forceinline__ void t0_cov_from_H(PoseSh* sh, double* ws) {
    double H[36], Hi[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) H[i] = sh->H[i];
    if (pm::inverse6_spd(H, Hi)) {  // Matrix6d::inverse(), :429 / :470 / :545
#pragma unroll
        for (int i = 0; i < 36; ++i) sh->cov[i] = Hi[i];
    } else {  // the pivoted LU on memory operands (see solve_normal_eq): a copy of H in ws, the inverse straight into sh->cov
#pragma unroll 1
        for (int i = 0; i < 36; ++i) ws[i] = sh->H[i];
        pm::inverse6_mem(ws, sh->cov);
    }
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

// lazy_eig (single-stream operation, results read by stvo_seq_read): the verdict of :372 comes from the certificate (exact when it
// says "good"; anything it cannot certify takes the eigen-decomposition here), and DT_cov_eig — an output, not an input of anything on
// the path — is left to the host, which runs the same routine in ~2 us: the decomposition is ~3.5 k dependent instructions, 8 us on
// one lane at the very end of the frame's chain.
// next_T (the device-resident pipeline under use_motion_model, or nullptr): the NEXT frame pair's initial DT by the rule of :317-324 —
// prev_frame->DT unless !isGoodSolution(prev DT, prev DT_cov, prev err_norm).  For a committed solution DT_cov and err_norm just
// passed that very test, so only is_finite(committed DT) is left to check; a rejected one has DT = I either way.
__device__ __forceinline__ void t0_commit(PoseSh* sh, stvo_pose_result* out, int status, int path, int it0, int it1, bool lazy_eig = false,
                                          double* next_T = nullptr) {
    // :372-391
    bool eig_pending = false;
    if (lazy_eig) {
        double C[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) C[i] = sh->cov[i];
        if (sh->err_out < 0.0 || sh->err_out > 1.0 || !pm::all_finite16(sh->DT)) {
            sh->good = 0;  // (the eigenvalues do not matter: the outputs of a rejected solution are zeros)
        } else if (pm::spd_unit_certificate(C) == 1) {
            sh->good = 1;
            eig_pending = true;
        } else {
            t0_is_good(sh, sh->DT, sh->err_out);
        }
    } else {
        t0_is_good(sh, sh->DT, sh->err_out);
    }
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
        if (next_T) {
            const bool fin = pm::all_finite16(T);
#pragma unroll
            for (int i = 0; i < 16; ++i) next_T[i] = fin ? T[i] : ((i % 5 == 0) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int i = 0; i < 36; ++i) out->cov[i] = sh->cov[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) out->cov_eig[i] = eig_pending ? 0.0 : sh->eig[i];
        if (eig_pending) path |= PATH_EIG_PENDING;
        out->err = sh->err_out;
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) out->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
        if (next_T) {
#pragma unroll
            for (int i = 0; i < 16; ++i) next_T[i] = (i % 5 == 0) ? 1.0 : 0.0;
        }
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
