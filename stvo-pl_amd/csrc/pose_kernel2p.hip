// pose_kernel2p.hip — K4/K5/K6, the batch formulation with THREAD-PRIVATE record storage: the whole of
// StereoFrameHandler::optimizePose (/root/reference/src/stereoFrameHandler.cpp:307-392) in one launch, one workgroup per
// frame pair, built so that FOUR frame pairs share a CU (two waves per pair at 256 VGPRs, a 40 KB LDS share).
//
// A pair is a chain of ~13.7 evaluate -> solve rounds with serial sections in which its other waves idle (NOTES.md), so what
// the CU's FP64 pipes do depends on how many independent chains it holds.  A thread owns prev points t + k BLOCK and numbers
// its matched records 0, 1, ... in that order (the ORDINAL); record r of thread t lives in LDS planes [r][part][t] — 16-byte
// parts of neighbouring threads are neighbours: conflict-free wide accesses, no cross-thread compaction, no barrier between
// staging and use.  Two kernels share the state machine below:
//   * pose2c_kernel (round 4) — COMPACT records, the device-resident pipeline's format (kernels.h: PoseArgs::prev_rc): a stereo
//     point is {u, v, disparity, level}; the LDS record is {u, v, ox, oy} as floats + b / disparity as a double = 24 bytes, P is
//     rebuilt with 2 subtractions + 3 products per use and sqrt(sigma2) comes from a 16-entry table.  TWELVE ordinals fit the
//     40 KB share (one more stays in registers): every record of the bench shape (~11.6 per thread) is on chip, nothing is
//     staged through HBM, and a record costs one 16-byte load per side (52 -> 36 bytes of gather per prev point);
//   * pose2p_kernel (round 3) — the general form for callers that hand over P / sigma2 / observations as doubles
//     (stvo_track_batched_dev): 48-byte records, the first K_lds ordinals in LDS, the others in a global arena with the same
//     plane layout, read with a three-deep register ring.
// Key-lines (~75 per pair, at most one per thread in practice) live in the arena in both.  Same arithmetic and association
// order in both (a thread accumulates its records in ascending ordinal), so the results agree bit for bit.
#include <cstdlib>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "pose_block.h"

namespace stvo {
namespace {

__device__ __forceinline__ double uni(double v) {  // a value every lane holds identically -> SGPR pair
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xFFFFFFFFll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

struct PointRec2 {
    double X, Y, Z, ox, oy, q;  // q = sqrt(sigma2)
};

constexpr bool POSE2P_PRIO = true;  // serial sections at wave priority 3 (measured: 239 -> 230 us per 512 pairs with four waves per pair)

// ---------------- the optimizePose state machine (:332-370), shared by both kernels ----------------
// evaluate(robust): optimizeFunctions[Robust] at sh->DT -> sh->H / g / err (store_total); remove_outliers(): :988-1067 at
// sh->DT1.  Every decision is block-uniform (read from LDS after a barrier); wave 0 runs the serial sections.
struct PoseFlow {
    int status, path, it0, it1;
};
// prm(): the optimizer parameters, fetched where they are used (pose2c_kernel reads them from the kernel-argument segment)
template <typename Prm, typename Eval, typename Rem, typename Tick>
__device__ __forceinline__ PoseFlow optimize_pose_flow(PoseSh* sh, Prm&& prm, const bool w0, Eval&& evaluate,
                                                       Rem&& remove_outliers, Tick&& tick, long long* tprof, double* ws) {
    int status = STVO_POSE_OK, path = 0, it0 = 0, it1 = 0;
    if (sh->n_inl_p + sh->n_inl_l >= prm().min_features) {
        int stage = 0;        // 0 = first optimisation (:335-338), 1 = refinement (:345-350), 2 = robust fallback (:359)
        int alg = prm().mode;   // 0 GN, 1 robust GN, 2 LM
        int max_it = prm().max_iters;
        for (;;) {
            if (w0) {
                sh->err_prev = 999999999.9;
                sh->good = 1;
#pragma unroll
                for (int i = 0; i < 16; ++i) sh->DTr[i] = sh->DT[i];  // robust GN's entry pose (:441)
            }
            const int n_it = (alg == 2 && max_it < 1) ? 1 : max_it;  // LM always evaluates once (:493)
            int evals = 0, action = ACT_BREAK;
            for (int it = 0; it < n_it; ++it) {
                long long tq = tick();
                evaluate(alg == 1);
                tprof[0] += tick() - tq;
                tq = tick();
                ++evals;
                if (w0) {
                    // the serial section is one wave's dependent chain while the co-resident workgroup's waves evaluate on the
                    // same SIMD: let it win the issue arbitration
                    if (POSE2P_PRIO) __builtin_amdgcn_s_setprio(3);
                    if (alg == 0) t0_gn_iter(sh, prm().min_error, prm().min_error_change, it, ws);
                    else if (alg == 1) t0_gnr_iter(sh, prm().min_error, prm().min_error_change, ws);
                    else t0_lm_iter(sh, prm().min_error, prm().min_error_change, it == 0 ? 1 : 0, ws);
                    if (POSE2P_PRIO) __builtin_amdgcn_s_setprio(0);
                }
                __syncthreads();
                tprof[1] += tick() - tq;
                action = sh->action;
                if (action != ACT_CONTINUE) break;
            }
            long long tq2 = tick();
            if (w0) {
                if (alg == 0 && action == ACT_FAIL) {
                    sh->err_out = -1.0;  // :408-409, covariance left untouched
                } else if (alg == 1 && !sh->good) {  // :473-478
#pragma unroll
                    for (int i = 0; i < 16; ++i) sh->DT[i] = sh->DTr[i];
                    sh->err_out = -1.0;
#pragma unroll
                    for (int i = 0; i < 36; ++i) sh->cov[i] = (i % 7 == 0) ? 1.0 : 0.0;
                } else {
                    t0_cov_from_H(sh, ws);  // :429 / :470 / :545 — H of the last evaluation (damped for LM)
                    sh->err_out = evals > 0 ? sh->err : 0.0;
                }
            }
            __syncthreads();
            tprof[2] += tick() - tq2;
            if (stage != 0) {
                it1 = evals;
                break;
            }
            it0 = evals;
            tq2 = tick();
            if (w0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sh->DT1[i] = sh->DT[i];
                t0_is_good_fast(sh, sh->DT1, sh->err_out);
            }
            __syncthreads();
            tprof[2] += tick() - tq2;
            if (sh->good) {  // :341
                path |= STVO_PATH_STAGE1_GOOD;
                tq2 = tick();
                remove_outliers();
                tprof[3] += tick() - tq2;
                if (sh->n_inl_p + sh->n_inl_l >= prm().min_features) {  // :345 — restart from the INITIAL DT
                    path |= STVO_PATH_REFINED;
                    stage = 1;
                } else {
                    if (w0) pm::identity4(sh->DT);
                    status = STVO_POSE_FEW_INLIERS_AFTER;
                    __syncthreads();
                    break;
                }
            } else {  // :357-362 robust GN on everything, from the initial DT
                path |= STVO_PATH_ROBUST_FALLBACK;
                stage = 2;
                alg = 1;
            }
            max_it = prm().max_iters_ref;
            if (w0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sh->DT[i] = sh->DT0[i];
            }
            __syncthreads();
        }
    } else {
        if (w0) pm::identity4(sh->DT);
        status = STVO_POSE_FEW_INLIERS_BEFORE;
        __syncthreads();
    }

    return PoseFlow{status, path, it0, it1};
}

// NW waves per frame pair at 256 VGPRs (two waves per SIMD); k_lds record ordinals of every thread live in LDS
// PROF: the developer's phase-tick instrumentation (tools/pose_probe.py) as its own instantiation — as a run-time flag its ~26
// live counters cost the production kernel registers across the whole state machine (30 VGPRs spilled at the loop head)
template <int NW, bool PROF>
__global__ __launch_bounds__(NW * 64, 2) void pose2p_kernel(PoseArgs a, const int k_lds, double2* arena, const size_t arena_pair) {
    constexpr int BLOCK = NW * 64;
    constexpr int PPT = (STVO_POSE_MAX_POINTS + BLOCK - 1) / BLOCK;
    constexpr int LPT = (STVO_POSE_MAX_LINES + BLOCK - 1) / BLOCK;
    using Ops = BlockOps<NW>;
    extern __shared__ double2 s_pl[];  // [k_lds][3][BLOCK]: 16-byte part `part` of this thread's record ordinal r at (r * 3 + part) * BLOCK + tid
    __shared__ __align__(16) int s_hist[2][Ops::HIST_W];  // BlockOps::select2
    __shared__ double s_red[NW][28];
    __shared__ int s_ired[NW];
    __shared__ PoseSh s_sh;
    PoseSh* sh = &s_sh;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool w0 = wv == 0;   // the wave that also runs the serial sections (all of its lanes, redundantly)
    const bool t0 = tid == 0;  // the lane that writes results to global memory
    constexpr bool prof = PROF;
    long long tprof[5] = {0, 0, 0, 0, 0};
    long long wprof[3] = {0, 0, 0}, wave_busy = 0;  // developer aid: loop compute, fold, barrier + partial sums; per-wave busy ticks
    auto tick = [&]() -> long long {
        if constexpr (PROF) return (long long)__builtin_readcyclecounter();
        else return 0ll;
    };
    const long long t_begin = tick();
    const stvo_cam cam_f = a.cams ? a.cams[f] : a.cam;
    const pm::Cam5 cam{cam_f.fx, cam_f.fy, cam_f.cx, cam_f.cy};
    const stvo_opt_params prm = a.prm;
    const double inv_homog = 1.0 / prm.homog_th;

    // ---------------- ownership: thread t owns prev points t + k BLOCK and prev line BLOCK - 1 - t ----------------
    // (lines are handed out from the top: the last threads own the fewest points, a line term costs about two points)
    unsigned pmatched = 0u, pinl = 0u;
    const int n_prev_p = a.n_prev_pts != nullptr ? min(a.n_prev_pts[f], a.max_pts) : 0;
    const size_t pbase = (size_t)f * a.max_pts;
    // branch-free, clamped loads: the PPT match indices (and below the PPT records) of a thread are all in flight at once —
    // the prologue is a chain of dependent HBM round trips otherwise (index -> record -> LDS)
    int jj[PPT];
    {
        int init[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * BLOCK;
            const int ic = i < n_prev_p ? i : 0;
            jj[k] = a.m12p ? a.m12p[pbase + ic] : ic;
            init[k] = a.init_inl_p ? a.init_inl_p[pbase + ic] : 1;
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * BLOCK;
            if (i < n_prev_p && jj[k] >= 0) {
                pmatched |= 1u << k;
                if (init[k] != 0) pinl |= 1u << k;
            } else {
                jj[k] = 0;
            }
        }
    }
    unsigned lmatched = 0u, linl = 0u;
    const int n_prev_l = (a.n_prev_lines != nullptr && a.max_lines > 0) ? a.n_prev_lines[f] : 0;
    const size_t lbase = (size_t)f * a.max_lines;
    const int li0 = BLOCK - 1 - tid;  // line k of this thread: li0 + k BLOCK
    int jl[LPT];
    {   // branch-free, clamped loads as for the points: all LPT match indices of the thread in one round trip
        int initl[LPT];
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int li = li0 + k * BLOCK;
            const int lc = (li < n_prev_l && li < a.max_lines) ? li : 0;
            jl[k] = (a.m12l && a.max_lines > 0) ? a.m12l[lbase + lc] : lc;
            initl[k] = (a.init_inl_l && a.max_lines > 0) ? a.init_inl_l[lbase + lc] : 1;
        }
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int li = li0 + k * BLOCK;
            if (li < n_prev_l && li < a.max_lines && jl[k] >= 0) {
                lmatched |= 1u << k;
                if (initl[k] != 0) linl |= 1u << k;
            } else {
                jl[k] = 0;
            }
        }
    }

    // ---------------- thread-private records: LDS planes for the first k_lds ordinals, the global arena for the rest ----------------
    unsigned* const sel = reinterpret_cast<unsigned*>(&s_hist[0][0]);  // BlockOps::select2's scratch: zero before its first use
    int sel_rot = 0;
    for (int i = tid; i < 2 * Ops::HIST_W; i += BLOCK) sel[i] = 0u;  // (the barriers of the counts below come first)
    const int n_m_p = Ops::template sum_int<true>(__popc(pmatched), s_ired);
    const int n_m_l = Ops::template sum_int<true>(__popc(lmatched), s_ired);
    // (the index of a record is laundered through an empty asm at every use: otherwise the compiler hoists the 64-bit addresses
    //  of all PPT records out of the iteration loop and carries them through the whole kernel — see pose_kernel2.hip)
    auto launder = [](int v) -> int {
        asm volatile("" : "+v"(v));
        return v;
    };
    double2* ar = arena + (size_t)f * arena_pair;              // points: ordinal r, part p at (r * 3 + p) * BLOCK + tid
    double2* ar_l = ar + (size_t)PPT * 3 * BLOCK;               // lines:  ordinal r, part p at (r * 7 + p) * BLOCK + tid
    unsigned arena_mask = pmatched;                             // the owned matched points whose ordinal is >= k_lds
    for (int i = 0; i < k_lds; ++i) arena_mask &= arena_mask - 1u;
    const unsigned lds_mask = pmatched & ~arena_mask;
    auto ordinal = [&](int k) -> int { return __popc(pmatched & ((1u << k) - 1u)); };
    auto load_point_lds = [&](int k) -> PointRec2 {
        const int r = launder(ordinal(k));
        const double2* q = s_pl + (size_t)(r * 3) * BLOCK + tid;
        const double2 v0 = q[0], v1 = q[BLOCK], v2 = q[2 * BLOCK];
        PointRec2 rec;
        rec.X = v0.x; rec.Y = v0.y; rec.Z = v1.x; rec.ox = v1.y; rec.oy = v2.x; rec.q = v2.y;
        return rec;
    };
    auto load_point_arena = [&](int k) -> PointRec2 {
        const int r = launder(ordinal(k));
        const double2* q = ar + (size_t)(r * 3) * BLOCK + tid;
        const double2 v0 = q[0], v1 = q[BLOCK], v2 = q[2 * BLOCK];
        PointRec2 rec;
        rec.X = v0.x; rec.Y = v0.y; rec.Z = v1.x; rec.ox = v1.y; rec.oy = v2.x; rec.q = v2.y;
        return rec;
    };
    auto load_point = [&](int k) -> PointRec2 { return ((lds_mask >> k) & 1u) ? load_point_lds(k) : load_point_arena(k); };
    auto load_line = [&](int k) -> pm::LineRec {
        const int r = launder(__popc(lmatched & ((1u << k) - 1u)));
        const double2* q = ar_l + (size_t)(r * 7) * BLOCK + tid;
        pm::LineRec L;
        const double2 v0 = q[0], v1 = q[BLOCK], v2 = q[2 * BLOCK], v3 = q[3 * BLOCK], v4 = q[4 * BLOCK], v5 = q[5 * BLOCK], v6 = q[6 * BLOCK];
        L.sP[0] = v0.x; L.sP[1] = v0.y; L.sP[2] = v1.x; L.eP[0] = v1.y; L.eP[1] = v2.x; L.eP[2] = v2.y;
        L.le[0] = v3.x; L.le[1] = v3.y; L.le[2] = v4.x; L.spl[0] = v4.y; L.spl[1] = v5.x; L.epl[0] = v5.y; L.epl[1] = v6.x;
        L.sigma2 = v6.y;
        return L;
    };
    // stage this thread's own records (thread-private slots: no barrier between staging and use), STAGE_CH records at a time
    constexpr int STAGE_CH = PPT < 8 ? PPT : 8;  // 96 VGPRs of loaded values in flight: half as many round trips as with 4
#pragma unroll
    for (int k0 = 0; k0 < PPT; k0 += STAGE_CH) {
        PointRec2 rec[STAGE_CH];
#pragma unroll
        for (int c = 0; c < STAGE_CH; ++c) {  // unconditional loads from clamped (valid) addresses
            const int k = k0 + c;
            if (k >= PPT) continue;
            const int ik = tid + k * BLOCK;
            const size_t i = pbase + (size_t)(ik < n_prev_p ? ik : 0), j = pbase + (size_t)jj[k];
            rec[c].X = a.prev_P[i * 3 + 0];
            rec[c].Y = a.prev_P[i * 3 + 1];
            rec[c].Z = a.prev_P[i * 3 + 2];
            rec[c].q = a.prev_s2p[i];
            rec[c].ox = a.curr_pl[j * 2 + 0];
            rec[c].oy = a.curr_pl[j * 2 + 1];
        }
#pragma unroll
        for (int c = 0; c < STAGE_CH; ++c) {
            const int k = k0 + c;
            if (k >= PPT) continue;
            if ((pmatched >> k) & 1u) {
                const int r = ordinal(k);
                double2* q = (((lds_mask >> k) & 1u) ? s_pl : ar) + (size_t)(r * 3) * BLOCK + tid;
                q[0] = make_double2(rec[c].X, rec[c].Y);
                q[BLOCK] = make_double2(rec[c].Z, rec[c].ox);
                q[2 * BLOCK] = make_double2(rec[c].oy, sqrt(rec[c].q));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < LPT; ++k)
        if ((lmatched >> k) & 1u) {
            const size_t i = lbase + (size_t)(li0 + k * BLOCK);
            const size_t j = lbase + (size_t)jl[k];
            const int r = __popc(lmatched & ((1u << k) - 1u));
            double2* q = ar_l + (size_t)(r * 7) * BLOCK + tid;
            q[0] = make_double2(a.prev_sP[i * 3 + 0], a.prev_sP[i * 3 + 1]);
            q[BLOCK] = make_double2(a.prev_sP[i * 3 + 2], a.prev_eP[i * 3 + 0]);
            q[2 * BLOCK] = make_double2(a.prev_eP[i * 3 + 1], a.prev_eP[i * 3 + 2]);
            q[3 * BLOCK] = make_double2(a.curr_le[j * 3 + 0], a.curr_le[j * 3 + 1]);
            q[4 * BLOCK] = make_double2(a.curr_le[j * 3 + 2], a.prev_spl[i * 2 + 0]);
            q[5 * BLOCK] = make_double2(a.prev_spl[i * 2 + 1], a.prev_epl[i * 2 + 0]);
            q[6 * BLOCK] = make_double2(a.prev_epl[i * 2 + 1], sqrt(a.prev_s2l[i]));  // the record carries sqrt(sigma2) (pm::line_term_q)
        }
    // (a thread reads back only what it wrote itself: program order is enough, no fence)

    {
        const int nip = Ops::template sum_int<true>(__popc(pinl), s_ired);
        const int nil = Ops::template sum_int<true>(__popc(linl), s_ired);
        if (w0) {
            sh->n_m_p = n_m_p;
            sh->n_m_l = n_m_l;
            sh->n_inl_p = nip;
            sh->n_inl_l = nil;
            sh->good = 1;
            sh->err_out = -1.0;  // :313
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double v = a.init_T ? a.init_T[(size_t)f * 16 + i] : ((i % 5 == 0) ? 1.0 : 0.0);
                sh->DT[i] = v;
                sh->DT0[i] = v;
            }
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                sh->cov[i] = 0.0;
                sh->H[i] = 0.0;
            }
        }
        __syncthreads();
    }

    const long long t_prologue = tick() - t_begin;
    // ---------------- optimizeFunctions / optimizeFunctionsRobust at sh->DT ----------------
    auto pose_sgpr = [&](const double* src, double* DT) {  // rows 0..2 of the 4x4 pose, wave-uniform -> SGPRs
#pragma unroll
        for (int i = 0; i < 12; ++i) DT[i] = uni(src[i]);
    };
    auto evaluate = [&](bool robust) {
        double DT[12];
        pose_sgpr(sh->DT, DT);
        double sp = 1.0, sl = 1.0;
        if (robust) {  // pre-pass :710-781: MAD scale of the inlier residual norms
            double rp[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                rp[k] = 0.0;
                if ((pinl >> k) & 1u) {
                    const PointRec2 r = load_point(k);
                    rp[k] = pm::point_residual(DT, cam, r.X, r.Y, r.Z, r.ox, r.oy);
                }
            }
            double rlv[LPT];
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                rlv[k] = 0.0;
                if ((linl >> k) & 1u) rlv[k] = pm::line_residual(DT, cam, load_line(k));
            }
            Ops::template mad_sigma2<PPT, LPT, true>(rp, pinl, sh->n_inl_p, rlv, linl, sh->n_inl_l, sel, sh->xchg, sel_rot, sp, sl);
            sp = pm::clamp_scale(sp);
            sl = pm::clamp_scale(sl);
        }
        const double isp = 1.0 / sp, isl = 1.0 / sl;  // reciprocals of the robust scales: one division per evaluation, not per feature
        const long long tw0 = tick();
        double acc[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) acc[i] = 0.0;
        {
            // the arena-resident inliers (the thread's highest ordinals): three records in flight, the first three requested now
            // and consumed after the LDS-resident ones — same ascending order as pose_kernel2.hip's accumulation
            unsigned pend = pinl & arena_mask;
            PointRec2 r0{1.0, 1.0, 1.0, 0.0, 0.0, 1.0}, r1 = r0, r2 = r0;
            auto req = [&](PointRec2& dst) -> bool {
                if (!pend) return false;
                const int k = __builtin_ctz(pend);
                pend &= pend - 1u;
                dst = load_point_arena(k);
                return true;
            };
            bool v0 = req(r0), v1 = req(r1), v2 = req(r2);
            // the first inlier line of the thread travels with them
            pm::LineRec L0{};
            int kl0 = -1;
            if (linl) {
                kl0 = __builtin_ctz(linl);
                L0 = load_line(kl0);
            }
            // LDS-resident inliers, one record ahead
            unsigned todo = pinl & lds_mask;
            PointRec2 cur{1.0, 1.0, 1.0, 0.0, 0.0, 1.0};
            if (todo) cur = load_point_lds(__builtin_ctz(todo));
            while (todo) {
                todo &= todo - 1u;
                PointRec2 nxt = cur;
                if (todo) nxt = load_point_lds(__builtin_ctz(todo));
                pm::point_term_q(acc, DT, cam, prm.homog_th, inv_homog, cur.X, cur.Y, cur.Z, cur.ox, cur.oy, cur.q, robust, isp);
                cur = nxt;
            }
            for (;;) {
                if (!v0) break;
                pm::point_term_q(acc, DT, cam, prm.homog_th, inv_homog, r0.X, r0.Y, r0.Z, r0.ox, r0.oy, r0.q, robust, isp);
                v0 = req(r0);
                if (!v1) break;
                pm::point_term_q(acc, DT, cam, prm.homog_th, inv_homog, r1.X, r1.Y, r1.Z, r1.ox, r1.oy, r1.q, robust, isp);
                v1 = req(r1);
                if (!v2) break;
                pm::point_term_q(acc, DT, cam, prm.homog_th, inv_homog, r2.X, r2.Y, r2.Z, r2.ox, r2.oy, r2.q, robust, isp);
                v2 = req(r2);
            }
            if (kl0 >= 0) pm::line_term_q(acc, DT, cam, prm.homog_th, inv_homog, L0, robust, isl);
#pragma unroll 1
            for (int k = 0; k < LPT; ++k)
                if (((linl >> k) & 1u) && k != kl0) {
                    const pm::LineRec L = load_line(k);
                    pm::line_term_q(acc, DT, cam, prm.homog_th, inv_homog, L, robust, isl);
                }
        }
        const long long tw1 = tick();
        Ops::template sum28_fold<true>(acc, s_red);
        const long long tw2 = tick();
        wave_busy += tw2 - tw0;
        __syncthreads();
        if (w0) {  // wave partials summed in wave order => bit-reproducible
            if (lane < 28) {
                double s = s_red[0][lane];
#pragma unroll
                for (int w = 1; w < NW; ++w) s += s_red[w][lane];
                store_total(sh, lane, s);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        wprof[0] += tw1 - tw0;
        wprof[1] += tw2 - tw1;
        wprof[2] += tick() - tw2;
    };

    if (a.eval_only) {
        evaluate(a.eval_robust != 0);
        if (w0) {
            if (t0) {
                double* o = a.eval_out + (size_t)f * 44;
#pragma unroll
                for (int i = 0; i < 36; ++i) o[i] = sh->H[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) o[36 + i] = sh->g[i];
                o[42] = sh->err;
                o[43] = (double)(sh->n_inl_p + sh->n_inl_l);
            }
        }
        return;
    }

    // ---------------- removeOutliers at pose DT1 (:988-1067) ----------------
    auto remove_outliers = [&]() {
        double DT[12];
        pose_sgpr(sh->DT1, DT);
        double resp[PPT], resl[LPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {  // ALL matches, current outliers included (:998-1005)
            resp[k] = 0.0;
            if (prm.has_points && ((pmatched >> k) & 1u)) {
                const PointRec2 r = load_point(k);
                resp[k] = pm::point_residual(DT, cam, r.X, r.Y, r.Z, r.ox, r.oy) * r.q;
            }
        }
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            resl[k] = 0.0;
            if (prm.has_lines && ((lmatched >> k) & 1u)) {
                const pm::LineRec L = load_line(k);
                resl[k] = pm::line_residual(DT, cam, L) * L.sigma2;  // L.sigma2 = sqrt(sigma2)
            }
        }
        int cnt[2];
        Ops::template outlier_cut<PPT, LPT, true>(resp, pmatched, sh->n_m_p, prm.has_points != 0, resl, lmatched, sh->n_m_l, prm.has_lines != 0,
                                                  prm.inlier_k, pinl, linl, sel, sh->xchg, sel_rot, s_red, cnt);
        if (w0) {
            if (prm.has_points) sh->n_inl_p = cnt[0];
            if (prm.has_lines) sh->n_inl_l = cnt[1];
        }
        __syncthreads();
    };

    const PoseFlow fl = optimize_pose_flow(sh, [&]() -> const stvo_opt_params& { return prm; }, w0, evaluate, remove_outliers, tick, tprof, &s_red[0][0]);
    const int status = fl.status, path = fl.path, it0 = fl.it0, it1 = fl.it1;

    {
        const long long tq3 = tick();
        if (t0) t0_commit(sh, a.results + f, status, path, it0, it1, false, a.next_T ? a.next_T + (size_t)f * 16 : nullptr);
        tprof[2] += tick() - tq3;
    }
    if (prof && t0) {
        tprof[4] = tick() - t_begin;
#pragma unroll
        for (int i = 0; i < 5; ++i) a.prof_out[(size_t)f * 16 + i] = tprof[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) a.prof_out[(size_t)f * 16 + 5 + i] = wprof[i];
    }
    if (prof && lane == 0 && (wv < 6 || wv == NW - 1)) a.prof_out[(size_t)f * 16 + 8 + (wv < 6 ? wv : 7)] = wave_busy;  // waves 0..5 and the last
    if (prof && t0) a.prof_out[(size_t)f * 16 + 14] = t_prologue;

    if (a.inl_p_out) {
        const size_t base = (size_t)f * a.max_pts;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * BLOCK;
            if (i < a.max_pts) a.inl_p_out[base + i] = ((pmatched >> k) & 1u) ? (int)((pinl >> k) & 1u) : -1;
        }
    }
    if (a.inl_l_out && a.max_lines > 0) {
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int li = li0 + k * BLOCK;
            if (li < a.max_lines) a.inl_l_out[(size_t)f * a.max_lines + li] = ((lmatched >> k) & 1u) ? (int)((linl >> k) & 1u) : -1;
        }
    }
}

// ======================================================================================================================
// pose2c_kernel — compact records (PoseArgs::prev_rc / curr_rc).  Thread state: pmatched (bit k: prev point tid + k BLOCK is
// matched — only for the mask written at the end) and, in ORDINAL space (bit r: the thread's r-th matched record), inl_o
// (inliers), lv (4-bit pyramid level per ordinal), slow_o (records that are not on chip: ordinals beyond the planes + the one
// register-resident record, or a level the 16-entry sigma table does not hold — re-gathered at every use, rare by
// construction: the bench shape has none).
// ======================================================================================================================
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {  // f(integral_constant<int, I>) for I in [I, N): compile-time ordinals
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// N records {float4, double} as named members (see compact_inliers): a<R>() / b<R>() with compile-time R
typedef float f4_native __attribute__((ext_vector_type(4)));  // (a plain vector type: nothing for the scalar-replacement pass to trip over)
template <int N>
struct RecRegs {
    f4_native a0;
    double b0;
    RecRegs<N - 1> rest;
    template <int R>
    __device__ __forceinline__ f4_native& a() {
        if constexpr (R == 0) return a0;
        else return rest.template a<R - 1>();
    }
    template <int R>
    __device__ __forceinline__ double& b() {
        if constexpr (R == 0) return b0;
        else return rest.template b<R - 1>();
    }
};
template <>
struct RecRegs<0> {};

template <int NW>
constexpr int pose2c_planes() {  // record ordinals per thread in LDS: 12 x 128 x 24 bytes = 36 KB of the 40 KB share (two waves);
                                 // everything (PPT = 8) with four waves and two pairs per CU
    return NW == 2 ? 12 : STVO_POSE_MAX_POINTS / (NW * 64);
}

struct CompactRec {
    float4 uvo; // u, v of the prev stereo point; ox, oy: the matched key-point of the current frame
    double bd;  // b / disparity
    double q;   // sqrt(sigma2)
};

// a record that is not on chip: find the slot of ordinal r, follow the match index, gather both sides
__device__ __noinline__ CompactRec pose2c_fetch_slow(const float4* prev_rc, const float4* curr_rc, const int32_t* m12p, size_t pbase,
                                                     int tid, int block, unsigned pmatched, int r, double cam_b, double level_scale) {
    unsigned m = pmatched;
    for (int i = 0; i < r; ++i) m &= m - 1u;
    const size_t i = pbase + (size_t)(tid + __builtin_ctz(m) * block);
    const size_t j = m12p ? pbase + (size_t)m12p[i] : i;
    const float4 p = prev_rc[i], c = curr_rc[j];
    CompactRec rec;
    rec.uvo = make_float4(p.x, p.y, c.x, c.y);
    rec.bd = cam_b / (double)p.z;
    rec.q = sqrt(pm::level_sigma2((int)p.w, level_scale));
    return rec;
}

template <int NW, bool PROF>
__global__ __launch_bounds__(NW * 64, 2) void pose2c_kernel(PoseArgs a_by_value, double2* arena, const size_t arena_pair) {
    // The ~35 pointers and scalars of PoseArgs are read from the kernel-argument segment WHERE THEY ARE USED, through a pointer the
    // compiler cannot see through: as ordinary by-value arguments they are all loaded at the top of the kernel and stay live — most
    // of them until the epilogue — in SGPRs the evaluation loop needs (it then reloads spilled scalars with v_readlane on every
    // trip: 9 of its 137 vector instructions).  PoseArgs is the first kernel argument: offset 0 of the segment.
    (void)a_by_value;
    typedef const PoseArgs __attribute__((address_space(4))) KArgs;
    KArgs* const kargs = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    auto ka = [&]() -> KArgs& {
        KArgs* p = kargs;
        asm volatile("" : "+s"(p));
        return *p;
    };

    constexpr int BLOCK = NW * 64;
    constexpr int PPT = (STVO_POSE_MAX_POINTS + BLOCK - 1) / BLOCK;
    constexpr int LPT = (STVO_POSE_MAX_LINES + BLOCK - 1) / BLOCK;
    constexpr int KL = pose2c_planes<NW>();
    constexpr bool HAS_REG = KL < PPT;  // ordinal KL lives in registers
    static_assert(KL <= PPT && PPT <= 16, "4-bit levels of 16 ordinals in one 64-bit word");
    using Ops = BlockOps<NW>;
    extern __shared__ float4 s_ra[];                               // [KL][BLOCK] {u, v, ox, oy}
    double* const s_rb = reinterpret_cast<double*>(s_ra + KL * BLOCK);  // [KL][BLOCK] b / disparity
    __shared__ __align__(16) int s_hist[2][Ops::HIST_W];  // BlockOps::select2
    __shared__ double s_red[NW][28];
    __shared__ int s_ired[NW];
    __shared__ PoseSh s_sh;
    PoseSh* sh = &s_sh;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool w0 = wv == 0, t0 = tid == 0;
    long long rprof[3] = {0, 0, 0};
    long long tprof[5] = {0, 0, 0, 0, 0};
    long long wprof[3] = {0, 0, 0}, wave_busy = 0;
    auto tick = [&]() -> long long {
        if constexpr (PROF) return (long long)__builtin_readcyclecounter();
        else return 0ll;
    };
    const long long t_begin = tick();
    if (blockIdx.x == 0 && threadIdx.x == 0 && ka().start_flag)  // this launch has begun to take the CUs (kernels.h)
        __hip_atomic_store(ka().start_flag, ka().start_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    stvo_cam cam_f;
    if (ka().cams) {
        cam_f = ka().cams[f];
    } else {
        cam_f.fx = ka().cam.fx; cam_f.fy = ka().cam.fy; cam_f.cx = ka().cam.cx; cam_f.cy = ka().cam.cy; cam_f.b = ka().cam.b;
    }
    const pm::Cam5 cam{cam_f.fx, cam_f.fy, cam_f.cx, cam_f.cy};
    const double cam_b = cam_f.b;
    const double homog_th = ka().prm.homog_th, inv_homog = 1.0 / homog_th;

    // ---------------- ownership (as pose2p_kernel): thread t owns prev points t + k BLOCK and prev line BLOCK - 1 - t ----------------
    unsigned pmatched = 0u, pinl = 0u;
    const int n_prev_p = ka().n_prev_pts != nullptr ? min(ka().n_prev_pts[f], ka().max_pts) : 0;
    const size_t pbase = (size_t)f * ka().max_pts;
    int jj[PPT];
    {
        int init[PPT];
        const int last = ka().max_pts - 1;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {  // (addresses clamped to the pair's slot, not to its count: the loads do not wait for n_prev_pts)
            const int i = tid + k * BLOCK;
            const int ic = i < last ? i : last;
            jj[k] = ka().m12p ? ka().m12p[pbase + ic] : ic;
            init[k] = ka().init_inl_p ? ka().init_inl_p[pbase + ic] : 1;
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * BLOCK;
            if (i < n_prev_p && jj[k] >= 0) {
                pmatched |= 1u << k;
                if (init[k] != 0) pinl |= 1u << k;
            } else {
                jj[k] = 0;
            }
        }
    }
    unsigned lmatched = 0u, linl = 0u;
    const int n_prev_l = (ka().n_prev_lines != nullptr && ka().max_lines > 0) ? ka().n_prev_lines[f] : 0;
    const size_t lbase = (size_t)f * ka().max_lines;
    const int li0 = BLOCK - 1 - tid;
    int jl[LPT];
    {
        int initl[LPT];
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int li = li0 + k * BLOCK;
            const int lc = (li < n_prev_l && li < ka().max_lines) ? li : 0;
            jl[k] = (ka().m12l && ka().max_lines > 0) ? ka().m12l[lbase + lc] : lc;
            initl[k] = (ka().init_inl_l && ka().max_lines > 0) ? ka().init_inl_l[lbase + lc] : 1;
        }
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int li = li0 + k * BLOCK;
            if (li < n_prev_l && li < ka().max_lines && jl[k] >= 0) {
                lmatched |= 1u << k;
                if (initl[k] != 0) linl |= 1u << k;
            } else {
                jl[k] = 0;
            }
        }
    }
    const int n_mine = __popc(pmatched);
    unsigned* const sel = reinterpret_cast<unsigned*>(&s_hist[0][0]);  // BlockOps::select2's scratch: zero before its first use
    int sel_rot = 0;
    for (int i = tid; i < 2 * Ops::HIST_W; i += BLOCK) sel[i] = 0u;  // (the barriers of the counts below come first)
    const int n_m_p = Ops::template sum_int<true>(n_mine, s_ired);
    const int n_m_l = Ops::template sum_int<true>(__popc(lmatched), s_ired);

    // ---------------- staging: every matched record of the thread, once, ordinal by ordinal ----------------
    const unsigned m_o = n_mine >= 32 ? 0xFFFFFFFFu : ((1u << n_mine) - 1u);
    unsigned inl_o = 0u, slow_o = 0u;
    unsigned long long lv = 0ull;
    float4 xa = make_float4(0.f, 0.f, 0.f, 0.f);
    double xb = 1.0;
    {
        int r = 0;
        constexpr int STAGE_CH = PPT;  // both records of every slot in flight at once: one round trip (32 x 16 bytes per thread)
#pragma unroll
        for (int k0 = 0; k0 < PPT; k0 += STAGE_CH) {
            float4 pv[STAGE_CH], cv[STAGE_CH];
#pragma unroll
            for (int c = 0; c < STAGE_CH; ++c) {  // unconditional loads from clamped (valid) addresses: all in flight at once
                const int k = k0 + c;
                if (k >= PPT) continue;
                const int ik = tid + k * BLOCK;
                pv[c] = ka().prev_rc[pbase + (size_t)(ik < n_prev_p ? ik : 0)];  // (index 0 for the slots past the count: one cache line)
                cv[c] = ka().curr_rc[pbase + (size_t)jj[k]];
            }
#pragma unroll
            for (int c = 0; c < STAGE_CH; ++c) {
                const int k = k0 + c;
                if (k >= PPT) continue;
                if ((pmatched >> k) & 1u) {
                    const int level = (int)pv[c].w;
                    const bool in_tab = level >= 0 && level < STVO_POSE_QTAB - 1;
                    const float4 av = make_float4(pv[c].x, pv[c].y, cv[c].x, cv[c].y);
                    const double bd = cam_b / (double)pv[c].z;  // backProjection's quotient (src/pinholeStereoCamera.cpp:224)
                    if (r < KL) {
                        s_ra[r * BLOCK + tid] = av;
                        s_rb[r * BLOCK + tid] = bd;
                    } else if (HAS_REG && r == KL) {
                        xa = av;
                        xb = bd;
                    }
                    lv |= (unsigned long long)(in_tab ? level : STVO_POSE_QTAB - 1) << (4 * r);
                    if (r > (HAS_REG ? KL : KL - 1) || !in_tab) slow_o |= 1u << r;
                    if ((pinl >> k) & 1u) inl_o |= 1u << r;
                    ++r;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the number of LDS planes any lane of this WAVE uses: the trip count of the evaluation loops (wave-uniform, an SGPR)
    int n_trip = n_mine < KL ? n_mine : KL;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_trip = max(n_trip, __shfl_xor(n_trip, off, 64));
    n_trip = __builtin_amdgcn_readfirstlane(n_trip);
    // After removeOutliers the inliers are re-dealt to the threads (compact_inliers below): `dealt` (block-uniform) says that the
    // LDS planes hold ONLY inliers, in a dense layout the owner-space masks no longer describe — inl_w are then the plane ordinals
    // this thread holds.  inl_o / pmatched keep describing the thread's own prev points (the mask written at the end).
    bool dealt = false;
    unsigned inl_w = 0u;

    // key-lines: the arena, plane layout [ordinal][7 parts][thread] (as pose2p_kernel)
    double2* ar_l = arena + (size_t)f * arena_pair;
    auto load_line = [&](int k) -> pm::LineRec {
        int r = __popc(lmatched & ((1u << k) - 1u));
        asm volatile("" : "+v"(r));  // no hoisted 64-bit addresses across the iteration loop
        const double2* q = ar_l + (size_t)(r * 7) * BLOCK + tid;
        pm::LineRec L;
        const double2 v0 = q[0], v1 = q[BLOCK], v2 = q[2 * BLOCK], v3 = q[3 * BLOCK], v4 = q[4 * BLOCK], v5 = q[5 * BLOCK], v6 = q[6 * BLOCK];
        L.sP[0] = v0.x; L.sP[1] = v0.y; L.sP[2] = v1.x; L.eP[0] = v1.y; L.eP[1] = v2.x; L.eP[2] = v2.y;
        L.le[0] = v3.x; L.le[1] = v3.y; L.le[2] = v4.x; L.spl[0] = v4.y; L.spl[1] = v5.x; L.epl[0] = v5.y; L.epl[1] = v6.x;
        L.sigma2 = v6.y;
        return L;
    };
#pragma unroll
    for (int k = 0; k < LPT; ++k)
        if ((lmatched >> k) & 1u) {
            const size_t i = lbase + (size_t)(li0 + k * BLOCK);
            const size_t j = lbase + (size_t)jl[k];
            const int r = __popc(lmatched & ((1u << k) - 1u));
            double2* q = ar_l + (size_t)(r * 7) * BLOCK + tid;
            q[0] = make_double2(ka().prev_sP[i * 3 + 0], ka().prev_sP[i * 3 + 1]);
            q[BLOCK] = make_double2(ka().prev_sP[i * 3 + 2], ka().prev_eP[i * 3 + 0]);
            q[2 * BLOCK] = make_double2(ka().prev_eP[i * 3 + 1], ka().prev_eP[i * 3 + 2]);
            q[3 * BLOCK] = make_double2(ka().curr_le[j * 3 + 0], ka().curr_le[j * 3 + 1]);
            q[4 * BLOCK] = make_double2(ka().curr_le[j * 3 + 2], ka().prev_spl[i * 2 + 0]);
            q[5 * BLOCK] = make_double2(ka().prev_spl[i * 2 + 1], ka().prev_epl[i * 2 + 0]);
            q[6 * BLOCK] = make_double2(ka().prev_epl[i * 2 + 1], sqrt(ka().prev_s2l[i]));  // the record carries sqrt(sigma2) (pm::line_term_q)
        }
    // (a thread reads back only what it wrote itself: program order is enough, no fence)

    {
        const int nip = Ops::template sum_int<true>(__popc(inl_o), s_ired);
        const int nil = Ops::template sum_int<true>(__popc(linl), s_ired);
        if (w0) {
            sh->n_m_p = n_m_p;
            sh->n_m_l = n_m_l;
            sh->n_inl_p = nip;
            sh->n_inl_l = nil;
            sh->good = 1;
            sh->err_out = -1.0;  // :313
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double v = ka().init_T ? ka().init_T[(size_t)f * 16 + i] : ((i % 5 == 0) ? 1.0 : 0.0);
                sh->DT[i] = v;
                sh->DT0[i] = v;
            }
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                sh->cov[i] = 0.0;
                sh->H[i] = 0.0;
            }
        }
        __syncthreads();
    }
    const long long t_prologue = tick() - t_begin;

    // ---------------- record access ----------------
    // P = backProjection(u, v, disparity) from the record: the two subtractions and three products of
    // src/pinholeStereoCamera.cpp:225-227 on the stored quotient — what point_tail_write computed when the pair was built
    auto rebuild = [&](const float4& av, double bd, double* X, double* Y, double* Z, double* ox, double* oy) {
#pragma clang fp contract(off)
        *X = bd * ((double)av.x - cam_f.cx);
        *Y = bd * ((double)av.y - cam_f.cy);
        *Z = bd * cam_f.fx;
        *ox = (double)av.z;
        *oy = (double)av.w;
    };
    const double* const qtab = ka().q_tab;  // (hot: stays in two SGPRs)
    auto q_of = [&](unsigned nib) -> double { return qtab[nib]; };
    auto fetch_slow = [&](int r) -> CompactRec {
        return pose2c_fetch_slow(ka().prev_rc, ka().curr_rc, ka().m12p, pbase, tid, BLOCK, pmatched, r, cam_b, ka().level_scale);
    };
    // record ordinal R (compile-time) of this thread, wherever it lives
    auto fetch = [&](auto rc) -> CompactRec {
        constexpr int R = decltype(rc)::value;
        CompactRec rec;
        if (!dealt && ((slow_o >> R) & 1u)) return fetch_slow(R);
        if constexpr (R < KL) {
            rec.uvo = s_ra[R * BLOCK + tid];
            rec.bd = s_rb[R * BLOCK + tid];
        } else {
            rec.uvo = xa;
            rec.bd = xb;
        }
        rec.q = q_of((unsigned)(lv >> (4 * R)) & 15u);
        return rec;
    };
    auto residual_of = [&](const double* DT, const CompactRec& rec) -> double {
        double X, Y, Z, ox, oy;
        rebuild(rec.uvo, rec.bd, &X, &Y, &Z, &ox, &oy);
        return pm::point_residual(DT, cam, X, Y, Z, ox, oy);
    };
    // residual norms of the records in `mask` (ordinal space), by ordinal; scaled by sqrt(sigma2) when `weighted`
    auto residuals = [&](const double* DT, unsigned mask, bool weighted, double* res) {
        auto one = [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            res[R] = 0.0;
            if ((mask >> R) & 1u) {
                const CompactRec rec = fetch(rc);
                const double n = residual_of(DT, rec);
                res[R] = weighted ? n * rec.q : n;
            }
        };
        static_for<0, PPT>(one);
    };

    // ---------------- optimizeFunctions / optimizeFunctionsRobust at sh->DT ----------------
    auto pose_sgpr = [&](const double* src, double* DT) {
#pragma unroll
        for (int i = 0; i < 12; ++i) DT[i] = uni(src[i]);
    };
    auto evaluate = [&](bool robust) {
        double DT[12];
        pose_sgpr(sh->DT, DT);
        double sp = 1.0, sl = 1.0;
        if (robust) {  // pre-pass :710-781: MAD scale of the inlier residual norms
            double rp[PPT];
            const unsigned pre = dealt ? inl_w : inl_o;
            residuals(DT, pre, false, rp);
            double rlv[LPT];
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                rlv[k] = 0.0;
                if ((linl >> k) & 1u) rlv[k] = pm::line_residual(DT, cam, load_line(k));
            }
            Ops::template mad_sigma2<PPT, LPT, true>(rp, pre, sh->n_inl_p, rlv, linl, sh->n_inl_l, sel, sh->xchg, sel_rot, sp, sl);
            sp = pm::clamp_scale(sp);
            sl = pm::clamp_scale(sl);
        }
        const double isp = 1.0 / sp, isl = 1.0 / sl;  // reciprocals of the robust scales: one division per evaluation, not per feature
        const long long tw0 = tick();
        double acc[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) acc[i] = 0.0;
        auto term = [&](const float4& av, double bd, double q) {
            double X, Y, Z, ox, oy;
            rebuild(av, bd, &X, &Y, &Z, &ox, &oy);
            pm::point_term_q(acc, DT, cam, homog_th, inv_homog, X, Y, Z, ox, oy, q, robust, isp);
        };
        {
            // the thread's first inlier line is requested first (global memory), the points run while it is in flight
            pm::LineRec L0{};
            int kl0 = -1;
            if (linl) {
                kl0 = __builtin_ctz(linl);
                L0 = load_line(kl0);
            }
            // LDS planes, ascending ordinal, one record ahead; the plane pointers advance by a constant
            const unsigned hot = dealt ? inl_w : (inl_o & ~slow_o);
            const float4* pa = s_ra + tid;
            const double* pb = s_rb + tid;
            float4 av = pa[0];
            double bd = pb[0], q = q_of((unsigned)lv & 15u);
            unsigned long long lvr = lv >> 4;
            for (int r = 0; r < n_trip; ++r) {
                const int rn = r + 1 < KL ? r + 1 : r;
                const float4 avn = pa[rn * BLOCK];
                const double bdn = pb[rn * BLOCK], qn = q_of((unsigned)lvr & 15u);
                lvr >>= 4;
                if ((hot >> r) & 1u) term(av, bd, q);
                av = avn;
                bd = bdn;
                q = qn;
            }
            if constexpr (HAS_REG) {
                if ((hot >> KL) & 1u) term(xa, xb, q_of((unsigned)(lv >> (4 * KL)) & 15u));
            }
            unsigned slow = dealt ? 0u : (inl_o & slow_o);  // off-chip records (none in the usual shapes), still in ascending ordinal
            while (slow) {
                const int r = __builtin_ctz(slow);
                slow &= slow - 1u;
                const CompactRec rec = fetch_slow(r);
                term(rec.uvo, rec.bd, rec.q);
            }
            if (kl0 >= 0) pm::line_term_q(acc, DT, cam, homog_th, inv_homog, L0, robust, isl);
#pragma unroll 1
            for (int k = 0; k < LPT; ++k)
                if (((linl >> k) & 1u) && k != kl0) {
                    const pm::LineRec L = load_line(k);
                    pm::line_term_q(acc, DT, cam, homog_th, inv_homog, L, robust, isl);
                }
        }
        const long long tw1 = tick();
        Ops::template sum28_fold<true>(acc, s_red);
        const long long tw2 = tick();
        wave_busy += tw2 - tw0;
        __syncthreads();
        if (w0) {  // wave partials summed in wave order => bit-reproducible
            if (lane < 28) {
                double sum = s_red[0][lane];
#pragma unroll
                for (int w = 1; w < NW; ++w) sum += s_red[w][lane];
                store_total(sh, lane, sum);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        wprof[0] += tw1 - tw0;
        wprof[1] += tw2 - tw1;
        wprof[2] += tick() - tw2;
    };

    // ---------------- the refinement's working set: the inliers, dealt evenly to the threads ----------------
    // A thread owns the matches of ITS prev points (binomial: ~11.6 +- 1.1 of 13 at the bench shape, 13 trips per evaluation for
    // the wave) and after removeOutliers ~70 % of them are inliers, scattered over its ordinals — the refinement (8.6 of the 13.6
    // evaluations of a pair) would keep walking 13 planes for ~8 records per thread.  So the inliers are re-dealt: inlier number g
    // of the pair (threads in order, ordinals ascending) goes to plane g / BLOCK of thread g % BLOCK — every thread reads its
    // inlier records into registers, barrier, writes them to their new slots (the planes are big enough: at most KL + 1 records per
    // thread came from them), the 4-bit levels travel through the selection histogram's LDS (free between selections).  The
    // refinement then runs ceil(inliers / BLOCK) trips (~9) with every lane busy.  Skipped (block-uniform) when some inlier is
    // off chip or the dense layout would not fit the planes; the sums are then formed in a different order than pose_kernel.hip's
    // (a few roundings), the inlier masks / counts / iteration logic are untouched.
    auto compact_inliers = [&]() {
        unsigned char* const s_lvb = reinterpret_cast<unsigned char*>(&s_hist[0][0]);  // [KL * BLOCK] level nibbles
        static_assert(sizeof(s_hist) >= (size_t)KL * BLOCK, "the level bytes of a full set of planes fit the histogram scratch");
        const unsigned mov = inl_o & ~slow_o;
        const int c = __popc(mov);
        // exclusive scan of c over the threads + the number of off-chip inliers of the pair
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int n_slow = Ops::template sum_int<true>(__popc(inl_o & slow_o), s_ired);
        if (lane == 63) s_ired[wv] = incl;
        __syncthreads();
        int base = incl - c, total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int cw = s_ired[w];
            if (w < wv) base += cw;
            total += cw;
        }
        __syncthreads();
        if (n_slow != 0 || total > KL * BLOCK) return;  // block-uniform: keep the owner layout
        // (the records in a recursive struct of NAMED members, not in arrays: as `float4 ra[KL + 1]` the compiler kept twelve of them in a
        //  192-byte stack object — scratch_store_dwordx4 behind every ds_read_b128 — which was most of what the kernel still wrote to scratch
        //  after round 6 took the pivoted 6 x 6 fallbacks out of its registers: 25 MB per 1024 pairs)
        RecRegs<HAS_REG ? KL + 1 : KL> regs;
        auto take = [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            if constexpr (R < KL) {
                regs.template a<R>() = *reinterpret_cast<const f4_native*>(&s_ra[R * BLOCK + tid]);
                regs.template b<R>() = s_rb[R * BLOCK + tid];
            } else {
                regs.template a<R>() = f4_native{xa.x, xa.y, xa.z, xa.w};
                regs.template b<R>() = xb;
            }
        };
        static_for<0, HAS_REG ? KL + 1 : KL>(take);
        __syncthreads();  // every record has been read: the planes can be overwritten
        {
            int g = base;
            auto put = [&](auto rc) {
                constexpr int R = decltype(rc)::value;
                if ((mov >> R) & 1u) {
                    *reinterpret_cast<f4_native*>(&s_ra[g]) = regs.template a<R>();
                    s_rb[g] = regs.template b<R>();
                    s_lvb[g] = (unsigned char)((lv >> (4 * R)) & 15u);
                    ++g;
                }
            };
            static_for<0, HAS_REG ? KL + 1 : KL>(put);
        }
        __syncthreads();
        const int n_hold = tid < total ? (total - tid + BLOCK - 1) / BLOCK : 0;  // slots g = r BLOCK + tid below total
        unsigned long long lvn = 0ull;
#pragma unroll
        for (int r = 0; r < KL; ++r)
            if (r < n_hold) lvn |= (unsigned long long)s_lvb[r * BLOCK + tid] << (4 * r);
        lv = lvn;
        inl_w = n_hold > 0 ? ((1u << n_hold) - 1u) : 0u;
        dealt = true;
        const int first = wv * 64;  // the wave's first thread holds the most
        n_trip = first < total ? (total - first + BLOCK - 1) / BLOCK : 0;
        __syncthreads();  // every level byte has been read
        for (int i = tid; i < 2 * Ops::HIST_W; i += BLOCK) sel[i] = 0u;  // the selection scratch as BlockOps::select2 expects it
        __syncthreads();
    };

    // ---------------- removeOutliers at pose DT1 (:988-1067) ----------------
    auto remove_outliers = [&]() {
        double DT[12];
        pose_sgpr(sh->DT1, DT);
        const bool do_p = ka().prm.has_points != 0, do_l = ka().prm.has_lines != 0;
        const long long tr0 = tick();
        double resp[PPT], resl[LPT];
        residuals(DT, do_p ? m_o : 0u, true, resp);  // ALL matches, current outliers included (:998-1005)
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            resl[k] = 0.0;
            if (do_l && ((lmatched >> k) & 1u)) {
                const pm::LineRec L = load_line(k);
                resl[k] = pm::line_residual(DT, cam, L) * L.sigma2;  // L.sigma2 = sqrt(sigma2)
            }
        }
        const long long tr1 = tick();
        int cnt[2];
        Ops::template outlier_cut<PPT, LPT, true>(resp, m_o, sh->n_m_p, do_p, resl, lmatched, sh->n_m_l, do_l, ka().prm.inlier_k, inl_o, linl, sel,
                                                  sh->xchg, sel_rot, s_red, cnt);
        if (w0) {
            if (do_p) sh->n_inl_p = cnt[0];
            if (do_l) sh->n_inl_l = cnt[1];
        }
        __syncthreads();
        const long long tr2 = tick();
        compact_inliers();
        rprof[0] += tr1 - tr0;
        rprof[1] += tr2 - tr1;
        rprof[2] += tick() - tr2;
    };

    const PoseFlow fl = optimize_pose_flow(sh, [&]() -> const stvo_opt_params __attribute__((address_space(4)))& { return ka().prm; }, w0, evaluate,
                                           remove_outliers, tick, tprof, &s_red[0][0]);

    {
        const long long tq3 = tick();
        if (t0) t0_commit(sh, ka().results + f, fl.status, fl.path, fl.it0, fl.it1, false, ka().next_T ? ka().next_T + (size_t)f * 16 : nullptr);
        tprof[2] += tick() - tq3;
    }
    if (PROF && t0) {
        tprof[4] = tick() - t_begin;
#pragma unroll
        for (int i = 0; i < 5; ++i) ka().prof_out[(size_t)f * 16 + i] = tprof[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) ka().prof_out[(size_t)f * 16 + 5 + i] = wprof[i];
        ka().prof_out[(size_t)f * 16 + 14] = t_prologue;
#pragma unroll
        for (int i = 0; i < 3; ++i) ka().prof_out[(size_t)f * 16 + 10 + i] = rprof[i];  // removeOutliers: residuals, statistics, re-deal
    }
    if (PROF && lane == 0 && (wv < 6 || wv == NW - 1)) ka().prof_out[(size_t)f * 16 + 8 + (wv < 6 ? wv : 7)] = wave_busy;

    if (ka().inl_p_out) {  // back to slots: prev point tid + k BLOCK is the thread's popc(pmatched below k)-th record
        const size_t base = (size_t)f * ka().max_pts;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * BLOCK;
            const int r = __popc(pmatched & ((1u << k) - 1u));
            if (i < ka().max_pts) ka().inl_p_out[base + i] = ((pmatched >> k) & 1u) ? (int)((inl_o >> r) & 1u) : -1;
        }
    }
    if (ka().inl_l_out && ka().max_lines > 0) {
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int li = li0 + k * BLOCK;
            if (li < ka().max_lines) ka().inl_l_out[(size_t)f * ka().max_lines + li] = ((lmatched >> k) & 1u) ? (int)((linl >> k) & 1u) : -1;
        }
    }
}

struct ArenaBuf {
    double2* dev = nullptr;
    size_t bytes = 0;
    int users = 0;               // contexts that hold the stream (pose2p_retain_stream); the buffer goes with the last one
    std::vector<void*> retired;  // superseded blocks: kept until the stream is released (a captured step graph may hold the address)
};

std::mutex g_arena_mu;
std::map<std::pair<int, hipStream_t>, ArenaBuf> g_arenas;

// the record arena of a (device, stream): grown on demand, never shrunk; the last pose2p_release_stream frees it.  Returns the
// block itself (read under the lock): callers never hold a pointer into the map.
double2* arena_ptr(hipStream_t s, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_arena_mu);
    ArenaBuf& b = g_arenas[std::make_pair(dev, s)];
    if (b.bytes < bytes || !b.dev) {
        if (b.dev) b.retired.push_back(b.dev);
        b.dev = nullptr;
        b.bytes = 0;
        if (hipMalloc((void**)&b.dev, bytes > 256 ? bytes : 256) != hipSuccess) return nullptr;
        b.bytes = bytes;
    }
    return b.dev;
}

template <int NW>
constexpr int pose2p_static_lds() {
    return (int)(sizeof(PoseSh) + NW * 28 * 8 + NW * 4 + 2 * 260 * 4);
}

template <int NW>
int launch_pose2p_variant(hipStream_t s, const PoseArgs& a, int wgs_per_cu) {
    constexpr int BLOCK = NW * 64;
    constexpr int PPT = (STVO_POSE_MAX_POINTS + BLOCK - 1) / BLOCK, LPT = (STVO_POSE_MAX_LINES + BLOCK - 1) / BLOCK;
    // LDS planes: as many record ordinals per thread as the workgroup's share of the CU's 160 KB holds (a plane of one ordinal
    // is BLOCK x 48 bytes)
    const int share = (160 * 1024) / wgs_per_cu - pose2p_static_lds<NW>();
    int k_lds = share / (BLOCK * 48);
    k_lds = k_lds < 0 ? 0 : (k_lds > PPT ? PPT : k_lds);
    const int lds = k_lds * BLOCK * 48;
    const bool prof = a.prof_out != nullptr;
    const void* kfn = prof ? reinterpret_cast<const void*>(&pose2p_kernel<NW, true>) : reinterpret_cast<const void*>(&pose2p_kernel<NW, false>);
    if (lds > 48 * 1024 && !lds_opt_in(kfn, lds)) return STVO_ERR_CAPACITY;
    const size_t pair_d2 = (size_t)(PPT * 3 + LPT * 7) * BLOCK;
    double2* ab = arena_ptr(s, (size_t)a.B * pair_d2 * sizeof(double2));
    if (!ab) return STVO_ERR_HIP;
    if (prof) hipLaunchKernelGGL((pose2p_kernel<NW, true>), dim3(a.B), dim3(BLOCK), (size_t)lds, s, a, k_lds, ab, pair_d2);
    else hipLaunchKernelGGL((pose2p_kernel<NW, false>), dim3(a.B), dim3(BLOCK), (size_t)lds, s, a, k_lds, ab, pair_d2);
    return STVO_OK;
}

// the compact-record kernel: KL planes of BLOCK x 24 bytes; the arena holds the key-lines only
template <int NW>
int launch_pose2c_variant(hipStream_t s, const PoseArgs& a, int wgs_per_cu) {
    constexpr int BLOCK = NW * 64;
    constexpr int LPT = (STVO_POSE_MAX_LINES + BLOCK - 1) / BLOCK;
    constexpr int lds = pose2c_planes<NW>() * BLOCK * 24;
    static_assert(lds + pose2p_static_lds<NW>() <= (160 * 1024) / (NW == 2 ? 4 : 2), "the planes must leave room for the co-resident pairs");
    (void)wgs_per_cu;
    const bool prof = a.prof_out != nullptr;
    const void* kfn = prof ? reinterpret_cast<const void*>(&pose2c_kernel<NW, true>) : reinterpret_cast<const void*>(&pose2c_kernel<NW, false>);
    if (lds > 48 * 1024 && !lds_opt_in(kfn, lds)) return STVO_ERR_CAPACITY;
    const size_t pair_d2 = (size_t)(LPT * 7) * BLOCK;
    double2* ab = arena_ptr(s, (size_t)a.B * pair_d2 * sizeof(double2));
    if (!ab) return STVO_ERR_HIP;
    if (prof) hipLaunchKernelGGL((pose2c_kernel<NW, true>), dim3(a.B), dim3(BLOCK), (size_t)lds, s, a, ab, pair_d2);
    else hipLaunchKernelGGL((pose2c_kernel<NW, false>), dim3(a.B), dim3(BLOCK), (size_t)lds, s, a, ab, pair_d2);
    return STVO_OK;
}

}  // namespace

// A context that adopts a stream (its own, a borrowed one, the NULL stream) retains the stream's arena and releases it when it
// lets go; the blocks are freed with the LAST holder, so contexts that share a user stream (bench.py: torch's current stream) do
// not free each other's scratch.  The releasing caller has synchronised the stream and made its device current.
void pose2p_retain_stream(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_arena_mu);
    ++g_arenas[std::make_pair(dev, s)].users;
}

void pose2p_release_stream(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_arena_mu);
    const auto it = g_arenas.find(std::make_pair(dev, s));
    if (it == g_arenas.end()) return;
    if (--it->second.users > 0) return;
    if (it->second.dev) (void)hipFree(it->second.dev);
    for (void* p : it->second.retired) (void)hipFree(p);
    g_arenas.erase(it);
}

int launch_pose2p(hipStream_t s, const PoseArgs& a) {
    if (a.B <= 0) return STVO_OK;
    if (a.max_pts > STVO_POSE_MAX_POINTS || a.max_lines > STVO_POSE_MAX_LINES || a.eval_only) return STVO_ERR_CAPACITY;
    if (a.prev_rc && (!a.curr_rc || !a.q_tab)) return STVO_ERR_INVALID_ARG;
    // waves per frame pair: two (four pairs per CU) once the batch holds more than two pairs per CU, four (two pairs per CU,
    // half as many records per thread) below that — with 512 pairs on 256 CUs the two-wave variant leaves half of every CU's
    // wave slots empty (configs[3] leg, 512 streams: 701 k vs 774 k frame pairs/s).  STVO_POSE2P_NW overrides (developer).
    const int nw = dbg().pose2p_nw != DBG_UNSET && dbg().pose2p_nw > 0 ? dbg().pose2p_nw : (a.B > 2 * device_cu_count() ? 2 : 4);
    if (a.prev_rc) return nw >= 4 ? launch_pose2c_variant<4>(s, a, 2) : launch_pose2c_variant<2>(s, a, 4);
    return nw >= 4 ? launch_pose2p_variant<4>(s, a, 2) : launch_pose2p_variant<2>(s, a, 4);
}

}  // namespace stvo
