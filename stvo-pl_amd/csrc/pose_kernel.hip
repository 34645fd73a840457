// pose_kernel.hip — K4/K5/K6: the whole of StereoFrameHandler::optimizePose
// (/root/reference/src/stereoFrameHandler.cpp:307-392) as ONE kernel launch, one workgroup per
// frame pair, FP64 throughout.
//
//   * worker waves + ONE solver wave per workgroup (same source, template flag W): a worker thread owns
//     up to PPT matched points and LPT matched lines as two bitmasks (matched / inlier); the records
//     (52 B / 116 B of live data) are streamed from L2 at every evaluation
//     with the next record in flight, so that several workgroups share a CU;
//   * optimizeFunctions / optimizeFunctionsRobust (:549-962): fused transform + project + residual
//     + 1x6 gradient + Cauchy weight (x overlap for lines) per feature, 28 FP64 partial sums per
//     thread (21 upper-triangular H + 6 g + 1 e); wave reduction = reduce-scatter on
//     v_permlane32_swap / v_permlane16_swap + a 16-lane DPP row scan, wave partials summed in wave
//     order by the solver wave => bit-reproducible;
//   * removeOutliers (:988-1067) and the robust scale (:742-781): median / MAD by exact k-th element
//     selection on register-resident values (no sort, no LDS buffer), with the reference's fabsf
//     float truncation (src/auxiliar.cpp:387-460);
//   * the 6x6 algebra, SE(3) updates and every data-dependent branch of the GN / robust-GN / LM
//     loops (:394-547) run on one lane of the solver wave (pose_math.h) and are broadcast through
//     LDS, so the control flow is block-uniform and matches the reference iteration for iteration.
#include <algorithm>
#include <cstdlib>

#include "pose_block.h"

namespace stvo {

// LDSREC: the matched records of the frame pair live in LDS for the whole optimisation (latency variant: one workgroup
// per CU, up to 152 KB of its 160 KB LDS).  Each worker thread stages ITS OWN records once (gathered through m12 from
// HBM / L2) and reads them back at every evaluation: thread-private slots, so no barrier is involved.  Measured on one
// frame pair: with the records streamed from L2 at every evaluation the workers spend ~70 % of an evaluation waiting for
// them (a CU's L1 moves a whole sector per gathered 16-byte observation), 102 k vs 27 k ticks per frame.
template <int BLOCK, int PPT, int LPT, bool W, bool LDSREC>
__device__ __forceinline__ void pose_body(const PoseArgs& a, int (*s_hist)[BlockOps<BLOCK / 64>::HIST_W], double (*s_red)[28],
                                          int* s_ired, PoseSh* sh, double* s_rec) {
    using Ops = BlockOps<BLOCK / 64>;
    const int f = blockIdx.x;
    const int tid = threadIdx.x;              // workers: 0 .. BLOCK-1 ; solver wave: BLOCK .. BLOCK+63
    const bool t0 = !W && (tid == BLOCK);     // the one lane that runs the serial algebra
    // optional phase timing (solver lane, s_memtime ticks): [0] evaluate+reduce wait, [1] iteration
    // algebra, [2] covariance + isGood + commit, [3] removeOutliers, [4] total
    long long tprof[5] = {0, 0, 0, 0, 0};
    long long wprof[3] = {0, 0, 0};  // worker lane 0: evaluate compute, reduction, robust pre-pass
    long long wave_busy = 0;
    const bool prof = a.prof_out != nullptr;
    auto tick = [&]() -> long long { return prof ? (long long)__builtin_readcyclecounter() : 0ll; };
    const long long t_begin = tick();
    const stvo_cam cam_f = a.cams ? a.cams[f] : a.cam;  // per-sequence calibration (stvo_seq_create_multi) or one for the batch
    const pm::Cam5 cam{cam_f.fx, cam_f.fy, cam_f.cx, cam_f.cy};
    const stvo_opt_params prm = a.prm;
    const double inv_homog = 1.0 / prm.homog_th;

    // ---------------- which records does this thread own?  (bitmasks only) ----------------
    // Thread t of the worker waves owns prev features i = t + k*BLOCK, k < PPT.  Only two bitmasks
    // (matched, inlier) live in registers for the whole optimisation; the record itself
    // (52 B / point, 116 B / line, gathered through m12) is re-read at every evaluation.  It comes
    // from HBM the first time and from L2 afterwards: the working set of the workgroups resident on
    // one XCD is < 4 MiB.  Keeping the records out of the VGPR file is what lets three workgroups
    // share a CU, so one pair's serial 6x6 algebra overlaps the parallel phases of the others.
    unsigned* const sel = reinterpret_cast<unsigned*>(&s_hist[0][0]);  // BlockOps::select2's scratch: zero before its first use
    int sel_rot = 0;
    if (W)
        for (int i = tid; i < 2 * Ops::HIST_W; i += BLOCK) sel[i] = 0u;  // (the barriers of the counts below come first)
    unsigned pmatched = 0u, pinl = 0u;
    // like the reference, optimizeFunctions sums whatever is in matched_pt / matched_ls; has_points /
    // has_lines only gate the matching (caller) and the two blocks of removeOutliers (:991,1026)
    const int n_prev_p = (W && a.n_prev_pts != nullptr) ? a.n_prev_pts[f] : 0;
    const size_t pbase = (size_t)f * a.max_pts;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = tid + k * BLOCK;
        if (i < n_prev_p) {
            const int j = a.m12p ? a.m12p[pbase + i] : i;
            if (j >= 0) {
                pmatched |= 1u << k;
                if (a.init_inl_p == nullptr || a.init_inl_p[pbase + i] != 0) pinl |= 1u << k;
            }
        }
    }
    // Key-lines: with few of them (<= 64 LPT, the KITTI configuration's ~80) they belong to the SOLVER wave, lane = line: it has
    // nothing else to do while the workers evaluate their points, and the two worker waves that used to carry a line term on top of
    // their point trips were what every evaluation waited for (~4 k of 17 k cycles).  Otherwise (hundreds of key-lines) to the
    // workers as before, one or two each.  los is block-uniform; BlockOps' `sc` arguments carry it.
    unsigned lmatched = 0u, linl = 0u;
    const int n_l_all = (a.n_prev_lines != nullptr && a.max_lines > 0) ? a.n_prev_lines[f] : 0;
    const bool los = a.lines_on_solver != 0 && n_l_all <= 64 * LPT;
    const bool own_l = W ? !los : los;
    const int ltid = W ? tid : tid - BLOCK, lstride = W ? BLOCK : 64;
    const int n_prev_l = own_l ? n_l_all : 0;
    const size_t lbase = (size_t)f * a.max_lines;
#pragma unroll
    for (int k = 0; k < LPT; ++k) {
        const int i = ltid + k * lstride;
        if (i < n_prev_l) {
            const int j = a.m12l ? a.m12l[lbase + i] : i;
            if (j >= 0) {
                lmatched |= 1u << k;
                if (a.init_inl_l == nullptr || a.init_inl_l[lbase + i] != 0) linl |= 1u << k;
            }
        }
    }
    struct PointRec {
        double X, Y, Z, ox, oy, s2;
    };
    // latency variant (few records per thread): the match indices stay in registers, so that fetching a record
    // is one level of loads instead of two dependent ones
    constexpr bool CACHE_J = PPT <= 6 && !LDSREC;
    int jcache[CACHE_J ? PPT : 1];
    if (CACHE_J) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * BLOCK;
            jcache[k] = ((pmatched >> k) & 1u) ? (a.m12p ? a.m12p[pbase + i] : i) : 0;
        }
    }
    // record cache: prev points with index < cap_p and prev lines with index < cap_l (everything in the latency variant; what
    // fits half a CU's LDS in the throughput variant — the rest is gathered from L2 at every evaluation)
    // (PARTIAL is a compile-time property of the small-workgroup variant: the latency variant holds everything and must not pay
    // for the index tests — they cost it 87 spilled registers)
    constexpr bool PARTIAL = LDSREC && BLOCK < 256;
    const int cap_p = !LDSREC ? 0 : (PARTIAL ? a.lds_cap_pts : a.max_pts), cap_l = !LDSREC ? 0 : (PARTIAL ? a.lds_cap_lines : a.max_lines);
    double* s_pts = s_rec;                                  // [cap_p][6]  X Y Z ox oy s2
    double* s_lns = s_rec + (size_t)cap_p * 6;              // [cap_l][14] sP eP le spl epl s2
    auto load_point_global = [&](int k) -> PointRec {
        const size_t i = pbase + (size_t)(tid + k * BLOCK);
        size_t j;
        if (CACHE_J) {
            int jj = jcache[0];
#pragma unroll
            for (int q = 1; q < (CACHE_J ? PPT : 1); ++q) jj = (k == q) ? jcache[q] : jj;
            j = pbase + (size_t)jj;
        } else {
            j = a.m12p ? pbase + (size_t)a.m12p[i] : i;
        }
        PointRec r;
        if (a.prev_rc) {  // compact stereo points of the device-resident pipeline: one 16-byte load per side (kernels.h)
            const float4 p = a.prev_rc[i], c = a.curr_rc[j];
            double P[3];
            pm::back_project(cam_f.b, cam_f.fx, cam_f.cx, cam_f.cy, (double)p.x, (double)p.y, (double)p.z, P);
            r.X = P[0]; r.Y = P[1]; r.Z = P[2];
            r.s2 = sqrt(pm::level_sigma2((int)p.w, a.level_scale));
            r.ox = (double)c.x;
            r.oy = (double)c.y;
            return r;
        }
        r.X = a.prev_P[i * 3 + 0];
        r.Y = a.prev_P[i * 3 + 1];
        r.Z = a.prev_P[i * 3 + 2];
        r.s2 = sqrt(a.prev_s2p[i]);  // records carry sqrt(sigma2): computed once for the LDS-resident ones
        r.ox = a.curr_pl[j * 2 + 0];
        r.oy = a.curr_pl[j * 2 + 1];
        return r;
    };
    auto load_point = [&](int k) -> PointRec {
        if (!LDSREC || (PARTIAL && tid + k * BLOCK >= cap_p)) return load_point_global(k);
        const double2* q = reinterpret_cast<const double2*>(s_pts + (size_t)(tid + k * BLOCK) * 6);
        const double2 v0 = q[0], v1 = q[1], v2 = q[2];
        PointRec r;
        r.X = v0.x; r.Y = v0.y; r.Z = v1.x; r.ox = v1.y; r.oy = v2.x; r.s2 = v2.y;
        return r;
    };
    auto load_line_global = [&](int k) -> pm::LineRec {
        const size_t i = lbase + (size_t)(ltid + k * lstride);
        const size_t j = a.m12l ? lbase + (size_t)a.m12l[i] : i;
        pm::LineRec L;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            L.sP[c] = a.prev_sP[i * 3 + c];
            L.eP[c] = a.prev_eP[i * 3 + c];
            L.le[c] = a.curr_le[j * 3 + c];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            L.spl[c] = a.prev_spl[i * 2 + c];
            L.epl[c] = a.prev_epl[i * 2 + c];
        }
        L.sigma2 = sqrt(a.prev_s2l[i]);  // sqrt(sigma2), see pm::line_term_q
        return L;
    };
    auto load_line = [&](int k) -> pm::LineRec {
        if (!LDSREC || (PARTIAL && ltid + k * lstride >= cap_l)) return load_line_global(k);
        const double2* q = reinterpret_cast<const double2*>(s_lns + (size_t)(ltid + k * lstride) * 14);
        pm::LineRec L;
        const double2 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4], v5 = q[5], v6 = q[6];
        L.sP[0] = v0.x; L.sP[1] = v0.y; L.sP[2] = v1.x; L.eP[0] = v1.y; L.eP[1] = v2.x; L.eP[2] = v2.y;
        L.le[0] = v3.x; L.le[1] = v3.y; L.le[2] = v4.x; L.spl[0] = v4.y; L.spl[1] = v5.x; L.epl[0] = v5.y; L.epl[1] = v6.x;
        L.sigma2 = v6.y;
        return L;
    };
    if (LDSREC && W) {  // stage this thread's own records (thread-private slots: no barrier needed)
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (((pmatched >> k) & 1u) && (!PARTIAL || tid + k * BLOCK < cap_p)) {
                const PointRec r = load_point_global(k);
                double2* q = reinterpret_cast<double2*>(s_pts + (size_t)(tid + k * BLOCK) * 6);
                q[0] = make_double2(r.X, r.Y);
                q[1] = make_double2(r.Z, r.ox);
                q[2] = make_double2(r.oy, r.s2);
            }
    }
    if (LDSREC && own_l) {
#pragma unroll
        for (int k = 0; k < LPT; ++k)
            if (((lmatched >> k) & 1u) && (!PARTIAL || ltid + k * lstride < cap_l)) {
                const pm::LineRec L = load_line_global(k);
                double2* q = reinterpret_cast<double2*>(s_lns + (size_t)(ltid + k * lstride) * 14);
                q[0] = make_double2(L.sP[0], L.sP[1]);
                q[1] = make_double2(L.sP[2], L.eP[0]);
                q[2] = make_double2(L.eP[1], L.eP[2]);
                q[3] = make_double2(L.le[0], L.le[1]);
                q[4] = make_double2(L.le[2], L.spl[0]);
                q[5] = make_double2(L.spl[1], L.epl[0]);
                q[6] = make_double2(L.epl[1], L.sigma2);
            }
    }

    {
        // matched / inlier counts of both kinds in one reduction (the solver wave's points are none: its zeros do not matter)
        const int cnt4[4] = {(int)__popc(pmatched), (int)__popc(lmatched), (int)__popc(pinl), (int)__popc(linl)};
        int tot4[4];
        Ops::template sum_int4<W>(cnt4, s_red, tot4, los);
        const int nmp = tot4[0], nml = tot4[1], nip = tot4[2], nil = tot4[3];
        if (t0) {
            sh->n_m_p = nmp;
            sh->n_m_l = nml;
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

    double fX = 1.0, fY = 1.0, fZ = 1.0, fox = 0.0, foy = 0.0, fs2 = 1.0;  // prefetched first inlier record
    int first_k = -1;
    auto prefetch_first = [&]() {
        first_k = -1;
        if (!LDSREC && W && pinl) {
            first_k = __builtin_ctz(pinl);
            const PointRec r = load_point(first_k);
            fX = r.X; fY = r.Y; fZ = r.Z; fox = r.ox; foy = r.oy; fs2 = r.s2;
        }
    };
    prefetch_first();

    // ---------------- optimizeFunctions / optimizeFunctionsRobust at sh->DT ----------------
    auto evaluate = [&](bool robust) {
        double DT[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) DT[i] = sh->DT[i];
        double sp = 1.0, sl = 1.0;
        if (robust) {  // pre-pass :710-781: MAD scale of the inlier residual norms, both kinds of features in the same rounds
            double rp[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                rp[k] = 0.0;
                if ((pinl >> k) & 1u) {
                    const PointRec r = load_point(k);
                    rp[k] = pm::point_residual(DT, cam, r.X, r.Y, r.Z, r.ox, r.oy);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep one record in flight, not PPT of them
            }
            double rl[LPT];
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                rl[k] = 0.0;
                if ((linl >> k) & 1u) rl[k] = pm::line_residual(DT, cam, load_line(k));
                __builtin_amdgcn_sched_barrier(0);
            }
            Ops::template mad_sigma2<PPT, LPT, W>(rp, pinl, sh->n_inl_p, rl, linl, sh->n_inl_l, sel, sh->xchg, sel_rot, sp, sl, los);
            sp = pm::clamp_scale(sp);
            sl = pm::clamp_scale(sl);
        }
        const double isp = 1.0 / sp, isl = 1.0 / sl;  // reciprocals of the robust scales: one division per evaluation, not per feature
        const long long tw0 = tick();
        double acc[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) acc[i] = 0.0;
        {
            // this thread's inlier points, TWO per trip (pm::point_term_t<d2>: every operation on both records, adjacent in the
            // instruction stream): a term is a chain of ~60 dependent FP64 instructions and a wave retires one of those every
            // ~8 cycles — the second record fills the gaps (latency variant: 3.3 records per thread, two waves per SIMD; the pipe
            // was ~40 % busy with one record per trip).
            // The records of the NEXT trip are in flight while this one is evaluated; the FIRST record of an evaluation was
            // requested at the end of the previous one (or by prefetch_first), i.e. while the workgroup waited for the solver wave.
            // An odd count ends with the last record evaluated twice, the second time with weight 0: an exact no-op.
            unsigned todo = pinl;
            auto next_k = [&]() -> int {
                const int k = __builtin_ctz(todo);
                todo &= todo - 1u;
                return k;
            };
            PointRec c0{1.0, 1.0, 1.0, 0.0, 0.0, 1.0}, c1 = c0;
            double m1 = 0.0;
            bool go = todo != 0u;
            if (go) {
                const int k0 = next_k();
                if (first_k == k0) {
                    c0.X = fX; c0.Y = fY; c0.Z = fZ; c0.ox = fox; c0.oy = foy; c0.s2 = fs2;
                } else {
                    c0 = load_point(k0);
                }
                c1 = c0;
                if (todo) {
                    c1 = load_point(next_k());
                    m1 = 1.0;
                }
            }
            while (go) {
                PointRec n0 = c1, n1 = c1;
                double mn = 0.0;
                go = todo != 0u;
                if (go) {
                    n0 = load_point(next_k());
                    n1 = n0;
                    if (todo) {
                        n1 = load_point(next_k());
                        mn = 1.0;
                    }
                }
                pm::point_term_t<pm::d2>(acc, DT, cam, prm.homog_th, inv_homog, pm::d2{c0.X, c1.X}, pm::d2{c0.Y, c1.Y}, pm::d2{c0.Z, c1.Z},
                                         pm::d2{c0.ox, c1.ox}, pm::d2{c0.oy, c1.oy}, pm::d2{c0.s2, c1.s2}, robust, isp, pm::d2{1.0, m1});
                c0 = n0;
                c1 = n1;
                m1 = mn;
            }
        }
#pragma unroll 1
        for (int k = 0; k < LPT; ++k)
            if ((linl >> k) & 1u) {
                const pm::LineRec L = load_line(k);
                pm::line_term_q(acc, DT, cam, prm.homog_th, inv_homog, L, robust, isl);
            }
        const long long tw1 = tick();
        prefetch_first();  // for the next evaluation; completes while this one is reduced and solved
        Ops::template sum28_fold<W>(acc, s_red, los);
        const long long tw2 = tick();
        Ops::template sum28_finish<W>(s_red, sh, los);
        wprof[0] += tw1 - tw0;
        wprof[1] += tick() - tw2;
        wprof[2] += tw2 - tw1;
        wave_busy += tw2 - tw0;  // this wave's own compute + fold (per-wave load balance, lane 0 of every worker wave)
    };

    if (a.eval_only) {
        evaluate(a.eval_robust != 0);
        if (t0) {
            double* o = a.eval_out + (size_t)f * 44;
#pragma unroll
            for (int i = 0; i < 36; ++i) o[i] = sh->H[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) o[36 + i] = sh->g[i];
            o[42] = sh->err;
            o[43] = (double)(sh->n_inl_p + sh->n_inl_l);
        }
        return;
    }

    // ---------------- removeOutliers at pose DT1 (:988-1067) ----------------
    auto remove_outliers = [&]() {
        double DT[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) DT[i] = sh->DT1[i];
        double resp[PPT], resl[LPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {  // ALL matches, current outliers included (:998-1005)
            resp[k] = 0.0;
            if (prm.has_points && ((pmatched >> k) & 1u)) {
                const PointRec r = load_point(k);
                resp[k] = pm::point_residual(DT, cam, r.X, r.Y, r.Z, r.ox, r.oy) * r.s2;  // r.s2 = sqrt(sigma2)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            resl[k] = 0.0;
            if (prm.has_lines && ((lmatched >> k) & 1u)) {
                const pm::LineRec L = load_line(k);
                resl[k] = pm::line_residual(DT, cam, L) * L.sigma2;  // sqrt(sigma2)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        int cnt[2];
        Ops::template outlier_cut<PPT, LPT, W>(resp, pmatched, sh->n_m_p, prm.has_points != 0, resl, lmatched, sh->n_m_l, prm.has_lines != 0,
                                               prm.inlier_k, pinl, linl, sel, sh->xchg, sel_rot, s_red, cnt, los);
        if (t0) {
            if (prm.has_points) sh->n_inl_p = cnt[0];
            if (prm.has_lines) sh->n_inl_l = cnt[1];
        }
        prefetch_first();  // the inlier set changed
        __syncthreads();
    };

    // ---------------- optimizePose state machine (:332-370) ----------------
    // One generic iteration loop drives GN (:394-431), robust GN (:433-480) and LM (:482-547) for
    // stage 1, the refinement and the robust fallback, so that the (large) fused evaluation and
    // the outlier removal are instantiated exactly once in the instruction stream.
    int status = STVO_POSE_OK, path = 0, it0 = 0, it1 = 0;
    if (sh->n_inl_p + sh->n_inl_l >= prm.min_features) {
        int stage = 0;        // 0 = first optimisation (:335-338), 1 = refinement (:345-350), 2 = robust fallback (:359)
        int alg = prm.mode;   // 0 GN, 1 robust GN, 2 LM
        int max_it = prm.max_iters;
        for (;;) {
            if (t0) {
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
                if (t0) {
                    if (alg == 0) t0_gn_iter(sh, prm.min_error, prm.min_error_change, it, &s_red[0][0]);
                    else if (alg == 1) t0_gnr_iter(sh, prm.min_error, prm.min_error_change, &s_red[0][0]);
                    else t0_lm_iter(sh, prm.min_error, prm.min_error_change, it == 0 ? 1 : 0, &s_red[0][0]);
                }
                __syncthreads();
                tprof[1] += tick() - tq;
                action = sh->action;
                if (action != ACT_CONTINUE) break;
            }
            long long tq2 = tick();
            if (t0) {
                if (alg == 0 && action == ACT_FAIL) {
                    sh->err_out = -1.0;  // :408-409, covariance left untouched
                } else if (alg == 1 && !sh->good) {  // :473-478
#pragma unroll
                    for (int i = 0; i < 16; ++i) sh->DT[i] = sh->DTr[i];
                    sh->err_out = -1.0;
#pragma unroll
                    for (int i = 0; i < 36; ++i) sh->cov[i] = (i % 7 == 0) ? 1.0 : 0.0;
                } else {
                    t0_cov_from_H(sh, &s_red[0][0]);  // :429 / :470 / :545 — H of the last evaluation (damped for LM)
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
            if (t0) {
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
                if (sh->n_inl_p + sh->n_inl_l >= prm.min_features) {  // :345 — restart from the INITIAL DT
                    path |= STVO_PATH_REFINED;
                    stage = 1;
                } else {
                    if (t0) pm::identity4(sh->DT);
                    status = STVO_POSE_FEW_INLIERS_AFTER;
                    __syncthreads();
                    break;
                }
            } else {  // :357-362 robust GN on everything, from the initial DT
                path |= STVO_PATH_ROBUST_FALLBACK;
                stage = 2;
                alg = 1;
            }
            max_it = prm.max_iters_ref;
            if (t0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sh->DT[i] = sh->DT0[i];
            }
            __syncthreads();
        }
    } else {
        if (t0) pm::identity4(sh->DT);
        status = STVO_POSE_FEW_INLIERS_BEFORE;
        __syncthreads();
    }


    {
        const long long tq3 = tick();
        if (t0) t0_commit(sh, a.results + f, status, path, it0, it1, a.lazy_eig != 0, a.next_T ? a.next_T + (size_t)f * 16 : nullptr);
        tprof[2] += tick() - tq3;
    }
    if (prof && t0) {
        tprof[4] = tick() - t_begin;
#pragma unroll
        for (int i = 0; i < 5; ++i) a.prof_out[(size_t)f * 16 + i] = tprof[i];
    }
    if (prof && W && (tid & 63) == 0 && (tid >> 6) < 8) a.prof_out[(size_t)f * 16 + 8 + (tid >> 6)] = wave_busy;
    if (prof && W && tid == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) a.prof_out[(size_t)f * 16 + 5 + i] = wprof[i];
    }

    if (W && a.inl_p_out) {
        const size_t base = (size_t)f * a.max_pts;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * BLOCK;
            if (i < a.max_pts) a.inl_p_out[base + i] = ((pmatched >> k) & 1u) ? (int)((pinl >> k) & 1u) : -1;
        }
    }
    if (own_l && a.inl_l_out && a.max_lines > 0) {
        const size_t base = (size_t)f * a.max_lines;
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int i = ltid + k * lstride;
            if (i < a.max_lines) a.inl_l_out[base + i] = ((lmatched >> k) & 1u) ? (int)((linl >> k) & 1u) : -1;
        }
    }
    if (W && los && a.inl_l_out)  // the solver's lanes cover lines 0 .. 64 LPT - 1: nothing is matched beyond them
        for (int i = 64 * LPT + tid; i < a.max_lines; i += BLOCK) a.inl_l_out[(size_t)f * a.max_lines + i] = -1;
}

template <int BLOCK, int PPT, int LPT, bool LDSREC>
__global__ __launch_bounds__(BLOCK + 64, 2) void pose_kernel(PoseArgs a) {  // >= 2 waves/SIMD => <= 256 VGPRs, 2 workgroups per CU
    extern __shared__ double s_rec[];  // LDSREC: [max_pts][6] + [max_lines][14] doubles (dynamic, sized at launch)
    __shared__ __align__(16) int s_hist[2][BlockOps<BLOCK / 64>::HIST_W];  // BlockOps::select2
    __shared__ double s_red[BLOCK / 64 + 1][28];  // (+ 1: the solver wave's partial sums when it owns the key-lines)
    __shared__ int s_ired[BLOCK / 64 + 1];
    __shared__ PoseSh s_sh;
    // This kernel is latency-bound (dependent FP64 chains, barriers) and is meant to run CONCURRENTLY with
    // the VALU-saturating matching kernel of the next batch (stvo_ctx_set_overlap): give its waves issue
    // priority so its critical path is not stretched by the co-resident popcount waves.
    __builtin_amdgcn_s_setprio(3);
    if (a.wait_flag) {  // results of another stream (PoseArgs::wait_flag): normally long there — one L2 round trip
        __shared__ int s_timed_out;
        if (threadIdx.x == 0) {  // bounded (~2 s): a signal that never comes (a failed launch on the other stream) must not hang the device
            int spin = 0;
            for (; spin < (1 << 23) && (int)(__hip_atomic_load(a.wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - a.wait_value) < 0; ++spin)
                __builtin_amdgcn_s_sleep(8);
            s_timed_out = spin >= (1 << 23);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (s_timed_out) {
            // The key-line stream never signalled: its match indices and line set are not to be trusted.  No pose from half the
            // input — the pair reports STVO_POSE_INTERNAL in band (pose held: DT = I, err = -1, like a rejected solution) and the
            // by-product fetch is NOT published, so stvo_seq_fetch_matches times out on the host side instead of handing over indices.
            if (threadIdx.x == 0) {
                stvo_pose_result* out = a.results + blockIdx.x;
                for (int i = 0; i < 16; ++i) out->T[i] = out->T_opt[i] = (i % 5 == 0) ? 1.0 : 0.0;
                for (int i = 0; i < 36; ++i) out->cov[i] = 0.0;
                for (int i = 0; i < 6; ++i) out->cov_eig[i] = 0.0;
                out->err = out->err_opt = -1.0;
                out->status = STVO_POSE_INTERNAL;
                out->path = 0;
                out->iters[0] = out->iters[1] = 0;
                out->n_matched_pt = out->n_matched_ls = out->n_inliers_pt = out->n_inliers_ls = 0;
                if (a.next_T)
                    for (int i = 0; i < 16; ++i) a.next_T[(size_t)blockIdx.x * 16 + i] = (i % 5 == 0) ? 1.0 : 0.0;
            }
            // the by-products of this pair: nothing matched, nothing an inlier — stvo_seq_fetch_inliers / the counts of stvo_seq_read
            // must not hand the PREVIOUS step's flags over as this step's
            if (a.inl_p_out)
                for (int i = threadIdx.x; i < a.max_pts; i += blockDim.x) a.inl_p_out[(size_t)blockIdx.x * a.max_pts + i] = -1;
            if (a.inl_l_out)
                for (int i = threadIdx.x; i < a.max_lines; i += blockDim.x) a.inl_l_out[(size_t)blockIdx.x * a.max_lines + i] = -1;
            return;
        }
    }
    if (a.fetch_dst && blockIdx.x == 0 && threadIdx.x >= BLOCK) {  // the solver wave has nothing to do until the first reduction
        const int lane = threadIdx.x - BLOCK;
        for (unsigned i = lane; i < a.fetch_n16; i += 64) a.fetch_dst[i] = a.fetch_src[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // the wave's stores have reached the host
        if (lane == 0) __hip_atomic_store(a.fetch_flag, a.fetch_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x < BLOCK)
        pose_body<BLOCK, PPT, LPT, true, LDSREC>(a, s_hist, s_red, s_ired, &s_sh, s_rec);   // worker waves
    else
        pose_body<BLOCK, PPT, LPT, false, LDSREC>(a, s_hist, s_red, s_ired, &s_sh, s_rec);  // solver wave
}

// Two instantiations of the same kernel:
//   throughput: 3 worker waves + 1 solver wave (256 threads, <= 256 VGPRs): two workgroups per CU and room
//               next to the matching kernel in overlap mode; used when the batch has more workgroups than CUs;
//   latency:    7 worker waves + 1 solver wave (512 threads): the parallel phases of ONE frame pair run ~2.3x
//               faster; used for small batches (single-stream operation through the handler API), where every
//               workgroup has a CU to itself anyway.
constexpr int POSE_BLOCK_T = 192, POSE_BLOCK_L = 448;
constexpr int POSE_LATENCY_MAX_B = 256;
// dynamic LDS available to the record cache: 160 KB per CU minus the kernel's static LDS (PoseSh, partial sums, counters)
constexpr size_t POSE_LDSREC_MAX_BYTES = (size_t)160 * 1024 - 8 * 1024;
constexpr size_t POSE_LDSREC_T_BYTES = (size_t)80 * 1024 - 6 * 1024;  // throughput variant: two workgroups share a CU

// lds_budget: bytes of dynamic LDS the record cache may use (LDSREC only): lines first (their gather is the longer chain),
// points with what is left
template <int BLK, bool LDSREC>
static void launch_pose_variant(hipStream_t s, const PoseArgs& a_in, size_t lds_budget = 0) {
    constexpr int PPT = (STVO_POSE_MAX_POINTS + BLK - 1) / BLK;
    constexpr int LPT = (STVO_POSE_MAX_LINES + BLK - 1) / BLK;
    PoseArgs a = a_in;
    a.lds_cap_pts = a.lds_cap_lines = 0;
    a.lines_on_solver = dbg().pose_los != 0 ? 1 : 0;
    size_t lds = 0;
    if (LDSREC) {
        const size_t lb = 14 * sizeof(double), pb = 6 * sizeof(double);
        if ((size_t)a.max_lines * lb + (size_t)a.max_pts * pb <= lds_budget) {  // everything fits (latency variant)
            a.lds_cap_lines = a.max_lines;
            a.lds_cap_pts = a.max_pts;
        } else {  // an eighth for the lines (86 of them in 76 KB: the KITTI configuration detects ~100 per image)
            a.lds_cap_lines = (int)std::min<size_t>((size_t)a.max_lines, lds_budget / 8 / lb);
            a.lds_cap_pts = (int)std::min<size_t>((size_t)a.max_pts, (lds_budget - (size_t)a.lds_cap_lines * lb) / pb);
        }
        lds = ((size_t)a.lds_cap_pts * 6 + (size_t)a.lds_cap_lines * 14) * sizeof(double);
    }
    hipLaunchKernelGGL((pose_kernel<BLK, PPT, LPT, LDSREC>), dim3(a.B), dim3(BLK + 64), lds, s, a);
}

// More than the default 64 KB of dynamic LDS needs an explicit opt-in; done once.  false: the runtime refused, callers use
// the streamed-record variant instead.
template <int BLK>
static bool pose_ldsrec_available() {
    constexpr int PPT = (STVO_POSE_MAX_POINTS + BLK - 1) / BLK;
    constexpr int LPT = (STVO_POSE_MAX_LINES + BLK - 1) / BLK;
    return lds_opt_in(reinterpret_cast<const void*>(&pose_kernel<BLK, PPT, LPT, true>), (int)POSE_LDSREC_MAX_BYTES);
}

namespace {
__global__ void stream_signal_kernel(unsigned* flag, unsigned value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// one thread: leaves when *flag has reached `value` (wrap-safe) or after ~0.5 ms — a scheduling hint (kernels.h: launch_stream_gate)
__global__ void stream_gate_kernel(const unsigned* flag, unsigned value) {
    // (the wave holds a VGPR granule of its SIMD while it waits: one wave of the 256-register pose kernel does not fit beside it, so it
    //  must leave quickly once that kernel has begun — polls ~0.4 us apart)
    for (int spin = 0; spin < 3000; ++spin) {
        if ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) >= 0) break;
        __builtin_amdgcn_s_sleep(16);
    }
}
}  // namespace
void launch_stream_signal(hipStream_t s, unsigned* flag, unsigned value) {
    hipLaunchKernelGGL(stream_signal_kernel, dim3(1), dim3(1), 0, s, flag, value);
}
void launch_stream_gate(hipStream_t s, const unsigned* flag, unsigned value) {
    hipLaunchKernelGGL(stream_gate_kernel, dim3(1), dim3(1), 0, s, flag, value);
}
bool pose_start_flag_ok(const PoseArgs& a) {  // launch_pose below takes the batch kernel on compact records (pose2c_kernel)
    const int which = dbg().pose_kernel;
    return a.B > 0 && !a.eval_only && a.prev_rc != nullptr && (which == 4 || (which != 1 && a.B > POSE_LATENCY_MAX_B));
}
bool pose_inline_sync_ok(int B) { return B >= 1 && B <= 16 && dbg().pose_kernel != 4; }

int launch_pose(hipStream_t s, const PoseArgs& a) {
    if (a.B <= 0) return STVO_OK;
    if ((a.wait_flag || a.fetch_dst) && !pose_inline_sync_ok(a.B)) return STVO_ERR_INVALID_ARG;
    // Two formulations (round 4: the 128-VGPR / compacted-LDS kernel of round 2 and the owner + evaluator experiment of round 3
    // are gone — both measured slower than what is here, NOTES.md):
    //   * up to 256 frame pairs, and for single evaluations (stvo_normal_eq): this file's latency variant — one workgroup per CU,
    //     seven worker waves + a solver wave, every record resident in LDS (121 us for one pair);
    //   * larger batches: pose_kernel2p.hip — thread-private records, two waves per pair at 256 VGPRs, four pairs per CU.
    // STVO_POSE_KERNEL = 1 / 4 (debug_switches.h) force either for every batch size: the parity tests run both everywhere.
    const int which = dbg().pose_kernel;
    if (!a.eval_only && (which == 4 || (which != 1 && a.B > POSE_LATENCY_MAX_B))) return launch_pose2p(s, a);
    if (a.max_pts > STVO_POSE_MAX_POINTS || a.max_lines > STVO_POSE_MAX_LINES) return STVO_ERR_CAPACITY;
    const size_t rec_bytes = ((size_t)a.max_pts * 6 + (size_t)a.max_lines * 14) * sizeof(double);
    // throughput variant: two workgroups per CU, each with half of the CU's LDS as record cache — most records are then read
    // from HBM once instead of at every evaluation (the gathers of ~64 co-resident pairs overflow an XCD's 4 MB L2)
    const bool lds_t = dbg().pose_lds_t != 0;  // (unset: on)
    if (a.B <= POSE_LATENCY_MAX_B && rec_bytes <= POSE_LDSREC_MAX_BYTES && pose_ldsrec_available<POSE_BLOCK_L>())
        launch_pose_variant<POSE_BLOCK_L, true>(s, a, POSE_LDSREC_MAX_BYTES);   // one workgroup per CU: records resident in LDS
    else if (a.B <= POSE_LATENCY_MAX_B)
        launch_pose_variant<POSE_BLOCK_L, false>(s, a);
    else if (lds_t && pose_ldsrec_available<POSE_BLOCK_T>())
        launch_pose_variant<POSE_BLOCK_T, true>(s, a, POSE_LDSREC_T_BYTES);
    else
        launch_pose_variant<POSE_BLOCK_T, false>(s, a);
    return STVO_OK;
}

}  // namespace stvo
