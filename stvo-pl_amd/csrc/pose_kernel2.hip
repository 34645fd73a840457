// pose_kernel2.hip — K4/K5/K6, second formulation: the whole of StereoFrameHandler::optimizePose
// (/root/reference/src/stereoFrameHandler.cpp:307-392) in one launch, one workgroup per frame pair, built for
// OCCUPANCY instead of per-wave register space.
//
// pose_kernel.hip (worker waves + one solver wave, 256 VGPRs, two waves per SIMD) is bound by the dependent-instruction
// latency of its FP64 chains: ~3.4 k cycles per point and evaluation against ~0.9 k cycles of issue
// (profiles/r02_*_pose_probe.txt).  This kernel keeps the same arithmetic and the same state machine but
//   * fits every wave into 128 VGPRs (4 waves per SIMD, 16 per CU): the pose lives in SGPRs (v_readfirstlane; every
//     FMA of the rigid transform takes it as its one scalar operand), the record of the next feature is the only
//     prefetch, sqrt(sigma2) is computed once per record, and the 28 sums are accumulated weight-first (35 instead of 56
//     instructions per feature);
//   * makes EVERY wave a worker; wave 0 also runs the serial sections, with the 6x6 systems solved on rows (lane i holds
//     row i, pivot rows travel through v_readlane into SGPRs): nothing in the kernel needs a 36-element array per lane
//     except the pivoted fall-backs for uncertified (rank-deficient) systems;
//   * keeps the matched records of the frame pair in LDS, COMPACTED (only matched features, lines first), as far as the
//     workgroup's LDS share goes — 78 KB with two workgroups per CU (NW = 8, batches), 152 KB with one (NW = 16, the
//     latency variant) — and streams the few records beyond it from L2: HBM sees every record once.
//   Same contract, same PoseArgs, same results up to the rounding of the re-associated sums (tests/test_gpu_pose.py runs
//   both kernels against the oracle).
#include <cstdlib>

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

constexpr int REC_P_BYTES = 48, REC_L_BYTES = 112;
constexpr bool POSE2_PRIO = true;  // serial sections at wave priority 3 (measured: 239 -> 230 us per 512 pairs with four waves per pair)

// NW waves per frame pair; WPE = waves per SIMD the register budget is set for (4: 128 VGPRs, 2: 256 VGPRs)
template <int NW, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void pose2_kernel(PoseArgs a, int lds_rec_bytes, const int* list, const int* list_count) {
    constexpr int BLOCK = NW * 64;
    // 6x6 systems: on ROWS (row_solve_spd & co., no 36-element arrays per lane) in the 128-VGPR variants; with the serial
    // routines of pose_math.h, executed redundantly by every lane of wave 0, in the 256-VGPR variant, where the arrays fit and
    // the serial form is the faster one (80 k vs 88 k cycles of solver time per frame pair)
    constexpr bool POSE2_ROW = WPE >= 4;
    constexpr int PPT = (STVO_POSE_MAX_POINTS + BLOCK - 1) / BLOCK;
    constexpr int LPT = (STVO_POSE_MAX_LINES + BLOCK - 1) / BLOCK;
    using Ops = BlockOps<NW>;
    extern __shared__ double s_rec[];  // [cap_l][14] line records, then [cap_p][6] point records (compacted)
    __shared__ int s_hist[2][Ops::HIST_W];  // select_kth_hist
    __shared__ double s_red[NW][28];
    __shared__ int s_ired[NW];
    __shared__ PoseSh s_sh;
    PoseSh* sh = &s_sh;
    // list != nullptr: the frame pairs list[0 .. *list_count - 1] (what pose_kernel3 left aside); workgroups beyond the count leave
    if (list != nullptr && (int)blockIdx.x >= *list_count) return;
    const int f = list != nullptr ? list[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool w0 = wv == 0;   // the wave that also runs the serial sections (all of its lanes, redundantly)
    const bool t0 = tid == 0;  // the lane that writes results to global memory
    const bool prof = a.prof_out != nullptr;
    long long tprof[5] = {0, 0, 0, 0, 0};
    long long wprof[3] = {0, 0, 0}, wave_busy = 0;  // developer aid: loop compute, fold, barrier + partial sums; per-wave busy ticks
    auto tick = [&]() -> long long { return prof ? (long long)__builtin_readcyclecounter() : 0ll; };
    const long long t_begin = tick();
    const stvo_cam cam_f = a.cams ? a.cams[f] : a.cam;
    const pm::Cam5 cam{cam_f.fx, cam_f.fy, cam_f.cx, cam_f.cy};
    const stvo_opt_params prm = a.prm;

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
#pragma unroll
    for (int k = 0; k < LPT; ++k) {
        const int li = li0 + k * BLOCK;
        if (li < n_prev_l && li < a.max_lines) {
            const int j = a.m12l ? a.m12l[lbase + li] : li;
            if (j >= 0) {
                lmatched |= 1u << k;
                if (a.init_inl_l == nullptr || a.init_inl_l[lbase + li] != 0) linl |= 1u << k;
            }
        }
    }

    // ---------------- compacted record cache in LDS ----------------
    int n_m_p, n_m_l;
    const int base_p = Ops::template excl_scan<true>(__popc(pmatched), s_ired, &n_m_p);
    const int base_l = Ops::template excl_scan<true>(__popc(lmatched), s_ired, &n_m_l);
    const int cap_l = min(n_m_l, lds_rec_bytes / REC_L_BYTES);
    const int cap_p = (lds_rec_bytes - cap_l * REC_L_BYTES) / REC_P_BYTES;
    double* s_lns = s_rec;
    double* s_pts = s_rec + (size_t)cap_l * 14;
    // (the index of a record is laundered through an empty asm at every use: otherwise the compiler hoists the 64-bit addresses
    //  of all PPT records of all six arrays — and their LDS slots — out of the iteration loop and carries ~100 VGPRs of
    //  loop-invariant addresses through the whole kernel: 115 doubles of scratch per lane, written once = 250 KB per frame pair)
    auto launder = [](int v) -> int {
        asm volatile("" : "+v"(v));
        return v;
    };
    auto load_point_global = [&](int k) -> PointRec2 {
        const size_t i = pbase + (size_t)launder(tid + k * BLOCK);
        const size_t j = a.m12p ? pbase + (size_t)a.m12p[i] : i;
        PointRec2 r;
        r.X = a.prev_P[i * 3 + 0];
        r.Y = a.prev_P[i * 3 + 1];
        r.Z = a.prev_P[i * 3 + 2];
        r.q = sqrt(a.prev_s2p[i]);
        r.ox = a.curr_pl[j * 2 + 0];
        r.oy = a.curr_pl[j * 2 + 1];
        return r;
    };
    auto point_slot = [&](int k) -> int { return base_p + __popc(pmatched & ((1u << k) - 1u)); };
    auto load_point = [&](int k) -> PointRec2 {
        const int slot = launder(point_slot(k));
        if (slot < cap_p) {
            const double2* q = reinterpret_cast<const double2*>(s_pts + (size_t)slot * 6);
            const double2 v0 = q[0], v1 = q[1], v2 = q[2];
            PointRec2 r;
            r.X = v0.x; r.Y = v0.y; r.Z = v1.x; r.ox = v1.y; r.oy = v2.x; r.q = v2.y;
            return r;
        }
        return load_point_global(k);  // the few records beyond the LDS share: streamed from L2 at every evaluation
    };
    auto line_slot = [&](int k) -> int { return base_l + __popc(lmatched & ((1u << k) - 1u)); };
    auto load_line_global = [&](int k) -> pm::LineRec {
        const size_t i = lbase + (size_t)launder(li0 + k * BLOCK);
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
        L.sigma2 = sqrt(a.prev_s2l[i]);  // the record carries sqrt(sigma2) (pm::line_term_q)
        return L;
    };
    auto load_line = [&](int k) -> pm::LineRec {
        const int slot = launder(line_slot(k));
        if (slot < cap_l) {
            const double2* q = reinterpret_cast<const double2*>(s_lns + (size_t)slot * 14);
            pm::LineRec L;
            const double2 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4], v5 = q[5], v6 = q[6];
            L.sP[0] = v0.x; L.sP[1] = v0.y; L.sP[2] = v1.x; L.eP[0] = v1.y; L.eP[1] = v2.x; L.eP[2] = v2.y;
            L.le[0] = v3.x; L.le[1] = v3.y; L.le[2] = v4.x; L.spl[0] = v4.y; L.spl[1] = v5.x; L.epl[0] = v5.y; L.epl[1] = v6.x;
            L.sigma2 = v6.y;
            return L;
        }
        return load_line_global(k);
    };
    // stage this thread's own records (thread-private slots: no barrier between staging and use), STAGE_CH records at a
    // time: with all PPT records in flight at once (96 VGPRs of loaded values next to their address arithmetic) the compiler
    // spilled 115 doubles per lane here — 250 KB of scratch per frame pair written once, 2.7x the kernel's compulsory bytes
    constexpr int STAGE_CH = PPT < 4 ? PPT : 4;
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
                const int slot = point_slot(k);
                if (slot < cap_p) {
                    double2* q = reinterpret_cast<double2*>(s_pts + (size_t)slot * 6);
                    q[0] = make_double2(rec[c].X, rec[c].Y);
                    q[1] = make_double2(rec[c].Z, rec[c].ox);
                    q[2] = make_double2(rec[c].oy, sqrt(rec[c].q));
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int k = 0; k < LPT; ++k)
        if (((lmatched >> k) & 1u) && line_slot(k) < cap_l) {
            const pm::LineRec L = load_line_global(k);
            double2* q = reinterpret_cast<double2*>(s_lns + (size_t)line_slot(k) * 14);
            q[0] = make_double2(L.sP[0], L.sP[1]);
            q[1] = make_double2(L.sP[2], L.eP[0]);
            q[2] = make_double2(L.eP[1], L.eP[2]);
            q[3] = make_double2(L.le[0], L.le[1]);
            q[4] = make_double2(L.le[2], L.spl[0]);
            q[5] = make_double2(L.spl[1], L.epl[0]);
            q[6] = make_double2(L.epl[1], L.sigma2);
        }

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
            sp = pm::clamp_scale(Ops::template mad_sigma<PPT, true>(rp, pinl, sh->n_inl_p, s_hist, &sh->xchg));
            double rlv[LPT];
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                rlv[k] = 0.0;
                if ((linl >> k) & 1u) rlv[k] = pm::line_residual(DT, cam, load_line(k));
            }
            sl = pm::clamp_scale(Ops::template mad_sigma<LPT, true>(rlv, linl, sh->n_inl_l, s_hist, &sh->xchg));
        }
        const long long tw0 = tick();
        double acc[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) acc[i] = 0.0;
        {
            // one record ahead: the next inlier's record is requested before the current one is evaluated
            unsigned todo = pinl;
            PointRec2 cur{1.0, 1.0, 1.0, 0.0, 0.0, 1.0};
            if (todo) cur = load_point(__builtin_ctz(todo));
            while (todo) {
                todo &= todo - 1u;
                PointRec2 nxt = cur;
                if (todo) nxt = load_point(__builtin_ctz(todo));
                pm::point_term_q(acc, DT, cam, prm.homog_th, cur.X, cur.Y, cur.Z, cur.ox, cur.oy, cur.q, robust, sp);
                cur = nxt;
            }
        }
#pragma unroll 1
        for (int k = 0; k < LPT; ++k)
            if ((linl >> k) & 1u) {
                const pm::LineRec L = load_line(k);
                pm::line_term_q(acc, DT, cam, prm.homog_th, L, robust, sl);
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
                sh->tot[lane] = s;
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
            t0_unpack(sh);
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
        if (prm.has_points) {
            double res[PPT];
            const int tot = sh->n_m_p;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {  // ALL matches, current outliers included (:998-1005)
                res[k] = 0.0;
                if ((pmatched >> k) & 1u) {
                    const PointRec2 r = load_point(k);
                    res[k] = pm::point_residual(DT, cam, r.X, r.Y, r.Z, r.ox, r.oy) * r.q;
                }
            }
            const double stdv = Ops::template mad_sigma<PPT, true>(res, pmatched, tot, s_hist, &sh->xchg);
            double v[3] = {0.0, 0.0, 0.0};  // mean of the samples below 2 sigma, or of all samples (src/auxiliar.cpp:405-427)
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                if ((pmatched >> k) & 1u) {
                    if (res[k] < 2.0 * stdv) {
                        v[0] += res[k];
                        v[1] += 1.0;
                    }
                    v[2] += res[k];
                }
            double t[3];
            Ops::template sum_small<3, true>(v, s_red, t);
            double mean = 0.0;
            if (tot != 0) {
                const int ksel = (int)t[1];
                mean = (ksel >= (int)(0.2 * (double)tot)) ? t[0] / (double)ksel : t[2] / (double)tot;
            }
            const double th = prm.inlier_k * stdv;
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                if (((pinl >> k) & 1u) && fabs(res[k] - mean) > th) pinl &= ~(1u << k);
            const int nip = Ops::template sum_int<true>(__popc(pinl), s_ired);
            if (w0) sh->n_inl_p = nip;
        }
        if (prm.has_lines) {
            double res[LPT];
            const int tot = sh->n_m_l;
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                res[k] = 0.0;
                if ((lmatched >> k) & 1u) {
                    const pm::LineRec L = load_line(k);
                    res[k] = pm::line_residual(DT, cam, L) * L.sigma2;  // L.sigma2 = sqrt(sigma2)
                }
            }
            const double stdv = Ops::template mad_sigma<LPT, true>(res, lmatched, tot, s_hist, &sh->xchg);
            double v[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < LPT; ++k)
                if ((lmatched >> k) & 1u) {
                    if (res[k] < 2.0 * stdv) {
                        v[0] += res[k];
                        v[1] += 1.0;
                    }
                    v[2] += res[k];
                }
            double t[3];
            Ops::template sum_small<3, true>(v, s_red, t);
            double mean = 0.0;
            if (tot != 0) {
                const int ksel = (int)t[1];
                mean = (ksel >= (int)(0.2 * (double)tot)) ? t[0] / (double)ksel : t[2] / (double)tot;
            }
            const double th = prm.inlier_k * stdv;
#pragma unroll
            for (int k = 0; k < LPT; ++k)
                if (((linl >> k) & 1u) && fabs(res[k] - mean) > th) linl &= ~(1u << k);
            const int nil = Ops::template sum_int<true>(__popc(linl), s_ired);
            if (w0) sh->n_inl_l = nil;
        }
        __syncthreads();
    };

    // ---------------- optimizePose state machine (:332-370), as in pose_kernel.hip ----------------
    int status = STVO_POSE_OK, path = 0, it0 = 0, it1 = 0;
    if (sh->n_inl_p + sh->n_inl_l >= prm.min_features) {
        int stage = 0;        // 0 = first optimisation (:335-338), 1 = refinement (:345-350), 2 = robust fallback (:359)
        int alg = prm.mode;   // 0 GN, 1 robust GN, 2 LM
        int max_it = prm.max_iters;
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
                    if (POSE2_PRIO) __builtin_amdgcn_s_setprio(3);
                    if (alg == 0) t0_gn_iter<POSE2_ROW>(sh, prm.min_error, prm.min_error_change, it);
                    else if (alg == 1) t0_gnr_iter<POSE2_ROW>(sh, prm.min_error, prm.min_error_change);
                    else t0_lm_iter<POSE2_ROW>(sh, prm.min_error, prm.min_error_change, it == 0 ? 1 : 0);
                    if (POSE2_PRIO) __builtin_amdgcn_s_setprio(0);
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
                    t0_cov_from_H<POSE2_ROW>(sh);  // :429 / :470 / :545 — H of the last evaluation (damped for LM)
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
                t0_is_good_fast<POSE2_ROW>(sh, sh->DT1, sh->err_out);
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
            max_it = prm.max_iters_ref;
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

    {
        const long long tq3 = tick();
        if (t0) t0_commit(sh, a.results + f, status, path, it0, it1);
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

// LDS share of the record cache: 160 KB per CU, WG workgroups per CU, minus the kernel's static LDS and some slack
template <int NW, int WPE>
constexpr int pose2_lds_budget() {  // 4 WPE / NW workgroups per CU share its 160 KB
    return (160 * 1024) / ((4 * WPE) / NW) - (int)(sizeof(PoseSh) + NW * 28 * 8 + NW * 4 + 2 * 260 * 4) - 768;
}

template <int NW, int WPE>
bool pose2_attr_ok() {
    return lds_opt_in(reinterpret_cast<const void*>(&pose2_kernel<NW, WPE>), pose2_lds_budget<NW, WPE>());
}

template <int NW, int WPE>
void launch_pose2_variant(hipStream_t s, const PoseArgs& a, const int* list = nullptr, const int* list_count = nullptr) {
    // no more LDS than the records of the largest possible problem need (a small batch item leaves room for other kernels)
    const long long need = (long long)a.max_pts * REC_P_BYTES + (long long)a.max_lines * REC_L_BYTES;
    int lds = pose2_attr_ok<NW, WPE>() ? pose2_lds_budget<NW, WPE>() : 48 * 1024;
    if (need < lds) lds = (int)((need + 15) & ~15ll);
    hipLaunchKernelGGL((pose2_kernel<NW, WPE>), dim3(a.B), dim3(NW * 64), (size_t)lds, s, a, lds, list, list_count);
}

}  // namespace

constexpr int POSE2_LATENCY_MAX_B = 256;  // up to one workgroup per CU: 16 waves per frame pair, everything in LDS

int launch_pose2(hipStream_t s, const PoseArgs& a) {
    if (a.B <= 0) return STVO_OK;
    if (a.max_pts > STVO_POSE_MAX_POINTS || a.max_lines > STVO_POSE_MAX_LINES) return STVO_ERR_CAPACITY;
    const char* env = std::getenv("STVO_POSE2_NW");  // developer override of the waves per frame pair
    const int force_nw = env ? std::atoi(env) : 0;
    // waves per frame pair: as many as keep ~4096 wave slots (4 per SIMD) filled with INDEPENDENT problems — a frame pair is
    // a chain of ~14 evaluate / solve rounds, and the chains of co-resident workgroups overlap each other's serial sections
    // 16 waves per frame pair while every pair has a CU to itself, 8 (two pairs per CU, records still in LDS) beyond; the 4- and
    // 2-wave variants (more pairs per CU, records mostly streamed) were measured slower and spill (NOTES.md): override only
    int nw = force_nw;
    if (nw == 0) nw = a.B <= POSE2_LATENCY_MAX_B ? 16 : 40;
    // nw == 40: four waves per pair at 256 VGPRs — two workgroups per CU with ALL records in LDS and every wave a worker
    if (nw == 40) launch_pose2_variant<4, 2>(s, a);
    else if (nw >= 16) launch_pose2_variant<16, 4>(s, a);
    else if (nw >= 8) launch_pose2_variant<8, 4>(s, a);
    else if (nw >= 4) launch_pose2_variant<4, 4>(s, a);
    else launch_pose2_variant<2, 4>(s, a);
    return STVO_OK;
}

int launch_pose2_list(hipStream_t s, const PoseArgs& a, const int* list, const int* count) {
    if (a.B <= 0) return STVO_OK;
    if (a.max_pts > STVO_POSE_MAX_POINTS || a.max_lines > STVO_POSE_MAX_LINES) return STVO_ERR_CAPACITY;
    launch_pose2_variant<4, 2>(s, a, list, count);
    return STVO_OK;
}

}  // namespace stvo
