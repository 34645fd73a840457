// pose_kernel.hip — K4/K5/K6: the whole of StereoFrameHandler::optimizePose
// (/root/reference/src/stereoFrameHandler.cpp:307-392) as ONE kernel launch, one workgroup per
// frame pair, FP64 throughout.
//
//   * each thread owns up to PPT matched points and LPT matched lines; their records (52 B / 116 B
//     of live data) are read from HBM exactly ONCE and then stay in VGPRs
//     for all <= 15 optimizeFunctions evaluations, the outlier removal and the commit;
//   * optimizeFunctions / optimizeFunctionsRobust (:549-962): fused transform + project + residual
//     + 1x6 gradient + Cauchy weight (x overlap for lines) per feature, 28 FP64 partial sums per
//     thread (21 upper-triangular H + 6 g + 1 e), fixed-order reduction: xor-butterfly inside
//     each wave64, then wave partials summed in wave order through LDS => bit-reproducible;
//   * removeOutliers (:988-1067) and the robust scale (:742-781): residuals compacted into an LDS
//     buffer, bitonic sort, median / MAD with the reference's fabsf float truncation
//     (src/auxiliar.cpp:387-460);
//   * the 6x6 algebra, SE(3) updates and every data-dependent branch of the GN / robust-GN / LM
//     loops (:394-547) run on thread 0 from registers (pose_math.h) and are broadcast through LDS,
//     so the control flow is block-uniform and matches the reference iteration for iteration.
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
    int action, good, n_inl_p, n_inl_l, n_m_p, n_m_l, evals, itmp;
};

template <int BLOCK>
struct BlockOps {
    static constexpr int NW = BLOCK / 64;

    // xor-butterfly inside the wave: every lane ends with the wave total (fixed order)
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

    // 28-vector block sum -> sh->tot[0..27], valid for every thread after return
    static __device__ __forceinline__ void sum28(double* acc, double (*red)[28], PoseSh* sh) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 28; ++k) {
            const double s = wave_sum(acc[k]);
            if (lane == 0) red[wv][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < 28) {
            double s = red[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w][threadIdx.x];
            sh->tot[threadIdx.x] = s;
        }
        __syncthreads();
    }

    // up to 4 doubles, result in sh->stat[0..n) for every thread
    template <int N>
    static __device__ __forceinline__ void sum_small(const double* v, double (*red)[28], double* out) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double s = wave_sum(v[k]);
            if (lane == 0) red[wv][k] = s;
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

    static __device__ __forceinline__ int sum_int(int v, int* ired) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int s = wave_sum_i(v);
        if (lane == 0) ired[wv] = s;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += ired[w];
        __syncthreads();
        return t;
    }

    // exclusive scan of per-thread counts (thread order); returns this thread's offset, *total = sum
    static __device__ __forceinline__ int excl_scan(int count, int* ired, int* total) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int incl = count;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) ired[wv] = incl;
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

    // ascending bitonic sort of buf[0..n2), n2 a power of two
    static __device__ __forceinline__ void bitonic_sort(double* buf, int n2) {
        for (int k = 2; k <= n2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < n2; i += BLOCK) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const double a = buf[i], b = buf[ixj];
                        const bool asc = (i & k) == 0;
                        if ((a > b) == asc && a != b) {
                            buf[i] = b;
                            buf[ixj] = a;
                        }
                    }
                }
                __syncthreads();
            }
    }

    // 1.4826 * MAD of buf[0..n) (values already compacted; buf is destroyed).  Follows
    // vector_stdv_mad / the first half of vector_mean_stdv_mad, src/auxiliar.cpp:395-404,447-457.
    // n == 0 -> 0.  Result returned to every thread.
    static __device__ __forceinline__ double mad_sigma(double* buf, int n) {
        if (n == 0) return 0.0;  // block-uniform
        int n2 = 1;
        while (n2 < n) n2 <<= 1;
        for (int i = n + threadIdx.x; i < n2; i += BLOCK) buf[i] = __builtin_inf();
        __syncthreads();
        bitonic_sort(buf, n2);
        const double median = buf[n / 2];
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += BLOCK) buf[i] = (double)fabsf((float)(buf[i] - median));
        __syncthreads();
        bitonic_sort(buf, n2);
        const double s = 1.4826 * buf[n / 2];
        __syncthreads();
        return s;
    }
};

__device__ __forceinline__ double norm3(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// ---- thread-0 sections (kept out of line: each instantiates fully unrolled 6x6 algebra) --------

__device__ __noinline__ void t0_unpack(PoseSh* sh) {
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

// body of gaussNewtonOptimization after optimizeFunctions (:405-428)
__device__ __noinline__ void t0_gn_iter(PoseSh* sh, double min_error, double min_error_change, int it) {
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
    pm::solve6(sh->H, sh->g, inc, nullptr);
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
__device__ __noinline__ void t0_gnr_iter(PoseSh* sh, double min_error, double min_error_change) {
    t0_unpack(sh);
    const double err = sh->err;
    if (fabs(err - sh->err_prev) < min_error_change || err < min_error) {
        sh->action = ACT_BREAK;
        return;
    }
    double inc[6], DT[16], lad;
    pm::solve6(sh->H, sh->g, inc, &lad);
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
__device__ __noinline__ void t0_lm_iter(PoseSh* sh, double min_error, double min_error_change, int first) {
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
        pm::solve6(sh->H, sh->g, inc, nullptr);
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
    pm::solve6(sh->H, sh->g, inc, nullptr);
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

__device__ __noinline__ void t0_cov_from_H(PoseSh* sh) {
    double Hi[36];
    pm::inverse6(sh->H, Hi);
#pragma unroll
    for (int i = 0; i < 36; ++i) sh->cov[i] = Hi[i];
}

// isGoodSolution(DT, cov, err) -> sh->good; eigenvalues left in sh->eig
__device__ __noinline__ void t0_is_good(PoseSh* sh, const double* DT, double err) {
    double w[6];
    pm::eig6(sh->cov, w);
#pragma unroll
    for (int i = 0; i < 6; ++i) sh->eig[i] = w[i];
    sh->good = pm::is_good_solution(DT, w, err) ? 1 : 0;
}

__device__ __noinline__ void t0_commit(PoseSh* sh, stvo_pose_result* out, int status, int path, int it0, int it1) {
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

template <int BLOCK, int PPT, int LPT>
__global__ __launch_bounds__(BLOCK) void pose_kernel(PoseArgs a) {
    using Ops = BlockOps<BLOCK>;
    constexpr int NW = BLOCK / 64;
    __shared__ double s_sort[BLOCK * PPT];
    __shared__ double s_red[NW][28];
    __shared__ int s_ired[NW];
    __shared__ PoseSh s_sh;
    PoseSh* sh = &s_sh;

    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const pm::Cam5 cam{a.cam.fx, a.cam.fy, a.cam.cx, a.cam.cy};
    const stvo_opt_params prm = a.prm;

    // ---------------- load the matched records (HBM -> VGPRs, once) ----------------
    double Px[PPT], Py[PPT], Pz[PPT], ox[PPT], oy[PPT], s2[PPT];
    unsigned pmatched = 0u, pinl = 0u;
    // like the reference, optimizeFunctions sums whatever is in matched_pt / matched_ls; has_points /
    // has_lines only gate the matching (caller) and the two blocks of removeOutliers (:991,1026)
    const int n_prev_p = (a.n_prev_pts != nullptr) ? a.n_prev_pts[f] : 0;
    {
        const size_t base = (size_t)f * a.max_pts;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * BLOCK;
            Px[k] = Py[k] = Pz[k] = 1.0;
            ox[k] = oy[k] = 0.0;
            s2[k] = 1.0;
            if (i < n_prev_p) {
                const int j = a.m12p ? a.m12p[base + i] : i;
                if (j >= 0) {
                    Px[k] = a.prev_P[(base + i) * 3 + 0];
                    Py[k] = a.prev_P[(base + i) * 3 + 1];
                    Pz[k] = a.prev_P[(base + i) * 3 + 2];
                    s2[k] = a.prev_s2p[base + i];
                    ox[k] = a.curr_pl[(base + j) * 2 + 0];
                    oy[k] = a.curr_pl[(base + j) * 2 + 1];
                    pmatched |= 1u << k;
                    if (a.init_inl_p == nullptr || a.init_inl_p[base + i] != 0) pinl |= 1u << k;
                }
            }
        }
    }
    pm::LineRec L[LPT];
    unsigned lmatched = 0u, linl = 0u;
    const int n_prev_l = (a.n_prev_lines != nullptr && a.max_lines > 0) ? a.n_prev_lines[f] : 0;
    {
        const size_t base = (size_t)f * a.max_lines;
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int i = tid + k * BLOCK;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                L[k].sP[c] = 1.0;
                L[k].eP[c] = 1.0;
                L[k].le[c] = 0.0;
            }
            L[k].spl[0] = L[k].spl[1] = L[k].epl[0] = L[k].epl[1] = 0.0;
            L[k].sigma2 = 1.0;
            if (i < n_prev_l) {
                const int j = a.m12l ? a.m12l[base + i] : i;
                if (j >= 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        L[k].sP[c] = a.prev_sP[(base + i) * 3 + c];
                        L[k].eP[c] = a.prev_eP[(base + i) * 3 + c];
                        L[k].le[c] = a.curr_le[(base + j) * 3 + c];
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        L[k].spl[c] = a.prev_spl[(base + i) * 2 + c];
                        L[k].epl[c] = a.prev_epl[(base + i) * 2 + c];
                    }
                    L[k].sigma2 = a.prev_s2l[base + i];
                    lmatched |= 1u << k;
                    if (a.init_inl_l == nullptr || a.init_inl_l[base + i] != 0) linl |= 1u << k;
                }
            }
        }
    }

    {
        const int nmp = Ops::sum_int(__popc(pmatched), s_ired);
        const int nml = Ops::sum_int(__popc(lmatched), s_ired);
        const int nip = Ops::sum_int(__popc(pinl), s_ired);
        const int nil = Ops::sum_int(__popc(linl), s_ired);
        if (tid == 0) {
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

    // ---------------- optimizeFunctions / optimizeFunctionsRobust at sh->DT ----------------
    auto evaluate = [&](bool robust) {
        double DT[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) DT[i] = sh->DT[i];
        double sp = 1.0, sl = 1.0;
        if (robust) {  // pre-pass :710-781: MAD scale of the inlier residual norms
            int tot = 0;
            int off = Ops::excl_scan(__popc(pinl), s_ired, &tot);
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                if ((pinl >> k) & 1u) s_sort[off++] = pm::point_residual(DT, cam, Px[k], Py[k], Pz[k], ox[k], oy[k]);
            __syncthreads();
            sp = pm::clamp_scale(Ops::mad_sigma(s_sort, tot));
            off = Ops::excl_scan(__popc(linl), s_ired, &tot);
#pragma unroll
            for (int k = 0; k < LPT; ++k)
                if ((linl >> k) & 1u) s_sort[off++] = pm::line_residual(DT, cam, L[k]);
            __syncthreads();
            sl = pm::clamp_scale(Ops::mad_sigma(s_sort, tot));
        }
        double acc[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) acc[i] = 0.0;
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if ((pinl >> k) & 1u)
                pm::point_term(acc, DT, cam, prm.homog_th, Px[k], Py[k], Pz[k], ox[k], oy[k], s2[k], robust, sp);
#pragma unroll
        for (int k = 0; k < LPT; ++k)
            if ((linl >> k) & 1u) pm::line_term(acc, DT, cam, prm.homog_th, L[k], robust, sl);
        Ops::sum28(acc, s_red, sh);
    };

    if (a.eval_only) {
        evaluate(a.eval_robust != 0);
        if (tid == 0) {
            t0_unpack(sh);
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

    // ---------------- the three optimisers; each works on sh->DT, writes sh->cov / sh->err_out ----
    auto run_gn = [&](int max_iters) -> int {  // :394-431
        if (tid == 0) sh->err_prev = 999999999.9;
        int evals = 0, action = ACT_BREAK;
        for (int it = 0; it < max_iters; ++it) {
            evaluate(false);
            ++evals;
            if (tid == 0) t0_gn_iter(sh, prm.min_error, prm.min_error_change, it);
            __syncthreads();
            action = sh->action;
            if (action != ACT_CONTINUE) break;
        }
        if (tid == 0) {
            if (action == ACT_FAIL)
                sh->err_out = -1.0;  // :408-409, covariance left untouched
            else {
                t0_cov_from_H(sh);   // :429  (H of the last evaluation)
                sh->err_out = max_iters > 0 ? sh->err : 0.0;
            }
        }
        __syncthreads();
        return evals;
    };
    auto run_gnr = [&](int max_iters) -> int {  // :433-480
        if (tid == 0) {
            sh->err_prev = 999999999.9;
            sh->good = 1;
#pragma unroll
            for (int i = 0; i < 16; ++i) sh->DTr[i] = sh->DT[i];
        }
        int evals = 0;
        for (int it = 0; it < max_iters; ++it) {
            evaluate(true);
            ++evals;
            if (tid == 0) t0_gnr_iter(sh, prm.min_error, prm.min_error_change);
            __syncthreads();
            if (sh->action != ACT_CONTINUE) break;
        }
        if (tid == 0) {
            if (sh->good) {
                t0_cov_from_H(sh);
                sh->err_out = max_iters > 0 ? sh->err : 0.0;
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) sh->DT[i] = sh->DTr[i];
                sh->err_out = -1.0;
#pragma unroll
                for (int i = 0; i < 36; ++i) sh->cov[i] = (i % 7 == 0) ? 1.0 : 0.0;
            }
        }
        __syncthreads();
        return evals;
    };
    auto run_lm = [&](int max_iters) -> int {  // :482-547
        evaluate(false);
        int evals = 1;
        if (tid == 0) t0_lm_iter(sh, prm.min_error, prm.min_error_change, 1);
        __syncthreads();
        for (int it = 1; it < max_iters; ++it) {
            evaluate(false);
            ++evals;
            if (tid == 0) t0_lm_iter(sh, prm.min_error, prm.min_error_change, 0);
            __syncthreads();
            if (sh->action != ACT_CONTINUE) break;
        }
        if (tid == 0) {
            t0_cov_from_H(sh);  // :545 — the damped H of the last solve
            sh->err_out = sh->err;
        }
        __syncthreads();
        return evals;
    };
    auto run_mode = [&](int mode, int iters) -> int {
        if (mode == 1) return run_gnr(iters);
        if (mode == 2) return run_lm(iters);
        return run_gn(iters);
    };

    // ---------------- removeOutliers at pose DT1 (:988-1067) ----------------
    auto remove_outliers = [&]() {
        double DT[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) DT[i] = sh->DT1[i];
        if (prm.has_points) {
            double res[PPT];
            int tot = 0;
            int off = Ops::excl_scan(__popc(pmatched), s_ired, &tot);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                res[k] = 0.0;
                if ((pmatched >> k) & 1u) {  // ALL matches, current outliers included
                    res[k] = pm::point_residual(DT, cam, Px[k], Py[k], Pz[k], ox[k], oy[k]) * sqrt(s2[k]);
                    s_sort[off++] = res[k];
                }
            }
            __syncthreads();
            const double stdv = Ops::mad_sigma(s_sort, tot);
            // mean of the samples below 2 sigma, or of all samples (src/auxiliar.cpp:405-427)
            double v[3] = {0.0, 0.0, 0.0};
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
            Ops::template sum_small<3>(v, s_red, t);
            double mean = 0.0;
            if (tot != 0) {
                const int ksel = (int)t[1];
                mean = (ksel >= (int)(0.2 * (double)tot)) ? t[0] / (double)ksel : t[2] / (double)tot;
            }
            const double th = prm.inlier_k * stdv;
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                if (((pinl >> k) & 1u) && fabs(res[k] - mean) > th) pinl &= ~(1u << k);
            const int nip = Ops::sum_int(__popc(pinl), s_ired);
            if (tid == 0) sh->n_inl_p = nip;
        }
        if (prm.has_lines) {
            double res[LPT];
            int tot = 0;
            int off = Ops::excl_scan(__popc(lmatched), s_ired, &tot);
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                res[k] = 0.0;
                if ((lmatched >> k) & 1u) {
                    res[k] = pm::line_residual(DT, cam, L[k]) * sqrt(L[k].sigma2);
                    s_sort[off++] = res[k];
                }
            }
            __syncthreads();
            const double stdv = Ops::mad_sigma(s_sort, tot);
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
            Ops::template sum_small<3>(v, s_red, t);
            double mean = 0.0;
            if (tot != 0) {
                const int ksel = (int)t[1];
                mean = (ksel >= (int)(0.2 * (double)tot)) ? t[0] / (double)ksel : t[2] / (double)tot;
            }
            const double th = prm.inlier_k * stdv;
#pragma unroll
            for (int k = 0; k < LPT; ++k)
                if (((linl >> k) & 1u) && fabs(res[k] - mean) > th) linl &= ~(1u << k);
            const int nil = Ops::sum_int(__popc(linl), s_ired);
            if (tid == 0) sh->n_inl_l = nil;
        }
        __syncthreads();
    };

    // ---------------- optimizePose state machine (:332-370) ----------------
    int status = STVO_POSE_OK, path = 0, it0 = 0, it1 = 0;
    if (sh->n_inl_p + sh->n_inl_l >= prm.min_features) {
        it0 = run_mode(prm.mode, prm.max_iters);  // works on DT_ (= sh->DT, a copy of DT0)
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sh->DT1[i] = sh->DT[i];
            t0_is_good(sh, sh->DT1, sh->err_out);
        }
        __syncthreads();
        if (sh->good) {  // :341
            path |= STVO_PATH_STAGE1_GOOD;
            remove_outliers();
            if (sh->n_inl_p + sh->n_inl_l >= prm.min_features) {  // :345 — restart from the INITIAL DT
                path |= STVO_PATH_REFINED;
                if (tid == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sh->DT[i] = sh->DT0[i];
                }
                __syncthreads();
                it1 = run_mode(prm.mode, prm.max_iters_ref);
            } else {
                if (tid == 0) pm::identity4(sh->DT);
                status = STVO_POSE_FEW_INLIERS_AFTER;
                __syncthreads();
            }
        } else {  // :357-362 robust GN on everything, from the initial DT
            path |= STVO_PATH_ROBUST_FALLBACK;
            if (tid == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sh->DT[i] = sh->DT0[i];
            }
            __syncthreads();
            it1 = run_gnr(prm.max_iters_ref);
        }
    } else {
        if (tid == 0) pm::identity4(sh->DT);
        status = STVO_POSE_FEW_INLIERS_BEFORE;
        __syncthreads();
    }

    if (tid == 0) t0_commit(sh, a.results + f, status, path, it0, it1);

    if (a.inl_p_out) {
        const size_t base = (size_t)f * a.max_pts;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * BLOCK;
            if (i < a.max_pts) a.inl_p_out[base + i] = ((pmatched >> k) & 1u) ? (int)((pinl >> k) & 1u) : -1;
        }
    }
    if (a.inl_l_out && a.max_lines > 0) {
        const size_t base = (size_t)f * a.max_lines;
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int i = tid + k * BLOCK;
            if (i < a.max_lines) a.inl_l_out[base + i] = ((lmatched >> k) & 1u) ? (int)((linl >> k) & 1u) : -1;
        }
    }
}

constexpr int POSE_BLOCK = 256;
constexpr int POSE_PPT = STVO_POSE_MAX_POINTS / POSE_BLOCK;  // 8
constexpr int POSE_LPT = STVO_POSE_MAX_LINES / POSE_BLOCK;   // 2

int launch_pose(hipStream_t s, const PoseArgs& a) {
    if (a.B <= 0) return STVO_OK;
    if (a.max_pts > STVO_POSE_MAX_POINTS || a.max_lines > STVO_POSE_MAX_LINES) return STVO_ERR_CAPACITY;
    hipLaunchKernelGGL((pose_kernel<POSE_BLOCK, POSE_PPT, POSE_LPT>), dim3(a.B), dim3(POSE_BLOCK), 0, s, a);
    return STVO_OK;
}

}  // namespace stvo
