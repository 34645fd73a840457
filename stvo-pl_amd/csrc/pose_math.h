// pose_math.h — FP64 building blocks of the pose optimizer, usable from HIP device code and from
// the C++ host mirror (StereoFrameHandler glue: Tfw composition, motion-model check).
//
// Everything here is written for REGISTER residency on gfx950: fixed-size arrays, loops with
// compile-time trip counts and compile-time indices after unrolling (runtime-indexed private
// arrays would be demoted to scratch memory).  Pivoting is expressed as predicated swaps.
//
// What each routine replaces in the reference (/root/reference):
//   pm_expmap_se3 / pm_logmap_se3 / pm_inverse_se3 / pm_adjoint_se3 / pm_unccomp_se3
//                         src/auxiliar.cpp:113-197
//   pm_solve6             Eigen::ColPivHouseholderQR<Matrix6d>::solve + logAbsDeterminant
//                         (call sites src/stereoFrameHandler.cpp:417-418,453-455,507-508,526-527)
//   pm_inverse6           Matrix6d::inverse()            (:429,470,545)
//   pm_eig6               SelfAdjointEigenSolver<Matrix6d>::eigenvalues()  (:294-295,379-380)
//   pm_point_term / pm_line_term   the per-feature bodies of optimizeFunctions[Robust]
//                         (:563-606, :610-684, :785-874, :878-952)
//   pm_line_overlap       StereoFrame::lineSegmentOverlap  src/stereoFrame.cpp:510-616
#pragma once

#include <math.h>

#include "../../include/stvo_types.h"

#if defined(__HIPCC__)
#define PM_HD __host__ __device__ __forceinline__
#else
#define PM_HD inline
#endif

namespace pm {

PM_HD double dmax(double a, double b) { return a > b ? a : b; }  // std::max semantics (NaN in b is dropped)
PM_HD double dmin(double a, double b) { return b < a ? b : a; }  // std::min semantics

PM_HD void identity4(double* T) {
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

PM_HD void mat4_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
            C[i * 4 + j] = s;
        }
}

PM_HD void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

PM_HD void skew3(const double* v, double* S) {
    S[0] = 0.0; S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2]; S[4] = 0.0; S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0]; S[8] = 0.0;
}

// inverse_se3: [R^T, -R^T t]
PM_HD void inverse_se3(const double* T, double* Ti) {
    double o[16];
    identity4(o);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) o[i * 4 + j] = T[j * 4 + i];
        o[i * 4 + 3] = -(T[0 * 4 + i] * T[3] + T[1 * 4 + i] * T[7] + T[2 * 4 + i] * T[11]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Ti[i] = o[i];
}

// expmap_se3, twist = (t, w); below theta = 1e-6 R = I and t is NOT multiplied by V.
// sincos / acos as real calls on the device: inlined, every instance's ~40 polynomial coefficients are hoisted out of the
// optimiser's iteration loop as 64-bit register constants and then spilled (46 doubles of scratch per lane in pose_kernel2)
#if defined(__HIP_DEVICE_COMPILE__)
struct SinCos {
    double s, c;
};
// theta = |w| of a pose increment: a fraction of a radian in practice.  Up to 64 rad: Cody-Waite reduction by pi / 2 (two FMAs) + the
// fdlibm kernel polynomials, ~45 instructions, within an ulp of libm's result (the pose tolerance is 1e-4 rad; the same routine
// restated in oracle/stvo_lsd_oracle.c: orc_sincos_det) — the library call behind it costs ~200 and runs at every iteration of the
// serial section.  Beyond 64 rad (diverged iterations) the library call, for its exact argument reduction.
static __device__ __noinline__ SinCos sincos_call(double x) {
    SinCos r;
    if (!(fabs(x) <= 64.0)) {
        sincos(x, &r.s, &r.c);
        return r;
    }
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17, TWO_OVER_PI = 6.36619772367581382433e-01;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double k = __builtin_rint(x * TWO_OVER_PI);
    double t = __builtin_fma(-k, PIO2_HI, x);
    t = __builtin_fma(-k, PIO2_LO, t);
    const double z = t * t;
    const double sn = t + (z * t) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
    const double cs = 1.0 - (0.5 * z - z * (z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))));
    const int q = (int)k & 3;
    r.s = q == 0 ? sn : (q == 1 ? cs : (q == 2 ? -sn : -cs));
    r.c = q == 0 ? cs : (q == 1 ? -sn : (q == 2 ? -cs : sn));
    return r;
}
static __device__ __noinline__ double acos_call(double x) { return acos(x); }
#endif

PM_HD void expmap_se3(const double* x, double* T) {
    double t0 = x[0], t1 = x[1], t2 = x[2];
    const double w[3] = {x[3], x[4], x[5]};
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (!(theta < 0.000001)) {
        // one reciprocal of theta instead of the reference's eleven divisions by it, and one sincos (<= 1 ulp per term)
        const double itheta = 1.0 / theta;
        double sk[9], s[9], s2[9];
        skew3(w, sk);
#pragma unroll
        for (int i = 0; i < 9; ++i) s[i] = sk[i] * itheta;
        mat3_mul(s, s, s2);
        double sn, cs;
#if defined(__HIP_DEVICE_COMPILE__)
        const SinCos sc = sincos_call(theta);
        sn = sc.s;
        cs = sc.c;
#else
        sincos(theta, &sn, &cs);
#endif
        const double ka = (1.0 - cs) * itheta, kb = (theta - sn) * itheta;
        double V[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double I = (i % 4 == 0) ? 1.0 : 0.0;
            R[i] = I + s[i] * sn + s2[i] * (1.0 - cs);
            V[i] = I + s[i] * ka + s2[i] * kb;
        }
        const double a = V[0] * t0 + V[1] * t1 + V[2] * t2;
        const double b = V[3] * t0 + V[4] * t1 + V[5] * t2;
        const double c = V[6] * t0 + V[7] * t1 + V[8] * t2;
        t0 = a; t1 = b; t2 = c;
    }
    T[0] = R[0]; T[1] = R[1]; T[2] = R[2]; T[3] = t0;
    T[4] = R[3]; T[5] = R[4]; T[6] = R[5]; T[7] = t1;
    T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = t2;
    T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
}

// 3x3 inverse by cofactors (what Eigen uses for fixed 3x3).
PM_HD void inverse3(const double* A, double* Ai) {
    const double c00 = A[4] * A[8] - A[5] * A[7];
    const double c01 = A[5] * A[6] - A[3] * A[8];
    const double c02 = A[3] * A[7] - A[4] * A[6];
    const double id = 1.0 / (A[0] * c00 + A[1] * c01 + A[2] * c02);
    Ai[0] = c00 * id;
    Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id;
    Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = c01 * id;
    Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id;
    Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = c02 * id;
    Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

// logmap_se3: cos from the trace (clamped), sine = sqrt(1-cos^2), V^-1 by 3x3 inverse.
PM_HD void logmap_se3(const double* T, double* x) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double w[3] = {0.0, 0.0, 0.0};
    double cosine = (R[0] + R[4] + R[8] - 1.0) / 2.0;
    if (cosine > 1.0) cosine = 1.0;
    else if (cosine < -1.0) cosine = -1.0;
    double sine = sqrt(1.0 - cosine * cosine);
    if (sine > 1.0) sine = 1.0;
#if defined(__HIP_DEVICE_COMPILE__)
    const double theta = acos_call(cosine);
#else
    const double theta = acos(cosine);
#endif
    if (theta > 0.000001) {
        // skewcoords(theta (R - R^T) / (2 sine)) = (M(2,1), M(0,2), M(1,0))
        w[0] = theta * (R[7] - R[5]) / (2.0 * sine);
        w[1] = theta * (R[2] - R[6]) / (2.0 * sine);
        w[2] = theta * (R[3] - R[1]) / (2.0 * sine);
        double sk[9], s[9], s2[9];
        skew3(w, sk);
#pragma unroll
        for (int i = 0; i < 9; ++i) s[i] = sk[i] / theta;
        mat3_mul(s, s, s2);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double I = (i % 4 == 0) ? 1.0 : 0.0;
            V[i] = I + s[i] * (1.0 - cosine) / theta + s2[i] * (theta - sine) / theta;
        }
    }
    double Vi[9];
    inverse3(V, Vi);
    x[0] = Vi[0] * T[3] + Vi[1] * T[7] + Vi[2] * T[11];
    x[1] = Vi[3] * T[3] + Vi[4] * T[7] + Vi[5] * T[11];
    x[2] = Vi[6] * T[3] + Vi[7] * T[7] + Vi[8] * T[11];
    x[3] = w[0]; x[4] = w[1]; x[5] = w[2];
}

// adjoint_se3 = [R, [t]x R; 0, R]
PM_HD void adjoint_se3(const double* T, double* A) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const double t[3] = {T[3], T[7], T[11]};
    double sk[9], skR[9];
    skew3(t, sk);
    mat3_mul(sk, R, skR);
#pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A[i * 6 + j] = R[i * 3 + j];
            A[i * 6 + j + 3] = skR[i * 3 + j];
            A[(i + 3) * 6 + j + 3] = R[i * 3 + j];
        }
}

// unccomp_se3: cov1 + Ad(T1) covinc Ad(T1)^T
PM_HD void unccomp_se3(const double* T1, const double* cov1, const double* covinc, double* out) {
    double A[36], tmp[36];
    adjoint_se3(T1, A);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += A[i * 6 + k] * covinc[k * 6 + j];
            tmp[i * 6 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += tmp[i * 6 + k] * A[j * 6 + k];
            out[i * 6 + j] = cov1[i * 6 + j] + s;
        }
}

// uncTinv_se3: Ad(T^-1) cov Ad(T^-1)^T            (src/auxiliar.cpp:184-190)
PM_HD void uncTinv_se3(const double* T, const double* cov, double* out) {
    double Ti[16], A[36], tmp[36];
    inverse_se3(T, Ti);
    adjoint_se3(Ti, A);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += A[i * 6 + k] * cov[k * 6 + j];
            tmp[i * 6 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += tmp[i * 6 + k] * A[j * 6 + k];
            out[i * 6 + j] = s;
        }
}

// DT <- DT * inverse_se3(expmap_se3(inc))        (src/stereoFrameHandler.cpp:419)
PM_HD void step_pose(double* DT, const double* inc) {
    double E[16], Ei[16], o[16];
    expmap_se3(inc, E);
    inverse_se3(E, Ei);
    mat4_mul(DT, Ei, o);
#pragma unroll
    for (int i = 0; i < 16; ++i) DT[i] = o[i];
}

// Column-pivoted Householder QR solve of the 6x6 normal equations, fully unrolled.
// Pivot = largest remaining column norm; rank cut at machine precision; free components = 0.
// Returns the numerical rank; *log_abs_det = sum log|R_ii|.
PM_HD int solve6(const double* H, const double* g, double* x, double* log_abs_det) {
    double A[36], c[6], y[6];
    int perm[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = H[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        c[i] = g[i];
        perm[i] = i;
        y[i] = 0.0;
    }
    double maxnorm2 = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) s += A[i * 6 + j] * A[i * 6 + j];
        maxnorm2 = s > maxnorm2 ? s : maxnorm2;
    }
    const double eps = 2.220446049250313e-16;
    const double thr_helper = maxnorm2 * eps * eps / 6.0;
    int rank = 6;
    double lad = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int big = k;
        double bigsq = -1.0;
#pragma unroll
        for (int j = k; j < 6; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = k; i < 6; ++i) s += A[i * 6 + j] * A[i * 6 + j];
            if (s > bigsq) {
                bigsq = s;
                big = j;
            }
        }
        if (rank == 6 && bigsq < thr_helper * (double)(6 - k)) rank = k;
#pragma unroll
        for (int j = k + 1; j < 6; ++j) {  // predicated column swap k <-> big (static indices)
            const bool sw = (big == j);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double a = A[i * 6 + k], b = A[i * 6 + j];
                A[i * 6 + k] = sw ? b : a;
                A[i * 6 + j] = sw ? a : b;
            }
            const int pa = perm[k], pb = perm[j];
            perm[k] = sw ? pb : pa;
            perm[j] = sw ? pa : pb;
        }
        const double c0 = A[k * 6 + k];
        double tail = 0.0;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) tail += A[i * 6 + k] * A[i * 6 + k];
        double beta, tau;
        if (tail <= 2.2250738585072014e-308) {
            tau = 0.0;
            beta = c0;
#pragma unroll
            for (int i = k + 1; i < 6; ++i) A[i * 6 + k] = 0.0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            const double den = c0 - beta;
#pragma unroll
            for (int i = k + 1; i < 6; ++i) A[i * 6 + k] /= den;
            tau = (beta - c0) / beta;
        }
        A[k * 6 + k] = beta;
        lad += log(fabs(beta));
#pragma unroll
        for (int j = k + 1; j < 6; ++j) {
            double s = A[k * 6 + j];
#pragma unroll
            for (int i = k + 1; i < 6; ++i) s += A[i * 6 + k] * A[i * 6 + j];
            s *= tau;
            A[k * 6 + j] -= s;
#pragma unroll
            for (int i = k + 1; i < 6; ++i) A[i * 6 + j] -= s * A[i * 6 + k];
        }
        {
            double s = c[k];
#pragma unroll
            for (int i = k + 1; i < 6; ++i) s += A[i * 6 + k] * c[i];
            s *= tau;
            c[k] -= s;
#pragma unroll
            for (int i = k + 1; i < 6; ++i) c[i] -= s * A[i * 6 + k];
        }
    }
    if (log_abs_det) *log_abs_det = lad;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = c[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) s -= A[i * 6 + j] * y[j];  // y[j] == 0 for j >= rank
        y[i] = (i < rank) ? s / A[i * 6 + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (perm[i] == j) x[j] = y[i];
    return rank;
}

// 6x6 inverse by LU with partial pivoting (rows swapped by predicated moves), fully unrolled.
PM_HD void inverse6(const double* Ain, double* Ai) {
    double A[36], B[36];  // B accumulates P * I, then is solved in place
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        A[i] = Ain[i];
        B[i] = (i % 7 == 0) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double best = fabs(A[k * 6 + k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double v = fabs(A[i * 6 + k]);
            if (v > best) {
                best = v;
                p = i;
            }
        }
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const bool sw = (p == i);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double a = A[k * 6 + j], b = A[i * 6 + j];
                A[k * 6 + j] = sw ? b : a;
                A[i * 6 + j] = sw ? a : b;
                const double c = B[k * 6 + j], d = B[i * 6 + j];
                B[k * 6 + j] = sw ? d : c;
                B[i * 6 + j] = sw ? c : d;
            }
        }
        const double piv = A[k * 6 + k];
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double l = A[i * 6 + k] / piv;
            A[i * 6 + k] = l;
#pragma unroll
            for (int j = k + 1; j < 6; ++j) A[i * 6 + j] -= l * A[k * 6 + j];
#pragma unroll
            for (int j = 0; j < 6; ++j) B[i * 6 + j] -= l * B[k * 6 + j];  // forward substitution on the fly
        }
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
#pragma unroll
        for (int col = 0; col < 6; ++col) {
            double s = B[i * 6 + col];
#pragma unroll
            for (int j = i + 1; j < 6; ++j) s -= A[i * 6 + j] * B[j * 6 + col];
            B[i * 6 + col] = s / A[i * 6 + i];
        }
    }
#pragma unroll
    for (int i = 0; i < 36; ++i) Ai[i] = B[i];
}

// ---- the two pivoted routines above on MEMORY operands with run-time loops (round 6) --------------------------------------------
// solve6 / inverse6 keep a 6x6 matrix (and a second one) in registers with every index a compile-time constant and every swap a
// predicated move: ~150 live VGPRs, which the batch pose kernel pays as spills around its serial sections although the routines
// only run when the LDL^T fast path refuses a matrix (rank-deficient geometry).  These forms do the SAME operations in the SAME
// order — tests/test_pose_math_host.py holds them to bit-identical results — on arrays the caller provides (LDS on the device):
// a handful of registers, slow, rare.
// A [36] and c [6] are destroyed; ws_y [6] and ws_perm [6] are scratch.  Returns the numerical rank.
PM_HD int solve6_mem(double* A, double* c, double* ws_y, int* perm, double* x, double* log_abs_det) {
    double* y = ws_y;
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
        perm[i] = i;
        y[i] = 0.0;
    }
    double maxnorm2 = 0.0;
#pragma unroll 1
    for (int j = 0; j < 6; ++j) {
        double s = 0.0;
#pragma unroll 1
        for (int i = 0; i < 6; ++i) s += A[i * 6 + j] * A[i * 6 + j];
        maxnorm2 = s > maxnorm2 ? s : maxnorm2;
    }
    const double eps = 2.220446049250313e-16;
    const double thr_helper = maxnorm2 * eps * eps / 6.0;
    int rank = 6;
    double lad = 0.0;
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
        int big = k;
        double bigsq = -1.0;
#pragma unroll 1
        for (int j = k; j < 6; ++j) {
            double s = 0.0;
#pragma unroll 1
            for (int i = k; i < 6; ++i) s += A[i * 6 + j] * A[i * 6 + j];
            if (s > bigsq) {
                bigsq = s;
                big = j;
            }
        }
        if (rank == 6 && bigsq < thr_helper * (double)(6 - k)) rank = k;
        if (big != k) {  // column swap k <-> big (the predicated form swaps with exactly one j > k)
#pragma unroll 1
            for (int i = 0; i < 6; ++i) {
                const double a = A[i * 6 + k];
                A[i * 6 + k] = A[i * 6 + big];
                A[i * 6 + big] = a;
            }
            const int pa = perm[k];
            perm[k] = perm[big];
            perm[big] = pa;
        }
        const double c0 = A[k * 6 + k];
        double tail = 0.0;
#pragma unroll 1
        for (int i = k + 1; i < 6; ++i) tail += A[i * 6 + k] * A[i * 6 + k];
        double beta, tau;
        if (tail <= 2.2250738585072014e-308) {
            tau = 0.0;
            beta = c0;
#pragma unroll 1
            for (int i = k + 1; i < 6; ++i) A[i * 6 + k] = 0.0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            const double den = c0 - beta;
#pragma unroll 1
            for (int i = k + 1; i < 6; ++i) A[i * 6 + k] /= den;
            tau = (beta - c0) / beta;
        }
        A[k * 6 + k] = beta;
        lad += log(fabs(beta));
#pragma unroll 1
        for (int j = k + 1; j < 6; ++j) {
            double s = A[k * 6 + j];
#pragma unroll 1
            for (int i = k + 1; i < 6; ++i) s += A[i * 6 + k] * A[i * 6 + j];
            s *= tau;
            A[k * 6 + j] -= s;
#pragma unroll 1
            for (int i = k + 1; i < 6; ++i) A[i * 6 + j] -= s * A[i * 6 + k];
        }
        {
            double s = c[k];
#pragma unroll 1
            for (int i = k + 1; i < 6; ++i) s += A[i * 6 + k] * c[i];
            s *= tau;
            c[k] -= s;
#pragma unroll 1
            for (int i = k + 1; i < 6; ++i) c[i] -= s * A[i * 6 + k];
        }
    }
    if (log_abs_det) *log_abs_det = lad;
#pragma unroll 1
    for (int i = 5; i >= 0; --i) {
        double s = c[i];
#pragma unroll 1
        for (int j = i + 1; j < 6; ++j) s -= A[i * 6 + j] * y[j];  // y[j] == 0 for j >= rank
        y[i] = (i < rank) ? s / A[i * 6 + i] : 0.0;
    }
#pragma unroll 1
    for (int i = 0; i < 6; ++i) x[perm[i]] = y[i];  // (perm is a permutation: every x[j] is written exactly once)
    return rank;
}

// A [36] is destroyed (L \ U in place), B [36] receives the inverse.
PM_HD void inverse6_mem(double* A, double* B) {
#pragma unroll 1
    for (int i = 0; i < 36; ++i) B[i] = (i % 7 == 0) ? 1.0 : 0.0;
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double best = fabs(A[k * 6 + k]);
#pragma unroll 1
        for (int i = k + 1; i < 6; ++i) {
            const double v = fabs(A[i * 6 + k]);
            if (v > best) {
                best = v;
                p = i;
            }
        }
        if (p != k) {
#pragma unroll 1
            for (int j = 0; j < 6; ++j) {
                const double a = A[k * 6 + j];
                A[k * 6 + j] = A[p * 6 + j];
                A[p * 6 + j] = a;
                const double c = B[k * 6 + j];
                B[k * 6 + j] = B[p * 6 + j];
                B[p * 6 + j] = c;
            }
        }
        const double piv = A[k * 6 + k];
#pragma unroll 1
        for (int i = k + 1; i < 6; ++i) {
            const double l = A[i * 6 + k] / piv;
            A[i * 6 + k] = l;
#pragma unroll 1
            for (int j = k + 1; j < 6; ++j) A[i * 6 + j] -= l * A[k * 6 + j];
#pragma unroll 1
            for (int j = 0; j < 6; ++j) B[i * 6 + j] -= l * B[k * 6 + j];
        }
    }
#pragma unroll 1
    for (int i = 5; i >= 0; --i) {
#pragma unroll 1
        for (int col = 0; col < 6; ++col) {
            double s = B[i * 6 + col];
#pragma unroll 1
            for (int j = i + 1; j < 6; ++j) s -= A[i * 6 + j] * B[j * 6 + col];
            B[i * 6 + col] = s / A[i * 6 + i];
        }
    }
}

// Determinant by LU with partial pivoting (what Matrix6d::determinant() does for a 6x6: PartialPivLU).
PM_HD double det6(const double* Ain) {
    double A[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = Ain[i];
    double det = 1.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double best = fabs(A[k * 6 + k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double v = fabs(A[i * 6 + k]);
            if (v > best) {
                best = v;
                p = i;
            }
        }
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const bool sw = (p == i);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double a = A[k * 6 + j], b = A[i * 6 + j];
                A[k * 6 + j] = sw ? b : a;
                A[i * 6 + j] = sw ? a : b;
            }
        }
        if (p != k) det = -det;
        const double piv = A[k * 6 + k];
        det *= piv;
        if (piv == 0.0) return 0.0;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double l = A[i * 6 + k] / piv;
#pragma unroll
            for (int j = k + 1; j < 6; ++j) A[i * 6 + j] -= l * A[k * 6 + j];
        }
    }
    return det;
}

// Ascending eigenvalues of the symmetric matrix given by the LOWER triangle of Ain
// (cyclic Jacobi, 15 rotations per sweep unrolled; the sweep loop is dynamic).
PM_HD void eig6(const double* Ain, double* w) {
    double A[36];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) A[i * 6 + j] = (i >= j) ? Ain[i * 6 + j] : Ain[j * 6 + i];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (i != j) off += A[i * 6 + j] * A[i * 6 + j];
                else diag += A[i * 6 + j] * A[i * 6 + j];
            }
        if (!(off > 1e-32 * diag) || !(off > 0.0)) break;
#pragma unroll
        for (int p = 0; p < 5; ++p)
#pragma unroll
            for (int q = p + 1; q < 6; ++q) {
                const double apq = A[p * 6 + q];
                if (apq != 0.0) {
                    const double theta = (A[q * 6 + q] - A[p * 6 + p]) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double akp = A[k * 6 + p], akq = A[k * 6 + q];
                        A[k * 6 + p] = cs * akp - sn * akq;
                        A[k * 6 + q] = sn * akp + cs * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double apk = A[p * 6 + k], aqk = A[q * 6 + k];
                        A[p * 6 + k] = cs * apk - sn * aqk;
                        A[q * 6 + k] = sn * apk + cs * aqk;
                    }
                }
            }
    }
    double v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = A[i * 7];
    // ascending sorting network (odd-even transposition, 6 rounds); comparisons false on NaN
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = (r & 1); i + 1 < 6; i += 2) {
            const double a = v[i], b = v[i + 1];
            const bool sw = a > b;
            v[i] = sw ? b : a;
            v[i + 1] = sw ? a : b;
        }
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = v[i];
}

// ---- fast paths for the (normally) symmetric positive definite 6x6 systems of the optimizer ----------
// The reference solves H x = g with Eigen's ColPivHouseholderQR, inverts H with PartialPivLU and takes
// eigenvalues with SelfAdjointEigenSolver (tridiagonalisation + implicit QL/QR).  H = sum w J J^T is symmetric
// positive definite unless the geometry is degenerate, and for an SPD matrix an unpivoted LDL^T factorisation
// is backward stable: it gives the same x and H^-1 up to rounding (~cond * eps) at ~1/10 of the arithmetic
// (no column norms, no predicated swaps, 6 divisions instead of ~27).  The routines below return false when a
// pivot is not comfortably positive (<= 1e-10 of the largest diagonal entry, or NaN); callers then run the
// pivoted routines above, so degenerate / rank-deficient inputs behave exactly as before.

// LDL^T of the symmetric matrix given by the LOWER triangle of A.  L: unit lower triangular (strict lower part
// written, [i*6+j], j < i), d: pivots, dinv: their reciprocals.
PM_HD bool ldl6(const double* A, double* L, double* d, double* dinv) {
    double maxd = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) maxd = fabs(A[i * 7]) > maxd ? fabs(A[i * 7]) : maxd;
    const double tol = 1e-10 * maxd;
    bool ok = maxd > 0.0 && maxd < 1.0e300;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double w[6];  // w[j] = L[k][j] * d[j]
        double dk = A[k * 7];
#pragma unroll
        for (int j = 0; j < k; ++j) {
            w[j] = L[k * 6 + j] * d[j];
            dk -= L[k * 6 + j] * w[j];
        }
        d[k] = dk;
        ok = ok && (dk > tol);
        const double inv = 1.0 / dk;
        dinv[k] = inv;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            double t = A[i * 6 + k];
#pragma unroll
            for (int j = 0; j < k; ++j) t -= L[i * 6 + j] * w[j];
            L[i * 6 + k] = t * inv;
        }
    }
    return ok;
}

// x = H^-1 g and log|det H| through LDL^T.  false => not certified SPD, outputs undefined (use solve6).
PM_HD bool solve6_spd(const double* H, const double* g, double* x, double* log_abs_det) {
    double L[36], d[6], dinv[6], y[6];
    if (!ldl6(H, L, d, dinv)) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double t = g[i];
#pragma unroll
        for (int j = 0; j < i; ++j) t -= L[i * 6 + j] * y[j];
        y[i] = t;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double t = y[i] * dinv[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) t -= L[j * 6 + i] * x[j];
        x[i] = t;
    }
    if (log_abs_det) *log_abs_det = log(d[0] * d[1] * d[2]) + log(d[3] * d[4] * d[5]);
    return true;
}

// Ai = A^-1 (symmetric) through LDL^T: A^-1 = M^T D^-1 M with M = L^-1.  false => use inverse6.
PM_HD bool inverse6_spd(const double* A, double* Ai) {
    double L[36], d[6], dinv[6], M[36];
    if (!ldl6(A, L, d, dinv)) return false;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {  // column j of M = L^-1 below the unit diagonal
            double t = L[i * 6 + j];
#pragma unroll
            for (int k = j + 1; k < i; ++k) t += L[i * 6 + k] * M[k * 6 + j];
            M[i * 6 + j] = -t;
        }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            // sum over k >= i of M[k][i] * dinv[k] * M[k][j]   (M[i][i] = 1)
            double t = (i == j) ? dinv[i] : dinv[i] * M[i * 6 + j];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) t += M[k * 6 + i] * dinv[k] * M[k * 6 + j];
            Ai[i * 6 + j] = t;
            Ai[j * 6 + i] = t;
        }
    return true;
}

// Ascending eigenvalues of the symmetric matrix given by the LOWER triangle of Ain: Householder
// tridiagonalisation followed by implicit-shift QL iterations on the tridiagonal matrix — the scheme of
// Eigen's SelfAdjointEigenSolver (and of EISPACK tred1 / tql1), eigenvalues only.  Every array index is a
// compile-time constant after unrolling (the deflation point m is tracked by predication), so d[] and e[]
// stay in registers.  ~5x fewer operations than the cyclic Jacobi eig6 above.
PM_HD void eig6_ql(const double* Ain, double* w) {
    double a[36], d[6], e[6];
    double amax = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) amax = fabs(Ain[i * 6 + j]) > amax ? fabs(Ain[i * 6 + j]) : amax;
    if (!(amax > 0.0) || !(amax < 1.0e300)) {  // zero matrix, NaN or Inf: eigenvalues of the diagonal as is
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = Ain[i * 7];
    } else {
        int ex;
        (void)frexp(amax, &ex);  // exact power-of-two scaling keeps the squares in range
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) a[i * 6 + j] = (j <= i) ? ldexp(Ain[i * 6 + j], -ex) : 0.0;
        // ---- tridiagonalisation (rows 5 .. 1) ----
#pragma unroll
        for (int i = 5; i >= 1; --i) {
            const int l = i - 1;
            double h = 0.0;
            if (l > 0) {
                double scale = 0.0;
#pragma unroll
                for (int k = 0; k <= l; ++k) scale += fabs(a[i * 6 + k]);
                if (scale == 0.0) {
                    e[i] = a[i * 6 + l];
                } else {
                    const double iscale = 1.0 / scale;
#pragma unroll
                    for (int k = 0; k <= l; ++k) {
                        a[i * 6 + k] *= iscale;
                        h += a[i * 6 + k] * a[i * 6 + k];
                    }
                    double f = a[i * 6 + l];
                    double g = f >= 0.0 ? -sqrt(h) : sqrt(h);
                    e[i] = scale * g;
                    h -= f * g;
                    a[i * 6 + l] = f - g;
                    f = 0.0;
                    const double ih = 1.0 / h;
#pragma unroll
                    for (int j = 0; j <= l; ++j) {
                        g = 0.0;
#pragma unroll
                        for (int k = 0; k <= j; ++k) g += a[j * 6 + k] * a[i * 6 + k];
#pragma unroll
                        for (int k = j + 1; k <= l; ++k) g += a[k * 6 + j] * a[i * 6 + k];
                        e[j] = g * ih;
                        f += e[j] * a[i * 6 + j];
                    }
                    const double hh = f / (h + h);
#pragma unroll
                    for (int j = 0; j <= l; ++j) {
                        f = a[i * 6 + j];
                        g = e[j] - hh * f;
                        e[j] = g;
#pragma unroll
                        for (int k = 0; k <= j; ++k) a[j * 6 + k] -= f * e[k] + g * a[i * 6 + k];
                    }
                }
            } else {
                e[i] = a[i * 6 + l];
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = a[i * 7];
#pragma unroll
        for (int i = 1; i < 6; ++i) e[i - 1] = e[i];
        e[5] = 0.0;
        // ---- implicit QL on (d, e) ----
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            for (int iter = 0; iter < 60; ++iter) {
                int m = 5;  // first m >= l whose sub-diagonal e[m] is negligible (e[5] == 0)
#pragma unroll
                for (int mm = 4; mm >= l; --mm) {
                    const double dd = fabs(d[mm]) + fabs(d[mm + 1]);
                    if (fabs(e[mm]) <= 2.220446049250313e-16 * dd) m = mm;
                }
                if (m == l) break;
                double dm = d[5];
#pragma unroll
                for (int mm = 4; mm >= l; --mm) dm = (m == mm) ? d[mm] : dm;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = sqrt(g * g + 1.0);
                g = dm - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
                double sn = 1.0, cs = 1.0, p = 0.0;
                bool underflow = false;
#pragma unroll
                for (int i = 4; i >= l; --i) {
                    if (i < m && !underflow) {
                        double f = sn * e[i];
                        const double b = cs * e[i];
                        r = sqrt(f * f + g * g);
                        e[i + 1] = r;
                        if (r == 0.0) {
                            d[i + 1] -= p;
                            underflow = true;
                        } else {
                            sn = f / r;
                            cs = g / r;
                            g = d[i + 1] - p;
                            r = (d[i] - g) * sn + 2.0 * cs * b;
                            p = sn * r;
                            d[i + 1] = g + p;
                            g = cs * r - b;
                        }
                    }
                }
#pragma unroll
                for (int mm = 5; mm >= l; --mm)
                    if (m == mm) e[mm] = 0.0;
                if (!underflow) {
                    d[l] -= p;
                    e[l] = g;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = ldexp(d[i], ex);
    }
    // ascending sorting network (odd-even transposition, 6 rounds); comparisons false on NaN
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = (r & 1); i + 1 < 6; i += 2) {
            const double x = w[i], y = w[i + 1];
            const bool sw = x > y;
            w[i] = sw ? y : x;
            w[i + 1] = sw ? x : y;
        }
}

// Cheap certificate for the eigenvalue part of isGoodSolution (lambda_min >= 0 and lambda_max <= 1) that
// avoids the eigen-decomposition in the common case: for the symmetric matrix S given by the LOWER
// triangle of C,  lambda_max <= ||S||_inf,  and S is positive definite iff its LDL^T pivots are positive.
// Returns +1 (certified good: ||S||_inf <= 1 and every pivot comfortably positive), or 0 (undecided: the
// caller falls back to eig6, which is what the reference computes).  Never returns a wrong "good".
PM_HD int spd_unit_certificate(const double* C) {
    double S[36];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) S[i * 6 + j] = (i >= j) ? C[i * 6 + j] : C[j * 6 + i];
    double rmax = 0.0, dmax_ = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) r += fabs(S[i * 6 + j]);
        rmax = r > rmax ? r : rmax;
        dmax_ = S[i * 7] > dmax_ ? S[i * 7] : dmax_;
    }
    if (!(rmax <= 1.0)) return 0;  // also catches NaN
    const double tol = 1e-10 * dmax_;
    bool ok = dmax_ > 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {  // in-place LDL^T on the lower triangle
        const double piv = S[k * 7];
        ok = ok && (piv > tol);
        const double inv = 1.0 / piv;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double l = S[i * 6 + k] * inv;
#pragma unroll
            for (int j = k + 1; j <= i; ++j) S[i * 6 + j] -= l * S[j * 6 + k];
        }
    }
    return ok ? 1 : 0;
}

PM_HD bool all_finite16(const double* T) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 16; ++i) ok = ok && isfinite(T[i]);
    return ok;
}

PM_HD bool is_identity16(const double* T) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 16; ++i) ok = ok && (T[i] == ((i % 5 == 0) ? 1.0 : 0.0));
    return ok;
}

// isGoodSolution (src/stereoFrameHandler.cpp:292-305) given precomputed ascending eigenvalues.
PM_HD bool is_good_solution(const double* DT, const double* eig, double err) {
    return !(eig[0] < 0.0 || eig[5] > 1.0 || err < 0.0 || err > 1.0 || !all_finite16(DT));
}

PM_HD double overlap_from_lambdas(double ls, double le) {
    const double lmin = dmin(ls, le), lmax = dmax(ls, le);
    if (lmin < 0.0 && lmax > 1.0) return 1.0;
    if (lmax < 0.0 || lmin > 1.0) return 0.0;
    if (lmin < 0.0) return lmax;
    if (lmax > 1.0) return 1.0 - lmin;
    return lmax - lmin;
}

// StereoFrame::lineSegmentOverlap: fraction of the observed segment (so,eo) covered by the
// projection of (sp,ep) onto it; vertical / horizontal / general branches with 1 px thresholds.
PM_HD double line_overlap(double sox, double soy, double eox, double eoy, double spx, double spy, double epx,
                          double epy) {
    const double lx = eox - sox, ly = eoy - soy;
    if (fabs(sox - eox) < 1.0) return overlap_from_lambdas((spy - soy) / ly, (epy - soy) / ly);
    if (fabs(soy - eoy) < 1.0) return overlap_from_lambdas((spx - sox) / lx, (epx - sox) / lx);
    const double a = soy - eoy, b = eox - sox, c = sox * eoy - eox * soy;
    const double lxy = 1.0 / (a * a + b * b);
    const double fsx = (b * (b * spx - a * spy) - a * c) * lxy;
    const double fex = (b * (b * epx - a * epy) - a * c) * lxy;
    return overlap_from_lambdas((fsx - sox) / lx, (fex - sox) / lx);
}

// Stereo association of a line pair, shared by line_tail_kernel (seq_pipeline.hip) and the host mirror (host/stereoFrame.cpp).
// Fraction of the left segment's row span [min(y_s, y_e), max(y_s, y_e)] that the right segment's rows cover
// (StereoFrame::lineSegmentOverlapStereo, src/stereoFrame.cpp:473-508): 1 for a left segment flatter than horiz_th, and the
// reference normalises by  max(left rows) - min(right rows)  (not by the left span) and clips at 1.
PM_HD double stereo_row_overlap(double yl_s, double yl_e, double yr_s, double yr_e, double horiz_th) {
    if (!(fabs(yl_e - yl_s) > horiz_th)) return 1.0;
    const double l_top = dmin(yl_s, yl_e), l_bot = dmax(yl_s, yl_e);
    const double r_top = dmin(yr_s, yr_e), r_bot = dmax(yr_s, yr_e);
    double cover;
    if (r_bot < l_top || r_top > l_bot)
        cover = 0.0;                                   // disjoint row ranges
    else if (r_bot > l_bot && r_top < l_top)
        cover = l_bot - l_top;                         // the right segment spans the whole left one
    else
        cover = dmin(l_bot, r_bot) - dmax(l_top, r_top);
    const double denom = l_bot - r_top;
    cover = denom > (double)0.01f ? cover / denom : 0.0;
    return cover > 1.0 ? 1.0 : cover;
}
// End-point disparities of a stereo line; both become -1 when their ratio is below min_ratio
// (StereoFrame::filterLineSegmentDisparity, src/stereoFrame.cpp:405-415).
PM_HD void stereo_line_disparities(double xl_s, double xl_e, double xr_s, double xr_e, double min_ratio, double* disp_s,
                                   double* disp_e) {
    const double ds = xl_s - xr_s, de = xl_e - xr_e;
    const bool consistent = !(dmin(ds, de) / dmax(ds, de) < min_ratio);
    *disp_s = consistent ? ds : -1.0;
    *disp_e = consistent ? de : -1.0;
}

struct Cam5 {
    double fx, fy, cx, cy;
};

// P_ = R P + t ; projection (src/pinholeStereoCamera.cpp:231-237)
PM_HD void transform_project(const double* DT, double X, double Y, double Z, const Cam5& cam, double* Pc,
                             double* uv) {
    Pc[0] = DT[0] * X + DT[1] * Y + DT[2] * Z + DT[3];
    Pc[1] = DT[4] * X + DT[5] * Y + DT[6] * Z + DT[7];
    Pc[2] = DT[8] * X + DT[9] * Y + DT[10] * Z + DT[11];
    // one reciprocal instead of the reference's two divisions (<= 1 ulp apart; an FP64 division is ~13 dependent
    // instructions on gfx950 and this sits on the critical path of every feature of every evaluation)
    const double iz = 1.0 / Pc[2];
    uv[0] = cam.cx + cam.fx * Pc[0] * iz;
    uv[1] = cam.cy + cam.fy * Pc[1] * iz;
}

// 1x6 gradient of the scalar residual (translation first, rotation last; only fx appears).
PM_HD void grad6(const double* Pc, double dx, double dy, double fx, double homog_th, double* J) {
    const double gx = Pc[0], gy = Pc[1], gz = Pc[2];
    const double gz2 = gz * gz;
    const double fgz2 = fx / dmax(homog_th, gz2);
    J[0] = +fgz2 * dx * gz;
    J[1] = +fgz2 * dy * gz;
    J[2] = -fgz2 * (gx * dx + gy * dy);
    J[3] = -fgz2 * (gx * gy * dx + gy * gy * dy + gz * gz * dy);
    J[4] = +fgz2 * (gx * gx * dx + gz * gz * dx + gx * gy * dy);
    J[5] = +fgz2 * (gx * gz * dy - gy * gz * dx);
}

// acc[0..20] upper triangle of H (row-major i<=j), acc[21..26] g, acc[27] e
PM_HD void accumulate28(double* acc, const double* J, double r, double w) {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) acc[k++] += J[i] * J[j] * w;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] += J[i] * r * w;
    acc[27] += r * r * w;
}

// Reprojection residual norm of one point.
PM_HD double point_residual(const double* DT, const Cam5& cam, double X, double Y, double Z, double ox, double oy) {
    double Pc[3], uv[2];
    transform_project(DT, X, Y, Z, cam, Pc, uv);
    const double dx = uv[0] - ox, dy = uv[1] - oy;
    return sqrt(dx * dx + dy * dy);
}

// ---- compact stereo points of the device-resident pipeline: {u, v, disparity as floats, pyramid level} ----
// The reference builds a PointFeature from exactly these (src/stereoFrame.cpp:152-167: float key-point coordinates, a float
// difference, backProjection of the three, sigma2 from the octave), so P and sigma2 are functions of 13 bytes; the pipeline
// stores those and the pose kernels recompute.  Operations and their order are the reference's (src/pinholeStereoCamera.cpp:
// 221-229, src/stereoFeatures.cpp:41-47); none of them can contract into an FMA.
PM_HD void back_project(double b, double fx, double cx, double cy, double u, double v, double disp, double* P) {
    const double bd = b / disp;
    P[0] = bd * (u - cx);
    P[1] = bd * (v - cy);
    P[2] = bd * fx;
}
PM_HD double level_sigma2(int level, double scale) {
    double sg = 1.0;
    for (int q = 0; q < level; ++q) sg *= scale;
    return 1.0 / (sg * sg);
}

// One point of optimizeFunctions (robust == false: r = |e| sqrt(sigma2), w = Cauchy(r);
// robust == true: r = |e|, w = Cauchy(r / s_p)).
PM_HD void point_term(double* acc, const double* DT, const Cam5& cam, double homog_th, double X, double Y, double Z,
                      double ox, double oy, double sigma2, bool robust, double s_p) {
    double Pc[3], uv[2], J[6];
    transform_project(DT, X, Y, Z, cam, Pc, uv);
    const double dx = uv[0] - ox, dy = uv[1] - oy;
    const double nrm = sqrt(dx * dx + dy * dy);
    grad6(Pc, dx, dy, cam.fx, homog_th, J);
    const double iden = 1.0 / dmax(homog_th, nrm);  // J / max(homogTh, |e|) as one reciprocal + 6 products
#pragma unroll
    for (int i = 0; i < 6; ++i) J[i] = J[i] * iden;
    double r, w;
    if (!robust) {
        r = nrm * sqrt(sigma2);
        w = 1.0 / (1.0 + r * r);
    } else {
        r = nrm;
        const double xx = r / s_p;
        w = 1.0 / (1.0 + xx * xx);
    }
    accumulate28(acc, J, r, w);
}

// Reciprocal / square root for the per-feature hot loop of pose_kernel2.hip: hardware seed (v_rcp_f64 / v_rsq_f64) + two
// Newton steps, <= 1 ulp from the correctly rounded value for the normal, positive arguments that occur there (depths,
// residual norms, 1 + r^2), without the scaling / special-case instructions of the IEEE expansions (5 / 8 instead of
// ~12 / ~15 instructions; the kernel is bound by FP64 issue).  Host builds use the exact operations.
PM_HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
#else
    return 1.0 / x;
#endif
}
PM_HD double fast_sqrt(double x) {  // x >= 0 (NaN propagates)
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return x > 0.0 ? g : x;  // sqrt(0) = 0 (the seed is inf there)
#else
    return sqrt(x);
#endif
}

// The same sums with the weight folded into one factor first: Jw = J w (6 products), then one FMA per entry — 35
// instructions instead of 56 (pose_kernel2.hip; the products differ from accumulate28's by at most one rounding each).
PM_HD void accumulate28w(double* acc, const double* J, double r, double w) {
    double Jw[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Jw[i] = J[i] * w;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) acc[k++] += Jw[i] * J[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] += Jw[i] * r;
    acc[27] += (r * r) * w;
}

// ---- the per-feature term of the kernels, written ONCE for one record (T = double) and for two records side by side
// (T = d2, device only): every operation of the pair form is the scalar operation on both components, adjacent in the instruction
// stream.  A term is a chain of ~60 dependent FP64 instructions and a wave retires one of those every ~8 cycles: the pair form
// fills the gaps with the other record (the compiler does NOT interleave two inlined scalar terms on its own — it schedules them
// one after the other).  Same expression trees, hence the same contractions and the same values as the scalar form.
#if defined(__HIPCC__)
typedef double d2 __attribute__((ext_vector_type(2)));
PM_HD d2 t_rcp(d2 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    d2 y = {__builtin_amdgcn_rcp(x.x), __builtin_amdgcn_rcp(x.y)};
    d2 e = {fma(-x.x, y.x, 1.0), fma(-x.y, y.y, 1.0)};
    y = d2{fma(y.x, e.x, y.x), fma(y.y, e.y, y.y)};
    e = d2{fma(-x.x, y.x, 1.0), fma(-x.y, y.y, 1.0)};
    return d2{fma(y.x, e.x, y.x), fma(y.y, e.y, y.y)};
#else
    return d2{fast_rcp(x.x), fast_rcp(x.y)};  // (the host pass only parses the kernels)
#endif
}
PM_HD d2 t_sqrt(d2 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const d2 y = {__builtin_amdgcn_rsq(x.x), __builtin_amdgcn_rsq(x.y)};
    d2 g = x * y, h = 0.5 * y;
    const d2 r = {fma(-h.x, g.x, 0.5), fma(-h.y, g.y, 0.5)};
    g = d2{fma(g.x, r.x, g.x), fma(g.y, r.y, g.y)};
    h = d2{fma(h.x, r.x, h.x), fma(h.y, r.y, h.y)};
    const d2 d = {fma(-g.x, g.x, x.x), fma(-g.y, g.y, x.y)};
    g = d2{fma(d.x, h.x, g.x), fma(d.y, h.y, g.y)};
    return d2{x.x > 0.0 ? g.x : x.x, x.y > 0.0 ? g.y : x.y};
#else
    return d2{fast_sqrt(x.x), fast_sqrt(x.y)};
#endif
}
PM_HD d2 t_sel_gt(d2 a, double b, d2 x, double y) { return d2{a.x > b ? x.x : y, a.y > b ? x.y : y}; }  // a > b ? x : y
PM_HD void t_acc(double& acc, d2 a, d2 b) {  // acc += a b, component 0 first: the order of two consecutive scalar terms
    acc += a.x * b.x;
    acc += a.y * b.y;
}
#endif
PM_HD double t_rcp(double x) { return fast_rcp(x); }
PM_HD double t_sqrt(double x) { return fast_sqrt(x); }
PM_HD double t_sel_gt(double a, double b, double x, double y) { return a > b ? x : y; }
PM_HD void t_acc(double& acc, double a, double b) { acc += a * b; }

// One point of optimizeFunctions[Robust] (or two).  Against the reference's formulas (point_term above, src/stereoFrameHandler.cpp:
// 563-606): one reciprocal per division by a common denominator; fx / max(homogTh, gz^2) from the reciprocal depth the projection
// already has (iz^2 when gz^2 exceeds the threshold — every point in front of the camera — 1 / homogTh otherwise, a select);
// the gradient with the common sub-expression t = gx dx + gy dy factored out and the scale a = fgz2 / max(homogTh, |e|) applied
// once (J3 = -a (gy t + gz^2 dy), J4 = a (gx t + gz^2 dx) are the reference's terms re-associated: 15 operations instead of 32);
// the sums weight-first (Jw = J w, one FMA per entry).  Same values up to a few roundings; the kernels are bound by FP64 issue.
// sqrt_sigma2 comes with the record; robust == true takes the RECIPROCAL of the MAD scale (one division per evaluation); the two
// variants differ by two selects of block-uniform values, not by a branch: r = |e| sqrt(sigma2), w = Cauchy(r) or r = |e|,
// w = Cauchy(r / s_p).  wmask (1 or 0) multiplies the weight: 0 makes a (finite) record an exact no-op.  inv_homog = 1 / homogTh.
template <typename T>
PM_HD void point_term_t(double* acc, const double* DT, const Cam5& cam, double homog_th, double inv_homog, T X, T Y, T Z, T ox, T oy,
                        T sqrt_sigma2, bool robust, double inv_s_p, T wmask) {
    const T gx = DT[0] * X + DT[1] * Y + DT[2] * Z + DT[3];
    const T gy = DT[4] * X + DT[5] * Y + DT[6] * Z + DT[7];
    const T gz = DT[8] * X + DT[9] * Y + DT[10] * Z + DT[11];
    const T iz = t_rcp(gz);
    const T dx = (cam.cx + cam.fx * gx * iz) - ox, dy = (cam.cy + cam.fy * gy * iz) - oy;
    const T nrm = t_sqrt(dx * dx + dy * dy);
    const T gz2 = gz * gz;
    const T a = (cam.fx * t_sel_gt(gz2, homog_th, iz * iz, inv_homog)) * t_sel_gt(nrm, homog_th, t_rcp(nrm), inv_homog);
    const T t = gx * dx + gy * dy;
    const T ag = a * gz;
    T J[6];
    J[0] = ag * dx;
    J[1] = ag * dy;
    J[2] = -a * t;
    J[3] = -a * (gy * t + gz2 * dy);
    J[4] = a * (gx * t + gz2 * dx);
    J[5] = ag * (gx * dy - gy * dx);
    const T r = nrm * (robust ? T(1.0) : sqrt_sigma2);
    const T xx = r * (robust ? inv_s_p : 1.0);
    const T w = t_rcp(1.0 + xx * xx) * wmask;
    T Jw[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Jw[i] = J[i] * w;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) t_acc(acc[k++], Jw[i], J[j]);
#pragma unroll
    for (int i = 0; i < 6; ++i) t_acc(acc[21 + i], Jw[i], r);
    t_acc(acc[27], r * r, w);
}
PM_HD void point_term_q(double* acc, const double* DT, const Cam5& cam, double homog_th, double inv_homog, double X, double Y, double Z,
                        double ox, double oy, double sqrt_sigma2, bool robust, double inv_s_p, double wmask = 1.0) {
    point_term_t<double>(acc, DT, cam, homog_th, inv_homog, X, Y, Z, ox, oy, sqrt_sigma2, robust, inv_s_p, wmask);
}

// the gradient of a line end-point: grad6 scaled by `a`, factored as in point_term_t
PM_HD void grad6_scaled(const double* Pc, double dx, double dy, double a, double gz2, double* J) {
    const double gx = Pc[0], gy = Pc[1], gz = Pc[2];
    const double t = gx * dx + gy * dy;
    const double ag = a * gz;
    J[0] = ag * dx;
    J[1] = ag * dy;
    J[2] = -a * t;
    J[3] = -a * (gy * t + gz2 * dy);
    J[4] = a * (gx * t + gz2 * dx);
    J[5] = ag * (gx * dy - gy * dx);
}
PM_HD double fgz2_of(double fx, double homog_th, double inv_homog, double gz2, double iz) {
    return fx * (gz2 > homog_th ? iz * iz : inv_homog);
}
PM_HD double inv_clamped(double homog_th, double inv_homog, double x) {  // 1 / max(homog_th, x)
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = fast_rcp(x);  // (not selected for x == 0)
    return x > homog_th ? y : inv_homog;
#else
    return 1.0 / dmax(homog_th, x);
#endif
}

struct LineRec {
    double sP[3], eP[3], le[3], spl[2], epl[2], sigma2;
};

PM_HD double line_residual(const double* DT, const Cam5& cam, const LineRec& L) {
    double Pc[3], s[2], t[2];
    transform_project(DT, L.sP[0], L.sP[1], L.sP[2], cam, Pc, s);
    transform_project(DT, L.eP[0], L.eP[1], L.eP[2], cam, Pc, t);
    const double ds = L.le[0] * s[0] + L.le[1] * s[1] + L.le[2];
    const double de = L.le[0] * t[0] + L.le[1] * t[1] + L.le[2];
    return sqrt(ds * ds + de * de);
}

PM_HD void line_term(double* acc, const double* DT, const Cam5& cam, double homog_th, const LineRec& L, bool robust,
                     double s_l) {
    double sPc[3], ePc[3], s[2], t[2], Js[6], Je[6], J[6];
    transform_project(DT, L.sP[0], L.sP[1], L.sP[2], cam, sPc, s);
    transform_project(DT, L.eP[0], L.eP[1], L.eP[2], cam, ePc, t);
    const double ds = L.le[0] * s[0] + L.le[1] * s[1] + L.le[2];
    const double de = L.le[0] * t[0] + L.le[1] * t[1] + L.le[2];
    const double nrm = sqrt(ds * ds + de * de);
    grad6(sPc, L.le[0], L.le[1], cam.fx, homog_th, Js);
    grad6(ePc, L.le[0], L.le[1], cam.fx, homog_th, Je);
    const double iden = 1.0 / dmax(homog_th, nrm);
#pragma unroll
    for (int i = 0; i < 6; ++i) J[i] = (Js[i] * ds + Je[i] * de) * iden;
    double r, w;
    if (!robust) {
        r = nrm * sqrt(L.sigma2);
        w = 1.0 / (1.0 + r * r);
    } else {
        r = nrm;
        const double xx = r / s_l;
        w = 1.0 / (1.0 + xx * xx);
    }
    w *= line_overlap(L.spl[0], L.spl[1], L.epl[0], L.epl[1], s[0], s[1], t[0], t[1]);
    accumulate28(acc, J, r, w);
}

// line_term with L.sigma2 already holding sqrt(sigma2), weight-folded accumulation, the reciprocals / square root of the point
// term, and J = (Js ds + Je de) / max(homogTh, |err|) with the two scale factors folded into the end-point gradients
// (inv_s_l: reciprocal of the robust scale, as for the points).  The overlap weight is the reference's, operation for operation.
PM_HD void line_term_q(double* acc, const double* DT, const Cam5& cam, double homog_th, double inv_homog, const LineRec& L, bool robust,
                       double inv_s_l) {
    double sPc[3], ePc[3], Js[6], Je[6], J[6];
    sPc[0] = DT[0] * L.sP[0] + DT[1] * L.sP[1] + DT[2] * L.sP[2] + DT[3];
    sPc[1] = DT[4] * L.sP[0] + DT[5] * L.sP[1] + DT[6] * L.sP[2] + DT[7];
    sPc[2] = DT[8] * L.sP[0] + DT[9] * L.sP[1] + DT[10] * L.sP[2] + DT[11];
    ePc[0] = DT[0] * L.eP[0] + DT[1] * L.eP[1] + DT[2] * L.eP[2] + DT[3];
    ePc[1] = DT[4] * L.eP[0] + DT[5] * L.eP[1] + DT[6] * L.eP[2] + DT[7];
    ePc[2] = DT[8] * L.eP[0] + DT[9] * L.eP[1] + DT[10] * L.eP[2] + DT[11];
    const double izs = fast_rcp(sPc[2]), ize = fast_rcp(ePc[2]);
    const double s0 = cam.cx + cam.fx * sPc[0] * izs, s1 = cam.cy + cam.fy * sPc[1] * izs;
    const double t0 = cam.cx + cam.fx * ePc[0] * ize, t1 = cam.cy + cam.fy * ePc[1] * ize;
    const double ds = L.le[0] * s0 + L.le[1] * s1 + L.le[2];
    const double de = L.le[0] * t0 + L.le[1] * t1 + L.le[2];
    const double nrm = fast_sqrt(ds * ds + de * de);
    const double iden = inv_clamped(homog_th, inv_homog, nrm);
    const double gs2 = sPc[2] * sPc[2], ge2 = ePc[2] * ePc[2];
    grad6_scaled(sPc, L.le[0], L.le[1], fgz2_of(cam.fx, homog_th, inv_homog, gs2, izs) * (ds * iden), gs2, Js);
    grad6_scaled(ePc, L.le[0], L.le[1], fgz2_of(cam.fx, homog_th, inv_homog, ge2, ize) * (de * iden), ge2, Je);
#pragma unroll
    for (int i = 0; i < 6; ++i) J[i] = Js[i] + Je[i];
    const double r = nrm * (robust ? 1.0 : L.sigma2);
    const double xx = r * (robust ? inv_s_l : 1.0);
    const double w = fast_rcp(1.0 + xx * xx) * line_overlap(L.spl[0], L.spl[1], L.epl[0], L.epl[1], s0, s1, t0, t1);
    accumulate28w(acc, J, r, w);
}

PM_HD double clamp_scale(double s) {
    const double th_min = 0.0001, th_max = sqrt(7.815);  // src/stereoFrameHandler.cpp:744-745
    if (s < th_min) s = th_min;
    if (s > th_max) s = th_max;
    return s;
}

}  // namespace pm
