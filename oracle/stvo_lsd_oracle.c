/*
 * stvo_lsd_oracle.c — TEST INFRASTRUCTURE ONLY: CPU restatement of the key-line DETECTOR the reference obtains from
 *     lsd->detect(img, lines, Config::lsdScale(), 1, opts);   + the top-N cut by response
 * in StereoFrame::detectLineFeatures (/root/reference/src/stereoFrame.cpp:207-237), i.e.
 *   LSDDetectorC::detectImpl (3rdparty/line_descriptor/src/LSDDetector_custom.cpp:227-325; the reference HOLDS this wrapper):
 *       one octave (the image itself, :60-63), cv::createLineSegmentDetector(opts...)->detect, checkLineExtremes (:75-100),
 *       the min_length filter (:274-276), KeyLine fields (:278-299: lineLength, numOfPixels = cv::LineIterator count, angle,
 *       response = lineLength / max(cols, rows)), class_id in detection order;
 *   stereoFrame.cpp:231-240: if more than lsd_nfeatures lines, sort by response (descending), keep the first lsd_nfeatures;
 * and of the detector core itself, cv::LineSegmentDetector (OpenCV 3.x imgproc/src/lsd.cpp) — THIRD-PARTY code that is NOT
 * under /root/reference (OpenCV is found by CMake, CMakeLists.txt:4, no pinned version).  It is restated from its published
 * algorithm: R. Grompone von Gioi, J. Jakubowicz, J.-M. Morel, G. Randall, "LSD: a Line Segment Detector", IPOL 2012
 * (doi 10.5201/ipol.2012.gjmr-lsd), in the structure of OpenCV's implementation as the builder remembers it:
 *   flsd        Gaussian blur (sigma = sigma_scale / scale if scale < 1 else sigma_scale, kernel 1 + 2 ceil(sigma sqrt(2 * 3 ln 10)))
 *               + resize by `scale` (8-bit data, cv::GaussianBlur / cv::resize INTER_LINEAR in their fixed-point forms — the same
 *               restatements oracle/stvo_orb_oracle.c uses), ll_angle, then for every pixel in pseudo-order of decreasing gradient:
 *               region_grow, the minimum region size -log10(NT) / log10(p), region2rect, +0.5 offset, division by `scale`;
 *   ll_angle    2 x 2 gradient (gx = DA + BC, gy = DA - BC on integers), norm = sqrt((gx^2 + gy^2) / 4), angle =
 *               fastAtan2(gx, -gy) in degrees (float polynomial) x pi / 180, undefined where norm <= quant / sin(ang_th);
 *               pseudo-ordering into n_bins bins of int(norm (n_bins - 1) / max_norm), highest bin first;
 *   region_grow 8-neighbourhood, neighbours visited row by row, a pixel joins when unused and within `prec` of the running
 *               region angle fastAtan2(sum sin, sum cos) (float sums);
 *   region2rect modgrad-weighted centroid, principal direction from the inertia matrix (smallest eigenvalue), extent of the
 *               projections, width >= 1.
 * Only lsd_refine = 0 (LSD_REFINE_NONE) is restated: every shipped configuration uses it (config/config/ all .yaml: lsd_refine : 0,
 * src/config.cpp:105); the refinement / NFA branches (refine, rect_improve, rect_nfa) return STVO_ERR_UNSUPPORTED here.
 *
 * PARITY UNPINNED: no OpenCV in this image, no test vector in the reference.  Points where the restatement
 * DEFINES behaviour that OpenCV leaves to the platform, followed bit for bit by the HIP kernels (stvo-pl_amd/csrc/lsd_kernels.hip):
 *   (1) order inside a gradient bin: OpenCV sorts the pixel list with std::sort on the bin number only (not stable); here
 *       pixels of a bin keep row-major order — the order of the published algorithm's per-bin lists;
 *   (2) cos / sin: the float cos / sin of region_grow and the double ones of region2rect are evaluated with orc_sincos_det
 *       (Cody-Waite reduction + the fdlibm kernel polynomials, no fused multiply-adds except the two of the reduction) and, for
 *       the float ones, rounded to float — within an ulp of the platform's libm, identical on host and device;
 *   (3) ties of equal response at the top-N cut keep detection order (std::sort would leave them in unspecified order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/stvo_types.h"

float orc_fast_atan2(float y, float x);                                                        /* stvo_orb_oracle.c */
void orc_resize_linear_fxy(const uint8_t* src, int scols, int srows, uint8_t* dst, int dcols, int drows, double inv_x, double inv_y); /* stvo_orb_oracle.c */

#define LSD_NOTDEF (-1024.0)
#define LSD_PI 3.14159265358979323846
#define LSD_DEG_TO_RADS (LSD_PI / 180)
#define LSD_M_3_2_PI ((3 * LSD_PI) / 2)
#define LSD_M_2_PI (2 * LSD_PI)

/* ---- deterministic double sin / cos (see (2) above); |x| a few turns at most ---- */
void orc_sincos_det(double x, double* s, double* c) {
    static const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17, TWO_OVER_PI = 6.36619772367581382433e-01;
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                        S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                        C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double k = rint(x * TWO_OVER_PI);
    double r = fma(-k, PIO2_HI, x);
    r = fma(-k, PIO2_LO, r);
    const double z = r * r;
    const double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double sn = r + (z * r) * (S1 + z * ps);
    const double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double cs = 1.0 - (0.5 * z - z * pc);
    const int q = (int)((long long)k & 3);
    *s = q == 0 ? sn : (q == 1 ? cs : (q == 2 ? -sn : -cs));
    *c = q == 0 ? cs : (q == 1 ? -sn : (q == 2 ? -cs : sn));
}

static int reflect101(int p, int n) { /* BORDER_REFLECT_101 */
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

/* cv::GaussianBlur(img, out, Size(7, 7), sigma) on 8-bit data, the fixed-point form of orc_gaussian_blur7 with sigma a parameter */
void orc_lsd_kernel7(double sigma, int32_t* ki /* [7] */) {
    double k[7], sum = 0.0;
    for (int i = 0; i < 7; ++i) {
        const double x = i - 3;
        k[i] = exp(-x * x / (2.0 * sigma * sigma));
        sum += k[i];
    }
    for (int i = 0; i < 7; ++i) ki[i] = (int)lrint((float)(k[i] / sum) * 256.0);
}
static void gaussian_blur7_sigma(const uint8_t* img, int cols, int rows, double sigma, uint8_t* out) {
    int32_t ki[7];
    orc_lsd_kernel7(sigma, ki);
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)cols * rows);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int i = 0; i < 7; ++i) s += ki[i] * img[y * cols + reflect101(x + i - 3, cols)];
            tmp[y * cols + x] = s;
        }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int i = 0; i < 7; ++i) s += ki[i] * tmp[reflect101(y + i - 3, rows) * cols + x];
            s = (s + (1 << 15)) >> 16;
            out[y * cols + x] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
        }
    free(tmp);
}

typedef struct {
    int refine;  /* 0 only */
    double scale, sigma_scale, quant, ang_th, log_eps, density_th;
    int n_bins;
    double min_length; /* LSDOptions::min_length (LSDDetector_custom.cpp:275) */
    int nfeatures;     /* Config::lsdNFeatures(), 0: keep all (stereoFrame.cpp:233) */
} orc_lsd_opts;

/* one detected key-line: what KeyLine holds of it (octave 0: the in-octave coordinates equal the image ones) */
typedef struct {
    float sx, sy, ex, ey;
    float length, response, angle;
    int32_t num_pixels;
} orc_keyline;

/* the size of the scaled image: resize(..., Size(), scale, scale) = Size(cvRound(cols scale), cvRound(rows scale)) */
void orc_lsd_scaled_size(int cols, int rows, double scale, int* w, int* h) {
    *w = (int)lrint((double)cols * scale);
    *h = (int)lrint((double)rows * scale);
}
/* kernel size 1 + 2 h of the blur; 7 for every sigma between 0.54 and 0.80 */
int orc_lsd_ksize(double scale, double sigma_scale, double* sigma_out) {
    const double sigma = (scale < 1) ? (sigma_scale / scale) : sigma_scale;
    const unsigned h = (unsigned)ceil(sigma * sqrt(2 * 3.0 * log(10.0)));
    if (sigma_out) *sigma_out = sigma;
    return 1 + 2 * (int)h;
}
double orc_lsd_rho(double quant, double ang_th) { return quant / sin(LSD_PI * ang_th / 180); }
int orc_lsd_min_reg_size(int w, int h, double ang_th) {
    const double p = ang_th / 180;
    const double log_nt = 5 * (log10((double)w) + log10((double)h)) / 2 + log10(11.0);
    return (int)(size_t)(-log_nt / log10(p));
}

/* cv::LineIterator(img, Point2f(sx, sy), Point2f(ex, ey)).count: Point2f -> Point by cvRound, clipped to the image, 8-connected */
static int clip_line(int w, int h, long long* x1, long long* y1, long long* x2, long long* y2) {
    /* cv::clipLine(Size, Point&, Point&) (imgproc/src/drawing.cpp), on 64-bit integers */
    const long long right = w - 1, bottom = h - 1;
    if (w <= 0 || h <= 0) return 0;
    int c1 = (*x1 < 0) + (*x1 > right) * 2 + (*y1 < 0) * 4 + (*y1 > bottom) * 8;
    int c2 = (*x2 < 0) + (*x2 > right) * 2 + (*y2 < 0) * 4 + (*y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            *x1 += (long long)((double)(a - *y1) * (*x2 - *x1) / (*y2 - *y1));
            *y1 = a;
            c1 = (*x1 < 0) + (*x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            *x2 += (long long)((double)(a - *y2) * (*x2 - *x1) / (*y2 - *y1));
            *y2 = a;
            c2 = (*x2 < 0) + (*x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                *y1 += (long long)((double)(a - *x1) * (*y2 - *y1) / (*x2 - *x1));
                *x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                *y2 += (long long)((double)(a - *x2) * (*y2 - *y1) / (*x2 - *x1));
                *x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}
int orc_line_iterator_count(int cols, int rows, float sx, float sy, float ex, float ey) {
    long long x1 = lrintf(sx), y1 = lrintf(sy), x2 = lrintf(ex), y2 = lrintf(ey);
    if (!clip_line(cols, rows, &x1, &y1, &x2, &y2)) return 0;
    const long long dx = llabs(x2 - x1), dy = llabs(y2 - y1);
    return (int)((dx > dy ? dx : dy) + 1);
}

typedef struct {
    double x1, y1, x2, y2, width;
} lsd_rect;

/* developer aid (tools/lsd_probe.py): when set, 8 doubles per segment — cx, cy, Ixx, Iyy, Ixy, theta, l_min, l_max */
double* orc_lsd_debug_out = NULL;
void orc_lsd_set_debug(double* p) { orc_lsd_debug_out = p; }

static int is_aligned(const double* angles, int w, int h, int x, int y, double theta, double prec) {
    if (x < 0 || y < 0 || x >= w || y >= h) return 0;
    const double a = angles[(size_t)y * w + x];
    if (a == LSD_NOTDEF) return 0;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > LSD_M_3_2_PI) {
        n_theta -= LSD_M_2_PI;
        if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
}

static double angle_diff(double a, double b) {
    double diff = a - b;
    while (diff <= -LSD_PI) diff += LSD_M_2_PI;
    while (diff > LSD_PI) diff -= LSD_M_2_PI;
    return fabs(diff);
}

static double angle_diff_signed(double a, double b) {
    double diff = a - b;
    while (diff <= -LSD_PI) diff += LSD_M_2_PI;
    while (diff > LSD_PI) diff -= LSD_M_2_PI;
    return diff;
}

/* what the search works on (one image) */
typedef struct {
    int w, h;
    const double* angles;
    const double* modgrad;
    uint8_t* used;
    int32_t* reg;       /* the current region: pixel indices in the order they joined */
    double dbg[8];      /* region2rect's intermediates of the last rectangle (orc_lsd_set_debug) */
} lsd_search;

/* region_grow: the region of `seed` under the angle tolerance `prec`; returns its size, the list in c->reg, its angle in *reg_angle_out */
static int region_grow(lsd_search* c, int32_t seed, double prec, double* reg_angle_out) {
    const int w = c->w, h = c->h;
    int32_t* reg = c->reg;
    int n_reg = 0;
    double reg_angle = c->angles[seed];
    reg[n_reg++] = seed;
    double sn, cs;
    orc_sincos_det(reg_angle, &sn, &cs);
    float sumdx = (float)cs, sumdy = (float)sn;
    c->used[seed] = 1;
    for (int i = 0; i < n_reg; ++i) {
        const int px = reg[i] % w, py = reg[i] / w;
        const int xx_min = px - 1 > 0 ? px - 1 : 0, xx_max = px + 1 < w - 1 ? px + 1 : w - 1;
        const int yy_min = py - 1 > 0 ? py - 1 : 0, yy_max = py + 1 < h - 1 ? py + 1 : h - 1;
        for (int yy = yy_min; yy <= yy_max; ++yy)
            for (int xx = xx_min; xx <= xx_max; ++xx) {
                const size_t q = (size_t)yy * w + xx;
                if (!c->used[q] && is_aligned(c->angles, w, h, xx, yy, reg_angle, prec)) {
                    const double angle = c->angles[q];
                    c->used[q] = 1;
                    reg[n_reg++] = (int32_t)q;
                    orc_sincos_det((double)(float)angle, &sn, &cs); /* cos(float(angle)), sin(float(angle)) */
                    sumdx += (float)cs;
                    sumdy += (float)sn;
                    reg_angle = orc_fast_atan2(sumdy, sumdx) * LSD_DEG_TO_RADS;
                }
            }
    }
    *reg_angle_out = reg_angle;
    return n_reg;
}

/* region2rect: the rectangle of the region c->reg[0 .. n_reg) (coordinates of the scaled image, no offset yet) */
static void region2rect(lsd_search* c, int n_reg, double reg_angle, double prec, lsd_rect* rec) {
    const int w = c->w;
    const int32_t* reg = c->reg;
    double x = 0, y = 0, sum = 0;
    for (int i = 0; i < n_reg; ++i) {
        const double weight = c->modgrad[reg[i]];
        x += (double)(reg[i] % w) * weight;
        y += (double)(reg[i] / w) * weight;
        sum += weight;
    }
    x /= sum;
    y /= sum;
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (int i = 0; i < n_reg; ++i) {
        const double weight = c->modgrad[reg[i]];
        const double ddx = (double)(reg[i] % w) - x, ddy = (double)(reg[i] / w) - y;
        Ixx += ddy * ddy * weight;
        Iyy += ddx * ddx * weight;
        Ixy -= ddx * ddy * weight;
    }
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)orc_fast_atan2((float)(lambda - Ixx), (float)Ixy)
                                           : (double)orc_fast_atan2((float)Ixy, (float)(lambda - Iyy));
    theta *= LSD_DEG_TO_RADS;
    if (angle_diff(theta, reg_angle) > prec) theta += LSD_PI;
    double dx, dy;
    orc_sincos_det(theta, &dy, &dx);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int i = 0; i < n_reg; ++i) {
        const double regdx = (double)(reg[i] % w) - x, regdy = (double)(reg[i] / w) - y;
        const double l = regdx * dx + regdy * dy;
        const double ww = -regdx * dy + regdy * dx;
        if (l > l_max) l_max = l;
        else if (l < l_min) l_min = l;
        if (ww > w_max) w_max = ww;
        else if (ww < w_min) w_min = ww;
    }
    rec->x1 = x + l_min * dx;
    rec->y1 = y + l_min * dy;
    rec->x2 = x + l_max * dx;
    rec->y2 = y + l_max * dy;
    rec->width = w_max - w_min;
    if (rec->width < 1.0) rec->width = 1.0;
    c->dbg[0] = x; c->dbg[1] = y; c->dbg[2] = Ixx; c->dbg[3] = Iyy; c->dbg[4] = Ixy; c->dbg[5] = theta; c->dbg[6] = l_min; c->dbg[7] = l_max;
}

static double lsd_dist(double x1, double y1, double x2, double y2) { return sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)); }
static double lsd_dist_sq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
static double lsd_density(const lsd_rect* rec, int n_reg) { return (double)n_reg / (lsd_dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width); }

/* reduce_region_radius (lsd_refine >= 1): while the region is too sparse for its rectangle, drop the points farther than 75 % of
 * the current radius from the seed (a dropped point is free again; the last point of the list takes its place), fit again.
 * Returns the new size, 0 when fewer than two points are left (the region is given up). */
static int reduce_region_radius(lsd_search* c, int n_reg, double reg_angle, double prec, lsd_rect* rec, double density, double density_th) {
    const int w = c->w;
    int32_t* reg = c->reg;
    const double xc = (double)(reg[0] % w), yc = (double)(reg[0] / w);
    const double r1 = lsd_dist_sq(xc, yc, rec->x1, rec->y1), r2 = lsd_dist_sq(xc, yc, rec->x2, rec->y2);
    double rad_sq = r1 > r2 ? r1 : r2;
    while (density < density_th) {
        rad_sq *= 0.75 * 0.75;
        for (int i = 0; i < n_reg; ++i)
            if (lsd_dist_sq(xc, yc, (double)(reg[i] % w), (double)(reg[i] / w)) > rad_sq) {
                c->used[reg[i]] = 0;
                const int32_t t = reg[i];
                reg[i] = reg[n_reg - 1];
                reg[n_reg - 1] = t;
                --n_reg;
                --i; /* the point that took the place is looked at next */
            }
        if (n_reg < 2) return 0;
        region2rect(c, n_reg, reg_angle, prec, rec);
        density = lsd_density(rec, n_reg);
    }
    return n_reg;
}

/* refine (lsd_refine >= 1): a region dense enough for its rectangle stays; otherwise it is grown again from the same seed under a
 * tolerance of twice the standard deviation of the angles near the seed, and, if still too sparse, cut back by radius.
 * Returns the new size (*reg_angle_io updated), 0 when the region is given up. */
static int refine_region(lsd_search* c, int n_reg, double* reg_angle_io, double prec, lsd_rect* rec, double density_th) {
    const int w = c->w;
    int32_t* reg = c->reg;
    double density = lsd_density(rec, n_reg);
    if (density >= density_th) return n_reg;
    const double xc = (double)(reg[0] % w), yc = (double)(reg[0] / w);
    const double ang_c = c->angles[reg[0]];
    double sum = 0, s_sum = 0;
    int n = 0;
    for (int i = 0; i < n_reg; ++i) {
        c->used[reg[i]] = 0;
        if (lsd_dist(xc, yc, (double)(reg[i] % w), (double)(reg[i] / w)) < rec->width) {
            const double ang_d = angle_diff_signed(c->angles[reg[i]], ang_c);
            sum += ang_d;
            s_sum += ang_d * ang_d;
            ++n;
        }
    }
    const double mean_angle = sum / (double)n; /* (the seed itself is always within the width: n >= 1) */
    const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
    n_reg = region_grow(c, reg[0], tau, reg_angle_io);
    if (n_reg < 2) return 0;
    region2rect(c, n_reg, *reg_angle_io, prec, rec);
    density = lsd_density(rec, n_reg);
    if (density < density_th) return reduce_region_radius(c, n_reg, *reg_angle_io, prec, rec, density, density_th);
    return n_reg;
}

/* The detector core on an 8-bit image: segments (x1, y1, x2, y2) as cv::Vec4f in the coordinates of `img`, in detection order.
 * Returns the number found (all of them are counted, at most `cap` are stored), or < 0 on error.
 * lsd_refine: 0 (LSD_REFINE_NONE) and 1 (LSD_REFINE_STD: refine / reduce_region_radius above); 2 (LSD_REFINE_ADV: rect_improve + the
 * NFA test) is not restated. */
int orc_lsd_segments(const uint8_t* img, int cols, int rows, const orc_lsd_opts* o, float* seg /* [cap][4] */, int cap) {
    if (o->refine != 0 && o->refine != 1) return STVO_ERR_UNSUPPORTED;
    const double prec = LSD_PI * o->ang_th / 180;
    const double rho = orc_lsd_rho(o->quant, o->ang_th);
    int w = cols, h = rows;
    uint8_t* scaled = NULL;
    if (o->scale != 1) {
        double sigma;
        if (orc_lsd_ksize(o->scale, o->sigma_scale, &sigma) != 7) return STVO_ERR_UNSUPPORTED;
        uint8_t* blur = (uint8_t*)malloc((size_t)cols * rows);
        gaussian_blur7_sigma(img, cols, rows, sigma, blur);
        orc_lsd_scaled_size(cols, rows, o->scale, &w, &h);
        scaled = (uint8_t*)malloc((size_t)w * h);
        orc_resize_linear_fxy(blur, cols, rows, scaled, w, h, o->scale, o->scale); /* resize(.., Size(), scale, scale): positions from 1 / scale */
        free(blur);
    } else {
        scaled = (uint8_t*)malloc((size_t)w * h);
        memcpy(scaled, img, (size_t)w * h);
    }
    const size_t npx = (size_t)w * h;
    double* angles = (double*)malloc(sizeof(double) * npx);
    double* modgrad = (double*)calloc(npx, sizeof(double));
    uint8_t* used = (uint8_t*)calloc(npx, 1);
    int32_t* order = (int32_t*)malloc(sizeof(int32_t) * npx);
    int32_t* reg = (int32_t*)malloc(sizeof(int32_t) * npx);
    /* ---- ll_angle ---- */
    for (size_t i = 0; i < npx; ++i) angles[i] = LSD_NOTDEF; /* incl. the last row and column */
    double max_grad = -1;
    for (int y = 0; y < h - 1; ++y)
        for (int x = 0; x < w - 1; ++x) {
            const int DA = (int)scaled[(size_t)(y + 1) * w + x + 1] - (int)scaled[(size_t)y * w + x];
            const int BC = (int)scaled[(size_t)y * w + x + 1] - (int)scaled[(size_t)(y + 1) * w + x];
            const int gx = DA + BC, gy = DA - BC;
            const double norm = sqrt((gx * gx + gy * gy) / 4.0);
            modgrad[(size_t)y * w + x] = norm;
            if (norm <= rho) {
                angles[(size_t)y * w + x] = LSD_NOTDEF;
            } else {
                angles[(size_t)y * w + x] = orc_fast_atan2((float)gx, (float)-gy) * LSD_DEG_TO_RADS;
                if (norm > max_grad) max_grad = norm;
            }
        }
    /* pseudo-ordering: bins from the highest to the lowest, row-major inside a bin (see (1)); every pixel of the gradient's domain */
    const double bin_coef = (max_grad > 0) ? (double)(o->n_bins - 1) / max_grad : 0;
    int32_t* bin_start = (int32_t*)calloc((size_t)o->n_bins + 1, sizeof(int32_t));
    for (int y = 0; y < h - 1; ++y)
        for (int x = 0; x < w - 1; ++x) {
            int b = (int)(modgrad[(size_t)y * w + x] * bin_coef);
            b = b < 0 ? 0 : (b >= o->n_bins ? o->n_bins - 1 : b);
            bin_start[o->n_bins - 1 - b + 1]++; /* slot of the bin in descending order, shifted by one for the prefix sum */
        }
    for (int b = 0; b < o->n_bins; ++b) bin_start[b + 1] += bin_start[b];
    size_t n_order = 0;
    for (int y = 0; y < h - 1; ++y)
        for (int x = 0; x < w - 1; ++x) {
            int b = (int)(modgrad[(size_t)y * w + x] * bin_coef);
            b = b < 0 ? 0 : (b >= o->n_bins ? o->n_bins - 1 : b);
            order[bin_start[o->n_bins - 1 - b]++] = (int32_t)((size_t)y * w + x);
            ++n_order;
        }
    free(bin_start);
    /* ---- the search ---- */
    const int min_reg_size = orc_lsd_min_reg_size(w, h, o->ang_th);
    lsd_search c;
    c.w = w; c.h = h; c.angles = angles; c.modgrad = modgrad; c.used = used; c.reg = reg;
    int n_seg = 0;
    for (size_t oi = 0; oi < n_order; ++oi) {
        const int32_t seed = order[oi];
        if (used[seed] || angles[seed] == LSD_NOTDEF) continue;
        double reg_angle;
        int n_reg = region_grow(&c, seed, prec, &reg_angle);
        if (n_reg < min_reg_size) continue;
        lsd_rect rec;
        region2rect(&c, n_reg, reg_angle, prec, &rec);
        if (o->refine >= 1) {
            n_reg = refine_region(&c, n_reg, &reg_angle, prec, &rec, o->density_th);
            if (n_reg == 0) continue;
        }
        /* found: the offset, then back to the coordinates of the input image */
        rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
        if (o->scale != 1) {
            rec.x1 /= o->scale; rec.y1 /= o->scale; rec.x2 /= o->scale; rec.y2 /= o->scale;
        }
        if (orc_lsd_debug_out && n_seg < cap) memcpy(orc_lsd_debug_out + 8 * (size_t)n_seg, c.dbg, sizeof(c.dbg));
        if (n_seg < cap) {
            seg[4 * n_seg + 0] = (float)rec.x1;
            seg[4 * n_seg + 1] = (float)rec.y1;
            seg[4 * n_seg + 2] = (float)rec.x2;
            seg[4 * n_seg + 3] = (float)rec.y2;
        }
        ++n_seg;
    }
    free(scaled); free(angles); free(modgrad); free(used); free(order); free(reg);
    return n_seg;
}

/* LSDDetectorC::detectImpl for one octave + the top-N cut of stereoFrame.cpp:231-240.  Returns the number of key-lines written
 * (<= cap), < 0 on error. */
int orc_lsd_detect(const uint8_t* img, int cols, int rows, const orc_lsd_opts* o, orc_keyline* out, int cap) {
    const int seg_cap = (cols * rows) / 16 + 16;
    float* seg = (float*)malloc(sizeof(float) * 4 * (size_t)seg_cap);
    int n_seg = orc_lsd_segments(img, cols, rows, o, seg, seg_cap);
    if (n_seg < 0) {
        free(seg);
        return n_seg;
    }
    if (n_seg > seg_cap) n_seg = seg_cap;
    orc_keyline* kl = (orc_keyline*)malloc(sizeof(orc_keyline) * (size_t)(n_seg > 0 ? n_seg : 1));
    int n = 0;
    for (int k = 0; k < n_seg; ++k) {
        float e[4] = {seg[4 * k], seg[4 * k + 1], seg[4 * k + 2], seg[4 * k + 3]};
        /* checkLineExtremes (:75-100) */
        if (e[0] < 0) e[0] = 0;
        if (e[0] >= cols) e[0] = (float)cols - 1.0f;
        if (e[2] < 0) e[2] = 0;
        if (e[2] >= cols) e[2] = (float)cols - 1.0f;
        if (e[1] < 0) e[1] = 0;
        if (e[1] >= rows) e[1] = (float)rows - 1.0f;
        if (e[3] < 0) e[3] = 0;
        if (e[3] >= rows) e[3] = (float)rows - 1.0f;
        /* :274 length = (float) sqrt(pow(e0 - e2, 2) + pow(e1 - e3, 2)): float differences, squares and root in double */
        const double d0 = (double)(e[0] - e[2]), d1 = (double)(e[1] - e[3]);
        const double length = (double)(float)sqrt(d0 * d0 + d1 * d1);
        if (!(length > o->min_length)) continue;
        orc_keyline q;
        q.sx = e[0]; q.sy = e[1]; q.ex = e[2]; q.ey = e[3];  /* octaveScale = pow((float)scale, 0) = 1 */
        q.length = (float)length;
        q.num_pixels = orc_line_iterator_count(cols, rows, e[0], e[1], e[2], e[3]);
        q.angle = (float)atan2((double)(q.ey - q.sy), (double)(q.ex - q.sx));
        q.response = q.length / (float)(cols > rows ? cols : rows);
        kl[n++] = q;
    }
    free(seg);
    /* stereoFrame.cpp:231-240 */
    int n_out = n;
    if (o->nfeatures != 0 && n > o->nfeatures) {
        /* stable selection sort of the top nfeatures by response (see (3)) — insertion into a sorted prefix */
        orc_keyline* srt = (orc_keyline*)malloc(sizeof(orc_keyline) * (size_t)n);
        int m = 0;
        for (int k = 0; k < n; ++k) {
            int pos = m;
            while (pos > 0 && srt[pos - 1].response < kl[k].response) --pos;
            if (pos >= o->nfeatures) continue;
            const int last = m < o->nfeatures ? m : o->nfeatures - 1;
            for (int j = last; j > pos; --j) srt[j] = srt[j - 1];
            srt[pos] = kl[k];
            if (m < o->nfeatures) ++m;
        }
        memcpy(kl, srt, sizeof(orc_keyline) * (size_t)m);
        free(srt);
        n_out = m;
    }
    if (n_out > cap) n_out = cap;
    memcpy(out, kl, sizeof(orc_keyline) * (size_t)n_out);
    free(kl);
    return n_out;
}
