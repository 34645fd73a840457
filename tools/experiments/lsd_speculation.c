/* lsd_speculation.c — how much exact parallelism is there in LSD's search loop?  (round-4 experiment for round 5; CPU only)
 *
 * The search of cv::LineSegmentDetector is sequential: seeds in pseudo-order, each region grown against the pixels all earlier regions
 * took (oracle/stvo_lsd_oracle.c; the GPU kernel keeps that order with ONE wavefront per image, ~0.1 s per image).  An EXACT parallel
 * form is optimistic: grow the next K unused seeds against the same committed snapshot, then commit in seed order — a region is valid if
 * its seed is still free and none of its pixels was taken by a region committed before it in this round; the first invalid region ends the
 * round (it and everything after it are grown again against the new snapshot).  This program replays that scheme on the level-line field
 * of a synthetic image and reports, per K: rounds, the critical path in "pixels added one after the other" (sum over rounds of the largest
 * region grown in the round), the work wasted, against the sequential total.
 * Result on a KITTI-size synthetic scene (127.9 k defined pixels, 5467 regions), 2026-09:
 *     next K unused seeds in order         K 16: path 2.8x shorter, work 1.87x      K 64: 5.4x, 3.6x      (same-edge seeds: wasted growth)
 *     picks >= 24 px apart (look-ahead)    K 16: 3.4x shorter, work 1.04x           K 64: 9.0x, 1.11x     K 256: 10x, 1.23x
 * (the separated variant commits out-of-order picks as soon as they do not collide — exact only if validated at their turn in seed
 *  order, see NOTES.md; at K 256 four of 5467 regions differ, at K <= 64 none).
 *   gcc -O2 -o /tmp/lsd_spec tools/experiments/lsd_speculation.c oracle/stvo_lsd_oracle.c oracle/stvo_orb_oracle.c -lm && /tmp/lsd_spec
 * (links the oracle for fastAtan2 / sincos / resize; the growth loop is re-stated here because it needs the snapshot semantics) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

float orc_fast_atan2(float y, float x);
void orc_sincos_det(double x, double* s, double* c);
void orc_lsd_kernel7(double sigma, int32_t* ki);
void orc_resize_linear(const uint8_t* src, int scols, int srows, uint8_t* dst, int dcols, int drows);

#define NOTDEF (-1024.0)
#define PI 3.14159265358979323846
static int W, H;
static double *ang, *mod;
static float *csn; /* cos, sin of float(angle) */

static int aligned(int q, double theta, double prec) {
    const double a = ang[q];
    if (a == NOTDEF) return 0;
    double d = fabs(theta - a);
    if (d > 1.5 * PI) d = fabs(d - 2 * PI);
    return d <= prec;
}
/* grow from seed against `used` (read-only) + `mine` (this region's own marks, stamp = id); returns size, pixels in reg */
static int grow(int seed, const uint8_t* used, int32_t* mine, int32_t id, int32_t* reg, double prec) {
    int n = 0;
    double ra = ang[seed], s, c;
    orc_sincos_det(ra, &s, &c);
    float sx = (float)c, sy = (float)s;
    reg[n++] = seed; mine[seed] = id;
    for (int i = 0; i < n; ++i) {
        const int px = reg[i] % W, py = reg[i] / W;
        for (int yy = py > 0 ? py - 1 : 0; yy <= (py + 1 < H ? py + 1 : H - 1); ++yy)
            for (int xx = px > 0 ? px - 1 : 0; xx <= (px + 1 < W ? px + 1 : W - 1); ++xx) {
                const int q = yy * W + xx;
                if (!used[q] && mine[q] != id && aligned(q, ra, prec)) {
                    mine[q] = id; reg[n++] = q;
                    sx += csn[2 * q]; sy += csn[2 * q + 1];
                    ra = orc_fast_atan2(sy, sx) * (PI / 180);
                }
            }
    }
    return n;
}

int main(int argc, char** argv) {
    const int W0 = 1241, H0 = 376;   /* a KITTI-size image ... */
    W = 1489; H = 451;                /* ... at lsd_scale 1.2 */
    const int npx = W * H, npx0 = W0 * H0;
    uint8_t* img0 = malloc(npx0);
    uint8_t* img = malloc(npx);
    /* synthetic scene: rectangles + sinusoidal shading + noise (a stand-in for stvo_amd.synth.make_image; LCG) */
    uint32_t rs = 12345u;
#define RND() (rs = rs * 1664525u + 1013904223u, (rs >> 8) & 0xFFFF)
    double* f = malloc(sizeof(double) * npx0);
    for (int y = 0; y < H0; ++y) for (int x = 0; x < W0; ++x) f[y * W0 + x] = 110 + 25 * sin(x * 3.1 / W0) + 18 * cos(y * 2.3 / H0);
    for (int r = 0; r < 1000; ++r) {
        int w = 6 + RND() % 84, h = 6 + RND() % 64, x0 = (int)(RND() % (W0 + 20)) - 20, y0 = (int)(RND() % (H0 + 20)) - 20;
        double v = 15 + RND() % 225;
        for (int y = y0 < 0 ? 0 : y0; y < y0 + h && y < H0; ++y) for (int x = x0 < 0 ? 0 : x0; x < x0 + w && x < W0; ++x) f[y * W0 + x] = v;
    }
    for (int i = 0; i < npx0; ++i) {  /* noise sigma 3: sum of 12 uniforms - 6 */
        double n = -6; for (int t = 0; t < 12; ++t) n += (RND() % 10000) / 10000.0;
        double v = f[i] + 3.0 * n; img0[i] = v < 0 ? 0 : v > 255 ? 255 : (uint8_t)lrint(v);
    }
    {   /* the detector's own front: GaussianBlur(7 x 7, sigma 0.6) in 8-bit fixed point, resize x 1.2 (oracle functions) */
        int32_t ki[7]; orc_lsd_kernel7(0.6, ki);
        int32_t* tmp = malloc(sizeof(int32_t) * npx0); uint8_t* bl = malloc(npx0);
#define R101(p, n) ((p) < 0 ? -(p) : ((p) >= (n) ? 2 * (n) - 2 - (p) : (p)))
        for (int y = 0; y < H0; ++y) for (int x = 0; x < W0; ++x) { int a = 0; for (int i = 0; i < 7; ++i) a += ki[i] * img0[y * W0 + R101(x + i - 3, W0)]; tmp[y * W0 + x] = a; }
        for (int y = 0; y < H0; ++y) for (int x = 0; x < W0; ++x) { int a = 0; for (int i = 0; i < 7; ++i) a += ki[i] * tmp[R101(y + i - 3, H0) * W0 + x]; a = (a + (1 << 15)) >> 16; bl[y * W0 + x] = a < 0 ? 0 : a > 255 ? 255 : a; }
        orc_resize_linear(bl, W0, H0, img, W, H);
        free(tmp); free(bl);
    }
    ang = malloc(sizeof(double) * npx); mod = calloc(npx, sizeof(double)); csn = calloc(2 * npx, sizeof(float));
    const double prec = PI * 22.5 / 180, rho = 2.0 / sin(prec);
    double mx = -1;
    for (int i = 0; i < npx; ++i) ang[i] = NOTDEF;
    for (int y = 0; y < H - 1; ++y) for (int x = 0; x < W - 1; ++x) {
        int DA = img[(y + 1) * W + x + 1] - img[y * W + x], BC = img[y * W + x + 1] - img[(y + 1) * W + x], gx = DA + BC, gy = DA - BC;
        double nrm = sqrt((gx * gx + gy * gy) / 4.0); mod[y * W + x] = nrm;
        if (nrm > rho) { ang[y * W + x] = orc_fast_atan2((float)gx, (float)-gy) * (PI / 180); double s, c; orc_sincos_det((double)(float)ang[y * W + x], &s, &c); csn[2 * (y * W + x)] = (float)c; csn[2 * (y * W + x) + 1] = (float)s; if (nrm > mx) mx = nrm; }
    }
    /* order: bins descending, row-major inside */
    int32_t* order = malloc(sizeof(int32_t) * npx); int n_order = 0;
    { int* start = calloc(1026, sizeof(int)); double bc = 1023 / mx;
      for (int i = 0; i < npx; ++i) if (ang[i] != NOTDEF) start[1023 - (int)(mod[i] * bc) + 1]++;
      for (int b = 0; b < 1024; ++b) start[b + 1] += start[b];
      for (int i = 0; i < npx; ++i) if (ang[i] != NOTDEF) { order[start[1023 - (int)(mod[i] * bc)]++] = i; ++n_order; }
      free(start); }
    const int Ks[] = {1, 4, 16, 64, 256};
    const int min_sep[] = {0, 24, 24};  /* 0: the next K unused seeds; 24: additionally at least 24 px apart (Chebyshev) from the round's earlier picks; third pass: separation 24 with the STRICT (exact) commit rule */
    uint8_t* used = malloc(npx); int32_t* mine = malloc(sizeof(int32_t) * npx); int32_t* claimed = malloc(sizeof(int32_t) * npx);
    int32_t** regs = malloc(sizeof(int32_t*) * 256); for (int k = 0; k < 256; ++k) regs[k] = malloc(sizeof(int32_t) * npx / 4 + 4096);
    printf("%d defined pixels\n", n_order);
    for (int ms = 0; ms < 3; ++ms)
    for (int ki = 0; ki < 5; ++ki) {
        const int K = Ks[ki], sep = min_sep[ms];
        const int strict = ms == 2;
        if (K == 1 && ms) continue;
        memset(used, 0, npx); memset(mine, 0xFF, sizeof(int32_t) * npx); memset(claimed, 0xFF, sizeof(int32_t) * npx);
        long long rounds = 0, crit = 0, work = 0, useful = 0, regions = 0; int32_t id = 0; int ptr = 0;
        while (1) {
            while (ptr < n_order && used[order[ptr]]) ++ptr;
            if (ptr >= n_order) break;
            int seeds[256], sz[256], rank[256], ns = 0;
            for (int p = ptr; p < n_order && ns < K; ++p) {
                const int q = order[p]; if (used[q]) continue;
                int ok = 1;
                if (sep) for (int j = 0; j < ns; ++j) { int dx = abs(q % W - seeds[j] % W), dy = abs(q / W - seeds[j] / W); if ((dx > dy ? dx : dy) < sep) { ok = 0; break; } }
                if (ok) { rank[ns] = p; seeds[ns++] = q; }
                if (p - ptr > 4096) break; /* look-ahead window */
            }
            int big = 0;
            for (int j = 0; j < ns; ++j) { sz[j] = grow(seeds[j], used, mine, ++id, regs[j], prec); work += sz[j]; if (sz[j] > big) big = sz[j]; }
            /* commit in rank order; with min_sep the picks are NOT the next seeds in order, so only the first pick is certainly in turn:
             * a later pick commits if no skipped earlier seed could touch it — approximated optimistically here by the conflict test alone */
            const int32_t round_id = (int32_t)rounds;
            int scan = ptr;  /* strict: every seed of rank < rank[j] must be used before pick j may commit */
            for (int j = 0; j < ns; ++j) {
                if (strict) {
                    while (scan < rank[j] && used[order[scan]]) ++scan;
                    if (scan < rank[j]) break;  /* an unresolved seed of lower rank: the rest of the picks wait (here: are thrown away) */
                }
                if (used[seeds[j]]) continue;  /* taken by a region committed in this round: nothing to commit */
                int conflict = 0;
                for (int t = 0; t < sz[j] && !conflict; ++t) conflict = claimed[regs[j][t]] == round_id;
                if (conflict) break;
                for (int t = 0; t < sz[j]; ++t) { used[regs[j][t]] = 1; claimed[regs[j][t]] = round_id; }
                useful += sz[j]; ++regions;
            }
            crit += big; ++rounds;
        }
        printf("%sK %3d  separation %2d : rounds %7lld  critical path %8lld pixel steps (%.1fx shorter than sequential)  work %9lld (%.2fx the useful %lld)  regions %lld\n",
               strict ? "strict " : "", K, sep, rounds, crit, (double)useful / (double)crit, work, (double)work / (double)useful, useful, regions);
    }
    return 0;
}
