/* lsd_speculation2.c — the EXACT form of the optimistic search of lsd_speculation.c: speculative regions stay PENDING (they mark nothing)
 * until every seed of lower rank is resolved; at its turn a pending region commits if all its pixels are still free, otherwise the seed is
 * grown again then.  Round structure: walk the seeds in rank order committing valid pending regions until a free seed WITHOUT a valid pending
 * region blocks; that seed plus up to K - 1 new upcoming seeds (free, not pending, >= SEP px from each other and from the blocking seed) are
 * grown in parallel against the current flags; repeat.  Reports rounds, critical path (sum over rounds of the largest region grown), work.
 * The committed regions are exactly the sequential ones (checked: same count and same total size as K = 1).
 *   gcc -O2 -o /tmp/lsd_spec2 tools/experiments/lsd_speculation2.c oracle/stvo_lsd_oracle.c oracle/stvo_orb_oracle.c -lm && /tmp/lsd_spec2 */
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
static float* csn;

static int aligned(int q, double theta, double prec) {
    const double a = ang[q];
    if (a == NOTDEF) return 0;
    double d = fabs(theta - a);
    if (d > 1.5 * PI) d = fabs(d - 2 * PI);
    return d <= prec;
}
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

int main(void) {
    const int W0 = 1241, H0 = 376;
    W = 1489; H = 451;
    const int npx = W * H, npx0 = W0 * H0;
    uint8_t* img0 = malloc(npx0); uint8_t* img = malloc(npx);
    uint32_t rs = 12345u;
#define RND() (rs = rs * 1664525u + 1013904223u, (rs >> 8) & 0xFFFF)
    double* f = malloc(sizeof(double) * npx0);
    for (int y = 0; y < H0; ++y) for (int x = 0; x < W0; ++x) f[y * W0 + x] = 110 + 25 * sin(x * 3.1 / W0) + 18 * cos(y * 2.3 / H0);
    for (int r = 0; r < 1000; ++r) {
        int w = 6 + RND() % 84, h = 6 + RND() % 64, x0 = (int)(RND() % (W0 + 20)) - 20, y0 = (int)(RND() % (H0 + 20)) - 20;
        double v = 15 + RND() % 225;
        for (int y = y0 < 0 ? 0 : y0; y < y0 + h && y < H0; ++y) for (int x = x0 < 0 ? 0 : x0; x < x0 + w && x < W0; ++x) f[y * W0 + x] = v;
    }
    for (int i = 0; i < npx0; ++i) { double n = -6; for (int t = 0; t < 12; ++t) n += (RND() % 10000) / 10000.0; double v = f[i] + 3.0 * n; img0[i] = v < 0 ? 0 : v > 255 ? 255 : (uint8_t)lrint(v); }
    { int32_t ki[7]; orc_lsd_kernel7(0.6, ki); int32_t* tmp = malloc(sizeof(int32_t) * npx0); uint8_t* bl = malloc(npx0);
#define R101(p, n) ((p) < 0 ? -(p) : ((p) >= (n) ? 2 * (n) - 2 - (p) : (p)))
      for (int y = 0; y < H0; ++y) for (int x = 0; x < W0; ++x) { int a = 0; for (int i = 0; i < 7; ++i) a += ki[i] * img0[y * W0 + R101(x + i - 3, W0)]; tmp[y * W0 + x] = a; }
      for (int y = 0; y < H0; ++y) for (int x = 0; x < W0; ++x) { int a = 0; for (int i = 0; i < 7; ++i) a += ki[i] * tmp[R101(y + i - 3, H0) * W0 + x]; a = (a + (1 << 15)) >> 16; bl[y * W0 + x] = a < 0 ? 0 : a > 255 ? 255 : a; }
      orc_resize_linear(bl, W0, H0, img, W, H); free(tmp); free(bl); }
    ang = malloc(sizeof(double) * npx); mod = calloc(npx, sizeof(double)); csn = calloc(2 * npx, sizeof(float));
    const double prec = PI * 22.5 / 180, rho = 2.0 / sin(prec);
    double mx = -1;
    for (int i = 0; i < npx; ++i) ang[i] = NOTDEF;
    for (int y = 0; y < H - 1; ++y) for (int x = 0; x < W - 1; ++x) {
        int DA = img[(y + 1) * W + x + 1] - img[y * W + x], BC = img[y * W + x + 1] - img[(y + 1) * W + x], gx = DA + BC, gy = DA - BC;
        double nrm = sqrt((gx * gx + gy * gy) / 4.0); mod[y * W + x] = nrm;
        if (nrm > rho) { int q = y * W + x; ang[q] = orc_fast_atan2((float)gx, (float)-gy) * (PI / 180); double s, c; orc_sincos_det((double)(float)ang[q], &s, &c); csn[2 * q] = (float)c; csn[2 * q + 1] = (float)s; if (nrm > mx) mx = nrm; }
    }
    int32_t* order = malloc(sizeof(int32_t) * npx); int n_order = 0;
    { int* start = calloc(1026, sizeof(int)); double bc = 1023 / mx;
      for (int i = 0; i < npx; ++i) if (ang[i] != NOTDEF) start[1023 - (int)(mod[i] * bc) + 1]++;
      for (int b = 0; b < 1024; ++b) start[b + 1] += start[b];
      for (int i = 0; i < npx; ++i) if (ang[i] != NOTDEF) { order[start[1023 - (int)(mod[i] * bc)]++] = i; ++n_order; }
      free(start); }
    printf("%d defined pixels\n", n_order);
    uint8_t* used = malloc(npx); int32_t* mine = malloc(sizeof(int32_t) * npx);
    int32_t* pend_of = malloc(sizeof(int32_t) * npx);           /* seed pixel -> pending slot or -1 */
    int32_t** p_px = malloc(sizeof(int32_t*) * (1 << 20));         /* pixel lists of pending regions, freed when resolved */
    enum { MAXP = 1 << 20 };
    int32_t* p_sz = malloc(sizeof(int32_t) * MAXP);
    int32_t* tmpreg = malloc(sizeof(int32_t) * npx);
    const int Ks[] = {1, 8, 16, 32, 64, 128}, seps[] = {16, 24, 40};
    for (int si = 0; si < 3; ++si)
    for (int ki = 0; ki < 6; ++ki) {
        const int K = Ks[ki], SEP = seps[si];
        if (K == 1 && si) continue;
        memset(used, 0, npx); memset(mine, 0xFF, sizeof(int32_t) * npx); memset(pend_of, 0xFF, sizeof(int32_t) * npx);
        long long rounds = 0, crit = 0, work = 0, useful = 0, regions = 0, regrown = 0; int32_t id = 0; int n_pend = 0;
        int scan = 0;
        while (1) {
            /* in rank order: skip used seeds, commit valid pending regions, stop at the first seed that must be grown now */
            int block = -1;
            while (scan < n_order) {
                const int q = order[scan];
                if (used[q]) { ++scan; continue; }
                const int ps = pend_of[q];
                if (ps >= 0) {
                    int ok = 1;
                    for (int t = 0; t < p_sz[ps] && ok; ++t) ok = !used[p_px[ps][t]];
                    pend_of[q] = -1;
                    if (ok) for (int t = 0; t < p_sz[ps]; ++t) used[p_px[ps][t]] = 1;
                    free(p_px[ps]); p_px[ps] = NULL;
                    if (ok) { useful += p_sz[ps]; ++regions; ++scan; continue; }
                    ++regrown;  /* invalid: grow it again now */
                }
                block = q; break;
            }
            if (block < 0) break;
            /* picks: the blocking seed + upcoming free, not-pending seeds, separated from each other */
            int picks[256], np_ = 0; picks[np_++] = block;
            for (int p = scan + 1; p < n_order && np_ < K && p - scan < 16384; ++p) {
                const int q = order[p]; if (used[q] || pend_of[q] >= 0) continue;
                int ok = 1;
                for (int j = 0; j < np_ && ok; ++j) { int dx = abs(q % W - picks[j] % W), dy = abs(q / W - picks[j] / W); ok = (dx > dy ? dx : dy) >= SEP; }
                if (ok) picks[np_++] = q;
            }
            int big = 0;
            for (int j = 0; j < np_; ++j) {
                const int n = grow(picks[j], used, mine, ++id, tmpreg, prec); work += n; if (n > big) big = n;
                if (j == 0) { for (int t = 0; t < n; ++t) used[tmpreg[t]] = 1; useful += n; ++regions; ++scan; }  /* the blocking seed commits at once */
                else if (n_pend < MAXP) { p_px[n_pend] = malloc(sizeof(int32_t) * n); p_sz[n_pend] = n; memcpy(p_px[n_pend], tmpreg, sizeof(int32_t) * n); pend_of[picks[j]] = n_pend++; }
            }
            crit += big; ++rounds;
        }
        printf("K %3d  separation %2d : rounds %6lld  critical path %7lld pixel steps (%.1fx shorter)  work %8lld (%.2fx useful %lld)  regions %lld  regrown %lld\n",
               K, SEP, rounds, crit, (double)useful / (double)crit, work, (double)work / (double)useful, useful, regions, regrown);
    }
    return 0;
}
