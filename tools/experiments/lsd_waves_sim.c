/* lsd_waves_sim.c — event-driven replay of the EXACT multi-wave LSD search sketched in NOTES.md ("pending regions, validated lazily"):
 * one workgroup of W waves per image.  Wave 0 commits in seed order: a seed that has a finished pending region takes it if all its pixels
 * are still free (cost ~ n / 64: lane-parallel flag stores) and grows it again otherwise; a seed another wave is growing right now is waited
 * for; any other seed is grown by wave 0 itself.  Waves 1 .. W - 1 speculate: when idle, a wave takes the first seed after the committer's
 * position (within a window of LOOK ranks) that is free, has no pending region, is not being grown and lies >= SEP px (Chebyshev) from every
 * seed being grown, and grows it against the flags committed so far (a snapshot at its start: the validation at commit makes that exact).
 * Time unit: one pixel added by one wave (the sequential chain of the kernel); a region costs n + C0 units.  Output: makespan against the
 * sequential sum, for W, SEP, LOOK.  The committed regions are the sequential ones (count and total size are checked).
 *   gcc -O2 -o /tmp/lsd_waves tools/experiments/lsd_waves_sim.c oracle/stvo_lsd_oracle.c oracle/stvo_orb_oracle.c -lm && /tmp/lsd_waves [noise [image.raw]]
 * (image.raw: 1241 x 376 bytes instead of the built-in scene, e.g. python -c "import sys; sys.path[:0]=['stvo-pl_amd/python']; from stvo_amd import synth; synth.make_image(500).tofile('/tmp/img500.raw')") */
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
static int W_, H_;
static double *ang, *mod;
static float* csn;
static const double PREC = PI * 22.5 / 180;

static int aligned(int q, double theta) {
    const double a = ang[q];
    if (a == NOTDEF) return 0;
    double d = fabs(theta - a);
    if (d > 1.5 * PI) d = fabs(d - 2 * PI);
    return d <= PREC;
}
static int grow(int seed, const uint8_t* used, int32_t* mine, int32_t id, int32_t* reg) {
    int n = 0;
    double ra = ang[seed], s, c;
    orc_sincos_det(ra, &s, &c);
    float sx = (float)c, sy = (float)s;
    reg[n++] = seed; mine[seed] = id;
    for (int i = 0; i < n; ++i) {
        const int px = reg[i] % W_, py = reg[i] / W_;
        for (int yy = py > 0 ? py - 1 : 0; yy <= (py + 1 < H_ ? py + 1 : H_ - 1); ++yy)
            for (int xx = px > 0 ? px - 1 : 0; xx <= (px + 1 < W_ ? px + 1 : W_ - 1); ++xx) {
                const int q = yy * W_ + xx;
                if (!used[q] && mine[q] != id && aligned(q, ra)) {
                    mine[q] = id; reg[n++] = q;
                    sx += csn[2 * q]; sy += csn[2 * q + 1];
                    ra = orc_fast_atan2(sy, sx) * (PI / 180);
                }
            }
    }
    return n;
}

enum { MAXW = 64 };
typedef struct { int busy; double until; int seed; int32_t* px; int n; int is_commit_of_pending; } Wave;

int main(int argc, char** argv) {
    const double noise = argc > 1 ? atof(argv[1]) : 3.0;
    const int W0 = 1241, H0 = 376;
    W_ = 1489; H_ = 451;
    const int npx = W_ * H_, npx0 = W0 * H0;
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
    for (int i = 0; i < npx0; ++i) { double n = -6; for (int t = 0; t < 12; ++t) n += (RND() % 10000) / 10000.0; double v = f[i] + noise * n; img0[i] = v < 0 ? 0 : v > 255 ? 255 : (uint8_t)lrint(v); }
    if (argc > 2) { FILE* fp = fopen(argv[2], "rb"); if (!fp || fread(img0, 1, npx0, fp) != (size_t)npx0) { printf("cannot read %s\n", argv[2]); return 1; } fclose(fp); }
    { int32_t ki[7]; orc_lsd_kernel7(0.6, ki); int32_t* tmp = malloc(sizeof(int32_t) * npx0); uint8_t* bl = malloc(npx0);
#define R101(p, n) ((p) < 0 ? -(p) : ((p) >= (n) ? 2 * (n) - 2 - (p) : (p)))
      for (int y = 0; y < H0; ++y) for (int x = 0; x < W0; ++x) { int a = 0; for (int i = 0; i < 7; ++i) a += ki[i] * img0[y * W0 + R101(x + i - 3, W0)]; tmp[y * W0 + x] = a; }
      for (int y = 0; y < H0; ++y) for (int x = 0; x < W0; ++x) { int a = 0; for (int i = 0; i < 7; ++i) a += ki[i] * tmp[R101(y + i - 3, H0) * W0 + x]; a = (a + (1 << 15)) >> 16; bl[y * W0 + x] = a < 0 ? 0 : a > 255 ? 255 : a; }
      orc_resize_linear(bl, W0, H0, img, W_, H_); free(tmp); free(bl); }
    ang = malloc(sizeof(double) * npx); mod = calloc(npx, sizeof(double)); csn = calloc(2 * npx, sizeof(float));
    const double rho = 2.0 / sin(PREC);
    double mx = -1;
    for (int i = 0; i < npx; ++i) ang[i] = NOTDEF;
    for (int y = 0; y < H_ - 1; ++y) for (int x = 0; x < W_ - 1; ++x) {
        int DA = img[(y + 1) * W_ + x + 1] - img[y * W_ + x], BC = img[y * W_ + x + 1] - img[(y + 1) * W_ + x], gx = DA + BC, gy = DA - BC;
        double nrm = sqrt((gx * gx + gy * gy) / 4.0); mod[y * W_ + x] = nrm;
        if (nrm > rho) { int q = y * W_ + x; ang[q] = orc_fast_atan2((float)gx, (float)-gy) * (PI / 180); double s, c; orc_sincos_det((double)(float)ang[q], &s, &c); csn[2 * q] = (float)c; csn[2 * q + 1] = (float)s; if (nrm > mx) mx = nrm; }
    }
    int32_t* order = malloc(sizeof(int32_t) * npx); int n_order = 0;
    { int* start = calloc(1026, sizeof(int)); double bc = 1023 / mx;
      for (int i = 0; i < npx; ++i) if (ang[i] != NOTDEF) start[1023 - (int)(mod[i] * bc) + 1]++;
      for (int b = 0; b < 1024; ++b) start[b + 1] += start[b];
      for (int i = 0; i < npx; ++i) if (ang[i] != NOTDEF) { order[start[1023 - (int)(mod[i] * bc)]++] = i; ++n_order; }
      free(start); }
    uint8_t* used = malloc(npx); int32_t* mine = malloc(sizeof(int32_t) * npx);
    int32_t** pend = calloc(npx, sizeof(int32_t*)); int32_t* pend_n = calloc(npx, sizeof(int32_t));  /* finished pending regions by seed */
    int8_t* inflight = calloc(npx, 1);
    int32_t* tmpreg = malloc(sizeof(int32_t) * npx);
    const double C0 = 2.0;  /* per-region overhead in pixel steps (seed set-up, first round) */
    /* sequential reference */
    long long seq_regions = 0, seq_px = 0; double seq_time = 0;
    { memset(used, 0, npx); memset(mine, 0xFF, sizeof(int32_t) * npx); int32_t id = 0;
      for (int p = 0; p < n_order; ++p) { const int q = order[p]; if (used[q]) continue; const int n = grow(q, used, mine, ++id, tmpreg); for (int t = 0; t < n; ++t) used[tmpreg[t]] = 1; ++seq_regions; seq_px += n; seq_time += n + C0; } }
    printf("noise %.0f: %d defined pixels, %lld regions, sequential time %.0f pixel steps\n", noise, n_order, seq_regions, seq_time);
    const int Ws[] = {2, 4, 8, 16, 32}, SEPs[] = {0, 12, 24}, LOOKs[] = {256, 2048, 16384};
    for (int li = 0; li < 3; ++li) for (int si = 0; si < 3; ++si) for (int wi = 0; wi < 5; ++wi) {
        const int NW = Ws[wi], SEP = SEPs[si], LOOK = LOOKs[li];
        if (li != 1 && si != 1) continue;  /* vary one of SEP / LOOK at a time around (12, 2048) */
        memset(used, 0, npx); memset(mine, 0xFF, sizeof(int32_t) * npx); memset(inflight, 0, npx);
        for (int i = 0; i < npx; ++i) if (pend[i]) { free(pend[i]); pend[i] = NULL; }
        Wave wv[MAXW]; memset(wv, 0, sizeof(wv));
        int32_t id = 0; int scan = 0; double now = 0, work = 0; long long regions = 0, px = 0, regrown = 0, taken = 0, self = 0, waited = 0, wasted_regions = 0;
        int done = 0;
        while (!done) {
            /* finish every task that ends now */
            for (int w = 0; w < NW; ++w) if (wv[w].busy && wv[w].until <= now) {
                wv[w].busy = 0;
                if (w == 0) { for (int t = 0; t < wv[0].n; ++t) used[wv[0].px[t]] = 1; ++regions; px += wv[0].n; free(wv[0].px); wv[0].px = NULL; }
                else { inflight[wv[w].seed] = 0; pend[wv[w].seed] = wv[w].px; pend_n[wv[w].seed] = wv[w].n; wv[w].px = NULL; }
            }
            /* the committer */
            if (!wv[0].busy) {
                while (scan < n_order && used[order[scan]]) { const int q = order[scan]; if (pend[q]) { free(pend[q]); pend[q] = NULL; ++wasted_regions; } ++scan; }
                if (scan >= n_order) { done = 1; break; }
                const int q = order[scan];
                if (inflight[q]) { ++waited; /* wait: nothing to start now */ }
                else if (pend[q]) {
                    int ok = 1; for (int t = 0; t < pend_n[q] && ok; ++t) ok = !used[pend[q][t]];
                    if (ok) { wv[0].busy = 1; wv[0].px = pend[q]; wv[0].n = pend_n[q]; wv[0].seed = q; wv[0].until = now + 1.0 + pend_n[q] / 64.0; pend[q] = NULL; ++taken; }
                    else { free(pend[q]); pend[q] = NULL; ++regrown; }
                }
                if (!wv[0].busy && !inflight[q]) {
                    const int n = grow(q, used, mine, ++id, tmpreg); work += n;
                    wv[0].busy = 1; wv[0].px = malloc(sizeof(int32_t) * n); memcpy(wv[0].px, tmpreg, sizeof(int32_t) * n); wv[0].n = n; wv[0].seed = q; wv[0].until = now + n + C0; ++self;
                }
            }
            /* the speculators */
            for (int w = 1; w < NW; ++w) if (!wv[w].busy) {
                int pick = -1;
                for (int p = scan + 1; p < n_order && p - scan < LOOK; ++p) {
                    const int q = order[p]; if (used[q] || pend[q] || inflight[q]) continue;
                    int ok = 1;
                    for (int v = 0; v < NW && ok; ++v) if (wv[v].busy && !(v == 0 && wv[0].px && wv[0].until - now < 2)) { int dx = abs(q % W_ - wv[v].seed % W_), dy = abs(q / W_ - wv[v].seed / W_); ok = (dx > dy ? dx : dy) >= SEP; }
                    if (ok) { pick = q; break; }
                }
                if (pick < 0) continue;
                const int n = grow(pick, used, mine, ++id, tmpreg); work += n;
                wv[w].busy = 1; wv[w].seed = pick; inflight[pick] = 1; wv[w].px = malloc(sizeof(int32_t) * n); memcpy(wv[w].px, tmpreg, sizeof(int32_t) * n); wv[w].n = n; wv[w].until = now + n + C0;
            }
            /* advance to the next completion */
            double nxt = 1e300; for (int w = 0; w < NW; ++w) if (wv[w].busy && wv[w].until < nxt) nxt = wv[w].until;
            if (nxt == 1e300) { printf("deadlock\n"); return 1; }
            now = nxt;
        }
        printf("W %2d  SEP %2d  LOOK %5d : time %8.0f (%.2fx faster)  work grown %.2fx  | committer: took %lld pending, grew %lld itself, regrew %lld; %lld speculative regions swallowed; regions %lld px %lld %s\n",
               NW, SEP, LOOK, now, seq_time / now, work / (double)seq_px, taken, self, regrown, wasted_regions, regions, px, regions == seq_regions && px == seq_px ? "(= sequential)" : "(DIFFERS)");
    }
    return 0;
}
