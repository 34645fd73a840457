/* lsd_resolve_model.c — CPU model of the speculate-and-verify resolution of lsd_grow_kernel (stvo-pl_amd/csrc/lsd_kernels.hip,
 * LSD_RESOLVE 1): the candidates of a sub-group (lane = 9 x region point + neighbour, 63 lanes) are not taken one after the other with the
 * region angle updated in between; instead
 *     A  = the lanes aligned with the angle at the start of the sub-group                                (guess)
 *     repeat: A' = A without the later lanes of a pixel that a lower lane of A holds (walked from the lowest lane up)
 *             lane k: sums_k = start sums + the (cos, sin) of A's lanes below k, ADDED IN LANE ORDER;  angle_k = fastAtan2(sums_k) (the start
 *             angle if no lane of A' is below k);  dup_k = a lane of A' below k holds the same pixel;  D = lanes aligned with angle_k and not dup
 *             m = lowest lane where D and A' differ: lanes below m are final, lane m's true decision is D[m];  A = A' below m | D from m on
 *     until D == A'
 * which is the sequential result exactly (induction over the lanes: lane k's sums are the sequential ones whenever A is right below k).
 * This program grows every region of a synthetic KITTI-size scene both ways (the sequential loop of the oracle and the lane model with the
 * kernel's round structure: up to 14 region points per round, flags read at the start of the round) and compares the pixel lists element
 * by element; it also counts the verification passes.
 *   gcc -O2 -o /tmp/lsd_model tools/experiments/lsd_resolve_model.c oracle/stvo_lsd_oracle.c oracle/stvo_orb_oracle.c -lm && /tmp/lsd_model */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

float orc_fast_atan2(float y, float x);
void orc_sincos_det(double x, double* s, double* c);
void orc_lsd_kernel7(double sigma, int32_t* ki);
void orc_resize_linear(const uint8_t* src, int scols, int srows, uint8_t* dst, int dcols, int drows);

#define PI 3.14159265358979323846
#define DEG2RAD (PI / 180)
static int W, H;
static float* angd;   /* degrees, < 0: undefined (the kernel's array) */
static float* csn;
static uint8_t* used;

static int aligned_d(double theta, double a, double prec) {
    double d = theta - a; if (d < 0) d = -d;
    if (d > 1.5 * PI) { d -= 2 * PI; if (d < 0) d = -d; }
    return d <= prec;
}
/* the oracle's loop (marks `used`) */
static int grow_seq(int seed, int32_t* reg, double prec, double* angle_out) {
    int n = 0;
    double ra = (double)angd[seed] * DEG2RAD, s, c;
    orc_sincos_det(ra, &s, &c);
    float sx = (float)c, sy = (float)s;
    reg[n++] = seed; used[seed] = 1;
    for (int i = 0; i < n; ++i) {
        const int px = reg[i] % W, py = reg[i] / W;
        for (int yy = py - 1; yy <= py + 1; ++yy) for (int xx = px - 1; xx <= px + 1; ++xx) {
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            const int q = yy * W + xx;
            if (!used[q] && angd[q] >= 0.f && aligned_d(ra, (double)angd[q] * DEG2RAD, prec)) {
                used[q] = 1; reg[n++] = q;
                sx += csn[2 * q]; sy += csn[2 * q + 1];
                ra = (double)orc_fast_atan2(sy, sx) * DEG2RAD;
            }
        }
    }
    *angle_out = ra;
    return n;
}
static long long g_rounds, g_pred_ok, g_pref_used, g_pref_bad;
static long long g_passes, g_subgroups, g_flips, g_accepts;
/* the lane model (marks `used`) */
static int grow_model(int seed, int32_t* reg, double prec, double* angle_out) {
    enum { GR = 2 };
    int n_reg = 1;
    double reg_angle = (double)angd[seed] * DEG2RAD, s0, c0;
    orc_sincos_det(reg_angle, &s0, &c0);
    float sumdx = (float)c0, sumdy = (float)s0;
    reg[0] = seed; used[seed] = 1;
    /* early loads for the NEXT round (issued right after the prediction, before this round is resolved): pixel and flag per lane */
    int have_pref = 0, pref_i = 0, pref_cnt = 0, pref_q[GR][64], pref_u[GR][64];
    for (int i = 0; i < n_reg;) {
        const int cnt = n_reg - i < 7 * GR ? n_reg - i : 7 * GR;
        int qq[GR][64]; float cx[GR][64], cy[GR][64]; double ad[GR][64]; int cand[GR][64], uu[GR][64];
        for (int r = 0; r < GR; ++r) for (int lane = 0; lane < 64; ++lane) {
            const int slot0 = lane / 9, nb = lane - slot0 * 9, slot = 7 * r + slot0;
            int valid = lane < 63 && slot < cnt && nb != 4;
            int pq = valid ? reg[i + slot] : 0;
            const int xx = pq % W + (nb % 3) - 1, yy = pq / W + nb / 3 - 1;
            valid = valid && xx >= 0 && xx < W && yy >= 0 && yy < H;
            qq[r][lane] = valid ? yy * W + xx : -1;
            int u = 1; float a = -1.f; cx[r][lane] = cy[r][lane] = 0.f;
            if (valid) { u = used[qq[r][lane]]; a = angd[qq[r][lane]]; cx[r][lane] = csn[2 * qq[r][lane]]; cy[r][lane] = csn[2 * qq[r][lane] + 1]; }
            uu[r][lane] = u;
            cand[r][lane] = valid && u == 0 && a >= 0.f;
            ad[r][lane] = (double)a * DEG2RAD;
        }
        if (have_pref) {  /* what the early loads + the correction hold must be what the fetch of this round reads */
            ++g_pref_used;
            int bad = pref_i != i || pref_cnt != cnt;
            for (int r = 0; r < GR && !bad; ++r) for (int l = 0; l < 64; ++l) if (pref_q[r][l] != qq[r][l] || (qq[r][l] >= 0 && pref_u[r][l] != uu[r][l])) bad = 1;
            g_pref_bad += bad;
        }
        /* could the NEXT round's loads be issued before this round is resolved?  prediction from the state at the start of the round: the
         * first lane of every pixel aligned with the start angle, sub-group after sub-group */
        int pred[128], n_pred = 0;
        for (int r = 0; r < GR && 7 * r < cnt; ++r) for (int l = 0; l < 64; ++l) if (cand[r][l] && aligned_d(reg_angle, ad[r][l], prec)) {
            int dupe = 0; for (int t = 0; t < n_pred; ++t) dupe |= pred[t] == qq[r][l];
            if (!dupe) pred[n_pred++] = qq[r][l];
        }
        const int n_before = n_reg;
        int early_q[GR][64], early_u[GR][64];
        const int next_i = i + cnt, next_n = n_reg + n_pred, next_cnt = next_n - next_i < 7 * GR ? next_n - next_i : 7 * GR;
        for (int r = 0; r < GR; ++r) for (int lane = 0; lane < 64; ++lane) {  /* the points of the next round: queued list entries, then the prediction */
            const int slot0 = lane / 9, nb = lane - slot0 * 9, slot = 7 * r + slot0;
            int valid = lane < 63 && slot < next_cnt && nb != 4;
            const int idx = next_i + slot;
            const int pq = valid ? (idx < n_reg ? reg[idx] : pred[idx - n_reg]) : 0;
            const int xx = pq % W + (nb % 3) - 1, yy = pq / W + nb / 3 - 1;
            valid = valid && xx >= 0 && xx < W && yy >= 0 && yy < H;
            early_q[r][lane] = valid ? yy * W + xx : -1;
            early_u[r][lane] = valid ? used[early_q[r][lane]] : 1;   /* the flags as they are NOW: this round's acceptances are not in them */
        }
        for (int r = 0; r < GR; ++r) {
            if (7 * r >= cnt) break;
            uint64_t A = 0, acc = 0;
            for (int l = 0; l < 64; ++l) if (cand[r][l] && aligned_d(reg_angle, ad[r][l], prec)) A |= 1ull << l;  /* every aligned lane: the pass keeps the first lane of a pixel */
            ++g_subgroups;
            if (A) for (;;) {
                ++g_passes;
                /* the kernel's pass: walk the guess from the lowest lane, a lane taken removes the later lanes of the same pixel from the guess */
                float sx[64], sy[64]; int any[64] = {0}, dup[64] = {0};
                for (int k = 0; k < 64; ++k) { sx[k] = sumdx; sy[k] = sumdy; }
                uint64_t rem = A; acc = 0;
                while (rem) {
                    const int j = __builtin_ctzll(rem);
                    acc |= 1ull << j;
                    for (int k = 0; k < 64; ++k) {
                        const int same = qq[r][k] == qq[r][j];
                        if (same) rem &= ~(1ull << k);
                        if (k > j) { sx[k] += cx[r][j]; sy[k] += cy[r][j]; any[k] = 1; dup[k] |= same; }
                    }
                }
                uint64_t D = 0;
                for (int k = 0; k < 64; ++k) {
                    const double th = any[k] ? (double)orc_fast_atan2(sy[k], sx[k]) * DEG2RAD : reg_angle;
                    if (cand[r][k] && !dup[k] && aligned_d(th, ad[r][k], prec)) D |= 1ull << k;
                }
                if (D == acc) break;
                ++g_flips;
                const int m = __builtin_ctzll(D ^ acc);
                const uint64_t below = (1ull << m) - 1ull;
                A = (acc & below) | (D & ~below);
                if (!A) { printf("empty guess after a flip\n"); return -1; }
            }
            A = acc;
            /* apply: in lane order */
            for (int l = 0; l < 64; ++l) if (A >> l & 1) {
                used[qq[r][l]] = 1; reg[n_reg++] = qq[r][l]; sumdx += cx[r][l]; sumdy += cy[r][l]; ++g_accepts;
                for (int r2 = r + 1; r2 < GR; ++r2) for (int l2 = 0; l2 < 64; ++l2) if (qq[r2][l2] == qq[r][l]) cand[r2][l2] = 0;
            }
            if (A) reg_angle = (double)orc_fast_atan2(sumdy, sumdx) * DEG2RAD;
        }
        ++g_rounds;
        { int same = n_reg - n_before == n_pred; for (int t = 0; t < n_pred && same; ++t) same = reg[n_before + t] == pred[t]; g_pred_ok += same;
          have_pref = same && next_cnt > 0;
          if (have_pref) {
              pref_i = next_i; pref_cnt = next_cnt;
              for (int r = 0; r < GR; ++r) for (int l = 0; l < 64; ++l) {
                  pref_q[r][l] = early_q[r][l]; pref_u[r][l] = early_u[r][l];
                  for (int t = 0; t < n_pred; ++t) if (early_q[r][l] == pred[t]) pref_u[r][l] = 1;  /* the correction: pixels this round took */
              }
          } }
        i += cnt;
    }
    *angle_out = reg_angle;
    return n_reg;
}

int main(int argc, char** argv) {
    const double noise = argc > 1 ? atof(argv[1]) : 3.0;  /* standard deviation of the grey-level noise: try 12 and 30 for many small regions */
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
    for (int i = 0; i < npx0; ++i) { double n = -6; for (int t = 0; t < 12; ++t) n += (RND() % 10000) / 10000.0; double v = f[i] + noise * n; img0[i] = v < 0 ? 0 : v > 255 ? 255 : (uint8_t)lrint(v); }
    { int32_t ki[7]; orc_lsd_kernel7(0.6, ki); int32_t* tmp = malloc(sizeof(int32_t) * npx0); uint8_t* bl = malloc(npx0);
#define R101(p, n) ((p) < 0 ? -(p) : ((p) >= (n) ? 2 * (n) - 2 - (p) : (p)))
      for (int y = 0; y < H0; ++y) for (int x = 0; x < W0; ++x) { int a = 0; for (int i = 0; i < 7; ++i) a += ki[i] * img0[y * W0 + R101(x + i - 3, W0)]; tmp[y * W0 + x] = a; }
      for (int y = 0; y < H0; ++y) for (int x = 0; x < W0; ++x) { int a = 0; for (int i = 0; i < 7; ++i) a += ki[i] * tmp[R101(y + i - 3, H0) * W0 + x]; a = (a + (1 << 15)) >> 16; bl[y * W0 + x] = a < 0 ? 0 : a > 255 ? 255 : a; }
      orc_resize_linear(bl, W0, H0, img, W, H); free(tmp); free(bl); }
    angd = malloc(sizeof(float) * npx); csn = calloc(2 * npx, sizeof(float));
    double* mod = calloc(npx, sizeof(double));
    const double prec = PI * 22.5 / 180, rho = 2.0 / sin(prec);
    double mx = -1;
    for (int i = 0; i < npx; ++i) angd[i] = -1.f;
    for (int y = 0; y < H - 1; ++y) for (int x = 0; x < W - 1; ++x) {
        int DA = img[(y + 1) * W + x + 1] - img[y * W + x], BC = img[y * W + x + 1] - img[(y + 1) * W + x], gx = DA + BC, gy = DA - BC;
        double nrm = sqrt((gx * gx + gy * gy) / 4.0); mod[y * W + x] = nrm;
        if (!(nrm <= rho)) { int q = y * W + x; angd[q] = orc_fast_atan2((float)gx, (float)-gy); double a = (double)angd[q] * DEG2RAD, s, c; orc_sincos_det((double)(float)a, &s, &c); csn[2 * q] = (float)c; csn[2 * q + 1] = (float)s; if (nrm > mx) mx = nrm; }
    }
    int32_t* order = malloc(sizeof(int32_t) * npx); int n_order = 0;
    { int* start = calloc(1026, sizeof(int)); double bc = 1023 / mx;
      for (int i = 0; i < npx; ++i) if (angd[i] >= 0.f) start[1023 - (int)(mod[i] * bc) + 1]++;
      for (int b = 0; b < 1024; ++b) start[b + 1] += start[b];
      for (int i = 0; i < npx; ++i) if (angd[i] >= 0.f) { order[start[1023 - (int)(mod[i] * bc)]++] = i; ++n_order; }
      free(start); }
    uint8_t* used_a = calloc(npx, 1); uint8_t* used_b = calloc(npx, 1);
    int32_t* ra = malloc(sizeof(int32_t) * npx); int32_t* rb = malloc(sizeof(int32_t) * npx);
    long long regions = 0, bad = 0;
    for (int p = 0; p < n_order; ++p) {
        const int seed = order[p];
        if (used_a[seed]) continue;
        double aa, ab;
        used = used_a; const int na = grow_seq(seed, ra, prec, &aa);
        used = used_b; const int nb = grow_model(seed, rb, prec, &ab);
        ++regions;
        if (na != nb || aa != ab || memcmp(ra, rb, sizeof(int32_t) * na)) { if (bad++ < 5) printf("region %lld (seed %d): %d vs %d pixels, angle %.17g vs %.17g\n", regions, seed, na, nb, aa, ab); }
    }
    printf("%lld regions, %lld differ; flags equal: %d\n", regions, bad, !memcmp(used_a, used_b, npx));
    printf("sub-groups %lld, with candidates accepted at the first guess or later: verification passes %lld, passes that ended in a flip %lld, pixels accepted %lld\n",
           g_subgroups, g_passes, g_flips, g_accepts);
    printf("rounds %lld, of which the accepted pixels (in order) equal the prediction made at the start of the round: %lld (%.1f %%)\n", g_rounds, g_pred_ok, 100.0 * g_pred_ok / g_rounds);
    printf("rounds that could run on loads issued one round early (+ the correction for the pixels taken meanwhile): %lld, of which the data differ from a fetch at the round's start: %lld\n", g_pref_used, g_pref_bad);
    return bad != 0 || g_pref_bad != 0;
}
