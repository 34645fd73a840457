/* lsd_waves_model.c — random-interleaving model of lsd_grow_waves_kernel's protocol (stvo-pl_amd/csrc/lsd_kernels.hip; the kernel had no
 * GPU time at the end of round 4).  Every wave is a small state machine; a seeded scheduler runs ONE step of a random runnable wave at a
 * time, so growth, validation and commits interleave at the finest grain the kernel allows:
 *   committer   per seed in rank order: publish the position -> look at the table, then at the in-flight set (wait while the seed is in
 *               flight, look at the table once more otherwise) -> validate a pending region pixel chunk by pixel chunk -> commit it chunk
 *               by chunk (64 flags per step) or grow the seed itself, one region point per step, flags set as it goes
 *   speculator  pick under the lock (first free, not pending, not in flight, separated seed after the committer's position) -> grow one
 *               region point per step against the flags AS THEY ARE AT THAT STEP, own pixels in a stamp array -> record, then table entry,
 *               then leave the in-flight set (three separate steps, in the kernel's order)
 * and the committed regions (seed, pixel list) are compared with the sequential search.  What this checks: the validation argument
 * under arbitrary interleavings and the ordering of the table / in-flight updates; what it cannot check: the HIP code itself.
 *   gcc -O2 -o /tmp/lsd_wmodel tools/experiments/lsd_waves_model.c oracle/stvo_lsd_oracle.c oracle/stvo_orb_oracle.c -lm && /tmp/lsd_wmodel [runs [x]]
 * (a second argument switches the validation off: the negative control — those runs must differ from the sequential search) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

float orc_fast_atan2(float y, float x);
void orc_sincos_det(double x, double* s, double* c);

#define PI 3.14159265358979323846
#define NOTDEF (-1024.0)
enum { W = 360, H = 240, NPX = W * H, NW = 6, SEP = 10, LOOK = 512 };
static double ang[NPX], mod_[NPX];
static float csn[2 * NPX];
static int order[NPX], n_order;
static const double PREC = PI * 22.5 / 180;
static uint32_t rs;
#define RND() (rs = rs * 1664525u + 1013904223u, (rs >> 8) & 0xFFFFFF)

static int aligned(int q, double theta) {
    const double a = ang[q];
    if (a == NOTDEF) return 0;
    double d = fabs(theta - a);
    if (d > 1.5 * PI) d = fabs(d - 2 * PI);
    return d <= PREC;
}
/* one growth in progress: the sequential loop of the oracle, resumable after every region point */
typedef struct { int seed, n, i; double ra; float sx, sy; int* px; int cap; } Grow;
static void grow_begin(Grow* g, int seed) {
    double s, c;
    g->seed = seed; g->n = 0; g->i = 0; g->ra = ang[seed];
    orc_sincos_det(g->ra, &s, &c);
    g->sx = (float)c; g->sy = (float)s;
    g->px[g->n++] = seed;
}
/* processes region point i; `taken(q)` = flag || own mark.  mark = 1: set used[] (committer); else stamp[] = id.  returns 1 when finished */
static int grow_step(Grow* g, uint8_t* used, int32_t* stamp, int id, int mark) {
    if (g->i >= g->n) return 1;
    const int px = g->px[g->i] % W, py = g->px[g->i] / W;
    for (int yy = py - 1; yy <= py + 1; ++yy) for (int xx = px - 1; xx <= px + 1; ++xx) {
        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
        const int q = yy * W + xx;
        const int taken = used[q] || (!mark && stamp[q] == id);
        if (!taken && aligned(q, g->ra)) {
            if (mark) used[q] = 1; else stamp[q] = id;
            g->px[g->n++] = q;
            g->sx += csn[2 * q]; g->sy += csn[2 * q + 1];
            g->ra = orc_fast_atan2(g->sy, g->sx) * (PI / 180);
        }
    }
    ++g->i;
    return g->i >= g->n;
}

typedef struct { int seed, n; int* px; } Region;
static Region* seq_regions; static int n_seq;

static void make_scene(uint32_t seed, double noise) {
    static uint8_t img[NPX];
    static double f[NPX];
    rs = seed;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) f[y * W + x] = 110 + 25 * sin(x * 3.1 / W) + 18 * cos(y * 2.3 / H);
    for (int r = 0; r < 90; ++r) {
        int w = 6 + RND() % 60, h = 6 + RND() % 50, x0 = (int)(RND() % (W + 20)) - 20, y0 = (int)(RND() % (H + 20)) - 20;
        double v = 15 + RND() % 225;
        for (int y = y0 < 0 ? 0 : y0; y < y0 + h && y < H; ++y) for (int x = x0 < 0 ? 0 : x0; x < x0 + w && x < W; ++x) f[y * W + x] = v;
    }
    for (int i = 0; i < NPX; ++i) { double n = -6; for (int t = 0; t < 12; ++t) n += (RND() % 10000) / 10000.0; double v = f[i] + noise * n; img[i] = v < 0 ? 0 : v > 255 ? 255 : (uint8_t)lrint(v); }
    const double rho = 2.0 / sin(PREC);
    double mx = -1;
    for (int i = 0; i < NPX; ++i) { ang[i] = NOTDEF; mod_[i] = 0; csn[2 * i] = csn[2 * i + 1] = 0; }
    for (int y = 0; y < H - 1; ++y) for (int x = 0; x < W - 1; ++x) {
        int DA = img[(y + 1) * W + x + 1] - img[y * W + x], BC = img[y * W + x + 1] - img[(y + 1) * W + x], gx = DA + BC, gy = DA - BC;
        double nrm = sqrt((gx * gx + gy * gy) / 4.0); mod_[y * W + x] = nrm;
        if (nrm > rho) { int q = y * W + x; ang[q] = orc_fast_atan2((float)gx, (float)-gy) * (PI / 180); double s, c; orc_sincos_det((double)(float)ang[q], &s, &c); csn[2 * q] = (float)c; csn[2 * q + 1] = (float)s; if (nrm > mx) mx = nrm; }
    }
    static int start[1026];
    memset(start, 0, sizeof(start)); n_order = 0;
    const double bc = 1023 / mx;
    for (int i = 0; i < NPX; ++i) if (ang[i] != NOTDEF) start[1023 - (int)(mod_[i] * bc) + 1]++;
    for (int b = 0; b < 1024; ++b) start[b + 1] += start[b];
    for (int i = 0; i < NPX; ++i) if (ang[i] != NOTDEF) { order[start[1023 - (int)(mod_[i] * bc)]++] = i; ++n_order; }
}

int main(int argc, char** argv) {
    const int runs = argc > 1 ? atoi(argv[1]) : 6;
    const int no_validation = argc > 2;  /* negative control: take every pending region unchecked -> the runs must come out DIFFERENT */
    static uint8_t used[NPX];
    static int32_t stamp[NW][NPX];
    static int pend_wave[NPX], pend_idx[NPX];   /* by rank: -1 or the (wave, record) of a finished region */
    int bad_total = 0;
    for (int run = 0; run < runs; ++run) {
        const double noise = (run / 3) % 2 ? 4.0 : 1.5;               /* few long regions / many small ones */
        const int c_share = run % 3 == 0 ? NW : (run % 3 == 1 ? 2 : 24);  /* the committer runs 1 step in c_share: as often as a speculator, every other step, rarely */
        make_scene(1000u + 77u * (run / 3), noise);
        /* sequential reference */
        memset(used, 0, NPX);
        seq_regions = malloc(sizeof(Region) * NPX); n_seq = 0;
        { Grow g; g.px = malloc(sizeof(int) * NPX);
          for (int p = 0; p < n_order; ++p) { const int q = order[p]; if (used[q]) continue; used[q] = 1; grow_begin(&g, q); while (!grow_step(&g, used, NULL, 0, 1)) {}
              seq_regions[n_seq].seed = q; seq_regions[n_seq].n = g.n; seq_regions[n_seq].px = malloc(sizeof(int) * g.n); memcpy(seq_regions[n_seq].px, g.px, sizeof(int) * g.n); ++n_seq; }
          free(g.px); }
        /* the model */
        memset(used, 0, NPX); memset(stamp, 0, sizeof(stamp));
        for (int i = 0; i < NPX; ++i) pend_wave[i] = -1;
        Region* recs[NW]; int nrec[NW]; for (int w = 0; w < NW; ++w) { recs[w] = malloc(sizeof(Region) * NPX); nrec[w] = 0; }
        int if_rank[NW], if_seed[NW]; for (int w = 0; w < NW; ++w) if_rank[w] = if_seed[w] = -1;
        int s_scan = -1, s_done = 0, lock = -1;
        Grow gw[NW]; for (int w = 0; w < NW; ++w) { gw[w].px = malloc(sizeof(int) * NPX); }
        int ids[NW] = {0};
        /* committer state */
        enum { C_NEXT, C_LOOK, C_LOOK2, C_WAIT, C_VALIDATE, C_COMMIT, C_SELF } cs = C_NEXT;
        int c_rank = -1, c_seed = -1, c_pw = -1, c_pi = -1, c_t = 0, c_ok = 1;
        /* speculator state */
        enum { S_IDLE, S_LOCKED, S_GROW, S_RECORD, S_TABLE, S_LEAVE } ss[NW]; for (int w = 0; w < NW; ++w) ss[w] = S_IDLE;
        int s_rank[NW];
        Region* out = malloc(sizeof(Region) * NPX); int n_out = 0;
        rs = 4242u + 9973u * run;
        long long steps = 0, taken = 0, self = 0, regrown = 0, waits = 0;
        while (!s_done) {
            const int w = RND() % c_share == 0 ? 0 : 1 + RND() % (NW - 1); ++steps;
            if (w == 0) {
                switch (cs) {
                case C_NEXT:
                    ++c_rank;
                    while (c_rank < n_order && used[order[c_rank]]) ++c_rank;
                    if (c_rank >= n_order) { s_done = 1; break; }
                    c_seed = order[c_rank]; s_scan = c_rank; if_seed[0] = c_seed; cs = C_LOOK; break;
                case C_LOOK:   /* the table first */
                    if (pend_wave[c_rank] >= 0) { c_pw = pend_wave[c_rank]; c_pi = pend_idx[c_rank]; c_t = 0; c_ok = 1; cs = C_VALIDATE; } else cs = C_LOOK2;
                    break;
                case C_LOOK2: { /* then the in-flight set */
                    int inflight = 0; for (int v = 1; v < NW; ++v) inflight |= if_rank[v] == c_rank;
                    if (inflight) { cs = C_WAIT; ++waits; }
                    else if (pend_wave[c_rank] >= 0) { c_pw = pend_wave[c_rank]; c_pi = pend_idx[c_rank]; c_t = 0; c_ok = 1; cs = C_VALIDATE; }
                    else { used[c_seed] = 1; grow_begin(&gw[0], c_seed); cs = C_SELF; ++self; }
                    break; }
                case C_WAIT:
                    if (pend_wave[c_rank] >= 0) { c_pw = pend_wave[c_rank]; c_pi = pend_idx[c_rank]; c_t = 0; c_ok = 1; cs = C_VALIDATE; }
                    break;
                case C_VALIDATE: { /* 64 pixels per step */
                    const Region* r = &recs[c_pw][c_pi];
                    for (int k = 0; k < 64 && c_t < r->n; ++k, ++c_t) c_ok &= no_validation || !used[r->px[c_t]];
                    if (c_t >= r->n) { if (c_ok) { c_t = 0; cs = C_COMMIT; ++taken; } else { used[c_seed] = 1; grow_begin(&gw[0], c_seed); cs = C_SELF; ++regrown; } }
                    break; }
                case C_COMMIT: {
                    const Region* r = &recs[c_pw][c_pi];
                    for (int k = 0; k < 64 && c_t < r->n; ++k, ++c_t) used[r->px[c_t]] = 1;
                    if (c_t >= r->n) { out[n_out].seed = c_seed; out[n_out].n = r->n; out[n_out].px = r->px; ++n_out; cs = C_NEXT; }
                    break; }
                case C_SELF:
                    if (grow_step(&gw[0], used, NULL, 0, 1)) { out[n_out].seed = c_seed; out[n_out].n = gw[0].n; out[n_out].px = malloc(sizeof(int) * gw[0].n); memcpy(out[n_out].px, gw[0].px, sizeof(int) * gw[0].n); ++n_out; cs = C_NEXT; }
                    break;
                }
                continue;
            }
            switch (ss[w]) {
            case S_IDLE: if (lock < 0) { lock = w; ss[w] = S_LOCKED; } break;
            case S_LOCKED: {
                int pick = -1;
                for (int r = s_scan + 1; r < n_order && r - s_scan <= LOOK; ++r) {
                    const int q = order[r]; if (used[q] || pend_wave[r] >= 0) continue;
                    int ok = 1;
                    for (int v = 0; v < NW && ok; ++v) if (if_seed[v] >= 0) { int dx = abs(q % W - if_seed[v] % W), dy = abs(q / W - if_seed[v] / W); ok = (dx > dy ? dx : dy) >= SEP; }
                    if (ok) { pick = r; break; }
                }
                if (pick >= 0) { if_rank[w] = pick; if_seed[w] = order[pick]; s_rank[w] = pick; ++ids[w]; stamp[w][order[pick]] = ids[w]; grow_begin(&gw[w], order[pick]); ss[w] = S_GROW; } else ss[w] = S_IDLE;
                lock = -1;
                break; }
            case S_GROW: if (grow_step(&gw[w], used, stamp[w], ids[w], 0)) ss[w] = S_RECORD; break;
            case S_RECORD: { Region* r = &recs[w][nrec[w]]; r->seed = gw[w].seed; r->n = gw[w].n; r->px = malloc(sizeof(int) * gw[w].n); memcpy(r->px, gw[w].px, sizeof(int) * gw[w].n); ss[w] = S_TABLE; break; }
            case S_TABLE: pend_idx[s_rank[w]] = nrec[w]; pend_wave[s_rank[w]] = w; ++nrec[w]; ss[w] = S_LEAVE; break;
            case S_LEAVE: if_rank[w] = if_seed[w] = -1; ss[w] = S_IDLE; break;
            }
        }
        int bad = n_out != n_seq;
        for (int k = 0; k < n_out && k < n_seq && !bad; ++k) bad = out[k].seed != seq_regions[k].seed || out[k].n != seq_regions[k].n || memcmp(out[k].px, seq_regions[k].px, sizeof(int) * out[k].n);
        printf("run %d (noise %.1f, committer 1 step in %d): %d regions (sequential %d) %s | %lld steps, committer took %lld pending, grew %lld itself, regrew %lld, waited %lld times\n", run, noise, c_share, n_out, n_seq,
               bad ? "DIFFERENT" : "identical", steps, taken, self, regrown, waits);
        bad_total += bad;
    }
    return bad_total != 0;
}
