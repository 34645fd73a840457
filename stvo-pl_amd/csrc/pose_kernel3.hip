// pose_kernel3.hip — K4/K5/K6, third formulation (batches): StereoFrameHandler::optimizePose
// (/root/reference/src/stereoFrameHandler.cpp:307-392) for B frame pairs with TWO frame pairs per workgroup and the roles
// of a frame pair's work split between waves, so that the serial sections of one pair overlap the evaluations of the other.
//
// What the measurements of pose_kernel2.hip said (NOTES.md, round 2): a frame pair is a chain of ~13.7 evaluate -> solve
// rounds; the solve (6x6 LDL^T, SE(3) update, stop tests) is one wave's dependent FP64 chain during which the pair's other
// waves idle, only two pairs fit a CU with their records in LDS, and two co-resident workgroups run in lock step.  Here one
// workgroup per CU holds the records of two frame pairs in LDS and has
//   * two OWNER waves (wave p owns pair slot p): each runs the whole optimizePose state machine of its current pair as a
//     plain sequential program — staging, the serial 6x6 algebra, every data-dependent decision, removeOutliers (median / MAD
//     by exact selection inside the wave), commit — and loops over its pairs (pair f = 2 g + p + k 2 G of workgroup g of G);
//   * NW - 2 EVALUATOR waves shared by both pairs: they serve optimizeFunctions[Robust] jobs (pose, scales, inlier bits in
//     LDS -> 28 partial sums per wave), whichever owner posts one.  While owner A solves, the evaluators work for owner B.
// Owners and evaluators meet through LDS mailboxes (sequence number / completion counter, workgroup-scope release / acquire)
// and s_sleep polling; there is no s_barrier after the prologue, so the two pairs never wait for each other.  Every spin is
// bounded: on a protocol failure the workgroup aborts and reports STVO_POSE_INTERNAL in the status of its pairs.
// The sums are reduced in a fixed order (evaluator lanes own fixed record slots, wave partials are added in wave order), so
// results are bit-reproducible run to run and independent of which owner was served first.
// Records that do not fit the pair's LDS share (or observations that are not exactly representable in the compact format)
// make the pair a MISFIT: it is appended to a list and solved by pose_kernel2.hip's kernel right behind this launch.
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "pose_block.h"

namespace stvo {
namespace {

constexpr int P3_MAXP = STVO_POSE_MAX_POINTS, P3_MAXL = STVO_POSE_MAX_LINES;
constexpr int P3_PPT = P3_MAXP / 64, P3_LPT = P3_MAXL / 64;  // slots per owner lane
constexpr int P3_SPIN_LIMIT = 1 << 24;                        // polls before a wait gives up (~seconds)
constexpr int P3_SEQ_EXIT = -1;
constexpr bool POSE2_PRIO_P3 = true;  // owner waves at wave priority 3: their serial chains share the SIMDs with evaluating waves

struct P3Job {  // owner -> evaluators, one per pair slot
    double DT[12];
    double sp, sl;
    double fx, fy, cx, cy;
    int robust, n_p, n_l, pad;
    int seq;   // release-stored last: a new value = a new job; P3_SEQ_EXIT = the owner has no more pairs
    int done;  // evaluator waves that finished job `seq`
};

__device__ __forceinline__ double uni3(double v) {  // wave-uniform value -> SGPR pair
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xFFFFFFFFll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ int ld_acq(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_rel(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wave_sync_lds() {  // LDS written by some lanes of this wave, read by others
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// k-th smallest (0-based) of the keys {key(k) : bit k of mask} of ONE wave (lane L, index k = slot k * 64 + L):
// most-significant-first radix selection on eight bits per round, counts in a 256-bin LDS histogram (no-return ds_add), the bins
// scanned four per lane.  key(k) is recomputed from the lane's value registers in every round (a few VALU ops) instead of being
// kept: the values themselves already take 64 VGPRs.  Integer counts only — the result is the one std::sort would leave at
// position kth (src/auxiliar.cpp:395-404).  As soon as exactly one key is left under the prefix it is fetched directly.
template <int N, typename K, int BITS, typename KeyFn>
__device__ __forceinline__ K wave_select_kth(KeyFn key, const unsigned mask, const int kth, int* hist) {
    const int lane = threadIdx.x & 63;
    K prefix = 0;
    int kk = kth;
#pragma unroll 1
    for (int shift = BITS - 8; shift >= 0; shift -= 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) hist[lane + 64 * i] = 0;
        wave_sync_lds();
        const int up = shift + 8 >= BITS ? 0 : shift + 8;
        const bool top = shift + 8 >= BITS;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const K kv = key(k);
            const bool same = top ? true : ((kv ^ prefix) >> up) == 0;
            if (((mask >> k) & 1u) && same) atomicAdd(&hist[(int)((kv >> shift) & 255)], 1);
        }
        wave_sync_lds();
        const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        const int sum = c0 + c1 + c2 + c3;
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int excl = incl - sum;
        int bin = 4 * lane, below = excl, cnt = c0;
        if (kk >= below + cnt) { below += cnt; cnt = c1; bin += 1;
            if (kk >= below + cnt) { below += cnt; cnt = c2; bin += 1;
                if (kk >= below + cnt) { below += cnt; cnt = c3; bin += 1; } } }
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(excl <= kk && kk < incl);  // exactly one lane when 0 <= kk < #keys
        const int src = hit ? __builtin_ctzll(hit) : 0;
        bin = __shfl(bin, src, 64);
        below = __shfl(below, src, 64);
        cnt = __shfl(cnt, src, 64);
        wave_sync_lds();  // the bins are cleared again next round
        prefix |= (K)bin << shift;
        kk -= below;
        if (cnt == 1 && shift > 0) {  // wave-uniform: the single key under the prefix
            K found = 0;
            bool have = false;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const K kv = key(k);
                if (((mask >> k) & 1u) && ((kv ^ prefix) >> shift) == 0) {
                    found = kv;
                    have = true;
                }
            }
            const unsigned long long who = __builtin_amdgcn_ballot_w64(have);
            const int s2 = who ? __builtin_ctzll(who) : 0;
            if (sizeof(K) == 8) {
                const unsigned long long f = (unsigned long long)found;
                const unsigned lo = (unsigned)__shfl((int)(unsigned)f, s2, 64), hi = (unsigned)__shfl((int)(unsigned)(f >> 32), s2, 64);
                return (K)(((unsigned long long)hi << 32) | lo);
            }
            return (K)(unsigned)__shfl((int)(unsigned)found, s2, 64);
        }
    }
    return prefix;
}

// 1.4826 * MAD of the n values {v[k] : bit k of mask} of one wave (vector_stdv_mad, src/auxiliar.cpp:395-404,447-457):
// median = sorted[n / 2]; dev = fabsf(x - median) (FLOAT truncation); MAD = sorted dev[n / 2].  n == 0 -> 0.
template <int N>
__device__ __forceinline__ double wave_mad_sigma(const double* v, const unsigned mask, const int n, int* hist) {
    if (n == 0) return 0.0;  // wave-uniform
    const int kth = n / 2;
    auto dkey = [&](int k) -> unsigned long long {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v[k]);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);  // total order of IEEE doubles
    };
    const unsigned long long res = wave_select_kth<N, unsigned long long, 64>(dkey, mask, kth, hist);
    const unsigned long long mb = (res >> 63) ? (res & 0x7FFFFFFFFFFFFFFFull) : ~res;
    const double median = __longlong_as_double((long long)mb);
    auto fkey = [&](int k) -> unsigned { return __float_as_uint(fabsf((float)(v[k] - median))); };  // >= 0 (or NaN)
    const unsigned fres = wave_select_kth<N, unsigned, 32>(fkey, mask, kth, hist);
    return 1.4826 * (double)__uint_as_float(fres);
}

__device__ __forceinline__ double wave_sum_d(double v) {  // butterfly: every lane ends with the total (fixed order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// NW waves per workgroup: 2 owners + NW - 2 evaluators.  ROW: 6x6 systems on rows (128-VGPR budget, NW = 16) or with the
// serial routines of pose_math.h (256 VGPRs, NW = 8).  COMPACT: observations (curr_pl) stored as two floats per point.
template <int NW, bool ROW, bool COMPACT>
__global__ __launch_bounds__(NW * 64, NW / 4) void pose3_kernel(const PoseArgs a_in, const int lds_pair_bytes, int* misfit_cnt, int* misfit_cnt_next,
                                                                int* misfit_list) {
    constexpr int NEV = NW - 2, NL = NEV * 64;
    using Ops = BlockOps<NW>;
    constexpr int OBS_BYTES = COMPACT ? 8 : 16;
    constexpr int REC_P = 32 + OBS_BYTES, REC_L = 112;
    extern __shared__ double s_dyn[];  // pair slot p: [lines 7 x double2][XY double2][ZQ double2][OBS]
    __shared__ PoseSh s_sh[2];
    __shared__ P3Job s_job[2];
    __shared__ double s_red[2][NEV][28];
    __shared__ unsigned s_inl_p[2][P3_MAXP / 32], s_inl_l[2][P3_MAXL / 32];
    __shared__ int s_hist[2][256];
    __shared__ int s_abort;
    // the launch arguments, read from LDS where they are needed: ~35 pointers kept in SGPRs across the owners' persistent loop
    // left the register allocator ~30 SGPRs for everything else (hundreds of spills)
    __shared__ PoseArgs s_a;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) {
        s_a = a_in;
        s_abort = 0;
        if (blockIdx.x == 0) *misfit_cnt_next = 0;  // the counter of the NEXT launch (this launch appends to misfit_cnt)
    }
    if (tid < 2) {
        s_job[tid].seq = 0;
        s_job[tid].done = 0;
    }
    __syncthreads();  // the only barrier of the kernel
    const PoseArgs& a = s_a;

    if (wv >= 2) {
        // =================================== evaluator waves ===================================
        const int ew = wv - 2, el = ew * 64 + lane;  // evaluator lane 0 .. NL-1
        int served[2] = {0, 0};
        int idle = 0;
        for (;;) {
            int exits = 0;
            bool any = false;
#pragma unroll 1
            for (int p = 0; p < 2; ++p) {
                const int sq = ld_acq(&s_job[p].seq);
                if (sq == P3_SEQ_EXIT) {
                    ++exits;
                    continue;
                }
                if (sq == served[p]) continue;
                any = true;
                served[p] = sq;
                const P3Job* job = &s_job[p];
                double DT[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) DT[i] = uni3(job->DT[i]);
                const pm::Cam5 cam{uni3(job->fx), uni3(job->fy), uni3(job->cx), uni3(job->cy)};
                const bool robust = __builtin_amdgcn_readfirstlane(job->robust) != 0;
                const double sp = uni3(job->sp), sl = uni3(job->sl), homog_th = uni3(a.prm.homog_th);
                const int n_p = __builtin_amdgcn_readfirstlane(job->n_p), n_l = __builtin_amdgcn_readfirstlane(job->n_l);
                const char* base = reinterpret_cast<const char*>(s_dyn) + (size_t)p * lds_pair_bytes;
                const double2* s_ln = reinterpret_cast<const double2*>(base);
                const double2* s_xy = reinterpret_cast<const double2*>(base + (size_t)n_l * REC_L);
                const double2* s_zq = s_xy + n_p;
                const char* s_ob = reinterpret_cast<const char*>(s_zq + n_p);
                double acc[28];
#pragma unroll
                for (int i = 0; i < 28; ++i) acc[i] = 0.0;
#pragma unroll 1
                for (int s = el; s < n_p; s += NL) {
                    if (!((s_inl_p[p][s >> 5] >> (s & 31)) & 1u)) continue;
                    const double2 xy = s_xy[s], zq = s_zq[s];
                    double ox, oy;
                    if (COMPACT) {
                        const float2 o = reinterpret_cast<const float2*>(s_ob)[s];
                        ox = (double)o.x;
                        oy = (double)o.y;
                    } else {
                        const double2 o = reinterpret_cast<const double2*>(s_ob)[s];
                        ox = o.x;
                        oy = o.y;
                    }
                    pm::point_term_q(acc, DT, cam, homog_th, xy.x, xy.y, zq.x, ox, oy, zq.y, robust, sp);
                }
                // lines are handed out from the top lanes (they own the fewest points; a line costs about two points)
#pragma unroll 1
                for (int l = NL - 1 - el; l < n_l; l += NL) {
                    if (!((s_inl_l[p][l >> 5] >> (l & 31)) & 1u)) continue;
                    const double2* q = s_ln + (size_t)l * 7;
                    const double2 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4], v5 = q[5], v6 = q[6];
                    pm::LineRec L;
                    L.sP[0] = v0.x; L.sP[1] = v0.y; L.sP[2] = v1.x; L.eP[0] = v1.y; L.eP[1] = v2.x; L.eP[2] = v2.y;
                    L.le[0] = v3.x; L.le[1] = v3.y; L.le[2] = v4.x; L.spl[0] = v4.y; L.spl[1] = v5.x; L.epl[0] = v5.y; L.epl[1] = v6.x;
                    L.sigma2 = v6.y;
                    pm::line_term_q(acc, DT, cam, homog_th, L, robust, sl);
                }
                // 28 values x 64 lanes -> 28 wave totals (reduce-scatter on the lane-swap instructions, pose_block.h)
                {
                    double s14[14], s7[7];
#pragma unroll
                    for (int k = 0; k < 14; ++k) s14[k] = Ops::fold32(acc[k], acc[14 + k]);
#pragma unroll
                    for (int k = 0; k < 7; ++k) s7[k] = Ops::fold16(s14[k], s14[7 + k]);
#pragma unroll
                    for (int k = 0; k < 7; ++k) {
                        double v = s7[k];
                        v = Ops::template dpp_add<0x111, 0xf>(v);
                        v = Ops::template dpp_add<0x112, 0xf>(v);
                        v = Ops::template dpp_add<0x114, 0xf>(v);
                        v = Ops::template dpp_add<0x118, 0xf>(v);
                        if ((lane & 15) == 15) s_red[p][ew][(lane >> 4) * 7 + k] = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_fetch_add(&s_job[p].done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (exits == 2) break;
            if (any) {
                idle = 0;
            } else {
                __builtin_amdgcn_s_sleep(2);
                if (++idle > P3_SPIN_LIMIT || ld_acq(&s_abort)) break;
            }
        }
        return;
    }

    // =================================== owner waves ===================================
    const int p = wv;  // pair slot
    PoseSh* sh = &s_sh[p];
    P3Job* job = &s_job[p];
    int* hist = s_hist[p];
    unsigned* inl_p = s_inl_p[p];
    unsigned* inl_l = s_inl_l[p];
    char* base = reinterpret_cast<char*>(s_dyn) + (size_t)p * lds_pair_bytes;
    int njob = 0;
    bool aborted = false;
    const bool prof = a.prof_out != nullptr;
    auto tick = [&]() -> long long { return prof ? (long long)__builtin_readcyclecounter() : 0ll; };

#pragma unroll 1
    for (int f = blockIdx.x * 2 + p; f < a.B; f += 2 * gridDim.x) {
        const long long t_begin = tick();
        long long t_wait = 0, t_stage = 0, t_out = 0, t_alg = 0, t_cov = 0, t_rm = 0, t_commit = 0, t_pre = 0, t_post = 0, t_sum = 0;
        const stvo_opt_params& prm = a.prm;
        const stvo_cam cam_f = a.cams ? a.cams[f] : a.cam;
        const pm::Cam5 cam{cam_f.fx, cam_f.fy, cam_f.cx, cam_f.cy};
        const int n_prev_p = a.n_prev_pts != nullptr ? min(a.n_prev_pts[f], a.max_pts) : 0;
        const int n_prev_l = (a.n_prev_lines != nullptr && a.max_lines > 0) ? min(a.n_prev_lines[f], a.max_lines) : 0;
        const size_t pbase = (size_t)f * a.max_pts, lbase = (size_t)f * a.max_lines;

        // ---------------- staging pass 1: which prev features are matched (compaction order = prev index order) ----------------
        // chunk c = prev features 64 c .. 64 c + 63; lane c keeps the ballots (matched / initially inlier) of chunk c
        unsigned long long my_mp = 0ull, my_ip = 0ull, my_ml = 0ull, my_il = 0ull;
        const int nch = (n_prev_p + 63) >> 6, nchl = (n_prev_l + 63) >> 6;
#pragma unroll 1
        for (int c0 = 0; c0 < nch; c0 += 8) {
            int jj[8], in[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // eight chunks' loads in flight, clamped addresses
                const int i = (c0 + u) * 64 + lane;
                const int ic = i < n_prev_p ? i : 0;
                jj[u] = a.m12p ? a.m12p[pbase + ic] : ic;
                in[u] = a.init_inl_p ? a.init_inl_p[pbase + ic] : 1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool m = (c0 + u) * 64 + lane < n_prev_p && jj[u] >= 0;
                const unsigned long long bm = __builtin_amdgcn_ballot_w64(m), bi = __builtin_amdgcn_ballot_w64(m && in[u] != 0);
                if (lane == c0 + u) {
                    my_mp = bm;
                    my_ip = bi;
                }
            }
        }
#pragma unroll 1
        for (int c = 0; c < nchl; ++c) {
            const int i = c * 64 + lane;
            const int ic = i < n_prev_l ? i : 0;
            const int j = a.m12l ? a.m12l[lbase + ic] : ic;
            const int in = a.init_inl_l ? a.init_inl_l[lbase + ic] : 1;
            const bool m = i < n_prev_l && j >= 0;
            const unsigned long long bm = __builtin_amdgcn_ballot_w64(m), bi = __builtin_amdgcn_ballot_w64(m && in != 0);
            if (lane == c) {
                my_ml = bm;
                my_il = bi;
            }
        }
        // exclusive prefix of the chunk counts (lane c: first slot of chunk c), totals
        const int cnt_p = __popcll(my_mp), cnt_l = __popcll(my_ml);
        int inc_p = cnt_p, inc_l = cnt_l;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o1 = __shfl_up(inc_p, off, 64), o2 = __shfl_up(inc_l, off, 64);
            if (lane >= off) {
                inc_p += o1;
                inc_l += o2;
            }
        }
        const int n_m_p = __shfl(inc_p, 63, 64), n_m_l = __shfl(inc_l, 63, 64);
        const int first_p = inc_p - cnt_p, first_l = inc_l - cnt_l;
        const bool fits = (size_t)n_m_l * REC_L + (size_t)n_m_p * REC_P <= (size_t)lds_pair_bytes;
        auto chunk_word = [&](unsigned long long mine, int c) -> unsigned long long {  // lane c's 64-bit word (c wave-uniform)
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, c), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), c);
            return ((unsigned long long)hi << 32) | lo;
        };

        double2* s_ln = reinterpret_cast<double2*>(base);
        double2* s_xy = reinterpret_cast<double2*>(base + (size_t)n_m_l * REC_L);
        double2* s_zq = s_xy + n_m_p;
        char* s_ob = reinterpret_cast<char*>(s_zq + n_m_p);
        bool exact = true;  // COMPACT: every observation is exactly a float pair
        if (fits) {
            // ---------------- staging pass 2: gather the matched records into the pair's LDS share ----------------
            // inlier bit arrays: slot s -> bit (s & 31) of word s >> 5
            inl_p[lane] = 0u;
            if (lane < P3_MAXL / 32) inl_l[lane] = 0u;
            wave_sync_lds();
            // groups of four chunks; the match indices of the NEXT group are requested before the records of this one
            int jn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = u * 64 + lane;
                jn[u] = a.m12p ? a.m12p[pbase + (size_t)(i < n_prev_p ? i : 0)] : i;
            }
#pragma unroll 1
            for (int c0 = 0; c0 < nch; c0 += 4) {
                int jc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) jc[u] = jn[u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = (c0 + 4 + u) * 64 + lane;
                    jn[u] = a.m12p ? a.m12p[pbase + (size_t)(i < n_prev_p ? i : 0)] : i;
                }
                double X[4], Y[4], Z[4], S2[4], OX[4], OY[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {  // four chunks' records in flight (unmatched lanes read record 0)
                    const int c = c0 + u;
                    const bool m = (chunk_word(my_mp, c) >> lane) & 1ull;
                    const size_t ic = pbase + (size_t)(m ? c * 64 + lane : 0), jx = pbase + (size_t)(m ? jc[u] : 0);
                    X[u] = a.prev_P[ic * 3 + 0];
                    Y[u] = a.prev_P[ic * 3 + 1];
                    Z[u] = a.prev_P[ic * 3 + 2];
                    S2[u] = a.prev_s2p[ic];
                    OX[u] = a.curr_pl[jx * 2 + 0];
                    OY[u] = a.curr_pl[jx * 2 + 1];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + u;
                    const unsigned long long bm = chunk_word(my_mp, c), bi = chunk_word(my_ip, c);
                    const int fs = __builtin_amdgcn_readlane(first_p, c);
                    if ((bm >> lane) & 1ull) {
                        const int sl_ = fs + __popcll(bm & ((1ull << lane) - 1ull));
                        s_xy[sl_] = make_double2(X[u], Y[u]);
                        s_zq[sl_] = make_double2(Z[u], sqrt(S2[u]));
                        if (COMPACT) {
                            const float fx_ = (float)OX[u], fy_ = (float)OY[u];
                            if ((double)fx_ != OX[u] || (double)fy_ != OY[u]) exact = false;
                            reinterpret_cast<float2*>(s_ob)[sl_] = make_float2(fx_, fy_);
                        } else {
                            reinterpret_cast<double2*>(s_ob)[sl_] = make_double2(OX[u], OY[u]);
                        }
                        if ((bi >> lane) & 1ull) atomicOr(&inl_p[sl_ >> 5], 1u << (sl_ & 31));
                    }
                }
            }
#pragma unroll 1
            for (int c = 0; c < nchl; ++c) {
                const unsigned long long bm = chunk_word(my_ml, c), bi = chunk_word(my_il, c);
                const int fs = __builtin_amdgcn_readlane(first_l, c);
                if ((bm >> lane) & 1ull) {
                    const int sl_ = fs + __popcll(bm & ((1ull << lane) - 1ull));
                    const size_t i = lbase + (size_t)(c * 64 + lane);
                    const size_t j = a.m12l ? lbase + (size_t)a.m12l[i] : i;
                    double2* q = s_ln + (size_t)sl_ * 7;
                    q[0] = make_double2(a.prev_sP[i * 3 + 0], a.prev_sP[i * 3 + 1]);
                    q[1] = make_double2(a.prev_sP[i * 3 + 2], a.prev_eP[i * 3 + 0]);
                    q[2] = make_double2(a.prev_eP[i * 3 + 1], a.prev_eP[i * 3 + 2]);
                    q[3] = make_double2(a.curr_le[j * 3 + 0], a.curr_le[j * 3 + 1]);
                    q[4] = make_double2(a.curr_le[j * 3 + 2], a.prev_spl[i * 2 + 0]);
                    q[5] = make_double2(a.prev_spl[i * 2 + 1], a.prev_epl[i * 2 + 0]);
                    q[6] = make_double2(a.prev_epl[i * 2 + 1], sqrt(a.prev_s2l[i]));  // the record carries sqrt(sigma2) (pm::line_term_q)
                    if ((bi >> lane) & 1ull) atomicOr(&inl_l[sl_ >> 5], 1u << (sl_ & 31));
                }
            }
            wave_sync_lds();
        }
        const bool all_exact = __builtin_amdgcn_ballot_w64(!exact) == 0ull;
        if (!fits || !all_exact) {  // wave-uniform: left to pose_kernel2's kernel right behind this launch
            if (lane == 0) misfit_list[atomicAdd(misfit_cnt, 1)] = f;
            continue;
        }
        t_stage = tick() - t_begin;

        // ---------------- per-pair state ----------------
        auto count_bits = [&](const unsigned* w, int nwords) -> int {  // wave-uniform popcount of a bit array (nwords <= 64)
            int c = lane < nwords ? __popc(w[lane]) : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
            return c;
        };
        {
            const int nip = count_bits(inl_p, P3_MAXP / 32), nil = count_bits(inl_l, P3_MAXL / 32);
            sh->n_m_p = n_m_p;
            sh->n_m_l = n_m_l;
            sh->n_inl_p = nip;
            sh->n_inl_l = nil;
            sh->good = 1;
            sh->err_out = -1.0;  // :313
            if (lane < 16) {
                const double v = a.init_T ? a.init_T[(size_t)f * 16 + lane] : ((lane % 5 == 0) ? 1.0 : 0.0);
                sh->DT[lane] = v;
                sh->DT0[lane] = v;
            }
            if (lane < 36) {
                sh->cov[lane] = 0.0;
                sh->H[lane] = 0.0;
            }
            if (lane == 0) {
                job->fx = cam.fx; job->fy = cam.fy; job->cx = cam.cx; job->cy = cam.cy;
                job->n_p = n_m_p;
                job->n_l = n_m_l;
            }
            wave_sync_lds();
        }

        // residual norms of this wave's slots (slot k * 64 + lane), for removeOutliers and the robust scale
        auto point_res = [&](const double* DT, int s) -> double {
            const double2 xy = s_xy[s], zq = s_zq[s];
            double ox, oy;
            if (COMPACT) {
                const float2 o = reinterpret_cast<const float2*>(s_ob)[s];
                ox = (double)o.x;
                oy = (double)o.y;
            } else {
                const double2 o = reinterpret_cast<const double2*>(s_ob)[s];
                ox = o.x;
                oy = o.y;
            }
            return pm::point_residual(DT, cam, xy.x, xy.y, zq.x, ox, oy);
        };
        auto load_line = [&](int l) -> pm::LineRec {
            const double2* q = s_ln + (size_t)l * 7;
            const double2 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4], v5 = q[5], v6 = q[6];
            pm::LineRec L;
            L.sP[0] = v0.x; L.sP[1] = v0.y; L.sP[2] = v1.x; L.eP[0] = v1.y; L.eP[1] = v2.x; L.eP[2] = v2.y;
            L.le[0] = v3.x; L.le[1] = v3.y; L.le[2] = v4.x; L.spl[0] = v4.y; L.spl[1] = v5.x; L.epl[0] = v5.y; L.epl[1] = v6.x;
            L.sigma2 = v6.y;
            return L;
        };
        auto pose12 = [&](const double* src, double* DT) {
#pragma unroll
            for (int i = 0; i < 12; ++i) DT[i] = uni3(src[i]);
        };

        // ---------------- optimizeFunctions / optimizeFunctionsRobust at sh->DT: a job for the evaluator waves ----------------
        auto evaluate = [&](bool robust) {
            double sp = 1.0, sl = 1.0;
            const long long te0 = tick();
            if (robust) {  // pre-pass :710-781: MAD scale of the inlier residual norms (this wave alone)
                double DT[12];
                pose12(sh->DT, DT);
                {
                    double rp[P3_PPT];
                    unsigned msk = 0u;
#pragma unroll
                    for (int k = 0; k < P3_PPT; ++k) {
                        const int s = k * 64 + lane;
                        rp[k] = 0.0;
                        if (s < n_m_p && ((inl_p[s >> 5] >> (s & 31)) & 1u)) {
                            rp[k] = point_res(DT, s);
                            msk |= 1u << k;
                        }
                    }
                    sp = pm::clamp_scale(wave_mad_sigma<P3_PPT>(rp, msk, sh->n_inl_p, hist));
                }
                {
                    double rl_[P3_LPT];
                    unsigned msk = 0u;
#pragma unroll
                    for (int k = 0; k < P3_LPT; ++k) {
                        const int s = k * 64 + lane;
                        rl_[k] = 0.0;
                        if (s < n_m_l && ((inl_l[s >> 5] >> (s & 31)) & 1u)) {
                            rl_[k] = pm::line_residual(DT, cam, load_line(s));
                            msk |= 1u << k;
                        }
                    }
                    sl = pm::clamp_scale(wave_mad_sigma<P3_LPT>(rl_, msk, sh->n_inl_l, hist));
                }
            }
            const long long te1 = tick();
            t_pre += te1 - te0;
            if (lane < 12) job->DT[lane] = sh->DT[lane];
            if (lane == 0) {
                job->sp = sp;
                job->sl = sl;
                job->robust = robust ? 1 : 0;
                job->done = 0;
            }
            ++njob;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) st_rel(&job->seq, njob);
            const long long tw = tick();
            t_post += tw - te1;
            {
                int spins = 0;
                while (ld_acq(&job->done) < NEV) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > P3_SPIN_LIMIT || ld_acq(&s_abort)) {
                        aborted = true;
                        break;
                    }
                }
            }
            const long long tw2 = tick();
            t_wait += tw2 - tw;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lane < 28) {  // wave partials summed in wave order => bit-reproducible
                double s = s_red[p][0][lane];
#pragma unroll
                for (int w = 1; w < NEV; ++w) s += s_red[p][w][lane];
                sh->tot[lane] = s;
            }
            wave_sync_lds();
            t_sum += tick() - tw2;
        };

        // ---------------- removeOutliers at pose DT1 (:988-1067), this wave alone ----------------
        auto remove_outliers = [&]() {
            double DT[12];
            pose12(sh->DT1, DT);
            if (prm.has_points) {
                double res[P3_PPT];
                unsigned msk = 0u;
                const int tot = sh->n_m_p;
#pragma unroll
                for (int k = 0; k < P3_PPT; ++k) {  // ALL matches, current outliers included (:998-1005)
                    const int s = k * 64 + lane;
                    res[k] = 0.0;
                    if (s < n_m_p) {
                        res[k] = point_res(DT, s) * s_zq[s].y;
                        msk |= 1u << k;
                    }
                }
                const double stdv = wave_mad_sigma<P3_PPT>(res, msk, tot, hist);
                double v0 = 0.0, v1 = 0.0, v2 = 0.0;  // mean of the samples below 2 sigma, or of all samples (src/auxiliar.cpp:405-427)
#pragma unroll
                for (int k = 0; k < P3_PPT; ++k)
                    if ((msk >> k) & 1u) {
                        if (res[k] < 2.0 * stdv) {
                            v0 += res[k];
                            v1 += 1.0;
                        }
                        v2 += res[k];
                    }
                v0 = wave_sum_d(v0);
                v1 = wave_sum_d(v1);
                v2 = wave_sum_d(v2);
                double mean = 0.0;
                if (tot != 0) {
                    const int ksel = (int)v1;
                    mean = (ksel >= (int)(0.2 * (double)tot)) ? v0 / (double)ksel : v2 / (double)tot;
                }
                const double th = prm.inlier_k * stdv;
#pragma unroll
                for (int k = 0; k < P3_PPT; ++k) {
                    if (k * 64 >= n_m_p) break;  // wave-uniform
                    const bool out = ((msk >> k) & 1u) && fabs(res[k] - mean) > th;
                    const unsigned long long bo = __builtin_amdgcn_ballot_w64(out);
                    if (lane < 2) inl_p[2 * k + lane] &= ~(unsigned)(bo >> (32 * lane));
                }
                wave_sync_lds();
                sh->n_inl_p = count_bits(inl_p, P3_MAXP / 32);
            }
            if (prm.has_lines) {
                double res[P3_LPT];
                unsigned msk = 0u;
                const int tot = sh->n_m_l;
#pragma unroll
                for (int k = 0; k < P3_LPT; ++k) {
                    const int s = k * 64 + lane;
                    res[k] = 0.0;
                    if (s < n_m_l) {
                        const pm::LineRec L = load_line(s);
                        res[k] = pm::line_residual(DT, cam, L) * L.sigma2;  // L.sigma2 = sqrt(sigma2)
                        msk |= 1u << k;
                    }
                }
                const double stdv = wave_mad_sigma<P3_LPT>(res, msk, tot, hist);
                double v0 = 0.0, v1 = 0.0, v2 = 0.0;
#pragma unroll
                for (int k = 0; k < P3_LPT; ++k)
                    if ((msk >> k) & 1u) {
                        if (res[k] < 2.0 * stdv) {
                            v0 += res[k];
                            v1 += 1.0;
                        }
                        v2 += res[k];
                    }
                v0 = wave_sum_d(v0);
                v1 = wave_sum_d(v1);
                v2 = wave_sum_d(v2);
                double mean = 0.0;
                if (tot != 0) {
                    const int ksel = (int)v1;
                    mean = (ksel >= (int)(0.2 * (double)tot)) ? v0 / (double)ksel : v2 / (double)tot;
                }
                const double th = prm.inlier_k * stdv;
#pragma unroll
                for (int k = 0; k < P3_LPT; ++k) {
                    if (k * 64 >= n_m_l) break;
                    const bool out = ((msk >> k) & 1u) && fabs(res[k] - mean) > th;
                    const unsigned long long bo = __builtin_amdgcn_ballot_w64(out);
                    if (lane < 2) inl_l[2 * k + lane] &= ~(unsigned)(bo >> (32 * lane));
                }
                wave_sync_lds();
                sh->n_inl_l = count_bits(inl_l, P3_MAXL / 32);
            }
            wave_sync_lds();
        };

        // ---------------- optimizePose state machine (:332-370), as in pose_kernel2.hip ----------------
        if (POSE2_PRIO_P3) __builtin_amdgcn_s_setprio(3);
        int status = STVO_POSE_OK, path = 0, it0 = 0, it1 = 0;
        if (sh->n_inl_p + sh->n_inl_l >= prm.min_features) {
            int stage = 0;        // 0 = first optimisation (:335-338), 1 = refinement (:345-350), 2 = robust fallback (:359)
            int alg = prm.mode;   // 0 GN, 1 robust GN, 2 LM
            int max_it = prm.max_iters;
            for (;;) {
                sh->err_prev = 999999999.9;
                sh->good = 1;
                if (lane < 16) sh->DTr[lane] = sh->DT[lane];  // robust GN's entry pose (:441)
                wave_sync_lds();
                const int n_it = (alg == 2 && max_it < 1) ? 1 : max_it;  // LM always evaluates once (:493)
                int evals = 0, action = ACT_BREAK;
                for (int it = 0; it < n_it; ++it) {
                    evaluate(alg == 1);
                    if (aborted) break;
                    ++evals;
                    const long long ta = tick();
                    if (alg == 0) t0_gn_iter<ROW>(sh, prm.min_error, prm.min_error_change, it);
                    else if (alg == 1) t0_gnr_iter<ROW>(sh, prm.min_error, prm.min_error_change);
                    else t0_lm_iter<ROW>(sh, prm.min_error, prm.min_error_change, it == 0 ? 1 : 0);
                    wave_sync_lds();
                    t_alg += tick() - ta;
                    action = sh->action;
                    if (action != ACT_CONTINUE) break;
                }
                if (aborted) break;
                const long long tc = tick();
                if (alg == 0 && action == ACT_FAIL) {
                    sh->err_out = -1.0;  // :408-409, covariance left untouched
                } else if (alg == 1 && !sh->good) {  // :473-478
                    if (lane < 16) sh->DT[lane] = sh->DTr[lane];
                    sh->err_out = -1.0;
                    if (lane < 36) sh->cov[lane] = (lane % 7 == 0) ? 1.0 : 0.0;
                } else {
                    t0_cov_from_H<ROW>(sh);  // :429 / :470 / :545 — H of the last evaluation (damped for LM)
                    sh->err_out = evals > 0 ? sh->err : 0.0;
                }
                wave_sync_lds();
                if (stage != 0) {
                    it1 = evals;
                    t_cov += tick() - tc;
                    break;
                }
                it0 = evals;
                if (lane < 16) sh->DT1[lane] = sh->DT[lane];
                wave_sync_lds();
                t0_is_good_fast<ROW>(sh, sh->DT1, sh->err_out);
                wave_sync_lds();
                t_cov += tick() - tc;
                if (sh->good) {  // :341
                    path |= STVO_PATH_STAGE1_GOOD;
                    const long long tr = tick();
                    remove_outliers();
                    t_rm += tick() - tr;
                    if (sh->n_inl_p + sh->n_inl_l >= prm.min_features) {  // :345 — restart from the INITIAL DT
                        path |= STVO_PATH_REFINED;
                        stage = 1;
                    } else {
                        pm::identity4(sh->DT);
                        status = STVO_POSE_FEW_INLIERS_AFTER;
                        wave_sync_lds();
                        break;
                    }
                } else {  // :357-362 robust GN on everything, from the initial DT
                    path |= STVO_PATH_ROBUST_FALLBACK;
                    stage = 2;
                    alg = 1;
                }
                max_it = prm.max_iters_ref;
                if (lane < 16) sh->DT[lane] = sh->DT0[lane];
                wave_sync_lds();
            }
        } else {
            pm::identity4(sh->DT);
            status = STVO_POSE_FEW_INLIERS_BEFORE;
            wave_sync_lds();
        }
        if (aborted) {
            if (lane == 0) {
                st_rel(&s_abort, 1);
                a.results[f].status = STVO_POSE_INTERNAL;
            }
            break;
        }
        const long long tcm = tick();
        if (lane == 0) t0_commit(sh, a.results + f, status, path, it0, it1);
        if (POSE2_PRIO_P3) __builtin_amdgcn_s_setprio(0);
        wave_sync_lds();

        // ---------------- inlier flags out: -1 unmatched, 0 outlier, 1 inlier, in prev index order ----------------
        const long long to = tick();
        t_commit = to - tcm;
        if (a.inl_p_out) {
            const int nch_all = (a.max_pts + 63) >> 6;
#pragma unroll 1
            for (int c = 0; c < nch_all; ++c) {
                const unsigned long long bm = chunk_word(my_mp, c & 63);
                const int fs = __builtin_amdgcn_readlane(first_p, c & 63);
                const int i = c * 64 + lane;
                int v = -1;
                if ((bm >> lane) & 1ull) {
                    const int sl_ = fs + __popcll(bm & ((1ull << lane) - 1ull));
                    v = (int)((inl_p[sl_ >> 5] >> (sl_ & 31)) & 1u);
                }
                if (i < a.max_pts) a.inl_p_out[pbase + i] = v;
            }
        }
        if (a.inl_l_out && a.max_lines > 0) {
            const int nch_all = (a.max_lines + 63) >> 6;
#pragma unroll 1
            for (int c = 0; c < nch_all; ++c) {
                const unsigned long long bm = chunk_word(my_ml, c & 63);
                const int fs = __builtin_amdgcn_readlane(first_l, c & 63);
                const int i = c * 64 + lane;
                int v = -1;
                if ((bm >> lane) & 1ull) {
                    const int sl_ = fs + __popcll(bm & ((1ull << lane) - 1ull));
                    v = (int)((inl_l[sl_ >> 5] >> (sl_ & 31)) & 1u);
                }
                if (i < a.max_lines) a.inl_l_out[lbase + i] = v;
            }
        }
        t_out = tick() - to;
        if (prof && lane == 0) {
            long long* o = a.prof_out + (size_t)f * 16;
            o[0] = tick() - t_begin;  // the pair's whole chain
            o[1] = t_wait;            // waiting for evaluation jobs
            o[2] = t_stage;
            o[3] = t_out;
            o[4] = njob;
            o[5] = t_alg;     // serial algebra of the iterations
            o[6] = t_cov;     // covariance + isGoodSolution between the stages
            o[7] = t_rm;      // removeOutliers
            o[8] = t_commit;
            o[9] = t_pre;     // robust pre-pass
            o[10] = t_post;   // posting a job
            o[11] = t_sum;    // collecting the partial sums
        }
    }
    if (lane == 0) st_rel(&job->seq, P3_SEQ_EXIT);
}

struct MisfitBuf {
    int* dev = nullptr;  // [2] counters (alternating between launches), then the list
    int cap = 0;
    unsigned launches = 0;
    std::vector<void*> retired;  // superseded blocks: kept until the stream is released (a captured step graph may hold the address)
};

std::mutex g_misfit_mu;
std::map<std::pair<int, hipStream_t>, MisfitBuf> g_misfits;

MisfitBuf* misfit_buf(hipStream_t s, int B) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_misfit_mu);
    MisfitBuf& b = g_misfits[std::make_pair(dev, s)];
    if (b.cap < B) {
        if (b.dev) b.retired.push_back(b.dev);
        b.dev = nullptr;
        b.cap = 0;
        const int cap = B < 4096 ? 4096 : B;
        if (hipMalloc((void**)&b.dev, (size_t)(cap + 4) * sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemsetAsync(b.dev, 0, (size_t)(cap + 4) * sizeof(int), s) != hipSuccess) return nullptr;
        b.cap = cap;
        b.launches = 0;
    }
    return &b;
}

template <int NW>
constexpr int pose3_pair_bytes() {  // the dynamic LDS of one pair slot: what is left of the CU's 160 KB after the static arrays
    return ((160 * 1024 - (int)(sizeof(PoseArgs) + 2 * sizeof(PoseSh) + 2 * sizeof(P3Job) + 2 * (NW - 2) * 28 * 8 + 2 * (P3_MAXP / 32 + P3_MAXL / 32) * 4 +
                                2 * 256 * 4 + 64) - 1024) / 2) & ~15;
}

template <int NW, bool ROW, bool COMPACT>
int launch_pose3_variant(hipStream_t s, const PoseArgs& a, MisfitBuf* mb) {
    constexpr int pair_bytes = pose3_pair_bytes<NW>();
    if (!lds_opt_in(reinterpret_cast<const void*>(&pose3_kernel<NW, ROW, COMPACT>), 2 * pair_bytes)) return STVO_ERR_CAPACITY;
    const int cus = device_cu_count();
    const int wgs = (a.B + 1) / 2 < cus ? (a.B + 1) / 2 : cus;
    int* cnt = mb->dev + (mb->launches & 1u);
    int* cnt_next = mb->dev + ((mb->launches + 1u) & 1u);
    ++mb->launches;
    hipLaunchKernelGGL((pose3_kernel<NW, ROW, COMPACT>), dim3(wgs), dim3(NW * 64), (size_t)(2 * pair_bytes), s, a, pair_bytes, cnt, cnt_next,
                       mb->dev + 4);
    // the pairs this launch left aside (records beyond the LDS share, inexact compact observations): pose_kernel2's kernel
    return launch_pose2_list(s, a, mb->dev + 4, cnt);
}

}  // namespace

void pose3_release_stream(hipStream_t s) {  // the caller has synchronised the stream
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_misfit_mu);
    const auto it = g_misfits.find(std::make_pair(dev, s));
    if (it == g_misfits.end()) return;
    if (it->second.dev) (void)hipFree(it->second.dev);
    for (void* p : it->second.retired) (void)hipFree(p);
    g_misfits.erase(it);
}

int launch_pose3(hipStream_t s, const PoseArgs& a) {
    if (a.B <= 0) return STVO_OK;
    if (a.max_pts > STVO_POSE_MAX_POINTS || a.max_lines > STVO_POSE_MAX_LINES) return STVO_ERR_CAPACITY;
    if (a.eval_only) return launch_pose2(s, a);
    MisfitBuf* mb = misfit_buf(s, a.B);
    if (!mb) return STVO_ERR_HIP;
    const char* env = std::getenv("STVO_POSE3_NW");  // developer override: 16 (128 VGPRs, rows) or 8 (256 VGPRs, serial 6x6)
    const int nw = env ? std::atoi(env) : 16;
    const char* ec = std::getenv("STVO_POSE3_COMPACT");  // developer override of the hint (the device still verifies every pair)
    const bool compact = ec ? std::atoi(ec) != 0 : a.obs_f32 != 0;
    if (nw == 8) return compact ? launch_pose3_variant<8, false, true>(s, a, mb) : launch_pose3_variant<8, false, false>(s, a, mb);
    return compact ? launch_pose3_variant<16, true, true>(s, a, mb) : launch_pose3_variant<16, true, false>(s, a, mb);
}

}  // namespace stvo
