// stvo_capi.hip — the extern "C" boundary (include/stvo_hip.h) over the HIP kernels.
// Host-buffer entry points stage through a per-context device arena; *_dev entry points only
// enqueue.  No CPU fallback exists anywhere in this library: every path launches gfx950 kernels.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ctx_internal.h"

namespace stvo {
namespace {

int env_int(const char* name) {
    const char* e = std::getenv(name);
    return (e && *e) ? std::atoi(e) : DBG_UNSET;
}
DebugSwitches parse_switches() {
    DebugSwitches d;
    d.pose_kernel = env_int("STVO_POSE_KERNEL");
    d.pose2p_nw = env_int("STVO_POSE2P_NW");
    d.pose_prof = env_int("STVO_POSE_PROF");
    d.pose_lds_t = env_int("STVO_POSE_LDS_T");
    d.pose_los = env_int("STVO_POSE_LOS");
    d.knn_mfma = env_int("STVO_KNN_MFMA");
    d.knn_nseg = env_int("STVO_KNN_NSEG");
    d.seq_graph = env_int("STVO_SEQ_GRAPH");
    d.seq_prof = env_int("STVO_SEQ_PROF");
    d.seq_inline = env_int("STVO_SEQ_INLINE");
    const char* lf = std::getenv("STVO_LINE_FORK");
    d.line_fork_late = lf ? (lf[0] == 'l' ? 1 : (lf[0] == 'm' ? 2 : 0)) : DBG_UNSET;
    d.line_first = env_int("STVO_LINE_FIRST");
    d.cells_ahead = env_int("STVO_CELLS_AHEAD");
    d.line_fused = env_int("STVO_LINE_FUSED");
    d.match_small = env_int("STVO_MATCH_SMALL");
    d.match_lazy = env_int("STVO_MATCH_LAZY");
    d.grid_tail = env_int("STVO_GRID_TAIL");
    d.grid_fused = env_int("STVO_GRID_FUSED");
    d.grid_fused_cap = env_int("STVO_GRID_FUSED_CAP");
    d.grid_cells = env_int("STVO_GRID_CELLS");
    d.seq_pipe = env_int("STVO_SEQ_PIPE");
    d.lines_ahead = env_int("STVO_LINES_AHEAD");
    d.grid_dyn = env_int("STVO_GRID_DYN");
    d.lsd_grow = env_int("STVO_LSD_GROW");
    d.lsd_waves = env_int("STVO_LSD_WAVES");
    d.lsd_xcd_blocks = env_int("STVO_LSD_XCD_BLOCKS");
    d.lsd_feed_ahead = env_int("STVO_LSD_FEED_AHEAD");
    d.lsd_sep = env_int("STVO_LSD_SEP");
    d.lsd_ahead = env_int("STVO_LSD_AHEAD");
    d.lsd_multi = env_int("STVO_LSD_MULTI");
    return d;
}
DebugSwitches& switches() {
    static DebugSwitches d = parse_switches();
    return d;
}

}  // namespace

const DebugSwitches& dbg() { return switches(); }
void dbg_reparse() { switches() = parse_switches(); }

}  // namespace stvo

extern "C" {

void stvo_debug_reparse_env(void) { stvo::dbg_reparse(); }

const char* stvo_backend_name(void) { return "hip-gfx950"; }
int stvo_abi_version(void) { return STVO_ABI_VERSION; }

const char* stvo_error_string(int code) {
    switch (code) {
        case STVO_OK: return "ok";
        case STVO_ERR_INVALID_ARG: return "invalid argument";
        case STVO_ERR_HIP: return "HIP runtime error";
        case STVO_ERR_NO_DEVICE: return "no gfx950 device (this library has no CPU fallback)";
        case STVO_ERR_CAPACITY: return "problem exceeds the context capacity";
        case STVO_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown error";
    }
}

const char* stvo_ctx_last_error(const stvo_ctx* ctx) { return ctx ? ctx->last_error : ""; }

int stvo_ctx_create(int device_id, int max_rows, int max_batch, stvo_ctx** out) {
    if (!out || max_rows <= 0 || max_rows > STVO_MAX_ROWS_LIMIT || max_batch <= 0) return STVO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev)
        return STVO_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return STVO_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return STVO_ERR_NO_DEVICE;  // kernels exist for gfx950 only
    stvo_ctx* ctx = new (std::nothrow) stvo_ctx();
    if (!ctx) return STVO_ERR_HIP;
    ctx->device = device_id;
    ctx->max_rows = max_rows;
    ctx->max_batch = max_batch;
    bool ok = hip_ok(ctx, hipSetDevice(device_id), "hipSetDevice") &&
              hip_ok(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), "hipStreamCreate");
    ctx->own_stream = ok;
    if (ok) {
        stvo::pose_retain_stream(ctx->stream);
        ctx->stream_retained = true;
    }
    const size_t knn_elems = (size_t)max_rows * (size_t)max_batch;
    // [nseg][B][rows] with nseg = 2 for full batches, up to KNN_MAX_NSEG for a single problem (knn_pick_nseg)
    const size_t knn_seg_elems =
        (size_t)max_rows * (size_t)(2 * max_batch > stvo::KNN_MAX_NSEG ? 2 * max_batch : stvo::KNN_MAX_NSEG);
    ctx->knn_capacity = knn_seg_elems;
    // arena: descriptors + records + per-row scratch of one host-buffer call, with slack
    ctx->arena_size = (size_t)max_rows * 1024 + ((size_t)4 << 20);
    ok = ok && hip_ok(ctx, hipMalloc((void**)&ctx->knn12, knn_seg_elems * sizeof(uint2)), "hipMalloc knn12") &&
         hip_ok(ctx, hipMalloc((void**)&ctx->knn21, knn_seg_elems * sizeof(uint2)), "hipMalloc knn21") &&
         hip_ok(ctx, hipMalloc((void**)&ctx->cand, knn_elems * sizeof(int32_t)), "hipMalloc cand") &&
         hip_ok(ctx, hipMalloc((void**)&ctx->need, knn_elems * sizeof(int32_t)), "hipMalloc need") &&
         hip_ok(ctx, hipMalloc((void**)&ctx->qsel, knn_elems * sizeof(int32_t)), "hipMalloc qsel") &&
         hip_ok(ctx, hipMalloc((void**)&ctx->nsel, (size_t)max_batch * 5 * sizeof(int32_t)), "hipMalloc nsel") &&
         hip_ok(ctx, hipMalloc((void**)&ctx->arena, ctx->arena_size), "hipMalloc arena") &&
         hip_ok(ctx, hipHostMalloc((void**)&ctx->arena_host, ctx->arena_size, hipHostMallocDefault), "hipHostMalloc") &&
         hip_ok(ctx, hipMalloc((void**)&ctx->probe_sink, 256), "hipMalloc sink");
    if (!ok) {
        stvo_ctx_destroy(ctx);
        return STVO_ERR_HIP;
    }
    *out = ctx;
    return STVO_OK;
}

int stvo_ctx_destroy(stvo_ctx* ctx) {
    if (!ctx) return STVO_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->knn12) (void)hipFree(ctx->knn12);
    if (ctx->knn21) (void)hipFree(ctx->knn21);
    if (ctx->cand) (void)hipFree(ctx->cand);
    if (ctx->need) (void)hipFree(ctx->need);
    if (ctx->qsel) (void)hipFree(ctx->qsel);
    if (ctx->nsel) (void)hipFree(ctx->nsel);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->arena_host) (void)hipHostFree(ctx->arena_host);
    if (ctx->probe_sink) (void)hipFree(ctx->probe_sink);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->stream_retained) stvo::pose_release_stream(ctx->stream);
    if (ctx->aux_stream) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        stvo::pose_release_stream(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
        // (the aux stream may have been created by a sequence pipeline — pipelined steps — without the overlap mode's events)
        if (ctx->ev_match_done) (void)hipEventDestroy(ctx->ev_match_done);
        if (ctx->ev_pose_done) (void)hipEventDestroy(ctx->ev_pose_done);
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return STVO_OK;
}

int stvo_ctx_set_stream(stvo_ctx* ctx, void* hip_stream) {
    if (!ctx) return STVO_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream_retained) stvo::pose_release_stream(ctx->stream);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->own_stream = false;
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);  // may be the NULL stream
    stvo::pose_retain_stream(ctx->stream);
    ctx->stream_retained = true;
    return STVO_OK;
}

int stvo_ctx_synchronize(stvo_ctx* ctx) {
    if (!ctx) return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->aux_stream));
    ctx->pose_pending = false;
    return STVO_OK;
}

int stvo_ctx_set_overlap(stvo_ctx* ctx, int enable) {
    if (!ctx) return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (enable && !ctx->aux_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
        stvo::pose_retain_stream(ctx->aux_stream);
    }
    if (enable && !ctx->ev_match_done) {  // (a sequence pipeline may have created the aux stream before: its pipelined steps)
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_match_done, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_pose_done, hipEventDisableTiming));
    }
    if (!enable && ctx->aux_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->aux_stream));
    ctx->overlap = enable ? 1 : 0;
    return STVO_OK;
}

// ------------------------------------------------------------------------------------------------
int stvo_match_nnr_mutual(stvo_ctx* ctx, const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int mutual,
                          int32_t* m12, int32_t* n_matches) {
    if (!ctx || n1 < 0 || n2 < 0 || (n1 > 0 && (!d1 || !m12)) || (n2 > 0 && !d2)) return STVO_ERR_INVALID_ARG;
    if (!(nnr <= 1.0f)) return STVO_ERR_INVALID_ARG;  // tie handling is only order-independent for nnr <= 1
    if (n1 > ctx->max_rows || n2 > ctx->max_rows) return STVO_ERR_CAPACITY;
    if (n_matches) *n_matches = 0;
    if (n1 == 0) return STVO_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->arena_off = 0;
    ctx->upload_hi = 0;
    const int stride = n1 > n2 ? n1 : n2;
    uint8_t *dd1, *dd2;
    int32_t *dn1, *dn2, *dm12;
    TRY(upload(ctx, &dd1, d1, (size_t)n1 * STVO_DESC_BYTES, (size_t)stride * STVO_DESC_BYTES));
    TRY(upload(ctx, &dd2, d2, (size_t)n2 * STVO_DESC_BYTES, (size_t)stride * STVO_DESC_BYTES));
    TRY(upload(ctx, &dn1, &n1, 1));
    TRY(upload(ctx, &dn2, &n2, 1));
    TRY(flush_uploads(ctx));
    TRY(upload(ctx, &dm12, (const int32_t*)nullptr, (size_t)stride));
    if (mutual) {
        const stvo::LazyScratch w{ctx->knn12, ctx->knn21, ctx->cand, ctx->need, ctx->qsel, ctx->nsel, ctx->knn_capacity};
        stvo::launch_match_mutual_lazy(ctx->stream, 1, stride, dd1, dn1, dd2, dn2, nnr, w, dm12, 0, nullptr);
    } else {
        const int nseg = stvo::knn_pick_nseg(1, stride, ctx->knn_capacity);
        stvo::launch_hamming_knn2(ctx->stream, 1, stride, stride, dd1, dn1, dd2, dn2, ctx->knn12, ctx->knn21, 0, 0, 0,
                                  nullptr, nullptr, nseg);
        stvo::launch_nnr_mutual(ctx->stream, 1, stride, ctx->knn12, ctx->knn21, dn1, dn2, nnr, 0, dm12, nseg);
    }
    TRY(check_launch(ctx));
    TRY(download_begin(ctx, dm12, (size_t)n1 * sizeof(int32_t)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(m12, host_mirror(ctx, dm12), (size_t)n1 * sizeof(int32_t));
    if (n_matches) {
        int c = 0;
        for (int i = 0; i < n1; ++i) c += m12[i] >= 0;
        *n_matches = c;
    }
    return STVO_OK;
}

int stvo_match_nnr_mutual_batched_dev(stvo_ctx* ctx, int B, int row_stride, const uint8_t* d1, const int32_t* n1,
                                      const uint8_t* d2, const int32_t* n2, float nnr, int mutual, int32_t* m12) {
    if (!ctx || B < 0 || row_stride <= 0 || !d1 || !d2 || !n1 || !n2 || !m12 || !(nnr <= 1.0f))
        return STVO_ERR_INVALID_ARG;
    // per-problem scratch (nsel: [5][max_batch]) and per-row scratch (cand / need / qsel: max_batch x max_rows; the knn
    // arrays hold the forward top-2 plus the reverse-check lists: 2 x B x row_stride) are sized by the context
    if (B > ctx->max_batch || (size_t)B * row_stride > (size_t)ctx->max_batch * ctx->max_rows || row_stride > STVO_MAX_ROWS_LIMIT ||
        (size_t)2 * B * row_stride > ctx->knn_capacity)
        return STVO_ERR_CAPACITY;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (mutual) {
        const stvo::LazyScratch w{ctx->knn12, ctx->knn21, ctx->cand, ctx->need, ctx->qsel, ctx->nsel, ctx->knn_capacity};
        stvo::launch_match_mutual_lazy(ctx->stream, B, row_stride, d1, n1, d2, n2, nnr, w, m12, 0, nullptr);
    } else {
        const int nseg = stvo::knn_pick_nseg(B, row_stride, ctx->knn_capacity);
        stvo::launch_hamming_knn2(ctx->stream, B, row_stride, row_stride, d1, n1, d2, n2, ctx->knn12, ctx->knn21, 0, 0, 0,
                                  nullptr, nullptr, nseg);
        stvo::launch_nnr_mutual(ctx->stream, B, row_stride, ctx->knn12, ctx->knn21, n1, n2, nnr, 0, m12, nseg);
    }
    return check_launch(ctx);
}

// ------------------------------------------------------------------------------------------------
namespace {

// stage a stvo_matched (host) into the arena as a B = 1 records-mode problem
int stage_records(stvo_ctx* ctx, const stvo_matched* m, const double* T, stvo::PoseArgs* a, int32_t** d_inl_p,
                  int32_t** d_inl_l) {
    std::memset(a, 0, sizeof(*a));
    if (m->np < 0 || m->nl < 0) return STVO_ERR_INVALID_ARG;
    if (m->np > STVO_POSE_MAX_POINTS || m->nl > STVO_POSE_MAX_LINES) return STVO_ERR_CAPACITY;
    a->B = 1;
    a->max_pts = m->np > 0 ? m->np : 1;
    a->max_lines = m->nl;
    double *P, *obs, *s2p, *sP, *eP, *le, *spl, *epl, *s2l, *dT;
    int32_t *dnp, *dnl;
    TRY(upload(ctx, &P, m->P, (size_t)m->np * 3));
    TRY(upload(ctx, &obs, m->pl_obs, (size_t)m->np * 2));
    TRY(upload(ctx, &s2p, m->sigma2p, (size_t)m->np));
    TRY(upload(ctx, d_inl_p, (const int32_t*)m->inlier_p, (size_t)m->np));
    TRY(upload(ctx, &sP, m->sP, (size_t)m->nl * 3));
    TRY(upload(ctx, &eP, m->eP, (size_t)m->nl * 3));
    TRY(upload(ctx, &le, m->le_obs, (size_t)m->nl * 3));
    TRY(upload(ctx, &spl, m->spl, (size_t)m->nl * 2));
    TRY(upload(ctx, &epl, m->epl, (size_t)m->nl * 2));
    TRY(upload(ctx, &s2l, m->sigma2l, (size_t)m->nl));
    TRY(upload(ctx, d_inl_l, (const int32_t*)m->inlier_l, (size_t)m->nl));
    TRY(upload(ctx, &dnp, &m->np, 1));
    TRY(upload(ctx, &dnl, &m->nl, 1));
    TRY(upload(ctx, &dT, T, 16));
    TRY(flush_uploads(ctx));
    a->n_prev_pts = dnp;
    a->prev_P = P;
    a->prev_s2p = s2p;
    a->curr_pl = obs;
    a->m12p = nullptr;
    a->init_inl_p = *d_inl_p;
    a->n_prev_lines = dnl;
    a->prev_sP = sP;
    a->prev_eP = eP;
    a->prev_spl = spl;
    a->prev_epl = epl;
    a->prev_s2l = s2l;
    a->curr_le = le;
    a->m12l = nullptr;
    a->init_inl_l = *d_inl_l;
    a->init_T = dT;
    return STVO_OK;
}

}  // namespace

int stvo_normal_eq(stvo_ctx* ctx, const double T[16], const stvo_cam* cam, const stvo_opt_params* params,
                   const stvo_matched* m, int robust, double H[36], double g[6], double* e, int32_t* n_used) {
    if (!ctx || !T || !cam || !params || !m || !H || !g || !e) return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->arena_off = 0;
    ctx->upload_hi = 0;
    stvo::PoseArgs a{};
    int32_t *dip, *dil;
    TRY(stage_records(ctx, m, T, &a, &dip, &dil));
    a.cam = *cam;
    a.prm = *params;
    double* dout = arena_alloc<double>(ctx, 44);
    if (!dout) return STVO_ERR_CAPACITY;
    a.eval_only = 1;
    a.eval_robust = robust;
    a.eval_out = dout;
    TRY(stvo::launch_pose(ctx->stream, a));
    TRY(check_launch(ctx));
    TRY(download_begin(ctx, dout, 44 * sizeof(double)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double* out = host_mirror(ctx, dout);
    std::memcpy(H, out, 36 * sizeof(double));
    std::memcpy(g, out + 36, 6 * sizeof(double));
    *e = out[42];
    if (n_used) *n_used = (int32_t)out[43];
    return STVO_OK;
}

int stvo_optimize_pose(stvo_ctx* ctx, const double init_T[16], const stvo_cam* cam, const stvo_opt_params* params,
                       stvo_matched* m, stvo_pose_result* out) {
    if (!ctx || !init_T || !cam || !params || !m || !out) return STVO_ERR_INVALID_ARG;
    if ((m->np > 0 && (!m->P || !m->pl_obs || !m->sigma2p || !m->inlier_p)) ||
        (m->nl > 0 && (!m->sP || !m->eP || !m->le_obs || !m->spl || !m->epl || !m->sigma2l || !m->inlier_l)))
        return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->arena_off = 0;
    ctx->upload_hi = 0;
    stvo::PoseArgs a{};
    int32_t *dip, *dil;
    TRY(stage_records(ctx, m, init_T, &a, &dip, &dil));
    a.cam = *cam;
    a.prm = *params;
    stvo_pose_result* dres = arena_alloc<stvo_pose_result>(ctx, 1);
    int32_t* dop = arena_alloc<int32_t>(ctx, (size_t)a.max_pts);
    int32_t* dol = arena_alloc<int32_t>(ctx, (size_t)(a.max_lines > 0 ? a.max_lines : 1));
    if (!dres || !dop || !dol) return STVO_ERR_CAPACITY;
    a.results = dres;
    a.inl_p_out = dop;
    a.inl_l_out = dol;
    TRY(stvo::launch_pose(ctx->stream, a));
    TRY(check_launch(ctx));
    TRY(download_begin(ctx, dres, sizeof(*out)));
    if (m->np) TRY(download_begin(ctx, dop, (size_t)m->np * 4));
    if (m->nl) TRY(download_begin(ctx, dol, (size_t)m->nl * 4));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out, host_mirror(ctx, dres), sizeof(*out));
    const int32_t* ip = host_mirror(ctx, dop);
    const int32_t* il = host_mirror(ctx, dol);
    for (int i = 0; i < m->np; ++i) m->inlier_p[i] = ip[i] > 0 ? 1 : 0;
    for (int i = 0; i < m->nl; ++i) m->inlier_l[i] = il[i] > 0 ? 1 : 0;
    return STVO_OK;
}

// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kOverlapLdsPad = 26 * 1024;  // 6 matching workgroups (24 waves) per CU instead of 8

int fill_pose_args(const stvo_track_batch_dev* b, const stvo_cam* cam, const stvo_opt_params* prm, bool identity,
                   stvo::PoseArgs* a) {
    std::memset(a, 0, sizeof(*a));
    a->B = b->B;
    a->max_pts = b->max_pts;
    a->max_lines = b->max_lines;
    a->n_prev_pts = b->n_prev_pts;
    a->prev_P = b->prev_P;
    a->prev_s2p = b->prev_sigma2p;
    a->curr_pl = b->curr_pl;
    a->m12p = identity ? nullptr : b->m12_pts;
    a->n_prev_lines = b->max_lines > 0 ? b->n_prev_lines : nullptr;
    a->prev_sP = b->prev_sP;
    a->prev_eP = b->prev_eP;
    a->prev_spl = b->prev_spl;
    a->prev_epl = b->prev_epl;
    a->prev_s2l = b->prev_sigma2l;
    a->curr_le = b->curr_le;
    a->m12l = identity ? nullptr : b->m12_lines;
    a->init_T = b->init_T;
    a->cam = *cam;
    a->prm = *prm;
    a->results = b->results;
    a->inl_p_out = b->inlier_pts;
    a->inl_l_out = b->inlier_lines;
    return STVO_OK;
}

int check_batch(stvo_ctx* ctx, const stvo_track_batch_dev* b, bool need_desc) {
    if (!ctx || !b || b->B < 0 || b->max_pts <= 0 || b->max_lines < 0) return STVO_ERR_INVALID_ARG;
    if (!b->n_prev_pts || !b->prev_P || !b->prev_sigma2p || !b->curr_pl || !b->results) return STVO_ERR_INVALID_ARG;
    if (need_desc && (!b->prev_pdesc || !b->curr_pdesc || !b->n_curr_pts || !b->m12_pts)) return STVO_ERR_INVALID_ARG;
    if (b->max_lines > 0) {
        if (!b->n_prev_lines || !b->prev_sP || !b->prev_eP || !b->prev_spl || !b->prev_epl || !b->prev_sigma2l ||
            !b->curr_le)
            return STVO_ERR_INVALID_ARG;
        if (need_desc && (!b->prev_ldesc || !b->curr_ldesc || !b->n_curr_lines || !b->m12_lines))
            return STVO_ERR_INVALID_ARG;
    }
    if (b->max_pts > STVO_POSE_MAX_POINTS || b->max_lines > STVO_POSE_MAX_LINES) return STVO_ERR_CAPACITY;
    const size_t rows = (size_t)(b->max_pts > b->max_lines ? b->max_pts : b->max_lines);
    if (b->B > ctx->max_batch || (size_t)b->B * rows > (size_t)ctx->max_batch * (size_t)ctx->max_rows ||
        (size_t)2 * b->B * rows > ctx->knn_capacity)
        return STVO_ERR_CAPACITY;
    return STVO_OK;
}

}  // namespace

int stvo_track_batched_dev(stvo_ctx* ctx, const stvo_track_batch_dev* b, const stvo_cam* cam,
                           const stvo_opt_params* params, float nnr_points, float nnr_lines, int mutual) {
    if (!cam || !params || !(nnr_points <= 1.0f) || !(nnr_lines <= 1.0f)) return STVO_ERR_INVALID_ARG;
    TRY(check_batch(ctx, b, true));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (b->B == 0) return STVO_OK;
    // Overlap mode: the matching kernels of THIS call may run while the pose kernel of the PREVIOUS call
    // is still in flight on aux_stream.  K1 only writes the context's knn scratch (never read by the pose
    // kernel); K2 rewrites m12, which the previous pose kernel may still be reading => K2 waits for it.
    const int pad = ctx->overlap ? kOverlapLdsPad : 0;
    hipEvent_t prev_pose = (ctx->overlap && ctx->pose_pending) ? ctx->ev_pose_done : nullptr;
    const stvo::LazyScratch w{ctx->knn12, ctx->knn21, ctx->cand, ctx->need, ctx->qsel, ctx->nsel, ctx->knn_capacity};
    auto match_set = [&](int stride, const uint8_t* da, const int32_t* na, const uint8_t* db, const int32_t* nb,
                         float nnr, int32_t* m12) {
        if (mutual) {
            hipEvent_t* tev = nullptr;
            if (ctx->timing && da == b->prev_pdesc) {  // time the point-descriptor launches only
                while (ctx->ev_pool.size() < ctx->ev_used + 4) {
                    hipEvent_t e;
                    if (hipEventCreate(&e) != hipSuccess) break;
                    ctx->ev_pool.push_back(e);
                }
                if (ctx->ev_pool.size() >= ctx->ev_used + 4) {
                    tev = ctx->ev_pool.data() + ctx->ev_used;
                    ctx->ev_used += 4;
                }
            }
            stvo::launch_match_mutual_lazy(ctx->stream, b->B, stride, da, na, db, nb, nnr, w, m12, pad, prev_pose, tev);
        } else {
            const int nseg = stvo::knn_pick_nseg(b->B, stride, ctx->knn_capacity);
            stvo::launch_hamming_knn2(ctx->stream, b->B, stride, stride, da, na, db, nb, ctx->knn12, ctx->knn21, 0, pad, 0,
                                      nullptr, nullptr, nseg);
            if (prev_pose) (void)hipStreamWaitEvent(ctx->stream, prev_pose, 0);
            stvo::launch_nnr_mutual(ctx->stream, b->B, stride, ctx->knn12, ctx->knn21, na, nb, nnr, 0, m12, nseg);
        }
    };
    // matchF2FPoints (:131-153)
    if (params->has_points)
        match_set(b->max_pts, b->prev_pdesc, b->n_prev_pts, b->curr_pdesc, b->n_curr_pts, nnr_points, b->m12_pts);
    // matchF2FLines (:155-180)
    if (params->has_lines && b->max_lines > 0)
        match_set(b->max_lines, b->prev_ldesc, b->n_prev_lines, b->curr_ldesc, b->n_curr_lines, nnr_lines, b->m12_lines);
    if (prev_pose) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, prev_pose, 0));  // also when nothing was matched
    stvo::PoseArgs a{};
    fill_pose_args(b, cam, params, false, &a);
    // a feature kind that is switched off is never matched => matched_pt / matched_ls stay empty (:137,160)
    if (!params->has_points) a.n_prev_pts = nullptr;
    if (!params->has_lines) a.n_prev_lines = nullptr;
    hipStream_t pose_stream = ctx->stream;
    if (ctx->overlap) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_match_done, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_match_done, 0));
        pose_stream = ctx->aux_stream;
    }
    TRY(stvo::launch_pose(pose_stream, a));
    if (ctx->overlap) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_pose_done, ctx->aux_stream));
        ctx->pose_pending = true;
    }
    return check_launch(ctx);
}

int stvo_optimize_pose_batched_dev(stvo_ctx* ctx, const stvo_track_batch_dev* b, const stvo_cam* cam,
                                   const stvo_opt_params* params) {
    if (!cam || !params) return STVO_ERR_INVALID_ARG;
    TRY(check_batch(ctx, b, false));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stvo::PoseArgs a{};
    fill_pose_args(b, cam, params, true, &a);
    TRY(stvo::launch_pose(ctx->stream, a));
    return check_launch(ctx);
}

int stvo_time_stage_dev(stvo_ctx* ctx, const stvo_track_batch_dev* b, const stvo_cam* cam,
                        const stvo_opt_params* params, float nnr, int stage, int iters, float* avg_ms) {
    if (!cam || !params || !avg_ms || iters <= 0) return STVO_ERR_INVALID_ARG;
    TRY(check_batch(ctx, b, true));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    stvo::PoseArgs a{};
    fill_pose_args(b, cam, params, false, &a);
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    const stvo::LazyScratch lw{ctx->knn12, ctx->knn21, ctx->cand, ctx->need, ctx->qsel, ctx->nsel, ctx->knn_capacity};
    for (int it = 0; it < iters; ++it) {
        const int pad = ctx->overlap ? kOverlapLdsPad : 0;
        const int nseg = stvo::knn_pick_nseg(b->B, b->max_pts, ctx->knn_capacity);
        if (stage == 0 || stage == 2) {  // hamming_knn2: the forward top-2 scan of every prev row
            stvo::launch_hamming_knn2(ctx->stream, b->B, b->max_pts, b->max_pts, b->prev_pdesc, b->n_prev_pts,
                                      b->curr_pdesc, b->n_curr_pts, ctx->knn12, ctx->knn21, 0, pad, 0, nullptr, nullptr, nseg);
        } else if (stage == 3) {  // hamming_verify on the claims of the LAST stvo_track_batched_dev call
            stvo::launch_hamming_verify(ctx->stream, b->B, b->max_pts, b->prev_pdesc, b->n_prev_pts, b->curr_pdesc, nnr, lw,
                                        pad, nseg, b->m12_pts);
        } else if (stage == 4) {  // developer probe: both directions as full top-2 scans (the pre-lazy formulation)
            stvo::launch_hamming_knn2(ctx->stream, b->B, b->max_pts, b->max_pts, b->prev_pdesc, b->n_prev_pts,
                                      b->curr_pdesc, b->n_curr_pts, ctx->knn12, ctx->knn21, 1, pad, 0, nullptr, nullptr, nseg);
        } else
            stvo::launch_pose(ctx->stream, a);
    }
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / (float)iters;
    if (stage == 1 && stvo::dbg().pose_prof != stvo::DBG_UNSET) {  // developer aid: per-phase ticks of the solver lane
        long long* dprof = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&dprof, (size_t)b->B * 16 * sizeof(long long)));
        a.prof_out = dprof;
        stvo::launch_pose(ctx->stream, a);
        std::vector<long long> h((size_t)b->B * 16);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (the copy below rides the null stream, which does not wait for this one)
        HIP_TRY(ctx, hipMemcpy(h.data(), dprof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double m[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int f = 0; f < b->B; ++f)
            for (int i = 0; i < 16; ++i) m[i] += (double)h[(size_t)f * 16 + i] / b->B;
        std::fprintf(stderr, "[pose prof] mean ticks/frame: evaluate %.0f  iter-algebra %.0f  cov+isgood+commit %.0f  "
                             "remove_outliers %.0f  total %.0f | worker: eval-compute %.0f  barrier+solver-sum %.0f  prefetch+fold %.0f\n", m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
        std::fprintf(stderr, "[pose prof] per worker wave, compute + fold ticks/frame: %.0f %.0f %.0f %.0f %.0f %.0f | prologue %.0f | last wave %.0f\n", m[8], m[9], m[10],
                     m[11], m[12], m[13], m[14], m[15]);
        (void)hipFree(dprof);
    }
    return check_launch(ctx);
}

int stvo_last_reverse_counts(stvo_ctx* ctx, int B, int32_t* counts) {
    if (!ctx || !counts || B <= 0 || B > ctx->max_batch) return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(counts, ctx->nsel, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost));
    return STVO_OK;
}

int stvo_last_reverse_plan(stvo_ctx* ctx, int B, int32_t* plan) {
    if (!ctx || !plan || B <= 0 || B > ctx->max_batch) return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(plan, ctx->nsel, (size_t)B * 5 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return STVO_OK;
}

int stvo_ctx_set_kernel_timing(stvo_ctx* ctx, int enable) {
    if (!ctx) return STVO_ERR_INVALID_ARG;
    ctx->timing = enable ? 1 : 0;
    ctx->ev_used = 0;
    return STVO_OK;
}

int stvo_ctx_get_kernel_timing(stvo_ctx* ctx, float* avg_ms_forward, float* avg_ms_reverse, int32_t* n_pairs) {
    if (!ctx || !avg_ms_forward || !avg_ms_reverse || !n_pairs) return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double f = 0.0, r = 0.0;
    int n = 0;
    for (size_t k = 0; k + 4 <= ctx->ev_used; k += 4) {
        float a = 0.f, c = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->ev_pool[k], ctx->ev_pool[k + 1]));
        HIP_TRY(ctx, hipEventElapsedTime(&c, ctx->ev_pool[k + 2], ctx->ev_pool[k + 3]));
        f += a;
        r += c;
        ++n;
    }
    *n_pairs = n;
    *avg_ms_forward = n ? (float)(f / n) : 0.f;
    *avg_ms_reverse = n ? (float)(r / n) : 0.f;
    ctx->ev_used = 0;
    return STVO_OK;
}

int stvo_valu_peak_probe(stvo_ctx* ctx, double* lane_ops_per_s) {
    if (!ctx || !lane_ops_per_s) return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int blocks = 256 * 8, iters = 4000;
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    stvo::launch_valu_probe(ctx->stream, blocks, 64, ctx->probe_sink);  // warm-up
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    stvo::launch_valu_probe(ctx->stream, blocks, iters, ctx->probe_sink);
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *lane_ops_per_s = (double)blocks * 256.0 * (double)iters * stvo::kValuProbeOpsPerThreadIter / ((double)ms * 1e-3);
    return check_launch(ctx);
}

// K3 entry points live in grid_capi.hip
}  // extern "C"
