// ctx_internal.h — the context object and staging helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "kernels.h"

struct stvo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool stream_retained = false;  // this context holds a reference on the stream's pose-record arena (kernels.h)
    int max_rows = 0, max_batch = 0;
    // persistent scratch for the batched path
    uint2* knn12 = nullptr;
    uint2* knn21 = nullptr;
    size_t knn_capacity = 0;  // elements in each of knn12 / knn21
    int32_t *cand = nullptr, *need = nullptr, *qsel = nullptr, *nsel = nullptr;  // lazy reverse pass
    // bump arena for the host-buffer entry points
    char* arena = nullptr;
    char* arena_host = nullptr;  // pinned mirror of the arena: uploads are gathered here and sent as ONE H2D copy
    size_t arena_size = 0, arena_off = 0, upload_hi = 0;
    uint32_t* probe_sink = nullptr;
    // overlap mode: pose kernels go to aux_stream; events order them against the matching kernels
    int overlap = 0;
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_match_done = nullptr, ev_pose_done = nullptr;
    bool pose_pending = false;
    // optional live kernel timing (bench): event pairs around every hamming_knn2 launch of the batched path
    int timing = 0;
    std::vector<hipEvent_t> ev_pool;  // start/stop pairs, grown on demand
    size_t ev_used = 0;
    char last_error[256] = {0};
};

namespace stvo_detail {

inline bool hip_ok(stvo_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    if (ctx) std::snprintf(ctx->last_error, sizeof(ctx->last_error), "%s: %s", what, hipGetErrorString(e));
    return false;
}
#define HIP_TRY(ctx, call)                                   \
    do {                                                     \
        if (!hip_ok((ctx), (call), #call)) return STVO_ERR_HIP; \
    } while (0)

// Zeroes freshly allocated device memory ON THE CONTEXT'S STREAM and waits.  hipMemset runs on the null stream, which the context's
// non-blocking stream does not wait for, and it may return before the fill has run: a first call that followed at once had its
// host-to-device copy land UNDER the fill (the first image of a new detector came out empty on some boxes of the pool).
inline bool zero_device(stvo_ctx* ctx, void* p, size_t bytes, const char* what) {
    return hip_ok(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream), what) && hip_ok(ctx, hipStreamSynchronize(ctx->stream), what);
}

// Set-up data to the device ON THE CONTEXT'S STREAM, complete on return (same reason: nothing of a context's set-up rides the null stream).
inline bool upload_now(stvo_ctx* ctx, void* dst, const void* src, size_t bytes, const char* what) {
    return hip_ok(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream), what) && hip_ok(ctx, hipStreamSynchronize(ctx->stream), what);
}

template <typename T>
inline T* arena_alloc(stvo_ctx* ctx, size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    if (ctx->arena_off + bytes > ctx->arena_size) return nullptr;
    T* p = reinterpret_cast<T*>(ctx->arena + ctx->arena_off);
    ctx->arena_off += bytes;
    return p;
}

// Host data is first gathered in the pinned mirror (same offset as the device allocation); flush_uploads()
// then moves everything with a single asynchronous H2D copy.  A dozen small pageable hipMemcpyAsync calls
// cost ~10-20 us each, which dominated the single-stream latency of the host-buffer entry points.
template <typename T>
inline int upload(stvo_ctx* ctx, T** dst, const T* src, size_t count, size_t count_alloc = 0) {
    *dst = arena_alloc<T>(ctx, count_alloc > count ? count_alloc : count);
    if (!*dst) return STVO_ERR_CAPACITY;
    if (count && src) {
        const size_t off = (size_t)(reinterpret_cast<char*>(*dst) - ctx->arena);
        std::memcpy(ctx->arena_host + off, src, count * sizeof(T));
        if (off + count * sizeof(T) > ctx->upload_hi) ctx->upload_hi = off + count * sizeof(T);
    }
    return STVO_OK;
}

inline int flush_uploads(stvo_ctx* ctx) {
    if (ctx->upload_hi) {
        if (ctx->upload_hi <= ((size_t)4 << 20))  // arena offsets are multiples of 256: 16-byte granularity is safe
            stvo::launch_copy16(ctx->stream, ctx->arena_host, ctx->arena, ctx->upload_hi);
        else
            HIP_TRY(ctx, hipMemcpyAsync(ctx->arena, ctx->arena_host, ctx->upload_hi, hipMemcpyHostToDevice, ctx->stream));
        ctx->upload_hi = 0;
    }
    return STVO_OK;
}

// asynchronous D2H of an arena region into the pinned mirror; read it through host_mirror() after the sync
inline int download_begin(stvo_ctx* ctx, const void* dev, size_t bytes) {
    const size_t off = (size_t)(reinterpret_cast<const char*>(dev) - ctx->arena);
    if (bytes == 0) return STVO_OK;
    if (bytes <= ((size_t)256 << 10) && (off & 15) == 0)
        stvo::launch_copy16(ctx->stream, dev, ctx->arena_host + off, bytes);  // may write up to 15 bytes of slack (arena blocks are 256-aligned)
    else
        HIP_TRY(ctx, hipMemcpyAsync(ctx->arena_host + off, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return STVO_OK;
}
template <typename T>
inline const T* host_mirror(stvo_ctx* ctx, const T* dev) {
    return reinterpret_cast<const T*>(ctx->arena_host + (reinterpret_cast<const char*>(dev) - ctx->arena));
}
#define TRY(expr)                \
    do {                         \
        int _rc = (expr);        \
        if (_rc != STVO_OK) return _rc; \
    } while (0)

inline int check_launch(stvo_ctx* ctx) {
    HIP_TRY(ctx, hipGetLastError());
    return STVO_OK;
}

}  // namespace stvo_detail
using namespace stvo_detail;
