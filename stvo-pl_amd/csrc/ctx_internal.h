// ctx_internal.h — the context object and staging helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "kernels.h"

struct stvo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int max_rows = 0, max_batch = 0;
    // persistent scratch for the batched path
    uint2* knn12 = nullptr;
    uint2* knn21 = nullptr;
    size_t knn_capacity = 0;  // elements in each of knn12 / knn21
    int32_t *cand = nullptr, *need = nullptr, *qsel = nullptr, *nsel = nullptr;  // lazy reverse pass
    // bump arena for the host-buffer entry points
    char* arena = nullptr;
    size_t arena_size = 0, arena_off = 0;
    uint32_t* probe_sink = nullptr;
    // overlap mode: pose kernels go to aux_stream; events order them against the matching kernels
    int overlap = 0;
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_match_done = nullptr, ev_pose_done = nullptr;
    bool pose_pending = false;
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

template <typename T>
inline T* arena_alloc(stvo_ctx* ctx, size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    if (ctx->arena_off + bytes > ctx->arena_size) return nullptr;
    T* p = reinterpret_cast<T*>(ctx->arena + ctx->arena_off);
    ctx->arena_off += bytes;
    return p;
}

template <typename T>
inline int upload(stvo_ctx* ctx, T** dst, const T* src, size_t count) {
    *dst = arena_alloc<T>(ctx, count);
    if (!*dst) return STVO_ERR_CAPACITY;
    if (count && src) HIP_TRY(ctx, hipMemcpyAsync(*dst, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return STVO_OK;
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
