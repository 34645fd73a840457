// orb_kernels.hip — the ORB point front-end on gfx950: what the reference obtains from
//     cv::ORB::create(...)->detectAndCompute(img, Mat(), points, pdesc, false)     (/root/reference/src/stereoFrame.cpp:104-118)
// for ONE pyramid level (config_kitti.yaml: orb_nlevels 1), FAST_SCORE ranking (orb_score 1), WTA_K 2, patch 31:
//   orb_fast_kernel      FAST-9/16 score of every pixel (cornerScore<16>), LDS-tiled, cheap compass-point rejection first
//   orb_nms_kernel       3x3 non-maximum suppression + border filter, response histogram per image
//   orb_cut_kernel       KeyPointsFilter::retainBest as a histogram cut (ties kept) + row offsets of the survivors
//   orb_emit_kernel      ordered (row-major) emission of the key-points: one wave per image row
//   orb_blur_kernel      GaussianBlur 7x7, sigma 2, 8-bit fixed point, BORDER_REFLECT_101
//   orb_describe_kernel  intensity-centroid angle (ICAngles, fastAtan2) + rotated BRIEF, one wave per key-point
// OpenCV is third-party code that is not under /root/reference: the semantics are those of oracle/stvo_orb_oracle.c (a
// restatement of OpenCV's algorithm, parity unpinned), against which these kernels are bit-exact (tests/test_gpu_orb.py).
// Integer / byte work throughout; the only floating point is the angle polynomial and the rotation of the test pattern, in
// FP32 exactly as OpenCV evaluates them (no fused multiply-adds).
#include <cmath>
#include <cstring>
#include <new>

#include "ctx_internal.h"

#pragma clang fp contract(off)

namespace stvo {
namespace {

constexpr int ORB_HP = 15;  // half patch
constexpr int TILE_W = 64, TILE_H = 8;

struct OrbDev {
    int B, cols, rows, K, nfeatures, fast_th, edge_th;
    const uint8_t* img;   // [B][rows][cols]
    uint8_t* score;       // [B][rows][cols] FAST score of corners (>= fast_th), 0 elsewhere
    uint8_t* keep;        // [B][rows][cols] response of the key-points that survive NMS + border, 0 elsewhere
    uint8_t* blur;        // [B][rows][cols]
    int32_t* hist;        // [B][256]
    int32_t* rowcnt;      // [B][rows] key-points >= cut per row, then exclusive offsets
    int32_t* cut;         // [B]
    float* kp;            // [B][K][2]
    float* resp;          // [B][K]
    float* angle;         // [B][K]
    uint8_t* desc;        // [B][K][32]
    int32_t* n_kp;        // [B]
    const int8_t* pattern;  // [256][4]
};

// circle offsets in OpenCV's order (dx, dy)
__device__ __constant__ int8_t c_circle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                                  {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

__global__ __launch_bounds__(TILE_W* TILE_H) void orb_fast_kernel(OrbDev o) {
    __shared__ uint8_t tile[TILE_H + 6][TILE_W + 8];  // halo of 3 (row padded to a multiple of 4 bytes)
    const int b = blockIdx.z, x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
    const uint8_t* img = o.img + (size_t)b * o.rows * o.cols;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    for (int i = tid; i < (TILE_H + 6) * (TILE_W + 6); i += TILE_W * TILE_H) {
        const int ty = i / (TILE_W + 6), tx = i % (TILE_W + 6);
        const int gx = min(max(x0 + tx - 3, 0), o.cols - 1), gy = min(max(y0 + ty - 3, 0), o.rows - 1);
        tile[ty][tx] = img[(size_t)gy * o.cols + gx];
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= o.cols || y >= o.rows) return;
    int out = 0;
    if (x >= 3 && x < o.cols - 3 && y >= 3 && y < o.rows - 3) {
        const int lx = threadIdx.x + 3, ly = threadIdx.y + 3;
        const int v = tile[ly][lx];
        const int t = o.fast_th;
        // any arc of 9 contiguous circle pixels holds at least two of the four compass points: cheap rejection
        const int c0 = tile[ly + 3][lx] - v, c4 = tile[ly][lx + 3] - v, c8 = tile[ly - 3][lx] - v, c12 = tile[ly][lx - 3] - v;
        const int nb = (c0 > t) + (c4 > t) + (c8 > t) + (c12 > t), nd = (c0 < -t) + (c4 < -t) + (c8 < -t) + (c12 < -t);
        if (nb >= 2 || nd >= 2) {
            int d[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) d[k] = (int)tile[ly + c_circle[k][1]][lx + c_circle[k][0]] - v;
            // min / max over every window of 9 consecutive circle positions by doubling: 2, 4, 8, then + 1
            int mn2[16], mx2[16], mn4[16], mx4[16], mn8[16], mx8[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                mn2[k] = min(d[k], d[(k + 1) & 15]);
                mx2[k] = max(d[k], d[(k + 1) & 15]);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                mn4[k] = min(mn2[k], mn2[(k + 2) & 15]);
                mx4[k] = max(mx2[k], mx2[(k + 2) & 15]);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                mn8[k] = min(mn4[k], mn4[(k + 4) & 15]);
                mx8[k] = max(mx4[k], mx4[(k + 4) & 15]);
            }
            int sb = -255, sd = -255;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                sb = max(sb, min(mn8[k], d[(k + 8) & 15]));
                sd = max(sd, -max(mx8[k], d[(k + 8) & 15]));
            }
            const int s = max(sb, sd) - 1;  // cornerScore<16>
            if (s >= t) out = s > 0 ? s : 0;
        }
    }
    o.score[(size_t)b * o.rows * o.cols + (size_t)y * o.cols + x] = (uint8_t)out;
}

__global__ __launch_bounds__(256) void orb_nms_kernel(OrbDev o) {
    __shared__ int s_hist[256];
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)b * o.rows * o.cols;
    if (x < o.cols) {
        int kept = 0;
        if (x >= 3 && x < o.cols - 3 && y >= 3 && y < o.rows - 3) {
            const uint8_t* sc = o.score + base + (size_t)y * o.cols + x;
            const int s = sc[0];
            if (s > 0) {
                const int c = o.cols;
                const bool is_max = s > sc[-c - 1] && s > sc[-c] && s > sc[-c + 1] && s > sc[-1] && s > sc[1] && s > sc[c - 1] && s > sc[c] &&
                                    s > sc[c + 1];
                // KeyPointsFilter::runByImageBorder
                if (is_max && x >= o.edge_th && x < o.cols - o.edge_th && y >= o.edge_th && y < o.rows - o.edge_th) kept = s;
            }
        }
        o.keep[base + (size_t)y * o.cols + x] = (uint8_t)kept;
        if (kept) atomicAdd(&s_hist[kept], 1);
    }
    __syncthreads();
    if (s_hist[threadIdx.x]) atomicAdd(&o.hist[(size_t)b * 256 + threadIdx.x], s_hist[threadIdx.x]);
}

// retainBest(nfeatures): the smallest response `cut` such that at least nfeatures key-points are >= cut (or 1)
__global__ __launch_bounds__(256) void orb_cut_kernel(OrbDev o) {
    __shared__ int s_h[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    s_h[tid] = o.hist[(size_t)b * 256 + tid];
    __syncthreads();
    if (tid == 0) {
        int cut = 1, acc = 0;
        for (int s = 255; s >= 1; --s) {
            acc += s_h[s];
            if (acc >= o.nfeatures) {
                cut = s;
                break;
            }
        }
        o.cut[b] = cut;
    }
    o.hist[(size_t)b * 256 + tid] = 0;  // ready for the next frame
}

// one wave per image row: PASS 0 counts the survivors of the row, PASS 1 writes them at the row's offset in ascending x
template <int PASS>
__global__ __launch_bounds__(64) void orb_emit_kernel(OrbDev o) {
    const int b = blockIdx.y, y = blockIdx.x, lane = threadIdx.x;
    const int cut = o.cut[b];
    const uint8_t* kr = o.keep + (size_t)b * o.rows * o.cols + (size_t)y * o.cols;
    int run = PASS == 1 ? o.rowcnt[(size_t)b * o.rows + y] : 0;
    for (int x0 = 0; x0 < o.cols; x0 += 64) {
        const int x = x0 + lane;
        const int r = x < o.cols ? kr[x] : 0;
        const bool ok = r >= cut && r > 0;
        const unsigned long long bal = __ballot(ok);
        if (PASS == 1 && ok) {
            const int idx = run + __popcll(bal & ((1ull << lane) - 1ull));
            if (idx < o.K) {
                const size_t k = (size_t)b * o.K + idx;
                o.kp[2 * k] = (float)x;
                o.kp[2 * k + 1] = (float)y;
                o.resp[k] = (float)r;
            }
        }
        run += __popcll(bal);
    }
    if (PASS == 0 && lane == 0) o.rowcnt[(size_t)b * o.rows + y] = run;
}

// exclusive scan of the row counts of one image (<= 1024 rows per pass, looped), total -> n_kp (capped at K)
__global__ __launch_bounds__(256) void orb_rowscan_kernel(OrbDev o) {
    __shared__ int s_part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    int32_t* rc = o.rowcnt + (size_t)b * o.rows;
    const int per = (o.rows + 255) / 256;
    const int lo = tid * per, hi = min(lo + per, o.rows);
    int sum = 0;
    for (int y = lo; y < hi; ++y) sum += rc[y];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) {
            const int v = s_part[i];
            s_part[i] = run;
            run += v;
        }
        o.n_kp[b] = run < o.K ? run : o.K;
    }
    __syncthreads();
    int run = s_part[tid];
    for (int y = lo; y < hi; ++y) {
        const int v = rc[y];
        rc[y] = run;
        run += v;
    }
}

__device__ __forceinline__ int reflect101(int p, int n) {
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

struct BlurK {
    int k[7];
};

__global__ __launch_bounds__(TILE_W* TILE_H) void orb_blur_kernel(OrbDev o, BlurK kk) {
    __shared__ uint8_t tile[TILE_H + 6][TILE_W + 8];
    __shared__ int hrow[TILE_H + 6][TILE_W];
    const int b = blockIdx.z, x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
    const uint8_t* img = o.img + (size_t)b * o.rows * o.cols;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    for (int i = tid; i < (TILE_H + 6) * (TILE_W + 6); i += TILE_W * TILE_H) {
        const int ty = i / (TILE_W + 6), tx = i % (TILE_W + 6);
        const int gx = reflect101(x0 + tx - 3, o.cols), gy = reflect101(y0 + ty - 3, o.rows);
        tile[ty][tx] = img[(size_t)gy * o.cols + gx];
    }
    __syncthreads();
    for (int i = tid; i < (TILE_H + 6) * TILE_W; i += TILE_W * TILE_H) {  // horizontal pass (integers, kernel * 2^8)
        const int ty = i / TILE_W, tx = i % TILE_W;
        int s = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) s += kk.k[j] * tile[ty][tx + j];
        hrow[ty][tx] = s;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= o.cols || y >= o.rows) return;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) s += kk.k[j] * hrow[threadIdx.y + j][threadIdx.x];
    s = (s + (1 << 15)) >> 16;
    o.blur[(size_t)b * o.rows * o.cols + (size_t)y * o.cols + x] = (uint8_t)min(max(s, 0), 255);
}

// OpenCV fastAtan2 (degrees): 7th-order odd polynomial on [0, 1], octant folding
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale,
                p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = __fdiv_rn(ax, ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

struct Umax {
    int u[ORB_HP + 2];
};

// one wave per key-point: intensity-centroid angle on the image, rotated BRIEF on the blurred image
__global__ __launch_bounds__(256) void orb_describe_kernel(OrbDev o, Umax um) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= o.n_kp[b]) return;  // wave-uniform
    const size_t kk = (size_t)b * o.K + k;
    const int x = (int)o.kp[2 * kk], y = (int)o.kp[2 * kk + 1];
    const size_t base = (size_t)b * o.rows * o.cols;
    const uint8_t* img = o.img + base;
    // ICAngles: lane u + 15 owns column u of the circular patch; integer moments, so the summation order is free
    int m10 = 0, m01 = 0;
    if (lane <= 2 * ORB_HP) {
        const int u = lane - ORB_HP, au = u < 0 ? -u : u;
        const uint8_t* c = img + (size_t)y * o.cols + (x + u);
        m10 = u * (int)c[0];
        for (int v = 1; v <= ORB_HP; ++v)
            if (au <= um.u[v]) {
                const int vp = c[(ptrdiff_t)v * o.cols], vm = c[-(ptrdiff_t)v * o.cols];
                m10 += u * (vp + vm);
                m01 += v * (vp - vm);
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m10 += __shfl_xor(m10, off, 64);
        m01 += __shfl_xor(m01, off, 64);
    }
    const float ang = fast_atan2_deg((float)m01, (float)m10);
    if (lane == 0) o.angle[kk] = ang;
    // computeOrbDescriptors, WTA_K = 2: lane l evaluates tests 4 l .. 4 l + 3
    const float rad = ang * (float)(3.14159265358979323846 / 180.0);
    const float a = (float)cos((double)rad), sb = (float)sin((double)rad);
    const uint8_t* bl = o.blur + base + (size_t)y * o.cols + x;
    int nib = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int8_t* p = o.pattern + 4 * (4 * lane + t);
        const float px0 = (float)p[0], py0 = (float)p[1], px1 = (float)p[2], py1 = (float)p[3];
        const int ix0 = __float2int_rn(px0 * a - py0 * sb), iy0 = __float2int_rn(px0 * sb + py0 * a);
        const int ix1 = __float2int_rn(px1 * a - py1 * sb), iy1 = __float2int_rn(px1 * sb + py1 * a);
        const int t0 = bl[(ptrdiff_t)iy0 * o.cols + ix0], t1 = bl[(ptrdiff_t)iy1 * o.cols + ix1];
        nib |= (t0 < t1) << t;
    }
    const int hi = __shfl_down(nib, 1, 64);
    if ((lane & 1) == 0) o.desc[kk * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
}

}  // namespace
}  // namespace stvo

struct stvo_orb {
    stvo_ctx* ctx = nullptr;
    stvo::OrbDev d{};
    stvo_orb_params prm{};
    char* dev = nullptr;  // score | keep | blur | hist | rowcnt | cut | pattern
    char* io = nullptr;   // staging for the host-buffer entry point: images in, results out
    size_t io_bytes = 0;
    int8_t pattern[1024];
    stvo::BlurK blur_k{};
    stvo::Umax umax{};
};

namespace {

void default_pattern(int8_t* pattern) {  // seeded stand-in for OpenCV's learned table (see stvo_orb_set_pattern)
    uint32_t s = 31u;
    for (int i = 0; i < 256; ++i) {
        int v[4];
        do {
            for (int j = 0; j < 4; ++j) {
                s = s * 1664525u + 1013904223u;
                v[j] = (int)((s >> 16) % 27u) - 13;
            }
        } while (v[0] == v[2] && v[1] == v[3]);
        for (int j = 0; j < 4; ++j) pattern[4 * i + j] = (int8_t)v[j];
    }
}

}  // namespace

extern "C" {

int stvo_orb_create(stvo_ctx* ctx, int B, int cols, int rows, int max_keypoints, const stvo_orb_params* prm, stvo_orb** out) {
    if (!ctx || !out || !prm || B <= 0 || cols < 64 || rows < 64 || max_keypoints <= 0) return STVO_ERR_INVALID_ARG;
    // the patch (radius 15) and the rotated pattern (|coordinate| <= 13 sqrt 2) must stay inside the image: edge >= 19
    if (prm->nfeatures <= 0 || prm->fast_threshold < 1 || prm->fast_threshold > 254 || prm->edge_threshold < 19 ||
        2 * prm->edge_threshold >= cols || 2 * prm->edge_threshold >= rows)
        return STVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stvo_orb* o = new (std::nothrow) stvo_orb();
    if (!o) return STVO_ERR_HIP;
    o->ctx = ctx;
    o->prm = *prm;
    const size_t px = (size_t)B * rows * cols;
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_score = 0, o_keep = al(px), o_blur = o_keep + al(px), o_hist = o_blur + al(px), o_rowcnt = o_hist + al((size_t)B * 256 * 4),
                 o_cut = o_rowcnt + al((size_t)B * rows * 4), o_pat = o_cut + al((size_t)B * 4), total = o_pat + 1024;
    if (!hip_ok(ctx, hipMalloc((void**)&o->dev, total), "hipMalloc orb") || !hip_ok(ctx, hipMemset(o->dev, 0, total), "hipMemset orb")) {
        if (o->dev) (void)hipFree(o->dev);
        delete o;
        return STVO_ERR_HIP;
    }
    stvo::OrbDev& d = o->d;
    d.B = B; d.cols = cols; d.rows = rows; d.K = max_keypoints;
    d.nfeatures = prm->nfeatures; d.fast_th = prm->fast_threshold; d.edge_th = prm->edge_threshold;
    d.score = (uint8_t*)(o->dev + o_score); d.keep = (uint8_t*)(o->dev + o_keep); d.blur = (uint8_t*)(o->dev + o_blur);
    d.hist = (int32_t*)(o->dev + o_hist); d.rowcnt = (int32_t*)(o->dev + o_rowcnt); d.cut = (int32_t*)(o->dev + o_cut);
    d.pattern = (const int8_t*)(o->dev + o_pat);
    default_pattern(o->pattern);
    if (!hip_ok(ctx, hipMemcpy(o->dev + o_pat, o->pattern, 1024, hipMemcpyHostToDevice), "hipMemcpy pattern")) {
        (void)hipFree(o->dev);
        delete o;
        return STVO_ERR_HIP;
    }
    {   // getGaussianKernel(7, 2) in 8-bit fixed point; umax of ICAngles
        double k[7], sum = 0.0;
        for (int i = 0; i < 7; ++i) {
            const double x = i - 3;
            k[i] = std::exp(-x * x / (2.0 * 2.0 * 2.0));
            sum += k[i];
        }
        for (int i = 0; i < 7; ++i) o->blur_k.k[i] = (int)std::lrint((float)(k[i] / sum) * 256.0);
        const int hp = stvo::ORB_HP;
        const int vmax = (int)std::floor(hp * std::sqrt(2.0) / 2 + 1), vmin = (int)std::ceil(hp * std::sqrt(2.0) / 2);
        for (int v = 0; v <= vmax; ++v) o->umax.u[v] = (int)std::lrint(std::sqrt((double)hp * hp - (double)v * v));
        for (int v = hp, v0 = 0; v >= vmin; --v) {
            while (o->umax.u[v0] == o->umax.u[v0 + 1]) ++v0;
            o->umax.u[v] = v0;
            ++v0;
        }
    }
    *out = o;
    return STVO_OK;
}

int stvo_orb_destroy(stvo_orb* o) {
    if (!o) return STVO_OK;
    (void)hipSetDevice(o->ctx->device);
    (void)hipStreamSynchronize(o->ctx->stream);
    if (o->dev) (void)hipFree(o->dev);
    if (o->io) (void)hipFree(o->io);
    delete o;
    return STVO_OK;
}

int stvo_orb_set_pattern(stvo_orb* o, const int8_t* pattern) {
    if (!o || !pattern) return STVO_ERR_INVALID_ARG;
    for (int i = 0; i < 1024; ++i)
        if (pattern[i] < -13 || pattern[i] > 13) return STVO_ERR_INVALID_ARG;  // rotated points must stay within the border
    HIP_TRY(o->ctx, hipSetDevice(o->ctx->device));
    HIP_TRY(o->ctx, hipStreamSynchronize(o->ctx->stream));
    std::memcpy(o->pattern, pattern, 1024);
    HIP_TRY(o->ctx, hipMemcpy(const_cast<int8_t*>(o->d.pattern), pattern, 1024, hipMemcpyHostToDevice));
    return STVO_OK;
}

int stvo_orb_get_pattern(const stvo_orb* o, int8_t* pattern) {
    if (!o || !pattern) return STVO_ERR_INVALID_ARG;
    std::memcpy(pattern, o->pattern, 1024);
    return STVO_OK;
}

int stvo_orb_detect_dev(stvo_orb* o, const uint8_t* images, float* kp_xy, float* response, float* angle, uint8_t* desc,
                        int32_t* n_kp) {
    if (!o || !images || !kp_xy || !response || !angle || !desc || !n_kp) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stvo::OrbDev d = o->d;
    d.img = images; d.kp = kp_xy; d.resp = response; d.angle = angle; d.desc = desc; d.n_kp = n_kp;
    hipStream_t s = ctx->stream;
    const dim3 tiles((d.cols + stvo::TILE_W - 1) / stvo::TILE_W, (d.rows + stvo::TILE_H - 1) / stvo::TILE_H, d.B), tb(stvo::TILE_W, stvo::TILE_H);
    hipLaunchKernelGGL(stvo::orb_fast_kernel, tiles, tb, 0, s, d);
    hipLaunchKernelGGL(stvo::orb_blur_kernel, tiles, tb, 0, s, d, o->blur_k);
    hipLaunchKernelGGL(stvo::orb_nms_kernel, dim3((d.cols + 255) / 256, d.rows, d.B), dim3(256), 0, s, d);
    hipLaunchKernelGGL(stvo::orb_cut_kernel, dim3(d.B), dim3(256), 0, s, d);
    hipLaunchKernelGGL(stvo::orb_emit_kernel<0>, dim3(d.rows, d.B), dim3(64), 0, s, d);
    hipLaunchKernelGGL(stvo::orb_rowscan_kernel, dim3(d.B), dim3(256), 0, s, d);
    hipLaunchKernelGGL(stvo::orb_emit_kernel<1>, dim3(d.rows, d.B), dim3(64), 0, s, d);
    hipLaunchKernelGGL(stvo::orb_describe_kernel, dim3((d.K + 3) / 4, d.B), dim3(256), 0, s, d, o->umax);
    return check_launch(ctx);
}

int stvo_orb_detect(stvo_orb* o, const uint8_t* images, float* kp_xy, float* response, float* angle, uint8_t* desc, int32_t* n_kp) {
    if (!o || !images || !kp_xy || !response || !angle || !desc || !n_kp) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const stvo::OrbDev& d = o->d;
    const size_t px = (size_t)d.B * d.rows * d.cols, nk = (size_t)d.B * d.K;
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_img = 0, o_kp = al(px), o_resp = o_kp + al(nk * 8), o_ang = o_resp + al(nk * 4), o_desc = o_ang + al(nk * 4),
                 o_n = o_desc + al(nk * 32), total = o_n + al((size_t)d.B * 4);
    if (o->io_bytes < total) {
        if (o->io) (void)hipFree(o->io);
        o->io = nullptr;
        o->io_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&o->io, total));
        o->io_bytes = total;
    }
    char* D = o->io;
    HIP_TRY(ctx, hipMemcpyAsync(D + o_img, images, px, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(D + o_kp, 0, total - o_kp, ctx->stream));
    TRY(stvo_orb_detect_dev(o, (const uint8_t*)(D + o_img), (float*)(D + o_kp), (float*)(D + o_resp), (float*)(D + o_ang),
                            (uint8_t*)(D + o_desc), (int32_t*)(D + o_n)));
    HIP_TRY(ctx, hipMemcpyAsync(kp_xy, D + o_kp, nk * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(response, D + o_resp, nk * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(angle, D + o_ang, nk * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(desc, D + o_desc, nk * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(n_kp, D + o_n, (size_t)d.B * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return STVO_OK;
}

}  // extern "C"
