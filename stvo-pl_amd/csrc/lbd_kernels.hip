// lbd_kernels.hip — the LBD line descriptor on gfx950 (first half): what the reference obtains from
//     Ptr<BinaryDescriptor> lbd = BinaryDescriptor::createBinaryDescriptor();  lbd->compute(img, lines, ldesc);
// in StereoFrame::detectLineFeatures (/root/reference/src/stereoFrame.cpp:207-243,303) for key-lines that already exist (octave 0:
// the reference detects on one octave, :230).  The algorithm is the reference-held
// 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp (computeGaussianPyramid / computeSobel :350-398, computeLBD
// :1026-1340, binaryConversion :401-412, the weight tables of the constructor :217-258), restated in oracle/stvo_lbd_oracle.c,
// against which these kernels are bit-exact (tests/test_gpu_lbd.py).  The LSD / FLD detectors that produce the key-lines are
// not built.
//   launch_blur7_u8      GaussianBlur(5 x 5, sigma 1) in OpenCV 3's 8-bit fixed point (weights x 2^8, one rounding shift by 16): orb_kernels.hip's
//                        register-resident 7-tap blur with zero outer weights
//   lbd_sobel_kernel     Sobel 3 x 3 of the blurred image, BORDER_REFLECT_101, four pixels per thread: int16 (dx, dy) in one word per pixel
//   lbd_describe_kernel  one wave per key-line: lane = row of the 63-row line support region.  The row sums are FLOAT sums in the
//                        source's order (a sequential walk along the line per row — the rows are the parallelism), the 9 band
//                        statistics are accumulated row by row in the source's order by the lane that owns the band, and the
//                        two normalisations are serial sums like the source's; no fused multiply-adds.
#include <cmath>
#include <cstring>
#include <new>

#include "ctx_internal.h"

#pragma clang fp contract(off)

namespace stvo {
namespace {

constexpr int LBD_BANDS = 9, LBD_W = 7, LBD_ROWS = LBD_BANDS * LBD_W, LBD_DESC = LBD_BANDS * 8;

struct LbdDev {
    int B, cols, rows, M;        // images, image size, key-line capacity per image
    const uint8_t* img;          // [B][rows][cols]
    uint8_t* blur;               // [B][rows][cols]
    int32_t* dxy;                // [B][rows][cols] dx (low half) and dy (high half) of a pixel as int16 in ONE word: the support region is walked with one gather per step
    const stvo_keyline* lines;   // [B][M]
    const int32_t* n_lines;      // [B]
    uint8_t* desc;               // [B][M][32]
    float* desc_f;               // [B][M][72] or nullptr
    int k5[5];                   // blur weights x 2^8
    float coefG[LBD_ROWS];       // (float) gaussCoefG_[hID]
    float coefL[3 * LBD_W];      // (float) gaussCoefL_[k]
};

__device__ __forceinline__ int reflect101(int p, int n) {
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

// Sobel 3 x 3 (BORDER_REFLECT_101) of the blurred image: a thread produces FOUR adjacent pixels of SB_R consecutive rows.  Per input
// row it loads the six bytes x - 1 .. x + 4 as two (unaligned) words — byte by byte with reflected columns only at the two image
// borders —, forms the row's four differences p[i + 1] - p[i - 1] and four smoothed values p[i - 1] + 2 p[i] + p[i + 1], keeps three
// rows of them in registers and emits one output row per input row: (dx, dy) as the int16 halves of one word per pixel.
// (Round 6.  Until then one thread per pixel with eight cached byte loads, behind a 5 x 5 blur of the same kind: 14 ms per 4096
// KITTI-size images; the blur is now orb_kernels.hip's register-resident 7-tap kernel with zero outer weights — the same 8-bit
// fixed-point arithmetic —, together 2.x ms.)
typedef uint32_t __attribute__((aligned(1))) lbd_u32_unaligned;
typedef int32_t lbd_i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
constexpr int SB_R = 16, SB_T = 64;
__global__ __launch_bounds__(SB_T) void lbd_sobel_kernel(LbdDev o) {
    const int b = blockIdx.z, x = (blockIdx.x * SB_T + threadIdx.x) * 4, y0 = blockIdx.y * SB_R;
    if (x >= o.cols) return;
    const uint8_t* img = o.blur + (size_t)b * o.rows * o.cols;
    int32_t* out = o.dxy + (size_t)b * o.rows * o.cols;
    const bool inner = x >= 1 && x + 5 <= o.cols;
    int xr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) xr[i] = reflect101(x - 1 + i, o.cols);
    int hx[3][4], sx[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) hx[j][i] = sx[j][i] = 0;
#pragma unroll
    for (int r = 0; r < SB_R + 2; ++r) {
        const int yo = y0 + r - 2;  // the output row completed by this input row
        if (r >= 2 && yo >= o.rows) break;  // uniform over the workgroup
        const uint8_t* row = img + (size_t)reflect101(y0 + r - 1, o.rows) * o.cols;
        int p[6];
        if (inner) {
            const uint32_t wa = *reinterpret_cast<const lbd_u32_unaligned*>(row + x - 1), wb = *reinterpret_cast<const lbd_u32_unaligned*>(row + x + 1);
            p[0] = (int)(wa & 0xFFu); p[1] = (int)((wa >> 8) & 0xFFu); p[2] = (int)((wa >> 16) & 0xFFu); p[3] = (int)(wa >> 24);
            p[4] = (int)((wb >> 16) & 0xFFu); p[5] = (int)(wb >> 24);
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) p[i] = (int)row[xr[i]];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hx[0][i] = hx[1][i]; hx[1][i] = hx[2][i];
            sx[0][i] = sx[1][i]; sx[1][i] = sx[2][i];
            hx[2][i] = p[i + 2] - p[i];
            sx[2][i] = p[i] + 2 * p[i + 1] + p[i + 2];
        }
        if (r >= 2) {
            int32_t* dst = out + (size_t)yo * o.cols + x;
            int32_t g[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gx = hx[0][i] + 2 * hx[1][i] + hx[2][i], gy = sx[2][i] - sx[0][i];
                g[i] = (int32_t)(((uint32_t)gx & 0xFFFFu) | ((uint32_t)gy << 16));  // (|g| <= 1020: int16 halves)
            }
            if (x + 4 <= o.cols) {  // one 16-byte store (word-aligned: the rows of an image of odd width are not 16-byte aligned)
                lbd_i32x4_a4 v;
                v.x = g[0]; v.y = g[1]; v.z = g[2]; v.w = g[3];
                *reinterpret_cast<lbd_i32x4_a4*>(dst) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (x + i < o.cols) dst[i] = g[i];
            }
        }
    }
}

constexpr int LBD_LINES_PER_WG = 4;
__global__ __launch_bounds__(64 * LBD_LINES_PER_WG) void lbd_describe_kernel(LbdDev o) {
    __shared__ float s_row[LBD_LINES_PER_WG][LBD_ROWS][8];
    __shared__ float s_des[LBD_LINES_PER_WG][LBD_DESC];
    __shared__ float s_norm[LBD_LINES_PER_WG][2];
    const int b = blockIdx.y, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l = blockIdx.x * LBD_LINES_PER_WG + wv;
    const int n = min(max(o.n_lines[b], 0), o.M);
    if (l >= n) return;  // wave-uniform (no workgroup barrier below: every synchronisation is inside the wave)
    const stvo_keyline kl = o.lines[(size_t)b * o.M + l];
    const int32_t* pdxy = o.dxy + (size_t)b * o.rows * o.cols;
    float (*row)[8] = s_row[wv];
    float* des = s_des[wv];
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // ---- computeLBD :1101-1135
    const int imageWidth = o.cols - 1, imageHeight = o.rows - 1, realWidth = o.cols;
    const int halfHeight = (LBD_ROWS - 1) / 2;
    const int lengthOfLSP = (int)(short)kl.num_pixels, halfWidth = (lengthOfLSP - 1) / 2;
    const float midX = (float)(0.5 * (double)(kl.sx + kl.ex)), midY = (float)(0.5 * (double)(kl.sy + kl.ey));
    const float dL0 = (float)cos((double)kl.angle), dL1 = (float)sin((double)kl.angle);
    const float dO0 = -dL1, dO1 = dL0;
    float sCorX0 = -dL0 * (float)halfWidth + dL1 * (float)halfHeight + midX;
    float sCorY0 = -dL1 * (float)halfWidth - dL0 * (float)halfHeight + midY;
    // lane hID's row origin: the source steps the origin once per row (:1176-1177), hID roundings deep
    const int hID = lane;
    for (int k = 0; k < LBD_ROWS - 1; ++k)
        if (k < hID) {
            sCorX0 -= dL1;
            sCorY0 += dL0;
        }
    if (hID < LBD_ROWS) {  // ---- one row of the support region (:1144-1175), sequential along the line
        float sCorX = sCorX0, sCorY = sCorY0;
        float pgdL = 0.f, ngdL = 0.f, pgdO = 0.f, ngdO = 0.f;
#pragma unroll 8
        for (int wID = 0; wID < lengthOfLSP; ++wID) {  // (unrolled: the gathers of eight steps in flight — their addresses do not depend on what is loaded; the sums stay in order)
            int t = (int)roundf(sCorX);
            const int xCor = t < 0 ? 0 : (t > imageWidth ? imageWidth : t);
            t = (int)roundf(sCorY);
            const int yCor = t < 0 ? 0 : (t > imageHeight ? imageHeight : t);
            const int g = pdxy[yCor * realWidth + xCor];
            const float dx = (float)(int16_t)(g & 0xFFFF), dy = (float)(g >> 16);
            const float gDL = dx * dL0 + dy * dL1;
            const float gDO = dx * dO0 + dy * dO1;
            if (gDL > 0) pgdL += gDL; else ngdL -= gDL;
            if (gDO > 0) pgdO += gDO; else ngdO -= gDO;
            sCorX += dL0;
            sCorY += dL1;
        }
        const float cg = o.coefG[hID];  // :1178-1186
        pgdL = cg * pgdL;
        ngdL = cg * ngdL;
        pgdO = cg * pgdO;
        ngdO = cg * ngdO;
        row[hID][0] = pgdL; row[hID][1] = ngdL; row[hID][2] = pgdL * pgdL; row[hID][3] = ngdL * ngdL;
        row[hID][4] = pgdO; row[hID][5] = ngdO; row[hID][6] = pgdO * pgdO; row[hID][7] = ngdO * ngdO;
    }
    wave_sync();
    // ---- band statistics (:1188-1225): band bd receives, in row order, the rows of band bd - 1 (weights gaussCoefL_[r]), its own
    // rows (gaussCoefL_[r + w]) and the rows of band bd + 1 (gaussCoefL_[r + 2 w]) — exactly the order in which the source's
    // row loop reaches it
    if (lane < LBD_BANDS) {
        const int bd = lane;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int h = (bd - 1) * LBD_W; h < (bd + 2) * LBD_W; ++h) {
            if (h < 0 || h >= LBD_ROWS) continue;
            const int rb = h / LBD_W, r = h - rb * LBD_W;
            const float c = o.coefL[rb == bd - 1 ? r : (rb == bd ? r + LBD_W : r + 2 * LBD_W)];
            acc[0] += c * row[h][0];
            acc[1] += c * row[h][1];
            acc[2] += c * c * row[h][2];
            acc[3] += c * c * row[h][3];
            acc[4] += c * row[h][4];
            acc[5] += c * row[h][5];
            acc[6] += c * c * row[h][6];
            acc[7] += c * c * row[h][7];
        }
        // :1231-1262 — mean and standard deviation per band
        const float invN2 = (float)(1.0 / (LBD_W * 2.0)), invN3 = (float)(1.0 / (LBD_W * 3.0));
        const float invN = (bd == 0 || bd == LBD_BANDS - 1) ? invN2 : invN3;
        float* d = des + 8 * bd;
        float t = acc[0] * invN;
        d[0] = t;
        d[4] = sqrtf(acc[2] * invN - t * t);
        t = acc[1] * invN;
        d[1] = t;
        d[5] = sqrtf(acc[3] * invN - t * t);
        t = acc[4] * invN;
        d[2] = t;
        d[6] = sqrtf(acc[6] * invN - t * t);
        t = acc[5] * invN;
        d[3] = t;
        d[7] = sqrtf(acc[7] * invN - t * t);
    }
    wave_sync();
    if (lane == 0) {  // :1265-1281 — serial sums in the source's order
        float tM = 0.f, tS = 0.f;
        for (int base = 0; base < LBD_BANDS; ++base) {
            const float* d = des + 8 * base;
            tM += d[0] * d[0]; tM += d[1] * d[1]; tM += d[2] * d[2]; tM += d[3] * d[3];
            tS += d[4] * d[4]; tS += d[5] * d[5]; tS += d[6] * d[6]; tS += d[7] * d[7];
        }
        s_norm[wv][0] = 1 / sqrtf(tM);
        s_norm[wv][1] = 1 / sqrtf(tS);
    }
    wave_sync();
    for (int i = lane; i < LBD_DESC; i += 64) {  // :1283-1310
        float v = des[i] * s_norm[wv][(i & 4) ? 1 : 0];
        if ((double)v > 0.4) v = (float)0.4;
        des[i] = v;
    }
    wave_sync();
    if (lane == 0) {  // :1313-1318
        float t = 0.f;
        for (int i = 0; i < LBD_DESC; ++i) t += des[i] * des[i];
        s_norm[wv][0] = 1 / sqrtf(t);
    }
    wave_sync();
    for (int i = lane; i < LBD_DESC; i += 64) des[i] = des[i] * s_norm[wv][0];  // :1320-1323
    wave_sync();
    const size_t k = (size_t)b * o.M + l;
    if (o.desc_f)
        for (int i = lane; i < LBD_DESC; i += 64) o.desc_f[k * LBD_DESC + i] = des[i];
    if (lane < 32) {  // computeImpl :655-659: byte c compares bands combinations[c][0] and [c][1] (binaryConversion :401-412)
        // the 32 pairs (i, j), i < j, in the order of the source's table (:74-108): bands 0 and 1 pair with up to 6, the others with up to 8
        int c = lane, i = 0, j = 0;
        const int cnt[8] = {6, 5, 6, 5, 4, 3, 2, 1};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (c >= 0 && c < cnt[q]) {
                i = q;
                j = q + 1 + c;
                c = -1;
            } else if (c >= 0) {
                c -= cnt[q];
            }
        }
        const float* f1 = des + 8 * i;
        const float* f2 = des + 8 * j;
        unsigned r = 0u;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (f1[t] > f2[t]) r += 1u << t;
        o.desc[k * 32 + lane] = (uint8_t)r;
    }
}

}  // namespace
}  // namespace stvo

struct stvo_lbd {
    stvo_ctx* ctx = nullptr;
    stvo::LbdDev d{};
    char* dev = nullptr;  // blur | (dx, dy)
    char* io = nullptr;   // staging of the host-buffer entry point
    size_t io_bytes = 0;
};

extern "C" {

int stvo_lbd_create(stvo_ctx* ctx, int B, int cols, int rows, int max_keylines, stvo_lbd** out) {
    if (!ctx || !out || B <= 0 || cols < 8 || rows < 8 || max_keylines <= 0) return STVO_ERR_INVALID_ARG;
    if (cols > 32767 || rows > 32767) return STVO_ERR_CAPACITY;  // the source walks the support region in shorts
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stvo_lbd* o = new (std::nothrow) stvo_lbd();
    if (!o) return STVO_ERR_HIP;
    o->ctx = ctx;
    const size_t px = (size_t)B * rows * cols;
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_dxy = al(px), total = o_dxy + al(px * 4);
    if (!hip_ok(ctx, hipMalloc((void**)&o->dev, total), "hipMalloc lbd")) {
        delete o;
        return STVO_ERR_HIP;
    }
    stvo::LbdDev& d = o->d;
    d.B = B; d.cols = cols; d.rows = rows; d.M = max_keylines;
    d.blur = (uint8_t*)o->dev; d.dxy = (int32_t*)(o->dev + o_dxy);
    {   // getGaussianKernel(5, 1) in 8-bit fixed point (binary_descriptor_custom.cpp:358)
        double k[5], sum = 0.0;
        for (int i = 0; i < 5; ++i) {
            const double x = i - 2;
            k[i] = std::exp(-x * x / (2.0 * 1.0 * 1.0));
            sum += k[i];
        }
        for (int i = 0; i < 5; ++i) d.k5[i] = (int)std::lrint((float)(k[i] / sum) * 256.0);
    }
    {   // the weight tables of BinaryDescriptor's constructor (:225-257), integer divisions included
        const int w = stvo::LBD_W;
        double u = (w * 3 - 1) / 2, sigma = (w * 2 + 1) / 2, inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < w * 3; ++i) {
            const double dis = i - u;
            d.coefL[i] = (float)std::exp(dis * dis * inv);
        }
        u = (stvo::LBD_BANDS * w - 1) / 2;
        sigma = u;
        inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < stvo::LBD_ROWS; ++i) {
            const double dis = i - u;
            d.coefG[i] = (float)std::exp(dis * dis * inv);
        }
    }
    *out = o;
    return STVO_OK;
}

int stvo_lbd_destroy(stvo_lbd* o) {
    if (!o) return STVO_OK;
    (void)hipSetDevice(o->ctx->device);
    (void)hipStreamSynchronize(o->ctx->stream);
    if (o->dev) (void)hipFree(o->dev);
    if (o->io) (void)hipFree(o->io);
    delete o;
    return STVO_OK;
}

int stvo_lbd_compute_dev(stvo_lbd* o, const uint8_t* images, const stvo_keyline* lines, const int32_t* n_lines, uint8_t* desc,
                         float* desc_float) {
    if (!o || !images || !lines || !n_lines || !desc) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stvo::LbdDev d = o->d;
    d.img = images; d.lines = lines; d.n_lines = n_lines; d.desc = desc; d.desc_f = desc_float;
    hipStream_t s = ctx->stream;
    {   // GaussianBlur(5 x 5, sigma 1): the shared 7-tap blur with zero outer weights (the same fixed-point arithmetic: weights x 2^8 per pass, (s + 2^15) >> 16)
        const int k7[7] = {0, d.k5[0], d.k5[1], d.k5[2], d.k5[3], d.k5[4], 0};
        stvo::launch_blur7_u8(s, d.B, d.cols, d.rows, images, d.blur, k7);
    }
    hipLaunchKernelGGL(stvo::lbd_sobel_kernel, dim3((d.cols + 4 * stvo::SB_T - 1) / (4 * stvo::SB_T), (d.rows + stvo::SB_R - 1) / stvo::SB_R, d.B), dim3(stvo::SB_T), 0, s, d);
    hipLaunchKernelGGL(stvo::lbd_describe_kernel, dim3((d.M + stvo::LBD_LINES_PER_WG - 1) / stvo::LBD_LINES_PER_WG, d.B),
                       dim3(64 * stvo::LBD_LINES_PER_WG), 0, s, d);
    return check_launch(ctx);
}

int stvo_lbd_compute(stvo_lbd* o, const uint8_t* images, const stvo_keyline* lines, const int32_t* n_lines, uint8_t* desc, float* desc_float) {
    if (!o || !images || !lines || !n_lines || !desc) return STVO_ERR_INVALID_ARG;
    stvo_ctx* ctx = o->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const stvo::LbdDev& d = o->d;
    const size_t px = (size_t)d.B * d.rows * d.cols, nl = (size_t)d.B * d.M;
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_img = 0, o_ln = al(px), o_n = o_ln + al(nl * sizeof(stvo_keyline)), o_desc = o_n + al((size_t)d.B * 4), o_f = o_desc + al(nl * 32),
                 total = o_f + al(nl * 72 * 4);
    if (o->io_bytes < total) {
        if (o->io) (void)hipFree(o->io);
        o->io = nullptr;
        o->io_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&o->io, total));
        o->io_bytes = total;
    }
    char* D = o->io;
    HIP_TRY(ctx, hipMemcpyAsync(D + o_img, images, px, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(D + o_ln, lines, nl * sizeof(stvo_keyline), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(D + o_n, n_lines, (size_t)d.B * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(D + o_desc, 0, total - o_desc, ctx->stream));
    TRY(stvo_lbd_compute_dev(o, (const uint8_t*)(D + o_img), (const stvo_keyline*)(D + o_ln), (const int32_t*)(D + o_n), (uint8_t*)(D + o_desc),
                             desc_float ? (float*)(D + o_f) : nullptr));
    HIP_TRY(ctx, hipMemcpyAsync(desc, D + o_desc, nl * 32, hipMemcpyDeviceToHost, ctx->stream));
    if (desc_float) HIP_TRY(ctx, hipMemcpyAsync(desc_float, D + o_f, nl * 72 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return STVO_OK;
}

}  // extern "C"
