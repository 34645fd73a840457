// fp4_probe.hip — what K1m needs to know about v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands on gfx950, measured:
//   1. correctness of the operand convention K1m relies on: lane l feeds row / column (l & 31) of its operand and the 32 nibbles of
//      its four dwords are the k slice 32 (l >> 5) .. + 31 IN THE SAME ORDER FOR BOTH OPERANDS (the contraction is a sum over k, so
//      any order works as long as A and B agree); D[row of A][column of B] lands in the standard 32x32 layout
//      (row = (r & 3) + 8 (r >> 2) + 4 (l >> 5), column = l & 31);  values +-4 (0b0110 / 0b1110), scales 1.0, exact f32 sums;
//   2. the C operand passes an exact f32 through (the tile-relative train index of K1m's keys);
//   3. issue rate against v_mfma_i32_32x32x32_i8 on the same amount of 256-bit rows (register-only operands).
//   hipcc --offload-arch=gfx950 -O3 -o tools/fp4_probe tools/fp4_probe.hip && tools/fp4_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A: [32][64] nibbles as bytes (0 / 1 = +4 / -4), B: [64][32]; one wave
__global__ void gemm_fp4(const unsigned char* A, const unsigned char* B, float* D) {
    const int l = threadIdx.x, rc = l & 31, hf = l >> 5;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    for (int t = 0; t < 32; ++t) {
        const int k = 32 * hf + t;
        const unsigned na = A[rc * 64 + k] ? 0xEu : 0x6u, nb = B[k * 32 + rc] ? 0xEu : 0x6u;
        a[t >> 3] |= (int)(na << (4 * (t & 7)));
        b[t >> 3] |= (int)(nb << (4 * (t & 7)));
    }
    v16f c;
    for (int r = 0; r < 16; ++r) c[r] = (float)((r & 3) + 8 * (r >> 2) + 4 * hf);  // the row index through the C operand
    const v16f d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 127, 0, 127);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + rc] = d[r];
}

template <int CHAINS, bool FP4>
__global__ __launch_bounds__(256) void rate(int iters, float* sink) {
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    v8i a, b;
    for (int k = 0; k < 8; ++k) {
        h = h * 1664525u + 1013904223u;
        a[k] = FP4 ? (int)((h & 0x88888888u) | 0x66666666u) : (int)((h & 0x80808080u) | 0x40404040u);
        h = h * 1664525u + 1013904223u;
        b[k] = FP4 ? (int)((h & 0x88888888u) | 0x66666666u) : (int)((h & 0x80808080u) | 0x40404040u);
    }
    float s = 0.f;
    if (FP4) {
        v16f acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c)
            for (int r = 0; r < 16; ++r) acc[c][r] = (float)(c + r);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)  // 4 x K = 64: one 256-bit row pair
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[c], 4, 4, 0, 127, 0, 127);
        }
        for (int c = 0; c < CHAINS; ++c)
            for (int r = 0; r < 16; ++r) s += acc[c][r];
    } else {
        v16i acc[CHAINS];
        const v4i a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
        for (int c = 0; c < CHAINS; ++c)
            for (int r = 0; r < 16; ++r) acc[c][r] = c + r;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)  // 8 x K = 32
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, acc[c], 0, 0, 0);
        }
        for (int c = 0; c < CHAINS; ++c)
            for (int r = 0; r < 16; ++r) s += (float)acc[c][r];
    }
    if (s == 123456.789f) sink[0] = s;
}

template <int CHAINS, bool FP4>
static void run_rate(int waves_per_simd, float* sink) {
    const int blocks = 256 * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<CHAINS, FP4>), dim3(blocks), dim3(256), 0, 0, 10, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rate<CHAINS, FP4>), dim3(blocks), dim3(256), 0, 0, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double tiles = (double)blocks * 4 * iters * CHAINS;  // 32 x 32 tiles of 256-bit distances
    printf("%s  chains/wave %d  waves/SIMD %d : %.2f G tiles/s = %.0f T pairs/s  (%.1f cycles per 32x32x256-bit tile per SIMD at 2.4 GHz)\n",
           FP4 ? "fp4 4 x 32x32x64 " : "i8  8 x 32x32x32 ", CHAINS, waves_per_simd, tiles / (ms * 1e-3) / 1e9, tiles * 1024 / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * CHAINS));
}

int main() {
    std::vector<unsigned char> A(32 * 64), B(64 * 32);
    srand(7);
    for (auto& v : A) v = rand() & 1;
    for (auto& v : B) v = rand() & 1;
    unsigned char *dA, *dB;
    float* dD;
    (void)hipMalloc(&dA, A.size());
    (void)hipMalloc(&dB, B.size());
    (void)hipMalloc(&dD, 32 * 32 * 4);
    (void)hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gemm_fp4, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<float> D(32 * 32);
    (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float ref = (float)i;
            for (int k = 0; k < 64; ++k) ref += (A[i * 64 + k] ? -4.f : 4.f) * (B[k * 32 + j] ? -4.f : 4.f);
            if (D[i * 32 + j] != ref) {
                if (bad < 8) printf("mismatch D[%d][%d] = %g, expected %g\n", i, j, D[i * 32 + j], ref);
                ++bad;
            }
        }
    printf("fp4 32x32x64 with the K1m operand convention (+-4 x +-4, row index through C): %s (%d of 1024 entries differ)\n", bad ? "WRONG" : "exact", bad);
    float* sink;
    (void)hipMalloc(&sink, 64);
    for (int w = 1; w <= 4; ++w) {
        run_rate<2, false>(w, sink);
        run_rate<2, true>(w, sink);
        run_rate<4, true>(w, sink);
    }
    return bad ? 1 : 0;
}
