// mfma_rates.hip — issue-rate probe for v_mfma_i32_32x32x32_i8 on gfx950: TOP/s as a function of the number of
// independent accumulator chains per wave and of resident waves per SIMD (register-only operands, no memory traffic).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rates tools/mfma_rates.hip && /tmp/mfma_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// DENSE: operands are pseudo-random +-64 bytes (what K1m feeds the instruction) instead of a few small integers — the
// switching activity of the multiplier array, hence power and the sustained clock, depends on the data.
template <int CHAINS, bool DENSE>
__global__ __launch_bounds__(256) void probe(int iters, int* sink) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    if (DENSE) {
        unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            a[k] = (int)((h & 0x80808080u) | 0x40404040u);
            h = h * 1664525u + 1013904223u;
            b[k] = (int)((h & 0x80808080u) | 0x40404040u);
        }
    }
    v16i acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = c + r;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[c], 0, 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 0x7fffffff) sink[0] = s;
}

template <int CHAINS, bool DENSE>
static void run(int waves_per_simd, int* sink) {
    // 256 CUs x 4 SIMDs; a workgroup of 256 threads puts one wave on each SIMD of a CU
    const int blocks = 256 * waves_per_simd, iters = 20000;  // ~10 ms per launch: long enough for the clock to settle
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<CHAINS, DENSE>), dim3(blocks), dim3(256), 0, 0, 10, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<CHAINS, DENSE>), dim3(blocks), dim3(256), 0, 0, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 * iters * 8 * CHAINS * 2.0 * 32 * 32 * 32;
    printf("%s operands  chains/wave %d  waves/SIMD %d : %.0f TOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", DENSE ? "dense +-64" : "sparse     ", CHAINS, waves_per_simd,
           ops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 8 * CHAINS));
}

int main() {
    int* sink;
    (void)hipMalloc(&sink, 64);
    for (int w = 1; w <= 4; ++w) {
        run<2, false>(w, sink);
        run<2, true>(w, sink);
        run<4, true>(w, sink);
    }
    return 0;
}
