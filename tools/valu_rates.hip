// valu_rates.hip — measurement tool (not product): per-instruction VALU issue rates on gfx950,
// used to price K1's instruction mix.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
#define DEF_KERNEL(NAME, ASM)                                                                      \
    __global__ __launch_bounds__(256) void k_##NAME(int iters, uint32_t* sink) {                   \
        uint32_t v[CHAINS], a = threadIdx.x * 2654435761u + 1u, b = blockIdx.x * 40503u + 7u;      \
        for (int c = 0; c < CHAINS; ++c) v[c] = a * (c + 3);                                       \
        for (int it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                       \
                _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) {                               \
                    asm volatile(ASM : "+v"(v[c]) : "v"(a), "v"(b));                               \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        uint32_t s = 0;                                                                            \
        for (int c = 0; c < CHAINS; ++c) s ^= v[c];                                                \
        if (s == 0x12345u) sink[0] = s;                                                            \
    }

DEF_KERNEL(xor, "v_xor_b32 %0, %0, %1")
DEF_KERNEL(bcnt, "v_bcnt_u32_b32 %0, %1, %0")
DEF_KERNEL(min, "v_min_u32 %0, %0, %1")
DEF_KERNEL(med3, "v_med3_u32 %0, %0, %1, %2")
DEF_KERNEL(lshl_or, "v_lshl_or_b32 %0, %0, 16, %1")
DEF_KERNEL(add, "v_add_u32 %0, %0, %1")
DEF_KERNEL(add3, "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL(fma32, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(xad, "v_xad_u32 %0, %0, %1, %2")
DEF_KERNEL(bfi, "v_bfi_b32 %0, %0, %1, %2")
DEF_KERNEL(perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEF_KERNEL(pk_min_u16, "v_pk_min_u16 %0, %0, %1")
DEF_KERNEL(dot4_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
DEF_KERNEL(dot8_u4, "v_dot8_u32_u4 %0, %0, %1, %2")
DEF_KERNEL(sad_u8, "v_sad_u8 %0, %0, %1, %2")
DEF_KERNEL(mov_dpp_ror, "v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL(mov_dpp_wave_ror, "v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL(min_dpp, "v_min_u32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")

template <typename F>
static void run(const char* name, F kern, uint32_t* sink) {
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, 10, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 16.0 * CHAINS;
    printf("%-18s %8.3f ms  %7.2f T lane-ops/s  (%.2f lanes/clk/SIMD @2.4GHz)\n", name, ms, ops / (ms * 1e-3) / 1e12,
           ops / (ms * 1e-3) / (256.0 * 4 * 2.4e9));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    uint32_t* sink;
    hipMalloc(&sink, 256);
#define RUN(NAME) run(#NAME, k_##NAME, sink)
    RUN(xor); RUN(bcnt); RUN(min); RUN(med3); RUN(lshl_or); RUN(add); RUN(add3); RUN(fma32); RUN(and_or); RUN(xad);
    RUN(bfi); RUN(perm); RUN(pk_add_u16); RUN(pk_min_u16); RUN(dot4_u8); RUN(dot8_u4); RUN(sad_u8); RUN(mov_dpp_ror);
    RUN(mov_dpp_wave_ror); RUN(min_dpp); RUN(cndmask); RUN(mul_lo); RUN(mad_u24);
    hipFree(sink);
    return 0;
}
