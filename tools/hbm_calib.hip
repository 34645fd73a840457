// hbm_calib.hip — what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count?  Four streaming kernels over a buffer far
// beyond the 256 MiB Infinity Cache: 4-byte and 16-byte loads per lane, 4-byte and 16-byte stores per lane, every byte touched once.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_calib tools/hbm_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- tools/hbm_calib      (and once more with WRITE_SIZE): tools/hbm_calib.sh
// The factors (known bytes / counter bytes) go into profiles/r05_hbm_counters.txt and are applied by bench.py's committed_traffic().
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void calib_read4(const uint32_t* __restrict__ p, size_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;  // (never: keeps the loads)
}
__global__ __launch_bounds__(256) void calib_read16(const uint4* __restrict__ p, size_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void calib_write4(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void calib_write16(uint4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
// a scattered 16-byte gather: one 16-byte word out of every 64 bytes (the record gathers of the pose / grid kernels)
__global__ __launch_bounds__(256) void calib_gather16_of_64(const uint4* __restrict__ p, size_t n64, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n64; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i * 4];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB: 8 x the Infinity Cache
    void* buf = nullptr;
    uint32_t* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    const int grid = 256 * 32;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(calib_read4, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(calib_read16, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(calib_gather16_of_64, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 64, sink);
        hipLaunchKernelGGL(calib_write4, dim3(grid), dim3(256), 0, 0, (uint32_t*)buf, bytes / 4);
        hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("bytes per kernel: read4 / read16 / write4 / write16 = %zu, gather16_of_64 = %zu useful of %zu spanned\n", bytes, bytes / 4, bytes);
    return 0;
}
