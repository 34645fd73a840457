// Does hipMemset (null stream) return before its fill has run, and does a non-blocking stream overtake it?  (Round 6: the first image
// of a new LSD detector came back empty on some boxes.)  hipMalloc 600 MB, hipMemset 0, then at once a host-to-device copy of 1 MB of
// ones to an offset 40 MB into the buffer on a hipStreamNonBlocking stream; count the zero bytes that arrive back.  Build + run:
//   hipcc --offload-arch=gfx950 -o tools/experiments/memset_race tools/experiments/memset_race.hip && tools/experiments/memset_race
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t total = 600u << 20, off = 40u << 20, n = 1u << 20;
    std::vector<unsigned char> ones(n, 1), back(n);
    int bad_runs = 0;
    for (int rep = 0; rep < 20; ++rep) {
        char* d = nullptr;
        hipMalloc((void**)&d, total);
        hipMemset(d, 0, total);
        hipMemcpyAsync(d + off, ones.data(), n, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        hipDeviceSynchronize();
        hipMemcpy(back.data(), d + off, n, hipMemcpyDeviceToHost);
        size_t zeros = 0;
        for (size_t i = 0; i < n; ++i) zeros += back[i] == 0;
        if (zeros) ++bad_runs;
        printf("rep %2d: %zu of %zu bytes zeroed after the copy\n", rep, zeros, n);
        hipFree(d);
    }
    printf("runs in which the fill overtook the copy: %d of 20\n", bad_runs);
    return 0;
}
