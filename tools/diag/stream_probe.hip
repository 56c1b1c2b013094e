// stream_probe.hip -- which streams' kernels overlap on this device (diagnostic): N streams, one 2 ms kernel each, every
// kernel notes when it started and ended (wall clock); run twice (cold, warm) per N and for two stream-creation patterns.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/stream_probe tools/diag/stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(e)                                                                  \
    do {                                                                          \
        hipError_t r_ = (e);                                                      \
        if (r_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

__global__ void spin_k(unsigned long long cycles, unsigned long long* stamp)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamp[0] = t0;
        stamp[1] = wall_clock64();
    }
}

int main()
{
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    const unsigned long long cycles = (unsigned long long)khz * 2;
    unsigned long long* stamps = nullptr;
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&stamps), 64 * 2 * sizeof(unsigned long long), hipHostMallocDefault));
    for (int pattern = 0; pattern < 3; ++pattern) {
        // pattern 0: only the work streams; 1: two copy streams made first (as rcx_host.hpp does); 2: as 1, and a copy runs on them
        std::vector<hipStream_t> pre(2), ss(8);
        if (pattern >= 1)
            for (auto& s : pre) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto& s : ss) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        char *d = nullptr, *h = nullptr;
        CHECK(hipMalloc(&d, 256 << 20));
        CHECK(hipHostMalloc(&h, 256 << 20, hipHostMallocDefault));
        for (int n : {2, 4, 8}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (pattern == 2) {
                    CHECK(hipMemcpyAsync(d, h, 256 << 20, hipMemcpyHostToDevice, pre[0]));
                    CHECK(hipMemcpyAsync(h, d, 128 << 20, hipMemcpyDeviceToHost, pre[1]));
                }
                for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin_k, dim3(16), dim3(64), 0, ss[i], cycles, stamps + 2 * i);
                CHECK(hipDeviceSynchronize());
                unsigned long long first = ~0ull;
                for (int i = 0; i < n; ++i) first = stamps[2 * i] < first ? stamps[2 * i] : first;
                printf("pattern %d, %d streams, %s:", pattern, n, rep ? "warm" : "cold");
                for (int i = 0; i < n; ++i) printf("  [%.2f, %.2f]", (stamps[2 * i] - first) * 1e3 / khz / 1e3, (stamps[2 * i + 1] - first) * 1e3 / khz / 1e3);
                printf(" ms\n");
            }
        }
        for (auto& s : ss) CHECK(hipStreamDestroy(s));
        if (pattern >= 1)
            for (auto& s : pre) CHECK(hipStreamDestroy(s));
        CHECK(hipFree(d));
        CHECK(hipHostFree(h));
    }
    return 0;
}
