// stream_probe2.hip -- how kernels of several streams share the device when they look like this repo's chunk kernels
// (diagnostic): issue-bound work instead of a clock spin (so sharing a SIMD shows as a longer kernel), a large LDS
// allocation per workgroup, optionally a dependent second kernel per stream.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/stream_probe2 tools/diag/stream_probe2.hip && /tmp/stream_probe2
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

extern __shared__ unsigned lds[];

__global__ void work_k(unsigned iters, unsigned long long* stamp, unsigned* sink, unsigned* where)
{
    const unsigned long long t0 = wall_clock64();
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7;
    for (unsigned i = 0; i < iters; ++i) {
        a = a * 1664525u + 1013904223u;
        b = b * 22695477u + 1u;
        c = c * 1103515245u + 12345u;
        d = d * 134775813u + 1u;
    }
    lds[threadIdx.x] = a + b + c + d;
    __syncthreads();
    if (threadIdx.x == 0) {
        sink[blockIdx.x & 1023] = lds[(a >> 8) % blockDim.x];
        unsigned xcc = 0, hw = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        where[blockIdx.x] = (xcc & 0xF) << 16 | (hw & 0xFFFF); // XCC, and HW_ID (cu id bits 8..11, sh 12, se 13..15)
        if (blockIdx.x == 0) {
            stamp[0] = t0;
            stamp[1] = wall_clock64();
        }
    }
}

int main()
{
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    unsigned long long* stamps = nullptr;
    unsigned *sink = nullptr, *where = nullptr;
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&stamps), 64 * 2 * sizeof(unsigned long long), hipHostMallocDefault));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&where), 8 * 64 * sizeof(unsigned), hipHostMallocDefault));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(work_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<hipStream_t> ss(8);
    for (auto& s : ss) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Shape { const char* name; unsigned wgs, threads, lds; int followers; };
    const Shape shapes[] = {{"16 wg x 320 thr, 142 KiB LDS (encode chunk)", 16, 320, 142 * 1024, 0},
                            {"16 wg x 320 thr, 142 KiB LDS + 2 dependent small kernels", 16, 320, 142 * 1024, 2},
                            {"32 wg x 256 thr, 78 KiB LDS (decode chunk)", 32, 256, 78 * 1024, 0},
                            {"32 wg x 256 thr, 4 KiB LDS", 32, 256, 4 * 1024, 0},
                            {"64 wg x 256 thr, 78 KiB LDS", 64, 256, 78 * 1024, 0}};
    const unsigned iters = 400000; // ~ 4 x 4 cycles x 400k = 6.4 M cycles ~ 2.7 ms for a wave alone on its SIMD
    for (const Shape& sh : shapes) {
        for (int n : {1, 2, 3, 4, 8}) {
            for (int rep = 0; rep < 2; ++rep) {
                for (int i = 0; i < n; ++i) {
                    hipLaunchKernelGGL(work_k, dim3(sh.wgs), dim3(sh.threads), sh.lds, ss[i], iters, stamps + 2 * i, sink, where + 64 * i);
                    for (int f = 0; f < sh.followers; ++f) hipLaunchKernelGGL(work_k, dim3(1), dim3(64), 1024, ss[i], 10u, stamps + 32 + 2 * i, sink, where + 64 * i + 63);
                }
                CHECK(hipDeviceSynchronize());
                if (!rep) continue;
                unsigned long long first = ~0ull;
                for (int i = 0; i < n; ++i) first = stamps[2 * i] < first ? stamps[2 * i] : first;
                printf("%s, %d streams:", sh.name, n);
                for (int i = 0; i < n; ++i) printf("  [%.2f, %.2f]", (stamps[2 * i] - first) * 1.0 / khz, (stamps[2 * i + 1] - first) * 1.0 / khz);
                printf(" ms\n");
                if (n == 4 || n == 2) { // where the workgroups of each kernel ran: XCC.SE.CU lists
                    for (int i = 0; i < n; ++i) {
                        printf("    stream %d:", i);
                        for (unsigned w = 0; w < sh.wgs && w < 32; ++w) {
                            const unsigned v = where[64 * i + w];
                            printf(" %u.%u.%u", v >> 16, (v >> 13) & 7, (v >> 8) & 15);
                        }
                        printf("\n");
                    }
                }
            }
        }
    }
    return 0;
}
