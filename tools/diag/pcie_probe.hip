// pcie_probe.hip -- what the host-buffer entry points of rcx.h can hope for on this box (diagnostic, not product):
// link rates with pinned memory (each way, both at once), the runtime's own pageable copies, what registering a
// caller's buffer costs, how fast T host threads move pageable bytes into pinned memory, and how many streams'
// kernels really run at once.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/pcie_probe tools/diag/pcie_probe.hip -lpthread && gpurun_out/pcie_probe
#include <hip/hip_runtime.h>
#include <sched.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(e)                                                                        \
    do {                                                                                \
        hipError_t r_ = (e);                                                            \
        if (r_ != hipSuccess) {                                                         \
            printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__);       \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void spin_k(unsigned long long cycles, unsigned* sink)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}

static void par_copy(char* dst, const char* src, size_t bytes, int threads)
{
    std::vector<std::thread> pool;
    const size_t per = ((bytes + threads - 1) / threads + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; ++t) {
        const size_t a = (size_t)t * per;
        if (a >= bytes) break;
        const size_t len = bytes - a < per ? bytes - a : per;
        pool.emplace_back([=] { memcpy(dst + a, src + a, len); });
    }
    for (auto& th : pool) th.join();
}

int main()
{
    const size_t GiB = 1ull << 30;
    cpu_set_t set;
    CPU_ZERO(&set);
    sched_getaffinity(0, sizeof(set), &set);
    printf("hardware_concurrency %u, affinity %d cpus\n", std::thread::hardware_concurrency(), CPU_COUNT(&set));
    char *d_a = nullptr, *d_b = nullptr, *pin_a = nullptr, *pin_b = nullptr;
    CHECK(hipMalloc(&d_a, GiB));
    CHECK(hipMalloc(&d_b, GiB));
    double t = now();
    CHECK(hipHostMalloc(&pin_a, GiB, hipHostMallocDefault));
    printf("hipHostMalloc 1 GiB: %.1f ms\n", (now() - t) * 1e3);
    CHECK(hipHostMalloc(&pin_b, GiB, hipHostMallocDefault));
    memset(pin_a, 1, GiB);
    memset(pin_b, 2, GiB);
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        t = now();
        CHECK(hipMemcpyAsync(d_a, pin_a, GiB, hipMemcpyHostToDevice, s1));
        CHECK(hipStreamSynchronize(s1));
        const double h2d = now() - t;
        t = now();
        CHECK(hipMemcpyAsync(pin_b, d_b, GiB, hipMemcpyDeviceToHost, s2));
        CHECK(hipStreamSynchronize(s2));
        const double d2h = now() - t;
        t = now();
        CHECK(hipMemcpyAsync(d_a, pin_a, GiB, hipMemcpyHostToDevice, s1));
        CHECK(hipMemcpyAsync(pin_b, d_b, GiB, hipMemcpyDeviceToHost, s2));
        CHECK(hipStreamSynchronize(s1));
        CHECK(hipStreamSynchronize(s2));
        const double both = now() - t;
        printf("pinned 1 GiB: H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f ms (%.1f GB/s each way)\n", GiB / h2d / 1e9, GiB / d2h / 1e9, both * 1e3,
               GiB / both / 1e9);
    }
    // chunked pinned copies (32 MiB pieces back to back on one stream): the per-copy overhead
    for (size_t piece : {(size_t)4 << 20, (size_t)32 << 20, (size_t)128 << 20}) {
        t = now();
        for (size_t o = 0; o < GiB; o += piece) CHECK(hipMemcpyAsync(d_a + o, pin_a + o, piece, hipMemcpyHostToDevice, s1));
        CHECK(hipStreamSynchronize(s1));
        printf("pinned H2D in %zu MiB pieces: %.1f GB/s\n", piece >> 20, GiB / (now() - t) / 1e9);
    }
    // pageable memory through the runtime
    char* page_a = static_cast<char*>(malloc(GiB));
    char* page_b = static_cast<char*>(malloc(GiB));
    t = now();
    memset(page_a, 3, GiB);
    printf("first touch of 1 GiB (memset, one thread): %.1f ms\n", (now() - t) * 1e3);
    memset(page_b, 4, GiB);
    for (int rep = 0; rep < 2; ++rep) {
        t = now();
        CHECK(hipMemcpy(d_a, page_a, GiB, hipMemcpyHostToDevice));
        const double h2d = now() - t;
        t = now();
        CHECK(hipMemcpy(page_b, d_b, GiB, hipMemcpyDeviceToHost));
        const double d2h = now() - t;
        printf("pageable hipMemcpy 1 GiB: H2D %.1f GB/s, D2H %.1f GB/s\n", GiB / h2d / 1e9, GiB / d2h / 1e9);
    }
    // registering the caller's buffer
    for (int rep = 0; rep < 2; ++rep) {
        t = now();
        CHECK(hipHostRegister(page_a, GiB, hipHostRegisterDefault));
        const double reg = now() - t;
        t = now();
        CHECK(hipMemcpyAsync(d_a, page_a, GiB, hipMemcpyHostToDevice, s1));
        CHECK(hipStreamSynchronize(s1));
        const double h2d = now() - t;
        t = now();
        CHECK(hipHostUnregister(page_a));
        const double unreg = now() - t;
        printf("hipHostRegister 1 GiB: %.1f ms, H2D from it %.1f GB/s, unregister %.1f ms\n", reg * 1e3, GiB / h2d / 1e9, unreg * 1e3);
    }
    {
        const size_t piece = 64 << 20;
        t = now();
        for (size_t o = 0; o < GiB; o += piece) CHECK(hipHostRegister(page_a + o, piece, hipHostRegisterDefault));
        const double reg = now() - t;
        t = now();
        for (size_t o = 0; o < GiB; o += piece) CHECK(hipHostUnregister(page_a + o));
        printf("hipHostRegister in 64 MiB pieces: %.1f ms (%.2f ms each), unregister %.1f ms\n", reg * 1e3, reg * 1e3 / 16, (now() - t) * 1e3);
    }
    // host threads moving pageable bytes into pinned memory and back
    for (int threads : {1, 2, 4, 8, 12, 16, 24, 32, 64}) {
        par_copy(pin_a, page_a, GiB, threads);
        t = now();
        par_copy(pin_a, page_a, GiB, threads);
        const double in = now() - t;
        t = now();
        par_copy(page_b, pin_b, GiB, threads);
        const double out = now() - t;
        printf("memcpy with %2d threads: pageable->pinned %.1f GB/s, pinned->pageable %.1f GB/s\n", threads, GiB / in / 1e9, GiB / out / 1e9);
    }
    // threads copying while the link is busy both ways
    {
        const int threads = 16;
        t = now();
        CHECK(hipMemcpyAsync(d_a, pin_a, GiB, hipMemcpyHostToDevice, s1));
        CHECK(hipMemcpyAsync(pin_b, d_b, GiB, hipMemcpyDeviceToHost, s2));
        par_copy(page_b, page_a, GiB, threads);
        const double host = now() - t;
        CHECK(hipStreamSynchronize(s1));
        CHECK(hipStreamSynchronize(s2));
        printf("16-thread pageable memcpy under both DMA directions: %.1f GB/s, DMA pair done after %.1f ms\n", GiB / host / 1e9, (now() - t) * 1e3);
    }
    // how many streams' kernels run at once: K streams, one 2 ms spin kernel of one workgroup each
    {
        int clock_khz = 100000;
        (void)hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeWallClockRate, 0);
        const unsigned long long cycles = (unsigned long long)clock_khz * 2; // 2 ms
        std::vector<hipStream_t> ss(32);
        for (auto& s : ss) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipLaunchKernelGGL(spin_k, dim3(1), dim3(64), 0, ss[0], cycles, (unsigned*)nullptr);
        CHECK(hipDeviceSynchronize());
        for (int k : {1, 2, 4, 8, 16, 32}) {
            t = now();
            for (int i = 0; i < k; ++i) hipLaunchKernelGGL(spin_k, dim3(1), dim3(64), 0, ss[i], cycles, (unsigned*)nullptr);
            CHECK(hipDeviceSynchronize());
            printf("%2d streams x one 2 ms kernel: %.2f ms\n", k, (now() - t) * 1e3);
        }
    }
    return 0;
}
