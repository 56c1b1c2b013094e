// lds_unaligned.hip -- does the LDS serve reads at addresses that are not multiples of their size?  (diagnostic)
//   hipcc --offload-arch=gfx950 -O2 -o build/lds_unaligned tools/diag/lds_unaligned.hip && build/lds_unaligned
// Fills 512 bytes of LDS with their own offsets (mod 251) and has lane l read 16 / 8 / 4 bytes at offset 2 * l + 1 * odd.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void probe(unsigned* out, int odd)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned char)(i % 251);
    __syncthreads();
    const unsigned at = (unsigned)(size_t)lds + 2 * threadIdx.x + (odd ? 1 : 0);
    unsigned a0, a1, a2, a3, b0, b1, c0;
    asm volatile("ds_read_b128 v[40:43], %[at]\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %[a0], v40\n\tv_mov_b32 %[a1], v41\n\tv_mov_b32 %[a2], v42\n\tv_mov_b32 %[a3], v43\n\t"
                 "ds_read_b64 v[40:41], %[at]\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %[b0], v40\n\tv_mov_b32 %[b1], v41\n\t"
                 "ds_read_b32 %[c0], %[at]\n\ts_waitcnt lgkmcnt(0)"
                 : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1), [c0] "=&v"(c0)
                 : [at] "v"(at)
                 : "v40", "v41", "v42", "v43", "memory");
    unsigned* o = out + 8 * threadIdx.x;
    o[0] = a0, o[1] = a1, o[2] = a2, o[3] = a3, o[4] = b0, o[5] = b1, o[6] = c0, o[7] = at & 1023u;
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 64 * 8 * 4);
    for (int odd = 0; odd < 2; ++odd) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, odd);
        unsigned h[64 * 8];
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int bad128 = 0, bad64 = 0, bad32 = 0;
        for (int l = 0; l < 64; ++l) {
            unsigned char want[16];
            const unsigned base = h[8 * l + 7];
            for (int k = 0; k < 16; ++k) want[k] = (unsigned char)((base + k) % 251);
            bad128 += memcmp(want, &h[8 * l], 16) != 0;
            bad64 += memcmp(want, &h[8 * l + 4], 8) != 0;
            bad32 += memcmp(want, &h[8 * l + 6], 4) != 0;
        }
        printf("%s addresses (2 x lane%s): lanes with wrong data: b128 %d, b64 %d, b32 %d of 64\n", odd ? "odd" : "even", odd ? " + 1" : "", bad128, bad64, bad32);
    }
    return 0;
}
