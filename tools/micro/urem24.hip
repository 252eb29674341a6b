// tools/micro/urem24.hip -- does `y % n` come out right on gfx950 when the compiler knows both operands fit 24 bits (y = x >> 8, n = (r + 16383) >> 14)?
// The wide prefilter kernel lost every record whose subset number was 10 of 11 (profiles/r04_wide_kernel.txt); this checks the arithmetic alone.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/urem24.hip -o tools/micro/_build/urem24 && tools/micro/_build/urem24
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t recs, unsigned long long *bad, uint32_t *firstBad) {
    const uint32_t n = (recs + 16383u) / 16384u;                    // as in the kernel: known to be small
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < (1u << 24); x += gridDim.x * blockDim.x) {
        const uint32_t y = (x * 256u + 17u) >> 8;                   // provably < 2^24
        const uint32_t r = y % n;
        const uint32_t q = y / n;
        if (q * n + r != y || r >= n) { if (atomicAdd(bad, 1ull) == 0) { firstBad[0] = y; firstBad[1] = r; firstBad[2] = q; } }
    }
}
int main() {
    unsigned long long *dBad; uint32_t *dFirst;
    hipMalloc(&dBad, 8); hipMalloc(&dFirst, 16);
    int nBadDiv = 0;
    for (uint32_t n = 1; n <= 80; n++) {
        hipMemset(dBad, 0, 8);
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, n * 16384u - 5u, dBad, dFirst);
        unsigned long long b = 0; uint32_t f[3];
        hipMemcpy(&b, dBad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, dFirst, 12, hipMemcpyDeviceToHost);
        if (b) { nBadDiv++; printf("n = %u: %llu wrong of 2^24 (first: y = %u -> r = %u, q = %u; true r = %u)\n", n, b, f[0], f[1], f[2], f[0] % n); }
    }
    printf("divisors with wrong results: %d of 80\n", nBadDiv);
    return 0;
}
