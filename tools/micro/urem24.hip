// tools/micro/urem24.hip -- the root cause of "the lost subset" (profiles/r04_wide_kernel.txt): `y % n` for operands the compiler
// can PROVE to fit 24 bits (y = hash >> 8, n = (records + 16383) >> 14, as the wide prefilter kernel had them until commit cccf127).
//
// For such operands hipcc (ROCm 7.2 LLVM, AMDGPUCodeGenPrepare: expandDivRem24) replaces the integer division by
//     fq = trunc(float(y) * v_rcp_iflag_f32(float(n)));  fr = fma(-fq, float(n), float(y));  q = uint(fq) + (|fr| >= float(n));  r = (y - q n) & 0xFFFFFF
// which corrects a quotient that came out one too SMALL, never one that came out one too LARGE.  float(y) * rcp(n) has up to 2^-24 * q of rounding
// error; once q >= 2^20 or so that is more than the 1 / n by which the true quotient of a numerator with remainder n - 1 stays below the next
// integer, the product rounds UP to that integer, trunc() keeps it, fr = -1, and the "remainder" is 0xFFFFFF.  With n = 11 that happens for 476 625
// of the 2^24 numerators -- all of them >= 11 534 346 and all with true remainder 10 -- whenever 1.0f / n rounds up (3, 7, 11, 12, 13, 44, 46, 57 ...;
// not 10, whose reciprocal rounds down).  In the kernel `hash % nSets == set` therefore held for NO set when the class was cut into 11 (44, 46, 57)
// subsets and the record's hash was large with remainder nSets - 1: "every lost target had subset number 10 of 11".
//
// Round 4's version of this file compared q * n + r with y on the device; both came from the same wrong q, the identity holds by construction, and
// the compiler folded the whole kernel to s_endpgm -- "0 wrong" meant nothing.  This version stores the remainders and compares on the HOST, next to a
// host model of the instruction sequence (correctly rounded reciprocal), so the output says whether the device does what the model predicts.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/urem24.hip -o tools/micro/_build/urem24 && tools/micro/_build/urem24
// Exit status 0: every remainder right (a fixed compiler); 3: wrong remainders, all explained by the mechanism (true remainder n - 1, y >= 2^22);
// 1: anything else.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void rem24_kernel(uint32_t recs, const uint32_t *in, uint32_t *out) {
    const uint32_t n = (recs + 16383u) / 16384u;                    // as in the kernel: at most 18 bits
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < (1u << 24); x += gridDim.x * blockDim.x) {
        const uint32_t y = in[x] >> 8;                              // a "hash" from memory, = x: provably < 2^24, not foldable
        out[x] = y % n;
    }
}

// the emitted sequence with a correctly rounded reciprocal
static uint32_t model(uint32_t y, uint32_t n) {
    const float fa = (float) y, fb = (float) n, rc = 1.0f / fb;
    const float fq = truncf(fa * rc);
    const float fr = fmaf(-fq, fb, fa);
    const uint32_t q = (uint32_t) fq + (fabsf(fr) >= fb ? 1u : 0u);
    return (y - q * n) & 0xFFFFFFu;
}

int main() {
    uint32_t *dOut = nullptr, *dIn = nullptr;
    if (hipMalloc(&dOut, sizeof(uint32_t) << 24) != hipSuccess || hipMalloc(&dIn, sizeof(uint32_t) << 24) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    std::vector<uint32_t> h(1u << 24);
    for (uint32_t x = 0; x < (1u << 24); x++) h[x] = (x << 8) | ((x * 2654435761u) >> 24);
    if (hipMemcpy(dIn, h.data(), sizeof(uint32_t) << 24, hipMemcpyHostToDevice) != hipSuccess) { printf("hipMemcpy failed\n"); return 1; }
    const uint32_t divisors[] = {2, 3, 7, 10, 11, 12, 13, 16, 44, 46, 57, 64, 80};
    int wrongDivisors = 0, unexplained = 0;
    for (uint32_t n : divisors) {
        hipLaunchKernelGGL(rem24_kernel, dim3(1024), dim3(256), 0, 0, n * 16384u - 5u, (const uint32_t *) dIn, dOut);
        if (hipMemcpy(h.data(), dOut, sizeof(uint32_t) << 24, hipMemcpyDeviceToHost) != hipSuccess) { printf("hipMemcpy failed\n"); return 1; }
        uint64_t wrong = 0, asModel = 0, modelWrong = 0, notTop = 0;
        uint32_t first = 0;
        for (uint32_t y = 0; y < (1u << 24); y++) {
            const uint32_t m = model(y, n);
            if (m != y % n) modelWrong++;
            if (h[y] != y % n) {
                if (!wrong) first = y;
                wrong++;
                if (h[y] == m) asModel++;
                if (y % n != n - 1 || y < (1u << 22)) notTop++;
            }
        }
        printf("n = %2u: device wrong for %8llu of 2^24 numerators (first y = %8u, true remainder %2u, device says 0x%X); the host model of the sequence: %8llu wrong; "
               "device == model on %llu of the wrong ones; outside 'remainder n - 1, y >= 2^22': %llu\n",
               n, (unsigned long long) wrong, first, wrong ? first % n : 0u, wrong ? h[first] : 0u, (unsigned long long) modelWrong, (unsigned long long) asModel, (unsigned long long) notTop);
        if (wrong) wrongDivisors++;
        if (notTop) unexplained++;
    }
    printf("divisors with wrong remainders: %d of %zu; divisors with wrong remainders the mechanism does not explain: %d\n", wrongDivisors, sizeof(divisors) / sizeof(divisors[0]), unexplained);
    return unexplained ? 1 : (wrongDivisors ? 3 : 0);
}
