// tools/micro/hammer.hip -- interference microbenchmark (not part of the product): occupies ONE resource of the GPU for a while so
// that a bench run in another process shows how each stage reacts.   hammer hbm|valu|rate <seconds> <wavesPerCU>
//   rate: VALU issue rates of the instruction kinds the Smith-Waterman kernels use (one wave per SIMD .. 4 waves per SIMD)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>

__global__ __launch_bounds__(256) void k_hbm(const uint64_t *table, uint64_t mask, uint32_t iters, uint64_t *out) {
    const uint64_t gid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = gid * 0x9E3779B97F4A7C15ull + 12345, acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint64_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; v[k] = table[x & mask]; }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k];
    }
    if (acc == 0x1234567) out[gid] = acc;
}
typedef short pk16 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k_valu(uint32_t iters, uint32_t *out) {
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 7 + k;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (KIND == 0) a[k] = a[k] * 3u + 1u;                                             // v_mad_u32 / mul+add
                else if (KIND == 1) a[k] = max(a[k] + 3u, a[(k + 1) & 7]);                        // v_add + v_max (int32)
                else if (KIND == 2) { pk16 x = __builtin_bit_cast(pk16, a[k]), y = __builtin_bit_cast(pk16, a[(k + 1) & 7]); x = __builtin_elementwise_max(x + y, y); a[k] = __builtin_bit_cast(uint32_t, x); }   // v_pk_add_i16 + v_pk_max_i16
                else if (KIND == 3) a[k] = __builtin_amdgcn_perm(a[k], a[(k + 1) & 7], 0x07060302u);  // v_perm_b32
                else a[k] = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) a[k], 0x111, 0xf, 0xf, true) + 1u;   // DPP row_shr + add
            }
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += a[k];
    if (s == 0x12345) out[threadIdx.x] = s;
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const double secs = argc > 2 ? atof(argv[2]) : 10;
    const int wpc = argc > 3 ? atoi(argv[3]) : 8;
    int cus = 256;
    (void) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint64_t *out; (void) hipMalloc(&out, 64 << 20);
    if (!strcmp(argv[1], "rate")) {
        const char *names[5] = {"int32 mul+add", "int32 add+max", "packed i16 add+max", "v_perm_b32", "dpp row_shr + add"};
        const int opsPerInner[5] = {2, 2, 2, 1, 2};
        for (int kind = 0; kind < 5; kind++)
            for (int w : {4, 8, 16}) {
                const int blocks = cus * w / 4; const uint32_t iters = 20000;
                hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
                auto launch = [&](uint32_t n) {
                    switch (kind) { case 0: hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, n, (uint32_t *) out); break; case 1: hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, n, (uint32_t *) out); break;
                                    case 2: hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(256), 0, 0, n, (uint32_t *) out); break; case 3: hipLaunchKernelGGL(k_valu<3>, dim3(blocks), dim3(256), 0, 0, n, (uint32_t *) out); break;
                                    default: hipLaunchKernelGGL(k_valu<4>, dim3(blocks), dim3(256), 0, 0, n, (uint32_t *) out); } };
                launch(100);
                (void) hipEventRecord(a); launch(iters); (void) hipEventRecord(b); (void) hipEventSynchronize(b);
                float ms = 0; (void) hipEventElapsedTime(&ms, a, b);
                const double inst = (double) blocks * 4 * iters * 64 * opsPerInner[kind];            // wave-instructions
                printf("%-20s waves/CU %2d: %.3f T wave-instr/s = %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[kind], w, inst / ms / 1e9,
                       (double) cus * 4 * 2.4e9 / (inst / (ms * 1e-3)));
            }
        return 0;
    }
    uint64_t *t = nullptr; const uint64_t n = (2048ull << 20) / 8;
    if (!strcmp(argv[1], "hbm")) { (void) hipMalloc(&t, n * 8); (void) hipMemset(t, 1, n * 8); }
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        if (t) hipLaunchKernelGGL(k_hbm, dim3(cus * wpc / 4), dim3(256), 0, 0, t, n - 1, 400u, out);
        else hipLaunchKernelGGL(k_valu<2>, dim3(cus * wpc / 4), dim3(256), 0, 0, 40000u, (uint32_t *) out);
        (void) hipDeviceSynchronize();
        launches++;
    }
    printf("%s hammer: %d launches in %.1f s\n", argv[1], launches, secs);
    return 0;
}
