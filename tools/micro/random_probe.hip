// tools/micro/random_probe.hip -- calibration microbenchmark (not part of the product): the rate of dependent-free random 8-byte
// reads from a table of a given size on gfx950, as the prefilter's slot probes issue them, and what FETCH_SIZE reports for them.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/random_probe.hip -o gpurun_out/random_probe
//   ./random_probe [tableMB ...]                 (rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./random_probe 512)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

template <int ILP>
__global__ __launch_bounds__(256) void probe(const uint64_t *table, uint64_t mask, uint32_t iters, uint64_t *out) {
    const uint64_t gid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = gid * 0x9E3779B97F4A7C15ull + 12345;
    uint64_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint64_t v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;      // independent of the loaded values: pure throughput
            v[k] = table[x & mask];
        }
#pragma unroll
        for (int k = 0; k < ILP; k++) acc += v[k];
    }
    if (acc == 0x1234567) out[gid] = acc;
}

int main(int argc, char **argv) {
    std::vector<int> sizes;
    for (int a = 1; a < argc; a++) sizes.push_back(atoi(argv[a]));
    if (sizes.empty()) sizes = {16, 64, 128, 256, 512, 1024, 2048};
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint64_t *out; hipMalloc(&out, 8ull << 20);
    for (int mb : sizes) {
        const uint64_t n = ((uint64_t) mb << 20) / 8;
        uint64_t *t; if (hipMalloc(&t, n * 8) != hipSuccess) { printf("alloc %d MB failed\n", mb); continue; }
        hipMemset(t, 1, n * 8);
        for (int wavesPerCu : {8, 16, 32}) {
            const int blocks = cus * wavesPerCu / 4;
            const uint32_t iters = 2000;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(256), 0, 0, t, n - 1, 50u, out);   // warm
            hipEventRecord(a);
            hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(256), 0, 0, t, n - 1, iters, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            const double loads = (double) blocks * 256 * iters * 8;
            printf("table %5d MB  waves/CU %2d  ILP 8: %7.2f G loads/s  (%.2f TB/s at 64 B per load)  %.2f ms\n", mb, wavesPerCu, loads / ms / 1e6, loads * 64 / ms / 1e9, ms);
        }
        hipFree(t);
    }
    return 0;
}
