// tools/micro/mk_experiments.hip -- experiments on the prefilter's index probes (NOT part of the product: built into
// tools/micro/_build/libmk_experiments.so by tools/micro/build.sh, driven by tools/partition_probe_experiment.py).
//
// Question: the per-query kernels move 3.7 x their algorithmic bytes because every probe of the k-mer presence
// bitmap (8 MB), the slot table (512 MB) and the entries that misses L2 fetches a 128-byte line.  Would a RADIX-PARTITIONED probe -- write
// the similar k-mers as records into P cell-range partitions first, then let one XCD at a time probe a partition whose bitmap + slot
// slices fit its 4 MB L2 -- be faster?  The reference itself bins for cache residency (CacheFriendlyOperations.cpp:185-274).
//
// Measured here on the REAL probe stream of the headline workload (the similar k-mers of the first nq fragments against the real index,
// enumerated by the product's own enumerator), bitmap + slot only (the part of the probe that partitioning can make cache resident):
//   A   direct: the k-mers probed in enumeration order, as the per-query kernels do (presence bit, slot of the present ones)
//   B1  partition pass: the k-mer list -> P partitions by cell range (workgroup-level counting sort of a tile in LDS, one global
//       reservation per (tile, partition), runs written coalesced); records of 4 B (cell only) or 8 B (cell + origin, what a real
//       second pass needs to route its hits back)
//   B2  probe pass: partition p is probed by the workgroups of XCD p % 8 (block b runs on XCD b % 8), so its slices stay in that L2
// The enumeration itself (the same in both designs) is timed separately and not part of either figure.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../metaeuk_amd/csrc/mk_prefilter.hpp"
#include "../../metaeuk_amd/csrc/mk_enum.hpp"

namespace {

using mk::PrefilterDeviceView;
constexpr int WAVE = 64;

#define XCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", #x, hipGetErrorString(e_)); return -1; } } while (0)
char g_err[512];

// similar k-mers of every k-mer start: COUNT (list == nullptr) or FILL (cells at list[off[rel]...])
__global__ __launch_bounds__(256) void enumerate_kernel(PrefilterDeviceView V, uint64_t posBegin, uint64_t posEnd, uint32_t *count, const uint64_t *off, uint32_t *list) {
    __shared__ mk::enumk::EnumLds<2> sE[4];
    const int w = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    const uint64_t p = posBegin + (uint64_t) blockIdx.x * 4 + w;
    if (p >= posEnd) return;
    const int thr = (int) V.q_kmer_thr[p];
    const uint64_t rel = p - posBegin;
    if (thr < 0) { if (!list && lane == 0) count[rel] = 0; return; }
    uint32_t *dst = list ? list + off[rel] : nullptr;
    uint32_t done = 0;
    const uint32_t n = mk::enumk::enumerate_position<2>(V, V.q_res + p, thr, lane, sE[w], [&](const uint32_t (&kmer)[2], const bool (&has)[2]) -> bool {
        if (dst) {
#pragma unroll
            for (int u = 0; u < 2; u++) if (has[u]) dst[done + (uint32_t) (u * WAVE + lane)] = kmer[u];
        }
        done += 2 * WAVE;
        return true;
    });
    if (!list && lane == 0) count[rel] = n;
}

// A. direct probes in list order: ILP independent k-mers per lane, presence bit, slot of the present ones
template <int ILP>
__global__ __launch_bounds__(256) void direct_probe_kernel(const uint32_t *cells, uint64_t n, const uint32_t *bits, const uint64_t *slots, unsigned long long *out) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x * ILP;
    unsigned long long hits = 0, acc = 0;
    for (uint64_t base = (uint64_t) blockIdx.x * blockDim.x * ILP; base < n; base += stride) {
        uint32_t c[ILP];
        bool pres[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) { const uint64_t i = base + (uint64_t) k * blockDim.x + threadIdx.x; c[k] = i < n ? cells[i] : 0xFFFFFFFFu; }
#pragma unroll
        for (int k = 0; k < ILP; k++) pres[k] = c[k] != 0xFFFFFFFFu && ((bits[c[k] >> 5] >> (c[k] & 31u)) & 1u);
#pragma unroll
        for (int k = 0; k < ILP; k++) if (pres[k]) { acc += slots[c[k]]; hits++; }
    }
    for (int d = 32; d >= 1; d >>= 1) { hits += __shfl_xor((long long) hits, d, 64); acc += __shfl_xor((long long) acc, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], hits); atomicAdd(&out[1], acc); }
}

// B1. tile of TILE k-mers -> LDS counting sort by partition -> one reservation per (tile, partition), runs written coalesced
__global__ __launch_bounds__(256) void partition_count_kernel(const uint32_t *cells, uint64_t n, uint32_t cellsPerPart, uint32_t nPart, unsigned long long *count) {
    extern __shared__ uint32_t smem[];
    for (uint32_t k = threadIdx.x; k < nPart; k += 256) smem[k] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t) gridDim.x * 256;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += stride) atomicAdd(&smem[cells[i] / cellsPerPart], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nPart; k += 256) if (smem[k]) atomicAdd(&count[k], (unsigned long long) smem[k]);
}

// (cursor[p] starts at the partition's first record: the partitions are sized exactly by a counting pass -- the k-mer space is probed very
//  unevenly, the largest of 256 cell ranges holds 7 x the mean)
template <int TILE, int REC_WORDS>
__global__ __launch_bounds__(256) void partition_kernel(const uint32_t *cells, uint64_t n, uint32_t cellsPerPart, uint32_t nPart, unsigned long long *cursor /* [nPart]: next free record */,
                                                        uint32_t *records /* [n][REC_WORDS] */) {
    extern __shared__ uint32_t smem[];
    uint32_t *sCount = smem;                       // [nPart] counts, then starts
    unsigned long long *sBase = reinterpret_cast<unsigned long long *>(smem + 4 * nPart + (size_t) TILE * REC_WORDS);     // [nPart] global start of the tile's run (behind the cells)
    uint32_t *sFill = smem + 2 * nPart;            // [nPart] cursor inside the tile
    uint32_t *sCell = smem + 4 * nPart;            // [TILE] cells ordered by partition (4 nPart: keeps the 64-bit starts behind them aligned)
    uint32_t *sSrc = sCell + TILE;                 // [TILE] origin (REC_WORDS == 2)
    const uint64_t t0 = (uint64_t) blockIdx.x * TILE;
    const uint32_t m = (uint32_t) min((uint64_t) TILE, n - t0);
    for (uint32_t k = threadIdx.x; k < nPart; k += 256) { sCount[k] = 0; sFill[k] = 0; }
    __syncthreads();
    uint32_t mine[TILE / 256];
#pragma unroll
    for (int k = 0; k < TILE / 256; k++) {
        const uint32_t i = (uint32_t) k * 256 + threadIdx.x;
        mine[k] = i < m ? cells[t0 + i] : 0xFFFFFFFFu;
        if (i < m) atomicAdd(&sCount[mine[k] / cellsPerPart], 1u);
    }
    __syncthreads();
    // exclusive prefix over the partitions (one wave, nPart <= 1024) + the global reservations
    if (threadIdx.x < 64) {
        uint32_t carry = 0;
        for (uint32_t p0 = 0; p0 < nPart; p0 += 64) {
            const uint32_t p = p0 + threadIdx.x;
            const uint32_t c = p < nPart ? sCount[p] : 0u;
            uint32_t x = c;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t) __shfl_up((int) x, d, 64); if ((int) threadIdx.x >= d) x += y; }
            if (p < nPart) {
                sCount[p] = carry + x - c;
                sBase[p] = c ? atomicAdd(&cursor[p], (unsigned long long) c) : 0ull;
            }
            carry += (uint32_t) __shfl((int) x, 63, 64);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TILE / 256; k++) {
        const uint32_t i = (uint32_t) k * 256 + threadIdx.x;
        if (i < m) {
            const uint32_t p = mine[k] / cellsPerPart;
            const uint32_t at = sCount[p] + atomicAdd(&sFill[p], 1u);
            sCell[at] = mine[k];
            if (REC_WORDS == 2) sSrc[at] = (uint32_t) (t0 + i);
        }
    }
    __syncthreads();
    // runs out: consecutive threads write consecutive records of a run (the run of record x is found by binary search over the starts)
    for (uint32_t x = threadIdx.x; x < m; x += 256) {
        uint32_t lo = 0, hi = nPart;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sCount[mid] <= x) lo = mid; else hi = mid; }
        // (empty partitions share their start with the next one: take the last partition whose start is <= x)
        const uint64_t dst = (sBase[lo] + (x - sCount[lo])) * REC_WORDS;
        records[dst] = sCell[x];
        if (REC_WORDS == 2) records[dst + 1] = sSrc[x];
    }
}

// B2. partition p is probed by the persistent workgroups of XCD p % 8; G workgroups per XCD
template <int REC_WORDS, int ILP>
__global__ __launch_bounds__(256) void partition_probe_kernel(const uint32_t *records, const unsigned long long *partStart /* [nPart + 1] */, uint32_t nPart,
                                                              const uint32_t *bits, const uint64_t *slots, unsigned long long *out) {
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, G = gridDim.x >> 3;
    unsigned long long hits = 0, acc = 0;
    for (uint32_t p = xcd; p < nPart; p += 8) {
        const uint64_t n = partStart[p + 1] - partStart[p];
        const uint32_t *rec = records + partStart[p] * REC_WORDS;
        const uint64_t stride = (uint64_t) G * 256 * ILP;
        for (uint64_t base = (uint64_t) j * 256 * ILP; base < n; base += stride) {
            uint32_t c[ILP];
            bool pres[ILP];
#pragma unroll
            for (int k = 0; k < ILP; k++) { const uint64_t i = base + (uint64_t) k * 256 + threadIdx.x; c[k] = i < n ? rec[i * REC_WORDS] : 0xFFFFFFFFu; }
#pragma unroll
            for (int k = 0; k < ILP; k++) pres[k] = c[k] != 0xFFFFFFFFu && ((bits[c[k] >> 5] >> (c[k] & 31u)) & 1u);
#pragma unroll
            for (int k = 0; k < ILP; k++) if (pres[k]) { acc += slots[c[k]]; hits++; }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) { hits += __shfl_xor((long long) hits, d, 64); acc += __shfl_xor((long long) acc, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], hits); atomicAdd(&out[1], acc); }
}

template <class F>
int timed(hipStream_t s, int reps, float &bestMs, F &&launch) {
    hipEvent_t a, b;
    XCHK(hipEventCreate(&a)); XCHK(hipEventCreate(&b));
    bestMs = 1e30f;
    for (int r = 0; r < reps; r++) {
        XCHK(hipEventRecord(a, s));
        launch();
        XCHK(hipEventRecord(b, s));
        XCHK(hipStreamSynchronize(s));
        XCHK(hipGetLastError());
        float ms = 0;
        XCHK(hipEventElapsedTime(&ms, a, b));
        bestMs = std::min(bestMs, ms);
    }
    (void) hipEventDestroy(a); (void) hipEventDestroy(b);
    return 0;
}

}  // namespace

extern "C" const char *mkx_last_error() { return g_err; }

// view: mk::PrefilterDeviceView of a k = 6 database and a sequence batch (mk_debug_prefilter_view); qOff: the batch's host offsets.
// nParts[nCases] partitions to try (each <= 1024), recWords[nCases] 1 or 2.  out[0..7]: k-mers, index hits (present k-mers), ms of the count
// enumeration, ms of the fill enumeration, ms direct ILP 2, ms direct ILP 4, ms direct ILP 8, checksum agreement (1 = every variant found the
// same present k-mers and slot sum); out[8 + 4 c ...]: case c: ms partition pass, ms probe pass (best of 4 / 8 / 16 workgroups per CU),
// ms of the counting pass that sizes the partitions, largest partition / mean
extern "C" int mkx_partition_probe(const void *view, size_t viewBytes, const uint64_t *qOff, uint32_t nq, const int *nParts, const int *recWords, int nCases, double *out) {
    if (viewBytes != sizeof(PrefilterDeviceView)) { snprintf(g_err, sizeof(g_err), "view size mismatch"); return -1; }
    PrefilterDeviceView V = *static_cast<const PrefilterDeviceView *>(view);
    if (V.kmer_size != 6 || V.p_sorted) { snprintf(g_err, sizeof(g_err), "k = 6 sequence search only"); return -1; }
    hipStream_t s = nullptr;
    const uint64_t posBegin = qOff[0], posEnd = qOff[nq], nPos = posEnd - posBegin;
    if (nPos == 0 || nPos >= 0x7FFFFFFFull) { snprintf(g_err, sizeof(g_err), "empty or too large"); return -1; }
    uint32_t *dCount = nullptr; uint64_t *dOff = nullptr; uint32_t *dCells = nullptr; unsigned long long *dOut = nullptr;
    XCHK(hipMalloc(&dCount, (nPos + 1) * 4)); XCHK(hipMalloc(&dOff, (nPos + 2) * 8)); XCHK(hipMalloc(&dOut, 64));
    XCHK(hipMemset(dCount, 0, (nPos + 1) * 4));
    float msCount = 0, msFill = 0;
    const unsigned eb = (unsigned) ((nPos + 3) / 4);
    if (timed(s, 1, msCount, [&] { hipLaunchKernelGGL(enumerate_kernel, dim3(eb), dim3(256), 0, s, V, posBegin, posEnd, dCount, (const uint64_t *) nullptr, (uint32_t *) nullptr); })) return -1;
    {
        hipcub::TransformInputIterator<unsigned long long, hipcub::CastOp<unsigned long long>, uint32_t *> it(dCount, hipcub::CastOp<unsigned long long>());
        size_t tb = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, (unsigned long long *) dOff, (int) (nPos + 1), s);
        void *tmp = nullptr;
        XCHK(hipMalloc(&tmp, tb));
        XCHK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, it, (unsigned long long *) dOff, (int) (nPos + 1), s));
        XCHK(hipStreamSynchronize(s));
        (void) hipFree(tmp);
    }
    uint64_t nK = 0;
    XCHK(hipMemcpy(&nK, dOff + nPos, 8, hipMemcpyDeviceToHost));
    if (nK == 0 || nK >= (1ull << 32)) { snprintf(g_err, sizeof(g_err), "%llu similar k-mers: take fewer queries", (unsigned long long) nK); return -1; }
    XCHK(hipMalloc(&dCells, nK * 4));
    if (timed(s, 1, msFill, [&] { hipLaunchKernelGGL(enumerate_kernel, dim3(eb), dim3(256), 0, s, V, posBegin, posEnd, dCount, (const uint64_t *) dOff, dCells); })) return -1;
    out[0] = (double) nK; out[2] = msCount; out[3] = msFill;
    // ---- A: direct
    unsigned long long ref[2] = {0, 0};
    bool same = true;
    const unsigned cus = 256;
    float ms = 0;
    for (int v = 0; v < 3; v++) {
        XCHK(hipMemset(dOut, 0, 64));
        const unsigned grid = cus * 16;
        int rc = 0;
        if (v == 0) rc = timed(s, 3, ms, [&] { hipLaunchKernelGGL(direct_probe_kernel<2>, dim3(grid), dim3(256), 0, s, dCells, nK, V.kmer_bits, V.kmer_slot, dOut); });
        if (v == 1) rc = timed(s, 3, ms, [&] { hipLaunchKernelGGL(direct_probe_kernel<4>, dim3(grid), dim3(256), 0, s, dCells, nK, V.kmer_bits, V.kmer_slot, dOut); });
        if (v == 2) rc = timed(s, 3, ms, [&] { hipLaunchKernelGGL(direct_probe_kernel<8>, dim3(grid), dim3(256), 0, s, dCells, nK, V.kmer_bits, V.kmer_slot, dOut); });
        if (rc) return -1;
        unsigned long long h[2];
        XCHK(hipMemcpy(h, dOut, 16, hipMemcpyDeviceToHost));
        h[0] /= 3; h[1] /= 3;                       // three repetitions accumulated
        if (v == 0) { ref[0] = h[0]; ref[1] = h[1]; } else same = same && h[0] == ref[0] && h[1] == ref[1];
        out[4 + v] = ms;
    }
    out[1] = (double) ref[0];
    // ---- B: partitioned
    const uint64_t cellsTotal = 64000000ull;
    for (int c = 0; c < nCases; c++) {
        const uint32_t P = (uint32_t) nParts[c];
        const int RW = recWords[c];
        const uint32_t cellsPerPart = (uint32_t) ((cellsTotal + P - 1) / P);
        uint32_t *dRec = nullptr; unsigned long long *dCur = nullptr, *dStart = nullptr;
        XCHK(hipMalloc(&dRec, nK * RW * 4)); XCHK(hipMalloc(&dCur, (size_t) P * 8)); XCHK(hipMalloc(&dStart, (size_t) (P + 1) * 8));
        constexpr int TILE = 8192;
        const unsigned pb = (unsigned) ((nK + TILE - 1) / TILE);
        const size_t lds = (size_t) (4 * P + TILE * RW) * 4 + (size_t) P * 8;
        if (RW == 1) XCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&partition_kernel<TILE, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        else XCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&partition_kernel<TILE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        // counting pass (timed with the partition pass: a real implementation needs it too, or generous slack)
        std::vector<unsigned long long> hCnt(P), hStart(P + 1, 0);
        float msCountP = 0, msPart = 0, msProbe = 1e30f;
        int rc = timed(s, 3, msCountP, [&] {
            (void) hipMemsetAsync(dCur, 0, (size_t) P * 8, s);
            hipLaunchKernelGGL(partition_count_kernel, dim3(cus * 8), dim3(256), (size_t) P * 4, s, dCells, nK, cellsPerPart, P, dCur);
        });
        if (rc) return -1;
        XCHK(hipMemcpy(hCnt.data(), dCur, (size_t) P * 8, hipMemcpyDeviceToHost));
        unsigned long long mx = 0, sum = 0;
        for (uint32_t p = 0; p < P; p++) { hStart[p + 1] = hStart[p] + hCnt[p]; mx = std::max(mx, hCnt[p]); sum += hCnt[p]; }
        XCHK(hipMemcpy(dStart, hStart.data(), (size_t) (P + 1) * 8, hipMemcpyHostToDevice));
        rc = timed(s, 3, msPart, [&] {
            (void) hipMemcpyAsync(dCur, dStart, (size_t) P * 8, hipMemcpyDeviceToDevice, s);
            if (RW == 1) hipLaunchKernelGGL((partition_kernel<TILE, 1>), dim3(pb), dim3(256), lds, s, dCells, nK, cellsPerPart, P, dCur, dRec);
            else hipLaunchKernelGGL((partition_kernel<TILE, 2>), dim3(pb), dim3(256), lds, s, dCells, nK, cellsPerPart, P, dCur, dRec);
        });
        if (rc) return -1;
        for (unsigned perCu : {4u, 8u, 16u}) {
            XCHK(hipMemset(dOut, 0, 64));
            const unsigned grid = cus * perCu;                       // a multiple of 8: G workgroups per XCD
            float m2 = 0;
            if (RW == 1) rc = timed(s, 3, m2, [&] { hipLaunchKernelGGL((partition_probe_kernel<1, 4>), dim3(grid), dim3(256), 0, s, dRec, dStart, P, V.kmer_bits, V.kmer_slot, dOut); });
            else rc = timed(s, 3, m2, [&] { hipLaunchKernelGGL((partition_probe_kernel<2, 4>), dim3(grid), dim3(256), 0, s, dRec, dStart, P, V.kmer_bits, V.kmer_slot, dOut); });
            if (rc) return -1;
            unsigned long long h[2];
            XCHK(hipMemcpy(h, dOut, 16, hipMemcpyDeviceToHost));
            h[0] /= 3; h[1] /= 3;
            same = same && h[0] == ref[0] && h[1] == ref[1] && sum == nK;
            msProbe = std::min(msProbe, m2);
        }
        out[8 + 4 * c + 0] = msPart; out[8 + 4 * c + 1] = msProbe; out[8 + 4 * c + 2] = msCountP; out[8 + 4 * c + 3] = (double) mx / ((double) nK / P);
        (void) hipFree(dRec); (void) hipFree(dCur); (void) hipFree(dStart);
    }
    out[7] = same ? 1.0 : 0.0;
    (void) hipFree(dCount); (void) hipFree(dOff); (void) hipFree(dCells); (void) hipFree(dOut);
    return 0;
}
