// tools/micro/mk_experiments.hip -- experiments on the prefilter's index probes and on the score pass's layout (NOT part of the product:
// built into tools/micro/_build/libmk_experiments.so by tools/micro/build.sh, driven by tools/partition_probe_experiment.py and
// tools/interseq_experiment.py).  The second experiment (inter-sequence score pass) is at the end of the file.
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

// =================================================================================================================================
// Experiment 2 ("next round" item 3): an INTER-SEQUENCE score pass for queries of at most 64 rows.
//
// The product's score pass (mk_sw.hip: swp_kernel) runs a DP on a 16-lane group as an anti-diagonal wavefront, R = 2..4 rows per lane: per
// step 8 hand-over instructions next to 10 R of recurrence, 15 ramp steps per DP, rows padded to 32 / 48 / 64, and a wave holds at most 8
// pairs of ONE query.  tools/sw_schedule_model.py puts the alternatives in numbers (profiles/r04_sw_schedule_model.txt); this is the variant
// it found worth measuring:
//   * one lane = one packed PAIR of targets (low / high int16 halves), all R rows of the query in its registers (H, E), the column loop has
//     no lane-to-lane traffic at all: 10 instructions per pair of cells + ~ 12 per column;
//   * the lane's scores come from its query's profile in LDS, prof[residue][row] int16, one ds_read_b128 per 8 rows and target; a group of
//     G = 8 lanes shares one query (one profile, rows padded by 16 bytes so that the lanes' rows start in different banks), a wave has 8 groups;
//   * a group works through ONE query's pairs (falling target length): its lanes take the next two targets when theirs end; when the
//     query is used up and every lane of the group is idle the group takes the next query from the launch's work counter and rebuilds its
//     profile -- the other groups of the wave wait for that (~ 2 % by the model);
//   * target residues are fetched a block of 4 columns ahead (byte loads, clamped), a lane is retired at the end of the block in which its
//     longer target ended (the columns behind a target's end see the all-zero profile row and cannot raise the maximum).
// Scores only (the e-value gate needs nothing else).  Verified by the driver against mk_sw_pairs of the product on a sample; timed against
// sw_fwd_rows32 / 48 / 64 of the product's alignment stage alone on the same pairs.
namespace {

typedef short xpk16 __attribute__((ext_vector_type(2)));
typedef unsigned short xpku16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ xpk16 xpk_from(uint32_t v) { return __builtin_bit_cast(xpk16, v); }
__device__ __forceinline__ uint32_t xpk_bits(xpk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ xpk16 xpk_max(xpk16 a, xpk16 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ xpk16 xpk_splat(int v) { xpk16 r; r.x = (short) v; r.y = (short) v; return r; }
__device__ __forceinline__ xpk16 xpk_subs0(xpk16 a, xpk16 b) {          // a - b clamped at 0 for non-negative halves
    return __builtin_bit_cast(xpk16, __builtin_elementwise_sub_sat(__builtin_bit_cast(xpku16, a), __builtin_bit_cast(xpku16, b)));
}

struct InterArgs {
    const uint8_t *q_res; const int8_t *q_bias; const uint64_t *q_off;
    const uint8_t *t_res; const int8_t *mat;                              // mat[t * 21 + q], 21 x 21
    const uint64_t *j_tstart; const uint32_t *j_tlen; const uint32_t *j_q; // the ordered pairs: by query, inside a query by falling target length
    const uint32_t *unit_start; uint32_t n_units;                          // unit u = pairs [unit_start[u], unit_start[u + 1]) of one query
    int gap_open, gap_extend;
    int32_t *out;                                                          // score of every pair
    uint32_t *work_counter;
    unsigned long long *stats;                                             // [0] lane-blocks with a pair, [1] lane-blocks in all, [2] profile builds
};

template <int R>
__global__ __launch_bounds__(64) void interseq_kernel(InterArgs A) {
    constexpr int G = 8, NG = 64 / G;
    constexpr int ROWB = 2 * R + 16;                                       // bytes of a profile row (padded)
    constexpr int PROFB = 22 * ROWB;                                       // 21 residues + the all-zero row of "no column"
    static_assert(R % 8 == 0 && ROWB % 16 == 0, "a lane reads its scores 8 rows (16 bytes) at a time");
    extern __shared__ __attribute__((aligned(16))) char interSmem[];
    char *prof = interSmem;                                                // [group][residue][row]
    int8_t *sMat = reinterpret_cast<int8_t *>(interSmem + NG * PROFB);
    uint8_t *sQ = reinterpret_cast<uint8_t *>(sMat + 448);                 // [group][row]: the query's residues ...
    int8_t *sB = reinterpret_cast<int8_t *>(sQ + NG * R);                  // ... and composition bias, staged for the profile build
    const int lane = (int) threadIdx.x, g = lane / G, lig = lane % G;
    for (int k = lane; k < 441; k += 64) sMat[k] = A.mat[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const xpk16 go2 = xpk_splat(A.gap_open), ge2 = xpk_splat(A.gap_extend), zero2 = xpk_splat(0);
    char *myProf = prof + g * PROFB;
    // the group's query (the same in all its lanes): remaining pairs [next, end), and whether the work counter ran dry for it
    uint32_t next = 0, end = 0;
    bool exhausted = false;
    // the lane's pair
    bool active = false;
    xpk16 H[R], E[R];
    xpk16 best = zero2;
#pragma unroll
    for (int r = 0; r < R; r++) { H[r] = zero2; E[r] = zero2; }
    int col = 0, len = 0, tLenA = 0, tLenB = 0;
    uint64_t baseA = 0, baseB = 0;
    uint32_t jA = 0;
    bool haveB = false;
    uint32_t rowA[4] = {0, 0, 0, 0}, rowB[4] = {0, 0, 0, 0};               // byte offsets of the profile rows of the block's 4 columns
    unsigned long long busy = 0, blocks = 0, builds = 0;
    const auto rowOf = [&](uint64_t base, int tLen, int c) -> uint32_t {
        const uint32_t res = c < tLen ? (uint32_t) A.t_res[base + (uint64_t) min(c, max(tLen - 1, 0))] : 21u;
        return min(res, 21u) * (uint32_t) ROWB;
    };
    for (;;) {
        // ---- lanes without a pair take the next one of their group's query; groups without a query take the next one ----
        for (;;) {
            const unsigned long long am = __ballot(active);
            const uint32_t gAct = (uint32_t) (am >> (g * G)) & 0xFFu;
            const bool want = !active && !exhausted;
            const bool canPull = want && next >= end && gAct == 0u;        // (the same in every lane of the group)
            bool canTake = want && next < end;
            if (__ballot(canTake || canPull) == 0ull) break;
            if (canPull) {
                uint32_t u = 0;
                if (lig == 0) u = atomicAdd(A.work_counter, 1u);
                u = (uint32_t) __shfl((int) u, g * G, 64);
                if (u >= A.n_units) exhausted = true;
                else {
                    next = A.unit_start[u]; end = A.unit_start[u + 1];
                    const uint32_t qi = A.j_q[next];
                    const uint64_t q0 = A.q_off[qi];
                    const int qLen = (int) (A.q_off[qi + 1] - q0);
                    for (int row = lig; row < R; row += G) {
                        const bool real = row < qLen;
                        sQ[g * R + row] = real ? A.q_res[q0 + (uint64_t) row] : (uint8_t) 255;
                        sB[g * R + row] = real ? A.q_bias[q0 + (uint64_t) row] : (int8_t) 0;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int idx = lig; idx < 22 * R; idx += G) {
                        const int t = idx / R, row = idx - t * R;
                        const uint32_t qc = sQ[g * R + row];
                        const int v = (t < 21 && qc < 21u) ? (int) sMat[t * 21 + (int) qc] + (int) sB[g * R + row] : 0;
                        *reinterpret_cast<int16_t *>(myProf + t * ROWB + row * 2) = (int16_t) v;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lig == 0) builds++;
                    canTake = next < end;
                }
            }
            const unsigned long long tm = __ballot(canTake);
            const uint32_t gm = (uint32_t) (tm >> (g * G)) & 0xFFu;
            const uint32_t rank = (uint32_t) __popc(gm & ((1u << lig) - 1u)), cnt = (uint32_t) __popc(gm);
            if (canTake) {
                const uint32_t j = next + 2u * rank;
                if (j < end) {
                    jA = j; haveB = j + 1u < end;
                    baseA = A.j_tstart[j]; tLenA = (int) A.j_tlen[j];
                    baseB = haveB ? A.j_tstart[j + 1u] : baseA; tLenB = haveB ? (int) A.j_tlen[j + 1u] : 0;
                    len = max(tLenA, tLenB); col = 0; best = zero2;
#pragma unroll
                    for (int r = 0; r < R; r++) { H[r] = zero2; E[r] = zero2; }
#pragma unroll
                    for (int k = 0; k < 4; k++) { rowA[k] = rowOf(baseA, tLenA, k); rowB[k] = rowOf(baseB, tLenB, k); }
                    active = true;
                }
            }
            next = min(end, next + 2u * cnt);                              // (every lane of the group, whether it took a pair or not)
        }
        if (__ballot(active) == 0ull) break;                               // nothing left anywhere in this wave
        blocks++;
        if (active) {
            busy++;
            // the next block's residues first: their loads land while this block's columns are computed
            uint32_t nA[4], nB[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { nA[k] = rowOf(baseA, tLenA, col + 4 + k); nB[k] = rowOf(baseB, tLenB, col + 4 + k); }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const char *pa = myProf + rowA[k], *pb = myProf + rowB[k];
                xpk16 F = zero2, diag = zero2;                             // above row 0: nothing
#pragma unroll
                for (int r0 = 0; r0 < R; r0 += 8) {
                    const uint4 wa = *reinterpret_cast<const uint4 *>(pa + r0 * 2), wb = *reinterpret_cast<const uint4 *>(pb + r0 * 2);
                    const uint32_t a4[4] = {wa.x, wa.y, wa.z, wa.w}, b4[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int r = r0 + i;
                        const xpk16 sc = xpk_from(__builtin_amdgcn_perm(b4[i / 2], a4[i / 2], (i & 1) ? 0x07060302u : 0x05040100u));
                        const xpk16 d = diag + sc;
                        diag = H[r];
                        const xpk16 h = xpk_max(xpk_max(d, E[r]), F);      // E, F >= 0 keep H non-negative
                        best = xpk_max(best, h);
                        const xpk16 ho = xpk_subs0(h, go2);
                        E[r] = xpk_max(xpk_subs0(E[r], ge2), ho);
                        F = xpk_max(xpk_subs0(F, ge2), ho);
                        H[r] = h;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                         // a column's profile reads stay in their column: registers for 2 waves per SIMD
            }
            col += 4;
#pragma unroll
            for (int k = 0; k < 4; k++) { rowA[k] = nA[k]; rowB[k] = nB[k]; }
            if (col >= len) {
                A.out[jA] = (int32_t) (int16_t) (xpk_bits(best) & 0xFFFFu);
                if (haveB) A.out[jA + 1u] = (int32_t) (int16_t) (xpk_bits(best) >> 16);
                active = false;
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) { busy += __shfl_xor((long long) busy, d, 64); builds += __shfl_xor((long long) builds, d, 64); }
    if (lane == 0) { atomicAdd(&A.stats[0], busy); atomicAdd(&A.stats[1], blocks * 64ull); atomicAdd(&A.stats[2], builds); }
}

template <int R>
int launch_interseq(const InterArgs &A, unsigned grid, hipStream_t s) {
    constexpr int ROWB = 2 * R + 16;
    const size_t lds = (size_t) 8 * 22 * ROWB + 448 + 2 * 8 * R;
    hipLaunchKernelGGL((interseq_kernel<R>), dim3(grid), dim3(64), lds, s, A);
    return 0;
}

}  // namespace

// Host arrays in, scores out.  q_res / q_bias / q_off: the query batch (residue codes 0..20, int8 composition bias, offsets); t_res: the
// targets' residues (tBytes, unmasked); mat441: int8 scores [t * 21 + q]; the pairs (jTStart, jTLen, jQ) ordered by query and falling target
// length; unitStart[nUnits + 1] cuts them into runs of one query; rows = 32 / 48 / 64 (every query of the call has at most that many
// residues).  out[0] best ms of `reps` launches, out[1] lane-blocks with a pair / lane-blocks in all (how busy the lanes were),
// out[2] profile builds, out[3] LDS bytes per wave.
extern "C" int mkx_interseq_score(const uint8_t *qRes, const int8_t *qBias, const uint64_t *qOff, uint32_t nq, const uint8_t *tRes, uint64_t tBytes,
                                  const int8_t *mat441, const uint64_t *jTStart, const uint32_t *jTLen, const uint32_t *jQ, uint64_t nJobs,
                                  const uint32_t *unitStart, uint32_t nUnits, int rows, int gapOpen, int gapExtend, int wavesPerCu, int reps,
                                  int32_t *outScore, double *out) {
    if (rows != 32 && rows != 48 && rows != 64) { snprintf(g_err, sizeof(g_err), "rows: 32, 48 or 64"); return -1; }
    if (nJobs == 0 || nJobs >= 0x7FFFFFFFull || nUnits == 0) { snprintf(g_err, sizeof(g_err), "no pairs, or too many"); return -1; }
    const uint64_t qBytes = qOff[nq];
    uint8_t *dQ = nullptr, *dT = nullptr; int8_t *dB = nullptr, *dM = nullptr; uint64_t *dQOff = nullptr, *dJT = nullptr;
    uint32_t *dJL = nullptr, *dJQ = nullptr, *dU = nullptr, *dCounter = nullptr; int32_t *dOut = nullptr; unsigned long long *dStats = nullptr;
    XCHK(hipMalloc(&dQ, qBytes + 16)); XCHK(hipMalloc(&dB, qBytes + 16)); XCHK(hipMalloc(&dQOff, ((size_t) nq + 1) * 8));
    XCHK(hipMalloc(&dT, tBytes + 16)); XCHK(hipMalloc(&dM, 448));
    XCHK(hipMalloc(&dJT, nJobs * 8)); XCHK(hipMalloc(&dJL, nJobs * 4)); XCHK(hipMalloc(&dJQ, nJobs * 4)); XCHK(hipMalloc(&dU, ((size_t) nUnits + 1) * 4));
    XCHK(hipMalloc(&dOut, nJobs * 4)); XCHK(hipMalloc(&dCounter, 64)); XCHK(hipMalloc(&dStats, 64));
    XCHK(hipMemcpy(dQ, qRes, qBytes, hipMemcpyHostToDevice)); XCHK(hipMemcpy(dB, qBias, qBytes, hipMemcpyHostToDevice));
    XCHK(hipMemcpy(dQOff, qOff, ((size_t) nq + 1) * 8, hipMemcpyHostToDevice));
    XCHK(hipMemcpy(dT, tRes, tBytes, hipMemcpyHostToDevice)); XCHK(hipMemcpy(dM, mat441, 441, hipMemcpyHostToDevice));
    XCHK(hipMemcpy(dJT, jTStart, nJobs * 8, hipMemcpyHostToDevice)); XCHK(hipMemcpy(dJL, jTLen, nJobs * 4, hipMemcpyHostToDevice));
    XCHK(hipMemcpy(dJQ, jQ, nJobs * 4, hipMemcpyHostToDevice)); XCHK(hipMemcpy(dU, unitStart, ((size_t) nUnits + 1) * 4, hipMemcpyHostToDevice));
    XCHK(hipMemset(dOut, 0xFF, nJobs * 4));
    int dev = 0, cus = 256;
    XCHK(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    InterArgs A;
    A.q_res = dQ; A.q_bias = dB; A.q_off = dQOff; A.t_res = dT; A.mat = dM; A.j_tstart = dJT; A.j_tlen = dJL; A.j_q = dJQ;
    A.unit_start = dU; A.n_units = nUnits; A.gap_open = gapOpen; A.gap_extend = gapExtend; A.out = dOut; A.work_counter = dCounter; A.stats = dStats;
    const unsigned grid = (unsigned) std::min<uint64_t>((uint64_t) cus * (uint64_t) std::max(1, wavesPerCu), ((uint64_t) nUnits + 7) / 8);
    hipStream_t s = nullptr;
    float best = 1e30f;
    hipEvent_t e0, e1;
    XCHK(hipEventCreate(&e0)); XCHK(hipEventCreate(&e1));
    for (int r = 0; r < std::max(1, reps); r++) {
        XCHK(hipMemsetAsync(dCounter, 0, 64, s)); XCHK(hipMemsetAsync(dStats, 0, 64, s));
        XCHK(hipEventRecord(e0, s));
        if (rows == 32) launch_interseq<32>(A, grid, s); else if (rows == 48) launch_interseq<48>(A, grid, s); else launch_interseq<64>(A, grid, s);
        XCHK(hipEventRecord(e1, s));
        XCHK(hipStreamSynchronize(s));
        XCHK(hipGetLastError());
        float ms = 0;
        XCHK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    unsigned long long st[8] = {0};
    XCHK(hipMemcpy(st, dStats, 64, hipMemcpyDeviceToHost));
    XCHK(hipMemcpy(outScore, dOut, nJobs * 4, hipMemcpyDeviceToHost));
    out[0] = best; out[1] = st[1] ? (double) st[0] / (double) st[1] : 0.0; out[2] = (double) st[2]; out[3] = (double) (8 * 22 * (2 * rows + 16) + 448 + 16 * rows);
    for (void *p : {(void *) dQ, (void *) dB, (void *) dQOff, (void *) dT, (void *) dM, (void *) dJT, (void *) dJL, (void *) dJQ, (void *) dU, (void *) dOut, (void *) dCounter, (void *) dStats}) (void) hipFree(p);
    return 0;
}
