// metaeuk_amd/csrc/mk_index.hip -- tantan masking and the k-mer index of the target database, built in HBM (see mk_index.hpp).
// Replaces, for the whole database at once:
//   Masker::maskSequence -> tantan::maskSequences            M/src/commons/Masker.cpp:15-32, M/lib/tantan/tantan.cpp:320-449,475-527
//   IndexBuilder::fillDatabase                               M/src/prefiltering/IndexBuilder.cpp:55-239
//   IndexTable::addKmerCount / addSequence / sortDBSeqLists  M/src/prefiltering/IndexTable.h:133-173,348-401,182-189
//
// tantan_kernel      one LANE per sequence (the recurrence over positions is sequential; the 50 repeat offsets live in 100 VGPRs, the
//                    last 50 residues in 13 more), sequences dealt to the waves by falling length.  Forward-backward in double with the
//                    reference build's partial-sum association (mk_params.simd_lanes_double); the per-position scratch of a wave is
//                    interleaved (position-major, lane-minor) so that its traffic is coalesced.  -ffp-contract=off; f64 add / mul /
//                    div of gfx950 are correctly rounded: the masked residues are byte-identical to the host's.
// index_count_kernel one workgroup per sequence: table cell of every k-mer start (spaced seed, X and self-score filter), the
//                    reference's "first position of a k-mer in the sequence" by comparing every start with the earlier ones through
//                    an LDS tile (quadratic in the sequence length, ~40 ms for 4.4e9 residues), a bit per kept start, one atomic per
//                    kept start on the cell's counter.
// scan_*             list lengths -> list starts (three sweeps, 4096 cells per workgroup), slots + presence bits.
// index_fill_kernel  kept starts -> entries; the place inside the list comes from an atomic on the cell's counter, so a list is
//                    complete but in arrival order ...
// index_finalize_*   ... and is sorted by target here (a (cell, target) pair occurs once, so that IS the reference's (target, position)
//                    order): a wave takes 64 neighbouring cells, whose lists are neighbours in memory as well, stages them in LDS and
//                    every lane sorts its own short list; lists of 65 .. 4096 entries get a workgroup and an LDS bitonic network,
//                    longer ones (a k-mer that occurs in > 4096 targets) a bitonic network in HBM, one launch per stride.
//                    Single-entry lists move into their slot.
#include "mk_index.hpp"
#include "mk_kernels.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>

namespace mk {

namespace {

constexpr int WAVE = 64;
constexpr int TW = 50;                  // tantan's maxRepeatOffset
constexpr int TSTEP = 16;               // ... its rescaling interval
constexpr uint64_t M40 = (1ull << 40) - 1ull;
constexpr uint32_t INVALID_CELL = 0xFFFFFFFFu;

#define ICHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return MK_ERR_DEVICE; } } while (0)

template <typename T>
struct Tmp {                             // scoped device allocation
    T *p = nullptr;
    ~Tmp() { if (p) (void) hipFree(p); }
    hipError_t alloc(size_t n) { return n ? hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)) : hipSuccess; }
    T *take() { T *q = p; p = nullptr; return q; }
};

// =====================================================================================================
//  tantan
// =====================================================================================================
struct TantanArgs {
    const uint8_t *res; uint8_t *masked; const uint64_t *off;
    const uint32_t *order;               // sequences by falling length
    uint32_t n_seq; uint32_t wave0;      // first wave (of the whole run) of this launch
    uint32_t n_waves;                    // waves of this launch
    const uint64_t *post_off;            // [wave - wave0]: the wave's block of 64 * Lmax floats ...
    const uint64_t *scale_off;           // ... and of 64 * (Lmax / 16 + 1) doubles
    float *post; double *scale;
    const double *ratio;                 // [21][21] likelihood ratios
    double enter[TW];
    double min_mask_prob;
    unsigned long long *masked_count;
};

__device__ __forceinline__ uint32_t win_byte(const uint32_t (&win)[13], int i) { return (win[i >> 2] >> ((i & 3) * 8)) & 0xFFu; }

template <int LANES>
__device__ __forceinline__ double hsum_parts(const double (&p)[4]) {
    if (LANES == 4) return (p[0] + p[2]) + (p[1] + p[3]);
    if (LANES == 2) return p[0] + p[1];
    return p[0];
}
// offset i belongs to a complete group of LANES offsets below `reach`
template <int LANES>
__device__ __forceinline__ bool grouped(int i, int reach) { return (i / LANES) * LANES + LANES <= reach; }

template <int LANES>
__global__ __launch_bounds__(256) void tantan_kernel(TantanArgs A) {
    __shared__ double sRatio[21 * 21];
    for (int k = threadIdx.x; k < 21 * 21; k += blockDim.x) sRatio[k] = A.ratio[k];
    __syncthreads();
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t wl = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;       // wave of this launch
    if (wl >= A.n_waves) return;
    const uint64_t slot = (uint64_t) (A.wave0 + wl) * WAVE + (uint64_t) lane;
    const bool have = slot < (uint64_t) A.n_seq;
    const uint32_t s = have ? A.order[slot] : 0u;
    const uint64_t begin = have ? A.off[s] : 0ull;
    const int L = have ? (int) (A.off[s + 1] - begin) : 0;
    const int Lmax = __builtin_amdgcn_readfirstlane(L);                              // lane 0 holds the longest sequence of the wave
    const uint8_t *seq = A.res + begin;
    float *post = A.post + A.post_off[wl] + lane;
    double *scale = A.scale + A.scale_off[wl] + lane;
    const double pRepeat = 0.005, pEnd = 0.05;
    const double bgStay = 1 - pRepeat, fgStay = 1 - pEnd;
    (void) pRepeat;

    double fg[TW];
    uint32_t win[13];
#pragma unroll
    for (int k = 0; k < TW; k++) fg[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 13; k++) win[k] = 0u;
    double bgp = 1.0;
    // ---- forward
    for (int pos = 0; pos < Lmax; pos++) {
        if (pos < L) {
            const uint32_t cur = seq[pos];
            const double *rr = sRatio + cur * 21u;
            const int reach = pos < TW ? pos : TW;
            const double b = bgp;
            double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < TW; i++) if (grouped<LANES>(i, reach)) part[i % LANES] = part[i % LANES] + fg[i];
            double fromFg = hsum_parts<LANES>(part);
#pragma unroll
            for (int i = 0; i < TW; i++) if (!grouped<LANES>(i, reach) && i < reach) fromFg += fg[i];
#pragma unroll
            for (int i = 0; i < TW; i++) if (i < reach) fg[i] = (b * A.enter[i] + fg[i] * fgStay) * rr[win_byte(win, i)];
            bgp = b * bgStay + fromFg * pEnd;
            if (pos % TSTEP == TSTEP - 1) {
                const double sc = 1 / bgp;
                scale[(size_t) (pos / TSTEP) * WAVE] = sc;
                bgp *= sc;
#pragma unroll
                for (int k = 0; k < TW; k++) fg[k] *= sc;
            }
            post[(size_t) pos * WAVE] = (float) bgp;
#pragma unroll
            for (int k = 12; k >= 1; k--) win[k] = (win[k] << 8) | (win[k - 1] >> 24);
            win[0] = (win[0] << 8) | cur;
        }
    }
    double tail = 0.0;
#pragma unroll
    for (int k = 0; k < TW; k++) tail += fg[k];
    const double total = bgp * bgStay + tail * pEnd;
    // ---- backward (win: byte i = seq[L - 1 - i]; before position pos is used it becomes byte i = seq[pos - 1 - i])
    bgp = bgStay;
#pragma unroll
    for (int k = 0; k < TW; k++) fg[k] = pEnd;
    uint32_t nMasked = 0;
    uint8_t *outSeq = A.masked + begin;
    for (int pos = Lmax - 1; pos >= 0; pos--) {
        if (pos < L) {
#pragma unroll
            for (int k = 0; k < 12; k++) win[k] = (win[k] >> 8) | (win[k + 1] << 24);
            const uint32_t older = pos >= TW ? (uint32_t) seq[pos - TW] : 0u;
            win[12] = ((win[12] >> 8) & ~0xFF00u) | (older << 8);
            const double nonRepeat = (double) post[(size_t) pos * WAVE] * bgp / total;
            const float pr = 1 - (float) nonRepeat;
            if (pos % TSTEP == TSTEP - 1) {
                const double sc = scale[(size_t) (pos / TSTEP) * WAVE];
                bgp *= sc;
#pragma unroll
                for (int k = 0; k < TW; k++) fg[k] *= sc;
            }
            const uint32_t cur = seq[pos];
            const double *rr = sRatio + cur * 21u;
            const int reach = pos < TW ? pos : TW;
            const double toBg = pEnd * bgp;
            double part[4] = {0.0, 0.0, 0.0, 0.0};
            double toFg = 0.0;
#pragma unroll
            for (int i = 0; i < TW; i++)
                if (grouped<LANES>(i, reach)) {
                    const double f = fg[i] * rr[win_byte(win, i)];
                    part[i % LANES] = part[i % LANES] + A.enter[i] * f;
                    fg[i] = toBg + fgStay * f;
                }
            toFg = hsum_parts<LANES>(part);
#pragma unroll
            for (int i = 0; i < TW; i++)
                if (!grouped<LANES>(i, reach) && i < reach) {
                    const double f = fg[i] * rr[win_byte(win, i)];
                    toFg += A.enter[i] * f;
                    fg[i] = toBg + fgStay * f;
                }
            bgp = bgStay * bgp + toFg;
            const bool m = (double) pr >= A.min_mask_prob;
            outSeq[pos] = m ? (uint8_t) XCODE : (uint8_t) cur;
            nMasked += m ? 1u : 0u;
        }
    }
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) nMasked += (uint32_t) __shfl_xor((int) nMasked, d, WAVE);
    if (lane == 0 && nMasked) atomicAdd(A.masked_count, (unsigned long long) nMasked);
}

// =====================================================================================================
//  k-mer cells
// =====================================================================================================
struct CellGeom {
    int8_t self[21];                    // self score of every residue in the seed matrix
    uint8_t addr[20];                   // KMER_ADDR_LETTER: residue -> (quad << 2 | position in the quad)
    int kmer_thr;
    int tiled;                          // k = 6: the tiled address order
};

template <int K>
__device__ __forceinline__ uint32_t cell_of(const uint8_t *r, const CellGeom &G) {
    constexpr int SP6[6] = {0, 1, 3, 5, 8, 9};
    constexpr int SP7[7] = {0, 1, 3, 5, 6, 9, 10};
    uint32_t let[K];
    int score = 0;
    bool hasX = false;
#pragma unroll
    for (int p = 0; p < K; p++) {
        const uint32_t c = r[K == 7 ? SP7[p] : SP6[p]];
        hasX |= c >= 20u;
        score += (int) G.self[c < 21u ? c : 20u];
        let[p] = c < 20u ? c : 0u;
    }
    if (hasX || (G.kmer_thr > 0 && score < G.kmer_thr)) return INVALID_CELL;
    if (K == 6 && G.tiled) {
        uint32_t d[6];
#pragma unroll
        for (int p = 0; p < 6; p++) d[p] = G.addr[let[p]];
        const uint32_t a = (((d[0] >> 2) + 5u * (d[1] >> 2) + 25u * (d[2] >> 2)) << 6) | ((d[0] & 3u) + 4u * (d[1] & 3u) + 16u * (d[2] & 3u));
        const uint32_t b = (((d[3] >> 2) + 5u * (d[4] >> 2) + 25u * (d[5] >> 2)) << 6) | ((d[3] & 3u) + 4u * (d[4] & 3u) + 16u * (d[5] & 3u));
        return 4096u * ((a >> 6) + 125u * (b >> 6)) + (a & 63u) + 64u * (b & 63u);
    }
    uint32_t idx = 0, pw = 1;
#pragma unroll
    for (int p = 0; p < K; p++) { idx += let[p] * pw; pw *= 20u; }
    return idx;
}

struct CountArgs {
    const uint8_t *masked; const uint64_t *off; uint32_t seq0, n_seq;
    CellGeom G;
    uint32_t *count;                    // [cells]
    uint32_t *first_bits;               // bit p: residue position p starts a kept k-mer
    uint64_t *entries; const uint64_t *slots; uint64_t entry_shift;      // fill pass
};

constexpr int CT = 256;                 // threads per sequence
constexpr int TJ = 1024;                // cells per comparison tile

constexpr int LONG_STARTS = 4096;       // sequences with more k-mer starts are cut into 256-start work items of their own (index_count_long_kernel)

// the kept starts among i0 .. i0 + 255 of sequence s (whole workgroup): table cell, first-position test against the other starts, counter + bit
template <int K>
__device__ __forceinline__ void count_round(const CountArgs &A, uint32_t s, uint64_t begin, const uint8_t *seq, int nStart, int i0, uint32_t *sCell) {
    const int tid = threadIdx.x, w = tid / WAVE, lane = tid & (WAVE - 1);
    const int i = i0 + tid;
    const uint32_t mine = i < nStart ? cell_of<K>(seq + i, A.G) : INVALID_CELL;
    bool dup = false;
    if (nStart <= 65536) {
        for (int j0 = 0; j0 < i0 + CT && j0 < nStart; j0 += TJ) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (a barrier at a loop head gets its wait spelled out: profiles/r06_barrier_at_loop_head.txt)
            __syncthreads();
            for (int t = tid; t < TJ; t += CT) { const int j = j0 + t; sCell[t] = j < nStart ? cell_of<K>(seq + j, A.G) : INVALID_CELL; }
            __syncthreads();
            const int lim = min(TJ, i - j0);                                    // tile cells that lie before start i
            const int wlim = min(TJ, i0 + w * WAVE + WAVE - 1 - j0);            // ... before the wave's last start
            for (int t = 0; t < wlim; t += 4) {
                const uint4 c = *reinterpret_cast<const uint4 *>(&sCell[t]);
                dup |= (t < lim && c.x == mine) | (t + 1 < lim && c.y == mine) | (t + 2 < lim && c.z == mine) | (t + 3 < lim && c.w == mine);
            }
        }
    } else {
        // index positions are 16 bits wide (IndexEntryLocal::position_j): the reference sorts a sequence's k-mers by (k-mer, position as
        // stored) and keeps the first, so beyond 65536 residues "first" means the smallest WRAPPED position -- every start is compared
        const uint32_t key = (((uint32_t) i & 0xFFFFu) << 16) | ((uint32_t) i >> 16);
        for (int j0 = 0; j0 < nStart; j0 += TJ) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (a barrier at a loop head gets its wait spelled out: profiles/r06_barrier_at_loop_head.txt)
            __syncthreads();
            for (int t = tid; t < TJ; t += CT) { const int j = j0 + t; sCell[t] = j < nStart ? cell_of<K>(seq + j, A.G) : INVALID_CELL; }
            __syncthreads();
            for (int t = 0; t < TJ; t += 4) {
                const uint4 c = *reinterpret_cast<const uint4 *>(&sCell[t]);
                const uint32_t j = (uint32_t) (j0 + t);
                dup |= (c.x == mine && (((j & 0xFFFFu) << 16) | (j >> 16)) < key) | (c.y == mine && ((((j + 1u) & 0xFFFFu) << 16) | ((j + 1u) >> 16)) < key) |
                       (c.z == mine && ((((j + 2u) & 0xFFFFu) << 16) | ((j + 2u) >> 16)) < key) | (c.w == mine && ((((j + 3u) & 0xFFFFu) << 16) | ((j + 3u) >> 16)) < key);
            }
        }
    }
    const bool first = mine != INVALID_CELL && !dup;
    if (first) atomicAdd(&A.count[mine], 1u);
    const unsigned long long m = __ballot(first);
    if (m) {                                                                // (wave-uniform)
        const uint64_t p0 = begin + (uint64_t) (i0 + w * WAVE);
        const uint64_t word = p0 >> 5;
        const uint32_t sh = (uint32_t) (p0 & 31u);
        const uint32_t w0 = (uint32_t) (m << sh);
        const uint32_t w1 = sh ? (uint32_t) (m >> (32u - sh)) : (uint32_t) (m >> 32);
        const uint32_t w2 = sh ? (uint32_t) (m >> (64u - sh)) : 0u;
        if (lane == 0 && w0) atomicOr(&A.first_bits[word], w0);
        if (lane == 1 && w1) atomicOr(&A.first_bits[word + 1], w1);
        if (lane == 2 && w2) atomicOr(&A.first_bits[word + 2], w2);
    }
    (void) s;
}

template <int K>
__global__ __launch_bounds__(CT) void index_count_kernel(CountArgs A) {
    constexpr int SPAN = K == 7 ? 11 : 10;
    __shared__ __attribute__((aligned(16))) uint32_t sCell[TJ];
    const uint32_t s = A.seq0 + blockIdx.x;
    if (s >= A.n_seq) return;
    const uint64_t begin = A.off[s];
    const int L = (int) (A.off[s + 1] - begin);
    const int nStart = L >= SPAN ? L - SPAN + 1 : 0;
    if (nStart > LONG_STARTS) return;                                       // index_count_long_kernel
    for (int i0 = 0; i0 < nStart; i0 += CT) count_round<K>(A, s, begin, A.masked + begin, nStart, i0, sCell);
}

// long sequences: one workgroup per 256 starts (a titin-sized target alone would keep one workgroup busy for seconds)
template <int K>
__global__ __launch_bounds__(CT) void index_count_long_kernel(CountArgs A, const uint2 *items /* (sequence, first start) */, uint32_t nItems) {
    constexpr int SPAN = K == 7 ? 11 : 10;
    __shared__ __attribute__((aligned(16))) uint32_t sCell[TJ];
    if (blockIdx.x >= nItems) return;
    const uint32_t s = items[blockIdx.x].x;
    const uint64_t begin = A.off[s];
    const int L = (int) (A.off[s + 1] - begin);
    count_round<K>(A, s, begin, A.masked + begin, L - SPAN + 1, (int) items[blockIdx.x].y, sCell);
}

template <int K>
__global__ __launch_bounds__(CT) void index_fill_kernel(CountArgs A) {
    constexpr int SPAN = K == 7 ? 11 : 10;
    const uint32_t s = A.seq0 + blockIdx.x;
    if (s >= A.n_seq) return;
    const uint64_t begin = A.off[s];
    const int L = (int) (A.off[s + 1] - begin);
    const int nStart = L >= SPAN ? L - SPAN + 1 : 0;
    const uint8_t *seq = A.masked + begin;
    for (int i = threadIdx.x; i < nStart; i += CT) {
        const uint64_t p = begin + (uint64_t) i;
        if (!((A.first_bits[p >> 5] >> (p & 31u)) & 1u)) continue;
        const uint32_t cell = cell_of<K>(seq + i, A.G);
        const uint32_t r = atomicSub(&A.count[cell], 1u) - 1u;                  // a place of its own inside the list
        const uint64_t start = (A.slots[cell] & M40) - A.entry_shift;
        A.entries[start + r] = (uint64_t) s | ((uint64_t) ((uint32_t) i & 0xFFFFu) << 32);
    }
}

// =====================================================================================================
//  list lengths -> slots
// =====================================================================================================
constexpr int SCAN_T = 256, SCAN_PER = 16, SCAN_BLOCK = SCAN_T * SCAN_PER;      // 4096 cells per workgroup

__device__ __forceinline__ uint64_t block_excl_scan64(uint64_t v, uint64_t *sWave /* [SCAN_T / 64] */, uint64_t &total) {
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    uint64_t x = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t lo = (uint32_t) __shfl_up((int) (uint32_t) x, d, WAVE), hi = (uint32_t) __shfl_up((int) (uint32_t) (x >> 32), d, WAVE);
        if (lane >= d) x += ((uint64_t) hi << 32) | lo;
    }
    if (lane == WAVE - 1) sWave[w] = x;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (int k = 0; k < (int) (blockDim.x / WAVE); k++) { if (k < w) before += sWave[k]; all += sWave[k]; }
    __syncthreads();
    total = all;
    return before + x - v;
}

__global__ __launch_bounds__(SCAN_T) void scan_sums_kernel(const uint32_t *count, uint64_t cells, uint64_t *blockSum, uint32_t *maxList) {
    __shared__ uint64_t sWave[SCAN_T / WAVE];
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_BLOCK + (uint64_t) threadIdx.x * SCAN_PER;
    uint64_t sum = 0;
    uint32_t mx = 0;
    for (int k = 0; k < SCAN_PER; k++) if (base + k < cells) { const uint32_t c = count[base + k]; sum += c; mx = max(mx, c); }
    uint64_t total;
    (void) block_excl_scan64(sum, sWave, total);
    if (threadIdx.x == 0) blockSum[blockIdx.x] = total;
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) mx = max(mx, (uint32_t) __shfl_xor((int) mx, d, WAVE));
    if ((threadIdx.x & (WAVE - 1)) == 0 && mx) atomicMax(maxList, mx);
}

// exclusive scan of the workgroup sums by ONE workgroup (a few hundred thousand values)
__global__ __launch_bounds__(1024) void scan_blocks_kernel(uint64_t *blockSum, uint64_t nBlocks, uint64_t *grandTotal) {
    __shared__ uint64_t sWave[1024 / WAVE];
    const uint64_t per = (nBlocks + blockDim.x - 1) / blockDim.x;
    const uint64_t b = min(nBlocks, (uint64_t) threadIdx.x * per), e = min(nBlocks, b + per);
    uint64_t sum = 0;
    for (uint64_t k = b; k < e; k++) sum += blockSum[k];
    uint64_t total;
    uint64_t run = block_excl_scan64(sum, sWave, total);
    for (uint64_t k = b; k < e; k++) { const uint64_t c = blockSum[k]; blockSum[k] = run; run += c; }
    if (threadIdx.x == 0) *grandTotal = total;
}

__global__ __launch_bounds__(SCAN_T) void scan_slots_kernel(const uint32_t *count, uint64_t cells, const uint64_t *blockStart, uint64_t entryShift,
                                                            uint64_t *slots, uint32_t *bits) {
    __shared__ uint64_t sWave[SCAN_T / WAVE];
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_BLOCK + (uint64_t) threadIdx.x * SCAN_PER;
    uint32_t c[SCAN_PER];
    uint64_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; k++) { c[k] = base + k < cells ? count[base + k] : 0u; sum += c[k]; }
    uint64_t total;
    uint64_t run = blockStart[blockIdx.x] + block_excl_scan64(sum, sWave, total);
    uint32_t present = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; k++) {
        if (base + k < cells) slots[base + k] = (run + entryShift) | ((uint64_t) c[k] << 40);
        run += c[k];
        present |= (c[k] ? 1u : 0u) << k;
    }
    // 16 cells per thread: two threads share a presence word
    const uint32_t other = (uint32_t) __shfl_xor((int) present, 1, WAVE);
    if ((threadIdx.x & 1) == 0 && base < cells) bits[base >> 5] = present | (other << 16);
}

// =====================================================================================================
//  lists: sort by target, single entries into their slots
// =====================================================================================================
constexpr int FSPAN = 1024;             // entries a wave stages at most
constexpr uint32_t FLANE_MAX = 64;      // a lane sorts lists up to this length itself
constexpr uint32_t LDS_LIST = 4096;     // a workgroup sorts lists up to this length in LDS

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename P>
__device__ __forceinline__ void insertion_sort_by_target(P a, uint32_t n) {
    for (uint32_t x = 1; x < n; x++) {
        const uint64_t v = a[x];
        uint32_t y = x;
        while (y > 0 && (uint32_t) a[y - 1] > (uint32_t) v) { a[y] = a[y - 1]; y--; }
        a[y] = v;
    }
}

__global__ __launch_bounds__(256) void index_finalize_kernel(uint64_t *slots, uint64_t *entries, uint64_t cells, uint64_t entryShift,
                                                             uint32_t *longList, uint32_t *nLong, uint32_t longCap) {
    __shared__ uint64_t sE[4][FSPAN];
    const int w = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    const uint64_t cell = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t slot = 0;
    if (cell < cells) slot = slots[cell];
    const uint32_t len = cell < cells ? (uint32_t) (slot >> 40) & 0x7FFFFFu : 0u;
    const uint64_t start = (slot & M40) - (cell < cells ? entryShift : 0ull);
    const unsigned long long nonEmpty = __ballot(len != 0u);
    if (!nonEmpty) return;
    const int firstLane = __ffsll((long long) nonEmpty) - 1, lastLane = 63 - __clzll((long long) nonEmpty);
    // the lists of neighbouring cells are neighbours: [base, end) holds every list of the wave
    const uint64_t base = ((uint64_t) (uint32_t) __shfl((int) (uint32_t) (start >> 32), firstLane, WAVE) << 32) | (uint32_t) __shfl((int) (uint32_t) start, firstLane, WAVE);
    const uint64_t endL = start + len;
    const uint64_t end = ((uint64_t) (uint32_t) __shfl((int) (uint32_t) (endL >> 32), lastLane, WAVE) << 32) | (uint32_t) __shfl((int) (uint32_t) endL, lastLane, WAVE);
    uint32_t mx = len;
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) mx = max(mx, (uint32_t) __shfl_xor((int) mx, d, WAVE));
    if (end - base <= (uint64_t) FSPAN && mx <= FLANE_MAX) {
        const uint32_t span = (uint32_t) (end - base);
        for (uint32_t t = (uint32_t) lane; t < span; t += WAVE) sE[w][t] = entries[base + t];
        wave_fence();
        const bool sorts = __ballot(len >= 2u) != 0ull;
        if (len >= 2u) insertion_sort_by_target(&sE[w][(uint32_t) (start - base)], len);
        wave_fence();
        if (sorts) for (uint32_t t = (uint32_t) lane; t < span; t += WAVE) entries[base + t] = sE[w][t];
        if (len == 1u) slots[cell] = (1ull << 63) | sE[w][(uint32_t) (start - base)];
    } else {
        if (len == 1u) slots[cell] = (1ull << 63) | entries[start];
        else if (len >= 2u && len <= 16u) insertion_sort_by_target(entries + start, len);
        else if (len > 16u) {
            const uint32_t k = atomicAdd(nLong, 1u);
            if (k < longCap) longList[k] = (uint32_t) cell;
        }
    }
}

// all comparators ascending (the merge of a block starts with a FLIP: i against its mirror image, then half-cleaners), so that a
// list whose length is no power of two is sorted as if it were padded with +infinity: a comparator that reaches beyond the end is void
__device__ __forceinline__ void bitonic_partner(uint32_t i, uint32_t k, uint32_t j, bool flip, uint32_t &l, uint32_t &r) {
    if (flip) {
        const uint32_t half = k >> 1, blk = i / half, o = i % half;
        l = blk * k + o; r = blk * k + k - 1u - o;
    } else {
        l = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)); r = l + j;
    }
}

// every queued list in chunks of LDS_LIST entries: chunk-local network.  from_k == 2: sorts the chunks (a list of <= LDS_LIST entries is
// done); from_k == k > LDS_LIST: the half-cleaners j = LDS_LIST / 2 .. 1 of merge level k.
__global__ __launch_bounds__(1024) void long_lds_kernel(const uint64_t *slots, uint64_t *entries, uint64_t entryShift, const uint32_t *longList, uint32_t nLong,
                                                        uint32_t fromK) {
    __shared__ uint64_t sK[LDS_LIST];
    const uint32_t item = blockIdx.x;
    if (item >= nLong) return;
    const uint64_t slot = slots[longList[item]];
    const uint32_t len = (uint32_t) (slot >> 40) & 0x7FFFFFu;
    if (fromK > 2u && len <= LDS_LIST) return;
    if (fromK > 2u) { uint32_t P = 1; while (P < len) P <<= 1; if (P < fromK) return; }
    uint64_t *a = entries + ((slot & M40) - entryShift);
    for (uint32_t c0 = 0; c0 < len; c0 += LDS_LIST) {
        const uint32_t n = min(LDS_LIST, len - c0);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < LDS_LIST; t += blockDim.x) sK[t] = t < n ? a[c0 + t] : ~0ull;
        __syncthreads();
        if (fromK == 2u) {
            for (uint32_t k = 2; k <= LDS_LIST; k <<= 1) {
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    const bool flip = j == (k >> 1);
                    for (uint32_t i = threadIdx.x; i < LDS_LIST / 2; i += blockDim.x) {
                        uint32_t l, r;
                        bitonic_partner(i, k, j, flip, l, r);
                        const uint64_t x = sK[l], y = sK[r];
                        if ((uint32_t) x > (uint32_t) y) { sK[l] = y; sK[r] = x; }
                    }
                    __syncthreads();
                }
            }
        } else {
            for (uint32_t j = LDS_LIST / 2; j > 0; j >>= 1) {
                for (uint32_t i = threadIdx.x; i < LDS_LIST / 2; i += blockDim.x) {
                    uint32_t l, r;
                    bitonic_partner(i, fromK, j, false, l, r);
                    const uint64_t x = sK[l], y = sK[r];
                    if ((uint32_t) x > (uint32_t) y) { sK[l] = y; sK[r] = x; }
                }
                __syncthreads();
            }
        }
        for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) a[c0 + t] = sK[t];
    }
}

// one stride of merge level k in HBM (the flip, or a half-cleaner with j >= LDS_LIST), for every queued list that reaches level k
__global__ __launch_bounds__(1024) void long_global_step_kernel(const uint64_t *slots, uint64_t *entries, uint64_t entryShift, const uint32_t *longList, uint32_t nLong,
                                                                uint32_t k, uint32_t j, int flip) {
    const uint32_t item = blockIdx.x;
    if (item >= nLong) return;
    const uint64_t slot = slots[longList[item]];
    const uint32_t len = (uint32_t) (slot >> 40) & 0x7FFFFFu;
    uint32_t P = 1;
    while (P < len) P <<= 1;
    if (len <= LDS_LIST || P < k) return;
    uint64_t *a = entries + ((slot & M40) - entryShift);
    for (uint32_t i = threadIdx.x; i < (P >> 1); i += blockDim.x) {
        uint32_t l, r;
        bitonic_partner(i, k, j, flip != 0, l, r);
        if (r >= len) continue;
        const uint64_t x = a[l], y = a[r];
        if ((uint32_t) x > (uint32_t) y) { a[l] = y; a[r] = x; }
    }
}

// ---- index DB <-> device ------------------------------------------------------------------------------
// (a file is not trusted: offsets that fall, a list of 2^23 entries or more and target numbers beyond the database are counted in bad[0..1];
//  the prefilter kernels would read out of bounds through them)
__global__ __launch_bounds__(256) void lens_from_offsets_kernel(const uint64_t *off /* [n + 1], piece */, uint64_t n, uint32_t *count, unsigned long long *bad) {
    const uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t a = off[k], b = off[k + 1];
    const bool falls = b < a, tooLong = !falls && b - a >= (1ull << 23);      // falling offsets: a corrupt file; a list of 2^23 entries: this build's slot width
    const bool ok = !falls && !tooLong;
    count[k] = ok ? (uint32_t) (b - a) : 0u;
    if (falls) atomicAdd(&bad[0], 1ull);
    if (tooLong) atomicAdd(&bad[2], 1ull);
}
__global__ __launch_bounds__(256) void expand6_kernel(const unsigned char *raw, uint64_t n, uint64_t *out, uint32_t nSeq, unsigned long long *bad) {
    const uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const unsigned char *p = raw + k * 6;
    const uint32_t seq = (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24);
    const uint32_t pos = (uint32_t) p[4] | ((uint32_t) p[5] << 8);
    out[k] = (uint64_t) seq | ((uint64_t) pos << 32);
    if (seq >= nSeq) atomicAdd(&bad[1], 1ull);
}
__global__ __launch_bounds__(256) void pack6_kernel(const uint64_t *in, uint64_t n, unsigned char *raw) {
    const uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t e = in[k];
    unsigned char *p = raw + k * 6;
    p[0] = (unsigned char) e; p[1] = (unsigned char) (e >> 8); p[2] = (unsigned char) (e >> 16); p[3] = (unsigned char) (e >> 24);
    p[4] = (unsigned char) (e >> 32); p[5] = (unsigned char) (e >> 40);
}
__global__ __launch_bounds__(256) void lens_from_slots_kernel(const uint64_t *slots, uint64_t n, uint32_t *count) {
    const uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t s = slots[k];
    count[k] = (s >> 63) ? 1u : (uint32_t) (s >> 40) & 0x7FFFFFu;
}

template <typename T>
__global__ __launch_bounds__(256) void diff_count_kernel(const T *a, const T *b, uint64_t n, unsigned long long *out) {
    const uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const bool d = k < n && a[k] != b[k];
    const unsigned long long m = __ballot(d);
    if ((threadIdx.x & (WAVE - 1)) == 0 && m) atomicAdd(out, (unsigned long long) __popcll(m));
}

inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned) ((n + block - 1) / block); }

// the shared tail: lengths in dCount -> slots, bits; `fill` puts the entries in place; lists sorted, single entries inlined
int finish_lists(uint32_t *dCount, uint64_t cells, uint64_t entryShift, hipStream_t stream, DeviceIndex &out, std::string &err,
                 const std::function<int(uint64_t nEntries)> &fill) {
    const uint64_t nBlocks = (cells + SCAN_BLOCK - 1) / SCAN_BLOCK;
    Tmp<uint64_t> dBlockSum, dTotal;
    Tmp<uint32_t> dMax;
    ICHK(dBlockSum.alloc(nBlocks + 1));
    ICHK(dTotal.alloc(1));
    ICHK(dMax.alloc(2));
    ICHK(hipMemsetAsync(dMax.p, 0, 8, stream));
    hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned) nBlocks), dim3(SCAN_T), 0, stream, dCount, cells, dBlockSum.p, dMax.p);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, dBlockSum.p, nBlocks, dTotal.p);
    ICHK(hipGetLastError());
    uint64_t nEntries = 0;
    uint32_t maxList = 0;
    ICHK(hipMemcpyAsync(&nEntries, dTotal.p, 8, hipMemcpyDeviceToHost, stream));
    ICHK(hipMemcpyAsync(&maxList, dMax.p, 4, hipMemcpyDeviceToHost, stream));
    ICHK(hipStreamSynchronize(stream));
    if (maxList >= (1u << 23)) { err = "a k-mer occurs in 2^23 or more targets: the slot's length field holds 23 bits"; return MK_ERR_UNSUPPORTED; }
    if (nEntries + entryShift >= (1ull << 40)) { err = "index has >= 2^40 entries"; return MK_ERR_UNSUPPORTED; }
    out.cells = cells; out.n_entries = nEntries; out.max_list = maxList;
    ICHK(hipMalloc(reinterpret_cast<void **>(&out.slots), cells * sizeof(uint64_t)));
    ICHK(hipMalloc(reinterpret_cast<void **>(&out.bits), ((cells + 31) / 32 + 1) * sizeof(uint32_t)));
    hipLaunchKernelGGL(scan_slots_kernel, dim3((unsigned) nBlocks), dim3(SCAN_T), 0, stream, dCount, cells, dBlockSum.p, entryShift, out.slots, out.bits);
    ICHK(hipGetLastError());
    const int rc = fill(nEntries);
    if (rc != MK_OK) return rc;
    // sort + inline
    const uint32_t longCap = (uint32_t) std::min<uint64_t>(nEntries / 17 + 1, 0x7FFFFFFFull);
    Tmp<uint32_t> dLong;
    ICHK(dLong.alloc(longCap));
    uint32_t *dNLong = dMax.p + 1;
    hipLaunchKernelGGL(index_finalize_kernel, dim3(grid_for(cells, 256)), dim3(256), 0, stream, out.slots, out.entries, cells, entryShift, dLong.p, dNLong, longCap);
    ICHK(hipGetLastError());
    uint32_t nLong = 0;
    ICHK(hipMemcpyAsync(&nLong, dNLong, 4, hipMemcpyDeviceToHost, stream));
    ICHK(hipStreamSynchronize(stream));
    if (nLong > longCap) { err = "internal: long-list queue overflow"; return MK_ERR_DEVICE; }
    if (nLong) {
        hipLaunchKernelGGL(long_lds_kernel, dim3(nLong), dim3(1024), 0, stream, out.slots, out.entries, entryShift, dLong.p, nLong, 2u);
        uint32_t P = 1;
        while (P < maxList) P <<= 1;
        for (uint32_t k = LDS_LIST * 2; k <= P && k != 0; k <<= 1) {
            hipLaunchKernelGGL(long_global_step_kernel, dim3(nLong), dim3(1024), 0, stream, out.slots, out.entries, entryShift, dLong.p, nLong, k, k >> 1, 1);
            for (uint32_t j = k >> 2; j >= LDS_LIST; j >>= 1)
                hipLaunchKernelGGL(long_global_step_kernel, dim3(nLong), dim3(1024), 0, stream, out.slots, out.entries, entryShift, dLong.p, nLong, k, j, 0);
            hipLaunchKernelGGL(long_lds_kernel, dim3(nLong), dim3(1024), 0, stream, out.slots, out.entries, entryShift, dLong.p, nLong, k);
        }
        ICHK(hipGetLastError());
    }
    ICHK(hipStreamSynchronize(stream));
    return MK_OK;
}

}  // namespace

void DeviceIndex::release() {
    if (masked) (void) hipFree(masked);
    if (slots) (void) hipFree(slots);
    if (bits) (void) hipFree(bits);
    if (entries) (void) hipFree(entries);
    masked = nullptr; slots = nullptr; bits = nullptr; entries = nullptr;
}

int device_build_index(const uint8_t *dRes, const uint64_t *dOff, const std::vector<uint64_t> &offHost, uint32_t nSeq, const SubMat &km,
                       const IndexBuildParams &P, hipStream_t stream, DeviceIndex &out, std::string &err, timed_begin_fn tb, timed_end_fn te) {
    const uint64_t total = offHost[nSeq];
    const int K = P.kmer_size == 7 ? 7 : 6;
    const uint64_t cells = K == 7 ? 1280000000ull : 64000000ull;
    ICHK(hipMalloc(reinterpret_cast<void **>(&out.masked), std::max<uint64_t>(total, 1)));
    if (total) ICHK(hipMemcpyAsync(out.masked, dRes, total, hipMemcpyDeviceToDevice, stream));
    out.masked_residues = 0;
    // ---- tantan
    if (P.mask && nSeq) {
        ScopedHost sh("host_index_tantan");
        std::vector<uint32_t> order(nSeq);
        {   // counting sort by falling length
            uint32_t maxLen = 0;
            for (uint32_t s = 0; s < nSeq; s++) maxLen = std::max<uint32_t>(maxLen, (uint32_t) (offHost[s + 1] - offHost[s]));
            std::vector<uint64_t> at(static_cast<size_t>(maxLen) + 2, 0);
            for (uint32_t s = 0; s < nSeq; s++) at[maxLen - (uint32_t) (offHost[s + 1] - offHost[s]) + 1]++;
            for (size_t k = 1; k < at.size(); k++) at[k] += at[k - 1];
            for (uint32_t s = 0; s < nSeq; s++) order[at[maxLen - (uint32_t) (offHost[s + 1] - offHost[s])]++] = s;
        }
        Tmp<uint32_t> dOrder;
        Tmp<double> dRatio;
        Tmp<unsigned long long> dMasked;
        ICHK(dOrder.alloc(nSeq));
        ICHK(dRatio.alloc(21 * 21));
        ICHK(dMasked.alloc(1));
        ICHK(hipMemcpyAsync(dOrder.p, order.data(), (size_t) nSeq * 4, hipMemcpyHostToDevice, stream));
        double ratio[21 * 21];
        for (int i = 0; i < ALPH; i++)
            for (int j = 0; j < ALPH; j++) ratio[i * 21 + j] = km.prob[i][j] / (km.pback[i] * km.pback[j]);
        ICHK(hipMemcpyAsync(dRatio.p, ratio, sizeof(ratio), hipMemcpyHostToDevice, stream));
        ICHK(hipMemsetAsync(dMasked.p, 0, 8, stream));
        TantanArgs A;
        A.res = dRes; A.masked = out.masked; A.off = dOff; A.order = dOrder.p; A.n_seq = nSeq; A.ratio = dRatio.p;
        A.min_mask_prob = P.mask_prob; A.masked_count = dMasked.p;
        {
            const double pRepeat = 0.005, decay = 0.9;
            double p = pRepeat * ((1 - decay) / (1 - std::pow(decay, TW)));
            for (int i = 0; i < TW; i++) { A.enter[i] = p; p *= decay; }
        }
        const uint64_t nWaves = ((uint64_t) nSeq + WAVE - 1) / WAVE;
        const uint64_t BUDGET = 6ull << 30;                          // bytes of per-position scratch per launch
        std::vector<uint64_t> postOff, scaleOff;
        Tmp<uint64_t> dPostOff, dScaleOff;
        Tmp<float> dPost;
        Tmp<double> dScale;
        uint64_t capWaves = 0, capPost = 0, capScale = 0;
        for (uint64_t w0 = 0; w0 < nWaves;) {
            postOff.clear(); scaleOff.clear();
            uint64_t nPost = 0, nScale = 0, w1 = w0;
            while (w1 < nWaves) {
                const uint32_t s = order[w1 * WAVE];
                const uint64_t Lmax = offHost[s + 1] - offHost[s];
                const uint64_t addP = Lmax * WAVE, addS = (Lmax / TSTEP + 1) * WAVE;
                if (w1 > w0 && (nPost + addP) * 4 + (nScale + addS) * 8 > BUDGET) break;
                postOff.push_back(nPost); scaleOff.push_back(nScale);
                nPost += addP; nScale += addS; w1++;
            }
            const uint64_t nw = w1 - w0;
            if (nw > capWaves) {
                if (dPostOff.p) { (void) hipFree(dPostOff.take()); (void) hipFree(dScaleOff.take()); }
                capWaves = nw;
                ICHK(dPostOff.alloc(capWaves)); ICHK(dScaleOff.alloc(capWaves));
            }
            if (nPost > capPost) { if (dPost.p) (void) hipFree(dPost.take()); capPost = nPost; ICHK(dPost.alloc(capPost)); }
            if (nScale > capScale) { if (dScale.p) (void) hipFree(dScale.take()); capScale = nScale; ICHK(dScale.alloc(capScale)); }
            ICHK(hipMemcpyAsync(dPostOff.p, postOff.data(), nw * 8, hipMemcpyHostToDevice, stream));
            ICHK(hipMemcpyAsync(dScaleOff.p, scaleOff.data(), nw * 8, hipMemcpyHostToDevice, stream));
            A.wave0 = (uint32_t) w0; A.n_waves = (uint32_t) nw; A.post_off = dPostOff.p; A.scale_off = dScaleOff.p; A.post = dPost.p; A.scale = dScale.p;
            const int th = tb ? tb("index_tantan", (double) nPost / WAVE * 10.0, 0) : -1;
            const unsigned grid = (unsigned) ((nw + 3) / 4);
            if (P.tantan_lanes == 4) hipLaunchKernelGGL(tantan_kernel<4>, dim3(grid), dim3(256), 0, stream, A);
            else if (P.tantan_lanes == 2) hipLaunchKernelGGL(tantan_kernel<2>, dim3(grid), dim3(256), 0, stream, A);
            else hipLaunchKernelGGL(tantan_kernel<1>, dim3(grid), dim3(256), 0, stream, A);
            if (te) te(th);
            ICHK(hipGetLastError());
            ICHK(hipStreamSynchronize(stream));                      // (the offset vectors are reused)
            w0 = w1;
        }
        unsigned long long nm = 0;
        ICHK(hipMemcpy(&nm, dMasked.p, 8, hipMemcpyDeviceToHost));
        out.masked_residues = nm;
    }
    // ---- k-mer lists
    CountArgs C;
    C.masked = out.masked; C.off = dOff; C.n_seq = nSeq; C.entry_shift = P.entry_shift;
    for (int a = 0; a < ALPH; a++) C.G.self[a] = (int8_t) km.sub[a][a];
    std::memcpy(C.G.addr, kmer_addr_letters(), 20);
    C.G.kmer_thr = P.kmer_thr; C.G.tiled = (K == 6 && !P.reference_order) ? 1 : 0;
    Tmp<uint32_t> dCount, dFirst;
    ICHK(dCount.alloc(cells));
    const uint64_t nWords = total / 32 + 4;
    ICHK(dFirst.alloc(nWords));
    ICHK(hipMemsetAsync(dCount.p, 0, cells * 4, stream));
    ICHK(hipMemsetAsync(dFirst.p, 0, nWords * 4, stream));
    C.count = dCount.p; C.first_bits = dFirst.p; C.entries = nullptr; C.slots = nullptr;
    // workgroups per launch: the HIP runtime refuses launches of 2^32 threads or more (gridDim.x * blockDim.x), and CT = 256
    const uint32_t GRID_MAX = (uint32_t) knob_long("MK_TEST_INDEX_GRID_MAX", 1l << 22);
    {
        const int th = tb ? tb("index_count", (double) total * 5.0, 0) : -1;
        for (uint32_t s0 = 0; s0 < nSeq; s0 += GRID_MAX) {
            C.seq0 = s0;
            const unsigned grid = (unsigned) std::min<uint64_t>(GRID_MAX, (uint64_t) nSeq - s0);
            if (K == 7) hipLaunchKernelGGL(index_count_kernel<7>, dim3(grid), dim3(CT), 0, stream, C);
            else hipLaunchKernelGGL(index_count_kernel<6>, dim3(grid), dim3(CT), 0, stream, C);
        }
        std::vector<uint2> items;
        const uint64_t span = K == 7 ? 11 : 10;
        for (uint32_t s2 = 0; s2 < nSeq; s2++) {
            const uint64_t L = offHost[s2 + 1] - offHost[s2];
            if (L < span + LONG_STARTS) continue;
            for (uint64_t i0 = 0; i0 < L - span + 1; i0 += CT) items.push_back(make_uint2(s2, (uint32_t) i0));
        }
        Tmp<uint2> dItems;
        if (!items.empty()) {
            ICHK(dItems.alloc(items.size()));
            ICHK(hipMemcpyAsync(dItems.p, items.data(), items.size() * sizeof(uint2), hipMemcpyHostToDevice, stream));
            for (size_t b0 = 0; b0 < items.size(); b0 += GRID_MAX) {
                const uint32_t nb = (uint32_t) std::min<size_t>(GRID_MAX, items.size() - b0);
                if (K == 7) hipLaunchKernelGGL(index_count_long_kernel<7>, dim3(nb), dim3(CT), 0, stream, C, dItems.p + b0, nb);
                else hipLaunchKernelGGL(index_count_long_kernel<6>, dim3(nb), dim3(CT), 0, stream, C, dItems.p + b0, nb);
            }
            ICHK(hipStreamSynchronize(stream));                    // (the item list lives on the host stack)
        }
        if (te) te(th);
        ICHK(hipGetLastError());
    }
    const int rc = finish_lists(dCount.p, cells, P.entry_shift, stream, out, err, [&](uint64_t nEntries) -> int {
        ICHK(hipMalloc(reinterpret_cast<void **>(&out.entries), std::max<uint64_t>(nEntries, 1) * sizeof(uint64_t)));
        C.entries = out.entries; C.slots = out.slots;
        const int th = tb ? tb("index_fill", (double) nEntries * 24.0, 0) : -1;
        for (uint32_t s0 = 0; s0 < nSeq; s0 += GRID_MAX) {
            C.seq0 = s0;
            const unsigned grid = (unsigned) std::min<uint64_t>(GRID_MAX, (uint64_t) nSeq - s0);
            if (K == 7) hipLaunchKernelGGL(index_fill_kernel<7>, dim3(grid), dim3(CT), 0, stream, C);
            else hipLaunchKernelGGL(index_fill_kernel<6>, dim3(grid), dim3(CT), 0, stream, C);
        }
        if (te) te(th);
        ICHK(hipGetLastError());
        return MK_OK;
    });
    return rc;
}

int device_index_from_lists(uint32_t *dCount, uint64_t *dEntriesIn, uint64_t cells, uint64_t nEntries, uint64_t entryShift, hipStream_t stream,
                            DeviceIndex &out, std::string &err) {
    out.entries = dEntriesIn;
    const int rc = finish_lists(dCount, cells, entryShift, stream, out, err, [&](uint64_t n) -> int {
        if (n != nEntries) { err = "the k-mer list offsets of the index do not add up to its entries"; return MK_ERR_ARG; }
        return MK_OK;
    });
    return rc;
}

int device_index_from_file(const uint64_t *hostOffsets, const unsigned char *hostEntries6, uint64_t nEntries, int kmerSize, uint64_t entryShift,
                           uint32_t nSeq, hipStream_t stream, DeviceIndex &out, std::string &err) {
    const uint64_t cells = kmerSize == 7 ? 1280000000ull : 64000000ull;
    if (hostOffsets[0] != 0 || hostOffsets[cells] != nEntries) { err = "the k-mer list offsets of the index do not add up to its entries"; return MK_ERR_ARG; }
    Tmp<uint32_t> dCount;
    Tmp<unsigned long long> dBad;
    ICHK(dCount.alloc(cells));
    ICHK(dBad.alloc(3));
    ICHK(hipMemsetAsync(dBad.p, 0, 24, stream));
    const uint64_t PIECE = 1ull << 26;                                   // cells / entries per upload
    {
        Tmp<uint64_t> dOffPiece;
        ICHK(dOffPiece.alloc(PIECE + 1));
        for (uint64_t c0 = 0; c0 < cells; c0 += PIECE) {
            const uint64_t n = std::min(PIECE, cells - c0);
            ICHK(hipMemcpyAsync(dOffPiece.p, hostOffsets + c0, (n + 1) * 8, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(lens_from_offsets_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, dOffPiece.p, n, dCount.p + c0, dBad.p);
            ICHK(hipStreamSynchronize(stream));
        }
    }
    Tmp<uint64_t> dEntries;                                              // (handed over to the index at the end: an error on the way frees it)
    ICHK(dEntries.alloc(std::max<uint64_t>(nEntries, 1)));
    {
        Tmp<unsigned char> dRaw;
        ICHK(dRaw.alloc(PIECE * 6));
        for (uint64_t e0 = 0; e0 < nEntries; e0 += PIECE) {
            const uint64_t n = std::min(PIECE, nEntries - e0);
            ICHK(hipMemcpyAsync(dRaw.p, hostEntries6 + e0 * 6, n * 6, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(expand6_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, dRaw.p, n, dEntries.p + e0, nSeq, dBad.p);
            ICHK(hipStreamSynchronize(stream));
        }
    }
    unsigned long long bad[3] = {0, 0, 0};
    ICHK(hipMemcpy(bad, dBad.p, 24, hipMemcpyDeviceToHost));
    if (bad[0]) { err = "corrupt index DB: " + std::to_string(bad[0]) + " k-mer list offsets fall"; return MK_ERR_ARG; }
    // (a valid index of a very large or low-complexity database can hold such a list: a limit of this build, as in the build path)
    if (bad[2]) { err = std::to_string(bad[2]) + " k-mers of the index DB occur in 2^23 targets or more: the slots of this build hold 23 bits of list length"; return MK_ERR_UNSUPPORTED; }
    if (bad[1]) { err = "corrupt index DB: " + std::to_string(bad[1]) + " entries name a sequence beyond the database's " + std::to_string(nSeq); return MK_ERR_ARG; }
    uint64_t *ent = dEntries.take();
    const int rc = device_index_from_lists(dCount.p, ent, cells, nEntries, entryShift, stream, out, err);
    if (rc != MK_OK && out.entries != ent) (void) hipFree(ent);          // (normally out.entries == ent and the caller's release() frees it)
    return rc;
}

int device_index_offsets(const DeviceIndex &ix, hipStream_t stream, std::vector<uint64_t> &offsets, std::string &err) {
    const uint64_t cells = ix.cells;
    offsets.assign(cells + 1, 0);
    const uint64_t PIECE = 1ull << 26;
    Tmp<uint32_t> dLen;
    ICHK(dLen.alloc(PIECE));
    std::vector<uint32_t> hLen(PIECE);
    for (uint64_t c0 = 0; c0 < cells; c0 += PIECE) {
        const uint64_t n = std::min(PIECE, cells - c0);
        hipLaunchKernelGGL(lens_from_slots_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, ix.slots + c0, n, dLen.p);
        ICHK(hipMemcpyAsync(hLen.data(), dLen.p, n * 4, hipMemcpyDeviceToHost, stream));
        ICHK(hipStreamSynchronize(stream));
        for (uint64_t k = 0; k < n; k++) offsets[c0 + k + 1] = hLen[k];
    }
    for (uint64_t k = 0; k < cells; k++) offsets[k + 1] += offsets[k];
    if (offsets[cells] != ix.n_entries) { err = "internal: the slots' list lengths do not add up to the entries"; return MK_ERR_DEVICE; }
    return MK_OK;
}

int device_index_entries6(const DeviceIndex &ix, hipStream_t stream, const std::function<bool(const void *, size_t)> &sink, std::string &err) {
    const uint64_t PIECE = 1ull << 26;
    Tmp<unsigned char> dRaw;
    ICHK(dRaw.alloc(PIECE * 6));
    std::vector<unsigned char> hRaw(PIECE * 6);
    for (uint64_t e0 = 0; e0 < ix.n_entries; e0 += PIECE) {
        const uint64_t n = std::min(PIECE, ix.n_entries - e0);
        hipLaunchKernelGGL(pack6_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, ix.entries + e0, n, dRaw.p);
        ICHK(hipMemcpyAsync(hRaw.data(), dRaw.p, n * 6, hipMemcpyDeviceToHost, stream));
        ICHK(hipStreamSynchronize(stream));
        if (!sink(hRaw.data(), n * 6)) { err = "writing the index entries failed"; return MK_ERR_ARG; }
    }
    return MK_OK;
}

int device_index_compare(const DeviceIndex &a, const DeviceIndex &b, uint64_t totalResidues, hipStream_t stream, uint64_t diff[4], std::string &err) {
    diff[0] = diff[1] = diff[2] = diff[3] = ~0ull;
    if (a.cells != b.cells || a.n_entries != b.n_entries) { err = "the indices differ in size"; return MK_OK; }
    Tmp<unsigned long long> d;
    ICHK(d.alloc(4));
    ICHK(hipMemsetAsync(d.p, 0, 32, stream));
    hipLaunchKernelGGL(diff_count_kernel<uint64_t>, dim3(grid_for(a.cells, 256)), dim3(256), 0, stream, a.slots, b.slots, a.cells, d.p);
    hipLaunchKernelGGL(diff_count_kernel<uint32_t>, dim3(grid_for((a.cells + 31) / 32, 256)), dim3(256), 0, stream, a.bits, b.bits, (a.cells + 31) / 32, d.p + 1);
    if (a.n_entries) hipLaunchKernelGGL(diff_count_kernel<uint64_t>, dim3(grid_for(a.n_entries, 256)), dim3(256), 0, stream, a.entries, b.entries, a.n_entries, d.p + 2);
    if (totalResidues) hipLaunchKernelGGL(diff_count_kernel<uint8_t>, dim3(grid_for(totalResidues, 256)), dim3(256), 0, stream, a.masked, b.masked, totalResidues, d.p + 3);
    ICHK(hipGetLastError());
    unsigned long long h[4];
    ICHK(hipMemcpyAsync(h, d.p, 32, hipMemcpyDeviceToHost, stream));
    ICHK(hipStreamSynchronize(stream));
    for (int k = 0; k < 4; k++) diff[k] = h[k];
    return MK_OK;
}

}  // namespace mk
