// metaeuk_amd/csrc/mk_align.hip -- the alignment stage as a device-resident pipeline.
// Per batch (or per prefilter chunk, under mk_search): the prefilter hits go up once (12 B per pair) and only the accepted
// pairs' integers come back (24 B each).
//   expand_pairs_kernel      pair -> query (binary search in the per-query offsets), forward SwJob
//   job order (round 4: no device-wide radix sort)
//                            the score pass wants the jobs grouped by (tile configuration, query) with the targets of a query by falling
//                            length.  The pairs of a query are neighbours already, so: a scan over the QUERIES per tile configuration
//                            (plan_sums / plan_offsets: first job and first wave of every query) and a sort of every query's own pairs
//                            by target length -- one wave up to 64 pairs, a workgroup in LDS / over HBM beyond (order_wave / order_block,
//                            mk_segsort.hpp).  The wave list falls out of the same scan: a wave = the next 64/G (x2 packed) jobs of one query.
//   swp_kernel / sw_kernel   score pass (mk_sw.hip): packed int16, two targets per lane group, for tiles <= 768 rows;
//                            persistent launch (a fixed number of one-wave workgroups per CU pull waves from a counter)
//   gate_count / gate_emit   e-value gate on the score (table per query length) -> position jobs for the ~9 % survivors, numbered in
//                            pair order (block counts + scan: the collected records need no sort afterwards)
//   bin_scan / bin_scatter   position and reverse jobs by (tile configuration, target length class): a counting sort over 11 x 4096 bins
//   sw_kernel                position pass (end cell of the maximum), then rev_jobs_kernel + reverse pass on the reversed
//                            prefixes (start cell)
//   collect_kernel           AlnRaw records in pair order
#include "mk_align.hpp"
#include "mk_kernels.hpp"
#include "mk_segsort.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <omp.h>

namespace mk {

namespace {

constexpr uint32_t KEY_CLS = 4096;                    // target-length classes (16 residues each), falling length = ascending class
// (two more "configurations" behind the tile configurations: the position jobs of profile queries that the transposed kernel takes -- 512- and 768-row
//  profiles, targets of at most 256 residues --, binned by query instead of by target length so that a query's jobs are neighbours in the order)
constexpr int SW_NBINCFG = SW_NCFG + 2;
constexpr uint32_t N_BINS = (uint32_t) SW_NBINCFG * KEY_CLS;
static_assert(SW_NCFG <= 16, "the bounds arrays keep 16 slots per kind");

__device__ __forceinline__ uint32_t len_class(uint32_t tLen) { return KEY_CLS - 1 - min(tLen >> 4, KEY_CLS - 1); }
__device__ __forceinline__ uint32_t sort_key(uint32_t qLen, uint32_t tLen) { return (uint32_t) sw_cfg_of(qLen) * KEY_CLS + len_class(tLen); }

// short launches of this stage run beside the persistent workgroups of the prefilter (mk_search): they ask for issue priority
__device__ __forceinline__ void helper_prio() { __builtin_amdgcn_s_setprio(3); }

// pair -> forward job (the query of pair p is the one whose hit range holds p)
__global__ __launch_bounds__(256) void expand_pairs_kernel(AlignView V, const uint64_t *hitOff, const mk_hit *hits, uint64_t n,
                                                           SwJob *jobs, uint32_t *badTarget,
                                                           unsigned long long *work /* [2 * cfg]: bytes, [2 * cfg + 1]: cells of the forward pass (statistics) */) {
    helper_prio();
    __shared__ unsigned long long sWork[2 * SW_NCFG];
    for (int k = threadIdx.x; k < 2 * SW_NCFG; k += blockDim.x) sWork[k] = 0;
    __syncthreads();
    const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        uint32_t lo = 0, hi = V.n_queries;              // largest q with hitOff[q] <= p
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (hitOff[mid] <= p) lo = mid; else hi = mid; }
        const uint32_t q = lo, t = hits[p].seq_id;
        SwJob j;
        j.q_start = (uint32_t) V.q_off[q]; j.q_len = (uint32_t) (V.q_off[q + 1] - V.q_off[q]);
        if (t < V.n_targets) { j.t_start = V.t_off[t]; j.t_len = (uint32_t) (V.t_off[t + 1] - V.t_off[t]); }
        else { j.t_start = 0; j.t_len = 0; atomicMax(badTarget, (uint32_t) p + 1u); }     // reported to the caller, nothing is aligned
        j.q_step = 1; j.t_step = 1; j.slot = (uint32_t) p;
        jobs[p] = j;
        const int c = sw_cfg_of(j.q_len);
        atomicAdd(&sWork[2 * c], (unsigned long long) (j.t_len + 2u * j.q_len + (uint32_t) (sizeof(SwJob) + sizeof(SwOut))));
        atomicAdd(&sWork[2 * c + 1], (unsigned long long) j.q_len * j.t_len);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * SW_NCFG; k += blockDim.x) if (sWork[k]) atomicAdd(&work[k], sWork[k]);
}

// ---- the score pass's job order, per query -------------------------------------------------------------------------------------
// jobs grouped by tile configuration, inside it by query (ascending), inside a query by falling target length: the DPs of a wave share
// their query (one LDS profile per wave) and have similar numbers of columns.  The order inside a query is a matter of speed only --
// every job writes to its own slot -- so a query with more pairs than an LDS tile holds is ordered run by run (ORDER_TILE pairs each).
constexpr uint32_t ORDER_WAVE_MAX = 64, ORDER_TILE = 2048;
constexpr int PLAN_NV = 2 * SW_NCFG + 1;              // per tile configuration: pairs, waves; runs of the queries beyond a wave
constexpr uint32_t PLAN_BLOCK = 256;
struct PlanArgs {
    const uint64_t *hitOff; const uint64_t *q_off; uint32_t nq, nPairs; bool narrow;
    uint32_t *jobOff, *waveOff;          // per query: its first job / first wave in the ordered lists
    uint32_t *runQuery, *runIndex;       // one entry per run of ORDER_TILE pairs of the queries with more than ORDER_WAVE_MAX pairs
    uint32_t *blockSums;                 // [blocks][PLAN_NV + 1]
    uint32_t *bounds;                    // [0..SW_NCFG] job bounds, [16..16+SW_NCFG] wave bounds, [33] runs
    uint32_t *waveStart;
};
__device__ __forceinline__ void plan_values(const PlanArgs &A, uint32_t q, uint32_t (&v)[PLAN_NV], int &cfg, uint32_t &n) {
#pragma unroll
    for (int k = 0; k < PLAN_NV; k++) v[k] = 0;
    cfg = 0; n = 0;
    if (q >= A.nq) return;
    n = (uint32_t) (A.hitOff[q + 1] - A.hitOff[q]);
    if (n == 0) return;
    cfg = sw_cfg_of((uint32_t) (A.q_off[q + 1] - A.q_off[q]));
    const uint32_t dpw = sw_cfg_jobs_per_wave(cfg, A.narrow);
#pragma unroll
    for (int c = 0; c < SW_NCFG; c++) if (c == cfg) { v[2 * c] = n; v[2 * c + 1] = (n + dpw - 1u) / dpw; }
    v[2 * SW_NCFG] = n > ORDER_WAVE_MAX ? (n + ORDER_TILE - 1u) / ORDER_TILE : 0u;
}
__global__ __launch_bounds__(PLAN_BLOCK) void plan_sums_kernel(PlanArgs A) {
    helper_prio();
    __shared__ uint32_t sm[PLAN_NV * 16];
    uint32_t v[PLAN_NV], excl[PLAN_NV], total[PLAN_NV], n;
    int cfg;
    plan_values(A, blockIdx.x * PLAN_BLOCK + threadIdx.x, v, cfg, n);
    segsort::block_scan<PLAN_NV>(v, excl, total, sm);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < PLAN_NV; k++) A.blockSums[blockIdx.x * (PLAN_NV + 1) + k] = total[k];
}
__global__ __launch_bounds__(PLAN_BLOCK) void plan_offsets_kernel(PlanArgs A) {
    helper_prio();
    __shared__ uint32_t sm[PLAN_NV * 16];
    uint32_t v[PLAN_NV], w[PLAN_NV], excl[PLAN_NV], base[PLAN_NV], all[PLAN_NV], total[PLAN_NV], n;
#pragma unroll
    for (int k = 0; k < PLAN_NV; k++) { v[k] = 0; w[k] = 0; }
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += PLAN_BLOCK)
#pragma unroll
        for (int k = 0; k < PLAN_NV; k++) {
            const uint32_t x = A.blockSums[b * (PLAN_NV + 1) + k];
            if (b < blockIdx.x) v[k] += x;
            w[k] += x;
        }
    segsort::block_scan<PLAN_NV>(v, excl, base, sm);          // base = sums of the blocks before this one
    segsort::block_scan<PLAN_NV>(w, excl, all, sm);           // all = sums of every block
    const uint32_t q = blockIdx.x * PLAN_BLOCK + threadIdx.x;
    int cfg;
    plan_values(A, q, v, cfg, n);
    segsort::block_scan<PLAN_NV>(v, excl, total, sm);
    uint32_t jobBase = 0, waveBase = 0;                       // first job / wave of this query's tile configuration
#pragma unroll
    for (int c = 0; c < SW_NCFG; c++) if (c < cfg) { jobBase += all[2 * c]; waveBase += all[2 * c + 1]; }
    if (q < A.nq) {
        uint32_t jo = jobBase, wo = waveBase;
#pragma unroll
        for (int c = 0; c < SW_NCFG; c++) if (c == cfg) { jo += base[2 * c] + excl[2 * c]; wo += base[2 * c + 1] + excl[2 * c + 1]; }
        A.jobOff[q] = jo; A.waveOff[q] = wo;
        const uint32_t runs = v[2 * SW_NCFG], r0 = base[2 * SW_NCFG] + excl[2 * SW_NCFG];
        for (uint32_t r = 0; r < runs; r++) { A.runQuery[r0 + r] = q; A.runIndex[r0 + r] = r; }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        uint32_t jb = 0, wb = 0;
        for (int c = 0; c < SW_NCFG; c++) { A.bounds[c] = jb; A.bounds[16 + c] = wb; jb += all[2 * c]; wb += all[2 * c + 1]; }
        A.bounds[SW_NCFG] = jb; A.bounds[16 + SW_NCFG] = wb; A.bounds[33] = all[2 * SW_NCFG];
        A.waveStart[wb] = A.nPairs;                           // closes the wave list
    }
}
struct OrderArgs {
    const uint64_t *hitOff; const uint64_t *q_off; const mk_hit *hits; const uint64_t *t_off; uint32_t nq, nTargets; bool narrow;
    const uint32_t *jobOff, *waveOff, *runQuery, *runIndex;
    uint32_t *order, *waveStart;
};
__device__ __forceinline__ uint64_t order_key(const OrderArgs &A, uint64_t p) {
    const uint32_t t = A.hits[p].seq_id;
    const uint32_t tLen = t < A.nTargets ? (uint32_t) (A.t_off[t + 1] - A.t_off[t]) : 0u;
    return ((uint64_t) len_class(tLen) << 32) | (uint64_t) (uint32_t) p;
}
// queries of at most 64 pairs: one wave each, ranks by counting
__global__ __launch_bounds__(256) void order_wave_kernel(OrderArgs A) {
    helper_prio();
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (q >= A.nq) return;
    const uint64_t p0 = A.hitOff[q];
    const uint32_t n = (uint32_t) (A.hitOff[q + 1] - p0);
    if (n == 0 || n > ORDER_WAVE_MAX) return;
    const uint64_t key = lane < n ? order_key(A, p0 + lane) : ~0ull;
    const uint32_t r = segsort::wave_rank(key, n);
    const uint32_t jo = A.jobOff[q];
    if (lane < n) A.order[jo + r] = (uint32_t) (p0 + lane);
    const uint32_t dpw = sw_cfg_jobs_per_wave(sw_cfg_of((uint32_t) (A.q_off[q + 1] - A.q_off[q])), A.narrow);
    const uint32_t nW = (n + dpw - 1u) / dpw;
    if (lane < nW) A.waveStart[A.waveOff[q] + lane] = jo + lane * dpw;
}
// larger queries: one workgroup per run of ORDER_TILE pairs, sorted in LDS
__global__ __launch_bounds__(256) void order_block_kernel(OrderArgs A) {
    helper_prio();
    __shared__ uint64_t sK[ORDER_TILE];
    const uint32_t q = A.runQuery[blockIdx.x], run = A.runIndex[blockIdx.x];
    const uint64_t p0 = A.hitOff[q];
    const uint32_t n = (uint32_t) (A.hitOff[q + 1] - p0);
    const uint32_t b = run * ORDER_TILE, m = min(ORDER_TILE, n - b);
    const uint32_t P = segsort::pow2_at_least(m);
    for (uint32_t t = threadIdx.x; t < P; t += 256) sK[t] = t < m ? order_key(A, p0 + b + t) : ~0ull;
    __syncthreads();
    segsort::lds_sort<256>(sK, P);
    const uint32_t jo = A.jobOff[q] + b;
    for (uint32_t t = threadIdx.x; t < m; t += 256) A.order[jo + t] = (uint32_t) sK[t];
    const uint32_t dpw = sw_cfg_jobs_per_wave(sw_cfg_of((uint32_t) (A.q_off[q + 1] - A.q_off[q])), A.narrow);   // (divides ORDER_TILE)
    const uint32_t w0 = b / dpw, nW = (m + dpw - 1u) / dpw;
    for (uint32_t w = threadIdx.x; w < nW; w += 256) A.waveStart[A.waveOff[q] + w0 + w] = jo + w * dpw;
}
static_assert(ORDER_TILE % 8 == 0, "a run holds whole waves");

// ---- exclusive prefix sums of a uint32 array in two sweeps (1024 elements per workgroup) ----
constexpr uint32_t SCAN_TILE = 1024;
__global__ __launch_bounds__(256) void scan_sums_kernel(const uint32_t *in, uint32_t n, uint32_t *blockSums) {
    helper_prio();
    __shared__ uint32_t sm[16];
    const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x * 4u;
    uint32_t v[1] = {0}, excl[1], total[1];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) if (i0 + k < n) v[0] += in[i0 + k];
    segsort::block_scan<1>(v, excl, total, sm);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = total[0];
}
__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t *in, uint32_t n, uint32_t *out, const uint32_t *blockSums, uint32_t *grandTotal) {
    helper_prio();
    __shared__ uint32_t sm[16];
    uint32_t v[1] = {0}, excl[1], base[1], total[1];
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256) v[0] += blockSums[b];
    segsort::block_scan<1>(v, excl, base, sm);
    const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x * 4u;
    uint32_t x[4];
    v[0] = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { x[k] = i0 + k < n ? in[i0 + k] : 0u; v[0] += x[k]; }
    segsort::block_scan<1>(v, excl, total, sm);
    uint32_t run = base[0] + excl[0];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { if (i0 + k < n) out[i0 + k] = run; run += x[k]; }
    if (grandTotal && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *grandTotal = base[0] + total[0];
}

// ---- e-value gate on the forward score (table per query length); survivors get a position job (the same DP again, this time
// with end-position tracking).  Two sweeps: survivors per 256-pair block, (scan), then every survivor takes its number in pair
// order -- the records collected at the end are ordered by pair without a sort.
__device__ __forceinline__ bool gate_pass(const SwJob *fwdJobs, const SwOut *fwdOut, uint64_t n, const GateEntry *gate, uint64_t p, SwJob &j) {
    if (p >= n) return false;
    const int score = fwdOut[p].score;
    if (score <= 0) return false;
    j = fwdJobs[p];
    const GateEntry g = gate[j.q_len];
    bool pass = score >= g.s0;
    if (!pass && score < 256) pass = (g.mask[score >> 5] >> (score & 31)) & 1u;
    return pass;
}
__global__ __launch_bounds__(256) void gate_count_kernel(const SwJob *fwdJobs, const SwOut *fwdOut, uint64_t n, const GateEntry *gate, uint32_t *blockCount) {
    helper_prio();
    SwJob j;
    const int c = __syncthreads_count(gate_pass(fwdJobs, fwdOut, n, gate, (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, j));
    if (threadIdx.x == 0) blockCount[blockIdx.x] = (uint32_t) c;
}
__global__ __launch_bounds__(256) void gate_emit_kernel(const SwJob *fwdJobs, const SwOut *fwdOut, uint64_t n, const GateEntry *gate, const uint32_t *blockStart,
                                                        uint32_t *posPair, SwJob *posJobs, uint32_t *posKeys, int32_t *posScore, uint32_t *hist,
                                                        unsigned long long *work /* [2 * cfg]: bytes, [2 * cfg + 1]: cells of the position pass (statistics) */,
                                                        bool useBound, int tposFirstCfg /* >= 0: jobs of this tile configuration and the next go to the transposed kernel */) {
    helper_prio();
    __shared__ unsigned long long sWork[2 * SW_NCFG];
    __shared__ uint32_t sWave[4];
    for (int k = threadIdx.x; k < 2 * SW_NCFG; k += blockDim.x) sWork[k] = 0;
    const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    SwJob j;
    const bool pass = gate_pass(fwdJobs, fwdOut, n, gate, p, j);
    const unsigned long long m = __ballot(pass);
    if (lane == 0) sWave[w] = (uint32_t) __popcll(m);
    __syncthreads();
    if (pass) {
        uint32_t r = blockStart[blockIdx.x] + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
        for (int k = 0; k < w; k++) r += sWave[k];
        // the score pass bounds the column of the end cell (swp_kernel; the int32 forward kernels of the largest tile report the end cell itself):
        // the position pass asks for the FIRST column that reaches the maximum, which the columns behind that bound cannot change
        const int bound = fwdOut[p].end_col;
        if (useBound && bound >= 0 && (uint32_t) bound + 1u < j.t_len) j.t_len = (uint32_t) bound + 1u;
        const int c = sw_cfg_of(j.q_len);
        atomicAdd(&sWork[2 * c], (unsigned long long) (j.t_len + 2u * j.q_len + (uint32_t) (sizeof(SwJob) + sizeof(SwOut))));
        atomicAdd(&sWork[2 * c + 1], (unsigned long long) j.q_len * j.t_len);
        posPair[r] = (uint32_t) p;
        posScore[r] = fwdOut[p].score;
        j.slot = r;
        posJobs[r] = j;
        uint32_t key = sort_key(j.q_len, j.t_len);
        if (tposFirstCfg >= 0 && (c == tposFirstCfg || c == tposFirstCfg + 1) && j.t_len <= SW_TPOS_MAX_ROWS)
            key = (uint32_t) (SW_NCFG + c - tposFirstCfg) * KEY_CLS + ((j.q_start >> 7) & (KEY_CLS - 1u));      // (by query: its first column, 128 columns per bin)
        posKeys[r] = key;
        atomicAdd(&hist[key], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * SW_NCFG; k += blockDim.x) if (sWork[k]) atomicAdd(&work[k], sWork[k]);
}

// reverse jobs (reversed prefixes ending at the forward end cell) of the survivors; the position pass must reproduce the
// score the gate saw
__global__ __launch_bounds__(256) void rev_jobs_kernel(const SwJob *posJobs, const SwOut *posOut, const uint32_t *posPair, const SwOut *fwdOut, uint32_t n,
                                                       SwJob *revJobs, uint32_t *revKeys, uint32_t *hist, uint32_t *mismatch) {
    helper_prio();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const SwOut o = posOut[r];
    const SwJob f = posJobs[r];
    if (o.score != fwdOut[posPair[r]].score || o.end_row < 0 || o.end_col < 0) atomicAdd(mismatch, 1u);
    SwJob j;
    j.q_len = (uint32_t) max(o.end_row, 0) + 1; j.t_len = (uint32_t) max(o.end_col, 0) + 1;
    j.q_start = f.q_start + (uint32_t) max(o.end_row, 0); j.q_step = -1;
    j.t_start = f.t_start + (uint64_t) max(o.end_col, 0); j.t_step = -1;
    j.slot = r;
    revJobs[r] = j;
    const uint32_t key = sort_key(j.q_len, j.t_len);
    revKeys[r] = key;
    atomicAdd(&hist[key], 1u);
}

// ---- position / reverse jobs by (tile configuration, target length class): a counting sort over the N_BINS keys ----
// bin_scan: hist -> cursor (exclusive prefix), hist is cleared for the next use; bounds[c] = first job of tile configuration c,
// bounds[32 + c] = its first occupied key
constexpr uint32_t BIN_THREADS = 1024, BINS_PER_THREAD = N_BINS / BIN_THREADS;
static_assert(N_BINS % BIN_THREADS == 0, "bins per thread");
__global__ __launch_bounds__(BIN_THREADS) void bin_scan_kernel(uint32_t *hist, uint32_t *cursor, uint32_t *bounds) {
    helper_prio();
    __shared__ uint32_t sm[16];
    __shared__ uint32_t sFirst[SW_NBINCFG];
    if (threadIdx.x < SW_NBINCFG) sFirst[threadIdx.x] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t k0 = threadIdx.x * BINS_PER_THREAD;
    uint32_t v[1] = {0}, excl[1], total[1];
    for (uint32_t k = 0; k < BINS_PER_THREAD; k++) {
        const uint32_t h = hist[k0 + k];
        v[0] += h;
        if (h) atomicMin(&sFirst[(k0 + k) / KEY_CLS], k0 + k);
    }
    segsort::block_scan<1>(v, excl, total, sm);
    uint32_t run = excl[0];
    for (uint32_t k = 0; k < BINS_PER_THREAD; k++) {
        const uint32_t h = hist[k0 + k];
        cursor[k0 + k] = run;
        if ((k0 + k) % KEY_CLS == 0) bounds[(k0 + k) / KEY_CLS] = run;
        run += h;
        hist[k0 + k] = 0;
    }
    if (threadIdx.x == 0) bounds[SW_NBINCFG] = total[0];
    __syncthreads();
    if (threadIdx.x < SW_NBINCFG) bounds[32 + threadIdx.x] = sFirst[threadIdx.x] == 0xFFFFFFFFu ? 0u : sFirst[threadIdx.x];
}
__global__ __launch_bounds__(256) void bin_scatter_kernel(const uint32_t *keys, uint32_t n, uint32_t *cursor, uint32_t *order) {
    helper_prio();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) order[atomicAdd(&cursor[keys[r]], 1u)] = r;
}

__global__ __launch_bounds__(256) void collect_kernel(const uint32_t *posPair, uint32_t n, const SwOut *fwdOut, const SwOut *revOut, AlnRaw *out) {
    helper_prio();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SwOut f = fwdOut[i], b = revOut[i];          // both indexed by survivor number = pair order
    AlnRaw a;
    a.pair = posPair[i]; a.score = f.score; a.q_end = f.end_row; a.t_end = f.end_col;
    // reverse score kept in q_start when it disagrees (the reference EXITs on that, :466-473)
    if (b.score != f.score) { a.q_start = -2; a.t_start = b.score; }
    else { a.q_start = f.end_row - b.end_row; a.t_start = f.end_col - b.end_col; }
    out[i] = a;
}

// ---- device-side assembly of the accepted alignments -------------------------------------------------------------------------
struct AssembleView {
    const AlnRaw *raw; uint32_t n;
    const uint64_t *hitOff; const mk_hit *hits; uint32_t nq;
    const uint64_t *q_off; const uint64_t *t_off;
    const double *evalTab; const int32_t *lenIdx; uint32_t maxLen, smax; const int32_t *bitScore; const uint32_t *sortKey;
    double evalThr; int minAlnLen;
    mk_alignment *tmp; uint8_t *pass; uint32_t *flags /* [0] score beyond the table [1] forward/backward mismatch */;
    unsigned long long *revWork;     // [2 * cfg]: bytes, [2 * cfg + 1]: cells of the reverse pass (statistics)
};

// one lane per accepted pair: Matcher::getSWResult's tail (Matcher.cpp:100-164) + Alignment::checkCriteria (Alignment.cpp:548-567)
__global__ __launch_bounds__(256) void assemble_kernel(AssembleView A) {
    __shared__ unsigned long long sWork[2 * SW_NCFG];
    for (int k = threadIdx.x; k < 2 * SW_NCFG; k += blockDim.x) sWork[k] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.n) {
        const uint32_t ql = (uint32_t) A.raw[i].q_end + 1u, tl = (uint32_t) A.raw[i].t_end + 1u;
        const int c = sw_cfg_of(ql);
        atomicAdd(&sWork[2 * c], (unsigned long long) (tl + 2u * ql + (uint32_t) (sizeof(SwJob) + sizeof(SwOut))));
        atomicAdd(&sWork[2 * c + 1], (unsigned long long) ql * tl);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * SW_NCFG; k += blockDim.x) if (sWork[k]) atomicAdd(&A.revWork[k], sWork[k]);
    if (i >= A.n) return;
    const AlnRaw r = A.raw[i];
    A.pass[i] = 0;
    if (r.q_start == -2) { atomicAdd(&A.flags[1], 1u); return; }
    uint32_t lo = 0, hi = A.nq;                       // largest q with hitOff[q] <= pair
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (A.hitOff[mid] <= (uint64_t) r.pair) lo = mid; else hi = mid; }
    const uint32_t t = A.hits[r.pair].seq_id;
    const int qLen = (int) (A.q_off[lo + 1] - A.q_off[lo]);
    const int tLen = (int) (A.t_off[t + 1] - A.t_off[t]);
    if ((uint32_t) r.score >= A.smax || (uint32_t) qLen > A.maxLen || A.lenIdx[qLen] < 0) { atomicAdd(&A.flags[0], 1u); return; }
    mk_alignment a;
    a.db_key = t; a.q_len = qLen; a.db_len = tLen; a.raw_score = r.score;
    a.evalue = A.evalTab[(size_t) A.lenIdx[qLen] * A.smax + (uint32_t) r.score];
    const auto cov = [](unsigned s, unsigned e, unsigned len) { return (float) (min(len, max(s, e)) - min(s, e) + 1u) / (float) len; };
    a.qcov = cov((unsigned) r.q_start, (unsigned) r.q_end, (unsigned) qLen);
    a.dbcov = cov((unsigned) r.t_start, (unsigned) r.t_end, (unsigned) tLen);
    a.q_start = r.q_start; a.q_end = r.q_end; a.db_start = r.t_start; a.db_end = r.t_end;
    a.aln_len = max(abs(r.q_end - r.q_start), abs(r.t_end - r.t_start)) + 1;
    const unsigned qAln = max((unsigned) r.q_end - (unsigned) r.q_start, 1u), dbAln = max((unsigned) r.t_end - (unsigned) r.t_start, 1u);
    const uint16_t s16 = (uint16_t) r.score;
    float sid = (float) ((double) ((float) (int) s16 / (float) max(qAln, dbAln)) * 0.1656 + 0.1141);   // Matcher.cpp:160-164 (float / float, then double)
    sid = fminf(sid, 1.0f);
    a.seq_id = fmaxf(0.0f, sid);
    a.bit_score = A.bitScore[r.score];
    A.tmp[i] = a;
    A.pass[i] = (a.evalue <= A.evalThr && a.aln_len >= A.minAlnLen) ? 1 : 0;
}


// one lane per query: its records are raw[first .. next) (raw is ordered by pair = by query): how many pass
__global__ __launch_bounds__(256) void assemble_count_kernel(AssembleView A, uint32_t *cnt) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= A.nq) return;
    const auto first_at = [&](uint64_t want) { uint32_t lo = 0, hi = A.n; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t) A.raw[mid].pair < want) lo = mid + 1; else hi = mid; } return lo; };
    const uint32_t b = first_at(A.hitOff[q]), e = first_at(A.hitOff[q + 1]);
    uint32_t c = 0;
    for (uint32_t k = b; k < e; k++) c += A.pass[k];
    cnt[q] = c;
}

// The per-query order (Alignment.cpp:403-405, Matcher::compareHits), one lane per passing record: its place in the query's list is the
// number of the query's passing records that sort before it.  compareHits is a total order inside a list (the last key, the target's DB
// key, is unique there), so the ranks are a permutation.  A fragment has a handful of alignments; a profile query has hundreds to
// thousands, and the n^2 comparisons of a list spread over its n lanes.
__global__ __launch_bounds__(256) void assemble_rank_kernel(AssembleView A, const uint32_t *off, mk_alignment *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n || !A.pass[k]) return;
    const uint64_t pair = A.raw[k].pair;
    uint32_t q;
    { uint32_t lo = 0, hi = A.nq; while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (A.hitOff[mid] <= pair) lo = mid; else hi = mid; } q = lo; }
    const auto first_at = [&](uint64_t want) { uint32_t lo = 0, hi = A.n; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t) A.raw[mid].pair < want) lo = mid + 1; else hi = mid; } return lo; };
    const uint32_t b = first_at(A.hitOff[q]), e = first_at(A.hitOff[q + 1]);
    const mk_alignment a = A.tmp[k];
    const uint32_t ka = A.sortKey ? A.sortKey[a.db_key] : a.db_key;
    uint32_t rank = 0;
    for (uint32_t j = b; j < e; j++) {
        if (j == k || !A.pass[j]) continue;
        const mk_alignment &p = A.tmp[j];
        const double pe = p.evalue;
        bool before;
        if (pe != a.evalue) before = pe < a.evalue;
        else if (p.bit_score != a.bit_score) before = p.bit_score > a.bit_score;
        else if (p.db_len != a.db_len) before = p.db_len < a.db_len;
        else {
            const uint32_t kp = A.sortKey ? A.sortKey[p.db_key] : p.db_key;
            before = kp != ka ? kp < ka : j < k;        // (a caller-installed list may name a target twice: the input order then decides)
        }
        rank += before ? 1u : 0u;
    }
    out[off[q] + rank] = a;
}

}  // namespace

void build_assemble_tables(const Evaluer &ev, const std::vector<uint64_t> &qOff, AssembleTables &t, std::unordered_map<uint32_t, std::vector<double>> *rowCache) {
    uint32_t maxLen = 0;
    const size_t n = qOff.size() - 1;
    for (size_t i = 0; i < n; i++) maxLen = std::max<uint32_t>(maxLen, (uint32_t) (qOff[i + 1] - qOff[i]));
    static std::atomic<uint64_t> nextId{1};
    t.id = nextId.fetch_add(1);
    t.smax = 4096;
    t.lenIdx.assign((size_t) maxLen + 1, -1);
    std::vector<uint32_t> lens;
    for (size_t i = 0; i < n; i++) { const uint32_t L = (uint32_t) (qOff[i + 1] - qOff[i]); if (t.lenIdx[L] < 0) { t.lenIdx[L] = 0; lens.push_back(L); } }
    std::sort(lens.begin(), lens.end());
    for (size_t k = 0; k < lens.size(); k++) t.lenIdx[lens[k]] = (int32_t) k;
    t.evalue.assign(lens.size() * (size_t) t.smax, 0.0);
    std::vector<uint8_t> cached(lens.size(), 0);
    if (rowCache)
        for (size_t k = 0; k < lens.size(); k++) {
            auto it = rowCache->find(lens[k]);
            if (it != rowCache->end() && it->second.size() == t.smax) { std::memcpy(&t.evalue[k * t.smax], it->second.data(), t.smax * sizeof(double)); cached[k] = 1; }
        }
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t k = 0; k < lens.size(); k++)
        if (!cached[k])
            for (uint32_t s = 0; s < t.smax; s++) t.evalue[k * t.smax + s] = ev.evalue((double) s, (double) lens[k]);
    if (rowCache)
        for (size_t k = 0; k < lens.size(); k++)
            if (!cached[k]) (*rowCache)[lens[k]].assign(&t.evalue[k * t.smax], &t.evalue[k * t.smax] + t.smax);
    if (t.bitScore.empty()) {
        t.bitScore.resize(32768);
        for (int s = 0; s < 32768; s++) t.bitScore[s] = static_cast<int>(ev.bitScore((double) s) + 0.5);
    }
}

// pass(score) table per query length present in the batch
void build_gate_table(const Evaluer &ev, double evalThr, const std::vector<uint64_t> &qOff, std::vector<GateEntry> &table, const AssembleTables *T) {
    uint32_t maxLen = 0;
    const size_t n = qOff.size() - 1;
    for (size_t i = 0; i < n; i++) maxLen = std::max<uint32_t>(maxLen, (uint32_t) (qOff[i + 1] - qOff[i]));
    std::vector<uint8_t> present(maxLen + 1, 0);
    for (size_t i = 0; i < n; i++) present[qOff[i + 1] - qOff[i]] = 1;
    table.assign(maxLen + 1, GateEntry{1 << 30, {0, 0, 0, 0, 0, 0, 0, 0}});
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t L = 1; L <= maxLen; L++) {
        if (!present[L]) continue;
        GateEntry g;
        std::memset(&g, 0, sizeof(g));
        // e-values fall monotonically with the score beyond the finite-size regime; scan high -> low for the last failure
        const int SCAN = 4096;
        // (the e-values of the scores below smax are in the assembly table when there is one: computed once)
        const double *row = (T && L < T->lenIdx.size() && T->lenIdx[L] >= 0) ? T->evalue.data() + (size_t) T->lenIdx[L] * T->smax : nullptr;
        const auto eval = [&](int s) { return (row && (uint32_t) s < T->smax) ? row[s] : ev.evalue((double) s, (double) L); };
        int lastFail = 0;
        for (int s = SCAN; s >= 1; s--) {
            const bool pass = !(eval(s) > evalThr);
            if (!pass) { lastFail = s; break; }
        }
        g.s0 = lastFail >= SCAN ? (1 << 30) : lastFail + 1;
        for (int s = 1; s < 256 && s < g.s0; s++)
            if (!(eval(s) > evalThr)) g.mask[s >> 5] |= 1u << (s & 31);
        table[L] = g;
    }
}

#define ACHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return MK_ERR_DEVICE; } } while (0)
#define ANULL(p) do { if (!(p)) { err = "device scratch allocation failed (" #p ")"; return MK_ERR_DEVICE; } } while (0)

// exclusive prefix sums of n uint32 (in -> out, in == out allowed); the grand total goes to *total (device) when given
static int device_scan(const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *total, hipStream_t stream, std::string &err) {
    if (n == 0) { if (total) ACHK(hipMemsetAsync(total, 0, 4, stream)); return MK_OK; }
    const uint32_t nb = (n + SCAN_TILE - 1u) / SCAN_TILE;
    uint32_t *sums = (uint32_t *) dev_scratch("align_scansums", (size_t) nb * 4);
    ANULL(sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(nb), dim3(256), 0, stream, in, n, sums);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, stream, in, n, out, (const uint32_t *) sums, total);
    ACHK(hipGetLastError());
    return MK_OK;
}

// launch shapes of the stage, fixed at the first call (several alignment workers run this code at once)
struct AlignShapes {
    int cus = 256;
    uint32_t persistentBlocks[SW_NCFG];      // persistent score pass: one-wave workgroups per tile configuration
    uint32_t unitsPerBlock = 0;              // MK_SW_UNITS_PER_BLOCK: short-lived workgroups instead of the persistent launch
    int knownForce = -1, knownWaves = 12;
    bool fwdLargeFirst = false;              // MK_SW_FWD_LARGE_FIRST=1
    bool multiLargeFirst = true;             // MK_SW_MULTI_LARGE_FIRST=0: the register classes of the position / reverse passes small tiles first
    bool multiPrio = true;                   // MK_SW_MULTI_PRIO=0: the persistent position / reverse workgroups do not ask for issue priority
    uint32_t multiPerCu[3] = {24, 12, 12};   // MK_SW_MULTI_WAVES: one-wave workgroups per CU of the three register classes (<= 64, 96 .. 256, >= 384 rows)
};
// MK_SW_EARLY_EXIT=0 (MK_DEBUG=1): the position / reverse passes run every column of their jobs (no bound from the score pass, no stop at the known
// score).  Read per call, like MK_SW_MULTI, so that one process can compare both forms (tests/test_gpu_parity.py::test_full_dp_position_passes...)
static bool sw_early_exit() { return knob_long("MK_SW_EARLY_EXIT", 1) != 0; }
static const AlignShapes &align_shapes() {
    static AlignShapes S;
    static std::once_flag once;
    std::call_once(once, [] {
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&S.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || S.cus <= 0) S.cus = 256;
        // persistent forward launch: this many one-wave workgroups per CU and tile configuration (MK_SW_WAVES_PER_CU = one number or one per
        // tile configuration, comma separated); 12 for every tile (round 2: 16 for the small tiles, 6-8 for the large ones; profiles/r03_search_tuning.txt)
        int perCu[SW_NCFG];
        for (int c = 0; c < SW_NCFG; c++) perCu[c] = 12;
        if (const char *e = knob("MK_SW_WAVES_PER_CU")) {
            int k = 0, last = 16;
            for (const char *p = e; *p && k < SW_NCFG; k++) {
                last = std::max(1, atoi(p));
                perCu[k] = last;
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
            for (; k < SW_NCFG; k++) perCu[k] = last;
        }
        for (int c = 0; c < SW_NCFG; c++) S.persistentBlocks[c] = (uint32_t) (S.cus * perCu[c]);
        S.unitsPerBlock = (uint32_t) std::max(0L, knob_long("MK_SW_UNITS_PER_BLOCK", 0));
        S.knownForce = (int) knob_long("MK_SW_KNOWN", -1);
        S.knownWaves = (int) std::max(1L, knob_long("MK_SW_KNOWN_WAVES", 12));
        S.multiPrio = knob_long("MK_SW_MULTI_PRIO", 1) != 0;
        S.fwdLargeFirst = knob_long("MK_SW_FWD_LARGE_FIRST", 0) != 0;
        S.multiLargeFirst = knob_long("MK_SW_MULTI_LARGE_FIRST", 1) != 0;
        if (const char *e = knob("MK_SW_MULTI_WAVES")) {
            int k = 0;
            for (const char *p = e; *p && k < 3; k++) { S.multiPerCu[k] = (uint32_t) std::max(1, atoi(p)); while (*p && *p != ',') p++; if (*p == ',') p++; }
        }
    });
    return S;
}

// the transposed position pass (mk_sw.hip: swtp_kernel): profile queries, the 32-lane tiles (512 and 768 rows: configurations 8 and 9), not with the
// one-launch-per-register-class form (its bounds end at the tile configurations).  MK_SW_TPOS=0 switches it off (read per call)
static int tpos_first_cfg(const AlignView &V) {
    static_assert(SW_NCFG == 11, "configurations 8 / 9 = 512 / 768 rows");
    if (!V.q_prof || knob_long("MK_SW_TPOS", 1) == 0 || knob_long("MK_SW_MULTI", 0) != 0 || knob_long("MK_SW_NARROW", -1) == 0) return -1;
    return 8;
}

// position / reverse pass: the jobs (their keys and the key histogram are filled by the producing kernel) ordered by
// (tile configuration, target length class) with a counting sort, one launch per tile configuration
static int run_sorted_sw(const AlignView &V, const mk_params &P, const SwJob *jobs, SwOut *out, const uint32_t *keys, uint32_t *hist, uint32_t *order,
                         uint32_t n, const char *tag, hipStream_t stream, std::string &err,
                         timed_begin_fn tb, timed_end_fn te, int *handles /* SW_NCFG, may be null */,
                         const int32_t *knownScore /* device, by job slot: the maximum every job will reach (null: unknown) */) {
    if (handles) for (int c = 0; c < SW_NCFG; c++) handles[c] = -1;
    if (n == 0) return MK_OK;
    const AlignShapes &S = align_shapes();
    uint32_t *dCursor = (uint32_t *) dev_scratch("align_bincursor", (size_t) N_BINS * 4);
    uint32_t *dBounds = (uint32_t *) dev_scratch("align_bounds", 64 * sizeof(uint32_t));
    uint32_t *hb = (uint32_t *) pinned_scratch("align_bounds_h", 64 * sizeof(uint32_t));
    ANULL(dCursor); ANULL(dBounds); ANULL(hb);
    int th = tb("align_sort", 12.0 * n, 0);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(BIN_THREADS), 0, stream, hist, dCursor, dBounds);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys, n, dCursor, order);
    te(th);
    ACHK(hipGetLastError());
    // One persistent launch per register class, the tile configurations' bounds read on the device (launch_sw_multi): no copy-back, no host
    // synchronisation, 3 launches instead of up to 10 per pass.  Not for queries beyond the largest tile (row tiles need a border whose size
    // follows the bounds) and not where the packed known-score kernel is preferred (the stage alone on the GPU: mk_align).
    const bool knownPreferred = knownScore && !V.q_prof && (S.knownForce >= 0 ? S.knownForce != 0 : !V.co_resident);
    // MEASURED AND NOT THE DEFAULT (profiles/r05_search_engine.txt, calls c8 / c9, same box, two repetitions each): sw_pos_* + sw_rev_* kernel time
    // 700 -> 230 ms per config-2 step and host_align_total 1 910 -> 1 430 ms, but the STEP is 2.5 % slower (875-880 against 855-860 ms): the prefilter
    // chain is the critical path of the queued search, and the persistent position / reverse workgroups slow its large-tier kernel (450 -> 467 ms
    // per step) more than the short launches with their host round trips did.  MK_SW_MULTI=1 (MK_DEBUG=1) selects it; read per call so that the
    // tests can run both.
    const bool multiLaunch = knob_long("MK_SW_MULTI", 0) != 0;
    if (multiLaunch && !knownPreferred && V.max_q_len <= (uint32_t) sw_cfg_rows(SW_NCFG - 1)) {
        uint32_t *dCnt = (uint32_t *) dev_scratch("align_multicounters", 64);
        ANULL(dCnt);
        ACHK(hipMemsetAsync(dCnt, 0, 64, stream));
        SwLaunch L;
        L.q_res = V.q_res; L.q_bias8 = V.q_bias8; L.q_prof = V.q_prof; L.t_res = V.t_res; L.mat = V.mat_aln;
        L.jobs = jobs; L.out = out; L.n_jobs = n; L.order = order;
        L.boundary = nullptr; L.boundary_stride = 0; L.boundary_job0 = 0;
        L.wave_start = nullptr; L.n_waves = 0; L.work_counter = nullptr; L.persistent_blocks = 0; L.units_per_block = 0;
        L.known_score = sw_early_exit() ? knownScore : nullptr;
        L.gap_open = P.gap_open; L.gap_extend = P.gap_extend;
        static const char *clsName[3] = {"rows32_64", "rows96_256", "rows384_1024"};
        static const int clsFirst[3] = {0, 3, 7};
        for (int kk = 0; kk < 3; kk++) {
            const int k = S.multiLargeFirst ? 2 - kk : kk;
            char nm[64];
            snprintf(nm, sizeof(nm), "%s_%s", tag, clsName[k]);
            th = tb(nm, 0, 0);
            if (handles) handles[clsFirst[k]] = th;
            const uint32_t units = (n + 3u) / 4u + (uint32_t) SW_NCFG;          // upper bound of the units of the class
            ACHK(launch_sw_multi(L, dBounds, dCnt + k, k, std::min(units, (uint32_t) S.cus * S.multiPerCu[k]), stream, S.multiPrio));
            te(th);
        }
        return MK_OK;
    }
    ACHK(hipMemcpyAsync(hb, dBounds, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    ACHK(sync_wait(stream, "wait_align"));
    for (int c = 0; c < SW_NCFG; c++) {
        const uint32_t lo = hb[c], hi = hb[c + 1];
        if (hi <= lo) continue;
        SwLaunch L;
        L.q_res = V.q_res; L.q_bias8 = V.q_bias8; L.q_prof = V.q_prof; L.t_res = V.t_res; L.mat = V.mat_aln;
        L.jobs = jobs; L.out = out; L.n_jobs = hi - lo; L.order = order + lo;
        L.boundary = nullptr; L.boundary_stride = 0; L.boundary_job0 = 0;
        L.wave_start = nullptr; L.n_waves = 0; L.work_counter = nullptr; L.persistent_blocks = 0; L.units_per_block = 0; L.known_score = nullptr;
        L.gap_open = P.gap_open; L.gap_extend = P.gap_extend;
        if (c == SW_NCFG - 1 && V.max_q_len > (uint32_t) sw_cfg_rows(c)) {
            const uint32_t cls = KEY_CLS - 1 - (hb[32 + c] % KEY_CLS);          // largest target-length class in this bucket
            const uint32_t stride = std::min<uint32_t>(V.max_t_len, (cls + 1) * 16);
            L.boundary = (uint32_t *) dev_scratch("align_border", (size_t) (hi - lo) * stride * sizeof(uint32_t));
            ANULL(L.boundary);
            L.boundary_stride = stride;
        }
        char nm[64];
        snprintf(nm, sizeof(nm), "%s_rows%d", tag, sw_cfg_rows(c));
        th = tb(nm, 0, 0);
        if (handles) handles[c] = th;
        // the score is known: packed int16, eight independent DPs per wave, persistent -- when the stage has the GPU to itself (mk_align:
        // 108 ms instead of 121 ms per config-2 pass).  Beside the prefilter of mk_search its 11-22 KB of profiles per wave cost the other
        // stage more LDS than the kernel saves (measured: 1.12 s per step against 1.10 s), so the int32 kernels stay there.  MK_SW_KNOWN=0/1 forces.
        if (knownScore && sw_cfg_known(c) && !V.q_prof && (S.knownForce >= 0 ? S.knownForce != 0 : !V.co_resident)) {
            uint32_t *dWork = (uint32_t *) dev_scratch("align_knowncounters", 64 * sizeof(uint32_t));     // (a buffer per scratch lane = per worker)
            ANULL(dWork);
            uint32_t *counter = dWork + (strcmp(tag, "sw_rev") == 0 ? 16 : 0) + c;
            ACHK(hipMemsetAsync(counter, 0, sizeof(uint32_t), stream));
            L.known_score = knownScore; L.work_counter = counter;
            // 11 / 22.5 KB of profiles per wave: few waves per CU, the LDS is shared with the prefilter workgroups of the other stream
            L.persistent_blocks = (uint32_t) S.cus * (uint32_t) (sw_cfg_rows(c) <= 32 ? S.knownWaves : std::max(1, S.knownWaves / 2));
            ACHK(launch_sw_known(L, c, stream));
        } else {
            if (sw_early_exit()) L.known_score = knownScore;           // sw_unit stops a DP once its known maximum has been seen in a finished column
            ACHK(launch_sw(L, c, stream));
        }
        te(th);
    }
    // the jobs binned for the transposed position kernel (gate_emit_kernel, tposFirstCfg): behind the tile configurations in the order
    for (int pc = 0; pc < SW_NBINCFG - SW_NCFG; pc++) {
        const uint32_t lo = hb[SW_NCFG + pc], hi = hb[SW_NCFG + pc + 1];
        if (hi <= lo) continue;
        if (!knownScore || !V.q_prof) { err = "transposed position jobs without a known score"; return MK_ERR_DEVICE; }
        SwLaunch L;
        L.q_res = V.q_res; L.q_bias8 = V.q_bias8; L.q_prof = V.q_prof; L.t_res = V.t_res; L.mat = V.mat_aln;
        L.jobs = jobs; L.out = out; L.n_jobs = hi - lo; L.order = order + lo;
        L.boundary = nullptr; L.boundary_stride = 0; L.boundary_job0 = 0;
        L.wave_start = nullptr; L.n_waves = 0; L.work_counter = nullptr; L.persistent_blocks = 0; L.units_per_block = 0; L.known_score = knownScore;
        L.gap_open = P.gap_open; L.gap_extend = P.gap_extend;
        const int rows = sw_cfg_rows(8 + pc);
        char nm[64];
        snprintf(nm, sizeof(nm), "%s_rows%dt", tag, rows);
        th = tb(nm, 0, 0);
        ACHK(launch_sw_tpos(L, rows, (uint32_t) S.cus * (uint32_t) std::max(1L, knob_long("MK_SW_TPOS_WAVES", 32)), stream));
        te(th);
    }
    return MK_OK;
}

// forward pass: jobs ordered by (configuration, query, target length); one wave per (query, <= 64/G targets)
static int run_shared_fwd(const AlignView &V, const mk_params &P, const uint64_t *dHitOff, const mk_hit *dHits, const SwJob *jobs, SwOut *out,
                          uint32_t n, hipStream_t stream, std::string &err, timed_begin_fn tb, timed_end_fn te, int *handles /* SW_NCFG */) {
    for (int c = 0; c < SW_NCFG; c++) handles[c] = -1;
    if (n == 0) return MK_OK;
    const AlignShapes &S = align_shapes();
    const uint32_t nq = V.n_queries;
    const uint32_t nb = (nq + PLAN_BLOCK - 1u) / PLAN_BLOCK;
    const uint32_t maxRuns = n / ORDER_TILE + nq + 1;                      // every query adds at most one partial run
    uint32_t *dOrder = (uint32_t *) dev_scratch("align_order", (size_t) n * 4);
    uint32_t *dWave = (uint32_t *) dev_scratch("align_wavestart", ((size_t) n + 1) * 4);
    uint32_t *dPlan = (uint32_t *) dev_scratch("align_plan", ((size_t) nq * 2 + (size_t) maxRuns * 2 + (size_t) nb * (PLAN_NV + 1)) * 4);
    uint32_t *dBounds = (uint32_t *) dev_scratch("align_bounds", 64 * sizeof(uint32_t));
    uint32_t *hb = (uint32_t *) pinned_scratch("align_bounds_h", 64 * sizeof(uint32_t));
    uint32_t *dWork = (uint32_t *) dev_scratch("align_workcounters", 256);        // [cfg]: the score kernels' wave counters, [16 (1 + class) + cfg]: the transposed kernels'
    ANULL(dOrder); ANULL(dWave); ANULL(dPlan); ANULL(dBounds); ANULL(hb); ANULL(dWork);
    ACHK(hipMemsetAsync(dWork, 0, 256, stream));
    // profile queries meet short targets (ORF fragments): waves of 8 jobs on every packed tile, the transposed score kernel (mk_sw.hip: swt_kernel) for the
    // waves whose fragments fit 256 rows.  MK_SW_NARROW=0 / 1 forces (read per call: a test compares both forms in one process)
    const long narrowForce = knob_long("MK_SW_NARROW", -1);
    const bool narrow = narrowForce >= 0 ? narrowForce != 0 : V.q_prof != nullptr;
    PlanArgs A;
    A.hitOff = dHitOff; A.q_off = V.q_off; A.nq = nq; A.nPairs = n; A.narrow = narrow;
    A.jobOff = dPlan; A.waveOff = dPlan + nq; A.runQuery = dPlan + 2 * (size_t) nq; A.runIndex = A.runQuery + maxRuns; A.blockSums = A.runIndex + maxRuns;
    A.bounds = dBounds; A.waveStart = dWave;
    int th = tb("align_sort", 24.0 * n, 0);
    hipLaunchKernelGGL(plan_sums_kernel, dim3(nb), dim3(PLAN_BLOCK), 0, stream, A);
    hipLaunchKernelGGL(plan_offsets_kernel, dim3(nb), dim3(PLAN_BLOCK), 0, stream, A);
    ACHK(hipGetLastError());
    ACHK(hipMemcpyAsync(hb, dBounds, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    OrderArgs O;
    O.hitOff = dHitOff; O.q_off = V.q_off; O.hits = dHits; O.t_off = V.t_off; O.nq = nq; O.nTargets = V.n_targets; O.narrow = narrow;
    O.jobOff = A.jobOff; O.waveOff = A.waveOff; O.runQuery = A.runQuery; O.runIndex = A.runIndex; O.order = dOrder; O.waveStart = dWave;
    hipLaunchKernelGGL(order_wave_kernel, dim3((nq + 3) / 4), dim3(256), 0, stream, O);
    ACHK(hipGetLastError());
    ACHK(sync_wait(stream, "wait_align"));                                   // the bounds (and the number of runs) are on the host
    if (hb[33] > 0) {
        hipLaunchKernelGGL(order_block_kernel, dim3(hb[33]), dim3(256), 0, stream, O);
        ACHK(hipGetLastError());
    }
    te(th);
    // Small tiles first.  (Large first -- so that the handful of long DPs never form the tail of the pass -- was measured in round 5 and is worse beside
    // persistent workgroups: a launch takes the wave slots its predecessor IN THE SAME STREAM has just freed; behind the tiny planning kernels a
    // low-parallelism launch waits for another stream's persistent kernel to end instead.  sw_fwd_rows384 / 512 / 768: 64 / 46 / 41 -> 116 / 101 / 61 ms
    // of kernel time per step, profiles/r05_search_engine.txt.  MK_SW_FWD_LARGE_FIRST=1 restores it.)
    for (int cc = 0; cc < SW_NCFG; cc++) {
        const int c = S.fwdLargeFirst ? SW_NCFG - 1 - cc : cc;
        const uint32_t lo = hb[c], hi = hb[c + 1], wlo = hb[16 + c], whi = hb[16 + c + 1];
        if (hi <= lo || whi <= wlo) continue;
        SwLaunch L;
        L.q_res = V.q_res; L.q_bias8 = V.q_bias8; L.q_prof = V.q_prof; L.t_res = V.t_res; L.mat = V.mat_aln;
        L.jobs = jobs; L.out = out; L.n_jobs = hi - lo; L.order = dOrder;         // wave_start holds absolute ordered positions
        L.boundary = nullptr; L.boundary_stride = 0; L.boundary_job0 = lo;
        L.wave_start = dWave + wlo; L.n_waves = whi - wlo;
        L.work_counter = dWork + c; L.work_counter_t = dWork + 16 + c; L.persistent_blocks = S.persistentBlocks[c]; L.units_per_block = S.unitsPerBlock;
        L.gap_open = P.gap_open; L.gap_extend = P.gap_extend;
        L.narrow = narrow;
        // the transposed score pass: profile queries, the 32-lane tiles (512 / 768 rows; on the 16-lane tiles the classic layout already ramps over 15
        // steps only and wins: 5 -> 10, 12 -> 17, 54 -> 58 ms at 192 / 256 / 384 rows, profiles/r06_transposed_score_pass.txt)
        {
            static const long tMax = knob_long("MK_SW_T_MAXROWS", 256), tMinCfg = knob_long("MK_SW_T_MINCFG", 512);
            L.t_max_rows = (narrow && V.q_prof && sw_cfg_packed(c) && sw_cfg_rows(c) >= (int) tMinCfg) ? (uint32_t) std::max(0L, std::min(256L, tMax)) : 0u;
        }
        L.known_score = nullptr;
        if (c == SW_NCFG - 1 && V.max_q_len > (uint32_t) sw_cfg_rows(c)) {
            // queries beyond the largest tile run in row tiles with an HBM border per job; the border is as long as the
            // longest target of the bucket (the first key only bounds its own query)
            const uint32_t stride = V.max_t_len;
            L.boundary = (uint32_t *) dev_scratch("align_border", (size_t) (hi - lo) * stride * sizeof(uint32_t));
            ANULL(L.boundary);
            L.boundary_stride = stride;
        }
        char nm[64];
        snprintf(nm, sizeof(nm), "sw_fwd_rows%d", sw_cfg_rows(c));
        th = tb(nm, 0, 0);
        handles[c] = th;
        if (sw_cfg_packed(c)) ACHK(launch_sw_score(L, c, stream));   // score only, packed int16, two targets per lane group
        else ACHK(launch_sw(L, c, stream));
        te(th);
    }
    return MK_OK;
}

// a timing handle at tile configuration c stands for the configurations from c up to the next handle (one launch per register class covers
// several; with one launch per configuration the ones in between are empty)
static void set_range_work(const int *handles, const unsigned long long *work /* [2 * cfg]: bytes, [2 * cfg + 1]: cells */, timed_set_fn ts) {
    for (int c = 0; c < SW_NCFG; c++) {
        if (handles[c] < 0) continue;
        double bytes = 0, cells = 0;
        for (int k = c; k < SW_NCFG && (k == c || handles[k] < 0); k++) { bytes += (double) work[2 * k]; cells += (double) work[2 * k + 1]; }
        ts(handles[c], bytes, cells);
    }
}

int run_align_device(const AlignView &V, const uint64_t *hitOffHost, const mk_hit *hitsHost, uint64_t nPairs,
                     const std::vector<GateEntry> &gate, const mk_params &P, hipStream_t stream,
                     const double *fwdWork /* per cfg: bytes, cells; may be null */,
                     const AlnRaw **out, size_t *nOut, std::string &err, timed_begin_fn tb, timed_end_fn te, timed_set_fn ts,
                     AssembleArgs *assemble) {
    *out = nullptr; *nOut = 0;
    if (assemble) { assemble->done = assemble->tables != nullptr; assemble->nOut = 0; if (assemble->counts) std::fill(assemble->counts, assemble->counts + V.n_queries, 0u); }
    if (nPairs == 0) return MK_OK;
    if (nPairs >= 0x7FFFFFFFull) { err = "more than 2^31 pairs in one batch: split the batch"; return MK_ERR_UNSUPPORTED; }
    const uint32_t n = (uint32_t) nPairs;
    const uint32_t nGateBlocks = (n + 255u) / 256u;
    uint64_t *dHitOff = (uint64_t *) dev_scratch("align_hitoff", ((size_t) V.n_queries + 1) * sizeof(uint64_t));
    mk_hit *dHits = (mk_hit *) dev_scratch("align_hits", (size_t) n * sizeof(mk_hit));
    SwJob *dJobs = (SwJob *) dev_scratch("align_jobs", (size_t) n * sizeof(SwJob));
    SwOut *dOut = (SwOut *) dev_scratch("align_out", (size_t) n * sizeof(SwOut));
    GateEntry *dGate = (GateEntry *) dev_scratch("align_gate", gate.size() * sizeof(GateEntry));
    uint32_t *dCount = (uint32_t *) dev_scratch("align_count", 16);
    uint32_t *dHist = (uint32_t *) dev_scratch("align_binhist", (size_t) N_BINS * 4);
    uint32_t *dGateBlk = (uint32_t *) dev_scratch("align_gateblk", (size_t) nGateBlocks * 4);
    ANULL(dHitOff); ANULL(dHits); ANULL(dJobs); ANULL(dOut); ANULL(dGate); ANULL(dCount); ANULL(dHist); ANULL(dGateBlk);
    ACHK(hipMemcpyAsync(dHitOff, hitOffHost, ((size_t) V.n_queries + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    ACHK(hipMemcpyAsync(dHits, hitsHost, (size_t) n * sizeof(mk_hit), hipMemcpyHostToDevice, stream));
    ACHK(hipMemcpyAsync(dGate, gate.data(), gate.size() * sizeof(GateEntry), hipMemcpyHostToDevice, stream));
    ACHK(hipMemsetAsync(dOut, 0, (size_t) n * sizeof(SwOut), stream));
    ACHK(hipMemsetAsync(dCount, 0, 16, stream));
    ACHK(hipMemsetAsync(dHist, 0, (size_t) N_BINS * 4, stream));
    unsigned long long *dFwdWork = (unsigned long long *) dev_scratch("align_fwdwork", 2 * SW_NCFG * 8);
    unsigned long long *hFwdWork = (unsigned long long *) pinned_scratch("align_fwdwork_h", 2 * SW_NCFG * 8);
    ANULL(dFwdWork); ANULL(hFwdWork);
    ACHK(hipMemsetAsync(dFwdWork, 0, 2 * SW_NCFG * 8, stream));
    int th = tb("align_expand", 44.0 * n, 0);
    hipLaunchKernelGGL(expand_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, V, dHitOff, dHits, (uint64_t) n, dJobs, dCount + 1, dFwdWork);
    te(th);
    ACHK(hipGetLastError());
    ACHK(hipMemcpyAsync(hFwdWork, dFwdWork, 2 * SW_NCFG * 8, hipMemcpyDeviceToHost, stream));     // (lands before run_shared_fwd's synchronisation)
    int hFwd[SW_NCFG], hRev[SW_NCFG];
    int rc = run_shared_fwd(V, P, dHitOff, dHits, dJobs, dOut, n, stream, err, tb, te, hFwd);
    (void) fwdWork;
    for (int c = 0; c < SW_NCFG; c++) if (hFwd[c] >= 0) ts(hFwd[c], (double) hFwdWork[2 * c], (double) hFwdWork[2 * c + 1]);
    if (rc != MK_OK) return rc;
    // e-value gate on the forward scores; the survivors (at most n) get a position pass, then the reverse pass
    th = tb("align_gate", 52.0 * n, 0);
    unsigned long long *dPosWork = (unsigned long long *) dev_scratch("align_poswork", 2 * SW_NCFG * 8);
    unsigned long long *hPosWork = (unsigned long long *) pinned_scratch("align_poswork_h", 2 * SW_NCFG * 8);
    ANULL(dPosWork); ANULL(hPosWork);
    ACHK(hipMemsetAsync(dPosWork, 0, 2 * SW_NCFG * 8, stream));
    hipLaunchKernelGGL(gate_count_kernel, dim3(nGateBlocks), dim3(256), 0, stream, dJobs, dOut, (uint64_t) n, dGate, dGateBlk);
    ACHK(hipGetLastError());
    if ((rc = device_scan(dGateBlk, nGateBlocks, dGateBlk, dCount, stream, err)) != MK_OK) return rc;
    uint32_t *hCount = (uint32_t *) pinned_scratch("align_count_h", 16);
    ANULL(hCount);
    ACHK(hipMemcpyAsync(hCount, dCount, 8, hipMemcpyDeviceToHost, stream));
    ACHK(sync_wait(stream, "wait_align"));
    if (hCount[1] != 0) { err = "prefilter hit " + std::to_string(hCount[1] - 1) + " names a target outside the DB"; return MK_ERR_ARG; }
    const uint32_t nRev = hCount[0];
    if (nRev == 0) { te(th); return MK_OK; }
    uint32_t *dRevPair = (uint32_t *) dev_scratch("align_revpair", (size_t) nRev * 4);
    SwJob *dPosJobs = (SwJob *) dev_scratch("align_posjobs", (size_t) nRev * sizeof(SwJob));
    int32_t *dPosScore = (int32_t *) dev_scratch("align_posscore", (size_t) nRev * 4);
    uint32_t *dKeys = (uint32_t *) dev_scratch("align_keys", (size_t) nRev * 4), *dOrder = (uint32_t *) dev_scratch("align_posorder", (size_t) nRev * 4);
    SwOut *dPosOut = (SwOut *) dev_scratch("align_posout", (size_t) nRev * sizeof(SwOut));
    SwOut *dRevOut = (SwOut *) dev_scratch("align_revout", (size_t) nRev * sizeof(SwOut));
    SwJob *dRevJobs = (SwJob *) dev_scratch("align_revjobs", (size_t) nRev * sizeof(SwJob));
    ANULL(dRevPair); ANULL(dPosJobs); ANULL(dPosScore); ANULL(dKeys); ANULL(dOrder); ANULL(dPosOut); ANULL(dRevOut); ANULL(dRevJobs);
    hipLaunchKernelGGL(gate_emit_kernel, dim3(nGateBlocks), dim3(256), 0, stream, dJobs, dOut, (uint64_t) n, dGate, (const uint32_t *) dGateBlk,
                       dRevPair, dPosJobs, dKeys, dPosScore, dHist, dPosWork, sw_early_exit(), tpos_first_cfg(V));
    te(th);
    ACHK(hipGetLastError());
    ACHK(hipMemcpyAsync(hPosWork, dPosWork, 2 * SW_NCFG * 8, hipMemcpyDeviceToHost, stream));
    ACHK(hipMemsetAsync(dPosOut, 0, (size_t) nRev * sizeof(SwOut), stream));
    ACHK(hipMemsetAsync(dRevOut, 0, (size_t) nRev * sizeof(SwOut), stream));
    int hPos[SW_NCFG];
    rc = run_sorted_sw(V, P, dPosJobs, dPosOut, dKeys, dHist, dOrder, nRev, "sw_pos", stream, err, tb, te, hPos, dPosScore);
    if (rc != MK_OK) return rc;
    set_range_work(hPos, hPosWork, ts);
    hipLaunchKernelGGL(rev_jobs_kernel, dim3((nRev + 255) / 256), dim3(256), 0, stream, dPosJobs, dPosOut, dRevPair, dOut, nRev, dRevJobs, dKeys, dHist, dCount + 2);   // (bin_scan cleared the histogram)
    ACHK(hipGetLastError());
    rc = run_sorted_sw(V, P, dRevJobs, dRevOut, dKeys, dHist, dOrder, nRev, "sw_rev", stream, err, tb, te, hRev, dPosScore);    // (rev job r = survivor r: same score)
    if (rc != MK_OK) return rc;
    // collect: the survivors are numbered in pair order
    AlnRaw *dRaw = (AlnRaw *) dev_scratch("align_raw", (size_t) nRev * sizeof(AlnRaw));
    ANULL(dRaw);
    th = tb("align_collect", 64.0 * nRev, 0);
    hipLaunchKernelGGL(collect_kernel, dim3((nRev + 255) / 256), dim3(256), 0, stream, (const uint32_t *) dRevPair, nRev, (const SwOut *) dPosOut, (const SwOut *) dRevOut, dRaw);
    te(th);
    ACHK(hipGetLastError());
    AlnRaw *hRaw = (AlnRaw *) pinned_scratch("align_raw_host", (size_t) nRev * sizeof(AlnRaw));
    ANULL(hRaw);
    bool assembled = false;
    if (assemble && assemble->tables) {
        // e-value (table), bit score, sequence identity, coverage, criteria and the per-query order on the device: the host only
        // receives the finished records
        const AssembleTables &T = *assemble->tables;
        double *dEval = (double *) dev_scratch("asm_evalue", T.evalue.size() * sizeof(double) + 8);
        int32_t *dLenIdx = (int32_t *) dev_scratch("asm_lenidx", T.lenIdx.size() * sizeof(int32_t));
        int32_t *dBit = (int32_t *) dev_scratch("asm_bitscore", T.bitScore.size() * sizeof(int32_t));
        mk_alignment *dTmp = (mk_alignment *) dev_scratch("asm_tmp", (size_t) nRev * sizeof(mk_alignment));
        mk_alignment *dFinal = (mk_alignment *) dev_scratch("asm_final", (size_t) nRev * sizeof(mk_alignment));
        uint8_t *dPass = (uint8_t *) dev_scratch("asm_pass", nRev);
        uint32_t *dCnt = (uint32_t *) dev_scratch("asm_cnt", ((size_t) V.n_queries + 1) * 4), *dOff = (uint32_t *) dev_scratch("asm_off", ((size_t) V.n_queries + 1) * 4);
        uint32_t *dFlags = (uint32_t *) dev_scratch("asm_flags", 16);
        uint32_t *hFlags = (uint32_t *) pinned_scratch("asm_flags_h", 16);
        ANULL(dEval); ANULL(dLenIdx); ANULL(dBit); ANULL(dTmp); ANULL(dFinal); ANULL(dPass); ANULL(dCnt); ANULL(dOff); ANULL(dFlags); ANULL(hFlags);
        // the tables belong to the batch (same for every range of an mk_search): uploaded when they change.  The device copies live in the
        // scratch buffers of the calling thread's LANE (a worker of mk_search and the caller of mk_align can share lane 0), so what a lane holds
        // is remembered per lane -- table id and the buffer it went to (a scratch buffer that grew has lost its content)
        {
            constexpr int MAX_LANES = 16;
            struct Uploaded { uint64_t id = 0; const void *eval = nullptr, *lenIdx = nullptr, *bit = nullptr; uint64_t epoch = 0; };
            static Uploaded uploaded[MAX_LANES];
            static std::mutex upMutex;
            const int lane = scratch_lane();
            std::lock_guard<std::mutex> g(upMutex);
            Uploaded local;
            Uploaded &U = (lane >= 0 && lane < MAX_LANES) ? uploaded[lane] : local;
            if (U.id != T.id || U.eval != dEval || U.lenIdx != dLenIdx || U.bit != dBit || U.epoch != scratch_epoch()) {   // (epoch: mk_shutdown released every buffer)
                ACHK(hipMemcpyAsync(dEval, T.evalue.data(), T.evalue.size() * sizeof(double), hipMemcpyHostToDevice, stream));
                ACHK(hipMemcpyAsync(dLenIdx, T.lenIdx.data(), T.lenIdx.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
                ACHK(hipMemcpyAsync(dBit, T.bitScore.data(), T.bitScore.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
                // the copies read T's vectors, which the next batch may rebuild: they must have left the host before anybody can do that
                ACHK(sync_wait(stream, "wait_align"));
                U.id = T.id; U.eval = dEval; U.lenIdx = dLenIdx; U.bit = dBit; U.epoch = scratch_epoch();
            }
        }
        unsigned long long *dWork = (unsigned long long *) dev_scratch("asm_revwork", 2 * SW_NCFG * 8);
        unsigned long long *hWork = (unsigned long long *) pinned_scratch("asm_revwork_h", 2 * SW_NCFG * 8);
        ANULL(dWork); ANULL(hWork);
        ACHK(hipMemsetAsync(dWork, 0, 2 * SW_NCFG * 8, stream));
        ACHK(hipMemsetAsync(dFlags, 0, 16, stream));
        AssembleView AV;
        AV.revWork = dWork;
        AV.raw = dRaw; AV.n = nRev; AV.hitOff = dHitOff; AV.hits = dHits; AV.nq = V.n_queries; AV.q_off = V.q_off; AV.t_off = V.t_off;
        AV.evalTab = dEval; AV.lenIdx = dLenIdx; AV.maxLen = (uint32_t) T.lenIdx.size() - 1; AV.smax = T.smax; AV.bitScore = dBit; AV.sortKey = assemble->dSortKey;
        AV.evalThr = P.evalue_thr; AV.minAlnLen = P.min_aln_len; AV.tmp = dTmp; AV.pass = dPass; AV.flags = dFlags;
        th = tb("align_assemble", (double) nRev * (sizeof(AlnRaw) + 2.0 * sizeof(mk_alignment)), 0);
        hipLaunchKernelGGL(assemble_kernel, dim3((nRev + 255) / 256), dim3(256), 0, stream, AV);
        hipLaunchKernelGGL(assemble_count_kernel, dim3((V.n_queries + 255) / 256), dim3(256), 0, stream, AV, dCnt);
        ACHK(hipGetLastError());
        if ((rc = device_scan(dCnt, V.n_queries, dOff, dOff + V.n_queries, stream, err)) != MK_OK) return rc;
        hipLaunchKernelGGL(assemble_rank_kernel, dim3((nRev + 255) / 256), dim3(256), 0, stream, AV, (const uint32_t *) dOff, dFinal);
        te(th);
        ACHK(hipGetLastError());
        ACHK(hipMemcpyAsync(hFlags, dFlags, 8, hipMemcpyDeviceToHost, stream));
        ACHK(hipMemcpyAsync(hWork, dWork, 2 * SW_NCFG * 8, hipMemcpyDeviceToHost, stream));
        ACHK(hipMemcpyAsync(hFlags + 2, dOff + V.n_queries, 4, hipMemcpyDeviceToHost, stream));
        ACHK(hipMemcpyAsync(assemble->counts, dCnt, (size_t) V.n_queries * 4, hipMemcpyDeviceToHost, stream));
        ACHK(hipMemcpyAsync(hCount, dCount, 12, hipMemcpyDeviceToHost, stream));
        ACHK(sync_wait(stream, "wait_align"));
        if (hCount[2] != 0) { err = "internal: the position pass disagrees with the score pass for " + std::to_string(hCount[2]) + " pairs"; return MK_ERR_DEVICE; }
        if (hFlags[1] != 0) { err = "Score of forward/backward SW differ for " + std::to_string(hFlags[1]) + " pairs"; return MK_ERR_SW_MISMATCH; }
        if (hFlags[0] == 0) {
            const size_t nFinal = hFlags[2];
            if (nFinal > 0) {
                mk_alignment *dst = assemble->reserve(nFinal);
                if (!dst) { err = "pinned host allocation failed"; return MK_ERR_DEVICE; }
                ACHK(hipMemcpyAsync(dst, dFinal, nFinal * sizeof(mk_alignment), hipMemcpyDeviceToHost, stream));
                ACHK(sync_wait(stream, "wait_align"));
            }
            assemble->nOut = nFinal;
            assembled = true;
        } else {
            std::fill(assemble->counts, assemble->counts + V.n_queries, 0u);       // a score beyond the e-value table: the caller assembles this range
        }
    }
    if (assemble) assemble->done = assembled;
    if (assembled) {
        unsigned long long *hWork = (unsigned long long *) pinned_scratch("asm_revwork_h", 2 * SW_NCFG * 8);
        set_range_work(hRev, hWork, ts);
        return MK_OK;
    }
    ACHK(hipMemcpyAsync(hRaw, dRaw, (size_t) nRev * sizeof(AlnRaw), hipMemcpyDeviceToHost, stream));
    ACHK(hipMemcpyAsync(hCount, dCount, 12, hipMemcpyDeviceToHost, stream));
    ACHK(sync_wait(stream, "wait_align"));
    if (hCount[2] != 0) { err = "internal: the position pass disagrees with the score pass for " + std::to_string(hCount[2]) + " pairs"; return MK_ERR_DEVICE; }
    *out = hRaw; *nOut = nRev;
    // reverse-pass work per tile configuration, from the results
    {
        double w[2 * SW_NCFG];
        for (int c = 0; c < 2 * SW_NCFG; c++) w[c] = 0;
#pragma omp parallel
        {
            double wl[2 * SW_NCFG];
            for (int c = 0; c < 2 * SW_NCFG; c++) wl[c] = 0;
#pragma omp for schedule(static) nowait
            for (uint32_t i = 0; i < nRev; i++) {
                const uint32_t ql = (uint32_t) hRaw[i].q_end + 1, tl = (uint32_t) hRaw[i].t_end + 1;
                const int c = sw_cfg_of(ql);
                wl[2 * c] += (double) tl + 2.0 * ql + sizeof(SwJob) + sizeof(SwOut);
                wl[2 * c + 1] += (double) ql * (double) tl;
            }
#pragma omp critical(mk_align_revwork)
            for (int c = 0; c < 2 * SW_NCFG; c++) w[c] += wl[c];
        }
        unsigned long long wi[2 * SW_NCFG];
        for (int c = 0; c < 2 * SW_NCFG; c++) wi[c] = (unsigned long long) w[c];
        set_range_work(hRev, wi, ts);
    }
    return MK_OK;
}

// ---- persistent scratch buffers ------------------------------------------------------------------
namespace {
struct Scratch { void *p = nullptr; size_t cap = 0; bool pinned = false; };
std::map<std::string, Scratch> &scratch_map() { static std::map<std::string, Scratch> m; return m; }
}

static std::mutex &scratch_mutex() { static std::mutex m; return m; }
static thread_local int t_lane = 0;
void set_scratch_lane(int lane) { t_lane = lane; }
int scratch_lane() { return t_lane; }
static std::string scratch_key(const char *prefix, const char *name) {
    std::string k = std::string(prefix) + name;
    if (t_lane > 0) { k += '#'; k += std::to_string(t_lane); }
    return k;
}

void *dev_scratch(const char *name, size_t bytes) {
    std::lock_guard<std::mutex> g(scratch_mutex());
    Scratch &s = scratch_map()[scratch_key("d:", name)];
    if (bytes <= s.cap && s.p) return s.p;
    if (s.p) (void) hipFree(s.p);
    s.p = nullptr; s.cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 8, 256);
    ScopedHost sh("host_alloc_device");                       // (what a first call pays for its buffers shows in the statistics)
    if (hipMalloc(&s.p, want) != hipSuccess) {                // the pool of the query batches may sit on what is missing
        (void) hipGetLastError();
        dev_pool_release();
        s.p = nullptr;
        if (hipMalloc(&s.p, want) != hipSuccess) { (void) hipGetLastError(); s.p = nullptr; return nullptr; }
    }
    s.cap = want;
    return s.p;
}

void *pinned_scratch(const char *name, size_t bytes) {
    std::lock_guard<std::mutex> g(scratch_mutex());
    Scratch &s = scratch_map()[scratch_key("h:", name)];
    if (bytes <= s.cap && s.p) return s.p;
    if (s.p) (void) hipHostFree(s.p);
    s.p = nullptr; s.cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 8, 256);
    ScopedHost sh("host_alloc_pinned");
    if (hipHostMalloc(&s.p, want, hipHostMallocDefault) != hipSuccess) { s.p = nullptr; return nullptr; }
    s.cap = want; s.pinned = true;
    return s.p;
}

namespace {
struct FreeBlock { void *p; size_t cap; };
std::vector<FreeBlock> &block_pool() { static std::vector<FreeBlock> v; return v; }
std::mutex &block_mutex() { static std::mutex m; return m; }
constexpr size_t BLOCK_POOL_MAX = 6;
}

bool HostBlock::reserve(size_t bytes, size_t keepBytes) {
    if (bytes <= cap && p) return true;
    void *np = nullptr; size_t ncap = 0;
    {
        std::lock_guard<std::mutex> g(block_mutex());
        auto &pool = block_pool();
        int best = -1;                               // smallest pooled block that is large enough
        for (int i = 0; i < (int) pool.size(); i++)
            if (pool[i].cap >= bytes && (best < 0 || pool[i].cap < pool[best].cap)) best = i;
        if (best >= 0) { np = pool[best].p; ncap = pool[best].cap; pool.erase(pool.begin() + best); }
    }
    if (!np) {
        ncap = std::max<size_t>(bytes + bytes / 8, 4096);
        ScopedHost sh("host_alloc_pinned");
        if (hipHostMalloc(&np, ncap, hipHostMallocDefault) != hipSuccess) return false;
    }
    if (p && keepBytes) std::memcpy(np, p, std::min(keepBytes, cap));
    release();
    p = np; cap = ncap;
    return true;
}

void HostBlock::release() {
    if (!p) return;
    void *drop = nullptr;
    {
        std::lock_guard<std::mutex> g(block_mutex());
        auto &pool = block_pool();
        pool.push_back(FreeBlock{p, cap});
        if (pool.size() > BLOCK_POOL_MAX) {          // keep the largest blocks
            int small = 0;
            for (int i = 1; i < (int) pool.size(); i++) if (pool[i].cap < pool[small].cap) small = i;
            drop = pool[small].p;
            pool.erase(pool.begin() + small);
        }
    }
    if (drop) (void) hipHostFree(drop);
    p = nullptr; cap = 0;
}

static std::atomic<uint64_t> g_scratchEpoch{1};
uint64_t scratch_epoch() { return g_scratchEpoch.load(); }

void scratch_release_all() {
    std::lock_guard<std::mutex> gs(scratch_mutex());
    g_scratchEpoch++;
    for (auto &kv : scratch_map()) {
        if (!kv.second.p) continue;
        if (kv.first[0] == 'h') (void) hipHostFree(kv.second.p); else (void) hipFree(kv.second.p);
        kv.second = Scratch();
    }
    std::lock_guard<std::mutex> g(block_mutex());
    for (FreeBlock &b : block_pool()) (void) hipHostFree(b.p);
    block_pool().clear();
}

}  // namespace mk
