// metaeuk_amd/csrc/mk_kernels.hpp -- device-side data structures and launch wrappers shared by the
// HIP kernels (mk_sw.hip, mk_align.hip, mk_prefilter.hip) and the C-ABI glue (mk_abi.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mk {

// One Smith-Waterman DP: rows = query residues q_res[q_start + k*q_step], k < q_len (with their int8
// composition bias), columns = target residues t_res[t_start + c*t_step], c < t_len.
// `slot` = where the result goes (results are scattered back to the caller's pair order).
struct SwJob {
    uint64_t t_start;
    uint32_t q_start;
    uint32_t q_len;
    uint32_t t_len;
    int32_t q_step;      // +1 forward, -1 reverse pass
    int32_t t_step;
    uint32_t slot;
};
// result: score, first column (in pass order) where the maximum is reached, smallest row with the maximum there
struct SwOut { int32_t score; int32_t end_col; int32_t end_row; int32_t pad; };

struct SwLaunch {
    const uint8_t *q_res; const int8_t *q_bias8;
    const int8_t *q_prof = nullptr;   // profile queries: [column][32] alignment profile (mk_profile.hpp); the scores then come from it, not from mat + bias
    const uint8_t *t_res;
    const int8_t *mat;          // 21x21 int8 substitution scores, mat[t*21+q]
    const SwJob *jobs; SwOut *out; uint64_t n_jobs;
    const uint32_t *order;      // optional indirection: the kernel's i-th DP is jobs[order[i]]
    uint32_t *boundary;         // multi-tile scratch: per job boundary_stride entries (null when every job is single tile)
    uint32_t boundary_stride;
    int gap_open, gap_extend;
    // shared-query mode (forward pass of the pipeline): jobs are ordered by query; wave w runs the jobs
    // [wave_start[w], wave_start[w+1]) -- all of one query, at most 64/G of them -- on ONE LDS query profile.
    // null: every job builds its own profile (test path, reverse pass).
    const uint32_t *wave_start; uint64_t n_waves;
    uint32_t *work_counter;     // shared-query mode: zeroed device counter the persistent workgroups pull wave numbers from
    uint32_t *work_counter_t = nullptr;   // ... and the transposed score kernels' (profile queries: launch_sw_score runs them over the same waves)
    uint32_t t_max_rows = 0;    // > 0: waves whose longest target has at most this many residues take the transposed score kernels (mk_sw.hip: swt_kernel)
    uint32_t persistent_blocks; // ... and how many workgroups to launch (0: one per wave)
    uint32_t units_per_block;   // a workgroup retires after this many waves of jobs (0: runs until the counter is exhausted)
    uint64_t boundary_job0;     // job index that owns the first boundary_stride entries of `boundary`
    const int32_t *known_score; // position / reverse pass: the maximum score of job j is known_score[jobs[j].slot] (launch_sw_known needs it; launch_sw, per-job
                                // profiles, stops a DP once that score has been seen in a finished column; null: unknown)
    bool narrow = false;        // score pass: the 16-lane variant of the 384-row tile (the waves were cut with sw_cfg_jobs_per_wave(c, true))
};

// tile configurations: G lanes per DP x R rows per lane; a job uses the smallest one whose G*R >= q_len
// (the last one also handles longer queries in row tiles)
constexpr int SW_NCFG = 11;
__host__ __device__ inline int sw_cfg_rows(int c) {
    const int rows[SW_NCFG] = {32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024};
    return rows[c];
}
// forward pass of the pipeline: tiles of at most 768 rows run in packed int16, two targets per lane group (16 lanes up to 256
// rows: 8 jobs per wave; 32 lanes: 4 jobs per wave); larger tiles in int32 on 64 lanes (1 job per wave)
__host__ __device__ inline bool sw_cfg_packed(int c) { return sw_cfg_rows(c) <= 768; }
// narrow: the score pass of the 384-row tile on 16 lanes x 24 rows (8 jobs per wave) instead of 32 lanes x 12 rows -- fewer ramp steps per
// DP when the targets are short (profile queries against ORF fragments, ~40 columns: 87 -> 65 ms at config-4 scale; the same for the
// 512-row tile, 32 rows per lane and 133 VGPRs, was slower: 166 -> 225 ms)
__host__ __device__ inline uint32_t sw_cfg_jobs_per_wave(int c, bool narrow = false) {
    const int rows = sw_cfg_rows(c);
    return rows <= 256 ? 8u : (rows <= 768 ? (narrow ? 8u : 4u) : 1u);     // (round 6: every packed tile of a profile query cuts its waves at 8 jobs --
                                                                            //  the transposed unit of mk_sw.hip runs 8 on 16-lane groups, the 32-lane classic
                                                                            //  path takes such a wave in two rounds)
}
__host__ __device__ inline int sw_cfg_of(uint32_t qLen) {
    int c = 0;
    while (c < SW_NCFG - 1 && (uint32_t) sw_cfg_rows(c) < qLen) c++;
    return c;
}
hipError_t launch_sw(const SwLaunch &L, int cfg, hipStream_t stream);
// score-only forward pass in packed int16, two targets per lane group (sw_cfg_packed configurations, shared-query mode only)
hipError_t launch_sw_score(const SwLaunch &L, int cfg, hipStream_t stream);
// position of a KNOWN maximum (end cell: first column that reaches it, smallest row there) in packed int16, eight independent DPs per wave,
// for the tiles sw_cfg_known() names; L.order / L.n_jobs / L.work_counter / L.persistent_blocks as in the persistent launches
__host__ __device__ inline bool sw_cfg_known(int c) { return sw_cfg_rows(c) <= 64; }
hipError_t launch_sw_known(const SwLaunch &L, int cfg, hipStream_t stream);
// round 6: the position pass of profile queries, transposed (the fragment in the lanes' rows, every profile column walked, packed int16, the known score
// located by the tie rule row first); for forward jobs of at most SW_TPOS_MAX_ROWS target residues whose query is a profile of a 32-lane tile
constexpr uint32_t SW_TPOS_MAX_ROWS = 256;
hipError_t launch_sw_tpos(const SwLaunch &L, int rows, uint32_t blocksPerClass, hipStream_t stream);
// position / reverse pass over ALL tile configurations of a register class in one persistent launch, the bounds of the tile configurations
// read on the device (mk_sw.hip: sw_multi_kernel)
hipError_t launch_sw_multi(const SwLaunch &L, const uint32_t *bounds, uint32_t *counter, int cls, uint32_t blocks, hipStream_t stream, bool prio = true);

// Ungapped score of a (query, target, 16-bit diagonal) candidate: UngappedAlignment::scoreSingleSequence
// (M/src/prefiltering/UngappedAlignment.cpp:438-447).  Both sequences below 32768 residues: the diagonal is the signed 16-bit value.
// Otherwise the 16-bit diagonal is ambiguous and the reference takes the best of every real diagonal it can stand for
// (computeLongScore, :312-329): -d * 65536 + diagonal for d = 1 .. 1 + tLen / 32768, and d * 65536 + diagonal for d = 0 .. qLen / 65536.
// smat = 21 x 21 int8 scores [q * 21 + t]; corr = the query's int8 diagonal correction.  Same code on the device and on the host.
// score(query position, target residue) is the scorer: substitution matrix + int8 correction for a sequence query, the alignment profile
// for a profile query (UngappedAlignment::createProfile, :385-414: queryProfile[pos][aa], X column 0).
template <class ScoreFn>
__host__ __device__ inline int ungapped_on_diagonal_fn(ScoreFn score_of, uint32_t qLen, const uint8_t *t, uint32_t tLen,
                                                       int diagonal, uint32_t minDist) {      // computeSingelSequenceScores (:416-430)
    uint32_t len = 0, q0 = 0, t0 = 0;
    if (diagonal >= 0 && minDist < qLen) { len = tLen < qLen - minDist ? tLen : qLen - minDist; q0 = minDist; }
    else if (diagonal < 0 && minDist < tLen) { len = tLen - minDist < qLen ? tLen - minDist : qLen; t0 = minDist; }
    int score = 0, best = 0;
    for (uint32_t k = 0; k < len; k++) {
        const int curr = score_of(q0 + k, (uint32_t) t[t0 + k]);
        score = score + curr > 0 ? score + curr : 0;
        best = best > score ? best : score;
    }
    return best;
}
template <class ScoreFn>
__host__ __device__ inline int ungapped_score_fn(ScoreFn score_of, uint32_t qLen, const uint8_t *t, uint32_t tLen, uint32_t d16) {
    if (qLen >= 32768u || tLen >= 32768u) {
        int best = 0;
        for (uint32_t d = 1; d <= 1u + tLen / 32768u; d++) {
            const int real = (int) (0u - d * 65536u + d16);                   // unsigned wrap, then int: as the reference computes it
            const int m = ungapped_on_diagonal_fn(score_of, qLen, t, tLen, real, (uint32_t) (real < 0 ? -real : real));
            best = best > m ? best : m;
        }
        for (uint32_t d = 0; d <= qLen / 65536u; d++) {
            const int real = (int) (d * 65536u + d16);
            const int m = ungapped_on_diagonal_fn(score_of, qLen, t, tLen, real, (uint32_t) (real < 0 ? -real : real));
            best = best > m ? best : m;
        }
        return best;
    }
    const uint32_t dist = ((0x10000u - d16) & 0xFFFFu) < d16 ? ((0x10000u - d16) & 0xFFFFu) : d16;      // distanceFromDiagonal (:364-369)
    return ungapped_on_diagonal_fn(score_of, qLen, t, tLen, (int) (short) (uint16_t) d16, dist);
}
template <typename MatT>
__host__ __device__ inline int ungapped_score(const MatT *smat, const uint8_t *q, const int8_t *corr, uint32_t qLen, const uint8_t *t, uint32_t tLen, uint32_t d16) {
    return ungapped_score_fn([=](uint32_t i, uint32_t tr) -> int { return (int) (int8_t) ((int8_t) smat[q[i] * 21 + tr] + corr[i]); }, qLen, t, tLen, d16);
}
// profile query: aln = [column][32] alignment profile of the query (mk_profile.hpp)
__host__ __device__ inline int ungapped_score_profile(const int8_t *aln, uint32_t qLen, const uint8_t *t, uint32_t tLen, uint32_t d16) {
    return ungapped_score_fn([=](uint32_t i, uint32_t tr) -> int { return (int) aln[(size_t) i * 32 + tr]; }, qLen, t, tLen, d16);
}

struct UngappedJob { uint64_t t_start; uint32_t q_start; uint32_t q_len; uint32_t t_len; uint32_t diagonal; };
struct UngappedLaunch {
    const uint8_t *q_res; const int8_t *q_corr;
    const uint8_t *t_masked;
    const int8_t *mat;          // 21x21 int8: BLOSUM62 x2 (bias -0.2) scores, mat[q*21+t]
    const UngappedJob *jobs; int32_t *out; uint64_t n_jobs;
};
hipError_t launch_ungapped(const UngappedLaunch &L, hipStream_t stream);

// per-residue query-side inputs (k-mer thresholds, int8 diagonal correction, int8 SW composition bias), mk_derive.hip
struct SubMat;
hipError_t launch_derive(const uint8_t *dRes, const uint64_t *dOff, uint32_t nq, uint64_t total, const SubMat &kmerMat, const SubMat &alnMat,
                         int kmerThr, bool compBias, float scale, int16_t *dKthr, int8_t *dCorr, int8_t *dSw8, hipStream_t stream, int kmerSize = 6);

// wall-clock accounting of host-side phases (shows up in mk_kernel_stats with launches == 0)
void host_stat(const char *name, double ms);
hipError_t sync_wait(hipStream_t stream, const char *statName);   // hipStreamSynchronize, blocked time booked under statName
struct ScopedHost {
    const char *name; double t0;
    static double now_ms();
    explicit ScopedHost(const char *n) : name(n), t0(now_ms()) {}
    ~ScopedHost() { host_stat(name, now_ms() - t0); }
};

// persistent, growable device / pinned-host buffers (one set per process; no hipMalloc on the hot path)
void *dev_scratch(const char *name, size_t bytes);          // nullptr on allocation failure (after the idle query-batch blocks were given back and the request repeated)
void dev_pool_release();                                    // mk_abi.cpp: every idle block of the query batches' device pool back to the device
// device blocks that live as long as a batch (query residues, ORF fragments): from / back to that pool -- hipFree waits for the whole device and
// would serialise the batch in flight with the one being prepared.  *cap = what dev_block_free must be told
void *dev_block_alloc(size_t bytes, size_t *cap);
void dev_block_free(void *p, size_t cap);
void *pinned_scratch(const char *name, size_t bytes);
void scratch_release_all();
uint64_t scratch_epoch();                                   // changes whenever scratch_release_all has run (whoever remembers what a buffer holds compares it)
void set_scratch_lane(int lane);                            // of the calling thread: lane > 0 gets buffers of its own under the same names (a second worker of one stage)
int scratch_lane();

// Pinned host block holding a batch's results (prefilter hits / alignments).  Blocks come from a small pool and
// go back to it when the batch handle dies: the next batch writes into already-mapped, already-pinned pages
// (device DMA lands the results at their final place, no first-touch faults, no host-side append copy).
struct HostBlock {
    void *p = nullptr; size_t cap = 0;
    HostBlock() = default;
    HostBlock(const HostBlock &) = delete;
    HostBlock &operator=(const HostBlock &) = delete;
    ~HostBlock() { release(); }
    bool reserve(size_t bytes, size_t keepBytes);   // grow to >= bytes keeping [0, keepBytes); false when the allocation fails
    void release();
};

}  // namespace mk
