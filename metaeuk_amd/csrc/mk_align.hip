// metaeuk_amd/csrc/mk_align.hip -- the alignment stage as a device-resident pipeline.
// Per batch (or per prefilter chunk, under mk_search): the prefilter hits go up once (12 B per pair) and only the accepted
// pairs' integers come back (24 B each).
//   expand_pairs_kernel      pair -> query (binary search in the per-query offsets), forward SwJob + 64-bit sort key
//                            (tile configuration, query, target length descending)
//   hipcub radix sort        the jobs of a query become adjacent and similar in length
//   seg_mark / max-scan / wave_flag / select   cut every (configuration, query) segment into waves of jobs that share ONE
//                            LDS query profile
//   swp_kernel / sw_kernel   score pass (mk_sw.hip): packed int16, two targets per lane group, for tiles <= 256 rows;
//                            persistent launch (a fixed number of one-wave workgroups per CU pull waves from a counter)
//   gate_kernel              e-value gate on the score (table per query length) -> position jobs for the ~9 % survivors
//   sw_kernel                position pass (end cell of the maximum), then rev_jobs_kernel + reverse pass on the reversed
//                            prefixes (start cell)
//   collect_kernel           AlnRaw records in pair order
#include "mk_align.hpp"
#include "mk_kernels.hpp"
#include <hipcub/hipcub.hpp>
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

constexpr uint32_t KEY_CLS = 4096;
static_assert(SW_NCFG <= 16, "job sort keys keep 4 bits for the tile configuration");

__device__ __forceinline__ uint32_t sort_key(uint32_t qLen, uint32_t tLen) {
    return (uint32_t) sw_cfg_of(qLen) * KEY_CLS + (KEY_CLS - 1 - min(tLen >> 4, KEY_CLS - 1));
}

// forward jobs are ordered by (tile configuration, query, target length descending): the DPs of a wave share their query
// (one LDS profile per wave) and have similar numbers of columns
__global__ __launch_bounds__(256) void expand_pairs_kernel(AlignView V, const uint64_t *hitOff, const mk_hit *hits, uint64_t n,
                                                           SwJob *jobs, uint64_t *keys, uint32_t *idx, uint32_t *badTarget,
                                                           unsigned long long *work /* [2 * cfg]: bytes, [2 * cfg + 1]: cells of the forward pass (statistics) */) {
    __shared__ unsigned long long sWork[2 * SW_NCFG];
    for (int k = threadIdx.x; k < 2 * SW_NCFG; k += blockDim.x) sWork[k] = 0;
    __syncthreads();
    const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        uint32_t lo = 0, hi = V.n_queries;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (hitOff[mid] <= p) lo = mid; else hi = mid; }
        const uint32_t t = hits[p].seq_id;
        const uint32_t qLen = (uint32_t) (V.q_off[lo + 1] - V.q_off[lo]), tLen = t < V.n_targets ? (uint32_t) (V.t_off[t + 1] - V.t_off[t]) : 0u;
        const int c = sw_cfg_of(qLen);
        atomicAdd(&sWork[2 * c], (unsigned long long) (tLen + 2u * qLen + (uint32_t) (sizeof(SwJob) + sizeof(SwOut))));
        atomicAdd(&sWork[2 * c + 1], (unsigned long long) qLen * tLen);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * SW_NCFG; k += blockDim.x) if (sWork[k]) atomicAdd(&work[k], sWork[k]);
    if (p >= n) return;
    uint32_t lo = 0, hi = V.n_queries;              // largest q with hitOff[q] <= p
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (hitOff[mid] <= p) lo = mid; else hi = mid; }
    const uint32_t q = lo, t = hits[p].seq_id;
    SwJob j;
    j.q_start = (uint32_t) V.q_off[q]; j.q_len = (uint32_t) (V.q_off[q + 1] - V.q_off[q]);
    if (t < V.n_targets) { j.t_start = V.t_off[t]; j.t_len = (uint32_t) (V.t_off[t + 1] - V.t_off[t]); }
    else { j.t_start = 0; j.t_len = 0; atomicMax(badTarget, (uint32_t) p + 1u); }     // reported to the caller, nothing is aligned
    j.q_step = 1; j.t_step = 1; j.slot = (uint32_t) p;
    jobs[p] = j;
    keys[p] = ((uint64_t) sw_cfg_of(j.q_len) << 44) | ((uint64_t) q << 12) | (uint64_t) (KEY_CLS - 1 - min(j.t_len >> 4, KEY_CLS - 1));
    idx[p] = (uint32_t) p;
}

// segment = run of jobs with the same (configuration, query); value = own index at a segment head, 0 elsewhere (max-scan -> head)
__global__ __launch_bounds__(256) void seg_mark_kernel(const uint64_t *sortedKeys, uint32_t n, uint32_t *head) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i > 0 && (sortedKeys[i] >> 12) != (sortedKeys[i - 1] >> 12)) ? i : 0u;
}
// a wave starts at every (64/G)-th job of a segment
__global__ __launch_bounds__(256) void wave_flag_kernel(const uint64_t *sortedKeys, const uint32_t *head, uint32_t n, uint8_t *flag, bool narrow) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cfg = (int) (sortedKeys[i] >> 44);
    const uint32_t dpw = sw_cfg_jobs_per_wave(cfg, narrow);        // packed score kernel: 2 per 16-lane group; else 64 / G
    flag[i] = ((i - head[i]) % dpw) == 0 ? 1 : 0;
}
// job and wave ranges of every configuration; closes the wave list with n
__global__ void shared_bounds_kernel(const uint64_t *sortedKeys, uint32_t n, uint32_t *waveStart, const uint32_t *nWaves,
                                     uint32_t *out /* [0..SW_NCFG] job bounds, [16..16+SW_NCFG] wave bounds, [32..] first target-length class */) {
    const uint32_t c = threadIdx.x;
    const uint32_t nw = nWaves[0];
    if (c == 0) waveStart[nw] = n;
    if (c > SW_NCFG) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t) (sortedKeys[mid] >> 44) < c) lo = mid + 1; else hi = mid; }
    out[c] = lo;
    const uint32_t jb = lo;
    if (c < SW_NCFG) out[32 + c] = jb < n ? (uint32_t) (sortedKeys[jb] & (KEY_CLS - 1)) : 0u;
    lo = 0; hi = nw;                                                 // first wave starting at or after the job bound
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (waveStart[mid] < jb) lo = mid + 1; else hi = mid; }
    out[16 + c] = lo;
}

// first index whose key >= c*KEY_CLS, for c = 0..SW_NCFG; plus the key at that index
__global__ void bounds_kernel(const uint32_t *sortedKeys, uint32_t n, uint32_t *bounds /* SW_NCFG+1 */, uint32_t *firstKey /* SW_NCFG */) {
    const uint32_t c = threadIdx.x;
    if (c > SW_NCFG) return;
    const uint32_t want = c * KEY_CLS;
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sortedKeys[mid] < want) lo = mid + 1; else hi = mid; }
    bounds[c] = lo;
    if (c < SW_NCFG) firstKey[c] = lo < n ? sortedKeys[lo] : 0;
}

// e-value gate on the forward score (table per query length); survivors get a position job (the same DP again, this time
// with end-position tracking)
__global__ __launch_bounds__(256) void gate_kernel(const SwJob *fwdJobs, const SwOut *fwdOut, uint64_t n, const GateEntry *gate,
                                                   uint32_t *posCount, uint32_t *posPair, SwJob *posJobs, uint32_t *posKeys, uint32_t *posIdx, int32_t *posScore,
                                                   unsigned long long *work /* [2 * cfg]: bytes, [2 * cfg + 1]: cells of the position pass (statistics) */) {
    __shared__ unsigned long long sWork[2 * SW_NCFG];
    for (int k = threadIdx.x; k < 2 * SW_NCFG; k += blockDim.x) sWork[k] = 0;
    __syncthreads();
    const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    bool pass = false;
    SwJob j;
    if (p < n) {
        const int score = fwdOut[p].score;
        if (score > 0) {
            j = fwdJobs[p];
            const GateEntry g = gate[j.q_len];
            pass = score >= g.s0;
            if (!pass && score < 256) pass = (g.mask[score >> 5] >> (score & 31)) & 1u;
            if (pass) {
                const int c = sw_cfg_of(j.q_len);
                atomicAdd(&sWork[2 * c], (unsigned long long) (j.t_len + 2u * j.q_len + (uint32_t) (sizeof(SwJob) + sizeof(SwOut))));
                atomicAdd(&sWork[2 * c + 1], (unsigned long long) j.q_len * j.t_len);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * SW_NCFG; k += blockDim.x) if (sWork[k]) atomicAdd(&work[k], sWork[k]);
    if (!pass) return;
    const uint32_t r = atomicAdd(posCount, 1u);
    posPair[r] = (uint32_t) p;
    posScore[r] = fwdOut[p].score;
    j.slot = r;
    posJobs[r] = j;
    posKeys[r] = sort_key(j.q_len, j.t_len);
    posIdx[r] = r;
}

// reverse jobs (reversed prefixes ending at the forward end cell) of the survivors; the position pass must reproduce the
// score the gate saw
__global__ __launch_bounds__(256) void rev_jobs_kernel(const SwJob *posJobs, const SwOut *posOut, const uint32_t *posPair, const SwOut *fwdOut, uint32_t n,
                                                       SwJob *revJobs, uint32_t *revKeys, uint32_t *revIdx, uint32_t *mismatch) {
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
    revKeys[r] = sort_key(j.q_len, j.t_len);
    revIdx[r] = r;
}

__global__ void iota_kernel(uint32_t *p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

__global__ __launch_bounds__(256) void collect_kernel(const uint32_t *sortedPair, const uint32_t *sortedRev, uint32_t n,
                                                      const SwOut *fwdOut, const SwOut *revOut, AlnRaw *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = sortedPair[i], r = sortedRev[i];
    const SwOut f = fwdOut[r], b = revOut[r];          // both indexed by survivor number
    AlnRaw a;
    a.pair = p; a.score = f.score; a.q_end = f.end_row; a.t_end = f.end_col;
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

static int run_sorted_sw(const AlignView &V, const mk_params &P, const SwJob *jobs, SwOut *out, uint32_t *keys, uint32_t *idx,
                         uint32_t *keys2, uint32_t *idx2, uint32_t n, const char *tag, hipStream_t stream, std::string &err,
                         timed_begin_fn tb, timed_end_fn te, int *handles /* SW_NCFG, may be null */,
                         const int32_t *knownScore /* device, by job slot: the maximum every job will reach (null: unknown) */) {
    if (handles) for (int c = 0; c < SW_NCFG; c++) handles[c] = -1;
    if (n == 0) return MK_OK;
    hipcub::DoubleBuffer<uint32_t> kb(keys, keys2), vb(idx, idx2);
    size_t tempBytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, kb, vb, (int) n, 0, 16, stream);
    void *temp = dev_scratch("align_sort_temp", tempBytes);
    ANULL(temp);
    int th = tb("align_sort", 16.0 * n, 0);
    ACHK(hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, kb, vb, (int) n, 0, 16, stream));
    te(th);
    uint32_t *dBounds = (uint32_t *) dev_scratch("align_bounds", 64 * sizeof(uint32_t));
    ANULL(dBounds);
    hipLaunchKernelGGL(bounds_kernel, dim3(1), dim3(64), 0, stream, kb.Current(), n, dBounds, dBounds + 32);
    ACHK(hipGetLastError());
    uint32_t *hb = (uint32_t *) pinned_scratch("align_bounds_h", 64 * sizeof(uint32_t));
    ANULL(hb);
    ACHK(hipMemcpyAsync(hb, dBounds, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    ACHK(sync_wait(stream, "wait_align"));
    for (int c = 0; c < SW_NCFG; c++) {
        const uint32_t lo = hb[c], hi = hb[c + 1];
        if (hi <= lo) continue;
        SwLaunch L;
        L.q_res = V.q_res; L.q_bias8 = V.q_bias8; L.q_prof = V.q_prof; L.t_res = V.t_res; L.mat = V.mat_aln;
        L.jobs = jobs; L.out = out; L.n_jobs = hi - lo; L.order = vb.Current() + lo;
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
        static const int force = getenv("MK_SW_KNOWN") ? atoi(getenv("MK_SW_KNOWN")) : -1;
        if (knownScore && sw_cfg_known(c) && !V.q_prof && (force >= 0 ? force != 0 : !V.co_resident)) {
            static int cus = 0;
            if (!cus) { int dev = 0; (void) hipGetDevice(&dev); if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256; }
            uint32_t *dWork = (uint32_t *) dev_scratch("align_knowncounters", 64 * sizeof(uint32_t));
            ANULL(dWork);
            static uint32_t slot = 0;
            uint32_t *counter = dWork + (slot++ % 64);
            ACHK(hipMemsetAsync(counter, 0, sizeof(uint32_t), stream));
            L.known_score = knownScore; L.work_counter = counter;
            // 11 / 22.5 KB of profiles per wave: few waves per CU, the LDS is shared with the prefilter workgroups of the other stream
            static const int perCuSmall = getenv("MK_SW_KNOWN_WAVES") ? std::max(1, atoi(getenv("MK_SW_KNOWN_WAVES"))) : 12;
            L.persistent_blocks = (uint32_t) cus * (uint32_t) (sw_cfg_rows(c) <= 32 ? perCuSmall : std::max(1, perCuSmall / 2));
            ACHK(launch_sw_known(L, c, stream));
        } else {
            ACHK(launch_sw(L, c, stream));
        }
        te(th);
    }
    return MK_OK;
}

// forward pass: jobs sorted by (configuration, query, target length); one wave per (query, <= 64/G targets)
static int run_shared_fwd(const AlignView &V, const mk_params &P, const SwJob *jobs, SwOut *out, uint64_t *keys, uint32_t *idx,
                          uint64_t *keys2, uint32_t *idx2, uint32_t n, hipStream_t stream, std::string &err,
                          timed_begin_fn tb, timed_end_fn te, int *handles /* SW_NCFG */) {
    for (int c = 0; c < SW_NCFG; c++) handles[c] = -1;
    if (n == 0) return MK_OK;
    hipcub::DoubleBuffer<uint64_t> kb(keys, keys2);
    hipcub::DoubleBuffer<uint32_t> vb(idx, idx2);
    uint32_t *dHead = (uint32_t *) dev_scratch("align_seghead", (size_t) n * 4);
    uint8_t *dFlag = (uint8_t *) dev_scratch("align_waveflag", n);
    uint32_t *dWave = (uint32_t *) dev_scratch("align_wavestart", ((size_t) n + 1) * 4);
    uint32_t *dNum = (uint32_t *) dev_scratch("align_nwaves", 16);
    uint32_t *dBounds = (uint32_t *) dev_scratch("align_bounds", 64 * sizeof(uint32_t));
    uint32_t *hb = (uint32_t *) pinned_scratch("align_bounds_h", 64 * sizeof(uint32_t));
    uint32_t *dWork = (uint32_t *) dev_scratch("align_workcounters", 64);
    ANULL(dHead); ANULL(dFlag); ANULL(dWave); ANULL(dNum); ANULL(dBounds); ANULL(hb); ANULL(dWork);
    ACHK(hipMemsetAsync(dWork, 0, 64, stream));
    // persistent forward launch: this many one-wave workgroups per CU and tile configuration (MK_SW_WAVES_PER_CU = one number
    // or one per tile configuration, comma separated).  Half the wave slots for the small tiles; fewer for the tiles whose profiles are large, so
    // that the LDS-hungry prefilter workgroups of the other stream still find room on the CU.
    static uint32_t persistentBlocks[SW_NCFG] = {};
    static uint32_t unitsPerBlock = 0;              // MK_SW_UNITS_PER_BLOCK: short-lived workgroups instead of the persistent launch
    if (!persistentBlocks[0]) {
        int dev = 0, cus = 256;
        (void) hipGetDevice(&dev);
        (void) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        int perCu[SW_NCFG] = {12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12};     // (round 2: 16 for the small tiles, 6-8 for the large ones; profiles/r03_search_tuning.txt)
        if (const char *e = getenv("MK_SW_WAVES_PER_CU")) {
            int k = 0, last = 16;
            for (const char *p = e; *p && k < SW_NCFG; k++) {
                last = std::max(1, atoi(p));
                perCu[k] = last;
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
            for (; k < SW_NCFG; k++) perCu[k] = last;
        }
        for (int c = 0; c < SW_NCFG; c++) persistentBlocks[c] = (uint32_t) (cus * perCu[c]);
        if (const char *e = getenv("MK_SW_UNITS_PER_BLOCK")) unitsPerBlock = (uint32_t) std::max(0, atoi(e));
    }
    size_t t1 = 0, t2 = 0, t3 = 0;
    hipcub::CountingInputIterator<uint32_t> iota(0);
    hipcub::DeviceRadixSort::SortPairs(nullptr, t1, kb, vb, (int) n, 0, 48, stream);
    hipcub::DeviceScan::InclusiveScan(nullptr, t2, dHead, dHead, hipcub::Max(), (int) n, stream);
    hipcub::DeviceSelect::Flagged(nullptr, t3, iota, dFlag, dWave, dNum, (int) n, stream);
    void *temp = dev_scratch("align_sort_temp", std::max(t1, std::max(t2, t3)));
    ANULL(temp);
    int th = tb("align_sort", 6.0 * 24.0 * n, 0);
    ACHK(hipcub::DeviceRadixSort::SortPairs(temp, t1, kb, vb, (int) n, 0, 48, stream));
    te(th);
    th = tb("align_waves", 30.0 * n, 0);
    hipLaunchKernelGGL(seg_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, kb.Current(), n, dHead);
    ACHK(hipcub::DeviceScan::InclusiveScan(temp, t2, dHead, dHead, hipcub::Max(), (int) n, stream));
    static const int narrowEnv = getenv("MK_SW_NARROW") ? atoi(getenv("MK_SW_NARROW")) : -1;
    const bool narrow = narrowEnv >= 0 ? narrowEnv != 0 : V.q_prof != nullptr;      // profile queries meet short targets (ORF fragments)
    hipLaunchKernelGGL(wave_flag_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, kb.Current(), dHead, n, dFlag, narrow);
    ACHK(hipcub::DeviceSelect::Flagged(temp, t3, iota, dFlag, dWave, dNum, (int) n, stream));
    hipLaunchKernelGGL(shared_bounds_kernel, dim3(1), dim3(64), 0, stream, kb.Current(), n, dWave, dNum, dBounds);
    te(th);
    ACHK(hipGetLastError());
    ACHK(hipMemcpyAsync(hb, dBounds, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    ACHK(sync_wait(stream, "wait_align"));
    for (int c = 0; c < SW_NCFG; c++) {
        const uint32_t lo = hb[c], hi = hb[c + 1], wlo = hb[16 + c], whi = hb[16 + c + 1];
        if (hi <= lo || whi <= wlo) continue;
        SwLaunch L;
        L.q_res = V.q_res; L.q_bias8 = V.q_bias8; L.q_prof = V.q_prof; L.t_res = V.t_res; L.mat = V.mat_aln;
        L.jobs = jobs; L.out = out; L.n_jobs = hi - lo; L.order = vb.Current();      // wave_start holds absolute sorted positions
        L.boundary = nullptr; L.boundary_stride = 0; L.boundary_job0 = lo;
        L.wave_start = dWave + wlo; L.n_waves = whi - wlo;
        L.work_counter = dWork + c; L.persistent_blocks = persistentBlocks[c]; L.units_per_block = unitsPerBlock;
        L.gap_open = P.gap_open; L.gap_extend = P.gap_extend;
        L.narrow = narrow;
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
    uint64_t *dHitOff = (uint64_t *) dev_scratch("align_hitoff", ((size_t) V.n_queries + 1) * sizeof(uint64_t));
    mk_hit *dHits = (mk_hit *) dev_scratch("align_hits", (size_t) n * sizeof(mk_hit));
    SwJob *dJobs = (SwJob *) dev_scratch("align_jobs", (size_t) n * sizeof(SwJob));
    SwOut *dOut = (SwOut *) dev_scratch("align_out", (size_t) n * sizeof(SwOut));
    uint32_t *dKeys = (uint32_t *) dev_scratch("align_keys", (size_t) n * 4), *dKeys2 = (uint32_t *) dev_scratch("align_keys2", (size_t) n * 4);
    uint32_t *dIdx = (uint32_t *) dev_scratch("align_idx", (size_t) n * 4), *dIdx2 = (uint32_t *) dev_scratch("align_idx2", (size_t) n * 4);
    GateEntry *dGate = (GateEntry *) dev_scratch("align_gate", gate.size() * sizeof(GateEntry));
    uint32_t *dCount = (uint32_t *) dev_scratch("align_count", 16);
    ANULL(dHitOff); ANULL(dHits); ANULL(dJobs); ANULL(dOut); ANULL(dKeys); ANULL(dKeys2); ANULL(dIdx); ANULL(dIdx2); ANULL(dGate); ANULL(dCount);
    ACHK(hipMemcpyAsync(dHitOff, hitOffHost, ((size_t) V.n_queries + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    ACHK(hipMemcpyAsync(dHits, hitsHost, (size_t) n * sizeof(mk_hit), hipMemcpyHostToDevice, stream));
    ACHK(hipMemcpyAsync(dGate, gate.data(), gate.size() * sizeof(GateEntry), hipMemcpyHostToDevice, stream));
    ACHK(hipMemsetAsync(dOut, 0, (size_t) n * sizeof(SwOut), stream));
    ACHK(hipMemsetAsync(dCount, 0, 16, stream));
    uint64_t *dKeys64 = (uint64_t *) dev_scratch("align_keys64", (size_t) n * 8), *dKeys64b = (uint64_t *) dev_scratch("align_keys64b", (size_t) n * 8);
    ANULL(dKeys64); ANULL(dKeys64b);
    unsigned long long *dFwdWork = (unsigned long long *) dev_scratch("align_fwdwork", 2 * SW_NCFG * 8);
    unsigned long long *hFwdWork = (unsigned long long *) pinned_scratch("align_fwdwork_h", 2 * SW_NCFG * 8);
    ANULL(dFwdWork); ANULL(hFwdWork);
    ACHK(hipMemsetAsync(dFwdWork, 0, 2 * SW_NCFG * 8, stream));
    int th = tb("align_expand", 52.0 * n, 0);
    hipLaunchKernelGGL(expand_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, V, dHitOff, dHits, (uint64_t) n, dJobs, dKeys64, dIdx, dCount + 1, dFwdWork);
    te(th);
    ACHK(hipGetLastError());
    ACHK(hipMemcpyAsync(hFwdWork, dFwdWork, 2 * SW_NCFG * 8, hipMemcpyDeviceToHost, stream));     // (lands before run_shared_fwd's synchronisation)
    int hFwd[SW_NCFG], hRev[SW_NCFG];
    int rc = run_shared_fwd(V, P, dJobs, dOut, dKeys64, dIdx, dKeys64b, dIdx2, n, stream, err, tb, te, hFwd);
    (void) fwdWork;
    for (int c = 0; c < SW_NCFG; c++) if (hFwd[c] >= 0) ts(hFwd[c], (double) hFwdWork[2 * c], (double) hFwdWork[2 * c + 1]);
    if (rc != MK_OK) return rc;
    // e-value gate on the forward scores; the survivors (at most n) get a position pass, then the reverse pass
    uint32_t *dRevPair = (uint32_t *) dev_scratch("align_revpair", (size_t) n * 4);
    SwJob *dPosJobs = (SwJob *) dev_scratch("align_posjobs", (size_t) n * sizeof(SwJob));
    ANULL(dRevPair); ANULL(dPosJobs);
    th = tb("align_gate", 52.0 * n, 0);
    unsigned long long *dPosWork = (unsigned long long *) dev_scratch("align_poswork", 2 * SW_NCFG * 8);
    unsigned long long *hPosWork = (unsigned long long *) pinned_scratch("align_poswork_h", 2 * SW_NCFG * 8);
    ANULL(dPosWork); ANULL(hPosWork);
    ACHK(hipMemsetAsync(dPosWork, 0, 2 * SW_NCFG * 8, stream));
    int32_t *dPosScore = (int32_t *) dev_scratch("align_posscore", (size_t) n * 4);
    ANULL(dPosScore);
    hipLaunchKernelGGL(gate_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dJobs, dOut, (uint64_t) n, dGate, dCount, dRevPair, dPosJobs, dKeys, dIdx, dPosScore, dPosWork);
    te(th);
    ACHK(hipGetLastError());
    uint32_t *hCount = (uint32_t *) pinned_scratch("align_count_h", 16);
    ANULL(hCount);
    ACHK(hipMemcpyAsync(hPosWork, dPosWork, 2 * SW_NCFG * 8, hipMemcpyDeviceToHost, stream));
    ACHK(hipMemcpyAsync(hCount, dCount, 8, hipMemcpyDeviceToHost, stream));
    ACHK(sync_wait(stream, "wait_align"));
    if (hCount[1] != 0) { err = "prefilter hit " + std::to_string(hCount[1] - 1) + " names a target outside the DB"; return MK_ERR_ARG; }
    const uint32_t nRev = hCount[0];
    if (nRev == 0) return MK_OK;
    SwOut *dPosOut = (SwOut *) dev_scratch("align_posout", (size_t) nRev * sizeof(SwOut));
    SwOut *dRevOut = (SwOut *) dev_scratch("align_revout", (size_t) nRev * sizeof(SwOut));
    SwJob *dRevJobs = (SwJob *) dev_scratch("align_revjobs", (size_t) nRev * sizeof(SwJob));
    ANULL(dPosOut); ANULL(dRevOut); ANULL(dRevJobs);
    ACHK(hipMemsetAsync(dPosOut, 0, (size_t) nRev * sizeof(SwOut), stream));
    ACHK(hipMemsetAsync(dRevOut, 0, (size_t) nRev * sizeof(SwOut), stream));
    int hPos[SW_NCFG];
    rc = run_sorted_sw(V, P, dPosJobs, dPosOut, dKeys, dIdx, dKeys2, dIdx2, nRev, "sw_pos", stream, err, tb, te, hPos, dPosScore);
    if (rc != MK_OK) return rc;
    for (int c = 0; c < SW_NCFG; c++) if (hPos[c] >= 0) ts(hPos[c], (double) hPosWork[2 * c], (double) hPosWork[2 * c + 1]);
    hipLaunchKernelGGL(rev_jobs_kernel, dim3((nRev + 255) / 256), dim3(256), 0, stream, dPosJobs, dPosOut, dRevPair, dOut, nRev, dRevJobs, dKeys, dIdx, dCount + 2);
    ACHK(hipGetLastError());
    rc = run_sorted_sw(V, P, dRevJobs, dRevOut, dKeys, dIdx, dKeys2, dIdx2, nRev, "sw_rev", stream, err, tb, te, hRev, dPosScore);    // (rev job r = survivor r: same score)
    if (rc != MK_OK) return rc;
    // order the survivors by pair index and collect
    uint32_t *dSeq = (uint32_t *) dev_scratch("align_seq", (size_t) nRev * 4), *dSeq2 = (uint32_t *) dev_scratch("align_seq2", (size_t) nRev * 4);
    uint32_t *dPair2 = (uint32_t *) dev_scratch("align_revpair2", (size_t) nRev * 4);
    AlnRaw *dRaw = (AlnRaw *) dev_scratch("align_raw", (size_t) nRev * sizeof(AlnRaw));
    ANULL(dSeq); ANULL(dSeq2); ANULL(dPair2); ANULL(dRaw);
    hipLaunchKernelGGL(iota_kernel, dim3((nRev + 255) / 256), dim3(256), 0, stream, dSeq, nRev);
    ACHK(hipGetLastError());
    hipcub::DoubleBuffer<uint32_t> pb(dRevPair, dPair2), sb(dSeq, dSeq2);
    size_t tempBytes = 0;
    int bits = 1; while ((1ull << bits) < nPairs) bits++;
    hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, pb, sb, (int) nRev, 0, bits, stream);
    void *temp = dev_scratch("align_sort_temp", tempBytes);
    ANULL(temp);
    th = tb("align_sort", 16.0 * nRev, 0);
    ACHK(hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, pb, sb, (int) nRev, 0, bits, stream));
    te(th);
    th = tb("align_collect", 64.0 * nRev, 0);
    hipLaunchKernelGGL(collect_kernel, dim3((nRev + 255) / 256), dim3(256), 0, stream, pb.Current(), sb.Current(), nRev, dPosOut, dRevOut, dRaw);
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
        // the tables belong to the batch (same for every range of an mk_search): uploaded when they change
        static thread_local uint64_t uploadedId = 0; static std::mutex upMutex;     // (per host thread: a second worker of the stage has buffers of its own)
        {
            std::lock_guard<std::mutex> g(upMutex);
            if (uploadedId != T.id) {
                ACHK(hipMemcpyAsync(dEval, T.evalue.data(), T.evalue.size() * sizeof(double), hipMemcpyHostToDevice, stream));
                ACHK(hipMemcpyAsync(dLenIdx, T.lenIdx.data(), T.lenIdx.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
                ACHK(hipMemcpyAsync(dBit, T.bitScore.data(), T.bitScore.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
                uploadedId = T.id;
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
        size_t tS = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tS, dCnt, dOff, (int) V.n_queries + 1, stream);
        void *tempS = dev_scratch("align_sort_temp", tS);
        ANULL(tempS);
        ACHK(hipcub::DeviceScan::ExclusiveSum(tempS, tS, dCnt, dOff, (int) V.n_queries + 1, stream));
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
        for (int c = 0; c < SW_NCFG; c++) if (hRev[c] >= 0) ts(hRev[c], (double) hWork[2 * c], (double) hWork[2 * c + 1]);
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
        for (int c = 0; c < SW_NCFG; c++) if (hRev[c] >= 0) ts(hRev[c], w[2 * c], w[2 * c + 1]);
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
    if (hipMalloc(&s.p, want) != hipSuccess) { s.p = nullptr; return nullptr; }
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

void scratch_release_all() {
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
