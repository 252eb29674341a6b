// metaeuk_amd/csrc/mk_prefilter.hip -- the k-mer prefilter on gfx950.
// Replaces, for a whole batch of queries at once:
//   KmerGenerator::generateKmerList / calculateArrayProduct   M/src/prefiltering/KmerGenerator.cpp:107-216
//   QueryMatcher::match (index probes + gather)               M/src/prefiltering/QueryMatcher.cpp:213-346
//   CacheFriendlyOperations::findDuplicates (double hits)     M/src/prefiltering/CacheFriendlyOperations.cpp:185-274
//   UngappedAlignment::computeScores                          M/src/prefiltering/UngappedAlignment.cpp:331-362
//   keepMaxScoreElementOnly / threshold / getResult / sort    M/src/prefiltering/QueryMatcher.cpp:149-209
// Pipeline per chunk of queries -- everything stays in HBM, the host sees only the final hit lists:
//   1. probe_kernel<COUNT>   one wave per k-mer start: enumerate the similar k-mers (two sorted 3-mer rows,
//                            product order of the reference), read the index offset pair of each, sum list sizes
//   2. exclusive scan        (hipcub) -> canonical position of every index entry ("ordinal")
//   3. probe_kernel<GATHER>  same enumeration, copies the index lists: key = (query, target), value = (ordinal, diagonal)
//   4. stable radix sort     (hipcub) by key: per (query,target) the hits are now in the reference's arrival order
//   5. double_hit_flag       the sequential 8-bit-diagonal rule of findDuplicates as a neighbour test + short backward
//                            walk; survivors are compacted IN ORDER (hipcub select), i.e. sorted by (query,target,arrival)
//   6. cand_score_kernel     exact ungapped score of every surviving (query, target, diagonal)
//   7. keep_kernel           best diagonal per target (first maximum in arrival order), >= --min-ungapped-score,
//                            per-query counts; queries that reach --max-seqs are flagged for the exact host
//                            tie-order logic (mk::select_hits), everything else is final
//   8. radix sort + emit     hits ordered by (query, score desc, target asc) -> compact mk_hit array + counts
#include "mk_prefilter.hpp"
#include "mk_host.hpp"
#include "mk_kernels.hpp"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstring>

namespace mk {

namespace {

constexpr int WAVE = 64;
constexpr int ROWCACHE = 512;     // leading entries of the second 3-mer row staged in LDS per wave
constexpr int N3 = 8000;

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t y = (uint32_t) __shfl_up((int) x, d, WAVE);
        if ((int) (threadIdx.x & (WAVE - 1)) >= d) x += y;
    }
    total = (uint32_t) __shfl((int) x, WAVE - 1, WAVE);
    return x - v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) v += (uint32_t) __shfl_xor((int) v, d, WAVE);
    return v;
}

// number of leading entries of a descending int16 row that are >= cutoff
__device__ __forceinline__ int count_ge(const int16_t *lds, int nLds, const int16_t *row, int cutoff) {
    if (nLds > 0 && (int) lds[nLds - 1] < cutoff) {
        int lo = 0, hi = nLds;              // first index with value < cutoff
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int) lds[mid] >= cutoff) lo = mid + 1; else hi = mid; }
        return lo;
    }
    int lo = nLds, hi = N3;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int) row[mid] >= cutoff) lo = mid + 1; else hi = mid; }
    return lo;
}

// query index of global residue position p (largest q with q_off[q] <= p)
__device__ __forceinline__ uint32_t find_query(const uint64_t *qOff, uint32_t nq, uint64_t p) {
    uint32_t lo = 0, hi = nq;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (qOff[mid] <= p) lo = mid; else hi = mid; }
    return lo;
}

struct ProbeArgs {
    PrefilterDeviceView V;
    uint64_t pos_begin, pos_end;      // global residue range of this chunk
    uint32_t q_first;                 // first query of the chunk
    uint32_t seq_bits;                // key = qLocal << seq_bits | target
    uint32_t *hit_count;              // [pos] (COUNT: written; GATHER: exclusive prefix, read)
    uint32_t *kmer_count;             // [pos] statistics
    uint64_t *keys; uint64_t *vals;   // GATHER outputs
};

template <bool GATHER>
__global__ __launch_bounds__(256) void probe_kernel(ProbeArgs A) {
    __shared__ int16_t sRow1[4][ROWCACHE];
    __shared__ uint16_t sIdx1[4][ROWCACHE];
    __shared__ uint32_t sPref[4][WAVE + 1];
    __shared__ uint16_t sIdx0[4][WAVE];
    const int w = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    const uint64_t p = A.pos_begin + (uint64_t) blockIdx.x * 4 + w;
    if (p >= A.pos_end) return;
    const uint64_t rel = p - A.pos_begin;
    const int thr = (int) A.V.q_kmer_thr[p];
    if (thr < 0) {                                     // no k-mer starts here (X inside, or too close to the end)
        if (!GATHER && lane == 0) { A.hit_count[rel] = 0; A.kmer_count[rel] = 0; }
        return;
    }
    const uint8_t *r = A.V.q_res + p;
    const uint32_t idx0 = r[0] + 20u * r[1] + 400u * r[3];      // spaced seed 1101010011 -> offsets 0,1,3,5,8,9
    const uint32_t idx1 = r[5] + 20u * r[8] + 400u * r[9];
    const int16_t *s0 = A.V.score3 + (size_t) idx0 * N3;
    const uint16_t *i0 = A.V.index3 + (size_t) idx0 * N3;
    const int16_t *s1 = A.V.score3 + (size_t) idx1 * N3;
    const uint16_t *i1 = A.V.index3 + (size_t) idx1 * N3;
    for (int k = lane; k < ROWCACHE; k += WAVE) { sRow1[w][k] = s1[k]; if (GATHER) sIdx1[w][k] = i1[k]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int cutoff1 = (int) (short) (thr - (int) sRow1[w][0]);   // threshold - best score of the second half

    uint32_t qLocal = 0, iPos = 0, qFirstHit = 0;
    uint32_t hitBase = 0;
    if (GATHER) {
        const uint32_t q = find_query(A.V.q_off, A.V.n_queries, p);
        qLocal = q - A.q_first;
        iPos = (uint32_t) (p - A.V.q_off[q]);
        qFirstHit = A.hit_count[A.V.q_off[q] - A.pos_begin];
        hitBase = A.hit_count[rel];
    }
    uint32_t hits = 0, kmers = 0;
    for (int a0 = 0; a0 < N3; a0 += WAVE) {
        const int a = a0 + lane;
        const int sa = (a < N3) ? (int) s0[a] : -32768;
        const bool valid = sa >= cutoff1;
        uint32_t nb = 0;
        if (valid) nb = (uint32_t) count_ge(sRow1[w], ROWCACHE, s1, (int) (short) (thr - sa));
        uint32_t groupTotal;
        const uint32_t excl = wave_excl_scan(nb, groupTotal);
        sPref[w][lane] = excl;
        if (lane == 0) sPref[w][WAVE] = groupTotal;
        sIdx0[w][lane] = valid ? i0[a] : (uint16_t) 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        kmers += groupTotal;
        // enumerate the (a,b) products of this group 64 at a time, in product order
        for (uint32_t base = 0; base < groupTotal; base += WAVE) {
            const uint32_t pr = base + lane;
            uint32_t size = 0, o0 = 0;
            if (pr < groupTotal) {
                int lo = 0, hi = WAVE;                 // largest al with sPref[al] <= pr (valid lanes have nb >= 1)
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sPref[w][mid] <= pr) lo = mid; else hi = mid; }
                const uint32_t b = pr - sPref[w][lo];
                const uint32_t ib = (b < ROWCACHE && GATHER) ? (uint32_t) sIdx1[w][b] : (uint32_t) i1[b];
                const uint32_t kmer = (uint32_t) sIdx0[w][lo] + N3 * ib;
                o0 = A.V.kmer_off[kmer];
                size = A.V.kmer_off[kmer + 1] - o0;
            }
            if (!GATHER) {
                hits += size;
            } else {
                uint32_t tot;
                const uint32_t ex = wave_excl_scan(size, tot);
                uint64_t dst = (uint64_t) hitBase + hits + ex;
                for (uint32_t e = 0; e < size; e++) {
                    const uint64_t ent = A.V.entries[o0 + e];
                    const uint32_t seq = (uint32_t) ent;
                    const uint32_t posj = (uint32_t) (ent >> 32) & 0xFFFFu;
                    const uint32_t diag = (iPos - posj) & 0xFFFFu;
                    A.keys[dst + e] = ((uint64_t) qLocal << A.seq_bits) | seq;
                    A.vals[dst + e] = ((uint64_t) (dst + e - qFirstHit) << 16) | diag;
                }
                hits += tot;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (__any(!valid)) break;
    }
    if (!GATHER) {
        hits = wave_sum(hits);
        if (lane == 0) { A.hit_count[rel] = hits; A.kmer_count[rel] = kmers; }
    }
}

// totals of a chunk after the scan + the reference's per-query databaseHits capacity check
// (QueryMatcher.cpp:43,281-316: a query gathering >= 2*max(1e6,dbSize) entries takes the overflow path)
__global__ __launch_bounds__(256) void chunk_totals_kernel(const uint64_t *qOff, uint32_t qFirst, uint32_t nq, uint64_t posBegin, uint64_t nPos,
                                                          const uint32_t *scan, const uint32_t *lastCount, uint64_t maxDbMatches,
                                                          unsigned long long *totals /* [0]=hits [1]=first overflowing query+1 */) {
    const uint32_t ql = blockIdx.x * blockDim.x + threadIdx.x;
    if (ql == 0) totals[0] = (unsigned long long) scan[nPos - 1] + lastCount[0];
    if (ql >= nq) return;
    const uint64_t b = qOff[qFirst + ql] - posBegin, e = qOff[qFirst + ql + 1] - posBegin;
    if (e == b) return;
    const uint64_t endv = (e < nPos) ? scan[e] : (uint64_t) scan[nPos - 1] + lastCount[0];
    if (endv - scan[b] >= maxDbMatches) atomicMax(&totals[1], (unsigned long long) (qFirst + ql) + 1ull);
}

// findDuplicates (computeTotalScore == false) on the (query,target)-sorted hit stream.
//   kept(t)    : low 8 bits of the diagonal equal those of the previous hit of the same (query,target);
//                the first hit of a target is compared with 0 (duplicateBitArray starts zeroed)
//   emitted(t) : kept(t) and the nearest earlier kept hit of the run has a different low byte (or none exists)
__global__ __launch_bounds__(256) void double_hit_flag_kernel(const uint64_t *keys, const uint64_t *vals, uint32_t n, uint8_t *flag) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t key = keys[t];
    const uint32_t lo = (uint32_t) vals[t] & 0xFFu;
    const bool samePrev = t > 0 && keys[t - 1] == key;
    const uint32_t prevLo = samePrev ? ((uint32_t) vals[t - 1] & 0xFFu) : 0u;
    uint8_t emit = 0;
    if (lo == prevLo) {
        emit = 1;
        if (samePrev) {
            uint32_t u = t - 1;
            while (true) {
                const uint32_t ulo = (uint32_t) vals[u] & 0xFFu;
                const bool uSame = u > 0 && keys[u - 1] == key;
                const uint32_t uprev = uSame ? ((uint32_t) vals[u - 1] & 0xFFu) : 0u;
                if (ulo == uprev) { emit = (ulo != lo) ? 1 : 0; break; }
                if (!uSame) break;
                u--;
            }
        }
    }
    flag[t] = emit;
}

// exact ungapped diagonal score of candidate c (sorted hit index sel[c])
struct CandArrays { uint32_t *q; uint32_t *id; uint32_t *ordinal; uint16_t *diag; int32_t *score; };

__global__ __launch_bounds__(256) void cand_score_kernel(PrefilterDeviceView V, uint32_t qFirst, uint32_t seqBits, const uint64_t *keys, const uint64_t *vals,
                                                         const uint32_t *sel, uint32_t n, CandArrays C) {
    __shared__ int8_t smat[21 * 21 + 3];
    for (int i = threadIdx.x; i < 21 * 21; i += blockDim.x) smat[i] = V.mat_ung[i];
    __syncthreads();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const uint32_t t = sel[c];
    const uint64_t key = keys[t], val = vals[t];
    const uint32_t ql = (uint32_t) (key >> seqBits), id = (uint32_t) (key & ((1ull << seqBits) - 1));
    const uint32_t d16 = (uint32_t) val & 0xFFFFu;
    const uint32_t q = qFirst + ql;
    const uint64_t qs = V.q_off[q], ts = V.t_off[id];
    const uint32_t qLen = (uint32_t) (V.q_off[q + 1] - qs), tLen = (uint32_t) (V.t_off[id + 1] - ts);
    const int diag = (int) (short) (uint16_t) d16;
    const uint32_t dist = min((0x10000u - d16) & 0xFFFFu, d16);
    uint32_t len = 0, q0 = 0, t0 = 0;
    if (diag >= 0 && dist < qLen) { len = min(tLen, qLen - dist); q0 = dist; }
    else if (diag < 0 && dist < tLen) { len = min(tLen - dist, qLen); t0 = dist; }
    const uint8_t *qr = V.q_res + qs + q0;
    const int8_t *corr = V.q_corr + qs + q0;
    const uint8_t *tr = V.t_masked + ts + t0;
    int score = 0, best = 0;
    for (uint32_t k = 0; k < len; k++) {
        const int curr = (int) (int8_t) (smat[qr[k] * 21 + tr[k]] + corr[k]);
        score = max(score + curr, 0);
        best = max(best, score);
    }
    C.q[c] = ql; C.id[c] = id; C.ordinal[c] = (uint32_t) (val >> 16); C.diag[c] = (uint16_t) d16; C.score[c] = best;
}

// keepMaxScoreElementOnly on the (query,target,arrival)-sorted candidates: a candidate survives when its clamped
// score is the maximum of its (query,target) run and no earlier candidate of the run has the same clamped score.
// Survivors below --min-ungapped-score can never be reported (diagonalThr >= minDiagScoreThr) and are dropped here.
__global__ __launch_bounds__(256) void keep_kernel(CandArrays C, uint32_t n, int minDiag, uint8_t *kept, uint32_t *perQuery) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const uint32_t q = C.q[c], id = C.id[c];
    const int s = min(C.score[c], 255);
    bool keep = s >= minDiag;
    for (uint32_t u = c; keep && u > 0; ) {            // earlier candidates of the run: none may reach s
        u--;
        if (C.q[u] != q || C.id[u] != id) break;
        if (min(C.score[u], 255) >= s) keep = false;
    }
    for (uint32_t u = c + 1; keep && u < n; u++) {    // later candidates: none may exceed s
        if (C.q[u] != q || C.id[u] != id) break;
        if (min(C.score[u], 255) > s) keep = false;
    }
    kept[c] = keep ? 1 : 0;
    if (keep) atomicAdd(&perQuery[q], 1u);
}

// sort key of a reportable hit: (query, score descending, target ascending); everything else sorts last
__global__ __launch_bounds__(256) void outkey_kernel(CandArrays C, uint32_t n, const uint8_t *kept, const uint32_t *perQuery, uint32_t maxHits,
                                                     uint32_t seqBits, uint64_t *outKey, uint32_t *outIdx, uint8_t *hostFlag) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const uint32_t q = C.q[c];
    const bool flagged = perQuery[q] >= maxHits;       // --max-seqs reached: the tie order needs the reference's bin logic
    hostFlag[c] = flagged ? 1 : 0;
    uint64_t key = ~0ull;
    if (kept[c] && !flagged) {
        // 20 bits query | (44 - seqBits) bits inverted score | seqBits bits target  (score field >= 17 bits)
        const uint32_t smax = (1u << min(44u - seqBits, 31u)) - 1u;
        const uint32_t sc = (uint32_t) min((uint32_t) C.score[c], smax);
        key = ((uint64_t) q << 44) | ((uint64_t) (smax - sc) << seqBits) | (uint64_t) C.id[c];
    }
    outKey[c] = key;
    outIdx[c] = c;
}

__global__ __launch_bounds__(256) void emit_kernel(CandArrays C, const uint32_t *sortedIdx, uint32_t nValid, mk_hit *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nValid) return;
    const uint32_t c = sortedIdx[i];
    mk_hit h;
    h.seq_id = C.id[c]; h.pref_score = C.score[c]; h.diagonal = C.diag[c]; h.pad_ = 0;
    out[i] = h;
}

struct HostCand { uint32_t q, id, ordinal; uint16_t diag; int32_t score; };
__global__ __launch_bounds__(256) void export_flagged_kernel(CandArrays C, const uint32_t *sel, uint32_t n, HostCand *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = sel[i];
    HostCand h;
    h.q = C.q[c]; h.id = C.id[c]; h.ordinal = C.ordinal[c]; h.diag = C.diag[c]; h.score = C.score[c];
    out[i] = h;
}

__global__ void first_invalid_kernel(const uint64_t *sortedKeys, uint32_t n, uint32_t *out) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sortedKeys[mid] != ~0ull) lo = mid + 1; else hi = mid; }
    out[0] = lo;
}

// exact ungapped self score of a query on diagonal 0 (QueryMatcher::rescoreHits, QueryMatcher.cpp:525-531);
// only needed when the score threshold saturates at 255
int self_score(const SubMat &ung, const uint8_t *q, const int8_t *corr, int L) {
    int s = 0, best = 0;
    for (int k = 0; k < L; k++) {
        const int curr = (int) (int8_t) ((int8_t) ung.sub[q[k]][q[k]] + corr[k]);
        s = std::max(s + curr, 0);
        best = std::max(best, s);
    }
    return best;
}

}  // namespace

#define PCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return MK_ERR_DEVICE; } } while (0)
#define PNULL(p) do { if (!(p)) { err = "device scratch allocation failed (" #p ")"; return MK_ERR_DEVICE; } } while (0)

int run_prefilter(const PrefilterDeviceView &V, const std::vector<uint64_t> &qOff, const std::vector<uint8_t> &qRes,
                  const int8_t *qCorrHost,
                  const std::vector<uint64_t> &tOff, const mk_params &P, int binCount, hipStream_t stream,
                  HostBlock &outBlk, size_t &nOut, std::vector<uint64_t> &outOff, std::string &err, timed_begin_fn tb, timed_end_fn te, timed_set_fn ts) {
    const uint32_t nq = V.n_queries;
    nOut = 0;
    outOff.assign((size_t) nq + 1, 0);
    // room for n more hits in the result block; the estimate of the final size follows the progress through the batch
    auto reserve_out = [&](size_t n, uint32_t qDone) -> bool {
        if ((nOut + n) * sizeof(mk_hit) <= outBlk.cap && outBlk.p) return true;
        if (hipStreamSynchronize(stream) != hipSuccess) return false;      // DMA into the old block must have landed
        const double frac = std::max(0.02, (double) qDone / (double) nq);
        const size_t want = (size_t) ((double) (nOut + n) / frac * 1.1) + 4096;
        return outBlk.reserve(std::max(want, nOut + n + 4096) * sizeof(mk_hit), nOut * sizeof(mk_hit));
    };
    const int maxHits = std::min<int>(P.max_seqs, (int) V.n_targets);
    const uint64_t dbSize = V.n_targets;
    const uint64_t maxDbMatches = std::max<uint64_t>(1000000, dbSize) * 2;   // QueryMatcher.cpp:43
    const size_t HIT_CAP = 768u << 20;                // index hits per chunk kept in HBM: 32 B each (key+value, double buffered)
    const uint64_t POS_CAP = 24u << 20;               // residues per chunk
    const uint32_t QCAP = 1u << 20;                   // queries per chunk (20-bit field of the output sort key)
    if (dbSize >= (1ull << 27)) { err = "more than 2^27 targets"; return MK_ERR_UNSUPPORTED; }
    uint32_t seqBits = 1; while ((1ull << seqBits) < dbSize) seqBits++;
    double hitsPerPos = 0;                            // running estimate used to size the next chunk
    SubMat ungMat;
    build_submat(ungMat, MAT_BLOSUM62, 2.0f, -0.2f);
    unsigned long long *dTotals = (unsigned long long *) dev_scratch("pf_totals", 64);
    unsigned long long *hTotals = (unsigned long long *) pinned_scratch("pf_totals_h", 64);
    PNULL(dTotals); PNULL(hTotals);
    uint32_t q0 = 0;
    while (q0 < nq) {
        uint32_t q1 = q0;
        {
            uint64_t posBudget = hitsPerPos > 0 ? std::min<uint64_t>(POS_CAP, (uint64_t) (0.8 * (double) HIT_CAP / hitsPerPos))
                                                : std::min<uint64_t>(POS_CAP, 1u << 20);   // first chunk: small probe
            while (q1 < nq && q1 - q0 < QCAP && (qOff[q1 + 1] - qOff[q0] <= posBudget || q1 == q0)) q1++;
        }
        uint64_t totalHits = 0, nPos = 0;
        uint32_t *dHit = nullptr, *dKmer = nullptr;
        int thCount = -1;
        for (;;) {
            nPos = qOff[q1] - qOff[q0];
            if (nPos == 0) break;
            dHit = (uint32_t *) dev_scratch("pf_hit", (nPos + 1) * 4);
            dKmer = (uint32_t *) dev_scratch("pf_kmer", (nPos + 1) * 4);
            uint32_t *dLast = (uint32_t *) dev_scratch("pf_last", 16);
            PNULL(dHit); PNULL(dKmer); PNULL(dLast);
            ProbeArgs A;
            A.V = V; A.pos_begin = qOff[q0]; A.pos_end = qOff[q1]; A.q_first = q0; A.seq_bits = seqBits;
            A.hit_count = dHit; A.kmer_count = dKmer; A.keys = nullptr; A.vals = nullptr;
            const unsigned blocks = (unsigned) ((nPos + 3) / 4);
            thCount = tb("kmer_probe_count", 0, 0);
            hipLaunchKernelGGL(probe_kernel<false>, dim3(blocks), dim3(256), 0, stream, A);
            te(thCount);
            PCHK(hipGetLastError());
            // totals: k-mers (reduce), hits (scan); the count of the last position is saved before the in-place scan
            PCHK(hipMemcpyAsync(dLast, dHit + (nPos - 1), 4, hipMemcpyDeviceToDevice, stream));
            size_t tempBytes = 0, tb2 = 0;
            hipcub::DeviceReduce::Sum(nullptr, tempBytes, dKmer, dTotals + 2, (int) nPos, stream);
            hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, dHit, dHit, (int) nPos, stream);
            void *temp = dev_scratch("pf_temp", std::max(tempBytes, tb2));
            PNULL(temp);
            // kmer_count is u32 per position; the sum can exceed 2^32 -> accumulate in 64 bit via a transform iterator
            {
                hipcub::TransformInputIterator<unsigned long long, hipcub::CastOp<unsigned long long>, uint32_t *> it(dKmer, hipcub::CastOp<unsigned long long>());
                size_t t3 = 0;
                hipcub::DeviceReduce::Sum(nullptr, t3, it, dTotals + 2, (int) nPos, stream);
                temp = dev_scratch("pf_temp", std::max(std::max(tempBytes, tb2), t3));
                PNULL(temp);
                PCHK(hipcub::DeviceReduce::Sum(temp, t3, it, dTotals + 2, (int) nPos, stream));
            }
            int th = tb("scan", 8.0 * nPos, 0);
            PCHK(hipcub::DeviceScan::ExclusiveSum(temp, tb2, dHit, dHit, (int) nPos, stream));
            te(th);
            PCHK(hipMemsetAsync(dTotals, 0, 16, stream));
            hipLaunchKernelGGL(chunk_totals_kernel, dim3((q1 - q0 + 255) / 256), dim3(256), 0, stream, V.q_off, q0, q1 - q0, qOff[q0], nPos,
                               dHit, dLast, maxDbMatches, dTotals);
            PCHK(hipGetLastError());
            PCHK(hipMemcpyAsync(hTotals, dTotals, 24, hipMemcpyDeviceToHost, stream));
            PCHK(sync_wait(stream, "wait_prefilter"));
            totalHits = hTotals[0];
            if (hTotals[1] != 0) { err = "query " + std::to_string(hTotals[1] - 1) + " overflows the reference's databaseHits buffer (QueryMatcher.cpp:281-316 is not restated)"; return MK_ERR_UNSUPPORTED; }
            ts(thCount, 8.0 * (double) hTotals[2] + 2.0 * 2.0 * ROWCACHE * (double) nPos, (double) hTotals[2]);
            hitsPerPos = std::max(1.0, (double) totalHits / (double) nPos);
            if (totalHits > HIT_CAP && q1 - q0 > 1) { q1 = q0 + (q1 - q0) / 2; continue; }
            break;
        }
        const uint32_t nqc = q1 - q0;
        std::vector<uint32_t> chunkCnt(nqc, 0);
        const mk_hit *devHits = nullptr;               // device-final hits of the chunk, staged (chunks with --max-seqs queries)
        size_t nDevHits = 0;
        bool devDirect = false;                        // ... or already on their way to the result block
        std::vector<std::vector<mk_hit>> hostHits;     // per flagged query
        std::vector<uint32_t> hostQ;
        if (nPos > 0 && totalHits > 0) {
            if (totalHits >= 0x7FFFFFFFull) { err = "a single query produces >= 2^31 index hits"; return MK_ERR_UNSUPPORTED; }
            const uint32_t nHits = (uint32_t) totalHits;
            uint64_t *dKeys = (uint64_t *) dev_scratch("pf_keys", (size_t) nHits * 8), *dKeys2 = (uint64_t *) dev_scratch("pf_keys2", (size_t) nHits * 8);
            uint64_t *dVals = (uint64_t *) dev_scratch("pf_vals", (size_t) nHits * 8), *dVals2 = (uint64_t *) dev_scratch("pf_vals2", (size_t) nHits * 8);
            PNULL(dKeys); PNULL(dKeys2); PNULL(dVals); PNULL(dVals2);
            ProbeArgs A;
            A.V = V; A.pos_begin = qOff[q0]; A.pos_end = qOff[q1]; A.q_first = q0; A.seq_bits = seqBits;
            A.hit_count = dHit; A.kmer_count = dKmer; A.keys = dKeys; A.vals = dVals;
            const unsigned blocks = (unsigned) ((nPos + 3) / 4);
            // gather pass: offset pairs again + 8 B per index entry read + 16 B (key,value) written per entry
            int th = tb("kmer_probe_gather", 8.0 * (double) hTotals[2] + 24.0 * (double) totalHits + 4.0 * ROWCACHE * (double) nPos, (double) hTotals[2]);
            hipLaunchKernelGGL(probe_kernel<true>, dim3(blocks), dim3(256), 0, stream, A);
            te(th);
            PCHK(hipGetLastError());
            // 4. stable sort by (query, target)
            int qBits = 1; while ((1u << qBits) < nqc) qBits++;
            hipcub::DoubleBuffer<uint64_t> kb(dKeys, dKeys2), vb(dVals, dVals2);
            size_t tempBytes = 0;
            hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, kb, vb, (int) nHits, 0, (int) seqBits + qBits, stream);
            void *temp = dev_scratch("pf_temp", tempBytes);
            PNULL(temp);
            const int passes = ((int) seqBits + qBits + 7) / 8;
            th = tb("sort_hits", 32.0 * passes * (double) nHits, 0);
            PCHK(hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, kb, vb, (int) nHits, 0, (int) seqBits + qBits, stream));
            te(th);
            // 5. double-hit rule -> flags -> ordered compaction
            uint8_t *dFlag = (uint8_t *) dev_scratch("pf_flag", nHits);
            uint32_t *dSel = (uint32_t *) dev_scratch("pf_sel", (size_t) nHits * 4);
            uint32_t *dNum = (uint32_t *) dev_scratch("pf_num", 64);
            PNULL(dFlag); PNULL(dSel); PNULL(dNum);
            th = tb("double_hit", 17.0 * nHits, 0);
            hipLaunchKernelGGL(double_hit_flag_kernel, dim3((nHits + 255) / 256), dim3(256), 0, stream, kb.Current(), vb.Current(), nHits, dFlag);
            te(th);
            PCHK(hipGetLastError());
            {
                hipcub::CountingInputIterator<uint32_t> iota(0);
                size_t t2 = 0;
                hipcub::DeviceSelect::Flagged(nullptr, t2, iota, dFlag, dSel, dNum, (int) nHits, stream);
                temp = dev_scratch("pf_temp", t2);
                PNULL(temp);
                th = tb("select_candidates", 5.0 * nHits, 0);
                PCHK(hipcub::DeviceSelect::Flagged(temp, t2, iota, dFlag, dSel, dNum, (int) nHits, stream));
                te(th);
            }
            uint32_t *hNum = (uint32_t *) pinned_scratch("pf_num_h", 64);
            PNULL(hNum);
            PCHK(hipMemcpyAsync(hNum, dNum, 4, hipMemcpyDeviceToHost, stream));
            PCHK(sync_wait(stream, "wait_prefilter"));
            const uint32_t nCand = hNum[0];
            if (nCand > 0) {
                CandArrays C;
                C.q = (uint32_t *) dev_scratch("pf_cq", (size_t) nCand * 4); C.id = (uint32_t *) dev_scratch("pf_cid", (size_t) nCand * 4);
                C.ordinal = (uint32_t *) dev_scratch("pf_cord", (size_t) nCand * 4); C.diag = (uint16_t *) dev_scratch("pf_cdiag", (size_t) nCand * 2);
                C.score = (int32_t *) dev_scratch("pf_cscore", (size_t) nCand * 4);
                uint8_t *dKept = (uint8_t *) dev_scratch("pf_kept", nCand), *dHostFlag = (uint8_t *) dev_scratch("pf_hostflag", nCand);
                uint32_t *dPerQ = (uint32_t *) dev_scratch("pf_perq", (size_t) nqc * 4);
                uint64_t *dOutKey = (uint64_t *) dev_scratch("pf_okey", (size_t) nCand * 8), *dOutKey2 = (uint64_t *) dev_scratch("pf_okey2", (size_t) nCand * 8);
                uint32_t *dOutIdx = (uint32_t *) dev_scratch("pf_oidx", (size_t) nCand * 4), *dOutIdx2 = (uint32_t *) dev_scratch("pf_oidx2", (size_t) nCand * 4);
                PNULL(C.q); PNULL(C.id); PNULL(C.ordinal); PNULL(C.diag); PNULL(C.score); PNULL(dKept); PNULL(dHostFlag); PNULL(dPerQ);
                PNULL(dOutKey); PNULL(dOutKey2); PNULL(dOutIdx); PNULL(dOutIdx2);
                th = tb("diag_score", 28.0 * nCand, 0);
                hipLaunchKernelGGL(cand_score_kernel, dim3((nCand + 255) / 256), dim3(256), 0, stream, V, q0, seqBits, kb.Current(), vb.Current(), dSel, nCand, C);
                te(th);
                PCHK(hipGetLastError());
                PCHK(hipMemsetAsync(dPerQ, 0, (size_t) nqc * 4, stream));
                th = tb("select_hits", 20.0 * nCand, 0);
                hipLaunchKernelGGL(keep_kernel, dim3((nCand + 255) / 256), dim3(256), 0, stream, C, nCand, P.min_ungapped_score, dKept, dPerQ);
                hipLaunchKernelGGL(outkey_kernel, dim3((nCand + 255) / 256), dim3(256), 0, stream, C, nCand, dKept, dPerQ, (uint32_t) maxHits, seqBits, dOutKey, dOutIdx, dHostFlag);
                PCHK(hipGetLastError());
                hipcub::DoubleBuffer<uint64_t> ob(dOutKey, dOutKey2);
                hipcub::DoubleBuffer<uint32_t> ib(dOutIdx, dOutIdx2);
                size_t t2 = 0;
                hipcub::DeviceRadixSort::SortPairs(nullptr, t2, ob, ib, (int) nCand, 0, 64, stream);
                temp = dev_scratch("pf_temp", t2);
                PNULL(temp);
                PCHK(hipcub::DeviceRadixSort::SortPairs(temp, t2, ob, ib, (int) nCand, 0, 64, stream));
                hipLaunchKernelGGL(first_invalid_kernel, dim3(1), dim3(1), 0, stream, ob.Current(), nCand, dNum + 1);
                // candidates of flagged queries, in (query,target,arrival) order, for the host
                hipcub::CountingInputIterator<uint32_t> iota(0);
                size_t t3 = 0;
                uint32_t *dFlagSel = dSel;            // the first selection is consumed; reuse its buffer
                hipcub::DeviceSelect::Flagged(nullptr, t3, iota, dHostFlag, dFlagSel, dNum + 2, (int) nCand, stream);
                void *temp2 = dev_scratch("pf_temp2", t3);
                PNULL(temp2);
                PCHK(hipcub::DeviceSelect::Flagged(temp2, t3, iota, dHostFlag, dFlagSel, dNum + 2, (int) nCand, stream));
                te(th);
                PCHK(hipMemcpyAsync(hNum, dNum, 12, hipMemcpyDeviceToHost, stream));
                uint32_t *hPerQ = (uint32_t *) pinned_scratch("pf_perq_h", (size_t) nqc * 4);
                PNULL(hPerQ);
                PCHK(hipMemcpyAsync(hPerQ, dPerQ, (size_t) nqc * 4, hipMemcpyDeviceToHost, stream));
                PCHK(sync_wait(stream, "wait_prefilter"));
                const uint32_t nValid = hNum[1], nFlagged = hNum[2];
                mk_hit *dHitsOut = nullptr;
                if (nValid > 0) {
                    dHitsOut = (mk_hit *) dev_scratch("pf_hits_out", (size_t) nValid * sizeof(mk_hit));
                    PNULL(dHitsOut);
                    hipLaunchKernelGGL(emit_kernel, dim3((nValid + 255) / 256), dim3(256), 0, stream, C, ib.Current(), nValid, dHitsOut);
                    PCHK(hipGetLastError());
                }
                if (nFlagged == 0) {
                    // every query of the chunk is final on the device: DMA the compact hit array to its final place
                    if (nValid > 0) {
                        if (!reserve_out(nValid, q1)) { err = "pinned host allocation for the prefilter result failed"; return MK_ERR_DEVICE; }
                        PCHK(hipMemcpyAsync((mk_hit *) outBlk.p + nOut, dHitsOut, (size_t) nValid * sizeof(mk_hit), hipMemcpyDeviceToHost, stream));
                    }
                    for (uint32_t ql = 0; ql < nqc; ql++) chunkCnt[ql] = hPerQ[ql];
                    devDirect = true;
                    nDevHits = nValid;
                } else {
                    if (nValid > 0) {
                        mk_hit *hHitsOut = (mk_hit *) pinned_scratch("pf_hits_out_h", (size_t) nValid * sizeof(mk_hit));
                        PNULL(hHitsOut);
                        PCHK(hipMemcpyAsync(hHitsOut, dHitsOut, (size_t) nValid * sizeof(mk_hit), hipMemcpyDeviceToHost, stream));
                        devHits = hHitsOut; nDevHits = nValid;
                    }
                    for (uint32_t ql = 0; ql < nqc; ql++) chunkCnt[ql] = hPerQ[ql] >= (uint32_t) maxHits ? 0 : hPerQ[ql];
                    // exact reference logic for the queries that reached --max-seqs (tie order depends on BINSIZE)
                    HostCand *dHC = (HostCand *) dev_scratch("pf_hostcand", (size_t) nFlagged * sizeof(HostCand));
                    HostCand *hHC = (HostCand *) pinned_scratch("pf_hostcand_h", (size_t) nFlagged * sizeof(HostCand));
                    PNULL(dHC); PNULL(hHC);
                    hipLaunchKernelGGL(export_flagged_kernel, dim3((nFlagged + 255) / 256), dim3(256), 0, stream, C, dFlagSel, nFlagged, dHC);
                    PCHK(hipGetLastError());
                    PCHK(hipMemcpyAsync(hHC, dHC, (size_t) nFlagged * sizeof(HostCand), hipMemcpyDeviceToHost, stream));
                    PCHK(sync_wait(stream, "wait_prefilter"));
                    ScopedHost sh("host_prefilter_maxseqs");
                    std::vector<uint32_t> runStart;                    // candidate runs, one per flagged query (ascending query)
                    for (uint32_t k = 0; k < nFlagged; k++) if (k == 0 || hHC[k].q != hHC[k - 1].q) runStart.push_back(k);
                    runStart.push_back(nFlagged);
                    const size_t nRuns = runStart.size() - 1;
                    hostQ.resize(nRuns);
                    hostHits.resize(nRuns);
                    int failed = 0;
#pragma omp parallel
                    {
                        std::vector<Cand> perQuery;
#pragma omp for schedule(dynamic, 4)
                        for (size_t r = 0; r < nRuns; r++) {
                            const uint32_t k0 = runStart[r], k1 = runStart[r + 1];
                            const uint32_t ql = hHC[k0].q;
                            perQuery.clear();
                            for (uint32_t k = k0; k < k1; k++) perQuery.push_back(Cand{hHC[k].id, hHC[k].diag, hHC[k].score, hHC[k].ordinal});
                            std::sort(perQuery.begin(), perQuery.end(), [](const Cand &a, const Cand &b) { return a.ordinal < b.ordinal; });
                            const uint32_t q = q0 + ql;
                            int n255 = 0;
                            for (const Cand &c : perQuery) n255 += c.score >= 255;
                            int self = 0;
                            if (n255 >= maxHits) {                     // threshold saturates: QueryMatcher.cpp:525-531 needs the exact self score
                                const int L = (int) (qOff[q + 1] - qOff[q]);
                                std::vector<int8_t> corr((size_t) L);
                                if (qCorrHost) std::memcpy(corr.data(), qCorrHost + qOff[q], (size_t) L);
                                else {
#pragma omp critical(mk_pf_corr)
                                    if (hipMemcpy(corr.data(), V.q_corr + qOff[q], (size_t) L, hipMemcpyDeviceToHost) != hipSuccess) failed = 1;
                                }
                                self = self_score(ungMat, qRes.data() + qOff[q], corr.data(), L);
                            }
                            std::vector<mk_hit> hh((size_t) maxHits);
                            const int cnt = select_hits(perQuery, binCount, maxHits, P.min_ungapped_score, self, hh.data());
                            hh.resize((size_t) cnt);
                            hostQ[r] = ql;
                            hostHits[r] = std::move(hh);
                        }
                    }
                    if (failed) { err = "hipMemcpy of the diagonal correction failed"; return MK_ERR_DEVICE; }
                }
            }
        }
        // append the chunk: device-final hits are compact in query order; flagged queries come from the host lists
        if (devDirect || hostQ.empty()) {
            if (!devDirect && nDevHits) {                              // (not reached today: the staged copy implies flagged queries)
                if (!reserve_out(nDevHits, q1)) { err = "pinned host allocation for the prefilter result failed"; return MK_ERR_DEVICE; }
                std::memcpy((mk_hit *) outBlk.p + nOut, devHits, nDevHits * sizeof(mk_hit));
            }
            uint64_t o = outOff[q0];
            for (uint32_t ql = 0; ql < nqc; ql++) { o += chunkCnt[ql]; outOff[(size_t) q0 + ql + 1] = o; }
            nOut += nDevHits;
        } else {
            ScopedHost sh("host_prefilter_append");
            size_t extra = 0;
            for (const auto &hh : hostHits) extra += hh.size();
            if (!reserve_out(nDevHits + extra, q1)) { err = "pinned host allocation for the prefilter result failed"; return MK_ERR_DEVICE; }
            mk_hit *dst = (mk_hit *) outBlk.p;
            size_t dev = 0, hk = 0;
            for (uint32_t ql = 0; ql < nqc; ql++) {
                const size_t qg = (size_t) q0 + ql;
                if (hk < hostQ.size() && hostQ[hk] == ql) {
                    if (!hostHits[hk].empty()) std::memcpy(dst + nOut, hostHits[hk].data(), hostHits[hk].size() * sizeof(mk_hit));
                    nOut += hostHits[hk].size();
                    hk++;
                } else {
                    const uint32_t c = chunkCnt[ql];
                    if (c) std::memcpy(dst + nOut, devHits + dev, (size_t) c * sizeof(mk_hit));
                    dev += c; nOut += c;
                }
                outOff[qg + 1] = nOut;
            }
        }
        q0 = q1;
    }
    (void) tOff;
    PCHK(sync_wait(stream, "wait_prefilter"));         // the last DMA into the result block
    return MK_OK;
}

}  // namespace mk
