// metaeuk_amd/csrc/mk_prefilter.hip -- the k-mer prefilter on gfx950.
// Replaces, for a whole batch of queries at once:
//   KmerGenerator::generateKmerList / calculateArrayProduct   M/src/prefiltering/KmerGenerator.cpp:107-216
//   QueryMatcher::match (index probes + gather)               M/src/prefiltering/QueryMatcher.cpp:213-346
//   CacheFriendlyOperations::findDuplicates (double hits)     M/src/prefiltering/CacheFriendlyOperations.cpp:185-274
//   UngappedAlignment::computeScores                          M/src/prefiltering/UngappedAlignment.cpp:331-362
// Pipeline per chunk of queries (everything stays in HBM; only the surviving diagonals go to the host):
//   1. probe_kernel<COUNT>   one wave per k-mer start: enumerate the similar k-mers (two sorted 3-mer rows,
//                            product order of the reference), read the index offset pair of each, sum list sizes
//   2. exclusive scan        (hipcub) -> canonical position of every index entry ("ordinal")
//   3. probe_kernel<GATHER>  same enumeration, copies the index lists: key = (query, target), value = (ordinal, diagonal)
//   4. stable radix sort     (hipcub) by key: per (query,target) the hits are now in the reference's arrival order
//   5. double_hit_kernel     the sequential 8-bit-diagonal rule of findDuplicates as a neighbour test + short backward walk
//   6. diag_score_kernel     exact ungapped score of every surviving (query, target, diagonal)
// The host then applies the per-query best-diagonal / threshold / top-N logic (mk::select_hits).
#include "mk_prefilter.hpp"
#include "mk_host.hpp"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstring>

namespace mk {

namespace {

constexpr int WAVE = 64;
constexpr int ROWCACHE = 512;     // leading entries of the second 3-mer row staged in LDS per wave
constexpr int N3 = 8000;

struct DCand { uint32_t q; uint32_t id; uint32_t ordinal; uint32_t diag; };

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
                int lo = 0, hi = WAVE;                 // largest al with sPref[al] <= pr
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sPref[w][mid] <= pr) lo = mid; else hi = mid; }
                // skip over empty lanes that share the same prefix value
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
                    A.keys[dst + e] = ((uint64_t) qLocal << 32) | seq;
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

// findDuplicates (computeTotalScore == false) on the (query,target)-sorted hit stream.
//   kept(t)    : low 8 bits of the diagonal equal those of the previous hit of the same (query,target);
//                the first hit of a target is compared with 0 (duplicateBitArray starts zeroed)
//   emitted(t) : kept(t) and the nearest earlier kept hit of the run has a different low byte (or none exists)
__global__ __launch_bounds__(256) void double_hit_kernel(const uint64_t *keys, const uint64_t *vals, uint64_t n,
                                                         DCand *out, uint32_t *outCount, uint32_t outCap) {
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t key = keys[t];
    const uint32_t lo = (uint32_t) vals[t] & 0xFFu;
    const bool samePrev = t > 0 && keys[t - 1] == key;
    const uint32_t prevLo = samePrev ? ((uint32_t) vals[t - 1] & 0xFFu) : 0u;
    if (lo != prevLo) return;                          // not kept
    bool emit = true;
    if (samePrev) {
        // walk back to the nearest kept hit of this run
        uint64_t u = t - 1;
        while (true) {
            const uint32_t ulo = (uint32_t) vals[u] & 0xFFu;
            const bool uSame = u > 0 && keys[u - 1] == key;
            const uint32_t uprev = uSame ? ((uint32_t) vals[u - 1] & 0xFFu) : 0u;
            if (ulo == uprev) { emit = (ulo != lo); break; }
            if (!uSame) break;
            u--;
        }
    }
    if (!emit) return;
    const uint32_t slot = atomicAdd(outCount, 1u);
    if (slot < outCap) {
        DCand c;
        c.q = (uint32_t) (key >> 32); c.id = (uint32_t) key;
        c.ordinal = (uint32_t) (vals[t] >> 16); c.diag = (uint32_t) vals[t] & 0xFFFFu;
        out[slot] = c;
    }
}

__global__ __launch_bounds__(256) void diag_score_kernel(PrefilterDeviceView V, uint32_t qFirst, const DCand *cand, uint32_t n, int32_t *scores) {
    __shared__ int8_t smat[21 * 21 + 3];
    for (int i = threadIdx.x; i < 21 * 21; i += blockDim.x) smat[i] = V.mat_ung[i];
    __syncthreads();
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const DCand c = cand[id];
    const uint32_t q = qFirst + c.q;
    const uint64_t qs = V.q_off[q], ts = V.t_off[c.id];
    const uint32_t qLen = (uint32_t) (V.q_off[q + 1] - qs), tLen = (uint32_t) (V.t_off[c.id + 1] - ts);
    const int diag = (int) (short) (uint16_t) c.diag;
    const uint32_t d16 = c.diag & 0xFFFFu;
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
    scores[id] = best;
}

template <typename T>
struct Dev {
    T *p = nullptr; size_t cap = 0;
    ~Dev() { if (p) hipFree(p); }
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
};

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

int run_prefilter(const PrefilterDeviceView &V, const std::vector<uint64_t> &qOff, const std::vector<uint8_t> &qRes,
                  const int8_t *qCorrHost,
                  const std::vector<uint64_t> &tOff, const mk_params &P, int binCount, hipStream_t stream,
                  std::vector<mk_hit> &outHits, std::vector<uint64_t> &outOff, std::string &err, timed_begin_fn tb, timed_end_fn te, timed_set_fn ts) {
    const uint32_t nq = V.n_queries;
    outHits.clear();
    outOff.assign((size_t) nq + 1, 0);
    const int maxHits = std::min<int>(P.max_seqs, (int) V.n_targets);
    const uint64_t dbSize = V.n_targets;
    const uint64_t maxDbMatches = std::max<uint64_t>(1000000, dbSize) * 2;   // QueryMatcher.cpp:43
    const size_t HIT_CAP = 768u << 20;                // index hits per chunk kept in HBM: 32 B each (key+value, double buffered)
    const uint64_t POS_CAP = 24u << 20;               // residues per chunk
    double hitsPerPos = 0;                            // running estimate used to size the next chunk
    SubMat ungMat, kmerMat;
    build_submat(ungMat, MAT_BLOSUM62, 2.0f, -0.2f);
    Dev<uint32_t> dHit, dKmer, dCount;
    Dev<uint64_t> dKeys, dVals, dKeys2, dVals2;
    Dev<uint8_t> dTemp;
    Dev<DCand> dCand;
    Dev<int32_t> dScore;
    uint32_t q0 = 0;
    std::vector<uint32_t> hHit, hKmer;
    uint64_t totalKmers = 0;
    std::vector<DCand> hCand;
    std::vector<int32_t> hScore;
    std::vector<Cand> perQuery;
    PCHK(dCount.reserve(1));
    while (q0 < nq) {
        // chunk = as many whole queries as fit POS_CAP residues
        uint32_t q1 = q0;
        {
            uint64_t posBudget = POS_CAP;
            if (hitsPerPos > 0) posBudget = std::min<uint64_t>(POS_CAP, (uint64_t) (0.8 * (double) HIT_CAP / hitsPerPos));
            else posBudget = std::min<uint64_t>(POS_CAP, 1u << 20);   // first chunk: small probe
            while (q1 < nq && (qOff[q1 + 1] - qOff[q0] <= posBudget || q1 == q0)) q1++;
        }
        bool shrunk;
        uint64_t totalHits = 0;
        uint64_t nPos = 0;
        do {
            shrunk = false;
            nPos = qOff[q1] - qOff[q0];
            if (nPos == 0) break;
            PCHK(dHit.reserve(nPos + 1));
            PCHK(dKmer.reserve(nPos + 1));
            ProbeArgs A;
            A.V = V; A.pos_begin = qOff[q0]; A.pos_end = qOff[q1]; A.q_first = q0;
            A.hit_count = dHit.p; A.kmer_count = dKmer.p; A.keys = nullptr; A.vals = nullptr;
            const unsigned blocks = (unsigned) ((nPos + 3) / 4);
            int th = tb("kmer_probe_count", 0, 0);
            hipLaunchKernelGGL(probe_kernel<false>, dim3(blocks), dim3(256), 0, stream, A);
            te(th);
            PCHK(hipGetLastError());
            hHit.resize(nPos); hKmer.resize(nPos);
            PCHK(hipMemcpyAsync(hHit.data(), dHit.p, nPos * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            PCHK(hipMemcpyAsync(hKmer.data(), dKmer.p, nPos * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            PCHK(hipStreamSynchronize(stream));
            totalKmers = 0;
            for (uint64_t k = 0; k < nPos; k++) totalKmers += hKmer[k];
            // algorithmic bytes of the count pass: one offset pair per similar k-mer + the two 3-mer score rows' heads
            ts(th, 8.0 * (double) totalKmers + 2.0 * 2.0 * ROWCACHE * (double) nPos, (double) totalKmers);
            totalHits = 0;
            uint64_t perQ = 0;
            uint32_t qi = q0;
            for (uint64_t k = 0; k < nPos; k++) {
                while (qOff[qi + 1] - qOff[q0] <= k) { qi++; perQ = 0; }
                perQ += hHit[k];
                // reference overflow path (QueryMatcher.cpp:281-316): a query whose hit buffer would wrap
                if (perQ >= maxDbMatches) { err = "query " + std::to_string(qi) + " overflows the reference's databaseHits buffer (not restated)"; return MK_ERR_UNSUPPORTED; }
                totalHits += hHit[k];
            }
            if (nPos > 0) hitsPerPos = std::max(1.0, (double) totalHits / (double) nPos);
            if (totalHits > HIT_CAP && q1 - q0 > 1) { q1 = q0 + (q1 - q0) / 2; shrunk = true; }
        } while (shrunk);
        std::vector<uint32_t> chunkCnt(q1 - q0, 0), slot(q1 - q0 + 1, 0);
        std::vector<mk_hit> chunkHits;
        if (nPos > 0 && totalHits > 0) {
            if (totalHits >= 0xFFFFFFFFull) { err = "a single query produces >= 2^32 index hits"; return MK_ERR_UNSUPPORTED; }
            // 2. exclusive scan of per-position hit counts (in place)
            size_t tempBytes = 0;
            hipcub::DeviceScan::ExclusiveSum(nullptr, tempBytes, dHit.p, dHit.p, (int) nPos, stream);
            PCHK(dTemp.reserve(tempBytes));
            int th = tb("scan", 8.0 * nPos, 0);
            PCHK(hipcub::DeviceScan::ExclusiveSum(dTemp.p, tempBytes, dHit.p, dHit.p, (int) nPos, stream));
            te(th);
            // 3. gather
            PCHK(dKeys.reserve(totalHits)); PCHK(dVals.reserve(totalHits));
            PCHK(dKeys2.reserve(totalHits)); PCHK(dVals2.reserve(totalHits));
            ProbeArgs A;
            A.V = V; A.pos_begin = qOff[q0]; A.pos_end = qOff[q1]; A.q_first = q0;
            A.hit_count = dHit.p; A.kmer_count = dKmer.p; A.keys = dKeys.p; A.vals = dVals.p;
            const unsigned blocks = (unsigned) ((nPos + 3) / 4);
            // gather pass: offset pairs again + 8 B per index entry read + 16 B (key,value) written per entry
            th = tb("kmer_probe_gather", 8.0 * (double) totalKmers + 24.0 * (double) totalHits + 4.0 * ROWCACHE * (double) nPos, (double) totalKmers);
            hipLaunchKernelGGL(probe_kernel<true>, dim3(blocks), dim3(256), 0, stream, A);
            te(th);
            PCHK(hipGetLastError());
            // 4. stable sort by (query, target)
            int qBits = 1; while ((1u << qBits) < (q1 - q0)) qBits++;
            hipcub::DoubleBuffer<uint64_t> kb(dKeys.p, dKeys2.p), vb(dVals.p, dVals2.p);
            tempBytes = 0;
            hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, kb, vb, (int) totalHits, 0, 32 + qBits, stream);
            PCHK(dTemp.reserve(tempBytes));
            th = tb("sort_hits", 32.0 * totalHits, 0);
            PCHK(hipcub::DeviceRadixSort::SortPairs(dTemp.p, tempBytes, kb, vb, (int) totalHits, 0, 32 + qBits, stream));
            te(th);
            // 5. double-hit rule
            const size_t candCap = std::min<size_t>(totalHits, 64u << 20);
            PCHK(dCand.reserve(candCap));
            PCHK(hipMemsetAsync(dCount.p, 0, sizeof(uint32_t), stream));
            th = tb("double_hit", 16.0 * totalHits, 0);
            hipLaunchKernelGGL(double_hit_kernel, dim3((unsigned) ((totalHits + 255) / 256)), dim3(256), 0, stream,
                               kb.Current(), vb.Current(), totalHits, dCand.p, dCount.p, (uint32_t) candCap);
            te(th);
            PCHK(hipGetLastError());
            uint32_t nCand = 0;
            PCHK(hipMemcpyAsync(&nCand, dCount.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            PCHK(hipStreamSynchronize(stream));
            if (nCand > candCap) { err = "candidate buffer overflow"; return MK_ERR_UNSUPPORTED; }
            if (nCand > 0) {
                // 6. exact ungapped scores
                PCHK(dScore.reserve(nCand));
                th = tb("diag_score", 24.0 * nCand, 0);
                hipLaunchKernelGGL(diag_score_kernel, dim3((nCand + 255) / 256), dim3(256), 0, stream, V, q0, dCand.p, nCand, dScore.p);
                te(th);
                PCHK(hipGetLastError());
                hCand.resize(nCand); hScore.resize(nCand);
                PCHK(hipMemcpyAsync(hCand.data(), dCand.p, nCand * sizeof(DCand), hipMemcpyDeviceToHost, stream));
                PCHK(hipMemcpyAsync(hScore.data(), dScore.p, nCand * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                PCHK(hipStreamSynchronize(stream));
                // host: group by query, then QueryMatcher's selection logic
                std::vector<uint32_t> start(q1 - q0 + 1, 0);
                for (uint32_t k = 0; k < nCand; k++) start[hCand[k].q + 1]++;
                for (uint32_t k = 0; k < q1 - q0; k++) start[k + 1] += start[k];
                std::vector<uint32_t> orderIdx(nCand);
                {
                    std::vector<uint32_t> cur(start.begin(), start.end() - 1);
                    for (uint32_t k = 0; k < nCand; k++) orderIdx[cur[hCand[k].q]++] = k;
                }
                for (uint32_t ql = 0; ql < q1 - q0; ql++) slot[ql + 1] = slot[ql] + std::min<uint32_t>(start[ql + 1] - start[ql], (uint32_t) maxHits);
                chunkHits.resize(slot[q1 - q0]);
#pragma omp parallel for schedule(dynamic, 64) private(perQuery)
                for (uint32_t ql = 0; ql < q1 - q0; ql++) {
                    if (start[ql + 1] == start[ql]) continue;
                    perQuery.clear();
                    for (uint32_t k = start[ql]; k < start[ql + 1]; k++) {
                        const DCand &c = hCand[orderIdx[k]];
                        perQuery.push_back(Cand{c.id, (uint16_t) c.diag, hScore[orderIdx[k]], c.ordinal});
                    }
                    std::sort(perQuery.begin(), perQuery.end(), [](const Cand &a, const Cand &b) { return a.ordinal < b.ordinal; });
                    const uint32_t q = q0 + ql;
                    int self = 0;
                    {   // only used on the saturated path; cheap enough to always have the inputs at hand
                        int n255 = 0;
                        for (const Cand &c : perQuery) n255 += c.score >= 255;
                        if (n255 >= maxHits) {
                            const int L = (int) (qOff[q + 1] - qOff[q]);
                            self = self_score(ungMat, qRes.data() + qOff[q], qCorrHost + qOff[q], L);
                        }
                    }
                    chunkCnt[ql] = (uint32_t) select_hits(perQuery, binCount, maxHits, P.min_ungapped_score, self, chunkHits.data() + slot[ql]);
                }
            }
        }
        for (uint32_t ql = 0; ql < q1 - q0; ql++) {
            outOff[(size_t) q0 + ql + 1] = outOff[(size_t) q0 + ql] + chunkCnt[ql];
            if (chunkCnt[ql]) outHits.insert(outHits.end(), chunkHits.begin() + slot[ql], chunkHits.begin() + slot[ql] + chunkCnt[ql]);
        }
        q0 = q1;
    }
    (void) tOff; (void) kmerMat;
    return MK_OK;
}

}  // namespace mk
