// metaeuk_amd/csrc/mk_prefilter.hip -- the k-mer prefilter on gfx950.
// Replaces, for a whole batch of queries at once:
//   KmerGenerator::generateKmerList / calculateArrayProduct   M/src/prefiltering/KmerGenerator.cpp:107-216
//   QueryMatcher::match (index probes + gather)               M/src/prefiltering/QueryMatcher.cpp:213-346
//   CacheFriendlyOperations::findDuplicates (double hits)     M/src/prefiltering/CacheFriendlyOperations.cpp:185-274
//   UngappedAlignment::computeScores                          M/src/prefiltering/UngappedAlignment.cpp:331-362
//   keepMaxScoreElementOnly / threshold / getResult / sort    M/src/prefiltering/QueryMatcher.cpp:149-209
//
// Two front ends produce the CANDIDATES of a chunk of queries (the (query, target, diagonal) triples that survive
// the double-diagonal rule; the candidates of a (query, target) pair are contiguous and in arrival order):
//
//  A. stream_kernel  (every query whose index hits fit a 64 K-hit region -- practically all ORF fragments)
//     persistent workgroups pull queries from a counter; each owns one hit region in HBM that it rewrites query after
//     query (it stays in L2 / Infinity Cache).  The waves take the k-mer starts of the query, most expensive first,
//     enumerate their similar k-mers (mk_enum.hpp: product order of the reference, no searches), probe the index and
//     append 8-byte hit records target | diagonal | start | ordinal to the region: the arrival order is in the record.
//     Two LDS bitmaps tell which targets were hit more than once -- only those (and single hits with diagonal low byte 0)
//     can satisfy the double-diagonal rule; pass 2 streams the region back, keeps those survivors, sorts them in LDS by
//     (target, arrival) with a bitonic network and evaluates the sequential 8-bit-diagonal rule of findDuplicates as a
//     neighbour test + short backward walk; only the resulting candidates leave the CU.  A query with more survivors
//     than the LDS sort holds is processed in target classes, one pass each.  Four shapes (one wave per query for the
//     tiny fragments, 4 and 16 waves for the larger ones); the host picks the tier from the exact similar-k-mer count
//     of the query (kmer_count_kernel); a query that overflows its region is retried by the next larger tier in the same
//     stream, and by path B after the largest.  One enumeration, one probe of the index per similar k-mer.
//
//  B. global path   (everything else: very long queries, queries that overflow the largest region, databases with
//     more than 2^22 targets)
//     probe_kernel<COUNT> -> exclusive scan -> probe_kernel<GATHER> (one 64-bit record per index hit: query | target | low
//     diagonal byte | arrival number within the query, written query-major) -> per-query segmented radix sort over the target
//     bits (hipcub; stable, so a pair's records stay in arrival order) -> double_hit_count_kernel -> block scan ->
//     double_hit_emit_kernel (the rule evaluated twice instead of flag + select: candidates come out ordered).
//
// Common back end per chunk, everything in HBM, the host sees only the final hit lists:
//     diag_score_kernel   exact ungapped score of every candidate
//     keep_kernel         best diagonal per target (first maximum in arrival order), >= --min-ungapped-score,
//                         per-query counts; queries that reach --max-seqs are flagged for the exact host
//                         tie-order logic (mk::select_hits), everything else is final
//     radix sort + emit   hits ordered by (query, score desc, target asc) -> compact mk_hit array, DMA'd to its
//                         final place in the batch's pinned result block
#include "mk_prefilter.hpp"
#include "mk_host.hpp"
#include "mk_kernels.hpp"
#include "mk_enum.hpp"
#include "mk_profile.hpp"
#include "mk_kmer7.hpp"
#include "mk_segsort.hpp"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <type_traits>

namespace mk {

namespace {

constexpr int WAVE = 64;
constexpr int N3 = 8000;

// short launches of this stage run beside the persistent workgroups of the other one, whose older waves win the CU's issue arbitration
#ifndef MK_HELPER_PRIO
#define MK_HELPER_PRIO 3         // (profiles/r03_search_tuning.txt: the helpers of a config-2 step 210 -> 95 ms)
#endif
__device__ __forceinline__ void helper_prio() { if (MK_HELPER_PRIO) __builtin_amdgcn_s_setprio(MK_HELPER_PRIO); }

// The lanes of a wave that hold the same key act as one group: fn(mask of the group, its first lane) runs in every member.  The candidates
// of a query are neighbours, so a per-query counter sees one atomic per wave instead of 64 (which the L2 would serialise).
template <typename F>
__device__ __forceinline__ void wave_grouped(bool active, uint32_t key, F &&fn) {
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long) todo) - 1;
        const uint32_t k = (uint32_t) __shfl((int) key, leader, 64);
        const bool mine = active && key == k;
        const unsigned long long grp = __ballot(mine);
        if (mine) { fn(grp, leader); active = false; }
        todo &= ~grp;
    }
}

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

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 32-bit finaliser (MurmurHash3's fmix32): the partition functions of the per-query kernels -- target class, bitmap bucket, subset, sub-class --
// take different bit ranges of differently seeded mixes, so that they are independent of each other (a class must spread evenly over the
// buckets, a subset over the sub-classes); single multiplicative hashes scaled to a range were visibly correlated (profiles/r04_wide_kernel.txt)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
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

// candidates of a chunk (structure of arrays; q = chunk-local query index)
struct CandArrays { uint32_t *q; uint32_t *id; uint32_t *ordinal; uint16_t *diag; int32_t *score; };

// =====================================================================================================
//  B. global path
// =====================================================================================================
struct ProbeArgs {
    PrefilterDeviceView V;
    uint64_t pos_begin, pos_end;      // residue range (of the view) handled by this launch
    uint32_t q_first;                 // first view query of the range
    uint32_t seq_bits;                // group = (q - q_first) << seq_bits | target
    uint32_t hit_bits;                // record = group << (8 + hit_bits) | (diagonal & 255) << hit_bits | arrival number of the hit within its query
    uint32_t *hit_count;              // [pos] (COUNT: written; GATHER: exclusive prefix, read)
    uint32_t *kmer_count;             // [pos] statistics
    uint64_t *keys; uint8_t *diag_hi; // GATHER outputs: one 8-byte record per index hit + the diagonal's high byte
    uint8_t *list_start = nullptr;    // GATHER, overflow path only: 1 at the first hit of every k-mer's index list (the buffer arithmetic of
                                      // QueryMatcher::match works on whole lists)
    const uint32_t *order = nullptr;  // the launch's i-th wave takes k-mer start pos_begin + order[i] (null: pos_begin + i); n_order of them
    uint64_t n_order = 0;
};

// slot and entry reads are one-touch random probes of tables far larger than L2; MK_NT_LOADS=1 marks them non-temporal so that
// they do not push the k-mer presence bitmap (8 MB, re-read all the time) out of the 4 MB L2s
#ifndef MK_NT_LOADS
#define MK_NT_LOADS 0
#endif
__device__ __forceinline__ uint64_t ld_probe(const uint64_t *p) {
#if MK_NT_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
// index list of a k-mer from its slot: length, first entry index, and the first entry itself when the slot holds it
// (slot of a longer list: first entry index in bits 0..39, length in bits 40..62 -- 2^40 entries per index, 2^23 per list)
struct KmerList { uint32_t size; uint64_t first; uint64_t ent0; bool isInline; };
__device__ __forceinline__ KmerList load_kmer_list(const uint64_t *slots, uint32_t kmer) {
    const uint64_t s = ld_probe(slots + kmer);
    KmerList l;
    l.isInline = (s >> 63) != 0;
    l.size = l.isInline ? 1u : (uint32_t) (s >> 40) & 0x7FFFFFu;
    l.first = s & 0xFFFFFFFFFFull;
    l.ent0 = s & 0x0000FFFFFFFFFFFFull;
    return l;
}
__device__ __forceinline__ uint64_t wave_read_lane64(uint64_t v, uint32_t srcLane) {
    return (uint64_t) enumk::wave_read_lane((uint32_t) v, srcLane) | ((uint64_t) enumk::wave_read_lane((uint32_t) (v >> 32), srcLane) << 32);
}
__device__ __forceinline__ bool kmer_present(const uint32_t *bits, uint32_t kmer) { return (bits[kmer >> 5] >> (kmer & 31u)) & 1u; }

// The index probes of a batch of U * 64 k-mers, STAGE BY STAGE: the U presence words, then the U slots, then the U first entries of the longer lists are
// issued together and waited for once.  (Rounds 2-5 wrote `if (has && present(kmer)) slot = load` inside the loop over u: the compiler turns that into
// nested divergent regions with a `s_waitcnt vmcnt(0)` inside each, so the chains bitmap -> slot -> entry of the U groups ran one AFTER the other -- the
// "probe groups" never were in flight together, which is also why U = 4 only made the kernel slower: profiles/r06_prefilter_probe_groups.txt.)
// kmer[u] is 0 where has[u] is false (every enumerator's contract): the presence word is read without a branch.
// Measured (one MI355X per call, base = the code of round 5 on the same box): third tier alone 274 -> 265 ms per step, beside the alignment stage 439 -> 420,
// config-2 step 834 -> 822 ms; with U = 4 the kernel stays slower (559 ms): the fabric's request rate, not the requests in flight per wave, bounds it.
// The wide kernel (16 waves of 64 registers, 35-47 spills already) keeps the probe and tail code of rounds 2-5 unless the two switches below are set: at
// 60 M proteins the staged probes change nothing (2 049 against 2 043 ms per pass of 20 000 fragments) and the merged tails cost 7 % (2 180 ms, 45 spills).
#ifndef MK_WIDE_STAGED_PROBES
#define MK_WIDE_STAGED_PROBES 0
#endif
#ifndef MK_WIDE_MERGED_TAILS
#define MK_WIDE_MERGED_TAILS 0
#endif
// what a lane without a probe reads: cell 0 of the slot table (the entry array of a small database may be empty)
// (as an index into the entry array, so that the load stays a global one)
__device__ __forceinline__ uint64_t idle_entry(const PrefilterDeviceView &V) { return (uint64_t) (V.kmer_slot - V.entries); }
template <int U, bool STAGED = true>
__device__ __forceinline__ void probe_lists(const PrefilterDeviceView &V, const uint32_t (&kmer)[U], const bool (&has)[U], uint32_t (&size)[U], uint64_t (&o0)[U],
                                            uint64_t (&ent0)[U], bool FIRST_ENTRIES = true) {
    if constexpr (!STAGED) {                                   // the form of rounds 2-5 (the wide kernel keeps it: see MK_WIDE_STAGED_PROBES)
#pragma unroll
    for (int u = 0; u < U; u++) {
        size[u] = 0; o0[u] = 0; ent0[u] = 0;
        bool inl = true;
        if (has[u] && kmer_present(V.kmer_bits, kmer[u])) { const KmerList l = load_kmer_list(V.kmer_slot, kmer[u]); o0[u] = l.first; size[u] = l.size; ent0[u] = l.ent0; inl = l.isInline; }
        if (FIRST_ENTRIES && !inl) ent0[u] = ld_probe(V.entries + o0[u]);
    }
    return;
    }
    // No branch around a load: a lane without a probe reads cell 0 of the table (one more request per wave instruction, a line every wave shares) and drops
    // the value.  Inside divergent regions the compiler cannot count the loads in flight at the join and waits for ALL of them before the next group's load.
    uint32_t word[U];
#pragma unroll
    for (int u = 0; u < U; u++) word[u] = V.kmer_bits[kmer[u] >> 5];
    uint64_t s[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const bool present = ((word[u] >> (kmer[u] & 31u)) & (has[u] ? 1u : 0u)) != 0u;
        const uint64_t v = ld_probe(V.kmer_slot + (present ? kmer[u] : 0u));
        s[u] = present ? v : 0ull;
    }
    bool far[U];
#pragma unroll
    for (int u = 0; u < U; u++) {                       // (a slot is never 0: bit 63 marks the inline entry, a longer list has a length)
        const bool isInline = (s[u] >> 63) != 0;
        size[u] = isInline ? 1u : (uint32_t) (s[u] >> 40) & 0x7FFFFFu;
        o0[u] = s[u] & 0xFFFFFFFFFFull;
        ent0[u] = s[u] & 0x0000FFFFFFFFFFFFull;
        far[u] = !isInline && s[u] != 0;
    }
    if (FIRST_ENTRIES) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            // (the lane mask passes through an empty asm: as a select on far[u] the compiler puts the load back under a branch)
            uint32_t m = far[u] ? 0xFFFFFFFFu : 0u;
            asm volatile("" : "+v"(m));
            const uint64_t mm = ((uint64_t) m << 32) | m;
            const uint64_t idle = idle_entry(V);
            const uint64_t v = ld_probe(V.entries + (idle ^ ((o0[u] ^ idle) & mm)));
            ent0[u] = (v & mm) | (ent0[u] & ~mm);
        }
    }
}

#ifndef MK_PROBE_U
#define MK_PROBE_U 4
#endif
constexpr int PROBE_U = MK_PROBE_U;        // 64-k-mer groups whose index probes are issued together (global path)

// the similar k-mers of a start read back from a list in HBM (profile queries), in windows of U * 64 like enumerate_position
template <int U, class F>
__device__ __forceinline__ uint32_t enumerate_list(const uint32_t *list, uint32_t count, int lane, F &&onBatch) {
    for (uint32_t base = 0; base < count; base += U * WAVE) {
        uint32_t kmer[U];
        bool has[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t x = base + (uint32_t) (u * WAVE + lane);
            has[u] = x < count;
            kmer[u] = has[u] ? list[x] : 0u;
        }
        if (!onBatch(kmer, has)) break;
    }
    return count;
}

template <bool GATHER, bool LIST = false>
__global__ __launch_bounds__(256) void probe_kernel(ProbeArgs A) {
    __shared__ enumk::EnumLds<PROBE_U> sE[4];
    __shared__ uint8_t sMark[4][WAVE];
    __builtin_amdgcn_s_setprio(3);                    // latency-bound waves issue ahead of the ALU-bound ones of the other stream
    const int w = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    const uint64_t item = (uint64_t) blockIdx.x * 4 + w;
    if (A.order && item >= A.n_order) return;
    const uint64_t p = A.pos_begin + (A.order ? (uint64_t) A.order[item] : item);
    if (p >= A.pos_end) return;
    const uint64_t rel = p - A.pos_begin;
    const int thr = (int) A.V.q_kmer_thr[p];
    if (thr < 0) {                                     // no k-mer starts here (X inside, or too close to the end)
        if (!GATHER && lane == 0) { A.hit_count[rel] = 0; A.kmer_count[rel] = 0; }
        return;
    }
    uint32_t qLocal = 0, iPos = 0, qFirstHit = 0;
    uint32_t hitBase = 0;
    if (GATHER) {
        const uint32_t q = find_query(A.V.q_off, A.V.n_queries, p);
        qLocal = q - A.q_first;
        iPos = (uint32_t) (p - A.V.q_off[q]);
        qFirstHit = A.hit_count[A.V.q_off[q] - A.pos_begin];
        hitBase = A.hit_count[rel];
    }
    uint32_t hits = 0;
    const auto onBatch = [&](const uint32_t (&kmer)[PROBE_U], const bool (&has)[PROBE_U]) -> bool {
            uint32_t size[PROBE_U];
            uint64_t o0[PROBE_U];
            uint64_t ent0[PROBE_U];
            probe_lists<PROBE_U>(A.V, kmer, has, size, o0, ent0, GATHER);
            if (!GATHER) {
#pragma unroll
                for (int u = 0; u < PROBE_U; u++) hits += size[u];
            } else {
                const auto put = [&](uint64_t ent, uint64_t at) {
                    const uint32_t seq = (uint32_t) ent;
                    const uint32_t posj = (uint32_t) (ent >> 32) & 0xFFFFu;
                    const uint32_t diag = (iPos - posj) & 0xFFFFu;
                    // low bits: the hit's arrival number within its query (the sort only looks at the group bits and is stable)
                    A.keys[at] = (((((uint64_t) qLocal << A.seq_bits) | seq) << 8 | (diag & 0xFFu)) << A.hit_bits) | (at - qFirstHit);
                    A.diag_hi[at] = (uint8_t) (diag >> 8);
                };
#pragma unroll
                for (int u = 0; u < PROBE_U; u++) {
                    const uint32_t incl = enumk::wave_incl_scan(size[u]);
                    const uint64_t dst = (uint64_t) hitBase + hits + (incl - size[u]);
                    if (size[u]) { put(ent0[u], dst); if (A.list_start) A.list_start[dst] = 1; }
                    // the rest of the longer lists, one entry per lane
                    enumk::wave_deal_tail(size[u] > 1 ? size[u] - 1 : 0u, lane, sMark[w], [&](uint32_t owner, uint32_t e, bool valid) {
                        const uint64_t oFirst = wave_read_lane64(o0[u], owner);
                        const uint32_t oLo = enumk::wave_read_lane((uint32_t) dst, owner), oHi = enumk::wave_read_lane((uint32_t) (dst >> 32), owner);
                        if (valid) put(A.V.entries[oFirst + e], (((uint64_t) oHi << 32) | oLo) + e);
                    });
                    hits += enumk::wave_last(incl);
                }
            }
            return true;
        };
    uint32_t kmers;
    if constexpr (LIST) {
        const uint64_t l0 = A.V.klist_off[p - A.V.klist_pos0], l1 = A.V.klist_off[p - A.V.klist_pos0 + 1];
        kmers = enumerate_list<PROBE_U>(A.V.klist + l0, (uint32_t) (l1 - l0), lane, onBatch);
    } else {
        kmers = enumk::enumerate_position<PROBE_U>(A.V, A.V.q_res + p, thr, lane, sE[w], onBatch);
    }
    if (!GATHER) {
        hits = wave_sum(hits);
        if (lane == 0) { A.hit_count[rel] = hits; A.kmer_count[rel] = kmers; }
    }
}

// totals of a range after the scan + the reference's per-query databaseHits capacity check
// (QueryMatcher.cpp:43,281-316: a query gathering >= 2*max(1e6,dbSize) entries takes the overflow path)
__global__ __launch_bounds__(256) void chunk_totals_kernel(const uint64_t *qOff, uint32_t qFirst, uint32_t nq, uint64_t posBegin, uint64_t nPos,
                                                          const uint32_t *scan, const uint32_t *lastCount, uint64_t maxDbMatches,
                                                          unsigned long long *totals /* [0]=hits [1]=first overflowing query+1 [3]=most hits of a query */) {
    const uint32_t ql = blockIdx.x * blockDim.x + threadIdx.x;
    if (ql == 0) totals[0] = (unsigned long long) scan[nPos - 1] + lastCount[0];
    if (ql >= nq) return;
    const uint64_t b = qOff[qFirst + ql] - posBegin, e = qOff[qFirst + ql + 1] - posBegin;
    if (e == b) return;
    const uint64_t endv = (e < nPos) ? scan[e] : (uint64_t) scan[nPos - 1] + lastCount[0];
    if (endv - scan[b] >= maxDbMatches) atomicMax(&totals[1], (unsigned long long) (qFirst + ql) + 1ull);
    atomicMax(&totals[3], (unsigned long long) (endv - scan[b]));
}

// The "k-mers per position" statistic is a sum of per-query quotients: accumulated in units of 2^-24 with INTEGER atomics, so that the sum does not depend on
// the order in which the waves arrive (a double atomicAdd made the printed statistic differ in its last digits from run to run)
constexpr double KPP_UNIT = 16777216.0;
__device__ __forceinline__ void kpp_add(double *cell, double v) { atomicAdd(reinterpret_cast<unsigned long long *>(cell), (unsigned long long) __double2ll_rn(v * KPP_UNIT)); }
static inline double kpp_value(const void *cell) { unsigned long long u; std::memcpy(&u, cell, 8); return (double) u / KPP_UNIT; }
// statistics: sum over the queries of a piece of (similar k-mers of the query) / (its length), in double, one atomic per wave
__global__ __launch_bounds__(256) void kmers_per_pos_kernel(const uint64_t *qOff, uint32_t qFirst, uint32_t nq, uint64_t posBegin, const uint32_t *kmerCount, double *out) {
    const uint32_t ql = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    if (ql < nq) {
        const uint64_t b = qOff[qFirst + ql], e = qOff[qFirst + ql + 1];
        unsigned long long sum = 0;
        for (uint64_t p = b; p < e; p++) sum += kmerCount[p - posBegin];
        if (e > b) v = (double) sum / (double) (e - b);
    }
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) __double_as_longlong(v), d, WAVE), hi = (uint32_t) __shfl_xor((int) (uint32_t) (__double_as_longlong(v) >> 32), d, WAVE);
        v += __longlong_as_double((long long) (((uint64_t) hi << 32) | lo));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0 && v != 0.0) kpp_add(out, v);
}

// first record of every query of a piece (+ the end): the exclusive hit scan at the query's first position
__global__ __launch_bounds__(256) void segment_offsets_kernel(const uint64_t *qOff, uint32_t qFirst, uint32_t nq, uint64_t posBegin, uint64_t nPos,
                                                             const uint32_t *hitScan, uint32_t nHits, uint32_t *seg) {
    const uint32_t ql = blockIdx.x * blockDim.x + threadIdx.x;
    if (ql > nq) return;
    const uint64_t p = ql == nq ? nPos : qOff[qFirst + ql] - posBegin;
    seg[ql] = p >= nPos ? nHits : hitScan[p];                       // (empty queries at the end of the piece start at the end)
}

// findDuplicates (computeTotalScore == false) on the (query,target)-sorted hit stream.
//   kept(t)    : low 8 bits of the diagonal equal those of the previous hit of the same (query,target);
//                the first hit of a target is compared with 0 (duplicateBitArray starts zeroed)
//   emitted(t) : kept(t) and the nearest earlier kept hit of the run has a different low byte (or none exists)
// The overflow path of QueryMatcher::match (QueryMatcher.cpp:281-316) cuts the hits of a query into segments (by arrival number) and runs
// findDuplicates on every segment with a cleared state: Segments makes a run end at a segment boundary as well.  start == nullptr: one segment.
struct Segments { const uint32_t *start; uint32_t n; uint32_t stop; /* hits from this arrival number on are dropped */ };
__device__ __forceinline__ uint32_t segment_of(const Segments &S, uint32_t arrival) {      // largest s with start[s] <= arrival
    uint32_t lo = 0, hi = S.n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (S.start[mid] <= arrival) lo = mid; else hi = mid; }
    return lo;
}
template <bool OVF = false>
__device__ __forceinline__ bool double_hit_emits(const uint64_t *rec, uint32_t hitBits, uint32_t t, const Segments S = Segments{nullptr, 1, 0xFFFFFFFFu}) {
    const uint64_t ordMask = (1ull << hitBits) - 1;
    const uint64_t r = rec[t] >> hitBits;
    const uint64_t group = r >> 8;
    const uint32_t lo = (uint32_t) r & 0xFFu;
    uint32_t seg = 0;
    if (OVF) {
        const uint32_t ord = (uint32_t) (rec[t] & ordMask);
        if (ord >= S.stop) return false;
        seg = segment_of(S, ord);
    }
    const auto same_run = [&](uint32_t u) -> bool {        // hit u - 1 belongs to the run of hit u (u > 0)
        if ((rec[u - 1] >> (hitBits + 8)) != group) return false;
        if (OVF) return segment_of(S, (uint32_t) (rec[u - 1] & ordMask)) == seg;
        return true;
    };
    const bool samePrev = t > 0 && same_run(t);
    const uint32_t prevLo = samePrev ? ((uint32_t) (rec[t - 1] >> hitBits) & 0xFFu) : 0u;
    if (lo != prevLo) return false;
    bool emit = true;
    if (samePrev) {
        uint32_t u = t - 1;
        while (true) {
            const uint32_t ulo = (uint32_t) (rec[u] >> hitBits) & 0xFFu;
            const bool uSame = u > 0 && same_run(u);
            const uint32_t uprev = uSame ? ((uint32_t) (rec[u - 1] >> hitBits) & 0xFFu) : 0u;
            if (ulo == uprev) { emit = ulo != lo; break; }
            if (!uSame) break;
            u--;
        }
    }
    return emit;
}

// segment boundaries of an overflowing query from the arrival numbers of its k-mer lists' first hits (ascending): the buffer arithmetic of
// QueryMatcher::match (:281-316) -- a list that does not fit (n + size >= cap) closes the segment before it; a list that alone fills the
// buffer stops the query (everything is dropped, :313-315,318-334).  One lane: a few hundred thousand lists, once per such query.
__global__ void overflow_segments_kernel(const uint32_t *listPos, uint32_t nLists, uint32_t nHits, uint64_t cap, uint32_t maxSeg, uint32_t *segStart,
                                         uint32_t *out /* [0] segments [1] stop (arrival number) [2] stopped [3] more segments than maxSeg */) {
    uint32_t nSeg = 1, stop = nHits, stopped = 0, tooMany = 0;
    uint64_t n = 0;
    segStart[0] = 0;
    for (uint32_t l = 0; l < nLists; l++) {
        const uint64_t sz = (uint64_t) (l + 1 < nLists ? listPos[l + 1] : nHits) - listPos[l];
        if (n + sz >= cap) {
            if (nSeg >= maxSeg) { tooMany = 1; break; }
            segStart[nSeg++] = listPos[l];
            n = 0;
            if (sz >= cap) { stopped = 1; stop = listPos[l]; break; }
        }
        n += sz;
    }
    out[0] = nSeg; out[1] = stop; out[2] = stopped; out[3] = tooMany;
}

// ordered compaction of the emitted records in two sweeps: candidates per 256-record block, (scan), then every block writes its
// candidates behind those of the blocks before it -- runs of one (query, target) stay contiguous and in arrival order
template <bool OVF = false>
__global__ __launch_bounds__(256) void double_hit_count_kernel(const uint64_t *rec, uint32_t hitBits, uint32_t n, uint32_t *blockCount, Segments S) {
    helper_prio();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = __syncthreads_count(t < n && double_hit_emits<OVF>(rec, hitBits, t, S));
    if (threadIdx.x == 0) blockCount[blockIdx.x] = (uint32_t) c;
}

// second sweep: emitted records -> candidate arrays (appended at `base`); qMap translates the range-local query index into the
// chunk-local one (null: qLocal + qAdd)
template <bool OVF = false>
__global__ __launch_bounds__(256) void double_hit_emit_kernel(const uint64_t *rec, const uint8_t *diagHi, const uint32_t *blockStart, uint32_t n, uint32_t seqBits,
                                                              uint32_t hitBits, const uint64_t *qOff, uint32_t qFirst, uint64_t posBegin, const uint32_t *hitScan,
                                                              const uint32_t *qMap, uint32_t qAdd, CandArrays C, uint32_t base, Segments S) {
    helper_prio();
    __shared__ uint32_t sWave[4];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    const bool emit = t < n && double_hit_emits<OVF>(rec, hitBits, t, S);
    const unsigned long long m = __ballot(emit);
    if (lane == 0) sWave[w] = (uint32_t) __popcll(m);
    __syncthreads();
    if (!emit) return;
    uint32_t c = blockStart[blockIdx.x] + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
    for (int k = 0; k < w; k++) c += sWave[k];
    const uint64_t r = rec[t];
    const uint32_t ord = (uint32_t) (r & ((1ull << hitBits) - 1));
    const uint64_t group = r >> (hitBits + 8);
    const uint32_t ql = (uint32_t) (group >> seqBits);
    C.q[base + c] = qMap ? qMap[ql] : ql + qAdd;
    C.id[base + c] = (uint32_t) (group & ((1ull << seqBits) - 1));
    C.ordinal[base + c] = ord;
    const uint32_t hit = hitScan[qOff[qFirst + ql] - posBegin] + ord;          // where the gather pass put this hit
    C.diag[base + c] = (uint16_t) (((uint32_t) diagHi[hit] << 8) | ((uint32_t) (r >> hitBits) & 0xFFu));
}

// compact copy of selected queries (the ones the fused kernel could not take) into a small batch for the global path
__global__ __launch_bounds__(256) void gather_queries_kernel(PrefilterDeviceView V, const uint32_t *srcQuery, const uint64_t *miniOff, uint32_t nMini, uint64_t total,
                                                             uint8_t *res, int16_t *kthr, int8_t *corr) {
    const uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= total) return;
    const uint32_t i = find_query(miniOff, nMini, r);
    const uint64_t src = V.q_off[srcQuery[i]] + (r - miniOff[i]);
    res[r] = V.q_res[src]; kthr[r] = V.q_kmer_thr[src]; corr[r] = V.q_corr[src];
}

// Exact number of similar k-mers of every k-mer start, without enumerating them: with the per-row score histograms the
// staircase sum over (first half, second half) collapses to sum_s hist0[s] * cum1[thr - s].  Summed per query (the host sizes
// the tier of each query with it) and kept per start as the work estimate of the per-query kernels.
// A wave walks 64 consecutive starts; for one start the lanes hold the score levels (two each: the histogram rows are read as two
// coalesced lines instead of ~80 dependent scalar loads per lane), a DPP sum folds them, and the counts of consecutive starts of one
// query leave the wave in one atomic.
__global__ __launch_bounds__(256) void kmer_count_kernel(PrefilterDeviceView V, uint64_t posBegin, uint64_t posEnd, uint32_t qFirst, uint32_t *perQuery,
                                                        uint16_t *perPos /* [p - posBegin]: similar k-mers of the start / 4, saturated (work estimate) */) {
    helper_prio();
    const int lane = threadIdx.x & (WAVE - 1);
    const uint64_t wave = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const uint64_t p0 = posBegin + wave * WAVE;
    if (p0 >= posEnd) return;
    const uint64_t p1 = min(p0 + (uint64_t) WAVE, posEnd);
    const int R = V.hist_range, lo = V.hist_lo;
    uint32_t q = find_query(V.q_off, V.n_queries, p0);
    uint64_t qEnd = V.q_off[q + 1];
    uint32_t acc = 0;
    for (uint64_t p = p0; p < p1; p++) {
        while (p >= qEnd) {                                    // (wave-uniform) next query: flush
            if (acc && lane == 0) atomicAdd(&perQuery[q - qFirst], acc);
            acc = 0; q++; qEnd = V.q_off[q + 1];
        }
        const int thr = (int) V.q_kmer_thr[p];
        uint32_t total = 0;
        if (thr >= 0) {
            const uint8_t *r = V.q_res + p;
            const uint32_t idx0 = r[0] + 20u * r[1] + 400u * r[3];
            const uint32_t idx1 = r[5] + 20u * r[8] + 400u * r[9];
            const uint16_t *h0 = V.hist3 + (size_t) idx0 * R, *c1 = V.cum3 + (size_t) idx1 * R;
            uint32_t part = 0;
            for (int k = lane; k < R; k += WAVE) {
                const uint32_t h = h0[k];
                const int x = thr - (lo + k) - lo;             // index of the cutoff in the cumulative row
                const uint32_t c = x <= 0 ? (uint32_t) N3 : (x >= R ? 0u : (uint32_t) c1[x]);
                part += h * c;
            }
            total = wave_sum(part);
        }
        if (lane == 0) perPos[p - posBegin] = (uint16_t) min((total + 3u) >> 2, 65535u);
        acc += total;
    }
    if (acc && lane == 0) atomicAdd(&perQuery[q - qFirst], acc);
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// Bitonic network over P keys in LDS (P a power of two) by the BLOCK threads of a workgroup.  Pair i of a stage belongs to thread i mod BLOCK, so a wave
// owns the same 64 pairs in every stage -- and while the partners are at most 64 apart those pairs lie inside ONE 128-key block: between two such stages
// the wave's own LDS order is enough; a workgroup barrier follows a stage only when it or the next one crosses the blocks (rounds 2-5 had a __syncthreads
// after each of the log P (log P + 1) / 2 stages: 21 for the 64 k-mer starts of a short fragment, 66 for 2 048 survivors -- now 1 and 11).
// The barrier stands BEHIND the stage's loop and is preceded by an explicit s_waitcnt: with __syncthreads() at the HEAD of the stage loop this compiler
// (ROCm 7.2) drops the wait for the ds_writes that reach the barrier over the back edge -- `s_barrier` with LDS writes of the previous stage still in
// flight, a few wrong candidates per 2 * 10^6 queries, different ones every run (profiles/r06_barrier_at_loop_head.txt).
__device__ __forceinline__ void workgroup_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
#ifndef MK_SORT_ALWAYS_BARRIER
#define MK_SORT_ALWAYS_BARRIER 0     // 1: a workgroup barrier after every stage (the form of rounds 2-5; experiments)
#endif
template <int BLOCK, bool DESCENDING, typename T>
__device__ __forceinline__ void lds_bitonic_sort(T *key, uint32_t P, int tid) {
    workgroup_sync_lds();                                        // the caller's keys
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = (uint32_t) tid; i < (P >> 1); i += BLOCK) {
                const uint32_t l = ((i & ~(j - 1u)) << 1) | (i & (j - 1u));
                const uint32_t r2 = l | j;
                const T x = key[l], y = key[r2];
                const bool up = (l & k) == 0;
                if ((DESCENDING ? x < y : x > y) == up) { key[l] = y; key[r2] = x; }
            }
            const uint32_t jNext = j > 1u ? j >> 1 : k;           // partner distance of the next stage (the first one of the merge of 2 k keys: k)
            const bool last = j == 1u && k == P;
            if (MK_SORT_ALWAYS_BARRIER || last || j > (uint32_t) WAVE || jNext > (uint32_t) WAVE) workgroup_sync_lds(); else wave_sync_lds();
        }
    }
}

// =====================================================================================================
//  A. per-query path: the index hits of a query pass through a workgroup-private region in HBM
// =====================================================================================================
// One enumeration, any query the region holds.  The workgroups are persistent (a fixed number per CU pull queries from a
// counter), so every workgroup owns ONE hit region for its whole life: the region is rewritten query after query and stays in
// L2 / Infinity Cache.  Pass 1 is the fused kernel's (waves take k-mer starts from a shared counter, enumerate, probe) except
// that a hit does not go to LDS: every probe batch reserves its slots in the region with one LDS atomic and writes 8-byte
// records target | diagonal | k-mer start | ordinal within the start -- the arrival order is IN the record, no chunk tables.
// LDS only holds the two "target bucket seen once / twice" bitmaps, which can therefore be much larger per hit than in the
// fused kernel.  Pass 2 streams the region back (coalesced), keeps the records whose target bucket was hit more than once
// (plus single hits with diagonal low byte 0) and sorts those survivors in LDS by (target, arrival rank); the rule and the
// emission are the fused kernel's.  When a query has more survivors than the LDS sort holds (the share of targets hit twice
// by chance grows with hits^2 / targets), the targets are split into hash classes and pass 2 runs once per class: the runs
// of a (query, target) pair stay contiguous, which is all the back end needs.
struct StreamArgs {
    PrefilterDeviceView V;
    const uint32_t *queries;                          // (global) query ids assigned to the tiers by tier_scatter_kernel, tier after tier ...
    const uint32_t *own_first, *own_count;            // ... this tier's share of that list (device: nothing of the assignment passes through the host)
    const uint32_t *prev_list;                        // ... followed by the queries that overflowed the next smaller tier (chunk-local ids,
    const uint32_t *prev_count;                       //     count known on the device only)
    uint32_t q_first;
    CandArrays C; uint32_t cand_cap;
    uint32_t *counters;                               // [0] candidates appended
    uint32_t *overflow_list; uint32_t *overflow_count;
    unsigned long long *totals;                       // [0] k-mers [1] index hits [2] k-mer starts (statistics / tier sizing) [3..6] workgroup time: pass 1, collect + sort, rule + emit, overflowed; [7] mean wave time of pass 1 [8] overflowed queries [9] class passes beyond the first
    uint32_t *work_counter;                           // next item of (own list ++ overflow list)
    uint64_t *pool;                                   // gridDim.x regions of CAPH records
    const uint16_t *pos_cost; uint64_t pos_begin;     // work estimate of every k-mer start of the chunk (kmer_count_kernel), [p - pos_begin]
};

#ifndef MK_STREAM_U
#define MK_STREAM_U 2
#endif
// shapes of the two streamed tiers: waves per workgroup, LDS sort size, bitmap bits, workgroups per CU
#ifndef MK_STREAM_U_A
#define MK_STREAM_U_A MK_STREAM_U   // probe groups of the third / the largest tier (experiments: tools/build_variant.sh)
#endif
#ifndef MK_STREAM_U_B
#define MK_STREAM_U_B MK_STREAM_U
#endif
#ifndef MK_STREAM_SURV_1
#define MK_STREAM_SURV_1 256       // second tier (4096 hits, one wave): LDS sort size and bitmap bits
#define MK_STREAM_MBITS_1 8192
#endif
// Third tier: 32 768 hits and 1 024 k-mer starts on 8 waves since round 5 (rounds 2-4: 8 192 hits, 512 starts, 4 waves).  The queries of the largest tier
// average 11 400 hits -- just beyond the old third tier -- and ran on 16 waves with 79 KB of LDS, ONE workgroup per CU beside the alignment stage.
// Measured on config 2, queued search, same box, two repetitions (profiles/r05_prefilter_tiers.txt; kernel ms per step, third + largest tier):
//   8 192 x 4 waves: 149 + 450, step 859 | 16 384 x 4: 345 + 250, 860 | 32 768 x 4 (sort 1 024 / 2 048): 471 + 173 / 514 + 129, 900-919 |
//   16 384 x 8 waves: 372 + 192, 839 | 32 768 x 8: 448 + 110, 841-845 | 65 536 x 8: 525 + 80, 885 | 32 768 x 8, sort 4 096: 545 + 104, 910
#ifndef MK_STREAM_CAP_A
#define MK_STREAM_CAP_A 32768      // hits and k-mer starts of a query of the third tier
#define MK_STREAM_MAXPOS_A 1024
#endif
#ifndef MK_STREAM_NW_A
#define MK_STREAM_NW_A 8
#define MK_STREAM_SURV_A 2048
#define MK_STREAM_MBITS_A 65536
#define MK_STREAM_WG_A 4
#endif
#ifndef MK_STREAM_NW_B
#define MK_STREAM_NW_B 16
#define MK_STREAM_SURV_B 4096
#define MK_STREAM_MBITS_B 131072
#define MK_STREAM_WG_B 2
#endif
#ifndef MK_STREAM_CAP_B
#define MK_STREAM_CAP_B 131072     // hits of a query held by the largest tier, and its k-mer starts: beyond, the query takes the global path
                                   // (65 536 until late in round 3: 0.1 % of config 2's queries then cost 9 % of the prefilter chain on the global path)
#endif
#ifndef MK_STREAM_MAXPOS_B
#define MK_STREAM_MAXPOS_B 2048
#endif
constexpr uint32_t REC_T_BITS = 22, REC_POS_BITS = 12, REC_ORD_BITS = 14;     // + 16 bits of diagonal = 64
constexpr int STREAM_MAX_CLASSES = 64;

template <int CAPH, int SURV, int MBITS, int MAXPOS, int NW, int U>
__global__ __launch_bounds__(NW * 64, 8) void stream_kernel(StreamArgs A) {       // 8 waves per SIMD: the kernel lives on memory-level parallelism
    constexpr int BLOCK = NW * WAVE;
    constexpr int LOG_MBITS = ilog2(MBITS);
    constexpr int RBITS = ilog2(CAPH);                // arrival rank of a hit within its query (< CAPH)
    constexpr int TSHIFT = 16 + RBITS;                // sort key: target << TSHIFT | rank << 16 | diagonal
    static_assert((CAPH & (CAPH - 1)) == 0 && (SURV & (SURV - 1)) == 0 && (MBITS & (MBITS - 1)) == 0, "powers of two");
    static_assert(MAXPOS <= (1 << REC_POS_BITS) && REC_T_BITS + TSHIFT <= 64, "record / key fields");
    // the enumerator's scratch (pass 1) and the sort keys (pass 2) share their LDS
    struct Pass1Lds { enumk::EnumLds<U> e[NW]; uint8_t mark[NW][WAVE]; };
    constexpr size_t RAW = sizeof(Pass1Lds) > sizeof(uint64_t) * SURV ? sizeof(Pass1Lds) : sizeof(uint64_t) * SURV;
    __shared__ __attribute__((aligned(16))) uint8_t sRaw[RAW];
    uint64_t *sKey = reinterpret_cast<uint64_t *>(sRaw);
    Pass1Lds &P1 = *reinterpret_cast<Pass1Lds *>(sRaw);
    __shared__ uint32_t sBm1[MBITS / 32], sBm2[MBITS / 32];
    __shared__ uint32_t sPosBase[MAXPOS];             // hits of a k-mer start, then their exclusive prefix
    constexpr int ORDER_N = MAXPOS < 64 ? 64 : MAXPOS;
    static_assert(RAW >= sizeof(uint32_t) * ORDER_N, "the start order is sorted in the shared scratch");
    __shared__ uint16_t sOrder[ORDER_N];              // k-mer starts by falling work estimate: the waves take the expensive ones first,
                                                      // so they reach the end of pass 1 together
    __shared__ uint32_t sNumOrder;                    // starts with k-mers
    __shared__ uint32_t sFlagBits[SURV / 32 + 2], sWordPrefix[SURV / 32 + 2];
    __shared__ uint32_t sWaveHits[NW], sWaveKmers[NW], sWavePos[NW];
    __shared__ uint32_t sClassCnt[STREAM_MAX_CLASSES];
    __shared__ uint32_t sNextPos, sUsed, sOverflow, sItem, sSurv, sEmitBase, sEmitCount, sClassMax;

    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, w = tid / WAVE, lane = tid & (WAVE - 1);
    uint64_t *region = A.pool + (size_t) blockIdx.x * CAPH;
    const uint32_t nOwn = A.own_count[0], ownFirst = A.own_first[0];
    const uint32_t nItems = nOwn + A.prev_count[0];
    constexpr uint64_t TMASK = (1ull << REC_T_BITS) - 1ull;
    // A record can take part in the double-diagonal rule when its target was hit twice (or its diagonal's low byte is 0: the first hit of a target is compared
    // with 0).  Pass 1 knows "the target's BUCKET was hit twice" (sBm2): with 1 000 ... 10 000 hits in 8 192 ... 65 536 buckets 95 % of the records that pass are
    // two targets sharing a bucket (round 6, MK_PREFILTER_DEBUG: 1.63e8 such records for 8.2e6 candidates in a chunk of the third tier) -- and they were
    // SORTED, in several class passes once they exceeded the LDS sort.  Second round, over those records only and with an independent hash: the memory of the
    // first bitmap (not needed after pass 1) becomes a once / twice pair of half the size; two hits of one target land in one bucket of both rounds, so no
    // record the rule needs is lost, and of the chance pairs of round one 2-12 % survive round two.
    constexpr int H2_BITS = LOG_MBITS - 1;
    uint32_t *sBm3 = sBm1, *sBm4 = sBm1 + MBITS / 64;
    const auto hit_twice = [&](uint64_t rec) -> bool {
        const uint32_t hb = ((uint32_t) (rec & TMASK) * 2654435761u) >> (32 - LOG_MBITS);
        return ((sBm2[hb >> 5] >> (hb & 31u)) & 1u) != 0u;
    };
    const auto bucket2 = [&](uint64_t rec) -> uint32_t { return mix32((uint32_t) (rec & TMASK) ^ 0x9E3779B9u) >> (32 - H2_BITS); };
    const auto survives = [&](uint64_t rec) -> bool {
        if (((uint32_t) (rec >> REC_T_BITS) & 0xFFu) == 0u) return true;
        if (!hit_twice(rec)) return false;
        const uint32_t h2 = bucket2(rec);
        return ((sBm4[h2 >> 5] >> (h2 & 31u)) & 1u) != 0u;
    };
    // target class of a record: a 24-bit hash scaled to the range (multiply and shift).  NOT `hash % n`: for operands the compiler can bound below 2^24
    // it emits a float-reciprocal division that returns remainder 0xFFFFFF where the true one is n - 1 (large numerators; n = 11, 44, 46, 57 ... on this
    // chip) -- such a record is in NO class.  Rounds 2-4 had that form here and in the wide kernel ("the lost subset"; root cause, device reproducer and
    // host model: tools/micro/urem24.hip, profiles/r05_lost_subset_root_cause.txt; tests/test_div24.py scans every kernel for the sequence)
    const auto class_of = [&](uint64_t rec, uint32_t n) -> uint32_t { return ((mix32((uint32_t) (rec & TMASK) ^ 0x85EBCA6Bu) >> 8) * n) >> 24; };
    for (;;) {
        workgroup_sync_lds();                             // the previous query's LDS is no longer read (a barrier at a loop head: with its own wait)
        if (tid == 0) sItem = atomicAdd(A.work_counter, 1u);
        __syncthreads();
        const uint32_t item = sItem;
        if (item >= nItems) break;
        const uint32_t q = item < nOwn ? A.queries[ownFirst + item] : A.q_first + A.prev_list[item - nOwn];
        const uint64_t qs = A.V.q_off[q];
        const int L = (int) (A.V.q_off[q + 1] - qs);
        const int nStart = L >= 10 ? L - 9 : 0;
        if (tid == 0) { sUsed = 0; sOverflow = nStart > MAXPOS ? 1u : 0u; sNextPos = 0; sSurv = 0; }
        for (int k = tid; k < MBITS / 32; k += BLOCK) { sBm1[k] = 0; sBm2[k] = 0; }
        const int nOrd = min(nStart, MAXPOS);
        uint32_t PO = WAVE;
        while ((int) PO < nOrd) PO <<= 1;
        for (int k = tid; k < nOrd; k += BLOCK) sPosBase[k] = 0;
        uint32_t *sOrdKey = reinterpret_cast<uint32_t *>(sRaw);      // cost << 12 | start (the scratch is free until pass 1 begins)
        for (int k = tid; k < (int) PO; k += BLOCK) sOrdKey[k] = k < nOrd ? ((uint32_t) A.pos_cost[qs - A.pos_begin + (uint64_t) k] << 12) | (uint32_t) k : 0u;
        if (tid == 0) sNumOrder = 0;
        __syncthreads();
        // descending bitonic sort of the k-mer starts by cost (starts without k-mers, cost 0, come last)
        lds_bitonic_sort<BLOCK, true>(sOrdKey, PO, tid);
        for (int k = tid; k < nOrd; k += BLOCK) {
            const uint32_t key = sOrdKey[k];
            sOrder[k] = (uint16_t) (key & 0xFFFu);
            if ((key >> 12) != 0u && (k + 1 == nOrd || (sOrdKey[k + 1] >> 12) == 0u)) sNumOrder = (uint32_t) k + 1u;
        }
        __syncthreads();
        const int nWork = (int) sNumOrder;
        const unsigned long long tStart = wall_clock64();

        // ---- pass 1: enumerate + probe; hits -> region, target buckets -> the two bitmaps
        uint32_t whits = 0, kmers = 0, npos = 0;
        bool dead = false;
        while (!dead) {
            uint32_t iu = 0;
            if (lane == 0) iu = atomicAdd(&sNextPos, 1u);
            const int io = __builtin_amdgcn_readfirstlane((int) iu);
            if (io >= nWork) break;                        // (what is left are starts without k-mers)
            const int i = (int) sOrder[io];
            const uint64_t p = qs + (uint64_t) i;
            const int thr = (int) A.V.q_kmer_thr[p];
            if (thr < 0) continue;
            if (*(volatile uint32_t *) &sOverflow) { dead = true; break; }
            npos++;
            uint32_t wcount = 0;                           // hits of this k-mer start so far
            kmers += enumk::enumerate_position<U>(A.V, A.V.q_res + p, thr, lane, P1.e[w],
                [&](const uint32_t (&kmer)[U], const bool (&has)[U]) -> bool {
                    uint32_t size[U], ex[U];
                    uint64_t o0[U];
                    uint64_t ent0[U];
                    probe_lists<U>(A.V, kmer, has, size, o0, ent0);          // (the first entries of the longer lists are still in flight below)
                    uint32_t totAll = 0;
#pragma unroll
                    for (int u = 0; u < U; u++) { const uint32_t incl = enumk::wave_incl_scan(size[u]); ex[u] = incl - size[u] + totAll; totAll += enumk::wave_last(incl); }
                    if (totAll == 0) return true;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&sUsed, totAll);           // the batch's slots: contiguous, hence coalesced stores
                    base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
                    if (base + totAll > (uint32_t) CAPH || wcount + totAll > (1u << REC_ORD_BITS)) { dead = true; if (lane == 0) sOverflow = 1; return false; }
                    const auto put = [&](uint64_t ent, uint32_t rel) {
                        const uint32_t tgt = (uint32_t) ent;
                        const uint32_t diag = ((uint32_t) i - ((uint32_t) (ent >> 32) & 0xFFFFu)) & 0xFFFFu;
                        region[base + rel] = (uint64_t) tgt | ((uint64_t) diag << REC_T_BITS) | ((uint64_t) (uint32_t) i << (REC_T_BITS + 16)) |
                                             ((uint64_t) (wcount + rel) << (REC_T_BITS + 16 + REC_POS_BITS));
                        const uint32_t hb = (tgt * 2654435761u) >> (32 - LOG_MBITS);
                        const uint32_t bit = 1u << (hb & 31u);
                        if (atomicOr(&sBm1[hb >> 5], bit) & bit) atomicOr(&sBm2[hb >> 5], bit);
                    };
                    // the further entries of the longer lists, one per lane: the first window's loads go out before the first entries are stored
                    uint32_t rem[U], oLo[U], oHi[U];
#pragma unroll
                    for (int u = 0; u < U; u++) { rem[u] = size[u] > 1 ? size[u] - 1 : 0u; oLo[u] = (uint32_t) o0[u]; oHi[u] = (uint32_t) (o0[u] >> 32); }
                    enumk::TailDeal<U> D;
                    D.init(size, rem, ex, totAll);
                    const auto tail_of = [&](uint32_t tbase, uint64_t &ent, uint32_t &rel) -> bool {
                        uint32_t id, e;
                        bool valid;
                        D.window(tbase, rem, lane, P1.mark[w], id, e, valid);
                        const uint64_t oFirst = (uint64_t) enumk::TailDeal<U>::pick(oLo, id) | ((uint64_t) enumk::TailDeal<U>::pick(oHi, id) << 32);
                        rel = enumk::TailDeal<U>::pick(ex, id) + e;
                        ent = ld_probe(A.V.entries + (valid ? oFirst + e : idle_entry(A.V)));      // (no branch around the load: see probe_lists)
                        return valid;
                    };
                    uint64_t tEnt = 0;
                    uint32_t tRel = 0;
                    bool tValid = false;
                    if (D.total) tValid = tail_of(0, tEnt, tRel);
#pragma unroll
                    for (int u = 0; u < U; u++) if (size[u]) put(ent0[u], ex[u]);
                    if (tValid) put(tEnt, tRel);
                    for (uint32_t tbase = WAVE; tbase < D.total; tbase += WAVE) if (tail_of(tbase, tEnt, tRel)) put(tEnt, tRel);
                    wcount += totAll;
                    return true;
                });
            if (lane == 0) sPosBase[i] = wcount;
            whits += wcount;
        }
        if (lane == 0) { sWaveHits[w] = whits; sWaveKmers[w] = kmers; sWavePos[w] = npos; atomicAdd(&A.totals[7], (wall_clock64() - tStart) / NW); }
        __syncthreads();                                   // (also orders the region stores before the reads of pass 2)
        const unsigned long long tGather = wall_clock64();
        if (sOverflow) {
            if (tid == 0) { A.overflow_list[atomicAdd(A.overflow_count, 1u)] = q - A.q_first; atomicAdd(&A.totals[6], tGather - tStart); atomicAdd(&A.totals[8], 1ull); }
            continue;
        }
        if (tid == 0) {
            uint32_t hits = 0, km = 0, np = 0;
            for (int k = 0; k < NW; k++) { hits += sWaveHits[k]; km += sWaveKmers[k]; np += sWavePos[k]; }
            atomicAdd(&A.totals[0], (unsigned long long) km);
            atomicAdd(&A.totals[1], (unsigned long long) hits);
            atomicAdd(&A.totals[2], (unsigned long long) np);
        }
        const uint32_t used = sUsed;
        if (used == 0) continue;
        // ---- second round of the target filter (see `survives`): the first bitmap's memory, cleared, takes the bucket pairs of the records that passed round one
        for (int k = tid; k < MBITS / 32; k += BLOCK) sBm1[k] = 0;
        __syncthreads();
        for (uint32_t s = (uint32_t) tid; s < used; s += BLOCK) {
            const uint64_t rec = region[s];
            if (hit_twice(rec)) {
                const uint32_t h2 = bucket2(rec);
                const uint32_t bit = 1u << (h2 & 31u);
                if (atomicOr(&sBm3[h2 >> 5], bit) & bit) atomicOr(&sBm4[h2 >> 5], bit);
            }
        }
        __syncthreads();
        // arrival rank of a hit = hits of the earlier k-mer starts + its ordinal
        if (w == 0) {
            const uint32_t perLane = ((uint32_t) nStart + WAVE - 1) / WAVE;
            const uint32_t b = min((uint32_t) nStart, (uint32_t) lane * perLane), e = min((uint32_t) nStart, b + perLane);
            uint32_t sum = 0;
            for (uint32_t k = b; k < e; k++) sum += sPosBase[k];
            uint32_t run = enumk::wave_incl_scan(sum) - sum;
            for (uint32_t k = b; k < e; k++) { const uint32_t c = sPosBase[k]; sPosBase[k] = run; run += c; }
        }
        // ---- pass 2a: how many records survive the filter; target classes if they do not fit the LDS sort at once
        {
            uint32_t local = 0;
            for (uint32_t s = (uint32_t) tid; s < used; s += BLOCK) local += survives(region[s]) ? 1u : 0u;
            local = wave_sum(local);
            if (lane == 0 && local) atomicAdd(&sSurv, local);
        }
        __syncthreads();
        const uint32_t nSurvAll = sSurv;
        if (nSurvAll == 0) continue;
        if (tid == 0) atomicAdd(&A.totals[10], (unsigned long long) nSurvAll);     // (statistics: survivors of the bucket filter; [11]: candidates)
        uint32_t nClasses = 1;
        if (nSurvAll > (uint32_t) SURV) {
            nClasses = (nSurvAll + (uint32_t) (SURV * 3 / 4) - 1) / (uint32_t) (SURV * 3 / 4);
            for (;;) {
                if (nClasses > (uint32_t) STREAM_MAX_CLASSES) break;
                for (uint32_t k = (uint32_t) tid; k < nClasses; k += BLOCK) sClassCnt[k] = 0;
                if (tid == 0) sClassMax = 0;
                __syncthreads();
                for (uint32_t s = (uint32_t) tid; s < used; s += BLOCK) {
                    const uint64_t rec = region[s];
                    if (survives(rec)) atomicAdd(&sClassCnt[class_of(rec, nClasses)], 1u);
                }
                __syncthreads();
                for (uint32_t k = (uint32_t) tid; k < nClasses; k += BLOCK) atomicMax(&sClassMax, sClassCnt[k]);
                __syncthreads();
                if (sClassMax <= (uint32_t) SURV) break;
                nClasses += 1 + nClasses / 4;
                __syncthreads();
            }
            if (nClasses > (uint32_t) STREAM_MAX_CLASSES) {    // (an extreme query: the global path sorts it)
                if (tid == 0) { A.overflow_list[atomicAdd(A.overflow_count, 1u)] = q - A.q_first; atomicAdd(&A.totals[8], 1ull); }
                continue;
            }
            if (tid == 0) atomicAdd(&A.totals[9], (unsigned long long) (nClasses - 1));
        }
        unsigned long long tSortAcc = 0, tEmitAcc = 0;
        for (uint32_t cls = 0; cls < nClasses; cls++) {
            const unsigned long long tc0 = wall_clock64();
            __syncthreads();
            if (tid == 0) sSurv = 0;
            __syncthreads();
            // ---- pass 2b: survivors of this class -> LDS sort keys (any order: the key carries the arrival rank)
            for (uint32_t s0 = 0; s0 < used; s0 += BLOCK) {
                const uint32_t s = s0 + (uint32_t) tid;
                bool surv = false;
                uint64_t rec = 0;
                if (s < used) {
                    rec = region[s];
                    surv = survives(rec) && (nClasses == 1 || class_of(rec, nClasses) == cls);
                }
                const unsigned long long m = __ballot(surv);
                if (m == 0) continue;
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(&sSurv, (uint32_t) __popcll(m));
                wbase = (uint32_t) __builtin_amdgcn_readfirstlane((int) wbase);
                if (surv) {
                    const uint32_t pos = (uint32_t) (rec >> (REC_T_BITS + 16)) & ((1u << REC_POS_BITS) - 1u);
                    const uint32_t rank = sPosBase[pos] + (uint32_t) (rec >> (REC_T_BITS + 16 + REC_POS_BITS));
                    sKey[wbase + (uint32_t) __popcll(m & ((1ull << lane) - 1ull))] =
                        ((rec & TMASK) << TSHIFT) | ((uint64_t) rank << 16) | ((rec >> REC_T_BITS) & 0xFFFFull);
                }
            }
            __syncthreads();
            const uint32_t nSurv = sSurv;
            if (nSurv == 0) continue;
            uint32_t P = WAVE;
            while (P < nSurv) P <<= 1;
            for (uint32_t s = nSurv + (uint32_t) tid; s < P; s += BLOCK) sKey[s] = ~0ull;
            __syncthreads();
            // ---- bitonic sort (keys are distinct: (target, rank) is unique)
            lds_bitonic_sort<BLOCK, false>(sKey, P, tid);
            const unsigned long long tc1 = wall_clock64();
            // ---- the double-diagonal rule on the target runs -> flag bits
            for (uint32_t t0 = 0; t0 < P; t0 += BLOCK) {
                const uint32_t t = t0 + (uint32_t) tid;
                bool emit = false;
                const uint64_t key = t < P ? sKey[t] : ~0ull;
                if (key != ~0ull) {
                    const uint64_t target = key >> TSHIFT;
                    const uint32_t lo = (uint32_t) key & 0xFFu;
                    const bool samePrev = t > 0 && (sKey[t - 1] >> TSHIFT) == target;
                    const uint32_t prevLo = samePrev ? ((uint32_t) sKey[t - 1] & 0xFFu) : 0u;
                    if (lo == prevLo) {
                        emit = true;
                        if (samePrev) {
                            uint32_t u = t - 1;
                            while (true) {
                                const uint32_t ulo = (uint32_t) sKey[u] & 0xFFu;
                                const bool uSame = u > 0 && (sKey[u - 1] >> TSHIFT) == target;
                                const uint32_t uprev = uSame ? ((uint32_t) sKey[u - 1] & 0xFFu) : 0u;
                                if (ulo == uprev) { emit = (ulo != lo); break; }
                                if (!uSame) break;
                                u--;
                            }
                        }
                    }
                }
                const unsigned long long m = __ballot(emit);
                if (lane == 0 && t < P) { sFlagBits[t >> 5] = (uint32_t) m; sFlagBits[(t >> 5) + 1] = (uint32_t) (m >> 32); }
            }
            __syncthreads();
            const uint32_t nWords = P >> 5;
            if (w == 0) {
                const uint32_t perLane = (nWords + WAVE - 1) / WAVE;
                const uint32_t b = min(nWords, (uint32_t) lane * perLane), e = min(nWords, b + perLane);
                uint32_t sum = 0;
                for (uint32_t k = b; k < e; k++) sum += (uint32_t) __popc(sFlagBits[k]);
                uint32_t total;
                uint32_t run = wave_excl_scan(sum, total);
                for (uint32_t k = b; k < e; k++) { sWordPrefix[k] = run; run += (uint32_t) __popc(sFlagBits[k]); }
                if (lane == 0) { sEmitBase = total ? atomicAdd(&A.counters[0], total) : 0u; sEmitCount = total; if (total) atomicAdd(&A.totals[11], (unsigned long long) total); }
            }
            __syncthreads();
            const uint32_t nEmit = sEmitCount, ebase = sEmitBase;
            if (nEmit != 0 && (unsigned long long) ebase + nEmit <= (unsigned long long) A.cand_cap) {   // else: the host sees counters[0] > cap and retries
                for (uint32_t t = (uint32_t) tid; t < P; t += BLOCK) {
                    const uint32_t word = sFlagBits[t >> 5];
                    if (!((word >> (t & 31u)) & 1u)) continue;
                    const uint32_t dst = ebase + sWordPrefix[t >> 5] + (uint32_t) __popc(word & ((1u << (t & 31u)) - 1u));
                    const uint64_t key = sKey[t];
                    A.C.q[dst] = q - A.q_first;
                    A.C.id[dst] = (uint32_t) (key >> TSHIFT);
                    A.C.ordinal[dst] = (uint32_t) (key >> 16) & ((1u << RBITS) - 1u);
                    A.C.diag[dst] = (uint16_t) key;
                }
            }
            tSortAcc += tc1 - tc0; tEmitAcc += wall_clock64() - tc1;
        }
        if (tid == 0) { atomicAdd(&A.totals[3], tGather - tStart); atomicAdd(&A.totals[4], tSortAcc); atomicAdd(&A.totals[5], tEmitAcc); }
    }
}

// Tiers of the per-query path: region size (hits), waves per workgroup, most k-mer starts, persistent workgroups per CU.  The two
// small tiers are one-wave workgroups (no barriers; most ORF fragments are tiny), the larger ones spread the k-mer starts of a
// query over 4 / 16 waves.

// ---- the tier of every query of a chunk, on the device (round 5) -----------------------------------------------------------------
// Rounds 2-4 brought the per-query k-mer counts of kmer_count_kernel to the host, chose the tiers there and sent the lists back: 21 ms of
// host time per config-2 step and a synchronisation per chunk between two launches of the prefilter's own chain -- which is the critical
// path of the queued search.  Now: classify (tier from the expected index hits and the number of k-mer starts, 64 size classes inside a
// tier so that the persistent workgroups take the large queries first) -> scan of the (tier, class) histogram -> scatter.  The kernels
// of the tiers read their share of the list from `info`; the host sees the counts and the global path's list after the tier kernels,
// with the counters it waits for anyway.
constexpr int TIER_CLS = 64, TIER_BINS = (4 + 1) * TIER_CLS;       // tier 4 = the global path's list (chunk-local ids)
struct TierPlan {
    const uint32_t *qk; const uint64_t *q_off; uint32_t q0, nqc;
    double limit[4]; int maxpos[4]; int nTiersUsed, firstTier;
    uint32_t *hist, *cursor;        // [TIER_BINS]
    uint16_t *bin;                  // per query: tier * 64 + class, 0xFFFF = no k-mer, no hits
    uint32_t *list, *fallback;      // [nqc] each
    uint32_t *info;                 // [0..3] first entry of tier t in `list` [4..7] its count [8] queries for the global path
    double *kpp;                    // statistics: sum over the queries of k-mers / length (null: not wanted)
};
__global__ __launch_bounds__(256) void tier_classify_kernel(TierPlan T) {
    helper_prio();
    __shared__ uint32_t sHist[TIER_BINS];
    for (int k = threadIdx.x; k < TIER_BINS; k += 256) sHist[k] = 0;
    __syncthreads();
    const uint32_t ql = blockIdx.x * 256u + threadIdx.x;
    double v = 0.0;
    if (ql < T.nqc) {
        const uint32_t kmers = T.qk[ql];
        const uint32_t L = (uint32_t) (T.q_off[(size_t) T.q0 + ql + 1] - T.q_off[(size_t) T.q0 + ql]);
        uint32_t bin = 0xFFFFu;
        if (kmers != 0u) {
            const double km = (double) kmers;
            const int npos = (int) L - 9;
            int t = 0;
            while (t < T.nTiersUsed && (km > T.limit[t] || npos > T.maxpos[t])) t++;
            if (t == T.nTiersUsed || t < T.firstTier) bin = 4u * TIER_CLS;
            else {
                const double scale = 63.0 / (T.limit[t] > 1.0 ? T.limit[t] : 1.0);
                const int c = (int) (km * scale);
                bin = (uint32_t) t * TIER_CLS + (uint32_t) (63 - (c < 63 ? c : 63));
            }
            atomicAdd(&sHist[bin], 1u);
            if (L) v = km / (double) L;
        }
        T.bin[ql] = (uint16_t) bin;
    }
    if (T.kpp) {
#pragma unroll
        for (int d = WAVE / 2; d >= 1; d >>= 1) {
            const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) __double_as_longlong(v), d, WAVE), hi = (uint32_t) __shfl_xor((int) (uint32_t) (__double_as_longlong(v) >> 32), d, WAVE);
            v += __longlong_as_double((long long) (((uint64_t) hi << 32) | lo));
        }
        if ((threadIdx.x & (WAVE - 1)) == 0 && v != 0.0) kpp_add(T.kpp, v);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < TIER_BINS; k += 256) if (sHist[k]) atomicAdd(&T.hist[k], sHist[k]);
}
__global__ __launch_bounds__(64) void tier_scan_kernel(TierPlan T) {       // one wave: 320 bins, five per lane
    helper_prio();
    const int lane = threadIdx.x;
    uint32_t h[5], pre[5], sum = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) { h[k] = T.hist[lane * 5 + k]; sum += h[k]; T.hist[lane * 5 + k] = 0; }      // (cleared for the next chunk)
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { const uint32_t o = (uint32_t) __shfl_up((int) incl, d, WAVE); if (lane >= d) incl += o; }
    uint32_t run = incl - sum;
#pragma unroll
    for (int k = 0; k < 5; k++) { pre[k] = run; T.cursor[lane * 5 + k] = run; run += h[k]; }
    // first entry of tier t = the prefix at bin 64 t, held by lane (64 t) / 5 in slot (64 t) % 5
    const uint32_t f0 = (uint32_t) __shfl((int) pre[0], 0, WAVE), f1 = (uint32_t) __shfl((int) pre[4], 12, WAVE), f2 = (uint32_t) __shfl((int) pre[3], 25, WAVE),
                   f3 = (uint32_t) __shfl((int) pre[2], 38, WAVE), f4 = (uint32_t) __shfl((int) pre[1], 51, WAVE);
    const uint32_t total = (uint32_t) __shfl((int) incl, WAVE - 1, WAVE);
    if (lane == 0) {
        T.info[0] = f0; T.info[1] = f1; T.info[2] = f2; T.info[3] = f3;
        T.info[4] = f1 - f0; T.info[5] = f2 - f1; T.info[6] = f3 - f2; T.info[7] = f4 - f3; T.info[8] = total - f4;
        T.info[9] = f4;                                                    // base of the global path's cursor (tier_scatter_kernel subtracts it)
    }
}
__global__ __launch_bounds__(256) void tier_scatter_kernel(TierPlan T) {
    helper_prio();
    const uint32_t ql = blockIdx.x * 256u + threadIdx.x;
    if (ql >= T.nqc) return;
    const uint32_t bin = T.bin[ql];
    if (bin == 0xFFFFu) return;
    const uint32_t at = atomicAdd(&T.cursor[bin], 1u);
    if (bin >= 4u * TIER_CLS) T.fallback[at - T.info[9]] = ql; else T.list[at] = T.q0 + ql;
}

struct FusedTier { int cap; int waves; int maxpos; int wgPerCu; };
constexpr int N_TIERS = 4;
// production tiers, then a miniature set (MK_PREFILTER_TIERS=tiny) with which small test inputs exercise every
// workgroup shape, the overflow hand-over, the class passes and the global path
const FusedTier TIERS[2 * N_TIERS] = {{2048, 1, 64, 28}, {4096, 1, 128, 24},
                                      {MK_STREAM_CAP_A, MK_STREAM_NW_A, MK_STREAM_MAXPOS_A, MK_STREAM_WG_A}, {MK_STREAM_CAP_B, MK_STREAM_NW_B, MK_STREAM_MAXPOS_B, MK_STREAM_WG_B},
                                      {256, 1, 32, 8}, {512, 1, 64, 8}, {1024, 4, 64, 4}, {4096, 8, 256, 4}};

//                                    region (hits)  LDS sort  bitmap bits  k-mer starts  waves  probe groups
void launch_stream(int tier, const StreamArgs &A, unsigned grid, hipStream_t stream) {
    switch (tier) {
        case 0: hipLaunchKernelGGL((stream_kernel<2048, 256, 8192, 64, 1, MK_STREAM_U>), dim3(grid), dim3(64), 0, stream, A); break;
        case 1: hipLaunchKernelGGL((stream_kernel<4096, MK_STREAM_SURV_1, MK_STREAM_MBITS_1, 128, 1, MK_STREAM_U>), dim3(grid), dim3(64), 0, stream, A); break;
        case 2: hipLaunchKernelGGL((stream_kernel<MK_STREAM_CAP_A, MK_STREAM_SURV_A, MK_STREAM_MBITS_A, MK_STREAM_MAXPOS_A, MK_STREAM_NW_A, MK_STREAM_U_A>), dim3(grid), dim3(64 * MK_STREAM_NW_A), 0, stream, A); break;
        case 3: hipLaunchKernelGGL((stream_kernel<MK_STREAM_CAP_B, MK_STREAM_SURV_B, MK_STREAM_MBITS_B, MK_STREAM_MAXPOS_B, MK_STREAM_NW_B, MK_STREAM_U_B>), dim3(grid), dim3(64 * MK_STREAM_NW_B), 0, stream, A); break;
        case 4: hipLaunchKernelGGL((stream_kernel<256, 64, 1024, 32, 1, 2>), dim3(grid), dim3(64), 0, stream, A); break;
        case 5: hipLaunchKernelGGL((stream_kernel<512, 64, 1024, 64, 1, 2>), dim3(grid), dim3(64), 0, stream, A); break;
        case 6: hipLaunchKernelGGL((stream_kernel<1024, 128, 2048, 64, 4, 2>), dim3(grid), dim3(256), 0, stream, A); break;
        default: hipLaunchKernelGGL((stream_kernel<4096, 256, 8192, 256, 8, 2>), dim3(grid), dim3(512), 0, stream, A); break;
    }
}

// =====================================================================================================
//  A'. the wide per-query kernel: any k-mer source, 27-bit targets, a million hits per query
// =====================================================================================================
// Round 4.  stream_kernel holds a query's hits in ONE region, filters them with two LDS bitmaps over the targets and sorts the
// survivors in LDS; that works while the hits are few against the bitmap (128 K bits).  A fragment searched against a UniRef50-scale
// database gathers 2*10^5 ... 10^6 index hits (k = 7), a profile query 10^5: the bitmaps saturate, every hit "survives", and until
// round 4 such queries took the sort-based global path -- two probe passes, the similar k-mers materialised, a device-wide radix sort.
// Here the region is PARTITIONED: a hit goes to one of NCLS target classes (a mixing hash of the target; one LDS atomic per hit gives its
// slot in the class), and pass 2 walks the classes in groups of at most GROUP_MAX records -- a class beyond that in subsets of its targets --
// with the two bitmaps, an exact per-target filter, the survivor sort and the double-diagonal rule of stream_kernel per group.  A target
// lives in one class, one subset, one sub-class, so the candidates of a (query, target) pair still leave the kernel contiguous and in arrival
// order.  Records are 8 + 4 bytes: target | diagonal | k-mer start, and the hit's ordinal within its start; the sort key target | arrival
// rank | diagonal gives the rank the 48 bits the target leaves.
// The k-mers come from the enumerator of the k = 6 table (MODE 0), from lists in HBM (MODE 1: profile queries), or from the in-wave 7-mer
// enumerator (MODE 2).  DESIGN.md 4.13; what was measured on the way: profiles/r04_wide_kernel.txt.
constexpr uint32_t W_T_BITS_MAX = 27;                          // sort key = target | arrival rank | diagonal (16 bits): the target field is as wide as the
                                                               // database needs (WideArgs::t_bits <= 27), the rank takes the other 48 - t_bits bits -- a
                                                               // query may gather 2^21 hits against 2^27 targets, 4 M against UniRef50's 60 M (26 bits)
constexpr int W_MODE_ENUM6 = 0, W_MODE_LIST = 1, W_MODE_ENUM7 = 2;
// Round 6: the target CLASS of a hit is the top bits of a BIJECTION of the t-bit target ids (multiply, xor-shift, multiply, xor-shift, all modulo 2^t:
// every step is invertible), so a sort key of a group that lies inside one class needs only the other t - log2(classes) bits to name the target -- and
// the arrival rank gets the bits the class number took: 2^28 hits per query at 60 M targets (26 bits, 64 classes) where the whole id left room for 2^22
// and sent a third of the fragments of a UniRef50-scale search to the sort-based global path.  wide_inv gives the id back when a candidate is emitted.
constexpr uint32_t WIDE_MUL_A = 0x9E3779B1u, WIDE_MUL_B = 0x7FEB352Du;
constexpr uint32_t inverse_mod_2_32(uint32_t a) { uint32_t x = a; for (int k = 0; k < 6; k++) x *= 2u - a * x; return x; }
static_assert(WIDE_MUL_A * inverse_mod_2_32(WIDE_MUL_A) == 1u && WIDE_MUL_B * inverse_mod_2_32(WIDE_MUL_B) == 1u, "odd multipliers");
__host__ __device__ __forceinline__ uint32_t wide_fwd(uint32_t x, uint32_t tBits) {
    const uint32_t m = (1u << tBits) - 1u, s = (tBits + 1u) >> 1;     // (x ^= x >> s with 2 s >= t undoes itself)
    x = (x * WIDE_MUL_A) & m; x ^= x >> s; x = (x * WIDE_MUL_B) & m; x ^= x >> s;
    return x;
}
__host__ __device__ __forceinline__ uint32_t wide_inv(uint32_t y, uint32_t tBits) {
    const uint32_t m = (1u << tBits) - 1u, s = (tBits + 1u) >> 1;
    y ^= y >> s; y = (y * inverse_mod_2_32(WIDE_MUL_B)) & m; y ^= y >> s; y = (y * inverse_mod_2_32(WIDE_MUL_A)) & m;
    return y;
}
struct WideArgs {
    PrefilterDeviceView V;
    const uint32_t *queries; uint32_t n_queries;      // view query ids, most expensive first (an ITEM of the work list: a query, or a part of one)
    const uint32_t *parts;                            // per item: r | log2(M) << 8 -- the workgroup takes the target classes c with (c & (M - 1)) == r (null: all)
    uint32_t *overflow_parts;                         // the part word of every entry of overflow_list
    uint32_t q_first;                                 // candidates carry q - q_first
    CandArrays C; uint32_t cand_cap;
    uint32_t *counters;                               // [0] candidates appended
    uint32_t *overflow_list; uint32_t *overflow_count;   // queries this kernel could not hold (view id - q_first): the global path takes them
    unsigned long long *totals;                       // as StreamArgs::totals; [10] sub-classes beyond the LDS sort (the host redoes the piece)
                                                      // [11] (a double) MODE 2: sum over the finished queries of similar k-mers / length (run statistics)
    uint32_t *work_counter;
    uint32_t *pool;                                   // gridDim.x regions of NCLS * cls_cap records of THREE dwords: target | diagonal | k-mer start (64 bits), the hit's
                                                      // ordinal within its start -- one 12-byte store per hit (rounds 4-5: an 8-byte and a 4-byte array; the
                                                      // scattered stores of pass 1 are half of its time at 60 M proteins, profiles/r06_config5.txt)
    uint32_t cls_cap;                                 // records per target class (NCLS * cls_cap <= 2^(48 - t_bits) * NCLS and < 2^31: the arrival rank of a hit
                                                      // shares the sort key with the target bits a class leaves open)
    uint32_t t_bits;                                  // bits of a target id (>= 20)
    uint32_t max_db_matches;                          // QueryMatcher's maxDbMatches (clamped to 2^32 - 1): from here on the reference's overflow path decides
    uint32_t max_log_m;                               // a part that fills a class is halved by its workgroup up to M = 2^max_log_m; beyond: overflow_list
    uint32_t one_class_hits;                          // a query with at least this many hits takes its classes one by one (2^(48 - t_bits); the tests force 0)
    const uint16_t *pos_cost; uint64_t pos_begin;     // MODE 0: work estimate of every k-mer start (kmer_count_kernel)
};

#ifndef MK_WIDE_SWEEP_ILP
#define MK_WIDE_SWEEP_ILP 4        // records in flight per thread in the sweeps of pass 2 (they are bound by memory latency)
#endif
template <int NCLS, int GROUP_MAX, int SURV, int MBITS, int MAXPOS, int NW, int U, int MODE>
__global__ __launch_bounds__(NW * 64, 8) void wide_kernel(WideArgs A) {        // (8 waves per SIMD = 64 registers; a 128-register build ran 5 % faster with ONE workgroup
                                                                                //  per CU and lost to two: profiles/r04_wide_kernel.txt)
    constexpr int BLOCK = NW * WAVE;
    constexpr int LOG_MBITS = ilog2(MBITS), LOG_NCLS = ilog2(NCLS);
    static_assert((SURV & (SURV - 1)) == 0 && (MBITS & (MBITS - 1)) == 0 && (NCLS & (NCLS - 1)) == 0, "powers of two");
    static_assert(MAXPOS <= 4096, "record / key fields");
    static_assert(2 * (MBITS / 32) >= 2 * SURV, "the exact-target table of pass 2 lives in the bitmaps' memory");
    static_assert(GROUP_MAX >= SURV, "a group holds at least one LDS sort");
    using EnumScratch = typename std::conditional<MODE == W_MODE_ENUM7, enumk::Enum7Lds<U>, enumk::EnumLds<U>>::type;
    struct Pass1Lds { EnumScratch e[NW]; uint8_t mark[NW][WAVE]; };
    constexpr size_t RAW = sizeof(Pass1Lds) > sizeof(uint64_t) * SURV ? sizeof(Pass1Lds) : sizeof(uint64_t) * SURV;
    __shared__ __attribute__((aligned(16))) uint8_t sRaw[RAW];
    uint64_t *sKey = reinterpret_cast<uint64_t *>(sRaw);
    Pass1Lds &P1 = *reinterpret_cast<Pass1Lds *>(sRaw);
    __shared__ uint32_t sBm[2 * (MBITS / 32)];        // the two bitmaps; after a group's survivors are collected, the exact-target table
    uint32_t *sBm1 = sBm, *sBm2 = sBm + MBITS / 32;
    __shared__ uint32_t sPosBase[MAXPOS];
    constexpr int ORDER_N = MAXPOS < 64 ? 64 : MAXPOS;
    static_assert(RAW >= sizeof(uint32_t) * ORDER_N, "the start order is sorted in the shared scratch");
    __shared__ uint16_t sOrder[ORDER_N];
    __shared__ uint32_t sNumOrder;
    __shared__ uint32_t sFlagBits[SURV / 32 + 2], sWordPrefix[SURV / 32 + 2];
    __shared__ uint32_t sWaveHits[NW], sWaveKmers[NW], sWavePos[NW];
    __shared__ uint32_t sClsUsed[NCLS];               // records in every target class
    __shared__ uint32_t sSubCnt[STREAM_MAX_CLASSES];
    __shared__ uint32_t sNextPos, sOverflow, sItem, sSurv, sEmitBase, sEmitCount, sSubMax;
    __shared__ uint32_t sRedo[2 * 8 + 2], sRedoN, sCurQ, sCurPart;     // halves of a part that filled a class, still to do (depth first: two per level)

    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, w = tid / WAVE, lane = tid & (WAVE - 1);
    if (tid == 0) sRedoN = 0;
    const uint32_t T_BITS = A.t_bits, CLS_CAP = A.cls_cap;
    const uint32_t CLS_SHIFT = T_BITS - (uint32_t) LOG_NCLS;          // class of a target = the top bits of its mapped id
    struct Rec12 { uint32_t lo, hi, ord; };
    Rec12 *region = reinterpret_cast<Rec12 *>(A.pool) + (size_t) blockIdx.x * NCLS * A.cls_cap;
    const uint64_t TMASK = (1ull << T_BITS) - 1ull;
    static_assert(LOG_MBITS <= 17, "bucket = the top bits of the mix, subset = its low 15 bits");
    const auto survives = [&](uint64_t rec) -> bool {
        const uint32_t hb = mix32((uint32_t) (rec & TMASK)) >> (32 - LOG_MBITS);
        return ((sBm2[hb >> 5] >> (hb & 31u)) & 1u) || ((uint32_t) (rec >> T_BITS) & 0xFFu) == 0u;
    };
    // (subset / sub-class of a target = hash bits scaled to the range: multiply and shift, no division)
    const auto sub_of = [&](uint64_t rec, uint32_t nSub) -> uint32_t { return ((mix32((uint32_t) (rec & TMASK) + 0x9E3779B9u) >> 8) * nSub) >> 24; };
    for (;;) {
        workgroup_sync_lds();                             // the previous query's LDS is no longer read (a barrier at a loop head: with its own wait)
        if (tid == 0) {
            if (sRedoN > 0) { sCurPart = sRedo[--sRedoN]; sItem = 1; }     // a half of the part that has just filled a class (the query stays: sCurQ)
            else {
                const uint32_t it = atomicAdd(A.work_counter, 1u);
                sItem = it < A.n_queries ? 1u : 0u;
                if (it < A.n_queries) { sCurQ = A.queries[it]; sCurPart = A.parts ? A.parts[it] : 0u; }
            }
        }
        __syncthreads();
        if (!sItem) break;
        const uint32_t q = sCurQ;
        // Round 6: a query with more hits than a region holds is taken by M = 2, 4, 8 ... workgroups at once: each enumerates and probes ALL k-mers (a
        // seventh of the kernel's time at 60 M proteins: profiles/r06_config5.txt) but keeps only the hits of ITS targets -- the log2(M) bits of the
        // mapped id below the class bits == r -- in the NCLS classes of its region, and runs pass 2 on them: the query's hits lie in M x NCLS classes
        // of the usual size (M times larger classes cost pass 2 quadratically: subsets x sweeps).  Arrival ranks count all hits, so the parts'
        // candidates are what one workgroup would have emitted, target by target.
        const uint32_t part = sCurPart;
        const uint32_t PART_R = part & 0xFFu, LOG_M = part >> 8, PART_MASK = (1u << LOG_M) - 1u;
        const uint32_t RES_BITS = T_BITS - (uint32_t) LOG_NCLS - LOG_M;  // what names a target inside its class and part
        const uint64_t qs = A.V.q_off[q];
        const int L = (int) (A.V.q_off[q + 1] - qs);
        const int span = A.V.kmer_size == 7 ? 11 : 10;
        const int nStart = L >= span ? L - span + 1 : 0;
        if (tid == 0) { sOverflow = nStart > MAXPOS ? 1u : 0u; sNextPos = 0; sSurv = 0; }
        for (int k = tid; k < NCLS; k += BLOCK) sClsUsed[k] = 0;
        const int nOrd = min(nStart, MAXPOS);
        uint32_t PO = WAVE;
        while ((int) PO < nOrd) PO <<= 1;
        for (int k = tid; k < nOrd; k += BLOCK) sPosBase[k] = 0;
        uint32_t *sOrdKey = reinterpret_cast<uint32_t *>(sRaw);      // cost << 12 | start (the scratch is free until pass 1 begins)
        for (int k = tid; k < (int) PO; k += BLOCK) {
            uint32_t cost = 0;
            if (k < nOrd) {
                if (MODE == W_MODE_LIST) {
                    const uint64_t pl = qs + (uint64_t) k - A.V.klist_pos0;
                    cost = (uint32_t) min((A.V.klist_off[pl + 1] - A.V.klist_off[pl] + 3ull) >> 2, (unsigned long long) 65535);
                } else if (MODE == W_MODE_ENUM7) {
                    // no count pass in front of this mode: the margin of the k-mer's own best similar k-mer over the threshold stands for the
                    // number of similar k-mers (it grows with it); 0 = no k-mer start here
                    const int thrK = (int) A.V.q_kmer_thr[qs + (uint64_t) k];
                    if (thrK >= 0) {
                        const uint8_t *r = A.V.q_res + qs + (uint64_t) k;
                        const int best = (int) A.V.score2[(size_t) (r[0] + 20u * r[1]) * 400u] + (int) A.V.score2[(size_t) (r[3] + 20u * r[5]) * 400u] +
                                         (int) A.V.score3[(size_t) (r[6] + 20u * r[9] + 400u * r[10]) * N3];
                        cost = best >= thrK ? (uint32_t) min(best - thrK + 1, 65535) : 0u;
                    }
                } else cost = A.pos_cost[qs - A.pos_begin + (uint64_t) k];
            }
            sOrdKey[k] = k < nOrd ? (cost << 12) | (uint32_t) k : 0u;
        }
        if (tid == 0) sNumOrder = 0;
        __syncthreads();
        // descending bitonic sort of the k-mer starts by cost (starts without k-mers, cost 0, come last)
        lds_bitonic_sort<BLOCK, true>(sOrdKey, PO, tid);
        for (int k = tid; k < nOrd; k += BLOCK) {
            const uint32_t key = sOrdKey[k];
            sOrder[k] = (uint16_t) (key & 0xFFFu);
            if ((key >> 12) != 0u && (k + 1 == nOrd || (sOrdKey[k + 1] >> 12) == 0u)) sNumOrder = (uint32_t) k + 1u;
        }
        __syncthreads();
        const int nWork = (int) sNumOrder;
        const unsigned long long tStart = wall_clock64();

        // ---- pass 1: k-mers -> index probes; every hit into the region of its target class
        uint32_t whits = 0, kmers = 0, npos = 0;
        bool dead = false;
        while (!dead) {
            uint32_t iu = 0;
            if (lane == 0) iu = atomicAdd(&sNextPos, 1u);
            const int io = __builtin_amdgcn_readfirstlane((int) iu);
            if (io >= nWork) break;                        // (what is left are starts without k-mers)
            const int i = (int) sOrder[io];
            const uint64_t p = qs + (uint64_t) i;
            const int thr = (int) A.V.q_kmer_thr[p];
            if (thr < 0) continue;
            if (*(volatile uint32_t *) &sOverflow) { dead = true; break; }
            npos++;
            uint32_t wcount = 0;                           // hits of this k-mer start so far
            const auto onBatch = [&](const uint32_t (&kmer)[U], const bool (&has)[U]) -> bool {
                    uint32_t size[U], ex[U];
                    uint64_t o0[U];
                    uint64_t ent0[U];
                    probe_lists<U, MK_WIDE_STAGED_PROBES != 0>(A.V, kmer, has, size, o0, ent0);
                    uint32_t totAll = 0;
#pragma unroll
                    for (int u = 0; u < U; u++) { const uint32_t incl = enumk::wave_incl_scan(size[u]); ex[u] = incl - size[u] + totAll; totAll += enumk::wave_last(incl); }
                    if (totAll == 0) return true;
                    bool over = false;
                    const auto put = [&](uint64_t ent, uint32_t rel) {
                        const uint32_t tgt = (uint32_t) ent;
                        const uint32_t diag = ((uint32_t) i - ((uint32_t) (ent >> 32) & 0xFFFFu)) & 0xFFFFu;
                        const uint32_t f = wide_fwd(tgt, T_BITS);                          // (a function of its own: the bitmap buckets of pass 2 must not follow the class)
                        if (((f >> RES_BITS) & PART_MASK) != PART_R) return;               // another part's target
                        const uint32_t cls = f >> CLS_SHIFT;
                        const uint32_t slot = atomicAdd(&sClsUsed[cls], 1u);
                        if (slot < CLS_CAP) {
                            const size_t at = (size_t) cls * CLS_CAP + slot;
                            const uint64_t rec = (uint64_t) tgt | ((uint64_t) diag << T_BITS) | ((uint64_t) (uint32_t) i << (T_BITS + 16u));
                            region[at] = Rec12{(uint32_t) rec, (uint32_t) (rec >> 32), wcount + rel};
                        } else over = true;
                    };
#if !MK_WIDE_MERGED_TAILS
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const uint32_t r0 = ex[u];
                        if (size[u]) put(ent0[u], r0);
                        enumk::wave_deal_tail(size[u] > 1 ? size[u] - 1 : 0u, lane, P1.mark[w], [&](uint32_t owner, uint32_t e, bool valid) {
                            const uint64_t oFirst = wave_read_lane64(o0[u], owner);
                            const uint32_t oR0 = enumk::wave_read_lane(r0, owner);
                            if (valid) put(ld_probe(A.V.entries + oFirst + e), oR0 + e);
                        });
                    }
#else
                    uint32_t rem[U], oLo[U], oHi[U];
#pragma unroll
                    for (int u = 0; u < U; u++) { rem[u] = size[u] > 1 ? size[u] - 1 : 0u; oLo[u] = (uint32_t) o0[u]; oHi[u] = (uint32_t) (o0[u] >> 32); }
                    enumk::TailDeal<U> D;
                    D.init(size, rem, ex, totAll);
                    const auto tail_of = [&](uint32_t tbase, uint64_t &ent, uint32_t &rel) -> bool {
                        uint32_t id, e;
                        bool valid;
                        D.window(tbase, rem, lane, P1.mark[w], id, e, valid);
                        const uint64_t oFirst = (uint64_t) enumk::TailDeal<U>::pick(oLo, id) | ((uint64_t) enumk::TailDeal<U>::pick(oHi, id) << 32);
                        rel = enumk::TailDeal<U>::pick(ex, id) + e;
                        ent = ld_probe(A.V.entries + (valid ? oFirst + e : idle_entry(A.V)));      // (no branch around the load: see probe_lists)
                        return valid;
                    };
                    uint64_t tEnt = 0;
                    uint32_t tRel = 0;
                    bool tValid = false;
                    if (D.total) tValid = tail_of(0, tEnt, tRel);              // the first window's loads go out before the first entries are stored
#pragma unroll
                    for (int u = 0; u < U; u++) if (size[u]) put(ent0[u], ex[u]);
                    if (tValid) put(tEnt, tRel);
                    for (uint32_t tbase = WAVE; tbase < D.total; tbase += WAVE) if (tail_of(tbase, tEnt, tRel)) put(tEnt, tRel);
#endif
                    wcount += totAll;
                    if (__ballot(over) != 0ull) { dead = true; if (lane == 0) sOverflow = 1; return false; }   // a class is full: the global path takes the query
                    return true;
                };
            if constexpr (MODE == W_MODE_LIST) {
                const uint64_t pl = p - A.V.klist_pos0;
                const uint64_t l0 = A.V.klist_off[pl], l1 = A.V.klist_off[pl + 1];
                kmers += enumerate_list<U>(A.V.klist + l0, (uint32_t) (l1 - l0), lane, onBatch);
            } else if constexpr (MODE == W_MODE_ENUM7) {
                kmers += enumk::enumerate7_position<U>(A.V, A.V.q_res + p, thr, lane, P1.e[w], onBatch);
            } else {
                kmers += enumk::enumerate_position<U>(A.V, A.V.q_res + p, thr, lane, P1.e[w], onBatch);
            }
            if (lane == 0) sPosBase[i] = wcount;
            whits += wcount;
        }
        if (lane == 0) { sWaveHits[w] = whits; sWaveKmers[w] = kmers; sWavePos[w] = npos; atomicAdd(&A.totals[7], (wall_clock64() - tStart) / NW); }
        __syncthreads();                                   // (also orders the region stores before the reads of pass 2)
        const unsigned long long tGather = wall_clock64();
        if (sOverflow) {
            if (tid == 0) {
                if (LOG_M < A.max_log_m) {
                    // the part is done again at once, as its two halves (one bit more of the mapped target id): the pass 1 just made is lost, but no
                    // launch with a handful of workgroups follows (three retry launches were a fifth of a 60 M-protein search: profiles/r06_config5.txt)
                    sRedo[sRedoN++] = (2u * PART_R + 1u) | ((LOG_M + 1u) << 8);
                    sRedo[sRedoN++] = (2u * PART_R) | ((LOG_M + 1u) << 8);
                    atomicAdd(&A.totals[13], 1ull);
                } else {
                    const uint32_t at = atomicAdd(A.overflow_count, 1u);
                    A.overflow_list[at] = q - A.q_first; A.overflow_parts[at] = part;
                }
                atomicAdd(&A.totals[6], tGather - tStart); atomicAdd(&A.totals[8], 1ull);
            }
            continue;
        }
        uint32_t hitsAll = 0;
        for (int k = 0; k < NW; k++) hitsAll += sWaveHits[k];
        // A query with at least 2 max(10^6, #targets) index hits does not fit the REFERENCE's per-thread buffer: QueryMatcher::match then works in segments
        // with a double-diagonal rule each (QueryMatcher.cpp:281-334), which only the global path restates (replay_overflow).  Since round 6 a query of that
        // size would fit here (2^28 hits), so it is handed over explicitly: every part sees the same total and emits nothing, the first one reports it.
        if (hitsAll >= A.max_db_matches) {
            if (tid == 0 && PART_R == 0u) {
                const uint32_t at = atomicAdd(A.overflow_count, 1u);
                A.overflow_list[at] = q - A.q_first; A.overflow_parts[at] = 0x80000000u;
            }
            continue;
        }
        if (tid == 0 && PART_R == 0u) {                    // (the statistics of a query are its first part's: every part enumerates everything)
            uint32_t km = 0, np = 0;
            for (int k = 0; k < NW; k++) { km += sWaveKmers[k]; np += sWavePos[k]; }
            atomicAdd(&A.totals[0], (unsigned long long) km);
            atomicAdd(&A.totals[1], (unsigned long long) hitsAll);
            atomicAdd(&A.totals[2], (unsigned long long) np);
            if (MODE == W_MODE_ENUM7 && km != 0 && L > 0) kpp_add(reinterpret_cast<double *>(&A.totals[11]), (double) km / (double) L);
        }
        if (hitsAll == 0) continue;
        // arrival rank of a hit = hits of the earlier k-mer starts + its ordinal
        if (w == 0) {
            const uint32_t perLane = ((uint32_t) nStart + WAVE - 1) / WAVE;
            const uint32_t b = min((uint32_t) nStart, (uint32_t) lane * perLane), e = min((uint32_t) nStart, b + perLane);
            uint32_t sum = 0;
            for (uint32_t k = b; k < e; k++) sum += sPosBase[k];
            uint32_t run = enumk::wave_incl_scan(sum) - sum;
            for (uint32_t k = b; k < e; k++) { const uint32_t c = sPosBase[k]; sPosBase[k] = run; run += c; }
        }
        __syncthreads();
        unsigned long long tSortAcc = 0, tEmitAcc = 0, tFilterAcc = 0;
        // ---- pass 2: the classes in groups of at most GROUP_MAX records (a single class may hold more)
        // a group of several small classes sorts on the whole (mapped) target id and leaves the rank 48 - T_BITS bits; a query with more hits than that
        // takes its classes one by one (the class number is then implied: RES_BITS of target, the rank gets LOG_NCLS bits more)
        const bool oneClassGroups = hitsAll >= A.one_class_hits;
        for (uint32_t c0 = 0; c0 < (uint32_t) NCLS; ) {
            uint32_t c1 = c0, recs = 0;
            while (c1 < (uint32_t) NCLS && (c1 == c0 || (!oneClassGroups && recs + sClsUsed[c1] <= (uint32_t) GROUP_MAX))) { recs += sClsUsed[c1]; c1++; }
            const uint32_t g0 = c0, g1 = c1;
            c0 = c1;
            if (recs == 0) continue;
            const bool single = g1 == g0 + 1u;
            const uint32_t FIELD_BITS = single ? RES_BITS : T_BITS, TSHIFT = 64u - FIELD_BITS;
            const uint32_t FIELD_MASK = (1u << FIELD_BITS) - 1u;
            const unsigned long long tc0 = wall_clock64();
            // the records of the group, 4 x BLOCK at a time: four loads in flight per thread (the sweeps are bound by memory latency), and every
            // thread of the workgroup takes part in every step
            const auto sweep = [&](auto &&fn) {
                for (uint32_t c = g0; c < g1; c++) {
                    const uint32_t nC = sClsUsed[c];
                    const size_t base = (size_t) c * CLS_CAP;
                    for (uint32_t s0 = 0; s0 < nC; s0 += (uint32_t) MK_WIDE_SWEEP_ILP * BLOCK) {
                        Rec12 rec[MK_WIDE_SWEEP_ILP];
                        bool ok[MK_WIDE_SWEEP_ILP];
#pragma unroll
                        for (uint32_t k = 0; k < (uint32_t) MK_WIDE_SWEEP_ILP; k++) {
                            const uint32_t s = s0 + k * BLOCK + (uint32_t) tid;
                            ok[k] = s < nC;
                            rec[k] = ok[k] ? region[base + s] : Rec12{0u, 0u, 0u};
                        }
#pragma unroll
                        for (uint32_t k = 0; k < (uint32_t) MK_WIDE_SWEEP_ILP; k++) fn(ok[k], rec[k].ord, (uint64_t) rec[k].lo | ((uint64_t) rec[k].hi << 32));
                    }
                }
            };
            // a class with more records than a group holds is taken in SUBSETS of its targets (a second hash): the bitmaps stay sparse
            const uint32_t nSets = (recs + (uint32_t) GROUP_MAX - 1u) / (uint32_t) GROUP_MAX;
            for (uint32_t set = 0; set < nSets; set++) {
            const auto in_set = [&](uint64_t rec) -> bool { return nSets == 1u || ((mix32((uint32_t) (rec & TMASK)) & 0x7FFFu) * nSets) >> 15 == set; };
            const unsigned long long tSet0 = wall_clock64();
            __syncthreads();                               // (the previous group's bitmaps and keys are no longer read)
            for (int k = tid; k < MBITS / 32; k += BLOCK) { sBm1[k] = 0; sBm2[k] = 0; }
            __syncthreads();
            // ---- 2a: target buckets hit once / twice
            sweep([&](bool valid, uint32_t, uint64_t rec) {
                if (!valid || !in_set(rec)) return;
                const uint32_t hb = mix32((uint32_t) (rec & TMASK)) >> (32 - LOG_MBITS);
                const uint32_t bit = 1u << (hb & 31u);
                if (atomicOr(&sBm1[hb >> 5], bit) & bit) atomicOr(&sBm2[hb >> 5], bit);
            });
            __syncthreads();
            tFilterAcc += wall_clock64() - tSet0;
            // ---- 2b: the survivors of sub-class `sub` of `nSub` -> LDS sort keys (any order: the key carries the arrival rank); sSurv counts
            // them all, the keys beyond the LDS sort are dropped (the caller looks at sSurv)
            const auto collect = [&](uint32_t sub, uint32_t nSub) {
                __syncthreads();
                if (tid == 0) sSurv = 0;
                __syncthreads();
                sweep([&](bool valid, uint32_t ord, uint64_t rec) {
                    const bool surv = valid && in_set(rec) && survives(rec) && (nSub == 1 || sub_of(rec, nSub) == sub);
                    const unsigned long long m = __ballot(surv);
                    if (m == 0) return;
                    uint32_t wbase = 0;
                    if (lane == 0) wbase = atomicAdd(&sSurv, (uint32_t) __popcll(m));
                    wbase = (uint32_t) __builtin_amdgcn_readfirstlane((int) wbase);
                    if (surv) {
                        const uint32_t slot = wbase + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
                        if (slot < (uint32_t) SURV) {
                            const uint32_t pos = (uint32_t) (rec >> (T_BITS + 16u)) & 0xFFFu;
                            const uint32_t rank = sPosBase[pos] + ord;
                            sKey[slot] = ((uint64_t) (wide_fwd((uint32_t) (rec & TMASK), T_BITS) & FIELD_MASK) << TSHIFT) | ((uint64_t) rank << 16) | ((rec >> T_BITS) & 0xFFFFull);
                        }
                    }
                });
                __syncthreads();
            };
            // optimistically as ONE sub-class (no counting sweep): nearly every group's survivors fit the LDS sort
            const unsigned long long tOpt0 = wall_clock64();
            collect(0u, 1u);
            const uint32_t nSurvAll = sSurv;
            if (nSurvAll == 0) continue;
            uint32_t nSub = 1;
            bool collected = true;                         // sKey holds the survivors of sub-class 0 of 1
            if (nSurvAll > (uint32_t) SURV) {
                // target sub-classes that fit
                collected = false;
                nSub = (nSurvAll + (uint32_t) (SURV * 3 / 4) - 1) / (uint32_t) (SURV * 3 / 4);
                for (;;) {
                    if (nSub > (uint32_t) STREAM_MAX_CLASSES) nSub = STREAM_MAX_CLASSES;
                    __syncthreads();
                    for (uint32_t k = (uint32_t) tid; k < nSub; k += BLOCK) sSubCnt[k] = 0;
                    if (tid == 0) sSubMax = 0;
                    __syncthreads();
                    sweep([&](bool valid, uint32_t, uint64_t rec) {
                        if (valid && in_set(rec) && survives(rec)) atomicAdd(&sSubCnt[sub_of(rec, nSub)], 1u);
                    });
                    __syncthreads();
                    for (uint32_t k = (uint32_t) tid; k < nSub; k += BLOCK) atomicMax(&sSubMax, sSubCnt[k]);
                    __syncthreads();
                    if (sSubMax <= (uint32_t) SURV || nSub == (uint32_t) STREAM_MAX_CLASSES) break;
                    nSub += 1 + nSub / 4;
                }
                if (tid == 0) atomicAdd(&A.totals[9], (unsigned long long) (nSub - 1));
            }
            // (GROUP_MAX <= STREAM_MAX_CLASSES * SURV / 2 for every shape in use: the sub-classes always fit; a sub-class that does not -- a
            //  single target with more than SURV hits in the group -- is cut at SURV below and flagged)
            for (uint32_t sub = 0; sub < nSub; sub++) {
                const unsigned long long ts0 = sub == 0 ? tOpt0 : wall_clock64();
                if (!collected) collect(sub, nSub);
                collected = false;
                // ---- 2c: the exact filter.  The bitmaps passed every record whose target BUCKET was hit twice; most of those are two targets
                // sharing a bucket.  A target with a single record can only matter when that record's low diagonal byte is 0 (the first hit of
                // a target is compared with 0), so: count the records per TARGET in an open-addressing table (in the bitmaps' memory: they are
                // done when there is one sub-class) and keep what the rule can use -- the sort below then has a few dozen keys, not thousands
                if (nSub == 1) {
                    constexpr uint32_t HT = 2u * (MBITS / 32);
                    constexpr uint32_t KPT = (SURV + BLOCK - 1) / BLOCK;
                    const uint32_t nAll = min(sSurv, (uint32_t) SURV);
                    uint64_t mine[KPT];
                    for (uint32_t k = (uint32_t) tid; k < HT; k += BLOCK) sBm[k] = 0;
                    __syncthreads();
#pragma unroll
                    for (uint32_t k = 0; k < KPT; k++) {
                        const uint32_t t = k * BLOCK + (uint32_t) tid;
                        mine[k] = t < nAll ? sKey[t] : ~0ull;
                        if (t < nAll) {
                            const uint32_t tg = (uint32_t) (mine[k] >> TSHIFT) + 1u;
                            uint32_t h = mix32(tg) % HT;
                            for (;;) {
                                const uint32_t old = atomicCAS(&sBm[h], 0u, tg);
                                if (old == 0u) break;
                                if ((old & 0x7FFFFFFFu) == tg) { if (!(old >> 31)) atomicOr(&sBm[h], 0x80000000u); break; }
                                h = h + 1u == HT ? 0u : h + 1u;
                            }
                        }
                    }
                    __syncthreads();
                    if (tid == 0) sSurv = 0;
                    __syncthreads();
#pragma unroll
                    for (uint32_t k = 0; k < KPT; k++) {
                        bool keep = false;
                        if (mine[k] != ~0ull) {
                            const uint32_t tg = (uint32_t) (mine[k] >> TSHIFT) + 1u;
                            uint32_t h = mix32(tg) % HT;
                            while ((sBm[h] & 0x7FFFFFFFu) != tg) h = h + 1u == HT ? 0u : h + 1u;
                            keep = (sBm[h] >> 31) != 0u || ((uint32_t) mine[k] & 0xFFu) == 0u;
                        }
                        const unsigned long long m = __ballot(keep);
                        if (m != 0ull) {
                            uint32_t wbase = 0;
                            if (lane == 0) wbase = atomicAdd(&sSurv, (uint32_t) __popcll(m));
                            wbase = (uint32_t) __builtin_amdgcn_readfirstlane((int) wbase);
                            if (keep) sKey[wbase + (uint32_t) __popcll(m & ((1ull << lane) - 1ull))] = mine[k];
                        }
                    }
                    __syncthreads();
                }
                __syncthreads();
                uint32_t nSurv = sSurv;
                if (nSurv > (uint32_t) SURV) { if (tid == 0) atomicAdd(&A.totals[10], 1ull); nSurv = SURV; }     // (reported by the host as an error: never seen)
                if (nSurv == 0) continue;
                uint32_t P = WAVE;
                while (P < nSurv) P <<= 1;
                for (uint32_t s = nSurv + (uint32_t) tid; s < P; s += BLOCK) sKey[s] = ~0ull;
                __syncthreads();
                // ---- bitonic sort (keys are distinct: (target, rank) is unique)
                lds_bitonic_sort<BLOCK, false>(sKey, P, tid);
                const unsigned long long ts1 = wall_clock64();
                // ---- the double-diagonal rule on the target runs -> flag bits
                for (uint32_t t0 = 0; t0 < P; t0 += BLOCK) {
                    const uint32_t t = t0 + (uint32_t) tid;
                    bool emit = false;
                    const uint64_t key = t < P ? sKey[t] : ~0ull;
                    if (key != ~0ull) {
                        const uint64_t target = key >> TSHIFT;
                        const uint32_t lo = (uint32_t) key & 0xFFu;
                        const bool samePrev = t > 0 && (sKey[t - 1] >> TSHIFT) == target;
                        const uint32_t prevLo = samePrev ? ((uint32_t) sKey[t - 1] & 0xFFu) : 0u;
                        if (lo == prevLo) {
                            emit = true;
                            if (samePrev) {
                                uint32_t u = t - 1;
                                while (true) {
                                    const uint32_t ulo = (uint32_t) sKey[u] & 0xFFu;
                                    const bool uSame = u > 0 && (sKey[u - 1] >> TSHIFT) == target;
                                    const uint32_t uprev = uSame ? ((uint32_t) sKey[u - 1] & 0xFFu) : 0u;
                                    if (ulo == uprev) { emit = (ulo != lo); break; }
                                    if (!uSame) break;
                                    u--;
                                }
                            }
                        }
                    }
                    const unsigned long long m = __ballot(emit);
                    if (lane == 0 && t < P) { sFlagBits[t >> 5] = (uint32_t) m; sFlagBits[(t >> 5) + 1] = (uint32_t) (m >> 32); }
                }
                __syncthreads();
                const uint32_t nWords = P >> 5;
                if (w == 0) {
                    const uint32_t perLane = (nWords + WAVE - 1) / WAVE;
                    const uint32_t b = min(nWords, (uint32_t) lane * perLane), e = min(nWords, b + perLane);
                    uint32_t sum = 0;
                    for (uint32_t k = b; k < e; k++) sum += (uint32_t) __popc(sFlagBits[k]);
                    uint32_t total;
                    uint32_t run = wave_excl_scan(sum, total);
                    for (uint32_t k = b; k < e; k++) { sWordPrefix[k] = run; run += (uint32_t) __popc(sFlagBits[k]); }
                    if (lane == 0) { sEmitBase = total ? atomicAdd(&A.counters[0], total) : 0u; sEmitCount = total; }
                }
                __syncthreads();
                const uint32_t nEmit = sEmitCount, ebase = sEmitBase;
                if (nEmit != 0 && (unsigned long long) ebase + nEmit <= (unsigned long long) A.cand_cap) {   // else: the host sees counters[0] > cap and retries
                    for (uint32_t t = (uint32_t) tid; t < P; t += BLOCK) {
                        const uint32_t word = sFlagBits[t >> 5];
                        if (!((word >> (t & 31u)) & 1u)) continue;
                        const uint32_t dst = ebase + sWordPrefix[t >> 5] + (uint32_t) __popc(word & ((1u << (t & 31u)) - 1u));
                        const uint64_t key = sKey[t];
                        A.C.q[dst] = q - A.q_first;
                        A.C.id[dst] = wide_inv((uint32_t) (key >> TSHIFT) | (single ? (g0 << CLS_SHIFT) | (PART_R << RES_BITS) : 0u), T_BITS);
                        A.C.ordinal[dst] = (uint32_t) ((key << FIELD_BITS) >> (FIELD_BITS + 16u));
                        A.C.diag[dst] = (uint16_t) key;
                    }
                }
                tSortAcc += ts1 - ts0; tEmitAcc += wall_clock64() - ts1;
            }
            }   // subsets of the group
            (void) tc0;
        }
        if (tid == 0) { atomicAdd(&A.totals[3], tGather - tStart); atomicAdd(&A.totals[4], tSortAcc); atomicAdd(&A.totals[5], tEmitAcc); atomicAdd(&A.totals[12], tFilterAcc); }
    }
}

// shapes of the wide kernel: production -- 16 waves, groups / subsets of 16 K records, a class as large as the rank bits the database leaves and
// the memory budget allow (2^21 ... 2^24 hits per query), with 64 target classes for sequence queries and 16 for profile queries (16 / 64 / 128 /
// 256 measured: profiles/r04_wide_kernel.txt) -- and a miniature (MK_PREFILTER_TIERS=tiny) with which small test inputs fill classes, span
// several groups, split classes into subsets and need sub-classes
struct WideShape { int clsCap /* 0: from the rank bits and the memory budget */, nCls, maxpos, waves, wgPerCu; };
const WideShape WIDE_SHAPES[3] = {{0, 16, 2048, 16, 2}, {192, 4, 64, 4, 4}, {0, 64, 2048, 16, 2}};
template <int MODE>
void launch_wide(int shape, const WideArgs &A, unsigned grid, hipStream_t stream) {
    if (shape == 0) hipLaunchKernelGGL((wide_kernel<16, 16384, 4096, 131072, 2048, 16, 2, MODE>), dim3(grid), dim3(1024), 0, stream, A);
    else if (shape == 2) hipLaunchKernelGGL((wide_kernel<64, 16384, 4096, 131072, 2048, 16, 2, MODE>), dim3(grid), dim3(1024), 0, stream, A);
    else hipLaunchKernelGGL((wide_kernel<4, 64, 64, 4096, 64, 4, 2, MODE>), dim3(grid), dim3(256), 0, stream, A);
}

// =====================================================================================================
//  common back end
// =====================================================================================================
// ungapped_score for a sequence query when both sequences are shorter than 32 768 (the common case: one real diagonal), with the three byte
// streams -- query residues, their int8 correction, target residues -- read as aligned dwords, four cells per load (a lane walks its own
// diagonal: byte loads cost the texture unit a transaction per lane and cell, and the kernel runs beside both other stages).  Same arithmetic
// as ungapped_on_diagonal_fn (mk_kernels.hpp), cell by cell.
__device__ __forceinline__ int ungapped_score_dwords(const int8_t *smat, const uint8_t *q, const int8_t *corr, uint32_t qLen, const uint8_t *t, uint32_t tLen, uint32_t d16) {
    const uint32_t wrapped = (0x10000u - d16) & 0xFFFFu;
    const uint32_t dist = wrapped < d16 ? wrapped : d16;                     // distanceFromDiagonal
    const int diagonal = (int) (short) (uint16_t) d16;
    uint32_t len = 0, q0 = 0, t0 = 0;
    if (diagonal >= 0 && dist < qLen) { len = tLen < qLen - dist ? tLen : qLen - dist; q0 = dist; }
    else if (diagonal < 0 && dist < tLen) { len = tLen - dist < qLen ? tLen - dist : qLen; t0 = dist; }
    const uint8_t *qp = q + q0, *tp = t + t0;
    const uint8_t *cp = reinterpret_cast<const uint8_t *>(corr) + q0;
    int score = 0, best = 0;
    const auto cell = [&](uint32_t qb, uint32_t cb, uint32_t tb) {
        const int curr = (int) (int8_t) (uint8_t) ((uint32_t) (uint8_t) smat[qb * 21u + tb] + cb);
        score = score + curr > 0 ? score + curr : 0;
        best = best > score ? best : score;
    };
    uint32_t k = 0;
    if (len >= 8) {
        const uint32_t aq = (uint32_t) (reinterpret_cast<uintptr_t>(qp) & 3u), ac = (uint32_t) (reinterpret_cast<uintptr_t>(cp) & 3u), at = (uint32_t) (reinterpret_cast<uintptr_t>(tp) & 3u);
        const uint32_t *qw = reinterpret_cast<const uint32_t *>(qp - aq), *cw = reinterpret_cast<const uint32_t *>(cp - ac), *tw = reinterpret_cast<const uint32_t *>(tp - at);
        uint32_t q0w = qw[0], c0w = cw[0], t0w = tw[0];
        // chunk c = bytes 4c .. 4c + 3 of a stream = bytes a .. a + 3 of the dword pair (w[c], w[c + 1]); the pair ends at stream byte 4c + 7 - a
        for (uint32_t c = 0; k + 8 <= len; c++, k += 4) {
            const uint32_t q1w = qw[c + 1], c1w = cw[c + 1], t1w = tw[c + 1];
            const uint32_t qv = __builtin_amdgcn_alignbyte(q1w, q0w, aq), cv = __builtin_amdgcn_alignbyte(c1w, c0w, ac), tv = __builtin_amdgcn_alignbyte(t1w, t0w, at);
            q0w = q1w; c0w = c1w; t0w = t1w;
#pragma unroll
            for (int b = 0; b < 4; b++) cell((qv >> (8 * b)) & 0xFFu, (cv >> (8 * b)) & 0xFFu, (tv >> (8 * b)) & 0xFFu);
        }
    }
    for (; k < len; k++) cell((uint32_t) qp[k], (uint32_t) cp[k], (uint32_t) tp[k]);
    return best;
}

// exact ungapped diagonal score of every candidate
__global__ __launch_bounds__(256) void diag_score_kernel(PrefilterDeviceView V, uint32_t qFirst, uint32_t n, CandArrays C) {
    helper_prio();
    __shared__ int8_t smat[21 * 21 + 3];
    for (int i = threadIdx.x; i < 21 * 21; i += blockDim.x) smat[i] = V.mat_ung[i];
    __syncthreads();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const uint32_t q = qFirst + C.q[c], id = C.id[c];
    const uint32_t d16 = (uint32_t) C.diag[c];
    const uint64_t qs = V.q_off[q], ts = V.t_off[id];
    const uint32_t qLen = (uint32_t) (V.q_off[q + 1] - qs), tLen = (uint32_t) (V.t_off[id + 1] - ts);
    const int best = V.p_aln ? ungapped_score_profile(V.p_aln + qs * PROFILE_ALN_STRIDE, qLen, V.t_masked + ts, tLen, d16)
                   : ((qLen < 32768u && tLen < 32768u) ? ungapped_score_dwords(smat, V.q_res + qs, V.q_corr + qs, qLen, V.t_masked + ts, tLen, d16)
                                                       : ungapped_score(smat, V.q_res + qs, V.q_corr + qs, qLen, V.t_masked + ts, tLen, d16));
    C.score[c] = best;
}

// keepMaxScoreElementOnly on the candidates (per (query,target) contiguous, in arrival order): a candidate survives
// when its clamped score is the maximum of its run and no earlier candidate of the run has the same clamped score.
// Survivors below --min-ungapped-score can never be reported (diagonalThr >= minDiagScoreThr) and are dropped here.
__global__ __launch_bounds__(256) void keep_kernel(CandArrays C, uint32_t n, int minDiag, uint32_t scoreMax, bool hostCut, uint8_t *kept, uint32_t *perQuery,
                                                   uint32_t *perQuery255 /* survivors at the clamp value 255 | flags: the host finishes the query (P255_*) */) {
    helper_prio();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int) (threadIdx.x & 63u);
    uint32_t q = 0;
    int s = 0;
    bool keep = false;
    if (c < n) {
        q = C.q[c];
        const uint32_t id = C.id[c];
        s = min(C.score[c], 255);
        keep = s >= minDiag;
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
        if (keep) {
            if (hostCut || C.ordinal[c] >= (1u << 26)) atomicOr(&perQuery255[q], 1u << 30);       // P255_HOST_IF_CUT: does not fit the cut key below
            if ((uint32_t) C.score[c] > scoreMax) atomicOr(&perQuery255[q], 1u << 31);             // P255_HOST: does not fit the report key
        }
    }
    wave_grouped(keep, q, [&](unsigned long long grp, int leader) { if (lane == leader) atomicAdd(&perQuery[q], (uint32_t) __popcll(grp)); });
    wave_grouped(keep && s == 255, q, [&](unsigned long long grp, int leader) { if (lane == leader) atomicAdd(&perQuery255[q], (uint32_t) __popcll(grp)); });
}

// ---- the tail of QueryMatcher::matchQuery per query (QueryMatcher.cpp:149-209 + getResult :117-125), without a device-wide sort
// Queries with at least --max-seqs surviving targets: the reference keeps the first max-seqs of them in the order (clamped score
// descending, hash bin of the target = id & (BINSIZE - 1), arrival) -- radixSortByScoreSize :498-523 over the bin-major element
// order of CacheFriendlyOperations.  That order as ONE 64-bit key per survivor: 255 - clamped | bin | arrival (the "cut key").  The
// reported order (score descending, target ascending) as another, which also carries the hit: inverted score | target | diagonal.
// The survivors of a query are gathered into its segment (counting sort by query: counts from keep_kernel, offsets from a scan,
// one atomic cursor per query); then whoever owns the segment -- a wave, a workgroup in LDS, a workgroup over HBM -- takes the
// max-seqs smallest cut keys (if the query reaches the cut), sorts the report keys and writes the query's hits at their final
// offset.  Only the case in which max-seqs survivors sit AT the clamp value (the threshold saturates and is rescaled by the self
// score, :163-170), arrival numbers >= 2^26 and scores beyond the key's score field stay on the host (flags in perQuery255).
constexpr uint32_t P255_HOST_IF_CUT = 1u << 30, P255_HOST = 1u << 31;
constexpr uint32_t FIN_WAVE_MAX = 64, FIN_SMALL_MAX = 512, FIN_LDS_MAX = 4096, FIN_HUGE_TILE = 8192;
__host__ __device__ __forceinline__ bool fin_to_host(uint32_t n, uint32_t p255, uint32_t maxHits) { return (p255 & P255_HOST) != 0u || (n >= maxHits && p255 >= maxHits); }
__host__ __device__ __forceinline__ uint32_t fin_score_max(uint32_t seqBits) { return (1u << (48u - seqBits < 31u ? 48u - seqBits : 31u)) - 1u; }

struct FinishArgs {
    const uint32_t *perQ, *perQ255;      // survivors per query, survivors at the clamp value | flags
    uint32_t nq, maxHits, seqBits;
    uint32_t *segOff, *outOff;           // per query: first survivor in the segment arrays, first hit in the output
    uint32_t *listW, *listS, *listB, *listH;     // queries by segment size: <= 64, <= 512, <= 4096, longer
    uint32_t *hugeTmp;                   // per listH entry: offset of its scratch (cut queries)
    uint32_t *blockSums;                 // [blocks][8]
    uint32_t *totals;                    // [0] hits [1] survivors in segments [2..5] list sizes [6] scratch keys [8] candidates of host queries [9] export cursor
};
constexpr int FIN_NV = 7;
__device__ __forceinline__ void fin_values(const FinishArgs &F, uint32_t q, uint32_t (&v)[FIN_NV]) {
#pragma unroll
    for (int k = 0; k < FIN_NV; k++) v[k] = 0;
    if (q >= F.nq) return;
    const uint32_t n = F.perQ[q];
    if (n == 0 || fin_to_host(n, F.perQ255[q], F.maxHits)) return;
    v[0] = min(n, F.maxHits);
    v[1] = n;
    v[2] = n <= FIN_WAVE_MAX ? 1u : 0u;
    v[3] = (n > FIN_WAVE_MAX && n <= FIN_SMALL_MAX) ? 1u : 0u;
    v[4] = (n > FIN_SMALL_MAX && n <= FIN_LDS_MAX) ? 1u : 0u;
    v[5] = n > FIN_LDS_MAX ? 1u : 0u;
    v[6] = (n > FIN_LDS_MAX && n >= F.maxHits) ? n : 0u;
}
// (small workgroups: these kernels run beside the persistent ones of both stages and must find room on a CU)
constexpr uint32_t FIN_SCAN = 256;
__global__ __launch_bounds__(FIN_SCAN) void finish_sums_kernel(FinishArgs F) {
    helper_prio();
    __shared__ uint32_t sm[FIN_NV * 16];
    uint32_t v[FIN_NV], excl[FIN_NV], total[FIN_NV];
    fin_values(F, blockIdx.x * FIN_SCAN + threadIdx.x, v);
    segsort::block_scan<FIN_NV>(v, excl, total, sm);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < FIN_NV; k++) F.blockSums[blockIdx.x * 8u + k] = total[k];
}
__global__ __launch_bounds__(FIN_SCAN) void finish_offsets_kernel(FinishArgs F) {
    helper_prio();
    __shared__ uint32_t sm[FIN_NV * 16];
    uint32_t v[FIN_NV], excl[FIN_NV], total[FIN_NV], base[FIN_NV];
#pragma unroll
    for (int k = 0; k < FIN_NV; k++) v[k] = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += FIN_SCAN)
#pragma unroll
        for (int k = 0; k < FIN_NV; k++) v[k] += F.blockSums[b * 8u + k];
    segsort::block_scan<FIN_NV>(v, excl, base, sm);
    const uint32_t q = blockIdx.x * FIN_SCAN + threadIdx.x;
    fin_values(F, q, v);
    segsort::block_scan<FIN_NV>(v, excl, total, sm);
    if (q < F.nq) {
        F.outOff[q] = base[0] + excl[0];
        F.segOff[q] = base[1] + excl[1];
        if (v[2]) F.listW[base[2] + excl[2]] = q;
        if (v[3]) F.listS[base[3] + excl[3]] = q;
        if (v[4]) F.listB[base[4] + excl[4]] = q;
        if (v[5]) { F.listH[base[5] + excl[5]] = q; F.hugeTmp[base[5] + excl[5]] = base[6] + excl[6]; }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < FIN_NV; k++) F.totals[k] = base[k] + total[k];
}

// survivors into the segments of their queries; candidates of host queries are counted
__global__ __launch_bounds__(256) void finish_scatter_kernel(CandArrays C, uint32_t n, const uint8_t *kept, FinishArgs F, uint32_t binMask, uint32_t *cursor,
                                                             uint64_t *cutKey, uint64_t *outKey) {
    helper_prio();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int) (threadIdx.x & 63u);
    bool host = false, put = false;
    uint32_t qv = 0, nQv = 0;
    if (c < n) {
        const uint32_t q = C.q[c], nQ = F.perQ[q];
        qv = q; nQv = nQ;
        if (nQ != 0 && fin_to_host(nQ, F.perQ255[q], F.maxHits)) host = true;
        else put = kept[c] != 0;
    }
    uint32_t pos = 0;
    wave_grouped(put, qv, [&](unsigned long long grp, int leader) {
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&cursor[qv], (uint32_t) __popcll(grp));
        pos = F.segOff[qv] + (uint32_t) __shfl((int) base, leader, 64) + (uint32_t) __popcll(grp & ((1ull << lane) - 1ull));
    });
    if (put) {
        const uint32_t score = (uint32_t) C.score[c], id = C.id[c];
        outKey[pos] = ((uint64_t) (fin_score_max(F.seqBits) - score) << (16u + F.seqBits)) | ((uint64_t) id << 16) | (uint64_t) C.diag[c];
        if (nQv >= F.maxHits) cutKey[pos] = ((uint64_t) (255u - min(score, 255u)) << 36) | ((uint64_t) (id & binMask) << 26) | (uint64_t) C.ordinal[c];
    }
    const unsigned long long m = __ballot(host);
    if (m != 0 && (threadIdx.x & 63u) == (uint32_t) (__ffsll((long long) m) - 1)) atomicAdd(&F.totals[8], (uint32_t) __popcll(m));
}

__device__ __forceinline__ mk_hit fin_hit(uint64_t key, uint32_t seqBits) {
    mk_hit h;
    h.seq_id = (uint32_t) (key >> 16) & ((1u << seqBits) - 1u);
    h.pref_score = (int32_t) (fin_score_max(seqBits) - (uint32_t) (key >> (16u + seqBits)));
    h.diagonal = (uint16_t) key; h.pad_ = 0;
    return h;
}

// segments of at most 64 survivors: one wave each, ranks by counting
__global__ __launch_bounds__(256) void finish_wave_kernel(FinishArgs F, uint32_t nItems, const uint64_t *cutKey, const uint64_t *outKey, mk_hit *out) {
    helper_prio();
    const uint32_t item = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (item >= nItems) return;
    const uint32_t q = F.listW[item], n = F.perQ[q], b = F.segOff[q];
    uint64_t key = lane < n ? outKey[b + lane] : ~0ull;
    if (n >= F.maxHits) {
        const uint64_t k1 = lane < n ? cutKey[b + lane] : ~0ull;
        if (segsort::wave_rank(k1, n) >= F.maxHits) key = ~0ull;
    }
    const uint32_t r = segsort::wave_rank(key, n);
    if (key != ~0ull) out[F.outOff[q] + r] = fin_hit(key, F.seqBits);
}

// longer segments: one workgroup each; in LDS up to TILE survivors, over HBM beyond
template <int THREADS, uint32_t TILE>
__global__ __launch_bounds__(THREADS) void finish_block_kernel(FinishArgs F, const uint32_t *list, const uint64_t *cutKey, uint64_t *outKey, uint64_t *scratch, mk_hit *out) {
    helper_prio();
    __shared__ uint64_t sK[TILE];
    __shared__ uint32_t sM;
    const uint32_t q = list[blockIdx.x], n = F.perQ[q], b = F.segOff[q];
    const bool cut = n >= F.maxHits;
    const uint32_t m = cut ? F.maxHits : n;
    mk_hit *dst = out + F.outOff[q];
    if (n <= TILE) {
        uint32_t P = segsort::pow2_at_least(n);
        if (cut) {
            for (uint32_t t = threadIdx.x; t < P; t += THREADS) sK[t] = t < n ? cutKey[b + t] : ~0ull;
            if (threadIdx.x == 0) sM = 0;
            __syncthreads();
            segsort::lds_sort<THREADS>(sK, P);
            const uint64_t last = sK[m - 1];
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < n; t += THREADS)
                if (cutKey[b + t] <= last) sK[atomicAdd(&sM, 1u)] = outKey[b + t];
            __syncthreads();
            P = segsort::pow2_at_least(m);
            for (uint32_t t = m + threadIdx.x; t < P; t += THREADS) sK[t] = ~0ull;
        } else {
            for (uint32_t t = threadIdx.x; t < P; t += THREADS) sK[t] = t < n ? outKey[b + t] : ~0ull;
        }
        __syncthreads();
        segsort::lds_sort<THREADS>(sK, P);
        for (uint32_t t = threadIdx.x; t < m; t += THREADS) dst[t] = fin_hit(sK[t], F.seqBits);
        return;
    }
    uint64_t *work = outKey + b;
    if (cut) {
        work = scratch + F.hugeTmp[blockIdx.x];
        for (uint32_t t = threadIdx.x; t < n; t += THREADS) work[t] = cutKey[b + t];
        if (threadIdx.x == 0) sM = 0;
        __syncthreads();
        segsort::global_sort<THREADS, TILE>(work, n, sK);
        const uint64_t last = work[m - 1];
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < n; t += THREADS)
            if (cutKey[b + t] <= last) work[atomicAdd(&sM, 1u)] = outKey[b + t];
        __syncthreads();
    }
    segsort::global_sort<THREADS, TILE>(work, m, sK);
    for (uint32_t t = threadIdx.x; t < m; t += THREADS) dst[t] = fin_hit(work[t], F.seqBits);
}

struct HostCand { uint32_t q, id, ordinal; uint16_t diag; int32_t score; };
// every candidate of the queries the host finishes (any order: the host sorts a query's candidates by arrival)
__global__ __launch_bounds__(256) void export_host_kernel(CandArrays C, uint32_t n, FinishArgs F, HostCand *out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    bool host = false;
    uint32_t q = 0;
    if (c < n) {
        q = C.q[c];
        const uint32_t nQ = F.perQ[q];
        host = nQ != 0 && fin_to_host(nQ, F.perQ255[q], F.maxHits);
    }
    const unsigned long long m = __ballot(host);
    if (m == 0) return;
    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t) (__ffsll((long long) m) - 1);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&F.totals[9], (uint32_t) __popcll(m));
    base = (uint32_t) __shfl((int) base, (int) leader, 64);
    if (!host) return;
    HostCand h;
    h.q = q; h.id = C.id[c]; h.ordinal = C.ordinal[c]; h.diag = C.diag[c]; h.score = C.score[c];
    out[base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull))] = h;
}

// exact ungapped self score of a query on diagonal 0 (QueryMatcher::rescoreHits, QueryMatcher.cpp:525-531);
// only needed when the score threshold saturates at 255
int self_score(const SubMat &ung, const uint8_t *q, const int8_t *corr, int L) {
    int8_t m[21 * 21];
    for (int a = 0; a < 21; a++) for (int b = 0; b < 21; b++) m[a * 21 + b] = (int8_t) ung.sub[a][b];
    return ungapped_score(m, q, corr, (uint32_t) L, q, (uint32_t) L, 0u);
}

// sizing state carried from one batch to the next (same database): index hits per similar k-mer, candidates per query
struct SizingMemo {
    const void *entries = nullptr; uint32_t nTargets = 0;
    double candPerQuery = 0;
    double wideHitsPerUnit = 0;               // wide kernel: index hits per similar k-mer (per k-mer START when the 7-mers are enumerated in the kernel), learned
    double hitsPerKmer[4] = {0, 0, 0, 0};     // per LDS tier (short fragments and long ORFs differ in composition)
    double margin[4] = {1.15, 1.15, 1.15, 1.15};
};
SizingMemo g_memo;
std::mutex g_memoMutex;        // two batches can be in the prefilter at once (the search engine's two prefilter threads): the memo is shared

}  // namespace

#define PCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return MK_ERR_DEVICE; } } while (0)
#define PNULL(p) do { if (!(p)) { err = "device scratch allocation failed (" #p ")"; return MK_ERR_DEVICE; } } while (0)

namespace {

constexpr int RC_CAND_OVERFLOW = 1000;            // internal: the chunk produced more candidates than the buffers hold

struct Ctx {
    hipStream_t stream; std::string *err; timed_begin_fn tb; timed_end_fn te; timed_set_fn ts;
    uint32_t seqBits; uint64_t maxDbMatches;
    int lastHitBits = 16;
    CandArrays C; uint32_t candCap;
    unsigned long long *dTotals, *hTotals;
    // host side of the batch, for the overflow path's replay (the rare query that fills the reference's databaseHits buffer)
    const std::vector<uint64_t> *qOffHost = nullptr; const std::vector<uint8_t> *qResHost = nullptr; const int8_t *qCorrHost = nullptr;
    std::function<const uint8_t *()> tMaskedHost; const std::vector<uint64_t> *tOffHost = nullptr; const SubMat *ungMat = nullptr;
    uint32_t chunkQ0 = 0;              // batch index of the chunk's first query (candidates carry chunk-local indices)
    PrefilterStats *stats = nullptr;   // run statistics (Prefiltering.cpp:889-904)
    bool statsKmers = false;           // ... the global path adds the similar k-mers per query too (no per-query count exists yet)
};

// ---- the overflow path of QueryMatcher::match, host part (QueryMatcher.cpp:281-334; oracle/mko_prefilter.c: overflow_device_model) ----
// cands = the diagonals the per-segment double-diagonal rule kept, ordered by (target, arrival number).  Per target the merges of the
// overflow events are replayed: after event 0 the list is segment 0's; at every later event the list + the event's segment goes through
// mergeDiagonalKeepScoredHitsDuplicates (backwards: an element stays when it is scored or its low diagonal byte differs from the element
// behind it; the output is reversed), is scored, and keepMaxElement leaves the first maximum (and zero-score elements behind it); the
// last segment is appended and mergeDiagonalDuplicates drops an element whose low byte equals its predecessor's.  The reference's ARRAY
// order -- ties at the --max-seqs cut follow it -- is rebuilt from (segment, arrival): an event reverses the array, so after event e it
// reads [C_e descending] ++ reverse(order before); the last segment follows ascending.  Survivors come back ordered by (target, that
// order) with `ordinal` = their rank in that order.
struct OvfCand { uint32_t id, ordinal; uint16_t diag; };
template <class ScoreFn>
void replay_overflow(const std::vector<OvfCand> &cands, const std::vector<uint32_t> &segStart, ScoreFn score_of, std::vector<OvfCand> &out) {
    struct El { uint32_t arrival, seg; uint16_t diag; uint8_t count; };
    const size_t events = segStart.size() - 1;
    const auto seg_of = [&](uint32_t arrival) { size_t lo = 0, hi = segStart.size(); while (hi - lo > 1) { const size_t mid = (lo + hi) >> 1; if (segStart[mid] <= arrival) lo = mid; else hi = mid; } return (uint32_t) lo; };
    // rank and direction of every segment in the final array
    std::vector<uint32_t> os(1, 0u);
    std::vector<uint8_t> od(1, (uint8_t) 0);
    for (size_t e = 1; e + 1 <= events; e++) {
        std::vector<uint32_t> ns(1, (uint32_t) e);
        std::vector<uint8_t> nd(1, (uint8_t) 1);
        for (size_t x = os.size(); x-- > 0;) { ns.push_back(os[x]); nd.push_back((uint8_t) !od[x]); }
        os.swap(ns); od.swap(nd);
    }
    if (events > 0) { os.push_back((uint32_t) events); od.push_back(0); }
    std::vector<uint32_t> rank(events + 1, 0u);
    std::vector<uint8_t> desc(events + 1, (uint8_t) 0);
    for (size_t x = 0; x < os.size(); x++) { rank[os[x]] = (uint32_t) x; desc[os[x]] = od[x]; }
    struct Surv { uint32_t id; uint16_t diag; uint64_t key; };
    std::vector<Surv> surv;
    std::vector<El> A, F, T;
    for (size_t r0 = 0; r0 < cands.size(); ) {
        size_t r1 = r0;
        while (r1 < cands.size() && cands[r1].id == cands[r0].id) r1++;
        const uint32_t id = cands[r0].id;
        F.clear();
        size_t t = r0;
        for (size_t e = 0; e <= events; e++) {
            A = F;
            for (; t < r1 && seg_of(cands[t].ordinal) == e; t++) A.push_back(El{cands[t].ordinal, (uint32_t) e, cands[t].diag, 0});
            F.clear();
            if (e == events) {
                if (events == 0) F = A;
                else if (!A.empty()) {                                   // mergeDiagonalDuplicates (CacheFriendlyOperations.cpp:80-115)
                    uint8_t d = (uint8_t) ((uint8_t) A[0].diag + 1);
                    for (const El &x : A) { if (d != (uint8_t) x.diag) F.push_back(x); d = (uint8_t) x.diag; }
                }
            } else if (e == 0) {
                F = A;
            } else if (!A.empty()) {                                     // mergeDiagonalKeepScoredHitsDuplicates (:118-150), align, keepMaxElement (:350-380)
                uint8_t d = (uint8_t) ((uint8_t) A.back().diag + 1);
                T.clear();
                for (size_t x = A.size(); x-- > 0;) { if (A[x].count != 0 || d != (uint8_t) A[x].diag) T.push_back(A[x]); d = (uint8_t) A[x].diag; }
                uint8_t mx = 0;
                for (El &x : T) { const int sc = score_of(id, x.diag); x.count = (uint8_t) (sc < 255 ? sc : 255); if (x.count > mx) mx = x.count; }
                for (const El &x : T) if (mx == x.count) { F.push_back(x); mx = 0; }
            }
        }
        for (const El &x : F) surv.push_back(Surv{id, x.diag, ((uint64_t) rank[x.seg] << 32) | (uint64_t) (desc[x.seg] ? 0xFFFFFFFFu - x.arrival : x.arrival)});
        r0 = r1;
    }
    // ordinal = rank of the key among all survivors; output by (target, key): the per-target lists above are already in key order
    std::vector<uint32_t> idx(surv.size());
    std::iota(idx.begin(), idx.end(), 0u);
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return surv[a].key < surv[b].key; });
    std::vector<uint32_t> rnk(surv.size());
    for (size_t x = 0; x < idx.size(); x++) rnk[idx[x]] = (uint32_t) x;
    out.clear();
    out.reserve(surv.size());
    for (size_t x = 0; x < surv.size(); x++) out.push_back(OvfCand{surv[x].id, rnk[x], surv[x].diag});
}

// B. global path over the view queries [a, b): appends their candidates at C[nCand...]; qMap/qAdd translate the view's
// query index into the chunk-local one.  Splits the range so that the index hits of one piece fit HIT_CAP.
int global_candidates(Ctx &X, const PrefilterDeviceView &Vin, const uint64_t *hOff /* host offsets of the view */, uint32_t a, uint32_t b,
                      const uint32_t *qMap, uint32_t qAddBase, uint32_t &nCand, double &hitsPerPos) {
    PrefilterDeviceView V = Vin;                      // (the k-mer lists of a piece are attached to the copy)
    std::string &err = *X.err;
    hipStream_t stream = X.stream;
    const size_t HIT_CAP = 768u << 20;                // index hits per piece kept in HBM: 17 B each (record double buffered + high diagonal byte)
    const uint64_t POS_CAP = 24u << 20;               // residues per piece
    const uint32_t QCAP = 1u << 20;
    uint32_t q0 = a;
    while (q0 < b) {
        uint32_t q1 = q0;
        {
            uint64_t posBudget = hitsPerPos > 0 ? std::min<uint64_t>(POS_CAP, (uint64_t) (0.8 * (double) HIT_CAP / hitsPerPos))
                                                : std::min<uint64_t>(POS_CAP, 1u << 20);   // first piece: small probe
            // queries per piece: what the 64-bit hit record leaves after target, diagonal byte and arrival number (as wide as the last
            // piece needed, plus one bit of slack)
            const int qBitsLeft = 64 - 8 - (int) X.seqBits - std::min(X.lastHitBits + 1, 31);
            // ... and what keeps the sorted (query, target) bits at a whole number of 8-bit radix passes, pieces of >= 8192 queries
            const int qBitsPass = (((int) X.seqBits + 13 + 7) / 8) * 8 - (int) X.seqBits;
            // one sort per query (segmented, target bits only) or one over the piece's (query, target) bits: measured both ways
            // (profiles/r03_experiments.txt) -- against 10^5 .. 2*10^6 targets the segmented sort is 2 x faster (configs 2 and 4), against
            // 1.2*10^7 (config-5 scale: 2*10^5 hits per query, 24 target bits) the whole-piece sort is 1.7 x faster.  MK_PREFILTER_SEGSORT=0/1 forces.
            static const int sortEnv = (int) knob_long("MK_PREFILTER_SEGSORT", -1);
            const bool wholeSort = sortEnv >= 0 ? sortEnv == 0 : X.seqBits >= 23;
            const int qBitsCap = wholeSort ? std::min(qBitsLeft, qBitsPass) : qBitsLeft;
            const uint32_t qCap = qBitsCap >= 20 ? QCAP : (qBitsCap < 1 ? 1u : (1u << qBitsCap));
            while (q1 < b && q1 - q0 < qCap && (hOff[q1 + 1] - hOff[q0] <= posBudget || q1 == q0)) q1++;
        }
        uint64_t totalHits = 0, nPos = 0;
        uint32_t *dHit = nullptr, *dKmer = nullptr;
        int qBits = 1, hitBits = 1;
        bool ovf = false;
        for (;;) {
            nPos = hOff[q1] - hOff[q0];
            if (nPos == 0) break;
            dHit = (uint32_t *) dev_scratch("pf_hit", (nPos + 1) * 4);
            dKmer = (uint32_t *) dev_scratch("pf_kmer", (nPos + 1) * 4);
            uint32_t *dLast = (uint32_t *) dev_scratch("pf_last", 16);
            PNULL(dHit); PNULL(dKmer); PNULL(dLast);
            const bool listed = V.p_sorted || V.kmer_size == 7;
            Kmer7Tables T7;
            T7.score2 = V.score2; T7.index2 = V.index2; T7.score3 = V.score3; T7.index3 = V.index3; T7.num3 = V.num3; T7.cum3 = V.cum3;
            T7.hist_lo = V.hist_lo; T7.hist_range = V.hist_range;
            if (listed) {
                // profile queries / k = 7: the similar k-mers of the piece as lists in HBM (count, scan, fill), walked by both probe passes
                const size_t KLIST_CAP = (size_t) 1 << 30;                 // k-mers per piece (4 GB)
                uint32_t *dCnt = (uint32_t *) dev_scratch("pf_klcount", (nPos + 1) * 4);
                uint64_t *dKOff = (uint64_t *) dev_scratch("pf_kloff", (nPos + 2) * 8);
                unsigned long long *hKTot = (unsigned long long *) pinned_scratch("pf_kltot_h", 16);
                PNULL(dCnt); PNULL(dKOff); PNULL(hKTot);
                int th = X.tb(V.p_sorted ? "profile_kmer_count" : "kmer7_count", 46.0 * (double) nPos, 0);
                if (V.p_sorted) PCHK(launch_profile_kmer_count(V.p_sorted, V.q_kmer_thr, hOff[q0], hOff[q1], dCnt, stream, V.kmer_size));
                else PCHK(launch_kmer7_count(T7, V.q_res, V.q_kmer_thr, hOff[q0], hOff[q1], dCnt, stream));
                X.te(th);
                PCHK(hipMemsetAsync(dCnt + nPos, 0, 4, stream));           // one more element: the scan then ends with the total
                hipcub::TransformInputIterator<unsigned long long, hipcub::CastOp<unsigned long long>, uint32_t *> cit(dCnt, hipcub::CastOp<unsigned long long>());
                size_t tk = 0;
                hipcub::DeviceScan::ExclusiveSum(nullptr, tk, cit, (unsigned long long *) dKOff, (int) (nPos + 1), stream);
                void *tempK = dev_scratch("pf_temp", tk);
                PNULL(tempK);
                PCHK(hipcub::DeviceScan::ExclusiveSum(tempK, tk, cit, (unsigned long long *) dKOff, (int) (nPos + 1), stream));
                PCHK(hipMemcpyAsync(hKTot, dKOff + nPos, 8, hipMemcpyDeviceToHost, stream));
                PCHK(sync_wait(stream, "wait_prefilter"));
                const size_t nK = (size_t) hKTot[0];
                if (nK > KLIST_CAP) {
                    if (q1 - q0 > 1) { q1 = q0 + (q1 - q0) / 2; continue; }
                    err = "one query has more than 2^30 similar k-mers"; return MK_ERR_UNSUPPORTED;
                }
                uint32_t *dKList = (uint32_t *) dev_scratch("pf_klist", std::max<size_t>(nK, 1) * 4);
                PNULL(dKList);
                th = X.tb(V.p_sorted ? "profile_kmer_fill" : "kmer7_fill", 46.0 * (double) nPos + 4.0 * (double) nK, (double) nK);
                if (V.p_sorted) PCHK(launch_profile_kmer_fill(V.p_sorted, V.q_kmer_thr, V.addr3, hOff[q0], hOff[q1], dKOff, dKList, stream, V.kmer_size));
                else PCHK(launch_kmer7_fill(T7, V.q_res, V.q_kmer_thr, hOff[q0], hOff[q1], dKOff, dKList, stream));
                X.te(th);
                V.klist = dKList; V.klist_off = dKOff; V.klist_pos0 = hOff[q0];
            }
            ProbeArgs A;
            A.V = V; A.pos_begin = hOff[q0]; A.pos_end = hOff[q1]; A.q_first = q0; A.seq_bits = X.seqBits; A.hit_bits = 0;
            A.hit_count = dHit; A.kmer_count = dKmer; A.keys = nullptr; A.diag_hi = nullptr;
            const unsigned blocks = (unsigned) ((nPos + 3) / 4);
            const int thCount = X.tb("kmer_probe_count", 0, 0);
            if (listed) hipLaunchKernelGGL((probe_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, A);
            else hipLaunchKernelGGL(probe_kernel<false>, dim3(blocks), dim3(256), 0, stream, A);
            X.te(thCount);
            PCHK(hipGetLastError());
            // totals: k-mers (reduce), hits (scan); the count of the last position is saved before the in-place scan
            PCHK(hipMemcpyAsync(dLast, dHit + (nPos - 1), 4, hipMemcpyDeviceToDevice, stream));
            size_t tb2 = 0, t3 = 0;
            hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, dHit, dHit, (int) nPos, stream);
            // kmer_count is u32 per position; the sum can exceed 2^32 -> accumulate in 64 bit via a transform iterator
            hipcub::TransformInputIterator<unsigned long long, hipcub::CastOp<unsigned long long>, uint32_t *> it(dKmer, hipcub::CastOp<unsigned long long>());
            hipcub::DeviceReduce::Sum(nullptr, t3, it, X.dTotals + 2, (int) nPos, stream);
            void *temp = dev_scratch("pf_temp", std::max(tb2, t3));
            PNULL(temp);
            PCHK(hipcub::DeviceReduce::Sum(temp, t3, it, X.dTotals + 2, (int) nPos, stream));
            int th = X.tb("scan", 8.0 * nPos, 0);
            PCHK(hipcub::DeviceScan::ExclusiveSum(temp, tb2, dHit, dHit, (int) nPos, stream));
            X.te(th);
            PCHK(hipMemsetAsync(X.dTotals, 0, 16, stream));
            PCHK(hipMemsetAsync(X.dTotals + 3, 0, 8, stream));
            hipLaunchKernelGGL(chunk_totals_kernel, dim3((q1 - q0 + 255) / 256), dim3(256), 0, stream, V.q_off, q0, q1 - q0, hOff[q0], nPos,
                               dHit, dLast, X.maxDbMatches, X.dTotals);
            PCHK(hipGetLastError());
            PCHK(hipMemcpyAsync(X.hTotals, X.dTotals, 32, hipMemcpyDeviceToHost, stream));
            PCHK(sync_wait(stream, "wait_prefilter"));
            totalHits = X.hTotals[0];
            ovf = X.hTotals[1] != 0;               // a query of the piece fills the reference's databaseHits buffer: it is processed alone, in segments
            if (ovf && q1 - q0 > 1) { q1 = q0 + (q1 - q0) / 2; continue; }
            if (ovf && !X.tMaskedHost) { err = "a query overflows the reference's databaseHits buffer (QueryMatcher.cpp:281-316) and the caller gave no access to the masked target residues"; return MK_ERR_UNSUPPORTED; }
            X.ts(thCount, 4.0 * (double) X.hTotals[2] + 8.0 * (double) totalHits + 1280.0 * (double) nPos, (double) X.hTotals[2]);   // bitmap word per k-mer, slot per non-empty k-mer, row heads per start
            hitsPerPos = std::max(1.0, (double) totalHits / (double) nPos);
            if (totalHits > HIT_CAP && q1 - q0 > 1) { q1 = q0 + (q1 - q0) / 2; continue; }
            // one 64-bit record per hit: query | target | low diagonal byte | arrival number within the query must fit
            qBits = 1; while ((1u << qBits) < q1 - q0) qBits++;
            hitBits = 1; while ((1ull << hitBits) < X.hTotals[3]) hitBits++;
            X.lastHitBits = hitBits;
            if (8 + hitBits + (int) X.seqBits + qBits > 64) {
                if (q1 - q0 > 1) { q1 = q0 + (q1 - q0) / 2; continue; }
                err = "a single query produces more index hits than the 64-bit hit record can number"; return MK_ERR_UNSUPPORTED;
            }
            break;
        }
        if (X.stats && nPos > 0) {                 // the piece is final: its share of the run statistics
            X.stats->db_matches += totalHits;
            if (ovf) X.stats->overflows += 1;
            if (X.statsKmers) {
                double *dSum = (double *) dev_scratch("pf_kpp", 16);
                PNULL(dSum);
                double hSum = 0;
                PCHK(hipMemsetAsync(dSum, 0, 8, stream));
                hipLaunchKernelGGL(kmers_per_pos_kernel, dim3((q1 - q0 + 255) / 256), dim3(256), 0, stream, V.q_off, q0, q1 - q0, hOff[q0], dKmer, dSum);
                PCHK(hipMemcpyAsync(&hSum, dSum, 8, hipMemcpyDeviceToHost, stream));
                PCHK(sync_wait(stream, "wait_prefilter"));
                X.stats->kmers_per_pos += kpp_value(&hSum);
            }
        }
        if (nPos > 0 && totalHits > 0) {
            if (totalHits >= 0x7FFFFFFFull) { err = "a single query produces >= 2^31 index hits"; return MK_ERR_UNSUPPORTED; }
            const uint32_t nHits = (uint32_t) totalHits;
            uint64_t *dKeys = (uint64_t *) dev_scratch("pf_keys", (size_t) nHits * 8), *dKeys2 = (uint64_t *) dev_scratch("pf_keys2", (size_t) nHits * 8);
            uint8_t *dDiagHi = (uint8_t *) dev_scratch("pf_diaghi", (size_t) nHits);
            PNULL(dKeys); PNULL(dKeys2); PNULL(dDiagHi);
            ProbeArgs A;
            A.V = V; A.pos_begin = hOff[q0]; A.pos_end = hOff[q1]; A.q_first = q0; A.seq_bits = X.seqBits; A.hit_bits = (uint32_t) hitBits;
            A.hit_count = dHit; A.kmer_count = dKmer; A.keys = dKeys; A.diag_hi = dDiagHi;
            uint8_t *dListStart = nullptr;
            if (ovf) {
                dListStart = (uint8_t *) dev_scratch("pf_liststart", (size_t) nHits);
                PNULL(dListStart);
                PCHK(hipMemsetAsync(dListStart, 0, (size_t) nHits, stream));
                A.list_start = dListStart;
            }
            const unsigned blocks = (unsigned) ((nPos + 3) / 4);
            // gather pass: slots again + 8 B per index entry read + 9 B (record, high diagonal byte) written per entry
            int th = X.tb("kmer_probe_gather", 4.0 * (double) X.hTotals[2] + 25.0 * (double) totalHits + 1280.0 * (double) nPos, (double) X.hTotals[2]);
            if (V.klist) hipLaunchKernelGGL((probe_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, A);
            else hipLaunchKernelGGL(probe_kernel<true>, dim3(blocks), dim3(256), 0, stream, A);
            X.te(th);
            PCHK(hipGetLastError());
            Segments S{nullptr, 1, 0xFFFFFFFFu};
            std::vector<uint32_t> hSegStart;
            bool ovfStopped = false;
            if (ovf) {
                // segment boundaries from the list starts (arrival numbers, ascending)
                uint32_t *dListPos = (uint32_t *) dev_scratch("pf_listpos", (size_t) nHits * 4);
                uint32_t *dNumL = (uint32_t *) dev_scratch("pf_num", 64);
                const uint32_t maxSeg = (uint32_t) std::min<uint64_t>(1u << 20, 2 * (uint64_t) nHits / X.maxDbMatches + 8);
                uint32_t *dSegStart = (uint32_t *) dev_scratch("pf_segstart", (size_t) maxSeg * 4);
                uint32_t *dSegOut = (uint32_t *) dev_scratch("pf_segout", 64);
                uint32_t *hSegOut = (uint32_t *) pinned_scratch("pf_segout_h", 64);
                PNULL(dListPos); PNULL(dNumL); PNULL(dSegStart); PNULL(dSegOut); PNULL(hSegOut);
                hipcub::CountingInputIterator<uint32_t> iota(0);
                size_t tl = 0;
                hipcub::DeviceSelect::Flagged(nullptr, tl, iota, dListStart, dListPos, dNumL, (int) nHits, stream);
                void *tempL = dev_scratch("pf_temp", tl);
                PNULL(tempL);
                PCHK(hipcub::DeviceSelect::Flagged(tempL, tl, iota, dListStart, dListPos, dNumL, (int) nHits, stream));
                PCHK(hipMemcpyAsync(hSegOut + 8, dNumL, 4, hipMemcpyDeviceToHost, stream));
                PCHK(sync_wait(stream, "wait_prefilter"));
                const uint32_t nLists = hSegOut[8];
                hipLaunchKernelGGL(overflow_segments_kernel, dim3(1), dim3(1), 0, stream, dListPos, nLists, nHits, (uint64_t) X.maxDbMatches, maxSeg, dSegStart, dSegOut);
                PCHK(hipGetLastError());
                PCHK(hipMemcpyAsync(hSegOut, dSegOut, 16, hipMemcpyDeviceToHost, stream));
                PCHK(sync_wait(stream, "wait_prefilter"));
                if (hSegOut[3]) { err = "internal: more overflow segments than expected"; return MK_ERR_DEVICE; }
                ovfStopped = hSegOut[2] != 0;
                hSegStart.resize(hSegOut[0]);
                PCHK(hipMemcpy(hSegStart.data(), dSegStart, (size_t) hSegOut[0] * 4, hipMemcpyDeviceToHost));
                S.start = dSegStart; S.n = hSegOut[0]; S.stop = hSegOut[1];
                if (knob("MK_PREFILTER_DEBUG")) fprintf(stderr, "[prefilter] a query overflows the databaseHits buffer: %u index hits in %u lists, %u segments%s\n",
                                                          nHits, nLists, hSegOut[0], ovfStopped ? ", one list alone fills the buffer: dropped" : "");
            }
            if (ovfStopped) { q0 = q1; continue; }                        // (:313-315,318-334: nothing is reported for this query)
            // sort by (query, target): only those bits are sorted, the records of a pair stay in arrival order.  The gather pass wrote the
            // records query by query, so the sort is per query over the target bits alone (segments of ~20 K records stay in L2)
            hipcub::DoubleBuffer<uint64_t> kb(dKeys, dKeys2);
            static const int sortEnv2 = (int) knob_long("MK_PREFILTER_SEGSORT", -1);
            const bool segSort = sortEnv2 >= 0 ? sortEnv2 != 0 : X.seqBits < 23;
            void *temp = nullptr;
            if (segSort) {
                const uint32_t nqc = q1 - q0;
                uint32_t *dSeg = (uint32_t *) dev_scratch("pf_seg", ((size_t) nqc + 1) * 4);
                PNULL(dSeg);
                hipLaunchKernelGGL(segment_offsets_kernel, dim3((nqc + 256) / 256), dim3(256), 0, stream, V.q_off, q0, nqc, hOff[q0], nPos, dHit, nHits, dSeg);
                PCHK(hipGetLastError());
                const int bit0 = 8 + hitBits, bit1 = 8 + hitBits + (int) X.seqBits;
                size_t tempBytes = 0;
                hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, tempBytes, kb, (int) nHits, (int) nqc, dSeg, dSeg + 1, bit0, bit1, stream);
                temp = dev_scratch("pf_temp", tempBytes);
                PNULL(temp);
                th = X.tb("sort_hits", 16.0 * (double) nHits, 0);
                PCHK(hipcub::DeviceSegmentedRadixSort::SortKeys(temp, tempBytes, kb, (int) nHits, (int) nqc, dSeg, dSeg + 1, bit0, bit1, stream));
                X.te(th);
            } else {
                const int bit0 = 8 + hitBits, bit1 = 8 + hitBits + (int) X.seqBits + qBits;
                size_t tempBytes = 0;
                hipcub::DeviceRadixSort::SortKeys(nullptr, tempBytes, kb, (int) nHits, bit0, bit1, stream);
                temp = dev_scratch("pf_temp", tempBytes);
                PNULL(temp);
                const int passes = (bit1 - bit0 + 7) / 8;
                th = X.tb("sort_hits", 16.0 * passes * (double) nHits, 0);
                PCHK(hipcub::DeviceRadixSort::SortKeys(temp, tempBytes, kb, (int) nHits, bit0, bit1, stream));
                X.te(th);
            }
            // double-hit rule, two sweeps: candidates per block -> scan -> ordered write into the candidate arrays
            const uint32_t nBlocks = (nHits + 255) / 256;
            uint32_t *dBlk = (uint32_t *) dev_scratch("pf_blk", ((size_t) nBlocks + 1) * 4);
            uint32_t *dLastBlk = (uint32_t *) dev_scratch("pf_num", 64);
            PNULL(dBlk); PNULL(dLastBlk);
            th = X.tb("double_hit", 16.0 * nHits, 0);
            if (ovf) hipLaunchKernelGGL(double_hit_count_kernel<true>, dim3(nBlocks), dim3(256), 0, stream, kb.Current(), (uint32_t) hitBits, nHits, dBlk, S);
            else hipLaunchKernelGGL(double_hit_count_kernel<false>, dim3(nBlocks), dim3(256), 0, stream, kb.Current(), (uint32_t) hitBits, nHits, dBlk, S);
            PCHK(hipGetLastError());
            PCHK(hipMemcpyAsync(dLastBlk, dBlk + (nBlocks - 1), 4, hipMemcpyDeviceToDevice, stream));
            {
                size_t t2 = 0;
                hipcub::DeviceScan::ExclusiveSum(nullptr, t2, dBlk, dBlk, (int) nBlocks, stream);
                temp = dev_scratch("pf_temp", t2);
                PNULL(temp);
                PCHK(hipcub::DeviceScan::ExclusiveSum(temp, t2, dBlk, dBlk, (int) nBlocks, stream));
            }
            uint32_t *hNum = (uint32_t *) pinned_scratch("pf_num_h", 64);
            PNULL(hNum);
            PCHK(hipMemcpyAsync(hNum, dBlk + (nBlocks - 1), 4, hipMemcpyDeviceToHost, stream));
            PCHK(hipMemcpyAsync(hNum + 1, dLastBlk, 4, hipMemcpyDeviceToHost, stream));
            PCHK(sync_wait(stream, "wait_prefilter"));
            const uint32_t nSel = hNum[0] + hNum[1];
            if ((uint64_t) nCand + nSel > X.candCap) { X.te(th); return RC_CAND_OVERFLOW; }
            if (nSel > 0 && !ovf) {
                hipLaunchKernelGGL(double_hit_emit_kernel<false>, dim3(nBlocks), dim3(256), 0, stream, kb.Current(), dDiagHi, dBlk, nHits, X.seqBits,
                                   (uint32_t) hitBits, V.q_off, q0, hOff[q0], dHit, qMap ? qMap + q0 : nullptr, qAddBase + q0, X.C, nCand, S);
                PCHK(hipGetLastError());
                nCand += nSel;
            }
            X.te(th);
            if (nSel > 0 && ovf) {
                // the per-segment diagonals of the one query: emitted, brought to the host, merged as the overflow events merged them, the
                // survivors written back in the reference's array order
                hipLaunchKernelGGL(double_hit_emit_kernel<true>, dim3(nBlocks), dim3(256), 0, stream, kb.Current(), dDiagHi, dBlk, nHits, X.seqBits,
                                   (uint32_t) hitBits, V.q_off, q0, hOff[q0], dHit, qMap ? qMap + q0 : nullptr, qAddBase + q0, X.C, nCand, S);
                PCHK(hipGetLastError());
                ScopedHost sh("host_prefilter_overflow");
                std::vector<uint32_t> hId(nSel), hOrd(nSel), hQ(1);
                std::vector<uint16_t> hDiag(nSel);
                PCHK(hipMemcpyAsync(hId.data(), X.C.id + nCand, (size_t) nSel * 4, hipMemcpyDeviceToHost, stream));
                PCHK(hipMemcpyAsync(hOrd.data(), X.C.ordinal + nCand, (size_t) nSel * 4, hipMemcpyDeviceToHost, stream));
                PCHK(hipMemcpyAsync(hDiag.data(), X.C.diag + nCand, (size_t) nSel * 2, hipMemcpyDeviceToHost, stream));
                PCHK(hipMemcpyAsync(hQ.data(), X.C.q + nCand, 4, hipMemcpyDeviceToHost, stream));
                PCHK(sync_wait(stream, "wait_prefilter"));
                std::vector<OvfCand> cands(nSel), surv;
                for (uint32_t k = 0; k < nSel; k++) cands[k] = OvfCand{hId[k], hOrd[k], hDiag[k]};
                // (emitted in (target, arrival) order: the records were sorted by target, stably)
                const size_t qg = (size_t) X.chunkQ0 + hQ[0];
                const uint64_t qs = (*X.qOffHost)[qg];
                const uint32_t L = (uint32_t) ((*X.qOffHost)[qg + 1] - qs);
                const uint8_t *tMasked = X.tMaskedHost();
                if (!tMasked) { err = "cannot fetch the masked target residues from the device"; return MK_ERR_DEVICE; }
                if (V.p_sorted) {
                    // a profile query: its alignment profile rows [column][32] (what diag_score_kernel reads) come from the device
                    std::vector<int8_t> aln((size_t) L * PROFILE_ALN_STRIDE);
                    PCHK(hipMemcpy(aln.data(), V.p_aln + hOff[q0] * PROFILE_ALN_STRIDE, aln.size(), hipMemcpyDeviceToHost));
                    replay_overflow(cands, hSegStart, [&](uint32_t id, uint16_t diag) -> int {
                        const uint64_t ts = (*X.tOffHost)[id];
                        return ungapped_score_profile(aln.data(), L, tMasked + ts, (uint32_t) ((*X.tOffHost)[id + 1] - ts), (uint32_t) diag);
                    }, surv);
                } else {
                    std::vector<int8_t> corr(L);
                    if (X.qCorrHost) std::memcpy(corr.data(), X.qCorrHost + qs, L);
                    else PCHK(hipMemcpy(corr.data(), V.q_corr + hOff[q0], L, hipMemcpyDeviceToHost));
                    int8_t m8[21 * 21];
                    for (int a = 0; a < 21; a++) for (int b = 0; b < 21; b++) m8[a * 21 + b] = (int8_t) X.ungMat->sub[a][b];
                    const uint8_t *qr = X.qResHost->data() + qs;
                    replay_overflow(cands, hSegStart, [&](uint32_t id, uint16_t diag) -> int {
                        const uint64_t ts = (*X.tOffHost)[id];
                        return ungapped_score(m8, qr, corr.data(), L, tMasked + ts, (uint32_t) ((*X.tOffHost)[id + 1] - ts), (uint32_t) diag);
                    }, surv);
                }
                const uint32_t nSurv = (uint32_t) surv.size();
                for (uint32_t k = 0; k < nSurv; k++) { hId[k] = surv[k].id; hOrd[k] = surv[k].ordinal; hDiag[k] = surv[k].diag; }
                std::vector<uint32_t> hQs(nSurv, hQ[0]);
                if (nSurv) {
                    PCHK(hipMemcpyAsync(X.C.id + nCand, hId.data(), (size_t) nSurv * 4, hipMemcpyHostToDevice, stream));
                    PCHK(hipMemcpyAsync(X.C.ordinal + nCand, hOrd.data(), (size_t) nSurv * 4, hipMemcpyHostToDevice, stream));
                    PCHK(hipMemcpyAsync(X.C.diag + nCand, hDiag.data(), (size_t) nSurv * 2, hipMemcpyHostToDevice, stream));
                    PCHK(hipMemcpyAsync(X.C.q + nCand, hQs.data(), (size_t) nSurv * 4, hipMemcpyHostToDevice, stream));
                    PCHK(sync_wait(stream, "wait_prefilter"));              // the vectors die with this scope
                }
                nCand += nSurv;
            }
        }
        q0 = q1;
    }
    return MK_OK;
}

// similar k-mers of every query of a piece from the list offsets (the cost the persistent workgroups are dealt by)
__global__ __launch_bounds__(256) void query_kmers_kernel(const uint64_t *qOff, uint32_t qFirst, uint32_t nq, uint64_t posBegin, const uint64_t *listOff, uint32_t *perQuery) {
    const uint32_t ql = blockIdx.x * blockDim.x + threadIdx.x;
    if (ql >= nq) return;
    const uint64_t n = listOff[qOff[qFirst + ql + 1] - posBegin] - listOff[qOff[qFirst + ql] - posBegin];
    perQuery[ql] = (uint32_t) min(n, (uint64_t) 0xFFFFFFFFull);
}

// A'. the wide per-query kernel over the view queries [a, b) (one chunk).  List modes (profile queries, k = 7) work in pieces whose
// similar k-mers fit KLIST_CAP; the queries the kernel cannot hold (more k-mer starts than it numbers, a full target class) come back
// in `fallback` (chunk-local ids) for the global path.
int wide_candidates(Ctx &X, const PrefilterDeviceView &Vin, const uint64_t *hOff, uint32_t a, uint32_t b, int shape, bool coResident,
                    uint32_t &nCand, std::vector<uint32_t> &fallback, double &kmersPerPos, double &globalHitsPerPos, PrefilterStats *cs, bool &fallbackKmerStats) {
    std::string &err = *X.err;
    hipStream_t stream = X.stream;
    // target classes of the production shape (MK_PREFILTER_WIDE_CLASSES = 16 / 64 forces; 128 and 256 were measured too: profiles/r04_wide_kernel.txt): 64 for
    // sequence queries (6e5 ... 3e6 index hits per fragment against a UniRef50-scale database: no or few subsets per class), 16 for profile
    // queries (1e5 hits per profile: 64 classes of a thousand records are all group overhead -- 245 against 185 ms per config-4 pass)
    if (shape == 0) {
        const long n = knob_long("MK_PREFILTER_WIDE_CLASSES", Vin.p_sorted ? 16 : 64);
        shape = n == 16 ? 0 : 2;
    }
    const WideShape &W = WIDE_SHAPES[shape];
    // sequence queries with k = 7: the 7-mers are enumerated inside the kernel (no lists in HBM, no count pass); MK_PREFILTER_K7_LISTS=1 keeps the lists
    const bool k7enum = !Vin.p_sorted && Vin.kmer_size == 7 && Vin.hist_range <= 256 && knob_long("MK_PREFILTER_K7_LISTS", 0) == 0;
    const bool listed = (Vin.p_sorted || Vin.kmer_size == 7) && !k7enum;
    fallbackKmerStats = k7enum;                        // (no per-query k-mer counts on the host in that mode: the global path counts those of the queries it takes)
    const size_t KLIST_CAP = (size_t) 1 << 31;                             // similar k-mers per piece (8 GB of table cells)
    const uint64_t POS_CAP = 48u << 20;                                    // residues per piece
    const int span = Vin.kmer_size == 7 ? 11 : 10;
    static int cus = 0;
    if (!cus) { int dev = 0; (void) hipGetDevice(&dev); if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256; }
    uint32_t *dCtr = (uint32_t *) dev_scratch("pf_wcounters", 64);
    uint32_t *hCtr = (uint32_t *) pinned_scratch("pf_wcounters_h", 64);
    unsigned long long *dTot = (unsigned long long *) dev_scratch("pf_wtotals", 16 * 8);
    unsigned long long *hTot = (unsigned long long *) pinned_scratch("pf_wtotals_h", 16 * 8);
    PNULL(dCtr); PNULL(hCtr); PNULL(dTot); PNULL(hTot);
    uint32_t p0 = a;
    while (p0 < b) {
        PrefilterDeviceView V = Vin;
        uint32_t p1 = b;
        uint64_t nPos = 0;
        const size_t fallbackMark = fallback.size();    // (a piece that is redone by the global path takes its fallback queries along)
        uint32_t *dCnt = nullptr;                       // list modes: list length of every k-mer start of the piece
        uint32_t *dQK = nullptr;
        uint16_t *dPosCost = nullptr;
        if (listed) {
            const uint64_t posBudget = kmersPerPos > 0 ? std::min<uint64_t>(POS_CAP, (uint64_t) (0.8 * (double) KLIST_CAP / kmersPerPos)) : std::min<uint64_t>(POS_CAP, 1u << 20);
            p1 = p0;
            while (p1 < b && (hOff[p1 + 1] - hOff[p0] <= posBudget || p1 == p0)) p1++;
            Kmer7Tables T7;
            T7.score2 = V.score2; T7.index2 = V.index2; T7.score3 = V.score3; T7.index3 = V.index3; T7.num3 = V.num3; T7.cum3 = V.cum3;
            T7.hist_lo = V.hist_lo; T7.hist_range = V.hist_range;
            for (;;) {
                nPos = hOff[p1] - hOff[p0];
                if (nPos == 0) break;
                dCnt = (uint32_t *) dev_scratch("pf_klcount", (nPos + 1) * 4);
                uint64_t *dKOff = (uint64_t *) dev_scratch("pf_kloff", (nPos + 2) * 8);
                unsigned long long *hKTot = (unsigned long long *) pinned_scratch("pf_kltot_h", 16);
                PNULL(dCnt); PNULL(dKOff); PNULL(hKTot);
                int th = X.tb(V.p_sorted ? "profile_kmer_count" : "kmer7_count", 46.0 * (double) nPos, 0);
                if (V.p_sorted) PCHK(launch_profile_kmer_count(V.p_sorted, V.q_kmer_thr, hOff[p0], hOff[p1], dCnt, stream, V.kmer_size));
                else PCHK(launch_kmer7_count(T7, V.q_res, V.q_kmer_thr, hOff[p0], hOff[p1], dCnt, stream));
                X.te(th);
                PCHK(hipMemsetAsync(dCnt + nPos, 0, 4, stream));           // one more element: the scan then ends with the total
                hipcub::TransformInputIterator<unsigned long long, hipcub::CastOp<unsigned long long>, uint32_t *> cit(dCnt, hipcub::CastOp<unsigned long long>());
                size_t tk = 0;
                hipcub::DeviceScan::ExclusiveSum(nullptr, tk, cit, (unsigned long long *) dKOff, (int) (nPos + 1), stream);
                void *tempK = dev_scratch("pf_temp", tk);
                PNULL(tempK);
                PCHK(hipcub::DeviceScan::ExclusiveSum(tempK, tk, cit, (unsigned long long *) dKOff, (int) (nPos + 1), stream));
                PCHK(hipMemcpyAsync(hKTot, dKOff + nPos, 8, hipMemcpyDeviceToHost, stream));
                PCHK(sync_wait(stream, "wait_prefilter"));
                const size_t nK = (size_t) hKTot[0];
                kmersPerPos = std::max(1.0, (double) nK / (double) nPos);
                if (nK > KLIST_CAP) {
                    if (p1 - p0 > 1) { p1 = p0 + (p1 - p0) / 2; continue; }
                    err = "one query has more than 2^31 similar k-mers"; return MK_ERR_UNSUPPORTED;
                }
                uint32_t *dKList = (uint32_t *) dev_scratch("pf_klist", std::max<size_t>(nK, 1) * 4);
                PNULL(dKList);
                th = X.tb(V.p_sorted ? "profile_kmer_fill" : "kmer7_fill", 46.0 * (double) nPos + 4.0 * (double) nK, (double) nK);
                if (V.p_sorted) PCHK(launch_profile_kmer_fill(V.p_sorted, V.q_kmer_thr, V.addr3, hOff[p0], hOff[p1], dKOff, dKList, stream, V.kmer_size));
                else PCHK(launch_kmer7_fill(T7, V.q_res, V.q_kmer_thr, hOff[p0], hOff[p1], dKOff, dKList, stream));
                X.te(th);
                V.klist = dKList; V.klist_off = dKOff; V.klist_pos0 = hOff[p0];
                break;
            }
        } else {
            nPos = hOff[p1] - hOff[p0];
        }
        const uint32_t nqp = p1 - p0;
        if (nPos == 0) { p0 = p1; continue; }
        // ---- similar k-mers of every query: the order in which the persistent workgroups take them, and the run statistics
        dQK = (uint32_t *) dev_scratch("pf_qkmers", (size_t) nqp * 4);
        uint32_t *hQK = (uint32_t *) pinned_scratch("pf_qkmers_h", (size_t) nqp * 4);
        PNULL(dQK); PNULL(hQK);
        if (k7enum) {
            for (uint32_t ql = 0; ql < nqp; ql++) {         // dealt by length: the number of k-mer starts stands for the work
                const int64_t L = (int64_t) (hOff[(size_t) p0 + ql + 1] - hOff[(size_t) p0 + ql]);
                hQK[ql] = (uint32_t) std::max<int64_t>(0, L - span + 1);
            }
        } else if (listed) {
            hipLaunchKernelGGL(query_kmers_kernel, dim3((nqp + 255) / 256), dim3(256), 0, stream, V.q_off, p0, nqp, hOff[p0], V.klist_off, dQK);
        } else {
            dPosCost = (uint16_t *) dev_scratch("pf_poscost", (size_t) (nPos + 16) * 2);
            PNULL(dPosCost);
            PCHK(hipMemsetAsync(dQK, 0, (size_t) nqp * 4, stream));
            const int th = X.tb("kmer_count", 5.0 * (double) nPos, 0);
            hipLaunchKernelGGL(kmer_count_kernel, dim3((unsigned) (((nPos + WAVE - 1) / WAVE + 3) / 4)), dim3(256), 0, stream, V, hOff[p0], hOff[p1], p0, dQK, dPosCost);
            X.te(th);
        }
        PCHK(hipGetLastError());
        if (!k7enum) {
            PCHK(hipMemcpyAsync(hQK, dQK, (size_t) nqp * 4, hipMemcpyDeviceToHost, stream));
            PCHK(sync_wait(stream, "wait_prefilter"));
        }
        std::vector<uint32_t> order;
        std::vector<uint32_t> orderUnits;                  // hQK of order[k]: what the tier of a query is estimated from
        {
            ScopedHost sh("host_prefilter_tiers");
            double sum = 0;
            uint32_t most = 1;
            for (uint32_t ql = 0; ql < nqp; ql++) most = std::max(most, hQK[ql]);
            std::vector<uint32_t> byClass[64];             // roughly by falling size: 64 classes
            for (uint32_t ql = 0; ql < nqp; ql++) {
                const uint64_t L = hOff[(size_t) p0 + ql + 1] - hOff[(size_t) p0 + ql];
                if (L) sum += (double) hQK[ql] / (double) L;
                if (hQK[ql] == 0) continue;                 // no k-mer: no hits
                if ((int64_t) L - span + 1 > (int64_t) W.maxpos) { fallback.push_back(p0 + ql - a); continue; }
                byClass[63 - std::min<uint64_t>(63, (uint64_t) hQK[ql] * 63 / most)].push_back(p0 + ql);
            }
            for (int c = 0; c < 64; c++) order.insert(order.end(), byClass[c].begin(), byClass[c].end());
            orderUnits.resize(order.size());
            for (size_t k = 0; k < order.size(); k++) orderUnits[k] = hQK[order[k] - p0];
            if (cs && !k7enum) cs->kmers_per_pos += sum;
        }
        if (!order.empty()) {
            int perCu = W.wgPerCu;
            // beside the alignment stage of mk_search: profile queries leave it most of the work (config 4: at most 16 prefilter waves per CU);
            // a sequence search with k = 7 or beyond 2^22 targets is 90 % prefilter (2*10^5 ... 3*10^6 index hits per fragment, 300 pairs to align):
            // the prefilter keeps both workgroups per CU (25.6 k against 19.2 k fragments/s at 11.8 M proteins, profiles/r04_wide_kernel.txt)
            if (coResident && Vin.p_sorted) perCu = std::max(1, 16 / W.waves < perCu ? 16 / W.waves : perCu);
            if (const char *e = knob("MK_PREFILTER_WG_PER_CU_W")) perCu = std::max(1, atoi(e));
            // sort key = target | arrival rank | 16-bit diagonal: the rank takes the bits the targets leave (one-class groups: the class number's bits
            // as well, wide_fwd); a class holds what that allows, within a budget for the regions of the persistent workgroups (12 bytes per record):
            // 24 GB at most (MK_PREFILTER_WIDE_POOL_GB), and never more than 45 % of what the device had free when the pool was first sized (a database
            // that fills the HBM -- 60 M proteins leave 35 GB -- must not be refused a search for want of region space)
            static uint64_t byFree = 0;
            if (!byFree) { size_t f = 0, t = 0; byFree = (hipMemGetInfo(&f, &t) == hipSuccess && f) ? std::max<uint64_t>(1ull << 30, (uint64_t) ((double) f * 0.45)) : (24ull << 30); }
            const uint64_t budget = std::min<uint64_t>((uint64_t) std::max<long>(1, knob_long("MK_PREFILTER_WIDE_POOL_GB", 24)) << 30, byFree);
            WideArgs A;
            A.t_bits = std::max<uint32_t>(X.seqBits, 20u);
            A.one_class_hits = knob_long("MK_TEST_WIDE_ONE_CLASS", 0) ? 0u : (uint32_t) std::min<uint64_t>(1ull << (48u - A.t_bits), 0xFFFFFFFFull);
            // (the rank of a hit has 48 - t_bits + log2(classes) bits in the key of a one-class group; the arrival ordinal of a candidate is a uint32)
            const uint64_t byRank = std::min<uint64_t>(1ull << (48u - A.t_bits), (1ull << 31) / (uint64_t) W.nCls);
            const uint32_t nCandPiece = nCand;
            // Round 6: PARTS.  A query whose hits do not fit one region is taken by M = 2, 4, 8 ... workgroups, each keeping the target classes of one
            // residue modulo M (wide_kernel): every region stays in use, whatever the hits per query -- at 60 M proteins a region of the 24 GB pool
            // holds 4 M hits and a third of the fragments of a metagenome gather more; they went to the sort-based global path (rounds 4-5), and as a
            // first attempt of this round to launches with a quarter of the workgroups and four times the region, which ran at a third of the speed
            // (profiles/r06_config5.txt).  A part that fills a class has wasted its pass 1, so every query STARTS with the M its expected hits need --
            // similar k-mers (k-mer starts when the 7-mers are enumerated in the kernel) x the hits per unit this database has shown so far (before
            // the first launch: index entries per table cell, x 2 500 similar 7-mers per start), 40 % of headroom -- and only a misjudged part
            // is run again, as its two halves.
            A.cls_cap = W.clsCap != 0 ? (uint32_t) W.clsCap
                                      : (uint32_t) std::max<uint64_t>(4096, std::min<uint64_t>(byRank, budget / ((uint64_t) cus * perCu * W.nCls * 12ull)) & ~63ull);
            const unsigned gridMax = (unsigned) cus * perCu;
            for (;;) {                                         // a pool that cannot be had is halved (fuller classes: more parts per query)
                const size_t regionRecs = (size_t) W.nCls * A.cls_cap;
                A.pool = (uint32_t *) dev_scratch("pf_wpool", (size_t) gridMax * regionRecs * 12);
                if (A.pool || W.clsCap != 0 || A.cls_cap <= 4096) break;
                (void) hipGetLastError();
                A.cls_cap = std::max<uint32_t>(4096, (A.cls_cap / 2) & ~63u);
            }
            PNULL(A.pool);
            int nClsLog = 0; while ((1 << nClsLog) < W.nCls) nClsLog++;
            // (the part number takes bits of the mapped target id below the class: at most 16 parts, and 8 bits stay for the targets of a class and part)
            int maxLogM = std::max(0, std::min(4, (int) A.t_bits - nClsLog - 8));
            maxLogM = (int) std::min<long>(maxLogM, std::max(0L, knob_long("MK_PREFILTER_WIDE_MAX_LOGM", 4)));
            double perUnit;
            { std::lock_guard<std::mutex> lk(g_memoMutex); perUnit = g_memo.entries == (const void *) Vin.entries ? g_memo.wideHitsPerUnit : 0.0; }
            if (perUnit <= 0) perUnit = (double) Vin.n_entries / (Vin.kmer_size == 7 ? 1.28e9 : 6.4e7) * (k7enum ? 2500.0 : 1.0);
            std::vector<uint32_t> itemQ, itemPart;           // the work list: view query id, r | log2(M) << 8
            const double regionHits = (double) W.nCls * (double) A.cls_cap;
            for (size_t k = 0; k < order.size(); k++) {
                const double est = 1.4 * perUnit * (double) orderUnits[k];
                int lm = 0;
                while (lm < maxLogM && est > regionHits * (double) (1u << lm)) lm++;
                if (est > 4.0 * regionHits * (double) (1u << lm)) { fallback.push_back(order[k] - a); continue; }   // (far beyond every region: the global path)
                // (clearly beyond the reference's maxDbMatches: its overflow path decides, which the global path restates -- no pass 1 to find that out)
                if (perUnit * (double) orderUnits[k] > 1.5 * (double) X.maxDbMatches) { fallback.push_back(order[k] - a); continue; }
                for (uint32_t r = 0; r < (1u << lm); r++) { itemQ.push_back(order[k]); itemPart.push_back(r | ((uint32_t) lm << 8)); }
            }
            bool redoGlobal = false;
            double unitsDone = 0, hitsDone = 0, kmersPerPosDone = 0;
            unsigned long long dbMatchesDone = 0;
            for (int round = 0; !itemQ.empty(); round++) {
                const size_t nItems = itemQ.size();
                uint32_t *hItems = (uint32_t *) pinned_scratch("pf_witems_h", nItems * 8);
                uint32_t *dItems = (uint32_t *) dev_scratch("pf_witems", nItems * 8);
                uint32_t *dOvfItems = (uint32_t *) dev_scratch("pf_wovf", nItems * 8);
                PNULL(hItems); PNULL(dItems); PNULL(dOvfItems);
                std::memcpy(hItems, itemQ.data(), nItems * 4);
                std::memcpy(hItems + nItems, itemPart.data(), nItems * 4);
                const unsigned launch = (unsigned) std::min<size_t>(nItems, gridMax);
                PCHK(hipMemcpyAsync(dItems, hItems, nItems * 8, hipMemcpyHostToDevice, stream));
                PCHK(hipMemsetAsync(dCtr, 0, 64, stream));
                hCtr[0] = nCand;                               // (pinned: the copy below reads it when the stream gets there -- synchronised before it is reused)
                PCHK(hipMemcpyAsync(dCtr, hCtr, 4, hipMemcpyHostToDevice, stream));
                PCHK(hipMemsetAsync(dTot, 0, 16 * 8, stream));
                A.V = V; A.queries = dItems; A.parts = dItems + nItems; A.n_queries = (uint32_t) nItems; A.q_first = a;
                A.C = X.C; A.cand_cap = X.candCap; A.counters = dCtr;
                A.overflow_list = dOvfItems; A.overflow_parts = dOvfItems + nItems; A.overflow_count = dCtr + 4; A.totals = dTot; A.work_counter = dCtr + 8;
                A.pos_cost = dPosCost; A.pos_begin = hOff[p0];
                A.max_log_m = knob_long("MK_PREFILTER_WIDE_KERNEL_HALVES", 1) ? (uint32_t) maxLogM : 0u;
                A.max_db_matches = (uint32_t) std::min<uint64_t>(X.maxDbMatches, 0xFFFFFFFFull);
                const int th = X.tb(round == 0 ? "prefilter_query_wide" : "prefilter_query_wide_retry", 0, 0);
                if (k7enum) launch_wide<W_MODE_ENUM7>(shape, A, launch, stream);
                else if (listed) launch_wide<W_MODE_LIST>(shape, A, launch, stream);
                else launch_wide<W_MODE_ENUM6>(shape, A, launch, stream);
                X.te(th);
                PCHK(hipGetLastError());
                PCHK(hipMemcpyAsync(hCtr, dCtr, 64, hipMemcpyDeviceToHost, stream));
                PCHK(hipMemcpyAsync(hTot, dTot, 16 * 8, hipMemcpyDeviceToHost, stream));
                PCHK(sync_wait(stream, "wait_prefilter"));
                if (knob("MK_PREFILTER_DEBUG"))
                    fprintf(stderr, "[prefilter] wide piece %u..%u round %d (%s; %u regions of %d x %u records): items %zu overflowed %llu (halved in the kernel %llu) | kmers %.3g hits %.3g pos %.3g | wg-ticks gather %.3g filter %.3g collect+sort %.3g rule+emit %.3g overflowed %.3g | extra sub-class passes %llu | cand %u -> %u\n",
                            p0, p1, round, k7enum ? "7-mers in the kernel" : (listed ? "lists" : "k = 6 enumerator"), launch, W.nCls, A.cls_cap, nItems, hTot[8], hTot[13], (double) hTot[0], (double) hTot[1], (double) hTot[2], (double) hTot[3], (double) hTot[12], (double) hTot[4], (double) hTot[5],
                            (double) hTot[6], hTot[9], nCand, hCtr[0]);
                if (hCtr[0] > X.candCap) return RC_CAND_OVERFLOW;
                if (hTot[10] != 0) { redoGlobal = true; break; }
                X.ts(th, 16.0 * (double) hTot[0] + 6.0 * (double) hTot[1], (double) hTot[0]);
                nCand = hCtr[0];
                dbMatchesDone += hTot[1];
                if (k7enum) kmersPerPosDone += kpp_value(&hTot[11]);
                unitsDone += (double) (k7enum ? hTot[2] : hTot[0]); hitsDone += (double) hTot[1];
                const uint32_t nOvf = hCtr[4];
                itemQ.clear(); itemPart.clear();
                if (nOvf > 0) {
                    uint32_t *hOvf = (uint32_t *) pinned_scratch("pf_wovf_h", (size_t) nOvf * 8);
                    PNULL(hOvf);
                    PCHK(hipMemcpyAsync(hOvf, dOvfItems, (size_t) nOvf * 4, hipMemcpyDeviceToHost, stream));
                    PCHK(hipMemcpyAsync(hOvf + nOvf, dOvfItems + nItems, (size_t) nOvf * 4, hipMemcpyDeviceToHost, stream));
                    PCHK(sync_wait(stream, "wait_prefilter"));
                    for (uint32_t k = 0; k < nOvf && !redoGlobal; k++) {
                        if (hOvf[nOvf + k] == 0x80000000u) { fallback.push_back(hOvf[k]); continue; }      // the reference's databaseHits overflow: the global path
                        const uint32_t r = hOvf[nOvf + k] & 0xFFu, lm = hOvf[nOvf + k] >> 8;
                        // the part's two halves; a part that cannot be halved any more has emitted nothing, but the other parts of its query have:
                        // the piece is done again by the global path (never seen: the estimate would have to be off by the factor of 4 above)
                        if ((int) lm >= maxLogM) { redoGlobal = true; break; }
                        for (uint32_t h = 0; h < 2; h++) { itemQ.push_back(hOvf[k] + a); itemPart.push_back((2u * r + h) | ((lm + 1) << 8)); }
                    }
                }
                if (redoGlobal) break;
            }
            if (!redoGlobal) {
                if (cs) { cs->db_matches += dbMatchesDone; if (k7enum) cs->kmers_per_pos += kmersPerPosDone; }
                if (unitsDone > 0 && hitsDone > 0) {
                    std::lock_guard<std::mutex> lk(g_memoMutex);
                    if (g_memo.entries == (const void *) Vin.entries) g_memo.wideHitsPerUnit = g_memo.wideHitsPerUnit > 0 ? 0.5 * (g_memo.wideHitsPerUnit + hitsDone / unitsDone) : hitsDone / unitsDone;
                }
            }
            if (redoGlobal) {
                // a single target class held more double-hit survivors of one sub-class than the LDS sort: the piece's candidates are dropped
                // (nCand goes back to where the piece began) and the sort-based path does the piece
                const bool sk = X.statsKmers;
                X.statsKmers = X.stats != nullptr && k7enum;
                fallback.resize(fallbackMark);
                nCand = nCandPiece;
                const int rc = global_candidates(X, Vin, hOff, p0, p1, nullptr, (uint32_t) 0 - a, nCand, globalHitsPerPos);
                X.statsKmers = sk;
                if (rc != MK_OK) return rc;
            }
        }
        p0 = p1;
    }
    std::sort(fallback.begin(), fallback.end());
    return MK_OK;
}

}  // namespace

int run_prefilter(const PrefilterDeviceView &V, const std::vector<uint64_t> &qOff, const std::vector<uint8_t> &qRes,
                  const int8_t *qCorrHost,
                  const std::vector<uint64_t> &tOff, const mk_params &P, int binCount, hipStream_t stream,
                  HostBlock &outBlk, size_t &nOut, std::vector<uint64_t> &outOff, std::string &err, timed_begin_fn tb, timed_end_fn te, timed_set_fn ts,
                  const PrefilterHooks &hooks) {
    const uint32_t nq = V.n_queries;
    nOut = 0;
    outOff.assign((size_t) nq + 1, 0);
    // room for n more hits in the result block; the estimate of the final size follows the progress through the batch
    auto reserve_out = [&](size_t n, uint32_t qDone) -> bool {
        if ((nOut + n) * sizeof(mk_hit) <= outBlk.cap && outBlk.p) return true;
        if (hipStreamSynchronize(stream) != hipSuccess) return false;      // DMA into the old block must have landed
        if (hooks.before_grow) hooks.before_grow();
        const double frac = std::max(0.02, (double) qDone / (double) nq);
        const size_t want = (size_t) ((double) (nOut + n) / frac * 1.1) + 4096;
        return outBlk.reserve(std::max(want, nOut + n + 4096) * sizeof(mk_hit), nOut * sizeof(mk_hit));
    };
    const int maxHits = std::min<int>(P.max_seqs, (int) V.n_targets);
    const uint64_t dbSize = V.n_targets;
    const uint32_t QCAP = 1u << 20;                   // queries per chunk (20-bit field of the output sort key)
    // candidates per chunk held in HBM (~44 B each).  Against a database of more than 2^24 targets a fragment leaves 4e4 candidates (60 M proteins): twice
    // the room there -- a piece of the wide kernel's work then has twice the items per workgroup and half as many launch tails (a quarter of a
    // piece's time with 4.5 items per workgroup, profiles/r06_config5.txt)
    const uint32_t CAND_CAP = (V.n_targets > (1u << 24) ? 192u : 96u) << 20;
    if (dbSize >= (1ull << 27)) { err = "more than 2^27 targets"; return MK_ERR_UNSUPPORTED; }
    uint32_t seqBits = 1; while ((1ull << seqBits) < dbSize) seqBits++;
    // front end: the per-query kernels keep the target id in a 22-bit field of their hit records
    bool useFused = seqBits <= REC_T_BITS;
    if (V.p_sorted || V.kmer_size == 7) useFused = false;   // profile queries, k = 7: k-mer lists (stream_kernel enumerates two 3-mer rows)
    // ... and whatever the 22-bit per-query kernels do not take goes to the wide per-query kernel (round 4; up to 2^27 targets, lists or the k = 6
    // enumerator), the sort-based global path behind it.  MK_PREFILTER_PATH = auto | fused | wide | global forces (fused: where it applies)
    bool useWide = !useFused && seqBits <= W_T_BITS_MAX;
    if (const char *e = knob("MK_PREFILTER_PATH")) {
        if (!strcmp(e, "global")) { useFused = false; useWide = false; }
        else if (!strcmp(e, "wide")) { useFused = false; useWide = seqBits <= W_T_BITS_MAX; }
        else if (strcmp(e, "fused") && strcmp(e, "auto")) { err = "MK_PREFILTER_PATH must be auto, fused, wide or global"; return MK_ERR_ARG; }
    }
    int tierBase = 0;
    if (const char *e = knob("MK_PREFILTER_TIERS")) {
        if (!strcmp(e, "tiny")) tierBase = N_TIERS;
        else if (strcmp(e, "default")) { err = "MK_PREFILTER_TIERS must be default or tiny"; return MK_ERR_ARG; }
    }
    const FusedTier *tiers = TIERS + tierBase;
    int nTiersUsed = N_TIERS;                          // MK_PREFILTER_MAX_TIERS: leave the largest tier(s) to the global path
    if (const char *e = knob("MK_PREFILTER_MAX_TIERS")) nTiersUsed = std::min(N_TIERS, std::max(1, atoi(e)));
    if (hooks.max_tiers > 0) nTiersUsed = std::min(nTiersUsed, hooks.max_tiers);
    int firstTier = 0;                                 // MK_PREFILTER_FIRST_TIER: queries that would fit a smaller tier go to the global path
    if (const char *e = knob("MK_PREFILTER_FIRST_TIER")) firstTier = std::max(0, atoi(e));
    double candPerQuery, globalHitsPerPos = 0, wideKmersPerPos = 0;
    {
        std::lock_guard<std::mutex> lk(g_memoMutex);
        if (g_memo.entries != (const void *) V.entries || g_memo.nTargets != V.n_targets) { g_memo = SizingMemo(); g_memo.entries = V.entries; g_memo.nTargets = V.n_targets; }
        candPerQuery = g_memo.candPerQuery;
    }
    static_assert(N_TIERS == 4, "SizingMemo holds four tiers");
    SubMat ungMat;
    build_submat(ungMat, MAT_BLOSUM62, 2.0f, -0.2f);

    Ctx X;
    X.stream = stream; X.err = &err; X.tb = tb; X.te = te; X.ts = ts; X.seqBits = seqBits;
    X.maxDbMatches = std::max<uint64_t>(1000000, dbSize) * 2;   // QueryMatcher.cpp:43
    X.candCap = CAND_CAP;
    X.statsKmers = hooks.stats && !useFused && !useWide;
    X.qOffHost = &qOff; X.qResHost = &qRes; X.qCorrHost = V.p_sorted ? nullptr : qCorrHost; X.tMaskedHost = hooks.t_masked_host; X.tOffHost = &tOff; X.ungMat = &ungMat;
    X.dTotals = (unsigned long long *) dev_scratch("pf_totals", 64);
    X.hTotals = (unsigned long long *) pinned_scratch("pf_totals_h", 64);
    PNULL(X.dTotals); PNULL(X.hTotals);
    CandArrays &C = X.C;
    C.q = (uint32_t *) dev_scratch("pf_cq", (size_t) CAND_CAP * 4); C.id = (uint32_t *) dev_scratch("pf_cid", (size_t) CAND_CAP * 4);
    C.ordinal = (uint32_t *) dev_scratch("pf_cord", (size_t) CAND_CAP * 4); C.diag = (uint16_t *) dev_scratch("pf_cdiag", (size_t) CAND_CAP * 2);
    C.score = (int32_t *) dev_scratch("pf_cscore", (size_t) CAND_CAP * 4);
    PNULL(C.q); PNULL(C.id); PNULL(C.ordinal); PNULL(C.diag); PNULL(C.score);
    uint32_t *dCounters = (uint32_t *) dev_scratch("pf_fcounters", 64);
    unsigned long long *dFTotals = (unsigned long long *) dev_scratch("pf_ftotals", 16 * 8 * N_TIERS);
    uint32_t *hCounters = (uint32_t *) pinned_scratch("pf_fcounters_h", 64);
    unsigned long long *hFTotals = (unsigned long long *) pinned_scratch("pf_ftotals_h", 16 * 8 * (N_TIERS + 1));
    PNULL(dCounters); PNULL(dFTotals); PNULL(hCounters); PNULL(hFTotals);

    uint32_t q0 = 0;
    uint32_t chunkLimit = QCAP;
    // ---- chunk size: bounded by the query field of the sort key and by the candidate buffers
    const auto chunk_end = [&](uint32_t from, uint32_t limit) -> uint32_t {
        uint32_t want = limit;
        if (hooks.max_chunk_queries) {
            // the consumer of the chunks (the alignment stage of mk_search) starts when the FIRST chunk is done: the first chunks are small
            // (1/4, then 1/2 of the limit), the later ones large enough to keep the launches of both stages long
            uint32_t lim = hooks.max_chunk_queries;
            if (hooks.chunk_ramp) {
                const uint32_t full = lim;
                lim = from == 0 ? std::max(1024u, full / 4) : (from < full ? std::max(1024u, full / 2) : full);
                // (shrinking the LAST chunks as well, to shorten the consumer's tail, was measured and costs more in short launches than it saves)
            }
            want = std::min(want, lim);
        }
        if (candPerQuery > 0) want = (uint32_t) std::min<double>(want, std::max(1.0, 0.6 * (double) CAND_CAP / candPerQuery));
        else want = std::min<uint32_t>(want, 1u << 16);                    // nothing known yet: a small probe chunk
        return (uint32_t) std::min<uint64_t>(nq, (uint64_t) from + std::max<uint32_t>(want, 1));
    };
    // (Measured in round 5 and not kept: the similar-k-mer count of the NEXT chunk on a side stream beside this chunk's tier kernels -- 36 ms per step
    //  off the head of the prefilter chain, and the step 852-862 against 838-844 ms: the sizing kernel slows the largest tier by more than it saves,
    //  profiles/r05_prefilter_tiers.txt.)
    while (q0 < nq) {
        const uint32_t q1 = chunk_end(q0, chunkLimit);
        const uint32_t nqc = q1 - q0;
        uint32_t nCand = 0;
        X.chunkQ0 = q0;
        PrefilterStats cs;                                                 // this attempt at the chunk; committed when it is final
        X.stats = hooks.stats ? &cs : nullptr;
        std::vector<uint32_t> fallback;                                     // chunk-local ids for the global path
        bool fallbackKmerStats = false;
        int rc = MK_OK;
        if (useWide) {
            // ---- A'. the wide per-query kernel; what it cannot hold comes back in `fallback`
            rc = wide_candidates(X, V, qOff.data(), q0, q1, tierBase ? 1 : 0, hooks.co_resident, nCand, fallback, wideKmersPerPos, globalHitsPerPos, hooks.stats ? &cs : nullptr,
                                 fallbackKmerStats);
            X.statsKmers = hooks.stats && fallbackKmerStats;
        } else if (useFused) {
            // ---- A. fused kernels, one launch per LDS tier; the tier follows the expected number of index hits
            // similar-k-mer count per query (exact, kmer_count_kernel) -> expected index hits -> tier, all on the device (tier_*_kernel): nothing
            // between the count and the tiers' launches waits for the host
            constexpr uint32_t LIST_PEEK = 4096;                                 // entries of the global path's two lists fetched with the counters
            uint16_t *dPosCost = (uint16_t *) dev_scratch("pf_poscost", (size_t) (qOff[q1] - qOff[q0] + 16) * 2);
            uint32_t *dQK = (uint32_t *) dev_scratch("pf_qkmers", (size_t) nqc * 4);
            PNULL(dPosCost);
            uint32_t *dList = (uint32_t *) dev_scratch("pf_flist", (size_t) nqc * 4), *dFallback = (uint32_t *) dev_scratch("pf_ffallback", (size_t) nqc * 4);
            uint32_t *dOvf = (uint32_t *) dev_scratch("pf_fovf", (size_t) nqc * 4 * N_TIERS);
            uint32_t *dTier = (uint32_t *) dev_scratch("pf_tierplan", (2 * TIER_BINS + 16 + 4) * 4);      // hist, cursor, info, statistics (a double)
            uint16_t *dBin = (uint16_t *) dev_scratch("pf_tierbin", (size_t) nqc * 2);
            uint32_t *hTier = (uint32_t *) pinned_scratch("pf_tierplan_h", (16 + 4 + 2 * LIST_PEEK) * 4);
            PNULL(dQK); PNULL(dList); PNULL(dFallback); PNULL(dOvf); PNULL(dTier); PNULL(dBin); PNULL(hTier);
            uint32_t *dInfo = dTier + 2 * TIER_BINS;
            double *dKpp = reinterpret_cast<double *>(dInfo + 16);
            double limit[N_TIERS];                                      // most k-mers a query may have to be tried in tier t
            {
                std::lock_guard<std::mutex> lk(g_memoMutex);
                for (int t = 0; t < N_TIERS; t++) {
                    const double hpk = g_memo.hitsPerKmer[t] > 0 ? g_memo.hitsPerKmer[t] : std::max(0.05, (double) V.n_entries / 64.0e6);
                    limit[t] = (double) tiers[t].cap / (hpk * g_memo.margin[t]);
                }
            }
            {
                const uint64_t pb = qOff[q0], pe = qOff[q1];
                PCHK(hipMemsetAsync(dTier, 0, (2 * TIER_BINS + 16 + 4) * 4, stream));
                PCHK(hipMemsetAsync(dQK, 0, (size_t) nqc * 4, stream));
                if (pe > pb) {
                    const int th = tb("kmer_count", 5.0 * (double) (pe - pb), 0);
                    hipLaunchKernelGGL(kmer_count_kernel, dim3((unsigned) (((pe - pb + WAVE - 1) / WAVE + 3) / 4)), dim3(256), 0, stream, V, pb, pe, q0, dQK, dPosCost);
                    te(th);
                    PCHK(hipGetLastError());
                }
                TierPlan T;
                T.qk = dQK; T.q_off = V.q_off; T.q0 = q0; T.nqc = nqc;
                for (int t = 0; t < N_TIERS; t++) { T.limit[t] = limit[t]; T.maxpos[t] = tiers[t].maxpos; }
                T.nTiersUsed = nTiersUsed; T.firstTier = firstTier;
                T.hist = dTier; T.cursor = dTier + TIER_BINS; T.bin = dBin; T.list = dList; T.fallback = dFallback; T.info = dInfo;
                T.kpp = hooks.stats ? dKpp : nullptr;
                const int th = tb("tier_assign", 10.0 * nqc, 0);
                hipLaunchKernelGGL(tier_classify_kernel, dim3((nqc + 255) / 256), dim3(256), 0, stream, T);
                hipLaunchKernelGGL(tier_scan_kernel, dim3(1), dim3(64), 0, stream, T);
                hipLaunchKernelGGL(tier_scatter_kernel, dim3((nqc + 255) / 256), dim3(256), 0, stream, T);
                te(th);
                PCHK(hipGetLastError());
            }
            PCHK(hipMemsetAsync(dCounters, 0, 64, stream));
            PCHK(hipMemsetAsync(dFTotals, 0, 16 * 8 * N_TIERS, stream));
            int thFused[N_TIERS];
            for (int t = 0; t < N_TIERS; t++) {
                thFused[t] = -1;
                if (t < nTiersUsed) {
                    // streamed tier: persistent workgroups, each with its own hit region; how many queries it has is known on the device only
                    static int cus = 0;
                    if (!cus) { int dev = 0; (void) hipGetDevice(&dev); if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256; }
                    int perCu = tiers[t].wgPerCu;
                    if (hooks.co_resident) perCu = std::max(1, tiers[t].waves * perCu > 16 ? 16 / tiers[t].waves : perCu);   // at most 16 waves per CU
                    // ... and 8 for the one-wave tiers (round 6, profiles/r06_small_tier_occupancy.txt): their ~50 ms per step hardly depend on the number of
                    // workgroups (56 / 50 / 48 ms with 4 / 8 / 16 per CU: each wave walks its queries through chains of dependent loads), but what they occupy is
                    // taken from the alignment stage, whose backlog then meets the third tier: step 819 -> 792 ms with 8 instead of 16
                    if (hooks.co_resident && tiers[t].waves == 1) perCu = std::min(perCu, 8);
                    if (const char *e = knob(t == 2 ? "MK_PREFILTER_WG_PER_CU_A" : (t == 3 ? "MK_PREFILTER_WG_PER_CU_B" : "MK_PREFILTER_WG_PER_CU_S"))) perCu = std::max(1, atoi(e));
                    const unsigned launch = (unsigned) std::min<size_t>((size_t) nqc, (size_t) cus * perCu);     // (a chunk has at most nqc queries for any tier)
                    char pn[32];
                    snprintf(pn, sizeof(pn), "pf_pool%d", t);
                    StreamArgs A;
                    A.pool = (uint64_t *) dev_scratch(pn, (size_t) launch * tiers[t].cap * 8);
                    PNULL(A.pool);
                    A.V = V; A.queries = dList; A.own_first = dInfo + t; A.own_count = dInfo + 4 + t; A.q_first = q0;
                    A.prev_list = t > 0 ? dOvf + (size_t) (t - 1) * nqc : nullptr; A.prev_count = t > 0 ? dCounters + 4 + (t - 1) : dCounters + 15;
                    A.C = C; A.cand_cap = CAND_CAP; A.counters = dCounters;
                    A.overflow_list = dOvf + (size_t) t * nqc; A.overflow_count = dCounters + 4 + t; A.totals = dFTotals + 16 * t;
                    A.work_counter = dCounters + 8 + t;
                    A.pos_cost = dPosCost; A.pos_begin = qOff[q0];
                    char nm[48];
                    snprintf(nm, sizeof(nm), "prefilter_query_cap%d", tiers[t].cap);
                    thFused[t] = tb(nm, 0, 0);
                    launch_stream(tierBase + t, A, launch, stream);
                    te(thFused[t]);
                    PCHK(hipGetLastError());
                }
            }
            // the counts of the assignment, the statistics and the head of the two lists the global path takes (its own and what the largest tier
            // could not hold) come with the counters: one synchronisation per chunk here
            const uint32_t peek = std::min(nqc, LIST_PEEK);
            PCHK(hipMemcpyAsync(hTier, dInfo, (16 + 4) * 4, hipMemcpyDeviceToHost, stream));
            PCHK(hipMemcpyAsync(hTier + 20, dFallback, (size_t) peek * 4, hipMemcpyDeviceToHost, stream));
            PCHK(hipMemcpyAsync(hTier + 20 + LIST_PEEK, dOvf + (size_t) (nTiersUsed - 1) * nqc, (size_t) peek * 4, hipMemcpyDeviceToHost, stream));
            PCHK(hipMemcpyAsync(hCounters, dCounters, 64, hipMemcpyDeviceToHost, stream));
            PCHK(hipMemcpyAsync(hFTotals + 16, dFTotals, 16 * 8 * N_TIERS, hipMemcpyDeviceToHost, stream));
            PCHK(sync_wait(stream, "wait_prefilter"));
            const uint32_t *hInfo = hTier;
            size_t listed[N_TIERS];
            for (int t = 0; t < N_TIERS; t++) listed[t] = hInfo[4 + t];
            const uint32_t nFallback = hInfo[8];
            if (hooks.stats) cs.kmers_per_pos += kpp_value(hTier + 16);
            for (int k = 0; k < 16; k++) { hFTotals[k] = 0; for (int t = 0; t < N_TIERS; t++) hFTotals[k] += hFTotals[16 * (t + 1) + k]; }
            cs.db_matches += hFTotals[1];
            if (knob("MK_PREFILTER_DEBUG"))
                for (int t = 0; t < N_TIERS; t++) {
                    const unsigned long long *T = hFTotals + 16 * (t + 1);
                    fprintf(stderr, "[prefilter]   tier %d (%s %d): queries %zu overflowed %llu | kmers %.3g hits %.3g pos %.3g | wg-ticks gather %.3g sort %.3g emit %.3g overflowed %.3g | extra class passes %llu | survivors %.3g candidates %.3g\n",
                            t, "region", tiers[t].cap, listed[t], T[8], (double) T[0], (double) T[1], (double) T[2], (double) T[3], (double) T[4], (double) T[5], (double) T[6], T[9], (double) T[10], (double) T[11]);
                }
            const uint32_t nOvf = hCounters[4 + nTiersUsed - 1];               // what even the largest tier in use could not hold
            if (hCounters[0] > CAND_CAP) rc = RC_CAND_OVERFLOW;
            else {
                nCand = hCounters[0];
                // algorithmic bytes of the per-query launches, SURVEY.md 8(d): B_lookup = 16 B per probed k-mer + 6 B per index entry read
                // (+ 7 B per candidate written, booked with the back end)
                for (int t = 0; t < N_TIERS; t++)
                    if (thFused[t] >= 0) {
                        const unsigned long long *T = hFTotals + 16 * (t + 1);
                        ts(thFused[t], 16.0 * (double) T[0] + 6.0 * (double) T[1], (double) T[0]);
                    }
                if (knob("MK_PREFILTER_DEBUG"))
                    fprintf(stderr, "[prefilter] chunk %u..%u: tiers %zu/%zu/%zu/%zu too-long %zu overflow %u | kmers %.3g hits %.3g pos %.3g | wg-ticks gather %.3g (mean wave %.3g) sort %.3g emit %.3g overflowed %.3g | cand %u\n",
                            q0, q1, listed[0], listed[1], listed[2], listed[3], (size_t) nFallback, nOvf, (double) hFTotals[0], (double) hFTotals[1],
                            (double) hFTotals[2], (double) hFTotals[3], (double) hFTotals[7], (double) hFTotals[4], (double) hFTotals[5], (double) hFTotals[6], nCand);
                // per tier: hits per similar k-mer of the queries that fitted, and a safety margin that follows the overflow rate
                {
                    std::lock_guard<std::mutex> lkMemo(g_memoMutex);
                    for (int t = 0; t < N_TIERS; t++) {
                        const unsigned long long *T = hFTotals + 16 * (t + 1);
                        if (T[0] > 0) g_memo.hitsPerKmer[t] = std::max(0.01, (double) T[1] / (double) T[0]);
                        const double tried = (double) listed[t] + (t > 0 ? (double) hCounters[4 + t - 1] : 0.0);
                        if (tried >= 256) {
                            const double frac = (double) hCounters[4 + t] / tried;
                            if (frac > 0.04) g_memo.margin[t] = std::min(3.0, g_memo.margin[t] * 1.08);
                            else if (frac < 0.01) g_memo.margin[t] = std::max(1.05, g_memo.margin[t] * 0.98);
                        }
                    }
                }
                if (nFallback > 0 || nOvf > 0) {
                    // what the global path takes: the queries no tier was tried for and the ones the largest tier could not hold
                    const uint32_t *hFb = hTier + 20, *hOvf = hTier + 20 + LIST_PEEK;
                    if (nFallback > peek || nOvf > peek) {                     // (rare: the lists are longer than what came with the counters)
                        uint32_t *hMore = (uint32_t *) pinned_scratch("pf_fovf_h", ((size_t) nFallback + nOvf + 2) * 4);
                        PNULL(hMore);
                        if (nFallback) PCHK(hipMemcpyAsync(hMore, dFallback, (size_t) nFallback * 4, hipMemcpyDeviceToHost, stream));
                        if (nOvf) PCHK(hipMemcpyAsync(hMore + nFallback, dOvf + (size_t) (nTiersUsed - 1) * nqc, (size_t) nOvf * 4, hipMemcpyDeviceToHost, stream));
                        PCHK(sync_wait(stream, "wait_prefilter"));
                        hFb = hMore; hOvf = hMore + nFallback;
                    }
                    fallback.assign(hFb, hFb + nFallback);
                    fallback.insert(fallback.end(), hOvf, hOvf + nOvf);
                    std::sort(fallback.begin(), fallback.end());
                }
            }
        }
        if (rc == MK_OK && !useFused && !useWide) {
            // ---- B. the whole chunk through the global path
            rc = global_candidates(X, V, qOff.data(), q0, q1, nullptr, (uint32_t) 0 - q0, nCand, globalHitsPerPos);
        } else if (rc == MK_OK && !fallback.empty() && V.p_sorted) {
            // ---- B. profile queries the wide kernel could not take: their columns stay where they are (the sorted columns and the alignment
            // profile are indexed by the batch's residue positions), run by run of neighbouring queries
            for (size_t f0 = 0; f0 < fallback.size() && rc == MK_OK; ) {
                size_t f1 = f0 + 1;
                while (f1 < fallback.size() && fallback[f1] == fallback[f1 - 1] + 1) f1++;
                rc = global_candidates(X, V, qOff.data(), q0 + fallback[f0], q0 + fallback[f1 - 1] + 1, nullptr, (uint32_t) 0 - q0, nCand, globalHitsPerPos);
                f0 = f1;
            }
        } else if (rc == MK_OK && !fallback.empty()) {
            // ---- B. the queries the fused kernels could not take, as a compact mini batch
            const uint32_t nMini = (uint32_t) fallback.size();
            uint64_t *hMiniOff = (uint64_t *) pinned_scratch("pf_minioff_h", ((size_t) nMini + 1) * 8);
            uint32_t *hSrc = (uint32_t *) pinned_scratch("pf_minisrc_h", (size_t) nMini * 8);
            PNULL(hMiniOff); PNULL(hSrc);
            uint32_t *hMap = hSrc + nMini;
            hMiniOff[0] = 0;
            for (uint32_t i = 0; i < nMini; i++) {
                const size_t qg = (size_t) q0 + fallback[i];
                hSrc[i] = (uint32_t) qg; hMap[i] = fallback[i];
                hMiniOff[i + 1] = hMiniOff[i] + (qOff[qg + 1] - qOff[qg]);
            }
            const uint64_t total = hMiniOff[nMini];
            uint64_t *dMiniOff = (uint64_t *) dev_scratch("pf_minioff", ((size_t) nMini + 1) * 8);
            uint32_t *dSrc = (uint32_t *) dev_scratch("pf_minisrc", (size_t) nMini * 8);
            uint8_t *dMRes = (uint8_t *) dev_scratch("pf_minires", total + 16);
            int16_t *dMThr = (int16_t *) dev_scratch("pf_minithr", (total + 16) * 2);
            int8_t *dMCorr = (int8_t *) dev_scratch("pf_minicorr", total + 16);
            PNULL(dMiniOff); PNULL(dSrc); PNULL(dMRes); PNULL(dMThr); PNULL(dMCorr);
            PCHK(hipMemcpyAsync(dMiniOff, hMiniOff, ((size_t) nMini + 1) * 8, hipMemcpyHostToDevice, stream));
            PCHK(hipMemcpyAsync(dSrc, hSrc, (size_t) nMini * 8, hipMemcpyHostToDevice, stream));
            if (total > 0) {
                hipLaunchKernelGGL(gather_queries_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream, V, dSrc, dMiniOff, nMini, total, dMRes, dMThr, dMCorr);
                PCHK(hipGetLastError());
            }
            PrefilterDeviceView M = V;
            M.q_res = dMRes; M.q_off = dMiniOff; M.q_kmer_thr = dMThr; M.q_corr = dMCorr; M.n_queries = nMini;
            std::vector<uint64_t> miniOff(hMiniOff, hMiniOff + nMini + 1);     // the pinned copy may be overwritten by the next chunk
            rc = global_candidates(X, M, miniOff.data(), 0, nMini, dSrc + nMini, 0, nCand, globalHitsPerPos);
        }
        if (rc == RC_CAND_OVERFLOW) {
            if (nqc == 1) { err = "one query yields more double-diagonal candidates than the device buffers hold"; return MK_ERR_UNSUPPORTED; }
            chunkLimit = std::max<uint32_t>(1, nqc / 2);
            candPerQuery = std::max(candPerQuery, 1.0) * 2.0;
            continue;                                                      // same q0, smaller chunk
        }
        if (rc != MK_OK) return rc;
        if (hooks.stats) { hooks.stats->kmers_per_pos += cs.kmers_per_pos; hooks.stats->db_matches += cs.db_matches; hooks.stats->overflows += cs.overflows; }
        chunkLimit = QCAP;
        candPerQuery = std::max(1.0, (double) nCand / (double) nqc);

        // ---- common back end
        std::vector<uint32_t> chunkCnt(nqc, 0);
        const mk_hit *devHits = nullptr;               // device-final hits of the chunk, staged (chunks with --max-seqs queries)
        size_t nDevHits = 0;
        bool devDirect = false;                        // ... or already on their way to the result block
        std::vector<std::vector<mk_hit>> hostHits;     // per flagged query
        std::vector<uint32_t> hostQ;
        if (nCand > 0) {
            if (seqBits > 28) { err = "more than 2^28 targets: the report key of the prefilter's tail has no room"; return MK_ERR_UNSUPPORTED; }
            const uint32_t nb = (nqc + FIN_SCAN - 1u) / FIN_SCAN;
            uint8_t *dKept = (uint8_t *) dev_scratch("pf_kept", nCand);
            uint32_t *dPerQ = (uint32_t *) dev_scratch("pf_perq", (size_t) nqc * 12), *dPerQ255 = dPerQ ? dPerQ + nqc : nullptr, *dCursor = dPerQ ? dPerQ + 2 * (size_t) nqc : nullptr;
            uint32_t *dFin = (uint32_t *) dev_scratch("pf_fin", ((size_t) nqc * 7 + (size_t) nb * 8) * 4);
            uint64_t *dCutKey = (uint64_t *) dev_scratch("pf_cutkey", (size_t) nCand * 8), *dOutKey = (uint64_t *) dev_scratch("pf_okey", (size_t) nCand * 8);
            uint32_t *dNum = (uint32_t *) dev_scratch("pf_num", 64);
            uint32_t *hNum = (uint32_t *) pinned_scratch("pf_num_h", 64);
            PNULL(dKept); PNULL(dPerQ); PNULL(dFin); PNULL(dCutKey); PNULL(dOutKey); PNULL(dNum); PNULL(hNum);
            int th = tb("diag_score", 28.0 * nCand, 0);
            hipLaunchKernelGGL(diag_score_kernel, dim3((nCand + 255) / 256), dim3(256), 0, stream, V, q0, nCand, C);
            te(th);
            PCHK(hipGetLastError());
            PCHK(hipMemsetAsync(dPerQ, 0, (size_t) nqc * 12, stream));
            PCHK(hipMemsetAsync(dNum, 0, 64, stream));
            th = tb("select_hits", 35.0 * nCand, 0);
            // (min-ungapped-score 0 keeps zero-score elements under rules of their own: every query at the cut is left to the host's restatement)
            static const bool hostCutEnv = knob_long("MK_PREFILTER_HOST_MAXSEQS", 0) != 0;
            const bool hostCut = hostCutEnv || P.min_ungapped_score <= 0;
            hipLaunchKernelGGL(keep_kernel, dim3((nCand + 255) / 256), dim3(256), 0, stream, C, nCand, P.min_ungapped_score, fin_score_max(seqBits), hostCut, dKept, dPerQ, dPerQ255);
            FinishArgs F;
            F.perQ = dPerQ; F.perQ255 = dPerQ255; F.nq = nqc; F.maxHits = (uint32_t) maxHits; F.seqBits = seqBits;
            F.segOff = dFin; F.outOff = dFin + nqc; F.listW = dFin + 2 * (size_t) nqc; F.listS = dFin + 3 * (size_t) nqc; F.listB = dFin + 4 * (size_t) nqc;
            F.listH = dFin + 5 * (size_t) nqc; F.hugeTmp = dFin + 6 * (size_t) nqc; F.blockSums = dFin + 7 * (size_t) nqc; F.totals = dNum;
            hipLaunchKernelGGL(finish_sums_kernel, dim3(nb), dim3(FIN_SCAN), 0, stream, F);
            hipLaunchKernelGGL(finish_offsets_kernel, dim3(nb), dim3(FIN_SCAN), 0, stream, F);
            hipLaunchKernelGGL(finish_scatter_kernel, dim3((nCand + 255) / 256), dim3(256), 0, stream, C, nCand, dKept, F, (uint32_t) (binCount - 1), dCursor, dCutKey, dOutKey);
            PCHK(hipGetLastError());
            te(th);
            PCHK(hipMemcpyAsync(hNum, dNum, 40, hipMemcpyDeviceToHost, stream));
            uint32_t *hPerQ = (uint32_t *) pinned_scratch("pf_perq_h", (size_t) nqc * 8);
            PNULL(hPerQ);
            PCHK(hipMemcpyAsync(hPerQ, dPerQ, (size_t) nqc * 8, hipMemcpyDeviceToHost, stream));
            const uint32_t *hPerQ255 = hPerQ + nqc;
            // hits of a query in the device-final array: all survivors, or exactly max-seqs of them; 0 when the host selects
            const auto dev_count = [&](uint32_t ql) -> uint32_t { return fin_to_host(hPerQ[ql], hPerQ255[ql], (uint32_t) maxHits) ? 0u : std::min(hPerQ[ql], (uint32_t) maxHits); };
            PCHK(sync_wait(stream, "wait_prefilter"));
            const uint32_t nValid = hNum[0], nFlagged = hNum[8];
            mk_hit *dHitsOut = nullptr;
            if (nValid > 0) {
                dHitsOut = (mk_hit *) dev_scratch("pf_hits_out", (size_t) nValid * sizeof(mk_hit));
                PNULL(dHitsOut);
                th = tb("sort_query_hits", 28.0 * hNum[1], 0);
                if (hNum[2]) hipLaunchKernelGGL(finish_wave_kernel, dim3((hNum[2] + 3) / 4), dim3(256), 0, stream, F, hNum[2], dCutKey, dOutKey, dHitsOut);
                if (hNum[3]) hipLaunchKernelGGL((finish_block_kernel<64, FIN_SMALL_MAX>), dim3(hNum[3]), dim3(64), 0, stream, F, F.listS, dCutKey, dOutKey, (uint64_t *) nullptr, dHitsOut);
                if (hNum[4]) hipLaunchKernelGGL((finish_block_kernel<256, FIN_LDS_MAX>), dim3(hNum[4]), dim3(256), 0, stream, F, F.listB, dCutKey, dOutKey, (uint64_t *) nullptr, dHitsOut);
                if (hNum[5]) {
                    uint64_t *dHuge = (uint64_t *) dev_scratch("pf_hugekeys", ((size_t) hNum[6] + 1) * 8);
                    PNULL(dHuge);
                    hipLaunchKernelGGL((finish_block_kernel<1024, FIN_HUGE_TILE>), dim3(hNum[5]), dim3(1024), 0, stream, F, F.listH, dCutKey, dOutKey, dHuge, dHitsOut);
                }
                te(th);
                PCHK(hipGetLastError());
            }
            if (nFlagged == 0) {
                // every query of the chunk is final on the device: DMA the compact hit array to its final place
                if (nValid > 0) {
                    if (!reserve_out(nValid, q1)) { err = "pinned host allocation for the prefilter result failed"; return MK_ERR_DEVICE; }
                    PCHK(hipMemcpyAsync((mk_hit *) outBlk.p + nOut, dHitsOut, (size_t) nValid * sizeof(mk_hit), hipMemcpyDeviceToHost, stream));
                }
                for (uint32_t ql = 0; ql < nqc; ql++) chunkCnt[ql] = dev_count(ql);
                devDirect = true;
                nDevHits = nValid;
            } else {
                if (nValid > 0) {
                    mk_hit *hHitsOut = (mk_hit *) pinned_scratch("pf_hits_out_h", (size_t) nValid * sizeof(mk_hit));
                    PNULL(hHitsOut);
                    PCHK(hipMemcpyAsync(hHitsOut, dHitsOut, (size_t) nValid * sizeof(mk_hit), hipMemcpyDeviceToHost, stream));
                    devHits = hHitsOut; nDevHits = nValid;
                }
                for (uint32_t ql = 0; ql < nqc; ql++) chunkCnt[ql] = dev_count(ql);
                // exact reference logic for the queries that reached --max-seqs (tie order depends on BINSIZE)
                HostCand *dHC = (HostCand *) dev_scratch("pf_hostcand", (size_t) nFlagged * sizeof(HostCand));
                HostCand *hHC = (HostCand *) pinned_scratch("pf_hostcand_h", (size_t) nFlagged * sizeof(HostCand));
                PNULL(dHC); PNULL(hHC);
                hipLaunchKernelGGL(export_host_kernel, dim3((nCand + 255) / 256), dim3(256), 0, stream, C, nCand, F, dHC);
                PCHK(hipGetLastError());
                PCHK(hipMemcpyAsync(hHC, dHC, (size_t) nFlagged * sizeof(HostCand), hipMemcpyDeviceToHost, stream));
                PCHK(sync_wait(stream, "wait_prefilter"));
                ScopedHost sh("host_prefilter_maxseqs");
                // candidate runs of the flagged queries: one per query, or one per class pass of a streamed query (the runs of one
                // query need not be adjacent) -- grouped by query below
                std::vector<uint32_t> runStart;
                for (uint32_t k = 0; k < nFlagged; k++) if (k == 0 || hHC[k].q != hHC[k - 1].q) runStart.push_back(k);
                runStart.push_back(nFlagged);
                const size_t nRunsRaw = runStart.size() - 1;
                std::vector<uint32_t> runOrder(nRunsRaw);          // the merge below walks the queries in ascending order
                std::iota(runOrder.begin(), runOrder.end(), 0u);
                std::sort(runOrder.begin(), runOrder.end(), [&](uint32_t x, uint32_t y) {
                    const uint32_t qx = hHC[runStart[x]].q, qy = hHC[runStart[y]].q;
                    return qx != qy ? qx < qy : x < y;
                });
                std::vector<uint32_t> groupStart;                  // first entry of runOrder of every flagged query
                for (size_t r = 0; r < nRunsRaw; r++) if (r == 0 || hHC[runStart[runOrder[r]]].q != hHC[runStart[runOrder[r - 1]]].q) groupStart.push_back((uint32_t) r);
                groupStart.push_back((uint32_t) nRunsRaw);
                const size_t nRuns = groupStart.size() - 1;
                hostQ.resize(nRuns);
                hostHits.resize(nRuns);
                int failed = 0;
#pragma omp parallel
                {
                    std::vector<Cand> perQuery;
#pragma omp for schedule(dynamic, 4)
                    for (size_t r = 0; r < nRuns; r++) {
                        const uint32_t ql = hHC[runStart[runOrder[groupStart[r]]]].q;
                        perQuery.clear();
                        for (uint32_t g = groupStart[r]; g < groupStart[r + 1]; g++) {
                            const uint32_t k0 = runStart[runOrder[g]], k1 = runStart[runOrder[g] + 1];
                            for (uint32_t k = k0; k < k1; k++) perQuery.push_back(Cand{hHC[k].id, hHC[k].diag, hHC[k].score, hHC[k].ordinal});
                        }
                        std::sort(perQuery.begin(), perQuery.end(), [](const Cand &a, const Cand &b) { return a.ordinal < b.ordinal; });
                        const uint32_t q = q0 + ql;
                        int n255 = 0;
                        for (const Cand &c : perQuery) n255 += c.score >= 255;
                        int self = 0;
                        if (n255 >= maxHits && V.p_sorted) {       // ... of the profile's query letters against the profile itself
                            const uint32_t L = (uint32_t) (qOff[q + 1] - qOff[q]);
                            std::vector<int8_t> aln((size_t) L * PROFILE_ALN_STRIDE);
                            if (qCorrHost) std::memcpy(aln.data(), qCorrHost + qOff[q] * PROFILE_ALN_STRIDE, aln.size());
                            else {
#pragma omp critical(mk_pf_corr)
                                if (hipMemcpy(aln.data(), V.p_aln + qOff[q] * PROFILE_ALN_STRIDE, aln.size(), hipMemcpyDeviceToHost) != hipSuccess) failed = 1;
                            }
                            self = ungapped_score_profile(aln.data(), L, qRes.data() + qOff[q], L, 0u);
                        } else if (n255 >= maxHits) {              // threshold saturates: QueryMatcher.cpp:525-531 needs the exact self score
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
            // offsets first (serial, trivial), then the copies in parallel: runs of device-final queries between two
            // host-selected ones move as one block
            std::vector<size_t> runDst, runSrc, runLen;                       // device runs
            std::vector<size_t> hostDst(hostQ.size());
            {
                size_t dev = 0, hk = 0, at = nOut, runBegin = 0;
                bool open = false;
                for (uint32_t ql = 0; ql < nqc; ql++) {
                    const size_t qg = (size_t) q0 + ql;
                    if (hk < hostQ.size() && hostQ[hk] == ql) {
                        if (open) { runLen.push_back(dev - runBegin); open = false; }
                        hostDst[hk] = at;
                        at += hostHits[hk].size();
                        hk++;
                    } else {
                        const uint32_t c = chunkCnt[ql];
                        if (c && !open) { runDst.push_back(at); runSrc.push_back(dev); runBegin = dev; open = true; }
                        dev += c; at += c;
                    }
                    outOff[qg + 1] = at;
                }
                if (open) runLen.push_back(dev - runBegin);
                nOut = at;
            }
            const size_t nRunsDev = runDst.size();
#pragma omp parallel for schedule(dynamic, 1)
            for (size_t r = 0; r < nRunsDev + hostQ.size(); r++) {
                if (r < nRunsDev) std::memcpy(dst + runDst[r], devHits + runSrc[r], runLen[r] * sizeof(mk_hit));
                else if (!hostHits[r - nRunsDev].empty())
                    std::memcpy(dst + hostDst[r - nRunsDev], hostHits[r - nRunsDev].data(), hostHits[r - nRunsDev].size() * sizeof(mk_hit));
            }
        }
        if (hooks.on_chunk) {
            PCHK(sync_wait(stream, "wait_prefilter"));                     // the chunk's DMA has landed
            hooks.on_chunk(q0, q1);
        }
        q0 = q1;
    }
    { std::lock_guard<std::mutex> lk(g_memoMutex); if (g_memo.entries == (const void *) V.entries) g_memo.candPerQuery = candPerQuery; }
    (void) tOff;
    PCHK(sync_wait(stream, "wait_prefilter"));         // the last DMA into the result block
    return MK_OK;
}

}  // namespace mk
