// metaeuk_amd/csrc/mk_prefilter.hpp -- device pipeline for the k-mer prefilter
// (QueryMatcher::match + findDuplicates + UngappedAlignment::align on the GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>
#include "../../include/metaeuk_amd.h"

namespace mk {

struct PrefilterDeviceView {
    // query batch
    const uint8_t *q_res; const uint64_t *q_off; const int16_t *q_kmer_thr; const int8_t *q_corr; uint32_t n_queries;
    // target side
    const uint8_t *t_masked; const uint64_t *t_off; uint32_t n_targets;
    // one 8-byte slot per k-mer: a single-entry list is stored inline (bit 63 | the entry: target | position << 32), any other list as
    // first entry index | length << 32 -- the most frequent case costs one dependent random read instead of two
    const uint64_t *kmer_slot; const uint64_t *entries;
    const uint32_t *kmer_bits;     // 20^6 bits: k-mer has a non-empty index list (8 MB: stays in L2/MALL, filters the offset probes)
    const int16_t *score3; const uint16_t *index3;
    const uint16_t *hist3; const uint16_t *cum3; int hist_lo, hist_range;   // per-row score histograms (sizing)
    uint64_t n_entries;
    const int8_t *mat_ung;
    // profile queries (mk_profile.hpp; null for sequence queries): q_res = the profiles' query letters, q_kmer_thr as usual, q_corr unused
    const int8_t *p_sorted = nullptr;     // [column][40]: the 20 scores descending + their residue numbers
    const int8_t *p_aln = nullptr;        // [column][32]: the alignment profile (score / 4), what the diagonal scoring reads
    const uint16_t *addr3 = nullptr;      // 3-mer number -> address code of the index table (kmer3_address_table)
    // k = 7 (sequence queries against a database of 3.35e9 residues or more, or -k 7): 2-mer rows and the code -> 3-mer number map; the
    // k-mer table then has 20^7 cells in the reference's numbering and the similar k-mers come as lists too (mk_kmer7.hpp)
    int kmer_size = 6;
    const int16_t *score2 = nullptr; const uint16_t *index2 = nullptr; const uint16_t *num3 = nullptr;
    // the similar k-mers of the k-mer starts [klist_pos0, ...) as lists in HBM (filled per piece by the global path for profile queries)
    const uint32_t *klist = nullptr; const uint64_t *klist_off = nullptr; uint64_t klist_pos0 = 0;
};

typedef int (*timed_begin_fn)(const char *name, double bytes, double cells);
typedef void (*timed_end_fn)(int handle);
typedef void (*timed_set_fn)(int handle, double bytes, double cells);

// what the reference's prefilter logs about a run (Prefiltering.cpp:889-904,953-975), accumulated over the calls of a batch
struct PrefilterStats {
    double kmers_per_pos = 0;        // sum over the queries of (similar k-mers of the query / its length)
    uint64_t db_matches = 0;         // index entries gathered
    uint64_t overflows = 0;          // queries that filled the reference's databaseHits buffer (QueryMatcher.cpp:281-334)
};

// optional hooks for a caller that consumes the result chunk by chunk while the prefilter is still running (mk_search)
struct PrefilterHooks {
    uint32_t max_chunk_queries = 0;                                  // 0: no limit beyond the device buffers
    bool chunk_ramp = false;                                         // the first chunks are smaller (1/4, 1/2 of the limit): the consumer starts sooner
    int max_tiers = 0;                                               // > 0: use only the first tiers of the per-query front end
    bool co_resident = false;                                        // another stage (the Smith-Waterman waves of mk_search) shares the CUs: the
                                                                     // persistent prefilter workgroups take about half of the wave slots
    std::function<const uint8_t *()> t_masked_host;                  // host copy of the masked target residues (fetched on demand): the overflow path of a
                                                                     // query that fills the reference's databaseHits buffer scores a few diagonals on the host
    PrefilterStats *stats = nullptr;                                 // accumulates the run statistics when set
    std::function<void(uint32_t q0, uint32_t q1)> on_chunk;          // hits and offsets of [q0, q1) are final and in host memory
    std::function<void()> before_grow;                               // the result block is about to be re-allocated
};

// Runs the whole prefilter for the batch.  q_off_host / t_off_host mirror the device offset arrays.
int run_prefilter(const PrefilterDeviceView &V, const std::vector<uint64_t> &q_off_host, const std::vector<uint8_t> &q_res_host,
                  const int8_t *q_corr_host /* sequence queries: the int8 diagonal correction; profile queries: the alignment profile [column][32] */,
                  const std::vector<uint64_t> &t_off_host, const mk_params &P, int binCount, hipStream_t stream,
                  struct HostBlock &outHits, size_t &nOutHits, std::vector<uint64_t> &outOff, std::string &err, timed_begin_fn tb, timed_end_fn te, timed_set_fn ts,
                  const PrefilterHooks &hooks);

}  // namespace mk
