// metaeuk_amd/csrc/mk_align.hpp -- device-resident pipeline for the gapped alignment stage
// (Alignment::run's per-pair work: Matcher::getSWResult -> SmithWaterman::ssw_align, on the GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/metaeuk_amd.h"
#include "mk_host.hpp"
#include "mk_prefilter.hpp"

namespace mk {

struct AlignView {
    const uint8_t *q_res; const int8_t *q_bias8; const uint64_t *q_off; uint32_t n_queries;
    const int8_t *q_prof = nullptr;   // profile queries: [column][32] alignment profile
    const uint8_t *t_res; const uint64_t *t_off; uint32_t n_targets;
    const int8_t *mat_aln;
    uint32_t max_q_len, max_t_len;
    bool co_resident = false;      // the prefilter of mk_search shares the CUs: kernels with a small LDS footprint are preferred
};

// integer result of one accepted pair (everything else is derived on the host in double/float)
struct AlnRaw { uint32_t pair; int32_t score, q_end, t_end, q_start, t_start; };

// e-value gate as a per-query-length table: pass(score) = score >= s0 || bit(score) for score < 256
// (ssw_align_private's `evalue > evalueThr` early return, StripedSmithWaterman.cpp:390-398)
struct GateEntry { int32_t s0; uint32_t mask[8]; };
struct AssembleTables;
void build_gate_table(const Evaluer &ev, double evalThr, const std::vector<uint64_t> &qOff, std::vector<GateEntry> &table, const AssembleTables *T = nullptr);

// Matcher::getSWResult's float/double tail (Matcher.cpp:60-164), Alignment::checkCriteria and the per-query sort on the device.
// The e-value is the one transcendental quantity: it comes from a table the host fills with the reference's double arithmetic
// (evalue[lenIdx[qLen] * smax + score]); everything else is correctly rounded float/double add, multiply, divide and integers.
struct AssembleTables {
    std::vector<double> evalue;      // per query length present in the batch: evalue(score, L) for score < smax
    std::vector<int32_t> lenIdx;     // query length -> row of `evalue` (-1: no query of that length)
    std::vector<int32_t> bitScore;   // static_cast<int>(bitScore(score) + 0.5) for score < 32768
    uint32_t smax = 0;
    uint64_t id = 0;                 // changes with every build (the device copy is refreshed when it differs)
};
// rowCache (optional): e-value rows by query length that were computed before for the same database; missing ones are added
void build_assemble_tables(const Evaluer &ev, const std::vector<uint64_t> &qOff, AssembleTables &t,
                           std::unordered_map<uint32_t, std::vector<double>> *rowCache = nullptr);
struct AssembleArgs {
    const AssembleTables *tables = nullptr;      // null: no device assembly (the caller gets AlnRaw records)
    const uint32_t *dSortKey = nullptr;          // device array: DB key of every target for the last tie-break of the sort (null: the target index)
    // destination for the accepted alignments of the range, asked for once their number is known (pinned host memory)
    std::function<mk_alignment *(size_t n)> reserve;
    uint32_t *counts = nullptr;                  // host array [n_queries]: accepted alignments of every query of the range
    size_t nOut = 0;                             // records written at the reserved destination
    bool done = false;                           // false after the call: a score beyond the table, the caller assembles from AlnRaw
    double revWork[2 * 16] = {0};                // reverse-pass work per tile configuration (bytes, cells), statistics
};

// pairs = (query of pair p is the one whose hitOff range contains p, target hits[p].seq_id); hits on the host.
// out: accepted pairs (those passing the e-value gate) with start positions, ordered by pair index; the array is
// pinned scratch owned by the library and valid until the next call.
int run_align_device(const AlignView &V, const uint64_t *hitOffHost, const mk_hit *hitsHost, uint64_t nPairs,
                     const std::vector<GateEntry> &gate, const mk_params &P, hipStream_t stream,
                     const double *fwdWork /* per tile configuration: algorithmic bytes, cells (2*SW_NCFG) or null */,
                     const AlnRaw **out, size_t *nOut, std::string &err, timed_begin_fn tb, timed_end_fn te, timed_set_fn ts,
                     AssembleArgs *assemble = nullptr);

}  // namespace mk
