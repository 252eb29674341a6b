// metaeuk_amd/csrc/mk_align.hpp -- device-resident pipeline for the gapped alignment stage
// (Alignment::run's per-pair work: Matcher::getSWResult -> SmithWaterman::ssw_align, on the GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/metaeuk_amd.h"
#include "mk_host.hpp"
#include "mk_prefilter.hpp"

namespace mk {

struct AlignView {
    const uint8_t *q_res; const int8_t *q_bias8; const uint64_t *q_off; uint32_t n_queries;
    const uint8_t *t_res; const uint64_t *t_off; uint32_t n_targets;
    const int8_t *mat_aln;
    uint32_t max_q_len, max_t_len;
};

// integer result of one accepted pair (everything else is derived on the host in double/float)
struct AlnRaw { uint32_t pair; int32_t score, q_end, t_end, q_start, t_start; };

// e-value gate as a per-query-length table: pass(score) = score >= s0 || bit(score) for score < 256
// (ssw_align_private's `evalue > evalueThr` early return, StripedSmithWaterman.cpp:390-398)
struct GateEntry { int32_t s0; uint32_t mask[8]; };
void build_gate_table(const Evaluer &ev, double evalThr, const std::vector<uint64_t> &qOff, std::vector<GateEntry> &table);

// pairs = (query of pair p is the one whose hitOff range contains p, target hits[p].seq_id); hits on the host.
// out: accepted pairs (those passing the e-value gate) with start positions, ordered by pair index; the array is
// pinned scratch owned by the library and valid until the next call.
int run_align_device(const AlignView &V, const uint64_t *hitOffHost, const mk_hit *hitsHost, uint64_t nPairs,
                     const std::vector<GateEntry> &gate, const mk_params &P, hipStream_t stream,
                     const double *fwdWork /* per tile configuration: algorithmic bytes, cells (2*SW_NCFG) or null */,
                     const AlnRaw **out, size_t *nOut, std::string &err, timed_begin_fn tb, timed_end_fn te, timed_set_fn ts);

}  // namespace mk
