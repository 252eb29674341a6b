// metaeuk_amd/csrc/mk_exons.hpp -- exon sets from the alignments of a contig's ORF fragments
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/metaeuk_amd.h"

namespace mk {

void default_exon_params(mk_exon_params &P);
// orfs[k] = query k of the batch the alignments belong to; alns[alnOff[k] .. alnOff[k+1]) = its accepted alignments.
// targetKeys (may be null) maps a target index to the key the predictions are ordered by and carry.
// preds[contigOff[c] .. contigOff[c+1]) = predictions of contig c in the reference's order (target key ascending, plus strand first).
void predict_exons(const mk_orf *orfs, uint64_t nOrfs, uint32_t nContigs, const mk_alignment *alns, const uint64_t *alnOff, const uint32_t *targetKeys,
                   uint64_t dbResidues, const mk_exon_params &P, std::vector<mk_prediction> &preds, std::vector<uint64_t> &contigOff, std::vector<mk_exon> &exons);
size_t format_prediction_exon(char *buf, const mk_prediction &p, const mk_exon &e);

}  // namespace mk
