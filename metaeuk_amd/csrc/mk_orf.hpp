// metaeuk_amd/csrc/mk_orf.hpp -- ORF extraction + translation on the device (extractorfs --translate)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include "../../include/metaeuk_amd.h"
#include "mk_host.hpp"

namespace mk {

// one fragment: position on its strand (s_from = first nucleotide in strand coordinates), length in codons,
// flags: 1 = incomplete start, 2 = incomplete end (no stop codon), 4 = minus strand
struct OrfRecord { uint32_t contig; uint32_t s_from; uint32_t n_aa; uint32_t flags; };

struct OrfScanArgs {
    const char *nucl; const uint64_t *offsets; uint32_t n_contigs;
    uint32_t min_length, max_length; uint64_t max_gaps;              // codons (Orf::findAll's minLength / maxLength / maxGaps)
    OrfRecord *records; uint64_t *aa_off;
};

struct OrfDeviceResult {
    uint64_t n_frag = 0, n_aa = 0;
    OrfRecord *records = nullptr; uint64_t *aa_off = nullptr;        // [n_frag], [n_frag + 1]
    char *aa_ascii = nullptr; uint8_t *aa_code = nullptr;            // [n_aa]
    size_t cap[4] = {0, 0, 0, 0};                                    // the four blocks come from the batch pool (mk::dev_block_alloc)
    void release();
};

// contigs (ASCII, concatenated, offsets[n+1]) already in HBM -> fragments in the reference's output order
int run_extract_orfs(const char *dNucl, const uint64_t *dOffsets, uint32_t nContigs, uint32_t minLength, uint32_t maxLength, uint64_t maxGaps,
                     hipStream_t stream, OrfDeviceResult &R, std::string &err);

}  // namespace mk
