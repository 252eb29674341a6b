// metaeuk_amd/csrc/mk_kmer7.hpp -- similar k-mer lists for k = 7 (the reference's k-mer size for databases from 3.35e9 residues on,
// M/src/prefiltering/IndexTable.h:439-449; spaced seed 11010110011, M/src/commons/Sequence.h:25).
// KmerGenerator::setDivideStrategy (M/src/prefiltering/KmerGenerator.cpp:41-86) cuts a 7-mer into a 2-mer, a 2-mer and a 3-mer
// (window positions 0-1, 2-3, 4-6; multipliers 20^0, 20^2, 20^4); generateKmerList (:107-187) multiplies the rows step by step, keeping
// the partial list in order -- i.e. the list is in lexicographic order of the three ranks, pruned at every step against the best of the
// remaining rows.  As for profile queries (mk_profile.hpp) the lists are materialised in HBM and walked by the list-driven probe kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mk {

struct Kmer7Tables {
    const int16_t *score2; const uint16_t *index2;     // [400][400] similar 2-mers, descending; index = 2-mer number
    const int16_t *score3; const uint16_t *index3;     // [8000][8000] similar 3-mers (index = table address code of the k = 6 path)
    const uint16_t *num3;                              // address code -> 3-mer number
    const uint16_t *cum3; int hist_lo, hist_range;     // per 3-mer row: entries with score >= hist_lo + x
};

// counts[p - posBegin] = list length of the k-mer start p (thr < 0: no start)
hipError_t launch_kmer7_count(const Kmer7Tables &T, const uint8_t *dRes, const int16_t *dKthr, uint64_t posBegin, uint64_t posEnd, uint32_t *dCounts, hipStream_t stream);
// list[listOff[p - posBegin] ..] = the k-mers' table cells n2a + 400 n2b + 160000 n3, in list order
hipError_t launch_kmer7_fill(const Kmer7Tables &T, const uint8_t *dRes, const int16_t *dKthr, uint64_t posBegin, uint64_t posEnd,
                             const uint64_t *dListOff, uint32_t *dList, hipStream_t stream);

}  // namespace mk
