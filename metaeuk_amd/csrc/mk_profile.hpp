// metaeuk_amd/csrc/mk_profile.hpp -- profile queries: the reference's profile-target search
// (M/data/workflow/searchslicedtargetprofile.sh) makes the profiles the QUERIES of prefilter / align and the 6-frame fragments the
// indexed targets.  Device layout of a batch of profiles and the kernels that derive it and list its similar k-mers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mk {

constexpr int PROFILE_COL_BYTES = 25;      // Sequence::PROFILE_READIN_SIZE (M/src/commons/Sequence.h:471): 20 scores, query letter, consensus, neff, 2 gap bytes
constexpr int PROFILE_SORTED_STRIDE = 40;  // per column: 20 scores descending (int8) + their 20 residue numbers
constexpr int PROFILE_ALN_STRIDE = 32;     // per column: score / 4 per residue, X (20) and the "no column" code 21 score 0

// Sequence::mapProfile (M/src/commons/Sequence.cpp:241-292) for every column of the batch (raw = the 25-byte columns, concatenated):
//   letters[p]  the profile's query letter (numSequence: what Sequence::kmerContainsX looks at)
//   sorted[p]   Util::rankedDescSort20 of the 20 scores with their residue numbers (the exchange network of Util.cpp:88-114, ties included)
//   aln[p]      profile_for_alignment: score / 4, C division
//   kthr[p]     k-mer threshold of the k-mer start p (QueryMatcher.cpp:225-244 with a zero composition bias: max(kmerThr, 0)); -1: no start
hipError_t launch_profile_derive(const uint8_t *dRaw, const uint64_t *dOff, uint32_t nProfiles, uint64_t totalCols, int kmerThr, int kmerSize,
                                 uint8_t *dLetters, int8_t *dSorted, int8_t *dAln, int16_t *dKthr, hipStream_t stream);
// kthr alone, for the other k-mer size (a batch derived for k = 6 that meets a k = 7 database, or the other way round): span 11 and the
// seed 11010110011 for k = 7, threshold 149.15 - 6.85 s (Prefiltering.cpp:1041-1043)
hipError_t launch_profile_kthr(const uint8_t *dLetters, const uint64_t *dOff, uint32_t nProfiles, uint64_t totalCols, int kmerThr, int kmerSize, int16_t *dKthr, hipStream_t stream);

// Similar k-mers of the k-mer starts [posBegin, posEnd) (KmerGenerator::generateKmerList with the profile divide strategy,
// M/src/prefiltering/KmerGenerator.cpp:30-39,107-216): six columns under the spaced pattern, one position per step, a partner taken
// while score_j >= threshold - partial - best of the remaining columns.  The reference's list order (step by step, the partial lists
// kept in order) is the lexicographic order of the rank tuples; a thread walks its start depth first in exactly that order.
//   count: counts[p - posBegin] = list length;  fill: list[listOff[p - posBegin] ..] = the k-mers' index-table cells
// kmerSize 7: seven columns under 11010110011, cells in the reference's numbering (the 20^7-cell table of a database of 3.35e9 residues or more)
hipError_t launch_profile_kmer_count(const int8_t *dSorted, const int16_t *dKthr, uint64_t posBegin, uint64_t posEnd, uint32_t *dCounts, hipStream_t stream, int kmerSize = 6);
hipError_t launch_profile_kmer_fill(const int8_t *dSorted, const int16_t *dKthr, const uint16_t *dAddr3, uint64_t posBegin, uint64_t posEnd,
                                    const uint64_t *dListOff, uint32_t *dList, hipStream_t stream, int kmerSize = 6);

}  // namespace mk
