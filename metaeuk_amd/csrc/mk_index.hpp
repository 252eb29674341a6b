// metaeuk_amd/csrc/mk_index.hpp -- the target side built ON THE DEVICE: tantan masking and the k-mer index of
// IndexBuilder::fillDatabase (M/src/prefiltering/IndexBuilder.cpp:55-239) as HIP kernels, so that a database is masked and indexed
// at HBM speed and never exists twice in host memory (a UniRef50-scale database: 1.8e10 residues, 1.4e11 bytes of entries).
// Results are byte-identical to the host builder mk::build_index (mk_host.cpp), which stays as the no-GPU writer of index DBs and as
// the check of this one (tests/test_gpu_index.py, MK_INDEX_BUILD=host).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>
#include "mk_host.hpp"
#include "mk_prefilter.hpp"

namespace mk {

// what the builder leaves in HBM; the caller owns the allocations (hipFree)
struct DeviceIndex {
    uint8_t *masked = nullptr;      // [total residues]
    uint64_t *slots = nullptr;      // [cells]: single-entry list inline (bit 63 | entry), else first entry (40 bits, + entry_shift) | length << 40
    uint32_t *bits = nullptr;       // [(cells + 31) / 32]: list non-empty
    uint64_t *entries = nullptr;    // [n_entries]: target | position << 32, every list ascending by target
    uint64_t cells = 0, n_entries = 0, masked_residues = 0;
    uint32_t max_list = 0;          // longest list
    void release();
};

struct IndexBuildParams {
    int kmer_size = 6;              // 6: cells in the tiled address order (mk_host.cpp, KMER_ADDR_LETTER) unless reference_order; 7: the reference's numbering
    int kmer_thr = 0;               // self-score filter of the index (0: none, the profile search)
    bool mask = true;
    double mask_prob = 0.9;
    int tantan_lanes = 4;           // doubles per SIMD register of the reference build being reproduced
    bool reference_order = false;   // k = 6 cells in the reference's numbering (what an index DB holds)
    uint64_t entry_shift = 0;       // MK_TEST_ENTRY_BASE
};

// dRes / dOff: the unmasked residues and offsets in HBM; offHost mirrors dOff
int device_build_index(const uint8_t *dRes, const uint64_t *dOff, const std::vector<uint64_t> &offHost, uint32_t nSeq, const SubMat &kmerMat,
                       const IndexBuildParams &P, hipStream_t stream, DeviceIndex &out, std::string &err, timed_begin_fn tb, timed_end_fn te);

// the same tables from k-mer lists that exist already (an index DB): dCount[cells] = list lengths in the cells' order, dEntriesIn = the lists
// back to back in that order, every list sorted.  Takes ownership of dEntriesIn (it becomes out.entries).
int device_index_from_lists(uint32_t *dCount, uint64_t *dEntriesIn, uint64_t cells, uint64_t nEntries, uint64_t entryShift, hipStream_t stream,
                            DeviceIndex &out, std::string &err);

// index DB -> device (cells in the reference's numbering: what the device tables use for k = 7): list lengths from the offsets, the
// 6-byte IndexEntryLocal records expanded to 8 bytes.  hostOffsets[cells + 1], hostEntries6 = nEntries * 6 bytes (both may be mmap'd
// file contents: they are read once, in pieces)
// The file is not trusted: offsets that fall, lists of 2^23 entries or more and entries naming a sequence >= nSeq are an error, not a crash.
int device_index_from_file(const uint64_t *hostOffsets, const unsigned char *hostEntries6, uint64_t nEntries, int kmerSize, uint64_t entryShift,
                           uint32_t nSeq, hipStream_t stream, DeviceIndex &out, std::string &err);

// device index (cells in the reference's numbering) -> the arrays of an index DB: the k-mer list offsets (offsets[cells + 1]) ...
int device_index_offsets(const DeviceIndex &ix, hipStream_t stream, std::vector<uint64_t> &offsets, std::string &err);
// ... and the entries as 6-byte IndexEntryLocal records through `sink(ptr, bytes)`, in list order, piece by piece
int device_index_entries6(const DeviceIndex &ix, hipStream_t stream, const std::function<bool(const void *, size_t)> &sink, std::string &err);

// test hook: number of differing words / entries between two indices (slots, bits, entries, masked residues)
int device_index_compare(const DeviceIndex &a, const DeviceIndex &b, uint64_t totalResidues, hipStream_t stream, uint64_t diff[4], std::string &err);

}  // namespace mk
