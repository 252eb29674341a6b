// metaeuk_amd/csrc/mk_indexfile.hpp -- the precomputed target index of `createindex`: writer and reader of
// the reference's index DB (type 9), M/src/prefiltering/PrefilteringIndexReader.cpp:10-326 (layout), :356-437 (what a reader takes).
// Host code, no device needed: the file holds what mk_targetdb_create computes on the host (masked sequences, k-mer lists), so a
// database indexed once -- by this library or by the reference -- is uploaded without being masked and indexed again.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>
#include "mk_host.hpp"

namespace mk {

// the sequence DB an index was made from, in the order of its .index file (ids of the index = positions in this order)
struct SeqDbImage {
    std::vector<uint32_t> keys;
    std::vector<uint64_t> offsets;       // into `data`
    std::vector<uint32_t> lengths;       // entry length including "\n\0"
    std::vector<char> data;              // the data file(s)
    int dbtype = 0;
};

struct IndexFileMeta {                   // META (PrefilteringIndexReader.cpp:84-92)
    int maxSeqLen = 65535, kmerSize = KMER, compBiasCorr = 1, alphabetSize = ALPH, mask = 1, spacedKmer = 1, kmerThr = 0,
        seqType = 0, srcSeqType = 0, headers1 = 0, headers2 = 0, splits = 1;
};

struct IndexFileContent {
    IndexFileMeta meta;
    std::string matrixName;              // "VTML80.out"
    SeqDbImage seqs;
    TargetIndex index;                   // reference numbering of the k-mers (Indexer::int2index); index.masked = SequenceLookup data
    std::vector<uint64_t> seqOffsets;    // of the masked residues, n + 1
    // read_index_file(..., viewLists = true): the k-mer lists and the masked residues stay in the mapped file (a k = 7 index: 10 GB of
    // offsets, 6 bytes per entry) -- index.offsets / index.entries / index.masked are empty, these point into `mapping`
    const uint64_t *listOffsets = nullptr;        // [cells + 1]
    const unsigned char *listEntries6 = nullptr;  // nEntries IndexEntryLocal records
    uint64_t nEntries = 0;
    const uint8_t *maskedView = nullptr;
    std::shared_ptr<void> mapping;
};

// the k-mer lists and masked residues as a writer consumes them: from host vectors, or streamed out of HBM
struct IndexListSource {
    uint64_t cells = 0, nEntries = 0;
    std::function<const uint64_t *()> offsets;                                                   // [cells + 1], the reference's k-mer numbering
    std::function<bool(const std::function<bool(const void *, size_t)> &sink)> entries6;        // IndexEntryLocal records in list order, in pieces
    const uint8_t *masked = nullptr; uint64_t maskedSize = 0;
};

uint64_t kmer_table_cells(int kmerSize);       // 20^6, 20^7

// text of a substitution matrix in the reference's .out format (parsed back by SubstitutionMatrix::readProbMatrix to the same numbers)
std::string matrix_text(int which);

// "" on success, else the error.  base = path of the index DB (<targetDB>.idx); writes base, base.index, base.dbtype
std::string write_index_file(const std::string &base, const SubMat &kmerMat, const IndexFileContent &c);      // lists = c.index
std::string write_index_file(const std::string &base, const SubMat &kmerMat, const IndexFileContent &c /* meta, seqs, seqOffsets */, const IndexListSource &lists);
std::string read_index_file(const std::string &base, IndexFileContent &c, bool viewLists = false);
void materialize_lists(IndexFileContent &c);   // views -> c.index (host vectors)

}  // namespace mk
