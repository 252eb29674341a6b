// metaeuk_amd/csrc/mk_indexfile.hpp -- the precomputed target index of `createindex`: writer and reader of
// the reference's index DB (type 9), M/src/prefiltering/PrefilteringIndexReader.cpp:10-326 (layout), :356-437 (what a reader takes).
// Host code, no device needed: the file holds what mk_targetdb_create computes on the host (masked sequences, k-mer lists), so a
// database indexed once -- by this library or by the reference -- is uploaded without being masked and indexed again.
#pragma once
#include <cstdint>
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
};

// text of a substitution matrix in the reference's .out format (parsed back by SubstitutionMatrix::readProbMatrix to the same numbers)
std::string matrix_text(int which);

// "" on success, else the error.  base = path of the index DB (<targetDB>.idx); writes base, base.index, base.dbtype
std::string write_index_file(const std::string &base, const SubMat &kmerMat, const IndexFileContent &c);
std::string read_index_file(const std::string &base, IndexFileContent &c);

}  // namespace mk
