// metaeuk_amd/csrc/mk_indexfile.cpp -- writer / reader of the reference's precomputed index DB (see mk_indexfile.hpp).
// Layout restated from PrefilteringIndexReader::createIndexFile (M/src/prefiltering/PrefilteringIndexReader.cpp:54-326):
// one DB (type 9) whose entries are keyed by fixed numbers; every entry is followed by a NUL (DBWriter::writeEnd) and padded to
// the next page (DBWriter::alignToPageSize, DBWriter.cpp:430-443) so that a reader can use the arrays where they are mapped.
#include "mk_indexfile.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace mk {

#include "../data/matrices.inc"

namespace {

enum : uint32_t {   // PrefilteringIndexReader.cpp:10-34
    K_VERSION = 0, K_META = 1, K_SCOREMATRIXNAME = 2, K_SCOREMATRIX2MER = 3, K_SCOREMATRIX3MER = 4, K_DBR1INDEX = 5, K_DBR1DATA = 6,
    K_DBR2INDEX = 7, K_DBR2DATA = 8, K_ENTRIES = 9, K_ENTRIESOFFSETS = 10, K_ENTRIESNUM = 12, K_SEQCOUNT = 13, K_SEQINDEXDATA = 14,
    K_SEQINDEXDATASIZE = 15, K_SEQINDEXSEQOFFSET = 16, K_GENERATOR = 22, K_SPACEDPATTERN = 23
};
const char INDEX_VERSION[] = "16";       // MMSEQS_CURRENT_INDEX_VERSION (M/src/MMseqsBase.cpp:6)
const int DBTYPE_INDEX_DB = 9;           // Parameters.h:77
const size_t PAGE = 4096;
const size_t ROW3 = (8000 / 64 + 1) * 64, ROW2 = (400 / 64 + 1) * 64;   // ScoreMatrix rows, MAX_ALIGN_INT = 64 (ScoreMatrix.h:45-47)

// DBReader<unsigned int>::Index (DBReader.h:58-62) as the compiler lays it out: id, pad, offset, length, pad
struct SerializedIndexEntry { uint32_t id; uint32_t pad0; uint64_t offset; uint32_t length; uint32_t pad1; };
static_assert(sizeof(SerializedIndexEntry) == 24, "DBReader::Index is 24 bytes");

struct Writer {
    FILE *f = nullptr;
    uint64_t off = 0;
    struct Rec { uint32_t key; uint64_t offset, length; };
    std::vector<Rec> recs;
    bool ok = true;
    void raw(const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) ok = false; off += n; }
    uint64_t begin() const { return off; }
    void end(uint32_t key, uint64_t start) {                   // writeEnd: NUL, index entry; then alignToPageSize
        const char z = 0;
        raw(&z, 1);
        recs.push_back(Rec{key, start, off - start});
        static const char zeros[PAGE] = {0};
        if (off % PAGE) raw(zeros, PAGE - off % PAGE);
    }
    void put(uint32_t key, const void *p, size_t n) { const uint64_t s = begin(); raw(p, n); end(key, s); }
};

// ExtendedSubstitutionMatrix::calcScoreMatrix (ExtendedSubstitutionMatrix.cpp:20-69) for k = 2: rows by the reference's index,
// candidates stable-sorted by descending score over the cartesian order with the first letter slowest; padding -255 / 0
void scorematrix2(const SubMat &km, std::vector<int16_t> &score, std::vector<uint32_t> &index) {
    score.assign(400 * ROW2, -255);
    index.assign(400 * ROW2, 0);
    std::vector<std::pair<short, uint32_t>> tmp(400);
    for (int i0 = 0; i0 < 20; i0++)
        for (int i1 = 0; i1 < 20; i1++) {
            for (int a0 = 0; a0 < 20; a0++)
                for (int a1 = 0; a1 < 20; a1++)
                    tmp[a0 * 20 + a1] = {static_cast<short>(km.sub[i0][a0] + km.sub[i1][a1]), static_cast<uint32_t>(a0 + 20 * a1)};
            std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<short, uint32_t> &l, const std::pair<short, uint32_t> &r) { return l.first > r.first; });
            const size_t row = static_cast<size_t>(i0 + 20 * i1) * ROW2;
            for (int z = 0; z < 400; z++) { score[row + z] = tmp[z].first; index[row + z] = tmp[z].second; }
        }
}

struct Mapped {
    char *p = nullptr; size_t n = 0;
    Mapped() = default;
    Mapped(const Mapped &) = delete;
    Mapped &operator=(const Mapped &) = delete;
    ~Mapped() { if (p) munmap(p, n); }
    std::string open(const std::string &path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return "cannot open " + path;
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); return "cannot stat " + path; }
        n = static_cast<size_t>(st.st_size);
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); n = 0; return "cannot map " + path; }
            p = static_cast<char *>(m);
        }
        close(fd);
        return "";
    }
};

}  // namespace

uint64_t kmer_table_cells(int kmerSize) { return kmerSize == 7 ? 1280000000ull : 64000000ull; }

std::string matrix_text(int which) {
    const bool bl = which == MAT_BLOSUM62;
    const double (*S)[ALPH] = bl ? MK_BLOSUM62_SCORES : MK_VTML80_SCORES;
    const double *bg = bl ? MK_BLOSUM62_BACKGROUND : MK_VTML80_BACKGROUND;
    const char *alpha = bl ? MK_BLOSUM62_ALPHABET : MK_VTML80_ALPHABET;
    std::string t = bl ? "# BLOSUM62\n" : "# VTML80\n";
    char buf[64];
    t += "# Background (precomputed optional):";
    for (int i = 0; i < ALPH; i++) { snprintf(buf, sizeof(buf), " %.5f", bg[i]); t += buf; }
    snprintf(buf, sizeof(buf), "\n# Lambda     (precomputed optional): %.5f\n", bl ? MK_BLOSUM62_LAMBDA : MK_VTML80_LAMBDA);
    t += buf;
    t += "  ";
    for (int i = 0; i < ALPH; i++) { t += ' '; t += alpha[i]; }
    t += '\n';
    for (int i = 0; i < ALPH; i++) {
        t += alpha[i];
        for (int j = 0; j < ALPH; j++) { snprintf(buf, sizeof(buf), " %.4f", S[i][j]); t += buf; }
        t += '\n';
    }
    return t;
}

std::string write_index_file(const std::string &base, const SubMat &km, const IndexFileContent &c) {
    IndexListSource src;
    src.cells = kmer_table_cells(c.meta.kmerSize);
    if (c.index.offsets.size() != src.cells + 1) return "index content is inconsistent";
    src.nEntries = c.index.entries.size();
    src.offsets = [&c]() { return c.index.offsets.data(); };
    src.entries6 = [&c](const std::function<bool(const void *, size_t)> &sink) {
        const size_t ne = c.index.entries.size();
        std::vector<unsigned char> buf;
        const size_t CH = 1u << 20;
        for (size_t b = 0; b < ne; b += CH) {
            const size_t e = std::min(ne, b + CH);
            buf.resize((e - b) * 6);
            for (size_t k = b; k < e; k++) {
                const uint32_t seq = static_cast<uint32_t>(c.index.entries[k]);
                const uint16_t pos = static_cast<uint16_t>(c.index.entries[k] >> 32);
                std::memcpy(buf.data() + (k - b) * 6, &seq, 4); std::memcpy(buf.data() + (k - b) * 6 + 4, &pos, 2);
            }
            if (!sink(buf.data(), buf.size())) return false;
        }
        return true;
    };
    src.masked = c.index.masked.data(); src.maskedSize = c.index.masked.size();
    return write_index_file(base, km, c, src);
}

std::string write_index_file(const std::string &base, const SubMat &km, const IndexFileContent &c, const IndexListSource &src) {
    const size_t n = c.seqs.keys.size();
    const size_t TABLE = src.cells;
    if (TABLE != kmer_table_cells(c.meta.kmerSize) || c.seqOffsets.size() != n + 1) return "index content is inconsistent";
    remove((base + ".dbtype").c_str());
    Writer w;
    w.f = fopen(base.c_str(), "wb");
    if (!w.f) return "cannot create " + base;
    w.put(K_VERSION, INDEX_VERSION, strlen(INDEX_VERSION));
    const int meta[12] = {c.meta.maxSeqLen, c.meta.kmerSize, c.meta.compBiasCorr, c.meta.alphabetSize, c.meta.mask, c.meta.spacedKmer, c.meta.kmerThr,
                          c.meta.seqType, c.meta.srcSeqType, c.meta.headers1, c.meta.headers2, c.meta.splits};
    w.put(K_META, meta, sizeof(meta));
    {   // BaseMatrix::serialize (BaseMatrix.cpp:170-187): name ':' text of the matrix file
        const std::string s = km.name + ":" + matrix_text(MAT_VTML80);
        w.put(K_SCOREMATRIXNAME, s.data(), s.size());
    }
    w.put(K_SPACEDPATTERN, "", 0);                                     // written when the pattern is empty (:98-102)
    {
        const std::string gen = "metaeuk_amd";
        w.put(K_GENERATOR, gen.data(), gen.size());
    }
    // DBReader::serialize (DBReader.cpp:944-972)
    uint64_t offIndex, lenIndex, offData, lenData;
    {
        uint64_t size = n, dataSize = 0;
        uint32_t lastKey = 0, maxLen = 0;
        for (size_t i = 0; i < n; i++) { dataSize += c.seqs.lengths[i]; lastKey = std::max(lastKey, c.seqs.keys[i]); maxLen = std::max(maxLen, c.seqs.lengths[i]); }
        const int dbtype = c.seqs.dbtype;
        offIndex = w.begin();
        w.raw(&size, 8); w.raw(&dataSize, 8); w.raw(&lastKey, 4); w.raw(&dbtype, 4); w.raw(&maxLen, 4);
        std::vector<SerializedIndexEntry> ser(n);
        for (size_t i = 0; i < n; i++) ser[i] = SerializedIndexEntry{c.seqs.keys[i], 0, c.seqs.offsets[i], c.seqs.lengths[i], 0};
        w.raw(ser.data(), n * sizeof(SerializedIndexEntry));
        w.end(K_DBR1INDEX, offIndex);
        lenIndex = 28 + n * sizeof(SerializedIndexEntry);
        offData = w.begin();
        w.raw(c.seqs.data.data(), c.seqs.data.size());
        w.end(K_DBR1DATA, offData);
        lenData = c.seqs.data.size();
        w.recs.push_back(Writer::Rec{K_DBR2INDEX, offIndex, lenIndex + 1});       // same database on both sides (:133-135)
        w.recs.push_back(Writer::Rec{K_DBR2DATA, offData, lenData + 1});
    }
    {   // ScoreMatrix::serialize (ScoreMatrix.h:28-39): all scores, then all indices (reference numbering)
        ScoreMat3 sm;
        build_scoremat3(km, sm);
        uint16_t addrOf[8000];
        kmer3_address_table(addrOf);
        std::vector<uint16_t> canon(8000);
        for (int k = 0; k < 8000; k++) canon[addrOf[k]] = static_cast<uint16_t>(k);
        const uint64_t s = w.begin();
        std::vector<int16_t> srow(ROW3, -255);
        for (int r = 0; r < 8000; r++) { std::memcpy(srow.data(), sm.score.data() + static_cast<size_t>(r) * 8000, 8000 * 2); w.raw(srow.data(), ROW3 * 2); }
        std::vector<uint32_t> irow(ROW3, 0);
        for (int r = 0; r < 8000; r++) {
            const uint16_t *src = sm.index.data() + static_cast<size_t>(r) * 8000;
            for (int z = 0; z < 8000; z++) irow[z] = canon[src[z]];
            w.raw(irow.data(), ROW3 * 4);
        }
        w.end(K_SCOREMATRIX3MER, s);
        std::vector<int16_t> s2; std::vector<uint32_t> i2;
        scorematrix2(km, s2, i2);
        const uint64_t t = w.begin();
        w.raw(s2.data(), s2.size() * 2); w.raw(i2.data(), i2.size() * 4);
        w.end(K_SCOREMATRIX2MER, t);
    }
    {   // IndexEntryLocal{u32 seqId; u16 position_j} packed (IndexTable.h:25-27), offsets size_t[cells + 1]
        const uint64_t s = w.begin();
        uint64_t written = 0;
        if (!src.entries6([&](const void *p, size_t bytes) { w.raw(p, bytes); written += bytes; return w.ok; })) { fclose(w.f); return "write to " + base + " failed"; }
        if (written != src.nEntries * 6) { fclose(w.f); return "index content is inconsistent (entries)"; }
        w.end(K_ENTRIES, s);
        const uint64_t *off = src.offsets();
        if (!off || off[TABLE] != src.nEntries) { fclose(w.f); return "index content is inconsistent (offsets)"; }
        w.put(K_ENTRIESOFFSETS, off, (TABLE + 1) * 8);
        const uint64_t num = src.nEntries;
        w.put(K_ENTRIESNUM, &num, 8);
    }
    {
        const uint64_t count = n;
        w.put(K_SEQCOUNT, &count, 8);
        const int64_t dataSize = static_cast<int64_t>(src.maskedSize);
        w.put(K_SEQINDEXDATASIZE, &dataSize, 8);
        w.put(K_SEQINDEXSEQOFFSET, c.seqOffsets.data(), (n + 1) * 8);
        const uint64_t s = w.begin();
        w.raw(src.masked, src.maskedSize);
        const char z = 0;
        w.raw(&z, 1);                                                  // getDataSize() + 1 bytes (:303)
        w.end(K_SEQINDEXDATA, s);
    }
    if (fclose(w.f) != 0 || !w.ok) return "write to " + base + " failed";
    std::stable_sort(w.recs.begin(), w.recs.end(), [](const Writer::Rec &a, const Writer::Rec &b) { return a.key < b.key; });
    FILE *i = fopen((base + ".index").c_str(), "wb");
    if (!i) return "cannot create " + base + ".index";
    // lengths modulo 2^32, as the reference's own index files hold them (DBReader::Index::length is an unsigned int and DBWriter::sortIndex
    // writes the final .index through it): the 10 GB of k = 7 list offsets read 1650065417
    for (const Writer::Rec &r : w.recs) fprintf(i, "%u\t%llu\t%llu\n", r.key, (unsigned long long) r.offset, (unsigned long long) (r.length & 0xFFFFFFFFull));
    if (fclose(i) != 0) return "cannot close " + base + ".index";
    FILE *t = fopen((base + ".dbtype").c_str(), "wb");
    if (!t) return "cannot create " + base + ".dbtype";
    const int32_t v = DBTYPE_INDEX_DB;
    fwrite(&v, 4, 1, t);
    if (fclose(t) != 0) return "cannot close " + base + ".dbtype";
    return "";
}

std::string read_index_file(const std::string &base, IndexFileContent &c, bool viewLists) {
    struct stat st;
    std::string dataPath = base;
    if (stat(dataPath.c_str(), &st) != 0) {
        dataPath = base + ".0";                                        // an unmerged single-split DB (FileUtil::findDatafiles)
        if (stat(dataPath.c_str(), &st) != 0) return "index database " + base + " has no data file";
        if (stat((base + ".1").c_str(), &st) == 0) return "index databases in several data files (--split > 1) are not implemented";
    }
    Mapped idx;
    std::shared_ptr<Mapped> datp = std::make_shared<Mapped>();
    Mapped &dat = *datp;
    std::string e = idx.open(base + ".index");
    if (!e.empty()) return e;
    e = dat.open(dataPath);
    if (!e.empty()) return e;
    struct Ent { uint64_t offset, length; };
    std::map<uint32_t, Ent> ent;
    for (const char *p = idx.p, *end = idx.p + idx.n; p < end;) {
        char *q;
        const uint32_t key = static_cast<uint32_t>(strtoul(p, &q, 10));
        if (q == p) break;
        Ent x;
        x.offset = strtoull(q, &q, 10);
        x.length = strtoull(q, &q, 10);
        if (x.offset >= dat.n || x.length == 0) return "index of " + base + " points outside the data file";
        ent[key] = x;
        p = q;
        while (p < end && *p != '\n') p++;
        if (p < end) p++;
    }
    {   // the reference keeps an entry's length in 32 bits (DBReader::Index::length; DBWriter::sortIndex re-writes the .index through it), so an
        // entry of 4 GB or more -- the 10 GB of k = 7 list offsets, the entries of a large database -- is recorded modulo 2^32: its true length
        // is the value of that residue class that fills the room up to the next entry (entries are padded to pages only)
        std::vector<uint64_t> starts;
        for (const auto &kv : ent) starts.push_back(kv.second.offset);
        starts.push_back(dat.n);
        std::sort(starts.begin(), starts.end());
        for (auto &kv : ent) {
            const uint64_t room = *std::upper_bound(starts.begin(), starts.end(), kv.second.offset) - kv.second.offset;
            if (kv.second.length <= 0xFFFFFFFFull && room > kv.second.length) kv.second.length += ((room - kv.second.length) >> 32) << 32;
            if (kv.second.offset + kv.second.length > dat.n) return "index of " + base + " points outside the data file";
        }
    }
    auto need = [&](uint32_t key, const char *&p, uint64_t &len) -> bool {
        auto it = ent.find(key);
        if (it == ent.end()) return false;
        p = dat.p + it->second.offset; len = it->second.length - 1;     // without the entry's terminating NUL
        return true;
    };
    const char *p; uint64_t len;
    if (!need(K_VERSION, p, len) || strncmp(p, INDEX_VERSION, strlen(INDEX_VERSION)) != 0)
        return "Outdated index version. Please recompute it with 'createindex'!";                           // Prefiltering.cpp:157
    if (!need(K_META, p, len) || len < 11 * sizeof(int)) return base + ": no META entry";
    int meta[12] = {0};
    std::memcpy(meta, p, std::min<uint64_t>(len, sizeof(meta)));
    c.meta.maxSeqLen = meta[0]; c.meta.kmerSize = meta[1]; c.meta.compBiasCorr = meta[2]; c.meta.alphabetSize = meta[3]; c.meta.mask = meta[4];
    c.meta.spacedKmer = meta[5]; c.meta.kmerThr = meta[6]; c.meta.seqType = meta[7]; c.meta.srcSeqType = meta[8]; c.meta.headers1 = meta[9];
    c.meta.headers2 = meta[10]; c.meta.splits = meta[11] == 0 ? 1 : meta[11];
    if (c.meta.kmerSize != 6 && c.meta.kmerSize != 7) return "the index was built with -k " + std::to_string(c.meta.kmerSize) + ": k = 6 and 7 are implemented";
    const size_t TABLE = kmer_table_cells(c.meta.kmerSize);
    if (c.meta.alphabetSize != ALPH) return "the index was built with --alph-size " + std::to_string(c.meta.alphabetSize) + ": only 21 is implemented";
    if ((c.meta.seqType & 0xFFFF) != 0) return "only amino-acid target indices are implemented (profile targets: SURVEY 8f-4)";
    if (c.meta.splits != 1) return "index databases with --split > 1 are not implemented";
    if (c.meta.spacedKmer != 1) return "the index was built without spaced k-mers: not implemented";
    if (need(K_SPACEDPATTERN, p, len) && len > 0 && std::string(p, len) != (c.meta.kmerSize == 7 ? "11010110011" : "1101010011")) return "the index uses a custom spaced k-mer pattern: not implemented";
    if (!need(K_SCOREMATRIXNAME, p, len)) return base + ": no SCOREMATRIXNAME entry";
    {
        const std::string s(p, strnlen(p, len));
        const size_t colon = s.find(".out:");
        c.matrixName = colon == std::string::npos ? s : s.substr(0, colon + 4);
        if (c.matrixName != "VTML80.out") return "the index was built with seed matrix " + c.matrixName + ": only VTML80.out is implemented";
    }
    if (!need(K_DBR1INDEX, p, len) || len < 28) return base + ": no DBR1INDEX entry";
    {
        uint64_t size; int dbtype;
        std::memcpy(&size, p, 8); std::memcpy(&dbtype, p + 20, 4);
        if (28 + size * sizeof(SerializedIndexEntry) > len) return base + ": truncated DBR1INDEX entry";
        c.seqs.dbtype = dbtype;
        c.seqs.keys.resize(size); c.seqs.offsets.resize(size); c.seqs.lengths.resize(size);
        for (uint64_t i = 0; i < size; i++) {
            SerializedIndexEntry s;
            std::memcpy(&s, p + 28 + i * sizeof(SerializedIndexEntry), sizeof(s));
            c.seqs.keys[i] = s.id; c.seqs.offsets[i] = s.offset; c.seqs.lengths[i] = s.length;
        }
    }
    if (!need(K_DBR1DATA, p, len)) return base + ": the index holds no sequence data";
    c.seqs.data.assign(p, p + len);
    if (viewLists) madvise(dat.p, dat.n, MADV_SEQUENTIAL);
    for (size_t i = 0; i < c.seqs.keys.size(); i++)
        if (c.seqs.offsets[i] + c.seqs.lengths[i] > c.seqs.data.size()) return base + ": sequence index points outside the sequence data";
    const size_t n = c.seqs.keys.size();
    uint64_t nEntries = 0, seqCount = 0;
    int64_t maskedSize = 0;
    if (!need(K_ENTRIESNUM, p, len) || len < 8) return base + ": no ENTRIESNUM entry (index without k-mer lists)";
    std::memcpy(&nEntries, p, 8);
    if (!need(K_SEQCOUNT, p, len) || len < 8) return base + ": no SEQCOUNT entry";
    std::memcpy(&seqCount, p, 8);
    if (seqCount != n) return base + ": SEQCOUNT does not match the sequence index";
    if (!need(K_ENTRIESOFFSETS, p, len) || len < (TABLE + 1) * 8) return base + ": no ENTRIESOFFSETS entry";
    const uint64_t *offView = reinterpret_cast<const uint64_t *>(p);             // (entries are page-aligned in the file)
    if (offView[TABLE] != nEntries) return base + ": ENTRIESOFFSETS does not end at ENTRIESNUM";
    if (!need(K_ENTRIES, p, len) || len < nEntries * 6) return base + ": no ENTRIES entry";
    const unsigned char *entView = reinterpret_cast<const unsigned char *>(p);
    if (viewLists) {
        c.listOffsets = offView; c.listEntries6 = entView; c.nEntries = nEntries; c.mapping = datp;
    } else {
        c.index.offsets.assign(offView, offView + TABLE + 1);
        c.index.entries.resize(nEntries);
#pragma omp parallel for schedule(static)
        for (uint64_t k = 0; k < nEntries; k++) {
            uint32_t seq; uint16_t pos;
            std::memcpy(&seq, entView + k * 6, 4); std::memcpy(&pos, entView + k * 6 + 4, 2);
            c.index.entries[k] = static_cast<uint64_t>(seq) | (static_cast<uint64_t>(pos) << 32);
        }
    }
    if (!need(K_SEQINDEXDATASIZE, p, len) || len < 8) return base + ": no SEQINDEXDATASIZE entry";
    std::memcpy(&maskedSize, p, 8);
    if (!need(K_SEQINDEXSEQOFFSET, p, len) || len < (n + 1) * 8) return base + ": no SEQINDEXSEQOFFSET entry";
    c.seqOffsets.resize(n + 1);
    std::memcpy(c.seqOffsets.data(), p, (n + 1) * 8);
    if (c.seqOffsets[n] != static_cast<uint64_t>(maskedSize)) return base + ": SEQINDEXSEQOFFSET does not end at SEQINDEXDATASIZE";
    if (!need(K_SEQINDEXDATA, p, len) || len < static_cast<uint64_t>(maskedSize)) return base + ": no SEQINDEXDATA entry";
    if (viewLists) c.maskedView = reinterpret_cast<const uint8_t *>(p);
    else c.index.masked.assign(p, p + maskedSize);
    c.index.maskedResidues = 0;
    return "";
}

// the lists of a content read with viewLists = true -> host vectors (c.index), the mapping is released
void materialize_lists(IndexFileContent &c) {
    if (!c.listOffsets) return;
    const size_t TABLE = kmer_table_cells(c.meta.kmerSize);
    c.index.offsets.assign(c.listOffsets, c.listOffsets + TABLE + 1);
    c.index.entries.resize(c.nEntries);
    const unsigned char *entView = c.listEntries6;
#pragma omp parallel for schedule(static)
    for (uint64_t k = 0; k < c.nEntries; k++) {
        uint32_t seq; uint16_t pos;
        std::memcpy(&seq, entView + k * 6, 4); std::memcpy(&pos, entView + k * 6 + 4, 2);
        c.index.entries[k] = static_cast<uint64_t>(seq) | (static_cast<uint64_t>(pos) << 32);
    }
    c.index.masked.assign(c.maskedView, c.maskedView + c.seqOffsets.back());
    c.listOffsets = nullptr; c.listEntries6 = nullptr; c.maskedView = nullptr; c.mapping.reset();
}

}  // namespace mk
