// metaeuk_amd/csrc/mk_dbio.hpp -- MMseqs2 on-disk database format, reader and writer (host side).
// Format (M/src/commons/DBReader.cpp:101-215,600-640; DBWriter.cpp:124-213,401-428,531-623):
//   name[.0 .. .N]   data: entries back to back, each terminated by '\0'; split files concatenate logically
//   name.index       text lines "key \t offset \t length \n" (length includes the '\0'), sorted by key
//   name.dbtype      4-byte little-endian int: low 16 bits type (0 amino acids, 5 alignment result,
//                    7 prefilter result), bit 31 = compressed
// Sequence entries are "RESIDUES\n\0" (sequence length = length - 2, DBReader.h:211-231).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/stat.h>

namespace mk {

enum { DBTYPE_AMINO_ACIDS = 0, DBTYPE_ALIGNMENT_RES = 5, DBTYPE_PREFILTER_RES = 7 };

struct DbEntry { uint32_t key; uint64_t offset; uint64_t length; };

struct Database {
    std::vector<char> data;          // concatenation of the data file(s)
    std::vector<DbEntry> entries;    // in LINEAR_ACCCESS order: by data offset (DBReader.cpp:362-391)
    int dbtype = -1;

    static bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

    static bool slurp(const std::string &path, std::vector<char> &out) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return false;
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        const size_t old = out.size();
        out.resize(old + (size_t) n);
        const size_t got = n > 0 ? fread(out.data() + old, 1, (size_t) n, f) : 0;
        fclose(f);
        return got == (size_t) n;
    }

    // returns "" on success, else an error message
    std::string open(const std::string &name) {
        data.clear(); entries.clear();
        if (exists(name)) {
            if (!slurp(name, data)) return "cannot read " + name;
        } else {                                            // FileUtil::findDatafiles: name.0, name.1, ...
            int part = 0;
            while (exists(name + "." + std::to_string(part))) {
                if (!slurp(name + "." + std::to_string(part), data)) return "cannot read " + name + "." + std::to_string(part);
                part++;
            }
            if (part == 0) return "database " + name + " has no data file";
        }
        std::vector<char> idx;
        if (!slurp(name + ".index", idx)) return "cannot read " + name + ".index";
        const char *p = idx.data(), *end = idx.data() + idx.size();
        while (p < end) {
            char *q;
            DbEntry e;
            e.key = (uint32_t) strtoul(p, &q, 10);
            if (q == p) break;
            e.offset = strtoull(q, &q, 10);
            e.length = strtoull(q, &q, 10);
            if (e.offset + e.length > data.size()) return "index of " + name + " points outside the data file";
            entries.push_back(e);
            p = q;
            while (p < end && *p != '\n') p++;
            if (p < end) p++;
        }
        std::stable_sort(entries.begin(), entries.end(), [](const DbEntry &a, const DbEntry &b) { return a.offset < b.offset; });
        dbtype = -1;
        FILE *f = fopen((name + ".dbtype").c_str(), "rb");
        if (f) { int32_t t = 0; if (fread(&t, 4, 1, f) == 1) dbtype = t; fclose(f); }
        if (dbtype == -1) return "database " + name + " has no .dbtype file";
        if (dbtype & (1 << 31)) return "compressed databases are not supported (" + name + ")";
        return "";
    }

    // entry positions by ascending key (ties: data offset) -- DBReader's SORT_BY_ID_OFFSET
    std::vector<size_t> keyOrder() const {
        std::vector<size_t> ord(entries.size());
        for (size_t i = 0; i < ord.size(); i++) ord[i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return entries[x].key < entries[y].key; });
        return ord;
    }
    const char *entry(size_t i) const { return data.data() + entries[i].offset; }
    size_t seqLen(size_t i) const { return entries[i].length >= 2 ? (size_t) entries[i].length - 2 : 0; }
};

// Result DB writer: one data file; the index is sorted by key on close like DBWriter::mergeResults.
struct DatabaseWriter {
    std::string name;
    FILE *f = nullptr;
    uint64_t offset = 0;
    std::vector<DbEntry> index;
    int dbtype;

    DatabaseWriter(const std::string &n, int type) : name(n), dbtype(type) {}
    std::string open() {
        // never leave a stale "done" marker behind (blastp.sh:59,77 test for name.dbtype)
        remove((name + ".dbtype").c_str());
        f = fopen(name.c_str(), "wb");
        return f ? "" : "cannot create " + name;
    }
    void write(uint32_t key, const char *buf, size_t n) {
        if (n) fwrite(buf, 1, n, f);
        fputc('\0', f);
        index.push_back(DbEntry{key, offset, n + 1});
        offset += n + 1;
    }
    std::string close() {
        if (fclose(f) != 0) return "cannot close " + name;
        f = nullptr;
        std::stable_sort(index.begin(), index.end(), [](const DbEntry &a, const DbEntry &b) { return a.key < b.key; });
        FILE *i = fopen((name + ".index").c_str(), "wb");
        if (!i) return "cannot create " + name + ".index";
        for (const DbEntry &e : index) fprintf(i, "%u\t%llu\t%llu\n", e.key, (unsigned long long) e.offset, (unsigned long long) e.length);
        if (fclose(i) != 0) return "cannot close " + name + ".index";
        FILE *t = fopen((name + ".dbtype").c_str(), "wb");
        if (!t) return "cannot create " + name + ".dbtype";
        const int32_t v = dbtype;
        fwrite(&v, 4, 1, t);
        if (fclose(t) != 0) return "cannot close " + name + ".dbtype";
        return "";
    }
};

// DBReader::decomposeDomainByAminoAcid (DBReader.cpp:1216-1257): the contiguous entry range of worker `rank` of `world` when the
// entries (lengths as in the index, in processing order) are dealt out in chunks of ceil(total / world) bytes
inline void decomposeByLength(const std::vector<DbEntry> &entries, int rank, int world, size_t &start, size_t &count) {
    const size_t n = entries.size();
    start = 0; count = n;
    if (world <= 1) return;
    if (n <= (size_t) world) { start = (size_t) rank < n ? (size_t) rank : 0; count = (size_t) rank < n ? 1 : 0; return; }
    uint64_t total = 0;
    for (const DbEntry &e : entries) total += e.length;
    const uint64_t chunk = (total + (uint64_t) world - 1) / (uint64_t) world;
    std::vector<size_t> per((size_t) world, 0);
    size_t w = 0;
    uint64_t acc = 0;
    for (const DbEntry &e : entries) {
        if (acc >= chunk) { acc = 0; w++; }
        acc += e.length;
        per[w]++;
    }
    start = 0;
    for (int r = 0; r < rank; r++) start += per[(size_t) r];
    count = per[(size_t) rank];
}

// DBWriter::mergeResults (DBWriter.cpp:531-623) over the result DBs <out>_0 .. <out>_{n-1} the workers wrote: data files back to
// back, index offsets rebased, index sorted by key; the shards are removed.  "" on success.
inline std::string mergeShards(const std::string &out, int n, int dbtype) {
    remove((out + ".dbtype").c_str());
    FILE *d = fopen(out.c_str(), "wb");
    if (!d) return "cannot create " + out;
    std::vector<DbEntry> index;
    uint64_t base = 0;
    for (int r = 0; r < n; r++) {
        const std::string shard = out + "_" + std::to_string(r);
        std::vector<char> data, idx;
        if (!Database::slurp(shard, data) || !Database::slurp(shard + ".index", idx)) { fclose(d); return "cannot read shard " + shard; }
        if (!data.empty() && fwrite(data.data(), 1, data.size(), d) != data.size()) { fclose(d); return "cannot write " + out; }
        const char *p = idx.data(), *end = idx.data() + idx.size();
        while (p < end) {
            char *q;
            DbEntry e;
            e.key = (uint32_t) strtoul(p, &q, 10);
            if (q == p) break;
            e.offset = strtoull(q, &q, 10) + base;
            e.length = strtoull(q, &q, 10);
            index.push_back(e);
            p = q;
            while (p < end && *p != '\n') p++;
            if (p < end) p++;
        }
        base += data.size();
    }
    if (fclose(d) != 0) return "cannot close " + out;
    std::stable_sort(index.begin(), index.end(), [](const DbEntry &a, const DbEntry &b) { return a.key < b.key; });
    FILE *i = fopen((out + ".index").c_str(), "wb");
    if (!i) return "cannot create " + out + ".index";
    for (const DbEntry &e : index) fprintf(i, "%u\t%llu\t%llu\n", e.key, (unsigned long long) e.offset, (unsigned long long) e.length);
    if (fclose(i) != 0) return "cannot close " + out + ".index";
    for (int r = 0; r < n; r++) {
        const std::string shard = out + "_" + std::to_string(r);
        remove(shard.c_str()); remove((shard + ".index").c_str()); remove((shard + ".dbtype").c_str());
    }
    FILE *t = fopen((out + ".dbtype").c_str(), "wb");
    if (!t) return "cannot create " + out + ".dbtype";
    const int32_t v = dbtype;
    fwrite(&v, 4, 1, t);
    if (fclose(t) != 0) return "cannot close " + out + ".dbtype";
    return "";
}

}  // namespace mk
