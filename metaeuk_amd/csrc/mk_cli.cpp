// metaeuk_amd/csrc/mk_cli.cpp -- `metaeuk-amd prefilter|align|search|extractorfs|predictexons|createindex`: the two hot modules of
// `metaeuk predictexons` with the reference's process-level signature, flag names and on-disk DB format,
// on top of the C ABI (include/metaeuk_amd.h).
//
//   prefilter <i:queryDB> <i:targetDB> <o:prefilterDB> [flags]          M/src/MMseqsBase.cpp:591-598
//   align     <i:queryDB> <i:targetDB> <i:prefilterDB> <o:alignmentDB>  M/src/MMseqsBase.cpp:641-649
//
// blastp.sh:70,85 invokes exactly these two with the flag strings built by
// Parameters::createParameterString (Parameters.cpp:2811-2881), i.e. every flag of the module's list is
// passed explicitly.  All of them are accepted here; the ones the predictexons path actually varies are
// honoured, and a value that would change results in a way this build does not restate is a hard error
// in EVERY command (the reference parser also fails hard on anything it does not know).  Omitted flags take
// the module defaults of the reference (-e 0.001 for align / search, -s 4 for prefilter, 5.7 for search) with
// one exception: `align` and `search` must be given --alignment-mode 2, because the module default (0 = auto)
// resolves to score-only output, which is not built.  predictexons applies setPredictExonsDefaults.
// Sharded commands (RANK / WORLD_SIZE or --shard r/N): every worker removes its own leftovers and stamps its
// shard with the launch token before it computes, a failing worker leaves <out>_<r>.failed, worker 0 merges
// only shards of its own launch and gives up after MK_SHARD_TIMEOUT_S (7200) seconds.  On failure nothing named
// <out>.dbtype is left behind, because the workflows use that file as their "step done" marker
// (blastp.sh:59,77).  Exit status non-zero on any error, as `EXIT(EXIT_FAILURE)` does (Util.h:14).
#include "../../include/metaeuk_amd.h"
#include "mk_dbio.hpp"

#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <sys/stat.h>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <deque>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

namespace {

// the command's two doors to the environment: what a launcher tells a worker (RANK, WORLD_SIZE, LOCAL_RANK, TORCHELASTIC_RUN_ID,
// MK_LAUNCH_ID, MK_SHARD_TIMEOUT_S, the reference's own MMSEQS_IGNORE_INDEX) is always honoured; test knobs (MK_CLI_*) only under MK_DEBUG=1
const char *launcherEnv(const char *name) { return getenv(name); }
const char *knobEnv(const char *name) {
    const char *dbg = launcherEnv("MK_DEBUG");
    if (!dbg || atoi(dbg) == 0) return nullptr;
    const char *v = launcherEnv(name);
    return (v && *v) ? v : nullptr;
}

struct Flag { const char *name; const char *def; bool honoured; };
// union of the `prefilter` and `align` parameter lists (Parameters.cpp:387-455) + common ones
const Flag FLAGS[] = {
    {"--sub-mat", "aa:blosum62.out,nucl:nucleotide.out", false}, {"--seed-sub-mat", "aa:VTML80.out,nucl:nucleotide.out", false},
    {"-s", "4", true}, {"-k", "0", false}, {"--k-score", "seq:2147483647,prof:2147483647", true},
    {"--target-search-mode", "0", false}, {"--alph-size", "aa:21,nucl:5", false}, {"--max-seq-len", "65535", false},
    {"--max-seqs", "300", true}, {"--split", "0", false}, {"--split-mode", "2", false}, {"--split-memory-limit", "0", false},
    {"-c", "0", false}, {"--cov-mode", "0", false}, {"--comp-bias-corr", "1", true}, {"--comp-bias-corr-scale", "1", true},
    {"--diag-score", "1", false}, {"--exact-kmer-matching", "0", false}, {"--mask", "1", true}, {"--mask-prob", "0.9", true},
    {"--mask-lower-case", "0", false}, {"--mask-n-repeat", "0", false}, {"--min-ungapped-score", "15", true},
    {"--add-self-matches", "0", false}, {"--spaced-kmer-mode", "1", false}, {"--spaced-kmer-pattern", "", false},
    {"--local-tmp", "", false}, {"--db-load-mode", "0", false}, {"--pca", "", false}, {"--pcb", "", false},
    {"--taxon-list", "", false}, {"--threads", "", true}, {"--compressed", "0", false}, {"-v", "3", true},
    {"-a", "0", false}, {"--alignment-mode", "2", false}, {"--alignment-output-mode", "0", true}, {"--wrapped-scoring", "0", false},
    {"-e", "100", true}, {"--min-seq-id", "0", false}, {"--min-aln-len", "0", true}, {"--seq-id-mode", "0", false},
    {"--alt-ali", "0", false}, {"--max-rejected", "2147483647", false}, {"--max-accept", "2147483647", false},
    {"--score-bias", "0", false}, {"--realign", "0", false}, {"--realign-score-bias", "-0.2", false}, {"--realign-max-seqs", "2147483647", false},
    {"--corr-score-weight", "0", false}, {"--gap-open", "aa:11,nucl:5", true}, {"--gap-extend", "aa:1,nucl:2", true}, {"--zdrop", "40", false},
    // predictexons only: extractorfs (Parameters.cpp:2525-2554), search workflow extras, collectoptimalset (LocalParameters.h:92-104)
    {"--min-length", "15", true}, {"--max-length", "32734", false}, {"--max-gaps", "2147483647", false}, {"--contig-start-mode", "2", false},
    {"--contig-end-mode", "2", false}, {"--orf-start-mode", "1", false}, {"--forward-frames", "1,2,3", false}, {"--reverse-frames", "1,2,3", false},
    {"--translation-table", "1", false}, {"--translate", "0", false}, {"--use-all-table-starts", "0", false}, {"--id-offset", "0", false},
    {"--create-lookup", "0", false}, {"--add-orf-stop", "0", false}, {"--num-iterations", "1", false}, {"--start-sens", "4", false},
    {"--sens-steps", "1", false}, {"--exhaustive-search", "0", false}, {"--exhaustive-search-filter", "0", false}, {"--strand", "1", false},
    {"--remove-tmp-files", "0", false}, {"--reuse-latest", "0", false}, {"--force-reuse", "0", false}, {"--disk-space-limit", "0", false},
    {"--mpi-runner", "", false}, {"--reverse-fragments", "0", false},
    {"--metaeuk-eval", "0.001", true}, {"--metaeuk-tcov", "0.5", true}, {"--max-intron", "10000", true}, {"--min-intron", "15", true},
    {"--min-exon-aa", "11", true}, {"--max-overlap", "10", true}, {"--max-exon-sets", "1", true}, {"--set-gap-open", "-1", true},
    {"--set-gap-extend", "-1", true},
    // properties of the reference build/host being reproduced (see mk_params in the ABI header)
    {"--ref-simd", "avx2", true}, {"--ref-l2-bytes", "", true}, {"--gpu", "0", true}, {"--shard", "", true},
};

std::string g_failMarker;        // <out>_<rank>.failed of a sharded worker: written by die(), seen by worker 0 (which stops waiting)
int die(const char *fmt, const std::string &a = "") {
    fprintf(stderr, fmt, a.c_str());
    fprintf(stderr, "\n");
    if (!g_failMarker.empty()) { FILE *f = fopen(g_failMarker.c_str(), "w"); if (f) { fprintf(f, fmt, a.c_str()); fclose(f); } }
    return EXIT_FAILURE;
}

bool sameValue(const std::string &v, const char *def) {
    if (v == def) return true;
    // numeric equality ("0.000" vs "0") and the aa: part of MultiParams ("aa:21,nucl:5")
    char *e1, *e2;
    const double a = strtod(v.c_str(), &e1), b = strtod(def, &e2);
    if (*e1 == 0 && *e2 == 0 && e1 != v.c_str()) return a == b;
    return false;
}

std::string aaPart(const std::string &v) {          // "aa:11,nucl:5" -> "11"; "11" -> "11"
    const size_t p = v.find("aa:");
    if (p == std::string::npos) return v;
    const size_t c = v.find(',', p);
    return v.substr(p + 3, c == std::string::npos ? std::string::npos : c - p - 3);
}

struct Args {
    std::vector<std::string> pos;
    std::map<std::string, std::string> opt;
};

// the run statistics the reference's prefilter logs (Prefiltering::printStatistics, Prefiltering.cpp:953-975), same six lines, on stderr
void printPrefilterStatistics(const mk_queries *Q, const mk_params &P) {
    mk_prefilter_stats st;
    char buf[512];
    if (mk_prefilter_statistics(Q, &st) != MK_OK) return;
    const size_t n = mk_format_prefilter_statistics(buf, sizeof(buf), &st, (uint64_t) P.max_seqs);
    fwrite(buf, 1, n, stderr);
}

int parse(int argc, char **argv, Args &a) {
    for (int i = 2; i < argc; i++) {
        const std::string s = argv[i];
        if (s.size() >= 2 && s[0] == '-' && !(s[1] >= '0' && s[1] <= '9')) {
            const Flag *f = nullptr;
            for (const Flag &k : FLAGS) if (s == k.name) f = &k;
            if (!f) return die("Unrecognized parameter \"%s\"", s);
            if (i + 1 >= argc) return die("Missing argument %s", s);
            a.opt[s] = argv[++i];
        } else {
            a.pos.push_back(s);
        }
    }
    for (const Flag &k : FLAGS) {
        auto it = a.opt.find(k.name);
        if (it == a.opt.end() || k.honoured) continue;
        std::string v = it->second, d = k.def;
        if (std::string(k.name) == "--alph-size" || std::string(k.name) == "--sub-mat" || std::string(k.name) == "--seed-sub-mat") { v = aaPart(v); d = aaPart(d); }
        if (std::string(k.name) == "--split-memory-limit" || std::string(k.name) == "--local-tmp" ||
            std::string(k.name) == "--db-load-mode" || std::string(k.name) == "--split" || std::string(k.name) == "--split-mode" ||
            std::string(k.name) == "--pca" || std::string(k.name) == "--pcb" || std::string(k.name) == "--zdrop" || std::string(k.name) == "--realign-score-bias" ||
            std::string(k.name) == "--realign-max-seqs" || std::string(k.name) == "--seq-id-mode" || std::string(k.name) == "--mask-lower-case" ||
            std::string(k.name) == "--threads" || std::string(k.name) == "--remove-tmp-files" || std::string(k.name) == "--reuse-latest" ||
            std::string(k.name) == "--force-reuse" || std::string(k.name) == "--disk-space-limit" || std::string(k.name) == "--mpi-runner") continue;    // no effect on this path
        if (std::string(k.name) == "--start-sens") continue;          // only read when --sens-steps > 1, which is refused below
        if (std::string(k.name) == "--exhaustive-search") continue;   // predictexons turns it on by itself for profile targets (PredictExons.cpp:22-26); checked there
        if (std::string(k.name) == "--max-seq-len") {                 // sequences are never cut here: the value must admit everything the reference admits
            if (atol(v.c_str()) < 65535) { fprintf(stderr, "--max-seq-len %s: shorter limits than the default 65535 (sequence splitting) are not implemented\n", v.c_str()); return EXIT_FAILURE; }
            continue;
        }
        if (std::string(k.name) == "-k") {                    // 0 = automatic: 6 below 3.35e9 target residues, else 7 (IndexTable.h:439-449)
            if (v != "0" && v != "6" && v != "7") { fprintf(stderr, "-k %s: k-mer sizes 6 and 7 (or 0 = auto) are implemented\n", v.c_str()); return EXIT_FAILURE; }
            continue;
        }
        if (!sameValue(v, d.c_str())) {
            fprintf(stderr, "%s %s: only the default (%s) is implemented by the MI355X path\n", k.name, it->second.c_str(), k.def);
            return EXIT_FAILURE;
        }
    }
    return 0;
}

int fillParams(const Args &a, mk_params &P, int &gpu) {
    mk_default_params(&P);
    P.sensitivity = 4.0f;                                    // Parameters.cpp:2360 (search overrides with -s explicitly)
    P.min_aln_len = 0;                                       // module default; predictexons passes --min-aln-len 11
    auto get = [&](const char *k) -> const std::string * { auto it = a.opt.find(k); return it == a.opt.end() ? nullptr : &it->second; };
    if (auto v = get("-s")) P.sensitivity = (float) atof(v->c_str());
    if (auto v = get("--k-score")) {
        std::string s = *v;
        const size_t p = s.find("seq:");
        if (p != std::string::npos) s = s.substr(p + 4);
        P.kmer_score = atoi(s.c_str());
    }
    if (auto v = get("-k")) P.kmer_size = atoi(v->c_str());
    if (auto v = get("--max-seqs")) P.max_seqs = atoi(v->c_str());
    if (auto v = get("--min-ungapped-score")) P.min_ungapped_score = atoi(v->c_str());
    if (auto v = get("--comp-bias-corr")) P.comp_bias_corr = atoi(v->c_str());
    if (auto v = get("--comp-bias-corr-scale")) P.comp_bias_scale = (float) atof(v->c_str());
    if (auto v = get("--mask")) P.mask = atoi(v->c_str());
    if (auto v = get("--mask-prob")) P.mask_prob = (float) atof(v->c_str());
    if (auto v = get("-e")) P.evalue_thr = atof(v->c_str());
    if (auto v = get("--min-aln-len")) P.min_aln_len = atoi(v->c_str());
    if (auto v = get("--gap-open")) P.gap_open = atoi(aaPart(*v).c_str());
    if (auto v = get("--gap-extend")) P.gap_extend = atoi(aaPart(*v).c_str());
    if (auto v = get("--threads")) setenv("OMP_NUM_THREADS", v->c_str(), 0);
    if (auto v = get("--ref-simd")) {
        if (*v == "sse41") { P.simd_lanes_byte = 16; P.simd_lanes_word = 8; P.simd_lanes_double = 2; }
        else if (*v != "avx2") return die("--ref-simd must be avx2 or sse41 (got %s)", *v);
    }
    long l2 = sysconf(_SC_LEVEL2_CACHE_SIZE);                // Util::getL2CacheSize (Util.cpp:317-332) of THIS host
    P.host_l2_bytes = l2 > 0 ? (uint64_t) l2 : 262144;
    if (auto v = get("--ref-l2-bytes")) if (!v->empty()) P.host_l2_bytes = strtoull(v->c_str(), nullptr, 10);
    gpu = 0;
    if (auto v = get("--gpu")) gpu = atoi(v->c_str());
    if (const char *lr = launcherEnv("LOCAL_RANK")) if (!get("--gpu")) gpu = atoi(lr);
    if (P.gap_open != 11 || P.gap_extend != 1) return die("only --gap-open 11 --gap-extend 1 has a hard-coded Gumbel parameter set in the reference (EvalueComputation.h:64-69); other values are not implemented%s");
    return 0;
}

// sequence DB -> encoded residues + offsets, in LINEAR_ACCCESS (data offset) order
void encodeDb(const mk::Database &db, std::vector<uint8_t> &res, std::vector<uint64_t> &off) {
    off.assign(db.entries.size() + 1, 0);
    for (size_t i = 0; i < db.entries.size(); i++) off[i + 1] = off[i] + db.seqLen(i);
    res.assign(off.back() + 1, 0);
    for (size_t i = 0; i < db.entries.size(); i++) mk_encode(db.entry(i), db.seqLen(i), res.data() + off[i]);
}

// profile DB (type 2) -> the 25-byte columns of its entries without the trailing NULs + column offsets (DBReader::getSeqLen for
// profiles: (length - 1) / 25, DBReader.h:224-227), in the order of db.entries
constexpr int DBTYPE_HMM_PROFILE = 2;
void profileColumns(const mk::Database &db, std::vector<uint8_t> &cols, std::vector<uint64_t> &off) {
    off.assign(db.entries.size() + 1, 0);
    for (size_t i = 0; i < db.entries.size(); i++) off[i + 1] = off[i] + (std::max<uint64_t>(db.entries[i].length, 1) - 1) / 25;
    cols.assign(off.back() * 25 + 1, 0);
    for (size_t i = 0; i < db.entries.size(); i++) std::memcpy(cols.data() + off[i] * 25, db.entry(i), (size_t) (off[i + 1] - off[i]) * 25);
}
// DBReader::getAminoAcidDBSize of a profile DB (DBReader.cpp:589-598): dataSize / 25 - entries
uint64_t profileDbResidues(const mk::Database &db) {
    uint64_t dataSize = 0;
    for (const mk::DbEntry &e : db.entries) dataSize += e.length;
    return dataSize / 25 - db.entries.size();
}
// the e-value threshold of the inverted search: scaled by #queries / #targets of the ORIGINAL search (Search.cpp:366-368, float division)
// and handed to the modules as text (Parameters::createParameterString streams the double with 6 significant digits)
double invertedEvalue(double evalThr, size_t nFragments, size_t nProfiles) {
    evalThr *= ((float) nFragments) / nProfiles;
    char txt[64];
    snprintf(txt, sizeof(txt), "%g", evalThr);
    return strtod(txt, nullptr);
}

// target side of a command: from the precomputed index DB when there is one (the path itself is an index DB, or <path>.idx exists and
// MMSEQS_IGNORE_INDEX is unset -- PrefilteringIndexReader::searchForIndex, PrefilteringIndexReader.cpp:568-579), else built from the
// sequence DB.  keys[i] = DB key of target i.
struct TargetSide { mk_targetdb *T = nullptr; std::vector<uint32_t> keys; bool fromIndex = false; };
int openTargetDb(const mk::Database &tdb, const mk_params &P, TargetSide &ts);
int openTarget(const std::string &path, const mk_params &P, TargetSide &ts) {
    std::string idx;
    {
        FILE *f = fopen((path + ".dbtype").c_str(), "rb");
        int32_t t = -1;
        if (f) { if (fread(&t, 4, 1, f) != 1) t = -1; fclose(f); }
        if (t >= 0 && (t & 0xFFFF) == 9) idx = path;
        else if (!P.profile_search && !launcherEnv("MMSEQS_IGNORE_INDEX") && mk::Database::exists(path + ".idx.dbtype")) idx = path + ".idx";
        // (profile queries need their own masking background and an unfiltered index: a sequence-search .idx next to the DB is not used)
    }
    if (!idx.empty()) {
        if (mk_targetdb_open_index(idx.c_str(), &P, &ts.T) != MK_OK) return die("%s", mk_last_error());
        const uint32_t *k; uint32_t n;
        mk_targetdb_keys(ts.T, &k, &n);
        ts.keys.assign(k, k + n);
        ts.fromIndex = true;
        return 0;
    }
    mk::Database tdb;
    const std::string e = tdb.open(path);
    if (!e.empty()) return die("%s", e);
    return openTargetDb(tdb, P, ts);
}
int openTargetDb(const mk::Database &tdb, const mk_params &P, TargetSide &ts) {
    if ((tdb.dbtype & 0xFFFF) != mk::DBTYPE_AMINO_ACIDS) return die("only amino-acid target databases are implemented (profile targets: SURVEY 8f-4)%s");
    std::vector<uint8_t> tres;
    std::vector<uint64_t> toff;
    tres.reserve(tdb.data.size());
    {
        toff.assign(tdb.entries.size() + 1, 0);
        for (size_t i = 0; i < tdb.entries.size(); i++) toff[i + 1] = toff[i] + tdb.seqLen(i);
        tres.assign(toff.back() + 1, 0);
        for (size_t i = 0; i < tdb.entries.size(); i++) mk_encode(tdb.entry(i), tdb.seqLen(i), tres.data() + toff[i]);
    }
    if (mk_targetdb_create(tres.data(), toff.data(), (uint32_t) tdb.entries.size(), &P, &ts.T) != MK_OK) return die("%s", mk_last_error());
    ts.keys.resize(tdb.entries.size());
    for (size_t i = 0; i < ts.keys.size(); i++) ts.keys[i] = tdb.entries[i].key;
    // the alignment order's last tie-break is the DB key (Matcher::compareHits), which need not follow the target order
    if (mk_targetdb_set_keys(ts.T, ts.keys.data(), (uint32_t) ts.keys.size()) != MK_OK) return die("%s", mk_last_error());
    return 0;
}

// One process per GPU over one query DB: worker `rank` of `world` takes the reference's residue-balanced query range
// and writes <out>_<rank>; worker 0 waits for the other shards (their .dbtype is written last) and merges them into <out>.  No
// collective: the workers only meet in the file system.  --shard r/N, else RANK / WORLD_SIZE of the launcher (torch.distributed.run,
// the RUNNER hook of blastp.sh:70,85).
struct Shard { int rank = 0, world = 1; std::string token; };
// what identifies ONE launch of the workers: the launcher's run id, else a hash of the command line (without the per-worker flags) AND of
// the size and modification time of every input the command line names.  A shard left behind by an earlier, crashed launch carries
// another token (or none) and is never merged -- unless it was made by the very same command over the very same input files, in which
// case it holds what this launch would write (the commands are deterministic; a shard's .dbtype appears last, when it is complete).
std::string launchToken(int argc, char **argv) {
    if (const char *id = launcherEnv("TORCHELASTIC_RUN_ID")) if (*id && strcmp(id, "none") != 0) return std::string("run-") + id;
    if (const char *id = launcherEnv("MK_LAUNCH_ID")) if (*id) return std::string("id-") + id;
    uint64_t h = 1469598103934665603ull;
    int positional = 0;                                     // the first positional argument after the command is always an input DB, the second one
    const int inputs = argc > 1 && !strcmp(argv[1], "extractorfs") ? 1 : 2;     // too except for extractorfs; outputs are not looked at (they change while the workers run)
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--shard") || !strcmp(argv[i], "--gpu")) { i++; continue; }
        for (const char *c = argv[i]; *c; c++) { h ^= (unsigned char) *c; h *= 1099511628211ull; }
        h ^= 0xFF; h *= 1099511628211ull;
        if (argv[i][0] == '-' && !(argv[i][1] >= '0' && argv[i][1] <= '9')) { i++; if (i < argc) { for (const char *c = argv[i]; *c; c++) { h ^= (unsigned char) *c; h *= 1099511628211ull; } h ^= 0xFE; h *= 1099511628211ull; } continue; }
        if (i > 1 && positional++ < inputs)
            for (const char *ext : {"", ".index"}) {
                struct stat st;
                if (stat((std::string(argv[i]) + ext).c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
                const uint64_t v[3] = {(uint64_t) st.st_size, (uint64_t) st.st_mtim.tv_sec, (uint64_t) st.st_mtim.tv_nsec};
                for (uint64_t x : v) for (int b = 0; b < 8; b++) { h ^= (x >> (8 * b)) & 0xFF; h *= 1099511628211ull; }
            }
    }
    char buf[32];
    snprintf(buf, sizeof(buf), "argv-%016llx", (unsigned long long) h);
    return buf;
}
int shardOf(const Args &a, Shard &sh, int argc, char **argv) {
    auto it = a.opt.find("--shard");
    if (it != a.opt.end() && !it->second.empty()) {
        if (sscanf(it->second.c_str(), "%d/%d", &sh.rank, &sh.world) != 2) return die("--shard wants rank/world, got %s", it->second);
    } else if (launcherEnv("RANK") && launcherEnv("WORLD_SIZE")) {
        sh.rank = atoi(launcherEnv("RANK")); sh.world = atoi(launcherEnv("WORLD_SIZE"));
    }
    if (sh.world < 1 || sh.rank < 0 || sh.rank >= sh.world) return die("bad shard %s", std::to_string(sh.rank) + "/" + std::to_string(sh.world));
    sh.token = launchToken(argc, argv);
    return 0;
}
// first thing a sharded worker does, before any compute: its own leftovers of an earlier launch disappear, and its token says which launch
// the shard it is about to write belongs to
void beginShard(const std::string &out, const Shard &sh) {
    if (sh.world <= 1) return;
    const std::string mine = out + "_" + std::to_string(sh.rank);
    for (const char *ext : {".dbtype", ".index", "", ".orfs", ".failed", ".token"}) remove((mine + ext).c_str());
    g_failMarker = mine + ".failed";
    FILE *f = fopen((mine + ".token").c_str(), "w");
    if (f) { fputs(sh.token.c_str(), f); fclose(f); }
}
long shardTimeoutTicks() {            // 50 ms ticks; MK_SHARD_TIMEOUT_S (default two hours): a peer that died without a marker
    const char *e = launcherEnv("MK_SHARD_TIMEOUT_S");
    const long sec = e && atol(e) > 0 ? atol(e) : 7200;
    return sec * 20;
}
// waits until worker r has published `file` (a path under <out>_<r>) in THIS launch; false: the worker failed or never showed up
bool waitForPeer(const std::string &out, int r, const std::string &file, const std::string &token) {
    const std::string base = out + "_" + std::to_string(r);
    const long limit = shardTimeoutTicks();
    for (long waited = 0; waited < limit; waited++) {
        if (mk::Database::exists(base + ".failed")) { die("worker %s failed", std::to_string(r)); return false; }
        if (mk::Database::exists(file)) {
            char buf[128] = {0};
            FILE *f = fopen((base + ".token").c_str(), "r");
            if (f) { if (!fgets(buf, sizeof(buf), f)) buf[0] = 0; fclose(f); }
            if (token == buf) return true;                      // else: a stale file of an earlier launch, its owner has not started yet
        }
        usleep(50000);
    }
    die("gave up waiting for %s", file);
    return false;
}
int finishShards(const std::string &out, const Shard &sh, int dbtype) {
    if (sh.world <= 1 || sh.rank != 0) return EXIT_SUCCESS;
    for (int r = 1; r < sh.world; r++)
        if (!waitForPeer(out, r, out + "_" + std::to_string(r) + ".dbtype", sh.token)) return EXIT_FAILURE;
    const std::string e = mk::mergeShards(out, sh.world, dbtype);
    if (!e.empty()) return die("%s", e);
    for (int r = 0; r < sh.world; r++) remove((out + "_" + std::to_string(r) + ".token").c_str());
    return EXIT_SUCCESS;
}

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// returns 0 and *S = nullptr on a worker other than 0 of a sharded launch (its share went to worker 0)
int invertedProfileSearch(const mk::Database &pdb, mk_targetdb *F, uint64_t nFrag, const mk_params &P, mk_swapped **S, uint64_t &nHits, uint64_t &nAln,
                          const Shard &sh, const std::string &outBase);

// search <i:fragmentDB> <i:profileDB> <o:alignmentDB> <tmpDir>: what Search.cpp:357-399 + searchslicedtargetprofile.sh compute -- the profiles
// against the fragments, swapped: one record per fragment key with the profiles that hit it.  Fragment numbering = the DB's data order, like
// the reference's prefilter numbers its targets.
int searchProfileTargets(const Args &a, mk_params P, const mk::Database &qdb, const std::string &outBase, const Shard &sh, double t0) {
    const std::string outPath = outBase;                                        // (worker 0 writes the whole result: the shards are the profiles)
    mk::Database pdb;
    std::string e = pdb.open(a.pos[1]);
    if (!e.empty()) return die("%s", e);
    const size_t nFrag = qdb.entries.size(), nProf = pdb.entries.size();
    std::vector<uint8_t> fres;
    std::vector<uint64_t> foff;
    encodeDb(qdb, fres, foff);
    const double evalThrUser = P.evalue_thr;
    P.profile_search = 1;
    P.max_seqs = (int) std::max<uint64_t>(300, nFrag);                       // Search.cpp:372
    P.evalue_thr = nProf ? invertedEvalue(evalThrUser, nFrag, nProf) : evalThrUser;
    mk_targetdb *F = nullptr;
    if (mk_targetdb_create(fres.data(), foff.data(), (uint32_t) nFrag, &P, &F) != MK_OK) return die("%s", mk_last_error());
    std::vector<uint32_t> fkeys(nFrag);
    for (size_t i = 0; i < nFrag; i++) fkeys[i] = qdb.entries[i].key;
    if (mk_targetdb_set_keys(F, fkeys.data(), (uint32_t) nFrag) != MK_OK) return die("%s", mk_last_error());
    uint64_t nHits = 0, nAln = 0;
    mk_swapped *S = nullptr;
    if (int rc = invertedProfileSearch(pdb, F, nFrag, P, &S, nHits, nAln, sh, outBase)) return rc;
    if (!S) { mk_targetdb_destroy(F); return EXIT_SUCCESS; }                    // a worker whose alignments went to worker 0
    const mk_alignment *sw; const uint64_t *soff;
    mk_swapped_result(S, &sw, &soff);
    mk::DatabaseWriter w(outPath, mk::DBTYPE_ALIGNMENT_RES);
    e = w.open();
    if (!e.empty()) return die("%s", e);
    std::string buf;
    char line[512];
    for (size_t i = 0; i < nFrag; i++) {
        buf.clear();
        for (uint64_t k = soff[i]; k < soff[i + 1]; k++) buf.append(line, mk_format_alignment(line, &sw[k]));
        w.write(qdb.entries[i].key, buf.data(), buf.size());
    }
    e = w.close();
    if (!e.empty()) return die("%s", e);
    fprintf(stderr, "search (profile targets): %zu profiles x %zu fragments, %llu prefilter hits, %llu alignments, %.2f s\n", nProf, nFrag, (unsigned long long) nHits,
            (unsigned long long) nAln, now() - t0);
    mk_swapped_destroy(S);
    mk_targetdb_destroy(F);
    for (int r = 0; r < sh.world && sh.world > 1; r++) remove((outBase + "_" + std::to_string(r) + ".token").c_str());
    return EXIT_SUCCESS;
}


// ---- target splits (TARGET_DB_SPLIT) -------------------------------------------------------------------------------------------------
// Prefiltering::estimateMemoryConsumption (Prefiltering.cpp:1067-1104): what the REFERENCE needs in host memory for one split
uint64_t refMemoryConsumption(int split, uint64_t dbSize, uint64_t resSize, uint64_t maxResListLen, int alphabetSize, int kmerSize, int threads) {
    const uint64_t dbSizeSplit = dbSize / (uint64_t) split;
    const uint64_t residueSize = resSize / (uint64_t) split * 7;
    const uint64_t indexTableSize = static_cast<uint64_t>(pow(alphabetSize, kmerSize)) * 8;
    const uint64_t threadSize = (uint64_t) ((double) threads * ((double) (dbSizeSplit * 2 * 6) + (double) dbSizeSplit * 1.5 * 7.0 + (double) (maxResListLen * 12) + (double) (dbSizeSplit * 2 * 7 * 2)));
    const uint64_t dbReaderSize = dbSize * (24 + 4);
    const uint64_t extendedMatrix = 8 * static_cast<uint64_t>(pow(pow(alphabetSize, 3), 2)) + (uint64_t) (8 * pow(pow(alphabetSize, 2), 2));
    const uint64_t background = dbSize * 22;
    return residueSize + indexTableSize + threadSize + background + extendedMatrix + dbReaderSize;
}
// ByteParser::parse (commons/ByteParser.h:11-60): digits + B / K / M / G / T, no unit = M
uint64_t parseBytes(const std::string &v) {
    if (v.empty()) return 0;
    uint64_t unit = 1ull << 20;
    std::string digits = v;
    const char last = v[v.size() - 1];
    if (!(last >= '0' && last <= '9')) {
        digits = v.substr(0, v.size() - 1);
        switch (last) { case 't': case 'T': unit = 1ull << 40; break; case 'g': case 'G': unit = 1ull << 30; break; case 'm': case 'M': unit = 1ull << 20; break;
                        case 'k': case 'K': unit = 1ull << 10; break; case 'b': case 'B': unit = 1; break; default: return ~0ull; }
    }
    return strtoull(digits.c_str(), nullptr, 10) * unit;
}
// What Prefiltering::setupSplit (Prefiltering.cpp:273-377) makes of --split / --split-mode / --split-memory-limit:
//   --split N --split-mode 0         N target splits, as given;
//   --split 0 (the default), mode 0 or 2, --split-memory-limit L > 0: the smallest N (and, with -k 0, the k) whose estimated need fits 0.9 L --
//                                    the reference's own estimate and search order (optimizeSplit, :1143-1176), L less what its DB readers hold
//                                    (MemoryTracker: ~32 bytes per entry of the two sequence DBs, an approximation);
//   no limit given:                  the limit is this GPU's memory -- one split unless masked residues + index of the whole database cannot
//                                    live in HBM next to the search's scratch (then the smallest N that can).
// N > 1 changes the result the way it changes the reference's: k from the residues per split, --max-seqs cut per split, BINSIZE per split.
struct SplitPlan { int splits = 1; int kmerSize = 0; int maxSeqs = 300; bool chosen = false; };
int planTargetSplits(const Args &a, const mk_params &P, size_t nT, uint64_t aaSize, size_t nQ, SplitPlan &plan) {
    auto get = [&](const char *k) -> const std::string * { auto it = a.opt.find(k); return it == a.opt.end() ? nullptr : &it->second; };
    const int splitArg = get("--split") ? atoi(get("--split")->c_str()) : 0;
    const int modeArg = get("--split-mode") ? atoi(get("--split-mode")->c_str()) : 2;
    const uint64_t limitArg = get("--split-memory-limit") ? parseBytes(*get("--split-memory-limit")) : 0;
    if (limitArg == ~0ull) return die("--split-memory-limit %s: not a size", *get("--split-memory-limit"));
    const int threads = get("--threads") ? std::max(1, atoi(get("--threads")->c_str())) : mk_host_threads();
    const size_t maxRes = std::min<size_t>(nT, (size_t) P.max_seqs);                                      // Prefiltering.cpp:169
    plan.splits = 1; plan.kmerSize = P.kmer_size; plan.maxSeqs = P.max_seqs;
    if (modeArg == 1) return 0;                                                                            // query splits leave the results alone
    if (splitArg > 1 && modeArg == 0) plan.splits = splitArg;
    else if (splitArg == 0) {
        if (limitArg > 0) {
            const uint64_t tracker = 32ull * ((uint64_t) nT + (uint64_t) nQ);
            const uint64_t limit = limitArg > tracker ? limitArg - tracker : 0;
            const int kAll = P.kmer_size ? P.kmer_size : (aaSize < 3350000000ull ? 6 : 7);
            if ((double) refMemoryConsumption(1, nT, aaSize, maxRes, 20, kAll, threads) > 0.9 * (double) limit) {
                int found = -1, foundK = 0;
                for (int k = P.kmer_size ? P.kmer_size : 7; k >= (P.kmer_size ? P.kmer_size : 6) && found < 0; k--)
                    for (int sp = 1; sp < 1000; sp++) {
                        if (!P.kmer_size && k == 6 && aaSize / (uint64_t) sp >= 3350000000ull) continue;          // getUpperBoundAACountForKmerSize
                        if ((double) refMemoryConsumption(sp, nT, aaSize, 0, 20, k, threads) < 0.9 * (double) limit) { found = sp; foundK = k; break; }
                    }
                if (found < 0) return die("Cannot fit databases into %s. Please use a computer with more main memory.", *get("--split-memory-limit"));
                plan.splits = (int) std::min<size_t>(nT, (size_t) found);
                if (!P.kmer_size) plan.kmerSize = foundK;
                plan.chosen = true;
            }
        } else {
            // this GPU: masked + unmasked residues, 8-byte entries (one per residue at most), slots + bits of the k-mer table, ~24 GB of search scratch
            uint64_t freeB = 0, totalB = 0;
            if (mk_device_memory(&freeB, &totalB) != MK_OK) totalB = 288ull << 30;
            for (int sp = 1; sp < 1000; sp++) {
                const uint64_t per = aaSize / (uint64_t) sp;
                const int k = P.kmer_size ? P.kmer_size : (per < 3350000000ull ? 6 : 7);
                const double need = 2.0 * (double) aaSize + 9.0 * (double) per + (k == 7 ? 1.28e9 : 6.4e7) * 12.2 + 24.0 * 1073741824.0;
                if (need < 0.9 * (double) totalB) { plan.splits = sp; plan.chosen = sp > 1; break; }
                if (sp == 999) return die("the target database does not fit this GPU even in 999 splits%s");
            }
        }
    }
    if ((size_t) plan.splits > nT) return die("split was set to %s but the db to split has fewer sequences", std::to_string(plan.splits));
    if (plan.splits > 1) {
        if (!plan.kmerSize) plan.kmerSize = aaSize / (uint64_t) plan.splits < 3350000000ull ? 6 : 7;               // Prefiltering.cpp:352-355
        const size_t fourTimesStdDeviation = (size_t) (4 * sqrt(static_cast<double>(maxRes) / static_cast<double>(plan.splits)));
        plan.maxSeqs = (int) std::max<size_t>(1, (maxRes / (size_t) plan.splits) + fourTimesStdDeviation);         // :359-362
    }
    return 0;
}

// Prefilter of batch Q against the targets in `plan.splits` residue-balanced ranges (Prefiltering::runSplit with TARGET_DB_SPLIT, :733-750): every
// range is masked, indexed and searched on its own -- its own BINSIZE, the reduced --max-seqs -- and a query's lists are joined and sorted by
// (score, key) like Prefiltering::mergeTargetSplits (:379-496) does; the joined list is NOT cut again.  The result is installed in Q with
// seq_id = position in tdb (mk_prefilter_result_set), so that mk_align / the writers go on as after a one-piece prefilter.
int splitPrefilter(mk_queries *Q, size_t nq, const mk::Database &tdb, const mk_params &P, const SplitPlan &plan, uint64_t &total) {
    mk_params PS = P;
    PS.kmer_size = plan.kmerSize;
    PS.max_seqs = plan.maxSeqs;
    std::vector<std::vector<mk_hit>> merged(nq);
    for (int sp = 0; sp < plan.splits; sp++) {
        size_t first = 0, count = 0;
        mk::decomposeByLength(tdb.entries, sp, plan.splits, first, count);
        if (count == 0) continue;
        std::vector<uint64_t> toff(count + 1, 0);
        for (size_t i = 0; i < count; i++) toff[i + 1] = toff[i] + tdb.seqLen(first + i);
        std::vector<uint8_t> tres(toff.back() + 1, 0);
#pragma omp parallel for schedule(dynamic, 256)
        for (size_t i = 0; i < count; i++) mk_encode(tdb.entry(first + i), tdb.seqLen(first + i), tres.data() + toff[i]);
        mk_targetdb *TS = nullptr;
        if (mk_targetdb_create(tres.data(), toff.data(), (uint32_t) count, &PS, &TS) != MK_OK) return die("%s", mk_last_error());
        if (mk_prefilter(TS, Q, &PS) != MK_OK) return die("%s", mk_last_error());
        printPrefilterStatistics(Q, PS);                  // the reference logs the statistics of every split's run (Prefiltering::runSplit, :889-904)
        const mk_hit *hits; const uint64_t *hoff;
        mk_prefilter_result(Q, &hits, &hoff);
        for (size_t i = 0; i < nq; i++)
            for (uint64_t h = hoff[i]; h < hoff[i + 1]; h++) { mk_hit x = hits[h]; x.seq_id += (uint32_t) first; merged[i].push_back(x); }
        mk_targetdb_destroy(TS);
    }
    std::vector<uint64_t> off(nq + 1, 0);
    for (size_t i = 0; i < nq; i++) off[i + 1] = off[i] + merged[i].size();
    std::vector<mk_hit> all(off[nq] + 1);
#pragma omp parallel for schedule(dynamic, 1024)
    for (size_t i = 0; i < nq; i++) {
        std::vector<mk_hit> &v = merged[i];
        std::sort(v.begin(), v.end(), [&](const mk_hit &x, const mk_hit &y) {          // hit_t::compareHitsByScoreAndId on the parsed lines (seqId = key)
            if (std::abs(x.pref_score) != std::abs(y.pref_score)) return std::abs(x.pref_score) > std::abs(y.pref_score);
            return tdb.entries[x.seq_id].key < tdb.entries[y.seq_id].key;
        });
        std::copy(v.begin(), v.end(), all.begin() + (std::ptrdiff_t) off[i]);
    }
    total = off[nq];
    if (mk_prefilter_result_set(Q, all.data(), off.data()) != MK_OK) return die("%s", mk_last_error());
    return 0;
}

// the targets as the alignment stage needs them: residues in HBM, no index (mk_targetdb_create_sequences)
int openTargetSequences(const mk::Database &tdb, const mk_params &P, TargetSide &ts) {
    std::vector<uint64_t> toff(tdb.entries.size() + 1, 0);
    for (size_t i = 0; i < tdb.entries.size(); i++) toff[i + 1] = toff[i] + tdb.seqLen(i);
    std::vector<uint8_t> tres(toff.back() + 1, 0);
#pragma omp parallel for schedule(dynamic, 256)
    for (size_t i = 0; i < tdb.entries.size(); i++) mk_encode(tdb.entry(i), tdb.seqLen(i), tres.data() + toff[i]);
    if (mk_targetdb_create_sequences(tres.data(), toff.data(), (uint32_t) tdb.entries.size(), &P, &ts.T) != MK_OK) return die("%s", mk_last_error());
    ts.keys.resize(tdb.entries.size());
    for (size_t i = 0; i < ts.keys.size(); i++) ts.keys[i] = tdb.entries[i].key;
    if (mk_targetdb_set_keys(ts.T, ts.keys.data(), (uint32_t) ts.keys.size()) != MK_OK) return die("%s", mk_last_error());
    return 0;
}

// mode: 0 = prefilter, 1 = align, 2 = search (the `search` workflow's two modules as one pipelined pass: <queryDB> <targetDB> <alignmentDB>
// <tmpDir>, nothing written in between -- blastp.sh:70,85 without the pref_0 round trip)
int cmdPrefilterOrAlign(int mode, int argc, char **argv) {
    const bool isAlign = mode != 0, isSearch = mode == 2;
    Args a;
    if (int rc = parse(argc, argv, a)) return rc;
    const size_t need = isAlign ? 4 : 3;
    if (a.pos.size() != need) {
        fprintf(stderr, "usage: metaeuk-amd %s <i:queryDB> <i:targetDB> %s[options]\n", isSearch ? "search" : (isAlign ? "align" : "prefilter"),
                isSearch ? "<o:alignmentDB> <tmpDir> " : (isAlign ? "<i:resultDB> <o:alignmentDB> " : "<o:prefilterDB> "));
        return EXIT_FAILURE;
    }
    mk_params P;
    int gpu = 0;
    if (isSearch && a.opt.find("-s") == a.opt.end()) a.opt["-s"] = "5.7";   // the search workflow's own default (Search.cpp:24)
    if (isAlign && a.opt.find("-e") == a.opt.end()) a.opt["-e"] = "0.001"; // Parameters.cpp:2414 (predictexons passes -e 100 explicitly)
    // the module's own default --alignment-mode 0 resolves to score-only output (Alignment.cpp:168-178), which is not built:
    // the caller has to ask for mode 2, as predictexons does (PredictExons.cpp:13)
    if (isAlign && a.opt.find("--alignment-mode") == a.opt.end())
        return die("%s needs --alignment-mode 2 (the module default, 0 = score only, is not implemented)", isSearch ? "search" : "align");
    if (int rc = fillParams(a, P, gpu)) return rc;
    // --alignment-output-mode 1 (the per-slice `align` of the sliced profile workflow): the accepted targets' keys only
    const bool keyListOut = isAlign && a.opt.count("--alignment-output-mode") && a.opt["--alignment-output-mode"] == "1";
    if (a.opt.count("--alignment-output-mode") && a.opt["--alignment-output-mode"] != "0" && !keyListOut)
        return die("--alignment-output-mode %s: only 0 (alignment records) and 1 (key lists) are implemented", a.opt["--alignment-output-mode"]);
    const double t0 = now();
    mk::Database qdb;
    std::string e = qdb.open(a.pos[0]);
    if (!e.empty()) return die("%s", e);
    const bool profileQueries = (qdb.dbtype & 0xFFFF) == DBTYPE_HMM_PROFILE;      // the modules as searchslicedtargetprofile.sh calls them
    if (!profileQueries && (qdb.dbtype & 0xFFFF) != mk::DBTYPE_AMINO_ACIDS) return die("only amino-acid and profile query databases are implemented%s");
    if (profileQueries && isSearch) return die("search with a profile query DB is not a workflow of the reference: use predictexons with a profile target DB, or prefilter / align / swapresults%s");
    if (profileQueries) P.profile_search = 1;
    Shard sh;
    if (int rc = shardOf(a, sh, argc, argv)) return rc;
    beginShard(a.pos[isAlign && !isSearch ? 3 : 2], sh);
    if (sh.world > 1) {                                       // this worker's queries only
        size_t first = 0, count = 0;
        mk::decomposeByLength(qdb.entries, sh.rank, sh.world, first, count);
        qdb.entries.erase(qdb.entries.begin() + (std::ptrdiff_t) (first + count), qdb.entries.end());
        qdb.entries.erase(qdb.entries.begin(), qdb.entries.begin() + (std::ptrdiff_t) first);
    }
    const std::string outBase = a.pos[isAlign && !isSearch ? 3 : 2];
    const std::string outPath = sh.world > 1 ? outBase + "_" + std::to_string(sh.rank) : outBase;
    if (mk_init(gpu) != MK_OK) return die("%s", mk_last_error());
    std::vector<uint8_t> qres;
    std::vector<uint64_t> qoff;
    if (profileQueries) profileColumns(qdb, qres, qoff); else encodeDb(qdb, qres, qoff);
    // TARGET_DB_SPLIT (explicit, or chosen from --split-memory-limit / this GPU's memory: planTargetSplits) changes the prefilter's results
    // (per-split --max-seqs, BINSIZE, merge order); `prefilter` and `search` honour it, `align` reads whatever prefilter DB it is given
    if (profileQueries && a.opt.count("--split") && a.opt.count("--split-mode") && a.opt["--split-mode"] == "0" && atoi(a.opt["--split"].c_str()) > 1)
        return die("--split-mode 0 with profile queries is not implemented%s");
    {   // `search <fragmentDB> <profileDB>`: Search.cpp:357-399 turns a profile TARGET database into the inverted sliced search
        FILE *f = fopen((a.pos[1] + ".dbtype").c_str(), "rb");
        int32_t t = -1;
        if (f) { if (fread(&t, 4, 1, f) != 1) t = -1; fclose(f); }
        if (t >= 0 && (t & 0xFFFF) == DBTYPE_HMM_PROFILE) {
            if (!isSearch || profileQueries) return die("a profile target database is searched by `search` / `predictexons` (the inverted sliced search), not by %s", isAlign ? "align" : "prefilter");
            // Search.cpp:357-399 takes the inverted sliced search only with --exhaustive-search 1 (predictexons sets it itself for profile targets,
            // PredictExons.cpp:22-26); without it the reference runs the target-side k-mer search of searchtargetprofile.sh (:251-257), which is not built
            {
                auto ex = a.opt.find("--exhaustive-search");
                if (ex == a.opt.end() || ex->second == "0")
                    return die("search against a profile target database without --exhaustive-search 1 (the target-side k-mer profile search, Search.cpp:251-257) is not implemented: pass --exhaustive-search 1%s");
            }
            // sharded: the workers split the PROFILES (every worker indexes all fragments: their number enters the e-value threshold,
            // the e-values and --max-seqs); qdb was cut to this worker's queries above -- the fragments are needed whole
            mk::Database fdb;
            if (sh.world > 1) { e = fdb.open(a.pos[0]); if (!e.empty()) return die("%s", e); }
            return searchProfileTargets(a, P, sh.world > 1 ? fdb : qdb, outBase, sh, t0);
        }
    }
    if (isSearch) {
        auto ex = a.opt.find("--exhaustive-search");
        if (ex != a.opt.end() && ex->second != "0") return die("--exhaustive-search 1 with a sequence target database is not implemented%s");
    }
    // the target side: a sequence DB is looked at first (its size decides about target splits); an index DB holds one split
    TargetSide ts;
    SplitPlan plan;
    plan.kmerSize = P.kmer_size; plan.maxSeqs = P.max_seqs;
    mk::Database tdbSeq;
    bool haveSeqDb = false;
    {
        FILE *f = fopen((a.pos[1] + ".dbtype").c_str(), "rb");
        int32_t t = -1;
        if (f) { if (fread(&t, 4, 1, f) != 1) t = -1; fclose(f); }
        const bool explicitSplit = a.opt.count("--split") && atoi(a.opt["--split"].c_str()) > 1 && a.opt.count("--split-mode") && a.opt["--split-mode"] == "0";
        const bool hasIdx = (t >= 0 && (t & 0xFFFF) == 9) || (!P.profile_search && !launcherEnv("MMSEQS_IGNORE_INDEX") && mk::Database::exists(a.pos[1] + ".idx.dbtype"));
        if (explicitSplit && t >= 0 && (t & 0xFFFF) == 9) return die("--split with --split-mode 0 needs an amino-acid sequence DB as the target (index DBs hold one split)%s");
        if (t >= 0 && (t & 0xFFFF) == mk::DBTYPE_AMINO_ACIDS && (isAlign && !isSearch ? true : (explicitSplit || !hasIdx))) {
            e = tdbSeq.open(a.pos[1]);
            if (!e.empty()) return die("%s", e);
            haveSeqDb = true;
            if (!isAlign || isSearch) {
                uint64_t aaSize = 0;
                for (size_t i = 0; i < tdbSeq.entries.size(); i++) aaSize += tdbSeq.seqLen(i);
                if (!profileQueries) { if (int rc = planTargetSplits(a, P, tdbSeq.entries.size(), aaSize, qdb.entries.size(), plan)) return rc; }
            }
        }
    }
    const int targetSplits = plan.splits;
    if (isAlign && !isSearch && haveSeqDb) { if (int rc = openTargetSequences(tdbSeq, P, ts)) return rc; }      // align: residues only, no index is built
    else if (targetSplits == 1) { if (int rc = haveSeqDb ? openTargetDb(tdbSeq, P, ts) : openTarget(a.pos[1], P, ts)) return rc; }
    else if (isSearch) { if (int rc = openTargetSequences(tdbSeq, P, ts)) return rc; }                          // the alignment half of a split search
    mk_targetdb *T = ts.T;
    const std::vector<uint32_t> &tkeys = ts.keys;
    mk_queries *Q = nullptr;
    if ((profileQueries ? mk_profiles_create(qres.data(), qoff.data(), (uint32_t) qdb.entries.size(), &P, &Q)
                        : mk_queries_create(qres.data(), qoff.data(), (uint32_t) qdb.entries.size(), &P, &Q)) != MK_OK) return die("%s", mk_last_error());
    const size_t nq = qdb.entries.size();
    char line[512];
    std::string buf;
    if (!isAlign && targetSplits > 1) {
        // TARGET_DB_SPLIT (Prefiltering.cpp:352-361,379-496): splitPrefilter
        uint64_t total = 0;
        if (int rc = splitPrefilter(Q, nq, tdbSeq, P, plan, total)) return rc;
        const mk_hit *hits; const uint64_t *hoff;
        mk_prefilter_result(Q, &hits, &hoff);
        mk::DatabaseWriter w(outPath, mk::DBTYPE_PREFILTER_RES);
        e = w.open();
        if (!e.empty()) return die("%s", e);
        for (size_t i = 0; i < nq; i++) {
            buf.clear();
            for (uint64_t h = hoff[i]; h < hoff[i + 1]; h++) buf.append(line, mk_format_hit(line, tdbSeq.entries[hits[h].seq_id].key, hits[h].pref_score, hits[h].diagonal));
            w.write(qdb.entries[i].key, buf.data(), buf.size());
        }
        e = w.close();
        if (!e.empty()) return die("%s", e);
        fprintf(stderr, "prefilter: %zu queries x %zu targets in %d target splits%s (--max-seqs %d per split, k = %d), %llu hits, %.2f s\n", nq, tdbSeq.entries.size(), targetSplits,
                plan.chosen ? " [chosen from the memory limit]" : "", plan.maxSeqs, plan.kmerSize, (unsigned long long) total, now() - t0);
    } else if (!isAlign) {
        if (mk_prefilter(T, Q, &P) != MK_OK) return die("%s", mk_last_error());
        const mk_hit *hits; const uint64_t *hoff;
        mk_prefilter_result(Q, &hits, &hoff);
        mk::DatabaseWriter w(outPath, mk::DBTYPE_PREFILTER_RES);
        e = w.open();
        if (!e.empty()) return die("%s", e);
        for (size_t i = 0; i < nq; i++) {
            buf.clear();
            for (uint64_t h = hoff[i]; h < hoff[i + 1]; h++)    // seqId -> dbKey (Prefiltering.cpp:845-852)
                buf.append(line, mk_format_hit(line, tkeys[hits[h].seq_id], hits[h].pref_score, hits[h].diagonal));
            w.write(qdb.entries[i].key, buf.data(), buf.size());
        }
        e = w.close();
        if (!e.empty()) return die("%s", e);
        printPrefilterStatistics(Q, P);
        fprintf(stderr, "prefilter: %zu queries x %zu targets%s, %llu hits, %.2f s\n", nq, tkeys.size(), ts.fromIndex ? " (precomputed index)" : "", (unsigned long long) hoff[nq], now() - t0);
    } else {
        std::vector<mk_hit> hits;
        if (isSearch && targetSplits > 1) {
            // search over target splits: the prefilter of every split, the joined lists, then the alignment against the whole database
            uint64_t total = 0;
            if (int rc = splitPrefilter(Q, nq, tdbSeq, P, plan, total)) return rc;
            if (mk_align(T, Q, &P) != MK_OK) return die("%s", mk_last_error());
            hits.resize(total);
            fprintf(stderr, "search: %d target splits%s (--max-seqs %d per split, k = %d)\n", targetSplits, plan.chosen ? " [chosen from the memory limit]" : "", plan.maxSeqs, plan.kmerSize);
        } else if (isSearch) {
            if (mk_search(T, Q, &P) != MK_OK) return die("%s", mk_last_error());
            const mk_hit *hp; const uint64_t *ho;
            mk_prefilter_result(Q, &hp, &ho);
            hits.resize(ho[nq]);                                     // (only their number is reported)
            printPrefilterStatistics(Q, P);
        } else {
        // read the prefilter DB: key \t score \t diagonal lines (QueryMatcher::parsePrefilterHit, QueryMatcher.h:87-102)
        mk::Database pdb;
        e = pdb.open(a.pos[2]);
        if (!e.empty()) return die("%s", e);
        std::map<uint32_t, uint32_t> qKeyToIdx, tKeyToIdx;
        for (size_t i = 0; i < nq; i++) qKeyToIdx[qdb.entries[i].key] = (uint32_t) i;
        for (size_t i = 0; i < tkeys.size(); i++) tKeyToIdx[tkeys[i]] = (uint32_t) i;
        std::vector<std::vector<mk_hit>> perQ(nq);
        for (size_t i = 0; i < pdb.entries.size(); i++) {
            auto qi = qKeyToIdx.find(pdb.entries[i].key);
            if (qi == qKeyToIdx.end()) {
                if (sh.world > 1) continue;                      // another worker's query
                return die("Query sequence %s is required in the prefiltering, but is not contained in the query sequence database", std::to_string(pdb.entries[i].key));
            }
            const char *p = pdb.entry(i);
            while (*p != '\0') {
                char *q;
                mk_hit h;
                const uint32_t key = (uint32_t) strtoul(p, &q, 10);
                // a line with the key alone (the key lists of --alignment-output-mode 1 that the sliced workflow re-aligns) or with other
                // columns than a prefilter hit's three: diagonal 0 (Alignment.cpp:352-361)
                h.pref_score = 0; h.diagonal = 0;
                if (*q == '\t') {
                    h.pref_score = (int32_t) strtol(q, &q, 10);
                    if (*q == '\t') {
                        h.diagonal = (uint16_t) (short) strtol(q, &q, 10);
                        if (*q == '\t') { h.pref_score = 0; h.diagonal = 0; }      // more than three columns: not a prefilter hit
                    }
                }
                h.pad_ = 0;
                auto ti = tKeyToIdx.find(key);
                if (ti == tKeyToIdx.end()) return die("Sequence %s is required in the prefiltering, but is not contained in the target sequence database", std::to_string(key));
                h.seq_id = ti->second;
                perQ[qi->second].push_back(h);
                while (*q != '\n' && *q != '\0') q++;
                p = (*q == '\n') ? q + 1 : q;
            }
        }
        std::vector<uint64_t> hoff(nq + 1, 0);
        for (size_t i = 0; i < nq; i++) { hits.insert(hits.end(), perQ[i].begin(), perQ[i].end()); hoff[i + 1] = hits.size(); }
        if (mk_prefilter_result_set(Q, hits.data(), hoff.data()) != MK_OK) return die("%s", mk_last_error());
        if (mk_align(T, Q, &P) != MK_OK) return die("%s", mk_last_error());
        }
        const mk_alignment *alns; const uint64_t *aoff;
        mk_align_result(Q, &alns, &aoff);
        mk::DatabaseWriter w(outPath, keyListOut ? 6 /* DBTYPE_CLUSTER_RES, Alignment.cpp:250-252 */ : mk::DBTYPE_ALIGNMENT_RES);
        e = w.open();
        if (!e.empty()) return die("%s", e);
        for (size_t i = 0; i < nq; i++) {
            buf.clear();
            for (uint64_t k = aoff[i]; k < aoff[i + 1]; k++) {
                mk_alignment al = alns[k];
                al.db_key = tkeys[al.db_key];
                if (keyListOut) { buf.append(line, (size_t) snprintf(line, sizeof(line), "%u\n", al.db_key)); continue; }   // Alignment.cpp:499-503
                buf.append(line, mk_format_alignment(line, &al));
            }
            w.write(qdb.entries[i].key, buf.data(), buf.size());
        }
        e = w.close();
        if (!e.empty()) return die("%s", e);
        fprintf(stderr, "%s: %llu alignments calculated, %llu passed, %.2f s\n", isSearch ? "search" : "align", (unsigned long long) hits.size(), (unsigned long long) aoff[nq], now() - t0);
    }
    mk_queries_destroy(Q);
    if (T) mk_targetdb_destroy(T);
    return finishShards(outBase, sh, isAlign ? (keyListOut ? 6 : mk::DBTYPE_ALIGNMENT_RES) : mk::DBTYPE_PREFILTER_RES);
}

// swapresults <i:queryDB> <i:targetDB> <i:resultDB> <o:resultDB> [-e]      M/src/util/swapresults.cpp:356-360 (doswap, alignment results)
//   queryDB / targetDB = the two sides of the search that produced resultDB (in the sliced workflow: the PROFILE DB and the fragments);
//   the output lists, per target key, the queries that hit it, e-values recomputed for the search the other way round.  Host code.
int cmdSwapResults(int argc, char **argv) {
    Args a;
    if (int rc = parse(argc, argv, a)) return rc;
    if (a.pos.size() != 4) return die("usage: metaeuk-amd swapresults <i:queryDB> <i:targetDB> <i:resultDB> <o:resultDB> [-e EVAL]%s");
    mk_params P;
    int gpu = 0;
    if (int rc = fillParams(a, P, gpu)) return rc;
    if (a.opt.find("-e") == a.opt.end()) P.evalue_thr = 0.001;              // Parameters.cpp:2414
    mk::Database qdb, tdb, rdb;
    std::string e = qdb.open(a.pos[0]);
    if (e.empty()) e = tdb.open(a.pos[1]);
    if (e.empty()) e = rdb.open(a.pos[2]);
    if (!e.empty()) return die("%s", e);
    if ((rdb.dbtype & 0xFFFF) != mk::DBTYPE_ALIGNMENT_RES) return die("swapresults: only alignment result DBs (type 5) are implemented%s");
    uint64_t aaResSize = 0;                                                  // query.sequenceReader->getAminoAcidDBSize(), swapresults.cpp:76-77
    if ((qdb.dbtype & 0xFFFF) == DBTYPE_HMM_PROFILE) aaResSize = profileDbResidues(qdb);
    else for (size_t i = 0; i < qdb.entries.size(); i++) aaResSize += qdb.seqLen(i);
    // targets by key: the output has one entry per key of the target DB (targetElementExists, :84-91,337-339)
    std::vector<size_t> tord = tdb.keyOrder();
    std::map<uint32_t, uint32_t> tKeyToIdx;
    for (size_t i = 0; i < tord.size(); i++) tKeyToIdx[tdb.entries[tord[i]].key] = (uint32_t) i;
    std::vector<mk_alignment> alns;
    std::vector<uint64_t> off(1, 0);
    std::vector<uint32_t> qkeys;
    for (size_t i = 0; i < rdb.entries.size(); i++) {
        const char *p = rdb.entry(i);
        while (*p != '\0') {                                                 // Matcher::parseAlignmentRecord (Matcher.cpp:203-239), 10 columns
            mk_alignment x;
            std::memset(&x, 0, sizeof(x));
            unsigned key = 0; char sid[64], ev[64];
            if (sscanf(p, "%u\t%d\t%63s\t%63s\t%d\t%d\t%d\t%d\t%d\t%d", &key, &x.bit_score, sid, ev, &x.q_start, &x.q_end, &x.q_len, &x.db_start, &x.db_end, &x.db_len) != 10)
                return die("Invalid alignment result record in %s", a.pos[2]);
            auto ti = tKeyToIdx.find(key);
            if (ti == tKeyToIdx.end()) return die("swapresults: key %s of the result DB is not in the target DB", std::to_string(key));
            x.db_key = ti->second;
            x.seq_id = (float) strtod(sid, nullptr); x.evalue = strtod(ev, nullptr);
            alns.push_back(x);
            while (*p != '\n' && *p != '\0') p++;
            if (*p == '\n') p++;
        }
        off.push_back(alns.size());
        qkeys.push_back(rdb.entries[i].key);
    }
    mk_swapped *S = nullptr;
    if (mk_swap_alignments(alns.data(), off.data(), (uint32_t) qkeys.size(), qkeys.data(), (uint32_t) tord.size(), aaResSize, &P, &S) != MK_OK) return die("%s", mk_last_error());
    const mk_alignment *sw; const uint64_t *soff;
    mk_swapped_result(S, &sw, &soff);
    mk::DatabaseWriter w(a.pos[3], mk::DBTYPE_ALIGNMENT_RES);
    e = w.open();
    if (!e.empty()) return die("%s", e);
    std::string buf;
    char line[512];
    for (size_t t = 0; t < tord.size(); t++) {
        buf.clear();
        for (uint64_t k = soff[t]; k < soff[t + 1]; k++) buf.append(line, mk_format_alignment(line, &sw[k]));
        w.write(tdb.entries[tord[t]].key, buf.data(), buf.size());
    }
    e = w.close();
    if (!e.empty()) return die("%s", e);
    fprintf(stderr, "swapresults: %zu result lists -> %zu target entries, %llu records\n", qkeys.size(), tord.size(), (unsigned long long) soff[tord.size()]);
    mk_swapped_destroy(S);
    return EXIT_SUCCESS;
}

// end of the batch of contigs (positions in `ord`) that starts at c0: as many as fit the per-call nucleotide budget (MK_CLI_BATCH_NT,
// default 2^29: six-frame translation of a batch stays well below the library's 2^32 query residues), at least one
size_t contigBatchEnd(const mk::Database &contigs, const std::vector<size_t> &ord, size_t c0) {
    uint64_t budget = 1ull << 29;
    if (const char *e = knobEnv("MK_CLI_BATCH_NT")) if (atoll(e) > 0) budget = (uint64_t) atoll(e);
    uint64_t nt = 0;
    size_t c1 = c0;
    while (c1 < ord.size() && (c1 == c0 || nt + contigs.seqLen(ord[c1]) <= budget)) { nt += contigs.seqLen(ord[c1]); c1++; }
    return c1;
}

// extractorfs <i:contigDB> <o:orfDB> [--min-length 15] [--translate 0|1] [--aa-sibling NAME] [--gpu N]
//   = util/extractorfs.cpp for predictexons' settings: nucleotide (or, --translate 1, amino-acid) ORF fragments under
//   renumbered keys 0..N-1 plus the header DB <o>_h ("contigKey<TAB>from(+|-)len[<TAB>complete]").  --aa-sibling NAME also
//   writes the translated fragments as the DB NAME next to <o>: predictexons.sh:47 skips `translatenucs` when aa_6f exists.
int cmdExtractOrfs(int argc, char **argv) {
    std::vector<std::string> pos;
    int minLength = 15, translate = 0, gpu = 0;
    std::string sibling;
    if (const char *lr = launcherEnv("LOCAL_RANK")) gpu = atoi(lr);
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&](int &dst) { if (i + 1 >= argc) return false; dst = atoi(argv[++i]); return true; };
        if (a == "--min-length") { if (!val(minLength)) return die("missing value for %s", a); }
        else if (a == "--translate") { if (!val(translate)) return die("missing value for %s", a); }
        else if (a == "--gpu") { if (!val(gpu)) return die("missing value for %s", a); }
        else if (a == "--aa-sibling") { if (i + 1 >= argc) return die("missing value for %s", a); sibling = argv[++i]; }
        else if (a == "--orf-start-mode" || a == "--contig-start-mode" || a == "--contig-end-mode" || a == "--max-length" || a == "--max-gaps" ||
                 a == "--forward-frames" || a == "--reverse-frames" || a == "--translation-table" || a == "--use-all-table-starts" ||
                 a == "--threads" || a == "--compressed" || a == "-v" || a == "--id-offset" || a == "--create-lookup") {
            // accepted when they carry predictexons' values (Parameters.cpp:2525-2554, PredictExons.cpp:9-11); anything else is an error
            if (i + 1 >= argc) return die("missing value for %s", a);
            const std::string v = argv[++i];
            const char *want = a == "--orf-start-mode" ? "1" : a == "--contig-start-mode" ? "2" : a == "--contig-end-mode" ? "2" : a == "--max-length" ? "32734"
                             : a == "--max-gaps" ? "2147483647" : a == "--forward-frames" ? "1,2,3" : a == "--reverse-frames" ? "1,2,3" : a == "--translation-table" ? "1"
                             : a == "--use-all-table-starts" ? "0" : a == "--compressed" ? "0" : a == "--id-offset" ? "0" : a == "--create-lookup" ? "0" : nullptr;
            if (want && v != want) return die(("this build implements " + a + " " + want + " only, got %s").c_str(), v);
        } else if (!a.empty() && a[0] == '-') return die("Unrecognized parameter %s", a);
        else pos.push_back(a);
    }
    if (pos.size() != 2) return die("usage: metaeuk-amd extractorfs <i:sequenceDB> <o:sequenceDB> [--min-length N] [--translate 0|1]%s", "");
    if (mk_init(gpu) != MK_OK) return die("%s", mk_last_error());
    mk::Database contigs;
    std::string e = contigs.open(pos[0]);
    if (!e.empty()) return die("%s", e);
    const double t0 = now();
    const std::vector<size_t> ord = contigs.keyOrder();           // fragments are renumbered by ascending contig key (extractorfs.cpp:140-155)
    static const char *COMP =                                  // Orf::iupacReverseComplementTable, Orf.cpp:48-52
        "................................................................"
        ".TVGH..CD..M.KN...YSAABW.R.......tvgh..cd..m.kn...ysaabw.r......"
        "................................................................"
        "................................................................";
    mk::DatabaseWriter seqW(pos[1], translate ? mk::DBTYPE_AMINO_ACIDS : 1 /* DBTYPE_NUCLEOTIDES */), hdrW(pos[1] + "_h", 12 /* DBTYPE_GENERIC_DB */);
    if (!(e = seqW.open()).empty() || !(e = hdrW.open()).empty()) return die("%s", e);
    std::string sibPath;
    mk::DatabaseWriter *aaW = nullptr;
    if (!sibling.empty()) {
        const size_t slash = pos[1].find_last_of('/');
        sibPath = (slash == std::string::npos ? std::string() : pos[1].substr(0, slash + 1)) + sibling;
        aaW = new mk::DatabaseWriter(sibPath, mk::DBTYPE_AMINO_ACIDS);
        if (!(e = aaW->open()).empty()) return die("%s", e);
    }
    std::string buf;
    char hdr[128];
    uint64_t n = 0;                                             // fragments so far = key of the next one
    // the contigs go through the device in batches bounded by their nucleotides (the library takes < 2^30 per call); the fragment
    // keys run on from batch to batch
    std::vector<char> nucl;
    std::vector<uint64_t> off;
    for (size_t c0 = 0; c0 < ord.size(); ) {
        const size_t c1 = contigBatchEnd(contigs, ord, c0);
        nucl.clear(); off.assign(1, 0);
        for (size_t i = c0; i < c1; i++) {
            nucl.insert(nucl.end(), contigs.entry(ord[i]), contigs.entry(ord[i]) + contigs.seqLen(ord[i]));
            off.push_back(nucl.size());
        }
        mk_orfs *O = nullptr;
        static const char none = 0;
        if (mk_extract_orfs(nucl.empty() ? &none : nucl.data(), off.data(), (uint32_t) (c1 - c0), minLength, &O) != MK_OK) return die("%s", mk_last_error());
        const mk_orf *orfs; const uint64_t *aaOff; const char *aa; uint64_t nb = 0;
        mk_orfs_result(O, &orfs, &aaOff, &aa, &nb);
        for (uint64_t k = 0; k < nb; k++) {
            const mk_orf &o = orfs[k];
            const size_t ci = ord[c0 + o.contig];
            mk_orf keyed = o;
            keyed.contig = contigs.entries[ci].key;                       // the header names the contig's DB key
            size_t hl = mk_format_orf_header(hdr, &keyed);
            hdr[hl++] = '\n';
            hdrW.write((uint32_t) (n + k), hdr, hl);
            buf.assign(aa + aaOff[k], aa + aaOff[k + 1]);
            buf.push_back('\n');
            if (aaW) aaW->write((uint32_t) (n + k), buf.data(), buf.size());
            if (!translate) {                                        // the fragment's nucleotides as Orf::getSequence hands them out
                const char *c = contigs.entry(ci);
                const size_t nn = 3 * (size_t) (aaOff[k + 1] - aaOff[k]);
                buf.resize(nn);
                for (size_t j = 0; j < nn; j++) {
                    char ch;
                    if (!o.minus_strand) { ch = c[o.from + j]; if (ch == 'u') ch = 't'; }
                    else { ch = c[o.from - j]; if (ch == 'u') ch = 't'; ch = COMP[(unsigned char) ch]; if (ch == '.') ch = 'N'; }
                    buf[j] = ch;
                }
                buf.push_back('\n');
            }
            seqW.write((uint32_t) (n + k), buf.data(), buf.size());
        }
        n += nb;
        mk_orfs_destroy(O);
        c0 = c1;
    }
    if (!(e = seqW.close()).empty() || !(e = hdrW.close()).empty()) return die("%s", e);
    if (aaW) { if (!(e = aaW->close()).empty()) return die("%s", e); delete aaW; }
    fprintf(stderr, "extractorfs: %zu contigs -> %llu fragments, %.2f s\n", contigs.entries.size(), (unsigned long long) n, now() - t0);
    return EXIT_SUCCESS;
}

// The inverted search of searchslicedtargetprofile.sh on handles: the profiles of pdb (by key, in slices of at most MK_CLI_PROFILE_COLS
// columns, default 2^24) as queries against the fragment side F (built with profile_search = 1; P already carries the inverted e-value
// threshold and --max-seqs), then swapresults -e DBL_MAX (Search.cpp:378-381): *S lists, per fragment index, the profiles that hit it.
// The profiles against the indexed fragments, swapped.  Sharded (sh.world > 1): the workers split the profiles by columns (contiguous ranges in
// key order, DBReader::decomposeDomainByAminoAcid's rule), every worker holds the whole fragment index; a worker's accepted alignments travel
// to worker 0 as they are in memory (<out>_<r>.alnbin: full-precision e-values -- the order of a swapped list is decided by them, a text
// round trip would not do), worker 0 strings the shares together in key order and swaps once: from there on the unsharded computation.
int invertedProfileSearch(const mk::Database &pdb, mk_targetdb *F, uint64_t nFrag, const mk_params &P, mk_swapped **S, uint64_t &nHits, uint64_t &nAln,
                          const Shard &sh, const std::string &outBase) {
    const size_t nProfAll = pdb.entries.size();
    std::vector<size_t> pord = pdb.keyOrder();
    size_t first = 0, count = nProfAll;
    if (sh.world > 1) {
        std::vector<mk::DbEntry> inKeyOrder(nProfAll);
        for (size_t i = 0; i < nProfAll; i++) inKeyOrder[i] = pdb.entries[pord[i]];
        mk::decomposeByLength(inKeyOrder, sh.rank, sh.world, first, count);
    }
    const uint64_t sliceCols = knobEnv("MK_CLI_PROFILE_COLS") ? strtoull(knobEnv("MK_CLI_PROFILE_COLS"), nullptr, 10) : (1ull << 24);
    std::vector<mk_alignment> alnAll;
    std::vector<uint64_t> alnOff(1, 0);
    std::vector<uint32_t> pkeys;
    nHits = 0;
    *S = nullptr;
    for (size_t p0 = first; p0 < first + count; ) {
        size_t p1 = p0;
        uint64_t cols = 0;
        std::vector<uint64_t> coff(1, 0);
        while (p1 < first + count && (p1 == p0 || cols + (std::max<uint64_t>(pdb.entries[pord[p1]].length, 1) - 1) / 25 <= sliceCols)) {
            cols += (std::max<uint64_t>(pdb.entries[pord[p1]].length, 1) - 1) / 25;
            coff.push_back(cols);
            p1++;
        }
        std::vector<uint8_t> colBytes(cols * 25 + 1);
        for (size_t i = p0; i < p1; i++) std::memcpy(colBytes.data() + coff[i - p0] * 25, pdb.entry(pord[i]), (size_t) (coff[i - p0 + 1] - coff[i - p0]) * 25);
        mk_queries *Q = nullptr;
        if (mk_profiles_create(colBytes.data(), coff.data(), (uint32_t) (p1 - p0), &P, &Q) != MK_OK) return die("%s", mk_last_error());
        if (mk_search(F, Q, &P) != MK_OK) return die("%s", mk_last_error());
        const mk_hit *hp; const uint64_t *ho;
        mk_prefilter_result(Q, &hp, &ho);
        nHits += ho[p1 - p0];
        const mk_alignment *alns; const uint64_t *aoff;
        mk_align_result(Q, &alns, &aoff);
        const uint64_t before = alnOff.back();
        alnAll.insert(alnAll.end(), alns, alns + aoff[p1 - p0]);
        for (size_t i = p0; i < p1; i++) { alnOff.push_back(before + aoff[i - p0 + 1]); pkeys.push_back(pdb.entries[pord[i]].key); }
        mk_queries_destroy(Q);
        p0 = p1;
    }
    if (sh.world > 1) {
        const std::string mine = outBase + "_" + std::to_string(sh.rank) + ".alnbin";
        if (sh.rank != 0) {
            FILE *f = fopen((mine + ".tmp").c_str(), "wb");
            if (!f) return die("cannot write %s", mine);
            const uint64_t head[3] = {(uint64_t) count, alnOff.back(), nHits};
            bool ok = fwrite(head, 8, 3, f) == 3 && fwrite(alnOff.data(), 8, alnOff.size(), f) == alnOff.size() &&
                      (pkeys.empty() || fwrite(pkeys.data(), 4, pkeys.size(), f) == pkeys.size()) &&
                      (alnAll.empty() || fwrite(alnAll.data(), sizeof(mk_alignment), alnAll.size(), f) == alnAll.size());
            ok = fclose(f) == 0 && ok;
            if (!ok || rename((mine + ".tmp").c_str(), mine.c_str()) != 0) return die("cannot write %s", mine);
            nAln = alnOff.back();
            return 0;
        }
        for (int r = 1; r < sh.world; r++) {
            const std::string theirs = outBase + "_" + std::to_string(r) + ".alnbin";
            if (!waitForPeer(outBase, r, theirs, sh.token)) return EXIT_FAILURE;
            FILE *f = fopen(theirs.c_str(), "rb");
            uint64_t head[3] = {0, 0, 0};
            if (!f || fread(head, 8, 3, f) != 3) return die("cannot read %s", theirs);
            std::vector<uint64_t> off(head[0] + 1);
            std::vector<uint32_t> keys(head[0]);
            std::vector<mk_alignment> al(head[1]);
            bool ok = fread(off.data(), 8, off.size(), f) == off.size() && (keys.empty() || fread(keys.data(), 4, keys.size(), f) == keys.size()) &&
                      (al.empty() || fread(al.data(), sizeof(mk_alignment), al.size(), f) == al.size());
            fclose(f);
            if (!ok || off.back() != head[1]) return die("%s is truncated", theirs);
            const uint64_t before = alnOff.back();
            for (uint64_t i = 0; i < head[0]; i++) alnOff.push_back(before + off[i + 1]);
            pkeys.insert(pkeys.end(), keys.begin(), keys.end());
            alnAll.insert(alnAll.end(), al.begin(), al.end());
            nHits += head[2];
            remove(theirs.c_str());
        }
        if (pkeys.size() != nProfAll) return die("the workers' profile ranges do not add up%s");
    }
    nAln = alnOff.back();
    // the e-values of the swapped lists use the profile DB's column count (swapresults.cpp:76-77: the original target side)
    mk_params SP = P;
    SP.evalue_thr = std::numeric_limits<double>::max();
    if (mk_swap_alignments(alnAll.data(), alnOff.data(), (uint32_t) nProfAll, pkeys.data(), (uint32_t) nFrag, profileDbResidues(pdb), &SP, S) != MK_OK) return die("%s", mk_last_error());
    return 0;
}

// predictexons against a PROFILE database (BASELINE config 4): Search.cpp:357-399 + searchslicedtargetprofile.sh in one
// process.  The fragments of ALL contigs become the indexed target side (the reference indexes the whole aa_6f DB as well: its numbers --
// fragments, residues -- enter the e-value threshold, the e-values and --max-seqs), the profiles go through prefilter + align in slices
// bounded by their columns, swapresults turns the lists round, the exon stage runs on them.  The contigs are translated in
// nucleotide-bounded batches; their fragments are collected on the host because they form ONE target side.  The workflow's second `align` (the merged key
// lists again, to "keep the top hits") accepts exactly the pairs of the first with --max-accept / --max-rejected at their defaults, so one
// pass is the result.  Sharded launches split the PROFILES: every worker translates all contigs and indexes all fragments (their number
// enters the e-value threshold, the e-values and --max-seqs), worker 0 gathers the alignments, swaps and predicts (invertedProfileSearch).
int predictExonsProfileTargets(const Args &a, mk_params P, const mk_exon_params &X, int minLength, const mk::Database &contigs, const std::vector<size_t> &ord,
                               const Shard &sh, const std::string &outPath, double t0) {
    static const char none = 0;
    mk::Database pdb;
    std::string e = pdb.open(a.pos[1]);
    if (!e.empty()) return die("%s", e);
    // the fragments of all contigs (ORF extraction in nucleotide-bounded batches, like the sequence path): records, offsets, residue codes
    std::vector<mk_orf> orfAll;
    std::vector<uint64_t> aaOffAll(1, 0);
    std::vector<uint8_t> fres;
    {
        std::vector<char> nucl;
        std::vector<uint64_t> off;
        for (size_t c0 = 0; c0 < ord.size(); ) {
            const size_t c1 = contigBatchEnd(contigs, ord, c0);
            nucl.clear(); off.assign(1, 0);
            for (size_t i = c0; i < c1; i++) {
                nucl.insert(nucl.end(), contigs.entry(ord[i]), contigs.entry(ord[i]) + contigs.seqLen(ord[i]));
                off.push_back(nucl.size());
            }
            mk_orfs *O = nullptr;
            if (mk_extract_orfs(nucl.empty() ? &none : nucl.data(), off.data(), (uint32_t) (c1 - c0), minLength, &O) != MK_OK) return die("%s", mk_last_error());
            const mk_orf *ob; const uint64_t *ao; const char *aab; uint64_t nb = 0;
            mk_orfs_result(O, &ob, &ao, &aab, &nb);
            const uint64_t base = aaOffAll.back();
            fres.resize(base + ao[nb] + 1);
            mk_encode(aab, ao[nb], fres.data() + base);
            for (uint64_t k = 0; k < nb; k++) {
                mk_orf o = ob[k];
                o.contig += (uint32_t) c0;                                    // position in `ord`, over all batches
                orfAll.push_back(o);
                aaOffAll.push_back(base + ao[k + 1]);
            }
            mk_orfs_destroy(O);
            c0 = c1;
        }
        if (fres.empty()) fres.push_back(0);
    }
    const mk_orf *orfs = orfAll.data();
    const uint64_t *aaOff = aaOffAll.data();
    const uint64_t nFrag = orfAll.size();
    const size_t nProf = pdb.entries.size();
    const double evalThrUser = P.evalue_thr;
    P.profile_search = 1;
    P.max_seqs = (int) std::max<uint64_t>(300, nFrag);                       // Search.cpp:372
    P.evalue_thr = nProf ? invertedEvalue(evalThrUser, (size_t) nFrag, nProf) : evalThrUser;
    mk_targetdb *F = nullptr;
    if (mk_targetdb_create(fres.data(), aaOff, (uint32_t) nFrag, &P, &F) != MK_OK) return die("%s", mk_last_error());
    const double t1 = now();
    uint64_t nHits = 0, nAlnTotal = 0;
    mk_swapped *S = nullptr;
    if (int rc = invertedProfileSearch(pdb, F, nFrag, P, &S, nHits, nAlnTotal, sh, a.pos[2])) return rc;
    if (!S) { mk_targetdb_destroy(F); return EXIT_SUCCESS; }                    // a worker whose alignments went to worker 0: it is done
    const double t2 = now();
    const uint64_t profRes = profileDbResidues(pdb);
    const mk_alignment *sw; const uint64_t *soff;
    mk_swapped_result(S, &sw, &soff);
    mk_predictions *R = nullptr;
    if (mk_predict_exons_arrays(orfs, nFrag, (uint32_t) ord.size(), sw, soff, profRes, &X, nullptr, &R) != MK_OK) return die("%s", mk_last_error());
    const mk_prediction *preds; const uint64_t *coffs; const mk_exon *exons; uint64_t np = 0;
    mk_predictions_result(R, &preds, &coffs, &exons, &np);
    mk::DatabaseWriter w(outPath, 12 /* DBTYPE_GENERIC_DB, collectoptimalset.cpp:244 */);
    e = w.open();
    if (!e.empty()) return die("%s", e);
    std::string buf;
    char line[512];
    for (size_t c = 0; c < ord.size(); c++) {
        buf.clear();
        for (uint64_t k = coffs[c]; k < coffs[c + 1]; k++)
            for (uint64_t x = preds[k].first_exon; x < preds[k].first_exon + preds[k].n_exons; x++) buf.append(line, mk_format_prediction_exon(line, &preds[k], &exons[x]));
        w.write(contigs.entries[ord[c]].key, buf.data(), buf.size());
    }
    e = w.close();
    if (!e.empty()) return die("%s", e);
    fprintf(stderr, "predictexons (profile targets): %zu contigs -> %llu fragments x %zu profiles -> %llu prefilter hits, %llu alignments -> %llu predictions; "
            "%.2f s (fragment index %.2f s, search %.2f s, swap + exon sets %.2f s)\n", ord.size(), (unsigned long long) nFrag, nProf, (unsigned long long) nHits,
            (unsigned long long) nAlnTotal, (unsigned long long) np, now() - t0, t1 - t0, t2 - t1, now() - t2);
    mk_predictions_destroy(R);
    mk_swapped_destroy(S);
    mk_targetdb_destroy(F);
    for (int r = 0; r < sh.world && sh.world > 1; r++) remove((a.pos[2] + "_" + std::to_string(r) + ".token").c_str());
    return EXIT_SUCCESS;
}

// predictexons <i:contigsDB> <i:targetsDB> <o:calledExonsDB> <tmpDir> [flags]   src/workflow/PredictExons.cpp:18-57, data/predictexons.sh
//   the whole workflow in one process: extractorfs + translatenucs + search (prefilter, align) + resultspercontig +
//   collectoptimalset, with nothing written between the stages.  Output = the dp_predictions DB the script moves to <o> (:96): one
//   record per contig (key = contig key, empty when nothing was predicted), one line per exon.  tmpDir is accepted and not used.
//   Under a multi-process launcher the contigs are split over the workers like the queries of prefilter / align.
int cmdPredictExons(int argc, char **argv) {
    Args a;
    if (int rc = parse(argc, argv, a)) return rc;
    if (a.pos.size() != 4) return die("usage: metaeuk-amd predictexons <i:contigsDB> <i:targetsDB> <o:calledExonsDB> <tmpDir> [options]%s");
    mk_params P;
    int gpu = 0;
    if (int rc = fillParams(a, P, gpu)) return rc;
    auto get = [&](const char *k) -> const std::string * { auto it = a.opt.find(k); return it == a.opt.end() ? nullptr : &it->second; };
    mk_exon_params X;
    mk_default_exon_params(&X);
    int minLength = 15;
    if (auto v = get("--min-length")) minLength = atoi(v->c_str());
    if (auto v = get("--metaeuk-eval")) X.evalue_thr = (double) (float) atof(v->c_str());      // float parameters (LocalParameters.h:76-77)
    if (auto v = get("--metaeuk-tcov")) X.target_cov_thr = (double) (float) atof(v->c_str());
    if (auto v = get("--max-intron")) X.max_intron = strtoull(v->c_str(), nullptr, 10);
    if (auto v = get("--min-intron")) X.min_intron = strtoull(v->c_str(), nullptr, 10);
    if (auto v = get("--min-exon-aa")) X.min_exon_aa = strtoull(v->c_str(), nullptr, 10);
    if (auto v = get("--max-overlap")) X.max_aa_overlap = strtoull(v->c_str(), nullptr, 10);
    if (auto v = get("--max-exon-sets")) X.max_exon_sets = strtoull(v->c_str(), nullptr, 10);
    if (auto v = get("--set-gap-open")) X.gap_open = atoi(v->c_str());
    if (auto v = get("--set-gap-extend")) X.gap_extend = atoi(v->c_str());
    if (!get("-e")) P.evalue_thr = 100;                          // setPredictExonsDefaults (PredictExons.cpp:8-16)
    if (!get("--min-aln-len")) P.min_aln_len = (int) X.min_exon_aa;   // par.alnLenThr = par.minExonAaLength (:46)
    if (mk::Database::exists(a.pos[2] + ".dbtype")) return die("%s exists already!", a.pos[2]);          // predictexons.sh:32
    const double t0 = now();
    mk::Database contigs;
    std::string e = contigs.open(a.pos[0]);
    if (!e.empty()) return die("%s", e);
    if (mk_init(gpu) != MK_OK) return die("%s", mk_last_error());
    const double tInit = now();
    // contigs by ascending key: the order in which createRenumberedDB numbers their fragments (extractorfs.cpp:140-155)
    std::vector<size_t> ord = contigs.keyOrder();
    Shard sh;
    if (int rc = shardOf(a, sh, argc, argv)) return rc;
    beginShard(a.pos[2], sh);
    {   // a PROFILE target database: the inverted search of searchslicedtargetprofile.sh (PredictExons.cpp:22-26 forces it).  Its shards are
        // the profiles, not the contigs: every worker sees all contigs
        FILE *f = fopen((a.pos[1] + ".dbtype").c_str(), "rb");
        int32_t t = -1;
        if (f) { if (fread(&t, 4, 1, f) != 1) t = -1; fclose(f); }
        if (t >= 0 && (t & 0xFFFF) == DBTYPE_HMM_PROFILE) return predictExonsProfileTargets(a, P, X, minLength, contigs, ord, sh, a.pos[2], t0);
        if (get("--exhaustive-search") && *get("--exhaustive-search") != "0") return die("--exhaustive-search 1 with a sequence target database is not implemented%s");
    }
    if (sh.world > 1) {                                           // this worker's contigs (a contiguous range of the key order)
        std::vector<mk::DbEntry> inOrder(ord.size());
        for (size_t i = 0; i < ord.size(); i++) inOrder[i] = contigs.entries[ord[i]];
        size_t first = 0, count = 0;
        mk::decomposeByLength(inOrder, sh.rank, sh.world, first, count);
        ord = std::vector<size_t>(ord.begin() + (std::ptrdiff_t) first, ord.begin() + (std::ptrdiff_t) (first + count));
    }
    const std::string outPath = sh.world > 1 ? a.pos[2] + "_" + std::to_string(sh.rank) : a.pos[2];
    static const char none = 0;
    // The contigs go through the chain in batches bounded by their nucleotides (a metagenome assembly does not fit one library call).  The batch
    // follows the input: an eighth of the nucleotides, between 2^23 and 2^27 (MK_CLI_BATCH_NT fixes it) -- ~6 batches for 10 000 contigs of 5 kb, so
    // that the first search starts after a sixth of the reading and the last exon stage is a sixth of the work
    uint64_t budget = 0;
    if (const char *eb = knobEnv("MK_CLI_BATCH_NT")) { if (atoll(eb) > 0) budget = (uint64_t) atoll(eb); }
    const bool budgetFixed = budget != 0;
    if (budget == 0) {
        uint64_t totalNt = 0;
        for (size_t i = 0; i < ord.size(); i++) totalNt += contigs.seqLen(ord[i]);
        budget = std::min<uint64_t>(1ull << 27, std::max<uint64_t>(1ull << 23, totalNt / 8));
    }
    auto batchEnd = [&](size_t c0) {
        uint64_t nt = 0;
        size_t c1 = c0;
        while (c1 < ord.size() && (c1 == c0 || nt + contigs.seqLen(ord[c1]) <= budget)) { nt += contigs.seqLen(ord[c1]); c1++; }
        return c1;
    };
    // Round 6: the FIRST batch of contigs is read, scanned and translated by a helper thread WHILE this thread masks and indexes the target side
    // (mk_extract_orfs / mk_queries_from_orfs work on a stream of their own; neither needs the target database) -- the reference's workflow runs
    // extractorfs in front of everything (predictexons.sh:42-60); here it hides behind the index build.  Not with several workers (the counting pass
    // below walks all batches first) and not when target splits may change the batch size.  MK_CLI_WARM=0 switches it off.
    struct Prefetched { mk_orfs *O = nullptr; mk_queries *Q = nullptr; size_t c1 = 0; uint64_t nb = 0; int rc = 0; std::string err; double seconds = 0; bool valid = false; } pre;
    std::thread preThread;
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{preThread};
    {
        const bool splitAsked = (get("--split") && atoi(get("--split")->c_str()) > 1) || get("--split-memory-limit");
        const char *warm = knobEnv("MK_CLI_WARM");
        if (sh.world == 1 && !splitAsked && !ord.empty() && !(warm && !strcmp(warm, "0"))) {
            pre.c1 = batchEnd(0);
            pre.valid = true;
            preThread = std::thread([&]() {
                const double ta = now();
                std::vector<char> nuclP;
                std::vector<uint64_t> offP(1, 0);
                for (size_t i = 0; i < pre.c1; i++) {
                    nuclP.insert(nuclP.end(), contigs.entry(ord[i]), contigs.entry(ord[i]) + contigs.seqLen(ord[i]));
                    offP.push_back(nuclP.size());
                }
                if (mk_extract_orfs(nuclP.empty() ? &none : nuclP.data(), offP.data(), (uint32_t) pre.c1, minLength, &pre.O) != MK_OK) { pre.rc = EXIT_FAILURE; pre.err = mk_last_error(); return; }
                const mk_orf *orfs; const uint64_t *aaOff; const char *aa;
                mk_orfs_result(pre.O, &orfs, &aaOff, &aa, &pre.nb);
                if (mk_queries_from_orfs(pre.O, &P, &pre.Q) != MK_OK) { pre.rc = EXIT_FAILURE; pre.err = mk_last_error(); }
                pre.seconds = now() - ta;
            });
        }
    }
    // the target side; a sequence DB that needs (or is told to use) target splits keeps only its residues resident: every contig batch then
    // runs the prefilter split by split and aligns against the whole database (splitPrefilter)
    TargetSide ts;
    SplitPlan plan;
    plan.kmerSize = P.kmer_size; plan.maxSeqs = P.max_seqs;
    mk::Database tdbSeq;
    {
        FILE *f = fopen((a.pos[1] + ".dbtype").c_str(), "rb");
        int32_t t = -1;
        if (f) { if (fread(&t, 4, 1, f) != 1) t = -1; fclose(f); }
        const bool explicitSplit = get("--split") && atoi(get("--split")->c_str()) > 1 && get("--split-mode") && *get("--split-mode") == "0";
        const bool hasIdx = (t >= 0 && (t & 0xFFFF) == 9) || (!launcherEnv("MMSEQS_IGNORE_INDEX") && mk::Database::exists(a.pos[1] + ".idx.dbtype"));
        if (explicitSplit && t >= 0 && (t & 0xFFFF) == 9) return die("--split with --split-mode 0 needs an amino-acid sequence DB as the target (index DBs hold one split)%s");
        if (t >= 0 && (t & 0xFFFF) == mk::DBTYPE_AMINO_ACIDS && (explicitSplit || !hasIdx)) {
            e = tdbSeq.open(a.pos[1]);
            if (!e.empty()) return die("%s", e);
            uint64_t aaSize = 0, nucl = 0;
            for (size_t i = 0; i < tdbSeq.entries.size(); i++) aaSize += tdbSeq.seqLen(i);
            for (size_t i = 0; i < contigs.entries.size(); i++) nucl += contigs.seqLen(i);
            // (the fragment DB the reference would have open is not written here: ~1 fragment per 25 nucleotides stands in for its size)
            if (int rc = planTargetSplits(a, P, tdbSeq.entries.size(), aaSize, (size_t) (nucl / 25), plan)) return rc;
            if (int rc = plan.splits > 1 ? openTargetSequences(tdbSeq, P, ts) : openTargetDb(tdbSeq, P, ts)) return rc;
        } else if (int rc = openTarget(a.pos[1], P, ts)) return rc;
    }
    mk_targetdb *T = ts.T;
    const std::vector<uint32_t> &tkeys = ts.keys;
    const double t1 = now();
    if (preThread.joinable()) preThread.join();
    if (pre.rc) return die("%s", pre.err);
    if (plan.splits > 1) {                                         // target splits index every split again per batch: as few batches as possible
        if (!budgetFixed) budget = 1ull << 29;
        if (pre.valid) { mk_queries_destroy(pre.Q); mk_orfs_destroy(pre.O); pre.valid = false; }
    }
    const double t1b = now();
    std::vector<char> nucl;
    std::vector<uint64_t> off;
    auto loadBatch = [&](size_t c0, size_t c1) {
        nucl.clear(); off.assign(1, 0);
        for (size_t i = c0; i < c1; i++) {
            nucl.insert(nucl.end(), contigs.entry(ord[i]), contigs.entry(ord[i]) + contigs.seqLen(ord[i]));
            off.push_back(nucl.size());
        }
    };
    // fragment keys are numbered over ALL contigs (createRenumberedDB): a worker publishes how many fragments its contigs have (a counting
    // pass of the ORF kernels over its batches) and adds what the workers before it found -- a prefix over the workers through the file system
    uint64_t orfBase = 0;
    if (sh.world > 1) {
        uint64_t mine = 0;
        for (size_t c0 = 0; c0 < ord.size(); ) {
            const size_t c1 = contigBatchEnd(contigs, ord, c0);
            loadBatch(c0, c1);
            mk_orfs *O = nullptr;
            if (mk_extract_orfs(nucl.empty() ? &none : nucl.data(), off.data(), (uint32_t) (c1 - c0), minLength, &O) != MK_OK) return die("%s", mk_last_error());
            const mk_orf *orfs; const uint64_t *aaOff; const char *aa; uint64_t nb = 0;
            mk_orfs_result(O, &orfs, &aaOff, &aa, &nb);
            mine += nb;
            mk_orfs_destroy(O);
            c0 = c1;
        }
        FILE *f = fopen((outPath + ".orfs.tmp").c_str(), "w");
        if (!f) return die("cannot write %s", outPath + ".orfs.tmp");
        fprintf(f, "%llu\n", (unsigned long long) mine);
        fclose(f);
        rename((outPath + ".orfs.tmp").c_str(), (outPath + ".orfs").c_str());
        for (int r = 0; r < sh.rank; r++) {
            const std::string cf = a.pos[2] + "_" + std::to_string(r) + ".orfs";
            if (!waitForPeer(a.pos[2], r, cf, sh.token)) return EXIT_FAILURE;
            unsigned long long v = 0;
            FILE *g = fopen(cf.c_str(), "r");
            if (!g || fscanf(g, "%llu", &v) != 1) return die("cannot read %s", cf);
            fclose(g);
            orfBase += v;
        }
    }
    mk::DatabaseWriter w(outPath, 12 /* DBTYPE_GENERIC_DB, collectoptimalset.cpp:244 */);
    e = w.open();
    if (!e.empty()) return die("%s", e);
    std::string buf;
    char line[512];
    uint64_t nOrfs = 0, np = 0;
    // The contigs go through the chain in batches bounded by their nucleotides (a metagenome assembly does not fit one library call); the target
    // index stays resident, the fragment keys run on from batch to batch.  Round 6: the batches are QUEUED in the library's search engine
    // (mk_search_begin / mk_search_wait) -- the reference's `search` is one OpenMP loop over all fragments (Prefiltering.cpp:817-886,
    // Alignment.cpp:312-514), a caller's batches must not put a seam into it: while batch k is searched, the contigs of batch k + 1 are read,
    // scanned and translated (mk_extract_orfs / mk_queries_from_orfs run on a stream of their own and do not wait for the engine), and the exon
    // sets of batch k - 1 are chained and written by this thread.  The batch follows the input: an eighth of the nucleotides, between 2^23 and
    // 2^27 (MK_CLI_BATCH_NT fixes it) -- ~6 batches for 10 000 contigs of 5 kb, so that the first search starts after a sixth of the reading and
    // the last exon stage is a sixth of the work.  The results do not depend on the batches (tests/test_gpu_parity.py::test_cli_contig_batches).
    struct Batch { size_t c0, c1; mk_orfs *O; mk_queries *Q; uint64_t nb, orfFirst; bool begun; double tBegun; };
    std::deque<Batch> inflight;
    double tRead = 0, tExtract = 0, tSearchWait = 0, tExons = 0, tWrite = 0;
    size_t nBatches = 0;
    // collect batch b: wait for its search (if it was queued), chain its exon sets, write them
    auto finish = [&](Batch &b) -> int {
        double ta = now();
        if (b.begun && mk_search_wait(b.Q) != MK_OK) return die("%s", mk_last_error());
        double tb = now();
        tSearchWait += tb - ta;
        if (knobEnv("MK_CLI_TIMELINE")) fprintf(stderr, "[predictexons] batch of %llu fragments: begun at %.3f s, collected at %.3f s (blocked %.3f s)\n", (unsigned long long) b.nb, b.tBegun - t0, tb - t0, tb - ta);
        mk_predictions *R = nullptr;
        if (mk_predict_exons(T, b.O, b.Q, &X, tkeys.data(), &R) != MK_OK) return die("%s", mk_last_error());
        const mk_prediction *preds; const uint64_t *coff; const mk_exon *exons; uint64_t npb = 0;
        mk_predictions_result(R, &preds, &coff, &exons, &npb);
        ta = now();
        tExons += ta - tb;
        for (size_t c = b.c0; c < b.c1; c++) {
            buf.clear();
            for (uint64_t k = coff[c - b.c0]; k < coff[c - b.c0 + 1]; k++)
                for (uint64_t x = preds[k].first_exon; x < preds[k].first_exon + preds[k].n_exons; x++) {
                    mk_exon ex = exons[x];
                    ex.orf += (uint32_t) (orfBase + b.orfFirst);
                    buf.append(line, mk_format_prediction_exon(line, &preds[k], &ex));
                }
            w.write(contigs.entries[ord[c]].key, buf.data(), buf.size());
        }
        np += npb;
        mk_predictions_destroy(R);
        mk_queries_destroy(b.Q);
        mk_orfs_destroy(b.O);
        tWrite += now() - ta;
        return 0;
    };
    for (size_t c0 = 0; c0 < ord.size() || (c0 == 0 && ord.empty()); ) {
        const size_t c1 = (c0 == 0 && pre.valid) ? pre.c1 : batchEnd(c0);
        double ta = now();
        Batch b{c0, c1, nullptr, nullptr, 0, nOrfs, false, 0.0};
        if (c0 == 0 && pre.valid) {                                  // (read, scanned and translated beside the target index)
            b.O = pre.O; b.Q = pre.Q; b.nb = pre.nb;
            pre.valid = false;
        } else {
            loadBatch(c0, c1);
            double tb = now();
            tRead += tb - ta;
            if (mk_extract_orfs(nucl.empty() ? &none : nucl.data(), off.data(), (uint32_t) (c1 - c0), minLength, &b.O) != MK_OK) return die("%s", mk_last_error());
            const mk_orf *orfs; const uint64_t *aaOff; const char *aa;
            mk_orfs_result(b.O, &orfs, &aaOff, &aa, &b.nb);
            if (mk_queries_from_orfs(b.O, &P, &b.Q) != MK_OK) return die("%s", mk_last_error());
            tExtract += now() - tb;
        }
        nOrfs += b.nb;
        nBatches++;
        if (plan.splits > 1) {                                       // (blocking: every split is indexed and searched in turn, then the whole batch is aligned)
            ta = now();
            uint64_t total = 0;
            if (int rc = splitPrefilter(b.Q, (size_t) b.nb, tdbSeq, P, plan, total)) return rc;
            if (mk_align(T, b.Q, &P) != MK_OK) return die("%s", mk_last_error());
            tSearchWait += now() - ta;
        } else {
            if (mk_search_begin(T, b.Q, &P) != MK_OK) return die("%s", mk_last_error());
            b.begun = true;
            b.tBegun = now();
        }
        inflight.push_back(b);
        // two batches stay queued while the oldest is collected (MK_CLI_QUEUE_DEPTH): the engine's prefilter thread finishes a batch well before its
        // alignments are complete, and with one batch behind it it idled until this thread had collected, chained and written the previous one and
        // translated the next (the timeline of 10 000 contigs: batch k + 1 begun 50 ms before batch k was complete)
        static const size_t depth = (size_t) std::max(1L, knobEnv("MK_CLI_QUEUE_DEPTH") ? atol(knobEnv("MK_CLI_QUEUE_DEPTH")) : 2L);
        while (inflight.size() > depth) {                            // batches k and k - 1 are queued: collect batch k - 2 beside them
            if (int rc = finish(inflight.front())) return rc;
            inflight.pop_front();
        }
        if (c1 == c0) break;                                       // (no contigs at all)
        c0 = c1;
    }
    while (!inflight.empty()) {
        if (int rc = finish(inflight.front())) return rc;
        inflight.pop_front();
    }
    const double t2 = now();
    e = w.close();
    if (!e.empty()) return die("%s", e);
    if (plan.splits > 1) fprintf(stderr, "predictexons: %d target splits%s (--max-seqs %d per split, k = %d)\n", plan.splits, plan.chosen ? " [chosen from the memory limit]" : "", plan.maxSeqs, plan.kmerSize);
    fprintf(stderr, "predictexons: %zu contigs -> %llu fragments x %zu targets -> %llu predictions; %.2f s (target index %.2f s, fragments to exon sets %.2f s)\n",
            ord.size(), (unsigned long long) nOrfs, tkeys.size(), (unsigned long long) np, now() - t0, t1 - t0, t2 - t1);
    if (knobEnv("MK_CLI_TIMELINE")) {                               // the library's own account: kernel event times and host phases of the whole run
        static mk_kernel_stat st[512];
        const int n = mk_kernel_stats(st, 512);
        for (int k = 0; k < n; k++) if (st[k].ms >= 2.0) fprintf(stderr, "[predictexons] %-28s %9.1f ms %7llu launches\n", st[k].name, st[k].ms, (unsigned long long) st[k].launches);
    }
    // the per-stage account of this thread (what it did while the searches ran in the library's engine; `search (blocked)` = the time it sat in
    // mk_search_wait, i.e. what the other stages did NOT hide)
    fprintf(stderr, "predictexons stages: %zu batches of <= %llu nt; open + init %.3f s, target index %.3f s, first batch beside it %.3f s, waited for it %.3f s, read %.3f s, extract + upload %.3f s, search (blocked) %.3f s, exons %.3f s, "
                    "write %.3f s, close %.3f s\n", nBatches, (unsigned long long) budget, tInit - t0, t1 - tInit, pre.seconds, t1b - t1, tRead, tExtract, tSearchWait, tExons, tWrite, now() - t2);
    mk_targetdb_destroy(T);
    if (int rc = finishShards(a.pos[2], sh, 12)) return rc;
    if (sh.world > 1 && sh.rank == 0) for (int r = 0; r < sh.world; r++) remove((a.pos[2] + "_" + std::to_string(r) + ".orfs").c_str());
    return EXIT_SUCCESS;
}

// createindex <i:sequenceDB> <tmpDir> [-s 7.5 ...]   M/src/workflow/CreateIndex.cpp:108-175 -> indexdb (util/indexdb.cpp:42-186)
//   masks the targets, builds the k-mer lists and writes <sequenceDB>.idx in the reference's index DB format (type 9).  With a GPU the
//   lists are built in HBM and streamed into the file (k = 6, and k = 7 from 3.35e9 residues on or with -k 7); without one the host
//   builder writes a k = 6 index.  The reference's `prefilter` / `search` pick the file up like one of their own; so do the commands above.
int cmdCreateIndex(int argc, char **argv) {
    Args a;
    if (int rc = parse(argc, argv, a)) return rc;
    if (a.pos.size() != 2) return die("usage: metaeuk-amd createindex <i:sequenceDB> <tmpDir> [options]%s");
    mk_params P;
    int gpu = 0;
    if (a.opt.find("-s") == a.opt.end()) a.opt["-s"] = "7.5";           // CreateIndex.cpp:114
    if (int rc = fillParams(a, P, gpu)) return rc;
    const double t0 = now();
    mk::Database db;
    const std::string e = db.open(a.pos[0]);
    if (!e.empty()) return die("%s", e);
    const std::vector<size_t> ord = db.keyOrder();                     // DBReader NOSORT over a key-sorted .index (indexdb.cpp:67-68)
    std::vector<uint32_t> keys(ord.size()), lens(ord.size());
    std::vector<uint64_t> offs(ord.size());
    for (size_t i = 0; i < ord.size(); i++) { keys[i] = db.entries[ord[i]].key; offs[i] = db.entries[ord[i]].offset; lens[i] = (uint32_t) db.entries[ord[i]].length; }
    const std::string out = a.pos[0] + ".idx";
    const bool onGpu = mk_init(gpu) == MK_OK;                          // (no device: mk_index_write builds on the host)
    if (mk_index_write(out.c_str(), db.data.data(), db.data.size(), keys.data(), offs.data(), lens.data(), (uint32_t) ord.size(), db.dbtype, &P) != MK_OK)
        return die("%s", mk_last_error());
    fprintf(stderr, "createindex: %zu sequences -> %s, %.2f s (%s)\n", ord.size(), out.c_str(), now() - t0, onGpu ? "lists built on the GPU" : "lists built on the host");
    return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "metaeuk-amd: MI355X prefilter+align modules of `metaeuk predictexons`\n  metaeuk-amd prefilter <queryDB> <targetDB> <prefilterDB> [flags]\n  metaeuk-amd align <queryDB> <targetDB> <prefilterDB> <alignmentDB> [flags]\n  metaeuk-amd search <queryDB> <targetDB> <alignmentDB> <tmpDir> [flags]\n  metaeuk-amd extractorfs <contigDB> <orfDB> [--min-length N] [--translate 0|1] [--aa-sibling NAME]\n  metaeuk-amd predictexons <contigsDB> <targetsDB> <calledExonsDB> <tmpDir> [flags]\n  metaeuk-amd createindex <targetsDB> <tmpDir> [-s 7.5]\n");
        return EXIT_FAILURE;
    }
    const std::string cmd = argv[1];
    if (cmd == "prefilter") return cmdPrefilterOrAlign(0, argc, argv);
    if (cmd == "align") return cmdPrefilterOrAlign(1, argc, argv);
    if (cmd == "search") return cmdPrefilterOrAlign(2, argc, argv);
    if (cmd == "shardinfo" && argc >= 4) {                       // <db> r/N: the worker's entry range (first, count) -- for tests and scripts
        mk::Database db;
        const std::string e = db.open(argv[2]);
        if (!e.empty()) return die("%s", e);
        int r = 0, n = 1;
        if (sscanf(argv[3], "%d/%d", &r, &n) != 2 || n < 1 || r < 0 || r >= n) return die("bad shard %s", argv[3]);
        size_t first = 0, count = 0;
        mk::decomposeByLength(db.entries, r, n, first, count);
        printf("%zu %zu\n", first, count);
        return EXIT_SUCCESS;
    }
    if (cmd == "mergeshards" && argc >= 5) {                     // <out> <n> <dbtype>: merge <out>_0 .. <out>_{n-1} by hand
        const std::string e = mk::mergeShards(argv[2], atoi(argv[3]), atoi(argv[4]));
        if (!e.empty()) return die("%s", e);
        return EXIT_SUCCESS;
    }
    if (cmd == "predictexons") return cmdPredictExons(argc, argv);
    // the other commands are not sharded: under a multi-process launcher worker 0 does the job
    if (launcherEnv("RANK") && atoi(launcherEnv("RANK")) > 0) return EXIT_SUCCESS;
    if (cmd == "extractorfs") return cmdExtractOrfs(argc, argv);
    if (cmd == "predictexons") return cmdPredictExons(argc, argv);
    if (cmd == "createindex" || cmd == "indexdb") return cmdCreateIndex(argc, argv);
    if (cmd == "swapresults") return cmdSwapResults(argc, argv);
    fprintf(stderr, "Invalid Command: %s\n", cmd.c_str());
    return EXIT_FAILURE;
}
