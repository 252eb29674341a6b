// metaeuk_amd/csrc/mk_abi.cpp -- C ABI (include/metaeuk_amd.h) over the HIP kernels.
// One process drives one GPU (mk_init); all device work goes through one HIP stream owned by the
// library.  There is NO CPU fallback for the kernels: without a usable HIP device every compute
// entry point fails with MK_ERR_DEVICE.
#include "../../include/metaeuk_amd_debug.h"
#include "mk_align.hpp"
#include "mk_host.hpp"
#include "mk_kernels.hpp"
#include "mk_orf.hpp"
#include "mk_exons.hpp"
#include "mk_indexfile.hpp"
#include "mk_index.hpp"
#include "mk_prefilter.hpp"
#include "mk_profile.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <omp.h>
#include <thread>
#include <unordered_map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MK_ERR_DEVICE, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// Sequences of 32768 residues and more take the reference's wrapped-diagonal path in the prefilter (16-bit index positions and
// diagonals, UngappedAlignment::computeLongScore); the alignment kernels number target columns in 17 bits.
constexpr uint32_t MK_MAX_SEQ_LEN = 1u << 17;
std::atomic<double> g_hitsPerQuery{0.0}, g_alnsPerQuery{0.0};   // results per query of the last finished search (sequence queries): sizes the next batch's result blocks

bool g_ready = false;
int g_device = -1;
hipStream_t g_stream = nullptr;

// ---- per-kernel timing (HIP events on the library's stream) and host-phase wall clock -------------
// The two stages of mk_search run on two host threads with one stream each: pending events are per thread, the
// accumulators are shared.
struct StatAcc { double ms = 0; uint64_t launches = 0; double bytes = 0; double cells = 0; };
std::map<std::string, StatAcc> g_stats;
std::mutex g_statsMutex;
std::vector<std::string> g_statNames;
hipStream_t g_stream2 = nullptr;                         // alignment stage of mk_search
hipStream_t g_prefStream2 = nullptr;                     // the search engine's second prefilter thread
hipStream_t g_uploadStream = nullptr;                    // mk_queries_create: upload + derivation of the NEXT batch beside a search in flight
constexpr int MAX_ALIGN_WORKERS = 4;
constexpr int PREFILTER2_LANE = 5;                       // scratch lane of the second prefilter thread (the alignment workers: 0 .. 3)
constexpr int UPLOAD_LANE = 8;                           // scratch lane of mk_queries_create
hipStream_t g_alignStreams[MAX_ALIGN_WORKERS] = {};      // [0] = g_stream2; the further workers of the alignment stage
thread_local hipStream_t t_stream = nullptr;             // stream of the calling thread's stage (null: g_stream)
hipStream_t cur_stream() { return t_stream ? t_stream : g_stream; }

struct Timed { hipEvent_t a{}, b{}; std::string name; double bytes; double cells; };
thread_local std::vector<Timed> g_pending;

int timed_begin(const char *name, double bytes, double cells) {
    Timed t; t.name = name; t.bytes = bytes; t.cells = cells;
    if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return -1;
    (void) hipEventRecord(t.a, cur_stream());
    g_pending.push_back(t);
    return (int) g_pending.size() - 1;
}
void timed_end(int h) { if (h >= 0) (void) hipEventRecord(g_pending[h].b, cur_stream()); }
void timed_set(int h, double bytes, double cells) { if (h >= 0 && h < (int) g_pending.size()) { g_pending[h].bytes = bytes; g_pending[h].cells = cells; } }
void timed_flush() {
    for (Timed &t : g_pending) {
        float ms = 0;
        (void) hipEventSynchronize(t.b);
        (void) hipEventElapsedTime(&ms, t.a, t.b);
        {
            std::lock_guard<std::mutex> g(g_statsMutex);
            StatAcc &s = g_stats[t.name];
            s.ms += ms; s.launches += 1; s.bytes += t.bytes; s.cells += t.cells;
        }
        (void) hipEventDestroy(t.a); (void) hipEventDestroy(t.b);
    }
    g_pending.clear();
}

struct HostTimer {
    std::string name; std::chrono::steady_clock::time_point t0;
    explicit HostTimer(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> g(g_statsMutex);
        g_stats[name].ms += ms;
    }
};

// Device blocks of the query batches come from a small pool: a batch of ORF fragments is created and destroyed per call of the hot
// path, and hipFree waits for the whole device -- with a search of the previous batch still in flight (mk_search_begin) that would
// serialise what the pipeline overlaps.  Blocks of more than 1 GB are not kept.
struct DevPoolBlock { void *p; size_t cap; };
std::vector<DevPoolBlock> g_devPool;
std::mutex g_devPoolMutex;
size_t g_devPoolBytes = 0;                               // idle bytes in the pool (under the mutex)
constexpr size_t DEV_POOL_BLOCKS = 24, DEV_POOL_BLOCK_MAX = 1ull << 30;
constexpr size_t DEV_POOL_BYTES_MAX = 6ull << 30;       // idle blocks are worth about two batches of the largest shape in use, never more
void dev_pool_trim();
void *dev_pool_alloc(size_t bytes, size_t *cap) {
    {
        std::lock_guard<std::mutex> g(g_devPoolMutex);
        int best = -1;
        for (int i = 0; i < (int) g_devPool.size(); i++)
            if (g_devPool[i].cap >= bytes && g_devPool[i].cap <= 2 * bytes + 65536 && (best < 0 || g_devPool[i].cap < g_devPool[best].cap)) best = i;
        if (best >= 0) {
            void *p = g_devPool[best].p; *cap = g_devPool[best].cap;
            g_devPoolBytes -= g_devPool[best].cap;
            g_devPool.erase(g_devPool.begin() + best);
            return p;
        }
    }
    void *p = nullptr;
    const size_t want = bytes + bytes / 16 + 256;
    if (hipMalloc(&p, want) != hipSuccess) {                 // the idle blocks may hold what is missing: give them back and ask once more
        (void) hipGetLastError();
        dev_pool_trim();
        p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    }
    *cap = want;
    return p;
}
void dev_pool_free(void *p, size_t cap) {
    std::vector<void *> drop;
    if (cap > DEV_POOL_BLOCK_MAX) drop.push_back(p);
    else {
        std::lock_guard<std::mutex> g(g_devPoolMutex);
        g_devPool.push_back(DevPoolBlock{p, cap});
        g_devPoolBytes += cap;
        while (g_devPool.size() > DEV_POOL_BLOCKS || (g_devPoolBytes > DEV_POOL_BYTES_MAX && g_devPool.size() > 1)) {
            // too many blocks: the smallest goes; too many bytes: the OLDEST goes (varying batch sizes leave blocks behind that no later batch fits)
            int victim = 0;
            if (g_devPool.size() > DEV_POOL_BLOCKS)
                for (int i = 1; i < (int) g_devPool.size(); i++) if (g_devPool[i].cap < g_devPool[victim].cap) victim = i;
            drop.push_back(g_devPool[victim].p);
            g_devPoolBytes -= g_devPool[victim].cap;
            g_devPool.erase(g_devPool.begin() + victim);
        }
    }
    for (void *d : drop) (void) hipFree(d);
}

// every pooled block back to the device (before a database is built -- the index of a UniRef50-scale database takes most of the HBM -- and when an
// allocation fails, here or in the scratch buffers of the stages: mk::dev_pool_release)
void dev_pool_trim() {
    std::vector<DevPoolBlock> drop;
    {
        std::lock_guard<std::mutex> g(g_devPoolMutex);
        drop.swap(g_devPool);
        g_devPoolBytes = 0;
    }
    for (DevPoolBlock &b : drop) (void) hipFree(b.p);
}

template <typename T>
struct DevBuf {
    T *p = nullptr; size_t n = 0;
    bool pooled = false; size_t poolCap = 0;                 // pooled: the block comes from / returns to the device pool
    ~DevBuf() { drop(); }
    void drop() {
        if (!p) return;
        if (pooled && poolCap) dev_pool_free(p, poolCap); else (void) hipFree(p);
        p = nullptr; poolCap = 0;
    }
    hipError_t alloc(size_t count) {
        drop();
        n = count;
        if (count == 0) return hipSuccess;
        if (pooled) {
            p = reinterpret_cast<T *>(dev_pool_alloc(count * sizeof(T), &poolCap));
            return p ? hipSuccess : hipErrorOutOfMemory;
        }
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess) {                                // the pool of the query batches may hold what is missing
            (void) hipGetLastError();
            p = nullptr;
            dev_pool_trim();
            e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        }
        return e;
    }
    hipError_t upload(const T *h, size_t count, hipStream_t stream = nullptr) {
        hipError_t e = alloc(count);
        if (e != hipSuccess || count == 0) return e;
        return hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, stream ? stream : g_stream);
    }
};

}  // namespace

namespace mk {
void dev_pool_release() { dev_pool_trim(); }
void *dev_block_alloc(size_t bytes, size_t *cap) { return dev_pool_alloc(bytes, cap); }
void dev_block_free(void *p, size_t cap) { if (p) dev_pool_free(p, cap); }
void host_stat(const char *name, double ms) { std::lock_guard<std::mutex> g(g_statsMutex); g_stats[name].ms += ms; }
hipError_t sync_wait(hipStream_t stream, const char *statName) {
    const double t0 = ScopedHost::now_ms();
    const hipError_t e = hipStreamSynchronize(stream);
    host_stat(statName, ScopedHost::now_ms() - t0);
    return e;
}
double ScopedHost::now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace mk

namespace {

// CPUs this process may actually use: min(affinity mask, cgroup-v2 cpu.max quota).  The GPU boxes expose
// 256 hardware threads behind a 16-CPU quota; an OpenMP team sized from nproc would be throttled.
int effective_cpus() {
    int n = omp_get_num_procs();
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char a[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) {
            const long quota = atol(a);
            if (quota > 0) n = (int) std::min<long>(n, std::max<long>(1, (quota + period - 1) / period));
        }
        fclose(f);
    }
    return std::max(n, 1);
}

}  // namespace

struct mk_targetdb {
    uint32_t n = 0;
    std::vector<uint64_t> off;
    std::vector<uint8_t> maskedHost;       // host copy of the masked residues: fetched from HBM when somebody asks (masked_host)
    bool maskedHostReady = false;
    uint64_t cells = 0;                    // cells of the k-mer table (20^6 or 20^7)
    uint64_t maskedResidues = 0;
    uint32_t maxList = 0;
    std::vector<uint32_t> keys;      // DB keys of the targets when the database came from an index file
    mk::SubMat kmerMat, ungMat, alnMat;
    mk::Evaluer evaluer;
    int kmerThr = 0;
    uint64_t nEntries = 0;
    uint32_t maxLen = 0;
    DevBuf<uint8_t> dRes, dMasked;
    DevBuf<uint64_t> dOff;
    DevBuf<uint64_t> dKmerSlot;      // 20^6: inline single entry, or first entry index | list length << 32 (entries < 2^32)
    DevBuf<uint32_t> dKmerBits;      // 20^6 bits: list non-empty
    DevBuf<uint64_t> dEntries;       // seqId | pos << 32
    DevBuf<int16_t> dScore3;
    DevBuf<uint16_t> dIndex3, dHist3, dCum3;
    int histLo = 0, histRange = 0;
    DevBuf<int8_t> dMatAln, dMatUng;
    DevBuf<uint32_t> dKeys;          // device copy of `keys` (last tie-break of the alignment order)
    DevBuf<uint16_t> dAddr3;         // 3-mer number -> address code of the table cells (profile k-mer lists)
    int kmerSize = 6;                // 6 or 7 (mk_params.kmer_size / IndexTable::computeKmerSize)
    uint64_t entryShift = 0;         // MK_TEST_ENTRY_BASE: the slots' list starts are shifted by this much, the entries pointer of the views back
    DevBuf<int16_t> dScore2;         // k = 7: similar 2-mers [400][400]
    DevBuf<uint16_t> dIndex2, dNum3; // ... their numbers; address code -> 3-mer number
    bool profileSearch = false;      // built for profile queries (mk_params.profile_search)
    bool unindexed = false;          // mk_targetdb_create_sequences: residues only, what mk_align needs (no masking, no k-mer index)
    std::vector<int32_t> bitScoreTable;   // static_cast<int>(bitScore(score) + 0.5), score < 32768
    // score tables of the alignment stage, kept from call to call: the e-value row of a query length depends on the database alone, and
    // consecutive batches of ORF fragments bring the same lengths (a batch of other lengths adds its rows; the device copy follows the id)
    std::unordered_map<uint32_t, std::vector<double>> evalueRows;
    std::shared_ptr<const struct ScoreTabs> scoreTabs;     // the newest snapshot (immutable: the batches in flight hold the one they were given)
};

// score tables of the alignment stage for a set of query lengths: e-value per (length, score), bit scores, the gate
struct ScoreTabs { mk::AssembleTables tables; std::vector<mk::GateEntry> gate; std::vector<uint32_t> lens; double thr = -1.0; };

struct mk_queries {
    uint32_t n = 0;
    uint32_t maxLen = 0;
    std::vector<uint64_t> off;
    std::vector<uint8_t> res;
    DevBuf<uint8_t> dRes;
    DevBuf<uint64_t> dOff;
    DevBuf<int16_t> dKmerThr;
    DevBuf<int8_t> dCorr, dBias8;
    // profile queries (mk_profiles_create): res / dRes hold the profiles' query letters, off the column offsets
    bool isProfile = false;
    DevBuf<int8_t> dProfSorted, dProfAln;   // [column][40], [column][32] (mk_profile.hpp)
    int kmerSize = 6;                // the k the k-mer thresholds were derived for (re-derived when the database uses the other one)
    mk_params derivedWith;           // the parameters of that derivation
    // stage results (the reference hands these over through the pref_0 / search_res DBs)
    mk::HostBlock hits; size_t nHits = 0; std::vector<uint64_t> hitOff; bool havePref = false;
    mk::PrefilterStats pfStats;      // run statistics of the prefilter over this batch (Prefiltering.cpp:889-904)
    mk::HostBlock alns; std::vector<uint64_t> alnOff; bool haveAln = false;
    struct SearchJob *job = nullptr; // mk_search_begin: the search in flight over this batch (mk_search_wait clears it)
    // (the profile images too -- ADVICE round 5: a hipFree of theirs in mk_queries_destroy synchronised the device under the batch in flight)
    mk_queries() { dRes.pooled = dOff.pooled = dKmerThr.pooled = dCorr.pooled = dBias8.pooled = dProfSorted.pooled = dProfAln.pooled = true; }
};

// the alignment stage's score tables for batch q (e-value per query length and score, the gate), from the database's cache.  A snapshot
// serves every batch whose query lengths it covers; a batch that brings new lengths gets a new snapshot over the union (the rows come from
// the per-length cache), and the batches still in flight keep theirs.
static std::shared_ptr<const ScoreTabs> score_tables(mk_targetdb *db, const mk_queries *q, double evalThr) {
    std::vector<uint32_t> lens;
    {
        uint32_t maxLen = 0;
        for (uint32_t i = 0; i < q->n; i++) maxLen = std::max<uint32_t>(maxLen, (uint32_t) (q->off[i + 1] - q->off[i]));
        std::vector<uint8_t> present((size_t) maxLen + 1, 0);
        for (uint32_t i = 0; i < q->n; i++) present[q->off[i + 1] - q->off[i]] = 1;
        for (uint32_t L = 0; L <= maxLen; L++) if (present[L] && (L > 0 || q->n > 0)) lens.push_back(L);
    }
    std::shared_ptr<const ScoreTabs> cur = db->scoreTabs;
    const bool sameThr = cur && cur->thr == evalThr && !cur->tables.lenIdx.empty();
    if (sameThr && std::includes(cur->lens.begin(), cur->lens.end(), lens.begin(), lens.end())) return cur;
    std::vector<uint32_t> all = lens;
    if (sameThr) {
        std::vector<uint32_t> merged;
        std::set_union(cur->lens.begin(), cur->lens.end(), lens.begin(), lens.end(), std::back_inserter(merged));
        if (merged.size() <= 4096) all.swap(merged);                          // (bounded: 32 KB per length and snapshot)
    }
    std::vector<uint64_t> off(all.size() + 1, 0);                             // one stand-in query per length
    for (size_t k = 0; k < all.size(); k++) off[k + 1] = off[k] + all[k];
    auto nt = std::make_shared<ScoreTabs>();
    nt->tables.bitScore = db->bitScoreTable;
    mk::build_assemble_tables(db->evaluer, off, nt->tables, &db->evalueRows);
    mk::build_gate_table(db->evaluer, evalThr, off, nt->gate, &nt->tables);
    nt->lens = all; nt->thr = evalThr;
    if (db->evalueRows.size() > 8192) db->evalueRows.clear();                 // (bounded: 32 KB per length)
    db->scoreTabs = nt;
    return nt;
}



namespace {

void engine_drain_if_running();
// drain: the entry points that use the library's streams and scratch buffers first let the searches in flight (mk_search_begin) finish
int ensure_ready(bool drain = true) {
    if (!g_ready) return fail(MK_ERR_DEVICE, "mk_init() was not called or no HIP device is usable");
    // the device binding is per host thread: a caller's helper thread (the commands prefetch their first batch on one) works on the GPU mk_init chose
    static thread_local bool deviceBound = false;
    if (!deviceBound) { (void) hipSetDevice(g_device); deviceBound = true; }
    if (drain) engine_drain_if_running();
    return MK_OK;
}

mk::AlignView align_view(const mk_targetdb *db, const mk_queries *q) {
    mk::AlignView V;
    V.q_res = q->dRes.p; V.q_bias8 = q->dBias8.p; V.q_off = q->dOff.p; V.n_queries = q->n;
    V.q_prof = q->isProfile ? q->dProfAln.p : nullptr;
    V.t_res = db->dRes.p; V.t_off = db->dOff.p; V.n_targets = db->n; V.mat_aln = db->dMatAln.p;
    V.max_q_len = q->maxLen; V.max_t_len = db->maxLen;
    return V;
}

// Test-path SW: explicit jobs built on the host, grouped by tile configuration.
int run_sw_jobs(const mk_targetdb *db, const mk_queries *q, const mk_params *P, std::vector<mk::SwJob> &jobs,
                std::vector<mk::SwOut> &out, const char *statName) {
    const size_t n = jobs.size();
    out.assign(n, mk::SwOut{0, -1, -1, 0});
    if (n == 0) return MK_OK;
    std::vector<mk::SwJob> sorted;
    sorted.reserve(n);
    uint32_t bounds[mk::SW_NCFG + 1];
    for (int c = 0; c < mk::SW_NCFG; c++) {
        bounds[c] = (uint32_t) sorted.size();
        for (size_t i = 0; i < n; i++)
            if (mk::sw_cfg_of(jobs[i].q_len) == c) { mk::SwJob j = jobs[i]; j.slot = (uint32_t) i; sorted.push_back(j); }
    }
    bounds[mk::SW_NCFG] = (uint32_t) sorted.size();
    DevBuf<mk::SwJob> dJobs;
    DevBuf<mk::SwOut> dOut;
    DevBuf<uint32_t> dBorder;
    HIPCHK(dJobs.upload(sorted.data(), n));
    HIPCHK(dOut.alloc(n));
    for (int c = 0; c < mk::SW_NCFG; c++) {
        const uint32_t lo = bounds[c], hi = bounds[c + 1];
        if (hi == lo) continue;
        mk::SwLaunch L;
        L.q_res = q->dRes.p; L.q_bias8 = q->dBias8.p; L.q_prof = q->isProfile ? q->dProfAln.p : nullptr; L.t_res = db->dRes.p; L.mat = db->dMatAln.p;
        L.jobs = dJobs.p + lo; L.out = dOut.p; L.n_jobs = hi - lo; L.order = nullptr;
        L.boundary = nullptr; L.boundary_stride = 0; L.boundary_job0 = 0;
        L.wave_start = nullptr; L.n_waves = 0; L.work_counter = nullptr; L.persistent_blocks = 0; L.units_per_block = 0; L.known_score = nullptr;
        L.gap_open = P->gap_open; L.gap_extend = P->gap_extend;
        double cells = 0, bytes = 0;
        uint32_t maxT = 0; bool multi = false;
        for (uint32_t i = lo; i < hi; i++) {
            cells += (double) sorted[i].q_len * (double) sorted[i].t_len;
            bytes += (double) sorted[i].t_len + 2.0 * sorted[i].q_len + sizeof(mk::SwJob) + sizeof(mk::SwOut);
            maxT = std::max(maxT, sorted[i].t_len);
            if (sorted[i].q_len > (uint32_t) mk::sw_cfg_rows(c)) multi = true;
        }
        if (multi) {
            HIPCHK(dBorder.alloc((size_t) (hi - lo) * maxT));
            L.boundary = dBorder.p; L.boundary_stride = maxT;
        }
        char nm[64];
        snprintf(nm, sizeof(nm), "%s_rows%d", statName, mk::sw_cfg_rows(c));
        const int th = timed_begin(nm, bytes, cells);
        HIPCHK(mk::launch_sw(L, c, g_stream));
        timed_end(th);
    }
    HIPCHK(hipMemcpyAsync(out.data(), dOut.p, n * sizeof(mk::SwOut), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    timed_flush();
    return MK_OK;
}

}  // namespace

// bulk forms: pieces of the record range are formatted in parallel into per-piece strings, then laid out in order
template <typename T, typename F>
size_t format_bulk(char *buf, size_t cap, const T *rec, uint64_t n, size_t maxLine, F &&one) {
    if (!buf || (!rec && n)) return 0;
    if (cap < n * maxLine) return 0;
    const uint64_t PIECE = 1u << 16;
    const uint64_t nPieces = (n + PIECE - 1) / PIECE;
    std::vector<size_t> len(nPieces + 1, 0);
    // every piece is written at its worst-case position first, then moved down to close the gaps (pieces only move forward)
#pragma omp parallel for schedule(dynamic, 1)
    for (uint64_t p = 0; p < nPieces; p++) {
        char *dst = buf + p * PIECE * maxLine, *w = dst;
        const uint64_t e = std::min(n, (p + 1) * PIECE);
        for (uint64_t i = p * PIECE; i < e; i++) w += one(w, rec[i]);
        len[p + 1] = (size_t) (w - dst);
    }
    size_t at = 0;
    for (uint64_t p = 0; p < nPieces; p++) {
        if (p) std::memmove(buf + at, buf + p * PIECE * maxLine, len[p + 1]);
        at += len[p + 1];
    }
    return at;
}

extern "C" {

const char *mk_last_error(void) { return g_err.c_str(); }
int mk_host_threads(void) { return effective_cpus(); }

int mk_init(int device) {
    if (!mk::launcher_env("OMP_NUM_THREADS")) {
        int share = 1;                                      // one process per GPU: the ranks of a node split its cores
        if (const char *lw = mk::launcher_env("LOCAL_WORLD_SIZE")) share = std::max(1, atoi(lw));
        omp_set_num_threads(std::max(1, effective_cpus() / share));
    }
    kmp_set_blocktime(0);                                   // idle team threads sleep: two stages share the host cores in mk_search
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(MK_ERR_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(MK_ERR_ARG, "device ordinal %d out of range (%d devices)", device, count);
    HIPCHK(hipSetDevice(device));
    {
        // two streams for the two stages of mk_search; MK_STREAM_PRIORITY=1: the prefilter's (latency-bound, few instructions) ahead
        // of the alignment's in the dispatcher
        int least = 0, greatest = 0;
        const char *pr = mk::knob("MK_STREAM_PRIORITY");
        const bool prio = pr && atoi(pr) != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
        if (!g_stream) HIPCHK(prio ? hipStreamCreateWithPriority(&g_stream, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
        if (!g_stream2) HIPCHK(prio ? hipStreamCreateWithPriority(&g_stream2, hipStreamNonBlocking, least) : hipStreamCreateWithFlags(&g_stream2, hipStreamNonBlocking));
        g_alignStreams[0] = g_stream2;
        if (!g_uploadStream) HIPCHK(hipStreamCreateWithFlags(&g_uploadStream, hipStreamNonBlocking));
        if (!g_prefStream2) HIPCHK(hipStreamCreateWithFlags(&g_prefStream2, hipStreamNonBlocking));
        for (int w = 1; w < MAX_ALIGN_WORKERS; w++)
            if (!g_alignStreams[w]) HIPCHK(prio ? hipStreamCreateWithPriority(&g_alignStreams[w], hipStreamNonBlocking, least) : hipStreamCreateWithFlags(&g_alignStreams[w], hipStreamNonBlocking));
    }
    g_device = device;
    g_ready = true;
    return MK_OK;
}

int mk_device_name(char *buf, size_t cap) {
    int rc = ensure_ready();
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, g_device));
    snprintf(buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return MK_OK;
}

int mk_device_memory(uint64_t *freeBytes, uint64_t *totalBytes) {
    int rc = ensure_ready();
    if (rc) return rc;
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (freeBytes) *freeBytes = f;
    if (totalBytes) *totalBytes = t;
    return MK_OK;
}

void mk_default_params(mk_params *p) {
    p->sensitivity = 5.7f; p->kmer_score = INT_MAX; p->max_seqs = 300; p->min_ungapped_score = 15;
    p->comp_bias_corr = 1; p->comp_bias_scale = 1.0f; p->mask = 1; p->mask_prob = 0.9f;
    p->gap_open = 11; p->gap_extend = 1; p->evalue_thr = 100.0; p->min_aln_len = 11;
    p->simd_lanes_byte = 32; p->simd_lanes_word = 16; p->simd_lanes_double = 4;
    p->host_l2_bytes = 1048576;
    p->profile_search = 0;
    p->kmer_size = 0;
}

void mk_encode(const char *ascii, size_t len, uint8_t *codes) { mk::encode(ascii, len, codes); }

// k-mer lists that exist already (an index DB)
struct PrebuiltIndex {
    mk::TargetIndex *host = nullptr;             // k = 6: lists in the tiled address order + masked residues, in host memory
    // k = 7: views into the mapped index DB (reference numbering = the device's numbering), streamed to the device
    const uint64_t *fileOffsets = nullptr; const unsigned char *fileEntries6 = nullptr; uint64_t nEntries = 0; const uint8_t *masked = nullptr;
    int kmerSize = 6;
};

static const uint8_t *masked_host(mk_targetdb *db) {
    static std::mutex m;                                 // (two prefilter threads of the search engine may ask at once)
    std::lock_guard<std::mutex> lk(m);
    if (!db->maskedHostReady) {
        db->maskedHost.resize(db->off[db->n]);
        if (db->off[db->n] && hipMemcpy(db->maskedHost.data(), db->dMasked.p, db->off[db->n], hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
        db->maskedHostReady = true;
    }
    return db->maskedHost.data();
}

static void adopt_index(mk_targetdb *db, mk::DeviceIndex &ix, uint64_t total) {
    db->dMasked.p = ix.masked; db->dMasked.n = total;
    db->dKmerSlot.p = ix.slots; db->dKmerSlot.n = ix.cells;
    db->dKmerBits.p = ix.bits; db->dKmerBits.n = (ix.cells + 31) / 32;
    db->dEntries.p = ix.entries; db->dEntries.n = ix.n_entries;
    db->nEntries = ix.n_entries; db->cells = ix.cells; db->maskedResidues = ix.masked_residues; db->maxList = ix.max_list;
    ix = mk::DeviceIndex();
}

// target side: matrices, tables, masking + k-mer index (built in HBM, or taken from an index DB), upload
static int targetdb_create(const uint8_t *residues, const uint64_t *offsets, uint32_t n, const mk_params *P, const PrebuiltIndex *prebuilt, mk_targetdb **out,
                           bool noIndex = false) {
    dev_pool_trim();                                     // (a database is built rarely and may need every byte)
    int rc = ensure_ready();
    if (rc) return rc;
    if (!residues || !offsets || !P || !out) return fail(MK_ERR_ARG, "null argument");
    // IndexTable::computeKmerSize (IndexTable.h:439-449): from 3.35e9 target residues on the reference searches with k = 7
    if (P->kmer_size != 0 && P->kmer_size != 6 && P->kmer_size != 7) return fail(MK_ERR_UNSUPPORTED, "-k %d: k-mer sizes 6 and 7 are implemented", P->kmer_size);
    const int kmerSize = prebuilt ? prebuilt->kmerSize : (P->kmer_size ? P->kmer_size : (offsets[n] < 3350000000ull ? 6 : 7));
    if (prebuilt && P->kmer_size && P->kmer_size != kmerSize) return fail(MK_ERR_ARG, "-k %d, but the index DB was built with k = %d", P->kmer_size, kmerSize);
    mk_targetdb *db = new mk_targetdb();
    db->kmerSize = kmerSize;
    db->n = n;
    db->off.assign(offsets, offsets + n + 1);
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t L = offsets[i + 1] - offsets[i];
        if (L >= MK_MAX_SEQ_LEN) { delete db; return fail(MK_ERR_UNSUPPORTED, "target %u has %llu residues: at most %u are supported (17-bit column field of the alignment kernels)", i, (unsigned long long) L, MK_MAX_SEQ_LEN - 1); }
        db->maxLen = std::max<uint32_t>(db->maxLen, (uint32_t) L);
    }
    db->profileSearch = P->profile_search != 0;
    if (db->profileSearch && prebuilt) { delete db; return fail(MK_ERR_UNSUPPORTED, "a precomputed index cannot serve profile queries: it was masked with the seed matrix's background and filtered by the sequence threshold"); }
    // profile queries: kmerSubMat is --sub-mat x8 (only the background of the masking uses it), Prefiltering.cpp:72-76
    mk::build_submat(db->kmerMat, db->profileSearch ? mk::MAT_BLOSUM62 : mk::MAT_VTML80, 8.0f, -0.2f);     // Prefiltering.cpp:68
    mk::build_submat(db->ungMat, mk::MAT_BLOSUM62, 2.0f, -0.2f);    // Prefiltering.cpp:69
    mk::build_submat(db->alnMat, mk::MAT_BLOSUM62, 2.0f, 0.0f);     // Alignment.cpp:152
    // ... and the index keeps every k-mer (localKmerThr = 0, Prefiltering.cpp:525-527)
    db->kmerThr = db->profileSearch ? 0 : (kmerSize == 7 ? mk::kmer_threshold_k7(P->sensitivity, P->kmer_score) : mk::kmer_threshold(P->sensitivity, P->kmer_score));
    db->evaluer.init(offsets[n]);
    db->bitScoreTable.resize(32768);
    for (int sc = 0; sc < 32768; sc++) db->bitScoreTable[sc] = static_cast<int>(db->evaluer.bitScore((double) sc) + 0.5);
    // test hook: shift every list start by this many entries (and the device pointer back by as many), so that the slots' 40-bit starts
    // are exercised beyond 2^32 without a database of that size
    const uint64_t entryShift = mk::knob("MK_TEST_ENTRY_BASE") ? strtoull(mk::knob("MK_TEST_ENTRY_BASE"), nullptr, 10) : 0;
    if (entryShift >= (1ull << 39)) { delete db; return fail(MK_ERR_ARG, "MK_TEST_ENTRY_BASE too large"); }
    db->entryShift = entryShift;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    ok(db->dRes.upload(residues, offsets[n]));
    ok(db->dOff.upload(offsets, n + 1));
    if (e != hipSuccess) { delete db; return fail(MK_ERR_DEVICE, "target upload failed: %s", hipGetErrorString(e)); }
    if (noIndex) {
        // the alignment stage's view of the targets (Alignment.cpp opens the sequence DB alone): residues, offsets, matrices, e-values
        db->unindexed = true;
        int8_t mA[441], mU[441];
        for (int i = 0; i < 21; i++)
            for (int j = 0; j < 21; j++) { mA[i * 21 + j] = (int8_t) db->alnMat.sub[i][j]; mU[i * 21 + j] = (int8_t) db->ungMat.sub[i][j]; }
        ok(db->dMatAln.upload(mA, 441));
        ok(db->dMatUng.upload(mU, 441));
        ok(hipStreamSynchronize(g_stream));
        if (e != hipSuccess) { delete db; return fail(MK_ERR_DEVICE, "target upload failed: %s", hipGetErrorString(e)); }
        *out = db;
        return MK_OK;
    }
    // MK_INDEX_BUILD=host: mask and index on the host (mk::build_index, the reference of the device builder)
    const char *ib = mk::knob("MK_INDEX_BUILD");
    const bool hostBuild = !prebuilt && ib && strcmp(ib, "host") == 0;
    if ((prebuilt && prebuilt->host) || hostBuild) {
        mk::TargetIndex built;
        if (!prebuilt) mk::build_index(db->kmerMat, residues, offsets, n, db->kmerThr, P->mask != 0, P->mask_prob, P->simd_lanes_double, built, true, kmerSize);
        mk::TargetIndex &ix = prebuilt ? *prebuilt->host : built;
        if (ix.entries.size() + entryShift >= (1ull << 40)) { delete db; return fail(MK_ERR_UNSUPPORTED, "index has >= 2^40 entries"); }
        db->nEntries = ix.entries.size();
        db->maskedResidues = ix.maskedResidues;
        const size_t nKmers = ix.offsets.size() - 1;
        db->cells = nKmers;
        std::vector<uint64_t> slots(nKmers);
        std::vector<uint32_t> bits((nKmers + 31) / 32, 0u);
        int tooLong = 0;
        uint32_t maxList = 0;
#pragma omp parallel for schedule(static) reduction(| : tooLong) reduction(max : maxList)
        for (size_t wd = 0; wd < bits.size(); wd++) {
            uint32_t m = 0;
            const size_t k0 = wd * 32, k1 = std::min(k0 + 32, nKmers);
            for (size_t k = k0; k < k1; k++) {
                const uint64_t first = ix.offsets[k], len = ix.offsets[k + 1] - first;
                if (len) m |= 1u << (k - k0);
                if (len >= (1ull << 23)) tooLong |= 1;
                maxList = std::max<uint32_t>(maxList, (uint32_t) std::min<uint64_t>(len, 0xFFFFFFFFull));
                slots[k] = len == 1 ? ((1ull << 63) | ix.entries[first]) : ((first + entryShift) | (len << 40));
            }
            bits[wd] = m;
        }
        if (tooLong) { delete db; return fail(MK_ERR_UNSUPPORTED, "a k-mer occurs in 2^23 or more targets: the slot's length field holds 23 bits"); }
        db->maxList = maxList;
        ok(db->dMasked.upload(ix.masked.data(), ix.masked.size()));
        ok(db->dKmerSlot.upload(slots.data(), slots.size()));
        ok(db->dKmerBits.upload(bits.data(), bits.size()));
        ok(db->dEntries.upload(ix.entries.data(), ix.entries.size()));
        ok(hipStreamSynchronize(g_stream));
    } else {
        mk::DeviceIndex ix;
        std::string err;
        if (prebuilt) {
            ok(hipMalloc(reinterpret_cast<void **>(&ix.masked), std::max<uint64_t>(offsets[n], 1)));
            if (e == hipSuccess && offsets[n]) ok(hipMemcpy(ix.masked, prebuilt->masked, offsets[n], hipMemcpyHostToDevice));
            if (e == hipSuccess) rc = mk::device_index_from_file(prebuilt->fileOffsets, prebuilt->fileEntries6, prebuilt->nEntries, kmerSize, entryShift, n, g_stream, ix, err);
        } else {
            HostTimer ht("host_index_build_total");
            mk::IndexBuildParams B;
            B.kmer_size = kmerSize; B.kmer_thr = db->kmerThr; B.mask = P->mask != 0; B.mask_prob = static_cast<double>(P->mask_prob);
            B.tantan_lanes = P->simd_lanes_double; B.reference_order = false; B.entry_shift = entryShift;
            rc = mk::device_build_index(db->dRes.p, db->dOff.p, db->off, n, db->kmerMat, B, g_stream, ix, err, timed_begin, timed_end);
            timed_flush();
        }
        if (e != hipSuccess || rc != MK_OK) {
            ix.release();
            delete db;
            return e != hipSuccess ? fail(MK_ERR_DEVICE, "target upload failed: %s", hipGetErrorString(e)) : fail(rc, "%s", err.c_str());
        }
        adopt_index(db, ix, offsets[n]);
    }
    mk::ScoreMat3 sm;
    mk::build_scoremat3(db->kmerMat, sm);
    int8_t matAln[441], matUng[441];
    for (int i = 0; i < 21; i++)
        for (int j = 0; j < 21; j++) { matAln[i * 21 + j] = (int8_t) db->alnMat.sub[i][j]; matUng[i * 21 + j] = (int8_t) db->ungMat.sub[i][j]; }
    ok(db->dScore3.upload(sm.score.data(), sm.score.size()));
    ok(db->dIndex3.upload(sm.index.data(), sm.index.size()));
    ok(db->dHist3.upload(sm.hist.data(), sm.hist.size()));
    ok(db->dCum3.upload(sm.cum.data(), sm.cum.size()));
    db->histLo = sm.histLo; db->histRange = sm.histRange;
    ok(db->dMatAln.upload(matAln, 441));
    ok(db->dMatUng.upload(matUng, 441));
    uint16_t addr3[8000];
    mk::kmer3_address_table(addr3);
    ok(db->dAddr3.upload(addr3, 8000));
    std::vector<int16_t> score2;
    std::vector<uint16_t> index2;
    uint16_t num3[8000];
    if (kmerSize == 7) {
        mk::build_scoremat2(db->kmerMat, score2, index2);
        mk::kmer3_number_of_address(num3);
        ok(db->dScore2.upload(score2.data(), score2.size()));
        ok(db->dIndex2.upload(index2.data(), index2.size()));
        ok(db->dNum3.upload(num3, 8000));
    }
    ok(hipStreamSynchronize(g_stream));
    if (e != hipSuccess) { delete db; return fail(MK_ERR_DEVICE, "target upload failed: %s", hipGetErrorString(e)); }
    *out = db;
    return MK_OK;
}

void mk_targetdb_destroy(mk_targetdb *db) { engine_drain_if_running(); delete db; }
int mk_targetdb_create(const uint8_t *residues, const uint64_t *offsets, uint32_t n, const mk_params *P, mk_targetdb **out) {
    return targetdb_create(residues, offsets, n, P, nullptr, out);
}
int mk_targetdb_create_sequences(const uint8_t *residues, const uint64_t *offsets, uint32_t n, const mk_params *P, mk_targetdb **out) {
    return targetdb_create(residues, offsets, n, P, nullptr, out, true);
}

// ---- createindex's precomputed index DB (mk_indexfile.cpp) ----
static void encode_seq_db(const mk::SeqDbImage &db, std::vector<uint8_t> &res, std::vector<uint64_t> &off) {
    const size_t n = db.keys.size();
    off.assign(n + 1, 0);
    for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + (db.lengths[i] >= 2 ? db.lengths[i] - 2 : 0);      // entries are "SEQ\n\0"
    res.assign(off[n] + 1, 0);
#pragma omp parallel for schedule(dynamic, 256)
    for (size_t i = 0; i < n; i++) mk::encode(db.data.data() + db.offsets[i], off[i + 1] - off[i], res.data() + off[i]);
}

// createindex: mask + index the sequence DB and write the reference's index DB.  With a GPU (mk_init was called) the lists are built
// in HBM (mk_index.hip) and streamed into the file -- any size, k = 6 or 7; without one the host builder does it (k = 6, no GPU needed).
int mk_index_write(const char *indexDb, const char *seqData, uint64_t seqDataSize, const uint32_t *keys, const uint64_t *offsets,
                   const uint32_t *lengths, uint32_t n, int seqDbtype, const mk_params *P) {
    if (!indexDb || !keys || !offsets || !lengths || !P || (!seqData && seqDataSize)) return fail(MK_ERR_ARG, "null argument");
    if ((seqDbtype & 0xFFFF) != 0) return fail(MK_ERR_UNSUPPORTED, "only amino-acid sequence databases can be indexed (profile targets: SURVEY 8f-4)");
    mk::IndexFileContent c;
    c.seqs.keys.assign(keys, keys + n); c.seqs.offsets.assign(offsets, offsets + n); c.seqs.lengths.assign(lengths, lengths + n);
    c.seqs.data.assign(seqData, seqData + seqDataSize);
    c.seqs.dbtype = seqDbtype;
    for (uint32_t i = 0; i < n; i++) {
        if (offsets[i] + lengths[i] > seqDataSize || lengths[i] < 2) return fail(MK_ERR_ARG, "entry %u lies outside the sequence data", i);
        if (lengths[i] - 2 >= MK_MAX_SEQ_LEN) return fail(MK_ERR_UNSUPPORTED, "target %u has %u residues: at most %u are supported", i, lengths[i] - 2, MK_MAX_SEQ_LEN - 1);
    }
    std::vector<uint8_t> res;
    encode_seq_db(c.seqs, res, c.seqOffsets);
    if (P->kmer_size != 0 && P->kmer_size != 6 && P->kmer_size != 7) return fail(MK_ERR_UNSUPPORTED, "-k %d: index DBs are written with k = 6 or 7", P->kmer_size);
    const int kmerSize = P->kmer_size ? P->kmer_size : (c.seqOffsets[n] < 3350000000ull ? 6 : 7);     // IndexTable::computeKmerSize
    if (P->profile_search) return fail(MK_ERR_UNSUPPORTED, "index DBs for profile queries are not implemented (they need their own masking background and an unfiltered index)");
    mk::SubMat km;
    mk::build_submat(km, mk::MAT_VTML80, 8.0f, -0.2f);
    c.meta.kmerSize = kmerSize;
    c.meta.kmerThr = kmerSize == 7 ? mk::kmer_threshold_k7(P->sensitivity, P->kmer_score) : mk::kmer_threshold(P->sensitivity, P->kmer_score);
    c.meta.mask = P->mask != 0; c.meta.compBiasCorr = P->comp_bias_corr != 0; c.meta.seqType = seqDbtype; c.meta.srcSeqType = seqDbtype;
    const char *ib = mk::knob("MK_INDEX_BUILD");
    if (!g_ready || (ib && strcmp(ib, "host") == 0)) {
        mk::build_index(km, res.data(), c.seqOffsets.data(), n, c.meta.kmerThr, P->mask != 0, P->mask_prob, P->simd_lanes_double, c.index, false, kmerSize);
        const std::string e = mk::write_index_file(indexDb, km, c);
        if (!e.empty()) return fail(MK_ERR_ARG, "%s", e.c_str());
        return MK_OK;
    }
    // device build, cells in the reference's numbering; the file is written while the lists stream out of HBM
    DevBuf<uint8_t> dRes;
    DevBuf<uint64_t> dOff;
    HIPCHK(dRes.upload(res.data(), c.seqOffsets[n]));
    HIPCHK(dOff.upload(c.seqOffsets.data(), n + 1));
    mk::IndexBuildParams B;
    B.kmer_size = kmerSize; B.kmer_thr = c.meta.kmerThr; B.mask = P->mask != 0; B.mask_prob = static_cast<double>(P->mask_prob);
    B.tantan_lanes = P->simd_lanes_double; B.reference_order = true; B.entry_shift = 0;
    mk::DeviceIndex ix;
    std::string err;
    int rc = mk::device_build_index(dRes.p, dOff.p, c.seqOffsets, n, km, B, g_stream, ix, err, nullptr, nullptr);
    if (rc != MK_OK) { ix.release(); return fail(rc, "%s", err.c_str()); }
    std::vector<uint8_t> masked(c.seqOffsets[n]);
    if (c.seqOffsets[n] && hipMemcpy(masked.data(), ix.masked, c.seqOffsets[n], hipMemcpyDeviceToHost) != hipSuccess) { ix.release(); return fail(MK_ERR_DEVICE, "cannot fetch the masked residues"); }
    std::vector<uint64_t> listOff;
    mk::IndexListSource src;
    src.cells = ix.cells; src.nEntries = ix.n_entries;
    src.masked = masked.data(); src.maskedSize = masked.size();
    int rcStream = MK_OK;
    src.entries6 = [&](const std::function<bool(const void *, size_t)> &sink) { rcStream = mk::device_index_entries6(ix, g_stream, sink, err); return rcStream == MK_OK; };
    src.offsets = [&]() -> const uint64_t * { rcStream = mk::device_index_offsets(ix, g_stream, listOff, err); return rcStream == MK_OK ? listOff.data() : nullptr; };
    const std::string e = mk::write_index_file(indexDb, km, c, src);
    ix.release();
    if (rcStream != MK_OK) return fail(rcStream, "%s", err.c_str());
    if (!e.empty()) return fail(MK_ERR_ARG, "%s", e.c_str());
    return MK_OK;
}

int mk_targetdb_open_index(const char *indexDb, const mk_params *P, mk_targetdb **out) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!indexDb || !P || !out) return fail(MK_ERR_ARG, "null argument");
    mk::IndexFileContent c;
    const std::string e = mk::read_index_file(indexDb, c, true);
    if (!e.empty()) return fail(MK_ERR_UNSUPPORTED, "%s", e.c_str());
    std::vector<uint8_t> res;
    std::vector<uint64_t> off;
    encode_seq_db(c.seqs, res, off);
    if (off != c.seqOffsets) return fail(MK_ERR_ARG, "%s: the masked sequences do not line up with the sequence database", indexDb);
    const std::vector<uint32_t> keys = c.seqs.keys;
    { std::vector<char>().swap(c.seqs.data); }                           // the ASCII sequences are encoded: drop the copy
    PrebuiltIndex pre;
    pre.kmerSize = c.meta.kmerSize;
    if (c.meta.kmerSize == 7) {
        // cells of the device table = the reference's k-mer numbers: offsets and 6-byte entries go from the mapped file to HBM in pieces
        pre.fileOffsets = c.listOffsets; pre.fileEntries6 = c.listEntries6; pre.nEntries = c.nEntries; pre.masked = c.maskedView;
    } else {
        mk::materialize_lists(c);
        mk::index_to_address_order(c.index);
        pre.host = &c.index;
    }
    rc = targetdb_create(res.data(), off.data(), (uint32_t) keys.size(), P, &pre, out);
    if (rc == MK_OK) rc = mk_targetdb_set_keys(*out, keys.data(), (uint32_t) keys.size());
    return rc;
}

// test hook (host only): what a reader takes from an index DB, as text -- masked_targets.txt (one masked sequence per line),
// index.txt ("kmer seqId:pos ..." per non-empty list, the reference's k-mer numbering), seqs.txt ("key<TAB>sequence")
int mk_index_dump(const char *indexDb, const char *outDir) {
    if (!indexDb || !outDir) return fail(MK_ERR_ARG, "null argument");
    mk::IndexFileContent c;
    const std::string e = mk::read_index_file(indexDb, c);
    if (!e.empty()) return fail(MK_ERR_UNSUPPORTED, "%s", e.c_str());
    static const char LETTERS[] = "ACDEFGHIKLMNPQRSTVWYX";
    const std::string d = outDir;
    FILE *f = fopen((d + "/masked_targets.txt").c_str(), "w");
    if (!f) return fail(MK_ERR_ARG, "cannot write to %s", outDir);
    for (size_t i = 0; i + 1 < c.seqOffsets.size(); i++) {
        for (uint64_t p = c.seqOffsets[i]; p < c.seqOffsets[i + 1]; p++) fputc(LETTERS[c.index.masked[p] <= 20 ? c.index.masked[p] : 20], f);
        fputc('\n', f);
    }
    fclose(f);
    f = fopen((d + "/index.txt").c_str(), "w");
    if (!f) return fail(MK_ERR_ARG, "cannot write to %s", outDir);
    for (size_t k = 0; k + 1 < c.index.offsets.size(); k++) {
        if (c.index.offsets[k + 1] == c.index.offsets[k]) continue;
        fprintf(f, "%zu", k);
        for (uint64_t j = c.index.offsets[k]; j < c.index.offsets[k + 1]; j++) fprintf(f, " %u:%u", (unsigned) (uint32_t) c.index.entries[j], (unsigned) (uint16_t) (c.index.entries[j] >> 32));
        fputc('\n', f);
    }
    fclose(f);
    f = fopen((d + "/seqs.txt").c_str(), "w");
    if (!f) return fail(MK_ERR_ARG, "cannot write to %s", outDir);
    for (size_t i = 0; i < c.seqs.keys.size(); i++) {
        fprintf(f, "%u\t", c.seqs.keys[i]);
        fwrite(c.seqs.data.data() + c.seqs.offsets[i], 1, c.seqs.lengths[i] >= 2 ? c.seqs.lengths[i] - 2 : 0, f);
        fputc('\n', f);
    }
    fclose(f);
    f = fopen((d + "/meta.txt").c_str(), "w");
    if (!f) return fail(MK_ERR_ARG, "cannot write to %s", outDir);
    fprintf(f, "maxSeqLen %d\nkmerSize %d\ncompBiasCorr %d\nalphabetSize %d\nmask %d\nspacedKmer %d\nkmerThr %d\nseqType %d\nsrcSeqType %d\nheaders1 %d\nheaders2 %d\nsplits %d\nmatrix %s\n",
            c.meta.maxSeqLen, c.meta.kmerSize, c.meta.compBiasCorr, c.meta.alphabetSize, c.meta.mask, c.meta.spacedKmer, c.meta.kmerThr, c.meta.seqType,
            c.meta.srcSeqType, c.meta.headers1, c.meta.headers2, c.meta.splits, c.matrixName.c_str());
    fclose(f);
    return MK_OK;
}

int mk_targetdb_set_keys(mk_targetdb *db, const uint32_t *keys, uint32_t n) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!db || !keys || n != db->n) return fail(MK_ERR_ARG, "mk_targetdb_set_keys: one key per target");
    db->keys.assign(keys, keys + n);
    HIPCHK(db->dKeys.upload(keys, n));
    HIPCHK(hipStreamSynchronize(g_stream));
    return MK_OK;
}

int mk_targetdb_keys(const mk_targetdb *db, const uint32_t **keys, uint32_t *n) {
    if (!db || !keys || !n) return fail(MK_ERR_ARG, "null argument");
    *keys = db->keys.empty() ? nullptr : db->keys.data(); *n = db->n;
    return MK_OK;
}

uint64_t mk_targetdb_residues(const mk_targetdb *db) { return db ? db->off[db->n] : 0; }
uint64_t mk_targetdb_index_entries(const mk_targetdb *db) { return db ? db->nEntries : 0; }
int mk_targetdb_masked(const mk_targetdb *db, uint8_t *out) {
    if (!db || !out) return fail(MK_ERR_ARG, "null argument");
    if (db->unindexed) return fail(MK_ERR_ARG, "this target database holds no masked residues (mk_targetdb_create_sequences)");
    const uint8_t *m = masked_host(const_cast<mk_targetdb *>(db));
    if (!m) return fail(MK_ERR_DEVICE, "cannot fetch the masked residues from the device");
    std::memcpy(out, m, db->off[db->n]);
    return MK_OK;
}
uint64_t mk_targetdb_masked_residues(const mk_targetdb *db) { return db ? db->maskedResidues : 0; }
int mk_targetdb_kmer_size(const mk_targetdb *db) { return db ? db->kmerSize : 0; }
uint32_t mk_targetdb_longest_list(const mk_targetdb *db) { return db ? db->maxList : 0; }

// test hook: words / entries / residues in which the tables of two databases differ (slot table, presence bits, entries, masked residues);
// all ~0 when the sizes differ
int mk_targetdb_index_compare(const mk_targetdb *a, const mk_targetdb *b, uint64_t diff[4]) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!a || !b || !diff) return fail(MK_ERR_ARG, "null argument");
    mk::DeviceIndex x, y;
    x.masked = a->dMasked.p; x.slots = a->dKmerSlot.p; x.bits = a->dKmerBits.p; x.entries = a->dEntries.p; x.cells = a->cells; x.n_entries = a->nEntries;
    y.masked = b->dMasked.p; y.slots = b->dKmerSlot.p; y.bits = b->dKmerBits.p; y.entries = b->dEntries.p; y.cells = b->cells; y.n_entries = b->nEntries;
    std::string err;
    if (a->off[a->n] != b->off[b->n]) { diff[0] = diff[1] = diff[2] = diff[3] = ~0ull; return MK_OK; }
    rc = mk::device_index_compare(x, y, a->off[a->n], g_stream, diff, err);
    if (rc != MK_OK) return fail(rc, "%s", err.c_str());
    return MK_OK;
}

// residues come from the host (upload) or are already in HBM (devResidues: device-to-device copy); the host copy is kept
// for the rare exact self-score of the --max-seqs path
static int queries_create(const uint8_t *residues, const uint8_t *devResidues, const uint64_t *offsets, uint32_t n, const mk_params *P, mk_queries **out) {
    int rc = ensure_ready(false);          // does not wait for the searches in flight: the next batch is uploaded and derived beside them
    if (rc) return rc;
    if (!residues || !offsets || !P || !out) return fail(MK_ERR_ARG, "null argument");
    if (offsets[n] >= 0xFFFFFFFFull) return fail(MK_ERR_ARG, "query batch too large (>= 2^32 residues): split it");
    mk_queries *q = new mk_queries();
    q->n = n;
    q->off.assign(offsets, offsets + n + 1);
    q->res.assign(residues, residues + offsets[n]);
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t L = offsets[i + 1] - offsets[i];
        if (L >= MK_MAX_SEQ_LEN) { delete q; return fail(MK_ERR_UNSUPPORTED, "query %u has %llu residues: at most %u are supported", i, (unsigned long long) L, MK_MAX_SEQ_LEN - 1); }
        q->maxLen = std::max<uint32_t>(q->maxLen, (uint32_t) L);
    }
    mk::SubMat kmerMat, alnMat;
    mk::build_submat(kmerMat, mk::MAT_VTML80, 8.0f, -0.2f);
    mk::build_submat(alnMat, mk::MAT_BLOSUM62, 2.0f, 0.0f);
    // a stream of its own: g_stream carries the prefilter of the batch in flight
    hipStream_t up = g_uploadStream ? g_uploadStream : g_stream;
    hipStream_t callerStream = t_stream;
    t_stream = up;                               // (the timing events follow the calling thread's stream)
    const int callerLane = mk::scratch_lane();
    mk::set_scratch_lane(UPLOAD_LANE);           // ... and scratch buffers of its own (launch_derive's)
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    const uint64_t total = offsets[n];
    if (devResidues) {
        ok(q->dRes.alloc(total));
        if (e == hipSuccess && total) ok(hipMemcpyAsync(q->dRes.p, devResidues, total, hipMemcpyDeviceToDevice, up));
    } else {
        ok(q->dRes.upload(residues, total, up));
    }
    ok(q->dOff.upload(offsets, n + 1, up));
    ok(q->dKmerThr.alloc(total));
    ok(q->dCorr.alloc(total));
    ok(q->dBias8.alloc(total));
    if (e == hipSuccess) {
        const int th = timed_begin("query_derive", (double) total * 9.0, 0);
        q->kmerSize = P->kmer_size == 7 ? 7 : 6;
        q->derivedWith = *P;
        ok(mk::launch_derive(q->dRes.p, q->dOff.p, n, total, kmerMat, alnMat,
                             q->kmerSize == 7 ? mk::kmer_threshold_k7(P->sensitivity, P->kmer_score) : mk::kmer_threshold(P->sensitivity, P->kmer_score),
                             P->comp_bias_corr != 0, P->comp_bias_scale, q->dKmerThr.p, q->dCorr.p, q->dBias8.p, up, q->kmerSize));
        timed_end(th);
    }
    ok(hipStreamSynchronize(up));
    timed_flush();
    t_stream = callerStream;
    mk::set_scratch_lane(callerLane);
    if (e != hipSuccess) { delete q; return fail(MK_ERR_DEVICE, "query upload failed: %s", hipGetErrorString(e)); }
    // (round 6) the pinned blocks the results will land in are taken NOW, on the caller's thread: pinning is 0.3 ms per MB, and the search threads needed
    // the blocks in the middle of a batch's first chunk -- 0.1 s of the first two batches of a command (`metaeuk-amd predictexons` prepares a batch while
    // the previous one is searched, the first one beside the target index).  A guess -- 48 prefilter hits and 6 accepted alignments per query, 256 MB
    // each at most -- that the stages enlarge when it is too small; the blocks come from and go back to the pool of result blocks
    {
        const double hpq = g_hitsPerQuery.load(), apq = g_alnsPerQuery.load();     // what the last finished search of this process returned per query
        const size_t hb = hpq > 0 ? (size_t) ((double) n * hpq * 1.08) * sizeof(mk_hit) : std::min<size_t>((size_t) n * 48 * sizeof(mk_hit), (size_t) 256 << 20);
        const size_t ab = apq > 0 ? (size_t) ((double) n * apq * 1.08) * sizeof(mk_alignment) : std::min<size_t>((size_t) n * 6 * sizeof(mk_alignment), (size_t) 256 << 20);
        (void) q->hits.reserve(hb + 4096, 0);
        (void) q->alns.reserve(ab + 4096, 0);
    }
    *out = q;
    return MK_OK;
}

int mk_queries_create(const uint8_t *residues, const uint64_t *offsets, uint32_t n, const mk_params *P, mk_queries **out) {
    return queries_create(residues, nullptr, offsets, n, P, out);
}

// ---- profile queries (mk_profile.hip): Sequence::mapProfile for the whole batch on the device ----
int mk_profiles_create(const uint8_t *columns, const uint64_t *offsets, uint32_t n, const mk_params *P, mk_queries **out) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!offsets || !P || !out || (!columns && offsets[n] > 0)) return fail(MK_ERR_ARG, "null argument");
    if (offsets[n] >= 0xFFFFFFFFull / 64) return fail(MK_ERR_ARG, "profile batch too large: split it");
    mk_queries *q = new mk_queries();
    q->n = n;
    q->isProfile = true;
    q->off.assign(offsets, offsets + n + 1);
    const uint64_t total = offsets[n];
    for (uint32_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i]) { delete q; return fail(MK_ERR_ARG, "offsets are not ascending at profile %u", i); }
        const uint64_t L = offsets[i + 1] - offsets[i];
        if (L >= MK_MAX_SEQ_LEN) { delete q; return fail(MK_ERR_UNSUPPORTED, "profile %u has %llu columns: at most %u are supported", i, (unsigned long long) L, MK_MAX_SEQ_LEN - 1); }
        q->maxLen = std::max<uint32_t>(q->maxLen, (uint32_t) L);
    }
    q->res.resize(total);
#pragma omp parallel for schedule(static)
    for (uint64_t c = 0; c < total; c++) q->res[c] = columns[c * mk::PROFILE_COL_BYTES + 20];      // the query letters (host copy: exact self score)
    for (uint64_t c = 0; c < total; c++)
        if (q->res[c] > 20) { delete q; return fail(MK_ERR_ARG, "column %llu: query letter %u is not a residue code", (unsigned long long) c, (unsigned) q->res[c]); }
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    DevBuf<uint8_t> dRaw;
    const uint64_t pad = 512;                                   // the k-mer list kernels stage whole 256 + 9 column windows
    ok(dRaw.alloc((total + pad) * mk::PROFILE_COL_BYTES));
    if (e == hipSuccess) ok(hipMemsetAsync(dRaw.p, 0, (total + pad) * mk::PROFILE_COL_BYTES, g_stream));
    if (e == hipSuccess && total) ok(hipMemcpyAsync(dRaw.p, columns, total * mk::PROFILE_COL_BYTES, hipMemcpyHostToDevice, g_stream));
    ok(q->dOff.upload(offsets, n + 1));
    ok(q->dRes.alloc(total + pad));
    ok(q->dKmerThr.alloc(total + pad));
    ok(q->dCorr.alloc(total + pad));
    ok(q->dBias8.alloc(total + pad));
    ok(q->dProfSorted.alloc((total + pad) * mk::PROFILE_SORTED_STRIDE));
    ok(q->dProfAln.alloc((total + pad) * mk::PROFILE_ALN_STRIDE));
    if (e == hipSuccess) {
        ok(hipMemsetAsync(q->dCorr.p, 0, total + pad, g_stream));
        ok(hipMemsetAsync(q->dBias8.p, 0, total + pad, g_stream));
        ok(hipMemsetAsync(q->dKmerThr.p, 0xFF, (total + pad) * 2, g_stream));       // -1: no k-mer start
        ok(hipMemsetAsync(q->dProfSorted.p, 0, (total + pad) * mk::PROFILE_SORTED_STRIDE, g_stream));
        ok(hipMemsetAsync(q->dProfAln.p, 0, (total + pad) * mk::PROFILE_ALN_STRIDE, g_stream));
        const int th = timed_begin("profile_derive", (double) total * (25.0 + 75.0), 0);
        q->kmerSize = P->kmer_size == 7 ? 7 : 6;
        q->derivedWith = *P;
        ok(mk::launch_profile_derive(dRaw.p, q->dOff.p, n, total, q->kmerSize == 7 ? mk::kmer_threshold_profile_k7(P->sensitivity) : mk::kmer_threshold_profile(P->sensitivity),
                                     q->kmerSize, q->dRes.p, q->dProfSorted.p, q->dProfAln.p, q->dKmerThr.p, g_stream));
        timed_end(th);
    }
    ok(hipStreamSynchronize(g_stream));
    timed_flush();
    if (e != hipSuccess) { delete q; return fail(MK_ERR_DEVICE, "profile upload failed: %s", hipGetErrorString(e)); }
    *out = q;
    return MK_OK;
}

int mk_profiles_derived(const mk_queries *q, uint8_t *letters, int8_t *sorted40, int8_t *aln32, int16_t *kmerThr) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!q || !letters || !sorted40 || !aln32 || !kmerThr) return fail(MK_ERR_ARG, "null argument");
    if (!q->isProfile) return fail(MK_ERR_ARG, "not a profile batch");
    const uint64_t total = q->off[q->n];
    HIPCHK(hipMemcpy(letters, q->dRes.p, total, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sorted40, q->dProfSorted.p, total * mk::PROFILE_SORTED_STRIDE, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(aln32, q->dProfAln.p, total * mk::PROFILE_ALN_STRIDE, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(kmerThr, q->dKmerThr.p, total * sizeof(int16_t), hipMemcpyDeviceToHost));
    return MK_OK;
}

// ---- swapresults on arrays (host code) ----
struct mk_swapped {
    std::unique_ptr<mk_alignment[]> alns;      // (not a vector: 300 MB of records at config-4 scale must not be zero-filled by one thread first)
    std::vector<uint64_t> off;
};

int mk_swap_alignments(const mk_alignment *alns, const uint64_t *offsets, uint32_t nq, const uint32_t *queryKeys, uint32_t nTargets,
                       uint64_t swappedDbResidues, const mk_params *P, mk_swapped **out) {
    if (!offsets || !P || !out || (!alns && offsets[nq] > 0)) return fail(MK_ERR_ARG, "null argument");
    const uint64_t total = offsets[nq];
    HostTimer ht("host_swap_total");
    mk::Evaluer ev;
    ev.init(swappedDbResidues);                        // swapresults.cpp:76-77,102
    const double ln2 = std::log(2.0);
    // 1. per record: the target's list it goes to (none beyond -e; the workflow passes DBL_MAX) and its e-value against the swapped database;
    //    list sizes.  The buffers are written by the threads that read them again (no value-initialisation by one thread in front).
    std::unique_ptr<uint32_t[]> target(new uint32_t[std::max<uint64_t>(total, 1)]);
    std::unique_ptr<double[]> evalue(new double[std::max<uint64_t>(total, 1)]);
    std::vector<uint64_t> cnt((size_t) nTargets + 1, 0);
    uint64_t badRecord = ~0ull;
#pragma omp parallel for schedule(static) reduction(min : badRecord)
    for (uint64_t k = 0; k < total; k++) {
        const mk_alignment &a = alns[k];
        if (a.db_key >= nTargets) { badRecord = std::min(badRecord, k); target[k] = 0xFFFFFFFFu; evalue[k] = 0; continue; }
        // Matcher::result_t::swapResult (Matcher.h:93-115): the e-value of the bit score against the swapped database, the record's
        // target length as the query length
        const double rawScore = (ev.logK + (double) a.bit_score * ln2) / ev.lambda;            // EvalueComputation.h:22-24
        const double e = ev.evalue(rawScore, (double) a.db_len);
        evalue[k] = e;
        const bool keep = e <= P->evalue_thr;                                                    // swapresults.cpp:291-295 (-e)
        target[k] = keep ? a.db_key : 0xFFFFFFFFu;
        if (keep) __atomic_fetch_add(&cnt[(size_t) a.db_key + 1], 1ull, __ATOMIC_RELAXED);
    }
    if (badRecord != ~0ull) return fail(MK_ERR_ARG, "alignment %llu names target %u of %u", (unsigned long long) badRecord, alns[badRecord].db_key, nTargets);
    for (uint32_t t = 0; t < nTargets; t++) cnt[t + 1] += cnt[t];
    mk_swapped *s = new mk_swapped();
    s->alns.reset(new mk_alignment[std::max<uint64_t>(cnt[nTargets], 1)]);
    // 2. every kept record swapped straight into its target's list (parallel over the queries; the order inside a list is settled by the
    //    sort: compareHits is a total order here, its last key -- the query's DB key -- is unique within a list)
    {
        std::vector<uint64_t> fill(cnt.begin(), cnt.end() - 1);
#pragma omp parallel for schedule(dynamic, 64)
        for (uint32_t i = 0; i < nq; i++) {
            for (uint64_t k = offsets[i]; k < offsets[i + 1]; k++) {
                if (target[k] == 0xFFFFFFFFu) continue;
                mk_alignment a = alns[k];
                // the record as swapresults re-reads it from its text (Matcher::parseAlignmentRecord, Matcher.cpp:203-239): the identity has
                // three decimals there -- "0.xyz" with xyz = (int)(seqId * 1000) (Util::fastSeqIdToBuffer), "1.00" for 1.0 -- and the nearest
                // double of that decimal is the correctly rounded quotient xyz / 1000
                if (!(a.seq_id == 1.0f)) a.seq_id = (float) ((double) (int) (a.seq_id * 1000) / 1000.0);
                a.evalue = evalue[k];
                std::swap(a.q_start, a.db_start); std::swap(a.q_end, a.db_end); std::swap(a.q_len, a.db_len); std::swap(a.qcov, a.dbcov);
                a.db_key = queryKeys ? queryKeys[i] : i;
                s->alns[__atomic_fetch_add(&fill[target[k]], 1ull, __ATOMIC_RELAXED)] = a;
            }
        }
    }
    s->off.swap(cnt);
    mk_alignment *base = s->alns.get();
#pragma omp parallel for schedule(dynamic, 1024)
    for (uint32_t t = 0; t < nTargets; t++)
        if (s->off[t + 1] - s->off[t] > 1) std::sort(base + s->off[t], base + s->off[t + 1], mk::alignment_less);
    *out = s;
    return MK_OK;
}

int mk_swapped_result(const mk_swapped *s, const mk_alignment **alns, const uint64_t **offsets) {
    if (!s || !alns || !offsets) return fail(MK_ERR_ARG, "null argument");
    *alns = s->alns.get(); *offsets = s->off.data();
    return MK_OK;
}
void mk_swapped_destroy(mk_swapped *s) { delete s; }

// ---- extractorfs --translate on the device (mk_orf.hip) ----
struct mk_orfs {
    mk::OrfDeviceResult dev;
    std::vector<mk_orf> orfs;
    std::vector<uint64_t> aaOff;
    std::vector<char> aa;
    std::vector<uint8_t> codes;
    uint32_t nContigs = 0;
    ~mk_orfs() { dev.release(); }
};

int mk_extract_orfs(const char *nucleotides, const uint64_t *offsets, uint32_t nContigs, int minCodons, mk_orfs **out) {
    // (round 6) like mk_queries_create this does NOT wait for the searches in flight: the contigs of the next batch are uploaded, scanned and translated
    // on the upload stream with scratch buffers and pooled device blocks of their own, beside the search of the previous batch
    int rc = ensure_ready(false);
    if (rc) return rc;
    if (!offsets || !out || (!nucleotides && offsets[nContigs] > 0) || minCodons < 1) return fail(MK_ERR_ARG, "bad argument");
    for (uint32_t i = 0; i < nContigs; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(MK_ERR_ARG, "offsets are not ascending at contig %u", i);
        if (offsets[i + 1] - offsets[i] >= 0x7FFFFFF0ull) return fail(MK_ERR_UNSUPPORTED, "contig %u is >= 2^31 nucleotides", i);
    }
    HostTimer ht("host_extract_orfs_total");
    hipStream_t up = g_uploadStream ? g_uploadStream : g_stream;
    struct Restore {
        hipStream_t s; int lane;
        ~Restore() { t_stream = s; mk::set_scratch_lane(lane); }
    } restore{t_stream, mk::scratch_lane()};
    t_stream = up;
    mk::set_scratch_lane(UPLOAD_LANE);
    mk_orfs *o = new mk_orfs();
    o->nContigs = nContigs;
    DevBuf<char> dNucl;
    DevBuf<uint64_t> dOff;
    dNucl.pooled = true; dOff.pooled = true;
    hipError_t e = dNucl.alloc(std::max<uint64_t>(offsets[nContigs], 1));
    if (e == hipSuccess && offsets[nContigs]) e = hipMemcpyAsync(dNucl.p, nucleotides, offsets[nContigs], hipMemcpyHostToDevice, up);
    if (e == hipSuccess) e = dOff.upload(offsets, (size_t) nContigs + 1, up);
    if (e != hipSuccess) { delete o; return fail(MK_ERR_DEVICE, "contig upload failed: %s", hipGetErrorString(e)); }
    std::string err;
    const int th = timed_begin("extract_orfs", (double) offsets[nContigs] * 2.0, 0);
    rc = mk::run_extract_orfs(dNucl.p, dOff.p, nContigs, (uint32_t) minCodons, 32734u, (uint64_t) INT_MAX, up, o->dev, err);
    timed_end(th);
    timed_flush();
    if (rc != MK_OK) { delete o; return fail(rc, "%s", err.c_str()); }
    const uint64_t nf = o->dev.n_frag, na = o->dev.n_aa;
    o->orfs.resize(nf); o->aaOff.assign(nf + 1, 0); o->aa.resize(na); o->codes.resize(na);
    if (nf) {
        std::vector<mk::OrfRecord> rec(nf);
        hipError_t c = hipMemcpyAsync(rec.data(), o->dev.records, nf * sizeof(mk::OrfRecord), hipMemcpyDeviceToHost, up);
        if (c == hipSuccess) c = hipMemcpyAsync(o->aaOff.data(), o->dev.aa_off, (nf + 1) * 8, hipMemcpyDeviceToHost, up);
        if (c == hipSuccess) c = hipMemcpyAsync(o->aa.data(), o->dev.aa_ascii, na, hipMemcpyDeviceToHost, up);
        if (c == hipSuccess) c = hipMemcpyAsync(o->codes.data(), o->dev.aa_code, na, hipMemcpyDeviceToHost, up);
        if (c == hipSuccess) c = hipStreamSynchronize(up);
        if (c != hipSuccess) { delete o; return fail(MK_ERR_DEVICE, "fragment download failed: %s", hipGetErrorString(c)); }
#pragma omp parallel for schedule(static)
        for (uint64_t k = 0; k < nf; k++) {
            const mk::OrfRecord &r = rec[k];
            const uint32_t len = (uint32_t) (offsets[r.contig + 1] - offsets[r.contig]);
            const uint32_t sTo = r.s_from + 3 * r.n_aa - 1;           // strand coordinates; the minus strand is mirrored (extractorfs.cpp:88-93)
            mk_orf &h = o->orfs[k];
            h.contig = r.contig;
            h.minus_strand = (r.flags & 4u) ? 1 : 0;
            h.from = h.minus_strand ? (len - 1) - r.s_from : r.s_from;
            h.to = h.minus_strand ? (len - 1) - sTo : sTo;
            h.incomplete_start = (r.flags & 1u) ? 1 : 0; h.incomplete_end = (r.flags & 2u) ? 1 : 0; h.pad_ = 0;
        }
    }
    *out = o;
    return MK_OK;
}

int mk_orfs_result(const mk_orfs *o, const mk_orf **orfs, const uint64_t **aaOffsets, const char **aa, uint64_t *nOrfs) {
    if (!o || !orfs || !aaOffsets || !aa || !nOrfs) return fail(MK_ERR_ARG, "null argument");
    *orfs = o->orfs.data(); *aaOffsets = o->aaOff.data(); *aa = o->aa.data(); *nOrfs = o->orfs.size();
    return MK_OK;
}

int mk_queries_from_orfs(const mk_orfs *o, const mk_params *P, mk_queries **out) {
    if (!o || !P || !out) return fail(MK_ERR_ARG, "null argument");
    static const uint8_t none = 0;
    return queries_create(o->codes.empty() ? &none : o->codes.data(), o->dev.aa_code, o->aaOff.data(), (uint32_t) o->orfs.size(), P, out);
}

void mk_orfs_destroy(mk_orfs *o) { delete o; }
size_t mk_format_orf_header(char *buf, const mk_orf *o) {
    return mk::format_orf_header(buf, o->contig, o->from, o->to, o->incomplete_start != 0, o->incomplete_end != 0);
}


// ---- resultspercontig + collectoptimalset on the alignment arrays (mk_exons.cpp) ----
struct mk_predictions {
    std::vector<mk_prediction> preds;
    std::vector<uint64_t> contigOff;
    std::vector<mk_exon> exons;
};

void mk_default_exon_params(mk_exon_params *p) { if (p) mk::default_exon_params(*p); }

int mk_predict_exons(const mk_targetdb *db, const mk_orfs *orfs, const mk_queries *q, const mk_exon_params *P, const uint32_t *targetKeys, mk_predictions **out) {
    if (!db || !orfs || !q || !P || !out) return fail(MK_ERR_ARG, "null argument");
    if (!q->haveAln) return fail(MK_ERR_ARG, "no alignment result in this batch: run mk_search or mk_align first");
    if (q->n != orfs->orfs.size()) return fail(MK_ERR_ARG, "the batch has %u queries, the ORF set %zu fragments", q->n, orfs->orfs.size());
    HostTimer ht("host_predict_exons_total");
    mk_predictions *p = new mk_predictions();
    mk::predict_exons(orfs->orfs.data(), orfs->orfs.size(), orfs->nContigs, (const mk_alignment *) q->alns.p, q->alnOff.data(), targetKeys, mk_targetdb_residues(db), *P,
                      p->preds, p->contigOff, p->exons);
    *out = p;
    return MK_OK;
}

// the same on caller-owned arrays (no batch handle, no GPU): orfs[k] = fragment k, alns[aln_offsets[k] .. aln_offsets[k+1]) its alignments
int mk_predict_exons_arrays(const mk_orf *orfs, uint64_t nOrfs, uint32_t nContigs, const mk_alignment *alns, const uint64_t *alnOffsets,
                            uint64_t dbResidues, const mk_exon_params *P, const uint32_t *targetKeys, mk_predictions **out) {
    if (!orfs || !alnOffsets || !P || !out || (!alns && alnOffsets[nOrfs] > 0)) return fail(MK_ERR_ARG, "null argument");
    mk_predictions *p = new mk_predictions();
    mk::predict_exons(orfs, nOrfs, nContigs, alns, alnOffsets, targetKeys, dbResidues, *P, p->preds, p->contigOff, p->exons);
    *out = p;
    return MK_OK;
}

int mk_predictions_result(const mk_predictions *p, const mk_prediction **preds, const uint64_t **contigOff, const mk_exon **exons, uint64_t *n) {
    if (!p || !preds || !contigOff || !exons || !n) return fail(MK_ERR_ARG, "null argument");
    *preds = p->preds.data(); *contigOff = p->contigOff.data(); *exons = p->exons.data(); *n = p->preds.size();
    return MK_OK;
}

void mk_predictions_destroy(mk_predictions *p) { delete p; }
size_t mk_format_prediction_exon(char *buf, const mk_prediction *p, const mk_exon *e) { return mk::format_prediction_exon(buf, *p, *e); }

void mk_queries_destroy(mk_queries *q) { if (q && q->job) (void) mk_search_wait(q); delete q; }

// test hook: the per-residue arrays the device derived for this batch (kmer threshold per k-mer start,
// int8 diagonal correction, int8 SW composition bias)
int mk_queries_derived(const mk_queries *q, int16_t *kmer_thr, int8_t *diag_corr, int8_t *sw_bias8) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!q || !kmer_thr || !diag_corr || !sw_bias8) return fail(MK_ERR_ARG, "null argument");
    const size_t total = q->off[q->n];
    if (total == 0) return MK_OK;
    HIPCHK(hipMemcpy(kmer_thr, q->dKmerThr.p, total * sizeof(int16_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(diag_corr, q->dCorr.p, total, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sw_bias8, q->dBias8.p, total, hipMemcpyDeviceToHost));
    return MK_OK;
}

// forward + (optionally) reverse pass on explicit pairs: ssw_align_private<SEQ_SEQ> (StripedSmithWaterman.cpp:309-545)
int mk_sw_pairs(mk_targetdb *db, mk_queries *q, const mk_params *P, const uint32_t *qIdx, const uint32_t *tIdx,
                uint64_t n, int withStart, int32_t *out5) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!db || !q || !P || !out5) return fail(MK_ERR_ARG, "null argument");
    std::vector<mk::SwJob> jobs(n);
    for (uint64_t p = 0; p < n; p++) {
        if (qIdx[p] >= q->n || tIdx[p] >= db->n) return fail(MK_ERR_ARG, "pair %llu out of range", (unsigned long long) p);
        mk::SwJob &j = jobs[p];
        j.q_start = (uint32_t) q->off[qIdx[p]]; j.q_len = (uint32_t) (q->off[qIdx[p] + 1] - q->off[qIdx[p]]);
        j.t_start = db->off[tIdx[p]]; j.t_len = (uint32_t) (db->off[tIdx[p] + 1] - db->off[tIdx[p]]);
        j.q_step = 1; j.t_step = 1; j.slot = (uint32_t) p;
    }
    std::vector<mk::SwOut> fwd, rev;
    rc = run_sw_jobs(db, q, P, jobs, fwd, "sw_fwd");
    if (rc) return rc;
    std::vector<mk::SwJob> rj;
    std::vector<uint64_t> rp;
    for (uint64_t p = 0; p < n; p++) {
        out5[p * 5 + 0] = fwd[p].score; out5[p * 5 + 1] = fwd[p].end_row; out5[p * 5 + 2] = fwd[p].end_col;
        out5[p * 5 + 3] = -1; out5[p * 5 + 4] = -1;
        if (withStart && fwd[p].score > 0) {
            mk::SwJob j;
            j.q_len = (uint32_t) fwd[p].end_row + 1; j.t_len = (uint32_t) fwd[p].end_col + 1;
            j.q_start = jobs[p].q_start + (uint32_t) fwd[p].end_row; j.q_step = -1;
            j.t_start = jobs[p].t_start + (uint64_t) fwd[p].end_col; j.t_step = -1; j.slot = 0;
            rj.push_back(j); rp.push_back(p);
        }
    }
    rc = run_sw_jobs(db, q, P, rj, rev, "sw_rev");
    if (rc) return rc;
    for (size_t k = 0; k < rp.size(); k++) {
        const uint64_t p = rp[k];
        if (rev[k].score != fwd[p].score)
            return fail(MK_ERR_SW_MISMATCH, "Score of forward/backward SW differ: %d %d (pair %llu)", fwd[p].score, rev[k].score, (unsigned long long) p);
        out5[p * 5 + 3] = fwd[p].end_row - rev[k].end_row;
        out5[p * 5 + 4] = fwd[p].end_col - rev[k].end_col;
    }
    return MK_OK;
}

int mk_ungapped(mk_targetdb *db, mk_queries *q, const uint32_t *qIdx, const uint32_t *tIdx, const uint16_t *diag, uint64_t n, int32_t *outScores) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!db || !q || !outScores) return fail(MK_ERR_ARG, "null argument");
    if (db->unindexed) return fail(MK_ERR_ARG, "mk_ungapped needs a target database with masked residues (this one was made by mk_targetdb_create_sequences)");
    if (q->isProfile) return fail(MK_ERR_UNSUPPORTED, "mk_ungapped takes sequence queries (the prefilter scores profile diagonals itself)");
    std::vector<mk::UngappedJob> jobs(n);
    double bytes = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (qIdx[i] >= q->n || tIdx[i] >= db->n) return fail(MK_ERR_ARG, "candidate %llu out of range", (unsigned long long) i);
        mk::UngappedJob &j = jobs[i];
        j.q_start = (uint32_t) q->off[qIdx[i]]; j.q_len = (uint32_t) (q->off[qIdx[i] + 1] - q->off[qIdx[i]]);
        j.t_start = db->off[tIdx[i]]; j.t_len = (uint32_t) (db->off[tIdx[i] + 1] - db->off[tIdx[i]]);
        j.diagonal = diag[i];
        bytes += sizeof(mk::UngappedJob) + 4 + std::min(j.q_len, j.t_len);
    }
    DevBuf<mk::UngappedJob> dJobs;
    DevBuf<int32_t> dOut;
    HIPCHK(dJobs.upload(jobs.data(), n));
    HIPCHK(dOut.alloc(n));
    mk::UngappedLaunch L;
    L.q_res = q->dRes.p; L.q_corr = q->dCorr.p; L.t_masked = db->dMasked.p; L.mat = db->dMatUng.p;
    L.jobs = dJobs.p; L.out = dOut.p; L.n_jobs = n;
    const int th = timed_begin("ungapped", bytes, 0);
    HIPCHK(mk::launch_ungapped(L, g_stream));
    timed_end(th);
    HIPCHK(hipMemcpyAsync(outScores, dOut.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    timed_flush();
    return MK_OK;
}

static mk::PrefilterDeviceView prefilter_view(const mk_targetdb *db, const mk_queries *q) {
    mk::PrefilterDeviceView V;
    V.q_res = q->dRes.p; V.q_off = q->dOff.p; V.q_kmer_thr = q->dKmerThr.p; V.q_corr = q->dCorr.p; V.n_queries = q->n;
    V.t_masked = db->dMasked.p; V.t_off = db->dOff.p; V.n_targets = db->n;
    V.kmer_slot = db->dKmerSlot.p; V.kmer_bits = db->dKmerBits.p; V.entries = db->dEntries.p - db->entryShift; V.score3 = db->dScore3.p; V.index3 = db->dIndex3.p;
    V.hist3 = db->dHist3.p; V.cum3 = db->dCum3.p; V.hist_lo = db->histLo; V.hist_range = db->histRange; V.n_entries = db->nEntries;
    V.mat_ung = db->dMatUng.p;
    if (q->isProfile) { V.p_sorted = q->dProfSorted.p; V.p_aln = q->dProfAln.p; V.addr3 = db->dAddr3.p; }
    V.kmer_size = db->kmerSize;
    if (db->kmerSize == 7) { V.score2 = db->dScore2.p; V.index2 = db->dIndex2.p; V.num3 = db->dNum3.p; }
    return V;
}

// the k-mer thresholds of a sequence batch belong to one k-mer size (seed pattern, threshold formula): a batch that meets a database of
// the other size (k chosen from the database's residue count, IndexTable.h:439-449) has them derived again
static int match_kmer_size(const mk_targetdb *db, mk_queries *q) {
    if (q->kmerSize == db->kmerSize) return MK_OK;
    const mk_params &P = q->derivedWith;
    if (q->isProfile) {          // the k-mer starts of a profile batch follow the seed pattern and threshold of the database's k
        const uint64_t total = q->off[q->n];
        HIPCHK(mk::launch_profile_kthr(q->dRes.p, q->dOff.p, q->n, total, db->kmerSize == 7 ? mk::kmer_threshold_profile_k7(P.sensitivity) : mk::kmer_threshold_profile(P.sensitivity),
                                       db->kmerSize, q->dKmerThr.p, cur_stream()));
        HIPCHK(hipStreamSynchronize(cur_stream()));
        q->kmerSize = db->kmerSize;
        return MK_OK;
    }
    mk::SubMat kmerMat, alnMat;
    mk::build_submat(kmerMat, mk::MAT_VTML80, 8.0f, -0.2f);
    mk::build_submat(alnMat, mk::MAT_BLOSUM62, 2.0f, 0.0f);
    const uint64_t total = q->off[q->n];
    HIPCHK(mk::launch_derive(q->dRes.p, q->dOff.p, q->n, total, kmerMat, alnMat,
                             db->kmerSize == 7 ? mk::kmer_threshold_k7(P.sensitivity, P.kmer_score) : mk::kmer_threshold(P.sensitivity, P.kmer_score),
                             P.comp_bias_corr != 0, P.comp_bias_scale, q->dKmerThr.p, q->dCorr.p, q->dBias8.p, cur_stream(), db->kmerSize));
    HIPCHK(hipStreamSynchronize(cur_stream()));
    q->kmerSize = db->kmerSize;
    return MK_OK;
}

// a profile batch searches a target side built for it (and only that one)
static int check_indexed(const mk_targetdb *db, const char *what) {
    if (db->unindexed) return fail(MK_ERR_ARG, "%s needs a target database with a k-mer index: this one was made by mk_targetdb_create_sequences (residues only, for mk_align)", what);
    return MK_OK;
}
static int check_roles(const mk_targetdb *db, const mk_queries *q) {
    if (db->unindexed) return MK_OK;                    // (residues serve either kind of query)
    if (q->isProfile != db->profileSearch)
        return fail(MK_ERR_ARG, q->isProfile ? "profile queries need a target database created with params->profile_search = 1"
                                             : "this target database was created for profile queries (params->profile_search = 1)");
    return MK_OK;
}

int mk_prefilter(mk_targetdb *db, mk_queries *q, const mk_params *P) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!db || !q || !P) return fail(MK_ERR_ARG, "null argument");
    if ((rc = check_indexed(db, "mk_prefilter")) != MK_OK) return rc;
    if ((rc = check_roles(db, q)) != MK_OK) return rc;
    if ((rc = match_kmer_size(db, q)) != MK_OK) return rc;
    std::string err;
    const int binCount = mk::bin_count_for(db->n, P->host_l2_bytes);
    {
        HostTimer ht("host_prefilter_total");
        mk::PrefilterHooks hooks;
        hooks.t_masked_host = [db]() { return masked_host(db); };
        q->pfStats = mk::PrefilterStats();
        hooks.stats = &q->pfStats;
        rc = mk::run_prefilter(prefilter_view(db, q), q->off, q->res, nullptr, db->off, *P, binCount, g_stream, q->hits, q->nHits, q->hitOff, err,
                               timed_begin, timed_end, timed_set, hooks);
    }
    timed_flush();
    if (rc != MK_OK) return fail(rc, "%s", err.c_str());
    q->havePref = true;
    return MK_OK;
}

// experiment hook (not part of the documented ABI; tools/micro/mk_experiments.hip is its only user): the device view of a (database,
// batch) pair -- pointers into HBM, valid while both handles live -- and the batch's host offsets
int mk_debug_prefilter_view(mk_targetdb *db, mk_queries *q, void *viewOut, size_t viewBytes, const uint64_t **qOffHost) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!db || !q || !viewOut || viewBytes != sizeof(mk::PrefilterDeviceView)) return fail(MK_ERR_ARG, "bad argument (view of %zu bytes expected)", sizeof(mk::PrefilterDeviceView));
    if ((rc = check_indexed(db, "mk_debug_prefilter_view")) != MK_OK) return rc;
    if ((rc = match_kmer_size(db, q)) != MK_OK) return rc;
    const mk::PrefilterDeviceView V = prefilter_view(db, q);
    std::memcpy(viewOut, &V, sizeof(V));
    if (qOffHost) *qOffHost = q->off.data();
    return MK_OK;
}

// the reference's prefilter statistics (Prefiltering.cpp:889-904 -> printStatistics :953-975) of the last mk_prefilter / mk_search over the batch
int mk_prefilter_statistics(const mk_queries *q, mk_prefilter_stats *out) {
    if (!q || !out) return fail(MK_ERR_ARG, "null argument");
    if (!q->havePref) return fail(MK_ERR_ARG, "no prefilter result in this batch");
    const uint64_t n = q->n;
    std::memset(out, 0, sizeof(*out));
    out->n_queries = n;
    if (n == 0) return MK_OK;
    out->kmers_per_pos = q->pfStats.kmers_per_pos / static_cast<double>(n);
    out->db_matches_per_seq = q->pfStats.db_matches / n;
    out->overflows = q->pfStats.overflows;
    out->results_per_seq = q->hitOff[n] / n;
    std::vector<uint32_t> lens(n);
    uint64_t empty = 0;
    for (uint64_t i = 0; i < n; i++) { lens[i] = (uint32_t) (q->hitOff[i + 1] - q->hitOff[i]); empty += lens[i] == 0; }
    std::nth_element(lens.begin(), lens.begin() + n / 2, lens.end());
    out->median_result_len = lens[n / 2];
    out->empty_results = empty;
    return MK_OK;
}
// "246.638184 k-mers per position\n12 DB matches per sequence\n..." as Prefiltering::printStatistics writes it
size_t mk_format_prefilter_statistics(char *buf, size_t cap, const mk_prefilter_stats *s, uint64_t max_results) {
    if (!buf || !s) return 0;
    const int n = snprintf(buf, cap, "\n%f k-mers per position\n%llu DB matches per sequence\n%llu overflows\n%llu sequences passed prefiltering per query sequence%s\n"
                                     "%u median result list length\n%llu sequences with 0 size result lists\n",
                           s->kmers_per_pos, (unsigned long long) s->db_matches_per_seq, (unsigned long long) s->overflows, (unsigned long long) s->results_per_seq,
                           s->results_per_seq > max_results ? " (ATTENTION: max. results were written to the output prefiltering database)" : "",
                           s->median_result_len, (unsigned long long) s->empty_results);
    return n < 0 || (size_t) n >= cap ? 0 : (size_t) n;
}

int mk_prefilter_result(const mk_queries *q, const mk_hit **hits, const uint64_t **offsets) {
    if (!q || !hits || !offsets) return fail(MK_ERR_ARG, "null argument");
    if (!q->havePref) return fail(MK_ERR_ARG, "no prefilter result in this batch");
    *hits = (const mk_hit *) q->hits.p; *offsets = q->hitOff.data();
    return MK_OK;
}

int mk_prefilter_result_set(mk_queries *q, const mk_hit *hits, const uint64_t *offsets) {
    if (!q || !offsets || (!hits && offsets[q->n] > 0)) return fail(MK_ERR_ARG, "null argument");
    for (uint32_t i = 0; i < q->n; i++) if (offsets[i + 1] < offsets[i]) return fail(MK_ERR_ARG, "offsets are not ascending at query %u", i);
    if (offsets[0] != 0) return fail(MK_ERR_ARG, "offsets[0] must be 0");
    q->hitOff.assign(offsets, offsets + q->n + 1);
    q->nHits = offsets[q->n];
    q->pfStats = mk::PrefilterStats();                  // an installed result has no k-mer / index-match counters (they belonged to the run that made it)
    if (!q->hits.reserve(std::max<size_t>(q->nHits, 1) * sizeof(mk_hit), 0)) return fail(MK_ERR_DEVICE, "pinned host allocation failed");
    if (q->nHits) std::memcpy(q->hits.p, hits, q->nHits * sizeof(mk_hit));
    q->havePref = true;
    return MK_OK;
}

// Alignment::run over the queries [q0, q1) (Alignment.cpp:312-514): SW on the device, then Matcher::getSWResult's
// float/double tail (Matcher.cpp:60-142), Alignment::checkCriteria (:548-567) and the per-query sort (:403-405).
// Appends the accepted alignments at alns[nAlnOut...] and fills alnOff[q0+1 .. q1].
// (wait_turn: with several workers on one batch, called before anything of the batch's result block is touched -- chunks commit in order)
static int align_range(mk_targetdb *db, mk_queries *q, const mk_params *P, uint32_t q0, uint32_t q1, const std::vector<mk::GateEntry> &gate,
                       const mk::AssembleTables *tables, hipStream_t stream, size_t &nAlnOut, const std::function<void()> &wait_turn = nullptr) {
    const uint32_t nqc = q1 - q0;
    const uint64_t h0 = q->hitOff[q0];
    const size_t n = (size_t) (q->hitOff[q1] - h0);
    const mk_hit *hits = (const mk_hit *) q->hits.p + h0;
    std::vector<uint64_t> hitOff((size_t) nqc + 1);            // pair offsets of the range
    for (uint32_t i = 0; i <= nqc; i++) hitOff[i] = q->hitOff[(size_t) q0 + i] - h0;
    mk::AlignView V = align_view(db, q);
    V.q_off = q->dOff.p + q0; V.n_queries = nqc;
    V.co_resident = stream != g_stream;             // the alignment stage of mk_search runs beside the prefilter
    const mk::AlnRaw *raw = nullptr;
    size_t m = 0;
    std::string err;
    // the float/double tail of getSWResult, the criteria and the per-query order run on the device; the host path below only
    // serves a range in which a score lies beyond the e-value table
    mk::AssembleArgs asmArgs;
    static const bool hostAssemble = mk::knob_long("MK_ALIGN_HOST_ASSEMBLE", 0) != 0;
    asmArgs.tables = hostAssemble ? nullptr : tables;
    asmArgs.dSortKey = db->dKeys.p;
    asmArgs.counts = (uint32_t *) mk::pinned_scratch(stream == g_stream2 ? "asm_counts_h2" : "asm_counts_h", std::max<size_t>(nqc, 1) * 4);
    if (!asmArgs.counts) return fail(MK_ERR_DEVICE, "pinned host allocation failed");
    bool turned = false;
    const auto turn = [&]() { if (!turned && wait_turn) wait_turn(); turned = true; };
    asmArgs.reserve = [&](size_t cnt) -> mk_alignment * {
        turn();
        if (!q->alns.reserve(std::max<size_t>(nAlnOut + cnt, 1) * sizeof(mk_alignment), nAlnOut * sizeof(mk_alignment))) return nullptr;
        return (mk_alignment *) q->alns.p + nAlnOut;
    };
    int rc = mk::run_align_device(V, hitOff.data(), hits, n, gate, *P, stream, nullptr, &raw, &m, err, timed_begin, timed_end, timed_set, &asmArgs);
    timed_flush();
    if (rc != MK_OK) return fail(rc, "%s", err.c_str());
    turn();
    if (asmArgs.done) {
        uint64_t o = 0;
        for (uint32_t i = 0; i < nqc; i++) { o += asmArgs.counts[i]; q->alnOff[(size_t) q0 + i + 1] = nAlnOut + o; }
        nAlnOut += asmArgs.nOut;
        return MK_OK;
    }
    HostTimer ht("host_align_assemble");
    // raw is ordered by pair index == by query: query i owns raw[first[i] .. first[i+1])
    std::vector<uint64_t> first((size_t) nqc + 1);
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i <= nqc; i++) {
        const uint64_t want = i < nqc ? hitOff[i] : (uint64_t) n;
        size_t lo = 0, hi = m;                         // first record with pair >= want
        while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (raw[mid].pair < want) lo = mid + 1; else hi = mid; }
        first[i] = lo;
    }
    // records are written at their upper-bound position; rejected ones (rare) leave holes that are closed afterwards
    if (!q->alns.reserve(std::max<size_t>(nAlnOut + m, 1) * sizeof(mk_alignment), nAlnOut * sizeof(mk_alignment))) return fail(MK_ERR_DEVICE, "pinned host allocation failed");
    mk_alignment *alns = (mk_alignment *) q->alns.p + nAlnOut;
    std::vector<uint32_t> cnt(nqc);
    int mismatch = 0;
    uint64_t holes = 0;
#pragma omp parallel for schedule(dynamic, 512) reduction(+ : mismatch, holes)
    for (uint32_t i = 0; i < nqc; i++) {
        const int qLen = (int) (q->off[(size_t) q0 + i + 1] - q->off[(size_t) q0 + i]);
        const uint64_t begin = first[i];
        uint64_t w = begin;
        for (uint64_t k = first[i]; k < first[i + 1]; k++) {
            const mk::AlnRaw &r = raw[k];
            if (r.q_start == -2) { mismatch++; continue; }
            const uint32_t t = hits[r.pair].seq_id;
            const int tLen = (int) (db->off[t + 1] - db->off[t]);
            mk_alignment a;
            a.db_key = t; a.q_len = qLen; a.db_len = tLen; a.raw_score = r.score;
            a.evalue = db->evaluer.evalue((double) r.score, (double) qLen);
            a.qcov = mk::compute_cov((unsigned) r.q_start, (unsigned) r.q_end, (unsigned) qLen);
            a.dbcov = mk::compute_cov((unsigned) r.t_start, (unsigned) r.t_end, (unsigned) tLen);
            a.q_start = r.q_start; a.q_end = r.q_end; a.db_start = r.t_start; a.db_end = r.t_end;
            a.aln_len = std::max(std::abs(r.q_end - r.q_start), std::abs(r.t_end - r.t_start)) + 1;
            const unsigned int qAln = std::max((unsigned) r.q_end - (unsigned) r.q_start, 1u);
            const unsigned int dbAln = std::max((unsigned) r.t_end - (unsigned) r.t_start, 1u);
            const uint16_t s16 = (uint16_t) r.score;
            float sid = (s16 / static_cast<float>(std::max(qAln, dbAln))) * 0.1656 + 0.1141;   // Matcher.cpp:160-164
            sid = std::min(sid, 1.0f);
            a.seq_id = std::max(0.0f, sid);
            a.bit_score = static_cast<int>(db->evaluer.bitScore((double) r.score) + 0.5);
            if (a.evalue <= P->evalue_thr && a.aln_len >= P->min_aln_len) alns[w++] = a;
        }
        if (w - begin > 1) {
            if (db->keys.empty()) std::sort(alns + begin, alns + w, mk::alignment_less);
            else std::sort(alns + begin, alns + w, [&](const mk_alignment &x, const mk_alignment &y) {      // Matcher::compareHits ties on the DB key
                mk_alignment kx = x, ky = y; kx.db_key = db->keys[x.db_key]; ky.db_key = db->keys[y.db_key];
                return mk::alignment_less(kx, ky);
            });
        }
        cnt[i] = (uint32_t) (w - begin);
        holes += first[i + 1] - w;
    }
    if (mismatch) return fail(MK_ERR_SW_MISMATCH, "Score of forward/backward SW differ for %d pairs", mismatch);
    uint64_t o = 0;                                    // range-local offsets
    for (uint32_t i = 0; i < nqc; i++) {
        if (holes && cnt[i] && o != first[i]) std::memmove(alns + o, alns + first[i], (size_t) cnt[i] * sizeof(mk_alignment));   // blocks only move forward
        o += cnt[i];
        q->alnOff[(size_t) q0 + i + 1] = nAlnOut + o;
    }
    nAlnOut += o;
    return MK_OK;
}

int mk_align(mk_targetdb *db, mk_queries *q, const mk_params *P) {
    int rc = ensure_ready();
    if (rc) return rc;
    if (!db || !q || !P) return fail(MK_ERR_ARG, "null argument");
    if (!q->havePref) return fail(MK_ERR_ARG, "mk_align: the batch has no prefilter result");
    if ((rc = check_roles(db, q)) != MK_OK) return rc;
    HostTimer htAll("host_align_total");
    std::shared_ptr<const ScoreTabs> tabs;
    {
        HostTimer ht("host_gate_table");
        tabs = score_tables(db, q, P->evalue_thr);
    }
    q->alnOff.assign((size_t) q->n + 1, 0);
    size_t nAln = 0;
    rc = align_range(db, q, P, 0, q->n, tabs->gate, &tabs->tables, g_stream, nAln);
    if (rc != MK_OK) return rc;
    q->haveAln = true;
    return MK_OK;
}

// ---- the search engine --------------------------------------------------------------------------------------------------------
// prefilter + align of a batch as ONE pipelined pass (the reference's `search` workflow runs the two modules back to back,
// blastp.sh:70,85): the prefilter works through the queries in chunks on one stream; every finished chunk is aligned on another
// stream by another host thread while the prefilter is already on the next chunk.  The prefilter is bound by memory latency and the
// Smith-Waterman kernels by the vector ALUs, so the two stages share the GPU.  Results are identical to mk_prefilter followed by mk_align.
//
// Round 5: the threads are PERSISTENT and the batches QUEUE (mk_search_begin / mk_search_wait): one prefilter thread takes the batches
// in the order they were begun, the alignment workers take the chunks of all batches in that order.  The prefilter of batch k + 1 then
// runs beside the alignment of the last chunks of batch k, and the alignment of its first chunk starts the moment the workers are free:
// the fill and drain of the two-stage pipeline (2 x 8 ms of a 133 ms query shard, profiles/r04_shard_sweep.txt) are paid once per run,
// not once per batch.  mk_search = begin + wait.
struct SearchJob {
    mk_targetdb *db = nullptr; mk_queries *q = nullptr; mk_params P;
    std::shared_ptr<const ScoreTabs> tabs;
    int hostThreads = 1;                                 // OpenMP threads of the caller (mk_init sizes them): split between the stages
    uint32_t pushed = 0, turn = 0;                       // chunks handed over; the chunk whose alignments are appended next
    int outstanding = 0;                                 // chunks queued or being aligned
    bool prefilterDone = false, finished = false;
    int rc = MK_OK; std::string err;
    size_t nAln = 0;
    double tStart = 0;
};

namespace {

struct SearchEngine {
    struct Item { SearchJob *job; uint32_t seq, q0, q1; };
    std::mutex m; std::condition_variable cv;
    std::deque<SearchJob *> batches;                     // begun, not yet through the prefilter
    std::deque<Item> items;                              // finished prefilter chunks of all batches, in order
    int unfinished = 0;                                  // batches begun whose results are not complete yet
    bool started = false;
    bool stop = false;                                   // (under m) mk_shutdown / process exit: the threads leave their loops
    std::vector<std::thread> threads;
    int nWorkers = 3;
    int nPrefilter = 1;                                  // MK_PREFILTER_THREADS = 2: the prefilters of two batches at once.  Measured and NOT the default
                                                         // (profiles/r05_search_engine.txt: 923 against 880 ms per config-2 step): the GPU is saturated by one
                                                         // prefilter chain and the alignment workers, two chains only stretch each other
    std::mutex tablesMutex;                              // the databases' score-table caches
};
SearchEngine *g_engine = nullptr;                        // its threads sleep on the condition variable until mk_shutdown or the end of the process
std::mutex g_engineMutex;                                // creation and shutdown of the engine

void engine_finish_locked(SearchEngine &E, SearchJob *job) {      // (E.m held) prefilter done and no chunk outstanding
    if (job->finished || !job->prefilterDone || job->outstanding != 0) return;
    job->finished = true;
    E.unfinished--;
    mk::host_stat("host_search_total", mk::ScopedHost::now_ms() - job->tStart);
}

void engine_prefilter_thread(SearchEngine *Ep, int idx) {
    SearchEngine &E = *Ep;
    (void) hipSetDevice(g_device);
    kmp_set_blocktime(0);
    // thread 0 works on the library's first stream with the default scratch lane (what the blocking mk_prefilter uses, which waits for the engine
    // to drain first); thread 1 has a stream and a lane of its own
    hipStream_t myStream = idx == 0 ? g_stream : g_prefStream2;
    if (idx != 0) { t_stream = myStream; mk::set_scratch_lane(PREFILTER2_LANE); }
    for (;;) {
        SearchJob *job;
        {
            std::unique_lock<std::mutex> lk(E.m);
            E.cv.wait(lk, [&] { return E.stop || !E.batches.empty(); });
            if (E.batches.empty()) return;                 // (stop: only once nothing is queued)
            job = E.batches.front(); E.batches.pop_front();
        }
        mk_targetdb *db = job->db; mk_queries *q = job->q; const mk_params *P = &job->P;
        job->tStart = mk::ScopedHost::now_ms();
        const int half = std::max(1, job->hostThreads / (E.nPrefilter + E.nWorkers));
        omp_set_num_threads(half);
        int rc = match_kmer_size(db, q);
        std::string err = rc != MK_OK ? g_err : std::string();
        static const int profilePipe = (int) mk::knob_long("MK_SEARCH_PROFILE_PIPELINE", 1);
        if (rc == MK_OK) {
            HostTimer ht("host_gate_table");       // the per-length score tables of the batch (cached per database and query lengths)
            std::lock_guard<std::mutex> lk(E.tablesMutex);
            job->tabs = score_tables(db, q, P->evalue_thr);
        }
        q->alnOff.assign((size_t) q->n + 1, 0);
        mk::PrefilterHooks hooks;
        // chunks of 262 144 queries, the first ones smaller (65 536, 131 072: the alignment stage starts when the first chunk is done).  Measured on
        // config 2 (profiles/r03_search_tuning.txt): 131 072-query chunks 1.095 s per step, 262 144 with the ramp 0.99 s -- the position and reverse
        // passes are launches per tile configuration and chunk, and short launches beside the persistent prefilter workgroups run at a third of
        // their speed.  A shard of that batch (1/8 of it on an 8-GPU node: 251 k queries) would be three chunks, with next to nothing for the two
        // stages to overlap: the chunk follows the batch -- the largest power of two below a sixth of it, 65 536 at least (32 768-query chunks are
        // too many short launches beside the persistent workgroups: profiles/r04_shard_sweep.txt)
        {
            uint32_t lim = 1u << 18;
            while (lim > (1u << 16) && (uint64_t) lim * 6 > (uint64_t) q->n) lim >>= 1;
            hooks.max_chunk_queries = lim;
        }
        // the ramp (first chunks of 1/4 and 1/2 of the limit) lets the alignment stage start early on a batch that arrives at an idle engine; when
        // another batch is still in flight the workers are busy with its tail anyway and the small chunks only cost launches: 120.5 -> 111.6 ms per
        // 251 601-fragment shard, 231 -> 215-221 ms at 502 351, 836.8 -> 831.1 ms for the full batch (queued, profiles/r05_search_engine.txt)
        {
            std::lock_guard<std::mutex> lk(E.m);
            hooks.chunk_ramp = E.unfinished <= 1;
        }
        hooks.co_resident = true;
        hooks.t_masked_host = [db]() { return masked_host(db); };
        q->pfStats = mk::PrefilterStats();
        hooks.stats = &q->pfStats;
        if (q->isProfile) hooks.max_chunk_queries = (uint32_t) std::max(64L, mk::knob_long("MK_SEARCH_PROFILE_CHUNK", 8192));
        if (const char *e = mk::knob("MK_SEARCH_CHUNK_QUERIES")) hooks.max_chunk_queries = (uint32_t) std::max(1024L, atol(e));
        if (const char *e = mk::knob("MK_SEARCH_CHUNK_RAMP")) hooks.chunk_ramp = atoi(e) != 0;
        // the last chunk can be dealt out in pieces: its alignment is the tail of the pass and every worker takes a share of it
        static const int tailPieces = (int) std::max(1L, mk::knob_long("MK_ALIGN_TAIL_PIECES", 1));
        const bool backToBack = q->isProfile && !profilePipe;        // MK_SEARCH_PROFILE_PIPELINE=0: the whole batch is one chunk for the alignment stage
        hooks.on_chunk = [&](uint32_t a, uint32_t b) {
            if (backToBack && b != q->n) return;
            if (backToBack) a = 0;
            {
                std::lock_guard<std::mutex> lk(E.m);
                const uint32_t pieces = (b == q->n && b - a >= 4096u) ? (uint32_t) tailPieces : 1u;
                for (uint32_t k = 0; k < pieces; k++) {
                    const uint32_t lo = a + (uint32_t) ((uint64_t) (b - a) * k / pieces), hi = a + (uint32_t) ((uint64_t) (b - a) * (k + 1) / pieces);
                    if (hi > lo) { E.items.push_back(SearchEngine::Item{job, job->pushed++, lo, hi}); job->outstanding++; }
                }
            }
            E.cv.notify_all();
        };
        hooks.before_grow = [&]() {                                // the result block moves: nobody may be reading it
            std::unique_lock<std::mutex> lk(E.m);
            E.cv.wait(lk, [&] { return job->outstanding == 0; });
        };
        if (rc == MK_OK) {
            HostTimer ht("host_prefilter_total");
            const int binCount = mk::bin_count_for(db->n, P->host_l2_bytes);
            rc = mk::run_prefilter(prefilter_view(db, q), q->off, q->res, nullptr, db->off, *P, binCount, myStream, q->hits, q->nHits, q->hitOff, err,
                                   timed_begin, timed_end, timed_set, hooks);
        }
        timed_flush();
        {
            std::lock_guard<std::mutex> lk(E.m);
            if (rc != MK_OK && job->rc == MK_OK) { job->rc = rc; job->err = err; }
            job->prefilterDone = true;
            engine_finish_locked(E, job);
        }
        E.cv.notify_all();
    }
}

// the alignment stage: MK_ALIGN_WORKERS host threads, each with a stream and scratch buffers of its own, take the chunks in turn -- the
// position and reverse passes of one chunk (short launches, slow beside persistent workgroups) then run beside the forward pass of the
// next one instead of in front of it.  Results are appended in chunk order.  Three workers since the position / reverse passes became
// short (profiles/r04_sw_early_exit.txt: 926 -> 910 ms per step against two; the same within the noise before).
void engine_align_thread(SearchEngine *Ep, int w) {
    SearchEngine &E = *Ep;
    t_stream = g_alignStreams[w];
    mk::set_scratch_lane(w);
    (void) hipSetDevice(g_device);
    kmp_set_blocktime(0);
    for (;;) {
        SearchEngine::Item it;
        {
            std::unique_lock<std::mutex> lk(E.m);
            E.cv.wait(lk, [&] { return E.stop || !E.items.empty(); });
            if (E.items.empty()) return;
            it = E.items.front(); E.items.pop_front();
        }
        SearchJob *job = it.job;
        omp_set_num_threads(std::max(1, job->hostThreads / (E.nPrefilter + E.nWorkers)));
        int r = MK_OK;
        const auto wait_turn = [&]() {
            std::unique_lock<std::mutex> lk(E.m);
            E.cv.wait(lk, [&] { return job->turn == it.seq; });
        };
        bool healthy;
        { std::lock_guard<std::mutex> lk(E.m); healthy = job->rc == MK_OK; }
        if (healthy) {
            HostTimer ht("host_align_total");
            r = align_range(job->db, job->q, &job->P, it.q0, it.q1, job->tabs->gate, &job->tabs->tables, g_alignStreams[w], job->nAln, wait_turn);
        }
        wait_turn();
        {
            std::lock_guard<std::mutex> lk(E.m);
            if (r != MK_OK && job->rc == MK_OK) { job->rc = r; job->err = g_err; }
            job->outstanding--;
            job->turn = it.seq + 1;
            engine_finish_locked(E, job);
        }
        E.cv.notify_all();
    }
}

// Drains the engine (every batch begun is finished), stops and joins its threads.  Registered with atexit when the engine starts -- a host that
// exits between mk_search_begin and mk_search_wait (an exception, an interpreter shutting down) must not leave threads behind that use the
// statistics map, the scratch buffers and the HIP runtime while the process tears them down -- and callable as mk_shutdown.
void engine_shutdown() {
    SearchEngine *E;
    {
        std::lock_guard<std::mutex> g(g_engineMutex);
        E = g_engine;
        g_engine = nullptr;
    }
    if (!E) return;
    {
        std::unique_lock<std::mutex> lk(E->m);
        E->cv.wait(lk, [&] { return E->unfinished == 0; });
        E->stop = true;
    }
    E->cv.notify_all();
    for (std::thread &t : E->threads) if (t.joinable()) t.join();
    delete E;
}

SearchEngine *engine_ptr() {
    std::lock_guard<std::mutex> g(g_engineMutex);
    if (!g_engine) {
        SearchEngine *E = new SearchEngine();
        E->nWorkers = std::min(MAX_ALIGN_WORKERS, std::max(1, (int) mk::knob_long("MK_ALIGN_WORKERS", 3)));
        E->nPrefilter = std::min(2, std::max(1, (int) mk::knob_long("MK_PREFILTER_THREADS", 1)));
        for (int k = 0; k < E->nPrefilter; k++) E->threads.emplace_back(engine_prefilter_thread, E, k);
        for (int w = 0; w < E->nWorkers; w++) E->threads.emplace_back(engine_align_thread, E, w);
        E->started = true;
        static bool registered = false;
        if (!registered) { registered = true; atexit(engine_shutdown); }
        g_engine = E;
    }
    return g_engine;
}

// every batch begun so far has its results complete (not necessarily collected): what the blocking entry points wait for before they
// use the library's streams and scratch buffers themselves
void engine_drain_if_running() {
    SearchEngine *E;
    { std::lock_guard<std::mutex> g(g_engineMutex); E = g_engine; }
    if (!E) return;
    std::unique_lock<std::mutex> lk(E->m);
    E->cv.wait(lk, [&] { return E->unfinished == 0; });
}

}  // namespace

int mk_search_begin(mk_targetdb *db, mk_queries *q, const mk_params *P) {
    int rc = ensure_ready(false);
    if (rc) return rc;
    if (!db || !q || !P) return fail(MK_ERR_ARG, "null argument");
    if (q->job) return fail(MK_ERR_ARG, "mk_search_begin: a search of this batch is in flight (mk_search_wait collects it)");
    if ((rc = check_indexed(db, "mk_search")) != MK_OK) return rc;
    if ((rc = check_roles(db, q)) != MK_OK) return rc;
    SearchJob *job = new SearchJob();
    job->db = db; job->q = q; job->P = *P;
    job->hostThreads = omp_get_max_threads();
    q->havePref = false; q->haveAln = false;
    q->job = job;
    SearchEngine &E = *engine_ptr();
    {
        std::lock_guard<std::mutex> lk(E.m);
        E.batches.push_back(job);
        E.unfinished++;
    }
    E.cv.notify_all();
    return MK_OK;
}

int mk_search_wait(mk_queries *q) {
    if (!q) return fail(MK_ERR_ARG, "null argument");
    SearchJob *job = q->job;
    if (!job) return fail(MK_ERR_ARG, "mk_search_wait: no search of this batch is in flight");
    SearchEngine &E = *engine_ptr();
    {
        std::unique_lock<std::mutex> lk(E.m);
        E.cv.wait(lk, [&] { return job->finished; });
    }
    const int rc = job->rc;
    if (rc != MK_OK) g_err = job->err;
    else {
        q->havePref = true; q->haveAln = true;
        if (q->n > 0 && !q->isProfile) { g_hitsPerQuery.store((double) q->nHits / (double) q->n); g_alnsPerQuery.store((double) q->alnOff[q->n] / (double) q->n); }
    }
    q->job = nullptr;
    delete job;
    return rc;
}

void mk_shutdown(void) {
    engine_shutdown();
    // ... and the memory the library keeps between calls goes back: the device blocks of the query batches, the persistent device and pinned scratch
    // of the stages, the pinned result blocks (a host that is done searching, or about to hand the GPU to another process, gets the HBM back; the
    // next call allocates again).  Databases and batches the caller still holds are untouched
    if (g_ready) {
        (void) hipDeviceSynchronize();
        dev_pool_trim();
        mk::scratch_release_all();
    }
}

int mk_search(mk_targetdb *db, mk_queries *q, const mk_params *P) {
    const int rc = mk_search_begin(db, q, P);
    if (rc != MK_OK) return rc;
    return mk_search_wait(q);
}

int mk_align_result(const mk_queries *q, const mk_alignment **alns, const uint64_t **offsets) {
    if (!q || !alns || !offsets) return fail(MK_ERR_ARG, "null argument");
    if (!q->haveAln) return fail(MK_ERR_ARG, "no alignment result in this batch");
    *alns = (const mk_alignment *) q->alns.p; *offsets = q->alnOff.data();
    return MK_OK;
}

int mk_kernel_stats(mk_kernel_stat *out, int cap) {
    std::lock_guard<std::mutex> g(g_statsMutex);
    int k = 0;
    g_statNames.clear();
    for (auto &kv : g_stats) g_statNames.push_back(kv.first);
    for (auto &kv : g_stats) {
        if (k >= cap) break;
        out[k].name = g_statNames[k].c_str();
        out[k].ms = kv.second.ms; out[k].launches = kv.second.launches; out[k].alg_bytes = kv.second.bytes; out[k].cells = kv.second.cells;
        k++;
    }
    return k;
}
void mk_kernel_stats_reset(void) { std::lock_guard<std::mutex> g(g_statsMutex); g_stats.clear(); }

size_t mk_format_hit(char *buf, uint32_t key, int32_t score, uint16_t diag) { return mk::format_hit(buf, key, score, diag); }
size_t mk_format_alignment(char *buf, const mk_alignment *a) { return mk::format_alignment(buf, *a); }

size_t mk_format_hits(char *buf, size_t cap, const mk_hit *hits, uint64_t n, const uint32_t *targetKeys) {
    return format_bulk(buf, cap, hits, n, 32, [&](char *w, const mk_hit &h) { return mk::format_hit(w, targetKeys ? targetKeys[h.seq_id] : h.seq_id, h.pref_score, h.diagonal); });
}
size_t mk_format_alignments(char *buf, size_t cap, const mk_alignment *alns, uint64_t n) {
    return format_bulk(buf, cap, alns, n, 160, [&](char *w, const mk_alignment &a) { return mk::format_alignment(w, a); });
}

}  // extern "C"
