// metaeuk_amd/csrc/mk_synth.cpp -- seeded generator of the synthetic workload of SURVEY.md 8(d), native and parallel, for the tests and
// bench.py at database sizes the Python generator (metaeuk_amd/synth.py: 5 s per 100 000 proteins) cannot reach: protein families of
// ten (founder of 150 .. 600 residues drawn from the Robinson background, member j = the founder with every residue redrawn with
// probability 0.05 (1 + j)), and query fragments cut out of known targets and mutated ("planted homologs").  Every family / fragment has
// its own counter-based random stream (splitmix64 of seed and number -> xoshiro256**), so the bytes depend on the seed alone -- not on the
// thread count, and they are the same in the build container and on the GPU box.  Host code, no GPU, nothing of the search path.
#include "../../include/metaeuk_amd_debug.h"
#include <algorithm>
#include <cstring>
#include <omp.h>
#include <string>
#include <cstdio>
#include <vector>

namespace {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t stream) {
        uint64_t x = seed * 0xD1342543DE82EF95ull + stream * 0x2545F4914F6CDD1Dull + 0x1234567ull;
        for (int i = 0; i < 4; i++) s[i] = splitmix(x);
    }
    static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    uint32_t below(uint32_t n) { return (uint32_t) (((next() >> 32) * (uint64_t) n) >> 32); }
};

// Robinson & Robinson background in the alphabet order ACDEFGHIKLMNPQRSTVWY (= residue codes 0..19), as metaeuk_amd/synth.py
const double BG[20] = {0.07805, 0.01925, 0.05364, 0.06295, 0.03856, 0.07377, 0.02199, 0.05142, 0.05744, 0.09019,
                       0.02243, 0.04487, 0.05203, 0.04264, 0.05129, 0.07120, 0.05841, 0.06441, 0.01330, 0.03216};

const uint8_t *residue_table() {                       // 16 random bits -> residue
    static uint8_t table[65536];
    static bool ready = false;
    if (!ready) {
#pragma omp critical(mk_synth_table)
        if (!ready) {
            double sum = 0;
            for (double v : BG) sum += v;
            double acc = 0;
            int at = 0;
            for (int a = 0; a < 20; a++) {
                acc += BG[a] / sum;
                const int end = a == 19 ? 65536 : std::min(65536, (int) (acc * 65536.0 + 0.5));
                for (; at < end; at++) table[at] = (uint8_t) a;
            }
            ready = true;
        }
    }
    return table;
}

constexpr uint32_t FAMILY = 10;
inline uint32_t founder_length(uint64_t seed, uint64_t family) { Rng r(seed, family * 2 + 1); return 150u + r.below(451u); }

}  // namespace

extern "C" {

int mk_synth_targets(uint64_t nTargets, uint64_t seed, uint8_t *residues, uint64_t cap, uint64_t *offsets, uint64_t *total) {
    if (!offsets || !total) return MK_ERR_ARG;
    const uint64_t nFam = (nTargets + FAMILY - 1) / FAMILY;
    offsets[0] = 0;
#pragma omp parallel for schedule(static)
    for (uint64_t f = 0; f < nFam; f++) {
        const uint32_t L = founder_length(seed, f);
        for (uint64_t t = f * FAMILY; t < std::min(nTargets, (f + 1) * FAMILY); t++) offsets[t + 1] = L;
    }
    for (uint64_t t = 0; t < nTargets; t++) offsets[t + 1] += offsets[t];
    *total = offsets[nTargets];
    if (!residues) return MK_OK;
    if (cap < *total) return MK_ERR_ARG;
    const uint8_t *table = residue_table();
#pragma omp parallel for schedule(dynamic, 256)
    for (uint64_t f = 0; f < nFam; f++) {
        Rng r(seed, f * 2);
        const uint64_t t0 = f * FAMILY;
        uint8_t *founder = residues + offsets[t0];
        const uint32_t L = (uint32_t) (offsets[t0 + 1] - offsets[t0]);
        for (uint32_t i = 0; i < L; i++) founder[i] = table[r.next() >> 48];
        for (uint64_t t = t0 + 1; t < std::min(nTargets, t0 + FAMILY); t++) {
            uint8_t *m = residues + offsets[t];
            const uint32_t thr = (uint32_t) (0.05 * (double) (1 + (t - t0)) * 65536.0);
            for (uint32_t i = 0; i < L; i++) {
                const uint64_t x = r.next();
                m[i] = ((uint32_t) (x >> 16) & 0xFFFFu) < thr ? table[x >> 48] : founder[i];
            }
        }
    }
    return MK_OK;
}

int mk_synth_fragments(uint64_t nFragments, uint64_t seed, const uint8_t *tRes, const uint64_t *tOff, uint64_t nTargets, double mutationRate,
                       uint32_t minLen, uint32_t maxLen, uint64_t randomEvery, uint8_t *residues, uint64_t cap, uint64_t *offsets, uint32_t *source,
                       uint64_t *total) {
    if (!offsets || !total || !tRes || !tOff || nTargets == 0 || minLen == 0 || maxLen < minLen) return MK_ERR_ARG;
    offsets[0] = 0;
    std::vector<uint32_t> start(nFragments);
#pragma omp parallel for schedule(static)
    for (uint64_t k = 0; k < nFragments; k++) {
        Rng r(seed ^ 0x5851F42D4C957F2Dull, k * 2 + 1);
        const uint32_t t = (uint32_t) (r.next() % nTargets);
        const uint32_t tl = (uint32_t) (tOff[t + 1] - tOff[t]);
        uint32_t L = minLen + r.below(maxLen - minLen + 1);
        const bool random = randomEvery && (k % randomEvery) == randomEvery - 1;
        if (!random && L > tl) L = tl;
        offsets[k + 1] = L;
        start[k] = (!random && tl > L) ? r.below(tl - L + 1) : 0u;
        if (source) source[k] = random ? 0xFFFFFFFFu : t;
    }
    for (uint64_t k = 0; k < nFragments; k++) offsets[k + 1] += offsets[k];
    *total = offsets[nFragments];
    if (!residues) return MK_OK;
    if (cap < *total) return MK_ERR_ARG;
    const uint8_t *table = residue_table();
    const uint32_t thr = (uint32_t) (mutationRate * 65536.0);
#pragma omp parallel for schedule(dynamic, 1024)
    for (uint64_t k = 0; k < nFragments; k++) {
        Rng r(seed ^ 0x5851F42D4C957F2Dull, k * 2);
        Rng pick(seed ^ 0x5851F42D4C957F2Dull, k * 2 + 1);
        const uint32_t t = (uint32_t) (pick.next() % nTargets);
        const bool random = randomEvery && (k % randomEvery) == randomEvery - 1;
        uint8_t *q = residues + offsets[k];
        const uint32_t L = (uint32_t) (offsets[k + 1] - offsets[k]);
        const uint8_t *src = tRes + tOff[t] + start[k];
        for (uint32_t i = 0; i < L; i++) {
            const uint64_t x = r.next();
            q[i] = (random || ((uint32_t) (x >> 16) & 0xFFFFu) < thr) ? table[x >> 48] : src[i];
        }
    }
    return MK_OK;
}

// the residues as an MMseqs2 sequence DB in memory: data = "SEQ\n\0" entries back to back, rows of the .index (key = position)
int mk_synth_seqdb(const uint8_t *residues, const uint64_t *offsets, uint64_t n, char *data, uint32_t *keys, uint64_t *dataOffsets, uint32_t *lengths) {
    if (!residues || !offsets || !data || !keys || !dataOffsets || !lengths) return MK_ERR_ARG;
    static const char LETTERS[] = "ACDEFGHIKLMNPQRSTVWYX";
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t L = offsets[i + 1] - offsets[i], at = offsets[i] + 2 * i;
        const uint8_t *s = residues + offsets[i];
        for (uint64_t p = 0; p < L; p++) data[at + p] = LETTERS[s[p] <= 20 ? s[p] : 20];
        data[at + L] = '\n'; data[at + L + 1] = '\0';
        keys[i] = (uint32_t) i; dataOffsets[i] = at; lengths[i] = (uint32_t) (L + 2);
    }
    return MK_OK;
}

// ... and straight to disk (<base>, <base>.index, <base>.dbtype = amino acids; key = position), piece by piece: a 2.2e10-residue database is 23 GB
// of data and 60 M index rows, which a Python loop would write for minutes.  with_lines != 0: also <base>.txt, one sequence per line (what the
// reference harness reads).
int mk_synth_write_seqdb(const char *base, const uint8_t *residues, const uint64_t *offsets, uint64_t n, int with_lines) {
    if (!base || !residues || !offsets) return MK_ERR_ARG;
    static const char LETTERS[] = "ACDEFGHIKLMNPQRSTVWYX";
    const std::string b(base);
    FILE *fd = fopen(b.c_str(), "wb"), *fi = fopen((b + ".index").c_str(), "wb"), *ft = fopen((b + ".dbtype").c_str(), "wb");
    FILE *fl = with_lines ? fopen((b + ".txt").c_str(), "wb") : nullptr;
    if (!fd || !fi || !ft || (with_lines && !fl)) { if (fd) fclose(fd); if (fi) fclose(fi); if (ft) fclose(ft); if (fl) fclose(fl); return MK_ERR_ARG; }
    const int dbtype = 0;
    fwrite(&dbtype, 4, 1, ft);
    fclose(ft);
    const uint64_t PIECE = 1u << 20;                              // sequences per piece
    std::vector<char> data, lines, index;
    bool ok = true;
    for (uint64_t i0 = 0; i0 < n && ok; i0 += PIECE) {
        const uint64_t i1 = std::min(n, i0 + PIECE), r0 = offsets[i0], r1 = offsets[i1];
        data.resize((size_t) (r1 - r0) + 2 * (size_t) (i1 - i0));
        if (fl) lines.resize((size_t) (r1 - r0) + (size_t) (i1 - i0));
#pragma omp parallel for schedule(static)
        for (uint64_t i = i0; i < i1; i++) {
            const uint64_t L = offsets[i + 1] - offsets[i], at = (offsets[i] - r0) + 2 * (i - i0), lt = (offsets[i] - r0) + (i - i0);
            const uint8_t *s = residues + offsets[i];
            for (uint64_t p = 0; p < L; p++) { const char c = LETTERS[s[p] <= 20 ? s[p] : 20]; data[at + p] = c; if (fl) lines[lt + p] = c; }
            data[at + L] = '\n'; data[at + L + 1] = '\0';
            if (fl) lines[lt + L] = '\n';
        }
        index.clear();
        char row[64];
        for (uint64_t i = i0; i < i1; i++) {
            const int w = snprintf(row, sizeof(row), "%llu\t%llu\t%llu\n", (unsigned long long) i, (unsigned long long) (offsets[i] + 2 * i), (unsigned long long) (offsets[i + 1] - offsets[i] + 2));
            index.insert(index.end(), row, row + w);
        }
        ok = fwrite(data.data(), 1, data.size(), fd) == data.size() && fwrite(index.data(), 1, index.size(), fi) == index.size() &&
             (!fl || fwrite(lines.data(), 1, lines.size(), fl) == lines.size());
    }
    ok = (fclose(fd) == 0) & ok; ok = (fclose(fi) == 0) & ok;
    if (fl) ok = (fclose(fl) == 0) & ok;
    return ok ? MK_OK : MK_ERR_ARG;
}

}  // extern "C"
