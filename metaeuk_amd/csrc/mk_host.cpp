// metaeuk_amd/csrc/mk_host.cpp -- see mk_host.hpp.  Reference paths: M/ = lib/mmseqs/.
#include "mk_host.hpp"
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <climits>
#include <cstdlib>
#include <numeric>
#include <omp.h>

namespace mk {

// ---- the library's two doors to the environment ----
// launcher_env: what a launcher tells a rank (RANK, LOCAL_WORLD_SIZE, OMP_NUM_THREADS ...): always honoured.
// knob: experiment / test switches (MK_PREFILTER_PATH, MK_SW_WAVES_PER_CU, MK_TEST_ENTRY_BASE ...).  They change tiers, paths and
// launch shapes, so a stray variable must not reach a production run: they are read only when MK_DEBUG=1 is set as well.
const char *launcher_env(const char *name) { return std::getenv(name); }
const char *knob(const char *name) {
    const char *dbg = launcher_env("MK_DEBUG");
    if (!dbg || std::atoi(dbg) == 0) return nullptr;
    const char *v = launcher_env(name);
    return (v && *v) ? v : nullptr;
}
long knob_long(const char *name, long dflt) { const char *v = knob(name); return v ? std::atol(v) : dflt; }

#include "../data/matrices.inc"

const int SPACED6[6] = {0, 1, 3, 5, 8, 9};   // positions of the ones in 1101010011 (M/src/commons/Sequence.h:23)
const int SPACED7[7] = {0, 1, 3, 5, 6, 9, 10};

// SubstitutionMatrix::SubstitutionMatrix -> readProbMatrix (M/src/commons/SubstitutionMatrix.cpp:12-57,
// 326-404) and BaseMatrix::generateSubMatrix (M/src/commons/BaseMatrix.cpp:110-159).
void build_submat(SubMat &m, int which, float bitFactor, float scoreBias) {
    const bool bl = which == MAT_BLOSUM62;
    const double (*S)[ALPH] = bl ? MK_BLOSUM62_SCORES : MK_VTML80_SCORES;
    const double *bg = bl ? MK_BLOSUM62_BACKGROUND : MK_VTML80_BACKGROUND;
    m.lambda = bl ? MK_BLOSUM62_LAMBDA : MK_VTML80_LAMBDA;
    m.name = bl ? "blosum62.out" : "VTML80.out";
    const double px = bg[XCODE];
    for (int i = 0; i < ALPH; i++) m.pback[i] = i < XCODE ? bg[i] * (1.0 - px) : bg[i];
    for (int i = 0; i < ALPH; i++)
        for (int j = 0; j < ALPH; j++)
            m.prob[i][j] = std::exp(m.lambda * S[i][j]) * m.pback[i] * m.pback[j];
    double rowBg[ALPH];
    for (int i = 0; i < ALPH; i++) {
        double acc = 0;
        for (int j = 0; j < ALPH; j++) acc += m.prob[i][j];
        rowBg[i] = acc;
    }
    rowBg[XCODE] = 1E-5;
    const double bf = bitFactor, sb = scoreBias;
    for (int i = 0; i < ALPH; i++)
        for (int j = 0; j < ALPH; j++) {
            const double v = bf * std::log2(m.prob[i][j] / (rowBg[i] * rowBg[j])) + sb;
            m.sub[i][j] = static_cast<short>(v < 0.0 ? v - 0.5 : v + 0.5);
        }
}

// SubstitutionMatrix::setupLetterMapping (SubstitutionMatrix.cpp:138-179) + Sequence::mapSequence
void encode(const char *s, size_t n, uint8_t *codes) {
    static uint8_t lut[256];
    static bool ready = false;
    if (!ready) {
        uint8_t pos[256];
        std::memset(pos, XCODE, sizeof(pos));
        for (int i = 0; MK_BLOSUM62_ALPHABET[i]; i++) pos[static_cast<unsigned char>(MK_BLOSUM62_ALPHABET[i])] = static_cast<uint8_t>(i);
        for (int c = 0; c < 256; c++) {
            const int u = std::toupper(c);
            uint8_t v = XCODE;
            if (u == 'J') v = pos['L'];
            else if (u == 'Z') v = pos['E'];
            else if (u == 'B') v = pos['D'];
            else if (u == 'U' || u == 'O') v = XCODE;
            else if (u >= 'A' && u <= 'Z') v = pos[u];
            lut[c] = v;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) codes[i] = lut[static_cast<unsigned char>(s[i])];
}

// SubstitutionMatrix::calcLocalAaBiasCorrection (SubstitutionMatrix.cpp:79-109)
void comp_bias(const SubMat &m, const uint8_t *seq, int L, float scale, float *bias) {
    for (int i = 0; i < L; i++) {
        const int lo = std::max(0, i - 20), hi = std::min(L, i + 20);
        const short *row = m.sub[seq[i]];
        int sum = 0;
        for (int j = lo; j < hi; j++) sum += row[seq[j]];
        sum -= row[seq[i]];
        float d = static_cast<float>(sum);
        d /= -1.0 * static_cast<float>(hi - lo);               // double division, rounded to float
        for (int a = 0; a < ALPH; a++) d += m.pback[a] * static_cast<float>(row[a]);   // double accumulate, float store
        bias[i] = scale * d;
    }
}

// Prefiltering::getKmerThreshold (M/src/prefiltering/Prefiltering.cpp:1005-1065), sequence search, k = 6
int kmer_threshold(float sensitivity, int kmerScoreOverride) {
    if (kmerScoreOverride != INT_MAX) return kmerScoreOverride;
    float base = 163.2;
    float best = base - (sensitivity * 8.917);
    return static_cast<int>(best);
}

int kmer_threshold_k7(float sensitivity, int kmerScoreOverride) {
    if (kmerScoreOverride != INT_MAX) return kmerScoreOverride;
    float base = 186.15;
    float best = base - (sensitivity * 11.22);
    return static_cast<int>(best);
}

// the same for a profile search without context pseudo counts (:1036-1040), k = 6
int kmer_threshold_profile(float sensitivity) {
    float base = 134.35;
    float best = base - (sensitivity * 6.15);
    return static_cast<int>(best);
}

// ... and with k = 7 (:1041-1043)
int kmer_threshold_profile_k7(float sensitivity) {
    float base = 149.15;
    float best = base - (sensitivity * 6.85);
    return static_cast<int>(best);
}

// QueryMatcher::initDiagonalMatcher (M/src/prefiltering/QueryMatcher.cpp:422-450)
int bin_count_for(uint64_t dbSize, uint64_t l2) {
    for (int b = 2; b <= 1024; b <<= 1) if (dbSize / static_cast<uint64_t>(b) < l2) return b;
    return 2048;
}

// Table addresses of the k-mers (bitmap, slots, lists) -- addresses only: rows of the 3-mer score table are still looked up and
// ordered by the reference's numbering, so the enumeration order (and with it the order in which hits arrive) is the reference's.
// The similar k-mers of a query k-mer differ from it by substitutions of similar residues at several of the six positions, and every
// 8-byte probe that misses L2 costs a 128-byte line from HBM (1024 bitmap cells, 16 slots): the address order is TILED so that such a
// cloud falls into few lines.  The 20 letters form five quads of similar residues (VILM | FYWH | RKQE | DNST | AGPC); a letter is
// (quad, position in the quad).  A 3-mer's address code is tile << 6 | w with tile = the three quads (base 5) and w = the three in-quad
// positions (base 4): a bijection onto 0..7999.  A k-mer's cell is 4096 * (tile(first 3-mer) + 125 * tile(second)) + w(first) + 64 * w(second):
// all 4^6 k-mers that agree in their six quads share one 4096-cell tile (four 128-byte lines of the bitmap).
static const uint8_t KMER_ADDR_LETTER[20] = {
    /* A */ 16, /* C */ 19, /* D */ 12, /* E */ 11, /* F */ 4, /* G */ 17, /* H */ 7, /* I */ 1, /* K */ 9, /* L */ 2,
    /* M */ 3, /* N */ 13, /* P */ 18, /* Q */ 10, /* R */ 8, /* S */ 14, /* T */ 15, /* V */ 0, /* W */ 6, /* Y */ 5};
const uint8_t *kmer_addr_letters() { return KMER_ADDR_LETTER; }
static inline uint16_t addr3_of_letters(int l0, int l1, int l2) {
    const int d0 = KMER_ADDR_LETTER[l0], d1 = KMER_ADDR_LETTER[l1], d2 = KMER_ADDR_LETTER[l2];
    return static_cast<uint16_t>((((d0 >> 2) + 5 * (d1 >> 2) + 25 * (d2 >> 2)) << 6) | ((d0 & 3) + 4 * (d1 & 3) + 16 * (d2 & 3)));
}
uint32_t kmer_cell(uint32_t addrFirst, uint32_t addrSecond) {
    return 4096u * ((addrFirst >> 6) + 125u * (addrSecond >> 6)) + (addrFirst & 63u) + 64u * (addrSecond & 63u);
}

// ExtendedSubstitutionMatrix::calcScoreMatrix (M/src/prefiltering/ExtendedSubstitutionMatrix.cpp:20-69),
// kmerSize 3 over the 20-letter alphabet (Prefiltering.cpp:208-213).  std::stable_sort by descending
// score over candidates enumerated in cartesian order with the FIRST letter slowest.
void build_scoremat3(const SubMat &km, ScoreMat3 &out) {
    const int N = 8000;
    out.score.assign(static_cast<size_t>(N) * N, 0);
    out.index.assign(static_cast<size_t>(N) * N, 0);
    std::vector<uint16_t> enumIdx(N);      // enumeration order -> Indexer index (a0 + 20 a1 + 400 a2): the row of a query 3-mer
    std::vector<uint16_t> addrIdx(N);      // enumeration order -> table address digits of the candidate 3-mer
    std::vector<uint8_t> letter(N * 3);
    for (int e = 0; e < N; e++) {
        const int a0 = e / 400, a1 = (e / 20) % 20, a2 = e % 20;
        letter[e * 3] = a0; letter[e * 3 + 1] = a1; letter[e * 3 + 2] = a2;
        enumIdx[e] = static_cast<uint16_t>(a0 + 20 * a1 + 400 * a2);
        addrIdx[e] = addr3_of_letters(a0, a1, a2);            // (a0 = the letter of the reference's fastest digit, see enumIdx)
    }
#pragma omp parallel
    {
        std::vector<int16_t> sc(N);
        std::vector<int> start(512);
#pragma omp for schedule(static)
        for (int e = 0; e < N; e++) {
            const short *r0 = km.sub[letter[e * 3]], *r1 = km.sub[letter[e * 3 + 1]], *r2 = km.sub[letter[e * 3 + 2]];
            int lo = INT_MAX, hi = INT_MIN;
            for (int f = 0; f < N; f++) {
                const int v = static_cast<short>(r0[letter[f * 3]] + r1[letter[f * 3 + 1]] + r2[letter[f * 3 + 2]]);
                sc[f] = static_cast<int16_t>(v);
                lo = std::min(lo, v); hi = std::max(hi, v);
            }
            const int range = hi - lo + 1;
            if (static_cast<int>(start.size()) < range + 1) start.resize(range + 1);
            std::fill(start.begin(), start.begin() + range + 1, 0);
            for (int f = 0; f < N; f++) start[hi - sc[f] + 1]++;
            for (int r = 0; r < range; r++) start[r + 1] += start[r];
            const size_t base = static_cast<size_t>(enumIdx[e]) * N;
            for (int f = 0; f < N; f++) {
                const int p = start[hi - sc[f]]++;
                out.score[base + p] = sc[f];
                out.index[base + p] = addrIdx[f];
            }
        }
    }
    // score histograms of every row (rows are sorted descending: first/last entry = max/min)
    int lo = INT_MAX, hi = INT_MIN;
    for (int r = 0; r < N; r++) { hi = std::max<int>(hi, out.score[static_cast<size_t>(r) * N]); lo = std::min<int>(lo, out.score[static_cast<size_t>(r) * N + N - 1]); }
    out.histLo = lo; out.histRange = hi - lo + 1;
    out.hist.assign(static_cast<size_t>(N) * out.histRange, 0);
    out.cum.assign(static_cast<size_t>(N) * out.histRange, 0);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < N; r++) {
        uint16_t *h = out.hist.data() + static_cast<size_t>(r) * out.histRange, *c = out.cum.data() + static_cast<size_t>(r) * out.histRange;
        for (int f = 0; f < N; f++) h[out.score[static_cast<size_t>(r) * N + f] - lo]++;
        unsigned acc = 0;
        for (int k = out.histRange - 1; k >= 0; k--) { acc += h[k]; c[k] = static_cast<uint16_t>(acc); }
    }
}

// tantan::maskSequences as driven by Masker::maskSequence (M/src/commons/Masker.cpp:15-32,
// M/lib/tantan/tantan.cpp:320-449,475-527): forward-backward over 50 repeat offsets, no gap states.
// `lanes` = doubles per SIMD register of the build being reproduced (partial-sum association).
int tantan_mask(const SubMat &km, uint8_t *seq, int L, double minMaskProb, int lanes) {
    constexpr int W = 50, STEP = 16;
    if (L <= 0) return 0;
    double ratio[ALPH][ALPH];
    for (int i = 0; i < ALPH; i++)
        for (int j = 0; j < ALPH; j++) ratio[i][j] = km.prob[i][j] / (km.pback[i] * km.pback[j]);
    const double pRepeat = 0.005, pEnd = 0.05, decay = 0.9;
    const double bgStay = 1 - pRepeat, fgStay = 1 - pEnd;
    double enter[W], fg[W];
    {
        double p = pRepeat * ((1 - decay) / (1 - std::pow(decay, W)));
        for (int i = 0; i < W; i++) { enter[i] = p; p *= decay; }
    }
    std::vector<float> post(L);
    std::vector<double> scale(L / STEP + 1, 0.0);
    auto hsum = [lanes](const double *l) { return lanes == 4 ? (l[0] + l[2]) + (l[1] + l[3]) : (lanes == 2 ? l[0] + l[1] : l[0]); };
    double bgp = 1.0;
    std::fill(fg, fg + W, 0.0);
    for (int pos = 0; pos < L; pos++) {
        const double *rr = ratio[seq[pos]];
        const int reach = std::min(pos, W);
        const double b = bgp;
        double part[4] = {0, 0, 0, 0};
        int i = 0;
        for (; i + lanes <= reach; i += lanes)
            for (int l = 0; l < lanes; l++) {
                const double f = fg[i + l];
                part[l] = part[l] + f;
                fg[i + l] = (b * enter[i + l] + f * fgStay) * rr[seq[pos - 1 - (i + l)]];
            }
        double fromFg = hsum(part);
        for (; i < reach; i++) {
            const double f = fg[i];
            fromFg += f;
            fg[i] = (b * enter[i] + f * fgStay) * rr[seq[pos - 1 - i]];
        }
        bgp = b * bgStay + fromFg * pEnd;
        if (pos % STEP == STEP - 1) {
            const double s = 1 / bgp;
            scale[pos / STEP] = s;
            bgp *= s;
            for (int k = 0; k < W; k++) fg[k] *= s;
        }
        post[pos] = static_cast<float>(bgp);
    }
    double tail = 0.0;
    for (int k = 0; k < W; k++) tail += fg[k];
    const double total = bgp * bgStay + tail * pEnd;
    bgp = bgStay;
    std::fill(fg, fg + W, pEnd);
    for (int pos = L - 1; pos >= 0; pos--) {
        const double nonRepeat = post[pos] * bgp / total;
        post[pos] = 1 - static_cast<float>(nonRepeat);
        if (pos % STEP == STEP - 1) {
            const double s = scale[pos / STEP];
            bgp *= s;
            for (int k = 0; k < W; k++) fg[k] *= s;
        }
        const double *rr = ratio[seq[pos]];
        const int reach = std::min(pos, W);
        const double toBg = pEnd * bgp;
        double part[4] = {0, 0, 0, 0};
        int i = 0;
        for (; i + lanes <= reach; i += lanes)
            for (int l = 0; l < lanes; l++) {
                const double f = fg[i + l] * rr[seq[pos - 1 - (i + l)]];
                part[l] = part[l] + enter[i + l] * f;
                fg[i + l] = toBg + fgStay * f;
            }
        double toFg = hsum(part);
        for (; i < reach; i++) {
            const double f = fg[i] * rr[seq[pos - 1 - i]];
            toFg += enter[i] * f;
            fg[i] = toBg + fgStay * f;
        }
        bgp = bgStay * bgp + toFg;
    }
    int masked = 0;
    for (int pos = 0; pos < L; pos++)
        if (post[pos] >= minMaskProb) { seq[pos] = XCODE; masked++; }
    return masked;
}

// IndexBuilder::fillDatabase (M/src/prefiltering/IndexBuilder.cpp:55-239) + IndexTable::addKmerCount /
// addSequence / sortDBSeqLists (M/src/prefiltering/IndexTable.h:133-173,348-401,182-189), AA targets, k=6.
void kmer3_address_table(uint16_t addrOf[8000]) {
    for (int k = 0; k < 8000; k++)
        addrOf[k] = addr3_of_letters(k % 20, (k / 20) % 20, k / 400);
}

// k-mer lists in the reference's numbering -> in table-address order (what the device tables use)
void index_to_address_order(TargetIndex &ix) {
    const uint64_t TABLE = 64000000ull;
    uint16_t addr3[8000];
    kmer3_address_table(addr3);
    std::vector<uint64_t> offsets(TABLE + 1, 0);
#pragma omp parallel for schedule(static)
    for (uint64_t hi = 0; hi < 8000; hi++)
        for (uint64_t lo = 0; lo < 8000; lo++) {
            const uint64_t k = lo + 8000 * hi, a = kmer_cell(addr3[lo], addr3[hi]);
            offsets[a + 1] = ix.offsets[k + 1] - ix.offsets[k];
        }
    for (uint64_t a = 0; a < TABLE; a++) offsets[a + 1] += offsets[a];
    std::vector<uint64_t> entries(ix.entries.size());
#pragma omp parallel for schedule(static)
    for (uint64_t hi = 0; hi < 8000; hi++)
        for (uint64_t lo = 0; lo < 8000; lo++) {
            const uint64_t k = lo + 8000 * hi, a = kmer_cell(addr3[lo], addr3[hi]);
            std::copy(ix.entries.begin() + ix.offsets[k], ix.entries.begin() + ix.offsets[k + 1], entries.begin() + offsets[a]);
        }
    ix.offsets.swap(offsets);
    ix.entries.swap(entries);
}

void kmer3_number_of_address(uint16_t numOf[8000]) {
    uint16_t addr[8000];
    kmer3_address_table(addr);
    for (int k = 0; k < 8000; k++) numOf[addr[k]] = static_cast<uint16_t>(k);
}

void build_scoremat2(const SubMat &km, std::vector<int16_t> &score, std::vector<uint16_t> &index) {
    const int N = 400;
    score.assign(static_cast<size_t>(N) * N, 0);
    index.assign(static_cast<size_t>(N) * N, 0);
    for (int e = 0; e < N; e++) {                          // enumeration order: first letter slowest; row of the query 2-mer a0 + 20 a1
        const int a0 = e / 20, a1 = e % 20;
        std::vector<std::pair<int, int>> cand(N);          // (score, enumeration rank)
        for (int f = 0; f < N; f++) cand[f] = {static_cast<short>(km.sub[a0][f / 20] + km.sub[a1][f % 20]), f};
        std::stable_sort(cand.begin(), cand.end(), [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.first > y.first; });
        const size_t row = static_cast<size_t>(a0 + 20 * a1) * N;
        for (int r = 0; r < N; r++) {
            score[row + r] = static_cast<int16_t>(cand[r].first);
            index[row + r] = static_cast<uint16_t>(cand[r].second / 20 + 20 * (cand[r].second % 20));
        }
    }
}

void build_index(const SubMat &km, const uint8_t *residues, const uint64_t *seqOff, uint32_t nSeq,
                 int kmerThr, bool mask, float maskProb, int tantanLanes, TargetIndex &out, bool addressOrder, int kmerSize) {
    const uint64_t TABLE = kmerSize == 7 ? 1280000000ull : 64000000ull;
    const int K = kmerSize == 7 ? 7 : 6, SPANK = kmerSize == 7 ? 11 : SPAN;
    const int *SP = kmerSize == 7 ? SPACED7 : SPACED6;
    if (kmerSize == 7) addressOrder = false;              // k = 7 cells are the reference's numbering (no tiling)
    const uint64_t total = seqOff[nSeq];
    out.masked.assign(residues, residues + total);
    uint64_t maskedCount = 0;
    if (mask) {
        const double thr = static_cast<double>(maskProb);   // float flag widened (Prefiltering.cpp:39, Masker.cpp:15)
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : maskedCount)
        for (uint32_t s = 0; s < nSeq; s++)
            maskedCount += tantan_mask(km, out.masked.data() + seqOff[s], static_cast<int>(seqOff[s + 1] - seqOff[s]), thr, tantanLanes);
    }
    out.maskedResidues = maskedCount;
    signed char self[ALPH];
    for (int a = 0; a < ALPH; a++) self[a] = static_cast<signed char>(km.sub[a][a]);
    // per sequence: distinct k-mers with their first position, in parallel; then counting + fill
    std::vector<std::vector<uint64_t>> perSeq(nSeq);    // kmer << 16 | pos, sorted, first position per k-mer
#pragma omp parallel
    {
        std::vector<uint64_t> buf;
#pragma omp for schedule(dynamic, 64)
        for (uint32_t s = 0; s < nSeq; s++) {
            const uint8_t *seq = out.masked.data() + seqOff[s];
            const int L = static_cast<int>(seqOff[s + 1] - seqOff[s]);
            buf.clear();
            for (int i = 0; i + SPANK <= L; i++) {
                uint32_t idx = 0, pw = 1;
                int score = 0;
                bool hasX = false;
                uint8_t let[7];
                for (int p = 0; p < K; p++) {
                    const uint8_t c = seq[i + SP[p]];
                    hasX |= (c == XCODE);
                    score += self[c];
                    let[p] = c < 20 ? c : 0;
                    idx += (c < 20 ? c * pw : 0);                  // Indexer::int2index
                    pw *= 20;
                }
                if (hasX || (kmerThr > 0 && score < kmerThr)) continue;
                if (addressOrder) idx = kmer_cell(addr3_of_letters(let[0], let[1], let[2]), addr3_of_letters(let[3], let[4], let[5]));   // the table address
                buf.push_back((static_cast<uint64_t>(idx) << 16) | static_cast<uint64_t>(i & 0xFFFF));
            }
            std::sort(buf.begin(), buf.end());
            std::vector<uint64_t> &dst = perSeq[s];
            uint64_t prev = ~0ull;
            for (uint64_t v : buf) {
                if ((v >> 16) != prev) dst.push_back(v);
                prev = v >> 16;
            }
        }
    }
    out.offsets.assign(TABLE + 1, 0);
    for (uint32_t s = 0; s < nSeq; s++)
        for (uint64_t v : perSeq[s]) out.offsets[(v >> 16) + 1]++;
    for (uint64_t k = 0; k < TABLE; k++) out.offsets[k + 1] += out.offsets[k];
    out.entries.assign(out.offsets[TABLE], 0);
    std::vector<uint64_t> cursor(out.offsets.begin(), out.offsets.end() - 1);
    for (uint32_t s = 0; s < nSeq; s++)                      // ascending seqId => lists sorted by (seqId,pos)
        for (uint64_t v : perSeq[s]) out.entries[cursor[v >> 16]++] = static_cast<uint64_t>(s) | ((v & 0xFFFF) << 32);
}

// EvalueComputation (M/src/alignment/EvalueComputation.h:18-40,64-69) over Sls::AlignmentEvaluer
// (M/lib/alp/sls_alignment_evaluer.cpp:657-835,989-1029; sls_pvalues.cpp:342-520; sls_basic.hpp:195-198)
void Evaluer::init(uint64_t dbResidues) {
    lambda = 0.27359865037097330642; K = 0.044620920658722244834;
    a_J = 1.5938724404943873658; b_J = -19.959867650284412122; a_I = a_J; b_I = b_J;
    alpha_J = 30.455610143099914211; beta_J = -622.28684628915891608; alpha_I = alpha_J; beta_I = beta_J;
    sigma = 29.602444874818868215; tau = -601.81087985041381216;
    vi_y_thr = std::max(2.0 * alpha_I / lambda, 0.0);
    vj_y_thr = std::max(2.0 * alpha_J / lambda, 0.0);
    c_y_thr = std::max(2.0 * sigma / lambda, 0.0);
    logK = std::log(K);
    dbRes = static_cast<double>(dbResidues);
}

double Evaluer::evalue(double y, double qLen) const {
    const double pi = 3.1415926535897932384626433832795;
    const double norm = 1 / std::sqrt(2.0 * pi);
    auto side = [&](double len, double a, double b, double alpha, double beta, double vthr, double &P) {
        const double shortfall = len - (a * y + b);
        const double sd = std::sqrt(std::max(vthr, alpha * y + beta));
        const double z = (sd == 0.0) ? 1e100 : shortfall / sd;
        P = 0.5 * std::erfc(-std::sqrt(0.5) * z);
        const double E = -norm * std::exp(-0.5 * z * z);
        return shortfall * P - sd * E;
    };
    double Pm, Pn;
    const double p1 = side(dbRes, a_I, b_I, alpha_I, beta_I, vi_y_thr, Pm);
    const double p2 = side(qLen, a_J, b_J, alpha_J, beta_J, vj_y_thr, Pn);
    const double c = std::max(c_y_thr, sigma * y + tau);
    const double PmPn = Pm * Pn;
    const double cP = c * PmPn;
    const double p1p2 = p1 * p2;
    const double area = p1p2 + cP;
    const double perArea = K * std::exp(-lambda * y);
    return perArea * area;
}

double Evaluer::bitScore(double score) const { return (lambda * score - logK) / std::log(2.0); }

// Tail of QueryMatcher::matchQuery (QueryMatcher.cpp:149-209): keepMaxScoreElementOnly
// (CacheFriendlyOperations.cpp:350-380), score histogram + computeScoreThreshold (QueryMatcher.h:206-216),
// radixSortByScoreSize (:498-523), rescoreHits (:525-544), getResult<1> (:363-420), final sort.
// The reference's element order is bin-major (id & (BINSIZE-1)) then arrival order; that order decides
// ties at the --max-seqs cut, so it is rebuilt here from the canonical ordinals.
int select_hits(std::vector<Cand> &cands, int binCount, int maxHits, int minDiagScore, int selfScore, mk_hit *out) {
    const uint32_t mask = static_cast<uint32_t>(binCount - 1);
    std::stable_sort(cands.begin(), cands.end(), [mask](const Cand &a, const Cand &b) {
        const uint32_t ba = a.id & mask, bb = b.id & mask;
        if (ba != bb) return ba < bb;
        return a.ordinal < b.ordinal;
    });
    // best clamped score per id; the first element reaching it survives (zero-score ids keep every element)
    std::vector<Cand> kept;
    kept.reserve(cands.size());
    {
        std::vector<std::pair<uint32_t, int>> best;   // (id, max clamped), small: linear probe via sort
        best.reserve(cands.size());
        for (const Cand &c : cands) best.emplace_back(c.id, std::min(c.score, 255));
        std::sort(best.begin(), best.end(), [](const std::pair<uint32_t, int> &a, const std::pair<uint32_t, int> &b) {
            return a.first != b.first ? a.first < b.first : a.second > b.second;
        });
        best.erase(std::unique(best.begin(), best.end(), [](const std::pair<uint32_t, int> &a, const std::pair<uint32_t, int> &b) { return a.first == b.first; }), best.end());
        std::vector<uint8_t> taken(best.size(), 0);
        for (const Cand &c : cands) {
            const int clamped = std::min(c.score, 255);
            const size_t k = std::lower_bound(best.begin(), best.end(), c.id, [](const std::pair<uint32_t, int> &a, uint32_t id) { return a.first < id; }) - best.begin();
            if (best[k].second != clamped) continue;
            if (clamped != 0) {
                if (taken[k]) continue;
                taken[k] = 1;
            }
            kept.push_back(c);
        }
    }
    unsigned int hist[256];
    std::memset(hist, 0, sizeof(hist));
    for (const Cand &c : kept) hist[std::min(c.score, 255)]++;
    size_t acc = 0, t = 255;
    for (; t > 0; t--) { acc += hist[t]; if (acc >= static_cast<size_t>(maxHits)) break; }
    const unsigned int thr = std::max<unsigned int>(static_cast<unsigned int>(minDiagScore), static_cast<unsigned int>(t));
    std::vector<Cand> ranked;
    ranked.reserve(kept.size());
    for (const Cand &c : kept) if (static_cast<unsigned int>(std::min(c.score, 255)) >= thr) ranked.push_back(c);
    std::stable_sort(ranked.begin(), ranked.end(), [](const Cand &a, const Cand &b) { return std::min(a.score, 255) > std::min(b.score, 255); });
    int n = 0;
    if (thr >= 255) {
        int self = selfScore - 255;
        self = std::min(std::max(self, 1), static_cast<int>(USHRT_MAX));
        const float fself = static_cast<float>(self);
        struct R { Cand c; unsigned char resc; };
        std::vector<R> rs;
        for (const Cand &c : ranked) {
            unsigned int ns = static_cast<unsigned int>(c.score) - 255u;
            const float s = static_cast<float>(std::min(ns, static_cast<unsigned int>(USHRT_MAX)));
            rs.push_back({c, static_cast<unsigned char>((s / fself) * static_cast<float>(UCHAR_MAX) + 0.5)});
        }
        std::stable_sort(rs.begin(), rs.end(), [](const R &a, const R &b) { return a.resc > b.resc; });
        for (const R &r : rs) {
            if (n >= maxHits) break;
            out[n].seq_id = r.c.id; out[n].diagonal = r.c.diag; out[n].pad_ = 0;
            out[n].pref_score = static_cast<int32_t>(255u + (static_cast<unsigned int>(r.resc) * static_cast<unsigned int>(self) / 255u));
            n++;
        }
    } else {
        for (const Cand &c : ranked) {
            if (n >= maxHits) break;
            out[n].seq_id = c.id; out[n].diagonal = c.diag; out[n].pad_ = 0;
            out[n].pref_score = c.score;    // exact when the clamped value saturated, equal otherwise
            n++;
        }
    }
    std::sort(out, out + n, [](const mk_hit &a, const mk_hit &b) {
        const int sa = std::abs(a.pref_score), sb = std::abs(b.pref_score);
        if (sa != sb) return sa > sb;
        return a.seq_id < b.seq_id;
    });
    return n;
}

// Orf::iupacReverseComplementTable (Orf.cpp:48-52) and TranslateNucl::initConversionTable (TranslateNucl.h:305-343)
void build_orf_tables(char comp[256], uint8_t base[256]) {
    std::memset(comp, '.', 256);
    const char *from = "ACGTUMRWSYKVHDBNacgtumrwsykvhdbn", *to = "TGCAAKYWSRMBDHVNtgcaakywsrmbdhvn";
    for (int i = 0; from[i]; i++) comp[static_cast<unsigned char>(from[i])] = to[i];
    std::memset(base, 0, 256);
    const char *codes = "-ACMGRSVTWYHKDBN";                  // eBase_gap .. eBase_N: bit 1 = A, 2 = C, 4 = G, 8 = T
    for (int i = 0; i < 16; i++) {
        base[static_cast<unsigned char>(codes[i])] = static_cast<uint8_t>(i);
        base[static_cast<unsigned char>(std::tolower(codes[i]))] = static_cast<uint8_t>(i);
    }
    base['U'] = base['u'] = 8;
    base['X'] = base['x'] = 15;
    for (int i = 0; i < 16; i++) base[i] = static_cast<uint8_t>(i);
}

// TranslateNucl::initTranslationTable (TranslateNucl.h:344-483) for the canonical code: the residue common to all the
// unambiguous codons an IUPAC codon stands for, with the B (D/N), Z (E/Q), J (I/L) unions, else X
void build_translation_table(char table[4096]) {
    const char *aaOf = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";   // TCAG order
    auto tcag = [](int bit) { return bit == 8 ? 0 : bit == 2 ? 1 : bit == 1 ? 2 : 3; };
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++)
            for (int k = 0; k < 16; k++) {
                char aa = 0;
                for (int x = 1; x <= 8; x <<= 1) {
                    if (!(x & i)) continue;
                    for (int y = 1; y <= 8; y <<= 1) {
                        if (!(y & j)) continue;
                        for (int z = 1; z <= 8; z <<= 1) {
                            if (!(z & k)) continue;
                            // expansion order of the reference is A, C, G, T on every position (bits 1, 2, 4, 8)
                            const char ch = aaOf[16 * tcag(x) + 4 * tcag(y) + tcag(z)];
                            if (!aa) aa = ch;
                            else if (aa != ch) {
                                if ((aa == 'B' || aa == 'D' || aa == 'N') && (ch == 'D' || ch == 'N')) aa = 'B';
                                else if ((aa == 'Z' || aa == 'E' || aa == 'Q') && (ch == 'E' || ch == 'Q')) aa = 'Z';
                                else if ((aa == 'J' || aa == 'I' || aa == 'L') && (ch == 'I' || ch == 'L')) aa = 'J';
                                else aa = 'X';
                            }
                        }
                    }
                }
                table[256 * i + 16 * j + k] = aa ? aa : 'X';
            }
}

size_t format_orf_header(char *buf, uint32_t key, uint32_t from, uint32_t to, bool incompleteStart, bool incompleteEnd) {
    const int len = std::abs(static_cast<int>(from) - static_cast<int>(to));
    const int complete = (incompleteStart ? 1 : 0) | ((incompleteEnd ? 1 : 0) << 1);
    int n = std::snprintf(buf, 64, "%u\t%u%c%d", key, from, from < to ? '+' : '-', len);
    if (complete) n += std::snprintf(buf + n, 16, "\t%d", complete);
    return static_cast<size_t>(n);
}

float compute_cov(unsigned int s, unsigned int e, unsigned int len) {
    return (std::min(len, std::max(s, e)) - std::min(s, e) + 1) / static_cast<float>(len);
}

size_t format_hit(char *buf, uint32_t key, int32_t score, uint16_t diag) {
    return static_cast<size_t>(std::sprintf(buf, "%u\t%d\t%d\n", key, score, static_cast<int>(static_cast<short>(diag))));
}

// Matcher::resultToBuffer (Matcher.cpp:280-327); seq. id. through Util::fastSeqIdToBuffer (Util.cpp:222-251),
// whose 1.0 branch returns a pointer at its terminator so the following tab eats the third decimal.
size_t format_alignment(char *buf, const mk_alignment &a) {
    char *p = buf;
    p += std::sprintf(p, "%u\t%d\t", a.db_key, a.bit_score);
    if (a.seq_id == 1.0) {
        p += std::sprintf(p, "1.00");
    } else {
        *p++ = '0'; *p++ = '.';
        if (a.seq_id < 0.10) *p++ = '0';
        if (a.seq_id < 0.01) *p++ = '0';
        p += std::sprintf(p, "%d", static_cast<int>(a.seq_id * 1000));
    }
    p += std::sprintf(p, "\t%.3E\t%d\t%d\t%d\t%d\t%d\t%d\n", a.evalue, a.q_start, a.q_end, a.q_len, a.db_start, a.db_end, a.db_len);
    return static_cast<size_t>(p - buf);
}

// Matcher::compareHits (M/src/alignment/Matcher.h:157-168)
bool alignment_less(const mk_alignment &a, const mk_alignment &b) {
    if (a.evalue != b.evalue) return a.evalue < b.evalue;
    if (a.bit_score != b.bit_score) return a.bit_score > b.bit_score;
    if (a.db_len != b.db_len) return a.db_len < b.db_len;
    return a.db_key < b.db_key;
}

}  // namespace mk
