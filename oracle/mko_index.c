/* oracle/mko_index.c -- TEST INFRASTRUCTURE (parity oracle).  See mko.h.
 * tantan repeat masking of targets and the target k-mer index. */
#include "mko.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* tantan::maskSequences (M/lib/tantan/tantan.cpp:475-494 -> Tantan::calcRepeatProbs :416-449) as
 * called by Masker::maskSequence (M/src/commons/Masker.cpp:15-32): maxCycleLength 50,
 * repeatProb 0.005, repeatEndProb 0.05, repeatOffsetProbDecay 0.9, no gaps, minMaskProb 0.9,
 * likelihood-ratio matrix = ProbabilityMatrix (BaseMatrix.h:91-103): prob[i][j]/(pBack[i]*pBack[j]).
 * With firstGapProb == 0 only the gapless SIMD paths run (tantan.cpp:320-355, 357-392).  Their
 * partial sums are per SIMD lane (`simd_lanes` doubles: 4 for AVX2, 2 for SSE4.1, 1 scalar) and
 * are combined as simdHorizontalAddDbl does (mcf_simd.h:175-179 / :332-334).
 * Floating-point caveat: the reference compiler may contract a*b+c into FMA; this restatement
 * does not.  A masking decision could only differ if a posterior lands within ~1e-15 of 0.9. */
int mko_tantan_mask(const mko_submat *km, uint8_t *seq, int L, double minMaskProb, int simd_lanes) {
    enum { maxRepeatOffset = 50, scaleStepSize = 16 };
    if (L <= 0) return 0;
    const double repeatProb = 0.005, repeatEndProb = 0.05, decay = 0.9;
    double lr[MKO_ALPH][MKO_ALPH];
    for (int i = 0; i < MKO_ALPH; i++)
        for (int j = 0; j < MKO_ALPH; j++)
            lr[i][j] = km->prob[i][j] / (km->pback[i] * km->pback[j]);
    const double b2b = 1 - repeatProb, f2b = repeatEndProb, f2f0 = 1 - repeatEndProb;
    /* firstRepeatOffsetProb (tantan.cpp:25-30) */
    const double b2fFirst = repeatProb * ((1 - decay) / (1 - pow(decay, maxRepeatOffset)));
    double b2f[maxRepeatOffset], fg[maxRepeatOffset];
    double p = b2fFirst;
    for (int i = 0; i < maxRepeatOffset; i++) { b2f[i] = p; p *= decay; }
    float *probs = (float *) malloc((size_t) L * sizeof(float));
    double *scaleFactors = (double *) calloc((size_t) L / scaleStepSize + 1, sizeof(double));
    double background = 1.0;
    for (int i = 0; i < maxRepeatOffset; i++) fg[i] = 0.0;
    const int W = simd_lanes;
    /* forward (calcForwardTransitionAndEmissionProbs) */
    for (int pos = 0; pos < L; pos++) {
        const double *lrRow = lr[seq[pos]];
        const int maxOffset = pos < maxRepeatOffset ? pos : maxRepeatOffset;
        const double b = background;
        double lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = 0;
        for (; i <= maxOffset - W; i += W) {
            for (int l = 0; l < W; l++) {
                double f = fg[i + l];
                lane[l] = lane[l] + f;
                fg[i + l] = (b * b2f[i + l] + f * f2f0) * lrRow[seq[pos - (i + l) - 1]];
            }
        }
        double fromForeground;
        if (W == 4) fromForeground = (lane[0] + lane[2]) + (lane[1] + lane[3]);
        else if (W == 2) fromForeground = lane[0] + lane[1];
        else fromForeground = lane[0];
        for (; i < maxOffset; i++) {
            double f = fg[i];
            fromForeground += f;
            fg[i] = (b * b2f[i] + f * f2f0) * lrRow[seq[pos - i - 1]];
        }
        background = b * b2b + fromForeground * f2b;
        if (pos % scaleStepSize == scaleStepSize - 1) {   /* rescaleForward :400-407 */
            double scale = 1 / background;
            scaleFactors[pos / scaleStepSize] = scale;
            background *= scale;
            for (int k = 0; k < maxRepeatOffset; k++) fg[k] *= scale;
        }
        probs[pos] = (float) background;
    }
    /* forwardTotal :139-145 (std::accumulate = sequential) */
    double fromFg = 0.0;
    for (int k = 0; k < maxRepeatOffset; k++) fromFg += fg[k];
    const double z = background * b2b + fromFg * f2b;
    /* backward */
    background = b2b;
    for (int k = 0; k < maxRepeatOffset; k++) fg[k] = f2b;
    for (int pos = L - 1; pos >= 0; pos--) {
        double nonRepeatProb = probs[pos] * background / z;
        probs[pos] = 1 - (float) nonRepeatProb;
        if (pos % scaleStepSize == scaleStepSize - 1) {   /* rescaleBackward :409-414 */
            double scale = scaleFactors[pos / scaleStepSize];
            background *= scale;
            for (int k = 0; k < maxRepeatOffset; k++) fg[k] *= scale;
        }
        /* calcEmissionAndBackwardTransitionProbs :357-392 */
        const double *lrRow = lr[seq[pos]];
        const int maxOffset = pos < maxRepeatOffset ? pos : maxRepeatOffset;
        const double toBackground = f2b * background;
        double lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = 0;
        for (; i <= maxOffset - W; i += W) {
            for (int l = 0; l < W; l++) {
                double f = fg[i + l] * lrRow[seq[pos - (i + l) - 1]];
                lane[l] = lane[l] + b2f[i + l] * f;
                fg[i + l] = toBackground + f2f0 * f;
            }
        }
        double toForeground;
        if (W == 4) toForeground = (lane[0] + lane[2]) + (lane[1] + lane[3]);
        else if (W == 2) toForeground = lane[0] + lane[1];
        else toForeground = lane[0];
        for (; i < maxOffset; i++) {
            double f = fg[i] * lrRow[seq[pos - i - 1]];
            toForeground += b2f[i] * f;
            fg[i] = toBackground + f2f0 * f;
        }
        background = b2b * background + toForeground;
    }
    /* maskProbableLetters :513-527 with hardMaskTable == X, then Masker::finalizeMasking */
    int masked = 0;
    for (int pos = 0; pos < L; pos++) {
        if (probs[pos] >= minMaskProb) { seq[pos] = MKO_X; masked++; }
    }
    free(probs);
    free(scaleFactors);
    return masked;
}

typedef struct { uint32_t kmer; uint16_t pos; } kp_t;
static int kp_cmp(const void *a, const void *b) {
    const kp_t *x = (const kp_t *) a, *y = (const kp_t *) b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}

static const int SPACED6[6] = {0, 1, 3, 5, 8, 9};   /* spaced_seed_6 = 1101010011 (Sequence.h:23) */

/* IndexBuilder::fillDatabase (M/src/prefiltering/IndexBuilder.cpp:55-239) for AA targets, k=6:
 * pass 1 mask + SequenceLookup + IndexTable::addKmerCount (IndexTable.h:133-173), prefix sums
 * (IndexTable::init :220-229), pass 2 IndexTable::addSequence (:348-401) on the MASKED residues,
 * lists sorted by (seqId,pos) (:182-189).  Per (k-mer, sequence) only the smallest position is kept;
 * k-mers with an X or with self-score < kmer_thr are skipped. */
mko_index *mko_index_build(const mko_submat *km, const uint8_t *residues, const uint64_t *seq_off,
                           uint32_t n_seq, int kmer_thr, int mask, int simd_lanes) {
    return mko_index_build_k(km, residues, seq_off, n_seq, kmer_thr, mask, simd_lanes, 6);
}

void mko_index_list(const mko_index *ix, uint64_t kmer, uint64_t *o0, uint64_t *o1) {
    if (!ix->kmers) { *o0 = ix->offsets[kmer]; *o1 = ix->offsets[kmer + 1]; return; }
    uint64_t lo = 0, hi = ix->n_kmers;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (ix->kmers[mid] < kmer) lo = mid + 1; else hi = mid; }
    if (lo < ix->n_kmers && ix->kmers[lo] == kmer) { *o0 = ix->offsets[lo]; *o1 = ix->offsets[lo + 1]; }
    else { *o0 = 0; *o1 = 0; }
}

typedef struct { uint64_t kmer; uint32_t seq; uint16_t pos; } ksp_t;
static int ksp_cmp(const void *a, const void *b) {
    const ksp_t *x = (const ksp_t *) a, *y = (const ksp_t *) b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    if (x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
    return 0;
}

/* k = 7: the same table (20^7 cells) kept sparse -- the distinct k-mers in ascending order with their lists; same entries in the same
 * order as the dense build (per sequence the smallest position of a k-mer, lists by sequence). */
static void build_sparse(mko_index *ix, const mko_submat *km, int kmer_thr) {
    const int *sp; const int span = mko_spaced_pattern(ix->k, &sp);
    char idScore[MKO_ALPH];
    for (int a = 0; a < MKO_ALPH; a++) idScore[a] = (char) km->sub[a][a];
    size_t cap = 1 << 20, n = 0;
    ksp_t *all = (ksp_t *) malloc(cap * sizeof(ksp_t));
    for (uint32_t s = 0; s < ix->n_seq; s++) {
        const uint8_t *seq = ix->masked + ix->seq_off[s];
        const int L = (int) (ix->seq_off[s + 1] - ix->seq_off[s]);
        const size_t first = n;
        for (int i = 0; i + span <= L; i++) {
            int hasX = 0, score = 0;
            uint64_t idx = 0, pw = 1;
            for (int p = 0; p < ix->k; p++) { uint8_t c = seq[i + sp[p]]; hasX |= (c == MKO_X); score += idScore[c]; idx += c * pw; pw *= 20; }
            if (hasX || (kmer_thr > 0 && score < kmer_thr)) continue;
            if (n == cap) { cap *= 2; all = (ksp_t *) realloc(all, cap * sizeof(ksp_t)); }
            all[n].kmer = idx; all[n].seq = s; all[n].pos = (uint16_t) i; n++;
        }
        /* per (k-mer, sequence) only the smallest position: positions ascend, so keep the first of equal k-mers after a stable grouping */
        if (n - first > 1) {
            qsort(all + first, n - first, sizeof(ksp_t), ksp_cmp);   /* ties (same k-mer, same seq): any order, the minimum is taken below */
            size_t w = first;
            for (size_t r = first; r < n; ) {
                size_t e = r; uint16_t mn = all[r].pos;
                while (e < n && all[e].kmer == all[r].kmer) { if (all[e].pos < mn) mn = all[e].pos; e++; }
                all[w] = all[r]; all[w].pos = mn; w++;
                r = e;
            }
            n = w;
        }
    }
    if (n > 1) qsort(all, n, sizeof(ksp_t), ksp_cmp);
    ix->n_entries = n;
    ix->seq_id = (uint32_t *) malloc((n + 1) * sizeof(uint32_t));
    ix->pos = (uint16_t *) malloc((n + 1) * sizeof(uint16_t));
    ix->kmers = (uint64_t *) malloc((n + 1) * sizeof(uint64_t));
    ix->offsets = (uint64_t *) malloc((n + 2) * sizeof(uint64_t));
    uint64_t nk = 0;
    for (size_t j = 0; j < n; j++) {
        if (j == 0 || all[j].kmer != all[j - 1].kmer) { ix->kmers[nk] = all[j].kmer; ix->offsets[nk] = j; nk++; }
        ix->seq_id[j] = all[j].seq; ix->pos[j] = all[j].pos;
    }
    ix->offsets[nk] = n;
    ix->n_kmers = nk;
    free(all);
}

mko_index *mko_index_build_k(const mko_submat *km, const uint8_t *residues, const uint64_t *seq_off,
                             uint32_t n_seq, int kmer_thr, int mask, int simd_lanes, int k) {
    mko_index *ix = (mko_index *) calloc(1, sizeof(*ix));
    ix->k = k;
    ix->table_size = k == 7 ? 1280000000ull : 64000000ull;
    if (k == 7) {
        ix->n_seq = n_seq;
        const uint64_t total7 = seq_off[n_seq];
        ix->masked = (uint8_t *) malloc(total7 + 1);
        ix->seq_off = (uint64_t *) malloc((n_seq + 1) * sizeof(uint64_t));
        memcpy(ix->masked, residues, total7);
        memcpy(ix->seq_off, seq_off, (n_seq + 1) * sizeof(uint64_t));
        uint64_t maskedRes7 = 0;
        if (mask) {
#pragma omp parallel for schedule(dynamic, 100) reduction(+: maskedRes7)
            for (uint32_t s = 0; s < n_seq; s++)
                maskedRes7 += (uint64_t) mko_tantan_mask(km, ix->masked + seq_off[s], (int) (seq_off[s + 1] - seq_off[s]), (double) 0.9f, simd_lanes);
        }
        ix->masked_residues = maskedRes7;
        build_sparse(ix, km, kmer_thr);
        return ix;
    }
    ix->n_seq = n_seq;
    const uint64_t total = seq_off[n_seq];
    ix->masked = (uint8_t *) malloc(total + 1);
    ix->seq_off = (uint64_t *) malloc((n_seq + 1) * sizeof(uint64_t));
    memcpy(ix->masked, residues, total);
    memcpy(ix->seq_off, seq_off, (n_seq + 1) * sizeof(uint64_t));
    ix->offsets = (uint64_t *) calloc(ix->table_size + 1, sizeof(uint64_t));
    char idScore[MKO_ALPH];
    for (int a = 0; a < MKO_ALPH; a++) idScore[a] = (char) km->sub[a][a];   /* getScoreLookup :10-22 */
    uint64_t maskedRes = 0;
    if (mask) {
#pragma omp parallel for schedule(dynamic, 100) reduction(+: maskedRes)
        for (uint32_t s = 0; s < n_seq; s++)
            maskedRes += (uint64_t) mko_tantan_mask(km, ix->masked + seq_off[s], (int) (seq_off[s + 1] - seq_off[s]), (double) 0.9f /* float --mask-prob widened, Prefiltering.cpp:39 */, simd_lanes);
    }
    ix->masked_residues = maskedRes;
    /* collect (kmer,pos) per sequence; sequential so the lists come out seqId-sorted */
    size_t cap = 1 << 16;
    kp_t *buf = (kp_t *) malloc(cap * sizeof(kp_t));
    /* pass 1: count */
    for (int pass = 0; pass < 2; pass++) {
        uint64_t *cursor = NULL;
        if (pass == 1) {
            uint64_t off = 0;
            for (uint64_t k = 0; k < ix->table_size; k++) { uint64_t c = ix->offsets[k]; ix->offsets[k] = off; off += c; }
            ix->offsets[ix->table_size] = off;
            ix->n_entries = off;
            ix->seq_id = (uint32_t *) malloc((off + 1) * sizeof(uint32_t));
            ix->pos = (uint16_t *) malloc((off + 1) * sizeof(uint16_t));
            cursor = (uint64_t *) malloc(ix->table_size * sizeof(uint64_t));
            memcpy(cursor, ix->offsets, ix->table_size * sizeof(uint64_t));
        }
        for (uint32_t s = 0; s < n_seq; s++) {
            const uint8_t *seq = ix->masked + seq_off[s];
            const int L = (int) (seq_off[s + 1] - seq_off[s]);
            size_t n = 0;
            if ((size_t) L + 1 > cap) { cap = (size_t) L + 1; buf = (kp_t *) realloc(buf, cap * sizeof(kp_t)); }
            for (int i = 0; i + 10 <= L; i++) {       /* Sequence::hasNextKmer: (i+1)+span <= L+... (Sequence.h:92-94) */
                int hasX = 0, score = 0;
                uint32_t idx = 0, pw = 1;
                for (int p = 0; p < 6; p++) {
                    uint8_t c = seq[i + SPACED6[p]];
                    hasX |= (c == MKO_X);
                    score += idScore[c];
                    idx += c * pw;
                    pw *= 20;
                }
                if (hasX) continue;
                if (kmer_thr > 0 && score < kmer_thr) continue;
                buf[n].kmer = idx;
                buf[n].pos = (uint16_t) i;
                n++;
            }
            if (n > 1) qsort(buf, n, sizeof(kp_t), kp_cmp);
            uint32_t prev = 0xFFFFFFFFu;
            for (size_t j = 0; j < n; j++) {
                if (buf[j].kmer != prev) {
                    if (pass == 0) ix->offsets[buf[j].kmer]++;
                    else {
                        uint64_t o = cursor[buf[j].kmer]++;
                        ix->seq_id[o] = s;
                        ix->pos[o] = buf[j].pos;
                    }
                }
                prev = buf[j].kmer;
            }
        }
        if (cursor) free(cursor);
    }
    free(buf);
    return ix;
}

void mko_index_free(mko_index *ix) {
    if (!ix) return;
    free(ix->offsets); free(ix->kmers); free(ix->seq_id); free(ix->pos); free(ix->masked); free(ix->seq_off);
    free(ix);
}
