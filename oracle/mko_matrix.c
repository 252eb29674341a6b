/* oracle/mko_matrix.c -- TEST INFRASTRUCTURE (parity oracle).  See mko.h.
 * Substitution matrices, sequence encoding, composition bias, extended
 * 3-mer score table and the similar-k-mer generator. */
#include "mko.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "../metaeuk_amd/data/matrices.inc"

/* SubstitutionMatrix ctor (M/src/commons/SubstitutionMatrix.cpp:12-57):
 * readProbMatrix (:326-404) + BaseMatrix::generateSubMatrix (BaseMatrix.cpp:110-159). */
void mko_submat_init(mko_submat *m, int which, float bit_factor, float score_bias) {
    const double (*S)[MKO_ALPH] = which == MKO_MAT_BLOSUM62 ? MK_BLOSUM62_SCORES : MK_VTML80_SCORES;
    const double *bg = which == MKO_MAT_BLOSUM62 ? MK_BLOSUM62_BACKGROUND : MK_VTML80_BACKGROUND;
    m->lambda = which == MKO_MAT_BLOSUM62 ? MK_BLOSUM62_LAMBDA : MK_VTML80_LAMBDA;
    m->name = which == MKO_MAT_BLOSUM62 ? "blosum62.out" : "VTML80.out";
    for (int i = 0; i < MKO_ALPH; i++) m->pback[i] = bg[i];
    /* :389-393 X is never positive in these matrices -> rescale the 20 real letters */
    for (int i = 0; i < MKO_ALPH - 1; i++) m->pback[i] = m->pback[i] * (1.0 - m->pback[MKO_X]);
    /* :395-401 */
    for (int i = 0; i < MKO_ALPH; i++)
        for (int j = 0; j < MKO_ALPH; j++)
            m->prob[i][j] = exp(m->lambda * S[i][j]) * m->pback[i] * m->pback[j];
    /* BaseMatrix::computeBackground (BaseMatrix.cpp:97-108) -- a LOCAL background from row sums */
    double pb[MKO_ALPH];
    for (int i = 0; i < MKO_ALPH; i++) {
        pb[i] = 0;
        for (int j = 0; j < MKO_ALPH; j++) pb[i] += m->prob[i][j];
    }
    pb[MKO_ALPH - 1] = 1E-5; /* ANY_BACK */
    const double bf = (double) bit_factor, sb = (double) score_bias;
    for (int i = 0; i < MKO_ALPH; i++) {
        for (int j = 0; j < MKO_ALPH; j++) {
            double sm = log2(m->prob[i][j] / (pb[i] * pb[j]));
            double v = (bf * sm + sb);
            m->sub[i][j] = (short) ((v < 0.0) ? v - 0.5 : v + 0.5);   /* BaseMatrix.cpp:150-151 */
        }
    }
}

/* aa2num: SubstitutionMatrix::setupLetterMapping (SubstitutionMatrix.cpp:138-179) over the
 * alphabet order of the matrix header, used by Sequence::mapSequence (Sequence.cpp:307-324). */
void mko_map_sequence(const char *seq, int len, uint8_t *codes) {
    static uint8_t table[256];
    static int init = 0;
    if (!init) {
        const char *alpha = MK_BLOSUM62_ALPHABET;
        uint8_t base[256];
        memset(base, 255, sizeof(base));
        for (int i = 0; alpha[i]; i++) base[(unsigned char) alpha[i]] = (uint8_t) i;
        for (int c = 0; c < 256; c++) {
            int up = toupper(c);
            uint8_t v;
            switch (up) {
                case 'A': case 'T': case 'G': case 'C': case 'D': case 'E': case 'F': case 'H': case 'I': case 'K':
                case 'L': case 'M': case 'N': case 'P': case 'Q': case 'R': case 'S': case 'V': case 'W': case 'Y':
                case 'X': v = base[up]; break;
                case 'J': v = base['L']; break;
                case 'U': case 'O': v = base['X']; break;
                case 'Z': v = base['E']; break;
                case 'B': v = base['D']; break;
                default: v = base['X']; break;
            }
            table[c] = v;
        }
        init = 1;
    }
    for (int i = 0; i < len; i++) codes[i] = table[(unsigned char) seq[i]];
}

/* SubstitutionMatrix::calcLocalAaBiasCorrection (SubstitutionMatrix.cpp:79-109).
 * The float/double mix is deliberate and mirrors the C++ expression types. */
void mko_comp_bias(const mko_submat *m, const uint8_t *seq, int L, float scale, float *bias) {
    const int windowSize = 40;
    for (int i = 0; i < L; i++) {
        const int minPos = (i - windowSize / 2) > 0 ? (i - windowSize / 2) : 0;
        const int maxPos = (i + windowSize / 2) < L ? (i + windowSize / 2) : L;
        const int windowLength = maxPos - minPos;
        int sumSubScores = 0;
        const short *subMat = m->sub[seq[i]];
        for (int j = minPos; j < maxPos; j++) sumSubScores += subMat[seq[j]];
        sumSubScores -= subMat[seq[i]];
        float deltaS_i = (float) sumSubScores;
        deltaS_i = (float) ((double) deltaS_i / (-1.0 * (double) ((float) windowLength)));
        for (int a = 0; a < MKO_ALPH; a++)
            deltaS_i = (float) ((double) deltaS_i + m->pback[a] * (double) ((float) subMat[a]));
        bias[i] = scale * deltaS_i;
    }
}

/* ExtendedSubstitutionMatrix::calcScoreMatrix (ExtendedSubstitutionMatrix.cpp:20-69) for alphabet 20.
 * Row i holds all k-mers j stable-sorted by descending score; the pre-sort order is the cartesian
 * product order of createCartesianProduct (:100-123): FIRST letter slowest. */
mko_scoremat *mko_scoremat_build(const mko_submat *m, int kmer) {
    const int A = 20;
    int size = 1;
    for (int i = 0; i < kmer; i++) size *= A;
    int row = (size / 64 + 1) * 64;
    mko_scoremat *s = (mko_scoremat *) calloc(1, sizeof(*s));
    s->element_size = size;
    s->row_size = row;
    s->score = (short *) malloc((size_t) size * row * sizeof(short));
    s->index = (uint32_t *) malloc((size_t) size * row * sizeof(uint32_t));
    /* permutation p (cartesian order) -> letters and Indexer index (Indexer.h:21-45: sum a_p*20^p) */
    uint8_t *letters = (uint8_t *) malloc((size_t) size * kmer);
    uint32_t *pidx = (uint32_t *) malloc((size_t) size * sizeof(uint32_t));
    for (int p = 0; p < size; p++) {
        int rem = p;
        uint32_t idx = 0, pw = 1;
        uint8_t tmp[8];
        for (int d = kmer - 1; d >= 0; d--) { tmp[d] = (uint8_t) (rem % A); rem /= A; }
        for (int d = 0; d < kmer; d++) { letters[(size_t) p * kmer + d] = tmp[d]; idx += tmp[d] * pw; pw *= A; }
        pidx[p] = idx;
    }
#pragma omp parallel
    {
        short *sc = (short *) malloc((size_t) size * sizeof(short));
        int *cnt = (int *) malloc(65536 * sizeof(int));
#pragma omp for schedule(static)
        for (int i = 0; i < size; i++) {
            const uint8_t *li = letters + (size_t) i * kmer;
            int mn = 32767, mx = -32768;
            for (int j = 0; j < size; j++) {
                const uint8_t *lj = letters + (size_t) j * kmer;
                short v = 0;
                for (int d = 0; d < kmer; d++) v = (short) (v + m->sub[li[d]][lj[d]]);
                sc[j] = v;
                if (v < mn) mn = v;
                if (v > mx) mx = v;
            }
            /* stable sort by descending score == counting sort on (mx - score) */
            int range = mx - mn + 1;
            if (range < 1 || range > 65535) range = 1;
            memset(cnt, 0, (size_t) (range + 1) * sizeof(int));
            for (int j = 0; j < size; j++) cnt[mx - sc[j] + 1]++;
            for (int r = 0; r < range; r++) cnt[r + 1] += cnt[r];
            size_t base = (size_t) pidx[i] * row;
            for (int j = 0; j < size; j++) {
                int pos = cnt[mx - sc[j]]++;
                s->score[base + pos] = sc[j];
                s->index[base + pos] = pidx[j];
            }
            for (int z = size; z < row; z++) { s->score[base + z] = -255; s->index[base + z] = 0; }
        }
        free(sc);
        free(cnt);
    }
    free(letters);
    free(pidx);
    return s;
}

void mko_scoremat_free(mko_scoremat *s) {
    if (!s) return;
    free(s->score);
    free(s->index);
    free(s);
}

/* KmerGenerator::generateKmerList + calculateArrayProduct (KmerGenerator.cpp:107-216) with the
 * k=6 divide strategy {3,3} (setDivideStrategy :41-86).  `kmer` = the 6 residues under the spaced
 * pattern.  Output indices = idx(first 3) + idx(last 3) * 20^3, in the order the reference emits. */
size_t mko_kmer_list6(const mko_scoremat *three, const uint8_t *kmer, short threshold, uint64_t *out, size_t cap) {
    const size_t MAX_KMER_RESULT_SIZE = 262144 * 32;
    const int row = three->row_size;
    const uint32_t index0 = kmer[0] + 20u * kmer[1] + 400u * kmer[2];
    const uint32_t index1 = kmer[3] + 20u * kmer[4] + 400u * kmer[5];
    const short *s0 = three->score + (size_t) index0 * row;
    const uint32_t *i0 = three->index + (size_t) index0 * row;
    const short *s1 = three->score + (size_t) index1 * row;
    const uint32_t *i1 = three->index + (size_t) index1 * row;
    const short possibleRest0 = s1[0];                       /* highestScorePerArray[1] + possibleRest[1](=0) */
    const short cutoff1 = (short) (threshold - possibleRest0);
    const size_t n1 = (size_t) three->element_size, n2 = (size_t) three->element_size;
    size_t counter = 0;
    for (size_t i = 0; i < n1; i++) {
        const short score_i = s0[i];
        if (score_i < cutoff1) break;
        const short cutoff2 = (short) (threshold - score_i - 0);
        for (size_t j = 0; j < n2 && (counter + 1 < MAX_KMER_RESULT_SIZE) && (s1[j] >= cutoff2); j++) {
            if (counter < cap) out[counter] = (uint64_t) i0[i] + (uint64_t) i1[j] * 8000u;
            counter++;
        }
        if (counter + 1 >= MAX_KMER_RESULT_SIZE) return counter;
    }
    return counter;
}

/* k = 7: KmerGenerator::setDivideStrategy (KmerGenerator.cpp:41-86) gives the steps {3,2,2} and reverses them: a 2-mer, a 2-mer and a
 * 3-mer row (positions 0-1, 2-3, 4-6 of the window; multipliers 20^0, 20^2, 20^4).  generateKmerList (:107-187) multiplies the first two
 * rows (the first one cut at threshold - best of the others), then the partial list -- walked whole, cutoff1 = -1000 -- with the third. */
size_t mko_kmer_list7(const mko_scoremat *two, const mko_scoremat *three, const uint8_t *kmer, short threshold, uint64_t *out, size_t cap) {
    const size_t MAX_KMER_RESULT_SIZE = 262144 * 32;
    const uint32_t index0 = kmer[0] + 20u * kmer[1], index1 = kmer[2] + 20u * kmer[3], index2 = kmer[4] + 20u * kmer[5] + 400u * kmer[6];
    const short *s0 = two->score + (size_t) index0 * two->row_size, *s1 = two->score + (size_t) index1 * two->row_size;
    const short *s2 = three->score + (size_t) index2 * three->row_size;
    const uint32_t *i0 = two->index + (size_t) index0 * two->row_size, *i1 = two->index + (size_t) index1 * two->row_size;
    const uint32_t *i2 = three->index + (size_t) index2 * three->row_size;
    const short rest1 = s2[0], rest0 = (short) (s1[0] + rest1);
    short cutoff1 = (short) (threshold - rest0);
    size_t capA = 4096, nA = 0;
    short *scA = (short *) malloc(capA * sizeof(short));
    uint64_t *kmA = (uint64_t *) malloc(capA * sizeof(uint64_t));
    for (int a = 0; a < two->element_size; a++) {
        const short score_i = s0[a];
        if (score_i < cutoff1) break;
        const short cutoff2 = (short) (threshold - score_i - rest1);
        for (int b = 0; b < two->element_size && (nA + 1 < MAX_KMER_RESULT_SIZE) && s1[b] >= cutoff2; b++) {
            if (nA == capA) { capA *= 2; scA = (short *) realloc(scA, capA * sizeof(short)); kmA = (uint64_t *) realloc(kmA, capA * sizeof(uint64_t)); }
            scA[nA] = (short) (score_i + s1[b]);
            kmA[nA] = (uint64_t) i0[a] + (uint64_t) i1[b] * 400u;
            nA++;
        }
        if (nA + 1 >= MAX_KMER_RESULT_SIZE) break;
    }
    cutoff1 = -1000;
    size_t counter = 0;
    for (size_t e = 0; e < nA; e++) {
        const short score_i = scA[e];
        if (score_i < cutoff1) break;
        const short cutoff2 = (short) (threshold - score_i - 0);
        for (int c = 0; c < three->element_size && (counter + 1 < MAX_KMER_RESULT_SIZE) && s2[c] >= cutoff2; c++) {
            if (counter < cap) out[counter] = kmA[e] + (uint64_t) i2[c] * 160000u;
            counter++;
        }
        if (counter + 1 >= MAX_KMER_RESULT_SIZE) break;
    }
    free(scA); free(kmA);
    return counter;
}

int mko_spaced_pattern(int k, const int **offsets) {
    static const int P6[6] = {0, 1, 3, 5, 8, 9};          /* spaced_seed_6 = 1101010011 */
    static const int P7[7] = {0, 1, 3, 5, 6, 9, 10};      /* spaced_seed_7 = 11010110011 */
    *offsets = k == 7 ? P7 : P6;
    return k == 7 ? 11 : 10;
}
