/* oracle/mko_profile.c -- TEST INFRASTRUCTURE (parity oracle).  See mko.h.
 * The profile-target path of the reference: M/data/workflow/searchslicedtargetprofile.sh makes the
 * PROFILES the queries of prefilter/align and the 6-frame fragments the indexed targets, then swapresults turns the lists round.
 * Restated here: Sequence::mapProfile, Util::rankedDescSort20, the profile divide strategy of KmerGenerator, the PROFILE_SEQ
 * Smith-Waterman (same striped passes, scores from the position-specific profile), and Matcher::result_t::swapResult. */
#include "mko.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Util::rankedDescSort20 (M/src/commons/Util.cpp:88-114): a fixed compare-exchange network on (score, index) pairs; a pair is
 * exchanged only when val[x] < val[y], which fixes the order of equal scores. */
static const unsigned char SORT20[][2] = {
    {0,16},{1,17},{2,18},{3,19},{4,12},{5,13},{6,14},{7,15},
    {0,8},{1,9},{2,10},{3,11},
    {8,16},{9,17},{10,18},{11,19},{0,4},{1,5},{2,6},{3,7},
    {8,12},{9,13},{10,14},{11,15},{4,16},{5,17},{6,18},{7,19},{0,2},{1,3},
    {4,8},{5,9},{6,10},{7,11},{12,16},{13,17},{14,18},{15,19},{0,1},
    {4,6},{5,7},{8,10},{9,11},{12,14},{13,15},{16,18},{17,19},
    {2,16},{3,17},{6,12},{7,13},{18,19},
    {2,8},{3,9},{10,16},{11,17},
    {2,4},{3,5},{6,8},{7,9},{10,12},{11,13},{14,16},{15,17},
    {2,3},{4,5},{6,7},{8,9},{10,11},{12,13},{14,15},{16,17},
    {1,16},{3,18},{5,12},{7,14},
    {1,8},{3,10},{9,16},{11,18},
    {1,4},{3,6},{5,8},{7,10},{9,12},{11,14},{13,16},{15,18},
    {1,2},{3,4},{5,6},{7,8},{9,10},{11,12},{13,14},{15,16},{17,18}};

void mko_ranked_desc_sort20(short *val, uint8_t *index) {
    for (size_t k = 0; k < sizeof(SORT20) / sizeof(SORT20[0]); k++) {
        const int x = SORT20[k][0], y = SORT20[k][1];
        if (val[x] < val[y]) {
            short t1 = val[x]; val[x] = val[y]; val[y] = t1;
            uint8_t t2 = index[x]; index[x] = index[y]; index[y] = t2;
        }
    }
}

/* Sequence::mapProfile (M/src/commons/Sequence.cpp:241-292).  One column = PROFILE_READIN_SIZE = 25 bytes (Sequence.h:458-471):
 * 20 signed scores, the query letter, the consensus letter, neff, two gap bytes.  profile_for_alignment = score / 4 (C division,
 * towards zero) laid out [aa][pos], the X row 0 (:272-280); for the prefilter every column's scores are sorted descending with
 * their residue numbers (:283-291). */
mko_profile *mko_profile_map(const char *data, int seqLen) {
    mko_profile *p = (mko_profile *) calloc(1, sizeof(mko_profile));
    const int L = seqLen;
    p->L = L;
    p->query = (uint8_t *) malloc((size_t) L + 1);
    p->consensus = (uint8_t *) malloc((size_t) L + 1);
    p->aln = (int8_t *) calloc((size_t) 21 * (size_t) (L > 0 ? L : 1), 1);
    p->sorted_score = (short *) malloc((size_t) (L > 0 ? L : 1) * 20 * sizeof(short));
    p->sorted_idx = (uint8_t *) malloc((size_t) (L > 0 ? L : 1) * 20);
    for (int l = 0; l < L; l++) {
        const char *col = data + (size_t) l * 25;
        for (int a = 0; a < 20; a++) {
            const short s = (short) (signed char) col[a];
            p->sorted_score[l * 20 + a] = s;
            p->sorted_idx[l * 20 + a] = (uint8_t) a;
            p->aln[(size_t) a * L + l] = (int8_t) (s / 4);
        }
        p->query[l] = (uint8_t) col[20];
        p->consensus[l] = (uint8_t) col[21];
        mko_ranked_desc_sort20(p->sorted_score + l * 20, p->sorted_idx + l * 20);
    }
    return p;
}

void mko_profile_free(mko_profile *p) {
    if (!p) return;
    free(p->query); free(p->consensus); free(p->aln); free(p->sorted_score); free(p->sorted_idx); free(p);
}

/* KmerGenerator::generateKmerList with setDivideStrategy(ScoreMatrix **) (M/src/prefiltering/KmerGenerator.cpp:30-39, 107-216):
 * six steps of one position each (Sequence::nextProfileKmer, Sequence.cpp:294-305, points them at the sorted columns under the spaced
 * pattern); the k-mer window is zeroed for profiles (Sequence.h:404-410), so highestScorePerArray[i] is the first = best score of
 * column i.  Step by step the partial list is multiplied with the next column, a partner is taken while
 * score_j >= threshold - score_i - possibleRest[i+1]; the first list is cut at threshold - possibleRest[0], later ones are walked
 * whole (cutoff1 = -1000).  Index = sum of residue * 20^i (the index table's alphabet, QueryMatcher.cpp:34). */
size_t mko_profile_kmer_list(const mko_profile *p, int pos, short threshold, uint64_t *out, size_t cap) {
    static const int SP[6] = {0, 1, 3, 5, 8, 9};
    const size_t MAX_KMER_RESULT_SIZE = 262144 * 32;
    const short *sc[6]; const uint8_t *ix[6];
    short possibleRest[6];
    uint64_t pw[6];
    for (int i = 0; i < 6; i++) {
        sc[i] = p->sorted_score + (size_t) (pos + SP[i]) * 20;
        ix[i] = p->sorted_idx + (size_t) (pos + SP[i]) * 20;
        pw[i] = i ? pw[i - 1] * 20u : 1u;
    }
    possibleRest[5] = 0;
    for (int i = 5; i >= 1; i--) possibleRest[i - 1] = (short) (sc[i][0] + possibleRest[i]);
    /* the lists are small (a few hundred k-mers): two heap buffers that grow on demand */
    size_t bufCap = 4096;
    short *sA = (short *) malloc(bufCap * sizeof(short)), *sB = (short *) malloc(bufCap * sizeof(short));
    uint64_t *kA = (uint64_t *) malloc(bufCap * sizeof(uint64_t)), *kB = (uint64_t *) malloc(bufCap * sizeof(uint64_t));
    short cutoff1 = (short) (threshold - possibleRest[0]);
    size_t n = 0;
    for (int j = 0; j < 20 && sc[0][j] >= cutoff1; j++) { sA[n] = sc[0][j]; kA[n] = ix[0][j]; n++; }
    /* (the reference copies only the indices of the first list and reads the scores from the column itself, :131-135; the loop over
     *  array1 then breaks at score_i < cutoff1 -- the same cut) */
    for (int i = 0; i < 5; i++) {
        size_t counter = 0;
        int full = 0;
        for (size_t a = 0; a < n && !full; a++) {
            const short score_i = sA[a];
            if (score_i < cutoff1) break;
            const short cutoff2 = (short) (threshold - score_i - possibleRest[i + 1]);
            for (int j = 0; j < 20 && (counter + 1 < MAX_KMER_RESULT_SIZE) && sc[i + 1][j] >= cutoff2; j++) {
                if (counter == bufCap) {
                    bufCap *= 2;
                    sA = (short *) realloc(sA, bufCap * sizeof(short)); sB = (short *) realloc(sB, bufCap * sizeof(short));
                    kA = (uint64_t *) realloc(kA, bufCap * sizeof(uint64_t)); kB = (uint64_t *) realloc(kB, bufCap * sizeof(uint64_t));
                }
                sB[counter] = (short) (score_i + sc[i + 1][j]);
                kB[counter] = kA[a] + (uint64_t) ix[i + 1][j] * pw[i + 1];
                counter++;
            }
            if (counter + 1 >= MAX_KMER_RESULT_SIZE) full = 1;
        }
        short *ts = sA; sA = sB; sB = ts;
        uint64_t *tk = kA; kA = kB; kB = tk;
        n = counter;
        cutoff1 = -1000;
    }
    for (size_t k = 0; k < n && k < cap; k++) out[k] = kA[k];
    free(sA); free(sB); free(kA); free(kB);
    return n;
}

/* SmithWaterman::ssw_init for a profile query (StripedSmithWaterman.cpp:1216-1345): no composition bias, the byte-mode bias is
 * |min| over the 20 x L alignment profile (:1275-1284, matSize = L * PROFILE_AA_SIZE). */
int mko_sw_profile_bias(const mko_profile *p) {
    int b = 0;
    for (size_t i = 0; i < (size_t) p->L * 20; i++) if (p->aln[i] < b) b = p->aln[i];
    return abs(b);
}

/* Matcher::result_t::swapResult (M/src/alignment/Matcher.h:93-115) on a parsed 10-column record + the re-print of
 * util/swapresults.cpp:283-333.  `ev` must be initialised with the amino-acid size of the PROFILE DB (swapresults.cpp:76-77,
 * DBReader::getAminoAcidDBSize: dataSize / 25 - entries). */
void mko_swap_result(const mko_evaluer *ev, mko_aln_result *r, uint32_t new_db_key) {
    const double rawScore = (ev->logK + (double) r->bit_score * log(2.0)) / ev->lambda;   /* EvalueComputation.h:22-24 */
    r->evalue = mko_evalue(ev, rawScore, (double) r->db_len);
    const int qs = r->q_start, qe = r->q_end, ql = r->q_len;
    r->q_start = r->db_start; r->q_end = r->db_end; r->q_len = r->db_len;
    r->db_start = qs; r->db_end = qe; r->db_len = ql;
    r->db_key = new_db_key;
}
