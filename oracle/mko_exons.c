/* oracle/mko_exons.c -- TEST INFRASTRUCTURE (parity oracle), not product code.
 *
 * Plain-C restatement of `resultspercontig` + `collectoptimalset` with predictexons' defaults:
 *   joining ORF->target alignments with ORF->contig locations, sorted by (target, orf)   src/exonpredictor/resultspercontig.cpp:145-190
 *   PotentialExon::setByAln / comparePotentialExons / exonToBuffer                       src/commons/PredictionParser.h:15-186
 *   isPairCompatible / getPenaltyForProtCoords / findoptimalsetbydp                      src/exonpredictor/collectoptimalset.cpp:33-217
 *   the per-contig, target-by-target loop and the combined e-value                       src/exonpredictor/collectoptimalset.cpp:262-413
 *   Prediction (low/high contig coordinate) and predictionToBuffer                       src/commons/PredictionParser.h:189-215,357-384
 * Pinned against oracle/_ref/ref_harness `exons` (the reference's own collectoptimalset.cpp / PredictionParser.h).
 * The reference hands alignments over as TEXT: sequence identity and e-value are what strtod reads back from the
 * printed columns, which is reproduced here.
 */
#include "mko.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    unsigned int exonKey, targetKey; int strand;
    unsigned int bitScore; double seqId, evalue;
    int targetMatchStart, targetMatchEnd, targetLen; double targetCov;
    int contigStart, contigEnd, nucleotideLen, aaLen;
    int startBeforeTrim, endBeforeTrim;
    int isUsed;
} exon_t;

/* PotentialExon::setByAln (PredictionParser.h:15-62) from the alignment columns and the ORF's header coordinates */
static void exon_set(exon_t *e, const mko_exon_aln *a, unsigned int orfKey, int orfFrom, int orfTo) {
    e->targetKey = a->target; e->bitScore = (unsigned int) a->bit_score; e->seqId = a->seq_id; e->evalue = a->evalue;
    e->targetMatchStart = a->db_start; e->targetMatchEnd = a->db_end; e->targetLen = a->db_len;
    e->exonKey = orfKey; e->startBeforeTrim = orfFrom; e->endBeforeTrim = orfTo;
    if (orfFrom < orfTo) {
        e->contigStart = orfFrom + a->q_start * 3; e->contigEnd = orfFrom + a->q_end * 3 + 2; e->strand = 1;
    } else {
        e->contigStart = -1 * (orfFrom - a->q_start * 3); e->contigEnd = -1 * (orfFrom - a->q_end * 3 - 2); e->strand = -1;
    }
    e->nucleotideLen = e->contigEnd - e->contigStart + 1;
    e->aaLen = e->nucleotideLen / 3;
    e->targetCov = (double) (e->targetMatchEnd - e->targetMatchStart + 1) / e->targetLen;
    e->isUsed = 0;
}

static int exon_less(const exon_t *a, const exon_t *b) {                      /* comparePotentialExons (:139-153) */
    if (a->isUsed != b->isUsed) return a->isUsed < b->isUsed;
    if (a->contigStart != b->contigStart) return a->contigStart < b->contigStart;
    if (a->contigEnd != b->contigEnd) return a->contigEnd < b->contigEnd;
    return 0;
}
static void stable_sort_exons(exon_t *v, size_t n) {                           /* insertion sort: stable like std::stable_sort */
    for (size_t i = 1; i < n; i++) {
        exon_t x = v[i];
        size_t j = i;
        while (j > 0 && exon_less(&x, &v[j - 1])) { v[j] = v[j - 1]; j--; }
        v[j] = x;
    }
}

static int pair_compatible(const exon_t *f, const exon_t *s, size_t minIntron, size_t maxIntron, size_t maxAaOverlap, size_t *aaOverlap) {
    if (f->strand != s->strand) return 0;
    if (s->contigEnd < f->contigEnd) return 0;
    const int diffOnContig = s->contigStart - f->contigEnd - 1;
    if (diffOnContig < 0) return 0;
    const size_t d = (size_t) abs(diffOnContig);
    if (d < minIntron || d > maxIntron) return 0;
    const int diffAAs = s->targetMatchStart - f->targetMatchEnd - 1;
    *aaOverlap = 0;
    if (diffAAs < 0) { *aaOverlap = (size_t) abs(diffAAs); if (*aaOverlap > maxAaOverlap) return 0; }
    if (s->targetMatchStart < f->targetMatchStart) return 0;
    return 1;
}
static int transition_penalty(const exon_t *prev, const exon_t *curr, int gapOpen, int gapExtend) {
    const int diffAAs = curr->targetMatchStart - prev->targetMatchEnd - 1;
    if (diffAAs < 0) return gapOpen + gapExtend * (abs(diffAAs) - 1);
    if (diffAAs <= 1) return 0;
    return gapOpen + gapExtend * (diffAAs - 1);
}

/* findoptimalsetbydp (collectoptimalset.cpp:106-217): cand is sorted, truncated to the unused ones, and marked; returns the
 * best path score and the chosen exons (in path order) in set[0..*nSet) */
static int optimal_set(exon_t *cand, size_t *nCand, exon_t *set, size_t *nSet, const mko_exon_params *P) {
    *nSet = 0;
    size_t n = *nCand;
    if (n == 0) return 0;
    stable_sort_exons(cand, n);
    size_t firstUsed = n;
    for (size_t i = 0; i < n; i++) if (cand[i].isUsed) { firstUsed = i; break; }
    n = *nCand = firstUsed;
    if (n == 0) return 0;     /* (the reference reads potentialExonCandidates[0] here; with --max-exon-sets 1 the list is never empty) */
    const int targetLength = cand[0].targetLen;
    size_t *prev = (size_t *) malloc(n * sizeof(size_t)), *numExons = (size_t *) malloc(n * sizeof(size_t));
    int *score = (int *) malloc(n * sizeof(int)), *aaLen = (int *) malloc(n * sizeof(int));
    for (size_t i = 0; i < n; i++) { prev[i] = i; score[i] = (int) cand[i].bitScore; numExons[i] = 1; aaLen[i] = cand[i].aaLen; }
    int best = 0;
    size_t last = 0;
    for (size_t c = 0; c < n; c++) {
        for (size_t p = 0; p < c; p++) {
            size_t overlap = 0;
            if (!pair_compatible(&cand[p], &cand[c], P->min_intron, P->max_intron, P->max_aa_overlap, &overlap)) continue;
            const size_t ne = numExons[p] + 1;
            const int bonus = (int) log2((double) ne);
            const int s = score[p] + transition_penalty(&cand[p], &cand[c], P->gap_open, P->gap_extend) + (int) cand[c].bitScore + bonus;
            if (s > score[c]) { prev[c] = p; score[c] = s; numExons[c] = ne; aaLen[c] = aaLen[p] + cand[c].aaLen - (int) overlap; }
        }
        if ((double) aaLen[c] / (double) targetLength >= P->target_cov_thr && score[c] > best) { last = c; best = score[c]; }
    }
    if (best != 0) {
        size_t k = last, m = 0;
        while (prev[k] != k) { set[m++] = cand[k]; cand[k].isUsed = 1; k = prev[k]; }
        set[m++] = cand[k]; cand[k].isUsed = 1;
        for (size_t i = 0; i < m / 2; i++) { exon_t t = set[i]; set[i] = set[m - 1 - i]; set[m - 1 - i] = t; }
        *nSet = m;
    }
    free(prev); free(numExons); free(score); free(aaLen);
    return best;
}

/* PotentialExon::exonToBuffer (:88-137) */
static size_t exon_to_buffer(char *b, const exon_t *e) {
    char *p = b;
    p += sprintf(p, "%u\t%d\t", e->exonKey, (int) e->bitScore);
    const float f = (float) e->seqId;
    if (f == 1.0) p += sprintf(p, "1.000\t");
    else {
        *p++ = '0'; *p++ = '.';
        if (f < 0.10) *p++ = '0';
        if (f < 0.01) *p++ = '0';
        const int s = (int) (f * 1000);
        p += sprintf(p, "%d\t", s);
    }
    p += sprintf(p, "%.3E\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", e->evalue, e->targetMatchStart, e->targetMatchEnd, e->targetLen,
                 e->contigStart, e->contigEnd, e->nucleotideLen, e->startBeforeTrim, e->endBeforeTrim);
    return (size_t) (p - b);
}

typedef struct { char *s; size_t n, cap; } sbuf;
static void sb_add(sbuf *o, const char *t, size_t n) {
    if (o->n + n + 1 > o->cap) { o->cap = 2 * (o->n + n) + 256; o->s = (char *) realloc(o->s, o->cap); }
    memcpy(o->s + o->n, t, n); o->n += n; o->s[o->n] = 0;
}

static void write_prediction(sbuf *out, unsigned int targetKey, int strand, int totalBitScore, double combinedEvalue, const exon_t *set, size_t n) {
    /* Prediction::Prediction (:193-215) and predictionToBuffer (:357-384) */
    const unsigned int low = (unsigned int) (set[0].strand == 1 ? set[0].contigStart : -1 * set[n - 1].contigEnd);
    const unsigned int high = (unsigned int) (set[0].strand == 1 ? set[n - 1].contigEnd : -1 * set[0].contigStart);
    char line[2048];
    for (size_t i = 0; i < n; i++) {
        char *p = line;
        p += sprintf(p, "%u\t%d\t%u\t%.3E\t%u\t%u\t%u\t", targetKey, strand, (unsigned int) totalBitScore, combinedEvalue, (unsigned int) n, low, high);
        p += exon_to_buffer(p, &set[i]);
        sb_add(out, line, (size_t) (p - line));
    }
}

static int aln_order(const void *x, const void *y) {                          /* resultspercontig's compareByTarget: (target key, orf key) */
    const mko_exon_aln *a = (const mko_exon_aln *) x, *b = (const mko_exon_aln *) y;
    if (a->target != b->target) return a->target < b->target ? -1 : 1;
    if (a->orf != b->orf) return a->orf < b->orf ? -1 : 1;
    return 0;
}

/* one contig: its ORF->target alignments (any order) with their ORF header coordinates; returns the prediction text
 * (malloc'd, one line per exon) */
char *mko_predict_exons(mko_exon_aln *alns, size_t n, const mko_exon_params *P, size_t *n_predictions) {
    sbuf out = {NULL, 0, 0};
    sb_add(&out, "", 0);
    *n_predictions = 0;
    qsort(alns, n, sizeof(mko_exon_aln), aln_order);                          /* (target, orf) pairs are unique: order is total */
    exon_t *plus = (exon_t *) malloc((n + 1) * sizeof(exon_t)), *minus = (exon_t *) malloc((n + 1) * sizeof(exon_t));
    exon_t *setP = (exon_t *) malloc((n + 1) * sizeof(exon_t)), *setM = (exon_t *) malloc((n + 1) * sizeof(exon_t));
    size_t i = 0;
    while (i < n) {
        const unsigned int target = alns[i].target;
        size_t np = 0, nm = 0;
        for (; i < n && alns[i].target == target; i++) {
            exon_t e;
            exon_set(&e, &alns[i], alns[i].orf, alns[i].orf_from, alns[i].orf_to);
            if ((size_t) (abs(e.nucleotideLen) / 3) >= P->min_exon_aa) { if (e.strand == 1) plus[np++] = e; else minus[nm++] = e; }
        }
        size_t iter = 0;
        while (iter < P->max_exon_sets && (np > 0 || nm > 0)) {
            size_t sp = 0, sm = 0;
            const int scoreP = optimal_set(plus, &np, setP, &sp, P), scoreM = optimal_set(minus, &nm, setM, &sm, P);
            if (sp > 0) {
                const double ev = pow(2, log2((double) P->db_residues) + log2(2) - scoreP);
                if (ev <= P->evalue_thr) { write_prediction(&out, target, 1, scoreP, ev, setP, sp); (*n_predictions)++; }
            }
            if (sm > 0) {
                const double ev = pow(2, log2((double) P->db_residues) + log2(2) - scoreM);
                if (ev <= P->evalue_thr) { write_prediction(&out, target, -1, scoreM, ev, setM, sm); (*n_predictions)++; }
            }
            iter++;
        }
    }
    free(plus); free(minus); free(setP); free(setM);
    return out.s;
}

void mko_exon_params_default(mko_exon_params *P, uint64_t db_residues) {       /* LocalParameters.h:138-146 */
    P->evalue_thr = (double) 0.001f; P->target_cov_thr = (double) 0.5f;
    P->max_intron = 10000; P->min_intron = 15; P->min_exon_aa = 11; P->max_aa_overlap = 10; P->max_exon_sets = 1;
    P->gap_open = -1; P->gap_extend = -1; P->db_residues = db_residues;
}
