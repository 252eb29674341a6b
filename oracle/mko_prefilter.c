/* oracle/mko_prefilter.c -- TEST INFRASTRUCTURE (parity oracle).  See mko.h.
 * One query through QueryMatcher::matchQuery (M/src/prefiltering/QueryMatcher.cpp:85-211). */
#include "mko.h"
#include <stdlib.h>
#include <string.h>

static const int SPACED6[6] = {0, 1, 3, 5, 8, 9};

/* QueryMatcher::initDiagonalMatcher (QueryMatcher.cpp:422-450): BINSIZE from the host L2 size */
int mko_bin_count_for(uint64_t db_size, uint64_t l2) {
    for (int x = 2; x <= 1024; x *= 2) if (db_size / (uint64_t) x < l2) return x;
    return 2048;
}

/* UngappedAlignment::createProfile (UngappedAlignment.cpp:385-414): profile[pos][aa] (row of 21) */
void mko_ungapped_profile(const mko_submat *um, const uint8_t *q, int L, const float *bias, int8_t *profile) {
    for (int pos = 0; pos < L; pos++) {
        float aaCorrBias = bias[pos];
        /* float/4 is float; -+0.5 promotes to double; result stored back into a float, then char */
        aaCorrBias = (float) ((aaCorrBias < 0.0) ? (double) (aaCorrBias / 4) - 0.5 : (double) (aaCorrBias / 4) + 0.5);
        signed char corr = (signed char) aaCorrBias;
        for (int a = 0; a < MKO_ALPH; a++) profile[pos * 21 + a] = (int8_t) (um->sub[q[pos]][a] + corr);
    }
}

/* UngappedAlignment::scalarDiagonalScoring (:30-43) */
static int scalar_diag(const int8_t *profile, unsigned int len, const uint8_t *db) {
    int max = 0, score = 0;
    for (unsigned int pos = 0; pos < len; pos++) {
        int curr = profile[pos * 21 + db[pos]];
        score = curr + score;
        score = (score < 0) ? 0 : score;
        max = (score > max) ? score : max;
    }
    return max;
}

/* UngappedAlignment::computeSingelSequenceScores (:416-431) */
static int single_scores(const int8_t *profile, unsigned int qlen, const uint8_t *t, unsigned int tlen, int diagonal, unsigned int minDist) {
    int max = 0;
    if (diagonal >= 0 && minDist < qlen) {
        unsigned int n = tlen < (qlen - minDist) ? tlen : (qlen - minDist);
        int s = scalar_diag(profile + (size_t) minDist * 21, n, t);
        max = s > max ? s : max;
    } else if (diagonal < 0 && minDist < tlen) {
        unsigned int n = (tlen - minDist) < qlen ? (tlen - minDist) : qlen;
        int s = scalar_diag(profile, n, t + minDist);
        max = s > max ? s : max;
    }
    return max;
}

/* UngappedAlignment::scoreSingleSequence (:441-451) incl. computeLongScore (:312-329); returns the
 * exact (unclamped) score of `diagonal` (u16 as stored in CounterResult). */
int mko_ungapped_score(const int8_t *profile, int qlen, const uint8_t *t, int tlen, uint16_t diagonal) {
    if (qlen >= 32768 || tlen >= 32768) {
        int totalMax = 0;
        for (unsigned int d = 1; d <= 1 + (unsigned int) tlen / 32768; d++) {
            int realDiagonal = (int) (-d * 65536 + diagonal);   /* unsigned wrap then int, as in the reference */
            int minDist = abs(realDiagonal);
            int m = single_scores(profile, qlen, t, tlen, realDiagonal, (unsigned int) minDist);
            totalMax = totalMax > m ? totalMax : m;
        }
        for (unsigned int d = 0; d <= (unsigned int) qlen / 65536; d++) {
            int realDiagonal = (int) (d * 65536 + diagonal);
            int minDist = abs(realDiagonal);
            int m = single_scores(profile, qlen, t, tlen, realDiagonal, (unsigned int) minDist);
            totalMax = totalMax > m ? totalMax : m;
        }
        return totalMax;
    }
    /* distanceFromDiagonal (:364-369) */
    unsigned short d1 = (unsigned short) (0 - diagonal), d2 = diagonal;
    unsigned short minDist = d1 < d2 ? d1 : d2;
    return single_scores(profile, qlen, t, tlen, (int) (short) diagonal, minDist);
}

typedef struct { uint32_t id; uint16_t diagonal; uint8_t count; } cres_t;   /* CounterResult */

/* stable partition by id & (B-1): CacheFriendlyOperations::hashIndexEntry / hashElements */
static void bin_partition(const cres_t *in, size_t n, int B, cres_t *out, size_t *bin_start /* B+1 */) {
    size_t *cnt = (size_t *) calloc((size_t) B + 1, sizeof(size_t));
    for (size_t i = 0; i < n; i++) cnt[(in[i].id & (uint32_t) (B - 1)) + 1]++;
    for (int b = 0; b < B; b++) cnt[b + 1] += cnt[b];
    memcpy(bin_start, cnt, ((size_t) B + 1) * sizeof(size_t));
    for (size_t i = 0; i < n; i++) out[cnt[in[i].id & (uint32_t) (B - 1)]++] = in[i];
    free(cnt);
}

static int hit_cmp(const void *a, const void *b) {   /* hit_t::compareHitsByScoreAndId (QueryMatcher.h:38-48) */
    const mko_hit *x = (const mko_hit *) a, *y = (const mko_hit *) b;
    int ax = abs(x->score), ay = abs(y->score);
    if (ax != ay) return ax > ay ? -1 : 1;
    if (x->seq_id != y->seq_id) return x->seq_id < y->seq_id ? -1 : 1;
    return 0;
}

/* The k-mer list of one k-mer start: the sequence path multiplies two 3-mer rows (mko_kmer_list6), the profile path six
 * position rows (mko_profile_kmer_list).  Returns the list length, or (size_t) -1 for a start that is skipped (an X in the window). */
typedef size_t (*kmer_gen_fn)(void *user, int pos, uint64_t **klist, size_t *kcap, uint64_t *listLen);

static int prefilter_core(const mko_prefilter_ctx *ctx, const uint8_t *q, int L, const int8_t *profile, kmer_gen_fn gen, void *user,
                          mko_hit *out, mko_prefilter_stats *st);

typedef struct { const mko_prefilter_ctx *ctx; const uint8_t *q; const float *bias; } seq_gen_t;

static size_t seq_kmer_gen(void *user, int i, uint64_t **klist, size_t *kcap, uint64_t *listLen) {
    const seq_gen_t *g = (const seq_gen_t *) user;
    const int k = g->ctx->index->k;
    const int *sp; mko_spaced_pattern(k, &sp);
    uint8_t kmer[8];
    int hasX = 0;
    float biasCorrection = 0;
    for (int p = 0; p < k; p++) {
        kmer[p] = g->q[i + sp[p]];
        hasX |= (kmer[p] == MKO_X);
        biasCorrection += g->bias[i + sp[p]];
    }
    if (hasX) return (size_t) -1;
    short b = (short) ((biasCorrection < 0.0) ? (double) biasCorrection - 0.5 : (double) biasCorrection + 0.5);
    int kms = g->ctx->kmer_thr - b;
    short kmerMatchScore = (short) (kms > 0 ? kms : 0);
    size_t nk = k == 7 ? mko_kmer_list7(g->ctx->two, g->ctx->three, kmer, kmerMatchScore, *klist, *kcap)
                       : mko_kmer_list6(g->ctx->three, kmer, kmerMatchScore, *klist, *kcap);
    if (nk > *kcap) {
        *kcap = nk;
        *klist = (uint64_t *) realloc(*klist, *kcap * sizeof(uint64_t));
        nk = k == 7 ? mko_kmer_list7(g->ctx->two, g->ctx->three, kmer, kmerMatchScore, *klist, *kcap)
                    : mko_kmer_list6(g->ctx->three, kmer, kmerMatchScore, *klist, *kcap);
    }
    *listLen += nk;
    return nk;
}

int mko_prefilter_query(const mko_prefilter_ctx *ctx, const uint8_t *q, int L, mko_hit *out, mko_prefilter_stats *st) {
    float *bias = (float *) malloc((size_t) (L > 0 ? L : 1) * sizeof(float));
    int8_t *profile = (int8_t *) malloc((size_t) (L > 0 ? L : 1) * 21);
    mko_comp_bias(ctx->kmer_mat, q, L, ctx->bias_scale, bias);               /* QueryMatcher.cpp:91-99 */
    mko_ungapped_profile(ctx->ungapped_mat, q, L, bias, profile);            /* :100-102 */
    seq_gen_t g = {ctx, q, bias};
    int rc = prefilter_core(ctx, q, L, profile, seq_kmer_gen, &g, out, st);
    free(bias); free(profile);
    return rc;
}

/* Profile query (QueryMatcher.cpp:91-99: no composition bias; UngappedAlignment::createProfile :385-408: the diagonal scores come
 * from profile_for_alignment, the X column stays 0; Sequence::kmerContainsX looks at the profile's query letters). */
typedef struct { const mko_prefilter_ctx *ctx; const mko_profile *p; } prof_gen_t;

static size_t prof_kmer_gen(void *user, int i, uint64_t **klist, size_t *kcap, uint64_t *listLen) {
    const prof_gen_t *g = (const prof_gen_t *) user;
    for (int p = 0; p < 6; p++) if (g->p->query[i + SPACED6[p]] == MKO_X) return (size_t) -1;
    int kms = g->ctx->kmer_thr;                                              /* bias 0: max(kmerThr - 0, 0), :242-243 */
    short kmerMatchScore = (short) (kms > 0 ? kms : 0);
    size_t nk = mko_profile_kmer_list(g->p, i, kmerMatchScore, *klist, *kcap);
    if (nk > *kcap) {
        *kcap = nk;
        *klist = (uint64_t *) realloc(*klist, *kcap * sizeof(uint64_t));
        nk = mko_profile_kmer_list(g->p, i, kmerMatchScore, *klist, *kcap);
    }
    *listLen += nk;
    return nk;
}

int mko_prefilter_profile(const mko_prefilter_ctx *ctx, const mko_profile *p, mko_hit *out, mko_prefilter_stats *st) {
    const int L = p->L;
    int8_t *profile = (int8_t *) calloc((size_t) (L > 0 ? L : 1) * 21, 1);
    for (int pos = 0; pos < L; pos++)
        for (int a = 0; a < 20; a++) profile[pos * 21 + a] = p->aln[(size_t) a * L + pos];
    prof_gen_t g = {ctx, p};
    int rc = prefilter_core(ctx, p->query, L, profile, prof_kmer_gen, &g, out, st);
    free(profile);
    return rc;
}

/* ---- A model of the DEVICE algorithm for the overflow path (MKO_OVERFLOW_MODEL=1): what the GPU kernels will compute, written
 * the way they will compute it, so that the plan is checked against the literal restatement above before any kernel exists.
 *   1. all hits of the query in arrival order (no buffer), with the arrival index of every k-mer list's first hit;
 *   2. segment boundaries from the list sizes (the buffer arithmetic of :281-316), `stopped` when one list alone fills the buffer;
 *   3. hits sorted by target (stable); the double-diagonal rule per (target, segment) run -- kept: low diagonal byte equals the previous
 *      hit's of the run (0 for the first); emitted: kept and the nearest earlier kept hit of the run has another byte;
 *   4. per target the merges of the overflow events replayed on its emitted candidates (scores from the exact ungapped score). */
typedef struct { uint32_t id; uint16_t diagonal; uint32_t arrival; } mhit_t;
typedef struct { uint64_t k; size_t i; } kp2_t;
static int kp2_cmp(const void *pa, const void *pb) { const kp2_t *a = (const kp2_t *) pa, *b = (const kp2_t *) pb; return a->k < b->k ? -1 : a->k > b->k; }
static int mhit_cmp(const void *a, const void *b) {
    const mhit_t *x = (const mhit_t *) a, *y = (const mhit_t *) b;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    if (x->arrival != y->arrival) return x->arrival < y->arrival ? -1 : 1;
    return 0;
}
static size_t overflow_device_model(const mko_index *ix, const int8_t *profile, int L, const mhit_t *hitsIn, size_t nAll,
                                    const uint32_t *listStart, size_t nLists, size_t maxDbMatches, cres_t **foundOut) {
    /* 2. segments */
    size_t segCap = nAll / (maxDbMatches / 2 + 1) + 8, nSeg = 1;
    uint32_t *segStart = (uint32_t *) malloc(segCap * sizeof(uint32_t));
    segStart[0] = 0;
    size_t n = 0, stopAt = nAll;
    int stopped = 0;
    for (size_t l = 0; l < nLists; l++) {
        const size_t sz = (l + 1 < nLists ? listStart[l + 1] : nAll) - listStart[l];
        if (n + sz >= maxDbMatches) {
            segStart[nSeg++] = listStart[l];
            n = 0;
            if (sz >= maxDbMatches) { stopped = 1; stopAt = listStart[l]; break; }
        }
        n += sz;
    }
    const size_t events = nSeg - 1;                      /* overflow events; segment `events` is the last one */
    cres_t *found = (cres_t *) malloc((nAll + 1) * sizeof(cres_t));
    size_t nf = 0;
    if (!stopped) {
        /* 3. sort by (target, arrival), rule per (target, segment) */
        mhit_t *h = (mhit_t *) malloc((stopAt + 1) * sizeof(mhit_t));
        memcpy(h, hitsIn, stopAt * sizeof(mhit_t));
        qsort(h, stopAt, sizeof(mhit_t), mhit_cmp);
        uint8_t *segOf = (uint8_t *) malloc(stopAt + 1);   /* (fewer than 256 segments in the cases modelled here) */
        uint8_t *emit = (uint8_t *) calloc(stopAt + 1, 1);
        for (size_t t = 0; t < stopAt; t++) { size_t s = 0; while (s + 1 < nSeg && segStart[s + 1] <= h[t].arrival) s++; segOf[t] = (uint8_t) s; }
        for (size_t t = 0; t < stopAt; t++) {
            const int samePrev = t > 0 && h[t - 1].id == h[t].id && segOf[t - 1] == segOf[t];
            const uint8_t lo = (uint8_t) h[t].diagonal, prevLo = samePrev ? (uint8_t) h[t - 1].diagonal : 0;
            if (lo != prevLo) continue;
            int e = 1;
            if (samePrev) {
                size_t u = t - 1;
                for (;;) {
                    const uint8_t ulo = (uint8_t) h[u].diagonal;
                    const int uSame = u > 0 && h[u - 1].id == h[u].id && segOf[u - 1] == segOf[u];
                    const uint8_t uprev = uSame ? (uint8_t) h[u - 1].diagonal : 0;
                    if (ulo == uprev) { e = ulo != lo; break; }
                    if (!uSame) break;
                    u--;
                }
            }
            emit[t] = (uint8_t) e;
        }
        /* 4. per target: replay.  Every element remembers the segment and arrival number it came from: the reference's array order -- which
         * decides ties at the --max-seqs cut -- follows from them.  An overflow event e >= 1 reverses the whole array: after it the order is
         * [C_e descending] ++ reverse(order before), so with m = events - 1:  order_m = [C_m desc] ++ reverse(order_(m-1)),  order_0 = [C_0 asc];
         * the last segment's candidates follow in arrival order. */
        typedef struct { cres_t c; uint32_t seg, arrival; } mel_t;
        mel_t *A = (mel_t *) malloc((stopAt + 1) * sizeof(mel_t)), *F = (mel_t *) malloc((stopAt + 1) * sizeof(mel_t));
        mel_t *surv = (mel_t *) malloc((stopAt + 1) * sizeof(mel_t));
        size_t ns = 0;
        for (size_t r0 = 0; r0 < stopAt; ) {
            size_t r1 = r0;
            while (r1 < stopAt && h[r1].id == h[r0].id) r1++;
            const uint32_t id = h[r0].id;
            size_t nF = 0;
            size_t t = r0;
            for (size_t e = 0; e <= events; e++) {
                size_t nA = 0;
                for (size_t x = 0; x < nF; x++) A[nA++] = F[x];
                for (; t < r1 && segOf[t] == e; t++)
                    if (emit[t]) { A[nA].c.id = id; A[nA].c.diagonal = h[t].diagonal; A[nA].c.count = 0; A[nA].seg = (uint32_t) e; A[nA].arrival = h[t].arrival; nA++; }
                if (e == events) {
                    nF = 0;
                    if (events == 0) { for (size_t x = 0; x < nA; x++) F[nF++] = A[x]; }
                    else if (nA > 0) {       /* mergeDiagonalDuplicates */
                        uint8_t d = (uint8_t) ((uint8_t) A[0].c.diagonal + 1);
                        for (size_t x = 0; x < nA; x++) { if (d != (uint8_t) A[x].c.diagonal) F[nF++] = A[x]; d = (uint8_t) A[x].c.diagonal; }
                    }
                } else if (e == 0) {
                    nF = 0;
                    for (size_t x = 0; x < nA; x++) F[nF++] = A[x];
                } else if (nA > 0) {         /* keep-scored merge (backwards, output reversed), score, per-target maximum */
                    uint8_t d = (uint8_t) ((uint8_t) A[nA - 1].c.diagonal + 1);
                    nF = 0;
                    for (size_t x = nA; x-- > 0;) { if (A[x].c.count != 0 || d != (uint8_t) A[x].c.diagonal) F[nF++] = A[x]; d = (uint8_t) A[x].c.diagonal; }
                    uint8_t mx = 0;
                    for (size_t x = 0; x < nF; x++) {
                        int sc = mko_ungapped_score(profile, L, ix->masked + ix->seq_off[id], (int) (ix->seq_off[id + 1] - ix->seq_off[id]), F[x].c.diagonal);
                        F[x].c.count = (uint8_t) (sc < 255 ? sc : 255);
                        if (F[x].c.count > mx) mx = F[x].c.count;
                    }
                    size_t w = 0;
                    for (size_t x = 0; x < nF; x++) { const int fnd = mx == F[x].c.count; if (fnd) { F[w++] = F[x]; mx = 0; } }
                    nF = w;
                } else nF = 0;
            }
            for (size_t x = 0; x < nF; x++) surv[ns++] = F[x];
            r0 = r1;
        }
        /* the array order: rank and direction of every segment */
        {
            int32_t *rank = (int32_t *) malloc((events + 2) * sizeof(int32_t));
            uint8_t *desc = (uint8_t *) calloc(events + 2, 1);
            /* order_e as a list of (segment, descending) */
            uint32_t *os = (uint32_t *) malloc((events + 2) * sizeof(uint32_t)), *ot = (uint32_t *) malloc((events + 2) * sizeof(uint32_t));
            uint8_t *od = (uint8_t *) malloc(events + 2), *odt = (uint8_t *) malloc(events + 2);
            size_t no = 1;
            os[0] = 0; od[0] = 0;
            for (size_t e = 1; e + 1 <= events; e++) {
                ot[0] = (uint32_t) e; odt[0] = 1;
                for (size_t x = 0; x < no; x++) { ot[1 + x] = os[no - 1 - x]; odt[1 + x] = (uint8_t) !od[no - 1 - x]; }
                no++;
                memcpy(os, ot, no * sizeof(uint32_t)); memcpy(od, odt, no);
            }
            if (events > 0) { os[no] = (uint32_t) events; od[no] = 0; no++; }
            for (size_t x = 0; x < no; x++) { rank[os[x]] = (int32_t) x; desc[os[x]] = od[x]; }
            /* counting sort of the survivors by (rank, +-arrival): simple insertion into per-rank buckets via qsort on a key */
            uint64_t *key = (uint64_t *) malloc((ns + 1) * sizeof(uint64_t));
            for (size_t x = 0; x < ns; x++) {
                const uint32_t a = surv[x].arrival;
                key[x] = ((uint64_t) rank[surv[x].seg] << 40) | ((uint64_t) (desc[surv[x].seg] ? 0xFFFFFFFFu - a : a) << 8);
            }
            /* (keys are unique: arrival numbers are) */
            kp2_t *kp = (kp2_t *) malloc((ns + 1) * sizeof(kp2_t));
            for (size_t x = 0; x < ns; x++) { kp[x].k = key[x]; kp[x].i = x; }
            qsort(kp, ns, sizeof(kp2_t), kp2_cmp);
            for (size_t x = 0; x < ns; x++) found[nf++] = surv[kp[x].i].c;
            free(rank); free(desc); free(os); free(ot); free(od); free(odt); free(key); free(kp);
        }
        free(surv);
        free(h); free(segOf); free(emit); free(A); free(F);
    }
    free(segStart);
    *foundOut = found;
    return nf;
}

static int prefilter_core(const mko_prefilter_ctx *ctx, const uint8_t *q, int L, const int8_t *profile, kmer_gen_fn gen, void *user,
                          mko_hit *out, mko_prefilter_stats *st) {
    const mko_index *ix = ctx->index;
    const int B = ctx->bin_count;
    int shift = 0;
    while ((1 << shift) < B) shift++;
    const size_t dbSize = ix->n_seq;
    const size_t maxDbMatches = (dbSize > 1000000 ? dbSize : 1000000) * 2;
    const size_t foundDiagonalsSize = dbSize > 1000000 ? dbSize : 1000000;
    int maxHits = ctx->max_hits < (int) dbSize ? ctx->max_hits : (int) dbSize;

    /* ---- match() (:213-346): gather index entries in (i, k-mer list, index list) order ---- */
    size_t cap = 1 << 16, n = 0;
    cres_t *hits = (cres_t *) malloc(cap * sizeof(cres_t));
    size_t kcap = 1 << 20;
    uint64_t *klist = (uint64_t *) malloc(kcap * sizeof(uint64_t));
    uint64_t kmerListLen = 0;
    int rc = 0;
    const int *spUnused; const int span = mko_spaced_pattern(ix->k, &spUnused);
    /* the diagonal matcher's working state (CacheFriendlyOperations): bins, duplicateBitArray, tmpElementBuffer */
    uint8_t *dup = (uint8_t *) calloc((dbSize >> shift) + 2, 1);
    size_t *bs = (size_t *) malloc(((size_t) B + 1) * sizeof(size_t));
    size_t wcap = 1 << 16, tcap = 1 << 16;
    cres_t *binned = (cres_t *) malloc(wcap * sizeof(cres_t)), *tmp = (cres_t *) malloc(tcap * sizeof(cres_t));
    /* foundDiagonals: [0, overflowHitCount) = what earlier overflow segments left, behind it the current segment's diagonals */
    size_t fcap = 1 << 16, overflowHitCount = 0, hitCount = 0;
    cres_t *found = (cres_t *) malloc(fcap * sizeof(cres_t));
    uint64_t dbMatchesAll = 0;
    int stopped = 0;
#define GROW(ptr, capv, need) do { if ((need) > (capv)) { while ((need) > (capv)) (capv) *= 2; (ptr) = (cres_t *) realloc((ptr), (capv) * sizeof(cres_t)); } } while (0)
    /* findDuplicates of the gathered segment hits[0..n) into found[overflowHitCount..] (CacheFriendlyOperations.cpp:38-49,185-274, computeTotalScore == false) */
#define FIND_DUPLICATES(result) do { \
        GROW(binned, wcap, n + 1); GROW(tmp, tcap, n + 1); GROW(found, fcap, overflowHitCount + n + 1); \
        bin_partition(hits, n, B, binned, bs); \
        cres_t *cand_ = found + overflowHitCount; \
        const size_t outSize_ = foundDiagonalsSize - overflowHitCount; \
        size_t nc_ = 0; \
        for (int bin = 0; bin < B; bin++) { \
            const cres_t *bp = binned + bs[bin]; \
            const size_t cur = bs[bin + 1] - bs[bin]; \
            size_t ec = 0; \
            for (size_t k = 0; k < cur; k++) { \
                const uint32_t h = bp[k].id >> shift; \
                const uint8_t currDiagonal = (uint8_t) bp[k].diagonal; \
                const uint8_t prevDiagonal = dup[h]; \
                tmp[ec] = bp[k]; \
                ec += (currDiagonal == prevDiagonal) ? 1 : 0; \
                dup[h] = currDiagonal; \
            } \
            if (nc_ + (ec < cur / 2 ? ec : cur / 2) >= outSize_) break;   /* :214-216 */ \
            for (size_t k = ec; k-- > 0;) dup[tmp[k].id >> shift] = (uint8_t) ((uint8_t) tmp[k].diagonal + 1); \
            for (size_t k = 0; k < ec; k++) { \
                const uint32_t h = tmp[k].id >> shift; \
                cand_[nc_].id = tmp[k].id; \
                cand_[nc_].count = 0; \
                cand_[nc_].diagonal = tmp[k].diagonal; \
                nc_ += (dup[h] != (uint8_t) tmp[k].diagonal) ? 1 : 0; \
                dup[h] = (uint8_t) tmp[k].diagonal; \
            } \
            for (size_t k = 0; k < cur; k++) dup[bp[k].id >> shift] = 0; \
        } \
        (result) = nc_; \
    } while (0)
    if (getenv("MKO_OVERFLOW_MODEL")) {
        /* the device model instead of the literal buffer logic below (tests compare the two) */
        size_t mcap = 1 << 20, nAll = 0, lcap = 1 << 16, nLists = 0;
        mhit_t *mh = (mhit_t *) malloc(mcap * sizeof(mhit_t));
        uint32_t *ls = (uint32_t *) malloc(lcap * sizeof(uint32_t));
        for (int i = 0; i + span <= L; i++) {
            const size_t nk = gen(user, i, &klist, &kcap, &kmerListLen);
            if (nk == (size_t) -1) continue;
            for (size_t k = 0; k < nk; k++) {
                uint64_t o0, o1;
                mko_index_list(ix, klist[k], &o0, &o1);
                const size_t sz = (size_t) (o1 - o0);
                if (sz == 0) continue;
                if (nLists == lcap) { lcap *= 2; ls = (uint32_t *) realloc(ls, lcap * sizeof(uint32_t)); }
                ls[nLists++] = (uint32_t) nAll;
                if (nAll + sz > mcap) { while (nAll + sz > mcap) mcap *= 2; mh = (mhit_t *) realloc(mh, mcap * sizeof(mhit_t)); }
                for (size_t e = 0; e < sz; e++) {
                    mh[nAll].id = ix->seq_id[o0 + e];
                    mh[nAll].diagonal = (uint16_t) (i - (int) ix->pos[o0 + e]);
                    mh[nAll].arrival = (uint32_t) nAll;
                    nAll++;
                }
            }
        }
        cres_t *mf = NULL;
        hitCount = overflow_device_model(ix, profile, L, mh, nAll, ls, nLists, maxDbMatches, &mf);
        GROW(found, fcap, hitCount + 1);
        memcpy(found, mf, hitCount * sizeof(cres_t));
        free(mf); free(mh); free(ls);
        dbMatchesAll = nAll;
        n = 0;
        goto tail;
    }
    for (int i = 0; i + span <= L && !stopped; i++) {
        const size_t nk = gen(user, i, &klist, &kcap, &kmerListLen);
        if (nk == (size_t) -1) continue;
        for (size_t k = 0; k < nk; k++) {
            uint64_t o0, o1;
            mko_index_list(ix, klist[k], &o0, &o1);
            const size_t sz = (size_t) (o1 - o0);
            if (n + sz >= maxDbMatches) {
                /* the overflow path (:281-316): the hits gathered so far are one segment with a double-diagonal rule of its own; from the
                 * second overflow on the kept diagonals are merged (the later one of equal neighbours stays, order reversed), scored, and
                 * only the best one per target survives */
                size_t hc;
                FIND_DUPLICATES(hc);
                if (overflowHitCount != 0) {
                    size_t N = hc + overflowHitCount;
                    /* mergeDiagonalKeepScoredHitsDuplicates (CacheFriendlyOperations.cpp:118-150) */
                    GROW(binned, wcap, N + 1);
                    bin_partition(found, N, B, binned, bs);
                    size_t m = 0;
                    for (int bin = 0; bin < B; bin++) {
                        const cres_t *bp = binned + bs[bin];
                        const size_t cur = bs[bin + 1] - bs[bin];
                        for (size_t x = 0; x < cur; x++) dup[bp[x].id >> shift] = (uint8_t) ((uint8_t) bp[x].diagonal + 1);
                        for (size_t x = cur; x-- > 0;) {
                            const uint32_t h = bp[x].id >> shift;
                            found[m] = bp[x];
                            m += (found[m].count != 0 || dup[h] != (uint8_t) bp[x].diagonal) ? 1 : 0;
                            dup[h] = (uint8_t) bp[x].diagonal;
                        }
                    }
                    /* ungappedAlignment->align (:295) */
                    for (size_t x = 0; x < m; x++) {
                        const uint32_t id = found[x].id;
                        int sc = mko_ungapped_score(profile, L, ix->masked + ix->seq_off[id], (int) (ix->seq_off[id + 1] - ix->seq_off[id]), found[x].diagonal);
                        found[x].count = (uint8_t) (sc < 255 ? sc : 255);
                    }
                    /* keepMaxScoreElementOnly (:297; CacheFriendlyOperations.cpp:350-380) */
                    GROW(binned, wcap, m + 1);
                    bin_partition(found, m, B, binned, bs);
                    memset(dup, 0, (dbSize >> shift) + 2);
                    size_t r = 0;
                    for (int bin = 0; bin < B; bin++) {
                        const cres_t *bp = binned + bs[bin];
                        const size_t cur = bs[bin + 1] - bs[bin];
                        for (size_t x = 0; x < cur; x++) { const uint32_t h = bp[x].id >> shift; if (bp[x].count > dup[h]) dup[h] = bp[x].count; }
                        for (size_t x = 0; x < cur; x++) {
                            const uint32_t h = bp[x].id >> shift;
                            found[r] = bp[x];
                            int fnd = (dup[h] == bp[x].count) ? 1 : 0;
                            r += (size_t) fnd;
                            dup[h] = (uint8_t) (dup[h] * (1 - fnd));
                        }
                    }
                    overflowHitCount = r;
                } else {
                    overflowHitCount = hc;
                }
                dbMatchesAll += n;
                n = 0;
                if (n + sz >= maxDbMatches) { stopped = 1; break; }      /* :313-315: one list alone fills the buffer -> everything is dropped below */
            }
            if (n + sz > cap) { while (n + sz > cap) cap *= 2; hits = (cres_t *) realloc(hits, cap * sizeof(cres_t)); }
            for (size_t e = 0; e < sz; e++) {
                hits[n].id = ix->seq_id[o0 + e];
                hits[n].diagonal = (uint16_t) (i - (int) ix->pos[o0 + e]);   /* hashIndexEntry :337-347 */
                hits[n].count = 0;
                n++;
            }
        }
    }
    /* :318-334: the last segment; nothing at all when it is empty */
    if (n > 0) {
        FIND_DUPLICATES(hitCount);
        if (overflowHitCount != 0) {
            /* mergeDiagonalDuplicates (CacheFriendlyOperations.cpp:80-115): of equal neighbours the earlier one stays */
            const size_t N = overflowHitCount + hitCount;
            GROW(binned, wcap, N + 1);
            bin_partition(found, N, B, binned, bs);
            size_t m = 0;
            for (int bin = 0; bin < B; bin++) {
                const cres_t *bp = binned + bs[bin];
                const size_t cur = bs[bin + 1] - bs[bin];
                for (size_t x = cur; x-- > 0;) dup[bp[x].id >> shift] = (uint8_t) ((uint8_t) bp[x].diagonal + 1);
                for (size_t x = 0; x < cur; x++) {
                    const uint32_t h = bp[x].id >> shift;
                    found[m] = bp[x];
                    m += (dup[h] != (uint8_t) bp[x].diagonal) ? 1 : 0;
                    dup[h] = (uint8_t) bp[x].diagonal;
                }
            }
            hitCount = m;
        }
    }
    dbMatchesAll += n;
tail:
    if (st) { st->kmer_list_len = kmerListLen; st->db_matches = dbMatchesAll; st->diagonals = 0; }
    {
        cres_t *cand = found;
        size_t nc = hitCount;
        GROW(binned, wcap, nc + 1);
        if (st) st->diagonals = nc;
        if (nc >= foundDiagonalsSize / 2) { rc = -1; free(binned); free(bs); free(dup); free(found); free(tmp); goto done; }

        /* ---- UngappedAlignment::align / computeScores (:331-362): count = min(255, score) ---- */
        for (size_t k = 0; k < nc; k++) {
            const uint32_t id = cand[k].id;
            int s = mko_ungapped_score(profile, L, ix->masked + ix->seq_off[id], (int) (ix->seq_off[id + 1] - ix->seq_off[id]), cand[k].diagonal);
            cand[k].count = (uint8_t) (s < 255 ? s : 255);
        }
        /* ---- keepMaxScoreElementOnly (CacheFriendlyOperations.cpp:70-78,350-380) ---- */
        bin_partition(cand, nc, B, binned, bs);
        memset(dup, 0, (dbSize >> shift) + 2);                    /* keepMaxElement starts from a cleared array (:353) */
        size_t nr = 0;
        for (int bin = 0; bin < B; bin++) {
            const cres_t *bp = binned + bs[bin];
            const size_t cur = bs[bin + 1] - bs[bin];
            for (size_t k = 0; k < cur; k++) {
                const uint32_t h = bp[k].id >> shift;
                if (bp[k].count > dup[h]) dup[h] = bp[k].count;
            }
            for (size_t k = 0; k < cur; k++) {
                const uint32_t h = bp[k].id >> shift;
                cand[nr] = bp[k];
                int found = (dup[h] == bp[k].count) ? 1 : 0;
                nr += (size_t) found;
                dup[h] = (uint8_t) (dup[h] * (1 - found));
            }
        }
        /* ---- score histogram, threshold (QueryMatcher.h:206-216, .cpp:152-155) ---- */
        unsigned int scoreSizes[256];
        memset(scoreSizes, 0, sizeof(scoreSizes));
        for (size_t k = 0; k < nr; k++) scoreSizes[cand[k].count]++;
        unsigned int thr;
        {
            size_t found = 0;
            size_t t;
            for (t = 255; t > 0; t--) { found += scoreSizes[t]; if (found >= (size_t) maxHits) break; }
            thr = (unsigned int) t;
        }
        unsigned int diagonalThr = thr > (unsigned int) ctx->min_diag_score ? thr : (unsigned int) ctx->min_diag_score;
        /* ---- radixSortByScoreSize (:498-523): descending score, stable ---- */
        cres_t *sorted = binned;   /* reuse */
        size_t above = 0;
        {
            size_t ptr[256];
            size_t prev = nr;
            for (int s = 0; s < 256; s++) { ptr[s] = prev - scoreSizes[s]; prev = ptr[s]; }
            for (size_t k = 0; k < nr; k++) {
                if (cand[k].count >= diagonalThr) { above++; sorted[ptr[cand[k].count]++] = cand[k]; }
            }
        }
        int nout = 0;
        if (diagonalThr >= 255) {
            /* saturated threshold: rescoreHits (:525-544) then radix sort again, getResult with rescale */
            memset(scoreSizes, 0, sizeof(scoreSizes));
            int maxSelfScore = mko_ungapped_score(profile, L, q, L, 0);
            maxSelfScore = maxSelfScore - 255;
            maxSelfScore = maxSelfScore > 1 ? maxSelfScore : 1;
            maxSelfScore = maxSelfScore < 65535 ? maxSelfScore : 65535;
            float fltMax = (float) maxSelfScore;
            size_t elements = 0;
            for (size_t k = 0; k < above && sorted[k].count >= 255; k++) {
                const uint32_t id = sorted[k].id;
                unsigned int newScore = (unsigned int) mko_ungapped_score(profile, L, ix->masked + ix->seq_off[id], (int) (ix->seq_off[id + 1] - ix->seq_off[id]), sorted[k].diagonal);
                newScore -= 255;
                float score = (float) (newScore < 65535u ? newScore : 65535u);
                sorted[k].count = (uint8_t) ((double) ((score / fltMax) * (float) 255) + 0.5);
                scoreSizes[sorted[k].count] += 1;
                elements++;
            }
            size_t ptr[256];
            size_t prev = elements;
            for (int s = 0; s < 256; s++) { ptr[s] = prev - scoreSizes[s]; prev = ptr[s]; }
            for (size_t k = 0; k < elements; k++) cand[ptr[sorted[k].count]++] = sorted[k];
            for (size_t k = 0; k < elements && nout < maxHits; k++) {
                out[nout].seq_id = cand[k].id;
                out[nout].diagonal = cand[k].diagonal;
                unsigned int newScore = 255u + ((unsigned int) cand[k].count * (unsigned int) maxSelfScore / 255u);
                out[nout].score = (int32_t) newScore;
                nout++;
            }
        } else {
            /* getResult<UNGAPPED_DIAGONAL_SCORE> (:363-420) */
            for (size_t k = 0; k < above && nout < maxHits; k++) {
                out[nout].seq_id = sorted[k].id;
                out[nout].diagonal = sorted[k].diagonal;
                out[nout].score = sorted[k].count;
                if (sorted[k].count >= 255) {
                    const uint32_t id = sorted[k].id;
                    out[nout].score = mko_ungapped_score(profile, L, ix->masked + ix->seq_off[id], (int) (ix->seq_off[id + 1] - ix->seq_off[id]), sorted[k].diagonal);
                }
                nout++;
            }
        }
        if (nout > 1) qsort(out, (size_t) nout, sizeof(mko_hit), hit_cmp);   /* :203-209 */
        rc = nout;
        free(binned); free(bs); free(dup); free(found); free(tmp);
    }
done:
    free(hits); free(klist);
    return rc;
}

/* QueryMatcher::prefilterHitToBuffer (QueryMatcher.h:118-130) */
#include <stdio.h>
size_t mko_format_hit(char *buf, const mko_hit *h) {
    return (size_t) sprintf(buf, "%u\t%d\t%d\n", h->seq_id, h->score, (int) (short) h->diagonal);
}
