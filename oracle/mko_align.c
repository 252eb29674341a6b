/* oracle/mko_align.c -- TEST INFRASTRUCTURE (parity oracle).  See mko.h.
 * Gapped Smith-Waterman with the reference's striped-SIMD semantics, ALP e-values,
 * Matcher result assembly and formatting. */
#include "mko.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* SmithWaterman::ssw_init (M/src/alignment/StripedSmithWaterman.cpp:1216-1345): int8 composition
 * bias (round half away from zero, :1230-1235) and the byte-mode bias |min(mat)|+|min(compBias)|
 * (:1275-1284).  mat = Matcher::setSubstitutionMatrix (Matcher.cpp:27-34): int8 of subMatrix. */
void mko_sw_query_init(const mko_submat *m, const uint8_t *q, int L, float bias_scale, int8_t *comp_bias8, int *bias) {
    float *tmp = (float *) malloc((size_t) (L > 0 ? L : 1) * sizeof(float));
    mko_comp_bias(m, q, L, bias_scale, tmp);
    int compositionBias = 0;
    for (int i = 0; i < L; i++) {
        comp_bias8[i] = (int8_t) ((tmp[i] < 0.0) ? tmp[i] - 0.5 : tmp[i] + 0.5);
        compositionBias = (compositionBias < comp_bias8[i]) ? compositionBias : comp_bias8[i];
    }
    compositionBias = compositionBias < 0 ? compositionBias : 0;
    int b = 0;
    for (int i = 0; i < MKO_ALPH; i++)
        for (int j = 0; j < MKO_ALPH; j++)
            if ((int8_t) m->sub[i][j] < b) b = (int8_t) m->sub[i][j];
    *bias = abs(b) + abs(compositionBias);
    free(tmp);
}

/* One striped pass, sw_sse2_byte (:638-940) / sw_sse2_word (:942-1214), restated per cell.
 *   qs,cb : query residues / int8 composition bias, already oriented for this pass
 *   t     : target residues; columns are visited t0, t0+step, ... for n_cols columns
 *   lanes : SIMD lanes of the reference build -> stripe length segLen = ceil(qlen/lanes)
 * Semantics (derived in DESIGN.md "SW recurrence"):
 *   H'(q) = max(0, Hprev(q-1)+s, E(q), Fm(q));  Fm restarts at 0 at every stripe head (vF=0, :715)
 *   Enext(q) = max(0, E(q)-ge, H'(q)-go)   -- uses H', i.e. no E-open out of a lazy-F cell (:819-871)
 *   F(q)  = max(0, F(q-1)-ge, H'(q-1)-go) across stripe heads (lazy-F loop);  H(q) = max(H'(q), F(q))
 *   column max / global max / end coordinates over H (:873-912).
 * word != 0: int16 saturating add (:1063).  Returns through out params; *overflow set when byte
 * mode would have reported 255 (:879-883). */
/* Profile queries (ssw_align_private<PROFILE_SEQ>, :296-298; createQueryProfile<.., PROFILE> :175-181): prof != NULL, the score of
 * row q against residue tc is prof[tc * pL + pq0 + q * pqstep] (mat / mat_rev of ssw_init: [aa][pos], X row 0) and there is no
 * composition bias.  GAP_POS_SCORING is not defined in this reference, so posSpecificGaps changes nothing. */
typedef struct { const int8_t *prof; int pL, pq0, pqstep; } sw_prof_t;

static void sw_pass(const mko_submat *m, const uint8_t *qs, const int8_t *cb, int qlen, const sw_prof_t *P,
                    const uint8_t *t, int t0, int step, int n_cols,
                    int go, int ge, int lanes, int word, int bias, int terminate,
                    int *o_max, int *o_end_t, int *o_end_q, int *overflow) {
    const int segLen = (qlen + lanes - 1) / lanes;
    int *Hprev = (int *) calloc((size_t) qlen + 1, sizeof(int));
    int *Hcur = (int *) calloc((size_t) qlen + 1, sizeof(int));
    int *E = (int *) calloc((size_t) qlen + 1, sizeof(int));
    int *Hmax = (int *) calloc((size_t) qlen + 1, sizeof(int));
    int max = 0, end_t = word ? 0 : -1;
    *overflow = 0;
    for (int c = 0; c < n_cols; c++) {
        const int ti = t0 + c * step;
        const uint8_t tc = t[ti];
        int Fm = 0, F = 0, colmax = 0;
        for (int q = 0; q < qlen; q++) {
            const int s = P ? (int) P->prof[(size_t) tc * P->pL + P->pq0 + q * P->pqstep]
                            : (int) (int8_t) m->sub[tc][qs[q]] + (int) cb[q];   /* createQueryProfile :162-187 */
            int diag = (q > 0 ? Hprev[q - 1] : 0) + s;
            if (word && diag > 32767) diag = 32767;
            if (q % segLen == 0) Fm = 0;
            int hp = diag > 0 ? diag : 0;
            if (E[q] > hp) hp = E[q];
            if (Fm > hp) hp = Fm;                  /* H' */
            int h = hp > F ? hp : F;               /* H after lazy-F */
            Hcur[q] = h;
            if (h > colmax) colmax = h;
            int e = E[q] - ge; if (e < 0) e = 0;
            int ho = hp - go; if (ho < 0) ho = 0;
            E[q] = e > ho ? e : ho;
            Fm = Fm - ge; if (Fm < 0) Fm = 0; if (ho > Fm) Fm = ho;
            F = F - ge; if (F < 0) F = 0; if (ho > F) F = ho;
        }
        if (colmax > max) {
            max = colmax;
            if (!word && max + bias >= 255) { *overflow = 1; break; }
            end_t = ti;
            memcpy(Hmax, Hcur, (size_t) qlen * sizeof(int));
        }
        int *tmp = Hprev; Hprev = Hcur; Hcur = tmp;
        if (colmax == terminate) break;
    }
    int end_q = qlen - 1;
    for (int q = 0; q < qlen; q++) if (Hmax[q] == max) { if (q < end_q) end_q = q; }
    *o_max = max; *o_end_t = end_t; *o_end_q = end_q;
    free(Hprev); free(Hcur); free(E); free(Hmax);
}

/* ssw_align_private<SEQ_SEQ> forward part (:346-377) */
void mko_sw_forward(const mko_submat *m, const uint8_t *q, const int8_t *cb, int bias, int qlen,
                    const uint8_t *t, int tlen, int go, int ge, int lanes_byte, int lanes_word, mko_sw_result *r) {
    int mx, et, eq, ovf;
    r->word = 0; r->rev_mismatch = 0; r->q_start = -1; r->t_start = -1;
    sw_pass(m, q, cb, qlen, NULL, t, 0, 1, tlen, go, ge, lanes_byte, 0, bias, 255 /* UCHAR_MAX */, &mx, &et, &eq, &ovf);
    if (ovf) {
        sw_pass(m, q, cb, qlen, NULL, t, 0, 1, tlen, go, ge, lanes_word, 1, 0, 65535, &mx, &et, &eq, &ovf);
        r->word = 1;
    }
    r->score = mx; r->t_end = et; r->q_end = eq;
}

/* reverse pass (:400-476): reversed query prefix [0..q_end], target columns t_end..0, stop at score1 */
void mko_sw_reverse(const mko_submat *m, const uint8_t *q, const int8_t *cb, int bias, int qlen,
                    const uint8_t *t, int tlen, int go, int ge, int lanes_byte, int lanes_word, mko_sw_result *r) {
    (void) tlen; (void) qlen;
    const int n = r->q_end + 1;
    uint8_t *rq = (uint8_t *) malloc((size_t) n);
    int8_t *rcb = (int8_t *) malloc((size_t) n);
    for (int k = 0; k < n; k++) { rq[k] = q[r->q_end - k]; rcb[k] = cb[r->q_end - k]; }
    int mx, et, eq, ovf;
    sw_pass(m, rq, rcb, n, NULL, t, r->t_end, -1, r->t_end + 1, go, ge, r->word ? lanes_word : lanes_byte, r->word,
            r->word ? 0 : bias, r->score, &mx, &et, &eq, &ovf);
    if (mx != r->score) r->rev_mismatch = 1;
    r->t_start = et;
    r->q_start = r->q_end - eq;
    free(rq); free(rcb);
}

/* the two passes for a profile query: forward over the whole profile, reverse over the reversed profile prefix [0..q_end]
 * (mat_rev + queryOffset, :401-403 / :429-433) */
void mko_sw_forward_profile(const mko_profile *p, int bias, const uint8_t *t, int tlen, int go, int ge, int lanes_byte, int lanes_word,
                            mko_sw_result *r) {
    int mx, et, eq, ovf;
    const sw_prof_t P = {p->aln, p->L, 0, 1};
    r->word = 0; r->rev_mismatch = 0; r->q_start = -1; r->t_start = -1;
    sw_pass(NULL, NULL, NULL, p->L, &P, t, 0, 1, tlen, go, ge, lanes_byte, 0, bias, 255, &mx, &et, &eq, &ovf);
    if (ovf) {
        sw_pass(NULL, NULL, NULL, p->L, &P, t, 0, 1, tlen, go, ge, lanes_word, 1, 0, 65535, &mx, &et, &eq, &ovf);
        r->word = 1;
    }
    r->score = mx; r->t_end = et; r->q_end = eq;
}

void mko_sw_reverse_profile(const mko_profile *p, int bias, const uint8_t *t, int go, int ge, int lanes_byte, int lanes_word, mko_sw_result *r) {
    int mx, et, eq, ovf;
    const sw_prof_t P = {p->aln, p->L, r->q_end, -1};
    sw_pass(NULL, NULL, NULL, r->q_end + 1, &P, t, r->t_end, -1, r->t_end + 1, go, ge, r->word ? lanes_word : lanes_byte, r->word,
            r->word ? 0 : bias, r->score, &mx, &et, &eq, &ovf);
    if (mx != r->score) r->rev_mismatch = 1;
    r->t_start = et;
    r->q_start = r->q_end - eq;
}

/* EvalueComputation (M/src/alignment/EvalueComputation.h:64-69 hard-coded BLOSUM62 11/1 Gumbel set)
 * + Sls::AlignmentEvaluer::initParameters (M/lib/alp/sls_alignment_evaluer.cpp:657-835)
 * + pvalues::compute_tmp_values (sls_pvalues.cpp:342-364) */
void mko_evaluer_init(mko_evaluer *e, uint64_t db_residues) {
    e->lambda = 0.27359865037097330642; e->K = 0.044620920658722244834;
    e->a_J = 1.5938724404943873658; e->b_J = -19.959867650284412122;
    e->a_I = 1.5938724404943873658; e->b_I = -19.959867650284412122;
    e->alpha_J = 30.455610143099914211; e->beta_J = -622.28684628915891608;
    e->alpha_I = 30.455610143099914211; e->beta_I = -622.28684628915891608;
    e->sigma = 29.602444874818868215; e->tau = -601.81087985041381216;
    const double nat_cut_off_in_max = 2.0;
    double v;
    v = nat_cut_off_in_max * e->alpha_I / e->lambda; e->vi_y_thr = v > 0.0 ? v : 0.0;
    v = nat_cut_off_in_max * e->alpha_J / e->lambda; e->vj_y_thr = v > 0.0 ? v : 0.0;
    v = nat_cut_off_in_max * e->sigma / e->lambda; e->c_y_thr = v > 0.0 ? v : 0.0;
    e->logK = log(e->K);
    e->db_res = (double) db_residues;
}

/* pvalues::get_appr_tail_prob_with_cov_without_errors (sls_pvalues.cpp:366-520), compute_only_area;
 * called as area(score, seqlen1 = qlen, seqlen2 = dbRes) -> (y, m = seqlen2, n = seqlen1)
 * (sls_alignment_evaluer.cpp:989-1029). */
static double alp_area(const mko_evaluer *e, double y, double qlen) {
    const double pi = 3.1415926535897932384626433832795;
    const double const_val = 1 / sqrt(2.0 * pi);
    const double m_ = e->db_res, n_ = qlen;
    double m_li_y = m_ - (e->a_I * y + e->b_I);
    double vi_y = e->alpha_I * y + e->beta_I; if (e->vi_y_thr > vi_y) vi_y = e->vi_y_thr;
    double sqrt_vi_y = sqrt(vi_y);
    double m_F = (sqrt_vi_y == 0.0) ? 1e100 : m_li_y / sqrt_vi_y;
    double P_m_F = 0.5 * erfc(-sqrt(0.5) * m_F);             /* sls_basic.hpp:195-198 */
    double E_m_F = -const_val * exp(-0.5 * m_F * m_F);
    double p1 = m_li_y * P_m_F - sqrt_vi_y * E_m_F;
    double n_lj_y = n_ - (e->a_J * y + e->b_J);
    double vj_y = e->alpha_J * y + e->beta_J; if (e->vj_y_thr > vj_y) vj_y = e->vj_y_thr;
    double sqrt_vj_y = sqrt(vj_y);
    double n_F = (sqrt_vj_y == 0.0) ? 1e100 : n_lj_y / sqrt_vj_y;
    double P_n_F = 0.5 * erfc(-sqrt(0.5) * n_F);
    double E_n_F = -const_val * exp(-0.5 * n_F * n_F);
    double p2 = n_lj_y * P_n_F - sqrt_vj_y * E_n_F;
    double c_y = e->sigma * y + e->tau; if (e->c_y_thr > c_y) c_y = e->c_y_thr;
    double P_m_F_P_n_F = P_m_F * P_n_F;
    double c_y_P = c_y * P_m_F_P_n_F;
    double p1_p2 = p1 * p2;
    return p1_p2 + c_y_P;
}

double mko_evalue(const mko_evaluer *e, double score, double qlen) {
    const double epa = e->K * exp(-e->lambda * score);       /* evaluePerArea, sls_alignment_evaluer.hpp:154-157 */
    const double a = alp_area(e, score, qlen);
    return epa * a;                                          /* EvalueComputation.h:33-37 */
}

double mko_bitscore(const mko_evaluer *e, double score) {
    return (e->lambda * score - e->logK) / log(2.0);         /* sls_alignment_evaluer.hpp:159-162 */
}

/* SmithWaterman::computeCov (:1671-1673), unsigned arithmetic */
static float compute_cov(unsigned int startPos, unsigned int endPos, unsigned int len) {
    unsigned int mx = startPos > endPos ? startPos : endPos;
    unsigned int mn = startPos < endPos ? startPos : endPos;
    unsigned int a = len < mx ? len : mx;
    return (a - mn + 1) / (float) len;
}

/* Matcher::getSWResult (Matcher.cpp:60-142) in SCORE_COV mode + Alignment::checkCriteria
 * (Alignment.cpp:548-567) with covThr 0, seqIdThr 0. */
static int align_pair(const mko_align_ctx *ctx, const uint8_t *q, const int8_t *cb, const mko_profile *prof, int bias, int qlen,
                      const uint8_t *t, int tlen, uint32_t db_key, mko_aln_result *out);

int mko_align_pair(const mko_align_ctx *ctx, const uint8_t *q, const int8_t *cb, int bias, int qlen,
                   const uint8_t *t, int tlen, uint32_t db_key, mko_aln_result *out) {
    return align_pair(ctx, q, cb, NULL, bias, qlen, t, tlen, db_key, out);
}

/* the same for a profile query (Matcher::initQuery, Matcher.cpp:53-54) */
int mko_align_pair_profile(const mko_align_ctx *ctx, const mko_profile *prof, int bias, const uint8_t *t, int tlen, uint32_t db_key,
                           mko_aln_result *out) {
    return align_pair(ctx, NULL, NULL, prof, bias, prof->L, t, tlen, db_key, out);
}

static int align_pair(const mko_align_ctx *ctx, const uint8_t *q, const int8_t *cb, const mko_profile *prof, int bias, int qlen,
                      const uint8_t *t, int tlen, uint32_t db_key, mko_aln_result *out) {
    mko_sw_result r;
    if (prof) mko_sw_forward_profile(prof, bias, t, tlen, ctx->gap_open, ctx->gap_extend, ctx->lanes_byte, ctx->lanes_word, &r);
    else mko_sw_forward(ctx->mat, q, cb, bias, qlen, t, tlen, ctx->gap_open, ctx->gap_extend, ctx->lanes_byte, ctx->lanes_word, &r);
    memset(out, 0, sizeof(*out));
    out->db_key = db_key; out->q_len = qlen; out->db_len = tlen; out->raw_score = r.score;
    if (r.t_end == -1) return 0;   /* nothing aligned: the reference returns uninitialised fields here (:385-388) */
    double evalue = mko_evalue(ctx->evaluer, (double) r.score, (double) qlen);
    float qcov = compute_cov(0, (unsigned) r.q_end, (unsigned) qlen);
    float tcov = compute_cov(0, (unsigned) r.t_end, (unsigned) tlen);
    if (!(evalue > ctx->eval_thr)) {
        if (prof) mko_sw_reverse_profile(prof, bias, t, ctx->gap_open, ctx->gap_extend, ctx->lanes_byte, ctx->lanes_word, &r);
        else mko_sw_reverse(ctx->mat, q, cb, bias, qlen, t, tlen, ctx->gap_open, ctx->gap_extend, ctx->lanes_byte, ctx->lanes_word, &r);
        qcov = compute_cov((unsigned) r.q_start, (unsigned) r.q_end, (unsigned) qlen);
        tcov = compute_cov((unsigned) r.t_start, (unsigned) r.t_end, (unsigned) tlen);
    }
    const unsigned int qStartPos = (unsigned int) r.q_start, dbStartPos = (unsigned int) r.t_start;
    const unsigned int qEndPos = (unsigned int) r.q_end, dbEndPos = (unsigned int) r.t_end;
    int d1 = abs((int) qEndPos - (int) qStartPos), d2 = abs((int) dbEndPos - (int) dbStartPos);
    unsigned int alnLength = (unsigned int) ((d1 > d2 ? d1 : d2) + 1);                  /* computeAlnLength */
    unsigned int qAlnLen = (qEndPos - qStartPos) > 1u ? (qEndPos - qStartPos) : 1u;
    unsigned int dbAlnLen = (dbEndPos - dbStartPos) > 1u ? (dbEndPos - dbStartPos) : 1u;
    /* estimateSeqIdByScorePerCol (Matcher.cpp:160-164): uint16 score, float division, double fma-free */
    unsigned short s16 = (unsigned short) r.score;
    unsigned int ml = qAlnLen > dbAlnLen ? qAlnLen : dbAlnLen;
    float seqId = (float) ((double) (s16 / (float) ml) * 0.1656 + 0.1141);
    seqId = seqId < 1.0f ? seqId : 1.0f;
    seqId = seqId > 0.0f ? seqId : 0.0f;
    out->bit_score = (int) (mko_bitscore(ctx->evaluer, (double) r.score) + 0.5);
    out->seq_id = seqId; out->evalue = evalue;
    out->q_start = r.q_start; out->q_end = r.q_end; out->db_start = r.t_start; out->db_end = r.t_end;
    out->aln_len = (int) alnLength; out->qcov = qcov; out->dbcov = tcov;
    if (r.rev_mismatch) return -1;
    return (evalue <= ctx->eval_thr) && ((int) alnLength >= ctx->aln_len_thr) ? 1 : 0;
}

/* Matcher::compareHits (Matcher.h:157-168) */
int mko_aln_compare(const void *a, const void *b) {
    const mko_aln_result *x = (const mko_aln_result *) a, *y = (const mko_aln_result *) b;
    if (x->evalue != y->evalue) return x->evalue < y->evalue ? -1 : 1;
    if (x->bit_score != y->bit_score) return x->bit_score > y->bit_score ? -1 : 1;
    if (x->db_len != y->db_len) return x->db_len < y->db_len ? -1 : 1;
    if (x->db_key != y->db_key) return x->db_key < y->db_key ? -1 : 1;
    return 0;
}

/* Matcher::resultToBuffer (Matcher.cpp:280-327) + Util::fastSeqIdToBuffer (Util.cpp:222-251) */
size_t mko_format_aln(char *buf, const mko_aln_result *r) {
    char *p = buf;
    p += sprintf(p, "%u\t%d\t", r->db_key, r->bit_score);
    if (r->seq_id == 1.0) {
        /* fastSeqIdToBuffer returns a pointer AT its NUL for 1.0 (Util.cpp:223-234), so the caller's
         * `*(tmpBuff-1) = '\t'` (Matcher.cpp:286) overwrites the last digit: the reference prints "1.00" */
        p += sprintf(p, "1.00");
    } else {
        *p++ = '0'; *p++ = '.';
        if (r->seq_id < 0.10) *p++ = '0';
        if (r->seq_id < 0.01) *p++ = '0';
        p += sprintf(p, "%d", (int) (r->seq_id * 1000));
    }
    p += sprintf(p, "\t%.3E\t%d\t%d\t%d\t%d\t%d\t%d\n", r->evalue, r->q_start, r->q_end, r->q_len, r->db_start, r->db_end, r->db_len);
    return (size_t) (p - buf);
}
