/* oracle/mko_cli.c -- TEST INFRASTRUCTURE (parity oracle).  See mko.h.
 * Command-line driver producing the same files as oracle/_ref/ref_harness so the two can be
 * diffed byte-for-byte:
 *   mko_cli pipeline <targets.txt> <queries.txt> <outdir> [-s 5.7] [--dump] [--lanes-byte 32]
 *                    [--lanes-word 16] [--tantan-lanes 4] [--l2 BYTES] [--max-seqs 300]
 *   mko_cli sw <targets.txt> <queries.txt> <pairs.txt> <out.txt>
 *   mko_cli submat blosum62|vtml80 <bitFactor> <bias>
 */
#include "mko.h"
#include <limits.h>
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct { char **s; int *len; size_t n; } lines_t;

static lines_t read_lines(const char *path) {
    lines_t L = {0, 0, 0};
    FILE *f = fopen(path, "r");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    size_t cap = 1024;
    L.s = (char **) malloc(cap * sizeof(char *));
    L.len = (int *) malloc(cap * sizeof(int));
    char *line = NULL; size_t lc = 0; ssize_t r;
    while ((r = getline(&line, &lc, f)) >= 0) {
        while (r > 0 && (line[r - 1] == '\n' || line[r - 1] == '\r' || line[r - 1] == ' ')) r--;
        if (L.n == cap) { cap *= 2; L.s = (char **) realloc(L.s, cap * sizeof(char *)); L.len = (int *) realloc(L.len, cap * sizeof(int)); }
        L.s[L.n] = (char *) malloc((size_t) r + 1);
        memcpy(L.s[L.n], line, (size_t) r); L.s[L.n][r] = 0;
        L.len[L.n] = (int) r;
        L.n++;
    }
    free(line);
    fclose(f);
    return L;
}

static void encode(const lines_t *L, uint8_t **res, uint64_t **off) {
    uint64_t tot = 0;
    for (size_t i = 0; i < L->n; i++) tot += (uint64_t) L->len[i];
    *res = (uint8_t *) malloc(tot + 1);
    *off = (uint64_t *) malloc((L->n + 1) * sizeof(uint64_t));
    uint64_t o = 0;
    for (size_t i = 0; i < L->n; i++) {
        (*off)[i] = o;
        mko_map_sequence(L->s[i], L->len[i], *res + o);
        o += (uint64_t) L->len[i];
    }
    (*off)[L->n] = o;
}

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

static int kmer_threshold(float sensitivity) {   /* Prefiltering.cpp:1051-1053, k = 6 */
    float base = 163.2;
    float best = base - (sensitivity * 8.917);
    return (int) best;
}

static int cmd_pipeline(int argc, char **argv) {
    lines_t T = read_lines(argv[2]), Q = read_lines(argv[3]);
    const char *outdir = argv[4];
    float sens = 5.7f; int dump = 0, lb = 32, lw = 16, tl = 4, maxSeqs = 300;
    long l2 = sysconf(_SC_LEVEL2_CACHE_SIZE);
    if (l2 <= 0) l2 = 262144;   /* Util::getL2CacheSize, Util.cpp:317-332 */
    for (int a = 5; a < argc; a++) {
        if (!strcmp(argv[a], "-s")) sens = (float) atof(argv[++a]);
        else if (!strcmp(argv[a], "--dump")) dump = 1;
        else if (!strcmp(argv[a], "--lanes-byte")) lb = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--lanes-word")) lw = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--tantan-lanes")) tl = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--l2")) l2 = atol(argv[++a]);
        else if (!strcmp(argv[a], "--max-seqs")) maxSeqs = atoi(argv[++a]);
    }
    mkdir(outdir, 0755);
    uint8_t *tres, *qres; uint64_t *toff, *qoff;
    encode(&T, &tres, &toff);
    encode(&Q, &qres, &qoff);
    mko_submat kmerMat, ungMat, alnMat;
    mko_submat_init(&kmerMat, MKO_MAT_VTML80, 8.0f, -0.2f);      /* Prefiltering.cpp:68 */
    mko_submat_init(&ungMat, MKO_MAT_BLOSUM62, 2.0f, -0.2f);     /* :69 */
    mko_submat_init(&alnMat, MKO_MAT_BLOSUM62, 2.0f, 0.0f);      /* Alignment.cpp:152 */
    const int kmerThr = kmer_threshold(sens);
    double t0 = now();
    mko_scoremat *three = mko_scoremat_build(&kmerMat, 3);
    double tExt = now() - t0; t0 = now();
    mko_index *ix = mko_index_build(&kmerMat, tres, toff, (uint32_t) T.n, kmerThr, 1, tl);
    double tIdx = now() - t0;
    char path[4096];
    if (dump) {
        static const char *alpha = "ACDEFGHIKLMNPQRSTVWYX";
        snprintf(path, sizeof(path), "%s/masked_targets.txt", outdir);
        FILE *f = fopen(path, "w");
        for (size_t i = 0; i < T.n; i++) {
            for (uint64_t p = toff[i]; p < toff[i + 1]; p++) fputc(alpha[ix->masked[p]], f);
            fputc('\n', f);
        }
        fclose(f);
        snprintf(path, sizeof(path), "%s/index.txt", outdir);
        f = fopen(path, "w");
        for (uint64_t k = 0; k < ix->table_size; k++) {
            if (ix->offsets[k + 1] == ix->offsets[k]) continue;
            fprintf(f, "%llu", (unsigned long long) k);
            for (uint64_t e = ix->offsets[k]; e < ix->offsets[k + 1]; e++) fprintf(f, " %u:%u", ix->seq_id[e], (unsigned) ix->pos[e]);
            fputc('\n', f);
        }
        fclose(f);
    }
    mko_prefilter_ctx pc;
    pc.kmer_mat = &kmerMat; pc.ungapped_mat = &ungMat; pc.three = three; pc.index = ix; pc.kmer_thr = kmerThr;
    pc.max_hits = maxSeqs; pc.min_diag_score = 15; pc.bin_count = mko_bin_count_for(T.n, (uint64_t) l2); pc.bias_scale = 1.0f;
    mko_evaluer ev;
    mko_evaluer_init(&ev, toff[T.n]);
    mko_align_ctx ac;
    ac.mat = &alnMat; ac.evaluer = &ev; ac.gap_open = 11; ac.gap_extend = 1; ac.eval_thr = 100.0; ac.aln_len_thr = 11;
    ac.lanes_byte = lb; ac.lanes_word = lw; ac.bias_scale = 1.0f;

    char **prefOut = (char **) calloc(Q.n, sizeof(char *)), **alnOut = (char **) calloc(Q.n, sizeof(char *));
    char **statOut = (char **) calloc(Q.n, sizeof(char *));
    unsigned long long totalHits = 0, alignments = 0, passed = 0, dbMatches = 0;
    double kmersPerPos = 0, cells = 0;
    t0 = now();
#pragma omp parallel
    {
        mko_hit *hits = (mko_hit *) malloc((size_t) (maxSeqs + 1) * sizeof(mko_hit));
        mko_aln_result *res = (mko_aln_result *) malloc((size_t) (maxSeqs + 1) * sizeof(mko_aln_result));
        char buf[512];
#pragma omp for schedule(dynamic, 1) reduction(+: totalHits, alignments, passed, dbMatches, kmersPerPos, cells)
        for (size_t id = 0; id < Q.n; id++) {
            const uint8_t *q = qres + qoff[id];
            const int L = (int) (qoff[id + 1] - qoff[id]);
            mko_prefilter_stats st;
            int nh = mko_prefilter_query(&pc, q, L, hits, &st);
            if (nh < 0) { fprintf(stderr, "query %zu: unsupported overflow path\n", id); nh = 0; }
            totalHits += (unsigned long long) nh;
            dbMatches += st.db_matches;
            kmersPerPos += L > 0 ? (double) st.kmer_list_len / (double) L : 0;
            size_t cap = (size_t) nh * 40 + 1, n = 0;
            prefOut[id] = (char *) malloc(cap);
            for (int h = 0; h < nh; h++) n += mko_format_hit(prefOut[id] + n, &hits[h]);
            prefOut[id][n] = 0;
            if (dump) { statOut[id] = (char *) malloc(96); snprintf(statOut[id], 96, "%llu\t%llu\n", (unsigned long long) st.kmer_list_len, (unsigned long long) st.db_matches); }
            /* align */
            int nr = 0;
            if (nh > 0) {
                int8_t *cb = (int8_t *) malloc((size_t) L + 1);
                int bias;
                mko_sw_query_init(&alnMat, q, L, 1.0f, cb, &bias);
                for (int h = 0; h < nh; h++) {
                    const uint32_t t = hits[h].seq_id;
                    const int tl_ = (int) (toff[t + 1] - toff[t]);
                    int ok = mko_align_pair(&ac, q, cb, bias, L, tres + toff[t], tl_, t, &res[nr]);
                    alignments++;
                    cells += (double) L * (double) tl_;
                    if (ok < 0) { fprintf(stderr, "Score of forward/backward SW differ (q %zu t %u)\n", id, t); exit(1); }
                    if (ok) { nr++; passed++; }
                }
                free(cb);
            }
            if (nr > 1) qsort(res, (size_t) nr, sizeof(mko_aln_result), mko_aln_compare);
            alnOut[id] = (char *) malloc((size_t) nr * 128 + 1);
            n = 0;
            for (int r = 0; r < nr; r++) n += mko_format_aln(alnOut[id] + n, &res[r]);
            alnOut[id][n] = 0;
            (void) buf;
        }
        free(hits); free(res);
    }
    double tRun = now() - t0;
    snprintf(path, sizeof(path), "%s/pref.txt", outdir);
    FILE *f = fopen(path, "w");
    for (size_t id = 0; id < Q.n; id++) { fprintf(f, ">%zu\n", id); fputs(prefOut[id], f); }
    fclose(f);
    snprintf(path, sizeof(path), "%s/aln.txt", outdir);
    f = fopen(path, "w");
    for (size_t id = 0; id < Q.n; id++) { fprintf(f, ">%zu\n", id); fputs(alnOut[id], f); }
    fclose(f);
    if (dump) {
        snprintf(path, sizeof(path), "%s/stats.txt", outdir);
        f = fopen(path, "w");
        for (size_t id = 0; id < Q.n; id++) fputs(statOut[id], f);
        fclose(f);
    }
    printf("{\"queries\": %zu, \"targets\": %zu, \"kmer_thr\": %d, \"bin_count\": %d, \"masked_residues\": %llu, \"index_entries\": %llu, "
           "\"t_extmat\": %.4f, \"t_index\": %.4f, \"t_prefilter_align\": %.4f, \"pref_hits\": %llu, \"kmers_per_pos\": %.4f, "
           "\"db_matches\": %llu, \"alignments\": %llu, \"passed\": %llu, \"cells_fwd\": %.0f}\n",
           Q.n, T.n, kmerThr, pc.bin_count, (unsigned long long) ix->masked_residues, (unsigned long long) ix->n_entries,
           tExt, tIdx, tRun, totalHits, kmersPerPos / (double) (Q.n ? Q.n : 1), dbMatches, alignments, passed, cells);
    return 0;
}

static int cmd_sw(int argc, char **argv) {
    (void) argc;
    lines_t T = read_lines(argv[2]), Q = read_lines(argv[3]), P = read_lines(argv[4]);
    uint8_t *tres, *qres; uint64_t *toff, *qoff;
    encode(&T, &tres, &toff);
    encode(&Q, &qres, &qoff);
    uint64_t dbres = toff[T.n];
    int lb = 32, lw = 16;
    for (int a = 6; a < argc; a++) {
        if (!strcmp(argv[a], "--dbres")) dbres = (uint64_t) atoll(argv[++a]);
        else if (!strcmp(argv[a], "--lanes-byte")) lb = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--lanes-word")) lw = atoi(argv[++a]);
    }
    mko_submat alnMat;
    mko_submat_init(&alnMat, MKO_MAT_BLOSUM62, 2.0f, 0.0f);
    mko_evaluer ev;
    mko_evaluer_init(&ev, dbres);
    mko_align_ctx ac;
    ac.mat = &alnMat; ac.evaluer = &ev; ac.gap_open = 11; ac.gap_extend = 1; ac.eval_thr = DBL_MAX; ac.aln_len_thr = 0;
    ac.lanes_byte = lb; ac.lanes_word = lw; ac.bias_scale = 1.0f;
    FILE *f = fopen(argv[5], "w");
    char buf[512];
    for (size_t p = 0; p < P.n; p++) {
        int q, t;
        if (sscanf(P.s[p], "%d %d", &q, &t) != 2) continue;
        const int L = (int) (qoff[q + 1] - qoff[q]);
        int8_t *cb = (int8_t *) malloc((size_t) L + 1);
        int bias;
        mko_sw_query_init(&alnMat, qres + qoff[q], L, 1.0f, cb, &bias);
        mko_aln_result r;
        int ok = mko_align_pair(&ac, qres + qoff[q], cb, bias, L, tres + toff[t], (int) (toff[t + 1] - toff[t]), (uint32_t) t, &r);
        if (ok < 0) fprintf(stderr, "fwd/bwd mismatch q %d t %d\n", q, t);
        size_t n = mko_format_aln(buf, &r);
        fprintf(f, "%d\t%d\t", q, t);
        fwrite(buf, 1, n, f);
        free(cb);
    }
    fclose(f);
    return 0;
}

static int cmd_submat(int argc, char **argv) {
    (void) argc;
    mko_submat m;
    mko_submat_init(&m, !strcmp(argv[2], "blosum62") ? MKO_MAT_BLOSUM62 : MKO_MAT_VTML80, (float) atof(argv[3]), (float) atof(argv[4]));
    for (int i = 0; i < MKO_ALPH; i++) for (int j = 0; j < MKO_ALPH; j++) printf("%d%c", m.sub[i][j], j + 1 == MKO_ALPH ? '\n' : ' ');
    for (int i = 0; i < MKO_ALPH; i++) printf("%.17g%c", m.pback[i], i + 1 == MKO_ALPH ? '\n' : ' ');
    for (int i = 0; i < MKO_ALPH; i++) for (int j = 0; j < MKO_ALPH; j++) printf("%.17g%c", m.prob[i][j], j + 1 == MKO_ALPH ? '\n' : ' ');
    return 0;
}

/* extractorfs --translate for every contig (one per line): same output format as `ref_harness orfs` */
static int cmd_orfs(int argc, char **argv) {
    size_t minLength = 15;
    for (int i = 4; i + 1 < argc; i++) if (!strcmp(argv[i], "--min-length")) minLength = (size_t) atol(argv[i + 1]);
    lines_t C = read_lines(argv[2]);
    FILE *out = fopen(argv[3], "w");
    if (!out) return 1;
    size_t total = 0;
    char hdr[128];
    for (size_t key = 0; key < C.n; key++) {
        fprintf(out, ">%zu\n", key);
        mko_orf *orfs; char *aa; size_t *off;
        const size_t n = mko_extract_orfs(C.s[key], strlen(C.s[key]), minLength, 32734, (size_t) INT_MAX, 1, &orfs, &aa, &off);
        for (size_t k = 0; k < n; k++) {
            mko_format_orf_header(hdr, (unsigned int) key, &orfs[k]);
            fprintf(out, "%s\t%.*s\n", hdr, (int) (off[k + 1] - off[k]), aa + off[k]);
        }
        total += n;
        free(orfs); free(aa); free(off);
    }
    fclose(out);
    printf("{\"contigs\": %zu, \"orfs\": %zu}\n", C.n, total);
    return 0;
}

/* resultspercontig + collectoptimalset over the outputs of `orfs` and `pipeline`: same format as `ref_harness exons` */
static int cmd_exons(int argc, char **argv) {
    (void) argc;
    lines_t T = read_lines(argv[2]), C = read_lines(argv[3]), O = read_lines(argv[4]), A = read_lines(argv[5]);
    FILE *out = fopen(argv[6], "w");
    if (!out) return 1;
    uint64_t residues = 0;
    for (size_t i = 0; i < T.n; i++) residues += (uint64_t) T.len[i];
    mko_exon_params P;
    mko_exon_params_default(&P, residues);
    /* ORF k: contig, header coordinates */
    size_t nOrf = 0;
    for (size_t i = 0; i < O.n; i++) if (O.s[i][0] != '>') nOrf++;
    unsigned int *orfContig = (unsigned int *) malloc((nOrf + 1) * sizeof(unsigned int));
    int *orfFrom = (int *) malloc((nOrf + 1) * sizeof(int)), *orfTo = (int *) malloc((nOrf + 1) * sizeof(int));
    size_t k = 0;
    for (size_t i = 0; i < O.n; i++) {
        if (O.s[i][0] == '>') continue;
        unsigned int contig, from; int len; char sign;
        if (sscanf(O.s[i], "%u\t%u%c%d", &contig, &from, &sign, &len) != 4) { fprintf(stderr, "bad ORF header: %s\n", O.s[i]); return 1; }
        orfContig[k] = contig; orfFrom[k] = (int) from; orfTo[k] = sign == '+' ? (int) from + len : (int) from - len;
        k++;
    }
    /* alignment blocks: block k = ORF k; collect per contig */
    size_t nAln = 0;
    for (size_t i = 0; i < A.n; i++) if (A.s[i][0] != '>') nAln++;
    mko_exon_aln *alns = (mko_exon_aln *) malloc((nAln + 1) * sizeof(mko_exon_aln));
    unsigned int *alnContig = (unsigned int *) malloc((nAln + 1) * sizeof(unsigned int));
    size_t a = 0;
    long block = -1;
    for (size_t i = 0; i < A.n; i++) {
        if (A.s[i][0] == '>') { block++; continue; }
        mko_exon_aln *x = &alns[a];
        char sid[64], ev[64];
        int qlen;
        if (sscanf(A.s[i], "%u\t%d\t%63s\t%63s\t%d\t%d\t%d\t%d\t%d\t%d", &x->target, &x->bit_score, sid, ev, &x->q_start, &x->q_end, &qlen, &x->db_start, &x->db_end, &x->db_len) != 10) {
            fprintf(stderr, "bad alignment line: %s\n", A.s[i]); return 1;
        }
        x->seq_id = strtod(sid, NULL); x->evalue = strtod(ev, NULL);
        x->orf = (unsigned int) block; x->orf_from = orfFrom[block]; x->orf_to = orfTo[block];
        alnContig[a] = orfContig[block];
        a++;
    }
    /* ORFs of a contig are consecutive, so are their alignments */
    size_t total = 0, at = 0;
    for (size_t c = 0; c < C.n; c++) {
        fprintf(out, ">%zu\n", c);
        size_t e = at;
        while (e < nAln && alnContig[e] == c) e++;
        size_t np = 0;
        char *txt = mko_predict_exons(alns + at, e - at, &P, &np);
        fputs(txt, out);
        free(txt);
        total += np;
        at = e;
    }
    fclose(out);
    printf("{\"contigs\": %zu, \"predictions\": %zu}\n", C.n, total);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: mko_cli pipeline|sw|submat ...\n"); return 2; }
    if (!strcmp(argv[1], "pipeline") && argc >= 5) return cmd_pipeline(argc, argv);
    if (!strcmp(argv[1], "sw") && argc >= 6) return cmd_sw(argc, argv);
    if (!strcmp(argv[1], "submat") && argc >= 5) return cmd_submat(argc, argv);
    if (!strcmp(argv[1], "orfs") && argc >= 4) return cmd_orfs(argc, argv);
    if (!strcmp(argv[1], "exons") && argc >= 7) return cmd_exons(argc, argv);
    return 2;
}
