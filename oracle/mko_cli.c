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
#include <math.h>
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

static int merged_hit_cmp(const void *a, const void *b) {   /* hit_t::compareHitsByScoreAndId on the joined lists */
    const mko_hit *x = (const mko_hit *) a, *y = (const mko_hit *) b;
    int ax = abs(x->score), ay = abs(y->score);
    if (ax != ay) return ax > ay ? -1 : 1;
    if (x->seq_id != y->seq_id) return x->seq_id < y->seq_id ? -1 : 1;
    return 0;
}

static int cmd_pipeline(int argc, char **argv) {
    lines_t T = read_lines(argv[2]), Q = read_lines(argv[3]);
    const char *outdir = argv[4];
    float sens = 5.7f; int dump = 0, lb = 32, lw = 16, tl = 4, maxSeqs = 300, kmerSize = 6, splits = 1;
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
        else if (!strcmp(argv[a], "-k")) kmerSize = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--split")) splits = atoi(argv[++a]);      /* target splits: --split N --split-mode 0 */
    }
    if (splits < 1 || (splits > 1 && dump)) { fprintf(stderr, "--split N >= 1 (and no --dump with splits)\n"); return 2; }
    if (kmerSize != 6 && kmerSize != 7) { fprintf(stderr, "-k 6 or 7\n"); return 2; }
    mkdir(outdir, 0755);
    uint8_t *tres, *qres; uint64_t *toff, *qoff;
    encode(&T, &tres, &toff);
    encode(&Q, &qres, &qoff);
    mko_submat kmerMat, ungMat, alnMat;
    mko_submat_init(&kmerMat, MKO_MAT_VTML80, 8.0f, -0.2f);      /* Prefiltering.cpp:68 */
    mko_submat_init(&ungMat, MKO_MAT_BLOSUM62, 2.0f, -0.2f);     /* :69 */
    mko_submat_init(&alnMat, MKO_MAT_BLOSUM62, 2.0f, 0.0f);      /* Alignment.cpp:152 */
    int kmerThr = kmer_threshold(sens);
    if (kmerSize == 7) {                                        /* Prefiltering.cpp:1057-1059 */
        float base = 186.15;
        float best = base - (sens * 11.22);
        kmerThr = (int) best;
    }
    double t0 = now();
    mko_scoremat *three = mko_scoremat_build(&kmerMat, 3);
    mko_scoremat *two = kmerSize == 7 ? mko_scoremat_build(&kmerMat, 2) : NULL;
    double tExt = now() - t0; t0 = now();
    /* TARGET_DB_SPLIT (Prefiltering.cpp:352-362,733-750): residue-balanced target ranges (DBReader::decomposeDomainByAminoAcid over the
     * index lengths = residues + 2), an index per range, --max-seqs reduced to maxRes / N + 4 sqrt(maxRes / N) */
    uint32_t *splitFirst = (uint32_t *) calloc((size_t) splits + 1, sizeof(uint32_t));
    if (splits > 1) {
        uint64_t total = 0;
        for (size_t i = 0; i < T.n; i++) total += (uint64_t) T.len[i] + 2;
        const uint64_t chunk = (total + (uint64_t) splits - 1) / (uint64_t) splits;
        uint64_t acc = 0; int w = 0;
        uint32_t *per = (uint32_t *) calloc((size_t) splits, sizeof(uint32_t));
        for (size_t i = 0; i < T.n; i++) { if (acc >= chunk) { acc = 0; w++; } acc += (uint64_t) T.len[i] + 2; per[w]++; }
        for (int r = 0; r < splits; r++) splitFirst[r + 1] = splitFirst[r] + per[r];
        free(per);
        size_t maxRes = (size_t) maxSeqs < T.n ? (size_t) maxSeqs : T.n;
        size_t four = (size_t) (4 * sqrt((double) maxRes / (double) splits));
        maxSeqs = (int) (maxRes / (size_t) splits + four);
        if (maxSeqs < 1) maxSeqs = 1;
    } else splitFirst[1] = (uint32_t) T.n;
    mko_index **ixs = (mko_index **) calloc((size_t) splits, sizeof(mko_index *));
    for (int r = 0; r < splits; r++) {
        const uint32_t f = splitFirst[r], n = splitFirst[r + 1] - splitFirst[r];
        uint64_t *so = (uint64_t *) malloc(((size_t) n + 1) * sizeof(uint64_t));
        for (uint32_t i = 0; i <= n; i++) so[i] = toff[f + i] - toff[f];
        ixs[r] = mko_index_build_k(&kmerMat, tres + toff[f], so, n, kmerThr, 1, tl, kmerSize);
        free(so);
    }
    mko_index *ix = ixs[0];
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
        for (uint64_t k = 0; k < ix->table_size && !ix->kmers; k++) {
            if (ix->offsets[k + 1] == ix->offsets[k]) continue;
            fprintf(f, "%llu", (unsigned long long) k);
            for (uint64_t e = ix->offsets[k]; e < ix->offsets[k + 1]; e++) fprintf(f, " %u:%u", ix->seq_id[e], (unsigned) ix->pos[e]);
            fputc('\n', f);
        }
        fclose(f);
    }
    mko_prefilter_ctx pc;
    pc.kmer_mat = &kmerMat; pc.ungapped_mat = &ungMat; pc.three = three; pc.two = two; pc.index = ix; pc.kmer_thr = kmerThr;
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
        mko_hit *hits = (mko_hit *) malloc(((size_t) maxSeqs * (size_t) splits + 1) * sizeof(mko_hit));
        mko_aln_result *res = (mko_aln_result *) malloc(((size_t) maxSeqs * (size_t) splits + 1) * sizeof(mko_aln_result));
        char buf[512];
#pragma omp for schedule(dynamic, 1) reduction(+: totalHits, alignments, passed, dbMatches, kmersPerPos, cells)
        for (size_t id = 0; id < Q.n; id++) {
            const uint8_t *q = qres + qoff[id];
            const int L = (int) (qoff[id + 1] - qoff[id]);
            mko_prefilter_stats st;
            int nh = 0;
            if (splits == 1) nh = mko_prefilter_query(&pc, q, L, hits, &st);
            else {
                /* every split on its own (own BINSIZE), the lists joined and sorted by (score, key) without another cut (mergeTargetSplits, :379-496) */
                mko_prefilter_stats one;
                memset(&st, 0, sizeof(st));
                for (int r = 0; r < splits && nh >= 0; r++) {
                    mko_prefilter_ctx ps = pc;
                    ps.index = ixs[r];
                    ps.bin_count = mko_bin_count_for(ixs[r]->n_seq, (uint64_t) l2);
                    int k = mko_prefilter_query(&ps, q, L, hits + nh, &one);
                    if (k < 0) { nh = -1; break; }
                    for (int h = 0; h < k; h++) hits[nh + h].seq_id += splitFirst[r];
                    nh += k;
                    st.kmer_list_len += one.kmer_list_len; st.db_matches += one.db_matches;
                }
                if (nh > 1) qsort(hits, (size_t) nh, sizeof(mko_hit), merged_hit_cmp);
            }
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
    lines_t T = read_lines(argv[2]), C = read_lines(argv[3]), O = read_lines(argv[4]), A = read_lines(argv[5]);
    FILE *out = fopen(argv[6], "w");
    if (!out) return 1;
    uint64_t residues = 0;
    for (size_t i = 0; i < T.n; i++) residues += (uint64_t) T.len[i];
    for (int i = 7; i + 1 < argc; i++) if (!strcmp(argv[i], "--dbres")) residues = (uint64_t) atoll(argv[i + 1]);   /* a profile DB */
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


/* The profile-target search of predictexons (M/data/workflow/searchslicedtargetprofile.sh): profiles = queries, the
 * fragments = indexed targets; prefilter, align, swapresults.
 *   mko_cli profilesearch <profile DB data file> <its .index> <fragments.txt> <outdir> [-s 4] [-e 100] [--l2 BYTES] [--lanes-byte 32] ...
 * writes pref.txt / aln.txt ('>profile key' blocks in key order) and swapped.txt ('>fragment number' blocks, every fragment).
 * Parameters as Search.cpp:357-399 sets them: --max-seqs = max(300, #fragments) for the prefilter, the e-value threshold scaled by
 * #fragments / #profiles and passed on as text ("%g": Parameters::createParameterString streams the double), swapresults -e DBL_MAX.
 * The workflow aligns twice (key lists per slice, then the merged lists again): with --max-accept / --max-rejected at INT_MAX both
 * passes accept the same pairs, so one pass is the result. */
typedef struct { unsigned int key; size_t off, len; } pindex_t;
static int pindex_cmp(const void *a, const void *b) { unsigned int x = ((const pindex_t *) a)->key, y = ((const pindex_t *) b)->key; return x < y ? -1 : x > y; }

static int cmd_profilesearch(int argc, char **argv) {
    const char *outdir = argv[5];
    float sens = 4.0f; double evalThr = 100.0, evalAbs = -1.0; int lb = 32, lw = 16, tl = 4;
    const char *keysPath = NULL;   /* line i = DB key of fragment i; fragments.txt is in the order of the data offsets of the fragment DB
                                      (the prefilter's target numbering, DBReader LINEAR_ACCCESS: Prefiltering.cpp:163) */
    long l2 = sysconf(_SC_LEVEL2_CACHE_SIZE);
    if (l2 <= 0) l2 = 262144;
    for (int a = 6; a < argc; a++) {
        if (!strcmp(argv[a], "-s")) sens = (float) atof(argv[++a]);
        else if (!strcmp(argv[a], "-e")) evalThr = atof(argv[++a]);
        else if (!strcmp(argv[a], "--lanes-byte")) lb = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--lanes-word")) lw = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--tantan-lanes")) tl = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--l2")) l2 = atol(argv[++a]);
        else if (!strcmp(argv[a], "--keys")) keysPath = argv[++a];
        else if (!strcmp(argv[a], "--eval-abs")) evalAbs = atof(argv[++a]);   /* the alignment threshold as given (a sample of a larger profile set) */
    }
    mkdir(outdir, 0755);
    /* profile DB */
    FILE *f = fopen(argv[2], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
    fseek(f, 0, SEEK_END); const size_t dataSize = (size_t) ftell(f); fseek(f, 0, SEEK_SET);
    char *pdata = (char *) malloc(dataSize + 1);
    if (fread(pdata, 1, dataSize, f) != dataSize) return 2;
    fclose(f);
    lines_t IX = read_lines(argv[3]);
    const size_t nProf = IX.n;
    pindex_t *pix = (pindex_t *) malloc((nProf + 1) * sizeof(pindex_t));
    size_t lengthSum = 0;
    for (size_t i = 0; i < nProf; i++) {
        unsigned long long o, l;
        if (sscanf(IX.s[i], "%u\t%llu\t%llu", &pix[i].key, &o, &l) != 3) return 2;
        pix[i].off = (size_t) o; pix[i].len = (size_t) l; lengthSum += (size_t) l;
    }
    qsort(pix, nProf, sizeof(pindex_t), pindex_cmp);
    const uint64_t profDbRes = (uint64_t) (lengthSum / 25 - nProf);          /* DBReader::getAminoAcidDBSize, DBReader.cpp:589-598 */
    /* fragments */
    lines_t T = read_lines(argv[4]);
    uint8_t *tres; uint64_t *toff;
    encode(&T, &tres, &toff);
    uint32_t *fragKey = (uint32_t *) malloc((T.n + 1) * sizeof(uint32_t)), *fragOfKey = (uint32_t *) malloc((T.n + 1) * sizeof(uint32_t));
    for (size_t i = 0; i < T.n; i++) fragKey[i] = (uint32_t) i;
    if (keysPath) {
        lines_t K = read_lines(keysPath);
        if (K.n != T.n) { fprintf(stderr, "--keys: %zu keys for %zu fragments\n", K.n, T.n); return 2; }
        for (size_t i = 0; i < T.n; i++) fragKey[i] = (uint32_t) strtoul(K.s[i], NULL, 10);
    }
    for (size_t i = 0; i < T.n; i++) {
        if (fragKey[i] >= T.n) { fprintf(stderr, "fragment keys must be 0 .. n-1\n"); return 2; }
        fragOfKey[fragKey[i]] = (uint32_t) i;
    }
    mko_submat kmerMat, ungMat, alnMat;
    mko_submat_init(&kmerMat, MKO_MAT_BLOSUM62, 8.0f, -0.2f);   /* Prefiltering.cpp:72-76: for profile queries the k-mer matrix (here
                                                                    only the background of the target masking) is --sub-mat, not the seed matrix */
    mko_submat_init(&ungMat, MKO_MAT_BLOSUM62, 2.0f, -0.2f);
    mko_submat_init(&alnMat, MKO_MAT_BLOSUM62, 2.0f, 0.0f);
    float base = 134.35;                                                     /* Prefiltering.cpp:1038-1040: profile search, k = 6 */
    float kmerThrBest = base - (sens * 6.15);
    const int kmerThr = (int) kmerThrBest;
    mko_index *ix = mko_index_build(&kmerMat, tres, toff, (uint32_t) T.n, 0 /* Prefiltering.cpp:525-527 */, 1, tl);
    mko_prefilter_ctx pc;
    pc.kmer_mat = &kmerMat; pc.ungapped_mat = &ungMat; pc.three = NULL; pc.two = NULL; pc.index = ix; pc.kmer_thr = kmerThr;
    pc.max_hits = (int) (T.n > 300 ? T.n : 300);                              /* Search.cpp:372 */
    pc.min_diag_score = 15; pc.bin_count = mko_bin_count_for(T.n, (uint64_t) l2); pc.bias_scale = 1.0f;
    {   /* Search.cpp:366-368 and the text round trip of the parameter string */
        evalThr *= ((float) T.n) / nProf;
        char txt[64];
        snprintf(txt, sizeof(txt), "%g", evalThr);
        evalThr = strtod(txt, NULL);
        if (evalAbs >= 0) evalThr = evalAbs;
    }
    mko_evaluer ev, evSwap;
    mko_evaluer_init(&ev, toff[T.n]);                                         /* Alignment.cpp:262: the target DB = the fragments */
    mko_evaluer_init(&evSwap, profDbRes);                                     /* swapresults.cpp:76-77,102 */
    mko_align_ctx ac;
    ac.mat = &alnMat; ac.evaluer = &ev; ac.gap_open = 11; ac.gap_extend = 1; ac.eval_thr = evalThr; ac.aln_len_thr = 11;
    ac.lanes_byte = lb; ac.lanes_word = lw; ac.bias_scale = 1.0f;
    char **prefOut = (char **) calloc(nProf, sizeof(char *)), **alnOut = (char **) calloc(nProf, sizeof(char *));
    mko_aln_result **alnRes = (mko_aln_result **) calloc(nProf, sizeof(mko_aln_result *));
    int *alnCnt = (int *) calloc(nProf, sizeof(int));
    unsigned long long totalHits = 0, passed = 0, kmers = 0, dbm = 0, positions = 0;
#pragma omp parallel
    {
        mko_hit *hits = (mko_hit *) malloc(((size_t) pc.max_hits + 1) * sizeof(mko_hit));
#pragma omp for schedule(dynamic, 1) reduction(+: totalHits, passed, kmers, dbm, positions)
        for (size_t id = 0; id < nProf; id++) {
            const int L = (int) (((pix[id].len > 1 ? pix[id].len : 1) - 1) / 25);   /* DBReader::getSeqLen, DBReader.h:224-227 */
            mko_profile *p = mko_profile_map(pdata + pix[id].off, L);
            mko_prefilter_stats st;
            int nh = mko_prefilter_profile(&pc, p, hits, &st);
            if (nh < 0) { fprintf(stderr, "profile %u: unsupported overflow path\n", pix[id].key); nh = 0; }
            totalHits += (unsigned long long) nh; kmers += st.kmer_list_len; dbm += st.db_matches; positions += (unsigned long long) L;
            size_t n = 0;
            prefOut[id] = (char *) malloc((size_t) nh * 40 + 1);
            for (int h = 0; h < nh; h++) { mko_hit pr = hits[h]; pr.seq_id = fragKey[pr.seq_id]; n += mko_format_hit(prefOut[id] + n, &pr); }
            prefOut[id][n] = 0;
            mko_aln_result *res = (mko_aln_result *) malloc(((size_t) nh + 1) * sizeof(mko_aln_result));
            int nr = 0;
            const int bias = mko_sw_profile_bias(p);
            for (int h = 0; h < nh; h++) {
                const uint32_t t = hits[h].seq_id;
                int ok = mko_align_pair_profile(&ac, p, bias, tres + toff[t], (int) (toff[t + 1] - toff[t]), fragKey[t], &res[nr]);
                if (ok < 0) { fprintf(stderr, "Score of forward/backward SW differ (profile %u fragment %u)\n", pix[id].key, t); exit(1); }
                if (ok) { nr++; passed++; }
            }
            if (nr > 1) qsort(res, (size_t) nr, sizeof(mko_aln_result), mko_aln_compare);
            alnOut[id] = (char *) malloc((size_t) nr * 128 + 1);
            n = 0;
            for (int r = 0; r < nr; r++) n += mko_format_aln(alnOut[id] + n, &res[r]);
            alnOut[id][n] = 0;
            alnRes[id] = res; alnCnt[id] = nr;
            mko_profile_free(p);
        }
        free(hits);
    }
    char path[4096];
    snprintf(path, sizeof(path), "%s/pref.txt", outdir);
    f = fopen(path, "w");
    for (size_t id = 0; id < nProf; id++) { fprintf(f, ">%u\n", pix[id].key); fputs(prefOut[id], f); }
    fclose(f);
    snprintf(path, sizeof(path), "%s/aln.txt", outdir);
    f = fopen(path, "w");
    for (size_t id = 0; id < nProf; id++) { fprintf(f, ">%u\n", pix[id].key); fputs(alnOut[id], f); }
    fclose(f);
    /* swapresults (util/swapresults.cpp:283-333): every printed record is parsed back (bit score, 3-decimal identity), its e-value
     * recomputed for the swapped search, the lists sorted with Matcher::compareHits; every fragment gets an entry */
    size_t *cnt = (size_t *) calloc(T.n + 1, sizeof(size_t));
    for (size_t id = 0; id < nProf; id++) for (int r = 0; r < alnCnt[id]; r++) cnt[alnRes[id][r].db_key + 1]++;
    for (size_t t = 0; t < T.n; t++) cnt[t + 1] += cnt[t];
    mko_aln_result *sw = (mko_aln_result *) malloc((cnt[T.n] + 1) * sizeof(mko_aln_result));
    size_t *fill = (size_t *) malloc((T.n + 1) * sizeof(size_t));
    memcpy(fill, cnt, (T.n + 1) * sizeof(size_t));
    for (size_t id = 0; id < nProf; id++) {
        for (int r = 0; r < alnCnt[id]; r++) {
            mko_aln_result x = alnRes[id][r];
            const uint32_t frag = x.db_key;
            char buf[256], sid[32];
            mko_format_aln(buf, &x);
            sscanf(buf, "%*u\t%*d\t%31s", sid);
            x.seq_id = (float) strtod(sid, NULL);                            /* Matcher::parseAlignmentRecord, Matcher.cpp:218-219 */
            mko_swap_result(&evSwap, &x, pix[id].key);
            sw[fill[frag]++] = x;
        }
    }
    snprintf(path, sizeof(path), "%s/swapped.txt", outdir);
    f = fopen(path, "w");
    char buf[256];
    for (size_t t = 0; t < T.n; t++) {
        fprintf(f, ">%zu\n", t);
        const size_t n = cnt[t + 1] - cnt[t];
        if (n > 1) qsort(sw + cnt[t], n, sizeof(mko_aln_result), mko_aln_compare);
        for (size_t k = 0; k < n; k++) { size_t l = mko_format_aln(buf, &sw[cnt[t] + k]); fwrite(buf, 1, l, f); }
    }
    fclose(f);
    printf("{\"profiles\": %zu, \"fragments\": %zu, \"kmer_thr\": %d, \"eval_thr\": %.17g, \"profile_db_residues\": %llu, \"masked_residues\": %llu, \"pref_hits\": %llu, "
           "\"passed\": %llu, \"kmers_per_pos\": %.4f, \"db_matches_per_profile\": %.1f}\n",
           nProf, T.n, kmerThr, evalThr, (unsigned long long) profDbRes, (unsigned long long) ix->masked_residues, totalHits, passed,
           positions ? (double) kmers / (double) positions : 0.0, nProf ? (double) dbm / (double) nProf : 0.0);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: mko_cli pipeline|sw|submat ...\n"); return 2; }
    if (!strcmp(argv[1], "pipeline") && argc >= 5) return cmd_pipeline(argc, argv);
    if (!strcmp(argv[1], "sw") && argc >= 6) return cmd_sw(argc, argv);
    if (!strcmp(argv[1], "submat") && argc >= 5) return cmd_submat(argc, argv);
    if (!strcmp(argv[1], "orfs") && argc >= 4) return cmd_orfs(argc, argv);
    if (!strcmp(argv[1], "exons") && argc >= 7) return cmd_exons(argc, argv);
    if (!strcmp(argv[1], "profilesearch") && argc >= 6) return cmd_profilesearch(argc, argv);
    return 2;
}
