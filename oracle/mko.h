/* oracle/mko.h -- TEST INFRASTRUCTURE (the parity oracle), not product code.
 *
 * A plain-C, scalar CPU restatement of the reference's prefilter+align hot path
 * (MMseqs2 as vendored in soedinglab/metaeuk, lib/mmseqs = "M/").  Every
 * function cites the reference file:line it follows.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
 * under oracle/; the product (metaeuk_amd/) never links or imports it.
 *
 * Pinning: this restatement is checked byte-for-byte against the reference's
 * own compiled code (oracle/_ref/ref_harness, built by oracle/Makefile.ref from
 * the sources under /root/reference) by tests/test_oracle_vs_ref.py and against
 * the committed fixtures under tests/golden/ that were generated with it.
 */
#ifndef MKO_H
#define MKO_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MKO_ALPH 21          /* 20 amino acids + X (code 20) */
#define MKO_X 20

/* ---- substitution matrix (M/src/commons/SubstitutionMatrix.cpp, BaseMatrix.cpp) ---- */
typedef struct {
    short sub[MKO_ALPH][MKO_ALPH];    /* BaseMatrix::subMatrix */
    double prob[MKO_ALPH][MKO_ALPH];  /* BaseMatrix::probMatrix (joint probabilities) */
    double pback[MKO_ALPH];           /* BaseMatrix::pBack */
    double lambda;
    const char *name;                 /* "blosum62.out" / "VTML80.out" */
} mko_submat;

enum { MKO_MAT_BLOSUM62 = 0, MKO_MAT_VTML80 = 1 };
void mko_submat_init(mko_submat *m, int which, float bit_factor, float score_bias);
void mko_map_sequence(const char *seq, int len, uint8_t *codes);
void mko_comp_bias(const mko_submat *m, const uint8_t *seq, int L, float scale, float *bias);

/* ---- extended (3-mer) score table (M/src/prefiltering/ExtendedSubstitutionMatrix.cpp) ---- */
typedef struct {
    int element_size;   /* 20^k */
    int row_size;       /* (element_size/64+1)*64 */
    short *score;
    uint32_t *index;
} mko_scoremat;
mko_scoremat *mko_scoremat_build(const mko_submat *m, int kmer);   /* alphabet 20 */
void mko_scoremat_free(mko_scoremat *s);

/* similar k-mer list for k=6 (M/src/prefiltering/KmerGenerator.cpp:107-216) */
size_t mko_kmer_list6(const mko_scoremat *three, const uint8_t *kmer, short threshold, uint64_t *out, size_t cap);
/* ... and for k=7: divide strategy {2,2,3} (setDivideStrategy :41-86, kmerSize % 3 == 1, reversed), three steps */
size_t mko_kmer_list7(const mko_scoremat *two, const mko_scoremat *three, const uint8_t *kmer, short threshold, uint64_t *out, size_t cap);
/* spaced seeds of Sequence.h:23,25 */
int mko_spaced_pattern(int k, const int **offsets);   /* returns the span (10 for k=6, 11 for k=7) */

/* ---- tantan masking + k-mer index (M/lib/tantan, Masker.cpp, IndexBuilder.cpp, IndexTable.h) ---- */
int mko_tantan_mask(const mko_submat *kmer_mat, uint8_t *seq, int L, double min_mask_prob, int simd_lanes);

typedef struct {
    int k;                   /* 6 or 7 */
    uint64_t table_size;     /* 20^k */
    uint64_t *offsets;       /* k = 6: table_size+1 (dense).  k = 7: n_kmers+1 over `kmers` (the oracle keeps the 1.28e9-cell table sparse) */
    uint64_t *kmers;         /* k = 7: the distinct k-mers with a list, ascending; NULL for k = 6 */
    uint64_t n_kmers;
    uint32_t *seq_id;        /* entries, sorted by (seq_id,pos) inside each k-mer list */
    uint16_t *pos;
    uint64_t n_entries;
    uint32_t n_seq;
    uint8_t *masked;         /* SequenceLookup: masked residues, concatenated */
    uint64_t *seq_off;       /* n_seq+1 */
    uint64_t masked_residues;
} mko_index;
mko_index *mko_index_build(const mko_submat *kmer_mat, const uint8_t *residues, const uint64_t *seq_off,
                           uint32_t n_seq, int kmer_thr, int mask, int simd_lanes);
mko_index *mko_index_build_k(const mko_submat *kmer_mat, const uint8_t *residues, const uint64_t *seq_off,
                             uint32_t n_seq, int kmer_thr, int mask, int simd_lanes, int k);
/* index list of a k-mer: entries [*o0, *o1) of seq_id / pos */
void mko_index_list(const mko_index *ix, uint64_t kmer, uint64_t *o0, uint64_t *o1);
void mko_index_free(mko_index *ix);

/* ---- prefilter for one query (M/src/prefiltering/QueryMatcher.cpp) ---- */
typedef struct { uint32_t seq_id; int32_t score; uint16_t diagonal; } mko_hit;
typedef struct {
    const mko_submat *kmer_mat;      /* VTML80 x8, bias -0.2 */
    const mko_submat *ungapped_mat;  /* BLOSUM62 x2, bias -0.2 */
    const mko_scoremat *three;
    const mko_scoremat *two;         /* k = 7 only */
    const mko_index *index;          /* its k decides the spaced seed and the list generator */
    int kmer_thr;
    int max_hits;                    /* min(--max-seqs, n_targets) */
    int min_diag_score;              /* 15 */
    int bin_count;                   /* CacheFriendlyOperations BINSIZE (QueryMatcher.cpp:422-450) */
    float bias_scale;                /* 1.0 */
} mko_prefilter_ctx;
typedef struct { uint64_t kmer_list_len; uint64_t db_matches; uint64_t diagonals; } mko_prefilter_stats;
/* returns number of hits written (<= max_hits), sorted like the reference; -1 on the (unsupported) overflow path */
int mko_prefilter_query(const mko_prefilter_ctx *ctx, const uint8_t *q, int L, mko_hit *out, mko_prefilter_stats *st);
int mko_bin_count_for(uint64_t db_size, uint64_t l2_cache_bytes);
int mko_ungapped_score(const int8_t *profile /* L x 21 */, int qlen, const uint8_t *t, int tlen, uint16_t diagonal);
void mko_ungapped_profile(const mko_submat *ungapped_mat, const uint8_t *q, int L, const float *bias, int8_t *profile);

/* ---- Smith-Waterman (M/src/alignment/StripedSmithWaterman.cpp) ---- */
typedef struct {
    int score;       /* score1 (255-saturated marker never leaves: word rerun is done inside) */
    int q_end, t_end, q_start, t_start;   /* -1 when not computed */
    int word;        /* 1 if the int16 pass was needed */
    int rev_mismatch;/* 1 if forward/backward scores differ (reference EXITs) */
} mko_sw_result;
/* query composition bias int8 + matrix bias (ssw_init, StripedSmithWaterman.cpp:1216-1345) */
void mko_sw_query_init(const mko_submat *m, const uint8_t *q, int L, float bias_scale, int8_t *comp_bias8, int *bias);
/* forward pass only (score/ends); lanes_byte/lanes_word = SIMD lanes of the reference build (32/16 for AVX2) */
void mko_sw_forward(const mko_submat *m, const uint8_t *q, const int8_t *comp_bias8, int bias, int qlen,
                    const uint8_t *t, int tlen, int gap_open, int gap_extend, int lanes_byte, int lanes_word,
                    mko_sw_result *r);
/* reverse pass for start positions given a forward result */
void mko_sw_reverse(const mko_submat *m, const uint8_t *q, const int8_t *comp_bias8, int bias, int qlen,
                    const uint8_t *t, int tlen, int gap_open, int gap_extend, int lanes_byte, int lanes_word,
                    mko_sw_result *r);

/* ---- e-value (M/src/alignment/EvalueComputation.h + M/lib/alp) ---- */
typedef struct { double lambda, K, logK, a_I, b_I, a_J, b_J, alpha_I, beta_I, alpha_J, beta_J, sigma, tau,
                 vi_y_thr, vj_y_thr, c_y_thr; double db_res; } mko_evaluer;
void mko_evaluer_init(mko_evaluer *e, uint64_t db_residues);   /* BLOSUM62, gap 11/1 */
double mko_evalue(const mko_evaluer *e, double score, double qlen);
double mko_bitscore(const mko_evaluer *e, double score);

/* ---- align one query against its prefilter hits and format (Matcher.cpp, Alignment.cpp) ---- */
typedef struct {
    const mko_submat *mat;           /* BLOSUM62 x2, bias 0 */
    const mko_evaluer *evaluer;
    int gap_open, gap_extend;
    double eval_thr;                 /* 100 */
    int aln_len_thr;                 /* 11 */
    int lanes_byte, lanes_word;
    float bias_scale;
} mko_align_ctx;
typedef struct {
    uint32_t db_key; int bit_score; float seq_id; double evalue;
    int q_start, q_end, q_len, db_start, db_end, db_len; int aln_len; float qcov, dbcov;
    int raw_score;
} mko_aln_result;
/* computes one pair: returns 1 if it passes Alignment::checkCriteria */
int mko_align_pair(const mko_align_ctx *ctx, const uint8_t *q, const int8_t *comp_bias8, int bias, int qlen,
                   const uint8_t *t, int tlen, uint32_t db_key, mko_aln_result *out);
int mko_aln_compare(const void *a, const void *b);
size_t mko_format_aln(char *buf, const mko_aln_result *r);
size_t mko_format_hit(char *buf, const mko_hit *h);

/* ---- profile queries (Sequence::mapProfile, profile k-mer lists, PROFILE_SEQ Smith-Waterman, swapresults) ---- */
typedef struct {
    int L;
    uint8_t *query, *consensus;      /* the profile's query / consensus letters (numSequence / numConsensusSequence) */
    int8_t *aln;                     /* profile_for_alignment: [21][L], score / 4, X row 0 */
    short *sorted_score;             /* [L][20] descending (Util::rankedDescSort20) */
    uint8_t *sorted_idx;             /* [L][20] residue numbers in that order */
} mko_profile;
mko_profile *mko_profile_map(const char *data, int seqLen);     /* seqLen = (entry length - 1) / 25 (DBReader::getSeqLen) */
void mko_profile_free(mko_profile *p);
void mko_ranked_desc_sort20(short *val, uint8_t *index);
size_t mko_profile_kmer_list(const mko_profile *p, int pos, short threshold, uint64_t *out, size_t cap);
int mko_prefilter_profile(const mko_prefilter_ctx *ctx, const mko_profile *p, mko_hit *out, mko_prefilter_stats *st);
int mko_sw_profile_bias(const mko_profile *p);
void mko_sw_forward_profile(const mko_profile *p, int bias, const uint8_t *t, int tlen, int gap_open, int gap_extend,
                            int lanes_byte, int lanes_word, mko_sw_result *r);
void mko_sw_reverse_profile(const mko_profile *p, int bias, const uint8_t *t, int gap_open, int gap_extend,
                            int lanes_byte, int lanes_word, mko_sw_result *r);
int mko_align_pair_profile(const mko_align_ctx *ctx, const mko_profile *prof, int bias, const uint8_t *t, int tlen, uint32_t db_key,
                           mko_aln_result *out);
void mko_swap_result(const mko_evaluer *ev, mko_aln_result *r, uint32_t new_db_key);

/* ---- extractorfs --translate: M/src/commons/Orf.cpp, TranslateNucl.h, util/extractorfs.cpp ---- */
typedef struct { size_t from, to; int incomplete_start, incomplete_end, strand; } mko_orf;   /* from/to = header coordinates */
void mko_translation_table(char table[4096]);     /* amino acid of every IUPAC base-code triple, genetic code 1 */
size_t mko_extract_orfs(const char *contig, size_t len, size_t minLength, size_t maxLength, size_t maxGaps, int startMode,
                        mko_orf **orfs, char **aa, size_t **aa_off);
size_t mko_format_orf_header(char *buf, unsigned int key, const mko_orf *o);

/* ---- resultspercontig + collectoptimalset ---- */
typedef struct {     /* one ORF->target alignment as the 10 printed columns carry it, plus the ORF's header coordinates */
    unsigned int target; int bit_score; double seq_id, evalue; int q_start, q_end, db_start, db_end, db_len;
    unsigned int orf; int orf_from, orf_to;
} mko_exon_aln;
typedef struct {
    double evalue_thr, target_cov_thr; size_t max_intron, min_intron, min_exon_aa, max_aa_overlap, max_exon_sets;
    int gap_open, gap_extend; uint64_t db_residues;
} mko_exon_params;
void mko_exon_params_default(mko_exon_params *P, uint64_t db_residues);
char *mko_predict_exons(mko_exon_aln *alns, size_t n, const mko_exon_params *P, size_t *n_predictions);

#ifdef __cplusplus
}
#endif
#endif
