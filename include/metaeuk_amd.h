/* include/metaeuk_amd.h -- C ABI of the MI355X-native prefilter+align hot path.
 *
 * Plain C, opaque handles, caller-owned buffers, int return codes (0 = ok,
 * negative = error; mk_last_error() gives the message).  No C++/torch types
 * cross this boundary.  Every entry point names the reference interface it
 * replaces (paths relative to the reference tree, M/ = lib/mmseqs/).
 *
 * Drop-in seam: the reference's in-process seam for this path is
 *   QueryMatcher::matchQuery(Sequence*, unsigned, bool)   M/src/prefiltering/QueryMatcher.h:64
 *   Matcher::initQuery / Matcher::getSWResult(...)        M/src/alignment/Matcher.h:153-154,206
 * called once per query from Prefiltering::runSplit (Prefiltering.cpp:817-886)
 * and Alignment::run (Alignment.cpp:312-514).  A GPU wants batches, so the ABI
 * is the batch form of those two calls plus the one-time target set-up that
 * Prefiltering::getIndexTable (Prefiltering.cpp:514-553) performs.
 * INTEGRATION.md shows the `prefilter` / `align` command bodies a maintainer
 * would register in src/metaeuk.cpp:21-96 on top of this ABI.
 *
 * Process model: one process drives one GPU (mk_init(device)); the library owns its HIP streams, its device scratch and
 * a pool of pinned result blocks.  Calls into the library must come from one thread at a time (the reference's modules
 * are one process per step as well); the library itself runs the stages of mk_search on internal, persistent threads (one for the
 * prefilter, three alignment workers, a stream each).
 * Experiment and test switches (MK_PREFILTER_PATH, MK_PREFILTER_TIERS, MK_SW_WAVES_PER_CU, MK_SW_MULTI, MK_ALIGN_WORKERS, MK_SEARCH_CHUNK_QUERIES, ... --
 * the full list is DESIGN.md section 9) are read ONLY when MK_DEBUG=1 is set in the environment: a stray variable cannot change tiers, paths or launch
 * shapes of a production run, and setting one without MK_DEBUG has no effect.  Always honoured: RANK / WORLD_SIZE / LOCAL_RANK / LOCAL_WORLD_SIZE,
 * OMP_NUM_THREADS, MK_SHARD_TIMEOUT_S.  Defaults worth knowing: three alignment workers (each with per-chunk scratch in HBM), one prefilter thread,
 * query batches' device blocks pooled (at most 24 blocks of at most 1 GB kept).
 */
#ifndef METAEUK_AMD_H
#define METAEUK_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MK_OK 0
#define MK_ERR_ARG (-1)
#define MK_ERR_DEVICE (-2)       /* HIP error / no GPU: the product never falls back to the CPU */
#define MK_ERR_UNSUPPORTED (-3)  /* a reference corner this build does not restate (message says which) */
#define MK_ERR_SW_MISMATCH (-4)  /* forward/backward SW scores differ: the reference EXITs here too
                                    (StripedSmithWaterman.cpp:466-473) */

typedef struct mk_targetdb mk_targetdb;
typedef struct mk_queries mk_queries;

/* hit_t (M/src/prefiltering/QueryMatcher.h:33-49) */
typedef struct { uint32_t seq_id; int32_t pref_score; uint16_t diagonal; uint16_t pad_; } mk_hit;

/* Matcher::result_t fields that reach the 10-column output (M/src/alignment/Matcher.h:30-91) */
typedef struct {
    uint32_t db_key;
    int32_t bit_score;
    float seq_id;
    float qcov, dbcov;
    double evalue;
    int32_t q_start, q_end, q_len;
    int32_t db_start, db_end, db_len;
    int32_t aln_len;
    int32_t raw_score;
} mk_alignment;

/* Search parameters = the subset of the reference's `prefilter` / `align` flag lists
 * (M/src/commons/Parameters.cpp:387-455) that the predictexons path honours. */
typedef struct {
    float sensitivity;        /* -s                (5.7 for the BASELINE configs) */
    int kmer_score;           /* --k-score; INT32_MAX = derive from -s (Prefiltering.cpp:1005-1065) */
    int max_seqs;             /* --max-seqs 300 */
    int min_ungapped_score;   /* --min-ungapped-score 15 */
    int comp_bias_corr;       /* --comp-bias-corr 1 */
    float comp_bias_scale;    /* --comp-bias-corr-scale 1.0 */
    int mask;                 /* --mask 1 (tantan on targets) */
    float mask_prob;          /* --mask-prob 0.9 */
    int gap_open, gap_extend; /* --gap-open 11 --gap-extend 1 */
    double evalue_thr;        /* -e 100 (predictexons default, src/workflow/PredictExons.cpp:8-16) */
    int min_aln_len;          /* --min-aln-len = --min-exon-aa 11 */
    /* properties of the reference BUILD/HOST that leak into its results (SURVEY.md 8a-10,12): */
    int simd_lanes_byte;      /* 32 = AVX2, 16 = SSE4.1  (striped SW stripe length) */
    int simd_lanes_word;      /* 16 = AVX2,  8 = SSE4.1 */
    int simd_lanes_double;    /*  4 = AVX2,  2 = SSE4.1  (tantan partial sums) */
    uint64_t host_l2_bytes;   /* Util::getL2CacheSize() of the host being reproduced (tie order at --max-seqs) */
    int profile_search;       /* 1: the queries are profiles (mk_profiles_create) and the targets sequences -- the inverted search of
                                 searchslicedtargetprofile.sh.  Changes what the reference changes: k-mer threshold 134.35 - 6.15 s
                                 (Prefiltering.cpp:1038-1040), no self-score filter in the target index (:525-527), the background of
                                 the target masking from --sub-mat x8 instead of the seed matrix (:72-76) */
    int kmer_size;            /* -k: 0 = automatic = 6 below 3.35e9 target residues, else 7 (IndexTable::computeKmerSize, IndexTable.h:439-449);
                                 6 or 7 force it.  k = 7: spaced seed 11010110011 (Sequence.h:25), threshold 186.15 - 11.22 s
                                 (Prefiltering.cpp:1057-1059), 2-mer x 2-mer x 3-mer k-mer lists (KmerGenerator.cpp:41-86), a 20^7-cell
                                 table (10 GB of HBM).  The target side decides; a query batch follows the database it is searched against */
} mk_params;

/* ---- process-wide ---- */
int mk_init(int device_ordinal);            /* binds the calling process to one GPU (one process per GPU) */
const char *mk_last_error(void);
void mk_default_params(mk_params *p);       /* defaults of `metaeuk predictexons` (SURVEY.md 3.2 argv) */
int mk_device_name(char *buf, size_t cap);
int mk_device_memory(uint64_t *free_bytes, uint64_t *total_bytes);   /* HBM of the bound GPU (the split planner of the commands sizes the target side with it) */
int mk_host_threads(void);                  /* CPUs usable by this process (affinity and cgroup quota) */

/* ---- encoding: Sequence::mapSequence + SubstitutionMatrix::aa2num (Sequence.cpp:307-324) ---- */
void mk_encode(const char *ascii, size_t len, uint8_t *codes);

/* ---- target side: replaces Prefiltering::getIndexTable -> IndexBuilder::fillDatabase
 * (M/src/prefiltering/IndexBuilder.cpp:55-239): tantan masking, SequenceLookup, k-mer index;
 * plus ExtendedSubstitutionMatrix::calcScoreMatrix (ExtendedSubstitutionMatrix.cpp:20-69).
 * residues: encoded 0..20, concatenated; offsets[n+1].  Everything ends up resident in HBM. */
int mk_targetdb_create(const uint8_t *residues, const uint64_t *offsets, uint32_t n_targets,
                       const mk_params *params, mk_targetdb **out);
/* the target side as the ALIGNMENT stage needs it (Alignment.cpp opens the sequence DB, not the index): residues and offsets in HBM, the
 * alignment matrix, the e-value parameters of the whole database -- no masking, no k-mer index.  mk_align takes it (with a prefilter
 * result installed by mk_prefilter_result_set: the `align` command, the one-pass commands after a target-split prefilter); mk_prefilter /
 * mk_search refuse it. */
int mk_targetdb_create_sequences(const uint8_t *residues, const uint64_t *offsets, uint32_t n_targets, const mk_params *params, mk_targetdb **out);
void mk_targetdb_destroy(mk_targetdb *db);
uint64_t mk_targetdb_residues(const mk_targetdb *db);
uint64_t mk_targetdb_index_entries(const mk_targetdb *db);
/* copies of host-built artefacts, for tests */
int mk_targetdb_masked(const mk_targetdb *db, uint8_t *out /* residues */);

/* masking and indexing run on the device (mk_index.hip); these report what came out.  MK_INDEX_BUILD=host in the environment selects
 * the host builder (the check of the device one).  mk_targetdb_index_compare is a test hook: the number of differing slot words,
 * presence words, entries and masked residues between two databases (all ~0 when their sizes differ). */
uint64_t mk_targetdb_masked_residues(const mk_targetdb *db);
int mk_targetdb_kmer_size(const mk_targetdb *db);          /* 6 or 7 */
uint32_t mk_targetdb_longest_list(const mk_targetdb *db);  /* most targets one k-mer occurs in */
int mk_targetdb_index_compare(const mk_targetdb *a, const mk_targetdb *b, uint64_t diff[4]);

/* ---- precomputed index: the index DB of `createindex` / `indexdb` (type 9,
 * M/src/prefiltering/PrefilteringIndexReader.cpp:54-326): sequence DB + masked sequences + k-mer lists, so that a database is
 * masked and indexed once and every rank of a node loads the same file.  Both directions are interchangeable with the reference: it
 * reads what mk_index_write writes and mk_targetdb_open_index reads what its createindex writes (amino-acid targets, k = 6 and
 * k = 7, one split).  mk_index_write builds the lists in HBM and streams them into the file when mk_init was called (any size;
 * k = 7 from 3.35e9 residues on, IndexTable.h:439-449); without a GPU the host builder writes k = 6 index DBs.  A k = 7 index DB is
 * not materialised on the host when it is opened: offsets and 6-byte entries go from the mapped file to HBM in pieces.
 * The sequence DB is passed as it lies on disk: the data file(s) and the rows of its .index
 * (key, offset, length incl. "\n\0") in file order; target ids of the index = positions in that order (DBReader NOSORT,
 * util/indexdb.cpp:67-68).  Writes <index_db>, <index_db>.index, <index_db>.dbtype. */
int mk_index_write(const char *index_db, const char *seq_data, uint64_t seq_data_size, const uint32_t *keys, const uint64_t *offsets,
                   const uint32_t *lengths, uint32_t n, int seq_dbtype, const mk_params *params);
/* target side from an index DB instead of mk_targetdb_create: nothing is masked or indexed again */
int mk_targetdb_open_index(const char *index_db, const mk_params *params, mk_targetdb **out);
/* DB keys of the targets (Matcher::compareHits breaks its last tie on the DB key, Matcher.h:157-168): a database created from residues
 * has none until they are installed -- the alignment order then ties on the target index.  mk_alignment.db_key stays the target index. */
int mk_targetdb_set_keys(mk_targetdb *db, const uint32_t *keys, uint32_t n_targets);
/* DB keys of the targets of a database opened from an index or installed with mk_targetdb_set_keys (NULL otherwise) */
int mk_targetdb_keys(const mk_targetdb *db, const uint32_t **keys, uint32_t *n);
/* test hook, no GPU: the parts of an index DB a reader takes, as text files in out_dir (masked_targets.txt, index.txt, seqs.txt, meta.txt) */
int mk_index_dump(const char *index_db, const char *out_dir);

/* ---- query batch: Sequence::mapSequence + the per-residue inputs of calcLocalAaBiasCorrection /
 * createProfile / ssw_init.  Uploads the batch to HBM.  The batch handle also carries the results
 * of the two stages, the way the reference passes them through the pref_0 / search_res DBs. */
int mk_queries_create(const uint8_t *residues, const uint64_t *offsets, uint32_t n_queries,
                      const mk_params *params, mk_queries **out);
void mk_queries_destroy(mk_queries *q);
/* test hook: the per-residue arrays derived on the device for this batch -- k-mer threshold per k-mer start
 * (QueryMatcher.cpp:225-244; -1 = no k-mer / contains X), int8 diagonal correction (UngappedAlignment.cpp:391-396),
 * int8 SW composition bias (StripedSmithWaterman.cpp:1228-1235).  Each array has offsets[n_queries] entries. */
int mk_queries_derived(const mk_queries *q, int16_t *kmer_thr, int8_t *diag_corr, int8_t *sw_bias8);

/* ---- profile queries: `predictexons` against a PROFILE database runs the reference's sliced
 * target-profile search (M/data/workflow/searchslicedtargetprofile.sh:108,130,181,190): the profiles are the QUERIES of prefilter and
 * align, the ORF fragments the indexed targets (mk_targetdb_create with params->profile_search = 1), and swapresults turns the
 * lists round (mk_swap_alignments).  columns = the profiles' 25-byte columns (Sequence::PROFILE_READIN_SIZE, M/src/commons/Sequence.h:
 * 458-471: 20 int8 scores, query letter, consensus letter, neff, 2 gap bytes) concatenated WITHOUT the entries' trailing NUL;
 * col_offsets[n+1] in columns.  Replaces Sequence::mapProfile (Sequence.cpp:241-292), the profile k-mer lists
 * (KmerGenerator.cpp:30-39, Sequence.cpp:294-305), UngappedAlignment::createProfile's profile branch (UngappedAlignment.cpp:385-408)
 * and ssw_init / ssw_align_private<PROFILE_SEQ> (StripedSmithWaterman.cpp:296-298,1243-1300).  The handle is a query batch:
 * mk_prefilter, mk_align, mk_search and the result getters take it. */
int mk_profiles_create(const uint8_t *columns, const uint64_t *col_offsets, uint32_t n_profiles, const mk_params *params, mk_queries **out);
/* test hook: the derived per-column arrays -- query letters [cols], sorted scores + residue numbers [cols][40], alignment profile
 * [cols][32], k-mer threshold per start [cols] */
int mk_profiles_derived(const mk_queries *q, uint8_t *letters, int8_t *sorted40, int8_t *aln32, int16_t *kmer_thr);

/* swapresults (M/src/util/swapresults.cpp:283-333 + Matcher::result_t::swapResult, Matcher.h:93-115) on arrays: alns[offsets[i] ..
 * offsets[i+1]) = the accepted alignments of query i (db_key = target INDEX, as mk_align_result returns them); the result lists them
 * per target: db_key = query_keys[i] (NULL: i), query and target columns exchanged, the e-value recomputed from the printed bit score
 * for a search against `swapped_db_residues` (DBReader::getAminoAcidDBSize of the ORIGINAL target side = the profile DB: columns
 * - per-entry rounding, DBReader.cpp:589-598), the identity as re-read from its 3-decimal text, every list sorted with
 * Matcher::compareHits.  Host code, no GPU. */
typedef struct mk_swapped mk_swapped;
int mk_swap_alignments(const mk_alignment *alns, const uint64_t *offsets, uint32_t n_queries, const uint32_t *query_keys,
                       uint32_t n_targets, uint64_t swapped_db_residues, const mk_params *params, mk_swapped **out);
int mk_swapped_result(const mk_swapped *s, const mk_alignment **alns, const uint64_t **offsets /* n_targets + 1 */);
void mk_swapped_destroy(mk_swapped *s);

/* ---- prefilter: batch form of QueryMatcher::matchQuery (QueryMatcher.cpp:85-211).
 * Query i's hits are hits[offsets[i] .. offsets[i+1]), sorted like the reference (|score| desc,
 * seq_id asc); seq_id = target index.  The arrays are owned by the batch handle. */
int mk_prefilter(mk_targetdb *db, mk_queries *q, const mk_params *params);
int mk_prefilter_result(const mk_queries *q, const mk_hit **hits, const uint64_t **offsets /* n+1 */);
/* what the reference's prefilter logs about a run (Prefiltering.cpp:889-904, printStatistics :953-975), from the counters of the last
 * mk_prefilter / mk_search over the batch: mean over the queries of (similar k-mers / query length), index entries gathered per
 * query, queries that took the databaseHits overflow path, hits per query, median list length, queries without a hit */
typedef struct {
    double kmers_per_pos;
    uint64_t db_matches_per_seq, overflows, results_per_seq, empty_results, n_queries;
    uint32_t median_result_len, pad_;
} mk_prefilter_stats;
int mk_prefilter_statistics(const mk_queries *q, mk_prefilter_stats *out);
/* the six lines as the reference prints them ("246.638184 k-mers per position\n12 DB matches per sequence\n...") */
size_t mk_format_prefilter_statistics(char *buf, size_t cap, const mk_prefilter_stats *s, uint64_t max_results);

/* install a prefilter result produced elsewhere (the `align` command reading a pref_0 DB,
 * Alignment.cpp:312-358) */
int mk_prefilter_result_set(mk_queries *q, const mk_hit *hits, const uint64_t *offsets);

/* ---- align: batch form of Matcher::initQuery + getSWResult + Alignment::checkCriteria + sort
 * (Matcher.cpp:49-142, Alignment.cpp:346-405) over the batch's prefilter result.  Query i's accepted
 * alignments, in output order, are alns[offsets[i] .. offsets[i+1]). */
int mk_align(mk_targetdb *db, mk_queries *q, const mk_params *params);
int mk_align_result(const mk_queries *q, const mk_alignment **alns, const uint64_t **offsets /* n+1 */);

/* ---- producer of the query batch: `extractorfs --translate` as predictexons runs it
 * (util/extractorfs.cpp:19-159, Orf::findAll with orf-start-mode 1, all six frames, genetic code 1, contig start / end mode
 * 2).  Contigs are ASCII nucleotides (IUPAC codes, lower case, U allowed), concatenated, offsets[n_contigs + 1].
 * Fragment k is what the reference's renumbered ORF DB holds under key k. */
typedef struct mk_orfs mk_orfs;
typedef struct mk_orf {
    uint32_t contig;            /* index of the contig */
    uint32_t from, to;          /* Orf::writeOrfHeader coordinates on the contig (from > to on the minus strand) */
    uint8_t incomplete_start, incomplete_end, minus_strand, pad_;
} mk_orf;
int mk_extract_orfs(const char *nucleotides, const uint64_t *offsets, uint32_t n_contigs, int min_codons, mk_orfs **out);
/* fragments, their translations (ASCII, case preserved) as aa[aa_offsets[k] .. aa_offsets[k+1]) ; views owned by the handle */
int mk_orfs_result(const mk_orfs *o, const mk_orf **orfs, const uint64_t **aa_offsets, const char **aa, uint64_t *n_orfs);
/* the fragments as a query batch: the residue codes go from the translation kernel to the search without leaving HBM */
int mk_queries_from_orfs(const mk_orfs *o, const mk_params *params, mk_queries **out);
void mk_orfs_destroy(mk_orfs *o);
size_t mk_format_orf_header(char *buf, const mk_orf *o);   /* "contig<TAB>from(+|-)len[<TAB>complete]", Orf.cpp:434-452 */

/* ---- consumer of the alignments: `resultspercontig` + `collectoptimalset` of predictexons
 * (src/exonpredictor/resultspercontig.cpp:34-220, collectoptimalset.cpp:33-424) on the arrays in memory: per contig, target
 * and strand, the chain of compatible putative exons (ORF-fragment alignments) with the highest score.  Host code. */
typedef struct {               /* LocalParameters.h:138-146 / the predictexons flags */
    double evalue_thr;         /* --metaeuk-eval 0.001 (combined e-value of a set) */
    double target_cov_thr;     /* --metaeuk-tcov 0.5 */
    uint64_t max_intron, min_intron, min_exon_aa, max_aa_overlap, max_exon_sets;   /* 10000, 15, 11, 10, 1 */
    int32_t gap_open, gap_extend;                                                  /* --set-gap-open -1 --set-gap-extend -1 */
} mk_exon_params;
typedef struct {               /* Prediction (src/commons/PredictionParser.h:189-384) */
    uint32_t target; int32_t strand; uint32_t total_bit_score, n_exons;
    double evalue;
    uint32_t low_coord, high_coord;
    uint64_t first_exon;       /* exons[first_exon .. first_exon + n_exons) */
} mk_prediction;
typedef struct {               /* PotentialExon as printed by exonToBuffer (PredictionParser.h:88-137) */
    uint32_t orf; int32_t bit_score;
    double seq_id, evalue;     /* as the reference re-reads them from the alignment text */
    int32_t target_start, target_end, target_len, contig_start, contig_end, nucleotide_len, orf_from, orf_to;
} mk_exon;
typedef struct mk_predictions mk_predictions;
void mk_default_exon_params(mk_exon_params *p);
/* q = the batch made by mk_queries_from_orfs(orfs) after mk_search / mk_align.  target_keys (may be NULL: key = index) gives the
 * DB key of every target: the reference orders a contig's predictions by target KEY and prints it. */
int mk_predict_exons(const mk_targetdb *db, const mk_orfs *orfs, const mk_queries *q, const mk_exon_params *params,
                     const uint32_t *target_keys, mk_predictions **out);
/* the same on caller-owned arrays, no batch handle and no GPU: orfs[k] = fragment k (contig ascending), alns[aln_offsets[k] ..
 * aln_offsets[k+1]) its accepted alignments, db_residues = residues of the target database (the e-value of a set) */
int mk_predict_exons_arrays(const mk_orf *orfs, uint64_t n_orfs, uint32_t n_contigs, const mk_alignment *alns, const uint64_t *aln_offsets,
                            uint64_t db_residues, const mk_exon_params *params, const uint32_t *target_keys, mk_predictions **out);
/* predictions of contig c = predictions[contig_offsets[c] .. contig_offsets[c+1]); views owned by the handle */
int mk_predictions_result(const mk_predictions *p, const mk_prediction **predictions, const uint64_t **contig_offsets /* n_contigs+1 */,
                          const mk_exon **exons, uint64_t *n_predictions);
void mk_predictions_destroy(mk_predictions *p);
/* one line of the reference's prediction DB: predictionToBuffer + exonToBuffer */
size_t mk_format_prediction_exon(char *buf, const mk_prediction *p, const mk_exon *e);

/* ---- prefilter + align of the batch in one pipelined pass: the `search` workflow's two module calls
 * (blastp.sh:70,85 via predictexons.sh:68).  Same results as mk_prefilter followed by mk_align (both result
 * getters work afterwards); the stages run concurrently on two HIP streams, chunk by chunk. */
int mk_search(mk_targetdb *db, mk_queries *q, const mk_params *params);
/* The same pass without blocking the caller (round 5): mk_search_begin queues the batch and returns; mk_search_wait returns once every hit and
 * alignment of THAT batch is in host memory (its return code is the search's; the result getters work afterwards).  mk_search = begin + wait.
 * Batches are searched in the order they were begun, by the library's own threads (one for the prefilter, three alignment workers): the
 * prefilter of batch k + 1 runs beside the alignment of the last chunks of batch k -- what Prefiltering::runSplit / Alignment::run get from
 * one OpenMP loop over ALL queries (Prefiltering.cpp:817-886, Alignment.cpp:312-514: no barrier between a caller's batches) -- so a caller that
 * walks a DB in batches (the contig batches of `metaeuk-amd predictexons`, the steps of bench.py) pays the fill and drain of the two-stage
 * pipeline once per run instead of once per batch.  Between begin and wait the batch and the database must stay alive and untouched; the one
 * call meant to be made meanwhile is mk_queries_create for the NEXT batch (it uploads and derives on a stream of its own).  Every other
 * compute entry point first waits for the searches in flight to finish (their results stay collectable); mk_queries_destroy of a batch in
 * flight waits for it, mk_targetdb_destroy for all. */
int mk_search_begin(mk_targetdb *db, mk_queries *q, const mk_params *params);
int mk_search_wait(mk_queries *q);
/* Round 6: waits for every search in flight, then stops and JOINS the library's search threads (the reference's OpenMP team dies with its
 * parallel region, Prefiltering.cpp:817-886; a persistent engine needs an explicit end).  An in-process host calls it before it unloads the
 * library or tears its own state down; the library registers it with atexit when the first search begins, so a process that exits between
 * mk_search_begin and mk_search_wait leaves no thread behind.  The explicit call also gives the memory back that the library keeps between
 * calls (pooled device blocks, the stages' device and pinned scratch, pooled result blocks); handles the caller holds stay valid.  The next
 * mk_search / mk_search_begin starts the threads and allocates again.  Not to be called concurrently with another entry point. */
void mk_shutdown(void);

/* ---- kernel-level entry points (used by the parity tests and bench.py) ---- */
/* Smith-Waterman on explicit pairs: for pair p, query q_idx[p] vs target t_idx[p].
 * out5[p*5..] = score, q_end, t_end, q_start, t_start (starts = -1 unless with_start). */
int mk_sw_pairs(mk_targetdb *db, mk_queries *q, const mk_params *params,
                const uint32_t *q_idx, const uint32_t *t_idx, uint64_t n_pairs, int with_start,
                int32_t *out5);
/* ungapped diagonal scores (exact, unclamped) for (q_idx, t_idx, diagonal) triples:
 * UngappedAlignment::scoreSingelSequenceByCounterResult (UngappedAlignment.cpp:434-451) */
int mk_ungapped(mk_targetdb *db, mk_queries *q, const uint32_t *q_idx, const uint32_t *t_idx,
                const uint16_t *diagonal, uint64_t n, int32_t *out_scores);

/* timing of the last mk_prefilter / mk_align / mk_sw_pairs call, from HIP events on the
 * library's stream: names[i] -> ms and launches; used for the roofline line of bench.py */
typedef struct { const char *name; double ms; uint64_t launches; double alg_bytes; double cells; } mk_kernel_stat;
int mk_kernel_stats(mk_kernel_stat *out, int cap);
void mk_kernel_stats_reset(void);

/* ---- formatting: QueryMatcher::prefilterHitToBuffer (QueryMatcher.h:118-130),
 * Matcher::resultToBuffer (Matcher.cpp:280-327) ---- */
size_t mk_format_hit(char *buf, uint32_t db_key, int32_t score, uint16_t diagonal);
size_t mk_format_alignment(char *buf, const mk_alignment *a);
/* bulk forms (the result DB writers, bench.py's result digest, the at-scale parity test): the lines of hits[0..n) /
 * alns[0..n) back to back.  target_keys (may be NULL: key = index) maps seq_id to the DB key.  Return the bytes written;
 * 0 when cap is too small (32 B per hit, 160 B per alignment always suffice). */
size_t mk_format_hits(char *buf, size_t cap, const mk_hit *hits, uint64_t n, const uint32_t *target_keys);
size_t mk_format_alignments(char *buf, size_t cap, const mk_alignment *alns, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
