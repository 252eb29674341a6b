// metaeuk_amd/csrc/mk_host.hpp -- host-side (CPU) pieces of the prefilter+align path that are
// not kernels: substitution matrices, sequence encoding, composition bias, the similar-3-mer
// table, tantan masking + k-mer index construction, ALP e-values, per-query hit selection and
// result formatting.  Each function mirrors one reference routine (cited in mk_host.cpp); all
// floating-point expressions keep the reference's float/double mix because their rounded
// results feed integer decisions that must be bit-exact.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/metaeuk_amd.h"

namespace mk {

// Environment: launcher_env = what a launcher tells a rank (always honoured); knob = experiment / test switches (MK_*), read only
// under MK_DEBUG=1 -- a stray variable cannot change tiers, paths or launch shapes of a production run (list: DESIGN.md section 9)
const char *launcher_env(const char *name);
const char *knob(const char *name);                 // nullptr unless MK_DEBUG=1 and the variable is set and non-empty
long knob_long(const char *name, long dflt);

constexpr int ALPH = 21;       // 20 amino acids + X
constexpr int XCODE = 20;
constexpr int KMER = 6;
constexpr int SPAN = 10;       // spaced seed 1101010011
extern const int SPACED6[6];
extern const int SPACED7[7];   // spaced seed 11010110011 (k = 7, Sequence.h:25), span 11

struct SubMat {
    short sub[ALPH][ALPH];
    double prob[ALPH][ALPH];
    double pback[ALPH];
    double lambda;
    std::string name;
};
enum { MAT_BLOSUM62 = 0, MAT_VTML80 = 1 };
void build_submat(SubMat &m, int which, float bitFactor, float scoreBias);
void encode(const char *s, size_t n, uint8_t *codes);
void comp_bias(const SubMat &m, const uint8_t *seq, int L, float scale, float *bias);
int kmer_threshold(float sensitivity, int kmerScoreOverride);
int kmer_threshold_profile(float sensitivity);
int kmer_threshold_profile_k7(float sensitivity);
int kmer_threshold_k7(float sensitivity, int kmerScoreOverride);     // sequence search, k = 7 (Prefiltering.cpp:1057-1059)
int bin_count_for(uint64_t dbSize, uint64_t l2Bytes);

// similar-3-mer table: row r (= 3-mer index) lists all 8000 3-mers by descending score
struct ScoreMat3 {
    std::vector<int16_t> score;   // [8000][8000]
    std::vector<uint16_t> index;  // [8000][8000]
    // per row: how many of the 8000 3-mers score exactly s / at least s, s = histLo .. histLo + histRange - 1.
    // With these the number of similar k-mers of a position is a ~100-term sum (no enumeration): sizing only.
    int histLo = 0, histRange = 0;
    std::vector<uint16_t> hist, cum;   // [8000][histRange]
};
void build_scoremat3(const SubMat &kmerMat, ScoreMat3 &out);

int tantan_mask(const SubMat &kmerMat, uint8_t *seq, int L, double minMaskProb, int lanes);

struct TargetIndex {
    std::vector<uint64_t> offsets;   // 20^6 + 1
    std::vector<uint64_t> entries;   // packed: seqId | pos << 32
    std::vector<uint8_t> masked;     // masked residues (SequenceLookup)
    uint64_t maskedResidues = 0;
};
// addressOrder: lists ordered by the k-mers' device table address (mk_host.cpp, KMER_ADDR_LETTER); false = the reference's
// Indexer numbering (what an index file holds)
void build_index(const SubMat &kmerMat, const uint8_t *residues, const uint64_t *seqOff, uint32_t nSeq,
                 int kmerThr, bool mask, float maskProb, int tantanLanes, TargetIndex &out, bool addressOrder = true, int kmerSize = 6);
// k = 7 (databases from 3.35e9 residues on, IndexTable.h:439-449): the similar 2-mers of every 2-mer, like build_scoremat3 (rows of 400,
// descending score, stable over the cartesian order with the first letter slowest); index = the 2-mer's number a0 + 20 a1
void build_scoremat2(const SubMat &kmerMat, std::vector<int16_t> &score, std::vector<uint16_t> &index);
void kmer3_number_of_address(uint16_t numOf[8000]);   // inverse of kmer3_address_table
void index_to_address_order(TargetIndex &ix);
void kmer3_address_table(uint16_t addrOf[8000]);   // reference 3-mer number -> address code (tile << 6 | in-quad positions), a permutation of 0..7999
const uint8_t *kmer_addr_letters();                 // [20]: residue -> quad << 2 | position in the quad (the tiled address order)
uint32_t kmer_cell(uint32_t addrFirst, uint32_t addrSecond);   // table cell of the k-mer made of two 3-mers (their address codes)

struct Evaluer {
    double lambda, K, logK, a_I, b_I, a_J, b_J, alpha_I, beta_I, alpha_J, beta_J, sigma, tau, vi_y_thr, vj_y_thr, c_y_thr, dbRes;
    void init(uint64_t dbResidues);
    double evalue(double score, double qLen) const;
    double bitScore(double score) const;
};

// candidate diagonal after double-hit detection + scoring
struct Cand { uint32_t id; uint16_t diag; int32_t score; uint32_t ordinal; };
// QueryMatcher::matchQuery tail: best diagonal per target, threshold, top-N, sort.  `cands` in canonical
// hit order (ascending ordinal).  selfScore = exact ungapped self score (only used when saturated).
int select_hits(std::vector<Cand> &cands, int binCount, int maxHits, int minDiagScore, int selfScore, mk_hit *out);

// ---- extractorfs --translate (Orf.cpp, TranslateNucl.h) ----
// comp[c] = Orf::iupacReverseComplementTable ('.' = not a nucleotide code); base[c] = TranslateNucl::sm_BaseToIdx (4-bit IUPAC sets)
void build_orf_tables(char comp[256], uint8_t base[256]);
// residue of every base-code triple (256 i + 16 j + k) under genetic code 1, ambiguity resolved like TranslateNucl::initTranslationTable
void build_translation_table(char table[4096]);
// Orf::writeOrfHeader without the newline: "key<TAB>from(+|-)len[<TAB>complete]"
size_t format_orf_header(char *buf, uint32_t key, uint32_t from, uint32_t to, bool incompleteStart, bool incompleteEnd);

float compute_cov(unsigned int startPos, unsigned int endPos, unsigned int len);
size_t format_hit(char *buf, uint32_t key, int32_t score, uint16_t diag);
size_t format_alignment(char *buf, const mk_alignment &a);
bool alignment_less(const mk_alignment &a, const mk_alignment &b);

}  // namespace mk
