/* oracle/mko_orf.c -- TEST INFRASTRUCTURE (parity oracle), not product code.
 *
 * Plain-C restatement of `extractorfs --translate` as `metaeuk predictexons` runs it:
 *   Orf::setSequence / findAll / findForward     M/src/commons/Orf.cpp:118-345
 *   TranslateNucl (genetic code 1, IUPAC aware)   M/src/commons/TranslateNucl.h:252-500
 *   the extractorfs loop and header format        M/src/util/extractorfs.cpp:64-125, Orf.cpp:434-452
 * Pinned against oracle/_ref/ref_harness `orfs` (the reference's own Orf.cpp / TranslateNucl.h) by
 * tests/test_oracle_golden.py.
 */
#include "mko.h"
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

/* ---- TranslateNucl ---------------------------------------------------------------------------- */
/* base codes: 4-bit IUPAC sets in the order "-ACMGRSVTWYHKDBN" (eBase_gap .. eBase_N, TranslateNucl.h:252-269) */
static int base_code(unsigned char ch) {
    static const char charToBase[17] = "-ACMGRSVTWYHKDBN";
    static int lut[256], ready = 0;
    if (!ready) {
        for (int i = 0; i < 256; i++) lut[i] = 0;
        for (int i = 0; i < 16; i++) { lut[(unsigned char) charToBase[i]] = i; lut[(unsigned char) tolower(charToBase[i])] = i; }
        lut['U'] = 8; lut['u'] = 8; lut['X'] = 15; lut['x'] = 15;          /* :322-325 */
        for (int i = 0; i < 16; i++) lut[i] = i;                             /* :327-329 */
        ready = 1;
    }
    return lut[ch];
}

/* amino acid of every (i,j,k) base-code triple for the canonical code (initTranslationTable, :344-483) */
void mko_translation_table(char table[4096]) {
    static const char *ncbieaa = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    static const int expansions[4] = {1, 2, 4, 8};                           /* A C G T */
    static const int codonIdx[9] = {0, 2, 1, 0, 3, 0, 0, 0, 0};              /* T=0 C=1 A=2 G=3 */
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++)
            for (int k = 0; k < 16; k++) {
                char aa = '\0';
                for (int p = 0; p < 4; p++) {
                    const int x = expansions[p];
                    if (!(x & i)) continue;
                    for (int q = 0; q < 4; q++) {
                        const int y = expansions[q];
                        if (!(y & j)) continue;
                        for (int r = 0; r < 4; r++) {
                            const int z = expansions[r];
                            if (!(z & k)) continue;
                            const char ch = ncbieaa[16 * codonIdx[x] + 4 * codonIdx[y] + codonIdx[z]];
                            if (aa == '\0') aa = ch;
                            else if (aa != ch) {
                                if ((aa == 'B' || aa == 'D' || aa == 'N') && (ch == 'D' || ch == 'N')) aa = 'B';
                                else if ((aa == 'Z' || aa == 'E' || aa == 'Q') && (ch == 'E' || ch == 'Q')) aa = 'Z';
                                else if ((aa == 'J' || aa == 'I' || aa == 'L') && (ch == 'I' || ch == 'L')) aa = 'J';
                                else aa = 'X';
                            }
                        }
                    }
                }
                table[256 * i + 16 * j + k] = aa != '\0' ? aa : 'X';         /* gap in the codon: the 'X' the table starts with */
            }
}

/* TranslateNucl::translate (:488-503): one residue per codon, lower case when any base of the codon is */
static void translate(const char table[4096], const char *nucl, size_t len, char *aa) {
    for (size_t i = 0; i + 2 < len; i += 3) {
        int lower = 0;
        for (int k = 0; k < 3; k++) lower |= islower((unsigned char) nucl[i + k]) != 0;
        const char r = table[256 * base_code((unsigned char) nucl[i]) + 16 * base_code((unsigned char) nucl[i + 1]) + base_code((unsigned char) nucl[i + 2])];
        aa[i / 3] = lower ? (char) tolower((unsigned char) r) : r;
    }
}

/* ---- Orf ------------------------------------------------------------------------------------ */
static char complement_of(char c) {                                         /* Orf::iupacReverseComplementTable, Orf.cpp:48-52 */
    static const char *t =
        "................................................................"
        ".TVGH..CD..M.KN...YSAABW.R.......tvgh..cd..m.kn...ysaabw.r......"
        "................................................................"
        "................................................................";
    return t[(unsigned char) c];
}

static int is_codon(const char *c, const char *ref) { return c[0] == ref[0] && c[1] == ref[1] && c[2] == ref[2]; }
static int is_gap_or_n(const char *c) {                                      /* Orf.cpp:186-190 */
    return c[0] == 'N' || complement_of(c[0]) == '.' || c[1] == 'N' || complement_of(c[1]) == '.' || c[2] == 'N' || complement_of(c[2]) == '.';
}

typedef struct { mko_orf *v; size_t n, cap; } orf_vec;
static void push(orf_vec *o, mko_orf x) {
    if (o->n == o->cap) { o->cap = o->cap ? 2 * o->cap : 64; o->v = (mko_orf *) realloc(o->v, o->cap * sizeof(mko_orf)); }
    o->v[o->n++] = x;
}

/* Orf::findForward (Orf.cpp:220-345) on one strand; `seq` is padded with CHAR_MAX behind its end */
static void find_forward(const char *seq, size_t len, orf_vec *out, size_t minLength, size_t maxLength, size_t maxGaps, int startMode, int strand) {
    int inside[3] = {1, 1, 1}, hasStart[3] = {0, 0, 0};
    size_t gaps[3] = {0, 0, 0}, count[3] = {0, 0, 0}, from[3] = {0, 1, 2};
    for (size_t i = 0; i < len - 2; i += 3)
        for (size_t position = i; position < i + 3; position++) {
            char codon[3];
            for (int k = 0; k < 3; k++) codon[k] = seq[position + k] == CHAR_MAX ? CHAR_MAX : (char) (seq[position + k] & (unsigned char) ~0x20);
            const size_t frame = position % 3;
            const int thisIncomplete = codon[0] == CHAR_MAX || codon[1] == CHAR_MAX || codon[2] == CHAR_MAX;
            const char *nx = seq + position + 3;
            const int isLast = !thisIncomplete && (nx[0] == CHAR_MAX || nx[1] == CHAR_MAX || nx[2] == CHAR_MAX);
            int shouldStart;
            if (startMode == 0) shouldStart = !inside[frame] && is_codon(codon, "ATG");
            else if (startMode == 1) shouldStart = !inside[frame];
            else shouldStart = is_codon(codon, "ATG");
            if (shouldStart) { inside[frame] = 1; hasStart[frame] = 1; from[frame] = position; gaps[frame] = 0; count[frame] = 0; }
            const int stop = is_codon(codon, "TAA") || is_codon(codon, "TAG") || is_codon(codon, "TGA");
            if (inside[frame]) {
                if (!stop) count[frame]++;
                if (is_gap_or_n(codon)) gaps[frame]++;
            }
            if (inside[frame] && (stop || isLast)) {
                inside[frame] = 0;
                if (count[frame] == 0 && stop) continue;
                const size_t to = position + ((isLast && !stop) ? 2 : (size_t) -1);
                if (gaps[frame] > maxGaps || count[frame] > maxLength || count[frame] < minLength) continue;
                mko_orf o;
                o.from = from[frame]; o.to = to; o.incomplete_start = !hasStart[frame]; o.incomplete_end = !stop; o.strand = strand;
                push(out, o);
            }
        }
}

/* extractorfs for one contig (extractorfs.cpp:64-125 with contig start/end mode 2, both strands, all frames):
 * fragments in the order the reference writes them; from/to are the header coordinates (minus strand mirrored);
 * aa = concatenated translations, aa_off[k] .. aa_off[k+1] the k-th fragment.  Returns the number of fragments. */
size_t mko_extract_orfs(const char *contig, size_t len, size_t minLength, size_t maxLength, size_t maxGaps, int startMode,
                        mko_orf **orfs, char **aa, size_t **aa_off) {
    *orfs = NULL; *aa = NULL; *aa_off = (size_t *) calloc(1, sizeof(size_t));
    if (len < 3) return 0;                                                   /* Orf::setSequence */
    static char table[4096];
    static int ready = 0;
    if (!ready) { mko_translation_table(table); ready = 1; }
    const size_t PAD = 8;                                                    /* VECSIZE_INT CHAR_MAX sentinels behind the end */
    char *fwd = (char *) malloc(len + PAD), *rev = (char *) malloc(len + PAD);
    for (size_t i = 0; i < len; i++) {                                       /* the second assignment wins: only 'u' -> 't' (:131-134) */
        fwd[i] = (contig[i] == 'U') ? 'T' : contig[i];
        fwd[i] = (contig[i] == 'u') ? 't' : contig[i];
    }
    for (size_t i = 0; i < len; i++) { rev[i] = complement_of(fwd[len - i - 1]); if (rev[i] == '.') rev[i] = 'N'; }
    for (size_t i = len; i < len + PAD; i++) { fwd[i] = CHAR_MAX; rev[i] = CHAR_MAX; }
    orf_vec found = {NULL, 0, 0};
    find_forward(fwd, len, &found, minLength, maxLength, maxGaps, startMode, 0);
    find_forward(rev, len, &found, minLength, maxLength, maxGaps, startMode, 1);
    size_t total = 0;
    for (size_t k = 0; k < found.n; k++) total += (found.v[k].to - found.v[k].from + 1) / 3;
    char *out = (char *) malloc(total + 1);
    size_t *off = (size_t *) realloc(*aa_off, (found.n + 1) * sizeof(size_t));
    off[0] = 0;
    for (size_t k = 0; k < found.n; k++) {
        mko_orf *o = &found.v[k];
        const size_t nlen = o->to - o->from + 1;                             /* always a multiple of 3 */
        translate(table, (o->strand ? rev : fwd) + o->from, nlen, out + off[k]);
        off[k + 1] = off[k] + nlen / 3;
        if (o->strand) { o->from = (len - 1) - o->from; o->to = (len - 1) - o->to; }
    }
    free(fwd); free(rev);
    *orfs = found.v; *aa = out; *aa_off = off;
    return found.n;
}

/* Orf::writeOrfHeader (Orf.cpp:434-452) without the trailing newline: "key<TAB>from(+|-)len[<TAB>complete]" */
size_t mko_format_orf_header(char *buf, unsigned int key, const mko_orf *o) {
    const int len = abs((int) o->from - (int) o->to);
    const int complete = (o->incomplete_start ? 1 : 0) | ((o->incomplete_end ? 1 : 0) << 1);
    int n = sprintf(buf, "%u\t%u%c%d", key, (unsigned) o->from, o->from < o->to ? '+' : '-', len);
    if (complete) n += sprintf(buf + n, "\t%d", complete);
    return (size_t) n;
}
