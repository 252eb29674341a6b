// metaeuk_amd/csrc/mk_exons.cpp -- consumer of the hot path's alignments: the exon sets
// of `predictexons`, straight from the alignment arrays in memory.  Replaces
//   resultspercontig   src/exonpredictor/resultspercontig.cpp:34-220   (regroup the ORF alignments by contig, (target, orf) order)
//   collectoptimalset  src/exonpredictor/collectoptimalset.cpp:33-424  (compatible-exon chain of maximal score per target and strand)
// without the two result DBs and the text re-parse between them.  Host code: a few compatible-pair tests per contig and
// target, nothing for a GPU; contigs are independent and run in parallel.  The reference passes alignments on as TEXT, so
// the sequence identity and the e-value an exon carries are what strtod reads back from the printed columns; that round
// trip is reproduced for the exons that reach the output.
#include "mk_exons.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>

namespace mk {

namespace {

// PotentialExon (src/commons/PredictionParser.h:14-186)
struct Exon {
    uint32_t orf, target; int strand;
    int bitScore;
    uint64_t aln;                    // index of the alignment (sequence identity / e-value are fetched for output only)
    int targetStart, targetEnd, targetLen;
    int contigStart, contigEnd, nucleotideLen, aaLen;
    int orfFrom, orfTo;
    bool used;
};

// PotentialExon::setByAln (:15-62)
Exon make_exon(const mk_alignment &a, uint64_t alnIndex, uint32_t target, uint32_t orfKey, int orfFrom, int orfTo) {
    Exon e;
    e.orf = orfKey; e.target = target; e.bitScore = a.bit_score; e.aln = alnIndex;
    e.targetStart = a.db_start; e.targetEnd = a.db_end; e.targetLen = a.db_len;
    e.orfFrom = orfFrom; e.orfTo = orfTo;
    if (orfFrom < orfTo) { e.contigStart = orfFrom + a.q_start * 3; e.contigEnd = orfFrom + a.q_end * 3 + 2; e.strand = 1; }
    else { e.contigStart = -1 * (orfFrom - a.q_start * 3); e.contigEnd = -1 * (orfFrom - a.q_end * 3 - 2); e.strand = -1; }
    e.nucleotideLen = e.contigEnd - e.contigStart + 1;
    e.aaLen = e.nucleotideLen / 3;
    e.used = false;
    return e;
}

// isPairCompatible (collectoptimalset.cpp:33-74)
bool compatible(const Exon &f, const Exon &s, const mk_exon_params &P, size_t &aaOverlap) {
    if (f.strand != s.strand) return false;
    if (s.contigEnd < f.contigEnd) return false;
    const int diffOnContig = s.contigStart - f.contigEnd - 1;
    if (diffOnContig < 0) return false;
    const size_t d = (size_t) std::abs(diffOnContig);
    if (d < P.min_intron || d > P.max_intron) return false;
    const int diffAAs = s.targetStart - f.targetEnd - 1;
    aaOverlap = 0;
    if (diffAAs < 0) { aaOverlap = (size_t) std::abs(diffAAs); if (aaOverlap > P.max_aa_overlap) return false; }
    return s.targetStart >= f.targetStart;
}

// getPenaltyForProtCoords (:76-104)
int transition_penalty(const Exon &prev, const Exon &curr, const mk_exon_params &P) {
    const int diffAAs = curr.targetStart - prev.targetEnd - 1;
    if (diffAAs < 0) return P.gap_open + P.gap_extend * (std::abs(diffAAs) - 1);
    if (diffAAs <= 1) return 0;
    return P.gap_open + P.gap_extend * (diffAAs - 1);
}

// findoptimalsetbydp (:106-217)
int optimal_set(std::vector<Exon> &cand, std::vector<Exon> &set, const mk_exon_params &P) {
    set.clear();
    if (cand.empty()) return 0;
    std::stable_sort(cand.begin(), cand.end(), [](const Exon &a, const Exon &b) {        // PotentialExon::comparePotentialExons
        if (a.used != b.used) return a.used < b.used;
        if (a.contigStart != b.contigStart) return a.contigStart < b.contigStart;
        return a.contigEnd < b.contigEnd;
    });
    size_t n = cand.size();
    for (size_t i = 0; i < cand.size(); i++) if (cand[i].used) { n = i; break; }
    cand.resize(n);
    if (n == 0) return 0;
    const int targetLength = cand[0].targetLen;
    struct Row { size_t prev; int score; size_t numExons; int aaLen; };
    std::vector<Row> dp(n);
    for (size_t i = 0; i < n; i++) dp[i] = Row{i, cand[i].bitScore, 1, cand[i].aaLen};
    int best = 0;
    size_t last = 0;
    for (size_t c = 0; c < n; c++) {
        for (size_t p = 0; p < c; p++) {
            size_t overlap = 0;
            if (!compatible(cand[p], cand[c], P, overlap)) continue;
            const size_t ne = dp[p].numExons + 1;
            const int bonus = (int) std::log2((double) ne);
            const int s = dp[p].score + transition_penalty(cand[p], cand[c], P) + cand[c].bitScore + bonus;
            if (s > dp[c].score) dp[c] = Row{p, s, ne, dp[p].aaLen + cand[c].aaLen - (int) overlap};
        }
        if ((double) dp[c].aaLen / (double) targetLength >= P.target_cov_thr && dp[c].score > best) { last = c; best = dp[c].score; }
    }
    if (best == 0) return 0;
    size_t k = last;
    while (dp[k].prev != k) { set.push_back(cand[k]); cand[k].used = true; k = dp[k].prev; }
    set.push_back(cand[k]); cand[k].used = true;
    std::reverse(set.begin(), set.end());
    return best;
}

// what the reference reads back from the printed alignment columns (Matcher::resultToBuffer -> PotentialExon::setByAln)
void text_round_trip(const mk_alignment &a, double &seqId, double &evalue) {
    if (a.seq_id == 1.0) seqId = 1.0;                                       // printed "1.00"
    else seqId = (double) static_cast<int>(a.seq_id * 1000) / 1000.0;        // "0.xyz": strtod of a 3-digit decimal = correctly rounded xyz / 1000
    char buf[48];
    std::snprintf(buf, sizeof(buf), "%.3E", a.evalue);
    evalue = std::strtod(buf, nullptr);
}

struct ContigOut { std::vector<mk_prediction> preds; std::vector<mk_exon> exons; };

void emit(ContigOut &out, uint32_t target, int strand, int score, double evalue, const std::vector<Exon> &set, const mk_alignment *alns) {
    mk_prediction p;
    p.target = target; p.strand = strand; p.total_bit_score = (uint32_t) score; p.evalue = evalue; p.n_exons = (uint32_t) set.size();
    // Prediction::Prediction (PredictionParser.h:193-215)
    p.low_coord = (uint32_t) (set.front().strand == 1 ? set.front().contigStart : -1 * set.back().contigEnd);
    p.high_coord = (uint32_t) (set.front().strand == 1 ? set.back().contigEnd : -1 * set.front().contigStart);
    p.first_exon = out.exons.size();
    out.preds.push_back(p);
    for (const Exon &e : set) {
        mk_exon x;
        x.orf = e.orf; x.bit_score = e.bitScore;
        text_round_trip(alns[e.aln], x.seq_id, x.evalue);
        x.target_start = e.targetStart; x.target_end = e.targetEnd; x.target_len = e.targetLen;
        x.contig_start = e.contigStart; x.contig_end = e.contigEnd; x.nucleotide_len = e.nucleotideLen;
        x.orf_from = e.orfFrom; x.orf_to = e.orfTo;
        out.exons.push_back(x);
    }
}

}  // namespace

void default_exon_params(mk_exon_params &P) {                               // LocalParameters.h:138-146
    P.evalue_thr = (double) 0.001f; P.target_cov_thr = (double) 0.5f;
    P.max_intron = 10000; P.min_intron = 15; P.min_exon_aa = 11; P.max_aa_overlap = 10; P.max_exon_sets = 1;
    P.gap_open = -1; P.gap_extend = -1;
}

void predict_exons(const mk_orf *orfs, uint64_t nOrfs, uint32_t nContigs, const mk_alignment *alns, const uint64_t *alnOff, const uint32_t *targetKeys,
                   uint64_t dbResidues, const mk_exon_params &P, std::vector<mk_prediction> &preds, std::vector<uint64_t> &contigOff, std::vector<mk_exon> &exons) {
    // the fragments of a contig are consecutive (mk_extract_orfs writes them contig by contig)
    std::vector<uint64_t> firstOrf((size_t) nContigs + 1, nOrfs);
    {
        uint64_t k = 0;
        for (uint32_t c = 0; c <= nContigs; c++) {
            while (k < nOrfs && orfs[k].contig < c) k++;
            firstOrf[c] = k;
        }
    }
    std::vector<ContigOut> outs(nContigs);
    const double logDb = std::log2((double) dbResidues) + std::log2(2);
#pragma omp parallel
    {
        struct Item { uint32_t target, orf; uint64_t aln; };
        std::vector<Item> items;
        std::vector<Exon> plus, minus, setP, setM;
#pragma omp for schedule(dynamic, 8)
        for (uint32_t c = 0; c < nContigs; c++) {
            // resultspercontig.cpp:145-182: the contig's (orf -> target) records, ordered by (target, orf)
            items.clear();
            for (uint64_t k = firstOrf[c]; k < firstOrf[c + 1]; k++)
                for (uint64_t a = alnOff[k]; a < alnOff[k + 1]; a++) items.push_back(Item{targetKeys ? targetKeys[alns[a].db_key] : alns[a].db_key, (uint32_t) k, a});
            std::sort(items.begin(), items.end(), [](const Item &x, const Item &y) { return x.target != y.target ? x.target < y.target : x.orf < y.orf; });
            // collectoptimalset.cpp:262-413, target by target
            size_t i = 0;
            while (i < items.size()) {
                const uint32_t target = items[i].target;
                plus.clear(); minus.clear();
                for (; i < items.size() && items[i].target == target; i++) {
                    const mk_orf &o = orfs[items[i].orf];
                    const Exon e = make_exon(alns[items[i].aln], items[i].aln, target, items[i].orf, (int) o.from, (int) o.to);
                    if ((size_t) (std::abs(e.nucleotideLen) / 3) >= P.min_exon_aa) (e.strand == 1 ? plus : minus).push_back(e);
                }
                size_t iter = 0;
                while (iter < P.max_exon_sets && (!plus.empty() || !minus.empty())) {
                    const int scoreP = optimal_set(plus, setP, P), scoreM = optimal_set(minus, setM, P);
                    if (!setP.empty()) {
                        const double ev = std::pow(2, logDb - scoreP);
                        if (ev <= P.evalue_thr) emit(outs[c], target, 1, scoreP, ev, setP, alns);
                    }
                    if (!setM.empty()) {
                        const double ev = std::pow(2, logDb - scoreM);
                        if (ev <= P.evalue_thr) emit(outs[c], target, -1, scoreM, ev, setM, alns);
                    }
                    iter++;
                }
            }
        }
    }
    preds.clear(); exons.clear();
    contigOff.assign((size_t) nContigs + 1, 0);
    for (uint32_t c = 0; c < nContigs; c++) {
        for (mk_prediction p : outs[c].preds) { p.first_exon += exons.size(); preds.push_back(p); }
        exons.insert(exons.end(), outs[c].exons.begin(), outs[c].exons.end());
        contigOff[c + 1] = preds.size();
    }
}

// one line of the prediction DB: Prediction::predictionToBuffer + PotentialExon::exonToBuffer (PredictionParser.h:88-137,357-384)
size_t format_prediction_exon(char *buf, const mk_prediction &p, const mk_exon &e) {
    char *w = buf;
    w += std::sprintf(w, "%u\t%d\t%u\t%.3E\t%u\t%u\t%u\t%u\t%d\t", p.target, p.strand, p.total_bit_score, p.evalue, p.n_exons, p.low_coord, p.high_coord,
                      e.orf, e.bit_score);
    const float f = (float) e.seq_id;
    if (f == 1.0) w += std::sprintf(w, "1.000\t");
    else {
        *w++ = '0'; *w++ = '.';
        if (f < 0.10) *w++ = '0';
        if (f < 0.01) *w++ = '0';
        w += std::sprintf(w, "%d\t", (int) (f * 1000));
    }
    w += std::sprintf(w, "%.3E\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", e.evalue, e.target_start, e.target_end, e.target_len, e.contig_start, e.contig_end,
                      e.nucleotide_len, e.orf_from, e.orf_to);
    return (size_t) (w - buf);
}

}  // namespace mk
