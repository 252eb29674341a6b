// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A small driver (our code) that links the REFERENCE's own translation units
// (compiled in place from /root/reference by oracle/Makefile.ref into
// oracle/_ref/) and runs the prefilter+align hot path through the
// reference's classes exactly as its own drivers do:
//   * prefilter loop  = lib/mmseqs/src/prefiltering/Prefiltering.cpp:60-72,
//     192-218,514-553 (matrices, k-mer threshold, index build) and :790-887
//     (per-query QueryMatcher::matchQuery + hit formatting);
//   * align loop      = lib/mmseqs/src/alignment/Alignment.cpp:152-154,263,
//     279-514 (Matcher::initQuery / getSWResult / checkCriteria / compareHits
//     / resultToBuffer).
// The reference's Parameters singleton, workflows and DB writers are NOT
// built (they need cmake-generated headers); the loops above are restated
// here with the reference's defaults for `metaeuk predictexons`
// (SURVEY.md section 3.2 argv).  Everything arithmetic is the reference's
// compiled code.
//
// Modes:
//   ref_harness pipeline <matdir> <targets.txt> <queries.txt> <outdir> [-s 5.7] [-k 7] [--threads N] [--dump] [--index <indexDB>]
//     --index: the target side comes from a precomputed index DB through the reference's PrefilteringIndexReader (sequence DB,
//     SequenceLookup, IndexTable, score matrices, seed matrix), the way Prefiltering.cpp:84-160,530-545 uses it; targets.txt is ignored
//     --split N: TARGET_DB_SPLIT (Prefiltering.cpp:352-362,725-750,957-1003): N residue-balanced target ranges, each with its own
//     index, k from the residues of the range, --max-seqs reduced to max/N + 4 sqrt(max/N), the N prefilter DBs joined by the
//     reference's own Prefiltering::mergeTargetSplits (Prefiltering.cpp:379-496, compiled in place)
//   ref_harness profilesearch <matdir> <profile DB data file> <its .index> <fragments.txt> <outdir> [-s 4] [-e 100] [--eval-abs X]
//                [--keys keys.txt] [--threads N] [-k 7]
//     = the sliced target-profile search of `predictexons contigsDB profileDB` (searchslicedtargetprofile.sh; Search.cpp:357-399) for one
//       slice: prefilter with the PROFILES as queries (Sequence::mapProfile, profile k-mer lists, QueryMatcher with setProfileMatrix)
//       against the fragments, align (Matcher with a profile query), swapresults (Matcher::result_t::swapResult + compareHits).
//       Same arguments and output files as `mko_cli profilesearch`: pref.txt / aln.txt ('>profile key' blocks in key order) and
//       swapped.txt ('>fragment key' blocks, every fragment).  The loops are restated from Prefiltering.cpp:790-887, Alignment.cpp:
//       279-514 and util/swapresults.cpp:254-318; every class they drive is the reference's compiled code.
//   ref_harness createindex <matdir> <seqDB> [-s 5.7] [-k 7]
//     = indexdb (util/indexdb.cpp:67-186): PrefilteringIndexReader::createIndexFile over the sequence DB on disk -> <seqDB>.idx
//   ref_harness sw       <matdir> <targets.txt> <queries.txt> <pairs.txt> <out.txt> [--dbres N]
//   ref_harness submat   <matfile.out> <bitFactor> <bias>
//   ref_harness exons    <targets.txt> <contigs.txt> <orfs.txt> <aln.txt> <out.txt>
//     = resultspercontig + collectoptimalset on the outputs of the modes below/above: orfs.txt as
//       written by `orfs`, aln.txt as written by `pipeline` (one block per ORF fragment, same order).  The joining of the
//       two inputs per contig and the target-by-target loop are our restatement of resultspercontig.cpp:145-190 and
//       collectoptimalset.cpp:262-413; PotentialExon::setByAln, findoptimalsetbydp and Prediction::predictionToBuffer are
//       the reference's own code (src/exonpredictor/collectoptimalset.cpp, src/commons/PredictionParser.h).
//   ref_harness orfs     <contigs.txt> <out.txt> [--min-length 15]
//     = extractorfs --translate as `predictexons` runs it (util/extractorfs.cpp:19-159 loop, Orf::findAll with
//       orf-start-mode 1, both strands, all frames, translation table 1): per contig one ">key" line, then one line per
//       ORF fragment "header<TAB>protein" in the order extractorfs writes them (= the renumbered ORF ids)
// Sequence files: one amino-acid sequence per line; key = 0-based line number.
#include "SubstitutionMatrix.h"
#include "ExtendedSubstitutionMatrix.h"
#include "IndexTable.h"
#include "IndexBuilder.h"
#include "QueryMatcher.h"
#include "Matcher.h"
#include "EvalueComputation.h"
#include "Sequence.h"
#include "DBReader.h"
#include "Parameters.h"
#include "Util.h"
#include "FastSort.h"
#include "Orf.h"
#include "TranslateNucl.h"
#include "PredictionParser.h"
#include "PrefilteringIndexReader.h"
#include "Prefiltering.h"
#include "DBWriter.h"

#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cmath>
#include <string>
#include <vector>
#include <fstream>
#include <chrono>
#include <omp.h>
#include <sys/stat.h>

// Debug.h declares this global; the reference defines it in Application.cpp
// (which pulls the whole CLI).  It is only the program name for messages.
const char *binary_name = "ref_harness";
// the application's other two globals (mmseqs.cpp:12-13, metaeuk.cpp:14): the generator string written into an index DB and the
// index version it accepts (MMSEQS_CURRENT_INDEX_VERSION, MMseqsBase.cpp:6)
const char *version = "ref_harness";
const char *index_version_compatible = "16";

static std::vector<std::string> readLines(const char *path) {
    std::vector<std::string> v;
    std::ifstream in(path);
    if (!in) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    std::string line;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
        v.push_back(line);
    }
    return v;
}

// Write a sequence DB in the MMseqs2 on-disk format
// (DBWriter.cpp:401-428 index lines, :193-213 dbtype; entries are "SEQ\n\0").
static void writeSeqDb(const std::string &base, const std::vector<std::string> &seqs, const std::vector<unsigned int> *keys = NULL) {
    FILE *d = fopen(base.c_str(), "wb");
    FILE *i = fopen((base + ".index").c_str(), "wb");
    size_t off = 0;
    for (size_t k = 0; k < seqs.size(); k++) {
        fwrite(seqs[k].data(), 1, seqs[k].size(), d);
        fputc('\n', d); fputc('\0', d);
        fprintf(i, "%zu\t%zu\t%zu\n", keys ? (size_t) (*keys)[k] : k, off, seqs[k].size() + 2);
        off += seqs[k].size() + 2;
    }
    fclose(d); fclose(i);
    FILE *t = fopen((base + ".dbtype").c_str(), "wb");
    int dbtype = Parameters::DBTYPE_AMINO_ACIDS;
    fwrite(&dbtype, 4, 1, t);
    fclose(t);
}

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int kmerThreshold(float sensitivity, int kmerSize) {
    // Prefiltering.cpp:1048-1060 (sequence-sequence branch)
    float kmerThrBest = FLT_MAX;
    if (kmerSize == 5) { float base = 160.75; kmerThrBest = base - (sensitivity * 12.75); }
    else if (kmerSize == 6) { float base = 163.2; kmerThrBest = base - (sensitivity * 8.917); }
    else if (kmerSize == 7) { float base = 186.15; kmerThrBest = base - (sensitivity * 11.22); }
    return static_cast<int>(kmerThrBest);
}

static int cmdSubmat(int argc, char **argv) {
    SubstitutionMatrix m(argv[2], atof(argv[3]), atof(argv[4]));
    for (int i = 0; i < m.alphabetSize; i++) {
        for (int j = 0; j < m.alphabetSize; j++) printf("%d%c", m.subMatrix[i][j], j + 1 == m.alphabetSize ? '\n' : ' ');
    }
    for (int i = 0; i < m.alphabetSize; i++) printf("%.17g%c", m.pBack[i], i + 1 == m.alphabetSize ? '\n' : ' ');
    for (int i = 0; i < m.alphabetSize; i++) {
        for (int j = 0; j < m.alphabetSize; j++) printf("%.17g%c", m.probMatrix[i][j], j + 1 == m.alphabetSize ? '\n' : ' ');
    }
    return 0;
}

static int cmdPipeline(int argc, char **argv) {
    if (argc < 6) return 2;
    std::string matdir = argv[2];
    std::vector<std::string> targets = readLines(argv[3]);
    std::vector<std::string> queries = readLines(argv[4]);
    std::string outdir = argv[5];
    float sensitivity = 5.7f;
    int threads = 1;
    bool dump = false;
    bool doAlign = true;
    size_t maxResListLen = 300;
    std::string indexDb;
    int forcedK = 0;
    int splits = 1;
    for (int a = 6; a < argc; a++) {
        std::string s = argv[a];
        if (s == "-s") sensitivity = atof(argv[++a]);
        else if (s == "-k") forcedK = atoi(argv[++a]);
        else if (s == "--split") splits = std::max(1, atoi(argv[++a]));
        else if (s == "--threads") threads = atoi(argv[++a]);
        else if (s == "--dump") dump = true;
        else if (s == "--no-align") doAlign = false;
        else if (s == "--max-seqs") maxResListLen = atol(argv[++a]);
        else if (s == "--index") indexDb = argv[++a];
    }
    mkdir(outdir.c_str(), 0755);
    omp_set_num_threads(threads);
    std::string tdb = outdir + "/_tdb", qdb = outdir + "/_qdb";
    writeSeqDb(tdb, targets);
    writeSeqDb(qdb, queries);

    DBReader<unsigned int> *tidxdbr = NULL, *tdbrp = NULL;
    if (!indexDb.empty()) {                                 // Prefiltering.cpp:84-96
        tidxdbr = new DBReader<unsigned int>(indexDb.c_str(), (indexDb + ".index").c_str(), threads, DBReader<unsigned int>::USE_INDEX | DBReader<unsigned int>::USE_DATA);
        tidxdbr->open(DBReader<unsigned int>::NOSORT);
        if (!PrefilteringIndexReader::checkIfIndexFile(tidxdbr)) { fprintf(stderr, "Outdated index version\n"); return 3; }
        tdbrp = PrefilteringIndexReader::openNewReader(tidxdbr, PrefilteringIndexReader::DBR1DATA, PrefilteringIndexReader::DBR1INDEX, true, threads, false, false);
    } else {
        tdbrp = new DBReader<unsigned int>(tdb.c_str(), (tdb + ".index").c_str(), threads, DBReader<unsigned int>::USE_INDEX | DBReader<unsigned int>::USE_DATA);
        tdbrp->open(DBReader<unsigned int>::LINEAR_ACCCESS);
    }
    DBReader<unsigned int> &tdbr = *tdbrp;
    DBReader<unsigned int> qdbr(qdb.c_str(), (qdb + ".index").c_str(), threads, DBReader<unsigned int>::USE_INDEX | DBReader<unsigned int>::USE_DATA);
    qdbr.open(DBReader<unsigned int>::LINEAR_ACCCESS);

    const int querySeqType = Parameters::DBTYPE_AMINO_ACIDS, targetSeqType = Parameters::DBTYPE_AMINO_ACIDS;
    const size_t maxSeqLen = 65535;
    // Prefiltering.cpp:68-70
    std::string blosum = matdir + "/blosum62.out", vtml = matdir + "/VTML80.out";
    // with an index: the seed matrix it carries ("VTML80.out:<text>", Prefiltering.cpp:151 + getSubstitutionMatrix)
    if (tidxdbr) vtml = PrefilteringIndexReader::getSubstitutionMatrix(tidxdbr);
    BaseMatrix *kmerSubMat = new SubstitutionMatrix(vtml.c_str(), 8.0, -0.2f);
    BaseMatrix *ungappedSubMat = new SubstitutionMatrix(blosum.c_str(), 2.0, -0.2f);
    const int alphabetSize = kmerSubMat->alphabetSize;
    // Prefiltering::setupSplit (Prefiltering.cpp:352-362): k from the residues per split, --max-seqs reduced for TARGET_DB_SPLIT
    int kmerSize = forcedK ? forcedK : IndexTable::computeKmerSize(tdbr.getAminoAcidDBSize() / std::max(splits, 1));      // -k (Prefiltering.cpp:98-99,181)
    if (tidxdbr) kmerSize = PrefilteringIndexReader::getMetadata(tidxdbr).kmerSize;
    const int kmerThr = kmerThreshold(sensitivity, kmerSize);
    maxResListLen = std::min(tdbr.getSize(), maxResListLen);
    if (splits > 1) {
        if (tidxdbr) { fprintf(stderr, "--split with --index is not part of this harness\n"); return 2; }
        size_t fourTimesStdDeviation = 4 * sqrt(static_cast<double>(maxResListLen) / static_cast<double>(splits));
        maxResListLen = std::max(static_cast<size_t>(1), (maxResListLen / splits) + fourTimesStdDeviation);
    }
    double t0 = now();
    // Prefiltering.cpp:208-213
    ScoreMatrix _2mer, _3mer;
    if (tidxdbr) {                                          // Prefiltering.cpp:197-206
        _2mer = PrefilteringIndexReader::get2MerScoreMatrix(tidxdbr, Parameters::PRELOAD_MODE_MMAP);
        _3mer = PrefilteringIndexReader::get3MerScoreMatrix(tidxdbr, Parameters::PRELOAD_MODE_MMAP);
    } else {
        kmerSubMat->alphabetSize = kmerSubMat->alphabetSize - 1;
        _2mer = ExtendedSubstitutionMatrix::calcScoreMatrix(*kmerSubMat, 2);
        _3mer = ExtendedSubstitutionMatrix::calcScoreMatrix(*kmerSubMat, 3);
        kmerSubMat->alphabetSize = alphabetSize;
    }
    double tExt = now() - t0;
    const size_t nq = qdbr.getSize();
    std::vector<std::string> prefOut(nq), alnOut(nq);
    std::vector<std::string> statOut(nq);
    size_t totalHits = 0;
    double kmersPerPos = 0; size_t dbMatches = 0;
    double tIndex = 0, tPrefAcc = 0;
    std::vector<std::pair<std::string, std::string>> splitFiles;
    for (int split = 0; split < splits; split++) {
    // Prefiltering::runSplit, TARGET_DB_SPLIT (Prefiltering.cpp:733-750): the split's target range
    size_t dbFrom = 0, dbSize = tdbr.getSize();
    if (splits > 1) {
        tdbr.decomposeDomainByAminoAcid(split, splits, &dbFrom, &dbSize);
        if (dbSize == 0) continue;
    }
    // Prefiltering.cpp:514-553
    t0 = now();
    SequenceLookup *sequenceLookup = NULL;
    IndexTable *indexTable = NULL;
    if (tidxdbr) {                                          // Prefiltering.cpp:530-545
        indexTable = PrefilteringIndexReader::getIndexTable(0, tidxdbr, Parameters::PRELOAD_MODE_MMAP);
        sequenceLookup = PrefilteringIndexReader::getSequenceLookup(0, tidxdbr, Parameters::PRELOAD_MODE_MMAP);
    } else {
        Sequence tseq(maxSeqLen, targetSeqType, kmerSubMat, kmerSize, true, true, true, "");
        indexTable = new IndexTable(alphabetSize - 1, kmerSize, false);
        IndexBuilder::fillDatabase(indexTable, &sequenceLookup, *kmerSubMat, _3mer, _2mer, &tseq, &tdbr, dbFrom, dbFrom + dbSize,
                                   kmerThr, true /*mask*/, false /*maskLowerCase*/, 0.9f /*maskProb*/, 0 /*maskNrepeats*/, 0 /*targetSearchMode*/);
    }
    tIndex += now() - t0;

    if (dump) {
        FILE *f = fopen((outdir + "/masked_targets.txt").c_str(), "w");
        for (size_t id = 0; id < dbSize; id++) {
            std::pair<const unsigned char *, const unsigned int> s = sequenceLookup->getSequence(id);
            for (unsigned int p = 0; p < s.second; p++) fputc(kmerSubMat->num2aa[s.first[p]], f);
            fputc('\n', f);
        }
        fclose(f);
        f = fopen((outdir + "/index.txt").c_str(), "w");
        for (size_t k = 0; k < indexTable->getTableSize(); k++) {
            size_t n;
            IndexEntryLocal *e = indexTable->getDBSeqList(k, &n);
            if (n == 0) continue;
            fprintf(f, "%zu", k);
            for (size_t j = 0; j < n; j++) fprintf(f, " %u:%u", e[j].seqId, (unsigned) e[j].position_j);
            fputc('\n', f);
        }
        fclose(f);
    }

    std::vector<std::string> splitOut(splits > 1 ? nq : 0);
    t0 = now();
#pragma omp parallel num_threads(threads)
    {
        unsigned int thread_idx = (unsigned int) omp_get_thread_num();
        Sequence seq(qdbr.getMaxSeqLen(), querySeqType, kmerSubMat, kmerSize, true, true, true, "");
        QueryMatcher matcher(indexTable, sequenceLookup, kmerSubMat, ungappedSubMat, kmerThr, kmerSize, dbSize,
                             std::max(tdbr.getMaxSeqLen(), qdbr.getMaxSeqLen()), maxResListLen, true, 1.0f,
                             true, 15, false, false);
        matcher.setSubstitutionMatrix(&_3mer, &_2mer);
        char buffer[128];
#pragma omp for schedule(dynamic, 1) reduction(+: totalHits, kmersPerPos, dbMatches)
        for (size_t id = 0; id < nq; id++) {
            char *seqData = qdbr.getData(id, thread_idx);
            unsigned int qKey = qdbr.getDbKey(id);
            seq.mapSequence(id, qKey, seqData, qdbr.getSeqLen(id));
            std::pair<hit_t *, size_t> res = matcher.matchQuery(&seq, UINT_MAX, false);
            std::string &out = splits > 1 ? splitOut[id] : prefOut[id];
            for (size_t i = 0; i < res.second; i++) {
                hit_t *h = res.first + i;
                h->seqId = tdbr.getDbKey(h->seqId + dbFrom);
                int len = QueryMatcher::prefilterHitToBuffer(buffer, *h);
                out.append(buffer, len);
            }
            totalHits += res.second;
            kmersPerPos += matcher.getStatistics()->kmersPerPos;
            dbMatches += matcher.getStatistics()->dbMatches;
            if (dump) {
                char tmp[128];
                snprintf(tmp, sizeof(tmp), "%lld\t%zu\n", llround(matcher.getStatistics()->kmersPerPos * seq.L), matcher.getStatistics()->dbMatches);
                statOut[id] = tmp;
            }
        }
    }
    tPrefAcc += now() - t0;
    if (splits > 1) {
        // the split's prefilter DB, entries in id order (what runSplit leaves after sortDatafileByIdOrder, Prefiltering.cpp:920-937)
        const std::string base = outdir + "/_pref_split_" + std::to_string(split);
        FILE *d = fopen(base.c_str(), "wb"), *ix = fopen((base + ".index").c_str(), "wb");
        size_t off = 0;
        for (size_t id = 0; id < nq; id++) {
            fwrite(splitOut[id].data(), 1, splitOut[id].size(), d);
            fputc('\0', d);
            fprintf(ix, "%u\t%zu\t%zu\n", qdbr.getDbKey(id), off, splitOut[id].size() + 1);
            off += splitOut[id].size() + 1;
        }
        fclose(d); fclose(ix);
        FILE *t = fopen((base + ".dbtype").c_str(), "wb");
        int dbtype = Parameters::DBTYPE_PREFILTER_RES;
        fwrite(&dbtype, 4, 1, t);
        fclose(t);
        splitFiles.push_back(std::make_pair(base, base + ".index"));
        delete indexTable;
        delete sequenceLookup;
    }
    }   // splits
    t0 = now();
    if (splits > 1) {
        // the reference's own merge (Prefiltering.cpp:379-496), then the merged DB back into memory
        const std::string merged = outdir + "/_pref_merged";
        Prefiltering::mergeTargetSplits(merged, merged + ".index", splitFiles, threads);
        DBReader<unsigned int> mdbr(merged.c_str(), (merged + ".index").c_str(), 1, DBReader<unsigned int>::USE_INDEX | DBReader<unsigned int>::USE_DATA);
        mdbr.open(DBReader<unsigned int>::NOSORT);
        totalHits = 0;
        for (size_t id = 0; id < nq; id++) {
            const size_t mid = mdbr.getId(qdbr.getDbKey(id));
            prefOut[id] = mid == UINT_MAX ? std::string() : std::string(mdbr.getData(mid, 0));
            for (char ch : prefOut[id]) totalHits += ch == '\n';
        }
        mdbr.close();
    }
    tPrefAcc += now() - t0;
    const double tPref = tPrefAcc;

    // ---------------- align (Alignment.cpp) ----------------
    size_t alignmentsNum = 0, totalPassed = 0;
    double cellsFwd = 0;
    double tAln = 0;
    if (doAlign) {
        BaseMatrix *m = new SubstitutionMatrix(blosum.c_str(), 2.0, 0.0);
        const int gapOpen = 11, gapExtend = 1;
        EvalueComputation evaluer(tdbr.getAminoAcidDBSize(), m, gapOpen, gapExtend);
        const double evalThr = 100.0;
        const float covThr = 0.0f; const int covMode = 0; const int seqIdMode = 0;
        const int alnLenThr = 11; const double seqIdThr = 0.0;
        const unsigned int swMode = Matcher::SCORE_COV;
        t0 = now();
#pragma omp parallel num_threads(threads)
        {
            unsigned int thread_idx = (unsigned int) omp_get_thread_num();
            char buffer[1024 + 32768 * 4];
            Sequence qSeq(maxSeqLen, querySeqType, m, 0, false, true);
            Sequence dbSeq(maxSeqLen, targetSeqType, m, 0, false, true);
            Matcher matcher(querySeqType, targetSeqType, std::max(tdbr.getMaxSeqLen(), qdbr.getMaxSeqLen()), m, &evaluer, true, 1.0f, gapOpen, gapExtend, 0.0f, 40);
            std::vector<Matcher::result_t> swResults;
#pragma omp for schedule(dynamic, 5) reduction(+: alignmentsNum, totalPassed, cellsFwd)
            for (size_t id = 0; id < nq; id++) {
                std::string &pref = prefOut[id];
                char *data = (char *) pref.c_str();
                unsigned int queryDbKey = qdbr.getDbKey(id);
                if (*data != '\0') {
                    size_t qId = qdbr.getId(queryDbKey);
                    qSeq.mapSequence(qId, queryDbKey, qdbr.getData(qId, thread_idx), qdbr.getSeqLen(qId));
                    matcher.initQuery(&qSeq);
                }
                while (*data != '\0') {
                    hit_t hit = QueryMatcher::parsePrefilterHit(data);
                    const unsigned int dbKey = hit.seqId;
                    short diagonal = static_cast<short>(hit.diagonal);
                    data = Util::skipLine(data);
                    size_t dbId = tdbr.getId(dbKey);
                    dbSeq.mapSequence(dbId, dbKey, tdbr.getData(dbId, thread_idx), tdbr.getSeqLen(dbId));
                    Matcher::result_t res = matcher.getSWResult(&dbSeq, static_cast<int>(diagonal), false, covMode, covThr, evalThr, swMode, seqIdMode, false, false);
                    alignmentsNum++;
                    cellsFwd += (double) qSeq.L * (double) dbSeq.L;
                    const bool evalOk = (res.eval <= evalThr);
                    const bool seqIdOK = (res.seqId >= seqIdThr);
                    const bool covOK = Util::hasCoverage(covThr, covMode, res.qcov, res.dbcov);
                    const bool alnLenOK = Util::hasAlignmentLength(alnLenThr, res.alnLength);
                    if (evalOk && seqIdOK && covOK && alnLenOK) {
                        swResults.emplace_back(res);
                        totalPassed++;
                    }
                }
                if (swResults.size() > 1) {
                    SORT_SERIAL(swResults.begin(), swResults.end(), Matcher::compareHits);
                }
                std::string &out = alnOut[id];
                for (size_t r = 0; r < swResults.size(); r++) {
                    size_t len = Matcher::resultToBuffer(buffer, swResults[r], false);
                    out.append(buffer, len);
                }
                swResults.clear();
            }
        }
        tAln = now() - t0;
    }

    {
        FILE *f = fopen((outdir + "/pref.txt").c_str(), "w");
        for (size_t id = 0; id < nq; id++) { fprintf(f, ">%u\n", qdbr.getDbKey(id)); fputs(prefOut[id].c_str(), f); }
        fclose(f);
        if (doAlign) {
            f = fopen((outdir + "/aln.txt").c_str(), "w");
            for (size_t id = 0; id < nq; id++) { fprintf(f, ">%u\n", qdbr.getDbKey(id)); fputs(alnOut[id].c_str(), f); }
            fclose(f);
        }
        if (dump) {
            f = fopen((outdir + "/stats.txt").c_str(), "w");
            for (size_t id = 0; id < nq; id++) fputs(statOut[id].c_str(), f);
            fclose(f);
        }
    }
    size_t qres = 0; for (size_t i = 0; i < queries.size(); i++) qres += queries[i].size();
    printf("{\"queries\": %zu, \"targets\": %zu, \"query_residues\": %zu, \"target_residues\": %zu, \"k\": %d, \"kmer_thr\": %d, "
           "\"threads\": %d, \"t_extmat\": %.4f, \"t_index\": %.4f, \"t_prefilter\": %.4f, \"t_align\": %.4f, "
           "\"pref_hits\": %zu, \"kmers_per_pos\": %.4f, \"db_matches\": %zu, \"alignments\": %zu, \"passed\": %zu, \"cells_fwd\": %.0f}\n",
           queries.size(), targets.size(), qres, (size_t) tdbr.getAminoAcidDBSize(), kmerSize, kmerThr, threads, tExt, tIndex, tPref, tAln,
           totalHits, kmersPerPos / (double) nq, dbMatches, alignmentsNum, totalPassed, cellsFwd);
    return 0;
}

// indexdb (util/indexdb.cpp:67-186) for an amino-acid sequence DB with predictexons' target-side settings
static int cmdCreateIndex(int argc, char **argv) {
    if (argc < 4) return 2;
    std::string matdir = argv[2], db = argv[3];
    float sensitivity = 5.7f;
    int forcedK = 0;
    for (int a = 4; a < argc; a++) {
        if (std::string(argv[a]) == "-s") sensitivity = atof(argv[++a]);
        else if (std::string(argv[a]) == "-k") forcedK = atoi(argv[++a]);
    }
    DBReader<unsigned int> dbr(db.c_str(), (db + ".index").c_str(), 1, DBReader<unsigned int>::USE_INDEX | DBReader<unsigned int>::USE_DATA);
    dbr.open(DBReader<unsigned int>::NOSORT);
    std::string vtml = matdir + "/VTML80.out";
    BaseMatrix *seedSubMat = new SubstitutionMatrix(vtml.c_str(), 8.0, -0.2f);
    const int kmerSize = forcedK ? forcedK : IndexTable::computeKmerSize(dbr.getAminoAcidDBSize());     // indexdb.cpp:71-75
    const int kmerThr = kmerThreshold(sensitivity, kmerSize);
    PrefilteringIndexReader::createIndexFile(PrefilteringIndexReader::indexName(db), &dbr, NULL, NULL, NULL, NULL, seedSubMat, 65535,
                                             true, "", true, seedSubMat->alphabetSize, kmerSize, 1 /*maskMode*/, 0 /*maskLowerCase*/, 0.9f,
                                             0 /*maskNrepeats*/, kmerThr, 0 /*targetSearchMode*/, 1 /*splits*/, 0 /*indexSubset*/);
    printf("{\"index\": \"%s\", \"k\": %d, \"kmer_thr\": %d}\n", PrefilteringIndexReader::indexName(db).c_str(), kmerSize, kmerThr);
    return 0;
}

// Function-level SW goldens: every listed pair goes through
// Matcher::getSWResult (SCORE_COV mode) with an e-value threshold that never
// rejects, so the reverse pass always runs.
static int cmdSw(int argc, char **argv) {
    if (argc < 7) return 2;
    std::string matdir = argv[2];
    std::vector<std::string> targets = readLines(argv[3]);
    std::vector<std::string> queries = readLines(argv[4]);
    std::vector<std::string> pairs = readLines(argv[5]);
    size_t dbRes = 0;
    for (size_t i = 0; i < targets.size(); i++) dbRes += targets[i].size();
    for (int a = 7; a < argc; a++) if (std::string(argv[a]) == "--dbres") dbRes = atol(argv[++a]);
    std::string blosum = matdir + "/blosum62.out";
    BaseMatrix *m = new SubstitutionMatrix(blosum.c_str(), 2.0, 0.0);
    EvalueComputation evaluer(dbRes, m, 11, 1);
    size_t maxLen = 1;
    for (size_t i = 0; i < targets.size(); i++) maxLen = std::max(maxLen, targets[i].size());
    for (size_t i = 0; i < queries.size(); i++) maxLen = std::max(maxLen, queries[i].size());
    Sequence qSeq(maxLen + 1, Parameters::DBTYPE_AMINO_ACIDS, m, 0, false, true);
    Sequence dbSeq(maxLen + 1, Parameters::DBTYPE_AMINO_ACIDS, m, 0, false, true);
    Matcher matcher(Parameters::DBTYPE_AMINO_ACIDS, Parameters::DBTYPE_AMINO_ACIDS, maxLen + 1, m, &evaluer, true, 1.0f, 11, 1, 0.0f, 40);
    FILE *f = fopen(argv[6], "w");
    char buffer[4096];
    int lastQ = -1;
    for (size_t p = 0; p < pairs.size(); p++) {
        int q, t;
        if (sscanf(pairs[p].c_str(), "%d %d", &q, &t) != 2) continue;
        if (q != lastQ) {
            qSeq.mapSequence(q, q, queries[q].c_str(), queries[q].size());
            matcher.initQuery(&qSeq);
            lastQ = q;
        }
        dbSeq.mapSequence(t, t, targets[t].c_str(), targets[t].size());
        Matcher::result_t res = matcher.getSWResult(&dbSeq, 0, false, 0, 0.0f, DBL_MAX, Matcher::SCORE_COV, 0, false, false);
        size_t len = Matcher::resultToBuffer(buffer, res, false);
        fprintf(f, "%d\t%d\t", q, t);
        fwrite(buffer, 1, len, f);
    }
    fclose(f);
    return 0;
}

// extractorfs.cpp:64-125 for one contig after another (contig/orf start and end modes 2 = keep everything)
static int cmdOrfs(int argc, char **argv) {
    if (argc < 4) return 2;
    size_t minLength = 15;                                       // PredictExons.cpp:11
    for (int i = 4; i + 1 < argc; i++) if (!strcmp(argv[i], "--min-length")) minLength = (size_t) atol(argv[i + 1]);
    std::vector<std::string> contigs = readLines(argv[2]);
    FILE *out = fopen(argv[3], "w");
    if (!out) return 1;
    const size_t maxSeqLen = 65535;                              // Parameters.cpp default
    Orf orf(1, false);
    TranslateNucl translateNucl(static_cast<TranslateNucl::GenCode>(1));
    const unsigned int forwardFrames = Orf::getFrames("1,2,3"), reverseFrames = Orf::getFrames("1,2,3");
    std::vector<Orf::SequenceLocation> res;
    std::vector<char> aa(maxSeqLen + 3 + 1);
    char buffer[1024];
    size_t nOrf = 0;
    for (size_t key = 0; key < contigs.size(); key++) {
        fprintf(out, ">%zu\n", key);
        const std::string data = contigs[key] + "\n";            // DB entries end with a newline
        const size_t sequenceLength = contigs[key].size();
        if (!orf.setSequence(data.c_str(), sequenceLength)) continue;
        res.clear();
        orf.findAll(res, minLength, 32734, INT_MAX, forwardFrames, reverseFrames, 1);
        for (size_t k = 0; k < res.size(); k++) {
            Orf::SequenceLocation loc = res[k];
            std::pair<const char *, size_t> sequence = orf.getSequence(loc);
            size_t fromPos = loc.from, toPos = loc.to;
            if (loc.strand == Orf::STRAND_MINUS) { fromPos = (sequenceLength - 1) - loc.from; toPos = (sequenceLength - 1) - loc.to; }
            Orf::writeOrfHeader(buffer, (unsigned int) key, fromPos, toPos, loc.hasIncompleteStart, loc.hasIncompleteEnd);
            if ((data[sequence.second] != '\n' && sequence.second % 3 != 0) && (data[sequence.second - 1] == '\n' && (sequence.second - 1) % 3 != 0))
                sequence.second = sequence.second - (sequence.second % 3);
            if (sequence.second < 3) continue;
            if (sequence.second > (3 * maxSeqLen)) sequence.second = (3 * maxSeqLen);
            translateNucl.translate(aa.data(), sequence.first, sequence.second);
            buffer[strlen(buffer) - 1] = '\0';                   // drop the header's newline
            fprintf(out, "%s\t%.*s\n", buffer, (int) (sequence.second / 3), aa.data());
            nOrf++;
        }
    }
    fclose(out);
    printf("{\"contigs\": %zu, \"orfs\": %zu}\n", contigs.size(), nOrf);
    return 0;
}

// the reference's exon-chaining DP (src/exonpredictor/collectoptimalset.cpp:106-217), linked from its own object file
int findoptimalsetbydp(std::vector<PotentialExon> &potentialExonCandidates, std::vector<PotentialExon> &optimalExonSet,
                       const size_t minIntronLength, const size_t maxIntronLength, const size_t maxAaOvelap, const int setGapOpenPenalty,
                       const int setGapExtendPenalty, const double dMetaeukTargetCovThr);

static int cmdExons(int argc, char **argv) {
    if (argc < 7) return 2;
    std::vector<std::string> targets = readLines(argv[2]), contigs = readLines(argv[3]);
    size_t totNumOfAAsInTargetDb = 0;                              // DBReader::getAminoAcidDBSize
    for (size_t i = 0; i < targets.size(); i++) totNumOfAAsInTargetDb += targets[i].size();
    // LocalParameters defaults (src/commons/LocalParameters.h:138-146)
    const float metaeukEvalueThr = 0.001, metaeukTargetCovThr = 0.5;
    const size_t maxIntronLength = 10000, minIntronLength = 15, minExonAaLength = 11, maxAaOverlap = 10, maxExonSets = 1;
    const int setGapOpenPenalty = -1, setGapExtendPenalty = -1;
    const double dMetaeukEvalueThr = (double) metaeukEvalueThr, dMetaeukTargetCovThr = (double) metaeukTargetCovThr;
    FILE *forf = fopen(argv[4], "r"), *faln = fopen(argv[5], "r"), *out = fopen(argv[6], "w");
    if (!forf || !faln || !out) return 1;
    // ORF fragments: key = running index; per contig the list of (orfKey, header fields)
    struct OrfRec { unsigned int key, contig; Orf::SequenceLocation loc; };
    std::vector<std::vector<OrfRec>> orfsOfContig(contigs.size());
    {
        char *line = NULL; size_t cap = 0; ssize_t r;
        unsigned int key = 0;
        while ((r = getline(&line, &cap, forf)) >= 0) {
            if (line[0] == '>') continue;
            std::string hdr(line);
            const size_t lastTab = hdr.rfind('\t');                // drop the protein column
            hdr = hdr.substr(0, lastTab) + "\n";
            OrfRec o;
            o.key = key++;
            o.loc = Orf::parseOrfHeader(hdr.c_str());
            o.contig = o.loc.id;
            orfsOfContig[o.contig].push_back(o);
        }
        free(line);
    }
    // alignments: block k (">k") = ORF k, lines = Matcher::resultToBuffer
    std::vector<std::vector<std::string>> alnOfOrf;
    {
        char *line = NULL; size_t cap = 0; ssize_t r;
        while ((r = getline(&line, &cap, faln)) >= 0) {
            if (line[0] == '>') { alnOfOrf.emplace_back(); continue; }
            alnOfOrf.back().emplace_back(line);
        }
        free(line);
    }
    char buffer[65536], exonLineBuffer[2048];
    std::string predictionBuffer;
    size_t nPred = 0;
    for (size_t contigKey = 0; contigKey < contigs.size(); contigKey++) {
        fprintf(out, ">%zu\n", contigKey);
        // resultspercontig.cpp:145-190: (orf -> target, orf -> contig) pairs, stable-sorted by (target key, orf key)
        std::vector<std::pair<Matcher::result_t, Matcher::result_t>> results;
        const size_t contigLen = contigs[contigKey].size();
        for (size_t j = 0; j < orfsOfContig[contigKey].size(); j++) {
            const OrfRec &o = orfsOfContig[contigKey][j];
            if (o.key >= alnOfOrf.size()) continue;
            const size_t orfLen = std::max(o.loc.from, o.loc.to) - std::min(o.loc.from, o.loc.to) + 1;   // Orf::getFromDatabase, Orf.cpp:103-116
            Matcher::result_t orfToContig((unsigned int) contigKey, 1, 1, 0, 1, 0, orfLen, 0, (orfLen - 1), orfLen, o.loc.from, o.loc.to, contigLen, "");
            orfToContig.dbKey = o.key;
            for (size_t a = 0; a < alnOfOrf[o.key].size(); a++)
                results.emplace_back(std::make_pair(Matcher::parseAlignmentRecord(alnOfOrf[o.key][a].c_str(), true), orfToContig));
        }
        std::stable_sort(results.begin(), results.end(), [](const std::pair<Matcher::result_t, Matcher::result_t> &l, const std::pair<Matcher::result_t, Matcher::result_t> &r) {
            if (l.first.dbKey < r.first.dbKey) return true;
            if (l.first.dbKey > r.first.dbKey) return false;
            return l.second.dbKey < r.second.dbKey;
        });
        std::string ss;
        for (size_t i = 0; i < results.size(); i++) {
            size_t len = Matcher::resultToBuffer(buffer, results[i].first, false, false);
            ss.append(buffer, len - 1);
            ss.append("\t");
            len = Matcher::resultToBuffer(buffer, results[i].second, false, false);
            ss.append(buffer, len);
        }
        // collectoptimalset.cpp:262-413 on this contig's entry
        std::vector<PotentialExon> plusE, minusE, plusSet, minusSet;
        std::vector<char> data(ss.begin(), ss.end());
        data.push_back('\0');
        char *resultsPtr = data.data();
        const char *entry[255];
        unsigned int currTargetKey = 0;
        bool isFirstIteration = true;
        auto flush = [&]() {
            size_t numIters = 0;
            while ((numIters < maxExonSets) && (plusE.size() > 0 || minusE.size() > 0)) {
                int totalBitScorePlus = findoptimalsetbydp(plusE, plusSet, minIntronLength, maxIntronLength, maxAaOverlap, setGapOpenPenalty, setGapExtendPenalty, dMetaeukTargetCovThr);
                int totalBitScoreMinus = findoptimalsetbydp(minusE, minusSet, minIntronLength, maxIntronLength, maxAaOverlap, setGapOpenPenalty, setGapExtendPenalty, dMetaeukTargetCovThr);
                if (plusSet.size() > 0) {
                    double log2EvaluePlus = log2(totNumOfAAsInTargetDb) + log2(2) - totalBitScorePlus;
                    double combinedEvaluePlus = pow(2, log2EvaluePlus);
                    if (combinedEvaluePlus <= dMetaeukEvalueThr) {
                        Prediction predToWrite(currTargetKey, PLUS, totalBitScorePlus, combinedEvaluePlus, plusSet);
                        Prediction::predictionToBuffer(predictionBuffer, exonLineBuffer, predToWrite);
                        fwrite(predictionBuffer.c_str(), 1, predictionBuffer.size(), out);
                        predictionBuffer.clear();
                        nPred++;
                    }
                }
                if (minusSet.size() > 0) {
                    double log2EvalueMinus = log2(totNumOfAAsInTargetDb) + log2(2) - totalBitScoreMinus;
                    double combinedEvalueMinus = pow(2, log2EvalueMinus);
                    if (combinedEvalueMinus <= dMetaeukEvalueThr) {
                        Prediction predToWrite(currTargetKey, MINUS, totalBitScoreMinus, combinedEvalueMinus, minusSet);
                        Prediction::predictionToBuffer(predictionBuffer, exonLineBuffer, predToWrite);
                        fwrite(predictionBuffer.c_str(), 1, predictionBuffer.size(), out);
                        predictionBuffer.clear();
                        nPred++;
                    }
                }
                plusSet.clear(); minusSet.clear();
                numIters++;
            }
            plusE.clear(); minusE.clear(); plusSet.clear(); minusSet.clear();
        };
        while (*resultsPtr != '\0') {
            const size_t columns = Util::getWordsOfLine(resultsPtr, entry, 255);
            if (columns != 20) { fprintf(stderr, "expected 20 columns\n"); return 1; }
            PotentialExon currExon;
            currExon.setByAln(entry);
            const unsigned int targetKey = currExon.targetKey;
            if (isFirstIteration) { currTargetKey = targetKey; isFirstIteration = false; }
            if (targetKey != currTargetKey) { flush(); currTargetKey = targetKey; }
            const size_t potentialExonAALen = std::abs(currExon.nucleotideLen) / 3;
            if (potentialExonAALen >= minExonAaLength) { if (currExon.strand == PLUS) plusE.emplace_back(currExon); else minusE.emplace_back(currExon); }
            resultsPtr = Util::skipLine(resultsPtr);
        }
        flush();
    }
    fclose(out); fclose(forf); fclose(faln);
    printf("{\"contigs\": %zu, \"predictions\": %zu}\n", contigs.size(), nPred);
    return 0;
}

// The sliced target-profile search of `predictexons contigsDB profileDB` for one slice (see the header of this file).
static int cmdProfileSearch(int argc, char **argv) {
    if (argc < 7) return 2;
    const std::string matdir = argv[2], profData = argv[3], profIndex = argv[4], outdir = argv[6];
    std::vector<std::string> frags = readLines(argv[5]);
    float sensitivity = 4.0f;
    double evalThr = 100.0, evalAbs = -1.0;
    int threads = 1, forcedK = 0;
    std::string keysPath;
    for (int a = 7; a < argc; a++) {
        std::string s = argv[a];
        if (s == "-s") sensitivity = atof(argv[++a]);
        else if (s == "-k") forcedK = atoi(argv[++a]);
        else if (s == "-e") evalThr = atof(argv[++a]);
        else if (s == "--eval-abs") evalAbs = atof(argv[++a]);
        else if (s == "--keys") keysPath = argv[++a];
        else if (s == "--threads") threads = atoi(argv[++a]);
    }
    mkdir(outdir.c_str(), 0755);
    omp_set_num_threads(threads);
    // the fragment DB: data in the order of fragments.txt (= the prefilter's target numbering, LINEAR_ACCCESS), keys from --keys
    std::vector<unsigned int> fragKeys(frags.size());
    for (size_t i = 0; i < frags.size(); i++) fragKeys[i] = (unsigned int) i;
    if (!keysPath.empty()) {
        std::vector<std::string> k = readLines(keysPath.c_str());
        if (k.size() != frags.size()) { fprintf(stderr, "--keys: %zu keys for %zu fragments\n", k.size(), frags.size()); return 2; }
        for (size_t i = 0; i < frags.size(); i++) fragKeys[i] = (unsigned int) strtoul(k[i].c_str(), NULL, 10);
    }
    const std::string tdb = outdir + "/_tdb", qdb = outdir + "/_qdb";
    writeSeqDb(tdb, frags, &fragKeys);
    {   // the profile DB as a DB triple: data file linked, index copied, dbtype = profile
        unlink(qdb.c_str());
        char *real = realpath(profData.c_str(), NULL);
        if (!real || symlink(real, qdb.c_str()) != 0) { fprintf(stderr, "cannot link %s\n", profData.c_str()); return 2; }
        free(real);
        std::ifstream in(profIndex.c_str(), std::ios::binary);
        std::ofstream out((qdb + ".index").c_str(), std::ios::binary);
        out << in.rdbuf();
        FILE *t = fopen((qdb + ".dbtype").c_str(), "wb");
        int dbtype = Parameters::DBTYPE_HMM_PROFILE;
        fwrite(&dbtype, 4, 1, t);
        fclose(t);
    }
    DBReader<unsigned int> tdbr(tdb.c_str(), (tdb + ".index").c_str(), threads, DBReader<unsigned int>::USE_INDEX | DBReader<unsigned int>::USE_DATA);
    tdbr.open(DBReader<unsigned int>::LINEAR_ACCCESS);
    DBReader<unsigned int> qdbr(qdb.c_str(), (qdb + ".index").c_str(), threads, DBReader<unsigned int>::USE_INDEX | DBReader<unsigned int>::USE_DATA);
    qdbr.open(DBReader<unsigned int>::LINEAR_ACCCESS);
    const int querySeqType = qdbr.getDbtype(), targetSeqType = tdbr.getDbtype();
    if (!Parameters::isEqualDbtype(querySeqType, Parameters::DBTYPE_HMM_PROFILE)) { fprintf(stderr, "not a profile DB\n"); return 2; }
    const size_t maxSeqLen = 65535;
    const std::string blosum = matdir + "/blosum62.out";
    // Prefiltering.cpp:72-76: profile queries take --sub-mat for both matrices
    BaseMatrix *kmerSubMat = new SubstitutionMatrix(blosum.c_str(), 8.0, -0.2f);
    BaseMatrix *ungappedSubMat = new SubstitutionMatrix(blosum.c_str(), 2.0, -0.2f);
    const int alphabetSize = kmerSubMat->alphabetSize;
    const int kmerSize = forcedK ? forcedK : IndexTable::computeKmerSize(tdbr.getAminoAcidDBSize());
    float kmerThrBest = FLT_MAX;                                   // Prefiltering.cpp:1033-1046 (profile search, no context pseudo counts)
    if (kmerSize == 5) { float base = 108.8; kmerThrBest = base - (sensitivity * 4.7); }
    else if (kmerSize == 6) { float base = 134.35; kmerThrBest = base - (sensitivity * 6.15); }
    else if (kmerSize == 7) { float base = 149.15; kmerThrBest = base - (sensitivity * 6.85); }
    const int kmerThr = static_cast<int>(kmerThrBest);
    const size_t nFrag = tdbr.getSize(), nProf = qdbr.getSize();
    // Search.cpp:366-372: the e-value threshold scaled by #fragments / #profiles, passed on as text; --max-seqs = max(300, #fragments)
    {
        evalThr *= ((float) nFrag) / nProf;
        char txt[64];
        snprintf(txt, sizeof(txt), "%g", evalThr);
        evalThr = strtod(txt, NULL);
        if (evalAbs >= 0) evalThr = evalAbs;
    }
    size_t maxResListLen = std::max((size_t) 300, nFrag);
    maxResListLen = std::min(tdbr.getSize(), maxResListLen);
    ScoreMatrix _2mer, _3mer;                                      // not computed for profile queries (Prefiltering.cpp:208-213)
    SequenceLookup *sequenceLookup = NULL;
    Sequence tseq(maxSeqLen, targetSeqType, kmerSubMat, kmerSize, true, true, true, "");
    IndexTable *indexTable = new IndexTable(alphabetSize - 1, kmerSize, false);
    IndexBuilder::fillDatabase(indexTable, &sequenceLookup, *kmerSubMat, _3mer, _2mer, &tseq, &tdbr, 0, tdbr.getSize(),
                               0 /* localKmerThr, Prefiltering.cpp:525-527 */, true, false, 0.9f, 0, 0);
    std::vector<std::string> prefOut(nProf), alnOut(nProf);
    size_t totalHits = 0, passed = 0;
#pragma omp parallel num_threads(threads)
    {
        unsigned int thread_idx = (unsigned int) omp_get_thread_num();
        Sequence seq(qdbr.getMaxSeqLen(), querySeqType, kmerSubMat, kmerSize, true, true, true, "");
        QueryMatcher matcher(indexTable, sequenceLookup, kmerSubMat, ungappedSubMat, kmerThr, kmerSize, tdbr.getSize(),
                             std::max(tdbr.getMaxSeqLen(), qdbr.getMaxSeqLen()), maxResListLen, true, 1.0f, true, 15, false, false);
        matcher.setProfileMatrix(seq.profile_matrix);
        char buffer[128];
#pragma omp for schedule(dynamic, 1) reduction(+: totalHits)
        for (size_t id = 0; id < nProf; id++) {
            seq.mapSequence(id, qdbr.getDbKey(id), qdbr.getData(id, thread_idx), qdbr.getSeqLen(id));
            std::pair<hit_t *, size_t> res = matcher.matchQuery(&seq, UINT_MAX, false);
            for (size_t i = 0; i < res.second; i++) {
                hit_t *h = res.first + i;
                h->seqId = tdbr.getDbKey(h->seqId);
                prefOut[id].append(buffer, QueryMatcher::prefilterHitToBuffer(buffer, *h));
            }
            totalHits += res.second;
        }
    }
    // ---- align (Alignment.cpp:152-154,263,279-514)
    BaseMatrix *m = new SubstitutionMatrix(blosum.c_str(), 2.0, 0.0);
    const int gapOpen = 11, gapExtend = 1;
    EvalueComputation evaluer(tdbr.getAminoAcidDBSize(), m, gapOpen, gapExtend);
    std::vector<std::vector<Matcher::result_t>> kept(nProf);
#pragma omp parallel num_threads(threads)
    {
        unsigned int thread_idx = (unsigned int) omp_get_thread_num();
        char buffer[1024 + 32768 * 4];
        Sequence qSeq(maxSeqLen, querySeqType, m, 0, false, true);
        Sequence dbSeq(maxSeqLen, targetSeqType, m, 0, false, true);
        Matcher matcher(querySeqType, targetSeqType, std::max(tdbr.getMaxSeqLen(), qdbr.getMaxSeqLen()), m, &evaluer, true, 1.0f, gapOpen, gapExtend, 0.0f, 40);
#pragma omp for schedule(dynamic, 1) reduction(+: passed)
        for (size_t id = 0; id < nProf; id++) {
            char *data = (char *) prefOut[id].c_str();
            if (*data != '\0') {
                qSeq.mapSequence(id, qdbr.getDbKey(id), qdbr.getData(id, thread_idx), qdbr.getSeqLen(id));
                matcher.initQuery(&qSeq);
            }
            std::vector<Matcher::result_t> &swResults = kept[id];
            while (*data != '\0') {
                hit_t hit = QueryMatcher::parsePrefilterHit(data);
                data = Util::skipLine(data);
                const size_t dbId = tdbr.getId(hit.seqId);
                dbSeq.mapSequence(dbId, hit.seqId, tdbr.getData(dbId, thread_idx), tdbr.getSeqLen(dbId));
                Matcher::result_t res = matcher.getSWResult(&dbSeq, static_cast<int>(static_cast<short>(hit.diagonal)), false, 0, 0.0f, evalThr, Matcher::SCORE_COV, 0, false, false);
                if (res.eval <= evalThr && res.seqId >= 0.0 && Util::hasCoverage(0.0f, 0, res.qcov, res.dbcov) && Util::hasAlignmentLength(11, res.alnLength)) {
                    swResults.emplace_back(res);
                    passed++;
                }
            }
            if (swResults.size() > 1) SORT_SERIAL(swResults.begin(), swResults.end(), Matcher::compareHits);
            for (size_t r = 0; r < swResults.size(); r++) alnOut[id].append(buffer, Matcher::resultToBuffer(buffer, swResults[r], false));
        }
    }
    // '>profile key' blocks in key order
    std::vector<size_t> byKey(nProf);
    for (size_t i = 0; i < nProf; i++) byKey[i] = i;
    std::sort(byKey.begin(), byKey.end(), [&](size_t a, size_t b) { return qdbr.getDbKey(a) < qdbr.getDbKey(b); });
    {
        FILE *f = fopen((outdir + "/pref.txt").c_str(), "w");
        for (size_t k = 0; k < nProf; k++) { fprintf(f, ">%u\n", qdbr.getDbKey(byKey[k])); fputs(prefOut[byKey[k]].c_str(), f); }
        fclose(f);
        f = fopen((outdir + "/aln.txt").c_str(), "w");
        for (size_t k = 0; k < nProf; k++) { fprintf(f, ">%u\n", qdbr.getDbKey(byKey[k])); fputs(alnOut[byKey[k]].c_str(), f); }
        fclose(f);
    }
    // ---- swapresults (util/swapresults.cpp:74-103,254-318): every printed record parsed back, swapped with the e-value of a search
    // against the profile DB, lists sorted with compareHits; every fragment gets an entry
    EvalueComputation swapEvaluer(qdbr.getAminoAcidDBSize(), m, gapOpen, gapExtend);
    unsigned int maxKey = 0;
    for (size_t i = 0; i < nFrag; i++) maxKey = std::max(maxKey, tdbr.getDbKey(i));
    std::vector<std::vector<Matcher::result_t>> swapped(static_cast<size_t>(maxKey) + 1);
    for (size_t k = 0; k < nProf; k++) {
        const size_t id = byKey[k];
        char *data = (char *) alnOut[id].c_str();
        while (*data != '\0') {
            Matcher::result_t res = Matcher::parseAlignmentRecord(data, true);
            const unsigned int fragKey = res.dbKey;
            res.dbKey = qdbr.getDbKey(id);                          // swapresults.cpp: the record moves to the target's list under the query's key
            Matcher::result_t::swapResult(res, swapEvaluer, false);
            swapped[fragKey].emplace_back(res);
            data = Util::skipLine(data);
        }
    }
    {
        FILE *f = fopen((outdir + "/swapped.txt").c_str(), "w");
        char buffer[1024 + 32768 * 4];
        for (size_t key = 0; key <= maxKey; key++) {
            fprintf(f, ">%zu\n", key);
            std::vector<Matcher::result_t> &v = swapped[key];
            if (v.size() > 1) SORT_SERIAL(v.begin(), v.end(), Matcher::compareHits);
            for (size_t j = 0; j < v.size(); j++) fwrite(buffer, 1, Matcher::resultToBuffer(buffer, v[j], false, false), f);
        }
        fclose(f);
    }
    printf("{\"profiles\": %zu, \"fragments\": %zu, \"k\": %d, \"kmer_thr\": %d, \"eval_thr\": %.17g, \"profile_db_residues\": %zu, \"pref_hits\": %zu, \"passed\": %zu}\n",
           nProf, nFrag, kmerSize, kmerThr, evalThr, (size_t) qdbr.getAminoAcidDBSize(), totalHits, passed);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && std::string(argv[1]) == "createindex") return cmdCreateIndex(argc, argv);
    if (argc < 2) { fprintf(stderr, "usage: ref_harness pipeline|sw|submat ...\n"); return 2; }
    std::string cmd = argv[1];
    if (cmd == "submat") return cmdSubmat(argc, argv);
    if (cmd == "pipeline") return cmdPipeline(argc, argv);
    if (cmd == "profilesearch") return cmdProfileSearch(argc, argv);
    if (cmd == "sw") return cmdSw(argc, argv);
    if (cmd == "orfs") return cmdOrfs(argc, argv);
    if (cmd == "exons") return cmdExons(argc, argv);
    return 2;
}
