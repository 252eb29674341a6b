#!/bin/bash
# Process-level goldens of the PROFILE-target path (BASELINE config 4) from the REAL reference binary:
# `metaeuk predictexons contigsDB profileDB` runs M/data/workflow/searchslicedtargetprofile.sh -- the profiles are the QUERIES of
# prefilter/align, the 6-frame fragments the indexed targets, swapresults turns the lists round.  The binary is built outside the
# repository exactly like tests/golden/make_process_golden.sh does (run that first); only its output DBs become fixtures (data):
#   prof_db.bin.gz / prof_db.index    the profile DB itself (result2profile of the first 100 e2e targets searched against themselves:
#                                     25 bytes per column, Sequence.h:458-471); keys = line numbers of e2e_targets.txt.gz
#   prof_frag_order.txt.gz            the fragment keys in the order of their data offsets in aa_6f: the prefilter numbers its targets in
#                                     that order (DBReader LINEAR_ACCCESS), which decides ties and the order inside the index lists
#   prof_pref.txt.gz                  prefilter profileDB aa_6f   (per profile: fragment key, score, diagonal)
#   prof_aln.txt.gz                   align profileDB aa_6f       (per profile; the workflow's final `align` of the merged key lists)
#   prof_search_res.txt.gz            swapresults                 (per fragment: profile key ... with the e-value of the swapped search)
#   prof_calls.txt.gz                 dp_predictions of the whole predictexons run
# The stages are ALSO run one by one with the parameter strings the workflow printed, and the chained result is checked to be the
# workflow's own search_res, so prof_pref / prof_aln are what the workflow computed inside.  Fragments = e2e_process_orfs.txt.gz
# (checked).  Host L2 = 2097152 bytes.  predictexons default -s 4 -> k-mer threshold 109 (Prefiltering.cpp:1038-1040).
set -e
R=$(cd $(dirname $0)/../.. && pwd)
W=${1:-/tmp/ppg}
M=/tmp/ref-build/src/metaeuk
[ -x $M ] || { echo "build the reference binary first: tests/golden/make_process_golden.sh"; exit 1; }
rm -rf $W && mkdir -p $W/tmp0 $W/tmp1 $W/stage
cd $W
python3 - <<PY
import gzip
def lines(n): return gzip.open('$R/tests/golden/' + n, 'rt').read().splitlines()
open('targets.faa', 'w').write("".join(">t%d\n%s\n" % (i, s) for i, s in enumerate(lines('e2e_targets.txt.gz')[:100])))
open('contigs.fna', 'w').write("".join(">c%d\n%s\n" % (i, s) for i, s in enumerate(lines('e2e_contigs.txt.gz'))))
PY
$M createdb targets.faa targetsDB --shuffle 0 > /dev/null
$M createdb contigs.fna contigsDB --shuffle 0 > /dev/null
$M search targetsDB targetsDB selfres tmp0 --threads 4 > self.log 2>&1
$M result2profile targetsDB targetsDB selfres profDB --threads 4 > r2p.log 2>&1
$M predictexons contigsDB profDB calls tmp1 --remove-tmp-files 0 --threads 4 > run.log 2>&1
T=tmp1/latest
PREF=$(grep -m1 "^prefilter " run.log | sed 's/^prefilter [^ ]* [^ ]* [^ ]* //')
ALN_IT=$(grep "^align " run.log | sed -n 1p | sed 's/^align [^ ]* [^ ]* [^ ]* [^ ]* //')
ALN=$(grep "^align " run.log | sed -n 2p | sed 's/^align [^ ]* [^ ]* [^ ]* [^ ]* //')
SWAP=$(grep -m1 "^swapresults " run.log | sed 's/^swapresults [^ ]* [^ ]* [^ ]* [^ ]* //')
eval $M prefilter profDB $T/aa_6f stage/pref $PREF > stage/pref.log 2>&1
eval $M align profDB $T/aa_6f stage/pref stage/aln_it $ALN_IT > stage/aln_it.log 2>&1
eval $M align profDB $T/aa_6f stage/aln_it stage/aln $ALN > stage/aln.log 2>&1
eval $M swapresults profDB $T/aa_6f stage/aln stage/swapped $SWAP > stage/swap.log 2>&1
python3 - <<PY
import gzip, os, shutil
def read_db(base):
    if os.path.exists(base): data = open(base, 'rb').read()
    else:
        data, i = b'', 0
        while os.path.exists(base + '.%d' % i): data += open(base + '.%d' % i, 'rb').read(); i += 1
    return {int(l.split('\t')[0]): data[int(l.split('\t')[1]):int(l.split('\t')[1]) + int(l.split('\t')[2]) - 1].decode() for l in open(base + '.index')}
T, G = '$W/tmp1/latest/', '$R/tests/golden/'
blocks = lambda d: "".join(">%d\n%s" % (k, d[k]) for k in sorted(d))
def write(n, t):
    with gzip.open(G + n, 'wt', compresslevel=9) as f: f.write(t)
aa, hdr = read_db(T + 'aa_6f'), read_db(T + 'aa_6f_h')
assert "".join("%s\t%s" % (hdr[k].rstrip("\n"), aa[k]) for k in sorted(aa)) == gzip.open(G + 'e2e_process_orfs.txt.gz', 'rt').read(), "fragments differ from e2e_process_orfs"
swapped = blocks(read_db('$W/stage/swapped'))
assert swapped == blocks(read_db(T + 'search_res')), "the stage-by-stage chain differs from the workflow's search_res"
with open('$W/profDB', 'rb') as f, gzip.open(G + 'prof_db.bin.gz', 'wb', compresslevel=9) as g: g.write(f.read())
shutil.copy('$W/profDB.index', G + 'prof_db.index')
order = sorted((int(l.split('\t')[1]), int(l.split('\t')[0])) for l in open(T + 'aa_6f.index'))
write("prof_frag_order.txt.gz", "".join("%d\n" % k for _, k in order))
write("prof_pref.txt.gz", blocks(read_db('$W/stage/pref')))
write("prof_aln.txt.gz", blocks(read_db('$W/stage/aln')))
write("prof_search_res.txt.gz", swapped)
write("prof_calls.txt.gz", blocks(read_db('$W/calls')))
print("profile-path goldens written;", sum(len(v.splitlines()) for v in read_db('$W/stage/pref').values()), "prefilter hits,",
      sum(len(v.splitlines()) for v in read_db('$W/stage/aln').values()), "alignments,", len(read_db('$W/calls')), "call entries")
PY
grep -h "k-mer similarity threshold\|k-mers per position\|DB matches per\|Index table k-mer threshold" stage/pref.log
