#!/bin/bash
# Process-level goldens from the REAL reference binary (round 2): `metaeuk predictexons` itself, not the harness.
# The binary is built OUTSIDE the repository from a copy of the read-only reference tree (cmake + make, the reference's own
# recipe; the missing large blob K4000.crf, which nothing on this path reads, is a 3-line text stub) -- nothing of it is kept;
# only its OUTPUT DBs become fixtures (data), as '>key' blocks sorted by key:
#   e2e_process_orfs.txt.gz          aa_6f_h + aa_6f: "header<TAB>protein" per fragment        (extractorfs + translatenucs)
#   e2e_process_pref.txt.gz          pref_0                                                    (prefilter, -s 5.7)
#   e2e_process_aln.txt.gz           search_res                                                (align)
#   e2e_exons_expected.txt.gz        dp_predictions == the harness-made fixture of round 1, byte for byte (checked below)
#   e2e_process_calls_default_s4.txt.gz   dp_predictions of a run WITHOUT -s: predictexons passes its own default -s 4 to search
#   e2e_process_pref_k7.txt.gz       `prefilter aa_6f targetsDB -k 7 -s 5.7`: the k-mer size the reference switches to from 3.35e9 target
#                                    residues on (IndexTable.h:439-449), forced on the small fixture (threshold 122, 2119 k-mers per position)
#   e2e_process_pref_split3_maxseqs20.txt.gz   `prefilter ... --split 3 --split-mode 0 --max-seqs 20 -s 5.7`: TARGET_DB_SPLIT -- three target ranges
#                                    (68 / 60 / 72 proteins) searched one by one with --max-seqs 6 + 10 each, the lists joined and sorted
#   e2e_process_prefilter_stats.txt  the six statistics lines of that prefilter run's log ("246.638184 k-mers per position" ...)
# Inputs: tests/golden/e2e_targets.txt.gz, e2e_contigs.txt.gz (createdb --shuffle 0, so keys = line numbers).  Host L2 = 2097152.
set -e
R=$(cd $(dirname $0)/../.. && pwd)
W=${1:-/tmp/pg}
rm -rf /tmp/ref-src /tmp/ref-build $W && mkdir -p $W /tmp/ref-build
cp -r /root/reference /tmp/ref-src
printf 'stub\nstub\nstub\n' > /tmp/ref-src/lib/mmseqs/data/resources/K4000.crf
(cd /tmp/ref-build && cmake -DCMAKE_POLICY_VERSION_MINIMUM=3.5 -DCMAKE_BUILD_TYPE=Release -DHAVE_AVX2=1 /tmp/ref-src > cmake.log && make -j7 metaeuk > make.log)
M=/tmp/ref-build/src/metaeuk
cd $W
python3 - <<PY
import gzip
def lines(n): return gzip.open('$R/tests/golden/' + n, 'rt').read().splitlines()
open('targets.faa', 'w').write("".join(">t%d\n%s\n" % (i, s) for i, s in enumerate(lines('e2e_targets.txt.gz'))))
open('contigs.fna', 'w').write("".join(">c%d\n%s\n" % (i, s) for i, s in enumerate(lines('e2e_contigs.txt.gz'))))
PY
$M createdb targets.faa targetsDB --shuffle 0 > /dev/null
$M createdb contigs.fna contigsDB --shuffle 0 > /dev/null
mkdir tmp tmp2
$M predictexons contigsDB targetsDB calls tmp --remove-tmp-files 0 --threads 4 -s 5.7 > run.log 2>&1
grep -A5 "k-mers per position" run.log > $R/tests/golden/e2e_process_prefilter_stats.txt     # the prefilter's run statistics (Prefiltering.cpp:953-975)
$M predictexons contigsDB targetsDB calls4 tmp2 --threads 4 > run4.log 2>&1
$M prefilter tmp/latest/aa_6f targetsDB pref_k7 -k 7 -s 5.7 --threads 4 > k7.log 2>&1      # ~1.5 min: the reference initialises a 21^7 table
$M prefilter tmp/latest/aa_6f targetsDB pref_sp3 --split 3 --split-mode 0 -s 5.7 --max-seqs 20 --threads 4 > sp3.log 2>&1
python3 - <<PY
import glob, gzip, os
def read_db(base):
    if os.path.exists(base): data = open(base, 'rb').read()
    else:
        data, i = b'', 0
        while os.path.exists(base + '.%d' % i): data += open(base + '.%d' % i, 'rb').read(); i += 1
    return {int(l.split('\t')[0]): data[int(l.split('\t')[1]):int(l.split('\t')[1]) + int(l.split('\t')[2]) - 1].decode() for l in open(base + '.index')}
T, G = '$W/tmp/latest/', '$R/tests/golden/'
blocks = lambda d: "".join(">%d\n%s" % (k, d[k]) for k in sorted(d))
def write(n, t):
    with gzip.open(G + n, 'wt', compresslevel=9) as f: f.write(t)
aa, hdr = read_db(T + 'aa_6f'), read_db(T + 'aa_6f_h')
write("e2e_process_orfs.txt.gz", "".join("%s\t%s" % (hdr[k].rstrip("\n"), aa[k]) for k in sorted(aa)))
write("e2e_process_pref.txt.gz", blocks(read_db(glob.glob(T + 'tmp_search/*/pref_0.index')[0][:-6])))
write("e2e_process_aln.txt.gz", blocks(read_db(T + 'search_res')))
write("e2e_process_calls_default_s4.txt.gz", blocks(read_db('$W/calls4')))
write("e2e_process_pref_k7.txt.gz", blocks(read_db('$W/pref_k7')))
write("e2e_process_pref_split3_maxseqs20.txt.gz", blocks(read_db('$W/pref_sp3')))
assert blocks(read_db('$W/calls')) == gzip.open(G + 'e2e_exons_expected.txt.gz', 'rt').read(), "the real predictexons differs from the harness-made exon sets"
print("real metaeuk predictexons -s 5.7 == harness-made e2e_exons_expected: OK")
PY
