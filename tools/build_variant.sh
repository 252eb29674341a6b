#!/bin/bash
# Kernel-variant experiments: builds metaeuk_amd/lib/variants/lib<name>.so from the current sources with extra -D flags for
# mk_prefilter.hip / mk_sw.hip / mk_align.hip; run with METAEUK_AMD_LIB=<that file> (metaeuk_amd/api.py honours it).
#   tools/build_variant.sh nt1 -DMK_NT_LOADS=1
set -e
name=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
O=$R/metaeuk_amd/lib/variants/obj_$name
mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -ffp-contract=off -Wno-unused-value -Wno-unused-result"
pids=""
for f in mk_prefilter.hip mk_sw.hip mk_align.hip; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $R/metaeuk_amd/csrc/$f -o $O/$f.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
objs=""
for f in mk_host.cpp mk_exons.cpp mk_indexfile.cpp mk_abi.cpp mk_derive.hip mk_orf.hip mk_profile.hip mk_kmer7.hip mk_index.hip mk_synth.cpp; do objs="$objs $R/metaeuk_amd/lib/obj/$f.o"; done
/opt/rocm/bin/hipcc $FLAGS -shared $objs $O/mk_prefilter.hip.o $O/mk_sw.hip.o $O/mk_align.hip.o -o $R/metaeuk_amd/lib/variants/lib$name.so
echo $R/metaeuk_amd/lib/variants/lib$name.so
