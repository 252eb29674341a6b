#!/bin/bash
# a variant of the product library with other compile-time shapes of the prefilter kernels (experiments; the result is not tracked):
#   tools/build_variant.sh <name> -DMK_STREAM_CAP_A=32768 ...   ->  tools/_variants/libmetaeuk_amd_<name>.so   (use: METAEUK_AMD_LIB=<that> python bench.py ...)
set -e
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
O=$R/tools/_variants
mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -ffp-contract=off -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $R/metaeuk_amd/csrc/${SRC:-mk_prefilter.hip} -o $O/mk_prefilter_$NAME.o
OBJS=$(ls $R/metaeuk_amd/lib/obj/*.o | grep -v mk_prefilter.hip.o)
/opt/rocm/bin/hipcc $FLAGS -shared $OBJS $O/mk_prefilter_$NAME.o -o $O/libmetaeuk_amd_$NAME.so
rm -f $O/mk_prefilter_$NAME.o
echo $O/libmetaeuk_amd_$NAME.so
