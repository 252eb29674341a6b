#!/bin/bash
# builds the experiment library (not part of the product) against the product's headers: tools/micro/_build/libmk_experiments.so,
# and the stand-alone micro benchmarks / reproducers (hammer, random_probe, urem24)
set -e
R=$(cd $(dirname $0)/../.. && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p $R/tools/micro/_build
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-unused-result $R/tools/micro/mk_experiments.hip -o $R/tools/micro/_build/libmk_experiments.so
for t in hammer random_probe urem24; do
  if [ ! -e $R/tools/micro/_build/$t ] || [ $R/tools/micro/$t.hip -nt $R/tools/micro/_build/$t ]; then
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result $R/tools/micro/$t.hip -o $R/tools/micro/_build/$t
  fi
done
echo $R/tools/micro/_build/libmk_experiments.so
