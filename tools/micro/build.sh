#!/bin/bash
# builds the experiment library (not part of the product) against the product's headers: tools/micro/_build/libmk_experiments.so
set -e
R=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $R/tools/micro/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-unused-result $R/tools/micro/mk_experiments.hip -o $R/tools/micro/_build/libmk_experiments.so
echo $R/tools/micro/_build/libmk_experiments.so
