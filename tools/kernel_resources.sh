#!/bin/bash
# VGPRs / SGPRs / LDS / scratch of every gfx950 kernel in an object file of the product (metaeuk_amd/lib/obj/*.o) or of the experiment library:
#   tools/kernel_resources.sh metaeuk_amd/lib/obj/mk_sw.hip.o [name filter]
set -e
OBJ=$1; FILTER=${2:-.}
T=$(mktemp -d)
LL=/opt/rocm/lib/llvm/bin
$LL/llvm-objcopy -O binary --only-section=.hip_fatbin $OBJ $T/fat.bin
$LL/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.o
$LL/llvm-readelf --notes $T/dev.o | awk '
/\.lds_size:/ {lds=$NF} /\.name:/ {name=$NF} /\.private_segment_fixed_size:/ {scr=$NF} /\.sgpr_count:/ {sg=$NF} /\.vgpr_count:/ {vg=$NF}
/\.agpr_count:/ {ag=$NF} /\.vgpr_spill_count:/ {sp=$NF; printf "vgpr %-4s agpr %-3s sgpr %-4s lds(static) %-6s scratch %-5s spills %-3s %s\n", vg, ag, sg, lds, scr, sp, name}' | while read l; do
  n=$(echo "$l" | awk '{print $NF}'); echo "${l% *} $(echo $n | c++filt | sed 's/(anonymous namespace):://; s/mk:://g; s/(.*//')"; done | grep -E "$FILTER"
rm -rf $T
