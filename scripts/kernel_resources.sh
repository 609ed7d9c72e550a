#!/bin/bash
# Registers, spills, scratch and LDS of every kernel in the built objects (llvm-readelf --notes on the gfx950 code objects).
# usage: scripts/kernel_resources.sh [object...]      (default: every object of zukelang_amd/csrc)
B=/opt/rocm/lib/llvm/bin
cd "$(dirname "$0")/../zukelang_amd/csrc" || exit 1
for o in ${@:-*.o}; do
  $B/llvm-objcopy -O binary --only-section=.hip_fatbin $o /tmp/kr_$$.fb 2>/dev/null || continue
  $B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/kr_$$.fb --output=/tmp/kr_$$.co 2>/dev/null || continue
  $B/llvm-readelf --notes /tmp/kr_$$.co 2>/dev/null | awk -v obj=$o '
    /\.agpr_count:/ {agpr=$2} /\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {scr=$2}
    /\.sgpr_count:/ {sgpr=$2} /\.vgpr_count:/ {vgpr=$2} /\.vgpr_spill_count:/ {spill=$2; printf "%-14s vgpr %3d agpr %3d sgpr %3d spill %3d scratch %5d lds %6d  %s\n", obj, vgpr, agpr, sgpr, spill, scr, lds, name}'
  rm -f /tmp/kr_$$.co /tmp/kr_$$.fb
done | c++filt | sed 's/zk:://g; s/(.*//'
