#!/bin/bash
# VGPRs, SGPRs, scratch and LDS of every kernel of the built library (no GPU needed).
LIB=${1:-compeg_amd/libcompeg_hip.so}
T=$(mktemp -d)
cp "$LIB" $T/lib.so
(cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null)
for co in $T/*gfx950*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | awk '
    /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.agpr_count:/ {a=$2}
    /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
    /\.wavefront_size:/ {printf "%-70s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d\n", name, v, a, s, p, g}'
done | sed 's/_ZN6compeg[0-9]*//' | sort
rm -rf $T
