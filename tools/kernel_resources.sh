#!/bin/bash
# Registers, LDS and scratch of every kernel in pt_trace.hip, from the code object's metadata; leaves the ISA in /tmp/ptk/.
# usage: tools/kernel_resources.sh [extra hipcc flags...]
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p /tmp/ptk
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "$@" --cuda-device-only -save-temps=obj -c pt_trace.hip -o /tmp/ptk/pt_trace_dev.co || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/ptk/pt_trace-hip-amdgcn-amd-amdhsa-gfx950.out | awk '
  /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2}
  /\.wavefront_size:/ {printf "%-60s vgpr %4s sgpr %4s lds %6s scratch %5s\n", substr(name,1,60), v, s, l, p}' | sort -u
