#!/bin/bash
# tools/build_variant.sh <name> [-DNAME=VALUE ...]: builds variants/<name>.so with extra defines for pt_trace.hip
set -e
name=$1; shift
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c pt_trace.hip -o /tmp/variant_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../variants/$name.so /tmp/variant_$name.o pt_capi.o pt_build.o
