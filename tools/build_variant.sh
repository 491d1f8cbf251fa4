#!/bin/bash
# builds a measurement variant of libkartohip.so: tools/build_variant.sh <out.so> [--instrumented] <extra hipcc flags...>
# --instrumented: the round-4 measurement scaffolding of the LDS-staged scoring kernels (-DKH_LDS_EXP=n, -DKH_LDS_TIMING, eight waves
# per angle), which left the shipped sources in round 5, is patched back into a scratch copy first
# (tools/patches/r4_k_score_lds_instrumentation.patch; it applies to the matcher_kernels.hip of the commit that added it).
out=$1; shift
src="$(cd "$(dirname "$0")/../slam_toolbox_amd/csrc" && pwd)"
if [ "$1" = "--instrumented" ]; then
  shift
  tmp=$(mktemp -d) && cp "$src"/* "$tmp"/ && (cd "$tmp" && patch -p3 < "$src/../../tools/patches/r4_k_score_lds_instrumentation.patch") || exit 1
  src=$tmp
fi
cd "$src"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-value -shared -ldl "$@" -o "$out" \
  matcher_host.cpp matcher_seq.cpp matcher_seq.hip matcher_group.cpp matcher_kernels.hip spa_host.cpp spa_symbolic.cpp spa_kernels.hip graph.hip occupancy.hip lifelong.hip comm.cpp mapper_host.cpp
