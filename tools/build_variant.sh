#!/bin/bash
# builds a measurement variant of libkartohip.so: tools/build_variant.sh <out.so> <extra hipcc flags...>
out=$1; shift
cd "$(dirname "$0")/../slam_toolbox_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-value -shared -ldl "$@" -o "$out" \
  matcher_host.cpp matcher_group.cpp matcher_kernels.hip spa_host.cpp spa_symbolic.cpp spa_kernels.hip graph.hip occupancy.hip lifelong.hip comm.cpp mapper_host.cpp
