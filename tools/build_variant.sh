#!/bin/bash
# Builds build_variants/librt_hip_<tag>.so from the in-tree sources with extra hipcc flags (A/B experiments;
# tools/ab_variants.sh runs them on the GPU box). usage: tools/build_variant.sh <tag> [extra flags...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -I../../include -I."
mkdir -p ../../build_variants /tmp/bv_$tag
/opt/rocm/bin/hipcc $F "$@" -c rt_kernels.hip -o /tmp/bv_$tag/rt_kernels.o
/opt/rocm/bin/hipcc $F "$@" -x hip -c rt_api.cpp -o /tmp/bv_$tag/rt_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/librt_hip_$tag.so /tmp/bv_$tag/rt_kernels.o /tmp/bv_$tag/rt_api.o
echo built build_variants/librt_hip_$tag.so
