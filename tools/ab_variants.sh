#!/bin/bash
# A/B of prebuilt kernel variants (build_variants/librt_hip_<tag>.so): copies each over the in-tree
# library on the GPU box's scratch copy and runs tools/quick_perf.py. Scratch tool, not the bench.
# AB_ENV="VAR=v ..." adds environment for every run.
set -e
mkdir -p gpurun_out
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
for f in build_variants/librt_hip_*.so; do
  echo "=== $f $AB_ENV" | tee -a gpurun_out/ab_variants.log
  cp "$f" raytracer_amd/librt_hip.so
  env $AB_ENV timeout -k 10 200 python3 -u tools/quick_perf.py "$@" 2>&1 | grep -E "strict|fast" | cut -c1-100 | tee -a gpurun_out/ab_variants.log
done
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
