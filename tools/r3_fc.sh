#!/bin/bash
set -o pipefail
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
for tag in "$@"; do
  if [ "$tag" = base ]; then cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so; else cp build_variants/librt_hip_$tag.so raytracer_amd/librt_hip.so; fi
  echo "== $tag $FC_ENV" | tee -a gpurun_out/r3_fixed_cost.log
  env $FC_ENV timeout -k 10 300 python3 -u tools/r3_fixed_cost.py 2>&1 | grep "^rank" | tee -a gpurun_out/r3_fixed_cost.log
done
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
