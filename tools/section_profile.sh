#!/bin/bash
# Runs tools/section_profile.py with build_variants/librt_hip_profile.so swapped in (GPU box scratch copy only);
# then, when build_variants/librt_hip_lanes.so exists (-DRT_PROFILE_SECTIONS=2), the lane-occupancy form.
set -e
mkdir -p gpurun_out
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
if [ -f build_variants/librt_hip_profile.so ]; then
  cp build_variants/librt_hip_profile.so raytracer_amd/librt_hip.so
  timeout -k 10 200 python3 -u tools/section_profile.py 2>&1 | tee gpurun_out/section_profile.log
fi
if [ -f build_variants/librt_hip_lanes.so ]; then
  cp build_variants/librt_hip_lanes.so raytracer_amd/librt_hip.so
  SECTION_LANES=1 timeout -k 10 200 python3 -u tools/section_profile.py 2>&1 | tee gpurun_out/section_lanes.log
fi
if [ -f build_variants/librt_hip_hist.so ]; then
  cp build_variants/librt_hip_hist.so raytracer_amd/librt_hip.so
  SECTION_LANES=hist timeout -k 10 200 python3 -u tools/section_profile.py 2>&1 | tee gpurun_out/section_hist.log
fi
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
