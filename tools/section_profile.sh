#!/bin/bash
# Runs tools/section_profile.py with build_variants/librt_hip_profile.so swapped in (GPU box scratch copy only).
set -e
mkdir -p gpurun_out
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
cp build_variants/librt_hip_profile.so raytracer_amd/librt_hip.so
timeout -k 10 200 python3 -u tools/section_profile.py 2>&1 | tee gpurun_out/section_profile.log
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
