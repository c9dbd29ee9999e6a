#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -u tools/scaling_proxy.py > gpurun_out/r3_scaling_proxy.md 2> gpurun_out/r3_scaling_proxy.err || { tail -5 gpurun_out/r3_scaling_proxy.err; exit 1; }
cat gpurun_out/r3_scaling_proxy.md
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
cp build_variants/librt_hip_lanes.so raytracer_amd/librt_hip.so
SECTION_LANES=1 timeout -k 10 300 python3 -u tools/section_profile.py > gpurun_out/r3_section_lanes.log 2>&1 || { tail -5 gpurun_out/r3_section_lanes.log; cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so; exit 1; }
SECTION_LANES=1 SECTION_BATCH=20 timeout -k 10 400 python3 -u tools/section_profile.py >> gpurun_out/r3_section_lanes.log 2>&1 || { tail -5 gpurun_out/r3_section_lanes.log; cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so; exit 1; }
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
grep -A 8 "^C4 strict" gpurun_out/r3_section_lanes.log
