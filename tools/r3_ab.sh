#!/bin/bash
# A/B of prebuilt kernel variants on the bench command (round 3). usage: tools/r3_ab.sh "<configs>" <tag> [<tag> ...]
# tag "base" = the in-tree library. Every variant first has to pass the full-size golden digests (bit-exactness), then runs
# bench.py --config <c> for each config. Results: gpurun_out/r3_ab.log
set -o pipefail
mkdir -p gpurun_out
log=gpurun_out/r3_ab.log
cfgs=$1; shift
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
for tag in "$@"; do
  if [ "$tag" = base ]; then cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so; else cp build_variants/librt_hip_$tag.so raytracer_amd/librt_hip.so || exit 1; fi
  if [ -z "$AB_SKIP_GOLDEN" ]; then
    timeout -k 10 300 python3 -u -m pytest tests/test_gpu_golden.py -x -q -m gpu --timeout 120 --timeout-method thread -k "${AB_GOLDEN_K:-checksums}" > gpurun_out/r3_ab_golden_$tag.log 2>&1 || { echo "$tag: GOLDEN FAILED" | tee -a $log; tail -5 gpurun_out/r3_ab_golden_$tag.log; cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so; exit 1; }
  fi
  for c in $cfgs; do
    r=$(env $AB_ENV timeout -k 10 120 python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds ${AB_SECONDS:-1.5} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f Gsamples/s %.4f ms_per_step' % (d['value']/1e9, d['ms_per_step']))") || { cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so; exit 1; }
    echo "$tag $c: $r" | tee -a $log
  done
done
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
