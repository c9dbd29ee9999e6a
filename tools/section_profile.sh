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
if [ -f build_variants/librt_hip_only0.so ]; then  # one section timed per build (-DRT_PROFILE_SECTIONS=1 -DRT_PROFILE_ONLY=i)
  rm -f gpurun_out/section_single.log
  for i in 0 1 2 3 4; do
    cp build_variants/librt_hip_only$i.so raytracer_amd/librt_hip.so
    SECTION_ONLY=$i timeout -k 10 200 python3 -u tools/section_profile.py 2>&1 | grep ONLY | tee -a gpurun_out/section_single.log
  done
  python3 - <<'PY' | tee gpurun_out/section_single.md
import collections
rows = collections.OrderedDict()
for l in open("gpurun_out/section_single.log"):
    cfg, label, share, cyc, life = [x.strip() for x in l[5:].split("|")]
    rows.setdefault(cfg, []).append((label, float(share), float(cyc), float(life)))
print("| config | " + " | ".join(l for l, *_ in next(iter(rows.values()))) + " | [E][F][A] + loop | wave-cycles per iteration |")
print("|---|" + "---|" * (len(next(iter(rows.values()))) + 2))
for cfg, parts in rows.items():
    life = sum(p[3] for p in parts) / len(parts)
    print(f"| {cfg} | " + " | ".join(f"{p[1]:.1f} %" for p in parts) + f" | {100.0 - sum(p[1] for p in parts):.1f} % | {life:.0f} |")
PY
fi
if [ -f build_variants/librt_hip_hist.so ]; then
  cp build_variants/librt_hip_hist.so raytracer_amd/librt_hip.so
  SECTION_LANES=hist timeout -k 10 200 python3 -u tools/section_profile.py 2>&1 | tee gpurun_out/section_hist.log
fi
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
