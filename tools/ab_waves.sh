set -e
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
for pair in w6:24 w6:28 w7:28 w8:32; do
  tag=${pair%%:*}; w=${pair##*:}
  echo "=== $tag waves/CU $w"
  cp build_variants/librt_hip_$tag.so raytracer_amd/librt_hip.so
  RT_WAVES_PER_CU=$w timeout -k 10 200 python3 -u tools/quick_perf.py c3 c4 c2 2>&1 | grep -E "strict|fast" | cut -c1-100
done
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
