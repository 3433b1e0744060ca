#!/bin/bash
# A/B of library builds on the GPU box: tools/ab_lib.sh "<bench args>" <tag> [<tag> ...]   ('' = the release build); env settings may precede a tag as VAR=val:tag
# prints value / ms_per_step / serialized kernel times of every run
ARGS="$1"; shift
cp salt_amd/lib/libsalt_gpu.so /tmp/libsalt_gpu_release.so
for spec in "$@"; do
  envs=""; tag="$spec"
  while [[ "$tag" == *=*:* ]]; do envs="$envs ${tag%%:*}"; tag="${tag#*:}"; done
  if [ "$tag" = "release" ] || [ -z "$tag" ]; then cp /tmp/libsalt_gpu_release.so salt_amd/lib/libsalt_gpu.so; else cp "salt_amd/lib/libsalt_gpu_${tag}.so" salt_amd/lib/libsalt_gpu.so; fi
  out=$(env $envs python3 bench.py $ARGS 2>/dev/null | tail -1)
  echo "== $spec :: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernel_ms_serialized"], "parity", d.get("parity",{}).get("mismatching_reads"), "| pe", d.get("pe",{}).get("value"), d.get("pe",{}).get("kernel_ms_serialized"))')"
done
cp /tmp/libsalt_gpu_release.so salt_amd/lib/libsalt_gpu.so
