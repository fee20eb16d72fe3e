#!/bin/bash
# Ablation timings of k_policy_features (profiling only): builds side-by-side libraries with POL_ABLATE bits and times each.
# Usage (on the GPU box through gpurun): bash tools/ablate_policy.sh "0 1 2 4 8 3" [envs]
set -o pipefail
VARIANTS=${1:-"0 1 2 4 8"}
ENVS=${2:-65536}
SRC=marl-ctf-development_amd/csrc
mkdir -p tools/_ab gpurun_out/ablate_policy
SRCS=$(make -s -C $SRC print-srcs | sed "s#[^ ]*#$SRC/&#g")  # the shipped library's own source list (csrc/Makefile)
for v in $VARIANTS; do
  so=tools/_ab/libctf_hip_pol$v.so
  if [ ! -f $so ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -DPOL_ABLATE=$v ${EXTRA_DEFS:-} \
      -shared -o $so $SRCS || exit 1
  fi
done
if [ "${BUILD_ONLY:-0}" = "1" ]; then exit 0; fi
for v in $VARIANTS; do
  CTF_LIB_PATH=$PWD/tools/_ab/libctf_hip_pol$v.so timeout -k 10 200 python tools/policy_native_bench.py $ENVS > gpurun_out/ablate_policy/v$v.json 2> gpurun_out/ablate_policy/v$v.err || echo "variant $v failed"
  python - <<PY
import json
d=json.load(open("gpurun_out/ablate_policy/v$v.json"))
print("POL_ABLATE=$v features_ms", d["features"], "two_teams_per_agent", d.get("features_two_teams_per_agent"), "shared_view", d.get("features_two_teams_shared_view"))
PY
done
