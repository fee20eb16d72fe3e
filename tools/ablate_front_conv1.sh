#!/bin/bash
# Round 5, review item 7: what could ANY cheaper conv1 buy the factored front?  Timing-only builds (wrong values) without the h0 rebuild +
# shared conv1 (FACT_ABLATE bit 4) and also without the conv1 patches (bit 5), against the shipped kernel, interleaved on one box:
# the front's stage time (tools/policy_native_bench.py) and the whole rollout (bench_rollout.py --no-update).
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/_ab gpurun_out
SRC=marl-ctf-development_amd/csrc
SRCS=$(make -s -C $SRC print-srcs | sed "s#[^ ]*#$SRC/&#g")
for v in 0 16 48; do
  so=tools/_ab/libctf_hip_front$v.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -DFACT_ABLATE=$v -shared -o $so $SRCS || exit 1
done
for round in 1 2; do
  for v in 0 16 48; do
    so=$PWD/tools/_ab/libctf_hip_front$v.so
    CTF_LIB_PATH=$so timeout -k 10 300 python tools/policy_native_bench.py 65536 2> gpurun_out/ablate_front_$v.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $round FACT_ABLATE=$v front_x2 %.4f ms' % d['fact_front_x2'])"
    CTF_LIB_PATH=$so timeout -k 10 300 python bench_rollout.py --envs 65536 --steps 16 --no-update 2>> gpurun_out/ablate_front_$v.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $round FACT_ABLATE=$v rollout %.2f M env-steps/s (%.4f s)' % (d['rollout_env_steps_per_s']/1e6, d['rollout_s']))"
  done
done
