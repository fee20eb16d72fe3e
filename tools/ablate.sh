#!/bin/bash
# Ablation builds of the observe kernel (profiling only): what does each phase cost?
# Usage (on the GPU box): bash tools/ablate.sh "0 1 2 4 8 3"
set -o pipefail
cd "$(dirname "$0")/.."
CS=marl-ctf-development_amd/csrc
mkdir -p gpurun_out/ablate
for v in ${1:-0 1 2 4 8}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DOBS_ABLATE=$v -shared \
      -o gpurun_out/ablate/libctf_ab$v.so $CS/ctf_abi.hip $CS/ctf_kernels.hip || exit 1
  CTF_LIB_PATH=$PWD/gpurun_out/ablate/libctf_ab$v.so timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('OBS_ABLATE=$v', d['kernels_ms'], 'value=%.1fM' % (d['value']/1e6))"
done
