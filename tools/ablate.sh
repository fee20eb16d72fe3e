#!/bin/bash
# A/B builds of the kernels with extra -D flags, benchmarked on ONE box, interleaved (boxes differ by ~10 %,
# so numbers from different gpurun calls are not comparable).  Profiling only.
# Usage (on the GPU box): bash tools/ablate.sh "-DOBS_ABLATE=0" "-DOBS_ABLATE=4" "-DOBS_PREFETCH=0" ...
set -o pipefail
cd "$(dirname "$0")/.."
CS=marl-ctf-development_amd/csrc
mkdir -p gpurun_out/ablate
SRCS=$(make -s -C $CS print-srcs | sed "s#[^ ]*#$CS/&#g")  # the shipped library's own source list (csrc/Makefile)
i=0
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off $flags -shared \
      -o gpurun_out/ablate/lib$i.so $SRCS || exit 1
  i=$((i+1))
done
for round in 1 2; do
  i=0
  for flags in "$@"; do
    CTF_LIB_PATH=$PWD/gpurun_out/ablate/lib$i.so timeout -k 10 200 python bench.py --steps 150 --warmup 10 --no-cpu-baseline ${BENCH_ARGS:-} \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('round $round [$flags]', {k: [round(x,4) for x in v] for k,v in d['kernels_ms_p10_p50_p90'].items()}, 'value=%.1fM' % (d['value']/1e6))"
    i=$((i+1))
  done
done
