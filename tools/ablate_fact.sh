#!/bin/bash
# Ablations of the factored front (profiling only): alternative builds of the library, each timed by tools/policy_native_bench.py.
# usage (on the GPU box): bash tools/ablate_fact.sh "<-D flags variant 1>" "<variant 2>" ...
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/_ab gpurun_out
SRC=marl-ctf-development_amd/csrc
SRCS=$(make -s -C $SRC print-srcs | sed "s#[^ ]*#$SRC/&#g")  # the shipped library's own source list (csrc/Makefile)
for flags in "$@"; do
  tag=$(echo "$flags" | tr -c 'A-Za-z0-9' '_')
  so=tools/_ab/libctf_hip_$tag.so
  [ -f "$so" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off $flags -shared -o "$so" $SRCS || exit 1
  CTF_LIB_PATH=$so timeout -k 10 300 python tools/policy_native_bench.py 65536 2> gpurun_out/ablate_fact_$tag.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$flags', {k:v for k,v in d.items() if k.startswith(('fact','act_two'))})"
done
