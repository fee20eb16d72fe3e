#!/bin/bash
# tools/profile.sh for several workloads in one call: bash tools/profile_all.sh <round> <key> [<key> ...]
# keys: arena (the headline), arena_32768 (configs[3]'s per-GPU shard, rank 3 of 8), arena20, split
ROUND=$1; shift
for key in "$@"; do
  case $key in
    arena) tag=$ROUND; args="--no-secondary";;
    arena_32768) tag=${ROUND}_arena32768; args="--no-secondary --envs-per-gpu 32768 --env-offset 98304";;
    arena20) tag=${ROUND}_arena20; args="--no-secondary --workload arena20";;
    split) tag=${ROUND}_split; args="--no-secondary --workload split --envs-per-gpu 4096";;
    *) echo "unknown key $key"; exit 1;;
  esac
  BENCH_ARGS="$args" bash tools/profile.sh $tag > gpurun_out/profile_$tag.log 2>&1 || exit 1
  echo "$tag done: $(grep -c csv gpurun_out/profile_$tag.log) csv files"
done
# the raw rocprofv3 output is ~35 MB per workload (gpurun merges at most 64 MiB back): condense on the box, keep the summaries only
mkdir -p gpurun_out/${ROUND}_summaries
for key in "$@"; do
  case $key in
    arena) tag=$ROUND; wl=arena_65536; args="--no-secondary";;
    arena_32768) tag=${ROUND}_arena32768; wl=arena_32768; args="--no-secondary --envs-per-gpu 32768 --env-offset 98304";;
    arena20) tag=${ROUND}_arena20; wl=arena20_65536; args="--no-secondary --workload arena20";;
    split) tag=${ROUND}_split; wl=split_4096; args="--no-secondary --workload split --envs-per-gpu 4096";;
  esac
  PROFILE_BENCH_ARGS="$args" python3 tools/summarize_profile.py gpurun_out/prof_$tag $tag $wl || exit 1
  cp profiles/${tag}_kernel_stats.csv profiles/${tag}_pmc_summary.json profiles/${tag}_bench_under_rocprof.json profiles/${tag}_PROVENANCE.txt profiles/${tag}_timed_region_kernels.json gpurun_out/${ROUND}_summaries/
  rm -rf gpurun_out/prof_$tag
done
cp profiles/traffic.json gpurun_out/${ROUND}_summaries/
