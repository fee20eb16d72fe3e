#!/bin/bash
# Per-instance (L2 channel / XCD) split of the write-path counters for a fast and a slow buffer (tools/placement_pmc).
set -o pipefail
export TMPDIR=/tmp
BIN=$PWD/tools/placement_pmc
ROOT=$PWD
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o $BIN $PWD/tools/placement_pmc.hip || exit 1
cd /tmp
for c in TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_BUSY TCC_REQ TCP_TCR_TCP_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY; do
  rm -rf /tmp/dims_$c
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format json -d /tmp/dims_$c -- $BIN 64 > /tmp/dims_$c.log 2>&1 || { echo "$c: pass failed"; continue; }
  grep -E "^fast|^tile fill" /tmp/dims_$c.log | head -2
  python3 $ROOT/tools/placement_dims.py /tmp/dims_$c
done
