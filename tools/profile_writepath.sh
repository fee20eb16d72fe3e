#!/bin/bash
# Write-path PMC comparison: k_observe (inside bench.py) vs the bare store stream (tools/store_bw5).
# (A fourth pass with the TA_* counters hung on this pool — the profiler never returned and the run was killed for silence;
# it is left out on purpose.)
set -o pipefail
OUT=$PWD/gpurun_out/prof_wp
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="$PWD/bench.py --steps 60 --warmup 20 --no-cpu-baseline"
BW=$PWD/tools/store_bw5
cd /tmp
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_WRITE_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/obs_$i -- python3 $BENCH > $OUT/obs_$i.log 2>&1 || echo "obs pass $i failed"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/bw_$i -- $BW > $OUT/bw_$i.log 2>&1 || echo "bw pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for kind in ("obs", "bw"):
    agg = collections.defaultdict(list); dur = []
    for d in sorted(glob.glob("$OUT/%s_*/" % kind)):
        for f in glob.glob(d + "*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "k_observe" in r["Kernel_Name"] or "k_store_stream" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(kind, "mean duration us", sum(dur) / len(dur) / 1e3)
    for k, v in sorted(agg.items()):
        v = v[len(v) // 2:]  # the warm half
        print("   %-40s %14.0f" % (k, sum(v) / len(v)))
PY
