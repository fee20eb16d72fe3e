#!/bin/bash
# What distinguishes a slow observation buffer?  tools/placement_pmc finds a fast and a slow allocation in ONE process and fills them
# alternately with the render's store pattern (k_fill_tagged<0> = fast, <1> = slow); separate --pmc passes compare the write path.
# (TA_* counters hang the profiler on this pool and are left out.)   bash tools/profile_placement.sh   (on the GPU box)
set -o pipefail
OUT=$PWD/gpurun_out/prof_placement
mkdir -p $OUT
export TMPDIR=/tmp
BIN=$PWD/tools/placement_pmc
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o $BIN $PWD/tools/placement_pmc.hip || exit 1
cd /tmp
$BIN 64 > $OUT/plain_run.txt 2>&1
cat $OUT/plain_run.txt
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "TCC_WRITE_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass_$i -- $BIN 64 > $OUT/pass_$i.log 2>&1 || echo "pass $i failed"
done
# the per-instance split of one counter, if this rocprofv3 reports it (json output keeps the dimensions)
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_WRREQ --kernel-trace --output-format json -d $OUT/pass_dims -- $BIN 64 > $OUT/pass_dims.log 2>&1 || echo "dims pass failed"
python3 - <<PY
import csv, glob, collections, json, os
agg = {0: collections.defaultdict(list), 1: collections.defaultdict(list)}
dur = {0: [], 1: []}
for f in sorted(glob.glob("$OUT/pass_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_fill_tagged" not in k: continue
        tag = 0 if "<0>" in k or "ILi0E" in k else (1 if "<1>" in k or "ILi1E" in k else None)
        if tag is None: continue
        agg[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[tag].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for tag, name in ((0, "fast"), (1, "slow")):
    print(name, "mean duration us %.1f over %d launches" % (sum(dur[tag]) / max(1, len(dur[tag])) / 1e3, len(dur[tag])))
print("%-44s %16s %16s %8s" % ("counter (mean per launch)", "fast", "slow", "slow/fast"))
for c in sorted(agg[0]):
    a = sum(agg[0][c]) / len(agg[0][c]); b = sum(agg[1][c]) / max(1, len(agg[1][c]))
    print("%-44s %16.0f %16.0f %8.3f" % (c, a, b, b / a if a else float("nan")))
for f in glob.glob("$OUT/pass_dims/**/*results.json", recursive=True):
    try:
        d = json.load(open(f))
        print("json keys:", list(d.keys())[:5])
        s = json.dumps(d)[:600]
        print(s)
    except Exception as e:
        print("json parse failed", e)
PY
