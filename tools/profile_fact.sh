#!/bin/bash
# SQ counters of the factored-fc1 kernels (profiling only): bash tools/profile_fact.sh
set -o pipefail
OUT=$PWD/gpurun_out/prof_fact
mkdir -p $OUT
export TMPDIR=/tmp
CMD="$PWD/tools/policy_native_bench.py 65536"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_a -- python3 $CMD > $OUT/pmc_a.log 2>&1 || echo "pmc a failed"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_b -- python3 $CMD > $OUT/pmc_b.log 2>&1 || echo "pmc b failed"
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_c -- python3 $CMD > $OUT/pmc_c.log 2>&1 || echo "pmc c failed"
python3 - <<PY
import csv,glob,collections
for p in ("pmc_a","pmc_b","pmc_c"):
    f=glob.glob("$OUT/"+p+"/**/*_counter_collection.csv",recursive=True)
    if not f: print(p,"missing"); continue
    vals=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        key=None
        for name in ("k_policy_features_fact","k_policy_fc1_patch","k_policy_features_team","k_fact_assign","k_fact_hist","k_fact_scan"):
            if name in k: key=name
        if key is None: continue
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[key][r["Dispatch_Id"]]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    for key in vals:
        print(p, key, "dur_us", round(sum(dur[key].values())/len(dur[key])/1e3,1), {c:round(sum(v)/len(v)) for c,v in vals[key].items()})
PY
