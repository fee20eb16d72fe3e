#!/bin/bash
# HBM traffic of the policy front kernels at a given occupancy (profiling only): bash tools/profile_team.sh <blocks_per_cu>
set -o pipefail
B=${1:-3}
OUT=$PWD/gpurun_out/prof_team_b$B
mkdir -p $OUT
export TMPDIR=/tmp
export CTF_POLICY_BLOCKS_PER_CU=$B
CMD="$PWD/tools/policy_native_bench.py 65536"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $CMD > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $CMD > $OUT/pmc_write.log 2>&1 || echo "pmc write failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $CMD > $OUT/pmc_sq.log 2>&1 || echo "pmc sq failed"
python3 - <<PY
import csv,glob,collections
for p in ("pmc_fetch","pmc_write","pmc_sq"):
    f=glob.glob("$OUT/"+p+"/**/*_counter_collection.csv",recursive=True)
    if not f: print(p,"missing"); continue
    vals=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "k_policy_features" not in k: continue
        key=("team" if "team" in k else "agent")
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[key][r["Dispatch_Id"]]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    for key in vals:
        print("blocks/CU $B", p, key, "dur_us", round(sum(dur[key].values())/len(dur[key])/1e3,1), {c:round(sum(v)/len(v)) for c,v in vals[key].items()})
PY
