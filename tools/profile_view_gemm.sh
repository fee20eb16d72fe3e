#!/bin/bash
# SQ counters of k_view_gemm against the library's GEMM kernel (profiling only): bash tools/profile_view_gemm.sh
set -o pipefail
OUT=$PWD/gpurun_out/prof_vg
mkdir -p $OUT
export TMPDIR=/tmp
CMD="$PWD/tools/view_gemm_bench.py"
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
        key="k_view_gemm" if "k_view_gemm" in k else ("library:"+k[:40] if ("Cijk" in k or "gemm" in k.lower()) else None)
        if key is None: continue
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[key][r["Dispatch_Id"]]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    for key in vals:
        print(p, key, "dur_us", round(sum(dur[key].values())/len(dur[key])/1e3,1), {c:round(sum(v)/len(v)) for c,v in vals[key].items()})
PY
rm -rf $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c
