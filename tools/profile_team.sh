set -o pipefail
OUT=$PWD/gpurun_out/prof_team
mkdir -p $OUT
export TMPDIR=/tmp
CMD="$PWD/tools/policy_native_bench.py 65536"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $CMD > $OUT/pmc_sq.log 2>&1 || echo "pmc sq failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $CMD > $OUT/pmc_sq2.log 2>&1 || echo "pmc sq2 failed"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/pmc_sq3 -- python3 $CMD > $OUT/pmc_sq3.log 2>&1 || echo "pmc sq3 failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $CMD > $OUT/pmc_write.log 2>&1 || echo "pmc write failed"
