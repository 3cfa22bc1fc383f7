#!/bin/bash
# PMC counter passes over the bench (separate runs per counter group; kernel-trace only).
# usage: PMC_GROUPS="sq1 sq2 fetch write cache" BENCH_ARGS="--stages mf" bash scripts/pmc.sh [outdir]
#        PMC_CMD="python scripts/one_shape.py 13223 12000" profiles that command instead of the bench
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
declare -A G
G[sq1]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G[sq2]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
G[cache]="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
G[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
# matrix-core pipe: busy cycles summed over the SIMDs, f16 math operations / 512, MFMA instructions, and the time base
G[mfma]="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"
for g in ${PMC_GROUPS:-fetch write}; do
  timeout 600 rocprofv3 --kernel-trace --pmc ${G[$g]} --output-format csv -d $R/$OUT/$g -o pmc -- ${PMC_CMD:-python $R/bench.py --steps 2 --warmup 1 --no-cpu ${BENCH_ARGS:-}} > $R/$OUT/$g.log 2>&1
  echo "group $g rc=$?"
done
cd $R
python scripts/pmc_summary.py $OUT --traffic $OUT/pmc_traffic.json | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +8M -delete
