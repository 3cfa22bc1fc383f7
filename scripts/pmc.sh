#!/bin/bash
# PMC counter passes over the bench (separate runs per counter group; kernel-trace only).
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc/g$i -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu ${PLAN:+--plan $PLAN} > $R/gpurun_out/pmc/g$i.log 2>&1
  echo "group $i rc=$?"
done
cd $R
python scripts/pmc_summary.py gpurun_out/pmc | tee gpurun_out/pmc/summary.txt
find gpurun_out/pmc -name "*.csv" -size +8M -delete
