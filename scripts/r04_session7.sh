#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "fk or pipeline or fuzz or design or reference" > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc|Error" $O/pytest_gpu.log | tail -4
(timeout 400 python scripts/time_fk_masks.py classic ninf hybrid dense step4; NX=11020 NS=12000 timeout 200 python scripts/time_fk_masks.py ninf classic) 2>/dev/null | grep "^{" | tee $O/time_fk_masks.txt | cut -c1-230
PMC_CMD="python $GRAFT_REPO_ROOT/scripts/time_fk_masks.py dense ninf" PMC_GROUPS="sq2" bash scripts/pmc.sh $O/pmcsq > $O/pmcsq.log 2>&1
cp $O/pmcsq/summary.txt $O/pmc_sq_fk_dense_ninf.txt; rm -rf $O/pmcsq; grep -A9 "fkf_passB" $O/pmc_sq_fk_dense_ninf.txt | grep -E "passB|CONFLICT|IDX_ACTIVE|ACTIVE_INST_VALU|WAIT_INST_ANY" | head -30
