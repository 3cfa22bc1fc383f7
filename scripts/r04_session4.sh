#!/bin/bash
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc" $O/pytest_gpu.log | tail -3
timeout 300 python scripts/time_xcorr_mm.py 2>&1 | grep "^{" | tee -a $O/time_xcorr_mm.txt
NX=11020 NS=12000 timeout 300 python scripts/time_xcorr_mm.py 2>&1 | grep "^{" | tee -a $O/time_xcorr_mm.txt
BENCH_ARGS="--stages mf --no-dense" PMC_GROUPS="sq1 sq2" bash scripts/pmc.sh $O/pmcsq > $O/pmcsq.log 2>&1
cp $O/pmcsq/summary.txt $O/pmc_sq_matched_filter.txt; rm -rf $O/pmcsq; grep -A18 "xcorr_mm" $O/pmc_sq_matched_filter.txt | head -20
