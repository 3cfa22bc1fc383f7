#!/bin/bash
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" > $O/bench_line.json; cut -c1-400 $O/bench_line.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o fk -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-dense > $R/$O/rocprof_bench.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && grep -E "d4w|Name" "$f" | cut -c1-200 | head -12
rm -rf $O/prof
BENCH_ARGS="--no-dense" PMC_GROUPS="fetch write" bash scripts/pmc.sh $O/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
cp $O/pmc/summary.txt $O/pmc_fetch_write_summary.txt; cp $O/pmc/pmc_traffic.json $O/pmc_traffic.json; rm -rf $O/pmc
BENCH_ARGS="--stages mf --no-dense" PMC_GROUPS="sq1 sq2" bash scripts/pmc.sh $O/pmcsq > $O/pmcsq.log 2>&1
cp $O/pmcsq/summary.txt $O/pmc_sq_matched_filter.txt; rm -rf $O/pmcsq; grep -A18 "xcorr_mm" $O/pmc_sq_matched_filter.txt | head -40
