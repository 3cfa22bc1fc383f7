#!/bin/bash
# round 2 evidence run: tests, smoke, bench lines, rocprofv3 kernel stats, PMC traffic, extra configs, CPU C1 column
set -u
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" > $O/bench_line.json; cut -c1-300 $O/bench_line.json
timeout 900 python bench.py --stages bp,fk,mf --steps 10 --warmup 3 --no-cpu --no-dense 2>/dev/null | grep "^{" > $O/bench_bp_fk_mf.json; cut -c1-400 $O/bench_bp_fk_mf.json
timeout 900 python bench.py --config stream --steps 10 --warmup 2 2>/dev/null | grep "^{" > $O/bench_stream_1gpu.json; cat $O/bench_stream_1gpu.json
timeout 900 python bench.py --shard channel --steps 5 --warmup 2 2>/dev/null | grep "^{" > $O/bench_shard_channel_1rank.json; cut -c1-300 $O/bench_shard_channel_1rank.json
timeout 600 python scripts/time_bp.py 2>/dev/null | grep "^{" > $O/time_bp.txt; cat $O/time_bp.txt
timeout 600 python scripts/pipeline_bench.py 2>/dev/null | grep "^{" > $O/pipeline_11020x12000.json; cat $O/pipeline_11020x12000.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o fk -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $R/$O/rocprof_bench.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && grep -E "d4w|Name" "$f" | cut -c1-220 | head -30
rm -rf $O/prof
PMC_GROUPS="fetch write" bash scripts/pmc.sh $O/pmc > $O/pmc.log 2>&1; tail -5 $O/pmc.log
cp $O/pmc/summary.txt $O/pmc_fetch_write_summary.txt; cp $O/pmc/pmc_traffic.json $O/pmc_traffic.json; rm -rf $O/pmc
timeout 900 python scripts/cpu_baseline_c1.py > $O/cpu_baseline_c1.json 2>/dev/null; cat $O/cpu_baseline_c1.json
