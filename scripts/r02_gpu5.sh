#!/bin/bash
# PMC traffic incl. the band-pass kernels; kernel stats for --stages bp,fk,mf; full GPU suite
set -u
O=gpurun_out/r02b
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostn\|^Librc"
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
BENCH_ARGS="--stages bp,fk,mf --no-dense" PMC_GROUPS="fetch write" bash scripts/pmc.sh $O/pmc > $O/pmc.log 2>&1; tail -4 $O/pmc.log
cp $O/pmc/summary.txt $O/pmc_fetch_write_summary_bp_fk_mf.txt; cp $O/pmc/pmc_traffic.json $O/pmc_traffic_bp_fk_mf.json; rm -rf $O/pmc
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o fk -- python $R/bench.py --stages bp,fk,mf --steps 5 --warmup 2 --no-cpu --no-dense > $R/$O/rocprof_bench.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_bp_fk_mf.csv && grep -E "d4w|Name" "$f" | cut -c1-160 | head -14
rm -rf $O/prof
timeout 600 python bench.py --shard channel --steps 5 --warmup 2 2>/dev/null | grep "^{" > $O/bench_shard_channel_1rank.json; cut -c1-250 $O/bench_shard_channel_1rank.json
timeout 600 python bench.py --config stream --steps 10 --warmup 2 2>/dev/null | grep "^{" > $O/bench_stream_1gpu.json; cut -c1-600 $O/bench_stream_1gpu.json
timeout 600 python scripts/pipeline_bench.py 2>/dev/null | grep "^{" > $O/pipeline_11020x12000.json; cat $O/pipeline_11020x12000.json
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" > $O/bench_line.json; cut -c1-200 $O/bench_line.json
