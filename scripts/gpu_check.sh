#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace, PMC traffic passes.
# Logs -> gpurun_out/ (copied into profiles/ afterwards).
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -x -q -m gpu -s 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench.log
if [ "${PROF:-1}" = "1" ]; then
echo "== rocprof kernel trace"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o fk -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $R/gpurun_out/rocprof_bench.log 2>&1
cd $R
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "d4w|Name" "$f" | cut -c1-200
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== pmc FETCH_SIZE / WRITE_SIZE"
PMC_GROUPS="fetch write" bash scripts/pmc.sh gpurun_out/pmc
cp gpurun_out/pmc/pmc_traffic.json gpurun_out/pmc_traffic.json
fi
