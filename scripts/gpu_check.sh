#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace, PMC traffic passes.
# Logs -> gpurun_out/ (copied into profiles/ afterwards).
set -u
mkdir -p gpurun_out
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
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc/g$i -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/pmc/g$i.log 2>&1
  echo "group $i rc=$?"
done
cd $R
python scripts/pmc_summary.py gpurun_out/pmc --traffic gpurun_out/pmc_traffic.json | tee gpurun_out/pmc/summary.txt
find gpurun_out/pmc -name "*.csv" -size +8M -delete
fi
