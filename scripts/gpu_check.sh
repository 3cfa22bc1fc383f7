#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace.  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi"; rocm-smi --showmeminfo vram 2>/dev/null | head -8
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -x -q -m gpu -s 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== rocprof"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o fk -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*kernel_stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
# keep the merge small: drop the raw trace, keep stats
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
