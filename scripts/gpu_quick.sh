#!/bin/bash
# Short GPU session: selected parity tests + bench (no profiling).  usage: TESTS="tests/test_fk_gpu.py" bash scripts/gpu_quick.sh
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest ${TESTS:-tests} -m gpu"
timeout 1500 python -m pytest ${TESTS:-tests} -x -q -m gpu -s 2>&1 | tail -${TAIL:-30} | tee gpurun_out/pytest_gpu.log
if [ "${BENCH:-1}" = "1" ]; then
echo "== bench"
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS:---no-cpu} 2>&1 | tail -3 | tee gpurun_out/bench.log
fi
