#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc|Error" $O/pytest_gpu.log | tail -5
timeout 600 python scripts/time_host_api.py 2>&1 | grep "^{" | tee $O/time_host_api.txt
timeout 900 python bench.py --config stream --steps 10 --warmup 2 2>$O/bench_stream_resident.err | grep "^{" > $O/bench_stream_resident.json; cut -c1-700 $O/bench_stream_resident.json
timeout 900 python bench.py --config stream --from-host --steps 10 --warmup 2 2>$O/bench_stream_from_host.err | grep "^{" > $O/bench_stream_from_host.json; cut -c1-900 $O/bench_stream_from_host.json; tail -3 $O/bench_stream_from_host.err
tail -5 $O/bench_stream_resident.err
