#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc|Error" $O/pytest_gpu.log | tail -4
timeout 300 python scripts/time_stft.py 2>/dev/null | grep "^{" | tee $O/time_stft.txt
D4W_STFT_MM=0 timeout 300 python scripts/time_stft.py 2>/dev/null | grep "^{" | tee -a $O/time_stft.txt
timeout 900 python bench.py --config stream --steps 10 --warmup 2 2>$O/bench_stream_resident.err | grep "^{" > $O/bench_stream_resident.json; cut -c1-600 $O/bench_stream_resident.json
timeout 900 python bench.py --config stream --from-host --files 24 --steps 10 --warmup 2 2>$O/bench_stream_from_host.err | grep "^{" > $O/bench_stream_from_host.json; cut -c1-900 $O/bench_stream_from_host.json
timeout 600 python scripts/pipeline_bench.py 2>/dev/null | grep "^{" | tee $O/pipeline_11020x12000.json
