#!/bin/bash
# round 2, GPU call 2: whole GPU suite, sharded bench at world 1 (packed path), default bench, pipeline bench
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -s > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
python bench.py --shard channel --steps 5 --warmup 2 --no-gather > gpurun_out/bench_shard_w1.json 2> gpurun_out/bench_shard_w1.err
tail -c 1500 gpurun_out/bench_shard_w1.json
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 3000 gpurun_out/bench_default.json
python scripts/pipeline_bench.py > gpurun_out/pipeline_11020x12000.json 2> gpurun_out/pipeline.err
cat gpurun_out/pipeline_11020x12000.json
