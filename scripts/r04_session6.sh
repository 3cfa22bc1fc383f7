#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --config stream --steps 10 --warmup 2 2>$O/bench_stream_resident.err | grep "^{" > $O/bench_stream_resident.json; cut -c1-700 $O/bench_stream_resident.json
timeout 900 python bench.py --config stream --from-host --steps 10 --warmup 2 2>$O/bench_stream_from_host.err | grep "^{" > $O/bench_stream_from_host.json; cut -c1-900 $O/bench_stream_from_host.json
tail -5 $O/bench_stream_resident.err
