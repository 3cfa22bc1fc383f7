#!/bin/bash
# Round-5 GPU sessions (gpurun -- 'bash scripts/r05_session.sh <name> <what...>'); results under gpurun_out/<name>/
set -u
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
for what in "$@"; do
case $what in
tests_mf)
    timeout 1500 python -m pytest tests/test_rowops_gpu.py tests/test_fuzz_gpu.py tests/test_fk_gpu.py tests/test_stream_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -s 2>&1 | tail -60 > $OUT/pytest_mf.log; tail -25 $OUT/pytest_mf.log ;;
tests_all)
    timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log ;;
xcorr_long)
    timeout 600 python scripts/time_xcorr_long.py > $OUT/time_xcorr_long.txt 2>&1
    NX=11020 NS=12000 timeout 600 python scripts/time_xcorr_long.py >> $OUT/time_xcorr_long.txt 2>&1; cat $OUT/time_xcorr_long.txt ;;
xcorr_mm)
    timeout 600 python scripts/time_xcorr_mm.py > $OUT/time_xcorr_mm.txt 2>&1; cat $OUT/time_xcorr_mm.txt ;;
bench)
    timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench_line.json ;;
bench_nocpu)
    timeout 900 python bench.py --no-cpu > $OUT/bench_line_nocpu.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench_line_nocpu.json ;;
smoke)
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -5 $OUT/smoke.log ;;
*) echo "unknown step $what" ;;
esac
done
