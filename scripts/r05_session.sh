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
stream_kernels)
    timeout 600 python scripts/stream_kernels.py > $OUT/stream_kernels.txt 2>&1; head -60 $OUT/stream_kernels.txt ;;
bench_stream)
    timeout 900 python bench.py --config stream --no-cpu > $OUT/bench_stream_1gpu.json 2> $OUT/bench_stream.err; tail -c 1500 $OUT/bench_stream_1gpu.json; tail -3 $OUT/bench_stream.err ;;
tests_stream)
    timeout 1500 python -m pytest tests/test_stream_gpu.py tests/test_pipeline_gpu.py tests/test_spectral_gpu.py tests/test_rowops_gpu.py -x -q -m gpu 2>&1 | tail -30 > $OUT/pytest_stream.log; tail -12 $OUT/pytest_stream.log ;;
time_bp)
    timeout 600 python scripts/time_bp.py > $OUT/time_bp.txt 2>&1; D4W_SOS_LANES=0 timeout 600 python scripts/time_bp.py > $OUT/time_bp_lane_per_row.txt 2>&1
    cat $OUT/time_bp.txt; echo "-- D4W_SOS_LANES=0:"; cat $OUT/time_bp_lane_per_row.txt ;;
tests_bp)
    timeout 1500 python -m pytest tests/test_rowops_gpu.py tests/test_stream_gpu.py tests/test_fuzz_gpu.py tests/test_reference_suite_gpu.py -x -q -m gpu 2>&1 | tail -30 > $OUT/pytest_bp.log; tail -8 $OUT/pytest_bp.log ;;
prof_bp)
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_bp -o bp -- python $OLDPWD/scripts/prof_bp.py > $OLDPWD/$OUT/prof_bp.log 2>&1)
    f=$(find $OUT/prof_bp -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_bp_file_shape.csv 2>/dev/null; head -12 $OUT/kernel_stats_bp_file_shape.csv | cut -c1-160; rm -rf $OUT/prof_bp ;;
pmc_mfma)
    # matrix-core pipe counters of the three kernels that use it: two-template correlator (bench shape), the
    # 16-step one-template correlator (450 taps), the detector's STFT (stream), filter2d of the Gabor detector
    PMC_GROUPS=mfma BENCH_ARGS="--stages mf" bash scripts/pmc.sh $OUT/pmc_mfma_xcorr > /dev/null 2>&1; cp $OUT/pmc_mfma_xcorr/summary.txt $OUT/pmc_mfma_xcorr.txt
    PMC_GROUPS=mfma PMC_CMD="env LENS=450 python $GRAFT_REPO_ROOT/scripts/time_xcorr_long.py" bash scripts/pmc.sh $OUT/pmc_mfma_xcorr_long > /dev/null 2>&1; cp $OUT/pmc_mfma_xcorr_long/summary.txt $OUT/pmc_mfma_xcorr_long.txt
    PMC_GROUPS=mfma PMC_CMD="env FILES=5 python $GRAFT_REPO_ROOT/scripts/stream_kernels.py" bash scripts/pmc.sh $OUT/pmc_mfma_stream > /dev/null 2>&1; cp $OUT/pmc_mfma_stream/summary.txt $OUT/pmc_mfma_stream.txt
    PMC_GROUPS=mfma PMC_CMD="python $GRAFT_REPO_ROOT/scripts/time_image.py" bash scripts/pmc.sh $OUT/pmc_mfma_image > /dev/null 2>&1; cp $OUT/pmc_mfma_image/summary.txt $OUT/pmc_mfma_image.txt
    rm -rf $OUT/pmc_mfma_xcorr $OUT/pmc_mfma_xcorr_long $OUT/pmc_mfma_stream $OUT/pmc_mfma_image
    grep -h -A 6 "xcorr_mm_rows\|stft_mm_rows\|filter2d_mm_rows" $OUT/pmc_mfma_*.txt | head -80 ;;
seed_shifts)
    # the whole GPU suite on shifted seeds (tests/conftest.py: D4W_SEED_SHIFT): different random cases through the same tests
    for k in ${SHIFTS:-1 2 3 4}; do
        D4W_SEED_SHIFT=$k timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $OUT/pytest_gpu_seed_shift_$k.log
        echo "shift $k: $(grep -E "passed|failed|error" $OUT/pytest_gpu_seed_shift_$k.log | tail -1)"
    done ;;
*) echo "unknown step $what" ;;
esac
done
