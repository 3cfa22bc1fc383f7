#!/bin/bash
# One GPU-box session that produces everything the round's measured claims rest on (copied into profiles/rNN/ afterwards):
# parity tests, smoke, the bench lines (default, bp+fk+mf, stream, channel-sharded on one rank), rocprofv3 kernel statistics,
# PMC FETCH_SIZE / WRITE_SIZE passes, band-pass and pipeline timings, the CPU column at the config-1 shape.
#   gpurun --timeout 2400 -- 'bash scripts/evidence_run.sh gpurun_out/r02c'
set -u
O=${1:-gpurun_out/evidence}
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; grep "smoke" $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" > $O/bench_line.json; cut -c1-300 $O/bench_line.json
# round 6: the default line times the PUBLIC calls; the composition of rounds 1-5 (plan.apply_stats + _xcorr_device, no tail term) beside it
timeout 900 python bench.py --steps 20 --warmup 5 --api private --no-cpu --no-dense 2>/dev/null | grep "^{" > $O/bench_line_private_api.json; cut -c1-200 $O/bench_line_private_api.json
timeout 900 python bench.py --stages bp,fk,mf --steps 10 --warmup 3 --no-cpu --no-dense 2>/dev/null | grep "^{" > $O/bench_bp_fk_mf.json; cut -c1-200 $O/bench_bp_fk_mf.json
timeout 900 python bench.py --config stream --steps 10 --warmup 2 2>/dev/null | grep "^{" > $O/bench_stream_1gpu.json; cut -c1-600 $O/bench_stream_1gpu.json
# ... with the two detectors of a file on a side stream each (the default of round 5; since round 6 everything runs on one stream)
timeout 900 python bench.py --config stream --steps 10 --warmup 2 --detector-streams 2 --no-cpu 2>/dev/null | grep "^{" > $O/bench_stream_1gpu_detector_side_streams.json; cut -c1-300 $O/bench_stream_1gpu_detector_side_streams.json
# round 5: the chain with the detectors on side streams against itself on one stream (which result differs: nothing, since the
# cross-stream fence), and every stage beside the matrix-core STFT with the fence switched off (the hazard itself)
timeout 300 python scripts/probe/stream_race.py 2>/dev/null | grep "^{" | cut -c1-300 > $O/stream_race.txt; tail -3 $O/stream_race.txt
(D4W_HAZARD_FENCE=0 LOAD=5 NSTAGES=7 timeout 300 python scripts/probe/stream_race2.py) 2>/dev/null | grep "trials" > $O/stream_race2_fence_off.txt; cut -c1-160 $O/stream_race2_fence_off.txt
# round 6: the victims beside kernels of OTHER libraries on side streams (hipBLASLt / rocBLAS GEMMs, MIOpen conv), 10 trials per pair here
# (the 40-trial run is profiles/r06e)
for fam in "matmul f16" "matmul bf16" "conv2d"; do D4W_FOREIGN_ONLY="$fam" D4W_CONC_TRIALS=10 D4W_CONC_REPORT=$R/$O/concurrency_trials_foreign.txt timeout 600 python -u -m pytest tests/test_concurrent_gpu.py -q -m gpu -s -k foreign > /dev/null 2>&1; done; tail -3 $O/concurrency_trials_foreign.txt
# the same chain with the raw files in pinned host memory (double-buffered upload on a side stream): 8 files per run as above, and
# 24 files per run (the first upload and the drain weigh less: the steady-state rate against the PCIe bound)
timeout 900 python bench.py --config stream --from-host --steps 10 --warmup 2 2>/dev/null | grep "^{" > $O/bench_stream_from_host.json; cut -c1-300 $O/bench_stream_from_host.json
timeout 900 python bench.py --config stream --from-host --files 24 --steps 10 --warmup 2 2>/dev/null | grep "^{" > $O/bench_stream_from_host_24files.json; cut -c1-300 $O/bench_stream_from_host_24files.json
# matched filter: matrix-core kernel against the FFT kernel, errors against float64 (bench shape and the file shape)
(timeout 300 python scripts/time_xcorr_mm.py; NX=11020 NS=12000 timeout 300 python scripts/time_xcorr_mm.py) 2>/dev/null | grep "^{" > $O/time_xcorr_mm.txt; cut -c1-400 $O/time_xcorr_mm.txt
# round 5: templates beyond 241 samples on the matrix cores (the reference script's 450-sample template) against the direct form
(timeout 400 python scripts/time_xcorr_long.py; NX=11020 NS=12000 timeout 300 python scripts/time_xcorr_long.py) 2>/dev/null | grep "^{" > $O/time_xcorr_long.txt; cut -c1-300 $O/time_xcorr_long.txt
# round 5: the kernels one file of the stream chain launches in steady state (every non-d4w row would be glue)
timeout 300 python scripts/stream_kernels.py 2>/dev/null | grep -v "^$" > $O/stream_kernels.txt; head -3 $O/stream_kernels.txt
# what a notebook call costs: host float64 in -> host float64 out, upload / kernel / download
timeout 600 python scripts/time_host_api.py 2>/dev/null | grep "^{" > $O/time_host_api.txt; cat $O/time_host_api.txt
timeout 900 python bench.py --shard channel --steps 5 --warmup 2 --force-replicas 2>/dev/null | grep "^{" > $O/bench_shard_channel_1rank.json; cut -c1-200 $O/bench_shard_channel_1rank.json
timeout 600 python scripts/time_bp.py 2>/dev/null | grep "^{" > $O/time_bp.txt; cat $O/time_bp.txt
timeout 600 python scripts/pipeline_bench.py 2>/dev/null | grep "^{" > $O/pipeline_11020x12000.json; cat $O/pipeline_11020x12000.json
timeout 600 python scripts/time_shapes.py 13223x12000 8000x12000 11020x12000 5510x12000 4000x12000 2>/dev/null | grep "^{" > $O/time_shapes.txt; cat $O/time_shapes.txt
# pass order and pass times per mask (time-first / channel-first), bench shape and the scripts' own 13223-channel selection
(timeout 400 python scripts/time_fk_masks.py classic ninf hybrid dense step4; NX=13223 NS=12000 timeout 200 python scripts/time_fk_masks.py ninf classic; NX=11020 NS=12000 timeout 200 python scripts/time_fk_masks.py ninf classic) 2>/dev/null | grep "^{" > $O/time_fk_masks.txt; cut -c1-200 $O/time_fk_masks.txt
# shapes beyond the direct kernels: prime channel counts / record lengths (Bluestein forms), loop-free prime radices (generic kernels)
timeout 600 python scripts/time_shapes.py 4099x12000 10007x12000 19997x12000 11020x12014 11020x12002 10007x12014 2>/dev/null | grep "^{" > $O/time_any_shape.txt; cut -c1-60,180-330 $O/time_any_shape.txt
timeout 300 python scripts/time_bluestein_rows.py 2>/dev/null | grep "^{" > $O/time_bluestein_rows.txt; cat $O/time_bluestein_rows.txt
[ -x scripts/probe/lds_isolation ] && (for sz in "42608 38912" "42608 45056" "43008 43008"; do timeout 120 scripts/probe/lds_isolation $sz 30; done) > $O/lds_isolation.txt 2>&1
timeout 500 python scripts/probe/time_bz_band.py 2>/dev/null | grep "^{" > $O/time_bz_band.txt; cut -c1-200 $O/time_bz_band.txt
(NX=4000 timeout 250 python scripts/time_long_rows.py; NX=20000 timeout 250 python scripts/time_long_rows.py) 2>/dev/null | grep "^{" > $O/time_long_rows.txt
timeout 250 python scripts/time_fk_filt.py 2>/dev/null | grep "^{" > $O/time_fk_filt.txt
(timeout 300 python scripts/time_api_sweep.py; NX=11020 NS=12000 timeout 300 python scripts/time_api_sweep.py) 2>/dev/null | grep "^{" > $O/time_api_sweep.txt
NX=11020 timeout 600 python scripts/time_spectral.py 2>/dev/null | grep "^{" > $O/time_spectral_11020x12000.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o fk -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $R/$O/rocprof_bench.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && grep -E "d4w|Name" "$f" | cut -c1-160 | head -16
rm -rf $O/prof
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o fk -- python $R/bench.py --stages bp,fk,mf --steps 5 --warmup 2 --no-cpu --no-dense > $R/$O/rocprof_bench_bp.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_bp_fk_mf.csv
rm -rf $O/prof
PMC_SOURCE="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --stages bp,fk,mf (scripts/evidence_run.sh $O, scripts/pmc.sh)" BENCH_ARGS="--stages bp,fk,mf --no-dense" PMC_GROUPS="fetch write" bash scripts/pmc.sh $O/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
cp $O/pmc/summary.txt $O/pmc_fetch_write_summary.txt; cp $O/pmc/pmc_traffic.json $O/pmc_traffic.json; rm -rf $O/pmc
timeout 900 python scripts/cpu_baseline_c1.py > $O/cpu_baseline_c1.json 2>/dev/null; cut -c1-300 $O/cpu_baseline_c1.json
timeout 300 python scripts/time_picks3.py 2>/dev/null | grep -E "^\{|find_peaks" > $O/time_picks.txt; head -2 $O/time_picks.txt | cut -c1-300
timeout 300 python scripts/time_stft.py 2>/dev/null | grep "^{" > $O/time_stft.txt; cat $O/time_stft.txt
# round 4: detect.pick_times on raw long-row correlograms (windowed turning-point sweep), the Gabor image pipeline (filter2d on the matrix cores)
(timeout 300 python scripts/time_pick_long.py; NX=11020 NS=12000 timeout 200 python scripts/time_pick_long.py) 2>/dev/null | grep "^{" > $O/time_pick_long.txt; cut -c1-300 $O/time_pick_long.txt
(timeout 300 python scripts/time_image.py --nx 11020 --ns 12000; D4W_F2D_MM=0 timeout 300 python scripts/time_image.py --nx 11020 --ns 12000) 2>/dev/null | grep "^{" > $O/time_image.txt; cut -c1-300 $O/time_image.txt
timeout 300 python scripts/time_bp_parts.py 2>/dev/null | grep "^{" > $O/time_bp_parts.txt
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o st -- python $R/bench.py --config stream --files 8 --steps 16 --warmup 2 > $R/$O/rocprof_stream.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_stream.csv
rm -rf $O/prof
