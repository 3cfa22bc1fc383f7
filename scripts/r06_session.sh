#!/bin/bash
# Round-6 GPU sessions (gpurun -- 'bash scripts/r06_session.sh <name> <what...>'); results under gpurun_out/<name>/
set -u
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for what in "$@"; do
case $what in
foreign)
    # the victims of the cross-stream hazard beside kernels of OTHER libraries (hipBLASLt / rocBLAS GEMMs, MIOpen
    # conv), one neighbour family per process (a crash inside one library must not take the others' rows with it)
    rm -f $OUT/concurrency_trials.txt
    for fam in "matmul f16" "matmul bf16" "conv2d"; do
        tag=$(echo $fam | tr ' ' '_')
        D4W_FOREIGN_ONLY="$fam" D4W_CONC_TRIALS=${TRIALS:-40} D4W_CONC_REPORT=$R/$OUT/concurrency_trials.txt PYTHONFAULTHANDLER=1 \
            timeout 900 python -u -m pytest tests/test_concurrent_gpu.py -q -m gpu -s -k foreign > $OUT/pytest_foreign_$tag.log 2>&1
        echo "foreign neighbours '$fam': rc $? $(grep -E "passed|failed|error" $OUT/pytest_foreign_$tag.log | tail -1)"
    done
    cat $OUT/concurrency_trials.txt ;;
concurrent)
    D4W_CONC_TRIALS=${TRIALS:-40} D4W_CONC_REPORT=$R/$OUT/concurrency_trials_own.txt timeout 1500 python -m pytest tests/test_concurrent_gpu.py -q -m gpu -s -k "not foreign" 2>&1 | tail -30 > $OUT/pytest_concurrent.log
    tail -8 $OUT/pytest_concurrent.log ;;
ab_r4)
    # the round-4 library (_ab/r4 = 35dbda3, built here) against HEAD on ONE box, alternating three times
    for i in 1 2 3; do
        for tree in _ab/r4 .; do
            tag=$([ $tree = . ] && echo HEAD || echo r4)
            (cd $tree && timeout 300 python scripts/time_xcorr_mm.py 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$tag pass $i  xcorr_mm two templates %.3f ms (min %.3f)  one template %.3f  no normalise %.3f  fft form %.3f' % (d['mm_ms_median_min'][0], d['mm_ms_median_min'][1], d['mm_ms_one_template'][0], d['mm_ms_no_normalise'][0], d['fft_ms_median_min'][0]))")
            (cd $tree && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-dense 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$tag pass $i  bench ms/step %.3f  frac %.3f ' % (d['ms_per_step'], r['frac']), ' '.join('%s=%.2f' % (k.replace('fk_pass', ''), v) for k, v in r.get('kernel_ms', {}).items()), ' '.join('%s=%.2f' % (k, v) for k, v in r.get('stage_ms', {}).items()))")
        done
    done 2>&1 | tee $OUT/ab_r4_vs_head.txt ;;
copy_ceiling)
    timeout 600 scripts/probe/copy_ceiling > $OUT/copy_ceiling.txt 2>&1; grep -c TB $OUT/copy_ceiling.txt; sort -t'|' -k8 $OUT/copy_ceiling.txt | grep "^copy" | sort -k2 -t'|' | awk -F'|' '{print}' | sort -t'|' -k8 -r | head -12 ;;
fk_policy)
    # cache-policy bits of the f-k passes' block accesses (csrc/d4w_internal.h: D4W_FK_LD / D4W_FK_ST): variant builds of the library
    # (scripts/probe/build_variant.sh fknt|fkntld|fkntst fk_filter.hip -DD4W_FK_LD=1 ...) against the packaged one, same box
    for rep in 1 2; do
      for tag in base fknt fkntld fkntst; do
        lib=$R/das4whales_amd/lib/probe/libd4w_$tag.so
        [ $tag = base ] && lib=$R/das4whales_amd/lib/libd4w.so
        [ -f $lib ] || continue
        D4W_LIB=$lib timeout 300 python -W ignore scripts/time_fk_masks.py classic ninf dense 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('$tag rep $rep %-8s %-13s total %6.2f ms  passes %s  frac %.3f' % (d['mask'], d['order'], d['total_ms'], d['ms'], d['frac_24B']))"
      done
    done 2>&1 | tee $OUT/fk_policy_ab.txt ;;
splitk)
    # the foreign neighbour that moved the overlap-save kernels' results (r06f): which knob makes it go away
    (timeout 200 python scripts/probe/splitk_neighbour.py
     D4W_XF_LDS_CLAIM=159 timeout 300 python scripts/probe/splitk_neighbour.py
     D4W_XF_LDS_CLAIM=80 timeout 300 python scripts/probe/splitk_neighbour.py
     SHAPE=256,8192,256 timeout 200 python scripts/probe/splitk_neighbour.py
     SHAPE=1024,32768,1024 timeout 200 python scripts/probe/splitk_neighbour.py
     DT=bf16 D4W_XF_LDS_CLAIM=159 timeout 300 python scripts/probe/splitk_neighbour.py) 2>/dev/null | grep "^{" | tee $OUT/splitk_neighbour.txt
    # what the library runs for that product
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_sk -o sk -- python -c "
import torch
a = torch.randn((256, 32768), device='cuda').half(); b = torch.randn((32768, 256), device='cuda').half()
for _ in range(5): c = torch.matmul(a, b)
torch.cuda.synchronize()" > /dev/null 2>&1)
    f=$(find $OUT/prof_sk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -8 | tee $OUT/splitk_kernels.txt; rm -rf $OUT/prof_sk ;;
claimed)
    # the overlap-save kernels claim the CU's LDS (xcorr_fft_blocks SUBS = 4, 159 KiB): the split-K neighbour again, the library's own
    # neighbours WITHOUT the cross-stream fence, and what the band-pass costs now
    (timeout 200 python scripts/probe/splitk_neighbour.py; DT=bf16 timeout 200 python scripts/probe/splitk_neighbour.py
     D4W_XF_LDS_CLAIM=0 timeout 200 python scripts/probe/splitk_neighbour.py) 2>/dev/null | grep "^{" | tee $OUT/splitk_neighbour_claimed.txt
    D4W_HAZARD_FENCE=0 D4W_CONC_TRIALS=${TRIALS:-40} timeout 1200 python -u -m pytest tests/test_concurrent_gpu.py -q -m gpu -s -k "other_streams" 2>&1 | tail -5 | tee $OUT/pytest_concurrent_fence_off.log
    (timeout 600 python scripts/time_bp.py; D4W_XF_LDS_CLAIM=0 timeout 600 python scripts/time_bp.py) 2>/dev/null | grep "^{" | tee $OUT/time_bp_claimed_then_unclaimed.txt ;;
bp_forms)
    # band-pass with the LDS-claiming kernel: sub-block barriers, one group per workgroup (default) against the walking form (probe build)
    (timeout 600 python scripts/time_bp.py; D4W_LIB=$R/das4whales_amd/lib/probe/libd4w_xfloop.so timeout 600 python -W ignore scripts/time_bp.py
     D4W_LIB=$R/das4whales_amd/lib/probe/libd4w_xfspin.so timeout 600 python -W ignore scripts/time_bp.py) 2>/dev/null | grep "^{" | cut -c1-160 | tee $OUT/time_bp_forms.txt
    (timeout 200 python scripts/probe/splitk_neighbour.py; D4W_LIB=$R/das4whales_amd/lib/probe/libd4w_xfloop.so timeout 200 python -W ignore scripts/probe/splitk_neighbour.py) 2>/dev/null | grep "^{" | cut -c1-420 | tee $OUT/splitk_neighbour_forms.txt ;;
mm_split)
    # two templates: the wave-split kernel at three workgroups per CU (default) against the kernel of rounds 4-5 (D4W_MM_FUSED=2)
    for rep in 1 2; do for f in 3 2; do D4W_MM_FUSED=$f timeout 300 python scripts/time_xcorr_mm.py 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('D4W_MM_FUSED=$f rep $rep two templates %.3f (min %.3f)  no normalise %.3f  with tail %.3f (min %.3f)  err %s tail err %s' % (d['mm_ms_median_min'][0], d['mm_ms_median_min'][1], d['mm_ms_no_normalise'][0], d['mm_tail_ms_median_min'][0], d['mm_tail_ms_median_min'][1], d['mm_err_vs_f64'], d['mm_tail_err_vs_f64']))"; done; done | tee $OUT/mm_wave_split.txt
    (NX=11020 NS=12000 D4W_MM_FUSED=3 timeout 300 python scripts/time_xcorr_mm.py; NX=11020 NS=12000 D4W_MM_FUSED=2 timeout 300 python scripts/time_xcorr_mm.py) 2>/dev/null | grep "^{" | cut -c1-700 | tee -a $OUT/mm_wave_split.txt ;;
bp_now)
    timeout 600 python scripts/time_bp.py 2>/dev/null | grep "^{" | cut -c1-170 | tee $OUT/time_bp.txt
    (timeout 200 python scripts/probe/splitk_neighbour.py; DT=bf16 timeout 200 python scripts/probe/splitk_neighbour.py) 2>/dev/null | grep "^{" | cut -c1-420 | tee $OUT/splitk_neighbour.txt ;;
mm_wgs)
    # how much the third workgroup per CU is worth to the one-template correlator (the two-template one has registers for two)
    for w in 3 2 1; do D4W_MM_WGS=$w timeout 300 python scripts/time_xcorr_mm.py 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('D4W_MM_WGS=$w  two templates %.3f  one template %.3f  tail two %.3f  tail one %.3f' % (d['mm_ms_median_min'][0], d['mm_ms_one_template'][0], d['mm_tail_ms_median_min'][0], d['mm_tail_ms_one_template'][0]))"; done | tee $OUT/mm_wgs.txt ;;
mm_variants)
    bash scripts/probe/mm_variants.sh run $OUT ;;
tickets)
    timeout 600 scripts/probe/copy_ceiling tickets > $OUT/copy_ceiling_tickets.txt 2>&1; cat $OUT/copy_ceiling_tickets.txt ;;
tests_all)
    timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log ;;
smoke)
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -5 $OUT/smoke.log ;;
tests_mf)
    timeout 1800 python -u -m pytest tests/test_rowops_gpu.py tests/test_fuzz_gpu.py tests/test_fk_gpu.py tests/test_stream_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -40 > $OUT/pytest_mf.log; tail -25 $OUT/pytest_mf.log ;;
bench)
    timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench_line.json ;;
bench_nocpu)
    timeout 900 python bench.py --no-cpu > $OUT/bench_line_nocpu.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench_line_nocpu.json ;;
api_sweep)
    (timeout 300 python scripts/time_api_sweep.py; NX=11020 NS=12000 timeout 300 python scripts/time_api_sweep.py) 2>/dev/null | grep "^{" > $OUT/time_api_sweep.txt; cat $OUT/time_api_sweep.txt ;;
xcorr_mm)
    timeout 600 python scripts/time_xcorr_mm.py > $OUT/time_xcorr_mm.txt 2>&1; cat $OUT/time_xcorr_mm.txt ;;
envelope_ab)    # (D4W_AN_PAIR exists only in a build with scripts/probe/analytic_pair.h pasted in: the build profiles/r06q measured)
    # the envelope at the file shape -- analytic_rows (one row's M-point transform, scalar butterflies) against
    # analytic_rows_pair (the row's two M/2-point sub-transforms in packed registers), same process order alternated, plus thread counts
    for i in 1 2; do
        for pair in 0 1; do
            echo "D4W_AN_PAIR=$pair pass $i: $(D4W_AN_PAIR=$pair NX=11020 NS=12000 timeout 300 python scripts/time_spectral.py 2>/dev/null | grep '^{' | cut -c1-330)"
        done
    done 2>&1 | tee $OUT/envelope_ab.txt
    for thr in 256 320 384 512; do
        echo "pair, $thr threads: $(D4W_AN_THREADS=$thr NX=11020 NS=12000 timeout 300 python scripts/time_spectral.py 2>/dev/null | grep '^{' | cut -c1-120)"
    done 2>&1 | tee -a $OUT/envelope_ab.txt ;;
spectral_tests)
    timeout 1800 python -u -m pytest tests/test_spectral_gpu.py tests/test_stream_gpu.py tests/test_pipeline_gpu.py tests/test_reference_suite_gpu.py tests/test_image_gpu.py -x -q -m gpu 2>&1 | tail -30 > $OUT/pytest_spectral.log; tail -8 $OUT/pytest_spectral.log ;;
stream)
    timeout 900 python bench.py --config stream --steps 10 --warmup 2 --no-cpu 2>/dev/null | grep "^{" > $OUT/bench_stream_1gpu.json; cut -c1-400 $OUT/bench_stream_1gpu.json
    timeout 300 python scripts/stream_kernels.py 2>/dev/null | grep -v "^$" > $OUT/stream_kernels.txt; head -30 $OUT/stream_kernels.txt ;;
valu_rate)
    timeout 120 scripts/ubench/valu_rate > $OUT/valu_rate.txt 2>&1; cat $OUT/valu_rate.txt ;;
envelope_pmc)
    # SQ counters of the two envelope kernels (own rocprofv3 passes, --kernel-trace only)
    for pair in 0 1; do
        i=0
        for g in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
            i=$((i + 1))
            (cd /tmp && D4W_AN_PAIR=$pair timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/$OUT/pmc_env/p${pair}_$i -o pmc -- python $R/scripts/time_env.py > /dev/null 2>&1)
        done
    done
    python - <<PY | tee $OUT/pmc_envelope.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_env/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "analytic_rows" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    for c, v in sorted(acc[k].items()):
        print("%-42s %-24s n=%d avg=%.4g" % (k, c, len(v), sum(v) / len(v)))
PY
    rm -rf $OUT/pmc_env ;;
*) echo "unknown step $what" ;;
esac
done
