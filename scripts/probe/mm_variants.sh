#!/bin/bash
# Probe builds of the matched filter for a same-box A/B (round 6: where the round-5 slowdown of xcorr_mm_rows comes from):
#   bash scripts/probe/mm_variants.sh build      (here, no GPU)
#   bash scripts/probe/mm_variants.sh run OUT    (GPU box: every variant + the round-4 tree in _ab/r4, alternating twice)
set -u
cd "$(dirname "$0")/../.."
case $1 in
build)
    bash scripts/probe/build_variant.sh mm_oldsplit xcorr_mm.hip -DD4W_MM_V_OLDSPLIT
    bash scripts/probe/build_variant.sh mm_nonan xcorr_mm.hip -DD4W_MM_V_NONAN
    bash scripts/probe/build_variant.sh mm_floatmean xcorr_mm.hip -DD4W_MM_V_FLOATMEAN
    bash scripts/probe/build_variant.sh mm_ch8192 xcorr_mm.hip -DD4W_MM_CH=8192
    bash scripts/probe/build_variant.sh mm_taildealt xcorr_mm.hip -DD4W_MM_V_TAIL_DEALT
    bash scripts/probe/build_variant.sh mm_tailnoscan xcorr_mm.hip -DD4W_MM_V_TAIL_NOSCAN ;;
build_tail)
    bash scripts/probe/build_variant.sh mm_taildealt xcorr_mm.hip -DD4W_MM_V_TAIL_DEALT
    bash scripts/probe/build_variant.sh mm_tailnoscan xcorr_mm.hip -DD4W_MM_V_TAIL_NOSCAN ;;
run)
    OUT=$2; mkdir -p $OUT
    fmt='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1], "two templates %.3f (min %.3f)  one template %.3f  no normalise %.3f  with tail %s / %s  err %s" % (
            d["mm_ms_median_min"][0], d["mm_ms_median_min"][1], d["mm_ms_one_template"][0], d["mm_ms_no_normalise"][0],
            ("%.3f" % d["mm_tail_ms_median_min"][0]) if "mm_tail_ms_median_min" in d else "-",
            ("%.3f" % d["mm_tail_ms_one_template"][0]) if "mm_tail_ms_one_template" in d else "-", d.get("mm_tail_err_vs_f64", d["mm_err_vs_f64"])))
'
    for rep in 1 2; do
        (cd _ab/r4 && timeout 300 python scripts/time_xcorr_mm.py 2>/dev/null | python -c "$fmt" "r4        rep $rep")
        for tag in ${TAGS:-base mm_oldsplit mm_nonan mm_floatmean mm_ch8192 mm_taildealt mm_tailnoscan}; do
            lib=$PWD/das4whales_amd/lib/probe/libd4w_$tag.so
            [ $tag = base ] && lib=$PWD/das4whales_amd/lib/libd4w.so
            [ -f $lib ] || continue
            D4W_LIB=$lib timeout 300 python -W ignore scripts/time_xcorr_mm.py 2>/dev/null | python -c "$fmt" "$(printf %-12s $tag) rep $rep"
        done
    done 2>&1 | tee $OUT/mm_variants.txt ;;
esac
