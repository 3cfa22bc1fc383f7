#!/bin/bash
# A/B of prebuilt library variants: das4whales_amd/lib/libd4w_<tag>.so (bench per tag, per-pass ms)
set -u
mkdir -p gpurun_out
fmt='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("ms/step %.2f  " % d["ms_per_step"], " ".join("%s=%.2f" % (k.replace("fk_pass",""), v) for k, v in r["kernel_ms"].items()),
              " ".join("%s=%.2f" % (k, v) for k, v in r["stage_ms"].items()), " plan", d["config"]["plan"])
    elif "Error" in l or "error" in l:
        print(l.rstrip())
'
cp das4whales_amd/lib/libd4w.so das4whales_amd/lib/libd4w_keep.so
for tag in "$@"; do
  cp das4whales_amd/lib/libd4w_$tag.so das4whales_amd/lib/libd4w.so
  echo "--- $tag ${ENVS:-} ${BENCH_ARGS:-}"
  env ${ENVS:-A=1} python bench.py --steps 5 --warmup 2 --no-cpu ${BENCH_ARGS:-} 2>&1 | python -c "$fmt"
done
cp das4whales_amd/lib/libd4w_keep.so das4whales_amd/lib/libd4w.so
