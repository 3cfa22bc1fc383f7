#!/bin/bash
# A/B of prebuilt library variants: das4whales_amd/lib/libd4w_<tag>.so
set -u
mkdir -p gpurun_out
fmt='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("ms/step %.2f  " % d["ms_per_step"], " ".join("%s=%.2f" % (k.replace("fk_pass",""), v) for k, v in r["kernel_ms"].items()), " plan", d["config"]["plan"])
'
for tag in "$@"; do
  cp das4whales_amd/lib/libd4w_$tag.so das4whales_amd/lib/libd4w.so
  for plan in "" "50,400,15,4000,8,16" "25,800,15,4000,16,8"; do
    echo "--- $tag plan=$plan ${ENVS:-}"
    env ${ENVS:-A=1} python bench.py --steps 5 --warmup 2 --no-cpu ${plan:+--plan $plan} 2>&1 | python -c "$fmt"
  done
done
