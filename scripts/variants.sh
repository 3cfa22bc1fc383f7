#!/bin/bash
# bench the f-k filter under several env settings: each argument is "VAR=val VAR2=val2"
set -u
fmt='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("ms/step %.2f  " % d["ms_per_step"], " ".join("%s=%.2f" % (k.replace("fk_pass",""), v) for k, v in r["kernel_ms"].items()), d["config"]["plan"])
'
for e in "$@"; do
  echo "--- $e"
  env $e python bench.py --steps 5 --warmup 2 --no-cpu --stages fk 2>&1 | python -c "$fmt"
done
