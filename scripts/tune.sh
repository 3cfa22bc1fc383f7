#!/bin/bash
# Plan / block-size sweep on the GPU box; prints per-pass milliseconds per variant.
set -u
mkdir -p gpurun_out
run() { echo "--- $*"; env "$@" python bench.py --steps 5 --warmup 2 --no-cpu ${PLAN:+--plan $PLAN} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('ms/step %.2f  ' % d['ms_per_step'], ' '.join('%s=%.2f' % (k.replace('fk_pass',''), v) for k, v in r['kernel_ms'].items()), ' plan', d['config']['plan'])
"; }
PLAN="" run A=1
PLAN="" run D4W_FK_THREADS_A=512 D4W_FK_THREADS_B=512 D4W_FK_THREADS_C=512
PLAN="50,400,15,4000,8,16" run D4W_FK_THREADS_A=512 D4W_FK_THREADS_B=512 D4W_FK_THREADS_C=512
PLAN="25,800,15,4000,16,8" run D4W_FK_THREADS_A=512 D4W_FK_THREADS_B=512 D4W_FK_THREADS_C=512
PLAN="40,500,30,2000,4,16" run D4W_FK_THREADS_A=512 D4W_FK_THREADS_B=256 D4W_FK_THREADS_C=512
PLAN="80,250,15,4000,4,16" run D4W_FK_THREADS_A=512 D4W_FK_THREADS_B=512 D4W_FK_THREADS_C=256
PLAN="40,500,15,4000,8,8" run D4W_FK_THREADS_A=512 D4W_FK_THREADS_B=512 D4W_FK_THREADS_C=256
