#!/bin/bash
# round 2, GPU call 1: slab sweep + new known-answer tests + existing f-k tests
mkdir -p gpurun_out
python scripts/slab_sweep.py > gpurun_out/slab_sweep.txt 2>&1
tail -40 gpurun_out/slab_sweep.txt
python -m pytest tests/test_fk_known_gpu.py tests/test_fk_gpu.py -x -q -m gpu -s > gpurun_out/pytest_fk.log 2>&1
tail -30 gpurun_out/pytest_fk.log
