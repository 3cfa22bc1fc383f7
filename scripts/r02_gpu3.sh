#!/bin/bash
mkdir -p gpurun_out
python scripts/time_bp.py > gpurun_out/time_bp.txt 2>&1
cat gpurun_out/time_bp.txt | grep "^{"
python scripts/slab_sweep.py 0 > gpurun_out/slab_sweep0.txt 2>&1
grep "^{" gpurun_out/slab_sweep0.txt | cut -c1-200
python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
tail -15 gpurun_out/pytest_gpu.log
