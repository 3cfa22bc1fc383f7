#!/bin/bash
OUT=gpurun_out/r04b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -7 $OUT/pytest_gpu.log
timeout 300 python scripts/time_xcorr_mm.py >> $OUT/time_xcorr_mm.txt 2>&1
D4W_MM_WGS=3 timeout 300 python scripts/time_xcorr_mm.py >> $OUT/time_xcorr_mm.txt 2>&1
NX=11020 NS=12000 timeout 300 python scripts/time_xcorr_mm.py >> $OUT/time_xcorr_mm.txt 2>&1
grep -v amdgpu.ids $OUT/time_xcorr_mm.txt
