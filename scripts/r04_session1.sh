#!/bin/bash
# round 4, first GPU session: parity of the matrix-core matched filter, its timing against the FFT kernel, LDS-DMA copy probe
OUT=gpurun_out/r04a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for w in 2 3 4; do D4W_MM_WGS=$w timeout 300 python scripts/time_xcorr_mm.py >> $OUT/time_xcorr_mm.txt 2>&1; done
NX=11020 NS=12000 timeout 300 python scripts/time_xcorr_mm.py >> $OUT/time_xcorr_mm.txt 2>&1
NX=11020 NS=12000 D4W_MM_WGS=2 timeout 300 python scripts/time_xcorr_mm.py >> $OUT/time_xcorr_mm.txt 2>&1
cat $OUT/time_xcorr_mm.txt
timeout 300 scripts/probe/dma_probe > $OUT/dma_probe.txt 2>&1; cat $OUT/dma_probe.txt
