# band-pass stage time vs the segmentation model's resident-wave slots (D4W_SOS_SLOTS)
export TMPDIR=/tmp
for shape in "4000 12000" "20000 120000"; do
  set -- $shape
  for slots in ${SLOTS:-2048 4096 8192 16384}; do
    D4W_SOS_SLOTS=$slots timeout 600 python bench.py --nx $1 --ns $2 --stages bp --steps 10 --warmup 3 --no-cpu 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2 slots $slots: bp ms', round(d['roofline']['stage_ms']['bp_sosfiltfilt'],3))"
  done
done
