# A/B of the run length of the statistics-carrying pass A' (D4W_FK_RUN_A)
export TMPDIR=/tmp
for r in ${RUNS:-10 15 25 50 75}; do
  echo "RUN_A=$r"
  D4W_FK_RUN_A=$r timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-dense 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), round(r['kernel_ms']['fk_passA_inv'],3), {k:round(v,3) for k,v in r['stage_ms'].items()})"
done
