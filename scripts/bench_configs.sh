# BASELINE configs[1] (4000 x 12000: band-pass + f-k [+ matched filter]) and the real OOI channel count
# (11020 x 12000, generic kernels with prime radices 19 and 29), plus configs[2] with a dense mask
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"][:60], "| ms/step", round(d["ms_per_step"],3), "| G samples/s", round(d["value"]/1e9,2), "|", {k:round(v,3) for k,v in r["stage_ms"].items()}, "| cpu", d.get("cpu_baseline",{}).get("value"))'
timeout 600 python bench.py --nx 4000 --ns 12000 --stages bp,fk --steps 20 --warmup 5 --cpu-sample 4000x12000 2>&1 | grep "^{" | tee gpurun_out/bench_4000x12000_bp_fk.json | python -c "$P"
timeout 600 python bench.py --nx 4000 --ns 12000 --stages bp,fk,mf --steps 20 --warmup 5 --no-cpu 2>&1 | grep "^{" | tee gpurun_out/bench_4000x12000_bp_fk_mf.json | python -c "$P"
timeout 600 python bench.py --nx 11020 --ns 12000 --stages bp,fk,mf --steps 20 --warmup 5 --no-cpu 2>&1 | grep "^{" | tee gpurun_out/bench_11020x12000_bp_fk_mf.json | python -c "$P"
