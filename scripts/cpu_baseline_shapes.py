"""CPU column of BASELINE.md section 3 at the two largest shapes the reference's form fits into host memory for --
20000 x 12000 and 4000 x 120000 (config 3's 20000 x 120000 needs >= 5 complex128 temporaries of 35.8 GiB each) -- plus
config 1's 4000 x 12000: the NumPy / SciPy float64 restatement (oracle/d4w_oracle.py; /root/reference is absent on the GPU
box), single thread as the reference runs, ONE timed run per stage after a warm-up on a small block, design time excluded;
the matched filter both as the batched port and in the reference's own row-loop form (detect.py:163-164) on a row sample.
One JSON line per shape; run once per round, kept under profiles/ (bench.py's cpu_baseline stays at config 1).
    python scripts/cpu_baseline_shapes.py [NXxNS ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import d4w_oracle as orc

fs, dx = 200.0, 2.0419046878814697
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4000, 12000), (20000, 12000), (4000, 120000)]


def once(fn):
    t0 = time.perf_counter(); fn(); return time.perf_counter() - t0


orc.fk_filter_filt(np.zeros((64, 256)), np.ones((64, 256)))            # imports, plan caches
for nx, ns in shapes:
    rng = np.random.default_rng(1234)
    x = rng.standard_normal((nx, ns))
    mask = np.ascontiguousarray(orc.hybrid_ninf_filter_design((nx, ns), [0, 4 * nx, 4], dx, fs, 1350., 1450., 3300, 3450, 14., 30.))
    t = np.arange(ns) / fs
    hf = orc.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
    lf = orc.gen_template_fincall(t, fs, 14.7, 21.8, 0.78)
    samples = float(nx) * ns
    res = {"shape": [nx, ns], "cores_available": len(os.sched_getaffinity(0)), "threads_used": 1, "runs": "one timed run per stage"}
    tf = once(lambda: orc.fk_filter_filt(x, mask))
    tb = once(lambda: orc.bp_filt(x, fs, 14, 30))
    tm = once(lambda: (orc.compute_cross_correlogram(x, hf), orc.compute_cross_correlogram(x, lf)))
    rows = min(nx, 400)
    tr = once(lambda: (orc.compute_cross_correlogram_reference_form(x[:rows], hf),
                       orc.compute_cross_correlogram_reference_form(x[:rows], lf))) * nx / rows
    res.update(fk_filter_filt_s=tf, bp_filt_s=tb, compute_cross_correlogram_x2_s=tm,
               compute_cross_correlogram_x2_reference_form_s=tr, reference_form_rows_timed=rows,
               channel_samples_per_s={"fk": samples / tf, "fk+mf": samples / (tf + tm), "fk+mf_reference_form": samples / (tf + tr),
                                      "bp+fk+mf": samples / (tb + tf + tm)})
    del mask
    print(json.dumps(res), flush=True)
