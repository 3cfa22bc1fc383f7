"""Matched filter with templates beyond 241 samples: the reference's own script correlates a 450-sample
template (scripts/main_mfdetect.py:70, gen_template_fincall(..., duration=2.25, window=False)).  The matrix-core form takes up
to 497 taps in one launch and longer templates in sections of 496 taps; HIP-event medians at NX x NS against the direct FIR
(the only form such templates had before round 5), errors of both against a float64 correlation of a few rows.  One JSON line
per template length."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from das4whales_amd import detect as ddet
nx, ns, fs = int(os.environ.get("NX", 20000)), int(os.environ.get("NS", 120000)), 200.0
lens = [int(v) for v in os.environ.get("LENS", "450,241,497,700,1024").split(",")]
gen = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((nx, ns), device="cuda", generator=gen) + 0.25
mean, mx = ddet._row_stats_cached(x)
t = np.arange(ns) / fs


def timeit(method, tl, n=8):
    ts = []
    for i in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ys = ddet._xcorr_device(x, tl, normalize=True, method=method, stats=(mean, mx))
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b)); del ys
    return float(np.median(ts[2:])), float(np.min(ts[2:]))


rows = [0, 1, nx // 2, nx - 1]
xs = x[rows].double().cpu().numpy()
xn = (xs - mean[rows].cpu().numpy()[:, None]) / mx[rows].double().cpu().numpy()[:, None]
for L in lens:
    if L == 450:      # the script's template: an un-windowed 2.25-s hyperbolic chirp
        tp = ddet._normalised_support(ddet.gen_template_fincall(t, fs, 15.0, 25.0, 2.25, window=False))
    else:
        tp = ddet._normalised_support(np.concatenate((np.random.default_rng(L).standard_normal(L), np.zeros(max(0, ns - L)))))
    L = len(tp)
    ref = np.stack([np.correlate(np.concatenate((r, np.zeros(L - 1))), tp, "valid") for r in xn])
    out = {"shape": [nx, ns], "support": L}
    for m in ("mm", "direct"):
        (y,) = ddet._xcorr_device(x, [tp], normalize=True, method=m, stats=(mean, mx))
        out[m + "_err_vs_f64"] = float(np.max(np.abs(y[rows].double().cpu().numpy() - ref)) / np.max(np.abs(ref)))
        del y
        out[m + "_ms_median_min"] = timeit(m, [tp], n=8 if m == "mm" else 4)
    out["mm_frac_of_hbm_8B_per_sample"] = 8.0 * nx * ns / 1e9 / (out["mm_ms_median_min"][0] * 1e-3) / 8000.0
    print(json.dumps(out), flush=True)
