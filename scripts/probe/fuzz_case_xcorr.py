"""Reproduce tests/test_fuzz_gpu.py::test_matched_filter_random_shapes with D4W_FUZZ_SEED=2: case (56, 19218, 19)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import das4whales_amd as dw
from oracle import d4w_oracle as orc
def rel(y, ref):
    return float(np.max(np.abs(np.asarray(y, dtype=np.float64) - ref)) / np.max(np.abs(ref)))
rng = np.random.default_rng(103 + 2)
for it in range(6):
    nx, ns = int(rng.integers(1, 200)), int(rng.integers(200, 30000))
    x = rng.standard_normal((nx, ns)) + 0.2
    if nx > 2:
        x[1] = 0.0
    L = int(rng.integers(2, min(161, ns // 2)))
    tpl = np.zeros(ns)
    tpl[:L] = rng.standard_normal(L) * np.hanning(L)
    keep = [r for r in range(nx) if np.any(x[r] != 0)]
    ref = orc.compute_cross_correlogram(x[keep], tpl)
    out = {}
    for meth in ("mm", "fft", "direct"):
        os.environ["D4W_XCORR_METHOD"] = meth
        for tail in (None, True, False):
            c = dw.detect.compute_cross_correlogram(x, tpl, exact_tail=tail)
            c = c.cpu().numpy() if hasattr(c, "cpu") else np.asarray(c)
            e = np.abs(c[keep] - ref) / np.max(np.abs(ref))
            out[(meth, tail)] = (float(e.max()), int(np.argmax(e.max(axis=0))), int(np.argmax(e.max(axis=1))))
    y = rng.standard_normal(ns)
    print((nx, ns, L), {k: ("%.2e" % v[0], v[1], v[2]) for k, v in out.items()})
