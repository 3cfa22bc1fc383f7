import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from tests.emu_util import load_emu, vp
from tests.test_emu_rowops import xcorr_mm_emu
emu = load_emu()
rng = np.random.default_rng(int(sys.argv[1]))
worst = 0.0
for case in range(int(sys.argv[2])):
    nx = int(rng.integers(1, 4)); ns = int(rng.choice([rng.integers(1, 64), rng.integers(64, 4200), rng.integers(4200, 13000)]))
    ntpl = int(rng.integers(1, 3))
    lens = [int(rng.integers(1, min(1200 if rng.random() < 0.3 else 242, max(2, ns + 1)))) for _ in range(ntpl)]
    taps = [rng.standard_normal(L) * rng.choice([1.0, 1e-3, 50.0]) for L in lens]
    sc_, of_ = rng.choice([1.0, 1e-6, 1e4]), rng.choice([0.0, 3.0, -1e3])
    x = rng.standard_normal((nx, ns)) * sc_ + of_
    if rng.random() < 0.3 and ns > 10:
        x[0, int(rng.integers(0, ns))] += 1e3 * np.abs(x).max()              # a spike: dynamic range
    norm = bool(rng.integers(0, 2))
    xf = np.ascontiguousarray(x, dtype=np.float32)
    ys = xcorr_mm_emu(emu, xf, taps, normalize=norm)
    xd = xf.astype(np.float64)
    if norm:
        mu = xd.mean(axis=1, keepdims=True); mx = np.abs(xd).max(axis=1, keepdims=True)
        xd = np.where(mx > 0, (xd - mu) / np.where(mx > 0, mx, 1), 0.0)
    for t, y in zip(taps, ys):
        tf = np.asarray(t, dtype=np.float32).astype(np.float64)
        ref = np.stack([np.correlate(np.concatenate((r, np.zeros(len(tf) - 1))), tf, "valid") for r in xd])
        scale = np.max(np.abs(ref), axis=1, keepdims=True); scale[scale == 0] = 1
        e = float(np.max(np.abs(y - ref) / scale))
        worst = max(worst, e)
        if not np.all(np.isfinite(y)) or e > 2e-6:
            print("BAD case", case, (nx, ns), lens, "norm", norm, "err", e, "scale", sc_, "offset", of_, "spike", bool(np.abs(xf).max() > 100 * (abs(of_) + sc_ * 6)))
print("worst", worst)
