import ctypes, sys, numpy as np, scipy.signal as sps
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from tests.emu_util import load_emu, vp
emu = load_emu()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    kind = rng.integers(0, 7)
    ns = int(rng.choice([rng.integers(3, 300), rng.integers(300, 16384), rng.integers(16385, 60000)]))
    if rng.random() < 0.5:
        ns = (ns // 4) * 4 or 4
    nx = int(rng.integers(1, 4))
    t = np.arange(ns)
    if kind == 0:
        x = rng.standard_normal((nx, ns))
    elif kind == 1:
        x = np.round(rng.standard_normal((nx, ns)) * rng.choice([1, 2, 4])) / rng.choice([1, 2, 4])       # plateaus
    elif kind == 2:
        x = np.cumsum(rng.choice([-1.0, 0.0, 1.0], size=(nx, ns)), axis=1)                                 # integer random walk
    elif kind == 3:
        x = np.sin(t * rng.uniform(0.01, 1.5))[None, :] * (1 + 0.3 * rng.standard_normal((nx, 1))) + 1e-3 * rng.standard_normal((nx, ns))
    elif kind == 4:
        x = np.abs(sps.hilbert(rng.standard_normal((nx, ns)), axis=1))
    elif kind == 5:
        x = np.tile(rng.standard_normal((nx, 7)), (1, ns // 7 + 1))[:, :ns]                                # periodic, many equal maxima
    else:
        x = np.where(rng.random((nx, ns)) < 0.01, rng.standard_normal((nx, ns)) * 10, 0.0)                   # sparse spikes on flat zero
    x = np.ascontiguousarray(x, dtype=np.float32)
    span = float(x.max() - x.min()) or 1.0
    for thr in (0.0, span * float(rng.uniform(0.01, 0.3)), span * float(rng.uniform(0.3, 1.1))):
        cap = ns // 2 + 1
        idx = np.empty((nx, cap), dtype=np.int32); cnt = np.empty(nx, dtype=np.int32)
        rc = emu.d4w_find_peaks_f32(vp(x), nx, ns, ctypes.c_double(thr), vp(idx), vp(cnt), cap, None)
        assert rc == 0, emu.d4w_last_error()
        for c in range(nx):
            ref = sps.find_peaks(x[c].astype(np.float64), prominence=thr)[0]
            if cnt[c] != len(ref) or not np.array_equal(idx[c, :cnt[c]], ref):
                bad += 1
                print("MISMATCH case", case, "kind", kind, "ns", ns, "thr", thr, "row", c, "got", cnt[c], "ref", len(ref))
print("done, mismatches:", bad)
