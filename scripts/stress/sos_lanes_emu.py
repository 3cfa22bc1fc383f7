"""Adversarial random cases for the zero-phase recursion with the sections of a row on adjacent lanes (csrc/rowops.hip:
sos_pass_lanes, through d4w_sosfiltfilt_f32 with one exact segment per row, and d4w_sosfiltfilt_ends_f32) on the CPU test build
against scipy.signal.sosfiltfilt:  python scripts/stress/sos_lanes_emu.py SEED NCASES"""
import sys, ctypes, numpy as np, scipy.signal as sps
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from tests.emu_util import load_emu, vp
from tests.test_emu_rowops import sosfiltfilt_emu
emu = load_emu()
emu.d4w_sosfiltfilt_ends_ws_bytes.restype = ctypes.c_size_t
rng = np.random.default_rng(int(sys.argv[1]))
worst = 0.0
for case in range(int(sys.argv[2])):
    order = int(rng.integers(1, 11))
    kind = rng.choice(["bp", "lp", "hp"])
    fs = 200.0
    if kind == "bp":
        f0 = float(rng.uniform(8, 40)); wn = [f0 / 100, min(0.95, (f0 + float(rng.uniform(8, 40))) / 100)]
        if order > 5: order = order // 2 or 1            # a band-pass of order n has n sections
    else:
        wn = float(rng.uniform(0.1, 0.8))
        order = min(2 * order, 20)                       # n / 2 sections
    sos = np.ascontiguousarray(sps.butter(order, wn, kind, output="sos"))
    if sos.shape[0] > 10:
        continue
    nx = int(rng.integers(1, 40)); ns = int(rng.integers(8 * sos.shape[0] + 8, 1500))
    padlen = min(3 * (2 * sos.shape[0] + 1), ns - 1)
    x = rng.standard_normal((nx, ns)) * float(rng.choice([1.0, 1e-4, 1e3])) + float(rng.choice([0.0, 5.0, -300.0]))
    ref = sps.sosfiltfilt(sos, x.astype(np.float32).astype(np.float64), axis=1, padlen=padlen)
    y = sosfiltfilt_emu(emu, x, sos, padlen=padlen)
    scale = max(np.max(np.abs(ref)), 1e-30)
    e = float(np.max(np.abs(y - ref)) / scale)
    worst = max(worst, e)
    if not np.all(np.isfinite(y)) or e > 1e-5:
        print("BAD rows", case, (nx, ns), sos.shape[0], kind, wn, "err", e)
    if ns >= 4 * padlen + 8:                             # the two row-end pieces in place
        piece = int(rng.integers(padlen + 1, ns // 2)); keep = int(rng.integers(1, piece + 1))
        xf = np.ascontiguousarray(x, dtype=np.float32); yy = np.full_like(xf, 3.25)
        zi = np.ascontiguousarray(sps.sosfilt_zi(sos))
        ws = np.empty(emu.d4w_sosfiltfilt_ends_ws_bytes(nx, piece, padlen), dtype=np.uint8)
        assert emu.d4w_sosfiltfilt_ends_f32(vp(xf), vp(yy), nx, ns, vp(sos), vp(zi), sos.shape[0], padlen, piece, keep, 0, vp(ws), None) == 0
        L = sps.sosfiltfilt(sos, xf[:, :piece].astype(np.float64), axis=1, padlen=padlen)
        R = sps.sosfiltfilt(sos, xf[:, ns - piece:].astype(np.float64), axis=1, padlen=padlen)
        sc = max(np.max(np.abs(L)), np.max(np.abs(R)), 1e-30)
        e2 = max(float(np.max(np.abs(yy[:, :keep] - L[:, :keep])) / sc), float(np.max(np.abs(yy[:, ns - keep:] - R[:, piece - keep:])) / sc))
        worst = max(worst, e2)
        if e2 > 1e-5 or not np.all(yy[:, keep:ns - keep] == 3.25):
            print("BAD ends", case, (nx, ns), piece, keep, sos.shape[0], "err", e2)
print("worst", worst)
