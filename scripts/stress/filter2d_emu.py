import sys, ctypes, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from tests.emu_util import load_emu, vp
from tests.known_answers import filter2d_loops
import scipy.ndimage as ndi
emu = load_emu()
emu.d4w_filter2d_ws_bytes.restype = ctypes.c_size_t
rng = np.random.default_rng(int(sys.argv[1]))
worst = 0.0
for case in range(int(sys.argv[2])):
    h, w = int(rng.integers(1, 60)), int(rng.integers(1, 700))
    kh, kw = int(rng.integers(1, 40)), int(rng.integers(1, 114))
    img = (rng.standard_normal((h, w)) * rng.choice([1.0, 1e-4, 255.0]) + rng.choice([0.0, 100.0])).astype(np.float32)
    ker = (rng.standard_normal((kh, kw)) * rng.choice([1.0, 1e-3])).astype(np.float32)
    out = np.full((h, w), np.nan, dtype=np.float32)
    ws = np.empty(int(emu.d4w_filter2d_ws_bytes(kh, kw)) + 16, dtype=np.uint8)
    rc = emu.d4w_filter2d_f32(vp(img), h, w, vp(ker), kh, kw, vp(out), 0, vp(ws), None)
    if rc != 0:
        print("refused", (h, w, kh, kw), emu.d4w_last_error()); continue
    ref = filter2d_loops(img.astype(np.float64), ker.astype(np.float64)) if h * w * kh * kw < 3e6 else None
    if ref is None:
        continue
    e = float(np.max(np.abs(out - ref)) / max(np.max(np.abs(ref)), 1e-300))
    worst = max(worst, e)
    if not np.all(np.isfinite(out)) or e > 3e-6:
        print("BAD", (h, w, kh, kw), e)
print("worst", worst)
