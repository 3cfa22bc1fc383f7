import sys, ctypes, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from tests.emu_util import load_emu, vp
from oracle import d4w_oracle as orc
emu = load_emu()
rng = np.random.default_rng(int(sys.argv[1]))
worst = 0.0
for case in range(int(sys.argv[2])):
    n_fft = int(rng.choice([32, 64, 96, 128, 160])); hop = int(rng.choice([8, 16, 24, 32]))
    lo = int(rng.integers(0, n_fft // 2)); hi = min(n_fft // 2, lo + int(rng.integers(0, 16)))
    ns = int(rng.choice([rng.integers(n_fft, 600), rng.integers(600, 9000)]))
    if not emu.d4w_stft_mm_eligible(n_fft, hop, lo, hi):
        continue
    nx = int(rng.integers(1, 4))
    x = rng.standard_normal((nx, ns)) * rng.choice([1.0, 1e-5, 1e4]) + rng.choice([0.0, 10.0])
    if rng.random() < 0.3:
        x[0, int(rng.integers(0, ns))] += 300 * np.abs(x).max()
    xf = np.ascontiguousarray(x, dtype=np.float32)
    nt = emu.d4w_stft_frames(ns, hop)
    S = np.full((nx, hi - lo + 1, nt), np.nan, dtype=np.float32)
    rc = emu.d4w_stft_mag_f32(vp(xf), vp(S), None, nx, ns, n_fft, hop, lo, hi, None)
    assert rc == 0, emu.d4w_last_error()
    for c in range(nx):
        ref = np.abs(orc.librosa_stft(xf[c].astype(np.float64), n_fft=n_fft, hop_length=hop))
        e = float(np.max(np.abs(S[c] - ref[lo:hi + 1])) / np.abs(ref).max())
        worst = max(worst, e)
        if not np.all(np.isfinite(S[c])) or e > 3e-6:
            print("BAD", (n_fft, hop, lo, hi, ns), c, e)
print("worst", worst)
