"""STFT magnitude kernel (n_fft 160, hop 8, kept bins 11..24, no row maximum -- the spectrogram-correlation detector's call)
on an 11020 x 12000 block: HIP events, median."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from das4whales_amd import dsp
nx, ns = int(os.environ.get("NX", 11020)), int(os.environ.get("NS", 12000))
x = torch.randn((nx, ns), device="cuda")
def ev(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
out = {"kept_bins_ms": ev(lambda: dsp._stft_mag(x, 160, 8, 11, 24, want_max=False)),
       "all_bins_rowmax_ms": ev(lambda: dsp._stft_mag(x, 160, 8, 11, 24, want_max=True)),
       "nfft256_hop12_ms": ev(lambda: dsp._stft_mag(x, 256, 12, 0, 128, want_max=True))}
print(json.dumps(out))
