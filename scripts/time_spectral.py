"""Timing of the row spectral operators on a resident 4000 x 12000 block (BASELINE configs[1]
geometry): HIP events around each C-ABI call, median of `reps`.  Prints one JSON line."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import das4whales_amd as dw  # noqa: E402
from das4whales_amd import dsp, detect  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    nx, ns, fs = int(os.environ.get("NX", 4000)), int(os.environ.get("NS", 12000)), 200.0
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn((nx, ns), device="cuda", generator=g)
    x = dsp.bp_filt(x, fs, 14, 30)
    out = {"shape": [nx, ns]}
    n = nx * ns
    ms = timed(lambda: dsp._analytic(x, 0))
    out["envelope_ms"] = ms
    out["envelope_GBps"] = 8 * n / ms / 1e6
    ms = timed(lambda: dsp.snr_tr_array(x, env=True))
    out["snr_env_ms"] = ms
    env = dsp._analytic(x, 0)
    thr = float(env.max()) * 0.45
    ms = timed(lambda: detect._find_peaks_device(env, thr), reps=5)
    out["find_peaks_ms(incl. host readback)"] = ms
    S = None

    def stft():
        nonlocal S
        S = dsp._stft_mag(x, 160, 8, 11, 24)[0]
    ms = timed(stft)
    out["stft_160_8_ms"] = ms
    out["stft_GBps"] = (4 * n + 4 * S.numel()) / ms / 1e6
    kernel = {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}
    ms = timed(lambda: detect.compute_cross_correlogram_spectrocorr(x, fs, [14., 30.], kernel, 0.8, 0.95), reps=5)
    out["spectrocorr_total_ms"] = ms
    ms = timed(lambda: dsp.get_fx(x[:, :400].contiguous(), 512))
    out["get_fx_512_ms"] = ms
    print(json.dumps(out))


if __name__ == "__main__":
    main()
