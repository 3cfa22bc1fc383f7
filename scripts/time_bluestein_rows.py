"""Row operators whose transform length has a prime factor > 31 (Bluestein inside the workgroup's LDS): envelope of 12002-sample
rows, spectrogram correlation with a 148-sample window, f-k at 12002 samples."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import das4whales_amd as dw
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 3)
out = {}
x = torch.randn((11020, 12002), device="cuda")
out["envelope_11020x12002_ms"] = ev(lambda: dw.dsp.envelope(x))
x = torch.randn((11020, 12000), device="cuda")
out["envelope_11020x12000_ms"] = ev(lambda: dw.dsp.envelope(x))
ker = {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}
out["spectrocorr_win148_ms"] = ev(lambda: dw.detect.compute_cross_correlogram_spectrocorr(x, 200.0, [14., 30.], ker, 0.74, 0.95))
out["spectrocorr_win160_ms"] = ev(lambda: dw.detect.compute_cross_correlogram_spectrocorr(x, 200.0, [14., 30.], ker, 0.8, 0.95))
print(json.dumps(out))
