"""Where find_peaks_prom spends its cycles at the file shape (probe build, scripts/probe/fp_timing.sh): mean cycles of
thread 0 per workgroup between the phase marks, for a few thresholds."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from das4whales_amd import dsp, detect
from das4whales_amd._lib import lib, check
nx, ns = int(os.environ.get("NX", 11020)), int(os.environ.get("NS", 12000))
torch.manual_seed(0)
x = dsp.bp_filt(torch.randn((nx, ns), device="cuda"), 200.0, 14, 30)
t = np.arange(ns) / 200.0
c = detect.compute_cross_correlogram(x, detect.gen_template_fincall(t, 200.0, 17.8, 28.8, 0.68))
env = dsp._analytic(c, 0)
cmax = float(c.max())
cap = 1024
idx = torch.empty((nx, cap), dtype=torch.int32, device="cuda"); cnt = torch.empty(nx, dtype=torch.int32, device="cuda")
lib.d4w_fp_timing_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["stage+summaries1", "summaries2", "fp_scan total", "emit", "  mark sweep", "  count+list", "  cfail clear", "  own walk",
         "  wait longest walk", "  accept"]
buf = (ctypes.c_ulonglong * 16)()
for label, thr in (("thr=inf", 1e30), ("thr=0.45 max(c)", 0.45 * cmax), ("thr=0.2 max(c)", 0.2 * cmax), ("thr=0", 0.0)):
    cc = cap if thr > 0 else ns // 2 + 1
    idx = torch.empty((nx, cc), dtype=torch.int32, device="cuda")
    run = lambda: check(lib.d4w_find_peaks_f32(env.data_ptr(), nx, ns, float(thr), idx.data_ptr(), cnt.data_ptr(), cc, None))
    run(); lib.d4w_fp_timing_read(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); b.synchronize()
    lib.d4w_fp_timing_read(buf, 1)
    print("%s: %.3f ms, %d picks; shader-clock cycles per workgroup:" % (label, a.elapsed_time(b), int(cnt.sum())))
    for k, n in enumerate(names):
        print("   %-22s %8.1f" % (n, buf[k] / nx))
