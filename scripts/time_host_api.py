"""What a notebook user gets (SURVEY 8d "H2D / D2H reported separately", reference contract data_handle.py:213: the callers hand
over host float64 arrays): wall time of bp_filt, fk_filter_sparsefilt and compute_cross_correlogram called with a HOST float64
block and returning a host float64 array, split into upload (H2D incl. the float64 -> float32 conversion), the device-resident
call, and download (D2H incl. float32 -> float64).  NX / NS select the block (default 11020 x 12000).  One JSON line."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import das4whales_amd as dw
from das4whales_amd import _device as dev
nx, ns = int(os.environ.get("NX", 11020)), int(os.environ.get("NS", 12000))
fs, dx = 200.0, 2.0419046878814697
rng = np.random.default_rng(0)
xh = rng.standard_normal((nx, ns))                      # pageable float64, as h5py + raw2strain leave it
t = np.arange(ns) / fs
hf = dw.detect.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
mask = dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx * 4, 4], dx, fs, 1350., 1450., 3300, 3450, 14., 30.)


def wall(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); del r
    return round(float(np.median(ts)), 2)


xd = dev.upload_f32(xh)
out = {"shape": [nx, ns], "host_block_MB_f64": round(xh.nbytes / 1e6, 1), "torch_threads": torch.get_num_threads()}
out["upload_f64_to_f32_ms"] = wall(lambda: dev.upload_f32(xh))
out["download_f32_to_f64_ms"] = wall(lambda: dev.download(xd, np.float64))
out["naive_astype_then_to_ms"] = wall(lambda: torch.from_numpy(xh.astype(np.float32)).to("cuda"), reps=2)
out["naive_cpu_then_astype_ms"] = wall(lambda: xd.cpu().numpy().astype(np.float64), reps=2)
pin = torch.empty((nx, ns), dtype=torch.float32).pin_memory()
pag = np.empty((nx, ns), dtype=np.float32)
out["convert_f64_to_pinned_f32_ms"] = wall(lambda: dev._convert(pin.numpy(), xh))
out["convert_f64_to_pageable_f32_ms"] = wall(lambda: dev._convert(pag, xh))
out["numpy_astype_f32_ms"] = wall(lambda: xh.astype(np.float32), reps=3)
xp = torch.from_numpy(xh.astype(np.float32)).pin_memory()
out["pinned_f32_h2d_ms"] = wall(lambda: xp.to("cuda", non_blocking=True))
out["pinned_f32_h2d_GBps"] = round(xp.numel() * 4 / out["pinned_f32_h2d_ms"] / 1e6, 1)
for name, host_call, dev_call in (
        ("bp_filt", lambda: dw.dsp.bp_filt(xh, fs, 14, 30), lambda: dw.dsp.bp_filt(xd, fs, 14, 30)),
        ("fk_filter_sparsefilt", lambda: dw.dsp.fk_filter_sparsefilt(xh, mask), lambda: dw.dsp.fk_filter_sparsefilt(xd, mask)),
        ("compute_cross_correlogram", lambda: dw.detect.compute_cross_correlogram(xh, hf), lambda: dw.detect.compute_cross_correlogram(xd, hf))):
    r = host_call()
    assert isinstance(r, np.ndarray) and r.dtype == np.float64 and r.shape == xh.shape
    out[name] = {"host_in_host_out_ms": wall(host_call), "device_resident_ms": wall(dev_call)}
print(json.dumps(out))
