"""Which device kernels one file of the streaming detection chain launches in steady state (stream.FileStream.push + envelope
picks + spectrogram correlation on resident float32 strain -- the bench's data generation stays outside the profiled region),
by name with calls and time per file: every row that is not a d4w:: kernel (or a runtime copy / fill) is glue the product
path should not need.  torch.profiler, device activities only."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from das4whales_amd import detect as ddet, dsp as ddsp, stream
nx, ns, fs, dx, halo = int(os.environ.get("NX", 11020)), int(os.environ.get("NS", 12000)), 200.0, 2.0419046878814697, 1024
F = int(os.environ.get("FILES", 8))
mask = ddsp.hybrid_ninf_filter_design((nx, ns), [0, nx * 4, 4], dx, fs, 1350., 1450., 3300, 3450, 14., 30.)
t = np.arange(ns) / fs
hf = ddet.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
lf = ddet.gen_template_fincall(t, fs, 14.7, 21.8, 0.78)
kernel = {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}
g = torch.Generator(device="cuda").manual_seed(5)
files = [torch.randn((nx, ns), device="cuda", generator=g) for _ in range(F)]


def detect_on(done):
    n = 0
    for r in done:
        rm = r.get("row_max")
        thr = 0.45 * ddet.correlogram_max(r["correlograms"][0], rm[0] if rm else None)
        for c in r["correlograms"]:
            n += ddet.pick_times_env(c, thr).total
        ddet.compute_cross_correlogram_spectrocorr(r["filtered"], fs, [14., 30.], kernel, 0.8, 0.95)
    return n


def run(files_):
    st = stream.FileStream(fs, 14, 30, templates=[hf, lf], fk_mask=mask, halo=halo)
    n = 0
    for x in files_:
        n += detect_on(st.push(x))
    return st, n


run(files)                                  # warm-up: plans, tables, capacity memos
torch.cuda.synchronize()
st = stream.FileStream(fs, 14, 30, templates=[hf, lf], fk_mask=mask, halo=halo)
for x in files[:3]:
    detect_on(st.push(x))                   # the pipeline is full: every further push finishes one file
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for x in files[3:]:
        detect_on(st.push(x))
    torch.cuda.synchronize()
nf = len(files) - 3
rows = []
for e in prof.key_averages():
    dt = getattr(e, "device_time_total", None)
    if dt is None:
        dt = getattr(e, "cuda_time_total", 0.0)
    if dt <= 0:
        continue
    rows.append((e.key, e.count / nf, dt / nf))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
glue = [r for r in rows if not ("d4w::" in r[0] or r[0].startswith("Memcpy") or r[0].startswith("Memset") or "rocclr" in r[0])]
print("per file: %.1f us of device time in %d kernel names; not d4w: %.1f us (%.1f %%)" % (tot, len(rows), sum(r[2] for r in glue), 100 * sum(r[2] for r in glue) / max(tot, 1e-9)))
for name, calls, us in rows:
    print("%9.1f us  %5.2f calls  %s%s" % (us, calls, "" if (name, calls, us) not in glue else "[GLUE] ", name[:150]))
