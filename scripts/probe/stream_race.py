"""Which result of the stream chain changes when the two detectors of a file run on a HIP stream each (bench.py --config stream
--detector-streams 2 gave pick totals that varied from run to run): the chain of bench.py on 6 files, once on one stream and a
few times with the detectors on side streams, per file: pick counts per template, checksums of the correlograms, the envelopes
and the spectrogram correlation."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from das4whales_amd import detect as ddet, dsp as ddsp, stream
nx, ns, fs, dx, halo, F = 11020, 12000, 200.0, 2.0419046878814697, 1024, 6
device = torch.device("cuda")
mask = ddsp.hybrid_ninf_filter_design((nx, ns), [0, nx * 4, 4], dx, fs, 1350., 1450., 3300, 3450, 14., 30.)
t = np.arange(ns) / fs
hf = ddet.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
lf = ddet.gen_template_fincall(t, fs, 14.7, 21.8, 0.78)
kernel = {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}
files = [torch.randn((nx, ns), device=device, generator=torch.Generator(device=device).manual_seed(1000 + i)) for i in range(F)]


def chk(x):
    return float(x.double().sum().cpu()), float(x.double().abs().max().cpu())


def run(nstreams, split_env):
    sides = [torch.cuda.Stream(device) for _ in range(nstreams)]
    main = torch.cuda.current_stream(device)
    st = stream.FileStream(fs, 14, 30, templates=[hf, lf], fk_mask=mask, halo=halo)
    keep, out = [], []

    def detect_on(done):
        for r in done:
            keep.append(r)
            for sd in sides:
                sd.wait_stream(main)
            rec = {"index": r["index"]}
            with torch.cuda.stream(sides[0]) if sides else torch.cuda.stream(main):
                rm = r.get("row_max")
                thr = ddet.Threshold(0.45, ddet.correlogram_max(r["correlograms"][0], rm[0] if rm else None, on_device=True))
                rec["thr"] = thr
                if split_env:
                    rec["env"] = [ddsp._analytic(c, 0) for c in r["correlograms"]]
                    rec["picks"] = [ddet._find_peaks_device(e, thr, lazy=True) for e in rec["env"]]
                else:
                    rec["picks"] = [ddet.pick_times_env(c, thr, lazy=True) for c in r["correlograms"]]
            with torch.cuda.stream(sides[-1]) if sides else torch.cuda.stream(main):
                rec["sc"] = ddet.compute_cross_correlogram_spectrocorr(r["filtered"], fs, [14., 30.], kernel, 0.8, 0.95)
            out.append(rec)
    for x in files:
        detect_on(st.push(x))
    detect_on(st.flush())
    for sd in sides:
        main.wait_stream(sd)
    torch.cuda.synchronize()
    res = []
    for rec, r in zip(out, keep):
        res.append({"index": rec["index"], "thr": float(rec["thr"]), "picks": [p.total for p in rec["picks"]],
                    "corr": [chk(c) for c in r["correlograms"]], "filtered": chk(r["filtered"]), "sc": chk(rec["sc"]),
                    "env": [chk(e) for e in rec.get("env", [])]})
    return res


ref = run(0, True)
print(json.dumps({"reference picks": [r["picks"] for r in ref]}))
for trial in range(4):
    got = run(2, True)
    diffs = [(a["index"], k) for a, b in zip(ref, got) for k in a if a[k] != b[k]]
    print(json.dumps({"trial": trial, "streams": 2, "differs": diffs, "picks": [r["picks"] for r in got] if diffs else "same"}))
for trial in range(2):
    got = run(1, True)
    diffs = [(a["index"], k) for a, b in zip(ref, got) for k in a if a[k] != b[k]]
    print(json.dumps({"trial": trial, "streams": 1, "differs": diffs}))
