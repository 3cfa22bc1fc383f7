"""Which stage of the filter chain changes its result when other kernels run beside it on two more HIP streams (follow-up of
stream_race.py: the FILTERED files differed).  Every stage alone -> reference; then the same call repeated with a detector load
on two side streams, compared bit for bit."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from das4whales_amd import detect as ddet, dsp as ddsp, stream
import scipy.signal as sp
nx, ns, fs, dx, halo = 11020, 12000, 200.0, 2.0419046878814697, 1024
device = torch.device("cuda")
mask = ddsp.hybrid_ninf_filter_design((nx, ns), [0, nx * 4, 4], dx, fs, 1350., 1450., 3300, 3450, 14., 30.)
t = np.arange(ns) / fs
hf = ddet.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
lf = ddet.gen_template_fincall(t, fs, 14.7, 21.8, 0.78)
taps = [ddet._normalised_support(hf), ddet._normalised_support(lf)]
kernel = {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}
g = torch.Generator(device=device).manual_seed(5)
a, b, c = (torch.randn((nx, ns), device=device, generator=g) for _ in range(3))
sos = sp.butter(8, [14 / (fs / 2), 30 / (fs / 2)], "bp", output="sos")
load_in = torch.randn((nx, ns), device=device, generator=g)
sides = [torch.cuda.Stream(device), torch.cuda.Stream(device)]
main = torch.cuda.current_stream(device)


LOAD = os.environ.get("LOAD", "aps")          # a: analytic, p: picks, s: spectrogram correlation, t: torch elementwise ops


def load():
    keep = []
    for sd in sides:
        sd.wait_stream(main)
    with torch.cuda.stream(sides[0]):
        for _ in range(2):
            e = ddsp._analytic(load_in, 0) if "a" in LOAD else load_in
            keep.append(e)
            if "p" in LOAD:
                keep.append(ddet._find_peaks_device(e, 1.5, lazy=True))
            if "t" in LOAD:
                keep.append(torch.sqrt(load_in * load_in + 1.0))
    with torch.cuda.stream(sides[1]):
        if "s" in LOAD:
            keep.append(ddet.compute_cross_correlogram_spectrocorr(load_in, fs, [14., 30.], kernel, 0.8, 0.95))
        if "1" in LOAD:                                      # the STFT alone
            keep.append(ddsp._stft_mag(load_in, 160, 8, 11, 23, want_max=False))
        if "5" in LOAD:                                      # the STFT alone into a buffer allocated once (no allocation in the load)
            from das4whales_amd._lib import lib, check
            check(lib.d4w_stft_mag_f32(load_in.data_ptr(), S0.data_ptr(), None, nx, ns, 160, 8, 11, 23, torch.cuda.current_stream().cuda_stream))
        if "6" in LOAD:                                      # ... on a few rows only (presence, not pressure)
            from das4whales_amd._lib import lib, check
            check(lib.d4w_stft_mag_f32(load_in.data_ptr(), S0.data_ptr(), None, 64, ns, 160, 8, 11, 23, torch.cuda.current_stream().cuda_stream))
        for kind, ch in ((0, "M"), (1, "V"), (2, "L"), (3, "F")):       # synthetic neighbours: matrix instructions / vector FMAs / LDS traffic only
            if ch in LOAD:
                import ctypes
                bl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libburners.so"))
                bl.burn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
                rc = bl.burn(kind, torch.cuda.current_stream().cuda_stream, BURN_OUT.data_ptr(), 256 * 3, int(os.environ.get("BURN_ITERS", 20000)))
                assert rc == 0, rc
        if "7" in LOAD:                                      # the overlap-save band-pass as the neighbour
            keep.append(ddsp._sosfiltfilt_between(load_in, a[:, -halo:], c[:, :halo], sos))
        if "4" in LOAD:                                      # the matrix-core matched filter as the neighbour
            keep.append(ddet._xcorr_device(load_in, taps, normalize=True))
        if "2" in LOAD:                                      # the median alone (on a prepared spectrogram)
            from das4whales_amd._lib import lib, check
            med = torch.empty(nx, dtype=torch.float32, device=device)
            check(lib.d4w_row_median_f32(S0.data_ptr(), nx, S0[0].numel(), med.data_ptr(), torch.cuda.current_stream().cuda_stream))
            keep.append(med)
        if "3" in LOAD:                                      # the correlation alone
            keep.append(ddet._spectrocorr_device(S0, KER, KER.shape[1] // 2, S0.shape[2], med=MED0))
        if "t" in LOAD:
            keep.append(torch.sqrt(load_in * load_in + 2.0))
    return keep


BURN_OUT = torch.empty(256 * 3 * 256, dtype=torch.float32, device=device)
S0, _ = ddsp._stft_mag(load_in, 160, 8, 11, 23, want_max=False)
KER = np.random.default_rng(0).random((13, 19))
MED0 = torch.ones(nx, dtype=torch.float32, device=device)
torch.cuda.synchronize()
yb = ddsp._sosfiltfilt_between(b, a[:, -halo:], c[:, :halo], sos)
plan = ddsp.get_fk_plan(nx, ns, device)
plan.set_mask(mask)
stages = {
    "bp halo (d4w_fir_fft_halo_f32)": lambda: ddsp._sosfiltfilt_between(b, a[:, -halo:], c[:, :halo], sos),
    "bp edge (d4w_fir_fft_cols_f32 + row ends)": lambda: ddsp._sosfiltfilt_device(b, sos, 51),
    "f-k apply + stats": lambda: ddsp._fk_apply_stats(yb, mask, prefix=True)[0],
    "f-k apply": lambda: ddsp.fk_filter_filt(yb, mask),
    "row stats + prefix": lambda: torch.cat([v.double() for v in ddet._row_stats_cached(yb + 0, prefix=True)]),
    "matched filter (mm, rowmax)": lambda: torch.cat(ddet._xcorr_device(yb, taps, normalize=True, cont=(c[:, :halo], 176), row_max=[])),
    "matched filter (FFT form)": lambda: torch.cat(ddet._xcorr_device(yb, taps, normalize=True, method="fft")),
    "envelope (analytic_rows)": lambda: ddsp._analytic(yb, 0),
    "picks (find_peaks_prom)": lambda: ddet._find_peaks_device(ENV, 0.9).packed.float(),
    "recursion (sosfiltfilt, lanes)": lambda: ddsp._sosfiltfilt_recursive(b, np.ascontiguousarray(sos), 51),
    "STFT (stft_mm_rows)": lambda: ddsp._stft_mag(b, 160, 8, 11, 23, want_max=False)[0],
    "spectrogram correlation (median + spectro_corr)": lambda: ddet._spectrocorr_device(S0, KER, KER.shape[1] // 2, S0.shape[2]),
}
ENV = ddsp._analytic(yb, 0)
only = os.environ.get("ONLY")
if only:
    stages = {k: v for k, v in stages.items() if only in k}
for name, fn in list(stages.items())[:int(os.environ.get("NSTAGES", 99))]:
    ref = fn().clone()
    torch.cuda.synchronize()
    bad = 0
    for trial in range(6):
        k = load()
        out = fn()
        for sd in sides:
            main.wait_stream(sd)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            d = (out.double() - ref.double()).abs()
            worst = float(d.max().cpu()) / float(ref.double().abs().max().cpu())
            if d.dim() == 2 and trial < 3 and tuple(d.shape) == (nx, ns) and os.environ.get("DETAIL"):
                nzr = (d > 0).any(dim=1).nonzero().flatten().cpu().numpy()
                nzc = (d > 0).any(dim=0).nonzero().flatten().cpu().numpy()
                runs = lambda v: [(int(a[0]), int(a[-1])) for a in np.split(v, np.nonzero(np.diff(v) > 1)[0] + 1)][:12]
                r0, c0_ = int(nzr[0]), int((d[int(nzr[0])] > 0).nonzero().flatten()[0].cpu())
                sl = slice(c0_, c0_ + 6)
                for rr in nzr[:6]:
                    rr = int(rr)
                    cm = int(d[rr].argmax().cpu())
                    big = (d[rr] > 0.02).nonzero().flatten().cpu().numpy()
                    print(json.dumps({"row": rr, "argmax col": cm, "col % 3172": cm % 3172, "max diff": float(d[rr, cm].cpu()), "cols with diff > 0.02": [int(v) for v in big[:10]],
                                      "x around": [round(float(v), 4) for v in b[rr, max(cm - 2, 0):cm + 3].cpu()]}), flush=True)
                print(json.dumps({"row": r0, "col": c0_, "ref": [round(float(v), 5) for v in ref[r0, sl].cpu()], "out": [round(float(v), 5) for v in out[r0, sl].cpu()],
                                  "ref row+1": [round(float(v), 5) for v in ref[r0 + 1, sl].cpu()], "out row+1": [round(float(v), 5) for v in out[r0 + 1, sl].cpu()],
                                  "out - ref": [round(float(v), 5) for v in (out[r0, sl] - ref[r0, sl]).cpu()],
                                  "diff stats (this row)": [float(d[r0].max().cpu()), float(d[r0][d[r0] > 0].min().cpu()), int((d[r0] > 0).sum().cpu())]}), flush=True)
                print(json.dumps({"stage": name, "trial": trial, "differing elements": int((d > 0).sum().cpu()), "row runs": runs(nzr), "column runs": runs(nzc),
                                  "out finite": bool(torch.isfinite(out).all().cpu())}), flush=True)
        del k
    print(json.dumps({"stage": name, "trials": 6, "differing": bad, "worst_rel": worst if bad else 0.0}), flush=True)
    if os.environ.get("SELFCHECK"):                    # probe build with -DD4W_XF_SELFCHECK (build_variant.sh)
        import ctypes
        from das4whales_amd._lib import lib as _l
        buf8 = (ctypes.c_ulonglong * 8)()
        _l.d4w_xf_selfcheck_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _l.d4w_xf_selfcheck_read(buf8, 1)
        print(json.dumps({"stage": name, "LDS words that did not hold what the lane wrote, by dword [right after the store x4 | after a barrier and a wait x4]": list(buf8)}), flush=True)
