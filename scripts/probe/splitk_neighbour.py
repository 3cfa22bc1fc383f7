"""Round 6: the overlap-save FFT kernels beside ONE foreign neighbour (a short-and-deep GEMM that the BLAS library splits along K),
which moved their results in 39-40 of 40 trials (profiles/r06f/concurrency_trials.txt).  Prints how many of TRIALS differ for each
victim; D4W_XF_LDS_CLAIM / D4W_HAZARD_FENCE etc. come from the environment.  SHAPE = M,K,N of the neighbour; DT = f16 | bf16;
SIDES = side streams that run it (1 or 2: two concurrent split-K products can wait for each other for ever)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, scipy.signal as sp
from das4whales_amd import detect as ddet, dsp as ddsp
FS = 200.0
nx, ns, halo = 11020, 12000, 1024
M, K, N = [int(v) for v in os.environ.get("SHAPE", "256,32768,256").split(",")]
dt = torch.float16 if os.environ.get("DT", "f16") == "f16" else torch.bfloat16
trials, nside = int(os.environ.get("TRIALS", 20)), int(os.environ.get("SIDES", 1))
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(6)
a, b, c = (torch.randn((nx, ns), device=dev, generator=g) for _ in range(3))
ka, kb = torch.randn((M, K), device=dev, generator=g).to(dt), torch.randn((K, N), device=dev, generator=g).to(dt)
t = np.arange(ns) / FS
taps = [ddet._normalised_support(ddet.gen_template_fincall(t, FS, 17.8, 28.8, 0.68)),
        ddet._normalised_support(ddet.gen_template_fincall(t, FS, 14.7, 21.8, 0.78))]
sos = sp.butter(8, [14 / (FS / 2), 30 / (FS / 2)], "bp", output="sos")
yb = ddsp._sosfiltfilt_between(b, a[:, -halo:], c[:, :halo], sos)
stages = {"bp_filt": lambda: ddsp.bp_filt(b, FS, 14.0, 30.0),
          "bp between files": lambda: ddsp._sosfiltfilt_between(b, a[:, -halo:], c[:, :halo], sos),
          "mf fft form": lambda: torch.cat(ddet._xcorr_device(yb, taps, normalize=True, method="fft")),
          "mf matrix cores": lambda: torch.cat(ddet._xcorr_device(yb, taps, normalize=True))}
sides = [torch.cuda.Stream(dev) for _ in range(nside)]
main = torch.cuda.current_stream(dev)
out = {"neighbour": [M, K, N, str(dt)], "sides": nside, "trials": trials, "claim_kib": os.environ.get("D4W_XF_LDS_CLAIM", "0")}
for name, fn in stages.items():
    ref = fn().clone()
    torch.cuda.synchronize()
    bad, worst, rows_bad = 0, 0.0, 0
    for _ in range(trials):
        keep = []
        for sd in sides:
            sd.wait_stream(main)
            with torch.cuda.stream(sd):
                keep.append([torch.matmul(ka, kb) for _ in range(3)])
        y = fn()
        with torch.cuda.stream(sides[0]):
            keep.append([torch.matmul(ka, kb) for _ in range(3)])
        for sd in sides:
            main.wait_stream(sd)
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad += 1
            d = (y.double() - ref.double()).abs()
            worst = max(worst, float(d.max() / ref.double().abs().max()))
            rows_bad = max(rows_bad, int((d.reshape(-1, ns).max(dim=1).values > 0).sum()))
        del keep, y
    out[name] = {"differ": bad, "worst_rel": worst, "rows_touched_max": rows_bad}
print(json.dumps(out), flush=True)
