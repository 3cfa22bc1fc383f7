"""f-k filter at 20000 x 120000: plain pass order against the slab order (D4W_FK_SLAB = column blocks per slab)
for the masks of SURVEY 8(d); also times the mask fold.  Output: one line per (mask, slab width)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import das4whales_amd as dw

nx, ns = 20000, 120000
fs, dx = 200.0, 2.0419046878814697
widths = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,5,6,10".split(","))]
gen = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((nx, ns), device="cuda", generator=gen)
y = torch.empty_like(x)
samples = float(nx) * ns
res = []
masks = {
    "dense": lambda: (torch.rand((nx, ns), device="cuda", generator=gen), 0.0),
    "classic": lambda: (dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], dx, fs), 0.0),
    "hybrid_ninf": lambda: (dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx, 1], dx, fs, 1350., 1450., 3300, 3450, 14., 30.), 0.0),
    "hybrid_ninf_pruned": lambda: (dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx, 1], dx, fs, 1350., 1450., 3300, 3450, 14., 30.), 4e-6),
}
for name, mk in masks.items():
    m, eps = mk()
    ref = None
    for sw in widths:
        if sw:
            os.environ["D4W_FK_SLAB"] = str(sw)
        else:
            os.environ.pop("D4W_FK_SLAB", None)
        plan = dw.dsp.FkPlan(nx, ns)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.set_mask(m, prune_eps=eps)
        torch.cuda.synchronize()
        t_fold = (time.perf_counter() - t0) * 1e3
        plan.apply(x, out=y)
        acc = [0.0] * 5
        reps = 4
        for _ in range(reps):
            _, ms = plan.apply_timed(x, out=y)
            acc = [a + b for a, b in zip(acc, ms)]
        acc = [a / reps for a in acc]
        # with the row statistics epilogue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        plan.apply_stats(x, out=y)
        e0.record()
        for _ in range(reps):
            plan.apply_stats(x, out=y)
        e1.record(); e1.synchronize()
        t_stats = e0.elapsed_time(e1) / reps
        if ref is None:
            ref = y.clone()
            same = True
        else:
            same = bool(torch.equal(ref, y))
        tot = sum(acc)
        line = {"mask": name, "slab": sw, "live_rows": plan.live_rows(), "set_mask_ms": round(t_fold, 2), "passes_ms": [round(a, 3) for a in acc],
                "fk_ms": round(tot, 3), "fk_stats_ms": round(t_stats, 3), "frac24": round(24 * samples / (tot * 1e-3) / 8e12, 4), "same_as_plain": same}
        print(json.dumps(line), flush=True)
        del plan
    del m, ref
    torch.cuda.empty_cache()
