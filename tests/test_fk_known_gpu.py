"""Known-answer parity tests of the f-k filter at the benchmarked configuration (20 000 x 120 000, the
shape-specialised kernels BENCH times) and at the 60-s file shapes: answers that are independent of any
other implementation of the filter (tests/known_answers.py), with the mask gains taken from the CPU
oracle's pointwise evaluators (oracle/d4w_oracle.py, pinned in tests/test_oracle_pointwise.py).

Reference: dsp.fk_filter_filt / fk_filter_sparsefilt dsp.py:725-786, designs dsp.py:85-171, 308-454,
the scripts' call scripts/main_mfdetect.py:46-55.  Tolerance: max|y - y_ref| <= 1e-5 max|y_ref|."""
import numpy as np
import pytest
import torch

from oracle import d4w_oracle as orc
from tests import known_answers as ka

pytestmark = pytest.mark.gpu
TOL = 1e-5
DX, FS = 2.0419046878814697, 200.0
NINF = dict(cs_min=1350., cp_min=1450., cp_max=3300, cs_max=3450, fmin=14., fmax=30.)   # scripts/main_mfdetect.py:46-47


@pytest.fixture(scope="module")
def dw():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import das4whales_amd as dw_
    from das4whales_amd import _lib
    assert "gfx950" in _lib.version()
    torch.set_float32_matmul_precision("highest")
    return dw_


def _design(dw, kind, shape, sel):
    if kind == "classic":
        m = dw.dsp.fk_filter_design(shape, sel, DX, FS)
        at = lambda i, j: orc.fk_filter_design_at(shape, sel, DX, FS, i, j)
    else:
        m = dw.dsp.hybrid_ninf_filter_design(shape, sel, DX, FS, NINF["cs_min"], NINF["cp_min"], NINF["cp_max"],
                                             NINF["cs_max"], NINF["fmin"], NINF["fmax"])
        at = lambda i, j: orc.hybrid_ninf_filter_design_at(shape, sel, DX, FS, i, j, **NINF)
    return m, at


def _plane_wave_check(plan, shape, sel, at, seed, host_rows=16):
    """Filter a superposition of on-grid plane waves with `plan` (mask already set): the output must be the
    same waves scaled by the oracle's folded gains -- on the whole block (float32 factors on the device)
    and on `host_rows` rows in float64 on the host."""
    nx, ns = shape
    rng = np.random.default_rng(seed)
    kx, kt, amp, ph = ka.pick_plane_waves(nx, ns, sel, DX, FS, rng)
    g = orc.folded_gain_at(at, shape, kx, kt)
    assert np.count_nonzero(g) >= 6 and np.count_nonzero(g == 0) >= 6
    A, B = ka.wave_factors(nx, ns, kx, kt, ph)
    A32, B32 = A.astype(np.float32), B.astype(np.float32)
    Ad, Bd = torch.from_numpy(A32).cuda(), torch.from_numpy(B32).cuda()
    x = (Ad * torch.from_numpy(np.tile(amp, 2).astype(np.float32)).cuda()) @ Bd
    y = plan.apply(x)
    del x
    ref = (Ad * torch.from_numpy(np.tile(amp * g, 2).astype(np.float32)).cuda()) @ Bd
    scale = float(ref.abs().max())
    err_dev = float((y - ref).abs().max()) / scale
    del ref
    rows = np.unique(np.concatenate(([0, 1, nx // 2, nx - 1], rng.integers(0, nx, host_rows))))
    ref64 = (A32[rows].astype(np.float64) * np.tile(amp * g, 2)) @ B32.astype(np.float64)
    err_host = float(np.max(np.abs(y[torch.from_numpy(rows).cuda()].cpu().numpy().astype(np.float64) - ref64))) / scale
    return err_dev, err_host, int(np.count_nonzero(g))


def _impulse_check(plan, mask_tensor, shape, c0, n0, deltas):
    """Unit impulse at (c0, n0): rows c0 + deltas of the output against Re ifft2(M') evaluated in float64
    (channel-axis sum on the device in float64, time-axis inverse transform on the host)."""
    nx, ns = shape
    x = torch.zeros(shape, dtype=torch.float32, device="cuda")
    x[c0, n0] = 1.0
    y = plan.apply(x)
    del x
    deltas = np.asarray(deltas)
    rows = (c0 + deltas) % nx
    accR = torch.zeros((len(deltas), ns), dtype=torch.float64, device="cuda")
    accI = torch.zeros_like(accR)
    dl = torch.from_numpy(deltas % nx).cuda()
    for s0 in range(0, nx, 1024):                   # shifted-grid rows s0:s1 hold wavenumbers k = s - nx//2 (mod nx)
        s1 = min(nx, s0 + 1024)
        k = (torch.arange(s0, s1, device="cuda") - nx // 2) % nx
        phase = 2 * np.pi * ((k[None, :] * dl[:, None]) % nx).double() / nx
        m = mask_tensor[s0:s1].double()
        accR += torch.cos(phase) @ m
        accI += torch.sin(phase) @ m
    G = (accR.cpu().numpy() + 1j * accI.cpu().numpy()) / nx
    G = np.fft.ifftshift(G, axes=1)                  # time axis of the mask back to the unshifted grid
    h = np.fft.ifft(G, axis=1).real
    got = np.roll(y[torch.from_numpy(rows).cuda()].cpu().numpy().astype(np.float64), -n0, axis=1)
    scale = float(y.abs().max())
    assert abs(scale - np.max(np.abs(h))) < 1e-4 * scale          # the peak of the response sits in the checked rows
    return float(np.max(np.abs(got - h))) / scale


@pytest.mark.parametrize("kind", ["classic", "hybrid_ninf"])
def test_bench_shape_plane_waves_and_impulse(dw, kind):
    """20 000 x 120 000 with the masks of SURVEY 8(d): the specialised kernels (the only instantiation the
    benchmark times) against closed-form answers."""
    shape, sel = (20000, 120000), [0, 20000, 1]
    mask, at = _design(dw, kind, shape, sel)
    plan = dw.dsp.get_fk_plan(*shape)
    plan.set_mask(mask)
    live = plan.live_rows()
    err_dev, err_host, npass = _plane_wave_check(plan, shape, sel, at, seed=20 + len(kind))
    print("20000x120000 %s (%d live rows): plane waves, %d with non-zero gain: device %.3e host-f64 %.3e"
          % (kind, live, npass, err_dev, err_host))
    assert err_dev < TOL and err_host < TOL
    err_imp = _impulse_check(plan, mask.tensor, shape, 12345, 67891, [0, 1, 2, 5, 40, 999, shape[0] // 2, shape[0] - 1, -7, 7654])
    print("20000x120000 %s: unit impulse, 10 rows vs float64 Re ifft2(M): %.3e" % (kind, err_imp))
    assert err_imp < TOL


@pytest.mark.parametrize("nx,ns,step", [(4000, 12000, 4), (11020, 12000, 4), (5510, 12000, 8), (13223, 12000, 4)])
@pytest.mark.parametrize("kind", ["classic", "hybrid_ninf"])
def test_file_shapes_plane_waves(dw, nx, ns, step, kind):
    """60-s file shapes (BASELINE configs[0..1], the real OOI selections incl. 13223 = 7 x 1889 channels):
    plane waves against the oracle's gains, impulse rows against float64 Re ifft2(M)."""
    shape, sel = (nx, ns), [0, nx * step, step]
    mask, at = _design(dw, kind, shape, sel)
    plan = dw.dsp.get_fk_plan(nx, ns)
    plan.set_mask(mask)
    err_dev, err_host, npass = _plane_wave_check(plan, shape, sel, at, seed=nx)
    print("%dx%d %s: plane waves device %.3e host %.3e (%d passing)" % (nx, ns, kind, err_dev, err_host, npass))
    assert err_dev < TOL and err_host < TOL
    err_imp = _impulse_check(plan, mask.tensor, shape, nx // 3, 4321, [0, 1, 3, nx // 2, nx - 1, -2])
    assert err_imp < TOL


def test_mask_edited_in_place_is_refolded(dw):
    """A NumPy mask modified in place between two calls must not reuse the folded copy of the first call
    (round-1 bug: the plan cache keyed on id(mask))."""
    rng = np.random.default_rng(5)
    nx, ns = 40, 480
    x = rng.standard_normal((nx, ns))
    m = rng.uniform(size=(nx, ns))
    y1 = dw.dsp.fk_filter_filt(x, m)
    assert np.max(np.abs(y1 - orc.fk_filter_filt(x, m))) < TOL * np.max(np.abs(y1))
    m *= 0.25
    m[:, ::3] = 0.0
    y2 = dw.dsp.fk_filter_filt(x, m)
    ref2 = orc.fk_filter_filt(x, m)
    assert np.max(np.abs(y2 - ref2)) < TOL * np.max(np.abs(ref2))
    # a torch mask edited in place bumps its version counter: also re-folded
    mt = torch.from_numpy(m.astype(np.float32)).cuda()
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    y3 = dw.dsp.fk_filter_filt(xt, mt)
    mt.mul_(2.0)
    y4 = dw.dsp.fk_filter_filt(xt, mt)
    assert float((y4 - 2.0 * y3).abs().max()) < TOL * float(y4.abs().max())


def test_pruned_mode_on_ocean_wave_block(dw):
    """Opt-in tail pruning (FkPlan.set_mask(m, prune_eps=4e-6), the only non-exact mode of the filter) against the exact
    filter on SURVEY 8(d)'s S-large recipe at 20 000 x 120 000: white noise 1e-9 + 40 ocean-wave plane waves of 1e-8 EACH
    + six fin-whale notes of 5e-9, the scripts' hybrid_ninf mask.  What the pruned mode drops is bounded by
    prune_eps * max|M_h| times the spectral L1 content of the input in the dropped rows; on this input (the ocean waves sit
    below 8 Hz, where the exact mask is zero outside the fan anyway) that is far below the 1e-5 budget -- asserted here --
    and a tone placed INSIDE the dropped region comes back as prune_eps-sized gain x amplitude, which the second half pins."""
    shape, sel = (20000, 120000), [0, 20000, 1]
    mask, at = _design(dw, "hybrid_ninf", shape, sel)
    x, calls = ka.synth_block_device(shape[0], shape[1], "cuda", ocean_amp=1e-8)
    plan = dw.dsp.get_fk_plan(*shape)
    plan.set_mask(mask)
    y_exact = plan.apply(x)
    plan.set_mask(mask, prune_eps=4e-6)
    live = plan.live_rows()
    y_pruned = plan.apply(x)
    scale = float(y_exact.abs().max())
    err = float((y_pruned - y_exact).abs().max()) / scale
    print("S-large, hybrid_ninf: pruned (eps 4e-6, %d live rows) vs exact: %.3e of max|y| = %.3e (input max %.3e)"
          % (live, err, scale, float(x.abs().max())))
    assert live < shape[0] // 4
    assert err < TOL
    # the calls survive, the ocean waves do not: the filtered block stays far below the 40 x 1e-8 of the input
    assert 1e-9 < scale < 0.25 * float(x.abs().max())
    del x, y_exact, y_pruned
    # a plane wave inside the dropped rows (|k| = 0.1 1/m, 46 Hz: gain 2 |H(46 Hz)|^2 ~ 1e-6) next to a pass-band wave
    nx, ns = shape
    dk, df = 1.0 / (nx * sel[2] * DX), FS / ns
    kx = np.array([int(round(20.0 / 2000.0 / dk)), int(round(0.1 / dk))])
    kt = np.array([int(round(20.0 / df)), int(round(46.0 / df))])
    g = orc.folded_gain_at(at, shape, kx, kt)
    assert g[0] > 0.5 and 0 < g[1] < 4e-6 * 2.0
    amp, ph = np.array([1.0, 10.0]), np.array([0.3, 1.1])
    A, B = ka.wave_factors(nx, ns, kx, kt, ph)
    Ad, Bd = torch.from_numpy(A.astype(np.float32)).cuda(), torch.from_numpy(B.astype(np.float32)).cuda()
    xw = (Ad * torch.from_numpy(np.tile(amp, 2).astype(np.float32)).cuda()) @ Bd
    yw = plan.apply(xw)                                     # pruned
    ref_exact = (Ad * torch.from_numpy(np.tile(amp * g, 2).astype(np.float32)).cuda()) @ Bd
    ref_pruned = (Ad * torch.from_numpy(np.tile(amp * g * np.array([1.0, 0.0]), 2).astype(np.float32)).cuda()) @ Bd
    sc = float(ref_exact.abs().max())
    e_pruned = float((yw - ref_pruned).abs().max()) / sc
    e_exact = float((yw - ref_exact).abs().max()) / sc
    print("tone in the dropped rows: pruned output vs (exact answer without that tone) %.3e; vs exact answer %.3e; "
          "bound gain x amplitude = %.3e" % (e_pruned, e_exact, g[1] * amp[1] / sc))
    assert e_pruned < TOL
    assert e_exact <= g[1] * amp[1] / sc + TOL            # the documented bound: dropped gain x amplitude of what sits there


def test_pass_order_is_chosen_per_mask(dw):
    """The planner models the bytes of both pass orders for every mask (include/d4w.h d4w_fk_plan_order): the scripts'
    hybrid_ninf design runs time-first (band 4-44 Hz through the channel transform, Butterworth skirts as tail columns, the
    rest never written) at ANY channel spacing; the classic fan and the sine-taper hybrid run channel-first with their dead
    wavenumber rows skipped; with the opt-in tail pruning hybrid_ninf has dead rows too and goes back to channel-first; a
    dense random mask stays channel-first.  Both orders give the same filter (the forced orders agree to float32 rounding)."""
    import os
    shape = (4000, 12000)
    plan = dw.dsp.get_fk_plan(*shape)
    x = torch.randn(shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    for step in (1, 4):
        sel = [0, shape[0] * step, step]
        ninf, _ = _design(dw, "hybrid_ninf", shape, sel)
        plan.set_mask(ninf)
        od = plan.order()
        assert od["order"] == "time-first" and od["tail_columns"] > 0 and od["band_columns"] < 0.5 * od["half_spectrum_columns"], od
        assert od["model_bytes_per_sample"]["time-first"] < 35.0 and od["model_bytes_per_sample"]["channel-first"] == 42.0
        y_tf = plan.apply(x)
        os.environ["D4W_FK_ORDER"] = "cf"
        try:
            plan.set_mask(ninf.tensor.clone())                    # a new tensor: not the cached fold
            assert plan.order()["order"] == "channel-first"
            y_cf = plan.apply(x)
        finally:
            del os.environ["D4W_FK_ORDER"]
        assert float((y_tf - y_cf).abs().max()) < 2e-6 * float(y_cf.abs().max())
    # opt-in tail pruning: at 2.04 m spacing the fan leaves most wavenumber rows dead -> channel-first wins; at 8.17 m half
    # of them stay alive and dropping the skirt columns time-first is cheaper -- whichever the model says
    ninf1, _ = _design(dw, "hybrid_ninf", shape, [0, shape[0], 1])
    plan.set_mask(ninf1, prune_eps=4e-6)
    assert plan.order()["order"] == "channel-first" and plan.live_rows() < shape[0] // 4
    plan.set_mask(ninf, prune_eps=4e-6)
    od = plan.order()
    mb = od["model_bytes_per_sample"]
    assert od["order"] == ("time-first" if mb["time-first"] < 0.97 * mb["channel-first"] else "channel-first") and od["tail_columns"] == 0
    classic, _ = _design(dw, "classic", shape, [0, shape[0], 1])
    plan.set_mask(classic)
    assert plan.order()["order"] == "channel-first" and plan.live_rows() < shape[0] // 2
    hyb = dw.dsp.hybrid_filter_design(shape, [0, shape[0], 1], DX, FS, 1350., 1450., 14., 30.)
    plan.set_mask(hyb)
    od = plan.order()
    assert od["order"] == ("time-first" if od["model_bytes_per_sample"]["time-first"] < 0.97 * od["model_bytes_per_sample"]["channel-first"] else "channel-first")
    plan.set_mask(torch.rand(shape, device="cuda"))
    assert plan.order()["order"] == "channel-first" and plan.live_rows() == shape[0]


@pytest.mark.parametrize("nx,ns", [(96, 480), (74, 480), (4000, 12000), (5510, 12000), (11020, 12000), (13223, 12000), (8000, 12000), (600, 120000)])
def test_time_first_equals_channel_first_on_every_kernel_configuration(dw, nx, ns):
    """Built-in and compiled-on-demand configurations (different radix triples, even and odd N1, prime c2 radices): a mask
    with band, wavenumber-independent skirt and zero columns through both pass orders -- same output to float32 rounding,
    and against the float64 oracle where it is cheap."""
    import os
    dw.dsp.compile_fk_shape(nx, ns)
    g = torch.Generator(device="cuda").manual_seed(nx + ns)
    x = torch.randn((nx, ns), device="cuda", generator=g)
    fb = torch.abs(torch.arange(ns, device="cuda") - ns // 2)                    # |f bin| of the shifted columns
    lo, hi, tail_to = ns // 20, ns // 7, ns // 4
    m = torch.zeros((nx, ns), device="cuda")
    band = (fb >= lo) & (fb < hi)
    m[:, band] = torch.rand((nx, int(band.sum())), device="cuda", generator=g)
    tail = (fb >= hi) & (fb < tail_to)
    m[:, tail] = (1e-3 * torch.exp(-(fb[tail] - hi).float() / (0.02 * ns)))[None, :]
    plan = dw.dsp.FkPlan(nx, ns)
    outs = {}
    for order in ("tf", "cf"):
        os.environ["D4W_FK_ORDER"] = order
        try:
            plan.set_mask(m.clone())
            od = plan.order()
            outs[order] = plan.apply(x, taper=True)
        finally:
            del os.environ["D4W_FK_ORDER"]
        if order == "tf":
            assert od["order"] == "time-first" and od["tail_columns"] > 0 and od["band_columns"] + od["tail_columns"] < ns // 2, od
        else:
            assert od["order"] == "channel-first"
    scale = float(outs["cf"].abs().max())
    err = float((outs["tf"] - outs["cf"]).abs().max()) / scale
    print("%d x %d: time-first vs channel-first %.3e" % (nx, ns, err))
    assert err < 3e-6
    if nx * ns <= 4000 * 12000:
        ref = orc.fk_filter_filt(x.cpu().numpy().astype(np.float64), m.cpu().numpy().astype(np.float64), tapering=True)
        e2 = float(np.max(np.abs(outs["tf"].cpu().numpy() - ref)) / np.max(np.abs(ref)))
        print("%d x %d: time-first vs oracle %.3e" % (nx, ns, e2))
        assert e2 < TOL
