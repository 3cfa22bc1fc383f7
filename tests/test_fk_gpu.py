"""GPU parity tests of the f-k filter: HIP path (through the C ABI) vs the reference's golden
outputs and vs the CPU oracle on seeded inputs; size-independent properties at full size.

Tolerance (north star): max|y - y_ref| <= 1e-5 * max|y_ref| with float32 arithmetic."""
import numpy as np
import pytest
import torch

from oracle import d4w_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(y, ref):
    return float(np.max(np.abs(np.asarray(y, dtype=np.float64) - ref)) / np.max(np.abs(ref)))


@pytest.fixture(scope="module")
def dw():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import das4whales_amd as dw_
    from das4whales_amd import _lib
    assert "gfx950" in _lib.version()
    return dw_


def test_golden_all_masks(dw, golden):
    g = golden("fk_40x480.npz")
    x = g["x"]
    for km, ky in [("m_classic", "y_classic"), ("m_ninf", "y_ninf"), ("m_hybrid", "y_hybrid"),
                   ("m_ninf_gs", "y_ninf_gs")]:
        y = dw.dsp.fk_filter_filt(x, g[km])
        assert y.dtype == np.float64 and y.shape == x.shape
        assert rel(y, g[ky]) < TOL, km
    # Fortran-ordered mask (what fk_filter_design returns, dsp.py:137) and the sparse entry point
    yf = dw.dsp.fk_filter_sparsefilt(x, np.asfortranarray(g["m_classic"]))
    assert rel(yf, g["y_classic"]) < TOL
    xc = x.copy()
    yt = dw.dsp.fk_filter_filt(xc, g["m_classic"], tapering=True)
    assert rel(yt, g["y_classic_taper"]) < TOL
    assert np.array_equal(xc, x)            # not modified in place (documented deviation)
    # opt-in: the reference's side effect (dsp.py:744-745 tapers the caller's array on the way), NumPy and CUDA inputs
    yt2 = dw.dsp.fk_filter_filt(xc, g["m_classic"], tapering=True, inplace_taper=True)
    assert rel(yt2, g["y_classic_taper"]) < TOL
    assert rel(xc, orc.taper_data(x.copy())) < TOL and not np.array_equal(xc, x)
    xg = torch.from_numpy(x).float().cuda()
    yt3 = dw.dsp.fk_filter_sparsefilt(xg, g["m_classic"], tapering=True, inplace_taper=True)
    assert rel(yt3.cpu().numpy(), g["y_classic_taper"]) < TOL and rel(xg.cpu().numpy(), orc.taper_data(x.copy())) < TOL


def test_golden_second_shape_and_coo(dw, golden):
    h = golden("fk_30x360.npz")

    class COO:                               # sparse.COO duck type: .todense(), .shape
        def __init__(self, d):
            self._d, self.shape = d, d.shape

        def todense(self):
            return self._d
    assert rel(dw.dsp.fk_filter_sparsefilt(h["x"], COO(h["m_ninf"])), h["y_ninf"]) < TOL
    assert rel(dw.dsp.fk_filter_filt(h["x"].astype(np.float32), h["m_classic"]), h["y_classic"]) < TOL


@pytest.mark.parametrize("nx,ns,opts", [
    (38, 406, (19, 2, 7, 29, 4, 4)),        # loop-based prime radices 19, 29, 7
    (93, 286, (31, 3, 11, 13, 2, 2)),
    (7, 14, None), (1, 64, None), (64, 2, None), (3, 4, None),
    (551, 1200, None),                       # 19 * 29 channels (OOI 5510 = 10 * 551)
    (250, 3000, (5, 50, 3, 500, 8, 16)),
    (250, 3000, (25, 10, 15, 100, 16, 16)),
    (1000, 2400, None),
])
def test_random_shapes_vs_oracle(dw, nx, ns, opts):
    rng = np.random.default_rng(nx * 7919 + ns)
    x = rng.standard_normal((nx, ns))
    m = rng.random((nx, ns))                 # arbitrary, non-Hermitian mask
    ref = orc.fk_filter_filt(x, m)
    plan = dw.dsp.FkPlan(nx, ns, opts=opts)
    plan.set_mask(m)
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    y = plan.apply(xt)
    assert rel(y.cpu().numpy(), ref) < TOL
    # in place
    plan.apply(xt, out=xt)
    assert rel(xt.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("nx,ns", [(18, 48), (8, 480), (100, 600)])
def test_shape_specialised_kernels_vs_oracle(dw, nx, ns):
    """Shapes with specialised (fat-stage register FFT) kernels, and the generic passes forced on
    the same shapes (opts[0] = -1)."""
    rng = np.random.default_rng(nx + ns)
    x = rng.standard_normal((nx, ns))
    m = rng.random((nx, ns))
    ref = orc.fk_filter_filt(x, m)
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    for opts in (None, (-1, 0, 0, 0, 0, 0)):
        plan = dw.dsp.FkPlan(nx, ns, opts=opts)
        plan.set_mask(m)
        assert rel(plan.apply(xt).cpu().numpy(), ref) < TOL
        assert rel(plan.apply(xt, taper=True).cpu().numpy(), orc.fk_filter_filt(x, m, tapering=True)) < TOL


def test_config1_block_vs_oracle(dw):
    """BASELINE configs[0]/[1] geometry: 4000 channels x 12000 samples, scripts' hybrid_ninf mask
    and the classic fan, synthetic OOI-like block (SURVEY 8d 'S-small')."""
    nx, ns, fs, dx = 4000, 12000, 200.0, 2.0419046878814697
    sel = [9794, 25794, 4]
    x = orc.synth_block(nx, ns, fs=fs, dx=dx, step=4, seed=1234, n_calls=6, n_waves=10) * 1e9
    m = orc.hybrid_ninf_filter_design((nx, ns), sel, dx, fs, cs_min=1350., cp_min=1450., cp_max=3300,
                                      cs_max=3450, fmin=14., fmax=30.)
    ref = orc.fk_filter_filt(x, m)
    y = dw.dsp.fk_filter_sparsefilt(x, m)
    e = rel(y, ref)
    print("config-1 block, hybrid_ninf: rel err %.3e" % e)
    assert e < TOL
    # tensor in -> tensor out, no host round trip
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    yt = dw.dsp.fk_filter_filt(xt, m)
    assert isinstance(yt, torch.Tensor) and yt.is_cuda and yt.dtype == torch.float32
    assert rel(yt.cpu().numpy(), ref) < TOL


def test_plane_wave_known_answer(dw):
    """a plane wave inside the speed fan passes, one outside is removed."""
    nx, ns, fs, dx = 1000, 2400, 200.0, 8.0
    m = orc.fk_filter_design((nx, ns), [0, nx, 1], dx, fs)
    t = np.arange(ns) / fs
    xpos = np.arange(nx) * dx
    f0 = 20.0

    def wave(c):
        k0 = np.round(f0 / c * nx * dx) / (nx * dx)          # on-grid wavenumber
        return np.cos(2 * np.pi * (f0 * t[None, :] - k0 * xpos[:, None]))
    y_in = dw.dsp.fk_filter_filt(wave(2000.0), m)
    y_out = dw.dsp.fk_filter_filt(wave(500.0), m)
    assert rel(y_in, wave(2000.0)) < 1e-4
    assert np.max(np.abs(y_out)) < 1e-4


def test_full_size_properties(dw):
    """BASELINE configs[2] shape 20000 x 120000 (9.6 GB): all-pass mask == identity, linearity,
    and zero mask == zero -- size-independent properties, no oracle at this size."""
    nx, ns = 20000, 120000
    free, _ = torch.cuda.mem_get_info()
    if free < 45e9:
        pytest.skip("needs ~45 GB of HBM")
    plan = dw.dsp.FkPlan(nx, ns)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    x = torch.randn((nx, ns), dtype=torch.float32, device="cuda", generator=gen)
    ones = torch.ones((nx, ns), dtype=torch.float32, device="cuda")
    plan.set_mask(ones)
    del ones
    y = plan.apply(x)
    err = float((y - x).abs().max() / x.abs().max())
    print("20000x120000 identity-mask error %.3e" % err)
    assert err < TOL
    # linearity: F(2x) == 2 F(x) with a non-trivial (half-strength) mask
    half = torch.full((nx, ns), 0.5, dtype=torch.float32, device="cuda")
    plan.set_mask(half)
    plan.apply(x, out=y)
    err2 = float((y - 0.5 * x).abs().max() / x.abs().max())
    assert err2 < TOL
    # the specialised kernels (default plan at this shape) against the generic five passes on a
    # random, non-Hermitian mask: two independent implementations of the same filter
    del half
    gen.manual_seed(6)
    mask = torch.rand((nx, ns), dtype=torch.float32, device="cuda", generator=gen)
    plan.set_mask(mask)
    plan.apply(x, out=y)
    plan_g = dw.dsp.FkPlan(nx, ns, opts=(-1, 0, 0, 0, 0, 0))
    plan_g.set_mask(mask)
    del mask
    yg = plan_g.apply(x)
    err3 = float((y - yg).abs().max() / yg.abs().max())
    print("20000x120000 specialised vs generic kernels: %.3e" % err3)
    assert err3 < TOL
    # dead-row skipping: the speed fan of fk_filter_design keeps ~30 % of the wavenumber rows; the
    # pruned run must equal the generic (unpruned) kernels' result
    del yg
    fan = dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], 2.0419046878814697, 200.0)
    plan.set_mask(fan)
    live = plan.live_rows()
    print("20000x120000 classic fan: %d of %d wavenumber rows live" % (live, nx))
    assert 0 < live < nx
    plan.apply(x, out=y)
    plan_g.set_mask(fan)
    del fan
    yg = plan_g.apply(x)
    err4 = float((y - yg).abs().max() / yg.abs().max())
    print("20000x120000 pruned specialised vs generic, classic fan: %.3e" % err4)
    assert err4 < TOL


def test_apply_stats_bench_shape(dw):
    """20 000 x 120 000: the row mean / max|.| formed in the last pass's epilogue equal the statistics of
    the filtered block (what detect.compute_cross_correlogram normalises by), the filter output is
    unchanged, and the matched filter fed with them matches the one that computes its own."""
    import torch
    nx, ns = 20000, 120000
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((nx, ns), device="cuda", generator=gen) + 0.25
    plan = dw.dsp.get_fk_plan(nx, ns)
    mask = dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], 2.0419046878814697, 200.0)
    plan.set_mask(mask)
    del mask
    y0 = plan.apply(x)
    y1, mean, mx = plan.apply_stats(x)
    assert torch.equal(y0, y1)
    ref_mean = y1.double().mean(dim=1)
    ref_max = y1.abs().amax(dim=1)
    scale = float(ref_max.max())
    assert float((mean.double() - ref_mean).abs().max()) < 1e-6 * scale
    assert torch.allclose(mx, ref_max, rtol=1e-6, atol=0)
    import numpy as np
    t = np.arange(ns) / 200.0
    hf = dw.detect.gen_template_fincall(t, 200.0, 17.8, 28.8, 0.68)
    taps = [dw.detect._normalised_support(hf)]
    rows = y1[:256].contiguous()
    c_own = dw.detect._xcorr_device(rows, taps, normalize=True)[0]
    c_fed = dw.detect._xcorr_device(rows, taps, normalize=True, stats=(mean[:256].contiguous(), mx[:256].contiguous()))[0]
    assert float((c_own - c_fed).abs().max()) < 2e-6 * float(c_own.abs().max())


@pytest.mark.parametrize("nx,ns", [(4000, 12000), (11020, 12000), (5510, 12000)])
def test_config_shapes_specialised_vs_generic(dw, nx, ns):
    """The 60-s file shapes (BASELINE configs[0..1]; the real OOI channel count 11020 = 20 x 19 x 29)
    run shape-specialised kernels: same result as the generic five passes, pruned and unpruned (the
    independent known answers at these shapes are in tests/test_fk_known_gpu.py)."""
    gen = torch.Generator(device="cuda").manual_seed(nx)
    x = torch.randn((nx, ns), device="cuda", generator=gen)
    dense = torch.rand((nx, ns), device="cuda", generator=gen)
    fan = dw.dsp.fk_filter_design((nx, ns), [0, nx * 4, 4], 2.0419046878814697, 200.0)
    fast, generic = dw.dsp.FkPlan(nx, ns), dw.dsp.FkPlan(nx, ns, opts=(-1, 0, 0, 0, 0, 0))
    for m in (dense, fan):
        fast.set_mask(m)
        generic.set_mask(m)
        y1, y2 = fast.apply(x), generic.apply(x)
        err = float((y1 - y2).abs().max()) / float(y2.abs().max())
        print("%d x %d specialised vs generic (%d live rows): %.3e" % (nx, ns, fast.live_rows(), err))
        assert err < 3e-6
    y1, mean, mx = fast.apply_stats(x)
    assert torch.allclose(mx, y1.abs().amax(dim=1), rtol=1e-6, atol=0)
    assert float((mean.double() - y1.double().mean(dim=1)).abs().max()) < 1e-6 * float(mx.max())
    ident = torch.ones((nx, ns), device="cuda")
    fast.set_mask(ident)
    assert float((fast.apply(x) - x).abs().max()) / float(x.abs().max()) < TOL


@pytest.mark.parametrize("nx,ns", [(300, 12002), (40, 2 * 3 * 1009), (74, 2 * 37 * 5)])
def test_record_length_with_large_prime_factor(dw, nx, ns):
    """ns / 2 with a prime factor > 31 -- numpy.fft.fft2 (dsp.py:748) takes any length: pass B runs Bluestein
    convolutions for the n2 sub-transforms.  12002 = 2 x 17 x 353 is a 60-s file cut two samples long."""
    rng = np.random.default_rng(ns)
    x = rng.standard_normal((nx, ns))
    m = orc.hybrid_ninf_filter_design((nx, ns), [0, nx * 4, 4], 2.0419046878814697, 200.0, 1350., 1450., 3300, 3450, 14., 30.)
    assert rel(dw.dsp.fk_filter_filt(x, m), orc.fk_filter_filt(x, m)) < TOL
    assert rel(dw.dsp.fk_filter_filt(x, m, tapering=True), orc.fk_filter_filt(x, m, tapering=True)) < TOL


def test_benchmark_step_is_the_public_calls_and_matches_the_oracle(dw):
    """bench.py's timed step since round 6: dsp.fk_filter_filt(x, mask) -> detect.compute_cross_correlograms(y, [hf, lf]) on a
    device-resident 20 000 x 120 000 block, full-length zero-padded templates (reference scripts/main_mfdetect.py:55-80).  The
    filter's last pass leaves the rows' mean / max|.| for the matched filter (detect._remember_row_stats), the correlator adds
    the template's DC tail (detect.py:158) itself.  Five rows of the composition against the oracle's
    compute_cross_correlogram of the same filtered rows in float64; then a drifting and a stepped row written INTO the filtered
    block (the remembered statistics must be dropped, the tail term -- 1e-3 of such a row's correlogram -- must be there)."""
    import numpy as np
    import torch
    from oracle import d4w_oracle as orc
    nx, ns, fs = 20000, 120000, 200.0
    free, _ = torch.cuda.mem_get_info()
    if free < 60e9:
        pytest.skip("needs ~60 GB of HBM")
    gen = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn((nx, ns), device="cuda", generator=gen) + 0.1
    mask = dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], 2.0419046878814697, fs)
    t = np.arange(ns) / fs
    hf = dw.detect.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
    lf = dw.detect.gen_template_fincall(t, fs, 14.7, 21.8, 0.78)
    y = dw.dsp.fk_filter_filt(x, mask)
    del x
    # the statistics travel with the result ...
    mean, mx = dw.detect._row_stats_cached(y)
    assert torch.allclose(mx, y.abs().amax(dim=1), rtol=1e-6, atol=0)
    assert float((mean - y.double().mean(dim=1)).abs().max()) < 1e-6 * float(mx.max())
    c_hf, c_lf = dw.detect.compute_cross_correlograms(y, [hf, lf])
    rows = [0, 1, 7777, 10000, nx - 1]
    yr = y[rows].double().cpu().numpy()
    for c, tp, name in ((c_hf, hf, "HF"), (c_lf, lf, "LF")):
        ref = orc.compute_cross_correlogram(yr, tp)
        got = c[rows].cpu().numpy()
        for k in range(len(rows)):
            e = float(np.max(np.abs(got[k] - ref[k])) / np.max(np.abs(ref[k])))
            assert e < 2e-6, (name, rows[k], e)
    del c_hf, c_lf
    # ... and are dropped when the block is written to: rows that drift / step, through the same public call
    tt = torch.arange(ns, device="cuda", dtype=torch.float32) / fs
    y[5] = 0.05 * y[5] + torch.sin(2 * np.pi * tt / 300.0)
    y[6] = torch.where(tt < 200.0, 1.0, -1.0) + 0.01 * y[6]
    c_hf, c_lf = dw.detect.compute_cross_correlograms(y, [hf, lf])
    rows = [4, 5, 6]
    yr = y[rows].double().cpu().numpy()
    for c, tp, name in ((c_hf, hf, "HF"), (c_lf, lf, "LF")):
        ref = orc.compute_cross_correlogram(yr, tp)
        got = c[rows].cpu().numpy()
        for k in range(len(rows)):
            e = float(np.max(np.abs(got[k] - ref[k])) / np.max(np.abs(ref[k])))
            # (the smooth rows' correlograms are small against the rows themselves -- the templates have no response at 1 / 300 Hz --
            # so float32 rounding of the products weighs more against the row's own maximum: 5e-6 here; without the tail term 1e-3)
            assert e < 1e-5, (name, "after the rows were rewritten", rows[k], e)
