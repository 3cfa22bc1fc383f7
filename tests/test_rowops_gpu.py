"""GPU parity tests of the row operators: zero-phase band-pass / sosfiltfilt and the matched
filter, through the C ABI, vs the reference's golden outputs and the CPU oracle.

Tolerance (north star): max|y - y_ref| <= 1e-5 * max|y_ref| with float32 arithmetic."""
import numpy as np
import pytest
import torch

from oracle import d4w_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5
FS = 200.0


def rel(y, ref):
    return float(np.max(np.abs(np.asarray(y, dtype=np.float64) - ref)) / np.max(np.abs(ref)))


@pytest.fixture(scope="module")
def dw():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import das4whales_amd as dw_
    return dw_


# ------------------------------------------------------------------------------------------
# band-pass
# ------------------------------------------------------------------------------------------
def test_bp_and_sos_golden(dw, golden):
    g = golden("fk_40x480.npz")
    x = g["x"]
    y = dw.dsp.bp_filt(x, FS, 14, 30)
    assert y.dtype == np.float64 and y.shape == x.shape
    assert rel(y, g["y_bp"]) < TOL
    sos_hp = dw.dsp.butterworth_filter([2, 5, "hp"], FS)
    sos_bp = dw.dsp.butterworth_filter([5, [10, 30], "bp"], FS)
    assert np.array_equal(sos_hp, g["sos_hp"]) and np.array_equal(sos_bp, g["sos_bp"])
    assert rel(dw.dsp.sosfiltfilt(sos_hp, x, axis=1), g["y_sos_hp"]) < TOL
    assert rel(dw.dsp.sosfiltfilt(sos_bp, x, axis=1), g["y_sos_bp"]) < TOL
    assert rel(dw.dsp.bp_filt(x[3], FS, 14, 30), g["y_bp"][3]) < TOL          # 1-D input
    with pytest.raises(ValueError, match="padlen, which is 51"):
        dw.dsp.bp_filt(x[:, :40], FS, 14, 30)


def test_bp_config1_block_segmented(dw):
    """4000 x 12000 (BASELINE configs[1] geometry): segmented rows vs the float64 filtfilt oracle."""
    nx, ns = 4000, 12000
    x = orc.synth_block(nx, ns, fs=FS, step=4, seed=1234, n_calls=6, n_waves=10) * 1e9
    y = dw.dsp.bp_filt(x, FS, 14, 30)
    rows = np.r_[0:64, 1990:2010, nx - 64:nx]
    ref = orc.bp_filt(x[rows], FS, 14, 30)
    e = rel(y[rows], ref)
    print("bp_filt 4000x12000: rel err %.3e" % e)
    assert e < TOL
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    yt = dw.dsp.bp_filt(xt, FS, 14, 30)
    assert isinstance(yt, torch.Tensor) and yt.is_cuda
    assert np.allclose(yt.cpu().numpy(), y.astype(np.float32), atol=0, rtol=0)   # deterministic


def test_bp_known_answer_tones(dw):
    """a 20 Hz tone passes with |H|^2 ~ 1, a 5 Hz tone is removed."""
    t = np.arange(24000) / FS
    x = np.stack([np.sin(2 * np.pi * 20 * t), np.sin(2 * np.pi * 5 * t)])
    y = dw.dsp.bp_filt(x, FS, 14, 30)
    mid = slice(4000, 20000)
    assert np.max(np.abs(y[0, mid] - x[0, mid])) < 2e-3          # pass band: zero phase, unit gain
    assert np.max(np.abs(y[1, mid])) < 1e-5


def test_bp_full_size_rows_vs_oracle(dw):
    """20000 x 120000: rows are independent, so a row subset is checked against the oracle."""
    nx, ns = 20000, 120000
    free, _ = torch.cuda.mem_get_info()
    if free < 40e9:
        pytest.skip("needs ~40 GB of HBM")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    x = torch.randn((nx, ns), dtype=torch.float32, device="cuda", generator=gen)
    y = dw.dsp.bp_filt(x, FS, 14, 30)
    rows = [0, 1, 63, 64, 7777, 12345, nx - 65, nx - 1]
    ref = orc.bp_filt(x[rows].cpu().numpy().astype(np.float64), FS, 14, 30)
    e = rel(y[rows].cpu().numpy(), ref)
    print("bp_filt 20000x120000 (8 rows): rel err %.3e" % e)
    assert e < TOL


def test_bp_row_ends_on_a_side_stream(dw, monkeypatch):
    """Long rows: the recursion on the row ends runs on a side stream underneath the overlap-save pass (dsp._sosfiltfilt_fft).
    Same answer, bit for bit, as with everything on the calling stream -- repeatedly, on the default stream and on a user
    stream, with the input freed right after the call."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(12)
    nx, ns = 256, 60000
    x = torch.randn((nx, ns), dtype=torch.float32, device="cuda", generator=gen)
    monkeypatch.setenv("D4W_BP_OVERLAP", "0")
    y0 = dw.dsp.bp_filt(x, FS, 14, 30)
    monkeypatch.setenv("D4W_BP_OVERLAP", "1")
    for _ in range(3):
        assert torch.equal(dw.dsp.bp_filt(x, FS, 14, 30), y0)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        outs = []
        for _ in range(3):
            xc = x.clone()
            outs.append(dw.dsp.bp_filt(xc, FS, 14, 30))
            del xc
    torch.cuda.current_stream().wait_stream(st)
    assert all(torch.equal(o, y0) for o in outs)
    assert rel(y0[:4].cpu().numpy(), orc.bp_filt(x[:4].cpu().numpy().astype(np.float64), FS, 14, 30)) < TOL


# ------------------------------------------------------------------------------------------
# matched filter
# ------------------------------------------------------------------------------------------
def test_templates_and_correlogram_golden(dw, golden):
    d = golden("detect_12x2000.npz")
    time = np.arange(2000) / FS
    hf = dw.detect.gen_template_fincall(time, FS, fmin=17.8, fmax=28.8, duration=0.68)
    lf = dw.detect.gen_template_fincall(time, FS, fmin=14.7, fmax=21.8, duration=0.78)
    assert np.allclose(hf, d["hf"], atol=1e-12) and np.allclose(lf, d["lf"], atol=1e-12)
    assert np.allclose(dw.detect.gen_linear_chirp(15., 25., 1.0, FS), d["lin_chirp"], atol=1e-12)
    assert np.allclose(dw.detect.gen_hyperbolic_chirp(15., 25., 1.0, FS), d["hyp_chirp"], atol=1e-12)
    assert np.allclose(dw.detect.gen_template_fincall(time, FS, 15., 25., 1.0, window=False), d["tpl_nowin"], atol=1e-12)
    x = d["x"]
    c_hf = dw.detect.compute_cross_correlogram(x, hf)
    assert c_hf.dtype == np.float64 and c_hf.shape == x.shape
    assert rel(c_hf, d["corr_hf"]) < TOL
    c2 = dw.detect.compute_cross_correlograms(x, [hf, lf])
    assert rel(c2[0], d["corr_hf"]) < TOL and rel(c2[1], d["corr_lf"]) < TOL
    assert rel(dw.detect.shift_xcorr(x[2], hf), d["xc"]) < TOL
    assert rel(dw.detect.shift_nxcorr(x[2], hf), d["nxc"]) < TOL
    # integer input (reference tests/test_detect.py:66-67 passes ints): floating result
    ci = dw.detect.compute_cross_correlogram(np.arange(20).reshape(2, 10), np.arange(10))
    assert ci.shape == (2, 10) and ci.dtype == np.float64
    assert rel(ci, orc.compute_cross_correlogram(np.arange(20).reshape(2, 10), np.arange(10))) < TOL
    # convert / select are index bookkeeping
    idx = dw.detect.convert_pick_times([np.array([3, 9]), np.array([], dtype=int), np.array([5])])
    assert np.array_equal(idx, np.array([[0, 0, 2], [3, 9, 5]]))


def test_correlogram_known_answer(dw):
    """a row holding the template at lag tau peaks there with value sum(tpl^2)/(A max|x|)."""
    ns, tau = 4000, 1234
    time = np.arange(ns) / FS
    hf = dw.detect.gen_template_fincall(time, FS, 17.8, 28.8, 0.68)
    L = 136
    x = np.zeros((3, ns))
    x[1, tau:tau + L] = 0.5 * hf[:L]
    c = dw.detect.compute_cross_correlogram(x, hf)
    assert np.all(c[0] == 0) and np.all(c[2] == 0)                 # all-zero rows -> zeros (documented)
    assert int(np.argmax(c[1])) == tau
    expect = 0.5 * np.sum(hf[:L] ** 2) / (np.max(np.abs(hf)) * np.max(np.abs(x[1])))
    assert abs(c[1, tau] - expect) / expect < 1e-4


def test_strict_reference_switch(dw):
    """das4whales_amd.set_strict_reference(True): what the reference does where the package's defaults differ on purpose --
    an all-zero channel correlates to NaN (0 / 0 at detect.py:157; the oracle, restating it, gives NaN too) and tapering=True
    tapers the caller's array in place (dsp.py:744-745).  Also as per-call options, and off again afterwards."""
    import torch
    import warnings
    ns = 2000
    time = np.arange(ns) / FS
    hf = dw.detect.gen_template_fincall(time, FS, 17.8, 28.8, 0.68)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((4, ns))
    x[2] = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = orc.compute_cross_correlogram(x, hf)
    assert np.all(np.isnan(ref[2]))
    c = dw.detect.compute_cross_correlogram(x, hf)
    assert np.all(c[2] == 0) and rel(c[[0, 1, 3]], ref[[0, 1, 3]]) < TOL
    cn = dw.detect.compute_cross_correlogram(x, hf, zero_rows="nan")
    assert np.all(np.isnan(cn[2])) and np.array_equal(cn[[0, 1, 3]], c[[0, 1, 3]])
    with pytest.raises(ValueError):
        dw.detect.compute_cross_correlogram(x, hf, zero_rows="inf")
    xt = torch.from_numpy(rng.standard_normal((64, 480)).astype(np.float32)).cuda()
    mask = dw.dsp.fk_filter_design((64, 480), [0, 64, 1], 2.04, FS)
    before = xt.clone()
    assert dw.set_strict_reference(True) is False
    try:
        cs = dw.detect.compute_cross_correlograms(x, [hf, hf])
        assert all(np.all(np.isnan(ci[2])) and np.array_equal(ci[[0, 1, 3]], c[[0, 1, 3]]) for ci in cs)
        y_strict = dw.dsp.fk_filter_filt(xt, mask, tapering=True)
        tapered = before.clone()
        dw.dsp.taper_data(tapered)
        assert torch.equal(xt, tapered) and not torch.equal(xt, before)
    finally:
        assert dw.set_strict_reference(False) is True
    x2 = before.clone()
    y = dw.dsp.fk_filter_filt(x2, mask, tapering=True)
    assert torch.equal(x2, before)                                  # default: the caller's array is left alone
    assert float((y - y_strict).abs().max()) <= 2e-6 * float(y.abs().max())
    assert np.all(dw.detect.compute_cross_correlogram(x, hf)[2] == 0)


def test_correlogram_config1_block(dw):
    nx, ns = 4000, 12000
    x = orc.synth_block(nx, ns, fs=FS, step=4, seed=99, n_calls=6, n_waves=4) * 1e9
    time = np.arange(ns) / FS
    hf = dw.detect.gen_template_fincall(time, FS, 17.8, 28.8, 0.68)
    lf = dw.detect.gen_template_fincall(time, FS, 14.7, 21.8, 0.78)
    c_hf, c_lf = dw.detect.compute_cross_correlograms(x, [hf, lf])
    rows = np.r_[0:16, 2000:2016, nx - 16:nx]
    e1 = rel(c_hf[rows], orc.compute_cross_correlogram(x[rows], hf))
    e2 = rel(c_lf[rows], orc.compute_cross_correlogram(x[rows], lf))
    print("correlogram 4000x12000: rel err HF %.3e LF %.3e" % (e1, e2))
    assert e1 < TOL and e2 < TOL


def test_correlogram_full_size_rows_vs_oracle(dw):
    nx, ns = 20000, 120000
    free, _ = torch.cuda.mem_get_info()
    if free < 40e9:
        pytest.skip("needs ~40 GB of HBM")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(12)
    x = torch.randn((nx, ns), dtype=torch.float32, device="cuda", generator=gen)
    time = np.arange(ns) / FS
    hf = dw.detect.gen_template_fincall(time, FS, 17.8, 28.8, 0.68)
    lf = dw.detect.gen_template_fincall(time, FS, 14.7, 21.8, 0.78)
    c_hf, c_lf = dw.detect.compute_cross_correlograms(x, [hf, lf])
    rows = [0, 1, 4999, 10000, nx - 1]
    xr = x[rows].cpu().numpy().astype(np.float64)
    e1 = rel(c_hf[rows].cpu().numpy(), orc.compute_cross_correlogram(xr, hf))
    e2 = rel(c_lf[rows].cpu().numpy(), orc.compute_cross_correlogram(xr, lf))
    print("correlogram 20000x120000 (5 rows): rel err HF %.3e LF %.3e" % (e1, e2))
    assert e1 < TOL and e2 < TOL


def test_correlogram_matrix_core_form_is_float32_grade(dw):
    """The default form (csrc/xcorr_mm.hip: binary16 hi / lo splits on the matrix cores) against the float64 oracle at a
    tolerance ten times below the north star's, on EVERY sample of a block large enough to meet the rare cases (a hi / lo
    pair that disagrees by one binary16 ulp is a 1e-4 error on one sample in ~10^4: 3.2 M samples would hold hundreds),
    and against the two float32 forms (overlap-save FFT, direct FIR)."""
    nx, ns = 64, 50000
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    x = torch.randn((nx, ns), dtype=torch.float32, device="cuda", generator=gen) * 3.0 + 0.7
    time = np.arange(ns) / FS
    tpl = [dw.detect._normalised_support(dw.detect.gen_template_fincall(time, FS, 17.8, 28.8, 0.68)),
           dw.detect._normalised_support(dw.detect.gen_template_fincall(time, FS, 14.7, 21.8, 0.78))]
    ym = dw.detect._xcorr_device(x, tpl, normalize=True, method="mm")
    yf = dw.detect._xcorr_device(x, tpl, normalize=True, method="fft")
    yd = dw.detect._xcorr_device(x, tpl, normalize=True, method="direct")
    xs = x.double().cpu().numpy()
    xn = (xs - xs.mean(axis=1, keepdims=True)) / np.abs(xs).max(axis=1, keepdims=True)
    for k in range(2):
        ref = np.stack([np.correlate(np.concatenate((r, np.zeros(len(tpl[k]) - 1))), tpl[k], "valid") for r in xn])
        em, ef, ed = (rel(y[k].cpu().numpy(), ref) for y in (ym, yf, yd))
        print("template %d: matrix cores %.2e, FFT %.2e, direct %.2e vs float64" % (k, em, ef, ed))
        assert em < 1e-6 and ef < 1e-6 and ed < 2e-6
    (y1,) = dw.detect._xcorr_device(x, tpl[1:], normalize=True, method="mm")          # the one-template kernel
    assert float((y1 - ym[1]).abs().max()) <= 1e-6 * float(ym[1].abs().max())
    # no row statistics from the caller: every chunk scales itself by a power of two
    yr = dw.detect._xcorr_device(x, tpl, normalize=False, method="mm")
    rd = dw.detect._xcorr_device(x, tpl, normalize=False, method="direct")
    assert float((yr[0] - rd[0]).abs().max()) <= 2e-6 * float(rd[0].abs().max())
    # a high-dynamic-range block: one spike of 10^4 x the noise per row, so that 1 / max|x| pushes the ordinary samples into
    # binary16's subnormal range (the hi half keeps ~7 bits there, the lo half the rest: absolute error 2^-24 / 2048 of the
    # row maximum).  Every sample again, and the lags AWAY from the spike separately -- relative to their own maximum, where
    # the spike's large correlation values do not hide an error of the small ones
    xs2 = x.clone()
    pos = torch.arange(nx, device="cuda") * 613 % (ns - 2000) + 1000
    xs2[torch.arange(nx, device="cuda"), pos] = 3.0e4
    ym2 = dw.detect._xcorr_device(xs2, tpl, normalize=True, method="mm")
    xs = xs2.double().cpu().numpy()
    xn = (xs - xs.mean(axis=1, keepdims=True)) / np.abs(xs).max(axis=1, keepdims=True)
    for k in range(2):
        L = len(tpl[k])
        ref = np.stack([np.correlate(np.concatenate((r, np.zeros(L - 1))), tpl[k], "valid") for r in xn])
        got = ym2[k].cpu().numpy()
        assert rel(got, ref) < 1e-6
        far = np.ones((nx, ns), dtype=bool)
        for c, p in enumerate(pos.cpu().numpy()):
            far[c, max(0, p - L):p + 1] = False
        e_far = np.max(np.abs(got - ref)[far]) / np.max(np.abs(ref[far]))
        print("spike block, template %d: lags away from the spike %.2e of their own maximum" % (k, e_far))
        assert e_far < 1e-5


@pytest.mark.parametrize("offset", [1e3, 1e4, 1e5, 1e6])
def test_offset_heavy_rows_hold_the_bar_on_every_form(dw, offset):
    """detect.py:157 de-means in float64 and accepts any row.  Rows whose offset is 10^3 .. 10^5 x their signal's deviation and a
    template with a non-zero sum (so that an error of the mean enters every lag): the float64 row means of d4w_row_stats_f32
    enter the correlators as two-float values and all three forms stay inside 1e-5 of a float64 correlation of the same
    float32 rows -- round 4's known limit (1.2 - 2.8e-5 at 1000 x with a float32 mean).  Also through the public call."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(int(offset) % 1000 + 5)
    nx, ns, L = 24, 30000, 137
    sig = 0.37
    sign = torch.tensor([1.0, -1.0, 0.731] * (nx // 3), device="cuda")[:, None]
    x = torch.randn((nx, ns), dtype=torch.float32, device="cuda", generator=gen) * sig + float(offset) * sig * sign
    rng = np.random.default_rng(int(offset))
    tpl = np.abs(rng.standard_normal(L)) + 0.3
    xs = x.double().cpu().numpy()
    xn = (xs - xs.mean(axis=1, keepdims=True)) / np.abs(xs).max(axis=1, keepdims=True)
    ref = np.stack([np.correlate(np.concatenate((r, np.zeros(L - 1))), tpl, "valid") for r in xn])
    mean, mx = dw.detect._row_stats_cached(x)
    assert mean.dtype == torch.float64
    assert float((mean.cpu() - torch.from_numpy(xs.mean(axis=1))).abs().max()) < 1e-7 * sig
    for how in ("mm", "fft", "direct"):
        (y,) = dw.detect._xcorr_device(x, [tpl], normalize=True, method=how)
        e = rel(y.cpu().numpy(), ref)
        print("offset %g, %s: %.2e" % (offset, how, e))
        assert e < TOL, (how, offset, e)
    # the public call: template zero-padded to the row length (its DC tail, detect.py:158, is added when it matters)
    tfull = np.zeros(ns)
    tfull[:L] = tpl
    got = dw.detect.compute_cross_correlogram(x, tfull).cpu().numpy()
    assert rel(got, orc.compute_cross_correlogram(xs, tfull)) < TOL


def test_zero_padded_template_on_rows_that_drift(dw):
    """detect.py:158 de-means the template over its zero-padded length, which leaves -mean/max on the padded part: a term of
    |coef| g x (a prefix sum of the de-meaned row).  3e-6 of the correlogram on white rows, 3e-8 on band-passed rows -- and
    3e-4 .. 1e-3 on rows with a slow drift or a step, which the rule of rounds 1-4 (a prediction from the template, white rows
    assumed) let pass.  The public call with its defaults against the oracle on white, band-limited, drifting and stepped
    rows, per row (every row against ITS OWN maximum), for one and for two templates; the rows that keep the term out are
    the band-limited ones only."""
    import scipy.signal as sps
    rng = np.random.default_rng(121)
    ns = 12000
    t = np.arange(ns) / FS
    hf = orc.gen_template_fincall(t, FS, 17.8, 28.8, 0.68)
    lf = orc.gen_template_fincall(t, FS, 14.7, 21.8, 0.78)
    sos = sps.butter(8, [14 / (FS / 2), 30 / (FS / 2)], "bp", output="sos")
    w = rng.standard_normal((40, ns))
    rows = []
    for k in range(10):
        rows += [w[4 * k], sps.sosfiltfilt(sos, w[4 * k + 1]), 0.05 * w[4 * k + 2] + np.sin(2 * np.pi * t / (20.0 + 7 * k) + k),
                 np.where(t < 5.0 * (k + 1), 1.0, -1.0) + 0.01 * w[4 * k + 3]]
    x = np.stack(rows) + 0.2
    nx = len(x)
    xs = x.astype(np.float32)
    refs = [orc.compute_cross_correlogram(xs.astype(np.float64), tp) for tp in (hf, lf)]
    per_row = lambda y, ref: np.max(np.abs(np.asarray(y, dtype=np.float64) - ref), axis=1) / np.max(np.abs(ref), axis=1)
    xd = torch.from_numpy(xs).cuda()
    got = dw.detect.compute_cross_correlograms(xd, [hf, lf])
    for y, ref in zip(got, refs):
        e = per_row(y.cpu().numpy(), ref)
        print("zero-padded template, worst row error by kind (white, band, drift, step): %s" % ["%.1e" % e[k::4].max() for k in range(4)])
        assert e.max() < TOL, e
    one = dw.detect.compute_cross_correlogram(xs, hf)                        # host in, host out
    assert per_row(one, refs[0]).max() < TOL
    # what the term is worth here
    bare = dw.detect.compute_cross_correlograms(xd, [hf, lf], exact_tail=False)
    e0 = per_row(bare[0].cpu().numpy(), refs[0])
    assert e0[2::4].min() > 1e-4 and e0[3::4].min() > 1e-4 and e0[1::4].max() < 1e-6, e0
    # round 6: the default adds the term inside the correlator (d4w_xcorr_mm_tail_f32), on EVERY row -- no row keeps the bare value,
    # the band-limited ones move by less than 1e-6 of their maximum
    same = (bare[0] == got[0]).all(dim=1).cpu().numpy()
    assert not same.any(), same
    moved = per_row(got[0].cpu().numpy(), bare[0].cpu().numpy().astype(np.float64))
    assert moved[1::4].max() < 1e-6 and moved[2::4].min() > 1e-4, moved
    full = dw.detect.compute_cross_correlograms(xd, [hf, lf], exact_tail=True)
    assert per_row(full[0].cpu().numpy(), refs[0]).max() < TOL
    # ... and the two-pass form of rounds 1-5 (prefix maxima + d4w_xcorr_dc_tail_rows_f32, decided per row) still stands behind
    # D4W_XCORR_TAIL=pass: the rows it leaves alone are the band-limited ones only, and it agrees with the in-kernel form
    import os
    os.environ["D4W_XCORR_TAIL"] = "pass"
    try:
        old = dw.detect.compute_cross_correlograms(xd, [hf, lf])
    finally:
        del os.environ["D4W_XCORR_TAIL"]
    same_old = (bare[0] == old[0]).all(dim=1).cpu().numpy()
    assert same_old[1::4].all() and not same_old[0::4].any() and not same_old[2::4].any() and not same_old[3::4].any(), same_old
    for a, b, ref in zip(old, got, refs):
        assert per_row(a.cpu().numpy(), ref).max() < TOL
        assert per_row(a.cpu().numpy(), b.cpu().numpy().astype(np.float64)).max() < 3e-6


def test_row_statistics_are_not_reused_after_a_raw_pointer_write(dw):
    """the library writes through raw pointers, which torch's version counter does not see -- `plan.apply(x_i,
    out=y); compute_cross_correlogram(y, tpl)` in a per-file loop must not find file 0's row statistics on file 1.  Every
    wrapper that writes into a caller's tensor bumps its version (dev.out_ptr)."""
    from das4whales_amd import detect as ddet
    nx, ns = 96, 480
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    plan = dw.dsp.get_fk_plan(nx, ns)
    plan.set_mask(np.ones((nx, ns)))
    y = torch.empty((nx, ns), dtype=torch.float32, device="cuda")
    time = np.arange(ns) / FS
    hf = dw.detect.gen_template_fincall(time, FS, 17.8, 28.8, 0.68)
    outs = []
    for k in range(2):
        x = torch.randn((nx, ns), device="cuda", generator=gen) * (1.0 + 3.0 * k) + 0.5 * k
        plan.apply(x, out=y)
        c = dw.detect.compute_cross_correlogram(y, hf)
        ref = orc.compute_cross_correlogram(y.double().cpu().numpy(), hf)
        assert rel(c.cpu().numpy(), ref) < TOL, k
        outs.append(ddet._row_stats_cached(y))
    assert outs[0][0] is not outs[1][0]
    v = y._version
    dw.dsp.taper_data(y)                                                  # in place through a raw pointer
    assert y._version > v


def test_row_statistics_are_remembered_per_tensor_version(dw):
    """Two correlogram calls on the SAME CUDA block (scripts/main_mfdetect.py:79-80: HF, then LF) form the row statistics once;
    an in-place edit of the block (version counter) or another tensor at the same address forms them again."""
    from das4whales_amd import detect as ddet
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    x = torch.randn((50, 6000), device="cuda", generator=gen) + 0.3
    time = np.arange(6000) / FS
    hf = dw.detect.gen_template_fincall(time, FS, 17.8, 28.8, 0.68)
    a = dw.detect.compute_cross_correlogram(x, hf)
    m1 = ddet._row_stats_cached(x)
    assert ddet._row_stats_cached(x)[0] is m1[0]                          # remembered
    x.mul_(2.0).add_(1.0)                                                  # same storage, new version
    m2 = ddet._row_stats_cached(x)
    assert m2[0] is not m1[0] and m2[0].dtype == torch.float64 and torch.allclose(m2[0], x.double().mean(dim=1), atol=1e-9)
    b = dw.detect.compute_cross_correlogram(x, hf)
    ref = orc.compute_cross_correlogram(x.cpu().numpy().astype(np.float64), hf)
    assert rel(b.cpu().numpy(), ref) < TOL and not torch.equal(a, b)


# ------------------------------------------------------------------------------------------
# fused ingest
# ------------------------------------------------------------------------------------------
def test_raw2strain_and_channel_selection(dw):
    """data_handle.raw2strain (data_handle.py:157-176) and the array part of load_das_data
    (data_handle.py:213-228) on an int32 'RawData' matrix."""
    rng = np.random.default_rng(21)
    nch, ns = 300, 12000
    raw = (rng.standard_normal((nch, ns)) * 4e4 + 2e5).astype(np.int32)
    meta = {"scale_factor": 1.7e-11, "fs": 200.0, "dx": 2.0419046878814697}
    sel = [10, 290, 4]
    ref = raw[sel[0]:sel[1]:sel[2]].astype(np.float64)
    ref -= np.mean(ref, axis=1, keepdims=True)
    ref *= meta["scale_factor"]
    y, tx, dist = dw.data_handle.load_das_data_array(raw, sel, meta)
    assert y.is_cuda and tuple(y.shape) == ref.shape
    assert rel(y.cpu().numpy(), ref) < 1e-6
    assert np.allclose(tx, np.arange(ns) / 200.0) and np.allclose(dist, (np.arange(ref.shape[0]) * 4 + 10) * meta["dx"])
    tr = raw[:50].astype(np.float64)
    out = dw.data_handle.raw2strain(tr.copy(), meta)
    exp = (tr - tr.mean(axis=1, keepdims=True)) * meta["scale_factor"]
    assert out.dtype == np.float64 and rel(out, exp) < 1e-6
