"""GPU parity tests of the row spectral operators through the Python mirror of the reference
namespace (which calls the C ABI): analytic signal / SNR / instantaneous frequency / get_fx,
spectrograms, spectrogram correlation and peak picking, vs the reference's golden outputs
(tests/golden/*.npz) and the CPU oracle.

Tolerance (north star): max|y - y_ref| <= 1e-5 * max|y_ref| with float32 arithmetic; picks must be
the identical index sets except peaks whose prominence is within 1e-4 * threshold of the threshold."""
import numpy as np
import pytest
import torch

from oracle import d4w_oracle as orc
from tests.known_answers import assert_picks_match

pytestmark = pytest.mark.gpu
TOL = 1e-5
FS = 200.0
KERNEL = {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}


def rel(y, ref):
    return float(np.max(np.abs(np.asarray(y, dtype=np.float64) - ref)) / np.max(np.abs(ref)))


@pytest.fixture(scope="module")
def dw():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import das4whales_amd as dw_
    return dw_


def test_snr_fx_ifreq_golden(dw, golden):
    g = golden("fk_40x480.npz")
    x = g["x"]
    for env, key in ((False, "snr"), (True, "snr_env")):
        y = dw.dsp.snr_tr_array(x, env=env)
        assert y.shape == x.shape and y.dtype == np.float64
        lin, ref = 10.0 ** (y / 10), 10.0 ** (g[key] / 10)          # power ratios: the quantity under the log
        assert np.max(np.abs(lin - ref)) / np.max(ref) < TOL
        top = g[key] > -40
        assert np.max(np.abs(y[top] - g[key][top])) < 1e-3
    r = golden("ref_test_vectors.npz")                              # reference tests/test_dsp.py:136-141
    assert np.allclose(dw.dsp.snr_tr_array(r["snr_in"]), r["snr_expected"], atol=1e-4)
    assert rel(dw.dsp.get_fx(x[:, :400], 512), g["fx"]) < TOL
    fi = dw.dsp.instant_freq(x[3], FS)
    assert fi.shape == g["ifreq"].shape
    d = np.abs(fi - g["ifreq"])
    d = np.minimum(d, np.abs(d - FS))                               # a +-pi phase step may take either sign
    z = orc.hilbert(x)
    # the angle of z[i+1] conj(z[i]) is conditioned like 1 / |z|: the error is weighed by the smaller envelope
    # sample of the pair relative to the row maximum (float32 analytic signal: absolute error ~1e-7 max|z|)
    w = np.minimum(np.abs(z[3][1:]), np.abs(z[3][:-1])) / np.max(np.abs(z[3]))
    assert np.max(d * w) < TOL * FS / 2
    assert np.max(d[w > 0.05]) < 20 * TOL * FS / 2
    assert rel(dw.dsp.envelope(x), np.abs(z)) < TOL
    assert rel(dw.dsp.hilbert_imag(x[7]), z[7].imag) < TOL
    xo = x[:, :475]                                                 # odd length: full complex transform
    assert rel(dw.dsp.envelope(xo), np.abs(orc.hilbert(xo))) < TOL


def test_envelope_config1_rows(dw):
    """12 000-sample rows (BASELINE configs[1] geometry), CUDA tensor in -> CUDA tensor out."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((300, 12000))
    xt = torch.from_numpy(x).cuda().float()
    e = dw.dsp.envelope(xt)
    assert e.is_cuda and e.dtype == torch.float32
    ref = orc.envelope(x[:40])
    assert rel(e[:40].cpu().numpy(), ref) < TOL


def test_envelope_long_rows(dw):
    """120 000-sample rows (BASELINE configs[2] row length) do not fit one workgroup's LDS: the
    four-step time-axis path through HBM (d4w_analytic_long_f32) gives the same envelope, SNR and picks."""
    import scipy.signal as sps
    rng = np.random.default_rng(8)
    nx, ns = 70, 120000
    x = rng.standard_normal((nx, ns))
    xt = torch.from_numpy(x).cuda().float()
    e = dw.dsp.envelope(xt)
    ref = orc.envelope(x[:6])
    err = rel(e[:6].cpu().numpy(), ref)
    print("envelope 70 x 120000 (6 rows): rel err %.3e" % err)
    assert err < TOL
    s = dw.dsp.snr_tr_array(x[:4], env=True)
    lin, lref = 10.0 ** (s / 10), np.abs(orc.hilbert(x[:4])) ** 2 / np.var(x[:4], axis=1, keepdims=True)
    assert np.max(np.abs(lin - lref)) / np.max(lref) < TOL
    thr = 3.0
    got = dw.detect.pick_times_env(x[:3], thr)
    for c in range(3):
        ndiff, nref = assert_picks_match(got[c], orc.envelope(x[c]), thr, "envelope picks, 120000-sample row %d" % c)
        print("row %d: %d picks, %d marginal differences" % (c, nref, ndiff))


def test_spectrograms_golden(dw, golden):
    g = golden("detect_12x2000.npz")
    x = g["x"]
    p, tt, ff = dw.dsp.get_spectrogram(x[5], FS, nfft=256, overlap_pct=0.95)
    assert p.shape == g["spec_p"].shape
    assert np.allclose(tt, g["spec_tt"]) and np.allclose(ff, g["spec_ff"])
    assert np.max(np.abs(10.0 ** (p / 20) - 10.0 ** (g["spec_p"] / 20))) < TOL
    S, sff, stt = dw.detect.get_sliced_nspectrogram(x[5], FS, 14., 30., 160, 8)
    assert rel(S, g["nspec"]) < TOL
    assert np.allclose(sff, g["nspec_ff"]) and np.allclose(stt, g["nspec_tt"])
    tvec, fvec, ker = dw.detect.buildkernel(27., 17., 4., 0.8, sff, stt, FS, 14., 30.)
    assert np.allclose(ker, g["ker"], rtol=1e-13, atol=1e-15) and np.allclose(tvec, g["ker_tvec"])
    assert rel(dw.detect.buildkernel_from_template(17., 27., 0.8, FS, 160, 8), g["ker_tpl"]) < TOL


def test_spectrocorr_golden(dw, golden):
    g = golden("detect_12x2000.npz")
    assert rel(dw.detect.xcorr2d(g["nspec"], g["ker"]), g["xcorr2d"]) < TOL
    sc = dw.detect.compute_cross_correlogram_spectrocorr(g["x"], FS, [14., 30.], KERNEL, 0.8, 0.95)
    assert sc.shape == g["spectrocorr"].shape
    assert rel(sc, g["spectrocorr"]) < TOL
    ts, cv = dw.detect.xcorr(g["nspec_tt"], g["nspec_ff"], g["nspec"], g["ker_tvec"], g["nspec_ff"], g["ker"])
    assert np.allclose(ts, g["xcorr_t"]) and rel(cv, g["xcorr_v"]) < TOL
    assert rel(dw.detect.nxcorr2d(g["nspec"], g["ker"]), g["nxcorr2d"]) < TOL


def test_spectrocorr_config1_block(dw):
    """4000 x 12000 block (detector settings of scripts/main_spectrodetect.py): all channels in one
    STFT / median / correlation launch vs the per-channel oracle on a row subset."""
    nx, ns = 4000, 12000
    x = orc.synth_block(nx, ns, fs=FS, step=4, seed=1234, n_calls=6, n_waves=10) * 1e9
    sc = dw.detect.compute_cross_correlogram_spectrocorr(x, FS, [14., 30.], KERNEL, 0.8, 0.95)
    assert sc.shape == (nx, 1 + ns // 8)
    rows = [0, 1, 777, 2048, 3999]
    ref = orc.compute_cross_correlogram_spectrocorr(x[rows], FS, [14., 30.], KERNEL, 0.8, 0.95)
    err = rel(sc[rows], ref)
    print("spectrocorr 4000x12000 (5 rows): rel err %.3e" % err)
    assert err < TOL


def _sets(picks):
    return set((int(r), int(t)) for r, row in enumerate(picks) for t in row)


def test_picks_golden(dw, golden):
    g = golden("detect_12x2000.npz")
    thr = float(g["thr"])
    pk = dw.detect.pick_times(g["corr_hf"], thr)
    pe = dw.detect.pick_times_env(g["corr_hf"], thr)
    assert len(pk) == 12 and all(p.dtype == np.int64 for p in pk)
    assert np.array_equal(dw.detect.convert_pick_times(pk), g["picks"])
    assert np.array_equal(dw.detect.convert_pick_times(pe), g["picks_env"])
    assert np.array_equal(dw.detect.process_corr(g["corr_hf"][4], thr), pe[4])
    par = dw.detect.pick_times_par(g["corr_hf"], thr)
    assert all(np.array_equal(a, b) for a, b in zip(par, pe))
    sel = dw.detect.select_picked_times(dw.detect.convert_pick_times(pe), 1.0, 8.0, FS)
    assert np.array_equal(sel[0], g["picks_env_sel0"]) and np.array_equal(sel[1], g["picks_env_sel1"])


def test_picks_random_rows_vs_scipy(dw):
    """Noise rows, plateaus, flat rows, edge maxima; also more peaks than the first capacity guess."""
    import scipy.signal as sps
    rng = np.random.default_rng(5)
    nx, ns = 64, 12000
    x = rng.standard_normal((nx, ns)).astype(np.float32)
    x[1] = np.round(x[1] * 3) / 3
    x[2] = 0.0
    x[4, -1] = 10.0
    x[5, 0] = 10.0
    for thr in (0.0, 1.5, 4.0):
        got = dw.detect.pick_times(x, thr)
        for c in range(nx):
            ref = sps.find_peaks(x[c].astype(np.float64), prominence=thr)[0]
            assert np.array_equal(got[c], ref), (thr, c)


def test_picks_with_few_candidates_vs_scipy(dw):
    """The regime of the detection chain: smooth envelopes, the threshold a fraction of the strongest peak, ~10 maxima worth a
    prominence walk per row (a group of lanes per walk side, hot blocks marked by half a wave each -- spectral.hip fp_scan);
    bursts cut by the row ends, plateaus, rows of 12 000 (staged in LDS) and 20 000 samples (read in place), every row
    against SciPy at several thresholds; the same picks with the threshold formed on the device (detect.Threshold) and
    with the result looked at later (lazy=True)."""
    import scipy.signal as sps
    for ns in (12000, 20000):
        rng = np.random.default_rng(300 + ns)
        nx = 96
        t = np.arange(ns)
        x = np.empty((nx, ns), dtype=np.float32)
        for c in range(nx):
            env = np.convolve(0.05 + 0.02 * np.abs(rng.standard_normal(ns)), np.ones(9) / 9, "same")
            for k in range(int(rng.integers(2, 40))):
                p, w, a = rng.integers(0, ns), rng.uniform(15, 300), rng.uniform(0.2, 1.0)
                env += a * np.exp(-0.5 * ((t - p) / w) ** 2)
            x[c] = env
        x[1] = np.round(x[1] * 40) / 40
        x[2, :40] += np.linspace(2.0, 0.0, 40)
        x[3, -25:] += np.linspace(0.0, 2.0, 25)
        x[4] = x[4].max() - x[4]
        xd = torch.from_numpy(x).cuda()
        top = xd.max().reshape(1)                                  # a device scalar
        for frac in (0.9, 0.45, 0.2, 0.08, 0.02):
            thr = frac * float(top.cpu()[0])
            got = dw.detect.pick_times(xd, thr)
            late = dw.detect.pick_times(xd, dw.detect.Threshold(frac, top), lazy=True)
            assert late._packed is None                            # nothing has been looked at yet
            for c in range(nx):
                ref = sps.find_peaks(x[c].astype(np.float64), prominence=thr)[0]
                assert np.array_equal(got[c], ref), (ns, frac, c, len(got[c]), len(ref))
            assert late.total == got.total and torch.equal(late.packed, got.packed) and torch.equal(late.counts, got.counts)


def test_picks_long_rows_window_sweep_vs_scipy(dw):
    """Rows beyond the LDS staging limit (detect.pick_times on raw 120 000-sample correlograms, detect.py:249-274): the
    windowed sweep over turning points (spectral.hip fp_sweep_segments) -- raw correlograms of noise (a maximum every ~9
    samples), their envelopes (a handful per window: the row falls back to the marking sweep), rows that change character
    half way, quantised rows (plateaus) -- every row against SciPy, with enough rows to have several workgroups per CU."""
    import scipy.signal as sps
    nx, ns = 640, 120000
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((nx, ns), device="cuda", generator=g)
    t = np.arange(ns) / FS
    hf = orc.gen_template_fincall(t, FS, 17.8, 28.8, 0.68)
    c = dw.detect.compute_cross_correlogram(x, hf)
    env = dw.dsp._analytic(c, 0)
    half = ns // 2
    c[1] = torch.round(c[1] * 8) / 8
    c[2, :half] = env[2, :half] * 0.2
    c[3, half:] = env[3, half:] * 0.2
    c[4] = env[4]
    c[5, 100:200] = 0.0
    cpu = c.cpu().numpy()
    rows = list(range(8)) + list(range(8, nx, 79))
    for frac in (0.45, 0.15, 0.9):
        thr = frac * float(c[6].max())
        got = dw.detect.pick_times(c, thr)
        for ch in rows:
            ref = sps.find_peaks(cpu[ch].astype(np.float64), prominence=thr)[0]
            assert np.array_equal(got[ch], ref), (frac, ch, len(got[ch]), len(ref))
    got = dw.detect.pick_times(c, 1e30)
    assert got.total == 0


def test_pick_pipeline_config1(dw):
    """f-k -> matched filter -> envelope picks on a 1000 x 12000 synthetic block: same picks as the
    float64 oracle pipeline except at prominences within 1e-4 of the threshold."""
    nx, ns = 1000, 12000
    x = orc.synth_block(nx, ns, fs=FS, step=4, seed=99, n_calls=5, n_waves=6) * 1e9
    x = orc.bp_filt(x, FS, 14, 30)
    t = np.arange(ns) / FS
    hf = orc.gen_template_fincall(t, FS, 17.8, 28.8, 0.68)
    ref_c = orc.compute_cross_correlogram(x, hf)
    thr = 0.45 * float(np.max(ref_c))
    ref_p = orc.pick_times_env(ref_c, thr)
    c = dw.detect.compute_cross_correlogram(x, hf)
    got = dw.detect.pick_times_env(c, thr)
    a, b = _sets(got), _sets(ref_p)
    print("picks: %d vs %d reference, symmetric difference %d" % (len(a), len(b), len(a ^ b)))
    assert len(b) > 50
    env = orc.envelope(ref_c)
    for ch in range(nx):                        # every differing pick must be marginal (prominence within 1e-4 thr of thr)
        assert_picks_match(got[ch], env[ch], thr, "pipeline picks, channel %d" % ch)


def test_detectors_on_long_rows(dw):
    """BASELINE configs[2] row length (120 000 samples) through the detector rows of SURVEY 8(a): matched
    filter -> envelope picks, spectrogram correlation, SNR; a few rows against the float64 oracle."""
    import scipy.signal as sps
    nx, ns = 600, 120000
    rng = np.random.default_rng(17)
    x = rng.standard_normal((nx, ns))
    t = np.arange(ns) / FS
    hf = orc.gen_template_fincall(t, FS, 17.8, 28.8, 0.68)
    for r in (0, 1, 599):                                   # a few calls so that picks exist
        for k in (5000, 60000, 110000):
            x[r, k:k + 136] += 6.0 * hf[:136]
    xt = torch.from_numpy(x).cuda().float()
    c = dw.detect.compute_cross_correlogram(xt, hf)
    rows = [0, 1, 599]
    ref_c = orc.compute_cross_correlogram(x[rows], hf)
    assert rel(c[rows].cpu().numpy(), ref_c) < TOL
    thr = 0.5 * float(ref_c.max())
    picks = dw.detect.pick_times_env(c, thr)
    assert len(picks) == nx
    for i, r in enumerate(rows):
        ndiff, nref = assert_picks_match(picks[r], orc.envelope(ref_c[i]), thr, "long-row picks, row %d" % r)
        assert nref >= 3
    sc = dw.detect.compute_cross_correlogram_spectrocorr(xt, FS, [14., 30.], KERNEL, 0.8, 0.95)
    assert tuple(sc.shape) == (nx, 1 + ns // 8)
    ref_sc = orc.compute_cross_correlogram_spectrocorr(x[rows], FS, [14., 30.], KERNEL, 0.8, 0.95)
    assert rel(sc[rows].cpu().numpy(), ref_sc) < TOL
    s = dw.dsp.snr_tr_array(xt[:8])
    ref_s = orc.snr_tr_array(x[:8])
    assert np.max(np.abs(10.0 ** (s.cpu().numpy() / 10) - 10.0 ** (ref_s / 10))) / np.max(10.0 ** (ref_s / 10)) < TOL


def test_envelope_of_the_bench_block(dw):
    """20 000 x 120 000: the long-row analytic signal on the shape-specialised f-k kernels (time phase + pass B with the Hilbert
    pair operation, every row its own Hermitian partner); rows are independent -- a few against scipy.signal.hilbert."""
    import scipy.signal as sps
    nx, ns = 20000, 120000
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((nx, ns), device="cuda", generator=gen)
    rows = [0, 1, 24, 25, 9999, 19999]
    env = dw.dsp.envelope(x)
    hil = dw.dsp.hilbert_imag(x[:, :])
    z = sps.hilbert(x[rows].cpu().numpy().astype(np.float64), axis=1)
    e1 = rel(env[rows].cpu().numpy(), np.abs(z))
    e2 = rel(hil[rows].cpu().numpy(), z.imag)
    print("envelope / Hilbert transform 20000x120000 (6 rows): rel err %.3e / %.3e" % (e1, e2))
    assert e1 < TOL and e2 < TOL
