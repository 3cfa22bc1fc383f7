"""Pins oracle/d4w_oracle.py against the golden fixtures produced by the REAL reference code
(tests/golden/make_golden.py) and against the reference's own value-pinning tests
(tests/test_dsp.py:85-88, :136-141 in the reference).  CPU only."""
import numpy as np
import pytest

from oracle import d4w_oracle as orc


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


ARGS_SCRIPTS = dict(cs_min=1350., cp_min=1450., cp_max=3300, cs_max=3450, fmin=14., fmax=30.)


@pytest.fixture(scope="module")
def g(golden):
    return golden("fk_40x480.npz")


@pytest.fixture(scope="module")
def d(golden):
    return golden("detect_12x2000.npz")


def test_reference_pinned_vectors(golden):
    v = golden("ref_test_vectors.npz")
    assert np.allclose(orc.taper_data(v["taper_in"]), v["taper_expected"])
    assert np.allclose(v["taper_out"], v["taper_expected"])
    assert np.allclose(orc.snr_tr_array(v["snr_in"]), v["snr_expected"])
    assert np.allclose(v["snr_out"], v["snr_expected"])


def test_designs_bit_exact(g):
    shape, sel, dx, fs = g["x"].shape, list(g["sel"]), float(g["dx"]), float(g["fs"])
    assert np.array_equal(orc.fk_filter_design(shape, sel, dx, fs), g["m_classic"])
    assert np.array_equal(orc.hybrid_filter_design(shape, sel, dx, fs, 1350., 1450., 14., 30.), g["m_hybrid"])
    assert np.array_equal(orc.hybrid_ninf_filter_design(shape, sel, dx, fs, **ARGS_SCRIPTS), g["m_ninf"])
    assert rel(orc.hybrid_gs_filter_design(shape, sel, dx, fs, 1350., 1450., 14., 30.), g["m_gs"]) < 1e-14
    assert rel(orc.hybrid_ninf_gs_filter_design(shape, sel, dx, fs, **ARGS_SCRIPTS), g["m_ninf_gs"]) < 1e-14


def test_designs_second_shape(golden):
    h = golden("fk_30x360.npz")
    shape, sel, dx, fs = h["x"].shape, list(h["sel"]), float(h["dx"]), float(h["fs"])
    assert np.array_equal(orc.fk_filter_design(shape, sel, dx, fs), h["m_classic"])
    assert np.array_equal(orc.hybrid_ninf_filter_design(shape, sel, dx, fs, **ARGS_SCRIPTS), h["m_ninf"])
    assert rel(orc.fk_filter_filt(h["x"], h["m_ninf"]), h["y_ninf"]) < 1e-13
    assert rel(orc.fk_filter_filt_half(h["x"], h["m_ninf"]), h["y_ninf"]) < 1e-12
    assert rel(orc.fk_filter_filt_half(h["x"], h["m_classic"]), h["y_classic"]) < 1e-12


def test_fk_apply(g):
    x = g["x"]
    assert rel(orc.fk_filter_filt(x, g["m_classic"]), g["y_classic"]) < 1e-13
    assert rel(orc.fk_filter_filt(x, g["m_classic"], tapering=True), g["y_classic_taper"]) < 1e-13
    assert rel(orc.fk_filter_filt(x, g["m_ninf"]), g["y_ninf"]) < 1e-13
    assert rel(orc.fk_filter_filt(x, g["m_hybrid"]), g["y_hybrid"]) < 1e-13
    assert rel(orc.fk_filter_filt(x, g["m_ninf_gs"]), g["y_ninf_gs"]) < 1e-13
    assert rel(orc.fk_filt(x, 1, float(g["fs"]), 4, float(g["dx"]), 1400., 3400.), g["y_fkfilt"]) < 1e-13
    assert rel(orc.taper_data(x), g["taper"]) < 1e-15


def test_half_spectrum_identity(g):
    """Re(ifft2(F*M)) == irfft2(rfft2(x) * fold(M)) even for the non-Hermitian masks."""
    x = g["x"]
    for key_m, key_y in [("m_classic", "y_classic"), ("m_ninf", "y_ninf"), ("m_hybrid", "y_hybrid"),
                         ("m_ninf_gs", "y_ninf_gs")]:
        assert rel(orc.fk_filter_filt_half(x, g[key_m]), g[key_y]) < 1e-12


def test_iir(g):
    x, fs = g["x"], float(g["fs"])
    assert rel(orc.bp_filt(x, fs, 14, 30), g["y_bp"]) < 1e-12
    assert np.array_equal(orc.butterworth_filter([2, 5, "hp"], fs), g["sos_hp"])
    assert np.array_equal(orc.butterworth_filter([5, [10, 30], "bp"], fs), g["sos_bp"])
    assert rel(orc.sosfiltfilt(g["sos_hp"], x), g["y_sos_hp"]) < 1e-12
    assert rel(orc.sosfiltfilt(g["sos_bp"], x), g["y_sos_bp"]) < 1e-12
    with pytest.raises(ValueError, match="padlen, which is 51"):
        orc.bp_filt(np.zeros((2, 51)), fs, 14, 30)


def test_metrics(g):
    x, fs = g["x"], float(g["fs"])
    assert rel(orc.snr_tr_array(x), g["snr"]) < 1e-13
    assert rel(orc.snr_tr_array(x, env=True), g["snr_env"]) < 1e-12
    assert rel(orc.get_fx(x[:, :400], 512), g["fx"]) < 1e-13
    assert rel(orc.instant_freq(x[3], fs), g["ifreq"]) < 1e-9


def test_templates(d):
    fs = float(d["fs"])
    t = np.arange(d["x"].shape[1]) / fs
    assert rel(orc.gen_template_fincall(t, fs, 17.8, 28.8, 0.68), d["hf"]) < 1e-13
    assert rel(orc.gen_template_fincall(t, fs, 14.7, 21.8, 0.78), d["lf"]) < 1e-13
    assert rel(orc.gen_linear_chirp(15., 25., 1.0, fs), d["lin_chirp"]) < 1e-13
    assert rel(orc.gen_hyperbolic_chirp(15., 25., 1.0, fs), d["hyp_chirp"]) < 1e-13
    assert rel(orc.gen_template_fincall(t, fs, 15., 25., 1.0, window=False), d["tpl_nowin"]) < 1e-13


def test_matched_filter(d):
    x = d["x"]
    assert rel(orc.compute_cross_correlogram(x, d["hf"]), d["corr_hf"]) < 1e-12
    assert rel(orc.compute_cross_correlogram(x, d["lf"]), d["corr_lf"]) < 1e-12
    assert rel(orc.shift_xcorr(x[2], d["hf"]), d["xc"]) < 1e-12
    assert rel(orc.shift_nxcorr(x[2], d["hf"]), d["nxc"]) < 1e-12
    assert rel(orc.snr_tr_array(d["corr_hf"], env=True), d["snr_env_hf"]) < 1e-11


def test_picks(d):
    thr = float(d["thr"])
    pe = orc.convert_pick_times(orc.pick_times_env(d["corr_hf"], thr))
    pk = orc.convert_pick_times(orc.pick_times(d["corr_hf"], thr))
    assert np.array_equal(pe, d["picks_env"])
    assert np.array_equal(pk, d["picks"])
    assert pe.shape[1] > 0
    s0, s1 = orc.select_picked_times(pe, 1.0, 8.0, float(d["fs"]))
    assert np.array_equal(s0, d["picks_env_sel0"]) and np.array_equal(s1, d["picks_env_sel1"])
    # the explicit prominence restatement agrees with scipy's find_peaks row by row
    for row in d["corr_hf"][:4]:
        assert np.array_equal(orc.find_peaks_prominence(row, thr), orc.pick_times(row[None, :], thr)[0])
        env = orc.envelope(row)
        assert np.array_equal(orc.find_peaks_prominence(env, thr), orc.pick_times_env(row[None, :], thr)[0])


def test_spectro(d):
    x, fs = d["x"], float(d["fs"])
    p, tt, ff = orc.get_spectrogram(x[5], fs, nfft=256, overlap_pct=0.95)
    assert p.shape == d["spec_p"].shape
    fin = np.isfinite(d["spec_p"])
    assert np.max(np.abs(p[fin] - d["spec_p"][fin])) < 1e-9
    assert np.allclose(tt, d["spec_tt"]) and np.allclose(ff, d["spec_ff"])
    S, sff, stt = orc.get_sliced_nspectrogram(x[5], fs, 14., 30., 160, 8)
    assert rel(S, d["nspec"]) < 1e-13 and np.allclose(sff, d["nspec_ff"]) and np.allclose(stt, d["nspec_tt"])
    tvec, fvec, ker = orc.buildkernel(27., 17., 4., 0.8, sff, stt, fs, 14., 30.)
    assert rel(ker, d["ker"]) < 1e-13 and np.allclose(tvec, d["ker_tvec"])
    assert rel(orc.xcorr2d(S, ker), d["xcorr2d"]) < 1e-12
    sc = orc.compute_cross_correlogram_spectrocorr(x, fs, [14., 30.], {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}, 0.8, 0.95)
    assert rel(sc, d["spectrocorr"]) < 1e-12


def test_spectro_alternatives(d):
    """The reference's unused correlation variants (detect.py:495-576,605-647)."""
    fs = float(d["fs"])
    S, sff, stt = d["nspec"], d["nspec_ff"], d["nspec_tt"]
    ker, tvec = d["ker"], d["ker_tvec"]
    assert rel(orc.nxcorr2d(S, ker), d["nxcorr2d"]) < 1e-11
    ts, cv = orc.xcorr(stt, sff, S, tvec, sff, ker)
    assert np.allclose(ts, d["xcorr_t"]) and rel(cv, d["xcorr_v"]) < 1e-12
    assert rel(orc.buildkernel_from_template(17., 27., 0.8, fs, 160, 8), d["ker_tpl"]) < 1e-12
