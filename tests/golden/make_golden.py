"""Generate the golden fixtures in tests/golden/*.npz from the REAL reference code.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports leabouffaut/DAS4Whales from /root/reference/src under oracle/ref_harness.py (stubs
for the absent I/O / plotting deps, a functional sparse.COO shim, and the librosa.stft
restatement) and records inputs + reference outputs (float64) for small seeded cases.  The
fixtures are what pins oracle/d4w_oracle.py (tests/test_oracle_golden.py) and what the GPU
parity tests compare against on the GPU box, where /root/reference does not exist.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.ref_harness import import_reference  # noqa: E402
from oracle import d4w_oracle as orc  # noqa: E402

FS = 200.0
DX = 2.0419046878814697


def dense(m):
    return np.asarray(m.todense() if hasattr(m, "todense") else m)


def main():
    dw = import_reference()
    import scipy.signal as sp
    rng = np.random.default_rng(20240807)

    # ------------------------------------------------------------------ f-k block: 40 x 480
    nx, ns = 40, 480
    sel = [0, nx * 4, 4]
    x = orc.synth_block(nx, ns, fs=FS, dx=DX, step=4, seed=7, n_calls=2, n_waves=8) * 1e9
    out = {"x": x, "sel": np.array(sel), "fs": FS, "dx": DX}
    args_classic = dict(cs_min=1400, cp_min=1450, cp_max=3400, cs_max=3500)
    args_scripts = dict(cs_min=1350., cp_min=1450., cp_max=3300, cs_max=3450, fmin=14., fmax=30.)
    m_classic = dw.dsp.fk_filter_design((nx, ns), sel, DX, FS, **args_classic)
    m_hybrid = dense(dw.dsp.hybrid_filter_design((nx, ns), sel, DX, FS, cs_min=1350., cp_min=1450., fmin=14., fmax=30.))
    m_ninf = dense(dw.dsp.hybrid_ninf_filter_design((nx, ns), sel, DX, FS, **args_scripts))
    m_gs = dense(dw.dsp.hybrid_gs_filter_design((nx, ns), sel, DX, FS, cs_min=1350., cp_min=1450., fmin=14., fmax=30.))
    m_ninf_gs = dense(dw.dsp.hybrid_ninf_gs_filter_design((nx, ns), sel, DX, FS, **args_scripts))
    out.update(m_classic=np.ascontiguousarray(m_classic), m_hybrid=m_hybrid, m_ninf=m_ninf,
               m_gs=m_gs, m_ninf_gs=m_ninf_gs)
    out["y_classic"] = dw.dsp.fk_filter_filt(x.copy(), m_classic)
    out["y_classic_taper"] = dw.dsp.fk_filter_filt(x.copy(), m_classic, tapering=True)
    coo = dw.dsp.hybrid_ninf_filter_design((nx, ns), sel, DX, FS, **args_scripts)
    out["y_ninf"] = dw.dsp.fk_filter_sparsefilt(x.copy(), coo)
    out["y_hybrid"] = dw.dsp.fk_filter_sparsefilt(x.copy(), dw.dsp.hybrid_filter_design(
        (nx, ns), sel, DX, FS, cs_min=1350., cp_min=1450., fmin=14., fmax=30.))
    out["y_ninf_gs"] = dw.dsp.fk_filter_sparsefilt(x.copy(), dw.dsp.hybrid_ninf_gs_filter_design(
        (nx, ns), sel, DX, FS, **args_scripts))
    out["y_fkfilt"] = dw.dsp.fk_filt(x.copy(), 1, FS, 4, DX, 1400., 3400.)
    out["taper"] = dw.dsp.taper_data(x.copy())
    out["y_bp"] = dw.dsp.bp_filt(x.copy(), FS, 14, 30)
    sos_hp = dw.dsp.butterworth_filter([2, 5, "hp"], FS)
    sos_bp = dw.dsp.butterworth_filter([5, [10, 30], "bp"], FS)
    out["sos_hp"] = sos_hp
    out["sos_bp"] = sos_bp
    out["y_sos_hp"] = sp.sosfiltfilt(sos_hp, x, axis=1)
    out["y_sos_bp"] = sp.sosfiltfilt(sos_bp, x, axis=1)
    out["snr"] = dw.dsp.snr_tr_array(x)
    out["snr_env"] = dw.dsp.snr_tr_array(x, env=True)
    out["fx"] = dw.dsp.get_fx(x[:, :400], 512)
    out["ifreq"] = dw.dsp.instant_freq(x[3], FS)
    np.savez_compressed(os.path.join(HERE, "fk_40x480.npz"), **out)

    # ------------------------------------------------------------------ odd-ish shape: 30 x 360 (nx not /4)
    nx2, ns2 = 30, 360
    sel2 = [10, 10 + nx2 * 2, 2]
    x2 = rng.standard_normal((nx2, ns2))
    m2 = dense(dw.dsp.hybrid_ninf_filter_design((nx2, ns2), sel2, DX, FS, **args_scripts))
    m2c = dw.dsp.fk_filter_design((nx2, ns2), sel2, DX, FS)
    np.savez_compressed(os.path.join(HERE, "fk_30x360.npz"), x=x2, sel=np.array(sel2), fs=FS, dx=DX,
                        m_ninf=m2, m_classic=np.ascontiguousarray(m2c),
                        y_ninf=dw.dsp.fk_filter_filt(x2.copy(), m2),
                        y_classic=dw.dsp.fk_filter_filt(x2.copy(), m2c))

    # ------------------------------------------------------------------ detection block: 12 x 2000
    nx3, ns3 = 12, 2000
    x3 = orc.synth_block(nx3, ns3, fs=FS, dx=DX, step=4, seed=11, n_calls=4, n_waves=4) * 1e9
    x3 = dw.dsp.bp_filt(x3, FS, 14, 30)
    time = np.arange(ns3) / FS
    hf = dw.detect.gen_template_fincall(time, FS, fmin=17.8, fmax=28.8, duration=0.68)
    lf = dw.detect.gen_template_fincall(time, FS, fmin=14.7, fmax=21.8, duration=0.78)
    det = {"x": x3, "fs": FS, "hf": hf, "lf": lf,
           "lin_chirp": dw.detect.gen_linear_chirp(15., 25., 1.0, FS),
           "hyp_chirp": dw.detect.gen_hyperbolic_chirp(15., 25., 1.0, FS),
           "tpl_nowin": dw.detect.gen_template_fincall(time, FS, 15., 25., 1.0, window=False)}
    det["corr_hf"] = dw.detect.compute_cross_correlogram(x3, hf)
    det["corr_lf"] = dw.detect.compute_cross_correlogram(x3, lf)
    det["xc"] = dw.detect.shift_xcorr(x3[2], hf)
    det["nxc"] = dw.detect.shift_nxcorr(x3[2], hf)
    det["snr_env_hf"] = dw.dsp.snr_tr_array(det["corr_hf"], env=True)
    thr = 0.5 * np.max(det["corr_hf"])
    det["thr"] = thr
    pk_env = dw.detect.pick_times_env(det["corr_hf"], thr)
    pk = dw.detect.pick_times(det["corr_hf"], thr)
    det["picks_env"] = dw.detect.convert_pick_times(pk_env)
    det["picks"] = dw.detect.convert_pick_times(pk)
    sel_t = dw.detect.select_picked_times(det["picks_env"], 1.0, 8.0, FS)
    det["picks_env_sel0"], det["picks_env_sel1"] = np.asarray(sel_t[0]), np.asarray(sel_t[1])
    p, tt, ff = dw.dsp.get_spectrogram(x3[5], FS, nfft=256, overlap_pct=0.95)
    det["spec_p"], det["spec_tt"], det["spec_ff"] = p, tt, ff
    sp_s, sff, stt = dw.detect.get_sliced_nspectrogram(x3[5], FS, 14., 30., 160, 8)
    det["nspec"], det["nspec_ff"], det["nspec_tt"] = sp_s, sff, stt
    tvec, fvec, ker = dw.detect.buildkernel(27., 17., 4., 0.8, sff, stt, FS, 14., 30.)
    det["ker_tvec"], det["ker"] = tvec, ker
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        det["spectrocorr"] = dw.detect.compute_cross_correlogram_spectrocorr(
            x3, FS, [14., 30.], {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}, 0.8, 0.95)
    det["xcorr2d"] = dw.detect.xcorr2d(sp_s, ker)
    det["nxcorr2d"] = dw.detect.nxcorr2d(sp_s, ker)
    xs, xv = dw.detect.xcorr(stt, sff, sp_s, tvec, fvec, ker)
    det["xcorr_t"], det["xcorr_v"] = np.asarray(xs), np.asarray(xv)
    det["ker_tpl"] = dw.detect.buildkernel_from_template(17., 27., 0.8, FS, 160, 8)
    np.savez_compressed(os.path.join(HERE, "detect_12x2000.npz"), **det)

    # ------------------------------------------------------------------ Gabor image pipeline: 240 x 1600
    # improcess functions are the reference's own code; cv2 / torchvision underneath are the shims of
    # oracle/ref_harness.py (cv2 restated, Resize = torch's own interpolate)
    import io
    import contextlib
    nx4, ns4 = 240, 1600
    sel4 = [0, nx4 * 4, 4]
    x4 = orc.synth_block(nx4, ns4, fs=FS, dx=DX, step=4, seed=23, n_calls=4, n_waves=6) * 1e9
    x4 = dw.dsp.bp_filt(x4, FS, 14., 30.)
    with contextlib.redirect_stdout(io.StringIO()):
        coo4 = dw.dsp.hybrid_ninf_filter_design((nx4, ns4), sel4, DX, FS, **args_scripts)
    trf = dw.dsp.fk_filter_sparsefilt(x4, coo4).astype(np.float32)          # stored as float32, fed as float64
    trf64 = trf.astype(np.float64)
    img = {"trf_fk": trf, "sel": np.array(sel4), "fs": FS, "dx": DX, "c0": 1500.}
    img["image"] = dw.improcess.trace2image(trf64)
    with contextlib.redirect_stdout(io.StringIO()):
        img["theta_c0"] = dw.improcess.angle_fromspeed(1500., FS, DX, sel4)
    img["imagebin"] = dw.improcess.binning(img["image"], 1 / 10, 1 / 10)
    up, down = dw.improcess.gabor_filt_design(img["theta_c0"])
    img["gab_up"], img["gab_down"] = up, down
    import cv2                                                                # the shim
    fimage = cv2.filter2D(img["imagebin"], cv2.CV_64F, up) + cv2.filter2D(img["imagebin"], cv2.CV_64F, down)
    img["fimage"] = fimage
    img["threshold"] = 0.5 * fimage.max()
    binary = fimage > img["threshold"]
    score = cv2.filter2D(binary.astype(float), cv2.CV_64F, up) + cv2.filter2D(binary.astype(float), cv2.CV_64F, down)
    img["score"] = score
    img["threshold2"] = 0.3 * score.max()
    mask = score > img["threshold2"]
    img["mask"] = mask
    img["smoothed_image"] = dw.improcess.apply_smooth_mask(img["imagebin"], mask)
    img["mask_sparse"] = dw.improcess.binning(mask, 10, 10)
    assert np.array_equal(dw.improcess.apply_smooth_mask(trf64, img["mask_sparse"]), trf64 * img["mask_sparse"])
    img["scale_pixels"] = dw.improcess.scale_pixels(trf64[:4])
    img["image"] = img["image"].astype(np.float32)                            # fixture size; 6e-8 relative
    np.savez_compressed(os.path.join(HERE, "image_240x1600.npz"), **img)

    # ------------------------------------------------------------------ reference's own pinned vectors
    # tests/test_dsp.py:85-88 (taper) and :136-141 (snr) -- literal values from the reference tests.
    lit = np.array([[1, 2, 3, 4, 5], [1, 2, 3, 4, 5]], dtype=float)
    np.savez_compressed(os.path.join(HERE, "ref_test_vectors.npz"),
                        taper_in=lit, taper_expected=np.array([[0, 2, 3, 4, 0], [0, 2, 3, 4, 0]], dtype=float),
                        taper_out=dw.dsp.taper_data(lit.copy()),
                        snr_in=lit,
                        snr_expected=np.array([[-3.01029996, 3.01029996, 6.53212514, 9.03089987, 10.96910013]] * 2),
                        snr_out=dw.dsp.snr_tr_array(lit.copy()))
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
