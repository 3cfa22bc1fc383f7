"""GPU parity of the Gabor image pipeline against the
fixture generated from the reference's improcess code and against the oracle on larger blocks."""
import os

import numpy as np
import pytest
import torch

from oracle import d4w_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "image_240x1600.npz"))


def rel(y, ref):
    return float(np.max(np.abs(np.asarray(y, dtype=np.float64) - ref)) / max(np.max(np.abs(ref)), 1e-300))


@pytest.fixture(scope="module")
def dw():
    assert torch.cuda.is_available()
    import das4whales_amd as dw_
    return dw_


def test_golden_stages(dw):
    ip = dw.improcess
    trf = G["trf_fk"].astype(np.float64)
    image = ip.trace2image(trf)
    assert image.dtype == np.float64 and rel(image, G["image"]) < TOL
    assert rel(ip.scale_pixels(trf[:4]), G["scale_pixels"]) < TOL
    assert abs(ip.angle_fromspeed(1500., float(G["fs"]), float(G["dx"]), G["sel"]) - float(G["theta_c0"])) < 1e-12
    up, down = ip.gabor_filt_design(float(G["theta_c0"]))
    assert rel(up, G["gab_up"]) < 1e-12 and rel(down, G["gab_down"]) < 1e-12
    imagebin = ip.binning(G["image"].astype(np.float64), 1 / 10, 1 / 10)
    assert imagebin.shape == (24, 160) and rel(imagebin, G["imagebin"]) < TOL
    fimage = ip.filter2d(G["imagebin"], up) + ip.filter2d(G["imagebin"], down)
    assert rel(fimage, G["fimage"]) < TOL
    binary = G["fimage"] > float(G["threshold"])
    score = ip.filter2d(binary, up) + ip.filter2d(binary, down)
    assert rel(score, G["score"]) < TOL
    up_mask = ip.binning(G["mask"], 10, 10)
    assert up_mask.dtype == bool and np.array_equal(up_mask, G["mask_sparse"])
    assert np.array_equal(ip.apply_smooth_mask(trf, G["mask_sparse"]), trf.astype(np.float32).astype(np.float64) * G["mask_sparse"])
    assert rel(ip.apply_smooth_mask(G["imagebin"], G["mask"]), G["smoothed_image"]) < TOL


def _near(img, thr, tol):
    return np.abs(img - thr) <= tol * np.max(np.abs(img))


def test_golden_pipeline_one_call(dw):
    trf = G["trf_fk"].astype(np.float64)
    r = dw.improcess.gabor_mask(trf, float(G["fs"]), float(G["dx"]), G["sel"], float(G["c0"]), float(G["threshold"]),
                                float(G["threshold2"]))
    assert rel(r["image"], G["image"]) < TOL and rel(r["imagebin"], G["imagebin"]) < TOL
    assert rel(r["fimage"], G["fimage"]) < TOL
    # binary decisions: identical except pixels whose score is within the float32 budget of the threshold
    flips1 = (G["fimage"] > float(G["threshold"])) != (r["fimage"] > float(G["threshold"]))
    assert not np.any(flips1 & ~_near(G["fimage"], float(G["threshold"]), TOL))
    if not flips1.any():
        assert rel(r["score"], G["score"]) < TOL
        flips2 = r["mask"] != G["mask"]
        assert not np.any(flips2 & ~_near(G["score"], float(G["threshold2"]), TOL))
        if not flips2.any():
            assert np.array_equal(r["mask_sparse"], G["mask_sparse"])
            assert np.array_equal(r["masked_tr"], trf.astype(np.float32).astype(np.float64) * G["mask_sparse"])
    # CUDA tensors in -> CUDA tensors out
    rt = dw.improcess.gabor_mask(torch.from_numpy(G["trf_fk"]).cuda(), float(G["fs"]), float(G["dx"]), G["sel"],
                                 float(G["c0"]), float(G["threshold"]), float(G["threshold2"]))
    assert rt["masked_tr"].is_cuda and rt["mask"].dtype == torch.bool
    assert np.array_equal(rt["mask"].cpu().numpy(), r["mask"])


@pytest.mark.parametrize("h,w,kh,kw", [(400, 1200, 101, 101), (77, 333, 9, 31), (30, 20, 101, 101), (5, 5, 3, 3)])
def test_filter2d_random(dw, h, w, kh, kw):
    rng = np.random.default_rng(h + kw)
    img, ker = rng.standard_normal((h, w)), rng.standard_normal((kh, kw))
    assert rel(dw.improcess.filter2d(img, ker), orc.filter2d(img, ker)) < TOL


@pytest.mark.parametrize("h,w,ft,fx", [(4000, 1200, 0.1, 0.1), (1102, 1200, 0.1, 0.1), (123, 457, 0.37, 0.21), (40, 120, 10, 10),
                                       (31, 77, 2.5, 3.0)])
def test_binning_random(dw, h, w, ft, fx):
    rng = np.random.default_rng(h)
    img = rng.standard_normal((h, w))
    ref = orc.binning(img, ft, fx)
    got = dw.improcess.binning(img, ft, fx)
    assert got.shape == ref.shape and rel(got, ref) < TOL
    m = rng.random((h, w)) > 0.97
    assert np.array_equal(dw.improcess.binning(m, ft, fx), orc.binning(m, ft, fx))


def test_trace2image_block_and_long_rows(dw):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((300, 12000)) * rng.uniform(0.5, 2, (300, 1))
    assert rel(dw.improcess.trace2image(x), orc.trace2image(x)) < TOL
    xl = rng.standard_normal((6, 120000))                                  # long-row analytic path
    assert rel(dw.improcess.trace2image(xl), orc.trace2image(xl)) < TOL


def test_errors(dw):
    with pytest.raises(ValueError):
        dw.improcess.trace2image(np.zeros(10))
    with pytest.raises(ValueError):
        dw.improcess.apply_smooth_mask(np.zeros((4, 4)), np.zeros((4, 5), dtype=bool))
    with pytest.raises(ValueError):                                        # int(1101.99..) * 10 != 11020-like mismatch
        dw.improcess.gabor_mask(np.random.default_rng(0).standard_normal((25, 95)), 200., 2.04, [0, 100, 4])


def test_cv2_stand_ins_against_the_documented_definitions(dw):
    """The product's filter2d (HIP) and get_gabor_kernel against OpenCV's documented definitions written as loops
    (tests/known_answers.py) -- not against the restatement."""
    from tests import known_answers as ka
    ka.check_filter2d(dw.improcess.filter2d, 2e-6)
    ka.check_gabor_kernel(dw.improcess.get_gabor_kernel, 1e-12)
