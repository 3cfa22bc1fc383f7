"""Oracle of the Gabor image pipeline against the fixture generated from the
reference's own improcess functions (tests/golden/make_golden.py) and against torch's CPU kernel for
the antialiased bilinear resize (the arithmetic torchvision 0.17 Resize delegates to)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import d4w_oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "image_240x1600.npz"))


def close(a, b, tol=1e-11):
    return np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) <= tol * max(np.max(np.abs(b)), 1e-300)


def test_trace2image_and_scale_pixels():
    trf = G["trf_fk"].astype(np.float64)
    assert close(orc.trace2image(trf), G["image"], 1e-6)            # fixture image is stored as float32
    assert close(orc.scale_pixels(trf[:4]), G["scale_pixels"])
    assert abs(orc.angle_fromspeed(1500., float(G["fs"]), float(G["dx"]), G["sel"]) - float(G["theta_c0"])) < 1e-12


def test_resize_restatement_matches_torch_cpu():
    rng = np.random.default_rng(3)
    for h, w, oh, ow in [(40, 120, 4, 12), (37, 101, 3, 10), (4, 12, 40, 120), (50, 33, 17, 80), (24, 160, 240, 1600)]:
        img = rng.standard_normal((h, w))
        ref = F.interpolate(torch.from_numpy(img)[None, None], size=(oh, ow), mode="bilinear", align_corners=False,
                            antialias=True)[0, 0].numpy()
        assert np.max(np.abs(orc.resize_bilinear_aa(img, oh, ow) - ref)) < 1e-13


def test_binning_down_and_bool_up():
    image = orc.trace2image(G["trf_fk"].astype(np.float64))
    assert close(orc.binning(image, 1 / 10, 1 / 10), G["imagebin"], 1e-9)
    up = orc.binning(G["mask"], 10, 10)
    assert up.dtype == bool and np.array_equal(up, G["mask_sparse"])


def test_gabor_kernels_and_pipeline():
    up, down = orc.gabor_filt_design(float(G["theta_c0"]))
    assert up.shape == (101, 101) and close(up, G["gab_up"]) and close(down, G["gab_down"])
    r = orc.gabor_mask_pipeline(G["trf_fk"].astype(np.float64), float(G["fs"]), float(G["dx"]), G["sel"], float(G["c0"]),
                                float(G["threshold"]), float(G["threshold2"]))
    assert close(r["fimage"], G["fimage"], 1e-9) and close(r["score"], G["score"], 1e-9)
    assert np.array_equal(r["mask"], G["mask"]) and np.array_equal(r["mask_sparse"], G["mask_sparse"])
    assert close(orc.apply_smooth_mask(r["imagebin"], r["mask"]), G["smoothed_image"], 1e-9)
    assert np.array_equal(r["masked_tr"], G["trf_fk"].astype(np.float64) * G["mask_sparse"])


def test_filter2d_reflect101_known_answer():
    """Identity kernel, shift kernel and a border case worked by hand (reflect-101: d c b | a b c d | c b a)."""
    img = np.arange(20, dtype=float).reshape(4, 5)
    k = np.zeros((3, 3))
    k[1, 1] = 1
    assert np.array_equal(orc.filter2d(img, k), img)
    k = np.zeros((3, 3))
    k[1, 0] = 1                                                       # out[y][x] = img[y][x-1]; x = 0 reads img[y][1]
    out = orc.filter2d(img, k)
    assert np.array_equal(out[:, 1:], img[:, :-1]) and np.array_equal(out[:, 0], img[:, 1])


def test_cv2_stand_ins_against_the_documented_definitions():
    """f3's cv2 pieces pinned independently of the restatement: OpenCV's documented definitions of
    filter2D / BORDER_REFLECT_101 / getGaborKernel written out as loops (tests/known_answers.py) -- impulse -> flipped kernel,
    constant -> sum of the kernel, a hand-worked border, kernels larger than the image, closed-form Gabor samples and the
    mirrored write."""
    from tests import known_answers as ka
    ka.check_filter2d(orc.filter2d, 1e-12)
    ka.check_gabor_kernel(orc.get_gabor_kernel)
