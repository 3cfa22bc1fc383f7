"""TEST INFRASTRUCTURE ONLY -- never imported by the das4whales_amd product path.

Imports the *real* reference package (leabouffaut/DAS4Whales, /root/reference/src) in this
build container so that golden vectors can be generated from the reference's own code
(tests/golden/make_golden.py).  /root/reference does not exist on the GPU box, so nothing
that runs there (bench.py, smoke(), -m gpu tests) may import this module.

The reference's `__init__` imports every sub-module (src/das4whales/__init__.py:1) and those
import I/O / plotting / image packages that are absent from this image (h5py, wget, dask,
nptdms: data_handle.py:11-18; librosa, sparse: dsp.py:12-13, detect.py:10; xarray: tools.py:1;
cv2, torchvision, skimage: improcess.py:12-18; pyproj: map.py:18).  None of them carries
arithmetic on the hot path except
  * `sparse.COO`  -- pure bookkeeping; replaced by the functional shim below,
  * `librosa.stft` -- absent; replaced by the restatement in oracle/d4w_oracle.py
    (`librosa_stft`, librosa >= 0.10 defaults).  Results that pass
    through it are labelled "restated-oracle" in DESIGN.md.
"""
import sys
import types
from unittest.mock import MagicMock

import numpy as np

REFERENCE_SRC = "/root/reference/src"


class COO:
    """Minimal functional stand-in for sparse.COO (dense-backed).

    Provides what the reference touches: COO.from_numpy (dsp.py:305,454,579,702), .shape,
    .data (tools.py:248), .todense() (dsp.py:784) and ndarray * COO -> COO (dsp.py:782).
    """
    __array_ufunc__ = None  # make ndarray.__mul__ defer to our __rmul__

    def __init__(self, dense):
        self._d = np.asarray(dense)

    @classmethod
    def from_numpy(cls, a):
        return cls(np.array(a))

    @property
    def shape(self):
        return self._d.shape

    @property
    def dtype(self):
        return self._d.dtype

    @property
    def data(self):
        return self._d[self._d != 0]

    @property
    def nnz(self):
        return int(np.count_nonzero(self._d))

    def todense(self):
        return self._d

    def __mul__(self, other):
        o = other._d if isinstance(other, COO) else other
        return COO(self._d * o)

    __rmul__ = __mul__


def _install_stubs():
    from oracle.d4w_oracle import librosa_stft

    for name in ["h5py", "wget", "dask", "dask.array", "nptdms", "xarray", "skimage", "skimage.transform",
                 "pyproj", "tqdm"]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock(name=name)
    _install_image_shims()
    # tqdm must be transparent: `for i in tqdm(range(n))`
    tq = types.ModuleType("tqdm")
    tq.tqdm = lambda it=None, *a, **k: it
    sys.modules["tqdm"] = tq

    sp = types.ModuleType("sparse")
    sp.COO = COO
    sys.modules["sparse"] = sp

    lr = types.ModuleType("librosa")
    lr.stft = librosa_stft
    sys.modules["librosa"] = lr

    import matplotlib
    matplotlib.use("Agg")


def _install_image_shims():
    """cv2 and torchvision.transforms as far as improcess.py's Gabor path touches them
    (improcess.py:116-123 getGaborKernel, :416-420 ToTensor / Resize; scripts/main_gabordetect.py:
    109,132 filter2D).  Both packages are absent here: cv2 is the oracle's restatement of OpenCV 4.9
    (parity unpinned), torchvision's Resize is the call torchvision 0.17.2 itself makes into torch,
    which IS installed (torch.nn.functional.interpolate, antialias=True)."""
    from oracle import d4w_oracle as orc

    cv2 = types.ModuleType("cv2")
    cv2.CV_64F = 6
    cv2.getGaborKernel = lambda ksize, sigma, theta, lambd, gamma, psi=np.pi * 0.5, ktype=6: \
        orc.get_gabor_kernel(ksize, sigma, theta, lambd, gamma, psi)
    cv2.filter2D = lambda src, ddepth, kernel: orc.filter2d(src, kernel)
    sys.modules["cv2"] = cv2

    import torch
    import torch.nn.functional as F

    class ToTensor:
        def __call__(self, pic):                       # torchvision functional.to_tensor for a 2-D ndarray
            t = torch.from_numpy(np.ascontiguousarray(pic[:, :, None].transpose(2, 0, 1)))
            return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t

    class Resize:
        def __init__(self, size):
            self.size = tuple(size)

        def __call__(self, img):                       # functional_tensor.resize, bilinear, antialias=True
            cast = img.dtype not in (torch.float32, torch.float64)
            x = img.to(torch.float32) if cast else img
            y = F.interpolate(x[None], size=self.size, mode="bilinear", align_corners=False, antialias=True)[0]
            if cast:
                if img.dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
                    y = torch.round(y)
                y = y.to(img.dtype)
            return y

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tr.ToTensor, tr.Resize = ToTensor, Resize
    tv.transforms = tr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr


def import_reference():
    """Return the reference package `das4whales` imported from /root/reference/src."""
    import os
    if not os.path.isdir(REFERENCE_SRC):
        raise RuntimeError("reference sources not present (expected only in the build container)")
    if "das4whales" in sys.modules and getattr(sys.modules["das4whales"], "__file__", "").startswith(REFERENCE_SRC):
        return sys.modules["das4whales"]
    _install_stubs()
    sys.path.insert(0, REFERENCE_SRC)
    try:
        import das4whales  # noqa
    finally:
        sys.path.remove(REFERENCE_SRC)
    return sys.modules["das4whales"]
