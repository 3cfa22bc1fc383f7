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

    for name in ["h5py", "wget", "dask", "dask.array", "nptdms", "xarray", "cv2",
                 "torchvision", "torchvision.transforms", "skimage", "skimage.transform",
                 "pyproj", "tqdm"]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock(name=name)
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
