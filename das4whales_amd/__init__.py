"""das4whales_amd -- MI355X-native hot path of DAS4Whales (f-k filter, band-pass, matched filter).

Mirrors the reference namespace: `import das4whales_amd as dw; dw.dsp.fk_filter_filt(...)`,
`dw.detect.compute_cross_correlogram(...)` (reference: src/das4whales/__init__.py:1).
All array arithmetic runs in hand-written HIP kernels (das4whales_amd/csrc, libd4w.so) behind the
C ABI in include/d4w.h; PyTorch-ROCm is used only for device memory, streams and RCCL.
"""
from . import _lib  # noqa: F401  (fails loudly if the native library is missing)
from . import dsp  # noqa: F401
from . import detect  # noqa: F401
from . import data_handle  # noqa: F401
from . import stream  # noqa: F401
from . import improcess  # noqa: F401
from . import fkjit  # noqa: F401

fkjit.load_cached()     # shape-specialised f-k kernels compiled on demand earlier (lib/jit/*.so)



def set_strict_reference(on=True):
    """Reproduce the reference where this package's defaults deliberately differ: dsp.fk_filter_filt / fk_filter_sparsefilt with
    tapering=True taper the caller's array in place (reference dsp.py:744-745), detect.compute_cross_correlogram gives a NaN row
    for an all-zero channel (the reference's 0 / 0, detect.py:157).  Returns the previous setting."""
    old = detect.STRICT_REFERENCE
    detect.STRICT_REFERENCE = bool(on)
    return old


__all__ = ["dsp", "detect", "data_handle", "stream", "improcess", "set_strict_reference"]
