"""data_handle -- the ingest step in front of the hot path (reference: src/das4whales/data_handle.py).

Only the array arithmetic is mirrored: channel selection + float conversion of the raw matrix and
raw2strain (de-mean, scale).  Reading the HDF5 / TDMS containers stays with the caller (h5py etc.
are I/O, out of scope here): hand the raw integer matrix over once it is in memory or on the GPU."""
import numpy as np
import torch

from . import _device as dev
from ._lib import lib, check

_DTYPES = {torch.int32: 0, torch.int16: 1, torch.float32: 2, torch.float64: 3}


def _to_device_raw(raw, device=None):
    dev.require_gpu()
    if dev.is_tensor(raw):
        t = raw if raw.is_cuda else raw.to(device or "cuda")
    else:
        a = np.asarray(raw)
        if a.dtype not in (np.int32, np.int16, np.float32, np.float64):
            a = a.astype(np.float64)
        t = torch.from_numpy(np.ascontiguousarray(a)).to(device or "cuda")
    if t.dtype not in _DTYPES:
        t = t.to(torch.float64)
    return t.contiguous()


def load_das_data_array(raw_data, selected_channels, metadata):
    """The array part of load_das_data (data_handle.py:181-230): rows
    selected_channels[0]:selected_channels[1]:selected_channels[2] of the raw [channel x time]
    matrix as strain (float32 CUDA tensor), plus the time and distance axes.  raw_data may be a NumPy
    array (int32 / int16 / float) or a torch tensor; the raw matrix crosses PCIe in its native width."""
    t = _to_device_raw(raw_data)
    nch, ns = t.shape
    c0, c1, step = int(selected_channels[0]), min(int(selected_channels[1]), nch), int(selected_channels[2])
    nx = len(range(c0, c1, step))
    y = torch.empty((nx, ns), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.d4w_raw2strain_f32(dev.ptr(t), _DTYPES[t.dtype], ns, c0, step, nx, float(metadata["scale_factor"]),
                                     dev.ptr(y), dev.stream_ptr(t)))
        # the kernel runs on torch's current stream, the stream a temporary `t` was allocated on: the caching allocator
        # re-uses its memory in stream order, no host synchronisation needed (a resident raw file keeps the pipeline asynchronous)
    tx = np.arange(ns) / metadata["fs"]                                          # data_handle.py:227
    dist = (np.arange(nx) * step + c0) * metadata["dx"]                          # data_handle.py:228
    return y, tx, dist


def raw2strain(trace, metadata):
    """trace - mean(trace, axis=1), times metadata['scale_factor'] -- data_handle.py:157-176.
    Returns a new array (the reference works in place on its float64 copy)."""
    if getattr(trace, "ndim", 0) != 2:
        raise ValueError("trace must be a 2-D [channel x time] array")
    t = _to_device_raw(trace)
    nx, ns = t.shape
    y = torch.empty((nx, ns), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.d4w_raw2strain_f32(dev.ptr(t), _DTYPES[t.dtype], ns, 0, 1, nx, float(metadata["scale_factor"]),
                                     dev.ptr(y), dev.stream_ptr(t)))
        torch.cuda.current_stream().synchronize()
    if dev.is_tensor(trace):
        return y if trace.is_cuda else y.cpu()
    a = np.asarray(trace)
    return y.cpu().numpy().astype(a.dtype if a.dtype.kind == "f" else np.float64)
