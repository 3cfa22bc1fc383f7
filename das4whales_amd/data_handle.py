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


class PinnedIngest:
    """Raw files from host memory to strain on the GPU with the upload hidden behind the previous file's processing
    (reference contract: data_handle.load_das_data, data_handle.py:181-230, whose h5py read lands in
    host memory).  `depth` pinned host buffers of one file's selected rows [nx, ns] (raw dtype: int32 OptaSense RawData,
    int16, float32 or float64) and as many device buffers; a side HIP stream carries the copies:

        ing = PinnedIngest((nx, ns), np.int32)
        dset.read_direct(ing.host_array(0), np.s_[c0:c1:step, :])     # the caller's reader fills pinned memory in place
        ing.upload(0)
        for i in range(nfiles):
            if i + 1 < nfiles:                                          # file i + 1 is read and uploaded while file i is
                ing.wait_host((i + 1) % 2)                              # processed on the device
                dset_next.read_direct(ing.host_array((i + 1) % 2), ...)
                ing.upload((i + 1) % 2)
            x, tx, dist = ing.strain(i % 2, [0, nx, 1], metadata)       # current stream waits for the copy only
            ...

    A 11 020 x 12 000 int32 file is 529 MB: ~10 ms of PCIe against ~7 ms of device work per file, so a stream of files
    runs at the link rate when the reader keeps up (bench.py --config stream --from-host)."""

    def __init__(self, shape, dtype=np.int32, device=None, depth=2):
        dev.require_gpu()
        self.device = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        tdt = {np.dtype(np.int32): torch.int32, np.dtype(np.int16): torch.int16, np.dtype(np.float32): torch.float32,
               np.dtype(np.float64): torch.float64}.get(np.dtype(dtype))
        if tdt is None:
            raise ValueError("raw dtype must be int32, int16, float32 or float64")
        self.shape = (int(shape[0]), int(shape[1]))
        self._host = [torch.empty(self.shape, dtype=tdt).pin_memory() for _ in range(depth)]
        self._dev = [torch.empty(self.shape, dtype=tdt, device=self.device) for _ in range(depth)]
        self._copy = torch.cuda.Stream(self.device)
        # the device buffers come from the caching allocator on the CURRENT stream: a recycled block may still be read by kernels
        # queued there, and the first upload of a slot waits on nothing else -- order the copy stream behind them
        self._copy.wait_stream(torch.cuda.current_stream(self.device))
        self._ready = [torch.cuda.Event() for _ in range(depth)]        # the upload of slot s has completed
        self._free = [torch.cuda.Event() for _ in range(depth)]         # the device copy of slot s has been consumed
        self._uploaded = [False] * depth
        self._consumed = [False] * depth
        self.timing = None            # set to a list: every upload appends its (start, end) events on the copy stream

    def host_array(self, slot):
        """NumPy view of pinned buffer `slot` for the reader to fill (h5py read_direct, np.copyto, file.readinto)."""
        return self._host[slot].numpy()

    def wait_host(self, slot):
        """Block the host until the last upload out of `slot` has left host memory (before refilling it)."""
        if self._uploaded[slot]:
            self._ready[slot].synchronize()

    def upload(self, slot, src=None):
        """Start the copy of host buffer `slot` (or of `src`, a pinned host tensor of the same shape and dtype that a
        reader with its own buffer pool filled) to device buffer `slot` on the side stream; returns immediately."""
        if src is not None and (tuple(src.shape) != self.shape or src.dtype != self._dev[slot].dtype or not src.is_pinned()):
            raise ValueError("src must be a pinned host tensor of shape %s and dtype %s" % (self.shape, self._dev[slot].dtype))
        with torch.cuda.device(self.device):
            if self._consumed[slot]:
                self._copy.wait_event(self._free[slot])                 # the kernel that read the previous content is done
            with torch.cuda.stream(self._copy):
                if self.timing is not None:
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record(self._copy)
                self._dev[slot].copy_(self._host[slot] if src is None else src, non_blocking=True)
                if self.timing is not None:
                    t1.record(self._copy)
                    self.timing.append((t0, t1))
                self._ready[slot].record(self._copy)
        self._uploaded[slot] = True

    def raw(self, slot):
        """The device copy of `slot` (raw dtype), ordered after its upload on the current stream."""
        torch.cuda.current_stream(self.device).wait_event(self._ready[slot])
        return self._dev[slot]

    def strain(self, slot, selected_channels, metadata):
        """load_das_data_array of the uploaded slot: (strain float32 CUDA tensor, tx, dist)."""
        out = load_das_data_array(self.raw(slot), selected_channels, metadata)
        self._free[slot].record(torch.cuda.current_stream(self.device))
        self._consumed[slot] = True
        return out


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
