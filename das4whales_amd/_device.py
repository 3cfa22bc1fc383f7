"""Tensor plumbing between NumPy / torch inputs and the C ABI (device pointers + stream).

Rules: numpy in -> numpy out with the input's floating dtype preserved (the
reference computes in float64; we compute in float32 and cast back); torch CUDA tensor in ->
torch CUDA tensor out with no PCIe round trip.
"""
import numpy as np
import torch


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("das4whales_amd needs a ROCm GPU (MI355X): torch.cuda.is_available() is "
                           "False and there is no CPU fallback")


def is_tensor(x):
    return isinstance(x, torch.Tensor)


# Host <-> device staging for the NumPy-in / NumPy-out calls (SURVEY.md 8d "H2D / D2H reported separately").  A notebook
# hands over pageable float64 arrays (reference data_handle.py:213).  Blocks above _STAGE_MIN bytes cross PCIe in row
# chunks through two pinned float32 buffers: a few host threads convert chunk k into one buffer (np.copyto releases the
# GIL) while chunk k - 1 is still on the wire out of the other, so the call costs about max(conversion, transfer) instead
# of their sum plus a pageable staging copy, and only float32 crosses the link.  Same in the other direction.
_STAGE_MIN = 32 << 20
_STAGE_BYTES = 64 << 20
_STAGE_THREADS = 8
_stage = {}            # (device index, thread id) -> [two pinned float32 buffers, two events]
_pool = None


def _staging(device):
    import threading
    key = (torch.device(device).index or 0, threading.get_ident())
    ent = _stage.get(key)
    if ent is None:
        if len(_stage) > 16:
            _stage.clear()
        bufs = [torch.empty(_STAGE_BYTES // 4, dtype=torch.float32).pin_memory() for _ in range(2)]
        ent = _stage[key] = [bufs, [torch.cuda.Event(), torch.cuda.Event()]]
    return ent


def _convert(dst, src):
    """dst[...] = src with dtype conversion, rows split over a small thread pool (NumPy copies release the GIL)."""
    global _pool
    n = dst.shape[0]
    if n < 2 * _STAGE_THREADS or dst.size < (1 << 20):
        np.copyto(dst, src, casting="unsafe")
        return
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=_STAGE_THREADS, thread_name_prefix="d4w-stage")
    step = -(-n // _STAGE_THREADS)
    futs = [_pool.submit(np.copyto, dst[r:r + step], src[r:r + step], "unsafe") for r in range(0, n, step)]
    for f in futs:
        f.result()


def upload_f32(a, device=None):
    """Host ndarray (any real dtype, any strides) -> contiguous float32 CUDA tensor, pipelined through pinned staging
    buffers on the current stream."""
    require_gpu()
    device = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
    a = np.asarray(a)
    if a.dtype.kind not in "fiub":
        a = a.astype(np.float64)
    if a.nbytes < _STAGE_MIN or a.ndim != 2 or a.shape[1] * 4 > _STAGE_BYTES:
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    nx, ns = a.shape
    out = torch.empty((nx, ns), dtype=torch.float32, device=device)
    bufs, evs = _staging(device)
    rows = max(1, (_STAGE_BYTES // 4) // ns)
    with torch.cuda.device(device):
        k = 0
        for r0 in range(0, nx, rows):
            r1 = min(nx, r0 + rows)
            if k >= 2:
                evs[k & 1].synchronize()         # the copy that last read this staging buffer has left the host
            st = bufs[k & 1][:(r1 - r0) * ns].view(r1 - r0, ns)
            _convert(st.numpy(), a[r0:r1])
            out[r0:r1].copy_(st, non_blocking=True)
            evs[k & 1].record()
            k += 1
        for e in evs[:min(k, 2)]:
            e.synchronize()                      # the staging buffers may be reused by the next call right away
    return out


def download(y, dtype=np.float32):
    """float32 CUDA tensor -> host ndarray of `dtype`, pipelined through the pinned staging buffers."""
    dtype = np.dtype(dtype)
    if (y.numel() * 4 < _STAGE_MIN or y.dim() != 2 or y.shape[1] * 4 > _STAGE_BYTES or not y.is_contiguous()
            or y.dtype != torch.float32):
        out = y.cpu().numpy()
        return out if out.dtype == dtype else out.astype(dtype)
    nx, ns = y.shape
    out = np.empty((nx, ns), dtype=dtype)
    bufs, evs = _staging(y.device)
    rows = max(1, (_STAGE_BYTES // 4) // ns)
    chunks = [(r0, min(nx, r0 + rows)) for r0 in range(0, nx, rows)]
    with torch.cuda.device(y.device):
        def issue(k):
            r0, r1 = chunks[k]
            bufs[k & 1][:(r1 - r0) * ns].view(r1 - r0, ns).copy_(y[r0:r1], non_blocking=True)
            evs[k & 1].record()
        issue(0)
        for k, (r0, r1) in enumerate(chunks):
            if k + 1 < len(chunks):
                issue(k + 1)                     # the next chunk crosses the link while this one is converted
            evs[k & 1].synchronize()
            _convert(out[r0:r1], bufs[k & 1][:(r1 - r0) * ns].view(r1 - r0, ns).numpy())
    return out


def to_device_f32(x, device=None):
    """Return a contiguous float32 CUDA tensor holding x (copying only when needed)."""
    require_gpu()
    if is_tensor(x):
        if not x.is_cuda:
            x = x.to(device or "cuda")
        elif device is not None and x.device != torch.device(device):
            x = x.to(device)              # a tensor on another GPU: the kernels run on `device`
        return x.to(torch.float32).contiguous()
    return upload_f32(x, device)


def like_input(y, template):
    """Convert the float32 CUDA result back to the caller's container / dtype."""
    if is_tensor(template):
        return y if template.is_cuda else y.to(template.device)
    t = np.asarray(template)
    return download(y, t.dtype if t.dtype.kind == "f" else np.float64)


def stream_ptr(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def ptr(t):
    return t.data_ptr()


def out_ptr(t):
    """Device pointer of a tensor a kernel is about to WRITE through.  The library writes through raw pointers, which
    torch does not see: bump the tensor's version counter as an in-place torch op would, so that whatever was remembered
    about its contents (detect._row_stats_cached keys on identity + version) is dropped -- `plan.apply(x_i, out=y)` in a
    per-file loop must not find file 0's row statistics on every later file."""
    torch.autograd.graph.increment_version(t)
    return t.data_ptr()
