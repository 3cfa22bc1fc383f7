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


def to_device_f32(x, device=None):
    """Return a contiguous float32 CUDA tensor holding x (copying only when needed)."""
    require_gpu()
    if is_tensor(x):
        if not x.is_cuda:
            x = x.to(device or "cuda")
        elif device is not None and x.device != torch.device(device):
            x = x.to(device)              # a tensor on another GPU: the kernels run on `device`
        return x.to(torch.float32).contiguous()
    a = np.asarray(x)
    if a.dtype != np.float32:
        a = a.astype(np.float32)
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a).to(device or "cuda")


def like_input(y, template):
    """Convert the float32 CUDA result back to the caller's container / dtype."""
    if is_tensor(template):
        return y if template.is_cuda else y.to(template.device)
    t = np.asarray(template)
    out = y.cpu().numpy()
    if t.dtype.kind == "f" and t.dtype != np.float32:
        out = out.astype(t.dtype)
    elif t.dtype.kind != "f":
        out = out.astype(np.float64)
    return out


def stream_ptr(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def ptr(t):
    return t.data_ptr()
