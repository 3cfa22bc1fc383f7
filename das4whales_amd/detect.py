"""detect -- MI355X-native mirror of das4whales.detect (reference: src/das4whales/detect.py).

Templates are generated on the host in float64 (a few hundred samples); correlations run in the
HIP library (include/d4w.h: d4w_row_stats_f32, d4w_xcorr_f32)."""
import numpy as np
import torch

from . import _device as dev
from ._lib import lib, check


# ---------------------------------------------------------------------------------------------
# templates (host)
# ---------------------------------------------------------------------------------------------
def gen_linear_chirp(fmin, fmax, duration, sampling_rate):
    """Linear down-sweep fmax -> fmin -- reference detect.py:20-41
    (scipy.signal.chirp(t, f0=fmax, f1=fmin, t1=duration, 'linear') restated)."""
    t = np.arange(0, duration, 1 / sampling_rate)
    return np.cos(2 * np.pi * (fmax * t + 0.5 * (fmin - fmax) / duration * t * t))


def gen_hyperbolic_chirp(fmin, fmax, duration, sampling_rate):
    """Hyperbolic down-sweep fmax -> fmin -- reference detect.py:44-65."""
    t = np.arange(0, duration, 1 / sampling_rate)
    f0, f1 = fmax, fmin
    if f0 == f1:
        return np.cos(2 * np.pi * f0 * t)
    sing = -f1 * duration / (f0 - f1)
    return np.cos(2 * np.pi * (-sing * f0) * np.log(np.abs(1 - t / sing)))


def gen_template_fincall(time, fs, fmin=15., fmax=25., duration=1., window=True):
    """Hann-windowed hyperbolic chirp zero-padded to len(time) -- reference detect.py:68-93."""
    chirp_signal = gen_hyperbolic_chirp(fmin, fmax, duration, fs)
    template = np.zeros(np.shape(time))
    if window:
        template[:len(chirp_signal)] = chirp_signal * np.hanning(len(chirp_signal))
    else:
        template[:len(chirp_signal)] = chirp_signal
    return template


# ---------------------------------------------------------------------------------------------
# matched filter
# ---------------------------------------------------------------------------------------------
def _support(v):
    nz = np.nonzero(v)[0]
    return int(nz[-1]) + 1 if len(nz) else 1


def _taps_tensor(taps_list, device):
    lt = max(4, -(-max(len(t) for t in taps_list) // 4) * 4)
    taps = np.zeros((len(taps_list), lt), dtype=np.float32)
    for i, t in enumerate(taps_list):
        taps[i, :len(t)] = t
    return torch.from_numpy(taps).to(device), lt


def _xcorr_device(x, taps_list, normalize):
    """x: float32 CUDA [nx, ns]; taps_list: 1..n host float64 vectors -> list of CUDA tensors."""
    nx, ns = x.shape
    outs = []
    with torch.cuda.device(x.device):
        mean = mx = None
        if normalize:
            mean = torch.empty(nx, dtype=torch.float32, device=x.device)
            mx = torch.empty(nx, dtype=torch.float32, device=x.device)
            check(lib.d4w_row_stats_f32(dev.ptr(x), nx, ns, dev.ptr(mean), dev.ptr(mx), dev.stream_ptr(x)))
        for i in range(0, len(taps_list), 2):                      # two templates per read of x
            grp = taps_list[i:i + 2]
            taps, lt = _taps_tensor(grp, x.device)
            ys = [torch.empty_like(x) for _ in grp]
            check(lib.d4w_xcorr_f32(dev.ptr(x), nx, ns, dev.ptr(mean) if normalize else None,
                                    dev.ptr(mx) if normalize else None, dev.ptr(taps), len(grp), lt,
                                    dev.ptr(ys[0]), dev.ptr(ys[1]) if len(ys) > 1 else None,
                                    dev.stream_ptr(x)))
            outs.extend(ys)
    return outs


def _host_vec(v):
    if dev.is_tensor(v):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float64).ravel()


def shift_xcorr(x, y):
    """Positive-lag cross-correlation c[k] = sum_n x[n+k] y[n] -- reference detect.py:96-112."""
    yv = _host_vec(y)
    x1 = x if dev.is_tensor(x) else np.asarray(x)
    if x1.ndim != 1:
        raise ValueError("shift_xcorr expects 1-D inputs")
    xd = dev.to_device_f32(x1.reshape(1, -1))
    c = _xcorr_device(xd, [yv[:_support(yv)]], normalize=False)[0][0]
    return dev.like_input(c, x)


def shift_nxcorr(x, y):
    """shift_xcorr / (std(x) std(y) len(x)) -- reference detect.py:115-137 (population std)."""
    xv, yv = _host_vec(x), _host_vec(y)
    c = shift_xcorr(x, y)
    return c / (np.std(xv) * np.std(yv) * len(xv))


def _normalised_support(template):
    """detect.py:158: (template - mean) / max|template| over the zero-padded length; returns the
    non-zero support of the ORIGINAL template (the constant -mean/max tail is dropped, see
    include/d4w.h)."""
    t = _host_vec(template)
    a = np.max(np.abs(t))
    if a == 0:
        raise ValueError("template is all zeros")
    return ((t - t.mean()) / a)[:_support(t)]


def compute_cross_correlograms(data, templates):
    """Several templates against one block in a single pass over `data` (two templates per kernel
    launch) -- what scripts/main_mfdetect.py:79-80 does with two separate calls."""
    if getattr(data, "ndim", 0) != 2:
        raise ValueError("data must be a 2-D [channel x time] array")
    xd = dev.to_device_f32(data)
    outs = _xcorr_device(xd, [_normalised_support(t) for t in templates], normalize=True)
    return [dev.like_input(o, data) for o in outs]


def compute_cross_correlogram(data, template):
    """Peak-normalised matched filter, every row against `template` -- reference detect.py:140-166.

    Rows: (x - mean) / max|x| (max of the un-de-meaned row, detect.py:157).  Output is floating
    point (the reference's np.empty_like would truncate integer input); an all-zero row gives
    zeros where the reference divides by zero."""
    return compute_cross_correlograms(data, [template])[0]


# ---------------------------------------------------------------------------------------------
# pick utilities (index bookkeeping, host)
# ---------------------------------------------------------------------------------------------
def convert_pick_times(peaks_indexes_m):
    """Ragged per-channel index lists -> 2 x K array, row 0 = channel index, row 1 = time index
    -- reference detect.py:277-303."""
    ch = [np.full(len(p), i, dtype=np.int64) for i, p in enumerate(peaks_indexes_m)]
    if not ch:
        return np.zeros((2, 0), dtype=np.int64)
    return np.asarray((np.concatenate(ch), np.concatenate([np.asarray(p, dtype=np.int64) for p in peaks_indexes_m])))


def select_picked_times(idx_tp, tstart, tend, fs):
    """Keep picks with tstart*fs <= time index <= tend*fs -- reference detect.py:306-330."""
    keep = (idx_tp[1] >= tstart * fs) & (idx_tp[1] <= tend * fs)
    return (idx_tp[0][keep], idx_tp[1][keep])
