"""detect -- MI355X-native mirror of das4whales.detect (reference: src/das4whales/detect.py).

Templates are generated on the host in float64 (a few hundred samples); correlations run in the
HIP library (include/d4w.h: d4w_row_stats_f32, d4w_xcorr_f32)."""
import numpy as np
import torch

import collections.abc

from . import _device as dev
from ._lib import lib, check
from .dsp import _cache_lock


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


_xf_tables = {}     # (templates, device, stream) -> [taps tensor, lt, workspace, tables built]


def _xf_prepared(grp, device, ws=True):
    """Device taps + the overlap-save workspace of a template group, kept per (templates, device, stream): the second
    call with the same templates on the same stream finds the template spectra already in the workspace (taps = NULL
    in d4w_xcorr_fft_cont_f32) and uploads nothing.  ws=False (the matrix-core form): the taps only."""
    key = (tuple(np.asarray(t, dtype=np.float64).tobytes() for t in grp), str(device),
           int(torch.cuda.current_stream(device).cuda_stream))
    with _cache_lock:
        ent = _xf_tables.get(key)
        if ent is None:
            if len(_xf_tables) > 32:
                _xf_tables.clear()
            taps, lt = _taps_tensor(grp, device)
            ent = _xf_tables[key] = [taps, lt, None, False]
        if ws and ent[2] is None:
            ent[2] = torch.empty(int(lib.d4w_xcorr_fft_ws_bytes()), dtype=torch.uint8, device=device)
    return ent


# das4whales_amd.set_strict_reference(True): the reference's side effects and degenerate values where this package's defaults
# differ: tapering=True tapers the caller's array (dsp.py:744-745), an all-zero row correlates to NaN (detect.py:157)
STRICT_REFERENCE = False

_row_stats_memo = {}        # (data_ptr, shape, device) -> (weakref to the tensor, its version, (mean, maxabs))


def _row_stats_cached(x, prefix=False):
    """(mean, max|.|) of the rows of a CUDA tensor, remembered while the SAME tensor (identity and version counter) is asked
    again: the reference's scripts call compute_cross_correlogram once per template on one block
    (scripts/main_mfdetect.py:79-80), which would read it twice just for the normalisation.  prefix=True: (mean, max|.|,
    prefix maxima) from one launch (d4w_row_stats_prefix_f32; what _apply_tails decides on)."""
    import weakref
    nx, ns = x.shape
    # per stream: statistics formed on one stream are not ordered before a kernel of another
    key = (x.data_ptr(), nx, ns, str(x.device), int(torch.cuda.current_stream(x.device).cuda_stream), bool(prefix))
    with _cache_lock:
        ent = _row_stats_memo.get(key)
        if ent is not None and ent[0]() is x and ent[1] == x._version:
            return ent[2]
    with torch.cuda.device(x.device):
        mean = torch.empty(nx, dtype=torch.float64, device=x.device)      # float64 (hi + lo for the kernels): d4w.h
        mx = torch.empty(nx, dtype=torch.float32, device=x.device)
        if prefix:
            pm = torch.empty(nx, dtype=torch.float32, device=x.device)
            check(lib.d4w_row_stats_prefix_f32(dev.ptr(x), nx, ns, dev.ptr(mean), dev.ptr(mx), dev.ptr(pm), dev.stream_ptr(x)))
            res = (mean, mx, pm)
        else:
            check(lib.d4w_row_stats_f32(dev.ptr(x), nx, ns, dev.ptr(mean), dev.ptr(mx), dev.stream_ptr(x)))
            res = (mean, mx)
    with _cache_lock:
        if len(_row_stats_memo) > 8:
            _row_stats_memo.clear()
        _row_stats_memo[key] = (weakref.ref(x), x._version, res)
    return res


def _remember_row_stats(x, stats):
    """Deposits (float64 row means, float32 row maxima) somebody else formed for the CUDA tensor x as it is now -- the f-k
    filter's last pass leaves them in its epilogue (dsp._fk_apply) -- so that the matched filter that follows does not read
    the block again for its normalisation (detect.py:157).  Dropped like any memo entry when x is written to."""
    import weakref
    nx, ns = x.shape
    key = (x.data_ptr(), nx, ns, str(x.device), int(torch.cuda.current_stream(x.device).cuda_stream), False)
    with _cache_lock:
        if len(_row_stats_memo) > 8:
            _row_stats_memo.clear()
        _row_stats_memo[key] = (weakref.ref(x), x._version, tuple(stats[:2]))


def _xcorr_method(taps_list, ns, method):
    """The kernel a correlation runs on: "mm" (banded-Toeplitz product on the matrix cores: two templates of <= 177 samples in
    one launch, one template of <= 497 per launch, longer ones in 496-tap sections up to d4w_xcorr_mm_max_support(); the
    default), "fft" (overlap-save, supports <= 161, rows >= 1024 samples) or "direct" (any support).  D4W_XCORR_METHOD
    overrides "auto" (measurements, A/B tests)."""
    import os
    import warnings
    explicit = method != "auto"
    if method == "auto":
        method = os.environ.get("D4W_XCORR_METHOD", "auto")
    if method not in ("auto", "mm", "fft", "direct"):
        raise ValueError("method must be 'auto', 'mm', 'fft' or 'direct', not %r" % (method,))
    longest = max(len(t) for t in taps_list)
    mm_ok = longest <= int(lib.d4w_xcorr_mm_max_support())
    fft_ok = ns >= 1024 and longest <= int(lib.d4w_xcorr_fft_max_support())
    if method == "mm" and not mm_ok:
        if explicit:
            raise ValueError("the matrix-core correlation takes supports <= %d samples" % int(lib.d4w_xcorr_mm_max_support()))
        warnings.warn("D4W_XCORR_METHOD=mm does not apply to a support of %d samples: choosing the form as 'auto' does" % longest)
        method = "auto"
    if method == "fft" and not fft_ok:
        # an override the shape does not admit falls back like 'auto' instead of surfacing as EINVAL from the C side
        warnings.warn("the overlap-save FFT correlation needs rows >= 1024 samples and supports <= %d (got %d, %d): "
                      "choosing the form as 'auto' does" % (int(lib.d4w_xcorr_fft_max_support()), ns, longest))
        method = "auto"
    if method == "mm" or (method == "auto" and mm_ok):
        return "mm"
    if method == "fft" or (method == "auto" and fft_ok):
        return "fft"
    return "direct"


def _tails_in_kernel(taps_list, coefs, ns, how):
    """Whether the zero-padded templates' constant tails (detect.py:158) can be added inside the matrix-core correlator
    (d4w_xcorr_mm_tail_f32: exact on every row, no second pass): the matrix-core form, supports of at most
    d4w_xcorr_mm_tail_max_support() samples (one launch per template) and shorter than the rows.
    D4W_XCORR_TAIL=pass keeps the two-pass form of rounds 1-5 (A/B measurements)."""
    import os
    if how != "mm" or not any(c != 0.0 for c in coefs) or os.environ.get("D4W_XCORR_TAIL", "kernel") == "pass":
        return False
    longest = max(len(t) for t in taps_list)
    return longest <= int(lib.d4w_xcorr_mm_tail_max_support()) and longest < ns


def _xcorr_device(x, taps_list, normalize, method="auto", stats=None, cont=None, row_max=None, tails=None):
    """x: float32 CUDA [nx, ns]; taps_list: 1..n host float64 vectors -> list of CUDA tensors.
    tails: optional list of the templates' DC-tail coefficients (_tail_coef) to be added inside the matrix-core kernel
    (the caller has checked _tails_in_kernel; needs normalize=True).
    method: "mm" (matrix cores), "fft" (overlap-save), "direct", or "auto" (_xcorr_method).
    stats: optional (mean, maxabs) CUDA tensors of the rows, e.g. from FkPlan.apply_stats.
    cont: optional (tensor [nx, >= n], n): the record continues -- the last lags read the first n samples of these rows
    instead of zeros (stream.FileStream); the matrix-core form, or exactly two templates on the FFT form.
    row_max: optional list that receives one CUDA tensor [nx] per template, max over the lags of every row, from the
    matrix-core kernel's epilogue (d4w_xcorr_mm_rowmax_f32); other forms leave it empty."""
    nx, ns = x.shape
    outs = []
    how = _xcorr_method(taps_list, ns, method)
    if tails is not None:
        if how != "mm" or not normalize or len(tails) != len(taps_list):
            raise ValueError("in-kernel template tails need the matrix-core form, normalize=True and one coefficient per template")
    if cont is not None:
        nxt, n_next = cont
        if not (how in ("mm", "fft") and (how == "mm" or len(taps_list) == 2) and nxt.is_cuda and nxt.dtype == torch.float32
                and nxt.stride(1) == 1 and nxt.shape[0] == nx and nxt.shape[1] >= n_next):
            raise ValueError("a continuation needs the matrix-core form (or two templates on the FFT form) and float32 CUDA rows")
    with torch.cuda.device(x.device):
        mean = mx = None
        if normalize and stats is not None:
            mean, mx = stats
            if (mean.dtype != torch.float64 or mx.dtype != torch.float32 or mean.numel() < nx or mx.numel() < nx
                    or not mean.is_contiguous() or not mx.is_contiguous()):
                raise ValueError("stats = (float64 row means, float32 row maxima), one contiguous value per row (d4w_row_stats_f32)")
        elif normalize:
            mean, mx = _row_stats_cached(x)
        for i in range(0, len(taps_list), 2):                      # two templates per read of x
            grp = taps_list[i:i + 2]
            # the correlograms of a fused pair sit back to back in one allocation: a consumer that treats both alike (envelope
            # picks at one threshold) can take them as ONE [2 nx, ns] block (stacked_pair below) -- one launch per operator
            pair = torch.empty((len(grp),) + tuple(x.shape), dtype=x.dtype, device=x.device)
            ys = [pair[k] for k in range(len(grp))]
            if how == "mm":
                taps, lt, _, _ = _xf_prepared(grp, x.device, ws=False)
                rm = [torch.empty(nx, dtype=torch.float32, device=x.device) for _ in grp] if row_max is not None else None
                tc = [float(c) for c in tails[i:i + 2]] if tails is not None else [0.0]
                check(lib.d4w_xcorr_mm_tail_f32(dev.ptr(x), nx, ns,
                                                dev.ptr(cont[0]) if cont is not None else None,
                                                int(cont[0].stride(0)) if cont is not None else 0,
                                                int(cont[1]) if cont is not None else 0,
                                                dev.ptr(mean) if normalize else None, dev.ptr(mx) if normalize else None,
                                                dev.ptr(taps), len(grp), lt, len(grp[0]), len(grp[-1]), tc[0], tc[-1] if len(grp) > 1 else 0.0,
                                                dev.ptr(ys[0]), dev.ptr(ys[1]) if len(ys) > 1 else None,
                                                dev.ptr(rm[0]) if rm else None, dev.ptr(rm[1]) if rm and len(rm) > 1 else None,
                                                dev.stream_ptr(x)))
                outs.extend(ys)
                if rm:
                    row_max.extend(rm)
                continue
            if how == "fft":
                ent = _xf_prepared(grp, x.device)
                taps, lt, ws, built = ent
                check(lib.d4w_xcorr_fft_cont_f32(dev.ptr(x), nx, ns,
                                                 dev.ptr(cont[0]) if cont is not None else None,
                                                 int(cont[0].stride(0)) if cont is not None else 0,
                                                 int(cont[1]) if cont is not None else 0,
                                                 dev.ptr(mean) if normalize else None,
                                                 dev.ptr(mx) if normalize else None, None if built else dev.ptr(taps), len(grp), lt,
                                                 len(grp[0]), len(grp[-1]),
                                                 dev.ptr(ys[0]), dev.ptr(ys[1]) if len(ys) > 1 else None,
                                                 dev.ptr(ws), dev.stream_ptr(x)))
                ent[3] = True
                outs.extend(ys)
                continue
            taps, lt = _taps_tensor(grp, x.device)
            check(lib.d4w_xcorr_lens_f32(dev.ptr(x), nx, ns, dev.ptr(mean) if normalize else None,
                                         dev.ptr(mx) if normalize else None, dev.ptr(taps), len(grp), lt,
                                         len(grp[0]), len(grp[-1]),
                                         dev.ptr(ys[0]), dev.ptr(ys[1]) if len(ys) > 1 else None,
                                         dev.stream_ptr(x)))
            outs.extend(ys)
    return outs


def stacked_pair(c0, c1):
    """The two correlograms of one fused correlator launch as ONE [2 nx, ns] tensor (no copy) when they sit back to back in one
    allocation (what _xcorr_device produces for a template pair), else None.  Rows 0 .. nx-1 are c0's, nx .. 2 nx-1 are c1's."""
    b0, b1 = getattr(c0, "_base", None), getattr(c1, "_base", None)
    if (b0 is None or b0 is not b1 or b0.dim() != 3 or b0.shape[0] != 2 or not b0.is_contiguous() or c0.shape != c1.shape
            or c0.data_ptr() != b0.data_ptr() or c1.data_ptr() != b0.data_ptr() + c0.numel() * c0.element_size()):
        return None
    return b0.view(2 * c0.shape[0], c0.shape[1])


def correlogram_max(c, row_max=None, on_device=False):
    """np.max(corr_m) of a correlogram as a Python float (scripts/main_mfdetect.py:82: the detection threshold is half the
    largest correlation): from the per-row maxima the matrix-core correlator left in its epilogue when they are given
    (stream.FileStream results carry them: one 2-value reduction over nx numbers), else one read of the block -- the
    library's own reduction either way, one 8-byte copy to the host.  on_device=True: the value stays on the device (a
    1-element float32 CUDA tensor, no copy, no wait) for `Threshold(0.45, value)` below."""
    src = row_max if row_max is not None else dev.to_device_f32(c)
    mm = torch.empty(2, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        check(lib.d4w_minmax_f32(dev.ptr(src), int(src.numel()), dev.ptr(mm), dev.stream_ptr(src)))
    return mm[1:2] if on_device else float(mm.cpu()[1])


class Threshold:
    """A pick threshold formed on the device: scale x value[0] (value: a 1-element float32 CUDA tensor, e.g.
    correlogram_max(c, on_device=True)) -- the reference's `0.45 * np.max(corr_m)` (scripts/main_mfdetect.py:82,95) without the
    host waiting for the maximum; the product is formed in float64 like the host's.  Accepted wherever pick_times /
    pick_times_env take a threshold."""

    def __init__(self, scale, value):
        if not (dev.is_tensor(value) and value.is_cuda and value.dtype == torch.float32 and value.numel() == 1):
            raise ValueError("value must be a 1-element float32 CUDA tensor")
        self.scale, self.value = float(scale), value

    def __float__(self):
        return self.scale * float(self.value.cpu().reshape(-1)[0])


def xcorr_continuation_ok(taps_list, ns):
    """Whether _xcorr_device(..., cont=...) applies: the matrix-core form, or two templates that run the fused overlap-save kernel."""
    import os
    how = _xcorr_method(taps_list, ns, "auto")
    if how == "mm":
        return True
    return (how == "fft" and len(taps_list) == 2
            and os.environ.get("D4W_XF_FUSED", "1") == "1" and os.environ.get("D4W_XF_TPAIR", "0") == "0")


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


_tpl_memo = {}      # digest of a template's bytes -> (normalised support, tail coefficient)


def _template_parts(template):
    """(normalised support, tail coefficient) of a template, remembered by the CONTENT of the vector (a 64-bit digest of its
    bytes: 0.1 ms for a 120 000-sample template against 0.8 ms for the mean / max / support sweeps -- the reference's scripts call
    compute_cross_correlogram once per template and block with the same two vectors, scripts/main_mfdetect.py:79-80)."""
    t = _host_vec(template)
    try:
        import xxhash
        key = (t.shape[0], xxhash.xxh3_64_intdigest(t.tobytes()))
    except Exception:                                    # noqa: BLE001  (no xxhash: a slower digest, same behaviour)
        import hashlib
        key = (t.shape[0], hashlib.blake2b(t.tobytes(), digest_size=8).digest())
    with _cache_lock:
        hit = _tpl_memo.get(key)
    if hit is not None:
        return hit
    a = np.max(np.abs(t))
    if a == 0:
        raise ValueError("template is all zeros")
    sup = _support(t)
    taps = ((t - t.mean()) / a)[:sup]
    taps.setflags(write=False)
    coef = 0.0 if sup >= len(t) else float(t.mean() / a)
    with _cache_lock:
        if len(_tpl_memo) > 64:
            _tpl_memo.clear()
        _tpl_memo[key] = (taps, coef)
    return taps, coef


def _normalised_support(template):
    """detect.py:158: (template - mean) / max|template| over the zero-padded length; returns the
    non-zero support of the ORIGINAL template (the constant -mean/max tail on the padded part is
    handled by _tail_coef / d4w_xcorr_mm_tail_f32 / d4w_xcorr_dc_tail_f32)."""
    return _template_parts(template)[0]


def _tail_coef(template):
    """mean(template) / max|template| over the zero-padded length: minus the constant that detect.py:158
    leaves on the padded part (0 when the template fills its whole length)."""
    return _template_parts(template)[1]


# The DC tail of a zero-padded template (detect.py:158) is |coef| g times a PREFIX SUM of the de-meaned row: 3-5e-6 of the
# correlogram's maximum for the fin-whale templates on white 60-s rows, 3e-8 on band-passed rows, 1e-3 on rows that drift --
# a property of the data, so the decision is taken per row on the data (round 5; rounds 1-4 predicted it from the template
# assuming white rows and let drifting rows pass with 2-9e-4): a row is left without the term only when the term cannot
# exceed TAIL_EPS of that row's own largest correlation (d4w_row_prefix_max_f32 bounds it, the correlator's epilogue gives
# the row maximum), every other row receives it.
TAIL_EPS = 1e-6


def _prefix_max(x, mean):
    """max_j |sum_{i<j} (x - mean)| per row (d4w_row_prefix_max_f32): one read of x, shared by all templates."""
    nx, ns = x.shape
    with torch.cuda.device(x.device):
        pm = torch.empty(nx, dtype=torch.float32, device=x.device)
        check(lib.d4w_row_prefix_max_f32(dev.ptr(x), nx, ns, dev.ptr(mean), dev.ptr(pm), dev.stream_ptr(x)))
    return pm


def _apply_tails(x, stats, outs, taps, coefs, row_max, exact_tail=None, pmax=None):
    """Adds the DC-tail term of every template with a non-zero coefficient to its correlogram in place.  exact_tail None:
    per-row decision (TAIL_EPS) where the correlator left its row maxima, every row otherwise; True: every row; False:
    none.  Row maxima of the rows that changed are formed again by the kernel.  pmax: the rows' prefix maxima when the
    caller has them already (dsp.FkPlan.apply_stats_prefix)."""
    if exact_tail is False or not any(c != 0.0 for c in coefs):
        return
    nx, ns = x.shape
    mean, mx = stats
    by_row = exact_tail is None and row_max is not None and len(row_max) == len(outs)
    pm = (pmax if pmax is not None else _prefix_max(x, mean)) if by_row else None
    with torch.cuda.device(x.device):
        for k, (o, tp, c) in enumerate(zip(outs, taps, coefs)):
            if c == 0.0:
                continue
            rm = row_max[k] if row_max is not None and len(row_max) == len(outs) else None
            check(lib.d4w_xcorr_dc_tail_rows_f32(dev.ptr(x), nx, ns, dev.ptr(mean), dev.ptr(mx), float(c), len(tp),
                                                 dev.out_ptr(o), dev.ptr(pm) if by_row else None,
                                                 dev.out_ptr(rm) if rm is not None else None,
                                                 TAIL_EPS if by_row else 0.0, dev.stream_ptr(x)))


def _nan_dead_rows(xd, outs):
    """zero_rows="nan": an all-zero row (a dead channel) is 0 / 0 in the reference's normalisation (detect.py:157) and its
    correlogram NaN; the kernels give zeros.  One masked fill per correlogram, only when asked for."""
    dead = _row_stats_cached(xd)[1] == 0
    for o in outs:
        o[dead] = float("nan")


def compute_cross_correlograms(data, templates, exact_tail=None, zero_rows=None):
    """Several templates against one block (detect.compute_cross_correlogram for each) -- what
    scripts/main_mfdetect.py:79-80 does with two separate calls.  exact_tail: True / False forces /
    skips the DC-tail term of the zero-padded template (detect.py:158) on every row; None (default) decides per row
    on the data and leaves out only what cannot exceed TAIL_EPS of the row's largest correlation (_apply_tails).
    zero_rows: "zeros" (default) / "nan" -- what an all-zero row gives; "nan" is the reference's 0 / 0
    (default "nan" under das4whales_amd.set_strict_reference(True))."""
    if zero_rows is None:
        zero_rows = "nan" if STRICT_REFERENCE else "zeros"
    if zero_rows not in ("zeros", "nan"):
        raise ValueError('zero_rows must be "zeros" or "nan"')
    xd, outs = _correlograms(data, templates, exact_tail)
    if zero_rows == "nan":
        _nan_dead_rows(xd, outs)
    return [dev.like_input(o, data) for o in outs]


def _correlograms(data, templates, exact_tail):
    if getattr(data, "ndim", 0) != 2:
        raise ValueError("data must be a 2-D [channel x time] array")
    xd = dev.to_device_f32(data)
    nx, ns = xd.shape
    taps = [_normalised_support(t) for t in templates]
    coefs = [_tail_coef(t) for t in templates]
    tails = exact_tail is not False and any(c != 0.0 for c in coefs)
    how = _xcorr_method(taps, ns, "auto")                   # decided once (an override that does not apply warns once)
    if tails and _tails_in_kernel(taps, coefs, ns, how):
        # round 6: the term is formed inside the correlator (prefix sums in its sample-conversion phase) -- exact on every row,
        # one pass over the block, no per-row decision
        return xd, _xcorr_device(xd, taps, normalize=True, method=how, stats=_row_stats_cached(xd), tails=coefs)
    by_row = tails and exact_tail is None and how == "mm"      # the form that leaves row maxima
    stats = _row_stats_cached(xd, prefix=by_row) if tails else None
    rmax = [] if by_row else None
    outs = _xcorr_device(xd, taps, normalize=True, method=how, stats=stats[:2] if stats else None, row_max=rmax)
    if tails:
        _apply_tails(xd, stats[:2], outs, taps, coefs, rmax, exact_tail, pmax=stats[2] if by_row else None)
    return xd, outs


def compute_cross_correlogram(data, template, exact_tail=None, zero_rows=None):
    """Peak-normalised matched filter, every row against `template` -- reference detect.py:140-166.

    Rows: (x - mean) / max|x| (max of the un-de-meaned row, detect.py:157).  Output is floating
    point (the reference's np.empty_like would truncate integer input); an all-zero row gives
    zeros where the reference divides by zero -- zero_rows="nan" gives the reference's NaN row."""
    return compute_cross_correlograms(data, [template], exact_tail=exact_tail, zero_rows=zero_rows)[0]


# ---------------------------------------------------------------------------------------------
# peak picking (d4w_find_peaks_f32; the envelope of the *_env variants is d4w_analytic_f32)
# ---------------------------------------------------------------------------------------------
class PickRows(collections.abc.Sequence):
    """The picks of every channel: a read-only Sequence that behaves like the list of per-channel int64 index arrays the
    reference returns (detect.py:169-274: len(), indexing, slicing, iteration, `in`, reversed(), in channel order), backed
    by ONE packed 2 x K table on the device (`packed`: row 0 = channel, row 1 = time index = detect.convert_pick_times'
    output) and `counts`.  Nothing is copied to the host until a row (or the table) is asked for; the first access copies
    the whole table once.  Deviation from the reference: it is not a `list` instance -- code that needs one (isinstance
    checks, `+`, append, np.asarray(..., dtype=object)) calls .tolist()."""

    def tolist(self):
        """A plain Python list of per-channel int64 ndarrays, exactly the reference's return type."""
        return list(self)

    def __add__(self, other):
        return self.tolist() + list(other)

    def __radd__(self, other):
        return list(other) + self.tolist()

    def __eq__(self, other):
        try:
            return len(self) == len(other) and all(np.array_equal(a, b) for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    __hash__ = None

    def __init__(self, packed, counts, resolve=None):
        """resolve: a callable returning the packed table, run on first use (pick_times*(..., lazy=True): the picker has
        been launched, the one host synchronisation of the call -- the table's size -- waits until somebody looks)."""
        self._packed, self.counts, self._resolve = packed, counts, resolve
        self._host = None

    @property
    def packed(self):
        if self._packed is None:
            self._packed = self._resolve()
            self._resolve = None
        return self._packed

    def _rows(self):
        if self._host is None:
            cnt = self.counts.cpu().numpy().astype(np.int64)
            self._host = (self.packed[1].cpu().numpy(), np.concatenate(([0], np.cumsum(cnt))))
        return self._host

    def __len__(self):
        return int(self.counts.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        flat, off = self._rows()
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("channel index out of range")
        return flat[off[i]:off[i + 1]]

    def __iter__(self):
        flat, off = self._rows()
        return (flat[off[i]:off[i + 1]] for i in range(len(self)))

    @property
    def total(self):
        return int(self.packed.shape[1])

    def table(self):
        """2 x K int64 ndarray (channel, time): detect.convert_pick_times of these picks."""
        return self.packed.cpu().numpy()


_PICK_CAP = {}                       # (nx, ns) -> index slots per row worth allocating up front
_PICK_CAP_BYTES = 8 << 30            # ... while the index table stays below this


def _find_peaks_device(c, threshold, cap0=1024, lazy=False):
    """c: float32 CUDA [nx, ns] -> PickRows.  One host synchronisation per call (total and largest per-row count);
    the ragged result is compacted on the device.  threshold: a number or a Threshold (formed on the device).  lazy=True:
    the picker is launched and the synchronisation (and the compaction behind it) is left to the first look at the result
    -- a chain over many blocks keeps the device busy and looks at the end; `c` stays alive until then."""
    nx, ns = c.shape
    # the widest row of the last call on this shape is the first guess of the next one: a stream of blocks with dense picks
    # (raw correlograms: thousands per row) would otherwise run the picker twice per block
    hint = _PICK_CAP.get((nx, ns), 0)
    if nx * hint * 4 > _PICK_CAP_BYTES:
        hint = 0
    cap = max(1, min(ns // 2 + 1, max(int(cap0), hint)))
    on_dev = isinstance(threshold, Threshold)
    if on_dev and threshold.value.device != c.device:
        raise ValueError("the threshold's value lives on another device than the block")
    cnt = torch.empty(nx, dtype=torch.int32, device=c.device)
    off = torch.empty(nx, dtype=torch.int64, device=c.device)
    summ = torch.empty(2, dtype=torch.int64, device=c.device)
    stream = torch.cuda.current_stream(c.device)

    def launch(cap):
        idx = torch.empty((nx, cap), dtype=torch.int32, device=c.device)
        if on_dev:
            check(lib.d4w_find_peaks_dthr_f32(dev.ptr(c), nx, ns, dev.ptr(threshold.value), threshold.scale, dev.ptr(idx),
                                              dev.ptr(cnt), cap, int(stream.cuda_stream)))
        else:
            check(lib.d4w_find_peaks_f32(dev.ptr(c), nx, ns, float(threshold), dev.ptr(idx), dev.ptr(cnt), cap,
                                         int(stream.cuda_stream)))
        check(lib.d4w_pick_offsets_i64(dev.ptr(cnt), nx, dev.ptr(off), dev.ptr(summ), int(stream.cuda_stream)))
        return idx

    def finish(idx, cap):
        with torch.cuda.device(c.device), torch.cuda.stream(stream):
            while True:
                need, total = (int(v) for v in summ.cpu())       # the call's one host synchronisation (a 16-byte copy)
                _PICK_CAP[(nx, ns)] = need + need // 4 + 16 if need > cap0 else 0
                if need <= cap:
                    break
                cap = min(ns // 2 + 1, max(need, 2 * cap))      # rare: a row with more peaks than the first guess
                idx = launch(cap)
            packed = torch.empty((2, total), dtype=torch.int64, device=c.device)
            check(lib.d4w_pack_picks_i64(dev.ptr(idx), dev.ptr(cnt), dev.ptr(off), nx, cap, total,
                                         dev.ptr(packed) if total else None, int(stream.cuda_stream)))
        return packed

    with torch.cuda.device(c.device):
        idx = launch(cap)
    if lazy:
        return PickRows(None, cnt, resolve=lambda: finish(idx, cap))
    return PickRows(finish(idx, cap), cnt)


def pick_times_env(corr_m, threshold, lazy=False):
    """Per row find_peaks(|hilbert(corr)|, prominence=threshold)[0] -- reference detect.py:169-195.  threshold: a number or a
    Threshold (formed on the device); lazy: see _find_peaks_device (both beyond the reference's signature)."""
    from . import dsp
    if getattr(corr_m, "ndim", 0) != 2:
        raise ValueError("corr_m must be a 2-D [channel x time] array")
    env = dsp._analytic(dev.to_device_f32(corr_m), 0)
    return _find_peaks_device(env, threshold, lazy=lazy)


def process_corr(corr, threshold):
    """One row of pick_times_env -- reference detect.py:198-218."""
    c = corr if dev.is_tensor(corr) else np.asarray(corr)
    return pick_times_env(c.reshape(1, -1), threshold)[0]


def pick_times_par(corr_m, threshold):
    """pick_times_env; the reference's thread-pool variant (detect.py:221-246) returns rows in
    completion order, here rows come back in channel order."""
    return pick_times_env(corr_m, threshold)


def pick_times(corr_m, threshold, lazy=False):
    """Per row find_peaks(corr, prominence=threshold)[0] -- reference detect.py:249-274 (threshold / lazy: pick_times_env)."""
    if getattr(corr_m, "ndim", 0) != 2:
        raise ValueError("corr_m must be a 2-D [channel x time] array")
    return _find_peaks_device(dev.to_device_f32(corr_m), threshold, lazy=lazy)


# ---------------------------------------------------------------------------------------------
# spectrogram correlation
# ---------------------------------------------------------------------------------------------
def _keep_bins(fs, nperseg, fmin, fmax):
    ff = np.linspace(0, fs / 2, num=nperseg // 2 + 1)                            # detect.py:386
    keep = np.where((ff >= fmin) & (ff <= fmax))[0]                              # detect.py:390
    if len(keep) == 0:
        raise ValueError("no STFT bin between fmin = %g and fmax = %g" % (fmin, fmax))
    return ff, int(keep[0]), int(keep[-1])


def get_sliced_nspectrogram(trace, fs, fmin, fmax, nperseg, nhop, plotflag=False):
    """|librosa.stft(trace, nperseg, nhop)| / max, rows with fmin <= f <= fmax; returns (p, ff, tt)
    -- reference detect.py:334-408 (plotflag is a plotting path of the reference and is ignored)."""
    from . import dsp
    if getattr(trace, "ndim", 0) != 1:
        raise ValueError("trace must be 1-D")
    ff, lo, hi = _keep_bins(fs, nperseg, fmin, fmax)
    x = dev.to_device_f32(trace.reshape(1, -1))
    S, mx = dsp._stft_mag(x, nperseg, nhop, lo, hi)
    dsp._scale_rows(S, mx, 0)                                                   # detect.py:387
    tt = np.linspace(0, trace.shape[0] / fs, num=S.shape[2])                    # detect.py:385
    return dev.like_input(S[0], trace), ff[lo:hi + 1], tt


def buildkernel(f0, f1, bdwdth, dur, f, t, samp, fmin, fmax, plotflag=False):
    """Hat-function kernel along a hyperbolic down-sweep, Hann-weighted in time; returns
    (tvec, fvec, BlueKernel) -- reference detect.py:411-492 (host, a few hundred values)."""
    t = np.asarray(t)
    tvec = np.linspace(0, dur, np.size(np.nonzero((t < dur * 8) & (t > dur * 7))))      # detect.py:456
    fvec = np.asarray(f)
    x = fvec[:, None] - (f0 * f1 * dur / ((f0 - f1) * tvec[None, :] + f1 * dur))        # detect.py:470
    kdist = (1 - np.square(x) / (bdwdth * bdwdth)) * np.exp(-np.square(x) / (2 * (bdwdth * bdwdth)))
    return tvec, fvec, kdist * np.hanning(len(tvec))[np.newaxis, :]                     # detect.py:474


def buildkernel_from_template(fmin, fmax, dur, fs, nperseg, nhop, plotflag=False):
    """Sliced normalised spectrogram of the windowed hyperbolic chirp -- reference detect.py:495-541."""
    template = gen_hyperbolic_chirp(fmin, fmax, dur, fs)
    template *= np.hanning(len(template))
    spectro, _, _ = get_sliced_nspectrogram(template, fs, fmin, fmax, nperseg, nhop, plotflag=False)
    return spectro


_kernel_memo = {}       # (kernel bytes, shape, device) -> device copy of a spectrogram-correlation kernel


def _kernel_on_device(K, device):
    key = (K.tobytes(), K.shape, str(device))
    with _cache_lock:
        Kd = _kernel_memo.get(key)
        if Kd is None:
            if len(_kernel_memo) > 16:
                _kernel_memo.clear()
            Kd = _kernel_memo[key] = torch.from_numpy(K).to(device)
    return Kd


def _spectrocorr_device(S, K, off, nout, med=None, zero_ends=False, out=None):
    """S: float32 CUDA [nx, nf, nt]; K: host [nf, nk] -> CUDA [nx, nout] (include/d4w.h d4w_spectrocorr_f32).
    out: optional contiguous [nx, nout] float32 CUDA rows to write into (a slice of a larger result)."""
    nx, nf, nt = S.shape
    K = np.ascontiguousarray(K, dtype=np.float32)
    if K.ndim != 2 or K.shape[0] != nf:
        raise ValueError("kernel has %s rows, the spectrogram %d" % (K.shape[:1], nf))
    Kd = _kernel_on_device(K, S.device)          # the same few hundred values file after file: uploaded once
    if out is None:
        out = torch.empty((nx, nout), dtype=torch.float32, device=S.device)
    with torch.cuda.device(S.device):
        if med is None:
            med = torch.empty(nx, dtype=torch.float32, device=S.device)
            check(lib.d4w_row_median_f32(dev.ptr(S), nx, nf * nt, dev.ptr(med), dev.stream_ptr(S)))   # detect.py:600
        check(lib.d4w_spectrocorr_f32(dev.ptr(S), nx, nf, nt, dev.ptr(Kd), K.shape[1], int(off), int(nout),
                                      dev.ptr(med), int(bool(zero_ends)), dev.ptr(out), dev.stream_ptr(S)))
        # Kd is a temporary on the stream the kernel runs on: the caching allocator re-uses it in stream order, no
        # host synchronisation (a stream of files stays asynchronous)
    return out


def _spectro_2d(spectro):
    if getattr(spectro, "ndim", 0) != 2:
        raise ValueError("spectro must be a 2-D [frequency x time] array")
    return dev.to_device_f32(spectro)[None]


def xcorr2d(spectro, kernel):
    """sum_f fftconvolve(spectro, flip(kernel, 1), 'same', axes=1), clipped at 0,
    / (median(spectro) * kernel.shape[1]) -- reference detect.py:579-602."""
    S = _spectro_2d(spectro)
    kernel = np.asarray(kernel)
    out = _spectrocorr_device(S, kernel, kernel.shape[1] // 2, S.shape[2])
    return dev.like_input(out[0], spectro)


def xcorr(t, f, Sxx, tvec, fvec, BlueKernel):
    """Valid-lag kernel x spectrogram correlation; returns [t_scale, CorrVal] -- reference detect.py:605-647."""
    nk, nfk = np.size(tvec), np.size(fvec)
    S = _spectro_2d(Sxx)
    med = torch.empty(1, dtype=torch.float32, device=S.device)
    with torch.cuda.device(S.device):                           # np.median(Sxx) is over the whole spectrogram
        check(lib.d4w_row_median_f32(dev.ptr(S), 1, S[0].numel(), dev.ptr(med), dev.stream_ptr(S)))
    S = S[:, :nfk].contiguous()                                 # detect.py:636: Sxx[:fvec_size, ...]
    nout = np.size(t) - (nk - 1)
    out = _spectrocorr_device(S, np.asarray(BlueKernel)[:nfk], 0, nout, med=med, zero_ends=True)
    t = np.asarray(t)
    t_scale = t[int(nk / 2) - 1:-int(np.ceil(nk / 2))]          # detect.py:645
    return [t_scale, dev.like_input(out[0], Sxx)]


def nxcorr2d(spectro, kernel):
    """max over the frequency lag of correlate(spectro, kernel, 'same') / (std(spectro) std(kernel) nt)
    -- reference detect.py:544-576.  One d4w_spectrocorr_f32 launch per frequency lag."""
    S = _spectro_2d(spectro)
    K = np.asarray(kernel, dtype=np.float64)
    nf, nt = S.shape[1], S.shape[2]
    nfk, nk = K.shape
    var = torch.empty(1, dtype=torch.float32, device=S.device)
    with torch.cuda.device(S.device):
        check(lib.d4w_row_var_f32(dev.ptr(S), 1, nf * nt, dev.ptr(var), dev.stream_ptr(S)))
    # out = raw / (med * nk) with med chosen so that the divisor is std(S) std(K) nt
    med = torch.sqrt(var) * float(np.std(K) * nt / nk)
    best = None
    for a in range(nf):                                          # output row a: K row fk meets S row a + fk - nfk//2
        Ka = np.zeros((nf, nk))
        lo = a - nfk // 2
        for fk in range(nfk):
            if 0 <= lo + fk < nf:
                Ka[lo + fk] = K[fk]
        # clipping at 0 is harmless only when the max is >= 0; keep exact semantics via two signs
        pos = _spectrocorr_device(S, Ka, nk // 2, nt, med=med)
        neg = _spectrocorr_device(S, -Ka, nk // 2, nt, med=med)
        row = pos - neg
        best = row if best is None else torch.maximum(best, row)
    return dev.like_input(best[0], spectro)


def compute_cross_correlogram_spectrocorr(data, fs, flims, kernel, win_size, overlap_pct):
    """Per-channel sliced spectrogram x hat kernel correlation -- reference detect.py:650-709.
    All channels go through one STFT launch, one median launch and one correlation launch; the
    spectrogram's max-normalisation (detect.py:387) cancels between numerator and median."""
    from . import dsp
    if getattr(data, "ndim", 0) != 2:
        raise ValueError("data must be a 2-D [channel x time] array")
    nperseg = int(win_size * fs)                                                 # detect.py:680
    nhop = int(np.floor(nperseg * (1 - overlap_pct)))                            # detect.py:681
    fmin, fmax = flims
    f1, f0, duration, bandwidth = kernel["f1"], kernel["f0"], kernel["dur"], kernel["bdwidth"]
    if fmax - f1 < 2 * bandwidth:                                                # detect.py:693-696
        fmax = f1 + 3 * bandwidth
    if f0 - fmin < 2 * bandwidth:
        fmin = f0 - 3 * bandwidth
    ff, lo, hi = _keep_bins(fs, nperseg, fmin, fmax)
    x = dev.to_device_f32(data)
    nx, ns = x.shape
    nt = int(lib.d4w_stft_frames(ns, nhop))
    tt = np.linspace(0, ns / fs, num=nt)                                         # detect.py:385
    _, _, ker = buildkernel(f0, f1, bandwidth, duration, ff[lo:hi + 1], tt, fs, fmin, fmax)   # detect.py:702
    out = torch.empty((nx, nt), dtype=torch.float32, device=x.device)
    per_ch = (hi - lo + 1) * nt * 4
    step = int(max(1, min(65535, (2 << 30) // per_ch)))          # <= 2 GiB of spectrogram in flight
    for a in range(0, nx, step):
        S, _ = dsp._stft_mag(x[a:a + step], nperseg, nhop, lo, hi, want_max=False)     # the max-normalisation cancels (docstring)
        _spectrocorr_device(S, ker, ker.shape[1] // 2, nt, out=out[a:a + step])        # rows a.. of the result, in place
    return dev.like_input(out, data)


# ---------------------------------------------------------------------------------------------
# pick utilities (index bookkeeping, host)
# ---------------------------------------------------------------------------------------------
def convert_pick_times(peaks_indexes_m):
    """Ragged per-channel index lists -> 2 x K array, row 0 = channel index, row 1 = time index
    -- reference detect.py:277-303."""
    if isinstance(peaks_indexes_m, PickRows):
        return peaks_indexes_m.table()
    ch = [np.full(len(p), i, dtype=np.int64) for i, p in enumerate(peaks_indexes_m)]
    if not ch:
        return np.zeros((2, 0), dtype=np.int64)
    return np.asarray((np.concatenate(ch), np.concatenate([np.asarray(p, dtype=np.int64) for p in peaks_indexes_m])))


def select_picked_times(idx_tp, tstart, tend, fs):
    """Keep picks with tstart*fs <= time index <= tend*fs -- reference detect.py:306-330."""
    keep = (idx_tp[1] >= tstart * fs) & (idx_tp[1] <= tend * fs)
    return (idx_tp[0][keep], idx_tp[1][keep])
