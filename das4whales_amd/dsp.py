"""dsp -- MI355X-native mirror of das4whales.dsp (reference: src/das4whales/dsp.py).

Same function names, positional signatures and array conventions as the reference; the array
arithmetic runs in the HIP library (include/d4w.h).  Filter *design* of 1-D IIR coefficients stays
on the host in float64 (SciPy), exactly like the reference does.
"""
import atexit
import ctypes
import threading
import weakref

import numpy as np
import torch

from . import _device as dev
from ._lib import lib, check


# ---------------------------------------------------------------------------------------------
# f-k plans (one per device x shape), with the most recent mask kept folded on the device
# ---------------------------------------------------------------------------------------------
class FkPlan:
    """Owns a d4w_fk_plan (include/d4w.h).  `opts` = (C1, C2, N1, N2, TA, TC) overrides."""

    def __init__(self, nx, ns, opts=None, device=None):
        dev.require_gpu()
        self.device = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.nx, self.ns = int(nx), int(ns)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            o = (ctypes.c_int * 6)(*[int(v) for v in opts]) if opts is not None else None
            check(lib.d4w_fk_plan_create_ex(self.nx, self.ns, o, ctypes.byref(self._h)))
        self._mask_key = None
        self._mask_ref = None
        # A plan is shared by every thread that filters this shape on this device (get_fk_plan): the public functions hold
        # this lock over [check the mask key, design, fold the mask, launch the filter], so that two threads with different
        # masks cannot interleave.  The launches are asynchronous -- the lock covers host time only -- and the library
        # orders applies of one plan across streams itself where they share plan-owned scratch (fk_apply_impl).
        self.lock = threading.RLock()

    def info(self):
        v = (ctypes.c_int * 8)()
        check(lib.d4w_fk_plan_info(self._h, v))
        return dict(zip(("nx", "ns", "C1", "C2", "N1", "N2", "TA", "TC"), list(v)))

    def live_rows(self):
        """Wavenumber rows the current mask keeps alive (nx = nothing is skipped), include/d4w.h."""
        return int(lib.d4w_fk_plan_live_rows(self._h))

    def order(self):
        """Order of the passes for the current mask and the modelled traffic of both orders (include/d4w.h
        d4w_fk_plan_order): time-first (half spectrum compacted to the frequency columns the mask needs) or
        channel-first (dead wavenumber rows skipped)."""
        v, b = (ctypes.c_int * 6)(), (ctypes.c_double * 2)()
        check(lib.d4w_fk_plan_order(self._h, v, b))
        return {"order": "time-first" if v[0] else "channel-first", "band_columns": v[1], "tail_columns": v[2],
                "half_spectrum_columns": v[3], "live_wavenumber_rows": v[4], "nx": v[5],
                "model_bytes_per_sample": {"channel-first": b[0], "time-first": b[1]}}

    def set_mask(self, fk_filter_matrix, prune_eps=0.0):
        """Dense ndarray (any order / float dtype), sparse.COO-like (.todense()) or CUDA tensor,
        on the fftshift-ed grid, shape [nx, ns] -- what the reference designs return.
        prune_eps > 0 (opt-in, not exact): wavenumber rows whose folded gains stay below
        prune_eps * max are skipped like all-zero rows (include/d4w.h d4w_fk_set_mask_dense_pruned_f32).
        prune_eps = 0 is exact to float32 rounding, not bit for bit: gains below D4W_FK_ROUND_EPS (default 2^-24)
        / sqrt(nx ns) * max|M_h| count as zero and columns whose gain spread stays below that are taken at their mean
        (what they add to any output sample is under half an ulp of rms(x) max|M_h|; the threshold follows the LARGEST
        gain, so one outlier gain raises it for all).  D4W_FK_ROUND_EPS=0 keeps exact zeros only."""
        m = fk_filter_matrix
        if isinstance(m, DesignedMask):
            return self._set_mask_design(m, prune_eps)
        if isinstance(m, DeviceMask):
            m = m.tensor
        # Only objects that carry a modification counter are cached (torch tensors, hence DeviceMask): a NumPy
        # array or a sparse.COO edited in place (`mask *= w`) keeps its id(), so those are re-folded every call.
        version = getattr(m, "_version", None) if dev.is_tensor(m) else None
        key = (id(m), version, float(prune_eps)) if version is not None else None
        if key is not None and self._mask_key == key and self._mask_ref is not None and self._mask_ref() is m:
            return
        self._mask_ref, self._mask_key = None, None
        if hasattr(m, "todense") and not dev.is_tensor(m):
            md = m.todense()
        else:
            md = m
        shape = tuple(md.shape)
        if shape != (self.nx, self.ns):
            raise ValueError("operands could not be broadcast together with shapes (%d,%d) %s"
                             % (self.nx, self.ns, shape))
        t = dev.to_device_f32(md, self.device)
        with torch.cuda.device(self.device):
            check(lib.d4w_fk_set_mask_dense_pruned_f32(self._h, dev.ptr(t), float(prune_eps), dev.stream_ptr(t)))
            torch.cuda.current_stream().synchronize()   # t may be a temporary
        if key is not None:
            self._mask_ref = weakref.ref(m)
            self._mask_key = key

    def set_mask_normalised(self, g, key=None):
        """g: float32 CUDA tensor [nx, ns] on the plan's device; the mask is (g - min) / (max - min) (dsp.fk_filt, reference
        dsp.py:945), applied while the mask is folded -- no separate normalisation pass over g.  key: what the caller wants
        to recognise this mask by later (FkPlan._mask_key), recorded after the fold."""
        if tuple(g.shape) != (self.nx, self.ns):
            raise ValueError("operands could not be broadcast together with shapes (%d,%d) %s" % (self.nx, self.ns, tuple(g.shape)))
        self._mask_ref, self._mask_key = None, None
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device).cuda_stream
            mm = torch.empty(2, dtype=torch.float32, device=self.device)
            check(lib.d4w_minmax_f32(dev.ptr(g), g.numel(), dev.ptr(mm), st))
            lohi = mm.cpu().numpy()
            lo, hi = np.float32(lohi[0]), np.float32(lohi[1])
            # a constant g: NumPy's 0 / 0 = NaN everywhere (as d4w_minmax_normalise_f32)
            a = np.float32(1.0) / (hi - lo) if hi > lo else np.float32(np.nan)
            b = -lo * a
            check(lib.d4w_fk_set_mask_dense_affine_f32(self._h, dev.ptr(g), float(a), float(b), st))
        self._mask_key = key

    def _set_mask_design(self, m, prune_eps):
        """A closed-form design (immutable) straight into the plan -- no dense mask."""
        key = (id(m), "design", float(prune_eps))
        if self._mask_key == key and self._mask_ref is not None and self._mask_ref() is m:
            return
        self._mask_ref, self._mask_key = None, None
        if m.shape != (self.nx, self.ns):
            raise ValueError("operands could not be broadcast together with shapes (%d,%d) %s" % (self.nx, self.ns, m.shape))
        h = m.hrow_on(self.device)
        p8 = (ctypes.c_double * 8)(*m.params)
        with torch.cuda.device(self.device):
            check(lib.d4w_fk_set_mask_design_f32(self._h, m.mode, m.k_spacing, m.t_spacing, p8, m.i0, m.i1,
                                                 dev.ptr(h) if h is not None else None, float(prune_eps),
                                                 torch.cuda.current_stream(self.device).cuda_stream))
        self._mask_ref, self._mask_key = weakref.ref(m), key

    def apply(self, x, out=None, taper=False):
        """x: float32 CUDA tensor [nx, ns]; returns the filtered tensor (out may alias x)."""
        if tuple(x.shape) != (self.nx, self.ns):
            raise ValueError("trace shape %s does not match the plan (%d, %d)" % (tuple(x.shape), self.nx, self.ns))
        if out is None:
            out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            check(lib.d4w_fk_apply_f32(self._h, dev.ptr(x), dev.out_ptr(out), int(bool(taper)), dev.stream_ptr(x)))
        return out

    def apply_stats(self, x, out=None, taper=False, timed=False):
        """apply() that also returns (mean, maxabs) of every filtered row -- what
        detect.compute_cross_correlogram normalises by -- formed in the last pass's epilogue.
        timed=True additionally returns the five pass times in ms."""
        if tuple(x.shape) != (self.nx, self.ns):
            raise ValueError("trace shape %s does not match the plan (%d, %d)" % (tuple(x.shape), self.nx, self.ns))
        if out is None:
            out = torch.empty_like(x)
        mean = torch.empty(self.nx, dtype=torch.float64, device=x.device)      # float64: include/d4w.h, d4w_row_stats_f32
        mx = torch.empty(self.nx, dtype=torch.float32, device=x.device)
        with torch.cuda.device(self.device):
            if timed:
                ms = (ctypes.c_float * 5)()
                check(lib.d4w_fk_apply_timed_stats_f32(self._h, dev.ptr(x), dev.out_ptr(out), int(bool(taper)), dev.ptr(mean),
                                                       dev.ptr(mx), dev.stream_ptr(x), ms))
                return out, mean, mx, list(ms)
            check(lib.d4w_fk_apply_stats_f32(self._h, dev.ptr(x), dev.out_ptr(out), int(bool(taper)), dev.ptr(mean),
                                             dev.ptr(mx), dev.stream_ptr(x)))
        return out, mean, mx

    def apply_stats_prefix(self, x, out=None, taper=False):
        """apply_stats() that also returns the rows' prefix maxima max_j |sum_{i<j} (y - mean)| (d4w_row_prefix_max_f32: what
        bounds the DC-tail term of a zero-padded template, detect._apply_tails).  Where the plan would sweep the result for the
        statistics anyway (d4w_fk_stats_in_epilogue == 0: the 60-s file shapes) one launch forms all three, the second
        sweep of a row served by L2."""
        if int(lib.d4w_fk_stats_in_epilogue(self._h)):
            out, mean, mx = self.apply_stats(x, out=out, taper=taper)
            pm = torch.empty(self.nx, dtype=torch.float32, device=x.device)
            with torch.cuda.device(self.device):
                check(lib.d4w_row_prefix_max_f32(dev.ptr(out), self.nx, self.ns, dev.ptr(mean), dev.ptr(pm), dev.stream_ptr(x)))
            return out, mean, mx, pm
        out = self.apply(x, out=out, taper=taper)
        mean = torch.empty(self.nx, dtype=torch.float64, device=x.device)
        mx = torch.empty(self.nx, dtype=torch.float32, device=x.device)
        pm = torch.empty(self.nx, dtype=torch.float32, device=x.device)
        with torch.cuda.device(self.device):
            check(lib.d4w_row_stats_prefix_f32(dev.ptr(out), self.nx, self.ns, dev.ptr(mean), dev.ptr(mx), dev.ptr(pm),
                                               dev.stream_ptr(x)))
        return out, mean, mx, pm

    def apply_timed(self, x, out=None, taper=False):
        """Like apply() but returns (out, [ms per pass A, C, B, C', A']) via HIP events."""
        if out is None:
            out = torch.empty_like(x)
        ms = (ctypes.c_float * 5)()
        with torch.cuda.device(self.device):
            check(lib.d4w_fk_apply_timed_f32(self._h, dev.ptr(x), dev.out_ptr(out), int(bool(taper)), dev.stream_ptr(x), ms))
        return out, list(ms)

    def __del__(self):
        try:
            if self._h:
                lib.d4w_fk_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def supported_length(n, even=False):
    """Largest length <= n whose prime factors are all <= 31 (what the mixed-radix kernels carry), optionally
    even.  Everything runs at other lengths too, slower: the f-k filter takes any shape (prime factors > 31 run as
    Bluestein convolutions -- several times slower when that part exceeds 4096 channels / 2048 of ns / 2), so do
    hilbert / spectrogram windows / get_fx up to their LDS limits.  Use this to trim a record or a selection onto
    the direct kernels, e.g. 12002 samples -> 12000."""
    n = int(n)
    while n > 1:
        m = n
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31):
            while m % p == 0:
                m //= p
        if m == 1 and (not even or n % 2 == 0):
            return n
        n -= 1
    return max(n, 1)


_plans = {}
_plans_lock = threading.Lock()
# one lock for the small per-(device, stream) caches of this package (side streams, zero-phase responses, overlap-save
# workspaces, template spectra): look-up-or-create is atomic, entries are keyed by the calling stream, so two Python
# threads on two streams never share a workspace (SURVEY 8b "threading")
_cache_lock = threading.RLock()


def compile_fk_shape(nx, ns, verbose=False, warn=False):
    """Compile (once, cached on disk) shape-specialised f-k kernels for [nx, ns] -- any shape whose axes factor into
    parts <= 32; see das4whales_amd/fkjit.py.  Returns True when the shape runs specialised kernels afterwards.
    Plans created before the call keep the kernels they were planned with (drop them with dsp.clear_fk_plans()).
    warn=True: a RuntimeWarning names the reason when the shape stays on the generic kernels."""
    from . import fkjit
    return fkjit.compile_fk_shape(nx, ns, verbose=verbose, warn=warn)


def _auto_specialise(nx, ns):
    """A new large block (>= 2^24 samples) gets its own f-k kernels (~15-40 s once, cached on disk).  Never raises: when it
    cannot (no configuration, no compiler, a failed build -- remembered, not retried) the generic kernels run, 3-8x slower
    at this size, and a RuntimeWarning says why."""
    import os
    if os.environ.get("D4W_FK_JIT", "1") == "0" or int(nx) * int(ns) < (1 << 24):
        return
    try:
        compile_fk_shape(nx, ns, warn=True)
    except Exception as e:                      # e.g. an unwritable cache directory
        import warnings
        warnings.warn("das4whales_amd: compiling f-k kernels for %d x %d failed (%r); the generic kernels run 3-8x slower "
                      "at this size" % (nx, ns, e), RuntimeWarning, stacklevel=3)


def clear_fk_plans():
    with _plans_lock:
        _plans.clear()


# cached plans own device memory: free it while the HIP runtime is still up, not during interpreter teardown
atexit.register(clear_fk_plans)


def get_fk_plan(nx, ns, device=None):
    device = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
    key = (str(device), int(nx), int(ns))
    with _plans_lock:
        known = key in _plans
    if not known:
        _auto_specialise(nx, ns)          # outside the lock: a compile takes tens of seconds (fkjit has its own lock)
    with _plans_lock:
        p = _plans.pop(key, None)
        if p is None:
            if len(_plans) >= 4:          # plans hold an nx*ns/2 float mask each: keep few, drop the least recently used
                _plans.pop(next(iter(_plans)))
            p = FkPlan(nx, ns, device=device)
        _plans[key] = p                   # (re-)insert at the most-recent end
        return p


def _fk_apply_odd(trace, fk_filter_matrix, tapering):
    """Odd record length, exactly, through the even-length (packed real) machinery: z[2n] = x[n], z[2n+1] = 0 has
    the spectrum of x repeated twice along f, so filtering z with the mask repeated twice along f returns y
    interleaved with zeros.  Twice the work of an even length; only data movement happens here."""
    nx, ns = trace.shape
    device = trace.device if dev.is_tensor(trace) and trace.is_cuda else None
    m = fk_filter_matrix.tensor if isinstance(fk_filter_matrix, DeviceMask) else fk_filter_matrix
    if hasattr(m, "todense") and not dev.is_tensor(m):
        m = m.todense()
    if tuple(m.shape) != (nx, ns):
        raise ValueError("operands could not be broadcast together with shapes (%d,%d) %s" % (nx, ns, tuple(m.shape)))
    plan = get_fk_plan(nx, 2 * ns, device)
    md = dev.to_device_f32(m, plan.device)
    mu = torch.roll(md, shifts=-(ns // 2), dims=1)              # time axis back to the unshifted grid
    x = dev.to_device_f32(trace, plan.device)
    if tapering:
        x = x.clone()
        check(lib.d4w_taper_f32(dev.ptr(x), nx, ns, dev.stream_ptr(x)))
    x2 = torch.zeros((nx, 2 * ns), dtype=torch.float32, device=x.device)
    x2[:, 0::2] = x
    with plan.lock:
        plan.set_mask(torch.cat((mu, mu), dim=1))                # periodic in f: fftshift by ns leaves it unchanged
        y = plan.apply(x2)[:, 0::2].contiguous()
    return dev.like_input(y, trace)


def _fk_apply(trace, fk_filter_matrix, tapering):
    if getattr(trace, "ndim", 0) != 2:
        raise ValueError("trace must be a 2-D [channel x time] array")
    nx, ns = trace.shape
    if ns % 2:
        return _fk_apply_odd(trace, fk_filter_matrix, tapering)
    device = trace.device if dev.is_tensor(trace) and trace.is_cuda else None
    plan = get_fk_plan(nx, ns, device)
    x = dev.to_device_f32(trace, plan.device)
    # A CUDA block in, a CUDA block out: the reference's chain goes on to the matched filter (scripts/main_mfdetect.py:55-80),
    # which normalises every row by its mean and max|.| (detect.py:157).  Where the filter's last pass can leave the two in its
    # epilogue for 2-3 % of its time (d4w_fk_stats_in_epilogue: the long-row shapes, where a separate sweep of the result costs
    # 12 %) it does, and the result carries them (detect._remember_row_stats) -- detect.compute_cross_correlogram on that
    # tensor then starts at the correlator.  D4W_FK_STATS_HINT=0: never.
    import os
    hint = (device is not None and trace.dtype == torch.float32 and os.environ.get("D4W_FK_STATS_HINT", "1") != "0")
    stats = None
    with plan.lock:                                              # the mask this call folds is the mask this call applies
        plan.set_mask(fk_filter_matrix)
        if hint and int(lib.d4w_fk_stats_in_epilogue(plan._h)):
            y, mean, mx = plan.apply_stats(x, taper=tapering)
            stats = (mean, mx)
        else:
            y = plan.apply(x, taper=tapering)
    if stats is not None:
        from . import detect
        detect._remember_row_stats(y, stats)
    return dev.like_input(y, trace)


def _fk_apply_stats(x, fk_filter_matrix, tapering=False, prefix=False):
    """fk_filter_filt of a float32 CUDA block that also returns (float64 row means, float32 row maxima) of the result from
    the last pass's epilogue -- what the matched filter that follows normalises by (detect.py:157); None for odd record
    lengths (the doubled-record form has no such epilogue).  prefix=True: (means, maxima, prefix maxima)
    (FkPlan.apply_stats_prefix)."""
    nx, ns = x.shape
    if ns % 2:
        return _fk_apply_odd(x, fk_filter_matrix, tapering), None
    plan = get_fk_plan(nx, ns, x.device)
    with plan.lock:
        plan.set_mask(fk_filter_matrix)
        if prefix:
            y, mean, mx, pm = plan.apply_stats_prefix(x, taper=tapering)
            return y, (mean, mx, pm)
        y, mean, mx = plan.apply_stats(x, taper=tapering)
    return y, (mean, mx)


def _strict_reference():
    from . import detect
    return bool(detect.STRICT_REFERENCE)


def fk_filter_filt(trace, fk_filter_matrix, tapering=False, inplace_taper=None):
    """Apply a pre-computed f-k mask (dense, on the fftshift-ed grid) -- reference dsp.py:725-756.

    By default `tapering=True` applies the Tukey window inside the filter's first pass and leaves `trace` as it was (SURVEY.md
    A.7 (N)); the reference tapers the caller's array IN PLACE on the way (dsp.py:744-745 -> taper_data, dsp.py:721).
    inplace_taper=True reproduces that side effect: `trace` comes back tapered, the result is the same (the default under
    das4whales_amd.set_strict_reference(True))."""
    if inplace_taper is None:
        inplace_taper = _strict_reference()
    if tapering and inplace_taper:
        taper_data(trace)
        tapering = False
    return _fk_apply(trace, fk_filter_matrix, tapering)


def fk_filter_sparsefilt(trace, fk_filter_matrix, tapering=False, inplace_taper=None):
    """Same with a sparse.COO mask -- reference dsp.py:759-786 (dense masks are accepted too).  inplace_taper: as above
    (the reference's dsp.py:775-776)."""
    if inplace_taper is None:
        inplace_taper = _strict_reference()
    if tapering and inplace_taper:
        taper_data(trace)
        tapering = False
    return _fk_apply(trace, fk_filter_matrix, tapering)


fk_filter = fk_filter_filt      # north-star spelling


def taper_data(trace):
    """trace *= tukey(ns, 0.03) along time, IN PLACE like the reference -- dsp.py:705-722."""
    if dev.is_tensor(trace) and trace.is_cuda and trace.dtype == torch.float32 and trace.is_contiguous():
        check(lib.d4w_taper_f32(dev.out_ptr(trace), trace.shape[0], trace.shape[1], dev.stream_ptr(trace)))
        return trace
    x = dev.to_device_f32(trace)
    check(lib.d4w_taper_f32(dev.ptr(x), x.shape[0], x.shape[1], dev.stream_ptr(x)))
    if dev.is_tensor(trace):
        trace.copy_(x.to(trace.dtype))
    else:
        trace[...] = x.cpu().numpy().astype(trace.dtype, copy=False)
    return trace


# ---------------------------------------------------------------------------------------------
# 1-D zero-phase filters (design on the host in float64, like the reference; application in HIP)
# ---------------------------------------------------------------------------------------------
def butterworth_filter(filterspec, fs):
    """SOS Butterworth design -- reference dsp.py:789-827 (host, SciPy, float64)."""
    import scipy.signal as sp
    filter_order, filter_critical_freq, filter_type_str = filterspec
    wn = np.array(filter_critical_freq) / (fs / 2)
    return sp.butter(filter_order, wn, btype=filter_type_str, output="sos")


_sos_host_cache = {}        # sos bytes -> {"decay": samples, "zi": steady-state initial conditions}: host work once per design


def _sos_host(sos):
    key = (sos.tobytes(), sos.shape)
    ent = _sos_host_cache.get(key)
    if ent is None:
        import scipy.signal as sp
        if len(_sos_host_cache) > 64:
            _sos_host_cache.clear()
        ent = _sos_host_cache[key] = {"zi": np.ascontiguousarray(sp.sosfilt_zi(sos), dtype=np.float64),
                                      "decay": _sos_decay_samples_uncached(sos)}
    return ent


def _sos_decay_samples(sos, tol=1e-9, nmax=1 << 17):
    """Samples after which the cascade's impulse response stays below tol * peak (host, float64; cached per design)."""
    if tol == 1e-9 and nmax == 1 << 17:
        return _sos_host(sos)["decay"]
    return _sos_decay_samples_uncached(sos, tol, nmax)


def _sos_decay_samples_uncached(sos, tol=1e-9, nmax=1 << 17):
    import scipy.signal as sp
    n = 4096
    while True:
        imp = np.zeros(n)
        imp[0] = 1.0
        h = np.abs(sp.sosfilt(sos, imp))
        big = np.nonzero(h > tol * h.max())[0]
        last = int(big[-1]) if len(big) else 0
        if last < n // 2 or n >= nmax:
            return last + 1
        n *= 2


def _sos_segment_length(nx, ns, warm, half_waves=None):
    """Time-segment length of the recursive filter.  A wave owns 64 rows x one segment and walks
    seg_len + warm samples; the GPU's rate of wave-steps saturates with the number of waves like
    waves / (waves + W_h) (measured on MI355X: W_h ~ 1000, 4.3 M wave-steps/ms when saturated), so
    the run time is about (seg_len + warm) * (groups * nseg + W_h): pick the nseg that minimises it.
    Short blocks (4000 x 12000) run a few dozen warm-started segments per row, long ones
    (20000 x 120000) ~26."""
    import os
    if half_waves is None:
        half_waves = int(os.environ.get("D4W_SOS_WH", 1024))
    groups = -(-nx // 64)
    best, best_cost = 0, float(ns + 2 * warm) * (groups + half_waves)      # nseg = 1: exact single segment
    for nseg in range(2, max(2, ns // 64) + 1):
        seg = -(-(-(-ns // nseg)) // 32) * 32
        if seg + 2 * warm >= ns:
            continue
        cost = float(seg + warm) * (groups * nseg + half_waves)
        if cost < best_cost:
            best, best_cost = seg, cost
    return best


def _zero_phase_response(sos, tol_taps=1e-8, tol_edge=1e-9, nmax=1 << 18):
    """Two-sided response g = h * h(-t) of the zero-phase cascade (host, float64): returns (taps [2K + 1], K, E) with
    K the even half width beyond which sum|g| is below tol_taps of the total, E >= K the distance from a row end
    beyond which the edge rule of filtfilt (odd extension, steady-state initial conditions) has decayed below
    tol_edge.  None when the response does not decay within nmax samples."""
    import scipy.signal as sp
    n = 4096
    while True:
        imp = np.zeros(n)
        imp[0] = 1.0
        h = sp.sosfilt(sos, imp)
        H = np.fft.rfft(h, 2 * n)
        g = np.fft.irfft(H * np.conj(H), 2 * n)[:n]             # g[m], m >= 0 (symmetric)
        a = np.abs(g)
        tail = np.cumsum(a[::-1])[::-1]
        tail = tail / max(tail[0], 1e-300)
        ok_t, ok_e = np.nonzero(tail < tol_taps)[0], np.nonzero(tail < tol_edge)[0]
        if len(ok_e) and ok_e[0] < n // 4:
            K = int(ok_t[0]) + (int(ok_t[0]) & 1)
            E = int(ok_e[0]) + (int(ok_e[0]) & 1)
            taps = np.concatenate((g[K:0:-1], g[:K + 1]))
            return taps, K, max(E, K)
        if n >= nmax:
            return None
        n *= 2


_zp_cache = {}


def _sosfiltfilt_fft(x, sos, padlen):
    """Interior by ONE overlap-save FFT pass with the truncated zero-phase response (d4w_fir_fft_f32, 8 B per sample),
    the E columns at either row end by the exact recursion on short row pieces (both ends of all rows in one call).
    Returns None when the form does not apply (response too long for the FFT block, rows shorter than a few pieces).
    Until round 5 rows shorter than ~24 pieces stayed with the segmented recursion (0.95 vs 1.8 ms at 4000 x 12000): the
    row-end pieces ran one row per lane, 0.5 ms per direction at 11 020 rows.  With the sections of a row on adjacent lanes
    (sos_pass_lanes) the pieces cost a tenth of that and the 60-s file shapes take this form too (D4W_BP_FFT_MINP = rows
    of at least that many pieces, default 4)."""
    import os
    if os.environ.get("D4W_BP_FFT", "1") == "0":
        return None
    nx, ns = x.shape
    zp = _zero_phase_taps(sos, x.device)
    if zp is None:
        return None
    t, K, E, dcg = zp
    P = 2 * E                                        # piece length: E kept + E for the artificial cut to decay
    if ns < int(os.environ.get("D4W_BP_FFT_MINP", "4")) * P or P <= padlen:
        return None
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        # the row ends: the left and the right piece of every row as one [2 nx, P] block (filtfilt's edge rule is not
        # symmetric under time reversal, so the right pieces stay in natural order and keep their LAST E outputs).  The
        # recursion on them is a chain of dependent steps on a few waves per CU: it runs on a side stream underneath the
        # bandwidth-bound overlap-save pass (D4W_BP_OVERLAP=0: one stream, one after the other)
        cur = torch.cuda.current_stream(x.device)
        side = _side_stream(x.device) if os.environ.get("D4W_BP_OVERLAP", "1") != "0" else cur
        ready = None
        if side is not cur:
            ready = torch.cuda.Event()
            ready.record(cur)                        # x is complete here: what the side stream has to wait for
        # the long kernel goes out FIRST: the device starts on it while the host is still preparing the row-end launches
        def cols(src, src_off, ld_src, dst, dst_off, ld_dst, ncols, stream):
            # one launch per strided piece (a torch slice copy of a > 2^31-element tensor is split into ~10)
            check(lib.d4w_copy_cols_f32(dev.ptr(src) + 4 * src_off, ld_src, dev.ptr(dst) + 4 * dst_off, ld_dst, nx, ncols,
                                        ctypes.c_void_p(stream.cuda_stream)))
        first = torch.empty(nx, dtype=torch.float32, device=x.device)
        cols(x, 0, ns, first, 0, 1, 1, cur)
        ent = _fir_workspace(t, x.device)
        # the interior kernel writes the columns [E, ns - E) only; the row-end columns belong to the pieces
        check(lib.d4w_fir_fft_cols_f32(dev.ptr(x), nx, ns, None if ent[1] else dev.ptr(t), int(K), dev.ptr(first), dcg, dev.ptr(y),
                                       int(E), int(ns - E), dev.ptr(ent[0]), dev.stream_ptr(x)))
        ent[1] = True
        # the row ends (d4w_sosfiltfilt_ends_f32): both pieces of every row read in place from x and their E outer outputs
        # written in place into y, the sections of a piece on adjacent lanes -- on the side stream, underneath the interior
        # kernel (disjoint columns of y: no order between the two is needed)
        zi = _sos_host(sos)["zi"]
        sos_p = sos.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        zi_p = zi.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        if side is not cur:
            side.wait_event(ready)
            x.record_stream(side)
            # (y needs no record_stream: the calling stream waits for the side stream below, before y can be used or freed;
            # recording it would make the allocator hold the 9.6-GB block back and cudaMalloc a new one per call)
        with torch.cuda.stream(side):
            ws = torch.empty(int(lib.d4w_sosfiltfilt_ends_ws_bytes(nx, P, padlen)), dtype=torch.uint8, device=x.device)
            check(lib.d4w_sosfiltfilt_ends_f32(dev.ptr(x), dev.ptr(y), nx, ns, sos_p, zi_p, sos.shape[0], int(padlen), int(P), int(E), 0,
                                               dev.ptr(ws), ctypes.c_void_p(side.cuda_stream)))
        if side is not cur:
            cur.wait_stream(side)
    return y


_side_streams = {}


def _side_stream(device):
    """One extra stream per (device, calling stream) for latency-bound work that runs underneath a bandwidth-bound kernel."""
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    with _cache_lock:
        st = _side_streams.get(key)
        if st is None:
            if len(_side_streams) > 32:
                _side_streams.clear()
            st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


def _zero_phase_taps(sos, device):
    """(taps tensor, K, E, dc gain) of the truncated zero-phase response of `sos` on `device`, cached; None when the
    response is too long for one FFT block."""
    key = (sos.tobytes(), sos.shape, str(device))
    with _cache_lock:
        zp = _zp_cache.get(key)
        if zp is None:
            if len(_zp_cache) > 16:
                _zp_cache.clear()
            # tol_edge 1e-8 (ten times below the truncation of the taps, a thousand below the parity bar): E = 532 instead of
            # 602 for the 14-30 Hz Butterworth-8, 12 % shorter row-end pieces
            r = _zero_phase_response(sos, tol_taps=1e-7, tol_edge=1e-8)
            if r is not None:
                taps, K, E = r
                r = (torch.from_numpy(np.ascontiguousarray(taps, dtype=np.float32)).to(device), K, E,
                     float(np.prod(sos[:, :3].sum(axis=1) / sos[:, 3:].sum(axis=1))) ** 2)
            zp = _zp_cache[key] = r or ()
    if not zp or zp[1] > int(lib.d4w_fir_fft_max_halfwidth()):
        return None
    return zp


_fir_ws = {}        # (taps tensor id, stream) -> [workspace, tables built]


def _fir_workspace(t, device):
    """Overlap-save workspace of a cached taps tensor per stream: after the first call it holds the taps' block spectrum
    and the next calls pass taps = NULL (include/d4w.h)."""
    key = (id(t), int(torch.cuda.current_stream(device).cuda_stream))
    with _cache_lock:
        ent = _fir_ws.get(key)
        if ent is None or ent[2] is not t:
            if len(_fir_ws) > 32:
                _fir_ws.clear()
            ent = _fir_ws[key] = [torch.empty(int(lib.d4w_xcorr_fft_ws_bytes()), dtype=torch.uint8, device=device), False, t]
    return ent


def _copy_cols(src, dst):
    """dst[:, :] = src[:, :] for float32 CUDA views [nx, n] whose samples are adjacent in a row (any row pitch): ONE
    d4w_copy_cols_f32 launch on the current stream instead of a torch slice copy (the halo bookkeeping of stream.py, the
    row-end pieces of the band-pass)."""
    nx, n = src.shape
    if tuple(dst.shape) != (nx, n) or src.dtype != torch.float32 or dst.dtype != torch.float32 \
            or (n > 1 and (src.stride(1) != 1 or dst.stride(1) != 1)):
        raise ValueError("_copy_cols: float32 views of one shape with adjacent samples in a row")
    if nx == 0 or n == 0:
        return dst
    with torch.cuda.device(src.device):
        check(lib.d4w_copy_cols_f32(src.data_ptr(), int(src.stride(0)), dev.out_ptr(dst), int(dst.stride(0)), nx, n,
                                    dev.stream_ptr(src)))
    return dst


def _concat_cols(parts):
    """torch.cat(parts, dim=1) of float32 CUDA views [nx, n_i] by one strided-copy launch per part."""
    nx = parts[0].shape[0]
    out = torch.empty((nx, sum(p.shape[1] for p in parts)), dtype=torch.float32, device=parts[0].device)
    a = 0
    for p_ in parts:
        _copy_cols(p_, out[:, a:a + p_.shape[1]])
        a += p_.shape[1]
    return out


def _sosfiltfilt_between(x, left, right, sos):
    """Zero-phase filter of rows that continue on both sides: x [nx, ns] with the samples before (left [nx, >= K], its LAST
    columns adjacent to x) and after (right [nx, >= K]) read in place by ONE overlap-save pass (d4w_fir_fft_halo_f32,
    8 B per sample, no concatenation).  Equal to filtering the concatenated record, to the 1e-7 truncation of the
    response.  Returns None when the form does not apply (response too long, halos shorter than its half width)."""
    sos = np.ascontiguousarray(np.atleast_2d(np.asarray(sos, dtype=np.float64)))
    zp = _zero_phase_taps(sos, x.device)
    if zp is None:
        return None
    t, K, E, dcg = zp
    ok = lambda h: dev.is_tensor(h) and h.is_cuda and h.dtype == torch.float32 and h.dim() == 2 and h.stride(1) == 1 \
        and h.shape[0] == x.shape[0] and h.shape[1] >= K
    if not (ok(left) and ok(right) and x.is_contiguous()):
        return None
    nx, ns = x.shape
    lv = left[:, left.shape[1] - K:]                     # views: the kernel takes a base pointer and a row pitch
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        first = torch.empty((nx, 1), dtype=torch.float32, device=x.device)
        _copy_cols(x[:, :1], first)
        ent = _fir_workspace(t, x.device)
        check(lib.d4w_fir_fft_halo_f32(dev.ptr(x), nx, ns, lv.data_ptr(), int(left.stride(0)), int(K), right.data_ptr(),
                                       int(right.stride(0)), int(K), None if ent[1] else dev.ptr(t), int(K), dev.ptr(first),
                                       dcg, dev.ptr(y), dev.ptr(ent[0]), dev.stream_ptr(x)))
        ent[1] = True
    return y


def _sosfiltfilt_one_neighbour(x, left, right, sos, padlen):
    """Zero-phase filter of rows that continue on ONE side (the first file of a record: `left` is None, its left end is a true
    record end; the last file: `right` is None): the overlap-save pass of _sosfiltfilt_between with a stand-in halo on the
    free side (the file's own columns: any finite samples do, the outputs they reach are replaced), then filtfilt's edge rule
    on that side's row-end pieces (d4w_sosfiltfilt_ends_sides_f32), whose E outer outputs overwrite the stand-in's reach.
    No concatenated copy of file + halo, no cropped copy of the result (round 5: 0.45 ms per edge file of 11 020 x 12 000).
    Returns None when the form does not apply."""
    if (left is None) == (right is None):
        return None
    sos = np.ascontiguousarray(np.atleast_2d(np.asarray(sos, dtype=np.float64)))
    zp = _zero_phase_taps(sos, x.device)
    if zp is None:
        return None
    t, K, E, dcg = zp
    P = 2 * E
    nx, ns = x.shape
    have = right if left is None else left
    if not (dev.is_tensor(have) and have.is_cuda and have.dtype == torch.float32 and have.dim() == 2 and have.stride(1) == 1
            and have.shape[0] == nx and have.shape[1] >= K and x.is_contiguous()) or ns < 4 * P or P <= padlen or E < K:
        return None
    lv = x[:, :K] if left is None else left[:, left.shape[1] - K:]
    rv = x[:, ns - K:] if right is None else right
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        first = torch.empty((nx, 1), dtype=torch.float32, device=x.device)
        _copy_cols(x[:, :1], first)
        ent = _fir_workspace(t, x.device)
        check(lib.d4w_fir_fft_halo_f32(dev.ptr(x), nx, ns, lv.data_ptr(), int(lv.stride(0)), int(K), rv.data_ptr(),
                                       int(rv.stride(0)), int(K), None if ent[1] else dev.ptr(t), int(K), dev.ptr(first),
                                       dcg, dev.out_ptr(y), dev.ptr(ent[0]), dev.stream_ptr(x)))
        ent[1] = True
        zi = _sos_host(sos)["zi"]
        ws = torch.empty(int(lib.d4w_sosfiltfilt_ends_ws_bytes(nx, P, padlen)), dtype=torch.uint8, device=x.device)
        check(lib.d4w_sosfiltfilt_ends_sides_f32(dev.ptr(x), dev.out_ptr(y), nx, ns, sos.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                 zi.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), sos.shape[0], int(padlen), int(P),
                                                 int(E), 0, 1 if left is None else 2, dev.ptr(ws), dev.stream_ptr(x)))
    return y


def _sosfiltfilt_recursive(x, sos, padlen, seg_len=None, warm=None):
    """The exact second-order-section recursion (forward + backward launch, 16 B per sample)."""
    import scipy.signal as sp
    nx, ns = x.shape
    zi = _sos_host(sos)["zi"]
    if warm is None:
        warm = -(-int(1.5 * _sos_decay_samples(sos)) // 32) * 32
    if seg_len is None:
        seg_len = _sos_segment_length(nx, ns, warm)
    if seg_len <= 0 or seg_len + 2 * warm >= ns:
        seg_len, warm = 0, 0                         # one exact segment per row
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        ws = torch.empty(int(lib.d4w_sosfiltfilt_ws_bytes(nx, ns, padlen)), dtype=torch.uint8, device=x.device)
        check(lib.d4w_sosfiltfilt_f32(dev.ptr(x), dev.ptr(y), nx, ns,
                                      sos.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                      zi.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                      sos.shape[0], int(padlen), int(seg_len), int(warm), dev.ptr(ws),
                                      dev.stream_ptr(x)))
    return y


def _rerun_check(fn, what):
    """D4W_VERIFY_RERUN=1 (opt-in): the overlap-save FFT kernels behind the zero-phase filters once returned blocks
    that depended on a kernel of ANOTHER stream resident beside them (fenced for the library's own pair, clean
    beside every foreign neighbour tried, mechanism open).  With the switch the filter runs twice and the two results are
    compared bit for bit; a difference raises instead of passing on.  Costs the second run and one host synchronisation."""
    import os
    y = fn()
    if os.environ.get("D4W_VERIFY_RERUN", "0") == "1":
        y2 = fn()
        if not torch.equal(y, y2):
            raise RuntimeError("%s: two runs on the same input differ (max %.3e): a result depended on what else was resident on the "
                               "device -- see DESIGN.md section 1" % (what, float((y - y2).abs().max())))
    return y


def _sosfiltfilt_device(x, sos, padlen, seg_len=None, warm=None):
    """x: float32 CUDA tensor [nx, ns] -> filtered tensor (new).  Long rows with a response short enough for one FFT
    block run the overlap-save form (interior) + the recursion at the row ends; everything else the recursion.
    seg_len / warm pin the recursion's segmentation (and select it)."""
    sos = np.ascontiguousarray(np.atleast_2d(np.asarray(sos, dtype=np.float64)))
    if sos.ndim != 2 or sos.shape[1] != 6:
        raise ValueError("sos array must be shape (n_sections, 6)")
    nx, ns = x.shape
    if ns <= padlen:
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % padlen)
    if seg_len is None and warm is None:
        y = _sosfiltfilt_fft(x, sos, padlen)
        if y is not None:
            return y
    return _sosfiltfilt_recursive(x, sos, padlen, seg_len, warm)


def _rows_2d(data):
    if getattr(data, "ndim", 0) == 1:
        return data[None, :], True
    if getattr(data, "ndim", 0) != 2:
        raise ValueError("expected a 1-D or 2-D [channel x time] array")
    return data, False


def sosfiltfilt(sos, x, axis=-1, padtype="odd", padlen=None):
    """scipy.signal.sosfiltfilt(sos, x, axis=1) on the GPU -- what the reference's users call on
    dsp.butterworth_filter designs (Example.py:55, DAS4Whales_ExampleNotebook.md:292).
    Only the time axis (last) and SciPy's default odd padding are supported."""
    if padtype != "odd":
        raise ValueError("only padtype='odd' (SciPy's default) is implemented")
    x2, was1d = _rows_2d(x)
    if axis not in (-1, x2.ndim - 1, 1 if not was1d else 0):
        raise ValueError("filtering runs along the time (last) axis")
    sos = np.atleast_2d(np.asarray(sos, dtype=np.float64))
    if padlen is None:                               # scipy/signal/_signaltools.py sosfiltfilt
        ntaps = 2 * sos.shape[0] + 1
        ntaps -= min((sos[:, 2] == 0).sum(), (sos[:, 5] == 0).sum())
        padlen = 3 * ntaps
    xd = dev.to_device_f32(x2)
    y = _rerun_check(lambda: _sosfiltfilt_device(xd, sos, int(padlen)), "sosfiltfilt")
    y = y[0] if was1d else y
    return dev.like_input(y, x)


def bp_filt(data, fs, fmin, fmax):
    """Zero-phase Butterworth-8 band-pass along time -- reference dsp.py:859-880.

    The reference runs scipy.signal.filtfilt on the 17-coefficient `ba` form in float64 (that
    recursion overflows in float32); here the same filter runs as float32
    second-order sections with filtfilt's edge rule (odd extension, padlen = 3*17 = 51)."""
    sos = _bp_sos(float(fs), float(fmin), float(fmax))
    x2, was1d = _rows_2d(data)
    xd = dev.to_device_f32(x2)
    y = _rerun_check(lambda: _sosfiltfilt_device(xd, sos, 51), "bp_filt")
    y = y[0] if was1d else y
    return dev.like_input(y, data)


_bp_sos_cache = {}


def _bp_sos(fs, fmin, fmax):
    """butter(8, [fmin, fmax] / (fs / 2), 'bp') as second-order sections (host, float64), designed once per band."""
    key = (fs, fmin, fmax)
    sos = _bp_sos_cache.get(key)
    if sos is None:
        import scipy.signal as sp
        if len(_bp_sos_cache) > 64:
            _bp_sos_cache.clear()
        sos = np.ascontiguousarray(sp.butter(8, [fmin / (fs / 2), fmax / (fs / 2)], "bp", output="sos"), dtype=np.float64)
        _bp_sos_cache[key] = sos
    return sos


bp_filter = bp_filt             # north-star spelling


# ---------------------------------------------------------------------------------------------
# f-k mask design (device kernels, one-off per shape)
# ---------------------------------------------------------------------------------------------
class DeviceMask:
    """An f-k mask that stays on the GPU: float32 CUDA tensor [nx, ns] on the fftshift-ed grid.

    Accepted directly by fk_filter_filt / fk_filter_sparsefilt (no host round trip).  It also
    quacks like what the reference's designers return: `.shape`, `.todense()` (sparse.COO, used by
    fk_filter_sparsefilt, dsp.py:784), `.data` (non-zero values, used by tools.disp_comprate,
    tools.py:248), and `np.asarray(mask)` (the dense ndarray fk_filter_design returns)."""

    def __init__(self, tensor):
        self._tensor = tensor
        self.shape = tuple(tensor.shape)
        self.ndim = 2
        self.dtype = np.dtype(np.float64)

    @property
    def tensor(self):
        return self._tensor

    def todense(self):
        return self.tensor.cpu().numpy().astype(np.float64)

    def __array__(self, dtype=None, copy=None):
        a = self.todense()
        return a if dtype is None else a.astype(dtype)

    @property
    def data(self):
        d = self.tensor[self.tensor != 0]
        return d.cpu().numpy().astype(np.float64)

    @property
    def nnz(self):
        return int(torch.count_nonzero(self.tensor))


class DesignedMask(DeviceMask):
    """What fk_filter_design / hybrid_filter_design / hybrid_ninf_filter_design return: the closed-form design itself.

    fk_filter_filt / fk_filter_sparsefilt write it straight into the plan's folded pass-B order
    (d4w_fk_set_mask_design_f32) -- bit-identical to folding the dense mask, which is never formed
    (9.6 GB at 20 000 x 120 000).  The dense [nx, ns] tensor is built on first use of `.tensor`,
    `np.asarray(mask)`, `.todense()`, `.data` or `.nnz`, on the device current at that moment."""

    def __init__(self, mode, trace_shape, k_spacing, t_spacing, params, i0=0, i1=0, hrow=None):
        self.mode = int(mode)
        self.shape = (int(trace_shape[0]), int(trace_shape[1]))
        self.ndim = 2
        self.dtype = np.dtype(np.float64)
        self.k_spacing, self.t_spacing = float(k_spacing), float(t_spacing)
        self.params = [float(v) for v in params] + [0.0] * (8 - len(params))
        self.i0, self.i1 = int(i0), int(i1)
        self.hrow = None if hrow is None else np.ascontiguousarray(hrow, dtype=np.float64)
        self._tensor = None
        self._hrow_dev = {}

    def hrow_on(self, device):
        """The |H|^2 row of hybrid_ninf as a float64 tensor on `device` (None for the other designs)."""
        if self.hrow is None:
            return None
        key = str(device)
        if key not in self._hrow_dev:
            self._hrow_dev[key] = torch.from_numpy(self.hrow).to(device)
        return self._hrow_dev[key]

    @property
    def tensor(self):
        if self._tensor is None:
            dev.require_gpu()
            device = torch.device("cuda:%d" % torch.cuda.current_device())
            out = torch.empty(self.shape, dtype=torch.float32, device=device)
            p8 = (ctypes.c_double * 8)(*self.params)
            h = self.hrow_on(device)
            with torch.cuda.device(device):
                check(lib.d4w_design_mask_f32(self.mode, self.shape[0], self.shape[1], self.k_spacing, self.t_spacing, p8,
                                              self.i0, self.i1, dev.ptr(h) if h is not None else None, dev.ptr(out),
                                              dev.stream_ptr(out)))
            self._tensor = out
        return self._tensor


def _shifted_axis(n, d):
    return np.fft.fftshift(np.fft.fftfreq(n, d=d))


def _design(mode, trace_shape, selected_channels, dx, fs, params, i0=0, i1=0, hrow=None, device=None):
    dev.require_gpu()
    nx, ns = int(trace_shape[0]), int(trace_shape[1])
    device = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
    out = torch.empty((nx, ns), dtype=torch.float32, device=device)
    p8 = (ctypes.c_double * 8)(*([float(v) for v in params] + [0.0] * (8 - len(params))))
    h = None
    if hrow is not None:
        h = torch.from_numpy(np.ascontiguousarray(hrow, dtype=np.float64)).to(device)
    with torch.cuda.device(device):
        check(lib.d4w_design_mask_f32(mode, nx, ns, float(selected_channels[2] * dx), 1.0 / float(fs), p8,
                                      int(i0), int(i1), dev.ptr(h) if h is not None else None, dev.ptr(out),
                                      dev.stream_ptr(out)))
        if h is not None:
            torch.cuda.current_stream().synchronize()       # h is a temporary
    return out


def _gaussian_filter(t, sigma):
    out, tmp = torch.empty_like(t), torch.empty_like(t)
    with torch.cuda.device(t.device):
        check(lib.d4w_gaussian_filter_f32(dev.ptr(t), dev.ptr(out), dev.ptr(tmp), t.shape[0], t.shape[1],
                                          float(sigma), dev.stream_ptr(t)))
    return out


def _first_index_ge(f, val):
    return int(np.argmax(f >= val))


def fk_filter_design(trace_shape, selected_channels, dx, fs, cs_min=1400, cp_min=1450, cp_max=3400, cs_max=3500):
    """Classic speed fan with sine tapers -- reference dsp.py:85-171.  Returns a DeviceMask
    (np.asarray(mask) gives the dense array; the reference returns a Fortran-ordered ndarray); the dense grid is only
    built if asked for -- the filter functions take the closed form (DesignedMask)."""
    return DesignedMask(0, trace_shape, selected_channels[2] * dx, 1.0 / float(fs), [cs_min, cp_min, cp_max, cs_max])


def hybrid_filter_design(trace_shape, selected_channels, dx, fs, cs_min=1400., cp_min=1450., fmin=15., fmax=25.,
                         display_filter=False):
    """Band-pass (4 Hz sine tapers) x speed high-pass -- reference dsp.py:174-305 (display_filter is
    a plotting path of the reference and is ignored)."""
    f = _shifted_axis(trace_shape[1], 1.0 / fs)
    i0, i1 = _first_index_ge(f, fmin - 4.0), _first_index_ge(f, fmax + 4.0)     # dsp.py:216-222
    return DesignedMask(1, trace_shape, selected_channels[2] * dx, 1.0 / float(fs), [cs_min, cp_min, fmin, fmax], i0, i1)


def hybrid_ninf_filter_design(trace_shape, selected_channels, dx, fs, cs_min=1400., cp_min=1450., cp_max=3400,
                              cs_max=3500, fmin=15., fmax=25., display_filter=False):
    """Butterworth-|H|^2 band-pass x speed band-pass, the design every reference script uses --
    reference dsp.py:308-454.  The 1-D |H|^2 row is designed on the host (SciPy, float64)."""
    import scipy.signal as sp
    ns = int(trace_shape[1])
    if ns % 2:
        raise ValueError("hybrid_ninf_filter_design needs an even number of time samples (dsp.py:349,372)")
    f = _shifted_axis(ns, 1.0 / fs)
    b, a = sp.butter(8, [fmin / (fs / 2), fmax / (fs / 2)], "bp")               # dsp.py:348
    H = np.concatenate((np.zeros(ns // 2), np.abs(sp.freqz(b, a, worN=ns // 2)[1]) ** 2))   # dsp.py:349
    i0, i1 = _first_index_ge(f, fmin - 14.0), _first_index_ge(f, fmax + 14.0)   # dsp.py:354-360
    return DesignedMask(2, trace_shape, selected_channels[2] * dx, 1.0 / float(fs), [cs_min, cp_min, cp_max, cs_max], i0, i1, H)


def hybrid_gs_filter_design(trace_shape, selected_channels, dx, fs, cs_min=1400., cp_min=1450., fmin=15., fmax=25.,
                            display_filter=False):
    """Box band x box |k| < f/cp_min, Gaussian-blurred (sigma 20) -- reference dsp.py:457-579."""
    f = _shifted_axis(trace_shape[1], 1.0 / fs)
    i0, i1 = _first_index_ge(f, fmin - 4.0), _first_index_ge(f, fmax + 4.0)     # dsp.py:503-505
    m = _design(3, trace_shape, selected_channels, dx, fs, [cs_min, cp_min, fmin, fmax], i0, i1)
    return DeviceMask(_gaussian_filter(m, 20))                                  # dsp.py:540


def hybrid_ninf_gs_filter_design(trace_shape, selected_channels, dx, fs, cs_min=1400., cp_min=1450., cp_max=3400,
                                 cs_max=3500, fmin=15., fmax=25., display_filter=False):
    """Box band x box speed band, blur THEN flips -- reference dsp.py:582-702."""
    f = _shifted_axis(trace_shape[1], 1.0 / fs)
    i0, i1 = _first_index_ge(f, fmin - 4.0), _first_index_ge(f, fmax + 4.0)     # dsp.py:628-630
    m = _design(4, trace_shape, selected_channels, dx, fs, [cs_min, cp_min, cp_max, cs_max, fmin, fmax], i0, i1)
    g = _gaussian_filter(m, 20)                                                 # dsp.py:659
    out = torch.empty_like(g)
    with torch.cuda.device(g.device):
        check(lib.d4w_flip_sum_f32(dev.ptr(g), dev.ptr(out), g.shape[0], g.shape[1], dev.stream_ptr(g)))   # dsp.py:660-661
    return DeviceMask(out)


def fk_filt(data, tint, fs, xint, dx, c_min, c_max):
    """Self-designing Gaussian-tapered speed band: design + apply -- reference dsp.py:883-953."""
    if getattr(data, "ndim", 0) != 2:
        raise ValueError("data must be a 2-D [channel x time] array")
    nx, ns = data.shape
    device = data.device if dev.is_tensor(data) and data.is_cuda else None
    # The reference designs this mask on every call (dsp.py:919-945): wedge, Gaussian blur, min-max normalisation.  It only
    # depends on the shape and on the six parameters, so the plan remembers which of them its folded mask was built from and a
    # second call with the same arguments goes straight to the filter (20 000 x 120 000: 52 -> 21 ms)
    if ns % 2 == 0:
        plan = get_fk_plan(nx, ns, device)
        key = ("fk_filt", float(tint), float(fs), float(xint), float(dx), float(c_min), float(c_max))
        x = dev.to_device_f32(data, plan.device)
        with plan.lock:
            if plan._mask_key != key:
                # axes: fftfreq(ns, tint/fs), fftfreq(nx, xint*dx) (dsp.py:923-924) -> spacing arguments
                g = _design(5, (nx, ns), [0, 0, xint], dx, fs / tint, [c_min, c_max], device=plan.device)    # dsp.py:930-936
                g = _gaussian_filter(g, 20)                                     # dsp.py:940
                plan.set_mask_normalised(g, key=key)                            # dsp.py:945, folded into the mask upload; the
                #                                                                 key is set only once the fold has been issued
            y = plan.apply(x)
        return dev.like_input(y, data)
    g = _design(5, (nx, ns), [0, 0, xint], dx, fs / tint, [c_min, c_max], device=device)
    g = _gaussian_filter(g, 20)
    with torch.cuda.device(g.device):
        check(lib.d4w_minmax_normalise_f32(dev.ptr(g), g.numel(), dev.stream_ptr(g)))        # dsp.py:945
    return _fk_apply(data, DeviceMask(g), False)


# ---------------------------------------------------------------------------------------------
# spectral views / metrics (row operators of csrc/spectral.hip)
# ---------------------------------------------------------------------------------------------
def _analytic(x2d, mode, fs=0.0, var=None):
    """x2d: float32 CUDA [nx, ns]; mode as in include/d4w.h d4w_analytic_f32."""
    nx, ns = x2d.shape
    y = torch.empty((nx, ns - 1 if mode == 3 else ns), dtype=torch.float32, device=x2d.device)
    with torch.cuda.device(x2d.device):
        if lib.d4w_analytic_row_fits_lds(ns):
            check(lib.d4w_analytic_f32(dev.ptr(x2d), dev.ptr(y), nx, ns, int(mode),
                                       dev.ptr(var) if var is not None else None, float(fs), dev.stream_ptr(x2d)))
        else:                                        # long rows: four-step time-axis transform through HBM (odd lengths as
                                                     # complex rows, lengths with a prime factor > 31 by Bluestein)
            if nx <= 65535:                          # shapes with specialised f-k kernels run their time phase + a Hilbert pass B
                _auto_specialise(nx, ns)             # (once per shape, cached on disk; the f-k filter of the block uses the same)
            for a in range(0, nx, 65535):
                xb, yb = x2d[a:a + 65535], y[a:a + 65535]
                ws = torch.empty(int(lib.d4w_analytic_long_ws_bytes(xb.shape[0], ns)), dtype=torch.uint8, device=x2d.device)
                check(lib.d4w_analytic_long_f32(dev.ptr(xb), dev.ptr(yb), xb.shape[0], ns, int(mode),
                                                dev.ptr(var[a:a + 65535]) if var is not None else None, float(fs),
                                                dev.ptr(ws), dev.stream_ptr(x2d)))
    return y


def envelope(trace):
    """|scipy.signal.hilbert(trace, axis=-1)| -- the envelope the reference forms inline at
    detect.py:192,217 and scripts/main_mfdetect.py:58."""
    x2, was1d = _rows_2d(trace)
    y = _analytic(dev.to_device_f32(x2), 0)
    return dev.like_input(y[0] if was1d else y, trace)


def hilbert_imag(trace):
    """imag(scipy.signal.hilbert(trace, axis=-1)): the Hilbert transform of every row."""
    x2, was1d = _rows_2d(trace)
    y = _analytic(dev.to_device_f32(x2), 1)
    return dev.like_input(y[0] if was1d else y, trace)


def instant_freq(channel, fs):
    """diff(unwrap(angle(hilbert(channel)))) / (2 pi) * fs -- reference dsp.py:830-856."""
    x2, was1d = _rows_2d(channel)
    y = _analytic(dev.to_device_f32(x2), 3, fs=fs)
    return dev.like_input(y[0] if was1d else y, channel)


def snr_tr_array(trace, env=False):
    """10 log10(trace^2 / std^2) per element, or with |hilbert|^2 (env=True) -- reference dsp.py:956-976."""
    if getattr(trace, "ndim", 0) != 2:
        raise ValueError("trace must be a 2-D [channel x time] array")
    x = dev.to_device_f32(trace)
    nx, ns = x.shape
    y = torch.empty_like(x)
    var = torch.empty(nx, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        if env and not lib.d4w_analytic_row_fits_lds(ns):
            check(lib.d4w_row_var_f32(dev.ptr(x), nx, ns, dev.ptr(var), dev.stream_ptr(x)))
            y = _analytic(x, 2, var=var)
        else:
            check(lib.d4w_snr_f32(dev.ptr(x), dev.ptr(y), nx, ns, int(bool(env)), dev.ptr(var), dev.stream_ptr(x)))
    return dev.like_input(y, trace)


def get_fx(trace, nfft):
    """2 |fftshift(fft(trace, nfft), axes=1)| / nfft * 1e9 -- reference dsp.py:18-38."""
    if getattr(trace, "ndim", 0) != 2:
        raise ValueError("trace must be a 2-D [channel x time] array")
    x = dev.to_device_f32(trace)
    nx, ns = x.shape
    y = torch.empty((nx, int(nfft)), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.d4w_fx_f32(dev.ptr(x), dev.ptr(y), nx, ns, int(nfft), dev.stream_ptr(x)))
    return dev.like_input(y, trace)


def _stft_mag(x2d, n_fft, hop, bin_lo, bin_hi, want_max=True):
    """|librosa.stft| of every row: returns (S [nx, bins, frames] raw magnitudes, rowmax [nx]).  want_max=False (frame
    lengths with a two-factor register transform only): the kept bins alone are formed and rowmax is None."""
    nx, ns = x2d.shape
    nt = int(lib.d4w_stft_frames(ns, int(hop)))
    S = torch.empty((nx, bin_hi - bin_lo + 1, nt), dtype=torch.float32, device=x2d.device)
    # without the row maximum: the matrix-core form (stft_mm.hip: few kept bins, hop a multiple of 8) or the two-factor
    # register transforms; every other frame length needs it
    want_max = want_max or not (int(n_fft) in (128, 160, 256, 512)
                                or lib.d4w_stft_mm_eligible(int(n_fft), int(hop), int(bin_lo), int(bin_hi)))
    mx = torch.empty(nx, dtype=torch.float32, device=x2d.device) if want_max else None
    with torch.cuda.device(x2d.device):
        check(lib.d4w_stft_mag_f32(dev.ptr(x2d), dev.ptr(S), dev.ptr(mx) if want_max else None, nx, ns, int(n_fft), int(hop),
                                   int(bin_lo), int(bin_hi), dev.stream_ptr(x2d)))
    return S, mx


def _scale_rows(S, denom, mode):
    nx = S.shape[0]
    with torch.cuda.device(S.device):
        check(lib.d4w_scale_rows_f32(dev.ptr(S), nx, S[0].numel(), dev.ptr(denom), int(mode), dev.stream_ptr(S)))
    return S


def get_spectrogram(waveform, fs, nfft=128, overlap_pct=0.8):
    """dB spectrogram of one channel normalised by its maximum; returns (p, tt, ff) -- reference
    dsp.py:41-78 (librosa.stft with hop = floor(nfft (1 - overlap_pct)))."""
    if getattr(waveform, "ndim", 0) != 1:
        raise ValueError("waveform must be 1-D")
    hop = int(np.floor(nfft * (1 - overlap_pct)))                                # dsp.py:68
    x = dev.to_device_f32(waveform.reshape(1, -1))
    S, mx = _stft_mag(x, nfft, hop, 0, nfft // 2)
    _scale_rows(S, mx, 1)                                                       # dsp.py:76
    tt = np.linspace(0, waveform.shape[0] / fs, num=S.shape[2])                 # dsp.py:74
    ff = np.linspace(0, fs / 2, num=S.shape[1])                                 # dsp.py:75
    return dev.like_input(S[0], waveform), tt, ff
