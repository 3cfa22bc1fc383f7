"""stream -- consecutive files of one cable processed as one continuous record (SURVEY.md 8f row f4,
BASELINE configs[4]).  No reference counterpart: the reference's scripts process one 60-s file at a
time (scripts/main_mfdetect.py:112-123), so every file starts and ends with filter transients and
the last L-1 lags of every correlogram are cut short.

What is carried across file boundaries
  * zero-phase band-pass (dsp.bp_filt): file i is filtered together with `halo` samples of file i-1
    and of file i+1.  The filter's two-sided impulse response decays below 1e-9 of its peak within
    ~650 samples, so with halo >= 1024 the kept part equals dsp.bp_filt of the
    CONCATENATED record to float32 rounding (parity target: the reference run on the concatenation);
  * matched filter (detect.compute_cross_correlogram): the lags ns-L+1 .. ns-1 of file i are
    completed with the first L-1 filtered samples of file i+1; normalisation stays per file (row mean
    and max|.| of file i, detect.py:157), as in the reference's per-file runs.
The f-k filter stays per file: it is a circular 2-D transform of one file in the reference
(dsp.py:748), and a transform over the concatenation would be a different filter.

Latency: the band-passed file i is complete when file i+1 has arrived, its correlograms when the
band-passed and f-k filtered file i+1 exists, i.e. when file i+2 has arrived.  push() returns the
results that became final; flush() closes the stream (the last file ends like a stand-alone file).
All arithmetic runs in the HIP library through dsp / detect; this module only moves halos around.
"""
import os

import numpy as np
import torch

from . import _device as dev
from . import detect, dsp


class FileStream:
    def __init__(self, fs, fmin, fmax, templates=(), fk_mask=None, halo=1024, prev_tail=None, on_filtered=None):
        """fs, fmin, fmax: band-pass (dsp.bp_filt arguments); templates: full-length or support-only
        template vectors (detect.gen_template_fincall output) for the matched filter; fk_mask: f-k mask
        for one file's shape (any form dsp.fk_filter_filt accepts) or None to skip the f-k filter.
        prev_tail: the last `halo` raw samples of the file BEFORE the first one pushed (a stream that continues a
        record another process holds, e.g. the previous rank's last file); on_filtered(index, y): called as soon as a
        file's band-passed (+ f-k filtered) version exists (the neighbour that needs its head can be served early)."""
        self.fs, self.fmin, self.fmax = float(fs), float(fmin), float(fmax)
        self.halo = int(halo)
        self.taps = [detect._normalised_support(t) for t in templates]
        # DC tail of a zero-padded full-length template (detect.py:158), applied per file exactly as
        # detect.compute_cross_correlograms does (0 for support-only vectors)
        self.tail = [detect._tail_coef(t) for t in templates]
        self.lmax = max((len(t) for t in self.taps), default=1)
        self.fk_mask = fk_mask
        self._raw = []          # raw files waiting for their right halo: [(index, tensor)]
        self._prev_tail = dev.to_device_f32(prev_tail) if prev_tail is not None else None   # last `halo` raw samples of the file before self._raw[0]
        self._on_filtered = on_filtered
        self._filt = []         # filtered (band-pass [+ f-k]) files waiting for the next file's head
        self._n = 0
        self._sos = None

    # ------------------------------------------------------------------------------------------
    def _bandpass(self, left, cur, right):
        """dsp.bp_filt of [left | cur | right], the part belonging to `cur`."""
        import scipy.signal as sp
        if self._sos is None:
            self._sos = sp.butter(8, [self.fmin / (self.fs / 2), self.fmax / (self.fs / 2)], "bp", output="sos")
        if left is not None and right is not None:
            # a file between two others: one overlap-save pass that reads the halos in place (no concatenated copy, no
            # cropped copy; the halo is longer than the filter's truncated response)
            y = dsp._sosfiltfilt_between(cur if cur.is_contiguous() else cur.contiguous(), left, right, self._sos)
            if y is not None:
                return y
        if (left is None) != (right is None) and os.environ.get("D4W_STREAM_EDGE_FIR", "1") != "0":
            # the first / last file of a record: the same pass with a stand-in halo on the free side, filtfilt's edge rule
            # on that side's row ends afterwards (no concatenated copy, no cropped copy)
            y = dsp._sosfiltfilt_one_neighbour(cur if cur.is_contiguous() else cur.contiguous(), left, right, self._sos, 51)
            if y is not None:
                return y
        # a response longer than the halo (or a stand-alone file): the pieces side by side in one buffer (one strided-copy
        # launch per piece, no torch.cat), filtered with filtfilt's edge rule at the true record ends, the file's own
        # columns copied out
        parts = [p for p in (left, cur, right) if p is not None]
        ext = dsp._concat_cols(parts) if len(parts) > 1 else (cur if cur.is_contiguous() else cur.contiguous())
        y = dsp._sosfiltfilt_device(ext, self._sos, 51)
        if len(parts) == 1:
            return y
        a = left.shape[1] if left is not None else 0
        return dsp._copy_cols(y[:, a:a + cur.shape[1]], torch.empty_like(cur))

    def _finish_bandpass(self, right_head):
        """The oldest waiting raw file now has its right halo (or the stream ends): filter it."""
        idx, cur = self._raw.pop(0)
        y = self._bandpass(self._prev_tail, cur, right_head)
        # the tail is COPIED (one strided-copy launch, nx x halo floats): a view would keep the whole raw file alive -- and
        # require the caller to leave it unmodified -- for one more push, just to hold 1024 columns of it
        self._prev_tail = dsp._copy_cols(cur[:, -self.halo:], torch.empty((cur.shape[0], self.halo), dtype=cur.dtype, device=cur.device)) \
            if self.halo > 0 else None
        stats = None
        if self.fk_mask is not None:
            # the row means / maxima the matched filter normalises by come out of the f-k filter's last pass
            # (and, for zero-padded templates, the prefix maxima their DC-tail term is decided on)
            y, stats = dsp._fk_apply_stats(y, self.fk_mask, prefix=self._want_prefix(y.shape[1])) if self.taps \
                else (dsp.fk_filter_filt(y, self.fk_mask), None)
        if self._on_filtered is not None:
            self._on_filtered(idx, y)
        return idx, y, stats

    def _tails_in_kernel(self, ns):
        """Round 6: the zero-padded templates' DC tail is added inside the matrix-core correlator where that applies
        (detect._tails_in_kernel) -- no prefix maxima, no second pass."""
        return bool(self.taps) and detect._tails_in_kernel(self.taps, self.tail, ns, detect._xcorr_method(self.taps, ns, "auto"))

    def _want_prefix(self, ns):
        return any(c != 0.0 for c in self.tail) and not self._tails_in_kernel(ns)

    def _correlate(self, idx, y, next_head, stats=None):
        """Correlograms of filtered file `y`, its last lags completed with `next_head` (or cut short); stats = (row means,
        row maxima) of y when the f-k filter left them."""
        out = {"index": idx, "filtered": y}
        if not self.taps:
            return out
        nx, ns = y.shape
        from ._lib import lib, check
        pm = None
        if stats is not None:
            mean, mx = stats[:2]
            pm = stats[2] if len(stats) > 2 else None
        else:
            with torch.cuda.device(y.device):
                mean = torch.empty(nx, dtype=torch.float64, device=y.device)
                mx = torch.empty(nx, dtype=torch.float32, device=y.device)
                if self._want_prefix(ns):
                    pm = torch.empty(nx, dtype=torch.float32, device=y.device)
                    check(lib.d4w_row_stats_prefix_f32(dev.ptr(y), nx, ns, dev.ptr(mean), dev.ptr(mx), dev.ptr(pm), dev.stream_ptr(y)))
                else:
                    check(lib.d4w_row_stats_f32(dev.ptr(y), nx, ns, dev.ptr(mean), dev.ptr(mx), dev.stream_ptr(y)))
        if next_head is not None and self.lmax > 1 and next_head.is_cuda and next_head.dtype == torch.float32 \
                and next_head.stride(1) == 1 and detect.xcorr_continuation_ok(self.taps, ns):
            # rows continue into the next file: the kernel reads the head of the next file's rows in place (de-meaned
            # like the file's own samples) -- no concatenated copy, no cropped copies of the correlograms
            rmax = []
            inker = self._tails_in_kernel(ns)
            cs = detect._xcorr_device(y, self.taps, normalize=True, stats=(mean, mx), cont=(next_head, self.lmax - 1),
                                      row_max=rmax, tails=self.tail if inker else None)
            if len(rmax) == len(cs):
                out["row_max"] = rmax        # max over the lags of every row, per template (detect.correlogram_max)
            if inker:
                out["correlograms"] = cs
                return out
        else:
            if next_head is not None and self.lmax > 1:
                # the padding must enter de-meaned like the file's own samples (the kernel subtracts the mean from
                # every sample it reads)
                ext = dsp._concat_cols([y, next_head[:, :self.lmax - 1]])
            else:
                ext = y
            # (a file without a successor -- the record's last -- still gets its row maxima from the correlator's epilogue:
            # the DC-tail decision per row and correlogram_max need no sweep of the correlograms then)
            rmax = [] if ext is y else None
            inker = ext is y and self._tails_in_kernel(ns)
            cs = detect._xcorr_device(ext, self.taps, normalize=True, stats=(mean, mx), row_max=rmax, tails=self.tail if inker else None)
            if rmax and len(rmax) == len(cs):
                out["row_max"] = rmax
            if inker:
                out["correlograms"] = cs
                return out
            cs = [dsp._copy_cols(c[:, :ns], torch.empty_like(y)) if c.shape[1] != ns else c for c in cs]
        # the DC tail of zero-padded templates, decided per row on the data (detect._apply_tails: the band-passed rows of
        # a stream have prefix sums of a few samples' size and are left alone; the row maxima stay valid either way)
        detect._apply_tails(y, (mean, mx), cs, self.taps, self.tail, out.get("row_max"), pmax=pm)
        out["correlograms"] = cs
        return out

    # ------------------------------------------------------------------------------------------
    def push(self, block):
        """Next file ([channel x time], NumPy or CUDA tensor).  Returns the list of results that became
        final (dicts with "index", "filtered" and "correlograms"), possibly empty.

        Lifetime of a pushed buffer: a float32 CUDA tensor is NOT copied -- file i is read in place by the work that
        push(i + 1) (or flush()) enqueues, and by nothing later (its last `halo` columns are copied out then).  A caller
        that recycles its input buffers may therefore overwrite file i's buffer, in stream order on the stream push() ran
        on, once push(i + 1) has returned: two alternating buffers are enough.  (NumPy blocks and tensors of another
        dtype are converted into a tensor the stream owns; the caller's array is free as soon as push() returns.)"""
        x = dev.to_device_f32(block)
        self._raw.append((self._n, x))
        self._n += 1
        done = []
        if len(self._raw) >= 2:                               # the older raw file has its right halo now
            head = x[:, :self.halo] if self.halo > 0 else None        # a view of the new file: no copy
            self._filt.append(self._finish_bandpass(head))
        if len(self._filt) >= 2:                              # the older filtered file has its successor
            idx, y, stats = self._filt.pop(0)
            done.append(self._correlate(idx, y, self._filt[0][1], stats))
        return done

    def flush(self, next_head=None, next_filtered_head=None):
        """End of the stream: finish the files still waiting.  By default their right edge is a true record end;
        when the record continues elsewhere, next_head = the first `halo` raw samples of the following file and
        next_filtered_head = the first lmax - 1 filtered samples of it (a tensor, or a callable returning one -- e.g.
        the wait on a receive posted earlier)."""
        done = []
        head = dev.to_device_f32(next_head)[:, :self.halo] if next_head is not None and self.halo > 0 else None
        while self._raw:
            self._filt.append(self._finish_bandpass(head if len(self._raw) == 1 else self._raw[1][1][:, :self.halo]))
            while len(self._filt) >= 2:
                idx, y, stats = self._filt.pop(0)
                done.append(self._correlate(idx, y, self._filt[0][1], stats))
        while self._filt:
            idx, y, stats = self._filt.pop(0)
            nxt = self._filt[0][1] if self._filt else None
            if nxt is None and next_filtered_head is not None:
                nxt = next_filtered_head() if callable(next_filtered_head) else next_filtered_head
            done.append(self._correlate(idx, y, nxt, stats))
        return done
