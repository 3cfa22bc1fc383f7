"""Shape-specialised f-k kernels compiled on demand.

The pass kernels of csrc/fk_fast.h are templates over the radix split of both axes (every sub-transform a register
butterfly, all index arithmetic constant-folded); libd4w.so carries the instantiations for the benchmark shape and the
60-s file shapes, and any other shape runs the generic runtime-radix kernels (3-8x slower).  This module is the code
generator for the rest: it picks a configuration (C1, C2A, C2B | N1, NA, NB, NC) for any [nx, ns] whose axes factor into
parts <= 32, instantiates the same templates in a one-file translation unit, compiles it with hipcc for gfx950
(~40 s, cached as das4whales_amd/lib/jit/*.so keyed by the configuration and a hash of the kernel headers) and
registers the resulting table entry with the library (d4w_fk_register_shape).  Host-side planning only -- the
arithmetic is the HIP kernels'.

    dw.dsp.compile_fk_shape(8000, 12000)      # explicitly, once per shape; later plans for it use the specialised kernels
    D4W_FK_JIT=0                               # switches off the automatic compilation for new large shapes (>= 2^24 samples)
"""
import ctypes
import glob
import hashlib
import math
import os
import subprocess
import threading

from ._lib import lib

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_INC = os.path.join(os.path.dirname(_HERE), "include")
_JITDIR = os.path.join(_HERE, "lib", "jit")
_lock = threading.Lock()
_loaded = {}            # path -> CDLL (kept alive: the kernels live in these libraries)
_REG = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t)


def _splits(n, parts, lim=32):
    """All ordered factorisations of n into `parts` factors <= lim."""
    if parts == 1:
        return [(n,)] if 1 <= n <= lim else []
    out = []
    for f in range(1, min(n, lim) + 1):
        if n % f == 0:
            out += [(f,) + r for r in _splits(n // f, parts - 1, lim)]
    return out


def _round64(v):
    return -(-v // 64) * 64


def choose_config(nx, ns):
    """(C1, C2A, C2B, N1, NA, NB, NC, TA, TC, thrA, thrC, thrB, wgA, wgC, wgB) or None.  Heuristics from the measured
    shapes: C1 and N1 around 20 (pass A's tile = C1 x N1 x 16 columns should hold ~10^4 elements),
    N2 = NA NB NC between ~500 and ~2500 samples (pass B's row pair in <= 48 KiB of LDS), 128-byte strips."""
    if nx < 1 or ns < 2 or ns % 2:
        return None
    M = ns // 2
    # a channel count with prime factors > 31 (e.g. 13223 = 7 x 1889, scripts/main_mfdetect.py's selection): that part of
    # the c2 axis runs the generic Bluestein pass C (C2X), passes A and B are specialised around it
    c2x, rest = 1, nx
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31):
        while rest % p == 0:
            rest //= p
    if rest > 1:
        if rest > 4096 or nx // rest > 32:
            return None
        c2x = rest
    best = None
    for N1, NA, NB, NC in _splits(M, 4):
        N2 = NA * NB * NC
        if N2 % 2 or NB * NC > 512 or NA < NB or NB < NC or N2 > 4096:
            continue
        TA = next((t for t in (16, 8, 4, 2) if N2 % t == 0 and M % t == 0), 0)
        if not TA:
            continue
        for C1, C2A, C2B in ([(nx // c2x, 1, 1)] if c2x > 1 else _splits(nx, 3)):
            if C2A > C2B or C1 * N1 * TA > 16000 or C2A * (C2B + 1) * TA > 16000:
                continue
            tileA = C1 * N1 * TA
            score = (0.5 * abs(math.log(tileA / 8000.0)) + abs(math.log(N2 / 1600.0)) + 0.3 * abs(math.log(C1 / 20.0))
                     + 0.2 * (C2B - C2A) / max(C2B, 1) + 0.1 * (NA - NC) / max(NA, 1) + (0.0 if TA == 16 else 1.0))
            if best is None or score < best[0]:
                best = (score, (C1, C2A, C2B, N1, NA, NB, NC, TA))
    if best is None:
        return None
    C1, C2A, C2B, N1, NA, NB, NC, TA = best[1]
    thrA = _round64(max(N1, C1) * TA)
    thrC = _round64(max(C2A, C2B) * TA)
    thrB = _round64(max(2 * NB * NC, NA * NB))
    if max(thrA, thrC, thrB) > 1024:
        return None
    ldsA = (C1 * N1 * TA + 2 * N1 * TA) * 8
    ldsC = (C2A * (C2B + 1) * TA + C2A * C2B) * 8
    ldsB = (2 * (NA * NB * NC + NA * NB) + 2 * NB * NC) * 8
    # resident workgroups per CU the persistent grids are sized for: what LDS and the 2048 threads of a CU admit, at most 8;
    # the large tiles of the measured shapes run 1-2 (their register pipelines hide the latency), small tiles need more
    wg = lambda l, thr: max(1, min(2 if l > 40 * 1024 else 8, (150 * 1024) // max(l, 1), 2048 // thr))
    return (C1, C2A, C2B, N1, NA, NB, NC, TA, TA, thrA, thrC, thrB, wg(ldsA, thrA), wg(ldsC, thrC), wg(ldsB, thrB), c2x)


def _header_hash():
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(_CSRC, "*.h"))) + [os.path.join(_INC, "d4w.h")]:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def _lib_path(nx, ns, cfg):
    tag = "_".join(str(v) for v in cfg[:12]) + ("_x%d" % cfg[15] if cfg[15] > 1 else "")
    return os.path.join(_JITDIR, "fk_%dx%d_%s_%s.so" % (nx, ns, tag, _header_hash()))


def _register(path):
    if path in _loaded:
        return
    jl = ctypes.CDLL(path)
    jl.d4w_jit_register.restype = ctypes.c_int
    jl.d4w_jit_register.argtypes = [ctypes.c_void_p]
    reg = ctypes.cast(lib.d4w_fk_register_shape, ctypes.c_void_p)
    rc = jl.d4w_jit_register(reg)
    if rc != 0:
        raise RuntimeError("d4w: registering %s failed: %s" % (path, lib.d4w_last_error().decode()))
    _loaded[path] = jl


def is_specialised(nx, ns):
    return bool(lib.d4w_fk_shape_is_specialised(int(nx), int(ns)))


_failed = {}      # (nx, ns, header hash) -> reason: a build that failed is not attempted again in this process


def failure_reason(nx, ns):
    """Why compile_fk_shape(nx, ns) returned False in this process (None: it did not, or has not been asked)."""
    return _failed.get((int(nx), int(ns), _header_hash()))


def compile_fk_shape(nx, ns, verbose=False, warn=False):
    """Make sure [nx, ns] runs shape-specialised f-k kernels; returns True when it does (built in, cached or freshly
    compiled), False when the shape has no admissible configuration or no compiler is available (generic kernels).
    A failure is remembered per process (and as a .failed marker next to the cached objects, keyed by the kernel headers'
    hash) so that the ~40 s compile is not repeated on every call; warn=True raises a RuntimeWarning when the shape falls
    back to the generic kernels."""
    nx, ns = int(nx), int(ns)

    def failed(reason, marker=None):
        _failed[(nx, ns, _header_hash())] = reason
        if marker:
            try:
                with open(marker, "w") as f:
                    f.write(reason[-4000:])
            except OSError:
                pass
        if warn:
            import warnings
            warnings.warn("das4whales_amd: no shape-specialised f-k kernels for %d x %d (%s); the generic kernels run "
                          "3-8x slower at this size" % (nx, ns, reason.splitlines()[0][:200]), RuntimeWarning, stacklevel=3)
        return False

    with _lock:
        if is_specialised(nx, ns):
            return True
        prev = _failed.get((nx, ns, _header_hash()))
        if prev is not None:
            return failed(prev)
        cfg = choose_config(nx, ns)
        if cfg is None:
            return failed("no admissible configuration: an axis has a prime factor > 31 beyond what the Bluestein pass carries")
        path = _lib_path(nx, ns, cfg)
        marker = path[:-3] + ".failed"
        if not os.path.exists(path):
            if os.path.exists(marker) and os.environ.get("D4W_FK_JIT", "") == "retry":
                try:
                    os.remove(marker)                 # the user asked for another attempt
                except OSError:
                    pass
            if os.path.exists(marker):
                hint = " (marker %s: delete it or set D4W_FK_JIT=retry to build again)" % marker
                try:
                    return failed("an earlier build of this configuration failed%s: %s" % (hint, open(marker).read()))
                except OSError:
                    return failed("an earlier build of this configuration failed" + hint)
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            if not os.path.exists(hipcc):
                return failed("no hipcc at %s" % hipcc)
            os.makedirs(_JITDIR, exist_ok=True)
            src = "%s.%d.hip" % (path[:-3], os.getpid())
            with open(src, "w") as f:
                f.write('// generated by das4whales_amd/fkjit.py for %d x %d\n#include "fk_entry.h"\n'
                        "using G = d4w::FkFastCfg<%s, false, 1, %d>;\n"
                        'extern "C" int d4w_jit_register(int (*reg)(const void*, size_t)) {\n'
                        "    d4w::FkFastEntry e = d4w::fast_entry<G>(%d, %d, %d);\n"
                        "    return reg(&e, sizeof(e));\n}\n" % (nx, ns, ", ".join(str(v) for v in cfg[:12]), cfg[15], cfg[12], cfg[13], cfg[14]))
            tmp = "%s.%d.tmp" % (path, os.getpid())          # several ranks may compile the same shape at once
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                   "-I", _CSRC, "-I", _INC, src, "-o", tmp]
            if verbose:
                print("[fkjit]", " ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            for leftover in ((src, tmp) if r.returncode != 0 else (src,)):
                try:
                    os.remove(leftover)
                except OSError:
                    pass
            if r.returncode != 0:
                if verbose:
                    print(r.stderr[-2000:], flush=True)
                # Only a compiler diagnostic is remembered on disk.  A compile that was killed (negative return code = signal,
                # e.g. the OOM killer), ran out of disk or was interrupted says nothing about the configuration: the next
                # process tries again.
                transient = (r.returncode < 0 or "No space left" in r.stderr or "error:" not in r.stderr)
                return failed("hipcc failed: " + (r.stderr.strip() or "exit code %d" % r.returncode), None if transient else marker)
            os.replace(tmp, path)
        _register(path)
        if not is_specialised(nx, ns):
            return failed("the compiled configuration did not register")
        return True


def prune_stale():
    """Delete cached objects built from other versions of the kernel headers (they are never loaded again)."""
    tag = "_" + _header_hash() + ".so"
    n = 0
    for path in glob.glob(os.path.join(_JITDIR, "fk_*")):
        if not path.endswith(tag) and not path.endswith(tag[:-3] + ".failed") and ".so." not in os.path.basename(path) and not path.endswith(".hip"):
            try:
                os.remove(path)
                n += 1
            except OSError:
                pass
    return n


def load_cached():
    """Register every cached configuration built from the current kernel headers (called at import: a dlopen each)."""
    tag = "_" + _header_hash() + ".so"
    for path in sorted(glob.glob(os.path.join(_JITDIR, "fk_*.so"))):
        if path.endswith(tag):
            try:
                with _lock:
                    _register(path)
            except Exception:
                pass
