"""Builds and loads the CPU *emulator* build of the HIP sources (tests/emu/hip_emu.h).

Test infrastructure: lets the kernel logic be exercised in the GPU-less build container through
the very same C ABI (host pointers instead of device pointers).  Never used by the product."""
import ctypes
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "_build", "libd4w_emu.so")


def _emu_fresh(deps):
    return os.path.exists(EMU_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(EMU_LIB) for d in deps)


def build_emu():
    srcs = sorted(glob.glob(os.path.join(ROOT, "das4whales_amd", "csrc", "*.hip")))
    deps = srcs + glob.glob(os.path.join(ROOT, "das4whales_amd", "csrc", "*.h")) + \
        [os.path.join(EMU_DIR, "hip_emu.h"), os.path.join(ROOT, "include", "d4w.h")]
    if _emu_fresh(deps):
        return EMU_LIB
    os.makedirs(os.path.dirname(EMU_LIB), exist_ok=True)
    # several test processes (pytest-xdist workers, the gloo ranks) may get here at once: one builds, the others wait
    import fcntl
    with open(os.path.join(EMU_DIR, "_build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if _emu_fresh(deps):
                return EMU_LIB
            objs = []
            procs = []
            for s in srcs:
                o = os.path.join(EMU_DIR, "_build", os.path.basename(s) + ".o")
                objs.append(o)
                procs.append(subprocess.Popen(["g++", "-O1", "-std=c++17", "-fPIC", "-DD4W_EMU", "-include",
                                               os.path.join(EMU_DIR, "hip_emu.h"), "-I", os.path.join(ROOT, "include"),
                                               "-x", "c++", "-c", s, "-o", o]))
            for p in procs:
                if p.wait() != 0:
                    raise RuntimeError("emulator build failed")
            tmp = EMU_LIB + ".%d.tmp" % os.getpid()
            subprocess.check_call(["g++", "-shared", "-o", tmp] + objs)
            os.replace(tmp, EMU_LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return EMU_LIB


def load_emu():
    lib = ctypes.CDLL(build_emu())
    lib.d4w_last_error.restype = ctypes.c_char_p
    return lib


def vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)
