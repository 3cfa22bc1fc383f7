"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/d4w.h declares, the product never reaches into oracle/, and a missing library fails
loudly.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "das4whales_amd", "lib", "libd4w.so")


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "d4w.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(d4w_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        ge.build()
    return LIB


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built)
    syms = declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), "libd4w.so does not export %s declared in include/d4w.h" % s
    lib.d4w_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.d4w_version()


def test_python_binding_covers_header(built):
    from das4whales_amd import _lib
    missing = [s for s in declared_symbols() if s not in _lib.SIGNATURES]
    assert not missing, "ctypes binding lacks %s" % missing


def test_argument_validation_without_gpu(built):
    """Pure host-side argument checks return D4W_EINVAL before any device work."""
    from das4whales_amd import _lib
    h = ctypes.c_void_p()
    assert _lib.lib.d4w_fk_plan_create(40, 481, ctypes.byref(h)) == -1     # odd ns
    assert b"even" in _lib.lib.d4w_last_error()
    assert _lib.lib.d4w_fk_plan_create(0, 480, ctypes.byref(h)) == -1
    with pytest.raises(ValueError):
        _lib.check(-1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "das4whales_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.replace("restated-oracle", ""), "%s mentions oracle/" % fn
                assert "/root/reference" not in txt


def test_missing_library_fails_loudly(tmp_path):
    code = ("import sys, os; sys.path.insert(0, %r); import das4whales_amd._lib as L" % ROOT)
    env = dict(os.environ)
    # run with the library hidden: copy package sans lib/
    import shutil
    dst = tmp_path / "das4whales_amd"
    shutil.copytree(os.path.join(ROOT, "das4whales_amd"), dst, ignore=shutil.ignore_patterns("lib", "__pycache__"))
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import das4whales_amd" % str(tmp_path)],
                       capture_output=True, text=True, env=env)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr or "not found" in r.stderr


def test_library_override_is_announced(built, tmp_path):
    """D4W_LIB swaps the shared object under the whole package (probe builds): it must say so -- a variable left
    over from a probe session would otherwise make every result come from that build unnoticed."""
    import shutil
    other = tmp_path / "libd4w_probe.so"
    shutil.copy(built, other)
    code = "import sys, warnings; sys.path.insert(0, %r); import das4whales_amd._lib as L; print(L.LIB_PATH)" % ROOT
    env = dict(os.environ, D4W_LIB=str(other))
    r = subprocess.run([sys.executable, "-W", "always", "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert str(other) in r.stdout and "D4W_LIB" in r.stderr and "replaces the packaged library" in r.stderr
    env.pop("D4W_LIB")
    r = subprocess.run([sys.executable, "-W", "always", "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "D4W_LIB" not in r.stderr
