"""CPU-only: the C-ABI library loads and exports every symbol include/nanocaller_hip.h declares; it fails
loudly (no CPU fallback) when no GPU is visible; the product package never imports the oracle."""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "nanocaller_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    from nanocaller_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 20
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "missing export %s" % n
    assert sorted(_lib.EXPORTS) == names
    assert L.nc_abi_version() == _lib.ABI_VERSION == 11


def test_header_compiles_as_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "nanocaller_hip.h"\nint main(void){nc_tile_entry e; e.start=1; return sizeof(e)==16 && sizeof(nc_readpack)>0 ? 0 : 1;}\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_no_gpu_fails_loudly_not_silently():
    import torch
    from nanocaller_amd import _lib
    L = _lib.lib()
    n = ctypes.c_int(-1)
    rc = L.nc_device_count(ctypes.byref(n))
    if torch.cuda.is_available():
        assert rc == 0 and n.value >= 1
        return
    ctx = ctypes.c_void_p()
    assert L.nc_ctx_create(0, ctypes.byref(ctx)) < 0 and not ctx.value
    import pytest
    from nanocaller_amd.engine import Engine
    with pytest.raises(_lib.NanoCallerHipError):
        Engine(0)


def test_product_never_imports_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "nanocaller_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "libnc_oracle" not in txt and "nc_oracle" not in txt, f
    code = "import sys; sys.path.insert(0, %r); import nanocaller_amd.snpCaller, nanocaller_amd.model_architect; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)" % ROOT
    subprocess.run([sys.executable, "-c", code], check=True)


def test_graft_entry_build_checks_pass():
    """__graft_entry__.build() (the driver's does-it-build check): compiles, loads and resolves every export"""
    import __graft_entry__ as g
    g.build()
