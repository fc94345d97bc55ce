"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/snmf.h declares, and the host mirror reproduces the reference's error behaviour.
No compute call is made (there is no GPU here and no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "snmf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(snmf_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from se_snmf_nat_amd import _lib
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_builds_loads_and_exports_every_declared_symbol(lib):
    for s in header_symbols():
        assert hasattr(lib, s), s
    from se_snmf_nat_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "snmf.h")).read()
    ver = int(re.search(r"#define SNMF_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.snmf_abi_version() == ver == _lib.ABI_VERSION == 5  # header, library and binding agree (the binding refuses another)
    assert lib.snmf_device_count() >= 0


def test_shared_object_is_a_gfx950_code_object():
    from se_snmf_nat_amd import _lib
    _lib.build()
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_hstep" in blob and b"k_wstats" in blob


def test_no_device_fails_loudly_without_fallback(lib):
    if lib.snmf_device_count() > 0:
        pytest.skip("a GPU is present")
    from se_snmf_nat_amd import SnmfError, sparse_nmf
    h = C.c_void_p()
    assert lib.snmf_ctx_create(C.byref(h), 0) == 5  # SNMF_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.snmf_last_error()
    with pytest.raises(SnmfError, match="NO_DEVICE"):
        sparse_nmf(np.ones((4, 4)), dict(r=2, cost_check=1))


def test_null_arguments_are_rejected(lib):
    assert lib.snmf_ctx_create(None, 0) == 1
    assert lib.snmf_plan_init(None) == 1
    assert lib.snmf_plan_run(None, 1, None) == 1
    assert lib.snmf_plan_stats_len(None) == 0


def test_host_mirror_reproduces_reference_errors():
    """Errors raised before any device work, as src/sparse_nmf.m does."""
    from se_snmf_nat_amd import SnmfError, sparse_nmf
    V = np.ones((6, 5))
    with pytest.raises(SnmfError, match="Number of components or initialization must be given"):
        sparse_nmf(V, dict(cost_check=1))  # src/sparse_nmf.m:117-119
    with pytest.raises(SnmfError, match="cost_check"):
        sparse_nmf(V, dict(r=2))  # src/sparse_nmf.m:260
    with pytest.raises(SnmfError, match="init_h"):
        sparse_nmf(V, dict(init_w=np.ones((6, 2)), init_h=np.ones((3, 5)), cost_check=1))
    with pytest.raises(SnmfError, match="init_w"):
        sparse_nmf(V, dict(init_w=np.ones((7, 2)), cost_check=1))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "se_snmf_nat_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), fn
