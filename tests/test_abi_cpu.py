"""CPU checks of the drop-in boundary: libwoft_hip.so builds for gfx950, loads, and exports
every symbol include/woft_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from woft_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def test_header_symbols_exported(lib):
    header = (ROOT / "include" / "woft_hip.h").read_text()
    declared = set(re.findall(r"^\s*(?:int|int64_t)\s+(woft_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 18
    raw = ctypes.CDLL(str(ROOT / "woft_amd" / "lib" / "libwoft_hip.so"))
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in woft_hip.h but not exported"
    from woft_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def test_struct_layout_and_version(lib):
    from woft_amd import _lib
    assert lib.woft_abi_version() >= 100
    assert lib.woft_sizeof(0) == ctypes.sizeof(_lib.ConvParams)
    assert lib.woft_sizeof(1) == ctypes.sizeof(_lib.LookupParams)


def test_argument_validation_without_gpu(lib):
    """Entry points reject bad arguments before touching the device."""
    from woft_amd import _lib
    p = _lib.ConvParams()
    assert lib.woft_conv2d(ctypes.byref(p), None) == -1
    assert lib.woft_conv2d(None, None) == -1
    lp = _lib.LookupParams()
    assert lib.woft_corr_lookup(ctypes.byref(lp), None) == -1
    assert lib.woft_hfit(None, None, None, 10, None, 0, 1.0, 0, None, None, None, None) == -1
    assert lib.woft_inorm_apply(None, None, None, None, None, None, 0, None, 0, 0, 0, None) == -1
    assert lib.woft_inorm_finalize(None, None, 0, 0, 0, 0, 0, 1e-5, None, None, None, None) == -1


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under woft_amd/ or pytracking/ may import it."""
    for pkg in ("woft_amd", "pytracking"):
        for f in (ROOT / pkg).rglob("*.py"):
            src = f.read_text()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
