"""CPU: the C-ABI shared library loads and exports every symbol include/bergen_hip.h declares.
No compute calls are made (there is no GPU here, and the library has no CPU fallback)."""
import ctypes
import os
import re

import pytest

from bergen_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bergen_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bh_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_list_the_same_symbols():
    assert declared_functions() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(handle, name), f"{name} missing from {_lib.LIB_PATH}"
    assert _lib.lib().bh_version() == 130


def test_library_contains_gfx950_code_object():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"bh_scan_topk_kernel" in blob and b"bh_merge_rescore_kernel" in blob


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.lib()
    assert lib.bh_device_count() == 0
    assert lib.bh_init(0) == _lib.BH_EHIP
    assert b"no HIP device" in lib.bh_last_error()
    with pytest.raises(_lib.BergenHipError):
        _lib.init(0)
    from bergen_amd import FlatIndex
    with pytest.raises(_lib.BergenHipError):
        FlatIndex(10, 64)


def test_argument_validation_needs_no_device():
    lib = _lib.lib()
    assert lib.bh_set_option(b"query_tile", 77) == _lib.BH_EINVAL
    assert lib.bh_set_option(b"nope", 1) == _lib.BH_EINVAL
    assert lib.bh_set_option(b"query_tile", 128) == _lib.BH_OK
    assert lib.bh_merge_topk(None, None, 0, 1, 1, None, None) == _lib.BH_EINVAL
    with pytest.raises(ValueError):
        _lib.check(lib.bh_index_finalize(None))


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "bergen_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "liboracle" not in text, f
