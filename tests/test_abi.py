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
    declared = int(re.search(r"^#define\s+BH_VERSION\s+(\d+)", open(HEADER).read(), re.M).group(1))
    assert _lib.lib().bh_version() == declared == _lib.BH_VERSION == _lib.header_version()


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


def test_binding_struct_layouts_mirror_the_header(tmp_path):
    """The ctypes structs are a hand-written mirror of include/bergen_hip.h: compile the header with gcc and compare the
    size and every field offset (a field added on one side only is how round 3's counters grew under an unchanged
    BH_VERSION)."""
    import subprocess
    structs = {"bh_counters": _lib.bh_counters, "bh_encoder_config": _lib.bh_encoder_config,
               "bh_encoder_counters": _lib.bh_encoder_counters}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf("{name}.{field} %zu\\n", offsetof({name}, {field}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == ctypes.sizeof(cls), name
        for field, _ in cls._fields_:
            assert int(got[f"{name}.{field}"]) == getattr(cls, field).offset, f"{name}.{field}"
        assert cls._fields_[0][0] == "struct_size" and cls().struct_size == ctypes.sizeof(cls)
    # and the header declares no struct field the binding lacks
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, cls in structs.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
        declared = re.findall(r"\b([a-z_0-9]+)\s*;", body)
        assert declared == [f for f, _ in cls._fields_], name


def test_per_handle_option_calls_validate_without_a_device():
    lib = _lib.lib()
    assert lib.bh_index_set_option(None, b"scan_kernel", 0) == _lib.BH_EINVAL
    assert lib.bh_sparse_set_option(None, b"sparse_head", 0) == _lib.BH_EINVAL
    assert lib.bh_set_option(b"scan_kernel", 1) == _lib.BH_EINVAL and b"scan_kernel must be" in lib.bh_last_error()
    assert lib.bh_set_option(b"filter256", 2) == _lib.BH_EINVAL
    assert lib.bh_set_option(b"filter256", 1) == _lib.BH_OK
