"""
ctypes loader of libbergen_hip.so (C ABI: include/bergen_hip.h).

BASELINE.json asks for a cffi layer; cffi is not installed in this image, ctypes
is the stdlib equivalent and binds the same `extern "C"` symbols.

There is deliberately NO fallback: if the shared library is missing or a symbol is absent the
import fails loudly, and every compute entry point fails if no gfx950 device is present.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BERGEN_HIP_LIB", os.path.join(_HERE, "lib", "libbergen_hip.so"))

BH_VERSION = 143  # the header version the struct layouts below mirror (tests/test_abi.py compares it with include/bergen_hip.h)

BH_OK = 0
BH_EINVAL = -1
BH_EHIP = -2
BH_ENOMEM = -3
BH_EINCOMPLETE = -4
BH_EUNSUPPORTED = -5

BH_F16 = 0
BH_F32 = 1
BH_METRIC_IP = 0
BH_METRIC_COS = 1


class _Sized(ctypes.Structure):
    """Structs whose first field is `struct_size` (include/bergen_hip.h, BH_VERSION 142): set on construction."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(type(self))


class bh_counters(_Sized):
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("reserved_head", ctypes.c_int32),
        ("n_rows", ctypes.c_int64),
        ("dim", ctypes.c_int32),
        ("dim_padded", ctypes.c_int32),
        ("query_tile", ctypes.c_int32),
        ("n_passes", ctypes.c_int32),
        ("n_workgroups", ctypes.c_int32),
        ("k_padded", ctypes.c_int32),
        ("scan_ms", ctypes.c_double),
        ("merge_ms", ctypes.c_double),
        ("total_ms", ctypes.c_double),
        ("algorithmic_bytes", ctypes.c_double),
        ("shader_mhz", ctypes.c_double),
        ("uncertified_queries", ctypes.c_int64),
        ("exact_ms", ctypes.c_double),
        ("tail_scan_ms", ctypes.c_double),
        ("tail_query_tile", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
        ("exact_passes", ctypes.c_int64),
        ("exact_rows_rescored", ctypes.c_int64),
        ("paired_scan_ms", ctypes.c_double),
        ("paired_launches", ctypes.c_int32),
        ("reserved1", ctypes.c_int32),
        ("balanced_scan_ms", ctypes.c_double),   # since BH_VERSION 142
        ("balanced_queries", ctypes.c_int32),
        ("reserved2", ctypes.c_int32),
    ]


class bh_encoder_config(_Sized):
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("n_layers", ctypes.c_int32),
        ("hidden", ctypes.c_int32),
        ("n_heads", ctypes.c_int32),
        ("intermediate", ctypes.c_int32),
        ("vocab_size", ctypes.c_int32),
        ("max_position", ctypes.c_int32),
        ("type_vocab_size", ctypes.c_int32),
        ("activation", ctypes.c_int32),
        ("ln_eps", ctypes.c_float),
        ("head_dim", ctypes.c_int32),
        ("position_offset", ctypes.c_int32),
        ("rotary_theta", ctypes.c_float),
        ("ffn_gated", ctypes.c_int32),
        ("rotary_scale", ctypes.c_float),   # since BH_VERSION 143
        ("alibi", ctypes.c_int32),
    ]


class bh_encoder_counters(_Sized):
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("batch", ctypes.c_int32),
        ("seq_len", ctypes.c_int32),
        ("real_tokens", ctypes.c_int64),
        ("packed_rows", ctypes.c_int64),
        ("forward_ms", ctypes.c_double),
        ("flops", ctypes.c_double),
        ("ln_fused", ctypes.c_int32),   # since BH_VERSION 142
        ("reserved0", ctypes.c_int32),
    ]


# name -> (restype, argtypes); must list EVERY symbol include/bergen_hip.h declares
_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
SYMBOLS = {
    "bh_init": (ctypes.c_int, [ctypes.c_int]),
    "bh_version": (ctypes.c_int, []),
    "bh_last_error": (ctypes.c_char_p, []),
    "bh_device_count": (ctypes.c_int, []),
    "bh_index_create": (ctypes.c_int, [ctypes.POINTER(_vp), _i64, _i32, _i32, _i32]),
    "bh_index_upload": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32]),
    "bh_index_upload_device": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32]),
    "bh_index_finalize": (ctypes.c_int, [_vp]),
    "bh_index_rows_uploaded": (_i64, [_vp]),
    "bh_index_destroy": (None, [_vp]),
    "bh_search": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "bh_search_device": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "bh_merge_topk": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "bh_merge_topk_device": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "bh_bench_counters": (ctypes.c_int, [_vp, ctypes.POINTER(bh_counters)]),
    "bh_debug_scan_timeline": (_i64, [_vp, _vp, _i64]),
    "bh_set_option": (ctypes.c_int, [ctypes.c_char_p, _i64]),
    "bh_index_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, _i64]),
    "bh_sparse_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, _i64]),
    "bh_encoder_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(bh_encoder_config)]),
    "bh_encoder_set_tensor": (ctypes.c_int, [_vp, ctypes.c_char_p, _vp, _i32, _i64]),
    "bh_encoder_commit": (ctypes.c_int, [_vp]),
    "bh_encoder_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, _i64]),
    "bh_encoder_set_rel_index": (ctypes.c_int, [_vp, _vp, _i32]),
    "bh_encoder_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32]),
    "bh_encoder_counters_get": (ctypes.c_int, [_vp, ctypes.POINTER(bh_encoder_counters)]),
    "bh_encoder_destroy": (None, [_vp]),
    "bh_op_gemm_f16": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _i32,
                                      _i32, _i32, ctypes.POINTER(ctypes.c_float)]),
    "bh_op_attention": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32]),
    "bh_op_layernorm": (ctypes.c_int, [_vp, _vp, _i64, _i32, ctypes.c_float, _vp, _vp]),
    "bh_op_rotary": (ctypes.c_int, [_vp, _i64, _i32, _vp, ctypes.c_float, _i32]),
    "bh_op_swiglu": (ctypes.c_int, [_vp, _vp, _i64, _i32]),
    "bh_op_gated_act": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32]),
    "bh_gemm_permlane_mode": (ctypes.c_int, []),
    "bh_sparse_create": (ctypes.c_int, [ctypes.POINTER(_vp), _i64, _i32]),
    "bh_sparse_upload_csr": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i32]),
    "bh_sparse_finalize": (ctypes.c_int, [_vp]),
    "bh_sparse_rows_uploaded": (_i64, [_vp]),
    "bh_sparse_nnz": (_i64, [_vp]),
    "bh_sparse_search": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "bh_sparse_counters": (ctypes.c_int, [_vp, ctypes.POINTER(bh_counters)]),
    "bh_sparse_destroy": (None, [_vp]),
}

_lib = None


class BergenHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libbergen_hip error {code}: {message}")
        self.code = code
        self.message = message


def lib():
    """Load (once) and return the ctypes handle with all prototypes set."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C bergen_amd/csrc`.  bergen_amd has no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        # the struct layouts above are those of ONE header version: refuse a library built from another
        if handle.bh_version() != BH_VERSION:
            raise ImportError(f"{LIB_PATH} reports ABI version {handle.bh_version()}, this binding is written for {BH_VERSION}: rebuild "
                              f"(`make -C bergen_amd/csrc`)")
        _lib = handle
    return _lib


def check(rc):
    """Map a bh_status to the reference's exception types (Errors)."""
    if rc == BH_OK:
        return
    msg = lib().bh_last_error().decode("utf-8", "replace")
    if rc == BH_EINCOMPLETE:
        raise IOError(msg)  # reference: modules/retrieve.py:165-166
    if rc == BH_EINVAL:
        raise ValueError(msg)
    if rc == BH_ENOMEM:
        raise MemoryError(msg)
    raise BergenHipError(rc, msg)


_initialised = {}


def init(device_id=0):
    """bh_init once per (process, device)."""
    if not _initialised.get(device_id):
        check(lib().bh_init(device_id))
        _initialised[device_id] = True
    else:
        check(lib().bh_init(device_id))  # cheap: re-selects the device for this thread


BH_OPTION_INHERIT = -(1 << 63)


def header_version():
    """BH_VERSION of include/bergen_hip.h as shipped beside this package (None when the header is not there)."""
    import re
    path = os.path.join(os.path.dirname(_HERE), "include", "bergen_hip.h")
    if not os.path.exists(path):
        return None
    m = re.search(r"^#define\s+BH_VERSION\s+(\d+)", open(path).read(), re.M)
    return int(m.group(1)) if m else None


def set_option(name, value):
    """PROCESS-WIDE default of a tuning option (every handle without an override of its own sees it from its next search)."""
    check(lib().bh_set_option(name.encode(), int(value)))
