"""ctypes binding of libchq.so (the C ABI in include/chq.h).  There is no fallback: if the HIP
library is missing or cannot be loaded this module raises, and so does every operation."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHQ_LIB_PATH: load another build of the same ABI (A/B timing of kernel variants); never a fallback
LIB_PATH = os.environ.get("CHQ_LIB_PATH") or os.path.join(_HERE, "lib", "libchq.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

ARROW_DEVICE_CPU = 1
ARROW_DEVICE_ROCM = 10
ARROW_FLAG_NULLABLE = 2


class ArrowSchema(C.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
    ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchema))), ("dictionary", C.POINTER(ArrowSchema)),
    ("release", C.c_void_p), ("private_data", C.c_void_p),
]


class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
    ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(ArrowArray))),
    ("dictionary", C.POINTER(ArrowArray)), ("release", C.c_void_p), ("private_data", C.c_void_p),
]


class ArrowDeviceArray(C.Structure):
    _fields_ = [("array", ArrowArray), ("device_id", C.c_int64), ("device_type", C.c_int32),
                ("sync_event", C.c_void_p), ("reserved", C.c_int64 * 3)]


class SelectItem(C.Structure):
    _fields_ = [("kind", C.c_int), ("expr", C.c_void_p), ("alias", C.c_char_p)]


class AliasList(C.Structure):
    _fields_ = [("aliases", C.POINTER(C.c_char_p)), ("n", C.c_int)]


class TableAliases(C.Structure):
    _fields_ = [("columns", C.POINTER(AliasList)), ("n_columns", C.c_int)]


class ColumnDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("format", C.c_char_p), ("nullable", C.c_int), ("null_count", C.c_int64),
                ("offset", C.c_int64), ("validity", C.c_void_p), ("values", C.c_void_p), ("data", C.c_void_p)]


class IpcMessage(C.Structure):
    pass


IpcMessage._fields_ = [("header", C.c_void_p), ("header_len", C.c_int64), ("body", C.c_void_p), ("body_len", C.c_int64),
                       ("body_device_type", C.c_int32), ("body_device_id", C.c_int32), ("end_of_stream", C.c_uint8 * 8),
                       ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ParquetImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_int64), ("release", C.c_void_p), ("private_data", C.c_void_p)]


# chq_read_range_fn: int (*)(void* user, int64_t offset, int64_t length, uint8_t* dst)
READ_RANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)


class CallStats(C.Structure):
    _fields_ = [("rows_in", C.c_int64), ("rows_out", C.c_int64), ("tiles", C.c_int64), ("launches", C.c_int64),
                ("bytes_read_alg", C.c_int64), ("bytes_written_alg", C.c_int64), ("kernel_ns", C.c_int64)]


# every symbol include/chq.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "chq_abi_version", "chq_status_name", "chq_ctx_create", "chq_ctx_destroy", "chq_ctx_last_error", "chq_ctx_stream",
    "chq_ctx_set_option", "chq_ctx_last_stats", "chq_expr_identifier", "chq_expr_compound_identifier", "chq_expr_number",
    "chq_expr_boolean", "chq_expr_single_quoted_string", "chq_expr_unsupported_value", "chq_expr_binary_op",
    "chq_expr_nested", "chq_expr_unsupported", "chq_expr_free", "chq_filter_record", "chq_filter_records", "chq_filter_records_coalesced", "chq_plan_describe", "chq_project_record",
    "chq_compute_value", "chq_filter_project_record", "chq_record_to_device", "chq_record_to_host", "chq_wrap_columns",
    "chq_record_copy_to_peer", "chq_record_to_ipc", "chq_record_from_ipc", "chq_ipc_describe",
    "chq_parquet_open", "chq_parquet_open_reader", "chq_parquet_num_columns", "chq_parquet_column_name", "chq_parquet_read_columns",
    "chq_parquet_close", "chq_parquet_num_row_groups", "chq_parquet_row_group_num_rows",
    "chq_parquet_describe", "chq_parquet_read_row_group", "chq_parquet_read_row_groups", "chq_record_to_parquet", "chq_records_to_parquet",
]


def build(force: bool = False) -> str:
    """Compile the HIP kernels and host code for gfx950 with the committed Makefile (hipcc cross-compiles
    without a GPU)."""
    args = ["make", "-s", "-C", CSRC_DIR, "-j4"]
    if force:
        subprocess.run(["make", "-s", "-C", CSRC_DIR, "clean"], check=True)
    subprocess.run(args, check=True)
    return LIB_PATH


_lib = None


def _preload_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; a process that loads both that
    copy and /opt/rocm's ends up with two HSA runtimes and the second one sees no GPU.  When torch is installed,
    bind to the runtime it ships (without importing torch) so both share one, whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(d, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C chapterhouseqe_amd/csrc). There is no CPU fallback.")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i64, cp, ci = C.c_void_p, C.c_int64, C.c_char_p, C.c_int
    PDA, PS = C.POINTER(ArrowDeviceArray), C.POINTER(ArrowSchema)
    PTA = C.POINTER(TableAliases)
    sig = {
        "chq_abi_version": (ci, []), "chq_status_name": (cp, [ci]),
        "chq_ctx_create": (ci, [ci, vp, C.POINTER(vp)]), "chq_ctx_destroy": (None, [vp]),
        "chq_ctx_last_error": (cp, [vp]), "chq_ctx_stream": (vp, [vp]), "chq_ctx_set_option": (ci, [vp, cp, i64]),
        "chq_ctx_last_stats": (None, [vp, C.POINTER(CallStats)]),
        "chq_expr_identifier": (vp, [cp]), "chq_expr_compound_identifier": (vp, [C.POINTER(cp), ci]),
        "chq_expr_number": (vp, [cp, ci]), "chq_expr_boolean": (vp, [ci]), "chq_expr_single_quoted_string": (vp, [cp, i64]),
        "chq_expr_unsupported_value": (vp, [cp]), "chq_expr_binary_op": (vp, [vp, ci, cp, vp]), "chq_expr_nested": (vp, [vp]),
        "chq_expr_unsupported": (vp, [cp]), "chq_expr_free": (None, [vp]),
        "chq_filter_record": (ci, [vp, PDA, PS, PTA, vp, ci, PDA, PS]),
        "chq_filter_records": (ci, [vp, ci, C.POINTER(PDA), PS, PTA, vp, ci, PDA, PS]),
        "chq_plan_describe": (ci, [PS, PTA, vp, i64, ci, C.c_char_p, C.c_size_t]),
        "chq_filter_records_coalesced": (ci, [vp, ci, C.POINTER(PDA), PS, PTA, vp, ci, PDA, PS, C.POINTER(i64)]),
        "chq_project_record": (ci, [vp, C.POINTER(SelectItem), ci, PDA, PS, PTA, ci, PDA, PS]),
        "chq_compute_value": (ci, [vp, PDA, PS, PTA, vp, ci, PDA, PS, C.POINTER(ci)]),
        "chq_filter_project_record": (ci, [vp, vp, C.POINTER(SelectItem), ci, PDA, PS, PTA, ci, PDA, PS]),
        "chq_record_to_device": (ci, [vp, PDA, PS, PDA, PS]), "chq_record_to_host": (ci, [vp, PDA, PS, PDA, PS]),
        "chq_wrap_columns": (ci, [vp, C.POINTER(ColumnDesc), ci, i64, ci, PDA, PS]),
        "chq_record_copy_to_peer": (ci, [vp, vp, PDA, PS, PDA, PS]),
        "chq_record_to_ipc": (ci, [vp, PDA, PS, ci, C.POINTER(IpcMessage)]),
        "chq_record_from_ipc": (ci, [vp, vp, i64, vp, i64, ci, ci, PDA, PS]),
        "chq_ipc_describe": (ci, [vp, i64, C.c_char_p, C.c_size_t]),
        "chq_parquet_open": (ci, [vp, i64, C.POINTER(vp), C.c_char_p, C.c_size_t]),
        "chq_parquet_open_reader": (ci, [i64, READ_RANGE_FN, vp, C.POINTER(vp), C.c_char_p, C.c_size_t]),
        "chq_parquet_num_columns": (C.c_int32, [vp]),
        "chq_parquet_column_name": (C.c_char_p, [vp, C.c_int32]),
        "chq_parquet_read_columns": (ci, [vp, vp, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, ci, vp, vp]),
        "chq_parquet_close": (None, [vp]),
        "chq_parquet_num_row_groups": (C.c_int32, [vp]),
        "chq_parquet_row_group_num_rows": (i64, [vp, C.c_int32]),
        "chq_parquet_describe": (ci, [vp, C.c_char_p, C.c_size_t]),
        "chq_parquet_read_row_group": (ci, [vp, vp, C.c_int32, ci, PDA, PS]),
        "chq_parquet_read_row_groups": (ci, [vp, vp, C.c_int32, C.c_int32, ci, vp, vp]),
        "chq_record_to_parquet": (ci, [vp, PDA, PS, C.POINTER(ParquetImage)]),
        "chq_records_to_parquet": (ci, [vp, ci, C.POINTER(PDA), PS, C.POINTER(ParquetImage)]),
    }
    for name, (res, args) in sig.items():
        try:
            f = getattr(L, name)
        except AttributeError:
            if os.environ.get("CHQ_LIB_PATH"):   # an older build under A/B timing: entry points added since are simply absent
                continue
            raise
        f.restype, f.argtypes = res, args
    _lib = L
    return L
