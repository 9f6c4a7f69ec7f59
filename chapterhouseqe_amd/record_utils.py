"""Host-side mirror of the reference's `record_utils` module on top of the chq C ABI.

Reference (RU = src/handlers/operator_handler/operators/record_utils):

    pub fn compute_value(rec, table_aliases, expr) -> Result<ArrayDatum>        RU/compute_value.rs:57-61
    pub fn filter_record(rec, table_aliases, expr) -> Result<RecordBatch>       RU/filter_record.rs:21-25
    pub fn project_record(fields, record, table_aliases) -> Result<RecordBatch> RU/record_projection.rs:16-20

Same names, same argument order and meaning, same error classes (ChqError.code mirrors the reference's
error enums, see include/chq.h).  Records are pyarrow RecordBatches (host; staged to HBM by the library) or
`DeviceRecordBatch`es (already in HBM; results stay there).  Every call runs HIP kernels on the GPU --
there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple, Union

import pyarrow as pa

from . import _lib as L
from . import sqlast as A

_OPS = {A.BinaryOperator.And: 0, A.BinaryOperator.Or: 1, A.BinaryOperator.Plus: 2, A.BinaryOperator.Minus: 3,
        A.BinaryOperator.Multiply: 4, A.BinaryOperator.Divide: 5, A.BinaryOperator.Modulo: 6,
        A.BinaryOperator.Eq: 7, A.BinaryOperator.NotEq: 8, A.BinaryOperator.Gt: 9, A.BinaryOperator.GtEq: 10,
        A.BinaryOperator.Lt: 11, A.BinaryOperator.LtEq: 12}
_OP_OTHER = 13


class ChqError(Exception):
    """An error returned by the library; `.code` is the chq_status (mirrors the reference's error enums)."""

    def __init__(self, code: int, message: str):
        self.code = code
        self.status_name = L.lib().chq_status_name(code).decode()
        self.message = message
        super().__init__(f"[{code} {self.status_name}] {message}")


# ------------------------------------------------------------------------------------------ context
class Context:
    """One per operator instance (the reference processes one batch at a time per instance)."""

    def __init__(self, device_id: int = 0, stream: Optional[int] = None):
        """`stream`: a hipStream_t handle to launch on (e.g. torch.cuda.current_stream().cuda_stream); 0 means the
        legacy default stream (passed to the C ABI as hipStreamLegacy); None lets the context create its own
        non-blocking stream (work issued on other streams must then be synchronised by the caller)."""
        self._h = C.c_void_p()
        if stream is not None and stream == 0:
            stream = 1   # hipStreamLegacy
        rc = L.lib().chq_ctx_create(device_id, C.c_void_p(stream) if stream else None, C.byref(self._h))
        if rc:
            raise ChqError(rc, f"chq_ctx_create(device {device_id}) failed: no usable MI355X/HIP device")
        self.device_id = device_id

    @property
    def handle(self):
        return self._h

    def set_option(self, key: str, value: int) -> None:
        rc = L.lib().chq_ctx_set_option(self._h, key.encode(), int(value))
        if rc:
            raise ChqError(rc, self.last_error())

    def last_error(self) -> str:
        return L.lib().chq_ctx_last_error(self._h).decode(errors="replace")

    def last_stats(self) -> Dict[str, int]:
        s = L.CallStats()
        L.lib().chq_ctx_last_stats(self._h, C.byref(s))
        return {k: getattr(s, k) for k, _ in L.CallStats._fields_}

    @property
    def stream(self) -> int:
        return L.lib().chq_ctx_stream(self._h) or 0

    def close(self) -> None:
        if self._h:
            L.lib().chq_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ------------------------------------------------------------------------------------------ batches
class _CBatch:
    """ArrowDeviceArray + ArrowSchema pair owned by Python (released on close)."""

    def __init__(self):
        self.array = L.ArrowDeviceArray()
        self.schema = L.ArrowSchema()

    def release(self):
        for s in (self.array.array, self.schema):
            if s.release:
                C.CFUNCTYPE(None, C.c_void_p)(s.release)(C.addressof(s))

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _export_host(rec: pa.RecordBatch) -> _CBatch:
    cb = _CBatch()
    rec._export_to_c(C.addressof(cb.array.array), C.addressof(cb.schema))
    cb.array.device_id = -1
    cb.array.device_type = L.ARROW_DEVICE_CPU
    return cb


def _import_host(cb: _CBatch) -> pa.RecordBatch:
    rb = pa.RecordBatch._import_from_c(C.addressof(cb.array.array), C.addressof(cb.schema))  # moves ownership
    return rb


class DeviceRecordBatch:
    """A record batch resident in HBM (Arrow C Device Data Interface, ARROW_DEVICE_ROCM)."""

    def __init__(self, ctx: Context, cb: _CBatch, keepalive=None):
        self.ctx = ctx
        self._cb = cb
        self._keep = keepalive

    @property
    def num_rows(self) -> int:
        return self._cb.array.array.length

    @property
    def num_columns(self) -> int:
        return self._cb.array.array.n_children

    @property
    def column_names(self) -> List[str]:
        return [self._cb.schema.children[i].contents.name.decode() for i in range(self.num_columns)]

    @property
    def column_formats(self) -> List[str]:
        """Arrow C format string of every column ("i", "f", "u", ...)."""
        return [self._cb.schema.children[i].contents.format.decode() for i in range(self.num_columns)]

    def describe_columns(self) -> List[dict]:
        """name / format / nullable / null_count / offset / buffer addresses of every column -- what
        `from_device_buffers` takes, e.g. to hand the buffers to RCCL without a copy."""
        out = []
        for i in range(self.num_columns):
            sch = self._cb.schema.children[i].contents
            arr = self._cb.array.array.children[i].contents
            bufs = [arr.buffers[k] or 0 for k in range(arr.n_buffers)]
            out.append({"name": sch.name.decode(), "format": sch.format.decode(), "nullable": bool(sch.flags & 2),
                        "null_count": int(arr.null_count), "offset": int(arr.offset), "length": int(arr.length),
                        "validity": bufs[0] if len(bufs) > 0 else 0, "values": bufs[1] if len(bufs) > 1 else 0,
                        "data": bufs[2] if len(bufs) > 2 else 0})
        return out

    def slice(self, offset: int, length: Optional[int] = None) -> "DeviceRecordBatch":
        """Rows [offset, offset + length) as a zero-copy view (Arrow slice: the same buffers, a larger element offset); the
        view keeps this batch alive.  Null counts of nullable columns become unknown (-1), as for any Arrow slice."""
        n = self.num_rows
        offset = max(0, min(int(offset), n))
        length = n - offset if length is None else max(0, min(int(length), n - offset))
        cols = self.describe_columns()
        for c in cols:
            c["offset"] += offset
            if c["validity"] and c["null_count"] != 0:
                c["null_count"] = -1
        return DeviceRecordBatch.from_device_buffers(cols, length, ctx=self.ctx, keepalive=[self])

    @staticmethod
    def from_device_buffers(columns: Sequence[dict], num_rows: int, ctx: Optional[Context] = None,
                            keepalive=None) -> "DeviceRecordBatch":
        """Wrap caller-owned HBM buffers without copying; `columns` as returned by `describe_columns`
        (validity / data optional)."""
        ctx = ctx or default_context()
        descs = (L.ColumnDesc * max(1, len(columns)))()
        for i, col in enumerate(columns):
            descs[i].name = col["name"].encode()
            descs[i].format = col["format"].encode()
            descs[i].nullable = 1 if col.get("nullable") else 0
            descs[i].null_count = col.get("null_count", 0)
            descs[i].offset = col.get("offset", 0)
            descs[i].validity = col.get("validity") or None
            descs[i].values = col.get("values") or None
            descs[i].data = col.get("data") or None
        out = _CBatch()
        rc = L.lib().chq_wrap_columns(ctx.handle, descs, len(columns), num_rows, L.ARROW_DEVICE_ROCM, C.byref(out.array), C.byref(out.schema))
        if rc:
            raise ChqError(rc, ctx.last_error())
        return DeviceRecordBatch(ctx, out, keepalive)

    def column_buffer_address(self, i: int, buffer: int = 1) -> int:
        """Device address of buffer `buffer` (0 validity, 1 values/offsets, 2 data) of column i."""
        return self._cb.array.array.children[i].contents.buffers[buffer] or 0

    @staticmethod
    def from_host(rec: pa.RecordBatch, ctx: Optional[Context] = None) -> "DeviceRecordBatch":
        ctx = ctx or default_context()
        src = _export_host(rec)
        out = _CBatch()
        rc = L.lib().chq_record_to_device(ctx.handle, C.byref(src.array), C.byref(src.schema), C.byref(out.array), C.byref(out.schema))
        src.release()
        if rc:
            raise ChqError(rc, ctx.last_error())
        return DeviceRecordBatch(ctx, out)

    @staticmethod
    def from_device_pointers(columns: Sequence[Tuple[str, str, int]], num_rows: int, ctx: Optional[Context] = None,
                             keepalive=None) -> "DeviceRecordBatch":
        """Wrap caller-owned HBM buffers (e.g. torch tensors) without copying.
        `columns`: (name, arrow_format, device_address_of_values[, device_address_of_utf8_bytes]) for non-null
        columns; for Utf8 ("u") the values buffer holds the int32 offsets."""
        ctx = ctx or default_context()
        descs = (L.ColumnDesc * max(1, len(columns)))()
        for i, col in enumerate(columns):
            name, fmt, addr = col[0], col[1], col[2]
            descs[i].name = name.encode()
            descs[i].format = fmt.encode()
            descs[i].values = addr
            if len(col) > 3:
                descs[i].data = col[3]
        out = _CBatch()
        rc = L.lib().chq_wrap_columns(ctx.handle, descs, len(columns), num_rows, L.ARROW_DEVICE_ROCM, C.byref(out.array), C.byref(out.schema))
        if rc:
            raise ChqError(rc, ctx.last_error())
        return DeviceRecordBatch(ctx, out, keepalive)

    def to_host(self) -> pa.RecordBatch:
        out = _CBatch()
        rc = L.lib().chq_record_to_host(self.ctx.handle, C.byref(self._cb.array), C.byref(self._cb.schema), C.byref(out.array), C.byref(out.schema))
        if rc:
            raise ChqError(rc, self.ctx.last_error())
        return _import_host(out)

    def copy_to_peer(self, dst_ctx: Context) -> "DeviceRecordBatch":
        """`chq_record_copy_to_peer`: this batch in the HBM of `dst_ctx`'s GPU (hipMemcpyPeerAsync over the xGMI link of
        the pair, asynchronous: the result carries a sync_event every chq call waits on).  Keep `self` alive until the
        result has been used once."""
        out = _CBatch()
        rc = L.lib().chq_record_copy_to_peer(self.ctx.handle, dst_ctx.handle, C.byref(self._cb.array), C.byref(self._cb.schema),
                                             C.byref(out.array), C.byref(out.schema))
        if rc:
            raise ChqError(rc, dst_ctx.last_error())
        return DeviceRecordBatch(dst_ctx, out, keepalive=self)

    # ---- zero-copy torch views of the buffers in HBM (tests, benches, handing a column to RCCL) -------------------------
    _TORCH_TYPES = {"c": ("int8", 1), "C": ("uint8", 1), "s": ("int16", 2), "i": ("int32", 4), "l": ("int64", 8),
                    "f": ("float32", 4), "g": ("float64", 8), "e": ("float16", 2)}

    def _hbm_view(self, torch, address: int, nbytes: int):
        dev = torch.device("cuda", self.ctx.device_id)
        if nbytes <= 0 or not address:
            return torch.empty(0, dtype=torch.uint8, device=dev)

        class _Range:   # __cuda_array_interface__ over a raw HBM range; keeps the batch (and so its buffers) alive
            def __init__(self, owner):
                self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (address, False), "version": 2}
                self._owner = owner
        return torch.as_tensor(_Range(self), device=dev)

    def column_tensor(self, i: int, torch):
        """The values of fixed-width column i (Arrow slice offset applied) as a torch tensor over the same HBM bytes."""
        arr = self._cb.array.array.children[i].contents
        fmt = self._cb.schema.children[i].contents.format.decode()
        name, width = self._TORCH_TYPES[fmt]
        raw = self._hbm_view(torch, (arr.buffers[1] or 0) + arr.offset * width, arr.length * width)
        return raw.view(getattr(torch, name))

    def utf8_tensors(self, i: int, torch):
        """(offsets, data) of Utf8 column i: int32 offsets [length + 1] (slice offset applied, values still absolute) and
        the uint8 data buffer from byte 0 up to the last offset."""
        arr = self._cb.array.array.children[i].contents
        offs = self._hbm_view(torch, (arr.buffers[1] or 0) + arr.offset * 4, (arr.length + 1) * 4).view(torch.int32)
        end = int(offs[-1].item()) if arr.length >= 0 and offs.numel() else 0
        return offs, self._hbm_view(torch, arr.buffers[2] or 0, end)

    def release(self) -> None:
        self._cb.release()


Record = Union[pa.RecordBatch, DeviceRecordBatch]


# ------------------------------------------------------------------------------------------ expr marshalling
def _expr_to_c(e: A.Expr):
    lib = L.lib()
    if isinstance(e, A.Nested):
        return lib.chq_expr_nested(_expr_to_c(e.expr))
    if isinstance(e, A.BinaryOp):
        return lib.chq_expr_binary_op(_expr_to_c(e.left), _OPS.get(e.op, _OP_OTHER), e.op.value.encode(), _expr_to_c(e.right))
    if isinstance(e, A.ValueExpr):
        v = e.value
        if isinstance(v, A.Number):
            return lib.chq_expr_number(v.text.encode(), int(v.long))
        if isinstance(v, A.Boolean):
            return lib.chq_expr_boolean(int(v.value))
        if isinstance(v, A.SingleQuotedString):
            b = v.value.encode()
            return lib.chq_expr_single_quoted_string(b, len(b))
        return lib.chq_expr_unsupported_value(v.debug.encode())
    if isinstance(e, A.Identifier):
        return lib.chq_expr_identifier(e.ident.value.encode())
    if isinstance(e, A.CompoundIdentifier):
        arr = (C.c_char_p * max(1, len(e.idents)))(*[i.value.encode() for i in e.idents])
        return lib.chq_expr_compound_identifier(arr, len(e.idents))
    if isinstance(e, A.UnsupportedExpr):
        return lib.chq_expr_unsupported(e.debug.encode())
    raise TypeError(f"not an Expr: {e!r}")


class _Aliases:
    def __init__(self, table_aliases: Optional[Sequence[Sequence[str]]]):
        self.ptr = None
        if table_aliases is None:
            return
        self._lists = (L.AliasList * max(1, len(table_aliases)))()
        self._keep = []
        for i, al in enumerate(table_aliases):
            arr = (C.c_char_p * max(1, len(al)))(*[a.encode() for a in al])
            self._keep.append(arr)
            self._lists[i].aliases = arr
            self._lists[i].n = len(al)
        self._ta = L.TableAliases(self._lists, len(table_aliases))
        self.ptr = C.pointer(self._ta)


def _items_to_c(fields: Sequence[A.SelectItem]):
    lib = L.lib()
    items = (L.SelectItem * max(1, len(fields)))()
    exprs = []
    for i, f in enumerate(fields):
        if isinstance(f, A.Wildcard):
            items[i].kind = 0
        elif isinstance(f, A.QualifiedWildcard):
            items[i].kind = 1
        elif isinstance(f, A.UnnamedExpr):
            items[i].kind = 2
            items[i].expr = _expr_to_c(f.expr)
            exprs.append(items[i].expr)
        elif isinstance(f, A.ExprWithAlias):
            items[i].kind = 3
            items[i].expr = _expr_to_c(f.expr)
            items[i].alias = f.alias.value.encode()
            exprs.append(items[i].expr)
        else:
            raise TypeError(f"not a SelectItem: {f!r}")
    return items, exprs


def _prepare(rec: Record, ctx: Optional[Context]):
    if isinstance(rec, DeviceRecordBatch):
        return rec.ctx if ctx is None else ctx, rec._cb, False, True
    return (ctx or default_context()), _export_host(rec), True, False


def _finish(ctx: Context, rc: int, out: _CBatch, device_result: bool):
    if rc:
        raise ChqError(rc, ctx.last_error())
    if device_result:
        return DeviceRecordBatch(ctx, out)
    return _import_host(out)


# ------------------------------------------------------------------------------------------ the path
def filter_record(rec: Record, table_aliases: Optional[Sequence[Sequence[str]]], expr: A.Expr, *,
                  ctx: Optional[Context] = None, device_result: Optional[bool] = None):
    """RU/filter_record.rs:21-39.  Keeps the rows where `expr` is true (and valid); every column, original
    order, same schema.  Result residency follows the input unless `device_result` says otherwise."""
    ctx, src, own_src, on_dev = _prepare(rec, ctx)
    dev_out = on_dev if device_result is None else device_result
    e = _expr_to_c(expr)
    al = _Aliases(table_aliases)
    out = _CBatch()
    try:
        rc = L.lib().chq_filter_record(ctx.handle, C.byref(src.array), C.byref(src.schema), al.ptr, e,
                                       L.ARROW_DEVICE_ROCM if dev_out else L.ARROW_DEVICE_CPU, C.byref(out.array), C.byref(out.schema))
    finally:
        L.lib().chq_expr_free(e)
        if own_src:
            src.release()
    return _finish(ctx, rc, out, dev_out)


def plan_describe(schema: pa.Schema, table_aliases: Optional[Sequence[Sequence[str]]], expr: A.Expr, num_rows: int = 2,
                  enable_minus: bool = False) -> str:
    """Host half of compute_value (typing, coercion, folding, lowering) as text -- `chq_plan_describe`; needs no GPU.
    Raises ChqError with the static status the record calls would return."""
    cs = L.ArrowSchema()
    schema._export_to_c(C.addressof(cs))
    e = _expr_to_c(expr)
    al = _Aliases(table_aliases)
    buf = C.create_string_buffer(8192)
    try:
        rc = L.lib().chq_plan_describe(C.byref(cs), al.ptr, e, num_rows, 1 if enable_minus else 0, buf, len(buf))
    finally:
        L.lib().chq_expr_free(e)
        if cs.release:
            C.CFUNCTYPE(None, C.c_void_p)(cs.release)(C.addressof(cs))
    text = buf.value.decode(errors="replace")
    if rc:
        raise ChqError(rc, text)
    return text


class IpcEncoded:
    """`chq_record_to_ipc`: an Arrow IPC stream in three parts -- `header` (host bytes: Schema message + the RecordBatch
    message's metadata), the body (`body_address`, `body_len`; in HBM unless `body_on_device` is False) and the 8-byte
    end-of-stream marker.  `to_bytes()` is the complete stream (host bodies only)."""

    def __init__(self, ctx: Context, msg: "L.IpcMessage"):
        self.ctx = ctx
        self._msg = msg
        self.header = C.string_at(msg.header, msg.header_len)
        self.body_address = msg.body or 0
        self.body_len = msg.body_len
        self.body_on_device = msg.body_device_type == L.ARROW_DEVICE_ROCM
        self.end_of_stream = bytes(msg.end_of_stream)

    def to_bytes(self) -> bytes:
        if self.body_on_device:
            raise ValueError("the body lives in HBM: ask for body_on_device=False, or move it with RCCL / a peer copy")
        return self.header + C.string_at(self.body_address, self.body_len) + self.end_of_stream

    def release(self) -> None:
        if self._msg is not None and self._msg.release:
            C.CFUNCTYPE(None, C.c_void_p)(self._msg.release)(C.addressof(self._msg))
        self._msg = None

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001
            pass


def record_to_ipc(record: Record, *, ctx: Optional[Context] = None, body_on_device: bool = False) -> IpcEncoded:
    """Arrow IPC stream encoding of one batch (messages/exchange.rs:145-197 of the reference) with the body assembled
    on the GPU (`chq_record_to_ipc`)."""
    ctx, src, own_src, _ = _prepare(record, ctx)
    msg = L.IpcMessage()
    try:
        rc = L.lib().chq_record_to_ipc(ctx.handle, C.byref(src.array), C.byref(src.schema),
                                       L.ARROW_DEVICE_ROCM if body_on_device else L.ARROW_DEVICE_CPU, C.byref(msg))
    finally:
        if own_src:
            src.release()
    if rc:
        raise ChqError(rc, ctx.last_error())
    return IpcEncoded(ctx, msg)


def record_from_ipc(stream: bytes, *, ctx: Optional[Context] = None, device_result: bool = True,
                    body_address: int = 0, body_len: int = 0, body_on_device: bool = True):
    """Inverse (`chq_record_from_ipc`, messages/exchange.rs:247-276): `stream` is a complete Arrow IPC stream, or -- with
    `body_address` -- only its metadata part, the body being read from that (device or host) address."""
    ctx = ctx or default_context()
    out = _CBatch()
    buf = (C.c_uint8 * len(stream)).from_buffer_copy(stream)
    rc = L.lib().chq_record_from_ipc(ctx.handle, C.addressof(buf), len(stream), body_address or None, body_len,
                                     L.ARROW_DEVICE_ROCM if body_on_device else L.ARROW_DEVICE_CPU,
                                     L.ARROW_DEVICE_ROCM if device_result else L.ARROW_DEVICE_CPU,
                                     C.byref(out.array), C.byref(out.schema))
    return _finish(ctx, rc, out, device_result)


def ipc_describe(stream: bytes) -> str:
    """Host half only: the metadata of an Arrow IPC stream as text (`chq_ipc_describe`); needs no GPU."""
    buf = C.create_string_buffer(1 << 16)
    data = (C.c_uint8 * max(1, len(stream))).from_buffer_copy(stream or b"\0")
    rc = L.lib().chq_ipc_describe(C.addressof(data), len(stream), buf, len(buf))
    text = buf.value.decode(errors="replace")
    if rc:
        raise ChqError(rc, text)
    return text


class ParquetFile:
    """A Parquet file whose pages are decoded on the GPU (`chq_parquet_*`) -- the device-side form
    of what `read_files_task.rs:233-282` does with `ParquetRecordBatchStreamBuilder`.  `source`: a path or the file's
    bytes; they stay referenced (the library borrows the memory) until `close()`."""

    def __init__(self, source, *, reader=None, size: Optional[int] = None):
        """`source`: a path or the file's bytes (the library borrows the memory).  Or `reader(offset, length) -> bytes` with
        `size` = the file's length (`chq_parquet_open_reader`): the footer is read at open, and each read call then fetches
        exactly the column chunks it decodes -- what the reference's opendal reader does (read_files_task.rs:233-250).
        `self.reads` records every (offset, length) the library asked for."""
        import numpy as np
        self._h = C.c_void_p()
        self.reads: List[Tuple[int, int]] = []
        err = C.create_string_buffer(1024)
        if reader is not None:
            if size is None:
                raise ValueError("a range reader needs the file's size")

            def _read(_user, offset, length, dst):
                try:
                    data = reader(offset, length)
                    if len(data) != length:
                        return 1
                    C.memmove(dst, bytes(data), length)
                    self.reads.append((int(offset), int(length)))
                    return 0
                except Exception:   # noqa: BLE001 -- a failing reader fails the call with a status, not a crash
                    return 2
            self._reader = L.READ_RANGE_FN(_read)   # (kept alive: the library calls it until close)
            rc = L.lib().chq_parquet_open_reader(int(size), self._reader, None, C.byref(self._h), err, len(err))
        else:
            if isinstance(source, (bytes, bytearray, memoryview)):
                self._bytes = np.frombuffer(source, dtype=np.uint8)
            else:
                self._bytes = np.fromfile(source, dtype=np.uint8)
            rc = L.lib().chq_parquet_open(self._bytes.ctypes.data, self._bytes.size, C.byref(self._h), err, len(err))
        if rc:
            raise ChqError(rc, err.value.decode(errors="replace"))

    @property
    def num_columns(self) -> int:
        return L.lib().chq_parquet_num_columns(self._h)

    @property
    def column_names(self) -> List[str]:
        return [L.lib().chq_parquet_column_name(self._h, i).decode() for i in range(self.num_columns)]

    @property
    def num_row_groups(self) -> int:
        return L.lib().chq_parquet_num_row_groups(self._h)

    def row_group_num_rows(self, i: int) -> int:
        return L.lib().chq_parquet_row_group_num_rows(self._h, i)

    def describe(self) -> str:
        """Host half only (no GPU): schema, row groups, column chunks and pages as text."""
        buf = C.create_string_buffer(1 << 22)
        rc = L.lib().chq_parquet_describe(self._h, buf, len(buf))
        if rc:
            raise ChqError(rc, "describe buffer too small")
        return buf.value.decode(errors="replace")

    def read_row_group(self, i: int, *, ctx: Optional[Context] = None, device_result: bool = True):
        ctx = ctx or default_context()
        out = _CBatch()
        rc = L.lib().chq_parquet_read_row_group(ctx.handle, self._h, i, L.ARROW_DEVICE_ROCM if device_result else L.ARROW_DEVICE_CPU,
                                                C.byref(out.array), C.byref(out.schema))
        return _finish(ctx, rc, out, device_result)

    def read_row_groups(self, first: int = 0, count: Optional[int] = None, *, ctx: Optional[Context] = None, device_result: bool = True,
                        columns: Optional[Sequence] = None):
        """Row groups [first, first + count) as one batch each (`chq_parquet_read_row_groups` / `chq_parquet_read_columns`):
        decoded together, two host synchronisations per call.  `columns`: names or indices of the columns to decode, in the
        order wanted (column pruning: only their chunks are fetched, uploaded and decoded); None = every column."""
        ctx = ctx or default_context()
        n = self.num_row_groups - first if count is None else count
        outs = (L.ArrowDeviceArray * max(n, 1))()
        schemas = (L.ArrowSchema * max(n, 1))()
        dev = L.ARROW_DEVICE_ROCM if device_result else L.ARROW_DEVICE_CPU
        if columns is None:
            rc = L.lib().chq_parquet_read_row_groups(ctx.handle, self._h, first, n, dev, outs, schemas)
        else:
            names = self.column_names
            idx = [names.index(c) if isinstance(c, str) else int(c) for c in columns]
            arr = (C.c_int32 * max(1, len(idx)))(*idx)
            rc = L.lib().chq_parquet_read_columns(ctx.handle, self._h, first, n, arr, len(idx), dev, outs, schemas)
        if rc:
            raise ChqError(rc, ctx.last_error())
        res = []
        for i in range(n):
            cb = _CBatch()   # struct copy = Arrow "move"
            C.memmove(C.addressof(cb.array), C.addressof(outs[i]), C.sizeof(L.ArrowDeviceArray))
            C.memmove(C.addressof(cb.schema), C.addressof(schemas[i]), C.sizeof(L.ArrowSchema))
            res.append(_finish(ctx, 0, cb, device_result))
        return res

    def close(self) -> None:
        if self._h:
            L.lib().chq_parquet_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ParquetImage:
    """The file image of `record_to_parquet(..., copy=False)`: `view` is a memoryview over the library's host buffer (no
    copy); `release()` hands the buffer back."""

    def __init__(self, img):
        self._img = img
        self.view = memoryview((C.c_uint8 * img.len).from_address(img.data)).cast("B") if img.len else memoryview(b"")

    def __len__(self):
        return self._img.len if self._img is not None else 0

    def release(self) -> None:
        if self._img is not None and self._img.release:
            self.view = memoryview(b"")
            C.CFUNCTYPE(None, C.c_void_p)(self._img.release)(C.addressof(self._img))
        self._img = None

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001
            pass


def record_to_parquet(record: Record, *, ctx: Optional[Context] = None, copy: bool = True):
    """One record batch (host or device resident) -> the bytes of one Parquet file with one row group, pages encoded on the
    GPU (`chq_record_to_parquet`; what materialize_files_task.rs:128-141 does with the parquet crate on the CPU).
    `copy=False` returns a `ParquetImage` over the library's own buffer instead of a `bytes` copy."""
    ctx, src, own_src, _ = _prepare(record, ctx)
    img = L.ParquetImage()
    try:
        rc = L.lib().chq_record_to_parquet(ctx.handle, C.byref(src.array), C.byref(src.schema), C.byref(img))
    finally:
        if own_src:
            src.release()
    if rc:
        raise ChqError(rc, ctx.last_error())
    if not copy:
        return ParquetImage(img)
    try:
        return C.string_at(img.data, img.len)
    finally:
        C.CFUNCTYPE(None, C.c_void_p)(img.release)(C.addressof(img))


def records_to_parquet(records: Sequence[Record], *, ctx: Optional[Context] = None, copy: bool = True):
    """Several record batches of one schema -> ONE Parquet file, one row group per batch in order
    (`chq_records_to_parquet`): the row-group compaction the reference plans for its materialize task (DEV_NOTES.md:117-121)."""
    if not records:
        raise ValueError("records_to_parquet needs at least one record batch")
    ctx = ctx or default_context()
    prepared = [_prepare(r, ctx) for r in records]
    arr = (C.POINTER(L.ArrowDeviceArray) * len(prepared))(*[C.pointer(p[1].array) for p in prepared])
    img = L.ParquetImage()
    try:
        rc = L.lib().chq_records_to_parquet(ctx.handle, len(prepared), arr, C.byref(prepared[0][1].schema), C.byref(img))
    finally:
        for _c, src, own, _ in prepared:
            if own:
                src.release()
    if rc:
        raise ChqError(rc, ctx.last_error())
    if not copy:
        return ParquetImage(img)
    try:
        return C.string_at(img.data, img.len)
    finally:
        C.CFUNCTYPE(None, C.c_void_p)(img.release)(C.addressof(img))


def scan_parquet(source, *, ctx: Optional[Context] = None, device_result: bool = True):
    """Every row group of a Parquet file as one batch each, decoded on the GPU (generator)."""
    f = ParquetFile(source)
    try:
        for batch in f.read_row_groups(ctx=ctx, device_result=device_result):
            yield batch
    finally:
        f.close()


class RecordGroup:
    """A prepared argument block for `filter_records`: the C pointer array over a list of same-schema batches.
    Building it once lets a caller that re-filters the same batches (benchmarks) keep Python out of the call."""

    def __init__(self, recs: Sequence[Record], ctx: Optional[Context] = None):
        if not recs:
            raise ValueError("empty record group")
        kinds = {isinstance(r, DeviceRecordBatch) for r in recs}
        if len(kinds) != 1:
            raise ValueError("a record group is all host batches or all device batches")
        self.on_device = kinds.pop()
        self.ctx = ctx or (recs[0].ctx if self.on_device else default_context())
        self._cbs = [r._cb if self.on_device else _export_host(r) for r in recs]
        self._own = not self.on_device
        self.n = len(recs)
        self.ptrs = (C.POINTER(L.ArrowDeviceArray) * self.n)(*[C.pointer(cb.array) for cb in self._cbs])
        self.schema = self._cbs[0].schema

    def release(self):
        if self._own:
            for cb in self._cbs:
                cb.release()
        self._cbs = []


def filter_records(recs, table_aliases: Optional[Sequence[Sequence[str]]], expr: A.Expr, *,
                   ctx: Optional[Context] = None, device_result: Optional[bool] = None, wrap: bool = True):
    """`[filter_record(r, table_aliases, expr) for r in recs]` in one call (`chq_filter_records`): the loop of
    filter_task.rs:78-126 below the boundary.  Same-schema batches; fixed-width null-free groups run in ONE
    kernel launch.  `recs` is a sequence of batches or a prepared `RecordGroup`.  With `wrap=False` the outputs
    are released immediately and only the per-batch row counts are returned (benchmarks)."""
    grp = recs if isinstance(recs, RecordGroup) else RecordGroup(recs, ctx)
    ctx = ctx or grp.ctx
    dev_out = grp.on_device if device_result is None else device_result
    e = _expr_to_c(expr)
    al = _Aliases(table_aliases)
    outs = (L.ArrowDeviceArray * grp.n)()
    schemas = (L.ArrowSchema * grp.n)()
    try:
        rc = L.lib().chq_filter_records(ctx.handle, grp.n, grp.ptrs, C.byref(grp.schema), al.ptr, e,
                                        L.ARROW_DEVICE_ROCM if dev_out else L.ARROW_DEVICE_CPU, outs, schemas)
    finally:
        L.lib().chq_expr_free(e)
        if grp is not recs:
            grp.release()
    if rc:
        raise ChqError(rc, ctx.last_error())
    results = []
    for i in range(grp.n):
        cb = _CBatch()   # struct copy = Arrow "move": the slots of the call arrays are never touched again
        C.memmove(C.addressof(cb.array), C.addressof(outs[i]), C.sizeof(L.ArrowDeviceArray))
        C.memmove(C.addressof(cb.schema), C.addressof(schemas[i]), C.sizeof(L.ArrowSchema))
        if not wrap:
            results.append(int(cb.array.array.length))
            cb.release()
        else:
            results.append(DeviceRecordBatch(ctx, cb) if dev_out else _import_host(cb))
    return results


def filter_records_coalesced(recs, table_aliases: Optional[Sequence[Sequence[str]]], expr: A.Expr, *,
                             ctx: Optional[Context] = None, device_result: Optional[bool] = None):
    """`filter_records` with the outputs joined (`chq_filter_records_coalesced`): returns (one batch holding the
    surviving rows of every input batch in input order, [surviving rows per input batch])."""
    grp = recs if isinstance(recs, RecordGroup) else RecordGroup(recs, ctx)
    ctx = ctx or grp.ctx
    dev_out = grp.on_device if device_result is None else device_result
    e = _expr_to_c(expr)
    al = _Aliases(table_aliases)
    out = _CBatch()
    rows = (C.c_int64 * grp.n)()
    try:
        rc = L.lib().chq_filter_records_coalesced(ctx.handle, grp.n, grp.ptrs, C.byref(grp.schema), al.ptr, e,
                                                  L.ARROW_DEVICE_ROCM if dev_out else L.ARROW_DEVICE_CPU,
                                                  C.byref(out.array), C.byref(out.schema), rows)
    finally:
        L.lib().chq_expr_free(e)
        if grp is not recs:
            grp.release()
    return _finish(ctx, rc, out, dev_out), list(rows)


def project_record(fields: Sequence[A.SelectItem], record: Record, table_aliases: Optional[Sequence[Sequence[str]]], *,
                   ctx: Optional[Context] = None, device_result: Optional[bool] = None):
    """RU/record_projection.rs:16-76."""
    ctx, src, own_src, on_dev = _prepare(record, ctx)
    dev_out = on_dev if device_result is None else device_result
    items, exprs = _items_to_c(fields)
    al = _Aliases(table_aliases)
    out = _CBatch()
    try:
        rc = L.lib().chq_project_record(ctx.handle, items, len(fields), C.byref(src.array), C.byref(src.schema), al.ptr,
                                        L.ARROW_DEVICE_ROCM if dev_out else L.ARROW_DEVICE_CPU, C.byref(out.array), C.byref(out.schema))
    finally:
        for x in exprs:
            L.lib().chq_expr_free(x)
        if own_src:
            src.release()
    return _finish(ctx, rc, out, dev_out)


def filter_project_record(predicate: A.Expr, fields: Sequence[A.SelectItem], record: Record,
                          table_aliases: Optional[Sequence[Sequence[str]]], *, ctx: Optional[Context] = None,
                          device_result: Optional[bool] = None):
    """filter_record followed by project_record on the survivors, without leaving the GPU
    (the reference's filter -> exchange -> materialize sequence, filter_task.rs:99 + materialize_files_task.rs:110)."""
    ctx, src, own_src, on_dev = _prepare(record, ctx)
    dev_out = on_dev if device_result is None else device_result
    e = _expr_to_c(predicate)
    items, exprs = _items_to_c(fields)
    al = _Aliases(table_aliases)
    out = _CBatch()
    try:
        rc = L.lib().chq_filter_project_record(ctx.handle, e, items, len(fields), C.byref(src.array), C.byref(src.schema), al.ptr,
                                               L.ARROW_DEVICE_ROCM if dev_out else L.ARROW_DEVICE_CPU, C.byref(out.array), C.byref(out.schema))
    finally:
        L.lib().chq_expr_free(e)
        for x in exprs:
            L.lib().chq_expr_free(x)
        if own_src:
            src.release()
    return _finish(ctx, rc, out, dev_out)


def compute_value(rec: Record, table_aliases: Optional[Sequence[Sequence[str]]], expr: A.Expr, *,
                  ctx: Optional[Context] = None) -> Tuple[pa.Array, bool]:
    """RU/compute_value.rs:57-344.  Returns (array, is_scalar) like the reference's ArrayDatum."""
    ctx, src, own_src, _ = _prepare(rec, ctx)
    e = _expr_to_c(expr)
    al = _Aliases(table_aliases)
    out = _CBatch()
    sc = C.c_int(0)
    try:
        rc = L.lib().chq_compute_value(ctx.handle, C.byref(src.array), C.byref(src.schema), al.ptr, e, L.ARROW_DEVICE_CPU,
                                       C.byref(out.array), C.byref(out.schema), C.byref(sc))
    finally:
        L.lib().chq_expr_free(e)
        if own_src:
            src.release()
    if rc:
        raise ChqError(rc, ctx.last_error())
    arr = pa.Array._import_from_c(C.addressof(out.array.array), C.addressof(out.schema))
    return arr, bool(sc.value)


def get_record_table_aliases(alias: Optional[str], record) -> List[List[str]]:
    """RU/record_aliases.rs:12-59: one alias list per column, `[alias]` or `[]`, taken from the producing
    TableFunc / Table operator."""
    n = record.num_columns
    return [[alias] for _ in range(n)] if alias is not None else [[] for _ in range(n)]
