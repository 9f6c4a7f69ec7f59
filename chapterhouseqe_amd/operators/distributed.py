"""One operator instance per GPU: batch sharding and the data-plane exchange over torch.distributed.

Record batches are independent units (each output keeps its input record id, filter_task.rs:106-113), so the path
shards with no data-path collective: rank r of W processes the records the shared RecordPool would hand to the
r-th instance -- statically `record_id % W == r` across processes.  Only two things cross ranks:
  * row counts / metrics: one all-reduce of a few int64 (RCCL on GPUs, gloo in the CPU tests);
  * a batch that the DAG forces onto another GPU (single materialize instance, rebalancing): point-to-point
    send/recv of its Arrow buffers, peer to peer over the xGMI link between the pair -- never a ring collective.
backend "nccl" IS RCCL on ROCm.
"""
from __future__ import annotations

import io
import json
from typing import Dict, List, Sequence, Tuple

import numpy as np
import pyarrow as pa


def shard_record_ids(record_ids: Sequence[int], rank: int, world_size: int) -> List[int]:
    """static sharding of record ids over operator instances"""
    return [r for r in record_ids if r % world_size == rank]


def all_reduce_counts(counts: Dict[str, int], device=None) -> Dict[str, int]:
    """Sum per-instance metrics (rows in / out, records) over all ranks."""
    import torch
    import torch.distributed as dist
    keys = sorted(counts)
    t = torch.tensor([counts[k] for k in keys], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: int(v) for k, v in zip(keys, t.tolist())}


def _batch_to_tensors(rec: pa.RecordBatch):
    """Arrow buffers of a host batch as uint8 tensors + a JSON description (schema, lengths, buffer sizes)."""
    import torch
    meta = {"num_rows": rec.num_rows, "schema": rec.schema.serialize().to_pybytes().hex(), "columns": []}
    tensors = []
    for col in rec.columns:
        col = pa.concat_arrays([col]) if col.offset else col    # normalise slices before shipping
        bufs = col.buffers()
        sizes = []
        for b in bufs:
            if b is None:
                sizes.append(-1)
                continue
            sizes.append(b.size)
            tensors.append(torch.frombuffer(memoryview(b), dtype=torch.uint8).clone() if b.size else torch.empty(0, dtype=torch.uint8))
        meta["columns"].append({"null_count": col.null_count, "buffers": sizes})
    return meta, tensors


def send_record(rec: pa.RecordBatch, record_id: int, dst: int, device=None) -> None:
    """Point-to-point transfer of one record batch (host representation) to rank `dst`."""
    import torch
    import torch.distributed as dist
    meta, tensors = _batch_to_tensors(rec)
    meta["record_id"] = record_id
    blob = torch.frombuffer(bytearray(json.dumps(meta).encode()), dtype=torch.uint8).to(device) if device is not None else \
        torch.frombuffer(bytearray(json.dumps(meta).encode()), dtype=torch.uint8)
    dist.send(torch.tensor([blob.numel()], dtype=torch.int64, device=device), dst)
    dist.send(blob, dst)
    for t in tensors:
        if t.numel():
            dist.send(t.to(device) if device is not None else t, dst)


def recv_record(src: int, device=None) -> Tuple[int, pa.RecordBatch]:
    import torch
    import torch.distributed as dist
    n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(n, src)
    blob = torch.zeros(int(n.item()), dtype=torch.uint8, device=device)
    dist.recv(blob, src)
    meta = json.loads(blob.cpu().numpy().tobytes().decode())
    schema = pa.ipc.read_schema(pa.py_buffer(bytes.fromhex(meta["schema"])))
    arrays = []
    for field, cm in zip(schema, meta["columns"]):
        bufs = []
        for size in cm["buffers"]:
            if size < 0:
                bufs.append(None)
                continue
            t = torch.zeros(size, dtype=torch.uint8, device=device)
            if size:
                dist.recv(t, src)
            bufs.append(pa.py_buffer(t.cpu().numpy().tobytes()))
        arrays.append(pa.Array.from_buffers(field.type, meta["num_rows"], bufs, null_count=cm["null_count"]))
    return meta["record_id"], pa.RecordBatch.from_arrays(arrays, schema=schema)


# ---------------------------------------------------------------------------------------------------------------
# Device-resident exchange: the Arrow buffers of a batch in HBM go peer to peer (RCCL send/recv over the xGMI link
# between the two GPUs) without touching the host; only the small JSON header is built on the CPU.
# ---------------------------------------------------------------------------------------------------------------
_WIDTHS = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "l": 8, "L": 8, "e": 2, "f": 4, "g": 8}


class _HbmRange:
    """a raw HBM range exposed to torch through __cuda_array_interface__ (zero copy); keeps its owner alive"""

    def __init__(self, ptr: int, nbytes: int, owner):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        self._owner = owner


def _as_tensor(ptr: int, nbytes: int, owner, device):
    import torch
    if nbytes == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(_HbmRange(ptr, nbytes, owner), device=device)


def _fixed_width(fmt: str) -> int:
    if fmt in _WIDTHS:
        return _WIDTHS[fmt]
    if fmt.startswith("w:"):
        return int(fmt[2:])
    if fmt.startswith("d:"):
        parts = fmt[2:].split(",")
        return int(parts[2]) // 8 if len(parts) == 3 else 16
    if fmt in ("tdD", "tts", "ttm"):
        return 4
    if fmt in ("tdm", "ttu", "ttn") or fmt.startswith("ts") or fmt.startswith("tD"):
        return 8
    raise ValueError(f"no fixed width for Arrow format {fmt!r}")


def device_record_to_tensors(rec, device=None):
    """(header, tensors): zero-copy uint8 views of every Arrow buffer of a DeviceRecordBatch, in header order.
    Buffers are sent from element 0 to offset + length, so the receiver keeps the same Arrow offsets."""
    import torch
    device = device if device is not None else torch.device("cuda", rec.ctx.device_id)
    header = {"num_rows": rec.num_rows, "columns": []}
    tensors = []
    for col in rec.describe_columns():
        end = col["offset"] + rec.num_rows
        fmt = col["format"]
        sizes = {"validity": 0, "values": 0, "data": 0}
        if col["validity"] and col["null_count"] != 0:
            sizes["validity"] = (end + 7) // 8
        if fmt == "b":
            sizes["values"] = (end + 7) // 8
        elif fmt == "u":
            # offsets stay absolute: only the bytes the logical rows use, [offsets[offset], offsets[end]), are sent and
            # the receiver rebases its data pointer by the first offset
            sizes["values"] = 4 * (end + 1)
            if col["values"] and rec.num_rows:
                first = int(_as_tensor(col["values"] + 4 * col["offset"], 4, rec, device).view(torch.int32).item())
                last = int(_as_tensor(col["values"] + 4 * end, 4, rec, device).view(torch.int32).item())
                sizes["data"] = last - first
                col = dict(col, data=col["data"] + first, data_base=first)
        else:
            sizes["values"] = _fixed_width(fmt) * end
        for key in ("validity", "values", "data"):
            if sizes[key]:
                tensors.append(_as_tensor(col[key], sizes[key], rec, device))
        header["columns"].append({k: col[k] for k in ("name", "format", "nullable", "null_count", "offset")} |
                                 {"sizes": sizes, "data_base": col.get("data_base", 0)})
    return header, tensors


def tensors_to_device_record(header, tensors, ctx):
    """Inverse of `device_record_to_tensors`: wraps the received tensors (kept alive by the batch) as a DeviceRecordBatch."""
    from ..record_utils import DeviceRecordBatch
    cols, k = [], 0
    for c in header["columns"]:
        d = {key: c[key] for key in ("name", "format", "nullable", "null_count", "offset")}
        for key in ("validity", "values", "data"):
            if c["sizes"][key]:
                d[key] = tensors[k].data_ptr() - (c.get("data_base", 0) if key == "data" else 0)
                k += 1
        if c["format"] == "u" and not c["sizes"]["data"]:
            d["data"] = 0
        if not c["sizes"]["validity"]:
            d["null_count"] = 0
        cols.append(d)
    return DeviceRecordBatch.from_device_buffers(cols, header["num_rows"], ctx, keepalive=list(tensors))


def send_device_record(rec, record_id: int, dst: int, table_aliases=None) -> None:
    """Point-to-point transfer of a batch that lives in HBM to the GPU of rank `dst` (backend "nccl" = RCCL)."""
    import torch
    import torch.distributed as dist
    header, tensors = device_record_to_tensors(rec)
    header["record_id"] = record_id
    header["table_aliases"] = table_aliases
    device = tensors[0].device if tensors else torch.device("cuda", rec.ctx.device_id)
    blob = torch.frombuffer(bytearray(json.dumps(header).encode()), dtype=torch.uint8).to(device)
    dist.send(torch.tensor([blob.numel()], dtype=torch.int64, device=device), dst)
    dist.send(blob, dst)
    for t in tensors:
        dist.send(t, dst)
    # With RCCL the sends are only enqueued (on torch's current stream).  The tensors are zero-copy views of the batch's
    # HBM buffers, not caching-allocator memory: once the caller acks the record the buffers may go back to the library's
    # pool and be reused by a chq call on another stream.  Drain the sends before returning (gloo sends are synchronous).
    if device.type == "cuda":
        torch.cuda.current_stream(device).synchronize()


def recv_device_record(src: int, ctx):
    """-> (record_id, DeviceRecordBatch, table_aliases); the buffers land in this rank's HBM and stay there."""
    import torch
    import torch.distributed as dist
    device = torch.device("cuda", ctx.device_id)
    n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(n, src)
    blob = torch.zeros(int(n.item()), dtype=torch.uint8, device=device)
    dist.recv(blob, src)
    header = json.loads(blob.cpu().numpy().tobytes().decode())
    tensors = []
    for c in header["columns"]:
        for key in ("validity", "values", "data"):
            if c["sizes"][key]:
                t = torch.empty(c["sizes"][key] + 16, dtype=torch.uint8, device=device)[: c["sizes"][key]]
                dist.recv(t, src)
                tensors.append(t)
    # the batch is handed to a chq context that launches on its OWN stream with no sync_event: every byte must have
    # landed before it is wrapped
    torch.cuda.current_stream(device).synchronize()
    return header["record_id"], tensors_to_device_record(header, tensors, ctx), header.get("table_aliases")


# ---------------------------------------------------------------------------------------------------------------
# Exchange-level forwarding: the DAG puts a consumer (e.g. the single materialize instance) on another rank than some of
# its producers (one filter instance per GPU).  A forwarder on the producer's rank drains the local exchange as the
# consumer operator would (get_next_record -> ship -> operator_completed_record_processing, so the pool's ack and
# garbage-collection rules are unchanged) and the receiving rank adds the records to its own exchange under their
# original record ids.  Host batches travel with send_record, batches in HBM with send_device_record (RCCL over xGMI).
# ---------------------------------------------------------------------------------------------------------------
_END_OF_STREAM = -1


def forward_exchange(ex, operator_id: str, instance_id: int, dst: int, device=None, poll_s: float = 0.002) -> int:
    """Drain `ex` on behalf of consumer operator `operator_id` and ship every record to rank `dst`; returns the number of
    records shipped.  Ends (after the exchange reports NONE_LEFT) with an end-of-stream marker."""
    import time
    from ..record_utils import DeviceRecordBatch
    from .exchange_operator import NONE_AVAILABLE, NONE_LEFT
    shipped = 0
    device = _default_device(device)
    while True:
        got = ex.get_next_record(operator_id, instance_id)
        if got == NONE_LEFT:
            break
        if got == NONE_AVAILABLE:
            time.sleep(poll_s)
            continue
        record_id, record, aliases = got
        if isinstance(record, DeviceRecordBatch):
            _send_kind(1, dst, device)
            send_device_record(record, record_id, dst, aliases)
        else:
            _send_kind(0, dst, device)
            send_record(record, record_id, dst, device)
            _send_json(aliases, dst, device)
        ex.operator_completed_record_processing(operator_id, record_id)
        shipped += 1
    _send_kind(_END_OF_STREAM, dst, device)
    return shipped


def receive_into_exchange(ex, srcs: Sequence[int], ctx=None, device=None) -> int:
    """Receive what `forward_exchange` ships from every rank in `srcs` (one after the other) and add it to `ex`; returns
    the number of records added.  `ctx`: the chq context that will own received HBM batches."""
    added = 0
    device = _default_device(device)
    for src in srcs:
        while True:
            kind = _recv_kind(src, device)
            if kind == _END_OF_STREAM:
                break
            if kind == 1:
                record_id, record, aliases = recv_device_record(src, ctx)
            else:
                record_id, record = recv_record(src, device)
                aliases = _recv_json(src, device)
            ex.send_record(record_id, record, aliases)
            added += 1
    return added


def _default_device(device):
    """control tensors must live where the backend can reach them: HBM for nccl (= RCCL), host memory for gloo"""
    import torch
    import torch.distributed as dist
    if device is None and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return device


def _send_kind(kind: int, dst: int, device) -> None:
    import torch
    import torch.distributed as dist
    dist.send(torch.tensor([kind], dtype=torch.int64, device=device), dst)


def _recv_kind(src: int, device) -> int:
    import torch
    import torch.distributed as dist
    t = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(t, src)
    return int(t.item())


def _send_json(obj, dst: int, device) -> None:
    import torch
    import torch.distributed as dist
    blob = torch.frombuffer(bytearray(json.dumps(obj).encode()), dtype=torch.uint8)
    blob = blob.to(device) if device is not None else blob
    dist.send(torch.tensor([blob.numel()], dtype=torch.int64, device=device), dst)
    if blob.numel():
        dist.send(blob, dst)


def _recv_json(src: int, device):
    import torch
    import torch.distributed as dist
    n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(n, src)
    if int(n.item()) == 0:
        return None
    blob = torch.zeros(int(n.item()), dtype=torch.uint8, device=device)
    dist.recv(blob, src)
    return json.loads(blob.cpu().numpy().tobytes().decode())
