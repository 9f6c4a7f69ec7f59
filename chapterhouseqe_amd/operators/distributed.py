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


# ---------------------------------------------------------------------------------------------------------------
# Wire format of one record = the reference's (messages/exchange.rs:180-197): a small envelope -- u8 variant, u64
# big-endian length, JSON metadata (record id, table aliases) -- followed by an Arrow IPC *stream* (Schema message,
# one RecordBatch message, end-of-stream).  A host batch travels as ONE byte string (pyarrow writes the stream).  A batch
# in HBM travels as the envelope + IPC metadata (host bytes, a few hundred) and the IPC *body* as ONE device buffer
# (`chq_record_to_ipc` assembles it in a single HBM allocation) sent peer to peer by RCCL over the xGMI link of the
# pair; the receiver wraps it with `chq_record_from_ipc`.  The rows never touch the host.
# ---------------------------------------------------------------------------------------------------------------
_VARIANT_SEND_RECORD = 1


def _envelope(meta: dict, ipc: bytes) -> bytes:
    import struct
    m = json.dumps(meta).encode()
    return struct.pack(">BQ", _VARIANT_SEND_RECORD, len(m)) + m + ipc


def _open_envelope(frame: bytes):
    import struct
    variant, mlen = struct.unpack(">BQ", frame[:9])
    if variant != _VARIANT_SEND_RECORD:
        raise ValueError(f"unexpected exchange message variant {variant}")
    return json.loads(frame[9:9 + mlen].decode()), frame[9 + mlen:]


def _send_bytes(data: bytes, dst: int, device) -> None:
    import torch
    import torch.distributed as dist
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8)
    t = t.to(device) if device is not None else t
    dist.send(torch.tensor([t.numel()], dtype=torch.int64, device=device), dst)
    if t.numel():
        dist.send(t, dst)


def _recv_bytes(src: int, device) -> bytes:
    import torch
    import torch.distributed as dist
    n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(n, src)
    if int(n.item()) == 0:
        return b""
    t = torch.zeros(int(n.item()), dtype=torch.uint8, device=device)
    dist.recv(t, src)
    return t.cpu().numpy().tobytes()


def record_to_stream(rec: pa.RecordBatch) -> bytes:
    """Arrow IPC stream of a host batch (what arrow::ipc::writer::StreamWriter produces in the reference)"""
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, rec.schema) as w:
        w.write_batch(rec)
    return sink.getvalue().to_pybytes()


def stream_to_record(stream: bytes) -> pa.RecordBatch:
    t = pa.ipc.open_stream(stream).read_all()
    batches = t.combine_chunks().to_batches()
    if len(batches) > 1:
        raise ValueError("received multiple record batches")       # exchange.rs:262-266
    return batches[0] if batches else pa.RecordBatch.from_pylist([], schema=t.schema)


def send_record(rec: pa.RecordBatch, record_id: int, dst: int, device=None, table_aliases=None) -> None:
    """Point-to-point transfer of one host record batch to rank `dst`: one framed byte string."""
    _send_bytes(_envelope({"record_id": record_id, "table_aliases": table_aliases}, record_to_stream(rec)), dst, device)


def recv_record(src: int, device=None):
    """-> (record_id, RecordBatch, table_aliases), like recv_device_record"""
    meta, stream = _open_envelope(_recv_bytes(src, device))
    return meta["record_id"], stream_to_record(stream), meta.get("table_aliases")


def send_device_record(rec, record_id: int, dst: int, table_aliases=None) -> None:
    """Point-to-point transfer of a batch that lives in HBM to the GPU of rank `dst` (backend "nccl" = RCCL): the IPC
    metadata in the envelope, the IPC body as one device buffer."""
    import torch
    import torch.distributed as dist
    from ..record_utils import record_to_ipc
    device = torch.device("cuda", rec.ctx.device_id)
    enc = record_to_ipc(rec, ctx=rec.ctx, body_on_device=True)
    try:
        _send_bytes(_envelope({"record_id": record_id, "table_aliases": table_aliases, "body_len": enc.body_len}, enc.header), dst, device)
        if enc.body_len:
            dist.send(_as_tensor(enc.body_address, enc.body_len, enc, device), dst)
        # With RCCL the send is only enqueued (on torch's current stream) and the tensor is a zero-copy view of the
        # library's HBM buffer, not caching-allocator memory: drain it before the buffer goes back to the pool.
        torch.cuda.current_stream(device).synchronize()
    finally:
        enc.release()


def recv_device_record(src: int, ctx):
    """-> (record_id, DeviceRecordBatch, table_aliases); the body lands in this rank's HBM and stays there."""
    import torch
    import torch.distributed as dist
    from ..record_utils import record_from_ipc
    device = torch.device("cuda", ctx.device_id)
    meta, header = _open_envelope(_recv_bytes(src, device))
    body_len = int(meta["body_len"])
    body = torch.empty(max(body_len, 1) + 64, dtype=torch.uint8, device=device)
    if body_len:
        dist.recv(body[:body_len], src)
    # the batch is handed to a chq context that launches on its OWN stream: every byte must have landed first
    torch.cuda.current_stream(device).synchronize()
    rec = record_from_ipc(header, ctx=ctx, device_result=True, body_address=body.data_ptr(), body_len=body_len, body_on_device=True)
    return meta["record_id"], rec, meta.get("table_aliases")


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
            send_record(record, record_id, dst, device, aliases)
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
                record_id, record, aliases = recv_record(src, device)
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
