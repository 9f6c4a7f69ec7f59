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
