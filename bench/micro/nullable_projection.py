#!/usr/bin/env python3
"""projection / filter of columns WITH nulls at scale: how much the null bookkeeping (validity words, null counts) costs next
to the same call on non-null columns.  python bench/micro/nullable_projection.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import chapterhouseqe_amd as chq  # noqa: E402
from chapterhouseqe_amd.sqlparse import parse_expr, parse_select  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
dev = torch.device("cuda", 0)
ctx = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(7)
a = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
b = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
words = (n + 63) // 64
m = torch.randint(-2**63, 2**63 - 1, (words,), dtype=torch.int64, device=dev, generator=g)
for _ in range(3):
    m &= torch.randint(-2**63, 2**63 - 1, (words,), dtype=torch.int64, device=dev, generator=g)
valid = ~m            # bit = 0 (null) where all four random bits were 1: 6.25 %
torch.cuda.synchronize()


def batch(nullable):
    cols = [("a", "f", a.data_ptr()), ("b", "f", b.data_ptr())]
    if nullable:
        nulls = n - int(sum(int(torch.sum(torch.bitwise_and(valid >> k, 1)).item()) for k in range(64)))   # (bits past n ignored: n is a multiple of 64 here)
        return chq.DeviceRecordBatch.from_device_buffers(
            [{"name": "a", "format": "f", "nullable": True, "null_count": nulls, "validity": valid.data_ptr(), "values": a.data_ptr()},
             {"name": "b", "format": "f", "values": b.data_ptr()}], n, ctx=ctx, keepalive=[a, b, valid])
    return chq.DeviceRecordBatch.from_device_pointers(cols, n, ctx=ctx, keepalive=[a, b])


def timed(fn, reps=5):
    fn().release()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn().release()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


al = [[], []]
sel = parse_select("select a + b as s, a * 2.0 as d from t")
pred = parse_expr("a > 10.0")
for nullable in (False, True):
    rec = batch(nullable)
    tp = timed(lambda: chq.project_record(sel.projection, rec, al, ctx=ctx))
    tf = timed(lambda: chq.filter_record(rec, al, pred, ctx=ctx))
    print(f"{n} rows, column a {'with 6 % nulls' if nullable else 'non-null'}: project_record {tp:.2f} ms, filter_record {tf:.2f} ms", flush=True)
