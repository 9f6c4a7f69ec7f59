"""One large HOST-resident batch through chq_filter_record (PCIe both ways): the serial path (upload, kernel, download)
against the chunked path whose uploads and downloads overlap.  usage: python bench/micro/host_large.py [rows]"""
import sys
import time

import numpy as np
import pyarrow as pa

sys.path.insert(0, ".")
import chapterhouseqe_amd as chq   # noqa: E402
from chapterhouseqe_amd.sqlparse import parse_expr   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
rng = np.random.default_rng(0)
rec = pa.record_batch({"id": pa.array(np.arange(n, dtype=np.int32)), "value1": pa.array((rng.random(n) * 100).astype(np.float32)),
                       "value2": pa.array((rng.random(n) * 100).astype(np.float32))})
al = [[], [], []]
e = parse_expr("value2 > 10.0")
ctx = chq.Context(0)
ref = None
for mode in (0, 1):
    ctx.set_option("large_host", mode)
    best = 1e9
    for it in range(5):
        t0 = time.perf_counter()
        out = chq.filter_record(rec, al, e, ctx=ctx)
        best = min(best, time.perf_counter() - t0)
    if ref is None:
        ref = out
    else:
        assert out.equals(ref)
    print(f"large_host={mode}: {best * 1e3:.1f} ms = {n / best / 1e9:.2f} G rows/s, {(n * 12 + out.num_rows * 12) / best / 1e9:.1f} GB/s over PCIe (both directions), launches {ctx.last_stats()['launches']}")
