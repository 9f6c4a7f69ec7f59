"""Per-call latency of filter_record at the reference's batch size (10 000 rows, physical_planner.rs:323) and a few
larger ones: device-resident input/output vs host batches through the same ABI (PCIe + staging included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa
import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr

ctx = chq.Context(0)
e = parse_expr("value2 > 10.0")
for n in [10_000, 100_000, 1_000_000, 10_000_000]:
    rng = np.random.default_rng(1)
    rb = pa.RecordBatch.from_arrays([pa.array(np.arange(n, dtype=np.int32)), pa.array((rng.random(n) * 100).astype(np.float32)),
                                     pa.array((rng.random(n) * 100).astype(np.float32))], names=["id", "value1", "value2"])
    al = [[], [], []]
    dev = chq.DeviceRecordBatch.from_host(rb, ctx)
    for label, rec in (("device", dev), ("host", rb)):
        for _ in range(5):
            out = chq.filter_record(rec, al, e, ctx=ctx)
            if label == "device": out.release()
        reps = 200 if n <= 100_000 else 30
        t0 = time.perf_counter()
        for _ in range(reps):
            out = chq.filter_record(rec, al, e, ctx=ctx)
            if label == "device": out.release()
        dt = (time.perf_counter() - t0) / reps
        print(f"n={n:>9d} {label:6s} {dt * 1e6:9.1f} us/call  {n / dt:12.3e} rows/s", flush=True)
