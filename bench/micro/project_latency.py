"""Per-call latency of project_record / filter_project_record at the reference's batch size, host batches (the
materialize task's calling pattern, materialize_files_task.rs:110) and device-resident ones."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa
import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_select

ctx = chq.Context(0)
sel = parse_select("select id, id + 10.0 as id_plus_10, (value2 + 10) / 100 as v2 from t where value2 > 10.0")
for n in [10_000, 100_000]:
    rng = np.random.default_rng(1)
    rb = pa.RecordBatch.from_arrays([pa.array(np.arange(n, dtype=np.int32)), pa.array((rng.random(n) * 100).astype(np.float32)),
                                     pa.array((rng.random(n) * 100).astype(np.float32))], names=["id", "value1", "value2"])
    al = [[], [], []]
    dev = chq.DeviceRecordBatch.from_host(rb, ctx)
    for label, rec in (("device", dev), ("host", rb)):
        for name, fn in (("project_record", lambda r: chq.project_record(sel.projection, r, al, ctx=ctx)),
                         ("filter_project_record", lambda r: chq.filter_project_record(sel.selection, sel.projection, r, al, ctx=ctx))):
            for _ in range(5):
                o = fn(rec)
            t0 = time.perf_counter()
            for _ in range(200):
                o = fn(rec)
                if label == "device": o.release()
            dt = (time.perf_counter() - t0) / 200
            print(f"n={n:>7d} {label:6s} {name:22s} {dt * 1e6:8.1f} us/call", flush=True)
