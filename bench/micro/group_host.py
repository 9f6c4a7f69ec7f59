"""Reference-shaped input: host-resident 10 000-row batches (physical_planner.rs:323), `simple` schema (id:i32, value1:Utf8(8),
value2:f32, create_sample_data.rs:113-155) and its numeric-only projection, filtered (a) one chq_filter_record call per
batch -- the reference's loop -- and (b) by one chq_filter_records call.  PCIe both ways included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa
import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr

nb, rows = 2000, 10_000
rng = np.random.default_rng(7)
ids = np.arange(nb * rows, dtype=np.int32)
v2 = (rng.random(nb * rows) * 100).astype(np.float32)
chars = rng.integers(ord("a"), ord("z") + 1, nb * rows * 8, dtype=np.uint8)
offs = (np.arange(nb * rows + 1, dtype=np.int64) * 8).astype(np.int32)
strs = pa.Array.from_buffers(pa.utf8(), nb * rows, [None, pa.py_buffer(offs.tobytes()), pa.py_buffer(chars.tobytes())])
full = pa.RecordBatch.from_arrays([pa.array(ids), strs, pa.array(v2)], names=["id", "value1", "value2"])
numeric = pa.RecordBatch.from_arrays([pa.array(ids), pa.array(v2), pa.array(v2)], names=["id", "value1", "value2"])
ctx = chq.Context(0)
e = parse_expr("value2 > 10.0")
al = [[], [], []]
for label, table in (("numeric (i32, f32, f32)", numeric), ("simple (i32, Utf8(8), f32)", full)):
    batches = [table.slice(b * rows, rows) for b in range(nb)]
    for rep in range(2):
        t0 = time.perf_counter()
        outs = [chq.filter_record(b, al, e, ctx=ctx) for b in batches]
        t_loop = time.perf_counter() - t0
        n_loop = sum(o.num_rows for o in outs)
        t0 = time.perf_counter()
        outs = chq.filter_records(batches, al, e, ctx=ctx)
        t_grp = time.perf_counter() - t0
        st = ctx.last_stats()
        assert sum(o.num_rows for o in outs) == n_loop
        grp = chq.RecordGroup(batches, ctx)           # Arrow structs exported once; outputs released right away
        t0 = time.perf_counter()
        counts = chq.filter_records(grp, al, e, ctx=ctx, wrap=False)
        t_raw = time.perf_counter() - t0
        grp.release()
        assert sum(counts) == n_loop
    print(f"{label}: {nb} x {rows} rows | per-batch loop {t_loop * 1e3:.1f} ms = {nb * rows / t_loop:.3e} rows/s ({t_loop / nb * 1e6:.0f} us/batch)"
          f" | one group call {t_grp * 1e3:.1f} ms = {nb * rows / t_grp:.3e} rows/s, {st['launches']} launches"
          f" | without the pyarrow export/import of {nb} batches {t_raw * 1e3:.1f} ms = {nb * rows / t_raw:.3e} rows/s", flush=True)
