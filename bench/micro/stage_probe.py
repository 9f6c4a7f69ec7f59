"""Per-stage latency of the device pipeline on 1 Mi-row row groups (development aid for bench/micro/pipeline.py)."""
import io, sys, time
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
sys.path.insert(0, ".")
import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_select
n = 8_000_000
rng = np.random.default_rng(0)
letters = rng.integers(ord("a"), ord("z") + 1, (n, 8), dtype=np.uint8)
value1 = pa.Array.from_buffers(pa.utf8(), n, [None, pa.py_buffer((np.arange(n + 1, dtype=np.int32) * 8).tobytes()), pa.py_buffer(letters.tobytes())])
t = pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "value1": value1, "value2": pa.array((rng.random(n) * 100).astype(np.float32))})
buf = io.BytesIO(); pq.write_table(t, buf, compression="none", row_group_size=1 << 20); raw = buf.getvalue()
sel = parse_select("select id, value1, value2 * 2.0 as twice from t where value2 > 10.0")
ctx = chq.Context(0)
al = [[], [], []]
for rep in range(3):
    f = chq.ParquetFile(raw)
    devs = f.read_row_groups(ctx=ctx)
    times = []
    for dev in devs:
        t0 = time.perf_counter()
        if rep == 2:
            k = chq.filter_record(dev, al, sel.selection, ctx=ctx)
            t1 = time.perf_counter()
            p = chq.project_record(sel.projection, k, al, ctx=ctx)
        else:
            k = dev
            t1 = time.perf_counter()
            p = chq.filter_project_record(sel.selection, sel.projection, dev, al, ctx=ctx)
        t2 = time.perf_counter()
        img = chq.record_to_parquet(p, ctx=ctx)
        t3 = time.perf_counter()
        dev.release(); p.release()
        if k is not dev: k.release()
        t4 = time.perf_counter()
        times.append(tuple(round(x * 1e3, 3) for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3)))
    f.close()
    print("rep", rep, "(filter, project, write, release) ms per row group:", times)
