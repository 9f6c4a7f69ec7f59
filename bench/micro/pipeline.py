"""The reference's sample query end to end -- read_files -> filter -> materialize (sample_queries/simple.sql:
`select * from read_files('simple/*.parquet') where value2 > 10.0`, plus a computed select item) -- with every stage on
the device: chq_parquet_read_row_group -> chq_filter_project_record -> chq_record_to_parquet; host memory holds only the
input file and the output files.  Next to it the same stages with pyarrow on the host (pyarrow is NOT the reference -- that
is arrow-rs / the parquet crate -- but it is the Arrow C++ equivalent available in this image).
usage: python bench/micro/pipeline.py [rows]"""
import io
import sys
import time

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.parquet as pq

sys.path.insert(0, ".")
import chapterhouseqe_amd as chq   # noqa: E402
from chapterhouseqe_amd.sqlparse import parse_select   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
rng = np.random.default_rng(0)
letters = rng.integers(ord("a"), ord("z") + 1, (n, 8), dtype=np.uint8)
value1 = pa.Array.from_buffers(pa.utf8(), n, [None, pa.py_buffer((np.arange(n + 1, dtype=np.int32) * 8).tobytes()), pa.py_buffer(letters.tobytes())])
t = pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "value1": value1, "value2": pa.array((rng.random(n) * 100).astype(np.float32))})
buf = io.BytesIO()
pq.write_table(t, buf, compression="none", row_group_size=1 << 20, data_page_size=1 << 20)
raw = buf.getvalue()
sel = parse_select("select id, value1, value2 * 2.0 as twice from t where value2 > 10.0")
print(f"input: {len(raw) / 1e6:.1f} MB, {n} rows")

best = 1e9
for _ in range(2):
    t0 = time.perf_counter()
    f = pq.ParquetFile(io.BytesIO(raw))
    out_bytes = rows = 0
    for g in range(f.metadata.num_row_groups):
        tab = f.read_row_group(g)
        tab = tab.filter(pc.greater(tab["value2"], pa.scalar(10.0, pa.float32())))
        tab = pa.table({"id": tab["id"], "value1": tab["value1"], "twice": pc.multiply(tab["value2"], pa.scalar(2.0, pa.float32()))})
        sink = io.BytesIO(); pq.write_table(tab, sink, compression="none", use_dictionary=False)
        out_bytes += len(sink.getvalue()); rows += tab.num_rows
    best = min(best, time.perf_counter() - t0)
print(f"host (pyarrow read -> filter -> project -> write, one thread pool): {best * 1e3:.0f} ms = {n / best / 1e6:.1f} M input rows/s, kept {rows}, wrote {out_bytes / 1e6:.1f} MB")

ctx = chq.Context(0)
best = 1e9
import gc
gc.collect(); gc.disable()     # (the collector's pauses -- 20+ ms with the tables above alive -- are not the library's)
for it in range(4):
    t0 = time.perf_counter()
    f = chq.ParquetFile(raw)
    out_bytes = rows = 0
    first = None
    ta = time.perf_counter()
    devs = f.read_row_groups(ctx=ctx)           # every row group decoded in one call (their uploads and decodes overlap)
    tb = time.perf_counter()
    stage = [tb - ta, 0.0, 0.0]
    per_call = []
    for dev in devs:
        t1 = time.perf_counter()
        res = chq.filter_project_record(sel.selection, sel.projection, dev, [[], [], []], ctx=ctx)
        t2 = time.perf_counter()
        img = chq.record_to_parquet(res, ctx=ctx, copy=False)      # the file image, in the library's host buffer: a writer
        image = bytes(img.view) if first is None else None         # (opendal in the reference) would take it from there
        nbytes = len(img); img.release()
        stage[1] += t2 - t1; stage[2] += time.perf_counter() - t2
        per_call.append((round((t2 - t1) * 1e3, 2), round((time.perf_counter() - t2) * 1e3, 2)))
        out_bytes += nbytes; rows += res.num_rows
        if first is None:
            first = image
        dev.release(); res.release()
    f.close()
    if time.perf_counter() - t0 < best:
        best = time.perf_counter() - t0; best_stage = stage; best_calls = per_call
    if it == 0:   # parity of the first output file against the host pipeline
        tab = pq.ParquetFile(io.BytesIO(raw)).read_row_group(0)
        tab = tab.filter(pc.greater(tab["value2"], pa.scalar(10.0, pa.float32())))
        exp = pa.table({"id": tab["id"], "value1": tab["value1"], "twice": pc.multiply(tab["value2"], pa.scalar(2.0, pa.float32()))})
        got = pq.read_table(io.BytesIO(first)).combine_chunks()
        assert got.num_rows == exp.num_rows and got.schema.names == exp.schema.names
        for name in exp.schema.names:   # (field nullability differs by design: project_record's rule is null_count > 0)
            assert got[name].combine_chunks().equals(exp[name].combine_chunks()), name
print("  per row group (filter + project, write) ms:", best_calls)
print("  stages: scan %.1f ms, filter + project %.1f ms, write %.1f ms" % tuple(x * 1e3 for x in best_stage))
print(f"device (scan -> filter + project -> write, pages decoded and encoded in HBM): {best * 1e3:.0f} ms = {n / best / 1e6:.1f} M input rows/s, kept {rows}, wrote {out_bytes / 1e6:.1f} MB")
