"""Parquet scan throughput: the reference's sample shape (id:Int32, value1:Utf8(8), value2:Float32,
create_sample_data.rs) written with the reference's writer settings (uncompressed, dictionary with PLAIN fallback, V1 pages,
1 Mi-row row groups), decoded (a) by pyarrow on the host CPU -- what read_files does today through the parquet crate --
and (b) by chq.scan_parquet: column chunks uploaded as they lie in the file, pages decoded in HBM.
usage: python bench/micro/parquet_scan.py [rows] [compression: none | snappy] [shape: sample | compressible]"""
import io
import sys
import time

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

sys.path.insert(0, ".")
import chapterhouseqe_amd as chq   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
codec = sys.argv[2] if len(sys.argv) > 2 else "none"
shape = sys.argv[3] if len(sys.argv) > 3 else "sample"
rng = np.random.default_rng(0)
letters = rng.integers(ord("a"), ord("z") + 1, (n, 8), dtype=np.uint8)
if shape == "compressible":   # few distinct letters per position, slowly changing: snappy finds matches (the sample data is random: it cannot)
    letters = (ord("a") + ((np.arange(n)[:, None] // np.array([1 << 14, 1 << 12, 1 << 10, 1 << 8, 64, 16, 4, 1])) % 4)).astype(np.uint8)
value1 = pa.Array.from_buffers(pa.utf8(), n, [None, pa.py_buffer((np.arange(n + 1, dtype=np.int32) * 8).tobytes()), pa.py_buffer(letters.tobytes())])
v2 = (rng.random(n) * 100).astype(np.float32)
if shape == "compressible":
    v2 = np.round(v2)   # 101 distinct values
t = pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "value1": value1, "value2": pa.array(v2)})
buf = io.BytesIO()
pq.write_table(t, buf, compression=codec, row_group_size=1 << 20, data_page_size=1 << 20, dictionary_pagesize_limit=1 << 20,
               use_dictionary=shape != "compressible")
print(f"compression {codec}, shape {shape}")
raw = buf.getvalue()
print(f"file: {len(raw) / 1e6:.1f} MB, {n} rows, {pq.ParquetFile(io.BytesIO(raw)).metadata.num_row_groups} row groups")

for threads in (False, True):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        tab = pq.read_table(io.BytesIO(raw), use_threads=threads)
        best = min(best, time.perf_counter() - t0)
    print(f"pyarrow read_table use_threads={threads}: {best * 1e3:.1f} ms = {len(raw) / best / 1e9:.2f} GB/s of file bytes, {n / best / 1e6:.1f} M rows/s")

ctx = chq.Context(0)
ctx.set_option("time_kernels", 0)
f = chq.ParquetFile(raw)
best, best_open = 1e9, 1e9
for it in range(4):
    t0 = time.perf_counter()
    g = chq.ParquetFile(raw)
    t1 = time.perf_counter()
    outs = g.read_row_groups(ctx=ctx)        # all row groups in flight together (one by one: see one_by_one below)
    t2 = time.perf_counter()
    rows = sum(o.num_rows for o in outs)
    assert rows == n
    if it == 0:   # parity of the first and last row group against pyarrow
        ref = pq.ParquetFile(io.BytesIO(raw))
        for i in (0, g.num_row_groups - 1):
            assert outs[i].to_host().equals(ref.read_row_group(i).to_batches()[0])
    for o in outs:
        o.release()
    g.close()
    best = min(best, t2 - t1); best_open = min(best_open, t1 - t0)
print(f"chq scan (metadata {best_open * 1e3:.2f} ms + upload and GPU decode of every row group, one call): {best * 1e3:.1f} ms = {len(raw) / best / 1e9:.2f} GB/s of file bytes, {n / best / 1e6:.1f} M rows/s")
one_by_one = 1e9
for it in range(3):
    g = chq.ParquetFile(raw)
    t1 = time.perf_counter()
    outs = [g.read_row_group(i, ctx=ctx) for i in range(g.num_row_groups)]
    one_by_one = min(one_by_one, time.perf_counter() - t1)
    for o in outs:
        o.release()
    g.close()
print(f"chq scan, one call per row group: {one_by_one * 1e3:.1f} ms")

# ---- f-4: the same table written back: pyarrow's writer on the host vs pages encoded on the GPU ---------------------------
m = min(n, 4_000_000)     # one row group / one page per column: keep every page below 2 GiB
rec = t.slice(0, m).to_batches()[0]
best = 1e9
for _ in range(3):
    sink = io.BytesIO()
    t0 = time.perf_counter()
    pq.write_table(pa.Table.from_batches([rec]), sink, compression="none")
    best = min(best, time.perf_counter() - t0)
print(f"pyarrow write_table ({m} rows, uncompressed, dictionary on): {best * 1e3:.1f} ms = {m / best / 1e6:.1f} M rows/s, {len(sink.getvalue()) / 1e6:.1f} MB")
best = 1e9
for _ in range(3):
    sink = io.BytesIO()
    t0 = time.perf_counter()
    pq.write_table(pa.Table.from_batches([rec]), sink, compression="none", use_dictionary=False)
    best = min(best, time.perf_counter() - t0)
print(f"pyarrow write_table ({m} rows, uncompressed, PLAIN): {best * 1e3:.1f} ms = {m / best / 1e6:.1f} M rows/s")
dev = chq.DeviceRecordBatch.from_host(rec, ctx=ctx)
best = 1e9
for it in range(4):
    t0 = time.perf_counter()
    image = chq.record_to_parquet(dev, ctx=ctx, copy=False)
    best = min(best, time.perf_counter() - t0)
    if it == 0:
        assert pq.read_table(io.BytesIO(bytes(image.view))).combine_chunks().equals(pa.Table.from_batches([rec]))
    size = len(image)
    image.release()
print(f"chq record_to_parquet from a device batch (file image in the library's host buffer): {best * 1e3:.1f} ms = {m / best / 1e6:.1f} M rows/s, {size / 1e6:.1f} MB")
