"""Parquet scan of PLAIN BYTE_ARRAY pages with RAGGED strings (the serial case of the length walk): uncompressed, no
dictionary, lengths uniform in [0, 20].  usage: python bench/micro/parquet_ragged.py [rows]"""
import io
import sys
import time

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

sys.path.insert(0, ".")
import chapterhouseqe_amd as chq   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
rng = np.random.default_rng(1)
lens = rng.integers(0, 21, n)
offs = np.zeros(n + 1, dtype=np.int32); np.cumsum(lens, out=offs[1:])
data = rng.integers(ord("a"), ord("z") + 1, int(offs[-1]), dtype=np.uint8)
s = pa.Array.from_buffers(pa.utf8(), n, [None, pa.py_buffer(offs.tobytes()), pa.py_buffer(data.tobytes())])
t = pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "s": s})
for use_dict in (False, True):
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="none", row_group_size=1 << 20, use_dictionary=use_dict)
    raw = buf.getvalue()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); pq.read_table(io.BytesIO(raw)); best = min(best, time.perf_counter() - t0)
    ctx = chq.Context(0)
    mine = 1e9
    for it in range(4):
        f = chq.ParquetFile(raw)
        t0 = time.perf_counter()
        outs = f.read_row_groups(ctx=ctx)
        mine = min(mine, time.perf_counter() - t0)
        if it == 0:
            assert outs[0].to_host().equals(pq.ParquetFile(io.BytesIO(raw)).read_row_group(0).to_batches()[0])
        for o in outs:
            o.release()
        f.close()
    print(f"ragged strings, use_dictionary={use_dict}: file {len(raw) / 1e6:.0f} MB, pyarrow {best * 1e3:.0f} ms, chq scan {mine * 1e3:.1f} ms = {len(raw) / mine / 1e9:.1f} GB/s")
    ctx.close()
