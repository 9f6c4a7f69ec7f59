"""Which pages cost what in the snappy inflate: one single-column file per kind of data, one row group, scanned once
(run under rocprofv3 --kernel-trace: the three pq_inflate_kernel launches per file are index / blocks / finish).
usage: python bench/micro/snappy_pages.py"""
import io
import sys

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

sys.path.insert(0, ".")
import chapterhouseqe_amd as chq   # noqa: E402

n = 1 << 20
rng = np.random.default_rng(0)
letters = rng.integers(ord("a"), ord("z") + 1, (n, 8), dtype=np.uint8)
value1 = pa.Array.from_buffers(pa.utf8(), n, [None, pa.py_buffer((np.arange(n + 1, dtype=np.int32) * 8).tobytes()), pa.py_buffer(letters.tobytes())])
cols = {
    "strings_plain": (value1, False),
    "strings_dict": (value1, True),
    "id_plain": (pa.array(np.arange(n, dtype=np.int32)), False),
    "id_dict": (pa.array(np.arange(n, dtype=np.int32)), True),
    "floats_plain": (pa.array((rng.random(n) * 100).astype(np.float32)), False),
    "floats_dict": (pa.array((rng.random(n) * 100).astype(np.float32)), True),
}
ctx = chq.Context(0)
for name, (arr, dic) in cols.items():
    buf = io.BytesIO()
    pq.write_table(pa.table({"c": arr}), buf, compression="snappy", row_group_size=n, data_page_size=1 << 20, dictionary_pagesize_limit=1 << 20, use_dictionary=dic)
    raw = buf.getvalue()
    f = chq.ParquetFile(raw)
    out = f.read_row_groups(ctx=ctx)
    assert out[0].num_rows == n
    for o in out:
        o.release()
    f.close()
    print(name, len(raw))
