"""Tables and writer settings shared by the CPU and GPU tiers of the Parquet-scan tests."""
import io

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq


def sample_table(n: int, seed: int = 0, nulls: bool = False, strings: str = "mixed") -> pa.Table:
    rng = np.random.default_rng(seed)
    m = (lambda p: rng.random(n) < p) if nulls else (lambda p: None)
    if strings == "mixed":
        words = ["", "a", "bb", "hello world", "x" * 40, "é日本", "tail"]
        s = [words[v] if v < len(words) else "u%06d" % v for v in rng.integers(0, 30, n)]
    elif strings == "unique":
        s = ["%09d" % v for v in rng.integers(0, 10**9, n)]
    else:
        s = ["k%d" % v for v in rng.integers(0, 5, n)]
    return pa.table({
        "id": pa.array(np.arange(n, dtype=np.int32)),
        "value1": pa.array(s, type=pa.utf8(), mask=m(0.15)),
        "value2": pa.array((rng.random(n) * 100).astype(np.float32), mask=m(0.05)),
        "i64": pa.array(rng.integers(-2**62, 2**62, n), type=pa.int64(), mask=m(0.3)),
        "f64": pa.array(rng.standard_normal(n), mask=m(0.01)),
        "small": pa.array(rng.integers(0, 7, n).astype(np.int32), mask=m(0.5)),
        "flag": pa.array(rng.random(n) < 0.3, mask=m(0.2)),
    })


def write_bytes(t: pa.Table, **kw) -> bytes:
    kw.setdefault("compression", "none")
    buf = io.BytesIO()
    pq.write_table(t, buf, **kw)
    return buf.getvalue()
