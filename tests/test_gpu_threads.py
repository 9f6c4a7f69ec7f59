"""GPU: several host threads, one library context each, at the same time -- group calls (host thread pool, pinned tables),
Parquet scans (auxiliary streams) and single-batch filters must not disturb one another (the library is re-entrant across
contexts: INTEGRATION.md section 2)."""
import io
import threading

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_four_contexts_on_four_threads():
    rng = np.random.default_rng(99)
    n_batches, rows = 3000, 2000                      # enough batches for the pooled import / export paths
    ids = np.arange(n_batches * rows, dtype=np.int32)
    v = (rng.random(n_batches * rows) * 100).astype(np.float32)
    words = np.array(["", "a", "bb", "ccc", "dddd", "eeeee"])
    s = words[rng.integers(0, len(words), n_batches * rows)]
    recs = [pa.record_batch({"id": pa.array(ids[b * rows:(b + 1) * rows]), "s": pa.array(s[b * rows:(b + 1) * rows], type=pa.utf8()),
                             "v": pa.array(v[b * rows:(b + 1) * rows])}) for b in range(n_batches)]
    al = [[], [], []]
    e = parse_expr("v > 25.0 and id % 3 = 0")
    expect_rows = [O.filter_record(r, al, e).num_rows for r in recs[:50]]
    table = pa.table({"id": pa.array(ids[:400_000]), "s": pa.array(s[:400_000], type=pa.utf8()), "v": pa.array(v[:400_000])})
    buf = io.BytesIO(); pq.write_table(table, buf, compression="none", row_group_size=50_000); raw = buf.getvalue()
    want_scan = pq.read_table(io.BytesIO(raw)).combine_chunks()
    errors = []

    def worker(k):
        try:
            ctx = chq.Context(0)
            devs = [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs] if k % 2 == 0 else recs
            for rep in range(3):
                outs = chq.filter_records(devs, al, e, ctx=ctx)
                got = [o.num_rows for o in outs[:50]]
                assert got == expect_rows, (k, rep)
                big, per = chq.filter_records_coalesced(devs, al, e, ctx=ctx)
                assert per[:50] == expect_rows and big.num_rows == sum(per)
                scanned = chq.ParquetFile(raw).read_row_groups(ctx=ctx)
                assert pa.Table.from_batches([b.to_host() for b in scanned]).combine_chunks().equals(want_scan)
                one = chq.filter_record(recs[k], al, e, ctx=ctx)
                assert one.num_rows == expect_rows[k]
            ctx.close()
        except BaseException as ex:  # noqa: BLE001
            errors.append((k, repr(ex)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)


def test_a_failed_call_does_not_leak_in_flight_work_into_the_pool():
    """A call that fails after it has queued its uploads (a typing error is found after staging) releases its blocks while
    the copies may still be in flight; the pools are process-wide, so the next call -- on ANOTHER context, i.e. another
    stream -- gets those blocks as its outputs.  Round 3's fuzz met the corruption twice (first output columns of a small
    filter overwritten); blocks released during unwinding now wait for the device (engine.cpp: Buffer::~Buffer)."""
    import numpy as np
    import pyarrow as pa
    from chapterhouseqe_amd.sqlparse import parse_expr
    from oracle import oracle as O
    from .helpers import batches_identical
    a, b = chq.Context(0), chq.Context(0)
    n = 600_000
    rng = np.random.default_rng(12)
    rec = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 1000, n).astype(np.int64)), pa.array(rng.random(n)),
                                      pa.array(rng.integers(0, 1000, n).astype(np.int64))], names=["x", "y", "z"])
    other = pa.RecordBatch.from_arrays([pa.array(np.arange(n, dtype=np.int64)), pa.array(np.arange(n, dtype=np.float64)),
                                        pa.array(np.arange(n, dtype=np.int64) * 3)], names=["x", "y", "z"])
    al = [[], [], []]
    bad, good = parse_expr("x + 1.5 > y"), parse_expr("x >= 0")        # Int64 + Float32: no common type (status 9); keeps every row
    want = O.filter_record(other, al, good)
    for it in range(150):
        with pytest.raises(chq.ChqError) as e:
            chq.filter_record(rec, al, bad, ctx=a)
        assert e.value.code == 9
        got = chq.filter_record(other, al, good, ctx=b)
        assert batches_identical(got, want), f"iteration {it}"
    a.close(); b.close()
