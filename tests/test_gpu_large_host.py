"""GPU: one large host-resident batch through chq_filter_record -- cut into chunks whose uploads and downloads overlap
(engine.cpp: filter_record_large_host) -- against the oracle and against the unchunked path: same rows, same order, same
status codes (a data-dependent error in a late chunk included)."""
import numpy as np
import pyarrow as pa
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    c = chq.Context(0)
    c.set_option("large_host_rows", 280_000)
    c.set_option("large_host_chunk", 65_536)       # many chunks at test sizes
    yield c
    c.close()


def table(n, seed):
    rng = np.random.default_rng(seed)
    return pa.record_batch({"id": pa.array(np.arange(n, dtype=np.int32)), "value1": pa.array((rng.random(n) * 100).astype(np.float32)),
                            "value2": pa.array((rng.random(n) * 100).astype(np.float32)), "k": pa.array(rng.integers(-5, 5, n).astype(np.int64)),
                            "h": pa.array(rng.integers(0, 30000, n).astype(np.int16))})


@pytest.mark.parametrize("n", [300_000, 327_680, 1_000_001])
def test_chunked_host_batches_match_the_oracle_and_the_unchunked_path(ctx, n):
    rec = table(n, n)
    al = chq.get_record_table_aliases(None, rec)
    for where in ["value2 > 10.0", "value2 > 99.9", "id % 7 = 0 and k > 0", "value1 < 0.0", "h > 15000 or value2 < 1.0"]:
        e = parse_expr(where)
        got = chq.filter_record(rec, al, e, ctx=ctx)
        launches = ctx.last_stats()["launches"]
        assert launches >= n // 65_536                     # it really ran chunk by chunk
        assert got.equals(O.filter_record(rec, al, e)), where
        ctx.set_option("large_host", 0)
        try:
            assert got.equals(chq.filter_record(rec, al, e, ctx=ctx))
        finally:
            ctx.set_option("large_host", 1)


def test_an_error_in_a_late_chunk_is_the_reference_error(ctx):
    n = 400_000
    a = np.ones(n, dtype=np.int32)
    a[n - 1000] = np.iinfo(np.int32).max                    # a + a overflows in the last chunk only
    rec = pa.record_batch({"a": pa.array(a), "v": pa.array(np.arange(n, dtype=np.float32))})
    al = chq.get_record_table_aliases(None, rec)
    e = parse_expr("a + a > 1")
    with pytest.raises(chq.ChqError) as got:
        chq.filter_record(rec, al, e, ctx=ctx)
    with pytest.raises(O.OracleError) as exp:
        O.filter_record(rec, al, e)
    assert got.value.code == exp.value.code


def test_columns_the_chunked_path_does_not_take_fall_back(ctx):
    n = 200_000
    rng = np.random.default_rng(3)
    rec = pa.record_batch({"id": pa.array(np.arange(n, dtype=np.int32)), "s": pa.array(["%04d" % v for v in rng.integers(0, 9999, n)]),
                           "o": pa.array(rng.random(n), mask=rng.random(n) < 0.1)})
    al = chq.get_record_table_aliases(None, rec)
    e = parse_expr("id % 3 = 0")
    assert chq.filter_record(rec, al, e, ctx=ctx).equals(O.filter_record(rec, al, e))
