"""GPU: Utf8 columns filtered inside filter_fused_kernel (device_program.h: Utf8Fold) against the CPU oracle and against the
separate Utf8 pass (`fold_utf8` = 0) -- bit-exact offsets, bytes, validity and null counts, for every tile kind the fold
serves, around tile / wave / 64-row-group boundaries, with empty, short, long and mixed strings, slices and nulls."""
import numpy as np
import pyarrow as pa
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = chq.Context(0)
    yield c
    c.close()


def strings(rng, n, kind):
    if kind == "fixed8":
        return ["%08d" % v for v in rng.integers(0, 10**8, n)]
    if kind == "ragged":   # 0..23 bytes: lengths that are not multiples of four, empty strings
        return ["x" * int(l) + str(i % 10) * (int(l) > 0) for i, l in enumerate(rng.integers(0, 23, n))]
    if kind == "mixed":    # mostly short, every ~40th row long: some 64-row groups take the wide copy, the average stays short
        out = []
        for i, l in enumerate(rng.integers(0, 12, n)):
            out.append(("L%05d-" % i) * 30 if rng.random() < 0.025 else "s" * int(l))
        return out
    if kind == "multibyte":
        words = ["", "é", "日本語", "aß", "🙂🙂", "plain"]
        return [words[v] for v in rng.integers(0, len(words), n)]
    raise AssertionError(kind)


def batch(rng, n, kind, nulls=False, extra_string_columns=0):
    mask = (rng.random(n) < 0.15) if nulls else None
    cols = {
        "id": pa.array(np.arange(n, dtype=np.int32)),
        "s": pa.array(strings(rng, n, kind), type=pa.utf8(), mask=mask),
        "v": pa.array(rng.random(n).astype(np.float32) * 100),
    }
    for k in range(extra_string_columns):
        cols[f"t{k}"] = pa.array(strings(rng, n, "ragged"), type=pa.utf8(), mask=(rng.random(n) < 0.1) if k % 2 else None)
    return pa.record_batch(cols)


def same(a: pa.RecordBatch, b: pa.RecordBatch):
    assert a.schema.names == b.schema.names and a.num_rows == b.num_rows
    for i in range(a.num_columns):
        x, y = a.column(i), b.column(i)
        assert x.null_count == y.null_count, a.schema.names[i]
        assert x.equals(y), a.schema.names[i]
        if pa.types.is_string(x.type) and len(x):   # the raw offsets too (rebased to 0 by both)
            ox = np.frombuffer(x.buffers()[1], dtype=np.int32, count=len(x) + 1, offset=4 * x.offset)
            oy = np.frombuffer(y.buffers()[1], dtype=np.int32, count=len(y) + 1, offset=4 * y.offset)
            assert np.array_equal(ox - ox[0], oy - oy[0])


SIZES = [1, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 16383, 16384, 16385, 40_000, 300_001]


@pytest.mark.parametrize("kind", ["fixed8", "ragged", "mixed", "multibyte"])
@pytest.mark.parametrize("tile_kind", [-1, 0, 1])
def test_fold_matches_oracle_and_the_separate_pass(ctx, kind, tile_kind):
    rng = np.random.default_rng(hash((kind, tile_kind)) & 0xFFFF)
    ctx.set_option("tile_kind", tile_kind)
    try:
        for n in SIZES:
            rec = batch(rng, n, kind, nulls=(n % 2 == 1))
            al = chq.get_record_table_aliases(None, rec)
            for where in ["id % 2 = 0", "v > 10.0", "v > 99.5", "id >= 0", "v < 0.0"]:
                e = parse_expr(where)
                exp = O.filter_record(rec, al, e)
                ctx.set_option("fold_utf8", 1)
                got = chq.filter_record(rec, al, e, ctx=ctx)
                launches = ctx.last_stats()["launches"]
                ctx.set_option("fold_utf8", 0)
                ref = chq.filter_record(rec, al, e, ctx=ctx)
                assert ctx.last_stats()["launches"] > launches or n == 0     # the fold really took the Utf8 pass away
                same(got, exp)
                same(got, ref)
    finally:
        ctx.set_option("tile_kind", -1)
        ctx.set_option("fold_utf8", 1)


def test_more_string_columns_than_the_fold_takes(ctx):
    """two Utf8 columns ride in the main kernel, the others take the separate pass -- in the same call"""
    rng = np.random.default_rng(5)
    for n in [100, 5000, 70_000]:
        rec = batch(rng, n, "ragged", nulls=True, extra_string_columns=3)
        al = chq.get_record_table_aliases(None, rec)
        for where in ["id % 3 = 0", "v > 50.0 and id > 10"]:
            same(chq.filter_record(rec, al, parse_expr(where), ctx=ctx), O.filter_record(rec, al, parse_expr(where)))


def test_sliced_string_columns(ctx):
    rng = np.random.default_rng(6)
    rec = batch(rng, 50_000, "mixed", nulls=True)
    for off, ln in [(1, 100), (7, 20_000), (63, 33_000), (4097, 40_000)]:
        sl = rec.slice(off, ln)
        al = chq.get_record_table_aliases(None, sl)
        e = parse_expr("v > 30.0")
        same(chq.filter_record(sl, al, e, ctx=ctx), O.filter_record(sl, al, e))


def test_utf8_predicate_with_folded_copy(ctx):
    """the predicate itself reads the string column that the same kernel then copies"""
    rng = np.random.default_rng(7)
    rec = batch(rng, 30_000, "ragged")
    al = chq.get_record_table_aliases(None, rec)
    for where in ["s >= 'xxxx'", "s = ''", "s < 'xx' or v > 90.0"]:
        same(chq.filter_record(rec, al, parse_expr(where), ctx=ctx), O.filter_record(rec, al, parse_expr(where)))


def test_long_strings_keep_the_separate_pass(ctx):
    """an average above 24 bytes per row stays with the offsets + wave-per-group copy kernels"""
    rng = np.random.default_rng(8)
    n = 20_000
    rec = pa.record_batch({"id": pa.array(np.arange(n, dtype=np.int32)), "s": pa.array(["w" * 100] * n, type=pa.utf8())})
    al = chq.get_record_table_aliases(None, rec)
    e = parse_expr("id % 2 = 1")
    got = chq.filter_record(rec, al, e, ctx=ctx)
    assert ctx.last_stats()["launches"] >= 2
    same(got, O.filter_record(rec, al, e))


def test_device_resident_column_of_unknown_span(ctx):
    """a caller's device pointers carry no byte span: one small read-back sizes the output, then the same single launch"""
    import torch
    dev = torch.device("cuda", 0)
    n, L = 1_000_000, 8
    g = torch.Generator(device=dev); g.manual_seed(9)
    ids = torch.arange(n, dtype=torch.int32, device=dev)
    chars = torch.randint(ord("a"), ord("z") + 1, (n * L,), dtype=torch.uint8, device=dev, generator=g)
    offs = (torch.arange(n + 1, dtype=torch.int64, device=dev) * L).to(torch.int32)
    c = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    rec = chq.DeviceRecordBatch.from_device_pointers([("id", "i", ids.data_ptr()), ("s", "u", offs.data_ptr(), chars.data_ptr())], n, ctx=c)
    out = chq.filter_record(rec, [[], []], parse_expr("id % 2 = 0"), ctx=c)
    assert out.num_rows == n // 2 and c.last_stats()["launches"] <= 2     # main kernel (+ tail tile)
    host = out.to_host()
    exp = chars.view(n, L)[::2].cpu().numpy()
    got = np.frombuffer(host.column(1).buffers()[2], dtype=np.uint8, count=(n // 2) * L).reshape(n // 2, L)
    assert np.array_equal(got, exp)
    assert np.array_equal(host.column(0).to_numpy(), np.arange(0, n, 2, dtype=np.int32))
    c.close()


def test_uniform_length_strings_take_the_fixed_width_copy():
    """Utf8 columns whose values all have one length (1, 2, 4, 8 or 16 bytes) are filtered as fixed-width columns once a pass
    over the offsets has proved it; anything else -- one ragged value, an unsupported length, nulls, a predicate that reads
    the column -- keeps the ordinary string paths.  Same answers either way (oracle), slices included."""
    import numpy as np
    import pyarrow as pa
    from chapterhouseqe_amd.sqlparse import parse_expr
    from oracle import oracle as O
    from .helpers import batches_identical, explain_diff
    c = chq.Context(0)
    c.set_option("uniform_utf8_rows", 1000)    # (default: batches of 2^24 rows and more)
    rng = np.random.default_rng(31)
    n = 70_001

    def fixed(width, seed):
        r = np.random.default_rng(seed)
        return pa.array(["".join(chr(97 + int(x)) for x in row) for row in r.integers(0, 26, (n, width))], pa.utf8())

    ragged = pa.array(["r" * int(l) for l in rng.integers(0, 12, n)], pa.utf8())
    almost = fixed(8, 5).to_pylist(); almost[n - 7] = "short"
    with_nulls = pa.array(fixed(8, 6).to_pylist(), pa.utf8(), mask=rng.random(n) < 0.1)
    rec = pa.RecordBatch.from_arrays(
        [pa.array(np.arange(n, dtype=np.int32)), fixed(8, 1), fixed(16, 2), fixed(1, 3), fixed(3, 4), ragged, pa.array(almost, pa.utf8()), with_nulls,
         pa.array((rng.random(n) * 100).astype(np.float32)), pa.array(["é%06d" % i for i in range(n)], pa.utf8())],   # 8 BYTES each: 2 + 6
        names=["id", "k8", "k16", "k1", "k3", "rag", "almost", "nul", "v", "utf8bytes"])
    al = [[] for _ in range(rec.num_columns)]
    for sl in (rec, rec.slice(13, 50_000), rec.slice(69_000)):
        dev = chq.DeviceRecordBatch.from_host(sl, c)
        for sql in ["id % 2 = 0", "v > 10.0", "v > 99.5", "id < 0", "k8 >= 'n' and v < 50.0", "k16 < 'c'"]:
            e = parse_expr(sql)
            want = O.filter_record(sl, al, e)
            got = chq.filter_record(dev, al, e, ctx=c).to_host()
            assert batches_identical(got, want), f"{sql} ({sl.num_rows} rows):\n{explain_diff(got, want)}"
            c.set_option("uniform_utf8_rows", 0)
            ref = chq.filter_record(dev, al, e, ctx=c).to_host()
            c.set_option("uniform_utf8_rows", 1000)
            assert batches_identical(got, ref), sql
    c.close()
