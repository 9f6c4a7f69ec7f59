"""GPU: chq_filter_records (one launch for a group of same-schema batches) against the CPU oracle and against
chq_filter_record called batch by batch -- the loop of filter_task.rs:78-126 that the group call replaces.
Bit-exact: the group call only moves values."""
import numpy as np
import pyarrow as pa
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr
from oracle import oracle as O

from .cases import empty_aliases
from .test_gpu_scale import _dtod
from .helpers import batches_identical, explain_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = chq.Context(0)
    yield c
    c.close()


def fixed_batch(n, seed, with_wide=True):
    rng = np.random.default_rng(seed)
    cols = {
        "id": pa.array(rng.integers(-10**6, 10**6, n).astype(np.int32)),
        "value1": pa.array((rng.random(n) * 40 - 10).astype(np.float32)),
        "value2": pa.array((rng.random(n) * 40 - 10).astype(np.float32)),
        "b": pa.array(rng.integers(-128, 128, n).astype(np.int8)),
        "h": pa.array(rng.integers(0, 65536, n).astype(np.uint16)),
    }
    if with_wide:
        cols["big"] = pa.array(rng.integers(-10**12, 10**12, n).astype(np.int64))
        cols["d"] = pa.array(rng.random(n) * 100)
    return pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys()))


def check_group(ctx, recs, sql, device):
    al = empty_aliases(recs[0])
    e = parse_expr(sql)
    exp = [O.filter_record(r, al, e) for r in recs]
    if device:
        devs = [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs]
        got = [g.to_host() for g in chq.filter_records(devs, al, e, ctx=ctx)]
    else:
        got = chq.filter_records(recs, al, e, ctx=ctx)
    assert len(got) == len(exp)
    for i, (g, x) in enumerate(zip(got, exp)):
        assert batches_identical(g, x), f"{sql}: batch {i} of {len(recs)} (rows {recs[i].num_rows}):\n{explain_diff(g, x)}"
    return ctx.last_stats()


RAGGED = [2, 3, 63, 64, 65, 2047, 2048, 2049, 10_000, 16_383, 16_384, 16_385, 40_000, 5]
PREDICATES = ["value2 > 10.0", "id % 2 = 0 and value1 < 20.0", "big / 3 > h or d * 2.0 < 50.0", "b < 0", "id = id", "id <> id"]


@pytest.mark.parametrize("device", [True, False], ids=["device", "host"])
@pytest.mark.parametrize("group_mode", [0, 1, 2], ids=["auto", "tile-table", "wave-packed"])
@pytest.mark.parametrize("tile_kind", [-1, 0, 1])
def test_group_matches_per_batch_results(tile_kind, group_mode, device):
    """ragged batch sizes: auto picks the per-tile table (one launch + the per-batch prefix gather); the wave-packed
    layout is forced as well (correct on any group, only wasteful on ragged ones)"""
    c = chq.Context(0)
    c.set_option("tile_kind", tile_kind)
    c.set_option("group_mode", group_mode)
    recs = [fixed_batch(n, 100 + i) for i, n in enumerate(RAGGED)]
    for sql in PREDICATES:
        st = check_group(c, recs, sql, device)
        assert st["launches"] == (1 if group_mode == 2 else 2), (sql, st["launches"])
        assert st["rows_in"] == sum(RAGGED)
    c.close()


@pytest.mark.parametrize("group_mode", [0, 1], ids=["auto", "tile-table"])
@pytest.mark.parametrize("rows", [10_000, 1024, 1025, 513, 9_216, 33_000])
def test_near_uniform_groups_are_wave_packed(rows, group_mode):
    """same-sized batches plus shorter tails (the last batch of a file): one launch, no idle lanes behind batch ends"""
    c = chq.Context(0)
    c.set_option("group_mode", group_mode)
    sizes = [rows] * 9 + [max(2, rows // 3), rows, max(2, rows - 1), 2]
    recs = [fixed_batch(n, 300 + i) for i, n in enumerate(sizes)]
    for sql in PREDICATES[:4]:
        st = check_group(c, recs, sql, True)
        if group_mode == 0 and rows >= 9_216:
            assert st["launches"] == 1, (sql, st)
    c.close()


def test_reference_sized_batches_in_one_launch(ctx):
    """the reference's planner emits 10 000-row batches (physical_planner.rs:323)"""
    recs = [fixed_batch(10_000, 7 + i, with_wide=False) for i in range(40)]
    st = check_group(ctx, recs, "value2 > 10.0", True)
    assert st["launches"] == 1 and st["rows_in"] == 400_000
    # the outputs are slices of one dense buffer per column: consecutive batches are adjacent in HBM
    devs = [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs[:3]]
    outs = chq.filter_records(devs, empty_aliases(recs[0]), parse_expr("value2 > 10.0"), ctx=ctx)
    a0, a1 = outs[0].column_buffer_address(0), outs[1].column_buffer_address(0)
    assert a1 - a0 == 4 * outs[0].num_rows


def test_group_result_outlives_its_siblings(ctx):
    """every output batch owns a share of the dense buffers: releasing some must not disturb the others"""
    recs = [fixed_batch(5000, 900 + i) for i in range(6)]
    al = empty_aliases(recs[0])
    e = parse_expr("value1 > 0.0")
    devs = [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs]
    outs = chq.filter_records(devs, al, e, ctx=ctx)
    keep = outs[4]
    for i, o in enumerate(outs):
        if i != 4:
            o.release()
    del outs
    junk = [chq.DeviceRecordBatch.from_host(fixed_batch(5000, 5), ctx) for _ in range(4)]   # would reuse freed blocks
    assert batches_identical(keep.to_host(), O.filter_record(recs[4], al, e))
    del junk


def mixed_batch(n, seed, nulls):
    r = np.random.default_rng(seed)
    m = (lambda p: (r.random(n) < p)) if nulls and n else (lambda p: None)
    words = np.array(["", "a", "ab", "zeta", "a much longer string value that spans more than sixty-four bytes of utf8 text ..."])
    return pa.RecordBatch.from_arrays(
        [pa.array(r.integers(0, 100, n).astype(np.int32), mask=m(0.2)),
         pa.array(words[r.integers(0, len(words), n)] if n else np.array([], dtype=object), type=pa.utf8(), mask=m(0.1)),
         pa.array(r.integers(0, 2, n).astype(bool), mask=m(0.15)),
         pa.array(r.random(n) * 10)], names=["a", "s", "f", "d"])


@pytest.mark.parametrize("device", [True, False], ids=["device", "host"])
def test_groups_outside_the_one_launch_path(ctx, device):
    """Utf8 / Boolean / nullable columns, 0- and 1-row batches, literal-only predicates: same results.  Such groups are
    joined -- on the host while staging, on the GPU when they live in HBM (next tests) -- or run batch by batch."""
    groups = [
        ([mixed_batch(n, 40 + n, False) for n in (100, 3000, 17)], "a > 50 and s <> 'ab'"),
        ([mixed_batch(n, 50 + n, True) for n in (100, 3000, 17)], "a > 50 or f"),
        ([fixed_batch(n, 60 + n) for n in (100, 0, 1, 5000)], "value2 > 10.0"),
        ([fixed_batch(n, 70 + n) for n in (100, 200)], "1 = 1"),
    ]
    for recs, sql in groups:
        check_group(ctx, recs, sql, device)


@pytest.mark.parametrize("chunk_bytes", [1 << 30, 20_000], ids=["one-chunk", "many-chunks"])
def test_host_groups_with_strings_booleans_and_nulls_are_concatenated(chunk_bytes):
    """the reference's sample tables carry a Utf8 column: 150 small host batches (some sliced, so every buffer carries
    an Arrow offset) are staged as ONE batch, filtered by a handful of launches and sliced per record id"""
    c = chq.Context(0)
    c.set_option("group_chunk_bytes", chunk_bytes)
    sizes = [int(x) for x in np.random.default_rng(1).integers(2, 900, 150)]
    recs = []
    for i, n in enumerate(sizes):
        b = mixed_batch(n + 11, 2000 + i, nulls=(i % 3 != 0))
        recs.append(b.slice(5 + i % 7, n) if i % 2 else b.slice(0, n))
    for sql in ["a > 50 and s <> 'ab'", "f or d * 2.0 > 15.0", "s >= 'ab'", "a % 7 = 0", "a = a", "d < 0.0"]:
        st = check_group(c, recs, sql, device=False)
        if chunk_bytes == 1 << 30:
            assert st["launches"] <= 12, (sql, st)          # not 150 x (main + follow-up kernels)
    # an error in one batch: the whole call fails with that batch's error, nothing is returned
    bad = list(recs)
    bad[77] = pa.RecordBatch.from_arrays([pa.array(np.full(20, 2**31 - 1, dtype=np.int32)), pa.array(["x"] * 20, type=pa.utf8()),
                                          pa.array([True] * 20), pa.array(np.zeros(20))], names=["a", "s", "f", "d"])
    with pytest.raises(chq.ChqError) as ei:
        chq.filter_records(bad, empty_aliases(bad[0]), parse_expr("a + 1 > 0"), ctx=c)
    assert ei.value.code == 20
    c.close()


@pytest.mark.parametrize("chunk_bytes", [1 << 30, 20_000], ids=["one-chunk", "many-chunks"])
def test_device_groups_with_strings_booleans_and_nulls_are_joined_on_the_gpu(chunk_bytes):
    """the same group resident in HBM (what a GPU scan / a previous GPU operator hands over): joined by the concat kernels,
    filtered as ONE batch, cut per record id -- a handful of launches, not 150 x (main + follow-up kernels)"""
    c = chq.Context(0)
    c.set_option("group_chunk_bytes", chunk_bytes)
    sizes = [int(x) for x in np.random.default_rng(2).integers(2, 900, 150)]
    recs = []
    for i, n in enumerate(sizes):
        b = mixed_batch(n + 11, 3000 + i, nulls=(i % 3 != 0))
        recs.append(b.slice(5 + i % 7, n) if i % 2 else b.slice(0, n))
    al = empty_aliases(recs[0])
    devs = [chq.DeviceRecordBatch.from_host(r, c) for r in recs]
    for sql in ["a > 50 and s <> 'ab'", "f or d * 2.0 > 15.0", "s >= 'ab'", "a % 7 = 0", "a = a", "d < 0.0"]:
        e = parse_expr(sql)
        exp = [O.filter_record(r, al, e) for r in recs]
        for out_host in (False, True):
            got = chq.filter_records(devs, al, e, ctx=c, device_result=not out_host)
            st = c.last_stats()
            assert len(got) == len(exp)
            for i, (g, x) in enumerate(zip(got, exp)):
                g = g.to_host() if hasattr(g, "to_host") else g
                assert batches_identical(g, x), f"{sql}: batch {i} ({recs[i].num_rows} rows):\n{explain_diff(g, x)}"
            if chunk_bytes == 1 << 30:
                assert st["launches"] <= 12, (sql, st)
    if chunk_bytes == 1 << 30:   # the joined form: one output batch + rows per input batch
        e = parse_expr("a > 50 or f")
        parts = [O.filter_record(r, al, e) for r in recs]
        whole = pa.Table.from_batches(parts).combine_chunks().to_batches()[0]
        got, rows = chq.filter_records_coalesced(devs, al, e, ctx=c)
        assert rows == [p.num_rows for p in parts]
        assert batches_identical(got.to_host(), whole, check_nullable=False), explain_diff(got.to_host(), whole)
    bad = list(recs)
    bad[77] = pa.RecordBatch.from_arrays([pa.array(np.full(20, 2**31 - 1, dtype=np.int32)), pa.array(["x"] * 20, type=pa.utf8()),
                                          pa.array([True] * 20), pa.array(np.zeros(20))], names=["a", "s", "f", "d"])
    bad_dev = [chq.DeviceRecordBatch.from_host(r, c) for r in bad]
    with pytest.raises(chq.ChqError) as ei:
        chq.filter_records(bad_dev, al, parse_expr("a + 1 > 0"), ctx=c)
    assert ei.value.code == 20
    c.close()


def test_reference_schema_group_resident_in_hbm(ctx):
    """the reference's own schema (id:Int32, value1:Utf8(8), value2:Float32, create_sample_data.rs:157-204) in its own batch
    size (10 000 rows, physical_planner.rs:323), 300 batches resident in HBM, the sample queries' predicates"""
    from chapterhouseqe_amd.sample_data import simple_batches
    recs = simple_batches(3_000_000, 8, 10_000)
    al = empty_aliases(recs[0])
    devs = [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs]
    for sql in ["id % 2 = 0", "value2 > 10.0", "value1 >= 'n' and id > 25"]:
        e = parse_expr(sql)
        got = chq.filter_records(devs, al, e, ctx=ctx)
        assert ctx.last_stats()["launches"] <= 8
        for i in (0, 1, 150, 299):
            assert batches_identical(got[i].to_host(), O.filter_record(recs[i], al, e)), (sql, i)
        rows = [g.num_rows for g in got]
        big, per = chq.filter_records_coalesced(devs, al, e, ctx=ctx)
        assert per == rows and big.num_rows == sum(rows)
        exp0 = O.filter_record(recs[0], al, e)
        head = big.to_host().slice(0, exp0.num_rows)
        assert batches_identical(head, exp0, check_nullable=False), sql


def test_a_single_null_stays_on_the_one_launch_path(ctx):
    """round 3: a device-resident, near-uniform group with validity bitmaps (here: ONE null in one batch) is filtered by the
    same single launch -- the bitmaps are compacted behind it -- instead of being joined first; host groups still concatenate"""
    recs = [fixed_batch(4000, 80 + i, with_wide=False) for i in range(4)]
    v = recs[2].column(1).to_numpy().copy()
    mask = np.zeros(len(v), dtype=bool)
    mask[1234] = True
    cols = list(recs[2].columns)
    cols[1] = pa.array(v, mask=mask)
    recs[2] = pa.RecordBatch.from_arrays(cols, names=recs[2].schema.names)
    for device in (True, False):
        check_group(ctx, recs, "value1 > 5.0", device)
    devs = [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs]
    chq.filter_records(devs, empty_aliases(recs[0]), parse_expr("value1 > 5.0"), ctx=ctx)
    st = ctx.last_stats()
    assert st["tiles"] < 64 and st["launches"] == 2, st      # the group launch + one bitmap compaction (value1's validity)


def nullable_group(nb, rows, seed, null_share=0.2):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(nb):
        n = rows - int(rng.integers(0, 40))

        def m(p):
            return rng.random(n) < p if (b % 3 != 1) else None      # every third batch carries no bitmap at all
        out.append(pa.RecordBatch.from_arrays([
            pa.array(rng.integers(-1000, 1000, n).astype(np.int32), mask=m(null_share)),
            pa.array((rng.random(n) * 100).astype(np.float32), mask=m(0.05)),
            pa.array(rng.integers(0, 2, n).astype(bool), mask=m(0.3)),
            pa.array(rng.integers(0, 2, n).astype(bool)),
            pa.array(rng.integers(-2**40, 2**40, n), type=pa.int64(), mask=m(0.5)),
            pa.array(["s%d" % v for v in rng.integers(0, 50, n)]),
            pa.array(rng.integers(0, 200, n).astype(np.uint8)),
        ], names=["a", "x", "flag", "flag2", "big", "name", "tiny"]))
    return out


@pytest.mark.parametrize("sql", [
    "tiny > 100",                                   # predicate on a non-null column; five bitmaps ride along
    "a > 0",                                        # a nullable predicate column: null -> the row is dropped
    "a + 1 > 0 and x < 50.0",                       # two nullable columns (generic interpreter, validity per batch)
    "flag",                                         # a nullable Boolean column as the predicate
    "flag = flag2 or a < 5",
    "big > 0 and flag2",                            # 64-bit + Boolean
])
@pytest.mark.parametrize("rows", [10_000, 3_000])
def test_device_groups_with_nulls_and_booleans_in_one_launch(ctx, sql, rows):
    """validity bitmaps (present in some batches only, at their own bit offsets) and Boolean columns of a device-resident,
    near-uniform group: one group launch + one bitmap compaction per Boolean column / nullable column, outputs identical to
    the per-batch oracle -- per-batch outputs, joined output, host results, sliced inputs"""
    recs = nullable_group(9, rows, rows)
    recs[4] = recs[4].slice(3, recs[4].num_rows - 7)          # Arrow offsets: bitmaps start mid-byte
    al = empty_aliases(recs[0])
    e = parse_expr(sql)
    exp = [O.filter_record(r, al, e) for r in recs]
    devs = [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs]
    got = chq.filter_records(devs, al, e, ctx=ctx)
    st = ctx.last_stats()
    if "big" not in sql:   # (a 64-bit predicate next to a Utf8 column has no one-launch form: joined on the device)
        assert st["launches"] <= 1 + 6, st                     # never the join (>= 7 concat launches) nor a per-batch loop
    for g, x in zip(got, exp):
        assert batches_identical(g.to_host(), x), explain_diff(g.to_host(), x)
    host_out = chq.filter_records(devs, al, e, ctx=ctx, device_result=False)
    for g, x in zip(host_out, exp):
        assert batches_identical(g, x), explain_diff(g, x)
    joined, counts = chq.filter_records_coalesced(devs, al, e, ctx=ctx)
    assert counts == [x.num_rows for x in exp]
    whole = pa.Table.from_batches(exp).combine_chunks().to_batches()
    if whole:
        assert batches_identical(joined.to_host(), whole[0], check_nullable=False)
    ctx.set_option("group_bits", 0)                            # the join path gives the same answer
    for g, x in zip(chq.filter_records(devs, al, e, ctx=ctx), exp):
        assert batches_identical(g.to_host(), x)
    ctx.set_option("group_bits", 1)


def test_group_errors_are_those_of_the_earliest_failing_batch(ctx):
    """no partial output; status and message of the first batch (array order) whose single call fails"""
    def ints(vals):
        return pa.RecordBatch.from_arrays([pa.array(np.asarray(vals, dtype=np.int32)), pa.array(np.arange(len(vals), dtype=np.int32))], names=["a", "d"])

    ok = ints(np.arange(1, 3001))
    overflow = ints([1, 2, 2**31 - 1] + [5] * 3000)     # a + 1 overflows
    zero_div = ints([7] * 5000)                          # a / d: d[0] = 0 -> divide by zero
    al = empty_aliases(ok)
    e = parse_expr("(a + 1) / d > 0")
    # ok batch: d[0] = 0 as well -> use a predicate that divides by a (never zero) for ok / overflow ordering
    e2 = parse_expr("(a + 1) / a > 0")
    for recs, expr in [([ok, overflow, zero_div], e2), ([ok, ints([3] * 10), overflow], e2)]:
        codes = []
        for r in recs:
            try:
                O.filter_record(r, al, expr)
                codes.append(None)
            except O.OracleError as err:
                codes.append(err.code)
        first = next(c for c in codes if c is not None)
        for device in (True, False):
            src = [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs] if device else recs
            with pytest.raises(chq.ChqError) as ei:
                chq.filter_records(src, al, expr, ctx=ctx)
            assert ei.value.code == first
    # div-by-zero in the first batch wins over the overflow in a later one
    with pytest.raises(chq.ChqError) as ei:
        chq.filter_records([zero_div, overflow, ok], al, e, ctx=ctx)
    assert ei.value.code == 21
    # the context stays usable
    check_group(ctx, [ok, ok], "a > 10", True)


def test_group_static_errors_and_argument_checks(ctx):
    recs = [fixed_batch(100, 1), fixed_batch(200, 2)]
    al = empty_aliases(recs[0])
    for sql in ["nope > 1", "id + 1", "id > 'x'"]:   # unknown column, not a Boolean predicate, no common type
        with pytest.raises(O.OracleError) as xi:
            O.filter_record(recs[0], al, parse_expr(sql))
        for src in (recs, [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs]):
            with pytest.raises(chq.ChqError) as ei:
                chq.filter_records(src, al, parse_expr(sql), ctx=ctx)
            assert ei.value.code == xi.value.code, sql
    with pytest.raises(ValueError):
        chq.RecordGroup([], ctx)
    with pytest.raises(ValueError):
        chq.RecordGroup([recs[0], chq.DeviceRecordBatch.from_host(recs[1], ctx)], ctx)


def test_large_group_many_small_batches(ctx):
    """2 000 batches x 10 000 rows resident in HBM, wrapped zero-copy from one allocation; counts checked against numpy"""
    import torch
    nb, rows = 2000, 10_000
    g = torch.Generator(device="cuda").manual_seed(3)
    cols = [torch.rand(nb * rows, generator=g, device="cuda", dtype=torch.float32) * 20.0 for _ in range(3)]
    torch.cuda.synchronize()
    devs = []
    for b in range(nb):
        devs.append(chq.DeviceRecordBatch.from_device_pointers(
            [(name, "f", c.data_ptr() + 4 * b * rows) for name, c in zip(("id", "value1", "value2"), cols)], rows, ctx))
    grp = chq.RecordGroup(devs, ctx)
    e = parse_expr("value2 > 10.0")
    outs = chq.filter_records(grp, [[], [], []], e, ctx=ctx)
    st = ctx.last_stats()
    assert st["launches"] == 1 and st["rows_in"] == nb * rows
    mask = cols[2] > 10.0
    exp_counts = mask.view(nb, rows).sum(dim=1).cpu().numpy()
    assert [o.num_rows for o in outs] == exp_counts.tolist()
    total = int(exp_counts.sum())
    for k in range(3):
        exp = torch.masked_select(cols[k], mask)
        base = outs[0].column_buffer_address(k)
        got = torch.empty(total, dtype=torch.float32, device="cuda")
        _dtod(got, base, 4 * total)
        assert torch.equal(got, exp)
    # spot-check a few batches through the ordinary export path
    for b in (0, 777, nb - 1):
        h = outs[b].to_host()
        lo, hi = b * rows, (b + 1) * rows
        assert np.array_equal(h.column(2).to_numpy(), cols[2][lo:hi][mask[lo:hi]].cpu().numpy())


@pytest.mark.parametrize("device", [True, False], ids=["device", "host"])
def test_coalesced_output_is_the_concatenation_of_the_per_batch_results(ctx, device):
    """chq_filter_records_coalesced: one output batch = the per-batch results back to back, plus rows per input batch"""
    cases = [
        ([fixed_batch(n, 500 + i) for i, n in enumerate([10_000] * 12 + [3333, 2])], "value2 > 10.0"),          # one launch
        ([fixed_batch(n, 520 + i) for i, n in enumerate(RAGGED)], "id % 2 = 0 and value1 < 20.0"),                # tile table
        ([mixed_batch(n, 540 + n, nulls=True) for n in (100, 3000, 17, 2500)], "a > 50 or f"),                    # general columns
        ([fixed_batch(n, 560 + n) for n in (100, 0, 1, 5000)], "value2 > 10.0"),                                  # batch by batch
        ([fixed_batch(n, 570 + n) for n in (40, 50)], "id <> id"),                                                # nothing survives
    ]
    for recs, sql in cases:
        al = empty_aliases(recs[0])
        e = parse_expr(sql)
        parts = [O.filter_record(r, al, e) for r in recs]
        exp = pa.Table.from_batches(parts).combine_chunks()
        exp = exp.to_batches()[0] if exp.num_rows else parts[0].slice(0, 0)
        src = [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs] if device else recs
        got, rows = chq.filter_records_coalesced(src, al, e, ctx=ctx)
        got = got.to_host() if device else got
        assert rows == [p.num_rows for p in parts], sql
        assert batches_identical(got, exp), f"{sql}:\n{explain_diff(got, exp)}"
    with pytest.raises(chq.ChqError):
        chq.filter_records_coalesced([fixed_batch(10, 1), fixed_batch(10, 2)], empty_aliases(fixed_batch(10, 1)), parse_expr("nope > 1"), ctx=ctx)


def test_device_groups_with_short_strings_are_filtered_straight_out_of_the_batches(ctx):
    """device-resident, non-null groups with one or two short-string Utf8 columns take the one-launch path (their offsets and
    bytes per batch ride in the group table, no join): uniform batches (wave-packed layout) and ragged ones (per-tile table),
    small and large tiles, every batch against the oracle, the joined form against the per-batch form, and the joined-first
    path (`group_fold` = 0) for the same answer"""
    rng = np.random.default_rng(2024)

    def batch(n, seed, two):
        r = np.random.default_rng(seed)
        cols = {"id": pa.array(r.integers(0, 1000, n).astype(np.int32)),
                "s": pa.array(["w" * int(l) + str(i % 7) for i, l in enumerate(r.integers(0, 20, n))]),
                "v": pa.array((r.random(n) * 100).astype(np.float32))}
        if two:
            cols["t"] = pa.array(["%x" % v for v in r.integers(0, 2**31, n)])
            cols["k"] = pa.array(r.integers(-9, 9, n).astype(np.int64))
        return pa.record_batch(cols)

    for sizes, two in [([10_000] * 40, False), ([10_000] * 40, True), ([int(x) for x in rng.integers(2, 30_000, 25)], True),
                       ([70_000, 3, 16_384, 16_385, 2, 50_000], False)]:
        recs = [batch(n, 500 + i, two) for i, n in enumerate(sizes)]
        al = empty_aliases(recs[0])
        devs = [chq.DeviceRecordBatch.from_host(r, ctx) for r in recs]
        for sql in ["id % 2 = 0", "v > 10.0", "v > 99.0 and id > 5", "id < 0"]:
            e = parse_expr(sql)
            for tile_kind in (-1, 0, 1):
                ctx.set_option("tile_kind", tile_kind)
                try:
                    got = chq.filter_records(devs, al, e, ctx=ctx)
                    assert ctx.last_stats()["launches"] <= 2, (sql, tile_kind)      # ONE kernel (+ the batch-end gather in tile mode)
                    for i, g in enumerate(got):
                        assert batches_identical(g.to_host(), O.filter_record(recs[i], al, e)), (sql, tile_kind, i)
                    big, per = chq.filter_records_coalesced(devs, al, e, ctx=ctx)
                    assert per == [g.num_rows for g in got]
                    joined = pa.Table.from_batches([g.to_host() for g in got]).combine_chunks().to_batches()
                    if big.num_rows:
                        assert batches_identical(big.to_host(), joined[0], check_nullable=False), (sql, tile_kind)
                    ctx.set_option("group_fold", 0)
                    ref = chq.filter_records(devs, al, e, ctx=ctx)
                    ctx.set_option("group_fold", 1)
                    for g, r in zip(got, ref):
                        assert batches_identical(g.to_host(), r.to_host())
                    host = chq.filter_records(devs, al, e, ctx=ctx, device_result=False)
                    for g, h in zip(got, host):
                        assert batches_identical(g.to_host(), h)
                finally:
                    ctx.set_option("tile_kind", -1)
                    ctx.set_option("group_fold", 1)


@pytest.mark.parametrize("nulls", [False, True], ids=["non-null", "with-nulls"])
def test_groups_too_large_for_one_launch_run_as_sub_groups(nulls):
    """a device group of short strings whose JOINED output would not fit int32 offsets (the reference's batch size at config-5
    scale: 10^5 batches, 8 GB of strings) runs as consecutive sub-groups, each through the one-launch path -- forced here on
    small data with `group_chunk_bytes`: the cut positions, the per-batch outputs (device and host form) and the launch count"""
    c = chq.Context(0)
    rng = np.random.default_rng(77)
    sizes = [3000] * 30 + [int(x) for x in rng.integers(2, 4000, 11)]
    recs = []
    for i, n in enumerate(sizes):
        r = np.random.default_rng(900 + i)
        v = pa.array((r.random(n) * 100).astype(np.float32), mask=(r.random(n) < 0.1) if nulls else None)
        recs.append(pa.record_batch({"id": pa.array(r.integers(0, 1000, n).astype(np.int32)),
                                     "value1": pa.array(["%08x" % x for x in r.integers(0, 2**32, n)]), "value2": v}))
    al = empty_aliases(recs[0])
    devs = [chq.DeviceRecordBatch.from_host(r, c) for r in recs]
    for chunk_bytes, min_launches in [(1 << 30, 1), (100_000, 8), (30_000, 25)]:   # 24 kB of strings per 3000-row batch
        c.set_option("group_chunk_bytes", chunk_bytes)
        for sql in ["id % 2 = 0", "value2 > 50.0"]:
            e = parse_expr(sql)
            got = chq.filter_records(devs, al, e, ctx=c)
            assert c.last_stats()["launches"] >= min_launches, (chunk_bytes, c.last_stats())
            assert c.last_stats()["rows_in"] == sum(sizes)
            host = chq.filter_records(devs, al, e, ctx=c, device_result=False)
            for i, (g, h) in enumerate(zip(got, host)):
                want = O.filter_record(recs[i], al, e)
                assert batches_identical(g.to_host(), want), f"{sql}, chunk {chunk_bytes}, batch {i}:\n{explain_diff(g.to_host(), want)}"
                assert batches_identical(h, want), (sql, chunk_bytes, i)
            # results of different sub-groups are independent blocks: releasing some does not touch the others
            keep = got[-1].to_host()
            for g in got[:-1]:
                g.release()
            assert batches_identical(got[-1].to_host(), keep)
    c.close()


@pytest.mark.parametrize("nulls", [False, True], ids=["non-null", "value2-with-nulls"])
def test_groups_of_uniform_length_strings_run_as_plain_groups(nulls):
    """device groups whose string columns all hold values of ONE length (the reference's sample strings) are proved uniform by
    one pass over the offsets and filtered as fixed-width columns: per-batch outputs, the joined form and the host form
    against the oracle, next to groups that must keep the string path (one ragged value; a 3-byte length)"""
    c = chq.Context(0)
    c.set_option("uniform_utf8_rows", 1000)   # (default: groups of 2^24 rows and more)
    rng = np.random.default_rng(5)
    sizes = [3000] * 20 + [int(x) for x in rng.integers(2, 4000, 9)]

    def group(fmt, spoil=None):
        recs = []
        for i, n in enumerate(sizes):
            r = np.random.default_rng(700 + i)
            s = [fmt % x for x in r.integers(0, 2**24, n)]
            if spoil is not None and i == spoil:
                s[n // 2] = "x"
            v = pa.array((r.random(n) * 100).astype(np.float32), mask=(r.random(n) < 0.1) if nulls else None)
            recs.append(pa.record_batch({"id": pa.array(r.integers(0, 1000, n).astype(np.int32)), "value1": pa.array(s), "value2": v,
                                         "key16": pa.array(["%016x" % x for x in r.integers(0, 2**62, n)])}))
        return recs

    for recs in (group("%08x"), group("%08x", spoil=17), group("%03x")):
        al = empty_aliases(recs[0])
        devs = [chq.DeviceRecordBatch.from_host(r, c) for r in recs]
        for sql in ["id % 2 = 0", "value2 > 50.0", "id < 0"]:
            e = parse_expr(sql)
            want = [O.filter_record(r, al, e) for r in recs]
            got = chq.filter_records(devs, al, e, ctx=c)
            host = chq.filter_records(devs, al, e, ctx=c, device_result=False)
            for i in range(len(recs)):
                assert batches_identical(got[i].to_host(), want[i]), f"{sql}, batch {i}:\n{explain_diff(got[i].to_host(), want[i])}"
                assert batches_identical(host[i], want[i]), (sql, i)
            big, per = chq.filter_records_coalesced(devs, al, e, ctx=c)
            assert per == [w.num_rows for w in want]
            whole = pa.Table.from_batches(want).combine_chunks().to_batches()
            if big.num_rows:
                assert batches_identical(big.to_host(), whole[0], check_nullable=False), sql
    c.close()
