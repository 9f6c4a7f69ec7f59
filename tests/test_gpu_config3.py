"""GPU: BASELINE.json config 3, the exact workload (SURVEY.md section 8 d):

    schema     a:Int32 U[0,1000), b:Float32 U[0,100), c:Float32 U[0,1100), d:Int32 U[0,10), e:Float32 U[0,2)
    predicate  a + b > c and d < 5.0 or e > 1.0          (parsed ((a + b > c) AND (d < 5.0)) OR (e > 1.0))
    projection a, a + b AS ab, d * 2 AS d2, e / 3.0 AS e3   on the surviving rows (filter -> exchange -> materialize)

Bit-exact against the oracle at 1 M rows (every tile kind, host and HBM inputs, the two reference steps and the one-pass
`chq_filter_project_record`); at the full 1 B rows through size-independent properties and torch's own compaction."""
import ctypes
import os

import numpy as np
import pyarrow as pa
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_select
from oracle import oracle as O

from .helpers import batches_identical, explain_diff

pytestmark = pytest.mark.gpu

SQL = "select a, a + b as ab, d * 2 as d2, e / 3.0 as e3 from t where a + b > c and d < 5.0 or e > 1.0"
ALIASES = [[], [], [], [], []]


def config3_host(n, seed=0xC0FFEE):
    rng = np.random.default_rng(seed)
    return pa.RecordBatch.from_arrays([
        pa.array(rng.integers(0, 1000, n).astype(np.int32)),
        pa.array((rng.random(n) * 100).astype(np.float32)),
        pa.array((rng.random(n) * 1100).astype(np.float32)),
        pa.array(rng.integers(0, 10, n).astype(np.int32)),
        pa.array((rng.random(n) * 2).astype(np.float32)),
    ], schema=pa.schema([pa.field(nm, t, nullable=False) for nm, t in
                         [("a", pa.int32()), ("b", pa.float32()), ("c", pa.float32()), ("d", pa.int32()), ("e", pa.float32())]]))


def test_the_predicate_parses_with_sql_precedence():
    sel = parse_select(SQL)
    assert sel.selection.op.value == "Or" and sel.selection.left.op.value == "And"
    assert [f.alias.value for f in sel.projection[1:]] == ["ab", "d2", "e3"]


@pytest.mark.parametrize("resident", ["host", "hbm"])
@pytest.mark.parametrize("tile_kind", [-1, 0, 1, 2])
def test_config3_against_the_oracle_1m_rows(tile_kind, resident):
    n = 1_000_003   # not a multiple of any tile size
    host = config3_host(n)
    sel = parse_select(SQL)
    exp_f = O.filter_record(host, ALIASES, sel.selection)
    exp_p = O.project_record(sel.projection, exp_f, ALIASES)
    assert 0.55 < exp_f.num_rows / n < 0.70   # s ~ 0.62
    ctx = chq.Context(0)
    ctx.set_option("tile_kind", tile_kind)
    rec = chq.DeviceRecordBatch.from_host(host, ctx) if resident == "hbm" else host
    # the two reference steps: filter_record (filter_task.rs:99), then project_record (materialize_files_task.rs:110)
    got_f = chq.filter_record(rec, ALIASES, sel.selection, ctx=ctx)
    got_p = chq.project_record(sel.projection, got_f, ALIASES, ctx=ctx)
    f_host = got_f.to_host() if hasattr(got_f, "to_host") else got_f
    p_host = got_p.to_host() if hasattr(got_p, "to_host") else got_p
    assert batches_identical(f_host, exp_f, nan_payload=True), explain_diff(f_host, exp_f)
    assert batches_identical(p_host, exp_p, nan_payload=True), explain_diff(p_host, exp_p)
    # one pass (chq_filter_project_record): library's choice, forced single-pass kernel, forced two steps
    for fuse in (1, 2, 0):
        ctx.set_option("fuse", fuse)
        got = chq.filter_project_record(sel.selection, sel.projection, rec, ALIASES, ctx=ctx)
        got = got.to_host() if hasattr(got, "to_host") else got
        assert batches_identical(got, exp_p, nan_payload=True), (fuse, explain_diff(got, exp_p))
    ctx.close()


def test_config3_in_reference_sized_batches():
    """the same workload as the reference would feed it: 10 000-row batches (physical_planner.rs:323) through one
    chq_filter_records call, every output batch against the oracle's"""
    host = config3_host(200_000, seed=3)
    sel = parse_select(SQL)
    parts = [host.slice(i, 10_000) for i in range(0, host.num_rows, 10_000)]
    ctx = chq.Context(0)
    for src in (parts, [chq.DeviceRecordBatch.from_host(p, ctx) for p in parts]):
        outs = chq.filter_records(src, ALIASES, sel.selection, ctx=ctx)
        assert len(outs) == len(parts)
        for o, p in zip(outs, parts):
            o = o.to_host() if hasattr(o, "to_host") else o
            assert batches_identical(o, O.filter_record(p, ALIASES, sel.selection), nan_payload=True)
    ctx.close()


def _dtod(dst_tensor, src_addr, nbytes):
    import torch
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(dst_tensor.data_ptr(), src_addr, nbytes, 3) == 0   # hipMemcpyDeviceToDevice


def test_config3_full_size_properties():
    """1 B rows: count, order, every survivor satisfies the predicate, compaction bit-equal to torch.masked_select for all
    five columns; projected columns: integer columns bit-exact, Float32 columns within 1 ULP of torch's own arithmetic
    (the north_star's tolerance for float results; bit-exactness against the oracle is checked at 1 M rows above)."""
    import torch
    n = 1_000_000_000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0xC0FFEE + 3)
    a = torch.randint(0, 1000, (n,), dtype=torch.int32, device=dev, generator=g)
    b = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
    c = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 1100, generator=g)
    d = torch.randint(0, 10, (n,), dtype=torch.int32, device=dev, generator=g)
    e = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 2, generator=g)
    ids = torch.arange(n, dtype=torch.int32, device=dev)   # order witness (an extra pass-through column)
    ctx = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    rec = chq.DeviceRecordBatch.from_device_pointers(
        [("a", "i", a.data_ptr()), ("b", "f", b.data_ptr()), ("c", "f", c.data_ptr()), ("d", "i", d.data_ptr()),
         ("e", "f", e.data_ptr()), ("id", "i", ids.data_ptr())], n, ctx=ctx)
    sel = parse_select(SQL)
    al = ALIASES + [[]]
    out = chq.filter_record(rec, al, sel.selection, ctx=ctx)
    # the same predicate in torch: values are finite, non-negative and a < 2^24, so IEEE compares = totalOrder compares
    # and Int32 -> Float32 is exact
    mask = ((a.to(torch.float32) + b > c) & (d.to(torch.float32) < 5.0)) | (e > 1.0)
    m = int(mask.sum().item())
    assert out.num_rows == m and 0.55 < m / n < 0.70
    kept = {}
    for i, (name, src) in enumerate([("a", a), ("b", b), ("c", c), ("d", d), ("e", e), ("id", ids)]):
        t = torch.empty(m, dtype=src.dtype, device=dev)
        _dtod(t, out.column_buffer_address(i, 1), m * 4)
        exp = torch.masked_select(src, mask)
        assert torch.equal(exp.view(torch.int32), t.view(torch.int32)), name     # bit-equal compaction
        del exp
        kept[name] = t
    torch.cuda.synchronize()
    assert bool((kept["id"][1:] > kept["id"][:-1]).all().item())                  # order preserved
    ok = ((kept["a"].to(torch.float32) + kept["b"] > kept["c"]) & (kept["d"] < 5)) | (kept["e"] > 1.0)
    assert bool(ok.all().item())                                                  # every survivor satisfies it
    del mask, ok
    # projection of the survivors (materialize step), two-step and one-pass
    proj = chq.project_record(sel.projection, out, al, ctx=ctx)
    ctx.set_option("fuse", 2)
    fused = chq.filter_project_record(sel.selection, sel.projection, rec, al, ctx=ctx)
    assert proj.num_rows == m and fused.num_rows == m

    def ulp_close(x, y):   # within 1 ULP: the integer keys of two finite floats of equal sign differ by at most 1
        return bool(((x.view(torch.int32) - y.view(torch.int32)).abs() <= 1).all().item())

    for res in (proj, fused):
        cols = []
        for i, dt in enumerate([torch.int32, torch.float32, torch.int32, torch.float32]):
            t = torch.empty(m, dtype=dt, device=dev)
            _dtod(t, res.column_buffer_address(i, 1), m * 4)
            cols.append(t)
        assert torch.equal(cols[0], kept["a"])
        assert ulp_close(cols[1], kept["a"].to(torch.float32) + kept["b"])
        assert torch.equal(cols[2], kept["d"] * 2)
        assert ulp_close(cols[3], kept["e"] / 3.0)
        del cols
    ctx.close()
