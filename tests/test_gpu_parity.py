"""GPU: parity of the HIP path (through the C ABI) with the reference's test vectors, the arrow-semantics table
and the CPU oracle.  Bit-exact for integers, booleans, strings, row selection and copied floats; computed floats
are compared bit-exact modulo NaN payload (tests/helpers.py)."""
import numpy as np
import pyarrow as pa
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr, parse_select, parse_statements
from oracle import oracle as O

from . import rules
from .cases import empty_aliases, run_golden_case, run_project_error, run_project_rule, run_rule
from .helpers import arrays_identical, batches_identical, explain_diff, load_golden

pytestmark = pytest.mark.gpu

GOLD = load_golden("reference_cases.json")
SIZES = [0, 1, 2, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4096 + 17, 16383, 16384, 16385, 50_000]


@pytest.fixture(scope="module")
def ctx():
    c = chq.Context(0)
    yield c
    c.close()


class _Impl:
    """the chq entry points bound to one context / tile kind"""

    def __init__(self, ctx):
        self.ctx = ctx

    def compute_value(self, rec, al, e):
        return chq.compute_value(rec, al, e, ctx=self.ctx)

    def filter_record(self, rec, al, e):
        return chq.filter_record(rec, al, e, ctx=self.ctx)

    def project_record(self, f, rec, al):
        return chq.project_record(f, rec, al, ctx=self.ctx)


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_reference_test_vectors(ctx, case):
    """pinned-by-reference: record_utils/test_*.rs"""
    run_golden_case(_Impl(ctx), case)


@pytest.mark.parametrize("rule", rules.RULES, ids=[r[0] for r in rules.RULES])
def test_arrow_semantics(ctx, rule):
    run_rule(_Impl(ctx), rule)


@pytest.mark.parametrize("rule", rules.PROJECT_RULES, ids=[r[0] for r in rules.PROJECT_RULES])
def test_project_record_rules(ctx, rule):
    run_project_rule(_Impl(ctx), rule)


@pytest.mark.parametrize("rule", rules.PROJECT_ERRORS, ids=[r[0] for r in rules.PROJECT_ERRORS])
def test_project_record_errors(ctx, rule):
    run_project_error(_Impl(ctx), rule)


# ------------------------------------------------------------------- the opt-in Minus extension (not reference)
@pytest.fixture
def minus_ctx():   # function-scoped: the oracle's extension mode must never leak into the reference-behaviour tests
    c = chq.Context(0)
    c.set_option("enable_minus", 1)
    with O.extension_minus():
        yield c
    c.close()


@pytest.mark.parametrize("rule", rules.MINUS_RULES, ids=[r[0] for r in rules.MINUS_RULES])
def test_minus_extension_rules(minus_ctx, rule):
    """`enable_minus` = arrow-arith numeric::sub; the reference itself rejects Minus (rule minus_not_implemented)"""
    run_rule(_Impl(minus_ctx), rule)


@pytest.mark.parametrize("tile_kind", [-1, 0, 1, 2])
@pytest.mark.parametrize("n", [1, 65, 2049, 16385, 40_000])
def test_minus_extension_fuzz_vs_oracle(minus_ctx, n, tile_kind):
    """OP_SUB in every arithmetic class (i8..u64, f32, f64), both operand orders, literal operands and folded
    constants, overflow / underflow statuses: GPU with enable_minus vs the oracle's extension mode"""
    minus_ctx.set_option("tile_kind", tile_kind)
    rng = np.random.default_rng(4242 + n + tile_kind)
    rec = make_batch(n, 17 * n + 3)
    al = empty_aliases(rec)
    outcomes = {"ok": 0, "error": 0, "unsupported": 0}
    ops = ("-", "-", "-", "+", "*", "/", "%")
    fixed = ["i8 - u8", "u8 - i8", "u16 - u8", "u8 - u16", "u64 - u64", "i64 - i32", "100 - i32", "i32 - 100", "2.5 - f32",
             "f32 - f64", "f64 - 0.5", "i16 - small - 1", "u32 - u32", "(i32 - 7) % 5", "i32 - i32 * 2 > small - 3",
             "f32 - f32 = 0.0", "1 - 2 - 3 + i32"]
    for sql in fixed:
        outcomes[check_same(minus_ctx, rec, al, sql, "value")] += 1
    for _ in range(20):
        outcomes[check_same(minus_ctx, rec, al, random_numeric(rng, 3, ops=ops), "value")] += 1
    for _ in range(6):
        cols, lits = pick_family(rng)
        sql = f"{random_numeric(rng, 2, cols, ops, lits=lits)} {rng.choice(['<', '>=', '=', '<>'])} {random_numeric(rng, 1, cols, ops, lits=lits)}"
        outcomes[check_same(minus_ctx, rec, al, sql, "filter")] += 1
    minus_ctx.set_option("tile_kind", -1)
    assert outcomes["ok"] >= 12 and outcomes["unsupported"] == 0, outcomes


def test_minus_extension_in_projection_and_one_pass(minus_ctx):
    rec = make_batch(30_000, 9, nulls=False)
    al = empty_aliases(rec)
    sel = parse_select("select i32 - small as d1, f32 - 1.5 as d2, 1000 - i16 as d3 from t where i32 - 50 > small - u8")
    exp = O.project_record(sel.projection, O.filter_record(rec, al, sel.selection), al)
    dev = chq.DeviceRecordBatch.from_host(rec, minus_ctx)
    two = chq.project_record(sel.projection, chq.filter_record(dev, al, sel.selection, ctx=minus_ctx), al, ctx=minus_ctx).to_host()
    assert batches_identical(two, exp, nan_payload=True), explain_diff(two, exp)
    minus_ctx.set_option("fuse", 2)
    one = chq.filter_project_record(sel.selection, sel.projection, dev, al, ctx=minus_ctx).to_host()
    minus_ctx.set_option("fuse", 1)
    assert batches_identical(one, exp, nan_payload=True), explain_diff(one, exp)


def test_minus_stays_rejected_without_the_option(ctx):
    rec = make_batch(100, 1)
    with pytest.raises(chq.ChqError) as ei:
        chq.compute_value(rec, empty_aliases(rec), parse_expr("i32 - 1"), ctx=ctx)
    assert ei.value.code == 3


# ---------------------------------------------------------------------------------------------- fuzz vs oracle
def make_batch(n, seed, nulls=True, tame=False):
    """`tame`: integer columns hold small magnitudes, so that sums and products of two or three of them stay inside their
    type -- checked arithmetic (arrow's add / mul fail the whole batch on the first overflow) then mostly computes VALUES for
    the fuzzers to compare; the full-range form keeps the boundary values (and ends most arithmetic in an overflow error)"""
    rng = np.random.default_rng(seed)

    def m(p):
        return (rng.random(n) < p) if nulls and n else None

    def ints(lo, hi, tlo, thi, dtype):
        return rng.integers(tlo, thi, n).astype(dtype) if tame else rng.integers(lo, hi, n).astype(dtype)

    words = np.array(["", "a", "ab", "abc", "b", "zeta", "a much longer string value that spans more than sixty-four bytes of utf8 text ..."])
    cols = {
        "i8": pa.array(ints(-128, 128, -5, 6, np.int8)),
        "i16": pa.array(ints(-3000, 3000, -30, 31, np.int16), mask=m(0.1)),
        "i32": pa.array(ints(-100000, 100000, -1000, 1001, np.int32)),
        "i64": pa.array(ints(-10**12, 10**12, -10**6, 10**6, np.int64), mask=m(0.05)),
        "u8": pa.array(ints(0, 256, 0, 6, np.uint8)),
        "u16": pa.array(ints(0, 65536, 0, 40, np.uint16)),
        "u32": pa.array(ints(0, 2**32, 0, 1500, np.uint32), mask=m(0.1)),
        "u64": pa.array(ints(0, 2**63, 0, 10**6, np.uint64)),
        "f32": pa.array((rng.random(n) * 200 - 100).astype(np.float32), mask=m(0.1)),
        "f64": pa.array(rng.random(n) * 2e6 - 1e6),
        "small": pa.array(rng.integers(1, 9, n).astype(np.int32)),
        "flag": pa.array(rng.integers(0, 2, n).astype(bool), mask=m(0.2)),
        "flag2": pa.array(rng.integers(0, 2, n).astype(bool)),
        "s": pa.array(words[rng.integers(0, len(words), n)] if n else np.array([], dtype=object), type=pa.utf8(), mask=m(0.1)),
        "k": pa.array(["k%d" % v for v in rng.integers(0, 20, n)], type=pa.utf8()),
    }
    return pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys()))


NUMERIC = ["i8", "i16", "i32", "i64", "u8", "u16", "u32", "u64", "f32", "f64", "small"]
# Column families in which EVERY pair of types -- and every pair of intermediate result types -- has a common type in the
# coercion table (compute_value.rs:350-431), with the literal kinds that coerce against all of them ("i": an integer literal
# is Int32, "f": a decimal literal is Float32): expressions drawn from one family evaluate instead of ending in
# UnsupportedTypeCoersion.  The last entry is the old free-for-all (mostly static errors), kept with a small weight.
FAMILIES = [
    (["i8", "i16", "i32", "small", "i64"], "i"),
    (["u8", "u16", "u32", "u64"], ""),
    (["u8", "u16", "i32", "i64", "small"], "i"),
    (["f32", "i8", "i16", "i32", "small"], "if"),
    (["f32", "i16", "i32", "u8", "small"], "if"),
    (["f32", "u8", "u16", "u32"], "f"),
    (["f64", "f32"], "if"),
    (["f64", "i32", "i64", "small"], "i"),
    (["f64", "u8", "u32", "u64"], ""),
    (NUMERIC, "if"),
]
FAMILY_WEIGHTS = np.array([3, 2, 2, 3, 2, 2, 2, 2, 1, 1], dtype=float) / 20.0
_INT_NONZERO = {"small"}          # integer columns that never hold 0: safe divisors
_FLOATS = {"f32", "f64"}


def pick_family(rng):
    return FAMILIES[int(rng.choice(len(FAMILIES), p=FAMILY_WEIGHTS))]


def random_numeric(rng, depth, cols=None, ops=("+", "*", "/", "%", "+", "/"), lits="if", wild=0.1):
    """a numeric expression over one family.  With probability 1 - `wild` the operators are chosen so that the expression
    cannot fail on a tame batch for a trivial reason: integer `/` and `%` take a non-zero divisor (a literal or `small`),
    `*` of integers multiplies by a literal or `small`"""
    if cols is None:
        cols, lits = pick_family(rng)
    cols = list(cols)
    careful = rng.random() >= wild

    def literal():
        kinds = [k for k in lits]
        if not kinds:
            return None
        if str(rng.choice(kinds)) == "i" or not (set(cols) & _FLOATS):
            return str(int(rng.integers(1, 9))) if "i" in kinds else "%.2f" % (rng.random() * 10 + 0.5)
        return "%.2f" % (rng.random() * 10 + 0.5)

    def leaf():
        lit = literal()
        if lit is not None and rng.random() >= 0.75:
            return lit
        return str(rng.choice(cols))

    def safe_factor():   # a right operand that neither is zero nor blows a small integer type up
        lit = literal()
        cand = [c for c in cols if c in _INT_NONZERO] + ([lit] if lit is not None else [])
        return str(rng.choice(cand)) if cand else None

    def gen(d):
        if d == 0 or rng.random() < 0.3:
            return leaf()
        op = str(rng.choice(list(ops)))
        left = gen(d - 1)
        floaty = bool(set(cols) & _FLOATS) and all(c in _FLOATS for c in cols if c in left.split())
        if careful and op in ("*", "/", "%") and not floaty:
            right = safe_factor()
            if right is None:
                op, right = "+", gen(d - 1)
        else:
            right = gen(d - 1)
        e = f"{left} {op} {right}"
        return f"({e})" if rng.random() < 0.5 else e
    return gen(depth)


def random_predicate(rng, depth):
    if depth == 0 or rng.random() < 0.35:
        r = rng.random()
        if r < 0.6:
            cols, lits = pick_family(rng)
            return f"{random_numeric(rng, 2, cols, lits=lits)} {rng.choice(['<', '<=', '>', '>=', '=', '<>'])} {random_numeric(rng, 1, cols, lits=lits)}"
        if r < 0.7:
            return rng.choice(["flag", "flag2"])
        if r < 0.8:
            return f"flag {rng.choice(['=', '<>', '<'])} flag2"
        if r < 0.9:
            return f"s {rng.choice(['<', '>=', '=', '<>'])} '{rng.choice(['a', 'ab', 'b', 'zz', ''])}'"
        return f"k {rng.choice(['=', '<>', '<='])} 'k{int(rng.integers(0, 20))}'"
    return f"({random_predicate(rng, depth - 1)} {rng.choice(['and', 'or'])} {random_predicate(rng, depth - 1)})"


def check_same(ctx, rec, al, sql, kind):
    e = parse_expr(sql)
    try:
        if kind == "filter":
            exp = O.filter_record(rec, al, e)
        else:
            exp = O.compute_value(rec, al, e)[0]
        exp_code = None
    except O.OracleError as err:
        exp, exp_code = None, err.code
    try:
        if kind == "filter":
            got = chq.filter_record(rec, al, e, ctx=ctx)
        else:
            got = chq.compute_value(rec, al, e, ctx=ctx)[0]
        got_code = None
    except chq.ChqError as err:
        got, got_code = None, err.code
    if exp_code is not None or got_code is not None:
        # NotSupported (30) only counts when BOTH sides say it (the documented out-of-scope shapes):
        # a GPU path that answers NotSupported where the oracle computes a value -- or the reverse -- is a failure
        assert got_code == exp_code, f"{sql}: oracle status {exp_code}, gpu status {got_code}"
        return "unsupported" if exp_code == 30 else "error"
    if kind == "filter":
        assert batches_identical(got, exp), f"{sql} (n={rec.num_rows}):\n{explain_diff(got, exp)}"
    else:
        assert arrays_identical(got, exp, nan_payload=True), f"{sql} (n={rec.num_rows}): value mismatch"
    return "ok"


@pytest.mark.parametrize("n", SIZES)
def test_fuzz_filter_vs_oracle(ctx, n):
    rng = np.random.default_rng(1000 + n)
    outcomes = {"ok": 0, "error": 0, "unsupported": 0}
    for tame in (True, False):   # tame values: the cases compare VALUES; full-range values: boundary values and error statuses
        rec = make_batch(n, n, tame=tame)
        al = empty_aliases(rec)
        for _ in range(12 if tame else 6):
            outcomes[check_same(ctx, rec, al, random_predicate(rng, 2), "filter")] += 1
    # "error" outcomes are checked too (same status on both sides), but a fuzz that mostly compares statuses proves little:
    # at least 10 of the 18 cases must have compared values
    assert outcomes["ok"] >= 10 and outcomes["unsupported"] <= 2, outcomes


@pytest.mark.parametrize("n", [1, 65, 2049, 20_000])
def test_fuzz_compute_value_vs_oracle(ctx, n):
    rng = np.random.default_rng(77 + n)
    outcomes = {"ok": 0, "error": 0, "unsupported": 0}
    for tame in (True, False):
        rec = make_batch(n, 5 * n + 1, tame=tame)
        al = empty_aliases(rec)
        for _ in range(25 if tame else 10):
            sql = random_numeric(rng, 3) if rng.random() < 0.6 else random_predicate(rng, 2)
            outcomes[check_same(ctx, rec, al, sql, "value")] += 1
    assert outcomes["ok"] >= 20 and outcomes["unsupported"] <= 3, outcomes


# ---- the reference's remaining type coverage: temporal / decimal same-type comparisons, Float16, Utf8 under AND / OR -------
def make_typed_batch(n, seed, nulls=True):
    import decimal
    rng = np.random.default_rng(seed)

    def m(p):
        return (rng.random(n) < p) if nulls and n else None

    def raw(typ, lo, hi, p=0.0):
        a = pa.array(rng.integers(lo, hi, n).astype(np.int32 if typ.bit_width == 32 else np.int64), mask=m(p))
        return a.view(typ)

    def dec(p=0.0):
        vals = [decimal.Decimal(int(x) * 2**40 + int(y)) / 100 for x, y in zip(rng.integers(-2**30, 2**30, n), rng.integers(0, 4, n))]
        return pa.array(vals, pa.decimal128(30, 2), mask=m(p)) if n else pa.array([], pa.decimal128(30, 2))

    halves = np.concatenate([np.array([0.0, -0.0, 1.0, 0.5, 65504.0, 6e-8, np.inf, -np.inf], dtype=np.float16),
                             (rng.standard_normal(64) * 8).astype(np.float16)])
    words = np.array(["true", "false", " yes", "NO ", "t", "0", "1", "off", "maybe", "", "on", "\u2003Y\u00a0", "fals", "truee"])
    cols = {
        "h": pa.array(halves[rng.integers(0, len(halves), n)], pa.float16(), mask=m(0.1)),
        "k": pa.array(halves[rng.integers(0, len(halves), n)], pa.float16()),
        "f32": pa.array((rng.random(n) * 20 - 10).astype(np.float32)),
        "f64": pa.array(rng.random(n) * 200 - 100, mask=m(0.05)),
        "i32": pa.array(rng.integers(-1000, 1001, n).astype(np.int32)),
        "d1": raw(pa.date32(), -5, 6, 0.1), "d2": raw(pa.date32(), -5, 6),
        "t1": raw(pa.timestamp("us"), -2**40, 2**40), "t2": raw(pa.timestamp("us"), -2**40, 2**40, 0.1),
        "u1": raw(pa.time32("s"), 0, 10), "u2": raw(pa.time32("s"), 0, 10),
        "x1": dec(0.1), "x2": dec(),
        "w": pa.array(words[rng.integers(0, len(words), n)] if n else np.array([], dtype=object), type=pa.utf8(), mask=m(0.1)),
        "flag": pa.array(rng.integers(0, 2, n).astype(bool), mask=m(0.2)),
    }
    return pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys()))


def random_typed_value(rng):
    ops = ["+", "*", "/", "%", "+"]
    r = rng.random()
    if r < 0.4:
        return f"h {rng.choice(ops)} k"
    if r < 0.6:
        return f"(h {rng.choice(ops)} k) {rng.choice(ops)} h"
    if r < 0.8:
        return f"h {rng.choice(ops)} {rng.choice(['f32', 'f64', '1.5', '(k + f32)'])}"
    return f"k {rng.choice(ops)} (f64 {rng.choice(ops)} h)"


def random_typed_predicate(rng, depth):
    cmp = lambda: str(rng.choice(["<", "<=", ">", ">=", "=", "<>"]))
    if depth == 0 or rng.random() < 0.4:
        r = rng.random()
        if r < 0.3:
            a, b = [("d1", "d2"), ("t1", "t2"), ("u1", "u2"), ("x1", "x2"), ("x2", "x1"), ("d2", "d2")][int(rng.integers(0, 6))]
            return f"{a} {cmp()} {b}"
        if r < 0.5:
            return f"h {cmp()} k"
        if r < 0.7:
            return f"{random_typed_value(rng)} {cmp()} {rng.choice(['h', 'f32', 'f64', '0.5'])}"
        if r < 0.85:
            return str(rng.choice(["w", "h", "flag", "k"]))
        return f"i32 {cmp()} {int(rng.integers(0, 500))}"
    return f"({random_typed_predicate(rng, depth - 1)} {rng.choice(['and', 'or'])} {random_typed_predicate(rng, depth - 1)})"


@pytest.mark.parametrize("n", [1, 64, 1000, 16385, 70_000])
def test_fuzz_typed_columns_vs_oracle(ctx, n):
    rng = np.random.default_rng(4000 + n)
    rec = make_typed_batch(n, n)
    al = empty_aliases(rec)
    outcomes = {"ok": 0, "error": 0, "unsupported": 0}
    for _ in range(14):
        outcomes[check_same(ctx, rec, al, random_typed_predicate(rng, 2), "filter")] += 1
    for _ in range(10):
        sql = random_typed_value(rng) if rng.random() < 0.5 else random_typed_predicate(rng, 1)
        outcomes[check_same(ctx, rec, al, sql, "value")] += 1
    # a top-level Float16 / Utf8 column is not a Boolean mask (CastToBooleanArrayFailed on both sides): a few errors are expected
    assert outcomes["ok"] >= 16 and outcomes["unsupported"] == 0, outcomes


def test_typed_columns_in_sliced_batches(ctx):
    """Arrow slices of the typed columns: Decimal128 values start at base + 16 * offset, the validity bitmaps and the Utf8
    offsets of the boolean-word column carry the slice offset into the typed-operation kernels"""
    rec = make_typed_batch(6000, 17)
    for start, length in [(1, 100), (7, 2049), (63, 64), (64, 4000), (5999, 1), (13, 0)]:
        sl = rec.slice(start, length)
        al = empty_aliases(sl)
        for sql in ["x1 < x2", "x1 >= x2 or w", "w and flag", "d1 <= d2 and t1 <> t2", "h < k", "h + k > f32 or u1 = u2"]:
            assert check_same(ctx, sl, al, sql, "filter") == "ok", (sql, start, length)
        for sql in ["x2 > x1", "w or flag", "h * k", "h + f64"]:
            assert check_same(ctx, sl, al, sql, "value") == "ok", (sql, start, length)
        dev = chq.DeviceRecordBatch.from_host(sl, ctx)
        e = parse_expr("x1 < x2 and (w or h >= k)")
        assert batches_identical(chq.filter_record(dev, al, e, ctx=ctx).to_host(), O.filter_record(sl, al, e))


def test_typed_columns_through_projection_and_groups(ctx):
    """the same operations behind the other entry points: projection outputs of type Float16, the fused filter + projection,
    and a device-resident batch group (per-batch fallback: typed operations materialise temporaries per batch)"""
    rec = make_typed_batch(30_000, 3)
    al = empty_aliases(rec)
    sel = parse_select("select h + k as s, h * f32 as p, x1 < x2 as lt, d1, w from t where (t1 <= t2 or w) and h >= k")
    exp = O.project_record(sel.projection, O.filter_record(rec, al, sel.selection), al)
    dev = chq.DeviceRecordBatch.from_host(rec, ctx)
    two = chq.project_record(sel.projection, chq.filter_record(dev, al, sel.selection, ctx=ctx), al, ctx=ctx).to_host()
    assert batches_identical(two, exp, nan_payload=True), explain_diff(two, exp)
    one = chq.filter_project_record(sel.selection, sel.projection, dev, al, ctx=ctx).to_host()
    assert batches_identical(one, exp, nan_payload=True), explain_diff(one, exp)
    parts = [make_typed_batch(5000, 10 + i) for i in range(6)]
    devs = [chq.DeviceRecordBatch.from_host(p, ctx) for p in parts]
    for sql in ["d1 < d2", "x1 >= x2 and i32 > 0", "h < k or w", "h + k > f32"]:
        e = parse_expr(sql)
        outs = chq.filter_records(devs, al, e, ctx=ctx)
        for p, o in zip(parts, outs):
            want = O.filter_record(p, al, e)
            assert batches_identical(o.to_host(), want), f"{sql}:\n{explain_diff(o.to_host(), want)}"


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("tile_kind", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 2048, 16384, 16385, 40_000])
def test_every_kernel_instantiation(n, tile_kind, split):
    """split=True forces the large-batch launch structure (complete tiles in the FULL-only instantiation, the
    incomplete tail tile in a second launch that continues the chained scan) on small inputs."""
    c = chq.Context(0)
    c.set_option("tile_kind", tile_kind)
    if split:
        c.set_option("split_rows", 1)
    rec = make_batch(n, 31 * n + tile_kind)
    al = empty_aliases(rec)
    for sql in ["f32 > 10.0", "i32 % 2 = 0 and u8 < 200", "flag or i16 * 2 > small", "k <> 'k3' and f32 / 3.0 < 20.0"]:
        assert check_same(c, rec, al, sql, "filter") == "ok", sql
    for sql in ["i32 * small + 1", "f32 * 2.0 + i8", "u16 % 7 + u8", "i16 > 5 or flag"]:
        assert check_same(c, rec, al, sql, "value") == "ok", sql
    c.close()


def test_sliced_inputs_keep_their_offsets(ctx):
    """Arrow slices: value buffers start at base + offset*width, bitmaps carry a bit offset."""
    rec = make_batch(5000, 42)
    for start, length in [(1, 100), (7, 2049), (63, 64), (64, 4000), (4999, 1), (13, 0)]:
        sl = rec.slice(start, length)
        al = empty_aliases(sl)
        for sql in ["f32 > 0.0 and flag", "s >= 'ab' or i64 % 3 = 0", "flag2 = flag"]:
            assert check_same(ctx, sl, al, sql, "filter") == "ok", (start, length, sql)
        assert check_same(ctx, sl, al, "i16 + i32", "value") == "ok"


def test_device_resident_pipeline(ctx):
    """filter -> project with the batch staying in HBM between the two operators (the reference's
    filter -> exchange -> materialize sequence)."""
    rec = make_batch(30_000, 9, nulls=False)
    al = empty_aliases(rec)
    sel = parse_select("select i32, k, i32 + 10.0 as p10, (f32 + 10) / 100 as v2, 1.0 / small as v3, small * small as v5 from t where i32 > 25 + 0.0")
    dev = chq.DeviceRecordBatch.from_host(rec, ctx)
    f_dev = chq.filter_record(dev, al, sel.selection, ctx=ctx)
    assert isinstance(f_dev, chq.DeviceRecordBatch)
    p_dev = chq.project_record(sel.projection, f_dev, al, ctx=ctx)
    got = p_dev.to_host()
    f_exp = O.filter_record(rec, al, sel.selection)
    exp = O.project_record(sel.projection, f_exp, al)
    assert batches_identical(got, exp, nan_payload=True), explain_diff(got, exp)
    fused = chq.filter_project_record(sel.selection, sel.projection, dev, al, ctx=ctx).to_host()
    assert batches_identical(fused, exp, nan_payload=True)
    assert batches_identical(dev.to_host(), rec)   # inputs are never modified


def test_wide_schema_needs_several_launches(ctx):
    rng = np.random.default_rng(3)
    n = 5000
    arrays = [pa.array(rng.integers(0, 1000, n).astype(np.int32)) for _ in range(95)]
    arrays += [pa.array(rng.random(n)), pa.array(rng.integers(0, 2, n).astype(bool)), pa.array(["x%d" % i for i in range(n)])]
    rec = pa.RecordBatch.from_arrays(arrays, names=[f"c{i}" for i in range(len(arrays))])
    al = empty_aliases(rec)
    e = parse_expr("c0 % 3 = 0 or c95 > 0.9")
    got = chq.filter_record(rec, al, e, ctx=ctx)
    assert batches_identical(got, O.filter_record(rec, al, e))
    assert ctx.last_stats()["launches"] >= 3


def test_opaque_fixed_width_types_pass_through(ctx):
    import datetime
    import decimal
    n = 3000
    rng = np.random.default_rng(5)
    rec = pa.RecordBatch.from_arrays([
        pa.array(np.arange(n, dtype=np.int32)),
        pa.array([datetime.date(2020, 1, 1) + datetime.timedelta(days=int(d)) for d in rng.integers(0, 3000, n)], mask=rng.random(n) < 0.1),
        pa.array(rng.integers(0, 10**15, n), type=pa.timestamp("us")),
        pa.array([decimal.Decimal(int(v)) / 100 for v in rng.integers(-10**9, 10**9, n)], type=pa.decimal128(20, 2)),
    ], names=["id", "d", "ts", "dec"])
    al = empty_aliases(rec)
    e = parse_expr("id % 5 <> 0")
    got = chq.filter_record(rec, al, e, ctx=ctx)
    exp = rec.filter(pa.array(np.arange(n) % 5 != 0))
    assert got.equals(exp)


def test_simple_sql_on_both_sample_data_sets(ctx):
    """Config 1: sample_queries/simple.sql; query 2 reads the 100-char wide-string data set."""
    from chapterhouseqe_amd.sample_data import simple_batches
    from .helpers import load_simple_sql
    SIMPLE_SQL = load_simple_sql()
    sets = {"simple": simple_batches(100, 8, 33), "simple_wide_string": simple_batches(100, 100, 33)}
    expect_ids = [list(range(0, 25)), list(range(26, 100)), list(range(0, 75)), list(range(26, 100)), list(range(0, 100, 2))]
    for sel, ids in zip(parse_statements(SIMPLE_SQL), expect_ids):
        name = "simple_wide_string" if "wide" in sel.from_.args[0] else "simple"
        got_ids = []
        for b in sets[name]:
            al = chq.get_record_table_aliases(sel.from_.alias, b)
            f = chq.filter_record(b, al, sel.selection, ctx=ctx)
            p = chq.project_record(sel.projection, f, al, ctx=ctx)
            exp = O.project_record(sel.projection, O.filter_record(b, al, sel.selection), al)
            assert batches_identical(p, exp, nan_payload=True), explain_diff(p, exp)
            got_ids += p.column(0).to_pylist()
        assert got_ids == ids


def test_wide_strings_mid_selectivity(ctx):
    """Config 4 shape: id:Int32, value1:Utf8(100 chars), value2:Float32; id > n/2 and a Utf8 equality."""
    from chapterhouseqe_amd.sample_data import simple_table
    n = 40_000
    rec = simple_table(n, 100, seed=11)
    al = empty_aliases(rec)
    for sql in [f"id > {n // 2}", "id > 25", "value2 < 10.0"]:
        e = parse_expr(sql)
        assert batches_identical(chq.filter_record(rec, al, e, ctx=ctx), O.filter_record(rec, al, e)), sql
    target = rec.column(1)[1234].as_py()
    e = parse_expr(f"value1 = '{target}'")
    got = chq.filter_record(rec, al, e, ctx=ctx)
    assert got.num_rows >= 1 and batches_identical(got, O.filter_record(rec, al, e))


def test_errors_leave_no_partial_output_and_context_stays_usable(ctx):
    rec = make_batch(10_000, 1)
    al = empty_aliases(rec)
    with pytest.raises(chq.ChqError) as ei:
        chq.filter_record(rec, al, parse_expr("i32 * i32 * i32 > 0"), ctx=ctx)
    assert ei.value.code == 20
    with pytest.raises(chq.ChqError) as ei:
        chq.filter_record(rec, al, parse_expr("i32 / (small % 1) > 0"), ctx=ctx)
    assert ei.value.code == 21
    assert check_same(ctx, rec, al, "i32 > 0", "filter") == "ok"


def test_malformed_arrays_are_refused_on_the_host(ctx):
    """a kernel must never see a null data pointer: arrays without the buffers their length needs fail with
    ArrowError::InvalidArgument before anything is launched"""
    good = chq.DeviceRecordBatch.from_host(pa.RecordBatch.from_arrays([pa.array(np.arange(100, dtype=np.int32))], names=["a"]), ctx)
    addr = good.column_buffer_address(0, 1)
    e = parse_expr("a > 5")
    cases = [
        [{"name": "a", "format": "i", "values": 0}],                                              # no values buffer
        [{"name": "a", "format": "i", "values": addr, "null_count": 3, "validity": 0}],          # nulls without a bitmap
        [{"name": "a", "format": "i", "values": addr}, {"name": "s", "format": "u", "values": 0}],  # Utf8 without offsets
    ]
    for cols in cases:
        with pytest.raises(chq.ChqError) as ei:     # chq_wrap_columns refuses them ...
            chq.DeviceRecordBatch.from_device_buffers(cols, 100, ctx)
        assert ei.value.code == 22, cols
    # ... and so does the import of hand-built Arrow structs on every record call
    import ctypes as C
    cb = good._cb
    child = cb.array.array.children[0].contents
    saved = child.buffers[1]
    child.buffers[1] = None
    try:
        for call in (lambda: chq.filter_record(good, [[]], e, ctx=ctx), lambda: chq.filter_records([good, good], [[]], e, ctx=ctx),
                     lambda: chq.compute_value(good, [[]], e, ctx=ctx)):
            with pytest.raises(chq.ChqError) as ei:
                call()
            assert ei.value.code == 22
    finally:
        child.buffers[1] = saved
    del C
    assert chq.filter_record(good, [[]], e, ctx=ctx).num_rows == 94


def test_power_of_two_literal_divisors(ctx):
    """x / 2^k and x % 2^k with a literal divisor take a shift/mask path on the device: truncation toward zero and the
    sign of the dividend must survive it, for every 32-bit-class integer type and the extremes"""
    rng = np.random.default_rng(17)
    n = 3000

    def col(dt, lo, hi):
        v = rng.integers(lo, hi, n, endpoint=True).astype(dt)
        v[:6] = np.array([lo, hi, 0, 1, lo + 1, hi - 1]).astype(dt)
        if lo < 0:
            v[6] = -1
        return pa.array(v)

    rec = pa.RecordBatch.from_arrays(
        [col(np.int8, -128, 127), col(np.int16, -32768, 32767), col(np.int32, -2**31, 2**31 - 1),
         col(np.uint8, 0, 255), col(np.uint16, 0, 65535), col(np.uint32, 0, 2**32 - 1)],
        names=["i8", "i16", "i32", "u8", "u16", "u32"])
    al = empty_aliases(rec)
    for c, divisors in [("i8", [1, 2, 64]), ("i16", [1, 2, 4096]), ("i32", [1, 2, 8, 65536, 1073741824]),
                        ("u8", [1, 2, 128]), ("u16", [1, 2, 32768])]:   # (UInt32 has no common type with an Int32 literal)
        for d in divisors:
            for sql in [f"{c} / {d}", f"{c} % {d}", f"({c} % {d}) = 0", f"{c} / {d} * {d} + {c} % {d} = {c}"]:
                assert check_same(ctx, rec, al, sql, "value") == "ok", sql
    # not powers of two, reversed operands, zero: the general path (and its errors) as before
    for sql, want in [("i32 % 3", "ok"), ("i32 / 7", "ok"), ("64 / u8", "error"), ("i32 % 0", "error"), ("u16 / 0", "error")]:
        assert check_same(ctx, rec, al, sql, "value") == want, sql


def wide_numeric_batch(n, seed, ncols=16):
    rng = np.random.default_rng(seed)
    cols = {f"c{i}": pa.array(rng.integers(1, 50, n).astype(np.int32)) for i in range(ncols)}
    cols["x"] = pa.array((rng.random(n) * 10 + 1).astype(np.float32))
    cols["y"] = pa.array(rng.random(n) * 10 + 1)
    cols["z"] = pa.array(np.zeros(n, dtype=np.int32))
    cols["big"] = pa.array(np.full(n, 2**31 - 1, dtype=np.int32))
    return pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols.keys()))


OVERSIZED = [
    # more numeric temporaries than a program has
    "((c0 + c1) * (c2 + c3)) / ((c4 + c5) * (c6 + c7)) + ((c0 + c2) * (c1 + c3)) * ((c4 + c6) * (c5 + c7))",
    "((x + y) * (y + x)) / ((x + 1.0) * (y + 2.0)) + ((x * y) + (y / x)) * ((x + x) / (y + y))",
    # more distinct columns than a program can reference
    " + ".join(f"c{i}" for i in range(16)),
    # more instructions than a program holds
    " + ".join(f"c{i % 16} * {i + 2} % {i + 3}" for i in range(30)),
]
OVERSIZED_PREDICATES = [
    " + ".join(f"c{i}" for i in range(16)) + " > 400",
    # more boolean temporaries than a program has
    "((c0 > c1 or c2 > c3) and (c4 > c5 or c6 > c7)) or ((c8 > c9 or c10 > c11) and (c12 > c13 or (c14 > c15 and (c0 > c2 or (c1 > c3 and (c4 > c6 or c5 > c7))))))",
    "((c0 + c1) * (c2 + c3)) / ((c4 + c5) * (c6 + c7)) > ((c8 + c9) * (c10 + c11)) / ((c12 + c13) * (c14 + c15))",
]


@pytest.mark.parametrize("n", [1, 777, 20_000])
def test_expressions_larger_than_one_device_program(ctx, n):
    """the reference has no size limits (one arrow kernel per AST node); here sub-trees become temporary columns until
    the rest fits one program -- values, and the error an arrow-rs evaluation would hit first, must not change"""
    rec = wide_numeric_batch(n, 4000 + n)
    al = empty_aliases(rec)
    for sql in OVERSIZED:
        assert check_same(ctx, rec, al, sql, "value") == "ok", sql
    for sql in OVERSIZED_PREDICATES:
        assert check_same(ctx, rec, al, sql, "filter") == "ok", sql
        assert check_same(ctx, rec, al, sql, "value") == "ok", sql
    # two different data-dependent errors: the small left operand (division by zero) is evaluated first by the
    # reference, although the large right operand (overflow) is the part that gets materialised first here
    huge_overflow = "((big + c0) * (c2 + c3)) / ((c4 + c5) * (c6 + c7)) + ((c0 + c2) * (c1 + c3)) * ((c4 + c6) * (c5 + c7))"
    for sql in [f"(c0 / z) + ({huge_overflow})", f"({huge_overflow}) + (c0 / z)", f"(c0 / z) + ({huge_overflow}) > 0"]:
        assert check_same(ctx, rec, al, sql, "value") == "error", sql
    assert check_same(ctx, rec, al, f"(c0 / z) + ({huge_overflow}) > 0", "filter") == "error"
    # a projection mixing ordinary and oversized items
    sel = parse_select(f"select c0, {OVERSIZED[0]} as big1, c1 + 1 as small, {OVERSIZED[2]} as sum16 from t")
    got = chq.project_record(sel.projection, rec, al, ctx=ctx)
    exp = O.project_record(sel.projection, rec, al)
    assert batches_identical(got, exp, nan_payload=True), explain_diff(got, exp)


def test_many_nullable_and_string_columns(ctx):
    """40 nullable Int32, 20 Utf8 (half of them nullable) and 6 Boolean columns: the follow-up kernels report through a
    fixed-size header (16 null counters, 8 Utf8 spans per round), wider batches take several rounds"""
    n = 3000
    rng = np.random.default_rng(23)
    arrays, names = [], []
    for i in range(40):
        arrays.append(pa.array(rng.integers(0, 100, n).astype(np.int32), mask=rng.random(n) < (0.02 * (i % 5))))
        names.append(f"n{i}")
    words = np.array(["", "a", "bb", "ccc", "a somewhat longer string " * 3])
    for i in range(20):
        arrays.append(pa.array(words[rng.integers(0, len(words), n)], type=pa.utf8(), mask=(rng.random(n) < 0.1) if i % 2 else None))
        names.append(f"s{i}")
    for i in range(6):
        arrays.append(pa.array(rng.integers(0, 2, n).astype(bool), mask=(rng.random(n) < 0.2) if i % 3 == 0 else None))
        names.append(f"b{i}")
    rec = pa.RecordBatch.from_arrays(arrays, names=names)
    al = empty_aliases(rec)
    for sql in ["n1 > 50", "n39 > 20 and s19 <> 'a'", "b0 or n7 % 2 = 0", "s0 >= 'b' or b5"]:
        assert check_same(ctx, rec, al, sql, "filter") == "ok", sql
    # host groups of such batches go through the same code once, concatenated
    parts = [rec.slice(0, 1000), rec.slice(1000, 1500), rec.slice(2500, 500)]
    e = parse_expr("n39 > 20 and s19 <> 'a'")
    got = chq.filter_records(parts, al, e, ctx=ctx)
    for g, p in zip(got, parts):
        exp = O.filter_record(p, al, e)
        assert batches_identical(g, exp), explain_diff(g, exp)


def test_default_nan_of_invalid_operations_matches_the_host(ctx):
    """0/0, inf + -inf, 0 * inf, fmod(x, 0): the NaN an invalid operation produces carries the sign of the reference's
    host (x86-64: 0xFFC00000 / 0xFFF8...), because totalOrder comparisons and filters downstream depend on it -- bit for
    bit against the oracle, which runs on that host"""
    inf = np.float32(np.inf)
    x = np.array([0.0, 1.0, inf, -inf, 0.0, 5.5, -0.0, inf], dtype=np.float32)
    y = np.array([0.0, 0.0, -inf, inf, inf, 0.0, 0.0, inf], dtype=np.float32)
    rec = pa.RecordBatch.from_arrays([pa.array(x), pa.array(y), pa.array(x.astype(np.float64)), pa.array(y.astype(np.float64)),
                                      pa.array(np.arange(8, dtype=np.int32))], names=["x", "y", "dx", "dy", "id"])
    al = empty_aliases(rec)
    for sql in ["x / y", "x + y", "x * y", "x % y", "dx / dy", "dx + dy", "dx * dy", "dx % dy", "id % y", "0.0 / x"]:
        e = parse_expr(sql)
        got = chq.compute_value(rec, al, e, ctx=ctx)[0]
        exp = O.compute_value(rec, al, e)[0]
        assert arrays_identical(got, exp, nan_payload=True), f"{sql}: {got.to_pylist()} vs {exp.to_pylist()}"
    for sql in ["x / y < 1.0", "x % y < 37.0 / x", "dx * dy >= dx", "x + y = x + y"]:
        assert check_same(ctx, rec, al, sql, "filter") == "ok", sql
    # NaN operands of either sign and payload against literals (the literal paths skip the fix-ups when they cannot matter)
    nans = np.frombuffer(np.array([0x7FC00000, 0xFFC00000, 0x7F800001, 0xFFA00123, 0x7FFFFFFF, 0x3F800000], dtype=np.uint32).tobytes(), dtype=np.float32)
    rn = pa.RecordBatch.from_arrays([pa.array(nans), pa.array(nans[::-1].copy())], names=["x", "y"])
    aln = empty_aliases(rn)
    for sql in ["x / 3.0", "3.0 / x", "x * 2.0", "x + 10.0", "x % 3.0", "3.0 % x", "x / 0.0", "x * 0.0", "x + y", "x * y", "x / y", "x % y"]:
        e = parse_expr(sql)
        got = chq.compute_value(rn, aln, e, ctx=ctx)[0]
        exp = O.compute_value(rn, aln, e)[0]
        assert arrays_identical(got, exp, nan_payload=True), f"{sql}: {[hex(v) for v in np.frombuffer(got.buffers()[1], dtype=np.uint32)[:6]]} vs {[hex(v) for v in np.frombuffer(exp.buffers()[1], dtype=np.uint32)[:6]]}"


@pytest.mark.parametrize("n", [2, 65, 2049, 10_000, 100_000, 262_144, 262_145])
def test_small_host_batches_take_the_single_sync_path(ctx, n):
    """the reference's calling pattern: a fixed-width host batch in, a host batch out -- staged through one pinned block
    each way (engine.cpp:filter_record_small_host); identical results with the path switched off"""
    rng = np.random.default_rng(n)
    rec = pa.RecordBatch.from_arrays(
        [pa.array(np.arange(n, dtype=np.int32)), pa.array((rng.random(n) * 100).astype(np.float32)),
         pa.array(rng.integers(-100, 100, n).astype(np.int8)), pa.array(rng.random(n) * 10),
         pa.array(rng.integers(0, 2**40, n).astype(np.int64)), pa.array(rng.integers(0, 60000, n).astype(np.uint16))],
        names=["id", "v", "b", "d", "l", "h"])
    al = empty_aliases(rec)
    off = chq.Context(0)
    off.set_option("small_host", 0)
    for sql in ["v > 10.0", "id % 2 = 0 and b < 0", "d * 2.0 > 5.0 or l % 3 = 0", "h > 70000", "id = id", "b + id > 100 and v < 50.0"]:
        e = parse_expr(sql)
        exp = O.filter_record(rec, al, e)
        got = chq.filter_record(rec, al, e, ctx=ctx)
        st = ctx.last_stats()
        assert batches_identical(got, exp), f"{sql} (n={n}):\n{explain_diff(got, exp)}"
        assert st["rows_in"] == n and st["rows_out"] == exp.num_rows
        if n <= 262_144:
            assert st["launches"] == 1
        assert batches_identical(chq.filter_record(rec, al, e, ctx=off), exp)
    if n > 10:   # Arrow slices (non-zero offsets) of host arrays go through the same path
        sl = rec.slice(3, n - 5)
        e = parse_expr("v > 10.0 and id % 2 = 0")
        got = chq.filter_record(sl, al, e, ctx=ctx)
        assert batches_identical(got, O.filter_record(sl, al, e)), n
    # data-dependent errors are the general path's to report, unchanged
    for sql, code in [("id * 100000 > 0", 20 if n > 21475 else None), ("v > 1.0 and 10 / (id % 2) > 1", 21)]:
        if code is None:
            continue
        with pytest.raises(chq.ChqError) as ei:
            chq.filter_record(rec, al, parse_expr(sql), ctx=ctx)
        assert ei.value.code == code, sql
    assert check_same(ctx, rec, al, "nope > 1", "filter") == "error"
    off.close()


def test_a_short_run_of_the_long_fuzz(monkeypatch):
    """tests/fuzz_long.py (random predicates, values, projections, groups, schemas; GPU vs oracle, strict NaN bits) for a
    few seconds with a fixed seed -- the long runs are recorded in profiles/r1/fuzz_long.txt"""
    import sys
    from . import fuzz_long
    monkeypatch.setattr(sys, "argv", ["fuzz_long", "8", "123"])
    assert fuzz_long.main() == 0
