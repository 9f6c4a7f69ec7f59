"""CPU: the host half of the library -- csrc/plan.cpp's typing, coercion table, literal typing, alias lookup, constant
folding, length rules and error codes -- through `chq_plan_describe` (no GPU, no context), against the CPU oracle's
compute_value on the same inputs: same result type, same scalar flag, same length class, same folded constants, same
static status codes.  (The device half is covered by the `-m gpu` tier.)"""
import struct

import numpy as np
import pyarrow as pa
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr
from oracle import oracle as O

from . import rules
from .cases import empty_aliases
from .helpers import load_golden

TYPE_NAMES = {pa.bool_(): "Boolean", pa.int8(): "Int8", pa.int16(): "Int16", pa.int32(): "Int32", pa.int64(): "Int64",
              pa.uint8(): "UInt8", pa.uint16(): "UInt16", pa.uint32(): "UInt32", pa.uint64(): "UInt64",
              pa.float16(): "Float16", pa.float32(): "Float32", pa.float64(): "Float64", pa.utf8(): "Utf8"}
DATA_DEPENDENT = {20, 21}   # overflow / divide by zero: only the data can tell (unless the operands are literals)


MINUS = {"on": False}   # tests of the opt-in `enable_minus` option flip this (and the oracle's extension mode) together


def describe(rec, al, sql):
    try:
        return 0, chq.plan_describe(rec.schema, al, parse_expr(sql), rec.num_rows, enable_minus=MINUS["on"])
    except chq.ChqError as e:
        return e.code, str(e)


def oracle_value(rec, al, sql):
    try:
        arr, is_scalar = O.compute_value(rec, al, parse_expr(sql))
        return 0, arr, is_scalar
    except O.OracleError as e:
        return e.code, None, None


def value_bits(arr):
    v = arr[0].as_py()
    t = arr.type
    if t == pa.float32():
        return struct.unpack("<I", struct.pack("<f", v))[0]
    if t == pa.float64():
        return struct.unpack("<Q", struct.pack("<d", v))[0]
    if t == pa.bool_():
        return int(v)
    return v & 0xFFFFFFFFFFFFFFFF


def check(rec, al, sql):
    code, text = describe(rec, al, sql)
    ocode, arr, is_scalar = oracle_value(rec, al, sql)
    if ocode in DATA_DEPENDENT and code != ocode:
        # the host half cannot see the data: it reports OK, or the static error the engine holds back until the sub-trees
        # evaluated before it have run on the device (TypedExpr::pending_code)
        return "data-dependent"
    if ocode == 30 and code == 30:
        return "unsupported"
    assert code == ocode, f"{sql}: oracle status {ocode}, host planner status {code} ({text})"
    if code:
        return "error"
    head = text.splitlines()[0].split()
    assert head[0] == "result" and head[1] == TYPE_NAMES[arr.type], f"{sql}: {head} vs {arr.type}"
    assert head[2] == f"scalar={int(bool(is_scalar))}", f"{sql}: {head} vs is_scalar={is_scalar}"
    if rec.num_rows != 1:
        assert head[3] == f"len1={int(len(arr) == 1)}", f"{sql}: {head} vs len {len(arr)}"
    if head[3] == "len1=1" and arr.type != pa.utf8() and arr[0].is_valid:
        got = int(text.splitlines()[1].split()[1], 16)
        width = {pa.bool_(): 1}.get(arr.type, arr.type.bit_width)
        mask = (1 << width) - 1 if width < 64 else 0xFFFFFFFFFFFFFFFF
        assert got & mask == value_bits(arr) & mask, f"{sql}: folded {got:#x} vs oracle {value_bits(arr):#x}"
    return "ok"


@pytest.mark.parametrize("rule", [r for r in rules.RULES if r[2] in ("value", "filter", "error")], ids=lambda r: r[0])
def test_rules_table_through_the_host_planner(rule):
    _, factory, _, sql, _ = rule
    rec = factory()
    check(rec, empty_aliases(rec), sql)


@pytest.fixture
def minus_mode():
    MINUS["on"] = True
    with O.extension_minus():
        yield
    MINUS["on"] = False


@pytest.mark.parametrize("rule", rules.MINUS_RULES, ids=lambda r: r[0])
def test_minus_rules_through_the_host_planner(rule, minus_mode):
    """`enable_minus` (not reference behaviour): typing, folding and static errors of Minus vs the oracle's extension mode"""
    _, factory, _, sql, _ = rule
    rec = factory()
    check(rec, empty_aliases(rec), sql)


def test_minus_every_pair_of_types_and_literals(minus_mode):
    rec = typed_batch(4)
    al = empty_aliases(rec)
    operands = list(COLS) + ["3", "2.5", "3000000000", "'x1'", "true"]
    counts = {}
    for a in operands:
        for b in operands:
            r = check(rec, al, f"{a} - {b}")
            counts[r] = counts.get(r, 0) + 1
    assert counts["ok"] > 50 and counts["error"] > 50, counts


def test_minus_is_rejected_without_the_option():
    rec = typed_batch(4)
    code, _ = describe(rec, empty_aliases(rec), "i32 - 1")
    assert code == 3   # BinaryOperatorNotImplemented (compute_value.rs:210-216)


def test_reference_vectors_type_the_same_way():
    """the expressions of the reference's own tests (tests/golden/reference_cases.json)"""
    from .helpers import batch_from_json, expr_from_json
    n = 0
    for case in load_golden("reference_cases.json")["cases"]:
        if "expr" not in case or "schema" not in case:
            continue
        rec = batch_from_json(case["schema"], case["columns"])
        e = expr_from_json(case["expr"])
        al = case["table_aliases"]
        try:
            text = chq.plan_describe(rec.schema, al, e, rec.num_rows)
            code = 0
        except chq.ChqError as err:
            code, text = err.code, str(err)
        try:
            arr, is_scalar = O.compute_value(rec, al, e)
            ocode = 0
        except O.OracleError as err:
            ocode = err.code
        assert code == ocode, (case["name"], code, ocode, text)
        if not code:
            head = text.splitlines()[0].split()
            assert head[1] == TYPE_NAMES[arr.type] and head[2] == f"scalar={int(bool(is_scalar))}", (case["name"], head)
        n += 1
    assert n >= 8


COLS = {"i8": pa.int8(), "i16": pa.int16(), "i32": pa.int32(), "i64": pa.int64(), "u8": pa.uint8(), "u16": pa.uint16(),
        "u32": pa.uint32(), "u64": pa.uint64(), "f32": pa.float32(), "f64": pa.float64(), "flag": pa.bool_(), "s": pa.utf8()}


def typed_batch(n):
    arrays = []
    for name, t in COLS.items():
        if t == pa.utf8():
            arrays.append(pa.array(["x%d" % i for i in range(n)], t))
        elif t == pa.bool_():
            arrays.append(pa.array([i % 2 == 0 for i in range(n)], t))
        else:
            arrays.append(pa.array(np.arange(1, n + 1), t))   # small positive values: no overflow, no zero divisors
    return pa.RecordBatch.from_arrays(arrays, names=list(COLS))


def test_every_pair_of_types_and_literals():
    """the whole coercion table (compute_value.rs:350-431) x arithmetic / comparison / and-or, columns and literals"""
    rec = typed_batch(4)
    al = empty_aliases(rec)
    operands = list(COLS) + ["3", "2.5", "3000000000", "'x1'", "true"]
    counts = {}
    for a in operands:
        for b in operands:
            for op in ["+", "*", "/", "%", "<", "=", "and", "or"]:
                r = check(rec, al, f"{a} {op} {b}")
                counts[r] = counts.get(r, 0) + 1
    assert counts["ok"] > 400 and counts["error"] > 400, counts


def test_every_pair_of_the_round_3_types():
    """Float16, temporal and decimal columns against each other, against the numeric / Boolean / Utf8 columns and against
    literals, under every operator: the host planner answers what the oracle answers (status and result type) -- the
    `left == right` arm and the Float16 rows of get_common_type (compute_value.rs:355, :387-390), arrow-arith's refusals for
    dates / times, arrow-cast's refusals under AND / OR"""
    import decimal
    n = 4
    ints32, ints64 = pa.array(np.arange(1, n + 1), pa.int32()), pa.array(np.arange(1, n + 1), pa.int64())
    cols = {
        "h": pa.array(np.arange(1, n + 1).astype(np.float16), pa.float16()), "h2": pa.array(np.arange(2, n + 2).astype(np.float16), pa.float16()),
        "d32": ints32.view(pa.date32()), "d32b": ints32.view(pa.date32()), "d64": ints64.view(pa.date64()),
        "ts": ints64.view(pa.timestamp("s")), "ts2": ints64.view(pa.timestamp("s")), "tms": ints64.view(pa.timestamp("ms")),
        "tz": ints64.view(pa.timestamp("s", tz="UTC")), "t32": ints32.view(pa.time32("s")), "t64": ints64.view(pa.time64("us")),
        "du": ints64.view(pa.duration("ms")), "du2": ints64.view(pa.duration("ms")),
        "dec": pa.array([decimal.Decimal(i) for i in range(1, n + 1)], pa.decimal128(20, 2)),
        "dec2": pa.array([decimal.Decimal(i) for i in range(2, n + 2)], pa.decimal128(20, 2)),
        "dec3": pa.array([decimal.Decimal(i) for i in range(1, n + 1)], pa.decimal128(20, 3)),
        "i32": ints32, "i64": ints64, "f32": pa.array(np.arange(1, n + 1).astype(np.float32)), "f64": pa.array(np.arange(1, n + 1).astype(np.float64)),
        "flag": pa.array([True, False, True, True]), "s": pa.array(["true", "no", "x", "1"]),
    }
    rec = pa.RecordBatch.from_arrays(list(cols.values()), names=list(cols))
    al = empty_aliases(rec)
    new = ["h", "h2", "d32", "d32b", "d64", "ts", "ts2", "tms", "tz", "t32", "t64", "du", "du2", "dec", "dec2", "dec3"]
    others = ["i32", "i64", "f32", "f64", "flag", "s", "3", "2.5", "'x'", "true"]
    counts = {}
    for a in new:
        for b in new + others:
            for op in ["+", "*", "/", "%", "<", "=", "<>", ">=", "and", "or"]:
                for sql in (f"{a} {op} {b}", f"{b} {op} {a}"):
                    r = check(rec, al, sql)
                    counts[r] = counts.get(r, 0) + 1
    assert counts["ok"] > 150 and counts["error"] > 2000 and counts.get("unsupported", 0) > 10, counts


def test_random_expressions_type_the_same_way():
    from .test_gpu_parity import random_numeric, random_predicate
    rng = np.random.default_rng(5)
    rec = pa.RecordBatch.from_arrays(
        [pa.array(np.arange(1, 6), t) for t in (pa.int8(), pa.int16(), pa.int32(), pa.int64(), pa.uint8(), pa.uint16(), pa.uint32(), pa.uint64(), pa.float32(), pa.float64(), pa.int32())] +
        [pa.array([True, False, True, True, False]), pa.array([False] * 5), pa.array(list("abcde")), pa.array(["k1"] * 5)],
        names=["i8", "i16", "i32", "i64", "u8", "u16", "u32", "u64", "f32", "f64", "small", "flag", "flag2", "s", "k"])
    al = empty_aliases(rec)
    counts = {}
    for _ in range(400):
        sql = random_numeric(rng, 3) if rng.random() < 0.5 else random_predicate(rng, 2)
        r = check(rec, al, sql)
        counts[r] = counts.get(r, 0) + 1
    assert counts.get("ok", 0) > 100, counts


def test_alias_qualified_lookup_and_length_rules():
    rec = typed_batch(3)
    al = [["t"]] * rec.num_columns
    for sql in ["t.i32 + 1", "x.i32 + 1", "t.nope", "a.b.c", "i32", "1", "i32 and true", "flag and flag", "flag or (1 = 1)",
                "(1 = 1) and (2 = 2)", "i32 = i32 and 1 = 1"]:
        check(rec, al, sql)
    one = typed_batch(1)       # a 1-row batch: len-1 literal arrays and columns have the same length
    for sql in ["flag and true", "i32 = i32 and 1 = 1", "i32 + 1"]:
        check(one, empty_aliases(one), sql)
    with pytest.raises(chq.ChqError):
        chq.plan_describe(pa.schema([("l", pa.list_(pa.int32()))]), None, parse_expr("l"))
