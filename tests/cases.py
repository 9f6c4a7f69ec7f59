"""Case runners shared by the CPU (oracle) and GPU (libchq) test files.  `impl` is any object exposing the
reference's three entry points -- compute_value(rec, aliases, expr) -> (array, is_scalar),
filter_record(rec, aliases, expr), project_record(fields, rec, aliases) -- and raising an exception with a
`.code` attribute (chq_status numbering) on failure: the oracle module and the chapterhouseqe_amd package
both qualify, so the same expectations check both."""
from __future__ import annotations

import pyarrow as pa

from chapterhouseqe_amd.sqlparse import parse_expr, parse_select

from .helpers import arrays_identical, batch_from_json, batches_identical, explain_diff, expr_from_json, pa_type


def empty_aliases(rec):
    return [[] for _ in range(rec.num_columns)]


def run_golden_case(impl, case):
    rec = batch_from_json(case["schema"], case["columns"])
    expr = expr_from_json(case["expr"])
    aliases = case["table_aliases"]
    if case["kind"] == "compute_value":
        arr, is_scalar = impl.compute_value(rec, aliases, expr)
        exp = pa.array(case["expected"]["values"], type=pa_type(case["expected"]["type"]))
        assert arrays_identical(arr, exp), f"{case['name']}: got {arr.to_pylist()} expected {exp.to_pylist()}"
        if "expected_is_scalar" in case:
            assert is_scalar == case["expected_is_scalar"], case["name"]
    elif case["kind"] == "filter_record":
        out = impl.filter_record(rec, aliases, expr)
        exp = batch_from_json(case["schema"], case["expected_columns"])
        assert out.schema.equals(rec.schema), f"{case['name']}: schema changed"
        assert batches_identical(out, exp), f"{case['name']}:\n{explain_diff(out, exp)}"
    else:
        raise ValueError(case["kind"])


def _expect_error(fn, code, name):
    try:
        fn()
    except Exception as e:  # noqa: BLE001
        assert getattr(e, "code", None) == code, f"{name}: expected status {code}, got {e!r}"
        return
    raise AssertionError(f"{name}: expected status {code}, call succeeded")


def run_rule(impl, rule):
    name, factory, kind, sql, expect = rule
    rec = factory()
    aliases = empty_aliases(rec)
    expr = parse_expr(sql)
    if kind == "value":
        typ, values = expect
        arr, _ = impl.compute_value(rec, aliases, expr)
        exp = pa.array(values, type=typ)
        assert arrays_identical(arr, exp, nan_payload=False), f"{name} ({sql}): got {arr.to_pylist()} expected {exp.to_pylist()}"
    elif kind == "filter":
        out = impl.filter_record(rec, aliases, expr)
        exp = rec.take(pa.array(expect, pa.int64())) if expect else rec.slice(0, 0)
        assert out.schema.equals(rec.schema), f"{name}: schema changed"
        assert batches_identical(out, exp), f"{name} ({sql}):\n{explain_diff(out, exp)}"
    elif kind == "error":
        _expect_error(lambda: impl.compute_value(rec, aliases, expr), expect, name)
        _expect_error(lambda: impl.filter_record(rec, aliases, expr), expect, name + " (via filter_record)")
    elif kind == "error_filter":
        _expect_error(lambda: impl.filter_record(rec, aliases, expr), expect, name)
    else:
        raise ValueError(kind)


def run_project_rule(impl, rule):
    name, factory, sql, expect = rule
    rec = factory()
    sel = parse_select(sql)
    out = impl.project_record(sel.projection, rec, empty_aliases(rec))
    assert out.num_columns == len(expect), f"{name}: {out.schema}"
    for i, (cname, typ, values, nullable) in enumerate(expect):
        f = out.schema.field(i)
        assert f.name == cname, f"{name}: field {i} named {f.name!r}, expected {cname!r}"
        assert f.type == typ, f"{name}: field {cname} type {f.type}"
        assert f.nullable == nullable, f"{name}: field {cname} nullable={f.nullable}, expected {nullable}"
        exp = pa.array(values, type=typ)
        assert arrays_identical(out.column(i), exp, nan_payload=False), f"{name}: column {cname}: {out.column(i).to_pylist()} vs {values}"


def run_project_error(impl, rule):
    name, factory, sql, code = rule
    rec = factory()
    sel = parse_select(sql)
    _expect_error(lambda: impl.project_record(sel.projection, rec, empty_aliases(rec)), code, name)
