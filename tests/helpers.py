"""Shared helpers for the parity tests (tests only; the oracle is never imported by the product)."""
from __future__ import annotations

import json
import os

import numpy as np
import pyarrow as pa

from chapterhouseqe_amd import sqlast as A
from chapterhouseqe_amd.sqlparse import parse_expr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_TYPES = {"bool": pa.bool_(), "int8": pa.int8(), "int16": pa.int16(), "int32": pa.int32(), "int64": pa.int64(),
          "uint8": pa.uint8(), "uint16": pa.uint16(), "uint32": pa.uint32(), "uint64": pa.uint64(),
          "float32": pa.float32(), "float64": pa.float64(), "utf8": pa.utf8()}


def pa_type(name: str) -> pa.DataType:
    return _TYPES[name]


def expr_from_json(j) -> A.Expr:
    """JSON encoding of the literal sqlparser structs the reference's tests build."""
    if "sql" in j:
        return parse_expr(j["sql"])
    if "BinaryOp" in j:
        b = j["BinaryOp"]
        return A.BinaryOp(expr_from_json(b["left"]), A.BinaryOperator[b["op"]], expr_from_json(b["right"]))
    if "Identifier" in j:
        return A.ident(j["Identifier"])
    if "CompoundIdentifier" in j:
        return A.compound(*j["CompoundIdentifier"])
    if "Number" in j:
        return A.number(j["Number"][0], bool(j["Number"][1]))
    if "Boolean" in j:
        return A.boolean(j["Boolean"])
    if "SingleQuotedString" in j:
        return A.string(j["SingleQuotedString"])
    if "Nested" in j:
        return A.Nested(expr_from_json(j["Nested"]))
    raise ValueError(j)


def batch_from_json(schema, columns) -> pa.RecordBatch:
    fields = [pa.field(f["name"], pa_type(f["type"]), f.get("nullable", False)) for f in schema]
    arrays = [pa.array(col, type=f.type) for col, f in zip(columns, fields)]
    return pa.RecordBatch.from_arrays(arrays, schema=pa.schema(fields))


def load_golden(name: str):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def arrays_identical(a: pa.Array, b: pa.Array, nan_payload: bool = True) -> bool:
    """Bit-exact comparison (signed zeros included), nulls compared by position only.  nan_payload=False treats
    every NaN as equal: the sign/payload of a NaN *produced* by an invalid operation (0/0, inf-inf) is hardware
    defined (x86 SSE: 0xFFC00000, gfx950: 0x7FC00000), so computed floats are compared modulo NaN payload;
    copied data (filter outputs) is always compared with payloads."""
    if a.type != b.type or len(a) != len(b) or a.null_count != b.null_count:
        return False
    if len(a) == 0:
        return True
    va = np.asarray(a.is_valid())
    vb = np.asarray(b.is_valid())
    if not np.array_equal(va, vb):
        return False
    if pa.types.is_floating(a.type):
        w = {16: np.uint16, 32: np.uint32, 64: np.uint64}[a.type.bit_width]
        fa = a.fill_null(0).to_numpy(zero_copy_only=False)
        fb = b.fill_null(0).to_numpy(zero_copy_only=False)
        xa, xb = fa.view(w)[va], fb.view(w)[vb]
        if not nan_payload:
            na, nb = np.isnan(fa[va]), np.isnan(fb[vb])
            return bool(np.array_equal(na, nb) and np.array_equal(xa[~na], xb[~nb]))
        return bool(np.array_equal(xa, xb))
    if pa.types.is_temporal(a.type):   # raw values (a timestamp may lie outside datetime's range)
        raw = pa.int32() if a.type.bit_width == 32 else pa.int64()
        return a.view(raw).to_pylist() == b.view(raw).to_pylist()
    if pa.types.is_string(a.type) or pa.types.is_boolean(a.type) or pa.types.is_decimal(a.type):
        return a.to_pylist() == b.to_pylist()
    xa = a.fill_null(0).to_numpy(zero_copy_only=False)
    xb = b.fill_null(0).to_numpy(zero_copy_only=False)
    return bool(np.array_equal(xa[va], xb[vb]))


def batches_identical(a: pa.RecordBatch, b: pa.RecordBatch, check_nullable: bool = True, nan_payload: bool = True) -> bool:
    if a.num_columns != b.num_columns or a.num_rows != b.num_rows:
        return False
    for i in range(a.num_columns):
        fa, fb = a.schema.field(i), b.schema.field(i)
        if fa.name != fb.name or fa.type != fb.type:
            return False
        if check_nullable and fa.nullable != fb.nullable:
            return False
        if not arrays_identical(a.column(i), b.column(i), nan_payload):
            return False
    return True


def explain_diff(a: pa.RecordBatch, b: pa.RecordBatch) -> str:
    out = [f"rows {a.num_rows} vs {b.num_rows}; cols {a.num_columns} vs {b.num_columns}"]
    for i in range(min(a.num_columns, b.num_columns)):
        fa, fb = a.schema.field(i), b.schema.field(i)
        if fa != fb:
            out.append(f"field {i}: {fa} (nullable={fa.nullable}) vs {fb} (nullable={fb.nullable})")
        if a.num_rows == b.num_rows and not arrays_identical(a.column(i), b.column(i)):
            la, lb = a.column(i).to_pylist(), b.column(i).to_pylist()
            bad = [k for k in range(len(la)) if la[k] != lb[k] and not (la[k] != la[k] and lb[k] != lb[k])][:5]
            out.append(f"column {i} ({fa.name}) differs at rows {bad}: {[la[k] for k in bad]} vs {[lb[k] for k in bad]} nulls {a.column(i).null_count} vs {b.column(i).null_count}")
    return "\n".join(out)


def load_simple_sql() -> str:
    """the five statements of the reference's sample_queries/simple.sql (BASELINE config 1), kept as a data fixture"""
    return ";\n".join(load_golden("simple_sql.json")["statements"]) + ";"
