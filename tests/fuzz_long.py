"""Long-running parity fuzz (not collected by pytest): random predicates / value expressions / projections / groups over
all column kinds and tile-boundary sizes, GPU (through the C ABI) against the CPU oracle -- values and status codes.

    python -m tests.fuzz_long [seconds] [seed]
"""
import sys
import time

import numpy as np
import pyarrow as pa

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr, parse_select
from oracle import oracle as O

from .cases import empty_aliases
from .helpers import arrays_identical, batches_identical, explain_diff
from .test_gpu_parity import (FAMILIES, FAMILY_WEIGHTS, make_batch, make_typed_batch, random_numeric, random_predicate,
                              random_typed_predicate, random_typed_value)

SIZES = [1, 2, 63, 64, 65, 511, 2047, 2048, 2049, 4100, 16383, 16384, 16385, 33000, 70001]


TYPE_POOL = [pa.int8(), pa.int16(), pa.int32(), pa.int64(), pa.uint8(), pa.uint16(), pa.uint32(), pa.uint64(), pa.float32(),
             pa.float64(), pa.bool_(), pa.utf8()]


def random_schema_batch(rng, n):
    """random column types / order / duplicate names / null masks + random table aliases"""
    k = int(rng.integers(1, 9))
    names, arrays, aliases = [], [], []
    for i in range(k):
        t = TYPE_POOL[int(rng.integers(0, len(TYPE_POOL)))]
        name = str(rng.choice(["a", "b", "c", "d", "val", "a"]))
        mask = (rng.random(n) < 0.15) if (n and rng.random() < 0.5) else None
        if t == pa.utf8():
            arr = pa.array(rng.choice(np.array(["", "x", "xy", "b", "long string " * 6]), n) if n else np.array([], dtype=object), type=t, mask=mask)
        elif t == pa.bool_():
            arr = pa.array(rng.integers(0, 2, n).astype(bool), mask=mask)
        elif pa.types.is_floating(t):
            vals = (rng.random(n) * 20 - 10)
            if n:
                vals[rng.integers(0, n, max(1, n // 50))] = rng.choice([0.0, -0.0, np.inf, -np.inf, np.nan])
            arr = pa.array(vals.astype(t.to_pandas_dtype()), mask=mask)
        else:
            hi = min(100, np.iinfo(t.to_pandas_dtype()).max)
            lo = -100 if np.iinfo(t.to_pandas_dtype()).min < 0 else 0
            arr = pa.array(rng.integers(lo, hi, n).astype(t.to_pandas_dtype()), mask=mask)
        names.append(name); arrays.append(arr)
        aliases.append([] if rng.random() < 0.3 else [str(rng.choice(["t", "u"]))])
    return pa.RecordBatch.from_arrays(arrays, names=names), aliases


def random_ident(rng, names):
    nm = str(rng.choice(names + ["nope"])) if rng.random() < 0.15 else str(rng.choice(names))
    r = rng.random()
    if r < 0.75:
        return nm
    if r < 0.97:
        return f"{rng.choice(['t', 'u', 't', 'u', 'w'])}.{nm}"
    return f"x.y.{nm}"


def random_expr2(rng, names, depth):
    if depth == 0 or rng.random() < 0.3:
        r = rng.random()
        if r < 0.7:
            return random_ident(rng, names)
        return str(rng.choice(["1", "2", "0", "2.5", "0.0", "3000000000", "'x'", "'b'", "true", "false"]))
    op = str(rng.choice(["+", "*", "/", "%", "<", "<=", "=", "<>", ">", ">=", "and", "or"]))
    e = f"{random_expr2(rng, names, depth - 1)} {op} {random_expr2(rng, names, depth - 1)}"
    return f"({e})" if rng.random() < 0.6 else e


_TYPE_TAG = {pa.int8(): "i8", pa.int16(): "i16", pa.int32(): "i32", pa.int64(): "i64", pa.uint8(): "u8", pa.uint16(): "u16",
             pa.uint32(): "u32", pa.uint64(): "u64", pa.float32(): "f32", pa.float64(): "f64"}


def typed_expr(rng, rec, want_bool):
    """a well-typed expression over a random-schema batch: the columns of ONE coercion family (so it evaluates instead of
    ending in UnsupportedTypeCoersion), bare names resolving to the first column of that name like the reference's
    column_by_name.  None when the schema has no usable column."""
    first = {}
    for f in rec.schema:
        first.setdefault(f.name, f.type)
    tags = {name: _TYPE_TAG.get(t) for name, t in first.items()}
    for _ in range(8):
        fam, lits = FAMILIES[int(rng.choice(len(FAMILIES) - 1, p=FAMILY_WEIGHTS[:-1] / FAMILY_WEIGHTS[:-1].sum()))]
        allowed = {"small": "i32"}
        names = [n for n, t in tags.items() if t is not None and (t in fam or allowed.get("small") == t and "small" in fam)]
        if not names:
            continue
        # (integer columns of random schemas hold values below 100: sums stay in range except for Int8 / UInt8 chains)
        ops = ("+", "+", "/", "%") if any(tags[n] in ("i8", "u8", "i16") for n in names) else ("+", "*", "/", "%", "+")
        num = lambda d: random_numeric(rng, d, names, ops=ops, lits=lits)   # noqa: E731
        if not want_bool:
            return num(int(rng.integers(1, 3)))
        leaf = lambda: f"{num(1)} {rng.choice(['<', '<=', '>', '>=', '=', '<>'])} {num(1)}"   # noqa: E731
        bools = [n for n, t in first.items() if t == pa.bool_()]
        strs = [n for n, t in first.items() if t == pa.utf8()]
        parts = []
        for _ in range(int(rng.integers(1, 4))):
            r = rng.random()
            if r < 0.15 and bools:
                parts.append(str(rng.choice(bools)))
            elif r < 0.3 and strs:
                parts.append(f"{rng.choice(strs)} {rng.choice(['<', '>=', '=', '<>'])} '{rng.choice(['x', 'b', 'xy', ''])}'")
            else:
                parts.append(leaf())
        e = parts[0]
        for q in parts[1:]:
            e = f"({e} {rng.choice(['and', 'or'])} {q})"
        return e
    return None


def dump_case(rec, exp, got):
    """small cases: the input and both results, bit patterns of the floats included (enough to replay the case by hand)"""
    if rec.num_rows > 40:
        return
    def show(x):
        try:
            cols = x.columns if hasattr(x, "columns") else [x]
            out = []
            for c in cols:
                if pa.types.is_floating(c.type):
                    w = {16: np.uint16, 32: np.uint32, 64: np.uint64}[c.type.bit_width]
                    out.append([None if v is None else hex(int(np.array([v], dtype=c.type.to_pandas_dtype()).view(w)[0])) for v in c.to_pylist()])
                else:
                    out.append(c.to_pylist())
            return out
        except Exception as e:   # noqa: BLE001
            return f"<{e}>"
    print(" input ", show(rec), "\n oracle", show(exp), "\n gpu   ", show(got), flush=True)


def outcome(fn):
    try:
        return None, fn()
    except (chq.ChqError, O.OracleError) as e:
        return e.code, None


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctxs = []
    for kind in (-1, 0, 1, 2):
        c = chq.Context(0)
        c.set_option("tile_kind", kind)
        if kind in (0, 2):   # two of the four contexts prove uniform-length string columns from 64 rows on (default: 2^24)
            c.set_option("uniform_utf8_rows", 64)
        ctxs.append(c)
    t0 = time.time()
    stats = {"filter": 0, "value": 0, "project": 0, "group": 0, "deep": 0, "group1": 0, "errors": 0, "unsupported": 0}
    it = 0
    while time.time() - t0 < budget:
        it += 1
        n = int(rng.choice(SIZES))
        rec = make_batch(n, int(rng.integers(0, 2**31)), nulls=bool(rng.random() < 0.7), tame=bool(rng.random() < 0.8))
        if rng.random() < 0.3 and n > 10:
            start = int(rng.integers(0, min(9, n - 1)))
            rec = rec.slice(start, n - start - int(rng.integers(0, 3)))
        # one case in seven runs on the temporal / decimal / Float16 / boolean-word columns (same-type comparisons, f16
        # arithmetic, Utf8 under AND / OR) through the same entry points
        typed_cols = rng.random() < 1.0 / 7.0
        if typed_cols:
            rec = make_typed_batch(n, int(rng.integers(0, 2**31)), nulls=bool(rng.random() < 0.7))
        gen_pred = (lambda d: random_typed_predicate(rng, min(d, 3))) if typed_cols else (lambda d: random_predicate(rng, d))
        gen_num = (lambda d: random_typed_value(rng)) if typed_cols else (lambda d: random_numeric(rng, d))
        al = empty_aliases(rec)
        ctx = ctxs[int(rng.integers(0, len(ctxs)))]
        mode = rng.random()
        if typed_cols and 0.9 <= mode < 0.95:
            mode = 0.97   # (the numeric-only group mode picks its columns by name)
        if rng.random() < 0.35:
            # random schemas (types, order, duplicate names, nulls, NaN / inf / signed zeros) and table aliases
            rec, al = random_schema_batch(rng, int(rng.choice([0, 1, 2, 5, 64, 65, 700, 2049, 20000])))
            names = rec.schema.names
            r = rng.random()
            typed = rng.random() < 0.85
            if r < 0.4:
                sql = (typed and typed_expr(rng, rec, False)) or random_expr2(rng, names, int(rng.integers(1, 4)))
                e = parse_expr(sql)
                ec, exp = outcome(lambda: O.compute_value(rec, al, e))
                gc, got = outcome(lambda: chq.compute_value(rec, al, e, ctx=ctx))
                kind = "schema-value"
                same = got is None or (got[1] == exp[1] and arrays_identical(got[0], exp[0], nan_payload=True))
            elif r < 0.7:
                sql = (typed and typed_expr(rng, rec, True)) or random_expr2(rng, names, int(rng.integers(1, 4)))
                e = parse_expr(sql)
                ec, exp = outcome(lambda: O.filter_record(rec, al, e))
                gc, got = outcome(lambda: chq.filter_record(rec, al, e, ctx=ctx))
                kind = "schema-filter"
                same = got is None or batches_identical(got, exp, nan_payload=True)
            else:
                items = []
                for _ in range(int(rng.integers(1, 5))):
                    q = rng.random()
                    if q < 0.15:
                        items.append("*")
                    elif q < 0.5:
                        items.append(random_ident(rng, names))
                    elif q < 0.8:
                        items.append((typed and typed_expr(rng, rec, bool(rng.random() < 0.3))) or random_expr2(rng, names, 2))
                    else:
                        items.append(((typed and typed_expr(rng, rec, bool(rng.random() < 0.3))) or random_expr2(rng, names, 2)) + f" as out{len(items)}")
                sql = "select " + ", ".join(items) + " from t"
                sel = parse_select(sql)
                ec, exp = outcome(lambda: O.project_record(sel.projection, rec, al))
                gc, got = outcome(lambda: chq.project_record(sel.projection, rec, al, ctx=ctx))
                kind = "schema-project"
                same = got is None or batches_identical(got, exp, nan_payload=True)
            stats[kind] = stats.get(kind, 0) + 1
            if ec == 30 and gc == 30:   # out of scope on BOTH sides; a one-sided 30 falls through to STATUS MISMATCH
                stats["unsupported"] += 1
                if stats["unsupported"] <= 12:
                    print(f"unsupported [{kind}] oracle {ec} gpu {gc}: {sql}   schema {[str(f.type) for f in rec.schema]}", flush=True)
                continue
            if ec is not None or gc is not None:
                stats["errors"] += 1
                if ec != gc:
                    print(f"STATUS MISMATCH [{kind}] n={rec.num_rows}: oracle {ec}, gpu {gc}: {sql}\n schema {rec.schema} aliases {al}", flush=True)
                    return 1
                continue
            if not same:
                print(f"VALUE MISMATCH [{kind}] n={rec.num_rows}: {sql}\n schema {rec.schema} aliases {al}", flush=True)
                dump_case(rec, exp, got)
                return 1
            continue
        if mode < 0.45:
            sql = gen_pred(int(rng.integers(1, 4)))
            e = parse_expr(sql)
            ec, exp = outcome(lambda: O.filter_record(rec, al, e))
            src = chq.DeviceRecordBatch.from_host(rec, ctx) if rng.random() < 0.5 else rec
            gc, got = outcome(lambda: chq.filter_record(src, al, e, ctx=ctx))
            if got is not None and hasattr(got, "to_host"):
                got = got.to_host()
            kind = "filter"
            same = got is None or batches_identical(got, exp)
        elif mode < 0.7:
            sql = gen_num(3) if rng.random() < 0.6 else gen_pred(2)
            e = parse_expr(sql)
            ec, exp = outcome(lambda: O.compute_value(rec, al, e)[0])
            gc, got = outcome(lambda: chq.compute_value(rec, al, e, ctx=ctx)[0])
            kind = "value"
            same = got is None or arrays_identical(got, exp, nan_payload=True)
        elif mode < 0.85:
            items = ", ".join([gen_num(2) + f" as c{k}" for k in range(int(rng.integers(1, 4)))] + (["*"] if rng.random() < 0.3 else []))
            sql = f"select {items} from t where {gen_pred(2)}"
            sel = parse_select(sql)
            ec, exp = outcome(lambda: O.project_record(sel.projection, O.filter_record(rec, al, sel.selection), al))
            if rng.random() < 0.5:
                ctx.set_option("fuse", 2)
            gc, got = outcome(lambda: chq.filter_project_record(sel.selection, sel.projection, rec, al, ctx=ctx))
            ctx.set_option("fuse", 1)
            kind = "project"
            same = got is None or batches_identical(got, exp, nan_payload=True)
        elif mode < 0.9:
            # deep expressions: beyond one device program -> sub-trees become temporary columns (fit_to_device)
            sql = gen_num(int(rng.integers(4, 7))) if rng.random() < 0.5 else gen_pred(int(rng.integers(3, 5)))
            e = parse_expr(sql)
            ec, exp = outcome(lambda: O.compute_value(rec, al, e)[0])
            src = chq.DeviceRecordBatch.from_host(rec, ctx) if rng.random() < 0.5 else rec
            gc, got = outcome(lambda: chq.compute_value(src, al, e, ctx=ctx)[0])
            kind = "deep"
            same = got is None or arrays_identical(got, exp, nan_payload=True)
        elif mode < 0.95:
            # numeric-only batches: the one-launch group path (wave-packed or tile table), per-batch and coalesced
            keep = [i for i, f in enumerate(rec.schema) if f.name in ("i8", "i16", "i32", "u8", "u16", "f32", "f64", "i64", "small")]
            num = pa.RecordBatch.from_arrays([pa.array(rec.column(i).to_numpy(zero_copy_only=False)) if rec.column(i).null_count == 0
                                              else pa.array(np.nan_to_num(rec.column(i).to_numpy(zero_copy_only=False)).astype(rec.schema.field(i).type.to_pandas_dtype()))
                                              for i in keep], names=[rec.schema.field(i).name for i in keep])
            if rng.random() < 0.45 and num.num_rows:   # + one or two non-null short-string columns: device groups then filter
                extra, names = [], []                   # them straight out of the batches (one launch, Utf8 inside the kernel)
                base = num.column(0).to_numpy(zero_copy_only=False)
                for u in range(int(rng.integers(1, 3))):
                    lens = rng.integers(0, 21, num.num_rows)
                    extra.append(pa.array([("s%d" % int(v))[:int(l)] + "x" * max(0, int(l) - 6) for v, l in zip(base, lens)], type=pa.utf8()))
                    names.append(f"str{u}")
                at = int(rng.integers(0, num.num_columns + 1))
                cols = list(num.columns); nm = list(num.schema.names)
                cols[at:at] = extra; nm[at:at] = names
                num = pa.RecordBatch.from_arrays(cols, names=nm)
            al = empty_aliases(num)
            sql = f"{random_numeric(rng, 2, ['i8', 'i16', 'i32', 'small'])} {rng.choice(['<', '>=', '='])} {random_numeric(rng, 1, ['i32', 'small', 'f32'])}"
            e = parse_expr(sql)
            k = int(rng.integers(2, 9))
            cut = sorted(set(int(x) for x in rng.integers(2, max(3, num.num_rows - 2), k))) if num.num_rows > 8 else []
            parts = [num.slice(a, b - a) for a, b in zip([0] + cut, cut + [num.num_rows])]
            ctx.set_option("group_mode", int(rng.integers(0, 3)))
            if rng.random() < 0.3:   # one fixed-width host batch in, host batch out: the single-synchronisation path
                ec, exp = outcome(lambda: O.filter_record(num, al, e))
                gc, got = outcome(lambda: chq.filter_record(num, al, e, ctx=ctx))
                same = got is None or batches_identical(got, exp)
                parts = None
            else:
                ec, exp = outcome(lambda: [O.filter_record(p, al, e) for p in parts])
            if parts is None:
                pass
            elif rng.random() < 0.33:
                gc, got = outcome(lambda: chq.filter_records(parts, al, e, ctx=ctx))
                same = got is None or all(batches_identical(g, x) for g, x in zip(got, exp))
            elif rng.random() < 0.5:
                devs = [chq.DeviceRecordBatch.from_host(p, ctx) for p in parts]
                gc, got = outcome(lambda: chq.filter_records(devs, al, e, ctx=ctx))
                same = got is None or all(batches_identical(g.to_host(), x) for g, x in zip(got, exp))
            else:
                devs = [chq.DeviceRecordBatch.from_host(p, ctx) for p in parts]
                gc, got = outcome(lambda: chq.filter_records_coalesced(devs, al, e, ctx=ctx))
                if got is not None:
                    whole = pa.Table.from_batches(exp).combine_chunks()
                    whole = whole.to_batches()[0] if whole.num_rows else exp[0].slice(0, 0)
                    same = got[1] == [x.num_rows for x in exp] and batches_identical(got[0].to_host(), whole, check_nullable=False)
                else:
                    same = True
            ctx.set_option("group_mode", 0)
            kind = "group1"
        else:
            sql = gen_pred(2)
            e = parse_expr(sql)
            k = int(rng.integers(2, 6))
            cut = sorted(set(int(x) for x in rng.integers(0, max(1, rec.num_rows), k)))
            parts = [rec.slice(a, b - a) for a, b in zip([0] + cut, cut + [rec.num_rows])]
            ec, exp = outcome(lambda: [O.filter_record(p, al, e) for p in parts])
            kind = "group"
            if rng.random() < 0.6:
                gc, got = outcome(lambda: chq.filter_records(parts, al, e, ctx=ctx))
                same = got is None or all(batches_identical(g, x) for g, x in zip(got, exp))
            else:   # joined output (general column kinds: concatenated on the host) from host or device batches
                src = [chq.DeviceRecordBatch.from_host(p, ctx) for p in parts] if rng.random() < 0.4 else parts
                gc, got = outcome(lambda: chq.filter_records_coalesced(src, al, e, ctx=ctx))
                same = True
                if got is not None:
                    whole = pa.Table.from_batches(exp).combine_chunks()
                    whole = whole.to_batches()[0] if whole.num_rows else exp[0].slice(0, 0)
                    joined = got[0].to_host() if hasattr(got[0], "to_host") else got[0]
                    same = got[1] == [x.num_rows for x in exp] and batches_identical(joined, whole, check_nullable=False)
        stats[kind] += 1
        if ec == 30 and gc == 30:   # out of scope on BOTH sides; a one-sided 30 falls through to STATUS MISMATCH
            stats["unsupported"] += 1
            continue
        if ec is not None or gc is not None:
            stats["errors"] += 1
            if ec != gc:
                print(f"STATUS MISMATCH [{kind}] n={rec.num_rows}: oracle {ec}, gpu {gc}: {sql}", flush=True)
                return 1
            continue
        if not same:
            print(f"VALUE MISMATCH [{kind}] n={rec.num_rows}: {sql}", flush=True)
            if kind != "group":
                dump_case(rec, exp, got)
            if kind in ("filter", "project"):
                print(explain_diff(got, exp))
            return 1
        if it % 200 == 0:
            print(f"{it} cases, {time.time() - t0:.0f} s: {stats}", flush=True)
    compared = it - stats["errors"] - stats["unsupported"]
    share = stats["errors"] / max(1, it)
    print(f"OK: {it} cases in {time.time() - t0:.0f} s: {compared} compared VALUES, {stats['errors']} ended in the same error status on both "
          f"sides ({100 * share:.1f} %), {stats['unsupported']} out of scope on both sides: {stats}", flush=True)
    if share > 0.35:   # a fuzz that mostly compares error codes proves little: the generators are biased towards evaluable cases
        print(f"FAILED: {100 * share:.1f} % of the cases ended in an error (bound: 35 %)", flush=True)
        return 2
    return 0


if __name__ == "__main__":
    sys.exit(main())
