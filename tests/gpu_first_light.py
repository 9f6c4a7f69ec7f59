"""Ad-hoc first run on the GPU: library vs oracle on a handful of cases (superseded by tests/test_gpu_*.py)."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyarrow as pa
import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr, parse_select
from oracle import oracle as O
from tests.helpers import batches_identical, arrays_identical, explain_diff

rng = np.random.default_rng(1)
def mk(n):
    return pa.RecordBatch.from_arrays([
        pa.array(np.arange(n, dtype=np.int32)),
        pa.array([("s%d" % (i * 7919 % 1000)) * (1 + i % 3) for i in range(n)]),
        pa.array((rng.random(n) * 100).astype(np.float32)),
        pa.array(rng.integers(0, 2, n).astype(bool)),
        pa.array(rng.integers(-1000, 1000, n).astype(np.int64)),
        pa.array(rng.random(n), mask=rng.random(n) < 0.2),
    ], names=["id", "value1", "value2", "flag", "big", "dnull"])

ok = True
for n in [5, 100, 3000, 70000, 300000]:
    rb = mk(n)
    al = [[] for _ in range(rb.num_columns)]
    for sql in ["value2 > 10.0", "id % 2 = 0", "id < 25", "id > 25 + 0.0", "flag", "flag = true and value2 < 50.0",
                "value1 = 's7'", "value1 <> 's7' and id > 3", "big * 2 > id", "dnull > 0.5", "dnull > 0.5 or flag",
                "(id + big) * 2 > big / 3 and value2 / 3.0 < 20.0 or id % 7 = 1", "value2 > 1000.0", "value2 >= 0.0"]:
        e = parse_expr(sql)
        try:
            exp = O.filter_record(rb, al, e)
            got = chq.filter_record(rb, al, e)
            good = batches_identical(got, exp)
        except Exception as ex:
            traceback.print_exc(); good = False; got = exp = None
        print(f"n={n:7d} filter {sql!r:70s} rows={got.num_rows if got is not None else -1:7d} {'ok' if good else 'FAIL'}", flush=True)
        if not good:
            ok = False
            if got is not None: print(explain_diff(got, exp))
    sel = parse_select("select id, value1, id + 10.0 as id_plus_10, (value2 + 10) / 100 as value2, 1.0 / id as value3, 1.0 / (id * id) as value4, id * 3 as value5, big % 7, dnull * 2.0 as d2, flag or id > 5 as f2, * from t where id > 25 + 0.0")
    small = rb.slice(0, min(n, 40000))  # id*id must not overflow
    try:
        f_exp = O.filter_record(small, al, sel.selection); p_exp = O.project_record(sel.projection, f_exp, al)
        f_got = chq.filter_record(small, al, sel.selection); p_got = chq.project_record(sel.projection, f_got, al)
        good = batches_identical(p_got, p_exp)
        fp = chq.filter_project_record(sel.selection, sel.projection, small, al)
        good = good and batches_identical(fp, p_exp)
    except Exception:
        traceback.print_exc(); good = False; p_got = None
    print(f"n={n:7d} project simple.sql q4 {'ok' if good else 'FAIL'}", flush=True)
    if not good:
        ok = False
        if p_got is not None: print(explain_diff(p_got, p_exp))
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
