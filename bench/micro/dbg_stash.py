import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa
import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr
from oracle import oracle as O
for n in [5, 100, 999, 3000, 70000]:
    rng = np.random.default_rng(n)
    rb = pa.RecordBatch.from_arrays([pa.array(np.arange(n, dtype=np.int32)), pa.array((rng.random(n)*100).astype(np.float32)), pa.array((rng.random(n)*100).astype(np.float32))], names=["id","v1","value2"])
    al=[[],[],[]]
    for sql in ["value2 > 10.0", "id % 2 = 0", "v1 < 50.0 and value2 > 10.0"]:
        e=parse_expr(sql)
        got=chq.filter_record(rb, al, e); exp=O.filter_record(rb, al, e)
        bad=[c for c in range(3) if got.column(c).to_pylist()!=exp.column(c).to_pylist()]
        print(n, sql, got.num_rows, exp.num_rows, "bad cols", bad)
        if bad:
            c=bad[0]; g=got.column(c).to_pylist(); x=exp.column(c).to_pylist()
            idx=[i for i in range(min(len(g),len(x))) if g[i]!=x[i]][:5]
            print("   first diffs at", idx, [g[i] for i in idx], [x[i] for i in idx])
