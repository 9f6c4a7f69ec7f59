"""python -m tests.stress_case [iterations]: two small host batches filtered over and over through four contexts (one per tile
kind), interleaved with larger calls that churn the buffer pools -- hunts timing-dependent faults the seeded fuzz only meets
by chance.  Prints the first mismatch with everything needed to replay it."""
import sys

import numpy as np
import pyarrow as pa

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr
from oracle import oracle as O

from .helpers import batches_identical
from .test_gpu_parity import make_batch


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    ctxs = []
    for kind in (-1, 0, 1, 2):
        c = chq.Context(0)
        c.set_option("tile_kind", kind)
        ctxs.append((kind, c))
    nan = np.frombuffer(np.uint64(0x7ff8000000000000).tobytes(), dtype=np.float64)[0]
    cases = [
        (pa.RecordBatch.from_arrays([pa.array([99], pa.int32()), pa.array([nan], pa.float64()), pa.array([22], pa.uint64())], names=["b", "a", "c"]),
         [["t"], ["t"], ["u"]], "(8 * 8 < 7 + b or 5 % 2 <> 5)"),
        (pa.RecordBatch.from_arrays([pa.array([1, 2, 3, 4, 5], pa.int8()), pa.array([1.5, 2.5, nan, 4.5, 5.5], pa.float64()),
                                     pa.array([1, 2, 3, 4, 5], pa.int32()), pa.array([7, 8, 9, 10, 11], pa.uint16()),
                                     pa.array(np.array([0.5, 1.5, 2.5, 3.5, 4.5], dtype=np.float32)), pa.array(["x", "yy", "", "zzz", "w"])],
                                    names=["val", "a", "b", "d", "c", "s"]),
         [[], [], [], [], [], []], "((c % 1 < (c / 8) or c <= c) or c + a >= (a + c))"),
    ]
    want = [O.filter_record(r, al, parse_expr(sql)) for r, al, sql in cases]
    big = make_batch(70_000, 5, tame=True)
    big_al = [[] for _ in big.schema]
    rng = np.random.default_rng(3)
    for it in range(iters):
        kind, ctx = ctxs[int(rng.integers(0, 4))]
        k = int(rng.integers(0, len(cases)))
        rec, al, sql = cases[k]
        got = chq.filter_record(rec, al, parse_expr(sql), ctx=ctx)
        if not batches_identical(got, want[k], nan_payload=True):
            print(f"MISMATCH at iteration {it}, tile_kind {kind}, case {k}: {sql}\n got  {got.to_pydict()}\n want {want[k].to_pydict()}", flush=True)
            return 1
        if rng.random() < 0.3:   # churn the pools
            kind2, ctx2 = ctxs[int(rng.integers(0, 4))]
            chq.filter_record(big, big_al, parse_expr("i32 % 2 = 0 and f32 > 1.0"), ctx=ctx2)
        if it % 2000 == 0:
            print(f"{it} iterations", flush=True)
    print(f"OK: {iters} iterations")
    return 0


if __name__ == "__main__":
    sys.exit(main())
