#!/usr/bin/env python3
"""Expected output digests of examples/c_abi_demo.c, produced by the ORACLE (oracle/chq_oracle.c) on the data the demo
generates (same LCG), for the parameter sets tests/test_c_example.py runs.  Run in the build container (CPU only):

    python scripts/make_c_abi_demo_fixture.py > tests/golden/c_abi_demo_expected.json
"""
import json
import os
import sys

import numpy as np
import pyarrow as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chapterhouseqe_amd.sqlparse import parse_expr, parse_select  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [(10_000, 64), (3, 5), (70_001, 7)]
W = (1, 3, 7)
M64 = (1 << 64) - 1


def demo_data(total):
    """the demo's generator: x <- 1664525 x + 1013904223 (mod 2^32), value = (x >> 8) * (100 / 2^24) as float32"""
    n = 2 * total
    xs = np.empty(n, dtype=np.uint32)
    x = 0xC0FFEE
    for i in range(n):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        xs[i] = x
    vals = (xs >> np.uint32(8)).astype(np.float32) * np.float32(100.0 / 16777216.0)
    return np.arange(total, dtype=np.int32), vals[0::2].copy(), vals[1::2].copy()


def digest(batch, ncols):
    d = 0
    for c in range(ncols):
        v = batch.column(c).to_numpy(zero_copy_only=False)
        bits = v.view(np.uint32).astype(np.uint64) if v.dtype != np.uint32 else v.astype(np.uint64)
        k = np.arange(1, len(bits) + 1, dtype=np.uint64)
        d = (d + W[c] * int((bits * k).sum(dtype=np.uint64))) & M64
    return d


def main():
    pred = parse_expr("value2 > 10.0")
    sel = parse_select("select id, value2 * 2.0 as twice from t where value2 > 10.0")
    out = {}
    for rows, nb in CASES:
        total = rows * nb
        idc, v1, v2 = demo_data(total)
        mk = lambda a, b: pa.RecordBatch.from_arrays([pa.array(idc[a:b]), pa.array(v1[a:b]), pa.array(v2[a:b])], names=["id", "value1", "value2"])
        al = [[] for _ in range(3)]
        per = [O.filter_record(mk(b * rows, (b + 1) * rows), al, pred) for b in range(nb)]
        d_loop = sum((b + 1) * digest(o, 3) for b, o in enumerate(per)) & M64
        n_loop = sum(o.num_rows for o in per)
        whole = O.filter_record(mk(0, total), al, pred)
        counts = sum((b + 1) * o.num_rows for b, o in enumerate(per)) & M64
        first = mk(0, rows)
        proj = O.project_record(sel.projection, O.filter_record(first, al, sel.selection), al)
        out[f"{rows}x{nb}"] = [
            f"digest filter_record rows={n_loop} sum={d_loop:016x}",
            f"digest filter_records rows={n_loop} sum={d_loop:016x}",
            f"digest filter_records_coalesced rows={whole.num_rows} sum={digest(whole, 3):016x} counts={counts:016x}",
            f"digest filter_project_record rows={proj.num_rows} sum={digest(proj, 2):016x}",
        ]
    json.dump({"generator": "scripts/make_c_abi_demo_fixture.py (oracle/chq_oracle.c on the demo's LCG data)", "cases": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
