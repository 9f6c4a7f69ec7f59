#!/usr/bin/env python3
"""Secondary measurements: the other BASELINE.json configs (SURVEY.md section 8 d) on ONE GPU.  bench.py stays the
headline (config 2); this script reports kernel-level numbers for configs 2b/3/4/5-shape so DESIGN.md can quote
them.  Every case is checked against the CPU oracle on a prefix before it is timed.

    python bench_configs.py [--scale 1.0] [--out profiles/r1/configs.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="scale every row count (1.0 = BASELINE sizes)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--where", default=None, help="replace the predicate of the selected case(s) (instruction-cost experiments)")
    ap.add_argument("--no-select", action="store_true", help="skip the projection / one-pass parts of the selected case(s)")
    args = ap.parse_args()

    import numpy as np
    import pyarrow as pa
    import torch

    import chapterhouseqe_amd as chq
    from chapterhouseqe_amd.sqlparse import parse_expr, parse_select
    from oracle import oracle as O
    from tests.helpers import batches_identical, explain_diff

    dev = torch.device("cuda", 0)
    ctx = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_option("time_kernels", 1)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    results = []

    def gen(n, spec, seed):
        """spec: list of (name, kind, *params); returns (tensors keepalive, column tuples, host builder)"""
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        keep, cols = [], []
        for item in spec:
            name, kind = item[0], item[1]
            if kind == "f32":
                t = torch.empty(n, dtype=torch.float32, device=dev).uniform_(item[2], item[3], generator=g)
                keep.append(t); cols.append((name, "f", t.data_ptr()))
            elif kind == "i32":
                t = torch.randint(item[2], item[3], (n,), dtype=torch.int32, device=dev, generator=g)
                keep.append(t); cols.append((name, "i", t.data_ptr()))
            elif kind == "id":
                t = torch.arange(n, dtype=torch.int32, device=dev)
                keep.append(t); cols.append((name, "i", t.data_ptr()))
            elif kind == "utf8":
                L = item[2]
                assert n * L < 2**31, "Utf8 (int32 offsets) holds < 2 GiB of bytes per array: lower the row count"
                chars = torch.randint(ord("a"), ord("z") + 1, (n * L,), dtype=torch.uint8, device=dev, generator=g)
                offs = (torch.arange(n + 1, dtype=torch.int64, device=dev) * L).to(torch.int32)
                keep += [chars, offs]; cols.append((name, "u", offs.data_ptr(), chars.data_ptr()))
        torch.cuda.synchronize()
        return keep, cols

    def host_prefix(keep, spec, m):
        arrays, fields = [], []
        k = 0
        for item in spec:
            name, kind = item[0], item[1]
            if kind == "utf8":
                L = item[2]
                chars, offs = keep[k], keep[k + 1]; k += 2
                arr = pa.Array.from_buffers(pa.utf8(), m, [None, pa.py_buffer(offs[: m + 1].cpu().numpy().tobytes()),
                                                          pa.py_buffer(chars[: m * L].cpu().numpy().tobytes())])
                arrays.append(arr); fields.append(pa.field(name, pa.utf8(), False))
            else:
                t = keep[k]; k += 1
                arrays.append(pa.array(t[:m].cpu().numpy())); fields.append(pa.field(name, arrays[-1].type, False))
        return pa.RecordBatch.from_arrays(arrays, schema=pa.schema(fields))

    def run_case(name, n, spec, where, select=None, seed=1, note=""):
        if args.only and args.only not in name:
            return
        n = max(1024, int(n * args.scale))
        if args.where:
            where = args.where
        if args.no_select:
            select = None
        keep, cols = gen(n, spec, seed)
        rec = chq.DeviceRecordBatch.from_device_pointers(cols, n, ctx=ctx, keepalive=keep)
        al = [[] for _ in cols]
        pred = parse_expr(where)
        fields = parse_select(f"select {select} from t").projection if select else None
        # parity on a prefix
        m = min(n, 2_000_000)
        host = host_prefix(keep, spec, m)
        sub = chq.DeviceRecordBatch.from_device_pointers(cols, m, ctx=ctx)
        got = chq.filter_record(sub, al, pred, ctx=ctx)
        exp = O.filter_record(host, al, pred)
        ok = batches_identical(got.to_host(), exp)
        if fields:
            gp = chq.project_record(fields, got, al, ctx=ctx).to_host()
            ok = ok and batches_identical(gp, O.project_record(fields, exp, al), nan_payload=False)
        if fields:   # the single-pass filter -> project kernel against the same two oracle steps
            ctx.set_option("fuse", 2)   # measure it even where the default heuristic would take the two steps
            one = chq.filter_project_record(pred, fields, sub, al, ctx=ctx)
            ok = ok and ctx.last_stats()["launches"] == 1 and \
                batches_identical(one.to_host(), O.project_record(fields, exp, al), nan_payload=False)
            one.release()
        got.release()
        if not ok:
            raise SystemExit(f"{name}: GPU result differs from the oracle")
        # timing
        fk, pk, wall = [], [], []
        for it in range(args.steps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = chq.filter_record(rec, al, pred, ctx=ctx)
            st = ctx.last_stats()
            t1 = time.perf_counter()
            pw = None
            if fields:
                p_out = chq.project_record(fields, out, al, ctx=ctx)
                torch.cuda.synchronize()
                pw = time.perf_counter() - t1
                p_out.release()
            rows_out = out.num_rows
            out.release()
            if it:
                fk.append(st["kernel_ns"] / 1e6); wall.append((t1 - t0) * 1e3)
                if pw is not None:
                    pk.append(pw * 1e3)
        fk.sort(); wall.sort(); pk.sort()
        one_pass = None
        if fields:
            ok_ms, ow_ms = [], []
            for it in range(args.steps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o = chq.filter_project_record(pred, fields, rec, al, ctx=ctx)
                t1 = time.perf_counter()
                st1 = ctx.last_stats()
                o.release()
                if it:
                    ok_ms.append(st1["kernel_ns"] / 1e6); ow_ms.append((t1 - t0) * 1e3)
            ok_ms.sort(); ow_ms.sort()
            ctx.set_option("fuse", 1)
            ob = st1["bytes_read_alg"] + st1["bytes_written_alg"]
            one_pass = {"kernel_ms": ok_ms[len(ok_ms) // 2], "wall_ms": ow_ms[len(ow_ms) // 2], "alg_bytes": ob, "launches": st1["launches"],
                        "GBps": ob / (ok_ms[len(ok_ms) // 2] * 1e-3) / 1e9, "frac_of_8TBps": ob / (ok_ms[len(ok_ms) // 2] * 1e-3) / 1e9 / HBM_PEAK}
        # the library's own count: every fixed-width input byte once, the selected output bytes once;
        # Utf8: 4 B of offsets per row read by two passes, the SELECTED rows' bytes read and written, 4 B of new offsets
        alg = st["bytes_read_alg"] + st["bytes_written_alg"]
        has_utf8 = any(item[1] == "utf8" for item in spec)
        one_kernel = st["launches"] <= 2          # main kernel (+ its tail-tile launch): the kernel time covers the whole call's bytes
        r = {"case": name, "rows": n, "rows_out": rows_out, "selectivity": rows_out / n, "where": where, "select": select,
             "filter_kernel_ms": fk[len(fk) // 2], "filter_wall_ms": wall[len(wall) // 2], "launches": st["launches"],
             "alg_bytes": alg,
             "fused_kernel_GBps": alg / (fk[len(fk) // 2] * 1e-3) / 1e9 if one_kernel or not has_utf8 else None,
             "whole_filter_GBps": alg / (wall[len(wall) // 2] * 1e-3) / 1e9,
             "rows_per_s_wall": n / (wall[len(wall) // 2] * 1e-3), "validated_rows": m, "note": note}
        r["fused_kernel_frac_of_8TBps"] = r["fused_kernel_GBps"] / HBM_PEAK if r["fused_kernel_GBps"] else None
        r["whole_filter_frac_of_8TBps"] = r["whole_filter_GBps"] / HBM_PEAK
        if pk:
            r["project_wall_ms"] = pk[len(pk) // 2]
        if one_pass:
            r["filter_project_one_pass"] = one_pass
        results.append(r)
        print(json.dumps(r), flush=True)
        del rec, keep
        ctx.set_option("trim_pool", 1)
        torch.cuda.empty_cache()

    def run_group_case(name, n, rows_per_batch, spec, where, seed=1, note=""):
        """the same rows cut into reference-sized batches (physical_planner.rs:323) and filtered by ONE
        chq_filter_records call; compared with calling chq_filter_record per batch on a subset"""
        if args.only and args.only not in name:
            return
        n = max(4 * rows_per_batch, int(n * args.scale)) // rows_per_batch * rows_per_batch
        nb = n // rows_per_batch
        keep, cols = gen(n, spec, seed)
        al = [[] for _ in cols]
        pred = parse_expr(where)
        widths = [4 for _ in cols]
        t0 = time.perf_counter()
        devs = [chq.DeviceRecordBatch.from_device_pointers(
            [(c[0], c[1], c[2] + w * b * rows_per_batch) for c, w in zip(cols, widths)], rows_per_batch, ctx=ctx) for b in range(nb)]
        grp = chq.RecordGroup(devs, ctx)
        wrap_s = time.perf_counter() - t0
        # parity: first batches against the oracle, every batch's row count against torch
        m = min(nb, 8)
        got = chq.filter_records(devs[:m], al, pred, ctx=ctx)
        host = host_prefix(keep, spec, m * rows_per_batch)
        for b in range(m):
            exp = O.filter_record(host.slice(b * rows_per_batch, rows_per_batch), al, pred)
            if not batches_identical(got[b].to_host(), exp):
                raise SystemExit(f"{name}: batch {b} differs from the oracle")
        del got
        counts = chq.filter_records(grp, al, pred, ctx=ctx, wrap=False)
        rows_out = sum(counts)
        fk, wall = [], []
        for it in range(args.steps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            counts = chq.filter_records(grp, al, pred, ctx=ctx, wrap=False)
            t1 = time.perf_counter()
            st = ctx.last_stats()
            if it:
                fk.append(st["kernel_ns"] / 1e6); wall.append((t1 - t0) * 1e3)
        fk.sort(); wall.sort()
        # the C call alone (no Python per-batch work): arguments prepared, outputs released afterwards through C callbacks
        import ctypes as C
        from chapterhouseqe_amd import _lib as L
        from chapterhouseqe_amd.record_utils import _Aliases, _expr_to_c
        ce, cal = _expr_to_c(pred), _Aliases(al)
        ccall, crel = [], []
        for it in range(3):
            outs = (L.ArrowDeviceArray * nb)(); schemas = (L.ArrowSchema * nb)()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = L.lib().chq_filter_records(ctx.handle, nb, grp.ptrs, C.byref(grp.schema), cal.ptr, ce, L.ARROW_DEVICE_ROCM, outs, schemas)
            t1 = time.perf_counter()
            assert rc == 0
            rel = C.CFUNCTYPE(None, C.c_void_p)
            for i in range(nb):
                rel(outs[i].array.release)(C.addressof(outs[i].array)); rel(schemas[i].release)(C.addressof(schemas[i]))
            t2 = time.perf_counter()
            ccall.append((t1 - t0) * 1e3); crel.append((t2 - t1) * 1e3)
        L.lib().chq_expr_free(ce)
        # the coalesced form: ONE output batch (the kernel's dense buffers), nothing exported per input batch
        coal = []
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            big, rows_per = chq.filter_records_coalesced(grp, al, pred, ctx=ctx)
            t1 = time.perf_counter()
            assert big.num_rows == rows_out and rows_per == counts
            big.release()
            if it:
                coal.append((t1 - t0) * 1e3)
        # the per-batch loop on a subset, same batches
        sub = min(nb, 2000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in range(sub):
            chq.filter_record(devs[b], al, pred, ctx=ctx).release()
        loop_s = time.perf_counter() - t0
        alg = st["bytes_read_alg"] + st["bytes_written_alg"]
        r = {"case": name, "rows": n, "batches": nb, "rows_per_batch": rows_per_batch, "rows_out": rows_out, "selectivity": rows_out / n,
             "where": where, "group_kernel_ms": fk[len(fk) // 2], "group_call_wall_ms": wall[len(wall) // 2], "launches": st["launches"],
             "tiles": st["tiles"], "alg_bytes": alg, "group_kernel_GBps": alg / (fk[len(fk) // 2] * 1e-3) / 1e9,
             "group_kernel_frac_of_8TBps": alg / (fk[len(fk) // 2] * 1e-3) / 1e9 / HBM_PEAK,
             "rows_per_s_group_call": n / (wall[len(wall) // 2] * 1e-3),
             "c_call_ms": min(ccall), "rows_per_s_c_call": n / (min(ccall) * 1e-3), "python_release_ms": min(crel),
             "coalesced_call_ms": min(coal), "rows_per_s_coalesced_call": n / (min(coal) * 1e-3),
             "per_batch_loop_us_per_batch": loop_s / sub * 1e6, "rows_per_s_per_batch_loop": sub * rows_per_batch / loop_s,
             "wrap_inputs_s": wrap_s, "note": note}
        results.append(r)
        print(json.dumps(r), flush=True)
        grp.release()
        del devs, keep
        ctx.set_option("trim_pool", 1)
        torch.cuda.empty_cache()

    def run_ref_group_case(name, nb, rows_per_batch, where, seed=5, note="", null_share=0.0):
        """the reference's own schema (id:Int32, value1:Utf8(8), value2:Float32) in its own batch size, resident in HBM:
        ONE group call (device-side join + the single-batch kernels) against the per-batch loop and the one-batch time"""
        if args.only and args.only not in name:
            return
        nb = max(8, int(nb * args.scale))
        n = nb * rows_per_batch
        L8 = 8
        g = torch.Generator(device=dev); g.manual_seed(seed)
        ids = torch.arange(n, dtype=torch.int32, device=dev)
        chars = torch.randint(ord("a"), ord("z") + 1, (n * L8,), dtype=torch.uint8, device=dev, generator=g)
        offs = (torch.arange(rows_per_batch + 1, dtype=torch.int64, device=dev) * L8).to(torch.int32)   # batch-local offsets
        v2 = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
        torch.cuda.synchronize()
        al = [[], [], []]
        pred = parse_expr(where)
        valid = None
        if null_share > 0:   # value2 is an optional column with real nulls (what read_files produces from `optional` Parquet columns)
            assert rows_per_batch % 8 == 0
            keepv = torch.rand(n, device=dev, generator=g) >= null_share
            valid = (keepv.view(-1, 8).to(torch.uint8) << torch.arange(8, device=dev, dtype=torch.uint8)).sum(dim=1).to(torch.uint8)   # LSB-first bitmap
            torch.cuda.synchronize()
            devs = [chq.DeviceRecordBatch.from_device_buffers(
                [{"name": "id", "format": "i", "values": ids.data_ptr() + 4 * b * rows_per_batch},
                 {"name": "value1", "format": "u", "values": offs.data_ptr(), "data": chars.data_ptr() + L8 * b * rows_per_batch},
                 {"name": "value2", "format": "f", "nullable": True, "null_count": -1, "values": v2.data_ptr() + 4 * b * rows_per_batch,
                  "validity": valid.data_ptr() + b * rows_per_batch // 8}], rows_per_batch, ctx=ctx) for b in range(nb)]
        else:
            devs = [chq.DeviceRecordBatch.from_device_pointers(
                [("id", "i", ids.data_ptr() + 4 * b * rows_per_batch), ("value1", "u", offs.data_ptr(), chars.data_ptr() + L8 * b * rows_per_batch),
                 ("value2", "f", v2.data_ptr() + 4 * b * rows_per_batch)], rows_per_batch, ctx=ctx) for b in range(nb)]
        grp = chq.RecordGroup(devs, ctx)
        # parity: a few batches against the oracle
        got = chq.filter_records(devs[:4], al, pred, ctx=ctx)
        for b in range(4):
            lo = b * rows_per_batch
            host = pa.RecordBatch.from_arrays([
                pa.array(ids[lo:lo + rows_per_batch].cpu().numpy()),
                pa.Array.from_buffers(pa.utf8(), rows_per_batch, [None, pa.py_buffer(offs.cpu().numpy().tobytes()),
                                                                  pa.py_buffer(chars[L8 * lo:L8 * (lo + rows_per_batch)].cpu().numpy().tobytes())]),
                pa.array(v2[lo:lo + rows_per_batch].cpu().numpy(), mask=None if valid is None else ~keepv[lo:lo + rows_per_batch].cpu().numpy())],
                names=["id", "value1", "value2"])
            if not batches_identical(got[b].to_host(), O.filter_record(host, al, pred), check_nullable=False):
                raise SystemExit(f"{name}: batch {b} differs from the oracle")
        del got
        per_call, coal = [], []
        counts = None
        for it in range(args.steps + 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            counts = chq.filter_records(grp, al, pred, ctx=ctx, wrap=False)
            t1 = time.perf_counter()
            if it: per_call.append((t1 - t0) * 1e3)
        # the C call alone (what a Rust / C caller pays): outputs released afterwards, outside the timed region
        import ctypes as C
        from chapterhouseqe_amd import _lib as L
        from chapterhouseqe_amd.record_utils import _Aliases, _expr_to_c
        ce, cal = _expr_to_c(pred), _Aliases(al)
        ccall = []
        for it in range(3):
            outs = (L.ArrowDeviceArray * nb)(); schemas = (L.ArrowSchema * nb)()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = L.lib().chq_filter_records(ctx.handle, nb, grp.ptrs, C.byref(grp.schema), cal.ptr, ce, L.ARROW_DEVICE_ROCM, outs, schemas)
            t1 = time.perf_counter()
            assert rc == 0
            rel = C.CFUNCTYPE(None, C.c_void_p)
            for i in range(nb):
                rel(outs[i].array.release)(C.addressof(outs[i].array)); rel(schemas[i].release)(C.addressof(schemas[i]))
            ccall.append((t1 - t0) * 1e3)
        L.lib().chq_expr_free(ce)
        coalesced_ms = None
        if n * L8 <= (1 << 30):
            for it in range(args.steps + 1):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                big, rows_per = chq.filter_records_coalesced(grp, al, pred, ctx=ctx)
                t1 = time.perf_counter()
                assert rows_per == counts
                big.release()
                if it: coal.append((t1 - t0) * 1e3)
            coalesced_ms = sorted(coal)[len(coal) // 2]
        sub = min(nb, 500)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in range(sub):
            chq.filter_record(devs[b], al, pred, ctx=ctx).release()
        loop_us = (time.perf_counter() - t0) / sub * 1e6
        rows_out = sum(counts)
        alg = n * (4 + 4 + L8 + 4) + rows_out * (4 + 4 + L8 + 4)
        per_call.sort()
        r = {"case": name, "rows": n, "batches": nb, "rows_per_batch": rows_per_batch, "where": where, "rows_out": rows_out,
             "group_call_ms": per_call[len(per_call) // 2], "c_call_ms": min(ccall), "us_per_batch_c_call": min(ccall) * 1e3 / nb,
             "coalesced_call_ms": coalesced_ms, "per_batch_loop_us_per_batch": loop_us,
             "us_per_batch_group_call": per_call[len(per_call) // 2] * 1e3 / nb, "alg_bytes": alg,
             "group_call_GBps": alg / (per_call[len(per_call) // 2] * 1e-3) / 1e9,
             "c_call_GBps": alg / (min(ccall) * 1e-3) / 1e9,
             "fields": "c_call_ms = chq_filter_records through ctypes (one Arrow struct per output batch: the product); group_call_ms = the "
                       "Python mirror chq.filter_records, which also builds one Python object per output batch (~2.6 us each)",
             "note": note}
        results.append(r)
        print(json.dumps(r), flush=True)
        grp.release()
        del devs, ids, chars, v2
        ctx.set_option("trim_pool", 1)
        torch.cuda.empty_cache()

    f3 = [("value0", "f32", 0, 100), ("value1", "f32", 0, 100), ("value2", "f32", 0, 100)]
    run_group_case("group2 value2>10, 10k-row batches", 1_000_000_000, 10_000, f3, "value2 > 10.0", seed=0xC0FFEE,
                   note="call wall time includes building the tile table, exporting and releasing every output batch through ctypes")
    run_group_case("group2 value2>10, 100k-row batches", 1_000_000_000, 100_000, f3, "value2 > 10.0", seed=0xC0FFEE)
    run_case("config2 value2>10 (s~0.9)", 1_000_000_000, f3, "value2 > 10.0", seed=0xC0FFEE)
    run_case("config2b value2>90 (s~0.1)", 1_000_000_000, f3, "value2 > 90.0", seed=0xC0FFEE)
    run_case("config2c value2>50 (s~0.5)", 1_000_000_000, f3, "value2 > 50.0", seed=0xC0FFEE)
    c3 = [("a", "i32", 0, 1000), ("b", "f32", 0, 100), ("c", "f32", 0, 1100), ("d", "i32", 0, 10), ("e", "f32", 0, 2)]
    run_case("config3 compound + projection", 1_000_000_000, c3, "a + b > c and d < 5.0 or e > 1.0",
             select="a, a + b as ab, d * 2 as d2, e / 3.0 as e3", seed=3)
    c8 = [(f"c{i}", "f32", 0, 100) for i in range(8)]
    run_case("config3b narrow projection of an 8-column table", 500_000_000, c8, "c3 > 10.0", select="c0 + c1 as s, c2", seed=8,
             note="the reference's filter copies all 8 columns before materialize reads 3 of them; the one-pass kernel never does")
    c4 = [("id", "id"), ("value1", "utf8", 100), ("value2", "f32", 0, 100)]
    run_case("config4 wide strings id>25 (s~1)", 20_000_000, c4, "id > 25", seed=4)
    n4 = max(1024, int(20_000_000 * args.scale))
    run_case("config4b wide strings id>n/2 (s~0.5)", 20_000_000, c4, f"id > {n4 // 2}", seed=4)
    run_case("config4c wide strings value2<10 (s~0.1)", 20_000_000, c4, "value2 < 10.0", seed=4)
    run_case("config4d wide strings value1>='n' (Utf8 predicate, s~0.5)", 20_000_000, c4, "value1 >= 'n'", seed=4)
    run_case("config4e wide strings value1<'b' (Utf8 predicate, s~0.04)", 20_000_000, c4, "value1 < 'b'", seed=4)
    c5 = [("id", "id"), ("value1", "utf8", 8), ("value2", "f32", 0, 100)]
    run_case("config5-shape id%2=0, one record batch", 250_000_000, c5, "id % 2 = 0", seed=5,
             note="huge_simple.sql shape; one Utf8 array holds < 2 GiB of bytes (int32 offsets), so a 1.25 B-row GPU shard is "
                  "five such batches")
    run_ref_group_case("refgroup id%2=0, 12 500 x 10k-row batches (1 GB of strings)", 12_500, 10_000, "id % 2 = 0",
                       note="reference schema id:Int32, value1:Utf8(8), value2:Float32; per-batch outputs and the joined output")
    run_ref_group_case("refgroup with nulls id%2=0, 12 500 x 10k-row batches, value2 optional with 5 % nulls", 12_500, 10_000, "id % 2 = 0",
                       note="validity bitmaps per batch: the one-launch path + one bitmap compaction (round 3)", null_share=0.05)
    run_ref_group_case("refgroup id%2=0, 100 000 x 10k-row batches", 100_000, 10_000, "id % 2 = 0",
                       note="1 B rows = 8 GB of string bytes: eight joined chunks (int32 offsets), per-batch outputs only")
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
