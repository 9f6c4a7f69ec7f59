#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X filter path (BASELINE.json / SURVEY.md section 8 d, config 2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One *step* = one `filter_record` call (`SELECT * WHERE value2 > 10.0`) over ONE device-resident record batch of
`--rows` rows (default 1e9) with three non-null Float32 columns U[0,100), through the C ABI (libchq.so): predicate
evaluation + order-preserving compaction of all three columns into freshly allocated HBM buffers, including the
read-back of the output row count (the Arrow length the caller needs).  Inputs are in HBM before the timed region.

N > 1: one operator instance per GPU (one process per GPU), each filtering its own `--rows`-row shard of an
N x rows table -- batches are independent in the reference (filter_task.rs:86-125), so there is no data-path
collective; ranks only all-reduce the elapsed time (MAX) and row counts (SUM) after the timed region ("weak").
Started without WORLD_SIZE in the environment, `--gpus N` launches the N ranks itself (child processes through
torch.distributed.run on 127.0.0.1, before this process has imported torch or touched a GPU) and exits with their code.

Prints ONE JSON line on rank 0 with the driver's fields plus `roofline` (dominant kernel vs HBM peak) and
`cpu_baseline` (the C oracle = CPU restatement of the reference path, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
PREDICATE = "value2 > 10.0"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="rows per GPU")
    ap.add_argument("--predicate", default=PREDICATE)
    ap.add_argument("--cpu-rows", type=int, default=1_000_000_000, help="rows of the same data timed on the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="context option key=value (experiments)")
    ap.add_argument("--validate-rows", type=int, default=4_000_000, help="prefix checked bit-exact against the oracle")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra config-3 / reference-schema lines (N = 1 only)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU: every rank joins a gloo group, runs the barrier / MAX / SUM protocol of the timed region on "
                         "dummy numbers and rank 0 prints the JSON line (tests the N-rank launch path on CPU)")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help="with --launcher-selftest: this rank exits non-zero")
    return ap.parse_args()


def self_launch(n_ranks, argv):
    """`--gpus N` without a launcher: start the N ranks as children (one process per GPU, rendezvous on 127.0.0.1) and
    return their exit code.  Runs before this process imports torch or touches a GPU -- never re-exec a process that has."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def launcher_selftest(args):
    """The multi-rank protocol of main() on CPU (gloo): barrier, timed region, barrier, MAX over ranks, SUM of counts,
    ONE JSON line from rank 0, non-zero exit when a rank fails."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    if rank == args.selftest_fail_rank:
        raise SystemExit(3)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rows = 1000 * (rank + 1)
    per_rank = [rows / elapsed]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rows], dtype=torch.int64)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rows = int(rr.item())
        gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor(per_rank, dtype=torch.float64))
        per_rank = [float(g.item()) for g in gathered]
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "value": rows / elapsed, "per_gpu_rows_per_s": per_rank,
                          "rows_total": rows}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def extra_lines(chq, torch, dev, ctx, n):
    """Secondary workloads carried by the same driver run (N = 1): BASELINE config 3 (compound predicate over five mixed
    Int32 / Float32 columns + the arithmetic projection of the survivors) and the reference's own schema in its own batch
    size (id:Int32, value1:Utf8(8), value2:Float32; 10 000-row batches resident in HBM, one group call), and the rows either
    side of the path: Parquet scan -> filter + project -> Parquet write on the device."""
    from chapterhouseqe_amd.sqlparse import parse_expr, parse_select
    out = {}
    g = torch.Generator(device=dev); g.manual_seed(3)
    try:   # ---- config 3 ----
        a = torch.randint(0, 1000, (n,), dtype=torch.int32, device=dev, generator=g)
        b = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
        c = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 1100, generator=g)
        d = torch.randint(0, 10, (n,), dtype=torch.int32, device=dev, generator=g)
        e = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 2, generator=g)
        torch.cuda.synchronize()
        rec = chq.DeviceRecordBatch.from_device_pointers(
            [("a", "i", a.data_ptr()), ("b", "f", b.data_ptr()), ("c", "f", c.data_ptr()), ("d", "i", d.data_ptr()), ("e", "f", e.data_ptr())], n, ctx=ctx)
        sel = parse_select("select a, a + b as ab, d * 2 as d2, e / 3.0 as e3 from t where a + b > c and d < 5.0 or e > 1.0")
        al = [[]] * 5
        fk, pw = [], []
        for it in range(4):
            o = chq.filter_record(rec, al, sel.selection, ctx=ctx)
            st = ctx.last_stats()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pr = chq.project_record(sel.projection, o, al, ctx=ctx)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            if it:
                fk.append(st["kernel_ns"] / 1e6); pw.append((t1 - t0) * 1e3)
            rows_out = o.num_rows
            pr.release(); o.release()
        fk.sort(); pw.sort()
        alg = st["bytes_read_alg"] + st["bytes_written_alg"]
        ms = fk[len(fk) // 2]
        out["config3"] = {"workload": "a + b > c and d < 5.0 or e > 1.0 over a:i32, b:f32, c:f32, d:i32, e:f32; select a, a+b, d*2, e/3.0 of the survivors",
                          "rows": n, "selectivity": rows_out / n, "filter_kernel_ms": ms, "algorithmic_bytes": alg,
                          "achieved_GBps": alg / (ms * 1e-3) / 1e9, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                          "rows_per_s": n / (ms * 1e-3), "project_call_ms": pw[len(pw) // 2]}
        del rec, a, b, c, d, e
    except Exception as err:  # noqa: BLE001 -- an extra line must never cost the headline
        out["config3"] = {"error": repr(err)}
    ctx.set_option("trim_pool", 1); torch.cuda.empty_cache()
    try:   # ---- the reference's schema and batch size, resident in HBM ----
        nb, rpb, L8 = 12_500, 10_000, 8
        m = nb * rpb
        ids = torch.arange(m, dtype=torch.int32, device=dev)
        chars = torch.randint(ord("a"), ord("z") + 1, (m * L8,), dtype=torch.uint8, device=dev, generator=g)
        offs = (torch.arange(rpb + 1, dtype=torch.int64, device=dev) * L8).to(torch.int32)
        v2 = torch.empty(m, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
        torch.cuda.synchronize()
        devs = [chq.DeviceRecordBatch.from_device_pointers(
            [("id", "i", ids.data_ptr() + 4 * k * rpb), ("value1", "u", offs.data_ptr(), chars.data_ptr() + L8 * k * rpb),
             ("value2", "f", v2.data_ptr() + 4 * k * rpb)], rpb, ctx=ctx) for k in range(nb)]
        grp = chq.RecordGroup(devs, ctx)
        pred = parse_expr("id % 2 = 0")
        ts = []
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            big, rows = chq.filter_records_coalesced(grp, [[], [], []], pred, ctx=ctx)
            t1 = time.perf_counter()
            kept = big.num_rows
            big.release()
            if it:
                ts.append((t1 - t0) * 1e3)
        ts.sort()
        t0 = time.perf_counter()
        for k in range(200):
            chq.filter_record(devs[k], [[], [], []], pred, ctx=ctx).release()
        loop_us = (time.perf_counter() - t0) / 200 * 1e6
        call_ms = ts[len(ts) // 2]
        alg = m * (4 + 4 + L8 + 4) + kept * (4 + 4 + L8 + 4)
        out["reference_schema_group"] = {
            "workload": f"{nb} x {rpb}-row batches of id:Int32, value1:Utf8(8), value2:Float32 resident in HBM, WHERE id % 2 = 0, "
                        "ONE chq_filter_records_coalesced call (one launch over the batches as they lie)",
            "rows": m, "rows_out": kept, "call_ms": call_ms, "rows_per_s": m / (call_ms * 1e-3), "us_per_batch": call_ms * 1e3 / nb,
            "per_batch_call_us": loop_us, "speedup_vs_per_batch_calls": loop_us * nb / (call_ms * 1e3),
            "algorithmic_GBps": alg / (call_ms * 1e-3) / 1e9}
        grp.release()
        del devs, ids, chars, v2
    except Exception as err:  # noqa: BLE001
        out["reference_schema_group"] = {"error": repr(err)}
    try:   # ---- the rows either side of the path: Parquet scan -> filter + project -> Parquet write, every stage on the device ----
        import io
        import numpy as np
        import pyarrow as pa
        import pyarrow.compute as pc
        import pyarrow.parquet as pq
        rows = 8_000_000
        rng = np.random.default_rng(0)
        letters = rng.integers(ord("a"), ord("z") + 1, (rows, 8), dtype=np.uint8)
        value1 = pa.Array.from_buffers(pa.utf8(), rows, [None, pa.py_buffer((np.arange(rows + 1, dtype=np.int32) * 8).tobytes()), pa.py_buffer(letters.tobytes())])
        t = pa.table({"id": pa.array(np.arange(rows, dtype=np.int32)), "value1": value1, "value2": pa.array((rng.random(rows) * 100).astype(np.float32))})
        sink = io.BytesIO()
        pq.write_table(t, sink, compression="none", row_group_size=1 << 20, data_page_size=1 << 20)   # the reference's writer settings
        raw = sink.getvalue()
        sel = parse_select("select id, value1, value2 * 2.0 as twice from t where value2 > 10.0")
        t0 = time.perf_counter(); host_tab = pq.read_table(io.BytesIO(raw)); host_read_ms = (time.perf_counter() - t0) * 1e3
        best, stages, kept_rows, out_bytes = 1e9, None, 0, 0
        for it in range(4):
            t0 = time.perf_counter()
            f = chq.ParquetFile(raw)
            devs = f.read_row_groups(ctx=ctx)
            t1 = time.perf_counter()
            kept_rows = out_bytes = 0
            tf = tw = 0.0
            for d in devs:
                a = time.perf_counter()
                res = chq.filter_project_record(sel.selection, sel.projection, d, [[], [], []], ctx=ctx)
                b = time.perf_counter()
                img = chq.record_to_parquet(res, ctx=ctx, copy=False)
                if it == 0 and kept_rows == 0:   # parity of the first result file: pyarrow reads it back
                    back = pq.read_table(io.BytesIO(bytes(img.view)))
                    m0 = host_tab.slice(0, d.num_rows).filter(pc.greater(host_tab["value2"].slice(0, d.num_rows), pa.scalar(10.0, pa.float32())))
                    assert back.num_rows == m0.num_rows and back["id"].combine_chunks().equals(m0["id"].combine_chunks()) and \
                        back["value1"].combine_chunks().equals(m0["value1"].combine_chunks())
                out_bytes += len(img); kept_rows += res.num_rows
                img.release(); res.release(); d.release()
                tf += b - a; tw += time.perf_counter() - b
            f.close()
            total = time.perf_counter() - t0
            if total < best:
                best, stages = total, {"scan_ms": (t1 - t0) * 1e3, "filter_project_ms": tf * 1e3, "write_ms": tw * 1e3}
        out["parquet_pipeline"] = {
            "workload": f"{rows} rows of id:Int32, value1:Utf8(8), value2:Float32 in an uncompressed Parquet file ({len(raw) / 1e6:.0f} MB, 1 Mi-row row groups) -> "
                        "chq_parquet_read_row_groups -> chq_filter_project_record (value2 > 10.0; id, value1, value2 * 2.0) -> chq_record_to_parquet",
            "call_ms": best * 1e3, "input_rows_per_s": rows / best, "rows_out": kept_rows, "file_bytes_out": out_bytes, **stages,
            "pyarrow_read_only_ms_on_this_host": host_read_ms}
        del t, value1, letters, raw, host_tab
    except Exception as err:  # noqa: BLE001
        out["parquet_pipeline"] = {"error": repr(err)}
    ctx.set_option("trim_pool", 1); torch.cuda.empty_cache()
    return out


def kernel_source_hash():
    """sha256 over the device sources: ties a committed PMC summary to the binary it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip", "device_program.h"):
        h.update(open(os.path.join(ROOT, "chapterhouseqe_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.launcher_selftest:
        return launcher_selftest(args)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the filter path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import chapterhouseqe_amd as chq
    from chapterhouseqe_amd.sqlparse import parse_expr

    n = args.rows
    dev = torch.device("cuda", local_rank)
    # synthetic data: schema value0,value1,value2 : Float32, i.i.d. U[0,100) (create_sample_data.rs:184-189),
    # counter-based generator seeded 0xC0FFEE + column (+ rank: every GPU holds a different shard)
    cols = []
    for c in range(3):
        g = torch.Generator(device=dev)
        g.manual_seed(0xC0FFEE + c + 1000 * rank)
        t = torch.empty(n, dtype=torch.float32, device=dev)
        t.uniform_(0.0, 100.0, generator=g)
        cols.append(t)
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream().cuda_stream
    ctx = chq.Context(local_rank, stream=stream)
    ctx.set_option("time_kernels", 1)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    rec = chq.DeviceRecordBatch.from_device_pointers(
        [(f"value{c}", "f", cols[c].data_ptr()) for c in range(3)], n, ctx=ctx, keepalive=cols)
    expr = parse_expr(args.predicate)
    aliases = chq.get_record_table_aliases(None, rec)

    import pyarrow as pa
    host_schema = pa.schema([pa.field(f"value{c}", pa.float32(), nullable=False) for c in range(3)])

    def step():
        out = chq.filter_record(rec, aliases, expr, ctx=ctx)   # result stays in HBM
        st = ctx.last_stats()
        out.release()
        return st

    # ---- validation of the timed configuration against the oracle on a prefix ---------------------------
    validated = None
    if rank == 0 and args.validate_rows > 0:
        import numpy as np
        import pyarrow as pa
        from oracle import oracle as O
        from tests.helpers import batches_identical
        m = min(n, args.validate_rows)
        host = pa.RecordBatch.from_arrays([pa.array(cols[c][:m].cpu().numpy()) for c in range(3)], schema=host_schema)
        exp = O.filter_record(host, aliases, expr)
        sub = chq.DeviceRecordBatch.from_device_pointers([(f"value{c}", "f", cols[c].data_ptr()) for c in range(3)], m, ctx=ctx)
        got = chq.filter_record(sub, aliases, expr, ctx=ctx)
        got_h = got.to_host()
        validated = bool(batches_identical(got_h, exp))
        got.release()
        if not validated:
            raise SystemExit("bench: GPU result differs from the oracle on the validation prefix")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ns = 0
    stats = None
    for _ in range(args.steps):
        stats = step()
        kernel_ns += stats["kernel_ns"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    rows_out = stats["rows_out"]
    per_gpu = [n * args.steps / elapsed]     # every rank's own rate; `value` uses the MAX elapsed time over the ranks
    if world > 1:
        gathered = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor(per_gpu, dtype=torch.float64, device=dev))
        per_gpu = [float(g.item()) for g in gathered]
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rows_out, kernel_ns], dtype=torch.int64, device=dev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rows_out_total = int(rr[0].item())
        kernel_ns = int(rr[1].item()) // world
    else:
        rows_out_total = rows_out

    if rank == 0:
        total_rows = n * world
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rows * args.steps / elapsed
        # algorithmic bytes of ONE launch of the dominant kernel (filter_fused_kernel) on one GPU:
        # 12 B/row read once + 12 B per surviving row written (SURVEY.md section 8 d)
        alg_bytes = stats["bytes_read_alg"] + stats["bytes_written_alg"]
        kern_ms = kernel_ns / args.steps / 1e6
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3 passes of
        # this same command; the counters cannot be read from inside the process, so the committed summary is quoted)
        # a summary is only quoted for the kernel sources it was measured on (it carries their hash)
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r2", "bench_pmc_hbm.json")
        if os.path.exists(pmc) and args.predicate == PREDICATE and n == 1_000_000_000:
            j = json.load(open(pmc))
            if j.get("kernel_source_sha256") == kernel_source_hash():
                traffic = j["fetch_bytes_corrected"] + j["write_bytes"]
                traffic_src = "profiles/r2/bench_pmc_hbm.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command, same kernel sources)"
            else:
                traffic_src = "profiles/r2/bench_pmc_hbm.json is stale: the kernel sources changed since that PMC pass"
        # a plain device copy measured in this same process, so the fraction is not hostage to the datasheet peak
        # (SURVEY.md section 8 d): one column copied into a scratch tensor, read + write bytes / time
        copy_gbps = None
        try:
            dst = torch.empty_like(cols[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dst.copy_(cols[0]); torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                dst.copy_(cols[0])
            e1.record(); torch.cuda.synchronize()
            copy_gbps = 2 * cols[0].numel() * 4 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del dst
        except RuntimeError:
            pass
        out = {
            "metric": "filtered rows/sec (input rows), SELECT * WHERE value2 > 10.0, 3 x f32, device-resident",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config 2: SELECT * WHERE {args.predicate} over one {n}-row record batch per GPU, "
                                   "3 x Float32 U[0,100), non-null, inputs and outputs in HBM",
                       "rows_per_gpu": n, "selectivity": rows_out / n, "parallelism": f"1 operator instance per GPU x {world}",
                       "rows_out_total": rows_out_total, "validated_vs_oracle_rows": args.validate_rows if validated else 0,
                       "per_gpu_rows_per_s": per_gpu},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "filter_fused_kernel<1024,16,FULL> (+ one-workgroup launch for the partial tail tile)",
                         "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "read_frac_of_peak": (stats["bytes_read_alg"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if kern_ms > 0 else None,
                         "device_copy_GBps_same_run": copy_gbps,
                         "frac_of_device_copy": (achieved / copy_gbps) if (achieved and copy_gbps) else None},
        }
        host = None
        if not args.no_cpu_baseline and world == 1:
            import pyarrow as pa
            m = min(n, args.cpu_rows)
            host = pa.RecordBatch.from_arrays([pa.array(cols[c][:m].cpu().numpy()) for c in range(3)], schema=host_schema)
        if world == 1 and not args.no_extra and n == 1_000_000_000:
            rec.release()
            del rec, cols
            ctx.set_option("trim_pool", 1)
            torch.cuda.empty_cache()
            out["extra"] = extra_lines(chq, torch, dev, ctx, n)
        if host is not None:   # the CPU leg is timed at N = 1 only
            from oracle import oracle as O
            kept, secs = O.filter_table_batched(host, aliases, expr, batch_rows=10_000)
            out["cpu_baseline"] = {"value": m / secs, "unit": "rows/s", "cores": 1, "kind": "port",
                                   "sample": f"first {m} rows of the same columns, 10 000-row batches, one thread = one "
                                             f"operator instance (filter_task.rs:86-125), {secs:.1f} s, kept {kept} rows",
                                   "host_cpus": os.cpu_count()}
            # (ii) one instance per host core over disjoint slices of the same rows (the C oracle runs outside the GIL)
            import threading
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            T = max(1, min(ncpu, 256))   # every host CPU this process may run on (the box's share for one GPU)
            if T > 1:
                per = (m // T) // 10_000 * 10_000
                if per > 0:
                    kept_t = [0] * T
                    def work(i):
                        kept_t[i] = O.filter_table_batched(host.slice(i * per, per), aliases, expr, batch_rows=10_000)[0]
                    ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
                    w0 = time.perf_counter()
                    for th in ths:
                        th.start()
                    for th in ths:
                        th.join()
                    wsecs = time.perf_counter() - w0
                    out["cpu_baseline"]["all_cores"] = {"value": per * T / wsecs, "unit": "rows/s", "cores": T,
                                                        "sample": f"{T} threads x {per} rows, {wsecs:.1f} s wall, kept {sum(kept_t)} rows"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
