#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X filter path (BASELINE.json / SURVEY.md section 8 d, config 2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One *step* = one `filter_record` call (`SELECT * WHERE value2 > 10.0`) over ONE device-resident record batch with three
non-null Float32 columns U[0,100), through the C ABI (libchq.so): predicate evaluation + order-preserving compaction of all
three columns into freshly allocated HBM buffers, including the read-back of the output row count (the Arrow length the
caller needs).  Inputs are in HBM before the timed region.

N > 1: one operator instance per GPU (one process per GPU).  Batches are independent in the reference
(filter_task.rs:86-125; the RecordPool hands disjoint records to instances, exchange_operator.rs:621-667), so there is no
data-path collective; ranks only all-reduce the elapsed time (MAX) and row counts (SUM) after the timed region.
  --scaling strong (default): the `--rows` rows (1e9) are split evenly, rows / N per GPU (SURVEY.md section 8 d: "repeat
                              config 2 at 2/4/8 GPUs, n split evenly"); the weak run rides along as `extra.weak`
  --scaling weak:             every GPU filters its own `--rows`-row shard of an N x rows table
  --config 5:                 the huge_simple.sql shape (sample_queries/huge_simple.sql:3-4): every GPU holds a 1.25 B-row
                              shard of id:Int32, value1:Utf8(8), value2:Float32 as device batches (string bytes below
                              2 GiB per batch: int32 offsets) and filters `id % 2 = 0` through chq_filter_records;
                              8 GPUs = the 10 B-row BASELINE config.  `--gathered` additionally ships the survivors to
                              rank 0 (Arrow IPC body in HBM + RCCL point-to-point over xGMI) and reports that number
                              separately: it is link-bound, not HBM-bound.
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
    ap.add_argument("--rows", type=int, default=None,
                    help="config 2: TOTAL rows (strong scaling, default 1e9: split evenly over the GPUs) or rows per GPU (weak); "
                         "config 5: rows per GPU (default 1.25e9 = one eighth of the 10 B-row table)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--config", type=int, choices=[2, 5], default=2)
    ap.add_argument("--batch-rows", type=int, default=125_000_000, help="config 5: rows per device batch (8-byte strings: < 2 GiB per batch)")
    ap.add_argument("--gathered", action="store_true", help="config 5, N > 1: also ship every survivor to rank 0 over RCCL and report it separately")
    ap.add_argument("--no-weak", action="store_true", help="strong scaling, N > 1: skip the extra weak-scaling measurement")
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


def plan_rows(args, world, rank):
    """-> (rows of this rank, total rows of the job, scaling label).  Config 2, strong: `--rows` (1e9) split evenly, the
    remainder going to the lowest ranks; weak: `--rows` per GPU.  Config 5: a fixed per-GPU shard (1.25e9 rows = one eighth
    of the 10 B-row table), so 8 GPUs are the BASELINE config -- weak by construction."""
    if args.config == 5:
        per = args.rows if args.rows is not None else 1_250_000_000
        return per, per * world, "weak"
    total = args.rows if args.rows is not None else 1_000_000_000
    if args.scaling == "weak":
        return total, total * world, "weak"
    return total // world + (1 if rank < total % world else 0), total, "strong"


def reduce_timing(dist, torch, world, dev, elapsed, rows_this_rank, steps, extra_ints=()):
    """MAX of the elapsed time over the ranks, SUM of the integer counters, every rank's own rate"""
    per_gpu = [rows_this_rank * steps / elapsed]
    sums = [int(x) for x in extra_ints]
    if world > 1:
        gathered = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor(per_gpu, dtype=torch.float64, device=dev))
        per_gpu = [float(g.item()) for g in gathered]
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        if sums:
            rr = torch.tensor(sums, dtype=torch.int64, device=dev)
            dist.all_reduce(rr, op=dist.ReduceOp.SUM)
            sums = [int(v) for v in rr.tolist()]
    return elapsed, per_gpu, sums


def launcher_selftest(args):
    """The multi-rank protocol of main() on CPU (gloo): the row plan of the selected mode (strong / weak / config 5),
    barrier, timed region, barrier, MAX over ranks, SUM of counts, ONE JSON line from rank 0, non-zero exit when a rank
    fails.  No GPU, no libchq: the "work" is a sleep proportional to the rank's rows."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    if rank == args.selftest_fail_rank:
        raise SystemExit(3)
    rows, total, scaling = plan_rows(args, world, rank)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, per_rank, sums = reduce_timing(dist, torch, world, None, elapsed, rows, 1, [rows, 1000 * (rank + 1)])
    line = {"selftest": True, "n_gpus": world, "value": sums[0] / elapsed, "per_gpu_rows_per_s": per_rank, "rows_total": sums[1],
            "scaling": scaling, "config": args.config, "rows_this_rank": rows, "rows_planned_total": total, "rows_summed": sums[0]}
    if args.gathered and world > 1:   # the gather protocol of config 5: every rank's survivors end up on rank 0, peer after peer
        from chapterhouseqe_amd.operators import distributed as D
        import pyarrow as pa
        got = 0
        if rank == 0:
            for src in range(1, world):
                for _ in range(2):
                    rid, rec, _al = D.recv_record(src)
                    got += rec.num_rows
        else:
            for b in range(2):
                D.send_record(pa.RecordBatch.from_pydict({"id": list(range(10 * rank + b))}), b, 0)
        line["gathered_rows"] = got
    if rank == 0:
        print(json.dumps(line), flush=True)
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
    try:   # ---- BASELINE config 5: ONE rank's 1.25 B-row shard of the 10 B-row huge_simple.sql table (what --config 5 runs per GPU) ----
        rows, batch_rows = 1_250_000_000, 125_000_000
        batches, keep = build_config5_shard(chq, torch, dev, ctx, rows, batch_rows)
        grp = chq.RecordGroup(batches, ctx)
        pred = parse_expr(C5_PREDICATE)
        outs = chq.filter_records(grp, [[], [], []], pred, ctx=ctx)
        check_config5_outputs(torch, keep, outs, 0); check_config5_outputs(torch, keep, outs, len(outs) - 1)
        for o in outs:
            o.release()
        run_config5_steps(chq, ctx, grp, pred, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rows_out, kernel_ns, alg = run_config5_steps(chq, ctx, grp, pred, 3)
        torch.cuda.synchronize(); secs = (time.perf_counter() - t0) / 3
        kms = kernel_ns / 3 / 1e6
        out["config5_shard"] = {
            "workload": f"one GPU's shard of config 5: WHERE {C5_PREDICATE} over {rows} rows of id:Int32, value1:Utf8(8), value2:Float32 in "
                        f"{len(batches)} device batches (10 GB of string bytes), ONE chq_filter_records call, one output per input batch",
            "rows": rows, "rows_out": rows_out, "call_ms": secs * 1e3, "rows_per_s": rows / secs, "kernel_ms": kms, "algorithmic_bytes": alg,
            "achieved_GBps": alg / (kms * 1e-3) / 1e9, "frac": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        grp.release()
        del batches, keep
    except Exception as err:  # noqa: BLE001
        out["config5_shard"] = {"error": repr(err)}
    ctx.set_option("trim_pool", 1); torch.cuda.empty_cache()
    return out


def kernel_source_hash():
    """sha256 over the device sources: ties a committed PMC summary to the binary it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip", "device_program.h"):
        h.update(open(os.path.join(ROOT, "chapterhouseqe_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def pmc_traffic(name, n_rows, predicate_ok=True):
    """HBM bytes per launch from a committed PMC summary (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate rocprofv3 passes:
    the counters cannot be read from inside the process).  Quoted only when the summary was measured on these kernel
    sources (it carries their hash) AND on this per-GPU row count."""
    for rnd in ("r3", "r2"):
        path = os.path.join(ROOT, "profiles", rnd, name)
        if not os.path.exists(path):
            continue
        j = json.load(open(path))
        rel = f"profiles/{rnd}/{name}"
        if not predicate_ok or j.get("rows", 1_000_000_000) != n_rows:
            return None, f"{rel} was measured on another workload / per-GPU row count"
        if j.get("kernel_source_sha256") != kernel_source_hash():
            return None, f"{rel} is stale: the kernel sources changed since that PMC pass"
        return j["fetch_bytes_corrected"] + j["write_bytes"], f"{rel} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command, same kernel sources)"
    return None, None


def make_config2_columns(torch, dev, n, rank):
    """schema value0,value1,value2 : Float32, i.i.d. U[0,100) (create_sample_data.rs:184-189), counter-based generator seeded
    0xC0FFEE + column (+ rank: every GPU holds a different shard)"""
    cols = []
    for c in range(3):
        g = torch.Generator(device=dev)
        g.manual_seed(0xC0FFEE + c + 1000 * rank)
        t = torch.empty(n, dtype=torch.float32, device=dev)
        t.uniform_(0.0, 100.0, generator=g)
        cols.append(t)
    torch.cuda.synchronize()
    return cols


def time_config2(chq, torch, dist, world, ctx, cols, n, expr, aliases, steps, warmup):
    """W untimed steps, then exactly K timed steps between barrier + synchronize on both sides -> (elapsed s, kernel ns, stats)"""
    rec = chq.DeviceRecordBatch.from_device_pointers([(f"value{c}", "f", cols[c].data_ptr()) for c in range(3)], n, ctx=ctx, keepalive=cols)

    def step():
        out = chq.filter_record(rec, aliases, expr, ctx=ctx)   # result stays in HBM
        st = ctx.last_stats()
        out.release()
        return st

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ns, stats = 0, None
    for _ in range(steps):
        stats = step()
        kernel_ns += stats["kernel_ns"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rec.release()
    return elapsed, kernel_ns, stats


# ---- config 5: the huge_simple.sql shape --------------------------------------------------------------------------------
C5_PREDICATE = "id % 2 = 0"
C5_STRLEN = 8


def build_config5_shard(chq, torch, dev, ctx, rows, batch_rows, rank=0):
    """One GPU's shard of the 10 B-row table of sample_queries/huge_simple.sql:3-4 (schema of create_sample_data.rs:124-133):
    id:Int32 = 0 .. rows-1 (per shard: 10 B overflows Int32), value1:Utf8 of 8 lowercase letters, value2:Float32 U[0,100),
    as device-resident batches of `batch_rows` rows -- a Utf8 array holds < 2 GiB of bytes (int32 offsets), so 1.25 B rows x
    8 bytes need at least five.  Every batch has its OWN offsets buffer (sharing one would let the offsets sit in L2).
    -> (batches, keepalive tensors)"""
    keep, batches = [], []
    L = C5_STRLEN
    assert batch_rows * L < 2**31
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED + 1000 * rank)
    for r0 in range(0, rows, batch_rows):
        m = min(batch_rows, rows - r0)
        ids = torch.arange(r0, r0 + m, dtype=torch.int32, device=dev)
        chars = torch.randint(ord("a"), ord("z") + 1, (m * L,), dtype=torch.uint8, device=dev, generator=g)
        offs = torch.arange(0, (m + 1) * L, L, dtype=torch.int32, device=dev)
        v2 = torch.empty(m, dtype=torch.float32, device=dev).uniform_(0.0, 100.0, generator=g)
        keep += [ids, chars, offs, v2]
        batches.append(chq.DeviceRecordBatch.from_device_pointers(
            [("id", "i", ids.data_ptr()), ("value1", "u", offs.data_ptr(), chars.data_ptr()), ("value2", "f", v2.data_ptr())], m, ctx=ctx))
    torch.cuda.synchronize()
    return batches, keep


def run_config5_steps(chq, ctx, grp, expr, steps):
    """`steps` passes over the shard: ONE chq_filter_records call per pass (the caller's loop of filter_task.rs:86-125 hoisted
    below the boundary, one output per input record id) -> (rows out of the last pass, kernel ns, alg. bytes of one pass)"""
    kernel_ns, rows_out, alg = 0, 0, 0
    for _ in range(steps):
        outs = chq.filter_records(grp, [[], [], []], expr, ctx=ctx)
        st = ctx.last_stats()
        kernel_ns += st["kernel_ns"]
        alg = st["bytes_read_alg"] + st["bytes_written_alg"]
        rows_out = sum(o.num_rows for o in outs)
        for o in outs:
            o.release()
    return rows_out, kernel_ns, alg


def check_config5_outputs(torch, keep, outs, batch_index=0):
    """size-independent properties of one output batch against its input batch: exactly the even ids, in order, each with
    its own string and float"""
    ids, chars, offs, v2 = keep[4 * batch_index: 4 * batch_index + 4]
    o = outs[batch_index]
    m = ids.numel()
    assert o.num_rows == (m + 1 - int(ids[0].item()) % 2) // 2, (o.num_rows, m)
    oid = o.column_tensor(0, torch)
    assert torch.equal(oid, ids[(ids % 2) == 0])
    ov = o.column_tensor(2, torch)
    assert torch.equal(ov.view(torch.int32), v2[(ids % 2) == 0].view(torch.int32))
    ooffs, odata = o.utf8_tensors(1, torch)
    assert torch.equal(ooffs - ooffs[0], torch.arange(0, (o.num_rows + 1) * C5_STRLEN, C5_STRLEN, dtype=torch.int32, device=ooffs.device))
    sel = chars.view(m, C5_STRLEN)[(ids % 2) == 0].reshape(-1)
    first = int(ooffs[0].item())
    assert torch.equal(odata[first: first + sel.numel()], sel)


def config5_main(args, chq, torch, dist, rank, local_rank, world, dev, ctx, comm_dev=None):
    comm_dev = comm_dev if comm_dev is not None else dev
    from chapterhouseqe_amd.sqlparse import parse_expr
    rows, total_rows, scaling = plan_rows(args, world, rank)
    batches, keep = build_config5_shard(chq, torch, dev, ctx, rows, args.batch_rows, rank)
    grp = chq.RecordGroup(batches, ctx)
    expr = parse_expr(C5_PREDICATE)
    validated = None
    if rank == 0 and args.validate_rows > 0:   # the first batch's prefix against the oracle, every batch by its properties
        import pyarrow as pa
        from oracle import oracle as O
        from tests.helpers import batches_identical
        m = min(batches[0].num_rows, args.validate_rows)
        ids, chars, offs, v2 = keep[:4]
        host = pa.RecordBatch.from_arrays(
            [pa.array(ids[:m].cpu().numpy()),
             pa.Array.from_buffers(pa.utf8(), m, [None, pa.py_buffer(offs[: m + 1].cpu().numpy().tobytes()), pa.py_buffer(chars[: m * C5_STRLEN].cpu().numpy().tobytes())]),
             pa.array(v2[:m].cpu().numpy())],
            schema=pa.schema([pa.field("id", pa.int32(), False), pa.field("value1", pa.utf8(), False), pa.field("value2", pa.float32(), False)]))
        sub = chq.DeviceRecordBatch.from_device_pointers(
            [("id", "i", ids.data_ptr()), ("value1", "u", offs.data_ptr(), chars.data_ptr()), ("value2", "f", v2.data_ptr())], m, ctx=ctx)
        got = chq.filter_record(sub, [[], [], []], expr, ctx=ctx)
        validated = bool(batches_identical(got.to_host(), O.filter_record(host, [[], [], []], expr)))
        got.release()
        if not validated:
            raise SystemExit("bench: GPU result differs from the oracle on the validation prefix (config 5)")
        outs = chq.filter_records(grp, [[], [], []], expr, ctx=ctx)
        for b in range(len(outs)):
            check_config5_outputs(torch, keep, outs, b)
        for o in outs:
            o.release()
    run_config5_steps(chq, ctx, grp, expr, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows_out, kernel_ns, alg = run_config5_steps(chq, ctx, grp, expr, args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed, per_gpu, (rows_out_total, kernel_ns_sum, alg_sum) = reduce_timing(dist, torch, world, comm_dev, elapsed, rows, args.steps, [rows_out, kernel_ns, alg])
    gathered = None
    if args.gathered and world > 1:
        gathered = gather_to_rank0(chq, torch, dist, rank, world, dev, ctx, grp, expr)
    if rank == 0:
        kern_ms = kernel_ns_sum / world / args.steps / 1e6          # per GPU and pass (the launches of one pass run back to back)
        alg_per_gpu = alg_sum / world
        achieved = alg_per_gpu / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
        out = {
            "metric": "filtered rows/sec (input rows), SELECT * WHERE id % 2 = 0, id:Int32, value1:Utf8(8), value2:Float32, device-resident",
            "value": total_rows * args.steps / elapsed, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "i32", "data": "synthetic",
            "config": {"workload": f"config 5 (huge_simple.sql shape): WHERE {C5_PREDICATE} over a {rows}-row shard per GPU in "
                                   f"{len(batches)} device batches of {args.batch_rows} rows (one chq_filter_records call per step, one output per "
                                   f"input batch); {world} GPU(s) = {total_rows} rows of the 10 B-row table",
                       "rows_per_gpu": rows, "rows_total": total_rows, "selectivity": rows_out_total / total_rows,
                       "parallelism": f"1 operator instance per GPU x {world}, no data-path collective",
                       "validated_vs_oracle_rows": args.validate_rows if validated else 0, "per_gpu_rows_per_s": per_gpu},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": None,
                         "kernel": "per batch: utf8_uniform_kernel (proves the strings all have 8 bytes) + filter_fused_kernel<1024,16,FULL> (+ tail) "
                                   "with the string column copied as a fixed-width column + iota_offsets_kernel; all three are timed",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_per_gpu,
                         "aggregate_GBps": (achieved * world) if achieved else None,
                         "frac_of_n_x_peak": (achieved / HBM_PEAK_GBPS) if achieved else None},
        }
        if gathered is not None:
            out["extra"] = {"gathered": gathered}
        print(json.dumps(out), flush=True)
    grp.release()


def gather_to_rank0(chq, torch, dist, rank, world, dev, ctx, grp, expr):
    """The xGMI-bound variant of config 5 (SURVEY.md section 8 e): every survivor is shipped to rank 0 -- each output batch as
    the reference's wire format with the body in HBM (chq_record_to_ipc) and ONE RCCL point-to-point send per body
    (operators/distributed.py).  Rank 0 drains its peers one after the other, so one xGMI link is busy at a time."""
    from chapterhouseqe_amd.operators import distributed as D
    outs = chq.filter_records(grp, [[], [], []], expr, ctx=ctx)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    rows = nbytes = 0
    if rank == 0:
        for src in range(1, world):
            for _ in range(len(outs)):
                rid, rec, _al = D.recv_device_record(src, ctx)
                rows += rec.num_rows
                rec.release()
    else:
        for b, o in enumerate(outs):
            D.send_device_record(o, b, 0, None)
    torch.cuda.synchronize()
    dist.barrier()
    secs = time.perf_counter() - t0
    mine = sum(o.num_rows for o in outs)
    for o in outs:
        o.release()
    shipped = torch.tensor([0 if rank == 0 else mine], dtype=torch.int64, device=dev)
    dist.all_reduce(shipped, op=dist.ReduceOp.SUM)
    shipped = int(shipped.item())
    nbytes = shipped * (4 + 4 + C5_STRLEN + 4)
    return {"what": "survivors of ranks 1..N-1 shipped to rank 0 (Arrow IPC body in HBM, one RCCL send per output batch, peers drained one after the other)",
            "rows_shipped": shipped, "rows_received": rows, "seconds": secs, "GBps_into_rank0": nbytes / secs / 1e9,
            "rows_per_s": shipped / secs}


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
    # CHQ_BENCH_REHEARSE=1: the N-rank code path on a box with ONE GPU (development boxes; tests/test_gpu_scale.py) -- every
    # rank runs on device 0 and the ranks talk through gloo (RCCL refuses two ranks on one device).  Timings of such a run
    # mean nothing (the ranks share the card); the row plan, the barriers, the reductions and the JSON line are the real ones.
    rehearse = os.environ.get("CHQ_BENCH_REHEARSE") == "1" and world > 1
    gpu_index = 0 if rehearse else local_rank
    torch.cuda.set_device(gpu_index)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import chapterhouseqe_amd as chq
    from chapterhouseqe_amd.sqlparse import parse_expr

    dev = torch.device("cuda", gpu_index)
    comm_dev = torch.device("cpu") if rehearse else dev   # where the reduced scalars live (gloo reduces host tensors)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = chq.Context(gpu_index, stream=stream)
    ctx.set_option("time_kernels", 1)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    if args.config == 5:
        if rehearse and args.gathered:
            raise SystemExit("--gathered needs RCCL: not available in the one-GPU rehearsal")
        config5_main(args, chq, torch, dist, rank, local_rank, world, dev, ctx, comm_dev)
        if world > 1:
            dist.destroy_process_group()
        return

    n, total_rows, scaling = plan_rows(args, world, rank)
    cols = make_config2_columns(torch, dev, n, rank)
    expr = parse_expr(args.predicate)
    import pyarrow as pa
    host_schema = pa.schema([pa.field(f"value{c}", pa.float32(), nullable=False) for c in range(3)])
    aliases = [[], [], []]

    # ---- validation of the timed configuration against the oracle on a prefix ---------------------------
    validated = None
    if rank == 0 and args.validate_rows > 0:
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

    elapsed, kernel_ns, stats = time_config2(chq, torch, dist, world, ctx, cols, n, expr, aliases, args.steps, args.warmup)
    rows_out = stats["rows_out"]
    elapsed, per_gpu, (rows_out_total, kernel_ns_sum) = reduce_timing(dist, torch, world, comm_dev, elapsed, n, args.steps, [rows_out, kernel_ns])
    kernel_ns = kernel_ns_sum // world

    # ---- N > 1, strong scaling: the weak-scaling run (every GPU its own `total_rows`-row shard) rides along -------------
    weak = None
    if world > 1 and scaling == "strong" and not args.no_weak:
        del cols
        torch.cuda.empty_cache()
        wcols = make_config2_columns(torch, dev, total_rows, rank)
        w_elapsed, w_kns, w_stats = time_config2(chq, torch, dist, world, ctx, wcols, total_rows, expr, aliases, args.steps, args.warmup)
        w_elapsed, w_per_gpu, (w_rows_out, w_kns_sum) = reduce_timing(dist, torch, world, comm_dev, w_elapsed, total_rows, args.steps, [w_stats["rows_out"], w_kns])
        w_alg = w_stats["bytes_read_alg"] + w_stats["bytes_written_alg"]
        w_kms = w_kns_sum / world / args.steps / 1e6
        weak = {"scaling": "weak", "rows_per_gpu": total_rows, "value": total_rows * world * args.steps / w_elapsed, "unit": "rows/s",
                "ms_per_step": w_elapsed / args.steps * 1e3, "per_gpu_rows_per_s": w_per_gpu, "kernel_ms": w_kms,
                "frac": (w_alg / (w_kms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if w_kms > 0 else None}
        cols = wcols

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rows * args.steps / elapsed
        # algorithmic bytes of ONE launch of the dominant kernel (filter_fused_kernel) on one GPU:
        # 12 B/row read once + 12 B per surviving row written (SURVEY.md section 8 d)
        alg_bytes = stats["bytes_read_alg"] + stats["bytes_written_alg"]
        kern_ms = kernel_ns / args.steps / 1e6
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
        traffic, traffic_src = pmc_traffic("bench_pmc_hbm.json", n, args.predicate == PREDICATE)
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
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config 2: SELECT * WHERE {args.predicate} over {total_rows} rows of 3 x Float32 U[0,100), non-null, "
                                   f"one {n}-row record batch per GPU ({'the table split evenly' if scaling == 'strong' else 'one table per GPU'}), "
                                   "inputs and outputs in HBM",
                       "rows_total": total_rows, "rows_per_gpu": n, "selectivity": rows_out / n,
                       "parallelism": f"1 operator instance per GPU x {world}, no data-path collective",
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
        if weak is not None:
            out["extra"] = {"weak": weak}
        host = None
        if not args.no_cpu_baseline and world == 1:
            m = min(n, args.cpu_rows)
            host = pa.RecordBatch.from_arrays([pa.array(cols[c][:m].cpu().numpy()) for c in range(3)], schema=host_schema)
        if world == 1 and not args.no_extra and n == 1_000_000_000:
            del cols
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
