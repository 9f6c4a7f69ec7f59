"""GPU: the operator pipeline on the HIP kernels -- filter instances sharing one exchange, batches staying in HBM
between filter and materialize."""
import os

import pyarrow.parquet as pq
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sample_data import simple_batches
from chapterhouseqe_amd.sqlparse import parse_select
from oracle import oracle as O

from .test_operators import run_pipeline

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("instances", [1, 2])
def test_pipeline_on_gpu(tmp_path, instances):
    batches = simple_batches(20_000, 8, 999)
    sql = "select id, value1, id + 10.0 as id_plus_10, (value2 + 10) / 100 as value2, 1.0 / id as v3 from read_files('x') where value2 > 10.0"
    ctxs = {}

    def filter_fn(rec, al, expr):   # one context per thread = per operator instance; results stay in HBM
        import threading
        c = ctxs.setdefault(threading.get_ident(), chq.Context(0))
        return chq.filter_record(rec, al, expr, ctx=c, device_result=True)

    runs, mrun = run_pipeline(tmp_path, filter_fn, None, instances, batches, sql)
    sel = parse_select(sql)
    d = os.path.dirname(mrun.task.files_written[0])
    for rid, b in enumerate(batches):
        al = [[] for _ in range(b.num_columns)]
        exp = O.project_record(sel.projection, O.filter_record(b, al, sel.selection), al)
        got = pq.read_table(os.path.join(d, f"rec_{rid}.parquet")).to_batches()[0]
        assert got.to_pydict() == exp.to_pydict(), rid


def test_filter_task_groups_queued_records_into_one_launch(tmp_path):
    """FilterTask(group_size=64): the queued 10 000-row records of one schema go through chq_filter_records"""
    import numpy as np
    import pyarrow as pa
    rng = np.random.default_rng(11)
    batches = [pa.RecordBatch.from_arrays([pa.array(np.arange(i * 10_000, (i + 1) * 10_000, dtype=np.int32)),
                                           pa.array((rng.random(10_000) * 100).astype(np.float32))], names=["id", "value2"])
               for i in range(40)]
    sql = "select id, value2 * 2.0 as v from read_files('x') where value2 > 10.0"
    runs, mrun = run_pipeline(tmp_path, None, None, 1, batches, sql, group_size=64)
    task = runs[0].task
    assert task.records_processed == 40 and 1 <= task.group_calls <= 3
    assert task._ctx.last_stats()["launches"] == 1   # the last group ran as one wave-packed launch
    sel = parse_select(sql)
    d = os.path.dirname(mrun.task.files_written[0])
    for rid, b in enumerate(batches):
        al = [[], []]
        exp = O.project_record(sel.projection, O.filter_record(b, al, sel.selection), al)
        got = pq.read_table(os.path.join(d, f"rec_{rid}.parquet")).to_batches()[0]
        assert got.to_pydict() == exp.to_pydict(), rid


def test_device_batch_round_trips_through_the_exchange_wire_format():
    """operators/distributed.py: a batch in HBM as the exchange ships it -- envelope + Arrow IPC metadata (host bytes) and
    the IPC body as ONE zero-copy device tensor (what RCCL send/recv moves) -- and back: every column kind, nulls, a
    sliced (offset) input; the clone stands in for the wire"""
    import torch
    from chapterhouseqe_amd.operators.distributed import _as_tensor, _envelope, _open_envelope
    from chapterhouseqe_amd.sqlparse import parse_expr
    from .helpers import batches_identical, explain_diff
    from .test_gpu_parity import make_batch
    ctx = chq.Context(0)
    device = torch.device("cuda", 0)
    for rec in (make_batch(5000, 321).slice(3, 4000), make_batch(70, 5, nulls=False), make_batch(2, 6)):
        dev = chq.DeviceRecordBatch.from_host(rec, ctx)
        enc = chq.record_to_ipc(dev, ctx=ctx, body_on_device=True)
        frame = _envelope({"record_id": 7, "table_aliases": [["t"]], "body_len": enc.body_len}, enc.header)
        body = _as_tensor(enc.body_address, enc.body_len, enc, device)
        assert body.is_cuda and body.data_ptr() == enc.body_address and body.numel() == enc.body_len      # zero copy, ONE buffer
        moved = body.clone()
        del body
        enc.release()
        meta, header = _open_envelope(frame)
        assert meta["record_id"] == 7 and meta["table_aliases"] == [["t"]]
        back = chq.record_from_ipc(header, ctx=ctx, device_result=True, body_address=moved.data_ptr(), body_len=meta["body_len"])
        del moved
        got = back.to_host()
        assert batches_identical(got, rec), explain_diff(got, rec)
        al = [[] for _ in range(rec.num_columns)]
        e = parse_expr("f32 > 0.0 and s >= 'ab' or i64 % 3 = 0")
        assert batches_identical(chq.filter_record(back, al, e, ctx=ctx).to_host(), O.filter_record(rec, al, e))
    ctx.close()


def test_peer_copy_between_two_contexts():
    """chq_record_copy_to_peer: the C-ABI data plane for a single-process worker with one context per GPU
    (hipMemcpyPeerAsync + sync_event).  One GPU here: the "peer" is a second context on the same device; with two or more
    GPUs the copy crosses xGMI (next test)."""
    import torch
    from chapterhouseqe_amd.sqlparse import parse_expr
    from .helpers import batches_identical, explain_diff
    from .test_gpu_parity import make_batch
    src_ctx = chq.Context(0)
    dst_dev = 1 if torch.cuda.device_count() > 1 else 0
    dst_ctx = chq.Context(dst_dev)
    for rec in (make_batch(30_000, 11), make_batch(5000, 12).slice(5, 3000), make_batch(1, 13), make_batch(0, 14)):
        dev = chq.DeviceRecordBatch.from_host(rec, src_ctx)
        moved = dev.copy_to_peer(dst_ctx)
        assert moved.ctx is dst_ctx and moved._cb.array.sync_event            # asynchronous: carries the event to wait on
        assert moved.column_buffer_address(0, 1) != dev.column_buffer_address(0, 1)
        al = [[] for _ in range(rec.num_columns)]
        e = parse_expr("f32 > 0.0 and s >= 'ab' or i64 % 3 = 0")
        got = chq.filter_record(moved, al, e, ctx=dst_ctx).to_host()          # waits on the sync_event inside the call
        assert batches_identical(got, O.filter_record(rec, al, e)), explain_diff(got, O.filter_record(rec, al, e))
        assert batches_identical(moved.to_host(), rec)
    src_ctx.close(); dst_ctx.close()


def _p2p_worker(rank, port, q):
    import torch
    import torch.distributed as dist
    from chapterhouseqe_amd.operators.distributed import recv_device_record, send_device_record
    from chapterhouseqe_amd.sqlparse import parse_expr
    from .test_gpu_parity import make_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", rank))
    ctx = chq.Context(rank)
    rec = make_batch(20_000, 99)
    al = [[] for _ in range(rec.num_columns)]
    e = parse_expr("f32 > 0.0 and flag")
    try:
        if rank == 0:   # filter on GPU 0, ship the survivors to GPU 1 over xGMI
            out = chq.filter_record(chq.DeviceRecordBatch.from_host(rec, ctx), al, e, ctx=ctx)
            send_device_record(out, 7, 1, al)
            # ... and a whole exchange of HBM-resident records, forwarded under their record ids
            from chapterhouseqe_amd.operators import ExchangeOperator
            from chapterhouseqe_amd.operators.distributed import forward_exchange
            ex = ExchangeOperator("mid", ["consumer"])
            for rid in range(3):
                ex.send_record(rid, chq.filter_record(chq.DeviceRecordBatch.from_host(rec.slice(rid * 5000, 5000), ctx), al, e, ctx=ctx), al)
            ex.producers_completed()
            shipped = forward_exchange(ex, "consumer", 0, dst=1)
            q.put(("sent", out.num_rows + shipped))
        else:
            rid, got, aliases = recv_device_record(0, ctx)
            from chapterhouseqe_amd.operators import ExchangeOperator
            from chapterhouseqe_amd.operators.distributed import receive_into_exchange
            from .helpers import batches_identical
            ok = rid == 7 and aliases == al and batches_identical(got.to_host(), O.filter_record(rec, al, e))
            ex = ExchangeOperator("mid_on_rank1", ["consumer"])
            ok = ok and receive_into_exchange(ex, [0], ctx=ctx) == 3
            for _ in range(3):
                rid2, r2, al2 = ex.get_next_record("consumer", 0)
                ok = ok and isinstance(r2, chq.DeviceRecordBatch) and al2 == al and \
                    batches_identical(r2.to_host(), O.filter_record(rec.slice(rid2 * 5000, 5000), al, e))
            q.put(("received", bool(ok)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_filtered_batch_moves_gpu_to_gpu_over_rccl():
    """two ranks, two GPUs, backend nccl (= RCCL): runs wherever at least two GPUs are visible"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the development boxes have one)")
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_p2p_worker, args=(r, 29731, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(180) for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    msgs = dict(q.get(timeout=5) for _ in range(2))
    assert msgs.get("received") is True and msgs.get("sent", 0) > 0


def test_steady_state_calls_do_not_grow_memory():
    """the HBM / host pools recycle: after a warm-up round, thousands of calls of every kind leave the free HBM and the
    process's resident set where they were"""
    import gc
    import resource
    import torch
    from chapterhouseqe_amd.sqlparse import parse_expr
    from .test_gpu_group import fixed_batch
    from .test_gpu_parity import make_batch
    ctx = chq.Context(0)
    sel = parse_select("select id, value1 + value2 as s from t where value2 > 10.0")
    mixed = [make_batch(n, n) for n in (1000, 5000)]
    plain = [fixed_batch(n, n, with_wide=False) for n in (1000, 10_000, 10_000, 3000)]
    e_mixed, e_plain = parse_expr("f32 > 0.0 and s >= 'ab' or flag"), parse_expr("value2 > 10.0")

    def one_round():
        for rec in mixed:
            al = [[] for _ in range(rec.num_columns)]
            chq.filter_record(rec, al, e_mixed, ctx=ctx)
            d = chq.DeviceRecordBatch.from_host(rec, ctx)
            chq.filter_record(d, al, e_mixed, ctx=ctx).release()
            chq.compute_value(rec, al, parse_expr("i32 * small + 1"), ctx=ctx)
            d.release()
        al = [[] for _ in range(plain[0].num_columns)]
        chq.filter_records(plain, al, e_plain, ctx=ctx)
        chq.filter_records_coalesced(plain, al, e_plain, ctx=ctx)
        chq.filter_records(mixed[:1] * 3, [[] for _ in range(mixed[0].num_columns)], e_mixed, ctx=ctx)
        chq.filter_project_record(sel.selection, sel.projection, plain[1], al, ctx=ctx)
        chq.project_record(sel.projection, plain[1], al, ctx=ctx)

    for _ in range(20):
        one_round()
    gc.collect(); torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for _ in range(300):
        one_round()
    gc.collect(); torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert free0 - free1 < 64 << 20, f"HBM in use grew by {(free0 - free1) >> 20} MiB over 300 rounds"
    assert rss1 - rss0 < 256 << 10, f"peak RSS grew by {(rss1 - rss0) >> 10} MiB over 300 rounds"      # ru_maxrss is in KiB
    ctx.close()


@pytest.mark.parametrize("compression", ["NONE", "SNAPPY"])
def test_read_files_filter_materialize_dag_on_the_device(tmp_path, compression):
    """The reference's whole DAG for `select ... from read_files('data/*.parquet') where ...` (README.md:88-92) through the
    operator mirrors: [read_files] -> exchange -> [filter] -> exchange -> [materialize], every operator on the GPU -- the scan
    decodes row groups in HBM (range reads, only the columns the query uses), records travel as zero-copy device slices of
    max_rows_per_batch rows, the result files are encoded from HBM.  Checked per record id against the oracle."""
    import threading

    import numpy as np
    import pyarrow as pa

    from chapterhouseqe_amd.operators.exchange_operator import ExchangeOperator
    from chapterhouseqe_amd.operators.tasks import (FilterOperatorTask, FilterTaskBuilder, MaterializeFilesOperatorTask, MaterializeFilesTaskBuilder,
                                                     OperatorInstanceConfig, OperatorTaskRegistry, ReadFilesOperatorTask, ReadFilesTaskBuilder)
    rng = np.random.default_rng(8)
    root = tmp_path / "storage"
    (root / "data").mkdir(parents=True)
    tables = []
    for k, n in enumerate([25_000, 7, 12_345]):
        t = pa.table({"id": pa.array(np.arange(n, dtype=np.int32) + 100_000 * k),
                      "value1": pa.array(["%08x" % v for v in rng.integers(0, 2**32, n)]),
                      "value2": pa.array((rng.random(n) * 100).astype(np.float32), mask=rng.random(n) < 0.05),
                      "unused": pa.array(rng.integers(0, 2**62, n).astype(np.int64)), "unused2": pa.array(rng.random(n))})
        pq.write_table(t, root / "data" / f"part{k}.parquet", compression=compression, row_group_size=10_000)
        tables.append(t)
    (root / "data" / "notes.txt").write_text("not a parquet file")
    sql = "select id, value1, value2 * 2.0 as twice from read_files('data/*.parquet') where value2 > 10.0"
    sel = parse_select(sql)
    ex0 = ExchangeOperator("operator_p0_exchange", ["operator_p1_producer"])
    ex1 = ExchangeOperator("operator_p1_exchange", ["operator_p2_producer"])
    out_root = tmp_path / "results"
    reg = (OperatorTaskRegistry()
           .add_table_func_task_builder("read_files", ReadFilesTaskBuilder(str(root), columns=["id", "value1", "value2"]))
           .add_filter_task_builder(FilterTaskBuilder(group_size=16))
           .add_materialize_files_builder(MaterializeFilesTaskBuilder(str(out_root)), ["parquet"]))
    rtask = ReadFilesOperatorTask("data/*.parquet", alias=None, max_rows_per_batch=4_000)
    rrun = reg.find_task_builder(rtask).build(OperatorInstanceConfig(1, "operator_p0_producer", 7, rtask), [], ex0)
    assert rrun() is None
    ex0.producers_completed()
    reader = rrun.task
    assert len(reader.files_read) == 3 and reader.host_fallbacks == 0
    sizes = sum(os.path.getsize(root / "data" / f"part{k}.parquet") for k in range(3))
    assert reader.bytes_fetched < sizes          # the `unused` column's chunks were never read
    # records: every 10 000-row row group in slices of at most 4 000 rows, in file order
    want_records = []
    for t in tables:
        for g in range(0, t.num_rows, 10_000):
            grp = t.slice(g, min(10_000, t.num_rows - g)).select(["id", "value1", "value2"])
            for at in range(0, grp.num_rows, 4_000):
                want_records.append(grp.slice(at, min(4_000, grp.num_rows - at)).combine_chunks().to_batches()[0])
    assert reader.record_id == len(want_records)
    ftask = FilterOperatorTask(sel.selection)
    frun = reg.find_task_builder(ftask).build(OperatorInstanceConfig(2, "operator_p1_producer", 7, ftask), [ex0], ex1)
    err = [None]
    th = threading.Thread(target=lambda: err.__setitem__(0, frun()))
    th.start(); th.join()
    assert err[0] is None and frun.task.records_processed == len(want_records) and frun.task.group_calls >= 1
    ex1.producers_completed()
    mtask = MaterializeFilesOperatorTask("parquet", sel.projection)
    mrun = reg.find_task_builder(mtask).build(OperatorInstanceConfig(3, "operator_p2_producer", 7, mtask), [ex1], None)
    assert mrun() is None
    d = os.path.dirname(mrun.task.files_written[0])
    for rid, rec in enumerate(want_records):
        al = [[] for _ in range(rec.num_columns)]
        exp = O.project_record(sel.projection, O.filter_record(rec, al, sel.selection), al)
        got = pq.read_table(os.path.join(d, f"rec_{rid}.parquet")).to_batches()
        got = got[0] if got else exp.slice(0, 0)
        assert got.to_pydict() == exp.to_pydict(), rid


def test_device_batch_slices_are_views():
    import numpy as np
    import pyarrow as pa
    from .helpers import batches_identical
    c = chq.Context(0)
    rng = np.random.default_rng(3)
    n = 5000
    rec = pa.RecordBatch.from_arrays([pa.array(np.arange(n, dtype=np.int32)), pa.array(["s%d" % i for i in range(n)]),
                                      pa.array(rng.random(n), mask=rng.random(n) < 0.2), pa.array(rng.integers(0, 2, n).astype(bool))],
                                     names=["id", "s", "x", "b"])
    dev = chq.DeviceRecordBatch.from_host(rec, c)
    for off, ln in [(0, n), (1, 100), (63, 65), (4999, 1), (17, 0), (4000, None)]:
        sl = dev.slice(off, ln)
        want = rec.slice(off, ln) if ln is not None else rec.slice(off)
        assert batches_identical(sl.to_host(), want), (off, ln)
        e = parse_select("select * from t where x > 0.5 and id % 3 = 0").selection
        al = [[] for _ in range(4)]
        assert batches_identical(chq.filter_record(sl, al, e, ctx=c).to_host(), O.filter_record(want, al, e)), (off, ln)
    sl = dev.slice(10, 20)
    del dev   # the view keeps the parent's buffers alive
    assert sl.to_host().column(0).to_pylist() == list(range(10, 30))
    c.close()
