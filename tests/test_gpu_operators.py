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
