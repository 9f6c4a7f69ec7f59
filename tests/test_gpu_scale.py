"""GPU: BASELINE.json's full sizes, checked through size-independent properties and against torch's own
compaction (torch.masked_select) -- count, order preservation, a checksum of every output column and full
bit-equality of the compacted columns."""
import ctypes

import numpy as np
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr

pytestmark = pytest.mark.gpu


def _dtod(dst_tensor, src_addr, nbytes):
    import os
    import torch
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rc = hip.hipMemcpy(dst_tensor.data_ptr(), src_addr, nbytes, 3)   # hipMemcpyDeviceToDevice
    assert rc == 0


@pytest.mark.parametrize("n,thr", [(1_000_000_000, 10.0), (250_000_000, 90.0), (100_000_001, 50.0)])
def test_config2_full_size_against_torch(n, thr):
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0xC0FFEE)
    ids = torch.arange(n, dtype=torch.int32, device=dev)           # order witness
    v1 = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
    v2 = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
    ctx = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    rec = chq.DeviceRecordBatch.from_device_pointers(
        [("id", "i", ids.data_ptr()), ("value1", "f", v1.data_ptr()), ("value2", "f", v2.data_ptr())], n, ctx=ctx)
    out = chq.filter_record(rec, chq.get_record_table_aliases(None, rec), parse_expr(f"value2 > {thr}"), ctx=ctx)
    mask = v2 > thr
    m = int(mask.sum().item())
    assert out.num_rows == m                                         # count
    got = []
    for i, dt in enumerate([torch.int32, torch.float32, torch.float32]):
        t = torch.empty(m, dtype=dt, device=dev)
        _dtod(t, out.column_buffer_address(i, 1), m * 4)
        got.append(t)
    torch.cuda.synchronize()
    assert bool((got[0][1:] > got[0][:-1]).all().item())             # order preserved (ids strictly increasing)
    assert bool((got[2] > thr).all().item())                         # every survivor satisfies the predicate
    for src, g_ in zip([ids, v1, v2], got):                          # bit-equality with torch's compaction
        exp = torch.masked_select(src, mask)
        assert torch.equal(exp.view(torch.int32), g_.view(torch.int32))
        assert int(exp.view(torch.int32).to(torch.int64).sum().item()) == int(g_.view(torch.int32).to(torch.int64).sum().item())
        del exp
    st = ctx.last_stats()
    assert st["rows_in"] == n and st["rows_out"] == m and st["launches"] == (1 if n % 16384 == 0 else 2)   # + the tail-tile launch
    out.release()
    ctx.close()


def test_idempotence_and_empty_result():
    import torch
    n = 50_000_000
    dev = torch.device("cuda", 0)
    v = torch.rand(n, device=dev) * 100
    ctx = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    rec = chq.DeviceRecordBatch.from_device_pointers([("value2", "f", v.data_ptr())], n, ctx=ctx)
    al = [[]]
    e = parse_expr("value2 > 10.0")
    once = chq.filter_record(rec, al, e, ctx=ctx)
    twice = chq.filter_record(once, al, e, ctx=ctx)                  # filtering the survivors keeps them all
    assert twice.num_rows == once.num_rows
    a = torch.empty(once.num_rows, device=dev)
    b = torch.empty(once.num_rows, device=dev)
    _dtod(a, once.column_buffer_address(0, 1), once.num_rows * 4)
    _dtod(b, twice.column_buffer_address(0, 1), once.num_rows * 4)
    assert torch.equal(a, b)
    none = chq.filter_record(rec, al, parse_expr("value2 > 1000.0"), ctx=ctx)
    assert none.num_rows == 0
    ctx.close()


@pytest.mark.parametrize("rows_per_batch,nb", [(10_000, 100_000), (9_999, 50_000)])
def test_reference_sized_batches_full_size_group(rows_per_batch, nb):
    """config 2 cut into the reference's 10 000-row batches (physical_planner.rs:323): 10^5 batches = 1 B rows through ONE
    chq_filter_records_coalesced call -- per-batch counts and the joined output against torch"""
    import torch
    dev = torch.device("cuda", 0)
    n = rows_per_batch * nb
    g = torch.Generator(device=dev)
    g.manual_seed(0xBEEF)
    ids = torch.arange(n, dtype=torch.int32, device=dev)
    v1 = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
    v2 = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
    ctx = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    recs = [chq.DeviceRecordBatch.from_device_pointers(
        [("id", "i", ids.data_ptr() + 4 * b * rows_per_batch), ("value1", "f", v1.data_ptr() + 4 * b * rows_per_batch),
         ("value2", "f", v2.data_ptr() + 4 * b * rows_per_batch)], rows_per_batch, ctx=ctx) for b in range(nb)]
    grp = chq.RecordGroup(recs, ctx)
    out, rows = chq.filter_records_coalesced(grp, [[], [], []], parse_expr("value2 > 10.0"), ctx=ctx)
    st = ctx.last_stats()
    mask = v2 > 10.0
    exp_rows = mask.view(nb, rows_per_batch).sum(dim=1)
    assert torch.equal(torch.tensor(rows, dtype=torch.int64), exp_rows.cpu())          # every batch's count
    m = int(exp_rows.sum().item())
    assert out.num_rows == m and st["rows_in"] == n and st["launches"] == 1
    for i, src in enumerate([ids, v1, v2]):                                             # the joined output, bit for bit
        t = torch.empty(m, dtype=src.dtype, device=dev)
        _dtod(t, out.column_buffer_address(i, 1), m * 4)
        exp = torch.masked_select(src, mask)
        assert torch.equal(exp.view(torch.int32), t.view(torch.int32))
        del exp, t
    out.release()
    grp.release()
    ctx.close()


def test_reference_schema_group_full_size():
    """the reference's own schema in its own batch size, resident in HBM, at scale: 12 500 x 10 000-row batches (1 GB of
    string bytes: the most one joined Utf8 column can hold with int32 offsets) through ONE chq_filter_records_coalesced
    call -- per-batch counts, order, ids and the Float32 column bit for bit against torch, string bytes by checksum"""
    import torch
    dev = torch.device("cuda", 0)
    nb, rpb, L8 = 12_500, 10_000, 8
    n = nb * rpb
    g = torch.Generator(device=dev); g.manual_seed(0xFEED)
    ids = torch.arange(n, dtype=torch.int32, device=dev)
    chars = torch.randint(ord("a"), ord("z") + 1, (n * L8,), dtype=torch.uint8, device=dev, generator=g)
    offs = (torch.arange(rpb + 1, dtype=torch.int64, device=dev) * L8).to(torch.int32)      # batch-local offsets, shared
    v2 = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
    ctx = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    recs = [chq.DeviceRecordBatch.from_device_pointers(
        [("id", "i", ids.data_ptr() + 4 * b * rpb), ("value1", "u", offs.data_ptr(), chars.data_ptr() + L8 * b * rpb),
         ("value2", "f", v2.data_ptr() + 4 * b * rpb)], rpb, ctx=ctx) for b in range(nb)]
    grp = chq.RecordGroup(recs, ctx)
    out, rows = chq.filter_records_coalesced(grp, [[], [], []], parse_expr("value2 > 50.0 and id % 2 = 0"), ctx=ctx)
    mask = (v2 > 50.0) & (ids % 2 == 0)
    exp_rows = mask.view(nb, rpb).sum(dim=1)
    assert torch.equal(torch.tensor(rows, dtype=torch.int64), exp_rows.cpu())
    m = int(exp_rows.sum().item())
    assert out.num_rows == m and ctx.last_stats()["launches"] <= 8
    got_ids = torch.empty(m, dtype=torch.int32, device=dev); _dtod(got_ids, out.column_buffer_address(0, 1), m * 4)
    got_v2 = torch.empty(m, dtype=torch.float32, device=dev); _dtod(got_v2, out.column_buffer_address(2, 1), m * 4)
    assert torch.equal(got_ids, torch.masked_select(ids, mask))
    assert torch.equal(got_v2.view(torch.int32), torch.masked_select(v2, mask).view(torch.int32))
    got_offs = torch.empty(m + 1, dtype=torch.int32, device=dev); _dtod(got_offs, out.column_buffer_address(1, 1), (m + 1) * 4)
    assert torch.equal(got_offs.to(torch.int64), torch.arange(m + 1, dtype=torch.int64, device=dev) * L8)     # 8 bytes per kept row
    got_chars = torch.empty(m * L8, dtype=torch.uint8, device=dev); _dtod(got_chars, out.column_buffer_address(1, 2), m * L8)
    exp_chars = chars.view(n, L8)[mask]
    assert torch.equal(got_chars.view(m, L8), exp_chars)
    out.release(); grp.release(); ctx.close()


def test_config4_100m_rows_as_five_batches_in_one_group():
    """BASELINE config 4 at its full 100 M rows: id:Int32, value1:Utf8(100), value2:Float32.  One Arrow Utf8 array holds
    < 2 GiB of bytes (int32 offsets), so 100 M rows are five 20 M-row batches; ONE chq_filter_records call filters them
    (`id > n/2` on batch-local ids) -- counts, ids, and a checksum of the copied string bytes against torch"""
    import torch
    dev = torch.device("cuda", 0)
    nb, rpb, L = 5, 20_000_000, 100
    g = torch.Generator(device=dev); g.manual_seed(4)
    ids = torch.arange(rpb, dtype=torch.int32, device=dev)                                   # the same ids in every batch
    offs = (torch.arange(rpb + 1, dtype=torch.int64, device=dev) * L).to(torch.int32)
    ctx = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    chars, v2s, recs = [], [], []
    for b in range(nb):
        c = torch.randint(ord("a"), ord("z") + 1, (rpb * L,), dtype=torch.uint8, device=dev, generator=g)
        v = torch.empty(rpb, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
        chars.append(c); v2s.append(v)
        recs.append(chq.DeviceRecordBatch.from_device_pointers(
            [("id", "i", ids.data_ptr()), ("value1", "u", offs.data_ptr(), c.data_ptr()), ("value2", "f", v.data_ptr())], rpb, ctx=ctx))
    outs = chq.filter_records(recs, [[], [], []], parse_expr(f"id > {rpb // 2}"), ctx=ctx)
    assert len(outs) == nb
    keep = rpb - rpb // 2 - 1
    for b, o in enumerate(outs):
        assert o.num_rows == keep
        got_ids = torch.empty(keep, dtype=torch.int32, device=dev); _dtod(got_ids, o.column_buffer_address(0, 1), keep * 4)
        assert torch.equal(got_ids, ids[rpb // 2 + 1:])
        got_chars = torch.empty(keep * L, dtype=torch.uint8, device=dev); _dtod(got_chars, o.column_buffer_address(1, 2), keep * L)
        assert torch.equal(got_chars, chars[b][(rpb // 2 + 1) * L:])
        got_v = torch.empty(keep, dtype=torch.float32, device=dev); _dtod(got_v, o.column_buffer_address(2, 1), keep * 4)
        assert torch.equal(got_v, v2s[b][rpb // 2 + 1:])
        del got_ids, got_chars, got_v
        o.release()
    ctx.close()


def test_config5_one_rank_shard_at_full_size():
    """BASELINE config 5 (sample_queries/huge_simple.sql:3-4, 10 B rows over 8 GPUs): ONE rank's 1.25 B-row shard of
    id:Int32, value1:Utf8(8), value2:Float32 -- 10 GB of string bytes, so ten device batches under the int32 offset limit --
    through ONE chq_filter_records call (`id % 2 = 0`), exactly as bench.py --config 5 runs it on every GPU.  Every output
    batch is checked by size-independent properties: exactly the even ids in order, each with its own string and float."""
    import torch

    import bench
    dev = torch.device("cuda", 0)
    ctx = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    rows, batch_rows = 1_250_000_000, 125_000_000
    batches, keep = bench.build_config5_shard(chq, torch, dev, ctx, rows, batch_rows)
    assert len(batches) == 10 and sum(b.num_rows for b in batches) == rows
    grp = chq.RecordGroup(batches, ctx)
    outs = chq.filter_records(grp, [[], [], []], parse_expr(bench.C5_PREDICATE), ctx=ctx)
    st = ctx.last_stats()
    assert len(outs) == 10 and sum(o.num_rows for o in outs) == rows // 2
    assert st["rows_in"] == rows and st["rows_out"] == rows // 2
    # 20 B/row read (id, offsets by both phases, float) + the selected strings; 4 + 4 + 8 + 4 B per surviving row written
    assert st["bytes_read_alg"] >= rows * 16 and st["bytes_written_alg"] >= (rows // 2) * 20
    for b in range(len(outs)):
        bench.check_config5_outputs(torch, keep, outs, b)
    for o in outs:
        o.release()
    grp.release()
    ctx.close()


@pytest.mark.parametrize("extra,rows_per_gpu", [(["--rows", "120000001"], None), (["--scaling", "weak", "--rows", "30000000"], 30_000_000),
                                                (["--config", "5", "--rows", "60000000", "--batch-rows", "25000000"], 60_000_000)])
def test_bench_n_rank_path_rehearsed_on_one_gpu(extra, rows_per_gpu):
    """`bench.py --gpus 3` end to end on this box's single GPU (CHQ_BENCH_REHEARSE=1: all ranks on device 0, gloo between them):
    the code path the driver runs on the 8-GPU node -- self-launch through torch.distributed.run, the row plan (strong:
    120 000 001 rows split 40 000 001 / 40 000 000 / 40 000 000), barriers, MAX / SUM reductions, the weak run riding along,
    config 5 per-rank shards -- with the real kernels.  Timings of a shared card mean nothing and are not checked."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CHQ_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--validate-rows", "200000"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = lines[0]
    assert j["n_gpus"] == 3 and j["steps"] == 3 and j["value"] > 0 and len(j["config"]["per_gpu_rows_per_s"]) == 3
    if "--config" in extra:
        assert j["scaling"] == "weak" and j["config"]["rows_per_gpu"] == rows_per_gpu and j["config"]["rows_total"] == 3 * rows_per_gpu
        assert abs(j["config"]["selectivity"] - 0.5) < 1e-6
    elif "weak" in extra:
        assert j["scaling"] == "weak" and j["config"]["rows_per_gpu"] == rows_per_gpu and j["config"]["rows_total"] == 3 * rows_per_gpu
        assert "extra" not in j or "weak" not in j["extra"]
    else:
        assert j["scaling"] == "strong" and j["config"]["rows_total"] == 120_000_001 and j["config"]["rows_per_gpu"] == 40_000_001
        assert j["extra"]["weak"]["rows_per_gpu"] == 120_000_001 and j["extra"]["weak"]["value"] > 0
        assert 0.89 < j["config"]["rows_out_total"] / 120_000_001 < 0.91     # value2 > 10.0 keeps ~90 %
        assert j["roofline"]["traffic"] is None                              # (the PMC file was measured on 1e9 rows per GPU)
