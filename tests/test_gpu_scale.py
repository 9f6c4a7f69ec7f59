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
