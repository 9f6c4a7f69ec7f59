"""Arrow IPC stream encoding of record batches with the body in HBM (the reference's wire format
for batches that leave the process: messages/exchange.rs:145-197 writer, :247-276 reader).  The checker is pyarrow's own IPC
reader / writer -- a standard format with an independent implementation.

CPU tier: the metadata (flatbuffer) READER against streams pyarrow writes, through the GPU-free `chq_ipc_describe`.
GPU tier: the WRITER (pyarrow must read our streams back bit for bit) and full round trips, every column kind, nulls,
slices, empty batches, host and HBM bodies."""
import datetime
import decimal

import numpy as np
import pyarrow as pa
import pytest

import chapterhouseqe_amd as chq

from .helpers import batches_identical, explain_diff


def sample_batch(n, seed=0, nulls=True):
    rng = np.random.default_rng(seed)

    def m(p):
        return (rng.random(n) < p) if (nulls and n) else None

    words = np.array(["", "a", "ab", "héllo", "a much longer string value that spans more than sixty-four bytes of utf8 text ......"])
    cols = [
        ("i8", pa.array(rng.integers(-128, 128, n).astype(np.int8), mask=m(0.1))),
        ("u16", pa.array(rng.integers(0, 65536, n).astype(np.uint16))),
        ("i32", pa.array(rng.integers(-10**6, 10**6, n).astype(np.int32), mask=m(0.2))),
        ("u64", pa.array(rng.integers(0, 2**63, n).astype(np.uint64))),
        ("f32", pa.array((rng.random(n) * 100).astype(np.float32), mask=m(0.1))),
        ("f64", pa.array(rng.random(n))),
        ("flag", pa.array(rng.integers(0, 2, n).astype(bool), mask=m(0.3))),
        ("s", pa.array(words[rng.integers(0, len(words), n)] if n else np.array([], dtype=object), type=pa.utf8(), mask=m(0.15))),
        ("day", pa.array([datetime.date(2020, 1, 1) + datetime.timedelta(days=int(d)) for d in rng.integers(0, 3000, n)], type=pa.date32())),
        ("ts", pa.array(rng.integers(0, 10**15, n), type=pa.timestamp("us", tz="UTC"))),
        ("dec", pa.array([decimal.Decimal(int(v)) / 100 for v in rng.integers(-10**9, 10**9, n)], type=pa.decimal128(20, 2))),
    ]
    return pa.RecordBatch.from_arrays([c[1] for c in cols], names=[c[0] for c in cols])


def pyarrow_stream(rec: pa.RecordBatch) -> bytes:
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, rec.schema) as w:
        w.write_batch(rec)
    return sink.getvalue().to_pybytes()


FORMATS = {pa.int8(): "c", pa.uint16(): "S", pa.int32(): "i", pa.uint64(): "L", pa.float32(): "f", pa.float64(): "g",
           pa.bool_(): "b", pa.utf8(): "u", pa.date32(): "tdD", pa.timestamp("us", tz="UTC"): "tsu:UTC", pa.decimal128(20, 2): "d:20,2"}


@pytest.mark.parametrize("n", [0, 1, 7, 1000])
def test_metadata_reader_against_pyarrow_streams(n):
    """no GPU: schema (names, types, nullability), row count, null counts and the buffer table of a pyarrow-written stream"""
    rec = sample_batch(n, seed=n)
    stream = pyarrow_stream(rec)
    lines = chq.ipc_describe(stream).splitlines()
    head = lines[0].split()
    assert head[0] == "rows" and int(head[1]) == n
    body_len, body_at = int(head[3]), int(head[5])
    assert body_at + body_len + 8 == len(stream)                 # metadata, body, 8-byte end-of-stream marker
    fields = [l.split() for l in lines if l.startswith("field ")]
    assert [f[1] for f in fields] == rec.schema.names
    assert [f[2] for f in fields] == [FORMATS[t] for t in rec.schema.types]
    assert [f[3] for f in fields] == [f"nullable={int(fl.nullable)}" for fl in rec.schema]
    assert [int(f[4].split("=")[1]) for f in fields] == [c.null_count for c in rec.columns]
    buffers = [tuple(int(x) for x in l.split()[1:]) for l in lines if l.startswith("buffer ")]
    assert len(buffers) == sum(3 if t == pa.utf8() else 2 for t in rec.schema.types)
    assert all(off % 8 == 0 and off + ln <= body_len for off, ln in buffers)


def test_metadata_reader_rejects_what_is_out_of_scope():
    with pytest.raises(chq.ChqError) as ei:
        chq.ipc_describe(b"")
    assert ei.value.code == 22
    rec = pa.RecordBatch.from_arrays([pa.array(["a", "b", "a"]).dictionary_encode()], names=["d"])
    with pytest.raises(chq.ChqError) as ei:
        chq.ipc_describe(pyarrow_stream(rec))
    assert ei.value.code == 30                                     # dictionaries: NotSupported
    nested = pa.RecordBatch.from_arrays([pa.array([[1, 2], [3]])], names=["l"])
    with pytest.raises(chq.ChqError) as ei:
        chq.ipc_describe(pyarrow_stream(nested))
    assert ei.value.code == 30
    stream = pyarrow_stream(sample_batch(10))
    with pytest.raises(chq.ChqError):
        chq.ipc_describe(stream[:40])                              # truncated


# ------------------------------------------------------------------------------------------------------------ GPU tier
@pytest.fixture(scope="module")
def ctx():
    c = chq.Context(0)
    yield c
    c.close()


def read_back(stream: bytes) -> pa.RecordBatch:
    t = pa.ipc.open_stream(stream).read_all()
    batches = t.to_batches()
    return batches[0] if batches else pa.RecordBatch.from_pylist([], schema=t.schema)


@pytest.mark.gpu
@pytest.mark.parametrize("resident", ["host", "hbm"])
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 70_001])
def test_pyarrow_reads_what_the_gpu_writes(ctx, n, resident):
    rec = sample_batch(n, seed=100 + n)
    src = chq.DeviceRecordBatch.from_host(rec, ctx) if resident == "hbm" else rec
    enc = chq.record_to_ipc(src, ctx=ctx)
    stream = enc.to_bytes()
    assert enc.body_len % 64 == 0 and len(enc.header) % 8 == 0
    got = read_back(stream)
    assert got.schema.equals(rec.schema, check_metadata=False)
    assert batches_identical(got, rec), explain_diff(got, rec)
    assert chq.ipc_describe(stream).splitlines()[0].startswith(f"rows {n} ")       # and our own reader agrees
    enc.release()


@pytest.mark.gpu
def test_sliced_batches_are_rebased(ctx):
    """Arrow slices: value buffers start mid-buffer, bitmaps at a bit offset, Utf8 offsets at a non-zero value"""
    rec = sample_batch(5000, seed=7)
    for start, length in [(1, 100), (7, 2049), (63, 64), (64, 4000), (4999, 1), (13, 0)]:
        sl = rec.slice(start, length)
        for src in (sl, chq.DeviceRecordBatch.from_host(sl, ctx)):
            got = read_back(chq.record_to_ipc(src, ctx=ctx).to_bytes())
            assert batches_identical(got, sl), (start, length, explain_diff(got, sl))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 5, 1000, 70_001])
def test_the_gpu_reads_what_pyarrow_writes(ctx, n):
    rec = sample_batch(n, seed=200 + n)
    stream = pyarrow_stream(rec)
    dev = chq.record_from_ipc(stream, ctx=ctx, device_result=True)
    assert isinstance(dev, chq.DeviceRecordBatch) and dev.num_rows == n
    back = dev.to_host()
    assert batches_identical(back, rec), explain_diff(back, rec)
    host = chq.record_from_ipc(stream, ctx=ctx, device_result=False)
    assert batches_identical(host, rec), explain_diff(host, rec)


@pytest.mark.gpu
def test_body_stays_in_hbm_between_two_contexts(ctx):
    """the exchange data plane: metadata over the control channel (host bytes), the body as ONE device buffer (what RCCL
    or a peer copy moves); the receiver wraps it and filters it without the rows ever touching the host"""
    from chapterhouseqe_amd.sqlparse import parse_expr
    from oracle import oracle as O
    rec = sample_batch(30_000, seed=3).select(["i8", "u16", "i32", "u64", "f32", "f64", "flag", "s"])   # (types the oracle knows)
    other = chq.Context(0)
    enc = chq.record_to_ipc(chq.DeviceRecordBatch.from_host(rec, ctx), ctx=ctx, body_on_device=True)
    assert enc.body_on_device and enc.body_address
    with pytest.raises(ValueError):
        enc.to_bytes()
    dev = chq.record_from_ipc(enc.header, ctx=other, device_result=True, body_address=enc.body_address, body_len=enc.body_len, body_on_device=True)
    enc.release()                                                  # the receiver owns its copy of the body
    al = [[] for _ in range(rec.num_columns)]
    e = parse_expr("i32 > 0 and s <> 'ab'")
    got = chq.filter_record(dev, al, e, ctx=other).to_host()
    assert batches_identical(got, O.filter_record(rec, al, e))
    other.close()


@pytest.mark.gpu
def test_filter_output_round_trips_through_ipc(ctx):
    """what the filter task would put on the wire: the compacted batch (fresh buffers, no slices)"""
    from chapterhouseqe_amd.sqlparse import parse_expr
    rec = sample_batch(50_000, seed=11)
    al = [[] for _ in range(rec.num_columns)]
    out = chq.filter_record(chq.DeviceRecordBatch.from_host(rec, ctx), al, parse_expr("f32 > 10.0 or flag"), ctx=ctx)
    got = read_back(chq.record_to_ipc(out, ctx=ctx).to_bytes())
    assert batches_identical(got, out.to_host())


@pytest.mark.gpu
def test_malformed_streams_fail_on_the_host(ctx):
    stream = bytearray(pyarrow_stream(sample_batch(100, seed=5)))
    with pytest.raises(chq.ChqError) as ei:
        chq.record_from_ipc(bytes(stream[: len(stream) // 2]), ctx=ctx)     # body cut short
    assert ei.value.code == 22
    with pytest.raises(chq.ChqError):
        chq.record_from_ipc(b"\xff\xff\xff\xff\x10\x00\x00\x00" + b"\x00" * 16, ctx=ctx)
