"""GPU: Parquet pages decoded by the kernels of csrc/parquet.hip against pyarrow's reader on the
same bytes -- bit-exact values, validity, null counts, offsets and string bytes, over encodings (PLAIN, RLE_DICTIONARY, the
writers' mid-chunk fallback from one to the other), data page versions, page sizes, nulls, row groups and edge cases."""
import io

import numpy as np
import pyarrow as pa
import pyarrow.compute  # noqa: F401
import pyarrow.parquet as pq
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr
from oracle import oracle as O
from tests.parquet_cases import sample_table, write_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = chq.Context(0)
    yield c
    c.close()


def check(raw: bytes, ctx, device_result=True):
    exp = pq.ParquetFile(io.BytesIO(raw))
    f = chq.ParquetFile(raw)
    assert f.num_row_groups == exp.metadata.num_row_groups
    for g in range(f.num_row_groups):
        want = exp.read_row_group(g).combine_chunks()
        got = f.read_row_group(g, ctx=ctx, device_result=device_result)
        if device_result:
            got = got.to_host()
        assert got.num_rows == want.num_rows and got.schema.names == want.schema.names
        for i, name in enumerate(want.schema.names):
            w = want.column(i).chunk(0) if want.num_rows else pa.array([], type=want.schema.field(i).type)
            x = got.column(i)
            assert x.type == w.type, name
            assert x.null_count == w.null_count, name
            assert x.equals(w), f"{name}: row group {g}"
    f.close()


@pytest.mark.parametrize("codec", ["none", "snappy"])
@pytest.mark.parametrize("kw", [
    dict(), dict(use_dictionary=False), dict(data_page_version="2.0"), dict(use_dictionary=False, data_page_version="2.0"),
    dict(data_page_size=300), dict(data_page_size=300, use_dictionary=False), dict(row_group_size=1500, data_page_size=2000),
    dict(dictionary_pagesize_limit=2048, data_page_size=1024), dict(write_statistics=False),
])
@pytest.mark.parametrize("nulls", [False, True])
@pytest.mark.parametrize("strings", ["mixed", "unique", "few"])
def test_row_groups_match_pyarrow(ctx, kw, nulls, strings, codec):
    """the whole matrix once as the reference's writers leave their files (uncompressed) and once as everybody else's do
    (snappy: pyarrow's default): the pages of a compressed chunk are inflated on the GPU (csrc/parquet_codec.hip)"""
    for n in [1, 63, 64, 65, 1000, 4097, 20_000]:
        check(write_bytes(sample_table(n, seed=n, nulls=nulls, strings=strings), compression=codec, **kw), ctx)


def test_edge_cases(ctx):
    # no rows at all; all-null columns; empty strings only; a string far longer than the walker's LDS window; one huge page
    check(write_bytes(sample_table(0)), ctx)
    n = 3000
    t = pa.table({"allnull": pa.array([None] * n, type=pa.int32()), "nullstr": pa.array([None] * n, type=pa.utf8()),
                  "empty": pa.array([""] * n), "bools": pa.array([True, False, None] * (n // 3))})
    for kw in [dict(), dict(use_dictionary=False), dict(data_page_version="2.0", data_page_size=100)]:
        check(write_bytes(t, **kw), ctx)
    big = ["x" * 40_000, "", "y" * 17_000, "short", "z" * 70_000] * 7
    for kw in [dict(), dict(use_dictionary=False), dict(use_dictionary=False, data_page_size=1 << 22)]:
        check(write_bytes(pa.table({"s": pa.array(big), "k": pa.array(np.arange(len(big), dtype=np.int64))}), **kw), ctx)
    rng = np.random.default_rng(5)
    one_page = pa.table({"v": pa.array(rng.integers(0, 1000, 300_000).astype(np.int32)), "s": pa.array(["%05d" % v for v in rng.integers(0, 99999, 300_000)])})
    for kw in [dict(data_page_size=1 << 26), dict(data_page_size=1 << 26, use_dictionary=False)]:
        check(write_bytes(one_page, **kw), ctx)


def test_snappy_pages_of_every_shape(ctx):
    """what the inflate kernel has to get right beyond the matrix above: pages far larger than its 64 KiB ring of output
    history, incompressible data (one literal per 64 KiB block), runs (copies with offset 1 that overlap themselves), pages
    whose first element is a literal longer than the ring, V2 pages with uncompressed level sections, all-null columns"""
    rng = np.random.default_rng(77)
    n = 400_000
    t = pa.table({
        "noise": pa.array(rng.integers(-2**62, 2**62, n), type=pa.int64()),                      # incompressible
        "zeros": pa.array(np.zeros(n, dtype=np.int32)),                                          # offset-1 copies
        "ramp": pa.array((np.arange(n) // 1000).astype(np.int32)),                               # long matches
        "text": pa.array(["the quick brown fox %d jumps over the lazy dog" % (v % 97) for v in range(n)]),
        "opt": pa.array(rng.integers(0, 5, n).astype(np.float64), mask=rng.random(n) < 0.4),
        "nulls": pa.array([None] * n, type=pa.int32()),
    })
    for kw in [dict(), dict(use_dictionary=False), dict(use_dictionary=False, data_page_size=1 << 24), dict(data_page_version="2.0"),
               dict(data_page_version="2.0", use_dictionary=False, data_page_size=1 << 22), dict(row_group_size=150_000, data_page_size=4096)]:
        check(write_bytes(t, compression="snappy", **kw), ctx)
    check(write_bytes(sample_table(0), compression="snappy"), ctx)
    check(write_bytes(sample_table(3000, seed=3, nulls=True), compression="snappy"), ctx, device_result=False)


def test_column_pruning_uploads_only_the_selected_chunks(ctx):
    """DEV_NOTES.md:123 (the reference's own TODO): a query that reads one column must not pay for the others -- a pruned
    read of 1 of 3 same-width columns uploads about a third of the bytes, columns come back in the order asked for"""
    n = 200_000
    rng = np.random.default_rng(13)
    t = pa.table({"a": pa.array(rng.integers(0, 2**31, n).astype(np.int32)), "b": pa.array(rng.random(n).astype(np.float32)),
                  "c": pa.array(rng.integers(0, 2**31, n).astype(np.int32))})
    for codec in ("none", "snappy"):
        raw = write_bytes(t, compression=codec, use_dictionary=False, row_group_size=50_000)
        f = chq.ParquetFile(raw)
        assert f.column_names == ["a", "b", "c"] and f.num_columns == 3
        full = f.read_row_groups(ctx=ctx)
        up_all = ctx.last_stats()["bytes_read_alg"]
        assert ctx.last_stats()["rows_in"] == n and ctx.last_stats()["bytes_written_alg"] == 12 * n
        one = f.read_row_groups(ctx=ctx, columns=["b"])
        up_one = ctx.last_stats()["bytes_read_alg"]
        assert 0.30 * up_all < up_one < 0.37 * up_all, (up_one, up_all)
        assert ctx.last_stats()["bytes_written_alg"] == 4 * n
        two = f.read_row_groups(ctx=ctx, columns=[2, 0])
        want = pq.ParquetFile(io.BytesIO(raw))
        for g in range(f.num_row_groups):
            w = want.read_row_group(g).to_batches()[0]
            assert full[g].to_host().equals(w)
            assert one[g].to_host().equals(w.select(["b"]))
            assert two[g].to_host().equals(w.select(["c", "a"]))
        with pytest.raises(chq.ChqError):
            f.read_row_groups(ctx=ctx, columns=[3])
        assert [b.num_columns for b in f.read_row_groups(ctx=ctx, columns=[])] == [0] * f.num_row_groups
        f.close()


def test_range_reader_fetches_the_footer_and_only_the_chunks_it_decodes(ctx):
    """the reference reads through opendal ranges (read_files_task.rs:233-250): the library asks the caller's reader for the
    file's tail at open and then for exactly the column chunks a call decodes -- never the whole file"""
    t = sample_table(60_000, seed=21, nulls=True)
    for codec in ("none", "snappy"):
        raw = write_bytes(t, compression=codec, row_group_size=20_000)
        f = chq.ParquetFile(None, reader=lambda off, ln: raw[off:off + ln], size=len(raw))
        assert len(f.reads) <= 2 and sum(l for _, l in f.reads) <= 70_000 and f.reads[0][0] + f.reads[0][1] == len(raw)
        assert f.num_row_groups == 3 and f.column_names == t.schema.names
        f.reads.clear()
        got = f.read_row_groups(1, 2, ctx=ctx, columns=["value1", "id"])
        md = pq.ParquetFile(io.BytesIO(raw)).metadata
        want_reads = []
        for g in (1, 2):
            for name in ("value1", "id"):
                cm = md.row_group(g).column(t.schema.names.index(name))
                first = cm.dictionary_page_offset if cm.dictionary_page_offset else cm.data_page_offset
                want_reads.append((first, cm.total_compressed_size))
        assert sorted(f.reads) == sorted(want_reads)                        # one read per selected chunk, nothing else
        want = pq.ParquetFile(io.BytesIO(raw))
        for k, g in enumerate((1, 2)):
            assert got[k].to_host().equals(want.read_row_group(g).to_batches()[0].select(["value1", "id"]))
        whole = f.read_row_groups(ctx=ctx)
        assert [b.to_host() for b in whole] == [want.read_row_group(g).to_batches()[0] for g in range(3)]
        f.close()
    # a reader that fails, and one that returns short reads: a status, not a crash
    raw = write_bytes(t, row_group_size=20_000)
    calls = []

    def flaky(off, ln):
        calls.append(off)
        if len(calls) > 1:
            raise IOError("storage went away")
        return raw[off:off + ln]
    f = chq.ParquetFile(None, reader=flaky, size=len(raw))
    with pytest.raises(chq.ChqError) as e:
        f.read_row_groups(ctx=ctx)
    assert e.value.code == 22 and "range reader" in str(e.value)
    with pytest.raises(chq.ChqError):
        chq.ParquetFile(None, reader=lambda off, ln: raw[off:off + ln - 1], size=len(raw))


def snappy_elements(raw: bytes):
    """(kind, length, offset) of every element of a raw snappy stream (format_description.txt) -- used to check that the data
    below really makes the compressor emit the elements the test is about"""
    pos, sh, n = 0, 0, 0
    while True:
        b = raw[pos]; pos += 1
        n |= (b & 0x7f) << sh; sh += 7
        if not b & 0x80:
            break
    out = []
    while pos < len(raw):
        tag = raw[pos]; kind = tag & 3
        if kind == 0:
            ln = tag >> 2; hdr = 1
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(raw[pos + 1:pos + 1 + nb], "little"); hdr = 1 + nb
            ln += 1
            out.append((0, ln, 0)); pos += hdr + ln
        elif kind == 1:
            out.append((1, ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | raw[pos + 1])); pos += 2
        elif kind == 2:
            out.append((2, (tag >> 2) + 1, int.from_bytes(raw[pos + 1:pos + 3], "little"))); pos += 3
        else:
            out.append((3, (tag >> 2) + 1, int.from_bytes(raw[pos + 1:pos + 5], "little"))); pos += 5
    return n, out


def test_snappy_elements_the_batched_parser_must_order(ctx):
    """the inflate kernel moves the elements of 64 input bytes together (csrc/parquet_codec.hip): copies that read what an
    earlier copy of the same batch wrote (chains of period-p runs), copies whose offset is close to the 64 KiB of history the
    ring holds (moved before the batch's own writes overrun their source), offsets beyond the ring (read back from HBM),
    literals too long for a batch, and a random mix of all of them.  The page is a required Int64 column's PLAIN values, so
    the stream the compressor sees is exactly these bytes"""
    rng = np.random.default_rng(2024)
    parts = []
    def rnd(k):
        return rng.integers(0, 256, k, dtype=np.uint8).tobytes()
    # one 64 KiB block per far offset: a key, a run (which leaves the compressor's match table alone), the key again
    for gap in (65536 - 48, 65440, 65473, 65400, 63000, 61500, 61000, 65487):
        key = rnd(48)
        block = key + bytes(gap - 48) + key
        parts.append(block + rnd(65536 - len(block)) if len(block) < 65536 else block[:65536])
    # chains: runs of every short period (each copy reads the bytes the previous one wrote), cut by literals of odd lengths
    for period in (1, 2, 3, 5, 7, 8, 11, 13, 24, 63, 64, 65, 100):
        unit = rnd(period)
        parts.append(unit * (1 + 3000 // period) + rnd(int(rng.integers(1, 9))))
    # a random mix: short literals, repeats of earlier bytes at any distance, runs, now and then a long literal
    mix = bytearray(rnd(200))
    while len(mix) < 3_000_000:
        c = rng.integers(0, 10)
        if c < 4:
            mix += rnd(int(rng.integers(1, 12)))
        elif c < 8:
            back = int(rng.integers(1, min(len(mix), 70000)))
            k = int(rng.integers(4, 80))
            at = len(mix) - back
            mix += mix[at:at + k]
        elif c < 9:
            unit = rnd(int(rng.integers(1, 12)))
            mix += unit * int(rng.integers(2, 40))
        else:
            mix += rnd(int(rng.integers(60, 400)))
    parts.append(bytes(mix))
    stream = b"".join(parts)
    stream += bytes(-len(stream) % 8)
    n_out, elements = snappy_elements(pa.Codec("snappy").compress(stream, asbytes=True))
    assert n_out == len(stream)
    offs = np.array([e[2] for e in elements if e[0]]); lens = np.array([e[1] for e in elements if e[0]])
    assert (offs > 65472).any() and ((offs > 61400) & (offs <= 65472)).sum() >= 3          # beyond the ring / moved first
    assert (offs < lens).sum() > 100 and any(e[0] == 0 and e[1] > 64 for e in elements)   # self-overlapping copies, long literals
    assert len(elements) > 40_000
    values = np.frombuffer(stream, dtype=np.int64)
    t = pa.table({"v": pa.array(values)}, schema=pa.schema([pa.field("v", pa.int64(), nullable=False)]))
    files = [write_bytes(t, compression="snappy", use_dictionary=False, **kw)
             for kw in [dict(data_page_size=1 << 26), dict(data_page_size=1 << 20), dict(data_page_size=70_000), dict(data_page_version="2.0", data_page_size=1 << 26)]]
    for raw in files:
        check(raw, ctx)
    # pages of three and more 64 KiB blocks were inflated one wave per block (option snappy_blocks, default 1); the same files
    # with one wave per page (0), and with the blocks giving up so that the FINISH job redoes the page (2)
    for mode in (0, 2, 3):                    # (3: large pages walked by one wave instead of one per segment)
        c = chq.Context(0)
        c.set_option("snappy_blocks", mode)
        for raw in files[:2]:
            check(raw, c)
        opt = pa.table({"v": pa.array(values[:300_000], mask=rng.random(300_000) < 0.3)})    # V1 optional: the descriptor patch
        check(write_bytes(opt, compression="snappy", use_dictionary=False, data_page_size=1 << 22), c)
        c.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_snappy_random_element_mixes(ctx, seed):
    """seeded streams of literals, repeats at any distance up to beyond the ring, runs of short periods and incompressible
    stretches, in Int64 / Int32 / Float64 / Utf8 columns with and without nulls, page sizes from a few KB to one page"""
    rng = np.random.default_rng(seed)
    def rnd(k):
        return rng.integers(0, 256, k, dtype=np.uint8).tobytes()
    mix = bytearray(rnd(64))
    short = int(rng.integers(3, 9))               # how literal-heavy this stream is
    while len(mix) < 700_000:
        c = rng.integers(0, 10)
        if c < short:
            mix += rnd(int(rng.integers(1, 1 << int(rng.integers(1, 8)))))
        elif c < 9:
            back = int(rng.integers(1, min(len(mix), 66000)))
            k = int(rng.integers(4, 1 << int(rng.integers(3, 9))))
            at = len(mix) - back
            for _ in range(k // max(back, 1) + 1):      # (a repeat longer than its distance overlaps itself)
                mix += mix[at:at + min(k, back)]
        else:
            mix += rnd(int(rng.integers(1, 12))) * int(rng.integers(2, 200))
    stream = bytes(mix[:len(mix) // 8 * 8])
    i64 = np.frombuffer(stream, dtype=np.int64)
    n = len(i64)
    text = (np.frombuffer(stream, dtype=np.uint8) % 26 + 97).astype(np.uint8).tobytes() * 2      # 16 bytes per row: enough
    lens = rng.integers(0, 11, n)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    t = pa.table({
        "a": pa.array(i64),
        "b": pa.array(np.frombuffer(stream, dtype=np.int32)[:n], mask=rng.random(n) < 0.2),
        "c": pa.array(np.frombuffer(stream, dtype=np.float64).view(np.int64)[:n] % 1000 / 8.0),
        "s": pa.Array.from_buffers(pa.utf8(), n, [None, pa.py_buffer(offs.tobytes()), pa.py_buffer(text[:int(offs[-1])])]),
    })
    page = int(rng.choice([4096, 70_000, 1 << 18, 1 << 20, 1 << 26]))
    kw = dict(use_dictionary=bool(rng.integers(0, 2)), data_page_size=page, data_page_version=str(rng.choice(["1.0", "2.0"])),
              row_group_size=int(rng.choice([n, n // 3 + 1])))
    raw = write_bytes(t, compression="snappy", **kw)
    check(raw, ctx)
    c = chq.Context(0)
    c.set_option("snappy_blocks", int(seed % 4))
    check(raw, c)
    c.close()


def test_damaged_snappy_pages_are_reported(ctx):
    """compressed bytes overwritten: offsets beyond the output so far, literals running past the page, a wrong uncompressed
    length -- the inflate kernel bounds every element and the call names the column"""
    n = 50_000
    rng = np.random.default_rng(5)
    t = pa.table({"k": pa.array((np.arange(n) // 7).astype(np.int32)), "s": pa.array(["row %06d" % (v % 500) for v in range(n)])})
    raw = write_bytes(t, compression="snappy", use_dictionary=False, data_page_size=8192)
    md = pq.ParquetFile(io.BytesIO(raw)).metadata
    reported = 0
    for col in range(2):
        cm = md.row_group(0).column(col)
        for frac in (0.05, 0.3, 0.6, 0.9):
            bad = bytearray(raw)
            at = cm.data_page_offset + int(cm.total_compressed_size * frac)
            bad[at:at + 48] = bytes(rng.integers(0, 256, 48, dtype=np.uint8))
            try:
                f = chq.ParquetFile(bytes(bad))
                got = f.read_row_group(0, ctx=ctx)
                assert got.num_rows == n       # (damage inside a literal decodes to different values: still a valid stream)
            except chq.ChqError as e:
                assert e.code in (22, 30), str(e)
                reported += 1
    assert reported >= 3


def test_damaged_snappy_pages_of_several_blocks(ctx):
    """the same for pages that are inflated block by block: damage in the element headers is found by the index walk (no
    block job runs), damage in an offset by a block job (the page is redone whole and judged there), damage in literal
    bytes decodes; a damaged page never keeps a later read from working"""
    n = 300_000
    rng = np.random.default_rng(6)
    t = pa.table({"k": pa.array((np.arange(n) // 3).astype(np.int64)), "s": pa.array(["row %06d of many" % (v % 5000) for v in range(n)])})
    reported = 0
    for page_size in (1 << 20, 1 << 26):      # (1 << 26: one page per chunk -- the strings' is walked in segments)
        raw = write_bytes(t, compression="snappy", use_dictionary=False, data_page_size=page_size)
        md = pq.ParquetFile(io.BytesIO(raw)).metadata
        if page_size == 1 << 26:
            assert md.row_group(0).column(1).total_compressed_size > (512 << 10)
        for col in range(2):
            cm = md.row_group(0).column(col)
            assert cm.total_uncompressed_size > 6 * 65536
            for frac in (0.01, 0.2, 0.45, 0.7, 0.97):
                for width in (1, 3, 64):
                    bad = bytearray(raw)
                    at = cm.data_page_offset + 64 + int((cm.total_compressed_size - 200) * frac)
                    bad[at:at + width] = bytes(rng.integers(0, 256, width, dtype=np.uint8))
                    try:
                        f = chq.ParquetFile(bytes(bad))
                        got = f.read_row_group(0, ctx=ctx)
                        assert got.num_rows == n
                        got.release()
                    except chq.ChqError as e:
                        assert e.code in (22, 30), str(e)
                        reported += 1
        check(raw, ctx)
    assert reported >= 10


def test_host_result_and_required_columns(ctx):
    n = 5000
    rng = np.random.default_rng(9)
    schema = pa.schema([pa.field("id", pa.int32(), nullable=False), pa.field("value1", pa.utf8(), nullable=False),
                        pa.field("value2", pa.float32(), nullable=False)])
    t = pa.table([pa.array(np.arange(n, dtype=np.int32)), pa.array(["%08x" % v for v in rng.integers(0, 2**32, n)]),
                  pa.array((rng.random(n) * 100).astype(np.float32))], schema=schema)
    for kw in [dict(), dict(use_dictionary=False), dict(data_page_version="2.0")]:
        check(write_bytes(t, **kw), ctx, device_result=False)
        check(write_bytes(t, **kw), ctx, device_result=True)


def test_scan_feeds_the_filter_kernels_without_leaving_hbm(ctx):
    """read_files -> filter as the reference's DAG runs them (read_files_task.rs:233-282 -> filter_task.rs:99), both on the
    device: the decoded row group goes straight into filter_record; checked against the oracle on pyarrow's decode"""
    n = 40_000
    rng = np.random.default_rng(11)
    t = pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "value1": pa.array(["%08x" % v for v in rng.integers(0, 2**32, n)]),
                  "value2": pa.array((rng.random(n) * 100).astype(np.float32))})
    raw = write_bytes(t, dictionary_pagesize_limit=16 * 1024, data_page_size=8 * 1024, row_group_size=25_000)
    e = parse_expr("value2 > 10.0 and id % 3 = 0")
    got_rows = 0
    for g, dev in enumerate(chq.scan_parquet(raw, ctx=ctx)):
        al = chq.get_record_table_aliases(None, dev)
        out = chq.filter_record(dev, al, e, ctx=ctx).to_host()
        host = pq.ParquetFile(io.BytesIO(raw)).read_row_group(g).to_batches()[0]
        exp = O.filter_record(host, al, e)
        assert out.equals(exp)
        got_rows += out.num_rows
    assert got_rows > 0


def test_unsupported_features_say_so(ctx):
    t = sample_table(100, seed=2)
    for codec, name in (("zstd", "ZSTD"), ("gzip", "GZIP"), ("lz4", "LZ4")):
        f = chq.ParquetFile(write_bytes(t, compression=codec))
        with pytest.raises(chq.ChqError) as e:
            f.read_row_group(0, ctx=ctx)
        assert e.value.code == 30 and "codec" in str(e.value) and name in str(e.value)
    f = chq.ParquetFile(write_bytes(pa.table({"d": pa.array([1, 2, 3], type=pa.date32())})))
    with pytest.raises(chq.ChqError) as e:
        f.read_row_group(0, ctx=ctx)
    assert e.value.code == 30
    f = chq.ParquetFile(write_bytes(pa.table({"v": pa.array(np.arange(1000, dtype=np.int64))}), use_dictionary=False,
                                    column_encoding={"v": "DELTA_BINARY_PACKED"}))
    with pytest.raises(chq.ChqError) as e:
        f.read_row_group(0, ctx=ctx)
    assert e.value.code == 30 and "encoding" in str(e.value)
    with pytest.raises(chq.ChqError):
        chq.ParquetFile(write_bytes(t)).read_row_group(5, ctx=ctx)


# ---- f-4: pages encoded on the GPU ---------------------------------------------------------------------------------------
def write_table_cases():
    rng = np.random.default_rng(21)
    for n in [0, 1, 7, 8, 9, 63, 64, 65, 1000, 4095, 4096, 4097, 50_000]:
        for nulls in (False, True):
            yield sample_table(n, seed=n + 100, nulls=nulls, strings="mixed")
    n = 10_000   # the reference's batch size and schema, non-nullable fields
    schema = pa.schema([pa.field("id", pa.int32(), nullable=False), pa.field("value1", pa.utf8(), nullable=False),
                        pa.field("value2", pa.float32(), nullable=False)])
    yield pa.table([pa.array(np.arange(n, dtype=np.int32)), pa.array(["%08x" % v for v in rng.integers(0, 2**32, n)]),
                    pa.array((rng.random(n) * 100).astype(np.float32))], schema=schema)


def test_written_files_are_read_back_by_pyarrow_and_by_the_scan(ctx):
    for t in write_table_cases():
        rec = t.to_batches()[0] if t.num_rows else pa.RecordBatch.from_arrays([pa.array([], type=f.type) for f in t.schema], schema=t.schema)
        for source in (rec, chq.DeviceRecordBatch.from_host(rec, ctx=ctx)):
            raw = chq.record_to_parquet(source, ctx=ctx)
            back = pq.read_table(io.BytesIO(raw))
            assert back.schema.names == rec.schema.names and back.num_rows == rec.num_rows
            md = pq.ParquetFile(io.BytesIO(raw)).metadata
            assert md.num_row_groups == 1 and md.row_group(0).num_rows == rec.num_rows
            for i, f in enumerate(rec.schema):
                got = back.column(i).combine_chunks()
                assert got.type == f.type and got.null_count == rec.column(i).null_count, f.name
                assert got.equals(rec.column(i)), f.name
                assert back.schema.field(i).nullable == f.nullable
            # ... and by this library's own scan (f-3 reads what f-4 writes)
            mine = chq.ParquetFile(raw).read_row_group(0, ctx=ctx).to_host()
            assert mine.equals(rec)


def test_written_chunks_carry_the_statistics_a_cpu_writer_records(ctx):
    """min / max / null_count of every column chunk equal what pyarrow's writer records for the same table (floats: NaNs
    take no part, a zero minimum is -0.0 and a zero maximum +0.0; strings: unsigned byte order)"""
    rng = np.random.default_rng(5)
    n = 20_000
    f32 = (rng.standard_normal(n) * 50).astype(np.float32)
    f32[::97] = np.nan; f32[5] = 0.0; f32[6] = -0.0
    zeros = np.zeros(n, dtype=np.float64); zeros[::2] = -0.0
    words = np.array(["", "a", "ab", "zeta", "Zulu", "\u00e9t\u00e9", "\U0001F600", "a" * 300])
    cases = [pa.table({
        "i32": pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32), mask=rng.random(n) < 0.1),
        "i64": pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64)),
        "f32": pa.array(f32, mask=rng.random(n) < 0.05),
        "f64": pa.array(rng.standard_normal(n) * 1e300),
        "z": pa.array(zeros),
        "b": pa.array(rng.integers(0, 2, n).astype(bool), mask=rng.random(n) < 0.3),
        "t": pa.array(np.ones(n, dtype=bool)),
        "s": pa.array(words[rng.integers(0, len(words), n)], type=pa.utf8(), mask=rng.random(n) < 0.2),
        "nulls": pa.array([None] * n, type=pa.int32()),
        "nans": pa.array(np.full(n, np.nan, dtype=np.float32)),
    }), sample_table(4097, seed=3, nulls=True, strings="mixed"), sample_table(1, seed=4, nulls=False)]
    for t in cases:
        rec = t.to_batches()[0]
        raw = chq.record_to_parquet(chq.DeviceRecordBatch.from_host(rec, ctx=ctx), ctx=ctx)
        want_raw = io.BytesIO()
        pq.write_table(t, want_raw, use_dictionary=False, compression="NONE")
        got_md = pq.ParquetFile(io.BytesIO(raw)).metadata.row_group(0)
        want_md = pq.ParquetFile(io.BytesIO(want_raw.getvalue())).metadata.row_group(0)
        for i, f in enumerate(rec.schema):
            g, w = got_md.column(i).statistics, want_md.column(i).statistics
            assert g is not None and g.null_count == w.null_count == rec.column(i).null_count, f.name
            assert g.has_min_max == w.has_min_max, (f.name, g.has_min_max, w.has_min_max)
            if w.has_min_max:
                same = lambda a, b: (a == b) and (not isinstance(a, float) or np.signbit(a) == np.signbit(b))
                assert same(g.min, w.min) and same(g.max, w.max), (f.name, g.min, w.min, g.max, w.max)
        from .helpers import batches_identical
        assert batches_identical(pq.read_table(io.BytesIO(raw)).to_batches()[0], rec)   # (bit-exact: NaNs included)


def test_chunks_are_cut_into_pages(ctx):
    """several data pages per chunk (4 096-row granularity, at most 64 per chunk): every column kind, nulls, slices at odd
    offsets; pyarrow and the scan read the same table back, and the page count is what the option asks for"""
    from .helpers import batches_identical
    c = chq.Context(0)
    for page_rows, n, want_pages in [(4096, 4096, 1), (4096, 4097, 2), (4096, 50_000, 13), (8192, 300_001, 37), (65_536, 300_001, 5),
                                     (4096, 1_000_000, 62)]:   # 245 pages asked -> capped: 64 pages of ceil(n / 64) rows, rounded up to 4 096
        c.set_option("parquet_page_rows", page_rows)
        for nulls in (False, True):
            t = sample_table(n, seed=n + page_rows, nulls=nulls, strings="mixed")
            rec = t.to_batches()[0] if n < 60_000 else pa.Table.from_batches(t.to_batches()).combine_chunks().to_batches()[0]
            for source in (chq.DeviceRecordBatch.from_host(rec, ctx=c), rec.slice(3, n - 5)):
                raw = chq.record_to_parquet(source, ctx=c)
                want = source.to_host() if hasattr(source, "to_host") else source
                assert batches_identical(pq.read_table(io.BytesIO(raw)).combine_chunks().to_batches()[0], want)
                f = chq.ParquetFile(raw)
                assert batches_identical(f.read_row_group(0, ctx=c).to_host(), want)
                rows = want.num_rows
                pages = {line.split()[1]: int(line.split()[-1]) for line in f.describe().splitlines() if line.startswith("chunk ")}
                expect = want_pages if rows == n else None
                for i, fld in enumerate(want.schema):
                    one_page = pa.types.is_boolean(fld.type) and want.column(i).null_count > 0   # bit-packed values without gaps
                    if expect is not None:
                        assert pages[str(i)] == (1 if one_page else expect), (fld.name, pages, expect)
    c.close()


def test_several_records_become_one_file_with_a_row_group_each(ctx):
    """chq_records_to_parquet: the row-group compaction the reference plans for its materialize task (DEV_NOTES.md:117-121)"""
    from .helpers import batches_identical
    parts = [sample_table(n, seed=50 + i, nulls=bool(i % 2), strings="mixed").to_batches()[0] for i, n in enumerate([10_000, 1, 70_000, 4096, 333])]
    schema = parts[0].schema
    parts = [p.cast(schema) if p.schema != schema else p for p in parts]
    for sources in (parts, [chq.DeviceRecordBatch.from_host(p, ctx=ctx) for p in parts], [parts[0], chq.DeviceRecordBatch.from_host(parts[1], ctx=ctx)] + parts[2:]):
        raw = chq.records_to_parquet(sources, ctx=ctx)
        f = pq.ParquetFile(io.BytesIO(raw))
        assert f.metadata.num_row_groups == len(parts) and f.metadata.num_rows == sum(p.num_rows for p in parts)
        mine = chq.ParquetFile(raw)
        for i, p in enumerate(parts):
            assert f.metadata.row_group(i).num_rows == p.num_rows
            assert batches_identical(f.read_row_group(i).combine_chunks().to_batches()[0], p)
            assert batches_identical(mine.read_row_group(i, ctx=ctx).to_host(), p)
            st = f.metadata.row_group(i).column(0).statistics
            assert st.has_min_max and st.min == pa.compute.min(p.column(0)).as_py() and st.max == pa.compute.max(p.column(0)).as_py()
        outs = mine.read_row_groups(ctx=ctx)
        assert [o.num_rows for o in outs] == [p.num_rows for p in parts]
    # one record: the same image as chq_record_to_parquet; mismatching schemas are refused
    assert chq.records_to_parquet([parts[0]], ctx=ctx) == chq.record_to_parquet(parts[0], ctx=ctx)
    other = pa.RecordBatch.from_arrays([pa.array([1, 2], pa.int64())], names=["id"])
    with pytest.raises(chq.ChqError) as e:
        chq.records_to_parquet([parts[0], other], ctx=ctx)
    assert e.value.code == 22


def test_sliced_batches_are_written_from_their_first_row(ctx):
    t = sample_table(5000, seed=77, nulls=True)
    for off, ln in [(1, 100), (3, 4000), (64, 4936), (4097, 500)]:
        rec = t.slice(off, ln).to_batches()[0]
        for source in (rec, chq.DeviceRecordBatch.from_host(rec, ctx=ctx)):
            back = pq.read_table(io.BytesIO(chq.record_to_parquet(source, ctx=ctx)))
            assert back.to_batches()[0].equals(rec) if ln else back.num_rows == 0


def test_filter_project_write_stays_on_the_device_until_the_file_image(ctx):
    """scan -> filter -> project -> write: the reference's whole query DAG for simple.sql, host memory only at both ends"""
    n = 30_000
    rng = np.random.default_rng(31)
    t = pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "value1": pa.array(["%08x" % v for v in rng.integers(0, 2**32, n)]),
                  "value2": pa.array((rng.random(n) * 100).astype(np.float32))})
    raw_in = write_bytes(t)
    from chapterhouseqe_amd.sqlparse import parse_select
    items = parse_select("select id, value2 * 2.0 as twice, value1 from t").projection
    dev = chq.ParquetFile(raw_in).read_row_group(0, ctx=ctx)
    al = chq.get_record_table_aliases(None, dev)
    e = parse_expr("value2 > 50.0")
    kept = chq.filter_record(dev, al, e, ctx=ctx)
    proj = chq.project_record(items, kept, al, ctx=ctx)
    got = pq.read_table(io.BytesIO(chq.record_to_parquet(proj, ctx=ctx))).to_batches()[0]
    host = t.to_batches()[0]
    exp = O.project_record(items, O.filter_record(host, al, e), al)
    assert got.equals(exp)


def test_unsupported_types_for_writing_say_so(ctx):
    rec = pa.record_batch({"d": pa.array([1, 2, 3], type=pa.date32())})
    with pytest.raises(chq.ChqError) as e:
        chq.record_to_parquet(rec, ctx=ctx)
    assert e.value.code == 30 and "date" in str(e.value).lower() or "tdD" in str(e.value)


def test_damaged_pages_are_reported_not_decoded_into_garbage_addresses(ctx):
    """bytes of the value streams overwritten: dictionary indices beyond the dictionary, length prefixes beyond the page,
    truncated level runs -- the kernels clamp every access to its page and the call reports the column"""
    n = 20_000
    rng = np.random.default_rng(41)
    t = pa.table({"k": pa.array(rng.integers(0, 50, n).astype(np.int32)), "s": pa.array(["v%05d" % v for v in rng.integers(0, 99999, n)]),
                  "o": pa.array(rng.integers(0, 9, n), type=pa.int64(), mask=rng.random(n) < 0.3)})
    raw = write_bytes(t, data_page_size=4096)
    md = pq.ParquetFile(io.BytesIO(raw)).metadata
    reported = 0
    for col in range(3):
        cm = md.row_group(0).column(col)
        start, size = cm.data_page_offset, cm.total_compressed_size
        for frac in (0.1, 0.5, 0.9):
            bad = bytearray(raw)
            at = start + int(size * frac * 0.8)
            bad[at:at + 64] = b"\xff" * 64
            try:
                f = chq.ParquetFile(bytes(bad))
                got = f.read_row_group(0, ctx=ctx)
                assert got.num_rows == n       # damage that still decodes (e.g. inside string bytes) is fine
            except chq.ChqError as e:
                assert e.code in (22, 30), str(e)
                reported += 1
    assert reported >= 3


def test_row_groups_read_together_equal_row_groups_read_one_by_one(ctx):
    t = sample_table(23_000, seed=61, nulls=True)
    raw = write_bytes(t, row_group_size=3000, data_page_size=2048)
    f = chq.ParquetFile(raw)
    one = [f.read_row_group(i, ctx=ctx).to_host() for i in range(f.num_row_groups)]
    for first, count in [(0, None), (2, 3), (7, 1), (f.num_row_groups, 0)]:
        got = f.read_row_groups(first, count, ctx=ctx)
        n = f.num_row_groups - first if count is None else count
        assert len(got) == n
        for k, g in enumerate(got):
            assert g.to_host().equals(one[first + k])
    host = f.read_row_groups(ctx=ctx, device_result=False)
    assert all(h.equals(o) for h, o in zip(host, one))
    with pytest.raises(chq.ChqError):
        f.read_row_groups(5, 100, ctx=ctx)
    assert pa.Table.from_batches(one).equals(pq.read_table(io.BytesIO(raw)).combine_chunks()) or \
        pa.Table.from_batches(one).combine_chunks().equals(pq.read_table(io.BytesIO(raw)).combine_chunks())


def test_file_image_without_a_copy(ctx):
    rec = sample_table(5000, seed=71, nulls=True).to_batches()[0]
    img = chq.record_to_parquet(rec, ctx=ctx, copy=False)
    assert len(img) == len(img.view) > 8 and bytes(img.view[:4]) == b"PAR1" and bytes(img.view[-4:]) == b"PAR1"
    assert pq.read_table(io.BytesIO(bytes(img.view))).to_batches()[0].equals(rec)
    img.release()
    assert len(img) == 0


def test_scan_on_a_context_that_borrows_torchs_default_stream():
    """bench.py (and any torch caller) hands the library torch's current stream, which the binding spells hipStreamLegacy.
    ROCm 7.2's hipStreamWaitEvent dereferences that handle when it finds it in an event (the round-2
    segfault); chq_ctx_create keeps the null spelling of the same stream, so the scan's auxiliary streams fork from / join
    into it with events again"""
    import torch
    c = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    assert c.stream == 0          # the legacy default stream under its null handle: no special handle reaches an event
    t = sample_table(30_000, seed=81, nulls=True)
    raw = write_bytes(t, row_group_size=7000)
    f = chq.ParquetFile(raw)
    got = f.read_row_groups(ctx=c)
    want = pq.ParquetFile(io.BytesIO(raw))
    for i, g in enumerate(got):
        assert g.to_host().equals(want.read_row_group(i).to_batches()[0])
    img = chq.record_to_parquet(got[0], ctx=c)
    assert pq.read_table(io.BytesIO(img)).to_batches()[0].equals(want.read_row_group(0).to_batches()[0])
    c.close()
