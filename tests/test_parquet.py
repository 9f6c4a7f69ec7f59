"""CPU tier of the Parquet scan: the library's footer / page-header reader (a hand-written Thrift
compact-protocol reader, `csrc/parquet_meta.cpp`) against pyarrow's own view of files pyarrow wrote.  No GPU: only
`chq_parquet_open` / `chq_parquet_describe`."""
import ctypes as C
import io

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

import chapterhouseqe_amd as chq
from chapterhouseqe_amd import _lib as L
from tests.parquet_cases import sample_table, write_bytes


def parse(desc: str):
    lines = desc.splitlines()
    head = dict(zip(lines[0].split()[::2], lines[0].split()[1::2]))
    cols = [l.split() for l in lines if l.startswith("column ")]
    rgs, cur = [], None
    for l in lines:
        t = l.split()
        if t[0] == "rg":
            cur = {"rows": int(t[3]), "chunks": []}; rgs.append(cur)
        elif t[0] == "chunk":
            cur["chunks"].append({"values": int(t[3]), "codec": int(t[5]), "pages": []})
        elif t[0] == "page":
            cur["chunks"][-1]["pages"].append({"type": int(t[1]), "values": int(t[3]), "enc": int(t[5]), "bytes": int(t[7]), "header": int(t[9])})
    return head, cols, rgs


@pytest.mark.parametrize("kw", [
    dict(), dict(use_dictionary=False), dict(data_page_version="2.0"), dict(data_page_size=512, row_group_size=3000),
    dict(compression="snappy"), dict(use_dictionary=False, data_page_version="2.0", row_group_size=1000, write_statistics=False),
])
def test_metadata_matches_pyarrow(kw):
    t = sample_table(7001, seed=3, nulls=True)
    raw = write_bytes(t, **kw)
    md = pq.ParquetFile(io.BytesIO(raw)).metadata
    f = chq.ParquetFile(raw)
    head, cols, rgs = parse(f.describe())
    assert int(head["rows"]) == md.num_rows and int(head["row_groups"]) == md.num_row_groups == f.num_row_groups
    assert [c[1] for c in cols] == [md.schema.column(i).name for i in range(md.num_columns)]
    assert [c[2] for c in cols] == [md.schema.column(i).physical_type for i in range(md.num_columns)]
    for g in range(md.num_row_groups):
        assert rgs[g]["rows"] == md.row_group(g).num_rows == f.row_group_num_rows(g)
        for c in range(md.num_columns):
            cm = md.row_group(g).column(c)
            ch = rgs[g]["chunks"][c]
            assert ch["values"] == cm.num_values
            assert (ch["codec"] == 0) == (cm.compression == "UNCOMPRESSED")
            data_pages = [p for p in ch["pages"] if p["type"] in (0, 3)]
            assert sum(p["values"] for p in data_pages) == cm.num_values
            assert (len([p for p in ch["pages"] if p["type"] == 2]) == 1) == cm.has_dictionary_page
            # the pages (header + payload each) tile the chunk exactly
            assert sum(p["bytes"] + p["header"] for p in ch["pages"]) == cm.total_compressed_size
    f.close()


def test_malformed_files_are_rejected_with_a_message():
    raw = write_bytes(sample_table(100, seed=1))
    for bad in [b"", b"PAR1", raw[:-1], raw[:-4] + b"XXXX", b"XXXX" + raw[4:], raw[:len(raw) // 2] + raw[-8:]]:
        with pytest.raises(chq.ChqError):
            chq.ParquetFile(bad)
    # a footer length that points outside the file
    broken = bytearray(raw); broken[-8:-4] = (len(raw) * 2).to_bytes(4, "little")
    with pytest.raises(chq.ChqError) as e:
        chq.ParquetFile(bytes(broken))
    assert "footer" in str(e.value)


def test_nested_schemas_are_refused_at_open():
    t = pa.table({"a": pa.array([[1, 2], [3]], type=pa.list_(pa.int32()))})
    with pytest.raises(chq.ChqError) as e:
        chq.ParquetFile(write_bytes(t))
    assert e.value.code == 30 and "nested" in str(e.value)


def test_the_reference_sample_shape():
    """create_sample_data.rs writes id:Int32, value1:Utf8, value2:Float32 with default writer properties: uncompressed,
    dictionary with PLAIN fallback, V1 pages -- the same settings here"""
    n = 20_000
    rng = np.random.default_rng(0)
    t = pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "value1": pa.array(["%08x" % v for v in rng.integers(0, 2**32, n)]),
                  "value2": pa.array(rng.random(n).astype(np.float32) * 100)})
    f = chq.ParquetFile(write_bytes(t, dictionary_pagesize_limit=32 * 1024, data_page_size=16 * 1024))
    head, cols, rgs = parse(f.describe())
    assert [c[2] for c in cols] == ["INT32", "BYTE_ARRAY", "FLOAT"] and "utf8" in cols[1]
    encs = {p["enc"] for ch in rgs[0]["chunks"] for p in ch["pages"] if p["type"] == 0}
    assert encs == {0, 8}      # dictionary pages first, PLAIN after the dictionary outgrew its limit: both in one chunk


def test_corrupted_metadata_never_crashes_the_reader():
    """random byte damage inside the footer and the page headers: every outcome is either a parsed file or a ChqError"""
    raw = bytearray(write_bytes(sample_table(3000, seed=4, nulls=True), data_page_size=1024))
    footer_len = int.from_bytes(raw[-8:-4], "little")
    rng = np.random.default_rng(12)
    opened = failed = 0
    for trial in range(400):
        bad = bytearray(raw)
        lo = len(raw) - 8 - footer_len if trial % 2 == 0 else 4
        hi = len(raw) - 8 if trial % 2 == 0 else len(raw) - 8 - footer_len
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(lo, hi))] = int(rng.integers(0, 256))
        try:
            f = chq.ParquetFile(bytes(bad))
            f.describe()
            f.close()
            opened += 1
        except chq.ChqError:
            failed += 1
    assert opened + failed == 400 and failed > 20


def _thrift_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _zigzag(v):
    return _thrift_varint((v << 1) ^ (v >> 63))


def test_sizes_near_int64_max_are_compared_by_subtraction():
    """Page and chunk sizes come from an untrusted file.  `payload_at + compressed_size > size` wraps for a size near
    INT64_MAX and would accept the page (read_files takes user files): every size in a page header or the footer that points
    outside its container must be refused at open, with INVALID_ARGUMENT, before any cast to 32 bits."""
    t = pa.table({"v": pa.array(np.arange(1000, dtype=np.int32))})
    raw = write_bytes(t, use_dictionary=False, write_statistics=False)
    md = pq.ParquetFile(io.BytesIO(raw)).metadata
    at = md.row_group(0).column(0).data_page_offset
    # the data page header: field 1 (type, i32), 2 (uncompressed size), 3 (compressed size) -- patch field 3 with a huge value
    # by rebuilding the three leading fields (the rest of the header follows unchanged)
    body = raw[at:]
    assert body[0] == 0x15            # field 1, i32
    p = 1
    while body[p] & 0x80:
        p += 1
    p += 1
    assert body[p] == 0x15            # field 2
    q = p + 1
    while body[q] & 0x80:
        q += 1
    q += 1
    assert body[q] == 0x15            # field 3
    r = q + 1
    while body[r] & 0x80:
        r += 1
    r += 1
    for evil in (2**62, 2**31 + 5, (1 << 63) - 1):
        patched = raw[:at] + body[:q + 1] + _zigzag(evil) + body[r:]
        # keep the footer where the file's tail says it is: only the page header grew
        f = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.lib().chq_parquet_open(patched, len(patched), C.byref(f), err, len(err))
        assert rc == 22, (rc, err.value)
        assert not f.value


def test_damaged_metadata_across_codecs_and_page_versions_never_crashes():
    """Mutation fuzz of the metadata reader (footer + page headers, the hand-written Thrift reader): random byte changes in
    valid files (uncompressed V1, snappy V2, small PLAIN pages) must end in a parsed description or in a ChqError, never in
    a crash or a hang -- read_files takes user files (sizes and counts of an untrusted file)."""
    rng = np.random.default_rng(9)
    n = 3000
    t = pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "s": pa.array(["v%d" % (i % 50) for i in range(n)]),
                  "x": pa.array(rng.random(n).astype(np.float32), mask=rng.random(n) < 0.1)})
    raws = []
    for kw in (dict(compression="NONE"), dict(compression="SNAPPY", data_page_version="2.0"),
               dict(compression="NONE", use_dictionary=False, data_page_size=2048)):
        b = io.BytesIO()
        pq.write_table(t, b, row_group_size=1000, **kw)
        raws.append(b.getvalue())
    parsed = rejected = 0
    for it in range(900):
        raw = bytearray(raws[it % len(raws)])
        footer_at = len(raw) - 8 - int.from_bytes(raw[-8:-4], "little")
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(footer_at, len(raw))) if rng.random() < 0.7 else int(rng.integers(4, footer_at))
            mode = rng.random()
            raw[pos] = int(rng.integers(0, 256)) if mode < 0.5 else (raw[pos] ^ (1 << int(rng.integers(0, 8)))) if mode < 0.8 else 0xFF
        try:
            f = chq.ParquetFile(bytes(raw))
            f.describe()
            f.close()
            parsed += 1
        except chq.ChqError:
            rejected += 1
    assert parsed + rejected == 900 and rejected > 100 and parsed > 100, (parsed, rejected)
