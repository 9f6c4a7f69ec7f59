// parquet_write.cpp -- one record batch -> one complete Parquet file image in host memory.
//
// Replaces the encode the reference does with the `parquet` crate behind the projection
// (materialize_files_task.rs:128-141: AsyncArrowWriter::try_new(writer, schema, None) / write / close -- one file, one
// row group per record).  The value streams of the data pages are produced in HBM (parquet_write.hip; a non-null
// fixed-width column needs no kernel, its Arrow buffer IS the PLAIN stream), copied once into their place in the file
// image; page headers and the footer (FileMetaData) are written here with a small Thrift compact-protocol writer.
// Output format: PLAIN encoding, UNCOMPRESSED, data pages V1, one page per column chunk, definition levels = the Arrow
// validity bitmap behind a one-run header.  Any Parquet reader decodes it (tests: pyarrow, and this library's own scan);
// it is NOT byte-identical to what the parquet crate writes (that one dictionary-encodes and adds statistics).
// Types: Int32, Int64, Float32, Float64, Boolean, Utf8; anything else CHQ_ERR_NOT_SUPPORTED.
#include <cstring>

#include "engine.hpp"
#include "parquet.hpp"
#include "parquet_device.h"

namespace chq {
namespace {

struct ThriftOut {
  std::vector<uint8_t> o;
  std::vector<int> stack;
  int last = 0;
  enum { T_I32 = 5, T_I64 = 6, T_BINARY = 8, T_LIST = 9, T_STRUCT = 12 };
  void byte(uint8_t b) { o.push_back(b); }
  void varint(uint64_t v) { while (v >= 0x80) { byte((uint8_t)(v | 0x80)); v >>= 7; } byte((uint8_t)v); }
  void zigzag(int64_t v) { varint(((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
  void field(int id, int type) {
    const int delta = id - last;
    if (delta > 0 && delta <= 15) byte((uint8_t)(delta << 4 | type));
    else { byte((uint8_t)type); zigzag(id); }
    last = id;
  }
  void i32(int id, int64_t v) { field(id, T_I32); zigzag(v); }
  void i64(int id, int64_t v) { field(id, T_I64); zigzag(v); }
  void str(int id, const std::string& s) { field(id, T_BINARY); varint(s.size()); o.insert(o.end(), s.begin(), s.end()); }
  void begin_struct(int id) { field(id, T_STRUCT); stack.push_back(last); last = 0; }
  void begin_element() { stack.push_back(last); last = 0; }   // a struct inside a list: no field header
  void end_struct() { byte(0); last = stack.back(); stack.pop_back(); }
  void list(int id, int elem_type, size_t n) {
    field(id, T_LIST);
    if (n < 15) byte((uint8_t)(n << 4 | elem_type)); else { byte((uint8_t)(0xf0 | elem_type)); varint(n); }
  }
};

struct Piece { int64_t at; const void* src; int64_t bytes; };   // device bytes to land at file offset `at`

struct ColumnPlan {
  int physical = 0;
  bool optional = false, string = false;
  std::vector<uint8_t> levels;      // host part of the level section: [u32 length][run header ...]; the bitmap bytes follow from the device
  const uint8_t* levels_dev = nullptr; int64_t levels_dev_bytes = 0;
  const void* stream = nullptr; int64_t stream_bytes = 0;
  std::vector<BufferPtr> keep;
  int64_t null_count = -1;
  BufferPtr block_sums;
  PwParams enc{};
  bool needs_scan = false, boolean = false;
  BufferPtr bool_bytes;
};

}  // namespace

ParquetImage record_to_parquet(Context& ctx, const Batch& in) {
  const Batch rec = to_device(ctx, in);
  const int64_t rows = rec.nrows;
  if (rows >= (1ll << 31)) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "parquet: a batch of " + std::to_string(rows) + " rows in one page"};
  const size_t nc = rec.cols.size();
  std::vector<ColumnPlan> plan(nc);
  const int grid = ctx.num_cus * 8;
  const int64_t n_blocks = std::max<int64_t>(1, (rows + PW_BLOCK_ROWS_HOST - 1) / PW_BLOCK_ROWS_HOST);
  auto totals = make_device_buffer(8 * (nc + 1), ctx.device);
  check_hip(hipMemsetAsync(totals->ptr, 0, 8 * (nc + 1), ctx.stream), "memset");

  // ---- phase 1: sizes (scan of the bytes every row contributes) ------------------------------------------------------
  for (size_t ci = 0; ci < nc; ++ci) {
    const Column& c = rec.cols[ci];
    ColumnPlan& pl = plan[ci];
    switch (c.type) {
      case T_BOOL: pl.physical = PQ_BOOLEAN; pl.boolean = true; break;
      case T_I32: pl.physical = PQ_INT32; break;
      case T_I64: pl.physical = PQ_INT64; break;
      case T_F32: pl.physical = PQ_FLOAT; break;
      case T_F64: pl.physical = PQ_DOUBLE; break;
      case T_UTF8: pl.physical = PQ_BYTE_ARRAY; pl.string = true; break;
      default: throw ChqError{CHQ_ERR_NOT_SUPPORTED, "parquet: column '" + c.name + "' of Arrow type '" + c.format + "' (Int32, Int64, Float32, Float64, Boolean and Utf8 are written)"};
    }
    pl.optional = c.nullable;
    const bool has_validity = c.validity && c.null_count != 0;
    if (has_validity && !c.nullable) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "parquet: column '" + c.name + "' is declared non-nullable but carries a validity bitmap"};
    pl.null_count = has_validity ? c.null_count : 0;
    PwParams& e = pl.enc;
    e.n_rows = rows; e.validity = has_validity ? c.validity : nullptr; e.bit_offset = c.offset;
    e.n_blocks = n_blocks; e.total_bytes = (unsigned long long*)totals->ptr + ci;
    if (pl.string) { e.offsets = (const int32_t*)c.values0(); e.data = c.data; }
    else if (pl.boolean) {
      e.width = 1;
      if (rows > 0) {   // one byte per row first (any bit offset), compacted below like a 1-byte column
        pl.bool_bytes = make_device_buffer((size_t)rows + 64, ctx.device);
        PwParams b{}; b.n_rows = rows; b.values = c.values; b.value_bit_offset = c.offset; b.out = (uint8_t*)pl.bool_bytes->ptr;
        check_hip(pw_launch_bits_to_bytes(b, grid, ctx.stream), "launch pw_bits_to_bytes_kernel");
        e.values = (const uint8_t*)pl.bool_bytes->ptr;
      }
    } else { e.width = c.width; e.values = (const uint8_t*)c.values0(); }
    pl.needs_scan = rows > 0 && (pl.string || has_validity || pl.boolean);
    if (pl.needs_scan) {
      pl.block_sums = make_device_buffer((size_t)n_blocks * 8 + 16, ctx.device);
      e.block_sums = (unsigned long long*)pl.block_sums->ptr;
      check_hip(pw_launch_scan(e, ctx.stream), "launch pw_counts_kernel / pw_scan_kernel");
    }
  }
  std::vector<unsigned long long> h_tot(nc + 1, 0);
  check_hip(hipMemcpyAsync(h_tot.data(), totals->ptr, 8 * nc, hipMemcpyDeviceToHost, ctx.stream), "read back");
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");

  // ---- phase 2: value streams and level bitmaps in HBM ---------------------------------------------------------------
  for (size_t ci = 0; ci < nc; ++ci) {
    const Column& c = rec.cols[ci];
    ColumnPlan& pl = plan[ci];
    PwParams& e = pl.enc;
    if (rows == 0) { pl.stream_bytes = 0; }
    else if (!pl.needs_scan) {   // the Arrow buffer is the stream
      pl.stream = e.values; pl.stream_bytes = rows * (int64_t)c.width;
      if (pl.stream_bytes >= (1ll << 31) - 64) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "parquet: column '" + c.name + "' needs a page of " + std::to_string(pl.stream_bytes) + " bytes"};
    }
    else {
      const int64_t bytes = (int64_t)h_tot[ci];
      if (bytes >= (1ll << 31) - 64) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "parquet: column '" + c.name + "' needs a page of " + std::to_string(bytes) + " bytes"};
      auto sb = make_device_buffer((size_t)bytes + 64, ctx.device);
      e.out = (uint8_t*)sb->ptr;
      check_hip(pw_launch_encode(e, ctx.stream), "launch pw_encode_kernel");
      pl.keep.push_back(sb);
      pl.stream = sb->ptr; pl.stream_bytes = bytes;
      if (!pl.string && !pl.boolean && e.validity) pl.null_count = rows - bytes / c.width;
      if (pl.boolean) {   // the compacted bytes -> bit-packed
        const int64_t nv = bytes;
        auto bits = make_device_buffer((size_t)((nv + 63) / 64) * 8 + 16, ctx.device);
        PwParams b{}; b.n_rows = nv; b.values = (const uint8_t*)sb->ptr; b.out = (uint8_t*)bits->ptr;
        if (nv > 0) check_hip(pw_launch_bytes_to_bits(b, grid, ctx.stream), "launch pw_bytes_to_bits_kernel");
        pl.keep.push_back(bits);
        pl.stream = bits->ptr; pl.stream_bytes = (nv + 7) / 8;
        if (e.validity) pl.null_count = rows - nv;
      }
    }
    // definition levels (optional columns): bit width 1
    if (pl.optional) {
      std::vector<uint8_t> run;
      auto varint = [&](uint64_t v) { while (v >= 0x80) { run.push_back((uint8_t)(v | 0x80)); v >>= 7; } run.push_back((uint8_t)v); };
      int64_t dev_bytes = 0;
      if (rows == 0) { /* no runs */ }
      else if (e.validity) {   // one bit-packed run: the validity bits from bit 0, LSB first
        const int64_t groups = (rows + 7) / 8;
        varint(((uint64_t)groups << 1) | 1u);
        dev_bytes = groups;
        if ((c.offset & 7) == 0) pl.levels_dev = c.validity + (c.offset >> 3);
        else {
          auto vb = make_device_buffer((size_t)((rows + 63) / 64) * 8 + 16, ctx.device);
          PwParams b = e; b.out = (uint8_t*)vb->ptr;
          check_hip(pw_launch_shift_bits(b, grid, ctx.stream), "launch pw_shift_bits_kernel");
          pl.keep.push_back(vb);
          pl.levels_dev = (const uint8_t*)vb->ptr;
        }
      } else { varint((uint64_t)rows << 1); run.push_back(1); }   // one RLE run: `rows` times level 1
      const uint32_t len = (uint32_t)(run.size() + dev_bytes);
      pl.levels.resize(4); memcpy(pl.levels.data(), &len, 4);
      pl.levels.insert(pl.levels.end(), run.begin(), run.end());
      pl.levels_dev_bytes = dev_bytes;
    }
  }

  // ---- the file image: PAR1, column chunks (page header + levels + values), footer, footer length, PAR1 ----------------
  struct ChunkMeta { int64_t page_at, total; };
  std::vector<ChunkMeta> meta(nc);
  std::vector<std::vector<uint8_t>> headers(nc);
  int64_t at = 4;
  for (size_t ci = 0; ci < nc; ++ci) {
    ColumnPlan& pl = plan[ci];
    const int64_t payload = (int64_t)pl.levels.size() + pl.levels_dev_bytes + pl.stream_bytes;
    ThriftOut t;
    t.i32(1, PQ_DATA_PAGE); t.i32(2, payload); t.i32(3, payload);
    t.begin_struct(5);
    t.i32(1, rows); t.i32(2, PQ_PLAIN); t.i32(3, PQ_RLE); t.i32(4, PQ_RLE);
    t.end_struct();
    t.byte(0);
    headers[ci] = std::move(t.o);
    meta[ci] = {at, (int64_t)headers[ci].size() + payload};
    at += meta[ci].total;
  }
  ThriftOut f;
  f.i32(1, 1);
  f.list(2, ThriftOut::T_STRUCT, nc + 1);
  f.begin_element(); f.str(4, "arrow_schema"); f.i32(5, (int64_t)nc); f.end_struct();
  for (size_t ci = 0; ci < nc; ++ci) {
    const ColumnPlan& pl = plan[ci];
    f.begin_element();
    f.i32(1, pl.physical); f.i32(3, pl.optional ? 1 : 0); f.str(4, rec.cols[ci].name);
    if (pl.string) { f.i32(6, 0); f.begin_struct(10); f.begin_struct(1); f.end_struct(); f.end_struct(); }   // UTF8 / LogicalType.STRING
    f.end_struct();
  }
  f.i64(3, rows);
  f.list(4, ThriftOut::T_STRUCT, 1);
  f.begin_element();
  f.list(1, ThriftOut::T_STRUCT, nc);
  int64_t total_bytes = 0;
  for (size_t ci = 0; ci < nc; ++ci) {
    const ColumnPlan& pl = plan[ci];
    f.begin_element();
    f.i64(2, meta[ci].page_at);
    f.begin_struct(3);
    f.i32(1, pl.physical);
    f.list(2, ThriftOut::T_I32, 2); f.zigzag(PQ_PLAIN); f.zigzag(PQ_RLE);
    f.list(3, ThriftOut::T_BINARY, 1); f.varint(rec.cols[ci].name.size()); f.o.insert(f.o.end(), rec.cols[ci].name.begin(), rec.cols[ci].name.end());
    f.i32(4, 0); f.i64(5, rows); f.i64(6, meta[ci].total); f.i64(7, meta[ci].total); f.i64(9, meta[ci].page_at);
    if (pl.null_count >= 0) { f.begin_struct(12); f.i64(3, pl.null_count); f.end_struct(); }
    f.end_struct();
    f.end_struct();
    total_bytes += meta[ci].total;
  }
  f.i64(2, total_bytes); f.i64(3, rows);
  f.end_struct();
  f.str(6, "chapterhouseqe_amd (MI355X page encoder)");
  f.byte(0);
  const int64_t file_len = at + (int64_t)f.o.size() + 8;

  ParquetImage img;
  img.bytes = make_host_buffer((size_t)file_len + 16);
  img.len = file_len;
  uint8_t* out = (uint8_t*)img.bytes->ptr;
  memcpy(out, "PAR1", 4);
  for (size_t ci = 0; ci < nc; ++ci) {
    const ColumnPlan& pl = plan[ci];
    int64_t p = meta[ci].page_at;
    memcpy(out + p, headers[ci].data(), headers[ci].size()); p += (int64_t)headers[ci].size();
    if (!pl.levels.empty()) { memcpy(out + p, pl.levels.data(), pl.levels.size()); p += (int64_t)pl.levels.size(); }
    if (pl.levels_dev_bytes) { check_hip(hipMemcpyAsync(out + p, pl.levels_dev, (size_t)pl.levels_dev_bytes, hipMemcpyDeviceToHost, ctx.stream), "copy levels"); p += pl.levels_dev_bytes; }
    if (pl.stream_bytes) check_hip(hipMemcpyAsync(out + p, pl.stream, (size_t)pl.stream_bytes, hipMemcpyDeviceToHost, ctx.stream), "copy values");
  }
  memcpy(out + at, f.o.data(), f.o.size());
  const uint32_t flen = (uint32_t)f.o.size();
  memcpy(out + at + f.o.size(), &flen, 4);
  memcpy(out + at + f.o.size() + 4, "PAR1", 4);
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  return img;
}

}  // namespace chq
