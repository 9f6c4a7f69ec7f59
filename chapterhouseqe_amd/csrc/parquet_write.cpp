// parquet_write.cpp -- one record batch -> one complete Parquet file image in host memory.
//
// Replaces the encode the reference does with the `parquet` crate behind the projection
// (materialize_files_task.rs:128-141: AsyncArrowWriter::try_new(writer, schema, None) / write / close -- one file, one
// row group per record).  The value streams of the data pages are produced in HBM (parquet_write.hip; a non-null
// fixed-width column needs no kernel, its Arrow buffer IS the PLAIN stream), copied once into their place in the file
// image; page headers and the footer (FileMetaData) are written here with a small Thrift compact-protocol writer.
// Output format: PLAIN encoding, UNCOMPRESSED, data pages V1 of `parquet_page_rows` rows (default 65 536, at most 64 pages
// per chunk), definition levels = the Arrow validity bitmap behind a one-run header per page, chunk statistics (null_count,
// min_value / max_value, TypeDefinedOrder) like the parquet crate's writer records them.  Any Parquet reader decodes it
// (tests: pyarrow, and this library's own scan); it is NOT byte-identical to what the parquet crate writes (that one
// dictionary-encodes first and adds page indexes).
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
  const uint8_t* levels_dev = nullptr;   // optional column with nulls: the validity bits from bit 0 (definition levels, bit width 1)
  const void* stream = nullptr; int64_t stream_bytes = 0;
  std::vector<BufferPtr> keep;
  int64_t null_count = -1;
  bool has_minmax = false;
  std::string min_value, max_value;   // Statistics.min_value / max_value: PLAIN encoding of the value (Utf8: the bytes)
  BufferPtr block_sums;
  PwParams enc{};
  bool needs_scan = false, boolean = false;
  BufferPtr bool_bytes;
  bool has_validity = false;
  std::vector<unsigned long long> prefix;   // several pages: bytes of the value stream in front of every 4096-row block
};

// one data page of a column chunk
struct PagePlan {
  int64_t rows = 0;
  std::vector<uint8_t> head;                // Thrift page header + the host part of the level section
  const uint8_t* levels_dev = nullptr; int64_t levels_dev_bytes = 0;
  const uint8_t* stream = nullptr; int64_t stream_bytes = 0;
  int64_t total() const { return (int64_t)head.size() + levels_dev_bytes + stream_bytes; }
};

struct ChunkMeta { int64_t page_at, total; };
// one row group of the file: the batch on the device, its columns' streams and pages, where its chunks lie in the file
struct RowGroupPlan {
  Batch rec;
  int64_t rows = 0;
  std::vector<ColumnPlan> plan;
  std::vector<std::vector<PagePlan>> pages;
  std::vector<ChunkMeta> meta;
  std::vector<BufferPtr> keep;
};

// Streams, statistics and page layout of ONE batch as one row group whose first chunk starts at file offset `at` (advanced
// past the row group's last page).
RowGroupPlan plan_row_group(Context& ctx, const Batch& in, int64_t& at) {
  RowGroupPlan rg;
  rg.rec = to_device(ctx, in);
  const Batch& rec = rg.rec;
  const int64_t rows = rec.nrows;
  rg.rows = rows;
  if (rows >= (1ll << 31)) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "parquet: a batch of " + std::to_string(rows) + " rows in one page"};
  const size_t nc = rec.cols.size();
  rg.plan.resize(nc);
  std::vector<ColumnPlan>& plan = rg.plan;
  const int grid = ctx.num_cus * 8;
  const int64_t n_blocks = std::max<int64_t>(1, (rows + PW_BLOCK_ROWS_HOST - 1) / PW_BLOCK_ROWS_HOST);
  // Several pages per chunk (round 3): pages are what a reader decodes in parallel (this library's scan: one workgroup per
  // page), the parquet crate cuts them at ~1 MiB.  Page boundaries sit on multiples of 4096 rows (the block size of the
  // stream scan, and a whole number of level bytes); at most 64 pages per chunk, so that a file image needs a bounded
  // number of device-to-host copies.
  int64_t page_rows = std::max<int64_t>(PW_BLOCK_ROWS_HOST, ctx.opt_parquet_page_rows / PW_BLOCK_ROWS_HOST * PW_BLOCK_ROWS_HOST);
  if (rows > 64 * page_rows) page_rows = ((rows + 63) / 64 + PW_BLOCK_ROWS_HOST - 1) / PW_BLOCK_ROWS_HOST * PW_BLOCK_ROWS_HOST;
  const int64_t n_pages = std::max<int64_t>(1, (rows + page_rows - 1) / page_rows);
  auto totals = make_device_buffer(8 * (nc + 1), ctx.device);
  check_hip(hipMemsetAsync(totals->ptr, 0, 8 * (nc + 1), ctx.stream), "memset");
  // chunk statistics (round 3): min / max per column, like the parquet crate's writer (EnabledStatistics default)
  std::vector<long long> h_stats(3 * nc);
  for (size_t ci = 0; ci < nc; ++ci) {
    const bool str = rec.cols[ci].type == T_UTF8;
    h_stats[3 * ci] = str ? -1 : INT64_MAX; h_stats[3 * ci + 1] = str ? -1 : INT64_MIN; h_stats[3 * ci + 2] = 0;
  }
  auto d_stats = make_device_buffer(24 * nc + 16, ctx.device);
  if (nc) check_hip(hipMemcpyAsync(d_stats->ptr, h_stats.data(), 24 * nc, hipMemcpyHostToDevice, ctx.stream), "init statistics");
  std::vector<BufferPtr> stats_keep;

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
    pl.has_validity = has_validity;
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
    if (rows > 0) {
      PwStatsParams sp{};
      sp.n_rows = rows; sp.validity = e.validity; sp.bit_offset = c.offset; sp.values = e.values;
      sp.offsets = e.offsets; sp.data = e.data; sp.out = (long long*)d_stats->ptr + 3 * ci;
      sp.kind = pl.string ? PW_STATS_UTF8 : pl.boolean ? PW_STATS_U8 : c.type == T_I32 ? PW_STATS_I32 : c.type == T_I64 ? PW_STATS_I64 :
                c.type == T_F32 ? PW_STATS_F32 : PW_STATS_F64;
      const int sgrid = (int)std::min<int64_t>((rows + 255) / 256, (int64_t)ctx.num_cus * 4);
      if (pl.string) {
        auto cand = make_device_buffer((size_t)sgrid * 16 + 16, ctx.device);
        stats_keep.push_back(cand);
        sp.cand = (long long*)cand->ptr;
        check_hip(pw_launch_stats(sp, sgrid, ctx.stream), "launch pw_stats_kernel");
        sp.n_cand = sgrid;
        check_hip(pw_launch_stats(sp, 1, ctx.stream), "launch pw_stats_kernel (candidates)");
      } else check_hip(pw_launch_stats(sp, sgrid, ctx.stream), "launch pw_stats_kernel");
    }
    pl.needs_scan = rows > 0 && (pl.string || has_validity || pl.boolean);
    if (pl.needs_scan) {
      pl.block_sums = make_device_buffer((size_t)n_blocks * 8 + 16, ctx.device);
      e.block_sums = (unsigned long long*)pl.block_sums->ptr;
      check_hip(pw_launch_scan(e, ctx.stream), "launch pw_counts_kernel / pw_scan_kernel");
      if (n_pages > 1 && !pl.boolean) {   // (a nullable Boolean column stays one page: its values are bit-packed without gaps)
        pl.prefix.resize((size_t)n_blocks);
        check_hip(hipMemcpyAsync(pl.prefix.data(), pl.block_sums->ptr, (size_t)n_blocks * 8, hipMemcpyDeviceToHost, ctx.stream), "read back block prefix");
      }
    }
  }
  std::vector<unsigned long long> h_tot(nc + 1, 0);
  check_hip(hipMemcpyAsync(h_tot.data(), totals->ptr, 8 * nc, hipMemcpyDeviceToHost, ctx.stream), "read back");
  if (nc) check_hip(hipMemcpyAsync(h_stats.data(), d_stats->ptr, 24 * nc, hipMemcpyDeviceToHost, ctx.stream), "read back statistics");
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  for (size_t ci = 0; ci < nc; ++ci) {
    const Column& c = rec.cols[ci];
    ColumnPlan& pl = plan[ci];
    const long long mn = h_stats[3 * ci], mx = h_stats[3 * ci + 1], cnt = h_stats[3 * ci + 2];
    if (rows == 0 || cnt == 0) continue;   // all null (or all NaN): no min / max
    auto raw = [](const void* v, size_t n) { return std::string((const char*)v, n); };
    if (pl.string) {
      if (mn < 0 || mx < 0) continue;
      std::string* dst[2] = {&pl.min_value, &pl.max_value};
      const long long row[2] = {mn, mx};
      bool ok = true;
      for (int k = 0; k < 2 && ok; ++k) {
        int32_t o[2];
        check_hip(hipMemcpy(o, (const int32_t*)c.values0() + row[k], 8, hipMemcpyDeviceToHost), "statistics: offsets");
        const int64_t len = (int64_t)o[1] - o[0];
        if (len > 4096) { ok = false; break; }   // (a reader gains nothing from a page-sized bound)
        dst[k]->resize((size_t)len);
        if (len) check_hip(hipMemcpy(&(*dst[k])[0], c.data + o[0], (size_t)len, hipMemcpyDeviceToHost), "statistics: bytes");
      }
      pl.has_minmax = ok;
      continue;
    }
    auto unkey32 = [](long long k) { int32_t b = (int32_t)k; return (int32_t)(b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1)); };
    auto unkey64 = [](long long k) { return (long long)(k ^ (long long)(((unsigned long long)(k >> 63)) >> 1)); };
    switch (c.type) {
      case T_I32: { int32_t a = (int32_t)mn, b = (int32_t)mx; pl.min_value = raw(&a, 4); pl.max_value = raw(&b, 4); } break;
      case T_I64: pl.min_value = raw(&mn, 8); pl.max_value = raw(&mx, 8); break;
      case T_F32: {   // the parquet writers' zero rule: a zero minimum is written as -0.0, a zero maximum as +0.0
        int32_t a = unkey32(mn), b = unkey32(mx);
        if ((a & 0x7fffffff) == 0) a = (int32_t)0x80000000; if ((b & 0x7fffffff) == 0) b = 0;
        pl.min_value = raw(&a, 4); pl.max_value = raw(&b, 4); } break;
      case T_F64: {
        long long a = unkey64(mn), b = unkey64(mx);
        if ((a & 0x7fffffffffffffffLL) == 0) a = (long long)0x8000000000000000ULL; if ((b & 0x7fffffffffffffffLL) == 0) b = 0;
        pl.min_value = raw(&a, 8); pl.max_value = raw(&b, 8); } break;
      default: { uint8_t a = (uint8_t)mn, b = (uint8_t)mx; pl.min_value = raw(&a, 1); pl.max_value = raw(&b, 1); } break;   // Boolean
    }
    pl.has_minmax = true;
  }

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
    // definition levels (optional columns), bit width 1: the validity bits from bit 0, LSB first (cut per page below)
    if (pl.optional && rows > 0 && e.validity) {
      if ((c.offset & 7) == 0) pl.levels_dev = c.validity + (c.offset >> 3);
      else {
        auto vb = make_device_buffer((size_t)((rows + 63) / 64) * 8 + 16, ctx.device);
        PwParams b = e; b.out = (uint8_t*)vb->ptr;
        check_hip(pw_launch_shift_bits(b, grid, ctx.stream), "launch pw_shift_bits_kernel");
        pl.keep.push_back(vb);
        pl.levels_dev = (const uint8_t*)vb->ptr;
      }
    }
  }

  // ---- the file image: PAR1, column chunks (pages: header + levels + values), footer, footer length, PAR1 -----------------
  rg.meta.resize(nc);
  rg.pages.resize(nc);
  std::vector<ChunkMeta>& meta = rg.meta;
  std::vector<std::vector<PagePlan>>& pages = rg.pages;
  for (size_t ci = 0; ci < nc; ++ci) {
    const Column& c = rec.cols[ci];
    ColumnPlan& pl = plan[ci];
    // Boolean values are bit-packed: a page may start inside a byte unless every row has a value and pages hold 8 k rows
    const bool paged = n_pages > 1 && !(pl.boolean && pl.has_validity);
    const int64_t np = paged ? n_pages : 1;
    int64_t total = 0;
    for (int64_t k = 0; k < np; ++k) {
      PagePlan pg;
      const int64_t r0 = paged ? k * page_rows : 0;
      pg.rows = paged ? std::min(page_rows, rows - r0) : rows;
      // the page's slice of the value stream
      if (pl.stream_bytes > 0) {
        int64_t b0, b1;
        if (!paged) { b0 = 0; b1 = pl.stream_bytes; }
        else if (pl.boolean) { b0 = r0 / 8; b1 = (r0 + pg.rows + 7) / 8; }
        else if (!pl.needs_scan) { b0 = r0 * (int64_t)c.width; b1 = (r0 + pg.rows) * (int64_t)c.width; }
        else {
          b0 = (int64_t)pl.prefix[(size_t)(r0 / PW_BLOCK_ROWS_HOST)];
          b1 = k + 1 < np ? (int64_t)pl.prefix[(size_t)((r0 + pg.rows) / PW_BLOCK_ROWS_HOST)] : pl.stream_bytes;
        }
        pg.stream = (const uint8_t*)pl.stream + b0; pg.stream_bytes = b1 - b0;
      }
      // the level section: [u32 length][one run]
      std::vector<uint8_t> lv;
      if (pl.optional) {
        std::vector<uint8_t> run;
        auto varint = [&](uint64_t v) { while (v >= 0x80) { run.push_back((uint8_t)(v | 0x80)); v >>= 7; } run.push_back((uint8_t)v); };
        if (pg.rows == 0) { /* no runs */ }
        else if (pl.levels_dev) {   // one bit-packed run of ceil(rows / 8) groups
          const int64_t groups = (pg.rows + 7) / 8;
          varint(((uint64_t)groups << 1) | 1u);
          pg.levels_dev = pl.levels_dev + r0 / 8; pg.levels_dev_bytes = groups;
        } else { varint((uint64_t)pg.rows << 1); run.push_back(1); }   // one RLE run: `rows` times level 1
        const uint32_t len = (uint32_t)(run.size() + pg.levels_dev_bytes);
        lv.resize(4); memcpy(lv.data(), &len, 4);
        lv.insert(lv.end(), run.begin(), run.end());
      }
      const int64_t payload = (int64_t)lv.size() + pg.levels_dev_bytes + pg.stream_bytes;
      ThriftOut t;
      t.i32(1, PQ_DATA_PAGE); t.i32(2, payload); t.i32(3, payload);
      t.begin_struct(5);
      t.i32(1, pg.rows); t.i32(2, PQ_PLAIN); t.i32(3, PQ_RLE); t.i32(4, PQ_RLE);
      t.end_struct();
      t.byte(0);
      pg.head = std::move(t.o);
      pg.head.insert(pg.head.end(), lv.begin(), lv.end());
      total += pg.total();
      pages[ci].push_back(std::move(pg));
    }
    meta[ci] = {at, total};
    at += total;
  }
  // the device blocks the pages point into live as long as the plan
  rg.keep.push_back(totals); rg.keep.push_back(d_stats);
  for (auto& b : stats_keep) rg.keep.push_back(b);
  return rg;
}
}  // namespace

// One file image from one or more batches of the same schema: one row group per batch, in order (the reference writes one
// file per record, materialize_files_task.rs:128-141; several records per file = the row-group compaction its DEV_NOTES.md
// 117-121 asks for).
ParquetImage records_to_parquet(Context& ctx, const std::vector<const Batch*>& ins) {
  if (ins.empty()) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "parquet: no record batch to write"};
  int64_t at = 4;
  std::vector<RowGroupPlan> rgs;
  rgs.reserve(ins.size());
  for (const Batch* in : ins) {
    if (!in) throw ChqError{CHQ_ERR_INVALID_HANDLE, "null record batch"};
    rgs.push_back(plan_row_group(ctx, *in, at));
    const RowGroupPlan& a = rgs.front(); const RowGroupPlan& b = rgs.back();
    bool same = a.rec.cols.size() == b.rec.cols.size();
    for (size_t ci = 0; same && ci < a.rec.cols.size(); ++ci)
      same = a.rec.cols[ci].name == b.rec.cols[ci].name && a.plan[ci].physical == b.plan[ci].physical && a.plan[ci].string == b.plan[ci].string &&
             a.plan[ci].optional == b.plan[ci].optional;
    if (!same) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "parquet: record batch " + std::to_string(rgs.size() - 1) + " has another schema than the first (names, types and nullability must agree)"};
  }
  const RowGroupPlan& first = rgs.front();
  const size_t nc = first.rec.cols.size();
  int64_t rows_all = 0;
  for (const RowGroupPlan& rg : rgs) rows_all += rg.rows;
  ThriftOut f;
  f.i32(1, 1);
  f.list(2, ThriftOut::T_STRUCT, nc + 1);
  f.begin_element(); f.str(4, "arrow_schema"); f.i32(5, (int64_t)nc); f.end_struct();
  for (size_t ci = 0; ci < nc; ++ci) {
    const ColumnPlan& pl = first.plan[ci];
    f.begin_element();
    f.i32(1, pl.physical); f.i32(3, pl.optional ? 1 : 0); f.str(4, first.rec.cols[ci].name);
    if (pl.string) { f.i32(6, 0); f.begin_struct(10); f.begin_struct(1); f.end_struct(); f.end_struct(); }   // UTF8 / LogicalType.STRING
    f.end_struct();
  }
  f.i64(3, rows_all);
  f.list(4, ThriftOut::T_STRUCT, rgs.size());
  for (const RowGroupPlan& rg : rgs) {
    f.begin_element();
    f.list(1, ThriftOut::T_STRUCT, nc);
    int64_t total_bytes = 0;
    for (size_t ci = 0; ci < nc; ++ci) {
      const ColumnPlan& pl = rg.plan[ci];
      const std::string& name = rg.rec.cols[ci].name;
      f.begin_element();
      f.i64(2, rg.meta[ci].page_at);
      f.begin_struct(3);
      f.i32(1, pl.physical);
      f.list(2, ThriftOut::T_I32, 2); f.zigzag(PQ_PLAIN); f.zigzag(PQ_RLE);
      f.list(3, ThriftOut::T_BINARY, 1); f.varint(name.size()); f.o.insert(f.o.end(), name.begin(), name.end());
      f.i32(4, 0); f.i64(5, rg.rows); f.i64(6, rg.meta[ci].total); f.i64(7, rg.meta[ci].total); f.i64(9, rg.meta[ci].page_at);
      if (pl.null_count >= 0 || pl.has_minmax) {
        f.begin_struct(12);
        if (pl.null_count >= 0) f.i64(3, pl.null_count);
        if (pl.has_minmax) { f.str(5, pl.max_value); f.str(6, pl.min_value); }
        f.end_struct();
      }
      f.end_struct();
      f.end_struct();
      total_bytes += rg.meta[ci].total;
    }
    f.i64(2, total_bytes); f.i64(3, rg.rows);
    f.end_struct();
  }
  f.str(6, "chapterhouseqe_amd (MI355X page encoder)");
  // column_orders: TypeDefinedOrder for every column -- without it readers must ignore min_value / max_value
  f.list(7, ThriftOut::T_STRUCT, nc);
  for (size_t ci = 0; ci < nc; ++ci) { f.begin_element(); f.begin_struct(1); f.end_struct(); f.end_struct(); }
  f.byte(0);
  const int64_t file_len = at + (int64_t)f.o.size() + 8;

  ParquetImage img;
  img.bytes = make_host_buffer((size_t)file_len + 16);
  img.len = file_len;
  uint8_t* out = (uint8_t*)img.bytes->ptr;
  memcpy(out, "PAR1", 4);
  // The body [4, at) is assembled in HBM (page heads uploaded as one blob, levels and value streams copied to their file
  // offsets by pw_assemble_kernel) and comes down in ONE copy: one copy per page part cost 6 ms for a 4 M-row batch in
  // 62 pages per chunk, against 1.6 ms for the single-page form.
  if (at > 4) {
    size_t n_pieces = 0, blob_bytes = 0;
    for (const RowGroupPlan& rg : rgs)
      for (size_t ci = 0; ci < nc; ++ci) for (const PagePlan& pg : rg.pages[ci]) { n_pieces += 1 + (pg.levels_dev_bytes > 0) + (pg.stream_bytes > 0); blob_bytes += pg.head.size(); }
    const size_t pieces_bytes = (n_pieces * sizeof(PwPiece) + 63) & ~(size_t)63;
    std::vector<uint8_t> up(pieces_bytes + blob_bytes);
    auto d_up = make_device_buffer(up.size() + 64, ctx.device);
    auto d_img = make_device_buffer((size_t)at + 64, ctx.device);
    PwPiece* pc = (PwPiece*)up.data();
    size_t k = 0, bo = pieces_bytes;
    unsigned long long longest = 0;
    for (const RowGroupPlan& rg : rgs) for (size_t ci = 0; ci < nc; ++ci) {
      int64_t p = rg.meta[ci].page_at;
      for (const PagePlan& pg : rg.pages[ci]) {
        memcpy(up.data() + bo, pg.head.data(), pg.head.size());
        pc[k++] = PwPiece{(unsigned long long)p, (const uint8_t*)d_up->ptr + bo, (unsigned long long)pg.head.size()};
        bo += pg.head.size(); p += (int64_t)pg.head.size();
        if (pg.levels_dev_bytes) { pc[k++] = PwPiece{(unsigned long long)p, pg.levels_dev, (unsigned long long)pg.levels_dev_bytes}; p += pg.levels_dev_bytes; }
        if (pg.stream_bytes) { pc[k++] = PwPiece{(unsigned long long)p, pg.stream, (unsigned long long)pg.stream_bytes}; p += pg.stream_bytes; }
      }
    }
    for (size_t i = 0; i < n_pieces; ++i) longest = std::max(longest, pc[i].len);
    unsigned long long seg = 256 * 1024;
    while ((longest + seg - 1) / seg > 65535) seg *= 2;
    check_hip(hipMemcpyAsync(d_up->ptr, up.data(), up.size(), hipMemcpyHostToDevice, ctx.stream), "upload page heads");
    PwAssembleParams ap{(const PwPiece*)d_up->ptr, (uint8_t*)d_img->ptr, seg};
    check_hip(pw_launch_assemble(ap, (int)n_pieces, (int)std::max<unsigned long long>(1, (longest + seg - 1) / seg), ctx.stream), "launch pw_assemble_kernel");
    check_hip(hipMemcpyAsync(out + 4, (const uint8_t*)d_img->ptr + 4, (size_t)at - 4, hipMemcpyDeviceToHost, ctx.stream), "copy the file body");
    check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");   // (`up` and the device blocks are released below)
  }
  memcpy(out + at, f.o.data(), f.o.size());
  const uint32_t flen = (uint32_t)f.o.size();
  memcpy(out + at + f.o.size(), &flen, 4);
  memcpy(out + at + f.o.size() + 4, "PAR1", 4);
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  return img;
}

ParquetImage record_to_parquet(Context& ctx, const Batch& in) { return records_to_parquet(ctx, std::vector<const Batch*>{&in}); }

}  // namespace chq
