// parquet_scan.cpp -- orchestration of the GPU page decode (parquet.hip) for one row group.  See parquet.hpp for scope.
#include <algorithm>
#include <cstring>
#include <functional>

#include "engine.hpp"
#include "parquet.hpp"
#include "parquet_device.h"

namespace chq {
namespace {

[[noreturn]] void unsupported(const std::string& what) { throw ChqError{CHQ_ERR_NOT_SUPPORTED, "parquet: " + what}; }
[[noreturn]] void malformed(const std::string& what) { throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "parquet: " + what}; }

// the columns' streams: forked off ctx.stream and joined back into it with events (engine.cpp: fork_aux_streams /
// join_aux_streams).  Round 2 synchronised on the host instead, after a segfault whose cause is now known: the context ran
// on the handle hipStreamLegacy, and ROCm 7.2's hipStreamWaitEvent dereferences that handle when it finds it in an event
// (chq_ctx_create now keeps the null spelling of the same stream).
void fork_streams(Context& ctx) { fork_aux_streams(ctx); }
void join_streams(Context& ctx) { join_aux_streams(ctx); }
// an error thrown while columns are in flight: wait for them before their temporaries go back to the pool
struct DrainOnUnwind {
  Context& ctx;
  ~DrainOnUnwind() {
    for (int i = 0; i < Context::kAuxStreams; ++i) if (ctx.aux[i]) (void)hipStreamSynchronize(ctx.aux[i]);
    (void)hipStreamSynchronize(ctx.stream);
  }
};

struct Scalars {   // per column, on the device and read back once per phase
  uint32_t err, total_values;
  unsigned long long total_bytes;
};

struct ColumnWork {
  const PqColumnSchema* schema = nullptr;
  const PqColumnChunk* chunk = nullptr;
  const char* format = nullptr;
  int width = 0;              // fixed-width types
  bool byte_array = false, boolean = false;
  bool has_levels = false;    // definition levels are decoded (optional column whose statistics do not rule nulls out)
  int64_t rows = 0;
  BufferPtr chunkb, pages_dev, nonnull, value_base, valid8, row_val, dense, vsrc, vlen, dict_src, dict_len, offsets, block_sums;
  BufferPtr rawb;                // a compressed chunk as it lies in the file (chunkb is then its uncompressed image)
  std::vector<BufferPtr> keep;   // page lists: alive until the stream has run
  // Sources of the asynchronous uploads (page tables, page lists, dictionary positions, a chunk fetched through the range
  // reader): pageable host memory the copy engine may read after the call that queued the copy has returned, so they live
  // as long as the column's work item -- until the host has synchronised with the column's stream.
  std::vector<std::shared_ptr<void>> host_keep;
  std::vector<PqPage> pages;     // the chunk's page headers (parsed here for a file behind a range reader)
  int64_t host_values = 0;    // non-null values when known on the host (no levels)
  int64_t uploaded = 0;       // file bytes sent to the GPU for this column
};

template <typename T>
BufferPtr upload(Context& ctx, hipStream_t stream, const std::vector<T>& v, ColumnWork& w) {
  auto b = make_device_buffer(v.size() * sizeof(T) + 16, ctx.device);
  if (!v.empty()) {
    auto keep = std::make_shared<std::vector<T>>(v);
    check_hip(hipMemcpyAsync(b->ptr, keep->data(), keep->size() * sizeof(T), hipMemcpyHostToDevice, stream), "upload page table");
    w.host_keep.push_back(keep);
  }
  return b;
}

}  // namespace

namespace {
// The stream of column `ci` of the `j`-th row group of a wave.  (ci + j * columns) % 12 put the same column of consecutive
// row groups on 12 / gcd(columns, 12) ... streams -- three columns: the string column of twenty row groups shared FOUR streams,
// and its pages (the slow ones: serial walks, serial inflates) queued five deep.  7 and 5 are coprime with 12: the columns of one
// row group get distinct streams, and one column visits all twelve over twelve row groups.
inline hipStream_t column_stream(Context& ctx, size_t ci, size_t j) { return ctx.aux[(ci * 7 + j * 5) % Context::kKernelStreams]; }

// the k-th upload event of this wave (created on first use, kept with the context)
hipEvent_t upload_event(Context& ctx, size_t k) {
  while (ctx.upload_events.size() <= k) {
    hipEvent_t e = nullptr;
    check_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    ctx.upload_events.push_back(e);
  }
  return ctx.upload_events[k];
}

// What phase A of a wave collects before anything is launched: the inflate jobs of every compressed page, and per column the
// closure that queues its decode kernels (they run behind the wave's ONE inflate launch).
struct WaveLaunch {
  std::vector<PqCodecJob> jobs;
  struct Col { hipStream_t stream; hipEvent_t uploaded; bool compressed; std::function<void()> launch; };
  std::vector<Col> cols;
  size_t n_events = 0;
};

struct RowGroupJob {   // one row group between its two phases
  int64_t rows = 0;
  std::vector<ColumnWork> work;
  BufferPtr scal;                 // Scalars per column, on the device
  std::vector<Scalars> hs;        // ... and read back
};

// ---- phase A: everything up to the Utf8 offsets, the columns side by side on the auxiliary streams (already forked) -----
void phase_a(Context& ctx, const PqFile& f, int row_group, RowGroupJob& job, size_t stream_shift, const std::vector<int>& sel, WaveLaunch& wave) {
  if (row_group < 0 || row_group >= (int)f.row_groups.size()) malformed("row group " + std::to_string(row_group) + " of " + std::to_string(f.row_groups.size()));
  const PqRowGroup& rg = f.row_groups[row_group];
  const int64_t rows = rg.num_rows;
  if (rows < 0 || rows >= (1ll << 31)) unsupported("row group of " + std::to_string(rows) + " rows");
  const size_t nc = sel.size();   // the columns this call decodes (work item ci <-> file column sel[ci])
  job.rows = rows;
  job.work.assign(nc, ColumnWork{});
  std::vector<ColumnWork>& work = job.work;
  job.hs.assign(nc + 1, Scalars{});
  Scalars* dscal = (Scalars*)job.scal->ptr;   // (allocated and zeroed by the caller, on ctx.stream, before the streams forked)
  for (size_t ci = 0; ci < nc; ++ci) {
    const hipStream_t cstream = column_stream(ctx, ci, stream_shift);   // the column's kernels
    const hipStream_t ustream = ctx.aux[Context::kCopyStream];          // every upload of the call (see engine.hpp)
    const PqColumnSchema& cs = f.columns[(size_t)sel[ci]];
    const PqColumnChunk& cc = rg.columns[(size_t)sel[ci]];
    ColumnWork& w = work[ci];
    w.schema = &cs; w.chunk = &cc; w.rows = rows;
    if (cs.repetition > 1) unsupported("repeated column '" + cs.name + "'");
    if (cs.logical_other) unsupported("logical type of column '" + cs.name + "'");
    if (cc.codec != 0 && cc.codec != 1) {
      static const char* kCodec[] = {"UNCOMPRESSED", "SNAPPY", "GZIP", "LZO", "BROTLI", "LZ4", "ZSTD", "LZ4_RAW"};
      unsupported(std::string("compression codec ") + (cc.codec >= 0 && cc.codec < 8 ? kCodec[cc.codec] : std::to_string(cc.codec).c_str()) +
                  " (column '" + cs.name + "'); UNCOMPRESSED and SNAPPY pages are decoded");
    }
    const bool compressed = cc.codec == 1;
    if (cc.num_values != rows) malformed("column '" + cs.name + "' holds " + std::to_string(cc.num_values) + " values for " + std::to_string(rows) + " rows");
    const bool is_string = cs.logical_string || cs.converted_type == 0;
    if (cs.converted_type > 0 && !((cs.converted_type == 17 && cs.type == PQ_INT32) || (cs.converted_type == 18 && cs.type == PQ_INT64)))
      unsupported("converted type " + std::to_string(cs.converted_type) + " of column '" + cs.name + "'");
    switch (cs.type) {
      case PQ_BOOLEAN: w.format = "b"; w.boolean = true; w.width = 1; break;
      case PQ_INT32: w.format = "i"; w.width = 4; break;
      case PQ_INT64: w.format = "l"; w.width = 8; break;
      case PQ_FLOAT: w.format = "f"; w.width = 4; break;
      case PQ_DOUBLE: w.format = "g"; w.width = 8; break;
      case PQ_BYTE_ARRAY:
        if (!is_string) unsupported("BYTE_ARRAY column '" + cs.name + "' without the String annotation");
        w.format = "u"; w.byte_array = true; break;
      default: unsupported("physical type " + std::to_string(cs.type) + " of column '" + cs.name + "'");
    }
    const int64_t first = rows ? cc.first_byte() : 0, csize = rows ? cc.total_compressed_size : 0;
    if (csize < 0 || first < 0 || csize > f.size - first) malformed("column chunk outside the file");
    if (csize >= (1ll << 32) - 64) unsupported("column chunk of " + std::to_string(csize) + " bytes");
    // the chunk's bytes: where they lie in the caller's memory, or fetched through the caller's range reader (exactly this
    // chunk: columns a call does not ask for are never read)
    const uint8_t* chunk_host = nullptr;
    if (csize > 0) {
      if (f.data) chunk_host = f.data + first;
      else {
        auto hb = make_host_buffer((size_t)csize + 16);
        if (f.read(f.read_user, first, csize, (uint8_t*)hb->ptr) != 0) malformed("the range reader failed on column '" + cs.name + "' (" + std::to_string(csize) + " bytes at " + std::to_string(first) + ")");
        chunk_host = (const uint8_t*)hb->ptr;
        w.host_keep.push_back(hb);
      }
    }
    const std::vector<PqPage>* chunk_pages = &cc.pages;
    if (!cc.pages_parsed) {   // (file behind a range reader: the headers are parsed from the bytes just fetched)
      PqColumnChunk tmp = cc;
      if (csize > 0) parquet_parse_pages(chunk_host, csize, tmp);
      w.pages = std::move(tmp.pages);
      chunk_pages = &w.pages;
    }
    BufferPtr raw = make_device_buffer((size_t)csize + 64, ctx.device);
    if (csize > 0) check_hip(hipMemcpyAsync(raw->ptr, chunk_host, (size_t)csize, hipMemcpyHostToDevice, ustream), "upload column chunk");
    check_hip(hipMemsetAsync((uint8_t*)raw->ptr + csize, 0, 64, ustream), "memset");
    w.uploaded = csize;
    if (!compressed) w.chunkb = raw; else w.rawb = raw;   // (compressed: chunkb = the uncompressed image, allocated below)

    const bool optional = cs.repetition == 1;
    w.has_levels = optional && cc.stat_null_count != 0;
    std::vector<PqPageDesc> descs;
    std::vector<int32_t> plain_list, dict_list, rle_list;
    std::vector<uint32_t> h_nonnull, h_base;
    uint32_t dict_at = 0, dict_len = 0, dict_count = 0;
    bool have_dict = false;
    int64_t row_at = 0;
    // Compressed chunk: every page is inflated into an uncompressed IMAGE of the chunk (pages back to back, 16-byte
    // aligned) and `rel` / the descriptors below point into that image; uncompressed chunk: the image is the chunk itself.
    std::vector<PqCodecJob> jobs;
    int64_t image_at = 0;
    for (const PqPage& pg : *chunk_pages) {
      if (!compressed && pg.compressed_size != pg.uncompressed_size) malformed("uncompressed page whose sizes differ");
      const int64_t raw_rel = pg.payload_at;   // (page offsets are relative to the chunk)
      if (raw_rel < 0 || pg.compressed_size < 0 || pg.compressed_size > csize - raw_rel) malformed("page outside its column chunk");
      if (pg.uncompressed_size < 0 || pg.uncompressed_size >= (1ll << 31)) malformed("page size");
      int64_t rel = raw_rel;
      int64_t v2_values_at = 0, v2_values_len = 0;   // compressed V2 page: where its values section lands in the image
      if (compressed) {
        if (pg.type == PQ_INDEX_PAGE) continue;
        rel = image_at;
        image_at += (pg.uncompressed_size + 15) / 16 * 16;
        if (image_at >= (1ll << 32) - 64) unsupported("column chunk of more than 4 GiB uncompressed");
        PqCodecJob j{};
        j.src_at = (uint32_t)raw_rel; j.src_len = (uint32_t)pg.compressed_size; j.dst_at = (uint32_t)rel; j.dst_len = (uint32_t)pg.uncompressed_size;
        j.codec = PQ_CODEC_SNAPPY; j.page = -1;
        if (pg.type == PQ_DATA_PAGE_V2) {
          // the level sections lie uncompressed in front of the values section, which is compressed unless the header says not
          const int64_t lv = pg.def_bytes + pg.rep_bytes;
          if (pg.def_bytes < 0 || pg.rep_bytes < 0 || lv > pg.compressed_size || lv > pg.uncompressed_size) malformed("level sections run past the page");
          if (lv > 0) {
            PqCodecJob l = j;
            l.src_len = (uint32_t)lv; l.dst_len = (uint32_t)lv; l.codec = PQ_CODEC_STORED;
            jobs.push_back(l);
          }
          // (the values section starts at an arbitrary byte of the image: the inflate kernel wants 16-byte aligned
          // destinations, so the values get their own aligned slot and the levels stay where they are)
          const int64_t vrel = image_at;
          image_at += (pg.uncompressed_size - lv + 15) / 16 * 16;
          j.src_at = (uint32_t)(raw_rel + lv); j.src_len = (uint32_t)(pg.compressed_size - lv);
          j.dst_at = (uint32_t)vrel; j.dst_len = (uint32_t)(pg.uncompressed_size - lv);
          if (!pg.v2_compressed) j.codec = PQ_CODEC_STORED;
          if (j.dst_len > 0 || j.src_len > 0) jobs.push_back(j);
          v2_values_at = vrel; v2_values_len = pg.uncompressed_size - lv;
        } else {
          if (pg.type == PQ_DATA_PAGE && cs.repetition == 1) {   // [4-byte length][levels][values]: told apart on the device
            j.page = -2;   // patched to the descriptor's index below
          }
          jobs.push_back(j);
        }
      }
      if (pg.type == PQ_DICTIONARY_PAGE) {
        if (pg.encoding != PQ_PLAIN && pg.encoding != PQ_PLAIN_DICTIONARY) unsupported("dictionary page encoding " + std::to_string(pg.encoding));
        if (pg.num_values < 0 || pg.num_values >= (1ll << 31)) malformed("dictionary size");
        dict_at = (uint32_t)rel; dict_len = (uint32_t)pg.uncompressed_size; dict_count = (uint32_t)pg.num_values; have_dict = true;
        continue;
      }
      if (pg.type == PQ_INDEX_PAGE) continue;
      if (pg.type != PQ_DATA_PAGE && pg.type != PQ_DATA_PAGE_V2) unsupported("page type " + std::to_string(pg.type));
      PqPageDesc d{};
      if (pg.num_values < 0 || pg.num_values > rows - row_at) malformed("page of column '" + cs.name + "' holds more rows than the row group has left");
      d.num_rows = (uint32_t)pg.num_values; d.first_row = row_at;
      int64_t at = rel, left = pg.uncompressed_size;
      if (pg.type == PQ_DATA_PAGE) {
        if (optional) {   // [4-byte length][RLE hybrid levels]
          if (pg.def_encoding != PQ_RLE) unsupported("definition level encoding " + std::to_string(pg.def_encoding));
          if (left < 4) malformed("page too short for its definition levels");
          if (compressed) {
            // the length prefix is inside the compressed payload: the inflate kernel reads it and fills in the descriptor
            // (levels_at / levels_len / values_at / values_len) before the decode kernels, queued behind it, read it
            for (size_t k = jobs.size(); k-- > 0;) if (jobs[k].page == -2) { jobs[k].page = (int32_t)descs.size(); jobs[k].flags = w.has_levels ? PQ_JOB_KEEP_LEVELS : 0; break; }
            d.levels_at = (uint32_t)at; d.levels_len = w.has_levels ? 1 : 0;   // (placeholder: "has levels", see the check below)
            at = rel; left = 0;
          } else {
            uint32_t l; memcpy(&l, chunk_host + raw_rel, 4);
            if ((int64_t)l + 4 > left) malformed("definition levels run past the page");
            d.levels_at = (uint32_t)(at + 4); d.levels_len = l;
            at += 4 + l; left -= 4 + l;
          }
        }
      } else {
        if (pg.rep_bytes != 0) unsupported("repetition levels");
        if (pg.def_bytes < 0 || pg.def_bytes > left) malformed("definition levels run past the page");
        d.levels_at = (uint32_t)at; d.levels_len = (uint32_t)pg.def_bytes;
        if (compressed) { at = v2_values_at; left = v2_values_len; }
        else { at += pg.def_bytes; left -= pg.def_bytes; }
        if (!optional && pg.def_bytes != 0) malformed("definition levels on a required column");
      }
      if (optional && w.has_levels && d.levels_len == 0 && d.num_rows > 0) malformed("optional column without definition levels");
      if (!w.has_levels) d.levels_len = 0;
      d.values_at = (uint32_t)at; d.values_len = (uint32_t)left;
      const int page_index = (int)descs.size();
      if (pg.encoding == PQ_PLAIN) plain_list.push_back(page_index);
      else if (pg.encoding == PQ_RLE_DICTIONARY || pg.encoding == PQ_PLAIN_DICTIONARY) {
        if (!have_dict) malformed("dictionary-encoded page without a dictionary page in front of it");
        dict_list.push_back(page_index);
      } else if (pg.encoding == PQ_RLE && w.boolean) rle_list.push_back(page_index);
      else unsupported("encoding " + std::to_string(pg.encoding) + " (column '" + cs.name + "')");
      h_nonnull.push_back(d.num_rows); h_base.push_back((uint32_t)row_at);
      row_at += pg.num_values;
      descs.push_back(d);
    }
    if (row_at != rows) malformed("pages of column '" + cs.name + "' hold " + std::to_string(row_at) + " rows, the row group has " + std::to_string(rows));
    const int n_pages = (int)descs.size();
    w.pages_dev = upload(ctx, ustream, descs, w);
    w.nonnull = w.has_levels ? make_device_buffer((size_t)n_pages * 4 + 16, ctx.device) : upload(ctx, ustream, h_nonnull, w);
    w.value_base = w.has_levels ? make_device_buffer((size_t)n_pages * 4 + 16, ctx.device) : upload(ctx, ustream, h_base, w);
    w.host_values = rows;
    // (every upload of the column is queued before its first kernel: the column's stream then waits ONCE, on one event)
    auto plain_dev = upload(ctx, ustream, plain_list, w), dict_dev = upload(ctx, ustream, dict_list, w), rle_dev = upload(ctx, ustream, rle_list, w);
    w.keep.push_back(plain_dev); w.keep.push_back(dict_dev); w.keep.push_back(rle_dev);
    if (w.byte_array && !dict_list.empty() && !compressed) {
      // The dictionary page is ONE page: on the GPU a single workgroup would walk its length prefixes (5 ms for 1 MB of
      // ragged strings).  The host has the bytes and walks them in ~0.2 ms while it prepares the launches.
      std::vector<uint32_t> h_src(dict_count), h_len(dict_count);
      const uint8_t* dp = chunk_host + dict_at;
      uint64_t pos = 0;
      for (uint32_t i = 0; i < dict_count; ++i) {
        if (pos + 4 > dict_len) malformed("dictionary page of column '" + cs.name + "' ends inside an entry");
        uint32_t l; memcpy(&l, dp + pos, 4);
        if ((uint64_t)l > dict_len - pos - 4) malformed("dictionary entry of column '" + cs.name + "' runs past its page");
        h_src[i] = dict_at + (uint32_t)pos + 4; h_len[i] = l;
        pos += 4 + (uint64_t)l;
      }
      w.dict_src = upload(ctx, ustream, h_src, w);
      w.dict_len = upload(ctx, ustream, h_len, w);
    }
    const hipEvent_t ev_uploaded = upload_event(ctx, wave.n_events++);
    check_hip(hipEventRecord(ev_uploaded, ustream), "hipEventRecord(upload)");
    if (compressed) {   // every page is inflated into the image by the wave's one inflate launch (parquet_read_row_groups)
      w.chunkb = make_device_buffer((size_t)image_at + 64, ctx.device);
      check_hip(hipMemsetAsync((uint8_t*)w.chunkb->ptr + image_at, 0, 64, ustream), "memset");
      for (PqCodecJob& j : jobs) {
        j.raw = (const uint8_t*)w.rawb->ptr; j.image = (uint8_t*)w.chunkb->ptr; j.pages = (PqPageDesc*)w.pages_dev->ptr; j.err = &dscal[ci].err;
        wave.jobs.push_back(j);
      }
    }

    PqDecodeParams p{};
    p.chunk = (const uint8_t*)w.chunkb->ptr; p.pages = (const PqPageDesc*)w.pages_dev->ptr; p.n_pages = n_pages; p.width = w.width;
    p.nonnull = (uint32_t*)w.nonnull->ptr; p.value_base = (uint32_t*)w.value_base->ptr;
    p.total_values = &dscal[ci].total_values; p.err = &dscal[ci].err;
    p.dict_at = dict_at; p.dict_len = dict_len; p.dict_count = dict_count;
    if (rows == 0 || n_pages == 0) continue;   // (an empty row group: phase B builds empty columns)
    // the column's decode kernels, queued once the wave's uploads and inflate launch are in place
    wave.cols.push_back(WaveLaunch::Col{cstream, ev_uploaded, compressed, [=, &ctx, &w]() mutable {
    if (w.has_levels) {
      w.valid8 = make_device_buffer((size_t)rows + 64, ctx.device);
      w.row_val = make_device_buffer((size_t)rows * 4 + 64, ctx.device);
      p.valid8 = (uint8_t*)w.valid8->ptr; p.row_val = (int32_t*)w.row_val->ptr;
      check_hip(pq_launch_levels(p, cstream), "launch pq_levels_kernel");
      check_hip(pq_launch_page_scan(p, cstream), "launch pq_page_scan_kernel");
      check_hip(pq_launch_rowval(p, cstream), "launch pq_rowval_kernel");
    }
    // dense values: at most `rows` of them
    if (w.byte_array) {
      w.vsrc = make_device_buffer((size_t)rows * 4 + 64, ctx.device);
      w.vlen = make_device_buffer((size_t)rows * 4 + 64, ctx.device);
      p.vsrc = (uint32_t*)w.vsrc->ptr; p.vlen = (uint32_t*)w.vlen->ptr;
      if (!dict_list.empty()) {
        if (compressed) {   // the page is only readable after inflation: one workgroup walks it on the device
          w.dict_src = make_device_buffer((size_t)dict_count * 4 + 16, ctx.device);
          w.dict_len = make_device_buffer((size_t)dict_count * 4 + 16, ctx.device);
          PqDecodeParams dpw = p;
          dpw.walk_dictionary = 1; dpw.dict_src = (uint32_t*)w.dict_src->ptr; dpw.dict_len_out = (uint32_t*)w.dict_len->ptr;
          if (dict_count > 0) check_hip(pq_launch_ba_walk(dpw, 1, cstream), "launch pq_ba_walk_kernel (dictionary)");
        }
        p.dict_src = (uint32_t*)w.dict_src->ptr; p.dict_len_out = (uint32_t*)w.dict_len->ptr;
        p.page_list = (const int32_t*)dict_dev->ptr;
        check_hip(pq_launch_dict_ba(p, (int)dict_list.size(), cstream), "launch pq_dict_ba_kernel");
      }
      if (!plain_list.empty()) {
        p.page_list = (const int32_t*)plain_dev->ptr;
        check_hip(pq_launch_ba_walk(p, (int)plain_list.size(), cstream), "launch pq_ba_walk_kernel");
      }
      // offsets = exclusive scan of the row lengths
      w.offsets = make_device_buffer((size_t)(rows + 2) * 4, ctx.device);
      const int64_t n_blocks = (rows + PQ_SCAN_ROWS_HOST - 1) / PQ_SCAN_ROWS_HOST;
      w.block_sums = make_device_buffer((size_t)n_blocks * 8 + 16, ctx.device);
      PqRowParams r{};
      r.n_rows = rows; r.row_val = w.has_levels ? (const int32_t*)w.row_val->ptr : nullptr;
      r.vlen = (const uint32_t*)w.vlen->ptr; r.out = w.offsets->ptr;
      r.block_sums = (unsigned long long*)w.block_sums->ptr; r.n_blocks = n_blocks; r.total_bytes = &dscal[ci].total_bytes;
      check_hip(pq_launch_rowlen(r, cstream), "launch pq_rowlen kernels");
    } else {
      if (dict_count > 0 && (uint64_t)dict_count * w.width > dict_len) malformed("dictionary page shorter than its entries");
      w.dense = make_device_buffer((size_t)rows * w.width + 64, ctx.device);
      p.dense = (uint8_t*)w.dense->ptr;
      if (!plain_list.empty()) {
        p.page_list = (const int32_t*)plain_dev->ptr;
        if (w.boolean) check_hip(pq_launch_bool(p, (int)plain_list.size(), cstream), "launch pq_bool_kernel");
        else check_hip(pq_launch_plain_copy(p, (int)plain_list.size(), cstream), "launch pq_plain_copy_kernel");
      }
      if (!rle_list.empty()) {
        p.page_list = (const int32_t*)rle_dev->ptr;
        check_hip(pq_launch_bool_rle(p, (int)rle_list.size(), cstream), "launch pq_bool_rle_kernel");
      }
      if (!dict_list.empty()) {
        if (w.boolean) unsupported("dictionary-encoded BOOLEAN column");
        p.page_list = (const int32_t*)dict_dev->ptr;
        check_hip(pq_launch_dict_fixed(p, (int)dict_list.size(), cstream), "launch pq_dict_fixed_kernel");
      }
    }
    }});
  }
}

// ---- phase B: per-row outputs (sizes are on the host now) -----------------------------------------------------------------
Batch phase_b(Context& ctx, RowGroupJob& job, size_t stream_shift) {
  const int64_t rows = job.rows;
  const size_t nc = job.work.size();
  std::vector<ColumnWork>& work = job.work;
  const std::vector<Scalars>& hs = job.hs;
  Batch out;
  out.on_device = true; out.device_id = ctx.device; out.nrows = rows;
  const int grid = ctx.num_cus * 8;
  for (size_t ci = 0; ci < nc; ++ci) {
    const hipStream_t cstream = column_stream(ctx, ci, stream_shift);
    ColumnWork& w = work[ci];
    if (hs[ci].err) malformed(std::string(hs[ci].err == PQ_ERR_CODEC ? "compressed pages" : hs[ci].err == PQ_ERR_LEVELS ? "definition levels" : "values") + " of column '" + w.schema->name + "' are malformed");
    Column o;
    o.name = w.schema->name; o.format = w.format; o.nullable = w.schema->repetition == 1; o.length = rows; o.offset = 0;
    parse_arrow_format(w.format, &o.type, &o.width);
    const int64_t nonnull = w.has_levels && rows > 0 ? (int64_t)hs[ci].total_values : rows;
    o.null_count = rows - nonnull;
    const int32_t* row_val = (w.has_levels && o.null_count > 0) ? (const int32_t*)w.row_val->ptr : nullptr;
    if (rows == 0) {
      auto vb = make_device_buffer(64, ctx.device);
      check_hip(hipMemsetAsync(vb->ptr, 0, 64, cstream), "memset");
      o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
      if (w.byte_array) { auto db = make_device_buffer(64, ctx.device); o.data = (const uint8_t*)db->ptr; o.owned.push_back(db); o.data_bytes = 0; }
      out.cols.push_back(std::move(o));
      continue;
    }
    if (o.null_count > 0) {   // validity bitmap from the one-byte-per-row levels
      auto vb = make_device_buffer((size_t)((rows + 63) / 64) * 8 + 16, ctx.device);
      PqRowParams r{};
      r.n_rows = rows; r.dense = (const uint8_t*)w.valid8->ptr; r.out = vb->ptr;
      check_hip(pq_launch_pack_bits(r, grid, cstream), "launch pq_pack_bits_kernel");
      o.validity = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
    }
    if (w.byte_array) {
      const unsigned long long total = hs[ci].total_bytes;
      if (total >= (1ull << 31)) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "offset overflow: column '" + o.name + "' of this row group holds " + std::to_string(total) + " bytes; Utf8 offsets are int32"};
      auto db = make_device_buffer((size_t)total + 64, ctx.device);
      PqRowParams r{};
      r.n_rows = rows; r.row_val = row_val; r.vsrc = (const uint32_t*)w.vsrc->ptr; r.vlen = (const uint32_t*)w.vlen->ptr;
      r.chunk = (const uint8_t*)w.chunkb->ptr; r.offsets = w.offsets->ptr; r.data_out = (uint8_t*)db->ptr;
      check_hip(pq_launch_utf8_copy(r, grid, cstream), "launch pq_utf8_copy_kernel");
      o.values = (const uint8_t*)w.offsets->ptr; o.owned.push_back(w.offsets);
      o.data = (const uint8_t*)db->ptr; o.owned.push_back(db); o.data_bytes = (int64_t)total;
    } else if (w.boolean) {
      auto vb = make_device_buffer((size_t)((rows + 63) / 64) * 8 + 16, ctx.device);
      PqRowParams r{};
      r.n_rows = rows; r.row_val = row_val; r.dense = (const uint8_t*)w.dense->ptr; r.out = vb->ptr;
      check_hip(pq_launch_pack_bits(r, grid, cstream), "launch pq_pack_bits_kernel");
      o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
    } else if (row_val) {
      auto vb = make_device_buffer((size_t)rows * w.width + 64, ctx.device);
      PqRowParams r{};
      r.n_rows = rows; r.row_val = row_val; r.dense = (const uint8_t*)w.dense->ptr; r.out = vb->ptr;
      check_hip(pq_launch_gather_fixed(r, w.width, grid, cstream), "launch pq_gather_fixed_kernel");
      o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
    } else {   // no nulls: the dense values ARE the column
      o.values = (const uint8_t*)w.dense->ptr; o.owned.push_back(w.dense);
    }
    out.cols.push_back(std::move(o));
  }
  return out;
}
}  // namespace

// Row groups [first, first + count): phase A of all of them is in flight before the first size is read back, so the two
// host synchronisations are paid once per call (per wave of about 1 GiB of column chunks), not once per row group, and the
// upload of one row group overlaps the decode of the previous ones.  Only the columns in `columns` are fetched and decoded.
std::vector<Batch> parquet_read_row_groups(Context& ctx, const PqFile& f, int first, int count, const int32_t* columns, int n_columns) {
  if (first < 0 || count < 0 || first + count > (int)f.row_groups.size())
    malformed("row groups [" + std::to_string(first) + ", " + std::to_string(first + count) + ") of " + std::to_string(f.row_groups.size()));
  std::vector<int> sel;
  if (!columns || n_columns < 0) { for (size_t i = 0; i < f.columns.size(); ++i) sel.push_back((int)i); }
  else {
    for (int k = 0; k < n_columns; ++k) {
      if (columns[k] < 0 || columns[k] >= (int)f.columns.size())
        throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "parquet: column index " + std::to_string(columns[k]) + " of " + std::to_string(f.columns.size())};
      sel.push_back(columns[k]);
    }
  }
  ctx.stats = chq_call_stats{};
  std::vector<Batch> outs;
  outs.reserve((size_t)count);
  int next = first;
  while (next < first + count) {
    int wave_end = next;
    int64_t wave_bytes = 0;
    while (wave_end < first + count) {   // a wave: about 1 GiB of column chunks resident at once (at least one row group)
      int64_t b = 0;
      for (int ci : sel) b += f.row_groups[(size_t)wave_end].columns[(size_t)ci].total_compressed_size;
      if (wave_end > next && wave_bytes + b > ((int64_t)1 << 30)) break;
      wave_bytes += b; ++wave_end;
    }
    std::vector<RowGroupJob> jobs((size_t)(wave_end - next));
    DrainOnUnwind drain{ctx};   // (declared after `jobs`: runs before their buffers are released)
    for (RowGroupJob& job : jobs) {
      job.scal = make_device_buffer(sizeof(Scalars) * (sel.size() + 1), ctx.device);
      check_hip(hipMemsetAsync(job.scal->ptr, 0, sizeof(Scalars) * (sel.size() + 1), ctx.stream), "memset");
    }
    fork_streams(ctx);
    WaveLaunch wave;
    // The compressed pages of the wave are inflated in up to three PARTS of consecutive row groups, each behind the uploads
    // of its own row groups (the GPU would otherwise idle until the wave's last upload); the decode closures of a part's
    // columns wait for that part's event.
    //   A snappy page of several 64 KiB blocks is inflated block by block (parquet_codec.hip): an INDEX job walks its element
    // chain and notes where each block starts, one BLOCK job per block moves the bytes, a FINISH job patches the page
    // descriptor (and redoes the page with one wave in the case the format allows and no compressor produces: blocks that
    // depend on each other): a CHAIN of three launches in stream order.
    //   The INDEX walk is serial per page and its time grows with the page's compressed size, so a few large pages (a 1 MiB
    // dictionary page of short strings: 13 ms) would hold back the BLOCK launch of all the others: pages of 512 KiB and more
    // of compressed bytes are collected over the whole wave and form ONE more chain on a stream of its own, behind the last
    // upload.  (Their walks all have to run side by side -- in one launch: a chain per part put them behind each other on
    // the few hardware queues the streams share, 33 -> 51 ms for the 20-row-group sample.)
    struct Chain { std::vector<PqCodecJob> seg, index, blocks, finish; size_t words = 0; };   // (words: of the index table)
    std::vector<std::shared_ptr<void>> host_keep;
    std::vector<BufferPtr> device_keep;
    std::vector<hipEvent_t> col_inflated;      // per entry of wave.cols
    const hipStream_t ustream = ctx.aux[Context::kCopyStream];
    auto launch_chain = [&](const Chain& ch, hipStream_t st) -> hipEvent_t {
      auto keep = std::make_shared<std::vector<PqCodecJob>>();
      for (const std::vector<PqCodecJob>* v : {&ch.seg, &ch.index, &ch.blocks, &ch.finish}) keep->insert(keep->end(), v->begin(), v->end());
      const size_t index_words = ch.words;
      if (index_words) {   // (the jobs carry offsets into the table until it exists)
        BufferPtr index_dev = make_device_buffer(index_words * sizeof(uint32_t) + 16, ctx.device);
        check_hip(hipMemsetAsync(index_dev->ptr, 0, index_words * sizeof(uint32_t), ustream), "memset");
        for (PqCodecJob& j : *keep) if (j.codec >= PQ_CODEC_SNAPPY_INDEX) j.index = (uint32_t*)((uint8_t*)index_dev->ptr + (uintptr_t)j.index);
        device_keep.push_back(index_dev);
      }
      BufferPtr jobs_dev = make_device_buffer(keep->size() * sizeof(PqCodecJob) + 16, ctx.device);
      check_hip(hipMemcpyAsync(jobs_dev->ptr, keep->data(), keep->size() * sizeof(PqCodecJob), hipMemcpyHostToDevice, ustream), "upload inflate jobs");
      host_keep.push_back(keep); device_keep.push_back(jobs_dev);
      // the chain starts behind everything uploaded so far: the pages, the job list, the cleared index table
      const hipEvent_t ev_ready = upload_event(ctx, wave.n_events++);
      check_hip(hipEventRecord(ev_ready, ustream), "hipEventRecord(jobs)");
      check_hip(hipStreamWaitEvent(st, ev_ready, 0), "hipStreamWaitEvent(jobs)");
      const PqCodecJob* at = (const PqCodecJob*)jobs_dev->ptr;
      const size_t counts[4] = {ch.seg.size(), ch.index.size(), ch.blocks.size(), ch.finish.size()};
      for (int k = 0; k < 4; ++k) {
        PqCodecParams cp{};
        cp.jobs = at; cp.n_jobs = (int32_t)counts[k];
        check_hip(k < 2 ? pq_launch_inflate_index(cp, st) : pq_launch_inflate(cp, st), "launch pq_inflate_kernel");
        at += counts[k];
      }
      const hipEvent_t ev_done = upload_event(ctx, wave.n_events++);
      check_hip(hipEventRecord(ev_done, st), "hipEventRecord(inflate)");
      return ev_done;
    };
    auto add_page = [&](Chain& ch, const PqCodecJob& j, bool segments) {   // an indexed page: its INDEX (or SEG + RESOLVE), BLOCK and FINISH jobs
      const uint32_t nblk = (j.dst_len + 65535u) / 65536u;
      PqCodecJob a = j;
      a.index = (uint32_t*)(uintptr_t)(ch.words * sizeof(uint32_t));   // (an offset until the table exists)
      ch.words += nblk + 2 + 4 * PQ_SNAPPY_SEGMENTS;
      PqCodecJob c = a; c.codec = PQ_CODEC_SNAPPY_FINISH;
      ch.finish.push_back(c);
      a.page = -1; a.flags = ctx.opt_snappy_blocks == 2 ? PQ_JOB_FORCE_FALLBACK : 0u;
      a.codec = PQ_CODEC_SNAPPY_BLOCK;
      for (uint32_t k = 0; k < nblk; ++k) { a.block = k; ch.blocks.push_back(a); }
      const uint32_t n_seg = segments ? std::min<uint32_t>(PQ_SNAPPY_SEGMENTS, j.src_len >> 15) : 1u;
      if (n_seg >= 2) {   // (the walk of the page: one wave per segment of its input, twice)
        for (uint32_t k = 0; k < n_seg; ++k) { a.block = k | (n_seg << 16); a.codec = PQ_CODEC_SNAPPY_SEG; ch.seg.push_back(a); a.codec = PQ_CODEC_SNAPPY_RESOLVE; ch.index.push_back(a); }
      } else {
        a.codec = PQ_CODEC_SNAPPY_INDEX; a.block = 0;
        ch.index.push_back(a);
      }
    };
    Chain large;
    const size_t n_parts = jobs.size() >= 12 ? 3 : (jobs.size() >= 6 ? 2 : 1);
    size_t job0 = 0, part = 0;
    for (size_t j = 0; j < jobs.size(); ++j) {
      phase_a(ctx, f, next + (int)j, jobs[j], j, sel, wave);
      if ((j + 1) * n_parts / jobs.size() == part) continue;     // (this part goes on)
      Chain ch;
      for (size_t q = job0; q < wave.jobs.size(); ++q) {
        const PqCodecJob& pj = wave.jobs[q];
        if (ctx.opt_snappy_blocks == 0 || pj.codec != PQ_CODEC_SNAPPY || pj.dst_len < 3u * 65536u) ch.blocks.push_back(pj);
        else if (pj.src_len >= (512u << 10)) add_page(large, pj, ctx.opt_snappy_blocks != 3);
        else add_page(ch, pj, ctx.opt_snappy_blocks != 3);
      }
      hipEvent_t ev = nullptr;
      if (!ch.blocks.empty()) ev = launch_chain(ch, ctx.aux[(part % 2) * 2]);   // (two streams: a part's INDEX launch next to the BLOCK launch before it)
      col_inflated.resize(wave.cols.size(), ev);
      job0 = wave.jobs.size(); ++part;
    }
    col_inflated.resize(wave.cols.size(), nullptr);
    const hipEvent_t ev_large = large.index.empty() ? nullptr : launch_chain(large, ctx.aux[1]);
    for (size_t k = 0; k < wave.cols.size(); ++k) {
      WaveLaunch::Col& c = wave.cols[k];
      check_hip(hipStreamWaitEvent(c.stream, c.compressed && col_inflated[k] ? col_inflated[k] : c.uploaded, 0), "hipStreamWaitEvent(upload)");
      if (c.compressed && ev_large) check_hip(hipStreamWaitEvent(c.stream, ev_large, 0), "hipStreamWaitEvent(inflate)");
      c.launch();
    }
    join_streams(ctx);
    for (RowGroupJob& job : jobs)
      if (!job.work.empty()) check_hip(hipMemcpyAsync(job.hs.data(), job.scal->ptr, sizeof(Scalars) * job.work.size(), hipMemcpyDeviceToHost, ctx.stream), "read back");
    check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
    fork_streams(ctx);
    for (size_t j = 0; j < jobs.size(); ++j) {
      outs.push_back(phase_b(ctx, jobs[j], j));
      // (chq_call_stats of a scan: rows decoded, file bytes sent to the GPU, Arrow bytes produced)
      ctx.stats.rows_in += jobs[j].rows; ctx.stats.rows_out += jobs[j].rows;
      for (const ColumnWork& w : jobs[j].work) ctx.stats.bytes_read_alg += w.uploaded;
    }
    join_streams(ctx);
    check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");   // the temporaries of `jobs` are released here
    next = wave_end;
  }
  for (const Batch& b : outs)
    for (const Column& c : b.cols) {
      if (c.type == T_UTF8) ctx.stats.bytes_written_alg += (c.length + 1) * 4 + std::max<int64_t>(0, c.data_bytes);
      else if (c.type == T_BOOL) ctx.stats.bytes_written_alg += (c.length + 7) / 8;
      else ctx.stats.bytes_written_alg += c.length * c.width;
      if (c.validity) ctx.stats.bytes_written_alg += (c.length + 7) / 8;
    }
  return outs;
}

Batch parquet_read_row_group(Context& ctx, const PqFile& f, int row_group) {
  if (row_group < 0 || row_group >= (int)f.row_groups.size()) malformed("row group " + std::to_string(row_group) + " of " + std::to_string(f.row_groups.size()));
  std::vector<Batch> one = parquet_read_row_groups(ctx, f, row_group, 1);
  return std::move(one[0]);
}

}  // namespace chq
