// ipc.cpp -- Arrow IPC *stream* encoding of one record batch with the message body in HBM.
//
// The reference serialises a batch that leaves the process with arrow-rs' ipc::writer::StreamWriter (schema message, one
// RecordBatch message, end-of-stream; src/handlers/message_handler/messages/exchange.rs:145-197) and reads it back with
// ipc::reader::StreamReader (exchange.rs:247-276).  Here the same wire format is produced for a batch that lives on the
// GPU: the two metadata flatbuffers (Schema, RecordBatch: a few hundred bytes) are built on the host, the message BODY --
// every Arrow buffer, rebased to offset 0, 64-byte aligned, back to back -- is assembled in ONE HBM allocation by device
// copies / small kernels, so a peer can receive it with a single RCCL send (or one D2H copy when it has to cross TCP).
// Decoding is the inverse: the body is moved to its destination with one copy and the columns are views into it.
//
// Format (arrow/format/{Message,Schema}.fbs, metadata version V5, little endian); flatbuffers are written and read by the
// few dozen lines below -- no dependency.  Dictionaries, compression and nested types are outside the path's scope
// (CHQ_ERR_NOT_SUPPORTED); the column kinds are those of engine.hpp: Boolean, Int8..UInt64, Float16/32/64, Utf8 and the
// fixed-width "opaque" types (date / time / timestamp / duration / decimal / fixed-size binary).
#include <cstring>

#include "engine.hpp"

namespace chq {

hipError_t launch_rebase_offsets(const int32_t* in, int32_t* out, int64_t n_plus_1, hipStream_t stream);
hipError_t launch_bit_shift_copy(const uint8_t* in, int64_t bit_offset, int64_t nbits, uint32_t* out, hipStream_t stream);
hipError_t launch_count_bits(const uint8_t* in, int64_t bit_offset, int64_t nbits, unsigned long long* out, hipStream_t stream);
hipError_t launch_validate_offsets(const int32_t* offs, int64_t n, int64_t data_len, uint32_t* flag, hipStream_t stream);

namespace {

// ---------------------------------------------------------------------------------------------------- flatbuffers: write
// Built back to front like the reference implementation: children first (higher addresses), `pos` = distance from the END.
class FlatWriter {
 public:
  FlatWriter() : buf_(1024), head_(1024) {}
  size_t size() const { return buf_.size() - head_; }
  void pad(size_t n) { reserve(n); head_ -= n; memset(&buf_[head_], 0, n); }
  void align(size_t a) { if (a > minalign_) minalign_ = a; pad((~size() + 1) & (a - 1)); }
  void prealign(size_t len, size_t a) { if (a > minalign_) minalign_ = a; pad((~(size() + len) + 1) & (a - 1)); }
  template <typename T> void push(T v) { align(sizeof(T)); reserve(sizeof(T)); head_ -= sizeof(T); memcpy(&buf_[head_], &v, sizeof(T)); }
  uint32_t string(const std::string& s) {
    prealign(s.size() + 1, 4);
    pad(1);
    reserve(s.size()); head_ -= s.size(); memcpy(&buf_[head_], s.data(), s.size());
    push<uint32_t>((uint32_t)s.size());
    return (uint32_t)size();
  }
  uint32_t offset_vector(const std::vector<uint32_t>& targets) {
    prealign(targets.size() * 4, 4);
    for (size_t i = targets.size(); i-- > 0;) { align(4); push<uint32_t>((uint32_t)(size() - targets[i] + 4)); }
    push<uint32_t>((uint32_t)targets.size());
    return (uint32_t)size();
  }
  uint32_t pair_vector(const std::vector<std::pair<int64_t, int64_t>>& v) {   // [FieldNode] / [Buffer]: structs of two longs
    prealign(v.size() * 16, 4);
    prealign(v.size() * 16, 8);
    for (size_t i = v.size(); i-- > 0;) { push<int64_t>(v[i].second); push<int64_t>(v[i].first); }
    push<uint32_t>((uint32_t)v.size());
    return (uint32_t)size();
  }
  void start_table() { fields_.clear(); }
  template <typename T> void scalar(int id, T v) { push<T>(v); fields_.push_back({(uint32_t)size(), id}); }
  void ref(int id, uint32_t target) {
    if (!target) return;
    align(4);
    push<uint32_t>((uint32_t)(size() - target + 4));
    fields_.push_back({(uint32_t)size(), id});
  }
  uint32_t end_table(uint32_t object_start) {
    align(4);
    push<int32_t>(0);   // soffset to the vtable, patched below
    const uint32_t table = (uint32_t)size();
    int maxid = -1;
    for (auto& f : fields_) maxid = std::max(maxid, f.id);
    for (int id = maxid; id >= 0; --id) {
      uint16_t off = 0;
      for (auto& f : fields_) if (f.id == id) off = (uint16_t)(table - f.pos);
      push<uint16_t>(off);
    }
    push<uint16_t>((uint16_t)(table - object_start));
    push<uint16_t>((uint16_t)((maxid + 1 + 2) * 2));
    const int32_t so = (int32_t)(size() - table);
    memcpy(&buf_[buf_.size() - table], &so, 4);
    return table;
  }
  uint32_t mark() const { return (uint32_t)size(); }
  std::vector<uint8_t> finish(uint32_t root) {
    prealign(4, std::max<size_t>(minalign_, 8));
    push<uint32_t>((uint32_t)(size() - root + 4));
    return std::vector<uint8_t>(buf_.begin() + (long)head_, buf_.end());
  }

 private:
  void reserve(size_t n) {
    if (head_ >= n) return;
    const size_t used = size(), grow = std::max(buf_.size(), n + 64);
    std::vector<uint8_t> nb(buf_.size() + grow);
    memcpy(&nb[nb.size() - used], &buf_[head_], used);
    head_ = nb.size() - used;
    buf_.swap(nb);
  }
  struct Loc { uint32_t pos; int id; };
  std::vector<uint8_t> buf_;
  size_t head_;
  size_t minalign_ = 1;
  std::vector<Loc> fields_;
};

// ----------------------------------------------------------------------------------------------------- flatbuffers: read
struct FlatReader {
  const uint8_t* base; size_t len;
  [[noreturn]] void bad() const { throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "malformed Arrow IPC metadata"}; }
  template <typename T> T rd(const uint8_t* p) const { if (p < base || p + sizeof(T) > base + len) bad(); T v; memcpy(&v, p, sizeof(T)); return v; }
  const uint8_t* root() const { return base + rd<uint32_t>(base); }
  const uint8_t* field(const uint8_t* t, int id) const {
    const uint8_t* vt = t - rd<int32_t>(t);
    const uint16_t vts = rd<uint16_t>(vt);
    if (4 + 2 * id + 2 > vts) return nullptr;
    const uint16_t off = rd<uint16_t>(vt + 4 + 2 * id);
    return off ? t + off : nullptr;
  }
  template <typename T> T scalar(const uint8_t* t, int id, T def) const { const uint8_t* p = field(t, id); return p ? rd<T>(p) : def; }
  const uint8_t* indirect(const uint8_t* t, int id) const { const uint8_t* p = field(t, id); return p ? p + rd<uint32_t>(p) : nullptr; }
  uint32_t vec_len(const uint8_t* v) const { return rd<uint32_t>(v); }
  std::string str(const uint8_t* t, int id) const {
    const uint8_t* s = indirect(t, id);
    if (!s) return "";
    const uint32_t n = rd<uint32_t>(s);
    if (s + 4 + n > base + len) bad();
    return std::string((const char*)s + 4, n);
  }
};

// ------------------------------------------------------------------------------------------------------- type <-> format
enum TypeTag : uint8_t { TY_Int = 2, TY_FloatingPoint = 3, TY_Utf8 = 5, TY_Bool = 6, TY_Decimal = 7, TY_Date = 8, TY_Time = 9,
                         TY_Timestamp = 10, TY_FixedSizeBinary = 15, TY_Duration = 18 };

int time_unit(char c) { return c == 's' ? 0 : c == 'm' ? 1 : c == 'u' ? 2 : 3; }
char unit_char(int u) { return "smun"[u & 3]; }

uint32_t write_type(FlatWriter& w, const std::string& f, uint8_t* tag) {
  const uint32_t start = w.mark();
  auto int_type = [&](int bits, bool sign) { w.start_table(); w.scalar<int32_t>(0, bits); w.scalar<uint8_t>(1, sign ? 1 : 0); *tag = TY_Int; return w.end_table(start); };
  auto fp = [&](int16_t prec) { w.start_table(); w.scalar<int16_t>(0, prec); *tag = TY_FloatingPoint; return w.end_table(start); };
  if (f == "b") { w.start_table(); *tag = TY_Bool; return w.end_table(start); }
  if (f == "u") { w.start_table(); *tag = TY_Utf8; return w.end_table(start); }
  if (f == "c") return int_type(8, true);   if (f == "C") return int_type(8, false);
  if (f == "s") return int_type(16, true);  if (f == "S") return int_type(16, false);
  if (f == "i") return int_type(32, true);  if (f == "I") return int_type(32, false);
  if (f == "l") return int_type(64, true);  if (f == "L") return int_type(64, false);
  if (f == "e") return fp(0); if (f == "f") return fp(1); if (f == "g") return fp(2);
  if (f == "tdD" || f == "tdm") { w.start_table(); w.scalar<int16_t>(0, f == "tdD" ? 0 : 1); *tag = TY_Date; return w.end_table(start); }
  if (f.size() == 3 && f[0] == 't' && f[1] == 't') {
    w.start_table(); w.scalar<int16_t>(0, (int16_t)time_unit(f[2])); w.scalar<int32_t>(1, (f[2] == 's' || f[2] == 'm') ? 32 : 64); *tag = TY_Time; return w.end_table(start);
  }
  if (f.size() >= 4 && f[0] == 't' && f[1] == 's' && f[3] == ':') {
    const std::string tz = f.substr(4);
    const uint32_t tzs = tz.empty() ? 0 : w.string(tz);
    const uint32_t st = w.mark();
    w.start_table(); w.scalar<int16_t>(0, (int16_t)time_unit(f[2])); w.ref(1, tzs); *tag = TY_Timestamp; return w.end_table(st);
  }
  if (f.size() == 3 && f[0] == 't' && f[1] == 'D') { w.start_table(); w.scalar<int16_t>(0, (int16_t)time_unit(f[2])); *tag = TY_Duration; return w.end_table(start); }
  if (f.rfind("d:", 0) == 0) {
    int p = 0, s = 0, bw = 128;
    if (sscanf(f.c_str(), "d:%d,%d,%d", &p, &s, &bw) < 2) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "decimal format '" + f + "'"};
    w.start_table(); w.scalar<int32_t>(0, p); w.scalar<int32_t>(1, s); w.scalar<int32_t>(2, bw); *tag = TY_Decimal; return w.end_table(start);
  }
  if (f.rfind("w:", 0) == 0) { w.start_table(); w.scalar<int32_t>(0, atoi(f.c_str() + 2)); *tag = TY_FixedSizeBinary; return w.end_table(start); }
  throw ChqError{CHQ_ERR_NOT_SUPPORTED, "Arrow type with format '" + f + "' cannot be written as Arrow IPC by this build"};
}

std::string read_type(const FlatReader& r, uint8_t tag, const uint8_t* t) {
  switch (tag) {
    case TY_Bool: return "b";
    case TY_Utf8: return "u";
    case TY_Int: {
      const int bits = r.scalar<int32_t>(t, 0, 0); const bool sign = r.scalar<uint8_t>(t, 1, 0) != 0;
      switch (bits) { case 8: return sign ? "c" : "C"; case 16: return sign ? "s" : "S"; case 32: return sign ? "i" : "I"; case 64: return sign ? "l" : "L"; default: break; }
      break;
    }
    case TY_FloatingPoint: { const int p = r.scalar<int16_t>(t, 0, 0); return p == 0 ? "e" : p == 1 ? "f" : "g"; }
    case TY_Date: return r.scalar<int16_t>(t, 0, 1) == 0 ? "tdD" : "tdm";
    case TY_Time: return std::string("tt") + unit_char(r.scalar<int16_t>(t, 0, 1));
    case TY_Timestamp: return std::string("ts") + unit_char(r.scalar<int16_t>(t, 0, 0)) + ":" + r.str(t, 1);
    case TY_Duration: return std::string("tD") + unit_char(r.scalar<int16_t>(t, 0, 1));
    case TY_Decimal: {
      const int p = r.scalar<int32_t>(t, 0, 0), s = r.scalar<int32_t>(t, 1, 0), bw = r.scalar<int32_t>(t, 2, 128);
      return "d:" + std::to_string(p) + "," + std::to_string(s) + (bw == 128 ? "" : "," + std::to_string(bw));
    }
    case TY_FixedSizeBinary: return "w:" + std::to_string(r.scalar<int32_t>(t, 0, 0));
    default: break;
  }
  throw ChqError{CHQ_ERR_NOT_SUPPORTED, "Arrow IPC type id " + std::to_string((int)tag) + " is outside this build's scope"};
}

void append_message(std::vector<uint8_t>& out, const std::vector<uint8_t>& fb) {   // continuation, size (padded to 8), flatbuffer
  const uint32_t cont = 0xFFFFFFFFu;
  const int32_t padded = (int32_t)((fb.size() + 7) / 8 * 8);
  const size_t at = out.size();
  out.resize(at + 8 + (size_t)padded, 0);
  memcpy(&out[at], &cont, 4); memcpy(&out[at + 4], &padded, 4);
  memcpy(&out[at + 8], fb.data(), fb.size());
}

constexpr int64_t kBodyAlign = 64;   // what arrow-rs and Arrow C++ writers use; the format asks for 8
int64_t align_up(int64_t v) { return (v + kBodyAlign - 1) / kBodyAlign * kBodyAlign; }

}  // namespace

// =====================================================================================================================
// encode
// =====================================================================================================================
IpcMessage record_to_ipc(Context& ctx, const Batch& dev, bool body_on_device) {
  const int64_t n = dev.nrows;
  if (!dev.on_device) throw ChqError{CHQ_ERR_INVALID_HANDLE, "record_to_ipc expects a device-resident batch"};
  // ---- layout: FieldNodes and Buffers in schema order, every buffer 64-byte aligned -------------------------------------
  struct Piece { int col; int kind; int64_t offset, length; };   // kind: 0 validity, 1 values / bitmap / offsets, 2 utf8 data
  std::vector<Piece> pieces;
  std::vector<std::pair<int64_t, int64_t>> nodes, buffers;
  std::vector<int64_t> null_counts(dev.cols.size(), 0);
  std::vector<std::pair<int32_t, int32_t>> utf8_ends(dev.cols.size(), {0, 0});
  // null counts the producer left unknown (-1) and the Utf8 byte ranges: small read-backs, one synchronisation
  {
    bool any = false;
    std::vector<BufferPtr> counters(dev.cols.size());
    std::vector<unsigned long long> host_counts(dev.cols.size(), 0);
    for (size_t c = 0; c < dev.cols.size(); ++c) {
      const Column& col = dev.cols[c];
      if (col.validity && col.null_count < 0 && n > 0) {
        counters[c] = make_device_buffer(16, ctx.device);
        check_hip(hipMemsetAsync(counters[c]->ptr, 0, 8, ctx.stream), "memset");
        check_hip(launch_count_bits(col.validity, col.offset, n, (unsigned long long*)counters[c]->ptr, ctx.stream), "launch count_bits_kernel");
        check_hip(hipMemcpyAsync(&host_counts[c], counters[c]->ptr, 8, hipMemcpyDeviceToHost, ctx.stream), "read back");
        any = true;
      }
      if (col.type == T_UTF8 && n > 0 && col.values) {
        const int32_t* offs = (const int32_t*)col.values0();
        check_hip(hipMemcpyAsync(&utf8_ends[c].first, offs, 4, hipMemcpyDeviceToHost, ctx.stream), "read offsets");
        check_hip(hipMemcpyAsync(&utf8_ends[c].second, offs + n, 4, hipMemcpyDeviceToHost, ctx.stream), "read offsets");
        any = true;
      }
    }
    if (any) check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
    for (size_t c = 0; c < dev.cols.size(); ++c) {
      const Column& col = dev.cols[c];
      if (!col.validity) null_counts[c] = 0;
      else if (col.null_count >= 0) null_counts[c] = col.null_count;
      else null_counts[c] = n - (int64_t)host_counts[c];
    }
  }
  int64_t at = 0;
  for (size_t c = 0; c < dev.cols.size(); ++c) {
    const Column& col = dev.cols[c];
    nodes.push_back({n, null_counts[c]});
    auto add = [&](int kind, int64_t len) {
      buffers.push_back({at, len});
      if (len > 0) pieces.push_back({(int)c, kind, at, len});
      at = align_up(at + len);
    };
    add(0, null_counts[c] > 0 ? (n + 7) / 8 : 0);
    if (col.type == T_BOOL) add(1, (n + 7) / 8);
    else if (col.type == T_UTF8) { add(1, (n + 1) * 4); add(2, (int64_t)utf8_ends[c].second - utf8_ends[c].first); }
    else add(1, n * col.width);
  }
  const int64_t body_len = at;

  // ---- metadata --------------------------------------------------------------------------------------------------------
  std::vector<uint8_t> header;
  {
    FlatWriter w;
    std::vector<uint32_t> fields;
    for (const Column& col : dev.cols) {
      uint8_t tag = 0;
      const uint32_t type = write_type(w, col.format, &tag);
      const uint32_t name = w.string(col.name);
      const uint32_t children = w.offset_vector({});
      const uint32_t start = w.mark();
      w.start_table();
      w.ref(0, name); w.scalar<uint8_t>(1, col.nullable ? 1 : 0); w.scalar<uint8_t>(2, tag); w.ref(3, type); w.ref(5, children);
      fields.push_back(w.end_table(start));
    }
    const uint32_t fvec = w.offset_vector(fields);
    uint32_t start = w.mark();
    w.start_table(); w.scalar<int16_t>(0, 0); w.ref(1, fvec);
    const uint32_t schema = w.end_table(start);
    start = w.mark();
    w.start_table(); w.scalar<int16_t>(0, 4 /* V5 */); w.scalar<uint8_t>(1, 1 /* Schema */); w.ref(2, schema); w.scalar<int64_t>(3, 0);
    append_message(header, w.finish(w.end_table(start)));
  }
  {
    FlatWriter w;
    const uint32_t bvec = w.pair_vector(buffers);
    const uint32_t nvec = w.pair_vector(nodes);
    uint32_t start = w.mark();
    w.start_table(); w.scalar<int64_t>(0, n); w.ref(1, nvec); w.ref(2, bvec);
    const uint32_t rb = w.end_table(start);
    start = w.mark();
    w.start_table(); w.scalar<int16_t>(0, 4); w.scalar<uint8_t>(1, 3 /* RecordBatch */); w.ref(2, rb); w.scalar<int64_t>(3, body_len);
    append_message(header, w.finish(w.end_table(start)));
  }

  // ---- body: one HBM allocation, every piece placed by a device copy or a small kernel ------------------------------------
  BufferPtr body = make_device_buffer((size_t)body_len + 64, ctx.device);
  uint8_t* bp = (uint8_t*)body->ptr;
  if (body_len > 0) check_hip(hipMemsetAsync(bp, 0, (size_t)body_len, ctx.stream), "memset body");   // padding bytes are zero
  for (const Piece& pc : pieces) {
    const Column& col = dev.cols[(size_t)pc.col];
    uint8_t* dst = bp + pc.offset;
    const bool bitmap = pc.kind == 0 || (pc.kind == 1 && col.type == T_BOOL);
    if (bitmap) {
      const uint8_t* src = pc.kind == 0 ? col.validity : col.values;
      if ((col.offset & 7) == 0) check_hip(hipMemcpyAsync(dst, src + (col.offset >> 3), (size_t)pc.length, hipMemcpyDeviceToDevice, ctx.stream), "copy bitmap");
      else check_hip(launch_bit_shift_copy(src, col.offset, n, (uint32_t*)dst, ctx.stream), "launch bit_shift_copy_kernel");
    } else if (pc.kind == 1 && col.type == T_UTF8) {
      const int32_t* offs = (const int32_t*)col.values0();
      if (utf8_ends[(size_t)pc.col].first == 0) check_hip(hipMemcpyAsync(dst, offs, (size_t)pc.length, hipMemcpyDeviceToDevice, ctx.stream), "copy offsets");
      else check_hip(launch_rebase_offsets(offs, (int32_t*)dst, n + 1, ctx.stream), "launch rebase_offsets_kernel");
    } else if (pc.kind == 2) {
      check_hip(hipMemcpyAsync(dst, col.data + utf8_ends[(size_t)pc.col].first, (size_t)pc.length, hipMemcpyDeviceToDevice, ctx.stream), "copy string bytes");
    } else {
      check_hip(hipMemcpyAsync(dst, col.values0(), (size_t)pc.length, hipMemcpyDeviceToDevice, ctx.stream), "copy values");
    }
  }
  IpcMessage msg;
  msg.header = std::move(header);
  msg.body_len = body_len;
  if (body_on_device) {
    msg.body = body; msg.body_on_device = true;
  } else {
    BufferPtr host = make_host_buffer((size_t)body_len + 64);
    if (body_len > 0) check_hip(hipMemcpyAsync(host->ptr, bp, (size_t)body_len, hipMemcpyDeviceToHost, ctx.stream), "download body");
    msg.body = host; msg.body_on_device = false;
    msg.keep = body;   // until the copy below has completed
  }
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  msg.keep.reset();
  return msg;
}

// =====================================================================================================================
// decode
// =====================================================================================================================
namespace {
struct FieldInfo { std::string name, format; bool nullable = false; };
struct ParsedStream {
  std::vector<FieldInfo> fields;
  int64_t n = 0, body_len = 0;
  std::vector<std::pair<int64_t, int64_t>> nodes, buffers;
  int64_t body_at = -1;   // position of the body inside the stream (right behind the batch message's metadata)
};

ParsedStream parse_stream(const uint8_t* stream, int64_t stream_len) {
  if (!stream || stream_len < 8) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "empty Arrow IPC stream"};
  ParsedStream ps;
  bool have_schema = false, have_batch = false;
  int64_t at = 0;
  while (at + 8 <= stream_len && !have_batch) {
    uint32_t first; memcpy(&first, stream + at, 4);
    int32_t msize;
    if (first == 0xFFFFFFFFu) { memcpy(&msize, stream + at + 4, 4); at += 8; }
    else { msize = (int32_t)first; at += 4; }   // pre-0.15 framing without the continuation marker
    if (msize == 0) break;   // end of stream
    if (msize < 0 || at + msize > stream_len) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "truncated Arrow IPC message"};
    FlatReader r{stream + at, (size_t)msize};
    const uint8_t* m = r.root();
    const uint8_t htype = r.scalar<uint8_t>(m, 1, 0);
    const uint8_t* h = r.indirect(m, 2);
    const int64_t blen = r.scalar<int64_t>(m, 3, 0);
    at += msize;
    if (htype == 1 && h) {   // Schema
      if (r.scalar<int16_t>(h, 0, 0) != 0) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "big-endian Arrow IPC streams are not supported"};
      const uint8_t* fv = r.indirect(h, 1);
      const uint32_t nf = fv ? r.vec_len(fv) : 0;
      for (uint32_t i = 0; i < nf; ++i) {
        const uint8_t* slot = fv + 4 + 4 * i;
        const uint8_t* f = slot + r.rd<uint32_t>(slot);
        if (r.field(f, 4)) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "dictionary-encoded fields are outside this build's scope"};
        const uint8_t* ch = r.indirect(f, 5);
        if (ch && r.vec_len(ch) != 0) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "nested Arrow types are outside this build's scope"};
        FieldInfo fi;
        fi.name = r.str(f, 0); fi.nullable = r.scalar<uint8_t>(f, 1, 0) != 0;
        const uint8_t* ty = r.indirect(f, 3);
        if (!ty) r.bad();
        fi.format = read_type(r, r.scalar<uint8_t>(f, 2, 0), ty);
        ps.fields.push_back(std::move(fi));
      }
      have_schema = true;
    } else if (htype == 3 && h) {   // RecordBatch
      if (!have_schema) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Arrow IPC record batch before its schema"};
      if (r.field(h, 3)) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "compressed Arrow IPC bodies are outside this build's scope"};
      ps.n = r.scalar<int64_t>(h, 0, 0);
      auto pairs = [&](int id, std::vector<std::pair<int64_t, int64_t>>& out) {
        const uint8_t* v = r.indirect(h, id);
        const uint32_t k = v ? r.vec_len(v) : 0;
        for (uint32_t i = 0; i < k; ++i) out.push_back({r.rd<int64_t>(v + 4 + 16 * i), r.rd<int64_t>(v + 4 + 16 * i + 8)});
      };
      pairs(1, ps.nodes); pairs(2, ps.buffers);
      ps.body_len = blen;
      ps.body_at = at;
      have_batch = true;
    } else if (htype == 2) {
      throw ChqError{CHQ_ERR_NOT_SUPPORTED, "dictionary batches are outside this build's scope"};
    } else {
      at += blen;   // a message kind we do not need
    }
  }
  if (!have_schema || !have_batch) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Arrow IPC stream without a schema and a record batch"};
  if (ps.n < 0 || ps.nodes.size() != ps.fields.size()) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Arrow IPC record batch does not match its schema"};
  return ps;
}
}  // namespace

// host only: what a stream's metadata says (the CPU test tier checks the flatbuffer reader against pyarrow's writer)
std::string describe_ipc(const uint8_t* stream, int64_t stream_len) {
  const ParsedStream ps = parse_stream(stream, stream_len);
  std::string out = "rows " + std::to_string(ps.n) + " body " + std::to_string(ps.body_len) + " body_at " + std::to_string(ps.body_at) + "\n";
  for (size_t c = 0; c < ps.fields.size(); ++c)
    out += "field " + ps.fields[c].name + " " + ps.fields[c].format + " nullable=" + (ps.fields[c].nullable ? "1" : "0") +
           " nulls=" + std::to_string(ps.nodes[c].second) + "\n";
  for (auto& b : ps.buffers) out += "buffer " + std::to_string(b.first) + " " + std::to_string(b.second) + "\n";
  return out;
}

Batch record_from_ipc(Context& ctx, const uint8_t* stream, int64_t stream_len, const void* body, int64_t body_len,
                      bool body_on_device, bool out_on_device) {
  const ParsedStream ps = parse_stream(stream, stream_len);
  const std::vector<FieldInfo>& fields = ps.fields;
  const int64_t n = ps.n, meta_body_len = ps.body_len;
  const std::vector<std::pair<int64_t, int64_t>>& nodes = ps.nodes;
  const std::vector<std::pair<int64_t, int64_t>>& buffers = ps.buffers;
  const uint8_t* inline_body = nullptr;
  if (!body) {
    if (ps.body_at + meta_body_len > stream_len) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Arrow IPC body is shorter than its metadata says"};
    inline_body = stream + ps.body_at;
  }
  const void* src_body = body ? body : (const void*)inline_body;
  const bool src_on_device = body ? body_on_device : false;
  if (body && body_len < meta_body_len) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Arrow IPC body is shorter than its metadata says"};

  // ---- the body goes to its destination with ONE copy; the columns are views into it -------------------------------------
  BufferPtr owned = out_on_device ? make_device_buffer((size_t)meta_body_len + 64, ctx.device) : make_host_buffer((size_t)meta_body_len + 64);
  if (meta_body_len > 0) {
    if (!out_on_device && !src_on_device) memcpy(owned->ptr, src_body, (size_t)meta_body_len);
    else {
      const hipMemcpyKind k = out_on_device ? (src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) : hipMemcpyDeviceToHost;
      check_hip(hipMemcpyAsync(owned->ptr, src_body, (size_t)meta_body_len, k, ctx.stream), "move Arrow IPC body");
    }
  }
  Batch out;
  out.nrows = n; out.on_device = out_on_device; out.device_id = out_on_device ? ctx.device : -1;
  size_t bi = 0;
  std::vector<std::pair<const int32_t*, int64_t>> utf8_checks;   // (offsets, data length) to validate after the copy landed
  auto take = [&](int64_t need, const char* what) -> const uint8_t* {
    if (bi >= buffers.size()) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Arrow IPC record batch has too few buffers"};
    const auto [off, len] = buffers[bi++];
    if (off < 0 || len < 0 || off + len > meta_body_len || len < need)
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, std::string("Arrow IPC buffer (") + what + ") outside the body or too short"};
    return (const uint8_t*)owned->ptr + off;
  };
  for (size_t c = 0; c < fields.size(); ++c) {
    Column col;
    col.name = fields[c].name; col.format = fields[c].format; col.nullable = fields[c].nullable;
    parse_arrow_format(col.format.c_str(), &col.type, &col.width);   // the C-data format parser knows type and width
    if (nodes[c].first != n) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Arrow IPC field node length differs from the batch length"};
    col.length = n; col.offset = 0; col.null_count = nodes[c].second;
    const uint8_t* validity = take(col.null_count > 0 ? (n + 7) / 8 : 0, "validity");
    col.validity = col.null_count > 0 ? validity : nullptr;
    if (col.type == T_BOOL) col.values = take((n + 7) / 8, "bitmap");
    else if (col.type == T_UTF8) {
      col.values = take((n + 1) * 4, "offsets");
      const size_t data_idx = bi;
      col.data = take(0, "string bytes");
      utf8_checks.push_back({(const int32_t*)col.values, buffers[data_idx].second});
    } else col.values = take(n * col.width, "values");
    col.owned.push_back(owned);
    out.cols.push_back(std::move(col));
  }
  // a kernel must never follow offsets out of the data buffer: EVERY offset is checked (0 <= off[i] <= off[i+1] <= data
  // length), on the device for a device result -- one small kernel per Utf8 column, one flag word read back with the
  // synchronisation that was needed anyway -- and by a plain loop for a host result
  BufferPtr flags;
  if (out_on_device && !utf8_checks.empty()) {
    flags = make_device_buffer(16, ctx.device);
    check_hip(hipMemsetAsync(flags->ptr, 0, 16, ctx.stream), "memset");
    for (auto& chk : utf8_checks)
      check_hip(launch_validate_offsets(chk.first, n, chk.second, (uint32_t*)flags->ptr, ctx.stream), "launch validate_offsets_kernel");
  }
  uint32_t bad = 0;
  if (flags) check_hip(hipMemcpyAsync(&bad, flags->ptr, 4, hipMemcpyDeviceToHost, ctx.stream), "read back");
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  if (!out_on_device) {
    for (auto& chk : utf8_checks)
      for (int64_t i = 0; i < n && !bad; ++i) bad |= chk.first[i] < 0 || chk.first[i + 1] < chk.first[i] || (int64_t)chk.first[i + 1] > chk.second;
  }
  if (bad) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Arrow IPC Utf8 offsets are not monotonic or point outside the data buffer"};
  return out;
}

}  // namespace chq
