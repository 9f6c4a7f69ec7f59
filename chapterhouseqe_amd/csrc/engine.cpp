// engine.cpp -- context, HBM pool, Arrow C Data import/export, and the record-level operations that
// drive the kernels.  Reference map: filter_record = RU/filter_record.rs:21-39, project_record =
// RU/record_projection.rs:16-76, compute_value = RU/compute_value.rs:57-344 (RU = src/handlers/
// operator_handler/operators/record_utils of the reference).
#include "engine.hpp"
#include <deque>
#include <exception>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>

namespace chq {

void check_hip(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    int code = (e == hipErrorOutOfMemory) ? CHQ_ERR_OUT_OF_MEMORY : CHQ_ERR_DEVICE;
    throw ChqError{code, std::string(what) + ": " + hipGetErrorString(e)};
  }
}

// =================================================================================================
// memory
// =================================================================================================
DevicePool& DevicePool::instance() {
  static DevicePool* pool = new DevicePool();   // intentionally leaked: must outlive late Arrow releases
  return *pool;
}
static size_t size_class(size_t bytes) {
  if (bytes < 256) return 256;
  if (bytes <= (1u << 20)) { size_t c = 256; while (c < bytes) c <<= 1; return c; }
  const size_t g = 2u << 20;
  return (bytes + g - 1) / g * g;
}
static uint64_t pool_key(size_t cap, int device) { return ((uint64_t)cap << 8) | (uint64_t)(device & 0xff); }
void* DevicePool::alloc(size_t bytes, int device, size_t* cap_out) {
  const size_t cap = size_class(bytes);
  *cap_out = cap;
  {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = free_.find(pool_key(cap, device));
    if (it != free_.end() && !it->second.empty()) { void* p = it->second.back(); it->second.pop_back(); return p; }
  }
  // allocate on the device the block is keyed by, whatever the calling thread's current device is (peer copies
  // allocate on the destination GPU from a call that started on the source context)
  int current = device;
  (void)hipGetDevice(&current);
  if (current != device) check_hip(hipSetDevice(device), "hipSetDevice");
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, cap);
  if (e == hipErrorOutOfMemory) { trim(); e = hipMalloc(&p, cap); }
  if (current != device) (void)hipSetDevice(current);
  check_hip(e, "hipMalloc");
  return p;
}
void DevicePool::free(void* p, size_t cap, int device) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(mu_);
  free_[pool_key(cap, device)].push_back(p);
}
void DevicePool::trim() {
  std::unordered_map<uint64_t, std::vector<void*>> f;
  { std::lock_guard<std::mutex> lk(mu_); f.swap(free_); }
  for (auto& kv : f) for (void* p : kv.second) (void)hipFree(p);
}

HostPool& HostPool::instance() {
  static HostPool* pool = new HostPool();   // leaked on purpose, like the device pool
  return *pool;
}
void* HostPool::alloc(size_t bytes, size_t* cap_out) {
  const size_t cap = size_class(bytes < 64 ? 64 : bytes);
  *cap_out = cap;
  if (cap >= ((size_t)1 << 20)) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = free_.find(cap);
    if (it != free_.end() && !it->second.empty()) { void* p = it->second.back(); it->second.pop_back(); cached_ -= cap; return p; }
  }
  void* p = nullptr;
  if (posix_memalign(&p, 64, cap) != 0) throw ChqError{CHQ_ERR_OUT_OF_MEMORY, "host allocation failed"};
  return p;
}
void HostPool::free(void* p, size_t cap) {
  if (!p) return;
  if (cap >= ((size_t)1 << 20)) {
    std::lock_guard<std::mutex> lk(mu_);
    if (cached_ + cap <= limit_) { free_[cap].push_back(p); cached_ += cap; return; }
  }
  ::free(p);
}
void HostPool::trim() {
  std::unordered_map<size_t, std::vector<void*>> f;
  { std::lock_guard<std::mutex> lk(mu_); f.swap(free_); cached_ = 0; }
  for (auto& kv : f) for (void* p : kv.second) ::free(p);
}
void HostPool::set_limit(size_t bytes) {
  { std::lock_guard<std::mutex> lk(mu_); limit_ = bytes; }
  if (bytes == 0) trim();
}

Buffer::~Buffer() {
  if (!ptr) return;
  // A block released while an exception unwinds the call may still be the source or target of work queued on the context's
  // streams (staged uploads ahead of a typing error, copies behind a kernel that reported a data error): the pools are
  // process-wide, and another context -- another stream -- would get it next.  The fuzz met exactly that: a call's first
  // output columns overwritten by the late upload of an earlier, failed call.  Errors are the slow path: wait for the device.
  if (std::uncaught_exceptions() > 0) (void)hipDeviceSynchronize();
  if (device) DevicePool::instance().free(ptr, cap, device_id); else HostPool::instance().free(ptr, cap);
}
BufferPtr make_device_buffer(size_t bytes, int device) {
  auto b = std::make_shared<Buffer>();
  b->ptr = DevicePool::instance().alloc(bytes ? bytes : 1, device, &b->cap);
  b->bytes = bytes; b->device = true; b->device_id = device;
  return b;
}
BufferPtr make_host_buffer(size_t bytes) {
  auto b = std::make_shared<Buffer>();
  b->ptr = HostPool::instance().alloc(bytes ? bytes : 1, &b->cap);
  b->bytes = bytes; b->device = false;
  return b;
}

// The auxiliary streams and their fork / join events: created together, on the context's device, the first time a call
// needs them (the ONE creation site; `aux_ready` is set only when every object exists, so a failure half way is retried
// from the first missing object instead of leaving null streams behind a non-null event).
void ensure_aux_streams(Context& ctx) {
  if (ctx.aux_ready) return;
  int current = ctx.device;
  (void)hipGetDevice(&current);
  if (current != ctx.device) check_hip(hipSetDevice(ctx.device), "hipSetDevice");
  if (!ctx.aux_fork) check_hip(hipEventCreateWithFlags(&ctx.aux_fork, hipEventDisableTiming), "hipEventCreate");
  for (int i = 0; i < Context::kAuxStreams; ++i) {
    if (!ctx.aux[i]) check_hip(hipStreamCreateWithFlags(&ctx.aux[i], hipStreamNonBlocking), "hipStreamCreate");
    if (!ctx.aux_join[i]) check_hip(hipEventCreateWithFlags(&ctx.aux_join[i], hipEventDisableTiming), "hipEventCreate");
  }
  ctx.aux_ready = true;
}
// Work on the auxiliary streams starts behind everything queued on ctx.stream ...
void fork_aux_streams(Context& ctx) {
  ensure_aux_streams(ctx);
  check_hip(hipEventRecord(ctx.aux_fork, ctx.stream), "hipEventRecord(fork)");
  for (int i = 0; i < Context::kAuxStreams; ++i) check_hip(hipStreamWaitEvent(ctx.aux[i], ctx.aux_fork, 0), "hipStreamWaitEvent(fork)");
}
// ... and ctx.stream goes on only behind everything queued on them (no host synchronisation on either end)
void join_aux_streams(Context& ctx) {
  for (int i = 0; i < Context::kAuxStreams; ++i) {
    check_hip(hipEventRecord(ctx.aux_join[i], ctx.aux[i]), "hipEventRecord(join)");
    check_hip(hipStreamWaitEvent(ctx.stream, ctx.aux_join[i], 0), "hipStreamWaitEvent(join)");
  }
}

Context::~Context() {
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
  if (aux_fork) (void)hipEventDestroy(aux_fork);
  for (hipEvent_t e : upload_events) (void)hipEventDestroy(e);
  for (int i = 0; i < kAuxStreams; ++i) { if (aux_join[i]) (void)hipEventDestroy(aux_join[i]); if (aux[i]) (void)hipStreamDestroy(aux[i]); }
  if (pinned) (void)hipHostFree(pinned);
  if (pinned_tbl) (void)hipHostFree(pinned_tbl);
  if (pinned_sizes) (void)hipHostFree(pinned_sizes);
  if (pinned_io) (void)hipHostFree(pinned_io);
  if (own_stream && stream) (void)hipStreamDestroy(stream);
}

// =================================================================================================
// Arrow C Data Interface
// =================================================================================================
const void* Column::values0() const {
  if (type == T_BOOL) return values;
  if (type == T_UTF8) return values ? values + 4 * offset : nullptr;
  return values ? values + (int64_t)width * offset : nullptr;
}

void parse_arrow_format(const char* f, DType* t, int* width);
static void parse_format(const char* f, DType* t, int* width) { parse_arrow_format(f, t, width); }
void parse_arrow_format(const char* f, DType* t, int* width) {
  *width = 0;
  if (f && f[0] && !f[1]) {   // the primitive types are one character: no string object on the per-batch import path
    switch (f[0]) {
      case 'b': *t = T_BOOL; return;
      case 'c': *t = T_I8; *width = 1; return;   case 'C': *t = T_U8; *width = 1; return;
      case 's': *t = T_I16; *width = 2; return;  case 'S': *t = T_U16; *width = 2; return;
      case 'i': *t = T_I32; *width = 4; return;  case 'I': *t = T_U32; *width = 4; return;
      case 'l': *t = T_I64; *width = 8; return;  case 'L': *t = T_U64; *width = 8; return;
      case 'e': *t = T_F16; *width = 2; return;  case 'f': *t = T_F32; *width = 4; return;
      case 'g': *t = T_F64; *width = 8; return;  case 'u': *t = T_UTF8; return;
      default: break;
    }
  }
  const std::string s(f ? f : "");   // multi-character formats: opaque fixed-width types that are only copied
  *t = T_FIXED_OPAQUE;
  if (s == "tdD" || s == "tts" || s == "ttm") { *width = 4; return; }
  if (s == "tdm" || s == "ttu" || s == "ttn" || s.rfind("ts", 0) == 0 || s.rfind("tD", 0) == 0) { *width = 8; return; }
  if (s.rfind("d:", 0) == 0) {   // decimal128 unless a bit width says otherwise
    int commas = (int)std::count(s.begin(), s.end(), ',');
    if (commas == 1) { *width = 16; return; }
    if (commas == 2) { int bw = atoi(s.substr(s.rfind(',') + 1).c_str()); if (bw == 32) { *width = 4; return; } if (bw == 64) { *width = 8; return; } if (bw == 128) { *width = 16; return; } }
  }
  if (s.rfind("w:", 0) == 0) { int w = atoi(s.c_str() + 2); if (w == 1 || w == 2 || w == 4 || w == 8 || w == 16) { *width = w; return; } }
  throw ChqError{CHQ_ERR_NOT_SUPPORTED, "Arrow type with format '" + s + "' is outside this build's scope"};
}

Batch import_batch(const ArrowDeviceArray* rec, const ArrowSchema* schema) {
  if (!rec || !schema || !schema->format) throw ChqError{CHQ_ERR_INVALID_HANDLE, "null record batch"};
  if (strcmp(schema->format, "+s") != 0) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "record batch must be a struct array"};
  const ArrowArray& a = rec->array;
  if (a.offset != 0) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "sliced struct arrays are not supported; slice the children"};
  if (a.n_children != schema->n_children) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "schema / array children mismatch"};
  Batch b;
  b.nrows = a.length;
  b.device_id = (int)rec->device_id;
  switch (rec->device_type) {
    case ARROW_DEVICE_CPU: case ARROW_DEVICE_ROCM_HOST: case ARROW_DEVICE_CUDA_HOST: b.on_device = false; break;
    case ARROW_DEVICE_ROCM: b.on_device = true; break;
    default: throw ChqError{CHQ_ERR_NOT_SUPPORTED, "unsupported Arrow device type"};
  }
  if (b.on_device && rec->sync_event) check_hip(hipEventSynchronize(*(hipEvent_t*)rec->sync_event), "hipEventSynchronize(sync_event)");
  b.cols.reserve((size_t)a.n_children);
  for (int64_t i = 0; i < a.n_children; ++i) {
    const ArrowArray* ca = a.children[i];
    const ArrowSchema* cs = schema->children[i];
    if (!ca || !cs) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "missing child array or schema"};
    Column c;
    c.name = cs->name ? cs->name : "";
    c.format = cs->format ? cs->format : "";
    parse_format(cs->format, &c.type, &c.width);
    c.nullable = (cs->flags & ARROW_FLAG_NULLABLE) != 0;
    if (ca->length < b.nrows) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "column shorter than the record batch"};
    if (ca->offset < 0 || b.nrows < 0) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "negative length or offset"};
    // a kernel must never be handed a null data pointer: refuse malformed arrays here, on the host
    if (b.nrows > 0 && (ca->n_buffers < 2 || !ca->buffers || !ca->buffers[1]))
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, std::string("column '") + (cs->name ? cs->name : "") + "' has no values buffer"};
    if (ca->null_count > 0 && (ca->n_buffers < 1 || !ca->buffers[0]))
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, std::string("column '") + (cs->name ? cs->name : "") + "' reports nulls but has no validity bitmap"};
    c.length = b.nrows;
    c.offset = ca->offset;
    c.validity = ca->n_buffers > 0 ? (const uint8_t*)ca->buffers[0] : nullptr;
    c.values = ca->n_buffers > 1 ? (const uint8_t*)ca->buffers[1] : nullptr;
    c.data = ca->n_buffers > 2 ? (const uint8_t*)ca->buffers[2] : nullptr;
    c.null_count = ca->null_count;   // -1 = unknown, resolved when staged
    if (!c.validity) c.null_count = 0;
    if (c.type == T_UTF8 && b.nrows > 0 && !c.data) {
      // legal only when every string is empty; host batches can be checked, device batches are taken at their word
      if (!b.on_device) {
        const int32_t* offs = (const int32_t*)c.values + c.offset;
        if (offs[b.nrows] != offs[0]) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Utf8 column '" + c.name + "' has offsets but no data buffer"};
      }
    }
    b.cols.push_back(std::move(c));
  }
  return b;
}

namespace {
struct ArrayHolder {
  std::vector<const void*> buffers;
  std::vector<ArrowArray*> child_ptrs;
  std::vector<ArrowArray> children;
  std::vector<BufferPtr> owned;
  hipEvent_t event = nullptr;   // ArrowDeviceArray::sync_event points here when set; destroyed with the array
};
struct SchemaHolder {
  std::string format, name;
  std::vector<ArrowSchema*> child_ptrs;
  std::vector<ArrowSchema> children;
};
void release_array(ArrowArray* a) {
  if (!a || !a->release) return;
  auto* h = (ArrayHolder*)a->private_data;
  if (h) {
    for (auto& c : h->children) if (c.release) c.release(&c);
    if (h->event) { (void)hipEventSynchronize(h->event); (void)hipEventDestroy(h->event); }   // buffers go back to the pool only once the copies behind the event are done
    delete h;
  }
  a->release = nullptr; a->private_data = nullptr;
}
void release_schema(ArrowSchema* s) {
  if (!s || !s->release) return;
  auto* h = (SchemaHolder*)s->private_data;
  if (h) { for (auto& c : h->children) if (c.release) c.release(&c); delete h; }
  s->release = nullptr; s->private_data = nullptr;
}
void fill_column_array(Column&& c, ArrowArray* out) {
  auto* h = new ArrayHolder();
  h->owned = std::move(c.owned);
  const bool keep_validity = c.validity && c.null_count != 0;
  h->buffers.push_back(keep_validity ? c.validity : nullptr);
  h->buffers.push_back(c.values);
  if (c.type == T_UTF8) h->buffers.push_back(c.data);
  memset(out, 0, sizeof(*out));
  out->length = c.length; out->null_count = keep_validity ? c.null_count : 0; out->offset = c.offset;
  out->n_buffers = (int64_t)h->buffers.size(); out->buffers = h->buffers.data();
  out->release = release_array; out->private_data = h;
}
void fill_column_schema(const Column& c, ArrowSchema* out) {
  auto* h = new SchemaHolder();
  h->format = c.format; h->name = c.name;
  memset(out, 0, sizeof(*out));
  out->format = h->format.c_str(); out->name = h->name.c_str();
  out->flags = c.nullable ? ARROW_FLAG_NULLABLE : 0;
  out->release = release_schema; out->private_data = h;
}
}  // namespace

void export_batch(Batch&& b, int device_type, ArrowDeviceArray* out, ArrowSchema* out_schema, hipEvent_t sync_event) {
  auto* ah = new ArrayHolder();
  ah->event = sync_event;
  auto* sh = new SchemaHolder();
  const size_t n = b.cols.size();
  ah->children.resize(n); sh->children.resize(n);
  for (size_t i = 0; i < n; ++i) {
    fill_column_schema(b.cols[i], &sh->children[i]);
    fill_column_array(std::move(b.cols[i]), &ah->children[i]);
  }
  for (size_t i = 0; i < n; ++i) { ah->child_ptrs.push_back(&ah->children[i]); sh->child_ptrs.push_back(&sh->children[i]); }
  ah->buffers.push_back(nullptr);
  memset(out, 0, sizeof(*out));
  out->array.length = b.nrows; out->array.null_count = 0; out->array.offset = 0;
  out->array.n_buffers = 1; out->array.buffers = ah->buffers.data();
  out->array.n_children = (int64_t)n; out->array.children = ah->child_ptrs.data();
  out->array.release = release_array; out->array.private_data = ah;
  out->device_id = device_type == ARROW_DEVICE_ROCM ? b.device_id : -1;
  out->device_type = device_type; out->sync_event = ah->event ? (void*)&ah->event : nullptr;
  sh->format = "+s"; sh->name = "";
  memset(out_schema, 0, sizeof(*out_schema));
  out_schema->format = sh->format.c_str(); out_schema->name = sh->name.c_str();
  out_schema->n_children = (int64_t)n; out_schema->children = sh->child_ptrs.data();
  out_schema->release = release_schema; out_schema->private_data = sh;
}

void export_single_column(Column&& c, bool on_device, int device_id, ArrowDeviceArray* out, ArrowSchema* out_schema) {
  fill_column_schema(c, out_schema);
  memset(out, 0, sizeof(*out));
  fill_column_array(std::move(c), &out->array);
  out->device_id = on_device ? device_id : -1;
  out->device_type = on_device ? ARROW_DEVICE_ROCM : ARROW_DEVICE_CPU;
}

// =================================================================================================
// host <-> device staging
// =================================================================================================
namespace {
// byte range of a bitmap covering bits [offset, offset+len), keeping the sub-byte phase
struct BitRange { int64_t first_byte, nbytes; };
BitRange bit_range(int64_t offset, int64_t len) {
  int64_t fb = offset >> 3, lb = (offset + len + 7) >> 3;
  return {fb, lb - fb};
}
enum class Dir { H2D, D2H, D2D, H2H, P2P };
// P2P: `ctx` is the DESTINATION context (buffers on its device, copies on its stream), `peer_device` the GPU `src` lives on
BufferPtr copy_bytes(Context& ctx, const uint8_t* src, int64_t nbytes, Dir dir, size_t pad = 16, int peer_device = -1) {
  BufferPtr out = (dir == Dir::D2H || dir == Dir::H2H) ? make_host_buffer((size_t)nbytes + pad) : make_device_buffer((size_t)nbytes + pad, ctx.device);
  if (dir == Dir::H2H) { if (nbytes > 0) memcpy(out->ptr, src, (size_t)nbytes); return out; }
  if (dir == Dir::P2P) {
    if (nbytes > 0) check_hip(hipMemcpyPeerAsync(out->ptr, ctx.device, src, peer_device, (size_t)nbytes, ctx.stream), "hipMemcpyPeerAsync");
    return out;
  }
  if (nbytes > 0) {
    hipMemcpyKind k = dir == Dir::H2D ? hipMemcpyHostToDevice : (dir == Dir::D2H ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    check_hip(hipMemcpyAsync(out->ptr, src, (size_t)nbytes, k, ctx.stream), "hipMemcpyAsync");
  }
  return out;
}

// Copy one column across (or within) memory spaces; the copy keeps `offset & 7` so that validity,
// boolean values and value buffers share one Arrow offset.
Column copy_column(Context& ctx, const Column& c, Dir dir, int peer_device = -1) {
  auto copy_bytes = [peer_device](Context& cx, const uint8_t* src, int64_t nbytes, Dir d, size_t pad = 16) {
    return chq::copy_bytes(cx, src, nbytes, d, pad, peer_device);
  };
  Column o;
  o.name = c.name; o.format = c.format; o.type = c.type; o.width = c.width; o.nullable = c.nullable;
  o.length = c.length; o.null_count = c.null_count;
  const int64_t phase = c.offset & 7, base = c.offset - phase, n = c.length;
  o.offset = phase;
  if (c.validity && c.null_count != 0) {
    BitRange r = bit_range(c.offset, n);
    auto vb = copy_bytes(ctx, c.validity + r.first_byte, r.nbytes, dir);
    o.validity = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
  }
  if (c.type == T_BOOL) {
    BitRange r = bit_range(c.offset, n);
    auto b = copy_bytes(ctx, c.values + r.first_byte, r.nbytes, dir);
    o.values = (const uint8_t*)b->ptr; o.owned.push_back(b);
  } else if (c.type == T_UTF8) {
    // offsets [base, offset+n]; data bytes [off[offset], off[offset+n]) -- need the two end offsets on the host
    int32_t ends[2] = {0, 0};
    if (c.values) {
      if (dir == Dir::H2D || dir == Dir::H2H) { const int32_t* offs = (const int32_t*)c.values; ends[0] = offs[c.offset]; ends[1] = offs[c.offset + n]; }
      else {   // (unified addressing: the copy finds the source GPU from the pointer, also for a peer's memory)
        check_hip(hipMemcpyAsync(&ends[0], c.values + 4 * c.offset, 4, hipMemcpyDeviceToHost, ctx.stream), "read offsets");
        check_hip(hipMemcpyAsync(&ends[1], c.values + 4 * (c.offset + n), 4, hipMemcpyDeviceToHost, ctx.stream), "read offsets");
        check_hip(hipStreamSynchronize(ctx.stream), "sync");
      }
    }
    auto ob = c.values ? copy_bytes(ctx, c.values + 4 * base, 4 * (n + phase + 1), dir) : copy_bytes(ctx, nullptr, 0, dir);
    if (!c.values) {   // empty array without an offsets buffer: synthesise [0]
      int32_t zero[9] = {0};
      if (dir == Dir::D2H || dir == Dir::H2H) memcpy(ob->ptr, zero, sizeof zero); else check_hip(hipMemcpyAsync(ob->ptr, zero, 16, hipMemcpyHostToDevice, ctx.stream), "memcpy");
    }
    o.values = (const uint8_t*)ob->ptr; o.owned.push_back(ob);
    const int64_t nb = (int64_t)ends[1] - ends[0];
    auto db = copy_bytes(ctx, c.data ? c.data + ends[0] : nullptr, c.data ? nb : 0, dir);
    o.data = (const uint8_t*)db->ptr - ends[0];   // offsets stay absolute
    o.data_bytes = nb;
    o.owned.push_back(db);
  } else {
    auto b = copy_bytes(ctx, c.values ? c.values + (int64_t)c.width * base : nullptr, c.values ? (int64_t)c.width * (n + phase) : 0, dir);
    o.values = (const uint8_t*)b->ptr; o.owned.push_back(b);
  }
  return o;
}

int64_t count_nulls_host(const uint8_t* validity, int64_t offset, int64_t n) {
  int64_t nulls = 0;
  for (int64_t i = 0; i < n; ++i) { int64_t b = offset + i; nulls += !((validity[b >> 3] >> (b & 7)) & 1); }
  return nulls;
}
}  // namespace

Batch to_device(Context& ctx, const Batch& b) {
  Batch o;
  o.nrows = b.nrows; o.on_device = true; o.device_id = ctx.device;
  for (const Column& c : b.cols) {
    Column cc = c;
    if (!b.on_device) {
      if (cc.validity && cc.null_count < 0) cc.null_count = count_nulls_host(cc.validity, cc.offset, cc.length);
      o.cols.push_back(copy_column(ctx, cc, Dir::H2D));
    } else {
      if (cc.validity && cc.null_count < 0) cc.null_count = 1;   // unknown: assume nulls may be present
      o.cols.push_back(cc);   // view, nothing owned
    }
  }
  return o;
}

Batch copy_to_peer(Context& src, Context& dst, const Batch& b, hipEvent_t* event_out) {
  if (!b.on_device) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "chq_record_copy_to_peer moves device-resident batches; stage host batches with chq_record_to_device on the destination context"};
  check_hip(hipSetDevice(dst.device), "hipSetDevice");
  if (src.device != dst.device) {   // direct xGMI path between the pair (the copy is staged through the host without it)
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dst.device, src.device) == hipSuccess && can) {
      const hipError_t e = hipDeviceEnablePeerAccess(src.device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) check_hip(e, "hipDeviceEnablePeerAccess");
      (void)hipGetLastError();
    }
  }
  Batch o;
  o.nrows = b.nrows; o.on_device = true; o.device_id = dst.device;
  for (const Column& c : b.cols) {
    Column cc = c;
    if (cc.validity && cc.null_count < 0) cc.null_count = 1;   // unknown: keep the bitmap
    o.cols.push_back(copy_column(dst, cc, Dir::P2P, src.device));
  }
  hipEvent_t ev = nullptr;
  check_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
  const hipError_t e = hipEventRecord(ev, dst.stream);
  if (e != hipSuccess) { (void)hipEventDestroy(ev); check_hip(e, "hipEventRecord"); }
  *event_out = ev;
  return o;
}

Batch to_host(Context& ctx, const Batch& b) {
  Batch o;
  o.nrows = b.nrows; o.on_device = false; o.device_id = -1;
  for (const Column& c : b.cols) o.cols.push_back(b.on_device ? copy_column(ctx, c, Dir::D2H) : c);
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  return o;
}

void GroupLite::set(size_t b, const Batch& r, const Batch& first, int device) {
  rows[b] = r.nrows;
  uint8_t f = 0;
  if (r.on_device && r.device_id == device) f |= GL_ON_DEVICE;
  if (r.nrows < 2) f |= GL_SHORT;
  if (r.cols.size() != ncols) { flags[b] = (uint8_t)(f | GL_SCHEMA_DIFFERS); return; }
  for (size_t i = 0; i < ncols; ++i) {
    const Column& c = r.cols[i];
    const Column& c0 = first.cols[i];
    if (c.type != c0.type || c.width != c0.width || c.format != c0.format) f |= GL_SCHEMA_DIFFERS;
    if (c.validity && c.null_count != 0) f |= GL_NULLS;
    if (c.type == T_UTF8 && c.data == nullptr) f |= GL_NO_UTF8_DATA;
    values0[b * ncols + i] = c.type == T_BOOL ? c.values : (const uint8_t*)c.values0();
    data[b * ncols + i] = c.data;
    validity[b * ncols + i] = (c.validity && c.null_count != 0) ? c.validity : nullptr;
    offset[b * ncols + i] = c.offset;
  }
  flags[b] = f;
}

std::vector<PlanColumn> plan_columns(const Batch& b, const chq_table_aliases* aliases) {
  std::vector<PlanColumn> out;
  for (size_t i = 0; i < b.cols.size(); ++i) {
    PlanColumn p;
    p.name = b.cols[i].name; p.type = b.cols[i].type; p.format = b.cols[i].format; p.width = b.cols[i].width;
    p.has_nulls = b.cols[i].validity && b.cols[i].null_count != 0;
    // without an explicit table_aliases argument every column has an (empty) alias list
    p.alias_entry_present = aliases ? (int)i < aliases->n_columns : true;
    if (aliases && (int)i < aliases->n_columns)
      for (int k = 0; k < aliases->columns[i].n; ++k) p.aliases.push_back(aliases->columns[i].aliases[k]);
    out.push_back(std::move(p));
  }
  return out;
}

namespace {
class WorkPool {
 public:
  WorkPool() {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n = std::min(16u, hw) - 1;   // + the calling thread
    for (unsigned i = 0; i < n; ++i) threads_.emplace_back([this] { worker(); });
  }
  ~WorkPool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  unsigned width() const { return (unsigned)threads_.size() + 1; }
  void run(unsigned tasks, const std::function<void(unsigned)>& f) {
    if (tasks == 0) return;
    std::unique_lock<std::mutex> busy(run_m_, std::try_to_lock);
    if (tasks == 1 || threads_.empty() || !busy.owns_lock()) { for (unsigned t = 0; t < tasks; ++t) f(t); return; }
    std::unique_lock<std::mutex> l(m_);
    job_ = &f; next_ = 0; total_ = tasks; finished_ = 0; error_ = nullptr; ++epoch_;
    cv_.notify_all();
    drain(l);
    done_cv_.wait(l, [this] { return finished_ == total_; });
    job_ = nullptr;
    if (error_) { auto e = error_; error_ = nullptr; l.unlock(); std::rethrow_exception(e); }
  }

 private:
  void drain(std::unique_lock<std::mutex>& l) {   // called with m_ held
    while (job_ && next_ < total_) {
      const unsigned t = next_++;
      const auto* f = job_;
      l.unlock();
      std::exception_ptr err;
      try { (*f)(t); } catch (...) { err = std::current_exception(); }
      l.lock();
      if (err && !error_) error_ = err;
      if (++finished_ == total_) done_cv_.notify_all();
    }
  }
  void worker() {
    std::unique_lock<std::mutex> l(m_);
    unsigned seen = 0;
    while (true) {
      cv_.wait(l, [&] { return stop_ || epoch_ != seen; });
      if (stop_) return;
      seen = epoch_;
      drain(l);
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_, run_m_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(unsigned)>* job_ = nullptr;
  unsigned next_ = 0, total_ = 0, finished_ = 0, epoch_ = 0;
  std::exception_ptr error_;
  bool stop_ = false;
};
WorkPool& work_pool() { static WorkPool p; return p; }
}  // namespace

void pool_run(unsigned tasks, const std::function<void(unsigned)>& f) { work_pool().run(tasks, f); }
unsigned pool_width() { return work_pool().width(); }
void pool_ranges(size_t n, size_t grain, const std::function<void(size_t, size_t)>& f) {
  if (n == 0) return;
  const size_t want = grain ? (n + grain - 1) / grain : 1;
  const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(want, pool_width()));
  const size_t per = (n + T - 1) / T;
  pool_run(T, [&](unsigned t) { const size_t i0 = std::min(n, (size_t)t * per), i1 = std::min(n, (size_t)(t + 1) * per); if (i0 < i1) f(i0, i1); });
}

PhaseTimer::PhaseTimer(const char* w) : what(w) {
  static const bool enabled = [] { const char* e = getenv("CHQ_TIMING"); return e && *e == '1'; }();
  on = enabled;
  if (on) t0 = last = std::chrono::steady_clock::now();
}
void PhaseTimer::mark(const char* phase) {
  if (!on) return;
  const auto now = std::chrono::steady_clock::now();
  line += std::string(" ") + phase + "=" + std::to_string(std::chrono::duration<double, std::micro>(now - last).count()).substr(0, 8) + "us";
  last = now;
}
PhaseTimer::~PhaseTimer() {
  if (!on) return;
  fprintf(stderr, "[chq timing] %s total=%.1fus%s\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), line.c_str());
}

// =================================================================================================
// kernel drivers
// =================================================================================================
namespace {

constexpr int64_t kTileRows[3] = {1024 * 16, 256 * 8, 256 * 8};
constexpr int kGridPerCu[3] = {1, 4, 4};
constexpr int kStashSlots[3] = {STASH_SLOTS_K0, STASH_SLOTS_K1, STASH_SLOTS_K2};

struct Scratch {   // header of ctx.small (device) and layout of ctx.pinned (host mirror)
  uint32_t ticket; uint32_t pad0;      // --- [0, kPerPass): re-cleared before every pass of a multi-pass filter
  unsigned long long total;
  uint32_t ticket2; uint32_t pad1;
  unsigned long long err;             // --- from here on: cleared once per call (errors accumulate over the passes)
  unsigned long long total_bytes;
  unsigned long long counters[24];
  int32_t utf8_ends[16];              // first / last input offset of each Utf8 column (output byte capacity)
  unsigned long long fold_bytes[MAX_FOLD_UTF8];   // output bytes of the Utf8 columns filtered inside the main kernel
};
constexpr size_t kPerPass = 24;
constexpr size_t kHeader = 512;       // status words start here
static_assert(sizeof(Scratch) <= kHeader, "scratch header");

void ensure_scratch(Context& ctx, int64_t ntiles) {
  if (!ctx.small || ctx.small_tiles < (size_t)ntiles + 64) {
    ctx.small_tiles = (size_t)ntiles + 64 + (size_t)ntiles / 4;
    ctx.small = make_device_buffer(kHeader + ctx.small_tiles * 8, ctx.device);
  }
  if (!ctx.pinned) { check_hip(hipHostMalloc(&ctx.pinned, sizeof(Scratch), hipHostMallocDefault), "hipHostMalloc"); ctx.pinned_bytes = sizeof(Scratch); }
}
Scratch* dev_scratch(Context& ctx) { return (Scratch*)ctx.small->ptr; }
u64* dev_status(Context& ctx) { return (u64*)((uint8_t*)ctx.small->ptr + kHeader); }

// FAST_UOPS (device_program.h): pre-decode a program whose every instruction works on non-null Int32 / UInt32 / Float32
// columns, 32-bit literals and boolean temporaries into (operand kind, loop body) pairs.  Returns false -- the generic
// interpreter runs -- as soon as one instruction falls outside that set.
bool encode_fast_uops(ProgramBlock& pb, const Lowered& lw, const Batch& rec) {
  if (lw.wide || lw.num_temps > 0 || !lw.strs.empty() || lw.prog.empty()) return false;
  auto is32 = [](int t) { return t == T_I32 || t == T_U32 || t == T_F32; };
  bool nullable_ref = false;
  for (int ci : lw.refs) {
    const Column& c = rec.cols[ci];
    if (!is32(c.type)) return false;
    nullable_ref |= c.validity && c.null_count != 0;
  }
  if (nullable_ref) {
    // columns WITH nulls: only `column <cmp> literal` (LOAD col; CMP const) -- the predicate of most sample queries, and Parquet
    // `optional` columns with real nulls are the normal case off read_files -- keeps the fast evaluators (they AND the column's
    // validity into the result; their boolean temporaries carry none, so anything longer takes the generic interpreter)
    const bool cmp_const = lw.prog.size() == 2 && lw.refs.size() == 1 && lw.prog[0].op == OP_LOAD && lw.prog[0].src_kind == SRC_COL &&
                           lw.prog[0].src_type == lw.prog[0].type && lw.prog[1].op >= OP_EQ && lw.prog[1].op <= OP_GE &&
                           lw.prog[1].src_kind == SRC_CONST && lw.prog[1].type == lw.prog[0].type;
    if (!cmp_const) return false;
  }
  for (size_t i = 0; i < lw.prog.size(); ++i) {
    const Instr& in = pb.prog[i];   // (never modified: incomplete waves run the same program through the generic interpreter)
    const bool rev = in.flags & IF_REV;
    uint8_t opd = FO_NONE, fop = FU_NOPS;
    bool negate = false;
    // ---- operand ----
    if (in.src_kind == SRC_COL) {
      if (!is32(in.src_type)) return false;
      if (in.src_type == in.type) opd = FO_COL;
      else if (in.type == T_F32 && in.src_type == T_I32) opd = FO_COL_I2F;
      else if (in.type == T_F32 && in.src_type == T_U32) opd = FO_COL_U2F;
      else return false;
    } else if (in.src_kind == SRC_CONST) {
      if (!is32(in.type) && !(in.op == OP_LOAD && in.type == T_BOOL)) return false;
      if (in.type == T_BOOL) return false;   // boolean literals take the generic path (length rules make them rare)
      opd = FO_CONST;
    } else if (in.src_kind == SRC_TEMP) {
      if (in.src_type != T_BOOL || in.src_idx >= MAX_BOOL_TEMPS) return false;
      opd = FO_BTEMP;
    }
    // ---- operation ----
    const uint32_t c = (uint32_t)in.imm;
    const bool pow2 = opd == FO_CONST && !rev && c != 0 && c <= 0x40000000u && (c & (c - 1)) == 0;
    switch (in.op) {
      case OP_LOAD:
        if (opd == FO_NONE) return false;
        if (opd == FO_BTEMP && in.type != T_BOOL) return false;
        fop = FU_LD;
        break;
      case OP_ADD: case OP_MUL: case OP_SUB: case OP_DIV: case OP_REM: {
        if (opd == FO_BTEMP || opd == FO_NONE) return false;
        const bool add = in.op == OP_ADD, mul = in.op == OP_MUL, sub = in.op == OP_SUB, div = in.op == OP_DIV;
        if (in.type == T_F32) {
          if (add) fop = FU_ADD_F; else if (mul) fop = FU_MUL_F; else if (sub) fop = rev ? FU_RSUB_F : FU_SUB_F;
          else if (div) fop = rev ? FU_RDIV_F : FU_DIV_F; else return false;   // fmod: generic path
        } else if (in.type == T_I32 || in.type == T_U32) {
          const bool s = in.type == T_I32;
          if (add) fop = s ? FU_ADD_I : FU_ADD_U; else if (mul) fop = s ? FU_MUL_I : FU_MUL_U;
          else if (sub) fop = rev ? (s ? FU_RSUB_I : FU_RSUB_U) : (s ? FU_SUB_I : FU_SUB_U);
          else if (pow2) fop = div ? (s ? FU_DIVP2_I : FU_DIVP2_U) : (s ? FU_REMP2_I : FU_REMP2_U);
          else return false;   // general integer division: generic path
        } else return false;
      } break;
      case OP_EQ: case OP_NE: case OP_LT: case OP_LE: case OP_GT: case OP_GE: {
        if (opd == FO_BTEMP || opd == FO_NONE || !is32(in.type)) return false;
        // primitives on (x = accumulator, y = operand): EQ, LT (x < y), GT (x > y); `rev` = operand (op) accumulator
        bool lt = false;   // else gt
        switch (in.op) {
          case OP_EQ: fop = FU_EQ; break;
          case OP_NE: fop = FU_EQ; negate = true; break;
          case OP_LT: lt = !rev; fop = 1; break;
          case OP_GT: lt = rev; fop = 1; break;
          case OP_LE: lt = rev; negate = true; fop = 1; break;    // x <= y == !(x > y)
          default: lt = !rev; negate = true; fop = 1; break;      // x >= y == !(x < y)
        }
        if (fop == 1) {
          if (in.type == T_I32) fop = lt ? FU_LT_I : FU_GT_I;
          else if (in.type == T_U32) fop = lt ? FU_LT_U : FU_GT_U;
          else if (opd == FO_CONST && (int32_t)c >= 0) fop = lt ? FU_LT_I : FU_GT_I;   // raw bits order like the keys (kernels.hip: run_cmp_const)
          else if (opd == FO_CONST) fop = lt ? FU_LT_FKC : FU_GT_FKC;   // (the device keys the literal once per instruction)
          else fop = lt ? FU_LT_F : FU_GT_F;
        }
      } break;
      case OP_AND: case OP_OR:
        if (opd != FO_BTEMP) return false;
        fop = in.op == OP_AND ? FU_AND : FU_OR;
        break;
      case OP_SPILL:
        if (in.type != T_BOOL || in.src_idx >= MAX_BOOL_TEMPS) return false;
        fop = FU_SPILL; opd = FO_NONE;
        break;
      case OP_CAST:
        if (in.type == T_F32 && in.src_type == T_I32) fop = FU_CVT_I2F;
        else if (in.type == T_F32 && in.src_type == T_U32) fop = FU_CVT_U2F;
        else return false;
        opd = FO_NONE;
        break;
      case OP_STORE: fop = FU_STORE; opd = FO_NONE; break;
      default: return false;
    }
    pb.fast_op[i] = (uint8_t)(fop | (negate ? FU_NEGATE : 0));
    pb.fast_opd[i] = opd;
  }
  return true;
}

void fill_refs(ProgramBlock& pb, const Lowered& lw, const Batch& rec, const std::vector<BufferPtr>& str_bufs) {
  pb.n_instr = (int32_t)lw.prog.size();
  pb.n_refs = (int32_t)lw.refs.size();
  pb.fast_kind = FAST_NONE;
  for (size_t i = 0; i < lw.prog.size(); ++i) pb.prog[i] = lw.prog[i];
  // Programs over non-null 32-bit columns are pre-decoded for the FASTK kernels (device_program.h); among them the shape
  // [LOAD col:T] [CMP literal] on a column of exactly the compare type has its own even leaner device path.
  if (encode_fast_uops(pb, lw, rec)) {
    pb.fast_kind = FAST_UOPS;
    if (lw.prog.size() == 2 && lw.refs.size() == 1 && lw.prog[0].op == OP_LOAD && lw.prog[0].src_kind == SRC_COL &&
        lw.prog[0].src_type == lw.prog[0].type && lw.prog[1].op >= OP_EQ && lw.prog[1].op <= OP_GE &&
        lw.prog[1].src_kind == SRC_CONST && lw.prog[1].type == lw.prog[0].type)
      pb.fast_kind = FAST_CMP_CONST;
  }
  for (size_t i = 0; i < lw.refs.size(); ++i) {
    const Column& c = rec.cols[lw.refs[i]];
    ColRef r{};
    r.values = c.values0();
    r.validity = (c.validity && c.null_count != 0) ? c.validity : nullptr;
    r.data = c.data;
    r.validity_bit_offset = c.offset;
    r.bool_bit_offset = c.offset;
    // a temporal column inside a program is one side of a same-type comparison: its values ARE Int32 / Int64 (plan.cpp)
    r.type = c.type == T_FIXED_OPAQUE ? (c.width == 4 ? T_I32 : T_I64) : c.type;
    pb.refs[i] = r;
  }
  for (size_t i = 0; i < lw.strs.size(); ++i) { pb.strs[i].bytes = (const uint8_t*)str_bufs[i]->ptr; pb.strs[i].len = (int64_t)lw.strs[i].size(); }
}

std::vector<BufferPtr> upload_strings(Context& ctx, const Lowered& lw) {
  std::vector<BufferPtr> out;
  for (const auto& s : lw.strs) {
    auto b = make_device_buffer(s.size() + 8, ctx.device);
    if (!s.empty()) check_hip(hipMemcpyAsync(b->ptr, s.data(), s.size(), hipMemcpyHostToDevice, ctx.stream), "upload string literal");
    out.push_back(b);
  }
  // literals live on the host stack of the caller: make the copies land before returning to it
  if (!out.empty()) check_hip(hipStreamSynchronize(ctx.stream), "sync");
  return out;
}

[[noreturn]] void throw_device_error(unsigned long long stored) {
  const unsigned long long err = ~stored;   // the device keeps the complement (see ERR_NONE)
  const int code = (int)(err & 0xff);
  const long long row = (long long)((err >> 8) & ((1ULL << 48) - 1));
  if (code == DE_DIV_ZERO) throw ChqError{CHQ_ERR_ARROW_DIVIDE_BY_ZERO, "Divide by zero error (row " + std::to_string(row) + ")"};
  throw ChqError{CHQ_ERR_ARROW_ARITHMETIC_OVERFLOW, "Overflow happened on row " + std::to_string(row)};
}

int pick_tile_kind(const Context& ctx, const Lowered& lw, int64_t rows) {
  if (lw.wide || lw.num_temps > 0) return 2;
  if (ctx.opt_tile_kind >= 0) return (int)ctx.opt_tile_kind;
  return rows >= (1 << 18) ? 0 : 1;
}

// (a Float16 operand reaches the stash widened to f32: not the column's bytes)
bool stashable(const Column& c) { return c.type != T_BOOL && c.type != T_UTF8 && c.type != T_F16 && c.width > 0 && c.width <= 4; }

// Up to `slots` narrow predicate input columns stay on chip between the predicate and copy phases (device_program.h:
// FilterParams::stash_refs); they move to the end of the launch's column order, slot k <-> the k-th of them.
void pick_stash(FilterParams& p, const Context& ctx, const Lowered& lw, const std::vector<Column>& cols, std::vector<int>& launch_cols, int tile_kind) {
  const int slots = std::min<int>(kStashSlots[tile_kind], ctx.opt_stash < 0 ? MAX_STASH : (int)ctx.opt_stash);
  p.n_stash = 0;
  std::vector<int> chosen;
  for (size_t r = 0; r < lw.refs.size() && r < 127 && p.n_stash < slots; ++r) {
    const int ci = lw.refs[r];
    if ((size_t)ci >= cols.size() || !stashable(cols[ci])) continue;   // (a temporary column is not an output column)
    if (std::find(chosen.begin(), chosen.end(), ci) != chosen.end()) continue;
    auto it = std::find(launch_cols.begin(), launch_cols.end(), ci);
    if (it == launch_cols.end()) continue;
    launch_cols.erase(it); launch_cols.push_back(ci);
    chosen.push_back(ci);
    p.stash_refs[p.n_stash++] = (int8_t)r;
  }
}

std::vector<Column> evaluate_dense(Context& ctx, const Batch& rec, const std::vector<PlanColumn>& pcols,
                                   const std::vector<const TypedExpr*>& exprs);

// ---- expressions that do not fit one device program ------------------------------------------------------------
// The reference has no size limits (one arrow kernel per AST node).  When lower_expr reports that a typed tree needs
// more instructions / columns / temporaries than a program holds, sub-trees are evaluated into temporary columns
// (appended to `work` / `wcols`, never part of any output) and replaced by column nodes, bottom-up and left to right
// -- i.e. in the reference's own evaluation order -- until the rest fits.  `strict` materialises EVERY inner node in
// that order (exactly the reference's strategy): used to find the first data-dependent error when a materialisation
// launch reports one, because a temporary may have been evaluated ahead of a smaller sub-tree to its left.
bool lowers_alone(const TypedExpr& te, int node, const std::vector<PlanColumn>& wcols) {
  try {
    Lowered trial;
    lower_expr(te, node, wcols, trial);
    return (int)trial.prog.size() + 1 <= MAX_INSTR;   // room for the STORE of a materialisation
  } catch (const ChqError& e) {
    if (e.code == CHQ_INTERNAL_PROGRAM_LIMIT) return false;
    throw;
  }
}
bool is_leaf_node(const Node& n) { return n.kind == Node::COL || n.kind == Node::CONST; }

void materialize_node(Context& ctx, Batch& work, std::vector<PlanColumn>& wcols, TypedExpr& te, int node) {
  TypedExpr sub;
  sub.nodes = te.nodes; sub.root = node;
  std::vector<const TypedExpr*> one{&sub};
  const chq_call_stats keep = ctx.stats;
  std::vector<Column> cols = evaluate_dense(ctx, work, wcols, one);
  ctx.stats = keep;
  Column c = std::move(cols[0]);
  c.name = "__chq_tmp_" + std::to_string(work.cols.size());
  PlanColumn pc;
  pc.name = c.name; pc.type = c.type; pc.has_nulls = c.validity && c.null_count != 0; pc.alias_entry_present = true;
  Node repl{};
  repl.kind = Node::COL; repl.type = te.nodes[node].type; repl.is_scalar = false; repl.len1 = false;
  repl.col = (int)work.cols.size(); repl.ref_order = te.nodes[node].ref_order;
  work.cols.push_back(std::move(c));
  wcols.push_back(std::move(pc));
  te.nodes[node] = repl;
}

// Decimal128 comparisons and Utf8 -> Boolean casts are not device-program instructions (plan.hpp: Node): their own
// kernels (typed_ops.hip) write a temporary Boolean column, which replaces the node like any other materialisation.
bool is_typed_op(const TypedExpr& te, int node) {
  const Node& n = te.nodes[node];
  return (n.kind == Node::CMP && n.from == T_FIXED_OPAQUE) || (n.kind == Node::TOBOOL && n.from == T_UTF8) ||
         (n.kind == Node::CONST && n.cval.null);   // `'maybe' AND ..` folded to NULL, next to a column of a one-row batch
}
void materialize_typed_op(Context& ctx, Batch& work, std::vector<PlanColumn>& wcols, TypedExpr& te, int node) {
  const Node n = te.nodes[node];
  const int64_t nrows = work.nrows;
  const size_t words = (size_t)((nrows + 63) / 64) + 1;
  auto bits = make_device_buffer(words * 8, ctx.device), valid = make_device_buffer(words * 8, ctx.device);
  auto count = make_device_buffer(16, ctx.device);
  check_hip(hipMemsetAsync(count->ptr, 0, 16, ctx.stream), "memset");
  if (n.kind == Node::CONST) {
    check_hip(hipMemsetAsync(bits->ptr, 0, words * 8, ctx.stream), "memset");
    check_hip(hipMemsetAsync(valid->ptr, 0, words * 8, ctx.stream), "memset");
    const u64 all = (u64)nrows;
    check_hip(hipMemcpyAsync(count->ptr, &all, 8, hipMemcpyHostToDevice, ctx.stream), "null count");
  } else if (n.kind == Node::CMP) {
    const Column& a = work.cols[(size_t)te.nodes[n.l].col];
    const Column& b = work.cols[(size_t)te.nodes[n.r].col];
    Cmp128Params p{};
    p.a = a.values0(); p.b = b.values0();
    p.a_validity = (a.validity && a.null_count != 0) ? a.validity : nullptr; p.a_validity_offset = a.offset;
    p.b_validity = (b.validity && b.null_count != 0) ? b.validity : nullptr; p.b_validity_offset = b.offset;
    p.nrows = nrows; p.op = n.op;
    p.out_bits = (u64*)bits->ptr; p.out_validity = (u64*)valid->ptr; p.null_count = (u64*)count->ptr;
    if (nrows > 0) check_hip(launch_cmp128(p, ctx.stream), "launch cmp128_kernel");
  } else {
    const Column& c = work.cols[(size_t)te.nodes[n.l].col];
    Utf8ToBoolParams p{};
    p.offsets = (const int32_t*)c.values0(); p.data = c.data;
    p.validity = (c.validity && c.null_count != 0) ? c.validity : nullptr; p.validity_offset = c.offset;
    p.nrows = nrows;
    p.out_bits = (u64*)bits->ptr; p.out_validity = (u64*)valid->ptr; p.null_count = (u64*)count->ptr;
    if (nrows > 0) check_hip(launch_utf8_to_bool(p, ctx.stream), "launch utf8_to_bool_kernel");
  }
  u64 nulls = 0;
  check_hip(hipMemcpyAsync(&nulls, count->ptr, 8, hipMemcpyDeviceToHost, ctx.stream), "read null count");
  check_hip(hipStreamSynchronize(ctx.stream), "sync");
  Column c;
  c.name = "__chq_tmp_" + std::to_string(work.cols.size());
  c.format = "b"; c.type = T_BOOL; c.width = 0; c.length = nrows; c.null_count = (int64_t)nulls; c.nullable = nulls != 0;
  c.values = (const uint8_t*)bits->ptr; c.validity = nulls ? (const uint8_t*)valid->ptr : nullptr;
  c.owned = {bits, valid};
  PlanColumn pc;
  pc.name = c.name; pc.type = T_BOOL; pc.has_nulls = nulls != 0; pc.alias_entry_present = true; pc.format = "b";
  Node repl{};
  repl.kind = Node::COL; repl.type = T_BOOL; repl.is_scalar = false; repl.len1 = false;
  repl.col = (int)work.cols.size(); repl.ref_order = n.ref_order;
  work.cols.push_back(std::move(c));
  wcols.push_back(std::move(pc));
  te.nodes[node] = repl;
}

void fit_subtree(Context& ctx, Batch& work, std::vector<PlanColumn>& wcols, TypedExpr& te, int node, bool strict, bool is_root) {
  if (is_typed_op(te, node)) { materialize_typed_op(ctx, work, wcols, te, node); return; }
  if (is_leaf_node(te.nodes[node])) return;
  const int l = te.nodes[node].l, r = te.nodes[node].r;
  if (l >= 0) fit_subtree(ctx, work, wcols, te, l, strict, false);
  if (r >= 0) fit_subtree(ctx, work, wcols, te, r, strict, false);
  if (strict) { if (!is_root) materialize_node(ctx, work, wcols, te, node); return; }
  if (lowers_alone(te, node, wcols)) return;
  // both children fit on their own but not together with this node: turn them into columns, left first
  if (l >= 0 && !is_leaf_node(te.nodes[l])) materialize_node(ctx, work, wcols, te, l);
  if (!lowers_alone(te, node, wcols) && r >= 0 && !is_leaf_node(te.nodes[r])) materialize_node(ctx, work, wcols, te, r);
}

// After this call lower_expr(te, te.root) succeeds.  `work` / `wcols` start as copies of the batch and its plan columns.
void fit_to_device(Context& ctx, const Batch& rec, const std::vector<PlanColumn>& pcols, const TypedExpr& original,
                   Batch& work, std::vector<PlanColumn>& wcols, TypedExpr& te) {
  for (int pass = 0; pass < 2; ++pass) {
    work = rec; wcols = pcols; te = original;
    try {
      fit_subtree(ctx, work, wcols, te, te.root, /*strict=*/pass == 1, true);
      return;
    } catch (const ChqError& e) {
      const bool data_error = e.code == CHQ_ERR_ARROW_ARITHMETIC_OVERFLOW || e.code == CHQ_ERR_ARROW_DIVIDE_BY_ZERO;
      if (pass == 1 || !data_error) throw;   // strict order reports the reference's first error
    }
  }
}

// type_expr + the reference's error order: data-dependent errors of subtrees evaluated before a static
// error take precedence over it
TypedExpr typed(Context& ctx, const Batch& rec, const std::vector<PlanColumn>& pcols, const Expr& expr) {
  TypedExpr te = type_expr(expr, pcols, rec.nrows, ctx.opt_enable_minus);
  if (te.pending_code) {
    if (rec.nrows > 0) {
      std::vector<TypedExpr> subs;
      for (int r : te.validate_roots) { TypedExpr s; s.nodes = te.nodes; s.root = r; subs.push_back(std::move(s)); }
      std::vector<const TypedExpr*> ptrs;
      for (auto& s : subs) ptrs.push_back(&s);
      (void)evaluate_dense(ctx, rec, pcols, ptrs);   // throws the data-dependent error if there is one
    }
    throw ChqError{te.pending_code, te.pending_msg};
  }
  return te;
}

Column empty_like(const Column& c) {
  Column o;
  o.name = c.name; o.format = c.format; o.type = c.type; o.width = c.width; o.nullable = c.nullable;
  return o;
}

}  // namespace

// =================================================================================================
// filter_record
// =================================================================================================
Batch filter_record(Context& ctx, const Batch& rec, const std::vector<PlanColumn>& pcols, const Expr& expr, SplitRequest* split) {
  const int64_t nrows = rec.nrows;
  TypedExpr te = typed(ctx, rec, pcols, expr);
  if (te.nodes[(size_t)te.root].kind == Node::CONST && te.nodes[(size_t)te.root].cval.null) {
    Scalar& v = te.nodes[(size_t)te.root].cval;   // a NULL mask slot drops its row (prep_null_mask_filter)
    v.null = false; v.bits = 0;
  }
  const Node& root = te.at(te.root);
  if (root.type != T_BOOL) {   // RU/filter_record.rs:27-35
    // the reference has already evaluated the expression at this point: data-dependent errors come first
    if (root.kind != Node::COL && !root.len1 && nrows > 0) { std::vector<const TypedExpr*> v{&te}; (void)evaluate_dense(ctx, rec, pcols, v); }
    throw ChqError{CHQ_ERR_CAST_TO_BOOLEAN_ARRAY_FAILED, std::string("cast to boolean array failed for array type: ") + dtype_name(root.type)};
  }
  // A literal-only predicate is a length-1 mask: arrow filters just the first row (and rejects a mask
  // longer than the columns) -- reproduced, not "fixed" (SURVEY.md section 8 a8).
  const int64_t mask_len = root.len1 ? 1 : nrows;
  if (mask_len > nrows && !rec.cols.empty())
    throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Filter predicate of length " + std::to_string(mask_len) +
                                                       " is larger than target array of length " + std::to_string(nrows)};
  Lowered lw;
  Batch work; std::vector<PlanColumn> wcols; TypedExpr fitted;
  const Batch* prog_rec = &rec;   // what the program's column refs index: the batch, or the batch + temporary columns
  try {
    lower_expr(te, te.root, pcols, lw);
  } catch (const ChqError& e) {
    if (e.code != CHQ_INTERNAL_PROGRAM_LIMIT) throw;
    fit_to_device(ctx, rec, pcols, te, work, wcols, fitted);   // sub-trees -> temporary columns until the rest fits
    lw = Lowered{};
    lower_expr(fitted, fitted.root, wcols, lw);
    prog_rec = &work;
  }

  // ---- Utf8 columns whose values all have the same length (keys, hashes, dates as text, the reference's own sample
  // strings -- create_sample_data.rs) are fixed-width columns in disguise: value i lies at data + offsets[0] + i L.  One
  // cheap pass over the offsets proves it (4 B/row); the column then goes through the fixed-width copy of the main kernel
  // (whole 8- or 16-byte values per lane) instead of the per-row string scatter, and its new offsets are 0, L, 2 L, ...
  // Config-5 shape: 0.99 -> 0.6 ms per 125 M-row batch.  Only columns the predicate does not read, without nulls.
  if (prog_rec == &rec && mask_len == nrows && ctx.opt_uniform_utf8_rows > 0 && nrows >= ctx.opt_uniform_utf8_rows && rec.on_device) {
    std::vector<int> cand;
    for (size_t i = 0; i < rec.cols.size(); ++i) {
      const Column& c = rec.cols[i];
      if (c.type != T_UTF8 || !c.values || !c.data || (c.validity && c.null_count != 0)) continue;
      if (std::find(lw.refs.begin(), lw.refs.end(), (int)i) != lw.refs.end()) continue;
      cand.push_back((int)i);
    }
    if (!cand.empty()) {
      // (with `time_kernels` the check and the offsets kernels are timed too and added to the call's kernel time: the
      // roofline of this path must not be flattered by leaving its extra passes out)
      hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};
      struct EventGuard { hipEvent_t* e; ~EventGuard() { for (int i = 0; i < 4; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } tev_guard{tev};
      if (ctx.opt_time_kernels) for (auto& e : tev) check_hip(hipEventCreate(&e), "hipEventCreate");
      auto d_chk = make_device_buffer(cand.size() * 16 + 16, ctx.device);
      check_hip(hipMemsetAsync(d_chk->ptr, 0, cand.size() * 16, ctx.stream), "memset");
      if (tev[0]) check_hip(hipEventRecord(tev[0], ctx.stream), "hipEventRecord");
      for (size_t k = 0; k < cand.size(); ++k) {
        Utf8UniformParams up{(const int32_t*)rec.cols[(size_t)cand[k]].values0(), nrows, (int32_t*)d_chk->ptr + 4 * k};
        check_hip(launch_utf8_uniform(up, ctx.stream), "launch utf8_uniform_kernel");
      }
      if (tev[1]) check_hip(hipEventRecord(tev[1], ctx.stream), "hipEventRecord");
      std::vector<int32_t> h_chk(cand.size() * 4);
      check_hip(hipMemcpyAsync(h_chk.data(), d_chk->ptr, cand.size() * 16, hipMemcpyDeviceToHost, ctx.stream), "read back");
      check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
      Batch view = rec;
      std::vector<PlanColumn> vcols = pcols;
      std::vector<int> turned;
      for (size_t k = 0; k < cand.size(); ++k) {
        const int32_t differs = h_chk[4 * k], len = h_chk[4 * k + 1], first = h_chk[4 * k + 2];
        if (differs || !(len == 1 || len == 2 || len == 4 || len == 8 || len == 16) || nrows * (int64_t)len >= (1ll << 31) - 64 || first < 0) continue;
        Column& v = view.cols[(size_t)cand[k]];
        v.type = T_FIXED_OPAQUE; v.format = "w:" + std::to_string(len); v.width = len;
        v.values = v.data + first; v.data = nullptr; v.data_bytes = -1; v.offset = 0; v.validity = nullptr; v.null_count = 0;
        vcols[(size_t)cand[k]].type = T_FIXED_OPAQUE; vcols[(size_t)cand[k]].format = v.format; vcols[(size_t)cand[k]].width = len;
        vcols[(size_t)cand[k]].has_nulls = false;
        turned.push_back(cand[k]);
      }
      if (!turned.empty()) {
        Batch res = filter_record(ctx, view, vcols, expr, split);   // (no eligible Utf8 column is left in the view: no further recursion)
        if (tev[2]) check_hip(hipEventRecord(tev[2], ctx.stream), "hipEventRecord");
        for (int ci : turned) {
          Column& o = res.cols[(size_t)ci];
          const Column& c = rec.cols[(size_t)ci];
          const int32_t len = o.width;
          auto ob = make_device_buffer((size_t)(res.nrows + 1) * 4 + 16, ctx.device);
          IotaOffsetsParams ip{(int32_t*)ob->ptr, res.nrows + 1, len, 0};
          check_hip(launch_iota_offsets(ip, ctx.stream), "launch iota_offsets_kernel");
          Column u = empty_like(c);
          u.length = res.nrows; u.null_count = 0; u.offset = 0;
          u.data = o.values; u.data_bytes = res.nrows * (int64_t)len;
          u.values = (const uint8_t*)ob->ptr;
          u.owned = std::move(o.owned); u.owned.push_back(ob);
          o = std::move(u);
          ctx.stats.bytes_read_alg += (nrows + 1) * 4;             // the offsets were read (by the check) ...
          ctx.stats.bytes_written_alg += (res.nrows + 1) * 4;      // ... and written
        }
        if (tev[3]) check_hip(hipEventRecord(tev[3], ctx.stream), "hipEventRecord");
        check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
        if (tev[0]) {
          float a = 0, b = 0;
          check_hip(hipEventElapsedTime(&a, tev[0], tev[1]), "hipEventElapsedTime");
          check_hip(hipEventElapsedTime(&b, tev[2], tev[3]), "hipEventElapsedTime");
          ctx.stats.kernel_ns += (int64_t)((a + b) * 1e6);
        }
        ctx.stats.launches += (int64_t)(cand.size() + turned.size());
        return res;
      }
    }
  }

  Batch out;
  out.on_device = true; out.device_id = ctx.device;
  ctx.stats = chq_call_stats{};
  ctx.stats.rows_in = nrows;
  for (const Column& c : rec.cols) out.cols.push_back(empty_like(c));
  if (mask_len == 0) {   // empty in, empty out (schema preserved)
    for (size_t i = 0; i < rec.cols.size(); ++i) {
      Column& o = out.cols[i];
      auto vb = make_device_buffer(16, ctx.device);
      check_hip(hipMemsetAsync(vb->ptr, 0, 16, ctx.stream), "memset");
      o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
      if (o.type == T_UTF8) { auto db = make_device_buffer(16, ctx.device); o.data = (const uint8_t*)db->ptr; o.owned.push_back(db); }
    }
    check_hip(hipStreamSynchronize(ctx.stream), "sync");
    out.nrows = 0;
    return out;
  }

  const int tile_kind = pick_tile_kind(ctx, lw, mask_len);
  const int64_t tile_rows = kTileRows[tile_kind];
  const int64_t ntiles = (mask_len + tile_rows - 1) / tile_rows;
  ensure_scratch(ctx, ntiles);
  Scratch* ds = dev_scratch(ctx);
  auto str_bufs = upload_strings(ctx, lw);

  // classify columns
  std::vector<int> fixed_cols, bool_cols, utf8_cols, nullable_cols;
  for (size_t i = 0; i < rec.cols.size(); ++i) {
    const Column& c = rec.cols[i];
    if (c.type == T_BOOL) bool_cols.push_back((int)i);
    else if (c.type == T_UTF8) utf8_cols.push_back((int)i);
    else fixed_cols.push_back((int)i);
    if (c.validity && c.null_count != 0) nullable_cols.push_back((int)i);
  }
  Scratch* hs = (Scratch*)ctx.pinned;
  // Utf8 columns of short strings are filtered inside the main kernel (device_program.h: Utf8Fold).  Their output
  // capacity is the input byte span: known when the library built the column itself (staged, joined, decoded), one
  // 8-byte read-back otherwise.
  // Long strings (more than 24 bytes per row on average) get only their new offsets from the main kernel; the bytes are
  // moved by utf8_copy_kernel (one wave per 64 rows), launched right behind it without a host round trip in between.
  std::vector<int> fold_cols; std::vector<int64_t> fold_cap; std::vector<bool> fold_data;
  if (ctx.opt_fold_utf8 && tile_kind != 2 && !utf8_cols.empty() && mask_len == nrows) {
    const size_t ncand = std::min<size_t>(MAX_FOLD_UTF8, utf8_cols.size());
    std::vector<int64_t> cap(ncand, -1);
    bool unknown = false;
    for (size_t k = 0; k < ncand; ++k) { cap[k] = rec.cols[utf8_cols[k]].data_bytes; unknown |= cap[k] < 0; }
    if (unknown) {
      GatherParams gp{};
      for (size_t k = 0; k < ncand; ++k) {
        const int32_t* offs = (const int32_t*)rec.cols[utf8_cols[k]].values0();
        gp.src[2 * k] = offs; gp.src[2 * k + 1] = offs + mask_len;
      }
      gp.n = (int32_t)(2 * ncand); gp.dst = ds->utf8_ends;
      check_hip(launch_gather_i32(gp, ctx.stream), "launch gather_i32_kernel");
      check_hip(hipMemcpyAsync(hs->utf8_ends, ds->utf8_ends, sizeof(hs->utf8_ends), hipMemcpyDeviceToHost, ctx.stream), "read back");
      check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
      for (size_t k = 0; k < ncand; ++k) cap[k] = (int64_t)hs->utf8_ends[2 * k + 1] - hs->utf8_ends[2 * k];
    }
    std::vector<int> rest;
    for (size_t k = 0; k < utf8_cols.size(); ++k) {
      if (k < ncand) { fold_cols.push_back(utf8_cols[k]); fold_cap.push_back(cap[k]); fold_data.push_back(cap[k] <= mask_len * 24); }
      else rest.push_back(utf8_cols[k]);
    }
    utf8_cols.swap(rest);
  }
  std::vector<BufferPtr> fold_status;
  const bool fold_long = std::find(fold_data.begin(), fold_data.end(), false) != fold_data.end();
  const bool need_followup = !bool_cols.empty() || !utf8_cols.empty() || !nullable_cols.empty() || (int)fixed_cols.size() > MAX_OUT ||
                             (split && !split->starts.empty()) || fold_long;
  const int64_t ngroups = (mask_len + 63) / 64;
  BufferPtr sel_mask, grp_base;
  if (need_followup) {
    sel_mask = make_device_buffer((size_t)(ngroups + 2) * 8, ctx.device);
    grp_base = make_device_buffer((size_t)(ngroups + 2) * 8, ctx.device);
  }

  // output buffers for fixed-width columns: capacity = mask_len rows
  for (int ci : fixed_cols) {
    Column& o = out.cols[ci];
    auto vb = make_device_buffer((size_t)mask_len * o.width + 16, ctx.device);
    o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
    ctx.stats.bytes_read_alg += mask_len * o.width;
  }
  // predicate inputs that are not output columns cannot occur for filter_record (SELECT * semantics):
  // every referenced column is also copied, so it is counted once above.

  const int grid_cap = ctx.num_cus * (ctx.opt_grid_per_cu > 0 ? (int)ctx.opt_grid_per_cu : kGridPerCu[tile_kind]);
  const int grid = (int)std::min<int64_t>(ntiles, grid_cap);
  size_t next_fixed = 0;
  bool first = true;
  if (ctx.opt_time_kernels && !ctx.ev0) { check_hip(hipEventCreate(&ctx.ev0), "hipEventCreate"); check_hip(hipEventCreate(&ctx.ev1), "hipEventCreate"); }

  do {
    FilterParams p{};
    p.nrows = mask_len;
    p.status = dev_status(ctx);
    p.ticket = &ds->ticket; p.total = &ds->total; p.err = &ds->err;
    p.sel_mask = (first && need_followup) ? (u64*)sel_mask->ptr : nullptr;
    p.grp_base = (first && need_followup) ? (u64*)grp_base->ptr : nullptr;
    if (first) fill_refs(p.pb, lw, *prog_rec, str_bufs);
    else {   // later passes re-read the selection bitmap as a Boolean column
      p.pb.n_instr = 1; p.pb.n_refs = 1;
      Instr in{}; in.op = OP_LOAD; in.type = T_BOOL; in.src_kind = SRC_COL; in.src_type = T_BOOL; in.src_idx = 0;
      p.pb.prog[0] = in;
      ColRef r{}; r.values = sel_mask->ptr; r.type = T_BOOL; p.pb.refs[0] = r;
    }
    std::vector<int> launch_cols;
    while (next_fixed < fixed_cols.size() && (int)launch_cols.size() < MAX_OUT) launch_cols.push_back(fixed_cols[next_fixed++]);
    // narrow predicate input columns stay on chip between the predicate and copy phases (placed last)
    p.n_stash = 0;
    if (first) pick_stash(p, ctx, lw, rec.cols, launch_cols, tile_kind);
    int n = 0;
    for (int ci : launch_cols) {
      p.outs[n].in = rec.cols[ci].values0(); p.outs[n].out = (void*)out.cols[ci].values; p.outs[n].width = (uint32_t)rec.cols[ci].width;
      ++n;
    }
    p.n_out = (int16_t)n;
    if (first) {
      for (size_t u = 0; u < fold_cols.size(); ++u) {
        const Column& c = rec.cols[fold_cols[u]];
        Column& o = out.cols[fold_cols[u]];
        auto offb = make_device_buffer((size_t)(mask_len + 2) * 4, ctx.device);
        auto db = make_device_buffer((size_t)fold_cap[u] + 16, ctx.device);
        auto st = make_device_buffer((size_t)(ntiles + 1) * 8, ctx.device);
        check_hip(hipMemsetAsync(st->ptr, 0, (size_t)(ntiles + 1) * 8, ctx.stream), "memset byte-scan status");
        fold_status.push_back(st);
        Utf8Fold& f = p.utf8[u];
        f.in_offsets = (const int32_t*)c.values0(); f.in_data = c.data;
        f.out_offsets = (int32_t*)offb->ptr; f.out_data = fold_data[u] ? (uint8_t*)db->ptr : nullptr;
        f.status = (u64*)st->ptr; f.total_bytes = &ds->fold_bytes[u];
        o.values = (const uint8_t*)offb->ptr; o.owned.push_back(offb);
        o.data = (const uint8_t*)db->ptr; o.owned.push_back(db);
        ctx.stats.bytes_read_alg += mask_len * 8;   // offsets, by both phases (as the separate Utf8 pass counts them)
      }
      p.n_utf8 = (int32_t)fold_cols.size();
    }
    if (first) check_hip(hipMemsetAsync(ds, 0, kHeader + (size_t)(ntiles + 1) * 8, ctx.stream), "memset scratch + status");
    else {
      check_hip(hipMemsetAsync(ds, 0, kPerPass, ctx.stream), "memset scratch");
      check_hip(hipMemsetAsync(dev_status(ctx), 0, (size_t)(ntiles + 1) * 8, ctx.stream), "memset status");
    }
    const int kind = first ? tile_kind : (tile_kind == 2 ? 1 : tile_kind);   // kinds 1 and 2 share a tile size
    if (ctx.opt_time_kernels && first) check_hip(hipEventRecord(ctx.ev0, ctx.stream), "hipEventRecord");
    // Large batches: all complete tiles run in the instantiation that contains no partial-tile code at all; the
    // (single) incomplete tail tile runs in a second one-workgroup launch that continues the same chained scan.
    const int64_t nfull = mask_len / tile_rows;
    if (mask_len >= ctx.opt_split_rows && nfull > 0) {
      p.tile_begin = 0; p.tile_end = nfull;
      check_hip(launch_filter(p, kind, false, (int)std::min<int64_t>(nfull, grid_cap), ctx.stream), "launch filter_fused_kernel");
      if (nfull < ntiles) {
        p.tile_begin = nfull; p.tile_end = ntiles; p.ticket = &ds->ticket2;
        check_hip(launch_filter(p, kind, true, 1, ctx.stream), "launch filter_fused_kernel (tail)");
        ++ctx.stats.launches;
      }
    } else {
      p.tile_begin = 0; p.tile_end = ntiles;
      check_hip(launch_filter(p, kind, true, grid, ctx.stream), "launch filter_fused_kernel");
    }
    if (ctx.opt_time_kernels && first) check_hip(hipEventRecord(ctx.ev1, ctx.stream), "hipEventRecord");
    ++ctx.stats.launches;
    if (first) {
      for (size_t u = 0; u < fold_cols.size(); ++u) {
        if (fold_data[u]) continue;
        Utf8Params up{};
        up.nrows = mask_len; up.sel_mask = (const u64*)sel_mask->ptr; up.grp_base = (const u64*)grp_base->ptr;
        up.in_offsets = p.utf8[u].in_offsets; up.in_data = p.utf8[u].in_data;
        up.out_offsets = p.utf8[u].out_offsets; up.out_data = (uint8_t*)out.cols[fold_cols[u]].data;
        check_hip(launch_utf8_copy(up, (int)std::min<int64_t>((ngroups + 3) / 4, (int64_t)ctx.num_cus * 16), ctx.stream), "launch utf8_copy_kernel");
        ++ctx.stats.launches;
      }
    }
    first = false;
  } while (next_fixed < fixed_cols.size());

  BufferPtr split_dev;
  if (split && !split->starts.empty()) {   // output position of every concatenated input batch
    const size_t n = split->starts.size();
    split_dev = make_device_buffer(n * 16 + 16, ctx.device);
    check_hip(hipMemcpyAsync(split_dev->ptr, split->starts.data(), n * 8, hipMemcpyHostToDevice, ctx.stream), "upload split rows");
    SplitBoundsParams sp{};
    sp.nrows = mask_len; sp.n = (int64_t)n; sp.starts = (const int64_t*)split_dev->ptr;
    sp.sel_mask = (const u64*)sel_mask->ptr; sp.grp_base = (const u64*)grp_base->ptr; sp.total = &ds->total;
    sp.out = (u64*)((uint8_t*)split_dev->ptr + n * 8);
    check_hip(launch_split_bounds(sp, ctx.stream), "launch split_bounds_kernel");
    split->bounds.assign(n, 0);
    check_hip(hipMemcpyAsync(split->bounds.data(), sp.out, n * 8, hipMemcpyDeviceToHost, ctx.stream), "read back split bounds");
  }
  // The follow-up kernels report through the fixed-size scratch header: 16 null counters and 8 Utf8 byte spans /
  // totals per round; wider batches simply take more rounds (each with its own read-back).
  constexpr size_t kNullPerRound = 16, kUtf8PerRound = 8;
  auto gather_utf8_ends = [&](size_t u0, size_t u1) {   // input byte span of Utf8 columns [u0, u1) = capacity of their outputs
    if (u1 <= u0) return;
    GatherParams gp{};
    for (size_t k = u0; k < u1; ++k) {
      const int32_t* offs = (const int32_t*)rec.cols[utf8_cols[k]].values0();
      gp.src[2 * (k - u0)] = offs; gp.src[2 * (k - u0) + 1] = offs + mask_len;
    }
    gp.n = (int32_t)(2 * (u1 - u0)); gp.dst = ds->utf8_ends;
    check_hip(launch_gather_i32(gp, ctx.stream), "launch gather_i32_kernel");
  };
  gather_utf8_ends(0, std::min(kUtf8PerRound, utf8_cols.size()));   // rides on the read-back of the row count
  check_hip(hipMemcpyAsync(hs, ds, sizeof(Scratch), hipMemcpyDeviceToHost, ctx.stream), "read back");
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  if (ctx.opt_time_kernels) { float ms = 0; check_hip(hipEventElapsedTime(&ms, ctx.ev0, ctx.ev1), "hipEventElapsedTime"); ctx.stats.kernel_ns = (int64_t)(ms * 1e6); }
  if (hs->err != ERR_NONE) throw_device_error(hs->err);
  const int64_t total = (int64_t)hs->total;
  out.nrows = total;
  ctx.stats.rows_out = total; ctx.stats.tiles = ntiles;
  for (int ci : fixed_cols) { out.cols[ci].length = total; ctx.stats.bytes_written_alg += total * out.cols[ci].width; }
  for (size_t u = 0; u < fold_cols.size(); ++u) {
    Column& o = out.cols[fold_cols[u]];
    o.length = total; o.data_bytes = (int64_t)hs->fold_bytes[u];
    ctx.stats.bytes_read_alg += o.data_bytes; ctx.stats.bytes_written_alg += (total + 1) * 4 + o.data_bytes;
  }

  if (need_followup) {
    const int fgrid = (int)std::min<int64_t>((ngroups + 31) / 32, (int64_t)ctx.num_cus * 8);
    // ---- Boolean value bitmaps and validity bitmaps -------------------------------------------------
    const size_t words = (size_t)(total + 31) / 32 + 2;
    auto bit_compact = [&](const uint8_t* in_bits, int64_t bit_off, u64* zero_counter) {
      auto ob = make_device_buffer(words * 4 + 8, ctx.device);
      check_hip(hipMemsetAsync(ob->ptr, 0, words * 4 + 8, ctx.stream), "memset bits");
      BitCompactParams bp{};
      bp.nrows = mask_len; bp.sel_mask = (const u64*)sel_mask->ptr; bp.grp_base = (const u64*)grp_base->ptr;
      bp.in_bits = in_bits; bp.in_bit_offset = bit_off; bp.out_bits = (uint32_t*)ob->ptr; bp.zero_count = zero_counter;
      check_hip(launch_bit_compact(bp, std::max(1, fgrid), ctx.stream), "launch bit_compact_kernel");
      ++ctx.stats.launches;
      return ob;
    };
    for (int ci : bool_cols) {
      auto ob = bit_compact(rec.cols[ci].values, rec.cols[ci].offset, nullptr);
      out.cols[ci].values = (const uint8_t*)ob->ptr; out.cols[ci].owned.push_back(ob); out.cols[ci].length = total;
    }
    std::vector<BufferPtr> byte_status(utf8_cols.size());
    size_t n0 = 0, u0 = 0;
    for (bool first_round = true; first_round || n0 < nullable_cols.size() || u0 < utf8_cols.size(); first_round = false) {
    const size_t n1 = std::min(n0 + kNullPerRound, nullable_cols.size()), u1 = std::min(u0 + kUtf8PerRound, utf8_cols.size());
    if (!first_round) {   // fresh counters, and the byte spans of this round's Utf8 columns
      check_hip(hipMemsetAsync(ds->counters, 0, sizeof(ds->counters), ctx.stream), "memset counters");
      if (u1 > u0) {
        gather_utf8_ends(u0, u1);
        check_hip(hipMemcpyAsync(hs, ds, sizeof(Scratch), hipMemcpyDeviceToHost, ctx.stream), "read back");
        check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
      }
    }
    for (size_t k = n0; k < n1; ++k) {
      const int ci = nullable_cols[k];
      auto ob = bit_compact(rec.cols[ci].validity, rec.cols[ci].offset, &ds->counters[k - n0]);
      out.cols[ci].validity = (const uint8_t*)ob->ptr; out.cols[ci].owned.push_back(ob);
    }
    // ---- Utf8 columns: one fused pass each (new offsets + bytes) --------------------------------------
    // Short strings: one fused pass (new offsets + bytes, 8192-row tiles).  Long strings: offsets pass (2048-row
    // tiles) then a copy pass with one wave per 64 rows, which spreads the byte copies over far more waves.
    for (size_t k = u0; k < u1; ++k) {
      const int ci = utf8_cols[k];
      const Column& c = rec.cols[ci];
      Column& o = out.cols[ci];
      const int64_t cap = (int64_t)hs->utf8_ends[2 * (k - u0) + 1] - hs->utf8_ends[2 * (k - u0)];
      const bool fused = cap <= mask_len * 24;
      const int64_t utile_rows = fused ? 8192 : 2048;
      const int64_t utiles = (mask_len + utile_rows - 1) / utile_rows;
      auto offb = make_device_buffer((size_t)(total + 2) * 4, ctx.device);
      auto db = make_device_buffer((size_t)cap + 16, ctx.device);
      byte_status[k] = make_device_buffer((size_t)(utiles + 64) * 8 + 16, ctx.device);
      check_hip(hipMemsetAsync(byte_status[k]->ptr, 0, (size_t)(utiles + 64) * 8 + 16, ctx.stream), "memset");
      Utf8Params up{};
      up.nrows = mask_len; up.sel_mask = (const u64*)sel_mask->ptr; up.grp_base = (const u64*)grp_base->ptr;
      up.in_offsets = (const int32_t*)c.values0(); up.in_data = c.data;
      up.out_offsets = (int32_t*)offb->ptr; up.out_data = (uint8_t*)db->ptr;
      up.byte_status = (u64*)byte_status[k]->ptr;
      up.ticket = (uint32_t*)((uint8_t*)byte_status[k]->ptr + (size_t)(utiles + 64) * 8);
      up.total_bytes = &ds->counters[kNullPerRound + (k - u0)];   // counters[16..23]: byte totals
      up.rows_out = total;
      if (fused) {
        check_hip(launch_utf8_filter(up, (int)std::min<int64_t>(utiles, (int64_t)ctx.num_cus * 2), ctx.stream), "launch utf8_filter_kernel");
        ++ctx.stats.launches;
      } else {
        check_hip(launch_utf8_offsets(up, (int)std::min<int64_t>(utiles, (int64_t)ctx.num_cus * 8), ctx.stream), "launch utf8_offsets_kernel");
        check_hip(launch_utf8_copy(up, (int)std::min<int64_t>((ngroups + 3) / 4, (int64_t)ctx.num_cus * 16), ctx.stream), "launch utf8_copy_kernel");
        ctx.stats.launches += 2;
      }
      o.values = (const uint8_t*)offb->ptr; o.owned.push_back(offb); o.length = total;
      o.data = (const uint8_t*)db->ptr; o.owned.push_back(db);
    }
    check_hip(hipMemcpyAsync(hs, ds, sizeof(Scratch), hipMemcpyDeviceToHost, ctx.stream), "read back");
    check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
    for (size_t k = n0; k < n1; ++k) {
      Column& o = out.cols[nullable_cols[k]];
      o.null_count = (int64_t)hs->counters[k - n0];
      if (o.null_count == 0) o.validity = nullptr;   // arrow drops an all-valid null buffer
    }
    for (size_t k = u0; k < u1; ++k) {   // offsets read by both passes, selected bytes read and written, new offsets
      Column& o = out.cols[utf8_cols[k]];
      o.data_bytes = (int64_t)hs->counters[kNullPerRound + (k - u0)];
      ctx.stats.bytes_read_alg += mask_len * 8 + o.data_bytes; ctx.stats.bytes_written_alg += (total + 1) * 4 + o.data_bytes;
    }
    n0 = n1; u0 = u1;
    }
  }
  for (Column& o : out.cols) o.length = total;
  return out;
}

// =================================================================================================
// filter_record_large_host: one LARGE host batch in, one host batch out.  The general path is three serial steps -- upload
// everything (10.6 ms for 50 M rows x 12 B), one 0.25 ms kernel, download everything (12.1 ms) -- on a link that is full
// duplex.  Here the batch is cut into chunks of a few million rows: chunk k is filtered by the ordinary single-batch path
// (so every semantic detail, the error order included, is the single-batch path's), its survivors start their way down to
// the host on a second stream, and the host thread moves on to uploading chunk k+1 while that download runs: uploads and
// downloads overlap, the call approaches max(upload, download) instead of their sum.  Fixed-width non-null columns and a
// non-literal predicate; anything else takes the general path.
// =================================================================================================
namespace { void add_stats(chq_call_stats& acc, const chq_call_stats& s); }
bool filter_record_large_host(Context& ctx, const Batch& rec, const chq_table_aliases* aliases, const Expr& expr, Batch* result) {
  const int64_t nrows = rec.nrows;
  const size_t ncols = rec.cols.size();
  if (!ctx.opt_large_host || rec.on_device || nrows < ctx.opt_large_host_rows || ncols == 0 || (int)ncols > MAX_OUT) return false;
  int64_t row_bytes = 0;
  for (const Column& c : rec.cols) {
    if (c.type == T_BOOL || c.type == T_UTF8 || c.width <= 0) return false;
    if (c.validity && c.null_count != 0 && (c.null_count > 0 || count_nulls_host(c.validity, c.offset, c.length) != 0)) return false;
    row_bytes += c.width;
  }
  const std::vector<PlanColumn> pcols = plan_columns(rec, aliases);
  try {
    TypedExpr te = type_expr(expr, pcols, nrows, ctx.opt_enable_minus);
    if (te.pending_code) return false;
    const Node& root = te.at(te.root);
    if (root.type != T_BOOL || root.len1) return false;
  } catch (const ChqError&) {
    return false;   // the general path reports it
  }
  // chunk: about 64 MB of input, a whole number of 16 384-row tiles
  int64_t chunk = std::max<int64_t>(1 << 20, ((int64_t)64 << 20) / std::max<int64_t>(1, row_bytes));
  if (ctx.opt_large_host_chunk > 0) chunk = ctx.opt_large_host_chunk;
  chunk = (chunk + 16383) / 16384 * 16384;
  ensure_aux_streams(ctx);   // (created once per context; also used by the Parquet scan)
  const hipStream_t down = ctx.aux[0];
  Batch out;
  out.on_device = false; out.device_id = -1;
  std::vector<BufferPtr> host_cols;
  for (const Column& c : rec.cols) {
    Column o = empty_like(c);
    auto hb = make_host_buffer((size_t)nrows * c.width + 64);
    o.values = (const uint8_t*)hb->ptr; o.owned.push_back(hb);
    host_cols.push_back(hb);
    out.cols.push_back(std::move(o));
  }
  // Host memory on both ends is pageable, so a copy call keeps its calling thread busy until the bytes have moved: the
  // downloads get a thread of their own.  It takes finished chunks off a queue (at most three wait: that bounds the HBM
  // held), copies their survivors to their place in the result and only then lets the chunk's buffers go back to the pool.
  struct Job { Batch dev_in, dev_out; int64_t base = 0; };
  std::mutex qm; std::condition_variable qcv;
  std::deque<Job> queue;
  bool closed = false;
  std::exception_ptr dl_error;
  const int device = ctx.device;
  std::thread downloader([&] {
    try {
      check_hip(hipSetDevice(device), "hipSetDevice");
      while (true) {
        Job job;
        {
          std::unique_lock<std::mutex> l(qm);
          qcv.wait(l, [&] { return closed || !queue.empty(); });
          if (queue.empty()) return;
          job = std::move(queue.front());
        }
        for (size_t i = 0; i < ncols; ++i) {
          const Column& rc = job.dev_out.cols[i];
          if (job.dev_out.nrows) check_hip(hipMemcpyAsync((uint8_t*)host_cols[i]->ptr + (size_t)job.base * rc.width, rc.values0(),
                                                         (size_t)job.dev_out.nrows * rc.width, hipMemcpyDeviceToHost, down), "download survivors");
        }
        check_hip(hipStreamSynchronize(down), "hipStreamSynchronize");
        { std::lock_guard<std::mutex> l(qm); queue.pop_front(); }   // (popped only now: the queue length bounds chunks in flight)
        qcv.notify_all();
      }
    } catch (...) {
      std::lock_guard<std::mutex> l(qm);
      dl_error = std::current_exception();
      queue.clear();
      qcv.notify_all();
    }
  });
  auto finish = [&] { { std::lock_guard<std::mutex> l(qm); closed = true; } qcv.notify_all(); if (downloader.joinable()) downloader.join(); };
  chq_call_stats acc{};
  int64_t total = 0;
  try {
    for (int64_t r0 = 0; r0 < nrows; r0 += chunk) {
      const int64_t n = std::min(chunk, nrows - r0);
      Batch view;
      view.nrows = n; view.on_device = false; view.device_id = -1;
      for (const Column& c : rec.cols) { Column v = c; v.offset = c.offset + r0; v.length = n; v.validity = nullptr; v.null_count = 0; view.cols.push_back(std::move(v)); }
      Batch dev = to_device(ctx, view);                                        // upload: this thread is busy with it
      Batch res = filter_record(ctx, dev, plan_columns(dev, aliases), expr);   // kernel + row count (synchronises ctx.stream)
      add_stats(acc, ctx.stats);
      Job job; job.base = total; total += res.nrows; job.dev_in = std::move(dev); job.dev_out = std::move(res);
      std::unique_lock<std::mutex> l(qm);
      qcv.wait(l, [&] { return dl_error || queue.size() < 3; });
      if (dl_error) break;
      queue.push_back(std::move(job));
      l.unlock();
      qcv.notify_all();
    }
  } catch (...) {
    finish();
    throw;
  }
  finish();
  if (dl_error) std::rethrow_exception(dl_error);
  out.nrows = total;
  for (Column& o : out.cols) { o.length = total; o.null_count = 0; o.validity = nullptr; }
  acc.rows_in = nrows; acc.rows_out = total;
  ctx.stats = acc;
  *result = std::move(out);
  return true;
}

// =================================================================================================
// filter_record_small_host: the reference's own calling pattern -- one 10 000-row host batch in, one host batch out --
// costs three pageable uploads, three pageable downloads (each of them synchronous) and two stream synchronisations on
// the general path: about 100 us, i.e. no faster than the CPU.  Here the columns are packed into ONE pinned block,
// uploaded with one asynchronous copy, the outputs are written into one device block at input capacity and come back
// with one asynchronous copy together with the row count: one synchronisation per call.
// =================================================================================================
bool filter_record_small_host(Context& ctx, const Batch& rec, const chq_table_aliases* aliases, const Expr& expr, Batch* result) {
  const int64_t nrows = rec.nrows;
  const size_t ncols = rec.cols.size();
  if (!ctx.opt_small_host || rec.on_device || nrows < 2 || nrows > (1 << 18) || ncols == 0 || (int)ncols > MAX_OUT) return false;
  constexpr size_t kAlign = 256;
  std::vector<size_t> at(ncols + 1, 0);
  for (size_t i = 0; i < ncols; ++i) {
    const Column& c = rec.cols[i];
    if (c.type == T_BOOL || c.type == T_UTF8 || c.width <= 0) return false;
    if (c.validity && c.null_count != 0 && (c.null_count > 0 || count_nulls_host(c.validity, c.offset, c.length) != 0)) return false;
    at[i + 1] = at[i] + ((size_t)nrows * c.width + kAlign - 1) / kAlign * kAlign;
  }
  const size_t block = at[ncols];
  if (block > ((size_t)8 << 20)) return false;
  const std::vector<PlanColumn> pcols = plan_columns(rec, aliases);
  Lowered lw;
  try {
    TypedExpr te = type_expr(expr, pcols, nrows, ctx.opt_enable_minus);
    if (te.pending_code) return false;
    const Node& root = te.at(te.root);
    if (root.type != T_BOOL || root.len1) return false;
    lower_expr(te, te.root, pcols, lw);
  } catch (const ChqError&) {
    return false;   // the general path reports it
  }
  if (!lw.strs.empty()) return false;

  if (ctx.pinned_io_bytes < 2 * block + 64) {
    if (ctx.pinned_io) (void)hipHostFree(ctx.pinned_io);
    ctx.pinned_io = nullptr; ctx.pinned_io_bytes = 0;
    const size_t cap = std::max<size_t>(2 * block + 64, (size_t)1 << 20);
    ctx.dev_io = make_device_buffer(cap, ctx.device);   // may throw: the capacity is recorded only once both halves exist
    check_hip(hipHostMalloc(&ctx.pinned_io, cap, hipHostMallocDefault), "hipHostMalloc (small host path)");
    ctx.pinned_io_bytes = cap;
  }
  uint8_t* h_in = (uint8_t*)ctx.pinned_io; uint8_t* h_out = h_in + block;
  uint8_t* d_in = (uint8_t*)ctx.dev_io->ptr; uint8_t* d_out = d_in + block;
  for (size_t i = 0; i < ncols; ++i) memcpy(h_in + at[i], rec.cols[i].values0(), (size_t)nrows * rec.cols[i].width);
  check_hip(hipMemcpyAsync(d_in, h_in, block, hipMemcpyHostToDevice, ctx.stream), "upload packed batch");

  const int tile_kind = (lw.wide || lw.num_temps > 0) ? 2 : 1;
  const int64_t tile_rows = kTileRows[tile_kind];
  const int64_t ntiles = (nrows + tile_rows - 1) / tile_rows;
  ensure_scratch(ctx, ntiles);
  Scratch* ds = dev_scratch(ctx);
  Scratch* hs = (Scratch*)ctx.pinned;

  // a view of the batch whose columns live in the device block (what the program's column refs resolve against)
  Batch dev;
  dev.nrows = nrows; dev.on_device = true; dev.device_id = ctx.device;
  for (size_t i = 0; i < ncols; ++i) {
    Column c = empty_like(rec.cols[i]);
    c.length = nrows; c.values = d_in + at[i];
    dev.cols.push_back(std::move(c));
  }
  FilterParams p{};
  p.nrows = nrows; p.tile_begin = 0; p.tile_end = ntiles;
  p.status = dev_status(ctx); p.ticket = &ds->ticket; p.total = &ds->total; p.err = &ds->err;
  fill_refs(p.pb, lw, dev, {});
  std::vector<int> launch_cols;
  for (size_t i = 0; i < ncols; ++i) launch_cols.push_back((int)i);
  pick_stash(p, ctx, lw, rec.cols, launch_cols, tile_kind);
  int n = 0;
  for (int ci : launch_cols) {
    p.outs[n].in = d_in + at[ci]; p.outs[n].out = d_out + at[ci]; p.outs[n].width = (uint32_t)rec.cols[ci].width;
    ++n;
  }
  p.n_out = (int16_t)n;
  ctx.stats = chq_call_stats{};
  ctx.stats.rows_in = nrows; ctx.stats.tiles = ntiles; ctx.stats.launches = 1;
  check_hip(hipMemsetAsync(ds, 0, kHeader + (size_t)(ntiles + 1) * 8, ctx.stream), "memset scratch + status");
  const int grid_cap = ctx.num_cus * kGridPerCu[tile_kind];
  check_hip(launch_filter(p, tile_kind, true, (int)std::min<int64_t>(ntiles, grid_cap), ctx.stream), "launch filter_fused_kernel (small host batch)");
  check_hip(hipMemcpyAsync(h_out, d_out, block, hipMemcpyDeviceToHost, ctx.stream), "download packed result");
  check_hip(hipMemcpyAsync(hs, ds, sizeof(Scratch), hipMemcpyDeviceToHost, ctx.stream), "read back");
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  if (hs->err != ERR_NONE) return false;   // the general path reports the error
  const int64_t total = (int64_t)hs->total;
  Batch out;
  out.on_device = false; out.device_id = -1; out.nrows = total;
  for (size_t i = 0; i < ncols; ++i) {
    Column o = empty_like(rec.cols[i]);
    auto hb = make_host_buffer((size_t)total * o.width + 16);
    if (total) memcpy(hb->ptr, h_out + at[i], (size_t)total * o.width);
    o.values = (const uint8_t*)hb->ptr; o.length = total; o.owned.push_back(hb);
    ctx.stats.bytes_read_alg += nrows * o.width; ctx.stats.bytes_written_alg += total * o.width;
    out.cols.push_back(std::move(o));
  }
  ctx.stats.rows_out = total;
  *result = std::move(out);
  return true;
}

// =================================================================================================
// filter_records: filter_record over a group of batches that share one schema, in ONE launch.
//
// The reference hands the filter operator one 10 000-row batch at a time (physical_planner.rs:323); at that size a
// call is bounded by launch + read-back latency, not by HBM.  A group call keeps the per-batch semantics (one output
// batch per input batch, same rows, same order) but runs one chained-scan compaction over all tiles of all batches:
// tiles never straddle batches, a per-tile table carries the batch-local row range and the column pointers, the output
// of every column is one dense buffer and batch b's output is the slice between the inclusive prefixes of the last
// tiles of batches b-1 and b.
//
// Fast path: every column fixed-width (not Boolean / Utf8) and free of nulls, every batch >= 2 rows, predicate not
// literal-only, no static error.  Anything else -- and any data-dependent error -- takes the per-batch loop, which
// reports exactly what chq_filter_record would for the earliest failing batch.
// =================================================================================================
namespace {
void add_stats(chq_call_stats& acc, const chq_call_stats& s) {
  acc.rows_in += s.rows_in; acc.rows_out += s.rows_out; acc.tiles += s.tiles; acc.launches += s.launches;
  acc.bytes_read_alg += s.bytes_read_alg; acc.bytes_written_alg += s.bytes_written_alg; acc.kernel_ns += s.kernel_ns;
}
// f(begin, end) over [0, n) on a few host threads when there is enough to copy (one thread moves ~10 GB/s, the PCIe
// link 55 GB/s: packing a group single-threaded would be the slowest step of a host-resident call)
template <class F>
void parallel_ranges(size_t n, size_t bytes, F&& f) {
  if (bytes < ((size_t)8 << 20) || n < 2) { f((size_t)0, n); return; }
  const size_t T = std::min<size_t>(std::min<size_t>(8, pool_width()), n);
  pool_ranges(n, (n + T - 1) / T, [&](size_t i0, size_t i1) { f(i0, i1); });
}

// ---- host-side concatenation of a group (general column kinds) -----------------------------------------------
// append n bits of src starting at bit src_bit (src == nullptr: ones) to dst at bit dst_bit; dst is zero-filled and has
// 8 spare bytes behind its last bit
void append_bits(uint8_t* dst, int64_t dst_bit, const uint8_t* src, int64_t src_bit, int64_t n) {
  while (n > 0) {
    const int k = (int)std::min<int64_t>(n, 56);
    uint64_t v;
    if (src) {
      const int sh = (int)(src_bit & 7);
      const int nbytes = (sh + k + 7) / 8;
      uint64_t raw = 0;
      memcpy(&raw, src + (src_bit >> 3), (size_t)nbytes);
      v = (raw >> sh) & ((1ULL << k) - 1ULL);
    } else {
      v = (1ULL << k) - 1ULL;
    }
    const int dsh = (int)(dst_bit & 7);
    uint64_t cur;
    uint8_t* dp = dst + (dst_bit >> 3);
    memcpy(&cur, dp, 8);
    cur |= v << dsh;                       // k + dsh <= 63
    memcpy(dp, &cur, 8);
    dst_bit += k; src_bit += k; n -= k;
  }
}

// one host batch holding the rows of recs[b0, b1) back to back (buffers from the recycling host pool)
Batch concat_host_batches(const std::vector<Batch>& recs, size_t b0, size_t b1) {
  Batch cat;
  cat.on_device = false; cat.device_id = -1;
  int64_t total = 0;
  for (size_t b = b0; b < b1; ++b) total += recs[b].nrows;
  cat.nrows = total;
  const size_t ncols = recs[b0].cols.size();
  for (size_t i = 0; i < ncols; ++i) {
    const Column& c0 = recs[b0].cols[i];
    Column o = empty_like(c0);
    o.length = total;
    bool any_nulls = false;
    for (size_t b = b0; b < b1; ++b) { const Column& c = recs[b].cols[i]; any_nulls |= c.validity && c.null_count != 0; }
    if (any_nulls) {
      auto vb = make_host_buffer((size_t)(total + 7) / 8 + 16);
      memset(vb->ptr, 0, (size_t)(total + 7) / 8 + 16);
      int64_t at = 0, nulls = 0;
      for (size_t b = b0; b < b1; ++b) {
        const Column& c = recs[b].cols[i];
        const bool has = c.validity && c.null_count != 0;
        append_bits((uint8_t*)vb->ptr, at, has ? c.validity : nullptr, c.offset, c.length);
        if (has) nulls += c.null_count > 0 ? c.null_count : count_nulls_host(c.validity, c.offset, c.length);
        at += c.length;
      }
      o.validity = (const uint8_t*)vb->ptr; o.null_count = nulls; o.owned.push_back(vb);
    }
    if (c0.type == T_BOOL) {
      auto vb = make_host_buffer((size_t)(total + 7) / 8 + 16);
      memset(vb->ptr, 0, (size_t)(total + 7) / 8 + 16);
      int64_t at = 0;
      for (size_t b = b0; b < b1; ++b) { const Column& c = recs[b].cols[i]; append_bits((uint8_t*)vb->ptr, at, c.values, c.offset, c.length); at += c.length; }
      o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
    } else if (c0.type == T_UTF8) {
      const size_t n = b1 - b0;
      std::vector<int64_t> row_at(n + 1, 0), byte_at(n + 1, 0);
      for (size_t k = 0; k < n; ++k) {
        const Column& c = recs[b0 + k].cols[i];
        int64_t nbytes = 0;
        if (c.values && c.length) { const int32_t* offs = (const int32_t*)c.values + c.offset; nbytes = (int64_t)offs[c.length] - offs[0]; }
        row_at[k + 1] = row_at[k] + c.length; byte_at[k + 1] = byte_at[k] + nbytes;
      }
      // Arrow Utf8 offsets are int32: a joined column of 2 GiB or more cannot be represented (arrow's concat reports
      // an offset overflow; wrapping silently would hand out negative offsets)
      if (byte_at[n] > (int64_t)INT32_MAX)
        throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "offset overflow: Utf8 column '" + c0.name + "' of the joined batches holds " +
                                                           std::to_string(byte_at[n]) + " bytes, more than int32 offsets can address"};
      auto ob = make_host_buffer((size_t)(total + 1) * 4 + 16);
      auto db = make_host_buffer((size_t)byte_at[n] + 16);
      int32_t* oo = (int32_t*)ob->ptr;
      oo[0] = 0;
      parallel_ranges(n, (size_t)byte_at[n] + (size_t)total * 4, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
          const Column& c = recs[b0 + k].cols[i];
          if (!c.values || !c.length) continue;
          const int32_t* offs = (const int32_t*)c.values + c.offset;
          const int32_t first = offs[0];
          const int64_t nbytes = byte_at[k + 1] - byte_at[k];
          if (nbytes) memcpy((uint8_t*)db->ptr + byte_at[k], c.data + first, (size_t)nbytes);
          const int32_t shift = (int32_t)(byte_at[k] - first);
          int32_t* dst = oo + row_at[k];
          for (int64_t r = 1; r <= c.length; ++r) dst[r] = offs[r] + shift;
        }
      });
      o.values = (const uint8_t*)ob->ptr; o.owned.push_back(ob);
      o.data = (const uint8_t*)db->ptr; o.owned.push_back(db);
    } else {
      const size_t n = b1 - b0;
      std::vector<int64_t> row_at(n + 1, 0);
      for (size_t k = 0; k < n; ++k) row_at[k + 1] = row_at[k] + recs[b0 + k].cols[i].length;
      auto vb = make_host_buffer((size_t)total * c0.width + 16);
      parallel_ranges(n, (size_t)total * c0.width, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
          const Column& c = recs[b0 + k].cols[i];
          if (c.length) memcpy((uint8_t*)vb->ptr + row_at[k] * c.width, c.values0(), (size_t)c.length * c.width);
        }
      });
      o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
    }
    cat.cols.push_back(std::move(o));
  }
  return cat;
}

void ensure_pinned_table(Context& ctx, size_t bytes) {
  if (ctx.pinned_tbl_bytes >= bytes) return;
  if (ctx.pinned_tbl) (void)hipHostFree(ctx.pinned_tbl);
  ctx.pinned_tbl = nullptr; ctx.pinned_tbl_bytes = 0;
  const size_t cap = bytes + bytes / 4 + 4096;
  check_hip(hipHostMalloc(&ctx.pinned_tbl, cap, hipHostMallocDefault), "hipHostMalloc (group table)");
  ctx.pinned_tbl_bytes = cap;
}

// ---- device-side concatenation of a group (general column kinds) ----------------------------------------------------
// The batches of recs[b0, b1) -- all resident in this GPU's HBM, one schema -- joined into ONE batch by the concat_*
// kernels: fixed-width values copied back to back, Utf8 offsets rebased onto one data buffer, Boolean values and validity
// bitmaps appended bit by bit.  `utf8_bytes[k][b]` = data bytes of the k-th Utf8 column of batch b (from gather_ends).
Batch concat_device_batches(Context& ctx, const std::vector<Batch>& recs, size_t b0, size_t b1,
                            const std::vector<int>& utf8_cols, const std::vector<std::vector<int64_t>>& utf8_bytes) {
  const size_t nb = b1 - b0, nc = recs[b0].cols.size();
  PhaseTimer pt("concat_device_batches");
  Batch cat;
  cat.on_device = true; cat.device_id = ctx.device;
  // ---- tables: [row_at (nb+1)] then per column [src nb] [aux nb | -] [bitoff nb | -] [byte_at nb+1 | -] [vsrc nb, vbitoff nb | -]
  struct ColPlan { size_t src = 0, aux = 0, bitoff = 0, byte_at = 0, vsrc = 0, vbitoff = 0; bool validity = false; int utf8_k = -1; };
  std::vector<ColPlan> plan(nc);
  // one pass over the batches on the pool's threads (every visit of a Batch is a cache miss at 10^4 batches): row counts
  // into a dense array, "does any batch carry nulls" per column
  std::vector<int64_t> batch_rows(nb, 0);
  {
    std::vector<std::atomic<int>> any_nulls(nc);
    for (auto& a : any_nulls) a.store(0);
    pool_ranges(nb, 2048, [&](size_t k0, size_t k1) {
      for (size_t k = k0; k < k1; ++k) {
        const Batch& rb = recs[b0 + k];
        batch_rows[k] = rb.nrows;
        for (size_t c = 0; c < nc; ++c) if (rb.cols[c].validity && rb.cols[c].null_count != 0) any_nulls[c].store(1, std::memory_order_relaxed);
      }
    });
    for (size_t c = 0; c < nc; ++c) plan[c].validity = any_nulls[c].load() != 0;
  }
  size_t words = nb + 1;
  for (size_t c = 0; c < nc; ++c) {
    const Column& c0 = recs[b0].cols[c];
    plan[c].src = words; words += nb;
    if (c0.type == T_UTF8) {
      plan[c].aux = words; words += nb;
      plan[c].byte_at = words; words += nb + 1;
      plan[c].utf8_k = (int)(std::find(utf8_cols.begin(), utf8_cols.end(), (int)c) - utf8_cols.begin());
    }
    if (c0.type == T_BOOL) { plan[c].bitoff = words; words += nb; }
    if (plan[c].validity) { plan[c].vsrc = words; words += nb; plan[c].vbitoff = words; words += nb; }
  }
  pt.mark("scan_batches");
  ensure_pinned_table(ctx, words * 8);
  u64* h = (u64*)ctx.pinned_tbl;
  int64_t total = 0;
  for (size_t k = 0; k < nb; ++k) { h[k] = (u64)total; total += batch_rows[k]; }
  h[nb] = (u64)total;
  cat.nrows = total;
  std::vector<int64_t> total_bytes(nc, 0);
  std::vector<int64_t> known_nulls(nc, 0);
  for (size_t c = 0; c < nc; ++c) {   // running byte positions first (serial, over a dense array)
    const ColPlan& pl = plan[c];
    if (recs[b0].cols[c].type != T_UTF8) continue;
    int64_t bytes = 0;
    const std::vector<int64_t>& ub = utf8_bytes[(size_t)pl.utf8_k];
    for (size_t k = 0; k < nb; ++k) { h[pl.byte_at + k] = (u64)bytes; bytes += ub[b0 + k]; }
    h[pl.byte_at + nb] = (u64)bytes; total_bytes[c] = bytes;
  }
  // the pointers: one pass over the batches (each batch's columns lie together in memory), on the pool's threads --
  // at 10^4 batches this is pointer chasing through ~10 MB of Batch / Column objects
  std::mutex nulls_m;
  pool_ranges(nb, 2048, [&](size_t k0, size_t k1) {
    std::vector<int64_t> nulls(nc, 0);
    for (size_t k = k0; k < k1; ++k) {
      const Batch& rb = recs[b0 + k];
      for (size_t c = 0; c < nc; ++c) {
        const ColPlan& pl = plan[c];
        const Column& col = rb.cols[c];
        h[pl.src + k] = (u64)(uintptr_t)(col.type == T_BOOL ? (const void*)col.values : col.values0());
        if (col.type == T_UTF8) h[pl.aux + k] = (u64)(uintptr_t)col.data;
        if (col.type == T_BOOL) h[pl.bitoff + k] = (u64)col.offset;
        if (pl.validity) {
          const bool has = col.validity && col.null_count != 0;
          h[pl.vsrc + k] = has ? (u64)(uintptr_t)col.validity : 0;
          h[pl.vbitoff + k] = (u64)col.offset;
          if (has && col.null_count > 0) nulls[c] += col.null_count;
        }
      }
    }
    std::lock_guard<std::mutex> l(nulls_m);
    for (size_t c = 0; c < nc; ++c) known_nulls[c] += nulls[c];
  });
  pt.mark("tables");
  auto d_tbl = make_device_buffer(words * 8 + 16, ctx.device);
  check_hip(hipMemcpyAsync(d_tbl->ptr, h, words * 8, hipMemcpyHostToDevice, ctx.stream), "upload concat tables");
  pt.mark("upload");
  const u64* d = (const u64*)d_tbl->ptr;
  const int grid = (int)std::min<int64_t>((int64_t)nb, (int64_t)ctx.num_cus * 16);
  for (size_t c = 0; c < nc; ++c) {
    const Column& c0 = recs[b0].cols[c];
    const ColPlan& pl = plan[c];
    Column o = empty_like(c0);
    o.length = total;
    ConcatParams cp{};
    cp.nb = (int64_t)nb; cp.row_at = (const int64_t*)d; cp.src = d + pl.src;
    if (pl.validity) {
      const size_t vbytes = (size_t)(total + 31) / 32 * 4 + 16;
      auto vb = make_device_buffer(vbytes, ctx.device);
      check_hip(hipMemsetAsync(vb->ptr, 0, vbytes, ctx.stream), "memset validity");
      ConcatParams vp = cp;
      vp.src = d + pl.vsrc; vp.bitoff = (const int64_t*)(d + pl.vbitoff); vp.dst = vb->ptr;
      check_hip(launch_concat(vp, 3, grid, ctx.stream), "launch concat_bits_kernel (validity)");
      o.validity = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
      o.null_count = known_nulls[c] > 0 ? known_nulls[c] : 1;   // "may contain nulls": the filter counts what survives
    }
    if (c0.type == T_BOOL) {
      const size_t bbytes = (size_t)(total + 31) / 32 * 4 + 16;
      auto vb = make_device_buffer(bbytes, ctx.device);
      check_hip(hipMemsetAsync(vb->ptr, 0, bbytes, ctx.stream), "memset bits");
      cp.bitoff = (const int64_t*)(d + pl.bitoff); cp.dst = vb->ptr;
      check_hip(launch_concat(cp, 3, grid, ctx.stream), "launch concat_bits_kernel");
      o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
    } else if (c0.type == T_UTF8) {
      auto ob = make_device_buffer((size_t)(total + 1) * 4 + 16, ctx.device);
      auto db = make_device_buffer((size_t)total_bytes[c] + 16, ctx.device);
      cp.aux = d + pl.aux; cp.byte_at = (const int64_t*)(d + pl.byte_at); cp.dst = ob->ptr; cp.dst2 = db->ptr;
      if (total == 0) check_hip(hipMemsetAsync(ob->ptr, 0, 4, ctx.stream), "memset offsets");
      check_hip(launch_concat(cp, 2, grid, ctx.stream), "launch concat_utf8_kernel");
      o.values = (const uint8_t*)ob->ptr; o.owned.push_back(ob);
      o.data = (const uint8_t*)db->ptr; o.owned.push_back(db);
      o.data_bytes = total_bytes[c];
    } else {
      auto vb = make_device_buffer((size_t)total * c0.width + 16, ctx.device);
      cp.dst = vb->ptr; cp.width = c0.width;
      check_hip(launch_concat(cp, 0, grid, ctx.stream), "launch concat_fixed_kernel");
      o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
    }
    cat.cols.push_back(std::move(o));
  }
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");   // the pinned table is reused by the next chunk
  pt.mark("join_kernels");
  return cat;
}

// data bytes of every Utf8 column of every batch of a device-resident group (first / last offset read by one kernel per
// column).  Two steps: `issue` queues the kernels and the read-back, `finish` waits for them -- the one-launch path builds
// its group table in between.
struct Utf8Sizes {
  std::vector<int> cols;
  size_t nb = 0;
  BufferPtr d_tbl;
  int32_t* h_ends = nullptr;   // [cols][2 nb] in ctx.pinned_sizes
};
Utf8Sizes device_utf8_bytes_issue(Context& ctx, const std::vector<Batch>& recs, const GroupLite* lite, const std::vector<int>& utf8_cols) {
  Utf8Sizes z;
  z.cols = utf8_cols; z.nb = lite ? lite->rows.size() : recs.size();
  const size_t nb = z.nb, nu = utf8_cols.size();
  if (nu == 0) return z;
  const size_t words = (nb + 1) + nu * nb;                 // row_at, then one pointer table per column
  const size_t need = words * 8 + nu * nb * 8;
  if (ctx.pinned_sizes_bytes < need) {
    if (ctx.pinned_sizes) (void)hipHostFree(ctx.pinned_sizes);
    ctx.pinned_sizes = nullptr; ctx.pinned_sizes_bytes = 0;
    check_hip(hipHostMalloc(&ctx.pinned_sizes, need + need / 4 + 4096, hipHostMallocDefault), "hipHostMalloc (utf8 sizes)");
    ctx.pinned_sizes_bytes = need + need / 4 + 4096;
  }
  u64* h = (u64*)ctx.pinned_sizes;
  int64_t total = 0;
  for (size_t b = 0; b < nb; ++b) { h[b] = (u64)total; total += lite ? lite->rows[b] : recs[b].nrows; }
  h[nb] = (u64)total;
  z.d_tbl = make_device_buffer(need + 16, ctx.device);
  z.h_ends = (int32_t*)(h + words);
  for (size_t k = 0; k < nu; ++k) {
    const size_t uc = (size_t)utf8_cols[k];
    u64* tbl = h + nb + 1 + k * nb;
    if (lite) for (size_t b = 0; b < nb; ++b) tbl[b] = (u64)(uintptr_t)lite->values0[b * lite->ncols + uc];
    else pool_ranges(nb, 2048, [&](size_t i0, size_t i1) { for (size_t b = i0; b < i1; ++b) tbl[b] = (u64)(uintptr_t)recs[b].cols[uc].values0(); });
  }
  check_hip(hipMemcpyAsync(z.d_tbl->ptr, h, words * 8, hipMemcpyHostToDevice, ctx.stream), "upload offsets tables");
  int32_t* d_ends = (int32_t*)((u64*)z.d_tbl->ptr + words);
  for (size_t k = 0; k < nu; ++k) {
    ConcatParams cp{};
    cp.nb = (int64_t)nb; cp.row_at = (const int64_t*)z.d_tbl->ptr; cp.src = (const u64*)z.d_tbl->ptr + nb + 1 + k * nb; cp.ends = d_ends + 2 * k * nb;
    check_hip(launch_concat(cp, 1, 1, ctx.stream), "launch gather_ends_kernel");
  }
  check_hip(hipMemcpyAsync(z.h_ends, d_ends, nu * nb * 8, hipMemcpyDeviceToHost, ctx.stream), "read back ends");
  return z;
}
std::vector<std::vector<int64_t>> device_utf8_bytes_finish(Context& ctx, const Utf8Sizes& z) {
  std::vector<std::vector<int64_t>> out(z.cols.size(), std::vector<int64_t>(z.nb, 0));
  if (z.cols.empty()) return out;
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  for (size_t k = 0; k < z.cols.size(); ++k) {
    const int32_t* e = z.h_ends + 2 * k * z.nb;
    for (size_t b = 0; b < z.nb; ++b) out[k][b] = (int64_t)e[2 * b + 1] - e[2 * b];
  }
  return out;
}
std::vector<std::vector<int64_t>> device_utf8_bytes(Context& ctx, const std::vector<Batch>& recs, const std::vector<int>& utf8_cols) {
  return device_utf8_bytes_finish(ctx, device_utf8_bytes_issue(ctx, recs, nullptr, utf8_cols));
}
}  // namespace

namespace {
// `co` != nullptr asks for ONE output batch (all surviving rows, input order) + rows per input batch; the one-launch
// path fills it directly and sets co->done, every other path returns per-batch outputs for the caller to join
struct Coalesced { Batch out; std::vector<int64_t> rows; bool done = false; };
std::vector<Batch> filter_records_impl(Context& ctx, const GroupInput& gi, const chq_table_aliases* aliases,
                                       const Expr& expr, bool out_on_device, Coalesced* co, GroupSliced* sliced);
}  // namespace

std::vector<Batch> filter_records(Context& ctx, const GroupInput& in, const chq_table_aliases* aliases,
                                  const Expr& expr, bool out_on_device, GroupSliced* sliced) {
  return filter_records_impl(ctx, in, aliases, expr, out_on_device, nullptr, sliced);
}

Batch filter_records_coalesced(Context& ctx, const GroupInput& in, const chq_table_aliases* aliases,
                               const Expr& expr, bool out_on_device, std::vector<int64_t>* rows_per_record) {
  if (!in.batches || in.batches->empty()) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "no record batches to coalesce"};
  Coalesced co;
  std::vector<Batch> outs = filter_records_impl(ctx, in, aliases, expr, out_on_device, &co, nullptr);
  if (!co.done) {   // join the per-batch results on the host (general column kinds), then move them where they are wanted
    const chq_call_stats st = ctx.stats;
    std::vector<Batch> host;
    for (Batch& o : outs) { co.rows.push_back(o.nrows); host.push_back(o.on_device ? to_host(ctx, o) : std::move(o)); }
    Batch cat = concat_host_batches(host, 0, host.size());
    co.out = out_on_device ? to_device(ctx, cat) : std::move(cat);
    if (out_on_device) check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
    ctx.stats = st;
  }
  if (rows_per_record) *rows_per_record = co.rows;
  return std::move(co.out);
}

namespace {
std::vector<Batch> filter_records_impl(Context& ctx, const GroupInput& gi, const chq_table_aliases* aliases,
                                       const Expr& expr, bool out_on_device, Coalesced* co, GroupSliced* sliced) {
  const std::vector<Batch>& recs = *gi.batches;   // (batch 0 is always there; the others after need_batches() when `lite` is set)
  const GroupLite* lite = gi.lite;
  if (lite && (lite->rows.empty() || recs.empty() || lite->ncols != recs[0].cols.size())) lite = nullptr;
  auto need_batches = [&]() { if (gi.materialise) gi.materialise(); };
  if (!lite) need_batches();
  const size_t nb = lite ? lite->rows.size() : recs.size();
  auto per_batch_loop = [&]() {
    need_batches();
    std::vector<Batch> outs;
    chq_call_stats acc{};
    for (const Batch& r : recs) {
      Batch dev = to_device(ctx, r);
      Batch o = filter_record(ctx, dev, plan_columns(dev, aliases), expr);
      add_stats(acc, ctx.stats);
      outs.push_back(out_on_device ? std::move(o) : to_host(ctx, o));
    }
    ctx.stats = acc;
    return outs;
  };
  if (nb < 2) return per_batch_loop();

  // ---- host batches with Utf8 / Boolean / nullable columns: concatenate on the host while staging, filter the one
  // big batch (every column kind is supported there), copy the result back once and slice it per input batch at the
  // positions the device reports (split_bounds_kernel).  Chunks keep every Utf8 column below 1 GiB of bytes.
  auto host_concat_path = [&]() -> std::vector<Batch> {
    need_batches();
    const size_t nc = recs[0].cols.size();
    std::vector<size_t> cuts{0};
    {
      std::vector<int64_t> bytes(nc, 0);
      int64_t rows = 0;
      for (size_t b = 0; b < nb; ++b) {
        bool over = rows + recs[b].nrows > (1ll << 30);
        const int64_t limit = ctx.opt_group_chunk_bytes;
        std::vector<int64_t> add(nc, 0);
        for (size_t i = 0; i < nc; ++i) {
          const Column& c = recs[b].cols[i];
          if (c.type == T_UTF8 && c.values && c.length) { const int32_t* o = (const int32_t*)c.values + c.offset; add[i] = (int64_t)o[c.length] - o[0]; }
          over |= bytes[i] + add[i] > limit;
        }
        if (over && b > cuts.back()) { cuts.push_back(b); std::fill(bytes.begin(), bytes.end(), 0); rows = 0; }
        for (size_t i = 0; i < nc; ++i) bytes[i] += add[i];
        rows += recs[b].nrows;
      }
      cuts.push_back(nb);
    }
    std::vector<Batch> outs;
    outs.reserve(nb);
    chq_call_stats acc{};
    for (size_t k = 0; k + 1 < cuts.size(); ++k) {
      const size_t b0 = cuts[k], b1 = cuts[k + 1];
      Batch cat = concat_host_batches(recs, b0, b1);
      SplitRequest split;
      int64_t at = 0;
      for (size_t b = b0; b < b1; ++b) { split.starts.push_back(at); at += recs[b].nrows; }
      split.starts.push_back(at);
      Batch dev = to_device(ctx, cat);
      Batch res = to_host(ctx, filter_record(ctx, dev, plan_columns(dev, aliases), expr, &split));
      add_stats(acc, ctx.stats);
      for (size_t b = b0; b < b1; ++b) {
        const int64_t begin = split.bounds[b - b0], end = split.bounds[b - b0 + 1];
        Batch o;
        o.on_device = false; o.device_id = -1; o.nrows = end - begin;
        for (const Column& c : res.cols) {
          Column sc = c;   // shares the result buffers
          sc.offset = c.offset + begin; sc.length = end - begin;
          if (sc.validity) {
            sc.null_count = count_nulls_host(sc.validity, sc.offset, sc.length);
            if (sc.null_count == 0) sc.validity = nullptr;   // arrow drops an all-valid null buffer
          } else sc.null_count = 0;
          o.cols.push_back(std::move(sc));
        }
        outs.push_back(std::move(o));
      }
    }
    ctx.stats = acc;
    return outs;
  };

  // ---- device-resident batches with Utf8 / Boolean / nullable columns (the reference's own schema is Int32, Utf8,
  // Float32: create_sample_data.rs:157-204): joined on the device by the concat kernels, filtered as ONE batch by the
  // ordinary kernels, cut at the batch boundaries the device reports (split_bounds_kernel).  Every output batch is a
  // slice (Arrow offset) of the shared result buffers; chunks keep every Utf8 column below the int32 offset range.
  auto device_concat_path = [&]() -> std::vector<Batch> {
    need_batches();
    const size_t nc = recs[0].cols.size();
    std::vector<int> utf8_cols;
    for (size_t i = 0; i < nc; ++i) if (recs[0].cols[i].type == T_UTF8) utf8_cols.push_back((int)i);
    PhaseTimer pt("device_concat_path");
    const std::vector<std::vector<int64_t>> ubytes = device_utf8_bytes(ctx, recs, utf8_cols);
    pt.mark("utf8_sizes");
    std::vector<size_t> cuts{0};
    {
      std::vector<int64_t> bytes(utf8_cols.size(), 0);
      int64_t rows = 0;
      for (size_t b = 0; b < nb; ++b) {
        bool over = rows + recs[b].nrows > (1ll << 30);
        for (size_t k = 0; k < utf8_cols.size(); ++k) over |= bytes[k] + ubytes[k][b] > ctx.opt_group_chunk_bytes;
        if (over && b > cuts.back()) { cuts.push_back(b); std::fill(bytes.begin(), bytes.end(), 0); rows = 0; }
        for (size_t k = 0; k < utf8_cols.size(); ++k) bytes[k] += ubytes[k][b];
        rows += recs[b].nrows;
      }
      cuts.push_back(nb);
    }
    if (co && cuts.size() > 2) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "offset overflow: the joined Utf8 output of this group does not fit int32 offsets; use chq_filter_records"};
    std::vector<Batch> outs;
    if (!co) outs.reserve(nb);
    chq_call_stats acc{};
    for (size_t k = 0; k + 1 < cuts.size(); ++k) {
      const size_t b0 = cuts[k], b1 = cuts[k + 1];
      // (a chunk of one batch -- e.g. a 2 GB Utf8 column on its own -- is filtered in place: nothing to join)
      Batch cat = b1 - b0 == 1 ? to_device(ctx, recs[b0]) : concat_device_batches(ctx, recs, b0, b1, utf8_cols, ubytes);
      pt.mark("join_launch");
      SplitRequest split;
      int64_t at = 0;
      for (size_t b = b0; b < b1; ++b) { split.starts.push_back(at); at += recs[b].nrows; }
      split.starts.push_back(at);
      Batch res = filter_record(ctx, cat, plan_columns(cat, aliases), expr, &split);
      pt.mark("filter_record");
      add_stats(acc, ctx.stats);
      if (co) {
        for (size_t b = b0; b < b1; ++b) co->rows.push_back(split.bounds[b - b0 + 1] - split.bounds[b - b0]);
        co->out = out_on_device ? std::move(res) : to_host(ctx, res);
        co->done = true;
        continue;
      }
      Batch whole = out_on_device ? std::move(res) : to_host(ctx, res);
      for (size_t b = b0; b < b1; ++b) {
        const int64_t begin = split.bounds[b - b0], end = split.bounds[b - b0 + 1];
        Batch o;
        o.on_device = out_on_device; o.device_id = out_on_device ? ctx.device : -1; o.nrows = end - begin;
        o.cols.reserve(whole.cols.size());
        for (const Column& c : whole.cols) {
          Column sc = c;   // shares the result buffers
          sc.offset = c.offset + begin; sc.length = end - begin;
          if (sc.validity) {
            if (out_on_device) sc.null_count = -1;   // unknown for the slice (Arrow C Data Interface: -1)
            else { sc.null_count = count_nulls_host(sc.validity, sc.offset, sc.length); if (sc.null_count == 0) sc.validity = nullptr; }
          } else sc.null_count = 0;
          o.cols.push_back(std::move(sc));
        }
        outs.push_back(std::move(o));
      }
    }
    ctx.stats = acc;
    return outs;
  };

  // ---- eligibility -----------------------------------------------------------------------------------
  PhaseTimer pt("filter_records (one-launch path)");
  const size_t ncols = recs[0].cols.size();
  if (ncols == 0) return per_batch_loop();
  bool plain = (int)ncols <= MAX_OUT, same_schema = true, all_host = !recs[0].on_device;
  // `foldable`: device-resident, non-null, fixed-width or Utf8 columns -- short-string Utf8 columns can then be filtered
  // straight out of the batches by the one-launch path (their offsets and bytes per batch ride in the group table)
  bool foldable = (int)ncols <= MAX_OUT && ctx.opt_fold_utf8 && ctx.opt_group_fold;
  bool has_bool = false, has_utf8 = false, need_bits = false;
  for (size_t i = 0; i < ncols; ++i) { has_bool |= recs[0].cols[i].type == T_BOOL; has_utf8 |= recs[0].cols[i].type == T_UTF8; }
  if (lite && recs[0].on_device) {
    // device-resident group: the facts were gathered per batch at import (GroupLite) -- one pass over nb bytes
    uint8_t any = 0, all = 0xff;
    for (uint8_t f : lite->flags) { any |= f; all &= f; }
    if (any & (GroupLite::GL_SHORT | GroupLite::GL_SCHEMA_DIFFERS)) { if (any & GroupLite::GL_SHORT) return per_batch_loop(); same_schema = false; }
    all_host = false;
    // validity bitmaps and Boolean columns ride along in the one-launch path when the group is wave-packed (decided below):
    // their bitmaps are compacted by bit_compact_group_kernel behind the main kernel
    need_bits = has_bool || (any & GroupLite::GL_NULLS);
    const bool all_dev = all & GroupLite::GL_ON_DEVICE;
    plain = plain && !has_utf8 && (!need_bits || (all_dev && ctx.opt_group_bits));
    foldable = foldable && all_dev && !(any & GroupLite::GL_NO_UTF8_DATA) && (!need_bits || ctx.opt_group_bits);
  } else {
  need_batches();
  for (const Batch& r : recs) {
    if (r.cols.size() != ncols || r.nrows < 2) return per_batch_loop();
    all_host &= !r.on_device;
    foldable &= r.on_device && r.device_id == ctx.device;
    for (size_t i = 0; i < ncols; ++i) {
      const Column& c = r.cols[i];
      same_schema &= c.type == recs[0].cols[i].type && c.width == recs[0].cols[i].width && c.format == recs[0].cols[i].format;
      plain &= c.type != T_BOOL && c.type != T_UTF8 && !(c.validity && c.null_count != 0 && (r.on_device || c.null_count > 0 ||
               count_nulls_host(c.validity, c.offset, c.length) != 0));
      foldable &= c.type != T_BOOL && !(c.validity && c.null_count != 0) && (c.type != T_UTF8 || c.data != nullptr);
    }
  }
  }
  if (!same_schema) return per_batch_loop();
  std::vector<int> fold_utf8;                          // the Utf8 columns of a foldable group
  std::vector<std::vector<int64_t>> fold_bytes;        // their data bytes per batch
  std::vector<int64_t> fold_cap;
  Utf8Sizes sizes_in_flight;
  int64_t fold_rows_all = 0;
  if (!plain && foldable) {
    for (size_t i = 0; i < ncols; ++i) if (recs[0].cols[i].type == T_UTF8) fold_utf8.push_back((int)i);
    if (lite) for (int64_t r : lite->rows) fold_rows_all += r; else for (const Batch& r : recs) fold_rows_all += r.nrows;
    foldable = !fold_utf8.empty() && (int)fold_utf8.size() <= MAX_FOLD_UTF8 && fold_rows_all < (1ll << 31) - 64;
    // the batches' string sizes are read back from the device: queued here, awaited only after the predicate has been
    // typed and the group table built (`finish_fold_sizes` below)
    if (foldable) sizes_in_flight = device_utf8_bytes_issue(ctx, recs, lite, fold_utf8);
  }
  bool fold = !plain && foldable;
  bool sizes_done = false, sizes_ok = false;
  auto finish_fold_sizes = [&]() -> bool {   // false: long strings, or more than one output column can address
    if (sizes_done) return sizes_ok;
    sizes_done = true;
    fold_bytes = device_utf8_bytes_finish(ctx, sizes_in_flight);
    bool ok = true;
    for (const auto& per_batch : fold_bytes) {
      int64_t cap = 0;
      for (int64_t v : per_batch) cap += v;
      fold_cap.push_back(cap);
      // short strings that fit ONE output column (int32 offsets: 2 x group_chunk_bytes = 2 GiB unless a test lowers the option)
      ok &= cap <= fold_rows_all * 24 && cap < 2 * ctx.opt_group_chunk_bytes - 64;
    }
    sizes_ok = ok;
    return ok;
  };
  pt.mark("eligibility+utf8_sizes");
  if (!plain && !fold) {
    bool all_device = true;
    for (const Batch& r : recs) all_device &= r.on_device && r.device_id == ctx.device;
    const bool host_case = all_host && !out_on_device;
    if (!host_case && !all_device) return per_batch_loop();
    try {
      TypedExpr probe = type_expr(expr, plan_columns(recs[0], aliases), recs[0].nrows, ctx.opt_enable_minus);
      if (probe.pending_code || probe.at(probe.root).type != T_BOOL || probe.at(probe.root).len1) return per_batch_loop();
    } catch (const ChqError&) {
      return per_batch_loop();   // reports the first batch's (static) error
    }
    try {
      return host_case ? host_concat_path() : device_concat_path();
    } catch (const ChqError& e) {
      if (e.code == CHQ_ERR_OUT_OF_MEMORY || e.code == CHQ_ERR_DEVICE) throw;
      return per_batch_loop();   // a data-dependent error: the loop reports the earliest failing batch's
    }
  }
  // a foldable group that turns out not to fit the one-launch path is joined on the device instead
  auto other_path = [&]() -> std::vector<Batch> {
    if (!sizes_in_flight.cols.empty()) check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");   // (the size gather reads back into pinned memory the next call reuses)
    if (!fold) return per_batch_loop();
    try {
      return device_concat_path();
    } catch (const ChqError& e) {
      if (e.code == CHQ_ERR_OUT_OF_MEMORY || e.code == CHQ_ERR_DEVICE) throw;
      return per_batch_loop();
    }
  };
  int64_t total_rows = 0, max_rows = 0;
  const bool host_in = !recs[0].on_device;
  if (lite && !host_in) {   // (everything this loop checks per batch is in the flags reduced above)
    for (int64_t r : lite->rows) { total_rows += r; max_rows = std::max(max_rows, r); }
    bool all_dev = true;
    for (uint8_t f : lite->flags) all_dev &= (f & GroupLite::GL_ON_DEVICE) != 0;
    if (!all_dev || (has_utf8 && !fold) || !(plain || fold)) return per_batch_loop();
  } else
  for (const Batch& r : recs) {
    if (r.cols.size() != ncols || r.nrows < 2 || r.on_device == host_in) return per_batch_loop();
    for (size_t i = 0; i < ncols; ++i) {
      const Column& c = r.cols[i];
      if (c.type == T_BOOL || (c.type == T_UTF8 && !fold) || c.type != recs[0].cols[i].type || c.width != recs[0].cols[i].width) return per_batch_loop();
      if (c.validity && c.null_count != 0) {
        if (r.on_device || c.null_count > 0) return per_batch_loop();
        if (count_nulls_host(c.validity, c.offset, c.length) != 0) return per_batch_loop();
      }
    }
    total_rows += r.nrows; max_rows = std::max(max_rows, r.nrows);
  }
  const std::vector<PlanColumn> pcols = plan_columns(recs[0], aliases);
  TypedExpr te = type_expr(expr, pcols, recs[0].nrows, ctx.opt_enable_minus);
  if (te.pending_code) return per_batch_loop();
  const Node& root = te.at(te.root);
  if (root.type != T_BOOL || root.len1) return per_batch_loop();
  Lowered lw;
  try {
    lower_expr(te, te.root, pcols, lw);
  } catch (const ChqError& e) {
    if (e.code != CHQ_INTERNAL_PROGRAM_LIMIT) throw;
    return per_batch_loop();   // oversized predicate: every batch materialises its own temporaries
  }
  if (!lw.strs.empty()) return other_path();
  // ---- a device group whose string columns all hold values of ONE length (the reference's sample strings, keys, hashes):
  // fixed-width columns in disguise, as in filter_record -- one pass over every batch's offsets proves it, then the group
  // runs as a PLAIN group (value pointer of batch b = its data + its first offset) and the joined output gets the offsets
  // 0, L, 2 L, ...; per-batch results are Arrow slices of that column as before.
  if (fold && lite && !host_in && (sliced || co) && ctx.opt_uniform_utf8_rows > 0 && total_rows >= ctx.opt_uniform_utf8_rows) {
    bool pred_reads_utf8 = false;
    for (int r : lw.refs) pred_reads_utf8 |= recs[0].cols[(size_t)r].type == T_UTF8;
    // (a group whose joined strings do not fit ONE output column is cut into sub-groups first: each comes back here)
    if (!pred_reads_utf8 && finish_fold_sizes()) {
      const size_t nu = fold_utf8.size();
      std::vector<unsigned long long> h_in((nu + 1) * nb);
      for (size_t b = 0; b < nb; ++b) h_in[b] = (unsigned long long)lite->rows[b];
      for (size_t k = 0; k < nu; ++k)
        for (size_t b = 0; b < nb; ++b) h_in[(k + 1) * nb + b] = (unsigned long long)(uintptr_t)lite->values0[b * ncols + (size_t)fold_utf8[k]];
      auto d_in = make_device_buffer(h_in.size() * 8 + 16, ctx.device);
      auto d_out = make_device_buffer(nu * nb * 12 + 16, ctx.device);
      check_hip(hipMemcpyAsync(d_in->ptr, h_in.data(), h_in.size() * 8, hipMemcpyHostToDevice, ctx.stream), "upload offsets table");
      check_hip(hipMemsetAsync(d_out->ptr, 0, nu * nb * 12, ctx.stream), "memset");
      for (size_t k = 0; k < nu; ++k) {
        Utf8UniformGroupParams up{(const unsigned long long*)d_in->ptr + (k + 1) * nb, (const long long*)d_in->ptr, (int64_t)nb, (int32_t*)d_out->ptr + 3 * k * nb};
        check_hip(launch_utf8_uniform_group(up, max_rows, ctx.stream), "launch utf8_uniform_group_kernel");
      }
      std::vector<int32_t> h_out(nu * nb * 3);
      check_hip(hipMemcpyAsync(h_out.data(), d_out->ptr, h_out.size() * 4, hipMemcpyDeviceToHost, ctx.stream), "read back");
      check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
      bool uniform = true;
      std::vector<int32_t> len_of(nu, 0);
      for (size_t k = 0; k < nu && uniform; ++k) {
        len_of[k] = h_out[3 * k * nb + 1];
        const int32_t L = len_of[k];
        uniform = (L == 1 || L == 2 || L == 4 || L == 8 || L == 16) && total_rows * (int64_t)L < (1ll << 31) - 64;
        for (size_t b = 0; b < nb && uniform; ++b) {
          const int32_t* o = &h_out[3 * (k * nb + b)];
          uniform = !o[0] && o[1] == L && o[2] >= 0 && lite->data[b * ncols + (size_t)fold_utf8[k]] != nullptr;
        }
      }
      if (uniform) {
        GroupLite sub = *lite;
        std::vector<Batch> head(1);
        head[0] = recs[0];
        for (size_t k = 0; k < nu; ++k) {
          const size_t i = (size_t)fold_utf8[k];
          for (size_t b = 0; b < nb; ++b) {
            sub.values0[b * ncols + i] = lite->data[b * ncols + i] + h_out[3 * (k * nb + b) + 2];
            sub.data[b * ncols + i] = nullptr;
            sub.offset[b * ncols + i] = 0;
          }
          Column& c = head[0].cols[i];
          c.owned.clear();
          c.type = T_FIXED_OPAQUE; c.format = "w:" + std::to_string(len_of[k]); c.width = len_of[k];
          c.values = sub.values0[i]; c.data = nullptr; c.data_bytes = -1; c.offset = 0;
        }
        for (uint8_t& f : sub.flags) f &= (uint8_t)~GroupLite::GL_NO_UTF8_DATA;
        GroupInput sgi; sgi.batches = &head; sgi.lite = &sub;
        // offsets 0, L, 2 L, ... for `n` values, where the results live
        auto iota = [&](int64_t n, int32_t L) -> BufferPtr {
          if (out_on_device) {
            auto ob = make_device_buffer((size_t)(n + 1) * 4 + 16, ctx.device);
            IotaOffsetsParams ip{(int32_t*)ob->ptr, n + 1, L, 0};
            check_hip(launch_iota_offsets(ip, ctx.stream), "launch iota_offsets_kernel");
            return ob;
          }
          auto ob = make_host_buffer((size_t)(n + 1) * 4 + 16);
          int32_t* o = (int32_t*)ob->ptr;
          for (int64_t r = 0; r <= n; ++r) o[r] = (int32_t)(r * L);
          return ob;
        };
        if (sliced) {
          GroupSliced part;
          (void)filter_records_impl(ctx, sgi, aliases, expr, out_on_device, nullptr, &part);
          if (part.filled) {
            std::vector<GroupSliced*> all{&part};
            for (GroupSliced& m : part.more) all.push_back(&m);
            for (GroupSliced* g : all) {
              const int64_t n_out = g->ends.empty() ? 0 : g->ends.back();
              for (size_t k = 0; k < nu; ++k) {
                const size_t i = (size_t)fold_utf8[k];
                g->proto[i] = empty_like(recs[0].cols[i]);
                g->data[i] = g->values[i];
                g->values[i] = iota(n_out, len_of[k]);
              }
              ctx.stats.bytes_written_alg += (n_out + 1) * 4 * (int64_t)nu;
            }
            if (out_on_device) check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
            ctx.stats.bytes_read_alg += (total_rows + (int64_t)nb) * 4 * (int64_t)nu;
            *sliced = std::move(part);
            return {};
          }
        } else {
          Coalesced part;
          (void)filter_records_impl(ctx, sgi, aliases, expr, out_on_device, &part, nullptr);
          if (part.done) {
            for (size_t k = 0; k < nu; ++k) {
              const size_t i = (size_t)fold_utf8[k];
              Column& o = part.out.cols[i];
              Column u = empty_like(recs[0].cols[i]);
              u.length = part.out.nrows; u.null_count = 0; u.offset = 0;
              u.data = o.values; u.data_bytes = part.out.nrows * (int64_t)len_of[k];
              BufferPtr ob = iota(part.out.nrows, len_of[k]);
              u.values = (const uint8_t*)ob->ptr;
              u.owned = std::move(o.owned); u.owned.push_back(ob);
              o = std::move(u);
            }
            if (out_on_device) check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
            ctx.stats.bytes_read_alg += (total_rows + (int64_t)nb) * 4 * (int64_t)nu;
            *co = std::move(part);
            return {};
          }
        }
        // (the plain path declined -- e.g. a data-dependent error that the per-batch loop must attribute: go on as before)
      }
    }
  }
  if (fold) {   // the predicate itself must not read a string column, and the wide / temporaries instantiation has no Utf8 form
    for (int r : lw.refs) if (recs[0].cols[(size_t)r].type == T_UTF8) return other_path();
    if (lw.wide || lw.num_temps > 0) return other_path();
  }

  ctx.stats = chq_call_stats{};
  ctx.stats.rows_in = total_rows;
  pt.mark("typing");

  // ---- inputs: device pointers per batch and column ------------------------------------------------------
  // host batches are packed column-wise into one staging block per column and uploaded with one copy each
  std::vector<BufferPtr> staged;
  struct PtrTable {   // [batch][column], flat: one allocation for 10^5 batches
    std::vector<const uint8_t*> v; size_t ncols;
    const uint8_t** operator[](size_t b) { return v.data() + b * ncols; }
  } in_ptr{std::vector<const uint8_t*>(nb * ncols), ncols};
  if (host_in) {
    for (size_t i = 0; i < ncols; ++i) {
      const int64_t w = recs[0].cols[i].width;
      auto pack = make_host_buffer((size_t)(total_rows * w) + 16);   // recycled block: no page faults
      auto db = make_device_buffer((size_t)(total_rows * w) + 16, ctx.device);
      std::vector<int64_t> at(nb + 1, 0);
      for (size_t b = 0; b < nb; ++b) { at[b + 1] = at[b] + recs[b].nrows * w; in_ptr[b][i] = (const uint8_t*)db->ptr + at[b]; }
      parallel_ranges(nb, (size_t)at[nb], [&](size_t k0, size_t k1) {
        for (size_t b = k0; b < k1; ++b) memcpy((uint8_t*)pack->ptr + at[b], recs[b].cols[i].values0(), (size_t)(recs[b].nrows * w));
      });
      check_hip(hipMemcpyAsync(db->ptr, pack->ptr, (size_t)at[nb], hipMemcpyHostToDevice, ctx.stream), "upload packed column");
      staged.push_back(db); staged.push_back(pack);
    }
    check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  } else if (lite) {
    memcpy(in_ptr.v.data(), lite->values0.data(), nb * ncols * sizeof(const uint8_t*));
  } else {
    for (size_t b = 0; b < nb; ++b)
      for (size_t i = 0; i < ncols; ++i) in_ptr[b][i] = (const uint8_t*)recs[b].cols[i].values0();
  }

  // ---- tiling ------------------------------------------------------------------------------------------------
  // Wave-granular packing when the batches are near-uniform (every batch gets the wave count of the longest one and
  // at most a fifth of the waves idle); otherwise whole tiles per batch, described by a per-tile table.
  constexpr int64_t kWaveRows[3] = {64 * 16, 64 * 8, 64 * 8};
  constexpr int64_t kWavesPerTile[3] = {16, 4, 4};
  int tile_kind;
  int64_t wpb = 0;   // > 0: wave-granular mode
  if (lw.wide || lw.num_temps > 0) tile_kind = 2;
  else if (ctx.opt_tile_kind >= 0) tile_kind = (int)ctx.opt_tile_kind;
  else tile_kind = -1;
  {
    const int k = tile_kind < 0 ? 0 : tile_kind;
    const int64_t w = (max_rows + kWaveRows[k] - 1) / kWaveRows[k];
    if (ctx.opt_group_mode != 1 && w * (int64_t)nb * kWaveRows[k] * 4 <= total_rows * 5 && w * (int64_t)nb < (1ll << 31) - 64) {
      wpb = w; tile_kind = k;
    } else if (ctx.opt_group_mode == 2) {
      wpb = w; tile_kind = k;
    }
  }
  if (tile_kind < 0) {   // the large tile unless padding every batch to a multiple of it idles more than a quarter of the lanes
    int64_t padded = 0;
    for (size_t b = 0; b < nb; ++b) { const int64_t r = lite ? lite->rows[b] : recs[b].nrows; padded += (r + kTileRows[0] - 1) / kTileRows[0] * kTileRows[0]; }
    tile_kind = padded * 4 <= total_rows * 5 ? 0 : 1;
  }
  const int64_t tile_rows = kTileRows[tile_kind];
  int64_t ntiles = 0;
  if (wpb > 0) ntiles = (wpb * (int64_t)nb + kWavesPerTile[tile_kind] - 1) / kWavesPerTile[tile_kind];
  else for (size_t b = 0; b < nb; ++b) ntiles += ((lite ? lite->rows[b] : recs[b].nrows) + tile_rows - 1) / tile_rows;
  ensure_scratch(ctx, ntiles);
  Scratch* ds = dev_scratch(ctx);

  // ---- bitmaps (validity of any column, values of Boolean columns): wave-packed groups only --------------------------
  // `proto` = batch 0 with every column that has nulls in ANY batch marked nullable: the program is lowered / pre-decoded
  // against it (a ref that may be null must take the generic interpreter even if batch 0 happens to be null-free)
  Batch proto = recs[0];
  std::vector<char> col_nulls(ncols, 0);
  struct BitCol { int col; bool validity; size_t word; };   // word: index of {bitmap, bit offset} in a batch's table row
  std::vector<BitCol> bit_cols;
  if (need_bits) {
    if (wpb == 0) return other_path();   // ragged group: joined on the device
    // (the bitmaps' addresses and offsets come from the flat per-batch arrays: no Batch objects, as in the null-free case)
    for (size_t b = 0; b < nb; ++b) {
      if (!(lite->flags[b] & GroupLite::GL_NULLS)) continue;
      for (size_t i = 0; i < ncols; ++i) if (lite->validity[b * ncols + i]) col_nulls[i] = 1;
    }
    for (size_t i = 0; i < ncols; ++i) {
      if (col_nulls[i]) { proto.cols[i].validity = (const uint8_t*)proto.cols[i].values; proto.cols[i].null_count = 1; }   // (never read: a marker)
      else { proto.cols[i].validity = nullptr; proto.cols[i].null_count = 0; }
      if (recs[0].cols[i].type == T_BOOL) bit_cols.push_back(BitCol{(int)i, false, 0});
      if (col_nulls[i]) bit_cols.push_back(BitCol{(int)i, true, 0});
    }
    if (bit_cols.size() > 16) return other_path();   // (one null counter each in the scratch header)
  }

  // column order of the launch: the stashed predicate column goes last (see filter_record)
  std::vector<int> launch_cols;
  for (size_t i = 0; i < ncols; ++i) if (recs[0].cols[i].type != T_UTF8 && recs[0].cols[i].type != T_BOOL) launch_cols.push_back((int)i);
  FilterParams p{};
  pick_stash(p, ctx, lw, proto.cols, launch_cols, tile_kind);
  const size_t nrefs = lw.refs.size(), nout = launch_cols.size(), nu = fold ? fold_utf8.size() : 0;
  size_t stride = (wpb > 0 ? 1 : 2) + nrefs + nout + 2 * nu;   // per Utf8 column: the batch's offsets and bytes
  const size_t bits_at = need_bits ? stride : 0;               // validity bitmap + bit offset of every program ref
  if (need_bits) { stride += 2 * nrefs; for (BitCol& q : bit_cols) { q.word = stride; stride += 2; } }
  if (fold && tile_kind == 2) return other_path();

  // ---- table (+ index of every batch's last tile in tile mode): built in pinned memory, one upload ---------------
  const size_t tbl_words = (wpb > 0 ? nb : (size_t)ntiles) * stride;
  const size_t bytes_tbl = tbl_words * 8, bytes_idx = wpb > 0 ? 0 : nb * 8, bytes_cnt = nb * 8;
  ensure_pinned_table(ctx, bytes_tbl + bytes_idx + bytes_cnt);
  u64* h_tbl = (u64*)ctx.pinned_tbl;
  int64_t* h_idx = (int64_t*)(h_tbl + tbl_words);
  u64* h_cnt = (u64*)((uint8_t*)h_tbl + bytes_tbl + bytes_idx);
  {
    u64* w = h_tbl;
    int64_t tile = 0;
    auto utf8_data = [&](size_t b, size_t k) -> u64 {
      return (u64)(uintptr_t)(lite ? lite->data[b * ncols + (size_t)fold_utf8[k]] : recs[b].cols[(size_t)fold_utf8[k]].data);
    };
    for (size_t b = 0; b < nb; ++b) {
      const int64_t rows = lite ? lite->rows[b] : recs[b].nrows;
      if (wpb > 0) {
        *w++ = (u64)rows;
        for (size_t k = 0; k < nrefs; ++k) *w++ = (u64)(uintptr_t)in_ptr[b][lw.refs[k]];
        for (size_t k = 0; k < nout; ++k) *w++ = (u64)(uintptr_t)in_ptr[b][launch_cols[k]];
        for (size_t k = 0; k < nu; ++k) { *w++ = (u64)(uintptr_t)in_ptr[b][fold_utf8[k]]; *w++ = utf8_data(b, k); }
        if (need_bits) {
          const size_t at = b * ncols;
          for (size_t k = 0; k < nrefs; ++k) *w++ = (u64)(uintptr_t)lite->validity[at + (size_t)lw.refs[k]];
          for (size_t k = 0; k < nrefs; ++k) *w++ = (u64)lite->offset[at + (size_t)lw.refs[k]];
          for (const BitCol& q : bit_cols) {
            *w++ = (u64)(uintptr_t)(q.validity ? lite->validity[at + (size_t)q.col] : lite->values0[at + (size_t)q.col]);   // (a Boolean column's values0 is its bitmap)
            *w++ = (u64)lite->offset[at + (size_t)q.col];
          }
        }
        continue;
      }
      for (int64_t r0 = 0; r0 < rows; r0 += tile_rows, ++tile) {
        *w++ = (u64)r0; *w++ = (u64)rows;
        for (size_t k = 0; k < nrefs; ++k) *w++ = (u64)(uintptr_t)in_ptr[b][lw.refs[k]];
        for (size_t k = 0; k < nout; ++k) *w++ = (u64)(uintptr_t)in_ptr[b][launch_cols[k]];
        for (size_t k = 0; k < nu; ++k) { *w++ = (u64)(uintptr_t)in_ptr[b][fold_utf8[k]]; *w++ = utf8_data(b, k); }
      }
      h_idx[b] = tile - 1;   // rows >= 2: every batch owns at least one tile
    }
  }
  pt.mark("table");
  if (fold && !finish_fold_sizes()) {
    // Short strings whose JOINED output would not fit int32 offsets (the reference's batch size at config-5 scale: 10^5
    // batches, 8 GB of strings): consecutive sub-groups, each through this one-launch path -- no join on the device
    // (2.7 ms per GiB chunk) and no Batch objects; the per-batch outputs are exported per sub-group.
    bool short_strings = lite && !host_in && sliced && !co;
    for (const auto& per_batch : fold_bytes) { int64_t cap = 0; for (int64_t v : per_batch) cap += v; short_strings = short_strings && cap <= fold_rows_all * 24; }
    if (short_strings) {
      std::vector<size_t> cuts{0};
      {
        std::vector<int64_t> bytes(fold_bytes.size(), 0);
        int64_t rows = 0;
        for (size_t b = 0; b < nb; ++b) {
          bool over = rows + lite->rows[b] > (1ll << 30);
          for (size_t k = 0; k < fold_bytes.size(); ++k) over |= bytes[k] + fold_bytes[k][b] > ctx.opt_group_chunk_bytes;
          if (over && b > cuts.back()) { cuts.push_back(b); std::fill(bytes.begin(), bytes.end(), 0); rows = 0; }
          for (size_t k = 0; k < fold_bytes.size(); ++k) bytes[k] += fold_bytes[k][b];
          rows += lite->rows[b];
        }
        cuts.push_back(nb);
      }
      bool groups_of_two = cuts.size() > 2;   // (a sub-group of ONE batch would take the per-batch path: nothing gained -- e.g. ten 1 GB batches)
      for (size_t k = 0; k + 1 < cuts.size(); ++k) groups_of_two = groups_of_two && cuts[k + 1] - cuts[k] >= 2;
      if (groups_of_two) {
        chq_call_stats acc{};
        bool ok = true;
        std::vector<GroupSliced> parts;
        for (size_t k = 0; k + 1 < cuts.size() && ok; ++k) {
          const size_t b0 = cuts[k], b1 = cuts[k + 1], n = b1 - b0;
          GroupLite sub;
          sub.resize(n, ncols);
          std::copy(lite->rows.begin() + b0, lite->rows.begin() + b1, sub.rows.begin());
          std::copy(lite->flags.begin() + b0, lite->flags.begin() + b1, sub.flags.begin());
          std::copy(lite->values0.begin() + b0 * ncols, lite->values0.begin() + b1 * ncols, sub.values0.begin());
          std::copy(lite->data.begin() + b0 * ncols, lite->data.begin() + b1 * ncols, sub.data.begin());
          std::copy(lite->validity.begin() + b0 * ncols, lite->validity.begin() + b1 * ncols, sub.validity.begin());
          std::copy(lite->offset.begin() + b0 * ncols, lite->offset.begin() + b1 * ncols, sub.offset.begin());
          // the sub-group's first batch as a view built from the flat arrays (schema and flags of batch 0)
          std::vector<Batch> head(1);
          head[0] = recs[0];
          head[0].nrows = lite->rows[b0];
          for (size_t i = 0; i < ncols; ++i) {
            Column& c = head[0].cols[i];
            const size_t at = b0 * ncols + i;
            c.owned.clear();
            c.offset = lite->offset[at]; c.length = lite->rows[b0];
            c.values = c.type == T_BOOL ? lite->values0[at] : lite->values0[at] - (int64_t)(c.type == T_UTF8 ? 4 : c.width) * c.offset;
            c.data = lite->data[at]; c.data_bytes = -1;
            c.validity = lite->validity[at]; c.null_count = c.validity ? 1 : 0;   // (unknown count: may have nulls)
          }
          GroupInput sgi; sgi.batches = &head; sgi.lite = &sub;
          GroupSliced part;
          (void)filter_records_impl(ctx, sgi, aliases, expr, out_on_device, nullptr, &part);
          add_stats(acc, ctx.stats);
          ok = part.filled && part.more.empty();
          parts.push_back(std::move(part));
        }
        if (ok) {
          *sliced = std::move(parts[0]);
          for (size_t k = 1; k < parts.size(); ++k) sliced->more.push_back(std::move(parts[k]));
          ctx.stats = acc;
          return {};
        }
      }
    }
    return other_path();   // (long strings / too many bytes for one column: joined on the device)
  }
  pt.mark("utf8_sizes");
  auto d_tbl = make_device_buffer(bytes_tbl + bytes_idx + bytes_cnt + 16, ctx.device);
  check_hip(hipMemcpyAsync(d_tbl->ptr, h_tbl, bytes_tbl + bytes_idx, hipMemcpyHostToDevice, ctx.stream), "upload group table");
  u64* d_cnt = (u64*)((uint8_t*)d_tbl->ptr + bytes_tbl + bytes_idx);

  // ---- dense outputs ---------------------------------------------------------------------------------------
  std::vector<BufferPtr> dense(ncols), dense_data(ncols), fold_status;
  for (size_t i = 0; i < ncols; ++i) {
    if (recs[0].cols[i].type == T_UTF8 || recs[0].cols[i].type == T_BOOL) continue;
    dense[i] = make_device_buffer((size_t)(total_rows * recs[0].cols[i].width) + 16, ctx.device);
    ctx.stats.bytes_read_alg += total_rows * recs[0].cols[i].width;
  }
  BufferPtr g_sel, g_base;
  const int64_t nslots = need_bits ? ntiles * kWavesPerTile[tile_kind] * (kWaveRows[tile_kind] / 64) : 0;
  if (need_bits) {
    g_sel = make_device_buffer((size_t)(nslots + 8) * 8, ctx.device);
    g_base = make_device_buffer((size_t)(nslots + 8) * 8, ctx.device);
    p.sel_mask = (u64*)g_sel->ptr; p.grp_base = (u64*)g_base->ptr; p.group_bits_at = (int32_t)bits_at; p.pb.group_bits_at = (int32_t)bits_at;
  }
  for (size_t k = 0; k < nu; ++k) {   // Utf8 columns: joined offsets (from 0) and bytes, capacity = the input bytes
    const size_t i = (size_t)fold_utf8[k];
    dense[i] = make_device_buffer((size_t)(total_rows + 2) * 4, ctx.device);
    dense_data[i] = make_device_buffer((size_t)fold_cap[k] + 64, ctx.device);
    auto st = make_device_buffer((size_t)(ntiles + 1) * 8, ctx.device);
    check_hip(hipMemsetAsync(st->ptr, 0, (size_t)(ntiles + 1) * 8, ctx.stream), "memset byte-scan status");
    fold_status.push_back(st);
    Utf8Fold& f = p.utf8[k];
    f.in_offsets = nullptr; f.in_data = nullptr;   // per batch, from the table
    f.out_offsets = (int32_t*)dense[i]->ptr; f.out_data = (uint8_t*)dense_data[i]->ptr;
    f.status = (u64*)st->ptr; f.total_bytes = &ds->fold_bytes[k];
    ctx.stats.bytes_read_alg += total_rows * 8;
  }
  p.n_utf8 = (int32_t)nu;
  p.nrows = ntiles * tile_rows;   // only locates the last tile; per-tile row ranges come from the table
  p.status = dev_status(ctx);
  p.ticket = &ds->ticket; p.total = &ds->total; p.err = &ds->err;
  fill_refs(p.pb, lw, proto, {});
  for (size_t k = 0; k < nout; ++k) {
    p.outs[k].in = nullptr; p.outs[k].out = dense[launch_cols[k]]->ptr; p.outs[k].width = (uint32_t)recs[0].cols[launch_cols[k]].width;
  }
  p.n_out = (int16_t)nout;
  p.group = (const u64*)d_tbl->ptr; p.group_stride = (int64_t)stride;
  p.group_wpb = (int32_t)wpb; p.group_nb = (int32_t)nb; p.group_batch_end = d_cnt;
  p.tile_begin = 0; p.tile_end = ntiles;
  check_hip(hipMemsetAsync(ds, 0, kHeader + (size_t)(ntiles + 1) * 8, ctx.stream), "memset scratch + status");
  const int grid_cap = ctx.num_cus * (ctx.opt_grid_per_cu > 0 ? (int)ctx.opt_grid_per_cu : kGridPerCu[tile_kind]);
  if (ctx.opt_time_kernels && !ctx.ev0) { check_hip(hipEventCreate(&ctx.ev0), "hipEventCreate"); check_hip(hipEventCreate(&ctx.ev1), "hipEventCreate"); }
  if (ctx.opt_time_kernels) check_hip(hipEventRecord(ctx.ev0, ctx.stream), "hipEventRecord");
  check_hip(launch_filter(p, tile_kind, true, (int)std::min<int64_t>(ntiles, grid_cap), ctx.stream), "launch filter_fused_kernel (group)");
  if (ctx.opt_time_kernels) check_hip(hipEventRecord(ctx.ev1, ctx.stream), "hipEventRecord");
  ctx.stats.launches = 1; ctx.stats.tiles = ntiles;
  if (wpb == 0) {   // tile mode: the inclusive prefix at every batch's last tile
    GatherStatusParams gp{};
    gp.status = dev_status(ctx); gp.idx = (const int64_t*)((const uint8_t*)d_tbl->ptr + bytes_tbl);
    gp.dst = d_cnt; gp.n = (int64_t)nb;
    check_hip(launch_gather_status(gp, ctx.stream), "launch gather_status_kernel");
    ctx.stats.launches = 2;
  }
  std::vector<BufferPtr> bit_out(bit_cols.size());
  const size_t bit_bytes = (size_t)(total_rows + 31) / 32 * 4 + 16;
  for (size_t q = 0; q < bit_cols.size(); ++q) {   // one joined bitmap per Boolean column / per column with nulls
    bit_out[q] = make_device_buffer(bit_bytes, ctx.device);
    check_hip(hipMemsetAsync(bit_out[q]->ptr, 0, bit_bytes, ctx.stream), "memset bits");
    BitCompactGroupParams bp{};
    bp.sel_mask = (const u64*)g_sel->ptr; bp.grp_base = (const u64*)g_base->ptr; bp.table = (const u64*)d_tbl->ptr; bp.stride = (int64_t)stride;
    bp.word_ptr = (int32_t)bit_cols[q].word; bp.word_off = (int32_t)bit_cols[q].word + 1; bp.wpb = (int32_t)wpb; bp.nb = (int32_t)nb;
    bp.rows_per_wave = (int32_t)kWaveRows[tile_kind]; bp.out_bits = (uint32_t*)bit_out[q]->ptr;
    bp.zero_count = bit_cols[q].validity ? &ds->counters[q] : nullptr;
    check_hip(launch_bit_compact_group(bp, (int)std::min<int64_t>((wpb * (int64_t)nb + 3) / 4, (int64_t)ctx.num_cus * 8), ctx.stream), "launch bit_compact_group_kernel");
    ++ctx.stats.launches;
  }
  Scratch* hs = (Scratch*)ctx.pinned;
  check_hip(hipMemcpyAsync(hs, ds, sizeof(Scratch), hipMemcpyDeviceToHost, ctx.stream), "read back");
  check_hip(hipMemcpyAsync(h_cnt, d_cnt, bytes_cnt, hipMemcpyDeviceToHost, ctx.stream), "read back batch prefixes");
  pt.mark("alloc+launch");
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  pt.mark("kernel+readback");
  if (ctx.opt_time_kernels) { float ms = 0; check_hip(hipEventElapsedTime(&ms, ctx.ev0, ctx.ev1), "hipEventElapsedTime"); ctx.stats.kernel_ns = (int64_t)(ms * 1e6); }
  if (hs->err != ERR_NONE) return per_batch_loop();   // reports the earliest failing batch, as the reference's loop would
  (void)fold_status;
  const int64_t total = (int64_t)hs->total;
  ctx.stats.rows_out = total;
  std::vector<int64_t> out_bytes(ncols, 0);   // Utf8 columns: bytes of the joined output
  for (size_t k = 0; k < nu; ++k) {
    out_bytes[(size_t)fold_utf8[k]] = (int64_t)hs->fold_bytes[k];
    ctx.stats.bytes_read_alg += (int64_t)hs->fold_bytes[k]; ctx.stats.bytes_written_alg += (total + 1) * 4 + (int64_t)hs->fold_bytes[k];
  }
  for (size_t i = 0; i < ncols; ++i) if (recs[0].cols[i].type != T_UTF8 && recs[0].cols[i].type != T_BOOL) ctx.stats.bytes_written_alg += total * recs[0].cols[i].width;

  // ---- the joined output columns: values (a Boolean column: its joined bitmap), Utf8 bytes, validity --------------------
  struct JoinedCol { BufferPtr values, data, validity; int64_t nulls = 0; };
  std::vector<JoinedCol> joined(ncols);
  for (size_t i = 0; i < ncols; ++i) { joined[i].values = dense[i]; joined[i].data = dense_data[i]; }
  for (size_t q = 0; q < bit_cols.size(); ++q) {
    JoinedCol& jc = joined[(size_t)bit_cols[q].col];
    if (!bit_cols[q].validity) jc.values = bit_out[q];
    else if (hs->counters[q] != 0) { jc.validity = bit_out[q]; jc.nulls = (int64_t)hs->counters[q]; }   // (arrow drops an all-valid null buffer)
  }
  if (!out_on_device) {   // host result: every joined buffer comes down once
    for (size_t i = 0; i < ncols; ++i) {
      const DType ty = recs[0].cols[i].type;
      auto down = [&](BufferPtr& b, size_t bytes) {
        if (!b) return;
        auto hb = make_host_buffer(bytes + 16);
        if (bytes) check_hip(hipMemcpyAsync(hb->ptr, b->ptr, bytes, hipMemcpyDeviceToHost, ctx.stream), "download joined column");
        b = hb;
      };
      down(joined[i].values, ty == T_UTF8 ? (size_t)(total + 1) * 4 : ty == T_BOOL ? (size_t)(total + 7) / 8 : (size_t)(total * recs[0].cols[i].width));
      down(joined[i].data, (size_t)out_bytes[i]);
      down(joined[i].validity, (size_t)(total + 7) / 8);
    }
    check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  }
  if (co) {   // the joined buffers ARE the coalesced batch
    co->out.on_device = out_on_device; co->out.device_id = out_on_device ? ctx.device : -1;
    co->out.nrows = total;
    for (size_t i = 0; i < ncols; ++i) {
      Column c = empty_like(recs[0].cols[i]);
      const JoinedCol& jc = joined[i];
      c.values = (const uint8_t*)jc.values->ptr; c.length = total; c.owned.push_back(jc.values);
      if (c.type == T_UTF8) { c.data = (const uint8_t*)jc.data->ptr; c.data_bytes = out_bytes[i]; c.owned.push_back(jc.data); }
      if (jc.validity) { c.validity = (const uint8_t*)jc.validity->ptr; c.null_count = jc.nulls; c.owned.push_back(jc.validity); }
      co->out.cols.push_back(std::move(c));
    }
    int64_t prev = 0;
    for (size_t b = 0; b < nb; ++b) { co->rows.push_back((int64_t)h_cnt[b] - prev); prev = (int64_t)h_cnt[b]; }
    co->done = true;
    return {};
  }
  if (sliced) {   // the caller cuts the joined buffers into Arrow structs itself (capi.cpp: one block for the whole group)
    sliced->filled = true; sliced->on_device = out_on_device; sliced->device_id = out_on_device ? ctx.device : -1;
    for (size_t i = 0; i < ncols; ++i) {
      sliced->proto.push_back(empty_like(recs[0].cols[i]));
      sliced->values.push_back(joined[i].values);
      sliced->data.push_back(joined[i].data);
      sliced->validity.push_back(joined[i].validity);
    }
    sliced->ends.assign(h_cnt, h_cnt + nb);
    return {};
  }
  std::vector<Batch> outs(nb);
  int64_t begin = 0;
  for (size_t b = 0; b < nb; ++b) {
    const int64_t end = (int64_t)h_cnt[b];
    Batch& o = outs[b];
    o.on_device = out_on_device; o.device_id = out_on_device ? ctx.device : -1;
    o.nrows = end - begin;
    for (size_t i = 0; i < ncols; ++i) {
      Column c = empty_like(recs[0].cols[i]);
      const JoinedCol& jc = joined[i];
      c.length = o.nrows;
      c.owned.push_back(jc.values);
      if (c.type == T_UTF8 || c.type == T_BOOL || jc.validity) {
        // a slice of the joined column: shared buffers, Arrow offset = first row (one offset serves values and validity)
        c.values = (const uint8_t*)jc.values->ptr; c.offset = begin;
        if (c.type == T_UTF8) { c.data = (const uint8_t*)jc.data->ptr; c.owned.push_back(jc.data); }
        if (jc.validity) {
          c.validity = (const uint8_t*)jc.validity->ptr; c.owned.push_back(jc.validity);
          if (out_on_device) c.null_count = -1;   // unknown for the slice (Arrow C Data Interface: -1)
          else { c.null_count = count_nulls_host(c.validity, c.offset, c.length); if (c.null_count == 0) c.validity = nullptr; }
        }
      } else c.values = (const uint8_t*)jc.values->ptr + begin * c.width;
      o.cols.push_back(std::move(c));
    }
    begin = end;
  }
  return outs;
}
}  // namespace

// =================================================================================================
// project_record / compute_value
// =================================================================================================
namespace {

// deep copy of a column inside HBM (the reference Arc-clones; inputs here are only borrowed)
Column clone_device_column(Context& ctx, const Column& c) { return copy_column(ctx, c, Dir::D2D); }

Column scalar_column(Context& ctx, const Scalar& s, const std::string& name) {
  Column o;
  o.name = name; o.type = s.type; o.length = 1; o.null_count = 0; o.nullable = false;
  static const char* fmts[T_NTYPES] = {"b", "c", "s", "i", "l", "C", "S", "I", "L", "e", "f", "g", "u", ""};
  o.format = fmts[s.type]; o.width = dtype_width(s.type);
  if (s.type == T_UTF8) {
    int32_t offs[2] = {0, (int32_t)s.str.size()};
    auto ob = make_device_buffer(16, ctx.device), db = make_device_buffer(s.str.size() + 16, ctx.device);
    check_hip(hipMemcpy(ob->ptr, offs, 8, hipMemcpyHostToDevice), "memcpy");
    if (!s.str.empty()) check_hip(hipMemcpy(db->ptr, s.str.data(), s.str.size(), hipMemcpyHostToDevice), "memcpy");
    o.values = (const uint8_t*)ob->ptr; o.data = (const uint8_t*)db->ptr; o.owned = {ob, db};
  } else {
    uint64_t bits[2] = {s.null ? 0 : s.bits, 0};   // second word: the validity bitmap of a NULL value
    auto vb = make_device_buffer(16, ctx.device);
    check_hip(hipMemcpy(vb->ptr, bits, 16, hipMemcpyHostToDevice), "memcpy");
    o.values = (const uint8_t*)vb->ptr; o.owned = {vb};
    if (s.null) { o.validity = (const uint8_t*)vb->ptr + 8; o.null_count = 1; o.nullable = true; }
  }
  return o;
}

struct ProjItem {   // one computed output of a launch
  int out_index;     // position in the output batch
  int null_slot;     // counter slot, -1 when the expression cannot produce nulls
};

bool subtree_can_null(const TypedExpr& t, int ni, const std::vector<PlanColumn>& cols) {
  const Node& n = t.at(ni);
  if (n.kind == Node::COL) return cols[n.col].has_nulls;
  if (n.kind == Node::CONST) return n.cval.null;
  bool r = n.kind == Node::TOBOOL && n.from == T_UTF8;   // a bad spelling is NULL
  if (n.l >= 0) r |= subtree_can_null(t, n.l, cols);
  if (n.r >= 0) r |= subtree_can_null(t, n.r, cols);
  return r;
}

const char* format_of(DType t) {
  static const char* fmts[T_NTYPES] = {"b", "c", "s", "i", "l", "C", "S", "I", "L", "e", "f", "g", "u", ""};
  return fmts[t];
}

// Evaluate the expressions `exprs[k]` (typed trees) densely over `rec`; returns one column per expression.
std::vector<Column> evaluate_dense(Context& ctx, const Batch& rec, const std::vector<PlanColumn>& pcols,
                                   const std::vector<const TypedExpr*>& exprs) {
  const int64_t nrows = rec.nrows;
  std::vector<Column> results(exprs.size());
  for (size_t k = 0; k < exprs.size(); ++k) {
    const Node& root = exprs[k]->at(exprs[k]->root);
    Column& o = results[k];
    o.type = root.type; o.format = format_of(root.type); o.width = dtype_width(root.type); o.length = nrows;
    if (root.type == T_UTF8 || root.type == T_FIXED_OPAQUE)
      throw ChqError{CHQ_ERR_NOT_SUPPORTED, std::string("expression result of type ") + dtype_name(root.type) + " is outside this build's scope"};
  }
  if (nrows == 0) {
    for (Column& o : results) { auto vb = make_device_buffer(16, ctx.device); o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb); }
    return results;
  }
  ensure_scratch(ctx, 1);
  Scratch* ds = dev_scratch(ctx);
  Scratch* hs = (Scratch*)ctx.pinned;
  size_t k = 0;
  while (k < exprs.size()) {
    if (!lowers_alone(*exprs[k], exprs[k]->root, pcols)) {
      // too large for one program even alone: sub-trees become temporary columns first (fit_to_device), then the
      // rest is evaluated like any other expression
      Batch work; std::vector<PlanColumn> wcols; TypedExpr fitted;
      fit_to_device(ctx, rec, pcols, *exprs[k], work, wcols, fitted);
      std::vector<const TypedExpr*> one{&fitted};
      Column c = std::move(evaluate_dense(ctx, work, wcols, one)[0]);
      results[k] = std::move(c);
      hs = (Scratch*)ctx.pinned; ds = dev_scratch(ctx);
      ++k;
      continue;
    }
    // pack as many expressions as fit the program limits into one launch
    Lowered lw;
    std::vector<ProjItem> items;
    int null_slots = 0;
    size_t k0 = k;
    for (; k < exprs.size() && (int)items.size() < MAX_PROJ; ++k) {
      Lowered trial = lw;
      try {
        lower_expr(*exprs[k], exprs[k]->root, pcols, trial);
        if ((int)trial.prog.size() + 1 > MAX_INSTR) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "program too long"};
      } catch (const ChqError& e) {
        if (k == k0) throw;   // does not fit even alone
        break;
      }
      Instr st{}; st.op = OP_STORE; st.src_idx = (uint16_t)items.size(); st.src_kind = SRC_NONE;
      trial.prog.push_back(st);
      lw = std::move(trial);
      const bool can_null = subtree_can_null(*exprs[k], exprs[k]->root, pcols);
      if (can_null && null_slots >= 16) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "too many nullable outputs in one projection"};
      items.push_back(ProjItem{(int)k, can_null ? null_slots++ : -1});
    }
    ProjectParams p{};
    p.nrows = nrows; p.err = &ds->err; p.n_proj = (int32_t)items.size();
    auto str_bufs = upload_strings(ctx, lw);
    fill_refs(p.pb, lw, rec, str_bufs);
    for (size_t i = 0; i < items.size(); ++i) {
      Column& o = results[items[i].out_index];
      const size_t vbytes = o.type == T_BOOL ? (size_t)((nrows + 63) / 64) * 8 + 16 : (size_t)nrows * o.width + 16;
      auto vb = make_device_buffer(vbytes, ctx.device);
      o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
      ProjOut po{};
      po.values = vb->ptr; po.type = o.type;
      if (items[i].null_slot >= 0) {
        auto nb = make_device_buffer((size_t)((nrows + 63) / 64) * 8 + 16, ctx.device);
        o.validity = (const uint8_t*)nb->ptr; o.owned.push_back(nb);
        po.validity = (u64*)nb->ptr; po.null_count = &ds->counters[items[i].null_slot];
      }
      p.outs[i] = po;
    }
    check_hip(hipMemsetAsync(ds, 0, sizeof(Scratch), ctx.stream), "memset scratch");
    const int tile_kind = pick_tile_kind(ctx, lw, nrows);
    const int64_t ntiles = (nrows + kTileRows[tile_kind] - 1) / kTileRows[tile_kind];
    const int64_t gcap = tile_kind == 0 ? (int64_t)ctx.num_cus * 2 : (int64_t)ctx.num_cus * 8;
    const int64_t nfull = nrows / kTileRows[tile_kind];
    if (nrows >= ctx.opt_split_rows && nfull > 0) {
      p.tile_begin = 0; p.tile_end = nfull;
      check_hip(launch_project(p, tile_kind, false, (int)std::min<int64_t>(nfull, gcap), ctx.stream), "launch project_kernel");
      if (nfull < ntiles) {
        p.tile_begin = nfull; p.tile_end = ntiles;
        check_hip(launch_project(p, tile_kind, true, 1, ctx.stream), "launch project_kernel (tail)");
        ++ctx.stats.launches;
      }
    } else {
      p.tile_begin = 0; p.tile_end = ntiles;
      check_hip(launch_project(p, tile_kind, true, (int)std::min<int64_t>(ntiles, gcap), ctx.stream), "launch project_kernel");
    }
    ++ctx.stats.launches;
    check_hip(hipMemcpyAsync(hs, ds, sizeof(Scratch), hipMemcpyDeviceToHost, ctx.stream), "read back");
    check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
    if (hs->err != ERR_NONE) {
      // Several expressions shared the launch and each numbers its nodes from zero, so the smallest (node, row) key may
      // belong to a later select item.  The reference evaluates the items one after the other: do the same to find
      // the error it would have reported.
      if (items.size() > 1) {
        for (const ProjItem& it : items) {
          std::vector<const TypedExpr*> one{exprs[it.out_index]};
          (void)evaluate_dense(ctx, rec, pcols, one);   // throws at the first failing item
        }
      }
      throw_device_error(hs->err);
    }
    for (const ProjItem& it : items) {
      Column& o = results[it.out_index];
      if (it.null_slot >= 0) { o.null_count = (int64_t)hs->counters[it.null_slot]; if (o.null_count == 0) o.validity = nullptr; }
    }
  }
  return results;
}

// the result of one compute_value call: a passthrough column, a literal-built length-1 array, or a program
struct Evaluated {
  enum Kind { PASSTHROUGH, SCALAR, COMPUTED } kind;
  int col = -1;
  Scalar value;
  TypedExpr typed;
  bool is_scalar = false;
};

Evaluated classify(Context& ctx, const Batch& rec, const Expr& e, const std::vector<PlanColumn>& pcols) {
  Evaluated ev;
  ev.typed = typed(ctx, rec, pcols, e);
  const Node& root = ev.typed.at(ev.typed.root);
  ev.is_scalar = root.is_scalar;
  if (root.kind == Node::COL) { ev.kind = Evaluated::PASSTHROUGH; ev.col = root.col; }
  else if (root.len1) { ev.kind = Evaluated::SCALAR; ev.value = fold_constant(ev.typed, ev.typed.root); }
  else ev.kind = Evaluated::COMPUTED;
  return ev;
}

}  // namespace

Column compute_value(Context& ctx, const Batch& rec, const std::vector<PlanColumn>& pcols, const Expr& expr, bool* is_scalar) {
  ctx.stats = chq_call_stats{};
  Evaluated ev = classify(ctx, rec, expr, pcols);
  if (is_scalar) *is_scalar = ev.is_scalar;
  Column out;
  switch (ev.kind) {
    case Evaluated::PASSTHROUGH: out = clone_device_column(ctx, rec.cols[ev.col]); check_hip(hipStreamSynchronize(ctx.stream), "sync"); break;
    case Evaluated::SCALAR: out = scalar_column(ctx, ev.value, ""); break;
    default: { std::vector<const TypedExpr*> v{&ev.typed}; out = std::move(evaluate_dense(ctx, rec, pcols, v)[0]); } break;
  }
  out.nullable = out.null_count > 0;
  return out;
}

Batch project_record(Context& ctx, const std::vector<chq_select_item>& fields, const Batch& rec,
                     const std::vector<PlanColumn>& pcols) {
  ctx.stats = chq_call_stats{};
  Batch out;
  out.on_device = true; out.device_id = ctx.device;
  std::vector<Evaluated> evs;            // computed items, evaluated together after the walk
  std::vector<int> computed_slot;        // output column index of each computed item
  size_t unnamed_idx = 0;
  for (const chq_select_item& f : fields) {
    switch (f.kind) {
      case CHQ_ITEM_WILDCARD:   // RU/record_projection.rs:27-32
        for (const Column& c : rec.cols) out.cols.push_back(clone_device_column(ctx, c));
        break;
      case CHQ_ITEM_QUALIFIED_WILDCARD:
        if (!evs.empty()) {
          std::vector<const TypedExpr*> ptrs;
          for (auto& p : evs) ptrs.push_back(&p.typed);
          (void)evaluate_dense(ctx, rec, pcols, ptrs);
        }
        throw ChqError{CHQ_ERR_PROJECT_NOT_IMPLEMENTED, "not implemented: SelectItem::QualifiedWildcard"};
      case CHQ_ITEM_UNNAMED_EXPR:
      case CHQ_ITEM_EXPR_WITH_ALIAS: {
        if (!f.expr) throw ChqError{CHQ_ERR_INVALID_HANDLE, "select item without expression"};
        const Expr& e = *(const Expr*)f.expr;
        Evaluated ev;
        try {
          ev = classify(ctx, rec, e, pcols);
        } catch (const ChqError&) {
          // the reference evaluates the items in order: errors of earlier computed items come first
          if (!evs.empty()) {
            std::vector<const TypedExpr*> ptrs;
            for (auto& p : evs) ptrs.push_back(&p.typed);
            (void)evaluate_dense(ctx, rec, pcols, ptrs);
          }
          throw;
        }
        std::string name;
        if (f.kind == CHQ_ITEM_EXPR_WITH_ALIAS) name = f.alias ? f.alias : "";
        else if (e.kind == Expr::IDENT) name = e.text;                 // RU/record_projection.rs:41-48
        else name = "unnamed_" + std::to_string(unnamed_idx);          // :49-53
        if (f.kind == CHQ_ITEM_UNNAMED_EXPR) ++unnamed_idx;           // :58, counts identifiers too
        Column col;
        if (ev.kind == Evaluated::PASSTHROUGH) col = clone_device_column(ctx, rec.cols[ev.col]);
        else if (ev.kind == Evaluated::SCALAR) col = scalar_column(ctx, ev.value, name);
        else { computed_slot.push_back((int)out.cols.size()); evs.push_back(std::move(ev)); }
        col.name = name;
        out.cols.push_back(std::move(col));
      } break;
      default: throw ChqError{CHQ_ERR_INVALID_HANDLE, "unknown select item kind"};
    }
  }
  if (!evs.empty()) {
    std::vector<const TypedExpr*> ptrs;
    for (auto& ev : evs) ptrs.push_back(&ev.typed);
    std::vector<Column> cols = evaluate_dense(ctx, rec, pcols, ptrs);
    for (size_t i = 0; i < cols.size(); ++i) {
      std::string name = out.cols[computed_slot[i]].name;
      out.cols[computed_slot[i]] = std::move(cols[i]);
      out.cols[computed_slot[i]].name = name;
    }
  }
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  // Field nullability: wildcard fields keep the schema flag; computed / identifier fields use
  // Array::is_nullable() = null_count > 0 (RU/record_projection.rs:45-47, 51-53, 62-66)
  {
    size_t k = 0;
    for (const chq_select_item& f : fields) {
      if (f.kind == CHQ_ITEM_WILDCARD) { k += rec.cols.size(); continue; }
      out.cols[k].nullable = out.cols[k].null_count > 0;
      ++k;
    }
  }
  // RecordBatch::try_new (RU/record_projection.rs:72-73)
  if (out.cols.empty()) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "must either specify a row count or at least one column"};
  const int64_t len = out.cols[0].length;
  for (const Column& c : out.cols) {
    if (c.length != len) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "all columns in a record batch must have the same length"};
    if (!c.nullable && c.null_count > 0)
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "Column '" + c.name + "' is declared as non-nullable but contains null values"};
  }
  out.nrows = len;
  return out;
}

// =================================================================================================
// filter_project_fused: filter_record + project_record in ONE pass when the inputs allow it.
// Returns false when the call is outside the fused kernel's scope or when the device flagged any error -- the caller
// then runs the two reference steps, which produce the reference's result or error exactly.
// =================================================================================================
bool filter_project_fused(Context& ctx, const Batch& rec, const std::vector<PlanColumn>& pcols, const Expr& pred,
                          const std::vector<chq_select_item>& fields, Batch* result) {
  const int64_t nrows = rec.nrows;
  if (ctx.opt_fuse == 0 || nrows < 2 || fields.empty()) return false;
  auto plain = [&](int ci) {   // fixed-width, no nulls
    const Column& c = rec.cols[ci];
    return c.type != T_BOOL && c.type != T_UTF8 && c.width > 0 && !(c.validity && c.null_count != 0);
  };
  struct Item { std::string name; bool nullable; int copy_col; int store_slot; DType type; };
  std::vector<Item> items;
  std::vector<TypedExpr> computed;
  Lowered lwp, lwq;
  try {
    TypedExpr tp = type_expr(pred, pcols, nrows, ctx.opt_enable_minus);
    if (tp.pending_code) return false;
    const Node& proot = tp.at(tp.root);
    if (proot.type != T_BOOL || proot.len1) return false;
    lower_expr(tp, tp.root, pcols, lwp);
    if (!lwp.strs.empty()) return false;
    for (int ci : lwp.refs) if (!plain(ci)) return false;

    size_t unnamed_idx = 0;
    for (const chq_select_item& f : fields) {
      if (f.kind == CHQ_ITEM_WILDCARD) {
        for (size_t ci = 0; ci < rec.cols.size(); ++ci) {
          if (!plain((int)ci)) return false;
          items.push_back(Item{rec.cols[ci].name, rec.cols[ci].nullable, (int)ci, -1, rec.cols[ci].type});
        }
        continue;
      }
      if ((f.kind != CHQ_ITEM_UNNAMED_EXPR && f.kind != CHQ_ITEM_EXPR_WITH_ALIAS) || !f.expr) return false;
      const Expr& e = *(const Expr*)f.expr;
      TypedExpr te = type_expr(e, pcols, nrows, ctx.opt_enable_minus);
      if (te.pending_code) return false;
      const Node& root = te.at(te.root);
      if (root.len1) return false;   // a literal-built column: RecordBatch::try_new decides on the filtered length
      std::string name;
      if (f.kind == CHQ_ITEM_EXPR_WITH_ALIAS) name = f.alias ? f.alias : "";
      else if (e.kind == Expr::IDENT) name = e.text;
      else name = "unnamed_" + std::to_string(unnamed_idx);
      if (f.kind == CHQ_ITEM_UNNAMED_EXPR) ++unnamed_idx;
      if (root.kind == Node::COL) {
        if (!plain(root.col)) return false;
        items.push_back(Item{name, false, root.col, -1, rec.cols[root.col].type});
        continue;
      }
      if (root.type == T_BOOL || root.type == T_UTF8 || root.type == T_F16 || root.type == T_FIXED_OPAQUE) return false;
      if ((int)computed.size() >= MAX_PROJ) return false;
      lower_expr(te, te.root, pcols, lwq);
      Instr st{}; st.op = OP_STORE; st.src_idx = (uint16_t)computed.size(); st.src_kind = SRC_NONE;
      lwq.prog.push_back(st);
      if ((int)lwq.prog.size() > MAX_INSTR || !lwq.strs.empty()) return false;
      items.push_back(Item{name, false, -1, (int)computed.size(), root.type});
      computed.push_back(std::move(te));
    }
    for (int ci : lwq.refs) if (!plain(ci)) return false;
  } catch (const ChqError&) {
    return false;   // static errors are the two-step path's to report, in the reference's order
  }
  int n_copy = 0;
  for (const Item& it : items) n_copy += it.copy_col >= 0;
  if (n_copy > MAX_FUSED_COPY) return false;
  if (ctx.opt_fuse == 1 && nrows > (1 << 18)) {
    // (small batches are latency-bound: one launch and one synchronisation always beat two of each)
    // Worth it?  The single pass is instruction-bound (interpreter + compacting stores in one kernel) while the two
    // steps run near the HBM roofline, so it only pays when it moves clearly fewer bytes: the filter step copies the
    // WHOLE table, the single pass touches only what the predicate and the select items need.  Selectivity is not
    // known yet; 0.5 is assumed.  (Measured: 8-column table, 3 columns used: 3.2 ms vs 7.4 ms; config 3, 5 columns,
    // 4 used: 12.9 ms vs 12.4 ms.)
    double w_table = 0, w_pred = 0, w_proj = 0, w_out = 0;
    std::vector<char> in_pred(rec.cols.size(), 0), in_proj(rec.cols.size(), 0);
    for (const Column& c : rec.cols) w_table += c.width > 0 ? c.width : 16;
    for (int ci : lwp.refs) if (!in_pred[ci]) { in_pred[ci] = 1; w_pred += rec.cols[ci].width; }
    for (int ci : lwq.refs) if (!in_proj[ci]) { in_proj[ci] = 1; w_proj += rec.cols[ci].width; }
    for (const Item& it : items) {
      if (it.copy_col >= 0 && !in_proj[it.copy_col]) { in_proj[it.copy_col] = 1; w_proj += rec.cols[it.copy_col].width; }
      w_out += it.copy_col >= 0 ? rec.cols[it.copy_col].width : dtype_width(it.type);
    }
    const double two_steps = w_table + 0.5 * w_table + 0.5 * (w_proj + w_out);
    const double one_pass = w_pred + w_proj + 0.5 * w_out;
    if (two_steps < 2.0 * one_pass) return false;
  }

  const int tile_kind = (lwp.wide || lwq.wide || lwp.num_temps > 0 || lwq.num_temps > 0) ? 2
                        : (ctx.opt_tile_kind == 0 || ctx.opt_tile_kind == 1) ? (int)ctx.opt_tile_kind
                        : (nrows >= (1 << 18) ? 0 : 1);
  const int64_t tile_rows = kTileRows[tile_kind];
  const int64_t ntiles = (nrows + tile_rows - 1) / tile_rows;
  ensure_scratch(ctx, ntiles);
  Scratch* ds = dev_scratch(ctx);
  Scratch* hs = (Scratch*)ctx.pinned;

  ctx.stats = chq_call_stats{};
  ctx.stats.rows_in = nrows;
  Batch out;
  out.on_device = true; out.device_id = ctx.device;
  FusedParams p{};
  p.nrows = nrows; p.tile_begin = 0; p.tile_end = ntiles;
  p.status = dev_status(ctx); p.ticket = &ds->ticket; p.total = &ds->total; p.err = &ds->err;
  fill_refs(p.pred, lwp, rec, {});
  fill_refs(p.proj, lwq, rec, {});
  p.n_proj = (int32_t)computed.size();
  std::vector<char> read_once(rec.cols.size(), 0);
  for (int ci : lwp.refs) read_once[ci] = 1;
  for (int ci : lwq.refs) read_once[ci] = 1;
  int64_t out_width = 0;
  for (const Item& it : items) {
    Column o;
    o.name = it.name; o.nullable = it.nullable; o.type = it.type;
    if (it.copy_col >= 0) {
      const Column& c = rec.cols[it.copy_col];
      o.format = c.format; o.width = c.width;
      read_once[it.copy_col] = 1;
    } else {
      o.format = format_of(it.type); o.width = dtype_width(it.type);
    }
    auto vb = make_device_buffer((size_t)nrows * o.width + 16, ctx.device);
    o.values = (const uint8_t*)vb->ptr; o.owned.push_back(vb);
    if (it.copy_col >= 0) {
      OutCol& oc = p.copies[p.n_copy++];
      oc.in = rec.cols[it.copy_col].values0(); oc.out = vb->ptr; oc.width = (uint32_t)o.width;
    } else {
      ProjOut& po = p.outs[it.store_slot];
      po.values = vb->ptr; po.type = (uint8_t)it.type;
    }
    out_width += o.width;
    out.cols.push_back(std::move(o));
  }
  for (size_t ci = 0; ci < rec.cols.size(); ++ci) if (read_once[ci]) ctx.stats.bytes_read_alg += nrows * rec.cols[ci].width;

  check_hip(hipMemsetAsync(ds, 0, kHeader + (size_t)(ntiles + 1) * 8, ctx.stream), "memset scratch + status");
  const int grid_cap = ctx.num_cus * (ctx.opt_grid_per_cu > 0 ? (int)ctx.opt_grid_per_cu : kGridPerCu[tile_kind]);
  if (ctx.opt_time_kernels && !ctx.ev0) { check_hip(hipEventCreate(&ctx.ev0), "hipEventCreate"); check_hip(hipEventCreate(&ctx.ev1), "hipEventCreate"); }
  if (ctx.opt_time_kernels) check_hip(hipEventRecord(ctx.ev0, ctx.stream), "hipEventRecord");
  check_hip(launch_filter_project(p, tile_kind, (int)std::min<int64_t>(ntiles, grid_cap), ctx.stream), "launch filter_project_kernel");
  if (ctx.opt_time_kernels) check_hip(hipEventRecord(ctx.ev1, ctx.stream), "hipEventRecord");
  check_hip(hipMemcpyAsync(hs, ds, sizeof(Scratch), hipMemcpyDeviceToHost, ctx.stream), "read back");
  check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  if (ctx.opt_time_kernels) { float ms = 0; check_hip(hipEventElapsedTime(&ms, ctx.ev0, ctx.ev1), "hipEventElapsedTime"); ctx.stats.kernel_ns = (int64_t)(ms * 1e6); }
  if (hs->err != ERR_NONE) return false;
  const int64_t total = (int64_t)hs->total;
  for (Column& o : out.cols) o.length = total;
  out.nrows = total;
  ctx.stats.rows_out = total; ctx.stats.tiles = ntiles; ctx.stats.launches = 1;
  ctx.stats.bytes_written_alg = total * out_width;
  *result = std::move(out);
  return true;
}

// =================================================================================================
// project_record_host: project_record for a HOST batch with a host result (the materialize task's calling pattern,
// materialize_files_task.rs:110).  Only the columns the computed select items read are uploaded and only the computed
// columns come back; identifier and wildcard items -- columns handed through unchanged, often the wide Utf8 ones --
// are copied host to host and never cross PCIe.  False = outside its scope (an item fails typing, literal-only items,
// QualifiedWildcard ...): the general path decides, in the reference's order.
// =================================================================================================
bool project_record_host(Context& ctx, const std::vector<chq_select_item>& fields, const Batch& rec,
                         const chq_table_aliases* aliases, Batch* result) {
  if (rec.on_device || fields.empty()) return false;
  const int64_t nrows = rec.nrows;
  Batch host = rec;   // shallow: null counts resolved below
  for (Column& c : host.cols) if (c.validity && c.null_count < 0) c.null_count = count_nulls_host(c.validity, c.offset, c.length);
  const std::vector<PlanColumn> pcols = plan_columns(host, aliases);
  struct Item { int copy_col; int computed; std::string name; bool keep_schema_flag; };
  std::vector<Item> items;
  std::vector<TypedExpr> computed;
  try {
    size_t unnamed_idx = 0;
    for (const chq_select_item& f : fields) {
      if (f.kind == CHQ_ITEM_WILDCARD) {
        for (size_t ci = 0; ci < host.cols.size(); ++ci) items.push_back(Item{(int)ci, -1, host.cols[ci].name, true});
        continue;
      }
      if ((f.kind != CHQ_ITEM_UNNAMED_EXPR && f.kind != CHQ_ITEM_EXPR_WITH_ALIAS) || !f.expr) return false;
      const Expr& e = *(const Expr*)f.expr;
      TypedExpr te = type_expr(e, pcols, nrows, ctx.opt_enable_minus);
      if (te.pending_code) return false;
      const Node& root = te.at(te.root);
      if (root.len1) return false;
      std::string name;
      if (f.kind == CHQ_ITEM_EXPR_WITH_ALIAS) name = f.alias ? f.alias : "";
      else if (e.kind == Expr::IDENT) name = e.text;                 // RU/record_projection.rs:41-48
      else name = "unnamed_" + std::to_string(unnamed_idx);          // :49-53
      if (f.kind == CHQ_ITEM_UNNAMED_EXPR) ++unnamed_idx;           // :58, counts identifiers too
      if (root.kind == Node::COL) { items.push_back(Item{root.col, -1, name, false}); continue; }
      items.push_back(Item{-1, (int)computed.size(), name, false});
      computed.push_back(std::move(te));
    }
  } catch (const ChqError&) {
    return false;
  }
  for (const Item& it : items)   // RecordBatch::try_new would refuse: let the general path say so
    if (it.copy_col >= 0 && it.keep_schema_flag && !host.cols[it.copy_col].nullable && host.cols[it.copy_col].null_count > 0) return false;

  ctx.stats = chq_call_stats{};
  ctx.stats.rows_in = nrows; ctx.stats.rows_out = nrows;
  std::vector<Column> results;
  if (!computed.empty()) {
    std::vector<char> needed(host.cols.size(), 0);
    for (const TypedExpr& te : computed) for (const Node& n : te.nodes) if (n.kind == Node::COL) needed[n.col] = 1;
    Batch dev;
    dev.nrows = nrows; dev.on_device = true; dev.device_id = ctx.device;
    for (size_t ci = 0; ci < host.cols.size(); ++ci) {
      if (needed[ci]) dev.cols.push_back(copy_column(ctx, host.cols[ci], Dir::H2D));
      else { Column ph = empty_like(host.cols[ci]); ph.length = nrows; dev.cols.push_back(std::move(ph)); }   // never dereferenced
    }
    std::vector<const TypedExpr*> ptrs;
    for (const TypedExpr& te : computed) ptrs.push_back(&te);
    std::vector<Column> dcols = evaluate_dense(ctx, dev, pcols, ptrs);
    for (Column& c : dcols) results.push_back(copy_column(ctx, c, Dir::D2H));
    check_hip(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize");
  }
  Batch out;
  out.on_device = false; out.device_id = -1; out.nrows = nrows;
  for (const Item& it : items) {
    Column c;
    if (it.copy_col >= 0) {
      c = copy_column(ctx, host.cols[it.copy_col], Dir::H2H);
      c.nullable = it.keep_schema_flag ? host.cols[it.copy_col].nullable : c.null_count > 0;
      if (c.validity && c.null_count == 0) c.validity = nullptr;
    } else {
      c = std::move(results[(size_t)it.computed]);
      c.nullable = c.null_count > 0;
    }
    c.name = it.name;
    out.cols.push_back(std::move(c));
  }
  if (out.cols.empty()) return false;
  *result = std::move(out);
  return true;
}

// =================================================================================================
// describe_plan: the host half of a call (typing, coercion, constant folding, lowering) without a GPU
// =================================================================================================
std::string describe_plan(const ArrowSchema* schema, const chq_table_aliases* aliases, const Expr& expr, int64_t nrows,
                          bool enable_minus) {
  if (!schema || !schema->format || strcmp(schema->format, "+s") != 0)
    throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "record batch schema must be a struct"};
  std::vector<PlanColumn> cols;
  for (int64_t i = 0; i < schema->n_children; ++i) {
    const ArrowSchema* cs = schema->children[i];
    PlanColumn p;
    int width = 0;
    p.name = cs->name ? cs->name : "";
    parse_format(cs->format, &p.type, &width);
    p.format = cs->format ? cs->format : ""; p.width = width;
    p.has_nulls = (cs->flags & ARROW_FLAG_NULLABLE) != 0;
    p.alias_entry_present = aliases ? (int)i < aliases->n_columns : true;
    if (aliases && (int)i < aliases->n_columns)
      for (int k = 0; k < aliases->columns[i].n; ++k) p.aliases.push_back(aliases->columns[i].aliases[k]);
    cols.push_back(std::move(p));
  }
  TypedExpr te = type_expr(expr, cols, nrows, enable_minus);
  if (te.pending_code) throw ChqError{te.pending_code, te.pending_msg};
  const Node& root = te.at(te.root);
  static const char* kOps[] = {"LOAD", "ADD", "SUB", "MUL", "DIV", "REM", "EQ", "NE", "LT", "LE", "GT", "GE", "AND", "OR",
                               "CAST", "TOBOOL", "SPILL", "STRCMP", "STORE"};
  static const char* kSrc[] = {"-", "col", "const", "tmp", "btmp"};
  std::string out = std::string("result ") + dtype_name(root.type) + " scalar=" + (root.is_scalar ? "1" : "0") + " len1=" + (root.len1 ? "1" : "0") + "\n";
  if (root.len1) {
    Scalar v = fold_constant(te, te.root);
    char hex[40];
    snprintf(hex, sizeof hex, "%016llx", (unsigned long long)v.bits);
    out += std::string("value ") + (v.type == T_UTF8 ? "'" + v.str + "'" : std::string("0x") + hex) + "\n";
    return out;
  }
  if (root.kind == Node::COL) { out += "column " + std::to_string(root.col) + "\n"; return out; }
  Lowered lw;
  try {
    lower_expr(te, te.root, cols, lw);
  } catch (const ChqError& e) {
    if (e.code != CHQ_INTERNAL_PROGRAM_LIMIT) throw;
    // valid, but more than one device program: the engine evaluates sub-trees into temporary columns first (fit_to_device)
    out += "split " + e.msg + "\n";
    return out;
  }
  out += "program wide=" + std::to_string((int)lw.wide) + " num_temps=" + std::to_string(lw.num_temps) + " refs=";
  for (size_t i = 0; i < lw.refs.size(); ++i) out += (i ? "," : "") + std::to_string(lw.refs[i]);
  out += "\n";
  for (const Instr& in : lw.prog) {
    char line[160];
    snprintf(line, sizeof line, "  %-6s %-7s %s%s%s idx=%u%s imm=0x%llx\n", in.op < 19 ? kOps[in.op] : "?", dtype_name((DType)in.type),
             in.src_kind < 5 ? kSrc[in.src_kind] : "?", in.src_kind == SRC_COL ? ":" : "", in.src_kind == SRC_COL ? dtype_name((DType)in.src_type) : "",
             (unsigned)in.src_idx, (in.flags & IF_REV) ? " rev" : "", (unsigned long long)in.imm);
    out += line;
  }
  return out;
}

}  // namespace chq
