// capi.cpp -- the extern "C" surface declared in include/chq.h.
#include <algorithm>
#include <cstring>
#include <exception>
#include <new>
#include <thread>

#include "engine.hpp"
#include "parquet.hpp"
#include <atomic>

using namespace chq;

struct chq_ctx { Context c; };
struct chq_expr { Expr e; };

namespace {

template <typename F>
chq_status guarded(chq_ctx* ctx, F&& f) {
  try {
    f();
    if (ctx) ctx->c.last_error.clear();
    return CHQ_OK;
  } catch (const ChqError& e) {
    if (ctx) ctx->c.last_error = e.msg;
    return e.code == CHQ_INTERNAL_PROGRAM_LIMIT ? CHQ_ERR_NOT_SUPPORTED : (chq_status)e.code;
  } catch (const std::bad_alloc&) {
    if (ctx) ctx->c.last_error = "out of host memory";
    return CHQ_ERR_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    if (ctx) ctx->c.last_error = e.what();
    return CHQ_ERR_DEVICE;
  }
}

void mark_released(ArrowDeviceArray* out, ArrowSchema* out_schema) {
  if (out) { memset(out, 0, sizeof(*out)); }
  if (out_schema) { memset(out_schema, 0, sizeof(*out_schema)); }
}

void finish(Context& c, Batch&& result_dev, int out_device, ArrowDeviceArray* out, ArrowSchema* out_schema) {
  if (out_device == ARROW_DEVICE_ROCM) export_batch(std::move(result_dev), ARROW_DEVICE_ROCM, out, out_schema);
  else if (out_device == ARROW_DEVICE_CPU) { Batch h = to_host(c, result_dev); export_batch(std::move(h), ARROW_DEVICE_CPU, out, out_schema); }
  else throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "out_device must be ARROW_DEVICE_CPU or ARROW_DEVICE_ROCM"};
}

void require(const void* p, const char* what) { if (!p) throw ChqError{CHQ_ERR_INVALID_HANDLE, std::string("null ") + what}; }

// f(i) for i in [0, n) on a few host threads when the group is large: importing / exporting 10^5 Arrow structs one after
// the other (~0.3 + 0.4 us each) used to be 93 % of a group call (round 1).  The first error, if any, is rethrown.
template <class F>
void for_each_parallel(int n, F&& f) {
  if (n < 2048) { for (int i = 0; i < n; ++i) f(i); return; }
  pool_ranges((size_t)n, 1024, [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; ++i) f((int)i); });
}

// ---- the outputs of a one-launch group call, exported from ONE block ---------------------------------------------------
// Every output batch of such a call is a slice of the same dense buffers (engine.hpp: GroupSliced).  Exporting them one by
// one (export_batch) costs ~20 allocations per batch and, worse, an atomic reference-count increment per column and batch
// on the SAME few shared buffers from sixteen threads -- 7.5 ms for 12 500 batches, far more than the kernel.  Here the
// Arrow structs of all batches live in a handful of arrays inside one reference-counted block that also holds the buffers:
// exporting a batch writes ~0.6 KB of plain structs, releasing one decrements one counter.
struct GroupBlock {
  std::atomic<int64_t> live{0};          // exported structs whose release has not run: (1 + C) arrays + (1 + C) schemas per batch
  std::vector<BufferPtr> buffers;
  std::vector<std::string> names, formats;
  std::string struct_format = "+s", empty;
  // (plain arrays, NOT value-initialised: every element is written by the export loop, which runs on the thread pool -- zero-
  // filling 60 MB of fresh pages on the calling thread first cost 12 ms for 100 000 batches)
  std::unique_ptr<ArrowArray[]> child_arrays;    // [nb * C]
  std::unique_ptr<ArrowArray*[]> child_ptrs;     // [nb * C]
  std::unique_ptr<const void*[]> bufs;           // [nb * C * 3]
  std::unique_ptr<const void*[]> parent_bufs;    // [nb] (the struct array's null validity)
  std::unique_ptr<ArrowSchema[]> child_schemas;  // [nb * C]
  std::unique_ptr<ArrowSchema*[]> schild_ptrs;   // [nb * C]
};
void group_drop(GroupBlock* blk) { if (blk->live.fetch_sub(1, std::memory_order_acq_rel) == 1) delete blk; }
void group_release_child_array(ArrowArray* a) {
  if (!a || !a->release) return;
  auto* blk = (GroupBlock*)a->private_data;
  a->release = nullptr; a->private_data = nullptr;
  group_drop(blk);
}
void group_release_array(ArrowArray* a) {
  if (!a || !a->release) return;
  auto* blk = (GroupBlock*)a->private_data;
  for (int64_t i = 0; i < a->n_children; ++i) if (a->children[i] && a->children[i]->release) a->children[i]->release(a->children[i]);
  a->release = nullptr; a->private_data = nullptr;
  group_drop(blk);
}
void group_release_child_schema(ArrowSchema* s) {
  if (!s || !s->release) return;
  auto* blk = (GroupBlock*)s->private_data;
  s->release = nullptr; s->private_data = nullptr;
  group_drop(blk);
}
void group_release_schema(ArrowSchema* s) {
  if (!s || !s->release) return;
  auto* blk = (GroupBlock*)s->private_data;
  for (int64_t i = 0; i < s->n_children; ++i) if (s->children[i] && s->children[i]->release) s->children[i]->release(s->children[i]);
  s->release = nullptr; s->private_data = nullptr;
  group_drop(blk);
}

void export_group(GroupSliced&& g, int device_type, ArrowDeviceArray* outs, ArrowSchema* out_schemas) {
  const size_t nb = g.ends.size(), C = g.proto.size();
  auto* blk = new GroupBlock();
  for (size_t i = 0; i < C; ++i) {
    blk->names.push_back(g.proto[i].name); blk->formats.push_back(g.proto[i].format);
    blk->buffers.push_back(g.values[i]);
    if (g.data[i]) blk->buffers.push_back(g.data[i]);
  }
  const size_t cells = std::max<size_t>(1, nb * C);
  blk->child_arrays.reset(new ArrowArray[cells]); blk->child_ptrs.reset(new ArrowArray*[cells]); blk->bufs.reset(new const void*[cells * 3]);
  blk->parent_bufs.reset(new const void*[std::max<size_t>(1, nb)]); blk->child_schemas.reset(new ArrowSchema[cells]); blk->schild_ptrs.reset(new ArrowSchema*[cells]);
  blk->live.store((int64_t)(nb * (2 + 2 * C)));
  std::vector<const uint8_t*> vbase(C), dbase(C), nbase(C);
  for (size_t i = 0; i < C; ++i) {
    vbase[i] = (const uint8_t*)g.values[i]->ptr; dbase[i] = g.data[i] ? (const uint8_t*)g.data[i]->ptr : nullptr;
    nbase[i] = (i < g.validity.size() && g.validity[i]) ? (const uint8_t*)g.validity[i]->ptr : nullptr;
    if (nbase[i]) blk->buffers.push_back(g.validity[i]);
  }
  const bool host_out = device_type != ARROW_DEVICE_ROCM;
  pool_ranges(nb, 1024, [&](size_t b0, size_t b1) {
    for (size_t b = b0; b < b1; ++b) {
      const int64_t begin = b ? g.ends[b - 1] : 0, rows = g.ends[b] - begin;
      for (size_t i = 0; i < C; ++i) {
        const Column& pc = g.proto[i];
        ArrowArray& ca = blk->child_arrays[b * C + i];
        const void** cb = &blk->bufs[(b * C + i) * 3];
        memset(&ca, 0, sizeof ca);
        cb[0] = nullptr; cb[2] = nullptr;
        ca.null_count = 0;
        if (pc.type == T_UTF8) { cb[1] = vbase[i]; cb[2] = dbase[i]; ca.offset = begin; ca.n_buffers = 3; }   // a slice of the joined column
        else if (pc.type == T_BOOL || nbase[i]) { cb[1] = vbase[i]; ca.offset = begin; ca.n_buffers = 2; }    // (one Arrow offset serves values and validity)
        else { cb[1] = vbase[i] + begin * pc.width; ca.offset = 0; ca.n_buffers = 2; }
        if (nbase[i]) {   // nulls of the slice: unknown on the device (-1), counted for a host result; an all-valid slice drops the bitmap
          cb[0] = nbase[i];
          if (!host_out) ca.null_count = -1;
          else {
            int64_t nulls = 0;
            for (int64_t r = begin; r < begin + rows; ++r) nulls += !((nbase[i][r >> 3] >> (r & 7)) & 1);
            ca.null_count = nulls;
            if (nulls == 0) cb[0] = nullptr;
          }
        }
        ca.length = rows; ca.buffers = cb; ca.release = group_release_child_array; ca.private_data = blk;
        blk->child_ptrs[b * C + i] = &ca;
        ArrowSchema& cs = blk->child_schemas[b * C + i];
        memset(&cs, 0, sizeof cs);
        cs.format = blk->formats[i].c_str(); cs.name = blk->names[i].c_str(); cs.flags = pc.nullable ? ARROW_FLAG_NULLABLE : 0;
        cs.release = group_release_child_schema; cs.private_data = blk;
        blk->schild_ptrs[b * C + i] = &cs;
      }
      ArrowDeviceArray& o = outs[b];
      memset(&o, 0, sizeof o);
      blk->parent_bufs[b] = nullptr;
      o.array.length = rows; o.array.n_buffers = 1; o.array.buffers = &blk->parent_bufs[b];
      o.array.n_children = (int64_t)C; o.array.children = C ? &blk->child_ptrs[b * C] : nullptr;
      o.array.release = group_release_array; o.array.private_data = blk;
      o.device_id = device_type == ARROW_DEVICE_ROCM ? g.device_id : -1; o.device_type = device_type; o.sync_event = nullptr;
      ArrowSchema& os = out_schemas[b];
      memset(&os, 0, sizeof os);
      os.format = blk->struct_format.c_str(); os.name = blk->empty.c_str();
      os.n_children = (int64_t)C; os.children = C ? &blk->schild_ptrs[b * C] : nullptr;
      os.release = group_release_schema; os.private_data = blk;
    }
  });
}

// Import of a group.  Batch 0 is imported in full.  A DEVICE-resident group then only gathers its GroupLite -- the checks of
// import_batch on every record, but no Batch objects: the one-launch path works from the flat arrays, and building and
// freeing 12 500 Batch objects cost ~0.6 ms of a 2.5 ms call; `gi.materialise` imports them when another path needs them.
// Host groups (staged and packed from the batches themselves) are imported in full, with their GroupLite on the side.
void lite_from_arrow(const ArrowDeviceArray* rec, const ArrowSchema* schema, const Batch& first, int device, GroupLite& lite, size_t b) {
  const ArrowArray& a = rec->array;
  const size_t nc = lite.ncols;
  uint8_t f = 0;
  if (a.offset != 0 || a.n_children != schema->n_children || (size_t)a.n_children != nc || rec->device_type != ARROW_DEVICE_ROCM || rec->sync_event) {
    lite.flags[b] = GroupLite::GL_SCHEMA_DIFFERS;   // (anything unusual: the full import decides -- and reports)
    return;
  }
  const int64_t rows = a.length;
  lite.rows[b] = rows;
  if ((int)rec->device_id == device) f |= GroupLite::GL_ON_DEVICE;
  if (rows < 2) f |= GroupLite::GL_SHORT;
  for (size_t i = 0; i < nc; ++i) {
    const ArrowArray* ca = a.children[i];
    const Column& c0 = first.cols[i];
    if (!ca || ca->length < rows || ca->offset < 0 || ca->n_buffers < 2 || !ca->buffers || !ca->buffers[1] ||
        (ca->null_count > 0 && !ca->buffers[0])) { f |= GroupLite::GL_SCHEMA_DIFFERS; continue; }
    const uint8_t* validity = (const uint8_t*)ca->buffers[0];
    const uint8_t* values = (const uint8_t*)ca->buffers[1];
    if (validity && ca->null_count != 0) { f |= GroupLite::GL_NULLS; lite.validity[b * nc + i] = validity; }
    lite.offset[b * nc + i] = ca->offset;
    if (c0.type == T_UTF8) {
      const uint8_t* data = ca->n_buffers > 2 ? (const uint8_t*)ca->buffers[2] : nullptr;
      if (!data) f |= GroupLite::GL_NO_UTF8_DATA;
      lite.values0[b * nc + i] = values + 4 * ca->offset; lite.data[b * nc + i] = data;
    } else if (c0.type == T_BOOL) lite.values0[b * nc + i] = values;
    else lite.values0[b * nc + i] = values + (int64_t)c0.width * ca->offset;
  }
  lite.flags[b] = f;
}

void import_group(int n_records, const ArrowDeviceArray* const* recs, const ArrowSchema* schema, int device, std::vector<Batch>& in,
                  GroupLite& lite, GroupInput& gi) {
  in.resize(1);
  in[0] = import_batch(recs[0], schema);
  lite.resize((size_t)n_records, in[0].cols.size());
  lite.set(0, in[0], in[0], device);
  gi.batches = &in; gi.lite = &lite;
  auto import_rest = [n_records, recs, schema, &in]() {
    if ((int)in.size() == n_records) return;
    in.resize((size_t)n_records);
    if (n_records > 1) for_each_parallel(n_records - 1, [&](int k) { in[(size_t)k + 1] = import_batch(recs[(size_t)k + 1], schema); });
  };
  if (in[0].on_device && n_records > 1) {
    const Batch& first = in[0];
    for_each_parallel(n_records - 1, [&](int k) { lite_from_arrow(recs[(size_t)k + 1], schema, first, device, lite, (size_t)k + 1); });
    gi.materialise = import_rest;
    return;
  }
  import_rest();
  const Batch& first = in[0];
  if (n_records > 1) for_each_parallel(n_records - 1, [&](int k) { lite.set((size_t)k + 1, in[(size_t)k + 1], first, device); });
}

}  // namespace

extern "C" {

int chq_abi_version(void) { return CHQ_ABI_VERSION; }

const char* chq_status_name(chq_status s) {
  switch (s) {
    case CHQ_OK: return "Ok";
    case CHQ_ERR_VALUE_TYPE_NOT_IMPLEMENTED: return "ComputeValueError::ValueTypeNotImplemented";
    case CHQ_ERR_EXPRESSION_TYPE_NOT_IMPLEMENTED: return "ComputeValueError::ExpressionTypeNotImplemented";
    case CHQ_ERR_BINARY_OPERATOR_NOT_IMPLEMENTED: return "ComputeValueError::BinaryOperatorNotImplemented";
    case CHQ_ERR_BINARY_OPERATION_CAST_FAILED: return "ComputeValueError::BinaryOperatinCastFailed";
    case CHQ_ERR_FAILED_TO_PARSE_AS_AN_INTEGER: return "ComputeValueError::FailedToParseAsAnInteger";
    case CHQ_ERR_FAILED_TO_PARSE_AS_A_FLOAT: return "ComputeValueError::FailedToParseAsAFloat";
    case CHQ_ERR_COLUMN_NOT_FOUND: return "ComputeValueError::ColumnNotFound";
    case CHQ_ERR_IDENTIFIER_NOT_FOUND: return "ComputeValueError::IdentifierNotFound";
    case CHQ_ERR_UNSUPPORTED_TYPE_COERSION: return "ComputeValueError::UnsupportedTypeCoersionForOperationBetweenTypes";
    case CHQ_ERR_CAST_TO_BOOLEAN_ARRAY_FAILED: return "FilterRecordError::CastToBooleanArrayFailedForArrayType";
    case CHQ_ERR_PROJECT_NOT_IMPLEMENTED: return "ProjectRecordError::NotImplemented";
    case CHQ_ERR_ARROW_ARITHMETIC_OVERFLOW: return "ArrowError::ArithmeticOverflow";
    case CHQ_ERR_ARROW_DIVIDE_BY_ZERO: return "ArrowError::DivideByZero";
    case CHQ_ERR_ARROW_INVALID_ARGUMENT: return "ArrowError::InvalidArgumentError";
    case CHQ_ERR_ARROW_COMPUTE: return "ArrowError::ComputeError";
    case CHQ_ERR_ARROW_CAST: return "ArrowError::CastError";
    case CHQ_ERR_NOT_SUPPORTED: return "NotSupported";
    case CHQ_ERR_INVALID_HANDLE: return "InvalidHandle";
    case CHQ_ERR_DEVICE: return "DeviceError";
    case CHQ_ERR_OUT_OF_MEMORY: return "OutOfMemory";
  }
  return "Unknown";
}

chq_status chq_ctx_create(int device_id, void* hip_stream, chq_ctx** out) {
  if (!out) return CHQ_ERR_INVALID_HANDLE;
  *out = nullptr;
  chq_ctx* ctx = new (std::nothrow) chq_ctx();
  if (!ctx) return CHQ_ERR_OUT_OF_MEMORY;
  chq_status st = guarded(ctx, [&] {
    int n = 0;
    check_hip(hipGetDeviceCount(&n), "hipGetDeviceCount");
    if (n <= 0 || device_id < 0 || device_id >= n) throw ChqError{CHQ_ERR_DEVICE, "no usable GPU for device id " + std::to_string(device_id)};
    check_hip(hipSetDevice(device_id), "hipSetDevice");
    hipDeviceProp_t prop;
    check_hip(hipGetDeviceProperties(&prop, device_id), "hipGetDeviceProperties");
    ctx->c.device = device_id;
    ctx->c.num_cus = prop.multiProcessorCount;
    if (hip_stream == (void*)hipStreamLegacy) {
      // The legacy default stream has two spellings, hipStreamLegacy ((hipStream_t)1) and the null handle; the library
      // keeps the NULL one.  ROCm 7.2's hipEventRecord stores the handle it was given in the event (guarding its own use
      // with `handle <= 1`), and hipStreamWaitEvent on such an event checks that field only against NULL before
      // dereferencing it: an event recorded on hipStreamLegacy makes any later hipStreamWaitEvent read address 0x249
      // (DESIGN.md section 5.1).  Same stream, no special handle ever reaches an event.
      ctx->c.stream = nullptr; ctx->c.own_stream = false;
    } else if (hip_stream) { ctx->c.stream = (hipStream_t)hip_stream; ctx->c.own_stream = false; }
    else { check_hip(hipStreamCreateWithFlags(&ctx->c.stream, hipStreamNonBlocking), "hipStreamCreate"); ctx->c.own_stream = true; }
  });
  if (st != CHQ_OK) { delete ctx; return st; }
  *out = ctx;
  return CHQ_OK;
}

void chq_ctx_destroy(chq_ctx* ctx) { delete ctx; }
const char* chq_ctx_last_error(const chq_ctx* ctx) { return ctx ? ctx->c.last_error.c_str() : "null context"; }
void* chq_ctx_stream(const chq_ctx* ctx) { return ctx ? (void*)ctx->c.stream : nullptr; }
void chq_ctx_last_stats(const chq_ctx* ctx, chq_call_stats* out) { if (ctx && out) *out = ctx->c.stats; }

chq_status chq_ctx_set_option(chq_ctx* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return CHQ_ERR_INVALID_HANDLE;
  return guarded(ctx, [&] {
    std::string k(key);
    if (k == "tile_kind") { if (value < -1 || value > 2) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "tile_kind must be -1..2"}; ctx->c.opt_tile_kind = value; }
    else if (k == "enable_minus") ctx->c.opt_enable_minus = value != 0;
    else if (k == "time_kernels") ctx->c.opt_time_kernels = value != 0;
    else if (k == "fold_utf8") ctx->c.opt_fold_utf8 = value != 0;
    else if (k == "group_fold") ctx->c.opt_group_fold = value != 0;
    else if (k == "group_bits") ctx->c.opt_group_bits = value != 0;
    else if (k == "uniform_utf8_rows") ctx->c.opt_uniform_utf8_rows = value < 0 ? 0 : value;
    else if (k == "snappy_blocks") { if (value < 0 || value > 3) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "snappy_blocks is 0, 1, 2 or 3"}; ctx->c.opt_snappy_blocks = value; }
    else if (k == "parquet_page_rows") { if (value < 1) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "parquet_page_rows must be positive"}; ctx->c.opt_parquet_page_rows = value; }
    else if (k == "large_host") ctx->c.opt_large_host = value != 0;
    else if (k == "large_host_chunk") { if (value < 0) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "large_host_chunk must not be negative"}; ctx->c.opt_large_host_chunk = value; }
    else if (k == "large_host_rows") { if (value < 1) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "large_host_rows must be positive"}; ctx->c.opt_large_host_rows = value; }
    else if (k == "stash") { if (value < -1 || value > MAX_STASH) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "stash must be -1..2"}; ctx->c.opt_stash = value; }
    else if (k == "small_host") ctx->c.opt_small_host = value != 0;
    else if (k == "fuse") { if (value < 0 || value > 2) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "fuse must be 0..2"}; ctx->c.opt_fuse = value; }
    else if (k == "grid_per_cu") ctx->c.opt_grid_per_cu = value;
    else if (k == "split_rows") ctx->c.opt_split_rows = value;
    else if (k == "group_mode") ctx->c.opt_group_mode = value;
    else if (k == "group_chunk_bytes") { if (value < 1 || value > (1ll << 30)) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "group_chunk_bytes must be 1..2^30"}; ctx->c.opt_group_chunk_bytes = value; }
    else if (k == "trim_pool") { DevicePool::instance().trim(); HostPool::instance().trim(); }
    else if (k == "host_pool_bytes") HostPool::instance().set_limit((size_t)(value < 0 ? 0 : value));
    else throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "unknown option: " + k};
  });
}

// ---- expressions ------------------------------------------------------------------------------------
static chq_expr* new_expr(Expr::Kind k) { chq_expr* e = new (std::nothrow) chq_expr(); if (e) e->e.kind = k; return e; }
chq_expr* chq_expr_identifier(const char* name) { chq_expr* e = new_expr(Expr::IDENT); if (e) e->e.text = name ? name : ""; return e; }
chq_expr* chq_expr_compound_identifier(const char* const* parts, int n) {
  chq_expr* e = new_expr(Expr::COMPOUND);
  if (e) for (int i = 0; i < n; ++i) e->e.parts.push_back(parts[i] ? parts[i] : "");
  return e;
}
chq_expr* chq_expr_number(const char* text, int is_long) { chq_expr* e = new_expr(Expr::NUMBER); if (e) { e->e.text = text ? text : ""; e->e.flag = is_long != 0; } return e; }
chq_expr* chq_expr_boolean(int value) { chq_expr* e = new_expr(Expr::BOOLEAN); if (e) e->e.flag = value != 0; return e; }
chq_expr* chq_expr_single_quoted_string(const char* bytes, int64_t len) {
  chq_expr* e = new_expr(Expr::STRING);
  if (e && bytes && len > 0) e->e.text.assign(bytes, (size_t)len);
  return e;
}
chq_expr* chq_expr_unsupported_value(const char* debug) { chq_expr* e = new_expr(Expr::VALUE_OTHER); if (e) e->e.text = debug ? debug : ""; return e; }
chq_expr* chq_expr_binary_op(chq_expr* left, chq_binary_operator op, const char* op_debug, chq_expr* right) {
  chq_expr* e = new_expr(Expr::BINARY);
  if (!e || !left || !right) { chq_expr_free(left); chq_expr_free(right); delete e; return nullptr; }
  e->e.op = (int)op; e->e.text = op_debug ? op_debug : "";
  // the wrapper structs own nothing but the Expr: move the trees in and drop the shells
  e->e.l.reset(new Expr(std::move(left->e))); e->e.r.reset(new Expr(std::move(right->e)));
  delete left; delete right;
  return e;
}
chq_expr* chq_expr_nested(chq_expr* inner) {
  chq_expr* e = new_expr(Expr::NESTED);
  if (!e || !inner) { chq_expr_free(inner); delete e; return nullptr; }
  e->e.l.reset(new Expr(std::move(inner->e)));
  delete inner;
  return e;
}
chq_expr* chq_expr_unsupported(const char* debug) { chq_expr* e = new_expr(Expr::OTHER); if (e) e->e.text = debug ? debug : ""; return e; }
void chq_expr_free(chq_expr* e) { delete e; }

// ---- the path -------------------------------------------------------------------------------------------
chq_status chq_filter_record(chq_ctx* ctx, const ArrowDeviceArray* rec, const ArrowSchema* schema,
                             const chq_table_aliases* table_aliases, const chq_expr* expr, int out_device,
                             ArrowDeviceArray* out, ArrowSchema* out_schema) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(ctx, [&] {
    require(expr, "expression"); require(out, "output array"); require(out_schema, "output schema");
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    Batch in = import_batch(rec, schema);
    if (out_device == ARROW_DEVICE_CPU && !in.on_device) {   // the reference's calling pattern: small host batch in, host batch out
      Batch small;
      if (filter_record_small_host(ctx->c, in, table_aliases, expr->e, &small) ||
          filter_record_large_host(ctx->c, in, table_aliases, expr->e, &small)) {
        export_batch(std::move(small), ARROW_DEVICE_CPU, out, out_schema);
        return;
      }
    }
    Batch dev = to_device(ctx->c, in);
    auto pcols = plan_columns(dev, table_aliases);
    Batch res = filter_record(ctx->c, dev, pcols, expr->e);
    finish(ctx->c, std::move(res), out_device, out, out_schema);
  });
}

chq_status chq_plan_describe(const ArrowSchema* schema, const chq_table_aliases* table_aliases, const chq_expr* expr,
                             int64_t n_rows, int enable_minus, char* buf, size_t buf_len) {
  std::string text;
  chq_status st = CHQ_OK;
  try {
    if (!expr) throw ChqError{CHQ_ERR_INVALID_HANDLE, "null expression"};
    text = describe_plan(schema, table_aliases, expr->e, n_rows, enable_minus != 0);
  } catch (const ChqError& e) {
    st = e.code == CHQ_INTERNAL_PROGRAM_LIMIT ? CHQ_ERR_NOT_SUPPORTED : (chq_status)e.code; text = e.msg;
  } catch (const std::exception& e) {
    st = CHQ_ERR_DEVICE; text = e.what();
  }
  if (buf && buf_len) { const size_t n = std::min(buf_len - 1, text.size()); memcpy(buf, text.data(), n); buf[n] = 0; }
  return st;
}

chq_status chq_filter_records(chq_ctx* ctx, int n_records, const ArrowDeviceArray* const* recs, const ArrowSchema* schema,
                              const chq_table_aliases* table_aliases, const chq_expr* expr, int out_device,
                              ArrowDeviceArray* outs, ArrowSchema* out_schemas) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  for (int i = 0; i < n_records; ++i) mark_released(outs ? &outs[i] : nullptr, out_schemas ? &out_schemas[i] : nullptr);
  return guarded(ctx, [&] {
    require(expr, "expression");
    if (n_records < 0) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "negative record count"};
    if (n_records == 0) return;
    require(recs, "record array"); require(outs, "output arrays"); require(out_schemas, "output schemas");
    if (out_device != ARROW_DEVICE_ROCM && out_device != ARROW_DEVICE_CPU)
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "out_device must be ARROW_DEVICE_CPU or ARROW_DEVICE_ROCM"};
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    PhaseTimer pt("chq_filter_records");
    std::vector<Batch> in;
    GroupLite lite;
    GroupInput gi;
    for (int i = 0; i < n_records; ++i) require(recs[i], "record");
    import_group(n_records, recs, schema, ctx->c.device, in, lite, gi);
    pt.mark("import");
    GroupSliced sliced;
    std::vector<Batch> res = filter_records(ctx->c, gi, table_aliases, expr->e, out_device == ARROW_DEVICE_ROCM, &sliced);
    pt.mark("filter");
    if (sliced.filled) {
      std::vector<GroupSliced> more = std::move(sliced.more);
      size_t at = sliced.ends.size();
      export_group(std::move(sliced), out_device, outs, out_schemas);
      for (GroupSliced& m : more) { const size_t n = m.ends.size(); export_group(std::move(m), out_device, outs + at, out_schemas + at); at += n; }
      pt.mark("export");
      return;
    }
    try {
      for_each_parallel(n_records, [&](int i) { export_batch(std::move(res[(size_t)i]), out_device, &outs[i], &out_schemas[i]); });
    } catch (...) {   // no partial output
      for (int i = 0; i < n_records; ++i) {
        if (outs[i].array.release) outs[i].array.release(&outs[i].array);
        if (out_schemas[i].release) out_schemas[i].release(&out_schemas[i]);
      }
      throw;
    }
  });
}

chq_status chq_filter_records_coalesced(chq_ctx* ctx, int n_records, const ArrowDeviceArray* const* recs, const ArrowSchema* schema,
                                        const chq_table_aliases* table_aliases, const chq_expr* expr, int out_device,
                                        ArrowDeviceArray* out, ArrowSchema* out_schema, int64_t* rows_per_record) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(ctx, [&] {
    require(expr, "expression"); require(out, "output array"); require(out_schema, "output schema");
    if (n_records <= 0) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "at least one record batch is needed"};
    require(recs, "record array");
    if (out_device != ARROW_DEVICE_ROCM && out_device != ARROW_DEVICE_CPU)
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "out_device must be ARROW_DEVICE_CPU or ARROW_DEVICE_ROCM"};
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    PhaseTimer pt("chq_filter_records_coalesced");
    std::vector<Batch> in;
    GroupLite lite;
    GroupInput gi;
    for (int i = 0; i < n_records; ++i) require(recs[i], "record");
    import_group(n_records, recs, schema, ctx->c.device, in, lite, gi);
    pt.mark("import");
    std::vector<int64_t> rows;
    Batch res = filter_records_coalesced(ctx->c, gi, table_aliases, expr->e, out_device == ARROW_DEVICE_ROCM, &rows);
    pt.mark("filter");
    if (rows_per_record) for (int i = 0; i < n_records; ++i) rows_per_record[i] = rows[(size_t)i];
    export_batch(std::move(res), out_device, out, out_schema);
    pt.mark("export");
  });
}

chq_status chq_project_record(chq_ctx* ctx, const chq_select_item* fields, int n_fields, const ArrowDeviceArray* rec,
                              const ArrowSchema* schema, const chq_table_aliases* table_aliases, int out_device,
                              ArrowDeviceArray* out, ArrowSchema* out_schema) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(ctx, [&] {
    require(out, "output array"); require(out_schema, "output schema");
    if (n_fields > 0) require(fields, "select items");
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    Batch in = import_batch(rec, schema);
    std::vector<chq_select_item> items(fields, fields + (n_fields > 0 ? n_fields : 0));
    if (out_device == ARROW_DEVICE_CPU && !in.on_device) {   // the materialize task's calling pattern: host in, host out
      Batch res_host;
      if (project_record_host(ctx->c, items, in, table_aliases, &res_host)) {
        export_batch(std::move(res_host), ARROW_DEVICE_CPU, out, out_schema);
        return;
      }
    }
    Batch dev = to_device(ctx->c, in);
    auto pcols = plan_columns(dev, table_aliases);
    Batch res = project_record(ctx->c, items, dev, pcols);
    finish(ctx->c, std::move(res), out_device, out, out_schema);
  });
}

chq_status chq_compute_value(chq_ctx* ctx, const ArrowDeviceArray* rec, const ArrowSchema* schema,
                             const chq_table_aliases* table_aliases, const chq_expr* expr, int out_device,
                             ArrowDeviceArray* out, ArrowSchema* out_schema, int* out_is_scalar) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(ctx, [&] {
    require(expr, "expression"); require(out, "output array"); require(out_schema, "output schema");
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    Batch in = import_batch(rec, schema);
    Batch dev = to_device(ctx->c, in);
    auto pcols = plan_columns(dev, table_aliases);
    bool sc = false;
    Column col = compute_value(ctx->c, dev, pcols, expr->e, &sc);
    if (out_is_scalar) *out_is_scalar = sc ? 1 : 0;
    // reuse the batch path for the host copy
    Batch one; one.nrows = col.length; one.on_device = true; one.device_id = ctx->c.device;
    one.cols.push_back(std::move(col));
    if (out_device == ARROW_DEVICE_CPU) { Batch h = to_host(ctx->c, one); export_single_column(std::move(h.cols[0]), false, -1, out, out_schema); }
    else if (out_device == ARROW_DEVICE_ROCM) export_single_column(std::move(one.cols[0]), true, ctx->c.device, out, out_schema);
    else throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "out_device must be ARROW_DEVICE_CPU or ARROW_DEVICE_ROCM"};
  });
}

chq_status chq_filter_project_record(chq_ctx* ctx, const chq_expr* predicate, const chq_select_item* fields, int n_fields,
                                     const ArrowDeviceArray* rec, const ArrowSchema* schema,
                                     const chq_table_aliases* table_aliases, int out_device, ArrowDeviceArray* out,
                                     ArrowSchema* out_schema) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(ctx, [&] {
    require(predicate, "predicate"); require(out, "output array"); require(out_schema, "output schema");
    if (n_fields > 0) require(fields, "select items");
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    Batch in = import_batch(rec, schema);
    Batch dev = to_device(ctx->c, in);
    auto pcols = plan_columns(dev, table_aliases);
    std::vector<chq_select_item> fused_items(fields, fields + (n_fields > 0 ? n_fields : 0));
    Batch fused;
    if (filter_project_fused(ctx->c, dev, pcols, predicate->e, fused_items, &fused)) {
      finish(ctx->c, std::move(fused), out_device, out, out_schema);
      return;
    }
    Batch filtered = filter_record(ctx->c, dev, pcols, predicate->e);   // stays in HBM
    chq_call_stats fs = ctx->c.stats;
    auto pcols2 = plan_columns(filtered, table_aliases);
    std::vector<chq_select_item> items(fields, fields + (n_fields > 0 ? n_fields : 0));
    Batch res = project_record(ctx->c, items, filtered, pcols2);
    fs.launches += ctx->c.stats.launches;
    ctx->c.stats = fs;
    finish(ctx->c, std::move(res), out_device, out, out_schema);
  });
}

chq_status chq_record_to_device(chq_ctx* ctx, const ArrowDeviceArray* rec, const ArrowSchema* schema, ArrowDeviceArray* out,
                                ArrowSchema* out_schema) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(ctx, [&] {
    require(out, "output array"); require(out_schema, "output schema");
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    Batch in = import_batch(rec, schema);
    if (in.on_device) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "record is already device resident"};
    Batch dev = to_device(ctx->c, in);
    check_hip(hipStreamSynchronize(ctx->c.stream), "hipStreamSynchronize");
    export_batch(std::move(dev), ARROW_DEVICE_ROCM, out, out_schema);
  });
}

chq_status chq_record_copy_to_peer(chq_ctx* src_ctx, chq_ctx* dst_ctx, const ArrowDeviceArray* rec, const ArrowSchema* schema,
                                   ArrowDeviceArray* out, ArrowSchema* out_schema) {
  if (!src_ctx || !dst_ctx) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(dst_ctx, [&] {
    require(out, "output array"); require(out_schema, "output schema");
    Batch in = import_batch(rec, schema);   // waits on rec->sync_event when the producer left one
    hipEvent_t ev = nullptr;
    Batch moved = copy_to_peer(src_ctx->c, dst_ctx->c, in, &ev);
    export_batch(std::move(moved), ARROW_DEVICE_ROCM, out, out_schema, ev);
  });
}

namespace {
struct IpcHolder { IpcMessage msg; };
void release_ipc(chq_ipc_message* m) {
  if (!m || !m->release) return;
  delete (IpcHolder*)m->private_data;
  memset(m, 0, sizeof(*m));
}
}  // namespace

chq_status chq_record_to_ipc(chq_ctx* ctx, const ArrowDeviceArray* rec, const ArrowSchema* schema, int body_device,
                             chq_ipc_message* out) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  if (out) memset(out, 0, sizeof(*out));
  return guarded(ctx, [&] {
    require(out, "output message");
    if (body_device != ARROW_DEVICE_ROCM && body_device != ARROW_DEVICE_CPU)
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "body_device must be ARROW_DEVICE_CPU or ARROW_DEVICE_ROCM"};
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    Batch in = import_batch(rec, schema);
    Batch dev = to_device(ctx->c, in);   // (a host batch is staged: the body is assembled by the device either way)
    auto* h = new IpcHolder();
    try {
      h->msg = record_to_ipc(ctx->c, dev, body_device == ARROW_DEVICE_ROCM);
    } catch (...) { delete h; throw; }
    out->header = h->msg.header.data(); out->header_len = (int64_t)h->msg.header.size();
    out->body = h->msg.body->ptr; out->body_len = h->msg.body_len;
    out->body_device_type = h->msg.body_on_device ? ARROW_DEVICE_ROCM : ARROW_DEVICE_CPU;
    out->body_device_id = h->msg.body_on_device ? ctx->c.device : -1;
    const uint8_t eos[8] = {0xff, 0xff, 0xff, 0xff, 0, 0, 0, 0};
    memcpy(out->end_of_stream, eos, 8);
    out->release = release_ipc; out->private_data = h;
  });
}

chq_status chq_record_from_ipc(chq_ctx* ctx, const uint8_t* stream, int64_t stream_len, const void* body, int64_t body_len,
                               int body_device_type, int out_device, ArrowDeviceArray* out, ArrowSchema* out_schema) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(ctx, [&] {
    require(out, "output array"); require(out_schema, "output schema"); require(stream, "stream");
    if (out_device != ARROW_DEVICE_ROCM && out_device != ARROW_DEVICE_CPU)
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "out_device must be ARROW_DEVICE_CPU or ARROW_DEVICE_ROCM"};
    if (body && body_device_type != ARROW_DEVICE_ROCM && body_device_type != ARROW_DEVICE_CPU)
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "body_device_type must be ARROW_DEVICE_CPU or ARROW_DEVICE_ROCM"};
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    Batch res = record_from_ipc(ctx->c, stream, stream_len, body, body_len, body_device_type == ARROW_DEVICE_ROCM,
                                out_device == ARROW_DEVICE_ROCM);
    export_batch(std::move(res), out_device, out, out_schema);
  });
}

chq_status chq_ipc_describe(const uint8_t* stream, int64_t stream_len, char* buf, size_t buf_len) {
  std::string text;
  chq_status st = CHQ_OK;
  try {
    text = describe_ipc(stream, stream_len);
  } catch (const ChqError& e) {
    text = e.msg; st = (chq_status)e.code;
  } catch (const std::exception& e) {
    text = e.what(); st = CHQ_ERR_ARROW_INVALID_ARGUMENT;
  }
  if (buf && buf_len) { const size_t k = std::min(buf_len - 1, text.size()); memcpy(buf, text.data(), k); buf[k] = 0; }
  return st;
}

// ---- Parquet scan -------------------------------------------------------------------------------------------------------
struct chq_parquet { chq::PqFile file; };

chq_status chq_parquet_open(const uint8_t* file, int64_t file_len, chq_parquet** out, char* err, size_t err_len) {
  if (!out) return CHQ_ERR_ARROW_INVALID_ARGUMENT;
  *out = nullptr;
  std::string text;
  chq_status st = CHQ_OK;
  try {
    auto* h = new chq_parquet();
    try { h->file = parquet_open(file, file_len); } catch (...) { delete h; throw; }
    *out = h;
  } catch (const ChqError& e) {
    text = e.msg; st = (chq_status)e.code;
  } catch (const std::exception& e) {
    text = e.what(); st = CHQ_ERR_ARROW_INVALID_ARGUMENT;
  }
  if (err && err_len) { const size_t k = std::min(err_len - 1, text.size()); memcpy(err, text.data(), k); err[k] = 0; }
  return st;
}
chq_status chq_parquet_open_reader(int64_t file_len, chq_read_range_fn read, void* user, chq_parquet** out, char* err, size_t err_len) {
  if (!out) return CHQ_ERR_ARROW_INVALID_ARGUMENT;
  *out = nullptr;
  std::string text;
  chq_status st = CHQ_OK;
  try {
    if (!read) throw ChqError{CHQ_ERR_INVALID_HANDLE, "null range reader"};
    auto* h = new chq_parquet();
    try { h->file = parquet_open_reader(file_len, read, user); } catch (...) { delete h; throw; }
    *out = h;
  } catch (const ChqError& e) {
    text = e.msg; st = (chq_status)e.code;
  } catch (const std::exception& e) {
    text = e.what(); st = CHQ_ERR_ARROW_INVALID_ARGUMENT;
  }
  if (err && err_len) { const size_t k = std::min(err_len - 1, text.size()); memcpy(err, text.data(), k); err[k] = 0; }
  return st;
}
void chq_parquet_close(chq_parquet* pq) { delete pq; }
int32_t chq_parquet_num_columns(const chq_parquet* pq) { return pq ? (int32_t)pq->file.columns.size() : 0; }
const char* chq_parquet_column_name(const chq_parquet* pq, int32_t column) {
  return pq && column >= 0 && column < (int32_t)pq->file.columns.size() ? pq->file.columns[(size_t)column].name.c_str() : nullptr;
}
int32_t chq_parquet_num_row_groups(const chq_parquet* pq) { return pq ? (int32_t)pq->file.row_groups.size() : 0; }
int64_t chq_parquet_row_group_num_rows(const chq_parquet* pq, int32_t row_group) {
  return pq && row_group >= 0 && row_group < (int32_t)pq->file.row_groups.size() ? pq->file.row_groups[row_group].num_rows : -1;
}
chq_status chq_parquet_describe(const chq_parquet* pq, char* buf, size_t buf_len) {
  if (!pq) return CHQ_ERR_INVALID_HANDLE;
  const std::string text = parquet_describe(pq->file);
  if (buf && buf_len) { const size_t k = std::min(buf_len - 1, text.size()); memcpy(buf, text.data(), k); buf[k] = 0; }
  return text.size() < buf_len ? CHQ_OK : CHQ_ERR_ARROW_INVALID_ARGUMENT;
}
chq_status chq_parquet_read_row_group(chq_ctx* ctx, const chq_parquet* pq, int32_t row_group, int out_device,
                                      ArrowDeviceArray* out, ArrowSchema* out_schema) {
  if (!ctx || !pq) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(ctx, [&] {
    require(out, "output array"); require(out_schema, "output schema");
    if (out_device != ARROW_DEVICE_ROCM && out_device != ARROW_DEVICE_CPU)
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "out_device must be ARROW_DEVICE_CPU or ARROW_DEVICE_ROCM"};
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    Batch res = parquet_read_row_group(ctx->c, pq->file, row_group);
    if (out_device == ARROW_DEVICE_CPU) res = to_host(ctx->c, res);
    export_batch(std::move(res), out_device, out, out_schema);
  });
}

chq_status chq_parquet_read_row_groups(chq_ctx* ctx, const chq_parquet* pq, int32_t first, int32_t count, int out_device,
                                       ArrowDeviceArray* outs, ArrowSchema* out_schemas) {
  return chq_parquet_read_columns(ctx, pq, first, count, nullptr, -1, out_device, outs, out_schemas);
}

chq_status chq_parquet_read_columns(chq_ctx* ctx, const chq_parquet* pq, int32_t first, int32_t count, const int32_t* columns,
                                    int32_t n_columns, int out_device, ArrowDeviceArray* outs, ArrowSchema* out_schemas) {
  if (!ctx || !pq) return CHQ_ERR_INVALID_HANDLE;
  if (outs && out_schemas) for (int32_t i = 0; i < count; ++i) mark_released(&outs[i], &out_schemas[i]);
  return guarded(ctx, [&] {
    if (count > 0) { require(outs, "output arrays"); require(out_schemas, "output schemas"); }
    if (out_device != ARROW_DEVICE_ROCM && out_device != ARROW_DEVICE_CPU)
      throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "out_device must be ARROW_DEVICE_CPU or ARROW_DEVICE_ROCM"};
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    if (!columns) n_columns = -1;
    std::vector<Batch> res = parquet_read_row_groups(ctx->c, pq->file, first, count, columns, n_columns);
    const chq_call_stats scan_stats = ctx->c.stats;
    try {
      for (size_t i = 0; i < res.size(); ++i) {
        if (out_device == ARROW_DEVICE_CPU) res[i] = to_host(ctx->c, res[i]);
        export_batch(std::move(res[i]), out_device, &outs[i], &out_schemas[i]);
      }
    } catch (...) {   // no partial output
      for (int32_t i = 0; i < count; ++i) {
        if (outs[i].array.release) outs[i].array.release(&outs[i].array);
        if (out_schemas[i].release) out_schemas[i].release(&out_schemas[i]);
      }
      throw;
    }
    ctx->c.stats = scan_stats;   // (to_host does not touch them today; the scan's counters are what the call reports)
  });
}

namespace {
struct ParquetImageHolder { chq::ParquetImage img; };
void release_parquet_image(chq_parquet_image* m) {
  if (!m || !m->release) return;
  delete (ParquetImageHolder*)m->private_data;
  m->data = nullptr; m->len = 0; m->private_data = nullptr; m->release = nullptr;
}
}  // namespace

chq_status chq_record_to_parquet(chq_ctx* ctx, const ArrowDeviceArray* rec, const ArrowSchema* schema, chq_parquet_image* out) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  if (out) { out->data = nullptr; out->len = 0; out->release = nullptr; out->private_data = nullptr; }
  return guarded(ctx, [&] {
    require(out, "output image");
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    Batch in = import_batch(rec, schema);
    auto* h = new ParquetImageHolder();
    try { h->img = record_to_parquet(ctx->c, in); } catch (...) { delete h; throw; }
    out->data = (const uint8_t*)h->img.bytes->ptr; out->len = h->img.len;
    out->release = release_parquet_image; out->private_data = h;
  });
}

chq_status chq_records_to_parquet(chq_ctx* ctx, int n_records, const ArrowDeviceArray* const* recs, const ArrowSchema* schema,
                                  chq_parquet_image* out) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  if (out) { out->data = nullptr; out->len = 0; out->release = nullptr; out->private_data = nullptr; }
  return guarded(ctx, [&] {
    require(out, "output image"); require(recs, "record batches");
    if (n_records < 1) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "chq_records_to_parquet needs at least one record batch"};
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    std::vector<Batch> in((size_t)n_records);
    std::vector<const Batch*> ptrs((size_t)n_records);
    for (int i = 0; i < n_records; ++i) { in[(size_t)i] = import_batch(recs[i], schema); ptrs[(size_t)i] = &in[(size_t)i]; }
    auto* h = new ParquetImageHolder();
    try { h->img = records_to_parquet(ctx->c, ptrs); } catch (...) { delete h; throw; }
    out->data = (const uint8_t*)h->img.bytes->ptr; out->len = h->img.len;
    out->release = release_parquet_image; out->private_data = h;
  });
}

chq_status chq_record_to_host(chq_ctx* ctx, const ArrowDeviceArray* rec, const ArrowSchema* schema, ArrowDeviceArray* out,
                              ArrowSchema* out_schema) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(ctx, [&] {
    require(out, "output array"); require(out_schema, "output schema");
    check_hip(hipSetDevice(ctx->c.device), "hipSetDevice");
    Batch in = import_batch(rec, schema);
    if (!in.on_device) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "record is already host resident"};
    for (Column& c : in.cols) if (c.validity && c.null_count < 0) c.null_count = 1;
    Batch h = to_host(ctx->c, in);
    // resolve unknown null counts on the host copy
    for (Column& c : h.cols) {
      if (c.validity) {
        int64_t nulls = 0;
        for (int64_t i = 0; i < c.length; ++i) { int64_t b = c.offset + i; nulls += !((c.validity[b >> 3] >> (b & 7)) & 1); }
        c.null_count = nulls;
      }
    }
    export_batch(std::move(h), ARROW_DEVICE_CPU, out, out_schema);
  });
}

chq_status chq_wrap_columns(chq_ctx* ctx, const chq_column_desc* cols, int n_cols, int64_t n_rows, int device_type,
                            ArrowDeviceArray* out, ArrowSchema* out_schema) {
  if (!ctx) return CHQ_ERR_INVALID_HANDLE;
  mark_released(out, out_schema);
  return guarded(ctx, [&] {
    require(out, "output array"); require(out_schema, "output schema");
    if (n_cols > 0) require(cols, "column descriptors");
    Batch b;
    b.nrows = n_rows; b.on_device = device_type == ARROW_DEVICE_ROCM; b.device_id = ctx->c.device;
    for (int i = 0; i < n_cols; ++i) {
      Column c;
      c.name = cols[i].name ? cols[i].name : ""; c.format = cols[i].format ? cols[i].format : "";
      // validate the format through the importer's table
      ArrowSchema tmp{}; tmp.format = c.format.c_str();
      {
        ArrowSchema parent{}; ArrowSchema* kids[1] = {&tmp}; parent.format = "+s"; parent.n_children = 1; parent.children = kids;
        ArrowArray ka{}; ArrowArray* akids[1] = {&ka}; ArrowDeviceArray da{}; da.array.n_children = 1; da.array.children = akids;
        da.device_type = ARROW_DEVICE_CPU;
        Batch probe = import_batch(&da, &parent);
        c.type = probe.cols[0].type; c.width = probe.cols[0].width;
      }
      if (n_rows > 0 && !cols[i].values) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "column '" + c.name + "' has no values buffer"};
      if (cols[i].null_count > 0 && !cols[i].validity) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "column '" + c.name + "' reports nulls but has no validity bitmap"};
      if (n_rows < 0 || cols[i].offset < 0) throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "negative length or offset"};
      c.nullable = cols[i].nullable != 0; c.length = n_rows; c.null_count = cols[i].validity ? cols[i].null_count : 0; c.offset = cols[i].offset;
      c.validity = (const uint8_t*)cols[i].validity; c.values = (const uint8_t*)cols[i].values; c.data = (const uint8_t*)cols[i].data;
      b.cols.push_back(std::move(c));
    }
    export_batch(std::move(b), device_type, out, out_schema);
  });
}

}  // extern "C"
