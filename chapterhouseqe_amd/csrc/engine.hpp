// engine.hpp -- context, device memory, Arrow C Data import/export and the record-level operations.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <mutex>
#include <chrono>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/chq.h"
#include "device_program.h"
#include "typed_ops.h"
#include "plan.hpp"

namespace chq {

// kernels.hip
hipError_t launch_filter(const FilterParams& p, int tile_kind, bool partial, int grid, hipStream_t stream);
hipError_t launch_filter_project(const FusedParams& p, int tile_kind, int grid, hipStream_t stream);
hipError_t launch_project(const ProjectParams& p, int tile_kind, bool partial, int grid, hipStream_t stream);
hipError_t launch_bit_compact(const BitCompactParams& p, int grid, hipStream_t stream);
hipError_t launch_bit_compact_group(const BitCompactGroupParams& p, int grid, hipStream_t stream);
hipError_t launch_cmp128(const Cmp128Params& p, hipStream_t stream);
hipError_t launch_utf8_to_bool(const Utf8ToBoolParams& p, hipStream_t stream);
hipError_t launch_utf8_uniform(const Utf8UniformParams& p, hipStream_t stream);
hipError_t launch_iota_offsets(const IotaOffsetsParams& p, hipStream_t stream);
hipError_t launch_utf8_uniform_group(const Utf8UniformGroupParams& p, int64_t max_rows, hipStream_t stream);
hipError_t launch_utf8_filter(const Utf8Params& p, int grid, hipStream_t stream);
hipError_t launch_gather_i32(const GatherParams& p, hipStream_t stream);
hipError_t launch_gather_status(const GatherStatusParams& p, hipStream_t stream);
hipError_t launch_split_bounds(const SplitBoundsParams& p, hipStream_t stream);
hipError_t launch_utf8_offsets(const Utf8Params& p, int grid, hipStream_t stream);
hipError_t launch_utf8_copy(const Utf8Params& p, int grid, hipStream_t stream);
hipError_t launch_concat(const ConcatParams& p, int kind, int grid, hipStream_t stream);   // kind: 0 fixed, 1 ends, 2 utf8, 3 bits

// ---- memory ---------------------------------------------------------------------------------------
// Process-wide caching allocator for HBM (outlives contexts: Arrow release callbacks may run after the
// context that produced a batch is gone).  Blocks are recycled by size class; nothing is returned to
// HIP until trim().
class DevicePool {
 public:
  static DevicePool& instance();
  void* alloc(size_t bytes, int device, size_t* cap);
  void free(void* p, size_t cap, int device);
  void trim();
 private:
  std::mutex mu_;
  std::unordered_map<uint64_t, std::vector<void*>> free_;   // (size class, device) -> cached blocks
};

// The same for host memory handed out as result buffers.  A fresh host allocation costs a page fault per 4 KiB the
// first time the D2H copy touches it (measured: 8.8 GB/s instead of 55 GB/s); recycled blocks are already mapped.
// Only blocks >= 1 MiB are cached, up to `limit` bytes in total; the rest goes straight back to the C allocator.
class HostPool {
 public:
  static HostPool& instance();
  void* alloc(size_t bytes, size_t* cap);
  void free(void* p, size_t cap);
  void trim();
  void set_limit(size_t bytes);
 private:
  std::mutex mu_;
  std::unordered_map<size_t, std::vector<void*>> free_;
  size_t cached_ = 0;
  size_t limit_ = (size_t)8 << 30;
};

// One allocation (device or host) owned by an exported batch.
struct Buffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  size_t cap = 0;          // size class the pool handed out
  int device_id = 0;
  bool device = false;
  ~Buffer();
  Buffer() = default;
  Buffer(const Buffer&) = delete;
  Buffer& operator=(const Buffer&) = delete;
};
using BufferPtr = std::shared_ptr<Buffer>;
BufferPtr make_device_buffer(size_t bytes, int device);
BufferPtr make_host_buffer(size_t bytes);

// ---- batches ----------------------------------------------------------------------------------------
// A column as stored (device or host pointers), Arrow layout.
struct Column {
  std::string name;
  std::string format;      // Arrow C format string
  DType type = T_FIXED_OPAQUE;
  int width = 0;           // bytes per value for fixed-width types
  bool nullable = false;   // schema flag
  int64_t length = 0;
  int64_t null_count = 0;
  int64_t offset = 0;      // Arrow slice offset (elements)
  const uint8_t* validity = nullptr;
  const uint8_t* values = nullptr;   // fixed: values buffer; bool: bitmap; utf8: int32 offsets buffer
  const uint8_t* data = nullptr;     // utf8 bytes
  int64_t data_bytes = -1;           // utf8: bytes the offsets of rows [offset, offset+length] span, when whoever built the column knew (else -1)
  std::vector<BufferPtr> owned;      // keeps staged / produced buffers alive

  // derived views (element 0 = first logical row)
  const void* values0() const;       // fixed: values + offset*width; utf8: offsets + offset
  int64_t bool_bit_offset() const { return offset; }
  int64_t validity_bit_offset() const { return offset; }
};

struct Batch {
  int64_t nrows = 0;
  bool on_device = false;
  int device_id = 0;
  std::vector<Column> cols;
};

// ---- context ------------------------------------------------------------------------------------------
struct Context {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cus = 0;
  std::string last_error;
  chq_call_stats stats{};
  // options
  int64_t opt_tile_kind = -1;       // -1 auto
  bool opt_enable_minus = false;
  bool opt_time_kernels = false;
  int64_t opt_large_host = 1;       // host batches of opt_large_host_rows rows or more: chunked, uploads and downloads overlapped
  int64_t opt_large_host_rows = 8 << 20;
  int64_t opt_large_host_chunk = 0;  // rows per chunk (0: about 64 MB of input)
  int64_t opt_group_bits = 1;       // wave-packed device groups with validity bitmaps / Boolean columns take the one-launch path (0: joined first)
  int64_t opt_uniform_utf8_rows = 1 << 24;   // batches from this size on (below it the extra pass and its read-back cost more than the copy saves): Utf8 columns whose values all have one length are filtered as fixed-width columns (0: never)
  int64_t opt_parquet_page_rows = 65536;   // chq_record_to_parquet: rows per data page (multiples of 4096; at most 64 pages per chunk)
  int64_t opt_snappy_blocks = 1;           // Parquet scan: snappy pages of 3+ blocks of 64 KiB are inflated block by block (0: one wave per page; 2: the blocks give up -- tests)
  int64_t opt_group_fold = 1;       // device-resident groups with short-string Utf8 columns: filtered straight out of the batches (0: joined first)
  int64_t opt_fold_utf8 = 1;        // short-string Utf8 columns are filtered inside filter_fused_kernel (0: always the separate Utf8 pass)
  int64_t opt_stash = -1;           // predicate input columns kept in LDS between the filter kernel's phases: -1 = as many as the tile kind has slots
  int64_t opt_fuse = 1;             // chq_filter_project_record's single-pass kernel: 0 never, 1 when it moves clearly fewer bytes, 2 whenever possible
  int64_t opt_grid_per_cu = 0;
  int64_t opt_split_rows = 1 << 20;   // batches at least this long run their complete tiles in the FULL-only kernels
  int64_t opt_group_chunk_bytes = 1ll << 30;   // host-concatenated groups: Utf8 bytes per column and chunk (int32 offsets)
  int64_t opt_group_mode = 0;       // batch-group launch: 0 auto, 1 force per-tile table, 2 force wave-granular packing
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // Parquet scan: the columns of a row group decode side by side (their per-page kernels are latency-bound, one wave per
  // page); created on first use, joined back into `stream` before anything is read back
  static constexpr int kAuxStreams = 13;    // twelve for kernels + kCopyStream for uploads
  // Uploads of the Parquet scan go through a stream of their own.  A copy from pageable host memory keeps the calling
  // thread until the copy has RUN, and on a column's kernel stream it runs behind every kernel queued there: with a slow
  // page decode on that stream (a 70 ms inflate) the host sat in the next upload instead of issuing the other columns' work,
  // and the columns decoded one after the other.  On the copy stream an upload waits for earlier uploads only; the column's
  // stream waits for it through an event.
  static constexpr int kCopyStream = 12;
  static constexpr int kKernelStreams = 12;
  std::vector<hipEvent_t> upload_events;    // pool: one per column in flight (parquet_scan.cpp)
  hipStream_t aux[kAuxStreams] = {};
  hipEvent_t aux_fork = nullptr, aux_join[kAuxStreams] = {};
  bool aux_ready = false;           // every auxiliary stream and event exists (ensure_aux_streams)
  // reusable device scratch
  BufferPtr small;          // [Scratch header (512 B)] [status words of the chained scan]: cleared by ONE memset per call
  size_t small_tiles = 0;   // status words the block has room for
  void* pinned = nullptr;                  // pinned host staging for small read-backs (256 B)
  size_t pinned_bytes = 0;
  void* pinned_tbl = nullptr;              // pinned staging of a batch-group launch (tile table, per-batch prefixes)
  size_t pinned_tbl_bytes = 0;
  void* pinned_sizes = nullptr;            // pinned staging of the Utf8 size gather of a group (in flight while the table is built)
  size_t pinned_sizes_bytes = 0;
  void* pinned_io = nullptr;               // pinned staging of the small host-batch path: [inputs | outputs]
  size_t pinned_io_bytes = 0;
  BufferPtr dev_io;                        // its device twin
  bool opt_small_host = true;              // host batches up to a few MB: one upload, one download, one synchronisation

  ~Context();
};

void check_hip(hipError_t e, const char* what);
// the auxiliary streams of a context: created on first use (one site), forked off / joined back into ctx.stream with events
void ensure_aux_streams(Context& ctx);
void fork_aux_streams(Context& ctx);
void join_aux_streams(Context& ctx);

// Arrow C Data Interface <-> Batch
Batch import_batch(const ArrowDeviceArray* rec, const ArrowSchema* schema);
// `sync_event`: optional event the consumer must wait on; the exported array owns it (destroyed on release)
void export_batch(Batch&& b, int device_type, ArrowDeviceArray* out, ArrowSchema* out_schema, hipEvent_t sync_event = nullptr);
void export_single_column(Column&& c, bool on_device, int device_id, ArrowDeviceArray* out, ArrowSchema* out_schema);

Batch to_device(Context& ctx, const Batch& b);   // stage host batch into HBM (no-op view when already there)
Batch to_host(Context& ctx, const Batch& b);
// device batch on src's GPU -> fresh buffers on dst's GPU (hipMemcpyPeerAsync on dst's stream); *event_out receives an
// event recorded behind the last copy (the caller owns it)
Batch copy_to_peer(Context& src, Context& dst, const Batch& b, hipEvent_t* event_out);

// the record-level operations (throw ChqError)
// `split`: optional input-row positions (ascending); on return bounds[i] = selected rows before starts[i], i.e. where the
// output of a batch that was concatenated into `rec_dev` at row starts[i] begins
struct SplitRequest { std::vector<int64_t> starts; std::vector<int64_t> bounds; };
Batch filter_record(Context& ctx, const Batch& rec_dev, const std::vector<PlanColumn>& pcols, const Expr& expr,
                    SplitRequest* split = nullptr);
// A small host batch, host result: columns staged through one pinned block each way (one H2D, one D2H, one stream
// synchronisation).  False = outside its scope or an error was flagged: take the general path.
bool filter_record_small_host(Context& ctx, const Batch& rec_host, const chq_table_aliases* aliases, const Expr& expr, Batch* result);
bool filter_record_large_host(Context& ctx, const Batch& rec_host, const chq_table_aliases* aliases, const Expr& expr, Batch* result);
// Per-batch facts of a group in flat arrays, gathered while the batches are imported (capi.cpp: the Arrow structs of 10^4
// batches are ~10^5 dependent cache misses, taken once, on the pool's threads).  Later passes of a group call -- eligibility,
// row totals, pointer tables -- read these arrays instead of walking the Batch objects again.
struct GroupLite {
  size_t ncols = 0;
  std::vector<int64_t> rows;               // [nb]
  std::vector<const uint8_t*> values0;     // [nb * ncols] Column::values0() (Boolean: the bitmap)
  std::vector<const uint8_t*> data;        // [nb * ncols] Utf8 bytes
  std::vector<const uint8_t*> validity;    // [nb * ncols] validity bitmap, null when the column has no nulls in that batch
  std::vector<int64_t> offset;             // [nb * ncols] Arrow slice offset: bit position of row 0 in the bitmaps
  std::vector<uint8_t> flags;              // [nb] GL_*
  enum : uint8_t { GL_NULLS = 1, GL_NO_UTF8_DATA = 2, GL_SCHEMA_DIFFERS = 4, GL_ON_DEVICE = 8, GL_SHORT = 16 };
  void resize(size_t nb, size_t nc) { ncols = nc; rows.assign(nb, 0); values0.assign(nb * nc, nullptr); data.assign(nb * nc, nullptr); validity.assign(nb * nc, nullptr); offset.assign(nb * nc, 0); flags.assign(nb, 0); }
  void set(size_t b, const Batch& r, const Batch& first, int device);
};
// The one-launch result of a group call before it is cut into per-batch Arrow structs: dense output buffers every batch's
// output is a slice of (capi.cpp exports them from one block instead of ~20 allocations per batch).
struct GroupSliced {
  bool filled = false, on_device = false;
  int device_id = -1;
  std::vector<Column> proto;               // per column: name / format / type / width / nullable flag
  std::vector<BufferPtr> values, data, validity;   // per column: values (Utf8: joined offsets; Boolean: joined bitmap), Utf8 bytes, joined validity bitmap or null
  std::vector<int64_t> ends;               // [nb] exclusive end row of every batch in the dense output
  // a group too large for ONE launch (int32 offsets of the joined Utf8 output, 2^31 rows) is run as consecutive sub-groups:
  // this struct then describes the first `ends.size()` batches and `more` the following ones, in order
  std::vector<GroupSliced> more;
};
// The batches of a group call.  `batches` always holds batch 0; with `lite` the others may be missing until `materialise`
// (imports every batch; idempotent) has run -- the one-launch path of a device-resident group works from `lite` alone, and
// 10^4 Batch objects it never looks at cost more to build and free than the rest of the call's host work.
struct GroupInput {
  std::vector<Batch>* batches = nullptr;
  const GroupLite* lite = nullptr;
  std::function<void()> materialise;
};
// one launch for a group of same-schema batches (host or device resident); outputs where `out_on_device` says.
// `sliced` (optional): filled instead of the returned vector when the one-launch path ran
std::vector<Batch> filter_records(Context& ctx, const GroupInput& in, const chq_table_aliases* aliases,
                                  const Expr& expr, bool out_on_device, GroupSliced* sliced = nullptr);
// the same, but ONE output batch holding every surviving row in input order (+ surviving rows per input batch)
Batch filter_records_coalesced(Context& ctx, const GroupInput& in, const chq_table_aliases* aliases,
                               const Expr& expr, bool out_on_device, std::vector<int64_t>* rows_per_record);
// (convenience for callers that hold complete batches)
inline GroupInput group_of(std::vector<Batch>& batches) { GroupInput g; g.batches = &batches; return g; }
Batch project_record(Context& ctx, const std::vector<chq_select_item>& fields, const Batch& rec_dev,
                     const std::vector<PlanColumn>& pcols);
// filter_record + project_record in one kernel pass; false = outside its scope (or an error was flagged): run the two steps
bool filter_project_fused(Context& ctx, const Batch& rec_dev, const std::vector<PlanColumn>& pcols, const Expr& pred,
                          const std::vector<chq_select_item>& fields, Batch* result);
// project_record for a host batch with a host result: pass-through columns never cross PCIe; false = general path
bool project_record_host(Context& ctx, const std::vector<chq_select_item>& fields, const Batch& rec_host,
                         const chq_table_aliases* aliases, Batch* result);
Column compute_value(Context& ctx, const Batch& rec_dev, const std::vector<PlanColumn>& pcols, const Expr& expr,
                     bool* is_scalar);

std::vector<PlanColumn> plan_columns(const Batch& b, const chq_table_aliases* aliases);

// Arrow C format string -> column kind and byte width (throws CHQ_ERR_NOT_SUPPORTED outside the build's scope)
void parse_arrow_format(const char* format, DType* type, int* width);

// A few persistent host threads for per-batch bookkeeping of large groups (importing / exporting 10^4..10^5 Arrow structs,
// building pointer tables): f(t) for t in [0, tasks), the caller takes part, the first exception is rethrown.  Creating
// threads per call cost more than the work at 12 500 batches (0.5 of 0.85 ms).  Re-entrant calls run inline.
void pool_run(unsigned tasks, const std::function<void(unsigned)>& f);
unsigned pool_width();   // threads that take part (including the caller)
// f(i0, i1) over [0, n) split into at most pool_width() ranges of at least `grain` items
void pool_ranges(size_t n, size_t grain, const std::function<void(size_t, size_t)>& f);

// Wall-clock phases of one call, printed to stderr when the environment has CHQ_TIMING=1 (development aid; off: one getenv
// per process).
struct PhaseTimer {
  const char* what; bool on; std::chrono::steady_clock::time_point t0, last; std::string line;
  explicit PhaseTimer(const char* w);
  void mark(const char* phase);
  ~PhaseTimer();
};

// ---- Arrow IPC stream with the body in HBM (ipc.cpp) ------------------------------------------------------------------
struct IpcMessage {
  std::vector<uint8_t> header;   // host: Schema message + the RecordBatch message's framing and metadata
  BufferPtr body;                // every Arrow buffer, 64-byte aligned, back to back
  int64_t body_len = 0;
  bool body_on_device = true;
  BufferPtr keep;
};
IpcMessage record_to_ipc(Context& ctx, const Batch& rec_dev, bool body_on_device);
// `body` null: the body follows the batch message inside `stream` (a complete host stream)
std::string describe_ipc(const uint8_t* stream, int64_t stream_len);   // host only
Batch record_from_ipc(Context& ctx, const uint8_t* stream, int64_t stream_len, const void* body, int64_t body_len,
                      bool body_on_device, bool out_on_device);

// host-only: result type / flags and the lowered device program of `expr` over a schema (text); throws the static error
std::string describe_plan(const ArrowSchema* schema, const chq_table_aliases* aliases, const Expr& expr, int64_t nrows,
                          bool enable_minus);

}  // namespace chq
