/*
 * chq.h -- C ABI of the MI355X-native filter / projection record kernels for ChapterhouseDB.
 *
 * This is the drop-in boundary for ONE path of the reference (alekLukanen/ChapterhouseQE): the
 * record_utils functions its `filter` and `materialize` operator tasks call once per RecordBatch.
 * Reference paths are relative to the reference repo root; RU = src/handlers/operator_handler/
 * operators/record_utils.
 *
 *   chq_filter_record   replaces  RU/filter_record.rs:21-25    pub fn filter_record(rec, table_aliases, expr)
 *                        called at operators/filter_tasks/filter_task.rs:99
 *   chq_project_record  replaces  RU/record_projection.rs:16-20 pub fn project_record(fields, record, table_aliases)
 *                        called at operators/materialize_tasks/materialize_files_task.rs:110
 *   chq_compute_value   replaces  RU/compute_value.rs:57-61     pub fn compute_value(rec, table_aliases, expr)
 *   chq_expr_*          carry     sqlparser::ast::Expr  (the variants RU/compute_value.rs:62-343 matches on)
 *   chq_select_item     carries   sqlparser::ast::SelectItem (RU/record_projection.rs:25-69)
 *   chq_table_aliases   carries   table_aliases: &Vec<Vec<String>> (RU/record_aliases.rs:12-59)
 *   status codes        mirror    ComputeValueError (RU/compute_value.rs:13-32), FilterRecordError
 *                                 (RU/filter_record.rs:12-15), ProjectRecordError (RU/record_projection.rs:11-14)
 *                                 and the ArrowError variants of the arrow-rs 53 kernels behind them.
 *
 * Record batches cross the boundary as Arrow C (Device) Data Interface structs (chq_arrow_abi.h): a
 * struct-typed ArrowDeviceArray with one child per column plus its ArrowSchema.  Inputs may live on
 * the host (ARROW_DEVICE_CPU; the library stages them to HBM) or already on the GPU (ARROW_DEVICE_ROCM,
 * zero copy).  Inputs are borrowed for the duration of the call and never modified or released.
 * Outputs are freshly allocated, owned by the caller and freed through the standard Arrow `release`
 * callbacks; the caller chooses whether they are returned in HBM (stay on the GPU for the next
 * operator) or copied back to host memory.
 *
 * No torch / C++ types appear here.  All functions are thread-safe per context: one chq_ctx per
 * operator instance (the reference runs one batch at a time per instance, filter_task.rs:86-125).
 * Everything is computed by HIP kernels on gfx950; there is no CPU fallback -- a missing or unusable
 * device is an error (CHQ_ERR_DEVICE).
 */
#ifndef CHQ_H
#define CHQ_H

#include <stdint.h>
#include "chq_arrow_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CHQ_ABI_VERSION 1

/* ---- status codes --------------------------------------------------------------------------- */
typedef enum chq_status {
  CHQ_OK = 0,
  /* ComputeValueError, RU/compute_value.rs:13-32 */
  CHQ_ERR_VALUE_TYPE_NOT_IMPLEMENTED = 1,
  CHQ_ERR_EXPRESSION_TYPE_NOT_IMPLEMENTED = 2,
  CHQ_ERR_BINARY_OPERATOR_NOT_IMPLEMENTED = 3,
  CHQ_ERR_BINARY_OPERATION_CAST_FAILED = 4,
  CHQ_ERR_FAILED_TO_PARSE_AS_AN_INTEGER = 5,
  CHQ_ERR_FAILED_TO_PARSE_AS_A_FLOAT = 6,
  CHQ_ERR_COLUMN_NOT_FOUND = 7,
  CHQ_ERR_IDENTIFIER_NOT_FOUND = 8,
  CHQ_ERR_UNSUPPORTED_TYPE_COERSION = 9,
  /* FilterRecordError, RU/filter_record.rs:12-15 */
  CHQ_ERR_CAST_TO_BOOLEAN_ARRAY_FAILED = 10,
  /* ProjectRecordError, RU/record_projection.rs:11-14 */
  CHQ_ERR_PROJECT_NOT_IMPLEMENTED = 11,
  /* ArrowError raised by the arrow-rs kernels the reference calls */
  CHQ_ERR_ARROW_ARITHMETIC_OVERFLOW = 20,
  CHQ_ERR_ARROW_DIVIDE_BY_ZERO = 21,
  CHQ_ERR_ARROW_INVALID_ARGUMENT = 22,
  CHQ_ERR_ARROW_COMPUTE = 23,
  CHQ_ERR_ARROW_CAST = 24,
  /* this implementation */
  CHQ_ERR_NOT_SUPPORTED = 30,   /* valid in the reference, outside this build's scope */
  CHQ_ERR_INVALID_HANDLE = 31,
  CHQ_ERR_DEVICE = 40,          /* HIP runtime / no usable gfx950 device */
  CHQ_ERR_OUT_OF_MEMORY = 41
} chq_status;

/* ---- context -------------------------------------------------------------------------------- */
typedef struct chq_ctx chq_ctx;

/* One context per operator instance.  `hip_stream` is a hipStream_t to run on (pass hipStreamLegacy /
 * hipStreamPerThread for the default streams), or NULL to let the context create its own non-blocking
 * stream -- in that case work the caller issued on other streams must be synchronised by the caller.  Fails with CHQ_ERR_DEVICE when no GPU is usable. */
chq_status chq_ctx_create(int device_id, void* hip_stream, chq_ctx** out);
void chq_ctx_destroy(chq_ctx* ctx);
/* Message of the last failing call on this context (valid until the next call on it). */
const char* chq_ctx_last_error(const chq_ctx* ctx);
/* hipStream_t the context launches on. */
void* chq_ctx_stream(const chq_ctx* ctx);
/* Options: "tile_kind" (-1 auto, 0: 16384-row tiles, 1/2: 2048-row tiles), "enable_minus",
 * "time_kernels", "trim_pool" (return every cached HBM / host block to the system), "host_pool_bytes" (cap of the
 * process-wide cache of host result buffers, default 8 GiB, 0 disables it), "fuse" (chq_filter_project_record's
 * single-pass kernel: 0 never, 1 = default: when it moves clearly fewer bytes than the two steps, 2 whenever the
 * inputs allow), "group_mode" (chq_filter_records layout: 0 auto, 1 per-tile table, 2 wave-packed).  Switches that exist for A/B
 * measurements and tests (all default 1): "fold_utf8" (short-string Utf8 columns inside the main kernel), "group_fold"
 * (the same for batch groups), "group_bits" (validity bitmaps / Boolean columns of a wave-packed device group in the
 * one-launch path), "stash" (-1 auto .. 2 predicate columns kept in LDS between the two phases), "split_rows" (rows from
 * which a batch is launched as complete tiles + tail), "uniform_utf8_rows" (batches of at least this many rows -- default 2^24, 0 = never -- have
 * their Utf8 columns checked for ONE value length and, if so, filtered as fixed-width columns), "parquet_page_rows" (chq_record_to_parquet: rows per data page,
 * default 65 536, rounded to multiples of 4 096; a chunk never gets more than 64 pages), "snappy_blocks" (Parquet scan: 1 = a
 * snappy page of three or more 64 KiB blocks is inflated one wave per BLOCK after a walk of its element chain, 0 = one wave
 * per page, 2 = the blocks give up and the page is redone whole -- the path a stream with blocks that depend on each other
 * takes --, 3 = as 1 but every chain walked by one wave instead of one wave per segment of the page's input; 2 and 3 for tests).
 * Unknown keys fail.
 *
 * Type coverage of expressions = the reference's (RU/compute_value.rs:350-431): the integer / float coercion table,
 * Utf8 and Boolean comparisons, Float16 (widening, f16 arithmetic and comparisons), same-type comparisons of Date32 /
 * Date64 / Time32 / Time64 / Timestamp / Duration / Decimal128 columns, and casts to Boolean under AND / OR from numeric
 * and Utf8 operands (a spelling arrow-cast does not know is NULL).  CHQ_ERR_NOT_SUPPORTED remains for ARITHMETIC on
 * decimals, durations and date / timestamp differences, and for comparisons of FixedSizeBinary / interval columns. */
chq_status chq_ctx_set_option(chq_ctx* ctx, const char* key, int64_t value);
/* Counters of the last filter call: rows in, rows out, tiles, kernel launches. */
typedef struct chq_call_stats {
  int64_t rows_in, rows_out, tiles, launches;
  int64_t bytes_read_alg, bytes_written_alg;   /* algorithmic bytes per SURVEY.md section 8(d) */
  int64_t kernel_ns;   /* duration of the main kernel launch(es), HIP events on the context's stream; 0 unless
                          the context option "time_kernels" is set */
} chq_call_stats;
void chq_ctx_last_stats(const chq_ctx* ctx, chq_call_stats* out);

int chq_abi_version(void);
const char* chq_status_name(chq_status s);

/* ---- expressions: sqlparser::ast::Expr ------------------------------------------------------- */
typedef struct chq_expr chq_expr;

/* sqlparser::ast::BinaryOperator; arms implemented by RU/compute_value.rs:70-209 plus Minus, which the
 * reference rejects (compute_value.rs:210-216) and so does this library unless the context option
 * "enable_minus" is set. Any other operator: CHQ_BINOP_OTHER. */
typedef enum chq_binary_operator {
  CHQ_BINOP_AND = 0, CHQ_BINOP_OR, CHQ_BINOP_PLUS, CHQ_BINOP_MINUS, CHQ_BINOP_MULTIPLY, CHQ_BINOP_DIVIDE,
  CHQ_BINOP_MODULO, CHQ_BINOP_EQ, CHQ_BINOP_NOTEQ, CHQ_BINOP_GT, CHQ_BINOP_GTEQ, CHQ_BINOP_LT, CHQ_BINOP_LTEQ,
  CHQ_BINOP_OTHER
} chq_binary_operator;

chq_expr* chq_expr_identifier(const char* name);                              /* Expr::Identifier */
chq_expr* chq_expr_compound_identifier(const char* const* parts, int n);      /* Expr::CompoundIdentifier */
chq_expr* chq_expr_number(const char* text, int is_long);                     /* Expr::Value(Value::Number(text, long)) */
chq_expr* chq_expr_boolean(int value);                                        /* Value::Boolean */
chq_expr* chq_expr_single_quoted_string(const char* bytes, int64_t len);      /* Value::SingleQuotedString */
chq_expr* chq_expr_unsupported_value(const char* debug);                      /* any other Value */
/* Expr::BinaryOp { left, op, right }; takes ownership of both children. `op_debug` is the operator's
 * Debug text, used in the BinaryOperatorNotImplemented message. */
chq_expr* chq_expr_binary_op(chq_expr* left, chq_binary_operator op, const char* op_debug, chq_expr* right);
chq_expr* chq_expr_nested(chq_expr* inner);                                   /* Expr::Nested; takes ownership */
chq_expr* chq_expr_unsupported(const char* debug);                            /* any other Expr variant */
void chq_expr_free(chq_expr* e);

/* sqlparser::ast::SelectItem, RU/record_projection.rs:25-69 */
typedef enum chq_select_item_kind {
  CHQ_ITEM_WILDCARD = 0, CHQ_ITEM_QUALIFIED_WILDCARD, CHQ_ITEM_UNNAMED_EXPR, CHQ_ITEM_EXPR_WITH_ALIAS
} chq_select_item_kind;
typedef struct chq_select_item {
  int kind;               /* chq_select_item_kind */
  const chq_expr* expr;   /* UNNAMED_EXPR / EXPR_WITH_ALIAS */
  const char* alias;      /* EXPR_WITH_ALIAS */
} chq_select_item;

/* table_aliases: &Vec<Vec<String>> -- one alias list per column (RU/record_aliases.rs:12-59).
 * `n_columns` may be smaller than the batch's column count (the reference's tests pass vec![]). */
typedef struct chq_alias_list { const char* const* aliases; int n; } chq_alias_list;
typedef struct chq_table_aliases { const chq_alias_list* columns; int n_columns; } chq_table_aliases;

/* ---- the path -------------------------------------------------------------------------------- */
/* `out_device`: ARROW_DEVICE_CPU or ARROW_DEVICE_ROCM. On success *out / *out_schema are filled and
 * must be released by the caller; on failure they are left released (release == NULL). */

/* RU/filter_record.rs:21-39: evaluate `expr` to a BooleanArray, keep the rows where it is true and
 * valid, every column, original order, same schema; possibly zero rows. */
chq_status chq_filter_record(chq_ctx* ctx, const struct ArrowDeviceArray* rec, const struct ArrowSchema* schema,
                             const chq_table_aliases* table_aliases, const chq_expr* expr, int out_device,
                             struct ArrowDeviceArray* out, struct ArrowSchema* out_schema);

/* chq_filter_record over `n_records` batches of ONE schema, in one call: out[i] is exactly what
 * chq_filter_record(recs[i]) returns, and on failure the status / message are those of the first
 * batch (in array order) whose single call fails; nothing is returned then.  This is the loop of
 * filter_task.rs:78-126 (get_next_record -> filter_record -> send) hoisted below the boundary: the
 * reference's batches are 10 000 rows (src/planner/physical_planner.rs:323), far too small to fill
 * the GPU one at a time.  When every column is fixed-width without nulls, all batches run in ONE
 * kernel launch (one chained compaction; out[i] are slices of one dense buffer per column); other
 * groups are processed batch by batch inside the call.  `outs` / `out_schemas`: n_records entries. */
chq_status chq_filter_records(chq_ctx* ctx, int n_records, const struct ArrowDeviceArray* const* recs,
                              const struct ArrowSchema* schema, const chq_table_aliases* table_aliases,
                              const chq_expr* expr, int out_device, struct ArrowDeviceArray* outs,
                              struct ArrowSchema* out_schemas);

/* The same call with the results joined: ONE output batch = the surviving rows of recs[0], recs[1], ... back to
 * back (input order), and rows_per_record[i] (optional, n_records entries) = how many of them came from recs[i].
 * This is the batch coalescing the reference plans for its exchange (DEV_NOTES.md:175-182):
 * downstream operators see one large batch instead of 10^5 small ones; a caller that needs the per-record ids back
 * slices the output at the running sums of rows_per_record.  In the one-launch case the output IS the kernel's dense
 * buffer, so nothing is copied or exported per input batch. */
chq_status chq_filter_records_coalesced(chq_ctx* ctx, int n_records, const struct ArrowDeviceArray* const* recs,
                                        const struct ArrowSchema* schema, const chq_table_aliases* table_aliases,
                                        const chq_expr* expr, int out_device, struct ArrowDeviceArray* out,
                                        struct ArrowSchema* out_schema, int64_t* rows_per_record);

/* RU/record_projection.rs:16-76 */
chq_status chq_project_record(chq_ctx* ctx, const chq_select_item* fields, int n_fields,
                              const struct ArrowDeviceArray* rec, const struct ArrowSchema* schema,
                              const chq_table_aliases* table_aliases, int out_device,
                              struct ArrowDeviceArray* out, struct ArrowSchema* out_schema);

/* RU/compute_value.rs:57-344: returns ArrayDatum { array, is_scalar } */
chq_status chq_compute_value(chq_ctx* ctx, const struct ArrowDeviceArray* rec, const struct ArrowSchema* schema,
                             const chq_table_aliases* table_aliases, const chq_expr* expr, int out_device,
                             struct ArrowDeviceArray* out, struct ArrowSchema* out_schema, int* out_is_scalar);

/* filter_record followed by project_record on the surviving rows -- the reference's
 * filter -> exchange -> materialize sequence (filter_task.rs:99, materialize_files_task.rs:110) fused
 * into one call so the filtered batch never leaves the GPU. */
chq_status chq_filter_project_record(chq_ctx* ctx, const chq_expr* predicate, const chq_select_item* fields,
                                     int n_fields, const struct ArrowDeviceArray* rec,
                                     const struct ArrowSchema* schema, const chq_table_aliases* table_aliases,
                                     int out_device, struct ArrowDeviceArray* out, struct ArrowSchema* out_schema);

/* ---- host half only (no GPU, no context) --------------------------------------------------------- */
/* Types `expr` against `schema` (a struct schema as in the record calls) exactly as chq_compute_value would --
 * literal typing, column / alias lookup, the coercion table, constant folding, the arrow length rules -- and writes
 * a description into `buf`: "result <Type> scalar=<0|1> len1=<0|1>", then the folded value, the column index or the
 * listing of the device program.  Returns the static status the record calls would return for a batch of n_rows rows
 * (message in `buf`).  Used by the CPU test tier and for debugging planners; touches no device. */
chq_status chq_plan_describe(const struct ArrowSchema* schema, const chq_table_aliases* table_aliases,
                             const chq_expr* expr, int64_t n_rows, int enable_minus, char* buf, size_t buf_len);

/* ---- device residency helpers ----------------------------------------------------------------- */
/* Copy a host batch into HBM / a device batch back to host memory. */
chq_status chq_record_to_device(chq_ctx* ctx, const struct ArrowDeviceArray* rec, const struct ArrowSchema* schema,
                                struct ArrowDeviceArray* out, struct ArrowSchema* out_schema);
chq_status chq_record_to_host(chq_ctx* ctx, const struct ArrowDeviceArray* rec, const struct ArrowSchema* schema,
                              struct ArrowDeviceArray* out, struct ArrowSchema* out_schema);

/* The data-plane half of the exchange step, GPU to GPU: copy a batch that lives in the HBM of `src_ctx`'s device into
 * the HBM of `dst_ctx`'s device with hipMemcpyPeerAsync -- peer to peer over the xGMI link of the pair, no host staging,
 * no collective.  This is what moves a record when the DAG puts its consumer on another GPU (one filter instance per
 * GPU feeding one materialize instance) for a worker that runs several operator instances in ONE process
 * (src/worker/query_worker.rs:35-39), one chq_ctx per GPU.  The reference's exchange semantics are the caller's and do
 * not change: the record keeps its record_id and table_aliases, the inbound exchange is acked only after the copy was
 * handed over (exchange_operator.rs:621-667, requests/send_record_request.rs:33-104).
 * The copies are enqueued on dst_ctx's stream; the call returns without waiting for them.  out->sync_event points to a
 * hipEvent_t (owned by `out`, destroyed by its release callback) recorded behind the last copy: every chq call that takes
 * `out` as input waits on it, any other consumer must (Arrow C Device Data Interface).  `rec` must stay alive until that
 * event has completed.  src_ctx == dst_ctx's device is allowed (a device-local deep copy). */
chq_status chq_record_copy_to_peer(chq_ctx* src_ctx, chq_ctx* dst_ctx, const struct ArrowDeviceArray* rec,
                                   const struct ArrowSchema* schema, struct ArrowDeviceArray* out,
                                   struct ArrowSchema* out_schema);

/* ---- Arrow IPC with the message body in HBM ------------------------------------------------------------------------ */
/* The wire format the reference puts a batch in when it leaves the process: an Arrow IPC *stream* -- Schema message,
 * one RecordBatch message, end-of-stream -- written by arrow::ipc::writer::StreamWriter and read back by
 * arrow::ipc::reader::StreamReader (src/handlers/message_handler/messages/exchange.rs:145-197, 247-276).  For a batch
 * that lives on the GPU the two metadata flatbuffers are built on the host and the message BODY (every Arrow buffer,
 * rebased to offset 0, 64-byte aligned, back to back) is assembled in ONE HBM allocation, so it travels with a single
 * RCCL send / peer copy, or one D2H copy when it must cross TCP.  The bytes
 *     header[0 .. header_len)  ++  body[0 .. body_len)  ++  end_of_stream[0 .. 8)
 * are exactly the stream an Arrow reader expects.  Dictionaries, compression and nested types: CHQ_ERR_NOT_SUPPORTED. */
typedef struct chq_ipc_message {
  const uint8_t* header;      /* host memory: Schema message + framing and metadata of the RecordBatch message */
  int64_t header_len;
  const void* body;           /* HBM (ARROW_DEVICE_ROCM) or host memory (ARROW_DEVICE_CPU), see body_device_type */
  int64_t body_len;
  int32_t body_device_type;
  int32_t body_device_id;
  uint8_t end_of_stream[8];   /* 0xFFFFFFFF 0x00000000 */
  void (*release)(struct chq_ipc_message*);
  void* private_data;
} chq_ipc_message;

/* `rec`: host or device resident.  `body_device`: ARROW_DEVICE_ROCM (the body stays in HBM) or ARROW_DEVICE_CPU. */
chq_status chq_record_to_ipc(chq_ctx* ctx, const struct ArrowDeviceArray* rec, const struct ArrowSchema* schema,
                             int body_device, chq_ipc_message* out);
/* Inverse.  `stream` (host memory) holds the Schema message and the RecordBatch message's metadata; the body is `body`
 * (in memory of kind `body_device_type`), or -- when `body` is NULL -- follows the metadata inside `stream` itself, i.e.
 * `stream` is a complete Arrow IPC stream as any Arrow writer produces it.  The body is moved to `out_device` with one
 * copy; the returned batch owns it. */
chq_status chq_record_from_ipc(chq_ctx* ctx, const uint8_t* stream, int64_t stream_len, const void* body, int64_t body_len,
                               int body_device_type, int out_device, struct ArrowDeviceArray* out,
                               struct ArrowSchema* out_schema);

/* Host half only (no GPU, no context): what the metadata of an Arrow IPC stream says -- "rows R body B body_at P", one
 * "field <name> <format> nullable=<0|1> nulls=<k>" line per column, one "buffer <offset> <length>" line per buffer.  Used
 * by the CPU test tier (the flatbuffer reader against pyarrow's writer) and for debugging. */
chq_status chq_ipc_describe(const uint8_t* stream, int64_t stream_len, char* buf, size_t buf_len);

/* ---- Parquet scan with the page decode on the GPU --------------------------------------------
 * Replaces the decode the reference does with the `parquet` crate in front of the filter path
 * (operators/table_func_tasks/read_files_task.rs:233-282: ParquetRecordBatchStreamBuilder::new(reader) ...
 * with_batch_size(max_rows_per_batch)): the caller hands over the file's bytes (host memory, borrowed until
 * chq_parquet_close), the library parses footer and page headers on the host, uploads each column chunk as it lies in the
 * file and decodes the pages into Arrow buffers in HBM.  One call yields one ROW GROUP as one device-resident batch (the
 * reference cuts 10 000-row batches; slices of the result are zero-copy, and the filter kernels prefer large batches).
 * Scope: flat schemas; BOOLEAN, INT32, INT64, FLOAT, DOUBLE, BYTE_ARRAY annotated String; required / optional columns;
 * PLAIN and RLE_DICTIONARY / PLAIN_DICTIONARY (also mixed inside a chunk: the writers' dictionary fallback); data pages
 * V1 and V2; UNCOMPRESSED (what the reference's writers produce: create_sample_data.rs:222,
 * materialize_files_task.rs:128-133) and SNAPPY (what pyarrow / Spark / DuckDB write by default; the pages are inflated on
 * the GPU).  Everything else (GZIP, ZSTD, LZ4, BROTLI pages; nested schemas; DELTA_* encodings): CHQ_ERR_NOT_SUPPORTED, the
 * message names the feature. */
typedef struct chq_parquet chq_parquet;
/* Footer + page headers; no GPU, no context.  On failure *out is NULL and `err` (if given) receives the message. */
chq_status chq_parquet_open(const uint8_t* file, int64_t file_len, chq_parquet** out, char* err, size_t err_len);
/* The same over a RANGE READER instead of the whole file in memory: the reference reads Parquet through opendal ranges
 * (read_files_task.rs:233-250: op.reader_with(path), ParquetRecordBatchStreamBuilder over it), i.e. the footer and then only
 * the column chunks a query decodes.  `read(user, offset, length, dst)` must fill dst[0 .. length) with the file's bytes
 * [offset, offset + length) and return 0; any other value fails the call.  Open reads the file's tail (footer) and nothing
 * else; every read call then fetches exactly the column chunks it decodes, one `read` per chunk, from the calling thread.
 * `read` / `user` must stay valid until chq_parquet_close. */
typedef int (*chq_read_range_fn)(void* user, int64_t offset, int64_t length, uint8_t* dst);
chq_status chq_parquet_open_reader(int64_t file_len, chq_read_range_fn read, void* user, chq_parquet** out, char* err, size_t err_len);
void chq_parquet_close(chq_parquet* pq);
int32_t chq_parquet_num_columns(const chq_parquet* pq);
/* Name of column `column` (valid until chq_parquet_close), or NULL. */
const char* chq_parquet_column_name(const chq_parquet* pq, int32_t column);
int32_t chq_parquet_num_row_groups(const chq_parquet* pq);
int64_t chq_parquet_row_group_num_rows(const chq_parquet* pq, int32_t row_group);
/* "rows R row_groups G columns C" / "column <name> <physical> <required|optional> [utf8]" / "rg <i> rows <n>" /
 * "chunk <col> values <n> codec <c> pages <p>" / "page <type> values <n> enc <e> bytes <b> header <h>" lines (CPU test tier) */
chq_status chq_parquet_describe(const chq_parquet* pq, char* buf, size_t buf_len);
/* Decode one row group.  `out_device`: ARROW_DEVICE_ROCM (stays in HBM) or ARROW_DEVICE_CPU (copied down). */
chq_status chq_parquet_read_row_group(chq_ctx* ctx, const chq_parquet* pq, int32_t row_group, int out_device,
                                      struct ArrowDeviceArray* out, struct ArrowSchema* out_schema);
/* Row groups [first, first + count), one batch each in outs[i] / out_schemas[i]: all of them are in flight together (the
 * upload of one overlaps the decode of the others) and the host synchronises twice per call instead of twice per row
 * group.  On failure nothing is returned. */
chq_status chq_parquet_read_row_groups(chq_ctx* ctx, const chq_parquet* pq, int32_t first, int32_t count, int out_device,
                                       struct ArrowDeviceArray* outs, struct ArrowSchema* out_schemas);

/* Column pruning (the reference's own TODO, DEV_NOTES.md:123): the same call for the `n_columns` columns listed in
 * `columns` (indices into the file's schema, in the order the output batch should carry them; NULL = every column).  Only
 * those columns' chunks are fetched (range reader), uploaded and decoded.  chq_ctx_last_stats afterwards: rows_in = rows
 * decoded, bytes_read_alg = file bytes sent to the GPU, bytes_written_alg = Arrow bytes produced. */
chq_status chq_parquet_read_columns(chq_ctx* ctx, const chq_parquet* pq, int32_t first, int32_t count, const int32_t* columns,
                                    int32_t n_columns, int out_device, struct ArrowDeviceArray* outs, struct ArrowSchema* out_schemas);

/* ---- Parquet write with the page encode on the GPU ---------------------------------------
 * Replaces the encode the reference does with the `parquet` crate behind project_record
 * (operators/materialize_tasks/materialize_files_task.rs:128-141: AsyncArrowWriter::try_new(writer, schema, None),
 * write(&rec), close() -- one file with one row group per record).  `rec` (host or device resident) becomes one complete
 * Parquet file image in host memory, ready for the storage writer: the value streams of the pages are produced in HBM and
 * copied once into place, headers and footer are written on the host.  Format: PLAIN, UNCOMPRESSED, data pages V1 of
 * "parquet_page_rows" rows, definition levels = the Arrow validity bitmap, chunk statistics (null_count, min / max).  Every
 * Parquet reader decodes it; it is not byte-identical to the parquet crate's output (which dictionary-encodes first and
 * adds page indexes).
 * Types: Int32, Int64, Float32, Float64, Boolean, Utf8; others CHQ_ERR_NOT_SUPPORTED. */
typedef struct chq_parquet_image {
  const uint8_t* data;   /* host memory, `len` bytes: "PAR1" ... footer ... "PAR1" */
  int64_t len;
  void (*release)(struct chq_parquet_image*);
  void* private_data;
} chq_parquet_image;
chq_status chq_record_to_parquet(chq_ctx* ctx, const struct ArrowDeviceArray* rec, const struct ArrowSchema* schema,
                                 chq_parquet_image* out);
/* The same for `n_records` batches of ONE schema (names, types and nullability must agree: CHQ_ERR_ARROW_INVALID_ARGUMENT
 * otherwise): one file, one row group per batch, in order -- the compaction of several records into a larger file that the
 * reference's DEV_NOTES.md:117-121 plans for the materialize task. */
chq_status chq_records_to_parquet(chq_ctx* ctx, int n_records, const struct ArrowDeviceArray* const* recs,
                                  const struct ArrowSchema* schema, chq_parquet_image* out);

/* Wrap caller-owned device (or host) buffers as a record batch without copying; the buffers must
 * outlive the returned structs, whose release callbacks free only the descriptors. `format` is an
 * Arrow C format string ("i","f","g","l","b","u", ...). */
typedef struct chq_column_desc {
  const char* name;
  const char* format;
  int nullable;
  int64_t null_count;
  int64_t offset;          /* logical offset in elements (Arrow slice offset) */
  const void* validity;    /* NULL when null_count == 0 */
  const void* values;      /* fixed width: values; bool: bitmap; utf8: int32 offsets */
  const void* data;        /* utf8: bytes */
} chq_column_desc;
chq_status chq_wrap_columns(chq_ctx* ctx, const chq_column_desc* cols, int n_cols, int64_t n_rows, int device_type,
                            struct ArrowDeviceArray* out, struct ArrowSchema* out_schema);

#ifdef __cplusplus
}
#endif
#endif /* CHQ_H */
