//! Raw FFI declarations for `include/chq.h` (libchq.so).  Source only: the build image has no Rust toolchain,
//! so this file is reviewed against the header, not compiled here.
#![allow(non_camel_case_types)]

use arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema};
use std::os::raw::{c_char, c_int, c_void};

pub const ARROW_DEVICE_CPU: c_int = 1;
pub const ARROW_DEVICE_ROCM: c_int = 10;

/// struct ArrowDeviceArray (Arrow C Device Data Interface)
#[repr(C)]
pub struct ArrowDeviceArray {
    pub array: FFI_ArrowArray,
    pub device_id: i64,
    pub device_type: i32,
    pub sync_event: *mut c_void,
    pub reserved: [i64; 3],
}
impl ArrowDeviceArray {
    /// an array slot for the library to fill (released: `array.release` is null)
    pub fn empty() -> ArrowDeviceArray {
        ArrowDeviceArray { array: FFI_ArrowArray::empty(), device_id: -1, device_type: ARROW_DEVICE_CPU, sync_event: std::ptr::null_mut(), reserved: [0; 3] }
    }
}

#[repr(C)]
pub struct chq_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct chq_expr {
    _private: [u8; 0],
}
#[repr(C)]
pub struct chq_parquet {
    _private: [u8; 0],
}

#[repr(C)]
pub struct chq_select_item {
    pub kind: c_int, // 0 Wildcard, 1 QualifiedWildcard, 2 UnnamedExpr, 3 ExprWithAlias
    pub expr: *const chq_expr,
    pub alias: *const c_char,
}

#[repr(C)]
pub struct chq_alias_list {
    pub aliases: *const *const c_char,
    pub n: c_int,
}
#[repr(C)]
pub struct chq_table_aliases {
    pub columns: *const chq_alias_list,
    pub n_columns: c_int,
}

/// struct chq_call_stats
#[repr(C)]
#[derive(Debug, Default, Clone, Copy)]
pub struct chq_call_stats {
    pub rows_in: i64,
    pub rows_out: i64,
    pub tiles: i64,
    pub launches: i64,
    pub bytes_read_alg: i64,
    pub bytes_written_alg: i64,
    pub kernel_ns: i64,
}

/// struct chq_column_desc (chq_wrap_columns)
#[repr(C)]
pub struct chq_column_desc {
    pub name: *const c_char,
    pub format: *const c_char,
    pub nullable: c_int,
    pub null_count: i64,
    pub offset: i64,
    pub validity: *const c_void,
    pub values: *const c_void,
    pub data: *const c_void,
}

/// struct chq_ipc_message: an Arrow IPC stream in three parts (header on the host, body in HBM or on the host, end marker)
#[repr(C)]
pub struct chq_ipc_message {
    pub header: *const u8,
    pub header_len: i64,
    pub body: *const c_void,
    pub body_len: i64,
    pub body_device_type: i32,
    pub body_device_id: i32,
    pub end_of_stream: [u8; 8],
    pub release: Option<unsafe extern "C" fn(*mut chq_ipc_message)>,
    pub private_data: *mut c_void,
}

/// chq_read_range_fn: fill dst[0 .. length) with the file's bytes [offset, offset + length); 0 = ok.  The reference's side of it
/// is an opendal reader (read_files_task.rs:233-250): `op.blocking().read_with(path).range(offset..offset + length)`.
pub type chq_read_range_fn = Option<unsafe extern "C" fn(user: *mut c_void, offset: i64, length: i64, dst: *mut u8) -> c_int>;

/// struct chq_parquet_image: a complete Parquet file in host memory
#[repr(C)]
pub struct chq_parquet_image {
    pub data: *const u8,
    pub len: i64,
    pub release: Option<unsafe extern "C" fn(*mut chq_parquet_image)>,
    pub private_data: *mut c_void,
}

// Every function include/chq.h declares, in header order (tests/test_abi.py checks names and arity against the header).
#[link(name = "chq")]
extern "C" {
    // ---- context ----
    pub fn chq_ctx_create(device_id: c_int, hip_stream: *mut c_void, out: *mut *mut chq_ctx) -> c_int;
    pub fn chq_ctx_destroy(ctx: *mut chq_ctx);
    pub fn chq_ctx_last_error(ctx: *const chq_ctx) -> *const c_char;
    pub fn chq_ctx_stream(ctx: *const chq_ctx) -> *mut c_void;
    pub fn chq_ctx_set_option(ctx: *mut chq_ctx, key: *const c_char, value: i64) -> c_int;
    pub fn chq_ctx_last_stats(ctx: *const chq_ctx, out: *mut chq_call_stats);
    pub fn chq_abi_version() -> c_int;
    pub fn chq_status_name(s: c_int) -> *const c_char;

    // ---- sqlparser::ast::Expr ----
    pub fn chq_expr_identifier(name: *const c_char) -> *mut chq_expr;
    pub fn chq_expr_compound_identifier(parts: *const *const c_char, n: c_int) -> *mut chq_expr;
    pub fn chq_expr_number(text: *const c_char, is_long: c_int) -> *mut chq_expr;
    pub fn chq_expr_boolean(value: c_int) -> *mut chq_expr;
    pub fn chq_expr_single_quoted_string(bytes: *const c_char, len: i64) -> *mut chq_expr;
    pub fn chq_expr_unsupported_value(debug: *const c_char) -> *mut chq_expr;
    pub fn chq_expr_binary_op(left: *mut chq_expr, op: c_int, op_debug: *const c_char, right: *mut chq_expr) -> *mut chq_expr;
    pub fn chq_expr_nested(inner: *mut chq_expr) -> *mut chq_expr;
    pub fn chq_expr_unsupported(debug: *const c_char) -> *mut chq_expr;
    pub fn chq_expr_free(e: *mut chq_expr);

    // ---- the path ----
    pub fn chq_filter_record(
        ctx: *mut chq_ctx, rec: *const ArrowDeviceArray, schema: *const FFI_ArrowSchema,
        table_aliases: *const chq_table_aliases, expr: *const chq_expr, out_device: c_int,
        out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema,
    ) -> c_int;
    /// the loop of filter_task.rs:78-126 over `n_records` same-schema batches, one kernel launch
    pub fn chq_filter_records(
        ctx: *mut chq_ctx, n_records: c_int, recs: *const *const ArrowDeviceArray, schema: *const FFI_ArrowSchema,
        table_aliases: *const chq_table_aliases, expr: *const chq_expr, out_device: c_int,
        outs: *mut ArrowDeviceArray, out_schemas: *mut FFI_ArrowSchema,
    ) -> c_int;
    /// the same, outputs joined into one batch + surviving rows per input record
    pub fn chq_filter_records_coalesced(
        ctx: *mut chq_ctx, n_records: c_int, recs: *const *const ArrowDeviceArray, schema: *const FFI_ArrowSchema,
        table_aliases: *const chq_table_aliases, expr: *const chq_expr, out_device: c_int,
        out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema, rows_per_record: *mut i64,
    ) -> c_int;
    pub fn chq_project_record(
        ctx: *mut chq_ctx, fields: *const chq_select_item, n_fields: c_int, rec: *const ArrowDeviceArray,
        schema: *const FFI_ArrowSchema, table_aliases: *const chq_table_aliases, out_device: c_int,
        out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema,
    ) -> c_int;
    pub fn chq_compute_value(
        ctx: *mut chq_ctx, rec: *const ArrowDeviceArray, schema: *const FFI_ArrowSchema,
        table_aliases: *const chq_table_aliases, expr: *const chq_expr, out_device: c_int,
        out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema, out_is_scalar: *mut c_int,
    ) -> c_int;
    /// filter_record followed by project_record on the survivors, one pass (filter_task.rs:99 + materialize_files_task.rs:110)
    pub fn chq_filter_project_record(
        ctx: *mut chq_ctx, predicate: *const chq_expr, fields: *const chq_select_item, n_fields: c_int,
        rec: *const ArrowDeviceArray, schema: *const FFI_ArrowSchema, table_aliases: *const chq_table_aliases,
        out_device: c_int, out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema,
    ) -> c_int;

    // ---- host half only ----
    pub fn chq_plan_describe(
        schema: *const FFI_ArrowSchema, table_aliases: *const chq_table_aliases, expr: *const chq_expr,
        n_rows: i64, enable_minus: c_int, buf: *mut c_char, buf_len: usize,
    ) -> c_int;

    // ---- device residency / exchange data plane ----
    pub fn chq_record_to_device(
        ctx: *mut chq_ctx, rec: *const ArrowDeviceArray, schema: *const FFI_ArrowSchema,
        out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema,
    ) -> c_int;
    pub fn chq_record_to_host(
        ctx: *mut chq_ctx, rec: *const ArrowDeviceArray, schema: *const FFI_ArrowSchema,
        out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema,
    ) -> c_int;
    /// HBM of src_ctx's GPU -> HBM of dst_ctx's GPU over xGMI (hipMemcpyPeerAsync); `out.sync_event` must be waited on
    pub fn chq_record_copy_to_peer(
        src_ctx: *mut chq_ctx, dst_ctx: *mut chq_ctx, rec: *const ArrowDeviceArray, schema: *const FFI_ArrowSchema,
        out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema,
    ) -> c_int;
    /// Arrow IPC stream of one batch, body assembled in one HBM allocation (messages/exchange.rs:145-197)
    pub fn chq_record_to_ipc(
        ctx: *mut chq_ctx, rec: *const ArrowDeviceArray, schema: *const FFI_ArrowSchema, body_device: c_int,
        out: *mut chq_ipc_message,
    ) -> c_int;
    /// inverse (messages/exchange.rs:247-276); `body` null = the body follows the metadata inside `stream`
    pub fn chq_record_from_ipc(
        ctx: *mut chq_ctx, stream: *const u8, stream_len: i64, body: *const c_void, body_len: i64,
        body_device_type: c_int, out_device: c_int, out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema,
    ) -> c_int;
    pub fn chq_ipc_describe(stream: *const u8, stream_len: i64, buf: *mut c_char, buf_len: usize) -> c_int;

    // ---- Parquet scan, page decode on the GPU (read_files_task.rs:233-282) ----
    pub fn chq_parquet_open(file: *const u8, file_len: i64, out: *mut *mut chq_parquet, err: *mut c_char, err_len: usize) -> c_int;
    pub fn chq_parquet_open_reader(
        file_len: i64, read: chq_read_range_fn, user: *mut c_void, out: *mut *mut chq_parquet, err: *mut c_char, err_len: usize,
    ) -> c_int;
    pub fn chq_parquet_close(pq: *mut chq_parquet);
    pub fn chq_parquet_num_columns(pq: *const chq_parquet) -> i32;
    pub fn chq_parquet_column_name(pq: *const chq_parquet, column: i32) -> *const c_char;
    pub fn chq_parquet_read_columns(
        ctx: *mut chq_ctx, pq: *const chq_parquet, first: i32, count: i32, columns: *const i32, n_columns: i32, out_device: c_int,
        outs: *mut ArrowDeviceArray, out_schemas: *mut FFI_ArrowSchema,
    ) -> c_int;
    pub fn chq_parquet_num_row_groups(pq: *const chq_parquet) -> i32;
    pub fn chq_parquet_row_group_num_rows(pq: *const chq_parquet, row_group: i32) -> i64;
    pub fn chq_parquet_describe(pq: *const chq_parquet, buf: *mut c_char, buf_len: usize) -> c_int;
    pub fn chq_parquet_read_row_group(
        ctx: *mut chq_ctx, pq: *const chq_parquet, row_group: i32, out_device: c_int,
        out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema,
    ) -> c_int;
    pub fn chq_parquet_read_row_groups(
        ctx: *mut chq_ctx, pq: *const chq_parquet, first: i32, count: i32, out_device: c_int,
        outs: *mut ArrowDeviceArray, out_schemas: *mut FFI_ArrowSchema,
    ) -> c_int;
    /// one record batch -> one Parquet file image in host memory, pages encoded on the GPU (materialize_files_task.rs:128-141)
    pub fn chq_record_to_parquet(
        ctx: *mut chq_ctx, rec: *const ArrowDeviceArray, schema: *const FFI_ArrowSchema, out: *mut chq_parquet_image,
    ) -> c_int;
    /// several batches of one schema -> one file, one row group per batch (the compaction DEV_NOTES.md:117-121 plans)
    pub fn chq_records_to_parquet(
        ctx: *mut chq_ctx, n_records: c_int, recs: *const *const ArrowDeviceArray, schema: *const FFI_ArrowSchema,
        out: *mut chq_parquet_image,
    ) -> c_int;
    pub fn chq_wrap_columns(
        ctx: *mut chq_ctx, cols: *const chq_column_desc, n_cols: c_int, n_rows: i64, device_type: c_int,
        out: *mut ArrowDeviceArray, out_schema: *mut FFI_ArrowSchema,
    ) -> c_int;
}
