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

#[repr(C)]
pub struct chq_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct chq_expr {
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

#[link(name = "chq")]
extern "C" {
    pub fn chq_ctx_create(device_id: c_int, hip_stream: *mut c_void, out: *mut *mut chq_ctx) -> c_int;
    pub fn chq_ctx_destroy(ctx: *mut chq_ctx);
    pub fn chq_ctx_last_error(ctx: *const chq_ctx) -> *const c_char;

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
}
