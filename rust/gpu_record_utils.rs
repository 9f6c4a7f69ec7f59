//! Drop-in replacements for `record_utils::filter_record` / `record_utils::project_record`
//! (src/handlers/operator_handler/operators/record_utils/{filter_record.rs:21-25, record_projection.rs:16-20})
//! that run on the MI355X through libchq.  Same signatures, same `anyhow::Result` error behaviour: every
//! non-zero chq_status becomes an error whose text is the library's message (status numbering mirrors
//! ComputeValueError / FilterRecordError / ProjectRecordError / ArrowError, include/chq.h).
//!
//! Source only (no Rust toolchain in the build image).  Needs `arrow = { version = "53.1", features = ["ffi"] }`
//! (the reference enables only "prettyprint", Cargo.toml:42).
use std::ffi::{CStr, CString};
use std::os::raw::c_char;
use std::sync::Arc;

use anyhow::{anyhow, Result};
use arrow::array::{Array, RecordBatch, StructArray};
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use sqlparser::ast::{BinaryOperator, Expr, SelectItem, Value};

use super::chq_sys::*;

/// One per operator instance (the reference runs one batch at a time per instance, filter_task.rs:86-125).
pub struct GpuContext(*mut chq_ctx);
unsafe impl Send for GpuContext {}

impl GpuContext {
    pub fn new(device_id: i32) -> Result<GpuContext> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { chq_ctx_create(device_id, std::ptr::null_mut(), &mut ctx) };
        if rc != 0 {
            return Err(anyhow!("chq_ctx_create failed with status {rc}: no usable GPU"));
        }
        Ok(GpuContext(ctx))
    }
    /// the handle, for entry points this module does not wrap (Parquet scan / write, IPC, peer copies)
    pub fn raw(&self) -> *mut chq_ctx {
        self.0
    }
    /// chq_status -> Result, with the library's message for this context
    pub fn check(&self, rc: i32) -> Result<()> {
        if rc == 0 { Ok(()) } else { Err(self.err(rc)) }
    }
    fn err(&self, rc: i32) -> anyhow::Error {
        let msg = unsafe { CStr::from_ptr(chq_ctx_last_error(self.0)) }.to_string_lossy().into_owned();
        anyhow!("[chq status {rc}] {msg}")
    }
}
impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { chq_ctx_destroy(self.0) }
    }
}

/// sqlparser::ast::Expr -> chq_expr (the variants compute_value.rs:62-343 matches on; everything else is
/// passed as "unsupported" so the library reports the same ExpressionTypeNotImplemented error).
fn lower_expr(e: &Expr) -> *mut chq_expr {
    let c = |s: &str| CString::new(s).unwrap();
    unsafe {
        match e {
            Expr::Nested(inner) => chq_expr_nested(lower_expr(inner)),
            Expr::BinaryOp { left, op, right } => {
                let code = match op {
                    BinaryOperator::And => 0, BinaryOperator::Or => 1, BinaryOperator::Plus => 2,
                    BinaryOperator::Minus => 3, BinaryOperator::Multiply => 4, BinaryOperator::Divide => 5,
                    BinaryOperator::Modulo => 6, BinaryOperator::Eq => 7, BinaryOperator::NotEq => 8,
                    BinaryOperator::Gt => 9, BinaryOperator::GtEq => 10, BinaryOperator::Lt => 11,
                    BinaryOperator::LtEq => 12, _ => 13,
                };
                chq_expr_binary_op(lower_expr(left), code, c(&format!("{:?}", op)).as_ptr(), lower_expr(right))
            }
            Expr::Value(Value::Number(text, long)) => chq_expr_number(c(text).as_ptr(), *long as i32),
            Expr::Value(Value::Boolean(b)) => chq_expr_boolean(*b as i32),
            Expr::Value(Value::SingleQuotedString(s)) => chq_expr_single_quoted_string(s.as_ptr() as *const c_char, s.len() as i64),
            Expr::Value(v) => chq_expr_unsupported_value(c(&format!("{:?}", v)).as_ptr()),
            Expr::Identifier(id) => chq_expr_identifier(c(&id.value).as_ptr()),
            Expr::CompoundIdentifier(ids) => {
                let owned: Vec<CString> = ids.iter().map(|i| c(&i.value)).collect();
                let ptrs: Vec<*const c_char> = owned.iter().map(|s| s.as_ptr()).collect();
                chq_expr_compound_identifier(ptrs.as_ptr(), ptrs.len() as i32)
            }
            other => chq_expr_unsupported(c(&format!("{:?}", other)).as_ptr()),
        }
    }
}

struct Aliases {
    _strings: Vec<Vec<CString>>,
    _ptrs: Vec<Vec<*const c_char>>,
    lists: Vec<chq_alias_list>,
}
fn lower_aliases(table_aliases: &Vec<Vec<String>>) -> Aliases {
    let strings: Vec<Vec<CString>> = table_aliases.iter().map(|l| l.iter().map(|a| CString::new(a.as_str()).unwrap()).collect()).collect();
    let ptrs: Vec<Vec<*const c_char>> = strings.iter().map(|l| l.iter().map(|s| s.as_ptr()).collect()).collect();
    let lists = ptrs.iter().map(|p| chq_alias_list { aliases: p.as_ptr(), n: p.len() as i32 }).collect();
    Aliases { _strings: strings, _ptrs: ptrs, lists }
}

fn export(rec: &RecordBatch) -> Result<(ArrowDeviceArray, FFI_ArrowSchema)> {
    let sa: StructArray = rec.clone().into();
    let (array, schema) = to_ffi(&sa.to_data())?;
    Ok((ArrowDeviceArray { array, device_id: -1, device_type: ARROW_DEVICE_CPU, sync_event: std::ptr::null_mut(), reserved: [0; 3] }, schema))
}
fn import(out: ArrowDeviceArray, schema: FFI_ArrowSchema) -> Result<RecordBatch> {
    let data = unsafe { from_ffi(out.array, &schema)? };
    Ok(RecordBatch::from(StructArray::from(data)))
}

/// record_utils::filter_record on the GPU (filter_record.rs:21-39)
pub fn filter_record(ctx: &GpuContext, rec: Arc<RecordBatch>, table_aliases: &Vec<Vec<String>>, expr: &Expr) -> Result<RecordBatch> {
    let (in_arr, in_schema) = export(&rec)?;
    let al = lower_aliases(table_aliases);
    let ta = chq_table_aliases { columns: al.lists.as_ptr(), n_columns: al.lists.len() as i32 };
    let e = lower_expr(expr);
    let mut out: ArrowDeviceArray = unsafe { std::mem::zeroed() };
    let mut out_schema = FFI_ArrowSchema::empty();
    let rc = unsafe { chq_filter_record(ctx.0, &in_arr, &in_schema, &ta, e, ARROW_DEVICE_CPU, &mut out, &mut out_schema) };
    unsafe { chq_expr_free(e) };
    if rc != 0 {
        return Err(ctx.err(rc));
    }
    import(out, out_schema)
}

/// `recs.iter().map(|r| filter_record(r, ..))` in one library call: every record the exchange has queued for this
/// operator instance is filtered by ONE kernel launch (chq_filter_records); results come back one per input, in
/// order, so each keeps its input record_id (filter_task.rs:109).
pub fn filter_records(ctx: &GpuContext, recs: &[Arc<RecordBatch>], table_aliases: &Vec<Vec<String>>, expr: &Expr) -> Result<Vec<RecordBatch>> {
    if recs.is_empty() {
        return Ok(vec![]);
    }
    let exported: Vec<(ArrowDeviceArray, FFI_ArrowSchema)> = recs.iter().map(|r| export(r)).collect::<Result<_>>()?;
    let ptrs: Vec<*const ArrowDeviceArray> = exported.iter().map(|(a, _)| a as *const ArrowDeviceArray).collect();
    let al = lower_aliases(table_aliases);
    let ta = chq_table_aliases { columns: al.lists.as_ptr(), n_columns: al.lists.len() as i32 };
    let e = lower_expr(expr);
    let n = recs.len();
    let mut outs: Vec<ArrowDeviceArray> = (0..n).map(|_| unsafe { std::mem::zeroed() }).collect();
    let mut schemas: Vec<FFI_ArrowSchema> = (0..n).map(|_| FFI_ArrowSchema::empty()).collect();
    let rc = unsafe {
        chq_filter_records(ctx.0, n as i32, ptrs.as_ptr(), &exported[0].1, &ta, e, ARROW_DEVICE_CPU, outs.as_mut_ptr(), schemas.as_mut_ptr())
    };
    unsafe { chq_expr_free(e) };
    if rc != 0 {
        return Err(ctx.err(rc)); // the error of the earliest failing batch; nothing was returned
    }
    outs.into_iter().zip(schemas.into_iter()).map(|(a, s)| import(a, s)).collect()
}

/// record_utils::project_record on the GPU (record_projection.rs:16-76)
pub fn project_record(ctx: &GpuContext, fields: &Vec<SelectItem>, record: Arc<RecordBatch>, table_aliases: &Vec<Vec<String>>) -> Result<RecordBatch> {
    let (in_arr, in_schema) = export(&record)?;
    let al = lower_aliases(table_aliases);
    let ta = chq_table_aliases { columns: al.lists.as_ptr(), n_columns: al.lists.len() as i32 };
    let mut exprs: Vec<*mut chq_expr> = Vec::new();
    let mut aliases: Vec<CString> = Vec::new();
    let mut items: Vec<chq_select_item> = Vec::new();
    for f in fields {
        match f {
            SelectItem::Wildcard(_) => items.push(chq_select_item { kind: 0, expr: std::ptr::null(), alias: std::ptr::null() }),
            SelectItem::QualifiedWildcard(_, _) => items.push(chq_select_item { kind: 1, expr: std::ptr::null(), alias: std::ptr::null() }),
            SelectItem::UnnamedExpr(e) => { let p = lower_expr(e); exprs.push(p); items.push(chq_select_item { kind: 2, expr: p, alias: std::ptr::null() }) }
            SelectItem::ExprWithAlias { expr, alias } => {
                let p = lower_expr(expr); exprs.push(p);
                aliases.push(CString::new(alias.value.as_str()).unwrap());
                items.push(chq_select_item { kind: 3, expr: p, alias: aliases.last().unwrap().as_ptr() })
            }
        }
    }
    let mut out: ArrowDeviceArray = unsafe { std::mem::zeroed() };
    let mut out_schema = FFI_ArrowSchema::empty();
    let rc = unsafe { chq_project_record(ctx.0, items.as_ptr(), items.len() as i32, &in_arr, &in_schema, &ta, ARROW_DEVICE_CPU, &mut out, &mut out_schema) };
    for p in exprs { unsafe { chq_expr_free(p) } }
    if rc != 0 {
        return Err(ctx.err(rc));
    }
    import(out, out_schema)
}
