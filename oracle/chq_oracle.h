/*
 * chq_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the ChapterhouseDB record_utils path
 *   src/handlers/operator_handler/operators/record_utils/{compute_value,filter_record,
 *   record_projection}.rs
 * plus the arrow-rs 53 kernels that path calls (arrow = "53.1", Cargo.toml:42 -- a third-party
 * crate that is NOT present under /root/reference; its published semantics are restated here and
 * pinned by the reference's own unit-test vectors, see tests/golden/reference_cases.json).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * The product (chapterhouseqe_amd/csrc) never links, loads or calls it.
 *
 * Structure deliberately mirrors the reference: a recursive tree walk that materialises one
 * freshly allocated array per AST node (compute_value.rs:57-344), then a per-column gather
 * (filter_record.rs:37 -> arrow filter_record_batch).
 */
#ifndef CHQ_ORACLE_H
#define CHQ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* data types (subset of arrow DataType that compute_value.rs:350-431 can coerce) */
enum {
  OC_BOOL = 0, OC_I8, OC_I16, OC_I32, OC_I64, OC_U8, OC_U16, OC_U32, OC_U64,
  OC_F16, OC_F32, OC_F64, OC_UTF8,
  /* temporal / decimal primitives: copied by filters and projections, compared with the SAME type only
   * (get_common_type's `_ if left == right` arm, compute_value.rs:355; `subtype` carries unit / time zone /
   * precision+scale so that DataType equality can be decided) */
  OC_DATE32, OC_DATE64, OC_TIME32, OC_TIME64, OC_TIMESTAMP, OC_DURATION, OC_DECIMAL128, OC_NTYPES
};

/* sqlparser::ast::BinaryOperator subset; anything else -> OC_OP_OTHER */
enum {
  OC_OP_AND = 0, OC_OP_OR, OC_OP_PLUS, OC_OP_MINUS, OC_OP_MULTIPLY, OC_OP_DIVIDE, OC_OP_MODULO,
  OC_OP_EQ, OC_OP_NOTEQ, OC_OP_GT, OC_OP_GTEQ, OC_OP_LT, OC_OP_LTEQ, OC_OP_OTHER
};

/* status codes: 1..11 mirror ComputeValueError (compute_value.rs:13-32), FilterRecordError
 * (filter_record.rs:12-15), ProjectRecordError (record_projection.rs:11-14); 20..24 mirror the
 * ArrowError variants the arrow kernels raise. */
enum {
  OC_OK = 0,
  OC_ERR_VALUE_TYPE_NOT_IMPLEMENTED = 1,
  OC_ERR_EXPRESSION_TYPE_NOT_IMPLEMENTED = 2,
  OC_ERR_BINARY_OPERATOR_NOT_IMPLEMENTED = 3,
  OC_ERR_BINARY_OPERATION_CAST_FAILED = 4,
  OC_ERR_FAILED_TO_PARSE_AS_AN_INTEGER = 5,
  OC_ERR_FAILED_TO_PARSE_AS_A_FLOAT = 6,
  OC_ERR_COLUMN_NOT_FOUND = 7,
  OC_ERR_IDENTIFIER_NOT_FOUND = 8,
  OC_ERR_UNSUPPORTED_TYPE_COERSION = 9,
  OC_ERR_CAST_TO_BOOLEAN_ARRAY_FAILED = 10,
  OC_ERR_PROJECT_NOT_IMPLEMENTED = 11,
  OC_ERR_ARROW_ARITHMETIC_OVERFLOW = 20,
  OC_ERR_ARROW_DIVIDE_BY_ZERO = 21,
  OC_ERR_ARROW_INVALID_ARGUMENT = 22,
  OC_ERR_ARROW_COMPUTE = 23,
  OC_ERR_ARROW_CAST = 24,
  OC_ERR_NOT_SUPPORTED = 30
};

typedef struct oc_array oc_array;
typedef struct oc_batch oc_batch;
typedef struct oc_expr oc_expr;

/* ---- expressions (sqlparser::ast::Expr subset handled by compute_value.rs:62-343) ---- */
oc_expr* oc_expr_identifier(const char* name);
oc_expr* oc_expr_compound_identifier(const char* const* parts, int nparts);
oc_expr* oc_expr_number(const char* text, int is_long);
oc_expr* oc_expr_boolean(int v);
oc_expr* oc_expr_string(const char* bytes, int64_t len);
oc_expr* oc_expr_value_other(const char* desc);           /* Value::Null, DoubleQuotedString, ... */
oc_expr* oc_expr_binary(int op, const char* op_desc, oc_expr* left, oc_expr* right); /* takes ownership */
oc_expr* oc_expr_nested(oc_expr* inner);                    /* takes ownership */
oc_expr* oc_expr_other(const char* desc);                   /* UnaryOp, Function, ... */
void oc_expr_free(oc_expr* e);

/* ---- record batches; buffers are BORROWED (caller keeps them alive) ---- */
oc_batch* oc_batch_new(int ncols, int64_t nrows);
/* values: fixed width -> typed values (already advanced to the first row); bool -> LSB-first bitmap
 * read from bit `bit_offset`; utf8 -> int32 offsets[nrows+1] with `data` the byte buffer.
 * validity: NULL or LSB-first bitmap read from bit `validity_bit_offset`. */
int oc_batch_set_column(oc_batch* b, int idx, const char* name, int type, int nullable,
                        const void* values, int64_t bit_offset, const uint8_t* data,
                        const uint8_t* validity, int64_t validity_bit_offset);
/* DataType parameters of a temporal / decimal column as one caller-chosen id (equal ids <=> equal DataTypes) */
int oc_batch_set_column_subtype(oc_batch* b, int idx, int subtype);
int oc_batch_set_aliases(oc_batch* b, int idx, const char* const* aliases, int n);
/* leave aliases shorter than the column count (test_compute_value.rs passes vec![]) */
void oc_batch_truncate_aliases(oc_batch* b, int n);
void oc_batch_free(oc_batch* b);

int oc_batch_num_columns(const oc_batch* b);
int64_t oc_batch_num_rows(const oc_batch* b);
const char* oc_batch_field_name(const oc_batch* b, int idx);
int oc_batch_field_nullable(const oc_batch* b, int idx);
const oc_array* oc_batch_column(const oc_batch* b, int idx);

int oc_array_type(const oc_array* a);
int oc_array_subtype(const oc_array* a);
int64_t oc_array_length(const oc_array* a);
int64_t oc_array_null_count(const oc_array* a);
const void* oc_array_values(const oc_array* a);      /* fixed: values; bool: bitmap; utf8: offsets */
int64_t oc_array_bit_offset(const oc_array* a);
const uint8_t* oc_array_data(const oc_array* a);     /* utf8 bytes */
const uint8_t* oc_array_validity(const oc_array* a); /* NULL or bitmap */
int64_t oc_array_validity_bit_offset(const oc_array* a);
void oc_array_free(oc_array* a);

/* ---- the path ---- */
/* compute_value.rs:57-344 */
int oc_compute_value(const oc_batch* rec, const oc_expr* expr, oc_array** out, int* out_is_scalar,
                     char* err, int errlen);
/* filter_record.rs:21-39 */
int oc_filter_record(const oc_batch* rec, const oc_expr* expr, oc_batch** out, char* err, int errlen);

/* record_projection.rs:16-76; item kinds mirror sqlparser::ast::SelectItem */
enum { OC_ITEM_WILDCARD = 0, OC_ITEM_QUALIFIED_WILDCARD, OC_ITEM_UNNAMED_EXPR, OC_ITEM_EXPR_WITH_ALIAS };
typedef struct { int kind; const oc_expr* expr; const char* alias; } oc_select_item;
int oc_project_record(const oc_select_item* items, int nitems, const oc_batch* rec, oc_batch** out,
                      char* err, int errlen);

/* CPU baseline helper: run filter_record over `rec` cut into reference-sized batches
 * (physical_planner.rs:323: 10 000 rows), one batch at a time on the calling thread, exactly like
 * FilterTask::async_main (filter_task.rs:86-125). Returns status; *rows_out = total rows kept. */
int oc_filter_table_batched(const oc_batch* rec, const oc_expr* expr, int64_t batch_rows,
                            int64_t* rows_out, double* seconds, char* err, int errlen);
/* same for filter followed by project (materialize_files_task.rs:110) */
int oc_filter_project_table_batched(const oc_batch* rec, const oc_expr* pred,
                                    const oc_select_item* items, int nitems, int64_t batch_rows,
                                    int64_t* rows_out, double* seconds, char* err, int errlen);

/* NON-REFERENCE extension switch, default 0.  The reference rejects BinaryOperator::Minus
 * (compute_value.rs:210-216: BinaryOperatorNotImplemented); with 1 the oracle evaluates it as arrow-arith
 * numeric::sub (checked integers, IEEE floats) so the product's opt-in "enable_minus" option has a checker.
 * Process-wide; tests switch it on around the Minus cases only. */
void oc_set_extension_minus(int on);

#ifdef __cplusplus
}
#endif
#endif
