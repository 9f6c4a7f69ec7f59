/*
 * chq_oracle.c -- CPU ORACLE (test infrastructure, NOT product code). See chq_oracle.h.
 *
 * Restates, in plain C, what the reference computes on its filter / projection hot path:
 *   compute_value      record_utils/compute_value.rs:57-344
 *   get_common_type    record_utils/compute_value.rs:350-431
 *   cast_to_common_type record_utils/compute_value.rs:433-461
 *   filter_record      record_utils/filter_record.rs:21-39
 *   project_record     record_utils/record_projection.rs:16-76
 * and the arrow-rs 53 kernels those call (arrow-arith numeric::{add,mul,div,rem}, boolean::{and,or};
 * arrow-ord cmp::{eq,neq,lt,lt_eq,gt,gt_eq}; arrow-cast cast; arrow-select filter_record_batch).
 * arrow-rs is an un-vendored third-party dependency (Cargo.toml:42, arrow = "53.1"): its published
 * semantics are restated here and pinned by the reference's own test vectors
 * (record_utils/test_*.rs -> tests/golden/reference_cases.json).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math  (no FMA contraction: arrow-rs
 * evaluates one IEEE operation per element per kernel).
 */
#define _POSIX_C_SOURCE 200809L
#include "chq_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>   /* strcasecmp */
#include <time.h>
#include <stdarg.h>

/* ------------------------------------------------------------------ structures */
struct oc_array {
  int type;
  int subtype;                /* temporal / decimal types: DataType parameters as an id (equal ids <=> equal DataTypes) */
  int64_t length;
  int64_t null_count;
  const void* values;         /* fixed: typed values; bool: bitmap; utf8: int32 offsets */
  int64_t bit_offset;         /* bool only */
  const uint8_t* data;        /* utf8 bytes */
  const uint8_t* validity;    /* NULL = all valid */
  int64_t validity_bit_offset;
  int owned;                  /* buffers malloc'ed by the oracle */
};

struct oc_batch {
  int ncols;
  int64_t nrows;
  char** names;
  int* nullable;
  oc_array** cols;
  int* cols_owned;            /* array struct owned by batch */
  char*** aliases;            /* aliases[i][k] */
  int* nalias;
  int naliases_vec;           /* length of the table_aliases vec (may be < ncols) */
};

enum { OE_NESTED, OE_BINARY, OE_NUMBER, OE_BOOLEAN, OE_STRING, OE_VALUE_OTHER, OE_IDENT, OE_COMPOUND, OE_OTHER };
struct oc_expr {
  int kind;
  int op;
  char* text;       /* number text / identifier / string bytes / description */
  int64_t text_len;
  int flag;         /* is_long / bool value */
  char** parts; int nparts;
  oc_expr *l, *r;
};

typedef struct { oc_array* arr; int is_scalar; int borrowed; } datum;

static const int TYPE_WIDTH[OC_NTYPES] = {0, 1, 2, 4, 8, 1, 2, 4, 8, 2, 4, 8, 0, 4, 8, 4, 8, 8, 8, 16};
static const char* TYPE_NAME[OC_NTYPES] = {"Boolean", "Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16",
                                           "UInt32", "UInt64", "Float16", "Float32", "Float64", "Utf8",
                                           "Date32", "Date64", "Time32", "Time64", "Timestamp", "Duration", "Decimal128"};

static int fail(char* err, int errlen, int code, const char* fmt, ...) {
  if (err && errlen > 0) {
    va_list ap; va_start(ap, fmt); vsnprintf(err, (size_t)errlen, fmt, ap); va_end(ap);
  }
  return code;
}

static inline int bit_get(const uint8_t* b, int64_t i) { return (b[i >> 3] >> (i & 7)) & 1; }
static inline void bit_set(uint8_t* b, int64_t i) { b[i >> 3] |= (uint8_t)(1u << (i & 7)); }
static inline int arr_valid(const oc_array* a, int64_t i) {
  return a->validity == NULL || bit_get(a->validity, a->validity_bit_offset + i);
}
static inline int arr_bool(const oc_array* a, int64_t i) { return bit_get((const uint8_t*)a->values, a->bit_offset + i); }

/* ------------------------------------------------------------------ expr builders */
static char* dupn(const char* s, int64_t n) { char* p = (char*)malloc((size_t)n + 1); memcpy(p, s, (size_t)n); p[n] = 0; return p; }
static oc_expr* enew(int kind) { oc_expr* e = (oc_expr*)calloc(1, sizeof(oc_expr)); e->kind = kind; return e; }
oc_expr* oc_expr_identifier(const char* name) { oc_expr* e = enew(OE_IDENT); e->text = dupn(name, (int64_t)strlen(name)); return e; }
oc_expr* oc_expr_compound_identifier(const char* const* parts, int nparts) {
  oc_expr* e = enew(OE_COMPOUND); e->nparts = nparts; e->parts = (char**)calloc((size_t)(nparts > 0 ? nparts : 1), sizeof(char*));
  for (int i = 0; i < nparts; ++i) e->parts[i] = dupn(parts[i], (int64_t)strlen(parts[i]));
  return e;
}
oc_expr* oc_expr_number(const char* text, int is_long) { oc_expr* e = enew(OE_NUMBER); e->text = dupn(text, (int64_t)strlen(text)); e->flag = is_long; return e; }
oc_expr* oc_expr_boolean(int v) { oc_expr* e = enew(OE_BOOLEAN); e->flag = v != 0; return e; }
oc_expr* oc_expr_string(const char* bytes, int64_t len) { oc_expr* e = enew(OE_STRING); e->text = dupn(bytes, len); e->text_len = len; return e; }
oc_expr* oc_expr_value_other(const char* desc) { oc_expr* e = enew(OE_VALUE_OTHER); e->text = dupn(desc, (int64_t)strlen(desc)); return e; }
oc_expr* oc_expr_binary(int op, const char* op_desc, oc_expr* l, oc_expr* r) {
  oc_expr* e = enew(OE_BINARY); e->op = op; e->l = l; e->r = r; e->text = dupn(op_desc ? op_desc : "", op_desc ? (int64_t)strlen(op_desc) : 0); return e;
}
oc_expr* oc_expr_nested(oc_expr* inner) { oc_expr* e = enew(OE_NESTED); e->l = inner; return e; }
oc_expr* oc_expr_other(const char* desc) { oc_expr* e = enew(OE_OTHER); e->text = dupn(desc, (int64_t)strlen(desc)); return e; }
void oc_expr_free(oc_expr* e) {
  if (!e) return;
  oc_expr_free(e->l); oc_expr_free(e->r);
  for (int i = 0; i < e->nparts; ++i) free(e->parts[i]);
  free(e->parts); free(e->text); free(e);
}

/* ------------------------------------------------------------------ arrays / batches */
static oc_array* arr_new(int type, int64_t len) {
  oc_array* a = (oc_array*)calloc(1, sizeof(oc_array));
  a->type = type; a->length = len; a->owned = 1;
  return a;
}
static oc_array* arr_alloc_fixed(int type, int64_t len) {
  oc_array* a = arr_new(type, len);
  if (type == OC_BOOL) a->values = calloc((size_t)((len + 7) / 8) + 8, 1);
  else a->values = calloc((size_t)(len > 0 ? len : 1) * (size_t)TYPE_WIDTH[type] + 8, 1);
  return a;
}
void oc_array_free(oc_array* a) {
  if (!a) return;
  if (a->owned) { free((void*)a->values); free((void*)a->data); free((void*)a->validity); }
  free(a);
}
static void datum_free(datum* d) { if (d->arr && !d->borrowed) oc_array_free(d->arr); d->arr = NULL; }

oc_batch* oc_batch_new(int ncols, int64_t nrows) {
  oc_batch* b = (oc_batch*)calloc(1, sizeof(oc_batch));
  int n = ncols > 0 ? ncols : 1;
  b->ncols = ncols; b->nrows = nrows;
  b->names = (char**)calloc((size_t)n, sizeof(char*));
  b->nullable = (int*)calloc((size_t)n, sizeof(int));
  b->cols = (oc_array**)calloc((size_t)n, sizeof(oc_array*));
  b->cols_owned = (int*)calloc((size_t)n, sizeof(int));
  b->aliases = (char***)calloc((size_t)n, sizeof(char**));
  b->nalias = (int*)calloc((size_t)n, sizeof(int));
  b->naliases_vec = ncols;
  return b;
}
int oc_batch_set_column(oc_batch* b, int idx, const char* name, int type, int nullable, const void* values,
                        int64_t bit_offset, const uint8_t* data, const uint8_t* validity, int64_t validity_bit_offset) {
  if (idx < 0 || idx >= b->ncols || type < 0 || type >= OC_NTYPES) return OC_ERR_ARROW_INVALID_ARGUMENT;
  oc_array* a = (oc_array*)calloc(1, sizeof(oc_array));
  a->type = type; a->length = b->nrows; a->values = values; a->bit_offset = bit_offset; a->data = data;
  a->validity = validity; a->validity_bit_offset = validity_bit_offset; a->owned = 0;
  if (validity) { int64_t nc = 0; for (int64_t i = 0; i < a->length; ++i) nc += !bit_get(validity, validity_bit_offset + i); a->null_count = nc;
    if (nc == 0) a->validity = NULL; }
  free(b->names[idx]); b->names[idx] = dupn(name, (int64_t)strlen(name));
  b->nullable[idx] = nullable;
  if (b->cols[idx] && b->cols_owned[idx]) oc_array_free(b->cols[idx]);
  b->cols[idx] = a; b->cols_owned[idx] = 1;
  return OC_OK;
}
int oc_batch_set_column_subtype(oc_batch* b, int idx, int subtype) {
  if (idx < 0 || idx >= b->ncols || !b->cols[idx]) return OC_ERR_ARROW_INVALID_ARGUMENT;
  b->cols[idx]->subtype = subtype;
  return OC_OK;
}
int oc_batch_set_aliases(oc_batch* b, int idx, const char* const* aliases, int n) {
  if (idx < 0 || idx >= b->ncols) return OC_ERR_ARROW_INVALID_ARGUMENT;
  for (int k = 0; k < b->nalias[idx]; ++k) free(b->aliases[idx][k]);
  free(b->aliases[idx]);
  b->aliases[idx] = (char**)calloc((size_t)(n > 0 ? n : 1), sizeof(char*));
  for (int k = 0; k < n; ++k) b->aliases[idx][k] = dupn(aliases[k], (int64_t)strlen(aliases[k]));
  b->nalias[idx] = n;
  return OC_OK;
}
void oc_batch_truncate_aliases(oc_batch* b, int n) { b->naliases_vec = n; }
void oc_batch_free(oc_batch* b) {
  if (!b) return;
  int n = b->ncols;
  for (int i = 0; i < n; ++i) {
    free(b->names[i]);
    if (b->cols[i] && b->cols_owned[i]) oc_array_free(b->cols[i]);
    for (int k = 0; k < b->nalias[i]; ++k) free(b->aliases[i][k]);
    free(b->aliases[i]);
  }
  free(b->names); free(b->nullable); free(b->cols); free(b->cols_owned); free(b->aliases); free(b->nalias); free(b);
}
int oc_batch_num_columns(const oc_batch* b) { return b->ncols; }
int64_t oc_batch_num_rows(const oc_batch* b) { return b->nrows; }
const char* oc_batch_field_name(const oc_batch* b, int idx) { return b->names[idx]; }
int oc_batch_field_nullable(const oc_batch* b, int idx) { return b->nullable[idx]; }
const oc_array* oc_batch_column(const oc_batch* b, int idx) { return b->cols[idx]; }
int oc_array_type(const oc_array* a) { return a->type; }
int oc_array_subtype(const oc_array* a) { return a->subtype; }
int64_t oc_array_length(const oc_array* a) { return a->length; }
int64_t oc_array_null_count(const oc_array* a) { return a->null_count; }
const void* oc_array_values(const oc_array* a) { return a->values; }
int64_t oc_array_bit_offset(const oc_array* a) { return a->bit_offset; }
const uint8_t* oc_array_data(const oc_array* a) { return a->data; }
const uint8_t* oc_array_validity(const oc_array* a) { return a->validity; }
int64_t oc_array_validity_bit_offset(const oc_array* a) { return a->validity_bit_offset; }

/* ------------------------------------------------------------------ literal parsing (compute_value.rs:219-265) */
/* Rust <f32 as FromStr>: [+-]? ( "inf"|"infinity"|"nan" | digits [. digits*]? [exp] | . digits+ [exp] ) */
static int rust_float_syntax_ok(const char* s) {
  const char* p = s;
  if (*p == '+' || *p == '-') ++p;
  if (!strcasecmp(p, "inf") || !strcasecmp(p, "infinity") || !strcasecmp(p, "nan")) return 1;
  int nd = 0;
  while (*p >= '0' && *p <= '9') { ++p; ++nd; }
  if (*p == '.') { ++p; while (*p >= '0' && *p <= '9') { ++p; ++nd; } }
  if (nd == 0) return 0;
  if (*p == 'e' || *p == 'E') {
    ++p; if (*p == '+' || *p == '-') ++p;
    int ne = 0; while (*p >= '0' && *p <= '9') { ++p; ++ne; }
    if (ne == 0) return 0;
  }
  return *p == 0;
}
/* Rust <i32/i64 as FromStr>: [+-]? digits+, overflow is an error */
static int rust_parse_int(const char* s, int64_t lo, int64_t hi, int64_t* out) {
  const char* p = s; int neg = 0;
  if (*p == '+') ++p; else if (*p == '-') { neg = 1; ++p; }
  if (!*p) return 0;
  __int128 v = 0;
  for (; *p; ++p) {
    if (*p < '0' || *p > '9') return 0;
    v = v * 10 + (*p - '0');
    if (v > ((__int128)1 << 64)) return 0;
  }
  if (neg) v = -v;
  if (v < lo || v > hi) return 0;
  *out = (int64_t)v; return 1;
}

static int scalar_of(int type, const void* v, datum* out) {
  oc_array* a = arr_alloc_fixed(type, 1);
  memcpy((void*)a->values, v, (size_t)TYPE_WIDTH[type]);
  out->arr = a; out->is_scalar = 1; out->borrowed = 0;
  return OC_OK;
}

/* ------------------------------------------------------------------ arrow-cast: compute::cast */
#define LOADV(T, a, i) (((const T*)(a)->values)[(i)])

/* Float16 <-> f32 as the `half` crate (2.x, arrow's f16) converts: widening is exact and quiets a signalling NaN;
 * narrowing rounds to nearest even, NaN keeps its top payload bits with the quiet bit set.  f16 arithmetic in
 * arrow-arith is half's operator impls: from_f32(to_f32(a) OP to_f32(b)).  (unpinned-by-reference: the reference
 * holds no Float16 vector.) */
static float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = h & 0x7C00u, man = h & 0x03FFu, b;
  if ((h & 0x7FFFu) == 0) b = sign;
  else if (exp == 0x7C00u) b = man == 0 ? (sign | 0x7F800000u) : (sign | 0x7FC00000u | (man << 13));
  else if (exp == 0) {   /* subnormal: normalise */
    int e = 0; while (!(man & 0x0400u)) { man <<= 1; ++e; }
    b = sign | (uint32_t)(127 - 15 - e + 1) << 23 | ((man & 0x03FFu) << 13);
  } else b = sign | (((exp >> 10) + (127 - 15)) << 23) | (man << 13);
  float f; memcpy(&f, &b, 4); return f;
}
static uint16_t f2h(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = x & 0x80000000u, exp = x & 0x7F800000u, man = x & 0x007FFFFFu;
  if (exp == 0x7F800000u) return (uint16_t)((sign >> 16) | 0x7C00u | (man ? 0x0200u : 0) | (man >> 13));
  uint32_t hs = sign >> 16;
  int32_t e = (int32_t)(exp >> 23) - 127 + 15;
  if (e >= 0x1F) return (uint16_t)(hs | 0x7C00u);            /* overflow -> infinity */
  if (e <= 0) {                                              /* subnormal or zero */
    if (14 - e > 24) return (uint16_t)hs;
    man |= 0x00800000u;
    uint32_t hm = man >> (14 - e);
    uint32_t round_bit = 1u << (13 - e);
    if ((man & round_bit) && (man & (3 * round_bit - 1))) ++hm;
    return (uint16_t)(hs | hm);
  }
  uint32_t he = (uint32_t)e << 10, hm = man >> 13, round_bit = 0x00001000u;
  if ((man & round_bit) && (man & (3 * round_bit - 1))) return (uint16_t)((hs | he | hm) + 1);
  return (uint16_t)(hs | he | hm);
}

static double load_as_f64(const oc_array* a, int64_t i) {
  switch (a->type) {
    case OC_F16: return h2f(LOADV(uint16_t, a, i));
    case OC_I8: return LOADV(int8_t, a, i); case OC_I16: return LOADV(int16_t, a, i);
    case OC_I32: return LOADV(int32_t, a, i); case OC_I64: return (double)LOADV(int64_t, a, i);
    case OC_U8: return LOADV(uint8_t, a, i); case OC_U16: return LOADV(uint16_t, a, i);
    case OC_U32: return LOADV(uint32_t, a, i); case OC_U64: return (double)LOADV(uint64_t, a, i);
    case OC_F32: return LOADV(float, a, i); case OC_F64: return LOADV(double, a, i);
    default: return 0;
  }
}
static int is_int_type(int t) { return t >= OC_I8 && t <= OC_U64; }
static int is_signed_int(int t) { return t >= OC_I8 && t <= OC_I64; }
static int is_float_type(int t) { return t == OC_F16 || t == OC_F32 || t == OC_F64; }

static oc_array* copy_validity_from(oc_array* dst, const oc_array* src) {
  if (src->validity && src->null_count > 0) {
    uint8_t* v = (uint8_t*)calloc((size_t)((src->length + 7) / 8) + 8, 1);
    for (int64_t i = 0; i < src->length; ++i) if (arr_valid(src, i)) bit_set(v, i);
    dst->validity = v; dst->validity_bit_offset = 0; dst->null_count = src->null_count;
  }
  return dst;
}

/* Rust's char::is_whitespace (Unicode White_Space) at the start / end of a UTF-8 byte string: length of the
 * white-space character that starts at p (0 = none) / that ends just before p + n */
static int ws_at(const uint8_t* p, int64_t n) {
  if (n >= 1 && ((p[0] >= 0x09 && p[0] <= 0x0D) || p[0] == 0x20)) return 1;
  if (n >= 2 && p[0] == 0xC2 && (p[1] == 0x85 || p[1] == 0xA0)) return 2;
  if (n >= 3) {
    if (p[0] == 0xE1 && p[1] == 0x9A && p[2] == 0x80) return 3;                                   /* U+1680 */
    if (p[0] == 0xE2 && p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF)) return 3;
    if (p[0] == 0xE2 && p[1] == 0x81 && p[2] == 0x9F) return 3;                                   /* U+205F */
    if (p[0] == 0xE3 && p[1] == 0x80 && p[2] == 0x80) return 3;                                   /* U+3000 */
  }
  return 0;
}
static int ws_before(const uint8_t* p, int64_t n) {
  for (int w = 1; w <= 3 && w <= n; ++w) if (ws_at(p + n - w, w) == w) return w;
  return 0;
}
/* 1 / 0 / -1 (not a boolean spelling) */
static int utf8_bool(const uint8_t* p, int64_t n) {
  int w;
  while (n > 0 && (w = ws_at(p, n)) > 0) { p += w; n -= w; }
  while (n > 0 && (w = ws_before(p, n)) > 0) n -= w;
  char buf[8];
  if (n < 1 || n > 5) return -1;
  for (int64_t i = 0; i < n; ++i) buf[i] = (char)((p[i] >= 'A' && p[i] <= 'Z') ? p[i] + 32 : p[i]);
  buf[n] = 0;
  static const char* T[] = {"t", "tr", "tru", "true", "y", "ye", "yes", "on", "1"};
  static const char* F[] = {"f", "fa", "fal", "fals", "false", "n", "no", "of", "off", "0"};
  for (size_t k = 0; k < sizeof T / sizeof *T; ++k) if ((int64_t)strlen(T[k]) == n && !memcmp(buf, T[k], (size_t)n)) return 1;
  for (size_t k = 0; k < sizeof F / sizeof *F; ++k) if ((int64_t)strlen(F[k]) == n && !memcmp(buf, F[k], (size_t)n)) return 0;
  return -1;
}

/* cast numeric -> numeric (as-style; safe mode nulls unrepresentable values, which the widening-only
 * coercion table never produces), numeric -> Boolean (value != 0), same type -> clone. */
static int arrow_cast(const oc_array* a, int to, oc_array** out, char* err, int errlen) {
  int from = a->type;
  int64_t n = a->length;
  if (to == OC_F16 && from != OC_F16) return fail(err, errlen, OC_ERR_NOT_SUPPORTED, "casts to Float16 are never requested by the coercion table");
  if (from == to) {
    oc_array* r;
    if (from == OC_UTF8) {
      r = arr_new(OC_UTF8, n);
      const int32_t* off = (const int32_t*)a->values;
      int32_t* no = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
      int64_t nb = off[n] - off[0];
      uint8_t* nd = (uint8_t*)malloc((size_t)nb + 8);
      memcpy(nd, a->data + off[0], (size_t)nb);
      for (int64_t i = 0; i <= n; ++i) no[i] = off[i] - off[0];
      r->values = no; r->data = nd;
    } else if (from == OC_BOOL) {
      r = arr_alloc_fixed(OC_BOOL, n);
      for (int64_t i = 0; i < n; ++i) if (arr_bool(a, i)) bit_set((uint8_t*)r->values, i);
    } else {
      r = arr_alloc_fixed(from, n);
      memcpy((void*)r->values, a->values, (size_t)n * (size_t)TYPE_WIDTH[from]);
    }
    r->subtype = a->subtype;
    copy_validity_from(r, a);
    *out = r; return OC_OK;
  }
  if (to == OC_BOOL && from == OC_UTF8) {
    /* arrow-cast 53 cast_utf8_to_boolean (reached from compute_value.rs:72-73 / :95-96): value.to_ascii_lowercase().trim()
     * matched against the accepted spellings; anything else is NULL because compute::cast runs with safe = true.
     * (unpinned-by-reference: the reference's tests never put a string under AND / OR.) */
    oc_array* r = arr_alloc_fixed(OC_BOOL, n);
    uint8_t* v = (uint8_t*)calloc((size_t)((n + 7) / 8) + 8, 1);
    int64_t nc = 0;
    const int32_t* off = (const int32_t*)a->values;
    for (int64_t i = 0; i < n; ++i) {
      int val = -1;
      if (arr_valid(a, i)) val = utf8_bool(a->data + off[i], off[i + 1] - off[i]);
      if (val < 0) { ++nc; continue; }
      bit_set(v, i);
      if (val) bit_set((uint8_t*)r->values, i);
    }
    if (nc > 0) { r->validity = v; r->null_count = nc; } else free(v);
    *out = r; return OC_OK;
  }
  if (to == OC_BOOL && (is_int_type(from) || is_float_type(from))) {
    oc_array* r = arr_alloc_fixed(OC_BOOL, n);
    for (int64_t i = 0; i < n; ++i) {
      if (!arr_valid(a, i)) continue;
      int nz;
      switch (from) {
        case OC_I64: nz = LOADV(int64_t, a, i) != 0; break;
        case OC_U64: nz = LOADV(uint64_t, a, i) != 0; break;
        default: nz = load_as_f64(a, i) != 0.0; break;   /* IEEE: NaN != 0 true, -0.0 != 0 false */
      }
      if (nz) bit_set((uint8_t*)r->values, i);
    }
    copy_validity_from(r, a);
    *out = r; return OC_OK;
  }
  if ((is_int_type(from) || is_float_type(from)) && (is_int_type(to) || is_float_type(to))) {
    oc_array* r = arr_alloc_fixed(to, n);
    uint8_t* newnull = NULL; int64_t extra_nulls = 0;
    for (int64_t i = 0; i < n; ++i) {
      if (!arr_valid(a, i)) continue;
      int ok = 1;
      /* go through the widest exact carrier for the source class */
      if (is_float_type(from)) {
        double x = load_as_f64(a, i);
        switch (to) {
          case OC_F32: ((float*)r->values)[i] = (float)x; break;
          case OC_F64: ((double*)r->values)[i] = x; break;
          default: {  /* float -> int: num::cast semantics: None when out of range / NaN */
            double t = trunc(x);
            double lo, hi;
            switch (to) {
              case OC_I8: lo = -128; hi = 127; break; case OC_I16: lo = -32768; hi = 32767; break;
              case OC_I32: lo = -2147483648.0; hi = 2147483647.0; break;
              case OC_I64: lo = -9223372036854775808.0; hi = 9223372036854775807.0; break;
              case OC_U8: lo = 0; hi = 255; break; case OC_U16: lo = 0; hi = 65535; break;
              case OC_U32: lo = 0; hi = 4294967295.0; break; default: lo = 0; hi = 18446744073709551615.0; break;
            }
            if (!(t >= lo && t <= hi) || (to == OC_I64 && t >= 9223372036854775808.0) || (to == OC_U64 && t >= 18446744073709551616.0)) ok = 0;
            else switch (to) {
              case OC_I8: ((int8_t*)r->values)[i] = (int8_t)t; break; case OC_I16: ((int16_t*)r->values)[i] = (int16_t)t; break;
              case OC_I32: ((int32_t*)r->values)[i] = (int32_t)t; break; case OC_I64: ((int64_t*)r->values)[i] = (int64_t)t; break;
              case OC_U8: ((uint8_t*)r->values)[i] = (uint8_t)t; break; case OC_U16: ((uint16_t*)r->values)[i] = (uint16_t)t; break;
              case OC_U32: ((uint32_t*)r->values)[i] = (uint32_t)t; break; default: ((uint64_t*)r->values)[i] = (uint64_t)t; break;
            }
          }
        }
      } else {
        /* integer source: carry in int64 / uint64 */
        int64_t sv = 0; uint64_t uv = 0; int src_signed = is_signed_int(from);
        switch (from) {
          case OC_I8: sv = LOADV(int8_t, a, i); break; case OC_I16: sv = LOADV(int16_t, a, i); break;
          case OC_I32: sv = LOADV(int32_t, a, i); break; case OC_I64: sv = LOADV(int64_t, a, i); break;
          case OC_U8: uv = LOADV(uint8_t, a, i); break; case OC_U16: uv = LOADV(uint16_t, a, i); break;
          case OC_U32: uv = LOADV(uint32_t, a, i); break; default: uv = LOADV(uint64_t, a, i); break;
        }
        switch (to) {
          case OC_F32: ((float*)r->values)[i] = src_signed ? (float)sv : (float)uv; break;   /* round-to-nearest-even */
          case OC_F64: ((double*)r->values)[i] = src_signed ? (double)sv : (double)uv; break;
          default: {
            __int128 v = src_signed ? (__int128)sv : (__int128)uv;
            __int128 lo, hi;
            switch (to) {
              case OC_I8: lo = -128; hi = 127; break; case OC_I16: lo = -32768; hi = 32767; break;
              case OC_I32: lo = INT32_MIN; hi = INT32_MAX; break; case OC_I64: lo = INT64_MIN; hi = INT64_MAX; break;
              case OC_U8: lo = 0; hi = 255; break; case OC_U16: lo = 0; hi = 65535; break;
              case OC_U32: lo = 0; hi = UINT32_MAX; break; default: lo = 0; hi = UINT64_MAX; break;
            }
            if (v < lo || v > hi) ok = 0;
            else switch (to) {
              case OC_I8: ((int8_t*)r->values)[i] = (int8_t)v; break; case OC_I16: ((int16_t*)r->values)[i] = (int16_t)v; break;
              case OC_I32: ((int32_t*)r->values)[i] = (int32_t)v; break; case OC_I64: ((int64_t*)r->values)[i] = (int64_t)v; break;
              case OC_U8: ((uint8_t*)r->values)[i] = (uint8_t)v; break; case OC_U16: ((uint16_t*)r->values)[i] = (uint16_t)v; break;
              case OC_U32: ((uint32_t*)r->values)[i] = (uint32_t)v; break; default: ((uint64_t*)r->values)[i] = (uint64_t)v; break;
            }
          }
        }
      }
      if (!ok) {
        if (!newnull) { newnull = (uint8_t*)malloc((size_t)((n + 7) / 8) + 8); memset(newnull, 0xff, (size_t)((n + 7) / 8) + 8); }
        newnull[i >> 3] &= (uint8_t)~(1u << (i & 7)); ++extra_nulls;
      }
    }
    copy_validity_from(r, a);
    if (newnull) {
      if (r->validity) { for (int64_t i = 0; i < n; ++i) if (!bit_get(newnull, i)) ((uint8_t*)r->validity)[i >> 3] &= (uint8_t)~(1u << (i & 7)); free(newnull); }
      else { for (int64_t i = n; i < ((n + 7) / 8) * 8; ++i) newnull[i >> 3] &= (uint8_t)~(1u << (i & 7)); r->validity = newnull; }
      r->null_count += extra_nulls;
    }
    *out = r; return OC_OK;
  }
  return fail(err, errlen, OC_ERR_ARROW_CAST, "Casting from %s to %s not supported", TYPE_NAME[from], TYPE_NAME[to]);
}

/* ------------------------------------------------------------------ get_common_type (compute_value.rs:350-431) */
static int get_common_type(int l, int r, int* out) {
  if (l == r) { *out = l; return 1; }
#define PAIR(a, b) ((l == (a) && r == (b)) || (l == (b) && r == (a)))
  if (PAIR(OC_I8, OC_I16)) { *out = OC_I16; return 1; }
  if (PAIR(OC_I8, OC_I32) || PAIR(OC_I16, OC_I32)) { *out = OC_I32; return 1; }
  if (PAIR(OC_I8, OC_I64) || PAIR(OC_I16, OC_I64) || PAIR(OC_I32, OC_I64)) { *out = OC_I64; return 1; }
  if (PAIR(OC_U8, OC_U16)) { *out = OC_U16; return 1; }
  if (PAIR(OC_U8, OC_U32) || PAIR(OC_U16, OC_U32)) { *out = OC_U32; return 1; }
  if (PAIR(OC_U8, OC_U64) || PAIR(OC_U16, OC_U64) || PAIR(OC_U32, OC_U64)) { *out = OC_U64; return 1; }
  if (PAIR(OC_U8, OC_I16)) { *out = OC_I16; return 1; }
  if (PAIR(OC_U8, OC_I32) || PAIR(OC_U16, OC_I32)) { *out = OC_I32; return 1; }
  if (PAIR(OC_U8, OC_I64) || PAIR(OC_U16, OC_I64) || PAIR(OC_U32, OC_I64)) { *out = OC_I64; return 1; }
  if (PAIR(OC_F16, OC_F32)) { *out = OC_F32; return 1; }
  if (PAIR(OC_F16, OC_F64) || PAIR(OC_F32, OC_F64)) { *out = OC_F64; return 1; }
  if (PAIR(OC_I8, OC_F32) || PAIR(OC_I16, OC_F32) || PAIR(OC_U8, OC_F32) || PAIR(OC_U16, OC_F32) ||
      PAIR(OC_I32, OC_F32) || PAIR(OC_U32, OC_F32)) { *out = OC_F32; return 1; }
  if (PAIR(OC_I8, OC_F64) || PAIR(OC_I16, OC_F64) || PAIR(OC_U8, OC_F64) || PAIR(OC_U16, OC_F64) ||
      PAIR(OC_I32, OC_F64) || PAIR(OC_U32, OC_F64) || PAIR(OC_I64, OC_F64) || PAIR(OC_U64, OC_F64)) { *out = OC_F64; return 1; }
#undef PAIR
  return 0;
}

/* cast_to_common_type (compute_value.rs:433-461): returns new datums (borrowing when no cast is needed) */
static int cast_to_common_type(const datum* l, const datum* r, datum* lo, datum* ro, char* err, int errlen) {
  int ct;
  if (!get_common_type(l->arr->type, r->arr->type, &ct) ||
      (l->arr->type == r->arr->type && l->arr->subtype != r->arr->subtype))   /* e.g. Timestamp(s) vs Timestamp(ms): different DataTypes */
    return fail(err, errlen, OC_ERR_UNSUPPORTED_TYPE_COERSION, "unsupported type coersion for operation between types %s and %s",
                TYPE_NAME[l->arr->type], TYPE_NAME[l->arr->type]);
  lo->is_scalar = l->is_scalar; ro->is_scalar = r->is_scalar;
  if (l->arr->type != ct) { int rc = arrow_cast(l->arr, ct, &lo->arr, err, errlen); if (rc) return rc; lo->borrowed = 0; }
  else { lo->arr = l->arr; lo->borrowed = 1; }
  if (r->arr->type != ct) { int rc = arrow_cast(r->arr, ct, &ro->arr, err, errlen); if (rc) { datum_free(lo); return rc; } ro->borrowed = 0; }
  else { ro->arr = r->arr; ro->borrowed = 1; }
  return OC_OK;
}

/* ------------------------------------------------------------------ arrow-arith numeric::{add,mul,div,rem} */
/* Output length / null rules of arrow-arith's op!/try_op! macros: scalar operands broadcast, a null
 * scalar gives an all-null result, arrays must have equal length, nulls = union of input nulls, the
 * operation (and so any error) is evaluated on valid slots only. */
static int binary_shape(const datum* l, const datum* r, int64_t* n, char* err, int errlen, const char* what) {
  if (l->is_scalar == r->is_scalar) {
    if (l->arr->length != r->arr->length)
      return fail(err, errlen, OC_ERR_ARROW_INVALID_ARGUMENT, "Cannot %s arrays of different lengths, got %lld vs %lld", what,
                  (long long)l->arr->length, (long long)r->arr->length);
    *n = l->arr->length;
  } else *n = l->is_scalar ? r->arr->length : l->arr->length;
  return OC_OK;
}
static uint8_t* union_validity(const datum* l, const datum* r, int64_t n, int64_t* null_count) {
  int ln = l->arr->null_count > 0, rn = r->arr->null_count > 0;
  *null_count = 0;
  if (!ln && !rn) return NULL;
  uint8_t* v = (uint8_t*)calloc((size_t)((n + 7) / 8) + 8, 1);
  int64_t nc = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t li = (l->is_scalar && !r->is_scalar) ? 0 : i, ri = (r->is_scalar && !l->is_scalar) ? 0 : i;
    if (arr_valid(l->arr, li) && arr_valid(r->arr, ri)) bit_set(v, i); else ++nc;
  }
  *null_count = nc;
  if (nc == 0) { free(v); return NULL; }
  return v;
}

enum { AR_ADD, AR_SUB, AR_MUL, AR_DIV, AR_REM };
/* Extension switch (default off = the reference's behaviour): BinaryOperator::Minus evaluates as arrow-arith
 * numeric::sub instead of returning BinaryOperatorNotImplemented.  Exists only so the tests can check the product's
 * opt-in "enable_minus" option (DEV_NOTES.md:54) against something. */
static int g_extension_minus = 0;
void oc_set_extension_minus(int on) { g_extension_minus = on != 0; }
static const char* AR_SYM[] = {"+", "-", "*", "/", "%"};

#define INT_ARITH(T, WIDE, TMIN, TMAX, IS_SIGNED)                                                            \
  for (int64_t i = 0; i < n; ++i) {                                                                          \
    if (vld && !bit_get(vld, i)) continue;                                                                   \
    T a = ((const T*)l->arr->values)[lb ? 0 : i], b = ((const T*)r->arr->values)[rb ? 0 : i];                 \
    WIDE w;                                                                                                  \
    switch (op) {                                                                                            \
      case AR_ADD: w = (WIDE)a + (WIDE)b; break;                                                             \
      case AR_SUB: w = (WIDE)a - (WIDE)b; break;                                                             \
      case AR_MUL: w = (WIDE)a * (WIDE)b; break;                                                             \
      case AR_DIV: if (b == 0) { rc = fail(err, errlen, OC_ERR_ARROW_DIVIDE_BY_ZERO, "Divide by zero error"); goto done; } \
                   w = (WIDE)a / (WIDE)b; break;                                                             \
      default:     if (b == 0) { rc = fail(err, errlen, OC_ERR_ARROW_DIVIDE_BY_ZERO, "Divide by zero error"); goto done; } \
                   if (IS_SIGNED && (WIDE)a == (WIDE)(TMIN) && (WIDE)b == (WIDE)-1) { w = (WIDE)(TMAX) + 1; break; }      \
                   w = (WIDE)a % (WIDE)b; break;                                                             \
    }                                                                                                        \
    if (w < (WIDE)(TMIN) || w > (WIDE)(TMAX)) {                                                              \
      rc = fail(err, errlen, OC_ERR_ARROW_ARITHMETIC_OVERFLOW, "Overflow happened on: %lld %s %lld", (long long)a, AR_SYM[op], (long long)b); \
      goto done; }                                                                                           \
    ((T*)out->values)[i] = (T)w;                                                                             \
  }

static int arrow_arith(int op, const datum* l, const datum* r, oc_array** res, char* err, int errlen) {
  int t = l->arr->type;
  /* arrow-arith 53 numeric.rs arithmetic_op: Decimal128 and Duration (+) have arithmetic the oracle does not restate;
   * Date / Timestamp / Duration reject + * / % (only subtraction-like forms exist), Time32/64 have no arithmetic. */
  if (t == r->arr->type && (t == OC_DECIMAL128 || (t == OC_DURATION && (op == AR_ADD || op == AR_SUB)) ||
                            ((t == OC_DATE32 || t == OC_DATE64 || t == OC_TIMESTAMP) && op == AR_SUB)))
    return fail(err, errlen, OC_ERR_NOT_SUPPORTED, "%s arithmetic is not restated by the oracle", TYPE_NAME[t]);
  if (t != r->arr->type || !(is_int_type(t) || is_float_type(t)))
    return fail(err, errlen, OC_ERR_ARROW_INVALID_ARGUMENT, "Invalid arithmetic operation: %s %s %s", TYPE_NAME[t], AR_SYM[op], TYPE_NAME[r->arr->type]);
  int64_t n; int rc = binary_shape(l, r, &n, err, errlen, "perform a binary operation on");
  if (rc) return rc;
  int lb = l->is_scalar && !r->is_scalar, rb = r->is_scalar && !l->is_scalar;
  oc_array* out = arr_alloc_fixed(t, n);
  int64_t nc; uint8_t* vld = union_validity(l, r, n, &nc);
  switch (t) {
    case OC_I8: INT_ARITH(int8_t, int32_t, INT8_MIN, INT8_MAX, 1) break;
    case OC_I16: INT_ARITH(int16_t, int32_t, INT16_MIN, INT16_MAX, 1) break;
    case OC_I32: INT_ARITH(int32_t, int64_t, INT32_MIN, INT32_MAX, 1) break;
    case OC_I64: INT_ARITH(int64_t, __int128, INT64_MIN, INT64_MAX, 1) break;
    case OC_U8: INT_ARITH(uint8_t, int32_t, 0, UINT8_MAX, 0) break;
    case OC_U16: INT_ARITH(uint16_t, int64_t, 0, UINT16_MAX, 0) break;
    case OC_U32: INT_ARITH(uint32_t, int64_t, 0, UINT32_MAX, 0) break;
    case OC_U64: INT_ARITH(uint64_t, __int128, 0, UINT64_MAX, 0) break;
    case OC_F16:
      for (int64_t i = 0; i < n; ++i) {
        float a = h2f(((const uint16_t*)l->arr->values)[lb ? 0 : i]), b = h2f(((const uint16_t*)r->arr->values)[rb ? 0 : i]), w;
        switch (op) { case AR_ADD: w = a + b; break; case AR_SUB: w = a - b; break; case AR_MUL: w = a * b; break;
                      case AR_DIV: w = a / b; break; default: w = fmodf(a, b); break; }
        ((uint16_t*)out->values)[i] = f2h(w);
      }
      break;
    case OC_F32:
      for (int64_t i = 0; i < n; ++i) {
        float a = ((const float*)l->arr->values)[lb ? 0 : i], b = ((const float*)r->arr->values)[rb ? 0 : i], w;
        switch (op) { case AR_ADD: w = a + b; break; case AR_SUB: w = a - b; break; case AR_MUL: w = a * b; break;
                      case AR_DIV: w = a / b; break; default: w = fmodf(a, b); break; }
        ((float*)out->values)[i] = w;
      }
      break;
    default:
      for (int64_t i = 0; i < n; ++i) {
        double a = ((const double*)l->arr->values)[lb ? 0 : i], b = ((const double*)r->arr->values)[rb ? 0 : i], w;
        switch (op) { case AR_ADD: w = a + b; break; case AR_SUB: w = a - b; break; case AR_MUL: w = a * b; break;
                      case AR_DIV: w = a / b; break; default: w = fmod(a, b); break; }
        ((double*)out->values)[i] = w;
      }
      break;
  }
done:
  if (rc) { free(vld); oc_array_free(out); return rc; }
  out->validity = vld; out->null_count = nc;
  *res = out; return OC_OK;
}

/* ------------------------------------------------------------------ arrow-ord cmp::* */
enum { CM_EQ, CM_NE, CM_LT, CM_LE, CM_GT, CM_GE };
/* IEEE-754 totalOrder keys (ArrowNativeTypeOp::compare = total_cmp; is_eq = bitwise equality) */
static inline int32_t f32_key(float f) { int32_t b; memcpy(&b, &f, 4); return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1); }
static inline int64_t f64_key(double f) { int64_t b; memcpy(&b, &f, 8); return b ^ (int64_t)(((uint64_t)(b >> 63)) >> 1); }

static inline int cmp_result(int op, int lt, int eq) {
  switch (op) { case CM_EQ: return eq; case CM_NE: return !eq; case CM_LT: return lt; case CM_LE: return lt || eq;
                case CM_GT: return !lt && !eq; default: return !lt; }
}

static int arrow_cmp(int op, const datum* l, const datum* r, oc_array** res, char* err, int errlen) {
  static const char* SYM[] = {"==", "!=", "<", "<=", ">", ">="};
  int t = l->arr->type;
  if (t != r->arr->type)
    return fail(err, errlen, OC_ERR_ARROW_INVALID_ARGUMENT, "Invalid comparison operation: %s %s %s", TYPE_NAME[t], SYM[op], TYPE_NAME[r->arr->type]);
  int64_t n; int rc = binary_shape(l, r, &n, err, errlen, "compare");
  if (rc) return rc;
  int lb = l->is_scalar && !r->is_scalar, rb = r->is_scalar && !l->is_scalar;
  oc_array* out = arr_alloc_fixed(OC_BOOL, n);
  uint8_t* ob = (uint8_t*)out->values;
  for (int64_t i = 0; i < n; ++i) {
    int64_t li = lb ? 0 : i, ri = rb ? 0 : i;
    int lt, eq;
    switch (t) {
#define CASE(TY, CT) case TY: { CT a = LOADV(CT, l->arr, li), b = LOADV(CT, r->arr, ri); lt = a < b; eq = a == b; } break;
      CASE(OC_I8, int8_t) CASE(OC_I16, int16_t) CASE(OC_I32, int32_t) CASE(OC_I64, int64_t)
      CASE(OC_U8, uint8_t) CASE(OC_U16, uint16_t) CASE(OC_U32, uint32_t) CASE(OC_U64, uint64_t)
#undef CASE
      /* temporal / decimal primitives compare their raw values (i32 / i64 / i128 natives) */
      case OC_DATE32: case OC_TIME32: { int32_t a = LOADV(int32_t, l->arr, li), b = LOADV(int32_t, r->arr, ri); lt = a < b; eq = a == b; } break;
      case OC_DATE64: case OC_TIME64: case OC_TIMESTAMP: case OC_DURATION: { int64_t a = LOADV(int64_t, l->arr, li), b = LOADV(int64_t, r->arr, ri); lt = a < b; eq = a == b; } break;
      case OC_DECIMAL128: { __int128 a, b; memcpy(&a, (const uint8_t*)l->arr->values + 16 * li, 16); memcpy(&b, (const uint8_t*)r->arr->values + 16 * ri, 16); lt = a < b; eq = a == b; } break;
      case OC_F16: { int16_t a = LOADV(int16_t, l->arr, li), b = LOADV(int16_t, r->arr, ri);   /* f16::total_cmp */
                     a ^= (int16_t)(((uint16_t)(a >> 15)) >> 1); b ^= (int16_t)(((uint16_t)(b >> 15)) >> 1); lt = a < b; eq = a == b; } break;
      case OC_F32: { int32_t a = f32_key(LOADV(float, l->arr, li)), b = f32_key(LOADV(float, r->arr, ri)); lt = a < b; eq = a == b; } break;
      case OC_F64: { int64_t a = f64_key(LOADV(double, l->arr, li)), b = f64_key(LOADV(double, r->arr, ri)); lt = a < b; eq = a == b; } break;
      case OC_BOOL: { int a = arr_bool(l->arr, li), b = arr_bool(r->arr, ri); lt = a < b; eq = a == b; } break;
      default: {  /* Utf8: byte-lexicographic */
        const int32_t* lo = (const int32_t*)l->arr->values; const int32_t* ro = (const int32_t*)r->arr->values;
        int64_t ll = lo[li + 1] - lo[li], rl = ro[ri + 1] - ro[ri];
        int c = memcmp(l->arr->data + lo[li], r->arr->data + ro[ri], (size_t)(ll < rl ? ll : rl));
        if (c == 0) c = (ll > rl) - (ll < rl);
        lt = c < 0; eq = c == 0;
      }
    }
    if (cmp_result(op, lt, eq)) bit_set(ob, i);
  }
  int64_t nc; out->validity = union_validity(l, r, n, &nc); out->null_count = nc;
  *res = out; return OC_OK;
}

/* ------------------------------------------------------------------ arrow-arith boolean::{and,or} (non-Kleene) */
static int arrow_and_or(int is_and, const oc_array* l, const oc_array* r, oc_array** res, char* err, int errlen) {
  if (l->length != r->length)
    return fail(err, errlen, OC_ERR_ARROW_COMPUTE, "Cannot perform bitwise operation on arrays of different length");
  int64_t n = l->length;
  oc_array* out = arr_alloc_fixed(OC_BOOL, n);
  uint8_t* ob = (uint8_t*)out->values;
  for (int64_t i = 0; i < n; ++i) {
    int a = arr_bool(l, i), b = arr_bool(r, i);
    if (is_and ? (a & b) : (a | b)) bit_set(ob, i);
  }
  datum dl = {(oc_array*)l, 0, 1}, dr = {(oc_array*)r, 0, 1};
  int64_t nc; out->validity = union_validity(&dl, &dr, n, &nc); out->null_count = nc;
  *res = out; return OC_OK;
}

/* ------------------------------------------------------------------ compute_value (compute_value.rs:57-344) */
static int column_by_name(const oc_batch* rec, const char* name) {
  for (int i = 0; i < rec->ncols; ++i) if (!strcmp(rec->names[i], name)) return i;   /* first match */
  return -1;
}

static int compute_value_rec(const oc_batch* rec, const oc_expr* e, datum* out, char* err, int errlen) {
  out->arr = NULL; out->is_scalar = 0; out->borrowed = 0;
  switch (e->kind) {
    case OE_NESTED: return compute_value_rec(rec, e->l, out, err, errlen);
    case OE_BINARY: {
      datum l, r;
      int rc = compute_value_rec(rec, e->l, &l, err, errlen);
      if (rc) return rc;
      rc = compute_value_rec(rec, e->r, &r, err, errlen);
      if (rc) { datum_free(&l); return rc; }
      if (e->op == OC_OP_AND || e->op == OC_OP_OR) {
        /* compute_value.rs:71-116: cast both sides to Boolean, then non-Kleene and/or; is_scalar=false */
        oc_array *lbool = NULL, *rbool = NULL;
        if (!rc) rc = arrow_cast(l.arr, OC_BOOL, &lbool, err, errlen);
        if (!rc) rc = arrow_cast(r.arr, OC_BOOL, &rbool, err, errlen);
        if (!rc) rc = arrow_and_or(e->op == OC_OP_AND, lbool, rbool, &out->arr, err, errlen);
        oc_array_free(lbool); oc_array_free(rbool);
        datum_free(&l); datum_free(&r);
        out->is_scalar = 0;
        return rc;
      }
      int arith = -1, cmp = -1;
      switch (e->op) {
        case OC_OP_MINUS:
          /* NOT the reference: it has no Minus arm (compute_value.rs:210-216). Only with oc_set_extension_minus(1), to
           * check the product's opt-in `enable_minus` option: arrow-arith numeric::sub (checked ints, IEEE floats). */
          if (g_extension_minus) { arith = AR_SUB; break; }
          datum_free(&l); datum_free(&r);
          return fail(err, errlen, OC_ERR_BINARY_OPERATOR_NOT_IMPLEMENTED, "binary operator not implemented: %s", e->text);
        case OC_OP_PLUS: arith = AR_ADD; break; case OC_OP_DIVIDE: arith = AR_DIV; break;
        case OC_OP_MULTIPLY: arith = AR_MUL; break; case OC_OP_MODULO: arith = AR_REM; break;
        case OC_OP_EQ: cmp = CM_EQ; break; case OC_OP_NOTEQ: cmp = CM_NE; break; case OC_OP_GT: cmp = CM_GT; break;
        case OC_OP_GTEQ: cmp = CM_GE; break; case OC_OP_LT: cmp = CM_LT; break; case OC_OP_LTEQ: cmp = CM_LE; break;
        default:  /* Minus and everything else: compute_value.rs:210-216 */
          datum_free(&l); datum_free(&r);
          return fail(err, errlen, OC_ERR_BINARY_OPERATOR_NOT_IMPLEMENTED, "binary operator not implemented: %s", e->text);
      }
      datum lc, rcst;
      rc = cast_to_common_type(&l, &r, &lc, &rcst, err, errlen);
      if (rc) { datum_free(&l); datum_free(&r); return rc; }
      if (arith >= 0) rc = arrow_arith(arith, &lc, &rcst, &out->arr, err, errlen);
      else rc = arrow_cmp(cmp, &lc, &rcst, &out->arr, err, errlen);
      out->is_scalar = lc.is_scalar && rcst.is_scalar;   /* ArrayDatum::new_binary_op, compute_value.rs:43-48 */
      datum_free(&lc); datum_free(&rcst); datum_free(&l); datum_free(&r);
      return rc;
    }
    case OE_NUMBER: {
      if (e->flag) return fail(err, errlen, OC_ERR_VALUE_TYPE_NOT_IMPLEMENTED, "value type not implemented: Number(\"%s\", true)", e->text);
      if (strchr(e->text, '.')) {
        if (!rust_float_syntax_ok(e->text)) return fail(err, errlen, OC_ERR_FAILED_TO_PARSE_AS_A_FLOAT, "failed to parse %s as a float", e->text);
        float f = strtof(e->text, NULL);   /* correctly rounded, like Rust's f32::from_str */
        return scalar_of(OC_F32, &f, out);
      }
      int64_t v;
      if (rust_parse_int(e->text, INT32_MIN, INT32_MAX, &v)) { int32_t x = (int32_t)v; return scalar_of(OC_I32, &x, out); }
      if (rust_parse_int(e->text, INT64_MIN, INT64_MAX, &v)) return scalar_of(OC_I64, &v, out);
      return fail(err, errlen, OC_ERR_FAILED_TO_PARSE_AS_AN_INTEGER, "failed to parse %s as an integer", e->text);
    }
    case OE_BOOLEAN: {
      oc_array* a = arr_alloc_fixed(OC_BOOL, 1);
      if (e->flag) bit_set((uint8_t*)a->values, 0);
      out->arr = a; out->is_scalar = 1; return OC_OK;
    }
    case OE_STRING: {
      oc_array* a = arr_new(OC_UTF8, 1);
      int32_t* off = (int32_t*)calloc(2, sizeof(int32_t)); off[1] = (int32_t)e->text_len;
      a->values = off; a->data = (uint8_t*)dupn(e->text, e->text_len);
      out->arr = a; out->is_scalar = 1; return OC_OK;
    }
    case OE_VALUE_OTHER: return fail(err, errlen, OC_ERR_VALUE_TYPE_NOT_IMPLEMENTED, "value type not implemented: %s", e->text);
    case OE_IDENT: {
      int idx = column_by_name(rec, e->text);
      if (idx < 0) return fail(err, errlen, OC_ERR_COLUMN_NOT_FOUND, "column not found: %s", e->text);
      out->arr = rec->cols[idx]; out->borrowed = 1; out->is_scalar = 0; return OC_OK;
    }
    case OE_COMPOUND: {
      if (e->nparts == 1) {
        int idx = column_by_name(rec, e->parts[0]);
        if (idx < 0) return fail(err, errlen, OC_ERR_COLUMN_NOT_FOUND, "column not found: %s", e->parts[0]);
        out->arr = rec->cols[idx]; out->borrowed = 1; return OC_OK;
      }
      if (e->nparts == 2) {
        for (int i = 0; i < rec->ncols; ++i) {
          if (strcmp(rec->names[i], e->parts[1])) continue;
          /* the reference .expect()s the alias vec entry: it panics when table_aliases is short */
          if (i >= rec->naliases_vec) return fail(err, errlen, OC_ERR_ARROW_INVALID_ARGUMENT, "table aliases vec has incorrect length");
          for (int k = 0; k < rec->nalias[i]; ++k)
            if (!strcmp(rec->aliases[i][k], e->parts[0])) { out->arr = rec->cols[i]; out->borrowed = 1; return OC_OK; }
        }
        return fail(err, errlen, OC_ERR_IDENTIFIER_NOT_FOUND, "identifier not found: \"%s.%s\"", e->parts[0], e->parts[1]);
      }
      {
        char buf[512]; size_t p = 0; buf[0] = 0;
        for (int i = 0; i < e->nparts && p < sizeof(buf) - 2; ++i) p += (size_t)snprintf(buf + p, sizeof(buf) - p, "%s%s", i ? "." : "", e->parts[i]);
        return fail(err, errlen, OC_ERR_IDENTIFIER_NOT_FOUND, "identifier not found: \"%s\"", buf);
      }
    }
    default: return fail(err, errlen, OC_ERR_EXPRESSION_TYPE_NOT_IMPLEMENTED, "expression type not implemented: %s", e->text);
  }
}

static oc_array* clone_array(const oc_array* a) { oc_array* r = NULL; char e[8]; arrow_cast(a, a->type, &r, e, 8); return r; }

int oc_compute_value(const oc_batch* rec, const oc_expr* expr, oc_array** out, int* out_is_scalar, char* err, int errlen) {
  datum d; int rc = compute_value_rec(rec, expr, &d, err, errlen);
  if (rc) return rc;
  if (d.borrowed) { *out = clone_array(d.arr); } else *out = d.arr;
  if (out_is_scalar) *out_is_scalar = d.is_scalar;
  return OC_OK;
}

/* ------------------------------------------------------------------ arrow-select filter_record_batch */
static oc_array* filter_array(const oc_array* a, const oc_array* mask, int64_t count) {
  int64_t m = mask->length;   /* rows beyond the mask are dropped (FilterBuilder iterates the predicate) */
  oc_array* r;
  int64_t k = 0;
#define SEL(i) (arr_bool(mask, (i)) && arr_valid(mask, (i)))   /* null mask slot = false (prep_null_mask_filter) */
  if (a->type == OC_UTF8) {
    r = arr_new(OC_UTF8, count);
    const int32_t* off = (const int32_t*)a->values;
    int64_t nb = 0;
    for (int64_t i = 0; i < m; ++i) if (SEL(i)) nb += off[i + 1] - off[i];
    int32_t* no = (int32_t*)calloc((size_t)count + 1, sizeof(int32_t));
    uint8_t* nd = (uint8_t*)malloc((size_t)nb + 8);
    int64_t p = 0;
    for (int64_t i = 0; i < m; ++i) if (SEL(i)) {
      int64_t len = off[i + 1] - off[i];
      memcpy(nd + p, a->data + off[i], (size_t)len); no[k++] = (int32_t)p; p += len;
    }
    no[count] = (int32_t)p;
    r->values = no; r->data = nd;
  } else if (a->type == OC_BOOL) {
    r = arr_alloc_fixed(OC_BOOL, count);
    for (int64_t i = 0; i < m; ++i) if (SEL(i)) { if (arr_bool(a, i)) bit_set((uint8_t*)r->values, k); ++k; }
  } else {
    int w = TYPE_WIDTH[a->type];
    r = arr_alloc_fixed(a->type, count);
    r->subtype = a->subtype;
    const uint8_t* src = (const uint8_t*)a->values; uint8_t* dst = (uint8_t*)r->values;
    for (int64_t i = 0; i < m; ++i) if (SEL(i)) { memcpy(dst + k * w, src + i * w, (size_t)w); ++k; }
  }
  if (a->null_count > 0) {
    uint8_t* v = (uint8_t*)calloc((size_t)((count + 7) / 8) + 8, 1);
    int64_t nc = 0; k = 0;
    for (int64_t i = 0; i < m; ++i) if (SEL(i)) { if (arr_valid(a, i)) bit_set(v, k); else ++nc; ++k; }
    if (nc > 0) { r->validity = v; r->null_count = nc; } else free(v);
  }
#undef SEL
  return r;
}

static void batch_copy_meta(oc_batch* dst, int di, const oc_batch* src, int si) {
  dst->names[di] = dupn(src->names[si], (int64_t)strlen(src->names[si]));
  dst->nullable[di] = src->nullable[si];
}

int oc_filter_record(const oc_batch* rec, const oc_expr* expr, oc_batch** out, char* err, int errlen) {
  datum d; int rc = compute_value_rec(rec, expr, &d, err, errlen);
  if (rc) return rc;
  if (d.arr->type != OC_BOOL) {   /* filter_record.rs:27-35 */
    rc = fail(err, errlen, OC_ERR_CAST_TO_BOOLEAN_ARRAY_FAILED, "cast to boolean array failed for array type: %s", TYPE_NAME[d.arr->type]);
    datum_free(&d); return rc;
  }
  const oc_array* mask = d.arr;
  if (mask->length > rec->nrows && rec->ncols > 0) {
    rc = fail(err, errlen, OC_ERR_ARROW_INVALID_ARGUMENT, "Filter predicate of length %lld is larger than target array of length %lld",
              (long long)mask->length, (long long)rec->nrows);
    datum_free(&d); return rc;
  }
  int64_t count = 0;
  for (int64_t i = 0; i < mask->length; ++i) count += arr_bool(mask, i) && arr_valid(mask, i);
  oc_batch* o = oc_batch_new(rec->ncols, count);
  for (int c = 0; c < rec->ncols; ++c) {
    batch_copy_meta(o, c, rec, c);
    o->cols[c] = filter_array(rec->cols[c], mask, count); o->cols_owned[c] = 1;
  }
  datum_free(&d);
  *out = o; return OC_OK;
}

/* ------------------------------------------------------------------ project_record (record_projection.rs:16-76) */
int oc_project_record(const oc_select_item* items, int nitems, const oc_batch* rec, oc_batch** out, char* err, int errlen) {
  int cap = 0;
  for (int i = 0; i < nitems; ++i) cap += items[i].kind == OC_ITEM_WILDCARD ? rec->ncols : 1;
  oc_batch* o = oc_batch_new(cap, 0);
  int k = 0, rc = OC_OK; size_t unnamed_idx = 0;
  for (int i = 0; i < nitems && !rc; ++i) {
    const oc_select_item* it = &items[i];
    switch (it->kind) {
      case OC_ITEM_WILDCARD:
        for (int c = 0; c < rec->ncols; ++c) { batch_copy_meta(o, k, rec, c); o->cols[k] = clone_array(rec->cols[c]); o->cols_owned[k] = 1; ++k; }
        break;
      case OC_ITEM_QUALIFIED_WILDCARD:
        rc = fail(err, errlen, OC_ERR_PROJECT_NOT_IMPLEMENTED, "not implemented: SelectItem::QualifiedWildcard"); break;
      case OC_ITEM_UNNAMED_EXPR: {
        datum d; rc = compute_value_rec(rec, it->expr, &d, err, errlen);
        if (rc) break;
        char nm[64];
        if (it->expr->kind == OE_IDENT) o->names[k] = dupn(it->expr->text, (int64_t)strlen(it->expr->text));
        else { snprintf(nm, sizeof nm, "unnamed_%zu", unnamed_idx); o->names[k] = dupn(nm, (int64_t)strlen(nm)); }
        o->cols[k] = d.borrowed ? clone_array(d.arr) : d.arr; o->cols_owned[k] = 1;
        o->nullable[k] = o->cols[k]->null_count > 0;   /* Array::is_nullable */
        ++k; ++unnamed_idx;
        break;
      }
      default: {
        datum d; rc = compute_value_rec(rec, it->expr, &d, err, errlen);
        if (rc) break;
        o->names[k] = dupn(it->alias, (int64_t)strlen(it->alias));
        o->cols[k] = d.borrowed ? clone_array(d.arr) : d.arr; o->cols_owned[k] = 1;
        o->nullable[k] = o->cols[k]->null_count > 0;
        ++k;
      }
    }
  }
  o->ncols = k;
  if (!rc) {
    /* RecordBatch::try_new (record_projection.rs:72-73) */
    if (k == 0) rc = fail(err, errlen, OC_ERR_ARROW_INVALID_ARGUMENT, "must either specify a row count or at least one column");
    else {
      int64_t len = o->cols[0]->length;
      for (int c = 0; c < k && !rc; ++c) {
        if (o->cols[c]->length != len) rc = fail(err, errlen, OC_ERR_ARROW_INVALID_ARGUMENT, "all columns in a record batch must have the same length");
        else if (!o->nullable[c] && o->cols[c]->null_count > 0)
          rc = fail(err, errlen, OC_ERR_ARROW_INVALID_ARGUMENT, "Column '%s' is declared as non-nullable but contains null values", o->names[c]);
      }
      o->nrows = len;
    }
  }
  if (rc) { o->ncols = cap; for (int c = k; c < cap; ++c) { o->cols[c] = NULL; } oc_batch_free(o); return rc; }
  *out = o; return OC_OK;
}

/* ------------------------------------------------------------------ batched CPU baseline helpers */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static oc_batch* slice_batch(const oc_batch* rec, int64_t start, int64_t len) {
  oc_batch* b = oc_batch_new(rec->ncols, len);
  for (int c = 0; c < rec->ncols; ++c) {
    const oc_array* a = rec->cols[c];
    oc_array* s = (oc_array*)calloc(1, sizeof(oc_array));
    *s = *a; s->owned = 0; s->length = len;
    if (a->type == OC_BOOL) s->bit_offset = a->bit_offset + start;
    else if (a->type == OC_UTF8) s->values = (const int32_t*)a->values + start;
    else s->values = (const uint8_t*)a->values + start * TYPE_WIDTH[a->type];
    if (a->validity) {
      s->validity_bit_offset = a->validity_bit_offset + start;
      int64_t nc = 0; for (int64_t i = 0; i < len; ++i) nc += !bit_get(a->validity, s->validity_bit_offset + i);
      s->null_count = nc; if (!nc) s->validity = NULL;
    }
    b->names[c] = dupn(rec->names[c], (int64_t)strlen(rec->names[c])); b->nullable[c] = rec->nullable[c];
    b->cols[c] = s; b->cols_owned[c] = 1;
    if (rec->nalias[c]) oc_batch_set_aliases(b, c, (const char* const*)rec->aliases[c], rec->nalias[c]);
  }
  b->naliases_vec = rec->naliases_vec;
  return b;
}

int oc_filter_project_table_batched(const oc_batch* rec, const oc_expr* pred, const oc_select_item* items, int nitems,
                                    int64_t batch_rows, int64_t* rows_out, double* seconds, char* err, int errlen) {
  int64_t total = 0; int rc = OC_OK;
  double t0 = now_s();
  for (int64_t s = 0; s < rec->nrows && !rc; s += batch_rows) {
    int64_t len = rec->nrows - s < batch_rows ? rec->nrows - s : batch_rows;
    oc_batch* in = slice_batch(rec, s, len);
    oc_batch* f = NULL;
    rc = oc_filter_record(in, pred, &f, err, errlen);
    if (!rc && items) { oc_batch* p = NULL; rc = oc_project_record(items, nitems, f, &p, err, errlen); if (!rc) { total += p->nrows; oc_batch_free(p); } }
    else if (!rc) total += f->nrows;
    oc_batch_free(f); oc_batch_free(in);
  }
  if (seconds) *seconds = now_s() - t0;
  if (rows_out) *rows_out = total;
  return rc;
}
int oc_filter_table_batched(const oc_batch* rec, const oc_expr* expr, int64_t batch_rows, int64_t* rows_out, double* seconds, char* err, int errlen) {
  return oc_filter_project_table_batched(rec, expr, NULL, 0, batch_rows, rows_out, seconds, err, errlen);
}
