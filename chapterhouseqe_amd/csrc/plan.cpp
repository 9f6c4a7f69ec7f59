// plan.cpp -- typing and lowering of sqlparser expressions (see plan.hpp for the reference map).
#include "plan.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <strings.h>

namespace chq {

static const char* kTypeNames[T_NTYPES] = {"Boolean", "Int8",    "Int16",   "Int32",   "Int64", "UInt8", "UInt16",
                                           "UInt32",  "UInt64",  "Float16", "Float32", "Float64", "Utf8", "FixedWidth"};
const char* dtype_name(DType t) { return t < T_NTYPES ? kTypeNames[t] : "?"; }
int dtype_width(DType t) {
  static const int w[T_NTYPES] = {0, 1, 2, 4, 8, 1, 2, 4, 8, 2, 4, 8, 0, 0};
  return t < T_NTYPES ? w[t] : 0;
}

static bool is_int(DType t) { return t >= T_I8 && t <= T_U64; }
static bool is_signed(DType t) { return t >= T_I8 && t <= T_I64; }
static bool is_float(DType t) { return t == T_F32 || t == T_F64; }
static bool is_numeric(DType t) { return is_int(t) || is_float(t); }
// (Float16 counts as wide: only the WIDE kernel instantiations carry its code, kernels.hip: f16_widen)
static bool is_wide(DType t) { return t == T_I64 || t == T_U64 || t == T_F64 || t == T_F16; }

[[noreturn]] static void fail(int code, const std::string& msg) { throw ChqError{code, msg}; }

// ---- arrow-cast 53 cast_utf8_to_boolean (reached from RU/compute_value.rs:72-73, :95-96) -------------------------
// value.to_ascii_lowercase().trim() against the accepted spellings; compute::cast runs with safe = true, so anything
// else is NULL.  str::trim removes Unicode White_Space.  (unpinned-by-reference)
static int ws_at(const uint8_t* p, int64_t n) {
  if (n >= 1 && ((p[0] >= 0x09 && p[0] <= 0x0D) || p[0] == 0x20)) return 1;
  if (n >= 2 && p[0] == 0xC2 && (p[1] == 0x85 || p[1] == 0xA0)) return 2;
  if (n >= 3) {
    if (p[0] == 0xE1 && p[1] == 0x9A && p[2] == 0x80) return 3;
    if (p[0] == 0xE2 && p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF)) return 3;
    if (p[0] == 0xE2 && p[1] == 0x81 && p[2] == 0x9F) return 3;
    if (p[0] == 0xE3 && p[1] == 0x80 && p[2] == 0x80) return 3;
  }
  return 0;
}
int utf8_to_bool(const uint8_t* p, int64_t n) {
  int w;
  while (n > 0 && (w = ws_at(p, n)) > 0) { p += w; n -= w; }
  for (bool again = true; again && n > 0;) {
    again = false;
    for (w = 1; w <= 3 && w <= n; ++w) if (ws_at(p + n - w, w) == w) { n -= w; again = true; break; }
  }
  if (n < 1 || n > 5) return -1;
  char b[6];
  for (int64_t i = 0; i < n; ++i) b[i] = (char)((p[i] >= 'A' && p[i] <= 'Z') ? p[i] + 32 : p[i]);
  auto is = [&](const char* w) { return (int64_t)strlen(w) == n && !memcmp(b, w, (size_t)n); };
  if (is("t") || is("tr") || is("tru") || is("true") || is("y") || is("ye") || is("yes") || is("on") || is("1")) return 1;
  if (is("f") || is("fa") || is("fal") || is("fals") || is("false") || is("n") || is("no") || is("of") || is("off") || is("0")) return 0;
  return -1;
}

// Temporal / decimal columns are compared with a column of the SAME DataType only (get_common_type's `left == right`
// arm, RU/compute_value.rs:355); arrow-ord then compares the native values: i32 (Date32, Time32), i64 (Date64, Time64,
// Timestamp, Duration), i128 (Decimal128).
DType opaque_compare_class(const PlanColumn& c) {
  const std::string& f = c.format;
  if (f == "tdD" || f == "tts" || f == "ttm") return T_I32;
  if (f == "tdm" || f == "ttu" || f == "ttn" || f.rfind("ts", 0) == 0 || f.rfind("tD", 0) == 0) return T_I64;
  if (f.rfind("d:", 0) == 0 && c.width == 16) return T_FIXED_OPAQUE;
  return T_NTYPES;
}
// "d:p,s" and "d:p,s,128" name the same DataType
static std::string canonical_format(const std::string& f) {
  if (f.rfind("d:", 0) == 0 && f.size() > 4 && f.compare(f.size() - 4, 4, ",128") == 0) return f.substr(0, f.size() - 4);
  return f;
}

// ---- literal parsing: RU/compute_value.rs:219-265 -------------------------------------------------
// Rust's <f32 as FromStr> grammar: [+-]? (inf|infinity|nan | digits [. digits*]? | . digits+) ([eE][+-]?digits)?
static bool rust_float_syntax(const std::string& s) {
  const char* p = s.c_str();
  if (*p == '+' || *p == '-') ++p;
  if (!strcasecmp(p, "inf") || !strcasecmp(p, "infinity") || !strcasecmp(p, "nan")) return true;
  int nd = 0;
  while (*p >= '0' && *p <= '9') { ++p; ++nd; }
  if (*p == '.') { ++p; while (*p >= '0' && *p <= '9') { ++p; ++nd; } }
  if (!nd) return false;
  if (*p == 'e' || *p == 'E') {
    ++p;
    if (*p == '+' || *p == '-') ++p;
    int ne = 0;
    while (*p >= '0' && *p <= '9') { ++p; ++ne; }
    if (!ne) return false;
  }
  return *p == 0;
}
// Rust's integer FromStr: [+-]? digits+ ; out-of-range is an error
static bool rust_parse_int(const std::string& s, int64_t lo, int64_t hi, int64_t* out) {
  const char* p = s.c_str();
  bool neg = false;
  if (*p == '+') ++p; else if (*p == '-') { neg = true; ++p; }
  if (!*p) return false;
  __int128 v = 0;
  for (; *p; ++p) {
    if (*p < '0' || *p > '9') return false;
    v = v * 10 + (*p - '0');
    if (v > ((__int128)1 << 64)) return false;
  }
  if (neg) v = -v;
  if (v < lo || v > hi) return false;
  *out = (int64_t)v;
  return true;
}

static Scalar scalar_bits(DType t, uint64_t bits) { Scalar s; s.type = t; s.bits = bits; return s; }
static uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float bits_f32(uint64_t b) { uint32_t u = (uint32_t)b; float f; memcpy(&f, &u, 4); return f; }
static uint64_t f64_bits(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static double bits_f64(uint64_t b) { double f; memcpy(&f, &b, 8); return f; }

// ---- get_common_type: RU/compute_value.rs:350-431 --------------------------------------------------
static bool common_type(DType l, DType r, DType* out) {
  if (l == r) { *out = l; return true; }
  auto pair = [&](DType a, DType b) { return (l == a && r == b) || (l == b && r == a); };
  if (pair(T_I8, T_I16)) { *out = T_I16; return true; }
  if (pair(T_I8, T_I32) || pair(T_I16, T_I32)) { *out = T_I32; return true; }
  if (pair(T_I8, T_I64) || pair(T_I16, T_I64) || pair(T_I32, T_I64)) { *out = T_I64; return true; }
  if (pair(T_U8, T_U16)) { *out = T_U16; return true; }
  if (pair(T_U8, T_U32) || pair(T_U16, T_U32)) { *out = T_U32; return true; }
  if (pair(T_U8, T_U64) || pair(T_U16, T_U64) || pair(T_U32, T_U64)) { *out = T_U64; return true; }
  if (pair(T_U8, T_I16)) { *out = T_I16; return true; }
  if (pair(T_U8, T_I32) || pair(T_U16, T_I32)) { *out = T_I32; return true; }
  if (pair(T_U8, T_I64) || pair(T_U16, T_I64) || pair(T_U32, T_I64)) { *out = T_I64; return true; }
  if (pair(T_F16, T_F32)) { *out = T_F32; return true; }
  if (pair(T_F16, T_F64) || pair(T_F32, T_F64)) { *out = T_F64; return true; }
  if (pair(T_I8, T_F32) || pair(T_I16, T_F32) || pair(T_U8, T_F32) || pair(T_U16, T_F32) || pair(T_I32, T_F32) ||
      pair(T_U32, T_F32)) { *out = T_F32; return true; }
  if (pair(T_I8, T_F64) || pair(T_I16, T_F64) || pair(T_U8, T_F64) || pair(T_U16, T_F64) || pair(T_I32, T_F64) ||
      pair(T_U32, T_F64) || pair(T_I64, T_F64) || pair(T_U64, T_F64)) { *out = T_F64; return true; }
  return false;
}

// ---- host evaluation of literal-only nodes (same semantics as the device interpreter) --------------
static void int_range_of(DType t, __int128* lo, __int128* hi) {
  switch (t) {
    case T_I8: *lo = INT8_MIN; *hi = INT8_MAX; break;
    case T_I16: *lo = INT16_MIN; *hi = INT16_MAX; break;
    case T_I32: *lo = INT32_MIN; *hi = INT32_MAX; break;
    case T_I64: *lo = INT64_MIN; *hi = INT64_MAX; break;
    case T_U8: *lo = 0; *hi = UINT8_MAX; break;
    case T_U16: *lo = 0; *hi = UINT16_MAX; break;
    case T_U32: *lo = 0; *hi = UINT32_MAX; break;
    default: *lo = 0; *hi = UINT64_MAX; break;
  }
}
static __int128 as_i128(const Scalar& s) { return is_signed(s.type) ? (__int128)(int64_t)s.bits : (__int128)(uint64_t)s.bits; }

static Scalar null_scalar(DType t) { Scalar s; s.type = t; s.null = true; return s; }

static Scalar cast_scalar(const Scalar& s, DType to) {
  if (s.type == to) return s;
  if (s.null) return null_scalar(to);
  if (to == T_BOOL && s.type == T_UTF8) {
    const int v = utf8_to_bool((const uint8_t*)s.str.data(), (int64_t)s.str.size());
    return v < 0 ? null_scalar(T_BOOL) : scalar_bits(T_BOOL, (uint64_t)v);
  }
  if (to == T_BOOL) {
    bool nz;
    if (s.type == T_F32) nz = bits_f32(s.bits) != 0.0f;
    else if (s.type == T_F64) nz = bits_f64(s.bits) != 0.0;
    else nz = s.bits != 0;
    return scalar_bits(T_BOOL, nz);
  }
  if (is_int(s.type)) {
    __int128 v = as_i128(s);
    if (to == T_F32) return scalar_bits(T_F32, f32_bits(is_signed(s.type) ? (float)(int64_t)v : (float)(uint64_t)v));
    if (to == T_F64) return scalar_bits(T_F64, f64_bits(is_signed(s.type) ? (double)(int64_t)v : (double)(uint64_t)v));
    return scalar_bits(to, (uint64_t)(int64_t)v);   // widening only: value preserved
  }
  if (s.type == T_F32 && to == T_F64) return scalar_bits(T_F64, f64_bits((double)bits_f32(s.bits)));
  fail(CHQ_ERR_ARROW_CAST, std::string("Casting from ") + dtype_name(s.type) + " to " + dtype_name(to) + " not supported");
}

static int32_t f32_key(uint32_t b) { int32_t s = (int32_t)b; return s ^ (int32_t)(((uint32_t)(s >> 31)) >> 1); }
static int64_t f64_key(uint64_t b) { int64_t s = (int64_t)b; return s ^ (int64_t)(((uint64_t)(s >> 63)) >> 1); }

static Scalar eval_arith(int op, const Scalar& a, const Scalar& b) {
  static const char* sym[] = {"+", "-", "*", "/", "%"};
  DType t = a.type;
  if (t == T_F32) {
    float x = bits_f32(a.bits), y = bits_f32(b.bits), w;
    switch (op) { case OP_ADD: w = x + y; break; case OP_SUB: w = x - y; break; case OP_MUL: w = x * y; break;
                  case OP_DIV: w = x / y; break; default: w = fmodf(x, y); break; }
    return scalar_bits(T_F32, f32_bits(w));
  }
  if (t == T_F64) {
    double x = bits_f64(a.bits), y = bits_f64(b.bits), w;
    switch (op) { case OP_ADD: w = x + y; break; case OP_SUB: w = x - y; break; case OP_MUL: w = x * y; break;
                  case OP_DIV: w = x / y; break; default: w = fmod(x, y); break; }
    return scalar_bits(T_F64, f64_bits(w));
  }
  __int128 x = as_i128(a), y = as_i128(b), w, lo, hi;
  int_range_of(t, &lo, &hi);
  switch (op) {
    case OP_ADD: w = x + y; break;
    case OP_SUB: w = x - y; break;
    case OP_MUL: w = x * y; break;
    case OP_DIV: if (y == 0) fail(CHQ_ERR_ARROW_DIVIDE_BY_ZERO, "Divide by zero error"); w = x / y; break;
    default:
      if (y == 0) fail(CHQ_ERR_ARROW_DIVIDE_BY_ZERO, "Divide by zero error");
      if (is_signed(t) && x == lo && y == -1) w = hi + 1; else w = x % y;
      break;
  }
  if (w < lo || w > hi)
    fail(CHQ_ERR_ARROW_ARITHMETIC_OVERFLOW, std::string("Overflow happened on: ") + std::to_string((long long)x) + " " +
                                                sym[op - OP_ADD] + " " + std::to_string((long long)y));
  return scalar_bits(t, (uint64_t)(int64_t)w);
}

static Scalar eval_cmp(int op, const Scalar& a, const Scalar& b) {
  bool lt, eq;
  switch (a.type) {
    case T_F32: { int32_t x = f32_key((uint32_t)a.bits), y = f32_key((uint32_t)b.bits); lt = x < y; eq = x == y; } break;
    case T_F64: { int64_t x = f64_key(a.bits), y = f64_key(b.bits); lt = x < y; eq = x == y; } break;
    case T_UTF8: { int c = a.str.compare(b.str); lt = c < 0; eq = c == 0; } break;   // byte-lexicographic
    default: { __int128 x = as_i128(a), y = as_i128(b); lt = x < y; eq = x == y; } break;
  }
  bool r;
  switch (op) {
    case OP_EQ: r = eq; break; case OP_NE: r = !eq; break; case OP_LT: r = lt; break;
    case OP_LE: r = lt || eq; break; case OP_GT: r = !lt && !eq; break; default: r = !lt; break;
  }
  return scalar_bits(T_BOOL, r);
}

Scalar fold_constant(const TypedExpr& t, int ni) {
  const Node& n = t.at(ni);
  switch (n.kind) {
    case Node::CONST: return n.cval;
    case Node::CAST: return cast_scalar(fold_constant(t, n.l), n.type);
    case Node::TOBOOL: return cast_scalar(fold_constant(t, n.l), T_BOOL);
    // (a NULL operand gives a NULL result; the operation -- and so its errors -- is evaluated on valid slots only)
    case Node::ARITH: { Scalar a = fold_constant(t, n.l), b = fold_constant(t, n.r); return a.null || b.null ? null_scalar(n.type) : eval_arith(n.op, a, b); }
    case Node::CMP: { Scalar a = fold_constant(t, n.l), b = fold_constant(t, n.r); return a.null || b.null ? null_scalar(T_BOOL) : eval_cmp(n.op, a, b); }
    case Node::ANDOR: {
      Scalar a = fold_constant(t, n.l), b = fold_constant(t, n.r);
      if (a.null || b.null) return null_scalar(T_BOOL);   // non-Kleene: validity = union
      return scalar_bits(T_BOOL, n.op == OP_AND ? (a.bits & b.bits & 1) : ((a.bits | b.bits) & 1));
    }
    default: fail(CHQ_ERR_INVALID_HANDLE, "fold_constant on a column");
  }
}

// ---- typing ------------------------------------------------------------------------------------------
namespace {
struct Typer {
  const std::vector<PlanColumn>& cols;
  int64_t nrows;
  bool enable_minus;
  TypedExpr t;
  int order = 0;

  int push(Node n) {
    n.ref_order = order < 250 ? order : 250;
    ++order;
    t.nodes.push_back(n);
    return (int)t.nodes.size() - 1;
  }
  int push_const(const Scalar& s, bool is_scalar) {
    Node n{}; n.kind = Node::CONST; n.type = s.type; n.is_scalar = is_scalar; n.len1 = true; n.cval = s;
    return push(n);
  }
  // fold a freshly built op node whose inputs are all literal-built
  int maybe_fold(int ni) {
    Node& n = t.nodes[ni];
    if (!n.len1 || n.kind == Node::CONST) return ni;
    Scalar v = fold_constant(t, ni);   // may throw the arithmetic error, at the reference's position
    bool is_scalar = n.is_scalar; int ord = n.ref_order;
    Node c{}; c.kind = Node::CONST; c.type = v.type; c.is_scalar = is_scalar; c.len1 = true; c.cval = v; c.ref_order = ord;
    t.nodes[ni] = c;
    return ni;
  }
  int column_by_name(const std::string& name) {
    for (size_t i = 0; i < cols.size(); ++i) if (cols[i].name == name) return (int)i;   // first match
    return -1;
  }
  int col_node(int idx) {
    Node n{}; n.kind = Node::COL; n.type = cols[idx].type; n.is_scalar = false; n.len1 = false; n.col = idx;
    return push(n);
  }
  int cast_to(int ni, DType to) {
    if (t.nodes[ni].type == to) return ni;
    DType from = t.nodes[ni].type;
    Node n{}; n.kind = Node::CAST; n.type = to; n.from = from; n.l = ni;
    n.is_scalar = t.nodes[ni].is_scalar; n.len1 = t.nodes[ni].len1;
    n.ref_order = t.nodes[ni].ref_order;
    t.nodes.push_back(n);
    return maybe_fold((int)t.nodes.size() - 1);
  }
  int to_bool(int ni) {   // compute::cast(x, &DataType::Boolean), RU/compute_value.rs:72-73,95-96
    DType from = t.nodes[ni].type;
    if (from == T_BOOL) return ni;
    if (!is_numeric(from) && from != T_F16 && from != T_UTF8) fail(CHQ_ERR_ARROW_CAST, std::string("Casting from ") + dtype_name(from) + " to Boolean not supported");
    Node n{}; n.kind = Node::TOBOOL; n.type = T_BOOL; n.from = from; n.l = ni;
    n.is_scalar = t.nodes[ni].is_scalar; n.len1 = t.nodes[ni].len1; n.ref_order = t.nodes[ni].ref_order;
    t.nodes.push_back(n);
    return maybe_fold((int)t.nodes.size() - 1);
  }

  int build(const Expr& e) {
    switch (e.kind) {
      case Expr::NESTED: return build(*e.l);
      case Expr::NUMBER: {
        if (e.flag) fail(CHQ_ERR_VALUE_TYPE_NOT_IMPLEMENTED, "value type not implemented: Number(\"" + e.text + "\", true)");
        if (e.text.find('.') != std::string::npos) {
          if (!rust_float_syntax(e.text)) fail(CHQ_ERR_FAILED_TO_PARSE_AS_A_FLOAT, "failed to parse " + e.text + " as a float");
          float f = strtof(e.text.c_str(), nullptr);   // correctly rounded like Rust's f32::from_str
          return push_const(scalar_bits(T_F32, f32_bits(f)), true);
        }
        int64_t v;
        if (rust_parse_int(e.text, INT32_MIN, INT32_MAX, &v)) return push_const(scalar_bits(T_I32, (uint64_t)v), true);
        if (rust_parse_int(e.text, INT64_MIN, INT64_MAX, &v)) return push_const(scalar_bits(T_I64, (uint64_t)v), true);
        fail(CHQ_ERR_FAILED_TO_PARSE_AS_AN_INTEGER, "failed to parse " + e.text + " as an integer");
      }
      case Expr::BOOLEAN: return push_const(scalar_bits(T_BOOL, e.flag ? 1 : 0), true);
      case Expr::STRING: { Scalar s; s.type = T_UTF8; s.str = e.text; return push_const(s, true); }
      case Expr::VALUE_OTHER: fail(CHQ_ERR_VALUE_TYPE_NOT_IMPLEMENTED, "value type not implemented: " + e.text);
      case Expr::IDENT: {
        int idx = column_by_name(e.text);
        if (idx < 0) fail(CHQ_ERR_COLUMN_NOT_FOUND, "column not found: " + e.text);
        return col_node(idx);
      }
      case Expr::COMPOUND: {
        if (e.parts.size() == 1) {
          int idx = column_by_name(e.parts[0]);
          if (idx < 0) fail(CHQ_ERR_COLUMN_NOT_FOUND, "column not found: " + e.parts[0]);
          return col_node(idx);
        }
        std::string joined;
        for (size_t i = 0; i < e.parts.size(); ++i) joined += (i ? "." : "") + e.parts[i];
        if (e.parts.size() == 2) {
          for (size_t i = 0; i < cols.size(); ++i) {
            if (cols[i].name != e.parts[1]) continue;
            // the reference .expect()s the alias entry and panics when table_aliases is too short
            if (!cols[i].alias_entry_present) fail(CHQ_ERR_ARROW_INVALID_ARGUMENT, "table aliases vec has incorrect length");
            for (const auto& a : cols[i].aliases) if (a == e.parts[0]) return col_node((int)i);
          }
        }
        fail(CHQ_ERR_IDENTIFIER_NOT_FOUND, "identifier not found: \"" + joined + "\"");
      }
      case Expr::BINARY: {
        int l = build(*e.l);
        int r = build(*e.r);
        return binary(e, l, r);
      }
      default: fail(CHQ_ERR_EXPRESSION_TYPE_NOT_IMPLEMENTED, "expression type not implemented: " + e.text);
    }
  }

  // length of a datum: 1 for literal-built arrays, nrows for everything else
  bool same_len(const Node& a, const Node& b) const { return a.len1 == b.len1 || nrows == 1; }

  int binary(const Expr& e, int l, int r) {
    if (e.op == CHQ_BINOP_AND || e.op == CHQ_BINOP_OR) {
      // RU/compute_value.rs:71-116: both sides cast to Boolean, then compute::and / compute::or on
      // &BooleanArray (no scalar broadcast), result flagged non-scalar
      l = to_bool(l);
      r = to_bool(r);
      const Node &ln = t.nodes[l], &rn = t.nodes[r];
      if (!same_len(ln, rn)) fail(CHQ_ERR_ARROW_COMPUTE, "Cannot perform bitwise operation on arrays of different length");
      Node n{}; n.kind = Node::ANDOR; n.op = e.op == CHQ_BINOP_AND ? OP_AND : OP_OR; n.type = T_BOOL;
      n.is_scalar = false; n.len1 = ln.len1 && rn.len1; n.l = l; n.r = r;
      return maybe_fold(push(n));
    }
    int aop = -1, cop = -1;
    switch (e.op) {
      case CHQ_BINOP_PLUS: aop = OP_ADD; break;
      case CHQ_BINOP_DIVIDE: aop = OP_DIV; break;
      case CHQ_BINOP_MULTIPLY: aop = OP_MUL; break;
      case CHQ_BINOP_MODULO: aop = OP_REM; break;
      case CHQ_BINOP_MINUS:
        if (enable_minus) { aop = OP_SUB; break; }
        fail(CHQ_ERR_BINARY_OPERATOR_NOT_IMPLEMENTED, "binary operator not implemented: " + (e.text.empty() ? std::string("Minus") : e.text));
      case CHQ_BINOP_EQ: cop = OP_EQ; break;
      case CHQ_BINOP_NOTEQ: cop = OP_NE; break;
      case CHQ_BINOP_GT: cop = OP_GT; break;
      case CHQ_BINOP_GTEQ: cop = OP_GE; break;
      case CHQ_BINOP_LT: cop = OP_LT; break;
      case CHQ_BINOP_LTEQ: cop = OP_LE; break;
      default: fail(CHQ_ERR_BINARY_OPERATOR_NOT_IMPLEMENTED, "binary operator not implemented: " + e.text);
    }
    // cast_to_common_type, RU/compute_value.rs:433-461
    DType lt = t.nodes[l].type, rt = t.nodes[r].type, ct;
    // (a temporal / decimal value is always a plain column: no literal and no operator produces one)
    const bool both_opaque = lt == T_FIXED_OPAQUE && rt == T_FIXED_OPAQUE;
    const bool same_opaque = both_opaque && canonical_format(cols[t.nodes[l].col].format) == canonical_format(cols[t.nodes[r].col].format);
    if (!common_type(lt, rt, &ct) || (both_opaque && !same_opaque))
      fail(CHQ_ERR_UNSUPPORTED_TYPE_COERSION, std::string("unsupported type coersion for operation between types ") +
                                                  dtype_name(lt) + " and " + dtype_name(lt));
    l = cast_to(l, ct);
    r = cast_to(r, ct);
    const Node ln = t.nodes[l], rn = t.nodes[r];
    static const char* asym[] = {"+", "-", "*", "/", "%"};
    static const char* csym[] = {"==", "!=", "<", "<=", ">", ">="};
    if (aop >= 0) {
      if (ct == T_FIXED_OPAQUE) {
        // arrow-arith 53 arithmetic_op: decimals and Duration +/- have arithmetic (not built here); dates and timestamps
        // only subtract; everything else is "Invalid arithmetic operation" like any non-numeric type
        const std::string& f = cols[ln.col].format;
        const bool sub = aop == OP_SUB, addsub = aop == OP_ADD || sub;
        if (f.rfind("d:", 0) == 0 || (f.rfind("tD", 0) == 0 && addsub) || ((f.rfind("td", 0) == 0 || f.rfind("ts", 0) == 0) && sub))
          fail(CHQ_ERR_NOT_SUPPORTED, "arithmetic on '" + f + "' columns is outside this build's scope");
      }
      if (!is_numeric(ct) && ct != T_F16)
        fail(CHQ_ERR_ARROW_INVALID_ARGUMENT, std::string("Invalid arithmetic operation: ") + dtype_name(ct) + " " +
                                                 asym[aop - OP_ADD] + " " + dtype_name(ct));
      if (ln.is_scalar == rn.is_scalar && !same_len(ln, rn))
        fail(CHQ_ERR_ARROW_COMPUTE, "Cannot perform a binary operation on arrays of different length");
    } else {
      if (ct == T_FIXED_OPAQUE) {
        const DType cls = opaque_compare_class(cols[ln.col]);
        if (cls == T_NTYPES)
          fail(CHQ_ERR_NOT_SUPPORTED, "comparison of '" + cols[ln.col].format + "' columns is outside this build's scope");
        if (cls != T_FIXED_OPAQUE) {   // the raw values ARE Int32 / Int64: the same columns, read as integers
          Node a = ln, b = rn; a.type = cls; b.type = cls;
          t.nodes.push_back(a); l = (int)t.nodes.size() - 1;
          t.nodes.push_back(b); r = (int)t.nodes.size() - 1;
          ct = cls;
        }
      }
      if (ln.is_scalar == rn.is_scalar && !same_len(ln, rn))
        fail(CHQ_ERR_ARROW_INVALID_ARGUMENT, std::string("Cannot compare arrays of different lengths (") + csym[cop - OP_EQ] + ")");
    }
    Node n{}; n.kind = aop >= 0 ? Node::ARITH : Node::CMP; n.op = aop >= 0 ? aop : cop;
    n.type = aop >= 0 ? ct : T_BOOL; n.from = ct; n.l = l; n.r = r;
    n.is_scalar = ln.is_scalar && rn.is_scalar;   // ArrayDatum::new_binary_op
    // output length: equal-flag operands -> their length; otherwise the non-scalar side's
    if (ln.is_scalar == rn.is_scalar) n.len1 = ln.len1 && rn.len1; else n.len1 = ln.is_scalar ? rn.len1 : ln.len1;
    return maybe_fold(push(n));
  }
};
}  // namespace

static bool has_checked_arith(const TypedExpr& t, int ni) {
  const Node& n = t.at(ni);
  if (n.kind == Node::ARITH && is_int(n.type) && !n.len1) return true;
  return (n.l >= 0 && has_checked_arith(t, n.l)) || (n.r >= 0 && has_checked_arith(t, n.r));
}

TypedExpr type_expr(const Expr& e, const std::vector<PlanColumn>& cols, int64_t nrows, bool enable_minus) {
  Typer ty{cols, nrows, enable_minus, {}, 0};
  try {
    int root = ty.build(e);
    ty.t.root = root;
  } catch (const ChqError& err) {
    // completed subtrees = nodes nobody points at; keep those that can fail on data, in evaluation order
    std::vector<char> is_child(ty.t.nodes.size(), 0);
    for (const Node& n : ty.t.nodes) { if (n.l >= 0) is_child[n.l] = 1; if (n.r >= 0) is_child[n.r] = 1; }
    std::vector<int> roots;
    for (size_t i = 0; i < ty.t.nodes.size(); ++i)
      if (!is_child[i] && has_checked_arith(ty.t, (int)i)) roots.push_back((int)i);
    if (roots.empty()) throw;
    ty.t.pending_code = err.code; ty.t.pending_msg = err.msg; ty.t.validate_roots = roots; ty.t.root = -1;
  }
  return std::move(ty.t);
}

// ---- lowering ------------------------------------------------------------------------------------------
namespace {
struct Gen {
  const TypedExpr& t;
  const std::vector<PlanColumn>& cols;
  Lowered& out;
  bool bool_used[MAX_BOOL_TEMPS] = {false, false, false, false};
  bool num_used[MAX_NUM_TEMPS] = {false, false};

  int ref_of(int col) {
    for (size_t i = 0; i < out.refs.size(); ++i) if (out.refs[i] == col) return (int)i;
    if ((int)out.refs.size() >= MAX_REFS) fail(CHQ_INTERNAL_PROGRAM_LIMIT, "expression references more than 12 distinct columns");
    out.refs.push_back(col);
    if (is_wide(cols[col].type)) out.wide = true;
    return (int)out.refs.size() - 1;
  }
  void emit(const Instr& in) {
    if ((int)out.prog.size() >= MAX_INSTR) fail(CHQ_INTERNAL_PROGRAM_LIMIT, "expression needs more than 40 instructions");
    out.prog.push_back(in);
  }
  static uint64_t const_bits(const Scalar& s) { return s.bits; }

  // can node be an instruction operand directly?
  bool operandable(int ni) const {
    const Node& n = t.at(ni);
    if (n.type == T_UTF8) return false;
    if (n.kind == Node::CONST) return !n.cval.null;
    if (n.kind == Node::COL) return true;
    if ((n.kind == Node::CAST || n.kind == Node::TOBOOL) && n.from != T_UTF8 && t.at(n.l).kind == Node::COL) return true;
    return false;
  }
  void set_operand(Instr& in, int ni) {
    const Node& n = t.at(ni);
    if (is_wide(n.type)) out.wide = true;
    if (n.kind == Node::CONST) { in.src_kind = SRC_CONST; in.src_type = n.type; in.imm = const_bits(n.cval); return; }
    const Node& c = n.kind == Node::COL ? n : t.at(n.l);
    in.src_kind = SRC_COL; in.src_idx = (uint16_t)ref_of(c.col); in.src_type = c.type;
  }
  int alloc_temp(DType ty) {
    if (ty == T_BOOL) {
      for (int i = 0; i < MAX_BOOL_TEMPS; ++i) if (!bool_used[i]) { bool_used[i] = true; return i; }
      fail(CHQ_INTERNAL_PROGRAM_LIMIT, "expression needs more than 4 boolean temporaries");
    }
    for (int i = 0; i < MAX_NUM_TEMPS; ++i) if (!num_used[i]) { num_used[i] = true; if (i + 1 > out.num_temps) out.num_temps = i + 1; return i; }
    fail(CHQ_INTERNAL_PROGRAM_LIMIT, "expression needs more than 2 numeric temporaries");
  }
  void free_temp(DType ty, int i) { if (ty == T_BOOL) bool_used[i] = false; else num_used[i] = false; }

  void gen(int ni) {
    const Node& n = t.at(ni);
    if (is_wide(n.type)) out.wide = true;
    Instr in{};
    in.ref_order = (uint8_t)n.ref_order;
    switch (n.kind) {
      case Node::COL:
        if (n.type == T_UTF8 || n.type == T_FIXED_OPAQUE)
          fail(CHQ_ERR_NOT_SUPPORTED, std::string("computing on a ") + dtype_name(n.type) + " column is outside this build's scope");
        in.op = OP_LOAD; in.type = n.type; set_operand(in, ni); emit(in);
        return;
      case Node::CONST:
        if (n.type == T_UTF8) fail(CHQ_ERR_NOT_SUPPORTED, "Utf8 scalar in this position is outside this build's scope");
        if (n.cval.null)   // (only met next to a column of a ONE-row batch: the engine makes it a one-row all-null temporary column)
          fail(CHQ_INTERNAL_PROGRAM_LIMIT, "a NULL literal-built value is not a program constant");
        in.op = OP_LOAD; in.type = n.type; set_operand(in, ni); emit(in);
        return;
      case Node::CAST:
      case Node::TOBOOL:
        if (n.from == T_UTF8)   // not a device-program operation: the engine parses the column into a temporary first
          fail(CHQ_INTERNAL_PROGRAM_LIMIT, "Utf8 -> Boolean cast runs as its own kernel");
        if (t.at(n.l).kind == Node::COL) { in.op = OP_LOAD; in.type = n.type; set_operand(in, ni); emit(in); return; }
        gen(n.l);
        in.op = n.kind == Node::CAST ? OP_CAST : OP_TOBOOL; in.type = n.type; in.src_type = n.from; in.src_kind = SRC_NONE;
        emit(in);
        return;
      case Node::CMP:
        if (n.from == T_UTF8) { gen_strcmp(n); return; }
        if (n.from == T_FIXED_OPAQUE) fail(CHQ_INTERNAL_PROGRAM_LIMIT, "Decimal128 comparison runs as its own kernel");
        [[fallthrough]];
      case Node::ARITH:
      case Node::ANDOR: {
        in.op = (uint8_t)n.op;
        in.type = n.kind == Node::ANDOR ? T_BOOL : n.from;   // type the operation is carried out in
        if (n.kind == Node::ARITH) in.type = n.type;
        if (operandable(n.r)) { gen(n.l); set_operand(in, n.r); emit(in); return; }
        if (operandable(n.l)) { gen(n.r); set_operand(in, n.l); in.flags |= IF_REV; emit(in); return; }
        gen(n.l);
        DType lt = t.at(n.l).type;
        int slot = alloc_temp(lt);
        Instr sp{}; sp.op = OP_SPILL; sp.type = lt; sp.src_idx = (uint16_t)slot; sp.src_kind = SRC_NONE; sp.ref_order = in.ref_order;
        emit(sp);
        gen(n.r);
        in.src_kind = SRC_TEMP; in.src_idx = (uint16_t)slot; in.src_type = lt; in.flags |= IF_REV;
        emit(in);
        free_temp(lt, slot);
        return;
      }
    }
  }

  void gen_strcmp(const Node& n) {
    const Node &l = t.at(n.l), &r = t.at(n.r);
    Instr in{};
    in.op = OP_STRCMP; in.type = T_UTF8; in.src_type = T_UTF8; in.src_kind = SRC_COL; in.ref_order = (uint8_t)n.ref_order;
    auto str_idx = [&](const Node& c) {
      if ((int)out.strs.size() >= MAX_CONST_STR) fail(CHQ_INTERNAL_PROGRAM_LIMIT, "more than 4 string literals in one expression");
      out.strs.push_back(c.cval.str);
      return (uint64_t)out.strs.size() - 1;
    };
    if (l.kind == Node::COL && r.kind == Node::CONST) { in.src_idx = (uint16_t)ref_of(l.col); in.imm = ((uint64_t)n.op << 56) | str_idx(r); }
    else if (l.kind == Node::CONST && r.kind == Node::COL) { in.src_idx = (uint16_t)ref_of(r.col); in.imm = ((uint64_t)n.op << 56) | str_idx(l); in.flags |= IF_REV; }
    else if (l.kind == Node::COL && r.kind == Node::COL) { in.src_idx = (uint16_t)ref_of(l.col); in.imm = ((uint64_t)n.op << 56) | (uint64_t)ref_of(r.col); in.flags |= IF_STR_RHS_COL; }
    else fail(CHQ_ERR_NOT_SUPPORTED, "Utf8 comparison operands must be columns or literals");
    emit(in);
  }
};
}  // namespace

void lower_expr(const TypedExpr& t, int node, const std::vector<PlanColumn>& cols, Lowered& out) {
  Gen g{t, cols, out};
  g.gen(node);
}

}  // namespace chq
