// plan.hpp -- host-side expression tree, typing and lowering.
//
// Mirrors what the reference decides on the host for every batch before it touches data:
//   literal typing          RU/compute_value.rs:219-265
//   column / alias lookup   RU/compute_value.rs:266-337
//   binary-op coercion      RU/compute_value.rs:350-461 (get_common_type / cast_to_common_type)
//   scalar flag             RU/compute_value.rs:34-55   (ArrayDatum::new_binary_op)
// and the static checks of the arrow-rs 53 kernels it calls (same-type requirement, length rules).
// The typed tree is then lowered into the accumulator-machine program of device_program.h.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/chq.h"
#include "device_program.h"

namespace chq {

struct ChqError {
  int code;
  std::string msg;
};
// internal: the expression is fine but does not fit ONE device program (instruction / column / temporary limits of
// device_program.h).  The engine reacts by evaluating sub-trees into temporary columns; at the C ABI it never appears.
constexpr int CHQ_INTERNAL_PROGRAM_LIMIT = 1030;

// ---- sqlparser::ast::Expr mirror -----------------------------------------------------------------
struct Expr {
  enum Kind { NESTED, BINARY, NUMBER, BOOLEAN, STRING, VALUE_OTHER, IDENT, COMPOUND, OTHER } kind;
  int op = 0;             // chq_binary_operator
  std::string text;       // number text / identifier / string bytes / debug text
  bool flag = false;      // is_long / boolean value
  std::vector<std::string> parts;
  std::unique_ptr<Expr> l, r;
};

const char* dtype_name(DType t);
int dtype_width(DType t);   // bytes of a fixed-width value (0 for bool / utf8)

// A column as the planner sees it
struct PlanColumn {
  std::string name;
  DType type;
  bool has_nulls;
  std::vector<std::string> aliases;
  bool alias_entry_present;   // table_aliases vec has an entry for this column
  std::string format;         // Arrow C format string: decides DataType equality of temporal / decimal columns
  int width = 0;              // bytes per value of a fixed-width column
};

struct Scalar {
  DType type = T_BOOL;
  uint64_t bits = 0;          // value bits (sign-extended ints, IEEE bits for floats, 0/1 for bool)
  std::string str;            // utf8
  bool null = false;          // only a Utf8 -> Boolean cast of a literal can make one (arrow-cast: bad spelling = NULL)
};

// arrow-cast 53 cast_utf8_to_boolean on one value: 1 / 0, or -1 when the spelling is not a boolean (= NULL, safe cast)
int utf8_to_bool(const uint8_t* bytes, int64_t len);
// how a temporal / decimal column compares with another of the SAME DataType: as I32 / I64 values, as 128-bit integers
// (T_FIXED_OPAQUE returned, width 16), or not at all in this build (T_NTYPES)
DType opaque_compare_class(const PlanColumn& c);

struct Node {
  // (a CMP whose `from` is T_FIXED_OPAQUE compares two Decimal128 columns, a TOBOOL whose `from` is T_UTF8 parses a Utf8
  // column: neither runs inside a device program -- the engine evaluates them into temporary Boolean columns first)
  enum Kind { COL, CONST, ARITH, CMP, ANDOR, CAST, TOBOOL } kind;
  DType type;
  bool is_scalar;   // ArrayDatum.is_scalar
  bool len1;        // array length is 1 regardless of the batch (built from literals only)
  int col = -1;     // COL
  Scalar cval;      // CONST
  int op = 0;       // ARITH: OP_ADD..OP_REM, CMP: OP_EQ..OP_GE, ANDOR: OP_AND/OP_OR
  int l = -1, r = -1;
  int ref_order = 0;
  DType from = T_BOOL;  // CAST / TOBOOL source type
};

struct TypedExpr {
  std::vector<Node> nodes;
  int root = -1;
  // A static error (type / lookup / literal-folding error) met while walking the tree.  The reference would
  // already have evaluated every subtree completed before that point, so a data-dependent error (integer
  // overflow, division by zero) in one of `validate_roots` wins over it; the engine checks them first.
  int pending_code = 0;
  std::string pending_msg;
  std::vector<int> validate_roots;
  const Node& at(int i) const { return nodes[i]; }
};

// Build the typed tree for `e` against the batch columns.  A static error is thrown directly when nothing
// evaluated before it could have failed on data; otherwise it is returned as `pending_*` (see TypedExpr).  `nrows` is needed for the arrow length rules between len-1 arrays and
// columns.  `enable_minus`: accept BinaryOperator::Minus (not implemented by the reference).
TypedExpr type_expr(const Expr& e, const std::vector<PlanColumn>& cols, int64_t nrows, bool enable_minus);

// Evaluate a len-1 (literal-only) subtree on the host with the device kernels' semantics.
// Throws ChqError for arithmetic errors.  `valid` is always true (literals are never null).
Scalar fold_constant(const TypedExpr& t, int node);

// Lowered program (host form; engine patches column pointers in)
struct Lowered {
  std::vector<Instr> prog;
  std::vector<int> refs;              // batch column index per ColRef slot
  std::vector<std::string> strs;      // scalar strings referenced by OP_STRCMP
  int num_temps = 0;                  // numeric temporaries used
  bool wide = false;                  // needs 64-bit value classes
};

// Append code that leaves the value of `node` in the accumulator. Throws CHQ_ERR_NOT_SUPPORTED when
// the expression exceeds the machine's limits (documented in DESIGN.md).
void lower_expr(const TypedExpr& t, int node, const std::vector<PlanColumn>& cols, Lowered& out);

}  // namespace chq
