// device_program.h -- the flat accumulator-machine program the host lowers a sqlparser Expr into, and
// the kernel parameter blocks.  Shared by host (plan.cpp / engine.cpp) and device (kernels.hip).
//
// The reference walks the Expr tree once per batch and calls one arrow-rs kernel per node, each
// materialising a full temporary array (RU/compute_value.rs:66-218).  Here the whole tree becomes one
// short program that every wavefront interprets over its 64 x R rows held in registers, so an
// expression costs one pass over its input columns and no temporaries in HBM.
#pragma once
#include <stdint.h>

namespace chq {

typedef unsigned long long u64;   // what HIP's 64-bit atomics take

// Arrow types in scope (the ones RU/compute_value.rs:350-431 can coerce, plus opaque pass-through).
enum DType : uint8_t {
  T_BOOL = 0, T_I8, T_I16, T_I32, T_I64, T_U8, T_U16, T_U32, T_U64, T_F16, T_F32, T_F64, T_UTF8,
  T_FIXED_OPAQUE,  // fixed-width type that is only copied (date/time/timestamp/decimal/...)
  T_NTYPES
};

enum Op : uint8_t {
  OP_LOAD = 0,   // acc = src
  OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_REM,             // acc = acc (op) src   [REV: src (op) acc]
  OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE,           // acc:bool = acc (cmp) src
  OP_AND, OP_OR,                                      // acc:bool = acc (and/or) src, non-Kleene
  OP_CAST,       // acc: src_type -> type
  OP_TOBOOL,     // acc:bool = acc != 0                (compute::cast(x, Boolean))
  OP_SPILL,      // temp[src_idx] = acc (type = acc type)
  OP_STRCMP,     // acc:bool = utf8 column refs[src_idx] (cmp imm>>56) const string #(imm & 0xffff) or column
  OP_STORE,      // projection: out[src_idx] = acc
  OP_NOPS
};

enum SrcKind : uint8_t { SRC_NONE = 0, SRC_COL, SRC_CONST, SRC_TEMP };
enum InstrFlags : uint8_t { IF_REV = 1, IF_STR_RHS_COL = 2 };

struct Instr {           // 16 bytes
  uint8_t op;
  uint8_t type;          // type the operation is carried out in / acc type after LOAD/CAST
  uint8_t src_kind;
  uint8_t src_type;      // stored type of the operand; converted to `type` while fetching
  uint16_t src_idx;      // column-ref index, temp slot, output index
  uint8_t flags;
  uint8_t ref_order;     // position in the reference's evaluation order (error priority)
  uint64_t imm;          // constant bits for SRC_CONST
};

struct ColRef {          // one input column referenced by the program (48 bytes)
  const void* values;    // fixed: values at row 0; bool: bitmap; utf8: int32 offsets at row 0
  const void* validity;  // bitmap or null
  const void* data;      // utf8 bytes
  int64_t validity_bit_offset;
  int64_t bool_bit_offset;
  uint8_t type;
  uint8_t pad[7];
};

struct ConstStr { const uint8_t* bytes; int64_t len; };  // device-resident scalar string

// fixed-width column compaction descriptor
struct OutCol {
  const void* in;        // values at row 0
  void* out;
  uint32_t width;        // 1,2,4,8,16 bytes
  uint32_t pad;
};

// projection output descriptor
struct ProjOut {
  void* values;          // typed values, or bitmap for bool
  u64* validity;    // bitmap (u64 words) or null when the expression cannot produce nulls
  u64* null_count;  // device counter
  uint8_t type;
  uint8_t pad[7];
};

constexpr int MAX_INSTR = 40;
constexpr int MAX_REFS = 12;
constexpr int MAX_OUT = 40;
constexpr int MAX_PROJ = 16;
constexpr int MAX_CONST_STR = 4;
constexpr int MAX_BOOL_TEMPS = 4;
constexpr int MAX_NUM_TEMPS = 2;
// LDS stash slots of the filter kernel's instantiations (tile kinds 0..2; device_program.h is shared with the host,
// which must not ask for more columns than the launched instantiation has slots)
constexpr int STASH_SLOTS_K0 = 1, STASH_SLOTS_K1 = 2, STASH_SLOTS_K2 = 1;
constexpr int MAX_STASH = 2;

// error word: the device keeps the bitwise complement of
//   [63:56] position of the node in the reference's evaluation order, [55:8] row index, [7:0] code
// and combines reports with atomicMax, so the smallest (earliest node, then earliest row) wins -- the reference's
// "first error ends the batch" order -- and "no error" is 0, i.e. the word is cleared by the same memset as the
// rest of the scratch block.
constexpr u64 ERR_NONE = 0ULL;
enum DevErr : uint32_t { DE_OVERFLOW = 1, DE_DIV_ZERO = 2 };

// Specialised predicate shapes that bypass the interpreter loop (host sets `fast_kind` after lowering)
enum FastKind : int32_t {
  FAST_NONE = 0,
  FAST_CMP_CONST = 1,   // refs[0] (Int32 / UInt32 / Float32, no nulls) <cmp prog[1].op> literal prog[1].imm, e.g. value2 > 10.0
  FAST_UOPS = 2,        // every instruction is pre-decoded into ProgramBlock::fast_op / fast_opd (below)
};

// Pre-decoded form of a program over non-null Int32 / UInt32 / Float32 columns (the host fills it when every instruction
// qualifies).  The generic interpreter classifies types, operand kinds and operators with chains of scalar compares --
// measured: about 50 branches and 150 scalar instructions per interpreted instruction and wave, more than the vector
// work of a 16-slot tile -- whereas here ONE jump table selects the operand fetch and one the straight-line loop body.
enum FastOperand : uint8_t { FO_NONE = 0, FO_COL, FO_COL_I2F, FO_COL_U2F, FO_CONST, FO_BTEMP };
enum FastOp : uint8_t {
  FU_LD = 0,                                            // acc = operand (numeric) / boolean temporary
  FU_ADD_F, FU_SUB_F, FU_RSUB_F, FU_MUL_F, FU_DIV_F, FU_RDIV_F,        // Float32; R*: operand (op) acc
  FU_ADD_I, FU_SUB_I, FU_RSUB_I, FU_MUL_I,              // Int32, checked
  FU_ADD_U, FU_SUB_U, FU_RSUB_U, FU_MUL_U,              // UInt32, checked
  FU_DIVP2_I, FU_REMP2_I, FU_DIVP2_U, FU_REMP2_U,       // by a literal power of two (imm)
  FU_EQ, FU_LT_I, FU_GT_I, FU_LT_U, FU_GT_U,            // acc (cmp) operand; FU_EQ is bitwise (all three types)
  FU_LT_F, FU_GT_F,                                     // Float32 totalOrder, both sides keyed
  FU_LT_FKC, FU_GT_FKC,                                 // Float32 totalOrder against a literal with the sign bit set
  FU_AND, FU_OR, FU_SPILL,                              // boolean temporaries
  FU_CVT_I2F, FU_CVT_U2F,                               // acc conversions (OP_CAST)
  FU_STORE,                                             // projection output
  FU_NOPS
};
constexpr uint8_t FU_NEGATE = 0x80;                     // compares: the result is complemented

struct ProgramBlock {
  int32_t n_instr;
  int32_t n_refs;
  int32_t fast_kind;
  int32_t group_bits_at;   // = FilterParams::group_bits_at (kept here so that the interpreter reads it from the kernel
                           // arguments where it is used instead of carrying it -- and a second row pointer -- in SGPRs)
  Instr prog[MAX_INSTR];
  ColRef refs[MAX_REFS];
  ConstStr strs[MAX_CONST_STR];
  uint8_t fast_op[MAX_INSTR];    // FAST_UOPS: FastOp (| FU_NEGATE) per instruction
  uint8_t fast_opd[MAX_INSTR];   // FAST_UOPS: FastOperand per instruction
};

// A Utf8 column filtered inside filter_fused_kernel (single-batch launches only): P(i) adds up the byte lengths of the
// tile's selected rows, a second chained scan (its own status words, resolved by its own wave while wave 0 resolves the
// row scan) gives the tile's first output byte, and C(i) writes the new offsets and copies the bytes.  Byte totals
// stay below 2^31 (int32 offsets), output capacity = the column's input byte span.
constexpr int MAX_FOLD_UTF8 = 2;
struct Utf8Fold {
  const int32_t* in_offsets;   // at row 0
  const uint8_t* in_data;
  int32_t* out_offsets;        // [rows_out + 1]
  uint8_t* out_data;           // null: only the new offsets are written here; utf8_copy_kernel, launched right behind, moves the bytes (long strings)
  u64* status;                 // per tile, like FilterParams::status; zeroed before launch
  u64* total_bytes;            // out
};

struct FilterParams {
  int64_t nrows;          // rows the mask covers
  int64_t tile_begin;     // this launch handles tiles [tile_begin, tile_end); the tail tile of a batch (the only one
  int64_t tile_end;       // that can be partial) runs in its own launch of the PARTIAL instantiation
  u64* status;       // per tile: flag(2) | value(62); zeroed before launch
  uint32_t* ticket;       // zeroed before launch
  u64* total;        // out: number of selected rows
  u64* err;          // ERR_NONE (0) before launch
  u64* sel_mask;     // optional: selection bitmap (one u64 per 64 rows), for follow-up kernels
  u64* grp_base;     // optional: output row index of each 64-row group's first selected row
  int16_t n_out;
  int16_t n_stash;        // program column-refs stash_refs[0..n_stash) keep their raw tile values in LDS between P and C;
  int8_t stash_refs[4];   // (MAX_STASH used) those columns are outs[n_out - n_stash ..] and are copied from LDS instead of fetched again
  int32_t n_utf8;         // Utf8 columns filtered inside this launch (utf8[0..n_utf8), MAX_FOLD_UTF8): see Utf8Fold
  int32_t pad1;
  // Batch-group launch (chq_filter_records): many batches of one schema, one launch, one dense compaction.  Tiles never
  // straddle batches; row `tile` of this table (group_stride words) = { first row of the tile inside its batch, rows of
  // that batch, value pointers of the n_refs program inputs, input pointers of the n_out copied columns }.
  // nullptr = single batch (pointers come from pb.refs / outs, rows are tile * TILE).  PARTIAL instantiation only.
  // Near-uniform groups (the reference's fixed 10 000-row batches) are packed at WAVE granularity instead
  // (group_wpb > 0): every batch owns group_wpb consecutive waves, so a tile's waves may belong to different batches
  // and no lanes idle behind a batch end.  Row b of the table = { rows of batch b, the same pointers }, wave w serves
  // batch w / group_wpb, and the last wave of a batch stores the batch's end offset in group_batch_end[b].
  const u64* group;
  int64_t group_stride;
  int32_t group_wpb;
  int32_t group_nb;
  u64* group_batch_end;
  // Wave-packed groups whose batches carry validity bitmaps or Boolean columns (round 3): word group_bits_at + k of a
  // batch's table row is the bitmap of program column-ref k (its validity, or for a Boolean column its values; 0 = no
  // bitmap: every row valid) and word group_bits_at + n_refs + k the bit position of the batch's row 0 in it.  0 = none.
  // sel_mask / grp_base of such a launch are indexed by wave slot: entry (tile * waves per tile + wave) * R + j.
  int32_t group_bits_at;
  int32_t pad2;
  Utf8Fold utf8[MAX_FOLD_UTF8];
  ProgramBlock pb;
  OutCol outs[MAX_OUT];
};

struct GatherStatusParams {   // dst[i] = value part of status[idx[i]]
  const u64* status;
  const int64_t* idx;
  u64* dst;
  int64_t n;
};

struct ProjectParams {
  int64_t nrows;
  int64_t tile_begin, tile_end;
  u64* err;
  int32_t n_proj;
  int32_t pad;
  ProgramBlock pb;
  ProjOut outs[MAX_PROJ];
};

// filter_project_kernel: predicate + compaction + projection of the surviving rows in one pass (the reference's
// filter -> exchange -> materialize sequence, filter_task.rs:99 + materialize_files_task.rs:110, without the
// intermediate batch).  Only fixed-width, null-free inputs; computed outputs are numeric.
constexpr int MAX_FUSED_COPY = 16;
struct FusedParams {
  int64_t nrows;
  int64_t tile_begin, tile_end;
  u64* status;
  uint32_t* ticket;
  u64* total;
  u64* err;
  int32_t n_proj;         // STORE slots of `proj`
  int32_t n_copy;         // columns passed through unchanged (identifier / wildcard select items)
  ProgramBlock pred;      // the WHERE predicate (all rows)
  ProgramBlock proj;      // the computed select items (evaluated for surviving rows only)
  ProjOut outs[MAX_PROJ];
  OutCol copies[MAX_FUSED_COPY];
};
static_assert(sizeof(FusedParams) <= 4096, "kernel arguments are limited to 4 KiB");

// follow-up kernels driven by the selection bitmap + per-tile inclusive prefixes left in `status`
struct BitCompactParams {   // bool values / validity bitmaps
  int64_t nrows;
  const u64* sel_mask;
  const u64* grp_base;
  const void* in_bits; int64_t in_bit_offset;
  uint32_t* out_bits;       // zero-initialised; bit k = k-th selected row
  u64* zero_count;     // optional: counts selected rows whose bit is 0 (null count)
};

// bit_compact_group_kernel: bit_compact_kernel for a wave-packed group -- out bit k = bit of the k-th selected row of the
// GROUP; chunk c (one wave slot of the main kernel: 64 R rows of one batch) reads batch c / wpb's bitmap.
struct BitCompactGroupParams {
  const u64* sel_mask;       // indexed by wave slot (see FilterParams::group_bits_at)
  const u64* grp_base;
  const u64* table;          // the main launch's group table
  int64_t stride;            // words per batch
  int32_t word_ptr, word_off;   // words of a table row: the bitmap (0: all ones) and the bit position of row 0
  int32_t wpb, nb;
  int32_t rows_per_wave;     // 64 R of the main kernel's tile kind
  int32_t pad;
  uint32_t* out_bits;        // zero-initialised
  u64* zero_count;
};

// typed_ops.hip: operations of the reference's type coverage that are not device-program instructions; each writes a
// temporary Boolean column (value bitmap + validity bitmap, bit i = row i) that the program then reads like any other.
struct Cmp128Params {        // Decimal128 (i128) comparison of two columns, arrow-ord cmp::* on the native values
  const void* a; const void* b;            // values at row 0 (16 bytes per row, any 4-byte alignment)
  const void* a_validity; const void* b_validity;   // bitmaps or null
  int64_t a_validity_offset, b_validity_offset;
  int64_t nrows;
  int32_t op;                // OP_EQ .. OP_GE
  int32_t pad;
  u64* out_bits; u64* out_validity;        // (nrows + 63) / 64 words each
  u64* null_count;           // zero-initialised
};
struct Utf8ToBoolParams {    // arrow-cast cast_utf8_to_boolean: trimmed, case-folded spelling -> true / false / NULL
  const int32_t* offsets;    // at row 0
  const uint8_t* data;
  const void* validity; int64_t validity_offset;
  int64_t nrows;
  u64* out_bits; u64* out_validity;
  u64* null_count;
};

struct SplitBoundsParams {   // split_bounds_kernel: out[i] = number of selected rows before input row starts[i]
  int64_t nrows;
  int64_t n;
  const int64_t* starts;
  const u64* sel_mask;
  const u64* grp_base;
  const u64* total;
  u64* out;
};

struct GatherParams {      // gather_i32_kernel: dst[i] = *src[i]  (batches tiny device->host read-backs into one copy)
  const int32_t* src[16];
  int32_t* dst;
  int32_t n;
  int32_t pad;
};

// concat_*_kernel: the batches of a device-resident group joined into one batch (engine.cpp: concat_device_batches).
// Per-batch tables of nb entries: `src` = buffer address (fixed width: values at row 0; Utf8: int32 offsets at row 0;
// bitmaps: the bitmap, 0 = "all ones"), `aux` = Utf8 data buffer, `bitoff` = first bit of a bitmap; row_at / byte_at are
// exclusive prefix sums (nb + 1 entries) of the batches' rows / Utf8 bytes.
struct ConcatParams {
  int64_t nb;
  const u64* src;
  const u64* aux;
  const int64_t* bitoff;
  const int64_t* row_at;
  const int64_t* byte_at;
  void* dst;           // values / offsets / bitmap (zero-initialised for bitmaps)
  void* dst2;          // Utf8 data
  int32_t* ends;       // gather_ends_kernel: first and last offset of every batch (2 nb entries)
  int32_t width;
  int32_t pad;
};

struct Utf8Params {
  int64_t nrows;
  const u64* sel_mask;
  const u64* grp_base;   // output row index per 64-row group (from the main kernel)
  const int32_t* in_offsets;  // at row 0
  const uint8_t* in_data;
  int32_t* out_offsets;       // [rows_out + 1]
  uint8_t* out_data;
  u64* byte_status;      // look-back state for the byte scan; zeroed before launch
  uint32_t* ticket;
  u64* total_bytes;
  int64_t rows_out;           // number of selected rows (known on the host by now)
};

}  // namespace chq
