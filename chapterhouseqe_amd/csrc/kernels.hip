// kernels.hip -- hand-written gfx950 kernels for the filter / projection record path.
//
// What each kernel replaces in the reference (RU = src/handlers/operator_handler/operators/record_utils):
//   filter_fused_kernel   RU/filter_record.rs:21-39  = compute_value (predicate) + arrow filter_record_batch;
//                         also the one-launch form of the caller's loop over many batches (filter_task.rs:78-126)
//   project_kernel        RU/record_projection.rs:16-76 = one compute_value per SelectItem
//   filter_project_kernel both of the above in one pass (filter_task.rs:99 -> materialize_files_task.rs:110)
//   bit_compact_kernel    arrow-select filter of Boolean values / validity bitmaps
//   utf8_*_kernel         arrow-select filter of Utf8 (offset rebuild + byte copy)
//
// Design (memory-bound path, no MFMA): one wavefront owns 64*R consecutive rows ("column chunk");
// lane l, slot j holds row  w0 + 64*j + l, so every global load/store instruction of a wave touches
// one contiguous 64-element run.  A comparison becomes one 64-bit ballot per slot, boolean combine is
// scalar arithmetic on those ballots, a row's output position is  tile_base + wave_prefix +
// popcount(earlier slots) + mbcnt(ballot).  Tile bases come from a single-pass chained scan
// (decoupled look-back on 8-byte {flag,value} words written/read with agent-scope relaxed atomics),
// software-pipelined so that a tile's aggregate is published one tile ahead of the point where its
// successors need it.  Tiles are handed out by an atomic ticket, so the scan can never wait on a
// workgroup that has not started (no co-residency assumption).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <type_traits>
#include "device_program.h"

namespace chq {


// ------------------------------------------------------------------------------------------------
// tile status words: bits 63..62 flag (0 invalid, 1 aggregate, 2 inclusive prefix), low 62 bits value
// ------------------------------------------------------------------------------------------------
#define ST_AGG (1ULL << 62)
#define ST_INC (2ULL << 62)
#define ST_VAL(x) ((x) & ((1ULL << 62) - 1))
#define ST_FLAG(x) ((x) >> 62)

// ------------------------------------------------------------------------------------------------
// Bounds-checked loads for incomplete waves: a raw buffer resource over the wave's rows [w0, w0 + nact) of one
// column; lanes past the end read 0 from the hardware range check instead of paying for clamped addresses.
// ------------------------------------------------------------------------------------------------
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wave_rows_rsrc(const void* col, int64_t w0, int width, int nact) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)col + w0 * width), 0, nact * width, 0x00020000);
}
// The lane index behind an optimisation barrier.  Every group of range-checked loads takes its own copy right in front of
// the loads: `copy + 64 j` then lives in the loads' basic block, where instruction selection folds the constant into the
// instructions' immediate offset field (one VGPR for all slots).  Without it the compiler shares `lane + 64 j` between all
// load groups of the kernel, computes the sixteen values once at the top and keeps them -- in every element size -- live
// across everything: the 192 spilled VGPRs of round 2's PARTIAL instantiations were these.
__device__ __forceinline__ int opaque_lane(int lane) { asm volatile("" : "+v"(lane)); return lane; }
// the lane index recomputed on the spot (two VALU instructions): nothing stays live for it between two uses
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
template <typename TY> __device__ __forceinline__ TY buf_load(__amdgpu_buffer_rsrc_t r, int elem);
template <> __device__ __forceinline__ uint8_t buf_load<uint8_t>(__amdgpu_buffer_rsrc_t r, int e) { return __builtin_amdgcn_raw_buffer_load_b8(r, e, 0, 0); }
template <> __device__ __forceinline__ int8_t buf_load<int8_t>(__amdgpu_buffer_rsrc_t r, int e) { return (int8_t)__builtin_amdgcn_raw_buffer_load_b8(r, e, 0, 0); }
template <> __device__ __forceinline__ uint16_t buf_load<uint16_t>(__amdgpu_buffer_rsrc_t r, int e) { return __builtin_amdgcn_raw_buffer_load_b16(r, e * 2, 0, 0); }
template <> __device__ __forceinline__ int16_t buf_load<int16_t>(__amdgpu_buffer_rsrc_t r, int e) { return (int16_t)__builtin_amdgcn_raw_buffer_load_b16(r, e * 2, 0, 0); }
template <> __device__ __forceinline__ uint32_t buf_load<uint32_t>(__amdgpu_buffer_rsrc_t r, int e) { return __builtin_amdgcn_raw_buffer_load_b32(r, e * 4, 0, 0); }
template <> __device__ __forceinline__ uint2 buf_load<uint2>(__amdgpu_buffer_rsrc_t r, int e) {
  const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(r, e * 8, 0, 0); return make_uint2(v.x, v.y); }
template <> __device__ __forceinline__ uint4 buf_load<uint4>(__amdgpu_buffer_rsrc_t r, int e) {
  const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(r, e * 16, 0, 0); return make_uint4(v.x, v.y, v.z, v.w); }

__device__ __forceinline__ void st_store(u64* p, u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 st_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 wave_sum(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// tell the compiler a wave-uniform value is uniform (moves it to SGPRs: scalar address math, saddr loads)
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ unsigned lane_rank(u64 m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// Sum the aggregates of the tiles before `tile` back to the nearest inclusive prefix (one wave).
// `first_tile` is the first tile of the chain (its predecessor is a virtual inclusive 0).
__device__ __forceinline__ u64 lookback_exclusive(const u64* status, int64_t tile, int64_t first_tile, int lane) {
  u64 excl = 0;
  int64_t look = tile - 1;
  while (true) {
    int64_t idx = look - lane;
    u64 w = ST_INC;
    if (idx >= first_tile) {
      w = st_load(&status[idx]);
      while (ST_FLAG(w) == 0) { __builtin_amdgcn_s_sleep(1); w = st_load(&status[idx]); }
    }
    u64 incm = __ballot(ST_FLAG(w) == 2);
    if (incm) {
      int first = __builtin_ctzll(incm);
      excl += wave_sum((lane <= first) ? ST_VAL(w) : 0ULL);
      break;
    }
    excl += wave_sum(ST_VAL(w));
    look -= 64;
  }
  return excl;
}

// The wide form: `lookback_issue` loads the status words of the 256 predecessors of `tile` (four per lane, all in flight
// together), `lookback_resolve` sums them back to the nearest inclusive prefix.  Returns false when a needed word was not
// published yet (or no inclusive prefix lies within 256 tiles).
__device__ __forceinline__ void lookback_issue(const u64* status, int64_t tile, int64_t first_tile, int lane, u64 (&w)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t idx = tile - 1 - lane - 64 * k;
    w[k] = idx >= first_tile ? st_load(&status[idx]) : ST_INC;
  }
}
__device__ __forceinline__ bool lookback_resolve(const u64 (&w)[4], int lane, u64& excl_out) {
  u64 excl = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u64 incm = __ballot(ST_FLAG(w[k]) == 2), zerom = __ballot(ST_FLAG(w[k]) == 0);
    if (incm) {
      const int first = __builtin_ctzll(incm);
      if (zerom & ((1ULL << first) - 1ULL)) return false;
      excl += wave_sum((lane <= first) ? ST_VAL(w[k]) : 0ULL);
      excl_out = excl;
      return true;
    }
    if (zerom) return false;
    excl += wave_sum(ST_VAL(w[k]));
  }
  return false;
}

// blocking form of the 256-wide look-back: re-reads the window until it resolves; the narrow loop takes over when the
// nearest inclusive prefix lies further back than the window
__device__ __forceinline__ u64 lookback_exclusive_wide(const u64* status, int64_t tile, int64_t first_tile, int lane) {
  for (int attempt = 0; attempt < 64; ++attempt) {
    u64 w[4];
    lookback_issue(status, tile, first_tile, lane, w);
    u64 excl;
    if (lookback_resolve(w, lane, excl)) return excl;
    bool all_published = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) all_published = all_published && (__ballot(ST_FLAG(w[k]) == 0) == 0);
    if (all_published) break;           // 256 aggregates and no inclusive prefix among them: walk further back
    __builtin_amdgcn_s_sleep(2);
  }
  return lookback_exclusive(status, tile, first_tile, lane);
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// Inclusive prefix sums over the 64 lanes of a wave on the DPP network (no LDS traffic): Kogge-Stone inside each row of
// 16 lanes (row_shr 1, 2, 4, 8; lanes shifted in from outside the row add 0), then lane 15 of rows 0 / 2 is added to
// rows 1 / 3 (row_bcast:15) and lane 31 to rows 2 and 3 (row_bcast:31).
// Two independent scans interleaved: each DPP add reads the register the add two instructions earlier wrote, so one
// s_nop covers the two wait states the hardware wants between a VALU write and a DPP read (9 issue slots per scan
// instead of the 18 of v_mov_dpp + v_add + s_nop the compiler emits for __builtin_amdgcn_update_dpp + add)
__device__ __forceinline__ void wave_incl_scan2(uint32_t& a, uint32_t& b) {
  asm volatile(
      "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "s_nop 0\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "s_nop 0\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_u32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "s_nop 0\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_u32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "s_nop 0\n"
      "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "v_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "s_nop 0\n"
      "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      "v_add_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void copy_rows_wide(const uint8_t* __restrict__ in_data, uint8_t* __restrict__ out_base,
                                               int rs, int rl, int rd, int cnt, int lane);
__device__ __forceinline__ u64 active_mask(int64_t start, int64_t nrows) {
  int64_t rem = nrows - start;
  return rem >= 64 ? ~0ULL : (rem <= 0 ? 0ULL : ((1ULL << rem) - 1ULL));
}

// 64 bits of an LSB-first bitmap starting at bit `bitpos` (wave-uniform), masked by `act`.
__device__ __forceinline__ u64 load_bits64(const void* bitmap, int64_t bitpos, u64 act) {
  if (act == 0) return 0;
  uint64_t abs = ((uint64_t)(uintptr_t)bitmap << 3) + (uint64_t)bitpos;
  const u64* p = (const u64*)(uintptr_t)((abs >> 6) << 3);
  int sh = (int)(abs & 63);
  u64 r = p[0] >> sh;
  int nbits = 64 - __builtin_clzll(act);
  if (sh != 0 && sh + nbits > 64) r |= p[1] << (64 - sh);
  return r & act;
}

// same, for a group that lies completely inside the batch (no masking, no bit counting)
__device__ __forceinline__ u64 load_bits64_full(const void* bitmap, int64_t bitpos) {
  uint64_t abs = ((uint64_t)(uintptr_t)bitmap << 3) + (uint64_t)bitpos;
  const u64* p = (const u64*)(uintptr_t)((abs >> 6) << 3);
  int sh = (int)(abs & 63);
  u64 r = p[0] >> sh;
  if (sh != 0) r |= p[1] << (64 - sh);
  return r;
}

__device__ __forceinline__ int32_t f32_key(uint32_t b) { int32_t s = (int32_t)b; return s ^ (int32_t)(((uint32_t)(s >> 31)) >> 1); }
__device__ __forceinline__ int64_t f64_key(uint64_t b) { int64_t s = (int64_t)b; return s ^ (int64_t)(((uint64_t)(s >> 63)) >> 1); }

__device__ __forceinline__ void report_error(u64* err, uint32_t ref_order, int64_t row, uint32_t code) {
  atomicMax(err, ~(((u64)ref_order << 56) | ((u64)row << 8) | (u64)code));
}

// Float16 values live in the interpreter as the f32 with the same value.  The widening is exact, and NaNs keep their
// payload WITHOUT being quieted (the hardware conversion would set the quiet bit), so the embedding is one-to-one and
// monotone under IEEE totalOrder: the f32 comparison code then gives f16's total order and bitwise equality unchanged.
// A real cast to Float32 / Float64 (arrow-cast through half::f16::to_f32) quiets the NaN: Interp::convert does that.
__device__ __forceinline__ uint32_t f16_widen(uint32_t h) {
  uint32_t w = __float_as_uint(__half2float(__ushort_as_half((unsigned short)h)));
  if ((h & 0x7fffu) > 0x7c00u) w = ((h & 0x8000u) << 16) | 0x7f800000u | ((h & 0x3ffu) << 13);
  return w;
}
// arrow's f16 arithmetic is half's: from_f32(to_f32(a) OP to_f32(b)), round to nearest even
__device__ __forceinline__ uint32_t f16_round(uint32_t f) {
  return __float_as_uint(__half2float(__float2half_rn(__uint_as_float(f))));
}
__device__ __forceinline__ uint16_t f16_bits(uint32_t f) { return __half_as_ushort(__float2half_rn(__uint_as_float(f))); }

// value classes the interpreter computes in
enum VClass { C_I32, C_U32, C_I64, C_U64, C_F32, C_F64, C_BOOL, C_NONE };
__device__ __forceinline__ int vclass(int t) {
  switch (t) {
    case T_I8: case T_I16: case T_I32: return C_I32;
    case T_U8: case T_U16: case T_U32: return C_U32;
    case T_I64: return C_I64; case T_U64: return C_U64;
    case T_F32: return C_F32; case T_F64: return C_F64; case T_BOOL: return C_BOOL;
    default: return C_NONE;
  }
}
// Float16 is computed in f32 (see f16_widen) and rounded back to f16 after every operation -- in the WIDE kernel
// instantiations only (plan.cpp marks Float16 programs wide): the 1024-thread kernels run at their 128-VGPR limit, and
// even an extra case label in their type switches brought spills back
template <bool WIDE>
__device__ __forceinline__ int vclass_of(int t) {
  if constexpr (WIDE) { if (t == T_F16) return C_F32; }
  return vclass(t);
}
__device__ __forceinline__ void int_range(int t, int64_t& lo, int64_t& hi) {
  switch (t) {
    case T_I8: lo = -128; hi = 127; break; case T_I16: lo = -32768; hi = 32767; break;
    case T_I32: lo = INT32_MIN; hi = INT32_MAX; break;
    case T_U8: lo = 0; hi = 255; break; case T_U16: lo = 0; hi = 65535; break;
    default: lo = 0; hi = 4294967295LL; break;
  }
}

#define PACK64(lo, hi) (((uint64_t)(hi) << 32) | (uint64_t)(lo))

// ------------------------------------------------------------------------------------------------
// The interpreter.  One call evaluates the whole program for the 64*R rows of one wave.
// Register state per lane: lo[R] (+ hi[R] when WIDE) = value of the lane's R rows; bitsv / validv =
// boolean value / validity of those rows packed one bit per slot (bit j <-> row w0 + 64 j + lane), so a
// boolean AND/OR over all R rows is one VALU op.  All arrays are indexed with compile-time constants
// only (loops fully unrolled) so they live in VGPRs.  WIDE=false instantiations carry no 64-bit types.
// ------------------------------------------------------------------------------------------------
// PARTIAL=false: every wave has all 64*R rows inside the batch (no clamping / masking code is generated at all)
// NS: predicate input columns (<= 4 bytes wide) whose raw values the filter kernel keeps on chip between its predicate and
// copy phases: they go to the LDS block `stash_lds` ([NS][R][BLOCK]) while the program fetches column-ref `stash_ref[k]`
// (0 for every other user of the interpreter).
template <int BLOCK, int R, bool WIDE, bool PARTIAL, int NS = 0>
struct Interp {
  static constexpr int NW = BLOCK / 64;
  static constexpr int RH = WIDE ? R : 1;
  static constexpr int NSA = NS > 0 ? NS : 1;
  static_assert(R <= 32, "slot flags are packed in 32-bit registers");

  uint32_t lo[R], hi[RH];
  uint32_t bitsv, validv, actv;

  int64_t w0;        // first row of the wave (uniform)
  int64_t nrows;
  int nact;          // active rows in [w0, w0 + 64R) (uniform)
  int lane, wv;
  uint32_t* stash_lds = nullptr;
  int stash_ref[NSA];
  int n_stash = 0;
  const u64* ptr_row = nullptr;   // batch-group launch: value pointer of column-ref k is ptr_row[k] (PARTIAL only)
  // ... and, when the group carries bitmaps (ProgramBlock::group_bits_at = B > 0): ptr_row[B - 1 + k] = validity bitmap of
  // column-ref k in this wave's batch (0: none), ptr_row[B - 1 + n_refs + k] = bit position of the batch's row 0 in its bitmaps

  __device__ __forceinline__ void set_rows(int64_t tile_start, int64_t nrows_, int lane_, int wv_) {
    lane = lane_; wv = wv_; nrows = nrows_;
    w0 = tile_start + (int64_t)wv * 64 * R;
    if constexpr (!PARTIAL) {
      nact = 64 * R;
      actv = R == 32 ? 0xffffffffu : ((1u << R) - 1u);
    } else {
      int64_t rem = nrows - w0;
      nact = rem <= 0 ? 0 : (rem >= 64 * R ? 64 * R : (int)rem);
      // slot j is active for this lane iff 64 j + lane < nact: all slots of the complete 64-row groups, plus -- for the lanes
      // below the remainder -- the slot of the incomplete one (scalar arithmetic and ONE vector compare; sixteen compares of
      // `lane | 64 j` kept sixteen VGPRs alive from the top of the kernel)
      const int full = nact >> 6, part = nact & 63;
      const uint32_t below = full >= 32 ? 0xffffffffu : ((1u << full) - 1u);
      actv = below | ((lane < part) ? (1u << (full & 31)) : 0u);
    }
  }
  __device__ __forceinline__ void stash_put(int ref_idx, const uint32_t (&v)[R]) {
    if constexpr (NS > 0) {
      if (n_stash == 0) return;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        if (ref_idx == stash_ref[k]) {
#pragma unroll
          for (int j = 0; j < R; ++j) stash_lds[(k * R + j) * BLOCK + wv * 64 + lane] = v[j];
        }
      }
    }
  }
  __device__ __forceinline__ u64 group_act(int j) const { return active_mask(w0 + 64 * j, nrows); }
  // per-lane flags (bit j) from a bitmap
  __device__ __forceinline__ uint32_t fetch_flags(const void* bitmap, int64_t bit_offset) const {
    uint32_t f = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      u64 wd;
      if constexpr (PARTIAL) wd = load_bits64(bitmap, bit_offset + w0 + 64 * j, group_act(j));
      else wd = load_bits64_full(bitmap, bit_offset + w0 + 64 * j);
      f |= (uint32_t)((wd >> lane) & 1) << j;
    }
    return f;
  }

  // ---- operand fetch ---------------------------------------------------------------------------
  // (Every case of the type switches below ends with a store to l[]: the compiler merges the cases' final stores into one
  // block, and when one of them went to h[] instead the merged store got a pointer phi (l or h), which keeps the whole
  // interpreter object -- every register array -- in scratch memory: the 152 bytes of scratch of round 2's WIDE kernels.)
  template <bool FULL>
  __device__ __forceinline__ void fetch_values(const ColRef& cr, const void* values, uint32_t (&l)[R], uint32_t (&h)[RH]) {
    struct { int type; const void* values; } c{cr.type, values};
    if constexpr (WIDE) {
      if (c.type == T_F16) {
        if constexpr (FULL) {
          const uint16_t* p = (const uint16_t*)c.values + w0;
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = f16_widen(p[j * 64 + lane]);
        } else {
          const int ol = opaque_lane(lane);
          const auto r = wave_rows_rsrc(c.values, w0, 2, nact);
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = f16_widen(buf_load<uint16_t>(r, j * 64 + ol));
        }
        return;
      }
    }
    // FULL: the wave's 64*R rows are all inside the batch -> element j*64+lane, immediate offsets off an SGPR base.
    // Otherwise (nact > 0 rows): range-checked buffer loads, rows past the end read as 0.
    if constexpr (FULL) {
      switch (c.type) {
        case T_I8: { const int8_t* p = (const int8_t*)c.values + w0;
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = (uint32_t)(int32_t)p[j * 64 + lane]; } break;
        case T_U8: { const uint8_t* p = (const uint8_t*)c.values + w0;
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = p[j * 64 + lane]; } break;
        case T_I16: { const int16_t* p = (const int16_t*)c.values + w0;
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = (uint32_t)(int32_t)p[j * 64 + lane]; } break;
        case T_U16: { const uint16_t* p = (const uint16_t*)c.values + w0;
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = p[j * 64 + lane]; } break;

        case T_I32: case T_U32: case T_F32: { const uint32_t* p = (const uint32_t*)c.values + w0;
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = p[j * 64 + lane]; } break;
        case T_I64: case T_U64: case T_F64:
          if constexpr (WIDE) { const uint2* p = (const uint2*)c.values + w0;
#pragma unroll
            for (int j = 0; j < R; ++j) { uint2 x = p[j * 64 + lane]; h[j] = x.y; l[j] = x.x; } }   // (l[] last, like every other case: see below)
          break;
        default: break;
      }
    } else {
      const int ol = opaque_lane(lane);
      switch (c.type) {
        case T_I8: { const auto r = wave_rows_rsrc(c.values, w0, 1, nact);
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = (uint32_t)(int32_t)buf_load<int8_t>(r, j * 64 + ol); } break;
        case T_U8: { const auto r = wave_rows_rsrc(c.values, w0, 1, nact);
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = buf_load<uint8_t>(r, j * 64 + ol); } break;
        case T_I16: { const auto r = wave_rows_rsrc(c.values, w0, 2, nact);
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = (uint32_t)(int32_t)buf_load<int16_t>(r, j * 64 + ol); } break;
        case T_U16: { const auto r = wave_rows_rsrc(c.values, w0, 2, nact);
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = buf_load<uint16_t>(r, j * 64 + ol); } break;

        case T_I32: case T_U32: case T_F32: { const auto r = wave_rows_rsrc(c.values, w0, 4, nact);
#pragma unroll
          for (int j = 0; j < R; ++j) l[j] = buf_load<uint32_t>(r, j * 64 + ol); } break;
        case T_I64: case T_U64: case T_F64:
          if constexpr (WIDE) { const auto r = wave_rows_rsrc(c.values, w0, 8, nact);
#pragma unroll
            for (int j = 0; j < R; ++j) { uint2 x = buf_load<uint2>(r, j * 64 + ol); h[j] = x.y; l[j] = x.x; } }
          break;
        default: break;
      }
    }
  }

  // validity flags of column-ref `ref_idx` for this wave's slots (all active slots when the column has no bitmap)
  __device__ __forceinline__ uint32_t col_valid(const ProgramBlock& pb, int ref_idx) const {
    const void* vb = pb.refs[ref_idx].validity;
    int64_t voff = pb.refs[ref_idx].validity_bit_offset;
    if constexpr (PARTIAL) {
      if (ptr_row && pb.group_bits_at) {   // batch-group launch with bitmaps: this wave's batch has its own
        const u64* ptr_bits = ptr_row + (pb.group_bits_at - 1);
        vb = (const void*)ptr_bits[ref_idx];
        voff = (int64_t)ptr_bits[pb.n_refs + ref_idx];
      }
      if (nact <= 0) return actv;          // (a wave past the end of its batch reads nothing)
    }
    return vb ? (fetch_flags(vb, voff) & actv) : actv;
  }

  __device__ __forceinline__ void fetch_col(const ProgramBlock& pb, const ColRef& c, int ref_idx, uint32_t (&l)[R], uint32_t (&h)[RH], uint32_t& b, uint32_t& v) {
    b = 0;
    if constexpr (!PARTIAL) {
      if (c.type == T_BOOL) b = fetch_flags(c.values, c.bool_bit_offset);
      else fetch_values<true>(c, c.values, l, h);
    } else {
      const void* vals = ptr_row ? (const void*)ptr_row[ref_idx] : c.values;
      // a batch-group launch with bitmaps (ProgramBlock::group_bits_at): this wave's batch has its own validity bitmap and
      // bit position of row 0.  Only the addresses differ -- ONE copy of the loads below serves both kinds of launch (a
      // second inlined copy of the type switch brought spills back into the 1024-thread kernels)
      const void* vb = c.validity;
      int64_t voff = c.validity_bit_offset, boff = c.bool_bit_offset;
      if (ptr_row && pb.group_bits_at) {
        const u64* ptr_bits = ptr_row + (pb.group_bits_at - 1);
        vb = (const void*)ptr_bits[ref_idx];
        voff = boff = (int64_t)ptr_bits[pb.n_refs + ref_idx];
      }
      if (nact == 64 * R && c.type != T_BOOL) {
        // a complete wave inside an incomplete tile (or a batch-group launch): same code as the FULL instantiation
        fetch_values<true>(c, vals, l, h);
      } else {
#pragma unroll
        for (int j = 0; j < R; ++j) l[j] = 0;
#pragma unroll
        for (int j = 0; j < RH; ++j) h[j] = 0;
        if (nact > 0) {
          if (c.type == T_BOOL) b = fetch_flags(vals, boff);   // (a Boolean column's value pointer is its bitmap)
          else fetch_values<false>(c, vals, l, h);
        }
      }
      v = (vb && nact > 0) ? (fetch_flags(vb, voff) & actv) : actv;   // (a wave past the end of its batch reads nothing)
      return;
    }
    v = c.validity ? (fetch_flags(c.validity, c.validity_bit_offset) & actv) : actv;
  }

  // convert values from the class of `from_t` to the class of `to_t` in place: exactly the widenings the
  // coercion table RU/compute_value.rs:350-431 can request, plus numeric -> Boolean (value != 0)
  __device__ __forceinline__ void convert(int from_t, int to_t, uint32_t (&l)[R], uint32_t (&h)[RH], uint32_t& b) {
    const int from = vclass_of<WIDE>(from_t), to = vclass_of<WIDE>(to_t);
    if constexpr (WIDE) {
      if (from_t == T_F16 && (to_t == T_F32 || to_t == T_F64)) {   // half::f16::to_f32 quiets a signalling NaN
#pragma unroll
        for (int j = 0; j < R; ++j) if ((l[j] & 0x7fffffffu) > 0x7f800000u) l[j] |= 0x00400000u;
      }
    }
    if (from == to || from == C_BOOL || from == C_NONE) return;
    // (every wave-uniform choice is made OUTSIDE the slot loops: left inside, the compiler evaluates both sides for
    // every slot and selects with a v_cndmask)
    if (to == C_BOOL) {
      b = 0;
      if (from == C_I32 || from == C_U32) {
#pragma unroll
        for (int j = 0; j < R; ++j) b |= (uint32_t)(l[j] != 0) << j;
      } else if (from == C_F32) {
#pragma unroll
        for (int j = 0; j < R; ++j) b |= (uint32_t)(__uint_as_float(l[j]) != 0.0f) << j;
      } else if constexpr (WIDE) {
        if (from == C_F64) {
#pragma unroll
          for (int j = 0; j < R; ++j) b |= (uint32_t)(__longlong_as_double((long long)PACK64(l[j], h[j])) != 0.0) << j;
        } else {
#pragma unroll
          for (int j = 0; j < R; ++j) b |= (uint32_t)((l[j] | h[j]) != 0) << j;
        }
      }
      return;
    }
    if (to == C_F32) {   // round-to-nearest-even
      if (from == C_I32) {
#pragma unroll
        for (int j = 0; j < R; ++j) l[j] = __float_as_uint((float)(int32_t)l[j]);
      } else {
#pragma unroll
        for (int j = 0; j < R; ++j) l[j] = __float_as_uint((float)l[j]);
      }
      return;
    }
    if (to == C_I32 || to == C_U32) return;   // u8/u16 -> i32: same bits
    if constexpr (WIDE) {
      // (one straight-line loop per source class: with the switch inside the slot loop the compiler kept the loop rolled and
      // indexed l[] / h[] dynamically -- the interpreter's whole register state then lived in scratch memory)
#define CVT_LOOP(EXPR)                                                                                  \
  { _Pragma("unroll") for (int j = 0; j < R; ++j) { const uint64_t x = (uint64_t)(EXPR); l[j] = (uint32_t)x; h[j] = (uint32_t)(x >> 32); } }
      if (to == C_F64) {
        switch (from) {
          case C_I32: CVT_LOOP(__double_as_longlong((double)(int32_t)l[j])) break;
          case C_U32: CVT_LOOP(__double_as_longlong((double)l[j])) break;
          case C_I64: CVT_LOOP(__double_as_longlong((double)(int64_t)PACK64(l[j], h[j]))) break;
          case C_U64: CVT_LOOP(__double_as_longlong((double)PACK64(l[j], h[j]))) break;
          default: CVT_LOOP(__double_as_longlong((double)__uint_as_float(l[j]))) break;
        }
      } else if (from == C_I32) {   // I64 / U64 from a 32-bit integer class
        CVT_LOOP((int64_t)(int32_t)l[j])
      } else {
        CVT_LOOP((uint64_t)l[j])
      }
#undef CVT_LOOP
    }
  }

  // ---- arithmetic: arrow-arith numeric::{add,sub,mul,div,rem}; checked for integers ----------------
  __device__ __forceinline__ void arith(int op, int t, bool rev, bool bconst, const uint32_t (&bl_)[R], const uint32_t (&bh_)[RH], u64* err, uint32_t ref_order) {
#define bl(j) bl_[j]
#define bh(j) bh_[j]
    const int cls = vclass_of<WIDE>(t);
    // x / 2^k and x % 2^k with a literal divisor (`id % 2 = 0` is in the reference's sample queries): shifts and masks
    // instead of the ~40-instruction software division; cannot fail (divisor > 0), truncates toward zero like arrow's
    // div / rem (sign of the dividend)
    if ((op == OP_DIV || op == OP_REM) && bconst && !rev && (cls == C_I32 || cls == C_U32)) {
      const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)bl(0));
      if (d != 0 && d <= 0x40000000u && (d & (d - 1)) == 0) {
        const int k = __builtin_ctz(d);
#pragma unroll
        for (int j = 0; j < R; ++j) {
          if (cls == C_I32) {
            const int32_t x = (int32_t)lo[j];
            const int32_t q = (x + ((x >> 31) & (int32_t)(d - 1))) >> k;
            lo[j] = op == OP_DIV ? (uint32_t)q : (uint32_t)(x - (int32_t)((uint32_t)q << k));
          } else {
            lo[j] = op == OP_DIV ? lo[j] >> k : lo[j] & (d - 1);
          }
        }
        return;
      }
    }
    int errj = -1; uint32_t errc = 0;   // first offending slot of this lane (= its smallest row)
#define BAD(code) do { if (live && errj < 0) { errj = j; errc = (code); } } while (0)
    if (cls == C_I32) {
      int64_t tlo, thi; int_range(t, tlo, thi);
      // computed in 64 bits and range-checked against the declared type; operator and operand order are chosen outside
      // the slot loops (see convert)
#define I32_LOOP(A, B, BODY)                                                                            \
  { _Pragma("unroll") for (int j = 0; j < R; ++j) {                                                       \
      const int32_t x = (int32_t)lo[j], y = (int32_t)bl(j);                                              \
      const int32_t a = (A), b = (B);                                                                    \
      const bool live = (validv >> j) & 1;                                                               \
      int64_t w;                                                                                         \
      BODY                                                                                               \
      if (w < tlo || w > thi) BAD(DE_OVERFLOW);                                                          \
      lo[j] = (uint32_t)w; } }
#define I32_DIR(BODY) { if (rev) I32_LOOP(y, x, BODY) else I32_LOOP(x, y, BODY) }
      switch (op) {
        case OP_ADD: I32_LOOP(x, y, w = (int64_t)a + b;) break;
        case OP_MUL: I32_LOOP(x, y, w = (int64_t)a * b;) break;
        case OP_SUB: I32_DIR(w = (int64_t)a - b;) break;
        // (the software division is ~40 instructions with a dozen temporaries: one slot at a time -- interleaved over all
        // sixteen slots by the scheduler it was the peak of the kernel's register pressure)
        case OP_DIV: I32_DIR(__builtin_amdgcn_sched_barrier(0); if (b == 0) { BAD(DE_DIV_ZERO); w = 0; } else if (a == INT32_MIN && b == -1) w = 2147483648LL; else w = a / b;) break;
        default: I32_DIR(__builtin_amdgcn_sched_barrier(0); if (b == 0) { BAD(DE_DIV_ZERO); w = 0; } else if (b == -1) { if ((int64_t)a == tlo) BAD(DE_OVERFLOW); w = 0; } else w = a % b;) break;
      }
#undef I32_DIR
#undef I32_LOOP
    } else if (cls == C_U32) {
      int64_t tlo, thi; int_range(t, tlo, thi);
#define U32_LOOP(A, B, BODY)                                                                            \
  { _Pragma("unroll") for (int j = 0; j < R; ++j) {                                                       \
      const uint32_t x = lo[j], y = bl(j);                                                               \
      const uint32_t a = (A), b = (B);                                                                   \
      const bool live = (validv >> j) & 1;                                                               \
      int64_t w;                                                                                         \
      BODY                                                                                               \
      if (w < tlo || w > thi) BAD(DE_OVERFLOW);                                                          \
      lo[j] = (uint32_t)w; } }
#define U32_DIR(BODY) { if (rev) U32_LOOP(y, x, BODY) else U32_LOOP(x, y, BODY) }
      switch (op) {
        case OP_ADD: U32_LOOP(x, y, w = (int64_t)a + b;) break;
        case OP_MUL: U32_LOOP(x, y, { uint64_t pr = (uint64_t)a * b; w = pr > 0x7fffffffffffffffULL ? -1 : (int64_t)pr; }) break;
        case OP_SUB: U32_DIR(w = (int64_t)a - b;) break;
        case OP_DIV: U32_DIR(__builtin_amdgcn_sched_barrier(0); if (b == 0) { BAD(DE_DIV_ZERO); w = 0; } else w = a / b;) break;
        default: U32_DIR(__builtin_amdgcn_sched_barrier(0); if (b == 0) { BAD(DE_DIV_ZERO); w = 0; } else w = a % b;) break;
      }
#undef U32_DIR
#undef U32_LOOP
    } else if (cls == C_F32) {
      // One IEEE operation per slot, the operator and operand order decided once outside the slot loop.  Only a NaN
      // result needs more work, and a wave sees one rarely: the fix-ups sit behind a wave-uniform test per slot.
      //   an invalid operation on non-NaN inputs (0/0, inf + -inf, 0 * inf, fmod(x, 0)) yields the DEFAULT NaN, whose sign
      //   is a property of the machine: the reference's hosts (x86-64 SSE) produce 0xFFC00000, gfx950 0x7FC00000.  arrow's
      //   comparisons are totalOrder, so the sign decides whether `nan < x` holds: follow the reference's host.
      //   NaN operands: SSE returns the first NaN operand, quieted (glibc's fmod reaches the same value through
      //   (x * y) / (x * y)); the GPU's fmod and a commuted hardware add / mul may pick the other one or a canonical NaN
#define F32_LOOP(A, B, EXPR)                                                                            \
  { _Pragma("unroll") for (int j = 0; j < R; ++j) {                                                       \
      const float x = __uint_as_float(lo[j]), y = __uint_as_float(bl(j));                                \
      const float a = (A), b = (B);                                                                      \
      float w = (EXPR);                                                                                  \
      if (__ballot(w != w) != 0) {                                                                       \
        const float fa = rev ? y : x, fb = rev ? x : y;   /* first / second operand of the SQL expression */ \
        if (w != w) {                                                                                    \
          if (fa != fa) w = __uint_as_float(__float_as_uint(fa) | 0x00400000u);                          \
          else if (fb != fb) w = __uint_as_float(__float_as_uint(fb) | 0x00400000u);                     \
          else w = __uint_as_float(0xFFC00000u);                                                         \
        }                                                                                                \
      }                                                                                                  \
      lo[j] = __float_as_uint(w); } }
      switch (op) {   // (`rev` is wave-uniform: each direction of a non-commutative operator is its own straight-line loop)
        case OP_ADD: F32_LOOP(x, y, a + b) break;
        case OP_MUL: F32_LOOP(x, y, a * b) break;
        case OP_SUB: if (rev) F32_LOOP(y, x, a - b) else F32_LOOP(x, y, a - b) break;
        case OP_DIV: if (rev) F32_LOOP(y, x, a / b) else F32_LOOP(x, y, a / b) break;
        default: if (rev) F32_LOOP(y, x, fmodf(a, b)) else F32_LOOP(x, y, fmodf(a, b)) break;
      }
#undef F32_LOOP
      if constexpr (WIDE) {
        if (t == T_F16) {
#pragma unroll
          for (int j = 0; j < R; ++j) lo[j] = f16_round(lo[j]);
        }
      }
    } else if constexpr (WIDE) {
      // 64-bit classes: as above, the operator and the operand order are chosen OUTSIDE the slot loops (a switch inside the
      // unrolled loop made the compiler index the register arrays dynamically, i.e. keep them in scratch memory)
#define W_LOOP(TY, A, B, BODY)                                                                          \
  { _Pragma("unroll") for (int j = 0; j < R; ++j) {                                                       \
      const TY x = (TY)PACK64(lo[j], hi[j]), y = (TY)PACK64(bl(j), bh(j));                               \
      const TY a = (A), b = (B);                                                                         \
      const bool live = (validv >> j) & 1;                                                               \
      TY w = 0; bool o = false;                                                                          \
      BODY                                                                                               \
      if (o) BAD(DE_OVERFLOW);                                                                           \
      lo[j] = (uint32_t)(uint64_t)w; hi[j] = (uint32_t)((uint64_t)w >> 32); } }
#define W_DIR(TY, BODY) { if (rev) W_LOOP(TY, y, x, BODY) else W_LOOP(TY, x, y, BODY) }
      if (cls == C_I64) {
        switch (op) {
          case OP_ADD: W_LOOP(long long, x, y, o = __builtin_saddll_overflow(a, b, &w);) break;
          case OP_MUL: W_LOOP(long long, x, y, o = __builtin_smulll_overflow(a, b, &w);) break;
          case OP_SUB: W_DIR(long long, o = __builtin_ssubll_overflow(a, b, &w);) break;
          case OP_DIV: W_DIR(long long, if (b == 0) BAD(DE_DIV_ZERO); else if (a == INT64_MIN && b == -1) o = true; else w = a / b;) break;
          default: W_DIR(long long, if (b == 0) BAD(DE_DIV_ZERO); else if (a == INT64_MIN && b == -1) o = true; else w = a % b;) break;
        }
      } else if (cls == C_U64) {
        switch (op) {
          case OP_ADD: W_LOOP(u64, x, y, o = __builtin_uaddll_overflow(a, b, &w);) break;
          case OP_MUL: W_LOOP(u64, x, y, o = __builtin_umulll_overflow(a, b, &w);) break;
          case OP_SUB: W_DIR(u64, o = __builtin_usubll_overflow(a, b, &w);) break;
          case OP_DIV: W_DIR(u64, if (b == 0) BAD(DE_DIV_ZERO); else w = a / b;) break;
          default: W_DIR(u64, if (b == 0) BAD(DE_DIV_ZERO); else w = a % b;) break;
        }
      } else if (cls == C_F64) {
#define F64_LOOP(A, B, EXPR)                                                                            \
  { _Pragma("unroll") for (int j = 0; j < R; ++j) {                                                       \
      const double x = __longlong_as_double((long long)PACK64(lo[j], hi[j])), y = __longlong_as_double((long long)PACK64(bl(j), bh(j))); \
      const double a = (A), b = (B);                                                                     \
      double w = (EXPR);                                                                                 \
      if (w != w && a == a && b == b) w = __longlong_as_double((long long)0xFFF8000000000000ULL);   /* x86-64 default NaN (see above) */ \
      if (a != a) w = __longlong_as_double(__double_as_longlong(a) | 0x0008000000000000LL);              \
      else if (b != b) w = __longlong_as_double(__double_as_longlong(b) | 0x0008000000000000LL);         \
      const uint64_t u = (uint64_t)__double_as_longlong(w);                                              \
      lo[j] = (uint32_t)u; hi[j] = (uint32_t)(u >> 32); } }
        switch (op) {
          case OP_ADD: if (rev) F64_LOOP(y, x, a + b) else F64_LOOP(x, y, a + b) break;   // (operand order decides which NaN propagates)
          case OP_MUL: if (rev) F64_LOOP(y, x, a * b) else F64_LOOP(x, y, a * b) break;
          case OP_SUB: if (rev) F64_LOOP(y, x, a - b) else F64_LOOP(x, y, a - b) break;
          case OP_DIV: if (rev) F64_LOOP(y, x, a / b) else F64_LOOP(x, y, a / b) break;
          default: if (rev) F64_LOOP(y, x, fmod(a, b)) else F64_LOOP(x, y, fmod(a, b)) break;
        }
#undef F64_LOOP
      }
#undef W_DIR
#undef W_LOOP
    }
#undef BAD
    // arrow's try_binary / try_unary stop at the first offending valid element: report (node, row, kind)
    if (__any(errj >= 0)) { if (errj >= 0) report_error(err, ref_order, w0 + (64 * errj + opaque_lane(lane)), errc); }   // (opaque: the 64-bit `w0 + lane` is not worth a register pair held across the whole program loop)
#undef bl
#undef bh
  }

  // ---- comparisons: arrow-ord cmp::*; floats by IEEE totalOrder ------------------------------------
  __device__ __forceinline__ void compare(int op, int t, bool rev, bool bconst, const uint32_t (&bl)[R], const uint32_t (&bh)[RH], uint32_t bb) {
    const int cls = vclass_of<WIDE>(t);
    if (cls == C_BOOL) {
      const uint32_t a = rev ? bb : bitsv, b = rev ? bitsv : bb;
      const uint32_t ltv = ~a & b, eqv = ~(a ^ b);
      uint32_t r;
      switch (op) {
        case OP_EQ: r = eqv; break; case OP_NE: r = ~eqv; break; case OP_LT: r = ltv; break;
        case OP_LE: r = ltv | eqv; break; case OP_GT: r = ~(ltv | eqv); break; default: r = ~ltv; break;
      }
      bitsv = r & actv;
      return;
    }
    // every operator is one primitive, possibly negated:  EQ: x==y  NE: !(x==y)  LT: a<b  GE: !(a<b)  GT: b<a  LE: !(b<a)
    // with (a, b) = rev ? (y, x) : (x, y); x = accumulator, y = operand
    const bool want_eq = op == OP_EQ || op == OP_NE;
    const bool negate = op == OP_NE || op == OP_GE || op == OP_LE;
    const bool swap = (op == OP_GT || op == OP_LE) != rev;      // compute y < x instead of x < y
    uint32_t r = 0;
#define CMP_LOOPS(TY, XJ, YJ)                                                                              \
  { if (want_eq) { _Pragma("unroll") for (int j = 0; j < R; ++j) { const TY x = (XJ), y = (YJ); r |= (uint32_t)(x == y) << j; } } \
    else if (swap) { _Pragma("unroll") for (int j = 0; j < R; ++j) { const TY x = (XJ), y = (YJ); r |= (uint32_t)(y < x) << j; } }  \
    else { _Pragma("unroll") for (int j = 0; j < R; ++j) { const TY x = (XJ), y = (YJ); r |= (uint32_t)(x < y) << j; } } }
    if (bconst && (cls == C_I32 || cls == C_U32 || cls == C_F32)) {
      // a literal operand lives in one scalar register; against a Float32 literal with a clear sign bit the raw bit
      // patterns already order like the totalOrder keys (see run_cmp_const)
      const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)bl[0]);
      if (cls == C_I32) CMP_LOOPS(int32_t, (int32_t)lo[j], (int32_t)c)
      else if (cls == C_U32) CMP_LOOPS(uint32_t, lo[j], c)
      else if ((int32_t)c >= 0) CMP_LOOPS(int32_t, (int32_t)lo[j], (int32_t)c)
      else { const int32_t kc = f32_key(c); CMP_LOOPS(int32_t, f32_key(lo[j]), kc) }
    }
    else if (cls == C_I32) CMP_LOOPS(int32_t, (int32_t)lo[j], (int32_t)bl[j])
    else if (cls == C_U32) CMP_LOOPS(uint32_t, lo[j], bl[j])
    else if (cls == C_F32) CMP_LOOPS(int32_t, f32_key(lo[j]), f32_key(bl[j]))   // IEEE totalOrder through the integer key
    else if constexpr (WIDE) {
      if (cls == C_I64) CMP_LOOPS(int64_t, (int64_t)PACK64(lo[j], hi[j]), (int64_t)PACK64(bl[j], bh[j]))
      else if (cls == C_U64) CMP_LOOPS(uint64_t, PACK64(lo[j], hi[j]), PACK64(bl[j], bh[j]))
      else CMP_LOOPS(int64_t, f64_key(PACK64(lo[j], hi[j])), f64_key(PACK64(bl[j], bh[j])))
    }
#undef CMP_LOOPS
    bitsv = (negate ? ~r : r) & actv;
  }

  // ---- Utf8 comparison of a column against a scalar string or another column ------------------------
  __device__ __forceinline__ void strcmp_op(int cmp, const ColRef& c, const uint8_t* rhs_bytes, int64_t rhs_len,
                                            const ColRef* rhs_col, bool rev) {
    uint32_t ltv = 0, eqv = 0;
#pragma unroll 1
    for (int j = 0; j < R; ++j) {
      if ((actv >> j) & 1) {
        const int32_t* offs = (const int32_t*)c.values + w0 + 64 * j + lane;
        const int32_t s0 = offs[0], s1 = offs[1];
        const uint8_t* a = (const uint8_t*)c.data + s0; int64_t alen = s1 - s0;
        const uint8_t* b = rhs_bytes; int64_t blen = rhs_len;
        if (rhs_col) {
          const int32_t* ro = (const int32_t*)rhs_col->values + w0 + 64 * j + lane;
          b = (const uint8_t*)rhs_col->data + ro[0]; blen = ro[1] - ro[0];
        }
        if (rev) { const uint8_t* t = a; a = b; b = t; const int64_t tl = alen; alen = blen; blen = tl; }
        const int64_t m = alen < blen ? alen : blen;
        int c3 = 0;
        for (int64_t k = 0; k < m; ++k) { const int d = (int)a[k] - (int)b[k]; if (d) { c3 = d; break; } }
        if (c3 == 0) c3 = (alen > blen) - (alen < blen);
        ltv |= (uint32_t)(c3 < 0) << j; eqv |= (uint32_t)(c3 == 0) << j;
      }
    }
    uint32_t r;
    switch (cmp) {
      case OP_EQ: r = eqv; break; case OP_NE: r = ~eqv; break; case OP_LT: r = ltv; break;
      case OP_LE: r = ltv | eqv; break; case OP_GT: r = ~(ltv | eqv); break; default: r = ~ltv; break;
    }
    bitsv = r & actv;
  }

  // ---- the pre-decoded program (device_program.h: FastOp / FastOperand) -----------------------------------------------
  __device__ __forceinline__ void fetch_raw(const ProgramBlock& pb, int ref_idx, uint32_t (&y)[R]) {
    const void* vals = (PARTIAL && ptr_row) ? (const void*)ptr_row[ref_idx] : pb.refs[ref_idx].values;
    if (!PARTIAL || nact == 64 * R) {
      const uint32_t* src = (const uint32_t*)vals + w0;
#pragma unroll
      for (int j = 0; j < R; ++j) y[j] = src[j * 64 + lane];
    } else {   // incomplete wave: range-checked loads, rows past the end read as 0 (they are inactive: actv masks them)
      const auto rs = wave_rows_rsrc(vals, w0, 4, nact);
      const int ol = opaque_lane(lane);
#pragma unroll
      for (int j = 0; j < R; ++j) y[j] = nact > 0 ? buf_load<uint32_t>(rs, j * 64 + ol) : 0u;
    }
  }

  template <typename StoreFn>
  __device__ __forceinline__ void run_fast(const ProgramBlock& pb, u64* err, StoreFn&& store) {
    bitsv = 0; validv = actv;
    int acc_type = T_BOOL;
    uint32_t y[R];
    uint32_t tb0 = 0, tb1 = 0, tb2 = 0, tb3 = 0;
    for (int pc = 0; pc < pb.n_instr; ++pc) {
      const Instr in = pb.prog[pc];
      const unsigned fop = pb.fast_op[pc] & 0x7fu;
      const bool neg = (pb.fast_op[pc] & FU_NEGATE) != 0;
      const unsigned opd = pb.fast_opd[pc];
      const uint32_t c = (uint32_t)in.imm;
      uint32_t tb = 0;
      if (opd == FO_COL || opd == FO_COL_I2F || opd == FO_COL_U2F) {
        // a column operand brings its validity (only `column <cmp> literal` programs reach this evaluator with a nullable
        // column -- engine.cpp: encode_fast_uops -- i.e. LOAD col, CMP const: boolean temporaries carry no validity here)
        const uint32_t cv = col_valid(pb, (int)in.src_idx);
        validv = (pb.fast_op[pc] & 0x7fu) == FU_LD ? cv : (validv & cv);
      }
      switch (opd) {   // ---- operand ----
        case FO_COL: fetch_raw(pb, in.src_idx, y); stash_put((int)in.src_idx, y); break;
        case FO_COL_I2F:
          fetch_raw(pb, in.src_idx, y); stash_put((int)in.src_idx, y);
#pragma unroll
          for (int j = 0; j < R; ++j) y[j] = __float_as_uint((float)(int32_t)y[j]);
          break;
        case FO_COL_U2F:
          fetch_raw(pb, in.src_idx, y); stash_put((int)in.src_idx, y);
#pragma unroll
          for (int j = 0; j < R; ++j) y[j] = __float_as_uint((float)y[j]);
          break;
        case FO_CONST:
#pragma unroll
          for (int j = 0; j < R; ++j) y[j] = c;
          break;
        case FO_BTEMP:
          switch (in.src_idx) { case 0: tb = tb0; break; case 1: tb = tb1; break; case 2: tb = tb2; break; default: tb = tb3; break; }
          break;
        default: break;
      }
      int errj = -1; uint32_t errc = 0;
      uint32_t r = 0;
#define FAST_F(EXPR, FIRST_IS_ACC)                                                                       \
  { _Pragma("unroll") for (int j = 0; j < R; ++j) {                                                       \
      const float x = __uint_as_float(lo[j]), yy = __uint_as_float(y[j]);                                \
      float w = (EXPR);                                                                                  \
      if (__ballot(w != w) != 0) {   /* rare: the x86 NaN rules, see Interp::arith */                   \
        const float fa = (FIRST_IS_ACC) ? x : yy, fb = (FIRST_IS_ACC) ? yy : x;                          \
        if (w != w) {                                                                                    \
          if (fa != fa) w = __uint_as_float(__float_as_uint(fa) | 0x00400000u);                          \
          else if (fb != fb) w = __uint_as_float(__float_as_uint(fb) | 0x00400000u);                     \
          else w = __uint_as_float(0xFFC00000u);                                                         \
        }                                                                                                \
      }                                                                                                  \
      lo[j] = __float_as_uint(w); }                                                                      \
    acc_type = T_F32; }
#define FAST_I(TY, WEXPR, TLO, THI, T)                                                                   \
  { _Pragma("unroll") for (int j = 0; j < R; ++j) {                                                       \
      const TY x = (TY)lo[j], yy = (TY)y[j];                                                             \
      const int64_t w = (WEXPR);                                                                         \
      if ((w < (TLO) || w > (THI)) && ((actv >> j) & 1) && errj < 0) { errj = j; errc = DE_OVERFLOW; }   \
      lo[j] = (uint32_t)w; }                                                                             \
    acc_type = (T); }
#define FAST_CMP(COND) { _Pragma("unroll") for (int j = 0; j < R; ++j) r |= (uint32_t)(COND) << j; bitsv = (neg ? ~r : r) & actv; acc_type = T_BOOL; }
      switch (fop) {   // ---- operation ----
        case FU_LD:
          if (opd == FO_BTEMP) { bitsv = tb; acc_type = T_BOOL; }
          else {
#pragma unroll
            for (int j = 0; j < R; ++j) lo[j] = y[j];
            acc_type = in.type;
          }
          break;
        case FU_ADD_F: FAST_F(x + yy, !(in.flags & IF_REV)) break;
        case FU_MUL_F: FAST_F(x * yy, !(in.flags & IF_REV)) break;
        case FU_SUB_F: FAST_F(x - yy, true) break;
        case FU_RSUB_F: FAST_F(yy - x, false) break;
        case FU_DIV_F: FAST_F(x / yy, true) break;
        case FU_RDIV_F: FAST_F(yy / x, false) break;
        case FU_ADD_I: FAST_I(int32_t, (int64_t)x + yy, (int64_t)INT32_MIN, (int64_t)INT32_MAX, T_I32) break;
        case FU_SUB_I: FAST_I(int32_t, (int64_t)x - yy, (int64_t)INT32_MIN, (int64_t)INT32_MAX, T_I32) break;
        case FU_RSUB_I: FAST_I(int32_t, (int64_t)yy - x, (int64_t)INT32_MIN, (int64_t)INT32_MAX, T_I32) break;
        case FU_MUL_I: FAST_I(int32_t, (int64_t)x * yy, (int64_t)INT32_MIN, (int64_t)INT32_MAX, T_I32) break;
        case FU_ADD_U: FAST_I(uint32_t, (int64_t)x + yy, 0, 4294967295LL, T_U32) break;
        case FU_SUB_U: FAST_I(uint32_t, (int64_t)x - yy, 0, 4294967295LL, T_U32) break;
        case FU_RSUB_U: FAST_I(uint32_t, (int64_t)yy - x, 0, 4294967295LL, T_U32) break;
        case FU_MUL_U: FAST_I(uint32_t, ((uint64_t)x * yy > 0x7fffffffffffffffULL ? -1 : (int64_t)((uint64_t)x * yy)), 0, 4294967295LL, T_U32) break;
        case FU_DIVP2_I: case FU_REMP2_I: {   // truncating division by 2^k (arrow div / rem: sign of the dividend)
          const int k = __builtin_ctz(c);
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const int32_t x = (int32_t)lo[j];
            const int32_t q = (x + ((x >> 31) & (int32_t)(c - 1))) >> k;
            lo[j] = fop == FU_DIVP2_I ? (uint32_t)q : (uint32_t)(x - (int32_t)((uint32_t)q << k));
          }
          acc_type = T_I32;
        } break;
        case FU_DIVP2_U: {
          const int k = __builtin_ctz(c);
#pragma unroll
          for (int j = 0; j < R; ++j) lo[j] = lo[j] >> k;
          acc_type = T_U32;
        } break;
        case FU_REMP2_U:
#pragma unroll
          for (int j = 0; j < R; ++j) lo[j] = lo[j] & (c - 1);
          acc_type = T_U32;
          break;
        case FU_EQ: FAST_CMP(lo[j] == y[j]) break;
        case FU_LT_I: FAST_CMP((int32_t)lo[j] < (int32_t)y[j]) break;
        case FU_GT_I: FAST_CMP((int32_t)lo[j] > (int32_t)y[j]) break;
        case FU_LT_U: FAST_CMP(lo[j] < y[j]) break;
        case FU_GT_U: FAST_CMP(lo[j] > y[j]) break;
        case FU_LT_F: FAST_CMP(f32_key(lo[j]) < f32_key(y[j])) break;
        case FU_GT_F: FAST_CMP(f32_key(lo[j]) > f32_key(y[j])) break;
        case FU_LT_FKC: { const int32_t kc = f32_key(c); FAST_CMP(f32_key(lo[j]) < kc) } break;
        case FU_GT_FKC: { const int32_t kc = f32_key(c); FAST_CMP(f32_key(lo[j]) > kc) } break;
        case FU_AND: bitsv &= tb; break;
        case FU_OR: bitsv |= tb; break;
        case FU_SPILL:
          switch (in.src_idx) { case 0: tb0 = bitsv; break; case 1: tb1 = bitsv; break; case 2: tb2 = bitsv; break; default: tb3 = bitsv; break; }
          break;
        case FU_CVT_I2F:
#pragma unroll
          for (int j = 0; j < R; ++j) lo[j] = __float_as_uint((float)(int32_t)lo[j]);
          acc_type = T_F32;
          break;
        case FU_CVT_U2F:
#pragma unroll
          for (int j = 0; j < R; ++j) lo[j] = __float_as_uint((float)lo[j]);
          acc_type = T_F32;
          break;
        case FU_STORE: store(in.src_idx, acc_type, *this); break;
        default: break;
      }
#undef FAST_CMP
#undef FAST_I
#undef FAST_F
      // arrow's try_binary stops at the first offending element: report (node, row, kind) -- as Interp::arith does
      if (__any(errj >= 0)) { if (errj >= 0) report_error(err, in.ref_order, w0 + (64 * errj + opaque_lane(lane)), errc); }
    }
  }

  // ---- specialised shape: 32-bit column <cmp> literal (FAST_CMP_CONST), complete waves only ------------------
  __device__ __forceinline__ void run_cmp_const(const ProgramBlock& pb) {
    const Instr cmp = pb.prog[1];
    const uint32_t* src = (const uint32_t*)((PARTIAL && ptr_row) ? (const void*)ptr_row[0] : pb.refs[0].values) + w0;
    uint32_t v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = src[j * 64 + lane];
    stash_put(0, v);
    const int op = cmp.op;
    const bool rev = cmp.flags & IF_REV;
    const bool want_eq = op == OP_EQ || op == OP_NE;
    const bool negate = op == OP_NE || op == OP_GE || op == OP_LE;
    const bool swap = (op == OP_GT || op == OP_LE) != rev;
    const uint32_t c = (uint32_t)cmp.imm;
    uint32_t r = 0;
#define FAST_LOOPS(TY, XJ, Y)                                                                              \
  { const TY y = (Y);                                                                                       \
    if (want_eq) { _Pragma("unroll") for (int j = 0; j < R; ++j) r |= (uint32_t)((XJ) == y) << j; }         \
    else if (swap) { _Pragma("unroll") for (int j = 0; j < R; ++j) r |= (uint32_t)(y < (XJ)) << j; }        \
    else { _Pragma("unroll") for (int j = 0; j < R; ++j) r |= (uint32_t)((XJ) < y) << j; } }
    switch (cmp.type) {
      case T_I32: FAST_LOOPS(int32_t, (int32_t)v[j], (int32_t)c) break;
      case T_U32: FAST_LOOPS(uint32_t, v[j], c) break;
      default:   // T_F32, IEEE totalOrder.  Against a literal with a clear sign bit the raw bit patterns already compare
                 // like the totalOrder keys: the key of a value with the sign bit set stays negative, every other key is
                 // the value's own bit pattern
        if ((int32_t)c >= 0) FAST_LOOPS(int32_t, (int32_t)v[j], (int32_t)c)
        else FAST_LOOPS(int32_t, f32_key(v[j]), f32_key(c))
        break;
    }
#undef FAST_LOOPS
    bitsv = (negate ? ~r : r) & actv;
    validv = col_valid(pb, 0);   // (round 3: `column <cmp> literal` on a column WITH nulls keeps this path)
  }

  // ---- the program loop --------------------------------------------------------------------------
  // numeric temporaries live in LDS: tmp_flags [slot][BLOCK] validity flags, tmp_num [slot][2][R][BLOCK] values
  template <typename StoreFn>
  __device__ __forceinline__ void run(const ProgramBlock& pb, u64* err, uint32_t* tmp_flags, uint32_t* tmp_num, StoreFn&& store) {
    bitsv = 0; validv = actv;
    int acc_type = T_BOOL;
    const int tix = wv * 64 + lane;
    uint32_t bl[R], bh[RH]; uint32_t bb = 0, bv = actv;
    uint32_t tb0 = 0, tb1 = 0, tb2 = 0, tb3 = 0, tv0 = 0, tv1 = 0, tv2 = 0, tv3 = 0;   // boolean temporaries
    static_assert(MAX_BOOL_TEMPS == 4, "boolean temporaries are four named registers");
    for (int pc = 0; pc < pb.n_instr; ++pc) {
      const Instr in = pb.prog[pc];
      const bool rev = in.flags & IF_REV;
      const bool bconst = in.src_kind == SRC_CONST;
      const uint32_t cl = (uint32_t)in.imm, ch = (uint32_t)(in.imm >> 32);
      if (in.op == OP_LOAD && in.src_kind == SRC_COL) {   // straight into the accumulator
        fetch_col(pb, pb.refs[in.src_idx], in.src_idx, lo, hi, bitsv, validv);
        stash_put((int)in.src_idx, lo);
        convert(in.src_type, in.type, lo, hi, bitsv);
        acc_type = in.type;
        continue;
      }
      if (in.src_kind == SRC_COL) {
        fetch_col(pb, pb.refs[in.src_idx], in.src_idx, bl, bh, bb, bv);
        stash_put((int)in.src_idx, bl);
        if (in.op != OP_STRCMP) convert(in.src_type, in.type, bl, bh, bb);
      } else if (bconst) {   // all slots hold the same value: after unrolling the compiler keeps one copy
#pragma unroll
        for (int j = 0; j < R; ++j) bl[j] = cl;
#pragma unroll
        for (int j = 0; j < RH; ++j) bh[j] = ch;
        bb = (in.imm & 1) ? actv : 0; bv = actv;
      } else if (in.src_kind == SRC_TEMP) {
        if (vclass(in.src_type) == C_BOOL) {
          switch (in.src_idx) {
            case 0: bb = tb0; bv = tv0; break; case 1: bb = tb1; bv = tv1; break;
            case 2: bb = tb2; bv = tv2; break; default: bb = tb3; bv = tv3; break;
          }
        } else {
          bb = 0; bv = tmp_flags[(size_t)in.src_idx * BLOCK + tix];
          const uint32_t* tl = tmp_num + (size_t)(in.src_idx * 2) * R * BLOCK;
#pragma unroll
          for (int j = 0; j < R; ++j) bl[j] = tl[j * BLOCK + tix];
          if constexpr (WIDE) {
#pragma unroll
            for (int j = 0; j < R; ++j) bh[j] = tl[(size_t)R * BLOCK + j * BLOCK + tix];
          }
          convert(in.src_type, in.type, bl, bh, bb);
        }
      }
      switch (in.op) {
        case OP_LOAD:   // constant or temporary
#pragma unroll
          for (int j = 0; j < R; ++j) lo[j] = bl[j];
#pragma unroll
          for (int j = 0; j < RH; ++j) hi[j] = bh[j];
          bitsv = bb; validv = bv; acc_type = in.type;
          break;
        case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_REM:
          validv &= bv;
          arith(in.op, in.type, rev, bconst, bl, bh, err, in.ref_order);
          acc_type = in.type;
          break;
        case OP_EQ: case OP_NE: case OP_LT: case OP_LE: case OP_GT: case OP_GE:
          validv &= bv;
          compare(in.op, in.type, rev, bconst, bl, bh, bb);
          acc_type = T_BOOL;
          break;
        case OP_AND: bitsv &= bb; validv &= bv; break;
        case OP_OR: bitsv |= bb; validv &= bv; break;
        case OP_CAST: convert(in.src_type, in.type, lo, hi, bitsv); acc_type = in.type; break;
        case OP_TOBOOL: convert(in.src_type, T_BOOL, lo, hi, bitsv); bitsv &= actv; acc_type = T_BOOL; break;
        case OP_SPILL: {
          if (vclass(in.type) == C_BOOL) {
            switch (in.src_idx) {
              case 0: tb0 = bitsv; tv0 = validv; break; case 1: tb1 = bitsv; tv1 = validv; break;
              case 2: tb2 = bitsv; tv2 = validv; break; default: tb3 = bitsv; tv3 = validv; break;
            }
          } else {
            tmp_flags[(size_t)in.src_idx * BLOCK + tix] = validv;
            uint32_t* tl = tmp_num + (size_t)(in.src_idx * 2) * R * BLOCK;
#pragma unroll
            for (int j = 0; j < R; ++j) tl[j * BLOCK + tix] = lo[j];
            if constexpr (WIDE) {
#pragma unroll
              for (int j = 0; j < R; ++j) tl[(size_t)R * BLOCK + j * BLOCK + tix] = hi[j];
            }
          }
        } break;
        case OP_STRCMP: {
          const int cmp = (int)(in.imm >> 56);
          const bool rhs_is_col = in.flags & IF_STR_RHS_COL;
          const int ridx = (int)(in.imm & 0xffff);
          uint32_t rv = actv;
          if (rhs_is_col && pb.refs[ridx].validity) rv = fetch_flags(pb.refs[ridx].validity, pb.refs[ridx].validity_bit_offset) & actv;
          strcmp_op(cmp, pb.refs[in.src_idx], rhs_is_col ? nullptr : pb.strs[ridx].bytes, rhs_is_col ? 0 : pb.strs[ridx].len,
                    rhs_is_col ? &pb.refs[ridx] : nullptr, rev);
          validv = bv & rv;
          acc_type = T_BOOL;
        } break;
        case OP_STORE: store(in.src_idx, acc_type, *this); break;
        default: break;
      }
    }
  }
};

// LDS scratch for interpreter temporaries (each thread only touches its own slots: no barriers needed)
template <int BLOCK, int R, int NUM_TEMPS>
struct TempLds {
  uint32_t flags[NUM_TEMPS > 0 ? NUM_TEMPS * BLOCK : 1];
  uint32_t num[NUM_TEMPS > 0 ? NUM_TEMPS * 2 * R * BLOCK : 1];
};

// ------------------------------------------------------------------------------------------------
// filter_fused_kernel: predicate + order-preserving compaction of every fixed-width column, one pass.
//   P(i): ticket -> interpret predicate -> per-lane selection flags to LDS, wave counts -> barrier -> tile
//         count -> publish AGGREGATE
//   C(i): look-back -> tile base -> barrier -> per column: load the wave's 64 R rows, store the selected ones
// The loop runs P(i+1) before C(i): an aggregate is published a whole tile ahead of the point where the
// successors' look-backs need it.  Everything a wave derives from the tile index / tile base goes through
// readfirstlane so that global accesses are SGPR-base + 32-bit lane offset.
// ------------------------------------------------------------------------------------------------
// FASTK: the instantiation for pre-decoded programs (FAST_UOPS / FAST_CMP_CONST); it does not contain the generic
// interpreter at all, and the generic instantiation does not contain the fast evaluators -- one evaluator per kernel keeps
// each of them inside the register budget.
// NU > 0: the instantiation that also filters up to NU Utf8 columns (FilterParams::utf8, single-batch launches): P(i)
// adds up the byte lengths of the selected rows, waves 1..NU publish / resolve one chained byte scan each while wave 0
// does the row scan, C(i) writes the new offsets and copies the bytes (short strings one lane per row in 4-byte
// chunks, 64-row groups of long strings through copy_rows_wide).  The separate Utf8 pass it replaces ran its own
// ticket / barrier / look-back chain per 8 192 rows and was latency-bound (2.45 ms for 250 M 8-byte strings).
template <int BLOCK, int R, bool WIDE, int NUM_TEMPS, bool PARTIAL, int NS, bool FASTK, int NU = 0>
__global__ __launch_bounds__(BLOCK) void filter_fused_kernel(const FilterParams p) {
  static_assert(NU + 1 <= BLOCK / 64, "one wave per scan");
  using I = Interp<BLOCK, R, WIDE, PARTIAL, NS>;
  constexpr int NW = BLOCK / 64;
  constexpr int64_t TILE = (int64_t)BLOCK * R;
  __shared__ uint32_t s_sel[2][BLOCK];       // bit j of word t: row (wave, slot j, lane) of thread t is selected
  __shared__ unsigned s_wave_cnt[2][NW];
  __shared__ unsigned s_tot[2];
  __shared__ int64_t s_tile[2];
  __shared__ u64 s_base;
  __shared__ TempLds<BLOCK, R, NUM_TEMPS> s_tmp;
  // Up to NS predicate input columns (<= 4 bytes wide) that are also output columns stay on chip between P(i) and
  // C(i): by the time C(i) runs (a whole tile later) they have left L2, re-reading one costs 4 B/row of HBM traffic
  // (measured: 16.0 GB fetched instead of 12.0 GB for config 2).  Double buffered -- P(i+1) fills one buffer while
  // C(i) drains the other -- so the 160 KiB of a CU take one 4-byte column of a 16 384-row tile (two of a 2 048-row
  // tile).  Measured dead ends for more (round 2): a single-buffered stash whose values wait in
  // registers across C(i) spills, and 8 192-row tiles with four stashed columns lose more to per-tile costs than the
  // re-reads cost.
  __shared__ uint32_t s_stash[2][NS * R * BLOCK];
  constexpr int NUA = NU > 0 ? NU : 1;
  __shared__ uint32_t s_wave_bytes[2][NUA][NW];   // Utf8 columns: bytes of the selected rows per wave / per tile
  __shared__ uint32_t s_utot[2][NUA];
  __shared__ uint32_t s_ubase[NUA];

  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ntiles32 = (int)p.tile_end;                    // this launch: tiles [tile_begin, tile_end)
  const int64_t last_tile = (p.nrows + TILE - 1) / TILE - 1;   // of the whole batch

  // Each phase derives its thread / lane index from its own opaque copy: the addresses built from them (LDS slots of the
  // selection words and the stash, lane byte offsets in every element size) then live inside the phase instead of being
  // computed once in front of the tile loop and held -- or spilled -- across both phases of every tile.
  auto P = [&](int buf) __attribute__((always_inline)) {   // (two call sites: an outlined copy would take the 2.4 KB parameter block through scratch)
    const int lane = fresh_lane(), tid = wv * 64 + lane;
    if (tid == 0) s_tile[buf] = p.tile_begin + (int64_t)atomicAdd(p.ticket, 1u);
    __syncthreads();
    const int64_t tile = uniform64(s_tile[buf]);
    if ((int)tile >= ntiles32) return;   // (tile indices fit 32 bits; a 64-bit signed compare of two scalars runs on the VALU and parks a copy of the bound in VGPRs)
    I it;
    int64_t row0 = tile * TILE, nr = p.nrows;
    if constexpr (PARTIAL) {
      if (p.group_wpb > 0) {   // batch-group launch, wave-granular: this wave's batch and its rows inside it
        const uint32_t wg = (uint32_t)tile * NW + (uint32_t)wv;
        const uint32_t b = wg / (uint32_t)p.group_wpb;
        const uint32_t lw = wg - b * (uint32_t)p.group_wpb;
        if (b < (uint32_t)p.group_nb) {
          const u64* brow = p.group + (int64_t)b * p.group_stride;
          nr = uniform64((int64_t)brow[0]);
          it.ptr_row = brow + 1;
          row0 = ((int64_t)lw - wv) * (64 * R);   // set_rows adds wv * 64 R back
        } else { nr = 0; row0 = 0; it.ptr_row = p.group; }
      } else if (p.group) {   // batch-group launch: the tile's rows and column pointers come from the tile table
        const u64* trow = p.group + tile * p.group_stride;
        row0 = uniform64((int64_t)trow[0]); nr = uniform64((int64_t)trow[1]);
        it.ptr_row = trow + 2;
      }
    }
    it.set_rows(row0, nr, lane, wv);
    it.n_stash = p.n_stash;
#pragma unroll
    for (int k = 0; k < NS; ++k) it.stash_ref[k] = k < p.n_stash ? (int)p.stash_refs[k] : -1;
    it.stash_lds = s_stash[buf];
    if constexpr (FASTK) {
      if (p.pb.fast_kind == FAST_CMP_CONST && it.nact == 64 * R) it.run_cmp_const(p.pb);
      else it.run_fast(p.pb, p.err, [](int, int, I&) {});
    } else it.run(p.pb, p.err, s_tmp.flags, s_tmp.num, [](int, int, I&) {});
    const uint32_t selv = it.bitsv & it.validv;   // null predicate slot = not selected (arrow prep_null_mask_filter)
    s_sel[buf][tid] = selv;
    if (p.sel_mask) {
      // single batch: one word per 64 rows of the batch; wave-packed group: one word per slot of every wave slot
      int64_t gidx = it.w0 >> 6;
      int nst = R;   // slots of this wave that exist: all of them in a complete tile and in a wave-packed group (whose table has a word per slot)
      if constexpr (PARTIAL) {
        if (p.group_wpb > 0) gidx = (int64_t)(((uint32_t)tile * NW + (uint32_t)wv) * R);   // (slot indices fit 32 bits: at most 2^31 / 64 slots)
        else nst = (it.nact + 63) >> 6;
      }
      // lane j keeps slot j's word: one store of R consecutive words per wave (and each ballot's SGPR pair is dead
      // after two selects -- written one by one from lane 0 the R masks were all live at once)
      u64 mine = 0;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const u64 m = __ballot((selv >> j) & 1);
        if (lane == j) mine = m;
      }
      if (lane < nst) p.sel_mask[gidx + lane] = mine;
    }
    unsigned cnt = __popc(selv);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) s_wave_cnt[buf][wv] = cnt;
    if constexpr (NU > 0) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u >= p.n_utf8) break;
        uint32_t bytes = 0;   // of this lane's selected rows; a wave's total stays below 2^31 (int32 offsets)
        // (batch-group launch: the wave's batch has its own offsets, behind the program's and the copied columns' pointers)
        const int32_t* uoffs = p.utf8[u].in_offsets;
        if constexpr (PARTIAL) { if (it.ptr_row) uoffs = (const int32_t*)it.ptr_row[p.pb.n_refs + p.n_out + 2 * u]; }
        if (it.nact == 64 * R) {
          const int32_t* offs = uoffs + it.w0;
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const int32_t o = offs[j * 64 + lane], n = offs[j * 64 + lane + 1];
            bytes += ((selv >> j) & 1) ? (uint32_t)(n - o) : 0u;
          }
        } else if (it.nact > 0) {   // rows [w0, w0 + nact): offsets w0 .. w0 + nact exist, anything past them reads 0
          const auto rs = wave_rows_rsrc(uoffs, it.w0, 4, it.nact + 1);
          const int ol = opaque_lane(lane);
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const uint32_t o = buf_load<uint32_t>(rs, j * 64 + ol), n = buf_load<uint32_t>(rs, j * 64 + ol + 1);
            bytes += ((selv >> j) & 1) ? n - o : 0u;
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bytes += __shfl_xor(bytes, o, 64);
        if (lane == 0) s_wave_bytes[buf][u][wv] = bytes;
      }
    }
    __syncthreads();
    if (wv == 0) {
      unsigned c = (lane < NW) ? s_wave_cnt[buf][lane] : 0u;
      u64 tot = wave_sum((u64)c);
      if (lane == 0) {
        s_tot[buf] = (unsigned)tot;
        st_store(&p.status[tile], (tile == 0 ? ST_INC : ST_AGG) | tot);
      }
    } else if constexpr (NU > 0) {
      if (wv <= p.n_utf8) {   // wave 1 + u owns the byte scan of Utf8 column u
        const int u = wv - 1;
        const uint32_t c = (lane < NW) ? s_wave_bytes[buf][u][lane] : 0u;
        const u64 tot = wave_sum((u64)c);
        if (lane == 0) {
          s_utot[buf][u] = (uint32_t)tot;
          st_store(&p.utf8[u].status[tile], (tile == 0 ? ST_INC : ST_AGG) | tot);
        }
      }
    }
  };

  auto C = [&](int buf) __attribute__((always_inline)) {
    const int lane = fresh_lane(), tid = wv * 64 + lane;
    const int64_t tile = uniform64(s_tile[buf]);
    if (wv == 0) {
      u64 excl = 0;
      if (tile > 0) {
        excl = lookback_exclusive(p.status, tile, 0, lane);
        if (lane == 0) st_store(&p.status[tile], ST_INC | (excl + s_tot[buf]));
      }
      if (lane == 0) { s_base = excl; if (tile == last_tile) *p.total = excl + s_tot[buf]; }
    } else if constexpr (NU > 0) {
      if (wv <= p.n_utf8) {
        const int u = wv - 1;
        u64 excl = 0;
        if (tile > 0) {
          excl = lookback_exclusive(p.utf8[u].status, tile, 0, lane);
          if (lane == 0) st_store(&p.utf8[u].status[tile], ST_INC | (excl + s_utot[buf][u]));
        }
        if (lane == 0) { s_ubase[u] = (uint32_t)excl; if (tile == last_tile) *p.utf8[u].total_bytes = excl + s_utot[buf][u]; }
      }
    }
    __syncthreads();
    u64 off0 = s_base;
    if constexpr (NU > 0) {   // the entry behind the last row: the column's total output bytes
      if (tile == last_tile && tid < p.n_utf8) p.utf8[tid].out_offsets[off0 + s_tot[buf]] = (int32_t)(s_ubase[tid] + s_utot[buf][tid]);
    }
    for (int w = 0; w < wv; ++w) off0 += s_wave_cnt[buf][w];
    off0 = (u64)uniform64((int64_t)off0);
    int64_t row0 = tile * TILE, nr = p.nrows;
    const u64* in_row = nullptr;   // batch-group launch: input pointers of the copied columns
    if constexpr (PARTIAL) {
      if (p.group_wpb > 0) {
        const uint32_t wg = (uint32_t)tile * NW + (uint32_t)wv;
        const uint32_t b = wg / (uint32_t)p.group_wpb;
        const uint32_t lw = wg - b * (uint32_t)p.group_wpb;
        if (b < (uint32_t)p.group_nb) {
          const u64* brow = p.group + (int64_t)b * p.group_stride;
          nr = uniform64((int64_t)brow[0]);
          in_row = brow + 1 + p.pb.n_refs;
          row0 = ((int64_t)lw - wv) * (64 * R);
          // the last wave of a batch knows where the batch's output ends
          if (lw + 1 == (uint32_t)p.group_wpb && lane == 0) p.group_batch_end[b] = off0 + s_wave_cnt[buf][wv];
        } else { nr = 0; row0 = 0; }
      } else if (p.group) {
        const u64* trow = p.group + tile * p.group_stride;
        row0 = uniform64((int64_t)trow[0]); nr = uniform64((int64_t)trow[1]);
        in_row = trow + 2 + p.pb.n_refs;
      }
    }
    auto in_ptr = [&](int k) -> const void* {
      if constexpr (PARTIAL) { if (in_row) return (const void*)in_row[k]; }
      return p.outs[k].in;
    };
    const int64_t w0 = row0 + (int64_t)wv * 64 * R;
    int nact = 64 * R;
    if constexpr (PARTIAL) {
      const int64_t rem = nr - w0;
      nact = rem <= 0 ? 0 : (rem >= 64 * R ? 64 * R : (int)rem);
      if (nact <= 0) return;
    }
    const uint32_t selv = s_sel[buf][tid];
    if (p.grp_base) {
      u64 run = off0;
      int64_t gidx = w0 >> 6;
      int nst = R;
      if constexpr (PARTIAL) {
        if (p.group_wpb > 0) gidx = (int64_t)(((uint32_t)tile * NW + (uint32_t)wv) * R);
        else nst = (nact + 63) >> 6;
      }
      u64 mine = 0;   // lane j keeps slot j's base: one store of R consecutive words per wave
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const u64 m = __ballot((selv >> j) & 1);
        if (lane == j) mine = run;
        run += __popcll(m);
      }
      if (lane < nst) p.grp_base[gidx + lane] = mine;
    }
    // Utf8 columns: the offsets of the first step are requested before the fixed-width copies, whose loads they join
    constexpr int CH = 4;                                   // 64-row groups per step (8 spill at the 128-VGPR budget of the 1024-thread tile: 3.0 ms vs 2.2 ms)
    uint32_t on[CH], nn[CH];
    auto utf8_ptr = [&](int u, int which) -> const void* {   // which: 0 offsets, 1 data; per batch in a group launch
      if constexpr (PARTIAL) { if (in_row) return (const void*)in_row[p.n_out + 2 * u + which]; }
      return which ? (const void*)p.utf8[u].in_data : (const void*)p.utf8[u].in_offsets;
    };
    auto load_offsets = [&](int u, int j0) __attribute__((always_inline)) {
      const int32_t* base = (const int32_t*)utf8_ptr(u, 0);
      const int32_t* offs = base + w0;
      const auto rs = wave_rows_rsrc(base, w0, 4, nact + 1);
      const int ol = opaque_lane(lane);
#pragma unroll
      for (int jj = 0; jj < CH; ++jj) {
        const uint32_t e = (uint32_t)((j0 + jj) * 64 + lane);
        if (nact == 64 * R) { on[jj] = (uint32_t)offs[e]; nn[jj] = (uint32_t)offs[e + 1]; }
        else { on[jj] = buf_load<uint32_t>(rs, (j0 + jj) * 64 + ol); nn[jj] = buf_load<uint32_t>(rs, (j0 + jj) * 64 + ol + 1); }
      }
    };
    if constexpr (NU > 0) { if (p.n_utf8 > 0) load_offsets(0, 0); }
    const int ncopy = p.n_out - p.n_stash;   // the stashed columns are outs[ncopy .. n_out)
    auto copy_columns = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      for (int c = 0; c < ncopy; ++c) {
        const OutCol oc = p.outs[c];
        const int ol = FULL ? lane : opaque_lane(lane);
        // CH values per lane are loaded (one contiguous 64-element run per instruction), then the selected ones are
        // stored at off0 + (selected rows in earlier slots) + rank of the lane among the selected lanes of its slot
        // complete wave: element j*64+lane off an SGPR base; incomplete wave: range-checked buffer load (0 past the end)
#define LOADV(TY, src, rs, j) (FULL ? (src)[(j) * 64 + lane] : buf_load<TY>(rs, (j) * 64 + ol))
#define COPY_COL(TY, CH)                                                                              \
  { const TY* src = (const TY*)in_ptr(c) + w0; TY* dst = (TY*)oc.out + off0; unsigned run = 0;        \
    const auto rs = wave_rows_rsrc(in_ptr(c), w0, sizeof(TY), nact);                                    \
    _Pragma("unroll") for (int j0 = 0; j0 < R; j0 += CH) {                                             \
      TY v[CH];                                                                                        \
      _Pragma("unroll") for (int jj = 0; jj < CH; ++jj) v[jj] = LOADV(TY, src, rs, j0 + jj);            \
      _Pragma("unroll") for (int jj = 0; jj < CH; ++jj) {                                              \
        const bool sel = (selv >> (j0 + jj)) & 1;                                                      \
        const u64 m = __ballot(sel);                                                                   \
        if (sel) dst[run + lane_rank(m)] = v[jj];                                                      \
        run += __popcll(m); } } }
        switch (oc.width) {
          case 1: COPY_COL(uint8_t, R) break;
          case 2: COPY_COL(uint16_t, R) break;
          case 4: {
            // runs of 4-byte columns are software-pipelined: the loads of the next column are issued before the
            // stores of the current one, so the CU always has loads in flight
            int cl = c;
            while (cl + 1 < ncopy && p.outs[cl + 1].width == 4) ++cl;
            uint32_t va[R], vb[R];
            auto ld = [&](int k, uint32_t (&v)[R]) {
              const uint32_t* src = (const uint32_t*)in_ptr(k) + w0;
              const auto rs = wave_rows_rsrc(in_ptr(k), w0, 4, nact);
#pragma unroll
              for (int j = 0; j < R; ++j) v[j] = LOADV(uint32_t, src, rs, j);
            };
            auto st = [&](int k, const uint32_t (&v)[R]) {
              uint32_t* dst = (uint32_t*)p.outs[k].out + off0;
              unsigned run = 0;
#pragma unroll
              for (int j = 0; j < R; ++j) {
                const bool sel = (selv >> j) & 1;
                const u64 m = __ballot(sel);
                if (sel) dst[run + lane_rank(m)] = v[j];
                run += __popcll(m);
              }
            };
            ld(c, va);
            for (int k = c; k <= cl; k += 2) {
              if (k + 1 <= cl) ld(k + 1, vb);
              st(k, va);
              if (k + 1 <= cl) { if (k + 2 <= cl) ld(k + 2, va); st(k + 1, vb); }
            }
            c = cl;
          } break;
          case 8: COPY_COL(uint2, (R >= 8 ? 8 : R)) break;
          default: {   // 16-byte values (decimal128 ...): named registers, a private array would be promoted to LDS
            const uint4* src = (const uint4*)in_ptr(c) + w0; uint4* dst = (uint4*)oc.out + off0; unsigned run = 0;
            const auto rs = wave_rows_rsrc(in_ptr(c), w0, 16, nact);
#pragma unroll
            for (int j0 = 0; j0 < R; j0 += 2) {
              const uint4 va = LOADV(uint4, src, rs, j0), vb = LOADV(uint4, src, rs, j0 + 1);
              const bool sa = (selv >> j0) & 1, sb = (selv >> (j0 + 1)) & 1;
              const u64 ma = __ballot(sa);
              if (sa) dst[run + lane_rank(ma)] = va;
              run += __popcll(ma);
              const u64 mb = __ballot(sb);
              if (sb) dst[run + lane_rank(mb)] = vb;
              run += __popcll(mb);
            }
          } break;
        }
#undef COPY_COL
#undef LOADV
      }
    };
    if constexpr (PARTIAL) {
      if (nact == 64 * R) copy_columns(std::true_type{});   // complete wave: unclamped, immediate-offset loads
      else copy_columns(std::false_type{});
    } else copy_columns(std::true_type{});
#pragma unroll
    for (int k = 0; k < NS; ++k) {   // the stashed predicate columns: values are still in LDS
      if (k >= p.n_stash) break;
      const OutCol oc = p.outs[ncopy + k];
      const uint32_t* sv = s_stash[buf] + k * R * BLOCK + tid;
      unsigned run = 0;
#define COPY_STASH(TY)                                                                                \
  { TY* dst = (TY*)oc.out + off0;                                                                     \
    _Pragma("unroll") for (int j = 0; j < R; ++j) {                                                    \
      const bool sel = (selv >> j) & 1;                                                                \
      const u64 m = __ballot(sel);                                                                     \
      if (sel) dst[run + lane_rank(m)] = (TY)sv[j * BLOCK];                                            \
      run += __popcll(m); } }
      if (oc.width == 4) COPY_STASH(uint32_t) else if (oc.width == 2) COPY_STASH(uint16_t) else COPY_STASH(uint8_t)
#undef COPY_STASH
    }
    if constexpr (NU > 0) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u >= p.n_utf8) break;
        Utf8Fold uf = p.utf8[u];
        uf.in_data = (const uint8_t*)utf8_ptr(u, 1);
        uint32_t boff = s_ubase[u];                         // first output byte of this wave's rows
        for (int w = 0; w < wv; ++w) boff += s_wave_bytes[buf][u][w];
        boff = __builtin_amdgcn_readfirstlane(boff);
        int32_t* oo = uf.out_offsets + off0;
        const bool copy = uf.out_data != nullptr;           // null: new offsets only, utf8_copy_kernel moves the (long) strings
        unsigned run = 0;
        // CH groups of 64 rows per step.  The offsets of the next step are requested right behind the data loads of the
        // current one, so one memory latency per step is exposed (each step's data) instead of two; the stores of the
        // new offsets are issued after those loads, so waiting for the data does not wait for them.  Everything
        // wave-uniform (byte bases, group counts, ballots) lives in SGPRs.
        if (u > 0) load_offsets(u, 0);
#pragma unroll 1
        for (int j0 = 0; j0 < R; j0 += CH) {
          uint32_t src[CH], len[CH], dpos[CH], gbase[CH], gcnt[CH], grow[CH];   // (32-bit lane offsets off uniform bases)
          uint32_t longm = 0;                               // bit jj: the group's rows average > 24 bytes -> copy_rows_wide
          uint32_t lor = 0;                                 // OR of the lengths this lane copies itself (>= their maximum)
          uint32_t incl[CH];
#pragma unroll
          for (int jj = 0; jj < CH; ++jj) {
            src[jj] = on[jj];
            len[jj] = ((selv >> (j0 + jj)) & 1) ? nn[jj] - on[jj] : 0u;
            incl[jj] = len[jj];
          }
#pragma unroll
          for (int jj = 0; jj < CH; jj += 2) wave_incl_scan2(incl[jj], incl[jj + 1]);
#pragma unroll
          for (int jj = 0; jj < CH; ++jj) {
            const bool sel = (selv >> (j0 + jj)) & 1;
            const uint32_t inc = incl[jj];
            const uint32_t gb = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            const uint32_t c = (uint32_t)__popcll(__ballot(sel));
            dpos[jj] = boff + inc - len[jj];
            if (gb > c * 24u) longm |= 1u << jj; else lor |= len[jj];
            gbase[jj] = boff; gcnt[jj] = c; grow[jj] = run;
            boff += gb; run += c;
          }
          const bool more = copy && __ballot(lor > 8u) != 0;                         // some row has bytes 8..15 to copy
          const bool tails = copy && __ballot((lor & 3u) != 0 || lor > 16u) != 0;    // some row is longer than 16 bytes or has 1-3 odd bytes
          if (!copy) longm = 0;
          // bytes 0..7 of every row this lane copies: one 8-byte access where the row has them (the address unit's cost
          // is per lane and instruction, not per byte), else one 4-byte access
          uint32_t w4[CH][2];
#pragma unroll
          for (int jj = 0; jj < CH; ++jj) {
            const uint8_t* sp = uf.in_data + src[jj];
            w4[jj][0] = 0; w4[jj][1] = 0;
            if (copy && !((longm >> jj) & 1)) {
              if (len[jj] >= 8u) __builtin_memcpy(&w4[jj][0], sp, 8);
              else if (len[jj] >= 4u) __builtin_memcpy(&w4[jj][0], sp, 4);
            }
          }
          if (j0 + CH < R) load_offsets(u, j0 + CH);
#pragma unroll
          for (int jj = 0; jj < CH; ++jj) {
            const bool sel = (selv >> (j0 + jj)) & 1;
            const u64 m = __ballot(sel);
            if (sel) oo[grow[jj] + lane_rank(m)] = (int32_t)dpos[jj];
          }
#pragma unroll
          for (int jj = 0; jj < CH; ++jj) {
            if (!copy || ((longm >> jj) & 1)) continue;
            uint8_t* dp = uf.out_data + dpos[jj];
            if (len[jj] >= 8u) __builtin_memcpy(dp, &w4[jj][0], 8);
            else if (len[jj] >= 4u) __builtin_memcpy(dp, &w4[jj][0], 4);
          }
          if (more) {   // bytes 8..15, a second round through the same registers
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
              const uint8_t* sp = uf.in_data + src[jj];
              w4[jj][0] = 0; w4[jj][1] = 0;
              if (!((longm >> jj) & 1)) {
                if (len[jj] >= 16u) __builtin_memcpy(&w4[jj][0], sp + 8, 8);
                else if (len[jj] >= 12u) __builtin_memcpy(&w4[jj][0], sp + 8, 4);
              }
            }
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
              if ((longm >> jj) & 1) continue;
              uint8_t* dp = uf.out_data + dpos[jj];
              if (len[jj] >= 16u) __builtin_memcpy(dp + 8, &w4[jj][0], 8);
              else if (len[jj] >= 12u) __builtin_memcpy(dp + 8, &w4[jj][0], 4);
            }
          }
          if (tails) {
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
              if ((longm >> jj) & 1) continue;
              const uint8_t* sp = uf.in_data + src[jj];
              uint8_t* dp = uf.out_data + dpos[jj];
              const int l = (int)len[jj];
              int b = l < 16 ? (l & ~3) : 16;
              for (; b + 4 <= l; b += 4) { uint32_t w; __builtin_memcpy(&w, sp + b, 4); __builtin_memcpy(dp + b, &w, 4); }
              for (; b < l; ++b) dp[b] = sp[b];
            }
          }
          if (longm) {
#pragma unroll 1
            for (int jj = 0; jj < CH; ++jj) {
              if (!((longm >> jj) & 1)) continue;
              const bool sel = (selv >> (j0 + jj)) & 1;
              const u64 m = __ballot(sel);
              // (src, len, dst) of the selected rows in rank order: lane k serves the row of rank k
              const unsigned dl = sel ? lane_rank(m) : 63u - lane_rank(~m);
              uint32_t s0 = src[0], l0 = len[0], d0 = dpos[0], g0 = gbase[0], c0 = gcnt[0];
#pragma unroll
              for (int k = 1; k < CH; ++k) if (jj == k) { s0 = src[k]; l0 = len[k]; d0 = dpos[k]; g0 = gbase[k]; c0 = gcnt[k]; }
              const int prs = __builtin_amdgcn_ds_permute((int)(dl << 2), (int)s0);
              const int prl = __builtin_amdgcn_ds_permute((int)(dl << 2), (int)l0);
              const int prd = __builtin_amdgcn_ds_permute((int)(dl << 2), (int)(d0 - g0));
              copy_rows_wide(uf.in_data, uf.out_data + g0, prs, prl, prd, (int)c0, lane);
            }
          }
        }
      }
    }
  };

  P(0);
  int itn = 0;
  while (true) {
    if (__builtin_amdgcn_readfirstlane((int)s_tile[itn & 1]) >= ntiles32) break;
    P((itn + 1) & 1);
    C(itn & 1);
    ++itn;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// filter_project_kernel: filter_fused_kernel's scan, but C(i) evaluates the select items on the tile's surviving rows
// and stores the results compacted -- the filtered batch itself never exists in HBM.
//   P(i): as in filter_fused_kernel.
//   C(i): look-back -> tile base -> the projection program runs with the active-row mask narrowed to the selected
//         rows (so validity, and with it checked-arithmetic error reporting, only sees rows the reference's
//         project_record would see); every STORE compacts with the same ballots / ranks as a column copy.
// Any error word set by either phase makes the host discard the result and take the two-call path, which reports
// the reference's error (predicate errors first, then the select items in order).
// ------------------------------------------------------------------------------------------------
template <int BLOCK, int R, bool WIDE, int NUM_TEMPS, bool FASTK>
__global__ __launch_bounds__(BLOCK) void filter_project_kernel(const FusedParams p) {
  constexpr int NW = BLOCK / 64;
  constexpr int64_t TILE = (int64_t)BLOCK * R;
  using I = Interp<BLOCK, R, WIDE, true>;
  __shared__ uint32_t s_sel[2][BLOCK];
  __shared__ unsigned s_wave_cnt[2][NW];
  __shared__ unsigned s_tot[2];
  __shared__ int64_t s_tile[2];
  __shared__ u64 s_base;
  __shared__ TempLds<BLOCK, R, NUM_TEMPS> s_tmp;

  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ntiles32 = (int)p.tile_end;   // (32-bit tile compares and per-phase lane indices: see filter_fused_kernel)
  const int64_t last_tile = (p.nrows + TILE - 1) / TILE - 1;

  auto P = [&](int buf) __attribute__((always_inline)) {
    const int lane = fresh_lane(), tid = wv * 64 + lane;
    if (tid == 0) s_tile[buf] = p.tile_begin + (int64_t)atomicAdd(p.ticket, 1u);
    __syncthreads();
    const int64_t tile = uniform64(s_tile[buf]);
    if ((int)tile >= ntiles32) return;
    I it;
    it.set_rows(tile * TILE, p.nrows, lane, wv);
    if constexpr (FASTK) {
      if (p.pred.fast_kind == FAST_CMP_CONST && it.nact == 64 * R) it.run_cmp_const(p.pred);
      else it.run_fast(p.pred, p.err, [](int, int, I&) {});
    } else it.run(p.pred, p.err, s_tmp.flags, s_tmp.num, [](int, int, I&) {});
    const uint32_t selv = it.bitsv & it.validv;
    s_sel[buf][tid] = selv;
    unsigned cnt = __popc(selv);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) s_wave_cnt[buf][wv] = cnt;
    __syncthreads();
    if (wv == 0) {
      unsigned c = (lane < NW) ? s_wave_cnt[buf][lane] : 0u;
      u64 tot = wave_sum((u64)c);
      if (lane == 0) {
        s_tot[buf] = (unsigned)tot;
        st_store(&p.status[tile], (tile == 0 ? ST_INC : ST_AGG) | tot);
      }
    }
  };

  auto C = [&](int buf) __attribute__((always_inline)) {
    const int lane = fresh_lane(), tid = wv * 64 + lane;
    const int64_t tile = uniform64(s_tile[buf]);
    if (wv == 0) {
      u64 excl = 0;
      if (tile > 0) {
        excl = lookback_exclusive(p.status, tile, 0, lane);
        if (lane == 0) st_store(&p.status[tile], ST_INC | (excl + s_tot[buf]));
      }
      if (lane == 0) { s_base = excl; if (tile == last_tile) *p.total = excl + s_tot[buf]; }
    }
    __syncthreads();
    u64 off0 = s_base;
    for (int w = 0; w < wv; ++w) off0 += s_wave_cnt[buf][w];
    off0 = (u64)uniform64((int64_t)off0);
    I it;
    it.set_rows(tile * TILE, p.nrows, lane, wv);
    if (it.nact <= 0) return;
    const uint32_t selv = s_sel[buf][tid];
    const int64_t w0 = it.w0;
    const int nact = it.nact;
    // ---- select items that are plain columns: compacting copies --------------------------------------------
    for (int c = 0; c < p.n_copy; ++c) {
      const OutCol oc = p.copies[c];
      const int ol = opaque_lane(lane);
#define FCOPY(TY)                                                                                     \
  { const TY* src = (const TY*)oc.in + w0; TY* dst = (TY*)oc.out + off0; unsigned run = 0;            \
    const auto rs = wave_rows_rsrc(oc.in, w0, sizeof(TY), nact);                                       \
    _Pragma("unroll") for (int j0 = 0; j0 < R; j0 += 4) {                                              \
      TY v[4];                                                                                         \
      _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                 \
        v[jj] = nact == 64 * R ? src[(j0 + jj) * 64 + lane] : buf_load<TY>(rs, (j0 + jj) * 64 + ol); \
      _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                               \
        const bool sel = (selv >> (j0 + jj)) & 1;                                                      \
        const u64 m = __ballot(sel);                                                                   \
        if (sel) dst[run + lane_rank(m)] = v[jj];                                                      \
        run += __popcll(m); } } }
      static_assert(R % 4 == 0, "copy loop handles four slots per step");
      switch (oc.width) {
        case 1: FCOPY(uint8_t) break;
        case 2: FCOPY(uint16_t) break;
        case 4: FCOPY(uint32_t) break;
        case 8: FCOPY(uint2) break;
        default: FCOPY(uint4) break;
      }
#undef FCOPY
    }
    // ---- computed select items ----------------------------------------------------------------------------
    if (p.n_proj > 0) {
      it.actv &= selv;
      auto compact_store = [&](int out_idx, int acc_type, I& s) {
        const ProjOut po = p.outs[out_idx];
        unsigned run = 0;
#define CSTORE(TY, EXPR)                                                                              \
  { TY* dst = (TY*)po.values + off0;                                                                  \
    _Pragma("unroll") for (int j = 0; j < R; ++j) {                                                    \
      const bool sel = (selv >> j) & 1;                                                                \
      const u64 m = __ballot(sel);                                                                     \
      if (sel) dst[run + lane_rank(m)] = (EXPR);                                                       \
      run += __popcll(m); } }
        if constexpr (WIDE) {
          if (acc_type == T_F16) { CSTORE(uint16_t, f16_bits(s.lo[j])) return; }
        }
        switch (acc_type) {
          case T_I8: case T_U8: CSTORE(uint8_t, (uint8_t)s.lo[j]) break;
          case T_I16: case T_U16: CSTORE(uint16_t, (uint16_t)s.lo[j]) break;
          case T_I32: case T_U32: case T_F32: CSTORE(uint32_t, s.lo[j]) break;
          default: if constexpr (WIDE) CSTORE(uint2, make_uint2(s.lo[j], s.hi[j])) break;
        }
#undef CSTORE
      };
      if constexpr (FASTK) it.run_fast(p.proj, p.err, compact_store);
      else it.run(p.proj, p.err, s_tmp.flags, s_tmp.num, compact_store);
    }
  };

  P(0);
  int itn = 0;
  while (true) {
    if (__builtin_amdgcn_readfirstlane((int)s_tile[itn & 1]) >= ntiles32) break;
    P((itn + 1) & 1);
    C(itn & 1);
    ++itn;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// project_kernel: evaluate every SelectItem expression densely (no compaction, no inter-tile state)
// ------------------------------------------------------------------------------------------------
template <int BLOCK, int R, bool WIDE, int NUM_TEMPS, bool PARTIAL, bool FASTK>
__global__ __launch_bounds__(BLOCK) void project_kernel(const ProjectParams p) {
  constexpr int64_t TILE = (int64_t)BLOCK * R;
  __shared__ TempLds<BLOCK, R, NUM_TEMPS> s_tmp;
  // null counts are gathered per workgroup in LDS and added to the outputs' counters once, at the end: one global atomic
  // per wave and tile put ~10^6 atomics on one address for a 10^9-row projection of a nullable column (9.6 ms for 400 M
  // rows against 1.5 ms for the same projection of non-null columns)
  __shared__ unsigned s_nulls[MAX_PROJ];   // (a workgroup sees fewer than 2^31 rows: batches are below that)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < MAX_PROJ) s_nulls[tid] = 0;
  __syncthreads();
  for (int64_t tile = p.tile_begin + blockIdx.x; tile < p.tile_end; tile += gridDim.x) {
    Interp<BLOCK, R, WIDE, PARTIAL> it;
    it.set_rows(tile * TILE, p.nrows, lane, wv);
    auto dense_store = [&](int out_idx, int acc_type, Interp<BLOCK, R, WIDE, PARTIAL>& s) {
      const ProjOut po = p.outs[out_idx];
      if (s.nact <= 0) return;
      const int cls = vclass(acc_type);
      if (cls == C_BOOL) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const u64 m = __ballot((s.bitsv >> j) & 1);
          if (lane == 0 && s.w0 + 64 * j < p.nrows) ((u64*)po.values)[(s.w0 >> 6) + j] = m;
        }
      } else {
        bool stored = false;
#define STORE_COL(TY, EXPR)                                                              \
  { TY* dst = (TY*)po.values + s.w0;                                                     \
    _Pragma("unroll") for (int j = 0; j < R; ++j) if ((s.actv >> j) & 1) dst[j * 64 + lane] = (EXPR); }
        if constexpr (WIDE) {
          if (acc_type == T_F16) { STORE_COL(uint16_t, f16_bits(s.lo[j])) stored = true; }
        }
        if (!stored) switch (acc_type) {
          case T_I8: case T_U8: STORE_COL(uint8_t, (uint8_t)s.lo[j]) break;
          case T_I16: case T_U16: STORE_COL(uint16_t, (uint16_t)s.lo[j]) break;
          case T_I32: case T_U32: case T_F32: STORE_COL(uint32_t, s.lo[j]) break;
          default: if constexpr (WIDE) STORE_COL(uint2, make_uint2(s.lo[j], s.hi[j])) break;
#undef STORE_COL
        }
      }
      if (po.validity) {
        unsigned nulls = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const u64 v = __ballot((s.validv >> j) & 1);
          const u64 a = s.group_act(j);
          if (a) { if (lane == 0) po.validity[(s.w0 >> 6) + j] = v; nulls += __popcll(a & ~v); }
        }
        if (nulls && lane == 0) atomicAdd(&s_nulls[out_idx], nulls);
      }
    };
    if constexpr (FASTK) it.run_fast(p.pb, p.err, dense_store);
    else it.run(p.pb, p.err, s_tmp.flags, s_tmp.num, dense_store);
    __syncthreads();   // temporaries in LDS are reused by the next tile
  }
  if (tid == 0) {
#pragma unroll 1
    for (int k = 0; k < p.n_proj; ++k) if (s_nulls[k]) atomicAdd(p.outs[k].null_count, (u64)s_nulls[k]);   // (uniform index: scalar loads of the argument block)
  }
}

// ------------------------------------------------------------------------------------------------
// bit_compact_kernel: out bit k = in bit of the k-th selected row (Boolean values, validity bitmaps), into a
// zero-initialised output.
// ------------------------------------------------------------------------------------------------
// Round 3: one LANE per 64-row group (the first version walked the groups of a chunk one after the other with a ds_permute +
// ballot each: ~100 wave-instructions per group, 0.85 ms per bitmap of a 400 M-row batch -- as long as the main kernel).
// Every lane compresses its group's 64 input bits under the group's selection word in registers (parallel-suffix compress,
// Hacker's Delight 7-4), the wave ORs the 64 variable-length pieces into a 4 Kibit LDS window at their output bit positions
// (known per group from the main kernel's scan), and the window goes out as whole words: plain coalesced stores inside,
// atomicOr for the two words it may share with the neighbouring waves.
__device__ __forceinline__ u64 pext64(u64 x, u64 m) {
  x &= m;
  u64 mk = ~m << 1;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    u64 mp = mk ^ (mk << 1);
    mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16; mp ^= mp << 32;
    const u64 mv = mp & m;
    m = (m ^ mv) | (mv >> (1 << i));
    const u64 t = x & mv;
    x = (x ^ t) | (t >> (1 << i));
    mk &= ~mp;
  }
  return x;
}

// lanes hold (comp, cnt, pos) of 64 consecutive groups (cnt = 0: nothing to write); s_win: this wave's 132 words of LDS
__device__ __forceinline__ void emit_compacted_bits(uint32_t* s_win, u64 comp, int cnt, u64 pos, uint32_t* out_bits, int lane) {
  const u64 live = __ballot(cnt > 0);
  if (!live) return;
  const int first = __builtin_ctzll(live), last = 63 - __builtin_clzll(live);
  const u64 pos_first = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pos, first)) | ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pos >> 32), first) << 32);
  const u64 pos_last = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pos, last)) | ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pos >> 32), last) << 32);
  const int cnt_last = __builtin_amdgcn_readlane(cnt, last);
  const u64 wbase = pos_first >> 5;
  const int nwords = (int)(((pos_last + (u64)cnt_last) - (wbase << 5) + 31) >> 5);   // <= 4096 / 32 + 1
  for (int i = lane; i < 132; i += 64) s_win[i] = 0;
  __builtin_amdgcn_wave_barrier();
  if (cnt > 0) {
    const uint32_t rel = (uint32_t)(pos - (wbase << 5));
    const int sh = (int)(rel & 31u), w = (int)(rel >> 5);
    const u64 lo = comp << sh;
    const uint32_t hi = sh ? (uint32_t)(comp >> (64 - sh)) : 0u;
    if ((uint32_t)lo) atomicOr(&s_win[w], (uint32_t)lo);
    if ((uint32_t)(lo >> 32)) atomicOr(&s_win[w + 1], (uint32_t)(lo >> 32));
    if (hi) atomicOr(&s_win[w + 2], hi);
  }
  __builtin_amdgcn_wave_barrier();   // (one wave: its LDS operations execute in order)
  for (int i = lane; i < nwords; i += 64) {
    const uint32_t v = s_win[i];
    if (!v) continue;                                         // (the output is zero-initialised)
    if (i == 0 || i == nwords - 1) atomicOr(&out_bits[wbase + i], v); else out_bits[wbase + i] = v;
  }
  __builtin_amdgcn_wave_barrier();
}

template <int BLOCK, int R>
__global__ __launch_bounds__(BLOCK) void bit_compact_kernel(const BitCompactParams p) {
  constexpr int NW = BLOCK / 64;
  __shared__ uint32_t s_win[NW][132];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t ngroups = (p.nrows + 63) >> 6;
  const int64_t nchunks = (ngroups + 63) >> 6;               // 64 groups = 4096 rows per wave and step
  u64 zeros_all = 0;   // ONE atomic per wave (one per chunk put 10^5 atomics on the same address)
  for (int64_t chunk = (int64_t)blockIdx.x * NW + wv; chunk < nchunks; chunk += (int64_t)gridDim.x * NW) {
    const int64_t g = chunk * 64 + lane;
    u64 comp = 0, pos = 0; int cnt = 0;
    if (g < ngroups) {
      const u64 act = active_mask(g << 6, p.nrows);
      const u64 m = p.sel_mask[g] & act;
      cnt = __popcll(m);
      if (cnt) {
        comp = pext64(load_bits64(p.in_bits, p.in_bit_offset + (g << 6), act), m);
        pos = p.grp_base[g];
      }
    }
    unsigned zeros = (unsigned)(cnt - __popcll(comp));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zeros += __shfl_xor(zeros, o, 64);
    zeros_all += zeros;
    emit_compacted_bits(s_win[wv], comp, cnt, pos, p.out_bits, lane);
  }
  if (p.zero_count && zeros_all && lane == 0) atomicAdd(p.zero_count, zeros_all);
}

// The same for a wave-packed batch group (FilterParams::group_bits_at): the selection words and bases are indexed by wave
// slot -- slot s = (batch * wpb + wave of the batch) * R + group of the wave -- and every batch has its own bitmap (none:
// every bit is 1); the output is ONE joined bitmap, batch b's bits are those of its output rows (per-batch results are
// Arrow slices of it, like the joined Utf8 column).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void bit_compact_group_kernel(const BitCompactGroupParams p) {
  constexpr int NW = BLOCK / 64;
  __shared__ uint32_t s_win[NW][132];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int64_t spb = (int64_t)p.wpb * (p.rows_per_wave >> 6);     // slots per batch
  const int64_t nslots = spb * p.nb;
  const int64_t nchunks = (nslots + 63) >> 6;
  u64 zeros_all = 0;
  for (int64_t chunk = (int64_t)blockIdx.x * NW + wv; chunk < nchunks; chunk += (int64_t)gridDim.x * NW) {
    const int64_t sl = chunk * 64 + lane;
    u64 comp = 0, pos = 0; int cnt = 0;
    if (sl < nslots) {
      const int64_t b = sl / spb;
      const int64_t w0 = (sl - b * spb) << 6;                      // first row of the slot inside its batch
      const u64* brow = p.table + b * p.stride;
      const u64 act = active_mask(w0, (int64_t)brow[0]);
      if (act) {
        const u64 m = p.sel_mask[sl] & act;
        cnt = __popcll(m);
        if (cnt) {
          const void* bits = (const void*)brow[p.word_ptr];
          comp = pext64(bits ? load_bits64(bits, (int64_t)brow[p.word_off] + w0, act) : act, m);
          pos = p.grp_base[sl];
        }
      }
    }
    unsigned zeros = (unsigned)(cnt - __popcll(comp));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zeros += __shfl_xor(zeros, o, 64);
    zeros_all += zeros;
    emit_compacted_bits(s_win[wv], comp, cnt, pos, p.out_bits, lane);
  }
  if (p.zero_count && zeros_all && lane == 0) atomicAdd(p.zero_count, zeros_all);
}

// ------------------------------------------------------------------------------------------------
// utf8_filter_kernel: arrow-select filter of a Utf8 column in one pass.  Tile = 64*G*NW rows.
//   pass 1: every lane loads the offset of its row (coalesced), lengths of the selected rows come from the
//           neighbour lane; wave / tile byte totals
//   chained scan over the tiles' byte totals (same look-back as the row scan)
//   pass 2: per 64-row group: exclusive scan of the lengths -> new offsets (written at the group's output row
//           position, known from the main kernel's grp_base), then the bytes are copied: one lane per row in 4-byte
//           chunks for short strings, eight lanes per row in 16-byte chunks for long ones (copy_rows_wide)
// ------------------------------------------------------------------------------------------------

// Long strings of one 64-row group: lane k holds (src offset, length, dst offset) of the selected row of rank k.
// Eight lanes per row move 16-byte chunks (128 bytes per pass), eight rows per step; the first pass of TWO steps is
// loaded before anything is stored, so 16 rows' loads are in flight together instead of one dependent load/store pair
// per two rows.
__device__ __forceinline__ void copy_rows_wide(const uint8_t* __restrict__ in_data, uint8_t* __restrict__ out_base,
                                               int rs, int rl, int rd, int cnt, int lane) {
  const int sub = lane >> 3, sl = lane & 7;
  for (int k = 0; k < cnt; k += 16) {
    const int ra = k + sub, rb = k + 8 + sub;
    const int ia = ra < cnt ? ra : cnt - 1, ib = rb < cnt ? rb : cnt - 1;
    // (all lanes take part in every shuffle: a lane disabled at the time would not supply its value)
    const int sa = __shfl(rs, ia, 64), da = __shfl(rd, ia, 64), sb = __shfl(rs, ib, 64), db = __shfl(rd, ib, 64);
    int la = __shfl(rl, ia, 64), lb = __shfl(rl, ib, 64);
    if (ra >= cnt) la = 0;
    if (rb >= cnt) lb = 0;
    const uint8_t* pa = in_data + sa; uint8_t* qa = out_base + da;
    const uint8_t* pb = in_data + sb; uint8_t* qb = out_base + db;
    uint4 wa, wb;
    const bool ha = sl * 16 + 16 <= la, hb = sl * 16 + 16 <= lb;
    if (ha) __builtin_memcpy(&wa, pa + sl * 16, 16);
    if (hb) __builtin_memcpy(&wb, pb + sl * 16, 16);
    if (ha) __builtin_memcpy(qa + sl * 16, &wa, 16);
    if (hb) __builtin_memcpy(qb + sl * 16, &wb, 16);
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // further passes of rows longer than 128 bytes, then the tail (< 16 bytes)
      const uint8_t* ps = h ? pb : pa; uint8_t* qd = h ? qb : qa; const int len = h ? lb : la;
      for (int b = 128 + sl * 16; b + 16 <= len; b += 128) { uint4 w; __builtin_memcpy(&w, ps + b, 16); __builtin_memcpy(qd + b, &w, 16); }
      const int tail = len & ~15, rem = len & 15;
      if (sl * 4 + 4 <= rem) { uint32_t w; __builtin_memcpy(&w, ps + tail + sl * 4, 4); __builtin_memcpy(qd + tail + sl * 4, &w, 4); }
      if (sl < (rem & 3)) qd[tail + (rem & ~3) + sl] = ps[tail + (rem & ~3) + sl];
    }
  }
}

template <int BLOCK, int G>
__global__ __launch_bounds__(BLOCK) void utf8_filter_kernel(const Utf8Params p) {
  constexpr int NW = BLOCK / 64;
  constexpr int64_t TILE = (int64_t)64 * G * NW;
  __shared__ u64 s_wave_bytes[NW];
  __shared__ u64 s_base_bytes;
  __shared__ int64_t s_tile;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int64_t ntiles = (p.nrows + TILE - 1) / TILE;
  while (true) {
    const int lane = fresh_lane(), tid = wv * 64 + lane;   // (per tile: nothing derived from it is held across tiles)
    if (tid == 0) s_tile = (int64_t)atomicAdd(p.ticket, 1u);
    __syncthreads();
    const int64_t tile = uniform64(s_tile);
    if (tile >= ntiles) break;
    const int64_t w0 = tile * TILE + (int64_t)wv * 64 * G;
    int32_t src[G]; uint32_t len[G]; u64 msk[G]; u64 gbase[G];
    u64 bytes = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t r0 = w0 + 64 * g;
      const u64 act = active_mask(r0, p.nrows);
      msk[g] = act ? (p.sel_mask[r0 >> 6] & act) : 0ULL;
      src[g] = 0; len[g] = 0; gbase[g] = 0;
      if (msk[g]) {   // uniform
        const int64_t last = p.nrows - r0 < 64 ? p.nrows - r0 : 64;             // offsets r0 .. r0+last are valid
        const int32_t o = p.in_offsets[r0 + (lane < last ? lane : last)];
        const int32_t oend = p.in_offsets[r0 + last];                            // broadcast load
        gbase[g] = p.grp_base[r0 >> 6];                                          // needed in pass 2: fetch it now
        int32_t nxt = __shfl_down(o, 1, 64);
        if (lane + 1 >= last) nxt = oend;
        src[g] = o;
        len[g] = ((msk[g] >> lane) & 1) ? (uint32_t)(nxt - o) : 0u;
      }
      bytes += len[g];
    }
    bytes = wave_sum(bytes);
    if (lane == 0) s_wave_bytes[wv] = bytes;
    __syncthreads();
    if (wv == 0) {
      u64 tb = 0;
      for (int w = 0; w < NW; ++w) tb += s_wave_bytes[w];
      u64 excl = 0;
      if (lane == 0) st_store(&p.byte_status[tile], (tile == 0 ? ST_INC : ST_AGG) | tb);
      if (tile > 0) {
        excl = lookback_exclusive_wide(p.byte_status, tile, 0, lane);
        if (lane == 0) st_store(&p.byte_status[tile], ST_INC | (excl + tb));
      }
      if (lane == 0) {
        s_base_bytes = excl;
        if (tile == ntiles - 1) { *p.total_bytes = excl + tb; p.out_offsets[p.rows_out] = (int32_t)(excl + tb); }
      }
    }
    __syncthreads();
    u64 boff = s_base_bytes;
    for (int w = 0; w < wv; ++w) boff += s_wave_bytes[w];
    boff = (u64)uniform64((int64_t)boff);
    // ---- pass 2a: new offsets; every group's byte base and every row's position inside it ---------------------
    uint32_t dsto[G]; u64 gboff[G]; uint32_t longg = 0;   // bit g of longg: the group takes the long-string path
#pragma unroll
    for (int g = 0; g < G; ++g) {
      dsto[g] = 0; gboff[g] = boff;
      if (!msk[g]) continue;
      const bool sel = (msk[g] >> lane) & 1;
      uint32_t inc = len[g];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
      dsto[g] = inc - len[g];                                     // byte position of this row inside the group's output
      const uint32_t group_bytes = __shfl(inc, 63, 64);
      if (sel) p.out_offsets[gbase[g] + lane_rank(msk[g])] = (int32_t)(boff + dsto[g]);
      if (group_bytes > (uint32_t)__popcll(msk[g]) * 24u) longg |= 1u << g;
      boff += group_bytes;
    }
    // ---- pass 2b: short strings, one lane per row.  The first 16 bytes of every selected row of ALL groups are
    // loaded before anything is stored (8 x 4 loads in flight per lane instead of one dependent load/store pair at a
    // time); whatever is longer than 16 bytes, and the 1-3 tail bytes, follow in a per-lane loop. --------------------
    {
      uint32_t w4[G][4];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const uint8_t* sp = p.in_data + src[g];
        const bool shortg = msk[g] && !((longg >> g) & 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          w4[g][q] = 0;
          if (shortg && (uint32_t)(4 * q + 4) <= len[g]) __builtin_memcpy(&w4[g][q], sp + 4 * q, 4);
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const bool shortg = msk[g] && !((longg >> g) & 1);
        if (!shortg) continue;
        const uint8_t* sp = p.in_data + src[g];
        uint8_t* dp = p.out_data + gboff[g] + dsto[g];
        const int l = (int)len[g];
#pragma unroll
        for (int q = 0; q < 4; ++q) if (4 * q + 4 <= l) __builtin_memcpy(dp + 4 * q, &w4[g][q], 4);
        int b = l < 16 ? (l & ~3) : 16;
        for (; b + 4 <= l; b += 4) { uint32_t w; __builtin_memcpy(&w, sp + b, 4); __builtin_memcpy(dp + b, &w, 4); }
        for (; b < l; ++b) dp[b] = sp[b];
      }
    }
    // ---- pass 2c: long strings: (src, len, dst) of the selected rows in rank order, half a wave per row ------------
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      if (!((longg >> g) & 1)) continue;
      const bool sel = (msk[g] >> lane) & 1;
      const int cnt = __popcll(msk[g]);
      uint8_t* gdst = p.out_data + gboff[g];
      const unsigned dl = sel ? lane_rank(msk[g]) : 63u - lane_rank(~msk[g]);
      const int rs = __builtin_amdgcn_ds_permute((int)(dl << 2), src[g]);
      const int rl = __builtin_amdgcn_ds_permute((int)(dl << 2), (int)len[g]);
      const int rd = __builtin_amdgcn_ds_permute((int)(dl << 2), (int)dsto[g]);
      copy_rows_wide(p.in_data, gdst, rs, rl, rd, cnt, lane);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Utf8 filter, kernel 1: byte length of every selected row -> new offsets.  A chained scan over tiles
// of 64*G*NW rows carries the running byte total; row positions come from grp_base.
// ------------------------------------------------------------------------------------------------
template <int BLOCK, int G>
__global__ __launch_bounds__(BLOCK) void utf8_offsets_kernel(const Utf8Params p) {
  constexpr int NW = BLOCK / 64;
  constexpr int64_t TILE = (int64_t)64 * G * NW;
  __shared__ u64 s_wave_bytes[NW];
  __shared__ u64 s_base_bytes;
  __shared__ int64_t s_tile;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t ntiles = (p.nrows + TILE - 1) / TILE;
  while (true) {
    if (tid == 0) s_tile = (int64_t)atomicAdd(p.ticket, 1u);
    __syncthreads();
    const int64_t tile = s_tile;
    if (tile >= ntiles) break;
    const int64_t w0 = tile * TILE + (int64_t)wv * 64 * G;
    uint32_t len[G]; u64 msk[G];     // a row is < 2 GiB (int32 offsets): 32-bit lengths, 64-bit totals
    u64 bytes = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t r0 = w0 + 64 * g;
      const u64 act = active_mask(r0, p.nrows);
      msk[g] = act ? (p.sel_mask[r0 >> 6] & act) : 0ULL;
      len[g] = 0;
      if ((msk[g] >> lane) & 1) { const int32_t* o = p.in_offsets + r0 + lane; len[g] = (uint32_t)(o[1] - o[0]); }
      bytes += len[g];
    }
    bytes = wave_sum(bytes);
    if (lane == 0) s_wave_bytes[wv] = bytes;
    __syncthreads();
    if (wv == 0) {
      u64 tb = 0;
      for (int w = 0; w < NW; ++w) tb += s_wave_bytes[w];
      u64 excl = 0;
      if (lane == 0) st_store(&p.byte_status[tile], (tile == 0 ? ST_INC : ST_AGG) | tb);
      if (tile > 0) {
        excl = lookback_exclusive_wide(p.byte_status, tile, 0, lane);
        if (lane == 0) st_store(&p.byte_status[tile], ST_INC | (excl + tb));
      }
      if (lane == 0) {
        s_base_bytes = excl;
        if (tile == ntiles - 1) { *p.total_bytes = excl + tb; p.out_offsets[p.rows_out] = (int32_t)(excl + tb); }
      }
    }
    __syncthreads();
    u64 boff = s_base_bytes;
    for (int w = 0; w < wv; ++w) boff += s_wave_bytes[w];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t r0 = w0 + 64 * g;
      if (msk[g]) {
        uint32_t inc = len[g];   // a 64-row group stays below 2^32 bytes only if rows do: guard with the 64-bit total
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if ((msk[g] >> lane) & 1) p.out_offsets[p.grp_base[r0 >> 6] + lane_rank(msk[g])] = (int32_t)(boff + inc - len[g]);
        boff += __shfl(inc, 63, 64);
      }
    }
    __syncthreads();
  }
}

// Utf8 filter, kernel 2: copy the bytes of the selected rows.  One wave per 64 rows.  Consecutive selected rows are
// contiguous in the input AND in the output, so where the selection comes in runs (range predicates, high selectivity)
// the unit of copying is the RUN: the whole wave moves it with 16-byte chunks, 1 KiB per instruction (global memory takes
// unaligned dwordx4 accesses) -- with the 100-character sample strings and most rows kept a 64-row group is one run of
// 6.4 KB (measured, config 4: 1.60 -> 1.18 ms).  A scattered selection is copied row by row (copy_rows_wide), which is
// faster there (runs of one or two rows would leave most of the wave idle).  Short strings -- a few bytes per row --
// take the fused utf8_filter_kernel instead.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void utf8_copy_kernel(const Utf8Params p) {
  constexpr int NW = BLOCK / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t ngroups = (p.nrows + 63) >> 6;
  const int64_t stride = (int64_t)gridDim.x * NW;
  for (int64_t g = (int64_t)blockIdx.x * NW + wv; g < ngroups; g += stride) {
    const u64 m = p.sel_mask[g] & active_mask(g << 6, p.nrows);
    if (!m) continue;
    // offsets of the group's 65 row boundaries (lane l: boundary l; the last one is broadcast), output byte base
    const int64_t r0 = g << 6;
    const int64_t last = p.nrows - r0 < 64 ? p.nrows - r0 : 64;
    const int32_t o = p.in_offsets[r0 + (lane < last ? lane : last)];
    const int32_t oend = p.in_offsets[r0 + last];
    int64_t dcur = (int64_t)p.out_offsets[p.grp_base[g]];
    const int cnt = __popcll(m);
    const int runs = __popcll(m & ~(m << 1));
    if (runs * 4 > cnt) {
      // scattered selection (runs of fewer than four rows on average): row by row, eight lanes per row, sixteen rows in
      // flight -- lane k describes the selected row of rank k
      int32_t onext = __shfl_down(o, 1, 64);
      if (lane + 1 >= (int)last) onext = oend;
      const bool sel = (m >> lane) & 1;
      const unsigned dstl = sel ? lane_rank(m) : 63u - lane_rank(~m);
      const int src0 = __builtin_amdgcn_ds_permute((int)(dstl << 2), o);
      const int len0 = __builtin_amdgcn_ds_permute((int)(dstl << 2), onext - o);
      int inc = lane < cnt ? len0 : 0;
      const int mylen = inc;
#pragma unroll
      for (int k = 1; k < 64; k <<= 1) { int t = __shfl_up(inc, k, 64); if (lane >= k) inc += t; }
      copy_rows_wide(p.in_data, p.out_data + dcur, src0, len0, inc - mylen, cnt, lane);
      continue;
    }
    u64 mm = m;
    while (mm) {   // wave-uniform: one iteration per run of selected rows
      const int first = __builtin_ctzll(mm);
      const u64 rest = ~(mm >> first);
      const int run = rest ? __builtin_ctzll(rest) : 64 - first;
      const int32_t sb = __shfl(o, first, 64);
      const int endrow = first + run;
      const int32_t se = endrow >= (int)last ? oend : __shfl(o, endrow < 64 ? endrow : 63, 64);
      const int64_t nbytes = (int64_t)se - sb;
      const uint8_t* src = p.in_data + sb;
      uint8_t* dst = p.out_data + dcur;
      int64_t b = (int64_t)lane * 16;
      for (; b + 16 + 3072 <= nbytes; b += 4096) {   // four chunks per lane in flight
        uint4 w0, w1, w2, w3;
        __builtin_memcpy(&w0, src + b, 16); __builtin_memcpy(&w1, src + b + 1024, 16);
        __builtin_memcpy(&w2, src + b + 2048, 16); __builtin_memcpy(&w3, src + b + 3072, 16);
        __builtin_memcpy(dst + b, &w0, 16); __builtin_memcpy(dst + b + 1024, &w1, 16);
        __builtin_memcpy(dst + b + 2048, &w2, 16); __builtin_memcpy(dst + b + 3072, &w3, 16);
      }
      for (; b + 16 <= nbytes; b += 1024) { uint4 w; __builtin_memcpy(&w, src + b, 16); __builtin_memcpy(dst + b, &w, 16); }
      const int64_t tail = nbytes & ~(int64_t)15;
      if (lane < (int)(nbytes & 15)) dst[tail + lane] = src[tail + lane];
      dcur += nbytes;
      mm = run + first >= 64 ? 0 : (mm >> (first + run)) << (first + run);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (extern, used by engine.cpp)
// ------------------------------------------------------------------------------------------------
// tile_kind 0: 1024 threads x 16 rows, 32-bit types only, no numeric temporaries (the streaming path), 1 stash slot
// tile_kind 1:  256 threads x  8 rows, 32-bit types only, 2 stash slots
// tile_kind 2:  256 threads x  8 rows, all types, numeric temporaries in LDS (general path), 1 stash slot
// partial: the launch may contain a tile that is not completely inside the batch
// The FASTK instantiations serve the pre-decoded programs (32-bit types only: tile kinds 0 and 1), the others the generic
// interpreter.
// The file is compiled as four translation units (Makefile: -DCHQ_TU=1..4, in parallel), each instantiating its share of the
// kernel templates above -- one unit took 3.5 minutes; without CHQ_TU everything is compiled in one.
//   1: filter_fused_kernel without Utf8 columns   2: filter_fused_kernel with Utf8 columns
//   3: filter_project_kernel, project_kernel       4: bit / Utf8 follow-up kernels, joins, IPC helpers
#ifndef CHQ_TU
#define CHQ_TU 0
#endif
hipError_t launch_filter_utf8(const FilterParams& p, int tile_kind, bool partial, int grid, hipStream_t stream);

#if CHQ_TU == 9   // development: ONE instantiation, for register-pressure experiments (scripts/kernel_resources.sh, -DCHQ_EXP_*)
#ifndef CHQ_EXP_ARGS
#define CHQ_EXP_ARGS 1024, 16, false, 0, true, 1, false, 0
#endif
template __global__ void filter_fused_kernel<CHQ_EXP_ARGS>(const FilterParams p);
#endif

#if CHQ_TU == 0 || CHQ_TU == 1
hipError_t launch_filter(const FilterParams& p, int tile_kind, bool partial, int grid, hipStream_t stream) {
  if (p.n_utf8 > 0) return launch_filter_utf8(p, tile_kind, partial, grid, stream);
  const bool fast = p.pb.fast_kind != FAST_NONE && tile_kind != 2;
#define LF(B, RR, W, T, S, F) do { if (partial) hipLaunchKernelGGL((filter_fused_kernel<B, RR, W, T, true, S, F>), dim3(grid), dim3(B), 0, stream, p); \
                                   else hipLaunchKernelGGL((filter_fused_kernel<B, RR, W, T, false, S, F>), dim3(grid), dim3(B), 0, stream, p); } while (0)
  switch (tile_kind) {
    case 0: if (fast) LF(1024, 16, false, 0, STASH_SLOTS_K0, true); else LF(1024, 16, false, 0, STASH_SLOTS_K0, false); break;
    case 1: if (fast) LF(256, 8, false, 0, STASH_SLOTS_K1, true); else LF(256, 8, false, 0, STASH_SLOTS_K1, false); break;
    default: LF(256, 8, true, MAX_NUM_TEMPS, STASH_SLOTS_K2, false); break;
  }
#undef LF
  return hipGetLastError();
}
#endif

#if CHQ_TU == 0 || CHQ_TU == 2
// launches with Utf8 columns filtered in the same pass (tile kinds 0 and 1; batch groups only in the PARTIAL instantiation)
hipError_t launch_filter_utf8(const FilterParams& p, int tile_kind, bool partial, int grid, hipStream_t stream) {
  const bool fast = p.pb.fast_kind != FAST_NONE && tile_kind != 2;
#define LFU(B, RR, S, F) do { if (partial) hipLaunchKernelGGL((filter_fused_kernel<B, RR, false, 0, true, S, F, MAX_FOLD_UTF8>), dim3(grid), dim3(B), 0, stream, p); \
                              else hipLaunchKernelGGL((filter_fused_kernel<B, RR, false, 0, false, S, F, MAX_FOLD_UTF8>), dim3(grid), dim3(B), 0, stream, p); } while (0)
  if (tile_kind == 2 || p.n_utf8 > MAX_FOLD_UTF8 || (p.group && !partial)) return hipErrorInvalidValue;
  if (tile_kind == 0) { if (fast) LFU(1024, 16, STASH_SLOTS_K0, true); else LFU(1024, 16, STASH_SLOTS_K0, false); }
  else { if (fast) LFU(256, 8, STASH_SLOTS_K1, true); else LFU(256, 8, STASH_SLOTS_K1, false); }
#undef LFU
  return hipGetLastError();
}
#endif

#if CHQ_TU == 0 || CHQ_TU == 3
hipError_t launch_filter_project(const FusedParams& p, int tile_kind, int grid, hipStream_t stream) {
  const bool fast = tile_kind != 2 && p.pred.fast_kind != FAST_NONE && (p.n_proj == 0 || p.proj.fast_kind == FAST_UOPS);
  switch (tile_kind) {
    case 0:
      if (fast) hipLaunchKernelGGL((filter_project_kernel<1024, 16, false, 0, true>), dim3(grid), dim3(1024), 0, stream, p);
      else hipLaunchKernelGGL((filter_project_kernel<1024, 16, false, 0, false>), dim3(grid), dim3(1024), 0, stream, p);
      break;
    case 1:
      if (fast) hipLaunchKernelGGL((filter_project_kernel<256, 8, false, 0, true>), dim3(grid), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((filter_project_kernel<256, 8, false, 0, false>), dim3(grid), dim3(256), 0, stream, p);
      break;
    default: hipLaunchKernelGGL((filter_project_kernel<256, 8, true, MAX_NUM_TEMPS, false>), dim3(grid), dim3(256), 0, stream, p); break;
  }
  return hipGetLastError();
}
hipError_t launch_project(const ProjectParams& p, int tile_kind, bool partial, int grid, hipStream_t stream) {
  const bool fast = p.pb.fast_kind == FAST_UOPS && tile_kind != 2;
#define LP(B, RR, W, T, F) do { if (partial) hipLaunchKernelGGL((project_kernel<B, RR, W, T, true, F>), dim3(grid), dim3(B), 0, stream, p); \
                                else hipLaunchKernelGGL((project_kernel<B, RR, W, T, false, F>), dim3(grid), dim3(B), 0, stream, p); } while (0)
  switch (tile_kind) {
    case 0: if (fast) LP(1024, 16, false, 0, true); else LP(1024, 16, false, 0, false); break;
    case 1: if (fast) LP(256, 8, false, 0, true); else LP(256, 8, false, 0, false); break;
    default: LP(256, 8, true, MAX_NUM_TEMPS, false); break;
  }
#undef LP
  return hipGetLastError();
}
#endif

#if CHQ_TU == 0 || CHQ_TU == 4
hipError_t launch_bit_compact(const BitCompactParams& p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL((bit_compact_kernel<256, 8>), dim3(grid), dim3(256), 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_bit_compact_group(const BitCompactGroupParams& p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL((bit_compact_group_kernel<256>), dim3(grid), dim3(256), 0, stream, p);
  return hipGetLastError();
}
__global__ void gather_i32_kernel(const GatherParams p) {
  const int i = threadIdx.x;
  if (i < p.n) p.dst[i] = *p.src[i];
}
// out[i] = selected rows among input rows [0, starts[i]): where the output of a batch concatenated at row starts[i] begins
__global__ void split_bounds_kernel(const SplitBoundsParams p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  const int64_t r = p.starts[i];
  if (r >= p.nrows) { p.out[i] = *p.total; return; }
  const int64_t g = r >> 6;
  p.out[i] = p.grp_base[g] + (u64)__popcll(p.sel_mask[g] & ((1ULL << (r & 63)) - 1ULL));
}
hipError_t launch_split_bounds(const SplitBoundsParams& p, hipStream_t stream) {
  if (p.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(split_bounds_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, stream, p);
  return hipGetLastError();
}
__global__ void gather_status_kernel(const GatherStatusParams p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < p.n) p.dst[i] = ST_VAL(p.status[p.idx[i]]);
}
hipError_t launch_gather_status(const GatherStatusParams& p, hipStream_t stream) {
  if (p.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_status_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_gather_i32(const GatherParams& p, hipStream_t stream) {
  hipLaunchKernelGGL(gather_i32_kernel, dim3(1), dim3(64), 0, stream, p);
  return hipGetLastError();
}
// ------------------------------------------------------------------------------------------------
// Joining the batches of a device-resident group into one batch (filter_task.rs:78-126 hoisted below the boundary for
// groups with Utf8 / Boolean / nullable columns: the joined batch goes through the ordinary single-batch kernels, the
// outputs are cut at the batch boundaries afterwards).  One workgroup per batch, grid-stride.
// ------------------------------------------------------------------------------------------------
template <typename TY>
__global__ __launch_bounds__(256) void concat_fixed_kernel(const ConcatParams p) {
  for (int64_t b = blockIdx.x; b < p.nb; b += gridDim.x) {
    const int64_t r0 = p.row_at[b], n = p.row_at[b + 1] - r0;
    const TY* src = (const TY*)p.src[b];
    TY* dst = (TY*)p.dst + r0;
    int64_t i = threadIdx.x;
    for (; i + 768 < n; i += 1024) {   // four loads in flight per thread
      const TY a = src[i], bq = src[i + 256], c = src[i + 512], d = src[i + 768];
      dst[i] = a; dst[i + 256] = bq; dst[i + 512] = c; dst[i + 768] = d;
    }
    for (; i < n; i += 256) dst[i] = src[i];
  }
}
__global__ __launch_bounds__(256) void gather_ends_kernel(const ConcatParams p) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.nb) return;
  const int32_t* offs = (const int32_t*)p.src[b];
  const int64_t n = p.row_at[b + 1] - p.row_at[b];
  p.ends[2 * b] = offs ? offs[0] : 0;
  p.ends[2 * b + 1] = offs ? offs[n] : 0;
}
__global__ __launch_bounds__(256) void concat_utf8_kernel(const ConcatParams p) {
  for (int64_t b = blockIdx.x; b < p.nb; b += gridDim.x) {
    const int64_t r0 = p.row_at[b], n = p.row_at[b + 1] - r0;
    const int64_t y0 = p.byte_at[b], nbytes = p.byte_at[b + 1] - y0;
    const int32_t* soff = (const int32_t*)p.src[b];
    int32_t* doff = (int32_t*)p.dst + r0;
    const int32_t first = (soff && n > 0) ? soff[0] : 0;
    const int32_t shift = (int32_t)(y0 - first);
    for (int64_t i = threadIdx.x; i < n; i += 256) doff[i] = soff[i] + shift;
    if (b == p.nb - 1 && threadIdx.x == 0) doff[n] = (int32_t)(y0 + nbytes);   // the closing offset of the joined column
    const uint8_t* sd = (const uint8_t*)p.aux[b] + first;
    uint8_t* dd = (uint8_t*)p.dst2 + y0;
    const int64_t words = nbytes >> 2;
    for (int64_t i = threadIdx.x; i < words; i += 256) { uint32_t w; __builtin_memcpy(&w, sd + 4 * i, 4); __builtin_memcpy(dd + 4 * i, &w, 4); }
    if (threadIdx.x < (nbytes & 3)) dd[4 * words + threadIdx.x] = sd[4 * words + threadIdx.x];
  }
}
// bits [bitoff[b], bitoff[b] + n_b) of every batch's bitmap appended at bit row_at[b] of the zero-initialised output; a batch
// without a bitmap (src 0) contributes ones.  32 bits per thread, merged with atomicOr (neighbouring batches share words).
__global__ __launch_bounds__(256) void concat_bits_kernel(const ConcatParams p) {
  for (int64_t b = blockIdx.x; b < p.nb; b += gridDim.x) {
    const int64_t r0 = p.row_at[b], n = p.row_at[b + 1] - r0;
    const uint8_t* src = (const uint8_t*)p.src[b];
    const int64_t so = p.bitoff[b];
    for (int64_t k = threadIdx.x; 32 * k < n; k += 256) {
      const int cnt = (int)(n - 32 * k < 32 ? n - 32 * k : 32);
      uint32_t v = 0xffffffffu;
      if (src) {
        const int64_t sb = so + 32 * k;               // first source bit
        const uint8_t* q = src + (sb >> 3);
        const int sh = (int)(sb & 7);
        const int need = (sh + cnt + 7) >> 3;          // 1..5 bytes hold the bits
        u64 raw = 0;
        for (int t = 0; t < need; ++t) raw |= (u64)q[t] << (8 * t);
        v = (uint32_t)(raw >> sh);
      }
      if (cnt < 32) v &= (1u << cnt) - 1u;
      const int64_t d = r0 + 32 * k;
      uint32_t* dw = (uint32_t*)p.dst + (d >> 5);
      const int ds = (int)(d & 31);
      if (v << ds) atomicOr(dw, v << ds);
      if (ds && (v >> (32 - ds))) atomicOr(dw + 1, v >> (32 - ds));
    }
  }
}
hipError_t launch_concat(const ConcatParams& p, int kind, int grid, hipStream_t stream) {   // kind: 0 fixed, 1 ends, 2 utf8, 3 bits
  if (p.nb <= 0) return hipSuccess;
  switch (kind) {
    case 0:
      switch (p.width) {
        case 1: hipLaunchKernelGGL((concat_fixed_kernel<uint8_t>), dim3(grid), dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((concat_fixed_kernel<uint16_t>), dim3(grid), dim3(256), 0, stream, p); break;
        case 4: hipLaunchKernelGGL((concat_fixed_kernel<uint32_t>), dim3(grid), dim3(256), 0, stream, p); break;
        case 8: hipLaunchKernelGGL((concat_fixed_kernel<uint2>), dim3(grid), dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL((concat_fixed_kernel<uint4>), dim3(grid), dim3(256), 0, stream, p); break;
      }
      break;
    case 1: hipLaunchKernelGGL(gather_ends_kernel, dim3((unsigned)((p.nb + 255) / 256)), dim3(256), 0, stream, p); break;
    case 2: hipLaunchKernelGGL(concat_utf8_kernel, dim3(grid), dim3(256), 0, stream, p); break;
    default: hipLaunchKernelGGL(concat_bits_kernel, dim3(grid), dim3(256), 0, stream, p); break;
  }
  return hipGetLastError();
}

// ---- Arrow IPC body assembly (ipc.cpp): the pieces a plain device copy cannot place --------------------------------------
__global__ __launch_bounds__(256) void rebase_offsets_kernel(const int32_t* in, int32_t* out, int64_t n) {
  const int32_t first = in[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = in[i] - first;
}
// out word k = bits [bit_offset + 32 k, +32) of `in` (bits past `nbits` are zero): a sliced bitmap rebased to bit 0
__global__ __launch_bounds__(256) void bit_shift_copy_kernel(const uint8_t* in, int64_t bit_offset, int64_t nbits, uint32_t* out) {
  const int64_t words = (nbits + 31) >> 5;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < words; k += (int64_t)gridDim.x * 256) {
    const int cnt = (int)(nbits - 32 * k < 32 ? nbits - 32 * k : 32);
    const int64_t sb = bit_offset + 32 * k;
    const uint8_t* q = in + (sb >> 3);
    const int sh = (int)(sb & 7);
    const int need = (sh + cnt + 7) >> 3;
    u64 raw = 0;
    for (int t = 0; t < need; ++t) raw |= (u64)q[t] << (8 * t);
    uint32_t v = (uint32_t)(raw >> sh);
    if (cnt < 32) v &= (1u << cnt) - 1u;
    out[k] = v;
  }
}
__global__ __launch_bounds__(256) void count_bits_kernel(const uint8_t* in, int64_t bit_offset, int64_t nbits, u64* out) {
  u64 local = 0;
  const int64_t words = (nbits + 31) >> 5;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < words; k += (int64_t)gridDim.x * 256) {
    const int cnt = (int)(nbits - 32 * k < 32 ? nbits - 32 * k : 32);
    const int64_t sb = bit_offset + 32 * k;
    const uint8_t* q = in + (sb >> 3);
    const int sh = (int)(sb & 7);
    const int need = (sh + cnt + 7) >> 3;
    u64 raw = 0;
    for (int t = 0; t < need; ++t) raw |= (u64)q[t] << (8 * t);
    uint32_t v = (uint32_t)(raw >> sh);
    if (cnt < 32) v &= (1u << cnt) - 1u;
    local += __popc(v);
  }
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}
// flag |= 1 unless 0 <= offs[i] <= offs[i+1] <= data_len for every i in [0, n): offsets that arrived over the wire are checked
// completely before any kernel follows them (the reference's arrow StreamReader validates the whole offsets buffer)
__global__ __launch_bounds__(256) void validate_offsets_kernel(const int32_t* offs, int64_t n, int64_t data_len, uint32_t* flag) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int32_t a = offs[i], b = offs[i + 1];
    bad |= a < 0 || b < a || (int64_t)b > data_len;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
static int ipc_grid(int64_t items) { const int64_t g = (items + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
hipError_t launch_validate_offsets(const int32_t* offs, int64_t n, int64_t data_len, uint32_t* flag, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(validate_offsets_kernel, dim3(ipc_grid(n)), dim3(256), 0, stream, offs, n, data_len, flag);
  return hipGetLastError();
}
hipError_t launch_rebase_offsets(const int32_t* in, int32_t* out, int64_t n_plus_1, hipStream_t stream) {
  hipLaunchKernelGGL(rebase_offsets_kernel, dim3(ipc_grid(n_plus_1)), dim3(256), 0, stream, in, out, n_plus_1);
  return hipGetLastError();
}
hipError_t launch_bit_shift_copy(const uint8_t* in, int64_t bit_offset, int64_t nbits, uint32_t* out, hipStream_t stream) {
  hipLaunchKernelGGL(bit_shift_copy_kernel, dim3(ipc_grid((nbits + 31) / 32)), dim3(256), 0, stream, in, bit_offset, nbits, out);
  return hipGetLastError();
}
hipError_t launch_count_bits(const uint8_t* in, int64_t bit_offset, int64_t nbits, u64* out, hipStream_t stream) {
  hipLaunchKernelGGL(count_bits_kernel, dim3(ipc_grid((nbits + 31) / 32)), dim3(256), 0, stream, in, bit_offset, nbits, out);
  return hipGetLastError();
}

hipError_t launch_utf8_filter(const Utf8Params& p, int grid, hipStream_t stream) {   // 8192-row tiles
  hipLaunchKernelGGL((utf8_filter_kernel<1024, 8>), dim3(grid), dim3(1024), 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_utf8_offsets(const Utf8Params& p, int grid, hipStream_t stream) {   // 2048-row tiles
  hipLaunchKernelGGL((utf8_offsets_kernel<256, 8>), dim3(grid), dim3(256), 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_utf8_copy(const Utf8Params& p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL((utf8_copy_kernel<256>), dim3(grid), dim3(256), 0, stream, p);
  return hipGetLastError();
}

#endif

}  // namespace chq
