// typed_ops.hip -- gfx950 kernels for the corners of the reference's type coverage that are not instructions of the
// accumulator machine (device_program.h): Decimal128 comparisons and the Utf8 -> Boolean cast under AND / OR.  Both are
// one pass over their input (32 B/row resp. the string bytes), one thread per row, one 64-bit ballot per wave as output.
#include <hip/hip_runtime.h>

#include "device_program.h"
#include "typed_ops.h"

namespace chq {
namespace {
__device__ __forceinline__ bool bit_at(const void* bits, int64_t i) {
  return (((const uint8_t*)bits)[i >> 3] >> (i & 7)) & 1;
}

// arrow-ord cmp::{eq,neq,lt,lt_eq,gt,gt_eq} on i128 natives (RU/compute_value.rs:118-209 after the `left == right` arm of
// get_common_type, :355).  Validity = both valid.
__global__ __launch_bounds__(256) void cmp128_kernel(const Cmp128Params p) {
  const int lane = threadIdx.x & 63;
  for (int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) & ~63LL; base < p.nrows; base += (int64_t)gridDim.x * 256) {
    const int64_t i = base + lane;
    bool r = false, valid = false;
    if (i < p.nrows) {
      valid = (!p.a_validity || bit_at(p.a_validity, p.a_validity_offset + i)) && (!p.b_validity || bit_at(p.b_validity, p.b_validity_offset + i));
      // (Arrow guarantees 8-byte alignment of the buffer only through its producer; 4-byte loads are always safe)
      const uint32_t* pa = (const uint32_t*)p.a + 4 * i;
      const uint32_t* pb = (const uint32_t*)p.b + 4 * i;
      const u64 alo = pa[0] | ((u64)pa[1] << 32), blo = pb[0] | ((u64)pb[1] << 32);
      const int64_t ahi = (int64_t)(pa[2] | ((u64)pa[3] << 32)), bhi = (int64_t)(pb[2] | ((u64)pb[3] << 32));
      const bool eq = alo == blo && ahi == bhi;
      const bool lt = ahi < bhi || (ahi == bhi && alo < blo);
      switch (p.op) {
        case OP_EQ: r = eq; break; case OP_NE: r = !eq; break; case OP_LT: r = lt; break;
        case OP_LE: r = lt || eq; break; case OP_GT: r = !lt && !eq; break; default: r = !lt; break;
      }
    }
    const u64 vm = __ballot(valid), rm = __ballot(r && valid);
    if (lane == 0) {
      p.out_bits[base >> 6] = rm;
      p.out_validity[base >> 6] = vm;
      const int64_t rows = p.nrows - base < 64 ? p.nrows - base : 64;
      const unsigned nulls = (unsigned)rows - (unsigned)__popcll(vm);
      if (nulls) atomicAdd(p.null_count, (u64)nulls);
    }
  }
}

// Rust's char::is_whitespace (Unicode White_Space): byte length of the white-space character starting at p (0 = none)
__device__ __forceinline__ int ws_at(const uint8_t* p, int n) {
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

// value.to_ascii_lowercase().trim() matched against arrow-cast's spellings: 1 / 0 / -1
__device__ int parse_bool(const uint8_t* p, int n) {
  int w;
  while (n > 0 && (w = ws_at(p, n)) > 0) { p += w; n -= w; }
  for (bool again = true; again && n > 0;) {
    again = false;
    for (w = 1; w <= 3 && w <= n; ++w)
      if (ws_at(p + n - w, w) == w) { n -= w; again = true; break; }
  }
  if (n < 1 || n > 5) return -1;
  // the trimmed spelling as one little-endian integer of lower-cased bytes
  u64 key = 0;
  for (int i = 0; i < n; ++i) {
    const uint8_t c = p[i];
    key |= (u64)((c >= 'A' && c <= 'Z') ? c + 32 : c) << (8 * i);
  }
#define K1(a) ((u64)(a))
#define K2(a, b) (K1(a) | (u64)(b) << 8)
#define K3(a, b, c) (K2(a, b) | (u64)(c) << 16)
#define K4(a, b, c, d) (K3(a, b, c) | (u64)(d) << 24)
#define K5(a, b, c, d, e) (K4(a, b, c, d) | (u64)(e) << 32)
  switch (key) {
    case K1('t'): case K2('t', 'r'): case K3('t', 'r', 'u'): case K4('t', 'r', 'u', 'e'):
    case K1('y'): case K2('y', 'e'): case K3('y', 'e', 's'): case K2('o', 'n'): case K1('1'):
      return 1;
    case K1('f'): case K2('f', 'a'): case K3('f', 'a', 'l'): case K4('f', 'a', 'l', 's'): case K5('f', 'a', 'l', 's', 'e'):
    case K1('n'): case K2('n', 'o'): case K2('o', 'f'): case K3('o', 'f', 'f'): case K1('0'):
      return 0;
    default: return -1;
  }
#undef K1
#undef K2
#undef K3
#undef K4
#undef K5
}

__global__ __launch_bounds__(256) void utf8_to_bool_kernel(const Utf8ToBoolParams p) {
  const int lane = threadIdx.x & 63;
  for (int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) & ~63LL; base < p.nrows; base += (int64_t)gridDim.x * 256) {
    const int64_t i = base + lane;
    int v = -1;
    if (i < p.nrows && (!p.validity || bit_at(p.validity, p.validity_offset + i))) {
      const int32_t b = p.offsets[i], e = p.offsets[i + 1];
      // a spelling is at most 5 bytes; white space around it is unbounded, so the whole value is scanned
      v = parse_bool(p.data + b, e - b);
    }
    const u64 vm = __ballot(v >= 0), rm = __ballot(v == 1);
    if (lane == 0) {
      p.out_bits[base >> 6] = rm;
      p.out_validity[base >> 6] = vm;
      const int64_t rows = p.nrows - base < 64 ? p.nrows - base : 64;
      const unsigned nulls = (unsigned)rows - (unsigned)__popcll(vm);
      if (nulls) atomicAdd(p.null_count, (u64)nulls);
    }
  }
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void utf8_uniform_kernel(const Utf8UniformParams p) {
  const int32_t first = p.offsets[0], len = p.offsets[1] - first;
  bool differs = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.nrows; i += (int64_t)gridDim.x * 256)
    differs |= p.offsets[i + 1] - p.offsets[i] != len;   // (every offset is read twice, by neighbouring lanes: one line)
  if (__ballot(differs) != 0 && (threadIdx.x & 63) == 0) atomicOr((unsigned*)&p.out[0], 1u);
  if (blockIdx.x == 0 && threadIdx.x == 0) { p.out[1] = len; p.out[2] = first; }
}
// grid = (batches, slices): a batch is walked by gridDim.y workgroups (few, large batches must not be ONE workgroup's work);
// out is zero-initialised, the flag is OR-ed in
__global__ __launch_bounds__(256) void utf8_uniform_group_kernel(const Utf8UniformGroupParams p) {
  for (int64_t b = blockIdx.x; b < p.nb; b += gridDim.x) {
    const int32_t* offs = (const int32_t*)(uintptr_t)p.offsets_of[b];
    const int64_t rows = p.rows_of[b];
    const int32_t first = offs[0], len = rows > 0 ? offs[1] - first : 0;
    bool differs = false;
    for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.y * 256) differs |= offs[i + 1] - offs[i] != len;
    if (__ballot(differs) != 0 && (threadIdx.x & 63) == 0) atomicOr((unsigned*)&p.out[3 * b], 1u);
    if (blockIdx.y == 0 && threadIdx.x == 0) { p.out[3 * b + 1] = len; p.out[3 * b + 2] = first; }
  }
}
__global__ __launch_bounds__(256) void iota_offsets_kernel(const IotaOffsetsParams p) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n_plus_1; i += (int64_t)gridDim.x * 256) p.out[i] = (int32_t)(i * p.step);
}
}  // namespace

static int grid_for(int64_t nrows) {
  const int64_t blocks = (nrows + 255) / 256;
  return (int)(blocks < 1 ? 1 : blocks > 256 * 32 ? 256 * 32 : blocks);
}
hipError_t launch_cmp128(const Cmp128Params& p, hipStream_t stream) {
  hipLaunchKernelGGL(cmp128_kernel, dim3(grid_for(p.nrows)), dim3(256), 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_utf8_uniform(const Utf8UniformParams& p, hipStream_t stream) {
  hipLaunchKernelGGL(utf8_uniform_kernel, dim3(grid_for(p.nrows)), dim3(256), 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_utf8_uniform_group(const Utf8UniformGroupParams& p, int64_t max_rows, hipStream_t stream) {
  if (p.nb <= 0) return hipSuccess;
  const unsigned gx = (unsigned)(p.nb < 65535 * 16 ? p.nb : 65535 * 16);
  int64_t gy = (max_rows + 16383) / 16384;             // ~64 offsets per thread ...
  const int64_t cap = (1 << 15) / (int64_t)gx + 1;     // ... and no more than ~2^15 workgroups in all
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  hipLaunchKernelGGL(utf8_uniform_group_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_iota_offsets(const IotaOffsetsParams& p, hipStream_t stream) {
  hipLaunchKernelGGL(iota_offsets_kernel, dim3(grid_for(p.n_plus_1)), dim3(256), 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_utf8_to_bool(const Utf8ToBoolParams& p, hipStream_t stream) {
  hipLaunchKernelGGL(utf8_to_bool_kernel, dim3(grid_for(p.nrows)), dim3(256), 0, stream, p);
  return hipGetLastError();
}
}  // namespace chq
