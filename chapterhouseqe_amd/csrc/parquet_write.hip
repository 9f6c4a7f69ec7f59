// parquet_write.hip -- gfx950 kernels that turn Arrow columns in HBM into the PLAIN-encoded value streams of Parquet data
// pages (the step behind the projection: materialize_files_task.rs:128-141 in the reference,
// where the `parquet` crate encodes on the CPU).  The host (parquet_write.cpp) writes page headers and the footer and
// copies the streams into the file image.
//   non-null fixed-width columns need no kernel at all: the Arrow values buffer IS the PLAIN stream;
//   definition levels are the Arrow validity bitmap behind a one-run header (bit width 1, LSB first: the same bit order);
//   pw_counts_kernel / pw_scan_kernel   per-block totals of "bytes this row contributes" and their exclusive scan
//   pw_compact_fixed_kernel             values of the non-null rows, densely packed
//   pw_utf8_encode_kernel               [4-byte length][bytes] per non-null row
//   pw_bits_to_bytes_kernel             Boolean bitmap (any bit offset) -> one byte per row (then compacted and re-packed)
// Memory-bound byte work: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "parquet_device.h"

namespace chq {

namespace {
constexpr int PW_BLOCK_ROWS = 4096;

__device__ __forceinline__ bool row_valid(const PwParams& p, int64_t r) {
  if (!p.validity) return true;
  const int64_t b = p.bit_offset + r;
  return (p.validity[b >> 3] >> (b & 7)) & 1;
}
// bytes row r contributes to the value stream
__device__ __forceinline__ uint32_t row_bytes(const PwParams& p, int64_t r) {
  if (r >= p.n_rows || !row_valid(p, r)) return 0;
  if (p.offsets) return 4u + (uint32_t)(p.offsets[r + 1] - p.offsets[r]);
  return (uint32_t)p.width;
}
}  // namespace

__global__ __launch_bounds__(256) void pw_counts_kernel(const PwParams p) {
  __shared__ unsigned long long s_w[4];
  const int64_t r0 = (int64_t)blockIdx.x * PW_BLOCK_ROWS;
  unsigned long long s = 0;
  for (int i = threadIdx.x; i < PW_BLOCK_ROWS; i += 256) s += row_bytes(p, r0 + i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) p.block_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void pw_scan_kernel(const PwParams p) {
  __shared__ unsigned long long s_part[256];
  __shared__ unsigned long long s_carry;
  const int tid = threadIdx.x;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < p.n_blocks; base += 256) {
    const int64_t i = base + tid;
    const unsigned long long v = i < p.n_blocks ? p.block_sums[i] : 0ull;
    s_part[tid] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const unsigned long long t = tid >= o ? s_part[tid - o] : 0ull;
      __syncthreads();
      s_part[tid] += t;
      __syncthreads();
    }
    if (i < p.n_blocks) p.block_sums[i] = s_carry + s_part[tid] - v;
    __syncthreads();
    if (tid == 255) s_carry += s_part[255];
    __syncthreads();
  }
  if (tid == 0) *p.total_bytes = s_carry;
}

// One block per PW_BLOCK_ROWS rows; 256 rows per step, positions from a block-local scan on top of the block's base.
// Fixed-width: out[pos] = value; Utf8: out[pos] = length, bytes behind it.
template <int W>
__global__ __launch_bounds__(256) void pw_encode_kernel(const PwParams p) {
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_dst[256], s_len[256];   // Utf8: where each row of the step goes, how long it is (~0: nothing to write)
  __shared__ int32_t s_src[256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * PW_BLOCK_ROWS;
  unsigned long long run = p.block_sums[blockIdx.x];
  for (int step = 0; step < PW_BLOCK_ROWS; step += 256) {
    const int64_t r = r0 + step + tid;
    const uint32_t nb = row_bytes(p, r);
    uint32_t inc = nb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wv; ++w) before += s_wave[w];
    const uint32_t all = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (W > 0) {
      if (nb) {
        uint8_t* dst = p.out + run + before + inc - nb;
        const uint8_t* src = p.values + r * W;
#pragma unroll
        for (int b = 0; b < W; ++b) dst[b] = src[b];          // (the stream is only byte aligned)
      }
    } else {
      // [4-byte length][bytes]: positions by one thread per row, the bytes by eight lanes per row in 16-byte chunks
      s_dst[tid] = before + inc - nb;                          // relative to this step's first byte
      s_len[tid] = nb ? nb - 4 : ~0u;
      s_src[tid] = nb ? p.offsets[r] : 0;
      __syncthreads();
      uint8_t* base = p.out + run;
      const int sl = tid & 7;
      for (int row = tid >> 3; row < 256; row += 32) {
        const uint32_t len = s_len[row];
        if (len == ~0u) continue;
        uint8_t* dst = base + s_dst[row];
        const uint8_t* src = p.data + s_src[row];
        if (sl == 0) __builtin_memcpy(dst, &len, 4);
        uint32_t b = (uint32_t)sl * 16;
        for (; b + 16 <= len; b += 128) { uint4 w; __builtin_memcpy(&w, src + b, 16); __builtin_memcpy(dst + 4 + b, &w, 16); }
        for (uint32_t k = (len & ~15u) + sl; k < len; k += 8) dst[4 + k] = src[k];
      }
    }
    run += all;
    __syncthreads();
  }
}

// Boolean values (bitmap at bit offset `value_bit_offset`) -> one byte per row
__global__ __launch_bounds__(256) void pw_bits_to_bytes_kernel(const PwParams p) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < p.n_rows; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = p.value_bit_offset + r;
    p.out[r] = (p.values[b >> 3] >> (b & 7)) & 1;
  }
}
// one byte per value -> bit-packed, LSB first (PLAIN BOOLEAN); n_rows = number of values
__global__ __launch_bounds__(256) void pw_bytes_to_bits_kernel(const PwParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (p.n_rows + 63) >> 6;
  for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwords; w += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const int64_t r = (w << 6) + lane;
    const unsigned long long m = __ballot(r < p.n_rows && p.values[r] != 0);
    if (lane == 0) ((unsigned long long*)p.out)[w] = m;
  }
}
// validity bitmap at a bit offset -> the same bits from bit 0 (definition levels of a sliced column); n_rows bits
__global__ __launch_bounds__(256) void pw_shift_bits_kernel(const PwParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (p.n_rows + 63) >> 6;
  for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwords; w += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const int64_t r = (w << 6) + lane;
    const unsigned long long m = __ballot(r < p.n_rows && row_valid(p, r));
    if (lane == 0) ((unsigned long long*)p.out)[w] = m;
  }
}

// ---- chunk statistics: min / max ------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ bool stats_valid(const PwStatsParams& p, int64_t r) {
  if (!p.validity) return true;
  const int64_t b = p.bit_offset + r;
  return (p.validity[b >> 3] >> (b & 7)) & 1;
}
// unsigned byte-lexicographic a < b (Parquet's order for BYTE_ARRAY / UTF8)
__device__ bool utf8_less(const PwStatsParams& p, int64_t a, int64_t b) {
  const int32_t ao = p.offsets[a], bo = p.offsets[b];
  const int32_t al = p.offsets[a + 1] - ao, bl = p.offsets[b + 1] - bo;
  const int32_t n = al < bl ? al : bl;
  const uint8_t* pa = p.data + ao; const uint8_t* pb = p.data + bo;
  for (int32_t i = 0; i < n; ++i) { if (pa[i] != pb[i]) return pa[i] < pb[i]; }
  return al < bl;
}
}  // namespace

__global__ __launch_bounds__(256) void pw_stats_kernel(const PwStatsParams p) {
  __shared__ long long s_min[4], s_max[4], s_cnt[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  long long mn = INT64_MAX, mx = INT64_MIN, cnt = 0;
  if (p.kind == PW_STATS_UTF8) {
    // candidates are rows (-1: none); the second launch (n_cand > 0) reduces the first launch's per-block candidates
    long long rmin = -1, rmax = -1;
    auto take = [&](long long r) {
      if (r < 0) return;
      if (rmin < 0 || utf8_less(p, r, rmin)) rmin = r;
      if (rmax < 0 || utf8_less(p, rmax, r)) rmax = r;
    };
    if (p.n_cand > 0) { for (int i = threadIdx.x; i < 2 * p.n_cand; i += blockDim.x) take(p.cand[i]); }
    else for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < p.n_rows; r += stride) if (stats_valid(p, r)) { take(r); ++cnt; }
    for (int o = 32; o > 0; o >>= 1) {
      const long long omin = __shfl_xor(rmin, o, 64), omax = __shfl_xor(rmax, o, 64);
      if (omin >= 0 && (rmin < 0 || utf8_less(p, omin, rmin))) rmin = omin;
      if (omax >= 0 && (rmax < 0 || utf8_less(p, rmax, omax))) rmax = omax;
      cnt += __shfl_xor(cnt, o, 64);
    }
    if (lane == 0) { s_min[wv] = rmin; s_max[wv] = rmax; s_cnt[wv] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) {
        if (s_min[w] >= 0 && (rmin < 0 || utf8_less(p, s_min[w], rmin))) rmin = s_min[w];
        if (s_max[w] >= 0 && (rmax < 0 || utf8_less(p, rmax, s_max[w]))) rmax = s_max[w];
        cnt += s_cnt[w];
      }
      if (p.n_cand > 0) { p.out[0] = rmin; p.out[1] = rmax; }
      else { p.cand[2 * blockIdx.x] = rmin; p.cand[2 * blockIdx.x + 1] = rmax; if (cnt) atomicAdd((unsigned long long*)&p.out[2], (unsigned long long)cnt); }
    }
    return;
  }
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < p.n_rows; r += stride) {
    if (!stats_valid(p, r)) continue;
    long long k;
    switch (p.kind) {
      case PW_STATS_I32: k = ((const int32_t*)p.values)[r]; break;
      case PW_STATS_I64: k = ((const long long*)p.values)[r]; break;
      case PW_STATS_F32: { const float f = ((const float*)p.values)[r]; if (f != f) continue;   // NaNs take no part (parquet-rs, parquet-cpp)
                           const int32_t b = __float_as_int(f); k = b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1); } break;
      case PW_STATS_F64: { const double f = ((const double*)p.values)[r]; if (f != f) continue;
                           const long long b = __double_as_longlong(f); k = b ^ (long long)(((unsigned long long)(b >> 63)) >> 1); } break;
      default: k = p.values[r]; break;
    }
    mn = k < mn ? k : mn; mx = k > mx ? k : mx; ++cnt;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const long long a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
    mn = a < mn ? a : mn; mx = b > mx ? b : mx; cnt += __shfl_xor(cnt, o, 64);
  }
  if (lane == 0) { s_min[wv] = mn; s_max[wv] = mx; s_cnt[wv] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) { mn = s_min[w] < mn ? s_min[w] : mn; mx = s_max[w] > mx ? s_max[w] : mx; cnt += s_cnt[w]; }
    if (cnt) { atomicMin(&p.out[0], mn); atomicMax(&p.out[1], mx); atomicAdd((unsigned long long*)&p.out[2], (unsigned long long)cnt); }
  }
}
hipError_t pw_launch_stats(const PwStatsParams& p, int grid, hipStream_t s) {
  hipLaunchKernelGGL(pw_stats_kernel, dim3(p.n_cand > 0 ? 1 : grid), dim3(256), 0, s, p);
  return hipGetLastError();
}

// one block per (piece, segment): 16-byte copies between arbitrarily aligned addresses (page headers have odd lengths)
__global__ __launch_bounds__(256) void pw_assemble_kernel(const PwAssembleParams p) {
  const PwPiece pc = p.pieces[blockIdx.x];
  const unsigned long long b0 = (unsigned long long)blockIdx.y * p.seg;
  if (b0 >= pc.len) return;
  const unsigned long long b1 = b0 + p.seg < pc.len ? b0 + p.seg : pc.len;
  uint8_t* dst = p.image + pc.dst;
  for (unsigned long long b = b0 + (unsigned long long)threadIdx.x * 16; b + 16 <= b1; b += 256 * 16) {
    uint4 w; __builtin_memcpy(&w, pc.src + b, 16); __builtin_memcpy(dst + b, &w, 16);
  }
  const unsigned long long tail = b0 + ((b1 - b0) & ~15ull);
  for (unsigned long long b = tail + threadIdx.x; b < b1; b += 256) dst[b] = pc.src[b];
}
hipError_t pw_launch_assemble(const PwAssembleParams& p, int n_pieces, int n_segs, hipStream_t s) {
  if (n_pieces <= 0) return hipSuccess;
  hipLaunchKernelGGL(pw_assemble_kernel, dim3((unsigned)n_pieces, (unsigned)n_segs), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t pw_launch_scan(const PwParams& p, hipStream_t s) {
  hipLaunchKernelGGL(pw_counts_kernel, dim3((unsigned)p.n_blocks), dim3(256), 0, s, p);
  hipLaunchKernelGGL(pw_scan_kernel, dim3(1), dim3(256), 0, s, p);
  return hipGetLastError();
}
hipError_t pw_launch_encode(const PwParams& p, hipStream_t s) {
  const dim3 g((unsigned)p.n_blocks), b(256);
  if (p.offsets) hipLaunchKernelGGL(pw_encode_kernel<0>, g, b, 0, s, p);
  else switch (p.width) {
    case 1: hipLaunchKernelGGL(pw_encode_kernel<1>, g, b, 0, s, p); break;
    case 4: hipLaunchKernelGGL(pw_encode_kernel<4>, g, b, 0, s, p); break;
    case 8: hipLaunchKernelGGL(pw_encode_kernel<8>, g, b, 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t pw_launch_bits_to_bytes(const PwParams& p, int grid, hipStream_t s) { hipLaunchKernelGGL(pw_bits_to_bytes_kernel, dim3(grid), dim3(256), 0, s, p); return hipGetLastError(); }
hipError_t pw_launch_bytes_to_bits(const PwParams& p, int grid, hipStream_t s) { hipLaunchKernelGGL(pw_bytes_to_bits_kernel, dim3(grid), dim3(256), 0, s, p); return hipGetLastError(); }
hipError_t pw_launch_shift_bits(const PwParams& p, int grid, hipStream_t s) { hipLaunchKernelGGL(pw_shift_bits_kernel, dim3(grid), dim3(256), 0, s, p); return hipGetLastError(); }

}  // namespace chq
