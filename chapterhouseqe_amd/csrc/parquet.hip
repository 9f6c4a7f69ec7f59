// parquet.hip -- gfx950 kernels that decode Parquet pages into Arrow buffers in HBM (the
// step in front of the filter path: read_files_task.rs:233-282 in the reference, where the `parquet` crate decodes on the
// CPU).  The host (parquet_meta.cpp / parquet_scan.cpp) parses only metadata and uploads each column chunk as it lies in
// the file; everything per value happens here.
//
// Work split: pages are independent, so the inherently serial parts -- walking the run headers of the RLE / bit-packed
// hybrid encoding, walking the length prefixes of PLAIN BYTE_ARRAY values -- run ONE WAVE PER PAGE (the wave expands each
// run with all 64 lanes; the length walk is staged through LDS so that its dependent loads have LDS, not HBM, latency),
// and every per-row step (rank -> value, gathers, offset scan, byte copies) is a plain data-parallel kernel.
//   levels      pq_levels_kernel        definition levels (bit width 1)   -> one byte per row + non-null count per page
//   page scan   pq_page_scan_kernel     non-null counts                  -> first value index of every page
//   row values  pq_rowval_kernel        valid bytes                      -> value index per row (-1 = null)
//   values      pq_plain_copy_kernel    PLAIN fixed-width pages          -> dense values (page headers squeezed out)
//               pq_dict_fixed_kernel    RLE_DICTIONARY fixed-width pages -> dense values (dictionary applied)
//               pq_bool_kernel, pq_bool_rle_kernel   PLAIN / RLE BOOLEAN pages -> one byte per value
//               pq_ba_walk_kernel       PLAIN BYTE_ARRAY pages / dictionary page -> (position, length) per value
//               pq_dict_ba_kernel       RLE_DICTIONARY BYTE_ARRAY pages  -> (position, length) per value
//   rows        pq_gather_fixed_kernel, pq_pack_bits_kernel, pq_rowlen_* (offset scan), pq_utf8_copy_kernel
// Memory-bound integer/byte work: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "parquet_device.h"

namespace chq {

namespace {

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ unsigned lane_rank64(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}
__device__ __forceinline__ void flag_error(uint32_t* err, uint32_t code) { atomicMax(err, code); }

// The RLE / bit-packed hybrid of Parquet (definition levels, dictionary indices): a sequence of runs, each introduced by
// a varint header h -- (h & 1) == 0: RLE, h >> 1 repetitions of one value stored in ceil(bw / 8) bytes; (h & 1) == 1:
// bit-packed, (h >> 1) groups of 8 values, bw bits each, LSB first.  The whole wave walks the headers together (uniform
// scalar work), then expands the run with all lanes; `emit(k, v)` receives value v for position k < n.  Returns the
// number of values emitted (n unless the stream ends early).
template <typename Emit>
__device__ __forceinline__ uint32_t hybrid_decode(const uint8_t* p, uint32_t len, int bw, uint32_t n, int lane, Emit&& emit) {
  uint32_t pos = 0, k = 0;
  const int vbytes = (bw + 7) >> 3;
  const uint32_t mask = bw >= 32 ? 0xffffffffu : ((1u << bw) - 1u);
  while (k < n && pos < len) {
    uint32_t h = 0;
    for (int shift = 0; shift < 35 && pos < len; shift += 7) { const uint8_t b = p[pos++]; h |= (uint32_t)(b & 0x7f) << shift; if (!(b & 0x80)) break; }
    if ((h & 1u) == 0) {
      uint32_t cnt = h >> 1;
      uint32_t v = 0;
      for (int b = 0; b < vbytes && pos < len; ++b) v |= (uint32_t)p[pos++] << (8 * b);
      v &= mask;
      if (cnt > n - k) cnt = n - k;
      for (uint32_t i = lane; i < cnt; i += 64) emit(k + i, v);
      k += cnt;
    } else {
      const uint32_t groups = h >> 1;
      uint32_t cnt = groups * 8u;
      const uint32_t bytes = groups * (uint32_t)bw;
      uint32_t avail = cnt;
      if (pos + bytes > len) avail = bw ? ((len - pos) * 8u) / (uint32_t)bw : cnt;   // truncated stream: what is there
      if (cnt > n - k) cnt = n - k;
      if (cnt > avail) cnt = avail;
      for (uint32_t i = lane; i < cnt; i += 64) {
        const uint64_t bit = (uint64_t)i * (uint32_t)bw;
        const uint8_t* q = p + pos + (bit >> 3);
        uint64_t w = 0;   // up to 5 bytes are needed; the chunk buffer is padded, but never read past the run itself
        const uint32_t left = pos + bytes - (uint32_t)(pos + (bit >> 3));
        if (left >= 8) __builtin_memcpy(&w, q, 8);
        else for (uint32_t b = 0; b < left; ++b) w |= (uint64_t)q[b] << (8 * b);
        emit(k + i, (uint32_t)(w >> (bit & 7)) & mask);
      }
      k += cnt;
      pos += bytes;
    }
  }
  return k;
}

}  // namespace

// ---- definition levels: one byte per row, non-null count per page -----------------------------------------------------
__global__ __launch_bounds__(64) void pq_levels_kernel(const PqDecodeParams p) {
  const int page = blockIdx.x, lane = threadIdx.x;
  const PqPageDesc d = p.pages[page];
  uint8_t* valid = p.valid8 + d.first_row;
  uint32_t ones = 0;
  const uint32_t got = hybrid_decode(p.chunk + d.levels_at, d.levels_len, 1, d.num_rows, lane,
                                     [&](uint32_t k, uint32_t v) { valid[k] = (uint8_t)v; ones += v; });
  if (got != d.num_rows) flag_error(p.err, PQ_ERR_LEVELS);
  ones = wave_sum_u32(ones);
  if (lane == 0) p.nonnull[page] = ones;
}

// ---- first value index of every page (exclusive scan of the non-null counts; one block) -------------------------------
__global__ __launch_bounds__(256) void pq_page_scan_kernel(const PqDecodeParams p) {
  __shared__ uint32_t s_part[256];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < p.n_pages; base += 256) {
    const int i = base + tid;
    const uint32_t v = i < p.n_pages ? p.nonnull[i] : 0u;
    s_part[tid] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const uint32_t t = tid >= o ? s_part[tid - o] : 0u;
      __syncthreads();
      s_part[tid] += t;
      __syncthreads();
    }
    if (i < p.n_pages) p.value_base[i] = s_carry + s_part[tid] - v;
    __syncthreads();
    if (tid == 255) s_carry += s_part[255];
    __syncthreads();
  }
  if (tid == 0) *p.total_values = s_carry;
}

// ---- value index of every row (-1 = null): one wave per page, 64 rows per step -----------------------------------------
__global__ __launch_bounds__(64) void pq_rowval_kernel(const PqDecodeParams p) {
  const int page = blockIdx.x, lane = threadIdx.x;
  const PqPageDesc d = p.pages[page];
  uint32_t run = p.value_base[page];
  for (uint32_t r = 0; r < d.num_rows; r += 64) {
    const bool in = r + lane < d.num_rows;
    const bool v = in && p.valid8[d.first_row + r + lane] != 0;
    const unsigned long long m = __ballot(v);
    if (in) p.row_val[d.first_row + r + lane] = v ? (int32_t)(run + lane_rank64(m)) : -1;
    run += (uint32_t)__popcll(m);
  }
}

// ---- PLAIN fixed-width pages: squeeze the page headers out ------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_plain_copy_kernel(const PqDecodeParams p) {
  const PqPageDesc d = p.pages[p.page_list[blockIdx.x]];
  const uint32_t n = p.nonnull[p.page_list[blockIdx.x]];
  const uint64_t bytes = (uint64_t)n * p.width;
  if (bytes > d.values_len) { if (threadIdx.x == 0) flag_error(p.err, PQ_ERR_VALUES); return; }
  const uint8_t* src = p.chunk + d.values_at;
  uint8_t* dst = p.dense + (uint64_t)p.value_base[p.page_list[blockIdx.x]] * p.width;
  // (source and destination are only byte aligned: 16-byte chunks through memcpy, global memory takes unaligned accesses)
  for (uint64_t b = ((uint64_t)blockIdx.y * blockDim.x + threadIdx.x) * 16; b < bytes; b += (uint64_t)gridDim.y * blockDim.x * 16) {
    if (b + 16 <= bytes) { uint4 w; __builtin_memcpy(&w, src + b, 16); __builtin_memcpy(dst + b, &w, 16); }
    else for (uint64_t k = b; k < bytes; ++k) dst[k] = src[k];
  }
}

// ---- RLE_DICTIONARY pages of a fixed-width column: decode the indices, apply the dictionary ---------------------------
template <typename TY>
__global__ __launch_bounds__(64) void pq_dict_fixed_kernel(const PqDecodeParams p) {
  const int page = p.page_list[blockIdx.x], lane = threadIdx.x;
  const PqPageDesc d = p.pages[page];
  const uint32_t n = p.nonnull[page];
  if (n == 0) return;
  if (d.values_len < 1) { flag_error(p.err, PQ_ERR_VALUES); return; }
  const uint8_t* v = p.chunk + d.values_at;
  const int bw = v[0];
  if (bw > 32) { flag_error(p.err, PQ_ERR_VALUES); return; }
  TY* out = (TY*)p.dense + p.value_base[page];
  const uint8_t* dict = p.chunk + p.dict_at;
  bool bad_index = false;
  const uint32_t got = hybrid_decode(v + 1, d.values_len - 1, bw, n, lane, [&](uint32_t k, uint32_t idx) {
    if (idx >= p.dict_count) { bad_index = true; idx = 0; }
    TY w; __builtin_memcpy(&w, dict + (uint64_t)idx * sizeof(TY), sizeof(TY));
    out[k] = w;
  });
  if (got != n || __ballot(bad_index)) flag_error(p.err, PQ_ERR_VALUES);
}

// ---- PLAIN BOOLEAN pages: bit-packed, LSB first -> one byte per value --------------------------------------------------
__global__ __launch_bounds__(256) void pq_bool_kernel(const PqDecodeParams p) {
  const int page = p.page_list[blockIdx.x];
  const PqPageDesc d = p.pages[page];
  const uint32_t n = p.nonnull[page];
  if ((uint64_t)d.values_len * 8 < n) { if (threadIdx.x == 0) flag_error(p.err, PQ_ERR_VALUES); return; }
  const uint8_t* src = p.chunk + d.values_at;
  uint8_t* out = p.dense + p.value_base[page];
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) out[k] = (src[k >> 3] >> (k & 7)) & 1;
}

// ---- RLE BOOLEAN pages (what V2 writers use): [4-byte length][hybrid runs, bit width 1] -> one byte per value ----------
__global__ __launch_bounds__(64) void pq_bool_rle_kernel(const PqDecodeParams p) {
  const int page = p.page_list[blockIdx.x], lane = threadIdx.x;
  const PqPageDesc d = p.pages[page];
  const uint32_t n = p.nonnull[page];
  if (n == 0) return;
  if (d.values_len < 4) { flag_error(p.err, PQ_ERR_VALUES); return; }
  const uint8_t* v = p.chunk + d.values_at;
  uint32_t len = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 24;
  if (len > d.values_len - 4) { flag_error(p.err, PQ_ERR_VALUES); len = d.values_len - 4; }
  uint8_t* out = p.dense + p.value_base[page];
  const uint32_t got = hybrid_decode(v + 4, len, 1, n, lane, [&](uint32_t k, uint32_t b) { out[k] = (uint8_t)b; });
  if (got != n) flag_error(p.err, PQ_ERR_VALUES);
}

// ---- PLAIN BYTE_ARRAY values (a data page, or the dictionary page): [4-byte length][bytes] ... --------------------------
// The position of value k+1 depends on the length of value k: the walk is serial per page.  One wave per page: a window
// of the page is staged in LDS by all lanes, lane 0 walks the length prefixes inside it (dependent LDS reads instead of
// dependent HBM reads), the (position, length) pairs it finds are collected in LDS and written out by all lanes.
constexpr int PQ_WALK_WINDOW = 16384;   // bytes
constexpr int PQ_WALK_BATCH = 1024;     // values
__global__ __launch_bounds__(64) void pq_ba_walk_kernel(const PqDecodeParams p) {
  __shared__ uint32_t s_win[PQ_WALK_WINDOW / 4 + 2];
  __shared__ uint32_t s_src[PQ_WALK_BATCH], s_len[PQ_WALK_BATCH];
  __shared__ uint32_t s_state[3];   // values found in this round, next position, error
  const int lane = threadIdx.x;
  uint32_t values_at, values_len, n; uint32_t* osrc; uint32_t* olen;
  if (p.walk_dictionary) {
    values_at = p.dict_at; values_len = p.dict_len; n = p.dict_count; osrc = p.dict_src; olen = p.dict_len_out;
  } else {
    const int page = p.page_list[blockIdx.x];
    const PqPageDesc d = p.pages[page];
    values_at = d.values_at; values_len = d.values_len; n = p.nonnull[page];
    osrc = p.vsrc + p.value_base[page]; olen = p.vlen + p.value_base[page];
  }
  const uint32_t end = values_at + values_len;
  uint32_t pos = values_at, k = 0;
  while (k < n) {
    // stage [wbase, wbase + WINDOW): wbase is pos rounded down to a multiple of 4 (dword loads of the chunk buffer)
    const uint32_t wbase = pos & ~3u;
    const uint32_t* g = (const uint32_t*)(p.chunk + wbase);
    const uint32_t wbytes = end - wbase < (uint32_t)PQ_WALK_WINDOW ? end - wbase : (uint32_t)PQ_WALK_WINDOW;
    for (uint32_t i = lane; i < (wbytes + 3) / 4 + 1; i += 64) s_win[i] = g[i];   // (the chunk buffer is padded by 16 bytes)
    __syncthreads();
    if (lane == 0) {
      uint32_t found = 0, q = pos, e = 0;
      while (k + found < n && found < (uint32_t)PQ_WALK_BATCH) {
        if (q + 4 > end) { e = 1; break; }
        const uint32_t off = q - wbase;
        if (off + 4 > wbytes) break;                       // the next length prefix lies outside the window: restage
        const uint32_t lo = s_win[off >> 2], hi = s_win[(off >> 2) + 1];
        const uint32_t len = __builtin_amdgcn_alignbyte(hi, lo, off & 3);
        if (len > end - q - 4) { e = 1; break; }
        s_src[found] = q + 4; s_len[found] = len;
        ++found;
        q += 4 + len;
      }
      s_state[0] = found; s_state[1] = q; s_state[2] = e;
    }
    __syncthreads();
    const uint32_t found = s_state[0];
    for (uint32_t i = lane; i < found; i += 64) { osrc[k + i] = s_src[i]; olen[k + i] = s_len[i]; }
    k += found; pos = s_state[1];
    if (s_state[2]) {   // malformed page: the rest reads as empty strings
      if (lane == 0) flag_error(p.err, PQ_ERR_VALUES);
      for (uint32_t i = k + lane; i < n; i += 64) { osrc[i] = values_at; olen[i] = 0; }
      break;
    }
    __syncthreads();
  }
}

// ---- RLE_DICTIONARY pages of a BYTE_ARRAY column: index -> (position, length) of the dictionary entry ------------------
__global__ __launch_bounds__(64) void pq_dict_ba_kernel(const PqDecodeParams p) {
  const int page = p.page_list[blockIdx.x], lane = threadIdx.x;
  const PqPageDesc d = p.pages[page];
  const uint32_t n = p.nonnull[page];
  if (n == 0) return;
  if (d.values_len < 1) { flag_error(p.err, PQ_ERR_VALUES); return; }
  const uint8_t* v = p.chunk + d.values_at;
  const int bw = v[0];
  if (bw > 32) { flag_error(p.err, PQ_ERR_VALUES); return; }
  uint32_t* osrc = p.vsrc + p.value_base[page];
  uint32_t* olen = p.vlen + p.value_base[page];
  bool bad_index = false;
  const uint32_t got = hybrid_decode(v + 1, d.values_len - 1, bw, n, lane, [&](uint32_t k, uint32_t idx) {
    if (idx >= p.dict_count) { bad_index = true; osrc[k] = d.values_at; olen[k] = 0; return; }
    osrc[k] = p.dict_src[idx]; olen[k] = p.dict_len_out[idx];
  });
  if (got != n || __ballot(bad_index)) flag_error(p.err, PQ_ERR_VALUES);
}

// ---- rows ------------------------------------------------------------------------------------------------------------
// out[row] = dense[row_val[row]] (0 for a null)
template <typename TY>
__global__ __launch_bounds__(256) void pq_gather_fixed_kernel(const PqRowParams p) {
  const TY* src = (const TY*)p.dense; TY* out = (TY*)p.out;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < p.n_rows; r += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = p.row_val[r];
    TY w{};
    if (v >= 0) w = src[v];
    out[r] = w;
  }
}

// Arrow bitmap (LSB first) from one byte per row: src8[row], or src8[row_val[row]] with nulls reading as 0
__global__ __launch_bounds__(256) void pq_pack_bits_kernel(const PqRowParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (p.n_rows + 63) >> 6;
  for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwords; w += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const int64_t r = (w << 6) + lane;
    bool bit = false;
    if (r < p.n_rows) {
      if (p.row_val) { const int32_t v = p.row_val[r]; bit = v >= 0 && p.dense[v] != 0; }
      else bit = p.dense[r] != 0;
    }
    const unsigned long long m = __ballot(bit);
    if (lane == 0) ((unsigned long long*)p.out)[w] = m;
  }
}

// Utf8 offsets = exclusive scan of the row lengths (vlen[row_val[row]], 0 for a null), three small kernels:
// block sums (PQ_SCAN_ROWS rows per block) -> scan of the block sums (one block) -> offsets
constexpr int PQ_SCAN_ROWS = 4096;
__device__ __forceinline__ uint32_t row_length(const PqRowParams& p, int64_t r) {
  if (r >= p.n_rows) return 0;
  if (!p.row_val) return p.vlen[r];
  const int32_t v = p.row_val[r];
  return v >= 0 ? p.vlen[v] : 0u;
}
__global__ __launch_bounds__(256) void pq_rowlen_sums_kernel(const PqRowParams p) {
  __shared__ unsigned long long s_w[4];
  const int64_t r0 = (int64_t)blockIdx.x * PQ_SCAN_ROWS;
  unsigned long long s = 0;
  for (int i = threadIdx.x; i < PQ_SCAN_ROWS; i += 256) s += row_length(p, r0 + i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) p.block_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void pq_rowlen_scan_kernel(const PqRowParams p) {
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
__global__ __launch_bounds__(256) void pq_rowlen_offsets_kernel(const PqRowParams p) {
  __shared__ uint32_t s_wave[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * PQ_SCAN_ROWS;
  unsigned long long run = p.block_sums[blockIdx.x];
  int32_t* offs = (int32_t*)p.out;
  for (int step = 0; step < PQ_SCAN_ROWS; step += 256) {
    const int64_t r = r0 + step + tid;
    const uint32_t len = row_length(p, r);
    uint32_t inc = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wv; ++w) before += s_wave[w];
    const uint32_t all = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (r < p.n_rows) offs[r] = (int32_t)(run + before + inc - len);   // (the host checks the total against 2^31 before trusting these)
    run += all;
    __syncthreads();
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) offs[p.n_rows] = (int32_t)*p.total_bytes;
}

// bytes of every row: eight lanes per row, 16-byte chunks, then the tail byte by byte
__global__ __launch_bounds__(256) void pq_utf8_copy_kernel(const PqRowParams p) {
  const int sl = threadIdx.x & 7;
  const int32_t* offs = (const int32_t*)p.offsets;
  for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; r < p.n_rows; r += ((int64_t)gridDim.x * blockDim.x) >> 3) {
    int32_t v = p.row_val ? p.row_val[r] : (int32_t)r;
    if (v < 0) continue;
    const uint32_t len = p.vlen[v];
    const uint8_t* src = p.chunk + p.vsrc[v];
    uint8_t* dst = p.data_out + offs[r];
    uint32_t b = (uint32_t)sl * 16;
    for (; b + 16 <= len; b += 128) { uint4 w; __builtin_memcpy(&w, src + b, 16); __builtin_memcpy(dst + b, &w, 16); }
    const uint32_t tail = len & ~15u;
    for (uint32_t k = tail + sl; k < len; k += 8) dst[k] = src[k];
  }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
hipError_t pq_launch_levels(const PqDecodeParams& p, hipStream_t s) { hipLaunchKernelGGL(pq_levels_kernel, dim3(p.n_pages), dim3(64), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_page_scan(const PqDecodeParams& p, hipStream_t s) { hipLaunchKernelGGL(pq_page_scan_kernel, dim3(1), dim3(256), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_rowval(const PqDecodeParams& p, hipStream_t s) { hipLaunchKernelGGL(pq_rowval_kernel, dim3(p.n_pages), dim3(64), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_plain_copy(const PqDecodeParams& p, int n_list, hipStream_t s) { hipLaunchKernelGGL(pq_plain_copy_kernel, dim3(n_list, 16), dim3(256), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_dict_fixed(const PqDecodeParams& p, int n_list, hipStream_t s) {
  switch (p.width) {
    case 4: hipLaunchKernelGGL(pq_dict_fixed_kernel<uint32_t>, dim3(n_list), dim3(64), 0, s, p); break;
    case 8: hipLaunchKernelGGL(pq_dict_fixed_kernel<uint64_t>, dim3(n_list), dim3(64), 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t pq_launch_bool(const PqDecodeParams& p, int n_list, hipStream_t s) { hipLaunchKernelGGL(pq_bool_kernel, dim3(n_list), dim3(256), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_bool_rle(const PqDecodeParams& p, int n_list, hipStream_t s) { hipLaunchKernelGGL(pq_bool_rle_kernel, dim3(n_list), dim3(64), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_ba_walk(const PqDecodeParams& p, int n_list, hipStream_t s) { hipLaunchKernelGGL(pq_ba_walk_kernel, dim3(n_list), dim3(64), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_dict_ba(const PqDecodeParams& p, int n_list, hipStream_t s) { hipLaunchKernelGGL(pq_dict_ba_kernel, dim3(n_list), dim3(64), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_gather_fixed(const PqRowParams& p, int width, int grid, hipStream_t s) {
  switch (width) {
    case 4: hipLaunchKernelGGL(pq_gather_fixed_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, p); break;
    case 8: hipLaunchKernelGGL(pq_gather_fixed_kernel<uint64_t>, dim3(grid), dim3(256), 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t pq_launch_pack_bits(const PqRowParams& p, int grid, hipStream_t s) { hipLaunchKernelGGL(pq_pack_bits_kernel, dim3(grid), dim3(256), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_rowlen(const PqRowParams& p, hipStream_t s) {
  hipLaunchKernelGGL(pq_rowlen_sums_kernel, dim3((unsigned)p.n_blocks), dim3(256), 0, s, p);
  hipLaunchKernelGGL(pq_rowlen_scan_kernel, dim3(1), dim3(256), 0, s, p);
  hipLaunchKernelGGL(pq_rowlen_offsets_kernel, dim3((unsigned)p.n_blocks), dim3(256), 0, s, p);
  return hipGetLastError();
}
hipError_t pq_launch_utf8_copy(const PqRowParams& p, int grid, hipStream_t s) { hipLaunchKernelGGL(pq_utf8_copy_kernel, dim3(grid), dim3(256), 0, s, p); return hipGetLastError(); }

}  // namespace chq
