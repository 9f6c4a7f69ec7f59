// parquet.hip -- gfx950 kernels that decode Parquet pages into Arrow buffers in HBM (the
// step in front of the filter path: read_files_task.rs:233-282 in the reference, where the `parquet` crate decodes on the
// CPU).  The host (parquet_meta.cpp / parquet_scan.cpp) parses only metadata and uploads each column chunk as it lies in
// the file; everything per value happens here.
//
// Work split: pages are independent, so the inherently serial parts run one workgroup (run headers of the RLE / bit-packed
// hybrid: 256 threads expand each run) or one wave (length prefixes of PLAIN BYTE_ARRAY values, 64 speculative positions
// per step) PER PAGE, staged through LDS so that the dependent loads have LDS, not HBM, latency; every per-row step
// (rank -> value, gathers, offset scan, byte copies) is a plain data-parallel kernel.
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

// The RLE / bit-packed hybrid of Parquet (definition levels, dictionary indices, RLE booleans): a sequence of runs, each
// introduced by a varint header h -- (h & 1) == 0: RLE, h >> 1 repetitions of one value stored in ceil(bw / 8) bytes;
// (h & 1) == 1: bit-packed, (h >> 1) groups of 8 values, bw bits each, LSB first.  `emit(k, v)` receives value v for
// position k < n; returns the number of values emitted (n unless the stream ends early).
// One workgroup per page, the stream staged through LDS: the run headers are parsed by every thread
// from LDS (identical scalar work, no dependent HBM loads), each run is expanded by all BLOCK threads, bit-packed values
// are extracted from LDS.  A bit-packed run that does not fit the window continues after the next staging.
constexpr int PQ_HYB_WINDOW = 16384;   // bytes
template <int BLOCK, typename Emit>
__device__ __forceinline__ uint32_t hybrid_decode_block(const uint8_t* g, uint32_t len, int bw, uint32_t n, uint32_t* s_win, Emit&& emit) {
  const int tid = threadIdx.x;
  const uint8_t* sb = (const uint8_t*)s_win;
  const int vbytes = (bw + 7) >> 3;
  const uint32_t mask = bw >= 32 ? 0xffffffffu : ((1u << bw) - 1u);
  uint32_t pos = 0, k = 0, pending = 0;   // pending: groups of a bit-packed run still to come
  while (k < n && pos < len) {
    const uint8_t* a = (const uint8_t*)((uintptr_t)(g + pos) & ~(uintptr_t)3);
    const uint32_t shift = (uint32_t)((g + pos) - a);
    const uint32_t wpos = pos;
    const uint32_t wbytes = len - pos < (uint32_t)(PQ_HYB_WINDOW - 16) ? len - pos : (uint32_t)(PQ_HYB_WINDOW - 16);
    for (uint32_t i = tid; i < (shift + wbytes + 3) / 4 + 2; i += BLOCK) s_win[i] = ((const uint32_t*)a)[i];   // (chunk buffers are padded by 64 bytes)
    __syncthreads();
    const uint32_t wend = pos + wbytes;
    bool stop = false;
    while (k < n && pos < wend && !stop) {
      uint32_t groups;
      if (pending) { groups = pending; pending = 0; }
      else {
        if (pos + 9 > wend && wend < len) break;            // header (<= 5 bytes) + RLE value (<= 4) may straddle the window
        uint32_t h = 0;
        for (int sh = 0; sh < 35 && pos < wend; sh += 7) { const uint8_t b = sb[pos - wpos + shift]; ++pos; h |= (uint32_t)(b & 0x7f) << sh; if (!(b & 0x80)) break; }
        if ((h & 1u) == 0) {
          uint32_t cnt = h >> 1, v = 0;
          for (int b = 0; b < vbytes && pos < wend; ++b) { v |= (uint32_t)sb[pos - wpos + shift] << (8 * b); ++pos; }
          v &= mask;
          if (cnt > n - k) cnt = n - k;
          for (uint32_t i = tid; i < cnt; i += BLOCK) emit(k + i, v);
          k += cnt;
          continue;
        }
        groups = h >> 1;
      }
      const uint32_t fit = bw ? (wend - pos) / (uint32_t)bw : groups;   // whole groups (bw bytes each) inside the window
      const uint32_t now = groups < fit ? groups : fit;
      if (now == 0) { if (wend >= len) stop = true; else pending = groups; break; }
      uint32_t cnt = now * 8u;
      if (cnt > n - k) cnt = n - k;
      for (uint32_t i = tid; i < cnt; i += BLOCK) {
        const uint64_t bit = (uint64_t)i * (uint32_t)bw;
        const uint32_t off = pos - wpos + shift + (uint32_t)(bit >> 3);
        const uint32_t d0 = s_win[off >> 2], d1 = s_win[(off >> 2) + 1], d2 = s_win[(off >> 2) + 2];
        const uint64_t w = (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, off & 3) | (uint64_t)__builtin_amdgcn_alignbyte(d2, d1, off & 3) << 32;
        emit(k + i, (uint32_t)(w >> (bit & 7)) & mask);
      }
      k += cnt;
      pos += now * (uint32_t)bw;
      if (now < groups) { pending = groups - now; break; }
    }
    __syncthreads();
    if (stop) break;
  }
  return k;
}
constexpr int PQ_HYB_BLOCK = 256;

}  // namespace

// ---- definition levels: one byte per row, non-null count per page -----------------------------------------------------
__global__ __launch_bounds__(PQ_HYB_BLOCK) void pq_levels_kernel(const PqDecodeParams p) {
  __shared__ uint32_t s_win[PQ_HYB_WINDOW / 4 + 4];
  __shared__ uint32_t s_ones;
  const int page = blockIdx.x;
  const PqPageDesc d = p.pages[page];
  uint8_t* valid = p.valid8 + d.first_row;
  if (threadIdx.x == 0) s_ones = 0;
  uint32_t ones = 0;
  const uint32_t got = hybrid_decode_block<PQ_HYB_BLOCK>(p.chunk + d.levels_at, d.levels_len, 1, d.num_rows, s_win,
                                                         [&](uint32_t k, uint32_t v) { valid[k] = (uint8_t)v; ones += v; });
  if (got != d.num_rows && threadIdx.x == 0) flag_error(p.err, PQ_ERR_LEVELS);
  ones = wave_sum_u32(ones);
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_ones, ones);
  __syncthreads();
  if (threadIdx.x == 0) p.nonnull[page] = s_ones;
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
__global__ __launch_bounds__(PQ_HYB_BLOCK) void pq_dict_fixed_kernel(const PqDecodeParams p) {
  __shared__ uint32_t s_win[PQ_HYB_WINDOW / 4 + 4];
  const int page = p.page_list[blockIdx.x];
  const PqPageDesc d = p.pages[page];
  const uint32_t n = p.nonnull[page];
  if (n == 0) return;
  if (d.values_len < 1) { if (threadIdx.x == 0) flag_error(p.err, PQ_ERR_VALUES); return; }
  const uint8_t* v = p.chunk + d.values_at;
  const int bw = v[0];
  if (bw > 32) { if (threadIdx.x == 0) flag_error(p.err, PQ_ERR_VALUES); return; }
  TY* out = (TY*)p.dense + p.value_base[page];
  const uint8_t* dict = p.chunk + p.dict_at;
  bool bad_index = false;
  const uint32_t got = hybrid_decode_block<PQ_HYB_BLOCK>(v + 1, d.values_len - 1, bw, n, s_win, [&](uint32_t k, uint32_t idx) {
    if (idx >= p.dict_count) { bad_index = true; idx = 0; }
    TY w; __builtin_memcpy(&w, dict + (uint64_t)idx * sizeof(TY), sizeof(TY));
    out[k] = w;
  });
  if (got != n || bad_index) flag_error(p.err, PQ_ERR_VALUES);
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
__global__ __launch_bounds__(PQ_HYB_BLOCK) void pq_bool_rle_kernel(const PqDecodeParams p) {
  __shared__ uint32_t s_win[PQ_HYB_WINDOW / 4 + 4];
  const int page = p.page_list[blockIdx.x];
  const PqPageDesc d = p.pages[page];
  const uint32_t n = p.nonnull[page];
  if (n == 0) return;
  if (d.values_len < 4) { if (threadIdx.x == 0) flag_error(p.err, PQ_ERR_VALUES); return; }
  const uint8_t* v = p.chunk + d.values_at;
  uint32_t len = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 24;
  if (len > d.values_len - 4) { if (threadIdx.x == 0) flag_error(p.err, PQ_ERR_VALUES); len = d.values_len - 4; }
  uint8_t* out = p.dense + p.value_base[page];
  const uint32_t got = hybrid_decode_block<PQ_HYB_BLOCK>(v + 4, len, 1, n, s_win, [&](uint32_t k, uint32_t b) { out[k] = (uint8_t)b; });
  if (got != n && threadIdx.x == 0) flag_error(p.err, PQ_ERR_VALUES);
}

// ---- PLAIN BYTE_ARRAY values (a data page, or the dictionary page): [4-byte length][bytes] ... --------------------------
// The position of value k+1 depends on the length of value k: the walk is serial per page.  One wave per page: a window
// of the page is staged in LDS by all lanes (dependent LDS reads instead of dependent HBM reads), and every step is
// SPECULATIVE: with L the length found at the current position, lane l looks at position + l * (4 + L) and checks that
// the length prefix there is L too; the lanes up to the first disagreement are confirmed together (each one's position
// follows from its predecessor's length).  Columns of equal-length strings (keys, codes, hashes, the reference's sample
// data) advance 256 values per step (four positions per lane); ragged ones fall back to bursts of plain serial steps.
constexpr int PQ_WALK_WINDOW = 32768;   // bytes
constexpr int PQ_WALK_SPEC = 4;         // speculative positions per lane and step
constexpr int PQ_WALK_BLOCK = 256;      // all four waves stage the window (8 x 16 bytes per thread, in flight together); wave 0 walks
__global__ __launch_bounds__(PQ_WALK_BLOCK) void pq_ba_walk_kernel(const PqDecodeParams p) {
  __shared__ __attribute__((aligned(16))) uint32_t s_win[PQ_WALK_WINDOW / 4 + 8];
  __shared__ uint32_t s_state[3];   // next position, values found so far, malformed
  const int tid = threadIdx.x, lane = tid & 63;
  const bool walker = tid < 64;
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
  uint32_t q = values_at, k = 0;
  bool failed = false;
  while (k < n && !failed) {
    // stage [wbase, wbase + wbytes): wbase is q rounded down to a multiple of 16 (dwordx4 loads of the chunk buffer)
    const uint32_t wbase = q & ~15u;
    const uint4* g = (const uint4*)(p.chunk + wbase);
    const uint32_t wbytes = end - wbase < (uint32_t)PQ_WALK_WINDOW ? end - wbase : (uint32_t)PQ_WALK_WINDOW;
    const uint32_t nvec = (wbytes + 15) / 16 + 1;   // (the chunk buffer is padded by 64 bytes)
    uint4 stage[PQ_WALK_WINDOW / 16 / PQ_WALK_BLOCK + 1];
#pragma unroll
    for (int u = 0; u < PQ_WALK_WINDOW / 16 / PQ_WALK_BLOCK + 1; ++u) { const uint32_t i = u * PQ_WALK_BLOCK + tid; if (i < nvec) stage[u] = g[i]; }
#pragma unroll
    for (int u = 0; u < PQ_WALK_WINDOW / 16 / PQ_WALK_BLOCK + 1; ++u) { const uint32_t i = u * PQ_WALK_BLOCK + tid; if (i < nvec) ((uint4*)s_win)[i] = stage[u]; }
    __syncthreads();
    if (walker) {
      auto length_at = [&](uint32_t pos) {   // pos + 4 <= wbase + wbytes
        const uint32_t off = pos - wbase;
        return (uint32_t)__builtin_amdgcn_alignbyte(s_win[(off >> 2) + 1], s_win[off >> 2], off & 3);
      };
      while (k < n) {
        if (q + 4 > end) { failed = true; break; }
        if (q + 4 > wbase + wbytes) break;                       // the next length prefix lies outside the window: restage
        const uint32_t L = length_at(q);                          // (uniform)
        if (L > end - q - 4) { failed = true; break; }
        // four positions per lane (256 per step): all four LDS reads are issued before the first ballot
        unsigned long long m[PQ_WALK_SPEC];
        uint32_t cpos[PQ_WALK_SPEC];
#pragma unroll
        for (int j = 0; j < PQ_WALK_SPEC; ++j) {
          const uint64_t c = (uint64_t)q + (uint64_t)(j * 64 + lane) * (4ull + L);
          bool ok = (uint64_t)k + j * 64 + lane < n && c + 4 <= (uint64_t)wbase + wbytes && c + 4 + L <= end;
          if (ok) ok = length_at((uint32_t)c) == L;
          m[j] = __ballot(ok);
          cpos[j] = (uint32_t)c + 4;
        }
        uint32_t total = 0;   // lane 0 of the first batch always agrees with itself: total >= 1
#pragma unroll
        for (int j = 0; j < PQ_WALK_SPEC; ++j) {
          const int cnt = m[j] == ~0ull ? 64 : __builtin_ctzll(~m[j]);
          if (lane < cnt) { osrc[k + total + lane] = cpos[j]; olen[k + total + lane] = L; }
          total += (uint32_t)cnt;
          if (cnt < 64) break;
        }
        k += total;
        q += total * (4u + L);
        if (total == 1) {   // ragged strings: the speculation bought nothing -- a burst of plain serial steps (uniform scalar
          // work, broadcast LDS reads, lane 0 stores) before it is tried again: 3 x the rate of one speculative step per value
          for (int burst = 0; burst < 48 && k < n; ++burst) {
            if (q + 4 > end) { failed = true; break; }
            if (q + 4 > wbase + wbytes) break;
            const uint32_t L2 = length_at(q);
            if (L2 > end - q - 4) { failed = true; break; }
            if (lane == 0) { osrc[k] = q + 4; olen[k] = L2; }
            ++k; q += 4u + L2;
          }
          if (failed) break;
        }
      }
      if (lane == 0) { s_state[0] = q; s_state[1] = k; s_state[2] = failed; }
    }
    __syncthreads();
    q = s_state[0]; k = s_state[1]; failed = s_state[2] != 0;
    __syncthreads();
  }
  if (failed && walker) {   // malformed page: the rest reads as empty strings
    if (lane == 0) flag_error(p.err, PQ_ERR_VALUES);
    for (uint32_t i = k + lane; i < n; i += 64) { osrc[i] = values_at; olen[i] = 0; }
  }
}

// ---- RLE_DICTIONARY pages of a BYTE_ARRAY column: index -> (position, length) of the dictionary entry ------------------
__global__ __launch_bounds__(PQ_HYB_BLOCK) void pq_dict_ba_kernel(const PqDecodeParams p) {
  __shared__ uint32_t s_win[PQ_HYB_WINDOW / 4 + 4];
  const int page = p.page_list[blockIdx.x];
  const PqPageDesc d = p.pages[page];
  const uint32_t n = p.nonnull[page];
  if (n == 0) return;
  if (d.values_len < 1) { if (threadIdx.x == 0) flag_error(p.err, PQ_ERR_VALUES); return; }
  const uint8_t* v = p.chunk + d.values_at;
  const int bw = v[0];
  if (bw > 32) { if (threadIdx.x == 0) flag_error(p.err, PQ_ERR_VALUES); return; }
  uint32_t* osrc = p.vsrc + p.value_base[page];
  uint32_t* olen = p.vlen + p.value_base[page];
  bool bad_index = false;
  const uint32_t got = hybrid_decode_block<PQ_HYB_BLOCK>(v + 1, d.values_len - 1, bw, n, s_win, [&](uint32_t k, uint32_t idx) {
    if (idx >= p.dict_count) { bad_index = true; osrc[k] = d.values_at; olen[k] = 0; return; }
    osrc[k] = p.dict_src[idx]; olen[k] = p.dict_len_out[idx];
  });
  if (got != n || bad_index) flag_error(p.err, PQ_ERR_VALUES);
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
hipError_t pq_launch_levels(const PqDecodeParams& p, hipStream_t s) { hipLaunchKernelGGL(pq_levels_kernel, dim3(p.n_pages), dim3(PQ_HYB_BLOCK), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_page_scan(const PqDecodeParams& p, hipStream_t s) { hipLaunchKernelGGL(pq_page_scan_kernel, dim3(1), dim3(256), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_rowval(const PqDecodeParams& p, hipStream_t s) { hipLaunchKernelGGL(pq_rowval_kernel, dim3(p.n_pages), dim3(64), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_plain_copy(const PqDecodeParams& p, int n_list, hipStream_t s) { hipLaunchKernelGGL(pq_plain_copy_kernel, dim3(n_list, 16), dim3(256), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_dict_fixed(const PqDecodeParams& p, int n_list, hipStream_t s) {
  switch (p.width) {
    case 4: hipLaunchKernelGGL(pq_dict_fixed_kernel<uint32_t>, dim3(n_list), dim3(PQ_HYB_BLOCK), 0, s, p); break;
    case 8: hipLaunchKernelGGL(pq_dict_fixed_kernel<uint64_t>, dim3(n_list), dim3(PQ_HYB_BLOCK), 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t pq_launch_bool(const PqDecodeParams& p, int n_list, hipStream_t s) { hipLaunchKernelGGL(pq_bool_kernel, dim3(n_list), dim3(256), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_bool_rle(const PqDecodeParams& p, int n_list, hipStream_t s) { hipLaunchKernelGGL(pq_bool_rle_kernel, dim3(n_list), dim3(PQ_HYB_BLOCK), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_ba_walk(const PqDecodeParams& p, int n_list, hipStream_t s) { hipLaunchKernelGGL(pq_ba_walk_kernel, dim3(n_list), dim3(PQ_WALK_BLOCK), 0, s, p); return hipGetLastError(); }
hipError_t pq_launch_dict_ba(const PqDecodeParams& p, int n_list, hipStream_t s) { hipLaunchKernelGGL(pq_dict_ba_kernel, dim3(n_list), dim3(PQ_HYB_BLOCK), 0, s, p); return hipGetLastError(); }
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
