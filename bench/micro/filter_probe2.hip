// Design probe (not product code): single-pass order-preserving filter of 3 x f32
// columns by `value2 > 10.0`, several kernel structures, timed with HIP events.
// Build: hipcc -O3 --offload-arch=gfx950 -o filter_probe filter_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t mix32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}

__global__ void gen_kernel(float* __restrict__ p, size_t n, uint64_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t r = mix32(i * 0x9E3779B97F4A7C15ULL + seed);
    p[i] = (float)(r >> 8) * (100.0f / 16777216.0f);
  }
}

__global__ void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) out[i] = in[i];
}

__device__ __forceinline__ int tot_key(float f) {
  int b = __float_as_int(f);
  return b ^ (int)(((unsigned)(b >> 31)) >> 1);
}

// ---- tile status words: bits 63..62 flag (0 invalid, 1 aggregate, 2 inclusive), low 62 bits value
#define ST_AGG (1ULL << 62)
#define ST_INC (2ULL << 62)
#define ST_VAL(x) ((x) & ((1ULL << 62) - 1))
#define ST_FLAG(x) ((x) >> 62)

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

// called by wave 0 only; returns exclusive prefix of this tile (valid in all lanes)
__device__ __forceinline__ u64 lookback(u64* status, int tile, u64 count) {
  const int lane = threadIdx.x & 63;
  if (tile == 0) {
    if (lane == 0) st_store(&status[0], ST_INC | count);
    return 0;
  }
  if (lane == 0) st_store(&status[tile], ST_AGG | count);
  u64 excl = 0;
  int look = tile - 1;
  while (true) {
    int idx = look - lane;
    u64 w = ST_INC;  // virtual tile before 0: inclusive 0
    if (idx >= 0) {
      w = st_load(&status[idx]);
      while (ST_FLAG(w) == 0) { __builtin_amdgcn_s_sleep(1); w = st_load(&status[idx]); }
    }
    u64 incm = __ballot(ST_FLAG(w) == 2);
    if (incm) {
      int first = __builtin_ctzll(incm);
      u64 v = (lane <= first) ? ST_VAL(w) : 0;
      excl += wave_sum(v);
      break;
    }
    excl += wave_sum(ST_VAL(w));
    look -= 64;
  }
  if (lane == 0) st_store(&status[tile], ST_INC | (excl + count));
  return excl;
}


__global__ void copy_kernel_u4(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  size_t i = ((size_t)blockIdx.x * blockDim.x) * 4 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (; i + 3 * blockDim.x < n4; i += stride) {
    float4 a = in[i], b = in[i + blockDim.x], c = in[i + 2 * blockDim.x], d = in[i + 3 * blockDim.x];
    out[i] = a; out[i + blockDim.x] = b; out[i + 2 * blockDim.x] = c; out[i + 3 * blockDim.x] = d;
  }
}
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void copy_kernel_nt(const f4v* __restrict__ in, f4v* __restrict__ out, size_t n4) {
  size_t i = ((size_t)blockIdx.x * blockDim.x) * 4 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (; i + 3 * blockDim.x < n4; i += stride) {
    f4v a = __builtin_nontemporal_load(in + i), b = __builtin_nontemporal_load(in + i + blockDim.x),
           c = __builtin_nontemporal_load(in + i + 2 * blockDim.x), d = __builtin_nontemporal_load(in + i + 3 * blockDim.x);
    __builtin_nontemporal_store(a, out + i); __builtin_nontemporal_store(b, out + i + blockDim.x);
    __builtin_nontemporal_store(c, out + i + 2 * blockDim.x); __builtin_nontemporal_store(d, out + i + 3 * blockDim.x);
  }
}

// MODE 0: all columns held in registers; MODE 1: column-at-a-time (predicate phase, then copy phase)
// LB 1: look-back; LB 0: tile-local output base (upper bound, no inter-WG dependency)
template <int BLOCK, int R, int NC, int MODE, int LB>
__global__ __launch_bounds__(BLOCK) void filter_reg(const float* const* __restrict__ cols_, float* const* __restrict__ outs_,
                                                   size_t n, float thr, u64* status, unsigned* ticket, u64* total) {
  constexpr int NW = BLOCK / 64;
  __shared__ unsigned s_wave_cnt[NW];
  __shared__ u64 s_base;
  __shared__ int s_tile[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* cols[NC]; float* outs[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { cols[c] = cols_[c]; outs[c] = outs_[c]; }
  const int kthr = tot_key(thr);
  const size_t TILE = (size_t)BLOCK * R;
  const int ntiles = (int)((n + TILE - 1) / TILE);
  if (tid == 0) s_tile[0] = (int)atomicAdd(ticket, 1u);
  __syncthreads();
  int it = 0;
  while (true) {
    const int tile = s_tile[it & 1];
    if (tile >= ntiles) break;
    if (tid == 0) s_tile[(it + 1) & 1] = (int)atomicAdd(ticket, 1u);  // prefetch next ticket
    const size_t w0 = (size_t)tile * TILE + (size_t)wv * 64 * R;
    float v[NC][R];
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        size_t r = w0 + (size_t)j * 64 + lane;
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c][j] = (r < n) ? cols[c][r] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        size_t r = w0 + (size_t)j * 64 + lane;
        v[NC - 1][j] = (r < n) ? cols[NC - 1][r] : 0.f;
      }
    }
    u64 m[R]; unsigned cnt = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      size_t r = w0 + (size_t)j * 64 + lane;
      bool sel = (r < n) && (tot_key(v[NC - 1][j]) > kthr);
      m[j] = __ballot(sel);
      cnt += __popcll(m[j]);
    }
    if (lane == 0) s_wave_cnt[wv] = cnt;
    __syncthreads();
    if (wv == 0) {
      unsigned c = (lane < NW) ? s_wave_cnt[lane] : 0u;
      u64 tot = wave_sum((u64)c);
      u64 e;
      if (LB) e = lookback(status, tile, tot); else e = (u64)tile * TILE;
      if (lane == 0) { s_base = e; if (tile == ntiles - 1) *total = LB ? e + tot : (u64)(0.9 * n); }
    }
    if (MODE == 1) {
      // prefetch column 0 while wave 0 is in the look-back
#pragma unroll
      for (int j = 0; j < R; ++j) {
        size_t r = w0 + (size_t)j * 64 + lane;
        v[0][j] = (r < n) ? cols[0][r] : 0.f;
      }
    }
    __syncthreads();
    u64 off0 = s_base;
    for (int w = 0; w < wv; ++w) off0 += s_wave_cnt[w];
    if (MODE == 0) {
      u64 off = off0;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        bool sel = (m[j] >> lane) & 1;
        unsigned rk = __builtin_amdgcn_mbcnt_hi((unsigned)(m[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[j], 0));
        if (sel) {
#pragma unroll
          for (int c = 0; c < NC; ++c) outs[c][off + rk] = v[c][j];
        }
        off += __popcll(m[j]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (c > 0 && c < NC - 1) {
#pragma unroll
          for (int j = 0; j < R; ++j) {
            size_t r = w0 + (size_t)j * 64 + lane;
            v[c][j] = (r < n) ? cols[c][r] : 0.f;
          }
        }
        u64 off = off0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
          bool sel = (m[j] >> lane) & 1;
          unsigned rk = __builtin_amdgcn_mbcnt_hi((unsigned)(m[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[j], 0));
          if (sel) outs[c][off + rk] = v[c][j];
          off += __popcll(m[j]);
        }
      }
    }
    ++it;
    __syncthreads();
  }
}

// two-pass: count kernel + scatter kernel with precomputed tile offsets
template <int BLOCK, int R>
__global__ __launch_bounds__(BLOCK) void count_kernel(const float* __restrict__ pc, size_t n, float thr, unsigned* __restrict__ counts) {
  constexpr int NW = BLOCK / 64;
  __shared__ unsigned s_wave_cnt[NW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int kthr = tot_key(thr);
  const size_t TILE = (size_t)BLOCK * R;
  const size_t w0 = (size_t)blockIdx.x * TILE + (size_t)wv * 64 * R;
  unsigned cnt = 0;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    size_t r = w0 + (size_t)j * 64 + lane;
    float x = (r < n) ? pc[r] : 0.f;
    cnt += __popcll(__ballot((r < n) && tot_key(x) > kthr));
  }
  if (lane == 0) s_wave_cnt[wv] = cnt;
  __syncthreads();
  if (tid == 0) { unsigned t = 0; for (int w = 0; w < NW; ++w) t += s_wave_cnt[w]; counts[blockIdx.x] = t; }
}
// single-block scan of tile counts -> exclusive offsets (u64)
__global__ void scan_kernel(const unsigned* __restrict__ counts, u64* __restrict__ offs, int ntiles, u64* total) {
  __shared__ u64 s_part[1024];
  const int tid = threadIdx.x;
  int per = (ntiles + 1023) / 1024;
  int b = tid * per, e = min(ntiles, b + per);
  u64 s = 0;
  for (int i = b; i < e; ++i) s += counts[i];
  s_part[tid] = s;
  __syncthreads();
  if (tid == 0) { u64 a = 0; for (int i = 0; i < 1024; ++i) { u64 t = s_part[i]; s_part[i] = a; a += t; } *total = a; }
  __syncthreads();
  u64 a = s_part[tid];
  for (int i = b; i < e; ++i) { offs[i] = a; a += counts[i]; }
}
template <int BLOCK, int R, int NC>
__global__ __launch_bounds__(BLOCK) void scatter_kernel(const float* const* __restrict__ cols_, float* const* __restrict__ outs_,
                                                        size_t n, float thr, const u64* __restrict__ offs) {
  constexpr int NW = BLOCK / 64;
  __shared__ unsigned s_wave_cnt[NW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* cols[NC]; float* outs[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { cols[c] = cols_[c]; outs[c] = outs_[c]; }
  const int kthr = tot_key(thr);
  const size_t TILE = (size_t)BLOCK * R;
  const int tile = blockIdx.x;
  const size_t w0 = (size_t)tile * TILE + (size_t)wv * 64 * R;
  float v[NC][R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    size_t r = w0 + (size_t)j * 64 + lane;
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c][j] = (r < n) ? cols[c][r] : 0.f;
  }
  u64 m[R]; unsigned cnt = 0;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    size_t r = w0 + (size_t)j * 64 + lane;
    m[j] = __ballot((r < n) && (tot_key(v[NC - 1][j]) > kthr));
    cnt += __popcll(m[j]);
  }
  if (lane == 0) s_wave_cnt[wv] = cnt;
  __syncthreads();
  u64 off = offs[tile];
  for (int w = 0; w < wv; ++w) off += s_wave_cnt[w];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    bool sel = (m[j] >> lane) & 1;
    unsigned rk = __builtin_amdgcn_mbcnt_hi((unsigned)(m[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[j], 0));
    if (sel) {
#pragma unroll
      for (int c = 0; c < NC; ++c) outs[c][off + rk] = v[c][j];
    }
    off += __popcll(m[j]);
  }
}
struct Bufs {
  float* cols[3]; float* outs[3];
  const float** d_cols; float** d_outs;
  u64* status; unsigned* ticket; u64* total;
  size_t n; size_t status_bytes;
};

template <typename F>
static double time_it(const char* name, Bufs& b, F launch, int iters, double alg_bytes_hint) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int it = 0; it < iters + 2; ++it) {
    CK(hipMemsetAsync(b.status, 0, b.status_bytes, 0));
    CK(hipMemsetAsync(b.ticket, 0, 4, 0));
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (it >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  double med = ms[ms.size() / 2];
  u64 total = 0; CK(hipMemcpy(&total, b.total, 8, hipMemcpyDeviceToHost));
  double bytes = alg_bytes_hint > 0 ? alg_bytes_hint : (double)b.n * 12.0 + (double)total * 12.0;
  printf("%-28s n=%zu sel=%llu  median %.3f ms  min %.3f ms  alg %.1f GB/s (min: %.1f)  rows/s %.3e\n", name, b.n,
         (unsigned long long)total, med, ms[0], bytes / med / 1e6, bytes / ms[0] / 1e6, (double)b.n / med * 1e3);
  fflush(stdout);
  return med;
}

static bool verify(Bufs& b, size_t nchk) {
  // check prefix of output against CPU on first nchk input rows
  std::vector<float> h[3];
  for (int c = 0; c < 3; ++c) { h[c].resize(nchk); CK(hipMemcpy(h[c].data(), b.cols[c], nchk * 4, hipMemcpyDeviceToHost)); }
  std::vector<float> e[3];
  for (size_t i = 0; i < nchk; ++i)
    if (h[2][i] > 10.0f) for (int c = 0; c < 3; ++c) e[c].push_back(h[c][i]);
  size_t m = e[0].size();
  bool ok = true;
  for (int c = 0; c < 3; ++c) {
    std::vector<float> g(m); CK(hipMemcpy(g.data(), b.outs[c], m * 4, hipMemcpyDeviceToHost));
    if (memcmp(g.data(), e[c].data(), m * 4) != 0) { ok = false; printf("  MISMATCH col %d\n", c); }
  }
  return ok;
}

int main(int argc, char** argv) {
  size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000000ULL;
  int iters = argc > 2 ? atoi(argv[2]) : 10;
  n &= ~(size_t)3;
  Bufs b; b.n = n;
  for (int c = 0; c < 3; ++c) {
    CK(hipMalloc(&b.cols[c], n * 4 + 64)); CK(hipMalloc(&b.outs[c], n * 4 + 64));
    gen_kernel<<<4096, 256>>>(b.cols[c], n, 0xC0FFEEULL + c * 7919);
  }
  CK(hipDeviceSynchronize());
  CK(hipMalloc(&b.d_cols, 3 * sizeof(void*))); CK(hipMalloc(&b.d_outs, 3 * sizeof(void*)));
  CK(hipMemcpy(b.d_cols, b.cols, 3 * sizeof(void*), hipMemcpyHostToDevice));
  CK(hipMemcpy(b.d_outs, b.outs, 3 * sizeof(void*), hipMemcpyHostToDevice));
  b.status_bytes = ((n + 1023) / 1024 + 64) * 8;
  CK(hipMalloc(&b.status, b.status_bytes)); CK(hipMalloc(&b.ticket, 4)); CK(hipMalloc(&b.total, 8));
  CK(hipMemset(b.total, 0, 8));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs=%d\n", prop.name, cus);


  unsigned* counts; u64* offs;
  CK(hipMalloc(&counts, ((n + 1023) / 1024 + 64) * 4)); CK(hipMalloc(&offs, ((n + 1023) / 1024 + 64) * 8));
  for (int g : {2048, 8192}) {
    char nm[64]; snprintf(nm, 64, "copy_f4 grid=%d", g);
    time_it(nm, b, [&] { for (int c = 0; c < 3; ++c) copy_kernel<<<g, 256>>>((const float4*)b.cols[c], (float4*)b.outs[c], n / 4); }, iters, (double)n * 24.0);
    snprintf(nm, 64, "copy_u4 grid=%d", g);
    time_it(nm, b, [&] { for (int c = 0; c < 3; ++c) copy_kernel_u4<<<g, 256>>>((const float4*)b.cols[c], (float4*)b.outs[c], n / 4); }, iters, (double)n * 24.0);
    snprintf(nm, 64, "copy_nt grid=%d", g);
    time_it(nm, b, [&] { for (int c = 0; c < 3; ++c) copy_kernel_nt<<<g, 256>>>((const f4v*)b.cols[c], (f4v*)b.outs[c], n / 4); }, iters, (double)n * 24.0);
  }
#define RUN_REG(BLOCK, R, WPC, MODE, LB) { char nm[96]; snprintf(nm, 96, "reg B=%d R=%d wg/cu=%d mode=%d lb=%d", BLOCK, R, WPC, MODE, LB); \
    time_it(nm, b, [&] { filter_reg<BLOCK, R, 3, MODE, LB><<<cus * WPC, BLOCK>>>(b.d_cols, b.d_outs, n, 10.0f, b.status, b.ticket, b.total); }, iters, 0); \
    if (LB) printf("   verify: %s\n", verify(b, std::min<size_t>(n, 8u << 20)) ? "ok" : "FAIL"); }
  RUN_REG(256, 16, 6, 0, 1); RUN_REG(256, 16, 6, 0, 0);
  RUN_REG(512, 16, 3, 0, 1); RUN_REG(512, 16, 3, 0, 0);
  RUN_REG(1024, 16, 1, 0, 1); RUN_REG(1024, 16, 2, 0, 1); RUN_REG(1024, 16, 2, 0, 0);
  RUN_REG(1024, 8, 2, 0, 1); RUN_REG(1024, 8, 2, 0, 0);
  RUN_REG(256, 16, 6, 1, 1); RUN_REG(512, 16, 3, 1, 1); RUN_REG(1024, 16, 2, 1, 1); RUN_REG(1024, 16, 2, 1, 0);
  RUN_REG(512, 32, 2, 1, 1); RUN_REG(1024, 32, 1, 1, 1); RUN_REG(256, 32, 4, 1, 1);
#define RUN_2P(BLOCK, R) { char nm[96]; snprintf(nm, 96, "twopass B=%d R=%d", BLOCK, R); \
    time_it(nm, b, [&] { int nt = (int)((n + (size_t)BLOCK * R - 1) / ((size_t)BLOCK * R)); \
      count_kernel<BLOCK, R><<<nt, BLOCK>>>(b.cols[2], n, 10.0f, counts); scan_kernel<<<1, 1024>>>(counts, offs, nt, b.total); \
      scatter_kernel<BLOCK, R, 3><<<nt, BLOCK>>>(b.d_cols, b.d_outs, n, 10.0f, offs); }, iters, 0); \
    printf("   verify: %s\n", verify(b, std::min<size_t>(n, 8u << 20)) ? "ok" : "FAIL"); }
  RUN_2P(256, 16); RUN_2P(512, 16); RUN_2P(256, 8);
  return 0;
}
