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

// ---------------- V_reg: registers only, dword loads strided by 64, ballot + contiguous dword stores
template <int R, int NC>
__global__ __launch_bounds__(256) void filter_reg(const float* const* __restrict__ cols_, float* const* __restrict__ outs_,
                                                   size_t n, float thr, u64* status, unsigned* ticket, u64* total) {
  __shared__ u64 s_wave_cnt[4];
  __shared__ u64 s_base;
  __shared__ int s_tile;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* cols[NC]; float* outs[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { cols[c] = cols_[c]; outs[c] = outs_[c]; }
  const int kthr = tot_key(thr);
  const size_t TILE = (size_t)256 * R;
  const int ntiles = (int)((n + TILE - 1) / TILE);
  while (true) {
    if (tid == 0) s_tile = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = s_tile;
    if (tile >= ntiles) break;
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
      bool sel = (r < n) && (tot_key(v[NC - 1][j]) > kthr);
      m[j] = __ballot(sel);
      cnt += __popcll(m[j]);
    }
    if (lane == 0) s_wave_cnt[wv] = cnt;
    __syncthreads();
    if (wv == 0) {
      u64 c = s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
      u64 e = lookback(status, tile, c);
      if (lane == 0) { s_base = e; if (tile == ntiles - 1) *total = e + c; }
    }
    __syncthreads();
    u64 off = s_base;
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
    __syncthreads();
  }
}

// ---------------- V_lds: stage columns in LDS (dwordx4 loads), eval from LDS, selection vector, gather from LDS
template <int R, int NC, bool X4STORE>
__global__ __launch_bounds__(256) void filter_lds(const float* const* __restrict__ cols_, float* const* __restrict__ outs_,
                                                   size_t n, float thr, u64* status, unsigned* ticket, u64* total) {
  constexpr int TILE = 256 * R;
  __shared__ __attribute__((aligned(16))) float s_col[NC][TILE];
  __shared__ unsigned short s_sel[TILE + 8];
  __shared__ u64 s_ballot[R * 4];
  __shared__ unsigned s_pref[R * 4 + 1];
  __shared__ u64 s_base;
  __shared__ int s_tile;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* cols[NC]; float* outs[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { cols[c] = cols_[c]; outs[c] = outs_[c]; }
  const int kthr = tot_key(thr);
  const int ntiles = (int)((n + TILE - 1) / TILE);
  while (true) {
    if (tid == 0) s_tile = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = s_tile;
    if (tile >= ntiles) break;
    const size_t t0 = (size_t)tile * TILE;
    const int rows = (int)((n - t0 < (size_t)TILE) ? (n - t0) : (size_t)TILE);
    // stage (n assumed multiple of 4 and 16B aligned in this probe)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int j = 0; j < R / 4; ++j) {
        int q = tid + j * 256;  // float4 index in tile
        if (q * 4 < rows) {
          float4 x = *reinterpret_cast<const float4*>(cols[c] + t0 + (size_t)q * 4);
          *reinterpret_cast<float4*>(&s_col[c][q * 4]) = x;
        }
      }
    }
    __syncthreads();
    // eval
    bool sel[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      int r = tid + j * 256;
      sel[j] = (r < rows) && (tot_key(s_col[NC - 1][r]) > kthr);
      u64 b = __ballot(sel[j]);
      if (lane == 0) s_ballot[j * 4 + wv] = b;
    }
    __syncthreads();
    if (wv == 0) {
      unsigned c = (lane < R * 4) ? (unsigned)__popcll(s_ballot[lane]) : 0u;
      unsigned inc = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
      if (lane < R * 4) s_pref[lane] = inc - c;
      unsigned totalc = __shfl(inc, R * 4 - 1, 64);
      if (lane == 0) s_pref[R * 4] = totalc;
      u64 e = lookback(status, tile, (u64)totalc);
      if (lane == 0) { s_base = e; if (tile == ntiles - 1) *total = e + totalc; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j) {
      u64 b = s_ballot[j * 4 + wv];
      unsigned rk = s_pref[j * 4 + wv] +
                    __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0));
      if (sel[j]) s_sel[rk] = (unsigned short)(tid + j * 256);
    }
    __syncthreads();
    const int cnt = (int)s_pref[R * 4];
    const u64 base = s_base;
    if (!X4STORE) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float* o = outs[c] + base;
        for (int k = tid; k < cnt; k += 256) o[k] = s_col[c][s_sel[k]];
      }
    } else {
      // align global stores to 16 B: slot k maps to output index base + k; choose k0 = -(base & 3)
      const int shift = (int)(base & 3);
      for (int q = tid; q * 4 - shift < cnt; q += 256) {
        int k = q * 4 - shift;
        if (k >= 0 && k + 3 < cnt) {
          unsigned short i0 = s_sel[k], i1 = s_sel[k + 1], i2 = s_sel[k + 2], i3 = s_sel[k + 3];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            float4 x = make_float4(s_col[c][i0], s_col[c][i1], s_col[c][i2], s_col[c][i3]);
            *reinterpret_cast<float4*>(outs[c] + base + k) = x;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int kk = k + e;
            if (kk >= 0 && kk < cnt) {
              unsigned short ii = s_sel[kk];
#pragma unroll
              for (int c = 0; c < NC; ++c) outs[c][base + kk] = s_col[c][ii];
            }
          }
        }
      }
    }
    __syncthreads();
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

  // copy roofline: read 12n write 12n bytes... use col0->out0 etc. over 3 arrays
  for (int g : {2048, 4096, 8192}) {
    char nm[64]; snprintf(nm, 64, "copy_f4 grid=%d", g);
    time_it(nm, b, [&] { for (int c = 0; c < 3; ++c) copy_kernel<<<g, 256>>>((const float4*)b.cols[c], (float4*)b.outs[c], n / 4); },
            iters, (double)n * 24.0);
  }
#define RUN_REG(R, WPC) { char nm[64]; snprintf(nm, 64, "reg R=%d wg/cu=%d", R, WPC); \
    time_it(nm, b, [&] { filter_reg<R, 3><<<cus * WPC, 256>>>(b.d_cols, b.d_outs, n, 10.0f, b.status, b.ticket, b.total); }, iters, 0); \
    printf("   verify: %s\n", verify(b, std::min<size_t>(n, 8u << 20)) ? "ok" : "FAIL"); }
#define RUN_LDS(R, WPC, X4) { char nm[64]; snprintf(nm, 64, "lds R=%d wg/cu=%d x4=%d", R, WPC, X4); \
    time_it(nm, b, [&] { filter_lds<R, 3, X4><<<cus * WPC, 256>>>(b.d_cols, b.d_outs, n, 10.0f, b.status, b.ticket, b.total); }, iters, 0); \
    printf("   verify: %s\n", verify(b, std::min<size_t>(n, 8u << 20)) ? "ok" : "FAIL"); }
  RUN_REG(8, 4); RUN_REG(8, 8); RUN_REG(16, 4); RUN_REG(16, 6); RUN_REG(4, 8);
  RUN_LDS(8, 4, false); RUN_LDS(8, 4, true); RUN_LDS(8, 6, true); RUN_LDS(16, 2, true); RUN_LDS(16, 3, true); RUN_LDS(4, 8, true);
  return 0;
}
