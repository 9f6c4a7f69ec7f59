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



// Pipelined single pass: P(i+1) [ticket, predicate, ballots->LDS, publish AGG] runs before C(i) [look-back, copy].
// Look-back uses LBW waves in parallel windows of 64 tiles.
template <int BLOCK, int R, int NC, int LBW, int MINW, bool STATIC>
__global__ __launch_bounds__(BLOCK, MINW) void filter_pipe(const float* const* __restrict__ cols_, float* const* __restrict__ outs_,
                                                    size_t n, float thr, u64* status, unsigned* ticket, u64* total) {
  constexpr int NW = BLOCK / 64;
  __shared__ u64 s_mask[2][R][NW];
  __shared__ unsigned s_wave_cnt[2][NW];
  __shared__ unsigned s_tot[2];
  __shared__ int s_tile[2];
  __shared__ u64 s_part[LBW];
  __shared__ int s_has[LBW];
  __shared__ u64 s_base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* cols[NC]; float* outs[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { cols[c] = cols_[c]; outs[c] = outs_[c]; }
  const int kthr = tot_key(thr);
  const size_t TILE = (size_t)BLOCK * R;
  const int ntiles = (int)((n + TILE - 1) / TILE);

  int pk = 0;
  auto P = [&](int buf) {
    if (STATIC) { if (tid == 0) s_tile[buf] = (int)(blockIdx.x + (unsigned)pk * gridDim.x); ++pk; }
    else if (tid == 0) s_tile[buf] = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = s_tile[buf];
    if (tile >= ntiles) return;
    const size_t w0 = (size_t)tile * TILE + (size_t)wv * 64 * R;
    float v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      size_t r = w0 + (size_t)j * 64 + lane;
      v[j] = (r < n) ? cols[NC - 1][r] : 0.f;
    }
    unsigned cnt = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      size_t r = w0 + (size_t)j * 64 + lane;
      u64 m = __ballot((r < n) && (tot_key(v[j]) > kthr));
      if (lane == 0) s_mask[buf][j][wv] = m;
      cnt += __popcll(m);
    }
    if (lane == 0) s_wave_cnt[buf][wv] = cnt;
    __syncthreads();
    if (wv == 0) {
      unsigned c = (lane < NW) ? s_wave_cnt[buf][lane] : 0u;
      u64 tot = wave_sum((u64)c);
      if (lane == 0) {
        s_tot[buf] = (unsigned)tot;
        if (tile == 0) st_store(&status[0], ST_INC | tot); else st_store(&status[tile], ST_AGG | tot);
      }
    }
  };

  auto C = [&](int buf) {
    const int tile = s_tile[buf];
    const size_t w0 = (size_t)tile * TILE + (size_t)wv * 64 * R;
    float v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      size_t r = w0 + (size_t)j * 64 + lane;
      v[j] = (r < n) ? cols[0][r] : 0.f;
    }
    if (wv < LBW && tile > 0) {
      int idx = tile - 1 - (wv * 64 + lane);
      u64 w = ST_INC;
      if (idx >= 0) {
        w = st_load(&status[idx]);
        while (ST_FLAG(w) == 0) { __builtin_amdgcn_s_sleep(1); w = st_load(&status[idx]); }
      }
      u64 incm = __ballot(ST_FLAG(w) == 2);
      u64 val;
      if (incm) { int first = __builtin_ctzll(incm); val = (lane <= first) ? ST_VAL(w) : 0; }
      else val = ST_VAL(w);
      val = wave_sum(val);
      if (lane == 0) { s_part[wv] = val; s_has[wv] = incm != 0; }
    }
    __syncthreads();
    if (wv == 0) {
      u64 excl = 0;
      if (tile > 0) {
        bool done = false;
        for (int w = 0; w < LBW; ++w) { excl += s_part[w]; if (s_has[w]) { done = true; break; } }
        int look = tile - 1 - LBW * 64;
        while (!done) {  // rare fallback: continue serially
          int idx = look - lane;
          u64 w = ST_INC;
          if (idx >= 0) {
            w = st_load(&status[idx]);
            while (ST_FLAG(w) == 0) { __builtin_amdgcn_s_sleep(1); w = st_load(&status[idx]); }
          }
          u64 incm = __ballot(ST_FLAG(w) == 2);
          if (incm) { int first = __builtin_ctzll(incm); excl += wave_sum((lane <= first) ? ST_VAL(w) : 0); done = true; }
          else { excl += wave_sum(ST_VAL(w)); look -= 64; }
        }
        if (lane == 0) st_store(&status[tile], ST_INC | (excl + s_tot[buf]));
      }
      if (lane == 0) { s_base = excl; if (tile == ntiles - 1) *total = excl + s_tot[buf]; }
    }
    __syncthreads();
    u64 off0 = s_base;
    for (int w = 0; w < wv; ++w) off0 += s_wave_cnt[buf][w];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c > 0) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
          size_t r = w0 + (size_t)j * 64 + lane;
          v[j] = (r < n) ? cols[c][r] : 0.f;
        }
      }
      u64 off = off0;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        u64 m = s_mask[buf][j][wv];
        bool sel = (m >> lane) & 1;
        unsigned rk = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (sel) outs[c][off + rk] = v[j];
        off += __popcll(m);
      }
    }
  };

  P(0);
  int it = 0;
  while (true) {
    if (s_tile[it & 1] >= ntiles) break;
    P((it + 1) & 1);
    C(it & 1);
    ++it;
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



  float thr = argc > 3 ? atof(argv[3]) : 10.0f;
#define RUN_PIPE(BLOCK, R, WPC, LBW, MINW, ST) { char nm[96]; snprintf(nm, 96, "pipe B=%d R=%d wg/cu=%d lbw=%d minw=%d static=%d", BLOCK, R, WPC, LBW, MINW, ST); \
    time_it(nm, b, [&] { filter_pipe<BLOCK, R, 3, LBW, MINW, ST><<<cus * WPC, BLOCK>>>(b.d_cols, b.d_outs, n, thr, b.status, b.ticket, b.total); }, iters, 0); \
    if (thr == 10.0f) printf("   verify: %s\n", verify(b, std::min<size_t>(n, 8u << 20)) ? "ok" : "FAIL"); }
  RUN_PIPE(1024, 16, 1, 1, 4, false); RUN_PIPE(1024, 16, 1, 1, 4, true);
  RUN_PIPE(1024, 16, 2, 2, 8, false); RUN_PIPE(1024, 16, 2, 2, 8, true);
  RUN_PIPE(1024, 8, 2, 2, 8, false); RUN_PIPE(1024, 8, 2, 2, 8, true);
  RUN_PIPE(512, 16, 4, 4, 8, false); RUN_PIPE(512, 16, 4, 4, 8, true);
  RUN_PIPE(512, 16, 2, 2, 4, true); RUN_PIPE(256, 16, 4, 4, 4, true); RUN_PIPE(256, 16, 8, 8, 8, true);
  return 0;
}
