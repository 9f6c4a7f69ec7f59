// config3_regs.hip -- round-3 experiment: BASELINE config 3 with every predicate / output column of a tile held
// in REGISTERS from the predicate to the compacting stores, C right behind P, no LDS stash and no second read.
//   predicate ((a + b > c) AND (d < 5.0)) OR (e > 1.0) over a:Int32, b:Float32, c:Float32, d:Int32, e:Float32; SELECT *
//   tile = BLOCK x R rows, BLOCK = 512 (8 waves), two workgroups per CU: one loads while the other waits on its look-back
// Hand-specialised (no interpreter): an upper bound for what a generic register-resident instantiation could reach.  It reuses
// the product's scan helpers (kernels.hip compiled without launchers).  Build + run: scripts/gpu_c3regs.sh
#define CHQ_TU 8
#include "../../chapterhouseqe_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>

namespace chq {

struct C3Params {
  const int32_t* a; const float* b; const float* c; const int32_t* d; const float* e;
  int32_t* oa; float* ob; float* oc; int32_t* od; float* oe;
  int64_t nrows;
  u64* status; uint32_t* ticket; u64* total;
};

template <int BLOCK, int R>
__global__ __launch_bounds__(BLOCK, 4) void
// (HIP: the second number of __launch_bounds__ is waves per SIMD -- 4 = sixteen waves per CU, a 128-VGPR budget)
c3_regs_kernel
(const C3Params p) {
  constexpr int NW = BLOCK / 64;
  constexpr int64_t TILE = (int64_t)BLOCK * R;
  __shared__ unsigned s_wave_cnt[NW];
  __shared__ int64_t s_tile;
  __shared__ u64 s_base;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ntiles = (int)(p.nrows / TILE);   // complete tiles only (the experiment's row count is a multiple of the tile)
  while (true) {
    const int lane = fresh_lane(), tid = wv * 64 + lane;
    if (tid == 0) s_tile = (int64_t)atomicAdd(p.ticket, 1u);
    __syncthreads();
    const int64_t tile = uniform64(s_tile);
    if ((int)tile >= ntiles) break;
    const int64_t w0 = tile * TILE + (int64_t)wv * 64 * R;
    uint32_t va[R], vb[R], vc[R], vd[R], ve[R];
    {
      const uint32_t* pa = (const uint32_t*)p.a + w0; const uint32_t* pb = (const uint32_t*)p.b + w0; const uint32_t* pc = (const uint32_t*)p.c + w0;
      const uint32_t* pd = (const uint32_t*)p.d + w0; const uint32_t* pe = (const uint32_t*)p.e + w0;
#pragma unroll
      for (int j = 0; j < R; ++j) va[j] = pa[j * 64 + lane];
#pragma unroll
      for (int j = 0; j < R; ++j) vb[j] = pb[j * 64 + lane];
#pragma unroll
      for (int j = 0; j < R; ++j) vc[j] = pc[j * 64 + lane];
#pragma unroll
      for (int j = 0; j < R; ++j) vd[j] = pd[j * 64 + lane];
#pragma unroll
      for (int j = 0; j < R; ++j) ve[j] = pe[j * 64 + lane];
    }
    uint32_t selv = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const float s = (float)(int32_t)va[j] + __uint_as_float(vb[j]);
      const bool t1 = f32_key(__float_as_uint(s)) > f32_key(vc[j]);
      const bool t2 = f32_key(__float_as_uint((float)(int32_t)vd[j])) < f32_key(__float_as_uint(5.0f));
      const bool t3 = (int32_t)ve[j] > (int32_t)__float_as_uint(1.0f);   // literal with a clear sign bit: raw bits order like the keys
      selv |= (uint32_t)((t1 && t2) || t3) << j;
    }
    unsigned cnt = __popc(selv);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) s_wave_cnt[wv] = cnt;
    __syncthreads();
    if (wv == 0) {
      unsigned c = (lane < NW) ? s_wave_cnt[lane] : 0u;
      const u64 tot = wave_sum((u64)c);
      if (lane == 0) st_store(&p.status[tile], (tile == 0 ? ST_INC : ST_AGG) | tot);
      u64 excl = 0;
      if (tile > 0) {
        excl = lookback_exclusive(p.status, tile, 0, lane);
        if (lane == 0) st_store(&p.status[tile], ST_INC | (excl + tot));
      }
      if (lane == 0) { s_base = excl; if ((int)tile == ntiles - 1) *p.total = excl + tot; }
    }
    __syncthreads();
    u64 off0 = s_base;
    for (int w = 0; w < wv; ++w) off0 += s_wave_cnt[w];
    off0 = (u64)uniform64((int64_t)off0);
    auto st = [&](uint32_t* out, const uint32_t (&v)[R]) {
      uint32_t* dst = out + off0;
      unsigned run = 0;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const bool sel = (selv >> j) & 1;
        const u64 m = __ballot(sel);
        if (sel) dst[run + lane_rank(m)] = v[j];
        run += __popcll(m);
      }
    };
    st((uint32_t*)p.oa, va); st((uint32_t*)p.ob, vb); st((uint32_t*)p.oc, vc); st((uint32_t*)p.od, vd); st((uint32_t*)p.oe, ve);
    __syncthreads();   // s_tile / s_wave_cnt / s_base are reused by the next tile
  }
}

// config 2 / 2b in the same structure: three Float32 columns, predicate `value2 > thr`, SELECT *
struct C2Params { const float* v[3]; float* o[3]; int64_t nrows; u64* status; uint32_t* ticket; u64* total; float thr; };
template <int BLOCK, int R>
__global__ __launch_bounds__(BLOCK, 4) void c2_regs_kernel(const C2Params p) {
  constexpr int NW = BLOCK / 64;
  constexpr int64_t TILE = (int64_t)BLOCK * R;
  __shared__ unsigned s_wave_cnt[NW];
  __shared__ int64_t s_tile;
  __shared__ u64 s_base;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ntiles = (int)(p.nrows / TILE);
  const int32_t thr = (int32_t)__float_as_uint(p.thr);
  while (true) {
    const int lane = fresh_lane(), tid = wv * 64 + lane;
    if (tid == 0) s_tile = (int64_t)atomicAdd(p.ticket, 1u);
    __syncthreads();
    const int64_t tile = uniform64(s_tile);
    if ((int)tile >= ntiles) break;
    const int64_t w0 = tile * TILE + (int64_t)wv * 64 * R;
    uint32_t v0[R], v1[R], v2[R];
    {
      const uint32_t* p2 = (const uint32_t*)p.v[2] + w0; const uint32_t* p0 = (const uint32_t*)p.v[0] + w0; const uint32_t* p1 = (const uint32_t*)p.v[1] + w0;
#pragma unroll
      for (int j = 0; j < R; ++j) v2[j] = p2[j * 64 + lane];
#pragma unroll
      for (int j = 0; j < R; ++j) v0[j] = p0[j * 64 + lane];
#pragma unroll
      for (int j = 0; j < R; ++j) v1[j] = p1[j * 64 + lane];
    }
    uint32_t selv = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) selv |= (uint32_t)((int32_t)v2[j] > thr) << j;
    unsigned cnt = __popc(selv);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) s_wave_cnt[wv] = cnt;
    __syncthreads();
    if (wv == 0) {
      unsigned c = (lane < NW) ? s_wave_cnt[lane] : 0u;
      const u64 tot = wave_sum((u64)c);
      if (lane == 0) st_store(&p.status[tile], (tile == 0 ? ST_INC : ST_AGG) | tot);
      u64 excl = 0;
      if (tile > 0) {
        excl = lookback_exclusive(p.status, tile, 0, lane);
        if (lane == 0) st_store(&p.status[tile], ST_INC | (excl + tot));
      }
      if (lane == 0) { s_base = excl; if ((int)tile == ntiles - 1) *p.total = excl + tot; }
    }
    __syncthreads();
    u64 off0 = s_base;
    for (int w = 0; w < wv; ++w) off0 += s_wave_cnt[w];
    off0 = (u64)uniform64((int64_t)off0);
    auto st = [&](uint32_t* out, const uint32_t (&v)[R]) {
      uint32_t* dst = out + off0;
      unsigned run = 0;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const bool sel = (selv >> j) & 1;
        const u64 m = __ballot(sel);
        if (sel) dst[run + lane_rank(m)] = v[j];
        run += __popcll(m);
      }
    };
    st((uint32_t*)p.o[0], v0); st((uint32_t*)p.o[1], v1); st((uint32_t*)p.o[2], v2);
    __syncthreads();
  }
}

__global__ void c3_gen_kernel(int32_t* a, float* b, float* c, int32_t* d, float* e, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0xC0FFEE;
    auto next = [&]() { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return (uint32_t)(x >> 32); };
    a[i] = (int32_t)(next() % 1000u);
    b[i] = (float)(next() >> 8) * (100.0f / 16777216.0f);
    c[i] = (float)(next() >> 8) * (1100.0f / 16777216.0f);
    d[i] = (int32_t)(next() % 10u);
    e[i] = (float)(next() >> 8) * (2.0f / 16777216.0f);
  }
}
// reference: count + checksum of the selected rows (order-independent part) and an order witness
__global__ void c3_ref_kernel(const int32_t* a, const float* b, const float* c, const int32_t* d, const float* e, int64_t n, u64* out) {
  u64 cnt = 0, sum = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float s = (float)a[i] + b[i];
    if ((s > c[i] && (float)d[i] < 5.0f) || e[i] > 1.0f) { ++cnt; sum += (u64)(uint32_t)a[i] * 31u + (u64)__float_as_uint(e[i]); }
  }
  cnt = wave_sum(cnt); sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], cnt); atomicAdd(&out[1], sum); }
}
__global__ void c3_sum_kernel(const int32_t* oa, const float* oe, int64_t m, u64* out) {
  u64 sum = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) sum += (u64)(uint32_t)oa[i] * 31u + (u64)__float_as_uint(oe[i]);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) atomicAdd(&out[2], sum);
}

}  // namespace chq

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int BLOCK, int R>
static int run(const char* name, chq::C3Params p, int grid_per_cu, int num_cus, chq::u64* d_chk, chq::u64 want_cnt, chq::u64 want_sum) {
  using namespace chq;
  const int64_t TILE = (int64_t)BLOCK * R;
  const int64_t ntiles = p.nrows / TILE;
  float best = 1e9f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 6; ++it) {
    CK(hipMemset(p.status, 0, (size_t)(ntiles + 1) * 8)); CK(hipMemset(p.ticket, 0, 4)); CK(hipMemset(p.total, 0, 8));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((c3_regs_kernel<BLOCK, R>), dim3(num_cus * grid_per_cu), dim3(BLOCK), 0, 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it) best = ms < best ? ms : best;
  }
  u64 total = 0; CK(hipMemcpy(&total, p.total, 8, hipMemcpyDeviceToHost));
  CK(hipMemset(d_chk + 2, 0, 8));
  hipLaunchKernelGGL(c3_sum_kernel, dim3(4096), dim3(256), 0, 0, p.oa, p.oe, (int64_t)total, d_chk);
  u64 chk[3]; CK(hipMemcpy(chk, d_chk, 24, hipMemcpyDeviceToHost));
  const double alg = (double)p.nrows * 20 + (double)total * 20;
  printf("%s: %.3f ms, %llu rows kept (want %llu) checksum %s, %.0f GB/s algorithmic = %.3f of 8 TB/s\n", name, best, (unsigned long long)total,
         (unsigned long long)want_cnt, (total == want_cnt && chk[2] == want_sum) ? "ok" : "MISMATCH", alg / (best * 1e-3) / 1e9, alg / (best * 1e-3) / 1e9 / 8000.0);
  return (total == want_cnt && chk[2] == want_sum) ? 0 : 2;
}

int main(int argc, char** argv) {
  using namespace chq;
  const int64_t n = (argc > 1 ? atoll(argv[1]) : 1000000000ll) / 16384 * 16384;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  C3Params p{};
  void* bufs[10];
  for (int i = 0; i < 10; ++i) CK(hipMalloc(&bufs[i], (size_t)n * 4 + 64));
  p.a = (int32_t*)bufs[0]; p.b = (float*)bufs[1]; p.c = (float*)bufs[2]; p.d = (int32_t*)bufs[3]; p.e = (float*)bufs[4];
  p.oa = (int32_t*)bufs[5]; p.ob = (float*)bufs[6]; p.oc = (float*)bufs[7]; p.od = (int32_t*)bufs[8]; p.oe = (float*)bufs[9];
  p.nrows = n;
  CK(hipMalloc((void**)&p.status, (size_t)(n / 2048 + 2) * 8)); CK(hipMalloc((void**)&p.ticket, 64)); CK(hipMalloc((void**)&p.total, 64));
  u64* d_chk; CK(hipMalloc((void**)&d_chk, 64)); CK(hipMemset(d_chk, 0, 64));
  hipLaunchKernelGGL(c3_gen_kernel, dim3(8192), dim3(256), 0, 0, (int32_t*)p.a, (float*)p.b, (float*)p.c, (int32_t*)p.d, (float*)p.e, n);
  hipLaunchKernelGGL(c3_ref_kernel, dim3(8192), dim3(256), 0, 0, p.a, p.b, p.c, p.d, p.e, n, d_chk);
  CK(hipDeviceSynchronize());
  u64 ref[2]; CK(hipMemcpy(ref, d_chk, 16, hipMemcpyDeviceToHost));
  printf("%lld rows, %d CUs, reference: %llu rows kept (s = %.4f)\n", (long long)n, prop.multiProcessorCount, (unsigned long long)ref[0], (double)ref[0] / n);
  int rc = 0;
  rc |= run<512, 16>("regs 512x16, 2 WG/CU", p, 2, prop.multiProcessorCount, d_chk, ref[0], ref[1]);
  rc |= run<512, 16>("regs 512x16, 3 WG/CU (launched)", p, 3, prop.multiProcessorCount, d_chk, ref[0], ref[1]);
  rc |= run<256, 16>("regs 256x16, 4 WG/CU", p, 4, prop.multiProcessorCount, d_chk, ref[0], ref[1]);
  rc |= run<512, 8>("regs 512x8, 2 WG/CU", p, 2, prop.multiProcessorCount, d_chk, ref[0], ref[1]);
  rc |= run<1024, 8>("regs 1024x8, 1 WG/CU", p, 1, prop.multiProcessorCount, d_chk, ref[0], ref[1]);
  rc |= run<1024, 8>("regs 1024x8, 2 WG/CU (launched)", p, 2, prop.multiProcessorCount, d_chk, ref[0], ref[1]);
  // ---- config 2 / 2b: columns b, c, e reused as value0..2 (e is U[0,2): thresholds 0.2 / 1.8 give s = 0.9 / 0.1) ----
  for (float thr : {0.2f, 1.0f, 1.8f}) {
    C2Params q{};
    q.v[0] = p.b; q.v[1] = p.c; q.v[2] = p.e; q.o[0] = p.ob; q.o[1] = p.oc; q.o[2] = p.oe;
    q.nrows = n; q.status = p.status; q.ticket = p.ticket; q.total = p.total; q.thr = thr;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant) {
      float best = 1e9f; u64 total = 0;
      for (int it = 0; it < 6; ++it) {
        const int64_t TILE = variant == 0 ? 512 * 16 : 1024 * 16;
        CK(hipMemset(p.status, 0, (size_t)(n / TILE + 1) * 8)); CK(hipMemset(p.ticket, 0, 4)); CK(hipMemset(p.total, 0, 8));
        CK(hipEventRecord(e0, 0));
        if (variant == 0) hipLaunchKernelGGL((c2_regs_kernel<512, 16>), dim3(prop.multiProcessorCount * 2), dim3(512), 0, 0, q);
        else hipLaunchKernelGGL((c2_regs_kernel<1024, 8>), dim3(prop.multiProcessorCount), dim3(1024), 0, 0, q);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it) best = ms < best ? ms : best;
      }
      CK(hipMemcpy(&total, p.total, 8, hipMemcpyDeviceToHost));
      const double alg = (double)n * 12 + (double)total * 12;
      printf("config 2 shape, value2 > %.1f (s = %.3f), regs %s: %.3f ms, %.0f GB/s algorithmic = %.3f of 8 TB/s\n", thr, (double)total / n,
             variant == 0 ? "512x16 2 WG/CU" : "1024x8 1 WG/CU", best, alg / (best * 1e-3) / 1e9, alg / (best * 1e-3) / 1e9 / 8000.0);
    }
  }
  return rc;
}
