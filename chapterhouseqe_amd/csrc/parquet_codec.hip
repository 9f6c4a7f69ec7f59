// parquet_codec.hip -- page decompression on the GPU ("page decompress").
//
// The reference reads arbitrary user Parquet through the `parquet` crate (read_files_task.rs:233-282), and the writers
// users have (pyarrow's default, Spark, DuckDB) compress pages with Snappy.  Here a compressed column chunk is uploaded as it
// lies in the file and every page is inflated in HBM into its place in an UNCOMPRESSED IMAGE of the chunk; the decode kernels
// of parquet.hip then run on that image exactly as they run on an uncompressed chunk.
//
// Snappy (raw format, github.com/google/snappy format_description.txt): a varint with the uncompressed length, then elements
//   tag & 3 == 0  literal: length (tag >> 2) + 1, or for tag >> 2 = 60..63 the next 1..4 bytes hold length - 1
//   tag & 3 == 1  copy, 1-byte offset: length ((tag >> 2) & 7) + 4, offset ((tag >> 5) << 8) | next byte
//   tag & 3 == 2  copy, 2-byte offset: length (tag >> 2) + 1, offset = next two bytes (little endian)
//   tag & 3 == 3  copy, 4-byte offset: length (tag >> 2) + 1, offset = next four bytes
// An element's position depends on every element before it and a copy may read what the previous element wrote, so a page is
// a serial chain; pages are independent.  One wavefront per page:
//   * the last 64 KiB of OUTPUT live in an LDS ring (snappy compressors match inside 64 KiB blocks, so copy offsets stay
//     below it; larger offsets -- legal, never produced -- read back from HBM): a copy is an LDS read + an LDS write, both
//     ~64 cycles, instead of a dependent HBM round trip per element;
//   * the tags are parsed from an LDS window of the INPUT by all 64 lanes at once (uniform control flow, broadcast reads);
//   * literals stream HBM -> ring with eight loads per lane in flight; the ring is written back to HBM 16 bytes per lane.
// Memory-bound byte work: no MFMA.  Rates: DESIGN.md section 3.5.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "parquet_device.h"

namespace chq {
namespace {

constexpr uint32_t SN_RING = 65536;         // bytes of output history in LDS
constexpr uint32_t SN_MASK = SN_RING - 1;
constexpr uint32_t SN_WIN = 4096;           // bytes of input staged for the tag parser
constexpr uint32_t SN_FLUSH = 16384;        // ring -> HBM once this much is pending

__device__ __forceinline__ void codec_error(uint32_t* err) { atomicMax(err, (uint32_t)PQ_ERR_CODEC); }

}  // namespace

// One wave per job.  job.raw: the chunk as it lies in the file; job.image: the uncompressed image (both padded by 64 bytes).
__global__ __launch_bounds__(64) void pq_inflate_kernel(const PqCodecParams p) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[SN_RING];
  __shared__ __attribute__((aligned(16))) uint8_t s_win[SN_WIN + 32];
  const int lane = threadIdx.x;
  const PqCodecJob job = p.jobs[blockIdx.x];
  const uint8_t* src = job.raw + job.src_at;
  uint8_t* dst = job.image + job.dst_at;        // (dst_at is a multiple of 16)
  const uint32_t slen = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.src_len), dlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.dst_len);
  bool failed = false;
  uint32_t first4 = 0;

  if (job.codec == PQ_CODEC_STORED) {           // a page (or the level bytes of a V2 page) that lies uncompressed in the file
    if (slen != dlen) failed = true;
    else {
      for (uint32_t i = lane * 16u; i + 16 <= dlen; i += 64 * 16) { uint4 w; __builtin_memcpy(&w, src + i, 16); *(uint4*)(dst + i) = w; }
      for (uint32_t i = (dlen & ~15u) + lane; i < dlen; i += 64) dst[i] = src[i];
      if (dlen >= 4) { __builtin_memcpy(&first4, src, 4); }
    }
  } else if (job.codec == PQ_CODEC_SNAPPY) {
    uint32_t pos = 0;                           // next input byte
    // ---- preamble: the uncompressed length must be what the page header promised ----
    {
      uint32_t v = 0; int sh = 0; bool done = false;
      while (!done && pos < slen && sh < 35) { const uint8_t b = src[pos++]; v |= (uint32_t)(b & 0x7f) << sh; sh += 7; done = !(b & 0x80); }
      if (!done || v != dlen) failed = true;
    }
    uint32_t out = 0, flushed = 0;              // bytes produced / bytes already written back (multiple of 16)
    uint32_t wlo = 0, wend = 0, wbias = 0;      // input bytes [wlo, wend) are staged: s_win[k] = input byte wbias + k
    bool got4 = false;                          // the first four output bytes were captured (while they are still in the ring)
    auto flush = [&](uint32_t upto) {           // ring [flushed, upto) -> HBM; upto is a multiple of 16 (or the end)
      for (uint32_t i = flushed + lane * 16u; i + 16 <= upto; i += 64 * 16) *(uint4*)(dst + i) = *(const uint4*)(s_ring + (i & SN_MASK));
      const uint32_t tail = upto & ~15u;
      if (tail >= flushed) for (uint32_t i = tail + lane; i < upto; i += 64) dst[i] = s_ring[i & SN_MASK];
      flushed = upto & ~15u;
    };
    // Input window: [wlo, wend) of the input is staged in s_win.  Everything that steers the loop -- positions, lengths,
    // offsets -- is wave-uniform and is kept in SCALAR registers: the element header is read from LDS by all lanes (same
    // address: a broadcast) and moved to SGPRs with readfirstlane, so the parse is SALU work (one cycle per instruction
    // instead of four, scalar branches) and only the byte movement runs on the vector unit.  The header is fetched one
    // element AHEAD: as soon as an element's length is known the next element's position is, and its header is requested
    // before the current element's bytes are moved -- the LDS round trips overlap instead of adding up.  (The first version
    // parsed in vector registers and fetched on demand: ~1000 cycles per element, 28 ms for a 1 MB page of short elements.)
    auto refill = [&](uint32_t at) {
      wlo = at & ~3u;
      const uint32_t n = slen - wlo < SN_WIN ? slen - wlo : SN_WIN;
      const uint32_t skew = (uint32_t)((uintptr_t)(src + wlo) & 3);
      const uint32_t* g = (const uint32_t*)(src + wlo - skew);   // aligned dword loads (the raw buffer is padded by 64 bytes)
      for (uint32_t i = lane; i < (n + skew + 3) / 4 + 2; i += 64) ((uint32_t*)s_win)[i] = g[i];
      wend = wlo + n;
      wbias = wlo - skew;                     // (may wrap below zero: only ever used in `pos - wbias`)
      __builtin_amdgcn_wave_barrier();        // one wave: its LDS accesses execute in order, the compiler must keep them so
    };
    auto staged = [&](uint32_t at) { return at >= wlo && (at + 5 <= wend || wend >= slen); };
    uint32_t n0 = 0, n1 = 0, nsh = 0;         // the two dwords holding the next header (still in flight) and its byte phase
    auto fetch = [&](uint32_t at) {
      const uint32_t o = at - wbias;
      const uint32_t* w = (const uint32_t*)(s_win + (o & ~3u));
      n0 = w[0]; n1 = w[1]; nsh = (o & 3u) * 8u;
    };
    if (!failed && pos < slen) { refill(pos); fetch(pos); }
    while (!failed && pos < slen && out < dlen) {
      // header bytes of the element at `pos`: tag, b1 .. b4 (bytes past the input are ignored below)
      const uint64_t hv = (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)n1) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)n0)) >> nsh;
      const uint32_t tag = (uint32_t)hv & 0xffu;
      const uint32_t b14 = (uint32_t)(hv >> 8);               // b1 | b2 << 8 | b3 << 16 | b4 << 24
      const uint32_t kind = tag & 3u;
      uint32_t len, off = 0, hdr;
      if (kind == 0) {
        len = tag >> 2; hdr = 1;
        if (len >= 60) {
          const uint32_t nb = len - 59;
          len = nb == 4 ? b14 : (b14 & ((1u << (8 * nb)) - 1u));
          hdr = 1 + nb;
        }
        if (len == 0xffffffffu || pos + hdr > slen) { failed = true; break; }
        len += 1;
        if (len > slen - pos - hdr || len > dlen - out) { failed = true; break; }
      } else {
        if (kind == 1) { len = ((tag >> 2) & 7u) + 4; off = ((tag >> 5) << 8) | (b14 & 0xffu); hdr = 2; }
        else if (kind == 2) { len = (tag >> 2) + 1; off = b14 & 0xffffu; hdr = 3; }
        else { len = (tag >> 2) + 1; off = b14; hdr = 5; }
        if (pos + hdr > slen || off == 0 || off > out || len > dlen - out) { failed = true; break; }
      }
      const uint32_t data = pos + hdr;                          // a literal's bytes start here
      const uint32_t next = kind == 0 ? data + len : data;      // the next element
      const bool ahead = next < slen && staged(next);
      if (ahead) fetch(next);
      if (kind == 0) {
        // ---- literal ----
        uint32_t done = 0;
        if (data + len <= wend && data >= wlo) {
          // a literal that lies inside the staged input window -- the common case by far: text-like data alternates copies
          // with literals of a few bytes -- moves LDS -> LDS (from HBM it would cost a memory round trip per element)
          const uint32_t sbase = data - wbias;
          for (; done < len; done += 64) {
            const uint32_t i = done + lane;
            uint8_t b = 0;
            if (i < len) b = s_win[sbase + i];
            if (i < len) s_ring[(out + i) & SN_MASK] = b;
          }
          done = len;
          if (!got4 && out + done >= 4) { first4 = *(const uint32_t*)s_ring; got4 = true; }
          if (out + done - flushed >= SN_FLUSH) flush((out + done) & ~15u);
        }
        // otherwise HBM -> ring, 512 bytes per round (eight byte loads per lane in flight); rounds are interleaved with
        // write-backs so that a long literal (incompressible data: one literal per 64 KiB block) never overruns the ring
        while (done < len) {
          const uint32_t n = len - done < 512 ? len - done : 512;
          uint8_t v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { const uint32_t i = u * 64 + lane; v[u] = i < n ? src[data + done + i] : 0; }
#pragma unroll
          for (int u = 0; u < 8; ++u) { const uint32_t i = u * 64 + lane; if (i < n) s_ring[(out + done + i) & SN_MASK] = v[u]; }
          done += n;
          if (!got4 && out + done >= 4) { first4 = *(const uint32_t*)s_ring; got4 = true; }   // (at most 512 bytes in: still there)
          if (out + done - flushed >= SN_FLUSH) flush((out + done) & ~15u);
        }
        out += len;
      } else {
        // ---- copy: byte i is byte (i mod off) of the `off` bytes in front of it -- every source byte was final before
        // this element started, so the (at most 64) bytes are independent of each other: one lane each ----
        uint32_t i = lane;
        if (off < len) i = (uint32_t)lane % off;
        uint8_t b = 0;
        if (off <= SN_RING - 64) {
          if ((uint32_t)lane < len) b = s_ring[(out - off + i) & SN_MASK];
        } else {                                // an offset beyond the ring: read what was written back (never produced by
          flush(out);                           // the snappy library; the tail bytes of a partial 16-byte group included)
          __threadfence();
          if ((uint32_t)lane < len) b = __hip_atomic_load(dst + (out - off + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((uint32_t)lane < len) s_ring[(out + lane) & SN_MASK] = b;
        out += len;
        if (!got4 && out >= 4) { first4 = *(const uint32_t*)s_ring; got4 = true; }   // (a copy is at most 64 bytes: still there)
        if (out - flushed >= SN_FLUSH) flush(out & ~15u);
      }
      pos = next;
      if (!ahead && pos < slen && out < dlen) { refill(pos); fetch(pos); }   // (the window is only replaced once the element that read from it is done)
    }
    if (!failed && (out != dlen || pos != slen)) failed = true;
    if (!failed) flush(out);
  } else {
    failed = true;
  }
  if (failed) { if (lane == 0) codec_error(job.err); return; }
  // ---- a V1 data page of an optional column starts with [4-byte length][definition levels]: only now can its parts be told
  // apart -- patch the page's descriptor (the decode kernels launched behind this one read it from HBM) ----
  if (job.page >= 0 && lane == 0) {
    PqPageDesc& d = job.pages[job.page];
    const uint32_t l = first4;
    if (dlen < 4 || l > dlen - 4) { codec_error(job.err); d.levels_len = 0; d.values_at = job.dst_at; d.values_len = 0; return; }
    d.levels_at = job.dst_at + 4;
    d.levels_len = (job.flags & PQ_JOB_KEEP_LEVELS) ? l : 0u;
    d.values_at = job.dst_at + 4 + l;
    d.values_len = dlen - 4 - l;
  }
}

hipError_t pq_launch_inflate(const PqCodecParams& p, hipStream_t s) {
  if (p.n_jobs <= 0) return hipSuccess;
  hipLaunchKernelGGL(pq_inflate_kernel, dim3(p.n_jobs), dim3(64), 0, s, p);
  return hipGetLastError();
}

}  // namespace chq
