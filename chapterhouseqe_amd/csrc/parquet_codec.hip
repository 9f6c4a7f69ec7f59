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
// a serial chain; pages are independent.  The parallelism around that chain (details at snappy_stream below):
//   * lanes: the chain is parsed 64 INPUT bytes at a time -- every lane decodes the header that would start at its byte, the
//     scalar unit follows the chain through them with readlane, a batch's literals and copies are moved together;
//   * waves: a page of three or more 64 KiB blocks of output is walked once without moving bytes (INDEX job) to find where
//     each block starts in the input, then inflated one wave per block (BLOCK jobs); a FINISH job patches the page
//     descriptor and redoes the page in the (never produced) case that its blocks depend on each other;
//   * the last 64 KiB of OUTPUT live in an LDS ring (the format's maximum offset within a compressor's block; larger
//     offsets -- legal, never produced -- read back from HBM): a copy is an LDS read + an LDS write instead of a dependent
//     HBM round trip per element; the ring is written back to HBM 16 bytes per lane;
//   * the input is staged through a 4 KiB LDS window; long literals stream HBM -> ring with eight loads per lane in flight.
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

// The wave's inclusive prefix sum (DPP: shifts inside the rows of 16 lanes, then the row totals broadcast to the rows behind).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1 (a lane without a source adds 0)
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return x;
}

enum { SN_FULL = 0, SN_INDEX = 1 };

// One raw snappy stream (`preamble`: it starts with its uncompressed length, which must be dlen), walked by one wave.
//   SN_FULL   the bytes are produced: src[0, slen) -> dst[0, dlen); first4 = the first four bytes of output.
//   SN_INDEX  nothing is moved: the element chain is walked only to learn where in the input every 64 KiB BLOCK of output
//             starts (index[1 + k] = input position of the element that produces output byte k * 65536).  The snappy
//             compressor works on 64 KiB blocks of input one at a time -- no element straddles a block boundary and no copy
//             reaches into the previous block -- so with that table the blocks of a page are inflated by one wave EACH
//             (the walk costs a few scalar instructions per element, the moves an LDS round trip or more).  The format
//             does not promise it: `aligned` comes back false if an element straddles a boundary, a block job that meets
//             an offset reaching below its block gives up, and either way the page is then redone by one wave (FINISH).
// Returns true if the stream is damaged.
// A part of a stream's element chain (SN_INDEX): the walk begins at input byte `start` with `out0` bytes of output behind it
// and ends at the first element that starts at or behind `stop`.  `whole`: it must end at the stream's end with dlen bytes of
// output.  `lead`: the start is a guess (see pq_inflate_index_kernel) -- an element too long for a batch is then taken for a
// misread, not followed.  Where the walk ended comes back in end_pos / end_out, where the elements begin (behind the
// preamble) in `first`.
struct SnappySpan { uint32_t start, stop, out0; bool whole, lead; uint32_t end_pos, end_out, first; };

template <int MODE>
__device__ __forceinline__ bool snappy_stream(const uint8_t* src, const uint32_t slen, uint8_t* dst, const uint32_t dlen, const bool preamble,
                                               uint32_t* index, bool& aligned, uint32_t& first4, uint8_t* s_ring, uint8_t* s_win, const int lane,
                                               SnappySpan* span = nullptr) {
  bool failed = false;
  const uint32_t stop = span ? span->stop : slen;
  uint32_t pos = span ? span->start : 0;      // next input byte
  // ---- preamble: the uncompressed length must be what the page header promised ----
  if (preamble) {
    uint32_t v = 0; int sh = 0; bool done = false;
    while (!done && pos < slen && sh < 35) { const uint8_t b = src[pos++]; v |= (uint32_t)(b & 0x7f) << sh; sh += 7; done = !(b & 0x80); }
    if (!done || v != dlen) failed = true;
  }
  // (every lane loaded the same bytes: tell the compiler, so that what steers the loops below lives in scalar registers)
  pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
  failed = __builtin_amdgcn_readfirstlane((int)failed) != 0;
  if (span) span->first = pos;
  uint32_t out = span ? span->out0 : 0, flushed = 0;   // bytes produced / bytes already written back (multiple of 16)
  uint32_t wlo = 0, wend = 0, wbias = 0;      // input bytes [wlo, wend) are staged: s_win[k] = input byte wbias + k
  bool got4 = false;                          // the first four output bytes were captured (while they are still in the ring)
  auto flush = [&](uint32_t upto) {           // ring [flushed, upto) -> HBM; upto is a multiple of 16 (or the end)
    for (uint32_t i = flushed + lane * 16u; i + 16 <= upto; i += 64 * 16) *(uint4*)(dst + i) = *(const uint4*)(s_ring + (i & SN_MASK));
    const uint32_t tail = upto & ~15u;
    if (tail >= flushed) for (uint32_t i = tail + lane; i < upto; i += 64) dst[i] = s_ring[i & SN_MASK];
    flushed = upto & ~15u;
  };
  // Input window: [wlo, wend) of the input is staged in s_win (refilled about once per 4000 input bytes).
  auto refill = [&](uint32_t at) {
    wlo = at & ~3u;
    const uint32_t n = slen - wlo < SN_WIN ? slen - wlo : SN_WIN;
    const uint32_t skew = (uint32_t)((uintptr_t)(src + wlo) & 3);
    const uint32_t* g = (const uint32_t*)(src + wlo - skew);   // aligned dword loads (the raw buffer is padded by 64 bytes)
    // (all of a lane's loads are issued before the first is waited for: as a loop of load - wait - store this was seventeen
    // HBM round trips per window, ~40 us, and the largest cost per input byte)
    const uint32_t cnt = (n + skew + 3) / 4 + 2;                 // at most SN_WIN / 4 + 3 dwords
    uint32_t t[SN_WIN / 256 + 1];
#pragma unroll
    for (uint32_t k = 0; k < SN_WIN / 256 + 1; ++k) { const uint32_t i = k * 64 + lane; t[k] = i < cnt ? g[i] : 0u; }
#pragma unroll
    for (uint32_t k = 0; k < SN_WIN / 256 + 1; ++k) { const uint32_t i = k * 64 + lane; if (i < cnt) ((uint32_t*)s_win)[i] = t[k]; }
    wend = wlo + n;
    wbias = wlo - skew;                     // (may wrap below zero: only ever used in `pos - wbias`)
    __builtin_amdgcn_wave_barrier();        // one wave: its LDS accesses execute in order, the compiler must keep them so
  };
  // One copy element moved by the whole wave (byte i is byte (i mod off) of the `off` bytes in front of it; every source
  // byte was final before the element started, so its at most 64 bytes are independent of each other: one lane each).
  auto wave_copy = [&](uint32_t at, uint32_t off, uint32_t len) {
    uint32_t i = lane;
    if (off < len) i = (uint32_t)lane % off;
    uint8_t b = 0;
    if (off <= SN_RING - 64) {
      if ((uint32_t)lane < len) b = s_ring[(at - off + i) & SN_MASK];
    } else {                                // an offset beyond the ring: read what was written back (`out` is where the
      flush(out);                           // finished output ends: the source lies below it; tail bytes included)
      __threadfence();
      if ((uint32_t)lane < len) b = __hip_atomic_load(dst + (at - off + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((uint32_t)lane < len) s_ring[(at + lane) & SN_MASK] = b;
  };
  if (dlen > 0x7fffffffu || slen > 0x7fffffffu) failed = true;   // (Parquet page sizes are i32; keeps the sums below in 32 bits)
  // The element chain is parsed SIXTY-FOUR INPUT BYTES AT A TIME.  Every lane decodes the header that would start at its
  // byte (tag, length, offset: vector work, done for all 64 positions at once, most of them not element starts); the chain
  // "element at lane c -> next element at lane c + advance(c)" is then followed by the scalar unit with readlane (a few
  // SALU instructions per element, no memory access), which also hands every element its output position.  The bytes are
  // moved for the whole batch of elements -- typically 20 to 30 -- together:
  //   1. copies whose source the batch's own writes would overrun in the ring (offsets near 64 KiB) go first, in order;
  //   2. every literal byte of the batch lies in the 64-byte input slice: lane p writes its byte to its place;
  //   3. the other copies run one LANE per element, in rounds: a copy is ready once its source ends at or below the first
  //      unfinished copy's output (everything below that is final).  Matches mostly point far back, so one or two rounds
  //      do; a chain of copies that each read the previous one's bytes degrades to one element per round.
  // (The version before this one walked the chain one element at a time: ~1400 cycles of dependent LDS round trips and
  // issue latency per element with one wave per SIMD, 60+ ms for a 1 MB page of short elements.)
  while (!failed && pos < stop && out < dlen) {
    if (!(pos >= wlo && (pos + 72 <= wend || wend >= slen))) refill(pos);
    const uint32_t p = pos + lane;            // this lane's input byte
    uint32_t tag = 0, b14 = 0;
    if (p < slen) {
      const uint32_t o = p - wbias;
      const uint32_t* w = (const uint32_t*)(s_win + (o & ~3u));
      const uint64_t hv = (((uint64_t)w[1] << 32) | w[0]) >> ((o & 3u) * 8u);
      tag = (uint32_t)hv & 0xffu;
      b14 = (uint32_t)(hv >> 8);              // b1 | b2 << 8 | b3 << 16 | b4 << 24
    }
    const uint32_t kind = tag & 3u;
    // (selects, not branches: all 64 lanes decode a header, whatever its kind)
    const uint32_t t6 = tag >> 2;
    const uint32_t nb = kind == 0 && t6 >= 60 ? t6 - 59 : 0u;                      // literal: bytes of extended length
    const uint32_t ext = nb == 4 ? b14 : (b14 & ((1u << (8 * nb)) - 1u));
    const uint32_t len0 = nb ? ext : (kind == 1 ? (t6 & 7u) + 3 : t6);            // length - 1
    const uint32_t hdr = kind == 0 ? 1 + nb : (kind == 1 ? 2u : (kind == 2 ? 3u : 5u));
    const uint32_t off = kind == 0 ? 0u : (kind == 1 ? ((tag >> 5) << 8) | (b14 & 0xffu) : (kind == 2 ? (b14 & 0xffffu) : b14));
    const uint32_t len = len0 + 1;
    // this header cannot be an element (only matters if the chain lands on it): it, or a literal's bytes, pass the input's end
    const bool bad = p + hdr > slen || (kind == 0 && (len0 == 0xffffffffu || len > slen - p - hdr));
    const uint32_t adv = kind != 0 ? hdr : (bad || len > 64 ? 4096u : hdr + len);   // input bytes to the next element
    // ---- the chain (scalar): element starts M; the batch ends at input byte `cur` ----
    const uint32_t lim = stop - pos < 64 ? stop - pos : 64;
    // (an element must END inside the slice -- lane + adv <= 63 -- so that `cur` stays a lane number and the loop below
    // has one test per element; a longer literal, or one that reaches the slice's end, starts the next batch)
    const uint64_t fits = __builtin_amdgcn_ballot_w64((uint32_t)lane < lim && (uint32_t)lane + adv <= 63);
    uint64_t M = 0;
    uint32_t cur = 0;
    while ((fits >> cur) & 1ull) {
      M |= 1ull << cur;
      cur += (uint32_t)__builtin_amdgcn_readlane((int)adv, (int)cur);
    }
    if (M == 0) {
      // ---- the element at `pos` is a literal that does not end inside the slice (or a damaged header): moved on its own ----
      if (__builtin_amdgcn_readfirstlane((int)bad) || (span && span->lead)) { failed = true; break; }
      const uint32_t llen = (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
      const uint32_t data = pos + (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr);
      if (llen > dlen - out) { failed = true; break; }
      if (MODE == SN_INDEX) {                     // where the 64 KiB blocks of output start in the input
        if (index && (out & 0xffffu) == 0 && lane == 0) index[1 + (out >> 16)] = pos;
        if ((out ^ (out + llen - 1)) >> 16) aligned = false;
      } else {
        uint32_t done = 0;
        if (data + llen <= wend && data >= wlo) {   // inside the staged input: LDS -> LDS
          const uint32_t sbase = data - wbias;
          for (; done < llen; done += 64) {
            const uint32_t i = done + lane;
            uint8_t b = 0;
            if (i < llen) b = s_win[sbase + i];
            if (i < llen) s_ring[(out + i) & SN_MASK] = b;
          }
          done = llen;
          if (!got4 && out + done >= 4) { first4 = *(const uint32_t*)s_ring; got4 = true; }
          if (out + done - flushed >= SN_FLUSH) flush((out + done) & ~15u);
        }
        // otherwise HBM -> ring, 512 bytes per round (eight byte loads per lane in flight); rounds are interleaved with
        // write-backs so that a long literal (incompressible data: one literal per 64 KiB block) never overruns the ring
        while (done < llen) {
          const uint32_t n = llen - done < 512 ? llen - done : 512;
          uint8_t v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { const uint32_t i = u * 64 + lane; v[u] = i < n ? src[data + done + i] : 0; }
#pragma unroll
          for (int u = 0; u < 8; ++u) { const uint32_t i = u * 64 + lane; if (i < n) s_ring[(out + done + i) & SN_MASK] = v[u]; }
          done += n;
          if (!got4 && out + done >= 4) { first4 = *(const uint32_t*)s_ring; got4 = true; }   // (at most 512 bytes in: still there)
          if (out + done - flushed >= SN_FLUSH) flush((out + done) & ~15u);
        }
      }
      out += llen;
      pos = data + llen;
      continue;
    }
    // output positions: a wave prefix sum of the lengths at the element starts (each at most 64: no overflow)
    const bool start = (M >> lane) & 1ull;
    const uint32_t incl = wave_inclusive_sum(start ? len : 0u);
    const uint32_t opos = out + incl - (start ? len : 0u);
    const uint32_t o_end = out + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    // ---- the batch: elements at the lanes of M, input [pos, pos + cur), output [out, o_end) ----
    const bool copy = start && kind != 0;
    if (__builtin_amdgcn_ballot_w64(start && (bad || opos + len > dlen || (MODE == SN_FULL && kind != 0 && (off == 0 || off > opos))))) { failed = true; break; }
    if (MODE == SN_INDEX) {
      if (index && start && (opos & 0xffffu) == 0) index[1 + (opos >> 16)] = pos + (uint32_t)lane;
      if (__builtin_amdgcn_ballot_w64(start && ((opos ^ (opos + len - 1)) >> 16) != 0)) aligned = false;
    } else {
      // 1. a later element of the batch writes ring slots up to o_end: what lies 64 KiB below that is overwritten.  Copies
      //    that read from there (their source is older than the batch, so it is final) are moved first, in element order
      {
        uint64_t fm = __builtin_amdgcn_ballot_w64(copy && off > SN_RING - (o_end - opos));
        while (fm) {
          const int e = __builtin_ctzll(fm);
          fm &= fm - 1;
          wave_copy((uint32_t)__builtin_amdgcn_readlane((int)opos, e), (uint32_t)__builtin_amdgcn_readlane((int)off, e),
                    (uint32_t)__builtin_amdgcn_readlane((int)len, e));
        }
      }
      const bool near = copy && off <= SN_RING - (o_end - opos);
      // 2. literal bytes: input byte p belongs to the last element starting at or below it
      {
        const uint64_t below = M & ((2ull << lane) - 1ull);          // (bit 0 of M is set: the batch starts at lane 0)
        const int e = 63 - __builtin_clzll(below);
        const uint32_t e_opos = (uint32_t)__shfl((int)opos, e, 64);
        const uint32_t e_hk = (uint32_t)__shfl((int)(hdr | (kind << 4)), e, 64);
        const uint32_t first = (uint32_t)e + (e_hk & 15u);         // the element's first data byte
        if ((uint32_t)lane < cur && (e_hk >> 4) == 0 && (uint32_t)lane >= first) s_ring[(e_opos + (uint32_t)lane - first) & SN_MASK] = (uint8_t)tag;
      }
      // 3. the copies, one lane per element
      {
        uint64_t un = __builtin_amdgcn_ballot_w64(near);
        const uint32_t src0 = opos - off;
        const uint32_t src_end = src0 + (off < len ? off : len);     // its distinct source bytes end here
        while (un) {
          const uint32_t F = (uint32_t)__builtin_amdgcn_readlane((int)opos, __builtin_ctzll(un));   // below F the output is final
          const bool ready = ((un >> lane) & 1ull) && src_end <= F;  // (the first unfinished copy always is)
          un &= ~__builtin_amdgcn_ballot_w64(ready);
          uint32_t r = 0;                                            // i mod off, kept incrementally
          // four bytes per lane and pass (most copies are 4 to 8 bytes long).  The reads are not predicated: any
          // address inside the ring may be read, and a lane that is not ready simply drops what it got
          for (uint32_t j = 0; __builtin_amdgcn_ballot_w64(ready && j < len); j += 4) {
            uint8_t v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
              v[u] = s_ring[(src0 + r) & SN_MASK];
              r = r + 1 == off ? 0 : r + 1;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
              if (ready && j + u < len) s_ring[(opos + j + u) & SN_MASK] = v[u];
          }
        }
      }
    }
    pos += cur;
    out = o_end;
    if (MODE == SN_FULL) {
      if (!got4 && out >= 4) { first4 = *(const uint32_t*)s_ring; got4 = true; }   // (a batch makes at most 4 KiB: still there)
      if (out - flushed >= SN_FLUSH) flush(out & ~15u);
    }
  }
  if (span) { span->end_pos = pos; span->end_out = out; }
  if (!failed && (!span || span->whole) && (out != dlen || pos != slen)) failed = true;
  if (MODE == SN_FULL && !failed) flush(out);
  return failed;
}

}  // namespace

// The walks of a launch (one wave each): only the 4 KiB input window in LDS, so these waves do not compete with the BLOCK
// jobs of other pages for the CUs' LDS (two 64 KiB rings fill a CU).
//   INDEX    the whole chain of a page, from its first element.
//   SEG      a page of 64 KiB and more of compressed bytes (a walk takes ~13 ms per MB of them) is cut into up to
//            PQ_SNAPPY_SEGMENTS equal ranges of at least 32 KiB of input and every range is walked by its own wave.  Where the chain enters a range is not known, so wave w
//            GUESSES: it starts 2 KiB in front of its range at an arbitrary byte; a chain started at a wrong byte reads
//            literal bytes as headers but, with element starts a few bytes apart, falls onto a true start within a few
//            elements and is the true chain from there on.  The wave records where it entered its range (g), where it left
//            it (e) and how many bytes of output the elements in between make (L).
//   RESOLVE  the guesses are ACCEPTED only if they prove each other: segment 0 starts at the true beginning, so its exit is
//            true; g of segment 1 must equal it EXACTLY, which makes segment 1's exit true, and so on.  Then the output
//            position at which segment w begins is the sum of the L in front of it and the wave walks its range again, this
//            time noting the block starts.  If any link fails (a guess that never met the chain: long literals, a misread
//            extended length), segment 0's wave walks the whole page as an INDEX job would.  Nothing is assumed: a wrong
//            guess costs time, not correctness.
__global__ __launch_bounds__(64) void pq_inflate_index_kernel(const PqCodecParams p) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[SN_WIN + 32];
  const int lane = threadIdx.x;
  const PqCodecJob job = p.jobs[blockIdx.x];
  const uint8_t* src = job.raw + job.src_at;
  const uint32_t slen = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.src_len), dlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.dst_len);
  const uint32_t codec = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.codec);
  const uint32_t nblk = (dlen + 65535u) / 65536u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(job.block & 0xffffu));        // this wave's segment ...
  const uint32_t n_seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(job.block >> 16)) | (codec == PQ_CODEC_SNAPPY_INDEX ? 1u : 0u);   // ... of so many
  uint32_t* seg = job.index + nblk + 2;        // [n_seg][4]: g, e, L, ok
  const uint32_t b0 = (uint32_t)((uint64_t)slen * w / n_seg);
  const uint32_t b1 = w + 1 >= n_seg ? slen : (uint32_t)((uint64_t)slen * (w + 1) / n_seg);
  bool aligned = true;
  uint32_t first4 = 0;
  SnappySpan sp{};
  SnappySpan* use = nullptr;                   // (null: the whole page)
  uint32_t* table = job.index;
  uint32_t limit = dlen;
  bool preamble = true, guessing = false;
  if (codec == PQ_CODEC_SNAPPY_SEG) {
    table = nullptr; use = &sp;
    if (w == 0) { sp.start = 0; sp.stop = b1; }
    else { sp.start = b0 - PQ_SNAPPY_LEAD; sp.stop = b0; sp.lead = true; guessing = true; preamble = false; limit = 0x7fffffffu; }
  } else if (codec == PQ_CODEC_SNAPPY_RESOLVE) {
    bool proven = true;
    uint32_t before = 0, mine = 0, prev_e = 0;
    for (uint32_t v = 0; v < n_seg; ++v) {
      const uint32_t g = seg[4 * v], e = seg[4 * v + 1], l = seg[4 * v + 2], ok = seg[4 * v + 3];
      if (!ok || (v > 0 && g != prev_e)) proven = false;
      if (v < w) before += l;
      if (v == w) mine = g;
      prev_e = e;
    }
    proven = __builtin_amdgcn_readfirstlane((int)proven) != 0;
    if (proven) { use = &sp; sp.start = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine); sp.stop = b1; sp.out0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)before); sp.whole = w + 1 == n_seg; preamble = false; }
    else if (w != 0) return;
  }
  bool failed = false, ok = true;
  uint32_t g = 0;
  for (int tries = 0;;) {
    failed = snappy_stream<SN_INDEX>(src, slen, nullptr, limit, preamble, table, aligned, first4, nullptr, s_win, lane, use);
    if (!guessing) break;
    if (!failed) {                             // the guess reached its range: now the range itself
      g = sp.end_pos; guessing = false;
      sp = SnappySpan{}; sp.start = g; sp.stop = b1;
      continue;
    }
    if (++tries >= 8 || sp.end_pos + 1 >= b0) { ok = false; break; }
    sp.start = sp.end_pos + 1;                 // behind whatever did not look like an element
  }
  if (codec == PQ_CODEC_SNAPPY_SEG) {
    if (w == 0) g = sp.first;
    if (lane == 0) { seg[4 * w] = g; seg[4 * w + 1] = sp.end_pos; seg[4 * w + 2] = sp.end_out; seg[4 * w + 3] = ok && !failed ? 1u : 0u; }
    return;
  }
  if (job.flags & PQ_JOB_FORCE_FALLBACK) aligned = false;
  if (lane == 0) {
    if (failed || !aligned) job.index[0] = 1u;                        // the page is inflated by its FINISH job
    else if (!use || sp.whole) job.index[1 + nblk] = slen;           // (the end of the last block)
    if (failed) codec_error(job.err);
  }
}

// One wave per job.  job.raw: the chunk as it lies in the file; job.image: the uncompressed image (both padded by 64 bytes).
__global__ __launch_bounds__(64) void pq_inflate_kernel(const PqCodecParams p) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[SN_RING];
  __shared__ __attribute__((aligned(16))) uint8_t s_win[SN_WIN + 32];
  const int lane = threadIdx.x;
  const PqCodecJob job = p.jobs[blockIdx.x];
  const uint8_t* src = job.raw + job.src_at;
  uint8_t* dst = job.image + job.dst_at;        // (dst_at is a multiple of 16)
  uint32_t slen = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.src_len), dlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.dst_len);
  const uint32_t codec = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.codec);
  bool failed = false, aligned = true;
  uint32_t first4 = 0;

  if (codec == PQ_CODEC_STORED) {               // a page (or the level bytes of a V2 page) that lies uncompressed in the file
    if (slen != dlen) failed = true;
    else {
      for (uint32_t i = lane * 16u; i + 16 <= dlen; i += 64 * 16) { uint4 w; __builtin_memcpy(&w, src + i, 16); *(uint4*)(dst + i) = w; }
      for (uint32_t i = (dlen & ~15u) + lane; i < dlen; i += 64) dst[i] = src[i];
      if (dlen >= 4) { __builtin_memcpy(&first4, src, 4); }
    }
  } else if (codec == PQ_CODEC_SNAPPY || codec == PQ_CODEC_SNAPPY_BLOCK || codec == PQ_CODEC_SNAPPY_FINISH) {
    bool run = true, preamble = true;
    if (codec != PQ_CODEC_SNAPPY) {
      const uint32_t given_up = __hip_atomic_load(job.index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (codec == PQ_CODEC_SNAPPY_BLOCK) {
        if (given_up) return;
        const uint32_t k = job.block;
        const uint32_t s0 = job.index[1 + k], s1 = job.index[2 + k];
        if (s0 > s1 || s1 > slen || (uint64_t)k * 65536u >= dlen) { if (lane == 0) __hip_atomic_store(job.index, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        src += s0; slen = (uint32_t)__builtin_amdgcn_readfirstlane((int)(s1 - s0));
        dst += (size_t)k * 65536u; dlen = dlen - k * 65536u < 65536u ? dlen - k * 65536u : 65536u;
        preamble = false;
      } else if (!given_up) {                   // every block was inflated by its own wave: only the page descriptor is left
        run = false;
        if (dlen >= 4) first4 = __hip_atomic_load((const uint32_t*)dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (run) failed = snappy_stream<SN_FULL>(src, slen, dst, dlen, preamble, nullptr, aligned, first4, s_ring, s_win, lane);
    if (codec == PQ_CODEC_SNAPPY_BLOCK) {       // (an offset that reaches below the block is not damage: the page is redone whole)
      if (failed && lane == 0) __hip_atomic_store(job.index, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
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

hipError_t pq_launch_inflate_index(const PqCodecParams& p, hipStream_t s) {
  if (p.n_jobs <= 0) return hipSuccess;
  hipLaunchKernelGGL(pq_inflate_index_kernel, dim3(p.n_jobs), dim3(64), 0, s, p);
  return hipGetLastError();
}

}  // namespace chq
