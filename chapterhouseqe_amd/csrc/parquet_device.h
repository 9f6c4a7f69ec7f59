// parquet_device.h -- structs shared by parquet_scan.cpp (host) and parquet.hip (kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chq {

enum : uint32_t { PQ_ERR_LEVELS = 1, PQ_ERR_VALUES = 2, PQ_ERR_CODEC = 3 };   // *err after a decode: which part of some page was malformed

// One data page of a column chunk.  Offsets are relative to the chunk buffer (the chunk as it lies in the file).
struct PqPageDesc {
  uint32_t levels_at;    // definition levels (RLE / bit-packed hybrid, bit width 1), behind their length prefix
  uint32_t levels_len;   // 0: none (required column, or statistics say there are no nulls)
  uint32_t values_at;
  uint32_t values_len;
  uint32_t num_rows;     // values of the page including nulls (flat schema: = rows)
  uint32_t pad;
  int64_t first_row;     // of the page inside the row group
};

struct PqDecodeParams {
  const uint8_t* chunk;
  const PqPageDesc* pages;
  int32_t n_pages;
  int32_t width;              // bytes per value (fixed-width types)
  const int32_t* page_list;   // the pages a values kernel serves (indices into `pages`)
  uint32_t* nonnull;          // [n_pages] non-null values per page (written by the levels kernel, or uploaded)
  uint32_t* value_base;       // [n_pages] index of the page's first value among the chunk's non-null values
  uint32_t* total_values;     // out: non-null values of the chunk
  uint8_t* valid8;            // [rows] one byte per row
  int32_t* row_val;           // [rows] value index, -1 = null
  uint8_t* dense;             // fixed-width: [values] decoded values (BOOLEAN: one byte each)
  uint32_t* vsrc;             // BYTE_ARRAY: [values] position of the bytes in the chunk buffer
  uint32_t* vlen;             //             [values] length
  uint32_t dict_at, dict_len, dict_count;   // dictionary page payload
  uint32_t walk_dictionary;   // pq_ba_walk_kernel: walk the dictionary page instead of data pages
  uint32_t* dict_src;         // BYTE_ARRAY dictionary: [dict_count] position / length of every entry
  uint32_t* dict_len_out;
  uint32_t* err;
};

// ---- page decompression (parquet_codec.hip) ---------------------------------------------------------------------------
// A compressed chunk is inflated page by page into an uncompressed image of the chunk; the page descriptors point into that
// image.  One job = one byte range of the raw chunk -> its place in the image.
enum : uint32_t { PQ_CODEC_STORED = 0, PQ_CODEC_SNAPPY = 1,
                  // a snappy page of several 64 KiB blocks, inflated block by block (parquet_codec.hip): three jobs in three launches
                  PQ_CODEC_SNAPPY_INDEX = 2,    // walks the element chain, writes where every block starts in the input
                  PQ_CODEC_SNAPPY_BLOCK = 3,    // inflates block `block` (one per block of the page)
                  PQ_CODEC_SNAPPY_FINISH = 4,   // patches the page descriptor; inflates the whole page if the blocks gave up
                  // ... and the INDEX walk of a large page as two launches of one wave per SEGMENT of its input (`block` = segment)
                  PQ_CODEC_SNAPPY_SEG = 5,      // walks a segment from a guessed start; notes where it entered, left, and the output in between
                  PQ_CODEC_SNAPPY_RESOLVE = 6 };// if the segments' entries and exits chain up exactly: walks the segment again, noting block starts
constexpr uint32_t PQ_SNAPPY_SEGMENTS = 16;     // at most; a segment is at least 32 KiB of compressed bytes (`block` = segment | segments << 16)
constexpr uint32_t PQ_SNAPPY_LEAD = 2048;       // bytes in front of a segment from where its wave looks for the chain
enum : uint32_t { PQ_JOB_KEEP_LEVELS = 1,       // with `page`: the column's definition levels are decoded (else only skipped)
                  PQ_JOB_FORCE_FALLBACK = 2 };  // INDEX job: report "not block-aligned" (tests of the FINISH path)
struct PqCodecJob {
  const uint8_t* raw;         // the chunk as it lies in the file (in HBM)
  uint8_t* image;             // the chunk's uncompressed image
  PqPageDesc* pages;          // the chunk's page descriptors (in HBM)
  uint32_t* err;              // the column's error word
  uint32_t src_at, src_len;   // in `raw`
  uint32_t dst_at, dst_len;   // in `image`; dst_at is a multiple of 16, dst_len is what the page header promises
  uint32_t codec;             // PQ_CODEC_*
  int32_t page;               // >= 0: a V1 data page of an optional column -- [4-byte length][levels][values] can only be told
                              // apart after inflation: the kernel fills levels_at / levels_len / values_at / values_len of pages[page]
  uint32_t flags;             // PQ_JOB_*
  uint32_t block;             // SNAPPY_BLOCK: which 64 KiB block of the page
  uint32_t* index;            // SNAPPY_INDEX / _BLOCK / _FINISH: [0] != 0: the blocks gave up; [1 + k]: input position of block k;
                              // [1 + blocks]: the page's compressed length; [2 + blocks ...]: PQ_SNAPPY_SEGMENTS x {entered at, left at,
                              // output bytes, ok} of a segmented walk (zero-filled before the launches)
};
// A launch serves every compressed page of every column and row group of a call's wave (jobs carry their own buffers):
// a page is a serial chain, so the pages (and, after the index walk, their 64 KiB blocks) in flight are the parallelism --
// per-column launches on the columns' streams left the twenty slowest pages of a twenty-row-group file sharing the handful of
// hardware queues the streams map to.  Launch order on a stream: INDEX jobs, then BLOCK + whole-page jobs, then FINISH jobs.
struct PqCodecParams {
  const PqCodecJob* jobs;
  int32_t n_jobs;
  int32_t pad;
};
hipError_t pq_launch_inflate(const PqCodecParams& p, hipStream_t s);         // STORED / SNAPPY / SNAPPY_BLOCK / SNAPPY_FINISH jobs
hipError_t pq_launch_inflate_index(const PqCodecParams& p, hipStream_t s);   // SNAPPY_INDEX jobs

struct PqRowParams {
  int64_t n_rows;
  const int32_t* row_val;     // null: row r holds value r (no nulls)
  const uint8_t* dense;
  const uint32_t* vsrc;
  const uint32_t* vlen;
  const uint8_t* chunk;
  void* out;                  // gather: values; pack_bits: bitmap words; rowlen: int32 offsets [n_rows + 1]
  const void* offsets;        // utf8 copy: the finished offsets
  uint8_t* data_out;
  unsigned long long* block_sums;   // [n_blocks]
  int64_t n_blocks;
  unsigned long long* total_bytes;
};

hipError_t pq_launch_levels(const PqDecodeParams& p, hipStream_t s);
hipError_t pq_launch_page_scan(const PqDecodeParams& p, hipStream_t s);
hipError_t pq_launch_rowval(const PqDecodeParams& p, hipStream_t s);
hipError_t pq_launch_plain_copy(const PqDecodeParams& p, int n_list, hipStream_t s);
hipError_t pq_launch_dict_fixed(const PqDecodeParams& p, int n_list, hipStream_t s);
hipError_t pq_launch_bool(const PqDecodeParams& p, int n_list, hipStream_t s);
hipError_t pq_launch_bool_rle(const PqDecodeParams& p, int n_list, hipStream_t s);
hipError_t pq_launch_ba_walk(const PqDecodeParams& p, int n_list, hipStream_t s);
hipError_t pq_launch_dict_ba(const PqDecodeParams& p, int n_list, hipStream_t s);
hipError_t pq_launch_gather_fixed(const PqRowParams& p, int width, int grid, hipStream_t s);
hipError_t pq_launch_pack_bits(const PqRowParams& p, int grid, hipStream_t s);
hipError_t pq_launch_rowlen(const PqRowParams& p, hipStream_t s);
hipError_t pq_launch_utf8_copy(const PqRowParams& p, int grid, hipStream_t s);
constexpr int PQ_SCAN_ROWS_HOST = 4096;   // = PQ_SCAN_ROWS of parquet.hip

// ---- Parquet write (parquet_write.hip) -------------------------------------------------------------------------------
struct PwParams {
  int64_t n_rows;
  const uint8_t* validity;      // Arrow validity bitmap or null (all valid)
  int64_t bit_offset;           // of row 0 in `validity`
  const uint8_t* values;        // fixed-width: values at row 0; bits_to_bytes: the Boolean bitmap; bytes_to_bits: one byte per value
  int64_t value_bit_offset;     // bits_to_bytes: of row 0 in `values`
  const int32_t* offsets;       // Utf8: offsets at row 0 (null for fixed-width columns)
  const uint8_t* data;          // Utf8: bytes (absolute offsets)
  int32_t width;                // fixed-width: bytes per value
  int32_t pad;
  uint8_t* out;
  unsigned long long* block_sums;   // [n_blocks]
  int64_t n_blocks;                 // ceil(n_rows / PW_BLOCK_ROWS_HOST)
  unsigned long long* total_bytes;
};
// pw_stats_kernel: min / max of the non-null (and, for floats, non-NaN) values of a column, for the chunk's Statistics.
// Numbers travel as order-preserving int64 keys (floats: the IEEE totalOrder key); Utf8 as the ROW of the smallest /
// largest value in unsigned byte order.
enum PwStatsKind { PW_STATS_I32 = 0, PW_STATS_I64, PW_STATS_F32, PW_STATS_F64, PW_STATS_U8, PW_STATS_UTF8 };
struct PwStatsParams {
  int64_t n_rows;
  const uint8_t* validity; int64_t bit_offset;
  const uint8_t* values;        // fixed-width values at row 0 (PW_STATS_U8: one byte per row)
  const int32_t* offsets; const uint8_t* data;   // Utf8
  int32_t kind; int32_t pad;
  long long* out;               // [3]: min key / row, max key / row, number of values that took part; host-initialised
  long long* cand;              // Utf8: [2 * grid] per-block candidates
  int32_t n_cand; int32_t pad2; // Utf8, second launch: number of blocks of the first
};
hipError_t pw_launch_stats(const PwStatsParams& p, int grid, hipStream_t s);
// pw_assemble_kernel: the file body is put together in HBM -- page headers (uploaded as one blob), level bytes and value
// streams copied to their file offsets -- so that the image needs ONE device-to-host copy however many pages it has
struct PwPiece { unsigned long long dst; const uint8_t* src; unsigned long long len; };
struct PwAssembleParams { const PwPiece* pieces; uint8_t* image; unsigned long long seg; };   // seg: bytes per block (multiple of 16)
hipError_t pw_launch_assemble(const PwAssembleParams& p, int n_pieces, int n_segs, hipStream_t s);
constexpr int PW_BLOCK_ROWS_HOST = 4096;   // = PW_BLOCK_ROWS of parquet_write.hip
hipError_t pw_launch_scan(const PwParams& p, hipStream_t s);
hipError_t pw_launch_encode(const PwParams& p, hipStream_t s);
hipError_t pw_launch_bits_to_bytes(const PwParams& p, int grid, hipStream_t s);
hipError_t pw_launch_bytes_to_bits(const PwParams& p, int grid, hipStream_t s);
hipError_t pw_launch_shift_bits(const PwParams& p, int grid, hipStream_t s);

}  // namespace chq
