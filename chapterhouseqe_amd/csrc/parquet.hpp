// parquet.hpp -- Parquet scan with the page decode on the GPU.
//
// Replaces, for the step in front of the filter path, what the reference does with the third-party `parquet` crate inside
// read_files_task.rs:233-282 (ParquetRecordBatchStreamBuilder ... with_batch_size(max_rows_per_batch)): here the file's
// bytes are parsed on the host only as far as the METADATA goes (footer, page headers: a few hundred bytes per MB), the
// column chunks are uploaded as they lie in the file, and the pages are decoded into Arrow buffers in HBM by the kernels
// of parquet.hip -- the batch the filter kernels consume never exists in host memory.
//
// Scope: flat schemas (no repetition), optional or required columns (definition level <= 1), physical types BOOLEAN,
// INT32, INT64, FLOAT, DOUBLE, BYTE_ARRAY (as Utf8); encodings PLAIN and RLE_DICTIONARY / PLAIN_DICTIONARY; data pages
// V1 and V2; codecs UNCOMPRESSED (what the reference's own writers produce: AsyncArrowWriter::try_new(.., None),
// create_sample_data.rs:222, materialize_files_task.rs:128-133) and SNAPPY (what the writers users have produce by default),
// inflated on the GPU (parquet_codec.hip).  Anything else: CHQ_ERR_NOT_SUPPORTED with the reason.
// Input: the whole file in host memory, or a range reader (the caller's storage reader fetches the footer and exactly the
// column chunks a call decodes: read_files_task.rs:233-250 reads through opendal ranges the same way); a call decodes the
// columns it is asked for (DEV_NOTES.md:123: column pruning).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace chq {

enum PqType : int { PQ_BOOLEAN = 0, PQ_INT32 = 1, PQ_INT64 = 2, PQ_INT96 = 3, PQ_FLOAT = 4, PQ_DOUBLE = 5, PQ_BYTE_ARRAY = 6, PQ_FIXED_LEN_BYTE_ARRAY = 7 };
enum PqEncoding : int { PQ_PLAIN = 0, PQ_PLAIN_DICTIONARY = 2, PQ_RLE = 3, PQ_BIT_PACKED = 4, PQ_DELTA_BINARY_PACKED = 5,
                        PQ_DELTA_LENGTH_BYTE_ARRAY = 6, PQ_DELTA_BYTE_ARRAY = 7, PQ_RLE_DICTIONARY = 8, PQ_BYTE_STREAM_SPLIT = 9 };
enum PqPageType : int { PQ_DATA_PAGE = 0, PQ_INDEX_PAGE = 1, PQ_DICTIONARY_PAGE = 2, PQ_DATA_PAGE_V2 = 3 };

struct PqColumnSchema {
  std::string name;
  int type = -1;            // PqType
  int type_length = 0;
  int repetition = 0;       // 0 required, 1 optional, 2 repeated
  int converted_type = -1;  // 0 = UTF8
  bool logical_string = false;
  bool logical_other = false;   // a logical type this scan does not map (decimal, timestamp, ...)
};

struct PqPage {
  int type = 0;                 // PqPageType
  int64_t header_at = 0;        // offset of the page header, relative to the first byte of its column chunk
  int64_t payload_at = 0;       // offset of the first byte behind the header, relative to the chunk
  int64_t compressed_size = 0, uncompressed_size = 0;
  int64_t num_values = 0;       // rows of a data page (flat schema), entries of a dictionary page
  int encoding = 0;             // of the values
  int def_encoding = PQ_RLE;
  int64_t num_nulls = -1;       // V2 only
  int64_t def_bytes = 0, rep_bytes = 0;   // V2: byte lengths of the level sections (uncompressed, in front of the values)
  bool v2_compressed = true;    // V2: is_compressed (the values section; the level sections never are)
};

struct PqColumnChunk {
  int type = -1;
  int codec = 0;
  int64_t num_values = 0;
  int64_t total_compressed_size = 0;
  int64_t data_page_offset = 0, dictionary_page_offset = -1;
  int64_t stat_null_count = -1;   // from the chunk's Statistics, -1 = not recorded
  std::vector<int> encodings;
  bool pages_parsed = false;    // a file opened over a range reader parses a chunk's page headers when the chunk is fetched
  std::vector<PqPage> pages;    // in file order, dictionary page (if any) first
  int64_t first_byte() const {   // (an empty chunk has a dictionary page and no data page: data_page_offset is 0 then)
    return dictionary_page_offset > 0 && (data_page_offset <= 0 || dictionary_page_offset < data_page_offset) ? dictionary_page_offset : data_page_offset;
  }
};

struct PqRowGroup {
  int64_t num_rows = 0;
  std::vector<PqColumnChunk> columns;
};

// 0 = `length` bytes of the file starting at `offset` were written to `dst`; anything else fails the call
typedef int (*PqReadRange)(void* user, int64_t offset, int64_t length, uint8_t* dst);

struct PqFile {
  const uint8_t* data = nullptr;   // borrowed: the whole file in host memory, or null for a file behind a range reader
  PqReadRange read = nullptr;      // range reader (data == null)
  void* read_user = nullptr;
  int64_t size = 0;
  int64_t num_rows = 0;
  std::string created_by;
  std::vector<PqColumnSchema> columns;   // leaves of a flat schema, in file order
  std::vector<PqRowGroup> row_groups;
};

// Footer + every page header of every column chunk.  Throws ChqError (INVALID_ARGUMENT for a malformed file,
// NOT_SUPPORTED for nested schemas).
PqFile parquet_open(const uint8_t* data, int64_t size);
// The same over a range reader: reads the tail of the file (footer), nothing else; page headers are parsed per chunk when a
// call fetches the chunk.
PqFile parquet_open_reader(int64_t size, PqReadRange read, void* user);
// Page headers of one column chunk whose `csize` bytes start at `chunk`: fills c.pages (offsets relative to the chunk)
void parquet_parse_pages(const uint8_t* chunk, int64_t csize, PqColumnChunk& c);
// "rows R row_groups G created_by ..." / "column <name> <physical> <required|optional> [utf8]" /
// "rg <i> rows <n>" / "chunk <col> values <n> codec <c> pages <p>" / "page <type> values <n> enc <e> bytes <b> header <h>" lines
std::string parquet_describe(const PqFile& f);

struct Context;
struct Batch;
struct Buffer;
// One row group decoded into a device-resident batch (parquet_scan.cpp + parquet.hip).
Batch parquet_read_row_group(Context& ctx, const PqFile& f, int row_group);
// Several consecutive row groups, one batch each: their decodes overlap, the host synchronises twice per call.
// `columns` (n_columns >= 0 entries, or null = every column): the columns to decode, in the order they are wanted -- only
// their chunks are fetched, uploaded and decoded.
std::vector<Batch> parquet_read_row_groups(Context& ctx, const PqFile& f, int first, int count, const int32_t* columns = nullptr,
                                           int n_columns = -1);

// One record batch -> one complete Parquet file (one row group) in host memory (parquet_write.cpp + parquet_write.hip).
struct ParquetImage {
  std::shared_ptr<Buffer> bytes;   // host memory
  int64_t len = 0;
};
ParquetImage record_to_parquet(Context& ctx, const Batch& rec);
// several batches of one schema -> one file, one row group per batch
ParquetImage records_to_parquet(Context& ctx, const std::vector<const Batch*>& recs);

}  // namespace chq
