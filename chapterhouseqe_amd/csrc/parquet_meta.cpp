// parquet_meta.cpp -- host half of the Parquet scan: the footer (FileMetaData) and every page header, read with a small
// Thrift compact-protocol reader (the image has no thrift / parquet library; the format is parquet-format's
// parquet.thrift, which the reference reaches through the `parquet` crate).  No GPU code here: tests/test_parquet.py
// checks this half on the CPU tier against pyarrow's own metadata.
#include "parquet.hpp"

#include <algorithm>
#include <cstring>

#include "engine.hpp"

namespace chq {
namespace {

[[noreturn]] void bad(const std::string& what) { throw ChqError{CHQ_ERR_ARROW_INVALID_ARGUMENT, "parquet: " + what}; }

// Thrift compact protocol, read-only, bounds-checked
struct Thrift {
  const uint8_t* p; const uint8_t* end;
  enum { T_STOP = 0, T_TRUE = 1, T_FALSE = 2, T_I8 = 3, T_I16 = 4, T_I32 = 5, T_I64 = 6, T_DOUBLE = 7, T_BINARY = 8, T_LIST = 9, T_SET = 10, T_MAP = 11, T_STRUCT = 12 };

  uint8_t byte() { if (p >= end) bad("metadata runs past its end"); return *p++; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) { const uint8_t b = byte(); v |= (uint64_t)(b & 0x7f) << shift; if (!(b & 0x80)) return v; }
    bad("varint longer than 64 bits");
  }
  int64_t zigzag() { const uint64_t v = varint(); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
  std::string binary() {
    const uint64_t n = varint();
    if (n > (uint64_t)(end - p)) bad("string runs past the metadata");
    std::string s((const char*)p, (size_t)n); p += n; return s;
  }
  // next field of the current struct: false at STOP.  `last` carries the previous field id (ids are delta-coded).
  bool field(int& id, int& type, int& last) {
    const uint8_t h = byte();
    if (h == 0) return false;
    type = h & 0x0f;
    const int delta = h >> 4;
    id = delta ? last + delta : (int)zigzag();
    last = id;
    return true;
  }
  void list_header(int& elem_type, uint64_t& n) {
    const uint8_t h = byte();
    elem_type = h & 0x0f; n = h >> 4;
    if (n == 15) n = varint();
    if (n > (uint64_t)(end - p) + 1) bad("list longer than the metadata");
  }
  void skip(int type, int depth = 0) {
    if (depth > 32) bad("metadata nested too deeply");
    switch (type) {
      case T_TRUE: case T_FALSE: break;
      case T_I8: byte(); break;
      case T_I16: case T_I32: case T_I64: varint(); break;
      case T_DOUBLE: if (end - p < 8) bad("truncated double"); p += 8; break;
      case T_BINARY: { const uint64_t n = varint(); if (n > (uint64_t)(end - p)) bad("truncated binary"); p += n; } break;
      case T_LIST: case T_SET: {
        int et; uint64_t n; list_header(et, n);
        for (uint64_t i = 0; i < n; ++i) { if (et == T_TRUE || et == T_FALSE) byte(); else skip(et, depth + 1); }
      } break;
      case T_MAP: {
        const uint64_t n = varint();
        if (n) { const uint8_t kv = byte(); for (uint64_t i = 0; i < n; ++i) { skip(kv >> 4, depth + 1); skip(kv & 0x0f, depth + 1); } }
      } break;
      case T_STRUCT: { int id, t, last = 0; while (field(id, t, last)) skip(t, depth + 1); } break;
      default: bad("unknown thrift type " + std::to_string(type));
    }
  }
};

struct RawSchemaElement { PqColumnSchema col; int num_children = 0; bool has_type = false; };

RawSchemaElement read_schema_element(Thrift& t) {
  RawSchemaElement e;
  int id, ty, last = 0;
  while (t.field(id, ty, last)) {
    switch (id) {
      case 1: e.col.type = (int)t.zigzag(); e.has_type = true; break;
      case 2: e.col.type_length = (int)t.zigzag(); break;
      case 3: e.col.repetition = (int)t.zigzag(); break;
      case 4: e.col.name = t.binary(); break;
      case 5: e.num_children = (int)t.zigzag(); break;
      case 6: e.col.converted_type = (int)t.zigzag(); break;
      case 10: {   // LogicalType union: 1 STRING, 10 INTEGER{bitWidth, isSigned}, anything else is not mapped
        int lid, lty, llast = 0;
        while (t.field(lid, lty, llast)) {
          if (lid == 1) { e.col.logical_string = true; t.skip(lty); }
          else if (lid == 10) {
            int iid, ity, ilast = 0; int width = 0; bool is_signed = true;
            while (t.field(iid, ity, ilast)) {
              if (iid == 1) width = (int8_t)t.byte();
              else if (iid == 2) is_signed = ity == Thrift::T_TRUE;
              else t.skip(ity);
            }
            e.col.logical_other |= !is_signed || (width != 32 && width != 64);
          } else { e.col.logical_other = true; t.skip(lty); }
        }
      } break;
      default: t.skip(ty);
    }
  }
  return e;
}

PqColumnChunk read_column_chunk(Thrift& t) {
  PqColumnChunk c;
  int id, ty, last = 0;
  bool have_meta = false;
  while (t.field(id, ty, last)) {
    if (id != 3) { t.skip(ty); continue; }   // ColumnMetaData
    have_meta = true;
    int mid, mty, mlast = 0;
    while (t.field(mid, mty, mlast)) {
      switch (mid) {
        case 1: c.type = (int)t.zigzag(); break;
        case 2: { int et; uint64_t n; t.list_header(et, n); for (uint64_t i = 0; i < n; ++i) c.encodings.push_back((int)t.zigzag()); } break;
        case 4: c.codec = (int)t.zigzag(); break;
        case 5: c.num_values = t.zigzag(); break;
        case 7: c.total_compressed_size = t.zigzag(); break;
        case 9: c.data_page_offset = t.zigzag(); break;
        case 11: c.dictionary_page_offset = t.zigzag(); break;
        case 12: {   // Statistics: only null_count (field 3) is used -- "no nulls" lets the scan skip the definition levels
          int sid, sty, slast = 0;
          while (t.field(sid, sty, slast)) { if (sid == 3) c.stat_null_count = t.zigzag(); else t.skip(sty); }
        } break;
        default: t.skip(mty);
      }
    }
  }
  if (!have_meta) bad("column chunk without inline metadata");
  return c;
}

// `file` .. `file + size`: the bytes of ONE column chunk; `at`: offset of the page header inside it
PqPage read_page_header(const uint8_t* file, int64_t size, int64_t at) {
  if (at < 0 || at >= size) bad("page offset outside its column chunk");
  Thrift t{file + at, file + size};
  PqPage pg;
  pg.header_at = at;
  int id, ty, last = 0;
  bool typed = false;
  while (t.field(id, ty, last)) {
    switch (id) {
      case 1: pg.type = (int)t.zigzag(); typed = true; break;
      case 2: pg.uncompressed_size = t.zigzag(); break;
      case 3: pg.compressed_size = t.zigzag(); break;
      case 5: {   // DataPageHeader
        int hid, hty, hlast = 0;
        while (t.field(hid, hty, hlast)) {
          if (hid == 1) pg.num_values = t.zigzag();
          else if (hid == 2) pg.encoding = (int)t.zigzag();
          else if (hid == 3) pg.def_encoding = (int)t.zigzag();
          else t.skip(hty);
        }
      } break;
      case 7: {   // DictionaryPageHeader
        int hid, hty, hlast = 0;
        while (t.field(hid, hty, hlast)) {
          if (hid == 1) pg.num_values = t.zigzag();
          else if (hid == 2) pg.encoding = (int)t.zigzag();
          else t.skip(hty);
        }
      } break;
      case 8: {   // DataPageHeaderV2
        int hid, hty, hlast = 0;
        while (t.field(hid, hty, hlast)) {
          switch (hid) {
            case 1: pg.num_values = t.zigzag(); break;
            case 2: pg.num_nulls = t.zigzag(); break;
            case 4: pg.encoding = (int)t.zigzag(); break;
            case 5: pg.def_bytes = t.zigzag(); break;
            case 6: pg.rep_bytes = t.zigzag(); break;
            case 7: pg.v2_compressed = hty == Thrift::T_TRUE; break;
            default: t.skip(hty);
          }
        }
      } break;
      default: t.skip(ty);
    }
  }
  if (!typed) bad("page header without a type");
  pg.payload_at = (int64_t)(t.p - file);
  // sizes come from an untrusted file: compare by subtraction (a sum with a size near INT64_MAX wraps and passes)
  if (pg.compressed_size < 0 || pg.compressed_size > size - pg.payload_at) bad("page payload runs past its column chunk");
  if (pg.uncompressed_size < 0 || pg.uncompressed_size >= (1ll << 31)) bad("page of " + std::to_string(pg.uncompressed_size) + " bytes");
  if (pg.num_values < 0 || pg.num_values >= (1ll << 31)) bad("page with " + std::to_string(pg.num_values) + " values");
  if (pg.def_bytes < 0 || pg.rep_bytes < 0) bad("negative level section length");
  return pg;
}

const char* type_name(int t) {
  static const char* n[] = {"BOOLEAN", "INT32", "INT64", "INT96", "FLOAT", "DOUBLE", "BYTE_ARRAY", "FIXED_LEN_BYTE_ARRAY"};
  return t >= 0 && t < 8 ? n[t] : "?";
}

}  // namespace

namespace {
// FileMetaData (the footer) -> schema, row groups, column chunk metadata; no page headers
void read_footer(const uint8_t* meta, int64_t meta_len, int64_t file_size, PqFile& f) {
  Thrift t{meta, meta + meta_len};
  std::vector<RawSchemaElement> schema;
  int id, ty, last = 0;
  while (t.field(id, ty, last)) {
    switch (id) {
      case 2: { int et; uint64_t n; t.list_header(et, n); for (uint64_t i = 0; i < n; ++i) schema.push_back(read_schema_element(t)); } break;
      case 3: f.num_rows = t.zigzag(); break;
      case 4: {
        int et; uint64_t n; t.list_header(et, n);
        for (uint64_t i = 0; i < n; ++i) {
          PqRowGroup rg;
          int rid, rty, rlast = 0;
          while (t.field(rid, rty, rlast)) {
            if (rid == 1) { int cet; uint64_t cn; t.list_header(cet, cn); for (uint64_t k = 0; k < cn; ++k) rg.columns.push_back(read_column_chunk(t)); }
            else if (rid == 3) rg.num_rows = t.zigzag();
            else t.skip(rty);
          }
          f.row_groups.push_back(std::move(rg));
        }
      } break;
      case 6: f.created_by = t.binary(); break;
      default: t.skip(ty);
    }
  }
  if (schema.empty()) bad("no schema");
  if (schema[0].num_children != (int)schema.size() - 1)
    throw ChqError{CHQ_ERR_NOT_SUPPORTED, "parquet: nested schema (groups below the root) is not supported"};
  for (size_t i = 1; i < schema.size(); ++i) {
    if (schema[i].num_children > 0 || !schema[i].has_type) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "parquet: nested schema (group '" + schema[i].col.name + "')"};
    f.columns.push_back(schema[i].col);
  }
  for (PqRowGroup& rg : f.row_groups) {
    if (rg.num_rows < 0) bad("row group with a negative row count");
    if (rg.columns.size() != f.columns.size()) bad("row group with " + std::to_string(rg.columns.size()) + " column chunks for " + std::to_string(f.columns.size()) + " columns");
    for (PqColumnChunk& c : rg.columns) {
      if (c.num_values < 0) bad("column chunk with a negative value count");
      if (c.num_values == 0) continue;   // an empty chunk: writers leave its offsets at 0
      // sizes come from an untrusted file: compare by subtraction (a sum with a size near INT64_MAX wraps and passes)
      const int64_t at = c.first_byte();
      if (at < 4 || c.total_compressed_size < 0 || c.total_compressed_size > file_size - 8 - at) bad("column chunk outside the file");
    }
  }
}
}  // namespace

void parquet_parse_pages(const uint8_t* chunk, int64_t csize, PqColumnChunk& c) {
  c.pages.clear();
  int64_t at = 0, values = 0;
  while (at < csize && values < c.num_values) {
    PqPage pg = read_page_header(chunk, csize, at);   // (a page may not run past its chunk)
    at = pg.payload_at + pg.compressed_size;
    if (pg.type == PQ_DATA_PAGE || pg.type == PQ_DATA_PAGE_V2) {
      if (pg.num_values > c.num_values - values) bad("page with more values than its chunk has left");
      values += pg.num_values;
    }
    c.pages.push_back(pg);
  }
  if (values != c.num_values) bad("pages hold " + std::to_string(values) + " values, the chunk's metadata says " + std::to_string(c.num_values));
  c.pages_parsed = true;
}

PqFile parquet_open(const uint8_t* data, int64_t size) {
  if (!data || size < 12) bad("file shorter than the magic numbers");
  if (memcmp(data, "PAR1", 4) != 0 || memcmp(data + size - 4, "PAR1", 4) != 0) {
    if (memcmp(data + size - 4, "PARE", 4) == 0) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "parquet: encrypted footer"};
    bad("magic number PAR1 missing");
  }
  uint32_t flen; memcpy(&flen, data + size - 8, 4);
  if ((int64_t)flen + 12 > size) bad("footer length larger than the file");
  PqFile f;
  f.data = data; f.size = size;
  read_footer(data + size - 8 - flen, (int64_t)flen, size, f);
  for (PqRowGroup& rg : f.row_groups)
    for (PqColumnChunk& c : rg.columns)
      if (c.num_values > 0) parquet_parse_pages(data + c.first_byte(), c.total_compressed_size, c);
      else c.pages_parsed = true;
  return f;
}

PqFile parquet_open_reader(int64_t size, PqReadRange read, void* user) {
  if (!read || size < 12) bad("file shorter than the magic numbers");
  // one read for the tail: the footer is almost always inside the last 64 KiB; a longer one costs a second read
  std::vector<uint8_t> tail((size_t)std::min<int64_t>(size, 64 << 10));
  if (read(user, size - (int64_t)tail.size(), (int64_t)tail.size(), tail.data()) != 0) bad("the range reader failed on the file tail");
  const uint8_t* e = tail.data() + tail.size();
  if (memcmp(e - 4, "PAR1", 4) != 0) {
    if (memcmp(e - 4, "PARE", 4) == 0) throw ChqError{CHQ_ERR_NOT_SUPPORTED, "parquet: encrypted footer"};
    bad("magic number PAR1 missing");
  }
  uint32_t flen; memcpy(&flen, e - 8, 4);
  if ((int64_t)flen + 12 > size) bad("footer length larger than the file");
  std::vector<uint8_t> meta;
  const uint8_t* mp;
  if ((size_t)flen + 8 <= tail.size()) mp = e - 8 - flen;
  else {
    meta.resize(flen);
    if (read(user, size - 8 - (int64_t)flen, (int64_t)flen, meta.data()) != 0) bad("the range reader failed on the footer");
    mp = meta.data();
  }
  PqFile f;
  f.read = read; f.read_user = user; f.size = size;
  read_footer(mp, (int64_t)flen, size, f);
  for (PqRowGroup& rg : f.row_groups) for (PqColumnChunk& c : rg.columns) if (c.num_values == 0) c.pages_parsed = true;
  return f;
}

std::string parquet_describe(const PqFile& f) {
  std::string o = "rows " + std::to_string(f.num_rows) + " row_groups " + std::to_string(f.row_groups.size()) + " columns " + std::to_string(f.columns.size()) + "\n";
  for (const PqColumnSchema& c : f.columns)
    o += "column " + c.name + " " + type_name(c.type) + (c.repetition == 0 ? " required" : c.repetition == 1 ? " optional" : " repeated") +
         ((c.logical_string || c.converted_type == 0) ? " utf8" : "") + (c.logical_other ? " logical-other" : "") + "\n";
  for (size_t g = 0; g < f.row_groups.size(); ++g) {
    const PqRowGroup& rg = f.row_groups[g];
    o += "rg " + std::to_string(g) + " rows " + std::to_string(rg.num_rows) + "\n";
    for (size_t k = 0; k < rg.columns.size(); ++k) {
      const PqColumnChunk& c = rg.columns[k];
      o += "chunk " + std::to_string(k) + " values " + std::to_string(c.num_values) + " codec " + std::to_string(c.codec) + " pages " + std::to_string(c.pages.size()) + "\n";
      for (const PqPage& pg : c.pages)
        o += "page " + std::to_string(pg.type) + " values " + std::to_string(pg.num_values) + " enc " + std::to_string(pg.encoding) + " bytes " + std::to_string(pg.compressed_size) +
             " header " + std::to_string(pg.payload_at - pg.header_at) + "\n";
    }
  }
  return o;
}

}  // namespace chq
