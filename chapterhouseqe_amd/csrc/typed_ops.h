// typed_ops.h -- parameter blocks of the small kernels of typed_ops.hip that are not part of the device program
// interface (device_program.h, whose hash ties the committed PMC summaries to the kernels they were measured on).
#pragma once
#include <stdint.h>

namespace chq {
// Uniform-length Utf8 columns (keys, hashes, the reference's sample strings): utf8_uniform_kernel checks that every value
// of a column has the length of the first one; out = {1 if some value differs, that length, offsets[0]}.
// iota_offsets_kernel writes the offsets of n values of L bytes: 0, L, 2 L, ...
struct Utf8UniformParams { const int32_t* offsets; int64_t nrows; int32_t* out; };
struct IotaOffsetsParams { int32_t* out; int64_t n_plus_1; int32_t step; int32_t pad; };
// the same check for every batch of a group: block b walks batch b's offsets; out[3 b ..] = {differs, length, offsets[0]}
struct Utf8UniformGroupParams { const unsigned long long* offsets_of; const long long* rows_of; int64_t nb; int32_t* out; };
}  // namespace chq
