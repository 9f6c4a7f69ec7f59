/*
 * Arrow C Data Interface + Arrow C Device Data Interface structure definitions
 * (https://arrow.apache.org/docs/format/CDataInterface.html and CDeviceDataInterface.html).
 * These are the ABI-stable public structs; arrow-rs (`arrow::ffi`, `arrow::ffi_stream`), Arrow C++ and
 * pyarrow all produce/consume exactly this layout, which is why the chq C ABI uses them for record
 * batches: the Rust operator shim exports its `RecordBatch` with `arrow::ffi::to_ffi` and imports the
 * result with `arrow::ffi::from_ffi` without copying.
 */
#ifndef CHQ_ARROW_ABI_H
#define CHQ_ARROW_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE

#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4

struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};

struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};

#endif /* ARROW_C_DATA_INTERFACE */

#ifndef ARROW_C_DEVICE_DATA_INTERFACE
#define ARROW_C_DEVICE_DATA_INTERFACE

typedef int32_t ArrowDeviceType;
#define ARROW_DEVICE_CPU 1
#define ARROW_DEVICE_CUDA 2
#define ARROW_DEVICE_CUDA_HOST 3
#define ARROW_DEVICE_OPENCL 4
#define ARROW_DEVICE_VULKAN 7
#define ARROW_DEVICE_METAL 8
#define ARROW_DEVICE_VPI 9
#define ARROW_DEVICE_ROCM 10
#define ARROW_DEVICE_ROCM_HOST 11

struct ArrowDeviceArray {
  struct ArrowArray array;
  int64_t device_id;
  ArrowDeviceType device_type;
  void* sync_event; /* hipEvent_t* for ARROW_DEVICE_ROCM, or NULL when the data is already visible */
  int64_t reserved[3];
};

#endif /* ARROW_C_DEVICE_DATA_INTERFACE */

#ifdef __cplusplus
}
#endif
#endif /* CHQ_ARROW_ABI_H */
