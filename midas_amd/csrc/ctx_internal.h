// Internals of the C-ABI context, shared by the translation units that implement entry points.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

struct midas_snps_ctx {
  int device = -1;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  int64_t err_read = -1;
  hipDeviceProp_t prop;
  // host staging of batch_create (packed records / payload), kept between batches: allocating, pinning or
  // unmapping a quarter of a gigabyte per batch costs more than packing and copying it
  void* stage_rec = nullptr;
  void* stage_blob = nullptr;
  size_t stage_rec_cap = 0, stage_blob_cap = 0;
};
