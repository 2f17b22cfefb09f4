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
};
