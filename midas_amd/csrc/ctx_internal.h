// Internals of the C-ABI context, shared by the translation units that implement entry points.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

struct midas_snps_ctx {
  int device = -1;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  int64_t err_read = -1;
  int pad_rule = 0;       // MIDAS_SNPS_PAD_SPEC: what P does to the query position (midas_snps_set_pad_rule)
  int row_coder = 0;      // MIDAS_SNPS_ROWS_DEVICE: who formats and deflates a batch's rows (midas_snps_set_row_coder)
  int default_path = 0;   // MIDAS_SNPS_PATH_AUTO: what batches created on this context take (midas_snps_set_default_path)
  // midas_snps_batch_write_part may be called from several host threads at once (one table each): the device part of a
  // call -- kernel, copies through the staging ring below -- is taken one at a time, the file writes run side by side
  std::mutex device_mutex;
  hipDeviceProp_t prop;
  // pinned staging ring for device -> pageable host copies, allocated on first use and kept for the context's lifetime
  // (pinning and unpinning a quarter of a gigabyte per batch costs more than the copy it would speed up)
  static constexpr int kStageSlots = 2;
  static constexpr size_t kStageBytes = (size_t)32 << 20;
  void* stage[kStageSlots] = {nullptr, nullptr};
  hipEvent_t stage_ev[kStageSlots] = {nullptr, nullptr};
};
