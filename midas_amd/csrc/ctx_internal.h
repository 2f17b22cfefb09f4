// Internals of the C-ABI context, shared by the translation units that implement entry points.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// The context's device arena: ONE allocation that the BAM decodes of a context take turns in (a hipMalloc / hipFree pair
// costs ~17 ms a gigabyte here, and a decode wants a few times the BAM's inflated size).  A decode borrows it (take), a BAM
// handle whose payload columns were cut into it keeps it until it is closed (give) -- whoever asks meanwhile gets an
// allocation of its own.  Shared with its borrowers: the context may be destroyed before the last handle is closed.
struct midas_arena_pool {
  std::mutex m;
  int device = -1;
  void* p = nullptr;
  size_t bytes = 0;
  bool lent = false, closing = false;
  void* take(size_t need, bool* pooled) {       // nullptr: out of device memory
    std::lock_guard<std::mutex> g(m);
    if (!lent && !closing) {
      if (bytes < need) {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        // (a little more than asked: the next BAM of about this size fits too)
        const size_t want = need + need / 16;
        if (hipMalloc(&p, want) == hipSuccess) bytes = want;
        else { (void)hipGetLastError(); p = nullptr; if (hipMalloc(&p, need) == hipSuccess) bytes = need; else { (void)hipGetLastError(); p = nullptr; } }
      }
      if (p) { lent = true; *pooled = true; return p; }
      return nullptr;
    }
    void* q = nullptr;
    *pooled = false;
    if (hipMalloc(&q, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return q;
  }
  void give(void* q) {
    std::lock_guard<std::mutex> g(m);
    if (q && q == p) {
      lent = false;
      if (closing) { (void)hipFree(p); p = nullptr; bytes = 0; }
    } else if (q) {
      (void)hipFree(q);
    }
  }
  // Columns a device decode has copied down and that still lie in a lent arena as well (midas_bam_load_device: the small
  // columns of a BAM whose payload stays on the device): a batch made from those very host arrays -- read-only views of the
  // decoder's buffers -- copies them device to device instead of sending them up the link again.
  struct Twin { const uint8_t* host; const uint8_t* dev; size_t bytes; const void* owner; };
  std::vector<Twin> twins;
  void add_twin(const void* host, const void* dev, size_t bytes, const void* owner) {
    if (!host || !dev || !bytes) return;
    std::lock_guard<std::mutex> g(m);
    twins.push_back(Twin{static_cast<const uint8_t*>(host), static_cast<const uint8_t*>(dev), bytes, owner});
  }
  void drop_twins(const void* owner) {
    std::lock_guard<std::mutex> g(m);
    size_t k = 0;
    for (size_t i = 0; i < twins.size(); ++i)
      if (twins[i].owner != owner) twins[k++] = twins[i];
    twins.resize(k);
  }
  const void* find_twin(const void* host, size_t bytes) {        // the device's copy of host[0, bytes), or nullptr
    const uint8_t* h = static_cast<const uint8_t*>(host);
    std::lock_guard<std::mutex> g(m);
    for (const Twin& t : twins)
      if (h >= t.host && bytes <= t.bytes && (size_t)(h - t.host) <= t.bytes - bytes) return t.dev + (h - t.host);
    return nullptr;
  }
  void close() {                                  // the context goes: free now, or when the borrower gives it back
    std::lock_guard<std::mutex> g(m);
    closing = true;
    if (!lent && p) { (void)hipFree(p); p = nullptr; bytes = 0; }
  }
};

struct midas_snps_ctx {
  int device = -1;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;             // last error text: written and read under err_mutex only (set_error / error_text):
                               // the table writers run on several host threads of one context
  std::mutex err_mutex;
  void set_error(const std::string& msg) { std::lock_guard<std::mutex> g(err_mutex); err = msg; }
  void clear_error() { std::lock_guard<std::mutex> g(err_mutex); err.clear(); }
  std::string error_text() { std::lock_guard<std::mutex> g(err_mutex); return err; }
  int64_t err_read = -1;
  int pad_rule = 0;       // MIDAS_SNPS_PAD_SPEC: what P does to the query position (midas_snps_set_pad_rule)
  int row_coder = 0;      // MIDAS_SNPS_ROWS_DEVICE: who formats and deflates a batch's rows (midas_snps_set_row_coder)
  int default_path = 0;   // MIDAS_SNPS_PATH_AUTO: what batches created on this context take (midas_snps_set_default_path)
  // midas_snps_batch_write_part may be called from several host threads at once (one table each): the device part of a
  // call -- kernel, copies through the staging ring below -- is taken one at a time, the file writes run side by side
  std::mutex device_mutex;
  hipDeviceProp_t prop;
  std::shared_ptr<midas_arena_pool> arena = std::make_shared<midas_arena_pool>();
  // pinned staging ring for device -> pageable host copies, allocated on first use and kept for the context's lifetime
  // (pinning and unpinning a quarter of a gigabyte per batch costs more than the copy it would speed up)
  static constexpr int kStageSlots = 2;
  static constexpr size_t kStageBytes = (size_t)32 << 20;
  // (the ring has ONE user at a time -- copy_mutex: a decode's uploader, a fetch, and the table writers' copies, which run on
  // copy_stream beside the next table's row kernel on the context's stream; taken AFTER device_mutex by whoever holds both)
  std::mutex copy_mutex;
  hipStream_t copy_stream = nullptr;
  // device buffers of the table writers (one a table: its coded streams + the members' tables), kept between tables: a hipMalloc /
  // hipFree pair per table is 2-3 ms of which the free waits for whatever the device is running
  struct RowBuffers {
    std::mutex m;
    std::vector<std::pair<void*, size_t>> idle;
    void* take(size_t bytes, size_t* got) {
      {
        std::lock_guard<std::mutex> g(m);
        size_t best = idle.size();
        for (size_t k = 0; k < idle.size(); ++k)
          if (idle[k].second >= bytes && (best == idle.size() || idle[k].second < idle[best].second)) best = k;
        if (best < idle.size()) {
          void* p = idle[best].first;
          *got = idle[best].second;
          idle.erase(idle.begin() + (long)best);
          return p;
        }
      }
      void* p = nullptr;
      const size_t want = bytes + bytes / 8;       // (the next table of about this size fits too)
      if (hipMalloc(&p, want) != hipSuccess) {      // (out of memory: what lies idle goes, then exactly what was asked for)
        (void)hipGetLastError();
        clear();
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        *got = bytes;
        return p;
      }
      *got = want;
      return p;
    }
    void give(void* p, size_t bytes) {
      std::vector<void*> drop;
      {
        std::lock_guard<std::mutex> g(m);
        idle.emplace_back(p, bytes);
        while (idle.size() > 16) {        // (more than any number of writers: the smallest go)
          size_t least = 0;
          for (size_t k = 1; k < idle.size(); ++k)
            if (idle[k].second < idle[least].second) least = k;
          drop.push_back(idle[least].first);
          idle.erase(idle.begin() + (long)least);
        }
      }
      for (void* q : drop) (void)hipFree(q);
    }
    void clear() {
      std::lock_guard<std::mutex> g(m);
      for (auto& e : idle) (void)hipFree(e.first);
      idle.clear();
    }
  } row_buffers;
  void* stage[kStageSlots] = {nullptr, nullptr};
  hipEvent_t stage_ev[kStageSlots] = {nullptr, nullptr};
  // Page-locking the ring costs ~0.2 ms a megabyte -- 14 ms that the first BAM decode of a process used to pay in front of its
  // upload.  midas_snps_create starts it on a thread of its own (the caller goes on to read its species and contigs); whoever
  // needs the ring first waits for that thread (stage_join), and allocates by itself what it did not get.
  std::thread stage_thread;
  void stage_join() { if (stage_thread.joinable()) stage_thread.join(); }
};
