// The path's one exchange between ranks, on RCCL itself: an all-gather of the per-species summary rows over xGMI (and the
// all-to-all of the genes path), behind four C-ABI calls -- no torch, no process group, nothing imported: a rank of an N-GPU
// job is as quick to start as a single process.  The reference's counterpart is the pickled (species_id, aln_stats) a pool
// worker returns through a pipe (midas/run/snps.py:225-241, midas/utility.py:81-107).
//
// librccl.so is loaded at run time (dlopen): the library itself does not depend on it, a single-GPU run never touches it.
// The 128-byte ncclUniqueId travels however the caller likes (midas_amd/dist.py: a file in <outdir>/snps/temp, written by
// rank 0 under a per-launch name).  Buffers at the ABI are the caller's host memory; the collectives run on device staging
// buffers on the context's stream.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/midas_snps.h"
#include "ctx_internal.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};

std::mutex g_lock;
Rccl g_rccl;

void set_err(char* err256, const char* fmt, const char* a = "", const char* b = "") {
  if (err256) snprintf(err256, 256, fmt, a, b);
}

// nullptr when RCCL is loaded; else what went wrong (in err256)
bool load_rccl(char* err256) {
  std::lock_guard<std::mutex> g(g_lock);
  if (g_rccl.lib) return true;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) { set_err(err256, "librccl.so not found: %s", dlerror()); return false; }
  Rccl r;
  r.lib = h;
#define SYM(field, name)                                                       \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                \
  if (!r.field) { set_err(err256, "librccl.so has no %s", name); dlclose(h); return false; }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(AllGather, "ncclAllGather")
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(GetErrorString, "ncclGetErrorString")
  SYM(GetVersion, "ncclGetVersion")
#undef SYM
  g_rccl = r;
  return true;
}

}  // namespace

struct midas_comm {
  midas_snps_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int32_t rank = 0, world = 1;
  uint8_t* d_send = nullptr;
  uint8_t* d_recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;
};

namespace {

int32_t nccl_fail(ncclResult_t r, const char* what, char* err256) {
  set_err(err256, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return MIDAS_SNPS_ERR_HIP;
}
int32_t hip_fail(hipError_t e, const char* what, char* err256) {
  set_err(err256, "%s: %s", what, hipGetErrorString(e));
  (void)hipGetLastError();
  return e == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP;
}
#define C_HIP(call)                                                   \
  do {                                                                \
    hipError_t e__ = (call);                                          \
    if (e__ != hipSuccess) return hip_fail(e__, #call, err256);       \
  } while (0)
#define C_NCCL(call)                                                  \
  do {                                                                \
    ncclResult_t r__ = (call);                                        \
    if (r__ != ncclSuccess) return nccl_fail(r__, #call, err256);     \
  } while (0)

int32_t room(midas_comm* c, size_t send_bytes, size_t recv_bytes, char* err256) {
  if (send_bytes > c->send_cap) {
    (void)hipFree(c->d_send);
    c->d_send = nullptr; c->send_cap = 0;
    const size_t cap = (send_bytes + 4095) & ~(size_t)4095;
    C_HIP(hipMalloc(&c->d_send, cap));
    c->send_cap = cap;
  }
  if (recv_bytes > c->recv_cap) {
    (void)hipFree(c->d_recv);
    c->d_recv = nullptr; c->recv_cap = 0;
    const size_t cap = (recv_bytes + 4095) & ~(size_t)4095;
    C_HIP(hipMalloc(&c->d_recv, cap));
    c->recv_cap = cap;
  }
  return MIDAS_SNPS_OK;
}

}  // namespace

extern "C" {

int32_t midas_comm_unique_id(uint8_t* out_id128, char* err256) {
  static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
  if (!out_id128) return MIDAS_SNPS_ERR_INVALID_ARG;
  if (!load_rccl(err256)) return MIDAS_SNPS_ERR_UNSUPPORTED;
  ncclUniqueId id;
  C_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(out_id128, &id, sizeof id);
  return MIDAS_SNPS_OK;
}

// Can this process use RCCL at all (the library loads, every entry point the binding needs is there, it answers)?  The ranks
// ask this BEFORE they call midas_comm_create together: a rank that could not would leave the others inside ncclCommInitRank,
// which waits for every rank of the communicator.
int32_t midas_comm_probe(int32_t* out_version, char* err256) {
  if (!load_rccl(err256)) return MIDAS_SNPS_ERR_UNSUPPORTED;
  int v = 0;
  C_NCCL(g_rccl.GetVersion(&v));
  if (out_version) *out_version = v;
  return MIDAS_SNPS_OK;
}

int32_t midas_comm_device_key(midas_snps_ctx* ctx, char* out64) {
  if (!ctx || !out64) return MIDAS_SNPS_ERR_INVALID_ARG;
  char bus[32] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof bus, ctx->device) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof bus, "device%d", ctx->device); }
  snprintf(out64, 64, "%s", bus);
  return MIDAS_SNPS_OK;
}

int32_t midas_comm_create(midas_snps_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world, midas_comm** out, char* err256) {
  if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return MIDAS_SNPS_ERR_INVALID_ARG;
  *out = nullptr;
  if (!load_rccl(err256)) return MIDAS_SNPS_ERR_UNSUPPORTED;
  C_HIP(hipSetDevice(ctx->device));
  midas_comm* c = new (std::nothrow) midas_comm();
  if (!c) { set_err(err256, "host allocation failed"); return MIDAS_SNPS_ERR_OUT_OF_MEMORY; }
  c->ctx = ctx; c->rank = rank; c->world = world;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return nccl_fail(r, "ncclCommInitRank", err256);
  }
  *out = c;
  return MIDAS_SNPS_OK;
}

void midas_comm_destroy(midas_comm* c) {
  if (!c) return;
  if (c->ctx) (void)hipSetDevice(c->ctx->device);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  (void)hipFree(c->d_send);
  (void)hipFree(c->d_recv);
  delete c;
}

int32_t midas_comm_all_gather(midas_comm* c, const void* send, void* recv, int64_t bytes_per_rank, char* err256) {
  if (!c || bytes_per_rank < 0 || (bytes_per_rank > 0 && (!send || !recv))) return MIDAS_SNPS_ERR_INVALID_ARG;
  if (bytes_per_rank == 0) return MIDAS_SNPS_OK;
  C_HIP(hipSetDevice(c->ctx->device));
  hipStream_t s = c->ctx->stream;
  const size_t n = (size_t)bytes_per_rank;
  int32_t st = room(c, n, n * (size_t)c->world, err256);
  if (st != MIDAS_SNPS_OK) return st;
  C_HIP(hipMemcpyAsync(c->d_send, send, n, hipMemcpyHostToDevice, s));
  C_NCCL(g_rccl.AllGather(c->d_send, c->d_recv, n, ncclUint8, c->comm, s));
  C_HIP(hipMemcpyAsync(recv, c->d_recv, n * (size_t)c->world, hipMemcpyDeviceToHost, s));
  C_HIP(hipStreamSynchronize(s));
  return MIDAS_SNPS_OK;
}

int32_t midas_comm_all_to_all_v(midas_comm* c, const void* send, const int64_t* send_bytes, void* recv, const int64_t* recv_bytes, char* err256) {
  if (!c || !send_bytes || !recv_bytes) return MIDAS_SNPS_ERR_INVALID_ARG;
  size_t ns = 0, nr = 0;
  for (int r = 0; r < c->world; ++r) {
    if (send_bytes[r] < 0 || recv_bytes[r] < 0) return MIDAS_SNPS_ERR_INVALID_ARG;
    ns += (size_t)send_bytes[r];
    nr += (size_t)recv_bytes[r];
  }
  if ((ns && !send) || (nr && !recv)) return MIDAS_SNPS_ERR_INVALID_ARG;
  C_HIP(hipSetDevice(c->ctx->device));
  hipStream_t s = c->ctx->stream;
  int32_t st = room(c, ns ? ns : 1, nr ? nr : 1, err256);
  if (st != MIDAS_SNPS_OK) return st;
  if (ns) C_HIP(hipMemcpyAsync(c->d_send, send, ns, hipMemcpyHostToDevice, s));
  C_NCCL(g_rccl.GroupStart());
  // (an open group is closed on EVERY way out: a thread left inside ncclGroupStart makes the communicator useless for what
  // follows -- the first failure is what is reported)
  size_t so = 0, ro = 0;
  ncclResult_t first = ncclSuccess;
  const char* where = "";
  for (int r = 0; r < c->world && first == ncclSuccess; ++r) {
    if (send_bytes[r]) { first = g_rccl.Send(c->d_send + so, (size_t)send_bytes[r], ncclUint8, r, c->comm, s); where = "ncclSend"; }
    if (first == ncclSuccess && recv_bytes[r]) { first = g_rccl.Recv(c->d_recv + ro, (size_t)recv_bytes[r], ncclUint8, r, c->comm, s); where = "ncclRecv"; }
    so += (size_t)send_bytes[r];
    ro += (size_t)recv_bytes[r];
  }
  const ncclResult_t ended = g_rccl.GroupEnd();
  if (first != ncclSuccess) return nccl_fail(first, where, err256);
  if (ended != ncclSuccess) return nccl_fail(ended, "ncclGroupEnd", err256);
  if (nr) C_HIP(hipMemcpyAsync(recv, c->d_recv, nr, hipMemcpyDeviceToHost, s));
  C_HIP(hipStreamSynchronize(s));
  return MIDAS_SNPS_OK;
}

}  // extern "C"
