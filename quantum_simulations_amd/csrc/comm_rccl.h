// comm_rccl.h -- multi-GPU reach of the C ABI: RCCL (xGMI) exchange inside the library.
// Part of the single translation unit qsim_hip.hip (included there, in order; not a standalone header).
//
// A reference maintainer who binds only libqsim_hip.so gets the partitioned state too (SURVEY 8b:
// `qsim_apply_1q_pair(h0, h1, U)` "RCCL inside", `qsim_swap_global_local` across GPUs): one process per
// GPU creates a `qsim_comm` from a 128-byte unique id that rank 0 makes and the host program hands to the
// other ranks (any transport: MPI, a file, torch.distributed); after that
//   qsim_comm_exchange      grouped send / receive of slices with any set of peers (one RCCL group),
//   qsim_comm_relayout      the all-to-all re-layout of m local <-> m global qubits of this rank's shard
//                           (pack all slabs in one pass, exchange, unpack; pipelined in pieces),
//   qsim_apply_1q_pair_remote / qsim_apply_2q_pair_*_remote
//                           the reference's partner-chunk butterflies (cpu_nonlocal.py:22-58) with the
//                           partner chunk on ANOTHER rank: full-shard exchange + the pair kernel.
// RCCL is resolved with dlopen at the first qsim_comm_* call (no link-time dependency: a process that has
// torch's bundled librccl loaded shares it; a plain C host gets /opt/rocm/lib/librccl.so.1).
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A host without the RCCL development headers still builds the library (RCCL is only ever dlopen'ed): the few
// types and constants of the NCCL API that the calls below use.
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
}
#endif
static_assert(ncclDouble == 8 && ncclSuccess == 0 && NCCL_UNIQUE_ID_BYTES == 128, "NCCL ABI constants");

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static int rccl_load() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_rccl.lib) return QSIM_OK;
  void* lib = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  if (!lib) return fail(QSIM_ERR_HIP, "RCCL is not available: dlopen(librccl.so.1) failed: %s", dlerror());
  RcclApi api;
  api.lib = lib;
  bool ok = true;
  auto sym = [&](const char* name) { void* p = dlsym(lib, name); ok = ok && p; return p; };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.Send = (decltype(api.Send))sym("ncclSend");
  api.Recv = (decltype(api.Recv))sym("ncclRecv");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { dlclose(lib); return fail(QSIM_ERR_HIP, "RCCL library lacks a required symbol"); }
  g_rccl = api;
  return QSIM_OK;
}

#define RCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess) return fail(QSIM_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
  } while (0)

struct qsim_comm {
  ncclComm_t comm;
  int rank, world, device;
  hipStream_t xfer_stream;     // re-layout pieces travel here while the next piece is packed on the chunk's stream
  hipEvent_t ev[16];           // piece s: ev[2s] = packed, ev[2s+1] = received
  hipEvent_t ev_bg[16];        // background exchanges (qsim_comm_exchange_bg): ticket t -> ev_bg[t % 16], recorded behind its group
  uint32_t bg_posted;          // tickets handed out so far
};

static int check_comm(const qsim_comm* c, const char* what) {
  if (!c || !c->comm) return fail(QSIM_ERR_INVALID, "%s: null communicator", what);
  return QSIM_OK;
}

// grouped exchange of [off, off + count) amplitude slices with each peer, on `stream`.  Ordering: the transfer is
// queued on `stream` only -- the caller must have BOTH buffers' producers / consumers on that stream (qsim_comm_exchange
// checks that the send and the receive chunk share it).
static int comm_exchange(qsim_comm* cm, int n_peers, const int32_t* peers, const double2* send, const uint64_t* send_off,
                         double2* recv, const uint64_t* recv_off, uint64_t count, hipStream_t stream) {
  RCCL_TRY(g_rccl.GroupStart());
  // (an error inside the group must not leave it open: every later RCCL call of the process would queue into it)
  ncclResult_t bad = ncclSuccess;
  const char* what = "";
  for (int i = 0; i < n_peers && bad == ncclSuccess; ++i) {
    bad = g_rccl.Send(send + send_off[i], 2 * count, ncclDouble, peers[i], cm->comm, stream);
    what = "ncclSend";
    if (bad == ncclSuccess) { bad = g_rccl.Recv(recv + recv_off[i], 2 * count, ncclDouble, peers[i], cm->comm, stream); what = "ncclRecv"; }
  }
  const ncclResult_t end = g_rccl.GroupEnd();
  if (bad != ncclSuccess) return fail(QSIM_ERR_HIP, "%s failed: %s", what, g_rccl.GetErrorString(bad));
  if (end != ncclSuccess) return fail(QSIM_ERR_HIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(end));
  return QSIM_OK;
}
