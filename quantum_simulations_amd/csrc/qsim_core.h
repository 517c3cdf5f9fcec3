// qsim_core.h -- error plumbing and the chunk handle.
// Part of the single translation unit qsim_hip.hip (included there, in order; not a standalone header).
// ------------------------------------------------------------------ error plumbing
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                 \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return fail(e_ == hipErrorOutOfMemory ? QSIM_ERR_NOMEM : QSIM_ERR_HIP,          \
                  "%s failed: %s", #expr, hipGetErrorString(e_));                     \
  } while (0)

// ------------------------------------------------------------------ chunk handle
struct qsim_chunk {
  int device;
  int k;                 // log2(amplitudes)
  double2* amp;          // device pointer
  hipStream_t stream;
  bool owns_memory;
  qsim_chunk* parent;    // for views (keeps nothing alive; caller orders destruction)
  hipEvent_t ev0, ev1;   // timing
  bool have_events;
  double* scratch;       // reduction workspace (lazily allocated, owned)
  int last_passes;       // HBM passes of the last qsim_apply_ops
  u64 span_bytes;        // size of the allocation the chunk lives in (cache-policy choice)
  struct PendingLast* pending;   // split form of qsim_apply_ops_io: the slab-storing pass, planned but not yet launched (owned)
  struct DeferredIo* deferred;   // an op list whose source arrives in pieces (qsim_ops_io::src_parts): planned, not yet launched (owned)
  bool own_in_chunk;     // the last qsim_apply_ops_io stored its own slab into THIS chunk (dst_own == src, one pass): qsim_apply_ops_io_own_slab
};

static const int kMaxDevices = 16;
static hipStream_t g_stream[kMaxDevices];
static bool g_stream_ready[kMaxDevices];
static std::mutex g_mu;

static int device_stream(int device, hipStream_t* out) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (device < 0 || device >= kMaxDevices) return fail(QSIM_ERR_INVALID, "device %d out of range", device);
  if (!g_stream_ready[device]) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&g_stream[device], hipStreamNonBlocking));
    g_stream_ready[device] = true;
  }
  *out = g_stream[device];
  return QSIM_OK;
}

static inline u64 amps(const qsim_chunk* c) { return 1ull << c->k; }
