// libqsim_hip.so -- gate-application kernels for MI355X (gfx950 / CDNA4), C ABI in
// include/qsim_hip.h.  Written for wave64, 16-byte (complex128) lane accesses and the
// HBM roofline: every gate is a streaming read-modify-write of the amplitudes it touches.
//
// Kernel family (one template, `k_gate<NM, ITEMS>`):
//   a gate is reduced on the host to
//     * a sorted list of index bit positions that are *removed* from the work-item
//       index (target bits and fixed-one control/diagonal bits),
//     * NM = 1, 2 or 4 "member" base pointers (the NM amplitudes one work item owns:
//       target-bit combinations, with fixed-one bits and partner-chunk selection folded
//       into the pointer),
//     * an NM x NM complex matrix.
//   NM=1: x *= d           diagonal Z/S/T/R (half the state), CZ/CR (a quarter)
//   NM=2: 2x2 butterfly    dense 1q, controlled-1q (CNOT/CY/CU: half), SWAP (half),
//                          apply_1q_pair across two chunks
//   NM=4: 4x4 butterfly    dense 2q, partner-chunk pair/quad forms
//   Each amplitude belongs to exactly one work item, so the update is in place with
//   no inter-thread hazard.  Lanes own consecutive work items => a wave's 16-B loads
//   cover contiguous runs of 2^(lowest removed bit) amplitudes (1 KiB when that bit >= 6).
//
// Algorithmic HBM bytes per launch: 32 * NM * count (read + write of every member).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/qsim_hip.h"

typedef unsigned long long u64;

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

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cfma(double2 a, double2 b, double2 acc) {
  // acc + a*b
  return make_double2(fma(a.x, b.x, fma(-a.y, b.y, acc.x)), fma(a.x, b.y, fma(a.y, b.x, acc.y)));
}

// Streaming (non-temporal) 16-byte accesses: `global_load/store_dwordx4 ... nt`.  Measured on
// MI355X (profiles/r01_bw_probe.txt): +6..15 % on the in-place butterfly when every wave
// instruction covers whole 128-B lines; harmful when a line is shared by two instructions,
// so the launcher only selects NT when the lowest removed index bit is >= 3.
template <bool NT>
__device__ __forceinline__ double2 ld_amp(const double2* p) {
  if (NT) {
    double2 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
  }
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st_amp(double2* p, double2 v) {
  if (NT) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
  } else {
    *p = v;
  }
}

// Re-insert zero bits at ascending positions pos[0..npos) of a compressed index.
__device__ __forceinline__ u64 expand_index(u64 c, int npos, int p0, int p1, int p2) {
  if (npos > 0) c = ((c >> p0) << (p0 + 1)) | (c & ((1ull << p0) - 1));
  if (npos > 1) c = ((c >> p1) << (p1 + 1)) | (c & ((1ull << p1) - 1));
  if (npos > 2) c = ((c >> p2) << (p2 + 1)) | (c & ((1ull << p2) - 1));
  return c;
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one).  With SWZ
// each XCD walks one contiguous eighth of the work-item space instead of every eighth block.
// (2-D grids only carry block counts whose x extent would overflow the 2^32 work-item limit of one
// grid dimension -- 33-qubit chunks; the linear id is y * gridDim.x + x.)
template <bool SWZ>
__device__ __forceinline__ u64 logical_block() {
  const u64 bid = (u64)blockIdx.y * gridDim.x + blockIdx.x;
  if (SWZ) {
    const u64 per_xcd = ((u64)gridDim.x * gridDim.y) >> 3;
    return (bid & 7) * per_xcd + (bid >> 3);
  }
  return bid;
}

constexpr unsigned kMaxGridX = 1u << 22;   // x * 256 threads stays below 2^32 work-items
static dim3 grid_for(u64 blocks) {
  if (blocks <= kMaxGridX) return dim3((unsigned)blocks);
  return dim3(kMaxGridX, (unsigned)((blocks + kMaxGridX - 1) / kMaxGridX));
}

template <int NM>
struct GateArgs {
  double2* member[NM];  // base pointer of each member (offsets folded in)
  u64 count;            // work items
  int npos;
  int pos[3];
  double2 u[NM * NM];   // row-major NM x NM
};

constexpr int kBlock = 256;

// All target bits resolved in registers: a work item owns NM amplitudes.
template <int NM, int ITEMS, bool NT, bool SWZ>
__global__ __launch_bounds__(kBlock) void k_gate(const GateArgs<NM> a) {
  const u64 first = (logical_block<SWZ>() * ITEMS) * kBlock + threadIdx.x;
  u64 idx[ITEMS];
  bool live[ITEMS];
  double2 x[ITEMS][NM];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const u64 c = first + (u64)r * kBlock;
    live[r] = c < a.count;
    idx[r] = expand_index(c, a.npos, a.pos[0], a.pos[1], a.pos[2]);
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    if (live[r]) {
#pragma unroll
      for (int m = 0; m < NM; ++m) x[r][m] = ld_amp<NT>(a.member[m] + idx[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    if (live[r]) {
#pragma unroll
      for (int row = 0; row < NM; ++row) {
        double2 acc = cmul(a.u[row * NM], x[r][0]);
#pragma unroll
        for (int col = 1; col < NM; ++col) acc = cfma(a.u[row * NM + col], x[r][col], acc);
        st_amp<NT>(a.member[row] + idx[r], acc);
      }
    }
  }
}

// Low target bits (index bit < 3: partners share a 128-B line) are resolved across lanes:
// every lane loads its own amplitude(s) with whole-line coalescing, fetches the partner
// values with ds_bpermute (`__shfl_xor`) and computes only its own output row.
//   NMR register members x 2^NSH lane states; canonical matrix index = (r << NSH) | s,
//   s bit b <-> lane bit lane_bit[b].
template <int NMR, int NSH>
struct ShuffleArgs {
  double2* member[NMR];
  u64 count;
  int npos;
  int pos[3];
  int lane_bit[2];
  double2 u[(NMR << NSH) * (NMR << NSH)];
};

template <int NMR, int NSH, int ITEMS, bool NT>
__global__ __launch_bounds__(kBlock) void k_gate_shuffle(const ShuffleArgs<NMR, NSH> a) {
  constexpr int NL = 1 << NSH;
  constexpr int DIM = NMR << NSH;
  const int lane = threadIdx.x & 63;
  int s = 0;
#pragma unroll
  for (int b = 0; b < NSH; ++b) s |= ((lane >> a.lane_bit[b]) & 1) << b;
  // per-lane coefficients: coef[r][r2][d] = U[(r<<NSH)|s][(r2<<NSH)|(s^d)]
  double2 coef[NMR][NMR][NL];
  int xmask[NL];
#pragma unroll
  for (int d = 0; d < NL; ++d) {
    xmask[d] = 0;
#pragma unroll
    for (int b = 0; b < NSH; ++b) xmask[d] |= ((d >> b) & 1) << a.lane_bit[b];
#pragma unroll
    for (int r = 0; r < NMR; ++r)
#pragma unroll
      for (int r2 = 0; r2 < NMR; ++r2) {
        double2 c = a.u[((r << NSH) | 0) * DIM + ((r2 << NSH) | (0 ^ d))];
#pragma unroll
        for (int sv = 1; sv < NL; ++sv) {
          const double2 alt = a.u[((r << NSH) | sv) * DIM + ((r2 << NSH) | (sv ^ d))];
          c.x = (s == sv) ? alt.x : c.x;
          c.y = (s == sv) ? alt.y : c.y;
        }
        coef[r][r2][d] = c;
      }
  }
  const u64 first = (logical_block<false>() * ITEMS) * kBlock + threadIdx.x;
  u64 idx[ITEMS];
  bool live[ITEMS];
  double2 x[ITEMS][NMR];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const u64 c = first + (u64)it * kBlock;
    live[it] = c < a.count;
    idx[it] = expand_index(c, a.npos, a.pos[0], a.pos[1], a.pos[2]);
#pragma unroll
    for (int r = 0; r < NMR; ++r) x[it][r] = make_double2(0.0, 0.0);
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if (live[it]) {
#pragma unroll
      for (int r = 0; r < NMR; ++r) x[it][r] = ld_amp<NT>(a.member[r] + idx[it]);
    }
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    double2 out[NMR];
#pragma unroll
    for (int r = 0; r < NMR; ++r) out[r] = make_double2(0.0, 0.0);
#pragma unroll
    for (int r2 = 0; r2 < NMR; ++r2) {
#pragma unroll
      for (int d = 0; d < NL; ++d) {
        double2 v = x[it][r2];
        if (d != 0) {  // all 64 lanes take part (inactive tail lanes hold zeros)
          v.x = __shfl_xor(x[it][r2].x, xmask[d], 64);
          v.y = __shfl_xor(x[it][r2].y, xmask[d], 64);
        }
#pragma unroll
        for (int r = 0; r < NMR; ++r) out[r] = cfma(coef[r][r2][d], v, out[r]);
      }
    }
    if (live[it]) {
#pragma unroll
      for (int r = 0; r < NMR; ++r) st_amp<NT>(a.member[r] + idx[it], out[r]);
    }
  }
}

// ------------------------------------------------------------------ host: gate plan
struct Group {        // 1, 2 or 4 chunks forming a virtual index space of k + g bits
  qsim_chunk* c[4];
  int n;              // number of chunks (1, 2, 4)
  int k;              // local bits
};

struct Plan {
  int nm;                   // register members: 1, 2, 4
  int nsh;                  // lane-resolved targets: 0, 1, 2
  double2* member[4];
  u64 count;                // work items
  int npos;
  int pos[3];
  int lane_bit[2];
  double2 u[16];            // (nm << nsh)^2 canonical matrix
  int low_removed;          // lowest removed index bit (64 if none)
  int high_removed;         // highest removed index bit (-1 if none)
  bool resident;            // the whole state fits the Infinity Cache: plain (cacheable) accesses
};

static inline bool is_zero(double re, double im) { return re == 0.0 && im == 0.0; }
static inline bool is_one(double re, double im) { return re == 1.0 && im == 0.0; }

// Index bits below this are resolved across lanes (partners share one 128-B line).
constexpr int kLaneCut = 3;

// Tunables (environment overrides are for profiling sweeps only).
struct Tuning {
  int swz_cut = 64;   // XCD-contiguous block order when the highest removed bit is below this (r01 scan: always)
  int force_nt = -1;  // -1 auto, 0 never, 1 always
  int items = 0;      // 0 auto
  int max_gates_per_pass = 128;
  int tile_special = 1;      // real / Y-like / -1 / +-i special-case opcodes in fused passes
  int tile_merge_diag = 1;   // merge phase gates that share their predicate (OPC_DIAGR)
  int debug_skip_gates = 0;  // QSIM_DEBUG_SKIP_GATES=1: tile passes move data but apply nothing (WRONG results)
  int debug_stats = 0;       // QSIM_DEBUG_STATS=1: print gates / groups per pass to stderr
  int tile_persistent = 0;   // resident grid + next-tile prefetch
  int tile_wgs_per_cu = 8;   // upper bound for the persistent grid (the occupancy query decides)
  int num_cus = 256;
  // States up to this size stay in the 256 MiB Infinity Cache between launches when accessed with
  // the default cache policy (tools/mall_probe.hip: 8.5-8.8 TB/s r+w for a 128-256 MiB region vs
  // 5.5 streaming); the NT policy bypasses it (6.0-6.2 at every size), so NT is for larger states.
  u64 mall_bytes = 256ull << 20;
  Tuning() {
    if (const char* e = getenv("QSIM_MALL_BYTES")) mall_bytes = strtoull(e, nullptr, 10);
    if (const char* e = getenv("QSIM_DEBUG_STATS")) debug_stats = atoi(e);
    if (const char* e = getenv("QSIM_TILE_PERSIST")) tile_persistent = atoi(e);
    if (const char* e = getenv("QSIM_TILE_WGS")) tile_wgs_per_cu = std::max(1, atoi(e));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess && prop.multiProcessorCount > 0)
      num_cus = prop.multiProcessorCount;
    if (const char* e = getenv("QSIM_SWZ_CUT")) swz_cut = atoi(e);
    if (const char* e = getenv("QSIM_NT")) force_nt = atoi(e);
    if (const char* e = getenv("QSIM_ITEMS")) items = atoi(e);
    if (const char* e = getenv("QSIM_PASS_GATES")) max_gates_per_pass = std::max(1, atoi(e));
    if (const char* e = getenv("QSIM_TILE_SPECIAL")) tile_special = atoi(e);
    if (const char* e = getenv("QSIM_TILE_MERGE_DIAG")) tile_merge_diag = atoi(e);
    if (const char* e = getenv("QSIM_DEBUG_SKIP_GATES")) debug_skip_gates = atoi(e);
  }
};
static const Tuning& tuning() {
  static Tuning t;
  return t;
}

// ---- per-launch HIP-event timing (bench.py roofline): events are recorded on the launch
// stream around every gate kernel while a profile is open; nothing is synchronised until
// qsim_profile_end.
struct LaunchRecord {
  int cls;            // kernel class id
  double bytes;       // algorithmic bytes: sum over the launch's gate-applications
  double hbm_bytes;   // bytes the launch itself has to move (32 B per amplitude it touches)
  hipEvent_t e0, e1;
};
struct ProfileState {
  bool open = false;
  hipStream_t stream = nullptr;
  std::vector<LaunchRecord> records;
  std::vector<hipEvent_t> pool;  // recycled events
};
static ProfileState g_prof;
static const char* const kClassNames[] = {
    "k_gate<1> scale (diagonal subset)", "k_gate<2> 2x2 butterfly", "k_gate<4> 4x4 butterfly",
    "k_gate_shuffle<1,1> lane 1q", "k_gate_shuffle<1,2> lane 2q", "k_gate_shuffle<2,1> lane+reg 2q",
    "k_tile fused pass"};
constexpr int kNumClasses = 7;

static hipEvent_t prof_event() {
  if (!g_prof.pool.empty()) {
    hipEvent_t e = g_prof.pool.back();
    g_prof.pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

struct ProfileScope {  // RAII around one launch
  bool on;
  LaunchRecord rec;
  ProfileScope(int cls, double bytes, hipStream_t stream, double hbm_bytes = -1.0) {
    on = g_prof.open && g_prof.stream == stream;
    if (!on) return;
    rec.cls = cls;
    rec.bytes = bytes;
    rec.hbm_bytes = hbm_bytes < 0 ? bytes : hbm_bytes;
    rec.e0 = prof_event();
    rec.e1 = prof_event();
    (void)hipEventRecord(rec.e0, stream);
  }
  void done(hipStream_t stream) {
    if (!on) return;
    (void)hipEventRecord(rec.e1, stream);
    g_prof.records.push_back(rec);
  }
};

// Do the allocations behind a group fit the Infinity Cache?
static bool group_resident(const Group& g) {
  bool one_parent = g.c[0]->parent != nullptr;
  u64 bytes = 0;
  for (int i = 0; i < g.n; ++i) {
    bytes += g.c[i]->span_bytes;
    one_parent = one_parent && g.c[i]->parent == g.c[0]->parent;
  }
  if (one_parent) bytes = g.c[0]->span_bytes;
  return bytes <= tuning().mall_bytes;
}

// Resolve a virtual offset (bits >= k select the chunk) to a device pointer.
static double2* resolve(const Group& g, u64 voff) {
  const u64 ci = voff >> g.k;
  return g.c[ci]->amp + (voff & ((1ull << g.k) - 1));
}

// Build a plan: `targets` (matrix order, MSB first), `fixed` one-bits, 2^nt x 2^nt matrix.
static int make_plan(const Group& g, const int* targets, int nt, const int* fixed, int nf,
                     const double* mat, Plan* p) {
  bool lane_t[2] = {false, false};
  int nr_bits = 0, nsh = 0;
  for (int j = 0; j < nt; ++j) {
    lane_t[j] = targets[j] < g.k && targets[j] < kLaneCut;
    if (lane_t[j]) ++nsh; else ++nr_bits;
  }
  p->nm = 1 << nr_bits;
  p->nsh = nsh;
  p->resident = group_resident(g);
  int removed[4];
  int nr = 0;
  for (int j = 0; j < nt; ++j) if (!lane_t[j] && targets[j] < g.k) removed[nr++] = targets[j];
  for (int i = 0; i < nf; ++i) if (fixed[i] < g.k) removed[nr++] = fixed[i];
  if (nr > 3) return fail(QSIM_ERR_INVALID, "internal: more than 3 removed bits");
  std::sort(removed, removed + nr);
  p->npos = nr;
  for (int i = 0; i < 3; ++i) p->pos[i] = i < nr ? removed[i] : 0;
  p->low_removed = nr ? removed[0] : 64;
  p->high_removed = nr ? removed[nr - 1] : -1;
  p->count = 1ull << (g.k - nr);
  // rank of each target among register / lane targets, in matrix (MSB-first) order
  int reg_rank[2] = {0, 0}, lane_rank[2] = {0, 0};
  for (int j = 0, rr = 0, lr = 0; j < nt; ++j) {
    if (lane_t[j]) lane_rank[j] = lr++; else reg_rank[j] = rr++;
  }
  p->lane_bit[0] = p->lane_bit[1] = 0;
  for (int j = 0; j < nt; ++j) {
    if (!lane_t[j]) continue;
    int below = 0;
    for (int i = 0; i < nr; ++i) if (removed[i] < targets[j]) ++below;
    p->lane_bit[nsh - 1 - lane_rank[j]] = targets[j] - below;
  }
  u64 fixed_off = 0;
  for (int i = 0; i < nf; ++i) fixed_off |= 1ull << fixed[i];
  for (int r = 0; r < p->nm; ++r) {
    u64 off = fixed_off;
    for (int j = 0; j < nt; ++j)
      if (!lane_t[j] && ((r >> (nr_bits - 1 - reg_rank[j])) & 1)) off |= 1ull << targets[j];
    p->member[r] = resolve(g, off);
  }
  // canonical index of matrix index m: (r << nsh) | s
  const int dim = 1 << nt;
  int canon[4];
  for (int m = 0; m < dim; ++m) {
    int r = 0, s = 0;
    for (int j = 0; j < nt; ++j) {
      const int bit = (m >> (nt - 1 - j)) & 1;
      if (lane_t[j]) s |= bit << (nsh - 1 - lane_rank[j]);
      else r |= bit << (nr_bits - 1 - reg_rank[j]);
    }
    canon[m] = (r << nsh) | s;
  }
  for (int a = 0; a < dim; ++a)
    for (int b = 0; b < dim; ++b)
      p->u[canon[a] * dim + canon[b]] = make_double2(mat[2 * (a * dim + b)], mat[2 * (a * dim + b) + 1]);
  return QSIM_OK;
}

template <int NM, int ITEMS, bool NT, bool SWZ>
static int launch_reg(const Plan& p, hipStream_t stream) {
  GateArgs<NM> a;
  for (int m = 0; m < NM; ++m) a.member[m] = p.member[m];
  a.count = p.count;
  a.npos = p.npos;
  for (int i = 0; i < 3; ++i) a.pos[i] = p.pos[i];
  for (int i = 0; i < NM * NM; ++i) a.u[i] = p.u[i];
  const u64 per_block = (u64)kBlock * ITEMS;
  const u64 blocks = (p.count + per_block - 1) / per_block;
  ProfileScope prof(NM == 1 ? 0 : (NM == 2 ? 1 : 2), 32.0 * NM * (double)p.count, stream);
  hipLaunchKernelGGL((k_gate<NM, ITEMS, NT, SWZ>), grid_for(blocks), dim3(kBlock), 0, stream, a);
  prof.done(stream);
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

template <int NMR, int NSH, int ITEMS, bool NT>
static int launch_shuffle(const Plan& p, hipStream_t stream) {
  ShuffleArgs<NMR, NSH> a;
  for (int m = 0; m < NMR; ++m) a.member[m] = p.member[m];
  a.count = p.count;
  a.npos = p.npos;
  for (int i = 0; i < 3; ++i) a.pos[i] = p.pos[i];
  a.lane_bit[0] = p.lane_bit[0];
  a.lane_bit[1] = p.lane_bit[1];
  constexpr int DIM = NMR << NSH;
  for (int i = 0; i < DIM * DIM; ++i) a.u[i] = p.u[i];
  const u64 per_block = (u64)kBlock * ITEMS;
  const u64 blocks = (p.count + per_block - 1) / per_block;
  ProfileScope prof(NMR == 2 ? 5 : (NSH == 1 ? 3 : 4), 32.0 * NMR * (double)p.count, stream);
  hipLaunchKernelGGL((k_gate_shuffle<NMR, NSH, ITEMS, NT>), grid_for(blocks), dim3(kBlock), 0, stream, a);
  prof.done(stream);
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

template <int NM, int ITEMS>
static int launch_reg_flags(const Plan& p, bool nt, bool swz, hipStream_t stream) {
  if (nt) return swz ? launch_reg<NM, ITEMS, true, true>(p, stream) : launch_reg<NM, ITEMS, true, false>(p, stream);
  return swz ? launch_reg<NM, ITEMS, false, true>(p, stream) : launch_reg<NM, ITEMS, false, false>(p, stream);
}

template <int NM>
static int launch_reg_items(const Plan& p, int items, bool nt, bool swz, hipStream_t stream) {
  switch (items) {
    case 1: return launch_reg_flags<NM, 1>(p, nt, swz, stream);
    case 2: return launch_reg_flags<NM, 2>(p, nt, swz, stream);
    default: return launch_reg_flags<NM, 4>(p, nt, swz, stream);
  }
}

static int launch_plan(const Plan& p, hipStream_t stream) {
  const Tuning& t = tuning();
  // NT only when every wave instruction covers whole 128-B lines
  bool nt = p.low_removed >= 3 && !p.resident;
  if (t.force_nt >= 0) nt = t.force_nt != 0;
  if (p.count > (1ull << 40)) return fail(QSIM_ERR_INVALID, "grid too large");
  if (p.nsh > 0) {
    if (p.nm == 1 && p.nsh == 1) return nt ? launch_shuffle<1, 1, 2, true>(p, stream) : launch_shuffle<1, 1, 2, false>(p, stream);
    if (p.nm == 1 && p.nsh == 2) return nt ? launch_shuffle<1, 2, 2, true>(p, stream) : launch_shuffle<1, 2, 2, false>(p, stream);
    if (p.nm == 2 && p.nsh == 1) return nt ? launch_shuffle<2, 1, 2, true>(p, stream) : launch_shuffle<2, 1, 2, false>(p, stream);
    return fail(QSIM_ERR_INVALID, "internal: bad shuffle plan %d/%d", p.nm, p.nsh);
  }
  // work items per thread (profiles/r01e_tune_items_swz.txt): one 2- or 4-member item per thread
  // is best up to removed bit 19 (0.77 vs 0.75 of peak); above it two items even out the
  // q mod 4 pattern of the HBM address hash (0.70-0.77)
  int items = p.nm == 4 ? 1 : ((p.nm == 2 && p.high_removed < 20) ? 1 : 2);
  bool swz = p.high_removed < t.swz_cut;
  if (!swz) items *= 2;
  if (t.items > 0) items = t.items;
  if (items != 1 && items != 2) items = 4;
  const u64 per_block = (u64)kBlock * items;
  const u64 blocks = (p.count + per_block - 1) / per_block;
  if (blocks < 64 || (blocks & 7)) swz = false;
  switch (p.nm) {
    case 1: return launch_reg_items<1>(p, items, nt, swz, stream);
    case 2: return launch_reg_items<2>(p, items, nt, swz, stream);
    case 4: return launch_reg_items<4>(p, items, nt, swz, stream);
  }
  return fail(QSIM_ERR_INVALID, "internal: bad member count %d", p.nm);
}

// Classify + launch a 1-qubit gate on virtual qubit `q` of the group.
static int gate_1q(const Group& g, int q, const double* U, hipStream_t stream) {
  Plan p;
  int rc;
  const bool diag = is_zero(U[2], U[3]) && is_zero(U[4], U[5]);
  // A diagonal bit inside a 128-B line (q < kLaneCut) leaves no untouched lines: the subset form
  // would still move every line, with partial-line accesses; the dense lane form streams whole
  // lines non-temporally instead (6.3 -> 5.8 ms at n = 30).
  const bool subline = q < g.k && q < kLaneCut;
  if (diag && is_one(U[0], U[1]) && !subline) {
    if (is_one(U[6], U[7])) return QSIM_OK;  // identity
    rc = make_plan(g, nullptr, 0, &q, 1, U + 6, &p);  // scale the bit-set half by U11
  } else if (diag && is_one(U[0], U[1]) && is_one(U[6], U[7])) {
    return QSIM_OK;  // identity
  } else {
    rc = make_plan(g, &q, 1, nullptr, 0, U, &p);
  }
  if (rc) return rc;
  return launch_plan(p, stream);
}

// Classify + launch a 2-qubit gate on virtual qubits (qa = MSB, qb = LSB).
static int gate_2q(const Group& g, int qa, int qb, const double* U, hipStream_t stream) {
  auto z = [&](int r, int c) { return is_zero(U[2 * (4 * r + c)], U[2 * (4 * r + c) + 1]); };
  auto one = [&](int r, int c) { return is_one(U[2 * (4 * r + c)], U[2 * (4 * r + c) + 1]); };
  Plan p;
  int rc;
  bool offdiag_zero = true;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c)
      if (r != c && !z(r, c)) offdiag_zero = false;
  // controlled on qa: [[I, 0], [0, V]]
  const bool ctrl_a = one(0, 0) && one(1, 1) && z(0, 1) && z(1, 0) &&
                      z(0, 2) && z(0, 3) && z(1, 2) && z(1, 3) &&
                      z(2, 0) && z(2, 1) && z(3, 0) && z(3, 1);
  // controlled on qb: identity on indices {0, 2}, V on {1, 3}
  const bool ctrl_b = one(0, 0) && one(2, 2) && z(0, 2) && z(2, 0) &&
                      z(0, 1) && z(0, 3) && z(2, 1) && z(2, 3) &&
                      z(1, 0) && z(1, 2) && z(3, 0) && z(3, 2);
  const bool swap = one(0, 0) && one(3, 3) && one(1, 2) && one(2, 1) && z(1, 1) && z(2, 2) &&
                    z(0, 1) && z(0, 2) && z(0, 3) && z(1, 0) && z(1, 3) &&
                    z(2, 0) && z(2, 3) && z(3, 0) && z(3, 1) && z(3, 2);
  if (offdiag_zero && one(0, 0) && one(1, 1) && one(2, 2)) {
    if (one(3, 3)) return QSIM_OK;  // identity
    const int fixed[2] = {qa, qb};
    rc = make_plan(g, nullptr, 0, fixed, 2, U + 2 * 15, &p);  // CZ / CR: quarter of the state
  } else if (ctrl_a && !(qa < g.k && qa < kLaneCut)) {   // (a sub-line control saves no traffic: dense form below)
    const double V[8] = {U[2 * 10], U[2 * 10 + 1], U[2 * 11], U[2 * 11 + 1],
                         U[2 * 14], U[2 * 14 + 1], U[2 * 15], U[2 * 15 + 1]};
    rc = make_plan(g, &qb, 1, &qa, 1, V, &p);  // CNOT / CY / CU: half of the state
  } else if (ctrl_b && !(qb < g.k && qb < kLaneCut)) {
    const double V[8] = {U[2 * 5], U[2 * 5 + 1], U[2 * 7], U[2 * 7 + 1],
                         U[2 * 13], U[2 * 13 + 1], U[2 * 15], U[2 * 15 + 1]};
    rc = make_plan(g, &qa, 1, &qb, 1, V, &p);
  } else if (swap) {
    // exchange |01> <-> |10>: 2-member work items (a=0,b=1) and (a=1,b=0), half the state.
    static const double X2[8] = {0, 0, 1, 0, 1, 0, 0, 0};
    // removed = {qa, qb}; member0 = |a=0,b=1>, member1 = |a=1,b=0>
    int removed[2];
    int nr = 0;
    if (qa < g.k) removed[nr++] = qa;
    if (qb < g.k) removed[nr++] = qb;
    std::sort(removed, removed + nr);
    p.nm = 2;
    p.nsh = 0;
    p.resident = group_resident(g);
    p.lane_bit[0] = p.lane_bit[1] = 0;
    p.npos = nr;
    for (int i = 0; i < 3; ++i) p.pos[i] = i < nr ? removed[i] : 0;
    p.low_removed = nr ? removed[0] : 64;
    p.high_removed = nr ? removed[nr - 1] : -1;
    p.count = 1ull << (g.k - nr);
    p.member[0] = resolve(g, 1ull << qb);
    p.member[1] = resolve(g, 1ull << qa);
    for (int i = 0; i < 4; ++i) p.u[i] = make_double2(X2[2 * i], X2[2 * i + 1]);
    rc = QSIM_OK;
  } else {
    const int t[2] = {qa, qb};
    rc = make_plan(g, t, 2, nullptr, 0, U, &p);
  }
  if (rc) return rc;
  return launch_plan(p, stream);
}

// ================================================================== fused tile passes
// One HBM round trip applies MANY gates (the GPU form of the reference's level batching,
// wenbo_engine/circuit/fusion.py:86-142, and of v3's fused independent-gate block,
// parallel_gate_applicator.py:169-204): a workgroup loads a *tile* of 2^T amplitudes into LDS,
// applies every gate of the pass whose target bits are tile bits, and stores the tile back.
//   tile bits = the kTileLow lowest index bits (every global access is a whole 128-B line, NT)
//   + T - kTileLow arbitrary higher bits chosen by the gates of the pass.
//   Control bits and diagonal bits may lie OUTSIDE the tile: they become a per-tile predicate.
// Inside the tile gates are applied in *register groups*: a group owns kGroupBits tile bits;
// each thread pulls the 2^kGroupBits amplitudes that differ in those bits from LDS into eight
// NAMED registers, applies every gate of the group on them, and writes them back once -- LDS
// traffic is paid per group, not per gate.
// The gate loop is instruction-issue bound (rocprofv3: SALU ~ VALU, one scalar unit per CU), so
// the host pre-decodes every gate into ONE opcode byte selecting a straight-line case (every
// kind x register target x register control combination, plus special cases for real matrices,
// Y-like gates and -1 / +-i phases) and ready-made predicate masks.
// Algorithmic bytes per pass: 32 B x 2^k (every amplitude read and written once), for g gates.
#ifndef QSIM_TILE_LOW
#define QSIM_TILE_LOW 3
#endif
#ifndef QSIM_TILE_SWZ
#define QSIM_TILE_SWZ 0
#endif
constexpr int kTileLow = QSIM_TILE_LOW;
#ifndef QSIM_TILE_BITS_MAX
#define QSIM_TILE_BITS_MAX 11
#endif
constexpr int kTileBitsMax = QSIM_TILE_BITS_MAX;       // 2^11 amplitudes = 32 KiB of LDS: 4-5 workgroups per CU
constexpr int kGroupBits = 3;
#ifndef QSIM_TILE_THREADS
#define QSIM_TILE_THREADS 256
#endif
constexpr int kTileThreads = QSIM_TILE_THREADS;
constexpr int kTileThreadBits = kTileThreads == 64 ? 6 : kTileThreads == 128 ? 7 : kTileThreads == 256 ? 8 : kTileThreads == 512 ? 9 : 10;
constexpr int kTileMaxGates = 144;     // entries incl. group headers  (2304 B of kernel arguments)
constexpr int kTileMaxMat = 104;       // complex matrix pool          (1664 B)

enum : uint8_t {
  OPC_NOP = 0,
  OPC_DENSE1 = 1,     // +variant 0..8: general 2x2                           (pool: 4)
  OPC_SWAP1 = 10,     // +variant: a <-> b                  X, CNOT           (pool: 0)
  OPC_ANTI1 = 19,     // +variant: a' = u01 b, b' = u10 a                     (pool: 4)
  OPC_PHASE = 28,     // +register mask 0..7: x *= m[0]     T, R, CR          (pool: 1)
  OPC_DENSE2 = 36,    // +3*JA + JB: general 4x4, SWAP                        (pool: 16)
  OPC_REAL1 = 45,     // +variant: 2x2 with real entries    H, RY, G          (pool: 4)
  OPC_YLIKE1 = 54,    // +variant: [[0,-i],[i,0]]           Y, CY             (pool: 0)
  OPC_PHASE_NEG = 63, // +mask: x = -x                      Z, CZ             (pool: 0)
  OPC_PHASE_I = 71,   // +mask: x = i x                     S                 (pool: 0)
  OPC_PHASE_NI = 79,  // +mask: x = -i x                                      (pool: 0)
  OPC_DIAGR = 87,     // +{0: bits 0,1; 1: bits 0,2; 2: bits 1,2; 3: bits 0,1,2}: several phase gates
                      // that share their predicate, one per register bit, merged by the host:
                      // x_i *= prod of the listed bits' phases that are set in i   (pool: 2 or 3)
  OPC_GROUP = 0xFE    // group header
};
// 1q variant: 0..2 = target register bit J, no register control; 3 + 2*J + k = control on the
// k-th of the two other register bits (ascending)
static inline int opc_1q_variant(int J, int C) {
  return C < 0 ? J : 3 + 2 * J + ((C > J ? C - 1 : C) == 0 ? 0 : 1);
}

struct alignas(16) TileGate {   // 16 bytes: one s_load_dwordx4
  uint8_t opcode;
  uint8_t count;           // group header: entries in the group
  uint16_t blk_mask;       // gate: tile bits OUTSIDE the group that must be 1;
                           // group header: s0 | s1 << 4 | s2 << 8 (ascending tile bits)
  uint16_t mat;            // gate: first TileArgs::mat entry of its matrix (0 when it has none)
  uint16_t pad;            // 0: the device reads mat | pad << 16 as one dword
  uint64_t outer_mask;     // absolute index bits outside the tile that must be 1
};

// The device reads a descriptor as ONE 128-bit scalar load (a struct copy is split by the
// compiler into per-field loads that each wait for scalar memory: four round trips per gate).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static_assert(sizeof(TileGate) == sizeof(u32x4), "descriptor = one dwordx4");

struct TileArgs {
  double2* amp;
  int ngates;
  int pad;
  uint8_t h[16];           // ascending absolute positions of the tile's high bits
  u32x4 g[kTileMaxGates];  // TileGate images
  double2 mat[kTileMaxMat];
};
static_assert(sizeof(TileArgs) <= 4096, "kernel arguments must fit 4 KiB");

static inline void put_gate(TileArgs* a, int i, const TileGate& g) { std::memcpy(&a->g[i], &g, sizeof g); }
static inline TileGate get_gate(const TileArgs* a, int i) { TileGate g; std::memcpy(&g, &a->g[i], sizeof g); return g; }

__device__ __forceinline__ unsigned insert_zero(unsigned c, int p) {
  return ((c >> p) << (p + 1)) | (c & ((1u << p) - 1));
}
// XOR-swizzled LDS slot (measured: within 1 % of five other swizzles and of none -- bank
// conflicts are not what limits the gate phase)
__device__ __forceinline__ unsigned lds_slot(unsigned t) { return t ^ ((t >> 4) & 15u); }

// Every case of the gate switch must leave x0..x7 in the registers it found them in: when a case
// defines a new value while the old one is still live the register allocator gives the whole PHI
// web of the switch a second register set and EVERY gate pays 32 v_mov_b64 (2/3 of the VALU work of
// a pass, r01f ISA audit).  So the last instruction of each output is inline asm whose destination
// is tied ("+v") to the old register; partial sums live in ordinary temporaries.
#define QS_IP_FMA(D, S, V, C)  asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(D) : "s"(S), "v"(V), "v"(C))    /* D = S*V + C  */
#define QS_IP_FNMA(D, S, V, C) asm volatile("v_fma_f64 %0, -%1, %2, %3" : "+v"(D) : "s"(S), "v"(V), "v"(C))   /* D = -S*V + C */
#define QS_IP_MOV(D, V)        asm volatile("v_mov_b64 %0, %1" : "+v"(D) : "v"(V))                              /* D = V        */
#define QS_IP_NEG(D, V)        asm volatile("v_mul_f64 %0, -1.0, %1" : "+v"(D) : "v"(V))                        /* D = -V       */
#if defined(QSIM_PLAIN_ALL)
#define QS_D1(A, B) { const double2 a_ = A, b_ = B; A = cfma(u01, b_, cmul(u00, a_)); B = cfma(u11, b_, cmul(u10, a_)); }
#else
#define QS_D1(A, B) {                                                                               \
    double t0_ = fma(-u00.y, A.y, u00.x * A.x), t1_ = fma(u00.y, A.x, u00.x * A.y);                 \
    double t2_ = fma(-u10.y, A.y, u10.x * A.x), t3_ = fma(u10.y, A.x, u10.x * A.y);                 \
    t0_ = fma(u01.x, B.x, t0_); t1_ = fma(u01.x, B.y, t1_);                                         \
    t2_ = fma(-u11.y, B.y, fma(u11.x, B.x, t2_)); t3_ = fma(u11.x, B.y, t3_);                       \
    QS_IP_FNMA(A.x, u01.y, B.y, t0_); QS_IP_FMA(A.y, u01.y, B.x, t1_);                              \
    QS_IP_FMA(B.y, u11.y, B.x, t3_); QS_IP_MOV(B.x, t2_); }
#endif
#if defined(QSIM_PLAIN_ALL)
#define QS_AN(A, B) { const double2 a_ = A, b_ = B; A = cmul(u01, b_); B = cmul(u10, a_); }
#else
#define QS_AN(A, B) {                                                                               \
    const double t2_ = fma(-u10.y, A.y, u10.x * A.x), t3_ = fma(u10.y, A.x, u10.x * A.y);           \
    const double t0_ = u01.x * B.x, t1_ = u01.x * B.y;                                              \
    QS_IP_FNMA(A.x, u01.y, B.y, t0_); QS_IP_FMA(A.y, u01.y, B.x, t1_);                              \
    QS_IP_MOV(B.x, t2_); QS_IP_MOV(B.y, t3_); }
#endif
#if defined(QSIM_PLAIN_PERM) || defined(QSIM_PLAIN_ALL)
#define QS_SW(A, B) { const double2 t_ = A; A = B; B = t_; }
#else
#if defined(QSIM_SWAP_MOV)
#define QS_SW(A, B) { const double t0_ = A.x, t1_ = A.y;                                            \
    QS_IP_MOV(A.x, B.x); QS_IP_MOV(A.y, B.y); QS_IP_MOV(B.x, t0_); QS_IP_MOV(B.y, t1_); }
#else
#define QS_SWAP64(P, Q) {                                                                           \
    unsigned pl_ = __double2loint(P), ph_ = __double2hiint(P), ql_ = __double2loint(Q), qh_ = __double2hiint(Q); \
    asm volatile("v_swap_b32 %0, %1" : "+v"(pl_), "+v"(ql_));                                       \
    asm volatile("v_swap_b32 %0, %1" : "+v"(ph_), "+v"(qh_));                                       \
    P = __hiloint2double(ph_, pl_); Q = __hiloint2double(qh_, ql_); }
#define QS_SW(A, B) { QS_SWAP64(A.x, B.x) QS_SWAP64(A.y, B.y) }
#endif
#endif
#if defined(QSIM_PLAIN_ALL)
#define QS_DR(A, B) { const double2 a_ = A, b_ = B;                                               \
    A = make_double2(fma(u01.x, b_.x, u00.x * a_.x), fma(u01.x, b_.y, u00.x * a_.y));               \
    B = make_double2(fma(u11.x, b_.x, u10.x * a_.x), fma(u11.x, b_.y, u10.x * a_.y)); }
#else
#define QS_DR(A, B) {                                                                               \
    const double tx_ = u00.x * A.x, ty_ = u00.x * A.y, sx_ = u10.x * A.x, sy_ = u10.x * A.y;        \
    QS_IP_FMA(A.x, u01.x, B.x, tx_); QS_IP_FMA(A.y, u01.x, B.y, ty_);                               \
    QS_IP_FMA(B.x, u11.x, B.x, sx_); QS_IP_FMA(B.y, u11.x, B.y, sy_); }
#endif
#if defined(QSIM_PLAIN_PERM) || defined(QSIM_PLAIN_ALL)
#define QS_YL(A, B) { const double2 a_ = A, b_ = B; A = make_double2(b_.y, -b_.x); B = make_double2(-a_.y, a_.x); }
#else
#define QS_YL(A, B) { const double t0_ = A.x, t1_ = A.y;                                            \
    QS_IP_MOV(A.x, B.y); QS_IP_NEG(A.y, B.x); QS_IP_NEG(B.x, t1_); QS_IP_MOV(B.y, t0_); }
#endif
#if defined(QSIM_PLAIN_ALL)
#define QS_PH(A) { A = cmul(u00, A); }
#else
#define QS_PH(A) { const double p_ = u00.y * A.x, t_ = u00.x * A.x;                                 \
    QS_IP_FNMA(A.x, u00.y, A.y, t_); QS_IP_FMA(A.y, u00.x, A.y, p_); }
#endif
#if defined(QSIM_PLAIN_PERM) || defined(QSIM_PLAIN_ALL)
#define QS_PN(A) { A = make_double2(-A.x, -A.y); }
#else
#define QS_PN(A) { QS_IP_NEG(A.x, A.x); QS_IP_NEG(A.y, A.y); }
#endif
#if defined(QSIM_PLAIN_PERM) || defined(QSIM_PLAIN_ALL)
#define QS_PI(A) { A = make_double2(-A.y, A.x); }
#else
#define QS_PI(A) { const double t_ = A.x; QS_IP_NEG(A.x, A.y); QS_IP_MOV(A.y, t_); }
#endif
#if defined(QSIM_PLAIN_PERM) || defined(QSIM_PLAIN_ALL)
#define QS_PM(A) { A = make_double2(A.y, -A.x); }
#else
#define QS_PM(A) { const double t_ = A.x; QS_IP_MOV(A.x, A.y); QS_IP_NEG(A.y, t_); }
#endif
// A *= W with W in vector registers (a product of two pool entries)
#define QS_IP_FMA_V(D, S, V, C)  asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(D) : "v"(S), "v"(V), "v"(C))
#define QS_IP_FNMA_V(D, S, V, C) asm volatile("v_fma_f64 %0, -%1, %2, %3" : "+v"(D) : "v"(S), "v"(V), "v"(C))
#define QS_PHS(A, U) { const double p_ = U.y * A.x, t_ = U.x * A.x;                                  \
    QS_IP_FNMA(A.x, U.y, A.y, t_); QS_IP_FMA(A.y, U.x, A.y, p_); }
#define QS_PHV(A, W) { const double p_ = W.y * A.x, t_ = W.x * A.x;                                  \
    QS_IP_FNMA_V(A.x, W.y, A.y, t_); QS_IP_FMA_V(A.y, W.x, A.y, p_); }
// register pairs (bit J clear / set) of each 1q variant
#define QS_PAIRS_0(OP) OP(x0, x1) OP(x2, x3) OP(x4, x5) OP(x6, x7)
#define QS_PAIRS_1(OP) OP(x0, x2) OP(x1, x3) OP(x4, x6) OP(x5, x7)
#define QS_PAIRS_2(OP) OP(x0, x4) OP(x1, x5) OP(x2, x6) OP(x3, x7)
#define QS_PAIRS_3(OP) OP(x2, x3) OP(x6, x7)
#define QS_PAIRS_4(OP) OP(x4, x5) OP(x6, x7)
#define QS_PAIRS_5(OP) OP(x1, x3) OP(x5, x7)
#define QS_PAIRS_6(OP) OP(x4, x6) OP(x5, x7)
#define QS_PAIRS_7(OP) OP(x1, x5) OP(x3, x7)
#define QS_PAIRS_8(OP) OP(x2, x6) OP(x3, x7)
#define QS_CASES_1Q(BASE, OP)                                                             \
  case BASE + 0: QS_PAIRS_0(OP) break;  case BASE + 1: QS_PAIRS_1(OP) break;             \
  case BASE + 2: QS_PAIRS_2(OP) break;  case BASE + 3: QS_PAIRS_3(OP) break;             \
  case BASE + 4: QS_PAIRS_4(OP) break;  case BASE + 5: QS_PAIRS_5(OP) break;             \
  case BASE + 6: QS_PAIRS_6(OP) break;  case BASE + 7: QS_PAIRS_7(OP) break;             \
  case BASE + 8: QS_PAIRS_8(OP) break;
#define QS_CASES_PHASE(BASE, OP)                                                                  \
  case BASE + 0: OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7) break;                   \
  case BASE + 1: OP(x1) OP(x3) OP(x5) OP(x7) break;  case BASE + 2: OP(x2) OP(x3) OP(x6) OP(x7) break; \
  case BASE + 3: OP(x3) OP(x7) break;                case BASE + 4: OP(x4) OP(x5) OP(x6) OP(x7) break; \
  case BASE + 5: OP(x5) OP(x7) break;                case BASE + 6: OP(x6) OP(x7) break;          \
  case BASE + 7: OP(x7) break;
// 4x4 on (v00, v01 = qb set, v10 = qa set, v11); the 16 entries are read where they are used
#define QS_D2(V0, V1, V2, V3) {                                                                         \
    const double2 a_ = V0, b_ = V1, c_ = V2, d_ = V3;                                                   \
    const double2 r0_ = cfma(a.mat[mq + 3], d_, cfma(a.mat[mq + 2], c_, cfma(a.mat[mq + 1], b_, cmul(a.mat[mq + 0], a_))));     \
    const double2 r1_ = cfma(a.mat[mq + 7], d_, cfma(a.mat[mq + 6], c_, cfma(a.mat[mq + 5], b_, cmul(a.mat[mq + 4], a_))));     \
    const double2 r2_ = cfma(a.mat[mq + 11], d_, cfma(a.mat[mq + 10], c_, cfma(a.mat[mq + 9], b_, cmul(a.mat[mq + 8], a_))));   \
    const double2 r3_ = cfma(a.mat[mq + 15], d_, cfma(a.mat[mq + 14], c_, cfma(a.mat[mq + 13], b_, cmul(a.mat[mq + 12], a_)))); \
    QS_IP_MOV(V0.x, r0_.x); QS_IP_MOV(V0.y, r0_.y); QS_IP_MOV(V1.x, r1_.x); QS_IP_MOV(V1.y, r1_.y);     \
    QS_IP_MOV(V2.x, r2_.x); QS_IP_MOV(V2.y, r2_.y); QS_IP_MOV(V3.x, r3_.x); QS_IP_MOV(V3.y, r3_.y); }

// min waves per SIMD asked of the register allocator: what the LDS footprint admits, capped at 4
// (5 forces spills at T = 11 and measured slower)
constexpr int tile_waves(int T) {
  return (160 * 1024) / ((1 << T) * 16) > 4 ? 4 : (160 * 1024) / ((1 << T) * 16);
}

// PERSIST: a resident grid walks tiles b, b + gridDim.x, ...; the global loads of the next tile are
// issued right after the current tile has been written to LDS, so they are in flight during the
// whole gate phase (software pipelining across tiles).
template <int T, bool PERSIST, bool NT>
__global__ __launch_bounds__(kTileThreads, tile_waves(T)) void k_tile(const TileArgs a, const unsigned ntiles) {
  constexpr int N = 1 << T;
  constexpr int LOW = kTileLow;
  constexpr int NH = T - LOW;                         // tile high bits
  constexpr int BLOCK = kTileThreads;
  constexpr int TB = kTileThreadBits;                 // thread id bits: LOW element bits + row bits
  constexpr int PER = N / BLOCK > 0 ? N / BLOCK : 1;  // tile elements per thread
  constexpr int NBLK = N >> kGroupBits;               // register blocks per tile (<= BLOCK)
  static_assert(NBLK <= BLOCK, "one register block per thread");
  __shared__ double2 lds[N];
  const int tid = threadIdx.x;
  const bool elem_ok = N >= BLOCK || tid < N;         // tiny tiles: surplus threads idle
  // global index of a tile's element 0: the tile number enumerates the non-tile bits
  auto tile_base = [&](unsigned tile_no) -> u64 {
    u64 base = (u64)tile_no << LOW;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int p = a.h[j];
      base = ((base >> p) << (p + 1)) | (base & ((1ull << p) - 1));
    }
    return base;
  };
  // element t = tid + BLOCK * j -> row = (tid >> LOW) | (j << (TB - LOW)): the thread part of the
  // offset is computed once, the j part is wave-uniform (scalar registers)
  u64 off_tid = tid & ((1 << LOW) - 1);
#pragma unroll
  for (int i = 0; i < TB - LOW && i < NH; ++i) off_tid |= (u64)((tid >> (LOW + i)) & 1) << a.h[i];
  auto off_j = [&](int j) -> u64 {
    u64 o = 0;
#pragma unroll
    for (int i = TB - LOW; i < NH; ++i) o |= (u64)((j >> (i - (TB - LOW))) & 1) << a.h[i];
    return o;
  };
  unsigned tile = blockIdx.x;
  u64 base = tile_base(tile);
  double2 v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) if (elem_ok) v[j] = ld_amp<NT>(a.amp + base + off_tid + off_j(j));
  for (;;) {
#pragma unroll
  for (int j = 0; j < PER; ++j) if (elem_ok) lds[lds_slot(tid + BLOCK * j)] = v[j];
  __syncthreads();
  const unsigned next = tile + gridDim.x;
  const bool has_next = PERSIST && next < ntiles;
  u64 next_base = 0;
  if (PERSIST && has_next) {
    next_base = tile_base(next);
#pragma unroll
    for (int j = 0; j < PER; ++j) if (elem_ok) v[j] = ld_amp<NT>(a.amp + next_base + off_tid + off_j(j));
  }

  const bool live = NBLK == BLOCK || tid < NBLK;
  int gi = 0;
  while (gi < a.ngates) {
    gi = __builtin_amdgcn_readfirstlane(gi);          // keep the descriptor reads scalar
    const u32x4 hd = a.g[gi++];                       // group header: opcode | count << 8 | bits << 16
    const unsigned hb = hd.x >> 16;
    const int s0 = hb & 15, s1 = (hb >> 4) & 15, s2 = (hb >> 8) & 15;
    const int ge = gi + ((hd.x >> 8) & 0xFF);
    const unsigned tb = insert_zero(insert_zero(insert_zero((unsigned)tid, s0), s1), s2);
    const unsigned b0 = 1u << s0, b1 = 1u << s1, b2 = 1u << s2;
    double2 x0, x1, x2, x3, x4, x5, x6, x7;
    if (live) {
      x0 = lds[lds_slot(tb)];            x1 = lds[lds_slot(tb | b0)];
      x2 = lds[lds_slot(tb | b1)];       x3 = lds[lds_slot(tb | b1 | b0)];
      x4 = lds[lds_slot(tb | b2)];       x5 = lds[lds_slot(tb | b2 | b0)];
      x6 = lds[lds_slot(tb | b2 | b1)];  x7 = lds[lds_slot(tb | b2 | b1 | b0)];
    }
    // The loop below runs on the CU's single scalar unit for every wave (SALU ~ 2x VALU per gate,
    // r01f ISA audit), so the bookkeeping is kept to: one add + one 128-bit load for the
    // descriptor (unsigned index), and the lane predicate evaluated only for gates that have one.
    for (unsigned q0 = (unsigned)gi, qe = (unsigned)ge; q0 != qe; ++q0) {
      const unsigned q = __builtin_amdgcn_readfirstlane(q0);
      const u32x4 g = a.g[q & 0xFF];                  // one s_load_dwordx4 (index the kernarg arrays directly:
                                                      // a pointer formed into them turns the loads into vector loads)
      const int mq = g.y & 0xFFFF;                    // (known-small indices fold into the load's offset); pool keeps 3 spare entries
      const double2 u00 = a.mat[mq], u01 = a.mat[mq + 1], u10 = a.mat[mq + 2], u11 = a.mat[mq + 3];
      const u64 outer = (u64)g.z | ((u64)g.w << 32);
      if ((base & outer) != outer) continue;
      // Lane predicate (control / diagonal bits that are tile bits outside the group).  EXEC is
      // narrowed by hand: a compiler-managed divergent region makes StructurizeCFG rewrite the
      // (uniform) opcode switch into flow blocks whose PHIs double-buffer x0..x7 (see QS_IP_*).
      // Everything up to the restore is VALU on x0..x7 / case-local temporaries + scalar branches.
      // (Threads that are not `live` -- tiles smaller than 8 x blockDim -- compute on registers
      // they never write back.)
      const unsigned bm = g.x >> 16;
      if (bm) {
        const u64 act = __builtin_amdgcn_ballot_w64((tb & bm) == bm);
        if (act == 0) continue;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_mov_b64 exec, %0" : : "s"(act) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      switch (g.x & 0xFF) {
        QS_CASES_1Q(OPC_DENSE1, QS_D1)
        QS_CASES_1Q(OPC_SWAP1, QS_SW)
        QS_CASES_1Q(OPC_ANTI1, QS_AN)
        QS_CASES_1Q(OPC_REAL1, QS_DR)
        QS_CASES_1Q(OPC_YLIKE1, QS_YL)
        QS_CASES_PHASE(OPC_PHASE, QS_PH)
        QS_CASES_PHASE(OPC_PHASE_NEG, QS_PN)
        QS_CASES_PHASE(OPC_PHASE_I, QS_PI)
        QS_CASES_PHASE(OPC_PHASE_NI, QS_PM)
        case OPC_DIAGR + 0: { const double2 w_ = cmul(u00, u01);     // bits 0 (u00) and 1 (u01)
          QS_PHS(x1, u00) QS_PHS(x5, u00) QS_PHS(x2, u01) QS_PHS(x6, u01) QS_PHV(x3, w_) QS_PHV(x7, w_) } break;
        case OPC_DIAGR + 1: { const double2 w_ = cmul(u00, u01);     // bits 0 (u00) and 2 (u01)
          QS_PHS(x1, u00) QS_PHS(x3, u00) QS_PHS(x4, u01) QS_PHS(x6, u01) QS_PHV(x5, w_) QS_PHV(x7, w_) } break;
        case OPC_DIAGR + 2: { const double2 w_ = cmul(u00, u01);     // bits 1 (u00) and 2 (u01)
          QS_PHS(x2, u00) QS_PHS(x3, u00) QS_PHS(x4, u01) QS_PHS(x5, u01) QS_PHV(x6, w_) QS_PHV(x7, w_) } break;
        case OPC_DIAGR + 3: {                                        // bits 0 (u00), 1 (u01), 2 (u10)
          const double2 w01_ = cmul(u00, u01), w02_ = cmul(u00, u10), w12_ = cmul(u01, u10), w012_ = cmul(w01_, u10);
          QS_PHS(x1, u00) QS_PHS(x2, u01) QS_PHS(x4, u10) QS_PHV(x3, w01_) QS_PHV(x5, w02_) QS_PHV(x6, w12_)
          QS_PHV(x7, w012_) } break;
        case OPC_DENSE2 + 1: QS_D2(x0, x2, x1, x3) QS_D2(x4, x6, x5, x7) break;   // qa = bit 0, qb = bit 1
        case OPC_DENSE2 + 2: QS_D2(x0, x4, x1, x5) QS_D2(x2, x6, x3, x7) break;   // qa = bit 0, qb = bit 2
        case OPC_DENSE2 + 3: QS_D2(x0, x1, x2, x3) QS_D2(x4, x5, x6, x7) break;   // qa = bit 1, qb = bit 0
        case OPC_DENSE2 + 5: QS_D2(x0, x4, x2, x6) QS_D2(x1, x5, x3, x7) break;   // qa = bit 1, qb = bit 2
        case OPC_DENSE2 + 6: QS_D2(x0, x1, x4, x5) QS_D2(x2, x3, x6, x7) break;   // qa = bit 2, qb = bit 0
        case OPC_DENSE2 + 7: QS_D2(x0, x2, x4, x6) QS_D2(x1, x3, x5, x7) break;   // qa = bit 2, qb = bit 1
        default: break;
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_mov_b64 exec, -1" : : : "memory");   // whole waves: blockDim is a multiple of 64
      __builtin_amdgcn_sched_barrier(0);
    }
    if (live) {
      lds[lds_slot(tb)] = x0;            lds[lds_slot(tb | b0)] = x1;
      lds[lds_slot(tb | b1)] = x2;       lds[lds_slot(tb | b1 | b0)] = x3;
      lds[lds_slot(tb | b2)] = x4;       lds[lds_slot(tb | b2 | b0)] = x5;
      lds[lds_slot(tb | b2 | b1)] = x6;  lds[lds_slot(tb | b2 | b1 | b0)] = x7;
    }
    __syncthreads();
    gi = ge;
  }
  {
    double2 w[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) if (elem_ok) w[j] = lds[lds_slot(tid + BLOCK * j)];
#pragma unroll
    for (int j = 0; j < PER; ++j) if (elem_ok) st_amp<NT>(a.amp + base + off_tid + off_j(j), w[j]);
  }
  if (!PERSIST || !has_next) break;
  tile = next;
  base = next_base;
  }  // tile loop (each thread re-writes only the LDS slots it just read: no barrier needed)
}

// ---- host planner: op list -> passes -> register groups ----------------------------------------
enum { TG_DENSE1 = 0, TG_PHASE = 1, TG_DENSE2 = 2, TG_ANTI1 = 3, TG_SWAP1 = 4 };   // FusedOp::kind

struct FusedOp {
  int kind;            // TG_DENSE1 / TG_ANTI1 / TG_SWAP1 (target, optional control), TG_PHASE, TG_DENSE2
  int target[2];       // 1q kinds: target[0]; TG_DENSE2: (qa, qb)
  int ntargets;
  int control;         // 1q kinds: control qubit or -1
  int bits[2];         // TG_PHASE: qubits that must be 1
  int nbits;
  int qubits[2];       // every qubit the op touches (for ordering)
  int nq;
  double2 m[16];
  int nm;              // matrix entries (4, 1 or 16)
  int halvings;        // algorithmic bytes = 32 B x 2^(k - halvings)  
};

static void set_1q_kind(FusedOp* o) {   // o->m holds the 2x2
  const bool zero_diag = o->m[0].x == 0 && o->m[0].y == 0 && o->m[3].x == 0 && o->m[3].y == 0;
  const bool ones = o->m[1].x == 1 && o->m[1].y == 0 && o->m[2].x == 1 && o->m[2].y == 0;
  o->kind = zero_diag ? (ones ? TG_SWAP1 : TG_ANTI1) : TG_DENSE1;
}

// Same classification as gate_1q / gate_2q; returns false for an identity.
static bool classify_op(int nq, const int32_t* q, const double* U, FusedOp* o) {
  o->nq = nq;
  o->qubits[0] = q[0];
  o->qubits[1] = nq == 2 ? q[1] : -1;
  o->control = -1;
  o->nbits = 0;
  o->ntargets = 0;
  o->halvings = 0;
  auto C = [&](int i) { return make_double2(U[2 * i], U[2 * i + 1]); };
  if (nq == 1) {
    const bool diag = is_zero(U[2], U[3]) && is_zero(U[4], U[5]);
    if (diag && is_one(U[0], U[1])) {
      if (is_one(U[6], U[7])) return false;
      o->kind = TG_PHASE; o->bits[0] = q[0]; o->nbits = 1; o->m[0] = C(3); o->nm = 1;
      return true;
    }
    o->target[0] = q[0]; o->ntargets = 1;
    for (int i = 0; i < 4; ++i) o->m[i] = C(i);
    o->nm = 4;
    set_1q_kind(o);
    return true;
  }
  auto z = [&](int r, int c) { return is_zero(U[2 * (4 * r + c)], U[2 * (4 * r + c) + 1]); };
  auto one = [&](int r, int c) { return is_one(U[2 * (4 * r + c)], U[2 * (4 * r + c) + 1]); };
  bool offdiag_zero = true;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c)
      if (r != c && !z(r, c)) offdiag_zero = false;
  const bool ctrl_a = one(0, 0) && one(1, 1) && z(0, 1) && z(1, 0) && z(0, 2) && z(0, 3) && z(1, 2) &&
                      z(1, 3) && z(2, 0) && z(2, 1) && z(3, 0) && z(3, 1);
  const bool ctrl_b = one(0, 0) && one(2, 2) && z(0, 2) && z(2, 0) && z(0, 1) && z(0, 3) && z(2, 1) &&
                      z(2, 3) && z(1, 0) && z(1, 2) && z(3, 0) && z(3, 2);
  if (offdiag_zero && one(0, 0) && one(1, 1) && one(2, 2)) {
    if (one(3, 3)) return false;
    o->kind = TG_PHASE; o->bits[0] = q[0]; o->bits[1] = q[1]; o->nbits = 2; o->m[0] = C(15); o->nm = 1;
    return true;
  }
  if (ctrl_a || ctrl_b) {
    o->control = ctrl_a ? q[0] : q[1];
    o->target[0] = ctrl_a ? q[1] : q[0];
    o->ntargets = 1;
    if (ctrl_a) { o->m[0] = C(10); o->m[1] = C(11); o->m[2] = C(14); o->m[3] = C(15); }
    else        { o->m[0] = C(5);  o->m[1] = C(7);  o->m[2] = C(13); o->m[3] = C(15); }
    o->nm = 4;
    set_1q_kind(o);
    if (o->kind == TG_DENSE1 && o->m[1].x == 0 && o->m[1].y == 0 && o->m[2].x == 0 && o->m[2].y == 0 &&
        o->m[0].x == 1 && o->m[0].y == 0) {   // controlled phase written as CU: diag(1, d)
      o->kind = TG_PHASE; o->bits[0] = q[0]; o->bits[1] = q[1]; o->nbits = 2; o->m[0] = o->m[3]; o->nm = 1;
      o->ntargets = 0; o->control = -1;
    }
    return true;
  }
  o->kind = TG_DENSE2; o->target[0] = q[0]; o->target[1] = q[1]; o->ntargets = 2;
  for (int i = 0; i < 16; ++i) o->m[i] = C(i);
  o->nm = 16;
  const bool swap = one(0, 0) && one(3, 3) && one(1, 2) && one(2, 1) && z(1, 1) && z(2, 2) && z(0, 1) && z(0, 2) &&
                    z(0, 3) && z(1, 0) && z(1, 3) && z(2, 0) && z(2, 3) && z(3, 0) && z(3, 1) && z(3, 2);
  o->halvings = swap ? 1 : 0;   // SWAP only exchanges |01> and |10>
  return true;
}

template <int T>
static int launch_tile(const TileArgs& a, const qsim_chunk* c, hipStream_t stream, double alg_bytes) {
  if constexpr (T > kTileBitsMax) {
    return fail(QSIM_ERR_INVALID, "internal: tile size %d not built", T);
  } else {
  const u64 ntiles = 1ull << (c->k - T);
  ProfileScope prof(6, alg_bytes, stream, 32.0 * (double)amps(c));
  if (tuning().tile_persistent) {
    static int resident = 0;            // workgroups per CU that registers and LDS admit
    if (!resident) {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_tile<T, true, true>, kTileThreads, 0) != hipSuccess || n < 1) n = 1;
      resident = n;
    }
    const int per_cu = std::max(1, std::min(tuning().tile_wgs_per_cu, resident));
    const u64 blocks = std::min<u64>(ntiles, (u64)tuning().num_cus * per_cu);
    hipLaunchKernelGGL((k_tile<T, true, true>), dim3((unsigned)blocks), dim3(kTileThreads), 0, stream, a, (unsigned)ntiles);
  } else {
    bool nt = c->span_bytes > tuning().mall_bytes;
    if (tuning().force_nt >= 0) nt = tuning().force_nt != 0;
    if (nt) hipLaunchKernelGGL((k_tile<T, false, true>), dim3((unsigned)ntiles), dim3(kTileThreads), 0, stream, a, (unsigned)ntiles);
    else hipLaunchKernelGGL((k_tile<T, false, false>), dim3((unsigned)ntiles), dim3(kTileThreads), 0, stream, a, (unsigned)ntiles);
  }
  prof.done(stream);
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
  }
}

static int launch_tile_any(const TileArgs& a, int T, const qsim_chunk* c, hipStream_t stream, double alg_bytes) {
  switch (T) {
    case 8: return launch_tile<8>(a, c, stream, alg_bytes);
    case 9: return launch_tile<9>(a, c, stream, alg_bytes);
    case 10: return launch_tile<10>(a, c, stream, alg_bytes);
    case 11: return launch_tile<11>(a, c, stream, alg_bytes);
    case 12: return launch_tile<12>(a, c, stream, alg_bytes);
  }
  return fail(QSIM_ERR_INVALID, "internal: tile size %d", T);
}

constexpr int kTileMinChunk = 8;   // smaller chunks run gate by gate

static inline u64 op_qmask(const FusedOp& o) {
  u64 m = 1ull << o.qubits[0];
  if (o.nq == 2) m |= 1ull << o.qubits[1];
  return m;
}

// Matrix-pool entries an op needs under the opcode it will get.
static int pool_entries(const FusedOp& o) {
  const bool sp = tuning().tile_special;
  auto is = [&](int e, double re, double im) { return o.m[e].x == re && o.m[e].y == im; };
  switch (o.kind) {
    case TG_SWAP1: return 0;
    case TG_PHASE: return (sp && (is(0, -1, 0) || is(0, 0, 1) || is(0, 0, -1))) ? 0 : 1;
    case TG_ANTI1: return (sp && is(1, 0, -1) && is(2, 0, 1)) ? 0 : 4;
    case TG_DENSE2: return 16;
    default: return 4;
  }
}

// Split one pass's ops (list order) into register groups of <= kGroupBits target tile bits and
// write the gate stream.  Ops that do not fit the argument budget stay un-emitted (they and
// everything that depends on them wait for the next launch).
static void emit_groups(const std::vector<FusedOp>& ops, const std::vector<size_t>& members,
                        const std::vector<int>& high, int T, TileArgs* a, std::vector<char>* emitted) {
  const int low = kTileLow;
  auto tile_pos = [&](int b) -> int {
    if (b < low) return b;
    for (size_t j = 0; j < high.size(); ++j) if (high[j] == b) return low + (int)j;
    return -1;
  };
  std::vector<char> done(members.size(), 0);
  size_t left = members.size();
  a->ngates = 0;
  int pool = 0;                       // next free matrix entry; 3 spare entries stay at the end
  while (left) {
    std::vector<int> S;               // tile bits of this group
    std::vector<size_t> grp;          // indices into members
    u64 blocked = 0;
    // Budget estimate in half units: a phase gate that may be merged with others (OPC_DIAGR) is
    // counted as half a descriptor and half a pool entry; the exact budget is enforced when the
    // group is written out (a group that overflows is cut there, the rest waits for the next pass).
    int slots2 = 0, pool2 = 0;
    const bool merge_on = tuning().tile_merge_diag != 0;
    for (size_t mi = 0; mi < members.size(); ++mi) {
      if (done[mi]) continue;
      const FusedOp& o = ops[members[mi]];
      const u64 qm = op_qmask(o);
      bool ok = !(blocked & qm);
      int need[2], nneed = 0;
      const bool mergeable = merge_on && o.kind == TG_PHASE && pool_entries(o) == 1;
      if (ok) {
        for (int t = 0; t < o.ntargets; ++t) {
          const int p = tile_pos(o.target[t]);
          if (std::find(S.begin(), S.end(), p) == S.end()) need[nneed++] = p;
        }
        if ((int)S.size() + nneed > kGroupBits) ok = false;
        if (2 * (a->ngates + 1) + slots2 + (mergeable ? 1 : 2) > 2 * kTileMaxGates || (int)grp.size() + 1 > 255) ok = false;
        if (2 * pool + pool2 + (mergeable ? 1 : 2 * pool_entries(o)) > 2 * (kTileMaxMat - 3)) ok = false;
      }
      if (!ok) { blocked |= qm; continue; }
      for (int t = 0; t < nneed; ++t) S.push_back(need[t]);
      grp.push_back(mi);
      slots2 += mergeable ? 1 : 2;
      pool2 += mergeable ? 1 : 2 * pool_entries(o);
    }
    if (grp.empty()) break;           // argument budget exhausted: the rest waits for the next launch
    // pad the group with the highest unused tile bits (high bits keep LDS accesses contiguous)
    for (int b = T - 1; (int)S.size() < kGroupBits && b >= 0; --b)
      if (std::find(S.begin(), S.end(), b) == S.end()) S.push_back(b);
    std::sort(S.begin(), S.end());
    auto reg_pos = [&](int tile_bit) -> int {
      for (int j = 0; j < kGroupBits; ++j) if (S[j] == tile_bit) return j;
      return -1;
    };
    TileGate hd;
    std::memset(&hd, 0, sizeof hd);
    hd.opcode = OPC_GROUP;
    hd.blk_mask = (uint16_t)(S[0] | (S[1] << 4) | (S[2] << 8));
    const int hd_at = a->ngates++;
    int n_emitted = 0;
    auto emit = [&](TileGate g, const double2* m, int nm) {
      if (nm) {
        g.mat = (uint16_t)pool;
        for (int e = 0; e < nm; ++e) a->mat[pool + e] = m[e];
        pool += nm;
      }
      put_gate(a, a->ngates++, g);
      ++n_emitted;
    };
    // Phase gates with ONE register bit and the same predicate (lane bits + outer bits) are merged
    // (the QFT's CR(k, a), CR(k, b), CR(k, c) for the group's register bits a, b, c): diagonal
    // gates commute with everything except a non-diagonal gate on one of their bits, so an open
    // accumulator is written out before such a gate on a register bit it has touched, or at the
    // end of the group.  One descriptor instead of up to three: the gate loop is scalar-issue bound.
    struct Acc { uint16_t blk; u64 outer; double2 phi[3]; unsigned touched; int count; TileGate single; };
    std::vector<Acc> open;
    bool cut = false;
    auto flush = [&](size_t i) {
      const Acc acc = open[i];
      open.erase(open.begin() + (long)i);
      if (acc.count == 1) { emit(acc.single, &acc.phi[__builtin_ctz(acc.touched)], 1); return; }
      TileGate g;
      std::memset(&g, 0, sizeof g);
      g.blk_mask = acc.blk;
      g.outer_mask = acc.outer;
      double2 m[3];
      int nm = 0;
      for (int r = 0; r < 3; ++r) if (acc.touched & (1u << r)) m[nm++] = acc.phi[r];
      if (nm == 1) g.opcode = (uint8_t)(OPC_PHASE + acc.touched);
      else g.opcode = (uint8_t)(OPC_DIAGR + (acc.touched == 3 ? 0 : acc.touched == 5 ? 1 : acc.touched == 6 ? 2 : 3));
      emit(g, m, nm);
    };
    for (size_t mi : grp) {
      const FusedOp& o = ops[members[mi]];
      TileGate g;
      std::memset(&g, 0, sizeof g);
      unsigned reg_mask = 0;
      int ctrl_reg = -1;
      auto require_one = [&](int qubit) {      // a control / phase bit
        const int p = tile_pos(qubit);
        if (p < 0) { g.outer_mask |= 1ull << qubit; return; }
        const int r = reg_pos(p);
        if (r >= 0) { reg_mask |= 1u << r; ctrl_reg = r; }
        else g.blk_mask |= (uint16_t)(1u << p);
      };
      auto is = [&](int e, double re, double im) { return o.m[e].x == re && o.m[e].y == im; };
      const bool sp = tuning().tile_special;
      {   // exact budget: descriptors and pool entries written so far + what the open runs will need
        int reserve_pool = 0;
        for (const Acc& acc : open) reserve_pool += __builtin_popcount(acc.touched);
        if (a->ngates + (int)open.size() + 1 > kTileMaxGates ||
            pool + reserve_pool + std::max(1, pool_entries(o)) > kTileMaxMat - 3) { cut = true; break; }
      }
      done[mi] = 1;
      (*emitted)[mi] = 1;
      --left;
      if (o.kind == TG_PHASE) {
        for (int t = 0; t < o.nbits; ++t) require_one(o.bits[t]);
        const int fam = !sp ? OPC_PHASE : (is(0, -1, 0) ? OPC_PHASE_NEG : (is(0, 0, 1) ? OPC_PHASE_I : (is(0, 0, -1) ? OPC_PHASE_NI : OPC_PHASE)));
        g.opcode = (uint8_t)(fam + reg_mask);
        if (tuning().tile_merge_diag && fam == OPC_PHASE && __builtin_popcount(reg_mask) == 1) {
          const int r = __builtin_ctz(reg_mask);
          size_t i = 0;
          while (i < open.size() && !(open[i].blk == g.blk_mask && open[i].outer == g.outer_mask)) ++i;
          if (i == open.size()) {
            Acc acc;
            acc.blk = g.blk_mask; acc.outer = g.outer_mask; acc.touched = 0; acc.count = 0; acc.single = g;
            for (int e = 0; e < 3; ++e) acc.phi[e] = make_double2(1.0, 0.0);
            open.push_back(acc);
          }
          Acc& acc = open[i];
          const double2 f = acc.phi[r], m = o.m[0];
          acc.phi[r] = make_double2(f.x * m.x - f.y * m.y, f.x * m.y + f.y * m.x);
          acc.touched |= 1u << r;
          ++acc.count;
          continue;
        }
      } else if (o.kind == TG_DENSE2) {
        g.opcode = (uint8_t)(OPC_DENSE2 + 3 * reg_pos(tile_pos(o.target[0])) + reg_pos(tile_pos(o.target[1])));
      } else {
        const int J = reg_pos(tile_pos(o.target[0]));
        if (o.control >= 0) require_one(o.control);
        int fam = o.kind == TG_DENSE1 ? OPC_DENSE1 : (o.kind == TG_SWAP1 ? OPC_SWAP1 : OPC_ANTI1);
        if (sp && o.kind == TG_DENSE1 && o.m[0].y == 0 && o.m[1].y == 0 && o.m[2].y == 0 && o.m[3].y == 0) fam = OPC_REAL1;
        if (sp && o.kind == TG_ANTI1 && is(1, 0, -1) && is(2, 0, 1)) fam = OPC_YLIKE1;
        g.opcode = (uint8_t)(fam + opc_1q_variant(J, ctrl_reg));
      }
      if (o.kind != TG_PHASE) {               // a non-diagonal gate: its targets end the open phase runs on them
        unsigned tmask = 0;
        for (int t = 0; t < o.ntargets; ++t) tmask |= 1u << reg_pos(tile_pos(o.target[t]));
        for (size_t i = open.size(); i-- > 0;) if (open[i].touched & tmask) flush(i);
      }
      emit(g, o.m, std::min(pool_entries(o), o.nm));
    }
    while (!open.empty()) flush(0);
    hd.count = (uint8_t)n_emitted;
    put_gate(a, hd_at, hd);
    if (cut) break;                   // argument budget exhausted inside the group
  }
}

// Greedy pass builder.  Ops are taken in list order; an op that does not fit the current tile
// blocks its qubits, and later ops on blocked qubits wait for the next pass, so any two ops
// sharing a qubit keep their order (ops on disjoint qubits commute).
static int run_fused(qsim_chunk* c, const std::vector<FusedOp>& ops, int* n_passes) {
  const int k = c->k;
  const Tuning& tune = tuning();
  const int T = k < kTileBitsMax ? k : kTileBitsMax;
  const int low = kTileLow;
  const int cap = T - low;                      // tile high-bit capacity
  std::vector<char> done(ops.size(), 0);
  size_t remaining = ops.size();
  size_t first = 0;
  *n_passes = 0;
  while (remaining) {
    std::vector<int> high;                      // chosen high bits
    std::vector<size_t> members;
    u64 blocked = 0;
    while (first < ops.size() && done[first]) ++first;
    for (size_t i = first; i < ops.size() && (int)members.size() < tune.max_gates_per_pass; ++i) {
      if (done[i]) continue;
      const FusedOp& o = ops[i];
      const u64 qmask = op_qmask(o);
      bool ok = !(blocked & qmask);
      int need[2], nneed = 0;
      if (ok) {
        for (int t = 0; t < o.ntargets; ++t) {
          const int b = o.target[t];
          if (b >= low && std::find(high.begin(), high.end(), b) == high.end()) need[nneed++] = b;
        }
        if ((int)high.size() + nneed > cap) ok = false;
      }
      if (!ok) { blocked |= qmask; continue; }
      for (int t = 0; t < nneed; ++t) high.push_back(need[t]);
      members.push_back(i);
    }
    if (members.empty()) return fail(QSIM_ERR_INVALID, "internal: fused planner made no progress");
    // fill the tile with the lowest unused bits so it always has T bits
    for (int b = low; (int)high.size() < cap && b < k; ++b)
      if (std::find(high.begin(), high.end(), b) == high.end()) high.push_back(b);
    std::sort(high.begin(), high.end());
    if (tune.debug_skip_gates == 2) for (int j = 0; j < cap; ++j) high[j] = low + j;   // contiguous tiles (floor probe)
    if (tune.debug_skip_gates == 3) for (int j = 0; j < cap; ++j) high[j] = k - cap + j; // far-strided tiles
    TileArgs a;
    std::memset(&a, 0, sizeof a);
    a.amp = c->amp;
    for (size_t j = 0; j < high.size(); ++j) a.h[j] = (uint8_t)high[j];
    std::vector<char> emitted(members.size(), 0);
    emit_groups(ops, members, high, T, &a, &emitted);
    size_t n_emitted = 0;
    double alg_bytes = 0;   // dense 32N, diagonal / controlled / SWAP 16N, CZ/CR 8N
    for (size_t mi = 0; mi < members.size(); ++mi)
      if (emitted[mi]) {
        const FusedOp& o = ops[members[mi]];
        const int halvings = o.kind == TG_PHASE ? o.nbits : (o.control >= 0 ? 1 : o.halvings);
        alg_bytes += 32.0 * (double)(amps(c) >> halvings);
        done[members[mi]] = 1; --remaining; ++n_emitted;
      }
    if (!n_emitted) return fail(QSIM_ERR_INVALID, "internal: fused planner emitted nothing");
    if (tune.debug_stats) {
      int groups = 0;
      for (int i = 0; i < a.ngates; ++i) groups += get_gate(&a, i).opcode == OPC_GROUP;
      std::fprintf(stderr, "[qsim] pass %d: %zu gates, %d groups, %d entries\n", *n_passes, n_emitted, groups, a.ngates);
    }
    if (tune.debug_skip_gates) a.ngates = 0;       // profiling aid: load -> LDS -> store only
    int rc = launch_tile_any(a, T, c, c->stream, alg_bytes);
    if (rc) return rc;
    ++*n_passes;
  }
  return QSIM_OK;
}

// ------------------------------------------------------------------ argument checks
static int check_chunk(const qsim_chunk* c, const char* what) {
  if (!c || !c->amp) return fail(QSIM_ERR_INVALID, "%s: null chunk", what);
  return QSIM_OK;
}

static int check_local_qubit(const qsim_chunk* c, int q) {
  if (q < 0) return fail(QSIM_ERR_INVALID, "qubit %d is negative", q);
  if (q >= c->k)
    return fail(QSIM_ERR_NONLOCAL,
                "qubit %d >= log2(chunk_size)=%d: non-local gate requires layout/collect step",
                q, c->k);
  return QSIM_OK;
}

static int check_group(qsim_chunk* const* cs, int n, const char* what) {
  for (int i = 0; i < n; ++i) {
    int rc = check_chunk(cs[i], what);
    if (rc) return rc;
    if (cs[i]->k != cs[0]->k) return fail(QSIM_ERR_INVALID, "%s: chunks differ in size", what);
    if (cs[i]->device != cs[0]->device)
      return fail(QSIM_ERR_INVALID, "%s: chunks live on different devices", what);
    for (int j = 0; j < i; ++j)
      if (cs[i]->amp == cs[j]->amp) return fail(QSIM_ERR_INVALID, "%s: the same chunk twice", what);
  }
  return QSIM_OK;
}

// ------------------------------------------------------------------ misc kernels
__global__ void k_fill_zero(double2* p, u64 n, int set0) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = make_double2((i == 0 && set0) ? 1.0 : 0.0, 0.0);
}

__device__ __forceinline__ u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// amplitude i = (u(2i), u(2i+1)), u(j) = (splitmix64(seed + j) >> 11) * 2^-52 - 1 in [-1, 1)
__global__ void k_fill_random(double2* p, u64 n, u64 seed) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double re = (double)(splitmix64(seed + 2 * i) >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    const double im = (double)(splitmix64(seed + 2 * i + 1) >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    p[i] = make_double2(re, im);
  }
}

__global__ void k_scale(double2* p, u64 n, double s) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double2 v = p[i];
    p[i] = make_double2(v.x * s, v.y * s);
  }
}

__global__ void k_copy(double2* __restrict__ dst, const double2* __restrict__ src, u64 n) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// dst[j] = src[insert(j, bit, value)]  /  inverse
__global__ void k_pack_half(double2* __restrict__ dst, const double2* __restrict__ src, u64 n_half,
                            int bit, u64 value_off) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_half; j += stride) {
    const u64 i = (((j >> bit) << (bit + 1)) | (j & ((1ull << bit) - 1))) | value_off;
    dst[j] = src[i];
  }
}
__global__ void k_unpack_half(double2* __restrict__ dst, const double2* __restrict__ src, u64 n_half,
                              int bit, u64 value_off) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_half; j += stride) {
    const u64 i = (((j >> bit) << (bit + 1)) | (j & ((1ull << bit) - 1))) | value_off;
    dst[i] = src[j];
  }
}

// slab gather / scatter for the all-to-all re-layout: j runs over the 2^(k-m) amplitudes whose
// bits `pos` equal the pattern folded into value_off
__global__ void k_pack_bits(double2* __restrict__ dst, const double2* __restrict__ src, u64 n_slab,
                            int npos, int p0, int p1, int p2, u64 value_off) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_slab; j += stride)
    dst[j] = ld_amp<true>(src + (expand_index(j, npos, p0, p1, p2) | value_off));
}
__global__ void k_unpack_bits(double2* __restrict__ dst, const double2* __restrict__ src, u64 n_slab,
                              int npos, int p0, int p1, int p2, u64 value_off) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_slab; j += stride)
    dst[expand_index(j, npos, p0, p1, p2) | value_off] = ld_amp<true>(src + j);
}

// exchange slab (bits pos == a_off pattern) of chunk A with slab (bits pos == b_off pattern) of chunk B
__global__ void k_swap_slabs(double2* __restrict__ a, double2* __restrict__ b, u64 n_slab,
                             int npos, int p0, int p1, int p2, u64 a_off, u64 b_off) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_slab; j += stride) {
    const u64 e = expand_index(j, npos, p0, p1, p2);
    const double2 x = a[e | a_off], y = b[e | b_off];
    a[e | a_off] = y;
    b[e | b_off] = x;
  }
}

constexpr int kReduceBlocks = 2048;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

template <bool MAX>
__device__ __forceinline__ void block_reduce_store(double v, double* out) {
  __shared__ double part[kBlock / 64];
  v = MAX ? wave_max(v) : wave_sum(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = part[0];
    for (int w = 1; w < kBlock / 64; ++w) r = MAX ? fmax(r, part[w]) : r + part[w];
    out[blockIdx.x] = r;
  }
}

__global__ __launch_bounds__(kBlock) void k_norm2_partial(const double2* p, u64 n, double* partial) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double2 v = p[i];
    acc = fma(v.x, v.x, fma(v.y, v.y, acc));
  }
  block_reduce_store<false>(acc, partial);
}

// kind 0: GHZ, kind 1: GHZ+QFT closed form.  `base` = global index of amp 0.
struct BitPerm { unsigned char to_logical[64]; int active; };   // physical index bit -> logical qubit

__global__ __launch_bounds__(kBlock) void k_closed_form_err(const double2* p, u64 n, int kind,
                                                            int n_total, u64 base, double* partial,
                                                            const BitPerm perm) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  const double inv_n = exp2(-(double)n_total);
  const double amp = exp2(-0.5 * (double)(n_total + 1));
  const u64 last = (n_total >= 64) ? ~0ull : ((1ull << n_total) - 1);
  double worst = 0.0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u64 y = base + i;
    if (perm.active) {                       // staged layout: logical index from the physical one
      const u64 x = y;
      y = 0;
      for (int b = 0; b < n_total; ++b) y |= ((x >> b) & 1ull) << perm.to_logical[b];
    }
    double er, ei;
    if (kind == 0) {
      er = (y == 0 || y == last) ? 0.70710678118654752440 : 0.0;
      ei = 0.0;
    } else {
      // exp(-2 pi i y / 2^n): y * 2^-n is exact in double for n <= 52
      double s, c;
      sincospi(-2.0 * ((double)y * inv_n), &s, &c);
      er = amp * (1.0 + c);
      ei = amp * s;
    }
    const double2 v = p[i];
    worst = fmax(worst, hypot(v.x - er, v.y - ei));
  }
  block_reduce_store<true>(worst, partial);
}

static int ensure_scratch(qsim_chunk* c) {
  if (!c->scratch) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc((void**)&c->scratch, sizeof(double) * kReduceBlocks));
  }
  return QSIM_OK;
}

static unsigned stream_grid(u64 n) {
  const u64 want = (n + kBlock - 1) / kBlock;
  return (unsigned)std::min<u64>(std::max<u64>(want, 1), 8192);
}

// ------------------------------------------------------------------ C ABI
extern "C" {

const char* qsim_last_error(void) { return g_err.c_str(); }
int qsim_version(void) { return 100; }

int qsim_device_count(int* count) {
  if (!count) return fail(QSIM_ERR_INVALID, "count is null");
  HIP_TRY(hipGetDeviceCount(count));
  return QSIM_OK;
}

static qsim_chunk* new_chunk() {
  qsim_chunk* c = new qsim_chunk();
  std::memset(c, 0, sizeof *c);
  return c;
}

int qsim_create(int device, int n_local_qubits, qsim_chunk** out) {
  if (!out) return fail(QSIM_ERR_INVALID, "out is null");
  if (n_local_qubits < 0 || n_local_qubits > 40)
    return fail(QSIM_ERR_INVALID, "n_local_qubits %d out of range [0, 40]", n_local_qubits);
  hipStream_t s;
  int rc = device_stream(device, &s);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  double2* p = nullptr;
  HIP_TRY(hipMalloc((void**)&p, sizeof(double2) << n_local_qubits));
  qsim_chunk* c = new_chunk();
  c->device = device;
  c->k = n_local_qubits;
  c->amp = p;
  c->stream = s;
  c->owns_memory = true;
  c->span_bytes = sizeof(double2) << n_local_qubits;
  *out = c;
  return QSIM_OK;
}

int qsim_create_view(qsim_chunk* parent, uint64_t offset_amps, int n_local_qubits, qsim_chunk** out) {
  int rc = check_chunk(parent, "qsim_create_view");
  if (rc) return rc;
  if (!out) return fail(QSIM_ERR_INVALID, "out is null");
  if (n_local_qubits < 0 || n_local_qubits > parent->k)
    return fail(QSIM_ERR_INVALID, "view of %d qubits does not fit a %d-qubit chunk", n_local_qubits, parent->k);
  const u64 len = 1ull << n_local_qubits;
  if (offset_amps % len != 0 || offset_amps + len > amps(parent))
    return fail(QSIM_ERR_INVALID, "view offset %llu not aligned/inside parent", (u64)offset_amps);
  qsim_chunk* c = new_chunk();
  c->device = parent->device;
  c->k = n_local_qubits;
  c->amp = parent->amp + offset_amps;
  c->stream = parent->stream;
  c->owns_memory = false;
  c->parent = parent;
  c->span_bytes = parent->span_bytes;
  *out = c;
  return QSIM_OK;
}

int qsim_wrap(int device, void* device_ptr, int n_local_qubits, void* stream, qsim_chunk** out) {
  if (!out || !device_ptr) return fail(QSIM_ERR_INVALID, "null pointer");
  if (n_local_qubits < 0 || n_local_qubits > 40) return fail(QSIM_ERR_INVALID, "n_local_qubits out of range");
  if (((uintptr_t)device_ptr & 15) != 0) return fail(QSIM_ERR_INVALID, "device pointer must be 16-byte aligned");
  qsim_chunk* c = new_chunk();
  c->device = device;
  c->k = n_local_qubits;
  c->amp = (double2*)device_ptr;
  c->stream = (hipStream_t)stream;
  c->owns_memory = false;
  c->span_bytes = sizeof(double2) << n_local_qubits;
  *out = c;
  return QSIM_OK;
}

int qsim_destroy(qsim_chunk* c) {
  if (!c) return QSIM_OK;
  (void)hipSetDevice(c->device);
  if (c->have_events) { (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1); }
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->owns_memory && c->amp) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->amp);
  }
  delete c;
  return QSIM_OK;
}

int qsim_n_local_qubits(const qsim_chunk* c) { return c ? c->k : -1; }
void* qsim_device_ptr(const qsim_chunk* c) { return c ? (void*)c->amp : nullptr; }

int qsim_init_zero(qsim_chunk* c, int set_amp0) {
  int rc = check_chunk(c, "qsim_init_zero");
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(k_fill_zero, dim3(stream_grid(amps(c))), dim3(kBlock), 0, c->stream, c->amp, amps(c), set_amp0);
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

int qsim_norm2(qsim_chunk* c, double* out) {
  int rc = check_chunk(c, "qsim_norm2");
  if (rc) return rc;
  if (!out) return fail(QSIM_ERR_INVALID, "out is null");
  if ((rc = ensure_scratch(c))) return rc;
  HIP_TRY(hipSetDevice(c->device));
  const unsigned grid = std::min<unsigned>(stream_grid(amps(c)), kReduceBlocks);
  hipLaunchKernelGGL(k_norm2_partial, dim3(grid), dim3(kBlock), 0, c->stream, c->amp, amps(c), c->scratch);
  HIP_TRY(hipGetLastError());
  std::vector<double> host(grid);
  HIP_TRY(hipMemcpyAsync(host.data(), c->scratch, sizeof(double) * grid, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  long double total = 0;
  for (double v : host) total += v;
  *out = (double)total;
  return QSIM_OK;
}

int qsim_init_random(qsim_chunk* c, uint64_t seed) {
  int rc = check_chunk(c, "qsim_init_random");
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(k_fill_random, dim3(stream_grid(amps(c))), dim3(kBlock), 0, c->stream, c->amp, amps(c), (u64)seed);
  HIP_TRY(hipGetLastError());
  double n2 = 0;
  if ((rc = qsim_norm2(c, &n2))) return rc;
  if (!(n2 > 0)) return fail(QSIM_ERR_INVALID, "random state has zero norm");
  hipLaunchKernelGGL(k_scale, dim3(stream_grid(amps(c))), dim3(kBlock), 0, c->stream, c->amp, amps(c), 1.0 / std::sqrt(n2));
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

int qsim_upload(qsim_chunk* c, const double* re_im, uint64_t offset_amps, uint64_t count) {
  int rc = check_chunk(c, "qsim_upload");
  if (rc) return rc;
  if (!re_im && count) return fail(QSIM_ERR_INVALID, "host buffer is null");
  if (offset_amps > amps(c) || count > amps(c) - offset_amps)
    return fail(QSIM_ERR_INVALID, "upload range [%llu, +%llu) outside chunk of %llu", (u64)offset_amps, (u64)count, amps(c));
  if (!count) return QSIM_OK;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(c->amp + offset_amps, re_im, count * sizeof(double2), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QSIM_OK;
}

int qsim_download(qsim_chunk* c, double* re_im, uint64_t offset_amps, uint64_t count) {
  int rc = check_chunk(c, "qsim_download");
  if (rc) return rc;
  if (!re_im && count) return fail(QSIM_ERR_INVALID, "host buffer is null");
  if (offset_amps > amps(c) || count > amps(c) - offset_amps)
    return fail(QSIM_ERR_INVALID, "download range [%llu, +%llu) outside chunk of %llu", (u64)offset_amps, (u64)count, amps(c));
  if (!count) return QSIM_OK;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(re_im, c->amp + offset_amps, count * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QSIM_OK;
}

int qsim_copy(qsim_chunk* dst, const qsim_chunk* src) {
  int rc = check_chunk(dst, "qsim_copy");
  if (rc || (rc = check_chunk(src, "qsim_copy"))) return rc;
  if (dst->k != src->k) return fail(QSIM_ERR_INVALID, "qsim_copy: sizes differ");
  if (dst->amp == src->amp) return QSIM_OK;
  HIP_TRY(hipSetDevice(dst->device));
  hipLaunchKernelGGL(k_copy, dim3(stream_grid(amps(dst))), dim3(kBlock), 0, dst->stream, dst->amp, src->amp, amps(dst));
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

int qsim_apply_1q(qsim_chunk* c, int qubit, const double U[8]) {
  int rc = check_chunk(c, "qsim_apply_1q");
  if (rc || (rc = check_local_qubit(c, qubit))) return rc;
  if (!U) return fail(QSIM_ERR_INVALID, "U is null");
  HIP_TRY(hipSetDevice(c->device));
  Group g = {{c, nullptr, nullptr, nullptr}, 1, c->k};
  return gate_1q(g, qubit, U, c->stream);
}

int qsim_apply_2q(qsim_chunk* c, int qa, int qb, const double U[32]) {
  int rc = check_chunk(c, "qsim_apply_2q");
  if (rc || (rc = check_local_qubit(c, qa)) || (rc = check_local_qubit(c, qb))) return rc;
  if (qa == qb) return fail(QSIM_ERR_INVALID, "apply_2q needs two distinct qubits, got %d twice", qa);
  if (!U) return fail(QSIM_ERR_INVALID, "U is null");
  HIP_TRY(hipSetDevice(c->device));
  Group g = {{c, nullptr, nullptr, nullptr}, 1, c->k};
  return gate_2q(g, qa, qb, U, c->stream);
}

static int validate_ops(qsim_chunk* c, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats) {
  int rc = check_chunk(c, "qsim_apply_ops");
  if (rc) return rc;
  if (n_ops < 0 || (n_ops && (!nq || !qubits || !mats))) return fail(QSIM_ERR_INVALID, "bad op list");
  for (int i = 0; i < n_ops; ++i) {  // validate everything before the first launch
    if (nq[i] != 1 && nq[i] != 2) return fail(QSIM_ERR_INVALID, "op %d: arity %d", i, nq[i]);
    for (int j = 0; j < nq[i]; ++j)
      if ((rc = check_local_qubit(c, qubits[2 * i + j]))) return rc;
    if (nq[i] == 2 && qubits[2 * i] == qubits[2 * i + 1])
      return fail(QSIM_ERR_INVALID, "op %d: repeated qubit", i);
  }
  return QSIM_OK;
}

int qsim_apply_ops_unfused(qsim_chunk* c, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats) {
  int rc = validate_ops(c, n_ops, nq, qubits, mats);
  if (rc) return rc;
  for (int i = 0; i < n_ops; ++i) {
    rc = nq[i] == 1 ? qsim_apply_1q(c, qubits[2 * i], mats + 32 * (size_t)i)
                    : qsim_apply_2q(c, qubits[2 * i], qubits[2 * i + 1], mats + 32 * (size_t)i);
    if (rc) return rc;
  }
  return QSIM_OK;
}

int qsim_apply_ops(qsim_chunk* c, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats) {
  int rc = validate_ops(c, n_ops, nq, qubits, mats);
  if (rc) return rc;
  if (n_ops < 2 || c->k < kTileMinChunk) {
    c->last_passes = n_ops;
    return qsim_apply_ops_unfused(c, n_ops, nq, qubits, mats);
  }
  HIP_TRY(hipSetDevice(c->device));
  std::vector<FusedOp> ops;
  ops.reserve(n_ops);
  for (int i = 0; i < n_ops; ++i) {
    FusedOp o;
    if (classify_op(nq[i], qubits + 2 * i, mats + 32 * (size_t)i, &o)) ops.push_back(o);
  }
  int passes = 0;
  rc = run_fused(c, ops, &passes);
  c->last_passes = passes;
  return rc;
}

int qsim_last_pass_count(const qsim_chunk* c) { return c ? c->last_passes : -1; }

int qsim_apply_1q_pair(qsim_chunk* c0, qsim_chunk* c1, const double U[8]) {
  qsim_chunk* cs[2] = {c0, c1};
  int rc = check_group(cs, 2, "qsim_apply_1q_pair");
  if (rc) return rc;
  if (!U) return fail(QSIM_ERR_INVALID, "U is null");
  HIP_TRY(hipSetDevice(c0->device));
  Group g = {{c0, c1, nullptr, nullptr}, 2, c0->k};
  return gate_1q(g, c0->k, U, c0->stream);
}

int qsim_apply_2q_pair_qa_local(qsim_chunk* c0, qsim_chunk* c1, int qa, const double U[32]) {
  qsim_chunk* cs[2] = {c0, c1};
  int rc = check_group(cs, 2, "qsim_apply_2q_pair_qa_local");
  if (rc || (rc = check_local_qubit(c0, qa))) return rc;
  if (!U) return fail(QSIM_ERR_INVALID, "U is null");
  HIP_TRY(hipSetDevice(c0->device));
  Group g = {{c0, c1, nullptr, nullptr}, 2, c0->k};
  return gate_2q(g, qa, c0->k, U, c0->stream);
}

int qsim_apply_2q_pair_qb_local(qsim_chunk* c0, qsim_chunk* c1, int qb, const double U[32]) {
  qsim_chunk* cs[2] = {c0, c1};
  int rc = check_group(cs, 2, "qsim_apply_2q_pair_qb_local");
  if (rc || (rc = check_local_qubit(c0, qb))) return rc;
  if (!U) return fail(QSIM_ERR_INVALID, "U is null");
  HIP_TRY(hipSetDevice(c0->device));
  Group g = {{c0, c1, nullptr, nullptr}, 2, c0->k};
  return gate_2q(g, c0->k, qb, U, c0->stream);
}

int qsim_apply_2q_quad(qsim_chunk* c00, qsim_chunk* c01, qsim_chunk* c10, qsim_chunk* c11, const double U[32]) {
  qsim_chunk* cs[4] = {c00, c01, c10, c11};
  int rc = check_group(cs, 4, "qsim_apply_2q_quad");
  if (rc) return rc;
  if (!U) return fail(QSIM_ERR_INVALID, "U is null");
  HIP_TRY(hipSetDevice(c00->device));
  // chunk index = 2*bit(qa) + bit(qb): qb is virtual bit k, qa is virtual bit k+1
  Group g = {{c00, c01, c10, c11}, 4, c00->k};
  return gate_2q(g, c00->k + 1, c00->k, U, c00->stream);
}

int qsim_pack_half(const qsim_chunk* src, int bit, int value, qsim_chunk* buf) {
  int rc = check_chunk(src, "qsim_pack_half");
  if (rc || (rc = check_chunk(buf, "qsim_pack_half"))) return rc;
  if (bit < 0 || bit >= src->k || buf->k != src->k - 1 || (value != 0 && value != 1))
    return fail(QSIM_ERR_INVALID, "qsim_pack_half: bit %d / buffer size mismatch", bit);
  HIP_TRY(hipSetDevice(src->device));
  const u64 n_half = amps(buf);
  hipLaunchKernelGGL(k_pack_half, dim3(stream_grid(n_half)), dim3(kBlock), 0, src->stream,
                     buf->amp, (const double2*)src->amp, n_half, bit, value ? (1ull << bit) : 0ull);
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

int qsim_unpack_half(qsim_chunk* dst, int bit, int value, const qsim_chunk* buf) {
  int rc = check_chunk(dst, "qsim_unpack_half");
  if (rc || (rc = check_chunk(buf, "qsim_unpack_half"))) return rc;
  if (bit < 0 || bit >= dst->k || buf->k != dst->k - 1 || (value != 0 && value != 1))
    return fail(QSIM_ERR_INVALID, "qsim_unpack_half: bit %d / buffer size mismatch", bit);
  HIP_TRY(hipSetDevice(dst->device));
  const u64 n_half = amps(buf);
  hipLaunchKernelGGL(k_unpack_half, dim3(stream_grid(n_half)), dim3(kBlock), 0, dst->stream,
                     dst->amp, (const double2*)buf->amp, n_half, bit, value ? (1ull << bit) : 0ull);
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

static int slab_args(const qsim_chunk* c, int m, const int32_t* bits, int pattern, const qsim_chunk* buf,
                     uint64_t buf_offset, int pos[3], u64* value_off, u64* n_slab) {
  if (m < 1 || m > 3 || !bits) return fail(QSIM_ERR_INVALID, "slab: 1..3 bits expected, got %d", m);
  if (m > c->k) return fail(QSIM_ERR_INVALID, "slab: more bits than the chunk has");
  if (pattern < 0 || pattern >= (1 << m)) return fail(QSIM_ERR_INVALID, "slab: pattern out of range");
  int sorted[3] = {0, 0, 0};
  *value_off = 0;
  for (int i = 0; i < m; ++i) {
    if (bits[i] < 0 || bits[i] >= c->k) return fail(QSIM_ERR_INVALID, "slab: bit %d out of range", bits[i]);
    for (int j = 0; j < i; ++j)
      if (bits[j] == bits[i]) return fail(QSIM_ERR_INVALID, "slab: repeated bit %d", bits[i]);
    sorted[i] = bits[i];
    if ((pattern >> i) & 1) *value_off |= 1ull << bits[i];
  }
  std::sort(sorted, sorted + m);
  for (int i = 0; i < 3; ++i) pos[i] = sorted[i];
  *n_slab = 1ull << (c->k - m);
  if (buf_offset > amps(buf) || *n_slab > amps(buf) - buf_offset)
    return fail(QSIM_ERR_INVALID, "slab: buffer range outside the buffer chunk");
  return QSIM_OK;
}

int qsim_pack_bits(const qsim_chunk* src, int m, const int32_t* bits, int pattern, qsim_chunk* buf,
                   uint64_t buf_offset_amps) {
  int rc = check_chunk(src, "qsim_pack_bits");
  if (rc || (rc = check_chunk(buf, "qsim_pack_bits"))) return rc;
  int pos[3];
  u64 voff, n_slab;
  if ((rc = slab_args(src, m, bits, pattern, buf, buf_offset_amps, pos, &voff, &n_slab))) return rc;
  HIP_TRY(hipSetDevice(src->device));
  hipLaunchKernelGGL(k_pack_bits, dim3(stream_grid(n_slab)), dim3(kBlock), 0, src->stream,
                     buf->amp + buf_offset_amps, (const double2*)src->amp, n_slab, m, pos[0], pos[1], pos[2], voff);
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

int qsim_unpack_bits(qsim_chunk* dst, int m, const int32_t* bits, int pattern, const qsim_chunk* buf,
                     uint64_t buf_offset_amps) {
  int rc = check_chunk(dst, "qsim_unpack_bits");
  if (rc || (rc = check_chunk(buf, "qsim_unpack_bits"))) return rc;
  int pos[3];
  u64 voff, n_slab;
  if ((rc = slab_args(dst, m, bits, pattern, buf, buf_offset_amps, pos, &voff, &n_slab))) return rc;
  HIP_TRY(hipSetDevice(dst->device));
  hipLaunchKernelGGL(k_unpack_bits, dim3(stream_grid(n_slab)), dim3(kBlock), 0, dst->stream,
                     dst->amp, (const double2*)buf->amp + buf_offset_amps, n_slab, m, pos[0], pos[1], pos[2], voff);
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

int qsim_swap_global_local(qsim_chunk* const* chunks, int n_chunks, const int32_t* global_bits,
                           const int32_t* local_bits, int m) {
  if (!chunks || !global_bits || !local_bits) return fail(QSIM_ERR_INVALID, "qsim_swap_global_local: null argument");
  if (m < 1 || m > 3) return fail(QSIM_ERR_INVALID, "qsim_swap_global_local: 1..3 qubit pairs expected, got %d", m);
  if (n_chunks < 2 || (n_chunks & (n_chunks - 1)) || n_chunks > 4096)
    return fail(QSIM_ERR_INVALID, "qsim_swap_global_local: chunk count must be a power of two >= 2");
  int rc = QSIM_OK;
  for (int i = 0; i < n_chunks; ++i) {
    if ((rc = check_chunk(chunks[i], "qsim_swap_global_local"))) return rc;
    if (chunks[i]->k != chunks[0]->k || chunks[i]->device != chunks[0]->device)
      return fail(QSIM_ERR_INVALID, "qsim_swap_global_local: chunks differ in size or device");
  }
  const int k = chunks[0]->k;
  int g_bits = 0;
  while ((1 << g_bits) < n_chunks) ++g_bits;
  int sorted[3] = {0, 0, 0};
  for (int i = 0; i < m; ++i) {
    if (local_bits[i] < 0 || local_bits[i] >= k) return fail(QSIM_ERR_NONLOCAL, "qsim_swap_global_local: local bit %d is non-local for 2^%d chunks", local_bits[i], k);
    if (global_bits[i] < 0 || global_bits[i] >= g_bits) return fail(QSIM_ERR_INVALID, "qsim_swap_global_local: chunk-index bit %d out of range", global_bits[i]);
    for (int j = 0; j < i; ++j)
      if (local_bits[j] == local_bits[i] || global_bits[j] == global_bits[i])
        return fail(QSIM_ERR_INVALID, "qsim_swap_global_local: repeated bit");
    sorted[i] = local_bits[i];
  }
  std::sort(sorted, sorted + m);
  HIP_TRY(hipSetDevice(chunks[0]->device));
  const u64 n_slab = 1ull << (k - m);
  auto local_offset = [&](int pattern) {
    u64 off = 0;
    for (int i = 0; i < m; ++i) if ((pattern >> i) & 1) off |= 1ull << local_bits[i];
    return off;
  };
  for (int c = 0; c < n_chunks; ++c) {
    int mine = 0;                                   // pattern of this chunk's swapped index bits
    for (int i = 0; i < m; ++i) mine |= ((c >> global_bits[i]) & 1) << i;
    for (int d = 0; d < (1 << m); ++d) {
      if (d == mine) continue;
      int peer = c;
      for (int i = 0; i < m; ++i) peer = (peer & ~(1 << global_bits[i])) | (((d >> i) & 1) << global_bits[i]);
      if (peer < c) continue;                       // each unordered pair once
      hipLaunchKernelGGL(k_swap_slabs, dim3(stream_grid(n_slab)), dim3(kBlock), 0, chunks[0]->stream,
                         chunks[c]->amp, chunks[peer]->amp, n_slab, m, sorted[0], sorted[1], sorted[2],
                         local_offset(d), local_offset(mine));
      HIP_TRY(hipGetLastError());
    }
  }
  return QSIM_OK;
}

int qsim_sync(qsim_chunk* c) {
  int rc = check_chunk(c, "qsim_sync");
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QSIM_OK;
}

int qsim_max_abs_err_closed_form_perm(qsim_chunk* c, int kind, int n_total_qubits, uint64_t base_index,
                                      const int32_t* log_to_phys, double* out);

int qsim_max_abs_err_closed_form(qsim_chunk* c, int kind, int n_total_qubits, uint64_t base_index, double* out) {
  return qsim_max_abs_err_closed_form_perm(c, kind, n_total_qubits, base_index, nullptr, out);
}

int qsim_max_abs_err_closed_form_perm(qsim_chunk* c, int kind, int n_total_qubits, uint64_t base_index,
                                      const int32_t* log_to_phys, double* out) {
  int rc = check_chunk(c, "qsim_max_abs_err_closed_form");
  if (rc) return rc;
  if (!out || (kind != 0 && kind != 1) || n_total_qubits < c->k || n_total_qubits > 52)
    return fail(QSIM_ERR_INVALID, "qsim_max_abs_err_closed_form: bad arguments");
  BitPerm perm;
  std::memset(&perm, 0, sizeof perm);
  if (log_to_phys) {
    u64 seen = 0;
    for (int q = 0; q < n_total_qubits; ++q) {
      const int ph = log_to_phys[q];
      if (ph < 0 || ph >= n_total_qubits || (seen >> ph) & 1)
        return fail(QSIM_ERR_INVALID, "qsim_max_abs_err_closed_form: log_to_phys is not a permutation");
      seen |= 1ull << ph;
      perm.to_logical[ph] = (unsigned char)q;
      if (ph != q) perm.active = 1;
    }
  }
  if ((rc = ensure_scratch(c))) return rc;
  HIP_TRY(hipSetDevice(c->device));
  const unsigned grid = std::min<unsigned>(stream_grid(amps(c)), kReduceBlocks);
  hipLaunchKernelGGL(k_closed_form_err, dim3(grid), dim3(kBlock), 0, c->stream, c->amp, amps(c), kind,
                     n_total_qubits, (u64)base_index, c->scratch, perm);
  HIP_TRY(hipGetLastError());
  std::vector<double> host(grid);
  HIP_TRY(hipMemcpyAsync(host.data(), c->scratch, sizeof(double) * grid, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  double worst = 0;
  for (double v : host) worst = std::max(worst, v);
  *out = worst;
  return QSIM_OK;
}

static int ensure_events(qsim_chunk* c) {
  if (!c->have_events) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    c->have_events = true;
  }
  return QSIM_OK;
}

int qsim_time_begin(qsim_chunk* c) {
  int rc = check_chunk(c, "qsim_time_begin");
  if (rc || (rc = ensure_events(c))) return rc;
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  return QSIM_OK;
}

int qsim_time_end(qsim_chunk* c, float* elapsed_ms) {
  int rc = check_chunk(c, "qsim_time_end");
  if (rc || (rc = ensure_events(c))) return rc;
  if (!elapsed_ms) return fail(QSIM_ERR_INVALID, "elapsed_ms is null");
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
  return QSIM_OK;
}

int qsim_profile_begin(qsim_chunk* c) {
  int rc = check_chunk(c, "qsim_profile_begin");
  if (rc) return rc;
  if (g_prof.open) return fail(QSIM_ERR_INVALID, "a profile is already open");
  HIP_TRY(hipSetDevice(c->device));
  g_prof.records.clear();
  g_prof.stream = c->stream;
  g_prof.open = true;
  return QSIM_OK;
}

int qsim_profile_end(qsim_chunk* c, int max_entries, int* n_entries, qsim_profile_entry* out) {
  int rc = check_chunk(c, "qsim_profile_end");
  if (rc) return rc;
  if (!g_prof.open) return fail(QSIM_ERR_INVALID, "no profile is open");
  if (!n_entries || (!out && max_entries > 0)) return fail(QSIM_ERR_INVALID, "null output");
  g_prof.open = false;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  uint64_t launches[kNumClasses] = {0};
  double ms[kNumClasses] = {0}, bytes[kNumClasses] = {0}, hbm[kNumClasses] = {0};
  for (LaunchRecord& r : g_prof.records) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, r.e0, r.e1));
    launches[r.cls] += 1;
    ms[r.cls] += t;
    bytes[r.cls] += r.bytes;
    hbm[r.cls] += r.hbm_bytes;
    g_prof.pool.push_back(r.e0);
    g_prof.pool.push_back(r.e1);
  }
  g_prof.records.clear();
  int n = 0;
  for (int cls = 0; cls < kNumClasses; ++cls) {
    if (!launches[cls]) continue;
    if (n < max_entries) {
      std::snprintf(out[n].kernel, sizeof out[n].kernel, "%s", kClassNames[cls]);
      out[n].launches = launches[cls];
      out[n].total_ms = ms[cls];
      out[n].algorithmic_bytes = bytes[cls];
      out[n].hbm_bytes = hbm[cls];
    }
    ++n;
  }
  *n_entries = n;
  return QSIM_OK;
}

}  // extern "C"
