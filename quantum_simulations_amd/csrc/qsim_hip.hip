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
#include <atomic>
#include <list>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/qsim_hip.h"

typedef unsigned long long u64;

// The sources are split by topic and compiled as ONE translation unit (kernels, their launchers and
// the C ABI share file-local state; hipcc needs no relocatable device code this way).
#include "qsim_core.h"
#include "gate_kernels.h"
#include "gate_plan.h"
#include "tile_kernel.h"
#include "tile_planner.h"
#include "misc_kernels.h"
#include "comm_rccl.h"

// An op list whose SOURCE arrives in pieces (qsim_ops_io::src_parts: the receive side of a fused re-layout): planned and
// prepared at the call, launched as the pieces are announced (qsim_apply_ops_io_load) -- the first pass as partial launches
// over the tiles whose source pieces are there, the rest when the source is complete.
struct DeferredIo {
  bool active = false;
  bool tiles = false;               // tile passes (else: chunk too small / no ops: gate by gate after the source is complete)
  int n_ops = 0;
  std::vector<int32_t> nq, qubits;  // the op list (kept for the gate-by-gate case)
  std::vector<double> mats;
  qsim_ops_io io;
  FusedIo fio;
  std::vector<CachedPass> passes;   // prepared passes (buffers in)
  int nb = 0;                       // 2^nb source pieces: the top nb index bits that are no source slab bits
  int piece_bit[3] = {0, 0, 0};     // ascending
  int nb_free = 0;                  // the first pass runs as 2^nb_free partial launches (0: whole, after the last piece)
  unsigned announced = 0, launched = 0;
};

// ------------------------------------------------------------------ C ABI
extern "C" {

const char* qsim_last_error(void) { return g_err.c_str(); }
int qsim_version(void) { return 100; }

int qsim_device_count(int* count) {
  if (!count) return fail(QSIM_ERR_INVALID, "count is null");
  HIP_TRY(hipGetDeviceCount(count));
  return QSIM_OK;
}

static bool parts_pending(const qsim_chunk* c) {
  return (c->pending && c->pending->mode != PendingLast::kNone) || (c->deferred && c->deferred->active);
}
// (error paths and re-initialisation: the chunk's contents are unspecified while pieces are pending, so dropping them loses nothing)
static void drop_pending(qsim_chunk* c) {
  if (c->pending) c->pending->mode = PendingLast::kNone;
  if (c->deferred) c->deferred->active = false;
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
  delete c->pending;
  delete c->deferred;
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
  drop_pending(c);
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

// complex64 transfers (the reference's chunk-file dtype): converted on the device through a staging buffer
static int c64_transfer(qsim_chunk* c, float* host, uint64_t offset_amps, uint64_t count, bool download, const char* what) {
  int rc = check_chunk(c, what);
  if (rc) return rc;
  if (!host && count) return fail(QSIM_ERR_INVALID, "%s: host buffer is null", what);
  if (offset_amps > amps(c) || count > amps(c) - offset_amps)
    return fail(QSIM_ERR_INVALID, "%s: range [%llu, +%llu) outside chunk of %llu", what, (u64)offset_amps, (u64)count, amps(c));
  if (!count) return QSIM_OK;
  HIP_TRY(hipSetDevice(c->device));
  const u64 piece = std::min<u64>(count, 1ull << 24);
  float2* stage = nullptr;
  if (hipMalloc((void**)&stage, sizeof(float2) * piece) != hipSuccess) return fail(QSIM_ERR_NOMEM, "%s: no device memory for the staging buffer", what);
  hipError_t e = hipSuccess;
  for (u64 done = 0; done < count && e == hipSuccess; done += piece) {
    const u64 n = std::min<u64>(piece, count - done);
    if (download) {
      hipLaunchKernelGGL(k_to_c64, dim3(stream_grid(n)), dim3(kBlock), 0, c->stream, stage, (const double2*)(c->amp + offset_amps + done), n);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(host + 2 * done, stage, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream);
    } else {
      e = hipMemcpyAsync(stage, host + 2 * done, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_from_c64, dim3(stream_grid(n)), dim3(kBlock), 0, c->stream, c->amp + offset_amps + done, (const float2*)stage, n);
        e = hipGetLastError();
      }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // (the staging buffer is reused)
  }
  (void)hipFree(stage);
  if (e != hipSuccess) return fail(QSIM_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return QSIM_OK;
}
int qsim_download_c64(qsim_chunk* c, float* re_im, uint64_t offset_amps, uint64_t count) {
  return c64_transfer(c, re_im, offset_amps, count, true, "qsim_download_c64");
}
int qsim_upload_c64(qsim_chunk* c, const float* re_im, uint64_t offset_amps, uint64_t count) {
  return c64_transfer(c, const_cast<float*>(re_im), offset_amps, count, false, "qsim_upload_c64");
}

int qsim_copy(qsim_chunk* dst, const qsim_chunk* src) {
  int rc = check_chunk(dst, "qsim_copy");
  if (rc || (rc = check_chunk(src, "qsim_copy"))) return rc;
  if (dst->k != src->k) return fail(QSIM_ERR_INVALID, "qsim_copy: sizes differ");
  if (dst->amp == src->amp) return QSIM_OK;
  HIP_TRY(hipSetDevice(dst->device));
  constexpr int kItems = 2;
  u64 blocks = (amps(dst) + (u64)kBlock * kItems - 1) / ((u64)kBlock * kItems);
  blocks = (blocks + 7) & ~7ull;                     // whole octets: logical_block<true> deals blocks over the 8 XCDs
  if ((sizeof(double2) << dst->k) > tuning().mall_bytes)
    hipLaunchKernelGGL((k_copy<true, kItems>), grid_for(blocks), dim3(kBlock), 0, dst->stream, dst->amp, src->amp, amps(dst));
  else
    hipLaunchKernelGGL((k_copy<false, kItems>), grid_for(blocks), dim3(kBlock), 0, dst->stream, dst->amp, src->amp, amps(dst));
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

// The streaming candidates behind bench.py's `stream_ceiling`: variant 0 = qsim_copy's own choice, 1 = the non-temporal
// copy kernel whatever the size, 2 = the plain (cached) copy kernel, 3 = hipMemcpyAsync device to device (the runtime's
// blit kernel).  Measurement aid: same arguments and stream semantics as qsim_copy.
int qsim_copy_variant(qsim_chunk* dst, const qsim_chunk* src, int variant) {
  if (variant == 0) return qsim_copy(dst, src);
  int rc = check_chunk(dst, "qsim_copy_variant");
  if (rc || (rc = check_chunk(src, "qsim_copy_variant"))) return rc;
  if (dst->k != src->k) return fail(QSIM_ERR_INVALID, "qsim_copy_variant: sizes differ");
  if (dst->amp == src->amp) return fail(QSIM_ERR_INVALID, "qsim_copy_variant: source and destination are the same buffer");
  HIP_TRY(hipSetDevice(dst->device));
  constexpr int kItems = 2;
  u64 blocks = (amps(dst) + (u64)kBlock * kItems - 1) / ((u64)kBlock * kItems);
  blocks = (blocks + 7) & ~7ull;
  if (variant == 1) hipLaunchKernelGGL((k_copy<true, kItems>), grid_for(blocks), dim3(kBlock), 0, dst->stream, dst->amp, src->amp, amps(dst));
  else if (variant == 2) hipLaunchKernelGGL((k_copy<false, kItems>), grid_for(blocks), dim3(kBlock), 0, dst->stream, dst->amp, src->amp, amps(dst));
  else if (variant == 3) HIP_TRY(hipMemcpyAsync(dst->amp, src->amp, sizeof(double2) << dst->k, hipMemcpyDeviceToDevice, dst->stream));
  else return fail(QSIM_ERR_INVALID, "qsim_copy_variant: variant %d", variant);
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
  if (parts_pending(c)) return fail(QSIM_ERR_INVALID, "qsim_apply_ops: slab pieces of a split qsim_apply_ops_io call are pending on this chunk");
  if (n_ops < 2 || c->k < kTileMinChunk || c->k > kTileMaxQubits) {
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
  rc = run_fused(c, ops, &passes, nullptr, n_ops, nq, qubits, mats);
  c->last_passes = passes;
  return rc;
}

// Dense k-qubit block: new[idx with the block's bits = out] = sum_in M[out][in] old[idx with the block's bits = in], pattern
// bit i <-> qubits[i] -- v3's `_apply_combined_matrix` (parallel_gate_applicator.py:315-385) for a genuinely dense 2^k x 2^k
// matrix (its tensor-product blocks are cheaper as butterflies inside a fused pass: qsim_apply_ops).  1 <= k <= 6.
// k = 1, 2: the pair kernels.  k >= 3 on chunks of >= 2^(k+4) amplitudes: the matrix cores (misc_kernels.h k_dense_mfma2: 16
// blocks per wave and step as the columns of v_mfma_f64_16x16x4_f64; the matrix image in registers for k = 3, 4, in LDS for
// k = 5, 6).  Smaller chunks: one workgroup per block through LDS (k_dense_small).
int qsim_apply_fused_k(qsim_chunk* c, int k, const int32_t* qubits, const double* M) {
  int rc = check_chunk(c, "qsim_apply_fused_k");
  if (rc) return rc;
  if (!qubits || !M) return fail(QSIM_ERR_INVALID, "qsim_apply_fused_k: null argument");
  if (k < 1 || k > 6) return fail(QSIM_ERR_INVALID, "qsim_apply_fused_k: 1 <= k <= 6 qubits expected, got %d", k);
  if (parts_pending(c)) return fail(QSIM_ERR_INVALID, "qsim_apply_fused_k: slab pieces of a split qsim_apply_ops_io call are pending on this chunk");
  for (int i = 0; i < k; ++i) {
    if ((rc = check_local_qubit(c, qubits[i]))) return rc;
    for (int j = 0; j < i; ++j) if (qubits[j] == qubits[i]) return fail(QSIM_ERR_INVALID, "qsim_apply_fused_k: repeated qubit %d", qubits[i]);
  }
  if (k == 1) return qsim_apply_1q(c, qubits[0], M);
  if (k == 2) return qsim_apply_2q(c, qubits[1], qubits[0], M);   // pattern = bit(q0) + 2 bit(q1) = the pair index with qa = q1
  HIP_TRY(hipSetDevice(c->device));
  if ((rc = ensure_scratch(c))) return rc;
  const int N = 1 << k;
  static_assert(kScratchDoubles * sizeof(double) >= 64 * 64 * sizeof(double2), "the chunk's scratch holds a 64 x 64 complex matrix");
  HIP_TRY(hipMemcpyAsync(c->scratch, M, sizeof(double2) * (size_t)N * N, hipMemcpyHostToDevice, c->stream));
  int sorted[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < k; ++i) sorted[i] = qubits[i];
  std::sort(sorted, sorted + k);
  const bool nt = c->span_bytes > tuning().mall_bytes && sorted[0] >= kLaneCut;
#ifdef QSIM_PROBES
  if (k <= 4 && tuning().dense_form == 0) {          // the round-4 kernels, kept in the probe build as A/B partners
    DenseArgs a;
    a.amp = c->amp;
    a.mat = reinterpret_cast<const double2*>(c->scratch);
    a.count = amps(c) >> k;
    for (int i = 0; i < 4; ++i) { a.bit[i] = i < k ? qubits[i] : 0; a.pos[i] = i < k ? sorted[i] : 0; }
    ProfileScope prof(7, 32.0 * (double)amps(c), c->stream, nt);
    if (c->k >= k + 4 && ((tuning().dense_mfma >> (k - 3)) & 1)) {
      DenseMfmaArgs d;
      d.amp = reinterpret_cast<double*>(c->amp);
      d.mat = a.mat;
      d.col_blocks = amps(c) >> (k + 4);
      for (int i = 0; i < 4; ++i) { d.pos[i] = a.pos[i]; d.bit[i] = a.bit[i]; }
      const u64 waves = (d.col_blocks + kDenseMfmaColBlocksPerWave - 1) / kDenseMfmaColBlocksPerWave;
      u64 wgs = (waves + kBlock / 64 - 1) / (kBlock / 64);
      wgs = (wgs + 7) & ~7ull;
      if (k == 3) { if (nt) hipLaunchKernelGGL((k_dense_mfma<3, true>), grid_for(wgs), dim3(kBlock), 0, c->stream, d); else hipLaunchKernelGGL((k_dense_mfma<3, false>), grid_for(wgs), dim3(kBlock), 0, c->stream, d); }
      else { if (nt) hipLaunchKernelGGL((k_dense_mfma<4, true>), grid_for(wgs), dim3(kBlock), 0, c->stream, d); else hipLaunchKernelGGL((k_dense_mfma<4, false>), grid_for(wgs), dim3(kBlock), 0, c->stream, d); }
    } else {
      u64 blocks = (a.count + kBlock - 1) / kBlock;
      blocks = (blocks + 7) & ~7ull;
      if (k == 3) { if (nt) hipLaunchKernelGGL((k_dense<3, true>), grid_for(blocks), dim3(kBlock), 0, c->stream, a); else hipLaunchKernelGGL((k_dense<3, false>), grid_for(blocks), dim3(kBlock), 0, c->stream, a); }
      else { if (nt) hipLaunchKernelGGL((k_dense<4, true>), grid_for(blocks), dim3(kBlock), 0, c->stream, a); else hipLaunchKernelGGL((k_dense<4, false>), grid_for(blocks), dim3(kBlock), 0, c->stream, a); }
    }
    prof.done(c->stream);
    HIP_TRY(hipGetLastError());
    return QSIM_OK;
  }
#endif
  ProfileScope prof(7, 32.0 * (double)amps(c), c->stream, nt);
  if (c->k >= k + 4) {
    DenseMfma2Args d;
    d.amp = c->amp;
    d.mat = reinterpret_cast<const double2*>(c->scratch);
    d.col_blocks = amps(c) >> (k + 4);
    d.consec_log2 = 0;
    d.skew = 0;
    for (int i = 0; i < 6; ++i) {
      d.pos[i] = i < k ? sorted[i] : 0;
      d.to_caller[i] = 0;
      for (int q = 0; q < k; ++q) if (i < k && qubits[q] == sorted[i]) d.to_caller[i] = q;
    }
    // k <= 4: a wave takes four CONSECUTIVE column groups (the matrix image costs a few loads per wave; 64 columns = 1 KiB
    // contiguous per pattern over its four steps: -3 / -5 % against groups strided through the region, the worst
    // placements -8 / -13 %); k = 5, 6: resident workgroups that walk the column groups (the image is built once per
    // workgroup in LDS: 32 / 128 KiB) -- for k = 5 TWO per CU, not the four that fit: fewer bytes in flight suit the
    // memory system better (-9 %; one per CU is as good, three and four are not).  profiles/r05q_dense_knob_scans.txt
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
    int pf = -1, groups = 4, wgs_per_cu = 2, consec_log2 = k <= 4 ? 2 : 0;       // (pf: the product's choice per k, kDensePf)
#ifdef QSIM_PROBES
    if (tuning().dense_pf >= 0) pf = tuning().dense_pf;
    if (tuning().dense_groups > 0) groups = tuning().dense_groups;
    if (tuning().dense_wgs > 0) wgs_per_cu = tuning().dense_wgs;
    if (tuning().dense_consec > 0) { consec_log2 = 0; while ((2 << consec_log2) <= tuning().dense_consec) ++consec_log2; }
    d.skew = (u64)tuning().dense_skew;
#endif
    d.consec_log2 = consec_log2;
    if (k <= 4) {
      const u64 waves = (d.col_blocks + groups - 1) / groups;
      const unsigned grid = (unsigned)std::min<u64>(((std::max<u64>((waves + 3) / 4, 1) + 7) & ~7ull), 1u << 20);   // (whole octets: one region per XCD)
      if (k == 3) { if (nt) launch_dense_mfma2<3, true, 256>(d, grid, c->stream, pf); else launch_dense_mfma2<3, false, 256>(d, grid, c->stream, pf); }
      else        { if (nt) launch_dense_mfma2<4, true, 256>(d, grid, c->stream, pf); else launch_dense_mfma2<4, false, 256>(d, grid, c->stream, pf); }
    } else if (k == 5) {
      const unsigned grid = (unsigned)std::min<u64>((std::max<u64>((d.col_blocks + 3) / 4, 1) + 7) & ~7ull, (u64)cus * wgs_per_cu);
      if (nt) launch_dense_mfma2<5, true, 256>(d, grid, c->stream, pf); else launch_dense_mfma2<5, false, 256>(d, grid, c->stream, pf);
    } else {
      const unsigned grid = (unsigned)std::min<u64>((std::max<u64>((d.col_blocks + 7) / 8, 1) + 7) & ~7ull, (u64)cus);
      if (nt) launch_dense_mfma2<6, true, 512>(d, grid, c->stream, pf); else launch_dense_mfma2<6, false, 512>(d, grid, c->stream, pf);
    }
  } else {
    DenseSmallArgs a;
    a.amp = c->amp;
    a.mat = reinterpret_cast<const double2*>(c->scratch);
    a.k = k;
    for (int i = 0; i < 6; ++i) { a.bit[i] = i < k ? qubits[i] : 0; a.pos[i] = i < k ? sorted[i] : 0; }
    hipLaunchKernelGGL(k_dense_small, dim3((unsigned)(amps(c) >> k)), dim3(64), 0, c->stream, a);
  }
  prof.done(c->stream);
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

// qsim_apply_ops with the high tile bits of the first n_tiles passes named by the caller (bit b of tile_masks[p]: index bit b
// is a tile bit of pass p): the pass builder takes them instead of searching; a mask that holds no op is ignored.
int qsim_apply_ops_tiled(qsim_chunk* c, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                         int n_tiles, const uint64_t* tile_masks) {
  int rc = validate_ops(c, n_ops, nq, qubits, mats);
  if (rc) return rc;
  if (n_tiles < 0 || (n_tiles && !tile_masks)) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_tiled: bad tile list");
  if (parts_pending(c)) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_tiled: slab pieces of a split qsim_apply_ops_io call are pending on this chunk");
  if (n_ops < 2 || c->k < kTileMinChunk || c->k > kTileMaxQubits) {
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
  const TileHint hint = {tile_masks, n_tiles};
  rc = run_fused(c, ops, &passes, nullptr, n_ops, nq, qubits, mats, nullptr, n_tiles ? &hint : nullptr);
  c->last_passes = passes;
  return rc;
}

int qsim_last_pass_count(const qsim_chunk* c) { return c ? c->last_passes : -1; }

// Forget the cached pass images (run_fused keeps those of the last few op lists): the next call plans again.  For callers
// that time a COLD call (bench.py `api_path`) or changed their mind about memory.
int qsim_plan_cache_clear(void) {
  std::lock_guard<std::mutex> lock(g_plan_cache_mu);
  g_plan_cache.clear();
  return QSIM_OK;
}

static_assert(sizeof(TileArgs) == QSIM_PASS_IMAGE_BYTES, "pass image = the kernel-argument block of k_tile");

int qsim_plan_ops_tiled(int n_local_qubits, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                        int n_tiles, const uint64_t* tile_masks, void* out, uint64_t out_capacity_bytes, int32_t* n_passes);
int qsim_plan_ops(int n_local_qubits, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                  void* out, uint64_t out_capacity_bytes, int32_t* n_passes) {
  return qsim_plan_ops_tiled(n_local_qubits, n_ops, nq, qubits, mats, 0, nullptr, out, out_capacity_bytes, n_passes);
}

int qsim_plan_ops_tiled(int n_local_qubits, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                        int n_tiles, const uint64_t* tile_masks, void* out, uint64_t out_capacity_bytes, int32_t* n_passes) {
  if (n_tiles < 0 || (n_tiles && !tile_masks)) return fail(QSIM_ERR_INVALID, "qsim_plan_ops_tiled: bad tile list");
  if (!n_passes) return fail(QSIM_ERR_INVALID, "qsim_plan_ops: n_passes is null");
  if (n_local_qubits < kTileMinChunk || n_local_qubits > kTileMaxQubits)
    return fail(QSIM_ERR_INVALID, "qsim_plan_ops: fused passes need %d..%d local qubits", kTileMinChunk, kTileMaxQubits);
  if (n_ops < 0 || (n_ops && (!nq || !qubits || !mats))) return fail(QSIM_ERR_INVALID, "bad op list");
  std::vector<FusedOp> ops;
  for (int i = 0; i < n_ops; ++i) {
    if (nq[i] != 1 && nq[i] != 2) return fail(QSIM_ERR_INVALID, "op %d: arity %d", i, nq[i]);
    for (int j = 0; j < nq[i]; ++j)
      if (qubits[2 * i + j] < 0 || qubits[2 * i + j] >= n_local_qubits)
        return fail(QSIM_ERR_NONLOCAL, "qubit %d >= log2(chunk_size)=%d: non-local gate requires layout/collect step",
                    qubits[2 * i + j], n_local_qubits);
    if (nq[i] == 2 && qubits[2 * i] == qubits[2 * i + 1]) return fail(QSIM_ERR_INVALID, "op %d: repeated qubit", i);
    FusedOp o;
    if (classify_op(nq[i], qubits + 2 * i, mats + 32 * (size_t)i, &o)) ops.push_back(o);
  }
  int passes = 0;
  char* dst = (char*)out;
  uint64_t used = 0;
  const TileHint hint = {tile_masks, n_tiles};
  int rc = plan_fused(n_local_qubits, ops, &passes, [&](TileArgs& a, int T, double, bool, bool) {
    if (dst) {
      if (used + sizeof(TileArgs) > out_capacity_bytes) return fail(QSIM_ERR_INVALID, "qsim_plan_ops: output buffer too small");
      std::memcpy(dst + used, &a, sizeof a);
    }
    used += sizeof(TileArgs);
    return (int)QSIM_OK;
  }, n_tiles ? &hint : nullptr);
  *n_passes = passes;
  return rc;
}

// The NEXT fused pass of a partly executed op list on a partitioned state (the partition planner's view of the pass builder,
// runner/partition_plan.py): qubits are index bits of the WHOLE state, the bits >= n_local_qubits are rank bits -- an op may
// use them as controls or phase bits (the rank applies or skips it by its own bits) but an op that TARGETS one has to wait
// for a re-layout and blocks what depends on it.  done[i] != 0: op i ran already.  Out: the high tile bits the pass builder
// would choose now (tile_mask, filled to a whole tile; need_mask: the ones its ops need), and the ops it would hold
// (members, ascending; capacity n_ops).  avoid_mask: bits the fill should leave out (slab bits of the re-layout that follows).
// hint_mask != 0: that tile instead of a searched one.  An empty pass (everything waits for a rank bit) is reported as
// *n_members = 0.  Host only, no device.
int qsim_plan_peek_pass(int n_local_qubits, int n_total_qubits, int n_ops, const int32_t* nq, const int32_t* qubits,
                        const double* mats, const uint8_t* done, uint64_t avoid_mask, uint64_t hint_mask, uint64_t* tile_mask,
                        uint64_t* need_mask, int32_t* n_members, int32_t* members) {
  if (!done || !tile_mask || !n_members || !members) return fail(QSIM_ERR_INVALID, "qsim_plan_peek_pass: null argument");
  if (n_local_qubits < kTileMinChunk || n_local_qubits > kTileMaxQubits)
    return fail(QSIM_ERR_INVALID, "qsim_plan_peek_pass: fused passes need %d..%d local qubits", kTileMinChunk, kTileMaxQubits);
  if (n_total_qubits < n_local_qubits || n_total_qubits > 63) return fail(QSIM_ERR_INVALID, "qsim_plan_peek_pass: bad total qubit count %d", n_total_qubits);
  if (n_ops < 0 || (n_ops && (!nq || !qubits || !mats))) return fail(QSIM_ERR_INVALID, "bad op list");
  std::vector<FusedOp> ops;
  std::vector<int32_t> origin;                 // classified op -> index in the caller's list (identities drop out)
  std::vector<uint8_t> done_ops;
  for (int i = 0; i < n_ops; ++i) {
    if (nq[i] != 1 && nq[i] != 2) return fail(QSIM_ERR_INVALID, "op %d: arity %d", i, nq[i]);
    for (int j = 0; j < nq[i]; ++j)
      if (qubits[2 * i + j] < 0 || qubits[2 * i + j] >= n_total_qubits) return fail(QSIM_ERR_INVALID, "op %d: qubit %d out of range", i, qubits[2 * i + j]);
    if (nq[i] == 2 && qubits[2 * i] == qubits[2 * i + 1]) return fail(QSIM_ERR_INVALID, "op %d: repeated qubit", i);
    FusedOp o;
    if (classify_op(nq[i], qubits + 2 * i, mats + 32 * (size_t)i, &o)) { ops.push_back(o); origin.push_back(i); done_ops.push_back(done[i]); }
  }
  std::vector<size_t> held;
  PeekPlan peek;
  peek.n_total = n_total_qubits;
  peek.done = done_ops.data();
  peek.avoid = avoid_mask;
  peek.tile_mask = peek.need_mask = 0;
  peek.members = &held;
  int passes = 0;
  const TileHint hint = {&hint_mask, 1};
  const int rc = plan_fused(n_local_qubits, ops, &passes, [](TileArgs&, int, double, bool, bool) { return (int)QSIM_OK; },
                            hint_mask ? &hint : nullptr, &peek);
  if (rc) return rc;
  *tile_mask = peek.tile_mask;
  if (need_mask) *need_mask = peek.need_mask;
  *n_members = (int32_t)held.size();
  for (size_t j = 0; j < held.size(); ++j) members[j] = origin[held[j]];
  return QSIM_OK;
}

// Pass counts of ONE op list under several qubit layouts (layouts[l * n_local_qubits + q] = the index bit of logical qubit q
// in layout l), planned in parallel on the host: the greedy pass builder's result depends on which three qubits live on the
// line bits (they belong to every tile) -- 17 to 20 passes for the 28-qubit bench circuit -- so an engine that is free to
// choose the layout (runner/engine.py) tries a few dozen and keeps the cheapest.  No device involved.
int qsim_plan_count_layouts(int n_local_qubits, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                            int n_layouts, const int32_t* layouts, int32_t* n_passes, int n_threads) {
  if (!n_passes || n_layouts < 0 || (n_layouts && !layouts)) return fail(QSIM_ERR_INVALID, "qsim_plan_count_layouts: bad arguments");
  if (n_local_qubits < kTileMinChunk || n_local_qubits > kTileMaxQubits)
    return fail(QSIM_ERR_INVALID, "qsim_plan_count_layouts: fused passes need %d..%d local qubits", kTileMinChunk, kTileMaxQubits);
  if (n_ops < 0 || (n_ops && (!nq || !qubits || !mats))) return fail(QSIM_ERR_INVALID, "bad op list");
  for (int i = 0; i < n_ops; ++i) {
    if (nq[i] != 1 && nq[i] != 2) return fail(QSIM_ERR_INVALID, "op %d: arity %d", i, nq[i]);
    for (int j = 0; j < nq[i]; ++j)
      if (qubits[2 * i + j] < 0 || qubits[2 * i + j] >= n_local_qubits) return fail(QSIM_ERR_NONLOCAL, "qubit %d >= log2(chunk_size)=%d: non-local gate requires layout/collect step", qubits[2 * i + j], n_local_qubits);
    if (nq[i] == 2 && qubits[2 * i] == qubits[2 * i + 1]) return fail(QSIM_ERR_INVALID, "op %d: repeated qubit", i);
  }
  for (int l = 0; l < n_layouts; ++l) {
    u64 seen = 0;
    for (int q = 0; q < n_local_qubits; ++q) {
      const int b = layouts[(size_t)l * n_local_qubits + q];
      if (b < 0 || b >= n_local_qubits || ((seen >> b) & 1)) return fail(QSIM_ERR_INVALID, "qsim_plan_count_layouts: layout %d is not a permutation", l);
      seen |= 1ull << b;
    }
  }
  (void)tuning();                                          // (initialised before the threads start)
  std::atomic<int> next{0}, bad{0};
  auto work = [&]() {
    std::vector<FusedOp> ops;
    for (;;) {
      const int l = next.fetch_add(1);
      if (l >= n_layouts) return;
      const int32_t* lay = layouts + (size_t)l * n_local_qubits;
      ops.clear();
      for (int i = 0; i < n_ops; ++i) {
        const int32_t q[2] = {lay[qubits[2 * i]], nq[i] == 2 ? lay[qubits[2 * i + 1]] : -1};
        FusedOp o;
        if (classify_op(nq[i], q, mats + 32 * (size_t)i, &o)) ops.push_back(o);
      }
      int passes = 0;
      const int rc = plan_fused(n_local_qubits, ops, &passes, [](TileArgs&, int, double, bool, bool) { return (int)QSIM_OK; });
      if (rc) bad.store(1);
      n_passes[l] = rc ? -1 : passes;
    }
  };
  const int nt = std::max(1, std::min(n_threads > 0 ? n_threads : 1, std::min(n_layouts, 64)));
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (std::thread& t : pool) t.join();
  if (bad.load()) return fail(QSIM_ERR_INVALID, "qsim_plan_count_layouts: a layout could not be planned");
  return QSIM_OK;
}

// Which index bit should every qubit live on so that the tiles of the given passes fall on index-bit sets with a good DRAM
// pattern?  Simulated annealing over the assignment (bits 0..2, the 128-byte line, stay) under the caller's cost model of
// a tile-bit set: c0 + sum_b bit_cost[b - 3] + sum_{a < b} pair_cost[(a - 3) * nb + (b - 3)], nb = top_bit - 2, bits above
// top_bit priced like top_bit (runner/tile_layout.py holds the coefficients: ridge fits to measured passes).  tile_masks[p] =
// the high tile bits of pass p as LOGICAL qubits; out_l2p[q] = the index bit chosen for qubit q.  Host only.
int qsim_choose_layout(int n_local_qubits, int n_tiles, const uint64_t* tile_masks, int top_bit, const double* bit_cost,
                       const double* pair_cost, const double* triple_cost, uint64_t seed, int sweeps, int32_t* out_l2p,
                       double* cost_identity, double* cost_chosen) {
  const int n = n_local_qubits, low = kTileLow;
  if (n < low + 2 || n > 62 || n_tiles < 0 || (n_tiles && !tile_masks) || !bit_cost || !pair_cost || !out_l2p || top_bit < low || top_bit > 62 || sweeps < 1)
    return fail(QSIM_ERR_INVALID, "qsim_choose_layout: bad arguments");
  const int nb = top_bit - low + 1;
  std::vector<std::vector<int>> tiles((size_t)n_tiles);
  std::vector<std::vector<int>> member((size_t)n);
  for (int t = 0; t < n_tiles; ++t)
    for (int q = low; q < n; ++q)
      if ((tile_masks[t] >> q) & 1) { tiles[(size_t)t].push_back(q); member[(size_t)q].push_back(t); }
  std::vector<double> sym((size_t)nb * nb, 0.0);
  for (int a = 0; a < nb; ++a)
    for (int b = a + 1; b < nb; ++b) sym[(size_t)a * nb + b] = sym[(size_t)b * nb + a] = pair_cost[(size_t)a * nb + b];
  std::vector<int> l2p((size_t)n);
  for (int q = 0; q < n; ++q) l2p[(size_t)q] = q;
  // (optional third-order terms: triple_cost[(a * nb + b) * nb + c] for a < b < c, zero elsewhere)
  auto cost_of = [&](int t) {
    int idx[64], m = 0;
    for (int q : tiles[(size_t)t]) idx[m++] = std::min(l2p[(size_t)q], top_bit) - low;
    double c = 0;
    for (int i = 0; i < m; ++i) {
      c += bit_cost[idx[i]];
      for (int j = i + 1; j < m; ++j) c += sym[(size_t)idx[i] * nb + idx[j]];
    }
    if (triple_cost) {
      std::sort(idx, idx + m);
      for (int i = 0; i < m; ++i)
        for (int j = i + 1; j < m; ++j) {
          if (idx[j] == idx[i]) continue;
          const double* row = triple_cost + ((size_t)idx[i] * nb + idx[j]) * nb;
          for (int l = j + 1; l < m; ++l) if (idx[l] != idx[j]) c += row[idx[l]];
        }
    }
    return c;
  };
  std::vector<double> costs((size_t)n_tiles);
  double cur = 0;
  for (int t = 0; t < n_tiles; ++t) cur += (costs[(size_t)t] = cost_of(t));
  const double identity = cur;
  double best = cur;
  std::vector<int> best_l2p = l2p;
  u64 rs = seed * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull;
  auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
  const int np_ = n - low;
  const long steps = (long)sweeps * np_ * np_ / 2;
  const double T0 = std::max(1e-3, 0.03 * identity / std::max(1, n_tiles));
  std::vector<int> touched;
  std::vector<double> fresh;
  for (long it = 0; it < steps && n_tiles > 0; ++it) {
    const int a = low + (int)(rnd() % (u64)np_);
    int b = low + (int)(rnd() % (u64)(np_ - 1));
    if (b >= a) ++b;
    touched.clear();
    for (int t : member[(size_t)a]) touched.push_back(t);
    for (int t : member[(size_t)b]) if (std::find(touched.begin(), touched.end(), t) == touched.end()) touched.push_back(t);
    if (touched.empty()) continue;
    std::swap(l2p[(size_t)a], l2p[(size_t)b]);
    fresh.clear();
    double delta = 0;
    for (int t : touched) { fresh.push_back(cost_of(t)); delta += fresh.back() - costs[(size_t)t]; }
    const double T = T0 * (1.0 - (double)it / (double)steps) + 1e-4;
    const double u = (double)(rnd() >> 11) * 0x1p-53;
    if (delta < 0 || u < std::exp(-delta / T)) {
      for (size_t i = 0; i < touched.size(); ++i) costs[(size_t)touched[i]] = fresh[i];
      cur += delta;
      if (cur < best - 1e-12) { best = cur; best_l2p = l2p; }
    } else {
      std::swap(l2p[(size_t)a], l2p[(size_t)b]);
    }
  }
  for (int q = 0; q < n; ++q) out_l2p[q] = best_l2p[(size_t)q];
  if (cost_identity) *cost_identity = identity;
  if (cost_chosen) *cost_chosen = best;
  return QSIM_OK;
}

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

static int slabs_all(qsim_chunk* state, int m, const int32_t* bits, qsim_chunk* buf, int skip_pattern, int piece,
                     int n_pieces, bool pack, const char* what) {
  int rc = check_chunk(state, what);
  if (rc || (rc = check_chunk(buf, what))) return rc;
  int pos[3];
  u64 voff, n_slab;
  if ((rc = slab_args(state, m, bits, 0, buf, 0, pos, &voff, &n_slab))) return rc;
  if (amps(buf) < amps(state)) return fail(QSIM_ERR_INVALID, "%s: the buffer must hold all 2^%d slabs", what, m);
  if (skip_pattern < -1 || skip_pattern >= (1 << m)) return fail(QSIM_ERR_INVALID, "%s: skip pattern out of range", what);
  int piece_bits = 0;
  while ((1 << piece_bits) < n_pieces) ++piece_bits;
  if (n_pieces < 1 || (1 << piece_bits) != n_pieces || piece_bits > 3 || piece_bits > state->k - m)
    return fail(QSIM_ERR_INVALID, "%s: n_pieces must be 1, 2, 4 or 8 and at most the slab length", what);
  if (piece < 0 || piece >= n_pieces) return fail(QSIM_ERR_INVALID, "%s: piece %d out of range", what, piece);
  int pb[3] = {0, 0, 0};      // the top piece_bits index bits that are NOT selected, ascending
  for (int b = state->k - 1, found = 0; b >= 0 && found < piece_bits; --b) {
    bool selected = false;
    for (int i = 0; i < m; ++i) selected = selected || bits[i] == b;
    if (!selected) pb[piece_bits - 1 - found++] = b;
  }
  HIP_TRY(hipSetDevice(state->device));
  const int b0 = bits[0], b1 = m > 1 ? bits[1] : 0, b2 = m > 2 ? bits[2] : 0;
  const int s_lo = pos[0], s_mid = m > 1 ? pos[1] : 0, s_hi = m > 2 ? pos[2] : 0;   // pos is sorted ascending
  const u64 n = amps(state) >> piece_bits;
  if (pack)
    hipLaunchKernelGGL((k_slabs_all<true>), dim3(stream_grid(n)), dim3(kBlock), 0, state->stream, state->amp, buf->amp,
                       n, m, b0, b1, b2, s_hi, s_mid, s_lo, state->k - m, skip_pattern, piece_bits, pb[0], pb[1], pb[2], piece);
  else
    hipLaunchKernelGGL((k_slabs_all<false>), dim3(stream_grid(n)), dim3(kBlock), 0, state->stream, state->amp, buf->amp,
                       n, m, b0, b1, b2, s_hi, s_mid, s_lo, state->k - m, skip_pattern, piece_bits, pb[0], pb[1], pb[2], piece);
  HIP_TRY(hipGetLastError());
  return QSIM_OK;
}

int qsim_pack_all(const qsim_chunk* src, int m, const int32_t* bits, qsim_chunk* buf, int skip_pattern, int piece,
                  int n_pieces) {
  return slabs_all(const_cast<qsim_chunk*>(src), m, bits, buf, skip_pattern, piece, n_pieces, true, "qsim_pack_all");
}

int qsim_unpack_all(qsim_chunk* dst, int m, const int32_t* bits, const qsim_chunk* buf, int skip_pattern, int piece,
                    int n_pieces) {
  return slabs_all(dst, m, bits, const_cast<qsim_chunk*>(buf), skip_pattern, piece, n_pieces, false, "qsim_unpack_all");
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

// The slab-storing end of an op list whose tile passes (all but a stashed last one) have been queued: the split form's
// bookkeeping (the pieces are stored by qsim_apply_ops_io_part), or the pack passes of a list that could not fuse them.
static int finish_out_side(qsim_chunk* c, const qsim_ops_io* io, FusedIo& fio) {
  int rc = QSIM_OK;
  if (io->dst && io->dst_parts != 0) {
    // split form: the slabs are stored piece by piece by qsim_apply_ops_io_part -- partial launches of the planned last
    // pass, or (nothing fusable) qsim_pack_all pieces of the final state, or nothing (stored already)
    PendingLast* p = c->pending ? c->pending : (c->pending = new PendingLast());
    const bool stashed = p->mode == PendingLast::kStashed;
    p->mode = stashed ? PendingLast::kTile : (fio.fused_out ? PendingLast::kDone : PendingLast::kPack);
    p->m = io->dst_m;
    for (int i = 0; i < io->dst_m; ++i) p->bits[i] = io->dst_bits[i];
    p->dst = io->dst; p->dst_own = io->dst_own; p->own_pattern = io->own_pattern;
    const int min_piece_bits = io->dst_parts < 0 ? kTileLow : 20;        // (negative: tests cut small shards too)
    const int want = io->dst_parts < 0 ? -io->dst_parts : io->dst_parts;
    plan_parts(p, c->k, want, p->mode == PendingLast::kTile ? p->a.h : nullptr, p->mode == PendingLast::kTile ? p->T - kTileLow : 0, min_piece_bits);
  } else if (io->dst && !fio.fused_out) {     // not fused: pack passes
    if ((rc = slabs_all(c, io->dst_m, io->dst_bits, io->dst, io->own_pattern, 0, 1, true, "qsim_apply_ops_io"))) return rc;
    if (io->own_pattern >= 0) {
      const uint64_t slab = 1ull << (c->k - io->dst_m);
      if ((rc = qsim_pack_bits(c, io->dst_m, io->dst_bits, io->own_pattern, io->dst_own, (uint64_t)io->own_pattern * slab))) return rc;
    }
  }
  return QSIM_OK;
}

// qsim_ops_io::src_parts: plan now, launch as the source pieces are announced (qsim_apply_ops_io_load)
static int apply_ops_io_deferred(qsim_chunk* c, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                                 const qsim_ops_io* io, const FusedIo& fio_in, const std::vector<FusedOp>& ops, bool tiles, int* n_passes) {
  DeferredIo* d = c->deferred ? c->deferred : (c->deferred = new DeferredIo());
  *d = DeferredIo();
  d->tiles = tiles;
  d->n_ops = n_ops;
  d->io = *io;
  d->fio = fio_in;
  const int want = io->src_parts < 0 ? -io->src_parts : io->src_parts;
  d->nb = piece_bits_for(c->k, io->src_m, want, io->src_parts < 0 ? kTileLow : 20);
  auto is_slab = [&](int b) { for (int i = 0; i < io->src_m; ++i) if (io->src_bits[i] == b) return true; return false; };
  int top[3] = {0, 0, 0}, found = 0;
  for (int b = c->k - 1; b >= 0 && found < d->nb; --b) if (!is_slab(b)) top[found++] = b;
  for (int i = 0; i < d->nb; ++i) d->piece_bit[i] = top[d->nb - 1 - i];
  int passes = 0, rc = QSIM_OK;
  if (io->src && !d->fio.src) ++passes;                     // the source cannot be read by a tile pass: unpack pieces
  if (tiles) {
    int p = 0;
    const TileHint hint = {io->tile_masks, io->n_tiles};
    if ((rc = run_fused(c, ops, &p, &d->fio, n_ops, nq, qubits, mats, &d->passes, io->n_tiles ? &hint : nullptr))) return rc;
    d->io.tile_masks = nullptr;                               // (the caller's array: used by the plan above only)
    d->io.n_tiles = 0;
    passes += p;
    c->own_in_chunk = d->fio.own_in_chunk;
    if (d->fio.src && !d->fio.fused_in) return fail(QSIM_ERR_INVALID, "internal: the first pass did not take the source buffer");
    // the first pass in partial launches: when it reads the source itself, is not also the slab-storing pass of a split
    // / fused destination, and the top piece bits are no tile bits of it
    const bool first_stores = d->passes.size() == 1 && io->dst != nullptr;
    if (d->fio.src && !first_stores && !d->passes.empty() && d->passes[0].T == kTileBitsMax) {
      auto is_tile = [&](int b) { for (int j = 0; j < kTileBitsMax - kTileLow; ++j) if (d->passes[0].a.h[j] == b) return true; return false; };
      while (d->nb_free < d->nb && !is_tile(top[d->nb_free])) ++d->nb_free;
    }
  } else {
    d->nq.assign(nq, nq + n_ops);
    d->qubits.assign(qubits, qubits + 2 * (size_t)n_ops);
    d->mats.assign(mats, mats + 32 * (size_t)n_ops);
    passes += n_ops;
  }
  if (io->dst && !d->fio.fused_out) ++passes;
  d->active = true;
  c->last_passes = passes;
  if (n_passes) *n_passes = passes;
  return QSIM_OK;
}

// Op list with a re-layout fused into its ends (staging.py:136-152 SWAP lists): the FIRST fused pass reads
// the state from io->src in the slab layout of qsim_pack_all over io->src_bits (what an all-to-all left in the receive
// buffer) instead of a separate unpack pass, the LAST one stores it into io->dst in the slab layout over io->dst_bits
// (slab io->own_pattern, which stays on this rank, into io->dst_own) instead of a separate pack pass.  Whatever cannot
// be fused (bits inside a 128-B line, chunks too small for tile passes, an empty op list, a slab bit that is a tile
// bit of the last pass) is done with the slab kernels, so the result is the same in every case; *n_passes counts the
// HBM passes really made.
int qsim_apply_ops_io(qsim_chunk* c, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                      const qsim_ops_io* io, int* n_passes) {
  int rc = validate_ops(c, n_ops, nq, qubits, mats);
  if (rc) return rc;
  if (!io) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io: io is null");
  if (io->struct_size != sizeof(qsim_ops_io))
    return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io: io->struct_size is %u, this library's qsim_ops_io has %zu bytes (zero the struct, "
                "set struct_size = sizeof(qsim_ops_io) and rebuild against this library's include/qsim_hip.h)", io->struct_size, sizeof(qsim_ops_io));
  if (io->n_tiles < 0 || (io->n_tiles && !io->tile_masks)) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io: bad tile list");
  auto check_side = [&](const qsim_chunk* b, int m, const int32_t* bits, const char* side) -> int {
    int r = check_chunk(b, "qsim_apply_ops_io");
    if (r) return r;
    if (b->k != c->k || b->device != c->device || b->amp == c->amp)
      return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io: the %s buffer must be a distinct chunk of the state's size and device", side);
    if (m < 1 || m > 3 || m > c->k) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io: %s: 1..3 slab bits expected, got %d", side, m);
    for (int i = 0; i < m; ++i) {
      if (bits[i] < 0 || bits[i] >= c->k) return fail(QSIM_ERR_NONLOCAL, "qsim_apply_ops_io: %s slab bit %d is non-local for 2^%d amplitudes", side, bits[i], c->k);
      for (int j = 0; j < i; ++j) if (bits[j] == bits[i]) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io: %s: repeated slab bit %d", side, bits[i]);
    }
    return QSIM_OK;
  };
  if (io->src && (rc = check_side(io->src, io->src_m, io->src_bits, "source"))) return rc;
  if (io->dst) {
    if ((rc = check_side(io->dst, io->dst_m, io->dst_bits, "destination"))) return rc;
    if (io->own_pattern < -1 || io->own_pattern >= (1 << io->dst_m)) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io: own_pattern out of range");
    if (io->own_pattern >= 0 && (rc = check_side(io->dst_own, io->dst_m, io->dst_bits, "own-slab"))) return rc;
    if (io->own_pattern >= 0 && io->dst_own->amp == io->dst->amp) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io: the own-slab buffer must differ from the destination");
    if (io->src && io->src->amp == io->dst->amp)
      return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io: the source buffer must differ from the destination buffer (one pass may read and write them at once)");
    // (dst_own == src is allowed: with two or more kernels the source has been consumed before the own slab is stored;
    // when ONE pass does everything the own slab goes into the chunk instead -- qsim_apply_ops_io_own_slab tells)
  }
  HIP_TRY(hipSetDevice(c->device));
  std::vector<FusedOp> ops;
  ops.reserve((size_t)n_ops);
  for (int i = 0; i < n_ops; ++i) {
    FusedOp o;
    if (classify_op(nq[i], qubits + 2 * i, mats + 32 * (size_t)i, &o)) ops.push_back(o);
  }
  const bool tiles = c->k >= kTileMinChunk && c->k <= kTileMaxQubits && !ops.empty();
  auto whole_lines = [](int m, const int32_t* bits) { for (int i = 0; i < m; ++i) if (bits[i] < kTileLow) return false; return true; };
  FusedIo fio;
  if (io->src && tiles && whole_lines(io->src_m, io->src_bits)) {
    fio.src = io->src;
    fio.in.m = io->src_m;
    for (int i = 0; i < io->src_m; ++i) fio.in.bits[i] = io->src_bits[i];
  }
  if (io->dst && tiles && whole_lines(io->dst_m, io->dst_bits)) {
    fio.dst = io->dst;
    fio.dst_own = io->dst_own;
    fio.own_pattern = io->own_pattern;
    fio.out.m = io->dst_m;
    for (int i = 0; i < io->dst_m; ++i) fio.out.bits[i] = io->dst_bits[i];
  }
  const bool parts = io->dst && io->dst_parts != 0;
  c->own_in_chunk = false;
  if (parts_pending(c))
    return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io: pieces of an earlier split call are pending on this chunk (qsim_apply_ops_io_part / _load)");
  fio.parts = parts;
  if (io->src && io->src_parts != 0) return apply_ops_io_deferred(c, n_ops, nq, qubits, mats, io, fio, ops, tiles, n_passes);
  int passes = 0;
  if (io->src && !fio.src) {           // not fusable: one unpack pass brings the state into the chunk
    if ((rc = slabs_all(c, io->src_m, io->src_bits, const_cast<qsim_chunk*>(io->src), -1, 0, 1, false, "qsim_apply_ops_io"))) return rc;
    ++passes;
  }
  if (tiles) {
    int p = 0;
    const TileHint hint = {io->tile_masks, io->n_tiles};
    if ((rc = run_fused(c, ops, &p, &fio, n_ops, nq, qubits, mats, nullptr, io->n_tiles ? &hint : nullptr))) return rc;
    passes += p;
    c->own_in_chunk = fio.own_in_chunk;
    if (fio.src && !fio.fused_in) return fail(QSIM_ERR_INVALID, "internal: the first pass did not take the source buffer");
  } else {
    if ((rc = qsim_apply_ops_unfused(c, n_ops, nq, qubits, mats))) return rc;
    passes += n_ops;
  }
  if (io->dst && !fio.fused_out) ++passes;                 // a pack pass (whole, or piece by piece)
  if ((rc = finish_out_side(c, io, fio))) return rc;
  c->last_passes = passes;
  if (n_passes) *n_passes = passes;
  return QSIM_OK;
}

// Split form of the slab-storing end of qsim_apply_ops_io (qsim_ops_io::dst_parts): store piece `part` of every slab.
int qsim_apply_ops_io_part(qsim_chunk* c, int part) {
  int rc = check_chunk(c, "qsim_apply_ops_io_part");
  if (rc) return rc;
  PendingLast* p = c->pending;
  if (!p || p->mode == PendingLast::kNone) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io_part: no split op list is pending on this chunk");
  const int n_parts = 1 << p->nb;
  if (part < 0 || part >= n_parts) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io_part: piece %d out of range", part);
  if ((p->stored >> part) & 1) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io_part: piece %d has been stored already", part);
  HIP_TRY(hipSetDevice(c->device));
  if (p->mode == PendingLast::kTile) {
    // the partial launch that holds this piece: the top nb_free piece bits select it, it stores 2^(nb - nb_free) pieces
    const int g = part >> (p->nb - p->nb_free);
    if (!((p->launched >> g) & 1)) {
      TileArgs a = p->a;
      a.nfix = (uint8_t)p->nb_free;
      a.fix_or = 0;
      for (int i = 0; i < p->nb_free; ++i) {
        const int bit = p->piece_bit[p->nb - p->nb_free + i];            // ascending: the top nb_free piece bits
        int below = 0;
        for (int j = 0; j < p->T - kTileLow; ++j) below += a.h[j] < bit;
        a.fix_pos[i] = (uint8_t)(bit - below);
        if ((g >> i) & 1) a.fix_or |= 1ull << bit;
        if (bit >= c->k || bit < kTileLow) return fail(QSIM_ERR_INVALID, "internal: piece bit %d", bit);
      }
      if ((rc = launch_tile_any(a, p->T, c, c->stream, p->alg_bytes / (double)(1 << p->nb_free)))) return rc;
      p->launched |= 1u << g;
    }
  } else if (p->mode == PendingLast::kPack) {
    // (the pieces of qsim_pack_all are the values of the top non-slab index bits: the same cut)
    if ((rc = slabs_all(c, p->m, p->bits, p->dst, p->own_pattern, part, n_parts, true, "qsim_apply_ops_io_part"))) return rc;
    if (p->own_pattern >= 0 && p->stored == 0) {            // the own slab goes to the receive buffer whole, with the first piece
      const uint64_t slab = 1ull << (c->k - p->m);
      if ((rc = qsim_pack_bits(c, p->m, p->bits, p->own_pattern, p->dst_own, (uint64_t)p->own_pattern * slab))) return rc;
    }
  }
  p->stored |= 1u << part;
  if (p->stored == (1u << n_parts) - 1u) p->mode = PendingLast::kNone;
  return QSIM_OK;
}

// Receive side of the split form (qsim_ops_io::src_parts): piece `part` of every slab of the source has arrived (its
// transfer is ordered before this call on the chunk's stream).  Launches what can run: an unpack piece, a partial launch of
// the first pass whose source pieces are all there, and -- with the last piece -- everything else of the op list.
int qsim_apply_ops_io_load(qsim_chunk* c, int part) {
  int rc = check_chunk(c, "qsim_apply_ops_io_load");
  if (rc) return rc;
  DeferredIo* d = c->deferred;
  if (!d || !d->active) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io_load: no op list with a split source is pending on this chunk");
  const int n_parts = 1 << d->nb;
  if (part < 0 || part >= n_parts) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io_load: piece %d out of range", part);
  if ((d->announced >> part) & 1) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io_load: piece %d has been announced already", part);
  HIP_TRY(hipSetDevice(c->device));
  struct Guard { qsim_chunk* c; bool armed; ~Guard() { if (armed) drop_pending(c); } } guard{c, true};   // an error leaves nothing pending
  d->announced |= 1u << part;
  const qsim_ops_io* io = &d->io;
  if (!d->fio.src) {                  // unpack mode: this piece goes home now
    if ((rc = slabs_all(c, io->src_m, io->src_bits, const_cast<qsim_chunk*>(io->src), -1, part, n_parts, false, "qsim_apply_ops_io_load"))) return rc;
  } else if (d->nb_free > 0) {        // partial launches of the first pass: group g = the top nb_free bits of the piece number
    const int shift = d->nb - d->nb_free;
    const int g = part >> shift;
    const unsigned group = ((1u << (1 << shift)) - 1u) << (g << shift);
    if ((d->announced & group) == group && !((d->launched >> g) & 1)) {
      CachedPass& p0 = d->passes[0];
      TileArgs a = p0.a;
      a.nfix = (uint8_t)d->nb_free;
      a.fix_or = 0;
      for (int i = 0; i < d->nb_free; ++i) {
        const int bit = d->piece_bit[shift + i];
        int below = 0;
        for (int j = 0; j < p0.T - kTileLow; ++j) below += a.h[j] < bit;
        a.fix_pos[i] = (uint8_t)(bit - below);
        if ((g >> i) & 1) a.fix_or |= 1ull << bit;
        if (bit >= c->k || bit < kTileLow) return fail(QSIM_ERR_INVALID, "internal: piece bit %d", bit);
      }
      if ((rc = launch_tile_any(a, p0.T, c, c->stream, p0.alg_bytes / (double)(1 << d->nb_free)))) return rc;
      d->launched |= 1u << g;
    }
  }
  if (d->announced != (n_parts >= 32 ? ~0u : (1u << n_parts) - 1u)) { guard.armed = false; return QSIM_OK; }
  // the source is complete: the rest of the op list
  d->active = false;
  if (d->tiles) {
    for (size_t i = 0; i < d->passes.size(); ++i) {
      if (i == 0 && d->nb_free > 0) continue;               // ran in partial launches
      CachedPass& p = d->passes[i];
      if ((rc = dispatch_planned(c, p.a, p.T, p.alg_bytes, i + 1 == d->passes.size(), &d->fio))) return rc;
    }
  } else {
    if ((rc = qsim_apply_ops_unfused(c, d->n_ops, d->nq.data(), d->qubits.data(), d->mats.data()))) return rc;
  }
  if ((rc = finish_out_side(c, io, d->fio))) return rc;
  guard.armed = false;
  return QSIM_OK;
}

// Where the last qsim_apply_ops_io on this chunk left the slab that stays on the rank: 0 = in io->dst_own, 1 = in the chunk
// itself (dst_own was the source buffer and one pass did everything; the exchange then has to deliver into the chunk).
int qsim_apply_ops_io_own_slab(const qsim_chunk* c, int32_t* in_chunk) {
  if (!c || !in_chunk) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io_own_slab: null argument");
  *in_chunk = c->own_in_chunk ? 1 : 0;
  return QSIM_OK;
}

// Source pieces of the pending op list (qsim_ops_io::src_parts): how many, their size, and how many partial launches the
// first pass takes (0: it runs whole after the last piece).
int qsim_apply_ops_io_source_parts(const qsim_chunk* c, int32_t* n_parts, uint64_t* piece_amps, int32_t* n_launches) {
  if (!c || !c->deferred || !c->deferred->active) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io_source_parts: nothing pending on this chunk");
  const DeferredIo* d = c->deferred;
  if (n_parts) *n_parts = 1 << d->nb;
  if (piece_amps) *piece_amps = (1ull << (c->k - d->io.src_m)) >> d->nb;
  if (n_launches) *n_launches = d->nb_free > 0 ? (1 << d->nb_free) : 0;
  return QSIM_OK;
}

// The pieces of the pending split op list: piece j of EVERY slab d is [d * 2^(k - m) + j * piece_amps, + piece_amps) of the
// send / receive buffers.  n_parts depends only on (k, m, dst_parts): the same on every rank.
int qsim_apply_ops_io_parts(const qsim_chunk* c, int32_t* n_parts, uint64_t* piece_amps, int32_t* n_launches) {
  if (!c || !c->pending || c->pending->mode == PendingLast::kNone) return fail(QSIM_ERR_INVALID, "qsim_apply_ops_io_parts: no split op list is pending on this chunk");
  const PendingLast* p = c->pending;
  if (n_parts) *n_parts = 1 << p->nb;
  if (piece_amps) *piece_amps = (1ull << (c->k - p->m)) >> p->nb;
  if (n_launches) *n_launches = p->mode == PendingLast::kTile ? (1 << p->nb_free) : (p->mode == PendingLast::kPack ? (1 << p->nb) : 0);
  return QSIM_OK;
}

// (k, m, pieces asked for) -> pieces made: the rule of the split form as a pure function (schedulers, dry runs, tests)
int qsim_split_piece_count(int n_local_qubits, int m, int dst_parts) {
  if (m < 1 || m > 3 || m > n_local_qubits || dst_parts == 0) return 1;
  return 1 << piece_bits_for(n_local_qubits, m, dst_parts < 0 ? -dst_parts : dst_parts, dst_parts < 0 ? kTileLow : 20);
}

// ---- multi-GPU reach of the C ABI (comm_rccl.h) ---------------------------------------------------
int qsim_comm_get_unique_id(uint8_t id[QSIM_COMM_ID_BYTES]) {
  static_assert(QSIM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  if (!id) return fail(QSIM_ERR_INVALID, "id is null");
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId uid;
  RCCL_TRY(g_rccl.GetUniqueId(&uid));
  std::memcpy(id, uid.internal, NCCL_UNIQUE_ID_BYTES);
  return QSIM_OK;
}

int qsim_comm_init(int device, int rank, int world, const uint8_t id[QSIM_COMM_ID_BYTES], qsim_comm** out) {
  if (!out || !id) return fail(QSIM_ERR_INVALID, "null argument");
  if (world < 1 || (world & (world - 1)) || rank < 0 || rank >= world)
    return fail(QSIM_ERR_INVALID, "qsim_comm_init: world %d must be a power of two and 0 <= rank %d < world", world, rank);
  int rc = rccl_load();
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  RCCL_TRY(g_rccl.CommInitRank(&comm, world, uid, rank));
  qsim_comm* c = new qsim_comm();
  c->comm = comm; c->rank = rank; c->world = world; c->device = device;
  c->xfer_stream = nullptr;
  for (auto& e : c->ev) e = nullptr;
  for (auto& e : c->ev_bg) e = nullptr;
  c->bg_posted = 0;
  *out = c;
  return QSIM_OK;
}

int qsim_comm_destroy(qsim_comm* c) {
  if (!c) return QSIM_OK;
  (void)hipSetDevice(c->device);
  if (c->xfer_stream) { (void)hipStreamSynchronize(c->xfer_stream); (void)hipStreamDestroy(c->xfer_stream); }
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : c->ev_bg) if (e) (void)hipEventDestroy(e);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
  return QSIM_OK;
}

int qsim_comm_rank(const qsim_comm* c) { return c ? c->rank : -1; }
int qsim_comm_world(const qsim_comm* c) { return c ? c->world : -1; }

int qsim_comm_exchange(qsim_comm* cm, int n_peers, const int32_t* peers, const qsim_chunk* send, const uint64_t* send_off,
                       qsim_chunk* recv, const uint64_t* recv_off, uint64_t count_amps) {
  int rc = check_comm(cm, "qsim_comm_exchange");
  if (rc || (rc = check_chunk(send, "qsim_comm_exchange")) || (rc = check_chunk(recv, "qsim_comm_exchange"))) return rc;
  if (n_peers < 0 || (n_peers && (!peers || !send_off || !recv_off))) return fail(QSIM_ERR_INVALID, "qsim_comm_exchange: bad peer list");
  if (send->amp == recv->amp) return fail(QSIM_ERR_INVALID, "qsim_comm_exchange: send and receive chunks must differ");
  if (send->stream != recv->stream) return fail(QSIM_ERR_INVALID, "qsim_comm_exchange: the send and the receive chunk must share a stream (the transfer is ordered on it)");
  for (int i = 0; i < n_peers; ++i) {
    if (peers[i] < 0 || peers[i] >= cm->world) return fail(QSIM_ERR_INVALID, "qsim_comm_exchange: peer %d out of range", peers[i]);
    if (send_off[i] > amps(send) || count_amps > amps(send) - send_off[i] || recv_off[i] > amps(recv) || count_amps > amps(recv) - recv_off[i])
      return fail(QSIM_ERR_INVALID, "qsim_comm_exchange: slice %d outside its chunk", i);
  }
  HIP_TRY(hipSetDevice(cm->device));
  return comm_exchange(cm, n_peers, peers, send->amp, send_off, recv->amp, recv_off, count_amps, send->stream);
}

// Background form: the group is queued on the communicator's transfer stream behind everything queued on the chunks'
// stream SO FAR; what is queued on the chunks' stream later runs beside it.  qsim_comm_join makes a chunk's stream wait
// for every background transfer posted so far (the piece pipeline of a fused re-layout: runner/distributed.py).
int qsim_comm_exchange_bg(qsim_comm* cm, int n_peers, const int32_t* peers, const qsim_chunk* send, const uint64_t* send_off,
                          qsim_chunk* recv, const uint64_t* recv_off, uint64_t count_amps, uint32_t* ticket) {
  int rc = check_comm(cm, "qsim_comm_exchange_bg");
  if (rc || (rc = check_chunk(send, "qsim_comm_exchange_bg")) || (rc = check_chunk(recv, "qsim_comm_exchange_bg"))) return rc;
  if (n_peers < 0 || (n_peers && (!peers || !send_off || !recv_off))) return fail(QSIM_ERR_INVALID, "qsim_comm_exchange_bg: bad peer list");
  if (send->amp == recv->amp) return fail(QSIM_ERR_INVALID, "qsim_comm_exchange_bg: send and receive chunks must differ");
  if (send->stream != recv->stream) return fail(QSIM_ERR_INVALID, "qsim_comm_exchange_bg: the send and the receive chunk must share a stream");
  for (int i = 0; i < n_peers; ++i) {
    if (peers[i] < 0 || peers[i] >= cm->world) return fail(QSIM_ERR_INVALID, "qsim_comm_exchange_bg: peer %d out of range", peers[i]);
    if (send_off[i] > amps(send) || count_amps > amps(send) - send_off[i] || recv_off[i] > amps(recv) || count_amps > amps(recv) - recv_off[i])
      return fail(QSIM_ERR_INVALID, "qsim_comm_exchange_bg: slice %d outside its chunk", i);
  }
  HIP_TRY(hipSetDevice(cm->device));
  if (!cm->xfer_stream) HIP_TRY(hipStreamCreateWithFlags(&cm->xfer_stream, hipStreamNonBlocking));
  for (auto& e : cm->ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : cm->ev_bg) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(cm->ev[14], send->stream));
  HIP_TRY(hipStreamWaitEvent(cm->xfer_stream, cm->ev[14], 0));
  if ((rc = comm_exchange(cm, n_peers, peers, send->amp, send_off, recv->amp, recv_off, count_amps, cm->xfer_stream))) return rc;
  const uint32_t t = cm->bg_posted++;
  HIP_TRY(hipEventRecord(cm->ev_bg[t % 16], cm->xfer_stream));
  if (ticket) *ticket = t;
  return QSIM_OK;
}

// The chunk's stream waits for background exchange `ticket` (and, transfers of one communicator running in order, for
// every one posted before it) -- not for later ones: the pieces of a re-layout are consumed as they arrive.
int qsim_comm_wait(qsim_comm* cm, qsim_chunk* c, uint32_t ticket) {
  int rc = check_comm(cm, "qsim_comm_wait");
  if (rc || (rc = check_chunk(c, "qsim_comm_wait"))) return rc;
  if (ticket >= cm->bg_posted) return fail(QSIM_ERR_INVALID, "qsim_comm_wait: ticket %u has not been handed out", ticket);
  if (cm->bg_posted - ticket > 16) return qsim_comm_join(cm, c);   // its event has been reused: wait for everything posted
  HIP_TRY(hipSetDevice(cm->device));
  HIP_TRY(hipStreamWaitEvent(c->stream, cm->ev_bg[ticket % 16], 0));
  return QSIM_OK;
}

int qsim_comm_join(qsim_comm* cm, qsim_chunk* c) {
  int rc = check_comm(cm, "qsim_comm_join");
  if (rc || (rc = check_chunk(c, "qsim_comm_join"))) return rc;
  if (!cm->xfer_stream) return QSIM_OK;                     // nothing was ever posted in the background
  HIP_TRY(hipSetDevice(cm->device));
  HIP_TRY(hipEventRecord(cm->ev[15], cm->xfer_stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, cm->ev[15], 0));
  return QSIM_OK;
}

// ---- all-to-all re-layout: ONE schedule, computed by a pure function --------------------------------------------
// Who sends what to whom when rank `rank` of `world` swaps its local bits local_bits[i] with the rank bits
// global_bits[i] (bit g of the rank = qubit k + g): the partner semantics of the reference's chunk groups
// (wenbo_engine/runner/single_node.py:222-245) with one chunk per rank.  Slab d of the send buffer (offset d * 2^(k-m):
// this rank's amplitudes whose local bits have the pattern d) goes to the rank whose global-bit pattern is d, and that
// rank's slab `own` (own = this rank's pattern) arrives at the same offset d of the receive buffer; the own slab stays.
// Pieces: every slab is cut into n_pieces equal parts that travel one after the other (pack / transfer / unpack overlap).
struct RelayoutPlan {
  int n_pieces, n_peers, own;
  int32_t peers[7];
  uint64_t offs[7];          // amplitude offset of the peer's slab in the send AND the receive buffer
  uint64_t slab, part;       // amplitudes per slab / per piece
};
static int relayout_plan(int rank, int world, int k, int m, const int32_t* local_bits, const int32_t* global_bits,
                         int n_pieces, RelayoutPlan* p) {
  if (world < 1 || (world & (world - 1)) || rank < 0 || rank >= world) return fail(QSIM_ERR_INVALID, "re-layout: bad rank %d / world %d", rank, world);
  if (m < 1 || m > 3 || !local_bits || !global_bits) return fail(QSIM_ERR_INVALID, "re-layout: 1..3 qubit pairs expected, got %d", m);
  int g_bits = 0;
  while ((1 << g_bits) < world) ++g_bits;
  for (int i = 0; i < m; ++i) {
    if (local_bits[i] < 0 || local_bits[i] >= k) return fail(QSIM_ERR_NONLOCAL, "re-layout: local bit %d is non-local for 2^%d shards", local_bits[i], k);
    if (global_bits[i] < 0 || global_bits[i] >= g_bits) return fail(QSIM_ERR_INVALID, "re-layout: rank bit %d out of range", global_bits[i]);
    for (int j = 0; j < i; ++j)
      if (local_bits[j] == local_bits[i] || global_bits[j] == global_bits[i]) return fail(QSIM_ERR_INVALID, "re-layout: repeated bit");
  }
  if (n_pieces != 1 && n_pieces != 2 && n_pieces != 4 && n_pieces != 8) return fail(QSIM_ERR_INVALID, "re-layout: n_pieces must be 1, 2, 4 or 8");
  while (n_pieces > 1 && (k - m) - (31 - __builtin_clz((unsigned)n_pieces)) < 20) n_pieces >>= 1;   // pieces stay >= 2^20 amplitudes
  if (k - m < 3) n_pieces = 1;
  p->n_pieces = n_pieces;
  p->own = 0;
  for (int i = 0; i < m; ++i) p->own |= ((rank >> global_bits[i]) & 1) << i;
  p->slab = 1ull << (k - m);
  p->part = p->slab / (u64)n_pieces;
  p->n_peers = 0;
  for (int d = 0; d < (1 << m); ++d) {
    if (d == p->own) continue;
    int peer = rank;
    for (int i = 0; i < m; ++i) peer = (peer & ~(1 << global_bits[i])) | (((d >> i) & 1) << global_bits[i]);
    p->peers[p->n_peers] = peer;
    p->offs[p->n_peers] = (u64)d * p->slab;
    ++p->n_peers;
  }
  return QSIM_OK;
}

int qsim_comm_relayout_plan(int rank, int world, int n_local_qubits, int m, const int32_t* local_bits, const int32_t* global_bits,
                            int n_pieces, int32_t* out_n_pieces, int32_t* out_n_peers, int32_t* out_own_pattern,
                            int32_t* out_peers, uint64_t* out_slab_offsets, uint64_t* out_piece_amps) {
  RelayoutPlan p;
  int rc = relayout_plan(rank, world, n_local_qubits, m, local_bits, global_bits, n_pieces, &p);
  if (rc) return rc;
  if (out_n_pieces) *out_n_pieces = p.n_pieces;
  if (out_n_peers) *out_n_peers = p.n_peers;
  if (out_own_pattern) *out_own_pattern = p.own;
  for (int i = 0; i < p.n_peers; ++i) {
    if (out_peers) out_peers[i] = p.peers[i];
    if (out_slab_offsets) out_slab_offsets[i] = p.offs[i];
  }
  if (out_piece_amps) *out_piece_amps = p.part;
  return QSIM_OK;
}

// pipeline: pack piece s+1 (chunk stream) while piece s is on the links (transfer stream); unpack behind it.
// loopback: every peer is this rank itself (the slabs come back unchanged): the same packs, events, streams, RCCL
// groups and unpacks as a real re-layout of the planned rank, runnable on one GPU.
static int relayout_run(qsim_comm* cm, qsim_chunk* state, qsim_chunk* buf0, qsim_chunk* buf1, int m, const int32_t* local_bits,
                        const RelayoutPlan& p, bool loopback) {
  int rc = QSIM_OK;
  if (buf0->k != state->k || buf1->k != state->k || buf0->amp == buf1->amp || buf0->amp == state->amp || buf1->amp == state->amp)
    return fail(QSIM_ERR_INVALID, "re-layout: two distinct exchange buffers of the shard's size are needed");
  HIP_TRY(hipSetDevice(cm->device));
  if (!cm->xfer_stream) HIP_TRY(hipStreamCreateWithFlags(&cm->xfer_stream, hipStreamNonBlocking));
  for (auto& e : cm->ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  int32_t peers[7];
  for (int i = 0; i < p.n_peers; ++i) peers[i] = loopback ? cm->rank : p.peers[i];
  for (int s = 0; s < p.n_pieces; ++s) {
    if ((rc = slabs_all(state, m, local_bits, buf0, p.own, s, p.n_pieces, true, "qsim_comm_relayout"))) return rc;
    HIP_TRY(hipEventRecord(cm->ev[2 * s], state->stream));
    HIP_TRY(hipStreamWaitEvent(cm->xfer_stream, cm->ev[2 * s], 0));
    uint64_t so[7];
    for (int i = 0; i < p.n_peers; ++i) so[i] = p.offs[i] + (u64)s * p.part;
    if ((rc = comm_exchange(cm, p.n_peers, peers, buf0->amp, so, buf1->amp, so, p.part, cm->xfer_stream))) return rc;
    HIP_TRY(hipEventRecord(cm->ev[2 * s + 1], cm->xfer_stream));
  }
  for (int s = 0; s < p.n_pieces; ++s) {
    HIP_TRY(hipStreamWaitEvent(state->stream, cm->ev[2 * s + 1], 0));
    if ((rc = slabs_all(state, m, local_bits, buf1, p.own, s, p.n_pieces, false, "qsim_comm_relayout"))) return rc;
  }
  return QSIM_OK;
}

// All-to-all re-layout of THIS rank's shard: local bits `local_bits[i]` trade places with rank bits
// `global_bits[i]` (bit g of the rank = qubit k + g).  buf0 / buf1: exchange buffers of the shard's size.
int qsim_comm_relayout(qsim_comm* cm, qsim_chunk* state, qsim_chunk* buf0, qsim_chunk* buf1, int m,
                       const int32_t* local_bits, const int32_t* global_bits, int n_pieces) {
  int rc = check_comm(cm, "qsim_comm_relayout");
  if (rc || (rc = check_chunk(state, "qsim_comm_relayout")) || (rc = check_chunk(buf0, "qsim_comm_relayout")) ||
      (rc = check_chunk(buf1, "qsim_comm_relayout"))) return rc;
  RelayoutPlan p;
  if ((rc = relayout_plan(cm->rank, cm->world, state->k, m, local_bits, global_bits, n_pieces, &p))) return rc;
  return relayout_run(cm, state, buf0, buf1, m, local_bits, p, false);
}

// The pipeline of qsim_comm_relayout as rank `as_rank` of a world of `as_world` would run it, with every transfer
// looped back to this rank: the state is unchanged afterwards and buf1 holds the slabs that were "received".
int qsim_comm_relayout_loopback(qsim_comm* cm, qsim_chunk* state, qsim_chunk* buf0, qsim_chunk* buf1, int m,
                                const int32_t* local_bits, const int32_t* global_bits, int n_pieces, int as_rank, int as_world) {
  int rc = check_comm(cm, "qsim_comm_relayout_loopback");
  if (rc || (rc = check_chunk(state, "qsim_comm_relayout_loopback")) || (rc = check_chunk(buf0, "qsim_comm_relayout_loopback")) ||
      (rc = check_chunk(buf1, "qsim_comm_relayout_loopback"))) return rc;
  RelayoutPlan p;
  if ((rc = relayout_plan(as_rank, as_world, state->k, m, local_bits, global_bits, n_pieces, &p))) return rc;
  return relayout_run(cm, state, buf0, buf1, m, local_bits, p, true);
}

// The fused re-layout as ONE call for a host without the Python runner (what runner/distributed.py does with
// qsim_apply_ops_io + its own exchange): shard := after( re-layout( before(shard) ) ).  The last fused pass of `before`
// stores the slabs piece by piece (qsim_ops_io::dst_parts) on the chunk's stream; the exchange of piece j -- every peer in
// one RCCL group, all links busy -- runs on the communicator's transfer stream as soon as piece j is stored, while piece
// j + 1 is computed; the first pass of `after` reads the received slabs from `recv`.  Two HBM passes fewer than
// qsim_comm_relayout between two op lists, and the compute of all pieces but the first hidden behind the links.
int qsim_comm_relayout_fused(qsim_comm* cm, qsim_chunk* shard, qsim_chunk* send, qsim_chunk* recv,
                             const qsim_op_list* before, const qsim_op_list* after, int m, const int32_t* local_bits,
                             const int32_t* global_bits, int n_pieces, int as_rank, int as_world, int* n_passes) {
  int rc = check_comm(cm, "qsim_comm_relayout_fused");
  if (rc || (rc = check_chunk(shard, "qsim_comm_relayout_fused")) || (rc = check_chunk(send, "qsim_comm_relayout_fused")) ||
      (rc = check_chunk(recv, "qsim_comm_relayout_fused"))) return rc;
  if (send->stream != shard->stream || recv->stream != shard->stream)
    return fail(QSIM_ERR_INVALID, "qsim_comm_relayout_fused: the shard and both buffers must share a stream");
  const bool loopback = as_world != 0;
  RelayoutPlan p;
  if ((rc = relayout_plan(loopback ? as_rank : cm->rank, loopback ? as_world : cm->world, shard->k, m, local_bits, global_bits, 1, &p))) return rc;
  static const qsim_op_list none = {0, nullptr, nullptr, nullptr};
  if (!before) before = &none;
  if (!after) after = &none;
  if (n_pieces != 1 && n_pieces != 2 && n_pieces != 4 && n_pieces != 8 && n_pieces != -2 && n_pieces != -4 && n_pieces != -8)
    return fail(QSIM_ERR_INVALID, "qsim_comm_relayout_fused: n_pieces must be 1, 2, 4 or 8");
  HIP_TRY(hipSetDevice(cm->device));
  if (!cm->xfer_stream) HIP_TRY(hipStreamCreateWithFlags(&cm->xfer_stream, hipStreamNonBlocking));
  for (auto& e : cm->ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  qsim_ops_io io;
  std::memset(&io, 0, sizeof io);
  io.struct_size = sizeof io;
  io.dst = send; io.dst_m = m; io.dst_own = recv; io.own_pattern = p.own;
  for (int i = 0; i < m; ++i) io.dst_bits[i] = local_bits[i];
  io.dst_parts = n_pieces == 1 ? -1 : n_pieces;            // (always the split form: -1 = one piece)
  int passes_before = 0, passes_after = 0;
  if ((rc = qsim_apply_ops_io(shard, before->n_ops, before->nq, before->qubits, before->mats, &io, &passes_before))) return rc;
  int32_t n_parts = 0;
  uint64_t piece_amps = 0;
  struct Guard { qsim_chunk* c; bool armed; ~Guard() { if (armed) drop_pending(c); } } guard{shard, true};   // an error below leaves nothing pending
  if ((rc = qsim_apply_ops_io_parts(shard, &n_parts, &piece_amps, nullptr))) return rc;
  for (int j = 0; j < n_parts; ++j) {
    if ((rc = qsim_apply_ops_io_part(shard, j))) return rc;
    HIP_TRY(hipEventRecord(cm->ev[j], shard->stream));
    HIP_TRY(hipStreamWaitEvent(cm->xfer_stream, cm->ev[j], 0));
    int32_t peers[8];
    uint64_t offs[8];
    for (int i = 0; i < p.n_peers; ++i) {
      peers[i] = loopback ? cm->rank : p.peers[i];
      offs[i] = p.offs[i] + (u64)j * piece_amps;
    }
    if ((rc = comm_exchange(cm, p.n_peers, peers, send->amp, offs, recv->amp, offs, piece_amps, cm->xfer_stream))) return rc;
    HIP_TRY(hipEventRecord(cm->ev[8 + j], cm->xfer_stream));       // piece j has arrived
  }
  // receive side: `after` is planned now (the links are busy meanwhile) and takes the pieces over as they arrive -- its
  // first pass runs on the tiles whose pieces are there, the rest with the last piece
  std::memset(&io, 0, sizeof io);
  io.struct_size = sizeof io;
  io.src = recv; io.src_m = m; io.own_pattern = -1;
  for (int i = 0; i < m; ++i) io.src_bits[i] = local_bits[i];
  io.src_parts = n_pieces == 1 ? -1 : n_pieces;
  if ((rc = qsim_apply_ops_io(shard, after->n_ops, after->nq, after->qubits, after->mats, &io, &passes_after))) return rc;
  int32_t n_in = 0;
  if ((rc = qsim_apply_ops_io_source_parts(shard, &n_in, nullptr, nullptr))) return rc;
  if (n_in != n_parts) return fail(QSIM_ERR_INVALID, "internal: %d source pieces for %d sent ones", n_in, n_parts);
  for (int j = 0; j < n_parts; ++j) {
    HIP_TRY(hipStreamWaitEvent(shard->stream, cm->ev[8 + j], 0));
    if ((rc = qsim_apply_ops_io_load(shard, j))) return rc;
  }
  guard.armed = false;
  if (n_passes) *n_passes = passes_before + passes_after;
  return QSIM_OK;
}

// cpu_nonlocal.apply_2q_quad (cpu_nonlocal.py:61-67; chunk groups of four, single_node.py:315-321) with the four chunks
// on four ranks: ranks[j] holds chunk j = 2 bit(qa) + bit(qb) (the argument order c00, c01, c10, c11) and this rank is
// ranks[my_index].  The local index range is cut into four quarters and every taking-part rank WORKS ON some of them: it
// receives those quarters of its partners' shards into `buf`, applies the matrix across the copies and its own quarter,
// and sends the results back -- 3/4 of a shard each way, twice, instead of three whole shards in.  A chunk the matrix
// leaves alone (its row and column are the identity's: the |0x> chunks of a gate controlled by qa) takes no part: its
// rank returns at once, nobody sends to it or waits for it (two active chunks: a 2x2 across the pair, half a shard each
// way).  ranks = {r, r, r, r} with r = this rank is the one-GPU loopback form: every transfer comes back, so the gate
// acts on the shard's own four quarters (chunk j = quarter j: local qubits k - 1 and k - 2).
int qsim_apply_2q_quad_remote(qsim_comm* cm, qsim_chunk* shard, qsim_chunk* buf, const int32_t ranks[4], int my_index, const double U[32]) {
  const char* what = "qsim_apply_2q_quad_remote";
  int rc = check_comm(cm, what);
  if (rc || (rc = check_chunk(shard, what)) || (rc = check_chunk(buf, what))) return rc;
  if (!ranks || !U) return fail(QSIM_ERR_INVALID, "%s: null argument", what);
  if (buf->k != shard->k || buf->amp == shard->amp || buf->stream != shard->stream)
    return fail(QSIM_ERR_INVALID, "%s: the buffer must be a distinct chunk of the shard's size on the shard's stream", what);
  if (shard->k < 2) return fail(QSIM_ERR_INVALID, "%s: shards of at least 4 amplitudes are needed", what);
  if (my_index < 0 || my_index > 3) return fail(QSIM_ERR_INVALID, "%s: my_index must be 0..3", what);
  bool loopback = true;
  for (int j = 0; j < 4; ++j) {
    if (ranks[j] < 0 || ranks[j] >= cm->world) return fail(QSIM_ERR_INVALID, "%s: rank %d out of range", what, ranks[j]);
    loopback = loopback && ranks[j] == cm->rank;
  }
  if (!loopback) {
    if (ranks[my_index] != cm->rank) return fail(QSIM_ERR_INVALID, "%s: ranks[my_index] must be this rank", what);
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < j; ++i)
        if (ranks[i] == ranks[j]) return fail(QSIM_ERR_INVALID, "%s: the four chunks live on four different ranks (or all on this one: loopback)", what);
  }
  // chunks the matrix touches (a unitary touches none, or at least two; exactly three: treated as all four)
  int act[4], n_act = 0;
  bool active[4];
  for (int j = 0; j < 4; ++j) {
    bool unit = true;
    for (int c = 0; c < 4; ++c) {
      const double want = c == j ? 1.0 : 0.0;
      unit = unit && U[2 * (4 * j + c)] == want && U[2 * (4 * j + c) + 1] == 0.0 && U[2 * (4 * c + j)] == want && U[2 * (4 * c + j) + 1] == 0.0;
    }
    active[j] = !unit;
  }
  for (int j = 0; j < 4; ++j) n_act += active[j];
  if (n_act == 0) return QSIM_OK;
  if (n_act == 3) { n_act = 4; for (bool& a : active) a = true; }
  if (!active[my_index]) return QSIM_OK;
  if (n_act == 1) {                                        // a phase on one chunk (CZ, CR with both qubits global): no exchange
    const double f[8] = {U[2 * (5 * my_index)], U[2 * (5 * my_index) + 1], 0, 0, 0, 0, U[2 * (5 * my_index)], U[2 * (5 * my_index) + 1]};
    return qsim_apply_1q(shard, 0, f);
  }
  for (int j = 0, i = 0; j < 4; ++j) if (active[j]) act[i++] = j;
  HIP_TRY(hipSetDevice(cm->device));
  const u64 Q = amps(shard) >> 2;
  auto owner = [&](int q) { return act[q % n_act]; };      // the chunk whose rank works on quarter q
  auto slot_of = [&](int q, int j) -> u64 {                 // where partner chunk j's quarter q sits in MY buffer (I work on q)
    u64 s = 0;
    for (int qq = 0; qq < 4; ++qq) {
      if (owner(qq) != my_index) continue;
      for (int i = 0; i < n_act; ++i) {
        if (act[i] == my_index) continue;
        if (qq == q && act[i] == j) return s;
        ++s;
      }
    }
    return 0;                                               // (unreachable)
  };
  // One RCCL group per direction.  Between two ranks the k-th send meets the k-th receive: both sides walk the quarters in
  // ascending order (and, inside a quarter, the partners in ascending chunk order).
  auto exchange = [&](bool back) -> int {
    RCCL_TRY(g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int q = 0; q < 4 && bad == ncclSuccess; ++q) {
      const int o = owner(q);
      if (o == my_index) {                                  // partners' copies of quarter q: in (there) / out (back)
        for (int i = 0; i < n_act && bad == ncclSuccess; ++i) {
          if (act[i] == my_index) continue;
          double2* copy = buf->amp + slot_of(q, act[i]) * Q;
          bad = back ? g_rccl.Send(copy, 2 * Q, ncclDouble, ranks[act[i]], cm->comm, shard->stream)
                     : g_rccl.Recv(copy, 2 * Q, ncclDouble, ranks[act[i]], cm->comm, shard->stream);
        }
      } else {                                              // my quarter q: out to the rank that works on it / back in
        double2* mine = shard->amp + (u64)q * Q;
        bad = back ? g_rccl.Recv(mine, 2 * Q, ncclDouble, ranks[o], cm->comm, shard->stream)
                   : g_rccl.Send(mine, 2 * Q, ncclDouble, ranks[o], cm->comm, shard->stream);
      }
    }
    const ncclResult_t end = g_rccl.GroupEnd();             // (closed on every path)
    if (bad != ncclSuccess) return fail(QSIM_ERR_HIP, "%s: RCCL send / receive failed: %s", what, g_rccl.GetErrorString(bad));
    if (end != ncclSuccess) return fail(QSIM_ERR_HIP, "%s: ncclGroupEnd failed: %s", what, g_rccl.GetErrorString(end));
    return QSIM_OK;
  };
  if ((rc = exchange(false))) return rc;
  for (int q = 0; q < 4; ++q) {
    if (owner(q) != my_index) continue;
    qsim_chunk view[4];
    for (int j = 0; j < 4; ++j) {
      view[j] = *shard;                                     // device, stream, cache policy of the shard's allocation
      view[j].k = shard->k - 2;
      view[j].owns_memory = false; view[j].scratch = nullptr; view[j].have_events = false; view[j].pending = nullptr;
      view[j].amp = j == my_index ? shard->amp + (u64)q * Q : (active[j] ? buf->amp + slot_of(q, j) * Q : nullptr);
    }
    if (n_act == 4) {
      Group g = {{&view[0], &view[1], &view[2], &view[3]}, 4, shard->k - 2};
      if ((rc = gate_2q(g, shard->k - 1, shard->k - 2, U, shard->stream))) return rc;
    } else {                                                // two active chunks a < b: the 2x2 [[U_aa, U_ab], [U_ba, U_bb]] across the pair
      const int a = act[0], b = act[1];
      const double W[8] = {U[2 * (4 * a + a)], U[2 * (4 * a + a) + 1], U[2 * (4 * a + b)], U[2 * (4 * a + b) + 1],
                           U[2 * (4 * b + a)], U[2 * (4 * b + a) + 1], U[2 * (4 * b + b)], U[2 * (4 * b + b) + 1]};
      Group g = {{&view[a], &view[b], nullptr, nullptr}, 2, shard->k - 2};
      if ((rc = gate_1q(g, shard->k - 2, W, shard->stream))) return rc;
    }
  }
  return exchange(true);
}

// The reference's partner-chunk butterflies with the partner chunk on ANOTHER rank: both ranks call with each
// other's rank; `my_side` = this rank's value of the global qubit (0: this shard is c0, 1: it is c1).  The
// partner's whole shard is received into `buf` and the pair kernel updates this rank's shard (the copy in
// `buf` is scratch afterwards).
static int pair_remote(qsim_comm* cm, qsim_chunk* shard, qsim_chunk* buf, int partner, int my_side, const char* what) {
  int rc = check_comm(cm, what);
  if (rc || (rc = check_chunk(shard, what)) || (rc = check_chunk(buf, what))) return rc;
  if (buf->k != shard->k || buf->amp == shard->amp) return fail(QSIM_ERR_INVALID, "%s: the receive buffer must be a distinct chunk of the shard's size", what);
  if (partner < 0 || partner >= cm->world) return fail(QSIM_ERR_INVALID, "%s: partner rank %d out of range", what, partner);
  if (my_side != 0 && my_side != 1) return fail(QSIM_ERR_INVALID, "%s: my_side must be 0 or 1", what);
  HIP_TRY(hipSetDevice(cm->device));
  const int32_t peer = partner;
  const uint64_t zero = 0;
  return comm_exchange(cm, 1, &peer, shard->amp, &zero, buf->amp, &zero, amps(shard), shard->stream);
}

int qsim_apply_1q_pair_remote(qsim_comm* cm, qsim_chunk* shard, qsim_chunk* buf, int partner_rank, int my_side, const double U[8]) {
  int rc = pair_remote(cm, shard, buf, partner_rank, my_side, "qsim_apply_1q_pair_remote");
  if (rc) return rc;
  return my_side == 0 ? qsim_apply_1q_pair(shard, buf, U) : qsim_apply_1q_pair(buf, shard, U);
}

int qsim_apply_2q_pair_qa_local_remote(qsim_comm* cm, qsim_chunk* shard, qsim_chunk* buf, int partner_rank, int my_side, int qa, const double U[32]) {
  int rc = pair_remote(cm, shard, buf, partner_rank, my_side, "qsim_apply_2q_pair_qa_local_remote");
  if (rc) return rc;
  return my_side == 0 ? qsim_apply_2q_pair_qa_local(shard, buf, qa, U) : qsim_apply_2q_pair_qa_local(buf, shard, qa, U);
}

int qsim_apply_2q_pair_qb_local_remote(qsim_comm* cm, qsim_chunk* shard, qsim_chunk* buf, int partner_rank, int my_side, int qb, const double U[32]) {
  int rc = pair_remote(cm, shard, buf, partner_rank, my_side, "qsim_apply_2q_pair_qb_local_remote");
  if (rc) return rc;
  return my_side == 0 ? qsim_apply_2q_pair_qb_local(shard, buf, qb, U) : qsim_apply_2q_pair_qb_local(buf, shard, qb, U);
}

// Sparse export: the amplitudes with |re| > eps or |im| > eps as rows (index, re, im), ascending by index.
int qsim_count_nonzero(qsim_chunk* c, double eps, uint64_t* count) {
  int rc = check_chunk(c, "qsim_count_nonzero");
  if (rc) return rc;
  if (!count || !(eps >= 0)) return fail(QSIM_ERR_INVALID, "qsim_count_nonzero: bad arguments");
  if ((rc = ensure_scratch(c))) return rc;
  HIP_TRY(hipSetDevice(c->device));
  unsigned long long* dcount = reinterpret_cast<unsigned long long*>(c->scratch);
  HIP_TRY(hipMemsetAsync(dcount, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_count_kept, dim3(stream_grid(amps(c))), dim3(kBlock), 0, c->stream, c->amp, amps(c), eps, dcount);
  HIP_TRY(hipGetLastError());
  unsigned long long host = 0;
  HIP_TRY(hipMemcpyAsync(&host, dcount, sizeof host, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *count = host;
  return QSIM_OK;
}

int qsim_export_nonzero(qsim_chunk* c, double eps, uint64_t capacity, uint64_t* out_idx, double* out_re_im, uint64_t* n_rows) {
  int rc = check_chunk(c, "qsim_export_nonzero");
  if (rc) return rc;
  if (!n_rows || !(eps >= 0) || (capacity && (!out_idx || !out_re_im))) return fail(QSIM_ERR_INVALID, "qsim_export_nonzero: bad arguments");
  if ((rc = ensure_scratch(c))) return rc;
  HIP_TRY(hipSetDevice(c->device));
  u64* didx = nullptr;
  double2* damp = nullptr;
  if (capacity) {
    if (hipMalloc((void**)&didx, sizeof(u64) * capacity) != hipSuccess) return fail(QSIM_ERR_NOMEM, "qsim_export_nonzero: no device memory for %llu rows", (u64)capacity);
    if (hipMalloc((void**)&damp, sizeof(double2) * capacity) != hipSuccess) { (void)hipFree(didx); return fail(QSIM_ERR_NOMEM, "qsim_export_nonzero: no device memory for %llu rows", (u64)capacity); }
  }
  unsigned long long* cursor = reinterpret_cast<unsigned long long*>(c->scratch);
  auto cleanup = [&]() { if (didx) (void)hipFree(didx); if (damp) (void)hipFree(damp); };
  hipError_t e = hipMemsetAsync(cursor, 0, sizeof(unsigned long long), c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_append_kept, dim3(stream_grid(amps(c))), dim3(kBlock), 0, c->stream, c->amp, amps(c), eps, cursor, (u64)capacity, didx, damp);
    e = hipGetLastError();
  }
  unsigned long long total = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&total, cursor, sizeof total, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  const u64 got = std::min<u64>(total, capacity);
  std::vector<u64> idx(got);
  std::vector<double2> amp(got);
  if (e == hipSuccess && got) e = hipMemcpy(idx.data(), didx, sizeof(u64) * got, hipMemcpyDeviceToHost);
  if (e == hipSuccess && got) e = hipMemcpy(amp.data(), damp, sizeof(double2) * got, hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return fail(QSIM_ERR_HIP, "qsim_export_nonzero: %s", hipGetErrorString(e));
  *n_rows = total;
  if (total > capacity) return QSIM_OK;             // the caller sees n_rows > capacity and comes back with room (nothing written)
  std::vector<u64> order(got);
  for (u64 i = 0; i < got; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](u64 a, u64 b) { return idx[a] < idx[b]; });
  for (u64 i = 0; i < got; ++i) {
    out_idx[i] = idx[order[i]];
    out_re_im[2 * i] = amp[order[i]].x;
    out_re_im[2 * i + 1] = amp[order[i]].y;
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
static int bit_perm_from(const int32_t* log_to_phys, int n_total_qubits, BitPerm* perm, const char* what);

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
  if ((rc = bit_perm_from(log_to_phys, n_total_qubits, &perm, "qsim_max_abs_err_closed_form"))) return rc;
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

static int bit_perm_from(const int32_t* log_to_phys, int n_total_qubits, BitPerm* perm, const char* what) {
  std::memset(perm, 0, sizeof *perm);
  if (!log_to_phys) return QSIM_OK;
  u64 seen = 0;
  for (int q = 0; q < n_total_qubits; ++q) {
    const int ph = log_to_phys[q];
    if (ph < 0 || ph >= n_total_qubits || (seen >> ph) & 1) return fail(QSIM_ERR_INVALID, "%s: log_to_phys is not a permutation", what);
    seen |= 1ull << ph;
    perm->to_logical[ph] = (unsigned char)q;
    if (ph != q) perm->active = 1;
  }
  return QSIM_OK;
}

// sum over the chunk's amplitudes whose logical index passes the filter of amp * w(logical index): see k_fingerprint
int qsim_fingerprint(qsim_chunk* c, int n_total_qubits, uint64_t base_index, const int32_t* log_to_phys, uint64_t seed,
                     uint64_t sel_mask, uint64_t sel_value, double out[2]) {
  int rc = check_chunk(c, "qsim_fingerprint");
  if (rc) return rc;
  if (!out || n_total_qubits < c->k || n_total_qubits > 52) return fail(QSIM_ERR_INVALID, "qsim_fingerprint: bad arguments");
  if (base_index & (amps(c) - 1)) return fail(QSIM_ERR_INVALID, "qsim_fingerprint: base_index must be a multiple of the chunk length");
  if (n_total_qubits < 64 && ((base_index + amps(c) - 1) >> n_total_qubits)) return fail(QSIM_ERR_INVALID, "qsim_fingerprint: the chunk does not fit a state of %d qubits at that base", n_total_qubits);
  if (sel_value & ~sel_mask) return fail(QSIM_ERR_INVALID, "qsim_fingerprint: sel_value has bits outside sel_mask");
  BitPerm perm;
  if ((rc = bit_perm_from(log_to_phys, n_total_qubits, &perm, "qsim_fingerprint"))) return rc;
  if ((rc = ensure_scratch(c))) return rc;
  HIP_TRY(hipSetDevice(c->device));
  const unsigned grid = std::min<unsigned>(stream_grid(amps(c)), kReduceBlocks / 2);
  hipLaunchKernelGGL(k_fingerprint, dim3(grid), dim3(kBlock), 0, c->stream, c->amp, amps(c), n_total_qubits, (u64)base_index,
                     fp_mix((u64)seed), (u64)sel_mask, (u64)sel_value, c->scratch, perm);
  HIP_TRY(hipGetLastError());
  std::vector<double> host(2 * (size_t)grid);
  HIP_TRY(hipMemcpyAsync(host.data(), c->scratch, sizeof(double) * host.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  long double re = 0, im = 0;
  for (unsigned b = 0; b < grid; ++b) { re += host[2 * b]; im += host[2 * b + 1]; }
  out[0] = (double)re;
  out[1] = (double)im;
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
  HIP_TRY(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (g_profs.count(c->stream)) return fail(QSIM_ERR_INVALID, "a profile is already open on this chunk's stream");
  g_profs[c->stream];
  g_prof_open.fetch_add(1);
  return QSIM_OK;
}

int qsim_profile_end(qsim_chunk* c, int max_entries, int* n_entries, qsim_profile_entry* out) {
  int rc = check_chunk(c, "qsim_profile_end");
  if (rc) return rc;
  if (!n_entries || (!out && max_entries > 0)) return fail(QSIM_ERR_INVALID, "null output");
  std::vector<LaunchRecord> records;
  {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    auto it = g_profs.find(c->stream);
    if (it == g_profs.end()) return fail(QSIM_ERR_INVALID, "no profile is open on this chunk's stream");
    records.swap(it->second.records);
    g_profs.erase(it);
    g_prof_open.fetch_sub(1);
  }
  HIP_TRY(hipSetDevice(c->device));
  hipError_t sync = hipStreamSynchronize(c->stream);
  uint64_t launches[kNumClasses] = {0}, streaming[kNumClasses] = {0};
  double ms[kNumClasses] = {0}, bytes[kNumClasses] = {0}, hbm[kNumClasses] = {0};
  hipError_t bad = sync;
  for (LaunchRecord& r : records) {
    float t = 0.f;
    if (bad == hipSuccess) bad = hipEventElapsedTime(&t, r.e0, r.e1);
    launches[r.cls] += 1;
    streaming[r.cls] += r.streaming ? 1 : 0;
    ms[r.cls] += t;
    bytes[r.cls] += r.bytes;
    hbm[r.cls] += r.hbm_bytes;
  }
  {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (LaunchRecord& r : records) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
  }
  if (bad != hipSuccess) return fail(QSIM_ERR_HIP, "qsim_profile_end: %s", hipGetErrorString(bad));
  int n = 0;
  for (int cls = 0; cls < kNumClasses; ++cls) {
    if (!launches[cls]) continue;
    if (n < max_entries) {
      std::snprintf(out[n].kernel, sizeof out[n].kernel, "%s", kClassNames[cls]);
      out[n].launches = launches[cls];
      out[n].total_ms = ms[cls];
      out[n].algorithmic_bytes = bytes[cls];
      out[n].hbm_bytes = hbm[cls];
      out[n].streaming_launches = streaming[cls];
    }
    ++n;
  }
  *n_entries = n;
  return QSIM_OK;
}

}  // extern "C"
