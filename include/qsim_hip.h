/*
 * qsim_hip.h -- C ABI of libqsim_hip.so, the MI355X (gfx950) gate-application engine.
 *
 * This is the drop-in boundary for the gate-application hot path of
 * onofreiandrea/quantum_simulations.  The reference calls six free Python functions on
 * caller-owned numpy chunks (wenbo_engine/runner/single_node.py:25-28,103-106,208-321):
 *
 *     cpu_scalar.apply_1q(chunk, qubit, U)              wenbo_engine/kernel/cpu_scalar.py:21
 *     cpu_scalar.apply_2q(chunk, qa, qb, U)             wenbo_engine/kernel/cpu_scalar.py:35
 *     cpu_nonlocal.apply_1q_pair(c0, c1, U)             wenbo_engine/kernel/cpu_nonlocal.py:22
 *     cpu_nonlocal.apply_2q_pair_qa_local(c0,c1,qa,U)   wenbo_engine/kernel/cpu_nonlocal.py:29
 *     cpu_nonlocal.apply_2q_pair_qb_local(c0,c1,qb,U)   wenbo_engine/kernel/cpu_nonlocal.py:45
 *     cpu_nonlocal.apply_2q_quad(c00,c01,c10,c11,U)     wenbo_engine/kernel/cpu_nonlocal.py:61
 *
 * Here a "chunk" is an opaque handle to 2^k complex128 amplitudes resident in HBM
 * (interleaved re,im doubles; amplitude i at byte offset 16*i).  Conventions are the
 * reference's: qubit q <-> bit q of the chunk-local index (little-endian); a 2-qubit
 * matrix is row-major 4x4, big-endian inside the pair (row/col = 2*bit(qa)+bit(qb)).
 * Matrices are passed as interleaved (re,im) doubles: U[8] for 2x2, U[32] for 4x4.
 *
 * Every function returns 0 on success or a negative QSIM_ERR_* code; the message is
 * available from qsim_last_error() (thread-local).  Nothing throws across the ABI.
 * Calls on one handle are serialised by the caller; kernels are enqueued on the
 * handle's stream and only qsim_sync/qsim_download/qsim_norm2/qsim_time_end block.
 */
#ifndef QSIM_HIP_H
#define QSIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QSIM_OK            0
#define QSIM_ERR_INVALID  (-1)  /* bad argument (null handle, qa == qb, range, ...)        */
#define QSIM_ERR_NONLOCAL (-2)  /* qubit >= log2(chunk): the reference's                    */
                                /* NotImplementedError("... non-local ...")                 */
#define QSIM_ERR_HIP      (-3)  /* a HIP runtime call failed                                */
#define QSIM_ERR_NOMEM    (-4)  /* device allocation failed                                 */

typedef struct qsim_chunk qsim_chunk;

/* ---- library / device ------------------------------------------------------------ */
const char* qsim_last_error(void);
int qsim_version(void);
int qsim_device_count(int* count);

/* ---- chunk lifetime (replaces block_store.read_chunk/init_zero_state,
 *      wenbo_engine/storage/block_store.py:31-65: chunks live in HBM, not on disk) --- */
int qsim_create(int device, int n_local_qubits, qsim_chunk** out);
/* A window [offset, offset + 2^n_local_qubits) of an existing chunk (chunk c of a
 * 2^n state held in one allocation: offset = c << k, block_store.py:14-15).          */
int qsim_create_view(qsim_chunk* parent, uint64_t offset_amps, int n_local_qubits,
                     qsim_chunk** out);
/* Adopt caller-owned device memory (e.g. a torch tensor used for RCCL exchange);
 * `stream` is a hipStream_t (NULL = default stream).  Memory is never freed here.    */
int qsim_wrap(int device, void* device_ptr, int n_local_qubits, void* stream,
              qsim_chunk** out);
int qsim_destroy(qsim_chunk* c);
int qsim_n_local_qubits(const qsim_chunk* c);
void* qsim_device_ptr(const qsim_chunk* c);

/* ---- state I/O ------------------------------------------------------------------- */
int qsim_init_zero(qsim_chunk* c, int set_amp0);          /* |0..0> when set_amp0 != 0 */
int qsim_init_random(qsim_chunk* c, uint64_t seed);       /* normalised, counter-based  */
int qsim_upload(qsim_chunk* c, const double* re_im, uint64_t offset_amps, uint64_t count);
int qsim_download(qsim_chunk* c, double* re_im, uint64_t offset_amps, uint64_t count);
/* complex64 (re, im as floats) transfers: the reference's chunk files are complex64 (wenbo_engine/storage/
 * block_store.py:11); the conversion (round to nearest even) runs on the device, 8 bytes per amplitude cross PCIe. */
int qsim_download_c64(qsim_chunk* c, float* re_im, uint64_t offset_amps, uint64_t count);
int qsim_upload_c64(qsim_chunk* c, const float* re_im, uint64_t offset_amps, uint64_t count);
int qsim_copy(qsim_chunk* dst, const qsim_chunk* src);    /* device-to-device, same k  */
/* measurement aid (bench.py `stream_ceiling`): the same copy through a named path -- 0: qsim_copy's own choice, 1: the
 * non-temporal kernel, 2: the plain (cached) kernel, 3: hipMemcpyAsync device-to-device (the runtime's blit kernel) */
int qsim_copy_variant(qsim_chunk* dst, const qsim_chunk* src, int variant);

/* ---- local butterflies (cpu_scalar.apply_1q / apply_2q) --------------------------- */
int qsim_apply_1q(qsim_chunk* c, int qubit, const double U[8]);
int qsim_apply_2q(qsim_chunk* c, int qa, int qb, const double U[32]);
/* Dense k-qubit block (1 <= k <= 6): v3's fused block as a genuine 2^k x 2^k contraction -- `_apply_combined_matrix`,
 * v3_hisvsim_spark/src/parallel_gate_applicator.py:315-385.  M row-major 2^k x 2^k (re, im interleaved), M[out][in], pattern
 * bit i <-> qubits[i] (little-endian over the list, :169-204):  new[idx | out] = sum_in M[out][in] old[idx | in].  A block that
 * is a tensor product of 1q gates is cheaper as butterflies of a fused pass (qsim_apply_ops); this entry is for matrices
 * that are dense to begin with: k = 3 .. 6 run on the matrix cores (v_mfma_f64_16x16x4_f64 over 16 blocks at a time, in
 * place; the real image of the matrix in registers for k = 3, 4 and in LDS for k = 5, 6: 8 * 2^k flop per amplitude over
 * the same 32 bytes, so k <= 5 is bound by HBM and k = 6 by the matrix cores), k = 1, 2 on the pair kernels; chunks of fewer
 * than 2^(k+4) amplitudes take a one-workgroup-per-block form. */
int qsim_apply_fused_k(qsim_chunk* c, int k, const int32_t* qubits, const double* M);
/* A pass of n_ops gates in order (single_node._process_local_chunk, :208-216).
 * nq[i] in {1,2}; qubits[2*i], qubits[2*i+1]; mats + 32*i holds U (8 or 32 doubles). */
int qsim_apply_ops(qsim_chunk* c, int n_ops, const int32_t* nq, const int32_t* qubits,
                   const double* mats);
/* qsim_apply_ops groups the pass into fused LDS-tile launches (one HBM round trip for many
 * gates; op order is kept for every pair of ops that share a qubit).  The _unfused form
 * issues one kernel per op, in list order; qsim_last_pass_count reports how many HBM round
 * trips the last qsim_apply_ops on this chunk took.                                        */
int qsim_apply_ops_unfused(qsim_chunk* c, int n_ops, const int32_t* nq, const int32_t* qubits,
                           const double* mats);
int qsim_last_pass_count(const qsim_chunk* c);
int qsim_plan_cache_clear(void);   /* forget the cached pass images of earlier op lists: the next qsim_apply_ops* call plans anew */
/* qsim_apply_ops with the high tile bits of the first n_tiles fused passes named by the caller (bit b of tile_masks[p]: index
 * bit b is a tile bit of pass p) instead of searched for: a host that planned the list once (qsim_plan_ops: every pass image
 * carries its tile bits) and then moved its qubits to other index bits -- a layout chosen for the DRAM pattern of the tiles,
 * runner/engine.py -- gets the same passes on the new bits.  A mask that holds no op is ignored (the search takes over). */
int qsim_apply_ops_tiled(qsim_chunk* c, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                         int n_tiles, const uint64_t* tile_masks);

/* ---- op list with a re-layout fused into its ends ------------------------------------------------
 * Multi-GPU re-layouts (the staging SWAP lists of wenbo_engine/circuit/staging.py:136-152, merged into one
 * all-to-all) move the shard through exchange buffers in SLAB layout (qsim_pack_all below).  Instead of one extra
 * HBM pass before the exchange (pack) and one after it (unpack), the last fused pass of the op list BEFORE the
 * exchange stores its tiles straight into the send buffer in slab order, and the first pass of the op list AFTER it
 * loads them from the receive buffer: same arithmetic as qsim_apply_ops on the chunk, two passes fewer.
 *   src  != NULL: the state is read from `src`, which holds it in the slab layout over src_bits (qsim_pack_all
 *                 with skip_pattern -1); the chunk's own contents are ignored and overwritten.
 *   dst  != NULL: after the ops the state is left in `dst` in the slab layout over dst_bits -- except slab
 *                 own_pattern (>= 0), which is left at its place in `dst_own` (the receive buffer: that slab stays
 *                 on this rank); the chunk's own contents are unspecified afterwards.
 *                 dst_own may be the SOURCE buffer (a rank then needs three shard-sized buffers, not four: the source
 *                 has been consumed when a later kernel stores the own slab).  When ONE pass reads the source and
 *                 stores the slabs, the own slab goes into the chunk itself instead (at the same place): ask
 *                 qsim_apply_ops_io_own_slab after the call -- the exchange then delivers into the chunk, and the
 *                 chunk and the source buffer trade roles.
 * Parts that cannot be fused (a slab bit inside a 128-byte line, fewer qubits than a tile holds, no ops, a slab bit
 * among the tile bits of the last pass) run as separate slab passes; *n_passes reports the HBM passes made. */
typedef struct {
  uint32_t struct_size;  /* = sizeof(qsim_ops_io) of the header the caller was built against: a host built against another
                          * layout of this struct is refused (QSIM_ERR_INVALID) instead of being read past its end.  Zero the
                          * struct (memset), set struct_size, then the fields you use. */
  const qsim_chunk* src; int32_t src_m; int32_t src_bits[3];
  qsim_chunk* dst; int32_t dst_m; int32_t dst_bits[3];
  qsim_chunk* dst_own; int32_t own_pattern;
  int32_t dst_parts;   /* 0: the slabs are stored by this call.  P = 2, 4, 8 (split form, dst != NULL): this call launches
                        * everything but the storing of the slabs, which is cut into PIECES -- piece j = the j-th of 2^nb <= P
                        * equal contiguous sub-ranges of every slab (pieces keep >= 2^20 amplitudes, so small shards get
                        * fewer; the count depends only on the chunk size, m and P: qsim_split_piece_count, the same on
                        * every rank whatever its own pass plan) -- and the caller stores them with
                        * qsim_apply_ops_io_part(c, j), posting the exchange of piece j (all peers at once: every link
                        * busy) while later pieces are still computed.  The storing pass runs as partial launches when the
                        * piece bits are no tile bits of it (else as one launch with the first piece), or as qsim_pack_all
                        * pieces when it cannot be fused.  (-P: the same without the 2^20 floor: tests.) */
  int32_t src_parts;   /* 0: the source is complete when this call is made.  P = 2, 4, 8 (src != NULL): the source arrives in the
                        * pieces of the same rule (the sender's dst_parts = P over the same slab bits): this call plans the op
                        * list and launches NOTHING; the caller announces every piece with qsim_apply_ops_io_load(c, j) once
                        * its transfer is ordered on the chunk's stream.  The first pass runs as partial launches over the
                        * tiles whose source pieces are there (when the top piece bits are no tile bits of it; else whole,
                        * with the last piece), everything else of the list with the last piece -- the receive side of a
                        * fused re-layout overlaps its first pass with the links.  With dst_parts too, the slab pieces of
                        * the destination are stored (qsim_apply_ops_io_part) after the last load. */
  int32_t n_tiles;     /* > 0: the high tile bits of the first n_tiles fused passes are named by the caller (tile_masks[p], bit b:
                        * index bit b is a tile bit of pass p), as in qsim_apply_ops_tiled: a host that planned stage boundaries
                        * and passes together (qsim_plan_peek_pass) gets the passes it planned on every rank. */
  const uint64_t* tile_masks;
} qsim_ops_io;
int qsim_apply_ops_io(qsim_chunk* c, int n_ops, const int32_t* nq, const int32_t* qubits,
                      const double* mats, const qsim_ops_io* io, int* n_passes);
int qsim_apply_ops_io_own_slab(const qsim_chunk* c, int32_t* in_chunk);   /* 1: the last call left the own slab in the chunk */
int qsim_apply_ops_io_load(qsim_chunk* c, int part);
int qsim_apply_ops_io_source_parts(const qsim_chunk* c, int32_t* n_parts, uint64_t* piece_amps, int32_t* n_launches);
/* Pieces of the pending split call: piece j of EVERY slab d is [d * 2^(k - m) + j * piece_amps, + piece_amps) of the send /
 * receive buffers; n_launches = the partial launches this chunk's pass is cut into (<= n_parts; 0: stored already). */
int qsim_apply_ops_io_parts(const qsim_chunk* c, int32_t* n_parts, uint64_t* piece_amps, int32_t* n_launches);
int qsim_apply_ops_io_part(qsim_chunk* c, int part);
int qsim_split_piece_count(int n_local_qubits, int m, int dst_parts);   /* the piece rule as a pure function */
/* The host planner of the fused passes WITHOUT a device (used by the CPU tests): plans the op list
 * for a 2^n_local_qubits chunk and writes one QSIM_PASS_IMAGE_BYTES pass image per planned pass to `out`
 * (layout = the kernel-argument block of k_tile, csrc/tile_kernel.h: record count, tile size T, tile high
 * bits, then the record stream the gate engine interprets).  `out` may be NULL to count passes only. */
#define QSIM_PASS_IMAGE_BYTES 4096
int qsim_plan_ops(int n_local_qubits, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                  void* out, uint64_t out_capacity_bytes, int32_t* n_passes);
int qsim_plan_ops_tiled(int n_local_qubits, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                        int n_tiles, const uint64_t* tile_masks, void* out, uint64_t out_capacity_bytes, int32_t* n_passes);

/* The next fused pass of a partly executed op list on a PARTITIONED state, without a device (the pass builder as the
 * partition planner sees it: runner/partition_plan.py, which replaces the per-stage view of wenbo_engine/circuit/
 * staging.py:447-519 `_local_sets_to_steps`).  Qubits are index bits of the whole state; bits >= n_local_qubits are rank
 * bits: fine as controls / phase bits, never tile bits or targets (an op that targets one waits and blocks its
 * dependents).  done[i] != 0: op i ran already.  Out: the tile the builder would choose now (tile_mask: 8 high bits incl.
 * the fill, which leaves out avoid_mask where it can; need_mask: those its ops need) and the ops it would hold (members,
 * capacity n_ops); *n_members = 0 when everything left waits for a rank bit.  hint_mask != 0 names the tile. */
int qsim_plan_peek_pass(int n_local_qubits, int n_total_qubits, int n_ops, const int32_t* nq, const int32_t* qubits,
                        const double* mats, const uint8_t* done, uint64_t avoid_mask, uint64_t hint_mask, uint64_t* tile_mask,
                        uint64_t* need_mask, int32_t* n_members, int32_t* members);

/* Pass counts of ONE op list under n_layouts qubit layouts (layouts[l * n_local_qubits + q] = index bit of logical qubit q),
 * planned in parallel on n_threads host threads, no device: the pass builder's result depends on which qubits live on the
 * three line bits (they belong to every tile), so a host that is free to choose the layout tries several. */
int qsim_plan_count_layouts(int n_local_qubits, int n_ops, const int32_t* nq, const int32_t* qubits, const double* mats,
                            int n_layouts, const int32_t* layouts, int32_t* n_passes, int n_threads);

/* The qubit layout search of runner/tile_layout.py as host code: simulated annealing over the assignment qubit -> index bit
 * (bits 0..2 stay) minimising sum over the passes of the caller's cost model of their tile-bit sets (see the .hip for the
 * model's form).  tile_masks[p] = high tile bits of pass p as logical qubits; out_l2p[q] = index bit for qubit q. */
int qsim_choose_layout(int n_local_qubits, int n_tiles, const uint64_t* tile_masks, int top_bit, const double* bit_cost,
                       const double* pair_cost, const double* triple_cost /* nb^3, a < b < c; may be NULL */, uint64_t seed,
                       int sweeps, int32_t* out_l2p, double* cost_identity, double* cost_chosen);

/* ---- partner-chunk butterflies (cpu_nonlocal.*) ----------------------------------- */
int qsim_apply_1q_pair(qsim_chunk* c0, qsim_chunk* c1, const double U[8]);
int qsim_apply_2q_pair_qa_local(qsim_chunk* c0, qsim_chunk* c1, int qa, const double U[32]);
int qsim_apply_2q_pair_qb_local(qsim_chunk* c0, qsim_chunk* c1, int qb, const double U[32]);
int qsim_apply_2q_quad(qsim_chunk* c00, qsim_chunk* c01, qsim_chunk* c10, qsim_chunk* c11,
                       const double U[32]);

/* ---- exchange helpers for the multi-GPU runner ------------------------------------- */
/* dst[j] = src[i] for the 2^(k-1) amplitudes i of src with bit `bit` == value, in
 * ascending order (pack), and the inverse scatter (unpack).  `buf` holds k-1 qubits.   */
int qsim_pack_half(const qsim_chunk* src, int bit, int value, qsim_chunk* buf);
int qsim_unpack_half(qsim_chunk* dst, int bit, int value, const qsim_chunk* buf);
/* Slab form for the all-to-all re-layout that replaces a staging SWAP list
 * (wenbo_engine/circuit/staging.py:136-152): the 2^(k-m) amplitudes of `src` whose index bits
 * bits[0..m) equal the bits of `pattern` (bit i of pattern <-> bits[i]), ascending, to/from
 * buf[buf_offset_amps ...).  1 <= m <= 3.                                               */
int qsim_pack_bits(const qsim_chunk* src, int m, const int32_t* bits, int pattern,
                   qsim_chunk* buf, uint64_t buf_offset_amps);
int qsim_unpack_bits(qsim_chunk* dst, int m, const int32_t* bits, int pattern,
                     const qsim_chunk* buf, uint64_t buf_offset_amps);
/* All 2^m slabs in one pass over the shard: slab p lands at buf[p * 2^(k-m) ...) (`buf` holds k
 * qubits); the slab `skip_pattern` (the part that stays on this rank; -1 = none) is left out.
 * Whole 128-B lines are read and written for every choice of bits, unlike the per-pattern form
 * when a bit is below 3.  `piece` of `n_pieces` (1, 2, 4 or 8) restricts the call to the same
 * contiguous 1/n_pieces sub-range of every slab, so runner/distributed.py can overlap packing,
 * the RCCL exchange and unpacking piece by piece.                                          */
int qsim_pack_all(const qsim_chunk* src, int m, const int32_t* bits, qsim_chunk* buf, int skip_pattern,
                  int piece, int n_pieces);
int qsim_unpack_all(qsim_chunk* dst, int m, const int32_t* bits, const qsim_chunk* buf, int skip_pattern,
                    int piece, int n_pieces);

/* All-to-all re-layout among the 2^g chunks of ONE device (chunks[c] = chunk index c): swaps
 * local qubit local_bits[i] with chunk-index bit global_bits[i] for i < m (m <= 3) -- the merged
 * form of a staging SWAP list ([p_out < k, p_in >= k], SWAP), staging.py:136-152.  Across GPUs:
 * qsim_comm_relayout below (the Python runner drives the same kernels over its own process group). */
int qsim_swap_global_local(qsim_chunk* const* chunks, int n_chunks, const int32_t* global_bits,
                           const int32_t* local_bits, int m);

/* ---- multi-GPU reach: RCCL (xGMI) inside the library --------------------------------------------
 * One process per GPU, shard g = amplitudes [g * 2^k, (g+1) * 2^k) (the reference's chunk g,
 * block_store.py:14-15), so a gate on qubit q >= k pairs rank g with g XOR 2^(q-k) exactly like the
 * partner chunks of cpu_nonlocal.py:7-15 / single_node.py:271-321.  Rank 0 makes a unique id and the host
 * program hands the 128 bytes to the other ranks (any transport); every rank then creates its
 * communicator.  RCCL is loaded with dlopen at the first of these calls: the rest of the ABI works
 * without it.  All calls are enqueued on the chunks' stream (qsim_sync to wait).                   */
#define QSIM_COMM_ID_BYTES 128
typedef struct qsim_comm qsim_comm;
int qsim_comm_get_unique_id(uint8_t id[QSIM_COMM_ID_BYTES]);
int qsim_comm_init(int device, int rank, int world, const uint8_t id[QSIM_COMM_ID_BYTES], qsim_comm** out);
int qsim_comm_destroy(qsim_comm* comm);
int qsim_comm_rank(const qsim_comm* comm);
int qsim_comm_world(const qsim_comm* comm);
/* One grouped exchange: for every i < n_peers send `count_amps` amplitudes of `send` starting at
 * send_off[i] to rank peers[i] and receive as many from it into `recv` at recv_off[i].            */
/* (send and recv must be chunks on the SAME stream: the transfer is ordered on it) */
int qsim_comm_exchange(qsim_comm* comm, int n_peers, const int32_t* peers, const qsim_chunk* send,
                       const uint64_t* send_off, qsim_chunk* recv, const uint64_t* recv_off, uint64_t count_amps);
/* Background form of qsim_comm_exchange: the group runs on the communicator's own transfer stream, behind everything
 * queued on the chunks' stream so far and BESIDE what is queued on it later (the next piece of a fused re-layout being
 * computed); qsim_comm_join makes a chunk's stream wait for every background transfer posted so far. */
int qsim_comm_exchange_bg(qsim_comm* comm, int n_peers, const int32_t* peers, const qsim_chunk* send,
                          const uint64_t* send_off, qsim_chunk* recv, const uint64_t* recv_off, uint64_t count_amps,
                          uint32_t* ticket);                 /* ticket: may be NULL */
int qsim_comm_join(qsim_comm* comm, qsim_chunk* c);
/* ... or only for the exchange `ticket` names (and those posted before it): the pieces of a re-layout are consumed as they
 * arrive (qsim_apply_ops_io_load), later pieces still on the links. */
int qsim_comm_wait(qsim_comm* comm, qsim_chunk* c, uint32_t ticket);
/* qsim_swap_global_local ACROSS GPUs: local qubit local_bits[i] of this rank's shard trades places
 * with rank bit global_bits[i] (qubit k + global_bits[i]) for i < m <= 3 -- the merged all-to-all form
 * of a staging SWAP list (staging.py:136-152).  Every rank of the communicator calls it with the same
 * arguments.  buf0 / buf1: two exchange buffers of the shard's size.  Packing, the RCCL exchange with
 * the 2^m - 1 peers (all links at once) and unpacking are pipelined over n_pieces (1, 2, 4, 8) sub-ranges.
 * (1 - 2^-m) of a shard crosses the links per rank.                                                */
int qsim_comm_relayout(qsim_comm* comm, qsim_chunk* state, qsim_chunk* buf0, qsim_chunk* buf1, int m,
                       const int32_t* local_bits, const int32_t* global_bits, int n_pieces);
/* The schedule of qsim_comm_relayout as a pure function (no GPU, no communicator): for rank `rank` of `world` it
 * returns the pieces really used (pieces keep >= 2^20 amplitudes), this rank's own pattern (the slab that stays), and
 * per peer its rank and the amplitude offset of its slab in the send AND the receive buffer; piece s of a slab is
 * [offset + s * piece_amps, + piece_amps).  Output arrays hold up to 7 entries.  The pairing is that of the
 * reference's partner groups (wenbo_engine/runner/single_node.py:222-245) with one chunk per rank. */
int qsim_comm_relayout_plan(int rank, int world, int n_local_qubits, int m, const int32_t* local_bits,
                            const int32_t* global_bits, int n_pieces, int32_t* out_n_pieces, int32_t* out_n_peers,
                            int32_t* out_own_pattern, int32_t* out_peers, uint64_t* out_slab_offsets,
                            uint64_t* out_piece_amps);
/* The same pipeline (packs, events, second stream, RCCL groups, unpacks) as rank `as_rank` of `as_world` would run
 * it, every transfer looped back to this rank: runnable on ONE GPU; the state is unchanged afterwards and buf1 holds
 * the slabs that were "received". */
int qsim_comm_relayout_loopback(qsim_comm* comm, qsim_chunk* state, qsim_chunk* buf0, qsim_chunk* buf1, int m,
                                const int32_t* local_bits, const int32_t* global_bits, int n_pieces, int as_rank,
                                int as_world);
/* cpu_nonlocal.apply_1q_pair / apply_2q_pair_qa_local / apply_2q_pair_qb_local (cpu_nonlocal.py:22-58)
 * with the partner chunk on rank `partner_rank` (both ranks call, naming each other): the partner's
 * shard is received into `buf` and this rank's shard is updated.  my_side = this rank's value of the
 * global qubit: 0 -> the shard is c0, 1 -> it is c1.  One shard crosses one xGMI link each way; a host
 * that may change the qubit layout uses qsim_comm_relayout with m = 1 instead (half a shard, no return). */
int qsim_apply_1q_pair_remote(qsim_comm* comm, qsim_chunk* shard, qsim_chunk* buf, int partner_rank, int my_side,
                              const double U[8]);
int qsim_apply_2q_pair_qa_local_remote(qsim_comm* comm, qsim_chunk* shard, qsim_chunk* buf, int partner_rank,
                                       int my_side, int qa, const double U[32]);
int qsim_apply_2q_pair_qb_local_remote(qsim_comm* comm, qsim_chunk* shard, qsim_chunk* buf, int partner_rank,
                                       int my_side, int qb, const double U[32]);

/* cpu_nonlocal.apply_2q_quad (cpu_nonlocal.py:61-67; the chunk groups of four of single_node.py:315-321) with the four
 * chunks on four ranks: ranks[j] holds chunk j = 2 bit(qa) + bit(qb) (argument order c00, c01, c10, c11), this rank is
 * ranks[my_index]; all taking-part ranks call with the same ranks[] and U.  Each rank works on a quarter (a half, when the
 * matrix touches only two chunks) of the local index range for the whole group: 3/4 of a shard crosses the links each way
 * and back instead of three shards in; a chunk the matrix leaves alone takes no part and moves nothing.  `buf`: scratch
 * of the shard's size.  ranks = {r, r, r, r} (r = this rank) is the one-GPU loopback form: with four chunks taking part
 * the gate then acts on the shard's own quarters (chunk j = quarter j: local qubits k - 1, k - 2), with two as their 2x2
 * on local qubit k - 2, with one as its phase on the whole shard.  A host that may change the qubit layout uses the re-layout (m = 2)
 * and a local gate instead: 3/4 of a shard once, nothing back. */
int qsim_apply_2q_quad_remote(qsim_comm* comm, qsim_chunk* shard, qsim_chunk* buf, const int32_t ranks[4], int my_index,
                              const double U[32]);
/* The fused re-layout as one call (what runner/distributed.py does with qsim_apply_ops_io and its own exchange):
 *     shard := after( re-layout( before(shard) ) )
 * The last fused pass of `before` stores the slabs piece by piece (qsim_ops_io::dst_parts, up to n_pieces = 1, 2, 4, 8) into
 * `send` (own slab into `recv`); the exchange of piece j with all 2^m - 1 peers (one RCCL group: every link busy) runs on
 * the communicator's transfer stream as soon as piece j is stored, while piece j + 1 is computed; the first pass of `after`
 * reads the received slabs from `recv` and leaves the state in `shard` in index order.  Two HBM passes fewer than
 * qsim_comm_relayout between two op lists.  before / after may be NULL or empty (then a pack / unpack pass is made).
 * as_world != 0: the schedule of rank as_rank in a world of as_world with every transfer looped back to this rank
 * (runnable on ONE GPU: the result is after(before(shard)) exactly).  Same arguments on every rank of the communicator;
 * the SWAP-list semantics are those of qsim_swap_global_local (staging.py:136-152). */
typedef struct { int32_t n_ops; const int32_t* nq; const int32_t* qubits; const double* mats; } qsim_op_list;
int qsim_comm_relayout_fused(qsim_comm* comm, qsim_chunk* shard, qsim_chunk* send, qsim_chunk* recv,
                             const qsim_op_list* before, const qsim_op_list* after, int m, const int32_t* local_bits,
                             const int32_t* global_bits, int n_pieces, int as_rank, int as_world, int* n_passes);

/* ---- synchronisation, reductions, timing ------------------------------------------- */
/* ---- sparse view (the v3 worker's rows; v3_hisvsim_spark parallel_gate_applicator.py:372-374, state_manager.py:95-106) ---
 * The amplitudes with |re| > eps or |im| > eps as rows (index, re, im), ascending by index, selected ON THE DEVICE:
 * qsim_count_nonzero counts them; qsim_export_nonzero writes up to `capacity` rows into host arrays and returns the
 * total in *n_rows (when it exceeds the capacity nothing is written: come back with room, or download the dense
 * state). */
int qsim_count_nonzero(qsim_chunk* c, double eps, uint64_t* count);
int qsim_export_nonzero(qsim_chunk* c, double eps, uint64_t capacity, uint64_t* out_idx, double* out_re_im,
                        uint64_t* n_rows);
int qsim_sync(qsim_chunk* c);
int qsim_norm2(qsim_chunk* c, double* out);               /* sum |amp|^2               */
/* max_i |amp_i - expected_i| for closed-form states, evaluated on the device:
 * kind 0: GHZ (1/sqrt2 at local index 0 of the first chunk and at the last index of the
 *         last), kind 1: GHZ+QFT  2^-(n+1)/2 (1 + exp(-2 pi i y / 2^n)), y = base + i.  */
int qsim_max_abs_err_closed_form(qsim_chunk* c, int kind, int n_total_qubits,
                                 uint64_t base_index, double* out);
/* Same for a staged layout: logical qubit q sits at physical index bit log_to_phys[q]
 * (atlas_stages' second return value, staging.py:587-634); NULL = identity.             */
int qsim_max_abs_err_closed_form_perm(qsim_chunk* c, int kind, int n_total_qubits,
                                      uint64_t base_index, const int32_t* log_to_phys,
                                      double* out);
/* Layout-aware fingerprint of a (partitioned, staged) state, evaluated on the device:
 *     out = sum over the chunk's amplitudes i whose LOGICAL index y passes (y & sel_mask) == sel_value of amp_i * w(y)
 * with y = the logical index of physical index base_index + i under log_to_phys (NULL = identity; the layout of
 * atlas_stages / permute_state, staging.py:587-658) and w(y) a counter-based pseudo-random complex weight: a = mix(y ^
 * mix(seed)), b = mix(a) (mix = one splitmix64 round), w = ((a >> 11) 2^-52 - 1) + i ((b >> 11) 2^-52 - 1).  The same
 * amplitudes give the same sum (up to summation order) wherever they live: shard g of a multi-GPU run in its staged
 * layout against the same index set of a one-GPU run of the circuit (ref_dense.simulate order, ref_dense.py:44-57)
 * selected with sel_mask / sel_value -- an amplitude-level check of states too large to gather (n = 33: 128 GiB). */
int qsim_fingerprint(qsim_chunk* c, int n_total_qubits, uint64_t base_index, const int32_t* log_to_phys, uint64_t seed,
                     uint64_t sel_mask, uint64_t sel_value, double out[2]);
int qsim_time_begin(qsim_chunk* c);                        /* hipEventRecord on stream  */
int qsim_time_end(qsim_chunk* c, float* elapsed_ms);       /* record + synchronize      */


/* ---- per-launch timing for roofline reports (bench.py) ------------------------------- */
/* Between begin and end every gate kernel launched on c's stream is bracketed by HIP events
 * (no synchronisation, no extra kernels).  One profile per STREAM: handles on different streams (qsim_wrap) may be
 * profiled at the same time from different host threads; a second begin on a stream whose profile is open fails.  qsim_profile_end synchronises the stream and
 * returns, per kernel class, the launch count, the summed event time, the summed
 * algorithmic bytes (per gate-application 32 B per amplitude the gate touches,
 * summed over the gates of a launch -- a fused pass counts every gate it applies) and
 * hbm_bytes, the bytes the launches themselves had to move (32 B per amplitude touched once);
 * streaming_launches counts the launches that ran the non-temporal (Infinity-Cache bypassing)
 * instantiation of their kernel -- the cache policy follows the size of the ALLOCATION a chunk
 * lives in (a small view of a large parent streams), and tests assert on it. */
typedef struct {
  char kernel[48];
  uint64_t launches;
  double total_ms;
  double algorithmic_bytes;
  double hbm_bytes;
  uint64_t streaming_launches;
} qsim_profile_entry;
int qsim_profile_begin(qsim_chunk* c);
int qsim_profile_end(qsim_chunk* c, int max_entries, int* n_entries, qsim_profile_entry* out);

#ifdef __cplusplus
}
#endif
#endif /* QSIM_HIP_H */
