// gate_plan.h -- host side of the per-gate kernels: plan, tuning knobs, launch profile, launch heuristics.
// Part of the single translation unit qsim_hip.hip (included there, in order; not a standalone header).
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

// Tunables (environment overrides exist in the probe build only).
#ifndef QSIM_TILE_HAD_DEFAULT
#define QSIM_TILE_HAD_DEFAULT 1
#endif
struct Tuning {
  int swz_cut = 64;   // XCD-contiguous block order when the highest removed bit is below this (r01 scan: always)
  int force_nt = -1;  // -1 auto, 0 never, 1 always
  int items = 0;      // 0 auto
  int max_gates_per_pass = 128;
  int plan_lookahead = -1;   // tile-bit look-ahead of the pass builder: -1 auto (>= 24 qubits; two passes deep from 26), 0 off, 1 on, 2 two deep
  int tile_special = 1;      // real / Y-like / -1 / +-i special-case opcodes in fused passes
  int tile_merge_diag = 1;   // merge phase gates that share their predicate (OPC_DIAGR)
  int tile_direct = 1;       // first / last register group above the line bits: global <-> registers without LDS (OPC_GROUP_DIRECT / OPC_END_DIRECT)
  int tile_last_search = 1;  // last register group chosen from the end of the pass among the triples above the line bits; value = fewest ops worth it
  int tile_commute_fuse = 0; // (off: neutral over 8 circuits -- 1 % less per pass, one pass more -- profiles/r02y_ab_commute_fusion.txt)
                             // a 1q gate moves past ops it commutes with and merges with its neighbour 1q gate on the qubit: 0 off, 1 forward (the earlier gate moves), 2 backward (the later one moves: fewer passes)
  int tile_mux = 1;          // a controlled gate with its control outside the tile + the 1q gate next to it on its target -> two predicated 2x2 records
  int plan_conflict_cost = -1; // pass builder, bank / row conflict pairs of tile bits: > 0 = ops a pass must hold more to be worth one (costs passes), -1 = break ties only (-1.9 % over 8 circuits), 0 = ignore
  int plan_commute = 2;      // (1: diagonal qubits only, 2: also X-type targets) pass builder: an op that has to wait blocks a qubit it acts on DIAGONALLY (control, phase bit) only for ops that
                             // act on it non-diagonally -- ops that are diagonal on every shared qubit commute, so later ones may pass it
  int plan_jitter = 0;       // probe: > 0 seeds pseudo-random tie-breaks (and losses of up to two ops) in the pass builder's growth choices
  int plan_force_low = 0;    // probe: the lowest N index bits above the line bits (3 .. 3 + N - 1) are tile bits of EVERY pass (2^(N+7)-byte
                             // contiguous pieces per tile: full DRAM rows on the write side, profiles/r03e_perm_windows_28q.txt)
  int plan_scan_window = 384; // pass builder: ops (not yet done) behind the first waiting op that a pass may take its ops from (bounds the host time per pass on long lists)
  int plan_anchor = 0;       // probe (round 4 evaluation): every tile shares >= N high bits with the tile of the pass before it
  int tile_sink_swaps = 1;   // X / CNOT that nothing later in their group touches: swap LDS addresses at write-back (OPC_ASWAP1)
  int tile_group_search = 1; // register groups: try every triple of pending target bits, not only first come
  int tile_had = QSIM_TILE_HAD_DEFAULT;          // uncontrolled c [[1,1],[1,-1]] as add/sub butterflies + one scale per pass (OPC_HAD1 / OPC_SCALE)
  int dense_form = 1;        // (probe build) qsim_apply_fused_k, k = 3, 4: 1 = k_dense_mfma2 (round 5: whole amplitudes per lane), 0 = the round-4 kernels chosen by dense_mfma; QSIM_DENSE_FORM
  int dense_pf = -1;         // (probe build) k_dense_mfma2's PF (0 none, 1 conditional, 2 unconditional next-group request); QSIM_DENSE_PF
  int dense_groups = 0;      // (probe build) column groups per wave, k <= 4 (product: 4); QSIM_DENSE_GROUPS
  int dense_wgs = 0;         // (probe build) resident workgroups per CU the k = 5 grid asks for (product: 4); QSIM_DENSE_WGS
  int dense_consec = 0;      // (probe build) k <= 4: a wave's column groups are consecutive (else strided through its XCD's region); QSIM_DENSE_CONSEC
  long long dense_skew = 0;  // (probe build) XCD x starts x * skew column groups into its region; QSIM_DENSE_SKEW
  int dense_mfma = 3;        // qsim_apply_fused_k: bit 0: k = 3, bit 1: k = 4 on the matrix cores (k_dense_mfma; else the vector-ALU k_dense; probe knob QSIM_DENSE_MFMA)
  int debug_skip_gates = 0;  // probe build only: QSIM_DEBUG_SKIP_GATES=1: tile passes move data but apply nothing (WRONG results)
  int tile_order = -1;       // probe build only: QSIM_TILE_ORDER=0/1/2 forces the tile order of k_tile
  int debug_stats = 0;       // probe build only: QSIM_DEBUG_STATS=1: print gates / groups per pass to stderr; 2: time every pass
  // States up to this size stay in the 256 MiB Infinity Cache between launches when accessed with
  // the default cache policy (tools/mall_probe.hip: 8.5-8.8 TB/s r+w for a 128-256 MiB region vs
  // 5.5 streaming); the NT policy bypasses it (6.0-6.2 at every size), so NT is for larger states.
  u64 mall_bytes = 256ull << 20;
  Tuning() {
#ifdef QSIM_PROBES
    // Every knob lives in the probe build only (`make probes` -> libqsim_hip_probes.so, loaded by tools/ and by the
    // planner tests through QSIM_LIBRARY): the product library reads nothing from the environment, so a plan -- and a
    // timed number -- never depends on it.  Planning knobs: every setting yields a correct program
    // (tests/test_tile_planner_cpu.py plans and interprets under each of them).
    if (const char* e = getenv("QSIM_DEBUG_STATS")) debug_stats = atoi(e);
    if (const char* e = getenv("QSIM_PASS_GATES")) max_gates_per_pass = std::max(1, atoi(e));
    if (const char* e = getenv("QSIM_PLAN_LOOKAHEAD")) plan_lookahead = atoi(e);
    if (const char* e = getenv("QSIM_TILE_SPECIAL")) tile_special = atoi(e);
    if (const char* e = getenv("QSIM_TILE_MERGE_DIAG")) tile_merge_diag = atoi(e);
    if (const char* e = getenv("QSIM_TILE_HAD")) tile_had = atoi(e);
    if (const char* e = getenv("QSIM_TILE_GROUP_SEARCH")) tile_group_search = atoi(e);
    if (const char* e = getenv("QSIM_TILE_SINK_SWAPS")) tile_sink_swaps = atoi(e);
    if (const char* e = getenv("QSIM_TILE_DIRECT")) tile_direct = atoi(e);
    if (const char* e = getenv("QSIM_PLAN_CONFLICT_COST")) plan_conflict_cost = atoi(e);
    if (const char* e = getenv("QSIM_PLAN_COMMUTE")) plan_commute = atoi(e);
    if (const char* e = getenv("QSIM_PLAN_JITTER")) plan_jitter = atoi(e);
    if (const char* e = getenv("QSIM_PLAN_FORCE_LOW")) plan_force_low = std::max(0, std::min(6, atoi(e)));
    if (const char* e = getenv("QSIM_PLAN_SCAN_WINDOW")) plan_scan_window = std::max(1, atoi(e));
    if (const char* e = getenv("QSIM_DENSE_MFMA")) dense_mfma = atoi(e);
    if (const char* e = getenv("QSIM_DENSE_FORM")) dense_form = atoi(e);
    if (const char* e = getenv("QSIM_DENSE_PF")) dense_pf = atoi(e);
    if (const char* e = getenv("QSIM_DENSE_GROUPS")) dense_groups = atoi(e);
    if (const char* e = getenv("QSIM_DENSE_WGS")) dense_wgs = atoi(e);
    if (const char* e = getenv("QSIM_DENSE_CONSEC")) dense_consec = atoi(e);
    if (const char* e = getenv("QSIM_DENSE_SKEW")) dense_skew = atoll(e);
    if (const char* e = getenv("QSIM_PLAN_ANCHOR")) plan_anchor = std::max(0, std::min(7, atoi(e)));
    if (const char* e = getenv("QSIM_TILE_MUX")) tile_mux = atoi(e);
    if (const char* e = getenv("QSIM_TILE_COMMUTE_FUSE")) tile_commute_fuse = atoi(e);
    if (const char* e = getenv("QSIM_TILE_LAST_SEARCH")) tile_last_search = atoi(e);
    // Probe knobs (cache policy, launch shapes, gate-less passes that give WRONG results)
    if (const char* e = getenv("QSIM_MALL_BYTES")) mall_bytes = strtoull(e, nullptr, 10);
    if (const char* e = getenv("QSIM_SWZ_CUT")) swz_cut = atoi(e);
    if (const char* e = getenv("QSIM_NT")) force_nt = atoi(e);
    if (const char* e = getenv("QSIM_ITEMS")) items = atoi(e);
    if (const char* e = getenv("QSIM_DEBUG_SKIP_GATES")) debug_skip_gates = atoi(e);
    if (const char* e = getenv("QSIM_TILE_ORDER")) tile_order = atoi(e);
#endif
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
  bool streaming;     // the non-temporal (Infinity-Cache bypassing) instantiation ran
  hipEvent_t e0, e1;
};
// One profile per STREAM (round 4: it was one per process, so a second host thread driving another handle could not
// open its own and interleaved its launches into the first one's -- the header promises that different handles may be
// driven from different host threads).  The table is guarded by a mutex; a launch costs one lookup while any profile is
// open and one relaxed load while none is.
struct ProfileState {
  std::vector<LaunchRecord> records;
};
static std::mutex g_prof_mu;
static std::map<hipStream_t, ProfileState> g_profs;       // open profiles, by stream
static std::vector<hipEvent_t> g_prof_pool;               // recycled events (guarded by g_prof_mu)
static std::atomic<int> g_prof_open{0};
static const char* const kClassNames[] = {
    "k_gate<1> scale (diagonal subset)", "k_gate<2> 2x2 butterfly", "k_gate<4> 4x4 butterfly",
    "k_gate_shuffle<1,1> lane 1q", "k_gate_shuffle<1,2> lane 2q", "k_gate_shuffle<2,1> lane+reg 2q",
    "k_tile fused pass", "k_dense dense k-qubit block"};
constexpr int kNumClasses = 8;

static hipEvent_t prof_event_locked() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;   // the launch goes untimed (ProfileScope drops it)
  return e;
}

struct ProfileScope {  // RAII around one launch
  bool on = false;
  LaunchRecord rec;
  ProfileScope(int cls, double bytes, hipStream_t stream, bool streaming, double hbm_bytes = -1.0) {
    if (g_prof_open.load(std::memory_order_relaxed) == 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (g_profs.find(stream) == g_profs.end()) return;
    rec.cls = cls;
    rec.bytes = bytes;
    rec.streaming = streaming;
    rec.hbm_bytes = hbm_bytes < 0 ? bytes : hbm_bytes;
    rec.e0 = prof_event_locked();
    rec.e1 = prof_event_locked();
    if (!rec.e0 || !rec.e1) {               // no event available: keep what we got for reuse, time nothing
      if (rec.e0) g_prof_pool.push_back(rec.e0);
      if (rec.e1) g_prof_pool.push_back(rec.e1);
      return;
    }
    on = true;
    (void)hipEventRecord(rec.e0, stream);
  }
  void done(hipStream_t stream) {
    if (!on) return;
    (void)hipEventRecord(rec.e1, stream);
    std::lock_guard<std::mutex> lock(g_prof_mu);
    auto it = g_profs.find(stream);
    if (it != g_profs.end()) it->second.records.push_back(rec);
    else { g_prof_pool.push_back(rec.e0); g_prof_pool.push_back(rec.e1); }   // (the profile was closed meanwhile)
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
  ProfileScope prof(NM == 1 ? 0 : (NM == 2 ? 1 : 2), 32.0 * NM * (double)p.count, stream, NT);
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
  ProfileScope prof(NMR == 2 ? 5 : (NSH == 1 ? 3 : 4), 32.0 * NMR * (double)p.count, stream, NT);
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
