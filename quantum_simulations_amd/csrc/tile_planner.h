// tile_planner.h -- host planner of the fused passes: op list -> passes -> register groups.
// Part of the single translation unit qsim_hip.hip (included there, in order; not a standalone header).
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
  double absorbed;     // algorithmic bytes of the gates fused into this one, as a fraction of 32 B x 2^k
  bool control_zero;   // the control must be 0 instead of 1 (only produced inside emit_groups, for a control outside the tile)
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
  o->absorbed = 0.0;
  o->control_zero = false;
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

// Tile order of a launch (k_tile).  Gate-less passes depend strongly on it (profiles/r02f_tile_order_probe.txt:
// 2.38 -> 1.58 ms for tile bits {3,4,5,6,18,19,20,21} with bit-reversed order, 1.90 -> 1.70 ms averaged over ten
// tile-bit sets with a bits-above-18 rule), but passes that carry their gates do not: per pass the best of the
// three orders is within 2 % of consecutive tiles on the bench circuit (profiles/r02i_pass_times_by_order.txt),
// so the product launches consecutive tiles; the other orders stay reachable in the probe build.
static int tile_order_for(const TileArgs&, int) {
#ifdef QSIM_PROBES
  if (tuning().tile_order >= 0) return tuning().tile_order;
#endif
  return 0;
}

// physical bit of logical index bit b under a slab layout (host mirror of k_tile's slab_base)
static uint8_t slab_position(const TileSlab& t, int b) {
  for (int j = 0; j < 3; ++j) if (t.bit[j] == b) return (uint8_t)(t.top + j);
  for (int i = 0; i < 4; ++i) if ((t.field[i] >> b) & 1) return (uint8_t)(b - i);
  return (uint8_t)b;
}

template <int T>
static int launch_tile(const TileArgs& a, const qsim_chunk* c, hipStream_t stream, double alg_bytes) {
  if constexpr (T > kTileBitsMax || T < kTileThreadBits) {
    return fail(QSIM_ERR_INVALID, "internal: tile size %d not built", T);
  } else {
  if (a.nfix && (T != kTileBitsMax || a.nfix > 3 || c->k - T < a.nfix)) return fail(QSIM_ERR_INVALID, "internal: partial launch of a %d-bit tile", T);
  const u64 ntiles = 1ull << (c->k - T - a.nfix);       // (a.nfix: the tiles of one destination slab of a fused re-layout)
  if (ntiles > 0xFFFFFFFFull) return fail(QSIM_ERR_INVALID, "internal: too many tiles");
  TileArgs args = a;
  args.ntiles = (uint32_t)ntiles;
  args.order = (uint8_t)((a.order & ~kTileOrderMask) | (tile_order_for(a, T) & kTileOrderMask));
  args.nbits = (uint8_t)c->k;
  if (!args.amp_out) args.amp_out = args.amp;
  // a pass that reads / writes a re-laid-out buffer: the thread layouts become physical bit positions
  for (int i = 0; i < T - kTileLow; ++i) {
    if (args.perm & kTilePermIn) args.lay_in[i] = slab_position(args.slab_in, args.lay_in[i]);
    if (args.perm & kTilePermOut) args.lay_out[i] = slab_position(args.slab_out, args.lay_out[i]);
  }
#ifdef QSIM_PROBES
  if (getenv("QSIM_DEBUG_OUT_OF_PLACE")) {   // memory probe (WRONG results): tiles are stored into a second buffer
    static double2* other = nullptr;
    static size_t other_bytes = 0;
    const size_t need = sizeof(double2) << c->k;
    if (other_bytes < need) { if (other) (void)hipFree(other); (void)hipMalloc((void**)&other, need); other_bytes = need; }
    args.amp_out = other;
  }
  {   // in-kernel stamps: QSIM_DEBUG_STAMPS=<file> dumps entry / loaded / engine / stored cycle stamps of every 64th workgroup
    static unsigned long long* dbuf = nullptr;
    static const char* path = getenv("QSIM_DEBUG_STAMPS");
    if (path && !dbuf) (void)hipMalloc((void**)&dbuf, sizeof(unsigned long long) * 4 * ((1u << 22)));
    unsigned long long* p = path ? dbuf : nullptr;
    std::memcpy(reinterpret_cast<unsigned char*>(&args) + kTileArgBytes - 8, &p, 8);
    if (p) (void)hipMemsetAsync(p, 0, sizeof(unsigned long long) * 4 * (ntiles / 64 + 1), stream);
  }
#endif
#ifndef QSIM_TILES_PER_WG
#define QSIM_TILES_PER_WG 2
#endif
  // thread part of an element offset: the first min(5, NH) bits of the load / store layouts: 32-bit addressing when
  // all are < 28
  bool wide = false;
  for (int i = 0; i < kTileThreadBits - kTileLow && i < T - kTileLow; ++i) wide = wide || args.lay_in[i] >= 28 || args.lay_out[i] >= 28;
  bool nt = c->span_bytes > tuning().mall_bytes;       // cache policy by state size (gate_plan.h)
  if (tuning().force_nt >= 0) nt = tuning().force_nt != 0;
  ProfileScope prof(6, alg_bytes, stream, nt, 32.0 * (double)(ntiles << T));
  // Two tiles per workgroup, the second one's loads in flight while the first is computed on
  // (profiles/r02r_ab_prefetch_before_engine.txt: -3.9 % per pass; 4 or 8 tiles per workgroup lose it again,
  // profiles/r02z_ab_paired_stores.txt).  Also for the 64-bit-offset form (110 VGPRs: the 32 KiB of LDS admit four
  // workgroups per CU, so 128 are there); not for small grids.
  constexpr int TPW = QSIM_TILES_PER_WG;
  if constexpr (T == kTileBitsMax) {
    if (args.nfix) {                                       // one destination slab: the PART instantiations (full tiles only)
      const bool two = TPW > 1 && ntiles >= (u64)TPW * 4096;
      const unsigned grid = (unsigned)(two ? ntiles / TPW : ntiles);
      if (two) {
        if (nt && wide) hipLaunchKernelGGL((k_tile<T, true, true, TPW, true>), dim3(grid), dim3(kTileThreads), 0, stream, args);
        else if (nt) hipLaunchKernelGGL((k_tile<T, true, false, TPW, true>), dim3(grid), dim3(kTileThreads), 0, stream, args);
        else if (wide) hipLaunchKernelGGL((k_tile<T, false, true, TPW, true>), dim3(grid), dim3(kTileThreads), 0, stream, args);
        else hipLaunchKernelGGL((k_tile<T, false, false, TPW, true>), dim3(grid), dim3(kTileThreads), 0, stream, args);
      } else {
        if (nt && wide) hipLaunchKernelGGL((k_tile<T, true, true, 1, true>), dim3(grid), dim3(kTileThreads), 0, stream, args);
        else if (nt) hipLaunchKernelGGL((k_tile<T, true, false, 1, true>), dim3(grid), dim3(kTileThreads), 0, stream, args);
        else if (wide) hipLaunchKernelGGL((k_tile<T, false, true, 1, true>), dim3(grid), dim3(kTileThreads), 0, stream, args);
        else hipLaunchKernelGGL((k_tile<T, false, false, 1, true>), dim3(grid), dim3(kTileThreads), 0, stream, args);
      }
      prof.done(stream);
      HIP_TRY(hipGetLastError());
      return QSIM_OK;
    }
  }
  if (TPW > 1 && ntiles >= (u64)TPW * 4096) {
    const unsigned grid = (unsigned)(ntiles / TPW);
    if (nt && wide) hipLaunchKernelGGL((k_tile<T, true, true, TPW>), dim3(grid), dim3(kTileThreads), 0, stream, args);
    else if (nt) hipLaunchKernelGGL((k_tile<T, true, false, TPW>), dim3(grid), dim3(kTileThreads), 0, stream, args);
    else if (wide) hipLaunchKernelGGL((k_tile<T, false, true, TPW>), dim3(grid), dim3(kTileThreads), 0, stream, args);
    else hipLaunchKernelGGL((k_tile<T, false, false, TPW>), dim3(grid), dim3(kTileThreads), 0, stream, args);
  } else {
    const unsigned grid = (unsigned)ntiles;
    if (nt && wide) hipLaunchKernelGGL((k_tile<T, true, true, 1>), dim3(grid), dim3(kTileThreads), 0, stream, args);
    else if (nt) hipLaunchKernelGGL((k_tile<T, true, false, 1>), dim3(grid), dim3(kTileThreads), 0, stream, args);
    else if (wide) hipLaunchKernelGGL((k_tile<T, false, true, 1>), dim3(grid), dim3(kTileThreads), 0, stream, args);
    else hipLaunchKernelGGL((k_tile<T, false, false, 1>), dim3(grid), dim3(kTileThreads), 0, stream, args);
  }
  prof.done(stream);
  HIP_TRY(hipGetLastError());
#ifdef QSIM_PROBES
  if (const char* path = getenv("QSIM_DEBUG_STAMPS")) {
    unsigned long long* p = nullptr;
    std::memcpy(&p, reinterpret_cast<unsigned char*>(&args) + kTileArgBytes - 8, 8);
    std::vector<unsigned long long> host(4 * (ntiles / 64 + 1));
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(host.data(), p, host.size() * 8, hipMemcpyDeviceToHost);
    if (FILE* f = std::fopen(path, "a")) {
      std::fprintf(f, "# pass nrec %d ntiles %llu\n", a.nrec, (u64)ntiles);
      for (size_t i = 0; i + 3 < host.size(); i += 4)
        if (host[i]) std::fprintf(f, "%llu %llu %llu %llu\n", host[i], host[i + 1], host[i + 2], host[i + 3]);
      std::fclose(f);
    }
  }
#endif
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

static inline u64 op_qmask(const FusedOp& o) {
  u64 m = 1ull << o.qubits[0];
  if (o.nq == 2) m |= 1ull << o.qubits[1];
  return m;
}

// ---- commutation-aware fusion of one-qubit gates -------------------------------------------------
// The host planners (circuit/fusion.py, the reference's fuse_1q_ops) only merge 1q gates that are ADJACENT on their
// qubit.  Inside the library a 1q gate G on qubit q also moves forward past every op it commutes with -- a
// controlled gate whose control is q when G is diagonal (Z, S, T, R), a controlled gate whose target is q when G
// commutes with its 2x2 (X through CNOT targets), a CZ / CR on q when G is diagonal -- and is multiplied into the
// next 1q gate on q.  Same unitary (rounding differs at 1e-16); on the bench circuit 1 in 5 1q gates disappears
// this way, among them every X (the costliest 1q record of the engine: 16 v_swap_b32 at half rate).
static inline double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
static inline double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
static bool op_1q_matrix(const FusedOp& o, double2 g[4]) {   // uncontrolled 1q op -> its 2x2
  if (o.kind == TG_PHASE && o.nbits == 1) {
    g[0] = make_double2(1, 0); g[1] = g[2] = make_double2(0, 0); g[3] = o.m[0];
    return true;
  }
  if ((o.kind == TG_DENSE1 || o.kind == TG_ANTI1 || o.kind == TG_SWAP1) && o.control < 0) {
    for (int i = 0; i < 4; ++i) g[i] = o.m[i];
    return true;
  }
  return false;
}
static int op_cost(const FusedOp& o);          // vector instructions per thread of the uncontrolled record (below)
constexpr int kRecordCost = 8;                 // a record's fetch + dispatch, in the same unit
static void commute_fuse_1q(std::vector<FusedOp>* ops, bool backward) {
  // forward: gate i moves later, into the next 1q gate j on its qubit (the product sits at j); backward: that gate j
  // moves earlier, into i (the product sits at i) -- legal under the same condition, everything between them on the
  // qubit commuting with the gate that moves... which for the backward form is gate j: its matrix is the one tested.
  const size_t n = ops->size();
  std::vector<char> dead(n, 0);
  constexpr size_t kWindow = 512;              // ops looked at behind a gate (bounds the cost on long lists)
  auto zero = [](double2 v) { return v.x == 0 && v.y == 0; };
  auto commutes_with = [&](const double2 g[4], const FusedOp& o, int q) {
    const bool diag = zero(g[1]) && zero(g[2]);
    if (o.kind == TG_PHASE) return diag;                                   // CZ / CR
    if (o.kind == TG_DENSE2) return false;
    if (o.control == q) return diag;                                       // q controls it
    if (o.control >= 0 && o.target[0] == q) {                              // q is its target: g v == v g ?
      const double2* v = o.m;
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 2; ++c) {
          const double2 gv = cadd(cmul(g[2 * r], v[c]), cmul(g[2 * r + 1], v[2 + c]));
          const double2 vg = cadd(cmul(v[2 * r], g[c]), cmul(v[2 * r + 1], g[2 + c]));
          if (gv.x != vg.x || gv.y != vg.y) return false;
        }
      return true;
    }
    return false;
  };
  for (size_t i = 0; i < n; ++i) {
    double2 g[4];
    if (dead[i] || !op_1q_matrix((*ops)[i], g)) continue;
    const int q = (*ops)[i].qubits[0];
    // the next 1q gate j on q, and whether everything on q in between commutes with the gate that moves
    size_t j = i + 1;
    bool g_passes = true, found = false;
    std::vector<size_t> between;
    for (; j < n && j <= i + kWindow; ++j) {
      const FusedOp& o = (*ops)[j];
      if (dead[j] || !((op_qmask(o) >> q) & 1)) continue;
      double2 h[4];
      if (op_1q_matrix(o, h)) { found = true; break; }
      between.push_back(j);
      if (!backward && !commutes_with(g, o, q)) { g_passes = false; break; }
    }
    if (!found || !g_passes) continue;
    double2 h[4];
    op_1q_matrix((*ops)[j], h);
    if (backward) {
      bool ok = true;
      for (size_t b : between) ok = ok && commutes_with(h, (*ops)[b], q);
      if (!ok) continue;
    }
    const double2 f[4] = {cadd(cmul(h[0], g[0]), cmul(h[1], g[2])), cadd(cmul(h[0], g[1]), cmul(h[1], g[3])),
                          cadd(cmul(h[2], g[0]), cmul(h[3], g[2])), cadd(cmul(h[2], g[1]), cmul(h[3], g[3]))};
    const double U[8] = {f[0].x, f[0].y, f[1].x, f[1].y, f[2].x, f[2].y, f[3].x, f[3].y};
    const int32_t qq[2] = {q, -1};
    auto frac = [](const FusedOp& x) { return x.absorbed + 1.0 / (double)(1ull << (x.kind == TG_PHASE ? x.nbits : x.halvings)); };
    const size_t keep = backward ? i : j, drop = backward ? j : i;
    const double moved = (*ops)[keep].absorbed + frac((*ops)[drop]);
    FusedOp fused;
    const bool identity = !classify_op(1, qq, U, &fused);
    // Worth it?  The engine's special cases are cheap (Z 8, H 16, T 16, S 20 vector instructions per thread against 32
    // for a real 2x2, 40 anti-diagonal, 68 complex): a product that costs more than its factors plus one record's
    // dispatch is left alone (measured: fusing everything that commutes made the pass 2 % slower at 15 % fewer records).
    if (!identity && op_cost(fused) > op_cost((*ops)[i]) + op_cost((*ops)[j]) + kRecordCost) continue;
    if (!identity) { fused.absorbed = moved; (*ops)[keep] = fused; }
    else dead[keep] = 1;                        // the product is the identity
    dead[drop] = 1;
    if (backward && !dead[i]) --i;              // the product may take the next gate in as well
  }
  size_t w = 0;
  for (size_t i = 0; i < n; ++i) if (!dead[i]) (*ops)[w++] = (*ops)[i];
  ops->resize(w);
}

constexpr int kTileMinChunk = kTileThreadBits > 8 ? kTileThreadBits : 8;   // smaller chunks run gate by gate (a tile holds at least one amplitude per thread)


// The case an op gets once its register positions are known depends only on its kind and matrix:
// family entry and matrix doubles of the record (tile_kernel.h, record stream).
struct OpShape { int family; int nd; };
static OpShape op_shape(const FusedOp& o) {
  const bool sp = tuning().tile_special;
  auto is = [&](int e, double re, double im) { return o.m[e].x == re && o.m[e].y == im; };
  switch (o.kind) {
    case TG_SWAP1: return {OPC_SWAP1, 0};
    case TG_PHASE:
      if (sp && is(0, -1, 0)) return {OPC_PHASE_NEG, 0};
      if (sp && is(0, 0, 1)) return {OPC_PHASE_I, 0};
      if (sp && is(0, 0, -1)) return {OPC_PHASE_NI, 0};
      return {OPC_PHASE, 2};
    case TG_ANTI1: return (sp && is(1, 0, -1) && is(2, 0, 1)) ? OpShape{OPC_YLIKE1, 0} : OpShape{OPC_ANTI1, 4};
    case TG_DENSE2: return {OPC_DENSE2, 32};
    default:   // a real 2x2 (H, RY, G) needs four doubles instead of eight; an uncontrolled c [[1,1],[1,-1]] none
      if (sp && tuning().tile_had && o.control < 0 && o.m[0].y == 0 && o.m[1].y == 0 && o.m[2].y == 0 && o.m[3].y == 0 &&
          o.m[0].x == o.m[1].x && o.m[0].x == o.m[2].x && o.m[0].x == -o.m[3].x && o.m[0].x != 0)
        return {OPC_HAD1, 0};
      return (sp && o.m[0].y == 0 && o.m[1].y == 0 && o.m[2].y == 0 && o.m[3].y == 0) ? OpShape{OPC_REAL1, 4}
                                                                                        : OpShape{OPC_DENSE1, 8};
  }
}

static int op_cost(const FusedOp& o) {
  switch (op_shape(o).family) {
    case OPC_PHASE_NEG: return 8;
    case OPC_PHASE: case OPC_HAD1: return 16;
    case OPC_PHASE_I: case OPC_PHASE_NI: return 20;
    case OPC_SWAP1: case OPC_REAL1: return 32;
    case OPC_ANTI1: case OPC_YLIKE1: return 40;
    case OPC_DENSE1: return 68;
    default: return 144;
  }
}

// Split one pass's ops (list order) into register groups of <= kGroupBits target tile bits and
// collect their descriptors.  Ops that do not fit the record budget stay un-emitted (they and
// everything that depends on them wait for the next launch).
static void emit_groups(const std::vector<FusedOp>& ops, const std::vector<size_t>& members,
                        const std::vector<int>& high, int T, std::vector<TileGroup>* out, std::vector<char>* emitted,
                        bool last_search = false, int direct_worth = 2) {
  double pass_scale = 1.0;            // product of the factors of the pass's unscaled Hadamard butterflies (OPC_HAD1)
  const int low = kTileLow;
  auto tile_pos = [&](int b) -> int {
    if (b < low) return b;
    for (size_t j = 0; j < high.size(); ++j) if (high[j] == b) return low + (int)j;
    return -1;
  };
  std::vector<char> done(members.size(), 0);
  size_t left = members.size();
  out->clear();
  int used = 0;                       // bytes of the records written so far
  const bool merge_on = tuning().tile_merge_diag != 0;
  auto run_bytes = [](unsigned touched) { const int n = __builtin_popcount(touched); return desc_bytes(n == 1 ? 2 : (n == 2 ? 6 : 14)); };
  const unsigned line_bits = (1u << low) - 1;
  const bool direct_ends = tuning().tile_direct && T == kTileBitsMax;   // (serialize_pass: full tiles only)
  // The LAST group of a full tile is free of its LDS write-back when it lies above the line bits (the kernel stores
  // the tile in that layout, OPC_END_DIRECT).  Chosen first, from the END of the list: the triple of bits above the
  // line bits that holds the most TERMINAL ops (ops that no op outside the set follows on any of their qubits);
  // those ops are set aside, the groups in front of them are built as before, and they are written last.
  bool cut = false;
  // Write one register group: the ops `grp` (indices into members, list order) on the tile bits `claimed`.
  // Returns true when the record budget ended inside the group.
  auto emit_group = [&](const std::vector<size_t>& grp, unsigned claimed) -> bool {
    std::vector<int> S;               // tile bits of this group
    for (unsigned m = claimed; m; m &= m - 1) S.push_back(__builtin_ctz(m));
    // pad the group with the highest unused tile bits (high bits keep LDS accesses contiguous)
    for (int b = T - 1; (int)S.size() < kGroupBits && b >= 0; --b)
      if (std::find(S.begin(), S.end(), b) == S.end()) S.push_back(b);
    std::sort(S.begin(), S.end());
    auto reg_pos = [&](int tile_bit) -> int {
      for (int j = 0; j < kGroupBits; ++j) if (S[j] == tile_bit) return j;
      return -1;
    };
    TileGroup tg;
    for (int j = 0; j < 3; ++j) tg.s[j] = S[j];
    tg.qmask = 0;
    used += kGroupRecordBytes;
    auto emit = [&](TileDesc d) {
      used += desc_bytes(d);
      tg.gates.push_back(d);
    };
    // Phase gates with ONE register bit and the same predicate (lane bits + outer bits) are merged
    // (the QFT's CR(k, a), CR(k, b), CR(k, c) for the group's register bits a, b, c): diagonal
    // gates commute with everything except a non-diagonal gate on one of their bits, so an open
    // accumulator is written out before such a gate on a register bit it has touched, or at the
    // end of the group.  One record instead of up to three: the gate loop is instruction-issue bound.
    struct Acc { uint16_t blk; u64 outer; double2 phi[3]; unsigned touched; };
    std::vector<Acc> open;
    auto cmul2 = [](double2 f, double2 m) { return make_double2(f.x * m.x - f.y * m.y, f.x * m.y + f.y * m.x); };
    auto flush = [&](size_t i) {
      const Acc acc = open[i];
      open.erase(open.begin() + (long)i);
      TileDesc d;
      std::memset(&d, 0, sizeof d);
      d.blk_mask = acc.blk;
      d.outer_mask = acc.outer;
      double2 m[3];
      int nm = 0, regs[3];
      for (int r = 0; r < 3; ++r) if (acc.touched & (1u << r)) { regs[nm] = r; m[nm++] = acc.phi[r]; }
      auto put = [&](int at, double2 v) { d.m[2 * at] = v.x; d.m[2 * at + 1] = v.y; };
      // one phase on one register bit: -1 / i / -i have their own families (2 / 3 / 3 vector instructions per
      // register instead of 4)
      auto single = [&](int r, double2 v) {
        TileDesc s1 = d;
        const bool sp = tuning().tile_special;
        const int fam = (sp && v.x == -1 && v.y == 0) ? OPC_PHASE_NEG : (sp && v.x == 0 && v.y == 1) ? OPC_PHASE_I
                        : (sp && v.x == 0 && v.y == -1) ? OPC_PHASE_NI : OPC_PHASE;
        s1.opcode = (uint8_t)(fam + (1u << r));
        if (fam == OPC_PHASE) { s1.m[0] = v.x; s1.m[1] = v.y; s1.nd = 2; }
        emit(s1);
      };
      auto cost1 = [&](double2 v) {   // vector instructions of the single form (4 registers)
        if (!tuning().tile_special) return 16;
        return (v.x == -1 && v.y == 0) ? 8 : ((v.x == 0 && (v.y == 1 || v.y == -1)) ? 12 : 16);
      };
      if (nm == 1) { single(regs[0], m[0]); return; }
      // merged run: 6 (two bits) or 7 (three bits) registers x 4 instructions, one record instead of nm
      int separate = 0;
      for (int e = 0; e < nm; ++e) separate += cost1(m[e]);
      if (separate < (nm == 2 ? 24 : 28)) {
        for (int e = 0; e < nm; ++e) single(regs[e], m[e]);
        return;
      }
      if (nm == 2) {
        d.opcode = (uint8_t)(OPC_DIAGR + (acc.touched == 3 ? 0 : acc.touched == 5 ? 1 : 2));
        put(0, m[0]); put(1, m[1]); put(2, cmul2(m[0], m[1])); d.nd = 6;
      } else {                          // a, b, c | ab, ac, bc, abc
        d.opcode = (uint8_t)(OPC_DIAGR + 3);
        const double2 ab = cmul2(m[0], m[1]);
        put(0, m[0]); put(1, m[1]); put(2, m[2]);
        put(3, ab); put(4, cmul2(m[0], m[2])); put(5, cmul2(m[1], m[2])); put(6, cmul2(ab, m[2]));
        d.nd = 14;
      }
      emit(d);
    };
    // The group's ops in emission order.  Peephole (tuning().tile_mux): a controlled gate C(V) whose control lies
    // OUTSIDE the tile (a per-tile predicate) next to an unconditional 1q gate U on its target -- nothing between
    // them touching the target -- becomes two predicated records at U's place: control = 1 -> U V (or V U when U comes
    // first), control = 0 -> U.  A tile runs exactly one of the two, so the pair costs one 2x2 instead of 2x2 + V; for
    // V = X (CNOT, 3 of 4 cases on the bench circuit) that removes 16 half-rate v_swap_b32 per thread.
    struct Item { FusedOp op; size_t mi; int with_next; };   // with_next: record bytes of the second half of a pair (both or neither fit)
    std::vector<Item> seq;
    seq.reserve(grp.size() + 4);
    {
      const size_t ng = grp.size();
      std::vector<char> gone(ng, 0), paired(ng, 0);
      std::vector<FusedOp> first_half(ng);     // for a paired U: the control = 1 record emitted in front of it
      auto is_plain_1q = [&](const FusedOp& u) {
        return (u.kind == TG_DENSE1 || u.kind == TG_ANTI1) && u.control < 0;
      };
      auto mul2 = [&](const double2* a, const double2* b, double2* out) {   // out = a b
        for (int r = 0; r < 2; ++r)
          for (int c = 0; c < 2; ++c) out[2 * r + c] = cadd(cmul(a[2 * r], b[c]), cmul(a[2 * r + 1], b[2 + c]));
      };
      if (tuning().tile_mux)
        for (size_t p = 0; p < ng; ++p) {
          const FusedOp& cv = ops[members[grp[p]]];
          if (gone[p] || paired[p] || cv.control < 0 || tile_pos(cv.control) >= 0) continue;
          if (cv.kind != TG_SWAP1 && cv.kind != TG_ANTI1 && cv.kind != TG_DENSE1) continue;
          const int t = cv.target[0];
          const u64 tbit = 1ull << t;
          long partner = -1;
          bool u_first = false;
          for (size_t q = p + 1; q < ng; ++q) {              // U after C(V)
            if (gone[q]) continue;
            const FusedOp& u = ops[members[grp[q]]];
            if (!(op_qmask(u) & tbit)) continue;
            if (is_plain_1q(u) && !paired[q]) partner = (long)q;
            break;
          }
          if (partner < 0) {                                 // U in front of C(V) -- unless C(V) can sink into the write-back
            bool touched_later = false;
            for (size_t q = p + 1; q < ng && !touched_later; ++q) touched_later = !gone[q] && (op_qmask(ops[members[grp[q]]]) & tbit);
            if (cv.kind == TG_SWAP1 && tuning().tile_sink_swaps && !touched_later) continue;
            for (size_t q = p; q-- > 0;) {
              if (gone[q]) continue;
              const FusedOp& u = ops[members[grp[q]]];
              if (!(op_qmask(u) & tbit)) continue;
              if (is_plain_1q(u) && !paired[q]) { partner = (long)q; u_first = true; }
              break;
            }
          }
          if (partner < 0) continue;
          const FusedOp& u = ops[members[grp[(size_t)partner]]];
          double2 m1[4];
          if (u_first) mul2(cv.m, u.m, m1); else mul2(u.m, cv.m, m1);
          FusedOp a = cv;                                    // control = 1 half: keeps C(V)'s control and bookkeeping
          for (int e = 0; e < 4; ++e) a.m[e] = m1[e];
          set_1q_kind(&a);
          first_half[(size_t)partner] = a;
          paired[(size_t)partner] = 1;
          gone[p] = 1;
          first_half[(size_t)partner].nq = (int)p;           // (slot reused below: which member the first half stands for)
        }
      for (size_t q = 0; q < ng; ++q) {
        if (gone[q]) continue;
        const FusedOp& u = ops[members[grp[q]]];
        if (!paired[q]) { seq.push_back(Item{u, grp[q], 0}); continue; }
        FusedOp a = first_half[q];
        const size_t p = (size_t)a.nq;
        a.nq = 2;
        FusedOp b = u;                                       // control = 0 half: U under the complementary predicate
        b.control = a.control;
        b.control_zero = true;
        b.nq = 2;
        b.qubits[1] = a.control;
        seq.push_back(Item{a, grp[p], desc_bytes(op_shape(b).nd)});
        seq.push_back(Item{b, grp[q], 0});
      }
    }
    // qubits touched by the ops AFTER position i of the group (X / CNOT that nothing later touches are sunk
    // into the write-back: OPC_ASWAP1)
    std::vector<u64> later(seq.size() + 1, 0);
    for (size_t i = seq.size(); i-- > 0;) later[i] = later[i + 1] | op_qmask(seq[i].op);
    size_t gi = 0;
    for (const Item& item : seq) {
      const size_t mi = item.mi;
      const u64 touched_later = later[++gi];
      const FusedOp& o = item.op;
      TileDesc d;
      std::memset(&d, 0, sizeof d);
      unsigned reg_mask = 0;
      int ctrl_reg = -1;
      auto require_one = [&](int qubit) {      // a control / phase bit
        const int p = tile_pos(qubit);
        if (p < 0) { d.outer_mask |= 1ull << qubit; return; }
        const int r = reg_pos(p);
        if (r >= 0) { reg_mask |= 1u << r; ctrl_reg = r; }
        else d.blk_mask |= (uint16_t)(1u << p);
      };
      const OpShape shape = op_shape(o);
      {   // exact budget: records written so far + what the open runs may need + this op (a mergeable
          // phase may grow a run to its largest form)
        int reserve = 0;
        for (const Acc& acc : open) reserve += run_bytes(acc.touched);
        const bool mergeable = merge_on && o.kind == TG_PHASE;
        if (used + reserve + (mergeable ? desc_bytes(14) : desc_bytes(shape.nd)) + item.with_next > kTileRecordBudget) { cut = true; break; }
      }
      done[mi] = 1;
      (*emitted)[mi] = 1;
      --left;
      tg.qmask |= op_qmask(o);
      auto put = [&](int at, double2 v) { d.m[2 * at] = v.x; d.m[2 * at + 1] = v.y; };
      if (o.kind == TG_PHASE) {
        for (int t = 0; t < o.nbits; ++t) require_one(o.bits[t]);
        d.opcode = (uint8_t)(shape.family + reg_mask);
        if (merge_on && __builtin_popcount(reg_mask) == 1) {   // (every phase family: -1 / +-i join the runs too)
          const int r = __builtin_ctz(reg_mask);
          size_t i = 0;
          while (i < open.size() && !(open[i].blk == d.blk_mask && open[i].outer == d.outer_mask)) ++i;
          if (i == open.size()) {
            Acc acc;
            acc.blk = d.blk_mask; acc.outer = d.outer_mask; acc.touched = 0;
            for (int e = 0; e < 3; ++e) acc.phi[e] = make_double2(1.0, 0.0);
            open.push_back(acc);
          }
          Acc& acc = open[i];
          acc.phi[r] = cmul2(acc.phi[r], o.m[0]);
          acc.touched |= 1u << r;
          continue;
        }
        if (shape.nd) { put(0, o.m[0]); d.nd = 2; }
      } else if (o.kind == TG_DENSE2) {
        d.opcode = (uint8_t)(OPC_DENSE2 + 3 * reg_pos(tile_pos(o.target[0])) + reg_pos(tile_pos(o.target[1])));
        for (int e = 0; e < 16; ++e) put(e, o.m[e]);
        d.nd = 32;
      } else {
        const int J = reg_pos(tile_pos(o.target[0]));
        if (o.control >= 0 && o.control_zero) { d.outer_mask |= 1ull << o.control; d.outer_zero = true; }   // (outside the tile by construction)
        else if (o.control >= 0) require_one(o.control);
        d.opcode = (uint8_t)(shape.family + opc_1q_variant(J, ctrl_reg));
        if (shape.family == OPC_SWAP1 && tuning().tile_sink_swaps &&
            !(touched_later & (1ull << o.target[0])) && !(ctrl_reg >= 0 && (touched_later & (1ull << o.control))))
          d.opcode = (uint8_t)(OPC_ASWAP1 + opc_1q_variant(J, ctrl_reg));
        if (shape.family == OPC_HAD1) {
          pass_scale *= o.m[0].x;
        } else if (shape.family == OPC_REAL1) {
          d.m[0] = o.m[0].x; d.m[1] = o.m[1].x; d.m[2] = o.m[2].x; d.m[3] = o.m[3].x; d.nd = 4;
        } else if (shape.family == OPC_ANTI1) {
          put(0, o.m[1]); put(1, o.m[2]); d.nd = 4;
        } else if (shape.family == OPC_DENSE1) {
          for (int e = 0; e < 4; ++e) put(e, o.m[e]);
          d.nd = 8;
        }
      }
      if (o.kind != TG_PHASE) {               // a non-diagonal gate: its targets end the open phase runs on them
        unsigned tmask = 0;
        for (int t = 0; t < o.ntargets; ++t) tmask |= 1u << reg_pos(tile_pos(o.target[t]));
        for (size_t i = open.size(); i-- > 0;) if (open[i].touched & tmask) flush(i);
      }
      emit(d);
    }
    while (!open.empty()) flush(0);
    if (!tg.gates.empty()) out->push_back(tg);
    else used -= kGroupRecordBytes;
    return cut;
  };
  std::vector<char> reserved(members.size(), 0);
  unsigned last_own = 0;
  int reserve_bytes = 0;
  size_t n_reserved = 0;
  if (direct_ends && last_search && members.size() > 1) {
    struct Tail { u64 qm; unsigned need; int bytes; };
    std::vector<Tail> tail(members.size());
    unsigned cand = 0;
    for (size_t mi = 0; mi < members.size(); ++mi) {
      const FusedOp& o = ops[members[mi]];
      unsigned need = 0;
      for (int t = 0; t < o.ntargets; ++t) need |= 1u << tile_pos(o.target[t]);
      tail[mi] = Tail{op_qmask(o), need, o.kind == TG_PHASE ? 48 : desc_bytes(op_shape(o).nd)};
      cand |= need & ~line_bits;
    }
    auto terminal = [&](unsigned own, std::vector<char>* mark) -> int {
      u64 blocked = 0;
      int count = 0, bytes = kGroupRecordBytes;
      for (size_t mi = members.size(); mi-- > 0;) {
        const Tail& t = tail[mi];
        if ((blocked & t.qm) || (t.need & ~own) || bytes + t.bytes > kTileRecordBudget / 2) { blocked |= t.qm; continue; }
        bytes += t.bytes;
        ++count;
        if (mark) (*mark)[mi] = 1;
      }
      if (mark) reserve_bytes = bytes;
      return count;
    };
    int best = 0;
    while (__builtin_popcount(cand) < kGroupBits)          // fewer than three target bits above the line bits: pad
      for (int b = T - 1; b >= low; --b) if (!(cand & (1u << b))) { cand |= 1u << b; break; }
    for (unsigned a = cand; a; a &= a - 1)
      for (unsigned b = a & (a - 1); b; b &= b - 1)
        for (unsigned c = b & (b - 1); c; c &= c - 1) {
          const unsigned own = (a & -a) | (b & -b) | (c & -c);
          const int count = terminal(own, nullptr);
          if (count > best) { best = count; last_own = own; }
        }
    if (best > 0 && (size_t)best < members.size()) {
      n_reserved = (size_t)terminal(last_own, &reserved);
      used = reserve_bytes;
    } else {
      last_own = 0;
    }
  }
  while (left > n_reserved) {
    // Which three tile bits does the group own?  First come (an op that still fits claims the bits it needs)
    // was the only rule up to r02a; now every triple of the pending ops' target bits is also tried and the one
    // that lets the group hold the most ops wins (ties: first come).  A group change is an LDS round trip of the
    // tile plus a barrier (~4 % of a tile's time each): 102 -> 86 groups on the 18 passes of the bench circuit.
    // Estimate of the record budget: a phase gate that may be merged with others (OPC_DIAGR) is counted as a
    // bare header; the exact budget is enforced when the group is written out (a group that overflows is cut
    // there, the rest waits for the next pass).
    struct Pending { size_t mi; u64 qm; int pos[2]; int npos; int bytes; };
    std::vector<Pending> pend;
    pend.reserve(left);
    unsigned cand_mask = 0;           // tile positions that pending ops target
    for (size_t mi = 0; mi < members.size(); ++mi) {
      if (done[mi] || reserved[mi]) continue;
      const FusedOp& o = ops[members[mi]];
      const OpShape shape = op_shape(o);
      Pending pd;
      pd.mi = mi;
      pd.qm = op_qmask(o);
      pd.npos = o.ntargets;
      for (int t = 0; t < o.ntargets; ++t) { pd.pos[t] = tile_pos(o.target[t]); cand_mask |= 1u << pd.pos[t]; }
      pd.bytes = (merge_on && shape.family == OPC_PHASE) ? 16 : desc_bytes(shape.nd);
      pend.push_back(pd);
    }
    // ops a group owning the tile bits `own` (mask) would hold, in list order, within the record budget;
    // own == 0: first come (bits are claimed as ops need them)
    auto select = [&](unsigned own, std::vector<size_t>* grp_out, unsigned* claimed_out) -> int {
      const bool first_come = own == 0;
      u64 blocked = 0;
      int est = used + kGroupRecordBytes, count = 0;
      unsigned claimed = own;
      for (const Pending& pd : pend) {
        bool ok = !(blocked & pd.qm);
        unsigned need = 0;
        for (int t = 0; t < pd.npos; ++t) need |= 1u << pd.pos[t];
        if (ok && first_come && __builtin_popcount(claimed | need) > kGroupBits) ok = false;
        if (ok && !first_come && (need & ~own)) ok = false;
        if (ok && est + pd.bytes > kTileRecordBudget) ok = false;
        if (!ok) { blocked |= pd.qm; continue; }
        claimed |= need;
        est += pd.bytes;
        ++count;
        if (grp_out) grp_out->push_back(pd.mi);
      }
      if (claimed_out) *claimed_out = first_come ? claimed : own;
      return count;
    };
    unsigned best_own = 0;
    unsigned fc_claimed = 0;
    int best_count = select(0, nullptr, &fc_claimed);
    // The FIRST group of a full tile is free of its LDS read when it lies above the line bits (the kernel loads the
    // tile in that layout, OPC_GROUP_DIRECT): such a triple wins unless another one holds `direct_worth` more ops.
    const bool want_direct = out->empty() && direct_ends;
    auto score = [&](int count, unsigned own_bits) { return count * 2 + ((want_direct && direct_worth && !(own_bits & line_bits)) ? 2 * direct_worth - 1 : 0); };
    int best_score = score(best_count, fc_claimed);
    if (tuning().tile_group_search && __builtin_popcount(cand_mask) > kGroupBits) {
      for (unsigned a = cand_mask; a; a &= a - 1)
        for (unsigned b = a & (a - 1); b; b &= b - 1)
          for (unsigned c = b & (b - 1); c; c &= c - 1) {
            const unsigned own = (a & -a) | (b & -b) | (c & -c);
            const int count = select(own, nullptr, nullptr);
            if (count > 0 && score(count, own) > best_score) { best_score = score(count, own); best_count = count; best_own = own; }
          }
    }
    std::vector<size_t> grp;          // indices into members
    unsigned claimed = 0;
    select(best_own, &grp, &claimed);
    if (grp.empty()) break;           // record budget exhausted: the rest waits for the next launch
    if (emit_group(grp, claimed)) break;   // record budget exhausted inside the group
  }
  if (n_reserved) {
    // the group set aside for the end: an op of it waits for the next launch with everything un-emitted that it follows
    used -= reserve_bytes;
    std::vector<size_t> grp;
    u64 blocked = 0;
    for (size_t mi = 0; mi < members.size(); ++mi) {
      if (done[mi]) continue;
      const u64 qm = op_qmask(ops[members[mi]]);
      if (!reserved[mi] || (blocked & qm)) { blocked |= qm; continue; }
      grp.push_back(mi);
    }
    if (!grp.empty()) emit_group(grp, last_own);
  }

  // The LAST group of a full tile is free of its LDS write-back when it lies above the line bits (OPC_END_DIRECT):
  // a last group that does not is moved in front of its predecessors as long as it shares no qubit with them
  // (groups on disjoint qubits commute), while that leaves a direct-capable group at the end.
  if (tuning().tile_direct && T == kTileBitsMax && out->size() > 1 && out->back().s[0] < low) {
    size_t at = out->size() - 1;
    while (at > 0 && !((*out)[at].qmask & (*out)[at - 1].qmask)) { std::swap((*out)[at], (*out)[at - 1]); --at; }
    if (out->back().s[0] < low)       // nothing gained: keep the original order
      while (at + 1 < out->size()) { std::swap((*out)[at], (*out)[at + 1]); ++at; }
  }
  if (pass_scale != 1.0 && !out->empty()) {   // a global factor commutes with everything: applied once
    // ... for free when the pass has an unconditional dense / real / anti-diagonal 1q gate (every amplitude goes
    // through its matrix: scale the matrix); else as one OPC_SCALE record at the end
    TileDesc* host = nullptr;
    for (TileGroup& g : *out)
      for (TileDesc& d : g.gates) {
        const bool fam = (d.opcode >= OPC_REAL1 && d.opcode < OPC_REAL1 + 3) || (d.opcode >= OPC_DENSE1 && d.opcode < OPC_DENSE1 + 3) ||
                         (d.opcode >= OPC_ANTI1 && d.opcode < OPC_ANTI1 + 3);      // variants 0..2: no register control
        if (fam && !d.blk_mask && !d.outer_mask && !host) host = &d;
      }
    if (host) {
      for (int e = 0; e < host->nd; ++e) host->m[e] *= pass_scale;
    } else {
      TileDesc d;
      std::memset(&d, 0, sizeof d);
      d.opcode = OPC_SCALE;
      d.m[0] = pass_scale;
      d.nd = 1;
      out->back().gates.push_back(d);
    }
  }
}

// Pass builder.  Ops are taken in list order; an op whose non-diagonal targets are not all tile
// bits has to wait and blocks its qubits: its TARGET qubits for every later op, its diagonal qubits (controls,
// phase bits) only for later ops that target them -- two ops that act diagonally on every qubit they share commute
// (CNOTs with a common control, a phase gate on a control, CZ / CR among themselves), so the later one may run a
// pass earlier than the one that waits; likewise ops that act as 1 or X on a shared qubit (CNOTs with a common
// target, X on a CNOT target) (round 3: 155 -> 148 passes over 8 random circuits at 28 qubits, 93 -> 84
// over 6 Clifford+T circuits, 21 -> 20 on the 30-qubit bench circuit; tuning().plan_commute).  Ops on disjoint
// qubits commute anyway; everything else keeps its list order.  WHICH qubits become the tile's
// high bits decides how many ops a pass holds:
//   * first come: an op that still fits claims the bits it needs (the only rule up to r01e);
//   * look-ahead: starting from the first 0, 2, 4, 6 of those bits, add the bit that lets the pass
//     hold the most ops, one at a time;
// the candidate that holds the most ops wins (first come on ties).  On the 28-qubit bench circuit
// this needs 20-21 passes instead of 24 (tests/test_tile_planner_cpu.py); the search costs about
// 0.3 ms per pass on the host, hidden behind the previous pass on the device for large states and
// switched off for small ones; from 26 qubits on each candidate is also scored by what the pass
// AFTER it could hold (one pass fewer on the bench circuits).  QSIM_PLAN_LOOKAHEAD = 0 / 1 / 2 forces.
// `sink(args, T, algorithmic_bytes, first, last)` receives every planned pass (first / last of the op list): the
// launcher on the device path, a serialiser in qsim_plan_ops (the planner itself never touches the GPU).
// `hint` (qsim_apply_ops_tiled): the caller names the high tile bits of the first passes itself -- a host that has planned
// the list once and moved its qubits to other index bits (a layout chosen for the memory pattern of the tiles:
// runner/engine.py) gets the SAME passes on the new bits instead of whatever the search finds there.  A hint that holds no
// op (or too many bits) is ignored and the search takes over for that pass.
struct TileHint { const uint64_t* masks; int n; };
// Peek mode (qsim_plan_peek_pass, the partition planner of runner/partition_plan.py): the op list lives on `n_total` >= k
// index bits; the bits >= k are RANK bits of a partitioned state -- never tile bits, never targets (an op that targets one
// waits, and blocks like any waiting op), but fine as controls and phase bits (a rank applies or skips such an op by its own
// rank bits).  `done` marks the ops that ran already; the builder chooses the tile of the NEXT pass exactly as it would in a
// full plan (candidates, look-ahead), reports it with its members and stops: nothing is emitted.
struct PeekPlan {
  int n_total;
  const uint8_t* done;         // per op of ops_in
  uint64_t avoid;              // index bits the tile should not hold unless its ops need them (the slab bits of a coming re-layout)
  uint64_t tile_mask;          // out: high tile bits (with the fill)
  uint64_t need_mask;          // out: ... of which the pass's ops need
  std::vector<size_t>* members;  // out: indices into ops_in
};
template <class Sink>
static int plan_fused(int k, const std::vector<FusedOp>& ops_in, int* n_passes, Sink&& sink, const TileHint* hint = nullptr,
                      PeekPlan* peek = nullptr) {
  const Tuning& tune = tuning();
  std::vector<FusedOp> ops = ops_in;
  if (tune.tile_commute_fuse == 1 || tune.tile_commute_fuse == 4) commute_fuse_1q(&ops, false);
  if (tune.tile_commute_fuse >= 2) commute_fuse_1q(&ops, true);
  if (tune.tile_commute_fuse == 3) commute_fuse_1q(&ops, false);
  const int T = k < kTileBitsMax ? k : kTileBitsMax;
  const int low = kTileLow;
  const int cap = T - low;                      // tile high-bit capacity
  const size_t n_ops = ops.size();
  std::vector<char> done(n_ops, 0);
  std::vector<u64> qm(n_ops), need(n_ops);     // all qubits of an op; its target bits above the low bits
  std::vector<u64> tm(n_ops);                  // the qubits an op acts on NON-diagonally (its targets); qm & ~tm: controls, phase bits
  std::vector<u64> xm(n_ops);                  // ... of those, the ones it acts on as 1 or X only (X, CNOT targets): X-diagonal
  for (size_t i = 0; i < n_ops; ++i) {
    qm[i] = op_qmask(ops[i]);
    need[i] = 0;
    tm[i] = xm[i] = 0;
    for (int t = 0; t < ops[i].ntargets; ++t) {
      tm[i] |= 1ull << ops[i].target[t];
      if (ops[i].target[t] >= low) need[i] |= 1ull << ops[i].target[t];
    }
    if (ops[i].kind == TG_SWAP1 && tune.plan_commute >= 2) xm[i] = tm[i];
    if (!tune.plan_commute) tm[i] = qm[i];     // (off: every qubit of a waiting op blocks everything on it)
  }
  // Two ops commute when each acts diagonally on every qubit they share (CNOTs with a common control, a phase gate on
  // a control, CZ / CR among themselves ...) -- or, plan_commute >= 2, both as 1 / X (CNOTs with a common target, X on a
  // CNOT target).  An op that has to wait therefore blocks its general TARGET qubits for everything, its diagonal
  // qubits for ops that target them and its X-type targets for everything but X-type targets:
  // `bt` = qubits blocked for all, `bd` = blocked for targets, `bx` = blocked for all but X-type targets.
  auto admissible3 = [&](size_t i, u64 bt, u64 bd, u64 bx) {
    return !(qm[i] & bt) && !(tm[i] & bd) && !((qm[i] & ~xm[i]) & bx);
  };
  const int n_bits = peek ? peek->n_total : k;
  const u64 all_qubits = n_bits >= 64 ? ~0ull : ((1ull << n_bits) - 1);
  const u64 rank_bits = all_qubits & ~(k >= 64 ? ~0ull : ((1ull << k) - 1));   // (peek mode only: no tile bit, no target)
  const bool lookahead = tune.plan_lookahead >= 0 ? tune.plan_lookahead != 0 : k >= 24;
  size_t remaining = n_ops;
  size_t first = 0;
  *n_passes = 0;
  if (peek) {
    remaining = 0;
    for (size_t i = 0; i < n_ops; ++i) { done[i] = peek->done[i] != 0; remaining += !done[i]; }
    peek->tile_mask = peek->need_mask = 0;
    peek->members->clear();
  }
  // How far behind `first` a pass looks for ops (counted in ops not yet done).  The scans below end early once every
  // qubit is blocked for everything (`bt`), which never happens while some qubit is used only as a control, phase bit or
  // X target (a GHZ root, a phase-estimation ancilla, an idle qubit): every scan then walked the whole remaining list --
  // 12 ms of host planning per pass on a 4000-op list, several times the device time of the pass.  An op
  // that far behind the front has a few thousand waiting ops on at most 35 qubits in front of it and is admissible only
  // by accident; it runs a pass later.  Lists shorter than the window (the bench circuits, QFT(33)) plan as before.
  // (byte-identical plans to the unbounded scan on the 28- / 30-qubit bench and Clifford+T circuits from 384 on; lists
  // of more than four windows use two thirds of it: 7.5 -> 1.4 ms per pass on a 4000-op list with a control-only qubit)
  int scan_window = tune.plan_scan_window;
  // ops a pass with the given high bits would hold (optionally listed)
  auto holds = [&](u64 tile_mask, std::vector<size_t>* out) -> int {
    u64 bt = 0, bd = 0, bx = 0;
    int count = 0, seen = 0;
    for (size_t i = first; i < n_ops && count < tune.max_gates_per_pass; ++i) {
      if (done[i]) continue;
      if (++seen > scan_window) break;
      if (!admissible3(i, bt, bd, bx) || (need[i] & ~tile_mask)) {
        bt |= tm[i] & ~xm[i];
        bx |= xm[i];
        bd |= qm[i] & ~tm[i];
        if (bt == all_qubits) break;
        continue;
      }
      ++count;
      if (out) out->push_back(i);
    }
    return count;
  };
  // A pass is memory bound up to ~32 descriptors and pays ~0.03 ms for each one beyond that
  // (tools/gate_cost_probe.py), so holding more than kSaturated ops buys nothing: a first-come pass
  // that is already that full is kept (phase-heavy circuits: QFT).
  constexpr int kSaturated = 64;
  // Memory pattern of a tile (profiles/r02z_tile_bits_samples.txt, 2500 gate-less passes): index bits b and b + 7
  // with b = 13..16 in ONE tile cost +0.13-0.17 ms each at 28 qubits (+9 %; byte-address bits 17-20 select the
  // bank, 24-27 are row bits folded into it: both varying inside a tile puts two rows on one bank), bits 18 / 26 and
  // 19 / 27 half of that.  Counted in units of ops a pass holds (tuning().plan_conflict_cost each).
  auto conflicts = [&](u64 mask) -> int {
    int c = 0;
    for (int b = 13; b <= 16; ++b) c += 2 * (int)(((mask >> b) & (mask >> (b + 7)) & 1));
    for (int b = 18; b <= 19; ++b) c += (int)(((mask >> b) & (mask >> (b + 8)) & 1));
    return c;
  };
  auto penalty = [&](u64 mask) -> int { return tune.plan_conflict_cost > 0 ? tune.plan_conflict_cost * conflicts(mask) / 2 : 0; };
  // plan_conflict_cost = -1: conflicts only break ties between tiles that hold the same number of ops
  const bool tie_break = tune.plan_conflict_cost < 0;
  // (the fitted per-bit + pair model of the same samples as the tie-break instead of the conflict count: 159 instead of
  // 155 passes over 8 circuits, +2.3 % time -- breaking every tie perturbs the greedy growth more than it saves)
  u64 jit_state = 0x9E3779B97F4A7C15ull * (u64)(tune.plan_jitter + 1);
  auto jitter = [&]() -> unsigned { jit_state ^= jit_state << 13; jit_state ^= jit_state >> 7; jit_state ^= jit_state << 17; return (unsigned)(jit_state >> 33); };
  auto keyed = [&](int count, u64 mask) -> int {
    if (tune.plan_jitter > 0) return (count - (int)(jitter() % 8 == 0) - (int)(jitter() % 16 == 0)) * 64 + (int)(jitter() % 64);
    return tie_break ? count * 64 - conflicts(mask) : count;
  };
  auto mask_of = [](const std::vector<int>& bits, size_t n) { u64 m = 0; for (size_t i = 0; i < n && i < bits.size(); ++i) m |= 1ull << bits[i]; return m; };
  // candidate tiles for the next pass from the current `done` / `first`: [0] = first come, then the
  // look-ahead ones grown from the first `seed` claimed bits
  u64 forced_static = 0;                        // (probe) bits that every tile holds
  for (int b = low; b < low + tune.plan_force_low && b < k && __builtin_popcountll(forced_static) < cap; ++b) forced_static |= 1ull << b;
  // Anchored tiles (tuning().plan_anchor = s > 0, evaluation knob of round 4): every tile after the first shares at
  // least s of its high bits with the tile of the pass before it (`prev`): those s qubits can stay on the physical
  // positions next to the line bits (2^(s+7)-byte contiguous pieces per tile, the fast DRAM pattern) while the pass's
  // store permutes the tile's qubits among the tile's positions.  The carried bits are grown greedily from the
  // previous tile (the bit that lets the pass hold the most ops, one at a time).
  const int anchor = tune.plan_anchor;
  auto candidates = [&](std::vector<u64>* out, size_t max_seed, size_t seed_step, u64 prev) {
    u64 forced = forced_static;
    if (anchor > 0 && prev) {
      for (int picked = 0; picked < anchor && picked < cap; ++picked) {
        int pick = -1, pick_count = -(1 << 20);
        for (int b = low; b < k; ++b) {
          if (!((prev >> b) & 1) || ((forced >> b) & 1)) continue;
          const int c = holds(forced | (1ull << b), nullptr);
          if (c > pick_count) { pick_count = c; pick = b; }
        }
        if (pick < 0) break;
        forced |= 1ull << pick;
      }
    }
    const int n_forced = __builtin_popcountll(forced);
    std::vector<int> claimed;                   // high bits in the order they were claimed
    {
      u64 bt = 0, bd = 0, bx = 0, mask = forced;
      int count = 0, seen = 0;
      for (size_t i = first; i < n_ops && count < tune.max_gates_per_pass; ++i) {
        if (done[i]) continue;
        if (++seen > scan_window) break;
        bool ok = admissible3(i, bt, bd, bx);
        const u64 extra = need[i] & ~mask;
        if (ok && (n_forced + (int)claimed.size() + __builtin_popcountll(extra) > cap || (extra & rank_bits))) ok = false;
        if (!ok) { bt |= tm[i] & ~xm[i]; bx |= xm[i]; bd |= qm[i] & ~tm[i]; if (bt == all_qubits) break; continue; }
        for (u64 e = extra; e; e &= e - 1) claimed.push_back(__builtin_ctzll(e));
        mask |= extra;
        ++count;
      }
    }
    out->clear();
    out->push_back(forced | mask_of(claimed, claimed.size()));
    if (!(lookahead && k - low > cap) || holds(out->front(), nullptr) >= kSaturated) return;
    for (size_t seed = 0; seed <= max_seed && seed <= claimed.size(); seed += seed_step) {
      u64 mask = forced | mask_of(claimed, seed);
      while (__builtin_popcountll(mask) < cap) {
        int pick = -1, pick_count = -(1 << 20);
        for (int b = low; b < k; ++b) {
          if ((mask >> b) & 1) continue;
          const int c = keyed(holds(mask | (1ull << b), nullptr) - penalty(mask | (1ull << b)), mask | (1ull << b));
          if (c > pick_count) { pick_count = c; pick = b; }
        }
        if (pick < 0) break;
        mask |= 1ull << pick;
      }
      out->push_back(mask);
    }
  };
  const bool depth2 = tune.plan_lookahead >= 0 ? tune.plan_lookahead >= 2 : k >= 26;
  std::vector<u64> cands, cands2;
  std::vector<size_t> trial;
  u64 prev_mask = 0;                            // high bits of the pass before (anchored tiles)
  while (remaining) {
    while (first < n_ops && done[first]) ++first;
    scan_window = remaining > 4 * (size_t)tune.plan_scan_window ? std::max(1, tune.plan_scan_window * 2 / 3) : tune.plan_scan_window;
    u64 best_mask = 0;
    bool hinted = false;
    if (hint && *n_passes < hint->n) {
      const u64 m = hint->masks[*n_passes] & all_qubits & ~((1ull << low) - 1);
      if (m && __builtin_popcountll(m) <= cap && holds(m, nullptr) > 0) { best_mask = m; hinted = true; }
    }
    if (!hinted) {
    candidates(&cands, 6, 2, prev_mask);
    best_mask = cands[0];
    int best_score = -(1 << 20);
    for (size_t ci = 0; ci < cands.size(); ++ci) {
      trial.clear();
      int score = std::min(holds(cands[ci], &trial), kSaturated) - penalty(cands[ci]);
      if (depth2 && cands.size() > 1 && trial.size() < remaining) {
        // what the pass AFTER this one could hold (a smaller candidate set)
        const size_t first_saved = first;
        for (size_t i : trial) done[i] = 1;
        while (first < n_ops && done[first]) ++first;
        candidates(&cands2, 4, 4, cands[ci]);
        int next_best = 0;
        for (u64 m2 : cands2) next_best = std::max(next_best, std::min(holds(m2, nullptr), kSaturated) - penalty(m2));
        for (size_t i : trial) done[i] = 0;
        first = first_saved;
        score += next_best;
      } else if (depth2 && trial.size() >= remaining) {
        score += 2 * kSaturated;                // finishes the list
      }
      score = keyed(score, cands[ci]);
      if (score > best_score) { best_score = score; best_mask = cands[ci]; }
    }
    }
    std::vector<int> high;                      // chosen high bits
    for (int b = low; b < k; ++b) if ((best_mask >> b) & 1) high.push_back(b);
    std::vector<size_t> members;
    holds(best_mask, &members);
    if (peek) {
      // (everything that is left may be waiting for a rank bit: an empty pass is an answer here, not an error)
      if (members.empty()) { *n_passes = 0; return QSIM_OK; }
      u64 needed = 0;
      for (size_t i : members) needed |= need[i];
      std::vector<int> kept;
      for (int b : high) if ((needed >> b) & 1) kept.push_back(b);
      for (int b = low; (int)kept.size() < cap && b < k; ++b)                        // the fill: not the bits to avoid
        if (!((peek->avoid >> b) & 1) && std::find(kept.begin(), kept.end(), b) == kept.end()) kept.push_back(b);
      high = kept;
      peek->need_mask = needed;
    }
    if (members.empty()) return fail(QSIM_ERR_INVALID, "internal: fused planner made no progress");
    // fill the tile with the lowest unused bits so it always has T bits
    for (int b = low; (int)high.size() < cap && b < k; ++b)
      if (std::find(high.begin(), high.end(), b) == high.end()) high.push_back(b);   // (bits 3.. conflict with nothing)
    std::sort(high.begin(), high.end());
    prev_mask = 0;
    for (int b : high) prev_mask |= 1ull << b;
#ifdef QSIM_PROBES
    if (tune.debug_skip_gates == 2) for (int j = 0; j < cap; ++j) high[j] = low + j;   // contiguous tiles (floor probe)
    if (tune.debug_skip_gates == 3) for (int j = 0; j < cap; ++j) high[j] = k - cap + j; // far-strided tiles
    if (tune.debug_skip_gates == 4) {   // tile bits from QSIM_DEBUG_TILE_BITS="b0,b1,..." (memory-pattern probe)
      if (const char* e = getenv("QSIM_DEBUG_TILE_BITS")) {
        std::vector<int> bits;
        for (const char* p = e; *p;) { bits.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
        if ((int)bits.size() == cap) { high = bits; std::sort(high.begin(), high.end()); }
      }
    }
#endif
    TileArgs a;
    std::memset(&a, 0, sizeof a);
    a.T = T;
    for (size_t j = 0; j < high.size(); ++j) a.h[j] = (uint8_t)high[j];
    std::vector<char> emitted(members.size(), 0);
    std::vector<TileGroup> groups;
    emit_groups(ops, members, high, T, &groups, &emitted);
    if (lookahead && tune.tile_direct && tune.tile_last_search && T == kTileBitsMax) {
      // second plan of the same pass with the last group chosen from the end (see emit_groups); LDS round trips of a
      // tile = group changes + a first group that is not loaded in place + a last group that is not stored in place
      auto trips = [&](const std::vector<TileGroup>& g) {
        return g.empty() ? 1 << 20 : (int)g.size() + (g.front().s[0] < kTileLow) + (g.back().s[0] < kTileLow);
      };
      auto count = [](const std::vector<char>& e) { size_t c = 0; for (char x : e) c += x != 0; return c; };
      std::vector<char> emitted2(members.size(), 0);
      std::vector<TileGroup> groups2;
      emit_groups(ops, members, high, T, &groups2, &emitted2, true);
      if (count(emitted2) > count(emitted) || (count(emitted2) == count(emitted) && trips(groups2) < trips(groups))) {
        groups.swap(groups2);
        emitted.swap(emitted2);
      }
      // (four more variants -- the first group's preference for a direct triple off / stronger, each with and without the
      // last-group search -- save one more round trip in 84 on the bench circuit: not worth three times the planning)
    }
    size_t n_emitted = 0;
    double alg_bytes = 0;   // dense 32N, diagonal / controlled / SWAP 16N, CZ/CR 8N
    for (size_t mi = 0; mi < members.size(); ++mi)
      if (emitted[mi]) {
        const FusedOp& o = ops[members[mi]];
        const int halvings = o.kind == TG_PHASE ? o.nbits : (o.control >= 0 ? 1 : o.halvings);
        alg_bytes += 32.0 * (double)((1ull << k) >> halvings) + 32.0 * (double)(1ull << k) * o.absorbed;
        done[members[mi]] = 1; --remaining; ++n_emitted;
      }
    if (!n_emitted) return fail(QSIM_ERR_INVALID, "internal: fused planner emitted nothing");
    if (peek) {
      // the ops whose records fit the pass (a rank runs this list without the ops its rank bits switch off and with the
      // rank-bit predicates gone: never more records than counted here)
      for (int b : high) peek->tile_mask |= 1ull << b;
      for (size_t mi = 0; mi < members.size(); ++mi) if (emitted[mi]) peek->members->push_back(members[mi]);
      *n_passes = 1;
      return QSIM_OK;
    }
#ifdef QSIM_PROBES
    if (tune.debug_skip_gates) {                   // profiling aid: load -> LDS -> store only (WRONG results)
      groups.resize(1);
      groups[0].gates.clear();
    }
#endif
    int rc = serialize_pass(groups, &a);
    if (rc) return rc;
#ifdef QSIM_PROBES
    // Engine-floor probe (WRONG results): QSIM_DEBUG_REMAP_TILE="b0,...,b7" keeps the pass's record stream but moves
    // its tile to these index bits (j-th smallest tile bit -> j-th listed bit), i.e. the same gates on another
    // memory pattern: what would the pass cost if its tile always had the fastest layout?
    if (const char* e = getenv("QSIM_DEBUG_REMAP_TILE")) {
      std::vector<int> bits;
      for (const char* q = e; *q;) { bits.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; }
      if ((int)bits.size() == cap && T == kTileBitsMax) {
        std::sort(bits.begin(), bits.end());
        uint8_t map[64];
        for (int b = 0; b < 64; ++b) map[b] = (uint8_t)b;
        for (int j = 0; j < cap; ++j) map[a.h[j]] = (uint8_t)bits[(size_t)j];
        for (int j = 0; j < cap; ++j) { a.lay_in[j] = map[a.lay_in[j]]; a.lay_out[j] = map[a.lay_out[j]]; }
        for (int j = 0; j < cap; ++j) a.h[j] = (uint8_t)bits[(size_t)j];
      }
    }
#endif
    if (tune.debug_stats) {
      size_t descs = 0;
      for (const TileGroup& g : groups) descs += g.gates.size();
      std::fprintf(stderr, "[qsim] pass %d: %zu gates, %zu groups, %zu descriptors\n", *n_passes, n_emitted, groups.size(), descs);
    }
    rc = sink(a, T, alg_bytes, *n_passes == 0, remaining == 0);
    if (rc) return rc;
    ++*n_passes;
  }
  return QSIM_OK;
}

// ---- re-layout fused into the first / last pass of an op list (qsim_apply_ops_io) -------------------------------
// Slab layout of an all-to-all re-layout over the local bits `bits` (qsim_pack_all): amplitude i of the chunk sits at
//   d * 2^(k - m) + (i with the m bits removed),   d = sum_j bit(i, bits[j]) << j
// of the buffer, i.e. logical index bit bits[j] is physical bit k - m + j and the others close ranks.
struct SlabLayout {
  int m = 0;
  int bits[3] = {0, 0, 0};
};
static void slab_descriptor(int k, const SlabLayout& L, TileSlab* t) {
  int sorted[3] = {64, 64, 64};
  for (int j = 0; j < L.m; ++j) sorted[j] = L.bits[j];
  std::sort(sorted, sorted + L.m);
  auto below = [](int b) -> u64 { return b >= 64 ? ~0ull : ((1ull << b) - 1); };
  t->field[0] = below(sorted[0]);
  for (int i = 1; i < 4; ++i) {
    const int lo = sorted[i - 1], hi = i < 3 ? sorted[i] : 64;
    t->field[i] = lo >= 64 ? 0 : (below(hi) & ~below(lo + 1));
  }
  for (int j = 0; j < 3; ++j) t->bit[j] = (uint8_t)(j < L.m ? L.bits[j] : 63);
  t->top = (uint8_t)(k - L.m);
  std::memset(t->pad, 0, sizeof t->pad);
}
struct FusedIo {
  const qsim_chunk* src = nullptr;   // first pass reads this buffer (slab layout `in`) instead of the chunk
  SlabLayout in;
  qsim_chunk* dst = nullptr;         // last pass stores into this buffer in slab layout `out` ...
  SlabLayout out;
  qsim_chunk* dst_own = nullptr;     // ... except slab `own_pattern`, which goes to this one (same layout)
  int own_pattern = -1;
  bool parts = false;                // the slab-storing pass is not launched here: it is left pending in the chunk and
                                     // launched slab by slab (qsim_apply_ops_io_part), so each slab's exchange can start early
  bool fused_in = false, fused_out = false;   // results
  bool own_in_chunk = false;         // result: the own slab went into the chunk itself (dst_own == src and ONE pass does it all)
};

// The slab-storing pass of an op list whose caller asked for the split form (qsim_ops_io::dst_parts): planned, not yet
// launched.  The slabs are stored PIECE by piece: piece j = the j-th of 2^nb equal contiguous sub-ranges of EVERY slab
// (the top nb index bits that are not slab bits have the value j), so the exchange of piece j uses all links at once
// while the pieces behind it are still being computed.  The cut depends ONLY on (k, m, pieces asked for): every rank of a
// multi-GPU run cuts alike whatever its own pass plan looks like (ranks plan different op lists: rank-bit phases,
// controlled gates with a global control) -- the k-th transfer between two ranks always has the same size on both sides.
// What a rank's plan decides is only how early its pieces are ready: kTile -- the top `nb_free` <= nb piece bits are not
// tile bits of the pass, which then runs as 2^nb_free partial launches (a launch stores 2^(nb - nb_free) pieces at once;
// nb_free = 0: one launch, everything ready with the first piece); kPack (nothing fusable: the state is final in the
// chunk): one qsim_pack_all piece per call; kDone: stored already (passes that cannot be launched in part).
struct PendingLast {
  enum Mode { kNone = 0, kStashed, kTile, kPack, kDone };   // kStashed: launch_planned has put the pass here, the caller's slab description is still missing
  int mode = kNone;
  TileArgs a;
  int T = 0;
  double alg_bytes = 0;
  int m = 0;
  int32_t bits[3] = {0, 0, 0};
  qsim_chunk* dst = nullptr;
  qsim_chunk* dst_own = nullptr;
  int own_pattern = -1;
  int nb = 0;                        // piece bits: 2^nb pieces
  int piece_bit[3] = {0, 0, 0};      // the top nb non-slab index bits, ascending (piece j: bit i of j <-> piece_bit[i])
  int nb_free = 0;                   // kTile: how many of them, from the top, are no tile bits of the pass
  unsigned stored = 0;               // bit j: piece j has been handed out
  unsigned launched = 0;             // kTile: bit g: partial launch g (the top nb_free bits of the piece number) is queued
};

// 2^nb pieces for a request of `want` (1, 2, 4, 8): as many as asked for while a piece keeps >= 2^min_piece_bits amplitudes.
static int piece_bits_for(int k, int m, int want, int min_piece_bits) {
  int nb = 0;
  while (nb < 3 && (2 << nb) <= want && (k - m) - (nb + 1) >= min_piece_bits && (k - m) - (nb + 1) >= kTileLow) ++nb;
  return nb;
}

static void plan_parts(PendingLast* p, int k, int want, const uint8_t* tile_high, int n_tile_high, int min_piece_bits) {
  auto is_slab = [&](int b) { for (int i = 0; i < p->m; ++i) if (p->bits[i] == b) return true; return false; };
  auto is_tile = [&](int b) { for (int j = 0; j < n_tile_high; ++j) if (tile_high[j] == b) return true; return false; };
  p->nb = piece_bits_for(k, p->m, want, min_piece_bits);
  int top[3], found = 0;
  for (int b = k - 1; b >= 0 && found < p->nb; --b) if (!is_slab(b)) top[found++] = b;    // descending
  for (int i = 0; i < p->nb; ++i) p->piece_bit[i] = top[p->nb - 1 - i];
  p->nb_free = 0;
  while (p->nb_free < p->nb && !is_tile(top[p->nb_free])) ++p->nb_free;
  p->stored = p->launched = 0;
}

// One planned pass on its way to the device: buffers and the re-layout of the op list's ends (prepare_planned), then the
// launch -- or, for the slab-storing pass of a split call, the stash (dispatch_planned).
static void prepare_planned(qsim_chunk* c, TileArgs& a, int T, bool first, bool last, FusedIo* io) {
  a.amp = c->amp;
  a.amp_out = c->amp;
  a.amp_out_own = nullptr;
  a.perm = 0;
  if (io && first && io->src) {
    a.amp = io->src->amp;
    a.perm |= kTilePermIn;
    slab_descriptor(c->k, io->in, &a.slab_in);
    io->fused_in = true;
  }
  if (io && last && io->dst) {
    // the slab that stays is chosen per tile from the tile's base: its bits must lie outside the tile
    bool clash = false;
    for (int j = 0; j < T - kTileLow; ++j)
      for (int i = 0; i < io->out.m; ++i) clash = clash || a.h[j] == io->out.bits[i];
    if (!clash || io->own_pattern < 0) {
      a.amp_out = io->dst->amp;
      a.perm |= kTilePermOut;
      slab_descriptor(c->k, io->out, &a.slab_out);
      if (io->own_pattern >= 0) {
        a.perm |= kTileOwnOut;
        a.amp_out_own = io->dst_own->amp;
        if (first && io->src && io->dst_own->amp == io->src->amp) {
          // ONE pass reads the source buffer and stores the slabs, and the caller named the source as the place of the
          // slab that stays (three shard-sized buffers per rank instead of four): a tile of that slab must not overwrite
          // source lines other tiles still read -- it goes into the chunk itself, whose contents nobody needs now
          a.amp_out_own = c->amp;
          io->own_in_chunk = true;
        }
        a.own_mask = a.own_value = 0;
        for (int i = 0; i < io->out.m; ++i) {
          a.own_mask |= 1ull << io->out.bits[i];
          if ((io->own_pattern >> i) & 1) a.own_value |= 1ull << io->out.bits[i];
        }
      }
      io->fused_out = true;
    }
  }
}

static int dispatch_planned(qsim_chunk* c, TileArgs& a, int T, double alg_bytes, bool last, FusedIo* io) {
  if (io && last && io->dst && io->parts && io->fused_out && T == kTileBitsMax) {
    PendingLast* p = c->pending ? c->pending : (c->pending = new PendingLast());
    p->mode = PendingLast::kStashed; p->a = a; p->T = T; p->alg_bytes = alg_bytes; p->launched = 0;
    return QSIM_OK;                  // (qsim_apply_ops_io fills in the slab description and keeps it pending)
  }
  if (tuning().debug_stats < 2) return launch_tile_any(a, T, c, c->stream, alg_bytes);
  // QSIM_DEBUG_STATS=2: time every pass synchronously and print its shape (profiling aid)
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, c->stream);
  const int rc = launch_tile_any(a, T, c, c->stream, alg_bytes);
  (void)hipEventRecord(e1, c->stream);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::fprintf(stderr, "[qsim] timed pass: %.3f ms, %d records, high bits", ms, a.nrec);
  for (int j = 0; j < T - kTileLow; ++j) std::fprintf(stderr, " %d", a.h[j]);
  std::fprintf(stderr, "\n");
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}

static int launch_planned(qsim_chunk* c, TileArgs& a, int T, double alg_bytes, bool first, bool last, FusedIo* io) {
  prepare_planned(c, a, T, first, last, io);
  return dispatch_planned(c, a, T, alg_bytes, last, io);
}

// ---- plan cache ---------------------------------------------------------------------------------------------------
// Runners execute the same op list again and again (a planned circuit repeated, one plan per chunk of a chunked run):
// the pass images of the last few op lists are kept, keyed by the exact bytes of the call (local qubits, arities,
// qubits, matrices), and launched again without planning.  Planning overlaps with the device for large states (the
// images of pass p + 1 are made while pass p runs), so this matters where passes are short: 13 % of a step at 22
// qubits, nothing at 28.  Images are stored without buffers or layouts: those are put in at
// launch time (launch_planned), so one entry serves every chunk of its size and every re-layout.
struct CachedPass { TileArgs a; int T; double alg_bytes; };
struct CachedPlan {
  int k = 0;
  u64 hash = 0;
  std::vector<unsigned char> key;        // nq | qubits | mats of the call
  std::vector<CachedPass> passes;
};
constexpr size_t kPlanCacheEntries = 8;
constexpr size_t kPlanCacheMaxKeyBytes = 4u << 20;      // longer op lists are planned every time
static std::list<CachedPlan> g_plan_cache;             // most recently used first
static std::mutex g_plan_cache_mu;

static u64 fnv1a(const unsigned char* p, size_t n, u64 h = 1469598103934665603ull) {
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

// `defer` (an op list whose source arrives in pieces, qsim_ops_io::src_parts): the passes are planned and prepared (buffers
// in) but not launched; the caller launches them when the source is complete.
static int run_fused(qsim_chunk* c, const std::vector<FusedOp>& ops, int* n_passes, FusedIo* io = nullptr,
                     int n_ops = 0, const int32_t* nq = nullptr, const int32_t* qubits = nullptr, const double* mats = nullptr,
                     std::vector<CachedPass>* defer = nullptr, const TileHint* hint = nullptr) {
  // key of the call (only when the caller handed the raw op list over)
  std::vector<unsigned char> key;
  u64 hash = 0;
  const size_t hint_bytes = hint ? sizeof(uint64_t) * (size_t)hint->n : 0;
  const size_t key_bytes = (size_t)n_ops * (sizeof(int32_t) * 3 + sizeof(double) * 32) + hint_bytes;
  const bool cacheable = nq && qubits && mats && n_ops > 0 && key_bytes <= kPlanCacheMaxKeyBytes && tuning().debug_stats == 0 &&
                         tuning().debug_skip_gates == 0;     // (probe passes take their tile bits from the environment at plan time)
  if (cacheable) {
    key.resize(key_bytes);
    unsigned char* w = key.data();
    std::memcpy(w, nq, sizeof(int32_t) * (size_t)n_ops); w += sizeof(int32_t) * (size_t)n_ops;
    std::memcpy(w, qubits, sizeof(int32_t) * 2 * (size_t)n_ops); w += sizeof(int32_t) * 2 * (size_t)n_ops;
    std::memcpy(w, mats, sizeof(double) * 32 * (size_t)n_ops); w += sizeof(double) * 32 * (size_t)n_ops;
    if (hint_bytes) std::memcpy(w, hint->masks, hint_bytes);
    hash = fnv1a(key.data(), key.size(), 1469598103934665603ull ^ (u64)c->k);
    std::vector<CachedPass> hit;
    {
      std::lock_guard<std::mutex> lock(g_plan_cache_mu);
      for (auto it = g_plan_cache.begin(); it != g_plan_cache.end(); ++it)
        if (it->hash == hash && it->k == c->k && it->key == key) {
          g_plan_cache.splice(g_plan_cache.begin(), g_plan_cache, it);
          hit = g_plan_cache.front().passes;
          break;
        }
    }
    if (!hit.empty()) {
      for (size_t p = 0; p < hit.size(); ++p) {
        if (defer) {
          prepare_planned(c, hit[p].a, hit[p].T, p == 0, p + 1 == hit.size(), io);
          defer->push_back(hit[p]);
          continue;
        }
        const int rc = launch_planned(c, hit[p].a, hit[p].T, hit[p].alg_bytes, p == 0, p + 1 == hit.size(), io);
        if (rc) return rc;
      }
      *n_passes = (int)hit.size();
      return QSIM_OK;
    }
  }
  std::vector<CachedPass> made;
  const int rc = plan_fused(c->k, ops, n_passes, [&](TileArgs& a, int T, double alg_bytes, bool first, bool last) {
    if (cacheable) made.push_back(CachedPass{a, T, alg_bytes});      // (before buffers and layouts go in)
    if (defer) {
      prepare_planned(c, a, T, first, last, io);
      defer->push_back(CachedPass{a, T, alg_bytes});
      return (int)QSIM_OK;
    }
    return launch_planned(c, a, T, alg_bytes, first, last, io);
  }, hint);
  if (rc == QSIM_OK && cacheable && !made.empty()) {
    std::lock_guard<std::mutex> lock(g_plan_cache_mu);
    g_plan_cache.emplace_front();
    CachedPlan& e = g_plan_cache.front();
    e.k = c->k; e.hash = hash; e.key.swap(key); e.passes.swap(made);
    while (g_plan_cache.size() > kPlanCacheEntries) g_plan_cache.pop_back();
  }
  return rc;
}
