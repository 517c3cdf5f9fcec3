/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's dense butterflies,
 * used (a) as a second checker next to oracle/dense_oracle.py and (b) as the timed
 * `cpu_baseline` ("port") leg of bench.py on the GPU node's host cores.
 *
 * Parity status: PINNED -- tests/test_oracle_c.py checks every function against the golden
 * vectors generated from the reference (tests/golden/kernels.npz, states.npz).
 *
 * Restates (paths under /root/reference):
 *   orc_apply_1q ........ wenbo_engine/kernel/cpu_scalar.py:21-32 (== ref_dense.py:13-23)
 *   orc_apply_2q ........ wenbo_engine/kernel/cpu_scalar.py:35-47 (== ref_dense.py:32-41)
 *   orc_apply_1q_pair ... wenbo_engine/kernel/cpu_nonlocal.py:22-26
 * Layout: interleaved (re, im) doubles, amplitude i at psi[2i], psi[2i+1]; qubit q <-> bit q
 * of i; 2q matrices row-major 4x4, big-endian inside the pair (qa = MSB).
 * The product library never links or loads this file.
 */
#include <stdint.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double re, im; } cplx;

static inline cplx cmul(cplx a, cplx b) {
  cplx r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
  return r;
}
static inline cplx cadd(cplx a, cplx b) { cplx r = { a.re + b.re, a.im + b.im }; return r; }

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_apply_1q(double* psi_, int n_qubits, int q, const double* U_) {
  if (q < 0 || q >= n_qubits) return -2;  /* the reference's non-local NotImplementedError */
  cplx* psi = (cplx*)psi_;
  const cplx* U = (const cplx*)U_;
  const int64_t pairs = (int64_t)1 << (n_qubits - 1);
  const int64_t low = ((int64_t)1 << q) - 1;
  const int64_t step = (int64_t)1 << q;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < pairs; ++c) {
    const int64_t i0 = ((c & ~low) << 1) | (c & low);
    const cplx a = psi[i0], b = psi[i0 + step];
    psi[i0] = cadd(cmul(U[0], a), cmul(U[1], b));
    psi[i0 + step] = cadd(cmul(U[2], a), cmul(U[3], b));
  }
  return 0;
}

int orc_apply_2q(double* psi_, int n_qubits, int qa, int qb, const double* U_) {
  if (qa < 0 || qb < 0 || qa >= n_qubits || qb >= n_qubits) return -2;
  if (qa == qb) return -1;
  cplx* psi = (cplx*)psi_;
  const cplx* U = (const cplx*)U_;
  const int lo = qa < qb ? qa : qb, hi = qa < qb ? qb : qa;
  const int64_t quads = (int64_t)1 << (n_qubits - 2);
  const int64_t lo_mask = ((int64_t)1 << lo) - 1;
  const int64_t hi_mask = ((int64_t)1 << hi) - 1;
  const int64_t sa = (int64_t)1 << qa, sb = (int64_t)1 << qb;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < quads; ++c) {
    int64_t i = ((c & ~lo_mask) << 1) | (c & lo_mask);
    i = ((i & ~hi_mask) << 1) | (i & hi_mask);
    const int64_t idx[4] = { i, i | sb, i | sa, i | sa | sb };
    cplx v[4], r[4];
    for (int m = 0; m < 4; ++m) v[m] = psi[idx[m]];
    for (int row = 0; row < 4; ++row) {
      cplx acc = cmul(U[4 * row], v[0]);
      for (int col = 1; col < 4; ++col) acc = cadd(acc, cmul(U[4 * row + col], v[col]));
      r[row] = acc;
    }
    for (int m = 0; m < 4; ++m) psi[idx[m]] = r[m];
  }
  return 0;
}

int orc_apply_1q_pair(double* c0_, double* c1_, int k, const double* U_) {
  cplx* c0 = (cplx*)c0_;
  cplx* c1 = (cplx*)c1_;
  const cplx* U = (const cplx*)U_;
  const int64_t n = (int64_t)1 << k;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const cplx a = c0[i], b = c1[i];
    c0[i] = cadd(cmul(U[0], a), cmul(U[1], b));
    c1[i] = cadd(cmul(U[2], a), cmul(U[3], b));
  }
  return 0;
}

/* |0..0> then ops in order: nq[i] in {1,2}, qubits[2i..], mats + 32*i (ref_dense.simulate,
 * ref_dense.py:44-57, after host-side gate-matrix construction). */
int orc_run_ops(double* psi, int n_qubits, int n_ops, const int32_t* nq, const int32_t* qubits,
                const double* mats, int init_zero_state) {
  const int64_t N = (int64_t)1 << n_qubits;
  if (init_zero_state) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < 2 * N; ++i) psi[i] = 0.0;
    psi[0] = 1.0;
  }
  for (int i = 0; i < n_ops; ++i) {
    int rc = nq[i] == 1 ? orc_apply_1q(psi, n_qubits, qubits[2 * i], mats + 32 * (size_t)i)
                        : orc_apply_2q(psi, n_qubits, qubits[2 * i], qubits[2 * i + 1],
                                       mats + 32 * (size_t)i);
    if (rc) return rc;
  }
  return 0;
}
