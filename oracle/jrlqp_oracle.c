/*
 * jrlqp_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see jrlqp_oracle.h).
 *
 * Plain-C restatement of the reference's dense Goldfarb-Idnani path.  Every function cites the
 * reference lines it follows (paths relative to the jrl-qp tree).
 *
 * CANON rules (oracle/README.md): the reference leaves summation order to Eigen; here it is fixed:
 *   C1  every accumulation of products is a chain  acc = fma(a_k, b_k, acc)  in the order in which
 *       the operands become available: increasing k for dots, GEMV rows/columns, Cholesky updates
 *       and forward substitution; decreasing k for backward substitution;
 *   C2  a*b+c written in the reference as one expression is one fma; every other operation is a
 *       single correctly-rounded IEEE-754 binary64 operation (/, sqrt included), no contraction
 *       (this file must be compiled with -ffp-contract=off and without -ffast-math);
 *   C3  plane rotations:  x' = fma(c, x, -(s*y)),  y' = fma(s, x, c*y).
 *   C5  the two scalar products of a pass that need the step direction z -- z.z (the ||z|| test, GoldfarbIdnaniSolver.cpp:174,309)
 *       and n+.z (ConstraintNormal::dot, ConstraintNormal.h:105-123: the step length :183-190, the objective update
 *       DualSolver.cpp:147, addInitialConstraint :309-321) -- are C4 reductions (orc_dot_tree, blockgi_oracle.c): one product per
 *       element up to n = 1024 (element e joins chain e mod 1024), combined by the adjacent-first pairwise tree.  Eigen's own
 *       dot() is a tree of SIMD partial sums as well (packet width unspecified), so this is as faithful to the reference as the
 *       single chain of C1 was -- and it is the shape a wavefront evaluates in log2(64) exchanges instead of n dependent fmas
 *       on the critical path of every pass.  n+.x stays the C1 chain: it is bit for bit the value C.col(p).dot(x) the constraint
 *       scan has just computed for the selected constraint (:97), and the kernels reuse it.
 *   C8  (round 4) beyond ORC_BLOCK_N = 256 unknowns -- i.e. for the host-driven solver of ONE large problem only; the batched kernels
 *       stop at 256 -- the three streaming products of a pass, d = J^T n+ (ConstraintNormal::preMultiplyByMt, ConstraintNormal.h:81-102),
 *       the scan's C.col(i).dot(x) (GoldfarbIdnaniSolver.cpp:97, and n+.x of :183-190, the same expression) and z = J2 d2 (:145), are
 *       BLOCKED: the index range of a sum is cut into segments of 64 consecutive indices (segment s = [64 s, 64 s + 63]), inside a
 *       segment the terms form one ascending fma chain from +0 (C1), and the at most 64 segment sums are combined by the adjacent-first
 *       pairwise tree of C4 / C5 (orc_dot_blk).  Eigen's own GEMV is blocked and vectorised too (its packet width and blocking are
 *       unspecified), so this is as faithful as the single chain was; on a GPU a 1024-term chain per output is ~4 us of dependent
 *       fmas whatever the bandwidth, a 64-term chain lets every CU stream J and C.  z sums over the columns q .. n-1: segments by
 *       ABSOLUTE column index, a segment's chain starting at max(64 s, q).
 */
#include "jrlqp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <omp.h>
#include <string.h>

#define FMA(a, b, c) __builtin_fma((a), (b), (c))

/* LITERAL mode (orc_gi_solve_literal, tests/test_oracle_literal.py).  Rounds 1-3 moved three evaluations away from the most
 * literal reading of the reference's source: C5 (z.z and n+.z as trees), C6 (which product of a rotation is rounded) and C7 (the
 * Givens coefficients of an addition from a suffix sum instead of the serial makeGivens recurrence of
 * GoldfarbIdnaniSolver.cpp:225-232).  Each is one of Eigen's admissible evaluations, but only a test can show that no DECISION
 * (termination status, iteration count, activation statuses, active order) depends on it.  The flags below switch each change off
 * per calling thread; the canonical entry points always run with 0.
 *   ORC_LIT_CHAIN_DOTS   z.z and n+.z as in-order chains (C1), as GoldfarbIdnaniSolver.cpp:174,183-190 read left to right
 *   ORC_LIT_ROT_C3       rotations as rounds 1-2 had them: x' = fma(c, x, -(s y)), y' = fma(s, x, c y)
 *   ORC_LIT_SERIAL_GIVENS  makeGivens(d_i, d_{i+1}) for i = n-2 .. q, radius carried in d_i, exactly :225-232
 *   ORC_LIT_NO_FMA       the dots and rotations above with every product and sum rounded on its own (what a build of Eigen
 *                        without FMA instructions evaluates, scalar path) */
static __thread int g_lit = 0;

static double lit_dot(int n, const double * a, const double * b)
{
  double acc = 0.0;
  if(g_lit & ORC_LIT_NO_FMA)
    for(int k = 0; k < n; ++k) acc = acc + a[k] * b[k];
  else
    for(int k = 0; k < n; ++k) acc = FMA(a[k], b[k], acc);
  return acc;
}

/* the scalar products of a pass that need z: C5 tree (canonical) or the literal chain */
static double zdot(int n, const double * a, const double * b)
{
  return (g_lit & (ORC_LIT_CHAIN_DOTS | ORC_LIT_NO_FMA)) ? lit_dot(n, a, b) : orc_dot_tree(n, a, b, 0);
}

/* ------------------------------------------------------------------------------------------- */
/* canonical primitives                                                                        */
/* ------------------------------------------------------------------------------------------- */

/* C1 dot product (Eigen dot/GEMV call sites: GoldfarbIdnaniSolver.cpp:73,97,183-190, ConstraintNormal.h:87-113) */
/* optional per-pass vector trace of the NEXT orc_gi_solve calls of this thread (tests only; NULL / 0: off) */
static __thread double * g_vtrace = NULL;
static __thread int g_vtrace_cap = 0;
void orc_set_vector_trace(double * buf, int cap)
{
  g_vtrace = cap > 0 ? buf : NULL;
  g_vtrace_cap = cap > 0 ? cap : 0;
}

double orc_dot(int n, const double * a, const double * b)
{
  double acc = 0.0;
  for(int k = 0; k < n; ++k) acc = FMA(a[k], b[k], acc);
  return acc;
}

/* CANON C8: the blocked product for vectors of more than ORC_BLOCK_N entries (shorter ones: the C1 chain).  Terms lo .. n-1. */
#define ORC_BLOCK_N 256
double orc_dot_blk(int n, const double * a, const double * b, int lo)
{
  if(n <= ORC_BLOCK_N || (g_lit & ORC_LIT_CHAIN_GEMV))
  {
    double acc = 0.0;
    for(int k = lo; k < n; ++k) acc = FMA(a[k], b[k], acc);
    return acc;
  }
  double p[64];
  const int S = (n + 63) >> 6;
  for(int s = 0; s < 64; ++s)
  {
    double acc = 0.0;
    if(s < S)
    {
      const int k0 = (64 * s > lo) ? 64 * s : lo, k1 = (64 * s + 64 < n) ? 64 * s + 64 : n;
      for(int k = k0; k < k1; ++k) acc = FMA(a[k], b[k], acc);
    }
    p[s] = acc;
  }
  for(int h = 1; h < 64; h <<= 1)
    for(int i = 0; i < 64; i += 2 * h) p[i] = p[i] + p[i + h];
  return p[0];
}

/* out[j] = orc_dot_blk(n, M + j*ldm, v, 0) for j < ncols, eight segment chains of a column in flight (a lone 64-term chain is bound
 * by the latency of its fmas) */
static void multi_dot_blk(int n, const double * M, long ldm, int ncols, const double * v, double * out)
{
  const int S = (n + 63) >> 6;
  for(int j = 0; j < ncols; ++j)
  {
    const double * c = M + (long)j * ldm;
    double p[64];
    int s = 0;
    for(; s + 8 <= S && 64 * (s + 8) <= n; s += 8) /* eight FULL segments */
    {
      const double *c0 = c + 64 * s, *v0 = v + 64 * s;
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
      for(int k = 0; k < 64; ++k)
      {
        a0 = FMA(c0[k], v0[k], a0);
        a1 = FMA(c0[k + 64], v0[k + 64], a1);
        a2 = FMA(c0[k + 128], v0[k + 128], a2);
        a3 = FMA(c0[k + 192], v0[k + 192], a3);
        a4 = FMA(c0[k + 256], v0[k + 256], a4);
        a5 = FMA(c0[k + 320], v0[k + 320], a5);
        a6 = FMA(c0[k + 384], v0[k + 384], a6);
        a7 = FMA(c0[k + 448], v0[k + 448], a7);
      }
      p[s] = a0; p[s + 1] = a1; p[s + 2] = a2; p[s + 3] = a3; p[s + 4] = a4; p[s + 5] = a5; p[s + 6] = a6; p[s + 7] = a7;
    }
    for(; s < S; ++s)
    {
      const int k1 = (64 * s + 64 < n) ? 64 * s + 64 : n;
      double acc = 0.0;
      for(int k = 64 * s; k < k1; ++k) acc = FMA(c[k], v[k], acc);
      p[s] = acc;
    }
    for(; s < 64; ++s) p[s] = 0.0;
    for(int h = 1; h < 64; h <<= 1)
      for(int i = 0; i < 64; i += 2 * h) p[i] = p[i] + p[i + h];
    out[j] = p[0];
  }
}

/* out[j] = orc_dot(n, M + j*ldm, v) for j < ncols: same per-output chain as orc_dot (bit-identical), eight
 * independent chains in flight so that the CPU baseline is not bound by fma latency.  When the column stride is a multiple
 * of 4 KiB (n = 1024: BASELINE c2) eight column streams plus v land in one set of an 8-way L1 and evict each other -- the
 * baseline would run 4x slower than at n = 1000 for no algorithmic reason -- so that case walks six columns at a time. */
static void multi_dot(int n, const double * M, long ldm, int ncols, const double * v, double * out)
{
  if(n > ORC_BLOCK_N && !(g_lit & ORC_LIT_CHAIN_GEMV)) /* CANON C8 */
  {
    multi_dot_blk(n, M, ldm, ncols, v, out);
    return;
  }
  int j = 0;
  if((ldm * (long)sizeof(double)) % 4096 == 0)
  {
    for(; j + 6 <= ncols; j += 6)
    {
      const double * c0 = M + (long)j * ldm;
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
      for(int k = 0; k < n; ++k)
      {
        double vk = v[k];
        a0 = FMA(c0[k], vk, a0);
        a1 = FMA(c0[k + ldm], vk, a1);
        a2 = FMA(c0[k + 2 * ldm], vk, a2);
        a3 = FMA(c0[k + 3 * ldm], vk, a3);
        a4 = FMA(c0[k + 4 * ldm], vk, a4);
        a5 = FMA(c0[k + 5 * ldm], vk, a5);
      }
      out[j] = a0; out[j + 1] = a1; out[j + 2] = a2; out[j + 3] = a3; out[j + 4] = a4; out[j + 5] = a5;
    }
    for(; j < ncols; ++j) out[j] = orc_dot(n, M + (long)j * ldm, v);
  }
  for(; j + 8 <= ncols; j += 8)
  {
    const double * c0 = M + (long)j * ldm;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    for(int k = 0; k < n; ++k)
    {
      double vk = v[k];
      a0 = FMA(c0[k], vk, a0);
      a1 = FMA(c0[k + ldm], vk, a1);
      a2 = FMA(c0[k + 2 * ldm], vk, a2);
      a3 = FMA(c0[k + 3 * ldm], vk, a3);
      a4 = FMA(c0[k + 4 * ldm], vk, a4);
      a5 = FMA(c0[k + 5 * ldm], vk, a5);
      a6 = FMA(c0[k + 6 * ldm], vk, a6);
      a7 = FMA(c0[k + 7 * ldm], vk, a7);
    }
    out[j] = a0; out[j + 1] = a1; out[j + 2] = a2; out[j + 3] = a3;
    out[j + 4] = a4; out[j + 5] = a5; out[j + 6] = a6; out[j + 7] = a7;
  }
  for(; j < ncols; ++j) out[j] = orc_dot(n, M + (long)j * ldm, v);
}

/* Eigen::internal::llt_inplace<double,Lower> (call sites GoldfarbIdnaniSolver.cpp:58, blockArrowLLT.cpp:64,78,
 * triBlockDiagLLT.cpp:18,31).  Returns -1 on success, else the index of the first non-positive pivot;
 * nothing is written at or after that column.  Only the lower triangle is read/written. */
int orc_llt(int n, double * A, int lda)
{
  /* Element (i, j) is A(i,j) minus L(i,k) L(j,k) for k = 0 .. j-1, one fma per k IN THAT ORDER, then the division by the pivot.
   * The loops below walk k outside and i inside, so that every access is down a column (the row-wise form runs through memory
   * with a stride of lda: at lda = 1024 every step of a chain falls in the same cache set); column j is built in a
   * scratch vector and only stored once its pivot is known to be positive. */
  double tmp_small[128];
  double * tmp = (n <= 128) ? tmp_small : (double *)malloc(sizeof(double) * (size_t)n);
  int info = -1;
  for(int j = 0; j < n; ++j)
  {
    double * cj = A + (long)j * lda;
    const int len = n - j;
    for(int i = 0; i < len; ++i) tmp[i] = cj[j + i];
    for(int k = 0; k < j; ++k)
    {
      const double * ck = A + (long)k * lda + j;
      const double ljk = ck[0];
      for(int i = 0; i < len; ++i) tmp[i] = FMA(-ck[i], ljk, tmp[i]);
    }
    if(tmp[0] <= 0.0)
    {
      info = j;
      break;
    }
    const double ljj = sqrt(tmp[0]);
    cj[j] = ljj;
    for(int i = 1; i < len; ++i) cj[j + i] = tmp[i] / ljj;
  }
  if(tmp != tmp_small) free(tmp);
  return info;
}

/* J.setIdentity(); L.transpose().solveInPlace(J)  (GoldfarbIdnaniSolver.cpp:64-66).
 * Backward substitution per column; entries below the diagonal are exact zeros. */
void orc_inv_lt(int n, const double * L, int ldl, double * J, int ldj)
{
  /* (the update of col[0 .. k) by row k of L runs down a column of a transposed copy of L: same operations, unit stride) */
  double lt_small[64 * 64];
  double * LT = (n <= 64) ? lt_small : (double *)malloc(sizeof(double) * (size_t)n * n);
  for(int i0 = 0; i0 < n; i0 += 32)
    for(int k = i0; k < n; ++k)
    {
      const int i1 = (i0 + 32 < k) ? i0 + 32 : k;
      for(int i = i0; i < i1; ++i) LT[i + (long)k * n] = L[k + (long)i * ldl];
    }
  for(int c = 0; c < n; ++c)
  {
    double * col = J + (long)c * ldj;
    for(int i = 0; i < n; ++i) col[i] = (i == c) ? 1.0 : 0.0;
    for(int k = c; k >= 0; --k)
    {
      double v = col[k] / L[k + (long)k * ldl];
      col[k] = v;
      const double * lk = LT + (long)k * n;
      for(int i = 0; i < k; ++i) col[i] = FMA(-lk[i], v, col[i]);
    }
  }
  if(LT != lt_small) free(LT);
}

/* Eigen::JacobiRotation<double>::makeGivens(p, q, &r), real case (call sites
 * GoldfarbIdnaniSolver.cpp:229,249).  [c -s; s c] * [p; q] = [r; 0]. */
void orc_givens(double p, double q, double * c, double * s, double * r)
{
  if(q == 0.0)
  {
    *c = p < 0.0 ? -1.0 : 1.0;
    *s = 0.0;
    *r = fabs(p);
  }
  else if(p == 0.0)
  {
    *c = 0.0;
    *s = q < 0.0 ? 1.0 : -1.0;
    *r = fabs(q);
  }
  else if(fabs(p) > fabs(q))
  {
    double t = q / p;
    double u = sqrt(FMA(t, t, 1.0));
    if(p < 0.0) u = -u;
    *c = 1.0 / u;
    *s = -t * (*c);
    *r = p * u;
  }
  else
  {
    double t = p / q;
    double u = sqrt(FMA(t, t, 1.0));
    if(q < 0.0) u = -u;
    *s = -1.0 / u;
    *c = -t * (*s);
    *r = q * u;
  }
}

/* C3: (x, y) <- (c x - s y, s x + c y); used for J.applyOnTheRight(i,i+1,G) and
 * R.applyOnTheLeft(i,i+1,G^T) (GoldfarbIdnaniSolver.cpp:231,251-252). */
static inline void rot(double c, double s, double * x, double * y)
{
  double xv = *x, yv = *y;
  if(__builtin_expect(g_lit & (ORC_LIT_ROT_C3 | ORC_LIT_NO_FMA), 0))
  {
    if(g_lit & ORC_LIT_NO_FMA)
    {
      *x = c * xv - s * yv;
      *y = s * xv + c * yv;
    }
    else
    {
      *x = FMA(c, xv, -(s * yv));
      *y = FMA(s, xv, c * yv);
    }
    return;
  }
  *x = FMA(-s, yv, c * xv); /* CANON C6: in each output the product of the OTHER operand's partner is rounded first -- c x here, */
  *y = FMA(s, xv, c * yv);  /* c y there -- so that a sweep's carried value (y in a descending sweep, x in an ascending one)
                             * enters one fma per rotation and nothing else */
}

/* ------------------------------------------------------------------------------------------- */
/* ActiveSet  (src/internal/ActiveSet.cpp:45-168, include/jrl-qp/internal/ActiveSet.h:76-148)  */
/* ------------------------------------------------------------------------------------------- */
struct orc_active_set
{
  int ncstr, nbnd;
  int * status; /* ncstr + nbnd */
  int * active; /* ordered list of active indices */
  int nact;
  int me, mi, ml, mu, mb, mbl, mbu, mbe;
};

static void as_count(orc_active_set * A, int status, int d)
{
  switch(status)
  {
    case ORC_LOWER: A->mi += d; A->ml += d; break;
    case ORC_UPPER: A->mi += d; A->mu += d; break;
    case ORC_EQUALITY: A->me += d; break;
    case ORC_LOWER_BOUND: A->mb += d; A->mbl += d; break;
    case ORC_UPPER_BOUND: A->mb += d; A->mbu += d; break;
    case ORC_FIXED: A->mb += d; A->mbe += d; break;
    default: break;
  }
}

void orc_as_reset(orc_active_set * A)
{
  for(int i = 0; i < A->ncstr + A->nbnd; ++i) A->status[i] = ORC_INACTIVE;
  A->nact = 0;
  A->me = A->mi = A->ml = A->mu = A->mb = A->mbl = A->mbu = A->mbe = 0;
}

orc_active_set * orc_as_new(int ncstr, int nbnd)
{
  orc_active_set * A = (orc_active_set *)malloc(sizeof(orc_active_set));
  A->ncstr = ncstr;
  A->nbnd = nbnd;
  A->status = (int *)malloc(sizeof(int) * (size_t)(ncstr + nbnd + 1));
  A->active = (int *)malloc(sizeof(int) * (size_t)(ncstr + nbnd + 1));
  orc_as_reset(A);
  return A;
}

void orc_as_free(orc_active_set * A)
{
  if(!A) return;
  free(A->status);
  free(A->active);
  free(A);
}

void orc_as_activate(orc_active_set * A, int idx, int status)
{
  A->active[A->nact++] = idx; /* push_back, ActiveSet.cpp:99 */
  A->status[idx] = status;
  as_count(A, status, +1);
}

void orc_as_deactivate(orc_active_set * A, int pos)
{
  int idx = A->active[pos];
  int status = A->status[idx];
  for(int k = pos; k + 1 < A->nact; ++k) A->active[k] = A->active[k + 1]; /* erase, ActiveSet.cpp:137 */
  A->nact--;
  A->status[idx] = ORC_INACTIVE;
  as_count(A, status, -1);
}

int orc_as_status(const orc_active_set * A, int idx) { return A->status[idx]; }
int orc_as_at(const orc_active_set * A, int pos) { return A->active[pos]; }
int orc_as_counters(const orc_active_set * A, int * c)
{
  c[0] = A->me + A->mi + A->mb;
  c[1] = A->me;
  c[2] = A->mi;
  c[3] = A->ml;
  c[4] = A->mu;
  c[5] = A->mb;
  c[6] = A->mbl;
  c[7] = A->mbu;
  return A->mbe;
}

/* ------------------------------------------------------------------------------------------- */
/* dense GI solver                                                                             */
/* ------------------------------------------------------------------------------------------- */
typedef struct
{
  int n, m, nb;
  const double *C, *bl, *bu, *xl, *xu;
  int ldc;
  double *J, *R, *d, *z, *x, *u, *r; /* J,R: n x n, ld n (GoldfarbIdnaniSolver.cpp:140-141) */
  double * Si; /* CANON C10 (more than ORC_BLOCK_N unknowns): S = R^-1, n x n, ld n, kept beside R; NULL otherwise */
  double f;
  orc_active_set A;
} gi_state;

long orc_gi_work_doubles(int n, int m, int nb)
{
  return 2L * n * n + 3L * n + 2L * (m + nb + 2) + m + 1 + 4L * n; /* + b_act, hCoeffs, alpha, beta (warm start) */
}

/* ConstraintNormal::dot (ConstraintNormal.h:105-123) with the step direction z (CANON C5: a C4 tree) */
static double np_dot(const gi_state * S, int p, int status, const double * v)
{
  switch(status)
  {
    case ORC_EQUALITY:
    case ORC_LOWER: return zdot(S->n, v, S->C + (long)p * S->ldc);
    case ORC_UPPER: return -zdot(S->n, v, S->C + (long)p * S->ldc);
    case ORC_FIXED:
    case ORC_LOWER_BOUND: return v[p - S->m];
    case ORC_UPPER_BOUND: return -v[p - S->m];
    default: return 0.0;
  }
}

/* computeStep_ (GoldfarbIdnaniSolver.cpp:136-148) with ConstraintNormal::preMultiplyByMt (ConstraintNormal.h:81-102) */
static void compute_step(gi_state * S, int p, int status)
{
  const int n = S->n, q = S->A.nact;
  double *d = S->d, *z = S->z, *r = S->r;
  const double * J = S->J;
  switch(status)
  {
    case ORC_EQUALITY:
    case ORC_LOWER:
      multi_dot(n, J, n, n, S->C + (long)p * S->ldc, d);
      break;
    case ORC_UPPER:
      multi_dot(n, J, n, n, S->C + (long)p * S->ldc, d);
      for(int j = 0; j < n; ++j) d[j] = -d[j];
      break;
    case ORC_FIXED:
    case ORC_LOWER_BOUND:
      for(int j = 0; j < n; ++j) d[j] = J[(p - S->m) + (long)j * n];
      break;
    case ORC_UPPER_BOUND:
      for(int j = 0; j < n; ++j) d[j] = -J[(p - S->m) + (long)j * n];
      break;
    default: break;
  }
  /* z = J2 d2 (:145) */
  if(n > ORC_BLOCK_N && !(g_lit & ORC_LIT_CHAIN_GEMV))
  {
    /* CANON C8: per row the columns in segments of 64 by absolute index (a segment's chain from max(64 s, q)), then the tree */
    const int S = (n + 63) >> 6;
    int P2 = 1;
    while(P2 < S) P2 <<= 1;
    double * P = (double *)malloc(sizeof(double) * (size_t)P2 * n);
    for(int s = 0; s < P2; ++s)
    {
      double * ps = P + (long)s * n;
      for(int i = 0; i < n; ++i) ps[i] = 0.0;
      if(s >= S) continue;
      const int j0 = (64 * s > q) ? 64 * s : q, j1 = (64 * s + 64 < n) ? 64 * s + 64 : n;
      for(int j = j0; j < j1; ++j)
      {
        const double * cj = J + (long)j * n;
        const double dj = d[j];
        for(int i = 0; i < n; ++i) ps[i] = FMA(cj[i], dj, ps[i]);
      }
    }
    for(int h = 1; h < P2; h <<= 1)
      for(int s = 0; s < P2; s += 2 * h)
      {
        double *pa = P + (long)s * n, *pb = P + (long)(s + h) * n;
        for(int i = 0; i < n; ++i) pa[i] = pa[i] + pb[i];
      }
    for(int i = 0; i < n; ++i) z[i] = P[i];
    free(P);
  }
  else
  {
  for(int i = 0; i < n; ++i) z[i] = 0.0;
  for(int j = q; j < n; ++j) /* per element: the same chain over increasing j; loop order only helps the compiler */
  {
    const double * cj = J + (long)j * n;
    const double dj = d[j];
    for(int i = 0; i < n; ++i) z[i] = FMA(cj[i], dj, z[i]);
  }
  }
  if(S->Si)
  {
    /* CANON C10: r = S d1 with S = R^-1 kept explicitly.  Row i is a blocked product in the manner of C8 over its terms k = i .. q-1:
     * the columns in segments of 64 by absolute index, a segment's fma chain ascending from max(64 s, i) and from +0, the segment sums
     * combined by the adjacent-first pairwise tree over 64 slots (empty slots +0).  The substitution below is q dependent steps; this
     * is q (q + 64) / 128 independent chains of at most 64 terms. */
    const int nseg = (q + 63) >> 6;
    for(int i = 0; i < q; ++i)
    {
      double p[64];
      const double * row = S->Si + i;
      for(int sg = 0; sg < 64; ++sg)
      {
        double acc = 0.0;
        if(sg < nseg)
        {
          const int k0 = (64 * sg > i) ? 64 * sg : i, k1 = (64 * sg + 64 < q) ? 64 * sg + 64 : q;
          for(int k = k0; k < k1; ++k) acc = FMA(row[(long)k * n], d[k], acc);
        }
        p[sg] = acc;
      }
      for(int h = 1; h < 64; h <<= 1)
        for(int t = 0; t < 64; t += 2 * h) p[t] = p[t] + p[t + h];
      r[i] = p[0];
    }
    return;
  }
  /* r = R^-1 d1 (:146), upper triangular back substitution */
  for(int k = 0; k < q; ++k) r[k] = d[k];
  for(int k = q - 1; k >= 0; --k)
  {
    double v = r[k] / S->R[k + (long)k * n];
    r[k] = v;
    for(int i = 0; i < k; ++i) r[i] = FMA(-S->R[i + (long)k * n], v, r[i]);
  }
}

/* CANON C7 -- the coefficients of an ADDITION's Givens sweep (GoldfarbIdnaniSolver.cpp:225-232: for i = n-2 .. q
 * makeGivens(d_i, d_{i+1}) with the radius left in d_i).  The radius r_i = hypot(d_i, r_{i+1}) is the norm of d[i .. n-1], and with
 * it c_i = d_i / r_i, s_i = -r_{i+1} / r_i (makeGivens' result, JacobiRotation.h, for a non-negative second argument).  The
 * recurrence is n - q dependent makeGivens evaluations -- the longest serial chain of a pass on a GPU -- whereas the norms are a
 * suffix sum: C7 fixes the radii as the square roots of a suffix sum of the squares taken in a fixed, parallel order, and every
 * coefficient as one division by its radius.
 *   j = n-1-k counts from the END of d;  e_j = d_k * d_k;  blocks of 64 consecutive j;  inside a block the Hillis-Steele
 *   inclusive scan  x_j += x_{j-off}  (off = 1, 2, 4, .. 32; all j of the block with j mod 64 >= off at once, from the values of
 *   the step before);  S_j = x_j + carry_b,  carry_0 = 0,  carry_{b+1} = S_{64 b + 63};
 *   rho_0 = d_{n-1} (signed, as the first makeGivens sees it),  rho_j = sqrt(S_j);
 *   rotation i = n-1-j (j >= 1):  rho_j == 0 ? (c, s) = (1, 0)  :  c = d_i / rho_j,  s = (-rho_{j-1}) / rho_j;   r_q = rho_{n-1-q}.
 * Squares leave the double range where makeGivens (which only forms ratios) does not: when an entry is beyond 2^500 in magnitude,
 * or a non-zero one below 2^-500, or anything is not finite, the serial recurrence is used instead -- the decision is a pure
 * function of d, so the kernels take the same branch.  Returns 1 for the tree, 0 for the recurrence. */
int orc_givens_chain(int n, int q, const double * d, double * c, double * s, double * rfin)
{
  const int m = n - q; /* entries d[q .. n-1]; rotations i = n-2 .. q */
  if(m <= 0) return 1;
  if(m == 1)
  {
    *rfin = d[n - 1];
    return 1;
  }
  double amax = 0.0, amin = INFINITY;
  int finite = 1;
  for(int k = q; k < n; ++k)
  {
    const double a = fabs(d[k]);
    if(!(a <= DBL_MAX)) finite = 0;
    if(a > amax) amax = a;
    if(a != 0.0 && a < amin) amin = a;
  }
  if(!finite || amax > 0x1p500 || amin < 0x1p-500 || (g_lit & ORC_LIT_SERIAL_GIVENS))
  {
    double r = d[n - 1];
    for(int i = n - 2; i >= q; --i) orc_givens(d[i], r, &c[i], &s[i], &r);
    *rfin = r;
    return 0;
  }
  double x_small[256];
  double * x = (m <= 256) ? x_small : (double *)malloc(sizeof(double) * (size_t)m);
  for(int j = 0; j < m; ++j) x[j] = d[n - 1 - j] * d[n - 1 - j];
  double carry = 0.0;
  for(int b0 = 0; b0 < m; b0 += 64)
  {
    const int cnt = (m - b0 < 64) ? m - b0 : 64;
    double * xb = x + b0;
    for(int off = 1; off < 64; off <<= 1)
      for(int j = cnt - 1; j >= off; --j) xb[j] = xb[j] + xb[j - off]; /* descending j: xb[j - off] is still the old value */
    for(int j = 0; j < cnt; ++j) xb[j] = xb[j] + carry;
    if(cnt == 64) carry = xb[63];
  }
  double rho_prev = d[n - 1]; /* rho_0 */
  for(int j = 1; j < m; ++j)
  {
    const int i = n - 1 - j;
    const double rho = sqrt(x[j]);
    if(rho == 0.0)
    {
      c[i] = 1.0;
      s[i] = 0.0;
    }
    else
    {
      c[i] = d[i] / rho;
      s[i] = (-rho_prev) / rho;
    }
    rho_prev = rho;
  }
  *rfin = rho_prev;
  if(x != x_small) free(x);
  return 1;
}

/* addConstraint (DualSolver.cpp:231-235) + addConstraint_ (GoldfarbIdnaniSolver.cpp:221-237) */
static void add_constraint(gi_state * S, int p, int status)
{
  const int n = S->n;
  orc_as_activate(&S->A, p, status);
  const int q = S->A.nact; /* already counts the new constraint */
  double *d = S->d, *J = S->J;
  double cs_small[2 * 256];
  double * cc = (n <= 256) ? cs_small : (double *)malloc(sizeof(double) * 2 * (size_t)n);
  double * ss = cc + n;
  double rfin = d[q - 1];
  orc_givens_chain(n, q - 1, d, cc, ss, &rfin); /* C7: all coefficients from d, then the sweep */
  for(int i = n - 2; i >= q - 1; --i)
  {
    double *ci = J + (long)i * n, *cj = J + (long)(i + 1) * n;
    const double c = cc[i], sn = ss[i];
    for(int rr = 0; rr < n; ++rr) rot(c, sn, &ci[rr], &cj[rr]);
  }
  d[q - 1] = rfin;
  for(int k = 0; k < q; ++k) S->R[k + (long)(q - 1) * n] = d[k];
  if(S->Si)
  {
    /* C10: the bordered inverse of [R d1; 0 rho]: S gets the column [-r / rho; 1 / rho], r = S d1 of this pass (still in S->r) */
    for(int k = 0; k < q - 1; ++k) S->Si[k + (long)(q - 1) * n] = -(S->r[k] / rfin);
    S->Si[(q - 1) + (long)(q - 1) * n] = 1.0 / rfin;
  }
  if(cc != cs_small) free(cc);
}

/* removeConstraint_ (GoldfarbIdnaniSolver.cpp:239-256) preceded by A_.deactivate(l) */
static void remove_constraint_raw(gi_state * S, int l)
{
  const int n = S->n;
  orc_as_deactivate(&S->A, l);
  const int q = S->A.nact; /* already counts the removal */
  double *R = S->R, *J = S->J;
  for(int i = l; i < q; ++i)
  {
    double c, s;
    for(int k = 0; k < i; ++k) R[k + (long)i * n] = R[k + (long)(i + 1) * n];
    orc_givens(R[i + (long)(i + 1) * n], R[(i + 1) + (long)(i + 1) * n], &c, &s, &R[i + (long)i * n]);
    for(int k = i + 2; k <= q; ++k) rot(c, s, &R[i + (long)k * n], &R[(i + 1) + (long)k * n]);
    double *ci = J + (long)i * n, *cj = J + (long)(i + 1) * n;
    for(int rr = 0; rr < n; ++rr) rot(c, s, &ci[rr], &cj[rr]);
    if(S->Si)
    {
      /* C10: S <- S G (the same rotation of columns i, i + 1 as J's: (G^T R)^-1 = S G); rows below i + 1 hold structural zeros */
      double *si = S->Si + (long)i * n, *sj = S->Si + (long)(i + 1) * n;
      for(int rr = 0; rr <= i + 1; ++rr)
      {
        if(rr == l) continue; /* (the row that leaves) */
        double xv = (rr <= i) ? si[rr] : 0.0, yv = sj[rr]; /* (row i + 1 meets its first rotation with a structural zero in column i) */
        rot(c, s, &xv, &yv);
        si[rr] = xv;
        sj[rr] = yv;
      }
    }
  }
  if(S->Si)
  {
    /* row l and the last column leave: rows l + 1 .. q move up by one (q: the new count; S had q + 1 rows and columns) */
    for(int j = l; j < q; ++j)
      for(int rr = l; rr <= j; ++rr) S->Si[rr + (long)j * n] = S->Si[(rr + 1) + (long)j * n];
  }
}

/* removeConstraint (DualSolver.cpp:237-244) + removeConstraint_ */
static void remove_constraint(gi_state * S, int l)
{
  const int q = S->A.nact;
  for(int k = l; k < q; ++k) S->u[k] = S->u[k + 1]; /* u.segment(l,q-l) = u.tail(q-l), u has q+1 entries */
  remove_constraint_raw(S, l);
}

/* addInitialConstraint (GoldfarbIdnaniSolver.cpp:295-338) */
static void add_initial_constraint(gi_state * S, int p, int status)
{
  const int n = S->n, q = S->A.nact;
  double *x = S->x, *z = S->z, *u = S->u, *r = S->r;
  u[q] = 0.0;
  compute_step(S, p, status);
  double t = 0.0;
  if(sqrt(zdot(n, z, z)) > 1e-14)
  {
    if(status == ORC_EQUALITY)
    {
      const double * cp = S->C + (long)p * S->ldc;
      t = (S->bl[p] - orc_dot_blk(n, cp, x, 0)) / zdot(n, cp, z);
    }
    else
    {
      int pb = p - S->m;
      t = (S->xl[pb] - x[pb]) / z[pb];
    }
  }
  for(int i = 0; i < n; ++i) x[i] = FMA(t, z[i], x[i]);
  S->f = FMA(t * np_dot(S, p, status, z), 0.5 * t + u[q], S->f);
  for(int k = 0; k < q; ++k) u[k] = FMA(-t, r[k], u[k]);
  u[q] += t;
  add_constraint(S, p, status);
}

/* ------------------------------------------------------------------------------------------- */
/* warm start: experimental::GoldfarbIdnaniSolver (src/experimental/GoldfarbIdnaniSolver.cpp)   */
/* ------------------------------------------------------------------------------------------- */

/* processInitialActiveSet (experimental/GoldfarbIdnaniSolver.cpp:309-386): bounds first, then general constraints.
 * Data equalities win over the guess; guesses on infinite bounds and FIXED guesses on non-fixed variables are
 * ignored.  Guess entries the reference rules out with assert() (a bound status on a general constraint, a general
 * status on a bound, values outside the enum) are ignored too.  as == NULL: no guess (the cold active set). */
static int process_initial_active_set(gi_state * S, const int * as, double big_bnd)
{
  const int n = S->n, m = S->m, nb = S->nb;
  orc_active_set * A = &S->A;
  orc_as_reset(A);
  for(int i = 0; i < nb; ++i)
  {
    const int bi = m + i;
    if(S->xl[i] == S->xu[i])
      orc_as_activate(A, bi, ORC_FIXED);
    else if(as && as[bi] != ORC_INACTIVE)
    {
      const int s = as[bi];
      if(s == ORC_LOWER_BOUND && !(S->xl[i] < -big_bnd)) orc_as_activate(A, bi, s);
      if(s == ORC_UPPER_BOUND && !(S->xu[i] > +big_bnd)) orc_as_activate(A, bi, s);
    }
  }
  for(int i = 0; i < m; ++i)
  {
    if(S->bl[i] == S->bu[i])
      orc_as_activate(A, i, ORC_EQUALITY);
    else if(as && as[i] != ORC_INACTIVE)
    {
      const int s = as[i];
      if(s == ORC_LOWER && !(S->bl[i] < -big_bnd)) orc_as_activate(A, i, s);
      if(s == ORC_UPPER && !(S->bu[i] > +big_bnd)) orc_as_activate(A, i, s);
      if(s == ORC_EQUALITY) orc_as_activate(A, i, s); /* :357-371 only filters FIXED: an EQUALITY guess is taken as is */
    }
  }
  if(A->nact > n)
  {
    if(A->me + A->mbe > n) return ORC_OVERCONSTRAINED_PROBLEM;
    /* work backward, deactivating inequality constraints until n are left (:373-383) */
    int i = A->nact;
    while(A->nact > n)
    {
      --i;
      for(;;)
      {
        int st = A->status[A->active[i]];
        if(st == ORC_EQUALITY || st == ORC_FIXED) --i; else break;
      }
      orc_as_deactivate(A, i);
    }
  }
  return ORC_SUCCESS;
}

/* initializeComputationData (experimental/GoldfarbIdnaniSolver.cpp:388-458).
 * N -> S->R (n x q), b_act; J = L^-T; B = L^-1 N; B = Q R (Householder, unblocked: Eigen's blocked variant with
 * block size 48 reduces to it for q <= 48; for larger q the CANON order below is still the definition); J = J Q.
 * CANON for the Householder steps (Eigen/src/Householder/Householder.h makeHouseholder / applyHouseholderOnTheLeft /
 * applyHouseholderOnTheRight): every sum an ascending fma chain; a*b+c one fma. */
static void initialize_computation_data(gi_state * S, const double * L, int ldl, double * b_act, double * hco)
{
  const int n = S->n, m = S->m, q = S->A.nact;
  double *N = S->R, *J = S->J;
  for(int j = 0; j < q; ++j)
  {
    const int idx = S->A.active[j];
    double * col = N + (long)j * n;
    switch(S->A.status[idx])
    {
      case ORC_LOWER:
      case ORC_EQUALITY:
        for(int i = 0; i < n; ++i) col[i] = S->C[i + (long)idx * S->ldc];
        b_act[j] = S->bl[idx];
        break;
      case ORC_UPPER:
        for(int i = 0; i < n; ++i) col[i] = -S->C[i + (long)idx * S->ldc];
        b_act[j] = -S->bu[idx];
        break;
      case ORC_LOWER_BOUND:
      case ORC_FIXED:
        for(int i = 0; i < n; ++i) col[i] = 0.0;
        col[idx - m] = 1.0;
        b_act[j] = S->xl[idx - m];
        break;
      case ORC_UPPER_BOUND:
        for(int i = 0; i < n; ++i) col[i] = 0.0;
        col[idx - m] = -1.0;
        b_act[j] = -S->xu[idx - m];
        break;
      default: break;
    }
  }
  orc_inv_lt(n, L, ldl, J, n); /* :421-423 */
  /* B = L^-1 N (:426), forward substitution; per element: acc = N_ij; acc = fma(-L_ik, B_kj, acc), k ascending; / L_ii */
  for(int j = 0; j < q; ++j)
  {
    double * col = N + (long)j * n;
    for(int i = 0; i < n; ++i)
    {
      double acc = col[i];
      for(int k = 0; k < i; ++k) acc = FMA(-L[i + (long)k * ldl], col[k], acc);
      col[i] = acc / L[i + (long)i * ldl];
    }
  }
  /* QR in place (:432-440) */
  for(int k = 0; k < q; ++k)
  {
    double * v = N + (long)k * n; /* essential part: v[k+1 .. n-1] */
    double tail = 0.0;
    for(int i = k + 1; i < n; ++i) tail = FMA(v[i], v[i], tail);
    const double c0 = v[k];
    double tau, beta;
    if(tail <= 2.2250738585072014e-308) /* numeric_limits<double>::min() */
    {
      tau = 0.0;
      beta = c0;
      for(int i = k + 1; i < n; ++i) v[i] = 0.0;
    }
    else
    {
      beta = sqrt(FMA(c0, c0, tail));
      if(c0 >= 0.0) beta = -beta;
      const double den = c0 - beta;
      for(int i = k + 1; i < n; ++i) v[i] = v[i] / den;
      tau = (beta - c0) / beta;
    }
    v[k] = beta;
    hco[k] = tau;
    if(tau != 0.0 && k + 1 < n)
      for(int j = k + 1; j < q; ++j) /* applyHouseholderOnTheLeft on the trailing columns */
      {
        double * cj = N + (long)j * n;
        double t = 0.0;
        for(int i = k + 1; i < n; ++i) t = FMA(v[i], cj[i], t);
        t = t + cj[k];
        cj[k] = FMA(-tau, t, cj[k]);
        const double st = tau * t;
        for(int i = k + 1; i < n; ++i) cj[i] = FMA(-st, v[i], cj[i]);
      }
  }
  /* J = J * Q,  Q = H_0 H_1 ... H_{q-1}  (:443-444): applyHouseholderOnTheRight on J.rightCols(n - k), k ascending */
  for(int k = 0; k < q; ++k)
  {
    const double * v = N + (long)k * n;
    const double tau = hco[k];
    if(k + 1 == n)
    {
      for(int i = 0; i < n; ++i) J[i + (long)k * n] *= (1.0 - tau); /* single column: *this *= 1 - tau */
      continue;
    }
    if(tau == 0.0) continue;
    for(int i = 0; i < n; ++i)
    {
      double t = 0.0;
      for(int j = k + 1; j < n; ++j) t = FMA(J[i + (long)j * n], v[j], t);
      t = t + J[i + (long)k * n];
      J[i + (long)k * n] = FMA(-tau, t, J[i + (long)k * n]);
      const double st = tau * t;
      for(int j = k + 1; j < n; ++j) J[i + (long)j * n] = FMA(-st, v[j], J[i + (long)j * n]);
    }
  }
}

/* initializePrimalDualPoints (experimental/GoldfarbIdnaniSolver.cpp:462-487) */
static void initialize_primal_dual(gi_state * S, const double * a, const double * b_act, double * alpha, double * beta)
{
  const int n = S->n, q = S->A.nact;
  const double *J = S->J, *R = S->R;
  double *x = S->x, *u = S->u;
  multi_dot(n, J, n, n, a, alpha); /* alpha = J^T a */
  for(int k = 0; k < q; ++k) /* beta = R^-T b_act */
  {
    double acc = b_act[k];
    for(int i = 0; i < k; ++i) acc = FMA(-R[i + (long)k * n], beta[i], acc);
    beta[k] = acc / R[k + (long)k * n];
  }
  for(int i = 0; i < n; ++i) /* x = J1 beta - J2 alpha2 */
  {
    double acc = 0.0;
    for(int k = 0; k < q; ++k) acc = FMA(J[i + (long)k * n], beta[k], acc);
    for(int k = q; k < n; ++k) acc = FMA(-J[i + (long)k * n], alpha[k], acc);
    x[i] = acc;
  }
  for(int k = 0; k < q; ++k) u[k] = alpha[k] + beta[k]; /* u = R^-1 (alpha1 + beta) */
  for(int k = q - 1; k >= 0; --k)
  {
    double v = u[k] / R[k + (long)k * n];
    u[k] = v;
    for(int i = 0; i < k; ++i) u[i] = FMA(-R[i + (long)k * n], v, u[i]);
  }
  double f1 = 0.0, f2 = 0.0; /* f = beta.(0.5 beta + alpha1) - 0.5 |alpha2|^2 */
  for(int k = 0; k < q; ++k) f1 = FMA(beta[k], FMA(0.5, beta[k], alpha[k]), f1);
  for(int k = q; k < n; ++k) f2 = FMA(alpha[k], alpha[k], f2);
  S->f = FMA(-0.5, f2, f1);
}

static int gi_solve_impl(int warm,
                         const int * as_in,
                         int n,
                         int m,
                         int nb,
                         double * G,
                 int ldg,
                 const double * a,
                 const double * C,
                 int ldc,
                 const double * bl,
                 const double * bu,
                 const double * xl,
                 const double * xu,
                 int max_iter,
                 double big_bnd,
                 double * xout,
                 double * uout,
                 int * status_out,
                 int * active_out,
                 int * nactive_out,
                 int * iters_out,
                 double * f_out,
                 orc_trace_rec * trace,
                 int trace_cap,
                 double * work)
{
  gi_state S;
  int rc = ORC_UNKNOWN, it = 0;
  double * own = NULL;
  if(!work) work = own = (double *)malloc(sizeof(double) * (size_t)orc_gi_work_doubles(n, m, nb));
  S.n = n; S.m = m; S.nb = nb; S.C = C; S.ldc = ldc; S.bl = bl; S.bu = bu; S.xl = xl; S.xu = xu;
  S.J = work; S.R = S.J + (long)n * n; S.d = S.R + (long)n * n; S.z = S.d + n; S.x = S.z + n;
  S.u = S.x + n; S.r = S.u + (m + nb + 2);
  double * cxs = S.r + (m + nb + 2); /* C^T x, all m columns (active ones are computed and ignored) */
  S.f = 0.0;
  /* CANON C10: beyond ORC_BLOCK_N unknowns S = R^-1 is kept beside R (cold and warm start of the dense solver; the closed-form box
   * start is a batched-size path) */
  S.Si = (n > ORC_BLOCK_N && warm != 2 && !(g_lit & ORC_LIT_R_SOLVE)) ? (double *)calloc((size_t)n * (size_t)n, sizeof(double)) : NULL;
  S.A.ncstr = m; S.A.nbnd = nb;
  S.A.status = (int *)malloc(sizeof(int) * (size_t)(m + nb + 1));
  S.A.active = (int *)malloc(sizeof(int) * (size_t)(m + nb + 1));
  orc_as_reset(&S.A);
  for(int i = 0; i < m + nb + 2; ++i) S.u[i] = S.r[i] = 0.0;
  for(int i = 0; i < n; ++i) S.x[i] = 0.0; /* defined output when init_ fails (the reference leaves x untouched) */
  double *x = S.x, *z = S.z, *u = S.u, *r = S.r;

  if(warm == 1)
  {
    /* ---- experimental init_ (experimental/GoldfarbIdnaniSolver.cpp:65-112) ---- */
    double * b_act = cxs + m + 1;
    double * hco = b_act + n;
    double * alpha = hco + n;
    double * beta = alpha + n;
    rc = process_initial_active_set(&S, as_in, big_bnd);
    if(rc != ORC_SUCCESS) goto done;
    if(orc_llt(n, G, ldg) >= 0) { rc = ORC_NON_POS_HESSIAN; goto done; }
    initialize_computation_data(&S, G, ldg, b_act, hco);
    if(S.Si)
    {
      /* C10: S = R^-1 of the QR factor just made.  Column j solves R s = e_j by column-oriented back substitution: s_j = 1 / R(j,j); once
       * s_k is final every row above adds its term, acc_i = fma(R(i,k), s_k, acc_i) (acc from +0: row i's chain runs over k = j .. i + 1,
       * descending), and s_{k-1} = -acc_{k-1} / R(k-1,k-1). */
      const int q0 = S.A.nact;
      for(int j = 0; j < q0; ++j)
      {
        double * sc = S.Si + (long)j * n;
        for(int i = 0; i < j; ++i) sc[i] = 0.0;
        sc[j] = 1.0 / S.R[j + (long)j * n];
        for(int k = j; k >= 1; --k)
        {
          const double sk = sc[k];
          for(int i = 0; i < k; ++i) sc[i] = FMA(S.R[i + (long)k * n], sk, sc[i]);
          sc[k - 1] = (-sc[k - 1]) / S.R[(k - 1) + (long)(k - 1) * n];
        }
      }
    }
    initialize_primal_dual(&S, a, b_act, alpha, beta);
    for(;;) /* constraints activated with u < 0 are deactivated (:78-109) */
    {
      const int q = S.A.nact;
      double umin = -1e-14;
      int lmin = -1;
      for(int l = 0; l < q; ++l)
      {
        int st = S.A.status[S.A.active[l]];
        if(S.u[l] < umin && st != ORC_FIXED && st != ORC_EQUALITY)
        {
          umin = S.u[l];
          lmin = l;
        }
      }
      if(lmin < 0) break;
      ++it;
      for(int k = lmin; k + 1 < q; ++k) b_act[k] = b_act[k + 1];
      remove_constraint_raw(&S, lmin);
      initialize_primal_dual(&S, a, b_act, alpha, beta);
    }
    S.u[S.A.nact] = 0.0;
    goto main_loop;
  }

  if(warm == 2)
  {
    /* ---- BoxAndSingleConstraintSolver::init_ (experimental/BoxAndSingleConstraintSolver.cpp:46-94): G = I is implicit,
     * `a` holds x0 itself (:38 "a will contain x0, not -x0"); the start point is x0 clamped to the box, every clamped
     * variable's bound is active with multiplier |x0_i - bound|, J is the permutation that puts the clamped coordinates
     * first (free coordinate number k, counted from 0, goes to column n-1-k) and R = diag(+1 lower / -1 upper).
     * f_ += 0.5 * u[q] * u[q] is read as fma(0.5 * u, u, f) (CANON C2). ---- */
    const double * x0 = a;
    int q = 0;
    for(long e = 0; e < (long)n * n; ++e) S.J[e] = 0.0;
    for(int i = 0; i < n; ++i)
    {
      if(x0[i] < xl[i])
      {
        x[i] = xl[i];
        u[q] = x[i] - x0[i];
        S.f = FMA(0.5 * u[q], u[q], S.f);
        for(int k = 0; k < q; ++k) S.R[k + (long)q * n] = 0.0;
        S.R[q + (long)q * n] = 1.0;
        S.J[i + (long)q * n] = 1.0;
        orc_as_activate(&S.A, m + i, ORC_LOWER_BOUND); /* (:72 "i+1": m == 1) */
        ++q;
      }
      else if(x0[i] > xu[i])
      {
        x[i] = xu[i];
        u[q] = x0[i] - x[i];
        S.f = FMA(0.5 * u[q], u[q], S.f);
        for(int k = 0; k < q; ++k) S.R[k + (long)q * n] = 0.0;
        S.R[q + (long)q * n] = -1.0;
        S.J[i + (long)q * n] = 1.0;
        orc_as_activate(&S.A, m + i, ORC_UPPER_BOUND);
        ++q;
      }
      else
      {
        x[i] = x0[i];
        S.J[i + (long)(n - i + q - 1) * n] = 1.0;
      }
    }
    goto main_loop;
  }

  /* ---- init_ (GoldfarbIdnaniSolver.cpp:56-82) ---- */
  if(orc_llt(n, G, ldg) >= 0) { rc = ORC_NON_POS_HESSIAN; goto done; }
  orc_inv_lt(n, G, ldg, S.J, n);
  for(int i = 0; i < n; ++i) x[i] = a[i];
  for(int k = 0; k < n; ++k) /* x = L.solve(a) */
  {
    double v = x[k] / G[k + (long)k * ldg];
    x[k] = v;
    for(int i = k + 1; i < n; ++i) x[i] = FMA(-G[i + (long)k * ldg], v, x[i]);
  }
  for(int k = n - 1; k >= 0; --k) /* L^T.solveInPlace(x) */
  {
    double v = x[k] / G[k + (long)k * ldg];
    x[k] = v;
    for(int i = 0; i < k; ++i) x[i] = FMA(-G[k + (long)i * ldg], v, x[i]);
  }
  for(int i = 0; i < n; ++i) x[i] = -x[i];
  S.f = 0.5 * orc_dot(n, a, x);

  /* initActiveSet (:268-287).  More than n equalities/fixed variables is UB in the reference; reported here. */
  {
    int neq = 0;
    for(int i = 0; i < m; ++i) neq += (bl[i] == bu[i]);
    for(int i = 0; i < nb; ++i) neq += (xl[i] == xu[i]);
    if(neq > n) { rc = ORC_OVERCONSTRAINED_PROBLEM; goto done; }
  }
  for(int i = 0; i < m; ++i)
    if(bl[i] == bu[i]) add_initial_constraint(&S, i, ORC_EQUALITY);
  for(int i = 0; i < nb; ++i)
    if(xl[i] == xu[i]) add_initial_constraint(&S, m + i, ORC_FIXED);

  /* ---- DualSolver::solve loop (DualSolver.cpp:96-167) ---- */
main_loop:
  {
    int skip_step1 = 0, p = -1, pstatus = ORC_INACTIVE;
    rc = ORC_MAX_ITER_REACHED;
    for(; it < max_iter; ++it)
    {
      int q = S.A.nact;
      if(!skip_step1)
      {
        /* selectViolatedConstraint_ (GoldfarbIdnaniSolver.cpp:84-134) */
        double smin = 0.0;
        p = -1;
        pstatus = ORC_INACTIVE;
        multi_dot(n, C, ldc, m, x, cxs);
        for(int i = 0; i < m; ++i)
        {
          if(S.A.status[i] != ORC_INACTIVE) continue;
          double cx = cxs[i]; /* == C.col(i).dot(x) (:97) */
          double sl = cx - bl[i];
          if(sl < smin) { smin = sl; p = i; pstatus = ORC_LOWER; }
          else
          {
            double su = bu[i] - cx;
            if(su < smin) { smin = su; p = i; pstatus = ORC_UPPER; }
          }
        }
        for(int i = 0; i < nb; ++i)
        {
          if(S.A.status[m + i] != ORC_INACTIVE) continue;
          double sl = x[i] - xl[i];
          if(sl < smin) { smin = sl; p = m + i; pstatus = ORC_LOWER_BOUND; }
          else
          {
            double su = xu[i] - x[i];
            if(su < smin) { smin = su; p = m + i; pstatus = ORC_UPPER_BOUND; }
          }
        }
        if(pstatus == ORC_INACTIVE) { rc = ORC_SUCCESS; break; }
        u[q] = 0.0;
      }

      compute_step(&S, p, pstatus);

      /* computeStepLength_ (GoldfarbIdnaniSolver.cpp:150-219) */
      double t1 = big_bnd, t2 = big_bnd;
      int l = 0;
      for(int k = 0; k < q; ++k)
      {
        /* reference quirk: the status is looked up with the POSITION k (:162) */
        int sk = S.A.status[k];
        if(sk != ORC_EQUALITY && sk != ORC_FIXED && r[k] > 0.0)
        {
          double tk = u[k] / r[k];
          if(tk < t1) { t1 = tk; l = k; }
        }
      }
      double cz = 0.0;
      if(sqrt(zdot(n, z, z)) > 1e-14) /* CANON C5 */
      {
        double b, cx;
        if(pstatus == ORC_LOWER || pstatus == ORC_UPPER)
        {
          const double * cp = C + (long)p * ldc;
          b = (pstatus == ORC_LOWER) ? bl[p] : bu[p];
          cx = orc_dot_blk(n, cp, x, 0);  /* C1 / C8: == cxs[p] of the scan when p was selected in this pass */
          cz = zdot(n, cp, z); /* C5 */
        }
        else
        {
          int pb = p - m;
          b = (pstatus == ORC_LOWER_BOUND) ? xl[pb] : xu[pb];
          cx = x[pb];
          cz = z[pb];
        }
        t2 = (b - cx) / cz;
      }
      double t = (t2 < t1) ? t2 : t1; /* std::min(t1, t2) (DualSolver.cpp:127) */

      if(g_vtrace && it < g_vtrace_cap)
      {
        /* the vectors the reference logs per pass (DualSolver.cpp:108: x, u, f_; :128-129: z, r): as they stand behind the step
         * computation, nothing of this pass applied yet -- the layout of jrlqp_dense_set_trace_vectors */
        const int mt1 = m + nb + 1;
        double * rec = g_vtrace + (long)it * (2 * n + 2 * mt1 + 2);
        for(int i = 0; i < n; ++i) rec[i] = S.x[i];
        for(int k = 0; k < mt1; ++k) rec[n + k] = (k <= q) ? S.u[k] : 0.0;
        for(int i = 0; i < n; ++i) rec[n + mt1 + i] = S.z[i];
        for(int k = 0; k < mt1; ++k) rec[2 * n + mt1 + k] = (k < q) ? S.r[k] : 0.0;
        rec[2 * n + 2 * mt1] = S.f;
        rec[2 * n + 2 * mt1 + 1] = (double)q;
      }
      if(trace && it < trace_cap)
      {
        trace[it].p = p; trace[it].status = pstatus; trace[it].l = l; trace[it].t1 = t1; trace[it].t2 = t2;
        trace[it].kind = (t2 >= big_bnd) ? 2 : ((t == t2) ? 0 : 1);
      }

      if(t >= big_bnd) { rc = ORC_INFEASIBLE; break; }

      if(t2 >= big_bnd)
      {
        for(int k = 0; k < q; ++k) u[k] = FMA(-t, r[k], u[k]);
        u[q] += t;
        remove_constraint(&S, l);
        skip_step1 = 1;
      }
      else
      {
        for(int i = 0; i < n; ++i) x[i] = FMA(t, z[i], x[i]);
        S.f = FMA(t * np_dot(&S, p, pstatus, z), 0.5 * t + u[q], S.f);
        for(int k = 0; k < q; ++k) u[k] = FMA(-t, r[k], u[k]);
        u[q] += t;
        if(t == t2)
        {
          add_constraint(&S, p, pstatus);
          skip_step1 = 0;
        }
        else
        {
          remove_constraint(&S, l);
          skip_step1 = 1;
        }
      }
    }
  }

done:
  for(int i = 0; i < n; ++i) xout[i] = x[i];
  if(uout)
  {
    /* DualSolver::multipliers (DualSolver.cpp:38-69) */
    for(int i = 0; i < m + nb; ++i) uout[i] = 0.0;
    for(int k = 0; k < S.A.nact; ++k)
    {
      int i = S.A.active[k];
      int st = S.A.status[i];
      uout[i] = (st == ORC_UPPER || st == ORC_UPPER_BOUND) ? u[k] : -u[k];
    }
  }
  if(status_out) for(int i = 0; i < m + nb; ++i) status_out[i] = S.A.status[i];
  if(active_out) for(int k = 0; k < S.A.nact; ++k) active_out[k] = S.A.active[k];
  if(nactive_out) *nactive_out = S.A.nact;
  if(iters_out) *iters_out = it;
  if(f_out) *f_out = S.f;
  free(S.Si);
  free(S.A.status);
  free(S.A.active);
  free(own);
  return rc;
}

int orc_gi_solve(int n, int m, int nb, double * G, int ldg, const double * a, const double * C, int ldc,
                 const double * bl, const double * bu, const double * xl, const double * xu, int max_iter,
                 double big_bnd, double * xout, double * uout, int * status_out, int * active_out, int * nactive_out,
                 int * iters_out, double * f_out, orc_trace_rec * trace, int trace_cap, double * work)
{
  return gi_solve_impl(0, NULL, n, m, nb, G, ldg, a, C, ldc, bl, bu, xl, xu, max_iter, big_bnd, xout, uout, status_out,
                       active_out, nactive_out, iters_out, f_out, trace, trace_cap, work);
}

int orc_gi_solve_warm(int n, int m, int nb, double * G, int ldg, const double * a, const double * C, int ldc,
                      const double * bl, const double * bu, const double * xl, const double * xu, const int * as_in,
                      int max_iter, double big_bnd, double * xout, double * uout, int * status_out, int * active_out,
                      int * nactive_out, int * iters_out, double * f_out, orc_trace_rec * trace, int trace_cap,
                      double * work)
{
  return gi_solve_impl(1, as_in, n, m, nb, G, ldg, a, C, ldc, bl, bu, xl, xu, max_iter, big_bnd, xout, uout, status_out,
                       active_out, nactive_out, iters_out, f_out, trace, trace_cap, work);
}

/* The cold solve with some of CANON C5 / C6 / C7 switched off for the calling thread (see g_lit above); flags = 0 is orc_gi_solve. */
int orc_gi_solve_literal(int flags, int n, int m, int nb, double * G, int ldg, const double * a, const double * C, int ldc,
                         const double * bl, const double * bu, const double * xl, const double * xu, int max_iter,
                         double big_bnd, double * xout, double * uout, int * status_out, int * active_out, int * nactive_out,
                         int * iters_out, double * f_out, orc_trace_rec * trace, int trace_cap, double * work)
{
  g_lit = flags;
  const int rc = gi_solve_impl(0, NULL, n, m, nb, G, ldg, a, C, ldc, bl, bu, xl, xu, max_iter, big_bnd, xout, uout, status_out,
                               active_out, nactive_out, iters_out, f_out, trace, trace_cap, work);
  g_lit = 0;
  return rc;
}

/* Packed batch through orc_gi_solve_literal; active: B x n ordered active lists (may be NULL), nactive: B. */
void orc_gi_solve_batch_literal(int flags, int B, int n, int m, int nb, double * G, const double * a, const double * C,
                                const double * bl, const double * bu, const double * xl, const double * xu, int max_iter,
                                double big_bnd, double * x, double * u, int * status, int * active, int * nactive, int * term,
                                int * iters, double * f, int nthreads)
{
  if(nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    double * work = (double *)malloc(sizeof(double) * (size_t)orc_gi_work_doubles(n, m, nb));
    int * act = (int *)malloc(sizeof(int) * (size_t)(m + nb + 1));
#pragma omp for schedule(static)
    for(int b = 0; b < B; ++b)
    {
      int na = 0;
      term[b] = orc_gi_solve_literal(flags, n, m, nb, G + (long)b * n * n, n, a + (long)b * n, C + (long)b * n * m, n,
                                     bl + (long)b * m, bu + (long)b * m, nb ? xl + (long)b * nb : NULL,
                                     nb ? xu + (long)b * nb : NULL, max_iter, big_bnd, x + (long)b * n, u + (long)b * (m + nb),
                                     status + (long)b * (m + nb), act, &na, iters + b, f + b, NULL, 0, work);
      if(nactive) nactive[b] = na;
      if(active)
        for(int k = 0; k < n; ++k) active[(long)b * n + k] = (k < na) ? act[k] : -1;
    }
    free(act);
    free(work);
  }
}

/* experimental::BoxAndSingleConstraintSolver::solve(x0, c, bl, xl, xu) (BoxAndSingleConstraintSolver.cpp:11-44):
 *   min 1/2 ||x - x0||^2  s.t.  c.x >= bl,  xl <= x <= xu      (one general constraint, bu = +inf, bounds mandatory)
 * closed-form init_ (:46-94), then the unchanged GoldfarbIdnaniSolver hooks.  u: 1 + n multipliers, status likewise. */
int orc_box_solve(int n, const double * x0, const double * c, double bl, const double * xl, const double * xu, int max_iter,
                  double big_bnd, double * xout, double * uout, int * status_out, int * active_out, int * nactive_out,
                  int * iters_out, double * f_out, orc_trace_rec * trace, int trace_cap)
{
  const double blv[1] = {bl}, buv[1] = {INFINITY};
  return gi_solve_impl(2, NULL, n, 1, n, NULL, n, x0, c, n, blv, buv, xl, xu, max_iter, big_bnd, xout, uout, status_out,
                       active_out, nactive_out, iters_out, f_out, trace, trace_cap, NULL);
}

void orc_gi_solve_batch(int B,
                        int n,
                        int m,
                        int nb,
                        double * G,
                        const double * a,
                        const double * C,
                        const double * bl,
                        const double * bu,
                        const double * xl,
                        const double * xu,
                        int max_iter,
                        double big_bnd,
                        double * x,
                        double * u,
                        int * status,
                        int * term,
                        int * iters,
                        double * f,
                        int nthreads)
{
  if(nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    /* one workspace per thread, as one solver object per thread in the reference's usage (benchmarks/Solvers.cpp:513-518) */
    double * work = (double *)malloc(sizeof(double) * (size_t)orc_gi_work_doubles(n, m, nb));
#pragma omp for schedule(static)
    for(int b = 0; b < B; ++b)
    {
      term[b] = orc_gi_solve(n, m, nb, G + (long)b * n * n, n, a + (long)b * n, C + (long)b * n * m, n, bl + (long)b * m,
                             bu + (long)b * m, nb ? xl + (long)b * nb : NULL, nb ? xu + (long)b * nb : NULL, max_iter,
                             big_bnd, x + (long)b * n, u + (long)b * (m + nb), status + (long)b * (m + nb), NULL, NULL,
                             iters + b, f + b, NULL, 0, work);
    }
    free(work);
  }
}

/* Timing loop of the CPU baseline (bench.py cpu_baseline leg) -- the reference's own benchmark loop
 * (benchmarks/Solvers.cpp:447-453 "G = original G" before every solve, :508-520 one solver object per measurement): every
 * thread owns ONE workspace and ONE n x n buffer; per solve it restores G into that buffer with a memcpy (the reference's
 * `G = pb.G` copy, inside the loop) and solves.  `rounds` passes over the B problems, static partition.  Nothing is allocated
 * or copied outside the threads, outputs land in caller-preallocated arrays (overwritten every round).  Returns the wall
 * seconds of the rounds alone, taken between two barriers INSIDE the parallel region (thread start-up excluded). */
double orc_gi_bench_rounds(int rounds, int B, int n, int m, int nb, const double * G, const double * a, const double * C,
                           const double * bl, const double * bu, const double * xl, const double * xu, int max_iter,
                           double big_bnd, double * x, double * u, int * status, int * term, int * iters, double * f,
                           int nthreads)
{
  double t0 = 0.0, t1 = 0.0;
  if(nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    double * work = (double *)malloc(sizeof(double) * (size_t)orc_gi_work_doubles(n, m, nb));
    double * Gw = (double *)malloc(sizeof(double) * (size_t)n * n);
    memset(Gw, 0, sizeof(double) * (size_t)n * n); /* (pages touched before the clock starts) */
#pragma omp barrier
#pragma omp master
    t0 = omp_get_wtime();
    for(int r = 0; r < rounds; ++r)
    {
#pragma omp for schedule(static) nowait
      for(int b = 0; b < B; ++b)
      {
        memcpy(Gw, G + (long)b * n * n, sizeof(double) * (size_t)n * n);
        term[b] = orc_gi_solve(n, m, nb, Gw, n, a + (long)b * n, C + (long)b * n * m, n, bl + (long)b * m, bu + (long)b * m,
                               nb ? xl + (long)b * nb : NULL, nb ? xu + (long)b * nb : NULL, max_iter, big_bnd,
                               x + (long)b * n, u + (long)b * (m + nb), status + (long)b * (m + nb), NULL, NULL, iters + b,
                               f + b, NULL, 0, work);
      }
    }
#pragma omp barrier
#pragma omp master
    t1 = omp_get_wtime();
    free(Gw);
    free(work);
  }
  return t1 - t0;
}

/* B warm-started solves; as_in: B x (m+nb) activation-status guesses (may be NULL: no guess). */
void orc_gi_solve_batch_warm(int B, int n, int m, int nb, double * G, const double * a, const double * C,
                             const double * bl, const double * bu, const double * xl, const double * xu,
                             const int * as_in, int max_iter, double big_bnd, double * x, double * u, int * status,
                             int * term, int * iters, double * f, int nthreads)
{
  if(nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    double * work = (double *)malloc(sizeof(double) * (size_t)orc_gi_work_doubles(n, m, nb));
    int * as = (int *)malloc(sizeof(int) * (size_t)(m + nb + 1));
#pragma omp for schedule(static)
    for(int b = 0; b < B; ++b)
    {
      if(as_in) memcpy(as, as_in + (long)b * (m + nb), sizeof(int) * (size_t)(m + nb)); /* status may alias as_in */
      term[b] = orc_gi_solve_warm(n, m, nb, G + (long)b * n * n, n, a + (long)b * n, C + (long)b * n * m, n,
                                  bl + (long)b * m, bu + (long)b * m, nb ? xl + (long)b * nb : NULL,
                                  nb ? xu + (long)b * nb : NULL, as_in ? as : NULL, max_iter, big_bnd, x + (long)b * n,
                                  u + (long)b * (m + nb), status + (long)b * (m + nb), NULL, NULL, iters + b, f + b,
                                  NULL, 0, work);
    }
    free(as);
    free(work);
  }
}
