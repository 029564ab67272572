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
 */
#include "jrlqp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FMA(a, b, c) __builtin_fma((a), (b), (c))

/* ------------------------------------------------------------------------------------------- */
/* canonical primitives                                                                        */
/* ------------------------------------------------------------------------------------------- */

/* C1 dot product (Eigen dot/GEMV call sites: GoldfarbIdnaniSolver.cpp:73,97,183-190, ConstraintNormal.h:87-113) */
double orc_dot(int n, const double * a, const double * b)
{
  double acc = 0.0;
  for(int k = 0; k < n; ++k) acc = FMA(a[k], b[k], acc);
  return acc;
}

/* out[j] = orc_dot(n, M + j*ldm, v) for j < ncols: same per-output chain as orc_dot (bit-identical), eight
 * independent chains in flight so that the CPU baseline is not bound by fma latency. */
static void multi_dot(int n, const double * M, long ldm, int ncols, const double * v, double * out)
{
  int j = 0;
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
  for(int j = 0; j < n; ++j)
  {
    double acc = A[j + (long)j * lda];
    for(int k = 0; k < j; ++k) acc = FMA(-A[j + (long)k * lda], A[j + (long)k * lda], acc);
    if(acc <= 0.0) return j;
    double ljj = sqrt(acc);
    A[j + (long)j * lda] = ljj;
    for(int i = j + 1; i < n; ++i)
    {
      double s = A[i + (long)j * lda];
      for(int k = 0; k < j; ++k) s = FMA(-A[i + (long)k * lda], A[j + (long)k * lda], s);
      A[i + (long)j * lda] = s / ljj;
    }
  }
  return -1;
}

/* J.setIdentity(); L.transpose().solveInPlace(J)  (GoldfarbIdnaniSolver.cpp:64-66).
 * Backward substitution per column; entries below the diagonal are exact zeros. */
void orc_inv_lt(int n, const double * L, int ldl, double * J, int ldj)
{
  for(int c = 0; c < n; ++c)
  {
    double * col = J + (long)c * ldj;
    for(int i = 0; i < n; ++i) col[i] = (i == c) ? 1.0 : 0.0;
    for(int k = c; k >= 0; --k)
    {
      double v = col[k] / L[k + (long)k * ldl];
      col[k] = v;
      for(int i = 0; i < k; ++i) col[i] = FMA(-L[k + (long)i * ldl], v, col[i]);
    }
  }
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
  *x = FMA(c, xv, -(s * yv));
  *y = FMA(s, xv, c * yv);
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
  double f;
  orc_active_set A;
} gi_state;

long orc_gi_work_doubles(int n, int m, int nb)
{
  return 2L * n * n + 3L * n + 2L * (m + nb + 2) + m + 1;
}

/* ConstraintNormal::dot (ConstraintNormal.h:105-123) */
static double np_dot(const gi_state * S, int p, int status, const double * v)
{
  switch(status)
  {
    case ORC_EQUALITY:
    case ORC_LOWER: return orc_dot(S->n, v, S->C + (long)p * S->ldc);
    case ORC_UPPER: return -orc_dot(S->n, v, S->C + (long)p * S->ldc);
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
  for(int i = 0; i < n; ++i) z[i] = 0.0;
  for(int j = q; j < n; ++j) /* per element: the same chain over increasing j; loop order only helps the compiler */
  {
    const double * cj = J + (long)j * n;
    const double dj = d[j];
    for(int i = 0; i < n; ++i) z[i] = FMA(cj[i], dj, z[i]);
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

/* addConstraint (DualSolver.cpp:231-235) + addConstraint_ (GoldfarbIdnaniSolver.cpp:221-237) */
static void add_constraint(gi_state * S, int p, int status)
{
  const int n = S->n;
  orc_as_activate(&S->A, p, status);
  const int q = S->A.nact; /* already counts the new constraint */
  double *d = S->d, *J = S->J;
  for(int i = n - 2; i >= q - 1; --i)
  {
    double c, s;
    orc_givens(d[i], d[i + 1], &c, &s, &d[i]);
    double *ci = J + (long)i * n, *cj = J + (long)(i + 1) * n;
    for(int rr = 0; rr < n; ++rr) rot(c, s, &ci[rr], &cj[rr]);
  }
  for(int k = 0; k < q; ++k) S->R[k + (long)(q - 1) * n] = d[k];
}

/* removeConstraint (DualSolver.cpp:237-244) + removeConstraint_ (GoldfarbIdnaniSolver.cpp:239-256) */
static void remove_constraint(gi_state * S, int l)
{
  const int n = S->n;
  int q = S->A.nact;
  for(int k = l; k < q; ++k) S->u[k] = S->u[k + 1]; /* u.segment(l,q-l) = u.tail(q-l), u has q+1 entries */
  orc_as_deactivate(&S->A, l);
  q = S->A.nact; /* already counts the removal */
  double *R = S->R, *J = S->J;
  for(int i = l; i < q; ++i)
  {
    double c, s;
    for(int k = 0; k < i; ++k) R[k + (long)i * n] = R[k + (long)(i + 1) * n];
    orc_givens(R[i + (long)(i + 1) * n], R[(i + 1) + (long)(i + 1) * n], &c, &s, &R[i + (long)i * n]);
    for(int k = i + 2; k <= q; ++k) rot(c, s, &R[i + (long)k * n], &R[(i + 1) + (long)k * n]);
    double *ci = J + (long)i * n, *cj = J + (long)(i + 1) * n;
    for(int rr = 0; rr < n; ++rr) rot(c, s, &ci[rr], &cj[rr]);
  }
}

/* addInitialConstraint (GoldfarbIdnaniSolver.cpp:295-338) */
static void add_initial_constraint(gi_state * S, int p, int status)
{
  const int n = S->n, q = S->A.nact;
  double *x = S->x, *z = S->z, *u = S->u, *r = S->r;
  u[q] = 0.0;
  compute_step(S, p, status);
  double t = 0.0;
  if(sqrt(orc_dot(n, z, z)) > 1e-14)
  {
    if(status == ORC_EQUALITY)
    {
      const double * cp = S->C + (long)p * S->ldc;
      t = (S->bl[p] - orc_dot(n, cp, x)) / orc_dot(n, cp, z);
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

int orc_gi_solve(int n,
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
  S.A.ncstr = m; S.A.nbnd = nb;
  S.A.status = (int *)malloc(sizeof(int) * (size_t)(m + nb + 1));
  S.A.active = (int *)malloc(sizeof(int) * (size_t)(m + nb + 1));
  orc_as_reset(&S.A);
  for(int i = 0; i < m + nb + 2; ++i) S.u[i] = S.r[i] = 0.0;
  for(int i = 0; i < n; ++i) S.x[i] = 0.0; /* defined output when init_ fails (the reference leaves x untouched) */
  double *x = S.x, *z = S.z, *u = S.u, *r = S.r;

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
      if(sqrt(orc_dot(n, z, z)) > 1e-14)
      {
        double b, cx;
        if(pstatus == ORC_LOWER || pstatus == ORC_UPPER)
        {
          const double * cp = C + (long)p * ldc;
          b = (pstatus == ORC_LOWER) ? bl[p] : bu[p];
          cx = orc_dot(n, cp, x);
          cz = orc_dot(n, cp, z);
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
  free(S.A.status);
  free(S.A.active);
  free(own);
  return rc;
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
