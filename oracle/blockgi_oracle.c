/*
 * blockgi_oracle.c -- CPU oracle for the STRUCTURED Goldfarb-Idnani inner loop (TEST INFRASTRUCTURE ONLY).
 *
 * Restates, in plain C, the reference's experimental::BlockGISolver together with the structured/ classes and the
 * orthonormal-sequence bookkeeping it is built on (jrl-umi3218/jrl-qp):
 *   src/experimental/BlockGISolver.cpp:17-288, :290-374, :446-483      solver hooks, init
 *   src/structured/StructuredG.cpp:22-113                              dispatch to the block decompositions
 *   src/structured/StructuredC.cpp:59-77                               col(i) as a single-segment vector, C^T x
 *   src/structured/StructuredJ.cpp:32-57                               d = Q^T L^-1 n+,  z = L^-T Q [0; d2]
 *   src/structured/StructuredQR.cpp:64-102                             R solve, Householder add, Givens remove
 *   src/internal/OrthonormalSequence.cpp:50-126, :189-207              sequence application
 *   src/DualSolver.cpp:91-168, :231-244, :38-69                        generic loop, multipliers
 *
 * Rounding contract: the CANON rules of jrlqp_oracle.c (C1-C3) for everything that works on one block (the block
 * factorisations / substitutions of structured_oracle.c, C^T x, the R solve, Givens rotations), plus
 *   C4  reductions over a whole vector of unknowns (Householder u.v, squared norms, a.x) are summed as 1024 strided
 *       partial fma chains -- element e of the vector (ABSOLUTE row index, 0-based) goes to chain e mod 1024, ascending
 *       e -- combined by an adjacent-first pairwise tree (chain 2i with 2i+1, then pairs of pairs, ...).  Eigen sums
 *       such reductions in SIMD-width partial sums too (width unspecified); this order maps one-to-one on a workgroup
 *       of 64..1024 lanes and is what the HIP kernel reproduces bit for bit (orc_dot_tree).
 */
#include "jrlqp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FMA(a, b, c) __builtin_fma((a), (b), (c))

/* ------------------------------------------------------------------------------------------- */
/* C4 reduction                                                                                */
/* ------------------------------------------------------------------------------------------- */
static unsigned short g_bitrev[1024];
static int g_bitrev_ready = 0;

static void bitrev_init(void)
{
  for(int v = 0; v < 1024; ++v)
  {
    int r = 0;
    for(int b = 0; b < 10; ++b)
      if(v & (1 << b)) r |= 1 << (9 - b);
    g_bitrev[v] = (unsigned short)r;
  }
  g_bitrev_ready = 1;
}

/* sum_i a[i] b[i], element i having absolute index phase + i.  The chains are kept in bit-reversed order so that
 * every level of the adjacent-first tree is one contiguous vector add: chains 2i and 2i+1 differ in bit 0, i.e. in
 * the TOP bit of their storage slot. */
double orc_dot_tree(int n, const double * a, const double * b, int phase)
{
  double p[1024];
  if(phase == 0 && n <= 1024)
  {
    /* every element has a chain of its own (one product): the tree over the first 2^k >= n slots, in natural order.  The slots
     * beyond are +0.0 and a partial sum is never -0.0 (fma(a, b, +0.0) is not), so leaving them out changes no bit. */
    int size = 1;
    while(size < n) size <<= 1;
    for(int i = 0; i < n; ++i) p[i] = FMA(a[i], b[i], 0.0);
    for(int i = n; i < size; ++i) p[i] = 0.0;
    for(int h = 1; h < size; h <<= 1)
      for(int i = 0; i < size; i += 2 * h) p[i] = p[i] + p[i + h];
    return n > 0 ? p[0] : 0.0;
  }
  if(!g_bitrev_ready) bitrev_init();
  memset(p, 0, sizeof p);
  for(int i = 0; i < n; ++i)
  {
    const int s = g_bitrev[(phase + i) & 1023];
    p[s] = FMA(a[i], b[i], p[s]);
  }
  for(int h = 512; h >= 1; h >>= 1)
    for(int i = 0; i < h; ++i) p[i] = p[i] + p[i + h];
  return p[0];
}

/* ------------------------------------------------------------------------------------------- */
/* state                                                                                       */
/* ------------------------------------------------------------------------------------------- */
typedef struct
{
  int kind;  /* 0 Householder (one reflector), 1 Givens sequence  (OSeqType, OrthonormalSequence.h:17-22) */
  int start; /* EmbeddedSeq::start */
  int n;     /* size of the embedded orthonormal matrix */
  int size;  /* number of elementary transformations */
  long off;  /* pool offset: Householder: n doubles (E(0) = 1, essential part below); Givens: c[size] then s[size] */
  double tau;
} oseq;

typedef struct
{
  /* StructuredG */
  int type; /* 0 TriBlockDiagonal, 1 BlockArrowUp, 2 BlockArrowDown  (StructuredG.h Type) */
  int nblk;
  const int * bs;
  double ** diag;
  const int * ldd;
  double ** off;
  const int * ldo;
  int n;
  /* StructuredC (block diagonal) */
  int ncb;
  const int *cvar, *ccstr, *cld;
  const double ** cblk;
  int m;
  int *to_block, *cum_var, *cum_cstr; /* StructuredC.cpp:9-23 */
  /* problem vectors */
  int nb;
  const double *bl, *bu, *xl, *xu;
  /* work */
  double *d, *z, *x, *u, *r, *cx;
  double * Rp; /* upper triangular R, packed by columns: R(i,j) at i + j(j+1)/2 */
  int q;       /* StructuredQR::q_ */
  oseq * seq;
  int nseq, seq_cap;
  double * pool;
  long pool_len, pool_cap;
  double f;
  orc_active_set * A;
} bgi;

#define RP(S, i, j) (S)->Rp[(i) + (long)(j) * ((j) + 1) / 2]

static double * pool_take(bgi * S, long cnt)
{
  if(S->pool_len + cnt > S->pool_cap)
  {
    long cap = S->pool_cap ? S->pool_cap : 1024;
    while(cap < S->pool_len + cnt) cap *= 2;
    S->pool = (double *)realloc(S->pool, sizeof(double) * (size_t)cap);
    S->pool_cap = cap;
  }
  double * p = S->pool + S->pool_len;
  S->pool_len += cnt;
  return p;
}

static oseq * seq_push(bgi * S)
{
  if(S->nseq == S->seq_cap)
  {
    S->seq_cap = S->seq_cap ? 2 * S->seq_cap : 64;
    S->seq = (oseq *)realloc(S->seq, sizeof(oseq) * (size_t)S->seq_cap);
  }
  return &S->seq[S->nseq++];
}

/* StructuredG::solveL (StructuredG.cpp:67-113): v <- L^-1 v, rows outside [start, end) of v are zero */
static void sg_solve_l(const bgi * S, double * v, int start, int end)
{
  if(S->type == 0)
    orc_tri_block_diag_lsolve(S->nblk, S->bs, S->diag, S->ldd, S->off, S->ldo, v, S->n, 1, start);
  else
    orc_block_arrow_lsolve(S->nblk, S->bs, S->diag, S->ldd, S->off, S->ldo, S->type == 1, v, S->n, 1, start, end);
}

/* StructuredG::solveInPlaceLTranspose (StructuredG.cpp:45-65) */
static void sg_solve_lt(const bgi * S, double * v)
{
  if(S->type == 0)
    orc_tri_block_diag_ltsolve(S->nblk, S->bs, S->diag, S->ldd, S->off, S->ldo, v, S->n, 1, -1);
  else
    orc_block_arrow_ltsolve(S->nblk, S->bs, S->diag, S->ldd, S->off, S->ldo, S->type == 1, v, S->n, 1, 0, -1);
}

/* (x, y) <- (c x - s y, s x + c y): rule C3 of jrlqp_oracle.c */
static inline void rot(double c, double s, double * x, double * y)
{
  double xv = *x, yv = *y;
  *x = FMA(-s, yv, c * xv); /* (CANON C3 / C6, as in jrlqp_oracle.c) */
  *y = FMA(s, xv, c * yv);
}

/* ElemOrthonormalSequence::applyToTheLeft / applyTransposeToTheLeft for one reflector (OrthonormalSequence.cpp:59-63,
 * :98-102; H is symmetric): v -= tau (E.v) E, the scalar tau*(E.v) formed first */
static void householder_apply(const bgi * S, const oseq * e, double * v)
{
  const double * E = S->pool + e->off;
  double * w = v + e->start;
  const double dd = orc_dot_tree(e->n, E, w, e->start);
  const double s = e->tau * dd;
  for(int i = 0; i < e->n; ++i) w[i] = FMA(-s, E[i], w[i]);
}

/* OrthonormalSequence::applyToTheLeft (OrthonormalSequence.cpp:189-197): v <- Q v, last element first */
static void q_apply(const bgi * S, double * v)
{
  for(int k = S->nseq - 1; k >= 0; --k)
  {
    const oseq * e = &S->seq[k];
    if(e->kind == 0)
      householder_apply(S, e, v);
    else
    {
      /* :76-78: for i = size-1 .. 0: v.applyOnTheLeft(i, i+1, G(c,s)) : (vi, vj) <- (c vi + s vj, -s vi + c vj) */
      const double *c = S->pool + e->off, *s = c + e->size;
      double * w = v + e->start;
      for(int i = e->size - 1; i >= 0; --i) rot(c[i], -s[i], &w[i], &w[i + 1]);
    }
  }
}

/* OrthonormalSequence::applyTransposeToTheLeft (:199-207): v <- Q^T v, first element first */
static void q_apply_t(const bgi * S, double * v)
{
  for(int k = 0; k < S->nseq; ++k)
  {
    const oseq * e = &S->seq[k];
    if(e->kind == 0)
      householder_apply(S, e, v);
    else
    {
      /* :115-117: for i = 0 .. size-1: v.applyOnTheLeft(i, i+1, G^T) : (vi, vj) <- (c vi - s vj, s vi + c vj) */
      const double *c = S->pool + e->off, *s = c + e->size;
      double * w = v + e->start;
      for(int i = 0; i < e->size; ++i) rot(c[i], s[i], &w[i], &w[i + 1]);
    }
  }
}

/* C.col(p) (StructuredC.cpp:59-64): pointer to the non-zero segment, its first row and its length */
static const double * c_col(const bgi * S, int p, int * start, int * len)
{
  const int b = S->to_block[p];
  *start = S->cum_var[b];
  *len = S->cvar[b];
  return S->cblk[b] + (long)(p - S->cum_cstr[b]) * S->cld[b];
}

/* SingleNZSegmentVector::dot (SingleNZSegmentVector.h:52-56) / bound rows: n+ . v without the sign */
static double cstr_dot(const bgi * S, int p, const double * v)
{
  if(p >= S->m) return v[p - S->m];
  int st, len;
  const double * c = c_col(S, p, &st, &len);
  return orc_dot(len, c, v + st);
}

/* BlockGISolver::computeStep_ (BlockGISolver.cpp:164-172) */
static void compute_step(bgi * S, int p, int status)
{
  const int n = S->n, q = S->q;
  double *d = S->d, *z = S->z, *r = S->r;
  /* StructuredJ::premultByJt (StructuredJ.cpp:43-57) */
  for(int i = 0; i < n; ++i) d[i] = 0.0; /* toFullVector */
  if(status <= ORC_EQUALITY)
  {
    int st, len;
    const double * c = c_col(S, p, &st, &len);
    for(int i = 0; i < len; ++i) d[st + i] = c[i];
    sg_solve_l(S, d, st, st + len);
    if(status == ORC_UPPER)
      for(int i = 0; i < n; ++i) d[i] = -d[i]; /* out *= -1 */
  }
  else
  {
    const int pb = p - S->m;
    d[pb] = (status == ORC_UPPER_BOUND) ? -1.0 : 1.0;
    sg_solve_l(S, d, pb, pb + 1);
  }
  q_apply_t(S, d);
  /* StructuredJ::premultByJ2 (StructuredJ.cpp:32-41) */
  for(int i = 0; i < q; ++i) z[i] = 0.0;
  for(int i = q; i < n; ++i) z[i] = d[i];
  q_apply(S, z);
  sg_solve_lt(S, z);
  /* StructuredQR::RSolve (StructuredQR.cpp:64-68) */
  for(int k = 0; k < q; ++k) r[k] = d[k];
  for(int k = q - 1; k >= 0; --k)
  {
    double v = r[k] / RP(S, k, k);
    r[k] = v;
    for(int i = 0; i < k; ++i) r[i] = FMA(-RP(S, i, k), v, r[i]);
  }
}

/* StructuredQR::add (StructuredQR.cpp:70-84) with Eigen's makeHouseholder (Householder.h, real case) */
static void qr_add(bgi * S)
{
  const int n = S->n, q = S->q;
  const double * d = S->d;
  const int len = n - q; /* d.tail(n - q) */
  oseq * e = seq_push(S);
  const long off = S->pool_len;
  double * E = pool_take(S, len);
  const double tail = orc_dot_tree(len - 1, d + q + 1, d + q + 1, q + 1);
  const double c0 = d[q];
  double tau, beta;
  if(tail <= 2.2250738585072014e-308) /* numeric_limits<double>::min() */
  {
    tau = 0.0;
    beta = c0;
    for(int i = 1; i < len; ++i) E[i] = 0.0;
  }
  else
  {
    beta = sqrt(FMA(c0, c0, tail));
    if(c0 >= 0.0) beta = -beta;
    const double den = c0 - beta;
    for(int i = 1; i < len; ++i) E[i] = d[q + i] / den;
    tau = (beta - c0) / beta;
  }
  E[0] = 1.0; /* ElemOrthonormalSequence::add (OrthonormalSequence.cpp:45) */
  for(int k = 0; k < q; ++k) RP(S, k, q) = d[k];
  RP(S, q, q) = beta;
  e->kind = 0; e->start = q; e->n = len; e->size = 1; e->off = off; e->tau = tau;
  S->q = q + 1;
}

/* StructuredQR::remove (StructuredQR.cpp:86-102) */
static void qr_remove(bgi * S, int l)
{
  const int q = --S->q; /* R still has q + 1 columns */
  oseq * e = seq_push(S);
  const int sz = q - l;
  const long off = S->pool_len;
  double * cs = pool_take(S, 2L * sz);
  for(int i = l; i < q; ++i)
  {
    double c, s;
    for(int k = 0; k < i; ++k) RP(S, k, i) = RP(S, k, i + 1);
    orc_givens(RP(S, i, i + 1), RP(S, i + 1, i + 1), &c, &s, &RP(S, i, i));
    for(int k = i + 2; k <= q; ++k) rot(c, s, &RP(S, i, k), &RP(S, i + 1, k));
    cs[i - l] = c;
    cs[sz + (i - l)] = s;
  }
  e->kind = 1; e->start = sz ? l : 0; e->n = sz + 1; e->size = sz; e->off = off; e->tau = 0.0;
}

/* ------------------------------------------------------------------------------------------- */
/* BlockGISolver::solve                                                                        */
/* ------------------------------------------------------------------------------------------- */
int orc_blockgi_solve(int type, int nblk, const int * bs, double ** diag, const int * ldd, double ** off, const int * ldo,
                      const double * a, int ncb, const int * cvar, const int * ccstr, const double ** cblk, const int * cld,
                      const double * bl, const double * bu, const double * xl, const double * xu, int nb, int max_iter,
                      double big_bnd, double * xout, double * uout, int * status_out, int * active_out, int * nactive_out,
                      int * iters_out, double * f_out, orc_trace_rec * trace, int trace_cap)
{
  bgi S;
  memset(&S, 0, sizeof S);
  S.type = type; S.nblk = nblk; S.bs = bs; S.diag = diag; S.ldd = ldd; S.off = off; S.ldo = ldo;
  S.ncb = ncb; S.cvar = cvar; S.ccstr = ccstr; S.cblk = cblk; S.cld = cld;
  S.bl = bl; S.bu = bu; S.xl = xl; S.xu = xu; S.nb = nb;
  int n = 0, m = 0, nvc = 0;
  for(int i = 0; i < nblk; ++i) n += bs[i];
  for(int i = 0; i < ncb; ++i) { m += ccstr[i]; nvc += cvar[i]; }
  S.n = n; S.m = m;
  int rc = ORC_UNKNOWN, it = 0;
  if(nvc != n || (nb != 0 && nb != n)) { rc = ORC_INCONSISTENT_INPUT; n = 0; }

  S.to_block = (int *)malloc(sizeof(int) * (size_t)(m + 1));
  S.cum_var = (int *)malloc(sizeof(int) * (size_t)(ncb + 1));
  S.cum_cstr = (int *)malloc(sizeof(int) * (size_t)(ncb + 1));
  {
    int cv = 0, cc = 0;
    for(int i = 0; i < ncb; ++i)
    {
      S.cum_var[i] = cv; S.cum_cstr[i] = cc;
      for(int k = 0; k < ccstr[i]; ++k) S.to_block[cc + k] = i;
      cv += cvar[i]; cc += ccstr[i];
    }
    S.cum_var[ncb] = cv; S.cum_cstr[ncb] = cc;
  }
  const int qmax = (S.n < m + nb ? S.n : m + nb) + 1;
  S.d = (double *)calloc((size_t)(3 * S.n + 2 * (m + nb + 2) + m + 1), sizeof(double));
  S.z = S.d + S.n; S.x = S.z + S.n; S.u = S.x + S.n; S.r = S.u + (m + nb + 2); S.cx = S.r + (m + nb + 2);
  S.Rp = (double *)calloc((size_t)((long)qmax * (qmax + 1) / 2 + 1), sizeof(double));
  S.A = orc_as_new(m, nb);
  double *x = S.x, *z = S.z, *u = S.u, *r = S.r;
  if(rc == ORC_INCONSISTENT_INPUT) goto done;

  /* ---- init_ (BlockGISolver.cpp:62-107) ---- */
  /* processInitialActiveSet (:290-374): data equalities / fixed variables would be activated here.  The reference's
   * initializePrimalDualPoints only handles an EMPTY initial active set (assert, :475) -- anything else is not
   * functional in the reference; it is reported as INCONSISTENT_INPUT here. */
  for(int i = 0; i < nb; ++i)
    if(xl[i] == xu[i]) { rc = ORC_INCONSISTENT_INPUT; goto done; }
  for(int i = 0; i < m; ++i)
    if(bl[i] == bu[i]) { rc = ORC_INCONSISTENT_INPUT; goto done; }
  /* lltInPlace (:71, StructuredG.cpp:22-43) */
  {
    int ok = (type == 0) ? orc_tri_block_diag_llt(nblk, bs, diag, ldd, off, ldo)
                         : orc_block_arrow_llt(nblk, bs, diag, ldd, off, ldo, type == 1);
    if(!ok) { rc = ORC_NON_POS_HESSIAN; goto done; }
  }
  /* initializePrimalDualPoints (:476-481): x = -G^-1 a, f = a.x / 2 */
  for(int i = 0; i < n; ++i) x[i] = a[i];
  sg_solve_l(&S, x, 0, n);
  sg_solve_lt(&S, x);
  for(int i = 0; i < n; ++i) x[i] = -x[i];
  S.f = 0.5 * orc_dot_tree(n, a, x, 0);

  /* ---- DualSolver::solve loop (DualSolver.cpp:96-167) ---- */
  {
    int skip_step1 = 0, p = -1, pstatus = ORC_INACTIVE;
    rc = ORC_MAX_ITER_REACHED;
    for(; it < max_iter; ++it)
    {
      const int q = S.q;
      if(!skip_step1)
      {
        /* selectViolatedConstraint_ (BlockGISolver.cpp:109-162); C^T x first (StructuredC.cpp:66-77) */
        for(int b = 0; b < ncb; ++b)
          for(int k = 0; k < ccstr[b]; ++k)
            S.cx[S.cum_cstr[b] + k] = orc_dot(cvar[b], cblk[b] + (long)k * cld[b], x + S.cum_var[b]);
        double smin = 0.0;
        p = -1;
        pstatus = ORC_INACTIVE;
        for(int i = 0; i < m; ++i)
        {
          if(orc_as_status(S.A, i) != ORC_INACTIVE) continue;
          const double sl = S.cx[i] - bl[i];
          if(sl < smin) { smin = sl; p = i; pstatus = ORC_LOWER; }
          else
          {
            const double su = bu[i] - S.cx[i];
            if(su < smin) { smin = su; p = i; pstatus = ORC_UPPER; }
          }
        }
        for(int i = 0; i < nb; ++i)
        {
          if(orc_as_status(S.A, m + i) != ORC_INACTIVE) continue;
          const double sl = x[i] - xl[i];
          if(sl < smin) { smin = sl; p = m + i; pstatus = ORC_LOWER_BOUND; }
          else
          {
            const double su = xu[i] - x[i];
            if(su < smin) { smin = su; p = m + i; pstatus = ORC_UPPER_BOUND; }
          }
        }
        if(pstatus == ORC_INACTIVE) { rc = ORC_SUCCESS; break; }
        u[q] = 0.0;
      }

      compute_step(&S, p, pstatus);

      /* computeStepLength_ (BlockGISolver.cpp:174-242); the status is looked up with the POSITION k (:186) */
      double t1 = big_bnd, t2 = big_bnd;
      int l = 0;
      for(int k = 0; k < q; ++k)
      {
        const int sk = orc_as_status(S.A, k);
        if(sk != ORC_EQUALITY && sk != ORC_FIXED && r[k] > 0.0)
        {
          const double tk = u[k] / r[k];
          if(tk < t1) { t1 = tk; l = k; }
        }
      }
      double cz = 0.0;
      if(sqrt(orc_dot_tree(n, z, z, 0)) > 1e-14)
      {
        const double b = (pstatus == ORC_LOWER) ? bl[p] : (pstatus == ORC_UPPER) ? bu[p]
                         : (pstatus == ORC_LOWER_BOUND) ? xl[p - m] : xu[p - m];
        const double cxp = cstr_dot(&S, p, x);
        cz = cstr_dot(&S, p, z);
        t2 = (b - cxp) / cz;
      }
      const double t = (t2 < t1) ? t2 : t1;

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
        for(int k = l; k < q; ++k) u[k] = u[k + 1]; /* DualSolver::removeConstraint (:237-244) */
        orc_as_deactivate(S.A, l);
        qr_remove(&S, l);
        skip_step1 = 1;
      }
      else
      {
        for(int i = 0; i < n; ++i) x[i] = FMA(t, z[i], x[i]);
        /* dot_ (BlockGISolver.cpp:255-273): sign of the normal */
        const double nz = (pstatus == ORC_UPPER || pstatus == ORC_UPPER_BOUND) ? -cz : cz;
        S.f = FMA(t * nz, 0.5 * t + u[q], S.f);
        for(int k = 0; k < q; ++k) u[k] = FMA(-t, r[k], u[k]);
        u[q] += t;
        if(t == t2)
        {
          orc_as_activate(S.A, p, pstatus);
          qr_add(&S);
          skip_step1 = 0;
        }
        else
        {
          for(int k = l; k < q; ++k) u[k] = u[k + 1];
          orc_as_deactivate(S.A, l);
          qr_remove(&S, l);
          skip_step1 = 1;
        }
      }
    }
  }

done:
  for(int i = 0; i < S.n; ++i) xout[i] = x[i];
  {
    int cnt[8];
    orc_as_counters(S.A, cnt);
    const int nact = cnt[0];
    if(uout)
    {
      for(int i = 0; i < m + nb; ++i) uout[i] = 0.0;
      for(int k = 0; k < nact; ++k)
      {
        const int i = orc_as_at(S.A, k);
        const int st = orc_as_status(S.A, i);
        uout[i] = (st == ORC_UPPER || st == ORC_UPPER_BOUND) ? u[k] : -u[k];
      }
    }
    if(status_out) for(int i = 0; i < m + nb; ++i) status_out[i] = orc_as_status(S.A, i);
    if(active_out) for(int k = 0; k < nact; ++k) active_out[k] = orc_as_at(S.A, k);
    if(nactive_out) *nactive_out = nact;
  }
  if(iters_out) *iters_out = it;
  if(f_out) *f_out = S.f;
  orc_as_free(S.A);
  free(S.to_block); free(S.cum_var); free(S.cum_cstr);
  free(S.d); free(S.Rp); free(S.seq); free(S.pool);
  return rc;
}

/* =========================================================================================== */
/* Contract C9 (round 5): the same solver without an orthonormal sequence                       */
/* =========================================================================================== */
/*
 * The reference keeps Q of the QR factorisation  L^-1 N = Q [R; 0]  as an OrthonormalSequence -- one Householder element per added
 * constraint, one Givens sequence per removed one (src/structured/StructuredQR.cpp:70-102) -- and applies EVERY element twice per
 * iteration (src/structured/StructuredJ.cpp:32-57, src/internal/OrthonormalSequence.cpp:189-207): O(n k) work and memory traffic per
 * iteration, k = elements made so far.  What an iteration needs of Q and R is
 *     r  = R^-1 Q1^T b            (b = L^-1 n+; the multipliers' direction, StructuredQR.cpp:64-68)
 *     y  = Q2 Q2^T b = b - Q1 Q1^T b = b - B r      (B = L^-1 N = Q1 R;  z = L^-T y)
 * i.e. the least-squares problem  min |b - B r|.  B is SPARSE -- column k is L^-1 n_k, non-zero only in the Hessian blocks under the
 * constraint's window and in the head block (arrows) -- so C9 keeps
 *     B   the columns L^-1 n_k of the active constraints (they are the b's of the iterations that added them), and
 *     S   = R^-1, upper triangular q x q (the inverse Cholesky factor of  B^T B = N^T G^-1 N),
 * and solves the least-squares problem by the corrected semi-normal equations (Bjorck 1987): r = S S^T B^T b, y = b - B r, then ONE
 * refinement step on the residual: r += S S^T B^T y, y -= B (S S^T B^T y).  No n x q object is ever touched: one iteration is O(q^2)
 * (four triangular matrix-vector products) plus O(q) sparse columns.
 *   add     S gets the column [-r / rho; 1 / rho], rho = |y| (the bordered inverse of [R d1; 0 rho]); B gets the column b
 *   remove  row l and column l leave: the rotations that fold S(l, l..q-1) into the last column -- orc_givens_chain (CANON C7) on that
 *           row reversed -- are applied to columns l .. q-1 of S, row by row (rows are independent); row l and the last column go
 * The iterates agree with the sequence form up to rounding (tests/test_oracle_blockgi_literal.py: identical decisions).
 *
 * Rounding rules:
 *   - b, L^-1, L^-T, C^T x, n+.x, n+.z, u, f: as in the sequence form (C1-C3);
 *   - BLOCK ORDER: off-head blocks in logical order then the head block for the arrows (the order of blockArrowLSolve's permuted
 *     result, blockArrowLLT.cpp:163-170), natural order for the tridiagonal shape;
 *   - a reduction over a whole vector of unknowns (|y|^2, |z|^2, a.x): per block the products summed by the adjacent-first pairwise
 *     tree (orc_dot_tree, phase 0), the block sums combined by the adjacent-first pairwise tree over the block order;
 *   - GROUPS: an active constraint belongs to the group of its block of C (general constraints, group = block index) or of the Hessian
 *     block that holds its variable (bounds, group = ncb + block-order index); reach(group) = the Hessian blocks in which its b can be
 *     non-zero: the off-head blocks under the window plus the head (arrows), every block from the first one under the window (tri);
 *   - g_k = B_k . v (v = b or y): per common block of reach(k) and reach(p) [v = y: reach(k)] one fma chain over the block's rows
 *     ascending from +0, the block values added up in block order (a chain of additions starting from the first block's value);
 *   - d1 = S^T g: per column an fma chain over the rows ascending from +0;  r = S d1: per row an fma chain over the columns DESCENDING
 *     from +0 (the order of a back substitution);
 *   - B r on block kappa: per group that reaches kappa (static list, group order) an fma chain over the group's active constraints in
 *     active-set order from +0, the group values combined by the adjacent-first pairwise tree over that list; y = v - that;
 *   - q == n: y = 0 (Q2 is empty: StructuredJ.cpp:36 multiplies an empty tail).
 */
typedef struct
{
  int nblk;
  int off_perm[4097], off_phys[4097], len[4097]; /* block order: offset in permuted / original coordinates, rows */
} blkorder;

static void blkorder_init(const bgi * S, blkorder * B)
{
  const int nblk = S->nblk;
  B->nblk = nblk;
  int phys[4097];
  for(int i = 0, o = 0; i < nblk; ++i) { phys[i] = o; o += S->bs[i]; }
  int op = 0;
  for(int k = 0; k < nblk; ++k)
  {
    const int b = (S->type == 1) ? (k + 1) % nblk : k; /* dsel: logical block k of the up arrow is physical block k + 1, head = 0 */
    B->off_perm[k] = op;
    B->off_phys[k] = phys[b];
    B->len[k] = S->bs[b];
    op += S->bs[b];
  }
}

static double tree_sum(int n, double * p) /* adjacent-first pairwise tree, p has room for the next power of two */
{
  int size = 1;
  while(size < n) size <<= 1;
  for(int i = n; i < size; ++i) p[i] = 0.0;
  for(int h = 1; h < size; h <<= 1)
    for(int i = 0; i < size; i += 2 * h) p[i] = p[i] + p[i + h];
  return n > 0 ? p[0] : 0.0;
}

/* C9 reduction of a.b over the block order; permuted: the vectors are in permuted coordinates (b, y) or original ones (x, z, a) */
static double blk_reduce(const blkorder * B, const double * a, const double * b, int permuted)
{
  double p[8192];
  for(int k = 0; k < B->nblk; ++k)
  {
    const int o = permuted ? B->off_perm[k] : B->off_phys[k];
    p[k] = orc_dot_tree(B->len[k], a + o, b + o, 0);
  }
  return tree_sum(B->nblk, p);
}

typedef struct
{
  int klo, khi; /* off-head blocks (block-order index) under the window; klo > khi: none.  tri: khi = nblk - 1 */
} reach_t;

/* reach of a constraint whose normal is non-zero on the ORIGINAL rows [wst, wen) */
static reach_t reach_of(const bgi * S, const blkorder * B, int wst, int wen)
{
  reach_t r = {B->nblk, -1};
  const int last = (S->type == 0) ? B->nblk : B->nblk - 1; /* arrows: the head is not an off-head block */
  for(int k = 0; k < last; ++k)
    if(B->off_phys[k] < wen && wst < B->off_phys[k] + B->len[k])
    {
      if(k < r.klo) r.klo = k;
      r.khi = k;
    }
  if(S->type == 0 && r.klo <= r.khi) r.khi = B->nblk - 1;
  return r;
}
static int reaches(const bgi * S, const blkorder * B, reach_t r, int k)
{
  if(S->type != 0 && k == B->nblk - 1) return 1; /* the head of an arrow */
  return r.klo <= k && k <= r.khi;
}

int orc_blockgi_solve_c9(int type, int nblk, const int * bs, double ** diag, const int * ldd, double ** off, const int * ldo,
                         const double * a, int ncb, const int * cvar, const int * ccstr, const double ** cblk, const int * cld,
                         const double * bl, const double * bu, const double * xl, const double * xu, int nb, int max_iter,
                         double big_bnd, double * xout, double * uout, int * status_out, int * active_out, int * nactive_out,
                         int * iters_out, double * f_out, orc_trace_rec * trace, int trace_cap)
{
  bgi S;
  memset(&S, 0, sizeof S);
  S.type = type; S.nblk = nblk; S.bs = bs; S.diag = diag; S.ldd = ldd; S.off = off; S.ldo = ldo;
  S.ncb = ncb; S.cvar = cvar; S.ccstr = ccstr; S.cblk = cblk; S.cld = cld;
  S.bl = bl; S.bu = bu; S.xl = xl; S.xu = xu; S.nb = nb;
  int n = 0, m = 0, nvc = 0;
  for(int i = 0; i < nblk; ++i) n += bs[i];
  for(int i = 0; i < ncb; ++i) { m += ccstr[i]; nvc += cvar[i]; }
  S.n = n; S.m = m;
  int rc = ORC_UNKNOWN, it = 0;
  if(nvc != n || (nb != 0 && nb != n) || nblk > 4096) { rc = ORC_INCONSISTENT_INPUT; n = 0; }

  S.to_block = (int *)malloc(sizeof(int) * (size_t)(m + 1));
  S.cum_var = (int *)malloc(sizeof(int) * (size_t)(ncb + 1));
  S.cum_cstr = (int *)malloc(sizeof(int) * (size_t)(ncb + 1));
  {
    int cv = 0, cc = 0;
    for(int i = 0; i < ncb; ++i)
    {
      S.cum_var[i] = cv; S.cum_cstr[i] = cc;
      for(int k = 0; k < ccstr[i]; ++k) S.to_block[cc + k] = i;
      cv += cvar[i]; cc += ccstr[i];
    }
    S.cum_var[ncb] = cv; S.cum_cstr[ncb] = cc;
  }
  const int qmax = (S.n < m + nb ? S.n : m + nb) + 1;
  S.d = (double *)calloc((size_t)(3 * S.n + 2 * (m + nb + 2) + m + 1), sizeof(double));
  S.z = S.d + S.n; S.x = S.z + S.n; S.u = S.x + S.n; S.r = S.u + (m + nb + 2); S.cx = S.r + (m + nb + 2);
  S.A = orc_as_new(m, nb);
  const long ld = S.n > 0 ? S.n : 1;
  const int ngrp = ncb + (nb ? nblk : 0);
  double * Bc = (double *)calloc((size_t)(ld * qmax), sizeof(double));       /* column storage by SLOT, permuted coordinates */
  double * Sm = (double *)calloc((size_t)qmax * (size_t)qmax, sizeof(double)); /* S(i,j) at i + j qmax, by position */
  double * wq = (double *)calloc((size_t)(6 * qmax + 8), sizeof(double));
  double *g = wq, *d1 = g + qmax, *dl = d1 + qmax, *gc = dl + qmax, *gs = gc + qmax, *grev = gs + qmax;
  int * slot = (int *)calloc((size_t)(4 * qmax + 4), sizeof(int)); /* position -> column slot */
  int *freeslot = slot + qmax, *grp = freeslot + qmax, *lpos = grp + qmax; /* free slots (stack), position -> group, scratch */
  reach_t * rch = (reach_t *)calloc((size_t)qmax + 1, sizeof(reach_t));    /* position -> reach */
  int nfree = 0;
  for(int k = qmax - 1; k >= 0; --k) freeslot[nfree++] = k;
  double *x = S.x, *z = S.z, *u = S.u, *r = S.r, *b = S.d;
  double * y = (double *)calloc((size_t)ld + 1, sizeof(double));
  blkorder * B = (blkorder *)malloc(sizeof(blkorder));
  reach_t * greach = (reach_t *)calloc((size_t)ngrp + 1, sizeof(reach_t));
  int * gstart = (int *)calloc((size_t)(ngrp + 2), sizeof(int)); /* per-iteration: positions of group g are lpos[gstart[g] .. gstart[g+1]) */
  double * slots = (double *)calloc((size_t)2 * (size_t)(ngrp + 2) + 2, sizeof(double));
  if(rc == ORC_INCONSISTENT_INPUT) goto done;
  blkorder_init(&S, B);
  for(int gI = 0; gI < ncb; ++gI) greach[gI] = reach_of(&S, B, S.cum_var[gI], S.cum_var[gI] + cvar[gI]);
  if(nb)
    for(int k = 0; k < nblk; ++k) greach[ncb + k] = reach_of(&S, B, B->off_phys[k], B->off_phys[k] + B->len[k]);

  /* ---- init_ (BlockGISolver.cpp:62-107), as in orc_blockgi_solve ---- */
  for(int i = 0; i < nb; ++i)
    if(xl[i] == xu[i]) { rc = ORC_INCONSISTENT_INPUT; goto done; }
  for(int i = 0; i < m; ++i)
    if(bl[i] == bu[i]) { rc = ORC_INCONSISTENT_INPUT; goto done; }
  {
    int ok = (type == 0) ? orc_tri_block_diag_llt(nblk, bs, diag, ldd, off, ldo)
                         : orc_block_arrow_llt(nblk, bs, diag, ldd, off, ldo, type == 1);
    if(!ok) { rc = ORC_NON_POS_HESSIAN; goto done; }
  }
  for(int i = 0; i < n; ++i) x[i] = a[i];
  sg_solve_l(&S, x, 0, n);
  sg_solve_lt(&S, x);
  for(int i = 0; i < n; ++i) x[i] = -x[i];
  S.f = 0.5 * blk_reduce(B, a, x, 0);

  /* ---- DualSolver::solve loop (DualSolver.cpp:96-167) ---- */
  {
    int skip_step1 = 0, p = -1, pstatus = ORC_INACTIVE;
    rc = ORC_MAX_ITER_REACHED;
    for(; it < max_iter; ++it)
    {
      const int q = S.q;
      if(!skip_step1)
      {
        for(int bb = 0; bb < ncb; ++bb)
          for(int k = 0; k < ccstr[bb]; ++k)
            S.cx[S.cum_cstr[bb] + k] = orc_dot(cvar[bb], cblk[bb] + (long)k * cld[bb], x + S.cum_var[bb]);
        double smin = 0.0;
        p = -1;
        pstatus = ORC_INACTIVE;
        for(int i = 0; i < m; ++i)
        {
          if(orc_as_status(S.A, i) != ORC_INACTIVE) continue;
          const double sl = S.cx[i] - bl[i];
          if(sl < smin) { smin = sl; p = i; pstatus = ORC_LOWER; }
          else
          {
            const double su = bu[i] - S.cx[i];
            if(su < smin) { smin = su; p = i; pstatus = ORC_UPPER; }
          }
        }
        for(int i = 0; i < nb; ++i)
        {
          if(orc_as_status(S.A, m + i) != ORC_INACTIVE) continue;
          const double sl = x[i] - xl[i];
          if(sl < smin) { smin = sl; p = m + i; pstatus = ORC_LOWER_BOUND; }
          else
          {
            const double su = xu[i] - x[i];
            if(su < smin) { smin = su; p = m + i; pstatus = ORC_UPPER_BOUND; }
          }
        }
        if(pstatus == ORC_INACTIVE) { rc = ORC_SUCCESS; break; }
        u[q] = 0.0;
      }

      /* ---- computeStep_ under C9 ---- */
      /* b = L^-1 n+ (StructuredJ.cpp:43-57, unchanged; permuted coordinates for the up arrow) */
      int pgrp;
      reach_t preach;
      for(int i = 0; i < n; ++i) b[i] = 0.0;
      if(pstatus <= ORC_EQUALITY)
      {
        int st, len;
        const double * c = c_col(&S, p, &st, &len);
        for(int i = 0; i < len; ++i) b[st + i] = c[i];
        sg_solve_l(&S, b, st, st + len);
        if(pstatus == ORC_UPPER)
          for(int i = 0; i < n; ++i) b[i] = -b[i];
        pgrp = S.to_block[p];
        preach = greach[pgrp];
      }
      else
      {
        const int pb = p - S.m;
        b[pb] = (pstatus == ORC_UPPER_BOUND) ? -1.0 : 1.0;
        sg_solve_l(&S, b, pb, pb + 1);
        preach = reach_of(&S, B, pb, pb + 1);
        pgrp = ncb + ((S.type != 0 && preach.klo > preach.khi) ? nblk - 1 : preach.klo);
      }
      double nrm2 = 0.0;
      /* positions of every group, in active-set order */
      for(int gI = 0; gI <= ngrp; ++gI) gstart[gI] = 0;
      for(int k = 0; k < q; ++k) ++gstart[grp[k] + 1];
      for(int gI = 0; gI < ngrp; ++gI) gstart[gI + 1] += gstart[gI];
      {
        int * fill = (int *)slots; /* scratch: next free entry per group (ngrp ints fit in the doubles) */
        for(int gI = 0; gI < ngrp; ++gI) fill[gI] = gstart[gI];
        for(int k = 0; k < q; ++k) lpos[fill[grp[k]]++] = k;
      }
      for(int k = 0; k < q; ++k) r[k] = 0.0;
      for(int i = 0; i < n; ++i) y[i] = b[i];
      for(int pass = 0; pass < 2 && q > 0; ++pass)
      {
        /* g = B^T y  (pass 0: y = b) */
        for(int k = 0; k < q; ++k)
        {
          const double * col = Bc + (long)slot[k] * ld;
          double acc = 0.0;
          int first = 1;
          for(int kap = 0; kap < nblk; ++kap)
          {
            if(!reaches(&S, B, rch[k], kap)) continue;
            if(pass == 0 && !reaches(&S, B, preach, kap)) continue; /* b is zero there */
            const int o = B->off_perm[kap];
            double v = 0.0;
            for(int i = 0; i < B->len[kap]; ++i) v = FMA(col[o + i], y[o + i], v);
            acc = first ? v : acc + v;
            first = 0;
          }
          g[k] = acc;
        }
        /* d1 = S^T g, dl = S d1 */
        for(int j = 0; j < q; ++j)
        {
          double acc = 0.0;
          for(int i = 0; i <= j; ++i) acc = FMA(Sm[i + (long)j * qmax], g[i], acc);
          d1[j] = acc;
        }
        for(int i = 0; i < q; ++i)
        {
          double acc = 0.0;
          for(int j = q - 1; j >= i; --j) acc = FMA(Sm[i + (long)j * qmax], d1[j], acc);
          dl[i] = acc;
        }
        for(int k = 0; k < q; ++k) r[k] = (pass == 0) ? dl[k] : r[k] + dl[k];
        if(q == n) /* y = 0 by rule: nothing to refine against */
        {
          for(int i = 0; i < n; ++i) y[i] = 0.0;
          break;
        }
        /* y -= B dl */
        for(int kap = 0; kap < nblk; ++kap)
        {
          /* groups that reach this block, in group order */
          int nT = 0, any = 0;
          const int o = B->off_perm[kap], len = B->len[kap];
          for(int gI = 0; gI < ngrp; ++gI)
            if(reaches(&S, B, greach[gI], kap)) { ++nT; any |= (gstart[gI + 1] > gstart[gI]); }
          if(!any) continue;
          for(int i = 0; i < len; ++i)
          {
            int t = 0;
            for(int gI = 0; gI < ngrp; ++gI)
            {
              if(!reaches(&S, B, greach[gI], kap)) continue;
              double acc = 0.0;
              for(int e = gstart[gI]; e < gstart[gI + 1]; ++e)
              {
                const int k = lpos[e];
                acc = FMA(dl[k], Bc[(long)slot[k] * ld + o + i], acc);
              }
              slots[t++] = acc;
            }
            y[o + i] = y[o + i] - tree_sum(nT, slots);
          }
        }
      }
      if(q != n) nrm2 = blk_reduce(B, y, y, 1);
      /* z = L^-T y */
      for(int i = 0; i < n; ++i) z[i] = y[i];
      sg_solve_lt(&S, z);

      /* computeStepLength_ (BlockGISolver.cpp:174-242) */
      double t1 = big_bnd, t2 = big_bnd;
      int l = 0;
      for(int k = 0; k < q; ++k)
      {
        const int sk = orc_as_status(S.A, k);
        if(sk != ORC_EQUALITY && sk != ORC_FIXED && r[k] > 0.0)
        {
          const double tk = u[k] / r[k];
          if(tk < t1) { t1 = tk; l = k; }
        }
      }
      double cz = 0.0;
      if(sqrt(blk_reduce(B, z, z, 0)) > 1e-14)
      {
        const double bnd = (pstatus == ORC_LOWER) ? bl[p] : (pstatus == ORC_UPPER) ? bu[p]
                           : (pstatus == ORC_LOWER_BOUND) ? xl[p - m] : xu[p - m];
        const double cxp = cstr_dot(&S, p, x);
        cz = cstr_dot(&S, p, z);
        t2 = (bnd - cxp) / cz;
      }
      const double t = (t2 < t1) ? t2 : t1;

      if(trace && it < trace_cap)
      {
        trace[it].p = p; trace[it].status = pstatus; trace[it].l = l; trace[it].t1 = t1; trace[it].t2 = t2;
        trace[it].kind = (t2 >= big_bnd) ? 2 : ((t == t2) ? 0 : 1);
      }

      if(t >= big_bnd) { rc = ORC_INFEASIBLE; break; }

      int drop;
      if(t2 >= big_bnd)
      {
        for(int k = 0; k < q; ++k) u[k] = FMA(-t, r[k], u[k]);
        u[q] += t;
        drop = 1;
      }
      else
      {
        for(int i = 0; i < n; ++i) x[i] = FMA(t, z[i], x[i]);
        const double nz = (pstatus == ORC_UPPER || pstatus == ORC_UPPER_BOUND) ? -cz : cz;
        S.f = FMA(t * nz, 0.5 * t + u[q], S.f);
        for(int k = 0; k < q; ++k) u[k] = FMA(-t, r[k], u[k]);
        u[q] += t;
        drop = !(t == t2);
      }
      if(!drop)
      {
        /* add: S gets the column [-r / rho; 1 / rho], B the column b */
        const double rho = sqrt(nrm2);
        orc_as_activate(S.A, p, pstatus);
        for(int k = 0; k < q; ++k) Sm[k + (long)q * qmax] = -(r[k] / rho);
        Sm[q + (long)q * qmax] = 1.0 / rho;
        const int sl = freeslot[--nfree];
        slot[q] = sl;
        grp[q] = pgrp;
        rch[q] = preach;
        double * col = Bc + (long)sl * ld;
        for(int i = 0; i < n; ++i) col[i] = b[i];
        S.q = q + 1;
        skip_step1 = 0;
      }
      else
      {
        /* remove (DualSolver.cpp:237-244): u, the active set, then S: fold row l into the last column (C7 on the row reversed) */
        for(int k = l; k < q; ++k) u[k] = u[k + 1];
        orc_as_deactivate(S.A, l);
        const int len = q - l;
        if(len > 1)
        {
          double rfin;
          for(int k = 0; k < len; ++k) grev[k] = Sm[l + (long)(q - 1 - k) * qmax];
          orc_givens_chain(len, 0, grev, gc, gs, &rfin);
        }
        for(int i = 0; i < q; ++i)
        {
          if(i == l) continue;
          const int i2 = (i > l) ? i - 1 : i; /* new position of the row */
          int t0 = 0;
          double carry;
          if(i < l)
            carry = Sm[i + (long)l * qmax];
          else
          {
            t0 = i - l - 1; /* the row's first non-zero is S(i, i) = the "next" operand of step t0 */
            carry = 0.0;
          }
          for(int tt = t0; tt < len - 1; ++tt)
          {
            /* rotation tt: (x, y) = (column l+tt+1, column l+tt) <- (c x - s y, s x + c y)  (rule C3 / C6) */
            const double c = gc[len - 2 - tt], sn = gs[len - 2 - tt];
            double xv = Sm[i + (long)(l + tt + 1) * qmax], yv = carry;
            rot(c, sn, &xv, &yv);
            Sm[i2 + (long)(l + tt) * qmax] = yv;
            carry = xv;
          }
          if(i2 != i)
            for(int j = 0; j < l + t0 && j < q - 1; ++j) Sm[i2 + (long)j * qmax] = 0.0; /* (below the diagonal: never read) */
        }
        freeslot[nfree++] = slot[l];
        for(int k = l; k < q - 1; ++k) { slot[k] = slot[k + 1]; grp[k] = grp[k + 1]; rch[k] = rch[k + 1]; }
        S.q = q - 1;
        skip_step1 = 1;
      }
    }
  }

done:
  for(int i = 0; i < S.n; ++i) xout[i] = x[i];
  {
    int cnt[8];
    orc_as_counters(S.A, cnt);
    const int nact = cnt[0];
    if(uout)
    {
      for(int i = 0; i < m + nb; ++i) uout[i] = 0.0;
      for(int k = 0; k < nact; ++k)
      {
        const int i = orc_as_at(S.A, k);
        const int st = orc_as_status(S.A, i);
        uout[i] = (st == ORC_UPPER || st == ORC_UPPER_BOUND) ? u[k] : -u[k];
      }
    }
    if(status_out) for(int i = 0; i < m + nb; ++i) status_out[i] = orc_as_status(S.A, i);
    if(active_out) for(int k = 0; k < nact; ++k) active_out[k] = orc_as_at(S.A, k);
    if(nactive_out) *nactive_out = nact;
  }
  if(iters_out) *iters_out = it;
  if(f_out) *f_out = S.f;
  orc_as_free(S.A);
  free(S.to_block); free(S.cum_var); free(S.cum_cstr);
  free(S.d); free(Bc); free(Sm); free(wq); free(slot); free(rch); free(y); free(B); free(greach); free(gstart); free(slots);
  return rc;
}

/* ------------------------------------------------------------------------------------------- */
/* stand-alone orthonormal-sequence object, for the identities of the reference's InternalTest  */
/* (tests/InternalTest.cpp:35-323): build a sequence, apply it / its transpose to vectors.      */
/* ------------------------------------------------------------------------------------------- */
struct orc_oseq
{
  bgi S;
};

orc_oseq * orc_oseq_new(int n)
{
  orc_oseq * o = (orc_oseq *)calloc(1, sizeof(orc_oseq));
  o->S.n = n;
  return o;
}

void orc_oseq_free(orc_oseq * o)
{
  if(!o) return;
  free(o->S.seq);
  free(o->S.pool);
  free(o);
}

/* prepare(Householder, len, 1) + add(start, essential, tau): makeHouseholder of v (len entries) -> beta returned */
double orc_oseq_add_householder(orc_oseq * o, int start, int len, const double * v)
{
  bgi * S = &o->S;
  /* reuse qr_add's arithmetic on a scratch d */
  double * d = (double *)calloc((size_t)S->n, sizeof(double));
  for(int i = 0; i < len; ++i) d[start + i] = v[i];
  double * save_d = S->d;
  double * save_R = S->Rp;
  const int save_n = S->n, save_q = S->q;
  S->d = d;
  S->Rp = (double *)calloc((size_t)((long)(start + 2) * (start + 3) / 2), sizeof(double));
  S->n = start + len;
  S->q = start;
  qr_add(S);
  const double beta = RP(S, start, start);
  free(S->Rp);
  free(d);
  S->d = save_d; S->Rp = save_R; S->n = save_n; S->q = save_q;
  return beta;
}

/* prepare(Givens, size + 1, size) + add(start + i, G(c[i], s[i])) */
void orc_oseq_add_givens(orc_oseq * o, int start, int size, const double * c, const double * s)
{
  bgi * S = &o->S;
  oseq * e = seq_push(S);
  const long off = S->pool_len;
  double * cs = pool_take(S, 2L * size);
  for(int i = 0; i < size; ++i) { cs[i] = c[i]; cs[size + i] = s[i]; }
  e->kind = 1; e->start = size ? start : 0; e->n = size + 1; e->size = size; e->off = off; e->tau = 0.0;
}

void orc_oseq_apply(const orc_oseq * o, double * v, int transpose)
{
  if(transpose) q_apply_t(&o->S, v); else q_apply(&o->S, v);
}
