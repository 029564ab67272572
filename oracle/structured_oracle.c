/*
 * structured_oracle.c -- CPU oracle for the structured Cholesky family (TEST INFRASTRUCTURE ONLY).
 *
 * Restates src/decomposition/blockArrowLLT.cpp and src/decomposition/triBlockDiagLLT.cpp of the
 * reference with the CANON rounding rules of jrlqp_oracle.c (fma chains, operands in order of
 * availability).  With those rules the block algorithms are bit-identical to the dense canonical
 * LLT / substitutions applied to the assembled matrix (skipped blocks only ever contribute exact
 * zeros) -- tests/test_oracle_structured.py checks exactly that, plus the reference's own 1e-8
 * identities against LAPACK.
 */
#include "jrlqp_oracle.h"

#include <stdlib.h>
#include <string.h>

#define FMA(a, b, c) __builtin_fma((a), (b), (c))
#define AT(P, ld, i, j) (P)[(i) + (long)(j) * (ld)]

static int imax(int a, int b) { return a > b ? a : b; }

/* X <- L^-1 X, L n x n lower (Eigen triangularView<Lower>::solveInPlace<OnTheLeft>), X n x k.
 * Lo: offset (row = col) of the sub-triangle L.bottomRightCorner used by the zero-window variants. */
static void trsm_left_lower(int n, const double * L, int ldl, double * X, int ldx, int k)
{
  for(int c = 0; c < k; ++c)
    for(int j = 0; j < n; ++j)
    {
      double v = AT(X, ldx, j, c) / AT(L, ldl, j, j);
      AT(X, ldx, j, c) = v;
      for(int i = j + 1; i < n; ++i) AT(X, ldx, i, c) = FMA(-AT(L, ldl, i, j), v, AT(X, ldx, i, c));
    }
}

/* X <- L^-T X (triangularView<Lower>.transpose().solveInPlace), backward */
static void trsm_left_lower_t(int n, const double * L, int ldl, double * X, int ldx, int k)
{
  for(int c = 0; c < k; ++c)
    for(int j = n - 1; j >= 0; --j)
    {
      double v = AT(X, ldx, j, c) / AT(L, ldl, j, j);
      AT(X, ldx, j, c) = v;
      for(int i = 0; i < j; ++i) AT(X, ldx, i, c) = FMA(-AT(L, ldl, j, i), v, AT(X, ldx, i, c));
    }
}

/* X <- X L^-T (solveInPlace<OnTheRight> on L.transpose()), X r x n: column j of X becomes available in increasing j */
static void trsm_right_lower_t(int n, const double * L, int ldl, double * X, int ldx, int r)
{
  for(int j = 0; j < n; ++j)
    for(int i = 0; i < r; ++i)
    {
      double s = AT(X, ldx, i, j);
      for(int k = 0; k < j; ++k) s = FMA(-AT(X, ldx, i, k), AT(L, ldl, j, k), s);
      AT(X, ldx, i, j) = s / AT(L, ldl, j, j);
    }
}

/* D.selfadjointView<Lower>().rankUpdate(B, -1): D (nd x nd, lower) -= B B^T, B nd x kb (transB=0) or B^T with B kb x nd (transB=1) */
static void syrk_lower_minus(int nd, double * D, int ldd, const double * B, int ldb, int kb, int transB)
{
  for(int c = 0; c < nd; ++c)
    for(int r = c; r < nd; ++r)
    {
      double s = AT(D, ldd, r, c);
      if(!transB)
        for(int k = 0; k < kb; ++k) s = FMA(-AT(B, ldb, r, k), AT(B, ldb, c, k), s);
      else
        for(int k = 0; k < kb; ++k) s = FMA(-AT(B, ldb, k, r), AT(B, ldb, k, c), s);
      AT(D, ldd, r, c) = s;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* block arrow (blockArrowLLT.cpp)                                                             */
/* ------------------------------------------------------------------------------------------- */
/* get<Up>::D (blockArrowLLT.cpp:22-46): logical diagonal block i */
static int dsel(int nblk, int up, int i) { return up ? (i + 1) % nblk : i; }

/* blockArrowLLT_ (blockArrowLLT.cpp:53-81) */
int orc_block_arrow_llt(int nblk, const int * bs, double ** diag, const int * ldd, double ** side, const int * lds, int up)
{
  const int ib = dsel(nblk, up, nblk - 1);
  double * Db = diag[ib];
  const int nbk = bs[ib];
  for(int i = 0; i < nblk - 1; ++i)
  {
    const int id = dsel(nblk, up, i);
    const int ni = bs[id];
    int ret = orc_llt(ni, diag[id], ldd[id]);
    if(ret > 0) return 0; /* reference quirk: a failure at column 0 is not detected (:65) */
    if(up)
    {
      /* side[i] (ni x nbk) <- Li^-1 side[i]; it then stores Bi^T (:68-70) */
      trsm_left_lower(ni, diag[id], ldd[id], side[i], lds[i], nbk);
      syrk_lower_minus(nbk, Db, ldd[ib], side[i], lds[i], ni, 1);
    }
    else
    {
      /* side[i] (nbk x ni) <- side[i] Li^-T (:71-72) */
      trsm_right_lower_t(ni, diag[id], ldd[id], side[i], lds[i], nbk);
      syrk_lower_minus(nbk, Db, ldd[ib], side[i], lds[i], ni, 0);
    }
  }
  int ret = orc_llt(nbk, Db, ldd[ib]);
  return ret <= 0; /* (:80) */
}

/* Mb -= B(i).middleCols(s, ni-s) * Mi   with B(i) = side[i] (down) or side[i]^T (up) */
static void arrow_gemm_minus(int up, const double * Si, int ldsi, int s, int ni, int nbk, const double * Mi, double * Mb, int ldm, int nrhs)
{
  for(int c = 0; c < nrhs; ++c)
    for(int r = 0; r < nbk; ++r)
    {
      double acc = AT(Mb, ldm, r, c);
      for(int k = s; k < ni; ++k)
      {
        double bv = up ? AT(Si, ldsi, k, r) : AT(Si, ldsi, r, k);
        acc = FMA(-bv, AT(Mi, ldm, k - s, c), acc);
      }
      AT(Mb, ldm, r, c) = acc;
    }
}

/* blockArrowLSolve_ (blockArrowLLT.cpp:92-152) on the already permuted right-hand side */
static void arrow_lsolve_(int nblk, const int * bs, double ** diag, const int * ldd, double ** side, const int * lds, int up,
                          double * M, int ldm, int nrhs, int start, int end)
{
  const int ib = dsel(nblk, up, nblk - 1);
  const int nbk = bs[ib];
  int ntot = 0;
  for(int i = 0; i < nblk; ++i) ntot += bs[i];
  double * Mb = M + (ntot - nbk);
  int n = 0;
  for(int i = 0; i < nblk - 1; ++i)
  {
    const int id = dsel(nblk, up, i);
    const int ni = bs[id];
    int s = imax(start - n, 0);
    if(ni < s || end <= n) { n += ni; continue; } /* (:121-127) */
    double * Mi = M + n + s;
    trsm_left_lower(ni - s, diag[id] + s + (long)s * ldd[id], ldd[id], Mi, ldm, nrhs);
    arrow_gemm_minus(up, side[i], lds[i], s, ni, nbk, Mi, Mb, ldm, nrhs);
    n += ni;
  }
  trsm_left_lower(nbk, diag[ib], ldd[ib], M + n, ldm, nrhs);
}

/* blockArrowLSolve (blockArrowLLT.cpp:154-176) */
void orc_block_arrow_lsolve(int nblk, const int * bs, double ** diag, const int * ldd, double ** side, const int * lds, int up,
                            double * M, int ldm, int nrhs, int start, int end)
{
  int ntot = 0;
  for(int i = 0; i < nblk; ++i) ntot += bs[i];
  if(end < 0) end = ntot;
  if(up)
  {
    const int n0 = bs[0], s = ntot - n0;
    double * tmp = (double *)malloc(sizeof(double) * (size_t)ntot);
    for(int c = 0; c < nrhs; ++c)
    {
      double * col = M + (long)c * ldm;
      memcpy(tmp, col, sizeof(double) * (size_t)ntot);
      memcpy(col, tmp + n0, sizeof(double) * (size_t)s);
      memcpy(col + s, tmp, sizeof(double) * (size_t)n0);
    }
    free(tmp);
    arrow_lsolve_(nblk, bs, diag, ldd, side, lds, 1, M, ldm, nrhs, imax(0, start - n0), imax(0, end - n0));
  }
  else
    arrow_lsolve_(nblk, bs, diag, ldd, side, lds, 0, M, ldm, nrhs, start, end);
}

/* blockArrowLTransposeSolve_ (blockArrowLLT.cpp:179-253) */
static void arrow_ltsolve_(int nblk, const int * bs, double ** diag, const int * ldd, double ** side, const int * lds, int up,
                           double * M, int ldm, int nrhs, int start, int end)
{
  const int ib = dsel(nblk, up, nblk - 1);
  const int nbk = bs[ib];
  int stot = 0;
  for(int i = 0; i < nblk; ++i) stot += bs[i];
  double * Mb = M + (stot - nbk);
  int zero = 0;
  if(end > stot - nbk)
  {
    int r = end - stot + nbk;
    trsm_left_lower_t(r, diag[ib], ldd[ib], Mb, ldm, nrhs);
  }
  else
    zero = 1;

  int n = 0;
  for(int i = 0; i < nblk - 1; ++i)
  {
    const int id = dsel(nblk, up, i);
    const int ni = bs[id];
    double * Mi = M + n;
    if(zero)
    {
      if(start >= n + ni) { n += ni; continue; }
    }
    else
    {
      /* Mi -= B(i)^T Mb, contributions in decreasing row index of Mb (backward substitution order) */
      for(int c = 0; c < nrhs; ++c)
        for(int r = 0; r < ni; ++r)
        {
          double acc = AT(Mi, ldm, r, c);
          for(int k = nbk - 1; k >= 0; --k)
          {
            double bv = up ? AT(side[i], lds[i], r, k) : AT(side[i], lds[i], k, r);
            acc = FMA(-bv, AT(Mb, ldm, k, c), acc);
          }
          AT(Mi, ldm, r, c) = acc;
        }
    }
    if(end >= n)
    {
      if(end >= n + ni)
        trsm_left_lower_t(ni, diag[id], ldd[id], Mi, ldm, nrhs);
      else
        trsm_left_lower_t(end - n, diag[id], ldd[id], Mi, ldm, nrhs);
    }
    n += ni;
  }
}

/* blockArrowLTransposeSolve (blockArrowLLT.cpp:255-277) */
void orc_block_arrow_ltsolve(int nblk, const int * bs, double ** diag, const int * ldd, double ** side, const int * lds, int up,
                             double * M, int ldm, int nrhs, int start, int end)
{
  int ntot = 0;
  for(int i = 0; i < nblk; ++i) ntot += bs[i];
  if(end < 0) end = ntot;
  if(up)
  {
    arrow_ltsolve_(nblk, bs, diag, ldd, side, lds, 1, M, ldm, nrhs, start, end);
    const int n0 = bs[0], s = ntot - n0;
    double * tmp = (double *)malloc(sizeof(double) * (size_t)ntot);
    for(int c = 0; c < nrhs; ++c)
    {
      double * col = M + (long)c * ldm;
      memcpy(tmp, col, sizeof(double) * (size_t)ntot);
      memcpy(col + n0, tmp, sizeof(double) * (size_t)s);
      memcpy(col, tmp + s, sizeof(double) * (size_t)n0);
    }
    free(tmp);
  }
  else
    arrow_ltsolve_(nblk, bs, diag, ldd, side, lds, 0, M, ldm, nrhs, start, end);
}

/* ------------------------------------------------------------------------------------------- */
/* block tridiagonal (triBlockDiagLLT.cpp)                                                     */
/* ------------------------------------------------------------------------------------------- */
/* triBlockDiagLLT (triBlockDiagLLT.cpp:9-35) */
int orc_tri_block_diag_llt(int nblk, const int * bs, double ** diag, const int * ldd, double ** sub, const int * lds)
{
  for(int i = 0; i < nblk - 1; ++i)
  {
    int ret = orc_llt(bs[i], diag[i], ldd[i]);
    if(ret > 0) return 0;
    trsm_right_lower_t(bs[i], diag[i], ldd[i], sub[i], lds[i], bs[i + 1]);
    syrk_lower_minus(bs[i + 1], diag[i + 1], ldd[i + 1], sub[i], lds[i], bs[i], 0);
  }
  int ret = orc_llt(bs[nblk - 1], diag[nblk - 1], ldd[nblk - 1]);
  if(ret > 0) return 0;
  return 1;
}

/* triBlockDiagLSolve (triBlockDiagLLT.cpp:37-96) */
void orc_tri_block_diag_lsolve(int nblk, const int * bs, double ** diag, const int * ldd, double ** sub, const int * lds,
                               double * M, int ldm, int nrhs, int start)
{
  int n = 0, l = 0, li = 0, zero = 1;
  for(int i = 0; i < nblk; ++i)
  {
    const int ni = bs[i];
    if(n + ni >= start)
    {
      if(zero)
      {
        int r = n + ni - start;
        int off = ni - r;
        trsm_left_lower(r, diag[i] + off + (long)off * ldd[i], ldd[i], M + start, ldm, nrhs);
        zero = 0;
      }
      else
      {
        double * Mi = M + n;
        const double * Ml = M + l;
        for(int c = 0; c < nrhs; ++c)
          for(int r = 0; r < ni; ++r)
          {
            double acc = AT(Mi, ldm, r, c);
            for(int k = 0; k < li; ++k) acc = FMA(-AT(sub[i - 1], lds[i - 1], r, k), AT(Ml, ldm, k, c), acc);
            AT(Mi, ldm, r, c) = acc;
          }
        trsm_left_lower(ni, diag[i], ldd[i], Mi, ldm, nrhs);
      }
    }
    l = n;
    li = ni;
    n += ni;
  }
}

/* triBlockDiagLTransposeSolve (triBlockDiagLLT.cpp:98-158) */
void orc_tri_block_diag_ltsolve(int nblk, const int * bs, double ** diag, const int * ldd, double ** sub, const int * lds,
                                double * M, int ldm, int nrhs, int end)
{
  int n = 0;
  for(int i = 0; i < nblk; ++i) n += bs[i];
  int l = 0, li = 0, zero = 1;
  if(end < 0) end = n;
  for(int i = nblk - 1; i >= 0; --i)
  {
    const int ni = bs[i];
    if(n - ni < end)
    {
      if(zero)
      {
        int r = end - n + ni;
        trsm_left_lower_t(r, diag[i], ldd[i], M + (n - ni), ldm, nrhs);
        zero = 0;
      }
      else
      {
        double * Mi = M + (n - ni);
        const double * Mn = M + (l - li); /* X[i+1] */
        for(int c = 0; c < nrhs; ++c)
          for(int r = 0; r < ni; ++r)
          {
            double acc = AT(Mi, ldm, r, c);
            for(int k = li - 1; k >= 0; --k) acc = FMA(-AT(sub[i], lds[i], k, r), AT(Mn, ldm, k, c), acc);
            AT(Mi, ldm, r, c) = acc;
          }
        trsm_left_lower_t(ni, diag[i], ldd[i], Mi, ldm, nrhs);
      }
    }
    l = n;
    li = ni;
    n -= ni;
  }
}
