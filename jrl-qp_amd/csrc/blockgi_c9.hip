// blockgi_c9.hip -- the structured Goldfarb-Idnani inner loop on gfx950 under contract C9: no orthonormal sequence.
//
//   reference (jrl-umi3218/jrl-qp):
//     experimental::BlockGISolver      src/experimental/BlockGISolver.cpp:62-288, :446-483   hooks + init
//     StructuredG / StructuredC        src/structured/StructuredG.cpp:45-113, StructuredC.cpp:59-77
//     StructuredJ / StructuredQR       src/structured/StructuredJ.cpp:32-57, src/structured/StructuredQR.cpp:64-102
//     generic loop                     src/DualSolver.cpp:91-168, :231-244, :38-69
//   oracle: oracle/blockgi_oracle.c orc_blockgi_solve_c9 (bit-identical results; the contract is stated there and in DESIGN.md 7.3)
//
// The reference keeps Q of  L^-1 N = Q [R; 0]  as a sequence of Householder / Givens elements and applies every element twice per
// iteration (O(n k) per iteration, k = elements ever made: 7.9 GB of reads per solve at BASELINE config c5).  What an iteration needs
// is the least-squares problem  min |b - B r|,  b = L^-1 n+,  B = L^-1 N:  r are the multipliers' direction, y = b - B r = Q2 Q2^T b is
// what z = L^-T y is made of.  B is SPARSE (a column is non-zero in the Hessian blocks under its constraint's window and in the head),
// so this kernel keeps the columns of B and S = R^-1 (q x q, upper triangular) and solves by the corrected semi-normal equations:
//     r = S S^T B^T b,  y = b - B r,   then once more on the residual:  r += S S^T B^T y,  y -= B (S S^T B^T y).
// One iteration is four triangular matrix-vector products (q^2 / 2 fmas each, every lane its own chain) plus O(q) sparse columns;
// nothing of size n x q exists.  add: S gets the column [-r / rho; 1 / rho], rho = |y|; remove: the rotations that fold row l of S into
// the last column come from a prefix sum of squares of that row (CANON C7 on the row reversed: no serial generator) and every row of
// S takes them as its own chain.
//
// One workgroup per QP, the whole state machine on the device, one launch per batch (as blockgi.hip).  LDS: b / y / z / x (n each),
// u, r and two q-vectors, the position -> slot map, the per-group lists of active slots; HBM (L2-resident for one problem): the
// columns of B by slot, S twice (by columns for r = S d1, by rows for d1 = S^T g: each product reads consecutive addresses across
// the lanes), L and C where the caller put them.
#include "blockgi.h"

#include "device_common.h"

namespace jrlqp
{
namespace
{
#include "blockgi_common.h"

struct C9
{
  double *Bc, *Sm, *St;
  int ldq;       // leading dimension of Sm / St (= qcap)
  double *ga, *gb; // two q-vectors (LDS)
  int * slot;      // position -> slot (LDS)
  int * roff;      // position -> 8 * slot * ldq: the row's BYTE offset in St (LDS; read as unsigned)
  unsigned * smeta; // slot -> first | (last + 1) << 16 off-head block of the column's reach (LDS)
  int *glist, *gcount, *goff; // per-group lists of active slots in active-set order (LDS)
  unsigned short * kapr;      // permuted row -> block-order index (LDS)
  unsigned char * lstat;      // activation status of every constraint (LDS copy of the output array: the scan reads it every pass)
  int *koff, *klen, *kpho;    // block-order index -> first permuted row, rows, first ORIGINAL row (LDS)
  double * bsum;              // pow2(nblk) doubles (LDS): block sums of a C9 reduction
  double * hred;              // 8 x 128 doubles (LDS): the waves' subtree roots of the head rows (c9_y_update)
  const int4 * tkl;           // block-order index -> ranges of groups that reach the block (LDS copy of BgiArgs::tk)
  int *sgrp, *freel;          // slot -> group, free-slot stack (HBM)
  int nfree;
};

// ---- C9 reduction over a whole vector: per block the adjacent-first pairwise tree of the products, then the same tree over the blocks
// (oracle blk_reduce).  fa / fb take the index in the vector's own coordinates; `perm`: the vector is in permuted coordinates.
template<int NT, bool BIG, class FA, class FB>
__device__ __forceinline__ double blk_reduce(Ctx & K, C9 & Q, bool perm, FA fa, FB fb)
{
  constexpr int NW = NT / 64;
  const int nblk = K.A->nblk, lane = K.lane;
  int P = 1;
  while(P < nblk) P <<= 1;
  if(P < 64) P = 64;
  for(int k = nblk + K.tid; k < P; k += NT) Q.bsum[k] = 0.0;
  for(int k = K.wave; k < nblk; k += NW)
  {
    const int len = Q.klen[k];
    const int o = perm ? Q.koff[k] : Q.kpho[k];
    double p = (lane < len) ? FMA(fa(o + lane), fb(o + lane), 0.0) : 0.0;
    double s = wave_sum_tree(p);
    if(BIG && len > 64) // (uniform) rows 64 .. 127: the second half of the 128-slot tree
    {
      const double p2 = (lane + 64 < len) ? FMA(fa(o + lane + 64), fb(o + lane + 64), 0.0) : 0.0;
      s = s + wave_sum_tree(p2);
    }
    if(lane == 0) Q.bsum[k] = s;
  }
  __syncthreads();
  // tree over the blocks (wave 0): lane l folds its aligned chunk of P / 64 slots in place, then the wave tree
  double * slot = K.red + 16 * K.par;
  K.par ^= 1;
  if(K.wave == 0)
  {
    const int chunk = P >> 6, base = lane * chunk;
    for(int h = 1; h < chunk; h <<= 1)
      for(int i = 0; i < chunk; i += 2 * h) Q.bsum[base + i] = Q.bsum[base + i] + Q.bsum[base + i + h];
    const double tot = wave_sum_tree(Q.bsum[base]);
    if(lane == 0) slot[0] = tot;
  }
  __syncthreads();
  return slot[0];
}

// ---- g_k = B_k . v for every active position k (v in LDS, permuted coordinates); pass 0: v = b, only the blocks b reaches ----------
// sum_i col[o + i] v[o + i], i < len, an fma chain from +0; the column entries of a group of 16 rows are requested together
__device__ __forceinline__ double c9_col_dot(const double * col, const double * v, int o, int len)
{
  double acc = 0.0;
  for(int i0 = 0; i0 < len; i0 += 16)
  {
    double cv[16];
#pragma unroll
    for(int u = 0; u < 16; ++u) cv[u] = col[o + min(i0 + u, len - 1)];
#pragma unroll
    for(int u = 0; u < 16; ++u)
    {
      const double t = FMA(cv[u], v[o + min(i0 + u, len - 1)], acc);
      acc = (i0 + u < len) ? t : acc;
    }
  }
  return acc;
}
template<int NT>
__device__ __forceinline__ void c9_bt_v(Ctx & K, C9 & Q, int q, const double * v, bool lim, int plo, int phi, double * out)
{
  const int nblk = K.A->nblk, np = K.A->n_pad;
  const bool arrow = K.A->kind != 2;
  const int oh = Q.koff[nblk - 1], lh = Q.klen[nblk - 1];
  for(int k = K.tid; k < q; k += NT)
  {
    const int s = Q.slot[k];
    const unsigned me = Q.smeta[s];
    int lo = (int)(me & 0xffffu), hi = (int)(me >> 16) - 1; // (last + 1 is stored: an empty range has last = -1)
    if(lim)
    {
      lo = max(lo, plo);
      hi = min(hi, phi);
    }
    const double * col = Q.Bc + (size_t)s * np;
    double acc = 0.0;
    if(arrow && lo == hi && lh <= 32 && Q.klen[lo] <= 32)
    {
      // the usual arrow column: one off-head block and the head, 2 x 16 entries requested together, twice
      const int o = Q.koff[lo], len = Q.klen[lo];
      double va = 0.0, vh = 0.0;
#pragma unroll
      for(int h0 = 0; h0 < 32; h0 += 16)
      {
        double ca[16], ch[16];
#pragma unroll
        for(int u = 0; u < 16; ++u) ca[u] = col[o + min(h0 + u, len - 1)];
#pragma unroll
        for(int u = 0; u < 16; ++u) ch[u] = col[oh + min(h0 + u, lh - 1)];
#pragma unroll
        for(int u = 0; u < 16; ++u)
        {
          const double t = FMA(ca[u], v[o + min(h0 + u, len - 1)], va);
          va = (h0 + u < len) ? t : va;
        }
#pragma unroll
        for(int u = 0; u < 16; ++u)
        {
          const double t = FMA(ch[u], v[oh + min(h0 + u, lh - 1)], vh);
          vh = (h0 + u < lh) ? t : vh;
        }
      }
      acc = va + vh;
    }
    else
    {
      bool first = true;
      for(int kap = lo; kap <= hi; ++kap)
      {
        const double val = c9_col_dot(col, v, Q.koff[kap], Q.klen[kap]);
        acc = first ? val : acc + val;
        first = false;
      }
      if(arrow)
      {
        const double val = c9_col_dot(col, v, oh, lh);
        acc = first ? val : acc + val;
      }
    }
    out[k] = acc;
  }
  __syncthreads();
}

// ---- d1 = S^T g: column j an fma chain over the rows (positions) ascending; lanes = consecutive columns of St -----------------------
// One wave per 64 columns.  Priced in VALU instructions per term (a wave64 fp64 instruction occupies its SIMD for four cycles): above
// the chunk's diagonal block every lane takes every term -- one add for the address (row offsets by position staged in LDS: `roff`),
// one fma --, only the diagonal block selects.  kPf loads are in flight while the previous kPf are consumed.
constexpr int kPf = 16;
template<int NT>
__device__ __forceinline__ void c9_st_g(Ctx & K, C9 & Q, int q, const double * g, double * d1)
{
  const int lane = K.lane;
  const char * sbase = reinterpret_cast<const char *>(Q.St);
  // row offsets in BYTES by position (uniform base + 32-bit offset: one add per term); the array is zero beyond position q - 1 (row 0
  // of St: always there), so a request that runs past the last row needs no clamp and the offsets are read four at a time
  const uint4 * roff4 = reinterpret_cast<const uint4 *>(Q.roff);
  auto ld = [&](unsigned off) { return *reinterpret_cast<const double *>(sbase + off); };
  auto request = [&](int i0, unsigned lo, double(&a)[kPf]) { // rows i0 .. i0 + kPf - 1 (i0 a multiple of 4)
    uint4 o[kPf / 4];
#pragma unroll
    for(int v = 0; v < kPf / 4; ++v) o[v] = roff4[i0 / 4 + v];
#pragma unroll
    for(int v = 0; v < kPf / 4; ++v)
    {
      a[4 * v + 0] = ld(o[v].x + lo);
      a[4 * v + 1] = ld(o[v].y + lo);
      a[4 * v + 2] = ld(o[v].z + lo);
      a[4 * v + 3] = ld(o[v].w + lo);
    }
  };
  auto operands = [&](int i0, double(&gv)[kPf]) { // g[i0 .. i0 + kPf - 1] (i0 even: two per LDS read)
    const double2 * g2 = reinterpret_cast<const double2 *>(g + i0);
#pragma unroll
    for(int v = 0; v < kPf / 2; ++v)
    {
      const double2 t = g2[v];
      gv[2 * v] = t.x;
      gv[2 * v + 1] = t.y;
    }
  };
  // A wave keeps 4 kPf = 64 loads in flight: all of a group of 64 rows are requested, then consumed in order (the wait counts of
  // straight-line code are exact; a load carried over the loop's back edge makes hipcc wait for EVERYTHING in flight at its first use).
  for(int c0 = 64 * K.wave; c0 < q; c0 += NT)
  {
    const int j = c0 + lane, jc = min(j, q - 1);
    const unsigned lo = 8u * (unsigned)jc;
    double acc = 0.0;
    double a0[kPf], a1[kPf], a2[kPf], a3[kPf], gv[kPf];
    // rows 0 .. c0-1 (c0 is a multiple of 64): no select
    for(int i0 = 0; i0 < c0; i0 += 4 * kPf)
    {
      request(i0, lo, a0);
      request(i0 + kPf, lo, a1);
      request(i0 + 2 * kPf, lo, a2);
      request(i0 + 3 * kPf, lo, a3);
      operands(i0, gv);
#pragma unroll
      for(int u = 0; u < kPf; ++u) acc = FMA(a0[u], gv[u], acc);
      operands(i0 + kPf, gv);
#pragma unroll
      for(int u = 0; u < kPf; ++u) acc = FMA(a1[u], gv[u], acc);
      operands(i0 + 2 * kPf, gv);
#pragma unroll
      for(int u = 0; u < kPf; ++u) acc = FMA(a2[u], gv[u], acc);
      operands(i0 + 3 * kPf, gv);
#pragma unroll
      for(int u = 0; u < kPf; ++u) acc = FMA(a3[u], gv[u], acc);
    }
    // the diagonal block: rows c0 .. c0 + 63, lane j takes rows <= j (rows beyond q - 1 read row 0 and are dropped by the select)
    {
      request(c0, lo, a0);
      request(c0 + kPf, lo, a1);
      request(c0 + 2 * kPf, lo, a2);
      request(c0 + 3 * kPf, lo, a3);
      operands(c0, gv);
#pragma unroll
      for(int u = 0; u < kPf; ++u)
      {
        const double t = FMA(a0[u], gv[u], acc);
        acc = (c0 + u <= jc) ? t : acc;
      }
      operands(c0 + kPf, gv);
#pragma unroll
      for(int u = 0; u < kPf; ++u)
      {
        const double t = FMA(a1[u], gv[u], acc);
        acc = (c0 + kPf + u <= jc) ? t : acc;
      }
      operands(c0 + 2 * kPf, gv);
#pragma unroll
      for(int u = 0; u < kPf; ++u)
      {
        const double t = FMA(a2[u], gv[u], acc);
        acc = (c0 + 2 * kPf + u <= jc) ? t : acc;
      }
      operands(c0 + 3 * kPf, gv);
#pragma unroll
      for(int u = 0; u < kPf; ++u)
      {
        const double t = FMA(a3[u], gv[u], acc);
        acc = (c0 + 3 * kPf + u <= jc) ? t : acc;
      }
    }
    if(j < q) d1[j] = acc;
  }
  __syncthreads();
}

// ---- dl = S d1: row i an fma chain over the columns DESCENDING; lanes = consecutive positions (rows of Sm by slot) -------------------
template<int NT>
__device__ __forceinline__ void c9_s_d(Ctx & K, C9 & Q, int q, const double * d1, double * dl)
{
  const int lane = K.lane;
  const size_t cstride = (size_t)Q.ldq * 8; // bytes per column
  for(int r0 = 64 * K.wave; r0 < q; r0 += NT)
  {
    const int i = r0 + lane, ic = min(i, q - 1);
    const unsigned ro = 8u * (unsigned)Q.slot[ic]; // the lane's row: a 32-bit offset against the (uniform) column base
    const char * sbase = reinterpret_cast<const char *>(Q.Sm);
    auto ld = [&](int j) { return *reinterpret_cast<const double *>(sbase + (size_t)j * cstride + ro); };
    double acc = 0.0;
    double a0[kPf], a1[kPf], a2[kPf], a3[kPf];
    // columns q-1 .. down to the last whole group of 64 above the chunk's diagonal block: every lane takes every term
    const int nfull = max(q - (r0 + 64), 0) / (4 * kPf) * (4 * kPf);
    int j0 = q - 1;
    for(int done = 0; done < nfull; done += 4 * kPf, j0 -= 4 * kPf)
    {
      double dv[kPf];
#pragma unroll
      for(int u = 0; u < kPf; ++u) a0[u] = ld(j0 - u);
#pragma unroll
      for(int u = 0; u < kPf; ++u) a1[u] = ld(j0 - kPf - u);
#pragma unroll
      for(int u = 0; u < kPf; ++u) a2[u] = ld(j0 - 2 * kPf - u);
#pragma unroll
      for(int u = 0; u < kPf; ++u) a3[u] = ld(j0 - 3 * kPf - u);
#pragma unroll
      for(int u = 0; u < kPf; ++u) dv[u] = d1[j0 - u];
#pragma unroll
      for(int u = 0; u < kPf; ++u) acc = FMA(a0[u], dv[u], acc);
#pragma unroll
      for(int u = 0; u < kPf; ++u) dv[u] = d1[j0 - kPf - u];
#pragma unroll
      for(int u = 0; u < kPf; ++u) acc = FMA(a1[u], dv[u], acc);
#pragma unroll
      for(int u = 0; u < kPf; ++u) dv[u] = d1[j0 - 2 * kPf - u];
#pragma unroll
      for(int u = 0; u < kPf; ++u) acc = FMA(a2[u], dv[u], acc);
#pragma unroll
      for(int u = 0; u < kPf; ++u) dv[u] = d1[j0 - 3 * kPf - u];
#pragma unroll
      for(int u = 0; u < kPf; ++u) acc = FMA(a3[u], dv[u], acc);
    }
    // the rest, down to the chunk's first row (fewer than 128 columns): lane i takes columns >= i
    for(; j0 >= r0; j0 -= 2 * kPf)
    {
#pragma unroll
      for(int u = 0; u < kPf; ++u) a0[u] = ld(max(j0 - u, r0));
#pragma unroll
      for(int u = 0; u < kPf; ++u) a1[u] = ld(max(j0 - kPf - u, r0));
#pragma unroll
      for(int u = 0; u < kPf; ++u)
      {
        const double t = FMA(a0[u], d1[max(j0 - u, r0)], acc);
        acc = (j0 - u >= ic) ? t : acc;
      }
#pragma unroll
      for(int u = 0; u < kPf; ++u)
      {
        const double t = FMA(a1[u], d1[max(j0 - kPf - u, r0)], acc);
        acc = (j0 - kPf - u >= ic) ? t : acc;
      }
    }
    if(i < q) dl[i] = acc;
  }
  __syncthreads();
}

// ---- y -= B dl.  Per group that reaches a row's block an fma chain over the group's active slots in active-set order, the group
// values combined by the adjacent-first pairwise tree over the static list of groups (general range, then bound range).  dls: dl by
// SLOT.  The tree is folded as the groups come (binary-counter stack: static register indices).
struct TreeStack
{
  double st[13];
  int cnt;
  __device__ __forceinline__ void reset()
  {
#pragma unroll
    for(int L = 0; L < 13; ++L) st[L] = 0.0;
    cnt = 0;
  }
  __device__ __forceinline__ void push(double v) // fold with the completed subtrees of equal size
  {
    bool placed = false;
#pragma unroll
    for(int L = 0; L < 13; ++L)
    {
      if(!placed)
      {
        if((cnt >> L) & 1)
          v = st[L] + v;
        else
        {
          st[L] = v;
          placed = true;
        }
      }
    }
    ++cnt;
  }
  __device__ __forceinline__ double flush() const // the pending subtrees, smallest first (padding with +0 changes no bit)
  {
    double tot = 0.0;
    bool have = false;
#pragma unroll
    for(int L = 0; L < 13; ++L)
    {
      if((cnt >> L) & 1)
      {
        tot = have ? st[L] + tot : st[L];
        have = true;
      }
    }
    return tot;
  }
};

// entries e0 .. of group gI for one row: acc <- fma(dl, B, acc) in list order
__device__ __forceinline__ double c9_group_chain(const C9 & Q, const double * dls, const double * colrow, int np, int gI, int e0, double acc)
{
  const int c = Q.gcount[gI], o = Q.goff[gI];
  for(; e0 < c; e0 += 4)
  {
    double bv[4], dv[4];
#pragma unroll
    for(int u = 0; u < 4; ++u)
    {
      const int s = Q.glist[o + min(e0 + u, c - 1)];
      dv[u] = dls[s];
      bv[u] = colrow[(size_t)s * np];
    }
#pragma unroll
    for(int u = 0; u < 4; ++u)
    {
      const double tt = FMA(dv[u], bv[u], acc);
      acc = (e0 + u < c) ? tt : acc;
    }
  }
  return acc;
}

template<int NT, bool BIG>
__device__ __forceinline__ void c9_y_update(Ctx & K, C9 & Q, double * y, const double * dls)
{
  constexpr int NW = NT / 64, RB = 4, EB = 8;
  const int n = K.A->n, np = K.A->n_pad, nblk = K.A->nblk, qc = K.A->qcap;
  const bool arrow = K.A->kind != 2;
  const int oh = Q.koff[nblk - 1], lh = Q.klen[nblk - 1];
  const int nrow = arrow ? n - lh : n; // (the head is the last block in permuted coordinates)
  // ---- A: every row but an arrow's head: thread per row, RB rows at a time; the first EB entries of each row's first group are
  // requested for all RB rows before any is consumed (the usual row has one group of a few entries: one memory round trip per RB rows)
  for(int i0 = K.tid; i0 < nrow; i0 += RB * NT)
  {
    double bv[RB][EB];
    int sv[RB][EB], g0[RB], c0[RB];
#pragma unroll
    for(int rr = 0; rr < RB; ++rr)
    {
      const int i = min(i0 + rr * NT, nrow - 1);
      const int4 t = Q.tkl[Q.kapr[i]];
      g0[rr] = (t.x <= t.y) ? t.x : t.z; // (a block no group reaches: t.z > t.w too -- c0 = 0 below)
      const bool any = (t.x <= t.y) || (t.z <= t.w);
      c0[rr] = any ? Q.gcount[g0[rr]] : 0;
      const int o = any ? Q.goff[g0[rr]] : 0;
#pragma unroll
      for(int u = 0; u < EB; ++u)
      {
        int sl = Q.glist[o + min(u, max(c0[rr] - 1, 0))];
        sl = (u < c0[rr]) ? sl : 0; // (an unused entry may hold anything: slot 0's storage is always there)
        sv[rr][u] = sl;
        bv[rr][u] = Q.Bc[(size_t)sl * np + i];
      }
    }
#pragma unroll
    for(int rr = 0; rr < RB; ++rr)
    {
      const int i = i0 + rr * NT;
      if(i < nrow)
      {
        const int4 t = Q.tkl[Q.kapr[i]];
        const double * colrow = Q.Bc + i;
        TreeStack T;
        T.reset();
        bool firstg = true;
        auto group = [&](int gI) {
          double acc = 0.0;
          int e0 = 0;
          if(firstg)
          {
#pragma unroll
            for(int u = 0; u < EB; ++u)
            {
              const double tt = FMA(dls[sv[rr][u]], bv[rr][u], acc);
              acc = (u < c0[rr]) ? tt : acc;
            }
            e0 = EB;
            firstg = false;
          }
          acc = c9_group_chain(Q, dls, colrow, np, gI, e0, acc);
          T.push(acc);
        };
        for(int gI = t.x; gI <= t.y; ++gI) group(gI);
        for(int gI = t.z; gI <= t.w; ++gI) group(gI);
        y[i] = y[i] - T.flush();
      }
    }
  }
  // ---- B: the head of an arrow -- every group reaches it.  The waves share the (padded) list of groups in aligned chunks: each folds
  // its chunk into a subtree, the NW subtree roots meet in LDS and one wave finishes the tree.
  if(arrow && !BIG && lh <= 32)
  {
    // head of at most 32 rows: lanes 0 .. 31 take the even slot of a pair of groups, lanes 32 .. 63 the odd one (the pair's sum is the
    // first level of the tree); the entries of the NEXT pair are requested before this pair's are consumed
    const int4 t = Q.tkl[nblk - 1];
    const int ng = max(t.y - t.x + 1, 0), nT = ng + max(t.w - t.z + 1, 0);
    int P = 1;
    while(P < nT) P <<= 1;
    if(P < 2 * NW) P = 2 * NW;
    const int G = P / NW, lane = K.lane, hb = lane >> 5, l5 = lane & 31;
    const double * rowp = Q.Bc + oh + min(l5, lh - 1);
    auto group_of = [&](int ts) { return (ts < nT) ? ((ts < ng) ? t.x + ts : t.z + (ts - ng)) : -1; };
    auto issue = [&](int ts, double(&bv)[EB], int(&sl)[EB], int & c) {
      const int gI = group_of(ts);
      c = (gI >= 0) ? Q.gcount[gI] : 0;
      const int o = (gI >= 0) ? Q.goff[gI] : 0;
#pragma unroll
      for(int u = 0; u < EB; ++u)
      {
        int sv = Q.glist[o + min(u, max(c - 1, 0))];
        sv = (u < c) ? sv : 0;
        sl[u] = sv;
        bv[u] = rowp[(size_t)sv * np];
      }
    };
    TreeStack T;
    T.reset();
    auto consume = [&](int ts, const double(&bv)[EB], const int(&sl)[EB], int c) {
      double acc = 0.0;
#pragma unroll
      for(int u = 0; u < EB; ++u)
      {
        const double tt = FMA(dls[sl[u]], bv[u], acc);
        acc = (u < c) ? tt : acc;
      }
      const int gI = group_of(ts);
      if(gI >= 0 && c > EB) acc = c9_group_chain(Q, dls, rowp, np, gI, EB, acc);
      T.push(acc + shfl_xor_d(acc, 32)); // (the same sum in both halves: a + b == b + a)
    };
    const int ts0 = K.wave * G + hb, npair = G / 2;
    // four pairs per trip: their entries are all requested, then consumed in order (no load is carried over the loop's back edge:
    // hipcc would wait for everything in flight at its first use)
    for(int r = 0; r < npair; r += 4)
    {
      double b0[EB], b1[EB], b2[EB], b3[EB];
      int s0[EB], s1[EB], s2[EB], s3[EB], c0, c1, c2, c3;
      issue(ts0 + 2 * r, b0, s0, c0);
      issue(ts0 + 2 * (r + 1), b1, s1, c1); // (a slot beyond the padded list has no group: no entries)
      issue(ts0 + 2 * (r + 2), b2, s2, c2);
      issue(ts0 + 2 * (r + 3), b3, s3, c3);
      consume(ts0 + 2 * r, b0, s0, c0);
      if(r + 1 < npair) consume(ts0 + 2 * (r + 1), b1, s1, c1); // (uniform)
      if(r + 2 < npair) consume(ts0 + 2 * (r + 2), b2, s2, c2);
      if(r + 3 < npair) consume(ts0 + 2 * (r + 3), b3, s3, c3);
    }
    __syncthreads(); // (part A's writes of y are done with; hred is free)
    if(hb == 0) Q.hred[K.wave * 128 + l5] = T.flush();
    __syncthreads();
    for(int r = K.tid; r < lh; r += NT)
    {
      double w[NW];
#pragma unroll
      for(int i = 0; i < NW; ++i) w[i] = Q.hred[i * 128 + r];
#pragma unroll
      for(int st = 1; st < NW; st <<= 1)
#pragma unroll
        for(int i = 0; i < NW; i += 2 * st) w[i] = w[i] + w[i + st];
      y[oh + r] = y[oh + r] - w[0];
    }
  }
  else if(arrow)
  {
    const int4 t = Q.tkl[nblk - 1];
    const int ng = max(t.y - t.x + 1, 0), nT = ng + max(t.w - t.z + 1, 0);
    int P = 1;
    while(P < nT) P <<= 1;
    if(P < NW) P = NW;
    const int G = P / NW, lane = K.lane;
    const int r0 = lane, r1 = lane + 64;
    const double * row0 = Q.Bc + oh + min(r0, lh - 1);
    const double * row1 = Q.Bc + oh + min(r1, lh - 1);
    TreeStack T0, T1;
    T0.reset();
    T1.reset();
    for(int ts = K.wave * G; ts < (K.wave + 1) * G; ++ts)
    {
      double a0 = 0.0, a1 = 0.0;
      if(ts < nT) // (uniform)
      {
        const int gI = (ts < ng) ? t.x + ts : t.z + (ts - ng);
        const int c = Q.gcount[gI], o = Q.goff[gI];
        for(int e0 = 0; e0 < c; e0 += EB)
        {
          double b0[EB], b1[EB];
          int sl[EB];
#pragma unroll
          for(int u = 0; u < EB; ++u)
          {
            sl[u] = Q.glist[o + min(e0 + u, c - 1)];
            b0[u] = row0[(size_t)sl[u] * np];
            if(BIG) b1[u] = row1[(size_t)sl[u] * np];
          }
#pragma unroll
          for(int u = 0; u < EB; ++u)
          {
            const double dv = dls[sl[u]];
            const double t0 = FMA(dv, b0[u], a0);
            a0 = (e0 + u < c) ? t0 : a0;
            if(BIG)
            {
              const double t1 = FMA(dv, b1[u], a1);
              a1 = (e0 + u < c) ? t1 : a1;
            }
          }
        }
      }
      T0.push(a0);
      if(BIG) T1.push(a1);
    }
    __syncthreads(); // (part A's writes of y are done with; hred is free)
    Q.hred[K.wave * 128 + r0] = T0.flush();
    if(BIG) Q.hred[K.wave * 128 + r1] = T1.flush();
    __syncthreads();
    for(int r = K.tid; r < lh; r += NT)
    {
      double w[NW];
#pragma unroll
      for(int i = 0; i < NW; ++i) w[i] = Q.hred[i * 128 + r];
#pragma unroll
      for(int st = 1; st < NW; st <<= 1)
#pragma unroll
        for(int i = 0; i < NW; i += 2 * st) w[i] = w[i] + w[i + st];
      y[oh + r] = y[oh + r] - w[0];
    }
  }
  (void)qc;
  __syncthreads();
}

// =========================================================================================================
// the kernel
// =========================================================================================================
template<int NT, bool BIG>
__global__ __launch_bounds__(NT) void blockgi_c9_kernel(const BgiArgs A)
{
  extern __shared__ double sm[];
  const int pb = blockIdx.x, tid = threadIdx.x;
  const int n = A.n, m = A.m, nb = A.nb, nblk = A.nblk;
  Ctx K;
  C9 Q;
  K.A = &A;
  K.tid = tid;
  K.lane = tid & 63;
  K.wave = tid >> 6;
  K.diag = A.diag + (size_t)pb * A.nblk;
  K.side = A.side + (size_t)pb * (A.nblk > 1 ? A.nblk - 1 : 0);
  K.cblk = A.cblk + (size_t)pb * A.ncb;
  // LDS: d | z | [x | u] | r | ga | gb | bsum | hred | tkl | red (32) | ired (32 doubles) | sh (8) | ints: slot, roff, smeta, gcount, goff, glist, koff, klen, kpho | kapr | bm
  double * p = sm;
  K.d = p; p += A.n_pad;
  K.z = p; p += A.n_pad;
  if(A.xu_global)
  {
    K.x = A.xu_ws + (size_t)pb * (A.n_pad + A.qcap);
    K.u = K.x + A.n_pad;
  }
  else
  {
    K.x = p; p += A.n_pad;
    K.u = p; p += A.qcap;
  }
  K.r = p; p += A.qcap;
  Q.ga = p; p += A.qcap + 64; // (+ 64: operands are read in groups that may run past the last position)
  Q.gb = p; p += A.qcap + 64;
  int P2 = 1;
  while(P2 < nblk) P2 <<= 1;
  if(P2 < 64) P2 = 64;
  Q.bsum = p; p += P2;
  Q.hred = p; p += 8 * 128;
  {
    int4 * tl = reinterpret_cast<int4 *>(p); p += 2 * nblk;
    for(int k = tid; k < nblk; k += NT) tl[k] = reinterpret_cast<const int4 *>(A.tk)[k];
    Q.tkl = tl;
  }
  K.red = p; p += 32;
  K.ired = reinterpret_cast<int *>(p); p += 32;
  double * sh = p; p += 8;
  K.tri = nullptr;
  K.par = 0;
  int * ip = reinterpret_cast<int *>(p);
  Q.slot = ip; ip += A.qcap;
  Q.roff = ip; ip += A.qcap + 64;
  Q.smeta = reinterpret_cast<unsigned *>(ip); ip += A.qcap;
  Q.gcount = ip; ip += A.ngrp;
  Q.goff = ip; ip += A.ngrp;
  Q.glist = ip; ip += (m + nb + 1);
  Q.koff = ip; ip += nblk;
  Q.klen = ip; ip += nblk;
  Q.kpho = ip; ip += nblk;
  Q.kapr = reinterpret_cast<unsigned short *>(ip); ip += (n + 2) / 2;
  Q.lstat = reinterpret_cast<unsigned char *>(ip); ip += (m + nb + 4) / 4;
  ip += (reinterpret_cast<size_t>(ip) & 7) ? 1 : 0; // (8-byte alignment of the block table)
  {
    BlkMeta * bm = reinterpret_cast<BlkMeta *>(ip);
    fill_block_table<NT>(A, pb, tid, K.diag, K.side, bm);
    K.bm = bm;
  }
  for(int k = tid; k < nblk; k += NT)
  {
    Q.koff[k] = A.off_perm[k];
    Q.klen[k] = A.bs[A.kap_phys[k]];
    Q.kpho[k] = A.boff[A.kap_phys[k]];
  }
  for(int i = tid; i < n; i += NT) Q.kapr[i] = (unsigned short)A.kap_prow[i];
  for(int g = tid; g < A.ngrp; g += NT)
  {
    Q.gcount[g] = 0;
    Q.goff[g] = A.goff[g];
  }
  K.pool = nullptr;
  K.Rp = nullptr;
  K.desc = nullptr;
  K.active = A.active + (size_t)pb * A.qcap;
  K.status = A.status + (size_t)pb * (m + nb);
  Q.ldq = A.qcap;
  Q.Bc = A.Bc + (size_t)pb * A.qcap * A.n_pad;
  Q.Sm = A.Sm + (size_t)pb * A.qcap * A.qcap;
  Q.St = A.St + (size_t)pb * A.qcap * A.qcap;
  Q.sgrp = A.sgrp + (size_t)pb * A.qcap;
  Q.freel = A.freel + (size_t)pb * A.qcap;
  Q.nfree = A.qcap;
  double *d = K.d, *z = K.z, *x = K.x, *u = K.u, *r = K.r;
  const double * a = A.a + (size_t)pb * n;
  const double *bl = A.bl + (size_t)pb * m, *bu = A.bu + (size_t)pb * m;
  const double *xl = nb ? A.xl + (size_t)pb * nb : nullptr, *xu = nb ? A.xu + (size_t)pb * nb : nullptr;
  jrlqp_trace_rec * trace = A.trace ? A.trace + (size_t)pb * A.trace_cap : nullptr;

  int q = 0, it = 0, rc = JRLQP_UNKNOWN;
  double f = 0.0;
  // diagnostic phase timer (thread 0, 100 MHz wall clock): 0 init, 1 scan, 2 L^-1 n+, 3 B^T v, 6 S^T g, 11 S d1, 12 r += dl, 4 y -= B dl, 5 L^-T y + |y|,
  // 7 step length, 8 update, 9 add, 10 drop
  long long pacc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tp = (A.prof && tid == 0) ? wall_clock64() : 0;
  const long long cyc0 = clock64(), wall0 = tp;
  auto stamp = [&](int slot) {
    if(A.prof && tid == 0)
    {
      const long long t = wall_clock64();
      pacc[slot] += t - tp;
      tp = t;
    }
  };

  for(int i = tid; i < m + nb; i += NT)
  {
    K.status[i] = JRLQP_INACTIVE;
    Q.lstat[i] = JRLQP_INACTIVE;
  }
  for(int i = tid; i < n; i += NT) x[i] = 0.0;
  for(int i = tid; i < A.qcap + 64; i += NT) Q.roff[i] = 0;
  for(int i = tid; i < A.qcap; i += NT)
  {
    u[i] = r[i] = 0.0;
    Q.ga[i] = Q.gb[i] = 0.0; // (operands are read in groups that may run past q: never uninitialised)
    if(i < 64) Q.ga[A.qcap + i] = Q.gb[A.qcap + i] = 0.0;
    Q.freel[i] = A.qcap - 1 - i; // (slot 0 is handed out first)
  }
  __syncthreads();

  // ---- init_ (BlockGISolver.cpp:62-107) ----
  {
    int bad = 0;
    for(int i = tid; i < m; i += NT) bad |= (bl[i] == bu[i]);
    for(int i = tid; i < nb; i += NT) bad |= (xl[i] == xu[i]);
    if(__syncthreads_or(bad))
    {
      rc = JRLQP_INCONSISTENT_INPUT;
      goto done;
    }
  }
  if(A.fail[pb]) // lltInPlace (:71-73), factorised by the kernels of structured.hip
  {
    rc = JRLQP_NON_POS_HESSIAN;
    goto done;
  }
  // initializePrimalDualPoints (:476-481): x = -G^-1 a, f = a.x / 2
  for(int i = tid; i < n; i += NT) x[i] = a[i];
  __syncthreads();
  sg_lsolve<NT, BIG>(K, x, 0, n, false);
  sg_ltsolve<NT, BIG>(K, x, false);
  for(int i = tid; i < n; i += NT) x[i] = -x[i];
  __syncthreads();
  f = 0.5 * blk_reduce<NT, BIG>(K, Q, false, [&](int e) { return a[e]; }, [&](int e) { return x[e]; });
  stamp(0);

  // ---- DualSolver::solve loop (DualSolver.cpp:96-167) ----
  {
    bool skip_step1 = false;
    int p = -1, pst = JRLQP_INACTIVE;
    rc = JRLQP_MAX_ITER_REACHED;
    for(; it < A.max_iter; ++it)
    {
      if(!skip_step1)
      {
        // selectViolatedConstraint_ (BlockGISolver.cpp:109-162): thread j owns constraints j, j + NT, ... in order
        double bval = 0.0;
        int bidx = 0x7fffffff, btag = JRLQP_INACTIVE;
        for(int j = tid; j < m; j += NT)
        {
          if(Q.lstat[j] != JRLQP_INACTIVE) continue; // (the LDS copy of the status: no memory round trip in front of the chain)
          const int b = A.to_block[j], len = A.cvar[b];
          const double * c = K.cblk[b] + (size_t)(j - A.cum_cstr[b]) * A.cld[b];
          const double * w = x + A.cum_var[b];
          const double cx = chain_neg8(0, 1, len, 0.0, [&](int k) { return -c[k]; }, [&](int k) { return w[k]; });
          const double sl = cx - bl[j];
          if(sl < bval)
          {
            bval = sl;
            bidx = j;
            btag = JRLQP_LOWER;
          }
          else
          {
            const double su = bu[j] - cx;
            if(su < bval)
            {
              bval = su;
              bidx = j;
              btag = JRLQP_UPPER;
            }
          }
        }
        for(int i = tid; i < nb; i += NT)
        {
          if(Q.lstat[m + i] != JRLQP_INACTIVE) continue;
          const double sl = x[i] - xl[i];
          if(sl < bval)
          {
            bval = sl;
            bidx = m + i;
            btag = JRLQP_LOWER_BOUND;
          }
          else
          {
            const double su = xu[i] - x[i];
            if(su < bval)
            {
              bval = su;
              bidx = m + i;
              btag = JRLQP_UPPER_BOUND;
            }
          }
        }
        block_argmin<NT>(K, bval, bidx, btag);
        p = bidx;
        pst = btag;
        if(pst == JRLQP_INACTIVE)
        {
          rc = JRLQP_SUCCESS;
          break;
        }
        if(tid == 0) u[q] = 0.0;
      }
      stamp(1);

      // ---- computeStep_ under C9 ----
      // b = L^-1 n+  (StructuredJ.cpp:43-57); group and reach of the constraint
      int wst, wen, pgrp, plo, phi;
      for(int i = tid; i < n; i += NT) d[i] = 0.0;
      __syncthreads();
      if(p < m)
      {
        const int b = A.to_block[p], len = A.cvar[b];
        const double * c = K.cblk[b] + (size_t)(p - A.cum_cstr[b]) * A.cld[b];
        wst = A.cum_var[b];
        wen = wst + len;
        for(int i = tid; i < len; i += NT) d[wst + i] = c[i];
        pgrp = b;
        plo = A.greach[2 * b];
        phi = A.greach[2 * b + 1];
      }
      else
      {
        wst = p - m;
        wen = wst + 1;
        if(tid == 0) d[wst] = (pst == JRLQP_UPPER_BOUND) ? -1.0 : 1.0;
        const int kap = A.kap_row[wst];
        pgrp = A.ncb + kap;
        // reach of a single variable: its own block (tri: every block from it on); a head variable of an arrow reaches the head only
        if(A.kind == 2)
        {
          plo = kap;
          phi = nblk - 1;
        }
        else if(kap == nblk - 1)
        {
          plo = nblk;
          phi = -1;
        }
        else
          plo = phi = kap;
      }
      __syncthreads();
      sg_lsolve<NT, BIG>(K, d, wst, wen, true, (A.kind == 2) ? 0 : plo, (A.kind == 2) ? 0x7fffffff : phi);
      if(pst == JRLQP_UPPER)
      {
        for(int i = tid; i < n; i += NT) d[i] = -d[i];
        __syncthreads();
      }
      stamp(2);
      // y = b (in z); r = 0; corrected semi-normal equations: two passes of  g = B^T y, d1 = S^T g, dl = S d1, r += dl, y -= B dl
      for(int i = tid; i < n; i += NT) z[i] = d[i];
      __syncthreads();
      for(int pass = 0; pass < 2 && q > 0; ++pass)
      {
        c9_bt_v<NT>(K, Q, q, z, pass == 0, plo, phi, Q.ga);
        stamp(3);
        c9_st_g<NT>(K, Q, q, Q.ga, Q.gb);
        stamp(6);
        c9_s_d<NT>(K, Q, q, Q.gb, Q.ga);
        stamp(11);
        for(int k = tid; k < q; k += NT)
        {
          const double dl = Q.ga[k];
          r[k] = (pass == 0) ? dl : r[k] + dl;
          Q.gb[Q.slot[k]] = dl; // dl by slot (d1 is dead)
        }
        __syncthreads();
        stamp(12);
        if(q == n) // Q2 is empty: y = 0 by rule, nothing to refine against
        {
          for(int i = tid; i < n; i += NT) z[i] = 0.0;
          __syncthreads();
          break;
        }
        c9_y_update<NT, BIG>(K, Q, z, Q.gb);
        stamp(4);
      }
      double nrm2 = 0.0;
      if(q != n && !(A.tune & 2)) nrm2 = blk_reduce<NT, BIG>(K, Q, true, [&](int e) { return z[e]; }, [&](int e) { return z[e]; });
      if(A.tune & 2) nrm2 = 1.0;
      // z = L^-T y  (StructuredJ.cpp:32-41)
      sg_ltsolve<NT, BIG>(K, z);
      stamp(5);

      // ---- computeStepLength_ (BlockGISolver.cpp:174-242) ----
      double t1 = A.big_bnd, t2 = A.big_bnd;
      int l = 0;
      {
        double tv = A.big_bnd;
        int ti = 0x7fffffff, tt = 0;
        for(int k = tid; k < q; k += NT)
        {
          if(r[k] > 0.0)
          {
            const double tk = u[k] / r[k];
            if(tk < tv)
            {
              tv = tk;
              ti = k;
            }
          }
        }
        block_argmin<NT>(K, tv, ti, tt);
        if(ti != 0x7fffffff)
        {
          t1 = tv;
          l = ti;
        }
      }
      const double zz = blk_reduce<NT, BIG>(K, Q, false, [&](int e) { return z[e]; }, [&](int e) { return z[e]; });
      double cz = 0.0;
      if(__builtin_sqrt(zz) > 1e-14)
      {
        if(tid == 0) sh[0] = cstr_dot(K, p, z);
        if(tid == 64 % NT) sh[1] = cstr_dot(K, p, x);
        __syncthreads();
        cz = sh[0];
        const double cxp = sh[1];
        const double b = (pst == JRLQP_LOWER) ? bl[p] : (pst == JRLQP_UPPER) ? bu[p] : (pst == JRLQP_LOWER_BOUND) ? xl[p - m] : xu[p - m];
        t2 = (b - cxp) / cz;
        __syncthreads(); // sh is reused
      }
      const double t = (t2 < t1) ? t2 : t1;
      stamp(7);

      if(trace && it < A.trace_cap && tid == 0)
      {
        trace[it].p = p;
        trace[it].status = pst;
        trace[it].l = l;
        trace[it].t1 = t1;
        trace[it].t2 = t2;
        trace[it].kind = (t2 >= A.big_bnd) ? 2 : ((t == t2) ? 0 : 1);
      }

      if(t >= A.big_bnd)
      {
        rc = JRLQP_INFEASIBLE;
        break;
      }

      bool drop;
      if(t2 >= A.big_bnd)
      {
        for(int k = tid; k < q; k += NT) u[k] = FMA(-t, r[k], u[k]);
        if(tid == 0) u[q] += t;
        drop = true;
      }
      else
      {
        for(int i = tid; i < n; i += NT) x[i] = FMA(t, z[i], x[i]);
        const double nz = (pst == JRLQP_UPPER || pst == JRLQP_UPPER_BOUND) ? -cz : cz; // dot_ (:255-273)
        const double uq = u[q];
        f = FMA(t * nz, 0.5 * t + uq, f);
        __syncthreads(); // every thread has read u[q]
        for(int k = tid; k < q; k += NT) u[k] = FMA(-t, r[k], u[k]);
        if(tid == 0) u[q] += t;
        drop = !(t == t2);
      }
      __syncthreads();
      stamp(8);

      if(!drop)
      {
        // ---- add: S gets the column [-r / rho; 1 / rho], B the column b (in d); the constraint joins its group's list ----
        const double rho = __builtin_sqrt(nrm2);
        const int sn = Q.freel[Q.nfree - 1];
        --Q.nfree;
        for(int k = tid; k < q; k += NT)
        {
          const double v = -(r[k] / rho);
          const int s = Q.slot[k];
          Q.Sm[(size_t)s + (size_t)q * Q.ldq] = v;
          Q.St[(size_t)s * Q.ldq + q] = v;
        }
        {
          // the column of B: the rows b reaches (the rest of the slot's storage is never read)
          double * col = Q.Bc + (size_t)sn * A.n_pad;
          for(int kap = plo; kap <= phi; ++kap)
          {
            const int o = Q.koff[kap], len = Q.klen[kap];
            for(int i = tid; i < len; i += NT) col[o + i] = d[o + i];
          }
          if(A.kind != 2)
          {
            const int o = Q.koff[nblk - 1], len = Q.klen[nblk - 1];
            for(int i = tid; i < len; i += NT) col[o + i] = d[o + i];
          }
        }
        if(tid == 0)
        {
          const double v = 1.0 / rho;
          Q.Sm[(size_t)sn + (size_t)q * Q.ldq] = v;
          Q.St[(size_t)sn * Q.ldq + q] = v;
          Q.slot[q] = sn;
          Q.roff[q] = 8 * sn * Q.ldq;
          Q.smeta[sn] = (unsigned)plo | ((unsigned)(phi + 1) << 16);
          Q.sgrp[sn] = pgrp;
          Q.glist[Q.goff[pgrp] + Q.gcount[pgrp]] = sn;
          Q.gcount[pgrp] += 1;
          K.status[p] = pst;
          Q.lstat[p] = (unsigned char)pst;
          K.active[q] = p;
        }
        ++q;
        skip_step1 = false;
        __syncthreads();
        stamp(9);
      }
      else
      {
        // ---- removeConstraint (DualSolver.cpp:237-244): u, the active list; then S: fold row l into the last column ----
        const int removed = K.active[l];
        const int sl = Q.slot[l];
        const int len = q - l;
        // coefficients: CANON C7 (orc_givens_chain) on the row reversed: v_t = S(l, l + t); rotation tt (columns l+tt, l+tt+1) has
        // c = v_{tt+1} / rho_{tt+1}, s = -rho_tt / rho_{tt+1}, rho_0 = v_0, rho_j = sqrt(prefix sum of squares, blocks of 64)
        double *gc = Q.ga, *gs = Q.gb; // (both dead here)
        if(len > 1)
        {
          // stage the row in r (dead until the next computeStep_): r[t] = v_t
          for(int tq = tid; tq < len; tq += NT) r[tq] = Q.St[(size_t)sl * Q.ldq + l + tq];
          __syncthreads();
          if(K.wave == 0)
          {
            const int lane = K.lane;
            bool ok = true;
            for(int t0 = 0; t0 < len; t0 += 64)
            {
              const int tq = t0 + lane;
              const double v = (tq < len) ? r[tq] : 0.0;
              ok = ok && __all(tq >= len || c7_in_range(v));
            }
            if(ok)
            {
              double carry = 0.0, rho_last = 0.0; // carry of the prefix sum; rho of the previous block's last entry
              for(int t0 = 0; t0 < len; t0 += 64)
              {
                const int tq = t0 + lane;
                const double v = (tq < len) ? r[tq] : 0.0;
                double e = v * v;
#pragma unroll
                for(int off = 1; off < 64; off <<= 1)
                {
                  const int lo2 = __shfl_up(__double2loint(e), off), hi2 = __shfl_up(__double2hiint(e), off);
                  const double o = __hiloint2double(hi2, lo2);
                  e = (lane >= off) ? e + o : e;
                }
                const double Sj = e + carry;
                double rho = __builtin_sqrt(Sj);
                if(tq == 0) rho = v; // rho_0: the signed first entry
                double rprev;
                {
                  const int lo2 = __shfl_up(__double2loint(rho), 1), hi2 = __shfl_up(__double2hiint(rho), 1);
                  rprev = __hiloint2double(hi2, lo2);
                }
                if(lane == 0) rprev = rho_last;
                if(tq >= 1 && tq < len)
                {
                  const bool zr = (rho == 0.0);
                  gc[tq - 1] = zr ? 1.0 : v / rho;
                  gs[tq - 1] = zr ? 0.0 : (-rprev) / rho;
                }
                carry = bcast_lane(Sj, 63);
                rho_last = bcast_lane(rho, 63);
              }
            }
            else if(lane == 0)
            {
              // the serial recurrence (an entry beyond 2^+-500 or not finite): orc_givens(d'[i], r, ...) for i = len-2 .. 0, d'[i] = v_{len-1-i}
              double rr = r[0];
              for(int tt = 0; tt < len - 1; ++tt)
              {
                double c, s, r2;
                make_givens(r[tt + 1], rr, c, s, r2);
                gc[tt] = c;
                gs[tt] = s;
                rr = r2;
              }
            }
          }
          __syncthreads();
        }
        // every remaining row takes the rotations as its own chain, in place (rows are independent); (x, y) = (next column, carried):
        // carried <- c x - s y [the value that moves on], stored <- s x + c y  (rule C3 / C6)
        for(int i = tid; i < q; i += NT)
        {
          if(i == l) continue;
          const int s = Q.slot[i];
          int t0 = 0;
          double carry = 0.0;
          if(i < l)
            carry = Q.Sm[(size_t)s + (size_t)l * Q.ldq];
          else
            t0 = i - l - 1;
          double * srow = Q.Sm + s;
          double * trow = Q.St + (size_t)s * Q.ldq;
          for(int tt = t0; tt < len - 1; ++tt)
          {
            const double c = gc[tt], sn = gs[tt];
            const double xv = srow[(size_t)(l + tt + 1) * Q.ldq];
            const double nx = FMA(-sn, carry, c * xv);
            const double ny = FMA(sn, xv, c * carry);
            srow[(size_t)(l + tt) * Q.ldq] = ny;
            trow[l + tt] = ny;
            carry = nx;
          }
        }
        __syncthreads();
        // u.segment(l, q - l) = u.tail(q - l) (u has q + 1 entries); ActiveSet::deactivate(l); the slot map follows
        for(int base = l; base < q; base += NT)
        {
          const int k = base + tid;
          const double tu = (k < q) ? u[k + 1] : 0.0;
          const int ta = (k + 1 < q) ? K.active[k + 1] : 0;
          const int ts = (k + 1 < q) ? Q.slot[k + 1] : 0;
          __syncthreads();
          if(k < q) u[k] = tu;
          if(k + 1 < q)
          {
            K.active[k] = ta;
            Q.slot[k] = ts;
            Q.roff[k] = 8 * ts * Q.ldq;
          }
          __syncthreads();
        }
        if(tid == 0)
        {
          K.status[removed] = JRLQP_INACTIVE;
          Q.lstat[removed] = JRLQP_INACTIVE;
          Q.freel[Q.nfree] = sl;
          Q.roff[q - 1] = 0; // (zero beyond the last position: c9_st_g)
          // leave the group's list, order kept
          const int g = Q.sgrp[sl], o = Q.goff[g], c = Q.gcount[g];
          int at = 0;
          while(at < c && Q.glist[o + at] != sl) ++at;
          for(int e = at; e + 1 < c; ++e) Q.glist[o + e] = Q.glist[o + e + 1];
          Q.gcount[g] = c - 1;
        }
        ++Q.nfree;
        --q;
        skip_step1 = true;
        __syncthreads();
        stamp(10);
      }
    }
  }

done:
  __syncthreads();
  {
    double * xo = A.x + (size_t)pb * n;
    for(int i = tid; i < n; i += NT) xo[i] = x[i];
    // DualSolver::multipliers (DualSolver.cpp:38-69)
    double * uo = A.u + (size_t)pb * (m + nb);
    for(int i = tid; i < m + nb; i += NT) uo[i] = 0.0;
    __syncthreads();
    for(int k = tid; k < q; k += NT)
    {
      const int i = K.active[k];
      const int st = K.status[i];
      uo[i] = (st == JRLQP_UPPER || st == JRLQP_UPPER_BOUND) ? u[k] : -u[k];
    }
    if(tid == 0)
    {
      A.term[pb] = rc;
      A.iters[pb] = it;
      A.f[pb] = f;
      if(A.prof)
      {
        for(int i = 0; i < 13; ++i) A.prof[(size_t)pb * A.prof_stride + i] = pacc[i];
        A.prof[(size_t)pb * A.prof_stride + 13] = clock64() - cyc0;       // core clock cycles
        A.prof[(size_t)pb * A.prof_stride + 14] = wall_clock64() - wall0; // 100 MHz ticks
      }
    }
  }
}

// transposed copies of one problem's blocks (BgiArgs::Tw): workgroup = (problem, block); reads walk the source's columns
__global__ __launch_bounds__(256) void bgi_pack_kernel(const BgiArgs A)
{
  const int per = 2 * A.nblk - 1, pb = blockIdx.x / per, t = blockIdx.x % per;
  const double * src;
  int ld, rows, cols;
  double * dst = A.Tw + (size_t)pb * A.tw_len;
  if(t < A.nblk)
  {
    src = A.diag[(size_t)pb * A.nblk + t];
    ld = A.ldd[t];
    rows = cols = A.bs[t];
    dst += A.tw_doff[t];
  }
  else
  {
    const int i = t - A.nblk;
    src = A.side[(size_t)pb * (A.nblk - 1) + i];
    ld = A.lds[i];
    cols = A.tw_sc[i];
    // rows of side block i: arrow down: the head's; arrow up / tridiagonal: block i + 1's
    rows = (A.kind == 0) ? A.bs[A.nblk - 1] : A.bs[i + 1];
    dst += A.tw_soff[i];
  }
  for(int e = threadIdx.x; e < rows * cols; e += 256)
  {
    const int r = e % rows, c = e / rows;
    dst[c + (size_t)r * cols] = src[r + (size_t)c * ld];
  }
}

template<int NT, bool BIG>
hipError_t launch_c9_b(const BgiArgs & a, int nprob, size_t lds, hipStream_t st)
{
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&blockgi_c9_kernel<NT, BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if(e != hipSuccess) return e;
  hipLaunchKernelGGL((blockgi_c9_kernel<NT, BIG>), dim3(nprob), dim3(NT), lds, st, a);
  return hipGetLastError();
}
template<int NT>
hipError_t launch_c9(const BgiArgs & a, int nprob, size_t lds, hipStream_t st)
{
  return (a.maxb > 64) ? launch_c9_b<NT, true>(a, nprob, lds, st) : launch_c9_b<NT, false>(a, nprob, lds, st);
}
} // namespace

long blockgi_c9_lds_bytes(int n, int m, int nb, int nblk, int ncb, int xu_global)
{
  const int n_pad = (n + 3) & ~3;
  int qcap = (n < m + nb ? n : m + nb) + 1;
  qcap = (qcap + 3) & ~3;
  const int ngrp = ncb + (nb ? nblk : 0);
  int P2 = 1;
  while(P2 < nblk) P2 <<= 1;
  if(P2 < 64) P2 = 64;
  const long dbl = (xu_global ? 2L * n_pad + 3L * qcap : 3L * n_pad + 4L * qcap) + 128 + P2 + 8 * 128 + 2L * nblk + 32 + 32 + 8;
  const long ints = 3L * qcap + 64 + 2L * ngrp + (m + nb + 1) + 3L * nblk + (n + 2) / 2 + (m + nb + 4) / 4 + 2;
  return (long)sizeof(double) * dbl + 4 * ints + (long)sizeof(BlkMeta) * nblk + 8;
}

hipError_t launch_blockgi_pack(const BgiArgs & a, int nprob, hipStream_t st)
{
  hipLaunchKernelGGL(bgi_pack_kernel, dim3(nprob * (2 * a.nblk - 1)), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_blockgi_c9(const BgiArgs & a, int nprob, hipStream_t st)
{
  const size_t lds = (size_t)blockgi_c9_lds_bytes(a.n, a.m, a.nb, a.nblk, a.ncb, a.xu_global);
  // four waves per problem up to 1024 unknowns (and in the throughput layout), eight beyond: the block-parallel phases (L^-T over ~100
  // blocks, the scan, y -= B dl) and the triangular products scale with the wave count; no rule of C9 depends on it
  if(a.n <= 1024 || a.xu_global) return launch_c9<256>(a, nprob, lds, st);
  return launch_c9<512>(a, nprob, lds, st);
}
} // namespace jrlqp
