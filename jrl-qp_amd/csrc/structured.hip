// structured.hip -- second kernel family: Cholesky factorisation and triangular solves of block-arrow and
// block-tridiagonal SPD matrices given as lists of diagonal / off-diagonal blocks (gfx950).
//
//   reference (jrl-umi3218/jrl-qp):
//     blockArrowLLT_              src/decomposition/blockArrowLLT.cpp:53-81
//     blockArrowLSolve(_)         :92-176      blockArrowLTransposeSolve(_)  :179-277
//     triBlockDiagLLT             src/decomposition/triBlockDiagLLT.cpp:9-35
//     triBlockDiagLSolve          :37-96       triBlockDiagLTransposeSolve   :98-158
//
// Parallel structure: the b-1 off-head blocks of an arrow matrix are independent
// (one workgroup each: LLT of D_i, B_i = S_i L_i^-T); the head block receives  D_b -= sum_i B_i B_i^T  -- under
// the CANON rounding contract each ELEMENT of D_b owns one in-order fma chain over all blocks (that is what
// makes the result bit-identical to the sequential reference order), so the head kernel gives one thread per
// element and walks the blocks; the block-tridiagonal chain is inherently serial across blocks and runs in one
// workgroup (parallelism comes from batching problems: grid.x = number of problems).
//
// Blocks are at most 64 x 64 (one wavefront row-per-lane); the substitutions keep the running column in
// registers and move pivots with v_readlane, like the dense kernel.  -ffp-contract=off; every fma explicit.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "structured.h"

namespace jrlqp
{
namespace
{

// ---- wave-level primitives on a block held in LDS (ld odd => conflict-free row-per-lane access) -----------
// in-place lower Cholesky, lane = row; returns -1 or the failing column (Eigen llt_inplace semantics)
__device__ int wave_llt(double * L, int ld, int n, int lane)
{
  for(int j = 0; j < n; ++j)
  {
    double acc = 0.0;
    {
      // (every lane runs the chain on a valid row: no divergent LDS traffic; rows outside [j, n) are discarded)
      const int rl = (lane >= j && lane < n) ? lane : j;
      acc = chain2_neg(0, 1, j, AT(L, ld, rl, j), [&](int k) { return AT(L, ld, rl, k); }, [&](int k) { return AT(L, ld, j, k); });
      acc = (lane >= j && lane < n) ? acc : 0.0;
    }
    double piv = bcast_lane(acc, j);
    if(piv <= 0.0) return j;
    double ljj = __builtin_sqrt(piv);
    if(lane == j)
      AT(L, ld, j, j) = ljj;
    else if(lane > j && lane < n)
      AT(L, ld, lane, j) = acc / ljj;
    wave_sync();
  }
  return -1;
}

__device__ __forceinline__ void load_block(double * dst, int ldd, const double * src, int lds_, int rows, int cols, int tid, int nthr)
{
  for(int e = tid; e < rows * cols; e += nthr)
  {
    int i = e % rows, j = e / rows;
    AT(dst, ldd, i, j) = AT(src, lds_, i, j);
  }
}

constexpr int kTriThreads = 256;

// in-place lower Cholesky of the n x n block in LDS; diagL receives the diagonal of the factor (the diagonal of L in
// LDS keeps the running sums).  Returns -1, or the failing column j: columns < j are final, nothing of column j on is
// meaningful (the caller writes back columns < j only, like Eigen's llt_inplace leaves the rest untouched).
__device__ int block_llt(double * L, int ld, int n, double * diagL, int tid)
{
  for(int c = tid; c < n; c += kTriThreads) diagL[c] = AT(L, ld, c, c);
  __syncthreads();
  const double p0 = AT(L, ld, 0, 0);
  if(p0 <= 0.0) return 0;
  const double l0 = __builtin_sqrt(p0);
  for(int r = tid; r < n; r += kTriThreads)
  {
    if(r == 0)
      diagL[0] = l0;
    else
      AT(L, ld, r, 0) = AT(L, ld, r, 0) / l0;
  }
  __syncthreads();
  for(int k = 0; k + 1 < n; ++k)
  {
    const int c1 = k + 1;
    const double lck = AT(L, ld, c1, k);
    const double piv = FMA(-lck, lck, AT(L, ld, c1, c1)); // every thread: the pivot of the column that becomes final now
    if(piv <= 0.0) return c1;
    const double lp = __builtin_sqrt(piv);
    // threads as a 16 x 16 tile over (row, column) of the trailing block: no integer division on the way
    for(int c = c1 + (tid >> 4); c < n; c += 16)
      for(int r = c1 + (tid & 15); r < n; r += 16)
      {
        if(r < c) continue;
        if(r == c1) // (c1, c1): its value is `piv`, kept in registers; only the factor's diagonal is recorded
        {
          diagL[c1] = lp;
          continue;
        }
        const double acc = FMA(-AT(L, ld, r, k), AT(L, ld, c, k), AT(L, ld, r, c));
        if(c == c1)
          AT(L, ld, r, c) = acc / lp;
        else
          AT(L, ld, r, c) = acc;
      }
    __syncthreads();
  }
  return -1;
}

// X (r x n) <- X L^-T, L lower with its diagonal in diagL
__device__ void block_trsm_right_lt(const double * L, int ldl, const double * diagL, int n, double * X, int ldx, int rows, int tid)
{
  const double d0 = diagL[0];
  for(int r = tid; r < rows; r += kTriThreads) AT(X, ldx, r, 0) = AT(X, ldx, r, 0) / d0;
  __syncthreads();
  for(int k = 0; k + 1 < n; ++k)
  {
    const int c1 = k + 1;
    const double d1 = diagL[c1];
    for(int j = c1 + (tid >> 4); j < n; j += 16)
      for(int r = (tid & 15); r < rows; r += 16)
      {
        const double acc = FMA(-AT(X, ldx, r, k), AT(L, ldl, j, k), AT(X, ldx, r, j));
        if(j == c1)
          AT(X, ldx, r, j) = acc / d1;
        else
          AT(X, ldx, r, j) = acc;
      }
    __syncthreads();
  }
}

// X (n x ncols) <- L^-1 X, L lower with its diagonal in diagL (Eigen triangularView<Lower>::solveInPlace on the left,
// blockArrowLLT.cpp:70): row k is final after its division; every row below then receives its k-th term
__device__ void block_trsm_left(const double * L, int ldl, const double * diagL, int n, double * X, int ldx, int ncols, int tid)
{
  const double d0 = diagL[0];
  for(int c = tid; c < ncols; c += kTriThreads) AT(X, ldx, 0, c) = AT(X, ldx, 0, c) / d0;
  __syncthreads();
  for(int k = 0; k + 1 < n; ++k)
  {
    const int r1 = k + 1;
    const double d1 = diagL[r1];
    for(int c = (tid >> 4); c < ncols; c += 16)
      for(int r = r1 + (tid & 15); r < n; r += 16)
      {
        const double acc = FMA(-AT(L, ldl, r, k), AT(X, ldx, k, c), AT(X, ldx, r, c));
        if(r == r1)
          AT(X, ldx, r, c) = acc / d1;
        else
          AT(X, ldx, r, c) = acc;
      }
    __syncthreads();
  }
}

// =========================================================================================================
// block arrow LLT, phase 1: one 256-thread workgroup per off-head block, right-looking out of LDS (see the comment on
// the block-tridiagonal kernel: same bits as the column-by-column order, one barrier + one division per column)
// =========================================================================================================
__global__ __launch_bounds__(kTriThreads) void arrow_block_kernel(const BlockDesc D)
{
  extern __shared__ double sm[];
  const int tid = threadIdx.x;
  const int i = blockIdx.x; // logical block index < nblk - 1
  const int pb = blockIdx.y; // problem in the batch
  const int id = dsel(D.nblk, D.up, i), ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int ni = D.bs[id], nbk = D.bs[ib];
  double * Dg = D.diag[(size_t)pb * D.nblk + id];
  double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
  const int ldd = D.ldd[id], ldsg = D.lds[i];
  const int ldl = ni | 1;
  double * Ls = sm;
  double * Xs = Ls + (size_t)ni * ldl;
  const int ldx = D.up ? (ni | 1) : (nbk | 1);
  double * diagL = Xs + (size_t)(D.up ? nbk : ni) * ldx;
  load_block(Ls, ldl, Dg, ldd, ni, ni, tid, kTriThreads);
  if(!D.up)
    load_block(Xs, ldx, Sg, ldsg, nbk, ni, tid, kTriThreads); // side[i]: nbk x ni
  else
    load_block(Xs, ldx, Sg, ldsg, ni, nbk, tid, kTriThreads); // side[i]: ni x nbk
  __syncthreads();
  const int ret = block_llt(Ls, ldl, ni, diagL, tid);
  __syncthreads();
  if(ret > 0 && tid == 0) atomicExch(&D.fail[pb], 1); // reference quirk: a failure at column 0 goes undetected (blockArrowLLT.cpp:65)
  const int cols = (ret < 0) ? ni : ret; // write the factor back: the finished columns only on failure
  for(int e = tid; e < ni * cols; e += kTriThreads)
  {
    const int r = e % ni, c = e / ni;
    if(r > c)
      AT(Dg, ldd, r, c) = AT(Ls, ldl, r, c);
    else if(r == c)
      AT(Dg, ldd, r, c) = diagL[c];
  }
  if(ret > 0) return;
  if(!D.up)
  {
    block_trsm_right_lt(Ls, ldl, diagL, ni, Xs, ldx, nbk, tid); // side[i] <- side[i] L_i^-T   (:71-72)
    for(int e = tid; e < nbk * ni; e += kTriThreads) AT(Sg, ldsg, e % nbk, e / nbk) = AT(Xs, ldx, e % nbk, e / nbk);
  }
  else
  {
    block_trsm_left(Ls, ldl, diagL, ni, Xs, ldx, nbk, tid); // side[i] <- L_i^-1 side[i]; it then stores B_i^T   (:68-70)
    for(int e = tid; e < ni * nbk; e += kTriThreads) AT(Sg, ldsg, e % ni, e / ni) = AT(Xs, ldx, e % ni, e / ni);
  }
}

// =========================================================================================================
// block arrow LLT, phase 2: head block.  D_b -= sum_i B_i B_i^T (one thread per lower-triangle element, blocks
// walked in order: one fma chain per element), then LLT(D_b).
// =========================================================================================================
__global__ __launch_bounds__(256) void arrow_head_kernel(const BlockDesc D)
{
  extern __shared__ double sm[]; // Ls (nbk x ldl) | Bs (64 x 65): the current side block, staged with coalesced loads
  const int tid = threadIdx.x;
  const int pb = blockIdx.x;
  const int ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int nbk = D.bs[ib];
  double * Dg = D.diag[(size_t)pb * D.nblk + ib];
  const int ldd = D.ldd[ib];
  const int ldl = nbk | 1;
  double * Ls = sm;
  double * Bs = sm + (size_t)nbk * ldl;
  if(D.fail[pb]) return; // a block already failed: the reference returns false before touching D_b further
  // each thread owns up to 16 lower-triangle elements (r, c) of D_b: e = tid + 256*s
  constexpr int SL = 16;
  double acc[SL];
#pragma unroll
  for(int s = 0; s < SL; ++s)
  {
    const int e = tid + 256 * s;
    const int r = e % nbk, c = e / nbk;
    acc[s] = (e < nbk * nbk && r >= c) ? AT(Dg, ldd, r, c) : 0.0;
  }
  // software pipeline over the blocks: the next block travels from L2 to registers while this one is consumed
  double nxt[SL];
  auto fetch = [&](int i) {
    const int id = dsel(D.nblk, D.up, i);
    const int ni = D.bs[id];
    const double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
    const int ldsg = D.lds[i];
    const int rows = D.up ? ni : nbk, cols = D.up ? nbk : ni; // stored shape of side[i]
#pragma unroll
    for(int s = 0; s < SL; ++s)
    {
      const int e = tid + 256 * s;
      nxt[s] = (e < rows * cols) ? AT(Sg, ldsg, e % rows, e / rows) : 0.0;
    }
  };
  if(D.nblk > 1) fetch(0);
  for(int i = 0; i < D.nblk - 1; ++i)
  {
    const int id = dsel(D.nblk, D.up, i);
    const int ni = D.bs[id];
    const int rows = D.up ? ni : nbk, cols = D.up ? nbk : ni;
    __syncthreads();
#pragma unroll
    for(int s = 0; s < SL; ++s)
    {
      const int e = tid + 256 * s;
      if(e < rows * cols) Bs[(e % rows) + (e / rows) * 65] = nxt[s];
    }
    __syncthreads();
    if(i + 1 < D.nblk - 1) fetch(i + 1);
#pragma unroll
    for(int s = 0; s < SL; ++s)
    {
      const int e = tid + 256 * s;
      const int r = e % nbk, c = e / nbk;
      if(e < nbk * nbk && r >= c)
      {
        if(!D.up) // B_i(r,k) B_i(c,k)
          acc[s] = chain2_neg(0, 1, ni, acc[s], [&](int k) { return Bs[r + k * 65]; }, [&](int k) { return Bs[c + k * 65]; });
        else // side stores B_i^T
          acc[s] = chain2_neg(0, 1, ni, acc[s], [&](int k) { return Bs[k + r * 65]; }, [&](int k) { return Bs[k + c * 65]; });
      }
    }
  }
  __syncthreads();
#pragma unroll
  for(int s = 0; s < SL; ++s)
  {
    const int e = tid + 256 * s;
    const int r = e % nbk, c = e / nbk;
    if(e < nbk * nbk && r >= c) AT(Ls, ldl, r, c) = acc[s];
  }
  __syncthreads();
  if(tid < 64)
  {
    int ret = wave_llt(Ls, ldl, nbk, tid);
    if(ret > 0 && tid == 0) atomicExch(&D.fail[pb], 1); // (:80) ret <= 0 is success
  }
  __syncthreads();
  for(int e = tid; e < nbk * nbk; e += 256)
  {
    int r = e % nbk, c = e / nbk;
    if(r >= c) AT(Dg, ldd, r, c) = AT(Ls, ldl, r, c);
  }
}

// =========================================================================================================
// block arrow solves.  M is N x nrhs column-major (already permuted for the up variant).
// =========================================================================================================
// L X = M, phase 1: X_i = L_i^-1 M_i for the off-head blocks (window [start, end) of non-zero rows, :113-140)
__global__ __launch_bounds__(64) void arrow_lsolve_blocks_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start, int end)
{
  const int lane = threadIdx.x, i = blockIdx.x, pb = blockIdx.y;
  const int id = dsel(D.nblk, D.up, i);
  const int ni = D.bs[id];
  int n0 = 0;
  for(int k = 0; k < i; ++k) n0 += D.bs[dsel(D.nblk, D.up, k)];
  int s = max(start - n0, 0);
  if(ni < s || end <= n0) return; // M_i is zero
  const double * Dg = D.diag[(size_t)pb * D.nblk + id];
  const int ldd = D.ldd[id];
  const double * Lsub = Dg + s + (size_t)s * ldd; // bottomRightCorner(ni - s, ni - s)
  double * Mp = M + (size_t)pb * D.mstride;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mi = Mp + n0 + s + (size_t)c * ldm;
    double xi = (lane < ni - s) ? Mi[lane] : 0.0;
    xi = wave_fwd(Lsub, ldd, ni - s, xi, lane);
    if(lane < ni - s) Mi[lane] = xi;
  }
}

// phase 2: M_b -= sum_i B_i(:, s:) X_i (one thread per (row, rhs) of M_b, blocks in order), then X_b = L_b^-1 M_b
__global__ __launch_bounds__(64) void arrow_lsolve_head_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start, int end)
{
  const int lane = threadIdx.x, pb = blockIdx.x;
  const int ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int nbk = D.bs[ib];
  int ntot = 0;
  for(int k = 0; k < D.nblk; ++k) ntot += D.bs[k];
  double * Mp = M + (size_t)pb * D.mstride;
  const double * Lb = D.diag[(size_t)pb * D.nblk + ib];
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mb = Mp + (ntot - nbk) + (size_t)c * ldm;
    double acc = (lane < nbk) ? Mb[lane] : 0.0;
    int n0 = 0;
    for(int i = 0; i < D.nblk - 1; ++i)
    {
      const int id = dsel(D.nblk, D.up, i);
      const int ni = D.bs[id];
      int s = max(start - n0, 0);
      if(!(ni < s || end <= n0) && lane < nbk)
      {
        const double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
        const int ldsg = D.lds[i];
        const double * Xi = Mp + n0 + (size_t)c * ldm;
        for(int k = s; k < ni; ++k)
        {
          double bv = D.up ? AT(Sg, ldsg, k, lane) : AT(Sg, ldsg, lane, k);
          acc = FMA(-bv, Xi[k], acc);
        }
      }
      n0 += ni;
    }
    acc = wave_fwd(Lb, D.ldd[ib], nbk, acc, lane);
    if(lane < nbk) Mb[lane] = acc;
  }
}

// L^T X = M, phase 1: head block X_b = L_b(0:r,0:r)^-T M_b(0:r) when end reaches into it (:203-212)
__global__ __launch_bounds__(64) void arrow_ltsolve_head_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start, int end)
{
  const int lane = threadIdx.x, pb = blockIdx.x;
  const int ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int nbk = D.bs[ib];
  int ntot = 0;
  for(int k = 0; k < D.nblk; ++k) ntot += D.bs[k];
  if(!(end > ntot - nbk)) return;
  const int r = end - ntot + nbk;
  const double * Lb = D.diag[(size_t)pb * D.nblk + ib];
  double * Mp = M + (size_t)pb * D.mstride;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mb = Mp + (ntot - nbk) + (size_t)c * ldm;
    double xi = (lane < r) ? Mb[lane] : 0.0;
    xi = wave_bwd(Lb, D.ldd[ib], r, xi, lane);
    if(lane < r) Mb[lane] = xi;
  }
}

// phase 2: per off-head block  M_i -= B_i^T X_b (descending k), then L_i^T solve (:217-251)
__global__ __launch_bounds__(64) void arrow_ltsolve_blocks_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start, int end)
{
  const int lane = threadIdx.x, i = blockIdx.x, pb = blockIdx.y;
  const int ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int nbk = D.bs[ib];
  int ntot = 0;
  for(int k = 0; k < D.nblk; ++k) ntot += D.bs[k];
  const bool zero = !(end > ntot - nbk);
  const int id = dsel(D.nblk, D.up, i);
  const int ni = D.bs[id];
  int n0 = 0;
  for(int k = 0; k < i; ++k) n0 += D.bs[dsel(D.nblk, D.up, k)];
  if(zero && start >= n0 + ni) return;
  const double * Li = D.diag[(size_t)pb * D.nblk + id];
  const int ldd = D.ldd[id];
  const double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
  const int ldsg = D.lds[i];
  double * Mp = M + (size_t)pb * D.mstride;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mi = Mp + n0 + (size_t)c * ldm;
    const double * Xb = Mp + (ntot - nbk) + (size_t)c * ldm;
    double acc = (lane < ni) ? Mi[lane] : 0.0;
    if(!zero && lane < ni)
      for(int k = nbk - 1; k >= 0; --k)
      {
        double bv = D.up ? AT(Sg, ldsg, lane, k) : AT(Sg, ldsg, k, lane);
        acc = FMA(-bv, Xb[k], acc);
      }
    if(end >= n0)
    {
      const int r = (end >= n0 + ni) ? ni : (end - n0);
      double solved = wave_bwd(Li, ldd, r, acc, lane);
      acc = (lane < r) ? solved : acc;
    }
    if(lane < ni) Mi[lane] = acc;
  }
}

// up-arrow row permutation of the right-hand side (blockArrowLLT.cpp:163-170, :266-271), through a scratch copy
__global__ void permute_rows_kernel(double * M, int ldm, int nrhs, int ntot, int n0, int backward, double * tmp, long mstride)
{
  double * Mp = M + (size_t)blockIdx.y * mstride;
  double * tp = tmp + (size_t)blockIdx.y * (size_t)ntot * nrhs;
  const int s = ntot - n0;
  for(int e = blockIdx.x * blockDim.x + threadIdx.x; e < ntot * nrhs; e += gridDim.x * blockDim.x)
  {
    int r = e % ntot, c = e / ntot;
    tp[e] = Mp[r + (size_t)c * ldm];
  }
  __syncthreads(); // single-block launch: gridDim.x == 1
  for(int e = threadIdx.x; e < ntot * nrhs; e += blockDim.x)
  {
    int r = e % ntot, c = e / ntot;
    int src;
    if(!backward)
      src = (r < s) ? (r + n0) : (r - s); // v.top(s) = tmp.bottom(s); v.bottom(n0) = tmp.top(n0)
    else
      src = (r < n0) ? (r + s) : (r - n0); // v.bottom(s) = tmp.top(s); v.top(n0) = tmp.bottom(n0)
    Mp[r + (size_t)c * ldm] = tp[src + (size_t)c * ntot];
  }
}

// =========================================================================================================
// block tridiagonal: serial chain over blocks, one workgroup per problem (grid.x = number of problems).
// Within a block the three steps run RIGHT-LOOKING on 256 threads out of LDS: as soon as column k of L (or of X) is
// final, every element to its right receives its k-th term -- each element still accumulates its terms in ascending k,
// which is all the rounding contract fixes, so the result is the bit pattern of the column-by-column reference order,
// while the critical path is one barrier + one division per COLUMN instead of one LDS round trip per TERM.
// =========================================================================================================
// (the chain from block i0 on; i0 > 0: the fast kernel below hands over here after a pivot failure at column 0 of block i0,
// the one failure the reference does not notice -- block i0 as it stands in memory is then the block to go on with)
__device__ void tri_llt_from(const BlockDesc & D, int i0, double * sm)
{
  const int tid = threadIdx.x, pb = blockIdx.x;
  int maxb = 0;
  for(int k = 0; k < D.nblk; ++k) maxb = max(maxb, D.bs[k]);
  const int ldl = maxb | 1;
  // three block buffers: the diagonal block being factorised, the sub-diagonal block, and the NEXT diagonal block,
  // which is fetched while this one is factorised and receives its rank update in LDS (no store -> load round trip
  // through HBM between consecutive blocks of the chain)
  double * Ls = sm;
  double * Xs = Ls + (size_t)maxb * ldl;
  double * Ns = Xs + (size_t)maxb * ldl;
  double * diagL = Ns + (size_t)maxb * ldl;
  load_block(Ls, ldl, D.diag[(size_t)pb * D.nblk + i0], D.ldd[i0], D.bs[i0], D.bs[i0], tid, kTriThreads);
  for(int i = i0; i < D.nblk; ++i)
  {
    const int ni = D.bs[i];
    double * Dg = D.diag[(size_t)pb * D.nblk + i];
    const int ldd = D.ldd[i];
    const bool last = (i == D.nblk - 1);
    const int n1 = last ? 0 : D.bs[i + 1];
    double * Sg = last ? nullptr : D.side[(size_t)pb * (D.nblk - 1) + i];
    const int ldsg = last ? 1 : D.lds[i];
    if(!last)
    {
      load_block(Xs, ldl, Sg, ldsg, n1, ni, tid, kTriThreads);
      load_block(Ns, ldl, D.diag[(size_t)pb * D.nblk + i + 1], D.ldd[i + 1], n1, n1, tid, kTriThreads);
    }
    __syncthreads();
    const int ret = block_llt(Ls, ldl, ni, diagL, tid);
    __syncthreads();
    // write the factor back: all of the lower triangle on success, the finished columns only on failure
    const int cols = (ret < 0) ? ni : ret;
    for(int e = tid; e < ni * cols; e += kTriThreads)
    {
      const int r = e % ni, c = e / ni;
      if(r > c)
        AT(Dg, ldd, r, c) = AT(Ls, ldl, r, c);
      else if(r == c)
        AT(Dg, ldd, r, c) = diagL[c];
    }
    if(ret > 0) // triBlockDiagLLT.cpp:19,32 (a failure at column 0 goes undetected there, and here)
    {
      if(tid == 0) D.fail[pb] = 1;
      return;
    }
    if(last) break;
    block_trsm_right_lt(Ls, ldl, diagL, ni, Xs, ldl, n1, tid); // S_i <- S_i L_i^-T
    for(int e = tid; e < n1 * ni; e += kTriThreads) AT(Sg, ldsg, e % n1, e / n1) = AT(Xs, ldl, e % n1, e / n1);
    // D_{i+1} -= S_i S_i^T (lower), in LDS: one element per thread, terms in ascending k
    for(int c = (tid >> 4); c < n1; c += 16)
    for(int r = (tid & 15); r < n1; r += 16)
    {
      if(r < c) continue;
      const double v = chain2_neg(0, 1, ni, AT(Ns, ldl, r, c), [&](int k) { return AT(Xs, ldl, r, k); }, [&](int k) { return AT(Xs, ldl, c, k); });
      AT(Ns, ldl, r, c) = v;
      AT(D.diag[(size_t)pb * D.nblk + i + 1], D.ldd[i + 1], r, c) = v; // (what a failing factorisation of block i+1 leaves behind)
    }
    __syncthreads();
    double * t = Ls; // the updated next block becomes the current one
    Ls = Ns;
    Ns = t;
  }
}

__global__ __launch_bounds__(kTriThreads) void tri_llt_kernel(const BlockDesc D)
{
  extern __shared__ double sm[];
  tri_llt_from(D, 0, sm);
}

// acc <- fma(-v, (src of lane LANE), acc) as ONE unit the compiler cannot pull apart: left to itself hipcc hoists the
// v_readlane pairs of a whole column in front of the fmas and then spills the scalar results through VGPR lanes (four more
// instructions per term, measured: the register kernel below no faster than the barrier kernel).
template<int LANE>
__device__ __forceinline__ void fma_neg_bcast(double & acc, double v, double src)
{
  const int lo = __double2loint(src), hi = __double2hiint(src);
  asm volatile("v_readlane_b32 s90, %2, %4\n\t"
               "v_readlane_b32 s91, %3, %4\n\t"
               "s_nop 1\n\t"
               "v_fma_f64 %0, -%1, s[90:91], %0"
               : "+v"(acc)
               : "v"(v), "v"(lo), "v"(hi), "n"(LANE)
               : "s90", "s91");
}
template<int J, int NB, int C = J + 1>
__device__ __forceinline__ void column_update(double (&a)[NB], double v, double src)
{
  if constexpr(C < NB)
  {
    fma_neg_bcast<C>(a[C], v, src);
    column_update<J, NB, C + 1>(a, v, src);
  }
}

// column J of the in-register factorisation (lane = row) and the columns after it
template<int J, int NB>
__device__ __forceinline__ void llt_columns(double (&a)[NB], int & ret, int lane, int ni, double * Dg, int ldd)
{
  if constexpr(J < NB)
  {
    if(ret < 0)
    {
      const double piv = bcast_lane(a[J], J);
      if(piv <= 0.0)
        ret = J;
      else
      {
        const double ljj = __builtin_sqrt(piv);
        const double lij = (lane == J) ? ljj : a[J] / ljj;
        a[J] = lij;
        if(lane >= J && lane < ni && J < ni) AT(Dg, ldd, lane, J) = lij; // (columns before a failing one are final)
        column_update<J, NB>(a, lij, lij);                               // a[c] -= L(lane, J) * L(c, J) for c > J
      }
    }
    llt_columns<J + 1, NB>(a, ret, lane, ni, Dg, ldd);
  }
}
// column J of X <- X L^-T (lane = row of X, L in the registers a[] of the same lanes) and the columns after it
template<int J, int NB>
__device__ __forceinline__ void trsm_columns(double (&x)[NB], const double (&a)[NB])
{
  if constexpr(J < NB)
  {
    const double xj = x[J] / bcast_lane(a[J], J);
    x[J] = xj;
    column_update<J, NB>(x, xj, a[J]); // x[j2] -= x(J) * L(j2, J) for j2 > J
    trsm_columns<J + 1, NB>(x, a);
  }
}

// ---- the same chain for blocks of at most NB rows, the serial part in REGISTERS -----------------------------------------
// One block costs the barrier-per-column kernel above ~40 us: sixty workgroup barriers around one division each.  Here wave 0
// keeps its row of the diagonal block in registers (lane = row, a[c] = D(lane, c)), factorises it right-looking -- the pivot
// and the column entries L(c, j) move by v_readlane, no LDS round trip and no barrier inside a block --, keeps its row of the
// sub-diagonal block in registers as well for S <- S L^-T, and only the rank update D_{i+1} -= S S^T, which has n^2 independent
// chains, goes to all four waves through LDS.  Per element the operations and their order are those of block_llt /
// block_trsm_right_lt (and of the oracle): same bits.  Blocks are padded to NB x NB with the identity (no per-element
// bounds tests): the padding factorises to the identity and contributes exact zeros.
template<int NB>
__global__ __launch_bounds__(kTriThreads) void tri_llt_fast_kernel(const BlockDesc D)
{
  extern __shared__ double sm[];
  const int tid = threadIdx.x, pb = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int ldl = NB | 1;
  double * Ls = sm;                 // the diagonal block being factorised (with its rank update applied)
  double * Xs = Ls + NB * ldl;      // S_i, then S_i L_i^-T
  double * Ns = Xs + NB * ldl;      // D_{i+1}
  __shared__ int s_ret;
  load_block(Ls, ldl, D.diag[(size_t)pb * D.nblk], D.ldd[0], D.bs[0], D.bs[0], tid, kTriThreads);
  for(int i = 0; i < D.nblk; ++i)
  {
    const int ni = D.bs[i];
    double * Dg = D.diag[(size_t)pb * D.nblk + i];
    const int ldd = D.ldd[i];
    const bool last = (i == D.nblk - 1);
    const int n1 = last ? 0 : D.bs[i + 1];
    double * Sg = last ? nullptr : D.side[(size_t)pb * (D.nblk - 1) + i];
    const int ldsg = last ? 1 : D.lds[i];
    if(!last)
    {
      load_block(Xs, ldl, Sg, ldsg, n1, ni, tid, kTriThreads);
      load_block(Ns, ldl, D.diag[(size_t)pb * D.nblk + i + 1], D.ldd[i + 1], n1, n1, tid, kTriThreads);
    }
    __syncthreads();
    if(wave == 0)
    {
      const bool row = lane < ni;
      const int rl = row ? lane : 0;
      double a[NB];
#pragma unroll
      for(int c = 0; c < NB; ++c)
      {
        const double v = AT(Ls, ldl, rl, (c < ni) ? c : 0);
        a[c] = (row && c < ni) ? v : ((lane == c) ? 1.0 : 0.0); // identity padding
      }
      int ret = -1;
      llt_columns<0, NB>(a, ret, lane, ni, Dg, ldd);
      if(lane == 0) s_ret = ret;
      if(ret < 0 && !last)
      {
        // S_i <- S_i L_i^-T, lane = row of S: x(j) = (x(j) - sum_{k<j} x(k) L(j,k)) / L(j,j), right-looking, L from the registers
        const bool srow = lane < n1;
        const int sl = srow ? lane : 0;
        double x[NB];
#pragma unroll
        for(int c = 0; c < NB; ++c)
        {
          const double v = AT(Xs, ldl, sl, (c < ni) ? c : 0);
          x[c] = (srow && c < ni) ? v : 0.0;
        }
        trsm_columns<0, NB>(x, a);
#pragma unroll
        for(int c = 0; c < NB; ++c)
          if(srow && c < ni)
          {
            AT(Xs, ldl, lane, c) = x[c];
            AT(Sg, ldsg, lane, c) = x[c];
          }
      }
    }
    __syncthreads();
    const int ret = s_ret;
    if(ret > 0) // triBlockDiagLLT.cpp:19,32: the failure the reference reports
    {
      if(tid == 0) D.fail[pb] = 1;
      return;
    }
    if(ret == 0) // (a failure at column 0 goes undetected there: the barrier kernel's code reproduces what follows)
    {
      __syncthreads();
      tri_llt_from(D, i, sm);
      return;
    }
    if(last) break;
    // D_{i+1} -= S_i S_i^T (lower), in LDS: one element per thread, terms in ascending k
    for(int c = (tid >> 4); c < n1; c += 16)
      for(int r = (tid & 15); r < n1; r += 16)
      {
        if(r < c) continue;
        const double v = chain2_neg(0, 1, ni, AT(Ns, ldl, r, c), [&](int k) { return AT(Xs, ldl, r, k); }, [&](int k) { return AT(Xs, ldl, c, k); });
        AT(Ns, ldl, r, c) = v;
        AT(D.diag[(size_t)pb * D.nblk + i + 1], D.ldd[i + 1], r, c) = v; // (what a failing factorisation of block i+1 leaves behind)
      }
    __syncthreads();
    double * t = Ls; // the updated next block becomes the current one
    Ls = Ns;
    Ns = t;
  }
}

// triBlockDiagLSolve (triBlockDiagLLT.cpp:37-96): forward block substitution with the zero-window `start`
__global__ __launch_bounds__(64) void tri_lsolve_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start)
{
  const int lane = threadIdx.x, pb = blockIdx.x;
  double * Mp = M + (size_t)pb * D.mstride;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mc = Mp + (size_t)c * ldm;
    int n = 0, l = 0, li = 0;
    bool zero = true;
    for(int i = 0; i < D.nblk; ++i)
    {
      const int ni = D.bs[i];
      const double * Li = D.diag[(size_t)pb * D.nblk + i];
      const int ldd = D.ldd[i];
      if(n + ni >= start)
      {
        if(zero)
        {
          const int r = n + ni - start, off = ni - r;
          double xi = (lane < r) ? Mc[start + lane] : 0.0;
          xi = wave_fwd(Li + off + (size_t)off * ldd, ldd, r, xi, lane);
          if(lane < r) Mc[start + lane] = xi;
          zero = false;
        }
        else
        {
          const double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i - 1];
          const int ldsg = D.lds[i - 1];
          double acc = (lane < ni) ? Mc[n + lane] : 0.0;
          if(lane < ni)
            for(int k = 0; k < li; ++k) acc = FMA(-AT(Sg, ldsg, lane, k), Mc[l + k], acc);
          acc = wave_fwd(Li, ldd, ni, acc, lane);
          if(lane < ni) Mc[n + lane] = acc;
        }
        __threadfence_block();
        wave_sync();
      }
      l = n;
      li = ni;
      n += ni;
    }
  }
}

// triBlockDiagLTransposeSolve (triBlockDiagLLT.cpp:98-158)
__global__ __launch_bounds__(64) void tri_ltsolve_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int end_in)
{
  const int lane = threadIdx.x, pb = blockIdx.x;
  double * Mp = M + (size_t)pb * D.mstride;
  int ntot = 0;
  for(int k = 0; k < D.nblk; ++k) ntot += D.bs[k];
  const int end = (end_in < 0) ? ntot : end_in;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mc = Mp + (size_t)c * ldm;
    int n = ntot, l = 0, li = 0;
    bool zero = true;
    for(int i = D.nblk - 1; i >= 0; --i)
    {
      const int ni = D.bs[i];
      const double * Li = D.diag[(size_t)pb * D.nblk + i];
      const int ldd = D.ldd[i];
      if(n - ni < end)
      {
        if(zero)
        {
          const int r = end - n + ni;
          double xi = (lane < r) ? Mc[n - ni + lane] : 0.0;
          xi = wave_bwd(Li, ldd, r, xi, lane);
          if(lane < r) Mc[n - ni + lane] = xi;
          zero = false;
        }
        else
        {
          const double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
          const int ldsg = D.lds[i];
          double acc = (lane < ni) ? Mc[n - ni + lane] : 0.0;
          if(lane < ni)
            for(int k = li - 1; k >= 0; --k) acc = FMA(-AT(Sg, ldsg, k, lane), Mc[(l - li) + k], acc);
          acc = wave_bwd(Li, ldd, ni, acc, lane);
          if(lane < ni) Mc[n - ni + lane] = acc;
        }
        __threadfence_block();
        wave_sync();
      }
      l = n;
      li = ni;
      n -= ni;
    }
  }
}

inline size_t lds_arrow_block(int maxb)
{
  return sizeof(double) * 2 * (size_t)maxb * (maxb | 1);
}
} // namespace

hipError_t launch_block_arrow_llt(const BlockDesc & d, int nprob, int maxb, hipStream_t st)
{
  hipError_t e = hipMemsetAsync(d.fail, 0, sizeof(int) * nprob, st);
  if(e != hipSuccess) return e;
  const size_t lds_blk = lds_arrow_block(maxb) + sizeof(double) * 64;
  const size_t lds_head = sizeof(double) * ((size_t)maxb * (maxb | 1) + 64 * 65);
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&arrow_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_blk);
  if(e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&arrow_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_head);
  if(e != hipSuccess) return e;
  if(d.nblk > 1) hipLaunchKernelGGL(arrow_block_kernel, dim3(d.nblk - 1, nprob), dim3(kTriThreads), lds_blk, st, d);
  hipLaunchKernelGGL(arrow_head_kernel, dim3(nprob), dim3(256), lds_head, st, d);
  return hipGetLastError();
}

hipError_t launch_block_arrow_solve(const BlockDesc & d, int nprob, bool transpose, double * M, int ldm, int nrhs, int start,
                                    int end, double * tmp, hipStream_t st)
{
  int ntot = 0;
  for(int k = 0; k < d.nblk; ++k) ntot += d.h_bs[k];
  if(end < 0) end = ntot;
  const int n0 = d.h_bs[0];
  if(!transpose)
  {
    if(d.up)
    {
      hipLaunchKernelGGL(permute_rows_kernel, dim3(1, nprob), dim3(256), 0, st, M, ldm, nrhs, ntot, n0, 0, tmp, d.mstride);
      start = start - n0 > 0 ? start - n0 : 0;
      end = end - n0 > 0 ? end - n0 : 0;
    }
    if(d.nblk > 1) hipLaunchKernelGGL(arrow_lsolve_blocks_kernel, dim3(d.nblk - 1, nprob), dim3(64), 0, st, d, M, ldm, nrhs, start, end);
    hipLaunchKernelGGL(arrow_lsolve_head_kernel, dim3(nprob), dim3(64), 0, st, d, M, ldm, nrhs, start, end);
  }
  else
  {
    hipLaunchKernelGGL(arrow_ltsolve_head_kernel, dim3(nprob), dim3(64), 0, st, d, M, ldm, nrhs, start, end);
    if(d.nblk > 1) hipLaunchKernelGGL(arrow_ltsolve_blocks_kernel, dim3(d.nblk - 1, nprob), dim3(64), 0, st, d, M, ldm, nrhs, start, end);
    if(d.up) hipLaunchKernelGGL(permute_rows_kernel, dim3(1, nprob), dim3(256), 0, st, M, ldm, nrhs, ntot, n0, 1, tmp, d.mstride);
  }
  return hipGetLastError();
}

hipError_t launch_tri_llt(const BlockDesc & d, int nprob, int maxb, hipStream_t st)
{
  hipError_t e = hipMemsetAsync(d.fail, 0, sizeof(int) * nprob, st);
  if(e != hipSuccess) return e;
  const size_t lds = lds_arrow_block(maxb) / 2 * 3 + sizeof(double) * 64;
  // blocks of at most 48 rows: the register-resident kernel (its LDS must also cover the barrier kernel's, for the hand-over).
  // It is the LATENCY kernel (one problem: 26 instead of 40 us per block of 30): with 200-300 VGPRs it holds one workgroup per
  // SIMD group, so that a batch that already fills the chip with the 256-thread barrier kernel stays there (measured: MultiIK
  // 9 x 43 at batch 1024 251 k vs 229 k factorisation+solves/s; 100 x 30 at batch 64 65 vs 43 us per problem).
  if(maxb <= 48 && (nprob <= 512 || maxb <= 32))
  {
    const int nbk = (maxb <= 32) ? 32 : 48;
    const size_t l2 = sizeof(double) * 3 * (size_t)nbk * (nbk | 1);
    const size_t lfast = l2 > lds ? l2 : lds;
    const void * fn = (nbk == 32) ? reinterpret_cast<const void *>(&tri_llt_fast_kernel<32>) : reinterpret_cast<const void *>(&tri_llt_fast_kernel<48>);
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lfast);
    if(e != hipSuccess) return e;
    if(nbk == 32)
      hipLaunchKernelGGL(tri_llt_fast_kernel<32>, dim3(nprob), dim3(kTriThreads), lfast, st, d);
    else
      hipLaunchKernelGGL(tri_llt_fast_kernel<48>, dim3(nprob), dim3(kTriThreads), lfast, st, d);
    return hipGetLastError();
  }
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tri_llt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if(e != hipSuccess) return e;
  hipLaunchKernelGGL(tri_llt_kernel, dim3(nprob), dim3(kTriThreads), lds, st, d);
  return hipGetLastError();
}

hipError_t launch_tri_solve(const BlockDesc & d, int nprob, bool transpose, double * M, int ldm, int nrhs, int start_or_end,
                            hipStream_t st)
{
  if(!transpose)
    hipLaunchKernelGGL(tri_lsolve_kernel, dim3(nprob), dim3(64), 0, st, d, M, ldm, nrhs, start_or_end);
  else
    hipLaunchKernelGGL(tri_ltsolve_kernel, dim3(nprob), dim3(64), 0, st, d, M, ldm, nrhs, start_or_end);
  return hipGetLastError();
}
} // namespace jrlqp
