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
#include <algorithm>
#include <climits>

#include "device_common.h"
#include "structured.h"

#include <cstdlib>

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
__device__ void arrow_block_body(const BlockDesc & D, double * sm)
{
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
  double * diagL = Ls + (size_t)ni * ldl;
  double * Xs = diagL + ((ni + 1) & ~1);
  // The side block goes through LDS in PANELS that fit behind L (round 4: blocks of up to 128 rows -- L alone is 132 KB then): rows of
  // side[i] for the down arrow (X <- X L^-T: a row meets only itself), columns for the up arrow (X <- L^-1 X: so does a column).  Up
  // to 64 rows everything is one panel, as before.
  const int prow = D.up ? ni : min(nbk, max(1, D.xs_cap / ni - 1));               // rows of a panel
  const int pcol = D.up ? min(nbk, max(1, D.xs_cap / (ni | 1))) : ni;             // columns of a panel
  const int ldx = prow | 1;
  load_block(Ls, ldl, Dg, ldd, ni, ni, tid, kTriThreads);
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
    for(int r0 = 0; r0 < nbk; r0 += prow) // side[i] <- side[i] L_i^-T   (:71-72), side[i]: nbk x ni
    {
      const int rows = min(prow, nbk - r0);
      __syncthreads();
      load_block(Xs, ldx, Sg + r0, ldsg, rows, ni, tid, kTriThreads);
      __syncthreads();
      block_trsm_right_lt(Ls, ldl, diagL, ni, Xs, ldx, rows, tid);
      for(int e = tid; e < rows * ni; e += kTriThreads) AT(Sg, ldsg, r0 + e % rows, e / rows) = AT(Xs, ldx, e % rows, e / rows);
    }
  }
  else
  {
    for(int c0 = 0; c0 < nbk; c0 += pcol) // side[i] <- L_i^-1 side[i]; it then stores B_i^T   (:68-70), side[i]: ni x nbk
    {
      const int cols = min(pcol, nbk - c0);
      __syncthreads();
      load_block(Xs, ldx, Sg + (size_t)c0 * ldsg, ldsg, ni, cols, tid, kTriThreads);
      __syncthreads();
      block_trsm_left(Ls, ldl, diagL, ni, Xs, ldx, cols, tid);
      for(int e = tid; e < ni * cols; e += kTriThreads) AT(Sg, ldsg, e % ni, c0 + e / ni) = AT(Xs, ldx, e % ni, e / ni);
    }
  }
}

__global__ __launch_bounds__(kTriThreads) void arrow_block_kernel(const BlockDesc D)
{
  extern __shared__ double sm[];
  arrow_block_body(D, sm);
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

// The head block on the matrix cores (round 5; heads of at most 64 rows): D_b -= sum_i B_i B_i^T with v_mfma_f64_16x16x4_f64, one wave
// per 16 x 16 tile of the lower triangle, the blocks in order and k ascending inside a block (4 per instruction: the instruction
// accumulates in k order, tests/test_gpu_arith.py) -- per element the SAME single fma chain as arrow_head_kernel's, which walked the
// blocks with two barriers and an LDS round trip each (550 us for 100 blocks of 30 rows; one problem's whole factorisation was 600 us).
// Four blocks' operands (64 loads per lane) are requested before their 32 instructions run; nothing is carried over the loop's edge.
typedef double double4_h __attribute__((ext_vector_type(4)));
template<int J, int NB>
__device__ __forceinline__ void llt_columns(double (&a)[NB], int & ret, int lane, double * Ls, int * ready, int base, double piv, double ljj, double lq); // (below)
constexpr int kHeadRingMaxBlocks = 2048; // side blocks whose descriptors the ring-fed head kernel keeps in LDS (16 bytes each)
__global__ __launch_bounds__(256) void arrow_head_mfma_kernel(const BlockDesc D)
{
  extern __shared__ double sm[]; // Ls (nbk x ldl)
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6;
  const int pb = blockIdx.x;
  const int ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int nbk = D.bs[ib];
  double * Dg = D.diag[(size_t)pb * D.nblk + ib];
  const int ldd = D.ldd[ib];
  const int ldl = nbk | 1;
  double * Ls = sm;
  if(D.fail[pb]) return; // a block already failed: the reference returns false before touching D_b further
  const int nt = (nbk + 15) >> 4, ntile = nt * (nt + 1) / 2;
  // ---- the side blocks through an LDS ring (round 5): arrow pointing down, blocks of at most 32 rows with an even number of rows,
  // even leading dimensions and 16-byte aligned blocks -- packed storage and the reference's views into an even-sized matrix alike.
  // The loop below asks for four blocks' operands, waits for them (~3 us), runs 32 instructions (0.7 us), 25 times over: 100 of the
  // kernel's 135 us at c5.  Here wave 3 -- no tile of its own: a head of at most 32 rows has three -- streams the blocks into a
  // ring of four slots of two blocks with LDS-direct loads (16 bytes per lane, eight instructions per block whatever its size: the
  // s_waitcnt constants count them), three slots in flight; waves 0-2 take their operands out of LDS.  One LDS-only barrier per
  // slot.  Per element still the one fma chain: the blocks in order, k ascending.
  __shared__ int s_ring_ok;
  {
    if(tid == 0) s_ring_ok = (!D.up && nbk <= 32 && (nbk & 1) == 0 && ntile <= 3 && D.nblk > 1 && D.nblk - 1 <= kHeadRingMaxBlocks) ? 1 : 0;
    __syncthreads();
    if(s_ring_ok)
    {
      int bad = 0;
      for(int i = tid; i < D.nblk - 1; i += 256)
      {
        const int ni = D.bs[dsel(D.nblk, D.up, i)];
        const size_t ad = reinterpret_cast<size_t>(D.side[(size_t)pb * (D.nblk - 1) + i]);
        bad |= (ni > 32) || (D.lds[i] & 1) || (ad & 15);
      }
      if(bad) s_ring_ok = 0; // (benign race: every writer stores 0)
    }
    __syncthreads();
    if(s_ring_ok)
    {
      constexpr int HB = 2, HSLOTS = 4, HNI = 8;          // blocks per slot, slots, load instructions per block
      constexpr int HIMG = HNI * 128;                     // doubles per block image (8 KiB: pieces of 16 bytes, 64 per instruction)
      constexpr int HSLOT = HB * HIMG;
      double * ring = sm + (size_t)nbk * ldl + ((nbk * ldl) & 1); // (16-byte aligned behind Ls)
      const int nside = D.nblk - 1, nchunk = (nside + HB - 1) / HB;
      // the blocks' descriptors in LDS: fetched inside the loader's loop they are vector loads -- in the same in-order queue as the
      // ring's, so every slot would wait for everything in flight
      const double ** m_ptr = reinterpret_cast<const double **>(ring + HSLOTS * HSLOT);
      int * m_ni = reinterpret_cast<int *>(m_ptr + nside);
      int * m_ld = m_ni + nside;
      for(int i = tid; i < nside; i += 256)
      {
        m_ptr[i] = D.side[(size_t)pb * nside + i];
        m_ni[i] = D.bs[dsel(D.nblk, D.up, i)];
        m_ld[i] = D.lds[i];
      }
      __syncthreads();
      const int ppc = nbk >> 1, ldi = nbk; // 16-byte pieces per column of a block; leading dimension of its image
      if(wave == 3)
      {
        // piece e = lane + 64 s of a block: column e / ppc, piece e % ppc of that column (the same for every block)
        int pcol[HNI], poff[HNI];
#pragma unroll
        for(int q = 0; q < HNI; ++q)
        {
          const int e = l + 64 * q;
          pcol[q] = e / ppc;
          poff[q] = 2 * (e - pcol[q] * ppc);
        }
        auto issue = [&](int k) {
          double * slot = ring + (k % HSLOTS) * HSLOT;
#pragma unroll
          for(int g = 0; g < HB; ++g)
          {
            const int i = min(k * HB + g, nside - 1);
            const int ni = m_ni[i];
            const double * Sg = m_ptr[i];
            const int ldsg = m_ld[i];
#pragma unroll
            for(int q = 0; q < HNI; ++q)
            {
              const double * p = Sg + (size_t)min(pcol[q], ni - 1) * ldsg + poff[q]; // (columns past the block: its last one, never read)
              __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)(slot + g * HIMG + q * 128), 16, 0, 0);
            }
          }
        };
        for(int k = 0; k < 3 && k < nchunk; ++k) issue(k);
        for(int k = 0; k < nchunk; ++k)
        {
          // slot k has landed when at most the loads of the slots behind it are outstanding (HB * HNI = 16 each, in order)
          if(k + 2 < nchunk)
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
          else if(k + 1 < nchunk)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          lds_barrier(); // B_k
          if(k + 3 < nchunk) issue(k + 3); // (slot (k + 3) mod 4 was last read for slot k - 1: everybody has passed B_k)
        }
      }
      else
      {
        const int t = wave; // (ntile <= 3)
        const bool work = t < ntile;
        int ti = 0, rem = t;
        while(rem > ti)
        {
          rem -= ti + 1;
          ++ti;
        }
        const int tj = rem;
        const int i0 = ti * 16, c0 = tj * 16;
        const int col = c0 + (l & 15);
        double4_h acc;
#pragma unroll
        for(int r = 0; r < 4; ++r)
        {
          const int row = i0 + (l >> 4) + 4 * r;
          acc[r] = (work && row < nbk && col < nbk && row >= col) ? AT(Dg, ldd, row, col) : 0.0;
        }
        const int arow = i0 + (l & 15), brow = c0 + (l & 15), kk = l >> 4;
        const int ar = min(arow, nbk - 1), br = min(brow, nbk - 1);
        const bool ain = arow < nbk, bin = brow < nbk;
        for(int k = 0; k < nchunk; ++k)
        {
          lds_barrier(); // B_k: slot k is in LDS
          if(!work) continue;
          const double * slot = ring + (k % HSLOTS) * HSLOT;
#pragma unroll
          for(int g = 0; g < HB; ++g)
          {
            const int i = k * HB + g;
            if(i < nside) // (uniform)
            {
              const int ni = m_ni[i];
              const double * img = slot + g * HIMG;
              double av[8], bv[8];
#pragma unroll
              for(int k8 = 0; k8 < 8; ++k8)
              {
                const int kq = min(4 * k8 + kk, ni - 1);
                av[k8] = img[ar + kq * ldi];
                bv[k8] = img[br + kq * ldi];
              }
#pragma unroll
              for(int k8 = 0; k8 < 8; ++k8)
              {
                if(4 * k8 < ni) // (uniform)
                {
                  const bool kin = 4 * k8 + kk < ni;
                  const double a = (kin && ain) ? -av[k8] : 0.0;
                  const double b = (kin && bin) ? bv[k8] : 0.0;
                  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
              }
            }
          }
        }
        if(work)
        {
#pragma unroll
          for(int r = 0; r < 4; ++r)
          {
            const int row = i0 + (l >> 4) + 4 * r;
            if(row < nbk && col < nbk && row >= col) AT(Ls, ldl, row, col) = acc[r];
          }
        }
      }
    }
  }
  if(!s_ring_ok) // (uniform; the ring needs no barrier of its own here: the one below orders Ls)
  for(int t = wave; t < ntile; t += 4)
  {
    int ti = 0, rem = t;
    while(rem > ti) // tile (ti, tj), tj <= ti, numbered row by row
    {
      rem -= ti + 1;
      ++ti;
    }
    const int tj = rem;
    const int i0 = ti * 16, c0 = tj * 16;
    const int col = c0 + (l & 15);
    double4_h acc;
#pragma unroll
    for(int r = 0; r < 4; ++r)
    {
      const int row = i0 + (l >> 4) + 4 * r;
      acc[r] = (row < nbk && col < nbk && row >= col) ? AT(Dg, ldd, row, col) : 0.0;
    }
    const int arow = i0 + (l & 15), brow = c0 + (l & 15), kk = l >> 4;
    const int ar = min(arow, nbk - 1), br = min(brow, nbk - 1);
    const bool ain = arow < nbk, bin = brow < nbk;
    constexpr int GB = 4; // blocks per trip
    for(int g0 = 0; g0 < D.nblk - 1; g0 += GB)
    {
      double av[GB][8], bv[GB][8];
      int nis[GB];
#pragma unroll
      for(int g = 0; g < GB; ++g)
      {
        const int i = min(g0 + g, D.nblk - 2);
        const int ni = D.bs[dsel(D.nblk, D.up, i)];
        nis[g] = (g0 + g < D.nblk - 1) ? ni : 0;
        const double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
        const int ldsg = D.lds[i];
#pragma unroll
        for(int k8 = 0; k8 < 8; ++k8)
        {
          const int kq = min(4 * k8 + kk, ni - 1);
          av[g][k8] = D.up ? AT(Sg, ldsg, kq, ar) : AT(Sg, ldsg, ar, kq);
          bv[g][k8] = D.up ? AT(Sg, ldsg, kq, br) : AT(Sg, ldsg, br, kq);
        }
      }
#pragma unroll
      for(int g = 0; g < GB; ++g)
      {
        const int ni = nis[g];
#pragma unroll
        for(int k8 = 0; k8 < 8; ++k8)
        {
          if(4 * k8 < ni) // (uniform)
          {
            const bool kin = 4 * k8 + kk < ni;
            const double a = (kin && ain) ? -av[g][k8] : 0.0;
            const double b = (kin && bin) ? bv[g][k8] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
          }
        }
        // a block of more than 32 columns: the rest, 4 at a time
        for(int k = 32; k < ni; k += 4)
        {
          const int i = g0 + g;
          const double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
          const int ldsg = D.lds[i];
          const int kq = min(k + kk, ni - 1);
          const bool kin = k + kk < ni;
          const double a0 = D.up ? AT(Sg, ldsg, kq, ar) : AT(Sg, ldsg, ar, kq);
          const double b0 = D.up ? AT(Sg, ldsg, kq, br) : AT(Sg, ldsg, br, kq);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64((kin && ain) ? -a0 : 0.0, (kin && bin) ? b0 : 0.0, acc, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for(int r = 0; r < 4; ++r)
    {
      const int row = i0 + (l >> 4) + 4 * r;
      if(row < nbk && col < nbk && row >= col) AT(Ls, ldl, row, col) = acc[r];
    }
  }
  __syncthreads();
  if(s_ring_ok)
  {
    // (heads of at most 32 rows: the factorisation in registers, as arrow_block_fast_kernel's -- the padded 32 x 33 factor goes to the
    // ring, which is free now; a pivot failure leaves Ls as it was and wave_llt below reproduces what the reference leaves behind)
    __shared__ int s_rdy, s_ret2;
    double * L2 = sm + (size_t)nbk * ldl + ((nbk * ldl) & 1);
    if(tid == 0) s_rdy = 0;
    if(wave == 0)
    {
      const bool row = l < nbk;
      const int rl = row ? l : 0;
      double a[32];
#pragma unroll
      for(int c = 0; c < 32; ++c)
      {
        const int cc = (c < nbk) ? c : 0;
        const double v = (rl >= cc) ? AT(Ls, ldl, rl, cc) : AT(Ls, ldl, cc, rl); // (Ls holds the lower triangle: a row's upper part by symmetry)
        a[c] = (row && c < nbk) ? v : ((l == c) ? 1.0 : 0.0); // identity padding
      }
      int ret = -1;
      const double p0 = bcast_lane(a[0], 0), l0 = __builtin_sqrt(p0);
      llt_columns<0, 32>(a, ret, l, L2, &s_rdy, 0, p0, l0, a[0] / l0);
      if(l == 0) s_ret2 = ret;
    }
    __syncthreads();
    if(s_ret2 < 0)
    {
      for(int e = tid; e < nbk * nbk; e += 256)
      {
        const int r = e % nbk, c = e / nbk;
        if(r >= c) AT(Dg, ldd, r, c) = L2[r + c * 33];
      }
      return;
    }
  }
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

// The head block when a block has more than 64 rows (round 4; up to 128): 64 lower-triangle elements per thread, ONE LDS region of
// maxb x (maxb | 1) doubles that holds the current side block while the products are accumulated and the head block afterwards
// (block_llt, 256 threads, right-looking -- per element the same chain as wave_llt's column-by-column order).  On a failure at column
// j the finished columns are written back and the columns from j on keep their rank-updated values, as the reference's in-place LLT
// leaves them.
__global__ __launch_bounds__(256) void arrow_head_big_kernel(const BlockDesc D)
{
  extern __shared__ double sm[];
  const int tid = threadIdx.x;
  const int pb = blockIdx.x;
  const int ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int nbk = D.bs[ib];
  double * Dg = D.diag[(size_t)pb * D.nblk + ib];
  const int ldd = D.ldd[ib];
  const int ldl = nbk | 1;
  if(D.fail[pb]) return; // a block already failed: the reference returns false before touching D_b further
  constexpr int SL = 64;
  double acc[SL];
#pragma unroll
  for(int s = 0; s < SL; ++s)
  {
    const int e = tid + 256 * s;
    const int r = e % nbk, c = e / nbk;
    acc[s] = (e < nbk * nbk && r >= c) ? AT(Dg, ldd, r, c) : 0.0;
  }
  for(int i = 0; i < D.nblk - 1; ++i)
  {
    const int id = dsel(D.nblk, D.up, i);
    const int ni = D.bs[id];
    const double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
    const int ldsg = D.lds[i];
    const int rows = D.up ? ni : nbk, cols = D.up ? nbk : ni; // stored shape of side[i]
    const int ldb = rows | 1;
    __syncthreads();
    load_block(sm, ldb, Sg, ldsg, rows, cols, tid, 256);
    __syncthreads();
#pragma unroll
    for(int s = 0; s < SL; ++s)
    {
      const int e = tid + 256 * s;
      const int r = e % nbk, c = e / nbk;
      if(e < nbk * nbk && r >= c)
      {
        if(!D.up) // B_i(r,k) B_i(c,k)
          acc[s] = chain2_neg(0, 1, ni, acc[s], [&](int k) { return sm[r + k * ldb]; }, [&](int k) { return sm[c + k * ldb]; });
        else // side stores B_i^T
          acc[s] = chain2_neg(0, 1, ni, acc[s], [&](int k) { return sm[k + r * ldb]; }, [&](int k) { return sm[k + c * ldb]; });
      }
    }
  }
  __syncthreads();
  double * Ls = sm;
  double * diagL = sm + (size_t)nbk * ldl;
#pragma unroll
  for(int s = 0; s < SL; ++s)
  {
    const int e = tid + 256 * s;
    const int r = e % nbk, c = e / nbk;
    if(e < nbk * nbk && r >= c) AT(Ls, ldl, r, c) = acc[s];
  }
  __syncthreads();
  const int ret = block_llt(Ls, ldl, nbk, diagL, tid);
  __syncthreads();
  if(ret > 0 && tid == 0) atomicExch(&D.fail[pb], 1); // (:80) ret <= 0 is success
  const int done = (ret < 0) ? nbk : ret;
#pragma unroll
  for(int s = 0; s < SL; ++s)
  {
    const int e = tid + 256 * s;
    const int r = e % nbk, c = e / nbk;
    if(e < nbk * nbk && r >= c) AT(Dg, ldd, r, c) = (c >= done) ? acc[s] : ((r == c) ? diagL[c] : AT(Ls, ldl, r, c));
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
  const int n0 = D.off[i];
  int s = max(start - n0, 0);
  if(ni < s || end <= n0) return; // M_i is zero
  const double * Dg = D.diag[(size_t)pb * D.nblk + id];
  const int ldd = D.ldd[id];
  const double * Lsub = Dg + s + (size_t)s * ldd; // bottomRightCorner(ni - s, ni - s)
  double * Mp = M + (size_t)pb * D.mstride;
  if(ni - s <= 32)
  {
    // (round 5) the lane's row of L, its diagonal entry and its right-hand side entry requested together, the substitution on registers:
    // one memory round trip instead of one per group of steps
    const int nr = ni - s, rl = min(lane, max(nr - 1, 0));
    double lv[32];
#pragma unroll
    for(int e = 0; e < 32; ++e) lv[e] = (nr > 0) ? AT(Lsub, ldd, rl, min(e, nr - 1)) : 0.0;
    const double dg = (nr > 0) ? AT(Lsub, ldd, rl, rl) : 1.0;
    for(int c = 0; c < nrhs; ++c)
    {
      double * Mi = Mp + n0 + s + (size_t)c * ldm;
      double xa = (lane < nr) ? Mi[lane] : 0.0;
      xa = wave_fwd_pre<32>(lv, dg, Lsub, ldd, nr, xa, lane);
      if(lane < nr) Mi[lane] = xa;
    }
    return;
  }
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mi = Mp + n0 + s + (size_t)c * ldm;
    const int nr = ni - s;
    double xa = (lane < nr) ? Mi[lane] : 0.0, xb = (lane + 64 < nr) ? Mi[lane + 64] : 0.0; // (rows lane, lane + 64: blocks of up to 128 rows)
    wave_fwd2(Lsub, ldd, nr, xa, xb, lane);
    if(lane < nr) Mi[lane] = xa;
    if(lane + 64 < nr) Mi[lane + 64] = xb;
  }
}

// phase 2: M_b -= sum_i B_i(:, s:) X_i (one thread per (row, rhs) of M_b, blocks in order), then X_b = L_b^-1 M_b
__global__ __launch_bounds__(64) void arrow_lsolve_head_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start, int end)
{
  const int lane = threadIdx.x, pb = blockIdx.x;
  const int ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int nbk = D.bs[ib];
  const int ntot = D.ntot;
  double * Mp = M + (size_t)pb * D.mstride;
  const double * Lb = D.diag[(size_t)pb * D.nblk + ib];
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mb = Mp + (ntot - nbk) + (size_t)c * ldm;
    double acc[2]; // rows lane, lane + 64 of the head block
#pragma unroll
    for(int h = 0; h < 2; ++h) acc[h] = (lane + 64 * h < nbk) ? Mb[lane + 64 * h] : 0.0;
    int n0 = 0;
    for(int i = 0; i < D.nblk - 1; ++i)
    {
      const int id = dsel(D.nblk, D.up, i);
      const int ni = D.bs[id];
      int s = max(start - n0, 0);
      if(!(ni < s || end <= n0))
      {
        const double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
        const int ldsg = D.lds[i];
        const double * Xi = Mp + n0 + (size_t)c * ldm;
#pragma unroll
        for(int h = 0; h < 2; ++h)
        {
          const int r = lane + 64 * h;
          if(r < nbk)
            for(int k = s; k < ni; ++k)
            {
              double bv = D.up ? AT(Sg, ldsg, k, r) : AT(Sg, ldsg, r, k);
              acc[h] = FMA(-bv, Xi[k], acc[h]);
            }
        }
      }
      n0 += ni;
    }
    wave_fwd2(Lb, D.ldd[ib], nbk, acc[0], acc[1], lane);
#pragma unroll
    for(int h = 0; h < 2; ++h)
      if(lane + 64 * h < nbk) Mb[lane + 64 * h] = acc[h];
  }
}

// phase 2 for blocks of at most 64 rows (round 5): the chain of a head row runs over every entry of every side block in order --
// 3000 dependent fmas at c5 -- and arrow_lsolve_head_kernel fed it one global load per term (433 us).  Here wave 0 only chains, out
// of LDS; waves 1 .. 3 stage the NEXT group of side blocks (and the solved X_i that go with them) while it does: one memory round
// trip per GROUP of blocks.  Same terms in the same order: same bits.
__global__ __launch_bounds__(256) void arrow_lsolve_head_staged_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start, int end, int gb,
                                                                         int bcap)
{
  extern __shared__ double sm[]; // [2][gb][bcap] side blocks | [2][gb][64] X_i
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pb = blockIdx.x;
  const int ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int nbk = D.bs[ib];
  const int ntot = D.ntot;
  double * Mp = M + (size_t)pb * D.mstride;
  const double * Lb = D.diag[(size_t)pb * D.nblk + ib];
  double * Bs = sm;
  double * Xs = sm + 2 * (size_t)gb * bcap;
  const int nside = D.nblk - 1, ngroup = (nside + gb - 1) / gb;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mb = Mp + (ntot - nbk) + (size_t)c * ldm;
    double acc = (wave == 0 && lane < nbk) ? Mb[lane] : 0.0;
    // stage group g into buffer g & 1 (threads t0 .. 255 step nthr)
    auto stage = [&](int g, int t0, int nthr) {
      double * Bg = Bs + (size_t)(g & 1) * gb * bcap;
      double * Xg = Xs + (size_t)(g & 1) * gb * 64;
      int n0 = D.off[min(g * gb, D.nblk)];
      for(int j = 0; j < gb; ++j)
      {
        const int i = g * gb + j;
        if(i >= nside) break;
        const int id = dsel(D.nblk, D.up, i), ni = D.bs[id];
        const double * Sg = D.side[(size_t)pb * nside + i];
        const int ldsg = D.lds[i];
        const int rows = D.up ? ni : nbk, cols = D.up ? nbk : ni, ldb = rows | 1;
        for(int e = tid - t0; e < rows * cols; e += nthr) Bg[(size_t)j * bcap + (e % rows) + (e / rows) * ldb] = AT(Sg, ldsg, e % rows, e / rows);
        const double * Xi = Mp + n0 + (size_t)c * ldm;
        for(int e = tid - t0; e < ni; e += nthr) Xg[j * 64 + e] = Xi[e];
        n0 += ni;
      }
    };
    if(ngroup > 0) stage(0, 0, 256);
    __syncthreads();
    int n0 = 0;
    for(int g = 0; g < ngroup; ++g)
    {
      if(wave != 0)
      {
        if(g + 1 < ngroup) stage(g + 1, 64, 192);
      }
      else
      {
        const double * Bg = Bs + (size_t)(g & 1) * gb * bcap;
        const double * Xg = Xs + (size_t)(g & 1) * gb * 64;
        for(int j = 0; j < gb; ++j)
        {
          const int i = g * gb + j;
          if(i >= nside) break;
          const int id = dsel(D.nblk, D.up, i), ni = D.bs[id];
          const int s = max(start - n0, 0);
          if(!(ni < s || end <= n0) && lane < nbk)
          {
            const int rows = D.up ? ni : nbk, ldb = rows | 1;
            const double * B = Bg + (size_t)j * bcap;
            for(int k = s; k < ni; ++k)
            {
              const double bv = D.up ? B[k + lane * ldb] : B[lane + k * ldb];
              acc = FMA(-bv, Xg[j * 64 + k], acc);
            }
          }
          n0 += ni;
        }
      }
      __syncthreads();
    }
    if(wave == 0)
    {
      double a1 = 0.0;
      wave_fwd2(Lb, D.ldd[ib], nbk, acc, a1, lane);
      if(lane < nbk) Mb[lane] = acc;
    }
    __syncthreads();
  }
}

// L^T X = M, phase 1: head block X_b = L_b(0:r,0:r)^-T M_b(0:r) when end reaches into it (:203-212)
__global__ __launch_bounds__(64) void arrow_ltsolve_head_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start, int end)
{
  const int lane = threadIdx.x, pb = blockIdx.x;
  const int ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int nbk = D.bs[ib];
  const int ntot = D.ntot;
  if(!(end > ntot - nbk)) return;
  const int r = end - ntot + nbk;
  const double * Lb = D.diag[(size_t)pb * D.nblk + ib];
  double * Mp = M + (size_t)pb * D.mstride;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mb = Mp + (ntot - nbk) + (size_t)c * ldm;
    double xa = (lane < r) ? Mb[lane] : 0.0, xb = (lane + 64 < r) ? Mb[lane + 64] : 0.0;
    wave_bwd2(Lb, D.ldd[ib], r, xa, xb, lane);
    if(lane < r) Mb[lane] = xa;
    if(lane + 64 < r) Mb[lane + 64] = xb;
  }
}

// phase 2: per off-head block  M_i -= B_i^T X_b (descending k), then L_i^T solve (:217-251)
__global__ __launch_bounds__(64) void arrow_ltsolve_blocks_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start, int end)
{
  const int lane = threadIdx.x, i = blockIdx.x, pb = blockIdx.y;
  const int ib = dsel(D.nblk, D.up, D.nblk - 1);
  const int nbk = D.bs[ib];
  const int ntot = D.ntot;
  const bool zero = !(end > ntot - nbk);
  const int id = dsel(D.nblk, D.up, i);
  const int ni = D.bs[id];
  const int n0 = D.off[i];
  if(zero && start >= n0 + ni) return;
  const double * Li = D.diag[(size_t)pb * D.nblk + id];
  const int ldd = D.ldd[id];
  const double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
  const int ldsg = D.lds[i];
  double * Mp = M + (size_t)pb * D.mstride;
  if(ni <= 32 && nbk <= 32 && end >= n0 + ni)
  {
    // (round 5, blocks and head of at most 32 rows, the whole block inside the window) the lane's column of L_i and of B_i, the diagonal
    // entry, the head's solution and the right-hand side entry requested together; the chain and the substitution on registers
    const int rl = min(lane, ni - 1);
    double lv[32], bv[32];
#pragma unroll
    for(int e = 0; e < 32; ++e) lv[e] = AT(Li, ldd, max(ni - 1 - e, 0), rl);
#pragma unroll
    for(int e = 0; e < 32; ++e)
    {
      const int k = max(nbk - 1 - e, 0);
      bv[e] = D.up ? AT(Sg, ldsg, rl, k) : AT(Sg, ldsg, k, rl);
    }
    const double dg = AT(Li, ldd, rl, rl);
    for(int c = 0; c < nrhs; ++c)
    {
      double * Mi = Mp + n0 + (size_t)c * ldm;
      const double * Xb = Mp + (ntot - nbk) + (size_t)c * ldm;
      double xv[32];
#pragma unroll
      for(int e = 0; e < 32; ++e) xv[e] = Xb[max(nbk - 1 - e, 0)];
      double acc = (lane < ni) ? Mi[lane] : 0.0;
      if(!zero)
      {
#pragma unroll
        for(int e = 0; e < 32; ++e)
          if(e < nbk) // (uniform)
          {
            const double t = FMA(-bv[e], xv[e], acc);
            acc = (lane < ni) ? t : acc;
          }
      }
      acc = wave_bwd_pre<32>(lv, dg, Li, ldd, ni, acc, lane);
      if(lane < ni) Mi[lane] = acc;
    }
    return;
  }
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mi = Mp + n0 + (size_t)c * ldm;
    const double * Xb = Mp + (ntot - nbk) + (size_t)c * ldm;
    double acc[2]; // rows lane, lane + 64 of block i
#pragma unroll
    for(int h = 0; h < 2; ++h)
    {
      const int r = lane + 64 * h;
      acc[h] = (r < ni) ? Mi[r] : 0.0;
      if(!zero && r < ni)
        for(int k = nbk - 1; k >= 0; --k)
        {
          double bv = D.up ? AT(Sg, ldsg, r, k) : AT(Sg, ldsg, k, r);
          acc[h] = FMA(-bv, Xb[k], acc[h]);
        }
    }
    if(end >= n0)
    {
      const int r = (end >= n0 + ni) ? ni : (end - n0);
      double sa = acc[0], sb = acc[1];
      wave_bwd2(Li, ldd, r, sa, sb, lane);
      acc[0] = (lane < r) ? sa : acc[0];
      acc[1] = (lane + 64 < r) ? sb : acc[1];
    }
#pragma unroll
    for(int h = 0; h < 2; ++h)
      if(lane + 64 * h < ni) Mi[lane + 64 * h] = acc[h];
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

// The same chain when a block has more than 64 rows (round 4; up to 128): three blocks no longer fit in LDS, so there is ONE block
// region -- the diagonal block while it is factorised, then the finished sub-diagonal block while D_{i+1} receives its rank update --
// and a panel region through which S_i <- S_i L_i^-T passes in row panels (as in arrow_block_kernel).  D_{i+1} is updated in memory
// (read, chain out of LDS, written) and fetched again as the next diagonal block.  Per element the chains are tri_llt_from's.
__global__ __launch_bounds__(kTriThreads) void tri_llt_big_kernel(const BlockDesc D)
{
  extern __shared__ double sm[];
  const int tid = threadIdx.x, pb = blockIdx.x;
  int maxb = 0;
  for(int k = 0; k < D.nblk; ++k) maxb = max(maxb, D.bs[k]);
  double * Ls = sm;
  double * diagL = Ls + (size_t)maxb * (maxb | 1);
  double * Xs = diagL + ((maxb + 1) & ~1);
  for(int i = 0; i < D.nblk; ++i)
  {
    const int ni = D.bs[i];
    const int ldl = ni | 1;
    double * Dg = D.diag[(size_t)pb * D.nblk + i];
    const int ldd = D.ldd[i];
    const bool last = (i == D.nblk - 1);
    __syncthreads();
    load_block(Ls, ldl, Dg, ldd, ni, ni, tid, kTriThreads);
    __syncthreads();
    const int ret = block_llt(Ls, ldl, ni, diagL, tid);
    __syncthreads();
    const int cols = (ret < 0) ? ni : ret; // the factor goes back: all of the lower triangle on success, the finished columns on failure
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
    const int n1 = D.bs[i + 1];
    double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
    const int ldsg = D.lds[i];
    const int prow = min(n1, max(1, D.xs_cap / ni - 1)), ldx = prow | 1;
    for(int r0 = 0; r0 < n1; r0 += prow) // S_i <- S_i L_i^-T, S_i: n1 x ni
    {
      const int rows = min(prow, n1 - r0);
      __syncthreads();
      load_block(Xs, ldx, Sg + r0, ldsg, rows, ni, tid, kTriThreads);
      __syncthreads();
      block_trsm_right_lt(Ls, ldl, diagL, ni, Xs, ldx, rows, tid);
      for(int e = tid; e < rows * ni; e += kTriThreads) AT(Sg, ldsg, r0 + e % rows, e / rows) = AT(Xs, ldx, e % rows, e / rows);
    }
    // D_{i+1} -= S_i S_i^T (lower): the finished S_i comes back into the block region (this workgroup's own stores: visible behind
    // the fence and the barrier), one element per thread and round, terms in ascending k
    __threadfence_block();
    __syncthreads();
    const int lds1 = n1 | 1;
    load_block(Ls, lds1, Sg, ldsg, n1, ni, tid, kTriThreads);
    __syncthreads();
    double * Dn = D.diag[(size_t)pb * D.nblk + i + 1];
    const int ldn = D.ldd[i + 1];
    for(int c = (tid >> 4); c < n1; c += 16)
      for(int r = (tid & 15); r < n1; r += 16)
      {
        if(r < c) continue;
        AT(Dn, ldn, r, c) = chain2_neg(0, 1, ni, AT(Dn, ldn, r, c), [&](int k) { return AT(Ls, lds1, r, k); }, [&](int k) { return AT(Ls, lds1, c, k); });
      }
    __threadfence_block();
  }
}

__global__ __launch_bounds__(kTriThreads) void tri_llt_kernel(const BlockDesc D)
{
  extern __shared__ double sm[];
  tri_llt_from(D, 0, sm);
}

template<int J, int NB, int C = J + 1, bool FIRST = true>
__device__ __forceinline__ void column_update(double (&a)[NB], double v, double src)
{
  if constexpr(C < NB)
  {
    fma_neg_bcast<C, FIRST>(a[C], v, src);
    column_update<J, NB, C + 1, false>(a, v, src);
  }
}

// LDS hand-off words between the waves of the register kernel below.  Values only ever grow (they carry the block number), so
// a word never has to be reset.  A waiting wave naps; the posting wave wakes the sleepers of its workgroup.  Every wait is
// bounded: a lost hand-off would otherwise hang the launch.
__device__ __forceinline__ void tri_post(int * word, int value, int lane)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if(lane == 0) __atomic_store_n(word, value, __ATOMIC_RELAXED);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory");
}
// The same without the waits, for a wave that posts a word behind LDS stores of its OWN: a wave's LDS operations are carried
// out in the order it issued them, so the word cannot become visible before the data; the reader polls in short naps.
__device__ __forceinline__ void tri_post_inorder(int * word, int value, int lane)
{
  asm volatile("" ::: "memory");
  if(lane == 0) __atomic_store_n(word, value, __ATOMIC_RELAXED);
  asm volatile("" ::: "memory");
}
// waits until *word > than; returns the value seen, or INT_MIN after about a tenth of a second
__device__ __forceinline__ int tri_wait_gt(const int * word, int than)
{
  int spins = 0, v;
  while((v = __atomic_load_n(word, __ATOMIC_RELAXED)) <= than)
  {
    if(++spins > (1 << 20)) return INT_MIN;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return v;
}

// column J of the in-register factorisation (lane = row) and the columns after it.  A finished column goes to LDS at once
// (Ls, leading dimension NB | 1) and `ready` is raised to base + J + 1: the wave that computes S L^-T follows one column behind.
// LOOK-AHEAD: the square root and the division of column J + 1 (piv, ljj = sqrt(piv), lq = a[J] / ljj arrive from the column
// before) are started as soon as column J + 1 has received its term of column J, and the terms of the columns after it are
// issued in their latency -- per element the operations and their order do not change.
template<int J, int NB>
__device__ __forceinline__ void llt_columns(double (&a)[NB], int & ret, int lane, double * Ls, int * ready, int base, double piv, double ljj, double lq)
{
  if constexpr(J < NB)
  {
    if(piv <= 0.0)
    {
      ret = J;
      return;
    }
    const double lij = (lane == J) ? ljj : lq;
    a[J] = lij;
    if(lane >= J && lane < NB) Ls[lane + J * (NB | 1)] = lij;
    tri_post_inorder(ready, base + J + 1, lane);
    // The terms of the columns after J + 1 take L(c, J) back from LDS (one address for the wave: a broadcast; two issue slots
    // per term where the v_readlane unit takes six).  Order of issue: the reads, then the unit for column J + 1 and the square
    // root and division of that column -- in the latency of the reads --, then the terms.
    const double * col = Ls + J * (NB | 1);
    double lc[NB];
#pragma unroll
    for(int c = J + 2; c < NB; ++c) lc[c] = col[c];
    double pn = 0.0, ln = 0.0, qn = 0.0;
    if constexpr(J + 1 < NB)
    {
      fma_neg_bcast<J + 1, true>(a[J + 1], lij, lij); // a[J+1] -= L(lane, J) * L(J+1, J)  (guarded on both sides: the pivot
                                                      // is read out of a[J+1] with v_readlane right behind it)
      pn = bcast_lane(a[J + 1], J + 1);
      ln = __builtin_sqrt(pn); // (a non-positive pivot gives a NaN nobody uses)
      qn = a[J + 1] / ln;
      asm volatile("" : "+v"(ln), "+v"(qn)); // (computed HERE: left alone, the compiler sinks both into the next column's block)
    }
#pragma unroll
    for(int c = J + 2; c < NB; ++c) a[c] = FMA(-lij, lc[c], a[c]); // a[c] -= L(lane, J) * L(c, J)
#pragma unroll
    for(int c = J + 2; c < NB; ++c) asm volatile("" : "+v"(a[c])); // (pinned to this column, see trsm_columns)
    llt_columns<J + 1, NB>(a, ret, lane, Ls, ready, base, pn, ln, qn);
  }
}
// column J of X <- X L^-T (lane = row of X) and the columns after it; column J of L is read from LDS (one address for the whole
// wave: a broadcast read) as soon as the factorising wave has published it
template<int J, int NB>
__device__ __forceinline__ void trsm_columns(double (&x)[NB], const double * Ls, const int * ready, int base, bool & dead)
{
  if constexpr(J < NB)
  {
    if(!dead)
    {
      const int seen = tri_wait_gt(ready, base + J);
      if(seen > base + NB) // INT_MIN never gets here; base + NB + 1: the factorisation failed, base + NB + 2: timeout elsewhere
        dead = true;
      else if(seen == INT_MIN)
        dead = true;
      else
      {
        const double * col = Ls + J * (NB | 1);
        double lc[NB];
#pragma unroll
        for(int c = J; c < NB; ++c) lc[c] = col[c];
        const double xj = x[J] / lc[J];
        x[J] = xj;
#pragma unroll
        for(int c = J + 1; c < NB; ++c) x[c] = FMA(-xj, lc[c], x[c]); // x[c] -= x(J) * L(c, J)
        // The terms are pinned to their column: left alone, the compiler sinks every term to the column that consumes x[c]
        // -- left-looking, all of L live in registers, hundreds of spills.  (One pin per column END: a pin after every term
        // also serialises the LDS reads, 40 cycles per term.)
#pragma unroll
        for(int c = J + 1; c < NB; ++c) asm volatile("" : "+v"(x[c]));
      }
    }
    trsm_columns<J + 1, NB>(x, Ls, ready, base, dead);
  }
}

// ---- the same chain for blocks of at most NB rows, the serial part in REGISTERS, the waves in different roles -----------------
// One block costs the barrier-per-column kernel above ~40 us: sixty workgroup barriers around one division each.  Here
//   wave 0 keeps its row of the diagonal block in registers (lane = row, a[c] = D(lane, c)) and factorises it right-looking --
//          pivot and column entries L(c, j) move by v_readlane, no LDS round trip, no barrier inside a block -- and publishes
//          every finished column in LDS;
//   wave 1 keeps its row of the sub-diagonal block in registers and computes S <- S L^-T one column behind wave 0;
//   waves 2, 3 fetch S_i and D_{i+1} from memory meanwhile;
// then all four waves store L_i and S_i and apply the rank update D_{i+1} -= S S^T (n^2 independent chains) through LDS.
// Per element the operations and their order are those of block_llt / block_trsm_right_lt (and of the oracle): same bits.
// Blocks are padded to NB x NB with the identity (no per-element bounds tests): the padding factorises to the identity and
// contributes exact zeros.
// rows x cols block from memory to LDS by ONE wave, lane = row: every load of the block in flight at once (clamped addresses,
// no predicate), then the stores
template<int NB>
__device__ __forceinline__ void load_rows(double * dst, const double * src, int ld, int rows, int cols, int lane)
{
  double v[NB];
  const int r = min(lane, rows - 1);
#pragma unroll
  for(int c = 0; c < NB; ++c) v[c] = src[r + (size_t)min(c, cols - 1) * ld];
#pragma unroll
  for(int c = 0; c < NB; ++c)
    if(lane < rows && c < cols) dst[lane + c * (NB | 1)] = v[c];
}

template<int NB>
__global__ __launch_bounds__(kTriThreads) void tri_llt_fast_kernel(const BlockDesc D)
{
  extern __shared__ double sm[];
  const int tid = threadIdx.x, pb = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int ldl = NB | 1;
  int opaque_zero = 0; // (rewritten by an empty asm statement per block, see below)
  double * Ls = sm;                 // the diagonal block being factorised (with its rank update applied), then L_i
  double * Ns = Ls + NB * ldl;      // D_{i+1}
  double * Xa = Ns + NB * ldl;      // S_i, then S_i L_i^-T (blocks with even i)
  double * Xb = Xa + NB * ldl;      // the same for odd i: S_{i+1} is fetched while block i is factorised
  __shared__ int s_ret, s_ready, s_lost;
#ifdef TRI_PROF
  long long tp_acc[6] = {0, 0, 0, 0, 0, 0}, tp_t = __builtin_amdgcn_s_memtime();
#define TP(k) { const long long now_ = __builtin_amdgcn_s_memtime(); tp_acc[k] += now_ - tp_t; tp_t = now_; }
#else
#define TP(k)
#endif
  if(tid == 0)
  {
    s_ready = 0;
    s_lost = 0;
  }
  double ** const diag = D.diag + (size_t)pb * D.nblk;
  double ** const side = D.side + (size_t)pb * (D.nblk - 1);
  if(wave < 2)
    load_block(Ls, ldl, diag[0], D.ldd[0], D.bs[0], D.bs[0], tid, 128);
  else if(wave == 2 && D.nblk > 1)
    load_rows<NB>(Xa, side[0], D.lds[0], D.bs[1], D.bs[0], lane);
  __syncthreads();
  // the descriptors of a block are read one block ahead (scalar loads share their counter with the LDS operations: a load
  // issued where its value is needed stalls the first LDS wait behind it for a memory round trip)
  int ni = D.bs[0], ldd = D.ldd[0];
  double * Dg = diag[0];
  int n1 = 0, ldn = 1, ldsg = 1;
  double *Dn = nullptr, *Sg = nullptr;
  int n_p = 0, ldd_p = 1, ldsg_p = 1; // the previous block (its factor is stored one iteration late)
  double *Dg_p = nullptr, *Sg_p = nullptr;
  if(D.nblk > 1)
  {
    n1 = D.bs[1];
    ldn = D.ldd[1];
    Dn = diag[1];
    Sg = side[0];
    ldsg = D.lds[0];
  }
  for(int i = 0; i < D.nblk; ++i)
  {
    const bool last = (i == D.nblk - 1);
    const bool more = (i + 2 < D.nblk);
    // block i + 2 / side block i + 1 (used from the next iteration on)
    const int n2 = more ? D.bs[i + 2] : 0, ld2 = more ? D.ldd[i + 2] : 1, lds2 = more ? D.lds[i + 1] : 1;
    double * const D2 = more ? diag[i + 2] : nullptr;
    double * const S2 = more ? side[i + 1] : nullptr;
    double * const Xs = (i & 1) ? Xb : Xa;
    double * const Xn = (i & 1) ? Xa : Xb;
    const int base = i * (NB + 4);
    asm volatile("" : "+v"(opaque_zero));
    TP(5)
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
      TP(0)
      {
        // (the lane number made opaque per block: the per-column lane masks are otherwise hoisted out of the block loop, a
        // hundred and fifty scalar registers that live in spill lanes and come back with two v_readlane each)
        const int lz = lane + opaque_zero;
        const double p0 = bcast_lane(a[0], 0), l0 = __builtin_sqrt(p0);
        llt_columns<0, NB>(a, ret, lz, Ls, &s_ready, base, p0, l0, a[0] / l0);
      }
      TP(1)
      if(lane == 0) s_ret = ret;
      if(ret >= 0) tri_post(&s_ready, base + NB + 1, lane); // releases the wave behind
    }
    else if(wave == 1)
    {
      if(!last)
      {
        // S_i <- S_i L_i^-T, lane = row of S: x(j) = (x(j) - sum_{k<j} x(k) L(j,k)) / L(j,j), right-looking
        TP(0)
        const bool srow = lane < n1;
        const int sl = srow ? lane : 0;
        double x[NB];
#pragma unroll
        for(int c = 0; c < NB; ++c)
        {
          const double v = AT(Xs, ldl, sl, (c < ni) ? c : 0);
          x[c] = (srow && c < ni) ? v : 0.0;
        }
        bool dead = false;
        trsm_columns<0, NB>(x, Ls, &s_ready, base, dead);
        TP(1)
        if(!dead)
        {
          // (all NB x NB entries, the exact zeros of the padding included: the rank update below runs over NB terms)
#pragma unroll
          for(int c = 0; c < NB; ++c)
            if(lane < NB) AT(Xs, ldl, lane, c) = x[c];
        }
        else if(__atomic_load_n(&s_ready, __ATOMIC_RELAXED) <= base + NB)
        {
          if(lane == 0) s_lost = 1; // a hand-off never came
        }
      }
    }
    else
    {
      // waves 2 and 3: the factor of the PREVIOUS block leaves for memory now (its buffers are about to be refilled), then the
      // next blocks are fetched -- both beside the factorisation, not in the phase every wave waits for
      if(wave == 2)
      {
        if(i > 0 && lane < ni) // S_{i-1} L_{i-1}^-T: n_i x n_{i-1}
          for(int c = 0; c < n_p; ++c) AT(Sg_p, ldsg_p, lane, c) = AT(Xn, ldl, lane, c);
        if(more) load_rows<NB>(Xn, S2, lds2, n2, n1, lane); // S_{i+1}
      }
      else
      {
        if(i > 0 && lane < n_p) // L_{i-1}, lower triangle
          for(int c = 0; c <= lane; ++c) AT(Dg_p, ldd_p, lane, c) = AT(Ns, ldl, lane, c);
        if(!last) load_rows<NB>(Ns, Dn, ldn, n1, n1, lane); // D_{i+1}
      }
      TP(1)
    }
    lds_barrier(); // (LDS only: the stores of the previous block's factor may still be in flight)
    TP(2)
    const int ret = s_ret;
    if(s_lost)
    {
      if(tid == 0) D.fail[pb] = 1;
      return;
    }
    if(ret == 0) // (a failure at column 0 goes undetected in the reference: the barrier kernel's code reproduces what follows)
    {
      __syncthreads(); // block i as the previous rank update stored it is read back from memory
      tri_llt_from(D, i, sm);
      return;
    }
    // the factor goes to memory.  A failing block: its finished columns, now; the last block: now; any other block: by waves
    // 2 and 3 at the start of the next iteration (above).
    if(ret > 0 || last)
    {
      const int cols = (ret < 0) ? ni : ret;
      for(int c = (tid >> 4); c < cols; c += 16)
        for(int r = (tid & 15); r < ni; r += 16)
          if(r >= c) AT(Dg, ldd, r, c) = AT(Ls, ldl, r, c);
      if(ret > 0) // triBlockDiagLLT.cpp:19,32: the failure the reference reports
      {
        if(tid == 0) D.fail[pb] = 1;
        return;
      }
      break;
    }
    // D_{i+1} -= S_i S_i^T (lower), in LDS, terms in ascending k.  The n1 (n1 + 1) / 2 elements are dealt out evenly: column cc
    // (n1 - cc elements) and column n1 - 1 - cc (cc + 1 elements) form one row of n1 + 1 elements of a rectangle.  Every chain
    // runs over all NB columns of the padded S L^-T -- the padding contributes exact zeros -- so the loop unrolls into two LDS
    // reads with immediate offsets and one fma per term.
    {
      const int w = n1 + 1, tot = w * ((n1 + 1) >> 1);
      for(int e = tid; e < tot; e += kTriThreads)
      {
        const int cc = e / w, rr = e - cc * w;
        int r, c;
        if(rr < n1 - cc)
        {
          c = cc;
          r = cc + rr;
        }
        else
        {
          c = n1 - 1 - cc;
          r = c + (rr - (n1 - cc));
          if(c == cc) continue; // (the middle column of an odd n1 is its own partner)
        }
        const double * xr = Xs + r;
        const double * xc = Xs + c;
        double acc = AT(Ns, ldl, r, c);
#pragma unroll
        for(int k = 0; k < NB; ++k) acc = FMA(-xr[k * ldl], xc[k * ldl], acc);
        AT(Ns, ldl, r, c) = acc;
        AT(Dn, ldn, r, c) = acc; // (what a failing factorisation of block i+1 leaves behind)
      }
    }
    TP(3)
    lds_barrier();
    TP(4)
    double * t = Ls; // the updated next block becomes the current one
    Ls = Ns;
    Ns = t;
    n_p = ni;
    ldd_p = ldd;
    Dg_p = Dg;
    Sg_p = Sg;
    ldsg_p = ldsg;
    ni = n1;
    ldd = ldn;
    Dg = Dn;
    n1 = n2;
    ldn = ld2;
    Dn = D2;
    Sg = S2;
    ldsg = lds2;
  }
#ifdef TRI_PROF
  if(pb == 0 && lane == 0)
    printf("tri wave %d: [0] %lld [1] %lld [2:barrierB] %lld [3:update] %lld [4:barrierC] %lld [5] %lld (cycles)\n", wave, tp_acc[0], tp_acc[1], tp_acc[2],
           tp_acc[3], tp_acc[4], tp_acc[5]);
#endif
}

// Block arrow (pointing down), phase 1 for blocks and a head of at most NB rows (round 5): ONE step of the chain above -- wave 0
// factorises the diagonal block in registers (llt_columns), wave 1 follows one column behind with B_i = side[i] L_i^-T
// (trsm_columns), no barrier and no LDS round trip per column -- instead of block_llt / block_trsm_right_lt with their barrier per
// column (40 us per block; 15 us here).  Per element the operations and their order are the same.  A pivot failure (any column) or
// a lost hand-off: nothing has been stored yet, and the barrier kernel's body redoes the block from memory with the reference's
// failure semantics.
template<int NB>
__global__ __launch_bounds__(kTriThreads) void arrow_block_fast_kernel(const BlockDesc D)
{
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x, pb = blockIdx.y;
  const int ni = D.bs[i], nbk = D.bs[D.nblk - 1]; // (down arrow: logical = physical)
  double * Dg = D.diag[(size_t)pb * D.nblk + i];
  double * Sg = D.side[(size_t)pb * (D.nblk - 1) + i];
  const int ldd = D.ldd[i], ldsg = D.lds[i];
  constexpr int ldl = NB | 1;
  double * Ls = sm;
  double * Xs = Ls + NB * ldl;
  __shared__ int s_ret, s_ready, s_lost;
  if(tid == 0)
  {
    s_ready = 0;
    s_lost = 0;
    s_ret = -1;
  }
  if(wave < 2)
    load_block(Ls, ldl, Dg, ldd, ni, ni, tid, 128);
  else if(wave == 2)
    load_rows<NB>(Xs, Sg, ldsg, nbk, ni, lane);
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
    const double p0 = bcast_lane(a[0], 0), l0 = __builtin_sqrt(p0);
    llt_columns<0, NB>(a, ret, lane, Ls, &s_ready, 0, p0, l0, a[0] / l0);
    if(lane == 0) s_ret = ret;
    if(ret >= 0) tri_post(&s_ready, NB + 1, lane); // releases the wave behind
  }
  else if(wave == 1)
  {
    const bool srow = lane < nbk;
    const int sl = srow ? lane : 0;
    double x[NB];
#pragma unroll
    for(int c = 0; c < NB; ++c)
    {
      const double v = AT(Xs, ldl, sl, (c < ni) ? c : 0);
      x[c] = (srow && c < ni) ? v : 0.0;
    }
    bool dead = false;
    trsm_columns<0, NB>(x, Ls, &s_ready, 0, dead);
    if(!dead)
    {
#pragma unroll
      for(int c = 0; c < NB; ++c)
        if(lane < NB) AT(Xs, ldl, lane, c) = x[c];
    }
    else if(lane == 0)
      s_lost = 1; // the factorisation failed, or a hand-off never came: the barrier kernel's body takes over
  }
  __syncthreads();
  if(s_lost || s_ret >= 0)
  {
    __syncthreads();
    arrow_block_body(D, sm); // (nothing of this attempt has reached memory)
    return;
  }
  for(int e = tid; e < ni * ni; e += kTriThreads)
  {
    const int r = e % ni, c = e / ni;
    if(r >= c) AT(Dg, ldd, r, c) = AT(Ls, ldl, r, c);
  }
  for(int e = tid; e < nbk * ni; e += kTriThreads) AT(Sg, ldsg, e % nbk, e / nbk) = AT(Xs, ldl, e % nbk, e / nbk);
}

// triBlockDiagLSolve (triBlockDiagLLT.cpp:37-96): forward block substitution with the zero-window `start`
__global__ __launch_bounds__(64) void tri_lsolve_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start)
{
  const int lane = threadIdx.x, pb = blockIdx.x;
  double * Mp = M + (size_t)pb * D.mstride;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mc = Mp + (size_t)c * ldm;
    int n = 0, li = 0;
    bool zero = true;
    double xp[2] = {0.0, 0.0}; // the solution of the block before (rows lane, lane + 64): its terms below come out of these registers
    // (round 5) the descriptors of a block are fetched one block ahead; the block before's solution stays in registers (it was stored
    // and read back through memory behind a fence); S(r, k) is requested 16 columns at a time
    int ni_n = D.bs[0], ldd_n = D.ldd[0], ldsg_n = 1;
    const double * Li_n = D.diag[(size_t)pb * D.nblk];
    const double * Sg_n = nullptr;
    for(int i = 0; i < D.nblk; ++i)
    {
      const int ni = ni_n, ldd = ldd_n, ldsg = ldsg_n;
      const double * Li = Li_n;
      const double * Sg = Sg_n;
      if(i + 1 < D.nblk)
      {
        ni_n = D.bs[i + 1];
        ldd_n = D.ldd[i + 1];
        Li_n = D.diag[(size_t)pb * D.nblk + i + 1];
        Sg_n = D.side[(size_t)pb * (D.nblk - 1) + i];
        ldsg_n = D.lds[i];
      }
      if(n + ni >= start)
      {
        if(zero)
        {
          const int r = n + ni - start, off = ni - r;
          double xa = (lane < r) ? Mc[start + lane] : 0.0, xb = (lane + 64 < r) ? Mc[start + lane + 64] : 0.0;
          wave_fwd2(Li + off + (size_t)off * ldd, ldd, r, xa, xb, lane);
          if(lane < r) Mc[start + lane] = xa;
          if(lane + 64 < r) Mc[start + lane + 64] = xb;
          zero = false;
          // (the solve above ran on the rows from `off` on, lane e = row off + e: the next block wants row k in lane k mod 64 of xp[k / 64])
          __threadfence_block();
          wave_sync();
          xp[0] = (lane < ni) ? Mc[n + lane] : 0.0; // (rows in front of `start`: whatever zero the caller stored)
          xp[1] = (lane + 64 < ni) ? Mc[n + lane + 64] : 0.0;
        }
        else
        {
          double acc[2]; // rows lane, lane + 64 of block i
#pragma unroll
          for(int h = 0; h < 2; ++h)
          {
            const int r = lane + 64 * h;
            const int rc = min(r, ni - 1);
            acc[h] = (r < ni) ? Mc[n + r] : 0.0;
            if(64 * h < ni) // (uniform)
              for(int k0 = 0; k0 < li; k0 += 16)
              {
                double sv[16];
#pragma unroll
                for(int e = 0; e < 16; ++e) sv[e] = AT(Sg, ldsg, rc, min(k0 + e, li - 1));
#pragma unroll
                for(int e = 0; e < 16; ++e)
                {
                  const int k = k0 + e;
                  if(k < li) // (uniform)
                  {
                    const double xk = (k < 64) ? bcast_lane_s(xp[0], k) : bcast_lane_s(xp[1], k - 64);
                    const double t = FMA(-sv[e], xk, acc[h]);
                    acc[h] = (r < ni) ? t : acc[h];
                  }
                }
              }
          }
          wave_fwd2(Li, ldd, ni, acc[0], acc[1], lane);
#pragma unroll
          for(int h = 0; h < 2; ++h)
            if(lane + 64 * h < ni) Mc[n + lane + 64 * h] = acc[h];
          xp[0] = acc[0];
          xp[1] = acc[1];
        }
      }
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
  const int ntot = D.ntot;
  const int end = (end_in < 0) ? ntot : end_in;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mc = Mp + (size_t)c * ldm;
    int n = ntot, li = 0;
    bool zero = true;
    double xp[2] = {0.0, 0.0}; // the solution of the block behind (rows lane, lane + 64); see tri_lsolve_kernel
    const int last = D.nblk - 1;
    int ni_n = D.bs[last], ldd_n = D.ldd[last], ldsg_n = 1;
    const double * Li_n = D.diag[(size_t)pb * D.nblk + last];
    const double * Sg_n = nullptr;
    for(int i = last; i >= 0; --i)
    {
      const int ni = ni_n, ldd = ldd_n, ldsg = ldsg_n;
      const double * Li = Li_n;
      const double * Sg = Sg_n;
      if(i > 0)
      {
        ni_n = D.bs[i - 1];
        ldd_n = D.ldd[i - 1];
        Li_n = D.diag[(size_t)pb * D.nblk + i - 1];
        Sg_n = D.side[(size_t)pb * (D.nblk - 1) + i - 1];
        ldsg_n = D.lds[i - 1];
      }
      if(n - ni < end)
      {
        if(zero)
        {
          const int r = end - n + ni;
          double xa = (lane < r) ? Mc[n - ni + lane] : 0.0, xb = (lane + 64 < r) ? Mc[n - ni + lane + 64] : 0.0;
          wave_bwd2(Li, ldd, r, xa, xb, lane);
          if(lane < r) Mc[n - ni + lane] = xa;
          if(lane + 64 < r) Mc[n - ni + lane + 64] = xb;
          zero = false;
          __threadfence_block();
          wave_sync();
          xp[0] = (lane < ni) ? Mc[n - ni + lane] : 0.0; // (rows from `end` on: whatever zero the caller stored)
          xp[1] = (lane + 64 < ni) ? Mc[n - ni + lane + 64] : 0.0;
        }
        else
        {
          double acc[2]; // rows lane, lane + 64 of block i
#pragma unroll
          for(int h = 0; h < 2; ++h)
          {
            const int r = lane + 64 * h;
            const int rc = min(r, ni - 1);
            acc[h] = (r < ni) ? Mc[n - ni + r] : 0.0;
            if(64 * h < ni) // (uniform)
              for(int k0 = li - 1; k0 >= 0; k0 -= 16)
              {
                double sv[16];
#pragma unroll
                for(int e = 0; e < 16; ++e) sv[e] = AT(Sg, ldsg, max(k0 - e, 0), rc);
#pragma unroll
                for(int e = 0; e < 16; ++e)
                {
                  const int k = k0 - e;
                  if(k >= 0) // (uniform)
                  {
                    const double xk = (k < 64) ? bcast_lane_s(xp[0], k) : bcast_lane_s(xp[1], k - 64);
                    const double t = FMA(-sv[e], xk, acc[h]);
                    acc[h] = (r < ni) ? t : acc[h];
                  }
                }
              }
          }
          wave_bwd2(Li, ldd, ni, acc[0], acc[1], lane);
#pragma unroll
          for(int h = 0; h < 2; ++h)
            if(lane + 64 * h < ni) Mc[n - ni + lane + 64 * h] = acc[h];
          xp[0] = acc[0];
          xp[1] = acc[1];
        }
      }
      li = ni;
      n -= ni;
    }
  }
}

inline size_t lds_arrow_block(int maxb)
{
  return sizeof(double) * 2 * (size_t)maxb * (maxb | 1);
}
// LDS of a kernel that keeps one maxb x (maxb | 1) block, its diagonal and a PANEL of a side block: the panel is a whole block up
// to 64 rows (everything at once, as before round 4) and whatever the CU's 160 KB leave beyond -- returns the panel's doubles
inline int panel_doubles(int maxb, size_t & lds_bytes)
{
  const size_t blk = (size_t)maxb * (maxb | 1), fixed = blk + (size_t)((maxb + 1) & ~1) + 8;
  const size_t room = (160 * 1024) / sizeof(double) - fixed;
  const size_t panel = blk + maxb < room ? blk + maxb : room;
  lds_bytes = sizeof(double) * (fixed + panel);
  return (int)panel;
}
// The head of the arrow's L solve (pointing down) for blocks of at most 32 rows, round 5: y_b -= sum_i B_i x_i is ONE chain of
// (number of unknowns) terms per head row -- 3000 at c5 --, and the staged kernel above spent 190 us on it for one problem: a scalar
// chain out of LDS (33 cycles a term) behind groups of side blocks staged by three waves with plain loads (a round trip per group).
// Here, as in arrow_head_mfma_kernel: wave 3 streams the side blocks into an LDS ring with LDS-direct loads (three slots of two
// blocks in flight) and wave 0 runs the chain, lane = head row, every operand of a block requested before the block's terms run.
// (The matrix cores were tried -- A = [-x_k] in row 0, B = B_i^T, so that D(0, j) is head row j's chain, bit for bit -- and lost:
// a dependent v_mfma_f64_16x16x4 costs ~170 cycles and one row of sixteen is used; 100 us for the 1400 of c5 against 41 us for the
// 3000 scalar terms.)  Per element: the staged kernel's chain.
constexpr int LB = 2, LSLOTS = 4, LNI = 8, LIMG = LNI * 128, LSLOT = LB * LIMG;
__global__ __launch_bounds__(256) void arrow_lsolve_head_ring_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start, int end)
{
  extern __shared__ double sm[]; // ring | x (the unknowns of the off-head blocks) | descriptors
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6, pb = blockIdx.x;
  const int ib = D.nblk - 1, nbk = D.bs[ib], nside = D.nblk - 1, nchunk = (nside + LB - 1) / LB;
  const int ntot = D.ntot;
  const int nx = ntot - nbk;
  double * Mp = M + (size_t)pb * D.mstride;
  const double * Lb = D.diag[(size_t)pb * D.nblk + ib];
  double * ring = sm;
  double * xs = ring + LSLOTS * LSLOT;
  const double ** m_ptr = reinterpret_cast<const double **>(xs + ((nx + 1) & ~1));
  int * m_ni = reinterpret_cast<int *>(m_ptr + nside);
  int * m_ld = m_ni + nside;
  int * m_n0 = m_ld + nside;
  for(int i = tid; i < nside; i += 256)
  {
    m_ptr[i] = D.side[(size_t)pb * nside + i];
    m_ni[i] = D.bs[i];
    m_ld[i] = D.lds[i];
  }
  __syncthreads();
  if(tid == 0) // (out of LDS: the same walk over D.bs is a dependent memory round trip per block)
  {
    int n0 = 0;
    for(int i = 0; i < nside; ++i)
    {
      m_n0[i] = n0;
      n0 += m_ni[i];
    }
  }
  const int ppc = nbk >> 1, ldi = nbk; // 16-byte pieces per column of a side block (nbk rows); leading dimension of its image
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mb = Mp + nx + (size_t)c * ldm;
    __syncthreads(); // (descriptors; the previous right-hand side is done with xs and the ring)
    for(int e = tid; e < nx; e += 256) xs[e] = Mp[e + (size_t)c * ldm];
    __syncthreads();
    if(wave == 3)
    {
      int pcol[LNI], poff[LNI];
#pragma unroll
      for(int q = 0; q < LNI; ++q)
      {
        const int e = l + 64 * q;
        pcol[q] = e / ppc;
        poff[q] = 2 * (e - pcol[q] * ppc);
      }
      auto issue = [&](int k) {
        double * slot = ring + (k % LSLOTS) * LSLOT;
#pragma unroll
        for(int g = 0; g < LB; ++g)
        {
          const int i = min(k * LB + g, nside - 1);
          const int ni = m_ni[i];
          const double * Sg = m_ptr[i];
          const int ldsg = m_ld[i];
#pragma unroll
          for(int q = 0; q < LNI; ++q)
          {
            const double * p = Sg + (size_t)min(pcol[q], ni - 1) * ldsg + poff[q];
            __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)(slot + g * LIMG + q * 128), 16, 0, 0);
          }
        }
      };
#ifdef LRING_PROF
      long long tw = 0, tb = 0, ti = 0;
#endif
      for(int k = 0; k < 3 && k < nchunk; ++k) issue(k);
      for(int k = 0; k < nchunk; ++k)
      {
#ifdef LRING_PROF
        const long long t0 = wall_clock64();
#endif
        if(k + 2 < nchunk)
          asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if(k + 1 < nchunk)
          asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LRING_PROF
        const long long t1 = wall_clock64();
#endif
        lds_barrier(); // B_k
#ifdef LRING_PROF
        const long long t2 = wall_clock64();
#endif
        if(k + 3 < nchunk) issue(k + 3);
#ifdef LRING_PROF
        tw += t1 - t0; tb += t2 - t1; ti += wall_clock64() - t2;
#endif
      }
#ifdef LRING_PROF
      if(pb == 0 && l == 0) printf("lring loader: wait loads %lld, barrier %lld, issue %lld (10 ns ticks)\n", tw, tb, ti);
#endif
    }
    else if(wave == 0)
    {
      // lane = head row: its chain over the blocks in order, k ascending -- every operand of a block requested before its chain runs
      // (clamped addresses, fixed counts; the terms outside [s, ni) are not executed)
      const int rl = min(l, nbk - 1);
      double acc = (l < nbk) ? Mb[l] : 0.0;
#ifdef LRING_PROF
      long long cb = 0, cc = 0, clast = wall_clock64();
#endif
      for(int k = 0; k < nchunk; ++k)
      {
#ifdef LRING_PROF
        { const long long t = wall_clock64(); cc += t - clast; clast = t; }
#endif
        lds_barrier(); // B_k: slot k is in LDS
#ifdef LRING_PROF
        { const long long t = wall_clock64(); cb += t - clast; clast = t; }
#endif
        const double * slot = ring + (k % LSLOTS) * LSLOT;
#pragma unroll
        for(int g = 0; g < LB; ++g)
        {
          const int i = k * LB + g;
          if(i >= nside) continue; // (uniform)
          const int ni = m_ni[i], n0 = m_n0[i];
          const int s = max(start - n0, 0);
          if(ni < s || end <= n0) continue; // (the reference's zero-window skipping, blockArrowLLT.cpp:139-151)
          const double * img = slot + g * LIMG + rl;
          const double * xi = xs + n0;
          double bv[32], xv[32];
#pragma unroll
          for(int kq = 0; kq < 32; ++kq)
          {
            const int kc = min(kq, ni - 1);
            bv[kq] = img[kc * ldi];
            xv[kq] = xi[kc];
          }
#pragma unroll
          for(int kq = 0; kq < 32; ++kq)
            if(kq >= s && kq < ni) acc = FMA(-bv[kq], xv[kq], acc); // (uniform)
        }
      }
#ifdef LRING_PROF
      if(pb == 0 && l == 0) printf("lring chain: barrier %lld, compute %lld (10 ns ticks)\n", cb, cc);
#endif
      double a1 = 0.0;
      wave_fwd2(Lb, D.ldd[ib], nbk, acc, a1, l);
      if(l < nbk) Mb[l] = acc;
    }
    else
    {
      for(int k = 0; k < nchunk; ++k) lds_barrier();
    }
  }
}

// The block-tridiagonal solves for blocks of at most 32 rows (round 5): everything block i needs -- its row (column) of L_i, the diagonal
// entry, its row (column) of the side block and its right-hand side entry -- is requested while block i - 1 (i + 1) is being solved, and
// the substitution runs on registers (wave_fwd_pre / wave_bwd_pre); the descriptors of all blocks sit in LDS.  The kernels above pay
// some five dependent memory round trips per block (4.4 us); per element the chains are theirs.
struct TriMeta
{
  const double ** mL;
  const double ** mS;
  int *ni, *ldd, *lds, *off;
};
__device__ __forceinline__ TriMeta tri_meta(double * sm, const BlockDesc & D, int pb, int tid, int nthr)
{
  TriMeta m;
  m.mL = reinterpret_cast<const double **>(sm);
  m.mS = m.mL + D.nblk;
  m.ni = reinterpret_cast<int *>(m.mS + D.nblk);
  m.ldd = m.ni + D.nblk;
  m.lds = m.ldd + D.nblk;
  m.off = m.lds + D.nblk;
  for(int i = tid; i < D.nblk; i += nthr)
  {
    m.mL[i] = D.diag[(size_t)pb * D.nblk + i];
    m.mS[i] = (i + 1 < D.nblk) ? D.side[(size_t)pb * (D.nblk - 1) + i] : nullptr;
    m.ni[i] = D.bs[i];
    m.ldd[i] = D.ldd[i];
    m.lds[i] = (i + 1 < D.nblk) ? D.lds[i] : 1;
    m.off[i] = D.off[i];
  }
  return m;
}

__global__ __launch_bounds__(64) void tri_lsolve_pre_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int start)
{
  extern __shared__ double sm[];
  const int lane = threadIdx.x, pb = blockIdx.x;
  const TriMeta m = tri_meta(sm, D, pb, lane, 64);
  wave_sync();
  __threadfence_block();
  double * Mp = M + (size_t)pb * D.mstride;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mc = Mp + (size_t)c * ldm;
    // the first block the window reaches: solved on its rows from `start` on (the kernel above's code)
    int i0 = 0;
    while(i0 < D.nblk && m.off[i0] + m.ni[i0] < start) ++i0;
    if(i0 >= D.nblk) continue;
    double xp;
    {
      const int n = m.off[i0], ni = m.ni[i0], ldd = m.ldd[i0];
      const double * Li = m.mL[i0];
      const int r = n + ni - start, off = ni - r;
      double xa = (lane < r) ? Mc[start + lane] : 0.0, xb = 0.0;
      wave_fwd2(Li + off + (size_t)off * ldd, ldd, r, xa, xb, lane);
      if(lane < r) Mc[start + lane] = xa;
      __threadfence_block();
      wave_sync();
      xp = (lane < ni) ? Mc[n + lane] : 0.0; // (rows in front of `start`: whatever zero the caller stored)
    }
    // operands of block j: lane = row; lv[e] = L_j(row, min(e, n_j - 1)), sv[e] = S_{j-1}(row, min(e, n_{j-1} - 1))
    double lv[32], sv[32], dg = 1.0, rhs = 0.0;
    auto fetch = [&](int j, double (&lo)[32], double (&so)[32], double & dgo, double & rho) {
      const int nj = m.ni[j], lj = m.ni[j - 1], rl = min(lane, nj - 1);
      const double * Lj = m.mL[j];
      const double * Sj = m.mS[j - 1];
      const int ldd = m.ldd[j], ldsg = m.lds[j - 1];
#pragma unroll
      for(int e = 0; e < 32; ++e) lo[e] = AT(Lj, ldd, rl, min(e, nj - 1));
#pragma unroll
      for(int e = 0; e < 32; ++e) so[e] = AT(Sj, ldsg, rl, min(e, lj - 1));
      dgo = AT(Lj, ldd, rl, rl);
      rho = (lane < nj) ? Mc[m.off[j] + lane] : 0.0;
    };
    if(i0 + 1 < D.nblk) fetch(i0 + 1, lv, sv, dg, rhs);
    for(int i = i0 + 1; i < D.nblk; ++i)
    {
      double lvn[32], svn[32], dgn = 1.0, rhsn = 0.0;
      if(i + 1 < D.nblk) fetch(i + 1, lvn, svn, dgn, rhsn);
      const int ni = m.ni[i], li = m.ni[i - 1];
      double acc = rhs;
#pragma unroll
      for(int k = 0; k < 32; ++k)
        if(k < li) // (uniform)
        {
          const double t = FMA(-sv[k], bcast_lane_s(xp, k), acc);
          acc = (lane < ni) ? t : acc;
        }
      acc = wave_fwd_pre<32>(lv, dg, m.mL[i], m.ldd[i], ni, acc, lane);
      if(lane < ni) Mc[m.off[i] + lane] = acc;
      xp = acc;
#pragma unroll
      for(int e = 0; e < 32; ++e)
      {
        lv[e] = lvn[e];
        sv[e] = svn[e];
      }
      dg = dgn;
      rhs = rhsn;
    }
  }
}

__global__ __launch_bounds__(64) void tri_ltsolve_pre_kernel(const BlockDesc D, double * M, int ldm, int nrhs, int end_in)
{
  extern __shared__ double sm[];
  const int lane = threadIdx.x, pb = blockIdx.x;
  const TriMeta m = tri_meta(sm, D, pb, lane, 64);
  wave_sync();
  __threadfence_block();
  double * Mp = M + (size_t)pb * D.mstride;
  const int ntot = D.ntot;
  const int end = (end_in < 0) ? ntot : end_in;
  for(int c = 0; c < nrhs; ++c)
  {
    double * Mc = Mp + (size_t)c * ldm;
    // the last block the window reaches: solved on its rows in front of `end`
    int i0 = D.nblk - 1;
    while(i0 >= 0 && m.off[i0] >= end) --i0;
    if(i0 < 0) continue;
    double xp;
    {
      const int n0 = m.off[i0], ni = m.ni[i0], ldd = m.ldd[i0];
      const int r = end - n0 < ni ? end - n0 : ni;
      double xa = (lane < r) ? Mc[n0 + lane] : 0.0, xb = 0.0;
      wave_bwd2(m.mL[i0], ldd, r, xa, xb, lane);
      if(lane < r) Mc[n0 + lane] = xa;
      __threadfence_block();
      wave_sync();
      xp = (lane < ni) ? Mc[n0 + lane] : 0.0; // (rows from `end` on: whatever zero the caller stored)
    }
    // operands of block j: lane = row r of the block; lv[e] = L_j(max(n_j - 1 - e, 0), r), sv[e] = S_j(max(n_{j+1} - 1 - e, 0), r)
    double lv[32], sv[32], dg = 1.0, rhs = 0.0;
    auto fetch = [&](int j, double (&lo)[32], double (&so)[32], double & dgo, double & rho) {
      const int nj = m.ni[j], lj = m.ni[j + 1], rl = min(lane, nj - 1);
      const double * Lj = m.mL[j];
      const double * Sj = m.mS[j];
      const int ldd = m.ldd[j], ldsg = m.lds[j];
#pragma unroll
      for(int e = 0; e < 32; ++e) lo[e] = AT(Lj, ldd, max(nj - 1 - e, 0), rl);
#pragma unroll
      for(int e = 0; e < 32; ++e) so[e] = AT(Sj, ldsg, max(lj - 1 - e, 0), rl);
      dgo = AT(Lj, ldd, rl, rl);
      rho = (lane < nj) ? Mc[m.off[j] + lane] : 0.0;
    };
    if(i0 > 0) fetch(i0 - 1, lv, sv, dg, rhs);
    for(int i = i0 - 1; i >= 0; --i)
    {
      double lvn[32], svn[32], dgn = 1.0, rhsn = 0.0;
      if(i > 0) fetch(i - 1, lvn, svn, dgn, rhsn);
      const int ni = m.ni[i], li = m.ni[i + 1];
      double acc = rhs;
#pragma unroll
      for(int e = 0; e < 32; ++e)
        if(e < li) // (uniform): the term of k = li - 1 - e
        {
          const double t = FMA(-sv[e], bcast_lane_s(xp, max(li - 1 - e, 0)), acc);
          acc = (lane < ni) ? t : acc;
        }
      acc = wave_bwd_pre<32>(lv, dg, m.mL[i], m.ldd[i], ni, acc, lane);
      if(lane < ni) Mc[m.off[i] + lane] = acc;
      xp = acc;
#pragma unroll
      for(int e = 0; e < 32; ++e)
      {
        lv[e] = lvn[e];
        sv[e] = svn[e];
      }
      dg = dgn;
      rhs = rhsn;
    }
  }
}

} // namespace

hipError_t launch_block_arrow_llt(const BlockDesc & d, int nprob, int maxb, hipStream_t st)
{
  hipError_t e = hipMemsetAsync(d.fail, 0, sizeof(int) * nprob, st);
  if(e != hipSuccess) return e;
  BlockDesc dd = d;
  size_t lds_blk = 0;
  dd.xs_cap = panel_doubles(maxb, lds_blk);
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&arrow_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_blk);
  if(e != hipSuccess) return e;
  static const bool slow_blocks = std::getenv("JRLQP_ARROW_BLOCKS_BARRIER") != nullptr; // (diagnostic: the barrier-per-column kernel)
  if(d.nblk > 1 && !d.up && maxb <= 32 && !slow_blocks)
  {
    // (the register kernel: two padded 32 x 33 blocks in LDS, or what its fall-back -- the barrier kernel's body -- needs)
    const size_t lds_fast = std::max(lds_blk, sizeof(double) * 2 * 32 * 33);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&arrow_block_fast_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fast);
    if(e != hipSuccess) return e;
    hipLaunchKernelGGL(arrow_block_fast_kernel<32>, dim3(d.nblk - 1, nprob), dim3(kTriThreads), lds_fast, st, dd);
  }
  else if(d.nblk > 1)
    hipLaunchKernelGGL(arrow_block_kernel, dim3(d.nblk - 1, nprob), dim3(kTriThreads), lds_blk, st, dd);
  if(maxb <= 64)
  {
    const bool mfma = std::getenv("JRLQP_ARROW_HEAD_LDS") == nullptr; // (diagnostic: the round 1-4 kernel that staged every block in LDS)
    // (the matrix-core kernel: Ls and, for heads of at most 32 rows, the ring of side blocks -- 4 slots x 2 blocks x 8 KiB)
    const size_t lds_head = sizeof(double) * ((size_t)maxb * (maxb | 1) + (mfma ? ((maxb <= 32 && d.nblk - 1 <= kHeadRingMaxBlocks) ? 1 + 4 * 2 * 8 * 128 + 2 * (size_t)d.nblk : 0) : 64 * 65));
    if(mfma)
    {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&arrow_head_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_head);
      if(e != hipSuccess) return e;
      hipLaunchKernelGGL(arrow_head_mfma_kernel, dim3(nprob), dim3(256), lds_head, st, dd);
    }
    else
    {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&arrow_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_head);
      if(e != hipSuccess) return e;
      hipLaunchKernelGGL(arrow_head_kernel, dim3(nprob), dim3(256), lds_head, st, dd);
    }
  }
  else
  {
    const size_t lds_head = sizeof(double) * ((size_t)maxb * (maxb | 1) + maxb + 2);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&arrow_head_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_head);
    if(e != hipSuccess) return e;
    hipLaunchKernelGGL(arrow_head_big_kernel, dim3(nprob), dim3(256), lds_head, st, dd);
  }
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
    int maxb = 0;
    for(int k = 0; k < d.nblk; ++k) maxb = d.h_bs[k] > maxb ? d.h_bs[k] : maxb;
    static const bool no_ring = std::getenv("JRLQP_ARROW_LSOLVE_STAGED") != nullptr; // (diagnostic: the staged scalar chain)
    if(d.ring_ok && !d.up && maxb <= 32 && d.nblk > 1 && !no_ring)
    {
      // (the ring of side blocks | the unknowns of the off-head blocks | 24 bytes of descriptors per block)
      const size_t lds = sizeof(double) * ((size_t)LSLOTS * LSLOT + (size_t)((ntot + 1) & ~1) + 3 * (size_t)d.nblk + 2);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&arrow_lsolve_head_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if(e != hipSuccess) return e;
      hipLaunchKernelGGL(arrow_lsolve_head_ring_kernel, dim3(nprob), dim3(256), lds, st, d, M, ldm, nrhs, start, end);
    }
    else if(maxb <= 64 && d.nblk > 1)
    {
      // side blocks staged through LDS in groups (arrow_lsolve_head_staged_kernel): as many blocks per group as 2 x gb x bcap doubles allow
      const int bcap = maxb * (maxb | 1);
      int gb = (int)((96 * 1024 / sizeof(double)) / (2 * (size_t)(bcap + 64)));
      gb = gb < 1 ? 1 : (gb > 16 ? 16 : gb);
      const size_t lds = sizeof(double) * 2 * (size_t)gb * (bcap + 64);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&arrow_lsolve_head_staged_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if(e != hipSuccess) return e;
      hipLaunchKernelGGL(arrow_lsolve_head_staged_kernel, dim3(nprob), dim3(256), lds, st, d, M, ldm, nrhs, start, end, gb, bcap);
    }
    else
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
    const size_t l2 = sizeof(double) * 4 * (size_t)nbk * (nbk | 1);
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
  if(maxb > 64) // (three blocks of more than 64 rows do not fit in LDS)
  {
    BlockDesc dd = d;
    size_t lbig = 0;
    dd.xs_cap = panel_doubles(maxb, lbig);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tri_llt_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lbig);
    if(e != hipSuccess) return e;
    hipLaunchKernelGGL(tri_llt_big_kernel, dim3(nprob), dim3(kTriThreads), lbig, st, dd);
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
  int maxb = 0;
  for(int k = 0; k < d.nblk; ++k) maxb = d.h_bs[k] > maxb ? d.h_bs[k] : maxb;
  static const bool plain = std::getenv("JRLQP_TRI_SOLVE_PLAIN") != nullptr; // (diagnostic: the kernels without the one-block-ahead requests)
  if(maxb <= 32 && d.nblk <= 2048 && !plain)
  {
    const size_t lds = (size_t)d.nblk * (2 * sizeof(double *) + 4 * sizeof(int));
    if(!transpose)
      hipLaunchKernelGGL(tri_lsolve_pre_kernel, dim3(nprob), dim3(64), lds, st, d, M, ldm, nrhs, start_or_end);
    else
      hipLaunchKernelGGL(tri_ltsolve_pre_kernel, dim3(nprob), dim3(64), lds, st, d, M, ldm, nrhs, start_or_end);
    return hipGetLastError();
  }
  if(!transpose)
    hipLaunchKernelGGL(tri_lsolve_kernel, dim3(nprob), dim3(64), 0, st, d, M, ldm, nrhs, start_or_end);
  else
    hipLaunchKernelGGL(tri_ltsolve_kernel, dim3(nprob), dim3(64), 0, st, d, M, ldm, nrhs, start_or_end);
  return hipGetLastError();
}
} // namespace jrlqp
