// gi_dense.hip -- Mode A: kernels behind the eight DualSolver hooks for ONE large dense QP whose state
// (J, R, C, x, ...) lives in HBM (gfx950).  The host keeps the active-set state machine (gi_dense_host.cpp).
//
//   hook (include/jrl-qp/DualSolver.h:101-121)      reference implementation            kernels here
//   init_                    GoldfarbIdnaniSolver.cpp:56-82    llt_panel / llt_update (MFMA) / jinv_block / x0
//   selectViolatedConstraint_                     :84-134      select_partial + select_final
//   computeStep_ + computeStepLength_ + dot_      :136-219     jtn (or row gather), z_rows, step_tail
//   (primal/dual update)   DualSolver.cpp:133-150              advance
//   addConstraint_                                :221-237     add_chain + rot_apply
//   removeConstraint_ (+ u shift, DualSolver.cpp:237-244) :239-256   remove_r + rot_apply
//
// CANON rounding contract: every output element owns one in-order fma chain, so the parallelism is
// across outputs (constraints, columns of J, rows of J, elements of the trailing matrix).  The only matrix-core
// use is the blocked-LLT trailing update: v_mfma_f64_16x16x4_f64 accumulates k-ordered fma chains
// (tests/test_gpu_arith.py), so feeding it the panel in increasing k keeps every element's chain exactly the
// oracle's.  Matrices are column-major; J and R have leading dimension n (Workspace.h, GoldfarbIdnaniSolver.cpp:140-141).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>

#include "device_common.h"
#include "gi_dense.h"
#include "gi_batch.h"

namespace jrlqp
{
namespace
{
#undef AT
#define AT(P, ld, i, j) (P)[(size_t)(i) + (size_t)(j) * (size_t)(ld)]
constexpr int NONE = 0x7fffffff;
enum : int { INACTIVE = 0, LOWER = 1, UPPER = 2, EQUALITY = 3, LOWER_BOUND = 4, UPPER_BOUND = 5, FIXED = 6 };

// lexicographic (value, index) minimum over the wave (general index order)
__device__ __forceinline__ void wave_argmin(double & val, int & idx, int & aux)
{
#pragma unroll
  for(int off = 32; off >= 1; off >>= 1)
  {
    double ov = shfl_xor_d(val, off);
    int oi = __shfl_xor(idx, off);
    int oa = __shfl_xor(aux, off);
    bool take = (oi != NONE) && (idx == NONE || ov < val || (ov == val && oi < idx));
    if(take)
    {
      val = ov;
      idx = oi;
      aux = oa;
    }
  }
}

#ifdef JRLQP_ROT_PROF
// diagnostic build only (EXTRA=-DJRLQP_ROT_PROF): where workgroup 0 of the sweep spends its time, in 10 ns ticks summed over launches
__device__ unsigned long long g_rotprof[32];
#define ROTP(slot, expr) do { if(blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_rotprof[slot], (unsigned long long)(expr)); } while(0)
#define ROTT() wall_clock64()
#define ROTQ(slot, expr) do { if(threadIdx.x == 0) atomicAdd(&g_rotprof[slot], (unsigned long long)(expr)); } while(0)
#else
#define ROTP(slot, expr) do { (void)sizeof(expr); } while(0)
#define ROTT() 0ull
#define ROTQ(slot, expr) do { (void)sizeof(expr); } while(0)
#endif
// ---- pass hand-off tags (PassSync, gi_dense.h) -----------------------------------------------------------------------------
// Whole workgroup.  The wait gives up after ~1 s (a producer that never ran: a dead stream), says so in the host-visible mailbox
// and lets the kernel run on -- the host voids the solve; nothing spins for ever.
__device__ __forceinline__ void flag_wait(const int * flag, int tag, Mailbox * mb)
{
  if(threadIdx.x == 0)
  {
    int spins = 0;
    // (relaxed polls: an acquire per poll would invalidate this XCD's cached lines under the kernels running beside it)
    while(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag)
    {
      __builtin_amdgcn_s_sleep(4);
      if(++spins > (1 << 20))
      {
        if(mb) mb->sync_err = 1;
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // every wave drops what its caches may hold of the producer's buffers
}
// after the workgroup's last global store; total = number of workgroups that arrive (the counter is left at zero again)
__device__ __forceinline__ void flag_post(int * cnt, int * flag, int total, int tag)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if(threadIdx.x == 0)
  {
    bool last = true;
    if(total > 1)
    {
      last = (__hip_atomic_fetch_add(cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == total - 1);
      if(last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if(last) __hip_atomic_store(flag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The scan of the pass BEHIND an addition enqueued ahead of the add / drop decision (StepDev, gi_dense.h): a one-wave gate kernel in
// front of the scan kernels waits for the step kernel's decision tag (the step kernel is in another stream; enqueued earlier, so it is
// never stuck behind the gate); the kernels behind the gate find the decision there and simply return if the pass does not add.
// The gate also waits for the x of the sweep's first phase (x_tag): the scan's workgroups must not sit on their LDS waiting for it --
// 512 of them, two per CU, left no CU with room for a workgroup of the sweep that was to publish it (a 0.4 s stall until the wait
// gave up, then the event-ordered rerun).  Behind the gate nothing waits.
__global__ void decision_gate_kernel(const StepDev * sd, int tag, const PassSync * sync, int wait_x, Mailbox * mb)
{
  int spins = 0;
  while(__hip_atomic_load(&sd->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag)
  {
    __builtin_amdgcn_s_sleep(4);
    if(++spins > (1 << 20))
    {
      if(mb) mb->sync_err = 1;
      return;
    }
  }
  if(__hip_atomic_load(&sd->kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kKindAdd || !wait_x) return;
  spins = 0;
  while(__hip_atomic_load(&sync->x_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != wait_x)
  {
    __builtin_amdgcn_s_sleep(4);
    if(++spins > (1 << 20))
    {
      if(mb) mb->sync_err = 1;
      return;
    }
  }
}
__device__ __forceinline__ bool gate_is_add(const StepDev * sd, int tag)
{
  return __hip_atomic_load(&sd->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag &&
         __hip_atomic_load(&sd->kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kKindAdd;
}

// ============================================================================================================
// K1  blocked right-looking Cholesky
// ============================================================================================================
constexpr int NB = 32; // panel width

// Panel j0, steps 1 and 2 in ONE launch (round 2): every workgroup (one wave) factorises the diagonal block A11 itself, out of
// registers (lane = row; llt_reg_columns, device_common.h: look-ahead pivot, the finished columns in LDS), and then computes its
// 64 rows of L21 = A21 L11^-T out of registers as well (trsm_reg_columns: L11 read back from LDS as broadcasts).  The redundant
// factorisation costs 6 us per workgroup and saves a dependent launch per panel.  L11 goes back over A11 -- which every
// workgroup of the launch has to READ first: it is written by the workgroup that was the LAST to have A11 in its registers (an
// arrival counter behind the loads, info[1]; every workgroup holds the same bits, so any of them can be the writer).  Rounds 2-3
// let workgroup 0 write it as soon as it had finished, which is only safe while all workgroups of the launch start together: with
// another tenant's kernels on the CUs a late workgroup read L11 for A11 and reported a non-positive pivot
// (tests/test_gpu_dense.py::test_dense_with_a_third_party_on_the_cus).  Blocks narrower than
// NB (the last panel) are padded with the identity, which factorises to itself and contributes exact zeros.  Per element: the
// terms in ascending k, then one division -- the bits of the two left-looking LDS kernels of round 1 (26 + 23 us per panel;
// this: 9 us).  A non-positive pivot: *info = its index, nothing at or after that column is written and
// A21 is left alone, as before.
__global__ __launch_bounds__(64) void llt_panel_fused_kernel(double * G, int ldg, int n, int j0, int * info)
{
  __shared__ double L11[NB * (NB + 1)];
  const int lane = threadIdx.x;
  const int nb = min(NB, n - j0);
  constexpr int ld1 = NB + 1;
  if(*info >= 0) return;
  double a[NB];
  {
    const bool row = lane < nb;
    const int r = j0 + (row ? lane : 0);
#pragma unroll
    for(int c = 0; c < NB; ++c)
    {
      const double v = AT(G, ldg, r, j0 + ((c < nb) ? c : 0));
      a[c] = (row && c < nb) ? v : ((lane == c) ? 1.0 : 0.0);
    }
  }
  // A11 is in registers: arrive.  (release: the loads above are complete before the increment is visible; acquire: the writer's
  // stores below come after every other workgroup's loads)
  bool writer = true;
  if(gridDim.x > 1)
  {
    int arrived = 0;
    if(lane == 0)
    {
      arrived = __hip_atomic_fetch_add(info + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if(arrived == (int)gridDim.x - 1) __hip_atomic_store(info + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (the next panel's launch finds zero)
    }
    writer = (__builtin_amdgcn_readfirstlane(arrived) == (int)gridDim.x - 1);
  }
  int bad = -1;
  {
    const double p0 = bcast_lane(a[0], 0), l0 = __builtin_sqrt(opaque_v(p0));
    llt_reg_columns<0, NB>(a, bad, lane, L11, ld1, p0, l0, a[0] / l0);
  }
  // (llt_reg_columns stores lanes >= J of column J for lane < NB: the padded rows too, harmless)
  const int cols = (bad < 0) ? nb : bad; // finished columns
  if(writer)
  {
    if(lane < nb)
    {
#pragma unroll
      for(int c = 0; c < NB; ++c)
        if(c < cols && lane >= c) AT(G, ldg, j0 + lane, j0 + c) = a[c];
    }
    if(bad >= 0 && lane == 0) *info = j0 + bad; // Eigen llt_inplace: index of the first non-positive pivot
  }
  if(bad >= 0) return;
  wave_sync();
  const int row = j0 + nb + blockIdx.x * 64 + lane;
  if(j0 + nb >= n) return; // (the last panel has no rows below it)
  const bool live = row < n;
  const int rr = live ? row : (n - 1);
  double x[NB];
#pragma unroll
  for(int c = 0; c < NB; ++c)
  {
    const double v = AT(G, ldg, rr, j0 + ((c < nb) ? c : 0));
    x[c] = (c < nb) ? v : 0.0;
  }
  trsm_reg_columns<0, NB>(x, L11, ld1);
  if(live)
  {
#pragma unroll
    for(int c = 0; c < NB; ++c)
      if(c < nb) AT(G, ldg, row, j0 + c) = x[c];
  }
}

typedef double double4_t __attribute__((ext_vector_type(4)));

// Trailing update A22 -= L21 L21^T (lower triangle) with v_mfma_f64_16x16x4_f64: one wave per 16 x 16 tile,
// k in increasing order (4 per instruction): per element acc = fma(-L_ik, L_jk, acc), k ascending -- CANON C1.
__global__ __launch_bounds__(64) void llt_update_kernel(double * G, int ldg, int n, int j0, const int * info)
{
  if(*info >= 0) return;
  const int nb = min(NB, n - j0);
  const int base = j0 + nb;
  const int ti = blockIdx.x, tj = blockIdx.y;
  if(tj > ti) return;
  const int l = threadIdx.x;
  const int i0 = base + ti * 16, c0 = base + tj * 16;
  if(i0 >= n || c0 >= n) return;
  const int col = c0 + (l & 15);
  double4_t acc;
#pragma unroll
  for(int r = 0; r < 4; ++r)
  {
    const int row = i0 + (l >> 4) + 4 * r;
    acc[r] = (row < n && col < n) ? AT(G, ldg, row, col) : 0.0;
  }
  const int arow = i0 + (l & 15); // A operand: A[i = l&15][k = l>>4]
  const int brow = c0 + (l & 15); // B operand: B[k = l>>4][j = l&15] = L21[c0 + j, k]
  for(int k = 0; k < nb; k += 4)
  {
    const int kk = k + (l >> 4);
    const bool kin = kk < nb;
    double a = (kin && arow < n) ? -AT(G, ldg, arow, j0 + kk) : 0.0;
    double b = (kin && brow < n) ? AT(G, ldg, brow, j0 + kk) : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for(int r = 0; r < 4; ++r)
  {
    const int row = i0 + (l >> 4) + 4 * r;
    if(row < n && col < n && row >= col) AT(G, ldg, row, col) = acc[r]; // strictly-upper part stays untouched
  }
}

// ============================================================================================================
// K2  J = L^-T (upper triangular), block row I = [i0, i0+NB) from the bottom up.  Lane = column c of J.
//     acc[r] for the NB rows of the block are NB independent chains per lane:
//       acc_i = delta_ic - sum_{k = c .. i0+NB, descending} L_ki J_kc      (rows below the block)
//     then the in-block back substitution.  J must be zero below its diagonal on entry (memset).
// ============================================================================================================
__global__ __launch_bounds__(256) void jinv_block_kernel(const double * __restrict__ L, int ldl, double * __restrict__ J, int ldj, int n, int i0)
{
  // 4 waves share one tile of 64 columns: wave w accumulates rows 8w .. 8w+7 of the block (8 independent chains per
  // lane instead of 32), the in-block substitution is done by wave 0 after the partial chains met in LDS
  __shared__ double Ls[64 * (NB + 1)]; // staged rows of L:  Ls[k - klo][r] = L[k, i0 + r]
  __shared__ double Jt[64 * 65];       // staged tile of J:  Jt[k - klo][c - c0] (coalesced loads, conflict-free reads)
  __shared__ double Ld[NB * (NB + 1)]; // the diagonal block of L
  __shared__ double As[NB * 65];       // accumulators handed to wave 0
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbk = min(NB, n - i0);
  const int c0 = i0 + blockIdx.x * 64;
  const int c = c0 + lane; // this lane's column
  const int cmax = min(n - 1, c0 + 63);
  const int ncol = cmax - c0 + 1;
  const bool live = c < n;
  // The accumulation over the rows below the block is a matrix product, acc(r, c) -= L(k, i0 + r) J(k, c) for k descending -- and the
  // f64 MFMA is a k-ordered fma chain (tests/test_gpu_arith.py), so it runs on the matrix cores with the bits of the scalar chains:
  // wave w owns the 16 columns c0 + 16 w .. of the tile and both 16-row halves of the block; lane l holds rows (l >> 4) + 4 r of a
  // half (r < 4) in column l & 15, feeds A(i = l & 15, k' = l >> 4) = -L and B(k' = l >> 4, j = l & 15) = J, and k' = 0 is the
  // HIGHEST row of a step of four.  (Rounds 1-3: lane = column with eight scalar chains per lane -- every L entry a broadcast LDS
  // read, which returns 512 bytes for eight: the kernel was bound by the LDS port, 60 us per block row.)
  const int mi = lane & 15, mk = lane >> 4;
  const int cw = c0 + 16 * wave + mi; // the column this lane accumulates
  double4_t acc2[2];
#pragma unroll
  for(int h = 0; h < 2; ++h)
#pragma unroll
    for(int r = 0; r < 4; ++r) acc2[h][r] = (cw < n && cw == i0 + 16 * h + mk + 4 * r) ? 1.0 : 0.0;
  for(int e = tid; e < nbk * nbk; e += 256)
  {
    int i = e % nbk, j = e / nbk;
    Ld[i + j * (NB + 1)] = AT(L, ldl, i0 + i, i0 + j);
  }
  // rows below the block, descending k, staged 64 at a time (J is zero below its diagonal, so rows k > c add nothing).
  // The next stage's entries of L and J are requested into registers BEFORE the chains of the current stage run (fixed
  // trip counts, clamped addresses: no predicated load, no branch between the requests and their use), so a stage costs
  // its 64 x RW fmas instead of a memory round trip plus the fmas.
  constexpr int NLS = (64 * NB) / 256, NJT = (64 * 64) / 256; // staged words per thread
  double pls[NLS], pjt[NJT];
  auto request = [&](int khi) {
    const int cnt = min(64, khi - (i0 + nbk) + 1);
    const int klo = khi - cnt + 1;
#pragma unroll
    for(int u = 0; u < NLS; ++u)
    {
      const int e = tid + 256 * u, kk = e & 63, r = e >> 6;
      pls[u] = AT(L, ldl, klo + min(kk, cnt - 1), i0 + min(r, nbk - 1));
    }
#pragma unroll
    for(int u = 0; u < NJT; ++u)
    {
      const int e = tid + 256 * u, kk = e & 63, cc = e >> 6;
      pjt[u] = AT(J, ldj, klo + min(kk, cnt - 1), c0 + min(cc, ncol - 1));
    }
  };
  int khi = cmax;
  if(khi >= i0 + nbk) request(khi);
  for(; khi >= i0 + nbk; khi -= 64)
  {
    const int cnt = min(64, khi - (i0 + nbk) + 1);
    __syncthreads();
#pragma unroll
    for(int u = 0; u < NLS; ++u)
    {
      const int e = tid + 256 * u, kk = e & 63, r = e >> 6;
      Ls[kk * (NB + 1) + r] = pls[u];
    }
#pragma unroll
    for(int u = 0; u < NJT; ++u)
    {
      const int e = tid + 256 * u, kk = e & 63, cc = e >> 6;
      Jt[kk * 65 + cc] = pjt[u];
    }
    lds_barrier();
    if(khi - 64 >= i0 + nbk) request(khi - 64);
    for(int kt = cnt - 1; kt >= 0; kt -= 4) // rows kt, kt-1, kt-2, kt-3 of the stage: one MFMA step per half (rows past the stage: A = 0)
    {
      const int kk = kt - mk;
      const bool kin = kk >= 0;
      const int kc = kin ? kk : 0;
      const double b = Jt[kc * 65 + 16 * wave + mi];
      const double a0 = kin ? -Ls[kc * (NB + 1) + mi] : 0.0, a1 = kin ? -Ls[kc * (NB + 1) + 16 + mi] : 0.0;
      acc2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc2[0], 0, 0, 0);
      acc2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc2[1], 0, 0, 0);
    }
  }
#pragma unroll
  for(int h = 0; h < 2; ++h)
#pragma unroll
    for(int r = 0; r < 4; ++r) As[(16 * h + mk + 4 * r) * 65 + 16 * wave + mi] = acc2[h][r];
  __syncthreads();
  if(wave != 0) return;
  // in-block back substitution (descending i), all columns of this tile at once
  double a2[NB];
#pragma unroll
  for(int r = 0; r < NB; ++r) a2[r] = As[r * 65 + lane];
#pragma unroll
  for(int ri = NB - 1; ri >= 0; --ri)
  {
    if(ri < nbk)
    {
      const double v = a2[ri] / Ld[ri + ri * (NB + 1)];
      if(live) AT(J, ldj, i0 + ri, c) = v;
#pragma unroll
      for(int r = 0; r < NB; ++r)
        if(r < ri) a2[r] = FMA(-Ld[ri + r * (NB + 1)], v, a2[r]);
    }
  }
}

// ============================================================================================================
// K3  x0 = -G^-1 a, f0 = a.x0 / 2: one workgroup, thread = row (strided), pivots through LDS
// ============================================================================================================
// ============================================================================================================
// column dots: out[j] = sign * sum_k M[k + j*ld] v[k]   (K4 with M = C, v = x; K5 with M = J, v = C[:,p])
// lane = column; the vector is staged in LDS; per lane the column is a contiguous stream.
// ============================================================================================================
// Tile geometry of the streaming column-dot kernels: a workgroup (one wave) owns CT = 16 columns; the matrix is
// streamed in tiles of KT = 64 rows x 16 columns.  ALL 64 lanes load (each column's 64 rows are 512 contiguous bytes:
// full cache lines), the tile lands in LDS, and lanes 0..15 run the 16 in-order chains out of LDS.  Compared with one
// lane per column streaming its own column, the matrix is spread over 4x as many CUs and every load is coalesced.
constexpr int CT = 16, KT = 64;

// acc <- in-order chain of up to 64 fmas, a[k * SA] * b[k], k = 0 .. cnt-1, out of LDS: the operands are fetched in four
// batches of 16, batch g+1 requested before the fmas of batch g, so that the chain pays one LDS latency per 64 steps
// instead of one per 8.  FULL: cnt == 64, no predication (a select per step would put one more dependent op on the chain).
template<int SA, bool FULL>
__device__ __forceinline__ double chain64(double acc, const double * a, const double * b, int cnt)
{
  double av[2][16], bv[2][16];
#pragma unroll
  for(int u = 0; u < 16; ++u)
  {
    const int k = FULL ? u : min(u, cnt - 1);
    av[0][u] = a[k * SA];
    bv[0][u] = b[k];
  }
#pragma unroll
  for(int g = 0; g < 4; ++g)
  {
    if(g < 3)
    {
#pragma unroll
      for(int u = 0; u < 16; ++u)
      {
        const int k = FULL ? ((g + 1) * 16 + u) : min((g + 1) * 16 + u, cnt - 1);
        av[(g + 1) & 1][u] = a[k * SA];
        bv[(g + 1) & 1][u] = b[k];
      }
    }
#pragma unroll
    for(int u = 0; u < 16; ++u)
    {
      const double nx = FMA(av[g & 1][u], bv[g & 1][u], acc);
      acc = (FULL || g * 16 + u < cnt) ? nx : acc;
    }
  }
  return acc;
}

// K3: x0 = -G^-1 a = -L^-T (L^-1 a), f0 = a.x0 / 2 -- two substitutions of n steps, blocked by 64 exactly as the back
// substitution of step_tail: wave 0 solves the 64 x 64 diagonal block out of statically indexed registers (lane = row,
// hoisted reciprocals, plain division on a failed range check), waves 1-7 advance the row blocks below (forward) / above
// (backward) through the block's 64 columns from tiles requested while wave 0 was solving; the running vector lives in LDS,
// two LDS-only barriers per block.  Row i's chain takes its terms in the same order as the unblocked loops (forward: k
// ascending, backward: k descending).  It replaces one workgroup barrier pair per STEP (1.2 ms at n = 1024).
//   TRANS = false: forward substitution with L (lower triangle of M);  TRANS = true: back substitution with L^T
template<bool TRANS>
__device__ __forceinline__ void blocked_subst(const double * __restrict__ M, int ld, int n, double * rs, int lane, int wave)
{
  // entry (i, k) of the triangular factor in use: L(i, k) = M(i, k), or L^T(i, k) = M(k, i)
  auto T = [&](int i, int k) -> double { return TRANS ? AT(M, ld, k, i) : AT(M, ld, i, k); };
  const int nblk = (n + 63) >> 6;
  double blk[64];
  double dg = 1.0;
  auto fetch_diag = [&](int cb) {
    const int kb = cb << 6, cnt = min(64, n - kb);
    const int row = kb + ((lane < cnt) ? lane : 0);
#pragma unroll
    for(int k = 0; k < 64; ++k) blk[k] = T(row, kb + ((k < cnt) ? k : 0));
    dg = (lane < cnt) ? AT(M, ld, row, row) : 1.0;
  };
  auto fetch_tile = [&](int rb, int kb, int cnt) {
    const int row = min((rb << 6) + lane, n - 1);
#pragma unroll
    for(int k = 0; k < 64; ++k) blk[k] = T(row, kb + ((k < cnt) ? k : 0));
  };
  const int cb0 = TRANS ? nblk - 1 : 0, cstep = TRANS ? -1 : 1;
  if(wave == 0) fetch_diag(cb0);
  lds_barrier();
  for(int it = 0, cb = cb0; it < nblk; ++it, cb += cstep)
  {
    const int kb = cb << 6, cnt = min(64, n - kb);
    // row blocks still to be advanced through this column block: above it (backward) or below it (forward)
    const int nrb = TRANS ? cb : nblk - 1 - cb;
    auto rbk = [&](int j) { return TRANS ? j : cb + 1 + j; }; // j-th of them
    const int j0 = wave - 1;
    if(wave > 0 && j0 < nrb) fetch_tile(rbk(j0), kb, cnt);
    if(wave == 0)
    {
      const double rk0 = (lane < cnt) ? rs[kb + lane] : 0.0;
      const double rinv = recip_refined(dg);
      // Steps as in wave_fwd / wave_bwd (device_common.h): the quotient of lane k is formed by every lane on its own entry
      // and read with one v_readlane pair, lane k keeps its numerator, all quotients are formed after the last step; no
      // branch per step (one per group of eight), steps beyond the block are masked.
      double rk = rk0;
#pragma unroll
      for(int gg = 0; gg < 8; ++gg)
      {
        const int g = TRANS ? 7 - gg : gg;
        if(g * 8 < cnt)
        {
#pragma unroll
          for(int uu = 0; uu < 8; ++uu)
          {
            const int k = g * 8 + (TRANS ? 7 - uu : uu);
            const double v = bcast_lane(div_pre(rk, dg, rinv), k);
            const double upd = FMA(-blk[k], v, rk);
            rk = (k < cnt && (TRANS ? lane < k : lane > k) && lane < cnt) ? upd : rk;
          }
        }
      }
      const double pre = rk;
      rk = div_pre(pre, dg, rinv);
      if(!__all(div_safe_den(dg) && div_safe_num(pre)))
      {
        rk = rk0;
#pragma unroll
        for(int kk = 0; kk < 64; ++kk)
        {
          const int k = TRANS ? 63 - kk : kk;
          if(k < cnt)
          {
            const double v = bcast_lane(rk, k) / bcast_lane(dg, k);
            const double upd = FMA(-blk[k], v, rk);
            rk = ((TRANS ? lane < k : lane > k) && lane < cnt) ? upd : rk;
            rk = (lane == k) ? v : rk;
          }
        }
      }
      if(lane < cnt) rs[kb + lane] = rk;
    }
    lds_barrier();
    if(wave == 0)
    {
      if(it + 1 < nblk) fetch_diag(cb + cstep);
    }
    else
    {
      for(int j = j0; j < nrb; j += 7)
      {
        const int rb = rbk(j);
        if(j != j0) fetch_tile(rb, kb, cnt);
        const int row = (rb << 6) + lane;
        double acc = rs[min(row, n - 1)];
#pragma unroll
        for(int g = 0; g < 8; ++g)
        {
          const int k0 = TRANS ? 56 - 8 * g : 8 * g;
          double rv[8];
#pragma unroll
          for(int t = 0; t < 8; ++t) rv[t] = rs[kb + min(k0 + t, cnt - 1)];
#pragma unroll
          for(int tt = 0; tt < 8; ++tt)
          {
            const int t = TRANS ? 7 - tt : tt;
            const double nx = FMA(-blk[k0 + t], rv[t], acc);
            acc = (k0 + t < cnt) ? nx : acc;
          }
        }
        if(row < n) rs[row] = acc;
      }
    }
    lds_barrier();
  }
}

__global__ __launch_bounds__(512) void x0_kernel(const double * __restrict__ L, int ldl, int n, const double * __restrict__ a, double * x, double * f0)
{
  extern __shared__ double xs[]; // n (the running vector) | n (a, for f0)
  double * as = xs + n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for(int i = tid; i < n; i += 512)
  {
    const double ai = a[i];
    xs[i] = ai;
    as[i] = ai;
  }
  blocked_subst<false>(L, ldl, n, xs, lane, wave); // (starts with a barrier)
  blocked_subst<true>(L, ldl, n, xs, lane, wave);
  for(int i = tid; i < n; i += 512)
  {
    const double xi = -xs[i];
    xs[i] = xi;
    x[i] = xi;
  }
  lds_barrier();
  if(wave == 0)
  {
    // f0 = a.x0 / 2: one in-order chain (every lane runs it on the same LDS words)
    double acc = 0.0;
    int k = 0;
    for(; k + 64 <= n; k += 64) acc = chain64<1, true>(acc, as + k, xs + k, 64);
    if(k < n) acc = chain64<1, false>(acc, as + k, xs + k, n - k);
    if(lane == 0) *f0 = 0.5 * acc;
  }
}

constexpr int NPRE = 3; // z_rows: tiles requested ahead of the one being consumed (registers)

// ---- column dots out of an LDS-direct tile stream (jtn: d = J^T n+; select_partial: C^T x) ------------------------------------
// A workgroup of TWO waves owns CT = 16 columns of M and runs their in-order chains over the n rows, 64 rows (a tile) at a time:
//   wave 1 (loader) streams the tiles into a ring of TS_NS LDS slots with global_load_lds_dwordx4 -- lane l fetches rows
//          2 (l mod 32), + 1 of column 2 i + (l div 32): an instruction moves two columns x 64 rows = 1 KiB, eight instructions a
//          tile, no registers -- up to TS_NS - 1 tiles ahead (round 2 kept three tiles in registers, two ahead: a tile's loads took
//          ~1.7 us to come back, the chains of a tile last 0.4 us, and the wave waited 0.9 us per tile); it alone waits for vmcnt;
//   wave 0 (chain)  lane c < 16 owns column c: one ds_read2_b64 fetches two rows of its column, the vector comes as broadcasts.
// A slot is [column pair][column in pair][row]: 1 KiB per pair as the loads deliver it, pairs 16 bytes apart so that the sixteen
// lanes of a step hit eight distinct bank groups (the two columns of a pair are 512 bytes apart: a two-way conflict, 2 cycles).
// One LDS-only barrier per tile.  Rows past n (last tile) are fetched from clamped addresses and never used; with an odd n the
// pair that starts at row n - 1 would read one double past its column -- possibly past the caller's buffer -- so it is fetched
// from rows n - 2, n - 1 instead and the loader then writes element n - 1 of its sixteen columns in place itself.
constexpr int TS_NS = 6;                 // slots
constexpr int TS_PAIR = 128 + 2;         // doubles between the column pairs of a slot
constexpr int TS_SLOT = (CT / 2) * TS_PAIR;
constexpr int TILE_THREADS = 128;
static_assert(8 * (TS_NS - 1) <= 63, "vmcnt");

// (t0 / t1: the tiles [t0, t1) only -- the blocked kernels of CANON C8 split the rows of a column group over several workgroups)
__device__ __forceinline__ void tile_loader(const double * __restrict__ M, size_t ld, int n, int col0, int ncols, double * slots, int lane, int t0 = 0,
                                            int t1 = -1)
{
  const int ntiles_all = (n + KT - 1) / KT;
  const int tb = t0, ntiles = (t1 < 0) ? ntiles_all : t1;
  const int kp = lane & 31, half = lane >> 5;
  const int cmax = col0 + ((ncols > 0) ? ncols - 1 : 0);
  auto issue = [&](int t) {
    if(n < 2) return; // (a single row: the odd-n path below fetches it; a 16-byte load could leave the caller's buffer)
    double * slot = slots + ((t - tb) % TS_NS) * TS_SLOT;
    int row = t * KT + 2 * kp;
    if(row + 1 >= n) row = n - 2; // past the end, or the pair that starts at the last row of an odd n (see above)
#pragma unroll
    for(int i = 0; i < CT / 2; ++i)
    {
      const double * src = M + (size_t)min(col0 + 2 * i + half, cmax) * ld + row;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(slot + i * TS_PAIR), 16, 0, 0);
    }
  };
  for(int t = tb; t < tb + TS_NS - 1 && t < ntiles; ++t) issue(t);
  for(int t = tb; t < ntiles; ++t)
  {
    if(t + TS_NS - 2 < ntiles) // the TS_NS - 2 tiles behind tile t are all requested: tile t has landed once only they are outstanding
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (TS_NS - 2)) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if((n & 1) && t == ntiles_all - 1 && lane < CT)
    {
      // odd n: element n - 1 of column `lane` (its pair was fetched one row low)
      const double vlast = M[(size_t)min(col0 + lane, cmax) * ld + (n - 1)];
      slots[((t - tb) % TS_NS) * TS_SLOT + (lane >> 1) * TS_PAIR + (lane & 1) * 64 + (n - 1 - t * KT)] = vlast;
    }
    lds_barrier(); // B_t
    if(t + TS_NS - 1 < ntiles) issue(t + TS_NS - 1);
  }
}

// the chain wave's side: acc of lane c < 16 = in-order chain of column col0 + c against vs (n entries, LDS)
__device__ __forceinline__ double tile_chain(int n, const double * vs, const double * slots, int lane)
{
  const int ntiles = (n + KT - 1) / KT;
  const int c = min(lane, CT - 1);
  const int coff = (c >> 1) * TS_PAIR + (c & 1) * 64;
  double acc = 0.0;
  for(int t = 0; t < ntiles; ++t)
  {
    lds_barrier(); // B_t: tile t is in its slot
    const double * a = slots + (t % TS_NS) * TS_SLOT + coff;
    const int k0 = t * KT, cnt = min(KT, n - k0);
    acc = (cnt == KT) ? chain64<1, true>(acc, a, vs + k0, cnt) : chain64<1, false>(acc, a, vs + k0, cnt);
  }
  return acc;
}

// wave-level wait for a pass tag (the chain wave of a two-wave kernel: the loader wave must not be held up)
__device__ __forceinline__ void flag_wait_wave(const int * flag, int tag, Mailbox * mb, int lane)
{
  if(lane == 0)
  {
    int spins = 0;
    while(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag)
    {
      __builtin_amdgcn_s_sleep(4);
      if(++spins > (1 << 20))
      {
        if(mb) mb->sync_err = 1;
        break;
      }
    }
  }
  wave_sync();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Hand-off of d (PassSync): what a jtn / jrow launch waits for before it touches psel and d, and what it publishes behind d
struct DSync
{
  PassSync * sync; // nullptr: plain launch
  Mailbox * mb;
  int wait_sel, wait_chain; // tags (0: none)
  int post, total;          // d_tag = post once `total` workgroups (of jtn and jrow together) have arrived
};

__global__ __launch_bounds__(TILE_THREADS) void jtn_kernel(const double * __restrict__ M, int ld, int n, int ncols, const double * __restrict__ v, double sign,
                                                           double * out, const int * psel = nullptr, const double * vbase = nullptr, int ldv = 0, int m = 0,
                                                           DSync ds = DSync{nullptr, nullptr, 0, 0, 0, 0})
{
  extern __shared__ double sh[]; // vector (n) | tile slots
  double * vs = sh;
  double * slots = sh + n + (n & 1); // (16-byte aligned: the loads deliver 16 bytes per lane)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col0 = blockIdx.x * CT;
  const int nc = min(CT, ncols - col0);
  if(wave == 1)
    tile_loader(M, (size_t)ld, n, col0, nc, slots, lane); // (M is complete when the kernel starts: the tiles do not wait for the selection)
  else
  {
    const unsigned long long tj0 = ROTT();
    if(ds.sync)
    {
      if(ds.wait_sel) flag_wait_wave(&ds.sync->sel_tag, ds.wait_sel, ds.mb, lane);
      if(ds.wait_chain) flag_wait_wave(&ds.sync->chain_tag, ds.wait_chain, ds.mb, lane);
    }
    const unsigned long long tj1 = ROTT();
    bool live = true;
    if(psel)
    {
      const int p = psel[0];
      live = !(p < 0 || p >= m); // none selected, or a bound (jrow_kernel)
      if(live)
      {
        v = vbase + (size_t)p * ldv;
        sign = (psel[1] == UPPER) ? -1.0 : 1.0;
      }
    }
    if(live)
      for(int i = lane; i < n; i += 64) vs[i] = v[i];
    const unsigned long long tj2 = ROTT();
    const double acc = tile_chain(n, vs, slots, lane); // (not live: the chains run on whatever LDS holds -- the loader's barriers need their partner)
    if(live && lane < nc) out[col0 + lane] = (sign < 0.0) ? -acc : acc;
    const unsigned long long tj3 = ROTT();
    ROTP(16, tj1 - tj0); // jtn: waiting for the selection / a discarded chain
    ROTP(17, tj2 - tj1); // jtn: psel, the constraint's column into LDS
    ROTP(18, tj3 - tj2); // jtn: the tiles
    ROTP(20, 1);
  }
  if(ds.sync && ds.post) flag_post(&ds.sync->d_cnt, &ds.sync->d_tag, ds.total, ds.post);
}


// ---- CANON C8 (oracle orc_dot_blk): the column dots BLOCKED, for more than kBlockN unknowns ---------------------------------------
// A column's sum is cut into segments of 64 rows -- exactly the tiles of the stream above --, each segment an ascending fma chain from
// +0, the segment sums combined by the adjacent-first pairwise tree.  The chain of an output is then 64 steps instead of n, and the
// rows of a column group are split over RS workgroups (grid.y): every CU streams its share of M.  Each workgroup leaves its segment
// sums in P[tile][column]; a second, small kernel behind it in the stream combines them (lane = output, the oracle's tree over 64
// slots in registers, the slots beyond the last segment +0).  (First version: the workgroup that arrived last at a per-group counter
// combined -- one agent-scope release per workgroup, i.e. an L2 write-back on this eight-XCD part, 384 of them beside a sweep that
// is dirtying 8 MiB of J: select 20 -> 52 us, jtn 20 -> 41 us.  A kernel boundary costs 1.5 us and no fence.)
constexpr int kBlockN = 256;
struct BlkArgs
{
  double * P; // segment sums: P[t * ldp + column]
  int ldp;
  int * cnt;  // one arrival counter per column group (zero between launches)
  int tps;    // tiles per workgroup
};
__device__ __forceinline__ void tile_chain_blk(int n, const double * vs, const double * slots, int lane, int t0, int t1, const BlkArgs & ba, int col0, int nc)
{
  const int c = min(lane, CT - 1);
  const int coff = (c >> 1) * TS_PAIR + (c & 1) * 64;
  for(int t = t0; t < t1; ++t)
  {
    lds_barrier(); // B_t: tile t is in its slot
    const double * a = slots + ((t - t0) % TS_NS) * TS_SLOT + coff;
    const int k0 = t * KT, cnt = min(KT, n - k0);
    const double acc = (cnt == KT) ? chain64<1, true>(0.0, a, vs + k0, cnt) : chain64<1, false>(0.0, a, vs + k0, cnt);
    if(lane < nc) ba.P[(size_t)t * ba.ldp + col0 + lane] = acc;
  }
}
__global__ __launch_bounds__(TILE_THREADS) void jtn_blk_kernel(const double * __restrict__ M, int ld, int n, int ncols, const double * __restrict__ v, double sign,
                                                               double * out, const int * psel, const double * vbase, int ldv, int m, DSync ds, BlkArgs ba)
{
  extern __shared__ double sh[]; // vector (n) | tile slots
  double * vs = sh;
  double * slots = sh + n + (n & 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col0 = blockIdx.x * CT;
  const int nc = min(CT, ncols - col0);
  const int ntiles = (n + KT - 1) / KT;
  const int t0 = blockIdx.y * ba.tps, t1 = min(ntiles, t0 + ba.tps);
  if(wave == 1)
    tile_loader(M, (size_t)ld, n, col0, nc, slots, lane, t0, t1);
  else
  {
    if(ds.sync)
    {
      if(ds.wait_sel) flag_wait_wave(&ds.sync->sel_tag, ds.wait_sel, ds.mb, lane);
      // (a discarded chain that may still be reading d is waited for by the combine kernel: this one only writes segment sums)
    }
    bool live = true;
    if(psel)
    {
      const int p = psel[0];
      live = !(p < 0 || p >= m); // none selected, or a bound (jrow_kernel)
      if(live) v = vbase + (size_t)p * ldv;
    }
    const int r0 = t0 * KT, r1 = min(n, t1 * KT);
    if(live)
      for(int i = r0 + lane; i < r1; i += 64) vs[i] = v[i];
    tile_chain_blk(n, vs, slots, lane, t0, t1, ba, col0, nc); // (not live: chains on whatever LDS holds -- the loader's barriers need their partner)
  }
}

// K4 partial, blocked: C^T x in segments (C8), then per CT constraints the first-index arg-min of min(c.x - bl, bu - c.x) below 0
__global__ __launch_bounds__(TILE_THREADS) void select_blk_kernel(const double * __restrict__ C, int ldc, int n, int m, const double * __restrict__ x,
                                                                  const double * __restrict__ bl, const double * __restrict__ bu,
                                                                  const unsigned char * __restrict__ st, int excl, double * pval, int * pidx, int * pstat,
                                                                  const PassSync * sync, int wait_x, Mailbox * mb, BlkArgs ba, const StepDev * gate = nullptr,
                                                                  int gate_tag = 0)
{
  extern __shared__ double sh[];
  double * vs = sh;
  double * slots = sh + n + (n & 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col0 = blockIdx.x * CT;
  const int nc = min(CT, m - col0);
  const int ntiles = (n + KT - 1) / KT;
  const int t0 = blockIdx.y * ba.tps, t1 = min(ntiles, t0 + ba.tps);
  if(gate && !gate_is_add(gate, gate_tag)) return; // (behind decision_gate_kernel: the pass does not add -- no scan)
  if(wave == 1)
    tile_loader(C, (size_t)ldc, n, col0, nc, slots, lane, t0, t1); // (C is the caller's: its tiles start at once, beside the wait for x)
  else
  {
    if(sync && wait_x) flag_wait_wave(&sync->x_tag, wait_x, mb, lane); // x of this pass comes from `advance` / the sweep on another stream
    const int r0 = t0 * KT, r1 = min(n, t1 * KT);
    for(int i = r0 + lane; i < r1; i += 64) vs[i] = x[i];
    tile_chain_blk(n, vs, slots, lane, t0, t1, ba, col0, nc);
  }
}

// The tree over the segment sums of an output (CANON C8), lane = output: P[t * ldp + o], t < nseg <= 64, the slots beyond nseg +0 --
// so the tree over the first power of two >= nseg slots is the whole tree (x + (+0) = x, and a segment sum is never -0)
template<int SL>
__device__ __forceinline__ double blk_tree_n(const double * P, int ldp, int nseg, int o)
{
  double p[SL];
#pragma unroll
  for(int t = 0; t < SL; ++t) p[t] = P[(size_t)min(t, nseg - 1) * ldp + o]; // (clamped, not predicated: every load in flight at once)
#pragma unroll
  for(int t = 0; t < SL; ++t) p[t] = (t < nseg) ? p[t] : 0.0;
#pragma unroll
  for(int h = 1; h < SL; h <<= 1)
#pragma unroll
    for(int i = 0; i < SL; i += 2 * h) p[i] = p[i] + p[i + h];
  return p[0];
}
// The C8 tree over the segment sums of row-chunk c of a triangular product (CANON C10: step_tail_kernel): slot s holds Pc[64 s] for
// c <= s < nseg and +0 otherwise
template<int SL>
__device__ __forceinline__ double tri_tree(const double * Pc, int c, int nseg)
{
  double p[SL];
#pragma unroll
  for(int t = 0; t < SL; ++t) p[t] = Pc[(size_t)min(max(t, c), nseg - 1) * 64];
#pragma unroll
  for(int t = 0; t < SL; ++t) p[t] = (t >= c && t < nseg) ? p[t] : 0.0;
#pragma unroll
  for(int h = 1; h < SL; h <<= 1)
#pragma unroll
    for(int i = 0; i < SL; i += 2 * h) p[i] = p[i] + p[i + h];
  return p[0];
}
__device__ __forceinline__ double blk_tree(const double * P, int ldp, int nseg, int o)
{
  if(nseg <= 8) return blk_tree_n<8>(P, ldp, nseg, o);
  if(nseg <= 16) return blk_tree_n<16>(P, ldp, nseg, o);
  if(nseg <= 32) return blk_tree_n<32>(P, ldp, nseg, o);
  return blk_tree_n<64>(P, ldp, nseg, o);
}
// two outputs at once, the loads of both in flight together (the step kernel's z: one latency instead of two)
template<int SL>
__device__ __forceinline__ void blk_tree2_n(const double * P, int ldp, int nseg, int o0, int o1, double & r0, double & r1)
{
  double p[SL], w[SL];
#pragma unroll
  for(int t = 0; t < SL; ++t) p[t] = P[(size_t)min(t, nseg - 1) * ldp + o0];
#pragma unroll
  for(int t = 0; t < SL; ++t) w[t] = P[(size_t)min(t, nseg - 1) * ldp + o1];
#pragma unroll
  for(int t = 0; t < SL; ++t) p[t] = (t < nseg) ? p[t] : 0.0;
#pragma unroll
  for(int t = 0; t < SL; ++t) w[t] = (t < nseg) ? w[t] : 0.0;
#pragma unroll
  for(int h = 1; h < SL; h <<= 1)
#pragma unroll
    for(int i = 0; i < SL; i += 2 * h)
    {
      p[i] = p[i] + p[i + h];
      w[i] = w[i] + w[i + h];
    }
  r0 = p[0];
  r1 = w[0];
}
// d = +-(tree of the segment sums), 64 outputs per workgroup (one wave); psel != nullptr: sign and "is there a general constraint at
// all" from the selection on the device; publishes d_tag like jtn_kernel.  Also the combine of z (plain: sign 1, no psel).
__global__ __launch_bounds__(64) void blk_comb_kernel(const double * __restrict__ P, int ldp, int nseg, int nout, double sign, double * out, const int * psel,
                                                      int m, DSync ds, PassSync * zsync = nullptr, int post_z = 0)
{
  const int o = blockIdx.x * 64 + threadIdx.x;
  if(ds.sync && ds.wait_chain) flag_wait(&ds.sync->chain_tag, ds.wait_chain, ds.mb); // a discarded chain may still be reading d
  bool live = true;
  if(psel)
  {
    const int p = psel[0];
    live = !(p < 0 || p >= m);
    sign = (psel[1] == UPPER) ? -1.0 : 1.0;
  }
  if(live && o < nout)
  {
    const double r = blk_tree(P, ldp, nseg, o);
    out[o] = (sign < 0.0) ? -r : r;
  }
  if(ds.sync && ds.post) flag_post(&ds.sync->d_cnt, &ds.sync->d_tag, ds.total, ds.post);
  if(zsync && post_z) flag_post(&zsync->z_cnt, &zsync->z_tag, gridDim.x, post_z);
}
// K4 partial behind select_blk_kernel: c.x of CT constraints per workgroup (the tree), then the first-index arg-min of
// min(c.x - bl, bu - c.x) below 0 -- what select_partial_kernel leaves for select_final_kernel.  (Folded into select_final_kernel it
// was one launch less and 29 us instead of 3 + 6: 2048 trees on one workgroup.)
__global__ __launch_bounds__(64) void select_comb_kernel(const double * __restrict__ P, int ldp, int nseg, int m, const double * __restrict__ bl,
                                                         const double * __restrict__ bu, const unsigned char * __restrict__ st, int excl, double * pval,
                                                         int * pidx, int * pstat, const StepDev * gate = nullptr, int gate_tag = 0, const int * psel = nullptr)
{
  if(gate)
  {
    if(!gate_is_add(gate, gate_tag)) return;
    excl = psel[0]; // the constraint being activated (its status byte is written by the sweep, concurrently)
  }
  const int lane = threadIdx.x;
  const int i = blockIdx.x * CT + lane;
  const bool on = lane < CT && i < m;
  const double cx = blk_tree(P, ldp, nseg, on ? i : 0);
  double bval = 0.0;
  int bidx = NONE, bstat = INACTIVE;
  if(on && st[i] == INACTIVE && i != excl) // `excl`: a constraint being activated concurrently (its status byte may not be set yet)
  {
    double sl = cx - bl[i];
    if(sl < 0.0)
    {
      bval = sl;
      bidx = i;
      bstat = LOWER;
    }
    else
    {
      double su = bu[i] - cx;
      if(su < 0.0)
      {
        bval = su;
        bidx = i;
        bstat = UPPER;
      }
    }
  }
  wave_argmin(bval, bidx, bstat);
  if(lane == 0)
  {
    pval[blockIdx.x] = bval;
    pidx[blockIdx.x] = bidx;
    pstat[blockIdx.x] = bstat;
  }
}

// K6 blocked (C8): z_i = sum over the columns q .. n-1 in segments of 64 by ABSOLUTE column index.  One wave per (64 rows, segment):
// lane = row (a column's 64 rows are 512 contiguous bytes), the segment's 64 entries of the row in registers -- all loads in flight at
// once --, d2 through LDS with the entries outside [q, n) replaced by +0 (a term fma(J_ij, +0, acc) leaves acc as it is).  The wave of
// a row group that arrives last adds the segment sums of its 64 rows (lane = row: the tree in registers).
__global__ __launch_bounds__(64) void z_blk_kernel(const double * __restrict__ J, int ld, int n, int q, const double * d, double * z,
                                                   PassSync * sync, int wait_d, int post_z, Mailbox * mb, BlkArgs ba, const double * __restrict__ Pd = nullptr,
                                                   int ldpd = 0, int nsegd = 0, const int * psel = nullptr, int m = 0, double sign = 1.0, double * dout = nullptr,
                                                   const PassSync * csync = nullptr, int wait_chain = 0, Mailbox * cmb = nullptr)
{
  __shared__ double dsm[64];
  // (the workgroups that store d: a coefficient chain of the pass before, discarded by a removal, may still be reading it)
  if(Pd && blockIdx.x == 0 && csync && wait_chain) flag_wait(&csync->chain_tag, wait_chain, cmb);
  if(sync && wait_d) flag_wait(&sync->d_tag, wait_d, mb); // on its own stream, beside the substitution: d (and with it J) of this pass
  const int lane = threadIdx.x;
  const int row = min((int)blockIdx.x * 64 + lane, n - 1);
  const int s = blockIdx.y, j0 = s * 64;
  const int j = j0 + lane;
  // Pd != nullptr (round 4): d has not been combined yet -- jtn_blk_kernel left the segment sums of d = J^T n+ in Pd, and this kernel
  // applies the C4 tree to the 64 entries of its segment itself (what blk_comb_kernel did in a launch of its own, on the critical path
  // of every pass); the workgroups of row group 0 also store d for the kernels behind (substitution, commit, coefficient chain).  With
  // the selection on the device (psel) the sign comes from there, and a BOUND has no segment sums: jrow_kernel wrote d itself.
  bool gen = Pd != nullptr;
  if(gen && psel)
  {
    const int p = psel[0];
    gen = !(p < 0 || p >= m);
    sign = (psel[1] == UPPER) ? -1.0 : 1.0;
  }
  const bool keeper = gen && blockIdx.x == 0; // (this workgroup stores its segment of d)
  double dv = 0.0;
  if((j0 + 64 > q || keeper) && j < n)
  {
    if(gen)
    {
      const double t = blk_tree(Pd, ldpd, nsegd, j);
      dv = (sign < 0.0) ? -t : t;
    }
    else
      dv = d[j];
  }
  if(keeper && j < n) dout[j] = dv;
  double acc = 0.0;
  if(j0 + 64 > q) // (a segment entirely below q has no term: +0)
  {
    double a[64];
#pragma unroll
    for(int t = 0; t < 64; ++t) a[t] = AT(J, ld, row, min(j0 + t, n - 1));
    dsm[lane] = (j >= q && j < n) ? dv : 0.0;
    wave_sync();
#pragma unroll
    for(int t0 = 0; t0 < 64; t0 += 8)
    {
      double dvv[8];
#pragma unroll
      for(int t = 0; t < 8; ++t) dvv[t] = dsm[t0 + t];
#pragma unroll
      for(int t = 0; t < 8; ++t) acc = FMA(a[t0 + t], dvv[t], acc);
    }
  }
  ba.P[(size_t)s * ba.ldp + blockIdx.x * 64 + lane] = acc; // (ldp >= 64 * gridDim.x); the step kernel adds the segment sums
}

// bound constraint: d = +-row pb of J
__global__ void jrow_kernel(const double * __restrict__ J, int ld, int n, int pb, double sign, double * out, const int * psel = nullptr, int m = 0,
                            DSync ds = DSync{nullptr, nullptr, 0, 0, 0, 0})
{
  if(ds.sync)
  {
    if(ds.wait_sel) flag_wait(&ds.sync->sel_tag, ds.wait_sel, ds.mb);
    if(ds.wait_chain) flag_wait(&ds.sync->chain_tag, ds.wait_chain, ds.mb);
  }
  bool live = true;
  if(psel)
  {
    const int p = psel[0];
    live = !(p < m); // none selected, or a general constraint (jtn_kernel)
    pb = p - m;
    sign = (psel[1] == UPPER_BOUND) ? -1.0 : 1.0;
  }
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if(live && j < n)
  {
    double v = AT(J, ld, pb, j);
    out[j] = (sign < 0.0) ? -v : v;
  }
  if(ds.sync && ds.post) flag_post(&ds.sync->d_cnt, &ds.sync->d_tag, ds.total, ds.post);
}

// K4 partial: per CT constraints the first-index arg-min of min(c.x - bl, bu - c.x) below 0
__global__ __launch_bounds__(TILE_THREADS) void select_partial_kernel(const double * __restrict__ C, int ldc, int n, int m, const double * __restrict__ x,
                                                                      const double * __restrict__ bl, const double * __restrict__ bu,
                                                                      const unsigned char * __restrict__ st, int excl, double * pval, int * pidx, int * pstat,
                                                                      const PassSync * sync = nullptr, int wait_x = 0, Mailbox * mb = nullptr)
{
  extern __shared__ double sh[];
  double * vs = sh;
  double * slots = sh + n + (n & 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col0 = blockIdx.x * CT;
  const int nc = min(CT, m - col0);
  if(wave == 1)
  {
    tile_loader(C, (size_t)ldc, n, col0, nc, slots, lane); // (C is the caller's: its tiles start at once, beside the wait for x)
    return;
  }
  if(sync && wait_x) flag_wait_wave(&sync->x_tag, wait_x, mb, lane); // x of this pass comes from `advance` on another stream
  for(int i = lane; i < n; i += 64) vs[i] = x[i];
  const double cx = tile_chain(n, vs, slots, lane);
  const int i = col0 + lane;
  double bval = 0.0;
  int bidx = NONE, bstat = INACTIVE;
  if(lane < nc && st[i] == INACTIVE && i != excl) // `excl`: a constraint being activated concurrently (its status byte may not be set yet)
  {
    double sl = cx - bl[i];
    if(sl < 0.0)
    {
      bval = sl;
      bidx = i;
      bstat = LOWER;
    }
    else
    {
      double su = bu[i] - cx;
      if(su < 0.0)
      {
        bval = su;
        bidx = i;
        bstat = UPPER;
      }
    }
  }
  wave_argmin(bval, bidx, bstat);
  if(lane == 0)
  {
    pval[blockIdx.x] = bval;
    pidx[blockIdx.x] = bidx;
    pstat[blockIdx.x] = bstat;
  }
}

// K4 final: combine the partials (in index order) and scan the bounds; result to the host-visible mailbox
__global__ __launch_bounds__(256) void select_final_kernel(int nparts, const double * pval, const int * pidx, const int * pstat, int m, int nb,
                                                          const double * __restrict__ x, const double * __restrict__ xl,
                                                          const double * __restrict__ xu, const unsigned char * __restrict__ st, int excl, Mailbox * mb,
                                                          int * psel, double * uq = nullptr, PassSync * sync = nullptr, int wait_x = 0, int post_sel = 0,
                                                          const StepDev * gate = nullptr, int gate_tag = 0)
{
  if(gate)
  {
    if(!gate_is_add(gate, gate_tag)) return;
    excl = psel[0];
  }
  if(sync && wait_x) flag_wait(&sync->x_tag, wait_x, mb); // (m == 0: no partial kernel ran ahead of this one)
  if(uq && threadIdx.x == 0) *uq = 0.0; // u[q] = 0 for the constraint about to be selected (DualSolver.cpp:121)
  __shared__ double sv[4];
  __shared__ int si[4], ss[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double bval = 0.0;
  int bidx = NONE, bstat = INACTIVE;
  for(int k = tid; k < nparts; k += 256)
  {
    int oi = pidx[k];
    double ov = pval[k];
    if(oi != NONE && (bidx == NONE || ov < bval || (ov == bval && oi < bidx)))
    {
      bval = ov;
      bidx = oi;
      bstat = pstat[k];
    }
  }
  for(int i = tid; i < nb; i += 256)
  {
    if(st[m + i] != INACTIVE || (m + i) == excl) continue;
    double xi = x[i];
    double sl = xi - xl[i];
    double cand;
    int cs;
    if(sl < 0.0) { cand = sl; cs = LOWER_BOUND; }
    else
    {
      double su = xu[i] - xi;
      if(!(su < 0.0)) continue;
      cand = su;
      cs = UPPER_BOUND;
    }
    // general constraints (index < m) win ties against bounds; bounds among themselves: first index
    if(bidx == NONE || cand < bval || (cand == bval && (m + i) < bidx))
    {
      bval = cand;
      bidx = m + i;
      bstat = cs;
    }
  }
  wave_argmin(bval, bidx, bstat);
  if(lane == 0)
  {
    sv[wave] = bval;
    si[wave] = bidx;
    ss[wave] = bstat;
  }
  __syncthreads();
  if(tid == 0)
  {
    double v = sv[0];
    int ix = si[0], sx = ss[0];
    for(int w = 1; w < 4; ++w)
      if(si[w] != NONE && (ix == NONE || sv[w] < v || (sv[w] == v && si[w] < ix)))
      {
        v = sv[w];
        ix = si[w];
        sx = ss[w];
      }
    mb->p = (ix == NONE) ? -1 : ix;
    mb->pstatus = sx;
    psel[0] = (ix == NONE) ? -1 : ix; // device copy: the step kernels of this pass read it without a host round trip
    psel[1] = sx;
  }
  if(sync && post_sel) flag_post(nullptr, &sync->sel_tag, 1, post_sel);
}

// K6  z = J2 d2.  A workgroup (one wave) owns 16 rows; all 64 lanes load (lane = (column offset, row): 4 columns x 16
// rows per instruction, 128 contiguous bytes per column), the 16 x 64 tile lands in LDS, lanes 0..15 run the
// in-order chains over columns q .. n-1.
__global__ __launch_bounds__(64) void z_rows_kernel(const double * __restrict__ J, int ld, int n, int q, const double * __restrict__ d, double * z,
                                                    PassSync * sync = nullptr, int wait_d = 0, int post_z = 0, Mailbox * mb = nullptr)
{
  __shared__ double tile[64 * 17]; // [column in tile][row], padded
  __shared__ double dsm[64];
  if(sync && wait_d) flag_wait(&sync->d_tag, wait_d, mb); // on its own stream, beside the substitution: d (and with it J) of this pass
  const int lane = threadIdx.x;
  const int r0 = blockIdx.x * 16;
  const int rr = lane & 15, cg = lane >> 4; // row in tile, column group
  double acc = 0.0;
  // (NPRE tiles in flight; clamped, not predicated, loads and a branch-free steady state: see tile_column_dot)
  double ring[NPRE][16], dring[NPRE];
  const int row = min(r0 + rr, n - 1);
  auto gload = [&](double (&dst)[16], double & dd, int j0) {
#pragma unroll
    for(int t = 0; t < 16; ++t) dst[t] = AT(J, ld, row, min(j0 + t * 4 + cg, n - 1));
    dd = d[min(j0 + lane, n - 1)];
  };
  auto consume = [&](double (&buf)[16], double & dd, int j0) {
    const int cnt = min(64, n - j0);
    wave_sync();
#pragma unroll
    for(int t = 0; t < 16; ++t) tile[(t * 4 + cg) * 17 + rr] = buf[t];
    dsm[lane] = dd;
    wave_sync();
    gload(buf, dd, j0 + NPRE * 64);
    const double * tr = tile + (lane & 15);
    const double nx = (cnt == 64) ? chain64<17, true>(acc, tr, dsm, cnt) : chain64<17, false>(acc, tr, dsm, cnt);
    acc = (lane < 16) ? nx : acc;
  };
#pragma unroll
  for(int s = 0; s < NPRE; ++s) gload(ring[s], dring[s], q + s * 64);
  int jb = q;
  for(; jb + NPRE * 64 <= n; jb += NPRE * 64)
  {
#pragma unroll
    for(int s = 0; s < NPRE; ++s) consume(ring[s], dring[s], jb + s * 64);
  }
#pragma unroll
  for(int s = 0; s < NPRE; ++s)
    if(jb + s * 64 < n) consume(ring[s], dring[s], jb + s * 64);
  if(lane < 16 && r0 + lane < n) z[r0 + lane] = acc;
  if(sync && post_z) flag_post(&sync->z_cnt, &sync->z_tag, gridDim.x, post_z);
}

// K7 + K8, two workgroups of 8 waves.
//
// block 0: r = R^-1 d1 and the ratio test.  The back substitution is blocked by 64 columns; row i's chain
// (acc = d_i; acc = fma(-R(i,k), r_k, acc) for k = q-1 .. i+1; r_i = acc / R(i,i)) is untouched, only WHO advances it
// changes: wave 0 solves the diagonal block of column block cb out of registers (lane = row, the block's row in 64
// statically indexed registers, hoisted reciprocals: ~5 dependent ops per step), then waves 1-7 each take a row block
// above it and advance those 64 chains through the block's 64 columns -- their tile of R was fetched into registers
// while wave 0 was solving, and wave 0 fetches the next diagonal block while they update.  Two LDS-only barriers per
// column block; the running right-hand side lives in LDS.
//
// block 1: z.z and n.z as C4 trees over the whole workgroup (CANON C5), n.x as an in-order chain in one lane beside them; the
// operands are staged in LDS by the whole workgroup first (a single lane chasing them through global memory pays one L2
// latency per eight entries).
__global__ __launch_bounds__(512) void step_tail_kernel(const double * __restrict__ R, int ld, int n, int q, const double * __restrict__ d, double * r,
                                                        const double * __restrict__ u, const unsigned char * __restrict__ st, double big,
                                                        const double * __restrict__ z, const double * __restrict__ x, const double * __restrict__ cp,
                                                        int pb, Mailbox * mb, const int * psel = nullptr, const double * cbase = nullptr, int ldc = 0,
                                                        int m = 0, int tag = 0, const PassSync * zsync = nullptr, int wait_z = 0,
                                                        StepDev * sd = nullptr, const double * gbl = nullptr, const double * gbu = nullptr,
                                                        const double * gxl = nullptr, const double * gxu = nullptr, const double * Pz = nullptr,
                                                        int ldpz = 0, int nsegz = 0, double * zout = nullptr, PassSync * dsync = nullptr, int post_d = 0,
                                                        const double * __restrict__ Sinv = nullptr, double * Pg = nullptr)
{
  extern __shared__ double rs[]; // block 0: the running right-hand side (q) ; block 1: z | x | n+ (3 n)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (round 4, blocked d: d was completed by the kernel in front of this one in the stream -- z_blk_kernel -- so the kernel boundary has
  // ordered it; whoever runs beside the main stream on d, the coefficient chain, is told here)
  if(dsync && post_d && blockIdx.x == 0 && tid == 0) __hip_atomic_store(&dsync->d_tag, post_d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  // the pass tag goes out last, behind a system-scope fence: the host polls it (gi_dense_host.cpp)
  auto post_tag = [&]() {
    __threadfence_system();
    if(tid == 0)
    {
      if(blockIdx.x == 0)
        mb->seq0 = tag;
      else
        mb->seq1 = tag;
    }
  };
  if(psel)
  {
    const int p = psel[0];
    if(blockIdx.x == 0 && tid == 0)
    {
      mb->sp = p;
      mb->sps = psel[1];
    }
    if(p < 0) // nothing violated: the host terminates on the p it reads with this pass's scalars
    {
      if(sd && blockIdx.x == 0 && tid == 0) // (a scan waiting at its gate learns that there is no decision to act on)
      {
        sd->kind = kKindNone;
        __threadfence();
        __hip_atomic_store(&sd->tag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      post_tag();
      return;
    }
    cp = (p < m) ? cbase + (size_t)p * ldc : nullptr;
    pb = p - m;
  }
  // The decision on the device (StepDev, gi_dense.h): called by the ONE thread of each workgroup that has just written that
  // workgroup's scalars; the second to arrive has both halves and forms t2, t and the kind of step exactly as the host does
  // (gi_dense_host.cpp; DualSolver.cpp:127-151; GoldfarbIdnaniSolver.cpp:174-218).
  // (the bound the decision compares with: fetched by the deciding lanes while the workgroup computes, not behind the hand-shake)
  double bsel = 0.0;
  if(sd && psel && tid == 0)
  {
    const int p = psel[0], ps = psel[1];
    if(ps == LOWER || ps == EQUALITY)
      bsel = gbl[p];
    else if(ps == UPPER)
      bsel = gbu[p];
    else if(ps == LOWER_BOUND || ps == FIXED)
      bsel = gxl[p - m];
    else
      bsel = gxu[p - m];
  }
  auto decide = [&]() {
    if(!sd || !psel) return;
    __threadfence();
    const int old = __hip_atomic_fetch_add(&sd->arrive, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if(old != 1) return;
    __hip_atomic_store(&sd->arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const volatile StepDev * sv = sd; // (behind the acquire fence: the other workgroup's half, out of L2)
    const double t1v = sv->t1, zzv = sv->zz, czv = sv->cz, cxv = sv->cx;
    const double b = bsel;
    double t2v = big;
    if(__builtin_sqrt(zzv) > 1e-14) t2v = (b - cxv) / czv;
    const double tv = (t2v < t1v) ? t2v : t1v;
    const int kind = (tv >= big) ? kKindInfeasible : ((t2v >= big) ? kKindDual : ((tv == t2v) ? kKindAdd : kKindDrop));
    sd->t = tv;
    sd->kind = kind;
    mb->dkind = kind;
    __threadfence();
    __hip_atomic_store(&sd->tag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  };
  if(blockIdx.x == 1)
  {
    // (x and n+ staged with one pad entry per 64: lane s of the blocked n+.x below walks segment s, a stride of 64 doubles would put
    // the 64 lanes of a step on one LDS bank)
    const int npad = n + (n >> 6) + 1;
    auto pd = [](int i) { return i + (i >> 6); };
    double * zl = rs;
    double * xl = rs + n;
    double * cl = xl + npad;
    const unsigned long long tb0 = ROTT();
    for(int i = tid; i < n; i += 512)
    {
      xl[pd(i)] = x[i];
      cl[pd(i)] = cp ? cp[i] : 0.0;
    }
    __syncthreads();
    // n+.x (bit for bit the scan's value for the constraint) on wave 1 -- it needs neither z nor r, and runs while z is still being
    // computed (zsync: z_rows runs on another stream, beside the substitution of block 0).  Up to kBlockN unknowns the C1 chain in one
    // lane; beyond, CANON C8: lane s runs the chain of segment s (64 terms), then the tree over the lanes.
    if(wave == 1 && n > kBlockN)
    {
      double acc = 0.0;
      if(cp)
      {
        const int k0 = lane * 64, cnt = min(64, n - k0); // (n <= 4096: at most 64 segments)
        if(cnt > 0)
        {
          const double *ca = cl + pd(k0), *xa = xl + pd(k0);
          for(int k = 0; k + 8 <= cnt; k += 8)
          {
            double av[8], bv[8];
#pragma unroll
            for(int t = 0; t < 8; ++t)
            {
              av[t] = ca[k + t];
              bv[t] = xa[k + t];
            }
#pragma unroll
            for(int t = 0; t < 8; ++t) acc = FMA(av[t], bv[t], acc);
          }
          for(int k = cnt & ~7; k < cnt; ++k) acc = FMA(ca[k], xa[k], acc);
        }
        acc = wave_sum_tree(acc);
      }
      else
        acc = xl[pd(pb)];
      if(lane == 0)
      {
        mb->cx = acc;
        if(sd) sd->cx = acc;
      }
    }
    else if(tid == 64 && n <= kBlockN)
    {
      double acc = 0.0;
      if(cp)
      {
        int k = 0;
        for(; k + 16 <= n; k += 16)
        {
          double av[16], bv[16];
#pragma unroll
          for(int t = 0; t < 16; ++t)
          {
            av[t] = cl[pd(k + t)];
            bv[t] = xl[pd(k + t)];
          }
#pragma unroll
          for(int t = 0; t < 16; ++t) acc = FMA(av[t], bv[t], acc);
        }
        for(; k < n; ++k) acc = FMA(cl[pd(k)], xl[pd(k)], acc);
      }
      else
        acc = xl[pd(pb)];
      mb->cx = acc;
      if(sd) sd->cx = acc;
    }
    if(zsync && wait_z) flag_wait(&zsync->z_tag, wait_z, mb);
    const unsigned long long tb1 = ROTT();
    if(Pz)
    {
      // CANON C8: z_blk_kernel (ahead of this kernel in the stream) left the segment sums of z: the tree here, z itself for the update
      if(nsegz <= 32)
      {
        // (two rows per thread and trip, their 2 nseg loads in flight together: the segment sums come from the other XCDs' kernel --
        // memory, not L2 -- and a trip costs one such latency; measured 7.7 us of the 15 us of this workgroup with one row per trip)
        for(int i0 = tid; i0 < n; i0 += 1024)
        {
          const int i1 = i0 + 512, i1c = min(i1, n - 1);
          double z0, z1;
          if(nsegz <= 16)
            blk_tree2_n<16>(Pz, ldpz, nsegz, i0, i1c, z0, z1);
          else
            blk_tree2_n<32>(Pz, ldpz, nsegz, i0, i1c, z0, z1);
          zl[i0] = z0;
          zout[i0] = z0;
          if(i1 < n)
          {
            zl[i1] = z1;
            zout[i1] = z1;
          }
        }
      }
      else
        for(int i = tid; i < n; i += 512)
        {
          const double zi = blk_tree(Pz, ldpz, nsegz, i);
          zl[i] = zi;
          zout[i] = zi;
        }
    }
    else
      for(int i = tid; i < n; i += 512) zl[i] = z[i];
    __syncthreads();
    const unsigned long long tb2 = ROTT();
    // z.z and n+.z: C4 trees (CANON C5, oracle orc_dot_tree): element e joins chain e mod 1024, thread t owns chains 2t and
    // 2t + 1 (ascending e), then the adjacent-first pairwise tree -- the two chains of a thread, the lanes of a wave
    // (wave_sum_tree), the eight waves.
    double pz0 = 0.0, pz1 = 0.0, pc0 = 0.0, pc1 = 0.0;
    for(int base = 0; base < n; base += 1024)
    {
      const int e0 = base + 2 * tid, e1 = e0 + 1;
      if(e0 < n)
      {
        const double zv = zl[e0];
        pz0 = FMA(zv, zv, pz0);
        pc0 = FMA(cl[pd(e0)], zv, pc0);
      }
      if(e1 < n)
      {
        const double zv = zl[e1];
        pz1 = FMA(zv, zv, pz1);
        pc1 = FMA(cl[pd(e1)], zv, pc1);
      }
    }
    const double sz = wave_sum_tree(pz0 + pz1), sc = wave_sum_tree(pc0 + pc1);
    double * red = cl + npad; // (inside the dynamic allocation: tail_lds; a static array on top of a raised dynamic limit does not fit)
    if(lane == 0)
    {
      red[wave] = sz;
      red[8 + wave] = sc;
    }
    __syncthreads();
    if(tid == 0)
    {
      double w[16];
#pragma unroll
      for(int i = 0; i < 16; ++i) w[i] = red[i];
      const double zzv = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
      const double czv = cp ? (((w[8] + w[9]) + (w[10] + w[11])) + ((w[12] + w[13]) + (w[14] + w[15]))) : zl[pb];
      mb->zz = zzv;
      mb->cz = czv;
      if(sd)
      {
        sd->zz = zzv;
        sd->cz = czv;
      }
      const unsigned long long tb3 = ROTT();
      decide();
      ROTQ(27, ROTT() - tb3);
    }
    const unsigned long long tb4 = ROTT();
    post_tag(); // (wave 0 of block 1 wrote the three scalars; thread 0 is one of its lanes)
    ROTQ(24, 1);
    ROTQ(25, tb1 - tb0);
    ROTQ(26, tb2 - tb1);
    ROTQ(28, tb4 - tb2);
    ROTQ(29, ROTT() - tb4);
    return;
  }

  const unsigned long long ta0 = ROTT();
  for(int i = tid; i < q; i += 512) rs[i] = d[i];
  if(Sinv)
  {
    // CANON C10 (more than kBlockN unknowns; oracle compute_step): r = S d1 with S = R^-1 kept beside R.  Row i is a blocked product in
    // the manner of C8 over its terms k = i .. q-1: the columns in segments of 64 by absolute index, a segment's chain ascending from
    // max(64 s, i), the segment sums combined by the tree.  The blocked substitution below is q dependent steps behind two barriers
    // per 64 columns (21.7 us of a 78 us pass at c2, q ~ 300); here a (row chunk c, segment s >= c) tile is 64 independent chains of 64
    // terms, lane = row, its 64 loads in flight at once, and the tiles go round the eight waves.  The tile sums: P[tile][lane], tiles
    // numbered row chunk by row chunk -- in LDS while they fit, in the workspace Pg beyond (q > ~1100: generic pointer).
    const int nsg = (q + 63) >> 6, ntile = nsg * (nsg + 1) / 2;
    double * ro = rs + 64 * nsg; // (the result; d padded to whole segments: the entries beyond q are read and never used)
    double * P = Pg ? Pg : ro + q;
    lds_barrier();
    {
      int c = 0, off = 0; // tile t = off(c) + (s - c), off(c) = the tiles of the row chunks above c
      for(int t = wave; t < ntile; t += 8)
      {
        while(t >= off + nsg - c)
        {
          off += nsg - c;
          ++c;
        }
        const int sg = c + (t - off);
        const int i = 64 * c + lane, ic = min(i, q - 1), k0 = 64 * sg, kmax = min(64, q - k0);
        const double * base = Sinv + ic + (size_t)k0 * ld;
        double a[64];
#pragma unroll
        for(int e = 0; e < 64; ++e) a[e] = base[(size_t)min(e, kmax - 1) * ld]; // (clamped, not predicated: every load in flight at once)
        double acc = 0.0;
        if(sg > c && kmax == 64)
        {
#pragma unroll
          for(int e = 0; e < 64; ++e) acc = FMA(a[e], rs[k0 + e], acc);
        }
        else
        {
#pragma unroll
          for(int e = 0; e < 64; ++e)
          {
            const double nx = FMA(a[e], rs[k0 + e], acc);
            acc = (k0 + e >= i && e < kmax) ? nx : acc;
          }
        }
        P[(size_t)t * 64 + lane] = acc;
      }
    }
    if(Pg) __threadfence();
    lds_barrier();
    for(int i = tid; i < q; i += 512)
    {
      const int c = i >> 6;
      const double * Pc = P + ((size_t)(c * nsg - c * (c - 1) / 2) - c) * 64 + (i & 63); // + 64 s: the sum of tile (c, s)
      ro[i] = (nsg <= 8) ? tri_tree<8>(Pc, c, nsg) : (nsg <= 16) ? tri_tree<16>(Pc, c, nsg) : (nsg <= 32) ? tri_tree<32>(Pc, c, nsg) : tri_tree<64>(Pc, c, nsg);
    }
    lds_barrier();
    if(wave != 0) return;
    const unsigned long long ta1 = ROTT();
    double tk = big;
    int kk = NONE, dummy = 0;
    for(int k = lane; k < q; k += 64)
    {
      double rv = ro[k];
      r[k] = rv;
      int sk = st[k];
      if(sk != EQUALITY && sk != FIXED && rv > 0.0)
      {
        double tt = u[k] / rv;
        if(tt < tk)
        {
          tk = tt;
          kk = k;
        }
      }
    }
    wave_argmin(tk, kk, dummy);
    if(lane == 0)
    {
      const double t1v = (kk == NONE) ? big : tk;
      mb->t1 = t1v;
      mb->l = (kk == NONE) ? 0 : kk;
      mb->uq = u[q];
      if(sd) sd->t1 = t1v;
      decide();
    }
    const unsigned long long ta2 = ROTT();
    post_tag();
    ROTQ(21, ta1 - ta0);
    ROTQ(22, ta2 - ta1);
    ROTQ(23, ROTT() - ta2);
    return;
  }
  const int nblk = (q + 63) >> 6;
  // One block of R per wave in 64 statically indexed registers, lane = row.  wave 0: the diagonal block it solves next,
  // blk[k] = R(kb + lane, kb + k), and dg = its pivot; waves 1-7: the tile they advance their row block through
  double blk[64];
  double dg = 1.0;
  auto fetch_diag = [&](int cb) {
    const int kb = cb << 6, cnt = min(64, q - kb);
    const int row = kb + ((lane < cnt) ? lane : 0);
#pragma unroll
    for(int k = 0; k < 64; ++k) blk[k] = AT(R, ld, row, kb + ((k < cnt) ? k : 0));
    dg = (lane < cnt) ? AT(R, ld, row, row) : 1.0;
  };
  if(wave == 0 && nblk > 0) fetch_diag(nblk - 1);
  lds_barrier();
  for(int cb = nblk - 1; cb >= 0; --cb)
  {
    const int kb = cb << 6, cnt = min(64, q - kb);
    // waves 1-7: the tile of the first row block they advance through this column block, requested before r exists
    const int rb0 = wave - 1;
    if(wave > 0 && rb0 < cb)
    {
      const int row = (rb0 << 6) + lane;
#pragma unroll
      for(int k = 0; k < 64; ++k) blk[k] = AT(R, ld, row, kb + ((k < cnt) ? k : 0));
    }
    if(wave == 0)
    {
      const double rk0 = (lane < cnt) ? rs[kb + lane] : 0.0;
      const double rinv = recip_refined(dg);
      // (steps as in wave_bwd, device_common.h: quotient formed by the owning lane, numerator left in its lane, all quotients
      // formed after the last step; one branch per group of eight steps, steps beyond the block masked)
      double rk = rk0;
#pragma unroll
      for(int g = 7; g >= 0; --g)
      {
        if(g * 8 < cnt)
        {
#pragma unroll
          for(int u = 7; u >= 0; --u)
          {
            const int k = g * 8 + u;
            const double v = bcast_lane(div_pre(rk, dg, rinv), k);
            const double upd = FMA(-blk[k], v, rk);
            rk = (k < cnt && lane < k) ? upd : rk;
          }
        }
      }
      const double pre = rk;
      rk = div_pre(pre, dg, rinv);
      if(!__all(div_safe_den(dg) && div_safe_num(pre)))
      {
        // an operand left the range in which the hoisted reciprocal is provably the compiler's own: redo with a / b
        rk = rk0;
#pragma unroll
        for(int k = 63; k >= 0; --k)
        {
          if(k < cnt)
          {
            const double v = bcast_lane(rk, k) / bcast_lane(dg, k);
            const double upd = FMA(-blk[k], v, rk);
            rk = (lane < k) ? upd : rk;
            rk = (lane == k) ? v : rk;
          }
        }
      }
      if(lane < cnt) rs[kb + lane] = rk;
    }
    lds_barrier();
    if(wave == 0)
    {
      if(cb > 0) fetch_diag(cb - 1);
    }
    else
    {
      for(int rb = rb0; rb < cb; rb += 7)
      {
        const int row = (rb << 6) + lane;
        if(rb != rb0)
        {
#pragma unroll
          for(int k = 0; k < 64; ++k) blk[k] = AT(R, ld, row, kb + ((k < cnt) ? k : 0));
        }
        double acc = rs[row];
#pragma unroll
        for(int k0 = 56; k0 >= 0; k0 -= 8)
        {
          double rv[8];
#pragma unroll
          for(int t = 0; t < 8; ++t) rv[t] = rs[kb + min(k0 + t, cnt - 1)];
#pragma unroll
          for(int t = 7; t >= 0; --t)
          {
            const double nx = FMA(-blk[k0 + t], rv[t], acc);
            acc = (k0 + t < cnt) ? nx : acc;
          }
        }
        rs[row] = acc;
      }
    }
    lds_barrier();
  }
  if(wave != 0) return;
  // ratio test: t1 = min u_k / r_k over r_k > 0 (status by POSITION k: GoldfarbIdnaniSolver.cpp:162)
  double tk = big;
  int kk = NONE, dummy = 0;
  for(int k = lane; k < q; k += 64)
  {
    double rv = rs[k];
    r[k] = rv;
    int sk = st[k];
    if(sk != EQUALITY && sk != FIXED && rv > 0.0)
    {
      double tt = u[k] / rv;
      if(tt < tk)
      {
        tk = tt;
        kk = k;
      }
    }
  }
  wave_argmin(tk, kk, dummy);
  if(lane == 0)
  {
    const double t1v = (kk == NONE) ? big : tk;
    mb->t1 = t1v;
    mb->l = (kk == NONE) ? 0 : kk;
    mb->uq = u[q];
    if(sd) sd->t1 = t1v;
    decide();
  }
  post_tag();
}

// K9  x += t z ; u[0:q] -= t r ; u[q] += t
__global__ void advance_kernel(int n, int q, double t, int primal, const double * __restrict__ z, double * x, const double * __restrict__ r, double * u,
                               PassSync * sync = nullptr, int post_x = 0)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(primal && i < n) x[i] = FMA(t, z[i], x[i]);
  if(i < q) u[i] = FMA(-t, r[i], u[i]);
  if(i == q) u[q] = u[q] + t;
  if(sync && post_x) flag_post(&sync->x_cnt, &sync->x_tag, gridDim.x, post_x);
}

// K10 part 1 (one workgroup of four waves): the Givens coefficients (c_i, s_i), i = n-2 .. q, of an addition and the final radius
// from d -- CANON C7 (oracle orc_givens_chain; device_common.h).  The squares of d are scanned from the END of d in blocks of 64 (a
// wave per block, lane 63 = the block's highest index: six shuffle-and-add steps), wave 0 chains the blocks (one addition each),
// then every thread takes a square root and two divisions per rotation.  Rounds 1-3 ran the radius recurrence r_i = hypot(d_i,
// r_{i+1}) here -- n - q dependent steps, 36 us at n = 1024 with seeds and a systolic wave: the longest kernel of a pass, and on its
// critical path.  Entries that cannot be squared (beyond 2^+-500, not finite) take the serial recurrence on wave 0, as the oracle does.
// Reads d only and writes only its own outputs, so it can run ahead of the add/drop decision beside z / r / the ratio test.
constexpr int CHAIN_THREADS = 256;
__global__ __launch_bounds__(CHAIN_THREADS) void add_chain_kernel(int n, int q, const double * d, double * cc, double * ss, double * rfin, int unused,
                                                                  PassSync * sync = nullptr, int tag = 0, Mailbox * mb = nullptr)
{
  extern __shared__ double sm[]; // d (n) | S, then rho (n + 1) | block carries (64) | flag
  if(sync && tag) flag_wait(&sync->d_tag, tag, mb); // launched beside jtn on another stream: d of THIS pass
  double * dsm = sm;
  double * S = sm + n;
  double * carry = S + (n + 1);
  int * okf = reinterpret_cast<int *>(carry + 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if(tid == 0) *okf = 1;
  for(int i = tid; i < n; i += CHAIN_THREADS) dsm[i] = d[i];
  __syncthreads();
  if(n - 1 <= q) // a single entry: no rotation, the radius is the entry itself
  {
    if(tid == 0) *rfin = dsm[n - 1];
    if(sync && tag) flag_post(nullptr, &sync->chain_tag, 1, tag);
    return;
  }
  bool ok = true;
  for(int k = q + tid; k < n; k += CHAIN_THREADS) ok = ok && c7_in_range(dsm[k]);
  if(!__all(ok) && lane == 0) *okf = 0;
  __syncthreads();
  if(*okf)
  {
    const int nblk = (n - q + 63) >> 6;
    for(int b = wave; b < nblk; b += CHAIN_THREADS / 64)
    {
      const int k = (n - 1 - 64 * b) - 63 + lane;
      const bool in = k >= q;
      const double dk = dsm[in ? k : q];
      const double x = c7_block_scan(in ? dk * dk : 0.0, lane, 63);
      if(in) S[k] = x;
      if(lane == 0) carry[b] = x; // the block's total (only read for a full block)
    }
    __syncthreads();
    if(wave == 0)
    {
      // carry_0 = 0, carry_b+1 = S_(64 b + 63) = x_(64 b + 63) + carry_b: lane b ends up with carry_b
      const double tb = (lane < nblk) ? carry[lane] : 0.0;
      double cy = 0.0, mine = 0.0;
      for(int b = 0; b < nblk; ++b)
      {
        mine = (lane == b) ? cy : mine;
        cy = bcast_lane(tb, b) + cy;
      }
      if(lane < nblk) carry[lane] = mine;
    }
    __syncthreads();
    for(int k = q + tid; k < n; k += CHAIN_THREADS)
    {
      const int b = (n - 1 - k) >> 6;
      S[k] = (k == n - 1) ? dsm[k] : __builtin_sqrt(S[k] + carry[b]); // rho
    }
    __syncthreads();
    for(int i = q + tid; i <= n - 2; i += CHAIN_THREADS)
    {
      const double ri = S[i], rup = S[i + 1];
      const bool z = (ri == 0.0);
      const double cq = dsm[i] / ri, sq = (-rup) / ri;
      cc[i] = z ? 1.0 : cq;
      ss[i] = z ? 0.0 : sq;
    }
    if(tid == 0) *rfin = S[q];
  }
  else if(wave == 0)
  {
    // the serial recurrence.  The loop body must stay free of stores and exec-mask regions (they stall the in-order wave: measured
    // +40 % on the batched kernel's chain).  The d values of the next 64 steps sit in a register (lane = step inside the block,
    // read with v_readlane); lane (i mod 64) keeps r_{i+1}; both are exchanged with LDS once per 64 steps.
    double * rho = S;
    double r = opaque_v(dsm[n - 1]);
    double kq = 0.0;
    for(int hi = n - 2; hi >= q; hi -= 64)
    {
      const int lo = (hi - 63 > q) ? (hi - 63) : q; // this block: i = hi .. lo
      const int mine = hi - lane;                   // lane j handles step i = hi - j
      const double dblk = (mine >= lo) ? dsm[mine] : 0.0;
      for(int i = hi; i >= lo; --i)
      {
        const double di = opaque_v(bcast_lane(dblk, hi - i));
        kq = (lane == hi - i) ? r : kq; // r_{i+1}
        r = givens_r_only(di, r);
      }
      if(mine >= lo) rho[mine + 1] = kq;
    }
    wave_sync();
    for(int i = q + lane; i <= n - 2; i += 64)
    {
      double c, s, rn;
      make_givens(dsm[i], rho[i + 1], c, s, rn);
      cc[i] = c;
      ss[i] = s;
    }
    if(lane == 0) *rfin = r;
  }
  if(sync && tag) flag_post(nullptr, &sync->chain_tag, 1, tag);
}

// K10 part 1b: R(:, q) = d(0:q) with the new diagonal; status byte of the activated constraint (GoldfarbIdnaniSolver.cpp:233)
__global__ __launch_bounds__(256) void add_commit_kernel(int q, const double * d, const double * rfin, double * R, int ld, unsigned char * st, int p,
                                                         int pstat, const PassSync * sync = nullptr, int wait_chain = 0, Mailbox * mb = nullptr,
                                                         double * S = nullptr, const double * r = nullptr)
{
  if(sync && wait_chain) flag_wait(&sync->chain_tag, wait_chain, mb); // rfin comes from the chain on the side stream
  const int k = blockIdx.x * 256 + threadIdx.x;
  if(k < q) AT(R, ld, k, q) = d[k];
  if(S && k < q) AT(S, ld, k, q) = -(r[k] / *rfin); // CANON C10
  if(k == q)
  {
    AT(R, ld, q, q) = *rfin;
    if(S) AT(S, ld, q, q) = 1.0 / *rfin;
    st[p] = (unsigned char)pstat;
  }
}

// K10 part 2 / K11 part 2: apply the rotation sequence to the columns of J, one workgroup of FOUR waves per 64 rows.
//   descending (add):   i = i_hi .. i_lo, columns (i, i+1), carry = column i+1
//   ascending (remove): i = i_lo .. i_hi, columns (i, i+1), carry = column i
// A row's sweep is one carry chain (two dependent ops per rotation); the point of this kernel is that the chain never
// waits for memory.  gfx9 counts loads AND stores in one in-order counter (vmcnt), and a wave that keeps three chunks of
// columns in flight in registers has none left for the (c, s) pairs -- hipcc then parks a chunk in AGPRs at the top of the
// steady loop behind an s_waitcnt vmcnt(0), i.e. every load in flight is waited for once per pass (round 2: 36 us per sweep of
// 870 rotations, of which ~10 us are the chain).  So each direction of the stream, and the chain, get a wave of their own:
//   wave 2 (loader)  streams the columns into LDS with LDS-direct loads (global_load_lds_dwordx4: no registers, 1 KiB = two
//                    columns of 64 rows per instruction), up to three chunks of 32 columns in flight in a ring of four slots;
//                    it alone waits for vmcnt, with the constant that lets the NEXT chunk land;
//   wave 0 (chain)   reads column values and (c, s) pairs out of LDS eight columns ahead of the rotations, rotates, and leaves the
//                    rotated columns in one of two LDS buffers -- it never has a memory operation outstanding in the loop;
//   waves 1, 3 (storers) copy the buffers of the finished chunks to memory, alternately, while wave 0 rotates the next ones.
// One LDS-only barrier per chunk for all four; (c_i, s_i) of the whole sweep are staged in LDS once.
constexpr int RC = 32;          // columns per chunk
constexpr int RSLOTS = 4;       // ring of chunk slots the loader fills (64 rows x RC columns each, column after column)
constexpr int RSLOT = RC * 64;  // doubles per slot
constexpr int RCB = 8;          // columns whose operands the chain wave requests per batch
constexpr int ROT_THREADS = 256;

template<bool DESC, bool FULL>
__device__ __forceinline__ void rot_chunk(const double * xs, double & carry, const double2 * csl, int t0, int T, double * obuf, int lane)
{
  // rotation number t = t0 + u; its (c, s) pair sits at index (DESC ? T-1-t : t) of the staged array: for a full chunk
  // that is one base address and immediate offsets
  const double2 * cb = csl + (DESC ? (T - 1 - t0) : t0);
  auto csat = [&](int u) -> double2 {
    if(FULL) return DESC ? cb[-u] : cb[u];
    const int t = min(t0 + u, T - 1);
    return csl[DESC ? (T - 1 - t) : t];
  };
  // operands are requested RCB columns at a time, one batch ahead of the rotations that use them (scheduling barriers: left
  // alone, hipcc requests each pair two rotations ahead and the chain waits for LDS at every step)
  double2 csb[2][RCB];
  double xb[2][RCB];
#pragma unroll
  for(int u = 0; u < RCB; ++u)
  {
    csb[0][u] = csat(u);
    xb[0][u] = xs[u * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for(int u = 0; u < RC; ++u)
  {
    if(u % RCB == 0 && u + RCB < RC)
    {
#pragma unroll
      for(int w = 0; w < RCB; ++w)
      {
        csb[((u / RCB) + 1) & 1][w] = csat(u + RCB + w);
        xb[((u / RCB) + 1) & 1][w] = xs[(u + RCB + w) * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const double c = csb[(u / RCB) & 1][u % RCB].x, sn = csb[(u / RCB) & 1][u % RCB].y, xv = xb[(u / RCB) & 1][u % RCB];
    double out, nc;
    if(DESC)
    {
      out = FMA(sn, xv, c * carry);
      nc = FMA(-sn, carry, c * xv);
    }
    else
    {
      out = FMA(-sn, xv, c * carry);
      nc = FMA(sn, carry, c * xv);
    }
    obuf[u * 64 + lane] = out;
    carry = (FULL || t0 + u < T) ? nc : carry;
  }
}

template<bool DESC>
__device__ __forceinline__ void rot_sweep(double * J, int ld, int n, int row0, int i_lo, int i_hi, const double2 * csl, double * slots, double * obufs)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = i_hi - i_lo + 1;
  const int nchunk = (T + RC - 1) / RC, nfull = T / RC;
  const int row = min(row0 + lane, n - 1);
  // rotation t consumes column fcol(t) and produces column wcol(t); consecutive rotations are one column apart
  auto fcol = [&](int t) { return DESC ? i_hi - t : i_lo + t + 1; };
  auto wcol = [&](int t) { return DESC ? i_hi - t + 1 : i_lo + t; };
  const ptrdiff_t cstep = DESC ? -(ptrdiff_t)ld : (ptrdiff_t)ld;
  // Every wave passes the same nchunk + 1 barriers B_0 .. B_nchunk.  Chunk k is in its slot before B_k (loader), rotated between
  // B_k and B_k+1 (chain), stored after B_k+1 (storer); slot (k + 3) mod 4 -- last read for chunk k - 1 -- is refilled after B_k.
  if(wave == 2)
  {
    // lane l fetches rows 2 (l mod 32), + 1 of column 2 i + (l div 32) of the chunk: 16 bytes; instruction i lands as 1 KiB at
    // slot + 2 i columns.  Rows past n are fetched from the block's first row pair instead (never used); the last pair of an odd
    // n reads one double past its column (the next column's first row; past the LAST column that is the slack J is allocated with).
    const int pr = 2 * (lane & 31), half = lane >> 5;
    const int lrow = (row0 + pr < n) ? row0 + pr : row0;
    auto issue = [&](int k) {
      double * slot = slots + (k % RSLOTS) * RSLOT;
      if(k < nfull)
      {
        const double * p = &AT(J, ld, lrow, fcol(k * RC + half));
#pragma unroll
        for(int i = 0; i < RC / 2; ++i)
        {
          __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)(slot + i * 128), 16, 0, 0);
          p += 2 * cstep;
        }
      }
      else
      {
#pragma unroll
        for(int i = 0; i < RC / 2; ++i)
        {
          const double * p = &AT(J, ld, lrow, fcol(min(k * RC + 2 * i + half, T - 1)));
          __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)(slot + i * 128), 16, 0, 0);
        }
      }
    };
    for(int k = 0; k < 3 && k < nchunk; ++k) issue(k);
    for(int k = 0; k < nchunk; ++k)
    {
      // chunk k has landed when at most the loads of the chunks behind it are outstanding (16 per chunk, in order)
      const unsigned long long tw = ROTT();
      if(k + 2 < nchunk)
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if(k + 1 < nchunk)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long tx = ROTT();
      ROTP(4, tx - tw); // loader: waiting for a chunk to land
      lds_barrier(); // B_k
      ROTP(5, ROTT() - tx); // loader: waiting at the barrier
      if(k + 3 < nchunk) issue(k + 3);
    }
    lds_barrier(); // B_nchunk
  }
  else if(wave == 0)
  {
    double carry = DESC ? AT(J, ld, row, i_hi + 1) : AT(J, ld, row, i_lo);
    for(int k = 0; k < nchunk; ++k)
    {
      const unsigned long long tb = ROTT();
      lds_barrier(); // B_k: chunk k is in its slot, the storer is done with the buffer of chunk k - 2
      ROTP(k == 0 ? 6 : 7, ROTT() - tb); // chain: waiting at B_0 (the first chunk) / at the other barriers
      ROTP(8, 1);
      const double * xs = slots + (k % RSLOTS) * RSLOT;
      double * ob = obufs + (k & 1) * RSLOT;
      if(k < nfull)
        rot_chunk<DESC, true>(xs, carry, csl, k * RC, T, ob, lane);
      else
        rot_chunk<DESC, false>(xs, carry, csl, k * RC, T, ob, lane);
    }
    lds_barrier(); // B_nchunk
    if(row0 + lane < n) AT(J, ld, row, DESC ? i_lo : i_hi + 1) = carry;
  }
  else
  {
    // Two storer waves (1 and 3), alternate chunks.  A wave has at most 63 memory operations outstanding (vmcnt, stores included on
    // gfx9), and a write acknowledgement takes a microsecond or two: one wave storing a column (512 bytes) per instruction moved one
    // chunk per microsecond and the whole workgroup waited for it at every barrier.  Here an instruction stores two columns (lane =
    // (column, row pair): 16 bytes), the two waves share the chunks, and a wave fetches a chunk out of its buffer right behind
    // the barrier that completes it but issues the stores only behind the NEXT barrier -- a stalled store holds nobody up.
    const int sidx = (wave == 1) ? 0 : 1;
    const int pr = 2 * (lane & 31), half = lane >> 5;
    const bool two = row0 + pr + 1 < n, one = row0 + pr < n;
    const int srow = min(row0 + pr, n - 1);
    double2 v[RC / 2];
    auto fetch = [&](int k) {
      const double * ib = obufs + (k & 1) * RSLOT + half * 64 + pr;
#pragma unroll
      for(int i = 0; i < RC / 2; ++i) v[i] = *reinterpret_cast<const double2 *>(ib + i * 128);
    };
    auto store = [&](int k) {
      double * p = &AT(J, ld, srow, wcol(k * RC + half));
#pragma unroll
      for(int i = 0; i < RC / 2; ++i)
      {
        if(k < nfull || k * RC + 2 * i + half < T)
        {
          if(two)
            *reinterpret_cast<double2 *>(p) = v[i];
          else if(one)
            *p = v[i].x;
        }
        p += 2 * cstep;
      }
    };
    lds_barrier(); // B_0
    int held = -1;
    for(int k = 0; k < nchunk; ++k)
    {
      const unsigned long long tb = ROTT();
      lds_barrier(); // B_k+1: chunk k is in its buffer
      if(sidx == 0) ROTP(9, ROTT() - tb); // storer A: waiting at the barriers
      if((k & 1) == sidx)
      {
        fetch(k);
        held = k;
      }
      else if(held >= 0)
      {
        store(held);
        held = -1;
      }
    }
    if(held >= 0) store(held);
  }
}

// The sweep of an ADDITION also commits it (K10 part 1b, add_commit_kernel's work): workgroup 0 writes R(:, q) = d(0:q)
// with the new diagonal and the status byte of the activated constraint -- one dispatch less on the critical path.
struct CommitArgs
{
  const double *d, *rfin;
  double * R;
  unsigned char * st;
  int q, ld, p, pstat;
  const PassSync * sync; // wait_chain != 0: (c_i, s_i) and rfin come from the chain on the side stream (PassSync::chain_tag)
  int wait_chain;
  Mailbox * mb;
  double * S = nullptr;       // CANON C10: S = R^-1 gets the column [-r / rho; 1 / rho] (r: this pass's S d1)
  const double * r = nullptr;
};
// The sweep enqueued AHEAD of the add / drop decision (StepDev, gi_dense.h): it runs only if the device's decision for pass `tag` is
// "add", takes (p, status) from psel and t from StepDev, and does the primal / dual update itself -- workgroup g advances its 64
// entries of x, workgroup 0 also u (DualSolver.cpp:146-150) -- before it commits and sweeps.
struct SpecArgs
{
  const StepDev * sd; // nullptr: not speculative (the host has decided; `advance` ran as its own kernel)
  int tag;
  const int * psel;
  const double *z, *r;
  double *x, *u;
  PassSync * xsync;
  int post_x;
};

__global__ __launch_bounds__(ROT_THREADS) void rot_apply_kernel(double * J, int ld, int n, int i_lo, int i_hi, int descending, const double * __restrict__ cc,
                                                                const double * __restrict__ ss, CommitArgs cm = CommitArgs{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, 0, nullptr},
                                                                SpecArgs sp = SpecArgs{nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0})
{
  extern __shared__ double rsm[]; // four chunk slots | two chunk buffers | (c_i, s_i) pairs (T)
  const unsigned long long tk0 = ROTT();
  if(sp.sd)
  {
    // same stream as the step kernel that took the decision: it is complete (and its stores visible) when this kernel starts
    if(sp.sd->tag != sp.tag || sp.sd->kind != kKindAdd) return; // this pass does not add: nothing was touched
    cm.p = sp.psel[0];
    cm.pstat = sp.psel[1];
    const double t = sp.sd->t;
    const int i = blockIdx.x * 64 + (int)threadIdx.x;
    if(threadIdx.x < 64 && i < n) sp.x[i] = FMA(t, sp.z[i], sp.x[i]);
    if(blockIdx.x == 0)
    {
      for(int k = threadIdx.x; k < cm.q; k += ROT_THREADS) sp.u[k] = FMA(-t, sp.r[k], sp.u[k]);
      if(threadIdx.x == 0) sp.u[cm.q] = sp.u[cm.q] + t;
    }
    if(sp.xsync && sp.post_x) flag_post(&sp.xsync->x_cnt, &sp.xsync->x_tag, gridDim.x, sp.post_x);
  }
  if(cm.sync && cm.wait_chain) flag_wait(&cm.sync->chain_tag, cm.wait_chain, cm.mb);
  const unsigned long long tk1 = ROTT();
  if(cm.d && blockIdx.x == 0)
  {
    for(int k = threadIdx.x; k < cm.q; k += ROT_THREADS) AT(cm.R, cm.ld, k, cm.q) = cm.d[k];
    if(cm.S)
    {
      const double rho = *cm.rfin;
      for(int k = threadIdx.x; k < cm.q; k += ROT_THREADS) AT(cm.S, cm.ld, k, cm.q) = -(cm.r[k] / rho);
      if(threadIdx.x == 0) AT(cm.S, cm.ld, cm.q, cm.q) = 1.0 / rho;
    }
    if(threadIdx.x == 0)
    {
      AT(cm.R, cm.ld, cm.q, cm.q) = *cm.rfin;
      cm.st[cm.p] = (unsigned char)cm.pstat;
    }
  }
  if(i_hi < i_lo) return;
  const int T = i_hi - i_lo + 1;
  double * slots = rsm;
  double * obufs = rsm + RSLOTS * RSLOT;
  double2 * csl = reinterpret_cast<double2 *>(obufs + 2 * RSLOT);
  for(int k = threadIdx.x; k < T; k += ROT_THREADS) csl[k] = make_double2(cc[i_lo + k], ss[i_lo + k]);
  __syncthreads();
  const unsigned long long tk2 = ROTT();
  if(descending)
    rot_sweep<true>(J, ld, n, blockIdx.x * 64, i_lo, i_hi, csl, slots, obufs);
  else
    rot_sweep<false>(J, ld, n, blockIdx.x * 64, i_lo, i_hi, csl, slots, obufs);
  if(threadIdx.x == 0)
  {
    ROTP(0, tk1 - tk0);    // waiting for the chain's tag
    ROTP(1, tk2 - tk1);    // commit + staging the (c, s) pairs
    ROTP(2, ROTT() - tk2); // the sweep, as the chain wave sees it
    ROTP(3, 1);            // launches
  }
}

// CANON C10, removal: S' = (S G) without row l and without its last column, G the rotations remove_r_kernel has just generated
// (cc[i], ss[i], i = l .. q-2, on columns (i, i + 1): the rotation J takes too -- (G^T R)^-1 = S G).  Every row is a chain of its own:
// lane = NEW row r' (old row r = r' + (r' >= l)), carried value x = S(r, i), next y = S(r, i + 1); a row below l meets its first
// rotation, i = r - 1, with a structural zero for x.  The new S is written into the OTHER buffer (a row below l moves up by one: in
// place it would overwrite what the lane of the row above still reads); the host swaps the two buffers afterwards.
// grid (row groups of 64, 2): y = 0 rotates -- the (c, s) pairs staged in LDS, 64 entries of the row requested together (a chain of q - l
// steps with one memory round trip per 64) --, y = 1 copies the columns left of l, which no rotation touches.
__global__ __launch_bounds__(64) void remove_s_kernel(const double * __restrict__ S, double * __restrict__ S2, int ld, int q, int l,
                                                      const double * __restrict__ cc, const double * __restrict__ ss)
{
  extern __shared__ double2 cs_l[]; // (c, s) of rotation l + t
  const int lane = threadIdx.x;
  const int r2 = blockIdx.x * 64 + lane; // new row
  if(blockIdx.y == 1)
  {
    // rows above l: columns r .. l - 1 as they are
    if((int)blockIdx.x * 64 >= l) return;
    const int r = min(r2, l - 1);
    for(int j0 = blockIdx.x * 64; j0 < l; j0 += 32)
    {
      double v[32];
#pragma unroll
      for(int e = 0; e < 32; ++e) v[e] = AT(S, ld, r, min(j0 + e, l - 1));
#pragma unroll
      for(int e = 0; e < 32; ++e)
        if(r2 < l && j0 + e >= r2 && j0 + e < l) AT(S2, ld, r2, j0 + e) = v[e];
    }
    return;
  }
  const int nrot = q - l - 1;
  for(int t = lane; t < nrot; t += 64) cs_l[t] = make_double2(cc[l + t], ss[l + t]);
  wave_sync();
  if(nrot <= 0) return;
  const bool live = r2 < q - 1;
  const int r2c = min(r2, q - 2);
  const int r = r2c + (r2c >= l ? 1 : 0); // old row
  const int t0 = (r < l) ? 0 : r - l - 1;
  double carry = (r < l) ? AT(S, ld, r, l) : 0.0;
  const double * yrow = S + r + (size_t)(l + 1) * ld; // y of rotation t: yrow[t * ld]
  double * orow = S2 + r2c + (size_t)l * ld;          // its output: orow[t * ld]
  for(int tb = __builtin_amdgcn_readfirstlane(t0); tb < nrot; tb += 64) // (t0 grows with the lane: lane 0 starts first)
  {
    double y[64];
#pragma unroll
    for(int e = 0; e < 64; ++e) y[e] = yrow[(size_t)min(tb + e, nrot - 1) * ld];
#pragma unroll
    for(int e = 0; e < 64; ++e)
    {
      const int tt = tb + e;
      const double2 cs = cs_l[min(tt, nrot - 1)];
      const double out = FMA(-cs.y, y[e], cs.x * carry);
      const double nc = FMA(cs.y, carry, cs.x * y[e]);
      const bool on = live && tt >= t0 && tt < nrot;
      if(on) orow[(size_t)tt * ld] = out;
      carry = on ? nc : carry;
    }
  }
}

// CANON C10, warm start: S = R^-1 of the QR factor (oracle gi_solve_impl).  One workgroup per column j solves R s = e_j by column-oriented
// back substitution in LDS: s_j = 1 / R(j,j); once s_k is final every row above adds its term, acc_i = fma(R(i,k), s_k, acc_i) (row i's
// chain runs over k = j .. i + 1, descending, from +0 -- the oracle's order), and s_{k-1} = -acc_{k-1} / R(k-1,k-1).  Column k of R is
// read coalesced; one barrier per step, the q columns side by side on q workgroups.
__global__ __launch_bounds__(256) void invert_upper_kernel(const double * __restrict__ R, double * __restrict__ S, int ld, int q)
{
  extern __shared__ double col[]; // acc_i until row i is final, s_i afterwards
  const int j = blockIdx.x, tid = threadIdx.x;
  if(j >= q) return;
  for(int i = tid; i < j; i += 256) col[i] = 0.0;
  if(tid == 0) col[j] = 1.0 / AT(R, ld, j, j);
  __syncthreads();
  for(int k = j; k >= 1; --k)
  {
    const double sk = col[k];
    for(int i = tid; i < k; i += 256)
    {
      double a = FMA(AT(R, ld, i, k), sk, col[i]);
      if(i == k - 1) a = (-a) / AT(R, ld, k - 1, k - 1);
      col[i] = a;
    }
    __syncthreads();
  }
  for(int i = tid; i <= j; i += 256) AT(S, ld, i, j) = col[i];
}

// K11 part 1 (one workgroup; thread tid owns the NEW columns tid + u * blockDim.x, u < NC, of R): delete column l of R,
// restore triangularity with q' - l rotations (q' = q - 1), record (c_i, s_i); shift u; clear the status byte.
// NC = 1 up to 1024 active constraints (one column per thread), NC = 4 up to 4096.
template<int NC>
__global__ __launch_bounds__(1024) void remove_r_kernel(double * R, int ld, int q, int l, double * cc, double * ss, double * u, unsigned char * st, int gone)
{
  __shared__ double gcs[2][2]; // (c, s) of the rotation in flight, two slots taking turns: ONE barrier per rotation -- rotation i + 1
                               // writes the other slot, and nobody can be two rotations ahead of a thread that has not read yet
  const int tid = threadIdx.x;
  const int nt = blockDim.x;
  const int qn = q - 1;
  // u.segment(l, q-l) = u.tail(q-l)   (u has q+1 entries)
  double uv[NC];
#pragma unroll
  for(int w = 0; w < NC; ++w)
  {
    const int k = tid + w * nt;
    uv[w] = (k >= l && k < q) ? u[k + 1] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for(int w = 0; w < NC; ++w)
  {
    const int k = tid + w * nt;
    if(k >= l && k < q) u[k] = uv[w];
  }
  if(tid == 0) st[gone] = INACTIVE;
  // head rows 0..l-1 of new column k come from old column k+1 (no rotation touches them): thread = row, columns
  // shifted left one after the other (a row is private to its thread: no barrier needed)
  double carry[NC];
#pragma unroll
  for(int w = 0; w < NC; ++w)
  {
    const int k = tid + w * nt;
    carry[w] = (k >= l && k < qn) ? AT(R, ld, l, k + 1) : 0.0;
  }
  for(int rr = tid; rr < l; rr += nt)
  {
    // (16 entries of the row requested together, then stored one column to the left: one memory round trip per 16
    // columns instead of one per column; across threads every access is coalesced)
    int k = l;
    for(; k + 24 <= qn; k += 24)
    {
      double v[24];
#pragma unroll
      for(int e = 0; e < 24; ++e) v[e] = AT(R, ld, rr, k + 1 + e);
#pragma unroll
      for(int e = 0; e < 24; ++e) AT(R, ld, rr, k + e) = v[e];
    }
    for(; k + 8 <= qn; k += 8)
    {
      double v[8];
#pragma unroll
      for(int e = 0; e < 8; ++e) v[e] = AT(R, ld, rr, k + 1 + e);
#pragma unroll
      for(int e = 0; e < 8; ++e) AT(R, ld, rr, k + e) = v[e];
    }
    for(; k < qn; ++k) AT(R, ld, rr, k) = AT(R, ld, rr, k + 1);
  }
  __syncthreads();
  // Rotation i needs row i+1 of the old R (one entry per column, a cache line each): the rows of the next PF
  // rotations are requested ahead -- clamped addresses, no predicate -- and the per-rotation hand-off of (c, s) goes
  // through LDS behind LDS-only barriers (a __syncthreads() would drain the prefetches at every step).  A thread only
  // ever rewrites entries of its own columns that it has already consumed.
  // (round 5: 12 rows ahead with one column per thread -- a rotation's store is in the same in-order queue as the prefetches, the row
  // requested four rotations ago was only there once the stores of the rotations between had been acknowledged: ~0.4 us a rotation)
  constexpr int PF = (NC == 1) ? 12 : 4;
  int colr[NC]; // old column read for slot w (clamped for the slots without a column)
  double yq[NC][PF];
#pragma unroll
  for(int w = 0; w < NC; ++w)
  {
    colr[w] = min(tid + w * nt + 1, q - 1);
#pragma unroll
    for(int e = 0; e < PF; ++e) yq[w][e] = AT(R, ld, min(l + 1 + e, q - 1), colr[w]);
  }
  for(int i0 = l; i0 < qn; i0 += PF)
  {
#pragma unroll
    for(int e = 0; e < PF; ++e)
    {
      const int i = i0 + e;
      if(i < qn) // (uniform)
      {
        double y[NC];
#pragma unroll
        for(int w = 0; w < NC; ++w)
        {
          const int k = tid + w * nt;
          const double yl = yq[w][e];
          yq[w][e] = AT(R, ld, min(i + 1 + PF, q - 1), colr[w]);
          y[w] = (k >= i && k < qn) ? yl : 0.0;
          if(k == i)
          {
            double c, s, rho;
            make_givens_lean(carry[w], y[w], c, s, rho);
            gcs[i & 1][0] = c;
            gcs[i & 1][1] = s;
            cc[i] = c;
            ss[i] = s;
            AT(R, ld, i, i) = rho;
          }
        }
        lds_barrier();
        const double c = gcs[i & 1][0], s = gcs[i & 1][1];
#pragma unroll
        for(int w = 0; w < NC; ++w)
        {
          const int k = tid + w * nt;
          if(k > i && k < qn)
          {
            AT(R, ld, i, k) = FMA(-s, y[w], c * carry[w]);
            carry[w] = FMA(s, carry[w], c * y[w]);
          }
        }
      }
    }
  }
}


// =====================================================================================================================
// Warm start of Mode A (experimental::GoldfarbIdnaniSolver, src/experimental/GoldfarbIdnaniSolver.cpp:388-487; oracle
// initialize_computation_data / initialize_primal_dual).  The n x q matrix N -> B = L^-1 N -> its Householder QR factors lives in
// the solver's R buffer (ld n), exactly where the reference keeps it: afterwards R IS the upper triangle of that array and
// the Householder vectors below the diagonal are never read again.  Coverage path: every element keeps the oracle's chain, the
// kernels are simple (one launch per reflector), the cost is a few milliseconds once per solve.
// =====================================================================================================================

// N (+-C columns / +-unit vectors) and b_act in activation order (:393-418); block j = column j
__global__ __launch_bounds__(256) void warm_build_kernel(const double * __restrict__ C, int ldc, int n, int m, const int * __restrict__ act,
                                                         const unsigned char * __restrict__ st, const double * __restrict__ bl,
                                                         const double * __restrict__ bu, const double * __restrict__ xl,
                                                         const double * __restrict__ xu, double * N, int ldn, double * bact)
{
  const int j = blockIdx.x, idx = act[j], sj = st[idx];
  double * col = N + (size_t)j * ldn;
  if(idx < m)
  {
    const double * c = C + (size_t)idx * ldc;
    for(int i = threadIdx.x; i < n; i += 256) col[i] = (sj == UPPER) ? -c[i] : c[i];
    if(threadIdx.x == 0) bact[j] = (sj == UPPER) ? -bu[idx] : bl[idx];
  }
  else
  {
    const int v = idx - m;
    for(int i = threadIdx.x; i < n; i += 256) col[i] = (i == v) ? ((sj == UPPER_BOUND) ? -1.0 : 1.0) : 0.0;
    if(threadIdx.x == 0) bact[j] = (sj == UPPER_BOUND) ? -xu[v] : xl[v];
  }
}

// B = L^-1 N (:426), one workgroup per column: the blocked forward substitution of x0_kernel (same per-element order)
__global__ __launch_bounds__(512) void warm_fwd_kernel(const double * __restrict__ L, int ldl, int n, double * B, int ldb)
{
  extern __shared__ double xs[];
  double * col = B + (size_t)blockIdx.x * ldb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for(int i = tid; i < n; i += 512) xs[i] = col[i];
  blocked_subst<false>(L, ldl, n, xs, lane, wave); // (starts and ends with a barrier)
  for(int i = tid; i < n; i += 512) col[i] = xs[i];
}

// in-order chain sum_{i = lo}^{hi-1} a[i] * b[i] over LDS operands, every lane of the calling wave the same value
__device__ __forceinline__ double lds_chain(const double * a, const double * b, int lo, int hi)
{
  double acc = 0.0;
  int k = lo;
  for(; k + 64 <= hi; k += 64) acc = chain64<1, true>(acc, a + k, b + k, 64);
  if(k < hi) acc = chain64<1, false>(acc, a + k, b + k, hi - k);
  return acc;
}

// Householder vector of column k (:432-440; Eigen makeHouseholder; oracle): tail = sum_{i > k} v_i^2 as ONE in-order chain,
// beta = -sign(c0) sqrt(c0^2 + tail), v_i /= (c0 - beta), tau = (beta - c0) / beta; v_k <- beta, hco[k] <- tau
__global__ __launch_bounds__(1024) void warm_hh_make_kernel(double * B, int ldb, int n, int k, double * hco)
{
  extern __shared__ double vs[]; // n
  __shared__ double sh[4];
  double * vk = B + (size_t)k * ldb;
  const int tid = threadIdx.x;
  for(int i = tid; i < n; i += 1024) vs[i] = vk[i];
  __syncthreads();
  if(tid < 64)
  {
    const double tailsq = lds_chain(vs, vs, k + 1, n);
    const double c0 = vs[k];
    const bool zero = tailsq <= 2.2250738585072014e-308;
    double beta = __builtin_sqrt(FMA(c0, c0, tailsq));
    beta = (c0 >= 0.0) ? -beta : beta;
    const double den = c0 - beta;
    double tau = (beta - c0) / beta;
    if(tid == 0)
    {
      sh[0] = zero ? 0.0 : tau;
      sh[1] = zero ? c0 : beta;
      sh[2] = den;
      sh[3] = zero ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  const double den = sh[2];
  const bool zero = sh[3] != 0.0;
  for(int i = k + 1 + tid; i < n; i += 1024) vk[i] = zero ? 0.0 : vs[i] / den;
  if(tid == 0)
  {
    vk[k] = sh[1];
    hco[k] = sh[0];
  }
}

// applyHouseholderOnTheLeft of reflector k on trailing column j = k + 1 + blockIdx.x (one wave per column): t = sum_{i > k} v_i c_i
// (one in-order chain), t += c_k, c_k = fma(-tau, t, c_k), c_i = fma(-(tau t), v_i, c_i)
__global__ __launch_bounds__(64) void warm_hh_apply_kernel(double * B, int ldb, int n, int k, const double * __restrict__ hco)
{
  extern __shared__ double sm[]; // v (n) | c (n)
  const double tau = hco[k];
  if(tau == 0.0 || k + 1 >= n) return;
  const int lane = threadIdx.x;
  const double * vk = B + (size_t)k * ldb;
  double * cj = B + (size_t)(k + 1 + blockIdx.x) * ldb;
  double * vs = sm;
  double * cs = sm + n;
  for(int i = k + lane; i < n; i += 64)
  {
    vs[i] = vk[i];
    cs[i] = cj[i];
  }
  wave_sync();
  double t = lds_chain(vs, cs, k + 1, n);
  const double top = cs[k];
  t = t + top;
  const double stt = tau * t;
  if(lane == 0) cj[k] = FMA(-tau, t, top);
  for(int i = k + 1 + lane; i < n; i += 64) cj[i] = FMA(-stt, vs[i], cs[i]);
}

// J = J Q, Q = H_0 H_1 ... H_{q-1} (:443-444): thread = row of J, reflector after reflector (a row only meets itself); the row's
// entries come in coalesced across the threads, sixteen in flight per thread
__global__ __launch_bounds__(64) void warm_jq_kernel(double * J, int ldj, int n, const double * __restrict__ B, int ldb, int q,
                                                     const double * __restrict__ hco)
{
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int ri = (i < n) ? i : (n - 1);
  for(int k = 0; k < q; ++k)
  {
    const double tau = hco[k];
    if(tau == 0.0 || k + 1 >= n) continue;
    const double * v = B + (size_t)k * ldb;
    double t = 0.0;
    int j = k + 1;
    for(; j + 16 <= n; j += 16)
    {
      double jv[16], vv[16];
#pragma unroll
      for(int u = 0; u < 16; ++u)
      {
        jv[u] = J[ri + (size_t)(j + u) * ldj];
        vv[u] = v[j + u];
      }
#pragma unroll
      for(int u = 0; u < 16; ++u) t = FMA(jv[u], vv[u], t);
    }
    for(; j < n; ++j) t = FMA(J[ri + (size_t)j * ldj], v[j], t);
    const double jk = J[ri + (size_t)k * ldj];
    t = t + jk;
    const double stt = tau * t;
    if(i < n)
    {
      J[ri + (size_t)k * ldj] = FMA(-tau, t, jk);
      for(j = k + 1; j < n; ++j) J[ri + (size_t)j * ldj] = FMA(-stt, v[j], J[ri + (size_t)j * ldj]);
    }
  }
}

// Triangular solves with R (upper, ld) of order q on one workgroup, one barrier per step, element e on thread e mod 1024:
//   FWD: out = R^-T b   (per element the terms in ascending i, then one division -- the oracle's beta)
//   else: out = R^-1 (a + b), back substitution (per element the terms in descending k)
template<bool FWD>
__global__ __launch_bounds__(1024) void warm_tri_kernel(const double * __restrict__ R, int ld, int q, const double * __restrict__ a,
                                                       const double * __restrict__ b, double * out)
{
  extern __shared__ double ws[]; // the running vector (q) | pivots' quotients (q)
  double * acc = ws;
  double * quo = ws + q;
  const int tid = threadIdx.x;
  for(int e = tid; e < q; e += 1024) acc[e] = FWD ? b[e] : (a[e] + b[e]);
  __syncthreads();
  if(FWD)
  {
    for(int i = 0; i < q; ++i)
    {
      if(tid == (i & 1023)) quo[i] = acc[i] / R[i + (size_t)i * ld];
      __syncthreads();
      const double v = quo[i];
      for(int e = tid; e < q; e += 1024)
        if(e > i) acc[e] = FMA(-R[i + (size_t)e * ld], v, acc[e]);
    }
  }
  else
  {
    for(int k = q - 1; k >= 0; --k)
    {
      if(tid == (k & 1023)) quo[k] = acc[k] / R[k + (size_t)k * ld];
      __syncthreads();
      const double v = quo[k];
      for(int e = tid; e < q; e += 1024)
        if(e < k) acc[e] = FMA(-R[e + (size_t)k * ld], v, acc[e]);
    }
  }
  __syncthreads();
  for(int e = tid; e < q; e += 1024) out[e] = quo[e];
}

// x = J1 beta - J2 alpha2 (:474-476): thread = row, one chain (k ascending)
__global__ __launch_bounds__(64) void warm_x_kernel(const double * __restrict__ J, int ldj, int n, int q, const double * __restrict__ beta,
                                                    const double * __restrict__ alpha, double * x)
{
  const int i = blockIdx.x * 64 + threadIdx.x;
  if(i >= n) return;
  double acc = 0.0;
  for(int k = 0; k < q; ++k) acc = FMA(J[i + (size_t)k * ldj], beta[k], acc);
  for(int k = q; k < n; ++k) acc = FMA(-J[i + (size_t)k * ldj], alpha[k], acc);
  x[i] = acc;
}

// f = beta.(0.5 beta + alpha1) - 0.5 |alpha2|^2 (:484-486): two in-order chains
__global__ void warm_f_kernel(int n, int q, const double * __restrict__ beta, const double * __restrict__ alpha, double * f)
{
  if(threadIdx.x != 0) return;
  double f1 = 0.0, f2 = 0.0;
  for(int k = 0; k < q; ++k) f1 = FMA(beta[k], FMA(0.5, beta[k], alpha[k]), f1);
  for(int k = q; k < n; ++k) f2 = FMA(alpha[k], alpha[k], f2);
  *f = FMA(-0.5, f2, f1);
}
} // namespace

// ============================================================================================================
// host-side launchers
// ============================================================================================================
static size_t rot_lds(int T)
{
  return sizeof(double) * (size_t)((RSLOTS + 2) * RSLOT + 2 * T); // (T = 4095: 163 824 of the CU's 163 840 bytes)
}
static size_t tile_lds(int n)
{
  return sizeof(double) * (size_t)(n + (n & 1) + TS_NS * TS_SLOT); // the vector | the tile slots (jtn, select_partial)
}
// CANON C10: do the tile sums of r = S d1 fit in LDS beside d1 and r?  (otherwise: DenseState::Ptri)
static bool tail_tiles_in_lds(int q)
{
  const size_t nsg = (size_t)(q + 63) >> 6;
  return sizeof(double) * (64 * nsg + q + 2 + nsg * (nsg + 1) / 2 * 64) <= 96 * 1024;
}
static size_t tail_lds(int n, int q)
{
  const int b1 = 3 * n + 2 * ((n >> 6) + 1) + 16; // block 1: z | x | n+ (the two padded, one entry per 64) | the per-wave partial sums of the two C4 trees
  size_t b0 = (size_t)q + 2;
  if(n > kBlockN) // CANON C10: d1 padded to whole segments | r | the tile sums while they fit
  {
    const size_t nsg = (size_t)(q + 63) >> 6;
    b0 = 64 * nsg + q + 2;
    if(tail_tiles_in_lds(q)) b0 += nsg * (nsg + 1) / 2 * 64;
  }
  return sizeof(double) * ((size_t)b1 > b0 ? (size_t)b1 : b0);
}
// Dynamic LDS beyond the default 64 KiB (step_tail: three n-vectors, n > 2730; rot_apply: the (c, s) pairs of a sweep,
// more than ~1900 rotations; add_chain: the third vector of the seeded recurrence): the limit is raised once, to the
// 160 KiB of a gfx950 CU.
constexpr size_t kLdsDefault = 65536, kLdsMax = 160 * 1024;
// hipFuncSetAttribute is per DEVICE: the "already raised" flags are one bit per device id, atomically set (one process may
// drive several GPUs, from several threads); beyond 64 devices the attribute is simply set on every call.
struct LdsRaised
{
  std::atomic<unsigned long long> mask{0};
};
template<class K>
static hipError_t raise_lds(K kernel, LdsRaised & done, size_t need)
{
  if(need <= kLdsDefault) return hipSuccess;
  if(need > kLdsMax) return hipErrorInvalidValue;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if(e != hipSuccess) return e;
  const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
  if(bit && (done.mask.load(std::memory_order_acquire) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
  if(e == hipSuccess && bit) done.mask.fetch_or(bit, std::memory_order_release);
  return e;
}
static hipError_t tail_attr(int n)
{
  static LdsRaised done;
  const size_t need = tail_lds(n, 0); // (block 1; block 0 of CANON C10 takes up to 96 KiB: tail_tiles_in_lds)
  return raise_lds(&step_tail_kernel, done, (n > kBlockN && need < 96 * 1024) ? 96 * 1024 : need);
}
static hipError_t tile_attr(int n) // (n > ~1800: the vector and six slots pass the default 64 KiB)
{
  static LdsRaised done_j, done_s;
  hipError_t e = raise_lds(&jtn_kernel, done_j, tile_lds(n));
  return e == hipSuccess ? raise_lds(&select_partial_kernel, done_s, tile_lds(n)) : e;
}
static hipError_t rot_attr(int T)
{
  static LdsRaised done;
  return raise_lds(&rot_apply_kernel, done, rot_lds(T));
}

// add_chain_kernel: the seeded recurrence needs a third LDS vector; beyond a CU's LDS it falls back to the plain one
static size_t chain_lds(int n)
{
  return sizeof(double) * (2 * (size_t)n + 1 + 64 + 2); // d | S / rho | block carries | flag
}
static hipError_t chain_attr(int n)
{
  static LdsRaised done;
  return raise_lds(&add_chain_kernel, done, chain_lds(n));
}

hipError_t dense_llt(double * G, int ldg, int n, int * d_info, hipStream_t st)
{
  for(int j0 = 0; j0 < n; j0 += NB)
  {
    const int nb = (n - j0 < NB) ? (n - j0) : NB;
    const int rows = n - j0 - nb;
    hipLaunchKernelGGL(llt_panel_fused_kernel, dim3(rows > 0 ? (rows + 63) / 64 : 1), dim3(64), 0, st, G, ldg, n, j0, d_info);
    if(rows > 0)
    {
      const int t = (rows + 15) / 16;
      hipLaunchKernelGGL(llt_update_kernel, dim3(t, t), dim3(64), 0, st, G, ldg, n, j0, d_info);
    }
  }
  return hipGetLastError();
}

hipError_t dense_jinv(const double * L, int ldl, double * J, int n, hipStream_t st)
{
  hipError_t e = hipMemsetAsync(J, 0, sizeof(double) * (size_t)n * n, st);
  if(e != hipSuccess) return e;
  const int nblk = (n + NB - 1) / NB;
  for(int b = nblk - 1; b >= 0; --b)
  {
    const int i0 = b * NB;
    const int cols = n - i0;
    hipLaunchKernelGGL(jinv_block_kernel, dim3((cols + 63) / 64), dim3(256), 0, st, L, ldl, J, n, n, i0);
  }
  return hipGetLastError();
}

hipError_t dense_x0(const double * L, int ldl, int n, const double * a, double * x, double * f0, hipStream_t st)
{
  hipLaunchKernelGGL(x0_kernel, dim3(1), dim3(512), sizeof(double) * 2 * n, st, L, ldl, n, a, x, f0);
  return hipGetLastError();
}

#ifndef JRLQP_BLK_TARGET
#define JRLQP_BLK_TARGET 512
#endif
// CANON C8 launch geometry: the 64-row tiles of a column group are split over `splits` workgroups of `tps` tiles each, so that about
// kBlkTarget workgroups stream the matrix (256 CUs; a workgroup of two waves)
constexpr int kBlkTarget = JRLQP_BLK_TARGET;
static void blk_split(int n, int groups, int & tps, int & splits)
{
  const int ntiles = (n + KT - 1) / KT;
  int want = (groups > 0) ? (kBlkTarget + groups - 1) / groups : 1;
  if(want < 1) want = 1;
  if(want > ntiles) want = ntiles;
  tps = (ntiles + want - 1) / want;
  splits = (ntiles + tps - 1) / tps;
}
static hipError_t tile_attr_blk(int n)
{
  static LdsRaised done_j, done_s;
  hipError_t e = raise_lds(&jtn_blk_kernel, done_j, tile_lds(n));
  return e == hipSuccess ? raise_lds(&select_blk_kernel, done_s, tile_lds(n)) : e;
}

hipError_t dense_select(const DenseViews & v, DenseState & s, int excl, hipStream_t st, int zero_u, int wait_x_tag, int post_sel_tag, int gate_tag)
{
  const int parts = (v.m + CT - 1) / CT;
  const StepDev * gate = nullptr;
  if(gate_tag)
  {
    if(!(v.n > kBlockN && parts > 0)) return hipErrorInvalidValue; // (only the blocked scan is ever enqueued ahead)
    gate = s.sd;
    hipLaunchKernelGGL(decision_gate_kernel, dim3(1), dim3(1), 0, st, s.sd, gate_tag, s.sync, wait_x_tag, s.mb);
    wait_x_tag = 0; // (the gate has waited for x: the scan's workgroups do not)
  }
  {
    hipError_t e = tile_attr(v.n);
    if(e != hipSuccess) return e;
  }
  if(parts > 0 && v.n > kBlockN)
  {
    hipError_t e = tile_attr_blk(v.n);
    if(e != hipSuccess) return e;
    int tps, splits;
    blk_split(v.n, parts, tps, splits);
    hipLaunchKernelGGL(select_blk_kernel, dim3(parts, splits), dim3(TILE_THREADS), tile_lds(v.n), st, v.C, v.ldc, v.n, v.m, s.x, v.bl, v.bu, s.st, excl,
                       s.pval, s.pidx, s.pstat, s.sync, wait_x_tag, s.mb, BlkArgs{s.partC, s.ldpC, s.cntC, tps}, gate, gate_tag);
    hipLaunchKernelGGL(select_comb_kernel, dim3(parts), dim3(64), 0, st, s.partC, s.ldpC, (v.n + KT - 1) / KT, v.m, v.bl, v.bu, s.st, excl, s.pval, s.pidx,
                       s.pstat, gate, gate_tag, s.psel);
  }
  else if(parts > 0)
    hipLaunchKernelGGL(select_partial_kernel, dim3(parts), dim3(TILE_THREADS), tile_lds(v.n), st, v.C, v.ldc, v.n, v.m, s.x, v.bl, v.bu, s.st,
                       excl, s.pval, s.pidx, s.pstat, s.sync, wait_x_tag, s.mb);
  hipLaunchKernelGGL(select_final_kernel, dim3(1), dim3(256), 0, st, parts, s.pval, s.pidx, s.pstat, v.m, v.nb, s.x, v.xl, v.xu, s.st, excl, s.mb, s.psel,
                     zero_u >= 0 ? s.u + zero_u : nullptr, s.sync, parts > 0 ? 0 : wait_x_tag, post_sel_tag, gate, gate_tag);
  return hipGetLastError();
}

hipError_t dense_step(const DenseViews & v, DenseState & s, int p, int pstat, int q, double big, hipStream_t st, hipStream_t spec, int tag,
                      int wait_sel_tag, int wait_chain_tag, hipStream_t zside, hipEvent_t ev_d, hipEvent_t ev_chain)
{
  const int n = v.n;
  const bool dev_p = (p == kSelOnDevice); // (p, status) are read from s.psel by the kernels
  const double sign = (pstat == UPPER || pstat == UPPER_BOUND) ? -1.0 : 1.0;
  const double * cp = nullptr;
  // (a problem without any constraint has no selection, so neither jtn nor jrow runs and nobody would ever publish d_tag: nothing
  // may be enqueued that waits for it -- the host loop returns before its first pass in that case, this is the second fence)
  const bool no_producer = dev_p && v.m == 0 && v.nb == 0;
  const bool chain = spec && q < n && !no_producer;
  const bool events = (ev_d != nullptr);   // event mode (kernels of different streams cannot run side by side: see PassSync): no tags
  const bool zs = zside && tag && !events && !no_producer && !(v.n > kBlockN); // z = J2 d2 on its own stream, beside the substitution (both only need d)
  const bool blk_early = v.n > kBlockN;
  // d_tag is published for whoever runs beside the main stream; blocked d (n > 256): by the step kernel, once z_blk_kernel has combined d
  // (round 5: posted by the row-group-0 workgroups of z_blk_kernel itself, one kernel earlier -- their 16 agent-scope releases made that
  // kernel 4.2 -> 8.1 us and the sweep waited no less: the chain was never late)
  const int dpost_any = (!events && (chain || zs)) ? tag : 0;
  const int dpost = blk_early ? 0 : dpost_any;
  if(events) wait_sel_tag = wait_chain_tag = 0;
  const int jt_groups = (n + CT - 1) / CT, jr_wgs = (n + 255) / 256;
  const bool blk = n > kBlockN; // CANON C8: the streaming products blocked (64-row segments + tree), their rows split over workgroups
  int jt_tps = 0, jt_splits = 1;
  if(blk) blk_split(n, jt_groups, jt_tps, jt_splits);
  const int nseg = (n + KT - 1) / KT, comb_wgs = (n + 63) / 64;
  const int jt_wgs = blk ? comb_wgs : jt_groups; // the workgroups that arrive at the d counter
  {
    hipError_t e = tile_attr(n);
    if(e == hipSuccess && blk) e = tile_attr_blk(n);
    if(e != hipSuccess) return e;
  }
  const BlkArgs bj{s.partJ, s.ldpJ, s.cntJ, jt_tps};
  if(dev_p)
  {
    // (with general constraints AND bounds both kernels run -- each returns at once unless the selection is of its kind -- and both
    // arrive at the d counter; the second one is behind the first in the stream: it has nothing left to wait for)
    DSync ds{s.sync, s.mb, wait_sel_tag, wait_chain_tag, dpost, (v.m > 0 ? jt_wgs : 0) + (v.nb > 0 ? jr_wgs : 0)};
    if(v.m > 0)
    {
      if(blk)
      {
        hipLaunchKernelGGL(jtn_blk_kernel, dim3(jt_groups, jt_splits), dim3(TILE_THREADS), tile_lds(n), st, s.J, n, n, n, nullptr, 1.0, s.d, s.psel, v.C, v.ldc,
                           v.m, ds, bj); // (the segment sums only: z_blk_kernel applies the tree)
      }
      else
        hipLaunchKernelGGL(jtn_kernel, dim3(jt_wgs), dim3(TILE_THREADS), tile_lds(n), st, s.J, n, n, n, nullptr, 1.0, s.d, s.psel, v.C, v.ldc, v.m, ds);
      ds.wait_sel = ds.wait_chain = 0;
    }
    if(v.nb > 0) hipLaunchKernelGGL(jrow_kernel, dim3(jr_wgs), dim3(256), 0, st, s.J, n, n, 0, 1.0, s.d, s.psel, v.m, ds);
  }
  else if(p < v.m)
  {
    cp = v.C + (size_t)p * v.ldc;
    if(blk)
    {
      const DSync ds{s.sync, s.mb, 0, wait_chain_tag, dpost, jt_wgs};
      hipLaunchKernelGGL(jtn_blk_kernel, dim3(jt_groups, jt_splits), dim3(TILE_THREADS), tile_lds(n), st, s.J, n, n, n, cp, sign, s.d, nullptr, nullptr, 0, 0,
                         ds, bj);
    }
    else
      hipLaunchKernelGGL(jtn_kernel, dim3(jt_wgs), dim3(TILE_THREADS), tile_lds(n), st, s.J, n, n, n, cp, sign, s.d, nullptr, nullptr, 0, 0,
                         DSync{s.sync, s.mb, 0, wait_chain_tag, dpost, jt_wgs});
  }
  else
    hipLaunchKernelGGL(jrow_kernel, dim3(jr_wgs), dim3(256), 0, st, s.J, n, n, p - v.m, sign, s.d, nullptr, 0,
                       DSync{s.sync, s.mb, 0, wait_chain_tag, dpost, jr_wgs});
  const bool gen_host = !dev_p && p < v.m; // (a general constraint selected by the host: the hook level)
  if(blk)
  {
    // z in the main stream, behind d and ahead of the step kernel, whose dot-product workgroup adds the segment sums itself: two
    // launches less than "z on the side stream + its own combine kernel" for a host that had become the bottleneck of a pass (ten
    // launches of ~4 us), at the price of z (a few us now that every CU streams J) no longer running beside the substitution.
    // Round 4: it also combines d (see the kernel) -- one more launch off the critical path of a pass.
    const dim3 zg((n + 63) / 64, (n + 63) / 64);
    const bool have_sums = (dev_p && v.m > 0) || gen_host;
    hipLaunchKernelGGL(z_blk_kernel, zg, dim3(64), 0, st, s.J, n, n, q, s.d, s.z, nullptr, 0, 0, nullptr, BlkArgs{s.partZ, s.ldpJ, s.cntZ, 1},
                       have_sums ? s.partJ : nullptr, s.ldpJ, nseg, dev_p ? s.psel : nullptr, v.m, sign, s.d, s.sync, events ? 0 : wait_chain_tag, s.mb);
  }
  if(chain)
  {
    // d is all the coefficient recurrence needs: it is enqueued now, on its own stream, and starts when d_tag says d is complete
    hipError_t e = chain_attr(n);
    if(e != hipSuccess) return e;
    if(events)
    {
      e = hipEventRecord(ev_d, st);
      if(e == hipSuccess) e = hipStreamWaitEvent(spec, ev_d, 0);
      if(e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(add_chain_kernel, dim3(1), dim3(CHAIN_THREADS), chain_lds(n), spec, n, q, s.d, s.cc2, s.ss2, s.rfin, 0, s.sync, events ? 0 : tag, s.mb);
    if(events)
    {
      e = hipEventRecord(ev_chain, spec);
      if(e != hipSuccess) return e;
    }
  }
  if(blk)
    ; // (launched above, in front of the coefficient chain)
  else if(zs)
    hipLaunchKernelGGL(z_rows_kernel, dim3((n + 15) / 16), dim3(64), 0, zside, s.J, n, n, q, s.d, s.z, s.sync, tag, tag, s.mb);
  else
    hipLaunchKernelGGL(z_rows_kernel, dim3((n + 15) / 16), dim3(64), 0, st, s.J, n, n, q, s.d, s.z);
  {
    hipError_t e = tail_attr(n);
    if(e != hipSuccess) return e;
  }
  if(dev_p)
    hipLaunchKernelGGL(step_tail_kernel, dim3(2), dim3(512), tail_lds(n, q), st, s.R, n, n, q, s.d, s.r, s.u, s.st, big, s.z, s.x,
                       nullptr, 0, s.mb, s.psel, v.C, v.ldc, v.m, tag, s.sync, zs ? tag : 0, s.sd, v.bl, v.bu, v.xl, v.xu, blk ? s.partZ : nullptr, s.ldpJ, nseg,
                       s.z, blk ? s.sync : nullptr, blk ? dpost_any : 0, (n > kBlockN) ? s.S : nullptr,
                       (n > kBlockN && !tail_tiles_in_lds(q)) ? s.Ptri : nullptr);
  else
    hipLaunchKernelGGL(step_tail_kernel, dim3(2), dim3(512), tail_lds(n, q), st, s.R, n, n, q, s.d, s.r, s.u, s.st, big, s.z, s.x, cp,
                       p - v.m, s.mb, nullptr, nullptr, 0, 0, tag, s.sync, zs ? tag : 0, nullptr, nullptr, nullptr, nullptr, nullptr, blk ? s.partZ : nullptr,
                       s.ldpJ, nseg, s.z, blk ? s.sync : nullptr, blk ? dpost_any : 0, (n > kBlockN) ? s.S : nullptr,
                       (n > kBlockN && !tail_tiles_in_lds(q)) ? s.Ptri : nullptr);
  return hipGetLastError();
}

hipError_t dense_advance(const DenseViews & v, DenseState & s, double t, int q, bool primal, hipStream_t st, int post_x_tag)
{
  const int cnt = (v.n > q + 1) ? v.n : (q + 1);
  hipLaunchKernelGGL(advance_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, v.n, q, t, primal ? 1 : 0, s.z, s.x, s.r, s.u, s.sync, post_x_tag);
  return hipGetLastError();
}

hipError_t dense_add(const DenseViews & v, DenseState & s, int p, int pstat, int q, hipStream_t st, int wait_chain_tag)
{
  const int n = v.n;
  const bool ran_ahead = (wait_chain_tag != 0);
  if(wait_chain_tag < 0) wait_chain_tag = 0; // event mode: the chain ran ahead and the stream is already ordered behind it
  if(!ran_ahead)
  {
    hipError_t e = chain_attr(n);
    if(e != hipSuccess) return e;
    hipLaunchKernelGGL(add_chain_kernel, dim3(1), dim3(CHAIN_THREADS), chain_lds(n), st, n, q, s.d, s.cc2, s.ss2, s.rfin, 0, nullptr, 0, nullptr);
  }
  {
    hipError_t e = rot_attr(n);
    if(e != hipSuccess) return e;
  }
  if(n - 2 >= q)
    hipLaunchKernelGGL(rot_apply_kernel, dim3((n + 63) / 64), dim3(ROT_THREADS), rot_lds(n - 1 - q), st, s.J, n, n, q, n - 2, 1, s.cc2, s.ss2,
                       CommitArgs{s.d, s.rfin, s.R, s.st, q, n, p, pstat, s.sync, wait_chain_tag, s.mb, (n > kBlockN) ? s.S : nullptr, s.r});
  else // q = n - 1: nothing to rotate
    hipLaunchKernelGGL(add_commit_kernel, dim3(q / 256 + 1), dim3(256), 0, st, q, s.d, s.rfin, s.R, n, s.st, p, pstat, s.sync, wait_chain_tag, s.mb,
                       (n > kBlockN) ? s.S : nullptr, s.r);
  return hipGetLastError();
}

hipError_t dense_add_speculative(const DenseViews & v, DenseState & s, int q, hipStream_t st, int tag, int wait_chain_tag, int post_x_tag)
{
  const int n = v.n;
  if(n - 2 < q) return hipErrorInvalidValue; // (q = n - 1: nothing to rotate -- the caller takes the decided path)
  hipError_t e = rot_attr(n);
  if(e != hipSuccess) return e;
  hipLaunchKernelGGL(rot_apply_kernel, dim3((n + 63) / 64), dim3(ROT_THREADS), rot_lds(n - 1 - q), st, s.J, n, n, q, n - 2, 1, s.cc2, s.ss2,
                     CommitArgs{s.d, s.rfin, s.R, s.st, q, n, -1, 0, s.sync, wait_chain_tag, s.mb, (n > kBlockN) ? s.S : nullptr, s.r},
                     SpecArgs{s.sd, tag, s.psel, s.z, s.r, s.x, s.u, s.sync, post_x_tag});
  return hipGetLastError();
}

void dense_rot_profile_dump()
{
#ifdef JRLQP_ROT_PROF
  unsigned long long h[32] = {0};
  if(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rotprof), sizeof(h)) != hipSuccess || (h[3] == 0 && h[24] == 0)) return;
  const double L = (double)h[3], us = 0.01;
  std::fprintf(stderr, "[rot prof] %llu launches, %.1f chunks each; per launch (us): wait chain tag %.2f, commit+stage %.2f, sweep %.2f | loader: wait data %.2f, at barriers %.2f | chain: wait B_0 %.2f, other barriers %.2f | storer A at barriers %.2f\n",
               h[3], h[8] / L, h[0] * us / L, h[1] * us / L, h[2] * us / L, h[4] * us / L, h[5] * us / L, h[6] * us / L, h[7] * us / L, h[9] * us / L);
  if(h[24])
    std::fprintf(stderr, "[tail prof] %llu launches; per launch (us): block 0: r = S d1 %.2f, ratio test + decision %.2f, pass tag %.2f | block 1: stage x, n+ and n+.x %.2f, z tree %.2f, dots %.2f, decision %.2f, pass tag %.2f\n",
                 h[24], h[21] * us / h[24], h[22] * us / h[24], h[23] * us / h[24], h[25] * us / h[24], h[26] * us / h[24], h[28] * us / h[24], h[27] * us / h[24], h[29] * us / h[24]);
  if(h[20])
    std::fprintf(stderr, "[jtn prof] %llu launches (workgroup 0); per launch (us): wait tags %.2f, stage the column %.2f, tiles %.2f, publish d %.2f\n", h[20],
                 h[16] * us / h[20], h[17] * us / h[20], h[18] * us / h[20], h[19] * us / h[20]);
#endif
}

// Can a kernel on one stream run while a kernel on another stream is running?  The pass tags (PassSync) rely on a consumer that was
// enqueued AFTER its producer never being executed BEFORE it; tools that let one kernel at a time onto the device and pick among the
// queues in an order of their own (rocprofv3 --pmc serialises every dispatch) break that.  The probe enqueues a waiting kernel on `a`
// and THEN the kernel it waits for on `b`: side by side the wait ends in microseconds; serialised, it gives up after ~2 ms and says so.
namespace
{
__global__ void probe_wait_kernel(const int * flag, int * seen)
{
  int ok = 0;
  for(int spins = 0; spins < 8192 && !ok; ++spins)
  {
    ok = (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1);
    if(!ok) __builtin_amdgcn_s_sleep(8);
  }
  *seen = ok;
}
__global__ void probe_post_kernel(int * flag)
{
  __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
} // namespace
hipError_t dense_probe_concurrency(int * dscratch2, hipStream_t a, hipStream_t b, bool * concurrent)
{
  hipError_t e = hipMemsetAsync(dscratch2, 0, 2 * sizeof(int), a);
  if(e == hipSuccess) e = hipStreamSynchronize(a);
  if(e != hipSuccess) return e;
  hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(1), 0, a, dscratch2, dscratch2 + 1);
  hipLaunchKernelGGL(probe_post_kernel, dim3(1), dim3(1), 0, b, dscratch2);
  e = hipStreamSynchronize(a);
  if(e == hipSuccess) e = hipStreamSynchronize(b);
  int seen = 0;
  if(e == hipSuccess) e = hipMemcpy(&seen, dscratch2 + 1, sizeof(int), hipMemcpyDeviceToHost);
  *concurrent = (seen == 1);
  return e;
}

hipError_t dense_remove(const DenseViews & v, DenseState & s, int l, int q, int gone, hipStream_t st)
{
  const int n = v.n;
  if(q <= 1024)
  {
    const int rthreads = ((q + 63) / 64) * 64; // thread = column: no more waves at the barriers than columns need
    hipLaunchKernelGGL(remove_r_kernel<1>, dim3(1), dim3(rthreads), 0, st, s.R, n, q, l, s.cc, s.ss, s.u, s.st, gone);
  }
  else // four columns per thread (q <= 4096 = JRLQP_DENSE_MAX_N)
    hipLaunchKernelGGL(remove_r_kernel<4>, dim3(1), dim3(1024), 0, st, s.R, n, q, l, s.cc, s.ss, s.u, s.st, gone);
  const int qn = q - 1;
  if(n > kBlockN && s.S)
  {
    // CANON C10: S follows R with the rotations just generated; the new S lands in the other buffer and the two swap roles
    if(q > 1)
      hipLaunchKernelGGL(remove_s_kernel, dim3((q - 1 + 63) / 64, 2), dim3(64), sizeof(double2) * (size_t)(q - l), st, s.S, s.S2, n, q, l, s.cc, s.ss);
    double * t = s.S;
    s.S = s.S2;
    s.S2 = t;
  }
  {
    hipError_t e = rot_attr(n);
    if(e != hipSuccess) return e;
  }
  if(qn > l) hipLaunchKernelGGL(rot_apply_kernel, dim3((n + 63) / 64), dim3(ROT_THREADS), rot_lds(qn - l), st, s.J, n, n, l, qn - 1, 0, s.cc, s.ss);
  return hipGetLastError();
}


// ---- warm start (experimental::GoldfarbIdnaniSolver): see the kernels above ------------------------------------------
hipError_t dense_warm_factor(const DenseViews & v, DenseState & s, int q, hipStream_t st)
{
  const int n = v.n;
  if(q > 0)
  {
    hipLaunchKernelGGL(warm_build_kernel, dim3(q), dim3(256), 0, st, v.C, v.ldc, n, v.m, s.wact, s.st, v.bl, v.bu, v.xl, v.xu, s.R, n, s.bact);
    hipLaunchKernelGGL(warm_fwd_kernel, dim3(q), dim3(512), sizeof(double) * n, st, v.G, v.ldg, n, s.R, n);
    static LdsRaised raised;
    hipError_t e = raise_lds(&warm_hh_apply_kernel, raised, sizeof(double) * 2 * (size_t)n + 16);
    if(e != hipSuccess) return e;
    for(int k = 0; k < q; ++k)
    {
      hipLaunchKernelGGL(warm_hh_make_kernel, dim3(1), dim3(1024), sizeof(double) * n, st, s.R, n, n, k, s.hco);
      if(k + 1 < q) hipLaunchKernelGGL(warm_hh_apply_kernel, dim3(q - k - 1), dim3(64), sizeof(double) * 2 * n, st, s.R, n, n, k, s.hco);
    }
  }
  if(q > 0 && n > kBlockN && s.S) // CANON C10: S = R^-1 of the factor just made
    hipLaunchKernelGGL(invert_upper_kernel, dim3(q), dim3(256), sizeof(double) * (size_t)q, st, s.R, s.S, n, q);
  hipError_t e = dense_jinv(v.G, v.ldg, s.J, n, st);
  if(e != hipSuccess) return e;
  if(q > 0) hipLaunchKernelGGL(warm_jq_kernel, dim3((n + 63) / 64), dim3(64), 0, st, s.J, n, n, s.R, n, q, s.hco);
  return hipGetLastError();
}

// initializePrimalDualPoints: alpha -> s.d, beta -> s.beta, x, u, f (into the mailbox's f0)
hipError_t dense_warm_point(const DenseViews & v, DenseState & s, int q, hipStream_t st)
{
  const int n = v.n;
  {
    hipError_t e = tile_attr(n);
    if(e != hipSuccess) return e;
  }
  if(n > kBlockN) // alpha = J^T a: the same product as d = J^T n+ (CANON C8 beyond kBlockN unknowns; oracle multi_dot)
  {
    hipError_t e = tile_attr_blk(n);
    if(e != hipSuccess) return e;
    const int groups = (n + CT - 1) / CT;
    int tps, splits;
    blk_split(n, groups, tps, splits);
    const DSync none{nullptr, nullptr, 0, 0, 0, 0};
    hipLaunchKernelGGL(jtn_blk_kernel, dim3(groups, splits), dim3(TILE_THREADS), tile_lds(n), st, s.J, n, n, n, v.a, 1.0, s.d, nullptr, nullptr, 0, 0, none,
                       BlkArgs{s.partJ, s.ldpJ, s.cntJ, tps});
    hipLaunchKernelGGL(blk_comb_kernel, dim3((n + 63) / 64), dim3(64), 0, st, s.partJ, s.ldpJ, (n + KT - 1) / KT, n, 1.0, s.d, nullptr, 0, none);
  }
  else
    hipLaunchKernelGGL(jtn_kernel, dim3((n + CT - 1) / CT), dim3(TILE_THREADS), tile_lds(n), st, s.J, n, n, n, v.a, 1.0, s.d, nullptr, nullptr, 0, 0);
  if(q > 0) hipLaunchKernelGGL(warm_tri_kernel<true>, dim3(1), dim3(1024), sizeof(double) * 2 * q, st, s.R, n, q, s.bact, s.bact, s.beta);
  hipLaunchKernelGGL(warm_x_kernel, dim3((n + 63) / 64), dim3(64), 0, st, s.J, n, n, q, s.beta, s.d, s.x);
  if(q > 0) hipLaunchKernelGGL(warm_tri_kernel<false>, dim3(1), dim3(1024), sizeof(double) * 2 * q, st, s.R, n, q, s.d, s.beta, s.u);
  hipLaunchKernelGGL(warm_f_kernel, dim3(1), dim3(64), 0, st, n, q, s.beta, s.d, &s.mb->f0);
  return hipGetLastError();
}

} // namespace jrlqp
