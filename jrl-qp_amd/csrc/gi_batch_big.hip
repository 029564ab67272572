// gi_batch_big.hip -- Mode B for 64 < n <= 128: the device-resident batched Goldfarb-Idnani solver when a row of the
// factors no longer fits one lane-per-row wavefront.
//
// Same entry point and same contract as gi_batch_kernel.h (jrlqp_gi_solve_batch: one workgroup solves one QP end to end, the
// whole dual active-set state machine on the device, inputs read once, results written once, results bit-identical to the
// CPU oracle), for the sizes the reference's own benchmark families reach (benchmarks/Solvers.cpp:613-623: n up to 100).
//
//   reference path restated here (jrl-umi3218/jrl-qp):
//     DualSolver::solve                      src/DualSolver.cpp:91-168
//     GoldfarbIdnaniSolver::init_            src/GoldfarbIdnaniSolver.cpp:56-82     (LLT, J = L^-T, x0, f0)
//     ::selectViolatedConstraint_            :84-134
//     ::computeStep_ / ::computeStepLength_  :136-219
//     ::addConstraint_ / ::removeConstraint_ :221-256
//     ::initActiveSet / addInitialConstraint :268-338
//     DualSolver::multipliers                src/DualSolver.cpp:38-69
//
// Layout.  J (n x n, ld = n|1) lives in LDS -- at n = 128 that is 132 KB of the CU's 160 KB, so one workgroup per CU; the
// Cholesky factor L shares that array (L is lower, J = L^-T is upper triangular: J is built row by row from the bottom
// in place, L's diagonal kept aside).  R (upper triangular, packed by columns) lives in LDS when it still fits
// beside J (n <= 104) and otherwise in a per-workgroup slice of a global workspace (66 KB: L2-resident).  C is streamed
// from global memory by the scan (one constraint per thread).  The grid is PERSISTENT -- one workgroup per CU slot, each
// looping over its share of the batch -- so that the workspace is sized by the machine, not by the batch.
//
// Work split (256 threads): one thread per row / column / constraint for the O(n^2) parts -- every output element still
// owns its in-order fma chain (CANON C1), which is what keeps the bits --, one thread for the scalar chains, barriers
// between the phases.  This kernel is the coverage path for large n: it is written for exactness and simplicity, not for
// the latency engineering of gi_batch_kernel.h.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "device_common.h"
#include "gi_batch.h"

namespace jrlqp
{
namespace
{
constexpr int WGB = 256;
constexpr int NONE_B = 0x7fffffff;
enum : int
{
  B_INACTIVE = 0,
  B_LOWER = 1,
  B_UPPER = 2,
  B_EQUALITY = 3,
  B_LOWER_BOUND = 4,
  B_UPPER_BOUND = 5,
  B_FIXED = 6
};
enum : int
{
  BT_SUCCESS = 0,
  BT_NON_POS_HESSIAN = 2,
  BT_INFEASIBLE = 3,
  BT_MAX_ITER = 4,
  BT_OVERCONSTRAINED = 6
};

struct CtlB
{
  double red_val[4];
  double bnd, cx, t1, f, rho; // bnd: the bound of the selected constraint, left by the thread that scanned it
  double dotp[8]; // per wave: its share of the C4 trees of z.z and n+.z
  double gc, gs, gr; // rotation of the removal step in flight
  int red_idx[4];
  int red_stat[4];
  int l, info, pad0, pad1;
  long long wt[4]; // (phase profile build: when each wave left its branch of computeStep_)
};

// acc <- chain over k = k0 .. k1-1 (ascending) of fma(+-a[k*sa], b[k*sb], acc); operands a group of CH ahead (flat pointers: LDS or
// global).  The left-over terms are ONE more group with clamped (in-footprint) addresses and a select per term, not a scalar loop
// (round 4: at n = 100 the four left-over terms were four exposed round trips per chain).  Requesting group g + 1 before group g
// is accumulated (two register buffers, a branch-free steady loop) was built and measured: no gain -- tools/ubench/chain_lat.hip:
// a wave moves one (LDS operand pair, fma) per ~33 cycles whatever the lookahead and however many lanes are active.
template<bool NEG, int CH = 8>
__device__ __forceinline__ double chain_asc(const double * a, long sa, const double * b, long sb, int k0, int k1, double acc)
{
  int k = k0;
  for(; k + CH <= k1; k += CH)
  {
    double av[CH], bv[CH];
#pragma unroll
    for(int u = 0; u < CH; ++u)
    {
      av[u] = a[(k + u) * sa];
      bv[u] = b[(k + u) * sb];
    }
#pragma unroll
    for(int u = 0; u < CH; ++u) acc = NEG ? FMA(-av[u], bv[u], acc) : FMA(av[u], bv[u], acc);
  }
  if(k < k1)
  {
    double av[CH], bv[CH];
#pragma unroll
    for(int u = 0; u < CH; ++u)
    {
      const int w = min(k + u, k1 - 1);
      av[u] = a[w * sa];
      bv[u] = b[w * sb];
    }
#pragma unroll
    for(int u = 0; u < CH; ++u)
    {
      const double t = NEG ? FMA(-av[u], bv[u], acc) : FMA(av[u], bv[u], acc);
      acc = (k + u < k1) ? t : acc;
    }
  }
  return acc;
}
// descending: k = k1-1 .. k0, first operand negated
template<int CH = 8>
__device__ __forceinline__ double chain_desc_neg(const double * a, long sa, const double * b, long sb, int k0, int k1, double acc)
{
  int k = k1 - 1;
  for(; k - CH + 1 >= k0; k -= CH)
  {
    double av[CH], bv[CH];
#pragma unroll
    for(int u = 0; u < CH; ++u)
    {
      av[u] = a[(k - u) * sa];
      bv[u] = b[(k - u) * sb];
    }
#pragma unroll
    for(int u = 0; u < CH; ++u) acc = FMA(-av[u], bv[u], acc);
  }
  if(k >= k0) // the left-over terms as one clamped group (see chain_asc)
  {
    double av[CH], bv[CH];
#pragma unroll
    for(int u = 0; u < CH; ++u)
    {
      const int w = max(k - u, k0);
      av[u] = a[w * sa];
      bv[u] = b[w * sb];
    }
#pragma unroll
    for(int u = 0; u < CH; ++u)
    {
      const double t = FMA(-av[u], bv[u], acc);
      acc = (k - u >= k0) ? t : acc;
    }
  }
  return acc;
}

// Lexicographic (value, index) minimum over the candidates of a wave (idx != NONE_B), every lane gets the winner and its aux word.
// DPP exchanges and readlanes only: the value's minimum, then the smallest index among the lanes that hold it, then one ballot for the
// lane (round 3 ran a butterfly of ds_bpermute round trips: 24 LDS crossbar trips per reduction, twice per pass).
__device__ __forceinline__ void argmin3(double & val, int & idx, int & aux)
{
  const bool cand = idx != NONE_B;
  const double mn = wave_min_d(cand ? val : __builtin_inf());
  const int ci = (cand && val == mn) ? idx : NONE_B;
  const int mi = wave_min_i(ci);
  if(mi == NONE_B) // (uniform) no candidate, or only NaNs
  {
    idx = NONE_B;
    return;
  }
  const int src = __builtin_ctzll(__ballot(ci == mi));
  val = mn;
  idx = mi;
  aux = __builtin_amdgcn_readlane(aux, src);
}

// The Givens coefficients of an addition on ONE wave, n up to 256 (GoldfarbIdnaniSolver.cpp:228-231), CANON C7 (oracle
// orc_givens_chain; device_common.h): the squares of d are scanned from the END of d in blocks of 64 (lane 63 = the block's highest
// index), the blocks chained by one addition each, then every lane takes a square root and two divisions per rotation.  Entries that
// cannot be squared: the serial recurrence r_i = hypot(d_i, r_{i+1}) with the plain makeGivens per rotation, as the oracle does.
// d, rho (n + 1), ssq (n + 1): LDS; cc, ss receive (c_i, s_i) for i = q .. n-2.  Returns the final radius R(q, q).
__device__ double wave_add_chain(int n, int q, const double * dsm, double * rho, double * ssq, double * cc, double * ss, int lane)
{
  n = __builtin_amdgcn_readfirstlane(n); // (block bounds and lane numbers from the SALU: see bcast_lane in device_common.h)
  q = __builtin_amdgcn_readfirstlane(q);
  if(n - 1 <= q) return dsm[n - 1]; // a single entry: no rotation
  bool ok = true;
  for(int k = q + lane; k < n; k += 64) ok = ok && c7_in_range(dsm[k]);
  if(__all(ok))
  {
    double carry = 0.0;
    for(int ktop = n - 1; ktop >= q; ktop -= 64)
    {
      const int k = ktop - 63 + lane;
      const bool in = k >= q;
      const double dk = dsm[in ? k : q];
      const double S = c7_block_scan(in ? dk * dk : 0.0, lane, 63) + carry;
      if(in) ssq[k] = S;
      carry = bcast_lane(S, 0); // (the entry 63 places from the block's end; only a full block has a successor)
    }
    wave_sync();
    for(int k = q + lane; k < n; k += 64) rho[k] = (k == n - 1) ? dsm[k] : __builtin_sqrt(ssq[k]);
    wave_sync();
    for(int i = q + lane; i <= n - 2; i += 64)
    {
      const double ri = rho[i], rup = rho[i + 1];
      const bool z = (ri == 0.0);
      const double cq = dsm[i] / ri, sq = (-rup) / ri;
      cc[i] = z ? 1.0 : cq;
      ss[i] = z ? 0.0 : sq;
    }
    return rho[q];
  }
  double r = dsm[n - 1];
  for(int i = n - 2; i >= q; --i) // the serial recurrence, every lane the same values
  {
    if(lane == 0) rho[i + 1] = r;
    r = givens_r_only(dsm[i], r);
  }
  wave_sync();
  for(int i = q + lane; i <= n - 2; i += 64)
  {
    double c, s, rn;
    make_givens(dsm[i], rho[i + 1], c, s, rn);
    cc[i] = c;
    ss[i] = s;
  }
  return r;
}

// A triangular solve on ONE wave with EPL entries per lane (rows lane, lane + 64, ...), nn <= 64 EPL unknowns: the lean step of
// wave_fwd / wave_bwd (device_common.h) -- the quotient of the pivot row formed by every lane on its own entry with the pivots'
// reciprocals hoisted (div_pre), one v_readlane pair, one fma and one select per entry block, the pivot column's coefficients
// requested a group of steps ahead, no branch inside a group.  FWD: pivots k = 0 .. nn-1, rows e > k receive fma(-coef(e, k), x_k, .);
// !FWD: pivots k = nn-1 .. 0, rows e < k.  coef(e, k) must be readable for every e < 64 EPL (the caller clamps), diag(e) for e < nn.
// Row e's chain is the sequential one.  Returns false if an operand left the range in which the hoisted reciprocal is provably the
// compiler's own quotient (the caller then redoes the solve with plain divisions); rv is then unspecified.
template<int EPL, bool FWD, class COEF, class DIAG>
__device__ __forceinline__ bool wave_tri_lean(double (&rv)[EPL], int nn, int lane, COEF coef, DIAG diag)
{
  nn = __builtin_amdgcn_readfirstlane(nn);
  double dg[EPL], rinv[EPL];
#pragma unroll
  for(int h = 0; h < EPL; ++h)
  {
    const int e = lane + 64 * h;
    dg[h] = (e < nn) ? diag(e) : 1.0;
    rinv[h] = recip_refined(dg[h]);
  }
  constexpr int TG = 8;
#pragma unroll
  for(int hq = 0; hq < EPL; ++hq)
  {
    const int hb = FWD ? hq : EPL - 1 - hq;
    const int klo = 64 * hb, khi = min(nn, 64 * hb + 64) - 1; // the pivots of entry block hb
    if(khi < klo) continue;
    double cn[EPL][TG], cv[EPL][TG];
    // group starting at pivot kg (FWD: kg, kg + 1, ...; else kg, kg - 1, ...), pivots past the block's end clamped to its last one
    auto fetch = [&](int kg) __attribute__((always_inline)) {
#pragma unroll
      for(int j = 0; j < TG; ++j)
      {
        const int k = FWD ? min(kg + j, khi) : max(kg - j, klo);
#pragma unroll
        for(int h = 0; h < EPL; ++h)
          if(FWD ? (h >= hb) : (h <= hb)) cn[h][j] = coef(lane + 64 * h, k);
      }
    };
    fetch(FWD ? klo : khi);
    for(int kg = FWD ? klo : khi; FWD ? (kg <= khi) : (kg >= klo); kg += FWD ? TG : -TG)
    {
#pragma unroll
      for(int j = 0; j < TG; ++j)
#pragma unroll
        for(int h = 0; h < EPL; ++h)
          if(FWD ? (h >= hb) : (h <= hb)) cv[h][j] = cn[h][j];
      fetch(FWD ? kg + TG : kg - TG); // (unconditional: clamped addresses; a branch here would make the steps wait for it)
#pragma unroll
      for(int j = 0; j < TG; ++j)
      {
        const int k = FWD ? kg + j : kg - j;
        const bool in = FWD ? (k <= khi) : (k >= klo);
        const int kk = in ? k : (FWD ? (1 << 30) : -1); // (scalar; past the block's end the compare below matches no lane)
        const double v = bcast_lane_s(div_pre(rv[hb], dg[hb], rinv[hb]), (FWD ? min(k, khi) : max(k, klo)) & 63);
#pragma unroll
        for(int h = 0; h < EPL; ++h)
          if(FWD ? (h >= hb) : (h <= hb))
          {
            const double upd = FMA(-cv[h][j], v, rv[h]);
            rv[h] = (FWD ? (lane + 64 * h > kk) : (lane + 64 * h < kk)) ? upd : rv[h];
          }
      }
    }
  }
  bool safe = true;
#pragma unroll
  for(int h = 0; h < EPL; ++h)
  {
    const double pre = rv[h]; // the numerators
    safe = safe && (lane + 64 * h >= nn || (div_safe_den(dg[h]) && div_safe_num(pre)));
    rv[h] = div_pre(pre, dg[h], rinv[h]);
  }
  return __all(safe);
}

// RLDS: R lives in LDS (compile-time, so that its accesses are ds_ instructions: with a run-time choice between LDS and the
// global scratch every access to R is a FLAT load, and the substitution -- two dependent loads per step -- pays ~500 cycles each)
// WARM: warm-started solve (experimental::GoldfarbIdnaniSolver, src/experimental/GoldfarbIdnaniSolver.cpp:20-112, :306-487; oracle
// orc_gi_solve_warm): the initial active set comes from the data equalities plus a guess, J and R from a Householder QR of
// B = L^-1 N, the start point in closed form, constraints with a negative multiplier purged one at a time; then the same loop.
// B (n x q) does not fit in LDS beside J: it lives in the workgroup's slice of the global workspace, behind R.
// WIDE: 128 < n <= 256: L / J (up to 526 KB) no longer fit in LDS at all -- they live in the workgroup's slice of the
// global workspace, behind R (L2 / Infinity-Cache resident: a slice is re-read every pass), LDS keeps the vectors; the wave-level
// substitutions carry four entries per lane instead of two.  Same code otherwise: every output element still owns its chain.
// BOX: experimental::BoxAndSingleConstraintSolver beyond one wavefront of variables (round 4; src/experimental/
// BoxAndSingleConstraintSolver.cpp:46-94, oracle gi_solve_impl warm == 2): G = I implicit, A.a holds x0, m == 1, nb == n, bu = +inf; the
// closed-form init_ (start point x0 clamped to the box, the clamped variables' bounds active, J a permutation, R = diag(+-1)) replaces
// the factorisation, J = L^-T, x0 and initActiveSet; then the same loop.
template<bool RLDS, bool WARM, bool WIDE = false, bool PROFB = false, int OCC = 1, bool BOX = false>
__global__ __launch_bounds__(WGB, WIDE ? 3 : OCC) void gi_batch_big_kernel(const BatchArgs A, double * ws, long ws_stride)
{
  static_assert(!(BOX && WARM), "the box solver starts cold");
  // (launch bounds: OCC = 2 -- two workgroups per CU while their LDS allows it, n <= 88: at most 256 VGPRs, a few spills; OCC = 1
  // beyond, where one workgroup fills the CU's LDS anyway and the spills would only cost; the wide variant keeps its factors in
  // memory and lives on occupancy: three waves per SIMD, 168 VGPRs)
  static_assert(!WIDE || !RLDS, "the wide variant keeps R and J in the workspace");
  constexpr int EPL = WIDE ? 4 : 2; // entries of a wave-level vector per lane (rows lane, lane + 64, ...)
  // entry k (scalar) of such a vector, in every lane
  auto vec_bcast = [&](const double (&v)[EPL], int k) __attribute__((always_inline)) {
    double r = 0.0;
#pragma unroll
    for(int h = 0; h < EPL; ++h)
      if((k >> 6) == h) r = bcast_lane(v[h], k & 63); // (k is scalar: a uniform branch)
    return r;
  };
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n = A.n, m = A.m, nb = A.nb, mt = m + nb;
  const int ld = n | 1;
  const long rlen = (long)n * (n + 1) / 2;
  const double big = A.big_bnd;

  // LDS carve-up (must match big_lds_bytes below)
  double * Rg = ws + (long)blockIdx.x * ((RLDS && !WARM) ? 0 : ws_stride); // !RLDS (per-workgroup slice of the scratch)
  double * M;                        // L (lower, during init) then J
  double * xs;
  if constexpr(WIDE)
  {
    M = Rg + ((rlen + 1) & ~1L);
    xs = smem;
  }
  else
  {
    M = smem;
    xs = M + (long)n * ld;
  }
  double * zs = xs + n;
  double * ds = zs + n;
  double * nps = ds + n;
  double * us = nps + n;
  double * rs = us + (n + 2);
  double * dgl = rs + (n + 2);       // diag(L); later the Givens radii r_{i+1} of an addition
  double * ccs = dgl + n;
  double * sss = ccs + n;
  double * ssq = sss + n;            // scratch of the coefficient chain (n + 2)
  double * tail = ssq + (n + 2);
  double * Rl = tail;                                   // RLDS
  double * Bw = Rg + (RLDS ? 0 : ((rlen + 1) & ~1L)) + (WIDE ? ((long)n * ld + 1) : 0); // WARM: N -> B = L^-1 N -> its QR factors, n x q, ld = n (behind R and, WIDE, behind L / J)
  if(RLDS) tail += rlen;
  CtlB * ctl = reinterpret_cast<CtlB *>(tail);
  int * act = reinterpret_cast<int *>(ctl + 1);
  unsigned char * st = reinterpret_cast<unsigned char *>(act + (n + 2));
  auto RP = [&](int i, int j) -> double & { // i <= j
    if constexpr(RLDS)
      return Rl[j * (j + 1) / 2 + i];
    else
      return Rg[(long)j * (j + 1) / 2 + i];
  };

  for(int b = blockIdx.x; b < A.B; b += gridDim.x)
  {
    const double * Gg = BOX ? nullptr : A.G + (size_t)b * n * n;
    const double * ag = A.a + (size_t)b * n;
    const double * Cg = A.C + (size_t)b * n * m;
    const double * blg = A.bl + (size_t)b * m;
    const double * bug = A.bu + (size_t)b * m;
    const double * xlg = nb ? A.xl + (size_t)b * nb : nullptr;
    const double * xug = nb ? A.xu + (size_t)b * nb : nullptr;
    __syncthreads(); // (the previous problem's results are out)
    // diagnostic phase profile (jrlqp_debug_set_profile, JRLQP_PROFILE_GI_BATCH): thread 0 accumulates shader-clock cycles per phase
    // -- slots: 0 load, 1 LLT, 2 x0 / J (/ warm factorisation), 3 scan + selection, 4 d = J^T n+, 5 z || R^-1 d || coefficient chain,
    // 6 scalar products, 7 step, 8 addition (sweep of J), 9 removal, 10 #additions, 11 #removals, 12-15 what waves 0-3 each needed of slot 5
    // (0: R^-1 d and the ratio test, 1: the coefficient chain, 2-3: z) -- into prof[b * stride + slot]
    const bool profb = PROFB && A.prof != nullptr && tid == 0; // (PROFB: a build of its own -- the accumulators cost the production kernel 5 %)
    long long pacc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // (16-22: the phases of a warm start's init)
    long long plast = profb ? clock64() : 0;
    auto stamp = [&](int slot) {
      if(profb)
      {
        const long long now = clock64();
        pacc[slot] += now - plast;
        plast = now;
      }
    };

    // ---------------------------------------------------------------- load G (lower triangle), reset
    if constexpr(BOX)
    {
      // J.setZero() (BoxAndSingleConstraintSolver.cpp:61); R.col(q).head(q).setZero() for whatever q comes out (:69, :80)
      for(int e = tid; e < n * ld; e += WGB) M[e] = 0.0;
      for(long e = tid; e < rlen; e += WGB) (RLDS ? Rl : Rg)[e] = 0.0;
    }
    else
    for(int e = tid; e < n * n; e += WGB)
    {
      const int i = e % n, j = e / n;
      M[i + (long)j * ld] = Gg[e];
    }
    for(int i = tid; i < mt; i += WGB) st[i] = 0;
    if(tid < n) xs[tid] = 0.0;
    if(tid == 0)
    {
      ctl->info = -1;
      ctl->f = 0.0;
    }
    __syncthreads();

    stamp(0);
    bool over = false;
    int nact0 = 0;
    int jnext = -2; // WARM, n <= 128: the next row of this thread's column of J = L^-T (built beside the QR by waves 2-3); -2: not started
    if constexpr(WARM)
    {
      // processInitialActiveSet (experimental/GoldfarbIdnaniSolver.cpp:309-386; oracle process_initial_active_set): the decision per
      // constraint -- data equalities win over the guess; guesses on infinite bounds, FIXED guesses on free variables and
      // ill-typed entries are ignored --, then the activation ORDER: bounds first, then general constraints, each by increasing
      // index; beyond n active constraints inequality entries are dropped from the back, i.e. every equality stays and the
      // first n - #eq inequality entries stay.  More equalities than unknowns: OVERCONSTRAINED, reported before the factorisation.
      const int * asg = A.as_in ? A.as_in + (size_t)b * mt : nullptr;
      for(int i = tid; i < mt; i += WGB)
      {
        int sdec = B_INACTIVE;
        const int g = asg ? asg[i] : B_INACTIVE;
        if(i < m)
        {
          const double lo = blg[i], hi = bug[i];
          if(lo == hi)
            sdec = B_EQUALITY;
          else if(g == B_LOWER && !(lo < -big))
            sdec = B_LOWER;
          else if(g == B_UPPER && !(hi > big))
            sdec = B_UPPER;
          else if(g == B_EQUALITY)
            sdec = B_EQUALITY; // (:357-371 only filters FIXED: an EQUALITY guess is taken as is)
        }
        else
        {
          const double lo = xlg[i - m], hi = xug[i - m];
          if(lo == hi)
            sdec = B_FIXED;
          else if(g == B_LOWER_BOUND && !(lo < -big))
            sdec = B_LOWER_BOUND;
          else if(g == B_UPPER_BOUND && !(hi > big))
            sdec = B_UPPER_BOUND;
        }
        st[i] = (unsigned char)sdec;
      }
      __syncthreads();
      // the order and the trimming (round 5: a ballot scan over the constraints IN ACTIVATION ORDER -- bounds first -- 256 at a time; one
      // thread walked all m + nb status bytes twice before: 380 k cycles at m + nb = 1100).  rank = inequality entries in front of this
      // one, kept while rank < n - #eq; pos = kept entries in front of it.
      if(tid == 0)
      {
        ctl->pad0 = 0;
        ctl->pad1 = 0;
        ctl->l = 0;
      }
      __syncthreads();
      {
        int c = 0;
        for(int i = tid; i < mt; i += WGB) c += (st[i] == B_EQUALITY || st[i] == B_FIXED) ? 1 : 0;
        if(c) atomicAdd(&ctl->pad0, c);
      }
      __syncthreads();
      const int neq0 = ctl->pad0;
      const int budget = n - neq0;
      __syncthreads();
      if(budget >= 0)
        for(int base = 0; base < mt; base += WGB)
        {
          const int o = base + tid;
          const int i = (o < nb) ? m + o : o - nb;
          const int sv = (o < mt) ? st[i] : B_INACTIVE;
          const bool iseq = (sv == B_EQUALITY || sv == B_FIXED);
          const bool ineq = (sv != B_INACTIVE) && !iseq;
          const unsigned long long lt = (1ull << lane) - 1ull;
          const unsigned long long m1 = __ballot(ineq);
          if(lane == 0) ctl->red_idx[wave] = __popcll(m1);
          __syncthreads();
          int rank = ctl->l;
          for(int w = 0; w < wave; ++w) rank += ctl->red_idx[w];
          rank += __popcll(m1 & lt);
          const bool keep = iseq || (ineq && rank < budget);
          const unsigned long long m2 = __ballot(keep);
          if(lane == 0) ctl->red_stat[wave] = __popcll(m2);
          __syncthreads();
          int pos = ctl->pad1;
          for(int w = 0; w < wave; ++w) pos += ctl->red_stat[w];
          pos += __popcll(m2 & lt);
          if(keep)
            act[pos] = i;
          else if(ineq)
            st[i] = B_INACTIVE;
          __syncthreads();
          if(tid == 0)
          {
            ctl->l += ctl->red_idx[0] + ctl->red_idx[1] + ctl->red_idx[2] + ctl->red_idx[3];
            ctl->pad1 += ctl->red_stat[0] + ctl->red_stat[1] + ctl->red_stat[2] + ctl->red_stat[3];
          }
          __syncthreads();
        }
      if(tid == 0) ctl->pad0 = (neq0 > n) ? 1 : 0;
      __syncthreads();
      over = ctl->pad0 != 0;
      nact0 = ctl->pad1;
      stamp(16); // processInitialActiveSet
    }
    // ---------------------------------------------------------------- LLT, left-looking column by column (orc_llt)
    int info = -1;
    for(int j = 0; j < n && !over && !BOX; ++j)
    {
      // rows i = j .. n-1 over the threads (two rounds when n - j > 256 never happens: n <= 128)
      const int i = j + tid;
      double acc = 0.0;
      if(i < n) acc = chain_asc<true>(M + i, ld, M + j, ld, 0, j, M[i + (long)j * ld]);
      if(i == j) ctl->t1 = acc; // pivot before the square root
      __syncthreads();
      const double piv = ctl->t1;
      if(piv <= 0.0)
      {
        info = j;
        break;
      }
      const double ljj = __builtin_sqrt(piv);
      if(i < n) M[i + (long)j * ld] = (i == j) ? ljj : acc / ljj;
      __syncthreads();
    }
    stamp(1);
    int term = BT_SUCCESS, it = 0, q = 0;
    if(over)
      term = BT_OVERCONSTRAINED;
    else if(info >= 0)
      term = BT_NON_POS_HESSIAN;
    else
    {
      if(A.Lout)
      {
        double * Lg = A.Lout + (size_t)b * n * n;
        for(int e = tid; e < n * n; e += WGB)
        {
          const int i = e % n, j = e / n;
          if(i >= j) Lg[e] = M[i + (long)j * ld];
        }
      }
      if constexpr(WARM)
      {
        // ---- initializeComputationData (experimental/GoldfarbIdnaniSolver.cpp:388-458; oracle initialize_computation_data)
        q = nact0;
        double * bact = ccs; // b_act (free until the first addition of the loop)
        double * hco = sss;  // Householder coefficients
        // N (+-C columns / +-unit vectors) and b_act in activation order (:393-418); a -> LDS
        for(int e = tid; e < q * n; e += WGB)
        {
          const int j = e / n, i = e % n;
          const int idx = act[j];
          const int sj = st[idx];
          double v;
          if(idx < m)
          {
            const double c = Cg[(size_t)idx * n + i];
            v = (sj == B_UPPER) ? -c : c;
          }
          else
            v = (i == idx - m) ? ((sj == B_UPPER_BOUND) ? -1.0 : 1.0) : 0.0;
          Bw[i + (long)j * n] = v;
        }
        for(int j = tid; j < q; j += WGB)
        {
          const int idx = act[j];
          const int sj = st[idx];
          bact[j] = (idx < m) ? ((sj == B_UPPER) ? -bug[idx] : blg[idx]) : ((sj == B_UPPER_BOUND) ? -xug[idx - m] : xlg[idx - m]);
        }
        if(tid < n)
        {
          nps[tid] = ag[tid];
          dgl[tid] = M[tid + (long)tid * ld];
        }
        __syncthreads();
        stamp(17); // N, b_act
        // J = L^-T for a warm start of at most 128 unknowns (round 5): it needs L alone, so waves 2-3 -- thread 128 + c runs column c, a
        // back substitution of its own that meets nobody else's entries -- build it NOW, beside the first columns of B = L^-1 N, and
        // the reflectors below are applied to J as they are made (one loop instead of a QR followed by a sweep over J).
        auto j_rows = [&](int rows) {
          if(WIDE || tid < 128 || tid - 128 >= n) return;
          const int c = tid - 128;
          if(jnext == -2) jnext = c;
          for(int r = 0; r < rows && jnext >= 0; ++r, --jnext)
          {
            const int i = jnext;
            double acc = (c == i) ? 1.0 : 0.0;
            acc = chain_desc_neg(M + (long)i * ld, 1, M + (long)c * ld, 1, i + 1, c + 1, acc); // L(k,i) * J(k,c), k = c .. i+1
            M[i + (long)c * ld] = acc / dgl[i];
          }
        };
        if constexpr(!WIDE)
        {
          if(tid == 0) ctl->pad0 = 0; // the next column of B nobody has taken yet
          __syncthreads();
          if(wave >= 2) j_rows(n);
        }
        // B = L^-1 N (:426): a wave per column, two rows per lane; per element the terms in ascending k, then one division (the oracle's
        // forward substitution).  Beyond 128 unknowns wave w takes columns w, w + 4, ...; up to 128 the waves draw columns from a
        // counter (waves 2-3 join when their columns of J are done).  L's diagonal comes from dgl: J's has replaced it in M.
        for(int j = WIDE ? wave : 0;; j += 4)
        {
          if constexpr(!WIDE)
          {
            int jj = 0;
            if(lane == 0) jj = atomicAdd(&ctl->pad0, 1);
            j = __builtin_amdgcn_readfirstlane(jj);
          }
          if(j >= q) break;
          double * col = Bw + (long)j * n;
          double bv[EPL];
#pragma unroll
          for(int h = 0; h < EPL; ++h) bv[h] = (lane + 64 * h < n) ? col[lane + 64 * h] : 0.0;
          const int ns = __builtin_amdgcn_readfirstlane(n);
          if constexpr(!WIDE)
          {
            // (round 5) the lean step of x0's substitution -- hoisted reciprocals, coefficients requested ahead, no division in a step:
            // 520 cycles a step with the plain division below, 782 k of a warm start's 3.2 M cycles at n = 100 -- licensed by its own
            // range check; the plain loop stays as the fall-back
            const int nl = n - 1;
            if(wave_tri_lean<EPL, true>(bv, n, lane, [&](int e, int k) { return M[min(e, nl) + (long)k * ld]; }, [&](int e) { return dgl[e]; }))
            {
#pragma unroll
              for(int h = 0; h < EPL; ++h)
                if(lane + 64 * h < n) col[lane + 64 * h] = bv[h];
              continue;
            }
#pragma unroll
            for(int h = 0; h < EPL; ++h) bv[h] = (lane + 64 * h < n) ? col[lane + 64 * h] : 0.0;
          }
          for(int k = 0; k < ns; ++k)
          {
            const double v = vec_bcast(bv, k) / dgl[k];
            const double * lk = M + (long)k * ld;
#pragma unroll
            for(int h = 0; h < EPL; ++h)
            {
              const int e = lane + 64 * h;
              if(e > k && e < n) bv[h] = FMA(-lk[e], v, bv[h]);
              if(e == k) bv[h] = v;
            }
          }
#pragma unroll
          for(int h = 0; h < EPL; ++h)
            if(lane + 64 * h < n) col[lane + 64 * h] = bv[h];
        }
        __syncthreads();
        if constexpr(!WIDE)
        {
          // (L is dead: the strictly lower part of J is zero -- the reflectors' scalar products below run over whole rows)
          for(int e = tid; e < n * n; e += WGB)
          {
            const int i = e % n, j = e / n;
            if(i > j) M[i + (long)j * ld] = 0.0;
          }
          __syncthreads();
        }
        stamp(18); // B = L^-1 N (and J = L^-T)
        // B = Q R, Householder, column by column (:432-440; Eigen makeHouseholder / applyHouseholderOnTheLeft), in place.
        // B lives in the workspace (global memory).  Round 5: the reflector of a step is staged in LDS (zs: free until the loop) -- its
        // norm is a chain over LDS instead of n dependent-by-eight global round trips, and the trailing columns read it from there;
        // a trailing column comes 16 entries per round trip.  The scalar product of a column with the reflector is a chain -- its
        // thread's --; the update of the column is not: the threads leave tau t_j in LDS and all four waves share the columns (lane =
        // row: a column's entries are contiguous).  Up to 128 unknowns the same reflector is applied to J = J Q (:443-444) in the same
        // step: threads 128 + i run row i's scalar product beside the trailing columns' (threads 0 .. q - k - 2 <= 127), and the
        // updates of both share the waves.  Per element the operations are unchanged.
        for(int k = 0; k < q; ++k)
        {
          double * vk = Bw + (long)k * n;
          double * vl = zs;
          if(wave == 0)
          {
#pragma unroll
            for(int h = 0; h < EPL; ++h)
              if(lane + 64 * h < n) vl[lane + 64 * h] = vk[lane + 64 * h];
            wave_sync();
            const double tailsq = chain_asc<false>(vl, 1, vl, 1, k + 1, n, 0.0); // (every lane the same chain)
            const double c0 = vl[k];
            const bool zero = tailsq <= 2.2250738585072014e-308;
            double beta = __builtin_sqrt(FMA(c0, c0, tailsq));
            beta = (c0 >= 0.0) ? -beta : beta;
            const double den = c0 - beta;
            double tau = (beta - c0) / beta;
            tau = zero ? 0.0 : tau;
            beta = zero ? c0 : beta;
            double mv[EPL];
#pragma unroll
            for(int h = 0; h < EPL; ++h) mv[h] = (lane + 64 * h > k && lane + 64 * h < n) ? vl[lane + 64 * h] : 0.0;
            wave_sync();
#pragma unroll
            for(int h = 0; h < EPL; ++h)
              if(lane + 64 * h > k && lane + 64 * h < n)
              {
                const double sv = zero ? 0.0 : mv[h] / den;
                vk[lane + 64 * h] = sv;
                vl[lane + 64 * h] = sv;
              }
            if(lane == 0)
            {
              vk[k] = beta;
              hco[k] = tau;
            }
          }
          __syncthreads();
          const double tau = hco[k];
          if(tau != 0.0 && k + 1 < n) // (uniform)
          {
            const int j = k + 1 + tid; // trailing column
            if(j < q && (WIDE || tid < 128))
            {
              double * cj = Bw + (long)j * n;
              double t = chain_asc<false, 16>(vl, 1, cj, 1, k + 1, n, 0.0);
              const double top = cj[k];
              t = t + top;
              cj[k] = FMA(-tau, t, top);
              rs[tid] = tau * t;
            }
            if(!WIDE && tid >= 128 && tid - 128 < n)
            {
              const int i = tid - 128; // row of J
              double t = chain_asc<false>(M + i, ld, vl, 1, k + 1, n, 0.0);
              const double jk = M[i + (long)k * ld];
              t = t + jk;
              M[i + (long)k * ld] = FMA(-tau, t, jk);
              ds[i] = tau * t;
            }
            __syncthreads();
            for(int jj = k + 1 + wave; jj < q; jj += WGB / 64)
            {
              double * cj = Bw + (long)jj * n;
              const double stt = rs[jj - k - 1];
#pragma unroll
              for(int h = 0; h < EPL; ++h)
              {
                const int i = lane + 64 * h;
                if(i > k && i < n) cj[i] = FMA(-stt, vl[i], cj[i]);
              }
            }
            if constexpr(!WIDE)
            {
              double sv[EPL];
#pragma unroll
              for(int h = 0; h < EPL; ++h) sv[h] = ds[min(lane + 64 * h, n - 1)];
              for(int jj = k + 1 + wave; jj < n; jj += WGB / 64) // a wave per column of J, lane = row
              {
                const double vj = vl[jj];
#pragma unroll
                for(int h = 0; h < EPL; ++h)
                {
                  const int i = lane + 64 * h;
                  if(i < n) M[i + (long)jj * ld] = FMA(-sv[h], vj, M[i + (long)jj * ld]);
                }
              }
            }
          }
          __syncthreads();
        }
        stamp(19); // QR
      }
      else if constexpr(BOX)
      {
        // ---- BoxAndSingleConstraintSolver::init_ (:46-94), thread = variable (n <= 256): the start point is x0 clamped to the box,
        // every clamped variable's bound is active (in index order) with multiplier |x0_i - bound_i|, J is the permutation that puts the
        // clamped coordinates first (free coordinate number k goes to column n-1-k) and R = diag(+1 lower / -1 upper)
        const bool in = tid < n;
        const double x0 = in ? ag[tid] : 0.0, lo = in ? xlg[tid] : 0.0, hi = in ? xug[tid] : 0.0;
        const bool blo = in && x0 < lo;
        const bool bup = in && !blo && x0 > hi;
        const unsigned long long mk = __ballot(blo || bup);
        unsigned long long * wm = reinterpret_cast<unsigned long long *>(ssq); // (scratch of the coefficient chain: free here)
        if(lane == 0) wm[wave] = mk;
        __syncthreads();
        int pos = __popcll(mk & ((1ull << lane) - 1ull)), total = 0;
        for(int w = 0; w < WGB / 64; ++w)
        {
          const int c = __popcll(wm[w]);
          if(w < wave) pos += c;
          total += c;
        }
        const double xv = blo ? lo : (bup ? hi : x0);
        const double uv = blo ? (xv - x0) : (x0 - xv);
        if(in)
        {
          xs[tid] = xv;
          const int col = (blo || bup) ? pos : (n - 1 - (tid - pos));
          M[tid + (long)col * ld] = 1.0;
          if(blo || bup)
          {
            RP(pos, pos) = blo ? 1.0 : -1.0;
            us[pos] = uv;
            act[pos] = m + tid; // (:72 "i + 1": one general constraint in front of the bounds)
            st[m + tid] = (unsigned char)(blo ? B_LOWER_BOUND : B_UPPER_BOUND);
          }
        }
        q = total;
        __syncthreads();
        if(tid == 0)
        {
          double f = 0.0; // f_ += 0.5 * u[q] * u[q], in activation order: one chain
          for(int k = 0; k < total; ++k) f = FMA(0.5 * us[k], us[k], f);
          ctl->f = f;
        }
        __syncthreads();
      }
      else
      // ---------------------------------------------------------------- x0 = -L^-T L^-1 a, f0 (GoldfarbIdnaniSolver.cpp:68-74)
      if constexpr(WIDE)
      {
        // thread i owns x_i; one barrier per substitution step (the pivot travels through LDS)
        double xi = (tid < n) ? ag[tid] : 0.0;
        for(int k = 0; k < n; ++k) // x = L.solve(a)
        {
          if(tid == k) zs[k] = xi / M[k + (long)k * ld];
          __syncthreads();
          const double v = zs[k];
          if(tid == k)
            xi = v;
          else if(tid > k && tid < n)
            xi = FMA(-M[tid + (long)k * ld], v, xi);
        }
        __syncthreads();
        for(int k = n - 1; k >= 0; --k) // L^T.solveInPlace(x)
        {
          if(tid == k) ds[k] = xi / M[k + (long)k * ld];
          __syncthreads();
          const double v = ds[k];
          if(tid == k)
            xi = v;
          else if(tid < k)
            xi = FMA(-M[k + (long)tid * ld], v, xi);
        }
        if(tid < n)
        {
          xs[tid] = -xi;
          dgl[tid] = M[tid + (long)tid * ld];
        }
        __syncthreads();
        if(tid == 0) ctl->f = 0.5 * chain_asc<false>(ag, 1, xs, 1, 0, n, 0.0);
      }
      else
      {
        if(tid < n) dgl[tid] = M[tid + (long)tid * ld];
        __syncthreads();
      }
      if constexpr(!BOX)
      {
      // ---------------------------------------------------------------- J = L^-T in place (orc_inv_lt):
      // J(i,c) = (delta_ic - sum_{k = c .. i+1, descending} L(k,i) J(k,c)) / L(i,i).  L (strictly lower) and J (upper) share the
      // array, L's diagonal is in dgl.
      if constexpr(WIDE)
      {
        // row by row from the bottom over all threads (column i of L is last read by row i of J)
        for(int i = n - 1; i >= 0; --i)
        {
          const int c = i + tid;
          if(c < n)
          {
            double acc = (c == i) ? 1.0 : 0.0;
            acc = chain_desc_neg(M + (long)i * ld, 1, M + (long)c * ld, 1, i + 1, c + 1, acc); // L(k,i) * J(k,c), k = c .. i+1
            M[i + (long)c * ld] = acc / dgl[i];
          }
          __syncthreads();
        }
      }
      else
      {
        // n <= 128 (round 4): a column of J is a back substitution of its own -- thread c runs column c from its diagonal up and meets
        // nobody else's entries, so there is no barrier between the rows (round 3 dealt the columns of a row to the lowest threads and
        // paid a barrier per row); waves 0-1 hold the columns.  Beside them wave 2 computes x0 with the lean wave-level substitutions
        // (two entries per lane, no barrier either: round 3 spent 2 n barrier steps with a division each on it).
        if(WARM ? (tid >= 128 && tid - 128 < n) : (tid < n))
        {
          // (a warm start: waves 2-3 hold the columns and have done most of their rows beside the QR)
          const int c = WARM ? tid - 128 : tid;
          for(int i = (WARM && jnext != -2) ? jnext : c; i >= 0; --i)
          {
            double acc = (c == i) ? 1.0 : 0.0;
            acc = chain_desc_neg(M + (long)i * ld, 1, M + (long)c * ld, 1, i + 1, c + 1, acc); // L(k,i) * J(k,c), k = c .. i+1
            M[i + (long)c * ld] = acc / dgl[i];
          }
        }
        else if(!WARM && wave == 2) // (a warm start computes its own point later)
        {
          double xv[EPL];
#pragma unroll
          for(int h = 0; h < EPL; ++h) xv[h] = (lane + 64 * h < n) ? ag[lane + 64 * h] : 0.0;
          const int nl = n - 1;
          bool ok = wave_tri_lean<EPL, true>(xv, n, lane, [&](int e, int k) { return M[min(e, nl) + (long)k * ld]; }, [&](int e) { return dgl[e]; });
          if(ok) ok = wave_tri_lean<EPL, false>(xv, n, lane, [&](int e, int k) { return M[k + (long)min(e, nl) * ld]; }, [&](int e) { return dgl[e]; });
          if(!ok)
          {
            // plain divisions, step by step (an operand outside the safe range of the hoisted reciprocals)
#pragma unroll
            for(int h = 0; h < EPL; ++h) xv[h] = (lane + 64 * h < n) ? ag[lane + 64 * h] : 0.0;
            for(int k = 0; k < n; ++k)
            {
              const double v = vec_bcast(xv, k) / dgl[k];
#pragma unroll
              for(int h = 0; h < EPL; ++h)
              {
                const int e = lane + 64 * h;
                if(e == k) xv[h] = v;
                if(e > k && e < n) xv[h] = FMA(-M[e + (long)k * ld], v, xv[h]);
              }
            }
            for(int k = n - 1; k >= 0; --k)
            {
              const double v = vec_bcast(xv, k) / dgl[k];
#pragma unroll
              for(int h = 0; h < EPL; ++h)
              {
                const int e = lane + 64 * h;
                if(e == k) xv[h] = v;
                if(e < k) xv[h] = FMA(-M[k + (long)e * ld], v, xv[h]);
              }
            }
          }
#pragma unroll
          for(int h = 0; h < EPL; ++h)
            if(lane + 64 * h < n) xs[lane + 64 * h] = -xv[h];
          wave_sync();
          if(lane == 0) ctl->f = 0.5 * chain_asc<false>(ag, 1, xs, 1, 0, n, 0.0);
        }
        __syncthreads();
      }
      if(!(WARM && !WIDE)) // (a warm start of at most 128 unknowns has done it, and has applied Q to J since)
      for(int e = tid; e < n * n; e += WGB) // the strictly lower part of J is zero
      {
        const int i = e % n, j = e / n;
        if(i > j) M[i + (long)j * ld] = 0.0;
      }
      }
      __syncthreads();
      if constexpr(WARM)
      {
        stamp(20); // J = L^-T
        // J = J Q, Q = H_0 H_1 ... (:443-444) beyond 128 unknowns: one row of J per thread, reflector after reflector (a row only meets
        // itself); the reflector (a column of B in the workspace) is staged in LDS -- zs and ds take turns, one barrier per reflector,
        // the next one requested before the current one is applied.  (Up to 128 unknowns the QR loop above has applied them.)
        if constexpr(WIDE)
        {
          double vnext = (tid < n && q > 0) ? Bw[tid] : 0.0;
          for(int k = 0; k < q; ++k)
          {
            double * v = (k & 1) ? ds : zs;
            if(tid < n) v[tid] = vnext;
            if(k + 1 < q && tid < n) vnext = Bw[(long)(k + 1) * n + tid];
            __syncthreads();
            const double tau = sss[k];
            if(tau != 0.0 && k + 1 < n && tid < n)
            {
              double t = chain_asc<false>(M + tid, ld, v, 1, k + 1, n, 0.0);
              const double jk = M[tid + (long)k * ld];
              t = t + jk;
              M[tid + (long)k * ld] = FMA(-tau, t, jk);
              const double stt = tau * t;
              for(int j = k + 1; j < n; ++j) M[tid + (long)j * ld] = FMA(-stt, v[j], M[tid + (long)j * ld]);
            }
          }
        }
        // R = upper triangle of the factored B
        for(int e = tid; e < q * q; e += WGB)
        {
          const int i = e % q, j = e / q;
          if(i <= j) RP(i, j) = Bw[i + (long)j * n];
        }
        __syncthreads();
        stamp(21); // J = J Q, R
      }
    }
    double * J = M;
    stamp(2);

    int p = -1, pstat = B_INACTIVE;

    // ---- shared pieces -------------------------------------------------------------------------------------------
    // n+ of a general constraint -> LDS
    auto load_np = [&](int pp) {
      if(pp < m && tid < n) nps[tid] = Cg[(size_t)pp * n + tid];
    };
    auto bound_of = [&](int pp, int ps) -> double {
      const bool lower = (ps == B_LOWER || ps == B_EQUALITY || ps == B_LOWER_BOUND || ps == B_FIXED);
      if(pp >= m) return lower ? xlg[pp - m] : xug[pp - m];
      return lower ? blg[pp] : (BOX ? __builtin_inf() : bug[pp]); // (box solver: bu_[0] = +inf, BoxAndSingleConstraintSolver.cpp:33)
    };
    // computeStep_ + the scalar products of computeStepLength_ (barriers inside: called by every thread)
    // fresh: pp was selected by a scan of the current x -- n+.x is then the scan's own chain value (ctl->cx, written by the thread
    // that owns the constraint: GoldfarbIdnaniSolver.cpp:97 and :183-190 are one expression) and is not recomputed
    double step_zz = 0.0, step_cz = 0.0; // z.z and n+.z of the step in flight (every thread holds them)
    auto compute_step = [&](int pp, int ps, bool fresh) {
      if(tid < n)
      {
        double dj = (pp < m) ? chain_asc<false>(J + (long)tid * ld, 1, nps, 1, 0, n, 0.0) : J[(pp - m) + (long)tid * ld];
        if(ps == B_UPPER || ps == B_UPPER_BOUND) dj = -dj;
        ds[tid] = dj;
      }
      // n+.x when the scan has not just left it (an initial constraint; the pass behind a removal): the C1 chain, on the last lane of
      // the workgroup BESIDE d (x does not move before the step is taken) -- round 3 ran it behind the trees, one chain on its own
      if(tid == WGB - 1 && !(fresh && pp < m)) ctl->cx = (pp < m) ? chain_asc<false>(nps, 1, xs, 1, 0, n, 0.0) : xs[pp - m];
      __syncthreads();
      stamp(4);
      if(tid >= 128)
      {
        // z = J2 d2: waves 2-3, one row per thread (two beyond n = 128)
        for(int i = tid - 128; i < n; i += 128) zs[i] = chain_asc<false>(J + i, ld, ds, 1, q, n, 0.0);
      }
      else if(tid >= 64)
      {
        // wave 1: the Givens coefficients of a possible ADDITION, speculatively -- they depend on d only, and most passes add
        // (a pass that drops simply does not use them).  rho in dgl (diag(L) is dead after the prologue).
        if(n - 2 >= q)
        {
          const double rfin = wave_add_chain(n, q, ds, dgl, ssq, ccs, sss, lane);
          if(lane == 0) ctl->rho = rfin;
        }
      }
      else
      {
        // r = R^-1 d1 on wave 0, EPL entries per lane (rows lane, lane + 64, ...), columns k = q-1 .. 0: the quotient of the
        // pivot row is formed in every lane (same operands, same bits)
        double rv[EPL], dg[EPL], rinv[EPL];
        const int qs = __builtin_amdgcn_readfirstlane(q); // (a scalar loop counter: see bcast_lane in device_common.h)
#pragma unroll
        for(int h = 0; h < EPL; ++h)
        {
          const int e = lane + 64 * h;
          rv[h] = (e < qs) ? ds[e] : 0.0;
          dg[h] = (e < qs) ? RP(e, e) : 1.0;
          rinv[h] = recip_refined(dg[h]); // every pivot's reciprocal at once (device_common.h, div_pre)
        }
        // The lean step of wave_bwd (device_common.h) with EPL entries per lane: the quotient of the pivot row is formed by every lane
        // on its own entry (three dependent operations, no broadcast of pivot or reciprocal), one v_readlane, one fma and one select
        // per entry block; the column's entries are requested a group of TG steps ahead (clamped into the column: in-footprint), no
        // branch inside a group.  Row e's chain is untouched: fma(-R(e, k), r_k, .) for k = q-1 .. e+1, then the quotient.  (Round 3
        // divided in every step and branched per step: 150-250 cycles per step, the longest phase of a pass at n = 100.)
        constexpr int TG = 8;
#pragma unroll
        for(int hb = EPL - 1; hb >= 0; --hb)
        {
          // pivots of entry block hb: k = min(qs, 64 hb + 64) - 1 .. 64 hb, in groups of TG from the top (steps past the block's
          // lower end are clamped to its first pivot and update no lane)
          const int ktop = min(qs, 64 * hb + 64) - 1, kbot = 64 * hb;
          if(ktop < kbot) continue;
          double cn[EPL][TG], cv[EPL][TG];
          auto fetch = [&](int kg) __attribute__((always_inline)) {
#pragma unroll
            for(int j = 0; j < TG; ++j)
            {
              const int k = max(kg - j, kbot);
#pragma unroll
              for(int h = 0; h <= hb; ++h) cn[h][j] = RP(RLDS ? lane + 64 * h : min(lane + 64 * h, k), k); // (LDS: an entry past the column's end is read and not used)
            }
          };
          fetch(ktop);
          for(int kg = ktop; kg >= kbot; kg -= TG)
          {
#pragma unroll
            for(int j = 0; j < TG; ++j)
#pragma unroll
              for(int h = 0; h <= hb; ++h) cv[h][j] = cn[h][j];
            fetch(kg - TG); // (unconditional: past the block's lower end the addresses are clamped, and a branch here would make the steps wait for it)
#pragma unroll
            for(int j = 0; j < TG; ++j)
            {
              const int k = kg - j;
              const int kk = (k >= kbot) ? k : -1; // (scalar; past the block's lower end the compare below matches no lane)
              const double v = bcast_lane_s(div_pre(rv[hb], dg[hb], rinv[hb]), max(k, kbot) & 63);
#pragma unroll
              for(int h = 0; h <= hb; ++h)
              {
                const double upd = FMA(-cv[h][j], v, rv[h]);
                rv[h] = (lane + 64 * h < kk) ? upd : rv[h];
              }
            }
          }
        }
        bool safe = true;
#pragma unroll
        for(int h = 0; h < EPL; ++h)
        {
          const double pre = rv[h]; // the numerators
          safe = safe && (lane + 64 * h >= qs || (div_safe_den(dg[h]) && div_safe_num(pre)));
          rv[h] = div_pre(pre, dg[h], rinv[h]);
        }
        if(!__all(safe))
        {
          // an operand left the range in which the hoisted reciprocal is provably the compiler's own quotient: redo with a / b
#pragma unroll
          for(int h = 0; h < EPL; ++h) rv[h] = (lane + 64 * h < qs) ? ds[lane + 64 * h] : 0.0;
          for(int k = qs - 1; k >= 0; --k)
          {
            double num = 0.0;
#pragma unroll
            for(int h = 0; h < EPL; ++h)
              if((k >> 6) == h) num = bcast_lane(rv[h], k & 63); // (k is scalar: a uniform branch)
            const double v = num / RP(k, k);
            const double * col = &RP(0, k);
#pragma unroll
            for(int h = 0; h < EPL; ++h)
            {
              const int e = lane + 64 * h;
              if(e < k) rv[h] = FMA(-col[e], v, rv[h]);
              if(e == k) rv[h] = v;
            }
          }
        }
#pragma unroll
        for(int h = 0; h < EPL; ++h)
          if(lane + 64 * h < q) rs[lane + 64 * h] = rv[h];
        // t1 = min u_k / r_k over r_k > 0, not EQUALITY/FIXED; status looked up by POSITION (GoldfarbIdnaniSolver.cpp:162)
        double tk = big;
        int kk = NONE_B, dummy = 0;
#pragma unroll
        for(int h = 0; h < EPL; ++h)
        {
          const int k = lane + 64 * h;
          const double rk = rv[h];
          if(k < q && rk > 0.0)
          {
            const int sk = st[k];
            const double tt = us[k] / rk;
            if(sk != B_EQUALITY && sk != B_FIXED && tt < tk)
            {
              tk = tt;
              kk = k;
            }
          }
        }
        if(kk == NONE_B) tk = big;
        argmin3(tk, kk, dummy);
        if(lane == 0)
        {
          ctl->t1 = (kk == NONE_B) ? big : tk;
          ctl->l = (kk == NONE_B) ? 0 : kk;
        }
      }
      if(PROFB && A.prof != nullptr && lane == 0) ctl->wt[wave] = clock64();
      __syncthreads();
      if(profb)
        for(int w = 0; w < 4; ++w) pacc[12 + w] += ctl->wt[w] - plast;
      stamp(5);
      // z.z and n+.z: C4 trees (CANON C5) -- one product per element, element e on thread e: each wave's adjacent-first tree over
      // its 64 elements, then the waves pairwise ((w0 + w1) + (w2 + w3); up to n = 128 the second pair is exactly +0).  (n+.x:
      // the C1 chain -- the scan's value for that constraint, or the chain beside d above)
      {
        const bool in = tid < n;
        const double zi = in ? zs[tid] : 0.0;
        const double ni = (in && pp < m) ? nps[tid] : 0.0;
        const double pz = wave_sum_tree(FMA(zi, zi, 0.0));
        const double pc = wave_sum_tree(FMA(ni, zi, 0.0));
        if(lane == 0)
        {
          ctl->dotp[2 * wave] = pz;
          ctl->dotp[2 * wave + 1] = pc;
        }
      }
      __syncthreads();
      // (the waves' shares combined by every thread for itself -- the same four additions on the same operands: no second barrier)
      step_zz = (ctl->dotp[0] + ctl->dotp[2]) + (ctl->dotp[4] + ctl->dotp[6]);
      step_cz = (pp < m) ? ((ctl->dotp[1] + ctl->dotp[3]) + (ctl->dotp[5] + ctl->dotp[7])) : zs[pp - m];
      stamp(6);
    };
    // x += t z (if primal), f, u -= t r, u[q] += t   (DualSolver.cpp:136-150)
    auto apply_step = [&](double t, double dotsc, bool dual_only) {
      if(!dual_only)
      {
        if(tid < n) xs[tid] = FMA(t, zs[tid], xs[tid]);
        if(tid == 0) ctl->f = FMA(t * dotsc, 0.5 * t + us[q], ctl->f); // uses u[q] BEFORE its update (the same thread, below)
      }
      if(tid < q) us[tid] = FMA(-t, rs[tid], us[tid]);
      if(tid == 0) us[q] = us[q] + t;
      __syncthreads();
    };
    // addConstraint_ (GoldfarbIdnaniSolver.cpp:221-237): the coefficients come from compute_step (wave 1, speculative); sweep of
    // J one row per thread, new column of R
    auto add_constraint = [&](int pp, int ps, auto && beside) {
      // ((c_i, s_i) and rho: computed beside the step by wave 1, compute_step)
      // beside(): what waves 2-3 do meanwhile when they have no row of J (n <= 128; the main loop: the NEXT pass's constraint scan)
      const bool split = !WIDE && n <= 128;
      if(tid < n)
      {
        double carry = J[tid + (long)(n - 1) * ld];
        int i = n - 2;
        for(; i - 3 >= q; i -= 4) // four columns per LDS round trip
        {
          double xv[4], cv[4], sv[4];
#pragma unroll
          for(int u = 0; u < 4; ++u)
          {
            xv[u] = J[tid + (long)(i - u) * ld];
            cv[u] = ccs[i - u];
            sv[u] = sss[i - u];
          }
#pragma unroll
          for(int u = 0; u < 4; ++u)
          {
            J[tid + (long)(i - u + 1) * ld] = FMA(sv[u], xv[u], cv[u] * carry);
            carry = FMA(-sv[u], carry, cv[u] * xv[u]);
          }
        }
        for(; i >= q; --i)
        {
          const double xv = J[tid + (long)i * ld];
          const double c = ccs[i], s = sss[i];
          J[tid + (long)(i + 1) * ld] = FMA(s, xv, c * carry);
          carry = FMA(-s, carry, c * xv);
        }
        if(n - 2 >= q) J[tid + (long)q * ld] = carry;
      }
      if(!split || tid < 128)
      {
        // R.col(q).head(q + 1) = d.head(q) | rho; the active set (beyond n = 128: threads from the top, whose rows come last)
        const int k = split ? tid : WGB - 1 - tid;
        if(k < q)
          RP(k, q) = ds[k];
        else if(k == q)
          RP(q, q) = (n - 2 >= q) ? ctl->rho : ds[q];
        if(k == 0)
        {
          act[q] = pp;
          st[pp] = (unsigned char)ps;
        }
      }
      else
        beside();
      __syncthreads();
    };
    // removeConstraint (DualSolver.cpp:237-244) + removeConstraint_ (GoldfarbIdnaniSolver.cpp:239-256), orc remove_constraint
    auto remove_constraint = [&](int l) {
      const double uv = (tid >= l && tid < q) ? us[tid + 1] : 0.0;
      const int av = (tid >= l && tid < q - 1) ? act[tid + 1] : 0;
      const int gone = act[l];
      __syncthreads();
      if(tid >= l && tid < q) us[tid] = uv;
      if(tid >= l && tid < q - 1) act[tid] = av;
      if(tid == 0) st[gone] = B_INACTIVE;
      const int qn = q - 1;
      for(int i = l; i < qn; ++i)
      {
        if(tid < i) RP(tid, i) = RP(tid, i + 1);
        if(tid == 128)
        {
          double c, s, r;
          make_givens(RP(i, i + 1), RP(i + 1, i + 1), c, s, r);
          ctl->gc = c;
          ctl->gs = s;
          ctl->gr = r;
        }
        __syncthreads();
        const double c = ctl->gc, s = ctl->gs;
        if(tid < n)
        {
          const double xv = J[tid + (long)i * ld], yv = J[tid + (long)(i + 1) * ld];
          J[tid + (long)i * ld] = FMA(-s, yv, c * xv);
          J[tid + (long)(i + 1) * ld] = FMA(s, xv, c * yv);
        }
        if(tid >= 128) // (beyond n = 128 these threads have a row of J as well: one after the other)
        {
          for(int k = i + 2 + (tid - 128); k <= qn; k += 128)
          {
            const double xv = RP(i, k), yv = RP(i + 1, k);
            RP(i, k) = FMA(-s, yv, c * xv);
            RP(i + 1, k) = FMA(s, xv, c * yv);
          }
          if(tid == 128) RP(i, i) = ctl->gr;
        }
        __syncthreads();
      }
      __syncthreads();
    };

    if constexpr(WARM)
    {
      // ---- initializePrimalDualPoints (experimental/GoldfarbIdnaniSolver.cpp:462-487), repeated after every purge (:78-109)
      double * bact = ccs;
      while(term == BT_SUCCESS)
      {
        if(tid < n) ds[tid] = chain_asc<false>(J + (long)tid * ld, 1, nps, 1, 0, n, 0.0); // alpha = J^T a
        if(wave == 1)
        {
          // beta = R^-T b_act: forward substitution, two entries per lane; per element the terms in ascending i, then one division
          double av[EPL];
#pragma unroll
          for(int h = 0; h < EPL; ++h) av[h] = (lane + 64 * h < q) ? bact[lane + 64 * h] : 0.0;
          const int qs = __builtin_amdgcn_readfirstlane(q);
          for(int i = 0; i < qs; ++i)
          {
            const double v = vec_bcast(av, i) / RP(i, i);
#pragma unroll
            for(int h = 0; h < EPL; ++h)
            {
              const int e = lane + 64 * h;
              if(e > i && e < q) av[h] = FMA(-RP(i, e), v, av[h]);
              if(e == i) av[h] = v;
            }
          }
#pragma unroll
          for(int h = 0; h < EPL; ++h)
            if(lane + 64 * h < q) rs[lane + 64 * h] = av[h];
        }
        __syncthreads();
        if(tid < n) // x = J1 beta - J2 alpha2
        {
          double acc = chain_asc<false>(J + tid, ld, rs, 1, 0, q, 0.0);
          acc = chain_asc<true>(J + tid, ld, ds, 1, q, n, acc);
          xs[tid] = acc;
        }
        if(wave == 2) // (not `else`: beyond n = 128 these threads have a row of x as well)
        {
          // u = R^-1 (alpha1 + beta), backward substitution, two entries per lane
          double uv[EPL];
#pragma unroll
          for(int h = 0; h < EPL; ++h) uv[h] = (lane + 64 * h < q) ? (ds[lane + 64 * h] + rs[lane + 64 * h]) : 0.0;
          const int qs = __builtin_amdgcn_readfirstlane(q);
          for(int k = qs - 1; k >= 0; --k)
          {
            const double v = vec_bcast(uv, k) / RP(k, k);
            const double * col = &RP(0, k);
#pragma unroll
            for(int h = 0; h < EPL; ++h)
            {
              const int e = lane + 64 * h;
              if(e < k) uv[h] = FMA(-col[e], v, uv[h]);
              if(e == k) uv[h] = v;
            }
          }
#pragma unroll
          for(int h = 0; h < EPL; ++h)
            if(lane + 64 * h < q) us[lane + 64 * h] = uv[h];
          // most negative multiplier of an inequality (first one on ties), threshold -1e-14 (:82-93)
          double cand = 0.0;
          int ci = NONE_B, dummy = 0;
#pragma unroll
          for(int h = 0; h < EPL; ++h)
          {
            const int k = lane + 64 * h;
            const double uk = uv[h];
            if(k < q)
            {
              const int sk = st[act[k]];
              if(uk < -1e-14 && sk != B_FIXED && sk != B_EQUALITY && (ci == NONE_B || uk < cand))
              {
                cand = uk;
                ci = k;
              }
            }
          }
          argmin3(cand, ci, dummy);
          if(lane == 0) ctl->l = (ci == NONE_B) ? -1 : ci;
        }
        if(tid == 192)
        {
          // f = beta.(0.5 beta + alpha1) - 0.5 |alpha2|^2
          double f1 = 0.0, f2 = 0.0;
          for(int k = 0; k < q; ++k) f1 = FMA(rs[k], FMA(0.5, rs[k], ds[k]), f1);
          for(int k = q; k < n; ++k) f2 = FMA(ds[k], ds[k], f2);
          ctl->f = FMA(-0.5, f2, f1);
        }
        __syncthreads();
        const int lmin = ctl->l;
        __syncthreads();
        if(lmin < 0) break;
        ++it;
        const double bv = (tid >= lmin && tid < q - 1) ? bact[tid + 1] : 0.0;
        __syncthreads();
        if(tid >= lmin && tid < q - 1) bact[tid] = bv;
        remove_constraint(lmin); // (its shift of u is harmless: u is recomputed)
        --q;
      }
      if(tid == 0) us[q] = 0.0;
      __syncthreads();
      stamp(22); // the start point and the purge
    }
    else if(term == BT_SUCCESS && !BOX) // (the box solver's init_ replaces initActiveSet as well: no pre-activation of xl == xu)
    {
      // ---------------------------------------------------------------- initActiveSet (GoldfarbIdnaniSolver.cpp:268-287)
      {
        // the equalities / fixed variables in index order, compacted into act[] (the very entries addConstraint_ writes again as it
        // activates them; q == 0 here): each thread tests its own constraints once, a ballot per wave and the waves' counts order them.
        // (Round 3 walked all m + nb constraints in every thread, two dependent global loads per constraint.)
        if(tid == 0) ctl->l = 0;
        __syncthreads();
        for(int base = 0; base < mt; base += WGB)
        {
          const int i = base + tid;
          const bool eq = (i < mt) && ((i < m) ? (blg[i] == bug[i]) : (xlg[i - m] == xug[i - m]));
          const unsigned long long mask = __ballot(eq);
          if(lane == 0) ctl->red_idx[wave] = __popcll(mask);
          __syncthreads();
          int off = ctl->l;
          for(int w = 0; w < wave; ++w) off += ctl->red_idx[w];
          off += __popcll(mask & ((1ull << lane) - 1ull));
          if(eq && off < n) act[off] = i;
          __syncthreads();
          if(tid == 0) ctl->l += ctl->red_idx[0] + ctl->red_idx[1] + ctl->red_idx[2] + ctl->red_idx[3];
          __syncthreads();
        }
        const int neq = ctl->l;
        __syncthreads();
        if(neq > n)
          term = BT_OVERCONSTRAINED; // UB in the reference; reported (as the oracle does)
        else if(neq > 0)
        {
          // n+ of the NEXT initial constraint is requested before the current one's step (a column of C is a global round trip)
          double npn = 0.0, bndn = 0.0;
          {
            const int p0 = act[0];
            if(p0 < m && tid < n) npn = Cg[(size_t)p0 * n + tid];
            bndn = bound_of(p0, (p0 < m) ? B_EQUALITY : B_FIXED);
          }
          for(int k = 0; k < neq; ++k)
          {
            p = act[k];
            pstat = (p < m) ? B_EQUALITY : B_FIXED;
            if(p < m && tid < n) nps[tid] = npn;
            const double bnd = bndn;
            if(k + 1 < neq)
            {
              const int pn = act[k + 1];
              if(pn < m && tid < n) npn = Cg[(size_t)pn * n + tid];
              bndn = bound_of(pn, (pn < m) ? B_EQUALITY : B_FIXED);
            }
            if(tid == 0) us[q] = 0.0;
            __syncthreads();
            compute_step(p, pstat, false);
            // addInitialConstraint (:295-338): full step, no ratio test
            const double cz = step_cz;
            double t = 0.0;
            if(__builtin_sqrt(step_zz) > 1e-14) t = (bnd - ctl->cx) / cz;
            apply_step(t, cz, false);
            add_constraint(p, pstat, [] {});
            ++q;
          }
        }
      }
    }

    // ---------------------------------------------------------------- main loop (DualSolver.cpp:103-166)
    if(term == BT_SUCCESS)
    {
      bool skip = false;
      // a thread's share of the constraint scan: constraints t, t + stride, ... (general ones, then the bounds), `excl` left out (the
      // constraint an addition is activating meanwhile: its status byte is being written)
      // general constraints g0 + t, g0 + t + stride, ... below g1, then (with_bounds) the bounds t, t + stride, ...; the accumulators may
      // arrive holding a candidate of LOWER index (the part scanned beside a sweep): strict < keeps it on a tie
      auto scan_part = [&](int t, int stride, int excl, double & bval, int & bidx, int & bstat, double & mycx, int & mybidx, double & mybnd, int g0, int g1,
                           bool with_bounds) {
        // (the limits of this thread's first bound are requested before the chains of its general constraints: one global round trip less
        // behind them)
        const double pxl = (with_bounds && t < nb) ? xlg[t] : 0.0, pxu = (with_bounds && t < nb) ? xug[t] : 0.0;
        for(int i = g0 + t; i < g1; i += stride)
        {
          if(i == excl || st[i] != B_INACTIVE) continue;
          const double cx = chain_asc<false, 32>(Cg + (size_t)i * n, 1, xs, 1, 0, n, 0.0); // (32 loads in flight: C comes from L2 / HBM)
          const double lo = blg[i];
          const double sl = cx - lo;
          if(sl < bval)
          {
            bval = sl;
            bidx = i;
            bstat = B_LOWER;
            mycx = cx;
            mybnd = lo;
          }
          else
          {
            const double up = BOX ? __builtin_inf() : bug[i];
            const double su = up - cx;
            if(su < bval)
            {
              bval = su;
              bidx = i;
              bstat = B_UPPER;
              mycx = cx;
              mybnd = up;
            }
          }
        }
        for(int i = t; with_bounds && i < nb; i += stride)
        {
          if(m + i == excl || st[m + i] != B_INACTIVE) continue;
          const double xi = xs[i];
          const double lo = (i == t) ? pxl : xlg[i];
          const double sl = xi - lo;
          if(sl < bval)
          {
            bval = sl;
            bidx = m + i;
            bstat = B_LOWER_BOUND;
            mybnd = lo;
          }
          else
          {
            const double up = (i == t) ? pxu : xug[i];
            const double su = up - xi;
            if(su < bval)
            {
              bval = su;
              bidx = m + i;
              bstat = B_UPPER_BOUND;
              mybnd = up;
            }
          }
        }
        mybidx = bidx; // this thread's own best constraint (before the reduction): it is the only thread that scanned it
      };
      // the scan of the pass BEHIND an addition starts beside that addition's sweep, on the waves that have no row of J (n <= 128): x is
      // final before the sweep, the scan does not read J, and the constraint being activated is left out explicitly.  One round of 128
      // general constraints (plus the bounds when there are no more general ones) -- the sweep is shorter than that; all four waves scan
      // the rest afterwards
      bool prescanned = false;
      double pre_val = 0.0, pre_cx = 0.0, pre_bnd = 0.0;
      int pre_idx = NONE_B, pre_stat = B_INACTIVE, pre_my = NONE_B;
      term = BT_MAX_ITER;
      for(; it < A.max_iter; ++it)
      {
        if(!skip)
        {
          // selectViolatedConstraint_ (:84-134): most negative slack, general constraints first, lower before upper, first index
          double bval = 0.0, mycx = 0.0, mybnd = 0.0;
          int bidx = NONE_B, bstat = B_INACTIVE, mybidx = NONE_B;
          if(!prescanned)
            scan_part(tid, WGB, -1, bval, bidx, bstat, mycx, mybidx, mybnd, 0, m, true);
          else
          {
            if(tid >= 128) // (waves 2-3 scanned the first constraints beside the sweep of the addition before)
            {
              bval = pre_val;
              bidx = pre_idx;
              bstat = pre_stat;
              mycx = pre_cx;
              mybidx = pre_my;
              mybnd = pre_bnd;
            }
            // the rest -- general constraints from 128 on and, if there were more than 128 of those, the bounds -- on all four waves
            if(m > 128) scan_part(tid, WGB, -1, bval, bidx, bstat, mycx, mybidx, mybnd, 128, m, true);
          }
          prescanned = false;
          argmin3(bval, bidx, bstat);
          if(lane == 0)
          {
            ctl->red_val[wave] = bval;
            ctl->red_idx[wave] = bidx;
            ctl->red_stat[wave] = bstat;
          }
          __syncthreads();
          double v = ctl->red_val[0];
          int ix = ctl->red_idx[0], sx = ctl->red_stat[0];
          for(int w = 1; w < 4; ++w)
          {
            const double rv = ctl->red_val[w];
            const int ri = ctl->red_idx[w], rst = ctl->red_stat[w];
            if(ri != NONE_B && (ix == NONE_B || rv < v || (rv == v && ri < ix)))
            {
              v = rv;
              ix = ri;
              sx = rst;
            }
          }
          __syncthreads();
          if(ix == NONE_B)
          {
            term = BT_SUCCESS; // DualSolver.cpp:114
            break;
          }
          p = ix;
          pstat = sx;
          load_np(p);
          if(mybidx == p) // (exactly one thread: the one that scanned constraint p; its candidate carries the selected status)
          {
            if(p < m) ctl->cx = mycx;
            ctl->bnd = mybnd; // (the limit the slack was formed with: bound_of(p, pstat) without another global round trip)
          }
          if(tid == 0) us[q] = 0.0; // DualSolver.cpp:121
          __syncthreads();
        }
        stamp(3);
        compute_step(p, pstat, !skip);
        // computeStepLength_ (:150-219)
        const double cz = step_cz;
        const double dotsc = (pstat == B_UPPER || pstat == B_UPPER_BOUND) ? -cz : cz; // ConstraintNormal::dot
        const double t1 = ctl->t1;
        const int l = ctl->l;
        double t2 = big;
        if(__builtin_sqrt(step_zz) > 1e-14) t2 = (ctl->bnd - ctl->cx) / cz; // (ctl->bnd == bound_of(p, pstat); it stays behind a removal: same p, same status)
        const double t = (t2 < t1) ? t2 : t1;
        if(A.trace && tid == 0 && it < A.trace_cap)
        {
          jrlqp_trace_rec * tr = A.trace + (size_t)b * A.trace_cap + it;
          tr->p = p;
          tr->status = pstat;
          tr->l = l;
          tr->kind = (t2 >= big) ? 2 : ((t == t2) ? 0 : 1);
          tr->t1 = t1;
          tr->t2 = t2;
        }
        if(t >= big)
        {
          term = BT_INFEASIBLE; // DualSolver.cpp:131
          break;
        }
        const bool dual_only = (t2 >= big);
        apply_step(t, dotsc, dual_only);
        stamp(7);
        if(!dual_only && t == t2)
        {
          add_constraint(p, pstat, [&] {
            pre_val = 0.0;
            pre_cx = 0.0;
            pre_idx = NONE_B;
            pre_stat = B_INACTIVE;
            pre_my = NONE_B;
            // one round of general constraints (the sweep beside it takes less than that); the bounds too if that was all of them
            scan_part(tid - 128, 128, p, pre_val, pre_idx, pre_stat, pre_cx, pre_my, pre_bnd, 0, min(m, 128), m <= 128);
          });
          prescanned = !WIDE && n <= 128;
          ++q;
          skip = false;
          stamp(8);
          if(profb) ++pacc[10];
        }
        else
        {
          remove_constraint(l);
          --q;
          skip = true; // same (p, status); u[q] keeps the shifted pending multiplier
          stamp(9);
          if(profb) ++pacc[11];
        }
      }
    }
    __syncthreads();

    // ---------------------------------------------------------------- results (sign-expanded multipliers, DualSolver.cpp:38-69)
    if(tid < n) A.x[(size_t)b * n + tid] = xs[tid];
    double * ug = A.u + (size_t)b * mt;
    int * sg = A.status + (size_t)b * mt;
    for(int i = tid; i < mt; i += WGB)
    {
      ug[i] = 0.0;
      sg[i] = st[i];
    }
    __syncthreads();
    if(tid < q)
    {
      const int i = act[tid];
      const int s = st[i];
      ug[i] = (s == B_UPPER || s == B_UPPER_BOUND) ? us[tid] : -us[tid];
    }
    if(tid == 0)
    {
      A.term[b] = term;
      A.iters[b] = it;
      A.f[b] = ctl->f;
    }
    if(profb)
      for(int k = 0; k < 24; ++k) A.prof[(size_t)b * A.prof_stride + k] = pacc[k];
  }
}

// J in the workgroup's workspace slice instead of LDS (the WIDE instantiations): always beyond 128 unknowns; JRLQP_BIG_WIDE_FROM=<n>
// (experiment) lowers the threshold
static int wide_from()
{
  static const int v = [] {
    const char * e = std::getenv("JRLQP_BIG_WIDE_FROM");
    const int t = e ? std::atoi(e) : 129;
    return (t >= 65 && t <= 129) ? t : 129;
  }();
  return v;
}
static bool is_wide(int n) { return n >= wide_from(); }
long big_lds_doubles(int n, int r_in_lds)
{
  const long ld = n | 1;
  long dbl = 4L * n + 2L * (n + 2) + 3L * n + (n + 2);
  if(!is_wide(n)) dbl += (long)n * ld; // J (beyond 128 unknowns it lives in the workspace: WIDE)
  if(r_in_lds) dbl += (long)n * (n + 1) / 2;
  return dbl;
}
} // namespace

bool batch_big_is_wide(int n) { return is_wide(n); }

// two_per_cu == 0: R stays in LDS wherever it fits -- the caller passed no workspace (the contract of the entries without `_ws`), or the
// batch has no more problems than the device has CUs (a workgroup alone on its CU: R in LDS is the faster place)
long batch_big_lds_bytes(int n, int m, int nb, int * r_in_lds, int two_per_cu)
{
  auto bytes_of = [&](int rl) { return ((big_lds_doubles(n, rl) * 8 + (long)sizeof(CtlB) + 4L * (n + 2) + (long)(m + nb)) + 15) & ~15L; };
  if(two_per_cu && !is_wide(n) && !std::getenv("JRLQP_BIG_R_IN_LDS"))
  {
    // (round 5) where J and the packed R fit once but not twice beside each other and J alone fits twice -- n = 79 .. 95 -- the packed R
    // goes to the workspace: two workgroups per CU with R behind L2 against one with R in LDS
    const long b1 = bytes_of(1), b0 = bytes_of(0);
    if(b1 <= kLdsPerCU && 2 * b1 > kLdsPerCU && 2 * b0 <= kLdsPerCU)
    {
      if(r_in_lds) *r_in_lds = 0;
      return b0;
    }
  }
  for(int rl = !is_wide(n) ? 1 : 0; rl >= 0; --rl)
  {
    const long bytes = bytes_of(rl);
    if(bytes <= kLdsPerCU)
    {
      if(r_in_lds) *r_in_lds = rl;
      return bytes;
    }
  }
  return -1;
}

// doubles of global workspace per resident workgroup: the packed R when it does not fit in LDS beside J (r_in_lds == 0), and for
// a warm start the n x n array of B = L^-1 N / its QR factors behind it
long batch_big_ws_doubles_per_wg(int n, int r_in_lds, int warm)
{
  long d = r_in_lds ? 0 : (((long)n * (n + 1) / 2 + 1) & ~1L);
  if(warm) d += (long)n * n;
  if(is_wide(n)) d += (long)n * (n | 1) + 1; // WIDE: L / J
  return d;
}

hipError_t launch_gi_batch_big(const BatchArgs & a, int nwg, double * ws, hipStream_t stream)
{
  int r_in_lds = 0;
  const long lds = batch_big_lds_bytes(a.n, a.m, a.nb, &r_in_lds, a.two_per_cu);
  if(lds < 0) return hipErrorInvalidValue;
  const long stride = batch_big_ws_doubles_per_wg(a.n, r_in_lds, a.warm);
  auto go = [&](auto kern) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if(e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(WGB), (size_t)lds, stream, a, ws, stride);
    return hipGetLastError();
  };
  if(a.box) // experimental::BoxAndSingleConstraintSolver beyond 64 variables (cold start only)
  {
    if(is_wide(a.n)) return go(&gi_batch_big_kernel<false, false, true, false, 1, true>);
    return r_in_lds ? go(&gi_batch_big_kernel<true, false, false, false, 1, true>) : go(&gi_batch_big_kernel<false, false, false, false, 1, true>);
  }
  if(is_wide(a.n)) return a.warm ? go(&gi_batch_big_kernel<false, true, true>) : go(&gi_batch_big_kernel<false, false, true>);
  if(a.prof && !a.warm) return r_in_lds ? go(&gi_batch_big_kernel<true, false, false, true>) : go(&gi_batch_big_kernel<false, false, false, true>);
#ifdef JRLQP_BIG_WARM_PROF // (diagnostic build: the phase profile of a warm start, tools/time_big_family.py warm ...)
  if(a.prof && a.warm) return r_in_lds ? go(&gi_batch_big_kernel<true, true, false, true>) : go(&gi_batch_big_kernel<false, true, false, true>);
#endif
  if(2 * lds <= kLdsPerCU && r_in_lds) // two workgroups per CU: the instantiations held to 256 VGPRs
    return a.warm ? go(&gi_batch_big_kernel<true, true, false, false, 2>) : go(&gi_batch_big_kernel<true, false, false, false, 2>);
  if(2 * lds <= kLdsPerCU && !is_wide(a.n)) // ... with the packed R in the workspace (batch_big_lds_bytes)
    return a.warm ? go(&gi_batch_big_kernel<false, true, false, false, 2>) : go(&gi_batch_big_kernel<false, false, false, false, 2>);
  if(a.warm) return r_in_lds ? go(&gi_batch_big_kernel<true, true>) : go(&gi_batch_big_kernel<false, true>);
  return r_in_lds ? go(&gi_batch_big_kernel<true, false>) : go(&gi_batch_big_kernel<false, false>);
}

} // namespace jrlqp
