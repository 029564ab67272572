// gi_batch_kernel.h -- Mode B: device-resident batched Goldfarb-Idnani solver for gfx950 (MI355X): the kernel template and its
// launch helpers; included by gi_batch.hip, gi_batch_wide.hip and gi_batch_wide2.hip, which instantiate it.
//
// One 256-thread workgroup (4 wave64) solves one dense QP end to end: the complete dual active-set state
// machine of the reference runs on-device, inputs are read from HBM exactly once and the results written once.
//
//   reference path restated here (jrl-umi3218/jrl-qp):
//     DualSolver::solve                      src/DualSolver.cpp:91-168
//     GoldfarbIdnaniSolver::init_            src/GoldfarbIdnaniSolver.cpp:56-82     (K1 LLT, K2 J=L^-T, K3 x0,f0)
//     ::selectViolatedConstraint_            :84-134   (K4)
//     ::computeStep_                         :136-148  (K5 d=J^T n+, K6 z=J2 d2, K7 r=R^-1 d1)
//     ::computeStepLength_                   :150-219  (K8)
//     ::addConstraint_ / ::removeConstraint_ :221-256  (K10 / K11 Givens updates of J and R)
//     ::initActiveSet / addInitialConstraint :268-338  (K12 equalities / fixed variables)
//     DualSolver::multipliers                src/DualSolver.cpp:38-69 (K13)
//
// Rounding contract (DESIGN.md §4 "CANON"): every accumulation is an in-order fma chain owned by ONE lane
// (one lane per output element), so the parallel kernel reproduces the sequential CPU oracle bit for bit;
// divisions and square roots are the correctly rounded IEEE ones; this file is compiled with -ffp-contract=off.
//
// Data placement per workgroup:
//   LDS  (doubles, ld = n|1 odd => "lane = row" and "lane = column" accesses of the n x n factors are both
//         bank-conflict free with ds_read_b64):
//         J[n*ld] | R[n*ld] (holds L during init) | x z d np [n each] | u r [n+2 each] | (C[m*ld] bl bu if CMODE==1)
//         | control block | active list int[n+2] | status bytes[m+nb]
//   VGPR (CMODE==2, compile-time n): thread i keeps column i of C (n doubles) and bl_i, bu_i in registers, so the
//         per-iteration constraint scan (K4, the largest streaming term) touches neither LDS nor HBM for C and
//         three workgroups fit one CU.
//   HBM  CMODE==0: C streamed from L2/HBM every scan (only when m is too large for the other two).
//
// Latency structure: the serial scalar chains (Givens coefficients, back-substitution pivots) set the critical
// path (216 cycles per Givens step is the measured floor for one wave: 27 dependent fp64 ops x 8 cycles).  The
// kernel therefore (i) software-pipelines every chain so that LDS latency, the 1/u division and the rotation of J
// run in the shadow of the coefficient chain, and (ii) runs the next iteration's constraint scan on waves 1-3 while
// wave 0 is inside the Givens sweep.
#pragma once
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "gi_batch.h"

namespace jrlqp
{
namespace
{
constexpr int WG = 256;
constexpr int NONE = 0x7fffffff;

enum : int
{
  INACTIVE = 0,
  LOWER = 1,
  UPPER = 2,
  EQUALITY = 3,
  LOWER_BOUND = 4,
  UPPER_BOUND = 5,
  FIXED = 6
};
enum : int
{
  T_SUCCESS = 0,
  T_NON_POS_HESSIAN = 2,
  T_INFEASIBLE = 3,
  T_MAX_ITER = 4,
  T_OVERCONSTRAINED = 6,
  T_UNKNOWN = 7
};

struct Ctl
{
  double red_val[4];
  double t1, zz, cz, cx;
  double f;
  double bnd; // bound value b of the selected constraint (bl, bu, xl or xu by status): kept out of global memory on the step path
  double bndw[4]; // CMODE 2: the same per scanning wave (each wave's own winner), see scan_publish
  double cxw[4];  // WS: n+.x of each scanning wave's winner -- the scan's own C1 chain C.col(p).dot(x), reused by the step length
  int red_idx[4];
  int red_stat[4];
  int info;
  int l;
  int eq_next;
  int cnt;
  int flag; // sequence number of the last add whose (c, s) coefficients are published (wave 0 -> wave 1 hand-off)
  int nact; // WARM: size of the initial active set
  int over; // WARM: more than n equalities / fixed variables activated
  int fack2;        // WS: pass number whose decision wave 2 has read (drop passes: nothing else ties that wave to MAIN)
  // WS (role-specialised main loop): hand-off words between the main wave and the helper waves, see ws_* below
  double rho;       // R(q, q) of the addition whose coefficients are published (chain wave -> main wave)
  int fd;           // pass number whose d (and p, pstat, q below) is published            main -> helpers
  int fz;           // pass number whose z, z.z, n.z, n.x and t2 are published               z wave -> main
  int fack0;        // pass number whose decision wave 0 has read (see fack2)
  int fx0;          // 1: x0 and f0 are published (overlapped prologue: wave 0 -> everybody)
  int fcs;          // pass number whose rotation coefficients are published (or abandoned) chain wave -> main
  int fdr;          // pass number whose d the chain wave has taken into registers          chain wave -> main
  int fstep;        // pass number whose step is taken (x, u final) and whose decision is in dec
  int dec;          // decision of pass fstep: 1 add, 2 drop, 3 stop
  int fscan[3];     // scan number whose winner is published, per scanning wave
  int wp, wps, wq;  // constraint, status and active-set size of the pass in flight
  int wfresh;       // 1: p was selected by a scan in this pass (x unchanged since: cxw holds n+.x); 0: the pass follows a removal
  int wit, wterm;   // final iteration count and termination status
  double t2;        // primal step length of the pass in flight (z wave -> main)
};

// ---- wave-level helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ int uniform(int v)
{
  return __builtin_amdgcn_readfirstlane(v);
}

// a double every lane holds identically -> SGPR pair (keeps the state machine's branches scalar)
__device__ __forceinline__ double uniform_d(double v)
{
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Lexicographic (value, index) minimum over the wave when the candidate index is monotone in the lane id
// (idx == NONE marks "no candidate" and must come with val == +big).  Butterfly min on the value, then one
// ballot picks the first lane holding it.  Every lane gets the winner.
__device__ __forceinline__ void wave_argmin_monotone(double & val, int & idx, int & aux)
{
  const double mn = wave_min_d(val);
  unsigned long long msk = __ballot(idx != NONE && val == mn);
  if(msk == 0)
  {
    idx = NONE;
    return;
  }
  int src = __builtin_ctzll(msk);
  val = mn;
  idx = __builtin_amdgcn_readlane(idx, src);
  aux = __builtin_amdgcn_readlane(aux, src);
}

// general version (index not monotone in the lane id)
__device__ __forceinline__ void wave_argmin(double & val, int & idx, int & aux)
{
#pragma unroll
  for(int off = 32; off >= 1; off >>= 1)
  {
    double ov = shfl_xor_d(val, off);
    int oi = __shfl_xor(idx, off);
    int oa = __shfl_xor(aux, off);
    bool take = (ov < val) || (ov == val && oi < idx);
    if(take)
    {
      val = ov;
      idx = oi;
      aux = oa;
    }
  }
}

// Eigen JacobiRotation::makeGivens, real case (oracle orc_givens), split in two so that the
// part the NEXT rotation depends on (r) is separated from the part only the application needs (c, s).
// Branch-free: both operand orders share one instruction stream; the special cases are selects.
struct GivensHalf
{
  double t, u; // quotient and signed sqrt(1 + t^2)
  double p, q; // inputs (for the special cases)
  bool big_p;  // |p| > |q|
};

__device__ __forceinline__ double givens_r(double p, double q, GivensHalf & h)
{
  h.p = p;
  h.q = q;
  h.big_p = fabs(p) > fabs(q);
  const double den = h.big_p ? p : q;
  const double num = h.big_p ? q : p;
  h.t = num / den;
  double u = __builtin_sqrt(FMA(h.t, h.t, 1.0));
  h.u = (den < 0.0) ? -u : u;
  double r = den * h.u;
  r = (p == 0.0) ? fabs(q) : r;
  r = (q == 0.0) ? fabs(p) : r;
  return r;
}

__device__ __forceinline__ void givens_cs(const GivensHalf & h, double & c, double & s)
{
  const double w = 1.0 / h.u; // 1/u
  const double tw = h.t * w;
  // |p|>|q|: c = 1/u, s = -t*c ;  else: s = -1/u, c = -t*s = t*(1/u)   (negations are exact)
  c = h.big_p ? w : tw;
  s = h.big_p ? -tw : -w;
  if(h.p == 0.0)
  {
    c = 0.0;
    s = (h.q < 0.0) ? 1.0 : -1.0;
  }
  if(h.q == 0.0)
  {
    c = (h.p < 0.0) ? -1.0 : 1.0;
    s = 0.0;
  }
}

// the same with v_min_f64 spelled out (three instructions per exchange instead of five); for operands that are numbers
__device__ __forceinline__ double vmin_d(double a, double b)
{
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double wave_min_d_fast(double v)
{
  v = vmin_d(v, dpp_move<0xB1>(v));
  v = vmin_d(v, dpp_move<0x4E>(v));
  v = vmin_d(v, dpp_move<0x141>(v));
  v = vmin_d(v, dpp_move<0x140>(v));
  const double r0 = bcast_lane(v, 0), r1 = bcast_lane(v, 16), r2 = bcast_lane(v, 32), r3 = bcast_lane(v, 48);
  return vmin_d(vmin_d(r0, r1), vmin_d(r2, r3));
}

// acc <- chain over k = k0 .. k1-1 (ascending) of fma(a[k*sa], b[k*sb], acc); loads are issued a chunk ahead so
// that LDS latency is paid once per chunk instead of once per step.
template<int CH = 8>
__device__ __forceinline__ double chain_up(const double * __restrict__ a, int sa, const double * __restrict__ b, int sb,
                                           int k0, int k1, double acc)
{
  int k = k0;
#pragma unroll 1
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
    for(int u = 0; u < CH; ++u) acc = FMA(av[u], bv[u], acc);
  }
  const int rem = k1 - k; // < CH: one more batched group, fmas predicated (a select costs 8 cycles, an exposed LDS load ~130)
  if(rem > 0)
  {
    double av[CH], bv[CH];
#pragma unroll
    for(int u = 0; u < CH - 1; ++u)
    {
      const int kk = (u < rem) ? (k + u) : (k1 - 1);
      av[u] = a[kk * sa];
      bv[u] = b[kk * sb];
    }
#pragma unroll
    for(int u = 0; u < CH - 1; ++u)
    {
      const double nx = FMA(av[u], bv[u], acc);
      acc = (u < rem) ? nx : acc;
    }
  }
  return acc;
}

// ascending with the first operand negated (Cholesky updates)
template<int CH = 8>
__device__ __forceinline__ double chain_up_neg(const double * __restrict__ a, int sa, const double * __restrict__ b, int sb,
                                               int k0, int k1, double acc)
{
  int k = k0;
#pragma unroll 1
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
    for(int u = 0; u < CH; ++u) acc = FMA(-av[u], bv[u], acc);
  }
  const int rem = k1 - k;
  if(rem > 0)
  {
    double av[CH], bv[CH];
#pragma unroll
    for(int u = 0; u < CH - 1; ++u)
    {
      const int kk = (u < rem) ? (k + u) : (k1 - 1);
      av[u] = a[kk * sa];
      bv[u] = b[kk * sb];
    }
#pragma unroll
    for(int u = 0; u < CH - 1; ++u)
    {
      const double nx = FMA(-av[u], bv[u], acc);
      acc = (u < rem) ? nx : acc;
    }
  }
  return acc;
}

// same with the first operand negated and indices descending: k = k1-1 .. k0
template<int CH = 8>
__device__ __forceinline__ double chain_down_neg(const double * __restrict__ a, int sa, const double * __restrict__ b, int sb,
                                                 int k0, int k1, double acc)
{
  int k = k1 - 1;
#pragma unroll 1
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
  const int rem = k - k0 + 1; // < CH
  if(rem > 0)
  {
    double av[CH], bv[CH];
#pragma unroll
    for(int u = 0; u < CH - 1; ++u)
    {
      const int kk = (u < rem) ? (k - u) : k0;
      av[u] = a[kk * sa];
      bv[u] = b[kk * sb];
    }
#pragma unroll
    for(int u = 0; u < CH - 1; ++u)
    {
      const double nx = FMA(-av[u], bv[u], acc);
      acc = (u < rem) ? nx : acc;
    }
  }
  return acc;
}

// PROF: diagnostic build only -- thread 0 accumulates shader-clock cycles per phase into A.prof[b*A.prof_stride + slot], 32 slots
// (slots: 0 load, 1 LLT, 2 J/x0, 3 scan (not overlapped), 4 d=J^T n, 5 trsv||z, 6 step+update, 7 add (+overlapped scan),
//  8 drop, 9 store, 10 #adds, 11 #drops, 12 s_memrealtime ticks (100 MHz) for the whole solve).
// The accumulators live in LDS (ds_add_u64, fire-and-forget) and are copied out once at the end: accumulating
// straight into global memory costs a load round trip per stamp and distorts every phase by 1-2 k cycles.
#define STAMP(slot)                                                                     \
  if(PROF && tid == 0)                                                                  \
  {                                                                                     \
    long long now_ = clock64();                                                         \
    atomicAdd(&prof_lds[(slot)], (unsigned long long)(now_ - last_));                   \
    last_ = clock64();                                                                  \
  }
// same for the first lane of wave 3 (the chain wave of the speculative path), with its own time base
#define STAMP3(slot)                                                                    \
  if(PROF && tid == 192)                                                                \
  {                                                                                     \
    long long now_ = clock64();                                                         \
    atomicAdd(&prof_lds[(slot)], (unsigned long long)(now_ - last3_));                  \
    last3_ = clock64();                                                                 \
  }
// same for the first lane of wave 2 (the helper that also scans the LDS-staged extra columns): slots 16 / 17
#define STAMP2(slot)                                                                    \
  if(PROF && tid == 128)                                                                \
  {                                                                                     \
    long long now_ = clock64();                                                         \
    atomicAdd(&prof_lds[(slot)], (unsigned long long)(now_ - last2_));                  \
    last2_ = clock64();                                                                 \
  }
#define PROF_COUNT(slot)                                   \
  if(PROF && tid == 0) atomicAdd(&prof_lds[(slot)], 1ull);
#define PROF_COUNT3(slot)                                  \
  if(PROF && tid == 192) atomicAdd(&prof_lds[(slot)], 1ull);

// ---- the kernel ------------------------------------------------------------------------------------------
// NT    : compile-time n (0 = runtime n)
// CMODE : 0 = C streamed from global memory, 1 = C staged in LDS, 2 = C in registers (needs NT > 0, m <= 256)
// WARM  : warm-started solve (src/experimental/GoldfarbIdnaniSolver.cpp): the initial active set comes from the data
//         equalities plus a guess, J and R from a Householder QR of L^-1 N; then the same loop
// EMB   : the problem is SMALLER than NT and runs embedded (see nr below); a separate instantiation, so that the
//         exact-size kernels -- BASELINE c3 among them -- keep nr a compile-time constant and their register allocation
// BOX   : experimental::BoxAndSingleConstraintSolver -- G = I implicit, A.a holds x0, m == 1, nb == n; the closed-form init_
//         of src/experimental/BoxAndSingleConstraintSolver.cpp:46-94 replaces LLT / J = L^-T / x0 / initActiveSet
// ---- J = L^-T out of registers (one wave, lane = column c of J, acc[i] = the running J_ic) ---------------------------------
// Row K of J is final once the rows below it are: J_Kc = acc[K] / L_KK; it then gives every row above its term
// acc[i] -= L_Ki J_Kc (L_Ki: one LDS address for the wave, a broadcast read).  Per element the terms arrive for k = n-1 .. i+1,
// descending, then one division: the order of the blocked left-looking loop this replaces (rows k > c contribute exact zeros
// in both).  Terms in groups of sixteen, pinned where they are written (see llt_reg_trail).
template<int K, int N, int I0>
__device__ __forceinline__ void jinv_reg_trail(double (&acc)[N], double v, const double * Lrow, int ld)
{
  if constexpr(I0 < K)
  {
    constexpr int I1 = (I0 + 16 < K) ? I0 + 16 : K;
    double lr[16];
#pragma unroll
    for(int i = I0; i < I1; ++i) lr[i - I0] = Lrow[i * ld];
#pragma unroll
    for(int i = I0; i < I1; ++i) acc[i] = FMA(-lr[i - I0], v, acc[i]);
#pragma unroll
    for(int i = I0; i < I1; ++i) asm volatile("" : "+v"(acc[i]));
    jinv_reg_trail<K, N, I1>(acc, v, Lrow, ld);
  }
}
template<int K, int N>
__device__ __forceinline__ void jinv_reg_rows(double (&acc)[N], int lane, const double * L, double * J, int ld)
{
  if constexpr(K >= 0)
  {
    const double v = acc[K] / L[K + K * ld];
    if(lane < N) J[K + lane * ld] = v;
    jinv_reg_trail<K, N, 0>(acc, v, L + K, ld);
    jinv_reg_rows<K - 1, N>(acc, lane, L, J, ld);
  }
}

#ifdef JRLQP_EXP_OCC4 // (experiment, tools/ubench/exp_c3.sh B -DJRLQP_EXP_OCC4: the register side of "four workgroups per CU" -- every role held to 128 VGPRs)
#define JRLQP_OCC_50 4
#else
#define JRLQP_OCC_50 3
#endif
template<int NT, int CMODE, bool PROF, bool WARM, bool EMB = false, bool BOX = false>
__global__ __launch_bounds__(WG, (NT > 0 && CMODE == 2) ? ((WARM || NT > 50) ? 2 : (NT <= 16 ? 5 : (NT <= 32 ? 4 : JRLQP_OCC_50))) : 1) void gi_batch_kernel(const BatchArgs A)
{
  long long last_ = 0, wall0_ = 0, last3_ = 0, last2_ = 0;
  if(PROF)
  {
    last_ = clock64();
    last3_ = last_;
    last2_ = last_;
    wall0_ = wall_clock64();
  }
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ unsigned long long prof_lds[PROF ? 32 : 1];
  if(PROF && threadIdx.x < 32) prof_lds[threadIdx.x] = 0;
  // (Measured in round 3 and taken out again: dealing the MAIN role out by SIMD.  The hardware starts a workgroup's four waves at a
  // SIMD of its own choice -- wave 3 lands on every SIMD equally often, so with three workgroups per CU two MAIN waves share a SIMD on
  // 5 CUs of 8.  Making the wave on SIMD (wave slot mod 4) the MAIN wave puts the three MAIN waves of a CU on three different SIMDs
  // (HW_ID confirmed it for 4096 of 4096 workgroups) -- and changes nothing: 2.35 M solves/s either way.  The serial wave is bound
  // by its own dependent latencies, not by issue slots lost to a neighbour.)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int b = blockIdx.x;
  const int n = NT ? NT : A.n;
  // A compile-time-n instantiation also takes SMALLER problems (cold start only): nr = the caller's n, the problem is
  // embedded as [G 0; 0 I], [a; 0], [C; 0] -- the factor is [L 0; 0 I], J starts as [J 0; 0 I], every d ends in nr..n-1
  // zeros, and the extra rotations of an addition act on (0, 0) pairs (identities) or on (d_{nr-1}, 0), which at most
  // negates the LAST column of the free part of J together with its d entry -- a sign that cancels in every product it
  // enters (z, the next rotation) and never reaches R, x, u or f.  So the padded run reproduces the unpadded one bit for
  // bit (up to the sign of exact zeros), and x, u, status are written back in the caller's sizes.  The warm start is
  // embedded the same way: N gets zero rows, so do B = L^-1 N and every Householder vector of its QR, and Q leaves the
  // identity block alone.
  const int nr = EMB ? A.n : n;
  const int ld = NT ? (NT | 1) : A.ld;
  const int m = A.m, nb = A.nb;
  const int mt = m + nb;
  constexpr int NR = NT ? NT : 1;

  // LDS carve-up (must match batch_lds_bytes below)
  double * J = smem;
  double * R = J + n * ld;
  double * xs = R + n * ld;
  double * zs = xs + n;
  double * ds = zs + n;
  double * nps = ds + n;
  double * us = nps + n;
  double * rs = us + (n + 2);
  double * tail = rs + (n + 2);
  // CMODE 2: one n-vector per scanning wave: the normal of that wave's most violated constraint, written by its owner
  // thread during the (overlapped) scan, so that nothing has to be published once the winner among the waves is known
  double * npw = tail;
  if(CMODE == 2) tail = npw + 3 * n;
  // CMODE 2: the state of the (sliced) speculative coefficient chain of wave 3 between its slices: seeds + radius, one
  // value per lane.  In registers it would be live across every other phase of the pass (all waves share one register
  // allocation) and push column entries of the scanning waves into scratch.
  double * gsc = tail;
  if(CMODE == 2) tail = gsc + 6 * 64;
  // gsc: [0, 2n + 1) the (c, s) pairs of an addition (csb, role-specialised loop); [128, 192) / [192, 256) bl / bu of the LDS-staged
  // extra columns when there are at most 64 of them (round 4: the wave that scans them fetched both from global memory in EVERY
  // scan -- two dependent loads, their pointers reloaded from scratch first, on the path every pass that adds waits for)
  double * ebl = gsc + 128;
  double * ebu = gsc + 192;
  // WARM: N -> B = L^-1 N -> its QR factors (n x q), b_act, Householder coefficients
  double * Bw = tail;
  double * bact = Bw + (WARM ? n * ld : 0);
  double * hco = bact + (WARM ? n : 0);
  tail = hco + (WARM ? n : 0);
  double * Cs = tail;
  double * bls = tail;
  double * bus = tail;
  if(CMODE == 1)
  {
    bls = Cs + (size_t)m * ld;
    bus = bls + m;
    tail = bus + m;
  }
  // CMODE 2: constraint i < 192 lives in the registers of thread i (waves 0-2; wave 3 owns no constraint: it runs
  // the Givens coefficient chain of an addition, speculatively, beside the other waves' work); the m - 192 remaining
  // columns are staged in LDS and scanned by wave 1 in a second, short pass
  constexpr int CREG_MAX = WG - 64;
  const int ne = (CMODE == 2 && m > CREG_MAX) ? (m - CREG_MAX) : 0;
  double * Ce = tail;
  if(CMODE == 2) tail = Ce + (size_t)ne * ld;
  Ctl * ctl = reinterpret_cast<Ctl *>(tail);
  int * act = reinterpret_cast<int *>(ctl + 1);
  unsigned char * st = reinterpret_cast<unsigned char *>(act + (n + 2));

  const double * Gg = A.G + (size_t)b * nr * nr;
  const double * ag = A.a + (size_t)b * nr;
  const double * Cg = A.C + (size_t)b * nr * m;
  const double * blg = A.bl + (size_t)b * m;
  const double * bug = A.bu + (size_t)b * m;
  const double * xlg = nb ? A.xl + (size_t)b * nb : nullptr;
  const double * xug = nb ? A.xu + (size_t)b * nb : nullptr;

  // ---------------------------------------------------------------- C column -> registers (CMODE 2)
  double creg[NR];
  double my_bl = 0.0, my_bu = 0.0;
  const int myc = (CMODE == 2 && tid < CREG_MAX) ? tid : -1; // constraint held in this thread's registers (waves 0-2)
  // At three workgroups per CU (168 VGPRs) the columns are fetched AFTER the initial factorisation: live across it they
  // push hipcc into spilling column entries for the whole kernel (reloaded one by one inside every scan).  The exposed
  // load latency is paid once per solve.
  constexpr bool LATE_C = (CMODE == 2 && NT == 50 && !WARM);
  // WS: role-specialised main loop (cold start, C in registers).  Wave 3 -- which owns no constraint column -- is the MAIN
  // wave: it runs every serial phase of a pass (d = J^T n+, the substitution, the ratio test, the step, the sweep of J, a
  // removal) in code of its own in which the 100 column registers of the other waves are not live, so that its loops can
  // keep whole columns / rows of J in flight.  Waves 0-2 are HELPERS: they own the columns of C (the scan), and beside it
  // wave 0 runs the Givens coefficient chain of a possible addition from the moment d exists -- the whole chain and the
  // (c, s) pairs, speculatively --, wave 1 z and z.z, wave 2 the two products with n+.  No workgroup barrier inside the
  // loop: the roles meet through sequence-numbered words in LDS (release/acquire), so a wave never waits for work it does
  // not depend on.
  constexpr bool WS = (CMODE == 2) && !WARM && !BOX && (NT >= 40); // (below 40 unknowns a pass is too short for the hand-offs to pay)
  auto load_columns = [&]() __attribute__((always_inline)) {
    if(m == 0) return; // no general constraint: C may be a null pointer (the loads below are unconditional)
    const bool own = myc >= 0 && myc < m;
    const double * ci = Cg + (size_t)(own ? myc : 0) * nr;
    if(nr == NR && (NR % 2 == 0) && ((reinterpret_cast<size_t>(Cg) & 15) == 0)) // a column is NR*8 bytes: 16-byte loads when aligned
    {
      const double2 * c2 = reinterpret_cast<const double2 *>(ci);
#pragma unroll
      for(int k = 0; k < NR / 2; ++k)
      {
        const double2 v = c2[k];
        creg[2 * k] = v.x; // (a thread that owns no constraint never reads creg)
        creg[2 * k + 1] = v.y;
      }
    }
    else
    {
#pragma unroll
      for(int k = 0; k < NR; ++k) creg[k] = (k < nr) ? ci[k] : 0.0; // (zero rows of an embedded smaller problem)
    }
    my_bl = own ? blg[myc] : 0.0;
    my_bu = own ? bug[myc] : 0.0;
  };
  if(CMODE == 2)
  {
    if(!LATE_C && !WS) load_columns();
    for(int e = tid; e < ne * n; e += WG)
    {
      int k = e % n, i = e / n;
      Ce[(size_t)i * ld + k] = (k < nr) ? Cg[(size_t)(CREG_MAX + i) * nr + k] : 0.0;
    }
    if(ne > 0 && ne <= 64 && tid < ne)
    {
      ebl[tid] = blg[CREG_MAX + tid];
      ebu[tid] = bug[CREG_MAX + tid];
    }
  }

  // ---------------------------------------------------------------- load G (coalesced) -> R region, status = 0
  for(int e = tid; e < n * n; e += WG)
  {
    int i = e % n, j = e / n;
    if constexpr(BOX)
    {
      R[i + j * ld] = 0.0; // (R.col(q).head(q).setZero() for whatever q comes out, BoxAndSingleConstraintSolver.cpp:69,80)
      J[i + j * ld] = 0.0; // J.setZero() (:61)
    }
    else
      R[i + j * ld] = (i < nr && j < nr) ? Gg[i + (size_t)j * nr] : ((i == j) ? 1.0 : 0.0);
  }
  for(int i = tid; i < mt; i += WG) st[i] = 0;
  if(tid < n) xs[tid] = 0.0;
  if(tid == 0)
  {
    ctl->info = -1;
    ctl->f = 0.0;
    ctl->eq_next = NONE;
    ctl->cnt = 0;
    ctl->flag = 0;
    // (hand-off words of the role-specialised loop: zeroed HERE, before anybody can post -- with the overlapped prologue the helpers
    // post their first scan while MAIN is still inside J = L^-T)
    ctl->fd = 0; ctl->fz = 0; ctl->fack0 = 0; ctl->fack2 = 0; ctl->fcs = 0; ctl->fdr = 0; ctl->fstep = 0; ctl->dec = 0; ctl->fx0 = 0;
    ctl->fscan[0] = 0; ctl->fscan[1] = 0; ctl->fscan[2] = 0;
    ctl->wit = 0; ctl->wterm = T_SUCCESS; ctl->wq = 0;
  }
  __syncthreads();
  STAMP(0)
  if(PROF && lane == 0 && (wave == 3 || wave == 0))
  {
    // where the hardware put this wave: HW_ID (SIMD [5:4], CU [11:8], SH [12], SE [15:13]) and XCC_ID -- slots 14 (wave 3) / 15 (wave 0)
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    prof_lds[wave == 3 ? 14 : 15] = ((unsigned long long)xcc << 32) | hw;
  }

  // ---------------------------------------------------------------- K1: LLT in LDS (wave 0), while waves 1-3 stage C
  double * L = R;
  if(BOX && wave == 0)
  {
    // BoxAndSingleConstraintSolver::init_ (:46-94), lane = variable: the start point is x0 clamped to the box, every clamped
    // variable's bound is active (in index order) with multiplier |x0_i - bound_i|, J is the permutation that puts the
    // clamped coordinates first (free coordinate number k goes to column n-1-k) and R = diag(+1 lower / -1 upper)
    const bool in = lane < n;
    const double x0 = in ? ag[lane] : 0.0, lo = in ? xlg[lane] : 0.0, hi = in ? xug[lane] : 0.0;
    const bool blo = in && x0 < lo;
    const bool bup = in && !blo && x0 > hi;
    const unsigned long long mk = __ballot(blo || bup);
    const int pos = __popcll(mk & ((1ull << lane) - 1ull));
    const double xv = blo ? lo : (bup ? hi : x0);
    const double uv = blo ? (xv - x0) : (x0 - xv);
    if(in)
    {
      xs[lane] = xv;
      const int col = (blo || bup) ? pos : (n - 1 - (lane - pos));
      J[lane + col * ld] = 1.0;
      if(blo || bup)
      {
        R[pos + pos * ld] = blo ? 1.0 : -1.0;
        us[pos] = uv;
        act[pos] = m + lane; // (:72 "i + 1": one general constraint in front of the bounds)
        st[m + lane] = (unsigned char)(blo ? LOWER_BOUND : UPPER_BOUND);
      }
    }
    double f = 0.0; // f_ += 0.5 * u[q] * u[q], in activation order: one chain, every lane the same
    for(unsigned long long rest = mk; rest; rest &= rest - 1)
    {
      const double uq = bcast_lane(uv, __builtin_ctzll(rest));
      f = FMA(0.5 * uq, uq, f);
    }
    if(lane == 0)
    {
      ctl->f = f;
      ctl->nact = __popcll(mk);
    }
  }
  else if(wave == 0)
  {
    // Left-looking by blocks of LB columns, lane = row: the LB accumulators of a row stay in registers while the
    // wave walks the columns k < j0 already factored (one lane-varying + LB broadcast LDS reads feed LB independent
    // fma chains per k), then the LB x LB diagonal block and the panel below it are finished from registers with
    // readlane.  Per element the updates still arrive in ascending k (CANON C1): bit-identical to the unblocked form.
    constexpr int LB = 8;
    int info = -1;
    const int rl = (lane < n) ? lane : (n - 1); // lanes >= n shadow the last row (results discarded)
    if constexpr(NT >= 16)
    {
      double a[NR];
#pragma unroll
      for(int c = 0; c < NR; ++c) a[c] = L[rl + c * ld];
      STAMP(18)
      const double p0 = bcast_lane(a[0], 0), l0 = __builtin_sqrt(opaque_v(p0));
      llt_reg_columns<0, NR>(a, info, lane, L, ld, p0, l0, a[0] / l0);
      STAMP(19)
    }
    else
    for(int j0 = 0; j0 < n && info < 0; j0 += LB)
    {
      double acc[LB];
#pragma unroll
      for(int c = 0; c < LB; ++c) acc[c] = L[rl + ((j0 + c < n) ? (j0 + c) : (n - 1)) * ld];
      // rows j0 .. j0+LB-1 of column k are LB consecutive doubles: one base address, immediate offsets.  In the last
      // block j0+c may pass n: the read then lands on the first rows of column k+1 (k+1 <= j0 < n, still inside
      // the factor's LDS region) and the accumulator it feeds is never used.
      // (loads of two steps issued together, then the fmas: the scheduling barrier keeps hipcc from interleaving
      // them one LDS round trip at a time to save registers)
      int k = 0;
#pragma unroll 1
      for(; k + 2 <= j0; k += 2)
      {
        double li[2], lj[2][LB];
#pragma unroll
        for(int u = 0; u < 2; ++u)
        {
          li[u] = L[rl + (k + u) * ld];
#pragma unroll
          for(int c = 0; c < LB; ++c) lj[u][c] = L[j0 + c + (k + u) * ld];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for(int u = 0; u < 2; ++u)
#pragma unroll
          for(int c = 0; c < LB; ++c) acc[c] = FMA(-li[u], lj[u][c], acc[c]);
        __builtin_amdgcn_sched_barrier(0);
      }
      static_assert(LB % 2 == 0, "j0 is a multiple of LB: the two-step loop leaves no remainder");
      STAMP(18)
#pragma unroll
      for(int c = 0; c < LB; ++c)
      {
        const int j = j0 + c;
        if(j < n && info < 0)
        {
          const double piv = bcast_lane(acc[c], j);
          if(piv <= 0.0)
            info = j;
          else
          {
            const double ljj = __builtin_sqrt(opaque_v(piv));
            const double lij = (lane == j) ? ljj : acc[c] / ljj;
            if(lane >= j && lane < n) L[lane + j * ld] = lij;
#pragma unroll
            for(int c2 = c + 1; c2 < LB; ++c2)
              if(j0 + c2 < n) acc[c2] = FMA(-lij, bcast_lane(lij, j0 + c2), acc[c2]);
          }
        }
      }
      STAMP(19)
    }
    if(lane == 0) ctl->info = info;
  }
  else
  {
    if(CMODE == 1)
    {
      for(int e = tid - 64; e < n * m; e += WG - 64)
      {
        int k = e % n, i = e / n;
        Cs[(size_t)i * ld + k] = Cg[e];
      }
      for(int i = tid - 64; i < m; i += WG - 64)
      {
        bls[i] = blg[i];
        bus[i] = BOX ? __builtin_inf() : bug[i]; // (box solver: bu_[0] = +inf, BoxAndSingleConstraintSolver.cpp:33)
      }
    }
    if constexpr(WARM)
    {
      // processInitialActiveSet, decision per constraint (experimental/GoldfarbIdnaniSolver.cpp:309-371): data
      // equalities win over the guess; guesses on infinite bounds, FIXED guesses and ill-typed entries are ignored
      const int * asg = A.as_in ? A.as_in + (size_t)b * mt : nullptr;
      int neq = 0;
      for(int i = tid - 64; i < mt; i += WG - 64)
      {
        int s = INACTIVE;
        const int g = asg ? asg[i] : INACTIVE;
        if(i < m)
        {
          const double lo = blg[i], hi = bug[i];
          if(lo == hi)
            s = EQUALITY;
          else if(g == LOWER && !(lo < -A.big_bnd))
            s = LOWER;
          else if(g == UPPER && !(hi > A.big_bnd))
            s = UPPER;
          else if(g == EQUALITY)
            s = EQUALITY; // :357-371 only filters FIXED: an EQUALITY guess is taken as is
        }
        else
        {
          const double lo = xlg[i - m], hi = xug[i - m];
          if(lo == hi)
            s = FIXED;
          else if(g == LOWER_BOUND && !(lo < -A.big_bnd))
            s = LOWER_BOUND;
          else if(g == UPPER_BOUND && !(hi > A.big_bnd))
            s = UPPER_BOUND;
        }
        st[i] = (unsigned char)s;
        neq += (s == EQUALITY || s == FIXED);
      }
      if(neq) atomicAdd(&ctl->cnt, neq);
    }
    else if constexpr(!BOX) // (the box solver's init_ replaces initActiveSet as well: no pre-activation of xl == xu)
    {
      // equalities / fixed variables present?  (GoldfarbIdnaniSolver.cpp:268-287)
      int first = NONE, cnt = 0;
      for(int i = tid - 64; i < mt; i += WG - 64)
      {
        bool eq = (i < m) ? (blg[i] == bug[i]) : (xlg[i - m] == xug[i - m]);
        if(eq)
        {
          ++cnt;
          if(i < first) first = i;
        }
      }
      if(first != NONE)
      {
        atomicMin(&ctl->eq_next, first);
        atomicAdd(&ctl->cnt, cnt);
      }
    }
  }
  __syncthreads();
  STAMP(1)

  if constexpr(WARM)
  {
    // activation ORDER (:319-371): bounds first, then general constraints, each by increasing index; if more than n
    // came out, inequality entries are dropped from the back until n remain (:373-383) -- i.e. every equality stays
    // and the first n - #eq inequality entries (in list order) stay
    if(wave == 0)
    {
      const int neq = ctl->cnt;
      const int budget = nr - neq; // (the caller's n)
      int base = 0, nrank = 0;
      const unsigned long long lt = (1ull << lane) - 1ull;
      for(int part = 0; part < 2; ++part)
      {
        const int lo = (part == 0) ? m : 0, hi = (part == 0) ? mt : m;
        for(int c0 = lo; c0 < hi; c0 += 64)
        {
          const int i = c0 + lane;
          const int s = (i < hi) ? st[i] : INACTIVE;
          const bool iseq = (s == EQUALITY || s == FIXED);
          const bool isne = (s != INACTIVE) && !iseq;
          const unsigned long long mne = __ballot(isne);
          const int rank = nrank + __popcll(mne & lt);
          const bool keep = iseq || (isne && rank < budget);
          const unsigned long long mk = __ballot(keep);
          const int pos = base + __popcll(mk & lt);
          if(keep && pos < nr) act[pos] = i;
          if(isne && !keep && budget >= 0) st[i] = INACTIVE;
          base += __popcll(mk);
          nrank += __popcll(mne);
        }
      }
      if(lane == 0)
      {
        ctl->nact = (base < nr) ? base : nr;
        ctl->over = (neq > nr);
      }
    }
    __syncthreads();
  }

  int term = T_SUCCESS;
  int it = 0;
  int q = 0;
  bool ovl = false; // overlapped prologue (see K2 below)

  if(WARM && ctl->over)
  {
    term = T_OVERCONSTRAINED; // reported before the factorisation is looked at (:70-71)
  }
  else if(ctl->info >= 0)
  {
    term = T_NON_POS_HESSIAN;
  }
  else
  {
    // optional: hand the factor back (the reference leaves it in G's lower triangle)
    if(A.Lout)
    {
      double * Lg = A.Lout + (size_t)b * nr * nr;
      for(int e = tid; e < nr * nr; e += WG)
      {
        int i = e % nr, j = e / nr;
        if(i >= j) Lg[e] = L[i + j * ld];
      }
    }
    // ---------------------------------------------------------------- K2 (wave 1): J = L^-T ; K3 (wave 0): x0, f0
    // Overlapped prologue (role-specialised kernels, no data equalities -- the pre-activation below needs every wave): J = L^-T runs on
    // the MAIN wave (wave 3, which owns no constraint column), x0 on wave 0, and there is NO barrier behind them: the helpers go
    // straight on to fetch their columns and, once wave 0 has posted x0 (fx0), to the first scan -- all of it beside MAIN's J, which
    // is the longest of the three.
    ovl = WS && uniform(ctl->eq_next) == NONE && !(A.tune & 8);
    if constexpr(BOX)
    {
      q = uniform(ctl->nact); // J, R, x, u, f, the active list: all set by the closed form above
    }
    else if(wave == (ovl ? 3 : 1))
    {
      // J = L^-T by blocks of JB rows from the bottom, lane = column c: J_ic = (delta_ic - sum_{k = c .. i+1,
      // descending} L_ki J_kc) / L_ii.  The JB accumulators of a column stay in registers while the wave walks the
      // rows k below the block; rows k > c contribute exact zeros (J_kc = +0), which leaves every accumulator bit
      // for bit as the triangular loop would (x + (+-0) = x, and an accumulator is never -0 here).
      constexpr int JB = 8;
      const int cl = (lane < n) ? lane : (n - 1);
      if constexpr(NT >= 16)
      {
        double acc[NR];
#pragma unroll
        for(int i = 0; i < NR; ++i) acc[i] = (i == lane) ? 1.0 : 0.0;
        jinv_reg_rows<NR - 1, NR>(acc, lane, L, J, ld);
      }
      else
      for(int i0 = ((n - 1) / JB) * JB; i0 >= 0; i0 -= JB)
      {
        double acc[JB];
#pragma unroll
        for(int r = 0; r < JB; ++r) acc[r] = (i0 + r == lane) ? 1.0 : 0.0;
        // (the loop is empty for the bottom block, the only one whose rows can pass n: i0 + r < n below)
        int k = n - 1;
#pragma unroll 1
        for(; k - 1 >= i0 + JB; k -= 2)
        {
          double jk[2], lk[2][JB];
#pragma unroll
          for(int u = 0; u < 2; ++u)
          {
            jk[u] = J[(k - u) + cl * ld];
#pragma unroll
            for(int r = 0; r < JB; ++r) lk[u][r] = L[(k - u) + (i0 + r) * ld];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for(int u = 0; u < 2; ++u)
#pragma unroll
            for(int r = 0; r < JB; ++r) acc[r] = FMA(-lk[u][r], jk[u], acc[r]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if(k >= i0 + JB) // odd count: one more step
        {
          const double jk = J[k + cl * ld];
#pragma unroll
          for(int r = 0; r < JB; ++r) acc[r] = FMA(-L[k + (i0 + r) * ld], jk, acc[r]);
        }
#pragma unroll
        for(int r = JB - 1; r >= 0; --r)
        {
          const int i = i0 + r;
          if(i < n)
          {
            const double v = acc[r] / L[i + i * ld];
            if(lane < n) J[i + lane * ld] = v;
#pragma unroll
            for(int r2 = 0; r2 < r; ++r2) acc[r2] = FMA(-L[i + (i0 + r2) * ld], v, acc[r2]);
          }
        }
      }
    }
    else if constexpr(WARM)
    {
      // waves 0, 2, 3: N (+-C columns / +-unit vectors) and b_act in activation order (:393-418); a -> LDS
      const int t3 = (wave == 0) ? lane : (tid - 64);
      const int qn = ctl->nact;
      for(int e = t3; e < qn * n; e += WG - 64)
      {
        const int j = e / n, i = e % n;
        const int idx = act[j];
        const int sj = st[idx];
        double v;
        if(idx < m)
        {
          const double c = (i < nr) ? Cg[(size_t)idx * nr + i] : 0.0;
          v = (sj == UPPER) ? -c : c;
        }
        else
          v = (i == idx - m) ? ((sj == UPPER_BOUND) ? -1.0 : 1.0) : 0.0;
        Bw[i + j * ld] = v;
      }
      for(int j = t3; j < qn; j += WG - 64)
      {
        const int idx = act[j];
        const int sj = st[idx];
        double bv;
        if(idx < m)
          bv = (sj == UPPER) ? -bug[idx] : blg[idx];
        else
          bv = (sj == UPPER_BOUND) ? -xug[idx - m] : xlg[idx - m];
        bact[j] = bv;
      }
      if(t3 < n) nps[t3] = (t3 < nr) ? ag[t3] : 0.0;
    }
    else if(wave == 0)
    {
      const int rl = (lane < n) ? lane : 0;
      double xi = (lane < nr) ? ag[lane] : 0.0;
      const double ai = xi;
      double dg = L[rl + rl * ld];
      dg = (lane < n) ? dg : 1.0;
      const double rinv = recip_refined(dg);
      const bool den_safe = __all(div_safe_den(dg));
      // First with hoisted reciprocals and the lean step of wave_fwd / wave_bwd (device_common.h): the quotient of lane k formed
      // by every lane on its own entry and read with one v_readlane pair, the numerator left in its lane, all quotients of a
      // substitution formed after its last step (so the numerators are checked once per substitution, not once per step).
      bool lean_ok = false;
      {
        double col = L[rl]; // column 0
        for(int k = 0; k < n; ++k) // x = L.solve(a)
        {
          const double coln = L[rl + ((k + 1 < n) ? (k + 1) : k) * ld];
          const double v = bcast_lane_s(div_pre(xi, dg, rinv), k);
          const double upd = FMA(-col, v, xi);
          xi = (lane > k && lane < n) ? upd : xi;
          col = coln;
        }
        const double pre1 = xi;
        xi = div_pre(pre1, dg, rinv);
        col = L[(n - 1) + rl * ld]; // row n-1 of L = column n-1 of L^T
        for(int k = n - 1; k >= 0; --k) // L^T.solveInPlace(x)
        {
          const double coln = L[((k > 0) ? (k - 1) : 0) + rl * ld];
          const double v = bcast_lane_s(div_pre(xi, dg, rinv), k);
          const double upd = FMA(-col, v, xi);
          xi = (lane < k) ? upd : xi;
          col = coln;
        }
        const double pre2 = xi;
        xi = div_pre(pre2, dg, rinv);
        lean_ok = den_safe && __all(div_safe_num(pre1) && div_safe_num(pre2));
      }
      for(int pass = 1; pass < 2 && !lean_ok; ++pass)
      {
        // (only if an operand left the range in which the hoisted reciprocal is provably the compiler's quotient): plain a/b
        bool safe = den_safe;
        xi = ai;
        double col = L[rl]; // column 0
        for(int k = 0; k < n; ++k) // x = L.solve(a), branch-free body
        {
          const double coln = L[rl + ((k + 1 < n) ? (k + 1) : k) * ld];
          const double num = opaque_v(bcast_lane(xi, k));
          const double den = bcast_lane(dg, k);
          const double v = (pass == 0) ? div_pre(num, den, bcast_lane(rinv, k)) : num / opaque_v(den);
          safe = safe && div_safe_num(num);
          const double upd = FMA(-col, v, xi);
          xi = (lane > k && lane < n) ? upd : xi;
          xi = (lane == k) ? v : xi;
          col = coln;
        }
        col = L[(n - 1) + rl * ld]; // row n-1 of L = column n-1 of L^T
        for(int k = n - 1; k >= 0; --k) // L^T.solveInPlace(x)
        {
          const double coln = L[((k > 0) ? (k - 1) : 0) + rl * ld];
          const double num = opaque_v(bcast_lane(xi, k));
          const double den = bcast_lane(dg, k);
          const double v = (pass == 0) ? div_pre(num, den, bcast_lane(rinv, k)) : num / opaque_v(den);
          safe = safe && div_safe_num(num);
          const double upd = FMA(-col, v, xi);
          xi = (lane < k) ? upd : xi;
          xi = (lane == k) ? v : xi;
          col = coln;
        }
        if(safe) break;
      }
      xi = -xi;
      if(lane < n) xs[lane] = xi;
      // f = 0.5 * a.dot(x): one in-order chain, operands fetched with readlane
      double acc = 0.0;
      for(int k = 0; k < n; ++k) acc = FMA(bcast_lane(ai, k), bcast_lane(xi, k), acc);
      if(lane == 0) ctl->f = 0.5 * acc;
      if(ovl)
      {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if(lane == 0) __atomic_store_n(&ctl->fx0, 1, __ATOMIC_RELAXED);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory");
      }
    }
    if(!ovl) __syncthreads();
    STAMP(2)
    if(CMODE == 2 && LATE_C && !WS) load_columns();

    const double big = A.big_bnd;
    constexpr bool SPEC = (CMODE == 2); // speculative Givens coefficient chain on wave 3 (which holds no constraint)
    constexpr int SPEC_A0 = 24, SPEC_B = 20; // steps run beside trsv/ratio (plus q/2) and beside the primal/dual update
    int p = -1, pstat = INACTIVE;
    int addno = 0; // number of constraints added so far: the hand-off token of add_chain / add_apply

    // ======================================================================================================
    // device-local lambdas (all inlined); barriers only ever appear in the kernel body, never inside them
    // ======================================================================================================

    // K4 for the calling thread's share: candidate (value < 0, index, status); `excl` is treated as active
    auto scan_local = [&](int excl, double & bval, int & bidx, int & bstat) __attribute__((always_inline)) {
      bval = 0.0;
      bidx = NONE;
      bstat = INACTIVE;
      if(CMODE == 2)
      {
        if(myc >= 0 && myc < m && st[myc] == INACTIVE && myc != excl)
        {
          double cx = 0.0;
          // static register indices need full unrolling; x is fetched ten values at a time so that the LDS
          // latency is paid once per group, and the scheduling barrier keeps the groups from being merged
          // (all 50 loads hoisted => spills)
#pragma unroll
          for(int k0 = 0; k0 < NR; k0 += 10)
          {
            double xv[10];
#pragma unroll
            for(int u = 0; u < 10; ++u)
              if(k0 + u < NR) xv[u] = xs[k0 + u];
#pragma unroll
            for(int u = 0; u < 10; ++u)
              if(k0 + u < NR) cx = FMA(creg[k0 + u], xv[u], cx);
            __builtin_amdgcn_sched_barrier(0);
          }
          double sl = cx - my_bl;
          double su = my_bu - cx;
          if(sl < 0.0)
          {
            bval = sl;
            bidx = myc;
            bstat = LOWER;
          }
          else if(su < 0.0)
          {
            bval = su;
            bidx = myc;
            bstat = UPPER;
          }
        }
      }
      else
      {
        for(int i = tid; i < m; i += WG)
        {
          if(st[i] != INACTIVE || i == excl) continue;
          const double * ci = (CMODE == 1) ? (Cs + (size_t)i * ld) : (Cg + (size_t)i * n);
          double cx = chain_up(ci, 1, xs, 1, 0, n, 0.0);
          double sl = cx - ((CMODE == 1) ? bls[i] : blg[i]);
          if(sl < bval)
          {
            bval = sl;
            bidx = i;
            bstat = LOWER;
          }
          else
          {
            double su = ((CMODE == 1) ? bus[i] : bug[i]) - cx;
            if(su < bval)
            {
              bval = su;
              bidx = i;
              bstat = UPPER;
            }
          }
        }
      }
      for(int i = tid; i < nb; i += (CMODE == 2) ? CREG_MAX : WG)
      {
        if(CMODE == 2 && tid >= CREG_MAX) break; // CMODE 2: wave 3 takes no part in the scan
        if(st[m + i] != INACTIVE || (m + i) == excl) continue;
        double xi = xs[i];
        double sl = xi - xlg[i];
        if(sl < bval)
        {
          bval = sl;
          bidx = m + i;
          bstat = LOWER_BOUND;
        }
        else
        {
          double su = xug[i] - xi;
          if(su < bval)
          {
            bval = su;
            bidx = m + i;
            bstat = UPPER_BOUND;
          }
        }
      }
    };

    // wave-level reduction of the scan candidates + hand-off of the wave's winner.  CMODE 2: waves 0-2 publish slots
    // 0-2; slot 3 (wave 3 holds no constraint) carries the winner among the LDS-staged extra columns, which wave 1
    // scans in a second pass (lane e = extra column e).
    auto scan_publish = [&](int excl, double bval, int bidx, int bstat) __attribute__((always_inline)) {
      if(CMODE == 2)
      {
        if(wave == 3) return;
        if(nb == 0)
        {
          if(bidx == NONE) bval = 1.0; // candidates are < 0
          wave_argmin_monotone(bval, bidx, bstat);
        }
        else
          wave_argmin(bval, bidx, bstat);
        if(lane == 0)
        {
          ctl->red_val[wave] = bval;
          ctl->red_idx[wave] = bidx;
          ctl->red_stat[wave] = bstat;
        }
        if(myc == bidx && myc < m) // this wave's winner: n+ and its bound into the wave's slot
        {
          double * slot = npw + wave * n;
#pragma unroll
          for(int k = 0; k < NR; ++k) slot[k] = creg[k];
          ctl->bndw[wave] = (bstat == LOWER) ? my_bl : my_bu;
        }
        if(wave == 1)
        {
          // the LDS-staged columns, 64 at a time (lane = column inside the pass); a lane keeps its first best one
          // (strict <: a later pass only wins with a smaller slack), then the general arg-min picks the first index
          // among equal values across lanes
          double ev = 1.0;
          int ei = NONE, es = INACTIVE;
          for(int e0 = 0; e0 < ne; e0 += 64)
          {
            const int i = CREG_MAX + e0 + lane;
            if(e0 + lane < ne && st[i] == INACTIVE && i != excl)
            {
              const double cx = chain_up(Ce + (size_t)(e0 + lane) * ld, 1, xs, 1, 0, n, 0.0);
              const double sl = cx - blg[i];
              const double su = bug[i] - cx;
              if(sl < 0.0)
              {
                if(sl < ev)
                {
                  ev = sl;
                  ei = i;
                  es = LOWER;
                }
              }
              else if(su < 0.0 && su < ev)
              {
                ev = su;
                ei = i;
                es = UPPER;
              }
            }
          }
          if(ne > 64)
            wave_argmin(ev, ei, es);
          else if(ne > 0)
            wave_argmin_monotone(ev, ei, es);
          if(lane == 0)
          {
            ctl->red_val[3] = ev;
            ctl->red_idx[3] = ei;
            ctl->red_stat[3] = es;
          }
        }
        return;
      }
      wave_argmin(bval, bidx, bstat);
      if(lane == 0)
      {
        ctl->red_val[wave] = bval;
        ctl->red_idx[wave] = bidx;
        ctl->red_stat[wave] = bstat;
      }
    };

    // after a barrier: every thread combines the four wave winners (first index wins ties)
    auto scan_combine = [&](int & pp, int & ps) {
      // all twelve words are fetched up front (one LDS round trip), then selected in registers
      double rv[4];
      int ri[4], rst[4];
#pragma unroll
      for(int w = 0; w < 4; ++w)
      {
        rv[w] = ctl->red_val[w];
        ri[w] = ctl->red_idx[w];
        rst[w] = ctl->red_stat[w];
      }
      double v = rv[0];
      int ix = ri[0], sx = rst[0];
#pragma unroll
      for(int w = 1; w < 4; ++w)
      {
        const bool take = (ri[w] != NONE) && (ix == NONE || rv[w] < v || (rv[w] == v && ri[w] < ix));
        v = take ? rv[w] : v;
        ix = take ? ri[w] : ix;
        sx = take ? rst[w] : sx;
      }
      pp = uniform((ix == NONE) ? -1 : ix);
      ps = uniform(sx);
    };

    // n+ of the selected general constraint -> LDS vector np (a pointer into staged C when CMODE == 1) and its bound
    // value.  (always_inline: a lambda that is not inlined captures creg[] by address and pushes it to scratch.)
    auto publish_np = [&](int pp, int ps, bool scanned) __attribute__((always_inline)) {
      const bool lower = (ps == LOWER || ps == EQUALITY || ps == LOWER_BOUND || ps == FIXED);
      if(WS && !scanned) // (equality pre-activation: the column registers are not loaded yet: from memory, n threads)
      {
        if(pp < CREG_MAX && pp < m)
        {
          if(tid < n) npw[(pp >> 6) * n + tid] = (tid < nr) ? Cg[(size_t)pp * nr + tid] : 0.0;
          if(tid == 0) ctl->bndw[pp >> 6] = lower ? blg[pp] : bug[pp];
        }
      }
      else if(CMODE == 2 && !scanned) // (a scan already left n+ and b in the owner wave's slot: scan_publish)
      {
        if(myc == pp && myc < m) // the owner thread (extra columns are used in place: np_ptr)
        {
          double * slot = npw + (pp >> 6) * n;
#pragma unroll
          for(int k = 0; k < NR; ++k) slot[k] = creg[k];
          ctl->bndw[pp >> 6] = lower ? my_bl : my_bu;
        }
      }
      if(CMODE == 0 && pp < m && tid < n) nps[tid] = Cg[(size_t)pp * n + tid];
    };
    // b of the selected constraint: from LDS when its owner thread published it, else from memory
    // (keeping every other case out of publish_np matters: a larger lambda there stops hipcc from keeping creg[] in
    // registers -- it lands in scratch and the kernel runs 2x slower)
    auto bound_of = [&](int pp, int ps) -> double {
      const bool lower = (ps == LOWER || ps == EQUALITY || ps == LOWER_BOUND || ps == FIXED);
      if(CMODE == 2 && pp < CREG_MAX && pp < m) return ctl->bndw[pp >> 6];
      if(pp >= m) return lower ? xlg[pp - m] : xug[pp - m];
      if(CMODE == 1) return lower ? bls[pp] : bus[pp];
      return lower ? blg[pp] : bug[pp];
    };
    auto np_ptr = [&](int pp) -> const double * {
      if(CMODE == 1) return Cs + (size_t)((pp < m) ? pp : 0) * ld;
      if(CMODE == 2 && pp >= CREG_MAX && pp < m) return Ce + (size_t)(pp - CREG_MAX) * ld;
      if(CMODE == 2 && pp < m) return npw + (pp >> 6) * n; // constraint i is held by thread i: wave i/64
      return nps;
    };

    // K5 (wave 0): d = +-J^T n+ (general) or +-row of J (bound)
    auto compute_d = [&](int pp, int ps) {
      if(lane < n)
      {
        double dj;
        if(pp < m)
          dj = chain_up(J + lane * ld, 1, np_ptr(pp), 1, 0, n, 0.0);
        else
          dj = J[(pp - m) + lane * ld];
        if(ps == UPPER || ps == UPPER_BOUND) dj = -dj;
        ds[lane] = dj;
      }
    };

    // K7 + ratio test (wave 0).  The loop body is branch-free on purpose: conditional LDS loads make hipcc wait
    // with lgkmcnt(0) right after issuing the prefetch (measured 267 cycles per step instead of ~130).
    double tr_t1 = 0.0; // (t1, l) of the last trsv_ratio, in every lane of the wave that ran it
    int tr_l = 0;
    double tr_r = 0.0, tr_u = 0.0, tr_uq = 0.0; // its r_k and the u_k it loaded (lane k < q), and u_q (every lane): WS keeps them for the update
    auto trsv_ratio = [&]() {
      const int rl = (lane < n) ? lane : 0;
      double rk = ds[rl];
      rk = (lane < q) ? rk : 0.0;
      double dg = R[rl + rl * ld];
      dg = (lane < q) ? dg : 1.0;
      const double rk0 = rk;
      const int sk_l = (lane < q) ? st[lane] : INACTIVE; // operands of the ratio test below
      const double u_l = us[rl];
      tr_uq = us[q]; // (the pending multiplier, a broadcast read in the same round trip; only the WS update uses it)
      const double rinv = recip_refined(dg); // every pivot's reciprocal at once (lane = pivot)
      // Columns are fetched FOUR steps ahead (the LDS is shared with 11 other waves: a one-step prefetch still exposes
      // most of its latency on this chain); the loop is unrolled by four so that the look-ahead registers need no
      // rotation.  Each lane keeps the numerator of its own division (`pre`): the range check that licenses the
      // hoisted reciprocals is made once, after the loop, instead of once per step.
      auto colat = [&](int k) { return R[rl + ((k > 0) ? k : 0) * ld]; };
      // One step: the quotient of lane k is formed by every lane on its own entry (no broadcast of pivot and reciprocal) and
      // read with one v_readlane pair; lane k itself is left alone -- after the step before, its entry IS the numerator of
      // r_k -- and all quotients are formed together after the last step (9 instructions per step instead of 18).
      auto step = [&](int k, double col) __attribute__((always_inline)) {
        const double v = bcast_lane_s(div_pre(rk, dg, rinv), k);
        const double upd = FMA(-col, v, rk);
        rk = (lane < k) ? upd : rk;
      };
      int k = uniform(q) - 1; // (a scalar counter: see bcast_lane in device_common.h)
      double c0 = colat(k), c1 = colat(k - 1), c2 = colat(k - 2), c3 = colat(k - 3);
#pragma unroll 1
      for(; k >= 3; k -= 4)
      {
        step(k, c0);
        c0 = colat(k - 4);
        step(k - 1, c1);
        c1 = colat(k - 5);
        step(k - 2, c2);
        c2 = colat(k - 6);
        step(k - 3, c3);
        c3 = colat(k - 7);
      }
      if(k >= 0) step(k, c0);
      if(k >= 1) step(k - 1, c1);
      if(k >= 2) step(k - 2, c2);
      const double pre = rk; // the numerators (lanes >= q: 0)
      rk = div_pre(pre, dg, rinv);
      const bool safe = __all(div_safe_den(dg) && div_safe_num(pre));
      if(!safe)
      {
        // an operand left the range in which the hoisted reciprocal is provably the compiler's own: redo with a/b
        rk = rk0;
        double col = R[rl + ((q > 0) ? (q - 1) : 0) * ld];
        for(int k = q - 1; k >= 0; --k)
        {
          const double coln = R[rl + ((k > 0) ? (k - 1) : 0) * ld];
          const double v = opaque_v(bcast_lane(rk, k)) / opaque_v(bcast_lane(dg, k));
          const double upd = FMA(-col, v, rk);
          rk = (lane < k) ? upd : rk;
          rk = (lane == k) ? v : rk;
          col = coln;
        }
      }
      if(lane < q) rs[lane] = rk;
      tr_r = rk;
      tr_u = u_l;
      STAMP(16)
      // t1 = min u_k / r_k over r_k > 0 (not EQUALITY/FIXED; status looked up by POSITION, GoldfarbIdnaniSolver.cpp:162)
      // (u and the status byte were fetched before the substitution loop; the quotient is formed in every lane and
      // selected afterwards: no LDS round trip and no divergent region on this path)
      const double tt = u_l / ((rk > 0.0) ? rk : 1.0);
      const bool cand = (lane < q) && sk_l != EQUALITY && sk_l != FIXED && rk > 0.0 && tt < big;
      double tk = cand ? tt : big;
      int kk = cand ? lane : NONE, dummy = 0;
      wave_argmin_monotone(tk, kk, dummy);
      tr_t1 = (kk == NONE) ? big : tk;
      tr_l = (kk == NONE) ? 0 : kk;
      if(!WS && lane == 0)
      {
        ctl->t1 = tr_t1;
        ctl->l = tr_l;
      }
      STAMP(17)
    };

    // K6 + the scalar products of K8: wave 1 computes z, then z.z and n+.z as C4 trees over its lanes (CANON C5: one product per
    // element, adjacent-first pairwise tree -- four DPP exchanges and three adds instead of two 50-step chains); wave 2 computes
    // n+.x (a C1 chain: the value the scan computed for the constraint; it does not need z) at the same time
    auto z_and_dots = [&](int pp) {
      const bool general = pp < m;
      if(wave == 2)
      {
        if(lane == 0) ctl->cx = general ? chain_up(np_ptr(pp), 1, xs, 1, 0, n, 0.0) : xs[pp - m];
        return;
      }
      const int rl = (lane < n) ? lane : 0;
      const double npv = general ? np_ptr(pp)[rl] : 0.0; // (requested before the chain)
      double zi = chain_up(J + rl, ld, ds, 1, q, n, 0.0);
      zi = (lane < n) ? zi : 0.0;
      if(lane < n) zs[lane] = zi;
      const double zz = wave_sum_tree(FMA(zi, zi, 0.0));
      double cz;
      if(general)
        cz = wave_sum_tree(FMA(npv, zi, 0.0)); // (lanes >= n: z = 0)
      else
        cz = bcast_lane(zi, pp - m);
      if(lane == 0)
      {
        ctl->zz = zz;
        ctl->cz = cz;
      }
    };

    // K9 (wave 0): primal/dual update
    auto apply_step = [&](double t, double dotsc, bool dual_only) {
      if(!dual_only)
      {
        if(lane < n) xs[lane] = FMA(t, zs[lane], xs[lane]);
        if(lane == 0) ctl->f = FMA(t * dotsc, 0.5 * t + us[q], ctl->f); // uses u[q] BEFORE its update
      }
      wave_sync();
      if(lane < q) us[lane] = FMA(-t, rs[lane], us[lane]);
      if(lane == 0) us[q] = us[q] + t;
      wave_sync();
    };

    // K10 (wave 0): addConstraint_ (GoldfarbIdnaniSolver.cpp:221-237).
    //   1. the coefficients (CANON C7, device_common.h c7_wave): lane i forms d_i^2, a six-step scan gives every lane the squared
    //      norm of d(i:n), and (c_i, s_i) are one square root and two divisions per lane -- no serial chain (rounds 1-3 ran the
    //      radius recurrence r_i = hypot(d_i, r_{i+1}) here: n - q dependent steps, seeded and systolic at best).  Entries too large
    //      or too small to be squared take the serial recurrence instead (chain_exact), as the oracle does;
    //   2. lane = row of J: sweep i = n-2 .. q applying the rotations, operands prefetched one step ahead
    // wave 0: phase 1; publishes (c_i, s_i) in LDS (z and r are dead after the primal/dual update) and raises
    // the flag wave 1 is waiting on; then the new column of R and the active-set bookkeeping
    double gc_d = 0.0, gc_rho = 0.0, gc_kq = 0.0;
    // (the slices of the old serial chain, interleaved with the other waves' phases: nothing left to interleave)
    auto chain_steps = [&](int, bool = false) {};
    // the serial recurrence: every lane runs it (uniform values); lane i keeps r_{i+1}
    auto chain_exact = [&]() {
      gc_rho = opaque_v(bcast_lane(gc_d, n - 1));
      gc_kq = 0.0;
      for(int i = n - 2; i >= q; --i)
      {
        const double p_i = opaque_v(bcast_lane(gc_d, i));
        gc_kq = (lane == i) ? gc_rho : gc_kq;
        gc_rho = givens_r_only(p_i, gc_rho);
      }
    };
    auto chain_finish = [&](int pp, int ps, int token) {
      STAMP(13)
      {
        const int rl = (lane < n) ? lane : (n - 1);
        const double dv = ds[rl];
        gc_d = (lane < n) ? dv : 0.0;
      }
      gc_rho = bcast_lane(gc_d, n - 1);
      if(n - 2 >= q)
      {
        const bool mine = lane >= q && lane <= n - 2;
        double cl, sl, rq;
        if(c7_wave(gc_d, lane, n, q, cl, sl, rq))
          gc_rho = rq;
        else
        {
          chain_exact();
          GivensHalf h;
          (void)givens_r(gc_d, gc_kq, h);
          givens_cs(h, cl, sl); // lane i now holds (c_i, s_i)
          if(PROF && CMODE == 2 && lane == 0) atomicAdd(&prof_lds[13], 1ULL); // (slot 13 is idle when wave 3 runs the chain)
        }
        if(mine)
        {
          zs[lane] = cl;
          rs[lane] = sl;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if(lane == 0) __atomic_store_n(&ctl->flag, token, __ATOMIC_RELAXED);
      STAMP(14)
      // R.col(q).head(q+1) = d.head(q+1)
      if(lane < q)
        R[lane + q * ld] = gc_d;
      else if(lane == q)
        R[q + q * ld] = gc_rho;
      if(lane == 0)
      {
        act[q] = pp;
        st[pp] = (unsigned char)ps;
      }
    };
    auto add_chain = [&](int pp, int ps, int token) {
      chain_steps(n, true);
      chain_finish(pp, ps, token);
    };

    // wave 1: phase 3, lane = row of J (lanes >= n shadow row n-1: same values, same addresses); runs while wave 0
    // scans its own constraints
    auto add_apply = [&](int qa, int token) {
      if(token >= 0) // coefficients published by another wave: wait for its flag
      {
        while(__atomic_load_n(&ctl->flag, __ATOMIC_RELAXED) != token) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      if(n - 2 < qa) return;
      const int rl = (lane < n) ? lane : (n - 1);
      double carry = J[rl + (n - 1) * ld];
      int i = n - 2;
#pragma unroll 1
      for(; i - 3 >= qa; i -= 4)
      {
        double xv[4], cv[4], sv[4];
#pragma unroll
        for(int u = 0; u < 4; ++u)
        {
          xv[u] = J[rl + (i - u) * ld];
          cv[u] = zs[i - u];
          sv[u] = rs[i - u];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for(int u = 0; u < 4; ++u)
        {
          J[rl + (i - u + 1) * ld] = FMA(sv[u], xv[u], cv[u] * carry);
          carry = FMA(-sv[u], carry, cv[u] * xv[u]);
        }
      }
#pragma unroll 1
      for(; i >= qa; --i)
      {
        const double xv = J[rl + i * ld];
        const double c2 = zs[i], s2 = rs[i];
        J[rl + (i + 1) * ld] = FMA(s2, xv, c2 * carry);
        carry = FMA(-s2, carry, c2 * xv);
      }
      J[rl + qa * ld] = carry;
    };

    // K11 (wave 0): removeConstraint (DualSolver.cpp:237-244, GoldfarbIdnaniSolver.cpp:239-256)
    auto do_drop = [&](int l) {
      l = uniform(l); // (scalar loop counters below: the rotation loop broadcasts by a running index, see bcast_lane)
      double uv = (lane >= l && lane < q) ? us[lane + 1] : 0.0;
      int av = (lane >= l && lane < q - 1) ? act[lane + 1] : 0;
      int gone = act[l];
      wave_sync();
      if(lane >= l && lane < q) us[lane] = uv;
      if(lane >= l && lane < q - 1) act[lane] = av;
      if(lane == 0) st[gone] = INACTIVE;
      const int qn = uniform(q) - 1;
      const bool mine = (lane >= l && lane < qn); // lane = NEW column index; old column = lane + 1
      // head rows 0 .. l-1 of the columns behind l move one column to the left: eight rows per round trip (all reads of a round
      // before its writes: lane j writes the column lane j-1 reads)
      for(int r0 = 0; r0 < l; r0 += 8)
      {
        double v[8];
        const double * src = R + (mine ? (lane + 1) : 0) * ld;
#pragma unroll
        for(int e = 0; e < 8; ++e) v[e] = src[min(r0 + e, l - 1)];
        wave_sync();
#pragma unroll
        for(int e = 0; e < 8; ++e)
          if(mine && r0 + e < l) R[(r0 + e) + lane * ld] = v[e];
        wave_sync();
      }
      double carry = mine ? R[l + (lane + 1) * ld] : 0.0;
      wave_sync();
      double y = (lane >= l && lane < qn) ? R[(l + 1) + (lane + 1) * ld] : 0.0;
      for(int i = l; i < qn; ++i)
      {
        double ynext = (i + 1 < qn && lane >= i + 1 && lane < qn) ? R[(i + 2) + (lane + 1) * ld] : 0.0;
        GivensHalf h;
        double rho = givens_r(bcast_lane_s(carry, i), bcast_lane_s(y, i), h);
        double c, s;
        givens_cs(h, c, s);
        if(lane == 0)
        {
          ds[i] = c; // d and z are dead here: reuse them to hand (c, s) to the J sweep
          zs[i] = s;
        }
        if(lane == i)
          R[i + i * ld] = rho;
        else if(lane > i && lane < qn)
        {
          R[i + lane * ld] = FMA(-s, y, c * carry);
          carry = FMA(s, carry, c * y);
        }
        y = ynext;
      }
      wave_sync();
      if(lane < n && qn > l)
      {
        double jc = J[lane + l * ld];
        double ynext = J[lane + (l + 1) * ld];
        double cn = ds[l], sn = zs[l]; // (c, s) of the next rotation requested one step ahead
        for(int i = l; i < qn; ++i)
        {
          const double c = cn, s = sn;
          const double yy = ynext;
          const int i1 = (i + 1 < qn) ? (i + 1) : i;
          cn = ds[i1];
          sn = zs[i1];
          ynext = J[lane + (i1 + 1) * ld];
          J[lane + i * ld] = FMA(-s, yy, c * jc);
          jc = FMA(s, jc, c * yy);
        }
        J[lane + qn * ld] = jc;
      }
    };

    // ======================================================================================================
    // K12: initActiveSet -- equalities (bl == bu) then fixed variables (xl == xu), in index order, full step each
    // ======================================================================================================
    if constexpr(WARM)
    {
      // ==================================================================================================
      // warm start: initializeComputationData + initializePrimalDualPoints + purge of the negative multipliers
      // (experimental/GoldfarbIdnaniSolver.cpp:388-487, 78-109); J = L^-T and N are in place
      // ==================================================================================================
      q = uniform(ctl->nact);
      // ---- B = L^-1 N (:426): wave w takes columns w, w+4, ...; lane = row, x0's forward substitution (wave_fwd: hoisted reciprocals,
      // the column entries of L requested a group of steps ahead, no branch in a group, its own range check and plain-division
      // fall-back -- round 5; the loop here took 325 cycles a step where wave_fwd takes ~60: 78 k of a warm start's 310 k cycles at n = 40)
      for(int j = wave; j < q; j += 4)
      {
        const double b0 = (lane < n) ? Bw[lane + j * ld] : 0.0;
        const double xi = wave_fwd(L, ld, n, b0, lane);
        if(lane < n) Bw[lane + j * ld] = xi;
      }
      __syncthreads();
      STAMP(24) // B = L^-1 N
      // ---- B = Q R, Householder, column by column (:432-440; Eigen makeHouseholder / applyHouseholderOnTheLeft) on
      // wave 0; wave 1 applies each reflector to J from the right as soon as it exists (J = J Q, :443-444)
      for(int k = 0; k < q; ++k)
      {
        if(wave == 0)
        {
          const double * vk = Bw + k * ld;
          const double tailsq = chain_up(vk, 1, vk, 1, k + 1, n, 0.0); // every lane the same chain (LDS broadcast)
          const double c0 = vk[k];
          const bool zero = tailsq <= 2.2250738585072014e-308;
          double beta = __builtin_sqrt(FMA(c0, c0, tailsq));
          beta = (c0 >= 0.0) ? -beta : beta;
          const double den = c0 - beta;
          double tau = (beta - c0) / beta;
          tau = zero ? 0.0 : tau;
          beta = zero ? c0 : beta;
          const double mine = (lane > k && lane < n) ? vk[lane] : 0.0;
          wave_sync();
          if(lane > k && lane < n) Bw[lane + k * ld] = zero ? 0.0 : mine / den;
          if(lane == 0)
          {
            Bw[k + k * ld] = beta;
            hco[k] = tau;
          }
        }
        __syncthreads();
        const double tau = uniform_d(hco[k]);
        if(tau != 0.0 && k + 1 < n)
        {
          const double * v = Bw + k * ld;
          if(wave == 0)
          {
            if(lane > k && lane < q) // lane = trailing column
            {
              double * cj = Bw + lane * ld;
              double t = chain_up(v, 1, cj, 1, k + 1, n, 0.0);
              const double top = cj[k];
              t = t + top;
              cj[k] = FMA(-tau, t, top);
              const double stt = tau * t;
              for(int i = k + 1; i < n; ++i) cj[i] = FMA(-stt, v[i], cj[i]);
            }
            wave_sync();
          }
          else if(wave == 1)
          {
            if(lane < n) // lane = row of J
            {
              double t = chain_up(J + lane, ld, v, 1, k + 1, n, 0.0);
              const double jk = J[lane + k * ld];
              t = t + jk;
              J[lane + k * ld] = FMA(-tau, t, jk);
              const double stt = tau * t;
              for(int j = k + 1; j < n; ++j) J[lane + j * ld] = FMA(-stt, v[j], J[lane + j * ld]);
            }
          }
        }
      }
      __syncthreads();
      STAMP(25) // QR and J = J Q
      // ---- R = upper triangle of the factored B (L is dead)
      for(int e = tid; e < q * q; e += WG)
      {
        const int i = e % q, j = e / q;
        if(i <= j) R[i + j * ld] = Bw[i + j * ld];
      }
      __syncthreads();
      STAMP(26) // R
      // ---- initializePrimalDualPoints (:462-487), repeated after every purge (:78-109)
      for(;;)
      {
        if(wave == 1)
        {
          if(lane < n) ds[lane] = chain_up(J + lane * ld, 1, nps, 1, 0, n, 0.0); // alpha = J^T a
        }
        else if(wave == 0)
        {
          // beta = R^-T b_act: forward substitution, lane = component
          const int cl = (lane < q) ? lane : 0;
          double acc = (lane < q) ? bact[lane] : 0.0;
          for(int i = 0; i < q; ++i)
          {
            const double v = opaque_v(bcast_lane(acc, i)) / opaque_v(R[i + i * ld]);
            const double upd = FMA(-R[i + cl * ld], v, acc);
            acc = (lane > i && lane < q) ? upd : acc;
            acc = (lane == i) ? v : acc;
          }
          if(lane < q) rs[lane] = acc;
        }
        __syncthreads();
        if(wave == 0)
        {
          if(lane < n) // x = J1 beta - J2 alpha2
          {
            double acc = chain_up(J + lane, ld, rs, 1, 0, q, 0.0);
            acc = chain_up_neg(J + lane, ld, ds, 1, q, n, acc);
            xs[lane] = acc;
          }
        }
        else if(wave == 1)
        {
          // u = R^-1 (alpha1 + beta)
          const int rl = (lane < q) ? lane : 0;
          double uk = (lane < q) ? (ds[rl] + rs[rl]) : 0.0;
          for(int k = q - 1; k >= 0; --k)
          {
            const double v = opaque_v(bcast_lane(uk, k)) / opaque_v(R[k + k * ld]);
            const double upd = FMA(-R[rl + k * ld], v, uk);
            uk = (lane < k) ? upd : uk;
            uk = (lane == k) ? v : uk;
          }
          if(lane < q) us[lane] = uk;
          // most negative multiplier of an inequality (first one on ties), threshold -1e-14 (:82-93)
          double cand = 1.0;
          int ci = NONE, dummy = 0;
          if(lane < q)
          {
            const int sk = st[act[lane]];
            if(uk < -1e-14 && sk != FIXED && sk != EQUALITY)
            {
              cand = uk;
              ci = lane;
            }
          }
          wave_argmin_monotone(cand, ci, dummy);
          if(lane == 0) ctl->l = (ci == NONE) ? -1 : ci;
        }
        else if(wave == 2)
        {
          // f = beta.(0.5 beta + alpha1) - 0.5 |alpha2|^2
          if(lane < q) zs[lane] = FMA(0.5, rs[lane], ds[lane]);
          wave_sync();
          if(lane == 0)
          {
            const double f1 = chain_up(rs, 1, zs, 1, 0, q, 0.0);
            const double f2 = chain_up(ds, 1, ds, 1, q, n, 0.0);
            ctl->f = FMA(-0.5, f2, f1);
          }
        }
        __syncthreads();
        const int lmin = uniform(ctl->l);
        if(lmin < 0) break;
        ++it;
        if(wave == 0)
        {
          const double bv = (lane >= lmin && lane < q - 1) ? bact[lane + 1] : 0.0;
          wave_sync();
          if(lane >= lmin && lane < q - 1) bact[lane] = bv;
          do_drop(lmin);
        }
        q = uniform(q - 1);
        __syncthreads();
      }
      STAMP(27) // start point + purge
    }
    else if(ctl->eq_next != NONE)
    {
      if(ctl->cnt > nr) // (the caller's n: an embedded problem has no more unknowns than that)
      {
        term = T_OVERCONSTRAINED; // UB in the reference; reported (as the oracle does)
      }
      else
      {
        int cursor = 0;
        for(;;)
        {
          int first = NONE;
          for(int i = cursor + tid; i < mt; i += WG)
          {
            bool eq = (i < m) ? (blg[i] == bug[i]) : (xlg[i - m] == xug[i - m]);
            if(eq && i < first) first = i;
          }
          __syncthreads();
          if(tid == 0) ctl->eq_next = NONE;
          __syncthreads();
          if(first != NONE) atomicMin(&ctl->eq_next, first);
          __syncthreads();
          p = uniform(ctl->eq_next);
          if(p == NONE) break;
          pstat = (p < m) ? EQUALITY : FIXED;
          cursor = p + 1;
          publish_np(p, pstat, false);
          if(tid == 0) us[q] = 0.0;
          __syncthreads();
          if(wave == 0) compute_d(p, pstat);
          __syncthreads();
          if(wave == 0)
            trsv_ratio();
          else if(wave <= 2)
            z_and_dots(p);
          __syncthreads();
          // addInitialConstraint (GoldfarbIdnaniSolver.cpp:295-338): full step, no ratio test
          const double cz = uniform_d(ctl->cz);
          double t = 0.0;
          if(__builtin_sqrt(uniform_d(ctl->zz)) > 1e-14)
          {
            t = (uniform_d(bound_of(p, pstat)) - uniform_d(ctl->cx)) / cz;
          }
          ++addno;
          if(wave == 0)
          {
            apply_step(t, cz, false);
            add_chain(p, pstat, addno);
          }
          else if(wave == 1)
            add_apply(q, addno);
          q = uniform(q + 1);
          __syncthreads();
        }
      }
    }

    // ======================================================================================================
    // main loop (DualSolver.cpp:103-166)
    // ======================================================================================================
    if(term == T_SUCCESS)
    {
      if(it >= A.max_iter) // it > 0 only after a warm start purged constraints
        term = T_MAX_ITER;
      else if constexpr(WS)
      {
        // ==================================================================================================
        // WS main loop: wave 3 = MAIN, waves 0-2 = HELPERS (see the comment at WS above).  Hand-off words in ctl:
        //   fd    = pass   : d, wp, wps, wq published                       MAIN -> helpers
        //   fz    = pass   : z, z.z, n.z, n.x, t2 published                     wave 1 -> MAIN
        //   fcs   = pass   : (c, s) pairs + rho published, or chain abandoned (also: d no longer read)  wave 0 -> MAIN
        //   fstep = pass   : x, u final, decision in dec                    MAIN -> helpers
        //   fscan[w] = scan number: winner of wave w published              helpers -> MAIN
        //   fack0 / fack2 = pass: wave 0 / wave 2 has read the decision of a DROP pass        helpers -> MAIN
        // A word is written after the data it announces (release) and polled by its readers (acquire).  The polls test for
        // equality, so a helper must never be more than one pass behind.  That holds because MAIN, before it leaves a pass,
        // has heard from every helper AFTER that helper read the pass's decision: wave 1 through fz of the next pass (it only
        // looks for the next d after the decision), waves 0 and 2 through fscan[w] in a pass that adds and through
        // fack0 / fack2 in a pass that drops.  So fd, fstep and dec are never rewritten before everybody has read them.
        // ==================================================================================================
        constexpr int MAINW = 3;
        // (c_i, s_i) at csb[2i], csb[2i+1], 16-byte aligned (the dynamic LDS base is): pointer arithmetic only, so that the
        // compiler keeps the LDS address space (an integer round trip turns every access into a flat load)
        double * csb = gsc + (((unsigned)reinterpret_cast<size_t>(gsc) >> 3) & 1u);
        // A waiting wave sleeps in long naps (an instruction it issues is an issue slot the working waves of its SIMD do not
        // get, and a helper waits most of the time); the wave that posts a word wakes the sleepers of its workgroup
        // (s_wakeup), so the hand-off does not wait for a nap to end.  A wake-up that arrives just before the s_sleep is
        // lost: the nap then bounds the delay.
        constexpr int NAP = 24; // x 64 cycles (6, 12, 24, 60 measured in round 3: 24 and 60 are 0.5-1 % ahead of 6 -- fewer polls in the issue slots of the working waves)
        // Every wait is BOUNDED (about a tenth of a second of naps): should a hand-off ever be lost, the waves give up one after
        // the other, the problem comes back with TerminationStatus UNKNOWN, and the grid drains -- a hung workgroup would hang
        // the whole launch.
        constexpr int SPIN_LIMIT = 1 << 17;
        auto ws_wait = [&](const int * word, int target) __attribute__((always_inline)) -> bool {
          int spins = 0;
          while(__atomic_load_n(word, __ATOMIC_RELAXED) != target)
          {
            if(++spins > SPIN_LIMIT) return false;
            __builtin_amdgcn_s_sleep(NAP);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          return true;
        };
        auto ws_post = [&](int * word, int value) __attribute__((always_inline)) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if(lane == 0) __atomic_store_n(word, value, __ATOMIC_RELAXED);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory");
        };
        // K4 on a helper wave, WS form: the column chain runs for every lane without a branch around it (a branch would put
        // the loads in an exec-mask region of their own), x arrives in double-buffered groups of XG as 16-byte broadcast
        // reads.  Then the wave's arg-min and the hand-off of the winner: (value, index, status) and, by its owner thread, n+
        // and the bound.  The LDS-staged extra columns (m > 192) are MAIN's (ws_extras): it has the registers and, after its
        // sweep, the time.
        auto ws_scan = [&](int excl) __attribute__((always_inline)) {
          constexpr int XG = 6;
          static_assert(NR % 2 == 0 || NT == 0, "x is read in pairs");
          const int sbyte = (myc < m) ? st[myc] : 1;
          double cx = 0.0;
          double xa[XG], xb[XG];
          auto xfetch = [&](double (&xq)[XG], int g) __attribute__((always_inline)) {
#pragma unroll
            for(int u = 0; u < XG; u += 2)
            {
              const int k = (g * XG + u < NR) ? (g * XG + u) : (NR - 2);
              const double2 v = *reinterpret_cast<const double2 *>(xs + k);
              xq[u] = v.x;
              xq[u + 1] = v.y;
            }
          };
          auto xchain = [&](const double (&xq)[XG], int g) __attribute__((always_inline)) {
#pragma unroll
            for(int u = 0; u < XG; ++u)
              if(g * XG + u < NR) cx = FMA(creg[g * XG + u], xq[u], cx);
          };
          constexpr int XNG = (NR + XG - 1) / XG;
          double ex = 0.0;
          const bool e_on = (wave == 2) && (ne > 0); // (uniform) wave 2 -- idle inside a pass -- also takes the extra columns
          if(e_on)
          {
            // lane e: extra column 192 + e as a SECOND chain in the same loop (a dependent fp64 chain leaves every other issue
            // slot free).  One buffer per operand stream, refilled in place right after use: eight terms of look-ahead.
            const double * ce = Ce + (size_t)((lane < ne) ? lane : 0) * ld;
            double eq[XG];
#pragma unroll
            for(int u = 0; u < XG; ++u) eq[u] = ce[u];
            xfetch(xa, 0);
#pragma unroll
            for(int k = 0; k < NR; ++k)
            {
              const int u = k % XG;
              cx = FMA(creg[k], xa[u], cx);
              ex = FMA(eq[u], xa[u], ex);
              if(k + XG < NR) eq[u] = ce[k + XG];
              if((u & 1) && k + XG < NR + 1)
              {
                const int kk = (k - 1 + XG < NR) ? (k - 1 + XG) : (NR - 2);
                const double2 v = *reinterpret_cast<const double2 *>(xs + kk);
                xa[u - 1] = v.x;
                xa[u] = v.y;
              }
              if(u == XG - 1) __builtin_amdgcn_sched_barrier(0);
            }
          }
          else
          {
            xfetch(xa, 0);
#pragma unroll
            for(int g = 0; g < XNG; g += 2)
            {
              if(g + 1 < XNG) xfetch(xb, g + 1);
              __builtin_amdgcn_sched_barrier(0);
              xchain(xa, g);
              if(g + 2 < XNG) xfetch(xa, g + 2);
              __builtin_amdgcn_sched_barrier(0);
              if(g + 1 < XNG) xchain(xb, g + 1);
            }
          }
          // candidates: value < 0 only; lower tested before upper (GoldfarbIdnaniSolver.cpp:99-110)
          double bval = 1.0;
          int bidx = NONE, bstat = INACTIVE;
          {
            const double sl = cx - my_bl, su = my_bu - cx;
            const bool ok = (myc < m) && sbyte == INACTIVE && myc != excl;
            const bool lo = ok && sl < 0.0, up = ok && !lo && su < 0.0;
            bval = lo ? sl : (up ? su : 1.0);
            bidx = (lo || up) ? myc : NONE;
            bstat = lo ? LOWER : (up ? UPPER : INACTIVE);
          }
          if(nb > 0)
          {
            // bounds: thread i < 192 takes variables i, i + 192, ... (general constraints come first: a bound wins a tie only
            // through the index order of the arg-min below)
            double v0 = (bidx == NONE) ? 0.0 : bval;
            for(int i = tid; i < nb; i += CREG_MAX)
            {
              if(st[m + i] != INACTIVE || (m + i) == excl) continue;
              const double xi = xs[i];
              const double sl = xi - xlg[i];
              if(sl < v0)
              {
                v0 = sl;
                bidx = m + i;
                bstat = LOWER_BOUND;
              }
              else
              {
                const double su = xug[i] - xi;
                if(su < v0)
                {
                  v0 = su;
                  bidx = m + i;
                  bstat = UPPER_BOUND;
                }
              }
            }
            bval = (bidx == NONE) ? 1.0 : v0;
            wave_argmin(bval, bidx, bstat);
          }
          else
          {
            // index monotone in the lane: minimum value, then the first lane holding it
            const double mn = wave_min_d_fast(bval);
            const unsigned long long msk = __ballot(bidx != NONE && bval == mn);
            if(msk == 0)
              bidx = NONE;
            else
            {
              const int src = __builtin_ctzll(msk);
              bval = mn;
              bidx = __builtin_amdgcn_readlane(bidx, src);
              bstat = __builtin_amdgcn_readlane(bstat, src);
            }
          }
          if(lane == 0)
          {
            ctl->red_val[wave] = bval;
            ctl->red_idx[wave] = bidx;
            ctl->red_stat[wave] = bstat;
          }
          if(myc == bidx && myc < m) // this wave's winner: n+ and its bound into the wave's slot
          {
            double * slot = npw + wave * n;
            if constexpr(NR % 2 == 0)
            {
#pragma unroll
              for(int k = 0; k < NR; k += 2) *reinterpret_cast<double2 *>(slot + k) = make_double2(creg[k], creg[k + 1]);
            }
            else
            {
#pragma unroll
              for(int k = 0; k < NR; ++k) slot[k] = creg[k];
            }
            ctl->bndw[wave] = (bstat == LOWER) ? my_bl : my_bu;
            ctl->cxw[wave] = cx; // n+.x for the step length: this very chain (GoldfarbIdnaniSolver.cpp:97 and :183-190 are one expression)
          }
          if(wave == 2 && ne <= 64)
          {
            double ev = 1.0;
            int ei = NONE, es = INACTIVE;
            const int i = CREG_MAX + lane;
            if(e_on && lane < ne && st[i] == INACTIVE && i != excl)
            {
              const double sl = ex - ebl[lane], su = ebu[lane] - ex; // (bl, bu of the staged columns: in LDS since the prologue)
              if(sl < 0.0)
              {
                ev = sl;
                ei = i;
                es = LOWER;
              }
              else if(su < 0.0)
              {
                ev = su;
                ei = i;
                es = UPPER;
              }
            }
            if(ne > 0) wave_argmin_monotone(ev, ei, es);
            if(lane == 0)
            {
              ctl->red_val[3] = ev;
              ctl->red_idx[3] = ei;
              ctl->red_stat[3] = es;
            }
            if(ei == i && e_on && lane < ne) ctl->cxw[3] = ex;
          }
        };
        // MAIN's share of K4 (only when there are more than 64 extra columns): the LDS-staged extra columns, 64 per round (lane e = column 192 + e0 + e); a lane keeps its
        // first best one (strict <), then the arg-min; the winner goes to slot 3 of the reduction words
        auto ws_extras = [&](int excl) __attribute__((always_inline)) {
          if(ne <= 64) return; // (wave 2 runs them beside its own columns)
          double ev = 1.0, ecx = 0.0;
          int ei = NONE, es = INACTIVE, mine = NONE;
          for(int e0 = 0; e0 < ne; e0 += 64)
          {
            const int i = CREG_MAX + e0 + lane;
            const bool on = e0 + lane < ne;
            const double * ce = Ce + (size_t)(on ? e0 + lane : 0) * ld;
            // (a plain chunked chain: MAIN waits for the helpers' scans afterwards anyway, and a second fully unrolled
            // pipeline here makes the register allocator spill the helpers' column registers)
            const double cxe = chain_up(ce, 1, xs, 1, 0, n, 0.0);
            if(on && st[i] == INACTIVE && i != excl)
            {
              const double sl = cxe - blg[i];
              const double su = bug[i] - cxe;
              if(sl < 0.0)
              {
                if(sl < ev)
                {
                  ev = sl;
                  ei = i;
                  es = LOWER;
                  ecx = cxe;
                }
              }
              else if(su < 0.0 && su < ev)
              {
                ev = su;
                ei = i;
                es = UPPER;
                ecx = cxe;
              }
            }
          }
          mine = ei;
          if(ne > 64)
            wave_argmin(ev, ei, es);
          else if(ne > 0)
            wave_argmin_monotone(ev, ei, es);
          if(lane == 0)
          {
            ctl->red_val[3] = ev;
            ctl->red_idx[3] = ei;
            ctl->red_stat[3] = es;
          }
          if(ei != NONE && mine == ei) ctl->cxw[3] = ecx; // (one lane: a constraint belongs to one lane) n+.x of the winner, see ws_scan
          wave_sync();
        };
        if(!ovl)
        {
          if(tid == 0)
          {
            ctl->wit = it; ctl->wterm = T_SUCCESS; ctl->wq = q;
          }
          __syncthreads();
        }
        const int tune = A.tune; // experiment bits (JRLQP_TUNE): 1 MAIN priority 3, 2 chain wave priority 2, 4 z wave priority 1
        if(wave == MAINW)
        {
          // ------------------------------------------------------------------------------------------ MAIN
          if(tune & 1) __builtin_amdgcn_s_setprio(3);
          int sq = 0, sc = 1;
          int wterm = T_SUCCESS;
          const int rl = (lane < n) ? lane : (n - 1);
          // x (lane i: x_i) and f live in MAIN's registers for the whole loop -- only MAIN changes them; xs is written through
          // for the scans, ctl->f once at the end
          double x_reg = 0.0, f_reg = 0.0;
#define WS_WAIT(word, target)      \
  if(!ws_wait((word), (target)))   \
  {                                \
    wterm = T_UNKNOWN;             \
    goto ws_stop;                  \
  }
          // first selection: the helpers scan with x = x0
          if(ovl && !ws_wait(&ctl->fx0, 1)) wterm = T_UNKNOWN; // (overlapped prologue: x0 comes from wave 0 without a barrier)
          ws_extras(-1);
          bool running = wterm == T_SUCCESS && ws_wait(&ctl->fscan[0], sc) && ws_wait(&ctl->fscan[1], sc) && ws_wait(&ctl->fscan[2], sc);
          x_reg = xs[rl];
          f_reg = ctl->f;
          if(running)
            scan_combine(p, pstat);
          else
            wterm = T_UNKNOWN;
          STAMP3(24)
          running = running && (p >= 0);
          if(running && lane == 0) us[q] = 0.0; // DualSolver.cpp:121
          bool fresh = true; // p comes from a scan of the current x (false in the pass behind a removal: skipStep1)
          while(running)
          {
            ++sq;
            // ---- K5: d = +-J^T n+ (general: every entry of the column in flight at once; bound: a row of J)
            {
              double dj;
              if(p < m)
              {
                const double * np = np_ptr(p);
                const double * jc = J + rl * ld;
                if constexpr(NT > 0)
                {
                  // A wave issues at most one instruction every four cycles, whatever its kind: the chain is written for the
                  // fewest instructions -- compile-time offsets (no address arithmetic), both operands by LDS reads (n+ as a
                  // broadcast read; a v_readlane pair per term would double the count), groups of DG terms, the next group
                  // requested before the current one is consumed.
                  constexpr int DG = 8;
                  constexpr int NG = (NR + DG - 1) / DG;
                  double ja[DG], na[DG], jb[DG], nb2[DG];
                  auto dfetch = [&](double (&jq)[DG], double (&nq)[DG], int g) __attribute__((always_inline)) {
#pragma unroll
                    for(int u = 0; u < DG; ++u)
                    {
                      const int k = (g * DG + u < NR) ? (g * DG + u) : (NR - 1);
                      jq[u] = jc[k];
                      nq[u] = np[k];
                    }
                  };
                  auto dchain = [&](const double (&jq)[DG], const double (&nq)[DG], int g) __attribute__((always_inline)) {
#pragma unroll
                    for(int u = 0; u < DG; ++u)
                      if(g * DG + u < NR) dj = FMA(jq[u], nq[u], dj);
                  };
                  dj = 0.0;
                  dfetch(ja, na, 0);
#pragma unroll
                  for(int g = 0; g < NG; g += 2)
                  {
                    if(g + 1 < NG) dfetch(jb, nb2, g + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    dchain(ja, na, g);
                    if(g + 2 < NG) dfetch(ja, na, g + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if(g + 1 < NG) dchain(jb, nb2, g + 1);
                  }
                }
                else
                  dj = chain_up(jc, 1, np, 1, 0, n, 0.0);
              }
              else
                dj = J[(p - m) + rl * ld];
              if(pstat == UPPER || pstat == UPPER_BOUND) dj = -dj;
              if(lane < n) ds[lane] = dj;
              if(lane == 0)
              {
                ctl->wp = p;
                ctl->wps = pstat;
                ctl->wq = q;
                ctl->wfresh = fresh ? 1 : 0;
              }
            }
            ws_post(&ctl->fd, sq);
            STAMP3(25)
            // ---- K7 + ratio test
            trsv_ratio();
            const double t1 = tr_t1;
            const int l = tr_l;
            STAMP3(26)
            WS_WAIT(&ctl->fz, sq)
            STAMP3(27)
            // ---- K8: step length (t2 comes from the z wave)
            const double zv = zs[rl]; // (requested with the two scalars: one LDS round trip for the whole update)
            const double cz = uniform_d(ctl->cz);
            const double dotsc = (pstat == UPPER || pstat == UPPER_BOUND) ? -cz : cz; // ConstraintNormal::dot
            const double t2 = uniform_d(ctl->t2);
            const double t = (t2 < t1) ? t2 : t1;
            if(A.trace && lane == 0 && it < A.trace_cap)
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
              wterm = T_INFEASIBLE; // DualSolver.cpp:131
              break;
            }
            const bool dual_only = (t2 >= big);
            const bool add = !dual_only && (t == t2);
            {
              // K9 out of registers (DualSolver.cpp:136-150): x += t z, f += t (n+.z) (t/2 + u_q) with u_q BEFORE its update,
              // u_k -= t r_k, u_q += t.  r, u and u_q (zero behind a fresh selection, the shifted pending multiplier behind a
              // removal) are still in the registers of trsv_ratio, x and f never left: no load, three stores.
              const double uq = tr_uq;
              if(!dual_only)
              {
                x_reg = FMA(t, zv, x_reg);
                if(lane < n) xs[lane] = x_reg;
                f_reg = FMA(t * dotsc, 0.5 * t + uq, f_reg);
              }
              if(lane < q) us[lane] = FMA(-t, tr_r, tr_u);
              if(lane == 0) us[q] = uq + t;
            }
            if(lane == 0) ctl->dec = add ? 1 : 2;
            ws_post(&ctl->fstep, sq);
            STAMP3(28)
            if(add)
            {
              // ---- K10: the coefficients come from the chain wave (speculative since d was published)
              WS_WAIT(&ctl->fcs, sq)
              STAMP3(29)
              const int qa = uniform(q);
              if(n - 2 >= qa)
              {
                // lane = row of J (lanes >= n shadow row n-1: same values, same addresses).  Columns n-2 .. qa, two dependent
                // operations per column.  With n a compile-time constant the sweep is unrolled over the column index: every LDS
                // address is a base register plus an immediate, the operands of column i - AD are requested while column i is
                // rotated, and one scalar compare-and-branch per column ends the sweep at qa -- nine instructions per column
                // (an instruction of any kind costs its wave four cycles).
                double carry = J[rl + (n - 1) * ld];
                if constexpr(NT > 0)
                {
                  // blocks of AB columns, straight-line code inside a block; the operands of block b+1 are requested before
                  // block b is rotated.  A block runs only if ALL its columns are inside the sweep (the `alive` chain: once
                  // false it stays false), so that no value loaded ahead is ever merged with another definition -- a merge
                  // would make the compiler wait for the load at once.  The up to AB-1 columns left go one by one.
                  constexpr int AB = 4;
                  constexpr int NB = (NR - 1) / AB;
                  const double * jr = J + rl;
                  double xa[AB], ca[AB], sa[AB], xb[AB], cb[AB], sb[AB];
                  auto afetch = [&](double (&xq)[AB], double (&cq)[AB], double (&sq2)[AB], int b) __attribute__((always_inline)) {
#pragma unroll
                    for(int u = 0; u < AB; ++u)
                    {
                      const int i = (NR - 2 - AB * b - u >= 0) ? (NR - 2 - AB * b - u) : 0;
                      xq[u] = jr[i * ld];
                      const double2 cs2 = *reinterpret_cast<const double2 *>(csb + 2 * i);
                      cq[u] = cs2.x;
                      sq2[u] = cs2.y;
                    }
                  };
                  auto arot = [&](const double (&xq)[AB], const double (&cq)[AB], const double (&sq2)[AB], int b) __attribute__((always_inline)) {
#pragma unroll
                    for(int u = 0; u < AB; ++u)
                    {
                      const int i = NR - 2 - AB * b - u;
                      J[rl + (i + 1) * ld] = FMA(sq2[u], xq[u], cq[u] * carry);
                      carry = FMA(-sq2[u], carry, cq[u] * xq[u]);
                    }
                  };
                  int inext = NR - 2;
                  bool alive = true;
                  afetch(xa, ca, sa, 0);
#pragma unroll
                  for(int b = 0; b < NB; ++b)
                  {
                    alive = alive && (NR - 2 - AB * b - (AB - 1) >= qa);
                    if(alive)
                    {
                      if(b & 1)
                      {
                        if(b + 1 < NB) afetch(xa, ca, sa, b + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        arot(xb, cb, sb, b);
                      }
                      else
                      {
                        if(b + 1 < NB) afetch(xb, cb, sb, b + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        arot(xa, ca, sa, b);
                      }
                      inext = NR - 2 - AB * (b + 1);
                    }
                  }
                  for(int i = inext; i >= qa; --i)
                  {
                    const double xv = J[rl + i * ld];
                    const double2 cs2 = *reinterpret_cast<const double2 *>(csb + 2 * i);
                    J[rl + (i + 1) * ld] = FMA(cs2.y, xv, cs2.x * carry);
                    carry = FMA(-cs2.y, carry, cs2.x * xv);
                  }
                }
                else
                {
                  for(int i = n - 2; i >= qa; --i)
                  {
                    const double xv = J[rl + i * ld];
                    const double c2 = csb[2 * i], s2 = csb[2 * i + 1];
                    J[rl + (i + 1) * ld] = FMA(s2, xv, c2 * carry);
                    carry = FMA(-s2, carry, c2 * xv);
                  }
                }
                J[rl + qa * ld] = carry;
              }
              STAMP3(22)
              // R.col(q).head(q+1) = d.head(q) | rho ; the active set
              {
                const double dq = ds[rl];
                if(lane < qa)
                  R[lane + qa * ld] = dq;
                else if(lane == qa)
                  R[qa + qa * ld] = ctl->rho;
                if(lane == 0)
                {
                  act[qa] = p;
                  st[p] = (unsigned char)pstat;
                }
              }
              q = uniform(q + 1);
              ++it;
              PROF_COUNT3(10)
              ws_extras(p); // (p's status byte is written above, after the helpers may have read it: excluded explicitly)
              STAMP3(30)
              ++sc;
              WS_WAIT(&ctl->fscan[0], sc)
              WS_WAIT(&ctl->fscan[1], sc)
              WS_WAIT(&ctl->fscan[2], sc)
              if(it >= A.max_iter)
              {
                wterm = T_MAX_ITER; // checked before the next selection (for(; it_ < maxIter_; ...), DualSolver.cpp:103)
                break;
              }
              scan_combine(p, pstat);
              STAMP3(31)
              if(p < 0) break; // SUCCESS (DualSolver.cpp:114)
              if(lane == 0) us[q] = 0.0; // DualSolver.cpp:121
              wave_sync();
              fresh = true;
            }
            else
            {
              fresh = false;
              WS_WAIT(&ctl->fdr, sq) // (do_drop hands its rotations on through the vectors d and z)
              do_drop(l);
              q = uniform(q - 1);
              ++it;
              PROF_COUNT3(11)
              WS_WAIT(&ctl->fcs, sq) // the chain wave is through with this pass (its chain was abandoned)
              WS_WAIT(&ctl->fack0, sq) // waves 0 and 2 have read this pass's decision (long ago: the removal took 9 k cycles)
              WS_WAIT(&ctl->fack2, sq)
              STAMP3(23)
              if(it >= A.max_iter)
              {
                wterm = T_MAX_ITER;
                break;
              }
              // skipStep1: same (p, pstat), u[q] keeps the shifted pending multiplier
            }
          }
        ws_stop:
#undef WS_WAIT
          // stop: tell the helpers (they wait either for the next d or for this pass's step)
          if(lane == 0)
          {
            ctl->wit = it;
            ctl->wterm = wterm;
            ctl->wq = q;
            ctl->dec = 3;
            ctl->f = f_reg;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if(lane == 0)
          {
            __atomic_store_n(&ctl->fstep, -1, __ATOMIC_RELAXED);
            __atomic_store_n(&ctl->fd, -1, __ATOMIC_RELAXED);
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory");
        }
        else
        {
          // ------------------------------------------------------------------------------------------ HELPERS
          load_columns();
          if(ovl && wave != 0 && !ws_wait(&ctl->fx0, 1)) { /* bounded wait: the scan below then works on a stale x and MAIN reports UNKNOWN */ }
          if((tune & 2) && wave == 0) __builtin_amdgcn_s_setprio(2);
          if((tune & 4) && wave == 1) __builtin_amdgcn_s_setprio(1);
          int sq = 0, sc = 0;
          double cx_next = 0.0; // wave 1: n+.x for the pass behind a removal (same p, x moved by the partial step)
          ++sc;
          ws_scan(-1);
          ws_post(&ctl->fscan[wave], sc);
          for(;;)
          {
            ++sq;
            // wait for this pass's d (or for the end)
            int got, spins = 0;
            while((got = __atomic_load_n(&ctl->fd, __ATOMIC_RELAXED)) != sq && got != -1)
            {
              if(++spins > SPIN_LIMIT) break;
              __builtin_amdgcn_s_sleep(NAP);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if(got != sq) break; // the end, or a hand-off lost (bounded wait)
            const int wp = uniform(ctl->wp), wq = uniform(ctl->wq);
            bool aborted = false;
            if(wave == 0)
            {
              // ---- the Givens coefficient chain of an addition (GoldfarbIdnaniSolver.cpp:228-231), speculative: seeded and
              // systolic (device_common.h), the state in registers; every eight steps it looks whether MAIN has decided to
              // drop instead, and gives up if so
              STAMP(3)
              const int rl = (lane < n) ? lane : (n - 1);
              const double dv = ds[rl];
              const double gd = (lane < n) ? dv : 0.0;
              ws_post(&ctl->fdr, sq); // d is in registers: MAIN may reuse its LDS vector (a removal does)
              double rho = bcast_lane(gd, n - 1);
              if(n - 2 >= wq)
              {
                // CANON C7 (device_common.h c7_wave): a six-step scan of the squares, a square root and two divisions per lane; the
                // serial recurrence only for entries that cannot be squared
                const bool mine = lane >= wq && lane <= n - 2;
                double cl, sl, rq;
                if(c7_wave(gd, lane, n, wq, cl, sl, rq))
                  rho = rq;
                else
                {
                  double rr = opaque_v(bcast_lane(gd, n - 1));
                  double kq = 0.0;
                  for(int i = n - 2; i >= wq; --i)
                  {
                    const double p_i = opaque_v(bcast_lane(gd, i));
                    kq = (lane == i) ? rr : kq;
                    rr = givens_r_only(p_i, rr);
                  }
                  rho = rr;
                  GivensHalf h;
                  (void)givens_r(gd, kq, h);
                  givens_cs(h, cl, sl); // lane i now holds (c_i, s_i)
                  PROF_COUNT(13)
                }
                STAMP(4)
                if(mine) *reinterpret_cast<double2 *>(csb + 2 * lane) = make_double2(cl, sl);
              }
              if(!aborted && lane == 0) ctl->rho = rho;
              ws_post(&ctl->fcs, sq);
              STAMP(5)
            }
            else if(wave == 1)
            {
              // ---- K6: z = J2 d2; then z.z and n+.z as C4 trees over the lanes (CANON C5) and, from them and n+.x, the primal
              // step length t2 (GoldfarbIdnaniSolver.cpp:174-218): MAIN only takes the minimum with t1.  n+.x is not computed here
              // in a pass that selected p: it is the scan's own chain for that constraint (cxw); behind a removal it was recomputed
              // beside MAIN's removal (cx_next below), off the critical path.
              const bool general = wp < m;
              const int wps = uniform(ctl->wps);
              const int rl = (lane < n) ? lane : 0;
              const double npv = general ? np_ptr(wp)[rl] : 0.0;
              double zi = chain_up(J + rl, ld, ds, 1, wq, n, 0.0);
              zi = (lane < n) ? zi : 0.0;
              if(lane < n) zs[lane] = zi;
              const double zz = wave_sum_tree(FMA(zi, zi, 0.0));
              const double czv = general ? wave_sum_tree(FMA(npv, zi, 0.0)) : bcast_lane(zi, wp - m);
              double cxv;
              if(uniform(ctl->wfresh))
                cxv = general ? uniform_d(ctl->cxw[(wp < CREG_MAX) ? (wp >> 6) : 3]) : uniform_d(xs[wp - m]);
              else
                cxv = cx_next;
              double t2v = big;
              if(__builtin_sqrt(zz) > 1e-14) t2v = (uniform_d(bound_of(wp, wps)) - cxv) / czv;
              if(lane == 0)
              {
                ctl->cz = czv;
                ctl->t2 = t2v;
              }
              ws_post(&ctl->fz, sq);
            }
            else
            {
              // (wave 2 has nothing to do inside a pass: it only scans)
            }
            // ---- the decision
            int fs;
            spins = 0;
            while((fs = __atomic_load_n(&ctl->fstep, __ATOMIC_RELAXED)) != sq && fs != -1)
            {
              if(++spins > SPIN_LIMIT) break;
              __builtin_amdgcn_s_sleep(NAP);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if(fs != sq) break;
            if(uniform(ctl->dec) == 1)
            {
              ++sc;
              STAMP(6)
              STAMP2(16)
              ws_scan(wp); // wp is treated as active (MAIN writes its status byte meanwhile)
              ws_post(&ctl->fscan[wave], sc);
              STAMP(7)
              STAMP2(17)
            }
            else if(wave == 0)
              ws_post(&ctl->fack0, sq);
            else if(wave == 2)
              ws_post(&ctl->fack2, sq);
            else
            {
              // wave 1, a pass that removes: the next pass keeps p but x has moved -- n+.x again (C1 chain, every lane the same
              // chain on broadcast reads), beside MAIN's removal
              cx_next = (wp < m) ? chain_up(np_ptr(wp), 1, xs, 1, 0, n, 0.0) : xs[wp - m];
            }
          }
        }
        __syncthreads();
        q = uniform(ctl->wq);
        it = uniform(ctl->wit);
        term = uniform(ctl->wterm);
      }
      else
      {
        // Step 1 for the first pass: everybody scans
        {
          double bv;
          int bi, bs;
          scan_local(-1, bv, bi, bs);
          scan_publish(-1, bv, bi, bs);
          __syncthreads();
          scan_combine(p, pstat);
          STAMP(3)
        }
        bool running = (p >= 0);
        if(running)
        {
          publish_np(p, pstat, true);
          if(tid == 0) us[q] = 0.0; // DualSolver.cpp:121
        }
        while(running)
        {
          // ---- selected constraint (p, pstat): n+ to LDS, u[q] = 0 only when it was freshly selected
          // (callers below take care of u[q]); K5
          __syncthreads();
          STAMP(20)
          if(wave == 0) compute_d(p, pstat);
          STAMP(21)
          __syncthreads();
          STAMP(4)
          // SPEC: d is all the coefficient recurrence of an addition needs, so wave 3 starts on it now, before the step
          // length says whether this pass adds or drops; it runs as many steps as the other waves' work is expected
          // to last (so that it neither holds the barrier up nor idles), and the result is simply discarded on a drop
          if(SPEC && wave == 3)
          {
            STAMP3(24)
            chain_steps(SPEC_A0 + (q >> 1), true);
            STAMP3(25)
          }
          else if(wave == 0)
            trsv_ratio();
          else if(wave <= 2)
            z_and_dots(p);
          __syncthreads();
          STAMP(5)
          if(SPEC && wave == 3)
          {
            STAMP3(26)
            chain_steps(SPEC_B); // beside the step-length computation and wave 0's primal/dual update
            STAMP3(27)
          }

          // ---- K8: step length (uniform, every thread)
          const double cz = uniform_d(ctl->cz);
          const double dotsc = (pstat == UPPER || pstat == UPPER_BOUND) ? -cz : cz; // ConstraintNormal::dot
          const double t1 = uniform_d(ctl->t1);
          const int l = uniform(ctl->l);
          double t2 = big;
          if(__builtin_sqrt(uniform_d(ctl->zz)) > 1e-14)
          {
            t2 = (uniform_d(bound_of(p, pstat)) - uniform_d(ctl->cx)) / cz;
          }
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
            term = T_INFEASIBLE; // DualSolver.cpp:131
            break;
          }
          const bool dual_only = (t2 >= big);
          const bool add = !dual_only && (t == t2);

          if(wave == 0) apply_step(t, dotsc, dual_only);
          STAMP(6)
          if(add)
          {
            __syncthreads(); // x is final: the other waves may scan for the NEXT pass during the rest of the sweep
            double bv;
            int bi, bs;
            ++addno;
            if(SPEC)
            {
              if(wave == 3)
              {
                STAMP3(28)
                chain_steps(n);
                STAMP3(29)
                chain_finish(p, pstat, addno);
                wave_sync();
                STAMP3(30)
                add_apply(q, -1); // same wave: the sweep of J follows at once, beside the other waves' scan
                STAMP3(31)
              }
            }
            else if(wave == 0)
            {
              add_chain(p, pstat, addno);
              wave_sync();
            }
            const int qa = q;
            q = uniform(q + 1);
            ++it;
            scan_local(p, bv, bi, bs); // p is treated as active (wave 0 has it in st[] by now, the others exclude it)
            scan_publish(p, bv, bi, bs);
            STAMP(15)
            if(!SPEC && wave == 1) add_apply(qa, addno); // overlaps wave 0's share of the scan
            PROF_COUNT(10)
            __syncthreads();
            STAMP(7)
            if(it >= A.max_iter)
            {
              term = T_MAX_ITER; // checked before the next selection (for(; it_ < maxIter_; ...), DualSolver.cpp:103)
              break;
            }
            scan_combine(p, pstat);
            STAMP(22)
            if(p < 0) break; // SUCCESS (DualSolver.cpp:114)
            publish_np(p, pstat, true);
            if(tid == 0) us[q] = 0.0; // DualSolver.cpp:121
            STAMP(23)
          }
          else
          {
            if(wave == 0) do_drop(l);
            q = uniform(q - 1);
            ++it;
            PROF_COUNT(11)
            STAMP(8)
            if(it >= A.max_iter)
            {
              term = T_MAX_ITER;
              break;
            }
            // skipStep1: same (p, pstat), u[q] keeps the shifted pending multiplier
          }
        }
      }
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- results (K13: sign-expanded multipliers)
  if(tid < nr) A.x[(size_t)b * nr + tid] = xs[tid];
  double * ug = A.u + (size_t)b * mt;
  int * sg = A.status + (size_t)b * mt;
  for(int i = tid; i < mt; i += WG)
  {
    ug[i] = 0.0;
    sg[i] = st[i];
  }
  __syncthreads();
  if(tid < q)
  {
    int i = act[tid];
    int s = st[i];
    ug[i] = (s == UPPER || s == UPPER_BOUND) ? us[tid] : -us[tid];
  }
  if(tid == 0)
  {
    A.term[b] = term;
    A.iters[b] = it;
    A.f[b] = ctl->f;
  }
  STAMP(9)
  if(PROF)
  {
    __syncthreads();
    if(tid < 32) A.prof[(size_t)b * A.prof_stride + tid] = (long long)prof_lds[tid];
    if(tid == 0) A.prof[(size_t)b * A.prof_stride + 12] = wall_clock64() - wall0_;
  }
}

template<int NT, int CMODE, bool PROF, bool WARM, bool EMB = false, bool BOX = false>
hipError_t launch_one(const BatchArgs & a, long lds, hipStream_t stream)
{
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gi_batch_kernel<NT, CMODE, PROF, WARM, EMB, BOX>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if(e != hipSuccess) return e;
  hipLaunchKernelGGL((gi_batch_kernel<NT, CMODE, PROF, WARM, EMB, BOX>), dim3(a.B), dim3(WG), (size_t)lds, stream, a);
  return hipGetLastError();
}

template<int NT, int CMODE>
hipError_t launch_prof(const BatchArgs & a, long lds, hipStream_t stream)
{
#ifdef JRLQP_EXP_ONLY_C3 // experimental fast build (tools/ubench/exp_c3.sh): the cold n = 50 kernel alone, never the product library
  if constexpr(NT != 50)
    return hipErrorInvalidValue;
  else
  {
    if(a.n != NT || a.warm) return hipErrorInvalidValue;
#ifdef JRLQP_EXP_NOPROF // (register-pressure experiments: the production instantiation alone, 8 s per build)
    return launch_one<NT, CMODE, false, false>(a, lds, stream);
#else
    return a.prof ? launch_one<NT, CMODE, true, false>(a, lds, stream) : launch_one<NT, CMODE, false, false>(a, lds, stream);
#endif
  }
#else
  if(NT > 0 && a.n != NT) // a smaller problem, embedded (no phase-profile build of these)
    return a.warm ? launch_one<NT, CMODE, false, true, true>(a, lds, stream) : launch_one<NT, CMODE, false, false, true>(a, lds, stream);
#ifdef JRLQP_WARM_PROF // (diagnostic build: the phases of a warm start in the generic kernel -- slots 24 .. 27, tools/time_big_family.py warm3 <n>)
  if constexpr(NT == 0)
    if(a.warm && a.prof) return launch_one<NT, CMODE, true, true>(a, lds, stream);
#endif
  if(a.warm) return launch_one<NT, CMODE, false, true>(a, lds, stream); // the phase profile covers the cold path only
  return a.prof ? launch_one<NT, CMODE, true, false>(a, lds, stream) : launch_one<NT, CMODE, false, false>(a, lds, stream);
#endif
}

} // namespace
} // namespace jrlqp
