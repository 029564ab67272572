// device_common.h -- device-side helpers shared by every kernel file (gfx950, wave64).
//
// Everything here is part of the CANON rounding contract: explicit fmas, correctly rounded / and
// sqrt, no contraction (the translation units are compiled with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

namespace jrlqp
{
#define FMA(a, b, c) __builtin_fma((a), (b), (c))
#define AT(P, ld, i, j) (P)[(i) + (size_t)(j) * (ld)]

// single-wave producer/consumer through LDS (or through memory the wave itself wrote): the hardware executes one
// wave's accesses in order, so only the compiler has to be told not to move accesses across this point
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier for exchanges that go through LDS only: waits for this wave's LDS traffic, NOT for its outstanding
// global loads (__syncthreads() carries a release fence, i.e. s_waitcnt vmcnt(0): every prefetch in flight would be
// drained at each reduction).  Global data produced for other waves still needs __syncthreads().
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// value of v in lane `src` (src wave-uniform): two v_readlane_b32, no LDS round trip.
// HAZARD NOTE (gfx950, hipcc 7.2, measured): when the compiler cannot prove `src` uniform it fetches the lane number with
// v_readfirstlane_b32 on the spot and puts the four wait states its hazard table asks for between that VALU write of an SGPR
// and the v_readlane that selects with it.  In a substitution loop whose counter lived in a VGPR that was not always enough:
// now and then a stale lane came back (9-18 of 61 parity tests of the batched kernel failing, different problems every run,
// only in the instantiations whose schedule put exactly four instructions there; with five explicit wait states, or with a
// scalar counter, every run was clean).  Rule: loops that broadcast by a running index make their bound uniform BEFORE the
// loop (`uniform()` / v_readfirstlane once), so that the SALU produces every lane number -- bcast_lane_s marks those sites.
__device__ __forceinline__ double bcast_lane(double v, int src)
{
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// the same, at a site whose `src` is a compile-time constant or SALU-produced (see the note above)
__device__ __forceinline__ double bcast_lane_s(double v, int src)
{
  return bcast_lane(v, src);
}

// acc <- fma(-v, (src of lane LANE), acc) as ONE unit the compiler cannot pull apart: left to itself hipcc hoists the
// v_readlane pairs of a whole column in front of the fmas and then spills the scalar results through VGPR lanes (four more
// instructions per term, measured: the register kernel below no faster than the barrier kernel).
// GUARD: the unit starts with a wait state.  The hardware wants one between a VALU write of a VGPR and a v_readlane of it, the
// compiler inserts it for its own instructions but cannot see into the unit: a unit issued right behind the v_cndmask that
// produced `src` read a stale low half (measured: factors off by 1e-8).  The first unit behind a new `src` is guarded.
template<int LANE, bool GUARD = false>
__device__ __forceinline__ void fma_neg_bcast(double & acc, double v, double src)
{
  const int lo = __double2loint(src), hi = __double2hiint(src);
  if constexpr(GUARD)
    asm volatile("s_nop 1\n\t"
                 "v_readlane_b32 s90, %2, %4\n\t"
                 "v_readlane_b32 s91, %3, %4\n\t"
                 "s_nop 1\n\t"
                 "v_fma_f64 %0, -%1, s[90:91], %0\n\t"
                 "s_nop 0"
                 : "+v"(acc)
                 : "v"(v), "v"(lo), "v"(hi), "n"(LANE)
                 : "s90", "s91");
  else
    asm volatile("v_readlane_b32 s90, %2, %4\n\t"
                 "v_readlane_b32 s91, %3, %4\n\t"
                 "s_nop 1\n\t"
                 "v_fma_f64 %0, -%1, s[90:91], %0"
                 : "+v"(acc)
                 : "v"(v), "v"(lo), "v"(hi), "n"(LANE)
                 : "s90", "s91");
}
__device__ __forceinline__ double shfl_xor_d(double v, int mask)
{
  int lo = __shfl_xor(__double2loint(v), mask);
  int hi = __shfl_xor(__double2hiint(v), mask);
  return __hiloint2double(hi, lo);
}

// keeps a wave-uniform value in a VGPR and hides its provenance from the compiler: without this hipcc scalarises
// parts of a dependent chain and every select costs a VALU->SALU->VALU round trip
__device__ __forceinline__ double opaque_v(double v)
{
  asm volatile("" : "+v"(v));
  return v;
}

// ---- DPP lane exchanges (no LDS round trip) and the in-wave part of the C4 reduction tree ---------------------------
template<int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v)
{
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// DPP move with a fallback value: lanes whose source lane does not exist (wave shifts: the first / last lane) get `old`
template<int CTRL>
__device__ __forceinline__ double dpp_shift_d(double v, double old)
{
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// DPP move, lanes without a source lane get zero (bound_ctrl): no preload of the destination, two instructions in all
template<int CTRL>
__device__ __forceinline__ double dpp_shift0_d(double v)
{
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// minimum over the 64 lanes, every lane gets it: four DPP exchanges inside each row of 16, then the four row
// minima through readlane -- no LDS crossbar (ds_bpermute) round trips
template<int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min_d(double v)
{
  double o;
  o = dpp_move<0xB1>(v); // quad_perm [1,0,3,2]
  v = (o < v) ? o : v;
  o = dpp_move<0x4E>(v); // quad_perm [2,3,0,1]
  v = (o < v) ? o : v;
  o = dpp_move<0x141>(v); // row_half_mirror
  v = (o < v) ? o : v;
  o = dpp_move<0x140>(v); // row_mirror
  v = (o < v) ? o : v;
  double r0 = bcast_lane(v, 0), r1 = bcast_lane(v, 16), r2 = bcast_lane(v, 32), r3 = bcast_lane(v, 48);
  double m01 = (r1 < r0) ? r1 : r0;
  double m23 = (r3 < r2) ? r3 : r2;
  return (m23 < m01) ? m23 : m01;
}

// the same for ints
template<int CTRL>
__device__ __forceinline__ int dpp_move_i(int v)
{
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_min_i(int v)
{
  int o;
  o = dpp_move_i<0xB1>(v);
  v = (o < v) ? o : v;
  o = dpp_move_i<0x4E>(v);
  v = (o < v) ? o : v;
  o = dpp_move_i<0x141>(v);
  v = (o < v) ? o : v;
  o = dpp_move_i<0x140>(v);
  v = (o < v) ? o : v;
  const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  const int m01 = (r1 < r0) ? r1 : r0, m23 = (r3 < r2) ? r3 : r2;
  return (m23 < m01) ? m23 : m01;
}

// Sum over the 64 lanes as an ADJACENT-FIRST pairwise tree (lane 2i with 2i+1, then pairs of pairs, ...): rule C4 of
// the oracle (oracle/blockgi_oracle.c orc_dot_tree).  Every lane returns the same bits: each level adds the totals of
// two sibling groups, and a + b == b + a.  quad_perm / row_half_mirror / row_mirror inside a row of 16, then the four
// row totals through v_readlane.
__device__ __forceinline__ double wave_sum_tree(double s)
{
  s = s + dpp_mov_d<0xB1>(s);  // lane ^ 1   (quad_perm [1,0,3,2])
  s = s + dpp_mov_d<0x4E>(s);  // lane ^ 2   (quad_perm [2,3,0,1])
  s = s + dpp_mov_d<0x141>(s); // the other quad of each group of 8 (row_half_mirror)
  s = s + dpp_mov_d<0x140>(s); // the other half of each row of 16 (row_mirror)
  const double r0 = bcast_lane(s, 0), r1 = bcast_lane(s, 16), r2 = bcast_lane(s, 32), r3 = bcast_lane(s, 48);
  return (r0 + r1) + (r2 + r3);
}

// Eigen JacobiRotation<double>::makeGivens(p, q, &r), real case (reference call sites GoldfarbIdnaniSolver.cpp:229,249;
// oracle orc_givens):  [c -s; s c] [p; q] = [r; 0]
__device__ __forceinline__ void make_givens(double p, double q, double & c, double & s, double & r)
{
  if(q == 0.0)
  {
    c = p < 0.0 ? -1.0 : 1.0;
    s = 0.0;
    r = fabs(p);
  }
  else if(p == 0.0)
  {
    c = 0.0;
    s = q < 0.0 ? 1.0 : -1.0;
    r = fabs(q);
  }
  else if(fabs(p) > fabs(q))
  {
    double t = q / p;
    double u = __builtin_sqrt(FMA(t, t, 1.0));
    if(p < 0.0) u = -u;
    c = 1.0 / u;
    s = -t * c;
    r = p * u;
  }
  else
  {
    double t = p / q;
    double u = __builtin_sqrt(FMA(t, t, 1.0));
    if(q < 0.0) u = -u;
    s = -1.0 / u;
    c = -t * s;
    r = q * u;
  }
}

// ---- CANON C7 (oracle/jrlqp_oracle.c orc_givens_chain): the coefficients of an addition's sweep from a suffix sum of squares ----
// value of lane + off (lanes whose partner is past lane 63 get their own value back: callers mask them)
__device__ __forceinline__ double shfl_down_d(double v, int off)
{
  int lo = __shfl_down(__double2loint(v), off);
  int hi = __shfl_down(__double2hiint(v), off);
  return __hiloint2double(hi, lo);
}
// may every entry of the chain be squared?  (0, or 2^-500 <= |d| <= 2^500; NaN / Inf: no)
__device__ __forceinline__ bool c7_in_range(double d)
{
  const double a = fabs(d);
  return a == 0.0 || (a >= 0x1p-500 && a <= 0x1p500);
}
// One block of the scan on a wave whose lane `lane` holds entry j = top - lane of the block (top = the block's lane of j = 0, i.e. the
// HIGHEST index k of the block sits in lane `top`; lanes above `top` are not part of the block): x_j += x_{j-off} <=> lane += lane + off.
__device__ __forceinline__ double c7_block_scan(double e, int lane, int top)
{
#pragma unroll
  for(int off = 1; off < 64; off <<= 1)
  {
    const double o = shfl_down_d(e, off);
    e = (lane + off <= top) ? e + o : e;
  }
  return e;
}
// The whole chain on one wave when n <= 64 (lane = index k, dk = d_k, anything in lanes >= n): rotation k's (c, s) for q <= k <= n-2
// and rq = r_q.  false: an entry is out of range -- nothing is computed, the caller runs the serial recurrence.
__device__ __forceinline__ bool c7_wave(double dk, int lane, int n, int q, double & c, double & s, double & rq)
{
  const bool in = lane >= q && lane < n;
  if(!__all(!in || c7_in_range(dk))) return false;
  const double S = c7_block_scan(in ? dk * dk : 0.0, lane, n - 1);
  double rho = __builtin_sqrt(S);
  rho = (lane == n - 1) ? dk : rho; // rho_0: the signed last entry, as the first makeGivens sees it
  const double rup = dpp_shift0_d<0x130>(rho); // lane k takes lane k+1's radius
  const bool z = (rho == 0.0);
  const double cq = dk / rho, sq = (-rup) / rho;
  c = z ? 1.0 : cq;
  s = z ? 0.0 : sq;
  rq = bcast_lane(rho, q);
  return true;
}

// sqrt(x) for x in [1, 2] (x = 1 + t^2, |t| <= 1): the Goldschmidt sequence hipcc expands __builtin_sqrt(double)
// into, minus its range scaling (ldexp by 0 here) and its inf/zero class fix-up -- bit-identical on this range,
// 9 dependent ops instead of 16 and no VALU->SALU round trip.
__device__ __forceinline__ double sqrt_1to2(double x)
{
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  double r = FMA(-h, g, 0.5);
  g = FMA(g, r, g);
  h = FMA(h, r, h);
  double d = FMA(-g, g, x);
  g = FMA(d, h, g);
  d = FMA(-g, g, x);
  g = FMA(d, h, g);
  return g;
}

// r of makeGivens(p, q) alone -- what the NEXT rotation of a sweep depends on.  r = |den| * sqrt(1 + t^2) with
// |t| = min(|p|,|q|) / max(|p|,|q|): the signs of t and u cancel in den*u and t enters only squared, so working on
// magnitudes gives the same bits as the signed formula; p == 0 or q == 0 fall out exactly (t = 0, sqrt = 1).
__device__ __forceinline__ double givens_r_only(double p, double q)
{
  const double a = fabs(p), b = fabs(q);
  const double hi = fmax(a, b), lo = fmin(a, b);
  const double t = lo / hi;
  const double r = hi * sqrt_1to2(FMA(t, t, 1.0));
  return (hi == 0.0) ? 0.0 : r;
}

// Division with the denominator-only part hoisted.  hipcc expands a/b (f64) into
//   s0 = div_scale(b,b,a); r = rcp(s0); two Newton steps on r; s1 = div_scale(a,b,a); q0 = s1*r; e = fma(-s0,q0,s1);
//   q = div_fmas(e, r, q0); div_fixup(q, b, a)
// and div_scale / div_fmas / div_fixup are identities unless an operand is zero/denormal/huge (exponent beyond
// +-768 or so).  Inside the safe range below the quotient is therefore exactly fma(e, r, q0) with r depending on the
// DENOMINATOR ONLY: a substitution chain can precompute r for all its pivots at once (one lane per pivot) and keep
// three dependent ops per step instead of thirteen.  Outside the safe range the caller falls back to a/b, so the
// result is bit-identical to the plain division in every case.
__device__ __forceinline__ double recip_refined(double den)
{
  double r = __builtin_amdgcn_rcp(den);
  double e = FMA(-den, r, 1.0);
  r = FMA(r, e, r);
  e = FMA(-den, r, 1.0);
  r = FMA(r, e, r);
  return r;
}
__device__ __forceinline__ double div_pre(double num, double den, double rinv)
{
  const double q0 = num * rinv;
  const double e = FMA(-den, q0, num);
  return FMA(e, rinv, q0);
}
__device__ __forceinline__ bool div_safe_den(double d)
{
  const double a = fabs(d);
  return a > 0x1p-400 && a < 0x1p400;
}
__device__ __forceinline__ bool div_safe_num(double v)
{
  const double a = fabs(v);
  return (a > 0x1p-400 && a < 0x1p400) || v == 0.0;
}


// make_givens with the same operations per result and fewer instructions on the way (round 4: the rotations of a REMOVAL are a serial
// recurrence, one Givens generator per step): the square root's argument 1 + t^2 lies in [1, 2] (sqrt_1to2: the compiler's sequence
// without its range scaling), 1 / u has |u| in [1, sqrt 2] and the quotient t = q / p (or p / q) of operands inside the safe range is
// div_pre's (the compiler's own division with the scaling and fix-up steps that are identities there).  Operands outside the safe
// range take the plain generator.  Bit-identical to make_givens (tests/test_gpu_arith.py drives both over random and edge operands).
__device__ __forceinline__ void make_givens_lean(double p, double q, double & c, double & s, double & r)
{
  if(q == 0.0 || p == 0.0 || !div_safe_den(p) || !div_safe_den(q))
  {
    make_givens(p, q, c, s, r);
    return;
  }
  const bool pbig = fabs(p) > fabs(q);
  const double num = pbig ? q : p, den = pbig ? p : q;
  const double t = div_pre(num, den, recip_refined(den));
  double u = sqrt_1to2(FMA(t, t, 1.0));
  if(den < 0.0) u = -u;
  const double iu = div_pre(1.0, u, recip_refined(u));
  if(pbig)
  {
    c = iu;
    s = -t * c;
  }
  else
  {
    s = -iu;
    c = -t * s;
  }
  r = den * u;
}

// ---- wave-level substitutions on one block (shared by the structured Cholesky family and the structured GI loop) ----
constexpr int kAhead = 16; // operands of a dependent chain are requested this many steps ahead (one HBM/L2 latency per group)

// x (n <= 64) <- L^-1 x, one element per lane (forward substitution, column oriented; Eigen triangularView<Lower>::
// solveInPlace call sites blockArrowLLT.cpp:139,151, triBlockDiagLLT.cpp:77,87).  L is read from HBM/L2: the column
// entries are fetched kAhead steps ahead with unconditional (in-footprint) loads so that one memory latency is paid
// per group, and the pivots' reciprocals are hoisted (div_pre): ~4 dependent fp64 ops per step instead of ~14.
// A post-hoc range check licenses the hoisted reciprocals; outside it the chain is redone with plain a/b.
__device__ __forceinline__ double wave_fwd(const double * L, int ldl, int n, double xi, int lane)
{
  n = __builtin_amdgcn_readfirstlane(n); // (scalar loop counters: see bcast_lane)
  const bool act = lane < n;
  const int rl = act ? lane : 0;
  const double dg = (n > 0) ? AT(L, ldl, rl, rl) : 1.0;
  const double rinv = recip_refined(dg);
  const double x0 = xi;
  // the operands of group g + 1 are requested before group g is consumed: one exposed memory latency per CALL
  double lv[kAhead], ln[kAhead];
#pragma unroll
  for(int j = 0; j < kAhead; ++j) ln[j] = (n > 0) ? AT(L, ldl, rl, min(j, n - 1)) : 0.0;
  for(int k0 = 0; k0 < n; k0 += kAhead)
  {
#pragma unroll
    for(int j = 0; j < kAhead; ++j) lv[j] = ln[j];
    if(k0 + kAhead < n)
    {
#pragma unroll
      for(int j = 0; j < kAhead; ++j) ln[j] = AT(L, ldl, rl, min(k0 + kAhead + j, n - 1));
    }
    // One step: the quotient of lane k -- three dependent operations, computed by EVERY lane on its own entry, so neither
    // the pivot nor its reciprocal is broadcast --, a v_readlane of it, one fma and one select for the lanes below.  Lane k
    // itself is left alone: after step k - 1 its entry IS the numerator of x_k, and all quotients are formed together
    // after the last step.  No branch inside a group (a uniform branch per step costs a pipeline refill and a wait for every
    // load in flight: 187 cycles per step measured, against ~50): a step beyond n - 1 only touches lanes >= n, whose
    // values nobody reads.
#pragma unroll
    for(int j = 0; j < kAhead; ++j)
    {
      const int k = k0 + j;
      const double v = bcast_lane_s(div_pre(xi, dg, rinv), min(k, 63));
      const double upd = FMA(-lv[j], v, xi);
      xi = (lane > k) ? upd : xi;
    }
  }
  const double pre = xi; // the numerators
  xi = div_pre(pre, dg, rinv);
  if(!__all(!act || (div_safe_den(dg) && div_safe_num(pre))))
  {
    xi = x0;
    for(int k = 0; k < n; ++k)
    {
      const double v = bcast_lane(xi, k) / bcast_lane(dg, k);
      if(lane == k)
        xi = v;
      else if(lane > k && act)
        xi = FMA(-AT(L, ldl, rl, k), v, xi);
    }
  }
  return act ? xi : x0;
}
// x <- L^-T x (backward substitution; call sites blockArrowLLT.cpp:211,241,248, triBlockDiagLLT.cpp:139,149)
__device__ __forceinline__ double wave_bwd(const double * L, int ldl, int n, double xi, int lane)
{
  n = __builtin_amdgcn_readfirstlane(n); // (scalar loop counters: see bcast_lane)
  const bool act = lane < n;
  const int rl = act ? lane : 0;
  const double dg = (n > 0) ? AT(L, ldl, rl, rl) : 1.0;
  const double rinv = recip_refined(dg);
  const double x0 = xi;
  double lv[kAhead], ln[kAhead];
#pragma unroll
  for(int j = 0; j < kAhead; ++j) ln[j] = (n > 0) ? AT(L, ldl, max(n - 1 - j, 0), rl) : 0.0;
  for(int k0 = n - 1; k0 >= 0; k0 -= kAhead)
  {
#pragma unroll
    for(int j = 0; j < kAhead; ++j) lv[j] = ln[j];
    if(k0 - kAhead >= 0)
    {
#pragma unroll
      for(int j = 0; j < kAhead; ++j) ln[j] = AT(L, ldl, max(k0 - kAhead - j, 0), rl);
    }
    // (steps as in wave_fwd, mirrored: lanes above k are updated; a step with k < 0 matches no lane)
#pragma unroll
    for(int j = 0; j < kAhead; ++j)
    {
      const int k = k0 - j;
      const double v = bcast_lane_s(div_pre(xi, dg, rinv), max(k, 0));
      const double upd = FMA(-lv[j], v, xi);
      xi = (lane < k) ? upd : xi;
    }
  }
  const double pre = xi;
  xi = div_pre(pre, dg, rinv);
  if(!__all(!act || (div_safe_den(dg) && div_safe_num(pre))))
  {
    xi = x0;
    for(int k = n - 1; k >= 0; --k)
    {
      const double v = bcast_lane(xi, k) / bcast_lane(dg, k);
      if(lane == k)
        xi = v;
      else if(lane < k)
        xi = FMA(-AT(L, ldl, k, rl), v, xi);
    }
  }
  return act ? xi : x0;
}

// wave_fwd / wave_bwd with the NP operands of each lane supplied by the caller (n <= NP <= 64), who requested them -- all at once, and
// beside whatever else the block needs -- before anything depended on them: one memory round trip per block instead of one per group
// of kAhead steps.  Same steps, same bits.  fwd: lv[j] = L(rl, min(j, n-1)); bwd: lv[j] = L(max(n-1-j, 0), rl); dg = L(rl, rl); rl = lane
// (lanes >= n: 0).
template<int NP>
__device__ __forceinline__ double wave_fwd_pre(const double (&lv)[NP], double dg, const double * L, int ldl, int n, double xi, int lane)
{
  n = __builtin_amdgcn_readfirstlane(n);
  const bool act = lane < n;
  const int rl = act ? lane : 0;
  const double rinv = recip_refined(dg);
  const double x0 = xi;
#pragma unroll
  for(int j = 0; j < NP; ++j)
  {
    const double v = bcast_lane_s(div_pre(xi, dg, rinv), j);
    const double upd = FMA(-lv[j], v, xi);
    xi = (lane > j) ? upd : xi;
  }
  const double pre = xi;
  xi = div_pre(pre, dg, rinv);
  if(!__all(!act || (div_safe_den(dg) && div_safe_num(pre))))
  {
    xi = x0;
    for(int k = 0; k < n; ++k)
    {
      const double v = bcast_lane(xi, k) / bcast_lane(dg, k);
      if(lane == k)
        xi = v;
      else if(lane > k && act)
        xi = FMA(-AT(L, ldl, rl, k), v, xi);
    }
  }
  return act ? xi : x0;
}
template<int NP>
__device__ __forceinline__ double wave_bwd_pre(const double (&lv)[NP], double dg, const double * L, int ldl, int n, double xi, int lane)
{
  n = __builtin_amdgcn_readfirstlane(n);
  const bool act = lane < n;
  const int rl = act ? lane : 0;
  const double rinv = recip_refined(dg);
  const double x0 = xi;
#pragma unroll
  for(int j = 0; j < NP; ++j)
  {
    const int k = n - 1 - j; // (a step with k < 0 matches no lane)
    const double v = bcast_lane_s(div_pre(xi, dg, rinv), max(k, 0));
    const double upd = FMA(-lv[j], v, xi);
    xi = (lane < k) ? upd : xi;
  }
  const double pre = xi;
  xi = div_pre(pre, dg, rinv);
  if(!__all(!act || (div_safe_den(dg) && div_safe_num(pre))))
  {
    xi = x0;
    for(int k = n - 1; k >= 0; --k)
    {
      const double v = bcast_lane(xi, k) / bcast_lane(dg, k);
      if(lane == k)
        xi = v;
      else if(lane < k)
        xi = FMA(-AT(L, ldl, k, rl), v, xi);
    }
  }
  return act ? xi : x0;
}

// acc <- fma(-a(k), b(k), acc) for k = k0, k0 + step, ... (cnt terms, in that order): the a() operands (HBM/L2) are
// fetched kAhead terms ahead with unconditional loads (the index is clamped into the valid range).  Two operand buffers
// take turns (no copy between them); a full group is sixteen plain fmas, only the last, partial group selects per term (no
// branch per term: see wave_fwd).
template<int G, class FA, class FB>
__device__ __forceinline__ double chain_group(int k0, int step, int cnt, int c0, double acc, const double (&av)[G], FB fb)
{
  if(c0 + G <= cnt)
  {
#pragma unroll
    for(int j = 0; j < G; ++j) acc = FMA(-av[j], fb(k0 + step * (c0 + j)), acc);
  }
  else
  {
#pragma unroll
    for(int j = 0; j < G; ++j)
    {
      const double t = FMA(-av[j], fb(k0 + step * min(c0 + j, cnt - 1)), acc);
      acc = (c0 + j < cnt) ? t : acc;
    }
  }
  return acc;
}
template<class FA, class FB>
__device__ __forceinline__ double chain_neg8(int k0, int step, int cnt, double acc, FA fa, FB fb)
{
  if(cnt <= 0) return acc;
  double a0[kAhead], a1[kAhead];
#pragma unroll
  for(int j = 0; j < kAhead; ++j) a0[j] = fa(k0 + step * min(j, cnt - 1));
  for(int c0 = 0; c0 < cnt; c0 += 2 * kAhead)
  {
    if(c0 + kAhead < cnt)
    {
#pragma unroll
      for(int j = 0; j < kAhead; ++j) a1[j] = fa(k0 + step * min(c0 + kAhead + j, cnt - 1));
    }
    acc = chain_group<kAhead, FA, FB>(k0, step, cnt, c0, acc, a0, fb);
    if(c0 + kAhead < cnt)
    {
      if(c0 + 2 * kAhead < cnt)
      {
#pragma unroll
        for(int j = 0; j < kAhead; ++j) a0[j] = fa(k0 + step * min(c0 + 2 * kAhead + j, cnt - 1));
      }
      acc = chain_group<kAhead, FA, FB>(k0, step, cnt, c0 + kAhead, acc, a1, fb);
    }
  }
  return acc;
}

// ---- blocks of 65 .. 128 rows (round 4): two entries per lane (rows lane and lane + 64), two 64-row stages ------------------------
// x = (a, b) <- L^-1 x, n <= 128.  Stage 1 is wave_fwd on the leading 64 x 64 block; then rows 64 .. n-1 receive their terms of the
// first 64 unknowns -- k ascending, the unknowns handed out by v_readlane, the coefficients fetched ahead --; stage 2 is wave_fwd on
// the trailing block.  Row e's chain is the sequential one (terms k = 0 .. e-1 in order), so the bits are those of a single
// substitution over all n rows.  n <= 64: exactly wave_fwd (b is not touched).
__device__ __forceinline__ void wave_fwd2(const double * L, int ldl, int n, double & a, double & b, int lane)
{
  n = __builtin_amdgcn_readfirstlane(n);
  if(n <= 64)
  {
    a = wave_fwd(L, ldl, n, a, lane);
    return;
  }
  a = wave_fwd(L, ldl, 64, a, lane);
  const int n1 = n - 64, rl = 64 + min(lane, n1 - 1);
  const double a0 = a;
  b = chain_neg8(0, 1, 64, b, [&](int k) { return AT(L, ldl, rl, k); }, [&](int k) { return bcast_lane(a0, k); });
  b = wave_fwd(L + 64 + (size_t)64 * ldl, ldl, n1, b, lane);
}
// x = (a, b) <- L^-T x, n <= 128: the trailing block first, then rows 0 .. 63 receive their terms of the unknowns 64 .. n-1 (k
// descending), then the leading block
__device__ __forceinline__ void wave_bwd2(const double * L, int ldl, int n, double & a, double & b, int lane)
{
  n = __builtin_amdgcn_readfirstlane(n);
  if(n <= 64)
  {
    a = wave_bwd(L, ldl, n, a, lane);
    return;
  }
  const int n1 = n - 64;
  b = wave_bwd(L + 64 + (size_t)64 * ldl, ldl, n1, b, lane);
  const double b0 = b;
  a = chain_neg8(n - 1, -1, n1, a, [&](int k) { return AT(L, ldl, k, lane); }, [&](int k) { return bcast_lane(b0, k - 64); });
  a = wave_bwd(L, ldl, 64, a, lane);
}

// same with BOTH operand streams fetched ahead (operands in LDS: a dependent loop that reads, waits and fmas one term at
// a time pays one LDS round trip per term; eight terms in flight hide it)
template<class FA, class FB>
__device__ __forceinline__ double chain2_neg(int k0, int step, int cnt, double acc, FA fa, FB fb)
{
  constexpr int W = 8;
  for(int c0 = 0; c0 < cnt; c0 += W)
  {
    double av[W], bv[W];
#pragma unroll
    for(int j = 0; j < W; ++j)
    {
      const int k = k0 + step * min(c0 + j, cnt - 1);
      av[j] = fa(k);
      bv[j] = fb(k);
    }
    if(c0 + W <= cnt)
    {
#pragma unroll
      for(int j = 0; j < W; ++j) acc = FMA(-av[j], bv[j], acc);
    }
    else
    {
#pragma unroll
      for(int j = 0; j < W; ++j)
      {
        const double t = FMA(-av[j], bv[j], acc);
        acc = (c0 + j < cnt) ? t : acc;
      }
    }
  }
  return acc;
}

// ---- in-register right-looking LLT of the n x n Hessian (n = N compile time), one wave, lane = row -------------------------
// The same scheme as the block-tridiagonal kernel's (structured.hip): a finished column goes to LDS; the term of column J + 1
// comes from a v_readlane unit so that the next pivot, its square root and its division start at once; the terms of the columns
// behind take L(c, J) back from LDS as broadcast reads, sixteen columns at a time (the register budget of this kernel), each
// group pinned where it is written.  Per element: the terms in ascending k, then one division -- the bits of the left-looking
// form it replaces (which walked the factored columns out of LDS for every block of eight: 64 k cycles at n = 50, this: 30 k).
template<int J, int N, int C0>
__device__ __forceinline__ void llt_reg_trail(double (&a)[N], double lij, const double * col)
{
  if constexpr(C0 < N)
  {
    constexpr int C1 = (C0 + 16 < N) ? C0 + 16 : N;
    double lc[16];
#pragma unroll
    for(int c = C0; c < C1; ++c) lc[c - C0] = col[c];
#pragma unroll
    for(int c = C0; c < C1; ++c) a[c] = FMA(-lij, lc[c - C0], a[c]);
#pragma unroll
    for(int c = C0; c < C1; ++c) asm volatile("" : "+v"(a[c]));
    llt_reg_trail<J, N, C1>(a, lij, col);
  }
}
template<int J, int N>
__device__ __forceinline__ void llt_reg_columns(double (&a)[N], int & info, int lane, double * L, int ld, double piv, double ljj, double lq)
{
  if constexpr(J < N)
  {
    if(piv <= 0.0)
    {
      info = J;
      return;
    }
    const double lij = (lane == J) ? ljj : lq;
    a[J] = lij;
    if(lane >= J && lane < N) L[lane + J * ld] = lij;
    double pn = 0.0, ln = 0.0, qn = 0.0;
    if constexpr(J + 1 < N)
    {
      fma_neg_bcast<J + 1, true>(a[J + 1], lij, lij);
      pn = bcast_lane(a[J + 1], J + 1);
      ln = __builtin_sqrt(opaque_v(pn)); // (a non-positive pivot gives a NaN nobody uses)
      qn = a[J + 1] / ln;
      asm volatile("" : "+v"(ln), "+v"(qn)); // (computed HERE: left alone, the compiler sinks both into the next column's block)
    }
    llt_reg_trail<J, N, J + 2>(a, lij, L + J * ld);
    llt_reg_columns<J + 1, N>(a, info, lane, L, ld, pn, ln, qn);
  }
}

// X <- X L^-T for one row of X per lane (x[c] = X(lane, c)), L (N x N, lower, leading dimension ld) in LDS: column J of X is
// final after one division, then gives the columns behind their term x[c] -= x(J) L(c, J) (L(c, J): a broadcast read).  Per
// element the terms in ascending k, then one division.
template<int J, int N>
__device__ __forceinline__ void trsm_reg_columns(double (&x)[N], const double * L, int ld)
{
  if constexpr(J < N)
  {
    const double * col = L + J * ld;
    double lc[N];
#pragma unroll
    for(int c = J; c < N; ++c) lc[c] = col[c];
    const double xj = x[J] / lc[J];
    x[J] = xj;
#pragma unroll
    for(int c = J + 1; c < N; ++c) x[c] = FMA(-xj, lc[c], x[c]);
#pragma unroll
    for(int c = J + 1; c < N; ++c) asm volatile("" : "+v"(x[c])); // (pinned to this column: see llt_reg_trail)
    trsm_reg_columns<J + 1, N>(x, L, ld);
  }
}

// logical diagonal block i of get<Up> (blockArrowLLT.cpp:22-46)
__device__ __forceinline__ int dsel(int nblk, int up, int i)
{
  return up ? (i + 1) % nblk : i;
}

} // namespace jrlqp
