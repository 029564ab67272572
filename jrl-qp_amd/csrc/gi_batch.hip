// gi_batch.hip -- Mode B: device-resident batched Goldfarb-Idnani solver for gfx950 (MI355X).
//
// One 256-thread workgroup (4 wave64) solves one dense QP end to end: the complete dual active-set state
// machine of the reference runs on-device out of LDS, inputs are read from HBM exactly once with coalesced
// loads and the results written once.
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
// LDS map per workgroup (doubles, ld = n|1 odd so that both "lane = row" and "lane = column" accesses of the
// n x n factors are bank-conflict free with ds_read_b64):
//   J[n*ld] | R[n*ld] (holds L during init) | x[n] z[n] d[n] u[n+2] r[n+2] | (C[m*ld] bl[m] bu[m] if staged)
//   | control block | active list int[n+1] | status bytes[m+nb]
#include <hip/hip_runtime.h>

#include "gi_batch.h"

namespace jrlqp
{
namespace
{
constexpr int WG = 256;

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
  T_OVERCONSTRAINED = 6
};

struct Ctl
{
  double red_val[4];
  double t1, zz, cz, cx;
  double f;
  int red_idx[4];
  int red_stat[4];
  int info;
  int l;
  int eq_next;
  int pad;
};

// ---- wave-level helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync()
{
  // single-wave producer/consumer through LDS: LDS executes one wave's accesses in order, so only the
  // compiler has to be told not to move accesses across this point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value of v in lane `src` (src must be wave-uniform): two v_readlane_b32, no LDS
__device__ __forceinline__ double bcast_lane(double v, int src)
{
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

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

__device__ __forceinline__ double shfl_xor_d(double v, int mask)
{
  int lo = __shfl_xor(__double2loint(v), mask);
  int hi = __shfl_xor(__double2hiint(v), mask);
  return __hiloint2double(hi, lo);
}

// lexicographic (value, index) minimum over the wave; every lane gets the winner
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

// Eigen JacobiRotation::makeGivens, real case (oracle orc_givens)
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
    double u = __builtin_sqrt(__builtin_fma(t, t, 1.0));
    if(p < 0.0) u = -u;
    c = 1.0 / u;
    s = -t * c;
    r = p * u;
  }
  else
  {
    double t = p / q;
    double u = __builtin_sqrt(__builtin_fma(t, t, 1.0));
    if(q < 0.0) u = -u;
    s = -1.0 / u;
    c = -t * s;
    r = q * u;
  }
}

#define FMA(a, b, c) __builtin_fma((a), (b), (c))

// ---- the kernel ------------------------------------------------------------------------------------------
// PROF: diagnostic build only -- thread 0 accumulates shader-clock cycles per phase into A.prof[b*16 + slot]
// (slots: 0 load, 1 LLT, 2 J/x0, 3 scan, 4 d=J^T n, 5 trsv||z, 6 step+update, 7 add, 8 drop, 9 store,
//  10 #adds, 11 #drops, 12 s_memrealtime ticks (100 MHz) for the whole solve).  Never used for a quoted time.
#define STAMP(slot)                                 \
  if(PROF && tid == 0)                              \
  {                                                 \
    long long now_ = clock64();                     \
    A.prof[(size_t)b * 16 + (slot)] += now_ - last_; \
    last_ = now_;                                   \
  }

template<bool C_LDS, bool PROF>
__global__ __launch_bounds__(WG) void gi_batch_kernel(const BatchArgs A)
{
  long long last_ = 0, wall0_ = 0;
  if(PROF)
  {
    last_ = clock64();
    wall0_ = wall_clock64();
  }
  extern __shared__ double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int b = blockIdx.x;
  const int n = A.n, m = A.m, nb = A.nb, ld = A.ld;
  const int mt = m + nb;

  // LDS carve-up (must match jrlqp::batch_lds_layout in gi_batch_host.cpp)
  double * J = smem;
  double * R = J + n * ld;
  double * xs = R + n * ld;
  double * zs = xs + n;
  double * ds = zs + n;
  double * us = ds + n;
  double * rs = us + (n + 2);
  double * tail = rs + (n + 2);
  double * Cs = tail;
  double * bls = tail;
  double * bus = tail;
  if(C_LDS)
  {
    bls = Cs + (size_t)m * ld;
    bus = bls + m;
    tail = bus + m;
  }
  Ctl * ctl = reinterpret_cast<Ctl *>(tail);
  int * act = reinterpret_cast<int *>(ctl + 1);
  unsigned char * st = reinterpret_cast<unsigned char *>(act + (n + 2));

  const double * Gg = A.G + (size_t)b * n * n;
  const double * ag = A.a + (size_t)b * n;
  const double * Cg = A.C + (size_t)b * n * m;
  const double * blg = A.bl + (size_t)b * m;
  const double * bug = A.bu + (size_t)b * m;
  const double * xlg = nb ? A.xl + (size_t)b * nb : nullptr;
  const double * xug = nb ? A.xu + (size_t)b * nb : nullptr;

  // ---------------------------------------------------------------- load G (coalesced) -> R region, status = 0
  for(int e = tid; e < n * n; e += WG)
  {
    int i = e % n, j = e / n;
    R[i + j * ld] = Gg[e];
  }
  for(int i = tid; i < mt; i += WG) st[i] = 0;
  if(tid < n) xs[tid] = 0.0;
  if(tid == 0)
  {
    ctl->info = -1;
    ctl->f = 0.0;
    ctl->eq_next = 0x7fffffff;
  }
  __syncthreads();
  STAMP(0)

  // ---------------------------------------------------------------- K1: LLT in LDS (wave 0), while waves 1-3 stage C
  double * L = R;
  if(wave == 0)
  {
    int info = -1;
    for(int j = 0; j < n; ++j)
    {
      double acc = 0.0;
      if(lane >= j && lane < n)
      {
        acc = L[lane + j * ld];
        for(int k = 0; k < j; ++k) acc = FMA(-L[lane + k * ld], L[j + k * ld], acc);
      }
      double piv = bcast_lane(acc, j);
      if(piv <= 0.0)
      {
        info = j;
        break;
      }
      double ljj = __builtin_sqrt(piv);
      if(lane == j)
        L[j + j * ld] = ljj;
      else if(lane > j && lane < n)
        L[lane + j * ld] = acc / ljj;
      wave_sync();
    }
    if(lane == 0) ctl->info = info;
  }
  else
  {
    if(C_LDS)
    {
      for(int e = tid - 64; e < n * m; e += WG - 64)
      {
        int k = e % n, i = e / n;
        Cs[(size_t)i * ld + k] = Cg[e];
      }
      for(int i = tid - 64; i < m; i += WG - 64)
      {
        bls[i] = blg[i];
        bus[i] = bug[i];
      }
    }
    // equalities / fixed variables present?  (GoldfarbIdnaniSolver.cpp:268-287)
    int first = 0x7fffffff;
    for(int i = tid - 64; i < mt; i += WG - 64)
    {
      bool eq = (i < m) ? (blg[i] == bug[i]) : (xlg[i - m] == xug[i - m]);
      if(eq && i < first) first = i;
    }
    if(first != 0x7fffffff) atomicMin(&ctl->eq_next, first);
  }
  __syncthreads();
  STAMP(1)

  int term = T_SUCCESS;
  int it = 0;
  int q = 0;

  if(ctl->info >= 0)
  {
    term = T_NON_POS_HESSIAN;
  }
  else
  {
    // optional: hand the factor back (the reference leaves it in G's lower triangle)
    if(A.Lout)
    {
      double * Lg = A.Lout + (size_t)b * n * n;
      for(int e = tid; e < n * n; e += WG)
      {
        int i = e % n, j = e / n;
        if(i >= j) Lg[e] = L[i + j * ld];
      }
    }
    // ---------------------------------------------------------------- K2 (wave 1): J = L^-T ; K3 (wave 0): x0, f0
    if(wave == 1)
    {
      const int c = lane;
      for(int i = n - 1; i >= 0; --i)
      {
        if(c < n)
        {
          double v = 0.0;
          if(c >= i)
          {
            double acc = (c == i) ? 1.0 : 0.0;
            for(int k = c; k > i; --k) acc = FMA(-L[k + i * ld], J[k + c * ld], acc);
            v = acc / L[i + i * ld];
          }
          J[i + c * ld] = v;
        }
      }
    }
    else if(wave == 0)
    {
      double xi = (lane < n) ? ag[lane] : 0.0;
      const double ai = xi;
      for(int k = 0; k < n; ++k) // x = L.solve(a)
      {
        double v = bcast_lane(xi, k) / L[k + k * ld];
        if(lane == k)
          xi = v;
        else if(lane > k && lane < n)
          xi = FMA(-L[lane + k * ld], v, xi);
      }
      for(int k = n - 1; k >= 0; --k) // L^T.solveInPlace(x)
      {
        double v = bcast_lane(xi, k) / L[k + k * ld];
        if(lane == k)
          xi = v;
        else if(lane < k)
          xi = FMA(-L[k + lane * ld], v, xi);
      }
      xi = -xi;
      if(lane < n) xs[lane] = xi;
      // f = 0.5 * a.dot(x): one in-order chain, operands fetched with readlane
      double acc = 0.0;
      for(int k = 0; k < n; ++k) acc = FMA(bcast_lane(ai, k), bcast_lane(xi, k), acc);
      if(lane == 0) ctl->f = 0.5 * acc;
    }
    __syncthreads();
    STAMP(2)

    // more equalities / fixed variables than unknowns is UB in the reference; reported (as the oracle does)
    int mode = (ctl->eq_next != 0x7fffffff) ? 0 : 1; // 0: initActiveSet, 1: main loop
    if(mode == 0)
    {
      int cnt = 0;
      for(int i = tid; i < mt; i += WG)
      {
        bool eq = (i < m) ? (blg[i] == bug[i]) : (xlg[i - m] == xug[i - m]);
        cnt += eq;
      }
      // block-wide sum through an LDS atomic on the (still unused) red_idx slot
      if(tid == 0) ctl->red_idx[0] = 0;
      __syncthreads();
      if(cnt) atomicAdd(&ctl->red_idx[0], cnt);
      __syncthreads();
      if(ctl->red_idx[0] > n)
      {
        term = T_OVERCONSTRAINED;
        mode = 2;
      }
      __syncthreads();
    }
    int eq_cursor = 0;
    int skip = 0;
    int p = -1, pstat = INACTIVE;
    const double big = A.big_bnd;

    while(mode != 2)
    {
      // ============================================================ selection
      if(mode == 0)
      {
        // next constraint with bl == bu (then bounds with xl == xu), in index order
        int first = 0x7fffffff;
        for(int i = eq_cursor + tid; i < mt; i += WG)
        {
          bool eq = (i < m) ? (blg[i] == bug[i]) : (xlg[i - m] == xug[i - m]);
          if(eq && i < first) first = i;
        }
        if(tid == 0) ctl->eq_next = 0x7fffffff;
        __syncthreads();
        if(first != 0x7fffffff) atomicMin(&ctl->eq_next, first);
        __syncthreads();
        p = uniform(ctl->eq_next);
        __syncthreads();
        if(p == 0x7fffffff)
        {
          mode = 1;
          continue;
        }
        pstat = (p < m) ? EQUALITY : FIXED;
        eq_cursor = p + 1;
        if(tid == 0) us[q] = 0.0;
      }
      else
      {
        if(it >= A.max_iter)
        {
          term = T_MAX_ITER;
          break;
        }
        if(!skip)
        {
          // ---- K4: s_i = min(C_i^T x - bl_i, bu_i - C_i^T x), first-index arg-min below 0
          double bval = 0.0;
          int bidx = 0x7fffffff, bstat = INACTIVE;
          for(int i = tid; i < m; i += WG)
          {
            if(st[i] != INACTIVE) continue;
            double cx = 0.0;
            if(C_LDS)
            {
              const double * ci = Cs + (size_t)i * ld;
              for(int k = 0; k < n; ++k) cx = FMA(ci[k], xs[k], cx);
            }
            else
            {
              const double * ci = Cg + (size_t)i * n;
              for(int k = 0; k < n; ++k) cx = FMA(ci[k], xs[k], cx);
            }
            double lo = C_LDS ? bls[i] : blg[i];
            double sl = cx - lo;
            if(sl < bval)
            {
              bval = sl;
              bidx = i;
              bstat = LOWER;
            }
            else
            {
              double hi = C_LDS ? bus[i] : bug[i];
              double su = hi - cx;
              if(su < bval)
              {
                bval = su;
                bidx = i;
                bstat = UPPER;
              }
            }
          }
          for(int i = tid; i < nb; i += WG)
          {
            if(st[m + i] != INACTIVE) continue;
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
          wave_argmin(bval, bidx, bstat);
          if(lane == 0)
          {
            ctl->red_val[wave] = bval;
            ctl->red_idx[wave] = bidx;
            ctl->red_stat[wave] = bstat;
          }
          __syncthreads();
          {
            double v = ctl->red_val[0];
            int ix = ctl->red_idx[0], sx = ctl->red_stat[0];
#pragma unroll
            for(int w = 1; w < 4; ++w)
            {
              double ov = ctl->red_val[w];
              int oi = ctl->red_idx[w];
              if(ov < v || (ov == v && oi < ix))
              {
                v = ov;
                ix = oi;
                sx = ctl->red_stat[w];
              }
            }
            p = uniform((ix == 0x7fffffff) ? -1 : ix);
            pstat = uniform(sx);
          }
          if(p < 0)
          {
            term = T_SUCCESS; // DualSolver.cpp:114
            break;
          }
          if(tid == 0) us[q] = 0.0; // DualSolver.cpp:121
        }
      }
      STAMP(3)

      // ============================================================ K5: d = J^T n+   (wave 0, lane = column of J)
      const bool general = (p < m);
      const int pb = p - m;
      const double * cp = C_LDS ? (Cs + (size_t)(general ? p : 0) * ld) : (Cg + (size_t)(general ? p : 0) * n);
      if(wave == 0 && lane < n)
      {
        double dj;
        if(general)
        {
          dj = 0.0;
          for(int k = 0; k < n; ++k) dj = FMA(J[k + lane * ld], cp[k], dj);
        }
        else
          dj = J[pb + lane * ld];
        if(pstat == UPPER || pstat == UPPER_BOUND) dj = -dj;
        ds[lane] = dj;
      }
      __syncthreads();
      STAMP(4)

      // ============================================================ K7 + ratio test (wave 0) || K6 + dots (wave 1)
      if(wave == 0)
      {
        double rk = (lane < q) ? ds[lane] : 0.0;
        for(int k = q - 1; k >= 0; --k)
        {
          double v = bcast_lane(rk, k) / R[k + k * ld];
          if(lane == k)
            rk = v;
          else if(lane < k)
            rk = FMA(-R[lane + k * ld], v, rk);
        }
        if(lane < q) rs[lane] = rk;
        // t1 = min u_k / r_k over r_k > 0 (not EQUALITY/FIXED; status looked up by POSITION, GoldfarbIdnaniSolver.cpp:162)
        double tk = big;
        int kk = 0x7fffffff, dummy = 0;
        if(lane < q)
        {
          int sk = st[lane];
          if(sk != EQUALITY && sk != FIXED && rk > 0.0)
          {
            double tt = us[lane] / rk;
            if(tt < big)
            {
              tk = tt;
              kk = lane;
            }
          }
        }
        wave_argmin(tk, kk, dummy);
        if(lane == 0)
        {
          ctl->t1 = tk;
          ctl->l = (kk == 0x7fffffff) ? 0 : kk;
        }
      }
      else if(wave == 1)
      {
        double zi = 0.0;
        if(lane < n)
        {
          for(int j = q; j < n; ++j) zi = FMA(J[lane + j * ld], ds[j], zi);
          zs[lane] = zi;
        }
        wave_sync();
        // three in-order chains in three lanes: z.z, c_p.z, c_p.x
        if(lane < 3)
        {
          double acc = 0.0;
          if(general)
          {
            const double * pa = (lane == 0) ? zs : cp;
            const double * pv = (lane == 2) ? xs : zs;
            for(int k = 0; k < n; ++k) acc = FMA(pa[k], pv[k], acc);
          }
          else
          {
            if(lane == 0)
              for(int k = 0; k < n; ++k) acc = FMA(zs[k], zs[k], acc);
            else
              acc = (lane == 1) ? zs[pb] : xs[pb];
          }
          if(lane == 0) ctl->zz = acc;
          if(lane == 1) ctl->cz = acc;
          if(lane == 2) ctl->cx = acc;
        }
      }
      __syncthreads();
      STAMP(5)

      // ============================================================ K8/K9: step length and primal/dual update
      const double cz = uniform_d(ctl->cz);
      const double zz = uniform_d(ctl->zz);
      const double cxp = uniform_d(ctl->cx);
      const double dotsc = (pstat == UPPER || pstat == UPPER_BOUND) ? -cz : cz; // ConstraintNormal::dot
      double t, t1 = big, t2 = big;
      int l = 0;
      if(mode == 0)
      {
        // addInitialConstraint (GoldfarbIdnaniSolver.cpp:295-338): full step, no ratio test
        t = 0.0;
        if(__builtin_sqrt(zz) > 1e-14)
        {
          double bb = uniform_d(general ? (C_LDS ? bls[p] : blg[p]) : xlg[pb]);
          t = (bb - cxp) / cz;
        }
      }
      else
      {
        t1 = uniform_d(ctl->t1);
        l = uniform(ctl->l);
        if(__builtin_sqrt(zz) > 1e-14)
        {
          double bb;
          if(pstat == LOWER)
            bb = C_LDS ? bls[p] : blg[p];
          else if(pstat == UPPER)
            bb = C_LDS ? bus[p] : bug[p];
          else if(pstat == LOWER_BOUND)
            bb = xlg[pb];
          else
            bb = xug[pb];
          t2 = (uniform_d(bb) - cxp) / cz;
        }
        t = (t2 < t1) ? t2 : t1;
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
      }
      const bool dual_only = (mode == 1) && (t2 >= big);
      const bool do_add = (mode == 0) || (!dual_only && t == t2);

      if(wave == 0)
      {
        if(!dual_only)
        {
          if(lane < n) xs[lane] = FMA(t, zs[lane], xs[lane]);
          if(lane == 0) ctl->f = FMA(t * dotsc, 0.5 * t + us[q], ctl->f); // uses u[q] BEFORE its update
        }
        wave_sync();
        if(lane < q) us[lane] = FMA(-t, rs[lane], us[lane]);
        if(lane == 0) us[q] = us[q] + t;
        wave_sync();
        STAMP(6)

        if(do_add)
        {
          // ---- K10: addConstraint_ (GoldfarbIdnaniSolver.cpp:221-237); lane = row of J
          double carry = (lane < n) ? J[lane + (n - 1) * ld] : 0.0;
          double dprev = ds[n - 1];
          for(int i = n - 2; i >= q; --i)
          {
            double c, s, rho;
            make_givens(ds[i], dprev, c, s, rho);
            dprev = rho;
            if(lane < n)
            {
              double xv = J[lane + i * ld];
              J[lane + (i + 1) * ld] = FMA(s, xv, c * carry);
              carry = FMA(c, xv, -(s * carry));
            }
          }
          if(lane < n) J[lane + q * ld] = carry;
          // R.col(q).head(q+1) = d.head(q+1)
          if(lane < q)
            R[lane + q * ld] = ds[lane];
          else if(lane == q)
            R[q + q * ld] = dprev;
          if(lane == 0)
          {
            act[q] = p;
            st[p] = (unsigned char)pstat;
          }
          STAMP(7)
          if(PROF && tid == 0) A.prof[(size_t)b * 16 + 10] += 1;
        }
        else
        {
          // ---- K11: removeConstraint (DualSolver.cpp:237-244, GoldfarbIdnaniSolver.cpp:239-256)
          double uv = (lane >= l && lane < q) ? us[lane + 1] : 0.0;
          int av = (lane >= l && lane < q - 1) ? act[lane + 1] : 0;
          int gone = act[l];
          wave_sync();
          if(lane >= l && lane < q) us[lane] = uv;
          if(lane >= l && lane < q - 1) act[lane] = av;
          if(lane == 0) st[gone] = INACTIVE;
          const int qn = q - 1;
          const bool mine = (lane >= l && lane < qn); // lane = NEW column index; old column = lane + 1
          for(int rr = 0; rr < l; ++rr)
          {
            double v = mine ? R[rr + (lane + 1) * ld] : 0.0;
            wave_sync();
            if(mine) R[rr + lane * ld] = v;
          }
          double carry = mine ? R[l + (lane + 1) * ld] : 0.0;
          wave_sync();
          for(int i = l; i < qn; ++i)
          {
            double y = (lane >= i && lane < qn) ? R[(i + 1) + (lane + 1) * ld] : 0.0;
            double c, s, rho;
            make_givens(bcast_lane(carry, i), bcast_lane(y, i), c, s, rho);
            if(lane == 0)
            {
              ds[i] = c; // d and z are dead here: reuse them to hand (c, s) to the J sweep
              zs[i] = s;
            }
            if(lane == i)
              R[i + i * ld] = rho;
            else if(lane > i && lane < qn)
            {
              R[i + lane * ld] = FMA(c, carry, -(s * y));
              carry = FMA(s, carry, c * y);
            }
            wave_sync();
          }
          if(lane < n && qn > l)
          {
            double jc = J[lane + l * ld];
            for(int i = l; i < qn; ++i)
            {
              double c = ds[i], s = zs[i];
              double y = J[lane + (i + 1) * ld];
              J[lane + i * ld] = FMA(c, jc, -(s * y));
              jc = FMA(s, jc, c * y);
            }
            J[lane + qn * ld] = jc;
          }
          STAMP(8)
          if(PROF && tid == 0) A.prof[(size_t)b * 16 + 11] += 1;
        }
      }
      if(do_add)
      {
        q = uniform(q + 1);
        skip = 0;
      }
      else
      {
        q = uniform(q - 1);
        skip = 1;
      }
      if(mode == 1) ++it;
      __syncthreads();
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- results (K13: sign-expanded multipliers)
  if(tid < n) A.x[(size_t)b * n + tid] = xs[tid];
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
  if(PROF && tid == 0) A.prof[(size_t)b * 16 + 12] += wall_clock64() - wall0_;
}

} // namespace

long batch_lds_bytes(int n, int m, int nb, bool c_lds)
{
  const long ld = n | 1;
  long dbl = 2L * n * ld + 3L * n + 2L * (n + 2);
  if(c_lds) dbl += (long)m * ld + 2L * m;
  long bytes = dbl * 8 + (long)sizeof(Ctl) + 4L * (n + 2) + (long)(m + nb);
  return (bytes + 15) & ~15L;
}

template<bool C_LDS, bool PROF>
static hipError_t launch_one(const BatchArgs & a, long lds, hipStream_t stream)
{
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gi_batch_kernel<C_LDS, PROF>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if(e != hipSuccess) return e;
  hipLaunchKernelGGL((gi_batch_kernel<C_LDS, PROF>), dim3(a.B), dim3(WG), (size_t)lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_gi_batch(const BatchArgs & a, bool c_lds, hipStream_t stream)
{
  const long lds = batch_lds_bytes(a.n, a.m, a.nb, c_lds);
  if(a.prof) return c_lds ? launch_one<true, true>(a, lds, stream) : launch_one<false, true>(a, lds, stream);
  return c_lds ? launch_one<true, false>(a, lds, stream) : launch_one<false, false>(a, lds, stream);
}

} // namespace jrlqp
