// blockgi.hip -- the STRUCTURED Goldfarb-Idnani inner loop on gfx950: one workgroup per QP, the whole state machine
// on the device, one launch per batch.
//
//   reference (jrl-umi3218/jrl-qp):
//     experimental::BlockGISolver      src/experimental/BlockGISolver.cpp:62-288, :446-483   hooks + init
//     StructuredG / StructuredC        src/structured/StructuredG.cpp:45-113, StructuredC.cpp:59-77
//     StructuredJ                      src/structured/StructuredJ.cpp:32-57     d = Q^T L^-1 n+,  z = L^-T Q [0; d2]
//     StructuredQR                     src/structured/StructuredQR.cpp:64-102   R solve, Householder add, Givens remove
//     OrthonormalSequence              src/internal/OrthonormalSequence.cpp:50-126, :189-207
//     generic loop                     src/DualSolver.cpp:91-168, :231-244, :38-69
//   oracle: oracle/blockgi_oracle.c (bit-identical results; rounding rules C1-C4 stated there and in DESIGN.md §4)
//
// Data placement: the vectors of unknowns (d, z, x) and the dual vectors (u, r) live in LDS; everything that grows
// with the iteration count -- R (packed upper triangle), the Householder vectors / Givens coefficients of the
// orthonormal sequence, the active list -- lives in a per-problem HBM workspace that stays L2-resident; the factor L
// (the caller's blocks, factorised in place by the kernels of structured.hip just before this launch) and C are read
// from where the caller put them.
//
// Parallel structure: J is never formed.  One iteration is a handful of O(n) vector passes:
//   * L^-1 n+ touches only the blocks under the constraint's non-zero window (reference zero-window logic);
//   * Q^T / Q applications are sequences of reflectors  v -= tau (E.v) E : the working vector stays in REGISTERS
//     (element e on thread (e mod 1024) / VPT), one C4 tree reduction and one barrier per reflector;
//   * L^-T is block parallel for the arrow shapes (one wave per off-head block), serial over blocks for the
//     tridiagonal shape;
//   * R^-1 d1 is a blocked back substitution (64-column diagonal blocks on one wave, pivots by v_readlane, row
//     updates on all waves);
//   * dropping a constraint pipelines the Givens rotations over the columns of R: one lane per column, one carried
//     value per lane, (c, s) broadcast with v_readlane.
#include "blockgi.h"

#include "device_common.h"

namespace jrlqp
{
namespace
{
#include "blockgi_common.h"

// ---- OrthonormalSequence::applyToTheLeft / applyTransposeToTheLeft (OrthonormalSequence.cpp:189-207) -----------------
// v (LDS, n entries) <- Q v (TR = false: last element first) or Q^T v (TR = true: first element first).
// Entry: v complete and visible to every thread (a barrier has been passed).  Exit: same.
template<int NT, int NCH, bool TR>
__device__ __forceinline__ void q_apply(Ctx & K, double * v, int nseq, long long * qprof)
{
  constexpr int VPT = 1024 / NT;
  const BgiArgs & A = *K.A;
  const int n = A.n, t0 = K.tid * VPT;
  if(nseq == 0) return;
  double vr[NCH][VPT];
#pragma unroll
  for(int c = 0; c < NCH; ++c)
#pragma unroll
    for(int i = 0; i < VPT; ++i)
    {
      const int e = 1024 * c + t0 + i;
      vr[c][i] = (e < n) ? v[e] : 0.0;
    }
  // Reflector rows stream from HBM / the Infinity Cache (one row per iteration ever made: far more than L2 holds), and
  // one reflector costs far less than one memory latency: the rows and descriptors of the next DEPTH elements are kept
  // in flight in a register ring (<= 36 doubles per thread for the large shapes; 16 for n <= 1024, where registers
  // decide how many problems share a CU).  Row addresses depend on the element index only -- the fetch pointers simply
  // step by one row / one record per element -- and whole VPT-packs inside the row are loaded unconditionally (a row
  // is zero outside its reflector's support).
  constexpr int DEPTH = (NCH == 1) ? 4 : ((36 / (NCH * VPT)) > 8 ? 8 : (36 / (NCH * VPT)) < 2 ? 2 : (36 / (NCH * VPT)));
  double ring[DEPTH][NCH][VPT];
  int rkind[DEPTH], rst[DEPTH], rsz[DEPTH];
  double rtau[DEPTH];
  // element fetched next (in application order kf = 0, 1, ...; clamped at the last one) and its row / record
  int kf = 0;
  const ptrdiff_t rstep = TR ? (ptrdiff_t)A.row_len : -(ptrdiff_t)A.row_len;
  const double * frow = K.pool + (size_t)(TR ? 0 : nseq - 1) * A.row_len + t0;
  const SeqRec * frec = K.desc + (TR ? 0 : nseq - 1);
  bool act[NCH];
#pragma unroll
  for(int c = 0; c < NCH; ++c) act[c] = (1024 * c + t0 < A.n_pad);
  auto fetch = [&](double(&en)[NCH][VPT], int & nkind, int & nst, int & nsz, double & ntau) {
    // (records are written by thread 0 behind a barrier and only ever read with vector loads)
    const int4 rec = *reinterpret_cast<const int4 *>(frec);
    nkind = rec.x;
    nst = rec.y;
    nsz = rec.z;
    ntau = frec->tau;
#pragma unroll
    for(int c = 0; c < NCH; ++c)
    {
      if(act[c])
      {
        if constexpr(VPT == 4)
        {
          const double2 lo = *reinterpret_cast<const double2 *>(frow + 1024 * c), hi = *reinterpret_cast<const double2 *>(frow + 1024 * c + 2);
          en[c][0] = lo.x;
          en[c][1] = lo.y;
          en[c][2] = hi.x;
          en[c][3] = hi.y;
        }
        else if constexpr(VPT == 2)
        {
          const double2 lo = *reinterpret_cast<const double2 *>(frow + 1024 * c);
          en[c][0] = lo.x;
          en[c][1] = lo.y;
        }
        else
        {
#pragma unroll
          for(int i = 0; i < VPT; ++i) en[c][i] = (1024 * c + t0 + i < A.n_pad) ? frow[1024 * c + i] : 0.0;
        }
      }
    }
    if(kf + 1 < nseq) // (uniform) the last element is simply fetched again by the slots that run past the end
    {
      ++kf;
      frow += rstep;
      frec += TR ? 1 : -1;
    }
  };
#pragma unroll
  for(int j = 0; j < DEPTH; ++j)
#pragma unroll
    for(int c = 0; c < NCH; ++c)
#pragma unroll
      for(int i = 0; i < VPT; ++i) ring[j][c][i] = 0.0;
  // rolling prefetch: slot j of the ring holds element kk0 + j; as soon as an element has been copied out of its slot
  // the slot is refilled with the element DEPTH positions further (clamped index: no branch around a load), so DEPTH
  // rows are in flight at any time.  The loop is unrolled by DEPTH: every slot index is a compile-time constant.
#pragma unroll
  for(int j = 0; j < DEPTH; ++j) fetch(ring[j], rkind[j], rst[j], rsz[j], rtau[j]);
  for(int kk0 = 0; kk0 < nseq; kk0 += DEPTH)
  {
#pragma unroll
  for(int j = 0; j < DEPTH; ++j)
  {
    const int kk = kk0 + j;
    if(kk < nseq)
    {
    const int k = TR ? kk : nseq - 1 - kk;
    const int kind = rkind[j], st = rst[j], sz = rsz[j];
    const double tau = rtau[j];
    const double * row = K.pool + (size_t)k * A.row_len;
    double er[NCH][VPT];
#pragma unroll
    for(int c = 0; c < NCH; ++c)
#pragma unroll
      for(int i = 0; i < VPT; ++i) er[c][i] = ring[j][c][i];
    fetch(ring[j], rkind[j], rst[j], rsz[j], rtau[j]);
    __builtin_amdgcn_sched_barrier(0); // the refill is issued HERE, ahead of this element's arithmetic
    const long long tq0 = qprof ? wall_clock64() : 0;
    if(kind == 0)
    {
      // one reflector: v -= tau (E.v) E on rows [st, n)  (OrthonormalSequence.cpp:59-63, :98-102).  The stored row is
      // ZERO outside [st, n) (written so when the reflector is made) and so is the working vector beyond n: the
      // products of those positions are exact zeros, a chain that receives fma(0, v, p) keeps p (p is never -0), and
      // fma(-s, 0, v) = v except that a NEGATIVE zero entry of v may come back as +0 -- no mask instruction anywhere.
      double p[VPT];
#pragma unroll
      for(int i = 0; i < VPT; ++i) p[i] = 0.0;
#pragma unroll
      for(int c = 0; c < NCH; ++c)
#pragma unroll
        for(int i = 0; i < VPT; ++i) p[i] = FMA(er[c][i], vr[c][i], p[i]);
#pragma unroll
      for(int s2 = 1; s2 < VPT; s2 <<= 1)
#pragma unroll
        for(int i = 0; i < VPT; i += 2 * s2) p[i] = p[i] + p[i + s2];
      const double dd = block_sum<NT>(K, p[0]);
      const double sc = tau * dd;
#pragma unroll
      for(int c = 0; c < NCH; ++c)
#pragma unroll
        for(int i = 0; i < VPT; ++i) vr[c][i] = FMA(-sc, er[c][i], vr[c][i]);
    }
    else if(sz > 0)
    {
      // a Givens sequence on rows st .. st + sz (OrthonormalSequence.cpp:76-78, :115-117): serial, one wave, v in LDS
#pragma unroll
      for(int c = 0; c < NCH; ++c)
#pragma unroll
        for(int i = 0; i < VPT; ++i)
        {
          const int e = 1024 * c + t0 + i;
          if(e < n) v[e] = vr[c][i];
        }
      __syncthreads();
      if(K.wave == 0)
      {
        const double *cp = row, *sp = row + A.qcap;
        double * w = v + st;
        const int lane = K.lane;
        // The rotations form one serial chain through the carried element.  Ascending order: ONE fma per step (the
        // product c*y does not depend on the carry).  Chunks of 64 rotations: lane j stages the operands of rotation j
        // in LDS, then EVERY lane runs the same scalar chain over uniform LDS reads, eight steps' operands requested
        // ahead of the eight dependent fmas; lane j keeps the carry that entered step j and forms the rotation's second
        // output afterwards.
        double *S = K.tri, *B = S + 64;
        if(TR)
        {
          // ascending i: (w_i, w_i+1) <- (c w_i - s w_i+1, s w_i + c w_i+1); carried: the new w_i+1
          double xc = w[0];
          double cn = (lane < sz) ? cp[lane] : 0.0, snn = (lane < sz) ? sp[lane] : 0.0;
          for(int i0 = 0; i0 < sz; i0 += 64)
          {
            const int cnt = min(64, sz - i0);
            const bool on = lane < cnt;
            const double c = cn, sn = snn;
            if(i0 + 64 < sz) // the next chunk's coefficients are requested before this chunk's chain runs
            {
              const int ix = min(i0 + 64 + lane, sz - 1);
              cn = cp[ix];
              snn = sp[ix];
            }
            const double y = on ? w[i0 + 1 + lane] : 0.0;
            S[lane] = sn;
            B[lane] = c * y;
            wave_sync();
            double tin = 0.0;
            for(int j0 = 0; j0 < cnt; j0 += 8)
            {
              double sv[8], bv[8];
#pragma unroll
              for(int u = 0; u < 8; ++u)
              {
                sv[u] = S[min(j0 + u, 63)];
                bv[u] = B[min(j0 + u, 63)];
              }
#pragma unroll
              for(int u = 0; u < 8; ++u)
              {
                const double nx = FMA(sv[u], xc, bv[u]);
                tin = (lane == j0 + u) ? xc : tin;
                xc = (j0 + u < cnt) ? nx : xc;
              }
            }
            if(on) w[i0 + lane] = FMA(-sn, y, c * tin);
            wave_sync();
          }
          if(lane == 0) w[sz] = xc;
        }
        else
        {
          // descending i: (w_i, w_i+1) <- (c w_i + s w_i+1, -s w_i + c w_i+1); carried: the new w_i = fma(s, w_i+1, c w_i)
          // (CANON C6: the product WITHOUT the carried operand is the rounded one, so a step is one fma on the carry -- c w_i is
          // the lane's own constant).  The systolic form is the faster one -- lane j owns rotation j, the carry
          // travels lane to lane with one DPP wave shift per step, every lane recomputes its step each time and lane
          // j's result is final from step j on.  The chunk is aligned to the TOP lanes (rotation i1-1 on lane 63); the
          // entry lane's result is known up front and it is given coefficients that reproduce it whatever arrives.
          double yc = w[sz];
          double cn, snn;
          {
            const int cnt = min(64, sz), j = lane - (64 - cnt);
            cn = (j >= 0) ? cp[sz - cnt + j] : 0.0;
            snn = (j >= 0) ? sp[sz - cnt + j] : 0.0;
          }
          for(int i1 = sz; i1 > 0; i1 -= 64)
          {
            const int cnt = min(64, i1), i0 = i1 - cnt;
            const int j = lane - (64 - cnt); // rotation i0 + j
            const bool on = j >= 0;
            const double c = cn, sn = snn;
            if(i0 > 0)
            {
              const int cnt2 = min(64, i0), j2 = lane - (64 - cnt2), ix = max(i0 - cnt2 + j2, 0);
              cn = cp[ix];
              snn = sp[ix];
              if(j2 < 0) cn = snn = 0.0;
            }
            const double xv = on ? w[i0 + j] : 0.0;
            double cl = c, xl = xv, sl = sn;
            if(lane == 63)
            {
              xl = FMA(sn, yc, c * xv);
              cl = 1.0;
              sl = 0.0;
            }
            const double pl = cl * xl;
            double out = 0.0, t = 0.0;
            for(int step = 0; step < cnt; ++step)
            {
              t = dpp_shift0_d<0x130>(out); // wave_shl:1 -- lane j takes lane j+1's carry
              out = FMA(sl, t, pl);
            }
            t = (lane == 63) ? yc : t;
            if(on) w[i0 + j + 1] = FMA(-sn, xv, c * t);
            yc = bcast_lane(out, 64 - cnt);
          }
          if(lane == 0) w[0] = yc;
        }
      }
      __syncthreads();
#pragma unroll
      for(int c = 0; c < NCH; ++c)
#pragma unroll
        for(int i = 0; i < VPT; ++i)
        {
          const int e = 1024 * c + t0 + i;
          vr[c][i] = (e < n) ? v[e] : 0.0;
        }
    }
    if(qprof) qprof[kind] += wall_clock64() - tq0;
    } // kk < nseq
  }
  }
#pragma unroll
  for(int c = 0; c < NCH; ++c)
#pragma unroll
    for(int i = 0; i < VPT; ++i)
    {
      const int e = 1024 * c + t0 + i;
      if(e < n) v[e] = vr[c][i];
    }
  __syncthreads();
}

// ---- StructuredQR::RSolve (StructuredQR.cpp:64-68): r <- R^-1 d1, R packed in HBM, r in LDS --------------------------
// Per element the order is the column-oriented back substitution's: contributions of columns k in descending order.
template<int NT>
__device__ __forceinline__ void r_solve(Ctx & K, int q)
{
  q = __builtin_amdgcn_readfirstlane(q); // (scalar loop counters: see bcast_lane)
  const double * R = K.Rp;
  double *r = K.r, *tri = K.tri;
  for(int k = K.tid; k < q; k += NT) r[k] = K.d[k];
  __syncthreads();
  for(int c1 = q; c1 > 0;)
  {
    const int c0 = (c1 - 1) & ~63, nc = c1 - c0;
    // stage the diagonal block (columns c0 .. c1-1, rows c0 .. column) in LDS, packed by columns: column jj of the block
    // holds jj + 1 entries starting at jj (jj + 1) / 2; thread = (row ii, column jj0 + multiples of NT / 64)
    for(int jj = K.tid >> 6; jj < nc; jj += NT / 64)
    {
      const int ii = K.tid & 63;
      if(ii <= jj) tri[jj * (jj + 1) / 2 + ii] = RP(R, c0 + ii, c0 + jj);
    }
    __syncthreads();
    if(K.wave == 0)
    {
      // back substitution inside the block: pivots by v_readlane, reciprocals of the diagonal hoisted out of the chain
      // (div_pre; redone with plain divisions if an operand leaves the range in which that is the compiler's quotient)
      const int lane = K.lane;
      const double x0 = (lane < nc) ? r[c0 + lane] : 0.0;
      const double dg = (lane < nc) ? tri[lane * (lane + 1) / 2 + lane] : 1.0;
      const double rinv = recip_refined(dg);
      double xi = x0;
      // (steps as in wave_bwd, device_common.h: the quotient formed by the owning lane, the numerator left in its lane, all
      // quotients formed after the last step; scalar loop counter)
      for(int k = nc - 1; k >= 0; --k)
      {
        const double * col = tri + k * (k + 1) / 2;
        const double v = bcast_lane_s(div_pre(xi, dg, rinv), k);
        const double upd = FMA(-col[min(lane, k)], v, xi);
        xi = (lane < k) ? upd : xi;
      }
      const double pre = xi;
      xi = div_pre(pre, dg, rinv);
      if(!__all(lane >= nc || (div_safe_den(dg) && div_safe_num(pre))))
      {
        xi = x0;
        for(int k = nc - 1; k >= 0; --k)
        {
          const double * col = tri + k * (k + 1) / 2;
          const double v = bcast_lane(xi, k) / col[k];
          if(lane == k)
            xi = v;
          else if(lane < k)
            xi = FMA(-col[lane], v, xi);
        }
      }
      if(lane < nc) r[c0 + lane] = xi;
    }
    __syncthreads();
    // rows above the block: r_i -= sum_k R(i,k) r_k, k descending
    for(int i = K.tid; i < c0; i += NT)
    {
      r[i] = chain_neg8(c1 - 1, -1, nc, r[i], [&](int k) { return RP(R, i, k); }, [&](int k) { return r[k]; });
    }
    __syncthreads();
    c1 = c0;
  }
}

// =========================================================================================================
// the kernel
// =========================================================================================================
template<int NT, int NCH, bool BIG>
__global__ __launch_bounds__(NT, (NT == 256 && NCH == 1) ? 3 : (NT == 256) ? 2 : 0) void blockgi_kernel(const BgiArgs A)
{
  extern __shared__ double sm[];
  const int pb = blockIdx.x, tid = threadIdx.x;
  const int n = A.n, m = A.m, nb = A.nb;
  Ctx K;
  K.A = &A;
  K.tid = tid;
  K.lane = tid & 63;
  K.wave = tid >> 6;
  K.diag = A.diag + (size_t)pb * A.nblk;
  K.side = A.side + (size_t)pb * (A.nblk > 1 ? A.nblk - 1 : 0);
  K.cblk = A.cblk + (size_t)pb * A.ncb;
  K.d = sm;
  K.z = K.d + A.n_pad;
  if(A.xu_global)
  {
    K.x = A.xu_ws + (size_t)pb * (A.n_pad + A.qcap);
    K.u = K.x + A.n_pad;
    K.r = K.z + A.n_pad;
  }
  else
  {
    K.x = K.z + A.n_pad;
    K.u = K.x + A.n_pad;
    K.r = K.u + A.qcap;
  }
  K.tri = K.r + A.qcap;
  K.red = K.tri + kTri;
  K.ired = reinterpret_cast<int *>(K.red + 32);
  K.par = 0;
  {
    BlkMeta * bm = reinterpret_cast<BlkMeta *>(K.red + 32 + 32 + 8);
    fill_block_table<NT>(A, pb, tid, K.diag, K.side, bm);
    K.bm = bm;
  }
  K.pool = A.pool + (size_t)pb * A.max_iter * A.row_len;
  K.Rp = A.Rp + (size_t)pb * A.rlen;
  K.desc = A.desc + (size_t)pb * A.max_iter;
  K.active = A.active + (size_t)pb * A.qcap;
  K.status = A.status + (size_t)pb * (m + nb);
  double *d = K.d, *z = K.z, *x = K.x, *u = K.u, *r = K.r;
  double * sh = K.red + 32 + 32; // a few doubles of broadcast scratch (after ired: 64 ints = 32 doubles)
  const double * a = A.a + (size_t)pb * n;
  const double *bl = A.bl + (size_t)pb * m, *bu = A.bu + (size_t)pb * m;
  const double *xl = nb ? A.xl + (size_t)pb * nb : nullptr, *xu = nb ? A.xu + (size_t)pb * nb : nullptr;
  jrlqp_trace_rec * trace = A.trace ? A.trace + (size_t)pb * A.trace_cap : nullptr;

  int q = 0, nseq = 0, it = 0, rc = JRLQP_UNKNOWN;
  double f = 0.0;
  // diagnostic phase timer (thread 0, 100 MHz wall clock): 0 init, 1 scan, 2 L^-1 n+, 3 Q^T d, 4 Q z, 5 L^-T z, 6 R solve,
  // 7 step length, 8 update, 9 add, 10 drop
  long long pacc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // 11, 12: reflectors / Givens sequences inside Q applications
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

  for(int i = tid; i < m + nb; i += NT) K.status[i] = JRLQP_INACTIVE;
  for(int i = tid; i < n; i += NT) x[i] = 0.0;
  for(int i = tid; i < A.qcap; i += NT) u[i] = r[i] = 0.0;
  __syncthreads();

  // ---- init_ (BlockGISolver.cpp:62-107) ----
  // processInitialActiveSet: data equalities / fixed variables are not functional in the reference (assert, :475)
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
  f = 0.5 * dot_tree<NT>(K, 0, n, [&](int e) { return a[e]; }, [&](int e) { return x[e]; });
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
          if(K.status[j] != JRLQP_INACTIVE) continue;
          const int b = A.to_block[j], len = A.cvar[b];
          const double * c = K.cblk[b] + (size_t)(j - A.cum_cstr[b]) * A.cld[b];
          const double * w = x + A.cum_var[b];
          // C^T x (StructuredC.cpp:66-77); -(-a) is exact, so the shared eight-ahead chain helper is used on -c
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
          if(K.status[m + i] != JRLQP_INACTIVE) continue;
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

      // ---- computeStep_ (BlockGISolver.cpp:164-172) ----
      // d = Q^T L^-1 n+  (StructuredJ.cpp:43-57)
      int wst, wen;
      for(int i = tid; i < n; i += NT) d[i] = 0.0;
      __syncthreads();
      if(p < m)
      {
        const int b = A.to_block[p], len = A.cvar[b];
        const double * c = K.cblk[b] + (size_t)(p - A.cum_cstr[b]) * A.cld[b];
        wst = A.cum_var[b];
        wen = wst + len;
        for(int i = tid; i < len; i += NT) d[wst + i] = c[i];
      }
      else
      {
        wst = p - m;
        wen = wst + 1;
        if(tid == 0) d[wst] = (pst == JRLQP_UPPER_BOUND) ? -1.0 : 1.0;
      }
      __syncthreads();
      sg_lsolve<NT, BIG>(K, d, wst, wen);
      if(pst == JRLQP_UPPER)
      {
        for(int i = tid; i < n; i += NT) d[i] = -d[i];
        __syncthreads();
      }
      stamp(2);
      q_apply<NT, NCH, true>(K, d, nseq, (A.prof && tid == 0) ? pacc + 11 : nullptr);
      stamp(3);
      // z = L^-T Q [0; d2]  (StructuredJ.cpp:32-41)
      for(int i = tid; i < n; i += NT) z[i] = (i < q) ? 0.0 : d[i];
      __syncthreads();
      q_apply<NT, NCH, false>(K, z, nseq, (A.prof && tid == 0) ? pacc + 11 : nullptr);
      stamp(4);
      sg_ltsolve<NT, BIG>(K, z);
      stamp(5);
      r_solve<NT>(K, q);
      stamp(6);

      // ---- computeStepLength_ (BlockGISolver.cpp:174-242) ----
      double t1 = A.big_bnd, t2 = A.big_bnd;
      int l = 0;
      {
        // (the position-indexed EQUALITY/FIXED exemption of :186 never fires: equalities are rejected above)
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
      const double zz = dot_tree<NT>(K, 0, n, [&](int e) { return z[e]; }, [&](int e) { return z[e]; });
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
        // ---- addConstraint: A_.activate + StructuredQR::add (StructuredQR.cpp:70-84), makeHouseholder of d.tail(n - q)
        const double tail = dot_tree<NT>(K, q + 1, n, [&](int e) { return d[e]; }, [&](int e) { return d[e]; });
        const double c0 = d[q];
        double tau, beta;
        double * row = K.pool + (size_t)nseq * A.row_len;
        if(tail <= 2.2250738585072014e-308)
        {
          tau = 0.0;
          beta = c0;
          for(int e = tid; e < A.n_pad; e += NT)
            if(e != q) row[e] = 0.0;
        }
        else
        {
          beta = __builtin_sqrt(FMA(c0, c0, tail));
          if(c0 >= 0.0) beta = -beta;
          const double den = c0 - beta;
          for(int e = tid; e < A.n_pad; e += NT)
            if(e != q) row[e] = (e > q && e < n) ? d[e] / den : 0.0; // zero outside (q, n): see q_apply
          tau = (beta - c0) / beta;
        }
        for(int k = tid; k < q; k += NT) RP(K.Rp, k, q) = d[k];
        if(tid == 0)
        {
          row[q] = 1.0;
          RP(K.Rp, q, q) = beta;
          K.desc[nseq] = SeqRec{0, q, 1, 0, tau, 0.0};
          K.status[p] = pst;
          K.active[q] = p;
        }
        ++q;
        ++nseq;
        skip_step1 = false;
        __syncthreads();
        stamp(9);
      }
      else
      {
        // ---- removeConstraint (DualSolver.cpp:237-244) + StructuredQR::remove (StructuredQR.cpp:86-102) ----
        // u.segment(l, q - l) = u.tail(q - l) (u has q + 1 entries); ActiveSet::deactivate(l) erases position l
        const int removed = K.active[l];
        for(int base = l; base < q; base += NT)
        {
          const int k = base + tid;
          const double tu = (k < q) ? u[k + 1] : 0.0;
          const int ta = (k + 1 < q) ? K.active[k + 1] : 0;
          __syncthreads();
          if(k < q) u[k] = tu;
          if(k + 1 < q) K.active[k] = ta;
          __syncthreads();
        }
        if(tid == 0) K.status[removed] = JRLQP_INACTIVE;
        const int q1 = q - 1, sz = q1 - l;
        double * R = K.Rp;
        double * row = K.pool + (size_t)nseq * A.row_len;
        double *cp = row, *sp = row + A.qcap;
        // rows above l: new column k-1 <- old column k; thread j owns row j (no cross-thread hazard)
        for(int j = tid; j < l; j += NT)
          for(int k = l + 1; k < q; ++k) RP(R, j, k - 1) = RP(R, j, k);
        // rows >= l: one lane per old column k = l+1 .. q-1, a carried value per lane, chunks of 64 columns
        if(K.wave == 0)
        {
          const int lane = K.lane;
          for(int cb = 0; cb < sz; cb += 64)
          {
            const int k = l + 1 + cb + lane;
            const bool valid = k < q;
            double xc = valid ? RP(R, l, k) : 0.0;
            const double dg = valid ? RP(R, k, k) : 0.0;
            // rotations generated by earlier chunks
            for(int i0 = l; i0 < l + cb; i0 += 64)
            {
              const double cl = cp[i0 - l + lane], sl = sp[i0 - l + lane]; // (i0 - l + lane < cb <= sz)
              for(int j = 0; j < 64; ++j)
              {
                const int i = i0 + j;
                const double c = bcast_lane(cl, j), s = bcast_lane(sl, j);
                if(valid)
                {
                  const double y = RP(R, i + 1, k);
                  const double xn = FMA(-s, y, c * xc);
                  xc = FMA(s, xc, c * y);
                  RP(R, i, k - 1) = xn;
                }
              }
            }
            // this chunk's own rotations: lane jj generates G_i, i = l + cb + jj
            const int cnt = min(64, sz - cb);
            double cmine = 0.0, smine = 0.0;
            for(int jj = 0; jj < cnt; ++jj)
            {
              const int i = l + cb + jj;
              double c, s, rr;
              make_givens(bcast_lane(xc, jj), bcast_lane(dg, jj), c, s, rr);
              if(lane == jj)
              {
                RP(R, i, i) = rr;
                cmine = c;
                smine = s;
              }
              else if(valid && lane > jj)
              {
                const double y = RP(R, i + 1, k);
                const double xn = FMA(-s, y, c * xc);
                xc = FMA(s, xc, c * y);
                RP(R, i, k - 1) = xn;
              }
            }
            if(lane < cnt)
            {
              cp[cb + lane] = cmine;
              sp[cb + lane] = smine;
            }
            __threadfence_block();
            wave_sync();
          }
        }
        if(tid == 0)
        {
          K.desc[nseq] = SeqRec{1, sz ? l : 0, sz, 0, 0.0, 0.0};
        }
        q = q1;
        ++nseq;
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

template<int NT, int NCH, bool BIG>
hipError_t launch_one_b(const BgiArgs & a, int nprob, size_t lds, hipStream_t st)
{
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&blockgi_kernel<NT, NCH, BIG>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if(e != hipSuccess) return e;
  hipLaunchKernelGGL((blockgi_kernel<NT, NCH, BIG>), dim3(nprob), dim3(NT), lds, st, a);
  return hipGetLastError();
}
template<int NT, int NCH>
hipError_t launch_one(const BgiArgs & a, int nprob, size_t lds, hipStream_t st)
{
  return (a.maxb > 64) ? launch_one_b<NT, NCH, true>(a, nprob, lds, st) : launch_one_b<NT, NCH, false>(a, nprob, lds, st);
}
} // namespace

long blockgi_lds_bytes(int n, int m, int nb, int nblk, int xu_global)
{
  const int n_pad = (n + 3) & ~3;
  int qcap = (n < m + nb ? n : m + nb) + 1;
  qcap = (qcap + 3) & ~3;
  const long vecs = xu_global ? (2L * n_pad + qcap) : (3L * n_pad + 2L * qcap);
  return (long)sizeof(double) * (vecs + kTri + 32 + 32 + 8) + (long)sizeof(BlkMeta) * nblk;
}

hipError_t launch_blockgi(const BgiArgs & a, int nprob, hipStream_t st)
{
  const size_t lds = (size_t)blockgi_lds_bytes(a.n, a.m, a.nb, a.nblk, a.xu_global);
  // four waves per problem up to 1024 unknowns; eight beyond (the block-parallel phases -- L^-T over ~100 blocks,
  // the constraint scan -- and the per-reflector work scale with the wave count; rule C4 does not depend on it).
  // Throughput layout (xu_global, chosen by the caller for large batches of large problems): four waves and half the
  // LDS, so that two problems share a CU and twice as many reflector rows are in flight per CU.
  if(a.n <= 1024) return launch_one<256, 1>(a, nprob, lds, st);
  if(a.xu_global)
    return (a.n <= 2048) ? launch_one<256, 2>(a, nprob, lds, st) : (a.n <= 3072) ? launch_one<256, 3>(a, nprob, lds, st) : launch_one<256, 4>(a, nprob, lds, st);
  if(a.n <= 2048) return launch_one<512, 2>(a, nprob, lds, st);
  if(a.n <= 3072) return launch_one<512, 3>(a, nprob, lds, st);
  return launch_one<512, 4>(a, nprob, lds, st); // (round 4: n <= 4096, a fourth chunk of 1024 chains per thread)
}
} // namespace jrlqp
