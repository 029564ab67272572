// blockgi_common.h -- device code shared by the two structured GI kernels (blockgi.hip: the reference's orthonormal-sequence form;
// blockgi_c9.hip: contract C9): the per-problem context, C4 reductions, StructuredG::solveL / solveInPlaceLTranspose on a vector held
// in LDS (block arrow up / down, block tridiagonal; blocks of up to 64 and of up to 128 rows), the (value, index) arg-min and n+ . v.
// Included inside namespace jrlqp { namespace { ... } } of each translation unit.
#pragma once
#define RP(R, i, j) (R)[(size_t)(i) + (size_t)(j) * ((size_t)(j) + 1) / 2]

constexpr int kTri = 64 * 65 / 2; // doubles of LDS scratch for one packed 64 x 64 upper triangle

// per-block description of L, staged in LDS once per solve: every block solve starts from these (a pointer fetched
// from HBM first would put one more memory latency in front of each of the ~100 block solves of an L^-T pass)
struct BlkMeta
{
  double * diag; // diagonal block b
  double * side; // side / sub-diagonal block b (b < nblk - 1)
  int bs, boff, ldd, lds;
  const double * diagT; // transposed copy of diag (ld bs): element (i, j) holds L(j, i)
  const double * sideT; // transposed copy of side (ld sct = columns of side): element (c, r) holds side(r, c)
  int sct, pad;
};

// fills the block table of one problem (LDS) -- shared by both kernels
template<int NT>
__device__ __forceinline__ void fill_block_table(const BgiArgs & A, int pb, int tid, double * const * diag, double * const * side, BlkMeta * bm)
{
  const double * tw = A.Tw + (size_t)pb * A.tw_len;
  for(int b = tid; b < A.nblk; b += NT)
  {
    bm[b].diag = diag[b];
    bm[b].side = (b < A.nblk - 1) ? side[b] : nullptr;
    bm[b].bs = A.bs[b];
    bm[b].boff = A.boff[b];
    bm[b].ldd = A.ldd[b];
    bm[b].lds = (b < A.nblk - 1) ? A.lds[b] : 0;
    bm[b].diagT = tw + A.tw_doff[b];
    bm[b].sideT = (b < A.nblk - 1) ? tw + A.tw_soff[b] : nullptr;
    bm[b].sct = (b < A.nblk - 1) ? A.tw_sc[b] : 0;
    bm[b].pad = 0;
  }
}

// wave_bwd (device_common.h) on the TRANSPOSED copy of the block: Lt(i, j) = L(j, i), so lane rl's operands L(k, rl) are Lt(rl, k) --
// consecutive addresses across the lanes (the original walks a column per lane: one cache line per lane and request).  Same steps.
__device__ __forceinline__ double wave_bwd_t(const double * Lt, int ldl, int n, double xi, int lane)
{
  n = __builtin_amdgcn_readfirstlane(n);
  const bool act = lane < n;
  const int rl = act ? lane : 0;
  const double dg = (n > 0) ? AT(Lt, ldl, rl, rl) : 1.0;
  const double rinv = recip_refined(dg);
  const double x0 = xi;
  double lv[kAhead], ln[kAhead];
#pragma unroll
  for(int j = 0; j < kAhead; ++j) ln[j] = (n > 0) ? AT(Lt, ldl, rl, max(n - 1 - j, 0)) : 0.0;
  for(int k0 = n - 1; k0 >= 0; k0 -= kAhead)
  {
#pragma unroll
    for(int j = 0; j < kAhead; ++j) lv[j] = ln[j];
    if(k0 - kAhead >= 0)
    {
#pragma unroll
      for(int j = 0; j < kAhead; ++j) ln[j] = AT(Lt, ldl, rl, max(k0 - kAhead - j, 0));
    }
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
        xi = FMA(-AT(Lt, ldl, rl, k), v, xi);
    }
  }
  return act ? xi : x0;
}

// One off-head block of an arrow's L^-T pass when the block and the head have at most 32 rows: acc_i = v_i - sum_k S(k, i) zh_k
// (k descending), then the backward substitution with the block (wave_bwd_t's steps) -- with the operands of BOTH requested up front:
// the coupling's 32 and the first 16 of the substitution go out together, the last 16 as soon as the coupling's registers are free, so
// a block costs about one memory round trip instead of four.  Sc(i, k) / Lt(i, k): transposed copies (consecutive lanes, consecutive
// addresses).  Same operations in the same order as chain_neg8 + wave_bwd_t: same bits.
__device__ __forceinline__ double arrow_block_bwd32(const double * Sc, int lds_, int nbk, const double * Lt, int ldl, int ni, const double * zh,
                                                    double acc, int lane)
{
  ni = __builtin_amdgcn_readfirstlane(ni);
  nbk = __builtin_amdgcn_readfirstlane(nbk);
  const bool act = lane < ni;
  const int rl = act ? lane : 0;
  double s0[16], s1[16], l0[16], l1[16];
#pragma unroll
  for(int j = 0; j < 16; ++j) s0[j] = AT(Sc, lds_, rl, max(nbk - 1 - j, 0));
#pragma unroll
  for(int j = 0; j < 16; ++j) s1[j] = AT(Sc, lds_, rl, max(nbk - 17 - j, 0));
  const double dg = AT(Lt, ldl, rl, rl);
#pragma unroll
  for(int j = 0; j < 16; ++j) l0[j] = AT(Lt, ldl, rl, max(ni - 1 - j, 0));
#pragma unroll
  for(int j = 0; j < 16; ++j)
  {
    const int k = nbk - 1 - j;
    const double t = FMA(-s0[j], zh[max(k, 0)], acc);
    acc = (k >= 0) ? t : acc;
  }
#pragma unroll
  for(int j = 0; j < 16; ++j)
  {
    const int k = nbk - 17 - j;
    const double t = FMA(-s1[j], zh[max(k, 0)], acc);
    acc = (k >= 0) ? t : acc;
  }
#pragma unroll
  for(int j = 0; j < 16; ++j) l1[j] = AT(Lt, ldl, rl, max(ni - 17 - j, 0));
  const double rinv = recip_refined(dg);
  const double x0 = acc;
  double xi = acc;
#pragma unroll
  for(int j = 0; j < 16; ++j)
  {
    const int k = ni - 1 - j; // (a step with k < 0 matches no lane)
    const double v = bcast_lane_s(div_pre(xi, dg, rinv), max(k, 0));
    const double upd = FMA(-l0[j], v, xi);
    xi = (lane < k) ? upd : xi;
  }
#pragma unroll
  for(int j = 0; j < 16; ++j)
  {
    const int k = ni - 17 - j;
    const double v = bcast_lane_s(div_pre(xi, dg, rinv), max(k, 0));
    const double upd = FMA(-l1[j], v, xi);
    xi = (lane < k) ? upd : xi;
  }
  const double pre = xi;
  xi = div_pre(pre, dg, rinv);
  if(!__all(!act || (div_safe_den(dg) && div_safe_num(pre))))
  {
    xi = x0;
    for(int k = ni - 1; k >= 0; --k)
    {
      const double v = bcast_lane(xi, k) / bcast_lane(dg, k);
      if(lane == k)
        xi = v;
      else if(lane < k)
        xi = FMA(-AT(Lt, ldl, rl, k), v, xi);
    }
  }
  return act ? xi : x0;
}

// arrow_block_bwd32 on TWO blocks at once: lanes 0 .. 31 work on block A, lanes 32 .. 63 on block B (niB = 0: no second block).  Half as
// many rounds per L^-T pass, and every request covers two blocks; a step's pivot comes from one v_readlane per half.  Per lane the
// operations are arrow_block_bwd32's: same bits.
__device__ __forceinline__ double arrow_block_bwd32x2(const double * ScA, int ldsA, const double * LtA, int niA, const double * ScB, int ldsB,
                                                      const double * LtB, int niB, int nbk, const double * zh, double acc, int lane)
{
  niA = __builtin_amdgcn_readfirstlane(niA);
  niB = __builtin_amdgcn_readfirstlane(niB);
  nbk = __builtin_amdgcn_readfirstlane(nbk);
  const bool hb = lane >= 32;
  const int l5 = lane & 31;
  const int ni = hb ? niB : niA;
  const bool act = l5 < ni;
  const int rl = act ? l5 : 0;
  const double * Sc = (hb && niB > 0) ? ScB : ScA;
  const double * Lt = (hb && niB > 0) ? LtB : LtA;
  const int lds_ = (hb && niB > 0) ? ldsB : ldsA, ldl = (hb && niB > 0) ? niB : niA, nl = (hb && niB > 0) ? niB : niA;
  double s0[16], s1[16], l0[16], l1[16];
#pragma unroll
  for(int j = 0; j < 16; ++j) s0[j] = AT(Sc, lds_, rl, max(nbk - 1 - j, 0));
#pragma unroll
  for(int j = 0; j < 16; ++j) s1[j] = AT(Sc, lds_, rl, max(nbk - 17 - j, 0));
  const double dg = AT(Lt, ldl, rl, rl);
#pragma unroll
  for(int j = 0; j < 16; ++j) l0[j] = AT(Lt, ldl, rl, max(nl - 1 - j, 0));
#pragma unroll
  for(int j = 0; j < 16; ++j)
  {
    const int k = nbk - 1 - j;
    const double t = FMA(-s0[j], zh[max(k, 0)], acc);
    acc = (k >= 0) ? t : acc;
  }
#pragma unroll
  for(int j = 0; j < 16; ++j)
  {
    const int k = nbk - 17 - j;
    const double t = FMA(-s1[j], zh[max(k, 0)], acc);
    acc = (k >= 0) ? t : acc;
  }
#pragma unroll
  for(int j = 0; j < 16; ++j) l1[j] = AT(Lt, ldl, rl, max(nl - 17 - j, 0));
  const double rinv = recip_refined(dg);
  const double x0 = acc;
  double xi = acc;
#pragma unroll
  for(int j = 0; j < 32; ++j)
  {
    const int kA = niA - 1 - j, kB = niB - 1 - j; // (a step with k < 0 matches no lane)
    const double qv = div_pre(xi, dg, rinv);
    const double vA = bcast_lane_s(qv, max(kA, 0)), vB = bcast_lane_s(qv, 32 + max(kB, 0));
    const double v = hb ? vB : vA;
    const double upd = FMA(-(j < 16 ? l0[j & 15] : l1[j & 15]), v, xi);
    xi = (l5 < (hb ? kB : kA)) ? upd : xi;
  }
  const double pre = xi;
  xi = div_pre(pre, dg, rinv);
  if(!__all(!act || (div_safe_den(dg) && div_safe_num(pre))))
  {
    xi = x0;
    for(int k = 31; k >= 0; --k)
    {
      const double vA = bcast_lane(xi, k) / bcast_lane(dg, k), vB = bcast_lane(xi, 32 + k) / bcast_lane(dg, 32 + k);
      const double v = hb ? vB : vA;
      if(k < ni)
      {
        if(l5 == k)
          xi = v;
        else if(l5 < k)
          xi = FMA(-AT(Lt, ldl, rl, k), v, xi);
      }
    }
  }
  return act ? xi : x0;
}

struct Ctx
{
  const BgiArgs * A;
  const BlkMeta * bm;
  int tid, lane, wave;
  double * const * diag;
  double * const * side;
  const double * const * cblk;
  double *d, *z, *x, *u, *r, *tri, *red;
  int * ired;
  int par; // parity of the reduction scratch slot (two slots: one barrier per reduction)
  double *pool, *Rp;
  SeqRec * desc;
  int *active, *status;
};

// ---- C4 reduction over the workgroup -------------------------------------------------------------------------------
template<int NT>
__device__ __forceinline__ double block_sum(Ctx & K, double s)
{
  constexpr int NW = NT / 64;
  s = wave_sum_tree(s);
  if(NW == 1) return s;
  double * slot = K.red + 16 * K.par;
  K.par ^= 1;
  if(K.lane == 0) slot[K.wave] = s;
  lds_barrier();
  double w[NW];
#pragma unroll
  for(int i = 0; i < NW; ++i) w[i] = slot[i];
#pragma unroll
  for(int st = 1; st < NW; st <<= 1)
#pragma unroll
    for(int i = 0; i < NW; i += 2 * st) w[i] = w[i] + w[i + st];
  return w[0];
}

// sum over e in [lo, hi) of fa(e) * fb(e); chain (e mod 1024) lives on thread (e mod 1024) / VPT
template<int NT, class FA, class FB>
__device__ __forceinline__ double dot_tree(Ctx & K, int lo, int hi, FA fa, FB fb)
{
  constexpr int VPT = 1024 / NT;
  double p[VPT];
#pragma unroll
  for(int i = 0; i < VPT; ++i) p[i] = 0.0;
  const int t0 = K.tid * VPT;
  for(int base = lo & ~1023; base < hi; base += 1024)
  {
#pragma unroll
    for(int i = 0; i < VPT; ++i)
    {
      const int e = base + t0 + i;
      if(e >= lo && e < hi) p[i] = FMA(fa(e), fb(e), p[i]);
    }
  }
#pragma unroll
  for(int st = 1; st < VPT; st <<= 1)
#pragma unroll
    for(int i = 0; i < VPT; i += 2 * st) p[i] = p[i] + p[i + st];
  return block_sum<NT>(K, p[0]);
}

// ---- StructuredG::solveL / solveInPlaceLTranspose on a vector held in LDS --------------------------------------------
// Block arrow, L v <- v with the non-zero window [start, end) of v (blockArrowLLT.cpp:92-176).  The up variant's row
// permutation (:163-170) is a change of offsets only: logical block i sits at physical rows boff[dsel(i)].
// [lo, hi]: the logical off-head blocks that can lie under the window (default: all; the C9 kernel knows the constraint's reach and
// spares the waves a scan of ~100 block descriptors per call).  Blocks of at most 32 rows take the operands of their substitution --
// and wave 0 those of the head's, before the barrier -- in one request (wave_fwd_pre): same steps, same bits.
template<int NT>
__device__ __forceinline__ void arrow_lsolve(Ctx & K, double * v, int start, int end, int lo = 0, int hi = 0x7fffffff)
{
  const BgiArgs & A = *K.A;
  constexpr int NW = NT / 64;
  const int up = A.kind == 1, nblk = A.nblk, lane = K.lane;
  const int ib = dsel(nblk, up, nblk - 1), nbk = K.bm[ib].bs;
  const int shift = up ? K.bm[0].bs : 0; // permuted row = physical row - shift for the off-head blocks
  if(up)
  {
    start = max(0, start - shift);
    end = max(0, end - shift);
  }
  hi = min(hi, nblk - 2);
  // the head's operands (wave 0), requested before anything else
  const bool hpre = nbk <= 32;
  double hl[32], hdg = 1.0;
  if(K.wave == 0 && hpre)
  {
    const double * Dh = K.bm[ib].diag;
    const int ldh = K.bm[ib].ldd, rl = (lane < nbk) ? lane : 0;
    hdg = AT(Dh, ldh, rl, rl);
#pragma unroll
    for(int j = 0; j < 32; ++j) hl[j] = AT(Dh, ldh, rl, min(j, nbk - 1));
  }
  for(int i = lo + ((K.wave - lo) % NW + NW) % NW; i <= hi; i += NW) // (block i on wave i mod NW, as the full loop deals them)
  {
    const int id = dsel(nblk, up, i), ni = K.bm[id].bs, n0 = K.bm[id].boff - shift;
    const int s = max(start - n0, 0);
    if(ni < s || end <= n0) continue; // (:121-127)
    const double * Dg = K.bm[id].diag;
    const int ldd = K.bm[id].ldd;
    double * vi = v + K.bm[id].boff + s;
    double xi = (lane < ni - s) ? vi[lane] : 0.0;
    if(ni - s <= 32)
    {
      const double * Ls = Dg + s + (size_t)s * ldd;
      const int nn = ni - s, rl = (lane < nn) ? lane : 0;
      double lv[32];
      const double dg = (nn > 0) ? AT(Ls, ldd, rl, rl) : 1.0;
#pragma unroll
      for(int j = 0; j < 32; ++j) lv[j] = (nn > 0) ? AT(Ls, ldd, rl, min(j, nn - 1)) : 0.0;
      xi = wave_fwd_pre<32>(lv, dg, Ls, ldd, nn, xi, lane);
    }
    else
      xi = wave_fwd(Dg + s + (size_t)s * ldd, ldd, ni - s, xi, lane);
    if(lane < ni - s) vi[lane] = xi;
  }
  __syncthreads();
  if(K.wave == 0)
  {
    double * vb = v + K.bm[ib].boff;
    double acc = (lane < nbk) ? vb[lane] : 0.0;
    for(int i = lo; i <= hi; ++i)
    {
      const int id = dsel(nblk, up, i), ni = K.bm[id].bs, n0 = K.bm[id].boff - shift;
      const int s = max(start - n0, 0);
      if(ni < s || end <= n0) continue;
      if(lane < nbk)
      {
        const double * Sg = K.bm[i].side;
        const int ldsg = K.bm[i].lds;
        const double * SgT = K.bm[i].sideT;
        const int sct = K.bm[i].sct;
        const double * Xi = v + K.bm[id].boff;
        acc = chain_neg8(s, 1, ni - s, acc, [&](int k) { return up ? AT(SgT, sct, lane, k) : AT(Sg, ldsg, lane, k); },
                         [&](int k) { return Xi[k]; });
      }
    }
    if(hpre)
      acc = wave_fwd_pre<32>(hl, hdg, K.bm[ib].diag, K.bm[ib].ldd, nbk, acc, lane);
    else
      acc = wave_fwd(K.bm[ib].diag, K.bm[ib].ldd, nbk, acc, lane);
    if(lane < nbk) vb[lane] = acc;
  }
  __syncthreads();
}

// L^T v <- v, full vector (blockArrowLLT.cpp:179-277 with start = 0, end = n)
template<int NT>
__device__ __forceinline__ void arrow_ltsolve(Ctx & K, double * v)
{
  const BgiArgs & A = *K.A;
  constexpr int NW = NT / 64;
  const int up = A.kind == 1, nblk = A.nblk, lane = K.lane;
  const int ib = dsel(nblk, up, nblk - 1), nbk = K.bm[ib].bs;
  double * vb = v + K.bm[ib].boff;
  if(K.wave == 0)
  {
    double xi = (lane < nbk) ? vb[lane] : 0.0;
    xi = wave_bwd_t(K.bm[ib].diagT, nbk, nbk, xi, lane);
    if(lane < nbk) vb[lane] = xi;
  }
  __syncthreads();
  if(A.maxb <= 32 && !(A.tune & 4))
  {
    // every block has at most 32 rows: two blocks per wave and round (arrow_block_bwd32x2)
    const int hb = lane >> 5, l5 = lane & 31;
    for(int p = K.wave; 2 * p < nblk - 1; p += NW)
    {
      const int iA = 2 * p, iB = min(2 * p + 1, nblk - 2);
      const bool hasB = 2 * p + 1 < nblk - 1;
      const int idA = dsel(nblk, up, iA), idB = dsel(nblk, up, iB);
      const int niA = K.bm[idA].bs, niB = hasB ? K.bm[idB].bs : 0;
      const int i = hb ? iB : iA, id = hb ? idB : idA, ni = hb ? niB : niA;
      double * vi = v + K.bm[id].boff;
      const double acc = (l5 < ni) ? vi[l5] : 0.0;
      const double solved = (A.tune & 1) ? acc
                                         : arrow_block_bwd32x2(up ? K.bm[iA].side : K.bm[iA].sideT, up ? K.bm[iA].lds : K.bm[iA].sct, K.bm[idA].diagT, niA,
                                                               up ? K.bm[iB].side : K.bm[iB].sideT, up ? K.bm[iB].lds : K.bm[iB].sct, K.bm[idB].diagT, niB,
                                                               nbk, vb, acc, lane);
      (void)i;
      if(l5 < ni) vi[l5] = solved;
    }
    __syncthreads();
    return;
  }
  for(int i = K.wave; i < nblk - 1; i += NW)
  {
    const int id = dsel(nblk, up, i), ni = K.bm[id].bs;
    const double * Sg = K.bm[i].side;
    const int ldsg = K.bm[i].lds;
    const double * SgT = K.bm[i].sideT;
    const int sct = K.bm[i].sct;
    double * vi = v + K.bm[id].boff;
    double acc = (lane < ni) ? vi[lane] : 0.0;
    double solved;
    if(A.tune & 1)
      solved = acc; // (timing experiment: no coupling, no substitution)
    else if(ni <= 32 && nbk <= 32)
      solved = arrow_block_bwd32(up ? Sg : SgT, up ? ldsg : sct, nbk, K.bm[id].diagT, ni, ni, vb, acc, lane);
    else
    {
      const int rl = (lane < ni) ? lane : 0; // idle lanes run the same (discarded) chain: no divergent loads
      acc = chain_neg8(nbk - 1, -1, nbk, acc, [&](int k) { return up ? AT(Sg, ldsg, rl, k) : AT(SgT, sct, rl, k); },
                       [&](int k) { return vb[k]; });
      solved = wave_bwd_t(K.bm[id].diagT, ni, ni, acc, lane);
    }
    if(lane < ni) vi[lane] = solved;
  }
  __syncthreads();
}

// Block tridiagonal (triBlockDiagLLT.cpp:37-96): serial over the blocks, one wave
__device__ __forceinline__ void tri_lsolve(Ctx & K, double * v, int start)
{
  const BgiArgs & A = *K.A;
  const int lane = K.lane;
  if(K.wave == 0)
  {
    int n = 0, l = 0, li = 0;
    bool zero = true;
    for(int i = 0; i < A.nblk; ++i)
    {
      const int ni = K.bm[i].bs;
      const double * Li = K.bm[i].diag;
      const int ldd = K.bm[i].ldd;
      if(n + ni >= start)
      {
        if(zero)
        {
          const int rr = n + ni - start, off = ni - rr;
          double xi = (lane < rr) ? v[start + lane] : 0.0;
          xi = wave_fwd(Li + off + (size_t)off * ldd, ldd, rr, xi, lane);
          if(lane < rr) v[start + lane] = xi;
          zero = false;
        }
        else
        {
          const double * Sg = K.bm[i - 1].side;
          const int ldsg = K.bm[i - 1].lds;
          double acc = (lane < ni) ? v[n + lane] : 0.0;
          {
            const int rl = (lane < ni) ? lane : 0;
            acc = chain_neg8(0, 1, li, acc, [&](int k) { return AT(Sg, ldsg, rl, k); }, [&](int k) { return v[l + k]; });
          }
          acc = wave_fwd(Li, ldd, ni, acc, lane);
          if(lane < ni) v[n + lane] = acc;
        }
        wave_sync();
      }
      l = n;
      li = ni;
      n += ni;
    }
  }
  __syncthreads();
}

// triBlockDiagLTransposeSolve (triBlockDiagLLT.cpp:98-158), full vector
__device__ __forceinline__ void tri_ltsolve(Ctx & K, double * v)
{
  const BgiArgs & A = *K.A;
  const int lane = K.lane;
  if(K.wave == 0)
  {
    int n = A.n, l = 0, li = 0;
    bool zero = true;
    for(int i = A.nblk - 1; i >= 0; --i)
    {
      const int ni = K.bm[i].bs;
      const double * Li = K.bm[i].diagT; // (the transposed copy: wave_bwd_t)
      const int ldd = ni;
      if(zero)
      {
        double xi = (lane < ni) ? v[n - ni + lane] : 0.0;
        xi = wave_bwd_t(Li, ldd, ni, xi, lane);
        if(lane < ni) v[n - ni + lane] = xi;
        zero = false;
      }
      else
      {
        const double * Sg = K.bm[i].sideT;
        const int ldsg = K.bm[i].sct;
        double acc = (lane < ni) ? v[n - ni + lane] : 0.0;
        {
          const int rl = (lane < ni) ? lane : 0;
          acc = chain_neg8(li - 1, -1, li, acc, [&](int k) { return AT(Sg, ldsg, rl, k); }, [&](int k) { return v[(l - li) + k]; });
        }
        acc = wave_bwd_t(Li, ldd, ni, acc, lane);
        if(lane < ni) v[n - ni + lane] = acc;
      }
      wave_sync();
      l = n;
      li = ni;
      n -= ni;
    }
  }
  __syncthreads();
}

// ---- the same four solves for blocks of 65 .. 128 rows (round 4): two rows per lane (wave_fwd2 / wave_bwd2, device_common.h); chosen
// per launch by BgiArgs::maxb so that problems with small blocks keep the single-entry code above and its register count
// (StructuredG::solveL / solveInPlaceLTranspose --------------------------------------------
// Block arrow, L v <- v with the non-zero window [start, end) of v (blockArrowLLT.cpp:92-176).  The up variant's row
// permutation (:163-170) is a change of offsets only: logical block i sits at physical rows boff[dsel(i)].
template<int NT>
__device__ __forceinline__ void arrow_lsolve_big(Ctx & K, double * v, int start, int end)
{
  const BgiArgs & A = *K.A;
  constexpr int NW = NT / 64;
  const int up = A.kind == 1, nblk = A.nblk, lane = K.lane;
  const int ib = dsel(nblk, up, nblk - 1), nbk = K.bm[ib].bs;
  const int shift = up ? K.bm[0].bs : 0; // permuted row = physical row - shift for the off-head blocks
  if(up)
  {
    start = max(0, start - shift);
    end = max(0, end - shift);
  }
  for(int i = K.wave; i < nblk - 1; i += NW)
  {
    const int id = dsel(nblk, up, i), ni = K.bm[id].bs, n0 = K.bm[id].boff - shift;
    const int s = max(start - n0, 0);
    if(ni < s || end <= n0) continue; // (:121-127)
    const double * Dg = K.bm[id].diag;
    const int ldd = K.bm[id].ldd;
    double * vi = v + K.bm[id].boff + s;
    const int nr = ni - s;
    double xa = (lane < nr) ? vi[lane] : 0.0, xb = (lane + 64 < nr) ? vi[lane + 64] : 0.0; // (rows lane, lane + 64: blocks of up to 128 rows)
    wave_fwd2(Dg + s + (size_t)s * ldd, ldd, nr, xa, xb, lane);
    if(lane < nr) vi[lane] = xa;
    if(lane + 64 < nr) vi[lane + 64] = xb;
  }
  __syncthreads();
  if(K.wave == 0)
  {
    double * vb = v + K.bm[ib].boff;
    double acc[2]; // rows lane, lane + 64 of the head block
#pragma unroll
    for(int h = 0; h < 2; ++h) acc[h] = (lane + 64 * h < nbk) ? vb[lane + 64 * h] : 0.0;
    for(int i = 0; i < nblk - 1; ++i)
    {
      const int id = dsel(nblk, up, i), ni = K.bm[id].bs, n0 = K.bm[id].boff - shift;
      const int s = max(start - n0, 0);
      if(ni < s || end <= n0) continue;
      const double * Sg = K.bm[i].side;
      const int ldsg = K.bm[i].lds;
      const double * Xi = v + K.bm[id].boff;
#pragma unroll
      for(int h = 0; h < 2; ++h)
      {
        const int r = lane + 64 * h;
        if(r < nbk)
          acc[h] = chain_neg8(s, 1, ni - s, acc[h], [&](int k) { return up ? AT(Sg, ldsg, k, r) : AT(Sg, ldsg, r, k); },
                              [&](int k) { return Xi[k]; });
      }
    }
    wave_fwd2(K.bm[ib].diag, K.bm[ib].ldd, nbk, acc[0], acc[1], lane);
#pragma unroll
    for(int h = 0; h < 2; ++h)
      if(lane + 64 * h < nbk) vb[lane + 64 * h] = acc[h];
  }
  __syncthreads();
}

// L^T v <- v, full vector (blockArrowLLT.cpp:179-277 with start = 0, end = n)
template<int NT>
__device__ __forceinline__ void arrow_ltsolve_big(Ctx & K, double * v)
{
  const BgiArgs & A = *K.A;
  constexpr int NW = NT / 64;
  const int up = A.kind == 1, nblk = A.nblk, lane = K.lane;
  const int ib = dsel(nblk, up, nblk - 1), nbk = K.bm[ib].bs;
  double * vb = v + K.bm[ib].boff;
  if(K.wave == 0)
  {
    double xa = (lane < nbk) ? vb[lane] : 0.0, xb = (lane + 64 < nbk) ? vb[lane + 64] : 0.0;
    wave_bwd2(K.bm[ib].diag, K.bm[ib].ldd, nbk, xa, xb, lane);
    if(lane < nbk) vb[lane] = xa;
    if(lane + 64 < nbk) vb[lane + 64] = xb;
  }
  __syncthreads();
  for(int i = K.wave; i < nblk - 1; i += NW)
  {
    const int id = dsel(nblk, up, i), ni = K.bm[id].bs;
    const double * Sg = K.bm[i].side;
    const int ldsg = K.bm[i].lds;
    double * vi = v + K.bm[id].boff;
    double acc[2]; // rows lane, lane + 64 of block i
#pragma unroll
    for(int h = 0; h < 2; ++h)
    {
      const int r = lane + 64 * h;
      acc[h] = (r < ni) ? vi[r] : 0.0;
      if(h == 0 || ni > 64) // (uniform)
      {
        const int rl = (r < ni) ? r : 0; // idle lanes run the same (discarded) chain: no divergent loads
        const double t = chain_neg8(nbk - 1, -1, nbk, acc[h], [&](int k) { return up ? AT(Sg, ldsg, rl, k) : AT(Sg, ldsg, k, rl); },
                                    [&](int k) { return vb[k]; });
        acc[h] = (r < ni) ? t : acc[h];
      }
    }
    wave_bwd2(K.bm[id].diag, K.bm[id].ldd, ni, acc[0], acc[1], lane);
#pragma unroll
    for(int h = 0; h < 2; ++h)
      if(lane + 64 * h < ni) vi[lane + 64 * h] = acc[h];
  }
  __syncthreads();
}

// Block tridiagonal (triBlockDiagLLT.cpp:37-96): serial over the blocks, one wave
__device__ __forceinline__ void tri_lsolve_big(Ctx & K, double * v, int start)
{
  const BgiArgs & A = *K.A;
  const int lane = K.lane;
  if(K.wave == 0)
  {
    int n = 0, l = 0, li = 0;
    bool zero = true;
    for(int i = 0; i < A.nblk; ++i)
    {
      const int ni = K.bm[i].bs;
      const double * Li = K.bm[i].diag;
      const int ldd = K.bm[i].ldd;
      if(n + ni >= start)
      {
        if(zero)
        {
          const int rr = n + ni - start, off = ni - rr;
          double xa = (lane < rr) ? v[start + lane] : 0.0, xb = (lane + 64 < rr) ? v[start + lane + 64] : 0.0;
          wave_fwd2(Li + off + (size_t)off * ldd, ldd, rr, xa, xb, lane);
          if(lane < rr) v[start + lane] = xa;
          if(lane + 64 < rr) v[start + lane + 64] = xb;
          zero = false;
        }
        else
        {
          const double * Sg = K.bm[i - 1].side;
          const int ldsg = K.bm[i - 1].lds;
          double acc[2]; // rows lane, lane + 64 of block i
#pragma unroll
          for(int h = 0; h < 2; ++h)
          {
            const int r = lane + 64 * h;
            acc[h] = (r < ni) ? v[n + r] : 0.0;
            if(h == 0 || ni > 64) // (uniform)
            {
              const int rl = (r < ni) ? r : 0;
              const double t = chain_neg8(0, 1, li, acc[h], [&](int k) { return AT(Sg, ldsg, rl, k); }, [&](int k) { return v[l + k]; });
              acc[h] = (r < ni) ? t : acc[h];
            }
          }
          wave_fwd2(Li, ldd, ni, acc[0], acc[1], lane);
#pragma unroll
          for(int h = 0; h < 2; ++h)
            if(lane + 64 * h < ni) v[n + lane + 64 * h] = acc[h];
        }
        wave_sync();
      }
      l = n;
      li = ni;
      n += ni;
    }
  }
  __syncthreads();
}

// triBlockDiagLTransposeSolve (triBlockDiagLLT.cpp:98-158), full vector
__device__ __forceinline__ void tri_ltsolve_big(Ctx & K, double * v)
{
  const BgiArgs & A = *K.A;
  const int lane = K.lane;
  if(K.wave == 0)
  {
    int n = A.n, l = 0, li = 0;
    bool zero = true;
    for(int i = A.nblk - 1; i >= 0; --i)
    {
      const int ni = K.bm[i].bs;
      const double * Li = K.bm[i].diag;
      const int ldd = K.bm[i].ldd;
      if(zero)
      {
        double xa = (lane < ni) ? v[n - ni + lane] : 0.0, xb = (lane + 64 < ni) ? v[n - ni + lane + 64] : 0.0;
        wave_bwd2(Li, ldd, ni, xa, xb, lane);
        if(lane < ni) v[n - ni + lane] = xa;
        if(lane + 64 < ni) v[n - ni + lane + 64] = xb;
        zero = false;
      }
      else
      {
        const double * Sg = K.bm[i].side;
        const int ldsg = K.bm[i].lds;
        double acc[2]; // rows lane, lane + 64 of block i
#pragma unroll
        for(int h = 0; h < 2; ++h)
        {
          const int r = lane + 64 * h;
          acc[h] = (r < ni) ? v[n - ni + r] : 0.0;
          if(h == 0 || ni > 64) // (uniform)
          {
            const int rl = (r < ni) ? r : 0;
            const double t = chain_neg8(li - 1, -1, li, acc[h], [&](int k) { return AT(Sg, ldsg, k, rl); }, [&](int k) { return v[(l - li) + k]; });
            acc[h] = (r < ni) ? t : acc[h];
          }
        }
        wave_bwd2(Li, ldd, ni, acc[0], acc[1], lane);
#pragma unroll
        for(int h = 0; h < 2; ++h)
          if(lane + 64 * h < ni) v[n - ni + lane + 64 * h] = acc[h];
      }
      wave_sync();
      l = n;
      li = ni;
      n -= ni;
    }
  }
  __syncthreads();
}

// The up arrow's row permutation (blockArrowLLT.cpp:163-170, :266-271): blockArrowLSolve returns its result in the
// PERMUTED order (first block moved last) and blockArrowLTransposeSolve expects its argument in that order -- d, and
// with it the orthonormal sequence, live in permuted coordinates.  forward: v <- [v(n0:), v(:n0)]; backward: the inverse.
template<int NT>
__device__ __forceinline__ void permute_up(Ctx & K, double * v, bool forward)
{
  constexpr int PER = kBgiMaxN / NT;
  const int n = K.A->n, n0 = K.bm[0].bs, s = n - n0;
  double tmp[PER];
#pragma unroll
  for(int j = 0; j < PER; ++j)
  {
    const int e = K.tid + j * NT;
    int src;
    if(forward)
      src = (e < s) ? e + n0 : e - s;
    else
      src = (e < n0) ? e + s : e - n0;
    tmp[j] = (e < n) ? v[src] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for(int j = 0; j < PER; ++j)
  {
    const int e = K.tid + j * NT;
    if(e < n) v[e] = tmp[j];
  }
  __syncthreads();
}

// StructuredG::solveL (StructuredG.cpp:67-113); permuted = false keeps the result in the original row order (only
// valid when the matching sg_ltsolve call is told the same: the two permutations of the reference cancel)
template<int NT, bool BIG>
__device__ __forceinline__ void sg_lsolve(Ctx & K, double * v, int start, int end, bool permuted = true, int lo = 0, int hi = 0x7fffffff)
{
  constexpr bool big = BIG; // (a kernel of its own for blocks of 65 .. 128 rows: the two-rows-per-lane variants cost the small-block kernel
                            // 5 % when they merely shared its code -- registers and instruction cache)
  if(K.A->kind == 2)
  {
    if(big)
      tri_lsolve_big(K, v, start);
    else
      tri_lsolve(K, v, start);
  }
  else
  {
    if(big)
      arrow_lsolve_big<NT>(K, v, start, end);
    else
      arrow_lsolve<NT>(K, v, start, end, lo, hi);
    if(K.A->kind == 1 && permuted) permute_up<NT>(K, v, true);
  }
}
// StructuredG::solveInPlaceLTranspose (StructuredG.cpp:45-65)
template<int NT, bool BIG>
__device__ __forceinline__ void sg_ltsolve(Ctx & K, double * v, bool permuted = true)
{
  constexpr bool big = BIG;
  if(K.A->kind == 2)
  {
    if(big)
      tri_ltsolve_big(K, v);
    else
      tri_ltsolve(K, v);
  }
  else
  {
    if(K.A->kind == 1 && permuted) permute_up<NT>(K, v, false);
    if(big)
      arrow_ltsolve_big<NT>(K, v);
    else
      arrow_ltsolve<NT>(K, v);
  }
}

// ---- lexicographic (value, index) minimum over the workgroup: strict '<' with first index winning ---------------------
template<int NT>
__device__ __forceinline__ void block_argmin(Ctx & K, double & val, int & idx, int & tag)
{
  constexpr int NW = NT / 64;
#pragma unroll
  for(int off = 32; off >= 1; off >>= 1)
  {
    const double ov = shfl_xor_d(val, off);
    const int oi = __shfl_xor(idx, off), ot = __shfl_xor(tag, off);
    if(ov < val || (ov == val && oi < idx))
    {
      val = ov;
      idx = oi;
      tag = ot;
    }
  }
  if(NW == 1) return;
  double * slot = K.red + 16 * K.par;
  int * islot = K.ired + 32 * K.par;
  K.par ^= 1;
  if(K.lane == 0)
  {
    slot[K.wave] = val;
    islot[K.wave] = idx;
    islot[16 + K.wave] = tag;
  }
  lds_barrier();
  val = slot[0];
  idx = islot[0];
  tag = islot[16];
#pragma unroll
  for(int w = 1; w < NW; ++w)
  {
    const double ov = slot[w];
    const int oi = islot[w], ot = islot[16 + w];
    if(ov < val || (ov == val && oi < idx))
    {
      val = ov;
      idx = oi;
      tag = ot;
    }
  }
}

// n+ . v without the sign: SingleNZSegmentVector::dot (SingleNZSegmentVector.h:52-56) or the bound's row
__device__ __forceinline__ double cstr_dot(const Ctx & K, int p, const double * v)
{
  const BgiArgs & A = *K.A;
  if(p >= A.m) return v[p - A.m];
  const int b = A.to_block[p], len = A.cvar[b];
  const double * c = K.cblk[b] + (size_t)(p - A.cum_cstr[b]) * A.cld[b];
  const double * w = v + A.cum_var[b];
  return chain_neg8(0, 1, len, 0.0, [&](int k) { return -c[k]; }, [&](int k) { return w[k]; });
}

