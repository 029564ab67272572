// microbenchmark: cycles of a 100-term fma chain out of LDS (lane = column of a column-major n x n array, ld = n|1; second
// operand a broadcast vector), the forms gi_batch_big.hip has used.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off chain_lat.hip -o chain_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define FMA(a, b, c) __builtin_fma(a, b, c)
template<int CH>
__device__ __forceinline__ double chain_simple(const double * a, long sa, const double * b, long sb, int k0, int k1, double acc)
{
  int k = k0;
  for(; k + CH <= k1; k += CH)
  {
    double av[CH], bv[CH];
#pragma unroll
    for(int u = 0; u < CH; ++u) { av[u] = a[(k + u) * sa]; bv[u] = b[(k + u) * sb]; }
#pragma unroll
    for(int u = 0; u < CH; ++u) acc = FMA(av[u], bv[u], acc);
  }
  for(; k < k1; ++k) acc = FMA(a[k * sa], b[k * sb], acc);
  return acc;
}
template<int CH>
__device__ __forceinline__ double chain_pipe(const double * a, long sa, const double * b, long sb, int k0, int k1, double acc)
{
  const int cnt = k1 - k0;
  if(cnt <= 0) return acc;
  const int nfull = cnt / CH, ng = (cnt + CH - 1) / CH;
  const double * pa = a + (long)k0 * sa;
  const double * pb = b + (long)k0 * sb;
  double a0[CH], b0[CH], a1[CH], b1[CH];
  auto fetch_full = [&](double (&av)[CH], double (&bv)[CH], int g) __attribute__((always_inline)) {
    const double * ga = pa + (long)g * CH * sa;
    const double * gb = pb + (long)g * CH * sb;
#pragma unroll
    for(int u = 0; u < CH; ++u) { av[u] = ga[u * sa]; bv[u] = gb[u * sb]; }
  };
  auto fetch_any = [&](double (&av)[CH], double (&bv)[CH], int g) __attribute__((always_inline)) {
#pragma unroll
    for(int u = 0; u < CH; ++u) { const int w = min(g * CH + u, cnt - 1); av[u] = pa[(long)w * sa]; bv[u] = pb[(long)w * sb]; }
  };
  auto run_full = [&](const double (&av)[CH], const double (&bv)[CH]) __attribute__((always_inline)) {
#pragma unroll
    for(int u = 0; u < CH; ++u) acc = FMA(av[u], bv[u], acc);
  };
  auto run_sel = [&](const double (&av)[CH], const double (&bv)[CH], int g) __attribute__((always_inline)) {
#pragma unroll
    for(int u = 0; u < CH; ++u) { const double t = FMA(av[u], bv[u], acc); acc = (g * CH + u < cnt) ? t : acc; }
  };
  int g = 0;
  if(nfull >= 3)
  {
    fetch_full(a0, b0, 0);
    for(; g + 3 <= nfull; g += 2) { fetch_full(a1, b1, g + 1); run_full(a0, b0); fetch_full(a0, b0, g + 2); run_full(a1, b1); }
  }
  else
    fetch_any(a0, b0, 0);
  if(g < ng)
  {
    fetch_any(a1, b1, g + 1);
    run_sel(a0, b0, g);
    if(g + 1 < ng) { fetch_any(a0, b0, g + 2); run_sel(a1, b1, g + 1); if(g + 2 < ng) run_sel(a0, b0, g + 2); }
  }
  return acc;
}
// the broadcast operand held in registers (lane e: b[e], b[e + 64]) and handed out by v_readlane: only the per-lane operand moves through LDS
__device__ __forceinline__ double rl(double v, int src)
{
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template<int CH>
__device__ __forceinline__ double chain_rl(const double * a, long sa, double b0, double b1, int k0, int k1, double acc)
{
  // k0 = 0, k1 = n <= 128 here
  int k = k0;
  for(; k + CH <= k1 && k + CH <= 64; k += CH)
  {
    double av[CH];
#pragma unroll
    for(int u = 0; u < CH; ++u) av[u] = a[(k + u) * sa];
#pragma unroll
    for(int u = 0; u < CH; ++u) acc = FMA(av[u], rl(b0, k + u), acc);
  }
  for(; k + CH <= k1; k += CH)
  {
    double av[CH];
#pragma unroll
    for(int u = 0; u < CH; ++u) av[u] = a[(k + u) * sa];
#pragma unroll
    for(int u = 0; u < CH; ++u) acc = FMA(av[u], rl(b1, k + u - 64), acc);
  }
  for(; k < k1; ++k) acc = FMA(a[k * sa], rl(k < 64 ? b0 : b1, k & 63), acc);
  return acc;
}
// b through SGPRs: the broadcast operand read once per group by lane 0 ... (here: every lane reads, then readfirstlane)
template<int MODE>
__global__ __launch_bounds__(256) void k(int n, int reps, int active, long long * out, double * sink, const double * __restrict__ gv)
{
  extern __shared__ double sm[];
  const int ld = n | 1, tid = threadIdx.x;
  double * J = sm, * v = sm + (long)n * ld, * d = v + n;
  for(int e = tid; e < n * ld; e += 256) J[e] = 1.0 / (1 + (e % 97));
  for(int e = tid; e < n; e += 256) v[e] = 0.5 + e;
  __syncthreads();
  double acc = 0.0;
  long long t0 = clock64();
  for(int r = 0; r < reps; ++r)
  {
    if(tid < active && tid < n)
    {
      double dj;
      if(MODE == 0) dj = chain_simple<8>(J + (long)tid * ld, 1, v, 1, 0, n, acc * 1e-300);
      if(MODE == 1) dj = chain_pipe<8>(J + (long)tid * ld, 1, v, 1, 0, n, acc * 1e-300);
      if(MODE == 2) dj = chain_simple<8>(J + tid, ld, v, 1, 0, n, acc * 1e-300);   // row of J (z): stride ld
      if(MODE == 3) dj = chain_pipe<8>(J + tid, ld, v, 1, 0, n, acc * 1e-300);
      if(MODE == 4) dj = chain_simple<16>(J + (long)tid * ld, 1, v, 1, 0, n, acc * 1e-300);
      if(MODE == 5) dj = chain_simple<4>(J + (long)tid * ld, 1, v, 1, 0, n, acc * 1e-300);
      if(MODE == 6) dj = chain_rl<8>(J + (long)tid * ld, 1, v[threadIdx.x & 63], v[(threadIdx.x & 63) + 36], 0, n, acc * 1e-300);
      if(MODE == 7) dj = chain_rl<8>(J + tid, ld, v[threadIdx.x & 63], v[(threadIdx.x & 63) + 36], 0, n, acc * 1e-300);
      if(MODE == 8) dj = chain_simple<8>(J + (long)tid * ld, 1, gv + (r & 7) * n, 1, 0, n, acc * 1e-300); // b: a wave-uniform vector in GLOBAL memory (const __restrict__)
      if(MODE == 9) dj = chain_simple<16>(J + (long)tid * ld, 1, gv + (r & 7) * n, 1, 0, n, acc * 1e-300);
      d[tid] = dj;
      acc += dj;
    }
    __syncthreads();
  }
  long long t1 = clock64();
  if(tid == 0) out[0] = (t1 - t0) / reps;
  sink[tid] = acc;
}
int main()
{
  long long * out; double * sink;
  double * gv; (void)hipMalloc(&out, 8); (void)hipMalloc(&sink, 8 * 256); (void)hipMalloc(&gv, 8 * 1024); (void)hipMemset(gv, 0, 8 * 1024);
  const int n = 100, reps = 2000;
  size_t lds = sizeof(double) * ((size_t)n * (n | 1) + 2 * n);
  const char * names[] = {"simple<8> column (d)", "pipelined<8> column (d)", "simple<8> row (z)", "pipelined<8> row (z)", "simple<16> column", "simple<4> column", "readlane<8> column (d)", "readlane<8> row (z)", "b from global memory <8>", "b from global memory <16>"};
  for(int active : {100, 64})
    for(int mode = 0; mode < 10; ++mode)
    {
      long long h = 0;
      for(int rep = 0; rep < 2; ++rep)
      {
        switch(mode)
        {
          case 0: hipLaunchKernelGGL(k<0>, 1, 256, lds, 0, n, reps, active, out, sink, gv); break;
          case 1: hipLaunchKernelGGL(k<1>, 1, 256, lds, 0, n, reps, active, out, sink, gv); break;
          case 2: hipLaunchKernelGGL(k<2>, 1, 256, lds, 0, n, reps, active, out, sink, gv); break;
          case 3: hipLaunchKernelGGL(k<3>, 1, 256, lds, 0, n, reps, active, out, sink, gv); break;
          case 4: hipLaunchKernelGGL(k<4>, 1, 256, lds, 0, n, reps, active, out, sink, gv); break;
          case 5: hipLaunchKernelGGL(k<5>, 1, 256, lds, 0, n, reps, active, out, sink, gv); break;
          case 6: hipLaunchKernelGGL(k<6>, 1, 256, lds, 0, n, reps, active, out, sink, gv); break;
          case 7: hipLaunchKernelGGL(k<7>, 1, 256, lds, 0, n, reps, active, out, sink, gv); break;
          case 8: hipLaunchKernelGGL(k<8>, 1, 256, lds, 0, n, reps, active, out, sink, gv); break;
          case 9: hipLaunchKernelGGL(k<9>, 1, 256, lds, 0, n, reps, active, out, sink, gv); break;
        }
        hipDeviceSynchronize();
        hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
      }
      printf("active %3d  %-26s %6lld clock64 ticks per 100-term chain + barrier\n", active, names[mode], h);
    }
  return 0;
}
