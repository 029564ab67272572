// microbenchmark: the constraint scan of gi_batch_big.hip -- thread = constraint, a 100-term chain c_i . x, C from global memory (L2),
// x from LDS -- as it is (every thread walks its own column: a load instruction touches 64 cache lines) and with a transposed C
// (coalesced: 8 lines per instruction).   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scan_lat.hip -o scan_lat.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define FMA(a, b, c) __builtin_fma(a, b, c)
template<int MODE>
__global__ __launch_bounds__(256) void k(int n, int m, int reps, const double * __restrict__ C, const double * __restrict__ Ct, long long * out, double * sink)
{
  __shared__ double xs[128];
  const int tid = threadIdx.x;
  const double * Cb = C + (size_t)blockIdx.x * n * m;
  const double * Ctb = Ct + (size_t)blockIdx.x * n * m;
  if(tid < n) xs[tid] = 0.5 + tid;
  __syncthreads();
  double tot = 0.0;
  long long t0 = clock64();
  for(int r = 0; r < reps; ++r)
  {
    double acc = tot * 1e-300;
    if(tid < m)
    {
      if(MODE == 0) // own column, 32 terms in flight (as in the kernel)
      {
        const double * a = Cb + (size_t)tid * n;
        int kk = 0;
        for(; kk + 32 <= n; kk += 32)
        {
          double av[32], bv[32];
#pragma unroll
          for(int u = 0; u < 32; ++u) { av[u] = a[kk + u]; bv[u] = xs[kk + u]; }
#pragma unroll
          for(int u = 0; u < 32; ++u) acc = FMA(av[u], bv[u], acc);
        }
        for(; kk < n; ++kk) acc = FMA(a[kk], xs[kk], acc);
      }
      if(MODE == 1) // transposed: Ct[k * m + i], 32 terms in flight
      {
        const double * a = Ctb + tid;
        int kk = 0;
        for(; kk + 32 <= n; kk += 32)
        {
          double av[32], bv[32];
#pragma unroll
          for(int u = 0; u < 32; ++u) { av[u] = a[(size_t)(kk + u) * m]; bv[u] = xs[kk + u]; }
#pragma unroll
          for(int u = 0; u < 32; ++u) acc = FMA(av[u], bv[u], acc);
        }
        for(; kk < n; ++kk) acc = FMA(a[(size_t)kk * m], xs[kk], acc);
      }
    }
    tot += acc;
    __syncthreads();
  }
  long long t1 = clock64();
  if(tid == 0) out[blockIdx.x] = (t1 - t0) / reps;
  sink[blockIdx.x * 256 + tid] = tot;
}
int main()
{
  const int n = 100, m = 200, reps = 200, nwg = 256;
  double *C, *Ct, *sink; long long * out;
  (void)hipMalloc(&C, sizeof(double) * n * m * nwg); (void)hipMalloc(&Ct, sizeof(double) * n * m * nwg);
  (void)hipMemset(C, 0, sizeof(double) * n * m * nwg); (void)hipMemset(Ct, 0, sizeof(double) * n * m * nwg);
  (void)hipMalloc(&sink, 8 * 256 * nwg); (void)hipMalloc(&out, 8 * nwg);
  const char * names[] = {"own column (as in the kernel)", "transposed C (coalesced)"};
  for(int wgs : {1, 256})
    for(int mode = 0; mode < 2; ++mode)
    {
      long long h[256];
      for(int rep = 0; rep < 2; ++rep)
      {
        if(mode == 0) hipLaunchKernelGGL(k<0>, wgs, 256, 0, 0, n, m, reps, C, Ct, out, sink);
        else hipLaunchKernelGGL(k<1>, wgs, 256, 0, 0, n, m, reps, C, Ct, out, sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, out, 8 * wgs, hipMemcpyDeviceToHost);
      }
      long long mx = 0; for(int i = 0; i < wgs; ++i) mx = h[i] > mx ? h[i] : mx;
      printf("%3d workgroups  %-32s %6lld cycles per scan of %d constraints x %d terms (slowest workgroup)\n", wgs, names[mode], mx, m, n);
    }
  return 0;
}
