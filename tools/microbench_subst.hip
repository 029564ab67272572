// Microbenchmark of the wave-level substitution (device_common.h wave_fwd / chain_neg8) on one wave: cycles per block of 43
// with the factor in LDS and in global memory.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I jrl-qp_amd/csrc ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "device_common.h"
using namespace jrlqp;
__global__ void k(const double * Lg, int n, int reps, long long * out, double * sink)
{
  extern __shared__ double sm[];
  const int lane = threadIdx.x, ld = n | 1;
  for(int e = lane; e < n * ld; e += 64) sm[e] = Lg[e];
  double * v = sm + n * ld;
  if(lane < n) v[lane] = 1.0 + lane;
  __syncthreads();
  double acc = lane < n ? v[lane] : 0.0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for(int r = 0; r < reps; ++r) acc = wave_fwd(sm, ld, n, acc * 1e-3 + 1.0, lane);
  long long t1 = __builtin_amdgcn_s_memtime();
  for(int r = 0; r < reps; ++r) acc = wave_fwd(Lg, ld, n, acc * 1e-3 + 1.0, lane);
  long long t2 = __builtin_amdgcn_s_memtime();
  const int rl = lane < n ? lane : 0;
  for(int r = 0; r < reps; ++r) acc = chain_neg8(0, 1, n, acc * 1e-3, [&](int kk) { return sm[rl + kk * ld]; }, [&](int kk) { return v[kk]; });
  long long t3 = __builtin_amdgcn_s_memtime();
  for(int r = 0; r < reps; ++r) acc = chain_neg8(0, 1, n, acc * 1e-3, [&](int kk) { return Lg[rl + kk * ld]; }, [&](int kk) { return v[kk]; });
  long long t4 = __builtin_amdgcn_s_memtime();
  if(lane == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; }
  sink[lane] = acc;
}
int main()
{
  const int n = 43, ld = n | 1, reps = 200;
  std::vector<double> L(n * ld, 0.0);
  for(int j = 0; j < n; ++j) for(int i = j; i < n; ++i) L[i + j * ld] = (i == j) ? 2.0 + 0.01 * i : 0.01 * ((i * 7 + j * 3) % 11 - 5);
  double *dL, *ds; long long * dout;
  hipMalloc(&dL, sizeof(double) * n * ld); hipMalloc(&ds, sizeof(double) * 64); hipMalloc(&dout, 64);
  hipMemcpy(dL, L.data(), sizeof(double) * n * ld, hipMemcpyHostToDevice);
  for(int it = 0; it < 2; ++it)
  {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), sizeof(double) * (n * ld + 64), 0, dL, n, reps, dout, ds);
    hipDeviceSynchronize();
  }
  long long o[4]; hipMemcpy(o, dout, 32, hipMemcpyDeviceToHost);
  printf("n=%d: wave_fwd LDS %lld cycles/call, global %lld; chain(n terms) LDS %lld, global %lld\n", n, o[0] / reps, o[1] / reps, o[2] / reps, o[3] / reps);
  return 0;
}
