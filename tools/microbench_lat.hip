// Dependent-issue latencies on one wave (gfx950): fp64 fma chain, fma + select, v_readlane round trip, LDS broadcast read.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long * out, double * sink, double a, double b)
{
  __shared__ double sm[64];
  const int lane = threadIdx.x;
  sm[lane] = a + lane;
  __syncthreads();
  double x = a + lane * 1e-3, y = b;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for(int r = 0; r < 100; ++r)
  {
#pragma unroll
    for(int j = 0; j < 32; ++j) x = __builtin_fma(x, y, y);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for(int r = 0; r < 100; ++r)
  {
#pragma unroll
    for(int j = 0; j < 32; ++j)
    {
      const double t = __builtin_fma(x, y, y);
      x = (lane > j) ? t : x;
    }
  }
  long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for(int r = 0; r < 100; ++r)
  {
#pragma unroll
    for(int j = 0; j < 32; ++j)
    {
      const int lo = __builtin_amdgcn_readlane(__double2loint(x), j), hi = __builtin_amdgcn_readlane(__double2hiint(x), j);
      x = __builtin_fma(__hiloint2double(hi, lo), y, x);
    }
  }
  long long t3 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for(int r = 0; r < 100; ++r)
  {
#pragma unroll
    for(int j = 0; j < 32; ++j)
    {
      const int idx = ((int)x) & 63; // dependent address
      x = __builtin_fma(sm[idx], y, x);
    }
  }
  long long t4 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for(int r = 0; r < 100; ++r)
  {
#pragma unroll
    for(int j = 0; j < 32; ++j) x = x * y; // mul chain
  }
  long long t5 = __builtin_amdgcn_s_memtime();
  float f = (float)x;
#pragma unroll 1
  for(int r = 0; r < 100; ++r)
  {
#pragma unroll
    for(int j = 0; j < 32; ++j) f = __builtin_fmaf(f, 0.5f, 1.0f);
  }
  long long t6 = __builtin_amdgcn_s_memtime();
  if(lane == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; out[5] = t6 - t5; }
  sink[lane] = x + f;
}
int main()
{
  double * ds; long long * dout;
  hipMalloc(&ds, 512); hipMalloc(&dout, 64);
  for(int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, ds, 1.0, 0.999); hipDeviceSynchronize(); }
  long long o[6]; hipMemcpy(o, dout, 48, hipMemcpyDeviceToHost);
  printf("per step (s_memtime ticks): fma64 chain %.1f, fma64+select %.1f, readlane+fma64 %.1f, dependent LDS read + fma64 %.1f, mul64 chain %.1f, fma32 chain %.1f\n",
         o[0] / 3200.0, o[1] / 3200.0, o[2] / 3200.0, o[3] / 3200.0, o[4] / 3200.0, o[5] / 3200.0);
  return 0;
}
