// Micro-benchmark: dependent-chain latency of fp64 fma / div / sqrt on gfx950 for 1..8 waves per SIMD and for a
// partially masked wave.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off fp64_latency.hip -o fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template<int OP>
__global__ void chain(double* out, long long* cyc, int iters, double a, double b, int active_lanes)
{
  double x = a + threadIdx.x * 1e-9;
  long long t0 = 0, t1 = 0;
  if ((threadIdx.x & 63) < active_lanes) {
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
      if (OP == 0) x = __builtin_fma(x, b, a);
      if (OP == 1) x = a / x + b;            // div + add
      if (OP == 2) x = __builtin_sqrt(x) + a; // sqrt + add
      if (OP == 3) { double t = a / x; double u = __builtin_sqrt(__builtin_fma(t, t, 1.0)); x = x * u; } // givens chain step
    }
    t1 = clock64();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template<int OP> void run(const char* name, int threads, int active)
{
  int blocks = 256, iters = 2000;
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, sizeof(long long) * blocks * 16);
  chain<OP><<<blocks, threads>>>(out, cyc, iters, 1.25, 0.75, active);
  hipDeviceSynchronize();
  chain<OP><<<blocks, threads>>>(out, cyc, iters, 1.25, 0.75, active);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("%-8s waves/SIMD=%d active_lanes=%2d : %.1f cycles per chain step\n", name, threads / 256, active, s / h.size() / iters);
  hipFree(out); hipFree(cyc);
}
int main()
{
  for (int threads : {256, 512, 1024}) {
    run<0>("fma", threads, 64); run<1>("div+add", threads, 64); run<2>("sqrt+add", threads, 64); run<3>("givens", threads, 64);
  }
  run<0>("fma", 256, 1); run<1>("div+add", 256, 1); run<3>("givens", 256, 1); run<3>("givens", 256, 16); run<3>("givens", 256, 32);
  return 0;
}
