// Micro-benchmark: where do the waves of co-resident 256-thread workgroups land?  Every wave records HW_REG_HW_ID
// (gfx9: WAVE_ID[3:0] SIMD_ID[5:4] CU_ID[11:8] SE_ID[15:13]) and the XCC id; the workgroups use 47 KB of LDS and ~168
// VGPRs' worth of occupancy limit (launch_bounds 256,3) like gi_batch_kernel<50,2>, and spin long enough to overlap.
// Also times a dependent fp64 chain on logical wave 0 with and without rotating the role by the hardware wave slot.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off wave_placement.hip -o wave_placement.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 3) void place(unsigned * hw, long long * cyc, int iters, int rotate, double a, double * sink)
{
  extern __shared__ double lds[];
  __shared__ int rot_s;
  const int pw = threadIdx.x >> 6;
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  if(threadIdx.x == 0) rot_s = rotate ? (id & 15) : 0; // hardware wave slot of physical wave 0
  __syncthreads();
  const int lw = (pw - rot_s) & 3; // logical wave
  if((threadIdx.x & 63) == 0)
  {
    hw[(blockIdx.x * 4 + pw) * 2] = id;
    hw[(blockIdx.x * 4 + pw) * 2 + 1] = (xcc & 15) | (lw << 8);
  }
  double x = a + threadIdx.x * 1e-9;
  long long t0 = clock64();
  if(lw == 0)
  {
    for(int i = 0; i < iters; ++i)
    {
      double t = a / x;
      double u = __builtin_sqrt(__builtin_fma(t, t, 1.0));
      x = x * u; // the Givens coefficient chain step
    }
  }
  else
  {
    // light periodic work like the scanning waves
    for(int i = 0; i < iters / 16; ++i)
    {
      x = __builtin_fma(x, 0.999, lds[(threadIdx.x + i) & 1023]);
      __builtin_amdgcn_s_sleep(20);
    }
  }
  long long t1 = clock64();
  sink[blockIdx.x * 256 + threadIdx.x] = x;
  if((threadIdx.x & 63) == 0 && lw == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
  const int blocks = 768 * 4, iters = 4000;
  unsigned * hw; long long * cyc; double * sink;
  hipMalloc(&hw, sizeof(unsigned) * blocks * 8); hipMalloc(&cyc, sizeof(long long) * blocks); hipMalloc(&sink, sizeof(double) * blocks * 256);
  hipFuncSetAttribute((const void *)place, hipFuncAttributeMaxDynamicSharedMemorySize, 47040);
  for(int rotate = 0; rotate < 2; ++rotate)
  {
    for(int rep = 0; rep < 2; ++rep)
    {
      place<<<blocks, 256, 47040>>>(hw, cyc, iters, rotate, 1.25, sink);
      hipDeviceSynchronize();
    }
    std::vector<unsigned> h(blocks * 8); std::vector<long long> c(blocks);
    hipMemcpy(h.data(), hw, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), cyc, sizeof(long long) * c.size(), hipMemcpyDeviceToHost);
    double s = 0; for(auto v : c) s += v;
    printf("rotate=%d: %.1f cycles per chain step (mean over %d workgroups)\n", rotate, s / blocks / iters, blocks);
    std::map<int, int> simd_of_lw0, patt;
    for(int b = 0; b < blocks; ++b)
    {
      int key = 0;
      for(int w = 0; w < 4; ++w)
      {
        unsigned id = h[(b * 4 + w) * 2], aux = h[(b * 4 + w) * 2 + 1];
        int simd = (id >> 4) & 3, lw = (aux >> 8) & 3;
        key = key * 4 + simd;
        if(lw == 0) simd_of_lw0[simd]++;
      }
      patt[key]++;
    }
    printf("  SIMD of logical wave 0: "); for(auto & kv : simd_of_lw0) printf("simd%d:%d ", kv.first, kv.second); printf("\n");
    printf("  physical wave -> SIMD patterns (w0 w1 w2 w3 : count): ");
    for(auto & kv : patt) printf("%d%d%d%d:%d ", (kv.first >> 6) & 3, (kv.first >> 4) & 3, (kv.first >> 2) & 3, kv.first & 3, kv.second);
    printf("\n");
    if(rotate == 0)
      for(int b = 0; b < 6; ++b)
      {
        printf("  wg %d:", b);
        for(int w = 0; w < 4; ++w)
        {
          unsigned id = h[(b * 4 + w) * 2], aux = h[(b * 4 + w) * 2 + 1];
          printf("  [xcc %u se %u cu %u simd %u slot %u]", aux & 15, (id >> 13) & 7, (id >> 8) & 15, (id >> 4) & 3, id & 15);
        }
        printf("\n");
      }
  }
  return 0;
}
