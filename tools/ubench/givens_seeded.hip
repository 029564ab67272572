// Micro-benchmark + check of the seeded Givens radius chain (device_common.h givens_seed / givens_seeded_step) against
// the plain recurrence givens_r_only: steps whose radius differs (expected: none in millions), cycles per step of both.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off givens_seeded.hip -o givens_seeded.bin
#include "../../jrl-qp_amd/csrc/device_common.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace jrlqp;

// one wave per vector of n <= 64 entries (lane i holds d_i), as in the batched kernel
__global__ __launch_bounds__(64) void chain64(const double * dall, int n, int reps, long long * cyc, int * mism, double * sfx_out)
{
  const int lane = threadIdx.x;
  const double * dv = dall + (size_t)blockIdx.x * n;
  int bad = 0;
  long long ce = 0, cs = 0;
  for(int rep = 0; rep < reps; ++rep)
  {
    const double d = (lane < n) ? dv[lane] * (1.0 + rep * 0x1p-30) : 0.0;
    // exact
    long long t0 = clock64();
    const double d_last = bcast_lane(d, n - 1);
    double r = opaque_v(d_last), kq = 0.0;
    for(int i = n - 2; i >= 0; --i)
    {
      const double p = opaque_v(bcast_lane(d, i));
      kq = (lane == i) ? r : kq;
      r = givens_r_only(p, r);
    }
    long long t1 = clock64();
    // seeded
    const double sq = (lane < n) ? d * d : 0.0;
    const double sincl = wave_suffix_sum_approx(sq, lane);
    const double snext = dpp_shift0_d<0x130>(sincl); // lane i <- lane i+1 (wave_shl:1), last lane 0
    if(rep == 0 && sfx_out) sfx_out[(size_t)blockIdx.x * 64 + lane] = snext;
    const GivensSeed g = givens_seed(d, snext);
    double rc = d; // lane n-1: d_{n-1}; the others are overwritten as the chain reaches them
    for(int k = 0; k < n - 1; ++k) rc = givens_seeded_step(dpp_shift0_d<0x130>(rc), g);
    double kq2 = dpp_shift0_d<0x130>(rc);
    kq2 = (lane == n - 2) ? d_last : kq2;
    const double r2 = bcast_lane(rc, 0);
    long long t2 = clock64();
    ce += t1 - t0; cs += t2 - t1;
    bad += (lane <= n - 2 && !(kq == kq2)) ? 1 : 0;
    bad += (lane == 0 && !(r == r2)) ? 1 : 0;
  }
  atomicAdd(mism, bad);
  if(lane == 0) { cyc[blockIdx.x * 2] = ce; cyc[blockIdx.x * 2 + 1] = cs; }
}

int main()
{
  for(int n : {50, 64, 20})
  for(int mode = 0; mode < 3; ++mode)
  {
    const int nv = 2048, reps = 50;
    std::vector<double> h((size_t)nv * n);
    srand48(7 + mode);
    for(int v = 0; v < nv; ++v)
      for(int i = 0; i < n; ++i)
      {
        double x = drand48() * 2 - 1;
        if(mode == 1) x *= pow(10.0, 6 * (drand48() * 2 - 1));          // wide dynamic range
        if(mode == 2 && (i > n - 1 - (v % n) || drand48() < 0.2)) x = 0; // zero tails and scattered zeros
        h[(size_t)v * n + i] = x;
      }
    double *d, *sfx; long long * cyc; int * mism;
    hipMalloc(&d, h.size() * 8); hipMalloc(&cyc, nv * 16); hipMalloc(&mism, 4); hipMalloc(&sfx, (size_t)nv * 64 * 8);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemset(mism, 0, 4);
    chain64<<<nv, 64>>>(d, n, reps, cyc, mism, sfx);
    hipDeviceSynchronize();
    int m = -1; hipMemcpy(&m, mism, 4, hipMemcpyDeviceToHost);
    std::vector<long long> c(nv * 2); hipMemcpy(c.data(), cyc, nv * 16, hipMemcpyDeviceToHost);
    std::vector<double> sf((size_t)nv * 64); hipMemcpy(sf.data(), sfx, sf.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for(int v = 0; v < nv; ++v) { double s = 0; for(int i = n - 1; i >= 0; --i) { double got = sf[(size_t)v * 64 + i]; if(s > 0) worst = fmax(worst, fabs(got - s) / s); else if(got != 0) worst = 1e300; s += h[(size_t)v * n + i] * h[(size_t)v * n + i]; } }
    double se = 0, ss = 0; for(int v = 0; v < nv; ++v) { se += c[2 * v]; ss += c[2 * v + 1]; }
    printf("n=%d mode=%d: steps=%lld radius mismatches=%d  suffix-sum rel err max=%.2e  cycles/step exact=%.1f seeded(incl. seeds)=%.1f\n", n, mode,
           (long long)nv * reps * (n - 1), m, worst, se / nv / reps / (n - 1), ss / nv / reps / (n - 1));
    hipFree(d); hipFree(cyc); hipFree(mism); hipFree(sfx);
  }
  return 0;
}
