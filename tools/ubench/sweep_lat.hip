// microbenchmark: the descending sweep of an addition out of LDS (gi_batch_big.hip add_constraint), cycles per sweep of T columns, n = 100 rows
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off sweep_lat.hip -o sweep_lat.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define FMA(a, b, c) __builtin_fma(a, b, c)
template<int MODE>
__global__ __launch_bounds__(256) void k(int n, int q, int reps, long long * out, double * sink)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int ld = n | 1, tid = threadIdx.x;
  double * J = sm, * ccs = sm + (long)n * ld + 1, * sss = ccs + n, * cs2 = sss + n + (n & 1);
  for(int e = tid; e < n * ld; e += 256) J[e] = 1.0 / (1 + (e % 97));
  for(int e = tid; e < n; e += 256) { ccs[e] = 0.8; sss[e] = 0.6; cs2[2 * e] = 0.8; cs2[2 * e + 1] = 0.6; }
  __syncthreads();
  long long t0 = clock64();
  for(int r = 0; r < reps; ++r)
  {
    if(MODE == 0 && tid < n) // as in round 3: four columns per round trip, c and s from two arrays
    {
      double carry = J[tid + (long)(n - 1) * ld];
      int i = n - 2;
      for(; i - 3 >= q; i -= 4)
      {
        double xv[4], cv[4], sv[4];
#pragma unroll
        for(int u = 0; u < 4; ++u) { xv[u] = J[tid + (long)(i - u) * ld]; cv[u] = ccs[i - u]; sv[u] = sss[i - u]; }
#pragma unroll
        for(int u = 0; u < 4; ++u) { J[tid + (long)(i - u + 1) * ld] = FMA(sv[u], xv[u], cv[u] * carry); carry = FMA(-sv[u], carry, cv[u] * xv[u]); }
      }
      for(; i >= q; --i) { const double xv = J[tid + (long)i * ld]; const double c = ccs[i], s = sss[i]; J[tid + (long)(i + 1) * ld] = FMA(s, xv, c * carry); carry = FMA(-s, carry, c * xv); }
      J[tid + (long)q * ld] = carry;
    }
    if(MODE == 1 && tid < n) // (c, s) interleaved: one 16-byte read per column; eight columns per round trip
    {
      double carry = J[tid + (long)(n - 1) * ld];
      int i = n - 2;
      for(; i - 7 >= q; i -= 8)
      {
        double xv[8]; double2 cs[8];
#pragma unroll
        for(int u = 0; u < 8; ++u) { xv[u] = J[tid + (long)(i - u) * ld]; cs[u] = *reinterpret_cast<const double2 *>(cs2 + 2 * (i - u)); }
#pragma unroll
        for(int u = 0; u < 8; ++u) { J[tid + (long)(i - u + 1) * ld] = FMA(cs[u].y, xv[u], cs[u].x * carry); carry = FMA(-cs[u].y, carry, cs[u].x * xv[u]); }
      }
      for(; i >= q; --i) { const double xv = J[tid + (long)i * ld]; const double c = cs2[2 * i], s = cs2[2 * i + 1]; J[tid + (long)(i + 1) * ld] = FMA(s, xv, c * carry); carry = FMA(-s, carry, c * xv); }
      J[tid + (long)q * ld] = carry;
    }
    if(MODE == 2 && tid < (n + 1) / 2) // two rows per lane (rows tid and tid + half), (c, s) interleaved, four columns per round trip
    {
      const int half = (n + 1) / 2, r1 = min(tid + half, n - 1);
      double ca = J[tid + (long)(n - 1) * ld], cb = J[r1 + (long)(n - 1) * ld];
      int i = n - 2;
      for(; i - 3 >= q; i -= 4)
      {
        double xa[4], xb[4]; double2 cs[4];
#pragma unroll
        for(int u = 0; u < 4; ++u) { xa[u] = J[tid + (long)(i - u) * ld]; xb[u] = J[r1 + (long)(i - u) * ld]; cs[u] = *reinterpret_cast<const double2 *>(cs2 + 2 * (i - u)); }
#pragma unroll
        for(int u = 0; u < 4; ++u)
        {
          J[tid + (long)(i - u + 1) * ld] = FMA(cs[u].y, xa[u], cs[u].x * ca); ca = FMA(-cs[u].y, ca, cs[u].x * xa[u]);
          J[r1 + (long)(i - u + 1) * ld] = FMA(cs[u].y, xb[u], cs[u].x * cb); cb = FMA(-cs[u].y, cb, cs[u].x * xb[u]);
        }
      }
      for(; i >= q; --i)
      {
        const double xa = J[tid + (long)i * ld], xb = J[r1 + (long)i * ld]; const double c = cs2[2 * i], s = cs2[2 * i + 1];
        J[tid + (long)(i + 1) * ld] = FMA(s, xa, c * ca); ca = FMA(-s, ca, c * xa);
        J[r1 + (long)(i + 1) * ld] = FMA(s, xb, c * cb); cb = FMA(-s, cb, c * xb);
      }
      J[tid + (long)q * ld] = ca; J[r1 + (long)q * ld] = cb;
    }
    if(MODE == 3 && tid < n) // as MODE 1 with four columns per round trip
    {
      double carry = J[tid + (long)(n - 1) * ld];
      int i = n - 2;
      for(; i - 3 >= q; i -= 4)
      {
        double xv[4]; double2 cs[4];
#pragma unroll
        for(int u = 0; u < 4; ++u) { xv[u] = J[tid + (long)(i - u) * ld]; cs[u] = *reinterpret_cast<const double2 *>(cs2 + 2 * (i - u)); }
#pragma unroll
        for(int u = 0; u < 4; ++u) { J[tid + (long)(i - u + 1) * ld] = FMA(cs[u].y, xv[u], cs[u].x * carry); carry = FMA(-cs[u].y, carry, cs[u].x * xv[u]); }
      }
      for(; i >= q; --i) { const double xv = J[tid + (long)i * ld]; const double c = cs2[2 * i], s = cs2[2 * i + 1]; J[tid + (long)(i + 1) * ld] = FMA(s, xv, c * carry); carry = FMA(-s, carry, c * xv); }
      J[tid + (long)q * ld] = carry;
    }
    if(MODE == 4 && tid < n) // round 3's form, the NEXT group's operands requested before the current group is rotated
    {
      double carry = J[tid + (long)(n - 1) * ld];
      int i = n - 2;
      if(i - 3 >= q)
      {
        double xv[4], cv[4], sv[4], xn[4], cn[4], sn[4];
#pragma unroll
        for(int u = 0; u < 4; ++u) { xn[u] = J[tid + (long)(i - u) * ld]; cn[u] = ccs[i - u]; sn[u] = sss[i - u]; }
        for(; i - 7 >= q; i -= 4)
        {
#pragma unroll
          for(int u = 0; u < 4; ++u) { xv[u] = xn[u]; cv[u] = cn[u]; sv[u] = sn[u]; }
#pragma unroll
          for(int u = 0; u < 4; ++u) { xn[u] = J[tid + (long)(i - 4 - u) * ld]; cn[u] = ccs[i - 4 - u]; sn[u] = sss[i - 4 - u]; }
#pragma unroll
          for(int u = 0; u < 4; ++u) { J[tid + (long)(i - u + 1) * ld] = FMA(sv[u], xv[u], cv[u] * carry); carry = FMA(-sv[u], carry, cv[u] * xv[u]); }
        }
#pragma unroll
        for(int u = 0; u < 4; ++u) { J[tid + (long)(i - u + 1) * ld] = FMA(sn[u], xn[u], cn[u] * carry); carry = FMA(-sn[u], carry, cn[u] * xn[u]); }
        i -= 4;
      }
      for(; i >= q; --i) { const double xv = J[tid + (long)i * ld]; const double c = ccs[i], s = sss[i]; J[tid + (long)(i + 1) * ld] = FMA(s, xv, c * carry); carry = FMA(-s, carry, c * xv); }
      J[tid + (long)q * ld] = carry;
    }
    if(MODE == 5 && tid < n) // eight columns per round trip, c and s from two arrays
    {
      double carry = J[tid + (long)(n - 1) * ld];
      int i = n - 2;
      for(; i - 7 >= q; i -= 8)
      {
        double xv[8], cv[8], sv[8];
#pragma unroll
        for(int u = 0; u < 8; ++u) { xv[u] = J[tid + (long)(i - u) * ld]; cv[u] = ccs[i - u]; sv[u] = sss[i - u]; }
#pragma unroll
        for(int u = 0; u < 8; ++u) { J[tid + (long)(i - u + 1) * ld] = FMA(sv[u], xv[u], cv[u] * carry); carry = FMA(-sv[u], carry, cv[u] * xv[u]); }
      }
      for(; i >= q; --i) { const double xv = J[tid + (long)i * ld]; const double c = ccs[i], s = sss[i]; J[tid + (long)(i + 1) * ld] = FMA(s, xv, c * carry); carry = FMA(-s, carry, c * xv); }
      J[tid + (long)q * ld] = carry;
    }
    __syncthreads();
  }
  long long t1 = clock64();
  if(tid == 0) out[0] = (t1 - t0) / reps;
  sink[tid] = J[tid];
}
int main()
{
  long long * out; double * sink;
  (void)hipMalloc(&out, 8); (void)hipMalloc(&sink, 8 * 256);
  const int n = 100, reps = 1000;
  size_t lds = sizeof(double) * ((size_t)n * (n | 1) + 6 * n + 8);
  const char * names[] = {"round 3 (4 cols, c / s arrays)", "(c,s) pairs, 8 cols", "two rows per lane, pairs, 4 cols", "(c,s) pairs, 4 cols", "round 3 + next group requested ahead", "8 cols, c / s arrays"};
  for(int q : {20, 60})
    for(int mode = 0; mode < 6; ++mode)
    {
      long long h = 0;
      for(int rep = 0; rep < 2; ++rep)
      {
        switch(mode)
        {
          case 0: hipLaunchKernelGGL(k<0>, 1, 256, lds, 0, n, q, reps, out, sink); break;
          case 1: hipLaunchKernelGGL(k<1>, 1, 256, lds, 0, n, q, reps, out, sink); break;
          case 2: hipLaunchKernelGGL(k<2>, 1, 256, lds, 0, n, q, reps, out, sink); break;
          case 3: hipLaunchKernelGGL(k<3>, 1, 256, lds, 0, n, q, reps, out, sink); break;
          case 4: hipLaunchKernelGGL(k<4>, 1, 256, lds, 0, n, q, reps, out, sink); break;
          case 5: hipLaunchKernelGGL(k<5>, 1, 256, lds, 0, n, q, reps, out, sink); break;
        }
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
      }
      printf("q %2d (%2d columns)  %-36s %6lld cycles per sweep = %5.1f per column\n", q, n - 1 - q, names[mode], h, (double)h / (n - 1 - q));
    }
  return 0;
}
