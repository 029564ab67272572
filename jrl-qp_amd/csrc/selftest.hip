// selftest.hip -- device-side probes of the arithmetic contract: correctly rounded fp64 division,
// square root and fma, the Givens generator, and the accumulation order of v_mfma_f64_16x16x4_f64.  The GPU
// tests compare these with the host's IEEE results bit for bit; they exist because the batched solver's
// bit-exact parity with the CPU oracle rests on them.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "selftest.h"

namespace jrlqp
{
namespace
{
__global__ void arith_kernel(int n, const double * a, const double * b, const double * c, double * odiv, double * osqrt,
                             double * ofma, double * gc, double * gs, double * gr)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  odiv[i] = a[i] / b[i];
  osqrt[i] = __builtin_sqrt(fabs(a[i]));
  ofma[i] = __builtin_fma(a[i], b[i], c[i]);
  double cc, ss, rr;
  make_givens(a[i], b[i], cc, ss, rr);
  {
    // the lean generator (device_common.h) must be make_givens bit for bit: a difference poisons the outputs the test compares
    double c2, s2, r2;
    make_givens_lean(a[i], b[i], c2, s2, r2);
    const bool same = __double_as_longlong(c2) == __double_as_longlong(cc) && __double_as_longlong(s2) == __double_as_longlong(ss)
                      && __double_as_longlong(r2) == __double_as_longlong(rr);
    if(!same && !(cc != cc || ss != ss || rr != rr)) cc = ss = rr = __builtin_nan("");
  }
  gc[i] = cc; gs[i] = ss; gr[i] = rr;
}

typedef double double4_t __attribute__((ext_vector_type(4)));

// one wave: D(16x16) = A(16x4) * B(4x16) + C, all row-major in memory.
// fragment layout (cdna_hip_programming.md §3): a = A[l&15][l>>4], b = B[l>>4][l&15]; c/d reg r: row (l>>4)+4r, col l&15
__global__ void mfma_f64_kernel(const double * A, const double * B, const double * C, double * D)
{
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  double4_t acc;
  for(int r = 0; r < 4; ++r) acc[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for(int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
} // namespace

hipError_t launch_selftest_arith(int n, const double * a, const double * b, const double * c, double * odiv, double * osqrt,
                                 double * ofma, double * gc, double * gs, double * gr, hipStream_t stream)
{
  if(n <= 0) return hipSuccess;
  hipLaunchKernelGGL(arith_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, a, b, c, odiv, osqrt, ofma, gc, gs, gr);
  return hipGetLastError();
}

hipError_t launch_selftest_mfma_f64(const double * A, const double * B, const double * C, double * D, hipStream_t stream)
{
  hipLaunchKernelGGL(mfma_f64_kernel, dim3(1), dim3(64), 0, stream, A, B, C, D);
  return hipGetLastError();
}
} // namespace jrlqp
