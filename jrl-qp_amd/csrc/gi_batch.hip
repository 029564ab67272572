// gi_batch.hip -- host-side helpers and dispatch of the batched kernel (gi_batch_kernel.h); the instantiations for n <= 32, the
// generic kernels and the box solver are compiled here, those for n = 40 .. 64 in gi_batch_wide.hip (two translation units: the
// kernel template unrolls over n, and one unit took six minutes to compile)
#include <cstdlib>
#include "gi_batch_kernel.h"

namespace jrlqp
{
hipError_t launch_gi_batch_wide(const BatchArgs & a, long lds, hipStream_t stream);

// instantiated sizes of the register-resident kernel; a cold solve of a smaller n runs embedded in the next one
int batch_pad_size(int n, int warm)
{
  static const int sizes[] = {10, 16, 20, 30, 32, 40, 50, 60, 64};
  for(int s : sizes)
  {
    if(s == n) return s;
    if(s > n) return (n < 8) ? 0 : s; // (tiny problems stay with the generic kernel)
  }
  return 0;
}

long batch_lds_bytes(int n, int m, int nb, int cmode, int warm)
{
  if(cmode == 2) n = batch_pad_size(n, warm);
  const long ld = n | 1;
  long dbl = 2L * n * ld + 4L * n + 2L * (n + 2);
  if(warm) dbl += (long)n * ld + 2L * n;
  if(cmode == 1) dbl += (long)m * ld + 2L * m;
  if(cmode == 2) dbl += 3L * n + 6L * 64;                          // per-wave n+ slots, parked chain state
  if(cmode == 2 && m > WG - 64) dbl += (long)(m - (WG - 64)) * ld; // extra columns beyond the 192 register-resident ones
  long bytes = dbl * 8 + (long)sizeof(Ctl) + 4L * (n + 2) + (long)(m + nb);
  return (bytes + 15) & ~15L;
}

int batch_pick_cmode(int n, int m, int nb, int warm)
{
  // registers: the specialised instantiations below; LDS when C fits beside the factors; else stream
  // registers (192 columns, the rest staged in LDS and scanned by one wave, 64 per pass).  Beyond 64 staged columns only
  // while two workgroups still share a CU: measured, n = 50 with 108 / 208 staged columns (one workgroup per CU) is 22-27 %
  // slower than streaming C, n = 30 / m = 300 and n = 20 / m = 500 are 2.6x / 1.8x faster than the generic kernel
  if(batch_pad_size(n, warm) > 0)
  {
    const long l2 = batch_lds_bytes(n, m, nb, 2, warm);
    if(l2 <= kLdsPerCU && (m <= 256 || 2 * l2 <= kLdsPerCU)) return 2;
  }
  if(batch_lds_bytes(n, m, nb, 1, warm) <= kLdsPerCU) return 1;
  if(batch_lds_bytes(n, m, nb, 0, warm) <= kLdsPerCU) return 0;
  return -1;
}

// experimental::BoxAndSingleConstraintSolver over a batch: the generic kernel (runtime n <= 64, c staged in LDS) with the
// closed-form init_; a.G is not read (G = I), a.a holds x0, a.m == 1, a.nb == a.n
hipError_t launch_gi_batch_box(const BatchArgs & a, hipStream_t stream)
{
  return launch_one<0, 1, false, false, false, true>(a, batch_lds_bytes(a.n, a.m, a.nb, 1, 0), stream);
}

hipError_t launch_gi_batch(const BatchArgs & a, int cmode, hipStream_t stream)
{
  long lds = batch_lds_bytes(a.n, a.m, a.nb, cmode, a.warm);
  // (experiment: JRLQP_EXP_LDS_PAD=<bytes> of unused dynamic LDS per workgroup -- fewer workgroups per CU with the same kernel: the
  // slope of throughput against occupancy)
  static const long pad = [] { const char * e = std::getenv("JRLQP_EXP_LDS_PAD"); return e ? std::atol(e) : 0L; }();
  if(pad > 0 && lds + pad <= kLdsPerCU) lds += pad;
  if(cmode == 2)
  {
    switch(batch_pad_size(a.n, a.warm))
    {
      case 10: return launch_prof<10, 2>(a, lds, stream);
      case 16: return launch_prof<16, 2>(a, lds, stream);
      case 20: return launch_prof<20, 2>(a, lds, stream);
      case 30: return launch_prof<30, 2>(a, lds, stream);
      case 32: return launch_prof<32, 2>(a, lds, stream);
      case 40:
      case 50: // BASELINE configs c3 / c4
      case 60:
      case 64: return launch_gi_batch_wide(a, lds, stream); // (their instantiations live in gi_batch_wide.hip)
      default: return hipErrorInvalidValue;
    }
  }
  if(cmode == 1) return launch_prof<0, 1>(a, lds, stream);
  return launch_prof<0, 0>(a, lds, stream);
}

} // namespace jrlqp
