// gi_batch_wide.hip -- the register-resident instantiations of the batched kernel for n = 40 and 50 (gi_batch_kernel.h);
// n = 60 and 64 are compiled in gi_batch_wide2.hip
#include "gi_batch_kernel.h"

namespace jrlqp
{
hipError_t launch_gi_batch_wide2(const BatchArgs & a, long lds, hipStream_t stream);

hipError_t launch_gi_batch_wide(const BatchArgs & a, long lds, hipStream_t stream)
{
  switch(batch_pad_size(a.n, a.warm))
  {
    case 40: return launch_prof<40, 2>(a, lds, stream);
    case 50: return launch_prof<50, 2>(a, lds, stream); // BASELINE configs c3 / c4
    case 60:
    case 64: return launch_gi_batch_wide2(a, lds, stream);
    default: return hipErrorInvalidValue;
  }
}
} // namespace jrlqp
