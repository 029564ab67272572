// gi_batch_wide2.hip -- the register-resident instantiations of the batched kernel for n = 60 and 64 (gi_batch_kernel.h)
#include "gi_batch_kernel.h"

namespace jrlqp
{
hipError_t launch_gi_batch_wide2(const BatchArgs & a, long lds, hipStream_t stream)
{
  switch(batch_pad_size(a.n, a.warm))
  {
    case 60: return launch_prof<60, 2>(a, lds, stream);
    case 64: return launch_prof<64, 2>(a, lds, stream);
    default: return hipErrorInvalidValue;
  }
}
} // namespace jrlqp
