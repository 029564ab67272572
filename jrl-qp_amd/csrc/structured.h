// structured.h -- launch interface of the structured Cholesky family (internal to libjrlqp_hip.so)
#pragma once
#include <hip/hip_runtime.h>

namespace jrlqp
{
struct BlockDesc
{
  int nblk;
  int up;          // block arrow pointing up (get<true> of blockArrowLLT.cpp:36-46)
  long mstride;    // doubles between the right-hand sides of consecutive problems of a batch
  const int * bs;  // [nblk]   block sizes (device)
  const int * ldd; // [nblk]   leading dimensions of the diagonal blocks (device)
  const int * lds; // [nblk-1] leading dimensions of the side / sub-diagonal blocks (device)
  double ** diag;  // [nprob*nblk]     device pointers
  double ** side;  // [nprob*(nblk-1)] device pointers
  int * fail;      // [nprob] set to 1 when a factorisation fails
  const int * h_bs; // host copy of bs (launch helpers only)
  int xs_cap;      // doubles of LDS the factorisation kernels may use for a panel of a side block (set by the launch helpers)
  const int * off = nullptr; // [nblk + 1] first unknown of LOGICAL block i (device; the kernels summed block sizes from memory before)
  int ntot = 0;              // number of unknowns
  int ring_ok = 0; // arrow pointing down, every block of at most 32 rows, an even number of rows in the head, even leading dimensions and
                   // 16-byte aligned side blocks (jrlqp_blocks_bind checks): the side blocks can be streamed into LDS 16 bytes per lane
};

hipError_t launch_block_arrow_llt(const BlockDesc & d, int nprob, int maxb, hipStream_t st);
hipError_t launch_block_arrow_solve(const BlockDesc & d, int nprob, bool transpose, double * M, int ldm, int nrhs, int start,
                                    int end, double * tmp, hipStream_t st);
hipError_t launch_tri_llt(const BlockDesc & d, int nprob, int maxb, hipStream_t st);
hipError_t launch_tri_solve(const BlockDesc & d, int nprob, bool transpose, double * M, int ldm, int nrhs, int start_or_end,
                            hipStream_t st);
} // namespace jrlqp
