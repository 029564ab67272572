// gi_batch.h -- launch interface of the batched GI kernel (internal to libjrlqp_hip.so)
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/jrlqp_hip.h"

namespace jrlqp
{
struct BatchArgs
{
  int n, m, nb, B;
  int ld; // n | 1: leading dimension (in doubles) of every n-row matrix staged in LDS
  int max_iter;
  int trace_cap;
  double big_bnd;
  const double *G, *a, *C, *bl, *bu, *xl, *xu;
  double *x, *u;
  int *status, *term, *iters;
  double * f;
  double * Lout;
  jrlqp_trace_rec * trace;
  long long * prof; // diagnostic only (jrlqp_debug_set_profile): prof_stride >= 32 int64 per QP, or nullptr
  int prof_stride;
  int warm;           // 1: warm-started solve (experimental::GoldfarbIdnaniSolver), as_in = activation-status guess
  const int * as_in;  // B x (m+nb) guesses, or nullptr ("no guess"); may alias `status`
  int box;            // 1: experimental::BoxAndSingleConstraintSolver (G = I implicit, a = x0, m == 1, nb == n, bu = +inf); n > 64: gi_batch_big.hip
  int tune;           // experiment bits (JRLQP_TUNE in the environment, capi.cpp env_tune; 0 in production)
  int two_per_cu = 0; // gi_batch_big, n = 79 .. 95: the packed R in the workspace so that two workgroups share a CU (batches of more
                      // problems than CUs, a workspace at hand: capi.cpp decides, batch_big_lds_bytes and the launcher follow)
};

constexpr long kLdsPerCU = 163840; // 160 KiB (MI355X_MICROARCH.md)

// cmode: where the constraint matrix lives during the solve: 0 = streamed from global memory, 1 = staged in LDS,
// 2 = one column per thread in registers (specialised compile-time-n instantiations)
int batch_pad_size(int n, int warm); // instantiated size a problem of n variables runs in (0: none, generic kernel)
long batch_lds_bytes(int n, int m, int nb, int cmode, int warm = 0);
int batch_pick_cmode(int n, int m, int nb, int warm = 0); // -1 if unsupported
hipError_t launch_gi_batch_box(const BatchArgs & a, hipStream_t stream);
// 64 < n <= JRLQP_BATCH_MAX_N (gi_batch_big.hip): persistent grid of nwg workgroups; ws: nwg * batch_big_ws_doubles_per_wg(...)
// doubles of global workspace: R when it does not fit in LDS beside J (*r_in_lds == 0), B = L^-1 N of a warm start
long batch_big_lds_bytes(int n, int m, int nb, int * r_in_lds, int two_per_cu = 1);
bool batch_big_is_wide(int n); // J in the workspace (always beyond JRLQP_BATCH_LDS_N unknowns)
long batch_big_ws_doubles_per_wg(int n, int r_in_lds, int warm);
hipError_t launch_gi_batch_big(const BatchArgs & a, int nwg, double * ws, hipStream_t stream);
hipError_t launch_gi_batch(const BatchArgs & a, int cmode, hipStream_t stream);
} // namespace jrlqp
