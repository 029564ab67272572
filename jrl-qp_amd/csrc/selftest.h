#pragma once
#include <hip/hip_runtime.h>
namespace jrlqp
{
hipError_t launch_selftest_arith(int n, const double * a, const double * b, const double * c, double * odiv, double * osqrt,
                                 double * ofma, double * gc, double * gs, double * gr, hipStream_t stream);
hipError_t launch_selftest_mfma_f64(const double * A, const double * B, const double * C, double * D, hipStream_t stream);
} // namespace jrlqp
