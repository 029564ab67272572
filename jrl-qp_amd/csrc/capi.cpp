// capi.cpp -- the C ABI of include/jrlqp_hip.h (thin: argument checks + kernel launches, no allocation, no sync)
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/jrlqp_hip.h"
#include "gi_batch.h"
#include "selftest.h"

namespace
{
std::atomic<int> g_last_hip_error{0};
std::atomic<long long *> g_prof{nullptr};

int hip_fail(hipError_t e)
{
  g_last_hip_error.store(static_cast<int>(e));
  return JRLQP_ERR_HIP;
}
} // namespace

extern "C" {

int jrlqp_abi_version(void)
{
  return JRLQP_ABI_VERSION;
}

const char * jrlqp_status_string(int status)
{
  switch(status)
  {
    case JRLQP_OK: return "ok";
    case JRLQP_ERR_INVALID_ARGUMENT: return "invalid argument";
    case JRLQP_ERR_UNSUPPORTED_SIZE: return "problem size not supported by this entry point";
    case JRLQP_ERR_NO_DEVICE: return "no HIP device";
    case JRLQP_ERR_HIP: return "HIP runtime error (see jrlqp_last_hip_error)";
    case JRLQP_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}

int jrlqp_last_hip_error(void)
{
  return g_last_hip_error.load();
}

int jrlqp_device_count(void)
{
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void jrlqp_default_options(jrlqp_options * o)
{
  if(!o) return;
  o->max_iter = 500;   // SolverOptions.h:16
  o->big_bnd = 1e100;  // SolverOptions.h:17
}

long jrlqp_gi_batch_lds_bytes(int n, int m, int nb, int * c_in_lds)
{
  if(n < 1 || n > JRLQP_BATCH_MAX_N || m < 0 || (nb != 0 && nb != n)) return -1;
  int cmode = jrlqp::batch_pick_cmode(n, m, nb);
  if(cmode < 0) return -1;
  if(c_in_lds) *c_in_lds = cmode;
  return jrlqp::batch_lds_bytes(n, m, nb, cmode);
}

int jrlqp_gi_solve_batch(int n,
                         int m,
                         int nb,
                         int B,
                         const double * dG,
                         const double * da,
                         const double * dC,
                         const double * dbl,
                         const double * dbu,
                         const double * dxl,
                         const double * dxu,
                         const jrlqp_options * opts,
                         double * dx,
                         double * du,
                         int * dstatus,
                         int * dterm,
                         int * diters,
                         double * df,
                         double * dL,
                         jrlqp_trace_rec * dtrace,
                         int trace_cap,
                         void * stream)
{
  if(n < 1 || m < 0 || B < 0 || (nb != 0 && nb != n)) return JRLQP_ERR_INVALID_ARGUMENT;
  if(B == 0) return JRLQP_OK;
  if(!dG || !da || !dx || !dterm || !diters || !df) return JRLQP_ERR_INVALID_ARGUMENT;
  if(m + nb > 0 && (!du || !dstatus)) return JRLQP_ERR_INVALID_ARGUMENT;
  if(m > 0 && (!dC || !dbl || !dbu)) return JRLQP_ERR_INVALID_ARGUMENT;
  if(nb > 0 && (!dxl || !dxu)) return JRLQP_ERR_INVALID_ARGUMENT;
  if(dtrace && trace_cap <= 0) return JRLQP_ERR_INVALID_ARGUMENT;
  if(n > JRLQP_BATCH_MAX_N) return JRLQP_ERR_UNSUPPORTED_SIZE;
  const int cmode = jrlqp::batch_pick_cmode(n, m, nb);
  if(cmode < 0) return JRLQP_ERR_UNSUPPORTED_SIZE;

  jrlqp_options o;
  jrlqp_default_options(&o);
  if(opts) o = *opts;

  jrlqp::BatchArgs a;
  a.n = n; a.m = m; a.nb = nb; a.B = B;
  a.ld = n | 1;
  a.max_iter = o.max_iter;
  a.trace_cap = dtrace ? trace_cap : 0;
  a.big_bnd = o.big_bnd;
  a.G = dG; a.a = da; a.C = dC; a.bl = dbl; a.bu = dbu; a.xl = dxl; a.xu = dxu;
  a.x = dx; a.u = du; a.status = dstatus; a.term = dterm; a.iters = diters; a.f = df;
  a.Lout = dL;
  a.trace = dtrace;
  a.prof = g_prof.load();
  hipError_t e = jrlqp::launch_gi_batch(a, cmode, static_cast<hipStream_t>(stream));
  if(e != hipSuccess) return hip_fail(e);
  return JRLQP_OK;
}

int jrlqp_debug_set_profile(long long * dprof)
{
  g_prof.store(dprof);
  return JRLQP_OK;
}

int jrlqp_selftest_arith(int n,
                          const double * da,
                          const double * db,
                          const double * dc,
                          double * ddiv,
                          double * dsqrt,
                          double * dfma,
                          double * dgc,
                          double * dgs,
                          double * dgr,
                          void * stream)
{
  if(n < 0 || !da || !db || !dc || !ddiv || !dsqrt || !dfma || !dgc || !dgs || !dgr) return JRLQP_ERR_INVALID_ARGUMENT;
  hipError_t e = jrlqp::launch_selftest_arith(n, da, db, dc, ddiv, dsqrt, dfma, dgc, dgs, dgr, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? JRLQP_OK : hip_fail(e);
}

int jrlqp_selftest_mfma_f64(const double * dA, const double * dB, const double * dC, double * dD, void * stream)
{
  if(!dA || !dB || !dC || !dD) return JRLQP_ERR_INVALID_ARGUMENT;
  hipError_t e = jrlqp::launch_selftest_mfma_f64(dA, dB, dC, dD, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? JRLQP_OK : hip_fail(e);
}

} // extern "C"
