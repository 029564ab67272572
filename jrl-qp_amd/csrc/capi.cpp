// capi.cpp -- the C ABI of include/jrlqp_hip.h (thin: argument checks + kernel launches, no allocation, no sync)
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/jrlqp_hip.h"
#include "blockgi.h"
#include "gi_batch.h"
#include "selftest.h"
#include "structured.h"

#include <cstdlib>
#include <new>
#include <vector>

namespace
{
std::atomic<int> g_last_hip_error{0};
// diagnostic phase-profile buffers, one per kernel family (jrlqp_debug_set_profile)
struct ProfBuf
{
  long long * p = nullptr;
  int stride = 0, nprob = 0;
};
ProfBuf g_prof[2];

// experiment knob (never part of the ABI): JRLQP_TUNE=<bits> in the environment, read once; travels in BatchArgs::tune
int env_tune()
{
  static const int v = [] {
    const char * e = std::getenv("JRLQP_TUNE");
    return e ? std::atoi(e) : 0;
  }();
  return v;
}

int hip_fail(hipError_t e)
{
  g_last_hip_error.store(static_cast<int>(e));
  return JRLQP_ERR_HIP;
}
} // namespace

// Pinned staging area for descriptor uploads: a bind fills it and enqueues asynchronous H2D copies; the event recorded
// behind them is waited for before the NEXT bind overwrites the area (binds are set-up calls, solves never wait).
struct PinnedStage
{
  char * p = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool pending = false;
  hipError_t init(size_t bytes)
  {
    cap = bytes ? bytes : 8;
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), cap, hipHostMallocDefault);
    if(e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    return e;
  }
  hipError_t reuse()
  {
    if(!pending) return hipSuccess;
    pending = false;
    return hipEventSynchronize(ev);
  }
  hipError_t submitted(hipStream_t st)
  {
    pending = true;
    return hipEventRecord(ev, st);
  }
  void release()
  {
    if(ev) (void)hipEventDestroy(ev);
    if(p) (void)hipHostFree(p);
    ev = nullptr;
    p = nullptr;
  }
};

// plan object of the structured family: owns the device-side block descriptors, the failure flags and the scratch
// used by the up-arrow row permutation, so that factorisations and solves allocate nothing
struct jrlqp_blocks
{
  int kind = 0, nblk = 0, nprob = 0, max_nrhs = 0, maxb = 0, ntot = 0;
  std::vector<int> h_bs;
  PinnedStage stage; // [ldd: nblk int][lds: ns int][diag: nprob*nblk ptr][side: nprob*ns ptr]
  int *d_bs = nullptr, *d_ldd = nullptr, *d_lds = nullptr, *d_fail = nullptr;
  double **d_diag = nullptr, **d_side = nullptr;
  double * d_tmp = nullptr;
  int * d_off = nullptr; // logical prefix offsets (nblk + 1)
  bool bound = false;
  int ring_ok = 0; // (jrlqp_blocks_bind)
};

// plan object of the structured GI solver: a jrlqp_blocks plan (factorisation of G) + the block-diagonal C description
// + the per-problem HBM workspace of the device-resident loop
struct jrlqp_blockgi
{
  jrlqp_blocks * blocks = nullptr;
  int kind = 0, nblk = 0, n = 0, m = 0, nb = 0, ncb = 0, nprob = 0, max_iter = 0;
  int n_pad = 0, qcap = 0, row_len = 0;
  long rlen = 0;
  std::vector<int> h_cvar;
  PinnedStage stage; // [cld: ncb int][cblk: nprob*ncb ptr]
  int *d_boff = nullptr, *d_cvar = nullptr, *d_ccstr = nullptr, *d_cumv = nullptr, *d_cumc = nullptr, *d_toblk = nullptr, *d_cld = nullptr;
  const double ** d_cblk = nullptr;
  double *d_pool = nullptr, *d_Rp = nullptr, *d_xu = nullptr;
  int xu_global = 0;
  jrlqp::SeqRec * d_desc = nullptr;
  int * d_active = nullptr;
  size_t ws_bytes = 0;
  bool bound = false;
  // contract C9 (blockgi_c9.hip): form 1
  int form = 0, ngrp = 0;
  int *d_greach = nullptr, *d_tk = nullptr, *d_kap_prow = nullptr, *d_kap_row = nullptr, *d_off_perm = nullptr, *d_kap_phys = nullptr, *d_goff = nullptr;
  double *d_Bc = nullptr, *d_Sm = nullptr, *d_St = nullptr;
  int *d_sgrp = nullptr, *d_freel = nullptr;
  // transposed copies of the factor's blocks (both forms)
  double * d_Tw = nullptr;
  int *d_tw_doff = nullptr, *d_tw_soff = nullptr, *d_tw_sc = nullptr;
  long tw_len = 0;
};

#define HIP_TRY(x)                      \
  do                                    \
  {                                     \
    hipError_t e_ = (x);                \
    if(e_ != hipSuccess) return hip_fail(e_); \
  } while(0)

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
  if(n > JRLQP_BATCH_WAVE_N && n <= JRLQP_BATCH_MAX_N && m >= 0 && (nb == 0 || nb == n))
  {
    if(c_in_lds) *c_in_lds = 0;
    return jrlqp::batch_big_lds_bytes(n, m, nb, nullptr);
  }
  if(n < 1 || n > JRLQP_BATCH_WAVE_N || m < 0 || (nb != 0 && nb != n)) return -1;
  int cmode = jrlqp::batch_pick_cmode(n, m, nb);
  if(cmode < 0) return -1;
  if(c_in_lds) *c_in_lds = cmode;
  return jrlqp::batch_lds_bytes(n, m, nb, cmode);
}

// grid of the persistent kernel of gi_batch_big.hip on the current device (0: no device)
static int device_cus()
{
  int dev = 0, ncu = 0;
  if(hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1)
    return 0;
  return ncu;
}

static int big_grid(int n, int m, int nb, int B, int two_per_cu = 1)
{
  const int ncu = device_cus();
  if(ncu < 1) return 0;
  const long ldsb = jrlqp::batch_big_lds_bytes(n, m, nb, nullptr, two_per_cu);
  const int per_cu = ldsb > 0 ? (int)(jrlqp::kLdsPerCU / ldsb) : 1;
  const int cap = jrlqp::batch_big_is_wide(n) ? 3 : 2; // (J in the workspace: more workgroups per CU hide its L2 latency; gi_batch_big.hip launch bounds)
  const int slots = ncu * (per_cu < 1 ? 1 : (per_cu > cap ? cap : per_cu));
  return B < slots ? B : slots;
}

long long jrlqp_gi_batch_workspace_bytes(int n, int m, int nb, int B)
{
  if(n < 1 || n > JRLQP_BATCH_MAX_N || m < 0 || B < 0 || (nb != 0 && nb != n)) return -1;
  if(n <= JRLQP_BATCH_WAVE_N) return jrlqp::batch_pick_cmode(n, m, nb) < 0 ? -1 : 0;
  int r_in_lds = 0;
  if(jrlqp::batch_big_lds_bytes(n, m, nb, &r_in_lds) < 0) return -1;
  if(B == 0) return 0;
  const int ncu = device_cus();
  if(ncu < 1) return -1;
  const int two = B > ncu ? 1 : 0; // (n = 79 .. 95: two workgroups per CU only pay when the batch has more problems than CUs)
  if(jrlqp::batch_big_lds_bytes(n, m, nb, &r_in_lds, two) < 0) return -1;
  // sized by the machine (one slice per resident workgroup of the persistent grid), not by the batch; the figure covers a
  // warm start (B = L^-1 N, n x n) as well as a cold one (R alone, and only beyond n = 104)
  return (long long)sizeof(double) * big_grid(n, m, nb, B, two) * jrlqp::batch_big_ws_doubles_per_wg(n, r_in_lds, 1);
}

static int gi_batch_common(int warm,
                           const int * das,
                           int n,
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
                           void * dws,
                           long long ws_bytes,
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
  const bool bign = n > JRLQP_BATCH_WAVE_N; // more rows than one lane-per-row wavefront: gi_batch_big.hip
  int r_in_lds = 0;
  // (n = 79 .. 95: two workgroups per CU with the packed R in the workspace -- when there is a workspace and the batch has more
  // problems than the device has CUs; a cold solve without a workspace keeps R in LDS wherever it fits)
  const int have_ws = ((dws != nullptr || warm) && bign && B > device_cus()) ? 1 : 0;
  if(bign && jrlqp::batch_big_lds_bytes(n, m, nb, &r_in_lds, have_ws) < 0) return JRLQP_ERR_UNSUPPORTED_SIZE;
  const int cmode = bign ? 0 : jrlqp::batch_pick_cmode(n, m, nb, warm);
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
  a.prof = g_prof[JRLQP_PROFILE_GI_BATCH].p;
  a.prof_stride = g_prof[JRLQP_PROFILE_GI_BATCH].stride;
  if(a.prof && B > g_prof[JRLQP_PROFILE_GI_BATCH].nprob) return JRLQP_ERR_INVALID_ARGUMENT;
  a.warm = warm;
  a.box = 0;
  a.as_in = das;
  a.tune = env_tune();
  a.two_per_cu = have_ws;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if(bign)
  {
    // persistent grid: as many workgroups per CU as their LDS allows (two up to n = 88, one beyond), each looping over its
    // share of the batch
    const int nwg = big_grid(n, m, nb, B, have_ws);
    if(nwg < 1) return JRLQP_ERR_NO_DEVICE;
    double * ws = nullptr;
    const long per_wg = jrlqp::batch_big_ws_doubles_per_wg(n, r_in_lds, warm);
    if(per_wg > 0)
    {
      // R (packed, 66 KB at n = 128) when it does not fit beside J, and B = L^-1 N of a warm start, live in the CALLER's workspace
      // (jrlqp_gi_batch_workspace_bytes) -- nothing is allocated inside a solve (tests/GoldfarbIdnaniSolverTest.cpp:109-116)
      const long long need = (long long)sizeof(double) * nwg * per_wg;
      if(!dws || ws_bytes < need || (reinterpret_cast<size_t>(dws) & 15)) return JRLQP_ERR_WORKSPACE;
      ws = static_cast<double *>(dws);
    }
    hipError_t e = jrlqp::launch_gi_batch_big(a, nwg, ws, st);
    return e == hipSuccess ? JRLQP_OK : hip_fail(e);
  }
  hipError_t e = jrlqp::launch_gi_batch(a, cmode, st);
  if(e != hipSuccess) return hip_fail(e);
  return JRLQP_OK;
}

long jrlqp_gi_batch_warm_lds_bytes(int n, int m, int nb, int * c_in_lds)
{
  if(n > JRLQP_BATCH_WAVE_N && n <= JRLQP_BATCH_MAX_N && m >= 0 && (nb == 0 || nb == n))
  {
    if(c_in_lds) *c_in_lds = 0;
    return jrlqp::batch_big_lds_bytes(n, m, nb, nullptr); // (B = L^-1 N lives in the workspace, not in LDS)
  }
  if(n < 1 || n > JRLQP_BATCH_WAVE_N || m < 0 || (nb != 0 && nb != n)) return -1;
  int cmode = jrlqp::batch_pick_cmode(n, m, nb, 1);
  if(cmode < 0) return -1;
  if(c_in_lds) *c_in_lds = cmode;
  return jrlqp::batch_lds_bytes(n, m, nb, cmode, 1);
}

int jrlqp_gi_solve_batch(int n, int m, int nb, int B, const double * dG, const double * da, const double * dC,
                         const double * dbl, const double * dbu, const double * dxl, const double * dxu,
                         const jrlqp_options * opts, double * dx, double * du, int * dstatus, int * dterm, int * diters,
                         double * df, double * dL, jrlqp_trace_rec * dtrace, int trace_cap, void * stream)
{
  return gi_batch_common(0, nullptr, n, m, nb, B, dG, da, dC, dbl, dbu, dxl, dxu, opts, dx, du, dstatus, dterm, diters, df,
                         dL, dtrace, trace_cap, nullptr, 0, stream);
}

int jrlqp_gi_solve_batch_ws(int warm, int n, int m, int nb, int B, const double * dG, const double * da, const double * dC,
                            const double * dbl, const double * dbu, const double * dxl, const double * dxu, const int * das,
                            const jrlqp_options * opts, double * dx, double * du, int * dstatus, int * dterm, int * diters,
                            double * df, double * dL, jrlqp_trace_rec * dtrace, int trace_cap, void * dworkspace,
                            long long workspace_bytes, void * stream)
{
  if(workspace_bytes < 0 || (workspace_bytes > 0 && !dworkspace)) return JRLQP_ERR_INVALID_ARGUMENT;
  return gi_batch_common(warm ? 1 : 0, warm ? das : nullptr, n, m, nb, B, dG, da, dC, dbl, dbu, dxl, dxu, opts, dx, du, dstatus, dterm,
                         diters, df, dL, dtrace, trace_cap, dworkspace, workspace_bytes, stream);
}

int jrlqp_gi_solve_batch_warm(int n, int m, int nb, int B, const double * dG, const double * da, const double * dC,
                              const double * dbl, const double * dbu, const double * dxl, const double * dxu,
                              const int * das, const jrlqp_options * opts, double * dx, double * du, int * dstatus,
                              int * dterm, int * diters, double * df, double * dL, jrlqp_trace_rec * dtrace,
                              int trace_cap, void * stream)
{
  return gi_batch_common(1, das, n, m, nb, B, dG, da, dC, dbl, dbu, dxl, dxu, opts, dx, du, dstatus, dterm, diters, df, dL,
                         dtrace, trace_cap, nullptr, 0, stream);
}

// ---- one process, several devices (the batch shards with no exchange during the solve) ---------------------------
void jrlqp_shard_range(int B_total, int nshards, int r, int * first, int * count)
{
  // contiguous and balanced: the first (B_total % nshards) shards hold one problem more (jrl-qp_amd/dist.py shard_range)
  int f = 0, c = 0;
  if(B_total >= 0 && nshards > 0 && r >= 0 && r < nshards)
  {
    const int base = B_total / nshards, rem = B_total % nshards;
    f = r * base + (r < rem ? r : rem);
    c = base + (r < rem ? 1 : 0);
  }
  if(first) *first = f;
  if(count) *count = c;
}

namespace
{
// the caller's current device, restored on every way out
struct DeviceGuard
{
  int prev = -1;
  DeviceGuard() { if(hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceGuard() { if(prev >= 0) (void)hipSetDevice(prev); }
};
} // namespace

long long jrlqp_gi_batch_workspace_bytes_on(int device, int n, int m, int nb, int B)
{
  DeviceGuard g;
  if(hipSetDevice(device) != hipSuccess) return -1;
  return jrlqp_gi_batch_workspace_bytes(n, m, nb, B);
}

// debug state of the multi-device entry (tests only): a device may be listed more than once (two shards, two streams, two workspaces
// on ONE device: the rehearsal of a multi-GPU run on a one-GPU box), and shard `g_multi_fail_shard` reports a failure instead of
// being enqueued; g_multi_failed: the shard whose enqueue failed in the last call (-1: none)
static int g_multi_shared = 0, g_multi_fail_shard = -1, g_multi_failed = -1;

int jrlqp_debug_multi_allow_shared_device(int on)
{
  g_multi_shared = on ? 1 : 0;
  return JRLQP_OK;
}
int jrlqp_debug_multi_inject_failure(int shard)
{
  g_multi_fail_shard = shard;
  return JRLQP_OK;
}
int jrlqp_gi_multi_failed_shard(void)
{
  return g_multi_failed;
}

static int multi_check_devices(int ndev, const int * device_ids)
{
  int visible = 0;
  if(hipGetDeviceCount(&visible) != hipSuccess || visible < 1) return JRLQP_ERR_NO_DEVICE;
  const bool shared = g_multi_shared || std::getenv("JRLQP_MULTI_SHARED_DEVICE");
  for(int r = 0; r < ndev; ++r)
  {
    if(device_ids[r] < 0 || device_ids[r] >= visible) return JRLQP_ERR_INVALID_ARGUMENT;
    for(int k = 0; k < r && !shared; ++k)
      if(device_ids[k] == device_ids[r]) return JRLQP_ERR_INVALID_ARGUMENT; // one shard per device
  }
  return JRLQP_OK;
}

// wait for the shards [0, upto) that hold problems (a shard without problems was never enqueued: its buffers need not be filled in)
static int multi_drain(int B_total, int ndev, const int * device_ids, const jrlqp_shard_buffers * shards, int upto)
{
  int rc = JRLQP_OK;
  for(int r = 0; r < upto; ++r)
  {
    int first = 0, count = 0;
    jrlqp_shard_range(B_total, ndev, r, &first, &count);
    if(count == 0) continue;
    if(hipSetDevice(device_ids[r]) != hipSuccess || hipStreamSynchronize(static_cast<hipStream_t>(shards[r].stream)) != hipSuccess) rc = JRLQP_ERR_HIP;
  }
  return rc;
}

int jrlqp_gi_solve_batch_multi(int warm, int n, int m, int nb, int B_total, int ndev, const int * device_ids,
                               const jrlqp_shard_buffers * shards, const jrlqp_options * opts, int synchronize)
{
  g_multi_failed = -1;
  if(ndev < 1 || !device_ids || !shards || B_total < 0) return JRLQP_ERR_INVALID_ARGUMENT;
  if(int rc = multi_check_devices(ndev, device_ids)) return rc;
  DeviceGuard guard;
  // enqueue every shard before waiting for any: the devices run side by side, the host thread only launches
  for(int r = 0; r < ndev; ++r)
  {
    int first = 0, count = 0;
    jrlqp_shard_range(B_total, ndev, r, &first, &count);
    if(count == 0) continue;
    const jrlqp_shard_buffers & b = shards[r];
    int rc = JRLQP_OK;
    if(hipSetDevice(device_ids[r]) != hipSuccess)
      rc = JRLQP_ERR_HIP;
    else if(r == g_multi_fail_shard)
      rc = JRLQP_ERR_INVALID_ARGUMENT; // (injected: tests/test_gpu_dist.py)
    else
      rc = gi_batch_common(warm ? 1 : 0, warm ? b.das : nullptr, n, m, nb, count, b.dG, b.da, b.dC, b.dbl, b.dbu, b.dxl, b.dxu, opts,
                           b.dx, b.du, b.dstatus, b.dterm, b.diters, b.df, b.dL, nullptr, 0, b.dworkspace, b.workspace_bytes, b.stream);
    if(rc != JRLQP_OK)
    {
      // the shards enqueued so far are running: wait for them, so that the caller may release or reuse every buffer on return; the
      // shards behind the failing one were never started (jrlqp_gi_multi_failed_shard says which one failed)
      g_multi_failed = r;
      (void)multi_drain(B_total, ndev, device_ids, shards, r);
      return rc;
    }
  }
  if(synchronize) return multi_drain(B_total, ndev, device_ids, shards, ndev);
  return JRLQP_OK;
}

int jrlqp_gi_gather_x_multi(int n, int B_total, int ndev, const int * device_ids, const jrlqp_shard_buffers * shards, int root,
                            double * dx_all_on_root, int synchronize)
{
  if(n < 1 || ndev < 1 || B_total < 0 || !device_ids || !shards || root < 0 || root >= ndev || !dx_all_on_root) return JRLQP_ERR_INVALID_ARGUMENT;
  if(int rc = multi_check_devices(ndev, device_ids)) return rc;
  DeviceGuard guard;
  for(int r = 0; r < ndev; ++r)
  {
    int first = 0, count = 0;
    jrlqp_shard_range(B_total, ndev, r, &first, &count);
    if(count == 0) continue;
    // every copy is ordered behind its shard's solve on the shard's OWN stream (a device-to-device copy over xGMI; on the root's
    // device it is a plain on-device copy), so nothing waits on the host
    HIP_TRY(hipSetDevice(device_ids[r]));
    const size_t bytes = sizeof(double) * (size_t)n * count;
    double * dst = dx_all_on_root + (size_t)first * n;
    hipStream_t st = static_cast<hipStream_t>(shards[r].stream);
    if(device_ids[r] == device_ids[root])
      HIP_TRY(hipMemcpyAsync(dst, shards[r].dx, bytes, hipMemcpyDeviceToDevice, st));
    else
      HIP_TRY(hipMemcpyPeerAsync(dst, device_ids[root], shards[r].dx, device_ids[r], bytes, st));
  }
  if(synchronize) return multi_drain(B_total, ndev, device_ids, shards, ndev);
  return JRLQP_OK;
}

static int box_common(int n, int B, const double * dx0, const double * dc, const double * dbl, const double * dxl, const double * dxu,
                      const jrlqp_options * opts, double * dx, double * du, int * dstatus, int * dterm, int * diters, double * df,
                      jrlqp_trace_rec * dtrace, int trace_cap, void * dws, long long ws_bytes, void * stream)
{
  if(n < 1 || B < 0) return JRLQP_ERR_INVALID_ARGUMENT;
  if(B == 0) return JRLQP_OK;
  if(!dx0 || !dc || !dbl || !dxl || !dxu || !dx || !du || !dstatus || !dterm || !diters || !df) return JRLQP_ERR_INVALID_ARGUMENT;
  if(dtrace && trace_cap <= 0) return JRLQP_ERR_INVALID_ARGUMENT;
  if(n > JRLQP_BATCH_MAX_N) return JRLQP_ERR_UNSUPPORTED_SIZE;
  jrlqp_options o;
  jrlqp_default_options(&o);
  if(opts) o = *opts;
  jrlqp::BatchArgs a{};
  a.n = n; a.m = 1; a.nb = n; a.B = B;
  a.ld = n | 1;
  a.max_iter = o.max_iter;
  a.trace_cap = dtrace ? trace_cap : 0;
  a.big_bnd = o.big_bnd;
  a.G = nullptr; a.a = dx0; a.C = dc; a.bl = dbl; a.bu = nullptr; // G = I and bu = +inf are implicit
  a.xl = dxl; a.xu = dxu;
  a.x = dx; a.u = du; a.status = dstatus; a.term = dterm; a.iters = diters; a.f = df;
  a.Lout = nullptr;
  a.trace = dtrace;
  a.prof = nullptr; a.prof_stride = 0;
  a.warm = 0; a.as_in = nullptr;
  a.box = 1;
  a.tune = env_tune();
  hipStream_t st = static_cast<hipStream_t>(stream);
  if(n <= JRLQP_BATCH_WAVE_N)
  {
    // the closed-form prologue as lane = variable on ONE wavefront (gi_batch_kernel.h, BOX)
    if(jrlqp::batch_lds_bytes(n, 1, n, 1, 0) > jrlqp::kLdsPerCU) return JRLQP_ERR_UNSUPPORTED_SIZE;
    hipError_t e = jrlqp::launch_gi_batch_box(a, st);
    return e == hipSuccess ? JRLQP_OK : hip_fail(e);
  }
  // beyond one wavefront of variables (round 4): the large-n batched kernel with the same closed form, thread = variable
  // (gi_batch_big.hip, BOX); R (past n = 104) and J (past n = 128) live in the caller's workspace like a cold dense solve's
  int r_in_lds = 0;
  // (the box instantiations run one workgroup per CU: R stays in LDS wherever it fits, whatever the caller passed)
  if(jrlqp::batch_big_lds_bytes(n, 1, n, &r_in_lds, 0) < 0) return JRLQP_ERR_UNSUPPORTED_SIZE;
  const int nwg = big_grid(n, 1, n, B, 0);
  if(nwg < 1) return JRLQP_ERR_NO_DEVICE;
  double * ws = nullptr;
  const long per_wg = jrlqp::batch_big_ws_doubles_per_wg(n, r_in_lds, 0);
  if(per_wg > 0)
  {
    const long long need = (long long)sizeof(double) * nwg * per_wg;
    if(!dws || ws_bytes < need || (reinterpret_cast<size_t>(dws) & 15)) return JRLQP_ERR_WORKSPACE;
    ws = static_cast<double *>(dws);
  }
  hipError_t e = jrlqp::launch_gi_batch_big(a, nwg, ws, st);
  return e == hipSuccess ? JRLQP_OK : hip_fail(e);
}

int jrlqp_box_solve_batch(int n, int B, const double * dx0, const double * dc, const double * dbl, const double * dxl,
                          const double * dxu, const jrlqp_options * opts, double * dx, double * du, int * dstatus, int * dterm,
                          int * diters, double * df, jrlqp_trace_rec * dtrace, int trace_cap, void * stream)
{
  return box_common(n, B, dx0, dc, dbl, dxl, dxu, opts, dx, du, dstatus, dterm, diters, df, dtrace, trace_cap, nullptr, 0, stream);
}

int jrlqp_box_solve_batch_ws(int n, int B, const double * dx0, const double * dc, const double * dbl, const double * dxl,
                             const double * dxu, const jrlqp_options * opts, double * dx, double * du, int * dstatus, int * dterm,
                             int * diters, double * df, jrlqp_trace_rec * dtrace, int trace_cap, void * dworkspace,
                             long long workspace_bytes, void * stream)
{
  if(workspace_bytes < 0 || (workspace_bytes > 0 && !dworkspace)) return JRLQP_ERR_INVALID_ARGUMENT;
  return box_common(n, B, dx0, dc, dbl, dxl, dxu, opts, dx, du, dstatus, dterm, diters, df, dtrace, trace_cap, dworkspace, workspace_bytes,
                    stream);
}

int jrlqp_blocks_create(int kind, int nblk, const int * block_sizes, int nprob, int max_nrhs, jrlqp_blocks ** out)
{
  if(!out || !block_sizes || kind < 0 || kind > 2 || nblk < 1 || nprob < 1 || max_nrhs < 1) return JRLQP_ERR_INVALID_ARGUMENT;
  int maxb = 0, ntot = 0;
  for(int i = 0; i < nblk; ++i)
  {
    if(block_sizes[i] < 1) return JRLQP_ERR_INVALID_ARGUMENT;
    if(block_sizes[i] > JRLQP_BLOCK_MAX) return JRLQP_ERR_UNSUPPORTED_SIZE;
    maxb = block_sizes[i] > maxb ? block_sizes[i] : maxb;
    ntot += block_sizes[i];
  }
  jrlqp_blocks * p = new(std::nothrow) jrlqp_blocks;
  if(!p) return JRLQP_ERR_WORKSPACE;
  p->kind = kind; p->nblk = nblk; p->nprob = nprob; p->max_nrhs = max_nrhs; p->maxb = maxb; p->ntot = ntot;
  p->h_bs.assign(block_sizes, block_sizes + nblk);
  const int ns = nblk > 1 ? nblk - 1 : 1;
  hipError_t e = hipMalloc(&p->d_bs, sizeof(int) * nblk);
  if(e == hipSuccess) e = hipMalloc(&p->d_ldd, sizeof(int) * nblk);
  if(e == hipSuccess) e = hipMalloc(&p->d_lds, sizeof(int) * ns);
  if(e == hipSuccess) e = hipMalloc(&p->d_fail, sizeof(int) * nprob);
  if(e == hipSuccess) e = hipMalloc(&p->d_diag, sizeof(double *) * (size_t)nprob * nblk);
  if(e == hipSuccess) e = hipMalloc(&p->d_side, sizeof(double *) * (size_t)nprob * ns);
  if(e == hipSuccess) e = hipMalloc(&p->d_tmp, sizeof(double) * (size_t)nprob * ntot * max_nrhs);
  if(e == hipSuccess) e = hipMemcpy(p->d_bs, block_sizes, sizeof(int) * nblk, hipMemcpyHostToDevice);
  if(e == hipSuccess) e = hipMalloc(&p->d_off, sizeof(int) * (nblk + 1));
  if(e == hipSuccess)
  {
    // (logical order: the up arrow's blocks are rotated by one, structured.hip dsel)
    std::vector<int> off(nblk + 1, 0);
    for(int i = 0; i < nblk; ++i) off[i + 1] = off[i] + block_sizes[(kind == 1) ? (i + 1) % nblk : i];
    e = hipMemcpy(p->d_off, off.data(), sizeof(int) * (nblk + 1), hipMemcpyHostToDevice);
  }
  if(e == hipSuccess) e = p->stage.init(sizeof(int) * 2 * (size_t)(nblk + ns + 2) + sizeof(double *) * (size_t)nprob * (nblk + ns));
  if(e != hipSuccess)
  {
    jrlqp_blocks_destroy(p);
    return hip_fail(e);
  }
  *out = p;
  return JRLQP_OK;
}

void jrlqp_blocks_destroy(jrlqp_blocks * p)
{
  if(!p) return;
  (void)hipFree(p->d_off); (void)hipFree(p->d_bs); (void)hipFree(p->d_ldd); (void)hipFree(p->d_lds); (void)hipFree(p->d_fail);
  (void)hipFree(p->d_diag); (void)hipFree(p->d_side); (void)hipFree(p->d_tmp);
  p->stage.release();
  delete p;
}

int jrlqp_blocks_bind(jrlqp_blocks * p, double * const * ddiag, const int * ldd, double * const * dside, const int * lds, void * stream)
{
  if(!p || !ddiag || !ldd || (p->nblk > 1 && (!dside || !lds))) return JRLQP_ERR_INVALID_ARGUMENT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ns = p->nblk - 1;
  for(int i = 0; i < p->nblk; ++i)
    if(ldd[i] < p->h_bs[i]) return JRLQP_ERR_INVALID_ARGUMENT;
  for(int i = 0; i < ns; ++i)
  { // rows of side block i: kind 0 n_last, kind 1 / kind 2 n_{i+1}
    const int rows = (p->kind == 0) ? p->h_bs[p->nblk - 1] : p->h_bs[i + 1];
    if(lds[i] < rows) return JRLQP_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(p->stage.reuse()); // the previous bind's copies have left the staging area
  const size_t ia = ((sizeof(int) * (size_t)(p->nblk + (ns > 0 ? ns : 0)) + 15) / 16) * 16;
  int * s_ldd = reinterpret_cast<int *>(p->stage.p);
  int * s_lds = s_ldd + p->nblk;
  double ** s_diag = reinterpret_cast<double **>(p->stage.p + ia);
  double ** s_side = s_diag + (size_t)p->nprob * p->nblk;
  for(int i = 0; i < p->nblk; ++i) s_ldd[i] = ldd[i];
  for(int i = 0; i < ns; ++i) s_lds[i] = lds[i];
  for(size_t i = 0; i < (size_t)p->nprob * p->nblk; ++i) s_diag[i] = ddiag[i];
  for(size_t i = 0; ns > 0 && i < (size_t)p->nprob * ns; ++i) s_side[i] = dside[i];
  HIP_TRY(hipMemcpyAsync(p->d_ldd, s_ldd, sizeof(int) * p->nblk, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(p->d_diag, s_diag, sizeof(double *) * (size_t)p->nprob * p->nblk, hipMemcpyHostToDevice, st));
  if(ns > 0)
  {
    HIP_TRY(hipMemcpyAsync(p->d_lds, s_lds, sizeof(int) * ns, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p->d_side, s_side, sizeof(double *) * (size_t)p->nprob * ns, hipMemcpyHostToDevice, st));
  }
  HIP_TRY(p->stage.submitted(st));
  p->bound = true;
  // (can the side blocks of a down arrow be streamed into LDS 16 bytes per lane?  structured.hip, arrow_lsolve_head_ring_kernel)
  p->ring_ok = (p->kind == 0 && ns >= 1 && ns <= 2048 && (p->h_bs[p->nblk - 1] & 1) == 0) ? 1 : 0;
  for(int i = 0; i < p->nblk && p->ring_ok; ++i)
    if(p->h_bs[i] > 32) p->ring_ok = 0;
  for(int i = 0; i < ns && p->ring_ok; ++i)
    if(lds[i] & 1) p->ring_ok = 0;
  for(size_t i = 0; ns > 0 && i < (size_t)p->nprob * ns && p->ring_ok; ++i)
    if(reinterpret_cast<size_t>(dside[i]) & 15) p->ring_ok = 0;
  return JRLQP_OK;
}

static jrlqp::BlockDesc make_desc(const jrlqp_blocks * p, long mstride)
{
  jrlqp::BlockDesc d;
  d.nblk = p->nblk;
  d.up = (p->kind == 1) ? 1 : 0;
  d.mstride = mstride;
  d.bs = p->d_bs; d.ldd = p->d_ldd; d.lds = p->d_lds;
  d.diag = p->d_diag; d.side = p->d_side; d.fail = p->d_fail;
  d.h_bs = p->h_bs.data();
  d.xs_cap = 0; // (set by the factorisation launch helpers)
  d.ring_ok = p->ring_ok;
  d.off = p->d_off;
  d.ntot = p->ntot;
  return d;
}

int jrlqp_blocks_llt(jrlqp_blocks * p, void * stream)
{
  if(!p || !p->bound) return JRLQP_ERR_INVALID_ARGUMENT;
  jrlqp::BlockDesc d = make_desc(p, 0);
  hipError_t e = (p->kind == 2) ? jrlqp::launch_tri_llt(d, p->nprob, p->maxb, static_cast<hipStream_t>(stream))
                                : jrlqp::launch_block_arrow_llt(d, p->nprob, p->maxb, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? JRLQP_OK : hip_fail(e);
}

int jrlqp_blocks_result(jrlqp_blocks * p, int * ok_host, void * stream)
{
  if(!p || !ok_host) return JRLQP_ERR_INVALID_ARGUMENT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(ok_host, p->d_fail, sizeof(int) * p->nprob, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for(int i = 0; i < p->nprob; ++i) ok_host[i] = ok_host[i] ? 0 : 1;
  return JRLQP_OK;
}

int jrlqp_blocks_solve(jrlqp_blocks * p, int transpose, double * dM, int ldm, long mstride, int nrhs, int start, int end, void * stream)
{
  if(!p || !p->bound || !dM || nrhs < 1 || nrhs > p->max_nrhs || ldm < p->ntot) return JRLQP_ERR_INVALID_ARGUMENT;
  jrlqp::BlockDesc d = make_desc(p, mstride);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e;
  if(p->kind == 2)
    e = jrlqp::launch_tri_solve(d, p->nprob, transpose != 0, dM, ldm, nrhs, transpose ? end : start, st);
  else
    e = jrlqp::launch_block_arrow_solve(d, p->nprob, transpose != 0, dM, ldm, nrhs, start, end, p->d_tmp, st);
  return e == hipSuccess ? JRLQP_OK : hip_fail(e);
}

int jrlqp_debug_set_profile(int kernel, long long * dprof, int slots_per_problem, int nprob)
{
  if(kernel != JRLQP_PROFILE_GI_BATCH && kernel != JRLQP_PROFILE_BLOCKGI) return JRLQP_ERR_INVALID_ARGUMENT;
  if(dprof && (slots_per_problem < JRLQP_PROFILE_SLOTS || nprob < 1)) return JRLQP_ERR_INVALID_ARGUMENT;
  g_prof[kernel].p = dprof;
  g_prof[kernel].stride = dprof ? slots_per_problem : 0;
  g_prof[kernel].nprob = dprof ? nprob : 0;
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

// ---- structured GI solver (experimental::BlockGISolver) ---------------------------------------------------------
int jrlqp_blockgi_create(int kind, int nblk, const int * block_sizes, int ncb, const int * cvar, const int * ccstr, int use_bounds,
                         int nprob, int max_iter, jrlqp_blockgi ** out)
{
  return jrlqp_blockgi_create_form(kind, nblk, block_sizes, ncb, cvar, ccstr, use_bounds, nprob, max_iter, JRLQP_BLOCKGI_FORM_AUTO, out);
}

int jrlqp_blockgi_form(const jrlqp_blockgi * h)
{
  return h ? h->form : -1;
}

int jrlqp_blockgi_create_form(int kind, int nblk, const int * block_sizes, int ncb, const int * cvar, const int * ccstr, int use_bounds,
                              int nprob, int max_iter, int form, jrlqp_blockgi ** out)
{
  if(form < JRLQP_BLOCKGI_FORM_AUTO || form > JRLQP_BLOCKGI_FORM_C9) return JRLQP_ERR_INVALID_ARGUMENT;
  if(!out || !block_sizes || kind < 0 || kind > 2 || nblk < 1 || ncb < 0 || nprob < 1 || max_iter < 1) return JRLQP_ERR_INVALID_ARGUMENT;
  if(ncb > 0 && (!cvar || !ccstr)) return JRLQP_ERR_INVALID_ARGUMENT;
  int n = 0, nv = 0, m = 0;
  for(int i = 0; i < nblk; ++i)
  {
    if(block_sizes[i] < 1) return JRLQP_ERR_INVALID_ARGUMENT;
    if(block_sizes[i] > JRLQP_BLOCKGI_BLOCK_MAX) return JRLQP_ERR_UNSUPPORTED_SIZE;
    n += block_sizes[i];
  }
  for(int i = 0; i < ncb; ++i)
  {
    if(cvar[i] < 0 || ccstr[i] < 0) return JRLQP_ERR_INVALID_ARGUMENT;
    nv += cvar[i];
    m += ccstr[i];
  }
  if(ncb > 0 && nv != n) return JRLQP_ERR_INVALID_ARGUMENT; // StructuredC::nbVar() == G.nbVar() (BlockGISolver.cpp:35)
  if(n > jrlqp::kBgiMaxN) return JRLQP_ERR_UNSUPPORTED_SIZE;
  const int nb = use_bounds ? n : 0;
  // Which form of the QR factors the loop keeps: the reference's orthonormal sequence (blockgi.hip, oracle
  // orc_blockgi_solve) or contract C9 (blockgi_c9.hip, oracle orc_blockgi_solve_c9: sparse B = L^-1 N and S = R^-1).  AUTO: C9 for the
  // arrows -- there a column of B is two blocks long --, the sequence for the tridiagonal shape (its columns of B are dense from the
  // window on, and the reference's fixtures of that shape take four iterations).  JRLQP_BLOCKGI_FORM=sequence|c9 overrides AUTO.
  const bool c9_fits = jrlqp::blockgi_c9_lds_bytes(n, m, nb, nblk, ncb) <= jrlqp::kLdsPerCU && nblk < 65535;
  if(form == JRLQP_BLOCKGI_FORM_AUTO)
  {
    form = (kind != 2 && n > 1024 && c9_fits) ? JRLQP_BLOCKGI_FORM_C9 : JRLQP_BLOCKGI_FORM_SEQUENCE;
    if(const char * e = std::getenv("JRLQP_BLOCKGI_FORM"))
      form = (e[0] == 'c' && c9_fits) ? JRLQP_BLOCKGI_FORM_C9 : (e[0] == 's') ? JRLQP_BLOCKGI_FORM_SEQUENCE : form;
  }
  if(form == JRLQP_BLOCKGI_FORM_C9 ? !c9_fits : jrlqp::blockgi_lds_bytes(n, m, nb, nblk) > jrlqp::kLdsPerCU) return JRLQP_ERR_UNSUPPORTED_SIZE;
  jrlqp_blockgi * h = new(std::nothrow) jrlqp_blockgi;
  if(!h) return JRLQP_ERR_WORKSPACE;
  h->form = form;
  int rc = jrlqp_blocks_create(kind, nblk, block_sizes, nprob, 1, &h->blocks);
  if(rc != JRLQP_OK)
  {
    delete h;
    return rc;
  }
  h->kind = kind; h->nblk = nblk; h->n = n; h->m = m; h->nb = nb; h->ncb = ncb; h->nprob = nprob; h->max_iter = max_iter;
  if(ncb > 0) h->h_cvar.assign(cvar, cvar + ncb);
  h->n_pad = (n + 3) & ~3;
  h->qcap = (((n < m + nb) ? n : (m + nb)) + 1 + 3) & ~3;
  h->row_len = h->n_pad > 2 * h->qcap ? h->n_pad : 2 * h->qcap;
  h->rlen = ((long)h->qcap * (h->qcap + 1) / 2 + 3) & ~3L;
  std::vector<int> boff(nblk), cumv(ncb + 1), cumc(ncb + 1), toblk(m + 1);
  for(int i = 0, o = 0; i < nblk; ++i)
  {
    boff[i] = o;
    o += block_sizes[i];
  }
  {
    int cv = 0, cc = 0;
    for(int i = 0; i < ncb; ++i)
    {
      cumv[i] = cv;
      cumc[i] = cc;
      for(int k = 0; k < ccstr[i]; ++k) toblk[cc + k] = i;
      cv += cvar[i];
      cc += ccstr[i];
    }
    cumv[ncb] = cv;
    cumc[ncb] = cc;
  }
  const size_t nc = ncb > 0 ? ncb : 1;
  const bool c9 = form == JRLQP_BLOCKGI_FORM_C9;
  const size_t pool_b = c9 ? 8 : sizeof(double) * (size_t)nprob * max_iter * h->row_len;
  const size_t r_b = c9 ? 8 : sizeof(double) * (size_t)nprob * h->rlen;
  const size_t desc_b = c9 ? 8 : sizeof(jrlqp::SeqRec) * (size_t)nprob * max_iter;
  hipError_t e = hipMalloc(&h->d_boff, sizeof(int) * nblk);
  if(e == hipSuccess) e = hipMalloc(&h->d_cvar, sizeof(int) * nc);
  if(e == hipSuccess) e = hipMalloc(&h->d_ccstr, sizeof(int) * nc);
  if(e == hipSuccess) e = hipMalloc(&h->d_cumv, sizeof(int) * (nc + 1));
  if(e == hipSuccess) e = hipMalloc(&h->d_cumc, sizeof(int) * (nc + 1));
  if(e == hipSuccess) e = hipMalloc(&h->d_toblk, sizeof(int) * (m + 1));
  if(e == hipSuccess) e = hipMalloc(&h->d_cld, sizeof(int) * nc);
  if(e == hipSuccess) e = hipMalloc(&h->d_cblk, sizeof(double *) * (size_t)nprob * nc);
  if(e == hipSuccess) e = hipMalloc(&h->d_pool, pool_b);
  if(e == hipSuccess) e = hipMalloc(&h->d_Rp, r_b);
  if(e == hipSuccess) e = hipMalloc(&h->d_desc, desc_b);
  if(e == hipSuccess) e = hipMalloc(&h->d_active, sizeof(int) * (size_t)nprob * h->qcap);
  // throughput layout: large problems (n > 1024) in batches of at least half the CUs keep x and u in HBM so that two
  // problems share a CU (same results: the layout does not enter the arithmetic)
  h->xu_global = (n > 1024 && nprob >= 128 && 2 * (c9 ? jrlqp::blockgi_c9_lds_bytes(n, m, nb, nblk, ncb, 1) : jrlqp::blockgi_lds_bytes(n, m, nb, nblk, 1)) <= jrlqp::kLdsPerCU) ? 1 : 0;
  if(const char * lay = std::getenv("JRLQP_BLOCKGI_LAYOUT")) // diagnostic override: "throughput" | "latency"
    if(n > 1024) h->xu_global = (lay[0] == 't') ? 1 : 0;
  if(e == hipSuccess) e = hipMalloc(&h->d_xu, sizeof(double) * (size_t)nprob * (h->n_pad + h->qcap));
  if(e == hipSuccess) e = h->stage.init(sizeof(int) * (nc + 4) + sizeof(double *) * (size_t)nprob * nc);
  if(e == hipSuccess) e = hipMemcpy(h->d_boff, boff.data(), sizeof(int) * nblk, hipMemcpyHostToDevice);
  if(e == hipSuccess && ncb > 0) e = hipMemcpy(h->d_cvar, cvar, sizeof(int) * ncb, hipMemcpyHostToDevice);
  if(e == hipSuccess && ncb > 0) e = hipMemcpy(h->d_ccstr, ccstr, sizeof(int) * ncb, hipMemcpyHostToDevice);
  if(e == hipSuccess) e = hipMemcpy(h->d_cumv, cumv.data(), sizeof(int) * (ncb + 1), hipMemcpyHostToDevice);
  if(e == hipSuccess) e = hipMemcpy(h->d_cumc, cumc.data(), sizeof(int) * (ncb + 1), hipMemcpyHostToDevice);
  if(e == hipSuccess) e = hipMemcpy(h->d_toblk, toblk.data(), sizeof(int) * (m + 1), hipMemcpyHostToDevice);
  if(e != hipSuccess)
  {
    jrlqp_blockgi_destroy(h);
    return hip_fail(e);
  }
  h->ws_bytes = pool_b + r_b + desc_b + sizeof(int) * (size_t)nprob * h->qcap;
  {
    // transposed copies of the blocks: diagT_b for every block, sideT_i for every side / sub-diagonal block
    std::vector<int> doff(nblk), soff(nblk > 1 ? nblk - 1 : 1), sc(nblk > 1 ? nblk - 1 : 1);
    long o = 0;
    for(int b = 0; b < nblk; ++b)
    {
      doff[b] = (int)o;
      o += (long)block_sizes[b] * block_sizes[b];
    }
    for(int i = 0; i + 1 < nblk; ++i)
    {
      // side block i: arrow down: G[head, block i] (head x bs_i); arrow up: G[block i+1, head] (bs_{i+1} x bs_0); tri: G[i+1, i]
      const int rows = (kind == 0) ? block_sizes[nblk - 1] : block_sizes[i + 1];
      const int cols = (kind == 0) ? block_sizes[i] : (kind == 1) ? block_sizes[0] : block_sizes[i];
      soff[i] = (int)o;
      sc[i] = cols;
      o += (long)rows * cols;
    }
    h->tw_len = (o + 3) & ~3L;
    e = hipMalloc(&h->d_Tw, sizeof(double) * (size_t)nprob * h->tw_len);
    if(e == hipSuccess) e = hipMalloc(&h->d_tw_doff, sizeof(int) * nblk);
    if(e == hipSuccess) e = hipMalloc(&h->d_tw_soff, sizeof(int) * soff.size());
    if(e == hipSuccess) e = hipMalloc(&h->d_tw_sc, sizeof(int) * sc.size());
    if(e == hipSuccess) e = hipMemcpy(h->d_tw_doff, doff.data(), sizeof(int) * nblk, hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMemcpy(h->d_tw_soff, soff.data(), sizeof(int) * soff.size(), hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMemcpy(h->d_tw_sc, sc.data(), sizeof(int) * sc.size(), hipMemcpyHostToDevice);
    if(e != hipSuccess)
    {
      jrlqp_blockgi_destroy(h);
      return hip_fail(e);
    }
    h->ws_bytes += sizeof(double) * (size_t)nprob * h->tw_len;
  }
  if(c9)
  {
    // static structure of contract C9: block order (off-head blocks in logical order, head last; natural for the tridiagonal shape),
    // groups (blocks of C, then one group of bounds per Hessian block) and which groups reach which block (oracle: reach_of / reaches)
    const int up = kind == 1, tri = kind == 2;
    const int ngrp = ncb + (nb ? nblk : 0);
    h->ngrp = ngrp;
    std::vector<int> kap_phys(nblk), off_perm(nblk), kap_of_phys(nblk), kap_prow(n > 0 ? n : 1), kap_row(n > 0 ? n : 1), greach(2 * (size_t)(ngrp > 0 ? ngrp : 1)),
        tk(4 * (size_t)nblk), goff(ngrp > 0 ? ngrp : 1);
    for(int k = 0, o = 0; k < nblk; ++k)
    {
      const int pbk = up ? (k + 1) % nblk : k;
      kap_phys[k] = pbk;
      kap_of_phys[pbk] = k;
      off_perm[k] = o;
      for(int i = 0; i < block_sizes[pbk]; ++i)
      {
        kap_prow[o + i] = k;
        kap_row[boff[pbk] + i] = k;
      }
      o += block_sizes[pbk];
    }
    const int last = tri ? nblk : nblk - 1; // arrows: the head is not an off-head block
    auto reach_of = [&](int wst, int wen, int & lo, int & hi) {
      lo = nblk;
      hi = -1;
      for(int k = 0; k < last; ++k)
      {
        const int o = boff[kap_phys[k]], len = block_sizes[kap_phys[k]];
        if(o < wen && wst < o + len)
        {
          if(k < lo) lo = k;
          hi = k;
        }
      }
      if(tri && lo <= hi) hi = nblk - 1;
    };
    for(int g = 0; g < ncb; ++g)
    {
      reach_of(cumv[g], cumv[g] + cvar[g], greach[2 * g], greach[2 * g + 1]);
      goff[g] = cumc[g];
    }
    if(nb)
      for(int k = 0; k < nblk; ++k)
      {
        reach_of(boff[kap_phys[k]], boff[kap_phys[k]] + block_sizes[kap_phys[k]], greach[2 * (ncb + k)], greach[2 * (ncb + k) + 1]);
        goff[ncb + k] = m + boff[kap_phys[k]];
      }
    auto reaches = [&](int g, int k) { return (!tri && k == nblk - 1) || (greach[2 * g] <= k && k <= greach[2 * g + 1]); };
    for(int k = 0; k < nblk; ++k)
    {
      int glo = 0, ghi = -1, blo = 0, bhi = -1;
      bool first = true;
      for(int g = 0; g < ncb; ++g)
        if(reaches(g, k))
        {
          if(first) glo = g;
          ghi = g;
          first = false;
        }
      first = true;
      for(int g = ncb; g < ngrp; ++g)
        if(reaches(g, k))
        {
          if(first) blo = g;
          bhi = g;
          first = false;
        }
      tk[4 * k] = glo; tk[4 * k + 1] = ghi; tk[4 * k + 2] = blo; tk[4 * k + 3] = bhi;
    }
    const size_t bc_b = sizeof(double) * (size_t)nprob * h->qcap * h->n_pad, s_b = sizeof(double) * (size_t)nprob * h->qcap * h->qcap;
    e = hipMalloc(&h->d_greach, sizeof(int) * greach.size());
    if(e == hipSuccess) e = hipMalloc(&h->d_tk, sizeof(int) * tk.size());
    if(e == hipSuccess) e = hipMalloc(&h->d_kap_prow, sizeof(int) * kap_prow.size());
    if(e == hipSuccess) e = hipMalloc(&h->d_kap_row, sizeof(int) * kap_row.size());
    if(e == hipSuccess) e = hipMalloc(&h->d_off_perm, sizeof(int) * nblk);
    if(e == hipSuccess) e = hipMalloc(&h->d_kap_phys, sizeof(int) * nblk);
    if(e == hipSuccess) e = hipMalloc(&h->d_goff, sizeof(int) * goff.size());
    if(e == hipSuccess) e = hipMalloc(&h->d_Bc, bc_b);
    if(e == hipSuccess) e = hipMalloc(&h->d_Sm, s_b);
    if(e == hipSuccess) e = hipMalloc(&h->d_St, s_b);
    if(e == hipSuccess) e = hipMalloc(&h->d_sgrp, sizeof(int) * (size_t)nprob * h->qcap);
    if(e == hipSuccess) e = hipMalloc(&h->d_freel, sizeof(int) * (size_t)nprob * h->qcap);
    if(e == hipSuccess) e = hipMemcpy(h->d_greach, greach.data(), sizeof(int) * greach.size(), hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMemcpy(h->d_tk, tk.data(), sizeof(int) * tk.size(), hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMemcpy(h->d_kap_prow, kap_prow.data(), sizeof(int) * kap_prow.size(), hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMemcpy(h->d_kap_row, kap_row.data(), sizeof(int) * kap_row.size(), hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMemcpy(h->d_off_perm, off_perm.data(), sizeof(int) * nblk, hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMemcpy(h->d_kap_phys, kap_phys.data(), sizeof(int) * nblk, hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMemcpy(h->d_goff, goff.data(), sizeof(int) * goff.size(), hipMemcpyHostToDevice);
    if(e != hipSuccess)
    {
      jrlqp_blockgi_destroy(h);
      return hip_fail(e);
    }
    h->ws_bytes += bc_b + 2 * s_b + 2 * sizeof(int) * (size_t)nprob * h->qcap;
  }
  *out = h;
  return JRLQP_OK;
}

void jrlqp_blockgi_destroy(jrlqp_blockgi * h)
{
  if(!h) return;
  jrlqp_blocks_destroy(h->blocks);
  (void)hipFree(h->d_boff); (void)hipFree(h->d_cvar); (void)hipFree(h->d_ccstr); (void)hipFree(h->d_cumv); (void)hipFree(h->d_cumc);
  (void)hipFree(h->d_toblk); (void)hipFree(h->d_cld); (void)hipFree(h->d_cblk); (void)hipFree(h->d_pool); (void)hipFree(h->d_Rp);
  (void)hipFree(h->d_desc); (void)hipFree(h->d_active); (void)hipFree(h->d_xu);
  (void)hipFree(h->d_greach); (void)hipFree(h->d_tk); (void)hipFree(h->d_kap_prow); (void)hipFree(h->d_kap_row); (void)hipFree(h->d_off_perm);
  (void)hipFree(h->d_kap_phys); (void)hipFree(h->d_goff); (void)hipFree(h->d_Bc); (void)hipFree(h->d_Sm); (void)hipFree(h->d_St);
  (void)hipFree(h->d_sgrp); (void)hipFree(h->d_freel);
  (void)hipFree(h->d_Tw); (void)hipFree(h->d_tw_doff); (void)hipFree(h->d_tw_soff); (void)hipFree(h->d_tw_sc);
  h->stage.release();
  delete h;
}

long long jrlqp_blockgi_workspace_bytes(const jrlqp_blockgi * h)
{
  return h ? (long long)h->ws_bytes : -1;
}

int jrlqp_blockgi_bind(jrlqp_blockgi * h, double * const * ddiag, const int * ldd, double * const * doff, const int * ldo,
                       const double * const * dcblk, const int * ldc, void * stream)
{
  if(!h || (h->ncb > 0 && (!dcblk || !ldc))) return JRLQP_ERR_INVALID_ARGUMENT;
  for(int i = 0; i < h->ncb; ++i)
    if(ldc[i] < (h->h_cvar[i] > 1 ? h->h_cvar[i] : 1)) return JRLQP_ERR_INVALID_ARGUMENT; // block i of C is cvar[i] x ccstr[i]
  int rc = jrlqp_blocks_bind(h->blocks, ddiag, ldd, doff, ldo, stream);
  if(rc != JRLQP_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if(h->ncb > 0)
  {
    HIP_TRY(h->stage.reuse());
    const size_t ia = ((sizeof(int) * (size_t)h->ncb + 15) / 16) * 16;
    int * s_cld = reinterpret_cast<int *>(h->stage.p);
    const double ** s_cblk = reinterpret_cast<const double **>(h->stage.p + ia);
    for(int i = 0; i < h->ncb; ++i) s_cld[i] = ldc[i];
    for(size_t i = 0; i < (size_t)h->nprob * h->ncb; ++i) s_cblk[i] = dcblk[i];
    HIP_TRY(hipMemcpyAsync(h->d_cld, s_cld, sizeof(int) * h->ncb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->d_cblk, s_cblk, sizeof(double *) * (size_t)h->nprob * h->ncb, hipMemcpyHostToDevice, st));
    HIP_TRY(h->stage.submitted(st));
  }
  h->bound = true;
  return JRLQP_OK;
}

int jrlqp_blockgi_solve(jrlqp_blockgi * h, const double * da, const double * dbl, const double * dbu, const double * dxl,
                        const double * dxu, const jrlqp_options * opts, double * dx, double * du, int * dstatus, int * dterm,
                        int * diters, double * df, jrlqp_trace_rec * dtrace, int trace_cap, void * stream)
{
  if(!h || !h->bound || !da || !dx || !dterm || !diters || !df) return JRLQP_ERR_INVALID_ARGUMENT;
  if(h->m + h->nb > 0 && (!du || !dstatus)) return JRLQP_ERR_INVALID_ARGUMENT;
  if(h->m > 0 && (!dbl || !dbu)) return JRLQP_ERR_INVALID_ARGUMENT;
  if(h->nb > 0 && (!dxl || !dxu)) return JRLQP_ERR_INVALID_ARGUMENT;
  if(dtrace && trace_cap <= 0) return JRLQP_ERR_INVALID_ARGUMENT;
  jrlqp_options o;
  jrlqp_default_options(&o);
  if(opts) o = *opts;
  if(o.max_iter > h->max_iter && h->form == JRLQP_BLOCKGI_FORM_SEQUENCE) return JRLQP_ERR_WORKSPACE; // the orthonormal sequence grows by one element per iteration
  // lltInPlace (BlockGISolver.cpp:71): the block factorisation kernels of structured.hip, same stream
  int rc = jrlqp_blocks_llt(h->blocks, stream);
  if(rc != JRLQP_OK) return rc;
  const jrlqp_blocks * p = h->blocks;
  jrlqp::BgiArgs a;
  a.kind = h->kind; a.nblk = h->nblk; a.n = h->n;
  a.bs = p->d_bs; a.boff = h->d_boff; a.ldd = p->d_ldd; a.lds = p->d_lds;
  a.diag = p->d_diag; a.side = p->d_side; a.fail = p->d_fail;
  a.m = h->m; a.nb = h->nb; a.ncb = h->ncb;
  a.cvar = h->d_cvar; a.ccstr = h->d_ccstr; a.cum_var = h->d_cumv; a.cum_cstr = h->d_cumc; a.to_block = h->d_toblk; a.cld = h->d_cld;
  a.cblk = h->d_cblk;
  a.a = da; a.bl = dbl; a.bu = dbu; a.xl = dxl; a.xu = dxu;
  a.max_iter = o.max_iter; a.big_bnd = o.big_bnd;
  a.maxb = h->blocks->maxb;
  a.n_pad = h->n_pad; a.qcap = h->qcap; a.row_len = h->row_len;
  a.pool = h->d_pool; a.Rp = h->d_Rp; a.rlen = h->rlen; a.desc = h->d_desc; a.active = h->d_active;
  a.xu_ws = h->d_xu; a.xu_global = h->xu_global;
  a.x = dx; a.u = du; a.status = dstatus; a.term = dterm; a.iters = diters; a.f = df;
  a.trace = dtrace; a.trace_cap = dtrace ? trace_cap : 0;
  a.tune = 0;
  if(const char * e = std::getenv("JRLQP_BGI_TUNE")) a.tune = std::atoi(e);
  a.prof = g_prof[JRLQP_PROFILE_BLOCKGI].p;
  a.prof_stride = g_prof[JRLQP_PROFILE_BLOCKGI].stride;
  if(a.prof && h->nprob > g_prof[JRLQP_PROFILE_BLOCKGI].nprob) return JRLQP_ERR_INVALID_ARGUMENT;
  a.form = h->form; a.ngrp = h->ngrp;
  a.greach = h->d_greach; a.tk = h->d_tk; a.kap_prow = h->d_kap_prow; a.kap_row = h->d_kap_row; a.off_perm = h->d_off_perm;
  a.kap_phys = h->d_kap_phys; a.goff = h->d_goff;
  a.Bc = h->d_Bc; a.Sm = h->d_Sm; a.St = h->d_St; a.sgrp = h->d_sgrp; a.freel = h->d_freel;
  a.Tw = h->d_Tw; a.tw_doff = h->d_tw_doff; a.tw_soff = h->d_tw_soff; a.tw_sc = h->d_tw_sc; a.tw_len = h->tw_len;
  hipError_t e = jrlqp::launch_blockgi_pack(a, h->nprob, static_cast<hipStream_t>(stream));
  if(e != hipSuccess) return hip_fail(e);
  e = (h->form == JRLQP_BLOCKGI_FORM_C9) ? jrlqp::launch_blockgi_c9(a, h->nprob, static_cast<hipStream_t>(stream))
                                                    : jrlqp::launch_blockgi(a, h->nprob, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? JRLQP_OK : hip_fail(e);
}

} // extern "C"
