// gi_dense_host.cpp -- Mode A host side: the DualSolver state machine (src/DualSolver.cpp:91-168) and the
// GoldfarbIdnaniSolver equality pre-activation (src/GoldfarbIdnaniSolver.cpp:268-338) in host C++17, driving the
// hook kernels of gi_dense.hip through the same entry points the C ABI exports.  Only scalars cross back per
// iteration (p, status, t1, l, three dot products, u[q]) through a pinned, mapped mailbox; J, R, C, x, z, d, u, r
// never leave HBM.  No allocation inside solve(): everything is sized in jrlqp_dense_create.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <new>
#include <vector>

#include "../../include/jrlqp_hip.h"
#include "gi_dense.h"

using namespace jrlqp;

struct jrlqp_dense
{
  int n = 0, m = 0, nb = 0;             // current problem sizes (jrlqp_dense_resize), within the capacities below
  int cap_n = 0, cap_m = 0, cap_nb = 0; // what the buffers were allocated for (jrlqp_dense_create): grow-only, like Workspace.h:37-43
  DenseViews v{};
  DenseState s{};
  Mailbox * h_mb = nullptr; // pinned host memory, mapped into the device address space
  bool bound = false;
  // host mirror of internal::ActiveSet (src/internal/ActiveSet.cpp): status per constraint + ordered active list
  std::vector<int> status, active;
  std::vector<double> h_bl, h_bu, h_xl, h_xu, h_u;
  std::vector<double> h_bact;          // warm start: b_act of the initial active set (host copy: shifted by a purge, then uploaded)
  std::vector<unsigned char> h_stb;    // warm start: the status bytes as uploaded to the device
  int it = 0, term = JRLQP_UNKNOWN;
  double f = 0.0;
  int last_hip = 0;
  hipStream_t side = nullptr;          // runs the next pass's constraint scan beside the Givens sweep
  hipStream_t spec = nullptr;          // runs the coefficient chain of an addition ahead of the add/drop decision
  int chain_pending = 0;               // tag of a chain launched on `spec` that no kernel of the main stream has waited for yet (0: none)
  // Hand-off between the streams of a pass: device-side tags (PassSync) where kernels of different streams run side by side -- the
  // normal case --, events where they do not (a profiler collecting counters serialises every dispatch and picks among the queues in
  // its own order: a consumer spinning on a tag would then run before its producer).  Decided once, at creation
  // (dense_probe_concurrency); JRLQP_DENSE_SYNC=events|tags overrides.
  bool use_events = false;
  hipEvent_t ev_x = nullptr, ev_sel = nullptr, ev_d = nullptr, ev_chain = nullptr;
  int tag = 0;                         // pass counter: the step kernel echoes it into the mailbox when its scalars are out
  jrlqp_trace_rec * trace = nullptr;   // jrlqp_dense_set_trace: one record per pass of the library's loop (host memory, caller-owned)
  int trace_cap = 0;
  double * vtrace = nullptr;           // jrlqp_dense_set_trace_vectors: x, u, z, r, f, q of every pass (host memory, caller-owned)
  int vtrace_cap = 0;
  bool speculate_scan = true;          // ... and the next pass's constraint scan with it (blocked scan only)
  bool speculate = true;               // enqueue the sweep of an addition ahead of the decision (StepDev; JRLQP_DENSE_SPECULATE=0 switches it off)
  int inject_fault_pass = -1;          // jrlqp_debug_dense_inject_sync_fault: the pass whose first kernel waits for a tag nobody posts
  int event_reruns = 0;                // solves that were re-run with event edges after a device-side wait gave up
};

namespace
{
#define TRY(x)                                   \
  do                                             \
  {                                              \
    hipError_t e_ = (x);                         \
    if(e_ != hipSuccess)                         \
    {                                            \
      h->last_hip = (int)e_;                     \
      return JRLQP_ERR_HIP;                      \
    }                                            \
  } while(0)

// Wait for the step kernel of pass `tag` by polling the two tags it writes last into the pinned, host-coherent mailbox.  A
// stream synchronisation returns tens of microseconds after the kernel has ended; the poll sees the scalars within a few.  The
// poll gives up after two seconds and lets the caller synchronise the stream (and find the error, if a kernel died).
inline bool wait_tag(const jrlqp::Mailbox * mb, int tag)
{
  const auto t0 = std::chrono::steady_clock::now();
  for(unsigned spin = 0;; ++spin)
  {
    if(mb->seq0 == tag && mb->seq1 == tag)
    {
      std::atomic_thread_fence(std::memory_order_acquire);
      return true;
    }
    if((spin & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return false;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
}

// The tag hand-off lets consumer kernels spin on the device while their producers run; that is only safe while the spinning
// workgroups cannot take the LDS a producer needs.  The workgroups of `jtn` / the scan hold n + 6240 doubles of LDS each and there
// are n / 16 of them: from n ~ 3700 on they fill every CU, and a solve at n = 3900 was measured to stall (a `jtn` spinning on the tag
// of a coefficient kernel that found no CU with 62 KB of LDS left; tools/check_max_size.py) until the wait gave up.  Beyond
// kTagsMaxN unknowns a handle therefore orders its streams by events (n = 3000 and 3500 were measured clean with tags).
constexpr int kTagsMaxN = 2048;

template<class T>
hipError_t dmalloc(T *& p, size_t count)
{
  return hipMalloc(reinterpret_cast<void **>(&p), sizeof(T) * (count ? count : 1));
}
} // namespace

extern "C" {

int jrlqp_dense_create(int n, int m, int nb, jrlqp_dense ** out)
{
  if(!out || n < 1 || m < 0 || (nb != 0 && nb != n)) return JRLQP_ERR_INVALID_ARGUMENT;
  if(n > JRLQP_DENSE_MAX_N) return JRLQP_ERR_UNSUPPORTED_SIZE;
  jrlqp_dense * h = new(std::nothrow) jrlqp_dense;
  if(!h) return JRLQP_ERR_WORKSPACE;
  h->n = n; h->m = m; h->nb = nb;
  h->cap_n = n; h->cap_m = m; h->cap_nb = nb;
  const size_t mt = (size_t)m + nb;
  DenseState & s = h->s;
  hipError_t e = dmalloc(s.J, (size_t)n * n + 2); // (+2: the sweep's 16-byte loads may read one double past the last column, gi_dense.hip rot_sweep)
  if(e == hipSuccess) e = dmalloc(s.R, (size_t)n * n);
  s.S = s.S2 = s.Ptri = nullptr;
  // CANON C10 (gi_dense.hip kBlockN).  JRLQP_DENSE_C10=0 (diagnostic): no S -- r by back substitution on R, the oracle's ORC_LIT_R_SOLVE
  const char * c10env = std::getenv("JRLQP_DENSE_C10");
  const bool c10 = n > 256 && !(c10env && c10env[0] == '0');
  if(e == hipSuccess && c10) e = dmalloc(s.S, (size_t)n * n);
  if(e == hipSuccess && c10) e = dmalloc(s.S2, (size_t)n * n);
  if(e == hipSuccess && c10) e = dmalloc(s.Ptri, (size_t)((n + 63) / 64) * ((n + 63) / 64 + 1) / 2 * 64);
  if(e == hipSuccess) e = dmalloc(s.d, n);
  if(e == hipSuccess) e = dmalloc(s.z, n);
  if(e == hipSuccess) e = dmalloc(s.x, n);
  if(e == hipSuccess) e = dmalloc(s.u, mt + 2);
  if(e == hipSuccess) e = dmalloc(s.r, mt + 2);
  if(e == hipSuccess) e = dmalloc(s.cc, n);
  if(e == hipSuccess) e = dmalloc(s.ss, n);
  if(e == hipSuccess) e = dmalloc(s.cc2, n);
  if(e == hipSuccess) e = dmalloc(s.ss2, n);
  if(e == hipSuccess) e = dmalloc(s.rfin, 1);
  if(e == hipSuccess) e = dmalloc(s.st, mt);
  const size_t parts = ((size_t)m + 15) / 16; // select_partial_kernel: one partial per 16 constraints
  if(e == hipSuccess) e = dmalloc(s.pval, parts);
  if(e == hipSuccess) e = dmalloc(s.pidx, parts);
  if(e == hipSuccess) e = dmalloc(s.pstat, parts);
  if(e == hipSuccess) e = dmalloc(s.info, 2); // {LLT info, arrival counter of llt_panel_fused_kernel (zero between launches)}
  if(e == hipSuccess) e = hipMemset(s.info, 0, 2 * sizeof(int));
  if(e == hipSuccess) e = dmalloc(s.psel, 2);
  if(e == hipSuccess) e = dmalloc(s.wact, (size_t)n + 1);
  if(e == hipSuccess) e = dmalloc(s.bact, n);
  if(e == hipSuccess) e = dmalloc(s.hco, n);
  if(e == hipSuccess) e = dmalloc(s.beta, n);
  if(e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&h->h_mb), sizeof(Mailbox), hipHostMallocMapped);
  if(e == hipSuccess) std::memset(h->h_mb, 0, sizeof(Mailbox));
  if(e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&s.mb), h->h_mb, 0);
  if(e == hipSuccess) e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
  if(e == hipSuccess) e = hipStreamCreateWithFlags(&h->spec, hipStreamNonBlocking);
  if(e == hipSuccess) e = dmalloc(s.sync, 1);
  if(e == hipSuccess) e = hipMemset(s.sync, 0, sizeof(PassSync));
  if(e == hipSuccess) e = dmalloc(s.sd, 1);
  if(e == hipSuccess) e = hipMemset(s.sd, 0, sizeof(StepDev));
  {
    // CANON C8: segment sums of the blocked products (64 rows per segment) and the arrival counters of their groups
    const size_t nseg = ((size_t)n + 63) / 64;
    s.ldpJ = (int)(nseg * 64);
    s.ldpC = m > 0 ? m : 1;
    const size_t gj = (size_t)n / 16 + 2, gz = nseg + 1, gc = (size_t)m / 16 + 2;
    if(e == hipSuccess) e = dmalloc(s.partJ, nseg * s.ldpJ);
    if(e == hipSuccess) e = dmalloc(s.partZ, nseg * s.ldpJ);
    if(e == hipSuccess) e = dmalloc(s.partC, nseg * s.ldpC);
    if(e == hipSuccess) e = dmalloc(s.cntJ, gj);
    if(e == hipSuccess) e = dmalloc(s.cntZ, gz);
    if(e == hipSuccess) e = dmalloc(s.cntC, gc);
    if(e == hipSuccess) e = hipMemset(s.cntJ, 0, sizeof(int) * gj);
    if(e == hipSuccess) e = hipMemset(s.cntZ, 0, sizeof(int) * gz);
    if(e == hipSuccess) e = hipMemset(s.cntC, 0, sizeof(int) * gc);
  }
  if(e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_d, hipEventDisableTiming);
  if(e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_chain, hipEventDisableTiming);
  if(e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_x, hipEventDisableTiming);
  if(e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_sel, hipEventDisableTiming);
  if(e == hipSuccess)
  {
    const char * spc = std::getenv("JRLQP_DENSE_SPECULATE");
    if(spc && std::strcmp(spc, "0") == 0) h->speculate = false;
    if(spc && std::strcmp(spc, "sweep") == 0) h->speculate_scan = false; // (A/B: the sweep ahead of the decision, the scan behind it)
    const char * mode = std::getenv("JRLQP_DENSE_SYNC");
    if(mode && std::strcmp(mode, "events") == 0)
      h->use_events = true;
    else if(mode && std::strcmp(mode, "tags") == 0)
      h->use_events = false;
    else if(n > kTagsMaxN)
      h->use_events = true;
    else
    {
      bool side_by_side = true;
      e = dense_probe_concurrency(s.psel, h->side, h->spec, &side_by_side); // (psel: two ints, rewritten by every selection)
      h->use_events = !side_by_side;
    }
  }
  if(e != hipSuccess)
  {
    jrlqp_dense_destroy(h);
    return JRLQP_ERR_HIP;
  }
  h->status.assign(mt, 0);
  h->active.reserve(n + 1);
  h->h_bl.resize(m); h->h_bu.resize(m); h->h_xl.resize(nb); h->h_xu.resize(nb); h->h_u.resize(n + 2);
  h->h_bact.reserve(n + 1);
  h->h_stb.assign(mt, 0);
  h->active.reserve(mt + 1); // (a warm start may list every constraint before it trims the list to n)
  *out = h;
  return JRLQP_OK;
}

void jrlqp_dense_destroy(jrlqp_dense * h)
{
  if(!h) return;
  DenseState & s = h->s;
  if(h->spec) (void)hipStreamSynchronize(h->spec);
  if(h->side) (void)hipStreamSynchronize(h->side);
  dense_rot_profile_dump();
  (void)hipFree(s.S); (void)hipFree(s.S2); (void)hipFree(s.Ptri);
  (void)hipFree(s.J); (void)hipFree(s.R); (void)hipFree(s.d); (void)hipFree(s.z); (void)hipFree(s.x);
  (void)hipFree(s.u); (void)hipFree(s.r); (void)hipFree(s.cc); (void)hipFree(s.ss); (void)hipFree(s.st);
  (void)hipFree(s.cc2); (void)hipFree(s.ss2); (void)hipFree(s.rfin);
  (void)hipFree(s.pval); (void)hipFree(s.pidx); (void)hipFree(s.pstat); (void)hipFree(s.info); (void)hipFree(s.psel);
  (void)hipFree(s.wact); (void)hipFree(s.bact); (void)hipFree(s.hco); (void)hipFree(s.beta);
  if(h->h_mb) (void)hipHostFree(h->h_mb);
  if(h->side) (void)hipStreamDestroy(h->side);
  if(h->spec) (void)hipStreamDestroy(h->spec);
  (void)hipFree(s.sync);
  (void)hipFree(s.sd);
  (void)hipFree(s.partJ); (void)hipFree(s.partZ); (void)hipFree(s.partC); (void)hipFree(s.cntJ); (void)hipFree(s.cntZ); (void)hipFree(s.cntC);
  if(h->ev_d) (void)hipEventDestroy(h->ev_d);
  if(h->ev_chain) (void)hipEventDestroy(h->ev_chain);
  if(h->ev_x) (void)hipEventDestroy(h->ev_x);
  if(h->ev_sel) (void)hipEventDestroy(h->ev_sel);
  delete h;
}

// DualSolver::resize (DualSolver.h:33) on a pre-sized solver: any problem that fits the capacities of jrlqp_dense_create is
// solved in the buffers already there -- no allocation, host or device (the "Multiple uses" contract of
// tests/GoldfarbIdnaniSolverTest.cpp:101-125).  J and R keep the reference's ld = n of the CURRENT problem.
int jrlqp_dense_resize(jrlqp_dense * h, int n, int m, int nb)
{
  if(!h || n < 1 || m < 0 || (nb != 0 && nb != n)) return JRLQP_ERR_INVALID_ARGUMENT;
  if(n > h->cap_n || m > h->cap_m || nb > h->cap_nb) return JRLQP_ERR_WORKSPACE;
  if(n == h->n && m == h->m && nb == h->nb) return JRLQP_OK;
  h->n = n; h->m = m; h->nb = nb;
  h->bound = false;
  h->status.assign((size_t)m + nb, 0); // (within the capacity reserved at creation: no allocation)
  h->h_stb.assign((size_t)m + nb, 0);
  h->active.clear();
  h->h_bl.resize(m); h->h_bu.resize(m); h->h_xl.resize(nb); h->h_xu.resize(nb);
  return JRLQP_OK;
}

int jrlqp_dense_last_hip_error(const jrlqp_dense * h)
{
  return h ? h->last_hip : 0;
}

// ---- hook-level entry points ---------------------------------------------------------------------------------
int jrlqp_dense_bind(jrlqp_dense * h, double * dG, int ldg, const double * da, const double * dC, int ldc, const double * dbl,
                     const double * dbu, const double * dxl, const double * dxu)
{
  if(!h || !dG || !da || ldg < h->n) return JRLQP_ERR_INVALID_ARGUMENT;
  if(h->m > 0 && (!dC || !dbl || !dbu || ldc < h->n)) return JRLQP_ERR_INVALID_ARGUMENT;
  if(h->nb > 0 && (!dxl || !dxu)) return JRLQP_ERR_INVALID_ARGUMENT;
  DenseViews & v = h->v;
  v.n = h->n; v.m = h->m; v.nb = h->nb;
  v.G = dG; v.ldg = ldg; v.a = da; v.C = dC; v.ldc = ldc; v.bl = dbl; v.bu = dbu; v.xl = dxl; v.xu = dxu;
  h->bound = true;
  return JRLQP_OK;
}

// init_ (GoldfarbIdnaniSolver.cpp:56-74) in two halves: the factorisation, which consumes G, and everything that can be redone from
// the factor (J = L^-T, x0, f0, the empty active set) -- a solve that has to be re-run (dense_solve_impl) only repeats the second.
static int dense_init_llt(jrlqp_dense * h, int * info_out, hipStream_t st)
{
  const DenseViews & v = h->v;
  DenseState & s = h->s;
  TRY(hipMemsetAsync(s.info, 0xff, sizeof(int), st)); // -1
  TRY(dense_llt(v.G, v.ldg, v.n, s.info, st));
  int info = -1;
  TRY(hipMemcpyAsync(&info, s.info, sizeof(int), hipMemcpyDeviceToHost, st));
  TRY(hipStreamSynchronize(st));
  *info_out = info;
  return JRLQP_OK;
}
static int dense_init_state(jrlqp_dense * h, hipStream_t st)
{
  const DenseViews & v = h->v;
  DenseState & s = h->s;
  TRY(hipMemsetAsync(s.st, 0, (size_t)v.m + v.nb, st));
  TRY(hipMemsetAsync(s.u, 0, sizeof(double) * ((size_t)v.m + v.nb + 2), st));
  TRY(hipMemsetAsync(s.x, 0, sizeof(double) * v.n, st));
  std::fill(h->status.begin(), h->status.end(), 0);
  h->active.clear();
  h->f = 0.0;
  return JRLQP_OK;
}
static int dense_init_point(jrlqp_dense * h, hipStream_t st)
{
  const DenseViews & v = h->v;
  DenseState & s = h->s;
  // J = L^-T and x0 = -G^-1 a both need the factor and nothing of each other: x0 (one workgroup, two substitutions of n steps:
  // 0.5 ms at n = 1024) runs on the side stream beside the n / 32 launches of J (1.5 ms) -- one event edge per solve
  TRY(hipEventRecord(h->ev_x, st)); // (the factor, and the zeroed x, are complete here)
  TRY(hipStreamWaitEvent(h->side, h->ev_x, 0));
  TRY(dense_x0(v.G, v.ldg, v.n, v.a, s.x, &s.mb->f0, h->side));
  TRY(dense_jinv(v.G, v.ldg, s.J, v.n, st));
  TRY(hipStreamSynchronize(st));
  TRY(hipStreamSynchronize(h->side));
  h->f = h->h_mb->f0;
  return JRLQP_OK;
}

// init_ (GoldfarbIdnaniSolver.cpp:56-74): LLT, J = L^-T, x0, f0, empty active set.  Synchronises the stream.
int jrlqp_dense_init(jrlqp_dense * h, int * llt_info, double * f0, void * stream)
{
  if(!h || !h->bound) return JRLQP_ERR_INVALID_ARGUMENT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = dense_init_state(h, st);
  if(rc != JRLQP_OK) return rc;
  int info = -1;
  rc = dense_init_llt(h, &info, st);
  if(rc != JRLQP_OK) return rc;
  if(llt_info) *llt_info = info;
  if(info >= 0)
  {
    if(f0) *f0 = 0.0;
    return JRLQP_OK;
  }
  rc = dense_init_point(h, st);
  if(rc != JRLQP_OK) return rc;
  if(f0) *f0 = h->f;
  return JRLQP_OK;
}

// selectViolatedConstraint_ (GoldfarbIdnaniSolver.cpp:84-134).  p = -1 / status INACTIVE when none is violated.  Also u[q] = 0 for
// the q the handle's active-set mirror holds (DualSolver.cpp:119-121: the loop re-seats u to q + 1 entries and clears the new one
// right behind the selection; u lives on the device, so the hook does it).
int jrlqp_dense_select(jrlqp_dense * h, int * p, int * status, void * stream)
{
  if(!h || !h->bound || !p || !status) return JRLQP_ERR_INVALID_ARGUMENT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TRY(dense_select(h->v, h->s, -1, st, (int)h->active.size()));
  TRY(hipStreamSynchronize(st));
  *p = h->h_mb->p;
  *status = h->h_mb->pstatus;
  return JRLQP_OK;
}

// computeStep_ + the device part of computeStepLength_ and dot_ (GoldfarbIdnaniSolver.cpp:136-219, :289-293):
// leaves d, z, r on the device and returns t1, l, z.z, n.z, n.x and u[q] (n = C[:,p] or e_{p-m}, unsigned).
int jrlqp_dense_step(jrlqp_dense * h, int p, int status, int q, double big_bnd, double * t1, int * l, double * zz, double * nz,
                     double * nx, double * uq, void * stream)
{
  if(!h || !h->bound || p < 0 || p >= h->m + h->nb || q < 0 || q > h->n) return JRLQP_ERR_INVALID_ARGUMENT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TRY(dense_step(h->v, h->s, p, status, q, big_bnd, st));
  TRY(hipStreamSynchronize(st));
  const Mailbox & mb = *h->h_mb;
  if(t1) *t1 = mb.t1;
  if(l) *l = mb.l;
  if(zz) *zz = mb.zz;
  if(nz) *nz = mb.cz;
  if(nx) *nx = mb.cx;
  if(uq) *uq = mb.uq;
  return JRLQP_OK;
}

// x += t z (if primal), u[0:q] -= t r, u[q] += t   (DualSolver.cpp:136-137, :146-150)
int jrlqp_dense_advance(jrlqp_dense * h, double t, int q, int primal, void * stream)
{
  if(!h || !h->bound) return JRLQP_ERR_INVALID_ARGUMENT;
  TRY(dense_advance(h->v, h->s, t, q, primal != 0, static_cast<hipStream_t>(stream)));
  return JRLQP_OK;
}

// addConstraint_ (GoldfarbIdnaniSolver.cpp:221-237); q = number of active constraints BEFORE the addition
int jrlqp_dense_add(jrlqp_dense * h, int p, int status, int q, void * stream)
{
  if(!h || !h->bound || q >= h->n || q != (int)h->active.size() || p < 0 || p >= h->m + h->nb) return JRLQP_ERR_INVALID_ARGUMENT;
  TRY(dense_add(h->v, h->s, p, status, q, static_cast<hipStream_t>(stream)));
  h->active.push_back(p); // A_.activate (DualSolver.cpp:231-235; ActiveSet.cpp:87-131): the handle's mirror, read by jrlqp_dense_results
  h->status[p] = status;
  return JRLQP_OK;
}

// removeConstraint (DualSolver.cpp:237-244) + removeConstraint_ (GoldfarbIdnaniSolver.cpp:239-256); q = number of
// active constraints BEFORE the removal, `index` = constraint index of the l-th active constraint
int jrlqp_dense_remove(jrlqp_dense * h, int l, int q, int index, void * stream)
{
  if(!h || !h->bound || l < 0 || l >= q || q > JRLQP_DENSE_MAX_N || q != (int)h->active.size() || h->active[l] != index)
    return JRLQP_ERR_INVALID_ARGUMENT;
  TRY(dense_remove(h->v, h->s, l, q, index, static_cast<hipStream_t>(stream)));
  h->status[index] = 0; // A_.deactivate(l) (DualSolver.cpp:237-244; ActiveSet.cpp:132-168)
  h->active.erase(h->active.begin() + l);
  return JRLQP_OK;
}

// initActiveSet (GoldfarbIdnaniSolver.cpp:268-287) -- the second half of the reference's init_ hook (:75-79): every data equality
// (bl_i == bu_i), then every fixed variable (xl_i == xu_i), by increasing index, each through addInitialConstraint (:295-338: step
// direction, full step, no ratio test, addConstraint).  hbl .. hxu: the bounds on the HOST (m, m, nb, nb entries).  *term =
// JRLQP_SUCCESS, or JRLQP_OVERCONSTRAINED_PROBLEM when there are more of them than unknowns (UB in the reference); *f is updated.
static int dense_init_active_set(jrlqp_dense * h, const double * hbl, const double * hbu, const double * hxl, const double * hxu,
                                 double big, int * term, void * stream)
{
  enum { EQUALITY = 3, FIXED = 6 };
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int n = h->n, m = h->m, nb = h->nb;
  int neq = 0;
  for(int i = 0; i < m; ++i) neq += (hbl[i] == hbu[i]);
  for(int i = 0; i < nb; ++i) neq += (hxl[i] == hxu[i]);
  *term = JRLQP_SUCCESS;
  if(neq > n)
  {
    *term = JRLQP_OVERCONSTRAINED_PROBLEM;
    return JRLQP_OK;
  }
  for(int i = 0; i < m + nb; ++i)
  {
    const bool eq = (i < m) ? (hbl[i] == hbu[i]) : (hxl[i - m] == hxu[i - m]);
    if(!eq) continue;
    const int status = (i < m) ? EQUALITY : FIXED;
    const int q = (int)h->active.size();
    TRY(hipMemsetAsync(h->s.u + q, 0, sizeof(double), st)); // u[q] = 0
    double t1, zz, nz, nx, uq;
    int l;
    int rc = jrlqp_dense_step(h, i, status, q, big, &t1, &l, &zz, &nz, &nx, &uq, stream);
    if(rc != JRLQP_OK) return rc;
    double t = 0.0;
    if(std::sqrt(zz) > 1e-14) t = (((i < m) ? hbl[i] : hxl[i - m]) - nx) / nz; // addInitialConstraint :309-322
    rc = jrlqp_dense_advance(h, t, q, 1, stream);
    if(rc != JRLQP_OK) return rc;
    h->f = std::fma(t * nz, 0.5 * t + uq, h->f);
    rc = jrlqp_dense_add(h, i, status, q, stream);
    if(rc != JRLQP_OK) return rc;
  }
  return JRLQP_OK;
}

int jrlqp_dense_init_active_set(jrlqp_dense * h, double big_bnd, int * term, int * nactive, int * active_host, int * status_host,
                                double * f, void * stream)
{
  if(!h || !h->bound || !term) return JRLQP_ERR_INVALID_ARGUMENT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const DenseViews & v = h->v;
  if(h->m) TRY(hipMemcpyAsync(h->h_bl.data(), v.bl, sizeof(double) * h->m, hipMemcpyDeviceToHost, st));
  if(h->m) TRY(hipMemcpyAsync(h->h_bu.data(), v.bu, sizeof(double) * h->m, hipMemcpyDeviceToHost, st));
  if(h->nb) TRY(hipMemcpyAsync(h->h_xl.data(), v.xl, sizeof(double) * h->nb, hipMemcpyDeviceToHost, st));
  if(h->nb) TRY(hipMemcpyAsync(h->h_xu.data(), v.xu, sizeof(double) * h->nb, hipMemcpyDeviceToHost, st));
  TRY(hipStreamSynchronize(st));
  const int rc = dense_init_active_set(h, h->h_bl.data(), h->h_bu.data(), h->h_xl.data(), h->h_xu.data(), big_bnd, term, stream);
  if(rc != JRLQP_OK) return rc;
  const int q = (int)h->active.size();
  if(nactive) *nactive = q;
  for(int k = 0; k < q; ++k)
  {
    if(active_host) active_host[k] = h->active[k];
    if(status_host) status_host[k] = h->status[h->active[k]];
  }
  if(f) *f = h->f;
  return JRLQP_OK;
}

int jrlqp_dense_trace_vectors_stride(const jrlqp_dense * h)
{
  return h ? 2 * h->n + 2 * (h->m + h->nb + 1) + 2 : -1;
}

int jrlqp_dense_set_trace_vectors(jrlqp_dense * h, double * vectors_host, int cap)
{
  if(!h || cap < 0 || (cap > 0 && !vectors_host)) return JRLQP_ERR_INVALID_ARGUMENT;
  h->vtrace = cap ? vectors_host : nullptr;
  h->vtrace_cap = cap;
  return JRLQP_OK;
}

int jrlqp_dense_set_trace(jrlqp_dense * h, jrlqp_trace_rec * trace_host, int trace_cap)
{
  if(!h || trace_cap < 0 || (trace_cap > 0 && !trace_host)) return JRLQP_ERR_INVALID_ARGUMENT;
  h->trace = trace_cap ? trace_host : nullptr;
  h->trace_cap = trace_cap;
  return JRLQP_OK;
}

int jrlqp_dense_sync_mode(const jrlqp_dense * h, int * event_reruns)
{
  if(!h) return -1;
  if(event_reruns) *event_reruns = h->event_reruns;
  return h->use_events ? 1 : 0;
}

int jrlqp_debug_dense_inject_sync_fault(jrlqp_dense * h, int pass)
{
  if(!h) return JRLQP_ERR_INVALID_ARGUMENT;
  h->inject_fault_pass = pass;
  return JRLQP_OK;
}

// ---- the complete solve: DualSolver::solve (src/DualSolver.cpp:91-168) on the host ----------------------------
static int dense_solve_attempts(jrlqp_dense * h, double * dG, int ldg, const double * da, const double * dC, int ldc, const double * dbl,
                                const double * dbu, const double * dxl, const double * dxu, const jrlqp_options * opts, int * term_out,
                                void * stream, bool warm, const int * as_host)
{
  int rc = jrlqp_dense_bind(h, dG, ldg, da, dC, ldc, dbl, dbu, dxl, dxu);
  if(rc != JRLQP_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  jrlqp_options o;
  jrlqp_default_options(&o);
  if(opts) o = *opts;
  const int n = h->n, m = h->m, nb = h->nb;
  const double big = o.big_bnd;
  enum { INACTIVE = 0, LOWER, UPPER, EQUALITY, LOWER_BOUND, UPPER_BOUND, FIXED };

  // bounds are needed on the host for t2 and for the equality detection
  if(m) TRY(hipMemcpyAsync(h->h_bl.data(), dbl, sizeof(double) * m, hipMemcpyDeviceToHost, st));
  if(m) TRY(hipMemcpyAsync(h->h_bu.data(), dbu, sizeof(double) * m, hipMemcpyDeviceToHost, st));
  if(nb) TRY(hipMemcpyAsync(h->h_xl.data(), dxl, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
  if(nb) TRY(hipMemcpyAsync(h->h_xu.data(), dxu, sizeof(double) * nb, hipMemcpyDeviceToHost, st));

  // kernels of the previous solve may still be in flight on the side streams (a discarded coefficient chain reading d, the z of a
  // final pass whose step found nothing to do): they ended long ago -- these calls return at once -- but nothing of this solve may
  // race with them
  TRY(hipStreamSynchronize(h->spec));
  TRY(hipStreamSynchronize(h->side));
  h->chain_pending = 0;
  h->h_mb->sync_err = 0;
  auto finish = [&](int term) {
    h->term = term;
    if(term_out) *term_out = term;
    (void)hipStreamSynchronize(h->spec); // nothing this solve enqueued outlives it (the exported hooks and the next solve write the
    (void)hipStreamSynchronize(h->side); // same buffers from the caller's stream)
    return JRLQP_OK;
  };
  // A device-side wait between the streams that gave up (PassSync, gi_dense.h: a consumer that did not see its producer within
  // ~1 s -- CUs held by someone else's kernels, several handles spinning at once) voids the pass, not the problem: the streams
  // are drained, the handle switches to event edges for good, and the solve is repeated from the factor (G holds L: the
  // factorisation is the one step that cannot be redone).  JRLQP_ERR_HIP is left for a rerun that fails too.
  bool rerun = false;
  int info = -1;
attempt:
  h->it = 0;
  h->term = JRLQP_UNKNOWN;
  if(!warm)
  {
    rc = dense_init_state(h, st);
    if(rc != JRLQP_OK) return rc;
    if(!rerun)
    {
      rc = dense_init_llt(h, &info, st);
      if(rc != JRLQP_OK) return rc;
      if(info >= 0) return finish(JRLQP_NON_POS_HESSIAN);
    }
    rc = dense_init_point(h, st);
    if(rc != JRLQP_OK) return rc;
    if(m + nb == 0) return finish(JRLQP_SUCCESS); // no constraint: x0 is the solution (and no kernel of a pass has anything to wait for)
  }

  auto bound_value = [&](int p, int status) -> double {
    switch(status)
    {
      case LOWER: case EQUALITY: return h->h_bl[p];
      case UPPER: return h->h_bu[p];
      case LOWER_BOUND: case FIXED: return h->h_xl[p - m];
      default: return h->h_xu[p - m];
    }
  };
  auto activate = [&](int p, int status) {
    h->active.push_back(p);
    h->status[p] = status;
  };

  if(warm)
  {
    // ==== experimental::GoldfarbIdnaniSolver::init_ (src/experimental/GoldfarbIdnaniSolver.cpp:65-112; oracle gi_solve_impl warm == 1)
    TRY(hipStreamSynchronize(st)); // the bounds are on the host
    DenseState & s = h->s;
    const int mt = m + nb;
    // processInitialActiveSet (:309-386): bounds first, then general constraints, each by increasing index; data equalities win over
    // the guess, guesses on infinite bounds / FIXED guesses on free variables / ill-typed entries are ignored
    std::fill(h->status.begin(), h->status.end(), 0);
    h->active.clear();
    h->f = 0.0;
    int neq = 0;
    for(int i = 0; i < nb; ++i)
    {
      const int bi = m + i;
      if(h->h_xl[i] == h->h_xu[i]) { activate(bi, FIXED); ++neq; }
      else if(as_host && as_host[bi] != INACTIVE)
      {
        const int g = as_host[bi];
        if(g == LOWER_BOUND && !(h->h_xl[i] < -big)) activate(bi, g);
        if(g == UPPER_BOUND && !(h->h_xu[i] > big)) activate(bi, g);
      }
    }
    for(int i = 0; i < m; ++i)
    {
      if(h->h_bl[i] == h->h_bu[i]) { activate(i, EQUALITY); ++neq; }
      else if(as_host && as_host[i] != INACTIVE)
      {
        const int g = as_host[i];
        if(g == LOWER && !(h->h_bl[i] < -big)) activate(i, g);
        if(g == UPPER && !(h->h_bu[i] > big)) activate(i, g);
        if(g == EQUALITY) { activate(i, g); ++neq; } // (:357-371 only filters FIXED: an EQUALITY guess is taken as is)
      }
    }
    TRY(hipMemsetAsync(s.u, 0, sizeof(double) * ((size_t)mt + 2), st));
    TRY(hipMemsetAsync(s.x, 0, sizeof(double) * n, st));
    if((int)h->active.size() > n)
    {
      if(neq > n)
      {
        h->active.clear(); // (the statuses stay, as in the reference; the multipliers are all zero)
        return finish(JRLQP_OVERCONSTRAINED_PROBLEM);
      }
      // work backward, deactivating inequality constraints until n are left (:373-383)
      int i = (int)h->active.size();
      while((int)h->active.size() > n)
      {
        --i;
        for(;;)
        {
          const int sv = h->status[h->active[i]];
          if(sv == EQUALITY || sv == FIXED) --i; else break;
        }
        h->status[h->active[i]] = INACTIVE;
        h->active.erase(h->active.begin() + i);
      }
    }
    // the factorisation (:66-71)
    if(!rerun)
    {
      rc = dense_init_llt(h, &info, st);
      if(rc != JRLQP_OK) return rc;
    }
    if(info >= 0)
    {
      h->active.clear();
      return finish(JRLQP_NON_POS_HESSIAN);
    }
    int q = (int)h->active.size();
    // status bytes, active list and b_act to the device; N -> B = L^-1 N -> QR, J = L^-T Q (initializeComputationData, :388-458)
    for(int i = 0; i < mt; ++i) h->h_stb[i] = (unsigned char)h->status[i];
    if(mt) TRY(hipMemcpyAsync(s.st, h->h_stb.data(), (size_t)mt, hipMemcpyHostToDevice, st));
    if(q) TRY(hipMemcpyAsync(s.wact, h->active.data(), sizeof(int) * q, hipMemcpyHostToDevice, st));
    h->h_bact.clear();
    for(int k = 0; k < q; ++k)
    {
      const int idx = h->active[k], sv = h->status[idx];
      h->h_bact.push_back((sv == UPPER || sv == UPPER_BOUND) ? -bound_value(idx, sv) : bound_value(idx, sv));
    }
    TRY(dense_warm_factor(h->v, s, q, st));
    // initializePrimalDualPoints + the purge of the negative multipliers (:78-109), each purge one iteration
    for(;;)
    {
      TRY(dense_warm_point(h->v, s, q, st));
      if(q) TRY(hipMemcpyAsync(h->h_u.data(), s.u, sizeof(double) * q, hipMemcpyDeviceToHost, st));
      TRY(hipStreamSynchronize(st));
      h->f = h->h_mb->f0;
      double umin = -1e-14;
      int lmin = -1;
      for(int l = 0; l < q; ++l)
      {
        const int sv = h->status[h->active[l]];
        if(h->h_u[l] < umin && sv != FIXED && sv != EQUALITY)
        {
          umin = h->h_u[l];
          lmin = l;
        }
      }
      if(lmin < 0) break;
      ++h->it;
      h->h_bact.erase(h->h_bact.begin() + lmin);
      rc = jrlqp_dense_remove(h, lmin, q, h->active[lmin], stream); // (its shift of u is harmless: u is recomputed; it also updates the mirror)
      if(rc != JRLQP_OK) return rc;
      --q;
      if(q) TRY(hipMemcpyAsync(s.bact, h->h_bact.data(), sizeof(double) * q, hipMemcpyHostToDevice, st));
    }
    TRY(hipMemsetAsync(s.u + q, 0, sizeof(double), st)); // u[q] = 0
  }
  else
  // initActiveSet (GoldfarbIdnaniSolver.cpp:268-287)
  {
    int t0 = JRLQP_SUCCESS;
    rc = dense_init_active_set(h, h->h_bl.data(), h->h_bu.data(), h->h_xl.data(), h->h_xu.data(), big, &t0, stream);
    if(rc != JRLQP_OK) return rc;
    if(t0 != JRLQP_SUCCESS) return finish(t0);
  }

  // optional host-side time line of the loop (JRLQP_DENSE_TIMING=1): where the host spends a pass
  static const bool timing = std::getenv("JRLQP_DENSE_TIMING") != nullptr;
  double t_launch_step = 0, t_wait = 0, t_rest = 0, t_mark = 0;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  struct Report
  {
    const bool & on;
    double &a, &b, &c;
    const int & it;
    ~Report()
    {
      if(on) std::fprintf(stderr, "[jrlqp dense] %d passes: launch step %.1f us, wait %.1f us, decide+launch rest %.1f us per pass\n", it, a / (it + 1), b / (it + 1), c / (it + 1));
    }
  } report{timing, t_launch_step, t_wait, t_rest, h->it};
  bool skip = false;
  bool scan_ahead = false; // ... and so is the next pass's constraint scan
  bool speculated = false; // the sweep of this pass's possible addition is already in the stream (dense_add_speculative)
  bool prefetched = false; // the scan of this pass already ran on the side stream, beside the previous pass's sweep
  int p = -1, pstat = INACTIVE;
  for(; h->it < o.max_iter; ++h->it)
  {
    const int q = (int)h->active.size();
    if(timing && t_mark > 0) t_rest += now() - t_mark;
    // ONE host round trip per pass: the selection (on the side stream beside the previous sweep, or here) leaves
    // (p, status) on the device, the step kernels read it there, and the host sees p together with the step's scalars
    const int tag = ++h->tag;
    int wait_sel = 0;
    if(!skip)
    {
      if(prefetched)
      {
        if(h->use_events)
          TRY(hipStreamWaitEvent(st, h->ev_sel, 0));
        else
          wait_sel = tag; // the first kernel of the step waits for the side stream's selection (PassSync::sel_tag)
        prefetched = false;
      }
      else
        TRY(dense_select(h->v, h->s, -1, st, q)); // (also u[q] = 0, DualSolver.cpp:121)
    }
    double t1, zz, nz, nx, uq;
    int l;
    {
      // computeStep_ with the coefficient chain of a possible addition started on the side as soon as d exists: most
      // passes add, and the chain (n - q dependent radius steps) is as long as z, r and the ratio test together.  A
      // chain that is still reading d must have ended before d is overwritten: the kernel that writes d waits for its tag.
      const bool spec = q < n;
      const double ta = timing ? now() : 0;
      if(h->use_events)
      {
        if(h->chain_pending) TRY(hipStreamWaitEvent(st, h->ev_chain, 0)); // a discarded chain may still be reading d
        TRY(dense_step(h->v, h->s, kSelOnDevice, 0, q, big, st, spec ? h->spec : nullptr, tag, 0, 0, nullptr, h->ev_d, h->ev_chain));
      }
      else
      {
        // (fault injection, tests only: this pass's first kernel waits for a chain tag nobody will ever post)
        const int wait_chain = (h->inject_fault_pass == h->it && !rerun) ? 0x7ffffff0 : h->chain_pending;
        TRY(dense_step(h->v, h->s, kSelOnDevice, 0, q, big, st, spec ? h->spec : nullptr, tag, wait_sel, wait_chain, h->side));
      }
      h->chain_pending = spec ? tag : 0;
      // The sweep of an addition -- most passes add -- goes into the stream NOW, behind the step kernel and ahead of the decision:
      // it runs only if the device's own decision for this pass (StepDev) is "add", and then also advances x and u.  The GPU no
      // longer idles while the scalars travel to the host and the launches travel back.
      speculated = h->speculate && !h->use_events && spec && q < n - 1 && !h->vtrace; // (a vector trace reads x and u of THIS pass: nothing may run ahead)
      if(speculated)
      {
        const bool more = h->it + 1 < o.max_iter;
        TRY(dense_add_speculative(h->v, h->s, q, st, tag, tag, more ? tag : 0));
        // ... and, with the blocked scan (more than 256 unknowns), the next pass's constraint scan on the side stream behind a one-wave
        // gate that waits for the decision: the scan no longer starts a host round trip after the step (it is what the next d waits
        // for).  (First version, round 4: every workgroup of the scan polling the decision itself -- 0.8 ms slower than no scan ahead.)
        scan_ahead = more && h->speculate_scan && n > 256 && m > 0;
        if(scan_ahead) TRY(dense_select(h->v, h->s, -1, h->side, q + 1, tag, tag + 1, tag));
      }
      const double tb = timing ? now() : 0;
      if(!wait_tag(h->h_mb, tag)) TRY(hipStreamSynchronize(st));
      if(timing)
      {
        const double tc = now();
        t_launch_step += tb - ta;
        t_wait += tc - tb;
        t_mark = tc; // (the rest of the pass is closed at the top of the next one)
      }
      const Mailbox & mb = *h->h_mb;
      if(mb.sync_err) // a device-side wait between the streams gave up (gi_dense.h, PassSync): nothing of this attempt can be trusted
      {
        TRY(hipStreamSynchronize(st)); // (every kernel runs on after its wait has given up: the streams drain)
        TRY(hipStreamSynchronize(h->spec));
        TRY(hipStreamSynchronize(h->side));
        h->chain_pending = 0;
        h->h_mb->sync_err = 0;
        if(h->use_events || rerun)
        {
          h->last_hip = (int)hipErrorLaunchTimeOut;
          return JRLQP_ERR_HIP;
        }
        h->use_events = true; // for good: whatever starved the hand-off may still be there for the next solve
        ++h->event_reruns;
        rerun = true;
        goto attempt;
      }
      if(!skip)
      {
        p = mb.sp; // (the step kernel's echo of the selection it worked on: mb.p may already hold the NEXT pass's, see Mailbox)
        pstat = mb.sps;
        if(p < 0 || pstat == INACTIVE) return finish(JRLQP_SUCCESS);
      }
      t1 = mb.t1; l = mb.l; zz = mb.zz; nz = mb.cz; nx = mb.cx; uq = mb.uq;
    }
    double t2 = big;
    if(std::sqrt(zz) > 1e-14) t2 = (bound_value(p, pstat) - nx) / nz;
    const double t = (t2 < t1) ? t2 : t1;
    if(h->trace && h->it < h->trace_cap) // the reference logs every pass (DualSolver.cpp:116,128-129,138,153,159)
      h->trace[h->it] = jrlqp_trace_rec{p, pstat, l, (t2 >= big) ? 2 : ((t == t2) ? 0 : 1), t1, t2};
    if(h->vtrace && h->it < h->vtrace_cap)
    {
      // the vectors the reference logs under ITERATION_BASIC_DETAILS: x, u, f at the top of the pass (DualSolver.cpp:108) and z, r
      // behind the step computation (:128-129).  Nothing of this pass has been applied yet (no speculation while this trace is on).
      const int mt1 = m + nb + 1, stride = 2 * n + 2 * mt1 + 2;
      double * rec = h->vtrace + (size_t)h->it * stride;
      TRY(hipStreamSynchronize(st));
      TRY(hipMemcpy(rec, h->s.x, sizeof(double) * n, hipMemcpyDeviceToHost));
      std::fill(rec + n, rec + n + mt1, 0.0);
      TRY(hipMemcpy(rec + n, h->s.u, sizeof(double) * (q + 1), hipMemcpyDeviceToHost));
      TRY(hipMemcpy(rec + n + mt1, h->s.z, sizeof(double) * n, hipMemcpyDeviceToHost));
      std::fill(rec + 2 * n + mt1, rec + 2 * n + 2 * mt1, 0.0);
      if(q) TRY(hipMemcpy(rec + 2 * n + mt1, h->s.r, sizeof(double) * q, hipMemcpyDeviceToHost));
      rec[2 * n + 2 * mt1] = h->f;
      rec[2 * n + 2 * mt1 + 1] = (double)q;
    }
    if(speculated)
    {
      // the device took the same decision from the same scalars with the same operations; a disagreement would mean that the sweep in
      // the stream did something the host does not know of
      const int kind = (t >= big) ? kKindInfeasible : ((t2 >= big) ? kKindDual : ((t == t2) ? kKindAdd : kKindDrop));
      if(h->h_mb->dkind != kind)
      {
        // void the attempt like a starved hand-off: drain, switch to event edges (no speculation there), repeat from the factor
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(h->spec);
        (void)hipStreamSynchronize(h->side);
        h->chain_pending = 0;
        if(h->use_events || rerun)
        {
          h->last_hip = (int)hipErrorAssert;
          return JRLQP_ERR_HIP;
        }
        h->use_events = true;
        ++h->event_reruns;
        rerun = true;
        goto attempt;
      }
    }
    if(t >= big) return finish(JRLQP_INFEASIBLE);
    const double dotsc = (pstat == UPPER || pstat == UPPER_BOUND) ? -nz : nz;
    if(t2 >= big)
    {
      rc = jrlqp_dense_advance(h, t, q, 0, stream);
      if(rc != JRLQP_OK) return rc;
      rc = jrlqp_dense_remove(h, l, q, h->active[l], stream); // (also A_.deactivate on the handle's mirror)
      if(rc != JRLQP_OK) return rc;
      skip = true;
    }
    else
    {
      const bool more = h->it + 1 < o.max_iter;
      const bool prefetch = (t == t2) && more;
      const bool swept = speculated && (t == t2); // the sweep in the stream is running: it advances x and u itself
      if(!swept) TRY(dense_advance(h->v, h->s, t, q, true, st, (prefetch && !h->use_events) ? tag : 0)); // (x_tag: the side stream's scan reads the new x)
      h->f = std::fma(t * dotsc, 0.5 * t + uq, h->f);
      if(t == t2)
      {
        // x is final once `advance` has run: the next pass's constraint scan (which only reads C, x, bounds and the
        // status bytes) runs on the side stream while the main stream does the Givens sweep; the constraint being
        // activated is excluded explicitly because its status byte is written by the sweep's first kernel
        // (the sweep is on the critical path: it is enqueued first, the side-stream scan behind it -- every launch costs the
        // host a few microseconds)
        if(h->use_events)
        {
          if(prefetch) TRY(hipEventRecord(h->ev_x, st)); // x is final here: the event sits BEFORE the sweep in the main stream
          if(h->chain_pending) TRY(hipStreamWaitEvent(st, h->ev_chain, 0));
          TRY(dense_add(h->v, h->s, p, pstat, q, st, h->chain_pending ? -1 : 0)); // (-1: the chain ran ahead, ordered by the event)
        }
        else if(!swept)
          TRY(dense_add(h->v, h->s, p, pstat, q, st, h->chain_pending)); // commit: the chain ran ahead (the sweep waits for chain_tag)
        h->chain_pending = 0;
        if(prefetch && swept && scan_ahead)
          prefetched = true; // (the scan went into the side stream with the sweep, ahead of the decision)
        else if(prefetch)
        {
          // (also u[q+1] = 0 for the next pass: a slot this pass does not touch)
          if(h->use_events)
          {
            TRY(hipStreamWaitEvent(h->side, h->ev_x, 0));
            TRY(dense_select(h->v, h->s, p, h->side, q + 1));
            TRY(hipEventRecord(h->ev_sel, h->side));
          }
          else
            TRY(dense_select(h->v, h->s, p, h->side, q + 1, tag, tag + 1));
          prefetched = true;
        }
        activate(p, pstat);
        skip = false;
      }
      else
      {
        rc = jrlqp_dense_remove(h, l, q, h->active[l], stream);
        if(rc != JRLQP_OK) return rc;
        skip = true;
      }
    }
  }
  return finish(JRLQP_MAX_ITER_REACHED);
}

// Every exit of the loop above that is NOT a termination status (a HIP error, a starved rerun that failed too) leaves through here:
// the caller's stream and the handle's two side streams are drained and the pending-chain flag is cleared, exactly as finish() does
// for a regular end -- a speculative sweep or a gated scan may still be running on x / u / J / psel, and the exported hooks,
// jrlqp_dense_results and the next solve touch the same buffers.
static int dense_solve_impl(jrlqp_dense * h, double * dG, int ldg, const double * da, const double * dC, int ldc, const double * dbl,
                            const double * dbu, const double * dxl, const double * dxu, const jrlqp_options * opts, int * term_out,
                            void * stream, bool warm, const int * as_host)
{
  const int rc = dense_solve_attempts(h, dG, ldg, da, dC, ldc, dbl, dbu, dxl, dxu, opts, term_out, stream, warm, as_host);
  if(rc != JRLQP_OK && h)
  {
    (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
    (void)hipStreamSynchronize(h->spec);
    (void)hipStreamSynchronize(h->side);
    h->chain_pending = 0;
  }
  return rc;
}

int jrlqp_dense_solve(jrlqp_dense * h, double * dG, int ldg, const double * da, const double * dC, int ldc, const double * dbl,
                      const double * dbu, const double * dxl, const double * dxu, const jrlqp_options * opts, int * term_out,
                      void * stream)
{
  return dense_solve_impl(h, dG, ldg, da, dC, ldc, dbl, dbu, dxl, dxu, opts, term_out, stream, false, nullptr);
}

int jrlqp_dense_solve_warm(jrlqp_dense * h, double * dG, int ldg, const double * da, const double * dC, int ldc, const double * dbl,
                           const double * dbu, const double * dxl, const double * dxu, const int * as_host, const jrlqp_options * opts,
                           int * term_out, void * stream)
{
  return dense_solve_impl(h, dG, ldg, da, dC, ldc, dbl, dbu, dxl, dxu, opts, term_out, stream, true, as_host);
}

// solution(), multipliers() (sign-expanded, DualSolver.cpp:38-69), activeSet(), iterations(), objectiveValue()
int jrlqp_dense_results(jrlqp_dense * h, double * x_host, double * u_host, int * status_host, int * iters, double * f, void * stream)
{
  if(!h) return JRLQP_ERR_INVALID_ARGUMENT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int q = (int)h->active.size();
  if(x_host) TRY(hipMemcpyAsync(x_host, h->s.x, sizeof(double) * h->n, hipMemcpyDeviceToHost, st));
  if(u_host && q) TRY(hipMemcpyAsync(h->h_u.data(), h->s.u, sizeof(double) * q, hipMemcpyDeviceToHost, st));
  TRY(hipStreamSynchronize(st));
  const int mt = h->m + h->nb;
  if(u_host)
  {
    for(int i = 0; i < mt; ++i) u_host[i] = 0.0;
    for(int k = 0; k < q; ++k)
    {
      const int i = h->active[k];
      const int s = h->status[i];
      u_host[i] = (s == 2 || s == 5) ? h->h_u[k] : -h->h_u[k];
    }
  }
  if(status_host) for(int i = 0; i < mt; ++i) status_host[i] = h->status[i];
  if(iters) *iters = h->it;
  if(f) *f = h->f;
  return JRLQP_OK;
}

} // extern "C"
