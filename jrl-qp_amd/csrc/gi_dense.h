// gi_dense.h -- Mode A (one large dense QP, host-driven hooks): shared types and launchers (internal)
#pragma once
#include <hip/hip_runtime.h>

namespace jrlqp
{
// host-visible (pinned, mapped) scalar hand-off between the hook kernels and the host state machine
struct Mailbox
{
  int p, pstatus;       // selectViolatedConstraint_
  int l, pad;           // computeStepLength_
  double t1, zz, cz, cx; // t1, z.z, n.z, n.x
  double uq;            // u[q] before the update (needed by the f update on the host)
  double f0;            // init_
  int llt_info, pad2;
  // pass tags written LAST by the two blocks of the step kernel (after a system-scope fence): the host state machine polls
  // them instead of synchronising the stream -- a stream synchronisation wakes up tens of microseconds after the kernel ends
  volatile int seq0, seq1;
};

// the caller's problem (device pointers; GoldfarbIdnaniSolver::Problem, GoldfarbIdnaniSolver.h:35-58)
struct DenseViews
{
  int n, m, nb;
  double * G;
  int ldg;
  const double * a;
  const double * C;
  int ldc;
  const double *bl, *bu, *xl, *xu;
};

// solver-owned device state (work_J_, work_R_, work_d_, work_x_, work_z_, work_u_, work_r_; DualSolver.h:128-141)
struct DenseState
{
  double *J, *R, *d, *z, *x, *u, *r, *cc, *ss;
  double *cc2, *ss2, *rfin; // output of a coefficient chain run ahead of the add/drop decision (its own buffers: a drop
                            // writes cc/ss on the main stream while a discarded chain may still be running)
  unsigned char * st; // activation status per constraint / bound
  double * pval;      // select partials
  int *pidx, *pstat;
  int * info;         // LLT info (device)
  int * psel;         // {p, status} of the last selection (device copy of the mailbox entries)
  int * wact;         // warm start: the initial active list (n ints)
  double *bact, *hco, *beta; // warm start: b_act, Householder coefficients, beta = R^-T b_act (n doubles each)
  Mailbox * mb;       // device view of the mailbox
};

constexpr int kSelOnDevice = -2; // dense_step: take (p, status) from DenseState::psel instead of the arguments

hipError_t dense_llt(double * G, int ldg, int n, int * d_info, hipStream_t st);
hipError_t dense_jinv(const double * L, int ldl, double * J, int n, hipStream_t st);
hipError_t dense_x0(const double * L, int ldl, int n, const double * a, double * x, double * f0, hipStream_t st);
// zero_u >= 0: also clears u[zero_u] (the multiplier slot of the constraint this selection is for; saves a memset launch)
hipError_t dense_select(const DenseViews & v, DenseState & s, int excl, hipStream_t st, int zero_u = -1);
// spec != nullptr: the Givens coefficient chain of an ADDITION is launched on `spec` as soon as d exists (ev_d), beside
// z / r / the ratio test on `st`; ev_chain marks its end.  dense_add(..., chain_ready = true) then only commits.
hipError_t dense_step(const DenseViews & v, DenseState & s, int p, int pstat, int q, double big, hipStream_t st,
                      hipStream_t spec = nullptr, hipEvent_t ev_d = nullptr, hipEvent_t ev_chain = nullptr, int tag = 0);
hipError_t dense_advance(const DenseViews & v, DenseState & s, double t, int q, bool primal, hipStream_t st);
hipError_t dense_add(const DenseViews & v, DenseState & s, int p, int pstat, int q, hipStream_t st, bool chain_ready = false,
                     hipEvent_t ev_chain = nullptr);
hipError_t dense_remove(const DenseViews & v, DenseState & s, int l, int q, int gone, hipStream_t st);
// warm start (experimental::GoldfarbIdnaniSolver.cpp:388-487): with the LLT in G, the active list in s.wact and the status bytes in
// s.st: N -> B = L^-1 N -> Householder QR (in s.R: R is its upper triangle), J = L^-T Q, b_act -> s.bact
hipError_t dense_warm_factor(const DenseViews & v, DenseState & s, int q, hipStream_t st);
// initializePrimalDualPoints: alpha = J^T a -> s.d, beta -> s.beta, x, u (q entries), f -> mailbox f0
hipError_t dense_warm_point(const DenseViews & v, DenseState & s, int q, hipStream_t st);
} // namespace jrlqp
