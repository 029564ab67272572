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
  int llt_info;
  int sp, sps;           // (p, status) of the pass whose step scalars these are, echoed by the step kernel: the NEXT pass's selection may
                         // already have overwritten p / pstatus above when its scan went into the stream ahead of the decision
  int dkind;             // the device's own decision for this pass (StepDev::kind), echoed for the host's cross-check; -1: none taken
  volatile int sync_err; // a device-side wait of the pass hand-off (PassSync) gave up: the results of this solve are void
  // pass tags written LAST by the two blocks of the step kernel (after a system-scope fence): the host state machine polls
  // them instead of synchronising the stream -- a stream synchronisation wakes up tens of microseconds after the kernel ends
  volatile int seq0, seq1;
};

// Device-side hand-off between the kernels of a pass that run on DIFFERENT streams (the coefficient chain beside z / r / the ratio
// test, the next pass's constraint scan beside the Givens sweep).  A producer publishes the pass tag with release semantics once
// its data are out (several workgroups: the last one to arrive at the counter); a consumer -- always enqueued AFTER its producer
// by the host, on whatever hardware queue its stream maps to, so the producer is never stuck behind it -- spins on the tag with
// acquire loads.  This replaces hipEventRecord / hipStreamWaitEvent pairs: a cross-queue event dependency costs 7-13 us of idle
// GPU per edge on this stack (four edges per pass; kernel trace in profiles/), a tag a fraction of a microsecond.
struct PassSync
{
  int d_cnt, d_tag;   // d = J^T n+ is complete            (jtn / jrow -> add_chain)
  int chain_tag;      // (c_i, s_i), the last radius        (add_chain -> rot_apply / add_commit; -> the next jtn after a drop)
  int x_cnt, x_tag;   // x, u advanced                      (advance -> the prefetched select)
  int sel_tag;        // psel = {p, status} of the NEXT pass (select_final -> jtn)
  int z_cnt, z_tag;   // z = J2 d2 is complete              (z_rows on the side stream -> the scalar products of step_tail)
};

// Device-side copy of a pass's step scalars and the add / drop decision they imply (round 4).  The two workgroups of the step kernel
// leave their scalars here as well as in the mailbox; the one that arrives last forms t2, t = min(t1, t2) and the decision with the
// very operations the host uses (DualSolver.cpp:127-151) and tags it with the pass number.  The sweep of an ADDITION, which also
// advances x and u (DualSolver.cpp:146-150), is enqueued right behind the step kernel, BEFORE the host has seen the scalars, and
// runs only if the decision of its pass is "add": the GPU goes from the ratio test to the sweep without the host round trip
// (8-12 us of an 87 us pass); the host still takes the same decision from the mailbox and drives everything else.
struct StepDev
{
  double t1, zz, cz, cx, uq, t;
  int l, arrive;
  int tag, kind; // kind: 0 full step + add, 1 partial step + drop, 2 dual-only step + drop, 3 infeasible (same coding as jrlqp_trace_rec)
};
constexpr int kKindAdd = 0, kKindDrop = 1, kKindDual = 2, kKindInfeasible = 3, kKindNone = 4; // (4: nothing was selected)

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
  double * Ptri;  // CANON C10: the tile sums of r = S d1 when they do not fit in LDS (nseg (nseg + 1) / 2 tiles of 64)
  double *S, *S2; // CANON C10 (capacity of more than 256 unknowns): S = R^-1 beside R (ld n, like R), and the buffer a removal writes the
                  // new S into (the two swap roles); nullptr otherwise
  double *cc2, *ss2, *rfin; // output of a coefficient chain run ahead of the add/drop decision (its own buffers: a drop
                            // writes cc/ss on the main stream while a discarded chain may still be running)
  unsigned char * st; // activation status per constraint / bound
  double * pval;      // select partials
  int *pidx, *pstat;
  int * info;         // {LLT info, arrival counter of the panel kernel} (device, two ints)
  int * psel;         // {p, status} of the last selection (device copy of the mailbox entries)
  int * wact;         // warm start: the initial active list (n ints)
  double *bact, *hco, *beta; // warm start: b_act, Householder coefficients, beta = R^-T b_act (n doubles each)
  Mailbox * mb;       // device view of the mailbox
  PassSync * sync;    // pass hand-off tags (zeroed at creation; tags only grow)
  StepDev * sd;       // device copy of the step scalars + the decision (zeroed at creation)
  // CANON C8 (more than 256 unknowns): segment sums of the three blocked products and the arrival counters of their column / row groups
  double *partJ, *partZ, *partC; // [segment][column] with leading dimensions ldpJ (>= n, multiple of 64) and ldpC (>= m)
  int ldpJ, ldpC;
  int *cntJ, *cntZ, *cntC;       // zero between launches
};

constexpr int kSelOnDevice = -2; // dense_step: take (p, status) from DenseState::psel instead of the arguments

hipError_t dense_llt(double * G, int ldg, int n, int * d_info, hipStream_t st); // d_info: two ints, d_info[1] zero (gi_dense.hip llt_panel_fused_kernel)
hipError_t dense_jinv(const double * L, int ldl, double * J, int n, hipStream_t st);
hipError_t dense_x0(const double * L, int ldl, int n, const double * a, double * x, double * f0, hipStream_t st);
// zero_u >= 0: also clears u[zero_u] (the multiplier slot of the constraint this selection is for; saves a memset launch).
// wait_x_tag != 0: the scan first waits for PassSync::x_tag == wait_x_tag (it runs on another stream than `advance`);
// post_sel_tag != 0: the selection publishes PassSync::sel_tag = post_sel_tag behind psel.
// gate_tag != 0: the scan goes into the stream AHEAD of the decision of pass gate_tag (StepDev): a one-wave gate kernel waits for the
// decision, the scan kernels behind it run only if that pass adds and exclude the constraint standing in psel (the one being activated)
hipError_t dense_select(const DenseViews & v, DenseState & s, int excl, hipStream_t st, int zero_u = -1, int wait_x_tag = 0, int post_sel_tag = 0,
                        int gate_tag = 0);
// spec != nullptr: the Givens coefficient chain of an ADDITION is launched on `spec` at once and starts as soon as d exists
// (PassSync::d_tag == tag), beside z / r / the ratio test on `st`; it publishes chain_tag = tag.  dense_add(..., wait_chain_tag = tag)
// then only commits.  wait_sel_tag / wait_chain_tag != 0: the first kernel waits for the selection made on the side stream / for a
// discarded chain of an earlier pass that may still be reading d.  zside != nullptr (with a tag): z = J2 d2 runs on `zside`, beside the
// back substitution on `st` (both need d only); the scalar products of the step wait for PassSync::z_tag.
hipError_t dense_step(const DenseViews & v, DenseState & s, int p, int pstat, int q, double big, hipStream_t st,
                      hipStream_t spec = nullptr, int tag = 0, int wait_sel_tag = 0, int wait_chain_tag = 0, hipStream_t zside = nullptr,
                      hipEvent_t ev_d = nullptr, hipEvent_t ev_chain = nullptr);
// ev_d / ev_chain != nullptr: EVENT MODE -- the chain is ordered behind d by events instead of tags and z stays on `st`; no tag is waited
// for or published (for environments in which kernels of different streams do not run side by side: dense_probe_concurrency).
// post_x_tag != 0: publishes PassSync::x_tag = post_x_tag once x and u are advanced
hipError_t dense_advance(const DenseViews & v, DenseState & s, double t, int q, bool primal, hipStream_t st, int post_x_tag = 0);
// wait_chain_tag != 0: the chain already ran on the side (dense_step with spec); the sweep waits for its tag and commits
hipError_t dense_add(const DenseViews & v, DenseState & s, int p, int pstat, int q, hipStream_t st, int wait_chain_tag = 0);
// The same sweep enqueued AHEAD of the decision (tags mode, q < n - 1): it runs only if StepDev says pass `tag` adds, takes p / status
// from psel and t from StepDev, advances x and u itself (no advance kernel) and publishes x_tag = post_x_tag when asked to
hipError_t dense_add_speculative(const DenseViews & v, DenseState & s, int q, hipStream_t st, int tag, int wait_chain_tag, int post_x_tag);
hipError_t dense_remove(const DenseViews & v, DenseState & s, int l, int q, int gone, hipStream_t st);
// warm start (experimental::GoldfarbIdnaniSolver.cpp:388-487): with the LLT in G, the active list in s.wact and the status bytes in
// s.st: N -> B = L^-1 N -> Householder QR (in s.R: R is its upper triangle), J = L^-T Q, b_act -> s.bact
hipError_t dense_warm_factor(const DenseViews & v, DenseState & s, int q, hipStream_t st);
// initializePrimalDualPoints: alpha = J^T a -> s.d, beta -> s.beta, x, u (q entries), f -> mailbox f0
hipError_t dense_warm_point(const DenseViews & v, DenseState & s, int q, hipStream_t st);
// see gi_dense.hip: do kernels of streams a and b run side by side?  dscratch2: two ints of device memory
hipError_t dense_probe_concurrency(int * dscratch2, hipStream_t a, hipStream_t b, bool * concurrent);
void dense_rot_profile_dump(); // diagnostic build (-DJRLQP_ROT_PROF) only: prints the sweep's time line to stderr; otherwise nothing
} // namespace jrlqp
