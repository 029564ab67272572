/*
 * jrlqp_hip.h -- C ABI of the MI355X-native Goldfarb-Idnani hot path (libjrlqp_hip.so).
 *
 * This is the drop-in boundary.  The reference (jrl-umi3218/jrl-qp) has no FFI: its extension
 * point is the C++ class jrl::qp::DualSolver with eight protected virtual hooks (include/jrl-qp/DualSolver.h:101-121)
 * filled in by jrl::qp::GoldfarbIdnaniSolver (src/GoldfarbIdnaniSolver.cpp).  Each entry point below states the
 * reference interface it replaces.  Plain pointers and sizes only; no C++/torch types; every function returns a
 * jrlqp_status (0 = ok) and never throws.  All matrix arguments are column-major fp64, exactly as the reference's
 * Eigen::Ref<MatrixXd> views (include/jrl-qp/defs.h:11-14).  Pointers named d* are DEVICE pointers (HBM).
 *
 * Rounding contract: the kernels follow the CANON rules of DESIGN.md §4 (in-order fma chains, correctly rounded
 * / and sqrt, no contraction), so results are bit-identical to the CPU oracle in oracle/ on the same inputs.
 */
#ifndef JRLQP_HIP_H
#define JRLQP_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JRLQP_ABI_VERSION 4 /* 4: jrlqp_blockgi_create_form / jrlqp_blockgi_form (contract C9 of the structured GI loop).  2: jrlqp_gi_batch_workspace_bytes / jrlqp_gi_solve_batch_ws; no allocation inside any solve.  3: hook level
                             * complete (jrlqp_dense_init_active_set, mirror kept by add / remove), jrlqp_dense_set_trace, multi-device entry */

/* jrl::qp::ActivationStatus, include/jrl-qp/enums.h:15-23 (same numeric values) */
enum jrlqp_activation
{
  JRLQP_INACTIVE = 0,
  JRLQP_LOWER = 1,
  JRLQP_UPPER = 2,
  JRLQP_EQUALITY = 3,
  JRLQP_LOWER_BOUND = 4,
  JRLQP_UPPER_BOUND = 5,
  JRLQP_FIXED = 6
};

/* jrl::qp::TerminationStatus, include/jrl-qp/enums.h:26-37 (same numeric values) */
enum jrlqp_termination
{
  JRLQP_SUCCESS = 0,
  JRLQP_INCONSISTENT_INPUT = 1,
  JRLQP_NON_POS_HESSIAN = 2,
  JRLQP_INFEASIBLE = 3,
  JRLQP_MAX_ITER_REACHED = 4,
  JRLQP_LINEAR_DEPENDENCY_DETECTED = 5,
  JRLQP_OVERCONSTRAINED_PROBLEM = 6,
  JRLQP_UNKNOWN = 7
};

/* error codes of the ABI itself (not solver outcomes) */
enum jrlqp_status
{
  JRLQP_OK = 0,
  JRLQP_ERR_INVALID_ARGUMENT = -1,
  JRLQP_ERR_UNSUPPORTED_SIZE = -2, /* batched path: n > JRLQP_BATCH_MAX_N or LDS budget exceeded */
  JRLQP_ERR_NO_DEVICE = -3,
  JRLQP_ERR_HIP = -4,          /* a HIP runtime call failed; see jrlqp_last_hip_error() */
  JRLQP_ERR_WORKSPACE = -5
};

#define JRLQP_BATCH_MAX_N 256 /* largest n of the batched entry points */
#define JRLQP_BATCH_LDS_N 128 /* up to here the factor J lives in LDS (gi_batch_big.hip); beyond, L / J live in the caller's workspace
                                 too (L2-resident), the same kernel with four vector entries per lane */
#define JRLQP_BATCH_WAVE_N 64 /* up to here a row of the factors is one lane of a wavefront (gi_batch_kernel.h); beyond it (cold
                                 start only) the solver of gi_batch_big.hip: J in LDS, R in LDS or in the caller's workspace
                                 (jrlqp_gi_batch_workspace_bytes), a persistent grid looping over the batch */

/* jrl::qp::SolverOptions, include/jrl-qp/SolverOptions.h:16-20 (maxIter_ = 500, bigBnd_ = 1e100).  warmStart_ and
 * the logging fields have no GPU counterpart: the production solver resets the active set anyway
 * (GoldfarbIdnaniSolver.cpp:75). */
typedef struct jrlqp_options
{
  int max_iter;
  double big_bnd;
} jrlqp_options;

/* per-iteration record (debug aid; the counterpart of the reference's LogFlags::ACTIVE_SET log lines,
 * DualSolver.cpp:116,138,153,159) */
typedef struct jrlqp_trace_rec
{
  int p, status, l, kind; /* kind: 0 full step + add, 1 partial step + drop, 2 dual-only step + drop */
  double t1, t2;
} jrlqp_trace_rec;

int jrlqp_abi_version(void);
const char * jrlqp_status_string(int status);
int jrlqp_last_hip_error(void);
/* number of visible devices; <= 0 when there is none (the library still loads on a CPU-only box) */
int jrlqp_device_count(void);
void jrlqp_default_options(jrlqp_options * opts);

/* ------------------------------------------------------------------------------------------------------------
 * Mode B -- device-resident batch: B independent dense QPs, one workgroup per QP, the whole GI state machine
 * on-device.  Replaces B calls of
 *     TerminationStatus GoldfarbIdnaniSolver::solve(G, a, C, bl, bu, xl, xu)   (GoldfarbIdnaniSolver.h:27-33,
 *                                                                              GoldfarbIdnaniSolver.cpp:18-54)
 * followed by solution() / multipliers() / objectiveValue() / iterations() / activeSet()
 * (DualSolver.h:40-57, DualSolver.cpp:33-84).
 *
 * Packed layout, problem b at offset b*stride:
 *   dG  [B][n*n]   col-major SPD; READ ONLY here (the reference clobbers G with its Cholesky factor; pass dL to get it)
 *   da  [B][n]
 *   dC  [B][n*m]   n x m col-major: column i = normal of constraint i
 *   dbl,dbu [B][m];  dxl,dxu [B][nb], nb == n or 0 (then they may be NULL)
 * Outputs:
 *   dx [B][n]; du [B][m+nb] sign-expanded multipliers (DualSolver.cpp:38-69); dstatus [B][m+nb] int32 activation
 *   status; dterm [B] int32 TerminationStatus; diters [B] int32; df [B] objective value.
 *   dL   optional [B][n*n]: lower triangle receives the Cholesky factor of G (what the reference leaves in G); the strictly
 *        upper part is not written.  dL MAY ALIAS dG: every problem's G is read completely before its factor is written, so
 *        dL == dG reproduces the reference's in-place contract (G's lower triangle <- L, upper part untouched).
 *   dtrace optional [B][trace_cap] records.
 * No allocation, no host synchronisation: the call enqueues one kernel on `stream` (a hipStream_t, may be NULL).
 * Requires 1 <= n <= JRLQP_BATCH_MAX_N.  Shapes that need a workspace (cold: from n = 105 .. 113 on, depending on m + nb, the
 * packed R factor -- and past n = 128 J itself -- no longer fit in LDS; warm: every n > 64) take the caller-owned workspace of
 * jrlqp_gi_solve_batch_ws below; this entry point then returns JRLQP_ERR_WORKSPACE instead of allocating.
 */
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
                         void * stream);

/* Warm-started variant.  Replaces B calls of
 *     experimental::GoldfarbIdnaniSolver::solve(G, a, C, bl, bu, xl, xu, as)  with options.warmStart_ = true
 *     (include/jrl-qp/experimental/GoldfarbIdnaniSolver.h, src/experimental/GoldfarbIdnaniSolver.cpp:20-112).
 * das [B][m+nb] int32: the activation-status guess per problem, general constraints first, then bounds -- exactly
 * what activeSet() returned for a previous solve (so a dstatus buffer can be passed back; das may alias dstatus:
 * that is the reference's "as.empty(): reuse the solver's own active set").  NULL = no guess.  Data equalities
 * (bl == bu, xl == xu) are activated whatever the guess says; guesses on infinite bounds (beyond +-big_bnd), FIXED
 * guesses on free variables and entries the reference rejects with assert() are ignored; if more than n constraints
 * come out active, inequality guesses are dropped from the back (src/experimental/GoldfarbIdnaniSolver.cpp:309-386).
 * J and R then come from a Householder QR of L^-1 N (:388-458); constraints whose multiplier is < -1e-14 are dropped
 * one at a time, each counting as one iteration (:78-109); then the same loop as the cold solver.
 * Every other argument as for jrlqp_gi_solve_batch. */
int jrlqp_gi_solve_batch_warm(int n,
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
                              const int * das,
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
                              void * stream);

/* Workspace contract of the batched solver -- the counterpart of "a pre-sized solver never allocates inside solve()"
 * (DualSolver::resize, include/jrl-qp/DualSolver.h:33; internal/Workspace.h:37-43; asserted by
 * tests/GoldfarbIdnaniSolverTest.cpp:109-116 under set_is_malloc_allowed(false)).
 *   jrlqp_gi_batch_workspace_bytes: device bytes a batch of B problems of shape (n, m, nb) needs beside its inputs and outputs
 *     on the CURRENT device, cold or warm-started: 0 for every shape whose state fits in LDS and registers (all n <= 64); beyond,
 *     per resident workgroup of the persistent grid (sized by the machine, not by B), the n x n array B = L^-1 N of a warm
 *     start, the packed R factor -- from n = 79 to 95 for batches of more problems than the device has CUs, so that two workgroups
 *     share a CU (round 5), and past n = 104 .. 112 (it depends on m + nb too) because it no longer fits in LDS --, and past n = 128 the factors L / J themselves.  A cold solve needs only the R / J
 *     part and accepts a smaller workspace accordingly; WITHOUT any workspace (NULL) it still runs wherever R fits in LDS (n <= 104 at
 *     least; one workgroup per CU from n = 79); -1 if the shape is unsupported.
 *   jrlqp_gi_solve_batch_ws: jrlqp_gi_solve_batch (warm == 0, das ignored) or jrlqp_gi_solve_batch_warm (warm != 0) with that
 *     workspace: dworkspace is a 16-byte aligned device pointer to at least workspace_bytes >= the figure above (NULL / 0 when
 *     that figure is 0).  The workspace is scratch: nothing survives the call, and one workspace must not be shared by solves
 *     that may run concurrently (different streams).  Too small a workspace returns JRLQP_ERR_WORKSPACE; nothing is launched.
 * No entry point of this library allocates or frees device or host memory inside a solve. */
long long jrlqp_gi_batch_workspace_bytes(int n, int m, int nb, int B);
int jrlqp_gi_solve_batch_ws(int warm,
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
                            const int * das,
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
                            void * dworkspace,
                            long long workspace_bytes,
                            void * stream);

/* experimental::BoxAndSingleConstraintSolver over a batch.  Replaces B calls of
 *     BoxAndSingleConstraintSolver::solve(x0, c, bl, xl, xu)      (include/jrl-qp/experimental/BoxAndSingleConstraintSolver.h:9-33,
 *                                                                  src/experimental/BoxAndSingleConstraintSolver.cpp:11-44)
 *   min 1/2 ||x - x0||^2   s.t.  c.x >= bl,   xl <= x <= xu
 * i.e. the dense GI solver on G = I with ONE general constraint (bu = +inf) and mandatory bounds, whose init_ is the closed
 * form of BoxAndSingleConstraintSolver.cpp:46-94: start at x0 clamped to the box with the clamped variables' bounds active
 * (multiplier |x0_i - bound_i|, J a permutation, R = diag(+-1)), no factorisation; then the unchanged GI loop.
 *   dx0, dc, dxl, dxu [B][n]; dbl [B].  Outputs: dx [B][n]; du, dstatus [B][1+n] (the general constraint first, then the
 *   bounds); dterm, diters, df [B]; df is 1/2 ||x - x0||^2 itself (the reference's objectiveValue(), constant included).
 * n <= JRLQP_BATCH_MAX_N (256).  Up to JRLQP_BATCH_WAVE_N variables the closed-form start is one wavefront (lane = variable); beyond
 * (round 4) the large-n batched kernel runs it with thread = variable.  jrlqp_box_solve_batch needs no workspace while R fits in LDS (n <= 104 at least);
 * past that the packed R (and past n = 128 the factor J) of the large-n kernel live in a caller-owned workspace of
 * jrlqp_gi_batch_workspace_bytes(n, 1, n, B) bytes, passed to jrlqp_box_solve_batch_ws (JRLQP_ERR_WORKSPACE without it).
 * One launch, no allocation, no host synchronisation. */
int jrlqp_box_solve_batch(int n,
                          int B,
                          const double * dx0,
                          const double * dc,
                          const double * dbl,
                          const double * dxl,
                          const double * dxu,
                          const jrlqp_options * opts,
                          double * dx,
                          double * du,
                          int * dstatus,
                          int * dterm,
                          int * diters,
                          double * df,
                          jrlqp_trace_rec * dtrace,
                          int trace_cap,
                          void * stream);
int jrlqp_box_solve_batch_ws(int n,
                          int B,
                          const double * dx0,
                          const double * dc,
                          const double * dbl,
                          const double * dxl,
                          const double * dxu,
                          const jrlqp_options * opts,
                          double * dx,
                          double * du,
                          int * dstatus,
                          int * dterm,
                          int * diters,
                          double * df,
                          jrlqp_trace_rec * dtrace,
                          int trace_cap,
                          void * dworkspace,
                          long long workspace_bytes,
                          void * stream);

/* ------------------------------------------------------------------------------------------------------------
 * One process, several devices.  The reference has no counterpart (it is single-threaded CPU code); this is the batched front
 * end of SURVEY.md section 8(e): QPs are independent, so a batch of B_total problems is cut into ndev CONTIGUOUS shards,
 *     shard r = [first_r, first_r + count_r),  jrlqp_shard_range(B_total, ndev, r, &first_r, &count_r)
 * (balanced: the first B_total % ndev shards hold one problem more -- the same arithmetic as the one-process-per-GPU launcher,
 * jrl-qp_amd/dist.py), shard r lives entirely on device device_ids[r] and is solved there by the batched kernel; there is NO
 * exchange between devices during the solve and no collective.  shards[r] holds DEVICE pointers into device_ids[r]'s memory for
 * the count_r problems of the shard, packed exactly as for jrlqp_gi_solve_batch (problem first_r + k of the batch is problem k of
 * the shard); `stream` is a stream of that device (NULL: its default stream); `dworkspace` / `workspace_bytes` is the shard's
 * workspace (jrlqp_gi_batch_workspace_bytes_on(device, n, m, nb, count_r); nothing is needed up to n = 64) -- nothing is
 * allocated inside the call.  Every shard is enqueued before any is waited for, so the devices run side by side while the
 * calling thread only launches; synchronize != 0 then waits for every shard's stream.  The caller's current device is restored.
 * jrlqp_gi_gather_x_multi copies the shards' solutions into ONE array of B_total x n doubles on device device_ids[root] (peer
 * copies over xGMI, each ordered behind its shard's solve on the shard's own stream; the epilogue of SURVEY.md section 5 for a
 * single consumer: 13 MB for 32768 problems of 50 unknowns) -- the only traffic between devices, and optional.
 * Errors: JRLQP_ERR_INVALID_ARGUMENT for a device id out of range or used twice; whatever jrlqp_gi_solve_batch_ws returns for a
 * shard otherwise (shards enqueued before the failing one keep running). */
typedef struct jrlqp_shard_buffers
{
  const double *dG, *da, *dC, *dbl, *dbu, *dxl, *dxu; /* inputs of the shard (dxl / dxu NULL when nb == 0, dC / dbl / dbu when m == 0) */
  const int * das;                                    /* activation-status guesses of a warm start, or NULL */
  double *dx, *du;                                    /* outputs of the shard */
  int *dstatus, *dterm, *diters;
  double * df;
  double * dL;                                        /* optional: the Cholesky factors (may alias dG), or NULL */
  void * dworkspace;
  long long workspace_bytes;
  void * stream;
} jrlqp_shard_buffers;
void jrlqp_shard_range(int B_total, int nshards, int r, int * first, int * count);
long long jrlqp_gi_batch_workspace_bytes_on(int device, int n, int m, int nb, int B);
int jrlqp_gi_solve_batch_multi(int warm, int n, int m, int nb, int B_total, int ndev, const int * device_ids,
                               const jrlqp_shard_buffers * shards, const jrlqp_options * opts, int synchronize);
int jrlqp_gi_gather_x_multi(int n, int B_total, int ndev, const int * device_ids, const jrlqp_shard_buffers * shards, int root,
                            double * dx_all_on_root, int synchronize);
/* Which shard failed to enqueue in the last jrlqp_gi_solve_batch_multi call of this process (-1: none).  On such a failure the call
 * returns the shard's error AFTER waiting for the shards it had already enqueued (every buffer may be released on return); the
 * shards behind the failing one were not started. */
int jrlqp_gi_multi_failed_shard(void);
/* Rehearsal on a box with fewer GPUs than shards (tests): with `on` != 0 (or JRLQP_MULTI_SHARED_DEVICE set in the environment) a
 * device may appear more than once in device_ids -- two shards, two streams, two workspaces on one device.  No scaling meaning.
 * jrlqp_debug_multi_inject_failure(r): shard r of the next calls reports an error instead of being enqueued (-1: off). */
int jrlqp_debug_multi_allow_shared_device(int on);
int jrlqp_debug_multi_inject_failure(int shard);

/* LDS bytes one workgroup of the batched kernel uses for (n, m, nb), and where the constraint matrix lives during the
 * solve: *c_placement = 2 one column per thread in registers, 1 staged in LDS, 0 streamed from L2/HBM; returns -1 if
 * unsupported.  Lets a caller reason about occupancy (163840 B of LDS per CU). */
long jrlqp_gi_batch_lds_bytes(int n, int m, int nb, int * c_placement);

/* the same for the warm-started kernel (it keeps N / the QR factors in LDS as well) */
long jrlqp_gi_batch_warm_lds_bytes(int n, int m, int nb, int * c_placement);

/* ------------------------------------------------------------------------------------------------------------
 * Mode A -- ONE large dense QP, state in HBM, the active-set state machine on the host.
 *
 * A handle plays the role of a pre-sized solver object (GoldfarbIdnaniSolver(nbVar, nbCstr, useBounds),
 * GoldfarbIdnaniSolver.h:19): it owns J, R, d, z, x, u, r on the device; solve() allocates nothing.
 * Two levels are exported:
 *   (1) the eight hooks DualSolver calls (DualSolver.h:101-121), fused where the reference calls them back to back, so
 *       that a jrl-qp maintainer can keep DualSolver::solve() as is and forward the hooks;
 *   (2) jrlqp_dense_solve = GoldfarbIdnaniSolver::solve(G, a, C, bl, bu, xl, xu) (GoldfarbIdnaniSolver.cpp:18-54): the same
 *       loop, run by the library's own host C++ restatement of DualSolver::solve.
 * All data pointers are DEVICE pointers; G (ld ldg) is overwritten by its Cholesky factor (lower triangle, strictly
 * upper part untouched) exactly like the reference; C is n x m column-major with leading dimension ldc.
 * Requires n <= JRLQP_DENSE_MAX_N.
 */
#define JRLQP_DENSE_MAX_N 4096
typedef struct jrlqp_dense jrlqp_dense;
int jrlqp_dense_create(int n, int m, int nb, jrlqp_dense ** out);
void jrlqp_dense_destroy(jrlqp_dense * h);
/* DualSolver::resize(nbVar, nbCstr, useBounds) (DualSolver.h:33) on an existing handle: buffers are grow-only like the
 * reference's workspaces (internal/Workspace.h:37-43) -- a problem with n, m, nb no larger than what the handle was created
 * for is solved in the existing buffers, with no allocation on the host or on the device; otherwise JRLQP_ERR_WORKSPACE
 * (create a larger handle).  This is what keeps "solver(5, 8, true); solve(problems of different shapes)" allocation-free
 * (tests/GoldfarbIdnaniSolverTest.cpp:101-125). */
int jrlqp_dense_resize(jrlqp_dense * h, int n, int m, int nb);
int jrlqp_dense_last_hip_error(const jrlqp_dense * h);
int jrlqp_dense_bind(jrlqp_dense * h, double * dG, int ldg, const double * da, const double * dC, int ldc, const double * dbl,
                     const double * dbu, const double * dxl, const double * dxu);
/* init_ without the equality pre-activation: *llt_info = -1 or the failing column (Eigen llt_inplace), *f0 = a.x0/2 */
int jrlqp_dense_init(jrlqp_dense * h, int * llt_info, double * f0, void * stream);
/* initActiveSet (GoldfarbIdnaniSolver.cpp:268-287), the second half of the reference's init_ hook (:75-79): activates every data
 * equality (bl_i == bu_i, status EQUALITY) and then every fixed variable (xl_i == xu_i, FIXED) by increasing index, each through
 * addInitialConstraint (:295-338).  Call it right behind a successful jrlqp_dense_init.  *term = JRLQP_SUCCESS, or
 * JRLQP_OVERCONSTRAINED_PROBLEM when there are more of them than unknowns (UB in the reference; nothing is activated then).
 * Optional outputs for the caller's own ActiveSet: *nactive, active_host[k] / status_host[k] (k < *nactive <= n, in activation
 * order), *f (the objective value after the full steps).  Synchronises the stream. */
int jrlqp_dense_init_active_set(jrlqp_dense * h, double big_bnd, int * term, int * nactive, int * active_host, int * status_host,
                                double * f, void * stream);
/* selectViolatedConstraint_: *p = -1 and *status = JRLQP_INACTIVE when no constraint is violated.  Also clears u[q] for the
 * current number q of active constraints (DualSolver.cpp:119-121 does that right behind the selection; u lives on the device). */
int jrlqp_dense_select(jrlqp_dense * h, int * p, int * status, void * stream);
/* computeStep_ + computeStepLength_'s reductions + dot_: d, z, r stay on the device; returns t1, l, z.z, n.z, n.x, u[q] */
int jrlqp_dense_step(jrlqp_dense * h, int p, int status, int q, double big_bnd, double * t1, int * l, double * zz, double * nz,
                     double * nx, double * uq, void * stream);
/* x += t z (primal != 0), u[0:q] -= t r, u[q] += t  (DualSolver.cpp:136-137, :146-150) */
int jrlqp_dense_advance(jrlqp_dense * h, double t, int q, int primal, void * stream);
/* addConstraint (DualSolver.cpp:231-235) + addConstraint_ (GoldfarbIdnaniSolver.cpp:221-237); q = active constraints BEFORE the
 * addition.  removeConstraint (DualSolver.cpp:237-244, incl. the shift of u) + removeConstraint_ (:239-256); q = active constraints
 * BEFORE the removal, index = the constraint at position l.  Both also keep the handle's own mirror of the ActiveSet (status per
 * constraint + ordered list), which jrlqp_dense_results expands the multipliers with: q (and index) must agree with it, otherwise
 * JRLQP_ERR_INVALID_ARGUMENT and nothing is launched. */
int jrlqp_dense_add(jrlqp_dense * h, int p, int status, int q, void * stream);
int jrlqp_dense_remove(jrlqp_dense * h, int l, int q, int index, void * stream);
int jrlqp_dense_solve(jrlqp_dense * h, double * dG, int ldg, const double * da, const double * dC, int ldc, const double * dbl,
                      const double * dbu, const double * dxl, const double * dxu, const jrlqp_options * opts, int * term,
                      void * stream);
/* Warm-started variant: experimental::GoldfarbIdnaniSolver::solve(G, a, C, bl, bu, xl, xu, as) with options.warmStart_ = true
 * (src/experimental/GoldfarbIdnaniSolver.cpp:20-112, :306-487).  as_host: m + nb activation-status guesses on the HOST (general
 * constraints first, then bounds) -- what jrlqp_dense_results returned in status_host for a previous solve -- or NULL for no
 * guess.  Same rules as jrlqp_gi_solve_batch_warm: data equalities are activated whatever the guess says, guesses on infinite
 * bounds and ill-typed entries are ignored, beyond n active constraints inequality guesses are dropped from the back; J and R
 * come from a Householder QR of L^-1 N (kept in the handle's R buffer, like the reference), constraints with a multiplier
 * < -1e-14 are dropped one at a time, each counting as one iteration; then the same loop as jrlqp_dense_solve.  No allocation. */
int jrlqp_dense_solve_warm(jrlqp_dense * h, double * dG, int ldg, const double * da, const double * dC, int ldc, const double * dbl,
                           const double * dbu, const double * dxl, const double * dxu, const int * as_host, const jrlqp_options * opts,
                           int * term, void * stream);
/* host outputs: x[n], sign-expanded multipliers u[m+nb], activation status[m+nb] -- valid behind jrlqp_dense_solve[_warm] AND
 * behind a loop of the caller's own over the hooks above --, and the iteration count and objective value of the LIBRARY's loop
 * (a caller that runs the loop itself owns it_ and f_, as DualSolver does) */
int jrlqp_dense_results(jrlqp_dense * h, double * x_host, double * u_host, int * status_host, int * iters, double * f, void * stream);
/* Iteration trace of jrlqp_dense_solve[_warm] -- the counterpart of the reference's per-pass log lines (DualSolver.cpp:105-108, 116,
 * 128-129, 138, 153, 159): record `it` of trace_host (HOST memory, caller-owned, trace_cap records; NULL / 0 switches it off)
 * is written by pass `it` of the loop, exactly like dtrace of the batched entry points. */
int jrlqp_dense_set_trace(jrlqp_dense * h, jrlqp_trace_rec * trace_host, int trace_cap);
/* The VECTORS of the reference's per-pass log (LogFlags::ITERATION_BASIC_DETAILS: x, u, f at the top of a pass, DualSolver.cpp:108; z, r
 * behind the step computation, :128-129), round 5.  Record `it` of vectors_host (HOST memory, caller-owned, `cap` records of
 * jrlqp_dense_trace_vectors_stride(h) doubles; NULL / 0 switches it off) holds
 *     x [n] | u [m+nb+1] (q + 1 entries valid, by POSITION in the active set) | z [n] | r [m+nb+1] (q valid) | f | q
 * as they stand when pass `it` has computed its step and applied nothing yet.  A debugging aid: while it is on, every pass is
 * synchronised and copied (no sweep or scan runs ahead of the decision); results are unchanged. */
int jrlqp_dense_trace_vectors_stride(const jrlqp_dense * h);
int jrlqp_dense_set_trace_vectors(jrlqp_dense * h, double * vectors_host, int cap);
/* How the handle orders the kernels of a pass that run on different streams: 0 = device-side tags, 1 = events.  Decided at
 * creation (JRLQP_DENSE_SYNC=events|tags overrides) and switched to events for good by a solve whose tag
 * hand-off starved (the solve is then repeated with events and returns its normal result); *event_reruns counts those. */
int jrlqp_dense_sync_mode(const jrlqp_dense * h, int * event_reruns);

/* ------------------------------------------------------------------------------------------------------------
 * Structured Cholesky family.  Replaces (include/jrl-qp/decomposition/blockArrowLLT.h:70-110, triBlockDiagLLT.h:41-72):
 *     bool blockArrowLLT(diag, side, up);            void blockArrowLSolve(diag, side, up, M, start, end);
 *     void blockArrowLTransposeSolve(diag, side, up, M, start, end);
 *     bool triBlockDiagLLT(diag, subDiag);           void triBlockDiagLSolve(diag, subDiag, M, start);
 *     void triBlockDiagLTransposeSolve(diag, subDiag, M, end);
 * The reference passes std::vector<Eigen::Ref<MatrixXd>> views into caller-owned storage; here a PLAN object holds
 * the same list as device pointers + leading dimensions (one list per problem of a batch of identically shaped
 * problems), so that the compute calls allocate nothing.  Blocks stay in caller-owned device memory and are factored
 * in place exactly like the reference does (diag[i] lower triangle <- L_i, side[i] <- B_i or B_i^T).
 *   kind: 0 = block arrow pointing down (up = false), 1 = block arrow pointing up (up = true), 2 = block tridiagonal
 *   block_sizes[nblk] (host); every block <= JRLQP_BLOCK_MAX rows (up to 64 rows: one wavefront, row per lane, everything of a
 *   block in LDS at once; 65 .. 128 rows (round 4): two rows per lane, side blocks pass through LDS in panels -- same results per
 *   element, slower kernels).
 *   side block i: kind 0: n_last x n_i; kind 1: n_{i+1} x n_0; kind 2: n_{i+1} x n_i.
 */
#define JRLQP_BLOCK_MAX 128
#define JRLQP_BLOCKGI_BLOCK_MAX 128 /* the structured GI loop (jrlqp_blockgi_*) */
typedef struct jrlqp_blocks jrlqp_blocks;
int jrlqp_blocks_create(int kind, int nblk, const int * block_sizes, int nprob, int max_nrhs, jrlqp_blocks ** out);
void jrlqp_blocks_destroy(jrlqp_blocks * plan);
/* ddiag: host array [nprob*nblk] of device pointers (problem-major), ldd[nblk]; dside: [nprob*(nblk-1)], lds[nblk-1] */
int jrlqp_blocks_bind(jrlqp_blocks * plan, double * const * ddiag, const int * ldd, double * const * dside, const int * lds, void * stream);
/* the factorisation (asynchronous).  Failure semantics are the reference's, including its `ret > 0` quirk
 * (blockArrowLLT.cpp:65, triBlockDiagLLT.cpp:19): query them with jrlqp_blocks_result. */
int jrlqp_blocks_llt(jrlqp_blocks * plan, void * stream);
/* synchronises the stream and writes the bool the reference returns, one per problem (1 = success) */
int jrlqp_blocks_result(jrlqp_blocks * plan, int * ok_host, void * stream);
/* in-place solve  P L X = M (transpose = 0)  or  L^T P^T X = M (transpose = 1); dM: device, N x nrhs column-major with
 * leading dimension ldm, problem b at dM + b*mstride.  start/end: first non-zero row / first row of the trailing zero
 * block (end < 0: none) -- the reference's zero-window arguments (tri-diagonal: start for L, end for L^T). */
int jrlqp_blocks_solve(jrlqp_blocks * plan, int transpose, double * dM, int ldm, long mstride, int nrhs, int start, int end, void * stream);

/* ------------------------------------------------------------------------------------------------------------
 * Structured GI solver.  Replaces experimental::BlockGISolver (include/jrl-qp/experimental/BlockGISolver.h:24-40):
 *     BlockGISolver(int nbVar, int nbCstr, bool useBounds);
 *     TerminationStatus solve(const structured::StructuredG & G, const VectorConstRef & a, const structured::StructuredC & C,
 *                             const VectorConstRef & bl, const VectorConstRef & bu, const VectorConstRef & xl,
 *                             const VectorConstRef & xu, const std::vector<ActivationStatus> & as = {});
 * and, behind it, the hooks of src/experimental/BlockGISolver.cpp:62-288 with structured::StructuredG / StructuredC /
 * StructuredJ / StructuredQR (src/structured/{StructuredG,StructuredC,StructuredJ,StructuredQR}.cpp) and
 * internal::OrthonormalSequence (src/internal/OrthonormalSequence.cpp).  J = L^-T Q is never formed: L stays in the
 * caller's blocks (factorised in place, like StructuredG::lltInPlace), Q is a growing sequence of Householder
 * reflectors (one per added constraint) and Givens sequences (one per removed constraint), R is dense upper triangular.
 *
 * A PLAN describes a batch of `nprob` identically shaped problems:
 *   StructuredG(Type, diag, offDiag)  -> kind / nblk / block_sizes as for jrlqp_blocks_create (kind 0 = BlockArrowDown,
 *                                        1 = BlockArrowUp, 2 = TriBlockDiagonal) + jrlqp_blockgi_bind(ddiag, ldd, doff, ldo)
 *   StructuredC(std::vector<MatrixConstRef>) (block diagonal, StructuredC.cpp:9-23)
 *                                     -> ncb blocks, block i is cvar[i] x ccstr[i] column-major (one constraint per
 *                                        column); sum cvar = nbVar; + jrlqp_blockgi_bind(dcblk, ldc)
 * The whole active-set loop runs on the device (one workgroup per problem, one launch per batch); the orthonormal
 * sequence grows by one element per iteration, so the plan's workspace is sized by `max_iter` at creation and
 * options.max_iter must not exceed it.  n <= 4096 unknowns per problem (the vectors of unknowns live in LDS: a shape whose vectors -- 3 n + 2 min(n, m + nb) doubles -- do not
 * fit is refused with JRLQP_ERR_UNSUPPORTED_SIZE; 3072 until round 4); blocks of at most
 * JRLQP_BLOCKGI_BLOCK_MAX rows.
 * As in the reference, only the cold start without data equalities / fixed variables is functional (the reference
 * asserts in initializePrimalDualPoints, BlockGISolver.cpp:475): such inputs return INCONSISTENT_INPUT per problem.
 * Outputs as jrlqp_gi_solve_batch: x [nprob*n], sign-expanded multipliers u and activation status [nprob*(m+nb)],
 * term / iters / f [nprob].  da: [nprob*n]; dbl, dbu: [nprob*m]; dxl, dxu: [nprob*n] or NULL (plan without bounds).
 */
typedef struct jrlqp_blockgi jrlqp_blockgi;
int jrlqp_blockgi_create(int kind, int nblk, const int * block_sizes, int ncb, const int * cvar, const int * ccstr, int use_bounds,
                         int nprob, int max_iter, jrlqp_blockgi ** out);
/* ABI 4 (round 5): which FORM of the QR factors of L^-1 N the loop keeps.
 *   SEQUENCE  the reference's: Q as an internal::OrthonormalSequence applied element by element (StructuredQR.cpp:70-102,
 *             OrthonormalSequence.cpp:189-207), R by a triangular solve -- O(n k) per iteration, k = elements ever made; the
 *             workspace is sized by max_iter and options.max_iter must not exceed it;
 *   C9        no Q: the sparse columns B = L^-1 N of the active constraints and S = R^-1, r and y = Q2 Q2^T b from the corrected
 *             semi-normal equations -- O(q^2) per iteration, q = active constraints; workspace independent of
 *             max_iter.  Same decisions as SEQUENCE on every tested problem, x / u to 1e-9 (tests/test_oracle_blockgi_literal.py);
 *   AUTO      C9 for the arrows of more than 1024 unknowns, SEQUENCE otherwise (the tridiagonal shape's columns of B are dense; small
 *             problems with few iterations batch better in the sequence kernel); what jrlqp_blockgi_create chooses.  The environment variable JRLQP_BLOCKGI_FORM=sequence|c9 overrides AUTO (diagnostic).
 * jrlqp_blockgi_form says which one a plan runs (the oracle to compare with: orc_blockgi_solve / orc_blockgi_solve_c9). */
#define JRLQP_BLOCKGI_FORM_AUTO (-1)
#define JRLQP_BLOCKGI_FORM_SEQUENCE 0
#define JRLQP_BLOCKGI_FORM_C9 1
int jrlqp_blockgi_create_form(int kind, int nblk, const int * block_sizes, int ncb, const int * cvar, const int * ccstr, int use_bounds,
                              int nprob, int max_iter, int form, jrlqp_blockgi ** out);
int jrlqp_blockgi_form(const jrlqp_blockgi * plan);
void jrlqp_blockgi_destroy(jrlqp_blockgi * plan);
long long jrlqp_blockgi_workspace_bytes(const jrlqp_blockgi * plan);
/* ddiag [nprob*nblk], doff [nprob*(nblk-1)], dcblk [nprob*ncb]: HOST arrays of DEVICE pointers (problem-major) */
int jrlqp_blockgi_bind(jrlqp_blockgi * plan, double * const * ddiag, const int * ldd, double * const * doff, const int * ldo,
                       const double * const * dcblk, const int * ldc, void * stream);
/* factorises the bound blocks in place and solves (asynchronous on `stream`); dtrace: optional, trace_cap records per problem */
int jrlqp_blockgi_solve(jrlqp_blockgi * plan, const double * da, const double * dbl, const double * dbu, const double * dxl,
                        const double * dxu, const jrlqp_options * opts, double * dx, double * du, int * dstatus, int * dterm,
                        int * diters, double * df, jrlqp_trace_rec * dtrace, int trace_cap, void * stream);

/* ------------------------------------------------------------------------------------------------------------
 * Self-tests of the arithmetic contract (no reference counterpart; used by tests/ to prove that device /, sqrt,
 * fma and the Givens generator are bit-identical to the host's IEEE-754 results, and to pin the accumulation
 * order of v_mfma_f64_16x16x4_f64).  Arrays of n doubles on the device:
 *   ddiv = a/b, dsqrt = sqrt(|a|), dfma = fma(a,b,c), (dgc,dgs,dgr) = makeGivens(a,b).
 */
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
                         void * stream);
/* Diagnostic: phase profile.  kernel: JRLQP_PROFILE_GI_BATCH (jrlqp_gi_solve_batch[_warm]) or JRLQP_PROFILE_BLOCKGI
 * (jrlqp_blockgi_solve); each keeps a buffer of its own.  When dprof != NULL the following solves of that family run an
 * instrumented build of the kernel that accumulates clock ticks per phase into dprof[b*slots_per_problem + slot], slot <
 * JRLQP_PROFILE_SLOTS (device int64, zero it first; the slots are listed in jrl-qp_amd/csrc/gi_batch_kernel.h / blockgi.hip).
 * slots_per_problem must be >= JRLQP_PROFILE_SLOTS, and a solve of more than nprob problems while a buffer is set returns
 * JRLQP_ERR_INVALID_ARGUMENT instead of writing past it.  dprof == NULL returns to the production kernel.  Never used for
 * quoted times. */
#define JRLQP_PROFILE_SLOTS 32
#define JRLQP_PROFILE_GI_BATCH 0
#define JRLQP_PROFILE_BLOCKGI 1
int jrlqp_debug_set_profile(int kernel, long long * dprof, int slots_per_problem, int nprob);
/* Fault injection for tests/test_gpu_dense.py: in pass `pass` (counted like iterations(); -1 = off) of the next solves of this
 * handle the first kernel of the step waits for a tag nobody posts, gives up after ~1 s and reports it -- the path a starved
 * hand-off takes.  The solve must then come back with its normal result through the event-ordered rerun. */
int jrlqp_debug_dense_inject_sync_fault(jrlqp_dense * h, int pass);
/* D(16x16) = A(16x4) * B(4x16) + C(16x16), row-major, one MFMA instruction on one wave */
int jrlqp_selftest_mfma_f64(const double * dA, const double * dB, const double * dC, double * dD, void * stream);

#ifdef __cplusplus
}
#endif
#endif
