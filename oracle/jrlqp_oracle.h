/*
 * jrlqp_oracle.h -- CPU oracle for the dense Goldfarb-Idnani hot path and the structured
 * block-arrow / block-tridiagonal Cholesky family.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the product
 * (jrl-qp_amd/, include/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * It is a plain-C restatement of the reference algorithm (jrl-umi3218/jrl-qp):
 *   src/DualSolver.cpp:91-168, :200-244           generic GI loop
 *   src/GoldfarbIdnaniSolver.cpp:56-338           dense hooks
 *   include/jrl-qp/internal/ConstraintNormal.h:81-123
 *   src/internal/ActiveSet.cpp:45-168
 *   src/decomposition/blockArrowLLT.cpp, triBlockDiagLLT.cpp
 * The arithmetic the reference delegates to Eigen (not vendored, absent from this image, so the
 * reference itself cannot be built here -- see DESIGN.md "Oracle") is restated with ONE canonical
 * rounding order, documented in oracle/README.md ("CANON rules"), that the HIP kernels reproduce
 * bit for bit.  Parity pinning: tests/test_oracle_*.py check this file against the reference's own
 * known answers and fixtures (GI-paper example, MultiIK fixtures, planted-solution problems, dense
 * identities for the structured family, the ActiveSet integer sequence).
 */
#ifndef JRLQP_ORACLE_H
#define JRLQP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* enums.h:15-37 of the reference; numeric values are part of the contract */
enum
{
  ORC_INACTIVE = 0,
  ORC_LOWER = 1,
  ORC_UPPER = 2,
  ORC_EQUALITY = 3,
  ORC_LOWER_BOUND = 4,
  ORC_UPPER_BOUND = 5,
  ORC_FIXED = 6
};
enum
{
  ORC_SUCCESS = 0,
  ORC_INCONSISTENT_INPUT = 1,
  ORC_NON_POS_HESSIAN = 2,
  ORC_INFEASIBLE = 3,
  ORC_MAX_ITER_REACHED = 4,
  ORC_LINEAR_DEPENDENCY_DETECTED = 5,
  ORC_OVERCONSTRAINED_PROBLEM = 6,
  ORC_UNKNOWN = 7
};

/* One trace record per pass of the GI loop (DualSolver.cpp:103-166). */
typedef struct
{
  int p;      /* selected constraint index (Step 1) */
  int status; /* its activation status */
  int l;      /* position of the blocking constraint (ratio test) */
  int kind;   /* 0 = full step + add, 1 = partial step + drop, 2 = dual-only step + drop */
  double t1, t2;
} orc_trace_rec;

/* Workspace size in doubles for orc_gi_solve (n*n J + n*n R + vectors). */
long orc_gi_work_doubles(int n, int m, int nb);

/*
 * Dense GI solve.  Mirrors GoldfarbIdnaniSolver::solve (GoldfarbIdnaniSolver.cpp:18-54).
 *   G   n x n col-major, ld ldg, SPD; lower triangle overwritten by its Cholesky factor
 *   a   n
 *   C   n x m col-major, ld ldc: one constraint per column
 *   bl,bu  m;  xl,xu  nb (nb == n or 0)
 * Outputs (any may be NULL except x):
 *   x n; u m+nb sign-expanded multipliers (DualSolver.cpp:38-69); status m+nb activation status;
 *   active (<= n entries) ordered active list; *nactive; *iters; *f objective value.
 *   trace/trace_cap: optional per-iteration records.
 * work: orc_gi_work_doubles doubles, or NULL (then malloc'ed internally).
 * Returns a TerminationStatus value.
 */
int orc_gi_solve(int n,
                 int m,
                 int nb,
                 double * G,
                 int ldg,
                 const double * a,
                 const double * C,
                 int ldc,
                 const double * bl,
                 const double * bu,
                 const double * xl,
                 const double * xu,
                 int max_iter,
                 double big_bnd,
                 double * x,
                 double * u,
                 int * status,
                 int * active,
                 int * nactive,
                 int * iters,
                 double * f,
                 orc_trace_rec * trace,
                 int trace_cap,
                 double * work);

/*
 * Batch of independent dense problems, packed contiguously (problem b at base + b*stride):
 * G: n*n, a: n, C: n*m, bl/bu: m, xl/xu: nb; outputs x: n, u: m+nb, status: m+nb, term/iters: 1, f: 1.
 * G is NOT preserved (as in the reference).  nthreads <= 1 -> serial.  Used by tests and as bench.py's cpu_baseline.
 */
void orc_gi_solve_batch(int B,
                        int n,
                        int m,
                        int nb,
                        double * G,
                        const double * a,
                        const double * C,
                        const double * bl,
                        const double * bu,
                        const double * xl,
                        const double * xu,
                        int max_iter,
                        double big_bnd,
                        double * x,
                        double * u,
                        int * status,
                        int * term,
                        int * iters,
                        double * f,
                        int nthreads);

/* LITERAL evaluation of the three places where the CANON contract (oracle/README.md) departs from the most literal reading of
 * the reference's source -- C5, C6, C7 -- one flag each, plus "no fma".  Test infrastructure for tests/test_oracle_literal.py:
 * the DECISIONS of a solve (termination, iterations, activation statuses, order of the active list) must not depend on them. */
enum
{
  ORC_LIT_CHAIN_DOTS = 1,    /* z.z and n+.z as in-order chains (GoldfarbIdnaniSolver.cpp:174,183-190) instead of C5 trees */
  ORC_LIT_ROT_C3 = 2,        /* rotations x' = fma(c, x, -(s y)), y' = fma(s, x, c y) instead of C6 */
  ORC_LIT_SERIAL_GIVENS = 4, /* makeGivens(d_i, d_{i+1}), i = n-2 .. q, radius in d_i (GoldfarbIdnaniSolver.cpp:225-232) instead of C7 */
  ORC_LIT_NO_FMA = 8,        /* those dots and rotations with every product and sum rounded separately */
  ORC_LIT_CHAIN_GEMV = 16,   /* d = J^T n+, C^T x, n+.x and z = J2 d2 as in-order chains at every size instead of C8 (blocked beyond 256 unknowns) */
  ORC_LIT_R_SOLVE = 32,      /* r = R^-1 d1 by back substitution with R at every size (GoldfarbIdnaniSolver.cpp:146) instead of C10 (S = R^-1 kept explicitly beyond 256 unknowns) */
  ORC_LIT_ALL = 55
};
int orc_gi_solve_literal(int flags, int n, int m, int nb, double * G, int ldg, const double * a, const double * C, int ldc,
                         const double * bl, const double * bu, const double * xl, const double * xu, int max_iter,
                         double big_bnd, double * x, double * u, int * status, int * active, int * nactive, int * iters,
                         double * f, orc_trace_rec * trace, int trace_cap, double * work);
/* packed batch; active: B x n (entries beyond nactive[b] are -1), may be NULL */
void orc_gi_solve_batch_literal(int flags, int B, int n, int m, int nb, double * G, const double * a, const double * C,
                                const double * bl, const double * bu, const double * xl, const double * xu, int max_iter,
                                double big_bnd, double * x, double * u, int * status, int * active, int * nactive, int * term,
                                int * iters, double * f, int nthreads);

/* Timing loop of bench.py's cpu_baseline leg: the reference's benchmark loop (benchmarks/Solvers.cpp:447-453, :508-520) --
 * per thread one workspace and one G buffer, G restored by memcpy INSIDE the loop before every solve, `rounds` passes over the
 * B problems (static partition), outputs into caller-preallocated arrays.  Returns the wall seconds of the rounds alone
 * (between two barriers inside the parallel region).  G is read-only here. */
double orc_gi_bench_rounds(int rounds, int B, int n, int m, int nb, const double * G, const double * a, const double * C,
                           const double * bl, const double * bu, const double * xl, const double * xu, int max_iter,
                           double big_bnd, double * x, double * u, int * status, int * term, int * iters, double * f,
                           int nthreads);

/* experimental::BoxAndSingleConstraintSolver (src/experimental/BoxAndSingleConstraintSolver.cpp:11-94):
 *   min 1/2 ||x - x0||^2  s.t.  c.x >= bl,  xl <= x <= xu ; closed-form initial point / active set / J / R, then the GI loop.
 *   u, status: 1 + n entries (the general constraint first, then the bounds).  f is 1/2 ||x - x0||^2 itself. */
int orc_box_solve(int n, const double * x0, const double * c, double bl, const double * xl, const double * xu, int max_iter,
                  double big_bnd, double * x, double * u, int * status, int * active, int * nactive, int * iters, double * f,
                  orc_trace_rec * trace, int trace_cap);

/*
 * Warm-started dense GI solve: experimental::GoldfarbIdnaniSolver::solve(..., as) with options.warmStart_ = true
 * (src/experimental/GoldfarbIdnaniSolver.cpp:20-112, 309-487).  Same arguments as orc_gi_solve plus
 *   as_in  m+nb activation-status guess (general constraints first, then bounds), or NULL for "no guess".
 * Data equalities (bl == bu, xl == xu) are activated up front (bounds first), the guessed constraints after them;
 * J and R come from a Householder QR of L^-1 N instead of one Givens sweep per constraint; constraints whose
 * multiplier comes out < -1e-14 are dropped one at a time (each counts as one iteration) before the usual loop.
 */
int orc_gi_solve_warm(int n, int m, int nb, double * G, int ldg, const double * a, const double * C, int ldc,
                      const double * bl, const double * bu, const double * xl, const double * xu, const int * as_in,
                      int max_iter, double big_bnd, double * x, double * u, int * status, int * active, int * nactive,
                      int * iters, double * f, orc_trace_rec * trace, int trace_cap, double * work);

void orc_gi_solve_batch_warm(int B, int n, int m, int nb, double * G, const double * a, const double * C,
                             const double * bl, const double * bu, const double * xl, const double * xu,
                             const int * as_in, int max_iter, double big_bnd, double * x, double * u, int * status,
                             int * term, int * iters, double * f, int nthreads);

/* Canonical primitives, exported so tests can pin them one by one. */
int orc_llt(int n, double * A, int lda);                         /* -1 ok, else failing column (Eigen llt_inplace) */
void orc_inv_lt(int n, const double * L, int ldl, double * J, int ldj); /* J = L^-T, dense n x n */
void orc_givens(double p, double q, double * c, double * s, double * r);
/* CANON C7: the coefficients (c_i, s_i), i = n-2 .. q, of an addition's sweep and the final radius, from d[q .. n-1] (jrlqp_oracle.c) */
int orc_givens_chain(int n, int q, const double * d, double * c, double * s, double * rfin);
double orc_dot(int n, const double * a, const double * b);
/* CANON C8: sum of a[k] b[k], k = lo .. n-1: the C1 chain up to 256 entries, beyond: 64-term segments (by absolute index) + the C4 tree */
double orc_dot_blk(int n, const double * a, const double * b, int lo);

/* ActiveSet restatement (ActiveSet.cpp:45-168) driven through a flat API for the integer-sequence test. */
typedef struct orc_active_set orc_active_set;
orc_active_set * orc_as_new(int ncstr, int nbnd);
void orc_as_free(orc_active_set *);
void orc_as_reset(orc_active_set *);
void orc_as_activate(orc_active_set *, int idx, int status);
void orc_as_deactivate(orc_active_set *, int pos);
int orc_as_status(const orc_active_set *, int idx);
int orc_as_at(const orc_active_set *, int pos);
/* counters[8] = {nbActive, me, mi, ml, mu, mb, mbl, mbu}; returns mbe */
int orc_as_counters(const orc_active_set *, int * counters);

/*
 * Structured family.  Blocks are described by flat arrays of pointers/sizes:
 *   diag[i]: ni x ni col-major with ld ldd[i]; side[i]: see blockArrowLLT.h:13-68 of the reference.
 * blockArrow: nblk diagonal blocks, nblk-1 side blocks.
 *   up == 0: side[i] is n_last x n_i (last block row);  up == 1: side[i] is n_{i+1} x n_0 (first block column).
 * Return 1 on success, 0 on failure (with the reference's ret>0 quirk, blockArrowLLT.cpp:65,80).
 */
int orc_block_arrow_llt(int nblk, const int * bs, double ** diag, const int * ldd, double ** side, const int * lds, int up);
void orc_block_arrow_lsolve(int nblk,
                            const int * bs,
                            double ** diag,
                            const int * ldd,
                            double ** side,
                            const int * lds,
                            int up,
                            double * M,
                            int ldm,
                            int nrhs,
                            int start,
                            int end);
void orc_block_arrow_ltsolve(int nblk,
                             const int * bs,
                             double ** diag,
                             const int * ldd,
                             double ** side,
                             const int * lds,
                             int up,
                             double * M,
                             int ldm,
                             int nrhs,
                             int start,
                             int end);
/* triBlockDiag: sub[i] is n_{i+1} x n_i (triBlockDiagLLT.h:13-39). */
int orc_tri_block_diag_llt(int nblk, const int * bs, double ** diag, const int * ldd, double ** sub, const int * lds);
void orc_tri_block_diag_lsolve(int nblk,
                               const int * bs,
                               double ** diag,
                               const int * ldd,
                               double ** sub,
                               const int * lds,
                               double * M,
                               int ldm,
                               int nrhs,
                               int start);
void orc_tri_block_diag_ltsolve(int nblk,
                                const int * bs,
                                double ** diag,
                                const int * ldd,
                                double ** sub,
                                const int * lds,
                                double * M,
                                int ldm,
                                int nrhs,
                                int end);

/*
 * Structured GI loop: experimental::BlockGISolver::solve (src/experimental/BlockGISolver.cpp:17-60) on a StructuredG
 * (type 0 TriBlockDiagonal: off[i] = sub-diagonal block n_{i+1} x n_i; 1 BlockArrowUp; 2 BlockArrowDown: off = side,
 * as for the decompositions above) and a block-diagonal StructuredC (ncb blocks, block i: cvar[i] x ccstr[i],
 * col-major with ld cld[i], one constraint per column).  diag/off are factorised in place like the reference.
 * Cold start without data equalities only -- the only configuration that is functional in the reference
 * (assert at BlockGISolver.cpp:475); anything else returns INCONSISTENT_INPUT.  blockgi_oracle.c.
 */
int orc_blockgi_solve(int type, int nblk, const int * bs, double ** diag, const int * ldd, double ** off, const int * ldo,
                      const double * a, int ncb, const int * cvar, const int * ccstr, const double ** cblk, const int * cld,
                      const double * bl, const double * bu, const double * xl, const double * xu, int nb, int max_iter,
                      double big_bnd, double * x, double * u, int * status, int * active, int * nactive, int * iters,
                      double * f, orc_trace_rec * trace, int trace_cap);

/* per-pass vector trace of the calling thread's next orc_gi_solve calls: record `it` = x [n] | u [m+nb+1] | z [n] | r [m+nb+1] | f | q
 * (the layout of jrlqp_dense_set_trace_vectors; the values the reference logs at DualSolver.cpp:108, :128-129).  NULL / 0: off. */
void orc_set_vector_trace(double * buf, int cap);

/* The same solver under contract C9 (round 5; blockgi_oracle.c): no orthonormal sequence -- the sparse columns B = L^-1 N of the active
 * constraints and S = R^-1 (inverse Cholesky factor of N^T G^-1 N), r and y = b - B r by the corrected semi-normal equations, S updated
 * by a bordered-inverse column on add and by C7 rotations of its columns on remove; reductions over whole vectors by structural block.
 * Same arguments and results as orc_blockgi_solve; decisions identical on every tested problem (tests/test_oracle_blockgi_literal.py). */
int orc_blockgi_solve_c9(int type, int nblk, const int * bs, double ** diag, const int * ldd, double ** off, const int * ldo,
                         const double * a, int ncb, const int * cvar, const int * ccstr, const double ** cblk, const int * cld,
                         const double * bl, const double * bu, const double * xl, const double * xu, int nb, int max_iter,
                         double big_bnd, double * x, double * u, int * status, int * active, int * nactive, int * iters, double * f,
                         orc_trace_rec * trace, int trace_cap);

/* rule C4 (blockgi_oracle.c): sum a[i] b[i] as 1024 strided fma chains (element i has absolute index phase + i)
 * combined by an adjacent-first pairwise tree */
double orc_dot_tree(int n, const double * a, const double * b, int phase);

/* OrthonormalSequence (src/internal/OrthonormalSequence.cpp) as a stand-alone object, for the identities of the
 * reference's tests/InternalTest.cpp */
typedef struct orc_oseq orc_oseq;
orc_oseq * orc_oseq_new(int n);
void orc_oseq_free(orc_oseq *);
double orc_oseq_add_householder(orc_oseq *, int start, int len, const double * v); /* returns beta */
void orc_oseq_add_givens(orc_oseq *, int start, int size, const double * c, const double * s);
void orc_oseq_apply(const orc_oseq *, double * v, int transpose);

#ifdef __cplusplus
}
#endif
#endif
