// jrlqp/GoldfarbIdnaniSolver.hpp -- host C++17 mirror of the reference's solver interface over the C ABI.
//
// Same public surface and semantics as jrl::qp::GoldfarbIdnaniSolver / jrl::qp::DualSolver
// (include/jrl-qp/GoldfarbIdnaniSolver.h:15-33, include/jrl-qp/DualSolver.h:26-60), with raw column-major views
// {pointer, rows, cols, leading dimension} where the reference takes Eigen::Ref (include/jrl-qp/defs.h:11-14):
//
//   jrlqp::GoldfarbIdnaniSolver qp(n, m, useBounds);          // pre-sized: solve() allocates nothing
//   auto st = qp.solve(G, a, C, bl, bu, xl, xu);              // HOST views; G is overwritten by its Cholesky factor
//   qp.solution(), qp.multipliers(), qp.objectiveValue(), qp.iterations(), qp.activeSet()
//
// The numerics run on the GPU (libjrlqp_hip.so): n <= 128 through the device-resident batched kernels with a batch of
// one (one launch per solve), larger problems through Mode A (state in HBM, host-driven active-set loop).  solve() copies
// the problem to the device and the factor back (the reference's "G is clobbered" contract); solveDevice() takes
// device pointers and copies nothing.  Contract of the reference kept: no exception leaves the class, and once sized
// the solver allocates nothing -- buffers are grow-only (tests/cpp/test_multiple_uses.cpp counts hipMalloc / hipFree).
// Header-only; link with -ljrlqp_hip and the HIP runtime.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../jrlqp_hip.h"

namespace jrlqp
{
enum class ActivationStatus : int // include/jrl-qp/enums.h:15-23
{
  INACTIVE = 0,
  LOWER,
  UPPER,
  EQUALITY,
  LOWER_BOUND,
  UPPER_BOUND,
  FIXED
};

enum class TerminationStatus : int // include/jrl-qp/enums.h:26-37
{
  SUCCESS = 0,
  INCONSISTENT_INPUT,
  NON_POS_HESSIAN,
  INFEASIBLE,
  MAX_ITER_REACHED,
  LINEAR_DEPENDENCY_DETECTED,
  OVERCONSTRAINED_PROBLEM,
  UNKNOWN
};

struct SolverOptions // include/jrl-qp/SolverOptions.h:14-88 (logging fields have no GPU counterpart)
{
  int maxIter_ = 500;
  double bigBnd_ = 1e100;
  bool warmStart_ = false; // ignored, as by the reference's production solver (GoldfarbIdnaniSolver.cpp:75)
  int maxIter() const { return maxIter_; }
  SolverOptions & maxIter(int m) { maxIter_ = m; return *this; }
  double bigBnd() const { return bigBnd_; }
  SolverOptions & bigBnd(double b) { bigBnd_ = b; return *this; }
  bool warmStart() const { return warmStart_; }
  SolverOptions & warmStart(bool w) { warmStart_ = w; return *this; }
};

struct MatrixRef // column-major view, inner stride 1
{
  double * data;
  int rows, cols, ld;
};
struct MatrixConstRef
{
  const double * data;
  int rows, cols, ld;
};
struct VectorConstRef
{
  const double * data;
  int size;
};

class GoldfarbIdnaniSolver
{
public:
  GoldfarbIdnaniSolver() = default;
  GoldfarbIdnaniSolver(int nbVar, int nbCstr, bool useBounds) { resize(nbVar, nbCstr, useBounds); }
  GoldfarbIdnaniSolver(const GoldfarbIdnaniSolver &) = delete;
  GoldfarbIdnaniSolver & operator=(const GoldfarbIdnaniSolver &) = delete;
  ~GoldfarbIdnaniSolver() { release(); }

  /** Pre-allocate for nbVar variables, nbCstr general constraints, and bounds if useBounds (DualSolver.h:33).
   *  GROW-ONLY like the reference's workspaces (internal/Workspace.h:37-43): a problem that fits what is already
   *  allocated -- fewer variables, fewer constraints, no bounds -- reuses it; nothing is freed before the destructor
   *  unless a larger problem arrives.  Never throws: a failed allocation is reported by the next solve() (UNKNOWN) and
   *  by lastError(). */
  void resize(int nbVar, int nbCstr, bool useBounds)
  {
    const int nb = useBounds ? nbVar : 0;
    if(nbVar < 1 || nbCstr < 0) { fail("resize: invalid sizes"); return; }
    if(!h_ || nbVar > capN_ || nbCstr > capM_ || nb > capNb_)
    {
      // grow: capacities never shrink (a solver pre-sized for (5, 8, true) keeps serving (5, 2, false) and (5, 8, true))
      const int cn = nbVar > capN_ ? nbVar : capN_, cm = nbCstr > capM_ ? nbCstr : capM_;
      const int cb = (nb > 0 || capNb_ > 0) ? cn : 0;
      release();
      capN_ = cn; capM_ = cm; capNb_ = cb;
      if(!ok(jrlqp_dense_create(cn, cm, cb, &h_), "jrlqp_dense_create")) { h_ = nullptr; return; }
      const size_t n = size_t(cn), m = size_t(cm), mt = m + size_t(cb);
      bool a = dev(&dG_, n * n) && dev(&da_, n) && dev(&dC_, n * m) && dev(&dbl_, m) && dev(&dbu_, m) && dev(&dxl_, cb) && dev(&dxu_, cb);
      a = a && dev(&dx_, n) && dev(&du_, mt) && dev(&dfv_, 1) && devi(&dst_, mt) && devi(&dti_, 2);
      // workspace of the batched path (jrlqp_gi_batch_workspace_bytes; > 0 for 104 < n <= 128 only): sized HERE for the largest
      // n the capacity admits -- it grows with n --, so that solve() never allocates
      wsBytes_ = 0;
      for(int k = (cn < JRLQP_BATCH_LDS_N ? cn : JRLQP_BATCH_LDS_N); k > JRLQP_BATCH_WAVE_N && wsBytes_ == 0; --k)
      {
        const long long b = jrlqp_gi_batch_workspace_bytes(k, 0, 0, 1);
        if(b > 0) wsBytes_ = b;
        if(b == 0) break;
      }
      if(wsBytes_ > 0) a = a && dev(&dws_, size_t(wsBytes_ + 7) / 8);
      if(!a) return;
      x_.reserve(n); u_.reserve(mt); as_.reserve(mt); ias_.reserve(mt);
    }
    if(!ok(jrlqp_dense_resize(h_, nbVar, nbCstr, nb), "jrlqp_dense_resize")) return;
    n_ = nbVar; m_ = nbCstr; nb_ = nb;
    // (within the reserved capacity: no allocation)
    x_.assign(size_t(n_), 0.0);
    u_.assign(size_t(m_) + nb_, 0.0);
    as_.assign(size_t(m_) + nb_, ActivationStatus::INACTIVE);
    ias_.assign(size_t(m_) + nb_, 0);
    err_.clear();
  }

  void options(const SolverOptions & o) { options_ = o; }
  const SolverOptions & options() const { return options_; }

  /** min 1/2 x^T G x + a^T x  s.t.  bl <= C^T x <= bu, xl <= x <= xu  (GoldfarbIdnaniSolver.h:22-33).
   *  HOST views.  C is n x m, one constraint per column; xl/xu of size n or 0.  Like the reference it never throws and
   *  (on a pre-sized solver) never allocates; size mismatches the reference only assert()s on (GoldfarbIdnaniSolver.cpp:33-39)
   *  return INCONSISTENT_INPUT, a failing runtime call returns UNKNOWN (see lastError()). */
  TerminationStatus solve(MatrixRef G, VectorConstRef a, MatrixConstRef C, VectorConstRef bl, VectorConstRef bu, VectorConstRef xl,
                          VectorConstRef xu) noexcept
  {
    if(G.rows < 1 || G.cols != G.rows || a.size != G.rows || (C.cols && C.rows != G.rows) || bl.size != C.cols || bu.size != C.cols
       || xu.size != xl.size || (xl.size != 0 && xl.size != G.rows) || G.ld < G.rows || (C.cols && C.ld < C.rows))
      return TerminationStatus::INCONSISTENT_INPUT;
    resize(G.rows, C.cols, xl.size > 0);
    if(!err_.empty()) return TerminationStatus::UNKNOWN;
    const size_t n = size_t(n_);
    bool c = hipok(hipMemcpy2D(dG_, sizeof(double) * n, G.data, sizeof(double) * size_t(G.ld), sizeof(double) * n, n, hipMemcpyHostToDevice));
    c = c && hipok(hipMemcpy(da_, a.data, sizeof(double) * n, hipMemcpyHostToDevice));
    if(m_)
    {
      c = c && hipok(hipMemcpy2D(dC_, sizeof(double) * n, C.data, sizeof(double) * size_t(C.ld), sizeof(double) * n, size_t(m_), hipMemcpyHostToDevice));
      c = c && hipok(hipMemcpy(dbl_, bl.data, sizeof(double) * m_, hipMemcpyHostToDevice));
      c = c && hipok(hipMemcpy(dbu_, bu.data, sizeof(double) * m_, hipMemcpyHostToDevice));
    }
    if(nb_)
    {
      c = c && hipok(hipMemcpy(dxl_, xl.data, sizeof(double) * nb_, hipMemcpyHostToDevice));
      c = c && hipok(hipMemcpy(dxu_, xu.data, sizeof(double) * nb_, hipMemcpyHostToDevice));
    }
    if(!c) return TerminationStatus::UNKNOWN;
    TerminationStatus st = solveDevice(dG_, n_, da_, dC_, n_, dbl_, dbu_, dxl_, dxu_);
    // the reference leaves the Cholesky factor in G's lower triangle (strictly upper part untouched)
    if(!hipok(hipMemcpy2D(G.data, sizeof(double) * size_t(G.ld), dG_, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyDeviceToHost)))
      return TerminationStatus::UNKNOWN;
    return st;
  }

  /** Same with DEVICE pointers: nothing is copied but the results.  Sizes are those of the last resize().
   *  Problems the device-resident batched kernels take (n <= 128, packed G and C) run there with a batch of one -- one launch
   *  for the whole solve instead of one host round trip per iteration (n = 100: ~2 ms against ~6 ms) --, larger ones through
   *  the host-driven hooks (Mode A); both leave the factor in G and give bit-identical results.  Beyond n = 104 the batched
   *  kernel keeps R in a workspace: it was allocated by resize() (jrlqp_gi_batch_workspace_bytes), so a sized solver never
   *  allocates inside solve() on either path. */
  TerminationStatus solveDevice(double * dG, int ldg, const double * da, const double * dC, int ldc, const double * dbl, const double * dbu,
                                const double * dxl, const double * dxu, hipStream_t stream = nullptr) noexcept
  {
    if(!h_) { fail("solveDevice: solver not sized"); return TerminationStatus::UNKNOWN; }
    jrlqp_options o{options_.maxIter_, options_.bigBnd_};
    int term = int(TerminationStatus::UNKNOWN);
    int cpl = 0;
    const long long wsNeed = (n_ <= JRLQP_BATCH_LDS_N) ? jrlqp_gi_batch_workspace_bytes(n_, m_, nb_, 1) : -1; // (a lone larger problem: Mode A)
    if(wsNeed >= 0 && wsNeed <= wsBytes_ && ldg == n_ && (m_ == 0 || ldc == n_) && jrlqp_gi_batch_lds_bytes(n_, m_, nb_, &cpl) >= 0)
    {
      // dL == dG: the kernel reads G completely before it writes the factor (jrlqp_hip.h), i.e. the reference's in-place contract
      if(!ok(jrlqp_gi_solve_batch_ws(0, n_, m_, nb_, 1, dG, da, dC, dbl, dbu, dxl, dxu, nullptr, &o, dx_, du_, dst_, dti_, dti_ + 1, dfv_, dG,
                                     nullptr, 0, wsNeed > 0 ? dws_ : nullptr, wsNeed > 0 ? wsBytes_ : 0, stream),
             "jrlqp_gi_solve_batch_ws"))
        return TerminationStatus::UNKNOWN;
      int ti[2] = {int(TerminationStatus::UNKNOWN), 0};
      bool c = hipok(hipMemcpyAsync(x_.data(), dx_, sizeof(double) * n_, hipMemcpyDeviceToHost, stream));
      if(m_ + nb_) c = c && hipok(hipMemcpyAsync(u_.data(), du_, sizeof(double) * (m_ + nb_), hipMemcpyDeviceToHost, stream));
      if(m_ + nb_) c = c && hipok(hipMemcpyAsync(ias_.data(), dst_, sizeof(int) * (m_ + nb_), hipMemcpyDeviceToHost, stream));
      c = c && hipok(hipMemcpyAsync(ti, dti_, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
      c = c && hipok(hipMemcpyAsync(&f_, dfv_, sizeof(double), hipMemcpyDeviceToHost, stream));
      c = c && hipok(hipStreamSynchronize(stream));
      if(!c) return TerminationStatus::UNKNOWN;
      term = ti[0];
      it_ = ti[1];
    }
    else
    {
      if(!ok(jrlqp_dense_solve(h_, dG, ldg, da, dC, ldc, dbl, dbu, dxl, dxu, &o, &term, stream), "jrlqp_dense_solve")) return TerminationStatus::UNKNOWN;
      if(!ok(jrlqp_dense_results(h_, x_.data(), u_.data(), ias_.data(), &it_, &f_, stream), "jrlqp_dense_results")) return TerminationStatus::UNKNOWN;
    }
    for(size_t i = 0; i < ias_.size(); ++i) as_[i] = static_cast<ActivationStatus>(ias_[i]);
    return static_cast<TerminationStatus>(term);
  }

  const std::vector<double> & solution() const { return x_; }                  // DualSolver::solution()
  const std::vector<double> & multipliers() const { return u_; }               // DualSolver::multipliers(), sign-expanded
  double objectiveValue() const { return f_; }                                 // DualSolver::objectiveValue()
  int iterations() const { return it_; }                                       // DualSolver::iterations()
  const std::vector<ActivationStatus> & activeSet() const { return as_; }      // DualSolver::activeSet()
  void resetActiveSet() {}                                                     // the production solver always cold-starts
  /** Empty unless an ABI / HIP runtime call failed (the reference has no such failure mode; it never throws either). */
  const std::string & lastError() const { return err_; }

private:
  void fail(const char * what) noexcept
  {
    try { err_ = what; } catch(...) {}
  }
  bool ok(int rc, const char * what) noexcept
  {
    if(rc == JRLQP_OK) return true;
    try { err_ = std::string(what) + ": " + jrlqp_status_string(rc); } catch(...) {}
    return false;
  }
  bool hipok(hipError_t e) noexcept
  {
    if(e == hipSuccess) return true;
    try { err_ = std::string("HIP: ") + hipGetErrorString(e); } catch(...) {}
    return false;
  }
  bool dev(double ** p, size_t count) noexcept { return hipok(hipMalloc(reinterpret_cast<void **>(p), sizeof(double) * (count ? count : 1))); }
  bool devi(int ** p, size_t count) noexcept { return hipok(hipMalloc(reinterpret_cast<void **>(p), sizeof(int) * (count ? count : 1))); }
  void release() noexcept
  {
    if(h_) jrlqp_dense_destroy(h_);
    h_ = nullptr;
    wsBytes_ = 0;
    for(double ** p : {&dG_, &da_, &dC_, &dbl_, &dbu_, &dxl_, &dxu_, &dx_, &du_, &dfv_, &dws_})
    {
      if(*p) (void)hipFree(*p);
      *p = nullptr;
    }
    for(int ** p : {&dst_, &dti_})
    {
      if(*p) (void)hipFree(*p);
      *p = nullptr;
    }
  }

  SolverOptions options_;
  jrlqp_dense * h_ = nullptr;
  int n_ = 0, m_ = 0, nb_ = 0, it_ = 0;
  int capN_ = 0, capM_ = 0, capNb_ = 0; // allocated capacities (grow-only)
  double f_ = 0.0;
  double *dG_ = nullptr, *da_ = nullptr, *dC_ = nullptr, *dbl_ = nullptr, *dbu_ = nullptr, *dxl_ = nullptr, *dxu_ = nullptr;
  double *dx_ = nullptr, *du_ = nullptr, *dfv_ = nullptr; // outputs of the batched (B = 1) path
  double * dws_ = nullptr;                                // its workspace (105 <= n <= 128), wsBytes_ bytes
  long long wsBytes_ = 0;
  int *dst_ = nullptr, *dti_ = nullptr;                   // activation status; {termination status, iterations}
  std::vector<double> x_, u_;
  std::vector<ActivationStatus> as_;
  std::vector<int> ias_;
  std::string err_;
};
} // namespace jrlqp
