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
// The numerics run on the GPU (libjrlqp_hip.so, Mode A: state in HBM, host-driven active-set loop).  solve() copies
// the problem to the device and the factor back (the reference's "G is clobbered" contract); solveDevice() takes
// device pointers and copies nothing.  Header-only; link with -ljrlqp_hip and the HIP runtime.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <stdexcept>
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

  /** Pre-allocate for nbVar variables, nbCstr general constraints, and bounds if useBounds (DualSolver.h:33). */
  void resize(int nbVar, int nbCstr, bool useBounds)
  {
    const int nb = useBounds ? nbVar : 0;
    if(h_ && nbVar == n_ && nbCstr == m_ && nb == nb_) return;
    release();
    n_ = nbVar; m_ = nbCstr; nb_ = nb;
    check(jrlqp_dense_create(n_, m_, nb_, &h_), "jrlqp_dense_create");
    const size_t n = size_t(n_), m = size_t(m_);
    hip(hipMalloc(&dG_, sizeof(double) * n * n));
    hip(hipMalloc(&da_, sizeof(double) * n));
    hip(hipMalloc(&dC_, sizeof(double) * (n * m ? n * m : 1)));
    hip(hipMalloc(&dbl_, sizeof(double) * (m ? m : 1)));
    hip(hipMalloc(&dbu_, sizeof(double) * (m ? m : 1)));
    hip(hipMalloc(&dxl_, sizeof(double) * (nb_ ? nb_ : 1)));
    hip(hipMalloc(&dxu_, sizeof(double) * (nb_ ? nb_ : 1)));
    x_.assign(n, 0.0);
    u_.assign(m + nb_, 0.0);
    as_.assign(m + nb_, ActivationStatus::INACTIVE);
    ias_.assign(m + nb_, 0);
  }

  void options(const SolverOptions & o) { options_ = o; }
  const SolverOptions & options() const { return options_; }

  /** min 1/2 x^T G x + a^T x  s.t.  bl <= C^T x <= bu, xl <= x <= xu  (GoldfarbIdnaniSolver.h:22-33).
   *  HOST views.  C is n x m, one constraint per column; xl/xu of size n or 0. */
  TerminationStatus solve(MatrixRef G, VectorConstRef a, MatrixConstRef C, VectorConstRef bl, VectorConstRef bu, VectorConstRef xl,
                          VectorConstRef xu)
  {
    resize(G.rows, C.cols, xl.size > 0);
    if(G.cols != n_ || a.size != n_ || (m_ && C.rows != n_) || bl.size != m_ || bu.size != m_ || xu.size != xl.size)
      return TerminationStatus::INCONSISTENT_INPUT;
    const size_t n = size_t(n_);
    hip(hipMemcpy2D(dG_, sizeof(double) * n, G.data, sizeof(double) * size_t(G.ld), sizeof(double) * n, n, hipMemcpyHostToDevice));
    hip(hipMemcpy(da_, a.data, sizeof(double) * n, hipMemcpyHostToDevice));
    if(m_)
    {
      hip(hipMemcpy2D(dC_, sizeof(double) * n, C.data, sizeof(double) * size_t(C.ld), sizeof(double) * n, size_t(m_), hipMemcpyHostToDevice));
      hip(hipMemcpy(dbl_, bl.data, sizeof(double) * m_, hipMemcpyHostToDevice));
      hip(hipMemcpy(dbu_, bu.data, sizeof(double) * m_, hipMemcpyHostToDevice));
    }
    if(nb_)
    {
      hip(hipMemcpy(dxl_, xl.data, sizeof(double) * nb_, hipMemcpyHostToDevice));
      hip(hipMemcpy(dxu_, xu.data, sizeof(double) * nb_, hipMemcpyHostToDevice));
    }
    TerminationStatus st = solveDevice(dG_, n_, da_, dC_, n_, dbl_, dbu_, dxl_, dxu_);
    // the reference leaves the Cholesky factor in G's lower triangle (strictly upper part untouched)
    hip(hipMemcpy2D(G.data, sizeof(double) * size_t(G.ld), dG_, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyDeviceToHost));
    return st;
  }

  /** Same with DEVICE pointers: nothing is copied but the results. */
  TerminationStatus solveDevice(double * dG, int ldg, const double * da, const double * dC, int ldc, const double * dbl, const double * dbu,
                                const double * dxl, const double * dxu, hipStream_t stream = nullptr)
  {
    jrlqp_options o{options_.maxIter_, options_.bigBnd_};
    int term = int(TerminationStatus::UNKNOWN);
    check(jrlqp_dense_solve(h_, dG, ldg, da, dC, ldc, dbl, dbu, dxl, dxu, &o, &term, stream), "jrlqp_dense_solve");
    check(jrlqp_dense_results(h_, x_.data(), u_.data(), ias_.data(), &it_, &f_, stream), "jrlqp_dense_results");
    for(size_t i = 0; i < ias_.size(); ++i) as_[i] = static_cast<ActivationStatus>(ias_[i]);
    return static_cast<TerminationStatus>(term);
  }

  const std::vector<double> & solution() const { return x_; }                  // DualSolver::solution()
  const std::vector<double> & multipliers() const { return u_; }               // DualSolver::multipliers(), sign-expanded
  double objectiveValue() const { return f_; }                                 // DualSolver::objectiveValue()
  int iterations() const { return it_; }                                       // DualSolver::iterations()
  const std::vector<ActivationStatus> & activeSet() const { return as_; }      // DualSolver::activeSet()
  void resetActiveSet() {}                                                     // the production solver always cold-starts

private:
  static void check(int rc, const char * what)
  {
    if(rc != JRLQP_OK) throw std::runtime_error(std::string(what) + ": " + jrlqp_status_string(rc));
  }
  static void hip(hipError_t e)
  {
    if(e != hipSuccess) throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e));
  }
  void release()
  {
    if(h_) jrlqp_dense_destroy(h_);
    h_ = nullptr;
    for(double ** p : {&dG_, &da_, &dC_, &dbl_, &dbu_, &dxl_, &dxu_})
    {
      if(*p) (void)hipFree(*p);
      *p = nullptr;
    }
  }

  SolverOptions options_;
  jrlqp_dense * h_ = nullptr;
  int n_ = 0, m_ = 0, nb_ = 0, it_ = 0;
  double f_ = 0.0;
  double *dG_ = nullptr, *da_ = nullptr, *dC_ = nullptr, *dbl_ = nullptr, *dbu_ = nullptr, *dxl_ = nullptr, *dxu_ = nullptr;
  std::vector<double> x_, u_;
  std::vector<ActivationStatus> as_;
  std::vector<int> ias_;
};
} // namespace jrlqp
