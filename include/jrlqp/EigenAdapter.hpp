// EigenAdapter.hpp -- the reference's Eigen-typed call syntax over the host mirror class (include/jrl-qp/GoldfarbIdnaniSolver.h:27-33,
// include/jrl-qp/defs.h:11-14).  The whole header is guarded: where Eigen's own headers are not installed (this repository's build
// image) it declares nothing, so it cannot rot unseen -- tests/cpp/test_eigen_adapter.cpp instantiates it wherever Eigen IS present
// and tests/test_host_class.py skips that test elsewhere (no stand-in for Eigen is ever written).  ABI 3 removed the unguarded
// round-3 version; this one restores the drop-in surface.
#pragma once
#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#define JRLQP_HAVE_EIGEN 1
#include <Eigen/Core>

#include "GoldfarbIdnaniSolver.hpp"

namespace jrlqp::eigen
{
using MatrixRef = Eigen::Ref<Eigen::MatrixXd>;
using MatrixConstRef = Eigen::Ref<const Eigen::MatrixXd>;
using VectorConstRef = Eigen::Ref<const Eigen::VectorXd>;

class GoldfarbIdnaniSolver
{
public:
  GoldfarbIdnaniSolver() = default;
  GoldfarbIdnaniSolver(int nbVar, int nbCstr, bool useBounds) : impl_(nbVar, nbCstr, useBounds) {}
  void resize(int nbVar, int nbCstr, bool useBounds) { impl_.resize(nbVar, nbCstr, useBounds); }
  void options(const SolverOptions & o) { impl_.options(o); }
  const SolverOptions & options() const { return impl_.options(); }

  /** GoldfarbIdnaniSolver::solve(MatrixRef G, a, C, bl, bu, xl, xu) (include/jrl-qp/GoldfarbIdnaniSolver.h:27-33): G is overwritten
   *  by its Cholesky factor (lower triangle), C is nbVar x nbCstr. */
  TerminationStatus solve(MatrixRef G, const VectorConstRef & a, const MatrixConstRef & C, const VectorConstRef & bl, const VectorConstRef & bu,
                          const VectorConstRef & xl, const VectorConstRef & xu)
  {
    return impl_.solve({G.data(), int(G.rows()), int(G.cols()), int(G.outerStride())}, {a.data(), int(a.size())},
                       {C.data(), int(C.rows()), int(C.cols()), int(C.outerStride())}, {bl.data(), int(bl.size())}, {bu.data(), int(bu.size())},
                       {xl.data(), int(xl.size())}, {xu.data(), int(xu.size())});
  }
  Eigen::Map<const Eigen::VectorXd> solution() const { return {impl_.solution().data(), Eigen::Index(impl_.solution().size())}; }
  Eigen::Map<const Eigen::VectorXd> multipliers() const { return {impl_.multipliers().data(), Eigen::Index(impl_.multipliers().size())}; }
  double objectiveValue() const { return impl_.objectiveValue(); }
  int iterations() const { return impl_.iterations(); }
  const std::vector<ActivationStatus> & activeSet() const { return impl_.activeSet(); }
  void resetActiveSet() { impl_.resetActiveSet(); }
  const std::string & lastError() const { return impl_.lastError(); }

private:
  jrlqp::GoldfarbIdnaniSolver impl_;
};
} // namespace jrlqp::eigen
#endif // __has_include(<Eigen/Core>)
#endif // defined(__has_include)
