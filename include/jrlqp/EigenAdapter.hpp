// jrlqp/EigenAdapter.hpp -- optional Eigen front end of the host mirror classes ("optional Eigen adapter header
// when Eigen is available").  Compiles to nothing when <Eigen/Core> is not on the include path (it is not in the build image of
// this repository: the header is then only parsed).
//
// With Eigen present a jrl-qp user keeps the reference's call syntax:
//
//   jrlqp::eigen::GoldfarbIdnaniSolver qp(n, m, useBounds);
//   auto st = qp.solve(G, a, C, bl, bu, xl, xu);             // Eigen::Ref arguments, exactly jrl::qp::GoldfarbIdnaniSolver::solve
//   qp.solution(), qp.multipliers()                          // Eigen::Map views of the results
//
// The arguments are the reference's typedefs (include/jrl-qp/defs.h:11-14): MatrixRef = Eigen::Ref<MatrixXd> (column-major, inner
// stride 1, any outer stride), VectorConstRef = const Eigen::Ref<const VectorXd>&.  Like Eigen::Ref<const ...> in the reference,
// a non-conforming const argument (e.g. C.transpose()) is silently evaluated into a temporary by Eigen before it gets here.
#pragma once

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#define JRLQP_HAVE_EIGEN 1
#endif
#endif

#ifdef JRLQP_HAVE_EIGEN
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
#endif // JRLQP_HAVE_EIGEN
