// eigen_crosscheck.cpp -- OPTIONAL cross-check of the oracle's primitives against Eigen itself ("if Eigen headers
// happen to exist on a box ... compile an Eigen-backed variant of the restatement's six primitives for an extra cross-check").
//
// TEST INFRASTRUCTURE ONLY, and only where Eigen is installed: this image has no Eigen, so `make -C oracle eigen`
// prints "Eigen not found" and builds nothing, and tests/test_oracle_eigen.py skips.  Nothing here is a stand-in for Eigen: the
// file includes the real headers or is not compiled.
//
// The six primitives are the Eigen call sites of the hot path: llt_inplace (GoldfarbIdnaniSolver.cpp:58),
// triangular solves (:66, :70-71, :146), dot / GEMV (ConstraintNormal.h:87-113, GoldfarbIdnaniSolver.cpp:97,145),
// JacobiRotation::makeGivens and its application (:228-231, :247-252), makeHouseholder / applyHouseholderOnTheLeft
// (experimental/GoldfarbIdnaniSolver.cpp:432-440).  Eigen's summation order is unspecified, so the comparison is to a tolerance of
// a few ulps times the condition of the operation -- except makeGivens, whose arithmetic is explicit in Eigen and must agree
// BIT FOR BIT with orc_givens.
#include <Eigen/Dense>
#include <Eigen/Jacobi>

#include <cmath>
#include <cstdio>
#include <random>

extern "C" {
#include "jrlqp_oracle.h"
}

using Eigen::MatrixXd;
using Eigen::VectorXd;

static int failures = 0;
#define EXPECT(cond, what)                                        \
  do                                                              \
  {                                                               \
    if(!(cond))                                                   \
    {                                                             \
      std::printf("FAILED %s (%s:%d)\n", what, __FILE__, __LINE__); \
      ++failures;                                                 \
    }                                                             \
  } while(0)

int main()
{
  std::mt19937_64 rng(20260703);
  std::normal_distribution<double> N01(0.0, 1.0);
  for(int n : {1, 2, 5, 10, 33, 50, 64, 100})
  {
    MatrixXd A(n + 3, n);
    for(int j = 0; j < n; ++j)
      for(int i = 0; i < n + 3; ++i) A(i, j) = N01(rng);
    const MatrixXd G = A.transpose() * A;
    const double scale = G.norm();
    // 1. LLT: Eigen::internal::llt_inplace<double, Lower>::blocked vs orc_llt
    MatrixXd Le = G, Lo = G;
    const Eigen::Index ret = Eigen::internal::llt_inplace<double, Eigen::Lower>::blocked(Le);
    const int info = orc_llt(n, Lo.data(), n);
    EXPECT(ret == -1 && info == -1, "llt succeeds on an SPD matrix");
    const MatrixXd Ltri_e = Le.triangularView<Eigen::Lower>(), Ltri_o = Lo.triangularView<Eigen::Lower>();
    EXPECT((Ltri_e - Ltri_o).norm() <= 1e-12 * std::sqrt(scale) * n, "orc_llt == Eigen llt_inplace (tolerance)");
    // 2. J = L^-T: J.setIdentity(); L.transpose().solveInPlace(J)  vs orc_inv_lt
    MatrixXd Je = MatrixXd::Identity(n, n), Jo(n, n);
    Ltri_o.triangularView<Eigen::Lower>().transpose().solveInPlace(Je);
    orc_inv_lt(n, Lo.data(), n, Jo.data(), n);
    EXPECT((Je - Jo).norm() <= 1e-11 * Je.norm() * n, "orc_inv_lt == Eigen triangular solve (tolerance)");
    // 3. dot / GEMV-T
    VectorXd x(n), y(n);
    for(int i = 0; i < n; ++i)
    {
      x[i] = N01(rng);
      y[i] = N01(rng);
    }
    EXPECT(std::abs(x.dot(y) - orc_dot(n, x.data(), y.data())) <= 1e-14 * (x.cwiseAbs().dot(y.cwiseAbs()) + 1e-300) * n, "orc_dot == Eigen dot (tolerance)");
    EXPECT(std::abs(x.dot(y) - orc_dot_tree(n, x.data(), y.data(), 0)) <= 1e-14 * (x.cwiseAbs().dot(y.cwiseAbs()) + 1e-300) * n,
           "orc_dot_tree == Eigen dot (tolerance)");
    // 4. makeGivens: explicit arithmetic in Eigen -> bit for bit, including the special cases
    for(int k = 0; k < 200; ++k)
    {
      double p = N01(rng) * std::pow(10.0, (k % 7) - 3), q = N01(rng);
      if(k % 11 == 0) q = 0.0;
      if(k % 13 == 0) p = 0.0;
      Eigen::JacobiRotation<double> Gr;
      double re = 0.0;
      Gr.makeGivens(p, q, &re);
      double c, s, r;
      orc_givens(p, q, &c, &s, &r);
      EXPECT(c == Gr.c() && s == Gr.s() && r == re, "orc_givens == JacobiRotation::makeGivens (bit for bit)");
    }
  }
  if(failures == 0) std::printf("eigen cross-check OK (Eigen %d.%d.%d)\n", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION);
  return failures ? 1 : 0;
}
