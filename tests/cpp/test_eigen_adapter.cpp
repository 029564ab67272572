// The Eigen-typed front end (include/jrlqp/EigenAdapter.hpp) on the GI paper's example (tests/GoldfarbIdnaniSolverTest.cpp:51-73) -- built
// and run only where Eigen's headers are installed (tests/test_host_class.py probes for them and skips otherwise).
#include <jrlqp/EigenAdapter.hpp>

#include <cmath>
#include <cstdio>

#ifndef JRLQP_HAVE_EIGEN
int main()
{
  std::puts("SKIP: Eigen not installed");
  return 77;
}
#else
int main()
{
  Eigen::MatrixXd G(2, 2);
  G << 4, -2, -2, 4;
  Eigen::VectorXd a(2), bl(3), bu(3), xl(0), xu(0);
  a << 6, 0;
  Eigen::MatrixXd C(2, 3);
  C << 1, 0, 1, 1, 1, 0; // rows of C^T: x1 + x2 >= 2, x2 >= 0, x1 >= 0  (column-wise: one constraint per column)
  bl << 2, 0, 0;
  bu << 1e100, 1e100, 1e100;
  jrlqp::eigen::GoldfarbIdnaniSolver solver(2, 3, false);
  const auto st = solver.solve(G, a, C, bl, bu, xl, xu);
  const auto x = solver.solution();
  const bool ok = st == jrlqp::TerminationStatus::SUCCESS && std::fabs(x[0] - 0.5) < 1e-12 && std::fabs(x[1] - 1.5) < 1e-12 &&
                  std::fabs(solver.objectiveValue() - 6.5) < 1e-12 && solver.iterations() == 1;
  std::printf("%s x = (%g, %g) f = %g\n", ok ? "OK" : "FAIL", x[0], x[1], solver.objectiveValue());
  return ok ? 0 : 1;
}
#endif
