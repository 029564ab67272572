// The reference's "Simple problem paper" test (tests/GoldfarbIdnaniSolverTest.cpp:51-73) written against the host
// C++ mirror of its interface (include/jrlqp/GoldfarbIdnaniSolver.hpp): same calls, same expectations.
#include <jrlqp/GoldfarbIdnaniSolver.hpp>

#include <cmath>
#include <cstdio>

using namespace jrlqp;

#define CHECK(c)                                              \
  do                                                          \
  {                                                           \
    if(!(c))                                                  \
    {                                                         \
      std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                               \
    }                                                         \
  } while(0)

int main()
{
  // example from the Goldfarb-Idnani paper
  double G[4] = {4, -2, -2, 4};
  double a[2] = {6, 0};
  double C[2] = {1, 1};
  double bl[1] = {2}, bu[1] = {10};
  double xl[2] = {0, 0}, xu[2] = {10, 10};

  GoldfarbIdnaniSolver qp(2, 1, true);
  auto st = qp.solve({G, 2, 2, 2}, {a, 2}, {C, 2, 1, 2}, {bl, 1}, {bu, 1}, {xl, 2}, {xu, 2});
  CHECK(st == TerminationStatus::SUCCESS);
  CHECK(std::fabs(qp.solution()[0] - 0.5) < 1e-14 && std::fabs(qp.solution()[1] - 1.5) < 1e-14);
  CHECK(std::fabs(qp.objectiveValue() - 6.5) < 1e-14);
  CHECK(qp.iterations() == 1);
  CHECK(std::fabs(qp.multipliers()[0] + 5.0) < 1e-13 && qp.multipliers()[1] == 0.0 && qp.multipliers()[2] == 0.0);
  CHECK(qp.activeSet()[0] == ActivationStatus::LOWER && qp.activeSet()[1] == ActivationStatus::INACTIVE);
  // G was overwritten by its Cholesky factor: L = [2 0; -1 sqrt(3)], strictly upper part untouched
  CHECK(G[0] == 2.0 && G[1] == -1.0 && G[2] == -2.0 && std::fabs(G[3] - std::sqrt(3.0)) < 1e-15);

  // "Multiple uses" (:101-125): the same pre-sized object solves again; options are honoured
  double G2[4] = {4, -2, -2, 4};
  qp.options(SolverOptions().maxIter(0));
  st = qp.solve({G2, 2, 2, 2}, {a, 2}, {C, 2, 1, 2}, {bl, 1}, {bu, 1}, {xl, 2}, {xu, 2});
  CHECK(st == TerminationStatus::MAX_ITER_REACHED && qp.iterations() == 0);
  double G3[4] = {4, -2, -2, -4};
  qp.options(SolverOptions());
  st = qp.solve({G3, 2, 2, 2}, {a, 2}, {C, 2, 1, 2}, {bl, 1}, {bu, 1}, {xl, 2}, {xu, 2});
  CHECK(st == TerminationStatus::NON_POS_HESSIAN);
  std::printf("host class OK\n");
  return 0;
}
