// The reference's structured-solver tests (tests/BlockGISolverTest.in.cpp:70-170, "Small problem tridiag obj, ineq only"
// and "Small problem arrow up obj, ineq only") written against the host C++ mirrors of its interface
// (include/jrlqp/BlockGISolver.hpp, include/jrlqp/GoldfarbIdnaniSolver.hpp): same construction -- a random block
// factor A, G = A A^T (tridiagonal) or A^T A (arrow), blocks packed as views into the dense G, block-diagonal
// double-sided constraints -- same calls, same expectations: retB == retD and solutionB.isApprox(solutionD, 1e-8).
#include <jrlqp/BlockGISolver.hpp>

#include <cmath>
#include <cstdio>
#include <vector>

using namespace jrlqp;

#define CHECK(c)                                                       \
  do                                                                   \
  {                                                                    \
    if(!(c))                                                           \
    {                                                                  \
      std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                        \
    }                                                                  \
  } while(0)

static unsigned long long g_state = 0x6A726C71ULL;
static double rnd() // uniform in [-1, 1]
{
  g_state = g_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return double((g_state >> 11) & ((1ULL << 53) - 1)) / double(1ULL << 52) - 1.0;
}

static bool isApprox(const std::vector<double> & a, const std::vector<double> & b, double prec)
{
  double d = 0, na = 0, nb = 0;
  for(size_t i = 0; i < a.size(); ++i)
  {
    d += (a[i] - b[i]) * (a[i] - b[i]);
    na += a[i] * a[i];
    nb += b[i] * b[i];
  }
  return std::sqrt(d) <= prec * std::sqrt(na < nb ? na : nb);
}

static int run(StructuredG::Type type, const std::vector<int> & n, const std::vector<int> & mi)
{
  const int nblk = int(n.size());
  std::vector<int> off(nblk + 1, 0);
  for(int i = 0; i < nblk; ++i) off[i + 1] = off[i] + n[i];
  const int N = off[nblk];
  // random factor with the block pattern (biBlockDiagRandom / blockDiagAndOneColDiagRandom)
  std::vector<double> A(size_t(N) * N, 0.0);
  auto fill = [&](int r0, int c0, int rows, int cols) {
    for(int j = 0; j < cols; ++j)
      for(int i = 0; i < rows; ++i) A[(r0 + i) + size_t(c0 + j) * N] = rnd();
  };
  for(int i = 0; i < nblk; ++i)
  {
    fill(off[i], off[i], n[i], n[i]);
    if(type == StructuredG::Type::TriBlockDiagonal && i + 1 < nblk) fill(off[i + 1], off[i], n[i + 1], n[i]);
    if(type == StructuredG::Type::BlockArrowUp) fill(off[i], 0, n[i], n[0]);
  }
  std::vector<double> G(size_t(N) * N, 0.0);
  for(int i = 0; i < N; ++i)
    for(int j = 0; j < N; ++j)
    {
      double s = (i == j) ? 0.05 : 0.0;
      for(int k = 0; k < N; ++k)
        s += (type == StructuredG::Type::TriBlockDiagonal) ? A[i + size_t(k) * N] * A[j + size_t(k) * N]  // A A^T
                                                           : A[k + size_t(i) * N] * A[k + size_t(j) * N]; // A^T A
      G[i + size_t(j) * N] = s;
    }
  std::vector<double> G0 = G, a(N);
  for(double & v : a) v = rnd();
  // block-diagonal double-sided constraints, feasible at a random point
  int M = 0;
  for(int v : mi) M += v;
  std::vector<double> C0(size_t(N) * M, 0.0), l(M), u(M);
  std::vector<MatrixConstRef> Cs;
  for(int i = 0, c = 0; i < nblk; ++i)
  {
    std::vector<double> x0(n[i]);
    for(double & v : x0) v = rnd();
    for(int j = 0; j < mi[i]; ++j)
    {
      double cx = 0;
      for(int r = 0; r < n[i]; ++r)
      {
        const double v = rnd();
        C0[(off[i] + r) + size_t(c + j) * N] = v;
        cx += v * x0[r];
      }
      l[c + j] = cx - 0.1 * std::fabs(rnd());
      u[c + j] = cx + 0.1 * std::fabs(rnd());
    }
    Cs.push_back({C0.data() + off[i] + size_t(c) * N, n[i], mi[i], N});
    c += mi[i];
  }
  // packing as vectors of views into the dense matrix (BlockGISolverTest.in.cpp:78-81, :130-133)
  std::vector<MatrixRef> D, S;
  for(int i = 0; i < nblk; ++i) D.push_back({G.data() + off[i] + size_t(off[i]) * N, n[i], n[i], N});
  for(int i = 0; i + 1 < nblk; ++i)
  {
    if(type == StructuredG::Type::TriBlockDiagonal)
      S.push_back({G.data() + off[i + 1] + size_t(off[i]) * N, n[i + 1], n[i], N});
    else
      S.push_back({G.data() + off[i + 1], n[i + 1], n[0], N});
  }
  StructuredG Gs(type, D, S);
  StructuredC C(Cs);

  GoldfarbIdnaniSolver solverD(N, M, false);
  auto retD = solverD.solve({G0.data(), N, N, N}, {a.data(), N}, {C0.data(), N, M, N}, {l.data(), M}, {u.data(), M}, {nullptr, 0}, {nullptr, 0});

  BlockGISolver solverB(N, M, false);
  auto retB = solverB.solve(Gs, {a.data(), N}, C, {l.data(), M}, {u.data(), M}, {nullptr, 0}, {nullptr, 0});

  CHECK(retB == retD);
  CHECK(retD == TerminationStatus::SUCCESS);
  CHECK(isApprox(solverB.solution(), solverD.solution(), 1e-8));
  CHECK(solverB.iterations() == solverD.iterations());
  for(int i = 0; i < M; ++i) CHECK(solverB.activeSet()[i] == solverD.activeSet()[i]);
  CHECK(solverB.iterations() > 0);
  // the structured factor landed in the caller's blocks: first diagonal block is the Cholesky factor of G's (tridiagonal case)
  if(type == StructuredG::Type::TriBlockDiagonal) CHECK(std::fabs(G[0] - G0[0]) < 1e-15 && G[0] > 0);
  return 0;
}

int main()
{
  for(int rep = 0; rep < 4; ++rep)
  {
    if(run(StructuredG::Type::TriBlockDiagonal, {3, 5, 2, 3}, {3, 3, 3, 3})) return 1;
    if(run(StructuredG::Type::BlockArrowUp, {3, 5, 2, 3}, {3, 3, 3, 3})) return 1;
  }
  if(run(StructuredG::Type::TriBlockDiagonal, {9, 7, 12, 8, 10}, {6, 5, 8, 4, 7})) return 1;
  std::printf("block host class OK\n");
  return 0;
}
