// The reference's "Multiple uses" test (tests/GoldfarbIdnaniSolverTest.cpp:101-125) against the host C++ mirror
// (include/jrlqp/GoldfarbIdnaniSolver.hpp): ONE solver pre-sized as (5, 8, true) solves five problems of different shapes
// -- no constraints, equalities, inequalities, both, and bounds -- and, like the reference under
// set_is_malloc_allowed(false), must not allocate while doing so.  Here "allocate" is every HIP runtime allocation entry
// point (hipMalloc, hipHostMalloc, hipMallocAsync, hipFreeAsync, hipFree, stream / event creation), counted by interposing them: the
// definitions below are exported from this executable (-rdynamic), so libjrlqp_hip.so's calls land here as well.
//
//   usage: test_multiple_uses <problems.bin> <results.bin>
//   problems.bin: int32 count, then per problem int32 n, m, nb and doubles G[n*n] a[n] C[n*m] bl[m] bu[m] xl[nb] xu[nb]
//   results.bin : per problem int32 term, iters, doubles f, x[n], u[m+nb], G-after[n*n], int32 status[m+nb]
#include <jrlqp/GoldfarbIdnaniSolver.hpp>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

static int g_alloc_calls = 0;
static bool g_counting = false;

template<class F>
static F next_symbol(const char * name)
{
  void * p = dlsym(RTLD_NEXT, name);
  if(!p)
  {
    std::fprintf(stderr, "no next %s\n", name);
    std::abort();
  }
  return reinterpret_cast<F>(p);
}

extern "C" {
hipError_t hipMalloc(void ** p, size_t s)
{
  static auto real = next_symbol<hipError_t (*)(void **, size_t)>("hipMalloc");
  if(g_counting) ++g_alloc_calls;
  return real(p, s);
}
hipError_t hipFree(void * p)
{
  static auto real = next_symbol<hipError_t (*)(void *)>("hipFree");
  if(g_counting) ++g_alloc_calls;
  return real(p);
}
hipError_t hipHostMalloc(void ** p, size_t s, unsigned int flags)
{
  static auto real = next_symbol<hipError_t (*)(void **, size_t, unsigned int)>("hipHostMalloc");
  if(g_counting) ++g_alloc_calls;
  return real(p, s, flags);
}
hipError_t hipHostFree(void * p)
{
  static auto real = next_symbol<hipError_t (*)(void *)>("hipHostFree");
  if(g_counting) ++g_alloc_calls;
  return real(p);
}
hipError_t hipMallocAsync(void ** p, size_t s, hipStream_t st)
{
  static auto real = next_symbol<hipError_t (*)(void **, size_t, hipStream_t)>("hipMallocAsync");
  if(g_counting) ++g_alloc_calls;
  return real(p, s, st);
}
hipError_t hipFreeAsync(void * p, hipStream_t st)
{
  static auto real = next_symbol<hipError_t (*)(void *, hipStream_t)>("hipFreeAsync");
  if(g_counting) ++g_alloc_calls;
  return real(p, st);
}
hipError_t hipStreamCreateWithFlags(hipStream_t * st, unsigned int flags)
{
  static auto real = next_symbol<hipError_t (*)(hipStream_t *, unsigned int)>("hipStreamCreateWithFlags");
  if(g_counting) ++g_alloc_calls;
  return real(st, flags);
}
hipError_t hipEventCreateWithFlags(hipEvent_t * ev, unsigned int flags)
{
  static auto real = next_symbol<hipError_t (*)(hipEvent_t *, unsigned int)>("hipEventCreateWithFlags");
  if(g_counting) ++g_alloc_calls;
  return real(ev, flags);
}
}

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

struct Pb
{
  int n, m, nb;
  std::vector<double> G, a, C, bl, bu, xl, xu;
};

static bool rd(std::FILE * f, void * p, size_t bytes)
{
  return bytes == 0 || std::fread(p, 1, bytes, f) == bytes;
}

int main(int argc, char ** argv)
{
  CHECK(argc == 3 || argc == 4);
  const bool sizes_mode = (argc == 4); // "sizes": one default-constructed solver over problems of any size, results only
  std::FILE * in = std::fopen(argv[1], "rb");
  CHECK(in);
  int count = 0;
  CHECK(rd(in, &count, 4) && count > 0);
  std::vector<Pb> pbs(count);
  for(Pb & p : pbs)
  {
    int hdr[3];
    CHECK(rd(in, hdr, 12));
    p.n = hdr[0]; p.m = hdr[1]; p.nb = hdr[2];
    p.G.resize(size_t(p.n) * p.n); p.a.resize(p.n); p.C.resize(size_t(p.n) * p.m); p.bl.resize(p.m); p.bu.resize(p.m);
    p.xl.resize(p.nb); p.xu.resize(p.nb);
    CHECK(rd(in, p.G.data(), 8 * p.G.size()) && rd(in, p.a.data(), 8 * p.a.size()) && rd(in, p.C.data(), 8 * p.C.size())
          && rd(in, p.bl.data(), 8 * p.bl.size()) && rd(in, p.bu.data(), 8 * p.bu.size()) && rd(in, p.xl.data(), 8 * p.xl.size())
          && rd(in, p.xu.data(), 8 * p.xu.size()));
  }
  std::fclose(in);
  std::FILE * out = std::fopen(argv[2], "wb");
  CHECK(out);

  if(sizes_mode)
  {
    // the back ends behind the one class: n <= 64 (batched kernel, batch of one), 64 < n <= 128 (large-n batched kernel; beyond
    // n = 104 with its R factor in the workspace resize() allocated), beyond (Mode A); the solver grows as the problems do and
    // then serves the smaller ones again -- in the second pass, with every capacity reached, WITHOUT a single allocation call
    // (n = 120 included: no hipMallocAsync / hipFreeAsync around the launch)
    GoldfarbIdnaniSolver any;
    for(int pass = 0; pass < 2; ++pass)
      for(Pb p : pbs)
      {
        g_alloc_calls = 0;
        g_counting = (pass == 1);
        auto st = any.solve({p.G.data(), p.n, p.n, p.n}, {p.a.data(), p.n}, {p.C.data(), p.n, p.m, p.n}, {p.bl.data(), p.m}, {p.bu.data(), p.m},
                            {p.xl.data(), p.nb}, {p.xu.data(), p.nb});
        g_counting = false;
        if(g_alloc_calls != 0) std::printf("solve() made %d allocation-type HIP calls (n=%d m=%d nb=%d)\n", g_alloc_calls, p.n, p.m, p.nb);
        CHECK(g_alloc_calls == 0);
        CHECK(any.lastError().empty());
        if(pass == 1)
        {
          const int term = int(st), it = any.iterations();
          const double f = any.objectiveValue();
          std::fwrite(&term, 4, 1, out);
          std::fwrite(&it, 4, 1, out);
          std::fwrite(&f, 8, 1, out);
          std::fwrite(any.solution().data(), 8, p.n, out);
          std::fwrite(any.multipliers().data(), 8, p.m + p.nb, out);
          std::fwrite(p.G.data(), 8, size_t(p.n) * p.n, out);
          for(auto s : any.activeSet())
          {
            const int v = int(s);
            std::fwrite(&v, 4, 1, out);
          }
        }
      }
    std::fclose(out);
    std::printf("sizes OK\n");
    return 0;
  }
  // two pre-sized solvers: the reference's (5, 8, true), which takes the one-launch batched path at these sizes, and one
  // forced through Mode A by a padded leading dimension (solveDevice with ldg > n), so that both back ends are held to
  // the no-allocation contract
  GoldfarbIdnaniSolver solver(5, 8, true);
  CHECK(solver.lastError().empty());
  // warm-up solve outside the counted region: the HIP runtime loads code objects / sizes its own pools on first use
  {
    Pb p = pbs[count - 1];
    auto st = solver.solve({p.G.data(), p.n, p.n, p.n}, {p.a.data(), p.n}, {p.C.data(), p.n, p.m, p.n}, {p.bl.data(), p.m}, {p.bu.data(), p.m},
                           {p.xl.data(), p.nb}, {p.xu.data(), p.nb});
    CHECK(st == TerminationStatus::SUCCESS);
  }
  for(int pass = 0; pass < 2; ++pass) // the five shapes, twice over: shrinking and growing again inside the capacity
    for(Pb p : pbs)
    {
      CHECK(p.n <= 5 && p.m <= 8);
      g_alloc_calls = 0;
      g_counting = true;
      auto st = solver.solve({p.G.data(), p.n, p.n, p.n}, {p.a.data(), p.n}, {p.C.data(), p.n, p.m, p.n}, {p.bl.data(), p.m},
                             {p.bu.data(), p.m}, {p.xl.data(), p.nb}, {p.xu.data(), p.nb});
      g_counting = false;
      if(g_alloc_calls != 0) std::printf("solve() made %d allocation-type HIP calls (n=%d m=%d nb=%d)\n", g_alloc_calls, p.n, p.m, p.nb);
      CHECK(g_alloc_calls == 0);
      CHECK(st == TerminationStatus::SUCCESS);
      CHECK(int(solver.solution().size()) == p.n && int(solver.multipliers().size()) == p.m + p.nb);
      if(pass == 0)
      {
        const int term = int(st), it = solver.iterations();
        const double f = solver.objectiveValue();
        std::fwrite(&term, 4, 1, out);
        std::fwrite(&it, 4, 1, out);
        std::fwrite(&f, 8, 1, out);
        std::fwrite(solver.solution().data(), 8, p.n, out);
        std::fwrite(solver.multipliers().data(), 8, p.m + p.nb, out);
        std::fwrite(p.G.data(), 8, size_t(p.n) * p.n, out); // G after the solve: lower triangle = Cholesky factor
        for(auto s : solver.activeSet())
        {
          const int v = int(s);
          std::fwrite(&v, 4, 1, out);
        }
      }
    }
  std::fclose(out);

  // Mode A under the same contract: device-resident problem with a padded leading dimension (ldg = n + 3)
  {
    const Pb & p = pbs[count - 1];
    const int ld = p.n + 3;
    GoldfarbIdnaniSolver big(5, 8, true);
    double *dG, *da, *dC, *dbl, *dbu, *dxl, *dxu;
    CHECK(hipMalloc((void **)&dG, 8 * size_t(ld) * p.n) == hipSuccess && hipMalloc((void **)&da, 8 * 8) == hipSuccess
          && hipMalloc((void **)&dC, 8 * size_t(ld) * 8) == hipSuccess && hipMalloc((void **)&dbl, 8 * 8) == hipSuccess
          && hipMalloc((void **)&dbu, 8 * 8) == hipSuccess && hipMalloc((void **)&dxl, 8 * 8) == hipSuccess
          && hipMalloc((void **)&dxu, 8 * 8) == hipSuccess);
    auto upload = [&]() {
      bool ok = hipMemcpy2D(dG, 8 * size_t(ld), p.G.data(), 8 * size_t(p.n), 8 * size_t(p.n), p.n, hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMemcpy(da, p.a.data(), 8 * p.n, hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMemcpy2D(dC, 8 * size_t(ld), p.C.data(), 8 * size_t(p.n), 8 * size_t(p.n), p.m, hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMemcpy(dbl, p.bl.data(), 8 * p.m, hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMemcpy(dbu, p.bu.data(), 8 * p.m, hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMemcpy(dxl, p.xl.data(), 8 * p.nb, hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMemcpy(dxu, p.xu.data(), 8 * p.nb, hipMemcpyHostToDevice) == hipSuccess;
      return ok;
    };
    big.resize(p.n, p.m, p.nb > 0);
    CHECK(upload());
    CHECK(big.solveDevice(dG, ld, da, dC, ld, dbl, dbu, dxl, dxu) == TerminationStatus::SUCCESS); // warm-up
    CHECK(upload());
    g_alloc_calls = 0;
    g_counting = true;
    big.resize(p.n, p.m, p.nb > 0);
    auto st = big.solveDevice(dG, ld, da, dC, ld, dbl, dbu, dxl, dxu);
    g_counting = false;
    CHECK(g_alloc_calls == 0);
    CHECK(st == TerminationStatus::SUCCESS);
    // same answer as the batched path on the same problem, bit for bit
    Pb q = p;
    CHECK(solver.solve({q.G.data(), q.n, q.n, q.n}, {q.a.data(), q.n}, {q.C.data(), q.n, q.m, q.n}, {q.bl.data(), q.m}, {q.bu.data(), q.m},
                       {q.xl.data(), q.nb}, {q.xu.data(), q.nb})
          == TerminationStatus::SUCCESS);
    CHECK(big.solution() == solver.solution() && big.multipliers() == solver.multipliers() && big.iterations() == solver.iterations());
    CHECK(big.objectiveValue() == solver.objectiveValue() && big.activeSet() == solver.activeSet());
  }

  // errors come back as statuses, never as exceptions: inconsistent sizes, then a solver that keeps working afterwards
  {
    Pb p = pbs[0];
    auto st = solver.solve({p.G.data(), p.n, p.n, p.n}, {p.a.data(), p.n - 1}, {p.C.data(), p.n, p.m, p.n}, {p.bl.data(), p.m}, {p.bu.data(), p.m},
                           {p.xl.data(), p.nb}, {p.xu.data(), p.nb});
    CHECK(st == TerminationStatus::INCONSISTENT_INPUT);
    st = solver.solve({p.G.data(), p.n, p.n, p.n}, {p.a.data(), p.n}, {p.C.data(), p.n, p.m, p.n}, {p.bl.data(), p.m}, {p.bu.data(), p.m},
                      {p.xl.data(), p.nb}, {p.xu.data(), p.nb});
    CHECK(st == TerminationStatus::SUCCESS);
  }
  std::printf("multiple uses OK\n");
  return 0;
}
