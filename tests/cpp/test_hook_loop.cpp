// The hook level of the C ABI driven the way a jrl-qp maintainer would drive it: a restatement of
// DualSolver::solve() (src/DualSolver.cpp:91-168) with its init() (:200-210 -> GoldfarbIdnaniSolver::init_, src/GoldfarbIdnaniSolver.cpp:
// 56-82) that calls ONLY
//     jrlqp_dense_bind / init / init_active_set / select / step / advance / add / remove / results
// -- never jrlqp_dense_solve -- and keeps its own ActiveSet, iteration counter and objective value, exactly as DualSolver does
// (A_, it_, f_).  For every problem of the input file the loop's result is compared bit for bit with jrlqp_dense_solve on a second
// handle, and written out so that tests/test_gpu_hooks.py can compare it with the CPU oracle.
//
//   test_hook_loop problems.bin results.bin
// problems.bin: int32 count; per problem: int32 n, m, nb, max_iter; then G (n*n), a, C (n*m col-major), bl, bu, xl, xu as float64.
// results.bin: per problem int32 term, iters; float64 f, x[n], u[m+nb]; int32 status[m+nb]; int32 nactive, active[nactive].
#include <hip/hip_runtime.h>
#include <jrlqp_hip.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(c)                                                        \
  do                                                                    \
  {                                                                     \
    if(!(c))                                                            \
    {                                                                   \
      std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);  \
      std::exit(1);                                                     \
    }                                                                   \
  } while(0)
#define OK(call) CHECK((call) == JRLQP_OK)
#define HIP(call) CHECK((call) == hipSuccess)

namespace
{
enum { INACTIVE = 0, LOWER, UPPER, EQUALITY, LOWER_BOUND, UPPER_BOUND, FIXED }; // enums.h:9-23

struct Problem
{
  int n, m, nb, max_iter;
  std::vector<double> G, a, C, bl, bu, xl, xu;
};

struct Device
{
  double *G = nullptr, *a = nullptr, *C = nullptr, *bl = nullptr, *bu = nullptr, *xl = nullptr, *xu = nullptr;
  void upload(const Problem & p)
  {
    auto up = [](double *& d, const std::vector<double> & h) {
      HIP(hipMalloc(reinterpret_cast<void **>(&d), sizeof(double) * (h.size() ? h.size() : 1)));
      if(!h.empty()) HIP(hipMemcpy(d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    };
    up(G, p.G); up(a, p.a); up(C, p.C); up(bl, p.bl); up(bu, p.bu); up(xl, p.xl); up(xu, p.xu);
  }
  void release()
  {
    for(double * d : {G, a, C, bl, bu, xl, xu}) (void)hipFree(d);
  }
};

struct Result
{
  int term = JRLQP_UNKNOWN, iters = 0;
  double f = 0.0;
  std::vector<double> x, u;
  std::vector<int> status, active;
};

// DualSolver::solve() over the hooks.  `A`, `it`, `f` are the caller's (DualSolver::A_, it_, f_).
Result hook_loop(jrlqp_dense * h, const Problem & pb, const Device & d, void * stream)
{
  const int n = pb.n, m = pb.m, nb = pb.nb;
  const double big = 1e100; // SolverOptions::bigBnd_
  Result R;
  R.x.assign(n, 0.0);
  R.u.assign(m + nb, 0.0);
  R.status.assign(m + nb, INACTIVE);
  std::vector<int> Astatus(m + nb, INACTIVE), Aactive; // internal::ActiveSet: status per constraint + ordered list
  int it = 0;                                           // DualSolver::it_
  double f = 0.0;                                       // DualSolver::f_
  auto finish = [&](int term) {
    R.term = term;
    R.iters = it;
    R.f = f;
    // solution(), multipliers(), activeSet() (DualSolver.cpp:33-84): the handle expands the condensed multipliers with its mirror
    OK(jrlqp_dense_results(h, R.x.data(), R.u.data(), R.status.data(), nullptr, nullptr, stream));
    R.active = Aactive;
    CHECK(R.status == Astatus); // the handle's mirror and the caller's ActiveSet agree
    return R;
  };
  auto bound_value = [&](int p, int status) {
    switch(status)
    {
      case LOWER: case EQUALITY: return pb.bl[p];
      case UPPER: return pb.bu[p];
      case LOWER_BOUND: case FIXED: return pb.xl[p - m];
      default: return pb.xu[p - m];
    }
  };

  OK(jrlqp_dense_bind(h, d.G, n, d.a, m ? d.C : nullptr, n, m ? d.bl : nullptr, m ? d.bu : nullptr, nb ? d.xl : nullptr, nb ? d.xu : nullptr));
  // ---- init() (DualSolver.cpp:200-210) -> init_ (GoldfarbIdnaniSolver.cpp:56-82)
  int info = -1;
  OK(jrlqp_dense_init(h, &info, &f, stream));
  if(info >= 0) return finish(JRLQP_NON_POS_HESSIAN);
  {
    int term = JRLQP_SUCCESS, q0 = 0;
    std::vector<int> act(n + 1), st(n + 1);
    OK(jrlqp_dense_init_active_set(h, big, &term, &q0, act.data(), st.data(), &f, stream)); // initActiveSet (:268-287)
    if(term != JRLQP_SUCCESS) return finish(term);
    for(int k = 0; k < q0; ++k)
    {
      Aactive.push_back(act[k]);
      Astatus[act[k]] = st[k];
    }
  }
  // ---- the loop (DualSolver.cpp:96-167)
  bool skipStep1 = false;
  int p = -1, pstat = INACTIVE;
  for(; it < pb.max_iter; ++it)
  {
    const int q = (int)Aactive.size();
    if(!skipStep1)
    {
      OK(jrlqp_dense_select(h, &p, &pstat, stream)); // Step 1 (+ u[q] = 0)
      if(pstat == INACTIVE) return finish(JRLQP_SUCCESS);
    }
    double t1, zz, nz, nx, uq;
    int l;
    OK(jrlqp_dense_step(h, p, pstat, q, big, &t1, &l, &zz, &nz, &nx, &uq, stream)); // Step 2: computeStep + the reductions of computeStepLength
    double t2 = big;
    if(std::sqrt(zz) > 1e-14) t2 = (bound_value(p, pstat) - nx) / nz; // GoldfarbIdnaniSolver.cpp:174-215
    const double t = (t2 < t1) ? t2 : t1;                            // std::min(t1, t2)
    if(t >= big) return finish(JRLQP_INFEASIBLE);
    if(t2 >= big)
    {
      OK(jrlqp_dense_advance(h, t, q, 0, stream)); // u += t [-r; 1]
      OK(jrlqp_dense_remove(h, l, q, Aactive[l], stream));
      Astatus[Aactive[l]] = INACTIVE;
      Aactive.erase(Aactive.begin() + l);
      skipStep1 = true;
    }
    else
    {
      OK(jrlqp_dense_advance(h, t, q, 1, stream)); // x += t z; u += t [-r; 1]
      const double dot = (pstat == UPPER || pstat == UPPER_BOUND) ? -nz : nz; // dot_ (ConstraintNormal::dot)
      f = std::fma(t * dot, 0.5 * t + uq, f);                                 // DualSolver.cpp:147
      if(t == t2)
      {
        OK(jrlqp_dense_add(h, p, pstat, q, stream));
        Aactive.push_back(p);
        Astatus[p] = pstat;
        skipStep1 = false;
      }
      else
      {
        OK(jrlqp_dense_remove(h, l, q, Aactive[l], stream));
        Astatus[Aactive[l]] = INACTIVE;
        Aactive.erase(Aactive.begin() + l);
        skipStep1 = true;
      }
    }
  }
  return finish(JRLQP_MAX_ITER_REACHED);
}

Result library_loop(jrlqp_dense * h, const Problem & pb, const Device & d, void * stream)
{
  const int n = pb.n, m = pb.m, nb = pb.nb;
  Result R;
  R.x.assign(n, 0.0);
  R.u.assign(m + nb, 0.0);
  R.status.assign(m + nb, INACTIVE);
  jrlqp_options o;
  jrlqp_default_options(&o);
  o.max_iter = pb.max_iter;
  OK(jrlqp_dense_solve(h, d.G, n, d.a, m ? d.C : nullptr, n, m ? d.bl : nullptr, m ? d.bu : nullptr, nb ? d.xl : nullptr, nb ? d.xu : nullptr, &o,
                       &R.term, stream));
  OK(jrlqp_dense_results(h, R.x.data(), R.u.data(), R.status.data(), &R.iters, &R.f, stream));
  return R;
}

template<class T>
void rd(std::FILE * f, T * p, size_t cnt)
{
  CHECK(std::fread(p, sizeof(T), cnt, f) == cnt);
}
} // namespace

int main(int argc, char ** argv)
{
  CHECK(argc == 3);
  std::FILE * fi = std::fopen(argv[1], "rb");
  std::FILE * fo = std::fopen(argv[2], "wb");
  CHECK(fi && fo);
  int count = 0;
  rd(fi, &count, 1);
  int nmax = 1, mmax = 0;
  std::vector<Problem> pbs(count);
  for(Problem & p : pbs)
  {
    int hdr[4];
    rd(fi, hdr, 4);
    p.n = hdr[0]; p.m = hdr[1]; p.nb = hdr[2]; p.max_iter = hdr[3];
    p.G.resize((size_t)p.n * p.n); p.a.resize(p.n); p.C.resize((size_t)p.n * p.m); p.bl.resize(p.m); p.bu.resize(p.m);
    p.xl.resize(p.nb); p.xu.resize(p.nb);
    rd(fi, p.G.data(), p.G.size()); rd(fi, p.a.data(), p.a.size()); rd(fi, p.C.data(), p.C.size()); rd(fi, p.bl.data(), p.bl.size());
    rd(fi, p.bu.data(), p.bu.size()); rd(fi, p.xl.data(), p.xl.size()); rd(fi, p.xu.data(), p.xu.size());
    nmax = p.n > nmax ? p.n : nmax;
    mmax = p.m > mmax ? p.m : mmax;
  }
  // two pre-sized handles (GoldfarbIdnaniSolver(nbVar, nbCstr, useBounds)), resized per problem like DualSolver::resize
  jrlqp_dense *hook = nullptr, *lib = nullptr;
  OK(jrlqp_dense_create(nmax, mmax, nmax, &hook));
  OK(jrlqp_dense_create(nmax, mmax, nmax, &lib));
  hipStream_t st;
  HIP(hipStreamCreate(&st));
  int k = 0;
  for(const Problem & p : pbs)
  {
    OK(jrlqp_dense_resize(hook, p.n, p.m, p.nb));
    OK(jrlqp_dense_resize(lib, p.n, p.m, p.nb));
    Device da, db;
    da.upload(p);
    db.upload(p);
    const Result A = hook_loop(hook, p, da, st);
    const Result B = library_loop(lib, p, db, st);
    // the caller's loop over the hooks and the library's own loop: the same bits
    CHECK(A.term == B.term && A.iters == B.iters);
    CHECK(std::memcmp(&A.f, &B.f, sizeof(double)) == 0);
    CHECK(std::memcmp(A.x.data(), B.x.data(), sizeof(double) * p.n) == 0);
    CHECK(p.m + p.nb == 0 || std::memcmp(A.u.data(), B.u.data(), sizeof(double) * (p.m + p.nb)) == 0);
    CHECK(A.status == B.status);
    // ... and the factor left in G
    std::vector<double> Ga(p.G.size()), Gb(p.G.size());
    HIP(hipMemcpy(Ga.data(), da.G, sizeof(double) * Ga.size(), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(Gb.data(), db.G, sizeof(double) * Gb.size(), hipMemcpyDeviceToHost));
    CHECK(std::memcmp(Ga.data(), Gb.data(), sizeof(double) * Ga.size()) == 0);
    const int hdr[2] = {A.term, A.iters};
    std::fwrite(hdr, sizeof(int), 2, fo);
    std::fwrite(&A.f, sizeof(double), 1, fo);
    std::fwrite(A.x.data(), sizeof(double), p.n, fo);
    std::fwrite(A.u.data(), sizeof(double), p.m + p.nb, fo);
    std::fwrite(A.status.data(), sizeof(int), p.m + p.nb, fo);
    const int na = (int)A.active.size();
    std::fwrite(&na, sizeof(int), 1, fo);
    std::fwrite(A.active.data(), sizeof(int), na, fo);
    da.release();
    db.release();
    std::printf("problem %d: n=%d m=%d nb=%d max_iter=%d -> term %d, %d iterations, %d active\n", k++, p.n, p.m, p.nb, p.max_iter, A.term, A.iters, na);
  }
  // argument checks of the hook level: a q or an index that disagrees with the handle's mirror launches nothing
  {
    const Problem & p = pbs[0];
    Device d;
    d.upload(p);
    OK(jrlqp_dense_resize(hook, p.n, p.m, p.nb));
    OK(jrlqp_dense_bind(hook, d.G, p.n, d.a, p.m ? d.C : nullptr, p.n, p.m ? d.bl : nullptr, p.m ? d.bu : nullptr, p.nb ? d.xl : nullptr,
                        p.nb ? d.xu : nullptr));
    int info = 0;
    OK(jrlqp_dense_init(hook, &info, nullptr, st));
    CHECK(jrlqp_dense_add(hook, 0, LOWER, 1, st) == JRLQP_ERR_INVALID_ARGUMENT);    // q != 0
    CHECK(jrlqp_dense_remove(hook, 0, 1, 0, st) == JRLQP_ERR_INVALID_ARGUMENT);     // nothing is active
    CHECK(jrlqp_dense_add(hook, p.m + p.nb, LOWER, 0, st) == JRLQP_ERR_INVALID_ARGUMENT); // no such constraint
    d.release();
  }
  jrlqp_dense_destroy(hook);
  jrlqp_dense_destroy(lib);
  HIP(hipStreamDestroy(st));
  std::fclose(fi);
  std::fclose(fo);
  std::printf("hook loop OK\n");
  return 0;
}
