// jrlqp/BlockGISolver.hpp -- host C++17 mirror of the reference's STRUCTURED solver interface over the C ABI.
//
// Same public surface and semantics as jrl::qp::experimental::BlockGISolver with structured::StructuredG and
// structured::StructuredC (include/jrl-qp/experimental/BlockGISolver.h:24-40, include/jrl-qp/structured/StructuredG.h:17-76,
// include/jrl-qp/structured/StructuredC.h:17-54), with raw column-major views where the reference takes Eigen::Ref:
//
//   std::vector<jrlqp::MatrixRef> D = {...}, S = {...};                       // views into the caller's dense G
//   jrlqp::StructuredG G(jrlqp::StructuredG::Type::TriBlockDiagonal, D, S);   // diag blocks, off-diagonal blocks
//   jrlqp::StructuredC C(Cblocks);                                            // block-diagonal C, one block per entry
//   jrlqp::BlockGISolver solver(nbVar, nbCstr, useBounds);
//   auto st = solver.solve(G, a, C, bl, bu, xl, xu);                          // HOST views; G's blocks receive the factor
//   solver.solution(), solver.multipliers(), solver.objectiveValue(), solver.iterations(), solver.activeSet()
//
// The numerics run on the GPU (libjrlqp_hip.so: jrlqp_blockgi_*, one workgroup per problem, the whole active-set loop on
// the device).  solve() stages the blocks on the device and copies the factor back into the caller's blocks (the
// reference factorises them in place, StructuredG::lltInPlace).  As in the reference only the cold start without data
// equalities is functional (BlockGISolver.cpp:475): other inputs return INCONSISTENT_INPUT.
// Header-only; link with -ljrlqp_hip and the HIP runtime.
#pragma once

#include <numeric>

#include "GoldfarbIdnaniSolver.hpp"

namespace jrlqp
{
/** structured::StructuredG (StructuredG.h:17-76): a symmetric matrix given by its diagonal blocks and one list of
 *  off-diagonal blocks whose meaning depends on the type (sub-diagonal blocks, first block column, last block row). */
class StructuredG
{
public:
  enum class Type
  {
    TriBlockDiagonal,
    BlockArrowUp,
    BlockArrowDown
  };
  StructuredG() = default;
  StructuredG(Type t, const std::vector<MatrixRef> & diag, const std::vector<MatrixRef> & offDiag) : type_(t), diag_(diag), offDiag_(offDiag)
  {
    nbVar_ = 0;
    for(const auto & D : diag_) nbVar_ += D.cols;
  }
  Type type() const { return type_; }
  int nbVar() const { return nbVar_; }
  int nbBlocks() const { return int(diag_.size()); }
  const std::vector<MatrixRef> & diag() const { return diag_; }
  const std::vector<MatrixRef> & offDiag() const { return offDiag_; }

private:
  Type type_ = Type::TriBlockDiagonal;
  int nbVar_ = 0;
  std::vector<MatrixRef> diag_, offDiag_;
};

/** structured::StructuredC (StructuredC.h:17-54): block-diagonal constraint matrix, block i is n_i x m_i with one
 *  constraint per column; the n_i add up to nbVar (the partition need not be G's). */
class StructuredC
{
public:
  StructuredC() = default;
  explicit StructuredC(const std::vector<MatrixConstRef> & C) : diag_(C)
  {
    for(const auto & Ci : diag_)
    {
      nbVar_ += Ci.rows;
      nbCstr_ += Ci.cols;
    }
  }
  int nbVar() const { return nbVar_; }
  int nbCstr() const { return nbCstr_; }
  const std::vector<MatrixConstRef> & diag() const { return diag_; }

private:
  int nbVar_ = 0, nbCstr_ = 0;
  std::vector<MatrixConstRef> diag_;
};

class BlockGISolver
{
public:
  BlockGISolver() = default;
  BlockGISolver(int nbVar, int nbCstr, bool useBounds) : n_(nbVar), m_(nbCstr), nb_(useBounds ? nbVar : 0) {}
  BlockGISolver(const BlockGISolver &) = delete;
  BlockGISolver & operator=(const BlockGISolver &) = delete;
  ~BlockGISolver() { release(); }

  void options(const SolverOptions & o) { options_ = o; }
  const SolverOptions & options() const { return options_; }

  /** min 1/2 x^T G x + a^T x  s.t.  bl <= C^T x <= bu, xl <= x <= xu  (BlockGISolver.h:33-40).  HOST views. */
  TerminationStatus solve(const StructuredG & G, VectorConstRef a, const StructuredC & C, VectorConstRef bl, VectorConstRef bu, VectorConstRef xl,
                          VectorConstRef xu)
  {
    const int n = G.nbVar(), m = C.nbCstr(), nb = xl.size > 0 ? n : 0;
    if(a.size != n || C.nbVar() != n || bl.size != m || bu.size != m || (xl.size != 0 && xl.size != n) || xu.size != xl.size)
      return TerminationStatus::INCONSISTENT_INPUT;
    err_.clear();
    plan(G, C, nb != 0);
    if(!err_.empty()) return TerminationStatus::UNKNOWN;
    // stage the blocks back to back on the device (column-major, tight leading dimensions)
    std::vector<double> hbuf(gTot_ + cTot_);
    size_t o = 0;
    auto pack = [&](const double * p, int rows, int cols, int ld) {
      for(int j = 0; j < cols; ++j)
        for(int i = 0; i < rows; ++i) hbuf[o++] = p[i + size_t(j) * ld];
    };
    for(const auto & D : G.diag()) pack(D.data, D.rows, D.cols, D.ld);
    for(const auto & S : G.offDiag()) pack(S.data, S.rows, S.cols, S.ld);
    for(const auto & Ci : C.diag()) pack(Ci.data, Ci.rows, Ci.cols, Ci.ld);
    hip(hipMemcpy(dblk_, hbuf.data(), sizeof(double) * hbuf.size(), hipMemcpyHostToDevice));
    hip(hipMemcpy(da_, a.data, sizeof(double) * n, hipMemcpyHostToDevice));
    if(m)
    {
      hip(hipMemcpy(dbl_, bl.data, sizeof(double) * m, hipMemcpyHostToDevice));
      hip(hipMemcpy(dbu_, bu.data, sizeof(double) * m, hipMemcpyHostToDevice));
    }
    if(nb)
    {
      hip(hipMemcpy(dxl_, xl.data, sizeof(double) * nb, hipMemcpyHostToDevice));
      hip(hipMemcpy(dxu_, xu.data, sizeof(double) * nb, hipMemcpyHostToDevice));
    }
    jrlqp_options opt{options_.maxIter_, options_.bigBnd_};
    check(jrlqp_blockgi_solve(h_, da_, dbl_, dbu_, nb ? dxl_ : nullptr, nb ? dxu_ : nullptr, &opt, dx_, du_, dst_, dterm_, dit_, df_, nullptr, 0,
                              nullptr),
          "jrlqp_blockgi_solve");
    hip(hipDeviceSynchronize());
    int term = 0;
    hip(hipMemcpy(&term, dterm_, sizeof(int), hipMemcpyDeviceToHost));
    hip(hipMemcpy(&it_, dit_, sizeof(int), hipMemcpyDeviceToHost));
    hip(hipMemcpy(&f_, df_, sizeof(double), hipMemcpyDeviceToHost));
    x_.resize(n);
    u_.assign(m + nb, 0.0);
    ias_.assign(m + nb, 0);
    hip(hipMemcpy(x_.data(), dx_, sizeof(double) * n, hipMemcpyDeviceToHost));
    if(m + nb)
    {
      hip(hipMemcpy(u_.data(), du_, sizeof(double) * (m + nb), hipMemcpyDeviceToHost));
      hip(hipMemcpy(ias_.data(), dst_, sizeof(int) * (m + nb), hipMemcpyDeviceToHost));
    }
    as_.resize(m + nb);
    for(size_t i = 0; i < ias_.size(); ++i) as_[i] = static_cast<ActivationStatus>(ias_[i]);
    // the factor goes back into the caller's blocks (StructuredG::lltInPlace factorises them in place)
    hip(hipMemcpy(hbuf.data(), dblk_, sizeof(double) * gTot_, hipMemcpyDeviceToHost));
    o = 0;
    auto unpack = [&](double * p, int rows, int cols, int ld) {
      for(int j = 0; j < cols; ++j)
        for(int i = 0; i < rows; ++i) p[i + size_t(j) * ld] = hbuf[o++];
    };
    for(const auto & D : G.diag()) unpack(D.data, D.rows, D.cols, D.ld);
    for(const auto & S : G.offDiag()) unpack(S.data, S.rows, S.cols, S.ld);
    if(!err_.empty()) return TerminationStatus::UNKNOWN;
    return static_cast<TerminationStatus>(term);
  }

  const std::vector<double> & solution() const { return x_; }
  const std::vector<double> & multipliers() const { return u_; }
  double objectiveValue() const { return f_; }
  int iterations() const { return it_; }
  const std::vector<ActivationStatus> & activeSet() const { return as_; }
  /** Empty unless an ABI / HIP runtime call failed during the last solve(). */
  const std::string & lastError() const { return err_; }

private:
  // No exception leaves the class (the reference's solvers never throw): the FIRST failing ABI / HIP call is recorded, the
  // calls after it fail harmlessly on their null handles, and solve() reports TerminationStatus::UNKNOWN (see lastError()).
  void check(int rc, const char * what) noexcept
  {
    if(rc == JRLQP_OK || !err_.empty()) return;
    try { err_ = std::string(what) + ": " + jrlqp_status_string(rc); } catch(...) {}
  }
  void hip(hipError_t e) noexcept
  {
    if(e == hipSuccess || !err_.empty()) return;
    try { err_ = std::string("HIP: ") + hipGetErrorString(e); } catch(...) {}
  }
  // (re)build the plan when the shape changes: block sizes, kinds, constraint partition
  void plan(const StructuredG & G, const StructuredC & C, bool useBounds)
  {
    std::vector<int> shape;
    shape.push_back(int(G.type()));
    shape.push_back(useBounds ? 1 : 0);
    shape.push_back(options_.maxIter_);
    for(const auto & D : G.diag()) shape.push_back(D.cols);
    shape.push_back(-1);
    for(const auto & Ci : C.diag())
    {
      shape.push_back(Ci.rows);
      shape.push_back(Ci.cols);
    }
    if(h_ && shape == shape_) return;
    release();
    shape_ = shape;
    const int nblk = G.nbBlocks(), ncb = int(C.diag().size());
    n_ = G.nbVar();
    m_ = C.nbCstr();
    nb_ = useBounds ? n_ : 0;
    std::vector<int> bs, cv, cc;
    for(const auto & D : G.diag()) bs.push_back(D.cols);
    for(const auto & Ci : C.diag())
    {
      cv.push_back(Ci.rows);
      cc.push_back(Ci.cols);
    }
    // ABI numbering: 0 BlockArrowDown, 1 BlockArrowUp, 2 TriBlockDiagonal
    const int kind = G.type() == StructuredG::Type::BlockArrowDown ? 0 : G.type() == StructuredG::Type::BlockArrowUp ? 1 : 2;
    check(jrlqp_blockgi_create(kind, nblk, bs.data(), ncb, cv.data(), cc.data(), useBounds ? 1 : 0, 1, options_.maxIter_, &h_), "jrlqp_blockgi_create");
    gTot_ = cTot_ = 0;
    for(const auto & D : G.diag()) gTot_ += size_t(D.rows) * D.cols;
    for(const auto & S : G.offDiag()) gTot_ += size_t(S.rows) * S.cols;
    for(const auto & Ci : C.diag()) cTot_ += size_t(Ci.rows) * Ci.cols;
    hip(hipMalloc(&dblk_, sizeof(double) * (gTot_ + cTot_ + 1)));
    hip(hipMalloc(&da_, sizeof(double) * (n_ + 1)));
    hip(hipMalloc(&dbl_, sizeof(double) * (m_ + 1)));
    hip(hipMalloc(&dbu_, sizeof(double) * (m_ + 1)));
    hip(hipMalloc(&dxl_, sizeof(double) * (nb_ + 1)));
    hip(hipMalloc(&dxu_, sizeof(double) * (nb_ + 1)));
    hip(hipMalloc(&dx_, sizeof(double) * (n_ + 1)));
    hip(hipMalloc(&du_, sizeof(double) * (m_ + nb_ + 1)));
    hip(hipMalloc(&dst_, sizeof(int) * (m_ + nb_ + 1)));
    hip(hipMalloc(&dterm_, sizeof(int)));
    hip(hipMalloc(&dit_, sizeof(int)));
    hip(hipMalloc(&df_, sizeof(double)));
    // device pointers of the staged blocks
    std::vector<double *> dp, sp;
    std::vector<const double *> cp;
    std::vector<int> ldd, lds, ldc;
    double * p = dblk_;
    for(const auto & D : G.diag())
    {
      dp.push_back(p);
      ldd.push_back(D.rows);
      p += size_t(D.rows) * D.cols;
    }
    for(const auto & S : G.offDiag())
    {
      sp.push_back(p);
      lds.push_back(S.rows);
      p += size_t(S.rows) * S.cols;
    }
    for(const auto & Ci : C.diag())
    {
      cp.push_back(p);
      ldc.push_back(Ci.rows > 0 ? Ci.rows : 1);
      p += size_t(Ci.rows) * Ci.cols;
    }
    if(sp.empty())
    {
      sp.push_back(nullptr);
      lds.push_back(1);
    }
    check(jrlqp_blockgi_bind(h_, dp.data(), ldd.data(), sp.data(), lds.data(), cp.data(), ldc.data(), nullptr), "jrlqp_blockgi_bind");
  }
  void release()
  {
    if(h_) jrlqp_blockgi_destroy(h_);
    h_ = nullptr;
    for(double ** p : {&dblk_, &da_, &dbl_, &dbu_, &dxl_, &dxu_, &dx_, &du_, &df_})
    {
      if(*p) (void)hipFree(*p);
      *p = nullptr;
    }
    for(int ** p : {&dst_, &dterm_, &dit_})
    {
      if(*p) (void)hipFree(*p);
      *p = nullptr;
    }
  }

  SolverOptions options_;
  jrlqp_blockgi * h_ = nullptr;
  std::vector<int> shape_;
  int n_ = 0, m_ = 0, nb_ = 0, it_ = 0;
  double f_ = 0.0;
  size_t gTot_ = 0, cTot_ = 0;
  double *dblk_ = nullptr, *da_ = nullptr, *dbl_ = nullptr, *dbu_ = nullptr, *dxl_ = nullptr, *dxu_ = nullptr, *dx_ = nullptr, *du_ = nullptr,
         *df_ = nullptr;
  int *dst_ = nullptr, *dterm_ = nullptr, *dit_ = nullptr;
  std::vector<double> x_, u_;
  std::vector<ActivationStatus> as_;
  std::vector<int> ias_;
  std::string err_;
};
} // namespace jrlqp
