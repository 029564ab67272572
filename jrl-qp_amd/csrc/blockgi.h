// blockgi.h -- launch interface of the structured GI kernel (internal to libjrlqp_hip.so)
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/jrlqp_hip.h"

namespace jrlqp
{
// One element of the orthonormal sequence (OrthonormalSequence.h:88-93 EmbeddedSeq): a Householder reflector
// (kind 0, size 1, rows [start, n)) or a Givens sequence (kind 1, `size` rotations on rows start .. start + size)
struct SeqRec
{
  int kind, start, size, pad;
  double tau, pad2;
};

// One batch of identically shaped structured QPs.  Every pointer is a DEVICE pointer.
struct BgiArgs
{
  // StructuredG (shared shape; per-problem block pointers)
  int kind;            // jrlqp_blocks numbering: 0 block arrow down, 1 block arrow up, 2 block tridiagonal
  int nblk, n;
  int maxb;            // largest block (rows): beyond 64 the block solves carry two rows per lane
  const int * bs;      // [nblk] block sizes
  const int * boff;    // [nblk] first row of block b in the vector of unknowns
  const int * ldd;     // [nblk]
  const int * lds;     // [nblk-1]
  double * const * diag; // [nprob*nblk]
  double * const * side; // [nprob*(nblk-1)]
  const int * fail;    // [nprob] set by the factorisation kernels
  // StructuredC (block diagonal)
  int m, nb, ncb;
  const int *cvar, *ccstr; // [ncb] rows / columns of block i
  const int *cum_var, *cum_cstr; // [ncb] prefix sums (StructuredC.cpp:13-21)
  const int * to_block;    // [m] constraint -> block (toBlock_)
  const int * cld;         // [ncb]
  const double * const * cblk; // [nprob*ncb]
  // problem vectors (problem b at base + b * length)
  const double *a, *bl, *bu, *xl, *xu;
  int max_iter;
  double big_bnd;
  // workspace, per problem
  int n_pad;   // n rounded up to a multiple of 4
  int qcap;    // capacity of the active list: min(n, m + nb) + 1, rounded up to a multiple of 4
  int row_len; // doubles per sequence element: max(n_pad, 2 * qcap)
  double * pool;  // [nprob * max_iter * row_len]   Householder vectors (absolute row index) / Givens (c | s)
  double * Rp;    // [nprob * rlen]   R packed by columns: R(i,j) at i + j(j+1)/2
  long rlen;
  SeqRec * desc;  // [nprob * max_iter]  one record per element of the orthonormal sequence
  int * active;   // [nprob * qcap]
  double * xu_ws; // [nprob * (n_pad + qcap)]  x and u of the throughput layout (see blockgi_lds_bytes)
  int xu_global;  // 1: x and u live in xu_ws (HBM/L2) instead of LDS, so that two large problems share a CU
  // outputs
  double *x, *u;
  int *status, *term, *iters;
  double * f;
  jrlqp_trace_rec * trace;
  int trace_cap;
  // ---- transposed copies of the factor's blocks (both forms; bgi_pack_kernel, behind the factorisation): a backward substitution and
  // the couplings that walk a block by rows read them with consecutive addresses across the lanes
  double * Tw;          // [nprob * tw_len]  per problem: diagT_b (bs_b x bs_b, ld bs_b) for every block, then sideT_i (cols_i x rows_i, ld cols_i)
  const int * tw_doff;  // [nblk]    offset of diagT_b
  const int * tw_soff;  // [nblk-1]  offset of sideT_i
  const int * tw_sc;    // [nblk-1]  columns of side block i (= leading dimension of its transposed copy)
  long tw_len;
  // ---- contract C9 (blockgi_c9.hip): no orthonormal sequence; workspace per problem ----
  int form;        // 0: the reference's orthonormal sequence (blockgi.hip), 1: contract C9
  int ngrp;        // groups of constraints: ncb blocks of C, then (bounds) one per Hessian block in block order
  const int * greach;   // [2 ngrp] first / last off-head block (block-order index) under the group's window (first > last: none)
  const int * tk;       // [4 nblk] per block (block order): range of general groups and range of bound groups that reach it
  const int * kap_prow; // [n] block-order index of the block that holds PERMUTED row i
  const int * kap_row;  // [n] the same for ORIGINAL row e
  const int * off_perm; // [nblk] first permuted row of block-order block k
  const int * kap_phys; // [nblk] physical block index of block-order block k
  const int * goff;     // [ngrp] first entry of group g in the per-problem active lists
  double * Bc;     // [nprob * qcap * n_pad]  columns L^-1 n_k of the active constraints, by SLOT, permuted coordinates
  double * Sm;     // [nprob * qcap * qcap]   S = R^-1: S(row of slot s, column of position j) at s + j qcap
  double * St;     // [nprob * qcap * qcap]   the same transposed: at s qcap + j
  int * sgrp;      // [nprob * qcap]  slot -> group
  int * freel;     // [nprob * qcap]  stack of free slots
  int tune;        // diagnostic only (JRLQP_BGI_TUNE): bits that switch parts of the kernels off for timing experiments; 0 in production
  long long * prof; // diagnostic only (jrlqp_debug_set_profile): prof_stride >= 32 int64 per problem (16 used), 100 MHz ticks per phase; or nullptr
  int prof_stride;
};

constexpr int kBgiMaxN = 4096; // unknowns per structured QP (vectors live in LDS; four chunks of 1024 chains: 3072 until round 4)

// xu_global = 1 (throughput layout for large problems in large batches): x and u are kept in the HBM workspace, which
// brings a 3000-unknown problem under half a CU's LDS
long blockgi_lds_bytes(int n, int m, int nb, int nblk, int xu_global = 0);
hipError_t launch_blockgi(const BgiArgs & a, int nprob, hipStream_t stream);
// contract C9 (blockgi_c9.hip)
long blockgi_c9_lds_bytes(int n, int m, int nb, int nblk, int ncb, int xu_global = 0);
hipError_t launch_blockgi_c9(const BgiArgs & a, int nprob, hipStream_t stream);
// transposed copies of the (factorised) blocks into BgiArgs::Tw: enqueue behind the factorisation, before either kernel
hipError_t launch_blockgi_pack(const BgiArgs & a, int nprob, hipStream_t stream);
} // namespace jrlqp
