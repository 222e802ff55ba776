// Multigrid preconditioner for the elasticity block K_el u = G c + f  (the F_m block, simulation_tumor_growth.py:110-113).
//
// What it stands in for: the reference solves the monolithic system with a sparse LU (simulation_tumor_growth.py:126-130,
// DOLFIN default) and names 'amg' as the alternative where LU no longer fits (simulation_tumor_growth_brain_quad.py:116-119).
// Block-Jacobi PCG needs O(1/h) iterations (800 at 1 M nodes, 1 650 at 10 M); one V-cycle of this hierarchy per PCG
// iteration makes the count independent of the mesh width (tools/proto_gmg.py: 47 / 51 / 53 at n = 16 / 24 / 32).
//
// Design (MI355X first): *geometric multigrid on auxiliary Cartesian grids*.
//   level 0   the mesh (any P1 simplex mesh): block SELL-64 operator; the smoother works in symmetrically scaled variables
//             (K~ = S K S, S = diag(1 / sqrt(k_ii))) on a HALF-precision copy of K~, with single-precision cycle vectors whose
//             iterate lives in 16-byte node records (one aligned gather per neighbour), fused into the SpMV (k_mg_fine)
//   level 1   a Cartesian grid of width H ~ 2h laid over the mesh; prolongation = d-linear interpolation from the
//             grid nodes onto the mesh nodes (8 parents per node, weights from coordinates alone -- no graph
//             algorithms, no aggregates, no QR); smoothed like level 0 (scaled variables, half-precision operator)
//   level 2.. 2:1 coarsenings with d-linear interpolation
//   coarse operators = Galerkin products P^T A P, stored as *dense stencils* [(2R+1)^d][d*d][nodes] in single
//   precision: no column indices, every load unit-stride over the grid nodes, the neighbour gather of x contiguous.
//   R = 1 (27-point) when the mesh nodes sit on a lattice that the grid can align with (the BASELINE box meshes),
//   R = 2 (125-point) for general meshes; mesh edges that reach further stay on level 0 (dropped from the products).
//   Rigid-body modes are d-linear, i.e. reproduced exactly on every level, which is what smoothed aggregation buys with
//   its near-nullspace vectors.
//   All set-up products are *gathers by the output entry* (one thread per (grid node, stencil offset)): no atomics,
//   bitwise reproducible hierarchies.  Grid nodes are walked in XCD-local bricks where neighbours share operands.
//   Smoother: Chebyshev of degree k (k = 1: damped block-Jacobi) on Dinv A, lambda_max by power iteration at set-up.
//   Coarsest grid (<= mg_coarse_nodes nodes): dense inverse computed once (device Gauss-Jordan), applied as one GEMV.
//   Partitioned runs: with glims_set_mg_frame every rank lays the SAME grids over the partitioned mesh -- level-0 passes
//   exact through halo exchanges, the first grid's operator and, per cycle, its restricted residual all-reduced, the
//   Cartesian levels computed redundantly: the iteration count does not depend on the rank count.  Without a frame the
//   hierarchy covers the rank's owned rows only (ghost couplings dropped inside the preconditioner = non-overlapping
//   additive Schwarz with one V-cycle per subdomain); the Krylov operator itself is exact either way.
#include "glims_internal.h"

#include <omp.h>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

struct GridDev {
  int n0, n1, n2;
  long long nn;
};
inline GridDev gdev(const MgGrid& g) { return GridDev{g.n[0], g.n[1], g.n[2], (long long)g.nn}; }

__device__ __forceinline__ void lin2v(long long I, const GridDev& g, int* v) {
  v[0] = (int)(I % g.n0);
  const long long t = I / g.n0;
  v[1] = (int)(t % g.n1);
  v[2] = (int)(t / g.n1);
}
__device__ __forceinline__ long long v2lin(const int* v, const GridDev& g) {
  return ((long long)v[2] * g.n1 + v[1]) * g.n0 + v[0];
}
__device__ __forceinline__ int gn(const GridDev& g, int a) { return a == 0 ? g.n0 : a == 1 ? g.n1 : g.n2; }

// Work box of a rank on a REPLICATED Cartesian level (partitioned runs with a global frame): the part of the grid this rank
// needs results on, grown by the smoothers' dependency margin.  nn = 0: the whole grid.  Threads enumerate the box, the
// arrays keep their global indexing; what lies outside the box is simply not recomputed (and is stale).
struct BoxDev {
  int lo0, lo1, lo2, n0, n1, n2;
  long long nn;
};
// node of thread t: false past the end; Iv / I in the level's global indexing
__device__ __forceinline__ bool box_node(const BoxDev& b, const GridDev& g, long long t, int* Iv, long long* I) {
  if (b.nn == 0) {
    if (t >= g.nn) return false;
    *I = t;
    lin2v(t, g, Iv);
    return true;
  }
  if (t >= b.nn) return false;
  Iv[0] = b.lo0 + (int)(t % b.n0);
  const long long q = t / b.n0;
  Iv[1] = b.lo1 + (int)(q % b.n1);
  Iv[2] = b.lo2 + (int)(q / b.n1);
  *I = v2lin(Iv, g);
  return true;
}

// Operator-sized arrays (stencil planes A / A16) of the first grid of a box-limited partitioned run are kept for the rank's
// STORAGE box only (= its work box at the smoother degree of the set-up): row of node Iv at op_row(), plane stride
// op_stride().  sb.nn = 0: the whole grid, global indexing.  (Vectors and the per-node arrays dinv / sc keep the level's
// global indexing either way: they are 1-3 % of the operator's bytes.)
__device__ __forceinline__ long long op_row(const BoxDev& sb, const int* Iv, long long I) {
  return sb.nn ? ((long long)(Iv[2] - sb.lo2) * sb.n1 + (Iv[1] - sb.lo1)) * sb.n0 + (Iv[0] - sb.lo0) : I;
}
__device__ __forceinline__ long long op_stride(const BoxDev& sb, const GridDev& g) { return sb.nn ? sb.nn : g.nn; }
// the lowest rank whose core box holds fine node iv (-1: nobody's): the owner rule of the masked restriction, also of the
// masked Galerkin product and of the replicated per-node arrays
__device__ __forceinline__ int core_owner(const int* __restrict__ cores, int n_ranks, const int* iv) {
  for (int q = 0; q < n_ranks; ++q) {
    const int* c = cores + q * 6;
    if (iv[0] >= c[0] && iv[0] <= c[3] && iv[1] >= c[1] && iv[1] <= c[4] && iv[2] >= c[2] && iv[2] <= c[5]) return q;
  }
  return -1;
}

template <int D>
__device__ __forceinline__ void off2v(int off, int R, int* o) {
  const int W = 2 * R + 1;
  o[0] = off % W - R;
  o[1] = (off / W) % W - R;
  o[2] = D == 3 ? off / (W * W) - R : 0;
}
template <int D>
__device__ __forceinline__ int v2off(const int* o, int R) {
  const int W = 2 * R + 1;
  return (D == 3 ? (o[2] + R) * W * W : 0) + (o[1] + R) * W + (o[0] + R);
}

// ---------------------------------------------------------------------------------------------------
// set-up kernels
// ---------------------------------------------------------------------------------------------------
// how far the Galerkin product mesh -> grid reaches: max over mesh edges (i, j) of the index distance between a parent
// of i and a parent of j (parents with non-zero weight only)
template <int D>
__global__ void k_mg_reach(int64_t n_own, int64_t n_col, GridDev g1, const int64_t* __restrict__ slice_ptr,
                           const int32_t* __restrict__ cols, const int32_t* __restrict__ cell0,
                           const double* __restrict__ wgt, unsigned long long* __restrict__ reach) {
  // reach[0]: stored entries seen, reach[1]: entries whose parents lie more than ONE grid cell apart (they need the
  // 125-point stencil), reach[2]: more than TWO apart (outside every stencil: dropped from the coarse operators)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_own) return;
  const int64_t base = slice_ptr[i >> 6];
  const int len = (int)((slice_ptr[(i >> 6) + 1] - base) >> 6);
  int ci[3];
  unsigned n_all = 0, n1 = 0, n2 = 0;
  lin2v(cell0[i], g1, ci);
  for (int k = 0; k < len; ++k) {
    const int64_t j = cols[base + (int64_t)k * GL_WAVE + (i & 63)];
    if (j >= n_col || j == i) continue;
    int cj[3], m = 0;
    lin2v(cell0[j], g1, cj);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const double wi = wgt[i * D + a], wj = wgt[j * D + a];
      const int lo_i = ci[a] + (wi == 1.0 ? 1 : 0), hi_i = ci[a] + (wi == 0.0 ? 0 : 1);
      const int lo_j = cj[a] + (wj == 1.0 ? 1 : 0), hi_j = cj[a] + (wj == 0.0 ? 0 : 1);
      m = max(m, max(hi_j - lo_i, hi_i - lo_j));
    }
    ++n_all;
    n1 += m > 1;
    n2 += m > 2;
  }
  if (n_all) atomicAdd(reach, (unsigned long long)n_all);
  if (n1) atomicAdd(reach + 1, (unsigned long long)n1);
  if (n2) atomicAdd(reach + 2, (unsigned long long)n2);
}

// level-1 cell and interpolation weights of every mesh node (internal numbering): cell = floor((x - lo) / H) clamped to
// the grid, weight towards the cell's upper node per axis; weights within 1e-6 of 0 / 1 are snapped (lattice-aligned nodes:
// exact weights, compact stencils)
template <int D>
__global__ void k_mg_node_cells(int64_t n, const double* __restrict__ xyz, double lo0, double lo1, double lo2, double H0,
                                double H1, double H2, GridDev g1, int32_t* __restrict__ cell0, double* __restrict__ wgt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double lo[3] = {lo0, lo1, lo2}, H[3] = {H0, H1, H2};
  long long lin = 0, stride = 1;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double t = (xyz[i * D + a] - lo[a]) / H[a];
    int c = (int)floor(t);
    c = max(0, min(gn(g1, a) - 2, c));
    double w = t - c;
    if (fabs(w) < 1e-6) w = 0.0;
    if (fabs(w - 1.0) < 1e-6) w = 1.0;
    w = fmax(0.0, fmin(1.0, w));
    wgt[i * D + a] = w;
    lin += (long long)c * stride;
    stride *= gn(g1, a);
  }
  cell0[i] = (int32_t)lin;
}
__global__ void k_mg_cell_keys(int64_t n, const int32_t* __restrict__ cell0, uint32_t* __restrict__ key,
                               int32_t* __restrict__ val) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  key[i] = (uint32_t)cell0[i];
  val[i] = (int32_t)i;
}

// bounding box (in level-1 cell coordinates) of the cells that hold this rank's OWNED mesh nodes: bb = {min[3], max[3]}
template <int D>
__global__ void k_mg_cell_bbox(int64_t n_own, GridDev g1, const int32_t* __restrict__ cell0, int* __restrict__ bb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_own) return;
  int cv[3];
  lin2v(cell0[i], g1, cv);
#pragma unroll
  for (int a = 0; a < D; ++a) {
    atomicMin(bb + a, cv[a]);
    atomicMax(bb + 3 + a, cv[a]);
  }
}

// first-grid residual of a box-limited partitioned cycle (GridExchange): message packing and the ordered sum
// sendbuf[(reg.off + local node) * BS + a] = r[a][node] over the nodes of every send region
template <int BS>
__global__ void k_gx_pack(GridDev g, int n_reg, const GridRegion* __restrict__ reg, long long n_send,
                          const double* __restrict__ r, double* __restrict__ sendbuf, const int* __restrict__ done) {
  if (done && *done) return;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_send) return;
  int k = 0;
  while (k + 1 < n_reg && reg[k + 1].off <= t) ++k;
  const GridRegion q = reg[k];
  const long long l = t - q.off;
  const int iv[3] = {q.lo[0] + (int)(l % q.n[0]), q.lo[1] + (int)((l / q.n[0]) % q.n[1]),
                     q.lo[2] + (int)(l / ((long long)q.n[0] * q.n[1]))};
  const long long I = v2lin(iv, g);
#pragma unroll
  for (int a = 0; a < BS; ++a) sendbuf[t * BS + a] = r[(long long)a * g.nn + I];
}
// r[.][I] = sum over the ranks, in ascending rank order on every rank (the same bits wherever two work boxes overlap), of
// their partial sums at I: this rank's own (already in r) and what the regions received hold; a thread per node of the box
template <int BS>
__global__ void k_gx_sum(GridDev g, BoxDev box, int my_rank, int n_reg, const GridRegion* __restrict__ reg,
                         const double* __restrict__ recvbuf, double* __restrict__ r, const int* __restrict__ done) {
  if (done && *done) return;
  int iv[3];
  long long I;
  if (!box_node(box, g, (long long)blockIdx.x * blockDim.x + threadIdx.x, iv, &I)) return;
  double own[BS], acc[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) {
    own[a] = r[(long long)a * g.nn + I];
    acc[a] = 0.0;
  }
  bool own_in = false;
  for (int k = 0; k < n_reg; ++k) {
    const GridRegion q = reg[k];
    if (!own_in && q.rank > my_rank) {
#pragma unroll
      for (int a = 0; a < BS; ++a) acc[a] += own[a];
      own_in = true;
    }
    const int d0 = iv[0] - q.lo[0], d1 = iv[1] - q.lo[1], d2 = iv[2] - q.lo[2];
    if (d0 < 0 || d0 >= q.n[0] || d1 < 0 || d1 >= q.n[1] || d2 < 0 || d2 >= q.n[2]) continue;
    const long long t = q.off + ((long long)d2 * q.n[1] + d1) * q.n[0] + d0;
#pragma unroll
    for (int a = 0; a < BS; ++a) acc[a] += recvbuf[t * BS + a];
  }
  if (!own_in) {
#pragma unroll
    for (int a = 0; a < BS; ++a) acc[a] += own[a];
  }
#pragma unroll
  for (int a = 0; a < BS; ++a) r[(long long)a * g.nn + I] = acc[a];
}

// Galerkin product mesh -> first grid, A1 = P^T K P, in two gathers (no atomics, fixed summation orders).
//
// Step 1, T = K P restricted to a box: the parents of the neighbours of mesh node i lie in the (2R + 2)^D grid nodes
// around i's cell (origin = cell - R); one WAVE per row, lane b owns box node b and walks the row's entries with the
// wave (broadcast loads), adding w_j,b K_ij where box node b is a parent of column j.  Entries whose parents fall outside
// the box are exactly the ones outside the stencil: dropped, as before.  T is kept in single precision (the product is
// stored in single precision anyway), rows [r0, r1) at a time so that its size stays bounded.
// Step 2, A1[I, off] += sum over the children i of I in [r0, r1) of w_iI T[i][I + off]: a thread per (grid node, stencil
// offset) walks the explicit restriction operator's entries of I.
// (Until round 3 one thread per (I, off) walked ALL rows of ALL children of the 2^D cells around I and tested every entry's
//  parents against I + off: 63x the operator's bytes in fetches, 29 ms at 1 M and 290 ms at 10 M mesh nodes --
//  profiles/r02_pmc_c5.json.)
template <int D, int BS>
__global__ __launch_bounds__(256) void k_mg_kp(GridDev g1, int R, int64_t r0, int64_t r1, int64_t n_col,
                                                const int32_t* __restrict__ cell0, const double* __restrict__ wgt,
                                                const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ cols,
                                                const double* __restrict__ vK, const uint8_t* __restrict__ fixed,
                                                float* __restrict__ T, int lump) {
  constexpr int B2 = BS * BS;
  const int E = 2 * R + 2, NB = D == 3 ? E * E * E : E * E;
  const int64_t i = r0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= r1) return;
  const int lane = threadIdx.x & 63;
  int ci[3];
  lin2v(cell0[i], g1, ci);
  const int64_t base = slice_ptr[i >> 6];
  const int len = (int)((slice_ptr[(i >> 6) + 1] - base) >> 6);
  const int rl = (int)(i & 63);
  int fi = 0;   // constrained components of the row's own node
  if (fixed) {
#pragma unroll
    for (int c = 0; c < BS; ++c) fi |= fixed[i * BS + c] ? 1 << c : 0;
  }
  // lane k holds what the wave needs to know about entry k of the row (column, its cell, its weights, whether it counts);
  // the loop below hands these round by cross-lane reads, so that the only memory accesses inside it are the (wave-uniform,
  // mutually independent) loads of the entry's values
  for (int k0 = 0; k0 < len; k0 += GL_WAVE) {   // rows longer than 64 entries: in batches (len <= 150)
    const int kk = k0 + lane;
    int64_t jl = 0;
    int cjl[3] = {0, 0, 0};
    double wl[3] = {0.0, 0.0, 0.0};
    int okl = 0;   // bit 0: the entry counts; bits 1 .. BS: constrained components of its column
    if (kk < len) {
      jl = cols[base + (int64_t)kk * GL_WAVE + rl];
      if (jl < n_col) {   // ghost column: dropped unless the hierarchy has a replicated (global) level
        okl = 1;
        lin2v(cell0[jl], g1, cjl);
#pragma unroll
        for (int a = 0; a < D; ++a) wl[a] = wgt[jl * D + a];
        if (fixed) {
#pragma unroll
          for (int c = 0; c < BS; ++c) okl |= fixed[jl * BS + c] ? 2 << c : 0;
        }
        // A POSITIVE coupling with parents more than R cells apart (a long edge opposite a dihedral angle near pi: slivers)
        // would lose its cross terms p_i p_j^T to the stencil radius while its share of the two diagonals stays: with
        // K = sum over edges of (-K_ij)(e_i - e_j)(e_i - e_j)^T + diag(row sums) that edge then contributes
        // -|K_ij| (p_i p_i^T + p_j p_j^T), negative semi-definite, and the first grid's operator becomes INDEFINITE
        // (measured on 300 k random points: eigenvalues of B A on the second grid down to -0.22, scalar RD operator
        // -0.88, by power iteration).  Such an edge is taken out whole instead -- its entry lumped onto the row's own
        // diagonal, symmetrised for blocks: A1' = A1 + |K_ij| (p_i - p_j)(p_i - p_j)^T stays positive definite and keeps the
        // row sums.  Far negative couplings keep losing their cross terms only (that errs on the definite side).
        if (lump) {
          int far = 0;   // some pair of parents with non-zero weights lies more than R cells apart
#pragma unroll
          for (int a = 0; a < D; ++a) {
            const double wi = wgt[i * D + a];
            const int lo_i = ci[a] + (wi >= 1.0 ? 1 : 0), hi_i = ci[a] + (wi > 0.0 ? 1 : 0);
            const int lo_j = cjl[a] + (wl[a] >= 1.0 ? 1 : 0), hi_j = cjl[a] + (wl[a] > 0.0 ? 1 : 0);
            far |= max(hi_i - lo_j, hi_j - lo_i) > R ? 1 : 0;
          }
          double tr = 0.0;
          const double* vk = vK + (base + (int64_t)kk * GL_WAVE) * B2 + rl;
#pragma unroll
          for (int a = 0; a < BS; ++a) tr += vk[(a * BS + a) * GL_WAVE];
          if (far && tr > 0.0 && jl != i) {
            okl |= 1 << 8;
#pragma unroll
            for (int a = 0; a < D; ++a) {
              cjl[a] = ci[a];
              wl[a] = wgt[i * D + a];
            }
          }
        }
      }
    }
    const int nk = min(GL_WAVE, len - k0);
    for (int b = lane; b < ((NB + 63) & ~63); b += GL_WAVE) {
      const bool live = b < NB;
      int Jv[3] = {0, 0, 0};
      Jv[0] = ci[0] - R + b % E;
      Jv[1] = ci[1] - R + (b / E) % E;
      if (D == 3) Jv[2] = ci[2] - R + b / (E * E);
      double acc[B2];
      float* t = T + ((i - r0) * NB + (live ? b : 0)) * B2;
#pragma unroll
      for (int e = 0; e < B2; ++e) acc[e] = k0 == 0 ? 0.0 : (double)t[e];
      for (int k = 0; k < nk; ++k) {
        const int ok = __shfl(okl, k, 64);
        double wj = live && (ok & 1) ? 1.0 : 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
          const int dj = Jv[a] - __shfl(cjl[a], k, 64);
          const double w = __shfl(wl[a], k, 64);
          wj *= dj == 0 ? 1.0 - w : dj == 1 ? w : 0.0;
        }
        const double* v = vK + (base + (int64_t)(k0 + k) * GL_WAVE) * B2 + rl;
#pragma unroll
        for (int e = 0; e < B2; ++e) {
          double val = v[e * GL_WAVE];
          if (BS > 1 && (ok >> 8)) val = 0.5 * (val + v[((e % BS) * BS + e / BS) * GL_WAVE]);   // lumped: symmetric part
          if (((fi >> (e / BS)) | (ok >> (1 + e % BS))) & 1) val = 0.0;
          acc[e] += wj * val;
        }
      }
      if (live) {
#pragma unroll
        for (int e = 0; e < B2; ++e) t[e] = (float)acc[e];
      }
    }
  }
}

template <int D, int BS>
__global__ void k_mg_ptkp(GridDev g1, int R, int S, int64_t r0, int64_t r1, const int64_t* __restrict__ pt_ptr,
                          const int32_t* __restrict__ pt_idx, const float* __restrict__ pt_w,
                          const int32_t* __restrict__ cell0, const float* __restrict__ T, double* __restrict__ A1d,
                          const BoxDev cb /*rows kept: this rank's core in a box-limited partitioned run, else the grid*/) {
  constexpr int B2 = BS * BS;
  const int E = 2 * R + 2, NB = D == 3 ? E * E * E : E * E;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long cnt = cb.nn ? cb.nn : g1.nn;
  if (t >= cnt * S) return;
  const long long tn = t / S;
  const int off = (int)(t - tn * S);
  int Iv[3], o[3], Jv[3] = {0, 0, 0};
  long long I;
  box_node(cb, g1, tn, Iv, &I);
  const long long Il = op_row(cb, Iv, I), ps = op_stride(cb, g1);
  off2v<D>(off, R, o);
  bool inside = true;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    Jv[a] = Iv[a] + o[a];
    inside = inside && Jv[a] >= 0 && Jv[a] < gn(g1, a);
  }
  if (!inside) return;   // (A1d is zero-initialised)
  double acc[B2];
#pragma unroll
  for (int e = 0; e < B2; ++e) acc[e] = A1d[((long long)off * B2 + e) * ps + Il];
  const long long sl = I >> 6;
  const int64_t base = pt_ptr[sl] + (I & 63);
  const int len = (int)((pt_ptr[sl + 1] - pt_ptr[sl]) >> 6);
  for (int k = 0; k < len; ++k) {
    const float w = pt_w[base + (int64_t)k * GL_WAVE];
    if (w == 0.0f) continue;   // padding
    const int64_t i = pt_idx[base + (int64_t)k * GL_WAVE];
    if (i < r0 || i >= r1) continue;
    int ci[3];
    lin2v(cell0[i], g1, ci);
    const int b = (Jv[0] - ci[0] + R) + E * ((Jv[1] - ci[1] + R) + (D == 3 ? E * (Jv[2] - ci[2] + R) : 0));
    const float* tp = T + ((i - r0) * NB + b) * B2;
#pragma unroll
    for (int e = 0; e < B2; ++e) acc[e] += (double)w * (double)tp[e];
  }
#pragma unroll
  for (int e = 0; e < B2; ++e) A1d[((long long)off * B2 + e) * ps + Il] = acc[e];
}

// Box-limited partitioned run: the first grid's operator rows travel by neighbour exchange instead of an all-reduce of the
// whole operator.  Rank q's partial rows (from its own mesh nodes) are non-zero on q's core only; rank p keeps the rows of its
// storage box, so p receives (box_p n core_q) from q -- the regions of the residual exchange (GridExchange), with S x B2
// values per node instead of BS.  Pack: sendbuf[(reg.off + local node) * SB + e] = A1d[e][core-local row].
__global__ void k_gx_pack_op(GridDev g, BoxDev cb, int SB, int n_reg, const GridRegion* __restrict__ reg, long long n_send,
                             const double* __restrict__ A1d, double* __restrict__ sendbuf) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_send * SB) return;
  const long long node = t / SB;
  const int e = (int)(t - node * SB);
  int k = 0;
  while (k + 1 < n_reg && reg[k + 1].off <= node) ++k;
  const GridRegion q = reg[k];
  const long long l = node - q.off;
  const int iv[3] = {q.lo[0] + (int)(l % q.n[0]), q.lo[1] + (int)((l / q.n[0]) % q.n[1]),
                     q.lo[2] + (int)(l / ((long long)q.n[0] * q.n[1]))};
  const long long I = v2lin(iv, g);
  sendbuf[t] = A1d[(long long)e * op_stride(cb, g) + op_row(cb, iv, I)];
}
// A[e][storage row of I] = sum over the ranks, in ascending rank order (the same bits wherever two storage boxes overlap), of
// their partial entries: this rank's own (if I lies in its core) and what the regions received hold.  Thread per (node, e).
__global__ void k_gx_sum_op(GridDev g, BoxDev sb, BoxDev cb, int my_rank, const int* __restrict__ my_core, int SB, int n_reg,
                            const GridRegion* __restrict__ reg, const double* __restrict__ recvbuf,
                            const double* __restrict__ A1d, float* __restrict__ A) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long cnt = sb.nn ? sb.nn : g.nn;
  if (t >= cnt * SB) return;
  const long long tn = t / SB;
  const int e = (int)(t - tn * SB);
  int iv[3];
  long long I;
  box_node(sb, g, tn, iv, &I);
  const bool mine = iv[0] >= my_core[0] && iv[0] <= my_core[3] && iv[1] >= my_core[1] && iv[1] <= my_core[4] &&
                    iv[2] >= my_core[2] && iv[2] <= my_core[5];
  const double own = mine ? A1d[(long long)e * op_stride(cb, g) + op_row(cb, iv, I)] : 0.0;
  double acc = 0.0;
  bool own_in = false;
  for (int k = 0; k < n_reg; ++k) {
    const GridRegion q = reg[k];
    if (!own_in && q.rank > my_rank) {
      acc += own;
      own_in = true;
    }
    const int d0 = iv[0] - q.lo[0], d1 = iv[1] - q.lo[1], d2 = iv[2] - q.lo[2];
    if (d0 < 0 || d0 >= q.n[0] || d1 < 0 || d1 >= q.n[1] || d2 < 0 || d2 >= q.n[2]) continue;
    acc += recvbuf[(q.off + ((long long)d2 * q.n[1] + d1) * q.n[0] + d0) * SB + e];
  }
  if (!own_in) acc += own;
  A[(long long)e * op_stride(sb, g) + op_row(sb, iv, I)] = (float)acc;
}
// x[.][I] = 0 unless this rank owns node I (lowest rank whose core holds it): after a sum over the ranks every node carries
// its owner's value.  `fill`: value for nodes nobody owns (added on rank 0 only).
template <int BS>
__global__ void k_mg_keep_owned(GridDev g, const int* __restrict__ cores, int n_ranks, int my_rank, double* __restrict__ x,
                                double fill) {
  const long long I = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= g.nn) return;
  int iv[3];
  lin2v(I, g, iv);
  const int owner = core_owner(cores, n_ranks, iv);
  if (owner == my_rank) return;
#pragma unroll
  for (int a = 0; a < BS; ++a) x[(long long)a * g.nn + I] = (owner < 0 && my_rank == 0) ? fill : 0.0;
}

// Transfer between two Cartesian levels.  Levels are boxes of ONE global index frame per level (partitioned runs:
// each rank stores the box its nodes touch; the levels below the first replicated one are such local boxes):
// global index = local index + offset, and the 2:1 relation (coarse J <-> fine 2J - 1, 2J, 2J + 1) holds between
// GLOBAL indices.  Single GPU: all offsets are 0.
struct Fac {
  int f[3];    // coarsening factor per axis (1 | 2)
  int of[3];   // global index of the fine box's node 0
  int oc[3];   // global index of the coarse box's node 0
};
// local fine index of the child d (-1, 0, 1) of local coarse index I along axis a
__device__ __forceinline__ int child_index(const Fac& fc, int a, int I, int d) {
  return (fc.f[a] == 2 ? 2 * (I + fc.oc[a]) + d : I + fc.oc[a]) - fc.of[a];
}
// 1-D interpolation weight of fine index j with respect to coarse index J (factor f)
__device__ __forceinline__ double w1d(int f, int j, int J) {
  if (f == 1) return j == J ? 1.0 : 0.0;
  const int dlt = j - 2 * J;
  return dlt == 0 ? 1.0 : (dlt == 1 || dlt == -1) ? 0.5 : 0.0;
}

// Galerkin product between two Cartesian levels, one thread per (coarse node, stencil offset)
template <int D, int BS>
__global__ void k_mg_rap(GridDev gf, GridDev gc, Fac fc, int R, int S, const float* __restrict__ Af,
                         float* __restrict__ Ac, const BoxDev sbf /*storage box of the fine operator*/,
                         const int* __restrict__ cores /*with a storage box: only the fine rows this rank owns count (the
                         partial products are all-reduced afterwards)*/, int n_ranks, int my_rank) {
  constexpr int B2 = BS * BS;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= gc.nn * S) return;
  const long long I = t / S;
  const int off = (int)(t - I * S);
  int Iv[3], o[3], Jv[3] = {0, 0, 0};
  lin2v(I, gc, Iv);
  off2v<D>(off, R, o);
  bool inside = true;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    Jv[a] = Iv[a] + o[a];
    inside = inside && Jv[a] >= 0 && Jv[a] < gn(gc, a);
  }
  double acc[B2];
#pragma unroll
  for (int e = 0; e < B2; ++e) acc[e] = 0.0;
  if (inside) {
    const int nch = D == 3 ? 27 : 9;
    for (int ch = 0; ch < nch; ++ch) {
      int dv[3] = {ch % 3 - 1, (ch / 3) % 3 - 1, D == 3 ? ch / 9 - 1 : 0};
      int iv[3] = {0, 0, 0};
      double wi = 1.0;
      bool ok = true;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        if (fc.f[a] == 1) {
          if (dv[a] != 0) ok = false;
        } else {
          wi *= dv[a] == 0 ? 1.0 : 0.5;
        }
        iv[a] = child_index(fc, a, Iv[a], dv[a]);
        ok = ok && iv[a] >= 0 && iv[a] < gn(gf, a);
      }
      if (ok && cores) ok = core_owner(cores, n_ranks, iv) == my_rank;
      if (!ok) continue;
      const long long i = op_row(sbf, iv, v2lin(iv, gf)), psf = op_stride(sbf, gf);
      // fine stencil offsets that land on a child of J: |(iv + of) - f J| <= f - 1 in global indices
      int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const int ctr = fc.f[a] * (Jv[a] + fc.oc[a]) - (iv[a] + fc.of[a]);
        lo[a] = max(-R, ctr - (fc.f[a] - 1));
        hi[a] = min(R, ctr + (fc.f[a] - 1));
      }
      int of[3] = {0, 0, 0};
      for (of[2] = lo[2]; of[2] <= hi[2]; ++of[2])
        for (of[1] = lo[1]; of[1] <= hi[1]; ++of[1])
          for (of[0] = lo[0]; of[0] <= hi[0]; ++of[0]) {
            double wj = 1.0;
            bool in = true;
#pragma unroll
            for (int a = 0; a < D; ++a) {
              const int jv = iv[a] + of[a];
              in = in && jv >= 0 && jv < gn(gf, a);
              wj *= w1d(fc.f[a], jv + fc.of[a], Jv[a] + fc.oc[a]);
            }
            if (!in || wj == 0.0) continue;
            const int oi = v2off<D>(of, R);
            const double ww = wi * wj;
#pragma unroll
            for (int e = 0; e < B2; ++e) acc[e] += ww * (double)Af[((long long)oi * B2 + e) * psf + i];
          }
    }
  }
#pragma unroll
  for (int e = 0; e < B2; ++e) Ac[((long long)off * B2 + e) * gc.nn + I] = (float)acc[e];
}

template <int BS>
__device__ __forceinline__ void inv_block(double (*A)[BS], double* o) {
  // dofs without any stiffness (no free child): identity, so that the block stays invertible and the dof stays zero
#pragma unroll
  for (int a = 0; a < BS; ++a)
    if (!(A[a][a] > 0.0)) {
#pragma unroll
      for (int b = 0; b < BS; ++b) A[a][b] = A[b][a] = 0.0;
      A[a][a] = 1.0;
    }
  if constexpr (BS == 1) {
    o[0] = 1.0 / A[0][0];
  } else if constexpr (BS == 2) {
    const double inv = 1.0 / (A[0][0] * A[1][1] - A[0][1] * A[1][0]);
    o[0] = A[1][1] * inv;
    o[1] = -A[0][1] * inv;
    o[2] = -A[1][0] * inv;
    o[3] = A[0][0] * inv;
  } else {
    const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1];
    const double c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    const double c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const double inv = 1.0 / (A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02);
    o[0] = c00 * inv;
    o[1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * inv;
    o[2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * inv;
    o[3] = c01 * inv;
    o[4] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * inv;
    o[5] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * inv;
    o[6] = c02 * inv;
    o[7] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * inv;
    o[8] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * inv;
  }
}

template <int D, int BS>
__global__ void k_mg_dinv(GridDev g, int S, const float* __restrict__ A, double* __restrict__ dinv, const BoxDev sb) {
  constexpr int B2 = BS * BS;
  int Iv[3];
  long long I;
  if (!box_node(sb, g, (long long)blockIdx.x * blockDim.x + threadIdx.x, Iv, &I)) return;
  const long long Il = op_row(sb, Iv, I), ps = op_stride(sb, g);
  const int ctr = S / 2;
  double M[BS][BS], o[B2];
#pragma unroll
  for (int a = 0; a < BS; ++a)
#pragma unroll
    for (int b = 0; b < BS; ++b) M[a][b] = (double)A[((long long)ctr * B2 + a * BS + b) * ps + Il];
  // the product is symmetric up to the single-precision rounding of its entries: symmetrise the block
#pragma unroll
  for (int a = 0; a < BS; ++a)
#pragma unroll
    for (int b = a + 1; b < BS; ++b) M[a][b] = M[b][a] = 0.5 * (M[a][b] + M[b][a]);
  inv_block<BS>(M, o);
#pragma unroll
  for (int e = 0; e < B2; ++e) dinv[(long long)e * g.nn + I] = o[e];
}

// ---------------------------------------------------------------------------------------------------
// cycle kernels, Cartesian levels (vectors component-major [BS][nn])
// ---------------------------------------------------------------------------------------------------
// what every Cartesian pass does with the row sum acc = (A xin)_I (one thread per node holds it)
template <int BS, int MODE>
__device__ __forceinline__ void cart_epilogue(long long nn, long long I, const double* acc,
                                              const double* __restrict__ dinv, const double* __restrict__ xin,
                                              const double* __restrict__ r, double* __restrict__ d,
                                              double* __restrict__ xout, double c1, double c2,
                                              const double* __restrict__ osc) {
  double t[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) t[a] = MODE == 2 ? acc[a] : r[(long long)a * nn + I] - acc[a];
  if (MODE == 0) {
#pragma unroll
    for (int a = 0; a < BS; ++a) xout[(long long)a * nn + I] = osc ? t[a] / osc[(long long)a * nn + I] : t[a];
    return;
  }
#pragma unroll
  for (int a = 0; a < BS; ++a) {
    double z = 0.0;
#pragma unroll
    for (int b = 0; b < BS; ++b) z += dinv[(long long)(a * BS + b) * nn + I] * t[b];
    if (MODE == 2) {
      xout[(long long)a * nn + I] = z;
    } else {
      const double dn = (c1 != 0.0 ? c1 * d[(long long)a * nn + I] : 0.0) + c2 * z;
      const double xn = xin[(long long)a * nn + I] + dn;
      d[(long long)a * nn + I] = dn;
      xout[(long long)a * nn + I] = osc ? osc[(long long)a * nn + I] * xn : xn;
    }
  }
}

// MODE 0: xout = r - A xin     MODE 1: d = c1 d + c2 Dinv (r - A xin); xout = xin + d     MODE 2: xout = Dinv A xin
// osc (levels smoothed in scaled variables, MgLevel::half): MODE 0 leaves them, xout = (r~ - A~ x~) / s; MODE 1 with osc
// is the last step of the cycle on this level, xout = s (x~ + d~)
template <int D, int BS, int MODE, class AT>
__global__ __launch_bounds__(256) void k_mg_cart(GridDev g, int R, const AT* __restrict__ A,
                                                  const double* __restrict__ dinv, const double* __restrict__ xin,
                                                  const double* __restrict__ r, double* __restrict__ d,
                                                  double* __restrict__ xout, double c1, double c2,
                                                  const int* __restrict__ done, const double* __restrict__ osc,
                                                  const BoxDev box, const BoxDev sb) {
  constexpr int B2 = BS * BS;
  if (done && *done) return;   // launches enqueued past the Krylov solver's convergence: nobody reads the result
  int Iv[3];
  long long I;
  if (!box_node(box, g, (long long)blockIdx.x * blockDim.x + threadIdx.x, Iv, &I)) return;
  const long long Il = op_row(sb, Iv, I), ps = op_stride(sb, g);
  double acc[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) acc[a] = 0.0;
  const int W = 2 * R + 1;
  const int zlo = D == 3 ? max(-R, -Iv[2]) : 0, zhi = D == 3 ? min(R, g.n2 - 1 - Iv[2]) : 0;
  const int ylo = max(-R, -Iv[1]), yhi = min(R, g.n1 - 1 - Iv[1]);
  const int xlo = max(-R, -Iv[0]), xhi = min(R, g.n0 - 1 - Iv[0]);
  for (int oz = zlo; oz <= zhi; ++oz)
    for (int oy = ylo; oy <= yhi; ++oy) {
      const long long nb0 = I + ((long long)oz * g.n1 + oy) * g.n0;
      const int off0 = (D == 3 ? (oz + R) * W * W : 0) + (oy + R) * W + R;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const long long nb = nb0 + ox;
        const AT* a0 = A + (long long)(off0 + ox) * B2 * ps + Il;
        double xb[BS];
#pragma unroll
        for (int b = 0; b < BS; ++b) xb[b] = xin[(long long)b * g.nn + nb];
#pragma unroll
        for (int a = 0; a < BS; ++a)
#pragma unroll
          for (int b = 0; b < BS; ++b) acc[a] += (double)(float)a0[(long long)(a * BS + b) * ps] * xb[b];
      }
    }
  cart_epilogue<BS, MODE>(g.nn, I, acc, dinv, xin, r, d, xout, c1, c2, osc);
}

// The same pass for SMALL grids: one wave per node, the stencil offsets dealt to the lanes, then a fixed shuffle tree.
// A thread per node walks its 27 (125) offsets one after the other -- on a grid of a few thousand nodes that serial
// chain, not bandwidth, is the whole kernel time (12-18 us per pass at 5^3 .. 26^3 nodes, measured).
template <int D, int BS, int MODE>
__global__ __launch_bounds__(256) void k_mg_cart_w(GridDev g, int R, int S, const float* __restrict__ A,
                                                    const double* __restrict__ dinv, const double* __restrict__ xin,
                                                    const double* __restrict__ r, double* __restrict__ d,
                                                    double* __restrict__ xout, double c1, double c2,
                                                    const int* __restrict__ done) {
  constexpr int B2 = BS * BS;
  if (done && *done) return;
  const long long I = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (I >= g.nn) return;
  const int lane = threadIdx.x & 63;
  int Iv[3];
  lin2v(I, g, Iv);
  double acc[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) acc[a] = 0.0;
  for (int off = lane; off < S; off += GL_WAVE) {
    int o[3], nv[3] = {0, 0, 0};
    off2v<D>(off, R, o);
    bool in = true;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      nv[a] = Iv[a] + o[a];
      in = in && nv[a] >= 0 && nv[a] < gn(g, a);
    }
    if (!in) continue;
    const long long nb = v2lin(nv, g);
    const float* a0 = A + (long long)off * B2 * g.nn + I;
    double xb[BS];
#pragma unroll
    for (int b = 0; b < BS; ++b) xb[b] = xin[(long long)b * g.nn + nb];
#pragma unroll
    for (int a = 0; a < BS; ++a)
#pragma unroll
      for (int b = 0; b < BS; ++b) acc[a] += (double)a0[(long long)(a * BS + b) * g.nn] * xb[b];
  }
#pragma unroll
  for (int a = 0; a < BS; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[a] += __shfl_down(acc[a], o, 64);
  }
  if (lane != 0) return;
  double t[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) t[a] = MODE == 2 ? acc[a] : r[(long long)a * g.nn + I] - acc[a];
  if (MODE == 0) {
#pragma unroll
    for (int a = 0; a < BS; ++a) xout[(long long)a * g.nn + I] = t[a];
    return;
  }
#pragma unroll
  for (int a = 0; a < BS; ++a) {
    double z = 0.0;
#pragma unroll
    for (int b = 0; b < BS; ++b) z += dinv[(long long)(a * BS + b) * g.nn + I] * t[b];
    if (MODE == 2) {
      xout[(long long)a * g.nn + I] = z;
    } else {
      const double dn = (c1 != 0.0 ? c1 * d[(long long)a * g.nn + I] : 0.0) + c2 * z;
      d[(long long)a * g.nn + I] = dn;
      xout[(long long)a * g.nn + I] = xin[(long long)a * g.nn + I] + dn;
    }
  }
}

// The same pass for MEDIUM grids (a few thousand to a few ten thousand nodes): G lanes per node, each walking every
// G-th stencil offset, partial sums combined by a fixed xor-shuffle tree.  Lane = sub * (64 / G) + node-in-wave, so that
// the lanes of one `sub` read 64 / G consecutive nodes of one operator plane.  A thread per node is a serial chain of
// 27 (125) x B2 loads with too few threads to fill the device at these sizes; a wave per node wastes 37 of 64 lanes.
template <int D, int BS, int MODE, int G, class AT>
__global__ __launch_bounds__(256) void k_mg_cart_g(GridDev g, int R, int S, const AT* __restrict__ A,
                                                    const double* __restrict__ dinv, const double* __restrict__ xin,
                                                    const double* __restrict__ r, double* __restrict__ d,
                                                    double* __restrict__ xout, double c1, double c2,
                                                    const int* __restrict__ done, const double* __restrict__ osc,
                                                    const BoxDev box, const BoxDev sb) {
  constexpr int B2 = BS * BS, NPW = GL_WAVE / G;   // nodes per wave
  if (done && *done) return;
  const int lane = threadIdx.x & 63;
  const int sub = lane / NPW;
  int Iv[3] = {0, 0, 0};
  long long I = 0;
  const bool live = box_node(box, g, ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * NPW + (lane % NPW), Iv, &I);
  double acc[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) acc[a] = 0.0;
  if (live)
    for (int off = sub; off < S; off += G) {
      int o[3], nv[3] = {0, 0, 0};
      off2v<D>(off, R, o);
      bool in = true;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        nv[a] = Iv[a] + o[a];
        in = in && nv[a] >= 0 && nv[a] < gn(g, a);
      }
      if (!in) continue;
      const long long nb = v2lin(nv, g);
      const long long ps = op_stride(sb, g);
      const AT* a0 = A + (long long)off * B2 * ps + op_row(sb, Iv, I);
      double xb[BS];
#pragma unroll
      for (int b = 0; b < BS; ++b) xb[b] = xin[(long long)b * g.nn + nb];
#pragma unroll
      for (int a = 0; a < BS; ++a)
#pragma unroll
        for (int b = 0; b < BS; ++b) acc[a] += (double)(float)a0[(long long)(a * BS + b) * ps] * xb[b];
    }
#pragma unroll
  for (int a = 0; a < BS; ++a) {
#pragma unroll
    for (int o = GL_WAVE / 2; o >= NPW; o >>= 1) acc[a] += __shfl_xor(acc[a], o, 64);
  }
  if (!live || sub != 0) return;
  cart_epilogue<BS, MODE>(g.nn, I, acc, dinv, xin, r, d, xout, c1, c2, osc);
}

// first smoothing step from a zero iterate: d = c2 Dinv r, x = d   (no operator pass)
// sc (MgLevel::half): the level enters its scaled variables here, r <- r~ = s r in place
template <int D, int BS>
__global__ void k_mg_first_cart(GridDev g, const double* __restrict__ dinv, double* __restrict__ r,
                                double* __restrict__ d, double* __restrict__ x, double c2,
                                const double* __restrict__ sc, const int* __restrict__ done, const BoxDev box) {
  if (done && *done) return;
  int Iv[3];
  long long I;
  if (!box_node(box, g, (long long)blockIdx.x * blockDim.x + threadIdx.x, Iv, &I)) return;
  double rv[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) {
    rv[a] = r[(long long)a * g.nn + I];
    if (sc) {
      rv[a] *= sc[(long long)a * g.nn + I];
      r[(long long)a * g.nn + I] = rv[a];
    }
  }
#pragma unroll
  for (int a = 0; a < BS; ++a) {
    double z = 0.0;
#pragma unroll
    for (int b = 0; b < BS; ++b) z += dinv[(long long)(a * BS + b) * g.nn + I] * rv[b];
    d[(long long)a * g.nn + I] = c2 * z;
    x[(long long)a * g.nn + I] = c2 * z;
  }
}
// level 0, first smoothing step from a zero iterate, entering the scaled variables: r~ = S r, d~ = c2 Dinv~ r~, x~ = d~
template <int BS, class XT>
__global__ void k_mg_first_fine(int64_t n_own, const float* __restrict__ dinv, const double* __restrict__ sc,
                                const double* __restrict__ r, XT* __restrict__ rs, XT* __restrict__ d,
                                XT* __restrict__ x, double c2, const int* __restrict__ done) {
  if (done && *done) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_own) return;
  double rv[BS], xv[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) {
    rv[a] = sc[i * BS + a] * r[i * BS + a];
    rs[i * BS + a] = (XT)rv[a];
  }
#pragma unroll
  for (int a = 0; a < BS; ++a) {
    double z = 0.0;
#pragma unroll
    for (int b = 0; b < BS; ++b) z += (double)dinv[i * BS * BS + a * BS + b] * rv[b];
    xv[a] = c2 * z;
    d[i * BS + a] = (XT)xv[a];
  }
  XNode<BS, XT>::store(x, i, xv);
}

// Interpolation weight of a mesh node towards corner `corner` of its grid cell, from the per-axis weights.  Rounded to
// single precision ONCE here: the explicit restriction operator stores these values and the prolongation evaluates the
// same expression, so that R = P^T holds bit for bit (on lattice meshes the factors are 0, 1/2, 1: nothing is rounded).
template <int D>
__device__ __forceinline__ double corner_weight(const double* __restrict__ w3, int corner) {
  double w = 1.0;
#pragma unroll
  for (int a = 0; a < D; ++a) w *= ((corner >> a) & 1) ? w3[a] : 1.0 - w3[a];
  return (double)(float)w;
}

// Restriction mesh -> grid as an EXPLICIT sparse operator, SELL-64 over the grid nodes (x fastest): entry k of grid node I
// = (child mesh node, weight) at pt_ptr[I / 64] + k * 64 + I % 64.  Until round 3 the restriction walked the children
// lists of the 2^D cells around a node and rebuilt every weight from the children's per-axis weights (24 B per child and
// visit, each child visited by its 8 parents): 462 us at 10 M mesh nodes with scalar unknowns, more than two operator
// passes of level 0.  The operator costs 8 B per (child, parent) pair with a non-zero weight -- 27 per grid node on a
// lattice mesh with H = 2 h, i.e. 3.4 per mesh node -- read in unit stride.
// set-up, pass 1: number of children with a non-zero weight
template <int D>
__global__ void k_mg_pt_count(GridDev g1, const int32_t* __restrict__ cell_ptr, const int32_t* __restrict__ cell_nodes,
                              const double* __restrict__ wgt, int32_t* __restrict__ cnt) {
  const long long I = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= g1.nn) return;
  int Iv[3];
  lin2v(I, g1, Iv);
  int n = 0;
  for (int corner = 0; corner < (1 << D); ++corner) {
    int cv[3] = {0, 0, 0};
    bool ok = true;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      cv[a] = Iv[a] - ((corner >> a) & 1);
      ok = ok && cv[a] >= 0 && cv[a] <= gn(g1, a) - 2;
    }
    if (!ok) continue;
    const long long c = v2lin(cv, g1);
    for (int32_t q = cell_ptr[c]; q < cell_ptr[c + 1]; ++q)
      n += corner_weight<D>(wgt + (int64_t)cell_nodes[q] * D, corner) != 0.0;
  }
  cnt[I] = n;
}
// set-up, pass 2: the entries, corner after corner and child after child (a fixed order); slots past a node's own count
// repeat a valid child index with weight 0
template <int D>
__global__ void k_mg_pt_fill(GridDev g1, const int32_t* __restrict__ cell_ptr, const int32_t* __restrict__ cell_nodes,
                             const double* __restrict__ wgt, const int64_t* __restrict__ pt_ptr,
                             int32_t* __restrict__ idx, float* __restrict__ wv) {
  const long long I = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // padded to whole slices
  const long long s = I >> 6;
  if (s >= (g1.nn + GL_WAVE - 1) / GL_WAVE) return;
  const int lane = (int)(I & 63);
  const int64_t base = pt_ptr[s];
  const int len = (int)((pt_ptr[s + 1] - base) >> 6);
  int k = 0;
  if (I < g1.nn) {
    int Iv[3];
    lin2v(I, g1, Iv);
    for (int corner = 0; corner < (1 << D); ++corner) {
      int cv[3] = {0, 0, 0};
      bool ok = true;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        cv[a] = Iv[a] - ((corner >> a) & 1);
        ok = ok && cv[a] >= 0 && cv[a] <= gn(g1, a) - 2;
      }
      if (!ok) continue;
      const long long c = v2lin(cv, g1);
      for (int32_t q = cell_ptr[c]; q < cell_ptr[c + 1]; ++q) {
        const int32_t i = cell_nodes[q];
        const double w = corner_weight<D>(wgt + (int64_t)i * D, corner);
        if (w == 0.0) continue;
        idx[base + (int64_t)k * GL_WAVE + lane] = i;
        wv[base + (int64_t)k * GL_WAVE + lane] = (float)w;
        ++k;
      }
    }
  }
  for (; k < len; ++k) {
    idx[base + (int64_t)k * GL_WAVE + lane] = 0;
    wv[base + (int64_t)k * GL_WAVE + lane] = 0.0f;
  }
}
// r1[I] = sum_k w_k res[child_k]: one thread per grid node, 8 (index, weight, gather) triples in flight
// With dinv1 the first smoothing step of the grid level (from a zero iterate: k_mg_first_cart) rides along: the thread
// has its node's restricted residual in registers.
template <int BS, class XT>
__global__ __launch_bounds__(256) void k_mg_restrict0(long long nn, const int64_t* __restrict__ pt_ptr,
                                                       const int32_t* __restrict__ idx, const float* __restrict__ wv,
                                                       const XT* __restrict__ res, double* __restrict__ r1,
                                                       const double* __restrict__ dinv1, const double* __restrict__ sc1,
                                                       double* __restrict__ d1, double* __restrict__ x1, double c2,
                                                       const int* __restrict__ done) {
  if (done && *done) return;
  const long long I = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // grid covers whole slices
  const long long s = I >> 6;
  if (s >= (nn + GL_WAVE - 1) / GL_WAVE) return;
  const int64_t base = pt_ptr[s] + (I & 63);
  const int len = (int)((pt_ptr[s + 1] - pt_ptr[s]) >> 6);
  double acc[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) acc[a] = 0.0;
  for (int k = 0; k < len; k += 8) {
    int32_t ci[8];
    float wk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t e = base + (int64_t)min(k + j, len - 1) * GL_WAVE;
      ci[j] = __builtin_nontemporal_load(idx + e);
      wk[j] = k + j < len ? __builtin_nontemporal_load(wv + e) : 0.0f;
    }
    double rv[8][BS];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int a = 0; a < BS; ++a) rv[j][a] = (double)res[(int64_t)ci[j] * BS + a];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int a = 0; a < BS; ++a) acc[a] += (double)wk[j] * rv[j][a];
  }
  if (I >= nn) return;
  if (dinv1 && sc1) {   // the level works in scaled variables from here on: r~ = s r
#pragma unroll
    for (int a = 0; a < BS; ++a) acc[a] *= sc1[(long long)a * nn + I];
  }
#pragma unroll
  for (int a = 0; a < BS; ++a) r1[(long long)a * nn + I] = acc[a];
  if (dinv1) {
#pragma unroll
    for (int a = 0; a < BS; ++a) {
      double z = 0.0;
#pragma unroll
      for (int b = 0; b < BS; ++b) z += dinv1[(long long)(a * BS + b) * nn + I] * acc[b];
      d1[(long long)a * nn + I] = c2 * z;
      x1[(long long)a * nn + I] = c2 * z;
    }
  }
}

// prolongation grid -> mesh, into the scaled level-0 variables: x~_i = x~_i + F_i S_i^-1 sum_{parents} w e_J
template <int D, int BS, class XT>
__global__ void k_mg_prolong0(GridDev g1, int64_t n_own, const int32_t* __restrict__ cell0,
                              const double* __restrict__ wgt, const uint8_t* __restrict__ fixed,
                              const double* __restrict__ sc, const double* __restrict__ e1,
                              const XT* __restrict__ xin, XT* __restrict__ xout, const int* __restrict__ done) {
  if (done && *done) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_own) return;
  int cv[3];
  lin2v(cell0[i], g1, cv);
  double acc[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) acc[a] = 0.0;
  for (int corner = 0; corner < (1 << D); ++corner) {
    int pv[3] = {0, 0, 0};
#pragma unroll
    for (int a = 0; a < D; ++a) pv[a] = cv[a] + ((corner >> a) & 1);
    const double w = corner_weight<D>(wgt + i * D, corner);   // the value the restriction operator stores
    if (w == 0.0) continue;
    const long long J = v2lin(pv, g1);
#pragma unroll
    for (int a = 0; a < BS; ++a) acc[a] += w * e1[(long long)a * g1.nn + J];
  }
  double xv[BS];
  XNode<BS, XT>::load(xin, i, xv);
#pragma unroll
  for (int a = 0; a < BS; ++a)
    xv[a] = (fixed && fixed[i * BS + a]) ? 0.0 : xv[a] + acc[a] / sc[i * BS + a];   // x~ = S^-1 x
  XNode<BS, XT>::store(xout, i, xv);
}

template <int D, int BS>
__global__ __launch_bounds__(256) void k_mg_restrict(GridDev gf, GridDev gc, Fac fc, const double* __restrict__ res,
                                                      double* __restrict__ rc, const double* __restrict__ dinv_c,
                                                      double* __restrict__ d_c, double* __restrict__ x_c, double c2,
                                                      const int* __restrict__ done, const int* __restrict__ cores,
                                                      int n_ranks, int my_rank) {
  // cores (partitioned runs, box-limited fine level): [n_ranks][6] = lo / hi (inclusive) of every rank's core box on the
  // FINE level; a child counts on the lowest rank whose core holds it (a partition of unity: the partial sums of the ranks
  // are all-reduced afterwards), children in nobody's core carry no residual
  if (done && *done) return;
  // one wave per coarse node, its (up to) 3^D children dealt to the lanes, fixed shuffle tree.  With dinv_c the first
  // smoothing step of the coarse level (from a zero iterate: d = x = c2 Dinv r, k_mg_first_cart) rides along.
  const long long I = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (I >= gc.nn) return;
  const int lane = threadIdx.x & 63;
  int Iv[3];
  lin2v(I, gc, Iv);
  double acc[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) acc[a] = 0.0;
  const int nch = D == 3 ? 27 : 9;
  if (lane < nch) {
    const int ch = lane;
    int dv[3] = {ch % 3 - 1, (ch / 3) % 3 - 1, D == 3 ? ch / 9 - 1 : 0};
    int iv[3] = {0, 0, 0};
    double wi = 1.0;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      if (fc.f[a] == 1) {
        if (dv[a] != 0) ok = false;
      } else {
        wi *= dv[a] == 0 ? 1.0 : 0.5;
      }
      iv[a] = child_index(fc, a, Iv[a], dv[a]);
      ok = ok && iv[a] >= 0 && iv[a] < gn(gf, a);
    }
    if (ok && cores) {
      int owner = -1;
      for (int q = 0; q < n_ranks && owner < 0; ++q) {
        const int* c = cores + q * 6;
        if (iv[0] >= c[0] && iv[0] <= c[3] && iv[1] >= c[1] && iv[1] <= c[4] && iv[2] >= c[2] && iv[2] <= c[5]) owner = q;
      }
      ok = owner == my_rank;
    }
    if (ok) {
      const long long i = v2lin(iv, gf);
#pragma unroll
      for (int a = 0; a < BS; ++a) acc[a] = wi * res[(long long)a * gf.nn + i];
    }
  }
#pragma unroll
  for (int a = 0; a < BS; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[a] += __shfl_down(acc[a], o, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < BS; ++a) rc[(long long)a * gc.nn + I] = acc[a];
    if (dinv_c) {
#pragma unroll
      for (int a = 0; a < BS; ++a) {
        double z = 0.0;
#pragma unroll
        for (int b = 0; b < BS; ++b) z += dinv_c[(long long)(a * BS + b) * gc.nn + I] * acc[b];
        d_c[(long long)a * gc.nn + I] = c2 * z;
        x_c[(long long)a * gc.nn + I] = c2 * z;
      }
    }
  }
}

template <int D, int BS>
__device__ __forceinline__ void prolong_node(const GridDev& gf, const GridDev& gc, const Fac& fc,
                                             const double* __restrict__ ec, const double* __restrict__ xin,
                                             double* __restrict__ xout, const double* __restrict__ isc, const int* iv,
                                             long long i);
// isc (fine level in scaled variables): x~ += (P e) / s
template <int D, int BS>
__global__ void k_mg_prolong(GridDev gf, GridDev gc, Fac fc, const double* __restrict__ ec,
                             const double* __restrict__ xin, double* __restrict__ xout,
                             const double* __restrict__ isc, const int* __restrict__ done, const BoxDev box) {
  if (done && *done) return;
  int iv[3];
  long long i;
  if (!box_node(box, gf, (long long)blockIdx.x * blockDim.x + threadIdx.x, iv, &i)) return;
  prolong_node<D, BS>(gf, gc, fc, ec, xin, xout, isc, iv, i);
}
template <int D, int BS>
__device__ __forceinline__ void prolong_node(const GridDev& gf, const GridDev& gc, const Fac& fc,
                                             const double* __restrict__ ec, const double* __restrict__ xin,
                                             double* __restrict__ xout, const double* __restrict__ isc, const int* iv,
                                             long long i) {
  double acc[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) acc[a] = 0.0;
  for (int corner = 0; corner < (1 << D); ++corner) {
    int pv[3] = {0, 0, 0};
    double w = 1.0;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const int up = (corner >> a) & 1;
      const int gfi = iv[a] + fc.of[a];   // global fine index
      if (fc.f[a] == 1) {
        if (up) ok = false;
        pv[a] = gfi - fc.oc[a];
      } else if ((gfi & 1) == 0) {
        if (up) ok = false;
        pv[a] = (gfi >> 1) - fc.oc[a];
      } else {
        pv[a] = (gfi >> 1) + up - fc.oc[a];
        w *= 0.5;
      }
      ok = ok && pv[a] >= 0 && pv[a] < gn(gc, a);
    }
    if (!ok) continue;
    const long long J = v2lin(pv, gc);
#pragma unroll
    for (int a = 0; a < BS; ++a) acc[a] += w * ec[(long long)a * gc.nn + J];
  }
#pragma unroll
  for (int a = 0; a < BS; ++a)
    xout[(long long)a * gf.nn + i] = xin[(long long)a * gf.nn + i] + (isc ? acc[a] / isc[(long long)a * gf.nn + i] : acc[a]);
}

// coarsest level: x = Ainv r, one wave per row of the dense inverse
__global__ __launch_bounds__(256) void k_mg_dense(int n, const double* __restrict__ Ainv, const double* __restrict__ r,
                                                   double* __restrict__ x, const int* __restrict__ done) {
  if (done && *done) return;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  double v = 0.0;
  for (int k = lane; k < n; k += GL_WAVE) v += Ainv[(size_t)row * n + k] * r[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if (lane == 0) x[row] = v;
}

__global__ void k_mg_f2d(int64_t n, const float* __restrict__ a, double* __restrict__ b) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = (double)a[i];
}
__global__ void k_mg_d2f(int64_t n, const double* __restrict__ a, float* __restrict__ b) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = (float)a[i];
}
__global__ void k_mg_mask_to_double(int64_t n, const uint8_t* __restrict__ m, double* __restrict__ d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = m[i] ? 1.0 : 0.0;
}
__global__ void k_mg_double_to_mask(int64_t n, const double* __restrict__ d, uint8_t* __restrict__ m) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) m[i] = d[i] != 0.0 ? 1 : 0;
}

__global__ void k_mg_fill(int64_t n, double* __restrict__ x, const uint8_t* __restrict__ fixed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // deterministic pseudo-random start vector of the power iteration
  unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  x[i] = (fixed && fixed[i]) ? 0.0 : (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}
__global__ void k_mg_scale(int64_t n, double* __restrict__ x, double s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
}

// A level enters symmetrically scaled variables (MgLevel::half): s = 1 / sqrt(a_ii) per dof (1 where the level's
// operator has no stiffness), Dinv <- S^-1 Dinv S^-1 = inverse diagonal blocks of S A S ...
template <int D, int BS>
__global__ void k_mg_level_scale(GridDev g, int S, const float* __restrict__ A, double* __restrict__ sc,
                                 double* __restrict__ dinv, const BoxDev sb) {
  constexpr int B2 = BS * BS;
  int Iv[3];
  long long I;
  if (!box_node(sb, g, (long long)blockIdx.x * blockDim.x + threadIdx.x, Iv, &I)) return;
  const long long Il = op_row(sb, Iv, I), ps = op_stride(sb, g);
  const int ctr = S / 2;
  double s[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) {
    const double aii = (double)A[((long long)ctr * B2 + a * BS + a) * ps + Il];
    s[a] = aii > 0.0 ? 1.0 / sqrt(aii) : 1.0;
    sc[(long long)a * g.nn + I] = s[a];
  }
#pragma unroll
  for (int a = 0; a < BS; ++a)
#pragma unroll
    for (int b = 0; b < BS; ++b) dinv[(long long)(a * BS + b) * g.nn + I] /= s[a] * s[b];
}
// ... and its operator becomes the half-precision copy of S A S (entries <= 1 in magnitude: no overflow, and whatever
// falls below the half-precision range is 1e-5 of a diagonal entry).  One thread per (node, stencil offset).
template <int D, int BS>
__global__ void k_mg_half_copy(GridDev g, int R, int S, const float* __restrict__ A, const double* __restrict__ sc,
                               _Float16* __restrict__ A16, const BoxDev sb) {
  constexpr int B2 = BS * BS;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long cnt = sb.nn ? sb.nn : g.nn;
  if (t >= cnt * S) return;
  const int off = (int)(t / cnt);
  int Iv[3], o[3], Jv[3] = {0, 0, 0};
  long long I;
  box_node(sb, g, t - (long long)off * cnt, Iv, &I);
  const long long Il = op_row(sb, Iv, I), ps = op_stride(sb, g);
  off2v<D>(off, R, o);
  bool in = true;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    Jv[a] = Iv[a] + o[a];
    in = in && Jv[a] >= 0 && Jv[a] < gn(g, a);
  }
  const long long J = in ? v2lin(Jv, g) : I;
#pragma unroll
  for (int a = 0; a < BS; ++a)
#pragma unroll
    for (int b = 0; b < BS; ++b) {
      const long long e = ((long long)off * B2 + a * BS + b) * ps + Il;
      A16[e] = in ? (_Float16)(float)(sc[(long long)a * g.nn + I] * (double)A[e] * sc[(long long)b * g.nn + J]) : (_Float16)0.0f;
    }
}

inline unsigned gridn(long long n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// Chebyshev coefficients of step m (0-based) for the interval [lmax / ratio, lmax] of Dinv A (glims_options.
// mg_cheb_ratio; default 30 on lattice meshes, 10 on general ones), lmax = 1.1 x the power-
// iteration estimate.  The lower end decides how much of the spectrum the smoother takes on itself: with lmax / 4 (the
// first version) degree 2 / 3 needed 31 / 25 PCG iterations on config C5, with lmax / 30 ... lmax / 60 they need
// 24 / 17 (Delaunay mesh: 42 / 34 -> 34 / 25; beyond lmax / 100 it gets worse again) -- tools/proto_gmg.py.
struct Cheb {
  double theta, delta, sigma, rho;
  Cheb(double lam, double ratio) {
    const double lmax = 1.1 * lam, lmin = lmax / ratio;
    theta = 0.5 * (lmax + lmin);
    delta = 0.5 * (lmax - lmin);
    sigma = theta / delta;
    rho = 1.0 / sigma;
  }
  void next(int m, double* c1, double* c2) {
    if (m == 0) {
      *c1 = 0.0;
      *c2 = 1.0 / theta;
      rho = 1.0 / sigma;
      return;
    }
    const double rn = 1.0 / (2.0 * sigma - rho);
    *c1 = rn * rho;
    *c2 = 2.0 * rn / delta;
    rho = rn;
  }
};

}  // namespace

// ===================================================================================================
// mesh metrics, HOST version (glims_create behind the test hook GLIMS_HOST_SYMBOLIC; the product computes them on the
// device, symbolic.hip): bounding box, lattice detection, mean edge length, coordinates in internal numbering
// ===================================================================================================
void gl_mesh_metrics(glims_ctx* h, const HostPattern& hp, const double* xyz_old) {
  MeshMetrics& mm = h->mm;
  const int d = h->dim;
  const int64_t n = h->n_own;
  std::vector<double> xyzn((size_t)h->n_nodes * d);   // coordinates in the internal numbering
  const int nth = gl_host_threads();   // not the runtime's default team (one thread per visible hardware thread)
  (void)nth;                           // (the device pass of the compiler does not see the OpenMP clauses)
#pragma omp parallel for schedule(static) num_threads(nth)
  for (int64_t i = 0; i < h->n_nodes; ++i)   // ghosts keep their place in the numbering (new2old = identity there)
    for (int a = 0; a < d; ++a) xyzn[i * d + a] = xyz_old[(int64_t)hp.new2old[i] * d + a];
  for (int a = 0; a < 3; ++a) {
    mm.lo[a] = 1e300;
    mm.hi[a] = -1e300;
  }
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < d; ++a) {
      mm.lo[a] = std::min(mm.lo[a], xyzn[i * d + a]);
      mm.hi[a] = std::max(mm.hi[a], xyzn[i * d + a]);
    }
  for (int a = d; a < 3; ++a) mm.lo[a] = mm.hi[a] = 0.0;
  // edges from the SELL pattern (owned columns only): smallest positive coordinate difference per axis, mean length
  double hmin[3] = {1e300, 1e300, 1e300}, esum = 0.0;
  int64_t ecnt = 0;
  double ext[3];
  for (int a = 0; a < 3; ++a) ext[a] = std::max(mm.hi[a] - mm.lo[a], 1e-300);
#pragma omp parallel num_threads(nth)
  {
    double hm[3] = {1e300, 1e300, 1e300}, es = 0.0;
    int64_t ec = 0;
#pragma omp for schedule(static) nowait
    for (int32_t s = 0; s < hp.n_slices; ++s) {
      const int64_t base = hp.slice_ptr[s];
      const int len = (int)((hp.slice_ptr[s + 1] - base) / GL_WAVE);
      for (int l = 0; l < GL_WAVE; ++l) {
        const int64_t i = (int64_t)s * GL_WAVE + l;
        if (i >= n) break;
        for (int k = 0; k < len; ++k) {
          const int64_t j = hp.cols[base + (int64_t)k * GL_WAVE + l];
          if (j <= i || j >= n) continue;
          double e2 = 0.0;
          for (int a = 0; a < d; ++a) {
            const double dl = std::fabs(xyzn[i * d + a] - xyzn[j * d + a]);
            e2 += dl * dl;
            if (dl > 1e-9 * ext[a]) hm[a] = std::min(hm[a], dl);
          }
          es += std::sqrt(e2);
          ++ec;
        }
      }
    }
#pragma omp critical
    {
      for (int a = 0; a < 3; ++a) hmin[a] = std::min(hmin[a], hm[a]);
      esum += es;
      ecnt += ec;
    }
  }
  mm.mean_edge = ecnt > 0 ? esum / (double)ecnt : ext[0];
  // lattice test: every coordinate an integer multiple of the axis' smallest edge component (box meshes)
  bool lat = true;
  for (int a = 0; a < d && lat; ++a) {
    if (!(hmin[a] < 1e299) || (mm.hi[a] - mm.lo[a]) / hmin[a] > 1e5) {
      lat = false;
      break;
    }
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad) num_threads(nth)
    for (int64_t i = 0; i < n; ++i) {
      const double t = (xyzn[i * d + a] - mm.lo[a]) / hmin[a];
      if (std::fabs(t - std::round(t)) > 1e-6) ++bad;
    }
    lat = bad == 0;
    mm.h_lattice[a] = hmin[a];
  }
  mm.lattice = lat;
  h->xyz_new.upload(xyzn, h->st);
  GL_HIP(hipStreamSynchronize(h->st));
}

// ===================================================================================================
// set-up
// ===================================================================================================
namespace {

// Work box of this rank on the replicated first grid for smoothers of degree `deg`: its core (the grid nodes its own mesh
// nodes interpolate from) grown by the dependency margin.  A pass of radius R is exact R nodes inside whatever its input
// was exact on; the input outside the box is stale.  Down: deg - 1 stencil passes and the residual (exact on box - deg R,
// needed on the core); up: the post-smoother starts from the pre-smoothed iterate (box - (deg - 1) R) and takes deg passes
// (exact on box - (2 deg - 1) R, needed on the core).  Hence the margin (2 deg - 1) R.
inline BoxDev sbox_of(const MgLevel& L) {
  return BoxDev{L.sb[0], L.sb[1], L.sb[2], L.sb[3], L.sb[4], L.sb[5], L.sb_nn};
}
template <int D>
BoxDev mg_work_box(const MgHierarchy& mg, const MgGrid& g, int deg) {
  const int m = (2 * deg - 1) * mg.R;
  int lo[3] = {0, 0, 0}, n[3] = {1, 1, 1};
  long long nn = 1;
  for (int a = 0; a < D; ++a) {
    lo[a] = std::max(0, mg.core[a] - m);
    const int hi = std::min(g.n[a] - 1, mg.core[3 + a] + m);
    n[a] = hi - lo[a] + 1;
    nn *= n[a];
  }
  if (nn >= g.nn) return BoxDev{0, 0, 0, 0, 0, 0, 0};
  return BoxDev{lo[0], lo[1], lo[2], n[0], n[1], n[2], nn};
}

// One operator pass on a Cartesian level.  Kernel by size: a wave per node up to 6 k nodes, 4 lanes per node for medium
// grids (<= 60 k nodes with 27-point stencils -- 26^3: -2.4 % per solve against a thread per node, 8 / 16 lanes the same,
// on 51^3 a thread per node is faster -- and every larger grid with 125-point stencils: 1 M-point Delaunay mesh 133.5 ->
// 126.7 ms per solve), a thread per node otherwise.  `osc`: see k_mg_cart (levels with MgLevel::half only).
template <int D, int BS>
void mg_apply_cart(glims_ctx* h, MgHierarchy& mg, MgLevel& L, int R, int mode, const double* xin, const double* r, double* d,
                   double* xout, double c1, double c2, const int* done = nullptr, const double* osc = nullptr,
                   const BoxDev box = BoxDev{0, 0, 0, 0, 0, 0, 0}) {
  const GridDev g = gdev(L.g);
  const int S = mg.S;
  const BoxDev sb = sbox_of(L);                   // where the level's operator planes keep their rows
  const long long cnt = box.nn ? box.nn : g.nn;   // threads' nodes: the work box of a partitioned run, or the grid
  GL_REQUIRE(!(L.half && mode == 2), "internal: the single-precision planes of this level are gone");   // (power iteration: before)
  const bool half = L.half;
  const _Float16* A16 = (const _Float16*)L.A16.p;
  if (g.nn <= 6000) {
    GL_REQUIRE(!half && box.nn == 0 && sb.nn == 0, "internal: half-precision operator / work box on a small grid");
    const unsigned gw = gridn(g.nn, 4);
    if (mode == 0)
      hipLaunchKernelGGL((k_mg_cart_w<D, BS, 0>), dim3(gw), dim3(256), 0, h->st, g, R, S, L.A.p, L.dinv.p, xin, r, d, xout, c1, c2, done);
    else if (mode == 1)
      hipLaunchKernelGGL((k_mg_cart_w<D, BS, 1>), dim3(gw), dim3(256), 0, h->st, g, R, S, L.A.p, L.dinv.p, xin, r, d, xout, c1, c2, done);
    else
      hipLaunchKernelGGL((k_mg_cart_w<D, BS, 2>), dim3(gw), dim3(256), 0, h->st, g, R, S, L.A.p, L.dinv.p, xin, r, d, xout, c1, c2, done);
    GL_HIP(hipGetLastError());
    return;
  }
  const bool lanes = g.nn <= 60000 || R >= 2;
  const unsigned grid = lanes ? gridn(cnt, 4 * (GL_WAVE / 4)) : gridn(cnt);
#define GL_CART(MODE)                                                                                                 \
  do {                                                                                                               \
    if (lanes && half)                                                                                               \
      hipLaunchKernelGGL((k_mg_cart_g<D, BS, MODE, 4, _Float16>), dim3(grid), dim3(256), 0, h->st, g, R, S, A16, L.dinv.p, \
                         xin, r, d, xout, c1, c2, done, osc, box, sb);                                               \
    else if (lanes)                                                                                                  \
      hipLaunchKernelGGL((k_mg_cart_g<D, BS, MODE, 4, float>), dim3(grid), dim3(256), 0, h->st, g, R, S, L.A.p, L.dinv.p, \
                         xin, r, d, xout, c1, c2, done, osc, box, sb);                                               \
    else if (half)                                                                                                   \
      hipLaunchKernelGGL((k_mg_cart<D, BS, MODE, _Float16>), dim3(grid), dim3(256), 0, h->st, g, R, A16, L.dinv.p, xin, r, \
                         d, xout, c1, c2, done, osc, box, sb);                                                       \
    else                                                                                                             \
      hipLaunchKernelGGL((k_mg_cart<D, BS, MODE, float>), dim3(grid), dim3(256), 0, h->st, g, R, L.A.p, L.dinv.p, xin, r,  \
                         d, xout, c1, c2, done, osc, box, sb);                                                       \
  } while (0)
  if (mode == 0) GL_CART(0); else if (mode == 1) GL_CART(1); else GL_CART(2);
#undef GL_CART
  GL_HIP(hipGetLastError());
}

// Dense inverse of the coarsest operator, in place on the device: Gauss-Jordan without pivoting (the matrix is SPD
// after the shift; a pivot <= 0 raises `bad`, the caller retries with a larger shift).  Two launches per pivot -- the
// old column is saved and the pivot row scaled by one block, then every other row is eliminated by a thread per entry:
// 3 k launches for the 1 536 unknowns of an 8^3 grid = 15 ms, where Cholesky + n triangular solves on the host's
// sixteen cores took 0.4 s (and 50 ms already for 6^3).
__global__ __launch_bounds__(1024) void k_gj_pivot(int n, int k, double* __restrict__ A, double* __restrict__ col,
                                                    int* __restrict__ bad) {
  if (*bad) return;
  const double p = A[(size_t)k * n + k];
  for (int i = threadIdx.x; i < n; i += blockDim.x) col[i] = A[(size_t)i * n + k];
  __syncthreads();   // every read of column k (the pivot among them) comes before the row is rewritten
  if (!(p > 0.0)) {
    if (threadIdx.x == 0) *bad = 1;
    return;
  }
  const double ip = 1.0 / p;
  for (int j = threadIdx.x; j < n; j += blockDim.x) A[(size_t)k * n + j] = j == k ? ip : A[(size_t)k * n + j] * ip;
}
__global__ __launch_bounds__(256) void k_gj_eliminate(int n, int k, double* __restrict__ A,
                                                       const double* __restrict__ col, const int* __restrict__ bad) {
  if (*bad) return;
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= n || i == k) return;
  const double f = col[i], rk = A[(size_t)k * n + j];
  A[(size_t)i * n + j] = j == k ? -f * rk : A[(size_t)i * n + j] - f * rk;
}
// A (device, n x n, row-major) -> inverse of A + shift_rel * max diag.  `diag_max` from the host copy.
bool dense_spd_inverse_device(glims_ctx* h, double* A, int n, double* col, int* bad) {
  GL_HIP(hipMemsetAsync(bad, 0, sizeof(int), h->st));
  for (int k = 0; k < n; ++k) {
    hipLaunchKernelGGL(k_gj_pivot, dim3(1), dim3(1024), 0, h->st, n, k, A, col, bad);
    hipLaunchKernelGGL(k_gj_eliminate, dim3((n + 255) / 256, n), dim3(256), 0, h->st, n, k, A, col, bad);
  }
  GL_HIP(hipGetLastError());
  int hb = 0;
  GL_HIP(hipMemcpyAsync(&hb, bad, sizeof(int), hipMemcpyDeviceToHost, h->st));
  GL_HIP(hipStreamSynchronize(h->st));
  return hb == 0;
}

template <int D, int BS>
void mg_setup_t(glims_ctx* h, MgHierarchy& mg) {
  constexpr int B2 = BS * BS;
  const MeshMetrics& mm = h->mm;
  const DevPattern& p = h->pat;
  const int64_t n = h->n_own;
  const double t_start = omp_get_wtime();
  const bool verbose = getenv("GLIMS_VERBOSE") != nullptr;
  double t_last = t_start;
  auto lap = [&](const char* what) {
    if (!verbose) return;
    GL_HIP(hipStreamSynchronize(h->st));
    const double t = omp_get_wtime();
    fprintf(stderr, "glims multigrid set-up: %-42s %8.1f ms\n", what, 1e3 * (t - t_last));
    t_last = t;
  };
  mg.clear();
  GL_REQUIRE(mg.op_vals && mg.op_dinv && mg.bs == BS, "multigrid set-up without an operator");
  GL_REQUIRE(h->xyz_new.n == (size_t)h->n_nodes * D, "mesh coordinates missing");
  const uint8_t* fx = mg.op_fixed;
  // spacing of the first grid in units of the mesh width.  One GPU: 2 (measured on config C5, zero guess: 16 / 22 / 28
  // iterations and 19.5 / 25.8 / 33.6 ms per solve with 2 / 3 / 4).  Partitioned runs with a global frame replicate the
  // Cartesian levels on every rank -- work that does not shrink with the rank count -- so a coarser first grid pays from
  // three ranks on: per-rank operator complexity 1 + 1.8 * 1.14 * ranks / factor^3, i.e. 2.03 -> 1.30 at 4 ranks with 3,
  // 3.05 -> 1.26 at 8 ranks with 4, and the residual all-reduce per cycle shrinks by the same factor^3 / 8.
  const double hf = h->opt.mg_h_factor > 0.5 ? h->opt.mg_h_factor
                    : !(h->world > 1 && h->mg_frame_set) || h->world <= 2 ? 2.0 : h->world <= 6 ? 3.0 : 4.0;
  const int coarse_max = std::max(8, h->opt.mg_coarse_nodes);

  // ---- partitioned runs: one global index frame for the auxiliary grids -------------------------------------------
  // With glims_set_mg_frame (the global bounding box) every rank lays the SAME Cartesian grids over the mesh and all
  // Cartesian levels are REPLICATED: each rank computes its part of the first grid's Galerkin operator (rows of its
  // owned mesh nodes, ghost columns included -- the product is additive over the rows), the parts are summed once by
  // an all-reduce, the coarser operators follow locally (identical on every rank); in every cycle the restricted
  // residual is all-reduced on the first grid, and the level-0 passes see the ghost values through a halo exchange.
  // The cycle is then the single-GPU cycle, evaluated in a distributed way: the iteration count does not depend on the
  // number of ranks (rank-local hierarchies: 31 -> 65+ iterations at 2 ranks).  The price is redundant work on the
  // Cartesian levels (1/8 of the mesh nodes on the first one) and 24 B per first-grid node of all-reduce per cycle;
  // intermediate rank-local boxes of the same frame were tried and rejected: their operators are partial row sums, and
  // smoothing with them next to an exact coarse operator made the iteration count WORSE (65 against 31 at 2 ranks).
  const bool multi = h->world > 1;
  const bool framed = multi && h->mg_frame_set;
  const int64_t n_all = framed ? h->n_nodes : n;        // nodes with grid coordinates (ghosts only in framed runs)
  bool lattice = mm.lattice;
  double h_lat[3] = {mm.h_lattice[0], mm.h_lattice[1], mm.h_lattice[2]}, mean_edge = mm.mean_edge;
  dvec<double> sc;                                       // scratch for the collective decisions of the set-up
  sc.alloc_zero(8, h->st);
  auto agree = [&](double* v, int m) {                   // sum over ranks (identity on one rank)
    if (!multi) return;
    GL_HIP(hipMemcpyAsync(sc.p, v, m * sizeof(double), hipMemcpyHostToDevice, h->st));
    gl_allreduce_bulk(h, sc.p, (size_t)m);
    GL_HIP(hipMemcpyAsync(v, sc.p, m * sizeof(double), hipMemcpyDeviceToHost, h->st));
    GL_HIP(hipStreamSynchronize(h->st));
  };
  if (framed) {   // every rank must take the same decisions: lattice only if all agree, spacings averaged
    double v[5] = {lattice ? 1.0 : 0.0, h_lat[0], h_lat[1], h_lat[2], mean_edge};
    agree(v, 5);
    lattice = v[0] == (double)h->world;
    for (int a = 0; a < 3; ++a) h_lat[a] = v[1 + a] / h->world;
    mean_edge = v[4] / h->world;
  }
  double flo[3], fhi[3];
  for (int a = 0; a < 3; ++a) {
    flo[a] = framed ? h->mg_frame_lo[a] : mm.lo[a];
    fhi[a] = framed ? h->mg_frame_hi[a] : mm.hi[a];
  }

  // ---- level 1 grid: spacing, origin, node -> cell map (host), children lists ------------------------------------
  double H[3] = {1, 1, 1};
  for (int a = 0; a < D; ++a) H[a] = lattice ? hf * h_lat[a] : hf * mean_edge / 1.2;
  MgGrid g1;
  int ng1[3] = {1, 1, 1}, o1[3] = {0, 0, 0};             // global dims of level 1, offset of this rank's box
  int S_try = D == 3 ? 27 : 9, widenings = 0;
  for (int attempt = 0;; ++attempt) {
    for (int a = 0; a < D; ++a) {
      const int cells = std::max(1, (int)std::ceil((fhi[a] - flo[a]) / H[a] - 1e-9));
      ng1[a] = cells + 1;
      mg.lo[a] = flo[a];
      mg.H[a] = H[a];
    }
    if (framed) {   // replicated grids must fit: widen the spacing (the same decision on every rank: global numbers only)
      double nng = 1.0;
      for (int a = 0; a < D; ++a) nng *= ng1[a];
      // the cap is a byte budget of the replicated first-grid operator: S_try = stencil entries assumed so far (27-point
      // until a pass of k_mg_reach has asked for the 125-point stencil)
      const double cap = (double)GL_MG_GLOBAL_BYTES / ((double)S_try * B2 * sizeof(float));
      if (nng > cap && widenings < 8) {
        const double fac = 1.02 * std::pow(nng / cap, 1.0 / D);
        for (int a = 0; a < D; ++a) H[a] *= fac;
        ++widenings;
        --attempt;   // not one of the (two) widenings for edges that reach too far
        continue;
      }
    }
    g1 = MgGrid();
    g1.nn = 1;
    for (int a = 0; a < D; ++a) {
      o1[a] = 0;                                         // (offsets: see Fac; every level is the whole grid)
      g1.n[a] = ng1[a];
      g1.nn *= g1.n[a];
    }
    GL_REQUIRE(g1.nn < (int64_t(1) << 31), "auxiliary grid too large");
    mg.cell0.alloc((size_t)n_all);
    mg.wgt.alloc((size_t)n_all * D);
    hipLaunchKernelGGL(k_mg_node_cells<D>, dim3(gridn(n_all)), dim3(256), 0, h->st, n_all, h->xyz_new.p, flo[0], flo[1],
                       flo[2], H[0], H[1], H[2], gdev(g1), mg.cell0.p, mg.wgt.p);
    GL_HIP(hipGetLastError());
    dvec<unsigned long long> reach;
    reach.alloc_zero(3, h->st);
    hipLaunchKernelGGL(k_mg_reach<D>, dim3(gridn(n)), dim3(256), 0, h->st, n, n_all, gdev(g1), p.slice_ptr.p, p.cols.p,
                       mg.cell0.p, mg.wgt.p, reach.p);
    GL_HIP(hipGetLastError());
    unsigned long long rc[3] = {0, 0, 0};
    GL_HIP(hipMemcpyAsync(rc, reach.p, sizeof(rc), hipMemcpyDeviceToHost, h->st));
    GL_HIP(hipStreamSynchronize(h->st));
    double fl[3] = {(double)rc[0], (double)rc[1], (double)rc[2]};
    if (framed) agree(fl, 3);                            // the stencil radius is a property of the whole hierarchy
    // Stencil radius: 1 if no edge's parents are more than one cell apart (lattice meshes), else 2.  Edges that reach
    // further than two cells -- the long slivers every Delaunay mesh has on its hull, a graded region -- are simply not
    // formed in the coarse operators (each (I, J) block and its transpose alike: the products gather by stencil
    // offset), level 0 keeps them.  Only when that concerns more than 2 % of the entries is the grid widened (twice at
    // most): widening until the LONGEST edge fits, as the first version did, ends with a 3 x 3 x 3 grid on a 1 M-point
    // Delaunay mesh (hull slivers span the domain) and 2 859 iterations.
    mg.dropped_fraction = fl[0] > 0.0 ? fl[2] / fl[0] : 0.0;
    if (mg.dropped_fraction <= 0.02 || attempt >= 2) {
      mg.R = fl[1] > 0.0 ? 2 : 1;
      const int S_need = D == 3 ? (mg.R == 2 ? 125 : 27) : (mg.R == 2 ? 25 : 9);
      if (framed && S_need > S_try) {   // the budget above was checked for the compact stencil: once more for the wide one
        S_try = S_need;
        --attempt;
        continue;
      }
      break;
    }
    for (int a = 0; a < D; ++a) H[a] *= 1.5;
  }
  // partitioned run: this rank's core on the first grid (bounding box of the grid nodes its OWN mesh nodes interpolate from)
  // and everybody's, for the box-limited cycle (mg_cycle_cart)
  mg.boxed = false;
  std::vector<int> all_cores;
  if (framed) {
    const int big = 1 << 30;
    std::vector<int> bb = {big, big, big, -big, -big, -big};
    dvec<int> dbb;
    dbb.upload(bb, h->st);
    if (n > 0) hipLaunchKernelGGL(k_mg_cell_bbox<D>, dim3(gridn(n)), dim3(256), 0, h->st, n, gdev(g1), mg.cell0.p, dbb.p);
    GL_HIP(hipGetLastError());
    GL_HIP(hipMemcpyAsync(bb.data(), dbb.p, 6 * sizeof(int), hipMemcpyDeviceToHost, h->st));
    GL_HIP(hipStreamSynchronize(h->st));
    for (int a = 0; a < 3; ++a) {
      const bool any = a < D && n > 0 && bb[a] <= bb[3 + a];
      mg.core[a] = any ? bb[a] : 0;
      mg.core[3 + a] = any ? std::min(g1.n[a] - 1, bb[3 + a] + 1) : (a < D && n > 0 ? -1 : 0);   // cell c touches nodes c, c + 1
    }
    std::vector<double> all((size_t)h->world * 6, 0.0);
    for (int q = 0; q < 6; ++q) all[(size_t)h->rank * 6 + q] = (double)mg.core[q];
    dvec<double> dall;
    dall.upload(all, h->st);
    gl_allreduce_bulk(h, dall.p, all.size());
    GL_HIP(hipMemcpyAsync(all.data(), dall.p, all.size() * sizeof(double), hipMemcpyDeviceToHost, h->st));
    GL_HIP(hipStreamSynchronize(h->st));
    all_cores.resize(all.size());
    for (size_t q = 0; q < all.size(); ++q) all_cores[q] = (int)std::lround(all[q]);
    mg.cores.upload(all_cores, h->st);
  }
  // ---- box-limited first grid (partitioned runs): decided BEFORE its operator is built, because the operator is then kept
  // for the rank's work box only.  Only where the thread-per-node / lanes-per-node kernels run and a coarser level exists; a
  // decision from replicated numbers, the same on every rank (the cycle then holds one more all-reduce).
  // Worth it when the passes saved outweigh the one small all-reduce the masked restriction adds per cycle (taken as 40 us
  // over RCCL): saving = (1 - largest box / grid, at the set-up's smoother degree) x (2 deg + 1) passes x the first grid's
  // half-precision operator bytes at 4 TB/s, required to be 1.5 x the cost -- e.g. a 55^3 grid of 27-point 3 x 3 stencils
  // (config C4's size on 8 ranks with spacing 4 h) with boxes of 22 %: 110 us saved.  GLIMS_MG_BOX_MIN_NODES (test hook)
  // replaces the estimate by a plain size threshold so that small test meshes take the path.
  mg.boxed = false;
  mg.box_fraction = 1.0;
  mg.gx.reset();
  mg.S = 1;
  for (int a = 0; a < D; ++a) mg.S *= 2 * mg.R + 1;
  BoxDev sbox1{0, 0, 0, 0, 0, 0, 0};   // storage box of the first grid's operator planes (nn = 0: the whole grid)
  if (framed && h->world > 1 && g1.nn > coarse_max && (h->opt.flags & GLIMS_FLAG_MG_WHOLE_GRID) == 0) {
    const char* e = getenv("GLIMS_MG_BOX_MIN_NODES");
    const long long min_nodes = e ? std::max(6001ll, atoll(e)) : 0;
    const int deg0 = BS == 1 ? (h->opt.rd_mg_smooth > 0 ? h->opt.rd_mg_smooth : (lattice ? 1 : 3)) : std::max(1, h->opt.mg_smooth);
    const MgGrid& g = g1;
    double worst = 0.0;
    MgHierarchy probe;   // (only core / R are read)
    probe.R = mg.R;
    for (int q = 0; q < h->world; ++q) {
      for (int a = 0; a < 6; ++a) probe.core[a] = all_cores[(size_t)q * 6 + a];
      const BoxDev b = mg_work_box<D>(probe, g, deg0);
      const double f = b.nn ? (double)b.nn / (double)g.nn : 1.0;
      worst = std::max(worst, f);
      if (q == h->rank) mg.box_fraction = f;
    }
    const double saved_us = (1.0 - worst) * (2.0 * deg0 + 1.0) * (double)g.nn * mg.S * B2 * 2.0 / 4.0e6;
    mg.boxed = g.nn > 6000 && (min_nodes ? g.nn >= min_nodes && worst <= 0.8 : saved_us >= 60.0);
    if (!mg.boxed) mg.box_fraction = 1.0;
    // the first-grid residual (every cycle) and operator rows (once, below) by neighbour exchange (GridExchange): regions
    // from the cores every rank holds
    if (mg.boxed) {
      sbox1 = mg_work_box<D>(mg, g, deg0);
      GridExchange& x = mg.gx;
      x.deg = deg0;
      auto box_of = [&](int q, int* lo, int* hi) {   // work box of rank q, inclusive (whole grid if mg_work_box says so)
        for (int a = 0; a < 6; ++a) probe.core[a] = all_cores[(size_t)q * 6 + a];
        const BoxDev b = mg_work_box<D>(probe, g, deg0);
        const int l3[3] = {b.lo0, b.lo1, b.lo2}, n3[3] = {b.n0, b.n1, b.n2};
        for (int a = 0; a < 3; ++a) {
          lo[a] = b.nn ? l3[a] : 0;
          hi[a] = b.nn ? l3[a] + n3[a] - 1 : g.n[a] - 1;
        }
      };
      auto cut = [&](const int* blo, const int* bhi, int q, GridRegion* out) {   // box n core_q
        long long vol = 1;
        for (int a = 0; a < 3; ++a) {
          const int lo = std::max(blo[a], all_cores[(size_t)q * 6 + a]), hi = std::min(bhi[a], all_cores[(size_t)q * 6 + 3 + a]);
          out->lo[a] = lo;
          out->n[a] = hi - lo + 1;
          if (hi < lo) return 0ll;
          vol *= out->n[a];
        }
        out->rank = q;
        return vol;
      };
      int mylo[3], myhi[3];
      box_of(h->rank, mylo, myhi);
      std::vector<GridRegion> sreg, rreg;
      x.send_ptr.assign(1, 0);
      x.recv_ptr.assign(1, 0);
      for (int q = 0; q < h->world; ++q) {
        if (q == h->rank) continue;
        GridRegion rs, rr;
        int qlo[3], qhi[3];
        box_of(q, qlo, qhi);
        const long long vs = cut(qlo, qhi, h->rank, &rs);   // what q needs of my partial sums: box_q n core_me
        const long long vr = cut(mylo, myhi, q, &rr);        // what I need of q's: box_me n core_q
        if (vs == 0 && vr == 0) continue;
        x.peers.push_back(q);
        if (vs > 0) {
          rs.rank = q;
          rs.off = x.n_send;
          sreg.push_back(rs);
          x.n_send += vs;
        }
        if (vr > 0) {
          rr.off = x.n_recv;
          rreg.push_back(rr);
          x.n_recv += vr;
        }
        x.send_ptr.push_back(x.n_send);
        x.recv_ptr.push_back(x.n_recv);
      }
      x.n_send_reg = (int)sreg.size();
      x.n_recv_reg = (int)rreg.size();
      if (!sreg.empty()) x.send_reg.upload(sreg, h->st);
      if (!rreg.empty()) x.recv_reg.upload(rreg, h->st);
      x.sendbuf.alloc((size_t)std::max<long long>(1, x.n_send) * BS);
      x.recvbuf.alloc((size_t)std::max<long long>(1, x.n_recv) * BS);
      GL_HIP(hipStreamSynchronize(h->st));
      x.ready = true;
    }
  }
  lap("grid choice, node -> cell map, reach");
  {   // children lists: the OWNED mesh nodes sorted by cell (stable radix sort: ascending node index inside a cell)
    dvec<uint32_t> k_in, k_out;
    dvec<int32_t> v_in;
    k_in.alloc((size_t)n);
    k_out.alloc((size_t)n);
    v_in.alloc((size_t)n);
    mg.cell_nodes.alloc((size_t)n);
    hipLaunchKernelGGL(k_mg_cell_keys, dim3(gridn(n)), dim3(256), 0, h->st, n, mg.cell0.p, k_in.p, v_in.p);
    GL_HIP(hipGetLastError());
    int bits = 1;
    while ((int64_t(1) << bits) <= g1.nn) ++bits;
    gl_sort_pairs_u32(h, k_in.p, k_out.p, v_in.p, mg.cell_nodes.p, (size_t)n, bits);
    mg.cell_ptr.alloc((size_t)g1.nn + 1);
    gl_offsets_of_sorted_keys(h, k_out.p, n, g1.nn, mg.cell_ptr.p);
    GL_HIP(hipStreamSynchronize(h->st));
  }
  lap("children lists (device sort)");
  {   // explicit restriction operator mesh -> grid (k_mg_restrict0): count, slice lengths on the host, fill
    dvec<int32_t> cnt;
    cnt.alloc((size_t)g1.nn);
    hipLaunchKernelGGL(k_mg_pt_count<D>, dim3(gridn(g1.nn)), dim3(256), 0, h->st, gdev(g1), mg.cell_ptr.p, mg.cell_nodes.p,
                       mg.wgt.p, cnt.p);
    GL_HIP(hipGetLastError());
    std::vector<int32_t> hc((size_t)g1.nn);
    GL_HIP(hipMemcpyAsync(hc.data(), cnt.p, hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->st));
    GL_HIP(hipStreamSynchronize(h->st));
    const int64_t ns = (g1.nn + GL_WAVE - 1) / GL_WAVE;
    std::vector<int64_t> ptr((size_t)ns + 1, 0);
    for (int64_t sl = 0; sl < ns; ++sl) {
      int m = 1;   // at least one (zero-weight) entry: the kernel clamps its slot index to len - 1
      for (int64_t I = sl * GL_WAVE; I < std::min<int64_t>(g1.nn, (sl + 1) * GL_WAVE); ++I) m = std::max(m, hc[I]);
      ptr[sl + 1] = ptr[sl] + (int64_t)m * GL_WAVE;
    }
    mg.pt_ptr.upload(ptr, h->st);
    mg.pt_idx.alloc((size_t)ptr[ns]);
    mg.pt_w.alloc((size_t)ptr[ns]);
    hipLaunchKernelGGL(k_mg_pt_fill<D>, dim3(gridn(ns * GL_WAVE)), dim3(256), 0, h->st, gdev(g1), mg.cell_ptr.p,
                       mg.cell_nodes.p, mg.wgt.p, mg.pt_ptr.p, mg.pt_idx.p, mg.pt_w.p);
    GL_HIP(hipGetLastError());
    GL_HIP(hipStreamSynchronize(h->st));
  }
  lap("restriction operator (count, fill)");
  const size_t nd0 = (size_t)h->n_nodes * BS;
  mg.x.alloc_zero(nd0, h->st);
  mg.x2.alloc_zero(nd0, h->st);
  mg.d.alloc_zero(nd0, h->st);
  mg.res.alloc_zero(nd0, h->st);
  // constrained-dof mask including the ghosts (their owners know): columns of constrained ghost dofs are eliminated
  // from this rank's part of the Galerkin products like every other constrained column
  dvec<uint8_t> fx_all;
  const uint8_t* fxr = fx;
  if (framed && fx) {
    dvec<double> fd;
    fd.alloc_zero(nd0, h->st);
    hipLaunchKernelGGL(k_mg_mask_to_double, dim3(gridn(n * BS)), dim3(256), 0, h->st, n * BS, fx, fd.p);
    gl_halo_exchange(h, fd.p, BS);
    fx_all.alloc(nd0);
    hipLaunchKernelGGL(k_mg_double_to_mask, dim3(gridn((long long)nd0)), dim3(256), 0, h->st, (int64_t)nd0, fd.p, fx_all.p);
    GL_HIP(hipGetLastError());
    GL_HIP(hipStreamSynchronize(h->st));
    fxr = fx_all.p;
  }

  // ---- Galerkin products ------------------------------------------------------------------------------------------
  auto new_level = [&](const MgGrid& g, const int* off, const int* ng, bool global, const BoxDev sb = BoxDev{0, 0, 0, 0, 0, 0, 0}) {
    MgLevel* L = new MgLevel();
    L->g = g;
    for (int a = 0; a < 3; ++a) {
      L->o[a] = off[a];
      L->ng[a] = ng[a];
    }
    L->global = global;
    const int sbv[6] = {sb.lo0, sb.lo1, sb.lo2, sb.n0, sb.n1, sb.n2};
    for (int a = 0; a < 6; ++a) L->sb[a] = sbv[a];
    L->sb_nn = sb.nn;
    L->A.alloc((size_t)mg.S * B2 * (sb.nn ? sb.nn : g.nn));
    L->dinv.alloc((size_t)B2 * g.nn);
    for (dvec<double>* v : {&L->x, &L->x2, &L->r, &L->d, &L->res}) v->alloc_zero((size_t)BS * g.nn, h->st);
    mg.lv.push_back(L);
    return L;
  };
  auto nn_of = [&](const int* ng) {
    int64_t v = 1;
    for (int a = 0; a < D; ++a) v *= ng[a];
    return v;
  };
  // replicate a level: sum of the ranks' partial operators (single precision planes summed in double)
  auto allreduce_operator = [&](MgLevel* L) {
    const size_t ne = (size_t)mg.S * B2 * L->g.nn;
    const size_t piece = std::min<size_t>(ne, (size_t)64 << 20);   // bounded staging buffer (512 MB of doubles at most)
    dvec<double> tmp;
    tmp.alloc(piece);
    for (size_t o = 0; o < ne; o += piece) {
      const size_t m = std::min(piece, ne - o);
      hipLaunchKernelGGL(k_mg_f2d, dim3(gridn((long long)m)), dim3(256), 0, h->st, (int64_t)m, L->A.p + o, tmp.p);
      gl_allreduce_bulk(h, tmp.p, m);
      hipLaunchKernelGGL(k_mg_d2f, dim3(gridn((long long)m)), dim3(256), 0, h->st, (int64_t)m, tmp.p, L->A.p + o);
    }
    GL_HIP(hipGetLastError());
    GL_HIP(hipStreamSynchronize(h->st));
  };
  const bool g1_global = framed;
  MgLevel* L1 = new_level(g1, o1, ng1, g1_global, sbox1);
  {   // A1 = P^T K P in two gathers (k_mg_kp, k_mg_ptkp); rows in chunks so that T = K P stays below ~8 GB
    const int E = 2 * mg.R + 2;
    const int64_t NB = D == 3 ? (int64_t)E * E * E : (int64_t)E * E;
    // this rank's partial operator (the rows of its own mesh nodes' parents): on its core box in a box-limited run
    BoxDev cb{0, 0, 0, 0, 0, 0, 0};
    if (mg.boxed) {
      long long cn = 1;
      for (int a = 0; a < 3; ++a) cn *= std::max(1, mg.core[3 + a] - mg.core[a] + 1);
      cb = BoxDev{mg.core[0], mg.core[1], mg.core[2], std::max(1, mg.core[3] - mg.core[0] + 1),
                  std::max(1, mg.core[4] - mg.core[1] + 1), std::max(1, mg.core[5] - mg.core[2] + 1), cn};
    }
    const long long rows = cb.nn ? cb.nn : g1.nn;
    const size_t ne = (size_t)mg.S * B2 * rows;
    dvec<double> A1d;
    A1d.alloc_zero(ne, h->st);
    const int64_t chunk = std::max<int64_t>(GL_WAVE, std::min<int64_t>(n, ((int64_t)8 << 30) / (NB * B2 * (int64_t)sizeof(float))));
    dvec<float> T;
    T.alloc((size_t)(std::min(chunk, n) * NB * B2));
    // (nothing is dropped on lattice meshes and on quality-controlled ones: no far couplings, the test costs nothing)
    const int lump_far = (h->opt.flags & GLIMS_FLAG_MG_NO_LUMPING) ? 0 : 1;
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
      const int64_t r1 = std::min(n, r0 + chunk);
      hipLaunchKernelGGL((k_mg_kp<D, BS>), dim3(gridn(r1 - r0, 4)), dim3(256), 0, h->st, gdev(g1), mg.R, r0, r1, n_all,
                         mg.cell0.p, mg.wgt.p, p.slice_ptr.p, p.cols.p, mg.op_vals, fxr, T.p, lump_far);
      hipLaunchKernelGGL((k_mg_ptkp<D, BS>), dim3(gridn(rows * mg.S)), dim3(256), 0, h->st, gdev(g1), mg.R, mg.S,
                         r0, r1, mg.pt_ptr.p, mg.pt_idx.p, mg.pt_w.p, mg.cell0.p, T.p, A1d.p, cb);
      GL_HIP(hipGetLastError());
    }
    if (mg.boxed) {
      // the rows of this rank's storage box = own partial rows + the neighbours' (box n their core), summed in rank order
      GridExchange& x = mg.gx;
      const int SB = mg.S * B2;
      dvec<double> sbuf, rbuf;
      dvec<int> my_core;
      sbuf.alloc((size_t)std::max<long long>(1, x.n_send) * SB);
      rbuf.alloc((size_t)std::max<long long>(1, x.n_recv) * SB);
      my_core.upload(std::vector<int>(mg.core, mg.core + 6), h->st);
      if (x.n_send > 0)
        hipLaunchKernelGGL(k_gx_pack_op, dim3(gridn(x.n_send * SB)), dim3(256), 0, h->st, gdev(g1), cb, SB, x.n_send_reg,
                           x.send_reg.p, x.n_send, A1d.p, sbuf.p);
      GL_HIP(hipGetLastError());
      gl_exchange(h, x.peers, x.send_ptr, x.recv_ptr, sbuf.p, rbuf.p, SB);
      hipLaunchKernelGGL(k_gx_sum_op, dim3(gridn(sbox1.nn * SB)), dim3(256), 0, h->st, gdev(g1), sbox1, cb, h->rank, my_core.p,
                         SB, x.n_recv_reg, x.recv_reg.p, rbuf.p, A1d.p, L1->A.p);
      GL_HIP(hipGetLastError());
      GL_HIP(hipStreamSynchronize(h->st));
    } else {
      hipLaunchKernelGGL(k_mg_d2f, dim3(gridn((long long)ne)), dim3(256), 0, h->st, (int64_t)ne, A1d.p, L1->A.p);
      GL_HIP(hipGetLastError());
      GL_HIP(hipStreamSynchronize(h->st));
    }
  }
  if (g1_global && !mg.boxed) allreduce_operator(L1);
  mg.entries = (int64_t)mg.S * B2 * (sbox1.nn ? sbox1.nn : g1.nn);   // per rank
  // coarsen until the GLOBAL grid is small enough (the level count is then the same on every rank)
  while (nn_of(mg.lv.back()->ng) > coarse_max) {
    MgLevel* Lf = mg.lv.back();
    MgGrid gc;
    gc.nn = 1;
    int ngc[3] = {1, 1, 1}, oc[3] = {0, 0, 0};
    bool any = false;
    for (int a = 0; a < D; ++a) {
      if (Lf->ng[a] > 2) {
        Lf->f[a] = 2;
        ngc[a] = Lf->ng[a] / 2 + 1;
        any = true;
      } else {
        Lf->f[a] = 1;
        ngc[a] = Lf->ng[a];
      }
    }
    if (!any) break;
    const bool glob = Lf->global;
    for (int a = 0; a < D; ++a) {
      if (glob) {
        oc[a] = 0;
        gc.n[a] = ngc[a];
      } else {   // box of the parents of this rank's fine box [o, o + n - 1]
        const int lo_g = Lf->o[a], hi_g = Lf->o[a] + Lf->g.n[a] - 1;
        oc[a] = Lf->f[a] == 2 ? lo_g >> 1 : lo_g;
        const int hc = Lf->f[a] == 2 ? (hi_g + 1) >> 1 : hi_g;
        gc.n[a] = std::min(hc, ngc[a] - 1) - oc[a] + 1;
      }
      gc.nn *= gc.n[a];
    }
    MgLevel* Lc = new_level(gc, oc, ngc, glob);
    Lf = mg.lv[mg.lv.size() - 2];
    Fac fc{{Lf->f[0], Lf->f[1], Lf->f[2]}, {Lf->o[0], Lf->o[1], Lf->o[2]}, {oc[0], oc[1], oc[2]}};
    // below a box-stored level every rank forms the product of the fine rows it OWNS (lowest rank whose core holds the node;
    // cores lie inside the storage boxes): a partition of the rows, summed over the ranks
    const bool masked = Lf->sb_nn != 0;
    hipLaunchKernelGGL((k_mg_rap<D, BS>), dim3(gridn((long long)gc.nn * mg.S)), dim3(256), 0, h->st, gdev(Lf->g), gdev(gc),
                       fc, mg.R, mg.S, Lf->A.p, Lc->A.p, sbox_of(*Lf), masked ? mg.cores.p : (const int*)nullptr, h->world,
                       h->rank);
    GL_HIP(hipGetLastError());
    if (glob && (!Lf->global || masked)) allreduce_operator(Lc);   // first replicated level: sum of the ranks' parts
    mg.entries += (int64_t)mg.S * B2 * gc.nn;
  }
  for (MgLevel* L : mg.lv) {
    hipLaunchKernelGGL((k_mg_dinv<D, BS>), dim3(gridn(L->sb_nn ? L->sb_nn : L->g.nn)), dim3(256), 0, h->st, gdev(L->g), mg.S,
                       L->A.p, L->dinv.p, sbox_of(*L));
    GL_HIP(hipGetLastError());
  }

  lap("Galerkin products, diagonal inverses");
  // ---- lambda_max(Dinv A) per smoothed level: power iteration -----------------------------------------------------
  // measured (tools/run_c5.py, degree 3): config C5 (lattice) 13.0 ms per solve with lmax / 30 against 16.6 with lmax / 10
  // and 18.7 with lmax / 4; 1 M-point Delaunay mesh 75 ms with lmax / 10 against 83 with lmax / 30
  mg.lattice = lattice;
  // default interval of the smoothers [lambda_max / ratio, lambda_max]: 30 on lattice meshes AND on general meshes of bounded
  // node spacing (no mesh edge spans more than two cells of the first grid: what a quality-controlled mesher emits -- the
  // brain-like workload at 1 M nodes: 27 iterations, 59.9 ms per solve with 30 against 31 / 68.5 with 10), 10 on meshes with
  // long edges / slivers (1 M random points: 75 ms per solve with 10 against 83 with 30)
  mg.default_ratio = (lattice || mg.dropped_fraction < 1e-4) ? 30.0 : 10.0;
  mg.cheb_ratio = h->opt.mg_cheb_ratio > 1.0 ? h->opt.mg_cheb_ratio : mg.default_ratio;
  mg.exact_level0 = framed;
  // (scalar hierarchy of a partitioned run: a float is half a double, the halo exchange moves whole doubles)
  mg.x32 = (h->opt.flags & GLIMS_FLAG_MG_FP64_VECTORS) == 0 && !(BS == 1 && framed);
  mg.half_smoother = (h->opt.flags & GLIMS_FLAG_MG_FP32_SMOOTHER) == 0;
  gl_make_smoother_copy(h, mg, mg.half_smoother, mg.exact_level0);
  mg.rs.alloc_zero(nd0, h->st);
  const int pit = 25;   // 12 under-estimate lambda_max on meshes with slivers (localised top modes); a pass costs 0.1 ms
  {
    const int64_t nd = n * BS;
    hipLaunchKernelGGL(k_mg_fill, dim3(gridn(nd)), dim3(256), 0, h->st, nd, mg.x.p, fx);
    double lam = 1.0;
    for (int it = 0; it < pit; ++it) {
      if (mg.exact_level0) gl_halo_exchange(h, mg.x.p, BS);
      gl_launch_mg_fine(h, mg, 2, mg.x.p, nullptr, nullptr, mg.x2.p, 0.0, 0.0);
      // = |Dinv A x| / |x| once x is normalised (it > 0); one global value in the distributed-exact mode, so that
      // every rank smooths with the same polynomial
      lam = std::sqrt(gl_dot(h, mg.x2.p, mg.x2.p, nd, mg.exact_level0));
      if (!(lam > 0.0) || !std::isfinite(lam)) break;
      hipLaunchKernelGGL(k_mg_scale, dim3(gridn(nd)), dim3(256), 0, h->st, nd, mg.x2.p, 1.0 / lam);
      std::swap(mg.x.p, mg.x2.p);
    }
    mg.lam0 = (std::isfinite(lam) && lam > 0.0) ? lam : 2.0;
    GL_HIP(hipMemsetAsync(mg.x.p, 0, nd0 * sizeof(double), h->st));
    GL_HIP(hipMemsetAsync(mg.x2.p, 0, nd0 * sizeof(double), h->st));
  }
  for (size_t l = 0; l + 1 < mg.lv.size(); ++l) {
    MgLevel& L = *mg.lv[l];
    const int64_t nd = L.g.nn * BS;
    hipLaunchKernelGGL(k_mg_fill, dim3(gridn(nd)), dim3(256), 0, h->st, nd, L.x.p, (const uint8_t*)nullptr);
    double lam = 1.0;
    for (int it = 0; it < pit; ++it) {
      if (L.sb_nn) {
        // box-stored operator: every rank applies the rows of its box to the (replicated) vector, keeps the rows it owns, and
        // the sum over the ranks is the whole product -- the single-rank iteration, evaluated in a distributed way
        mg_apply_cart<D, BS>(h, mg, L, mg.R, 2, L.x.p, nullptr, nullptr, L.x2.p, 0.0, 0.0, nullptr, nullptr, sbox_of(L));
        hipLaunchKernelGGL((k_mg_keep_owned<BS>), dim3(gridn(L.g.nn)), dim3(256), 0, h->st, gdev(L.g), mg.cores.p, h->world,
                           h->rank, L.x2.p, 0.0);
        GL_HIP(hipGetLastError());
        gl_allreduce_bulk(h, L.x2.p, (size_t)nd);
      } else {
        mg_apply_cart<D, BS>(h, mg, L, mg.R, 2, L.x.p, nullptr, nullptr, L.x2.p, 0.0, 0.0);
      }
      lam = std::sqrt(gl_dot(h, L.x2.p, L.x2.p, nd, false));
      if (!(lam > 0.0) || !std::isfinite(lam)) break;
      hipLaunchKernelGGL(k_mg_scale, dim3(gridn(nd)), dim3(256), 0, h->st, nd, L.x2.p, 1.0 / lam);
      std::swap(L.x.p, L.x2.p);
    }
    L.lam = (std::isfinite(lam) && lam > 0.0) ? lam : 2.0;
  }

  // The first grid carries 1 / 8 of the mesh's nodes but dense 27- (125-)point stencils of 3 x 3 blocks: 1 / 2 (general
  // meshes: more than 1 x) of level 0's operator bytes per pass.  It is smoothed like level 0: scaled variables, operator in
  // half precision (C5: k_mg_cart 29 -> see DESIGN.md section 7; the single-precision planes are released).
  if (mg.lv.size() >= 2 && mg.lv[0]->g.nn > 6000 && mg.half_smoother) {
    MgLevel& L = *mg.lv[0];
    const GridDev g = gdev(L.g);
    const BoxDev sb = sbox_of(L);
    const long long rows = sb.nn ? sb.nn : g.nn;
    L.sc.alloc_zero((size_t)BS * L.g.nn, h->st);
    L.A16.alloc((size_t)mg.S * B2 * rows);
    hipLaunchKernelGGL((k_mg_level_scale<D, BS>), dim3(gridn(rows)), dim3(256), 0, h->st, g, mg.S, L.A.p, L.sc.p, L.dinv.p, sb);
    if (sb.nn) {   // the half copy scales COLUMNS too: the factors of the nodes around the box come from their owners
      hipLaunchKernelGGL((k_mg_keep_owned<BS>), dim3(gridn(g.nn)), dim3(256), 0, h->st, g, mg.cores.p, h->world, h->rank,
                         L.sc.p, 1.0);
      GL_HIP(hipGetLastError());
      gl_allreduce_bulk(h, L.sc.p, (size_t)BS * g.nn);
    }
    hipLaunchKernelGGL((k_mg_half_copy<D, BS>), dim3(gridn(rows * mg.S)), dim3(256), 0, h->st, g, mg.R, mg.S, L.A.p,
                       L.sc.p, (_Float16*)L.A16.p, sb);
    GL_HIP(hipGetLastError());
    GL_HIP(hipStreamSynchronize(h->st));
    L.A.release();
    L.half = true;
  }
  lap("half copies, eigenvalue estimates");
  // ---- coarsest level: dense inverse ------------------------------------------------------------------------------
  {
    MgLevel& L = *mg.lv.back();
    const int64_t nn = L.g.nn;
    const int nc = (int)(nn * BS);
    GL_REQUIRE(nc <= 6000, "coarsest multigrid level too large for the dense solve (mg_coarse_nodes)");
    std::vector<float> Ah((size_t)mg.S * B2 * nn);
    GL_HIP(hipMemcpyAsync(Ah.data(), L.A.p, Ah.size() * sizeof(float), hipMemcpyDeviceToHost, h->st));
    GL_HIP(hipStreamSynchronize(h->st));
    std::vector<double> M((size_t)nc * nc, 0.0);
    const int W = 2 * mg.R + 1;
    for (int64_t I = 0; I < nn; ++I) {
      const int iv[3] = {(int)(I % L.g.n[0]), (int)((I / L.g.n[0]) % L.g.n[1]), (int)(I / ((int64_t)L.g.n[0] * L.g.n[1]))};
      for (int off = 0; off < mg.S; ++off) {
        const int o[3] = {off % W - mg.R, (off / W) % W - mg.R, D == 3 ? off / (W * W) - mg.R : 0};
        bool in = true;
        int jv[3] = {0, 0, 0};
        for (int a = 0; a < D; ++a) {
          jv[a] = iv[a] + o[a];
          in = in && jv[a] >= 0 && jv[a] < L.g.n[a];
        }
        if (!in) continue;
        const int64_t J = ((int64_t)jv[2] * L.g.n[1] + jv[1]) * L.g.n[0] + jv[0];
        for (int a = 0; a < BS; ++a)
          for (int b = 0; b < BS; ++b)
            M[(size_t)(a * nn + I) * nc + (b * nn + J)] = (double)Ah[((size_t)off * B2 + a * BS + b) * nn + I];
      }
    }
    for (int i = 0; i < nc; ++i)   // symmetrise (single-precision entries), identity on dofs without stiffness
      for (int j = i + 1; j < nc; ++j) M[(size_t)i * nc + j] = M[(size_t)j * nc + i] = 0.5 * (M[(size_t)i * nc + j] + M[(size_t)j * nc + i]);
    for (int i = 0; i < nc; ++i)
      if (!(M[(size_t)i * nc + i] > 0.0)) {
        for (int j = 0; j < nc; ++j) M[(size_t)i * nc + j] = M[(size_t)j * nc + i] = 0.0;
        M[(size_t)i * nc + i] = 1.0;
      }
    // a body without Dirichlet data has the rigid-body modes in the kernel of every level: the small relative shift
    // keeps the factorisation defined there and changes a regular operator by 1e-9
    double dmax = 0.0;
    for (int i = 0; i < nc; ++i) dmax = std::max(dmax, M[(size_t)i * nc + i]);
    dvec<double> col;
    dvec<int> bad;
    col.alloc((size_t)nc);
    bad.alloc(1);
    bool ok = false;
    for (double shift = 1e-9; shift < 1.0 && !ok; shift *= 100.0) {
      for (int i = 0; i < nc; ++i) M[(size_t)i * nc + i] += shift * dmax;
      mg.coarse_inv.upload(M, h->st);
      ok = dense_spd_inverse_device(h, mg.coarse_inv.p, nc, col.p, bad.p);
      for (int i = 0; i < nc; ++i) M[(size_t)i * nc + i] -= shift * dmax;
    }
    // (partitioned runs: this check and the size check above look at REPLICATED data -- the coarsest operator is the same
    //  on every rank -- so all ranks throw together and nobody is left waiting in a collective)
    GL_REQUIRE(ok, "multigrid: the coarsest operator is not positive definite");
  }
  lap("coarsest level: dense inverse");
  mg.ready = true;
  mg.n_levels = (int)mg.lv.size() + 1;
  mg.complexity = 1.0 + (double)mg.entries / ((double)p.total_entries * B2);
  mg.ms_setup = 1e3 * (omp_get_wtime() - t_start);
  if (getenv("GLIMS_VERBOSE")) {
    fprintf(stderr, "glims multigrid (%d dof / node): %s mesh, H = (%.4g, %.4g, %.4g), stencil radius %d (%.3f %% of the entries reach further and stay on level 0), levels:",
            BS, lattice ? "lattice" : "general", mg.H[0], mg.H[1], D == 3 ? mg.H[2] : 0.0, mg.R, 100.0 * mg.dropped_fraction);
    fprintf(stderr, " mesh(%lld nodes, lam %.2f)", (long long)n, mg.lam0);
    for (MgLevel* L : mg.lv)
      fprintf(stderr, " %dx%dx%d%s(lam %.2f)", L->g.n[0], L->g.n[1], L->g.n[2], L->global ? "[replicated]" : "", L->lam);
    fprintf(stderr, "; operator complexity %.2f; set-up %.1f ms\n", mg.complexity, mg.ms_setup);
    if (framed)
      fprintf(stderr, "glims multigrid, rank %d: level-0 passes %s (%d interior / %d boundary slices); first grid %s (work box %.0f %% of it)\n",
              h->rank, gl_mg_split_level0(h, mg) ? "split around the halo exchange" : "after the halo exchange",
              (int)p.n_interior, (int)p.n_boundary, mg.boxed ? "box-limited" : "whole on every rank", 100.0 * mg.box_fraction);
    if (framed && mg.boxed && mg.gx.ready)
      fprintf(stderr, "glims multigrid, rank %d: first-grid residual by neighbour exchange with %d peers, %.1f KB out / %.1f KB in per cycle (the all-reduce moves %.1f KB)\n",
              h->rank, (int)mg.gx.peers.size(), mg.gx.n_send * BS * 8e-3, mg.gx.n_recv * BS * 8e-3, mg.lv[0]->g.nn * BS * 8e-3);
  }
}

// one V-cycle on the Cartesian levels l.. : x_l = approx A_l^-1 r_l (result left in L.x)
// c2 of the first Chebyshev step of level l, if the restriction INTO level l may do that step itself (a smoothed
// level, and no all-reduce between the restriction and the step); 0 otherwise
inline double mg_fused_first_c2(const MgHierarchy& mg, size_t l, bool allreduce_before) {
  if (allreduce_before || l + 1 == mg.lv.size()) return 0.0;
  Cheb ch(mg.lv[l]->lam, mg.cheb_ratio);
  double c1, c2;
  ch.next(0, &c1, &c2);
  return c2;
}

// `first_done`: the restriction that produced L.r has taken the first smoothing step as well (L.d, L.x are set)
template <int D, int BS>
void mg_cycle_cart(glims_ctx* h, MgHierarchy& mg, int deg, size_t l, const int* done, bool first_done) {
  MgLevel& L = *mg.lv[l];
  const GridDev g = gdev(L.g);
  if (l + 1 == mg.lv.size()) {
    const int nc = (int)(L.g.nn * BS);
    hipLaunchKernelGGL(k_mg_dense, dim3((nc + 3) / 4), dim3(256), 0, h->st, nc, mg.coarse_inv.p, L.r.p, L.x.p, done);
    GL_HIP(hipGetLastError());
    return;
  }
  Cheb ch(L.lam, mg.cheb_ratio);
  double c1, c2;
  ch.next(0, &c1, &c2);
  const double* sc = L.half ? L.sc.p : nullptr;   // this level works in scaled variables between first_cart and its last pass
  // (first_done on a level with scaled variables: only the mesh -> grid restriction does that, and it applies the scaling)
  GL_REQUIRE(!(first_done && L.half && l > 0), "internal: fused first step on a level with scaled variables");
  // Replicated first grid of a partitioned run: this rank computes only its work box (mg_work_box) -- what it hands down
  // (the residual on its core, restricted under the owner mask and summed over the ranks) and up (the correction on its
  // core, interpolated to its own mesh nodes) is what the whole-grid sweep would have produced there.
  const bool boxed = l == 0 && mg.boxed;
  const BoxDev box = boxed ? mg_work_box<D>(mg, L.g, deg) : BoxDev{0, 0, 0, 0, 0, 0, 0};
  const long long cnt = box.nn ? box.nn : g.nn;
  if (!first_done)
    hipLaunchKernelGGL((k_mg_first_cart<D, BS>), dim3(gridn(cnt)), dim3(256), 0, h->st, g, L.dinv.p, L.r.p, L.d.p, L.x.p, c2, sc,
                       done, box);
  double *xa = L.x.p, *xb = L.x2.p;
  for (int m = 1; m < deg; ++m) {
    ch.next(m, &c1, &c2);
    mg_apply_cart<D, BS>(h, mg, L, mg.R, 1, xa, L.r.p, L.d.p, xb, c1, c2, done, nullptr, box);
    std::swap(xa, xb);
  }
  mg_apply_cart<D, BS>(h, mg, L, mg.R, 0, xa, L.r.p, nullptr, L.res.p, 0.0, 0.0, done, sc, box);
  MgLevel& C = *mg.lv[l + 1];
  const Fac fc{{L.f[0], L.f[1], L.f[2]}, {L.o[0], L.o[1], L.o[2]}, {C.o[0], C.o[1], C.o[2]}};
  const bool reduce_c = boxed || (C.global && !L.global && h->world > 1);
  const double c2c = mg_fused_first_c2(mg, l + 1, reduce_c);
  hipLaunchKernelGGL((k_mg_restrict<D, BS>), dim3(gridn(C.g.nn, 4)), dim3(256), 0, h->st, g, gdev(C.g), fc, L.res.p, C.r.p,
                     c2c != 0.0 ? C.dinv.p : nullptr, C.d.p, C.x.p, c2c, done, boxed ? mg.cores.p : nullptr, h->world,
                     h->rank);
  GL_HIP(hipGetLastError());
  // first replicated level of a partitioned run: every rank has restricted the residual of its own rows -> sum
  // (box-limited first grid: of the fine nodes it owns)
  if (reduce_c) gl_allreduce_bulk(h, C.r.p, (size_t)BS * C.g.nn);
  mg_cycle_cart<D, BS>(h, mg, deg, l + 1, done, c2c != 0.0);
  hipLaunchKernelGGL((k_mg_prolong<D, BS>), dim3(gridn(cnt)), dim3(256), 0, h->st, g, gdev(C.g), fc, C.x.p, xa, xb, sc, done, box);
  GL_HIP(hipGetLastError());
  std::swap(xa, xb);
  Cheb cp(L.lam, mg.cheb_ratio);
  for (int m = 0; m < deg; ++m) {
    cp.next(m, &c1, &c2);
    mg_apply_cart<D, BS>(h, mg, L, mg.R, 1, xa, L.r.p, L.d.p, xb, c1, c2, done, m == deg - 1 ? sc : nullptr, box);
    std::swap(xa, xb);
  }
  if (xa != L.x.p) std::swap(L.x.p, L.x2.p);   // the result is always handed up in L.x
}

}  // namespace
bool gl_mg_split_level0(const glims_ctx* h, const MgHierarchy& mg) {
  return mg.exact_level0 && h->world > 1 && h->n_peers > 0 && h->pat.n_interior > 0 && h->pat.n_boundary > 0;
}
namespace {

template <int D, int BS>
void mg_apply_t(glims_ctx* h, MgHierarchy& mg, int deg, const double* r, double* u, const int* done, double* pv) {
  const int64_t n = h->n_own;
  const uint8_t* fx = mg.op_fixed;
  const double* r_full = r;
  GL_REQUIRE(!(mg.lv[0]->sb_nn && deg > mg.gx.deg),
             "multigrid: the first grid's operator is stored for the work boxes of a lower smoother degree (rebuild the hierarchy)");
  Cheb ch(mg.lam0, mg.cheb_ratio);
  double c1, c2;
  ch.next(0, &c1, &c2);
  const bool x32 = mg.x32;
  const int xrec = gl_xrec_doubles(BS, x32);   // length of an iterate's node record in doubles (halo exchange)
  if (x32)
    hipLaunchKernelGGL((k_mg_first_fine<BS, float>), dim3(gridn(n)), dim3(256), 0, h->st, n, mg.dinv0.p, mg.sc.p, r,
                       (float*)mg.rs.p, (float*)mg.d.p, (float*)mg.x.p, c2, done);
  else
    hipLaunchKernelGGL((k_mg_first_fine<BS, double>), dim3(gridn(n)), dim3(256), 0, h->st, n, mg.dinv0.p, mg.sc.p, r,
                       mg.rs.p, mg.d.p, mg.x.p, c2, done);
  r = mg.rs.p;   // from here on the level-0 passes work in the scaled variables
  double *xa = mg.x.p, *xb = mg.x2.p;
  const bool ex = mg.exact_level0;   // the passes read ghost columns: bring them in (iterates are owned-row vectors)
  // ... and hide the exchange behind the slices that reference no ghost column, as the Krylov operator does
  const bool split = gl_mg_split_level0(h, mg);
  auto fine = [&](int mode, double* xin, double* dd, double* xout, double a1, double a2, double* uo, const double* rf,
                  double* pvv) {
    if (split) {
      gl_halo_start(h, xin, xrec);
      gl_launch_mg_fine(h, mg, mode, xin, r, dd, xout, a1, a2, done, uo, rf, pvv, 1);
      gl_halo_finish(h);
      gl_launch_mg_fine(h, mg, mode, xin, r, dd, xout, a1, a2, done, uo, rf, pvv, 2);
    } else {
      if (ex) gl_halo_exchange(h, xin, xrec);
      gl_launch_mg_fine(h, mg, mode, xin, r, dd, xout, a1, a2, done, uo, rf, pvv);
    }
  };
  for (int m = 1; m < deg; ++m) {
    ch.next(m, &c1, &c2);
    fine(1, xa, mg.d.p, xb, c1, c2, nullptr, nullptr, nullptr);
    std::swap(xa, xb);
  }
  fine(0, xa, nullptr, mg.res.p, 0.0, 0.0, nullptr, nullptr, nullptr);
  MgLevel& L1 = *mg.lv[0];
  const GridDev g1 = gdev(L1.g);
  // restriction through the explicit operator (a thread per grid node); the grid level's first smoothing step rides along
  // unless an all-reduce of the restricted residual comes first (replicated grid of a partitioned run) or the level is the
  // dense-solved one
  const bool reduce1 = L1.global && h->world > 1;
  const double c2_1 = mg_fused_first_c2(mg, 0, reduce1);
  const unsigned gr0 = (unsigned)(((g1.nn + 63) / 64 * 64 + 255) / 256);
  const double* sc1 = L1.half ? L1.sc.p : nullptr;
  if (x32)
    hipLaunchKernelGGL((k_mg_restrict0<BS, float>), dim3(gr0), dim3(256), 0, h->st, g1.nn, mg.pt_ptr.p, mg.pt_idx.p,
                       mg.pt_w.p, (const float*)mg.res.p, L1.r.p, c2_1 != 0.0 ? L1.dinv.p : nullptr, sc1, L1.d.p,
                       L1.x.p, c2_1, done);
  else
    hipLaunchKernelGGL((k_mg_restrict0<BS, double>), dim3(gr0), dim3(256), 0, h->st, g1.nn, mg.pt_ptr.p, mg.pt_idx.p,
                       mg.pt_w.p, (const double*)mg.res.p, L1.r.p, c2_1 != 0.0 ? L1.dinv.p : nullptr, sc1, L1.d.p,
                       L1.x.p, c2_1, done);
  GL_HIP(hipGetLastError());
  if (reduce1 && mg.boxed && mg.gx.ready && deg <= mg.gx.deg) {
    // box-limited cycle: this rank needs the sum on its work box only -> neighbour exchange of (box n core) regions
    GridExchange& x = mg.gx;
    if (x.n_send > 0)
      hipLaunchKernelGGL((k_gx_pack<BS>), dim3(gridn(x.n_send)), dim3(256), 0, h->st, g1, x.n_send_reg, x.send_reg.p, x.n_send,
                         L1.r.p, x.sendbuf.p, done);
    GL_HIP(hipGetLastError());
    gl_exchange(h, x.peers, x.send_ptr, x.recv_ptr, x.sendbuf.p, x.recvbuf.p, BS);
    const BoxDev box = mg_work_box<D>(mg, L1.g, deg);
    if (x.n_recv_reg > 0)
      hipLaunchKernelGGL((k_gx_sum<BS>), dim3(gridn(box.nn ? box.nn : g1.nn)), dim3(256), 0, h->st, g1, box, h->rank, x.n_recv_reg,
                         x.recv_reg.p, x.recvbuf.p, L1.r.p, done);
    GL_HIP(hipGetLastError());
  } else if (reduce1) {
    gl_allreduce_bulk(h, L1.r.p, (size_t)BS * L1.g.nn);
  }
  mg_cycle_cart<D, BS>(h, mg, deg, 0, done, c2_1 != 0.0);
  if (x32)
    hipLaunchKernelGGL((k_mg_prolong0<D, BS, float>), dim3(gridn(n)), dim3(256), 0, h->st, g1, n, mg.cell0.p, mg.wgt.p, fx,
                       mg.sc.p, L1.x.p, (const float*)xa, (float*)xb, done);
  else
    hipLaunchKernelGGL((k_mg_prolong0<D, BS, double>), dim3(gridn(n)), dim3(256), 0, h->st, g1, n, mg.cell0.p, mg.wgt.p, fx,
                       mg.sc.p, L1.x.p, (const double*)xa, xb, done);
  GL_HIP(hipGetLastError());
  std::swap(xa, xb);
  Cheb cp(mg.lam0, mg.cheb_ratio);
  for (int m = 0; m < deg; ++m) {
    cp.next(m, &c1, &c2);
    // the last step leaves the scaled variables and writes the preconditioned residual where the solver wants it -- and,
    // for the Krylov solver, the partial sums of (r, u) and (r, r) over its blocks (pv)
    const bool last = m == deg - 1;
    fine(1, xa, mg.d.p, xb, c1, c2, last ? u : nullptr, last ? r_full : nullptr, last ? pv : nullptr);
    std::swap(xa, xb);
  }
}

template <int D, int BS>
void mg_apply_cycle(glims_ctx* h, MgHierarchy& mg, int deg, const double* r, double* u, const int* done, double* pv) {
  // the smoothers' interval follows the options of the moment (no rebuild: the eigenvalue estimates do not depend on it)
  // (defaults, measured: elasticity 30 on lattice meshes / 10 on general ones with degree 3; the scalar RD hierarchy 10
  // with degree 1 -- tools/run_rd_precond.py)
  mg.cheb_ratio = h->opt.mg_cheb_ratio > 1.0 ? h->opt.mg_cheb_ratio : (mg.bs == 1 ? 10.0 : mg.default_ratio);
  mg.cycles++;
  // (Replaying the cycle -- or whole Krylov iterations -- from a captured hipGraph was built and measured in round 3: no
  //  gain at any size, 1.99 vs 1.88 ms per step on config C2, 4.37 vs 4.34 at 1 M rows, C5 12.15 vs 12.07: the idle time
  //  between dependent kernels is the write-back of the predecessor's dirty lines, not launch latency.)
  mg_apply_t<D, BS>(h, mg, deg, r, u, done, pv);
}

}  // namespace

void gl_mg_setup(glims_ctx* h, MgHierarchy& mg) {
  GL_REQUIRE(mg.bs == 1 || mg.bs == h->dim, "multigrid: block size must be 1 or the dimension");
  if (h->dim == 2) {
    if (mg.bs == 1) mg_setup_t<2, 1>(h, mg);
    else mg_setup_t<2, 2>(h, mg);
  } else {
    if (mg.bs == 1) mg_setup_t<3, 1>(h, mg);
    else mg_setup_t<3, 3>(h, mg);
  }
}

void gl_mg_apply(glims_ctx* h, MgHierarchy& mg, int degree, const double* r, double* u, const int* done, double* pv) {
  GL_REQUIRE(mg.ready, "multigrid hierarchy not built");
  const int deg = std::max(1, std::min(8, degree));
  if (h->dim == 2) {
    if (mg.bs == 1) mg_apply_cycle<2, 1>(h, mg, deg, r, u, done, pv);
    else mg_apply_cycle<2, 2>(h, mg, deg, r, u, done, pv);
  } else {
    if (mg.bs == 1) mg_apply_cycle<3, 1>(h, mg, deg, r, u, done, pv);
    else mg_apply_cycle<3, 3>(h, mg, deg, r, u, done, pv);
  }
}
