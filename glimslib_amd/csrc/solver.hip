// Time stepping on the device: Newton on the RD block, single-reduction (Chronopoulos-Gear) Jacobi-PCG for
// the linear solves, block-Jacobi PCG for the (linear, decoupled) mechanics block, RCCL halo exchange.
//
// Replaces, per time step, `self.solver.solve()` of the reference (simulation_base.py:302): DOLFIN residual and
// Jacobian assembly + PETSc SNES with a sparse LU of the monolithic (d+1)N system
// (simulation_tumor_growth.py:124-130).  The concentration block does not depend on the displacement
// (simulation_tumor_growth.py:115-120), so the monolithic Newton iteration and "Newton on c, then one linear
// solve for u" have the same fixed point; tests/ checks that claim against the monolithic oracle.
//
// Host/device protocol: the host never reads a scalar inside a Krylov iteration.  alpha/beta live in `scal`, a
// device-side `done` word turns the remaining enqueued kernels of a solve into no-ops.  Inside the Newton iteration the
// host does not even ask how a linear solve went: it enqueues the previous solve's iteration count + 2, goes on to the
// assembly sweep and receives {||R||, Krylov iterations, final rr, done} in one store sequence into pinned host memory
// (k_publish), on whose sequence number it spins -- one decision point per Newton iteration, no stream
// synchronisation, no copies.  Solves whose length cannot be predicted (first step, elasticity, projections) are
// polled once per `check_every` iterations through the same mailbox.
// In a partitioned run every rank enqueues the same kernel / collective sequence; the decision words are computed from
// all-reduced values (summed in rank order inside the final reduction block, see NodeMail) and are therefore
// identical on all ranks.
#include "glims_internal.h"

#include <omp.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>

namespace {

// block-level sum of two values -> pv[blockIdx.x*2 + {0,1}]   (256-thread blocks)
__device__ __forceinline__ void block_sum2(double a, double b, double* __restrict__ pv) {
  __shared__ double sm[4][2];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_down(a, o, 64);
    b += __shfl_down(b, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sm[threadIdx.x >> 6][0] = a;
    sm[threadIdx.x >> 6][1] = b;
  }
  __syncthreads();
  if (threadIdx.x < 2) pv[(size_t)blockIdx.x * 2 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

// u = Dinv r and the first (r.u, r.r) partials (p, s start undefined: the first update has beta = 0 and skips them); also clears the recurrence scalars and the decision word
// of the previous solve (two memset nodes on the stream cost ~20 us of idle device per solve)
template <int BS>
__global__ __launch_bounds__(256) void k_cg_init(int64_t n_own, const double* __restrict__ r,
                                                  const double* __restrict__ dinv, double* __restrict__ u,
                                                  double* __restrict__ p, double* __restrict__ s,
                                                  double* __restrict__ pv, double* __restrict__ scal,
                                                  int* __restrict__ done, const PackMap pm, int ext) {
  if (blockIdx.x == 0 && threadIdx.x < 2 * SC_COUNT + 2) scal[threadIdx.x] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *done = 0;
  if (ext) return;   // external preconditioner (multigrid): u and the (r.u, r.r) partials come out of the cycle's last kernel
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double pg = 0.0, pr = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_own; i += stride) {
    if constexpr (BS == 1) {
      const double ri = r[i], ui = dinv[i] * ri;
      u[i] = ui;           // p and s are not initialised: the first update (beta = 0) does not read them
      if (pm.ref) pack_row<1>(pm, i, &ui);
      pg += ri * ui;
      pr += ri * ri;
    } else {
      double rv[BS], uv[BS];
#pragma unroll
      for (int a = 0; a < BS; ++a) rv[a] = r[i * BS + a];
#pragma unroll
      for (int a = 0; a < BS; ++a) {
        double v = 0.0;
#pragma unroll
        for (int b = 0; b < BS; ++b) v += dinv[i * BS * BS + a * BS + b] * rv[b];
        u[i * BS + a] = v;
        uv[a] = v;
        pg += rv[a] * v;
        pr += rv[a] * rv[a];
      }
      if (pm.ref) pack_row<BS>(pm, i, uv);
    }
  }
  block_sum2(pg, pr, pv);
}

__device__ __forceinline__ double2 ntload2(const double2* p) {
  double2 v;
  v.x = __builtin_nontemporal_load(&p->x);
  v.y = __builtin_nontemporal_load(&p->y);
  return v;
}
__device__ __forceinline__ void ntstore2(double2* p, const double2 v) {
  __builtin_nontemporal_store(v.x, &p->x);
  __builtin_nontemporal_store(v.y, &p->y);
}

// Chronopoulos-Gear recurrence + vector update in one kernel.
//   red = (gamma = r.u, delta = w.u, rr = r.r), already global sums (gamma, rr from the previous launch of this kernel
//   or k_cg_init, delta from the SpMV).  Every thread derives alpha/beta from `red`
//   and the previous iteration's scalars (`prev`, read-only here); thread 0 of block 0 publishes the new scalars to
//   `cur` (ping-pong) and to `info` = {iterations done, last rr} for the host.  No separate scalar kernel.
//   p = u + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s;  u = Dinv r      (one pass over 7 vectors)
template <int BS>
__global__ __launch_bounds__(256) void k_cg_update(int64_t n_own, const double* __restrict__ red,
                                                    const double* __restrict__ prev, double* __restrict__ cur,
                                                    double* __restrict__ info, int* __restrict__ done, double tol2,
                                                    double* __restrict__ p, double* __restrict__ s,
                                                    double* __restrict__ x, double* __restrict__ r,
                                                    double* __restrict__ u, const double* __restrict__ w,
                                                    const double* __restrict__ dinv, double* __restrict__ pv, int nt,
                                                    const PackMap pm, int ext, double* __restrict__ hist) {
  if (*done) return;
  const double gamma = red[0], delta = red[1], rr = red[2];
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  if (lead) info[1] = rr;
  if (!(isfinite(gamma) && isfinite(delta) && isfinite(rr))) {
    if (lead) *done = 2;
    return;
  }
  if (rr <= tol2) {
    if (lead) *done = 1;
    return;
  }
  const double it = prev[SC_IT];
  const bool first = !(it > 0.0);
  double beta = 0.0, denom = delta;
  if (it > 0.0) {
    beta = gamma / prev[SC_GAMMA];
    denom = delta - beta * gamma / prev[SC_ALPHA];
  }
  if (!(denom > 0.0)) {   // operator not SPD on this Krylov space (or exact breakdown)
    if (lead) *done = 3;
    return;
  }
  const double alpha = gamma / denom;
  if (lead) {
    cur[SC_ALPHA] = alpha;
    cur[SC_BETA] = beta;
    cur[SC_GAMMA] = gamma;
    cur[SC_IT] = it + 1.0;
    info[0] = it + 1.0;
    if (hist && it < (double)GL_CG_HIST) {   // Lanczos coefficients of this solve (Ritz values -> interval of the dot-free solves)
      hist[2 * (int)it] = alpha;
      hist[2 * (int)it + 1] = beta;
    }
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double pg = 0.0, pr = 0.0;   // partials of the NEXT iteration's gamma = r.u and rr = r.r
  if (BS == 1 && !ext) {
    // 16 B per lane (two rows per thread): all seven streams are hipMalloc-aligned
    const int64_t n2 = n_own >> 1;
    double2* p2 = reinterpret_cast<double2*>(p);
    double2* s2 = reinterpret_cast<double2*>(s);
    double2* x2 = reinterpret_cast<double2*>(x);
    double2* r2 = reinterpret_cast<double2*>(r);
    double2* u2 = reinterpret_cast<double2*>(u);
    const double2* w2 = reinterpret_cast<const double2*>(w);
    const double2* d2 = reinterpret_cast<const double2*>(dinv);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
      // nt != 0: p, s, x, r, w, dinv are streamed once per iteration -> non-temporal, so that u -- the only vector the
      // following SpMV gathers from -- is what stays in L2 / Infinity Cache
      double2 uu, pp, ww, ss, rr2, dd, xx, pn, sn, rn, un;
      if (nt) {
        uu = ntload2(u2 + i); ww = ntload2(w2 + i);
        rr2 = ntload2(r2 + i); dd = ntload2(d2 + i); xx = ntload2(x2 + i);
      } else {
        uu = u2[i]; ww = w2[i]; rr2 = r2[i]; dd = d2[i]; xx = x2[i];
      }
      if (first) {          // beta = 0: p = u, s = w; the old p, s (uninitialised) are not read
        pp.x = pp.y = ss.x = ss.y = 0.0;
      } else if (nt) {
        pp = ntload2(p2 + i); ss = ntload2(s2 + i);
      } else {
        pp = p2[i]; ss = s2[i];
      }
      pn.x = uu.x + beta * pp.x;
      pn.y = uu.y + beta * pp.y;
      sn.x = ww.x + beta * ss.x;
      sn.y = ww.y + beta * ss.y;
      xx.x += alpha * pn.x;
      xx.y += alpha * pn.y;
      rn.x = rr2.x - alpha * sn.x;
      rn.y = rr2.y - alpha * sn.y;
      un.x = dd.x * rn.x;
      un.y = dd.y * rn.y;
      pg += rn.x * un.x + rn.y * un.y;
      pr += rn.x * rn.x + rn.y * rn.y;
      if (nt) {
        ntstore2(p2 + i, pn); ntstore2(s2 + i, sn); ntstore2(x2 + i, xx); ntstore2(r2 + i, rn);
      } else {
        p2[i] = pn; s2[i] = sn; x2[i] = xx; r2[i] = rn;
      }
      u2[i] = un;
      if (pm.ref) {
        pack_row<1>(pm, 2 * i, &un.x);
        pack_row<1>(pm, 2 * i + 1, &un.y);
      }
    }
    if ((n_own & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const int64_t i = n_own - 1;
      const double pi = u[i] + (first ? 0.0 : beta * p[i]);
      const double si = w[i] + (first ? 0.0 : beta * s[i]);
      p[i] = pi;
      s[i] = si;
      x[i] += alpha * pi;
      const double ri = r[i] - alpha * si;
      r[i] = ri;
      const double ui = dinv[i] * ri;
      u[i] = ui;
      if (pm.ref) pack_row<1>(pm, i, &ui);
      pg += ri * ui;
      pr += ri * ri;
    }
    block_sum2(pg, pr, pv);
    return;
  }
  if (ext) {   // external preconditioner: p, s, x, r only (u currently holds M^-1 of the old r; the cycle follows)
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_own * BS; j += stride) {
      const double pi = u[j] + (first ? 0.0 : beta * p[j]);
      const double si = w[j] + (first ? 0.0 : beta * s[j]);
      p[j] = pi;
      s[j] = si;
      x[j] += alpha * pi;
      const double rn = r[j] - alpha * si;
      r[j] = rn;
      pr += rn * rn;
    }
    block_sum2(0.0, pr, pv);   // |r|^2 of the new residual, for the early convergence check (k_early_done)
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_own; i += stride) {
    if constexpr (BS == 1) {
      // (handled above)
    } else {
      double rv[BS];
#pragma unroll
      for (int a = 0; a < BS; ++a) {
        const int64_t j = i * BS + a;
        const double pi = u[j] + (first ? 0.0 : beta * p[j]);
        const double si = w[j] + (first ? 0.0 : beta * s[j]);
        p[j] = pi;
        s[j] = si;
        x[j] += alpha * pi;
        rv[a] = r[j] - alpha * si;
        r[j] = rv[a];
      }
      double uv[BS];
#pragma unroll
      for (int a = 0; a < BS; ++a) {
        double v = 0.0;
#pragma unroll
        for (int b = 0; b < BS; ++b) v += dinv[i * BS * BS + a * BS + b] * rv[b];
        u[i * BS + a] = v;
        uv[a] = v;
        pg += rv[a] * v;
        pr += rv[a] * rv[a];
      }
      if (pm.ref) pack_row<BS>(pm, i, uv);
    }
  }
  block_sum2(pg, pr, pv);
}

// red[q] = sum_b partials[b*nq + q] in a fixed order -> bitwise reproducible.  Two stages: `nb1` blocks each sum a
// contiguous range into tmp[blk*nq + q], then one block sums tmp (a single block over ~1e5 partials took 98 us).
__global__ __launch_bounds__(256) void k_reduce_stage1(int n, int nq, int per_block,
                                                        const double* __restrict__ partials,
                                                        double* __restrict__ tmp, const int* __restrict__ done) {
  if (done && *done) return;
  __shared__ double sm[4];
  const int lo = blockIdx.x * per_block, hi = min(n, lo + per_block);
  for (int q = 0; q < nq; ++q) {
    double v = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) v += partials[(size_t)i * nq + q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tmp[(size_t)blockIdx.x * nq + q] = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
  }
}

// final stage: ONE block of 1024 threads, every thread keeps nq (<= 3) independent accumulators so that its loads
// are all in flight together; fixed summation order (thread-strided, then a fixed tree) -> reproducible.
__global__ __launch_bounds__(1024) void k_reduce(int n, int nq, const double* __restrict__ partials,
                                                  double* __restrict__ red, const int* __restrict__ done,
                                                  const NodeMail nm) {
  if (done && *done) return;
  __shared__ double sm[16][3];
  double v[3] = {0.0, 0.0, 0.0};
  const int total = n * nq;   // partials are [n][nq]: walk the flat array with a stride that is a multiple of nq
  const int stride = 1024 * nq;
  {
    // four independent strided chains per thread (the loop is latency-bound: 12 us for 16 k pairs with one chain)
    double a[4][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
    int i = threadIdx.x * nq;
    for (; i + 3 * stride < total; i += 4 * stride)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (q < nq) a[c][q] += partials[i + c * stride + q];
    for (; i < total; i += stride)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (q < nq) a[0][q] += partials[i + q];
#pragma unroll
    for (int q = 0; q < 3; ++q) v[q] = (a[0][q] + a[1][q]) + (a[2][q] + a[3][q]);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_down(v[q], o, 64);
  }
  if ((threadIdx.x & 63) == 0)
    for (int q = 0; q < 3; ++q) sm[threadIdx.x >> 6][q] = v[q];
  __syncthreads();
  if ((int)threadIdx.x < nq) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sm[w][threadIdx.x];
    red[threadIdx.x] = t;
  }
  if (nm.slots) node_allreduce(red, nq, nm);
}

// PCG reduction: red = ( sum pv[.][0] , sum ps[.] , sum pv[.][1] ) = (gamma, delta, rr), fixed order.
//   ps: one delta partial per SpMV block (up to ~40 k at 10 M rows: four independent chains per thread, the loop is
//   latency-bound), pv: one (gamma, rr) pair per block of the vector kernel.  In a partitioned run the same block
//   performs the node-local all-reduce (NodeMail).
__global__ __launch_bounds__(1024) void k_reduce_cg(int ns, const double* __restrict__ ps, int nv,
                                                     const double* __restrict__ pv, double* __restrict__ red,
                                                     const int* __restrict__ done, const NodeMail nm) {
  if (done && *done) return;
  __shared__ double sm[16][3];
  double v[3] = {0.0, 0.0, 0.0};
  {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int i = threadIdx.x;
    for (; i + 3072 < ns; i += 4096) {
      a0 += ps[i];
      a1 += ps[i + 1024];
      a2 += ps[i + 2048];
      a3 += ps[i + 3072];
    }
    for (; i < ns; i += 1024) a0 += ps[i];
    v[1] = (a0 + a1) + (a2 + a3);
  }
  {   // (up to ~40 k pairs when the multigrid cycle's last pass supplies them: two chains per value)
    double g0 = 0.0, g1 = 0.0, r0 = 0.0, r1 = 0.0;
    int i = threadIdx.x;
    for (; i + 1024 < nv; i += 2048) {
      g0 += pv[(size_t)i * 2];
      r0 += pv[(size_t)i * 2 + 1];
      g1 += pv[(size_t)(i + 1024) * 2];
      r1 += pv[(size_t)(i + 1024) * 2 + 1];
    }
    for (; i < nv; i += 1024) {
      g0 += pv[(size_t)i * 2];
      r0 += pv[(size_t)i * 2 + 1];
    }
    v[0] = g0 + g1;
    v[2] = r0 + r1;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_down(v[q], o, 64);
  }
  if ((threadIdx.x & 63) == 0)
    for (int q = 0; q < 3; ++q) sm[threadIdx.x >> 6][q] = v[q];
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sm[w][threadIdx.x];
    red[threadIdx.x] = t;
  }
  if (nm.slots) node_allreduce(red, 3, nm);
}

// Early convergence check.  The recurrence notices a converged residual one iteration late: the vector update of
// iteration k leaves the partial sums of |r_k|^2, and it is the update of iteration k + 1 that reads their total -- after
// another preconditioner application and another operator pass whose results nobody uses.  One such pair per solve: a
// V-cycle + SpMV of 1.15 ms on the 10 ms elasticity solve of config C5, a 0.36 ms SpMV on each of the 3.1 Newton solves of
// a C4 step (8 % of the step).  This kernel sums the partials right after the update (rr pairs at pv[i * 2 + 1]) and sets
// the decision word when the tolerance is met, so that everything enqueued behind it returns at once.  The host inserts
// it where convergence is expected (from the previous solve's count; after every update when each iteration carries a
// V-cycle).  `decide` = 0: partitioned run without the node mailbox -- the sum goes to red[3], an all-reduce follows,
// k_early_decide takes the decision.
__global__ __launch_bounds__(1024) void k_early_done(int nv, const double* __restrict__ pv, double tol2,
                                                      double* __restrict__ red, double* __restrict__ info,
                                                      int* __restrict__ done, int decide, const NodeMail nm) {
  if (*done) return;
  __shared__ double sm[16];
  double a0 = 0.0, a1 = 0.0;
  int i = threadIdx.x;
  for (; i + 1024 < nv; i += 2048) {
    a0 += pv[(size_t)i * 2 + 1];
    a1 += pv[(size_t)(i + 1024) * 2 + 1];
  }
  for (; i < nv; i += 1024) a0 += pv[(size_t)i * 2 + 1];
  double v = a0 + a1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sm[w];
    red[3] = t;
  }
  if (nm.slots) node_allreduce(red + 3, 1, nm);
  __syncthreads();
  if (decide && threadIdx.x == 0) {
    const double rr = red[3];
    if (isfinite(rr) && rr <= tol2) {
      info[1] = rr;
      *done = 1;
    }
  }
}
__global__ void k_early_decide(const double* __restrict__ red, double tol2, double* __restrict__ info,
                               int* __restrict__ done) {
  if (*done) return;
  const double rr = red[3];
  if (isfinite(rr) && rr <= tol2) {
    info[1] = rr;
    *done = 1;
  }
}

// Hands the host everything it decides on in ONE store sequence into pinned host memory:
//   mail = [seq | red[0..3] | Krylov info {iterations, rr} | done], seq written last (system-scope release).
// The host spins on seq instead of synchronising the stream and copying three scalars back (~25-30 us of idle
// device per decision point, measured on the 1 M-row configuration).
__global__ void k_publish(int n, const double* __restrict__ red, const double* __restrict__ info,
                          const int* __restrict__ done, const int* __restrict__ comm_err, double* mail,
                          unsigned long long seq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < n && i < 4; ++i) mail[1 + i] = red[i];
  if (info) {
    mail[5] = info[0];
    mail[6] = info[1];
  }
  mail[7] = done ? (double)*done : 0.0;
  mail[8] = comm_err ? (double)*comm_err : 0.0;
  __threadfence_system();
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(mail), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_pack(int64_t n, int bs, const int32_t* __restrict__ idx, const double* __restrict__ vec,
                       double* __restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * bs) return;
  const int64_t k = i / bs;
  const int a = (int)(i - k * bs);
  buf[i] = vec[(int64_t)idx[k] * bs + a];
}

__global__ void k_sub(int64_t n, double* __restrict__ y, const double* __restrict__ a, const double* __restrict__ b,
                      const uint8_t* __restrict__ fixed) {   // y = a - b, 0 on fixed dofs
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (fixed && fixed[i]) ? 0.0 : a[i] - b[i];
}

__global__ void k_mask_assign(int64_t n, double* __restrict__ y, const uint8_t* __restrict__ fixed,
                              const double* __restrict__ vals) {   // y[fixed] = vals ? vals[i] : 0
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && fixed[i]) y[i] = vals ? vals[i] : 0.0;
}

__global__ void k_extrapolate(int64_t n, double* __restrict__ c, double* __restrict__ c_old,
                              const uint8_t* __restrict__ fixed, int have_old) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double cur = c[i], old = c_old[i];
  c_old[i] = cur;
  if (have_old && !(fixed && fixed[i])) c[i] = 2.0 * cur - old;
}

// warm start of a step's first Newton solve: u = c - c_old (the previous step's increment), c_old = c
// The step's predicted increment, extrapolated linearly in time from the last two: d1 = c - c_old is the increment of the step
// just finished, du holds the one before it (second_order = 0 while only one is known): u = 2 d1 - du.  Measured against
// u = d1 (interleaved runs, profiles/r04_warm_start_order_ab.txt): C4 12.7 -> 11.8 PCG iterations per step, 10.2-10.7 -> 9.9
// ms; C3 17.95 -> 16.55, 1.60 -> 1.53 ms; brain-like mesh 29.65 -> 28.65.  A quadratic extrapolation (three increments) buys
// C3 another 1.6 iterations and costs C4 0.1 ms: not taken.
__global__ void k_ws_delta(int64_t n, const double* __restrict__ c, double* __restrict__ c_old, double* __restrict__ u,
                           double* __restrict__ du, int second_order) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double ci = c[i];
  const double d1 = ci - c_old[i];
  const double ui = second_order ? 2.0 * d1 - du[i] : d1;
  u[i] = ui;
  du[i] = d1;
  c_old[i] = ci;
}
// Guess of a step's second linear solve from the second corrections of the two previous steps: u = 2 d - d_prev (or d alone when
// only one is known); d_prev = d.
__global__ void k_d2_guess(int64_t n, const double* __restrict__ d, double* __restrict__ d_prev, double* __restrict__ u,
                           int second_order) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = d[i];
  u[i] = second_order ? 2.0 * a - d_prev[i] : a;
  d_prev[i] = a;
}
// Is a guess u better than none?  Partial sums of |r - A u|^2 and |r|^2 (w = A u), grid-stride, two per block.
__global__ __launch_bounds__(256) void k_guess_norms(int64_t n, const double* __restrict__ r, const double* __restrict__ w,
                                                      double* __restrict__ pv) {
  double a = 0.0, b = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double ri = r[i], t = ri - w[i];
    a += t * t;
    b += ri * ri;
  }
  block_sum2(a, b, pv);
}
// r -= A u (w = A u), c += u -- if the guess brings the residual down to a fifth (red = global sums of k_guess_norms), else nothing
__global__ void k_ws_apply_if(int64_t n, double* __restrict__ r, const double* __restrict__ w, double* __restrict__ c,
                              const double* __restrict__ u, const double* __restrict__ red) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !(red[0] < 0.04 * red[1])) return;   // (|r - A u| < 0.2 |r|: less than that is gone after PCG's first iterations anyway)
  r[i] -= w[i];
  c[i] += u[i];
}
// d = c - d (d held the state before a solve: now the correction the solve added)
__global__ void k_d2_from_state(int64_t n, const double* __restrict__ c, double* __restrict__ d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = c[i] - d[i];
}
// r -= A u (w = A u), c += u
__global__ void k_ws_apply(int64_t n, double* __restrict__ r, const double* __restrict__ w, double* __restrict__ c,
                           const double* __restrict__ u) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  r[i] -= w[i];
  c[i] += u[i];
}

__global__ __launch_bounds__(256) void k_dot_partials(int64_t n, const double* __restrict__ a,
                                                       const double* __restrict__ b, double* __restrict__ partials) {
  __shared__ double sm[4];
  double v = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) v += a[i] * b[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// up to three dot products against one vector in a single pass: partials[b][q] = (a_q, y) over block b
struct Dot3 {
  const double* a[3];
};
__global__ __launch_bounds__(256) void k_dot3(int64_t n, int nq, const Dot3 v, const double* __restrict__ y,
                                               double* __restrict__ partials) {
  __shared__ double sm[4][3];
  double acc[3] = {0.0, 0.0, 0.0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double yi = y[i];
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (q < nq) acc[q] += v.a[q][i] * yi;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o, 64);
  }
  if ((threadIdx.x & 63) == 0)
    for (int q = 0; q < 3; ++q) sm[threadIdx.x >> 6][q] = acc[q];
  __syncthreads();
  if ((int)threadIdx.x < nq)
    partials[(size_t)blockIdx.x * nq + threadIdx.x] =
        sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

// y = sum_k a[k] x_k  (k < m <= 16)
struct LinComb {
  const double* x[16];
  double a[16];
};
__global__ void k_lincomb(int64_t n, int m, const LinComb lc, double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  for (int k = 0; k < m; ++k) v += lc.a[k] * lc.x[k][i];
  y[i] = v;
}

// inverse of the (Dirichlet-modified) diagonal BS x BS blocks of K_el (BS = 1: of a scalar operator plane)
template <int BS>
__global__ void k_block_dinv(int64_t n_own, const int64_t* __restrict__ slice_ptr, const uint8_t* __restrict__ diag_k,
                             const double* __restrict__ vKel, const uint8_t* __restrict__ fixed,
                             double* __restrict__ dinv) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_own) return;
  const int64_t s = row >> 6;
  const int lane = (int)(row & 63);
  const double* v = vKel + (slice_ptr[s] + (int64_t)diag_k[row] * GL_WAVE) * (BS * BS) + lane;
  double A[BS][BS];
#pragma unroll
  for (int a = 0; a < BS; ++a)
#pragma unroll
    for (int b = 0; b < BS; ++b) {
      double x = v[(a * BS + b) * GL_WAVE];
      if (fixed && (fixed[row * BS + a] || fixed[row * BS + b])) x = (a == b) ? 1.0 : 0.0;
      A[a][b] = x;
    }
  double* o = dinv + row * BS * BS;
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

inline unsigned grid_for(int64_t n, int bs = 256, int64_t cap = 256 * 16) {
  int64_t g = (n + bs - 1) / bs;
  if (g < 1) g = 1;
  if (cap > 0 && g > cap) g = cap;
  return (unsigned)g;
}
inline unsigned grid_exact(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

// ===================================================================================================
// halo exchange (RCCL grouped send/recv on the communication stream)
// ===================================================================================================
// `prepacked`: the kernel that produced `vec` has written the send buffer already (PackMap)
static void halo_start(glims_ctx* h, double* vec, int bs, bool prepacked = false) {
  if (h->world <= 1 || h->n_peers == 0) return;
  h->stats.halo_exchanges++;
  h->stats.halo_bytes += h->n_send * bs * (int64_t)sizeof(double);
  if (h->n_send > 0 && !prepacked) {
    hipLaunchKernelGGL(k_pack, dim3(grid_exact(h->n_send * bs)), dim3(256), 0, h->st, h->n_send, bs,
                       h->send_idx.p, vec, h->sendbuf.p);
    GL_HIP(hipGetLastError());
  }
  if (h->tr_halo) {
    const int rc = h->tr_halo(h->tr_user, h->sendbuf.p, h->send_ptr.data(), vec + h->n_own * bs, h->recv_ptr.data(),
                              h->n_peers, h->peer_rank.data(), bs, (void*)h->st);
    if (rc != 0) throw glims_error(GLIMS_E_RCCL, "transport halo callback failed (" + std::to_string(rc) + ")");
    return;
  }
  GL_HIP(hipEventRecord(h->ev_pack, h->st));
  GL_HIP(hipStreamWaitEvent(h->st_comm, h->ev_pack, 0));
  GL_REQUIRE(h->comm_halo, "world > 1 but no communicator: call glims_comm_init or glims_set_transport");
  hipEvent_t* ce = h->comm_pair();   // glims_options.time_kernels: the exchange's duration on the communication stream
  if (ce) GL_HIP(hipEventRecord(ce[0], h->st_comm));
  GL_NCCL(ncclGroupStart());
  for (int p = 0; p < h->n_peers; ++p) {
    const int64_t ns = h->send_ptr[p + 1] - h->send_ptr[p], nr = h->recv_ptr[p + 1] - h->recv_ptr[p];
    if (ns > 0)
      GL_NCCL(ncclSend(h->sendbuf.p + h->send_ptr[p] * bs, (size_t)ns * bs, ncclDouble, h->peer_rank[p],
                       h->comm_halo, h->st_comm));
    if (nr > 0)
      GL_NCCL(ncclRecv(vec + (h->n_own + h->recv_ptr[p]) * bs, (size_t)nr * bs, ncclDouble, h->peer_rank[p],
                       h->comm_halo, h->st_comm));
  }
  GL_NCCL(ncclGroupEnd());
  if (ce) GL_HIP(hipEventRecord(ce[1], h->st_comm));
  GL_HIP(hipEventRecord(h->ev_halo, h->st_comm));
}
static void halo_finish(glims_ctx* h) {
  if (h->world <= 1 || h->n_peers == 0 || h->tr_halo) return;
  // ... and how long the compute stream really waits for it (what the interior slices did not hide): an event before the
  // wait and one after it
  hipEvent_t* we = h->wait_pair();
  if (we) GL_HIP(hipEventRecord(we[0], h->st));
  GL_HIP(hipStreamWaitEvent(h->st, h->ev_halo, 0));
  if (we) GL_HIP(hipEventRecord(we[1], h->st));
}
void gl_halo_exchange(glims_ctx* h, double* vec, int bs) {
  halo_start(h, vec, bs);
  halo_finish(h);
}
void gl_exchange(glims_ctx* h, const std::vector<int32_t>& peers, const std::vector<int64_t>& send_ptr,
                 const std::vector<int64_t>& recv_ptr, const double* sendbuf, double* recvbuf, int bs) {
  const int np = (int)peers.size();
  if (h->world <= 1 || np == 0) return;
  h->stats.halo_exchanges++;
  h->stats.halo_bytes += send_ptr[np] * bs * (int64_t)sizeof(double);
  if (h->tr_halo) {
    const int rc = h->tr_halo(h->tr_user, sendbuf, send_ptr.data(), recvbuf, recv_ptr.data(), np, peers.data(), bs, (void*)h->st);
    if (rc != 0) throw glims_error(GLIMS_E_RCCL, "transport halo callback failed (" + std::to_string(rc) + ")");
    return;
  }
  GL_REQUIRE(h->comm_halo, "world > 1 but no communicator: call glims_comm_init or glims_set_transport");
  GL_HIP(hipEventRecord(h->ev_pack, h->st));
  GL_HIP(hipStreamWaitEvent(h->st_comm, h->ev_pack, 0));
  GL_NCCL(ncclGroupStart());
  for (int p = 0; p < np; ++p) {
    const int64_t ns = send_ptr[p + 1] - send_ptr[p], nr = recv_ptr[p + 1] - recv_ptr[p];
    if (ns > 0)
      GL_NCCL(ncclSend(sendbuf + send_ptr[p] * bs, (size_t)ns * bs, ncclDouble, peers[p], h->comm_halo, h->st_comm));
    if (nr > 0)
      GL_NCCL(ncclRecv(recvbuf + recv_ptr[p] * bs, (size_t)nr * bs, ncclDouble, peers[p], h->comm_halo, h->st_comm));
  }
  GL_NCCL(ncclGroupEnd());
  GL_HIP(hipEventRecord(h->ev_halo, h->st_comm));
  GL_HIP(hipStreamWaitEvent(h->st, h->ev_halo, 0));
}
void gl_halo_start(glims_ctx* h, double* vec, int bs) { halo_start(h, vec, bs); }
void gl_halo_finish(glims_ctx* h) { halo_finish(h); }
// Sum over ranks of the values reduce_partials / k_reduce_cg just left in `dev` (= h->red).  With the node mailbox
// the final reduction block has already done it.
static void allreduce_sum(glims_ctx* h, double* dev, int n) {
  if (h->world <= 1) return;
  h->stats.allreduces++;
  h->stats.reduce_transport = h->nm.slots ? 1 : h->tr_allreduce ? 3 : 2;
  if (h->nm.slots) return;
  if (h->tr_allreduce) {
    const int rc = h->tr_allreduce(h->tr_user, dev, n, (void*)h->st);
    if (rc != 0) throw glims_error(GLIMS_E_RCCL, "transport allreduce callback failed (" + std::to_string(rc) + ")");
    return;
  }
  GL_REQUIRE(h->comm_red, "world > 1 but no communicator: call glims_comm_init or glims_set_transport");
  GL_NCCL(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, h->comm_red, h->st));
}
// In-place sum over ranks of a device vector of any length (multigrid set-up and its per-cycle coarse residual):
// always RCCL or the host transport -- the node mailbox carries a handful of scalars only.
void gl_allreduce_bulk(glims_ctx* h, double* dev, size_t n) {
  if (h->world <= 1 || n == 0) return;
  if (h->tr_allreduce) {
    for (size_t o = 0; o < n; o += (size_t)1 << 28) {   // the callback counts in int
      const int rc = h->tr_allreduce(h->tr_user, dev + o, (int)std::min<size_t>(n - o, (size_t)1 << 28), (void*)h->st);
      if (rc != 0) throw glims_error(GLIMS_E_RCCL, "transport allreduce callback failed (" + std::to_string(rc) + ")");
    }
    return;
  }
  GL_REQUIRE(h->comm_red, "world > 1 but no communicator: call glims_comm_init or glims_set_transport");
  GL_NCCL(ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, h->comm_red, h->st));
}
void gl_comm_destroy(glims_ctx* h) {
  if (h->comm_halo) (void)ncclCommDestroy(h->comm_halo);
  if (h->comm_red) (void)ncclCommDestroy(h->comm_red);
  h->comm_halo = h->comm_red = nullptr;
}

static void reduce_partials(glims_ctx* h, int n, int nq, const int* done) {
  if (n > 16384) {
    const int per_block = 1024, nb1 = (n + per_block - 1) / per_block;
    hipLaunchKernelGGL(k_reduce_stage1, dim3(nb1), dim3(256), 0, h->st, n, nq, per_block, h->partials.p,
                       h->partials2.p, done);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, nb1, nq, h->partials2.p, h->red.p, done, h->nm);
  } else {
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, n, nq, h->partials.p, h->red.p, done, h->nm);
  }
  GL_HIP(hipGetLastError());
}

// Publishes red[0..n) (+ the Krylov info and decision word) to the pinned mailbox and waits for it.
struct Mail {
  double red[4];
  double info[2];
  int done;
};
static Mail fetch(glims_ctx* h, int n_red, bool with_krylov) {
  const unsigned long long seq = ++h->mail_seq;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, h->st, n_red, h->red.p,
                     with_krylov ? h->scal.p + 2 * SC_COUNT : (const double*)nullptr,
                     with_krylov ? h->done.p : (const int*)nullptr, (const int*)h->nm.err, h->mail_dev, seq);
  GL_HIP(hipGetLastError());
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(h->h_pinned);
  for (long spins = 0; *flag != seq; ++spins) {
    if ((spins & 0xfff) == 0xfff) {
      const hipError_t q = hipStreamQuery(h->st);
      if (q == hipSuccess) {
        if (*flag == seq) break;
        throw glims_error(GLIMS_E_HIP, "stream drained without publishing the decision word");
      }
      if (q != hipErrorNotReady) throw glims_error(GLIMS_E_HIP, std::string("HIP error while waiting: ") + hipGetErrorString(q));
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  if (h->h_pinned[8] != 0.0)
    throw glims_error(GLIMS_E_RCCL, "node mailbox all-reduce timed out: a peer rank did not arrive within 60 s");
  Mail m;
  for (int i = 0; i < 4; ++i) m.red[i] = h->h_pinned[1 + i];
  m.info[0] = h->h_pinned[5];
  m.info[1] = h->h_pinned[6];
  m.done = (int)h->h_pinned[7];
  return m;
}
static void poll(glims_ctx* h, int* done, double* info2) {
  const Mail m = fetch(h, 0, true);
  info2[0] = m.info[0];
  info2[1] = m.info[1];
  *done = m.done;
}
static void read_red(glims_ctx* h, int n, double* out) {
  const Mail m = fetch(h, n, false);
  for (int i = 0; i < n; ++i) out[i] = m.red[i];
}
static double read_red0(glims_ctx* h) {
  double v;
  read_red(h, 1, &v);
  return v;
}

double gl_dot(glims_ctx* h, const double* a, const double* b, int64_t n, bool global) {
  const unsigned gd = grid_for(n, 256, 1024);
  hipLaunchKernelGGL(k_dot_partials, dim3(gd), dim3(256), 0, h->st, n, a, b, h->partials.p);
  if (global) {
    reduce_partials(h, (int)gd, 1, nullptr);
    allreduce_sum(h, h->red.p, 1);
  } else {   // rank-local value: no node-mailbox / RCCL step in the final block
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, (int)gd, 1, h->partials.p, h->red.p, (const int*)nullptr,
                       NodeMail());
    GL_HIP(hipGetLastError());
  }
  return read_red0(h);
}

// inverse diagonal blocks of the constrained K_el (block-Jacobi preconditioner; level-0 smoother of the multigrid)
void gl_block_dinv(glims_ctx* h) {
  const DevPattern& p = h->pat;
  const uint8_t* fx = h->have_fixed_u ? h->fixed_u.p : nullptr;
  if (h->dim == 2)
    hipLaunchKernelGGL(k_block_dinv<2>, dim3(grid_exact(h->n_own)), dim3(256), 0, h->st, h->n_own, p.slice_ptr.p,
                       p.diag_k.p, h->vKel.p, fx, h->m_dinv.p);
  else
    hipLaunchKernelGGL(k_block_dinv<3>, dim3(grid_exact(h->n_own)), dim3(256), 0, h->st, h->n_own, p.slice_ptr.p,
                       p.diag_k.p, h->vKel.p, fx, h->m_dinv.p);
  GL_HIP(hipGetLastError());
}

// ---- the two multigrid hierarchies (mg.hip) -------------------------------------------------------------------------
// elasticity: K_el with dim x dim blocks, Dirichlet dofs of the displacement eliminated
void gl_mg_setup_mech(glims_ctx* h) {
  GL_REQUIRE(h->vKel.n != 0, "multigrid set-up before the elasticity operator was assembled");
  gl_block_dinv(h);
  MgHierarchy& mg = h->mg;
  mg.bs = h->dim;
  mg.op_vals = h->vKel.p;
  mg.op_fixed = h->have_fixed_u ? h->fixed_u.p : nullptr;
  mg.op_dinv = h->m_dinv.p;
  gl_mg_setup(h, mg);
  h->stats.mg_levels = mg.n_levels;
  h->stats.mg_complexity = mg.complexity;
  h->stats.ms_mg_setup = mg.ms_setup;
  if (!mg.lv.empty()) {
    const MgLevel& L = *mg.lv[0];
    h->stats.mg_grid1_bytes = (int64_t)(L.half ? L.A16.n * sizeof(uint16_t) : L.A.n * sizeof(float));
  }
  h->stats.mg_box_fraction = mg.box_fraction;
}

// RD block: the hierarchy is built ONCE per glims_setup on the static part S = (1 - dt rho) M + dt K_D of the Newton
// Jacobian A(c) = S + 2 dt N(c).  N(c) is a mass-like matrix weighted with rho c: 0 <= 2 dt N(c) <= 2 dt rho c_max M, while
// S >= (1 - dt rho) M, so the spectrum of S^-1 A(c) lies in [1, 1 + 2 dt rho c_max / (1 - dt rho)] (= [1, 1.22] for the
// reference's dt rho = 0.1 at c = 1): a V-cycle for S preconditions every A(c) of the run as well as it preconditions S.
// The Krylov operator is the assembled fp64 A(c) as before.
void gl_mg_setup_rd(glims_ctx* h) {
  GL_REQUIRE(h->vS.n != 0, "multigrid set-up before the RD operators were assembled");
  MgHierarchy& mg = h->mg_rd;
  const uint8_t* fx = h->have_fixed_c ? h->fixed_c.p : nullptr;
  mg.own_dinv.alloc((size_t)h->n_own);
  hipLaunchKernelGGL(k_block_dinv<1>, dim3(grid_exact(h->n_own)), dim3(256), 0, h->st, h->n_own, h->pat.slice_ptr.p,
                     h->pat.diag_k.p, h->vS.p, fx, mg.own_dinv.p);
  GL_HIP(hipGetLastError());
  mg.bs = 1;
  mg.op_vals = h->vS.p;
  mg.op_fixed = fx;
  mg.op_dinv = mg.own_dinv.p;
  gl_mg_setup(h, mg);
  h->stats.rd_mg_levels = mg.n_levels;
  h->stats.rd_mg_complexity = mg.complexity;
  h->stats.ms_rd_mg_setup = mg.ms_setup;
}

// sum over the owned rows of S_ii / M_ii  (and the row count) -> partials of k_reduce, [blocks][2]
__global__ __launch_bounds__(256) void k_diag_ratio(int64_t n_own, const int64_t* __restrict__ slice_ptr,
                                                     const uint8_t* __restrict__ diag_k, const double* __restrict__ vS,
                                                     const double* __restrict__ vM, double lattice,
                                                     double* __restrict__ pv /*[blocks][3]*/) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double q = 0.0, cnt = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n_own; row += stride) {
    const int64_t e = slice_ptr[row >> 6] + (int64_t)diag_k[row] * GL_WAVE + (row & 63);
    q += vS[e] / vM[e];
    cnt += 1.0;
  }
  __shared__ double sm[4][2];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    q += __shfl_down(q, o, 64);
    cnt += __shfl_down(cnt, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sm[threadIdx.x >> 6][0] = q;
    sm[threadIdx.x >> 6][1] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double qs = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0], cs = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
    pv[(size_t)blockIdx.x * 3 + 0] = qs;
    pv[(size_t)blockIdx.x * 3 + 1] = cs;
    pv[(size_t)blockIdx.x * 3 + 2] = lattice * cs;   // rows that sit on a lattice (all or none of this rank's)
  }
}

// Which preconditioner the RD solves use.  The reference's sparse LU (simulation_tumor_growth.py:126-130) does not care
// how stiff a step is; Jacobi-PCG does: with q = mean_i S_ii / M_ii ~ 1 + C dt D / h^2 its iteration count per Newton
// solve grows like sqrt(2 q) (measured: 4-5 on the brain-extent configs C3 / C4 with q = 1.0-1.1, 66 / 129 on the unit
// cube with D = 0.1, dt = 1 at n = 32 / 64).  A multigrid-preconditioned iteration costs as much as ~5 Jacobi ones where
// the kernels are bandwidth-bound (>= ~0.5 M rows) and more where they are launch-bound, and needs 3-4 of them per solve:
// `auto` takes the hierarchy when the predicted Jacobi count exceeds that.  The decision uses all-reduced numbers, so
// every rank of a partitioned run takes the same one.
void gl_rd_choose_precond(glims_ctx* h) {
  const unsigned g = grid_for(h->n_own, 256, 1024);
  hipLaunchKernelGGL(k_diag_ratio, dim3(g), dim3(256), 0, h->st, h->n_own, h->pat.slice_ptr.p, h->pat.diag_k.p,
                     h->vS.p, h->vM.p, h->mm.lattice ? 1.0 : 0.0, h->partials.p);
  GL_HIP(hipGetLastError());
  reduce_partials(h, (int)g, 3, nullptr);
  allreduce_sum(h, h->red.p, 3);
  double out[3];
  read_red(h, 3, out);
  const double rows = std::max(1.0, out[1]);
  h->rd_stiffness_ratio = out[0] / rows;
  const double jacobi_its = std::sqrt(2.0 * std::max(0.0, h->rd_stiffness_ratio));
  // Break-even Jacobi count per Newton solve (tools/run_rd_precond.py).  Lattice meshes: the cycle needs 3-4 iterations
  // and costs ~5 (10 M rows) to ~20 (0.1 M rows, launch-bound) Jacobi iterations.  General meshes (measured on Delaunay
  // tetrahedralisations of random points, slivers included): 40 iterations per solve with degree-3 smoothers at 6.5 Jacobi
  // iterations each -- four times the lattice figure; the prediction sqrt(2 q) is also poor there (85 measured where it
  // says 14), which the observed count corrects after the first step (gl_step).
  double break_even = rows >= 4.0e6 ? 20.0 : rows >= 4.0e5 ? 45.0 : rows >= 5.0e4 ? 75.0 : 120.0;
  if (out[2] != out[1]) break_even *= 4.0;   // (a partitioned mesh is a lattice if every rank's part is)
  h->rd_break_even = break_even;
  int pick = h->opt.rd_precond;
  if (pick == GLIMS_RD_PRECOND_AUTO)
    pick = jacobi_its > break_even ? GLIMS_RD_PRECOND_MULTIGRID : GLIMS_RD_PRECOND_JACOBI;
  h->rd_precond_active = pick;
  h->stats.rd_precond_used = pick;
  h->stats.rd_stiffness_ratio = h->rd_stiffness_ratio;
  if (getenv("GLIMS_VERBOSE"))
    fprintf(stderr, "glims RD preconditioner: mean S_ii / M_ii = %.4g over %.0f rows -> predicted Jacobi-PCG iterations per "
            "solve %.0f (break-even %.0f): %s\n", h->rd_stiffness_ratio, rows, jacobi_its, break_even,
            pick == GLIMS_RD_PRECOND_MULTIGRID ? "multigrid V-cycle" : "Jacobi");
}

// c = Dirichlet value on the constrained nodes (owned and ghost alike: every rank lists the constrained nodes of its
// whole sub-mesh), then the state-dependent caches are stale
void gl_apply_dirichlet_c(glims_ctx* h) {
  h->dirichlet_c_dirty = false;
  if (!h->have_fixed_c || !h->fixed_c_val.p || !h->have_state) return;
  hipLaunchKernelGGL(k_mask_assign, dim3(grid_exact(h->n_nodes)), dim3(256), 0, h->st, h->n_nodes, h->c.p,
                     h->fixed_c.p, (const double*)h->fixed_c_val.p);
  GL_HIP(hipGetLastError());
  h->have_c_old = false;
  h->have_d2 = false;
}

// ===================================================================================================
// Dot-free RD linear solves: Chebyshev semi-iteration on the interval the run's right-hand sides excite
// ===================================================================================================
// Start of a solve from a zero guess: the first iterate needs no operator pass -- y_1 = d_1 = Dinv b / theta (+ its halo payload).
__global__ __launch_bounds__(256) void k_cheb_start(int64_t n_own, const double* __restrict__ b, const double* __restrict__ dinv,
                                                     double inv_theta, double* __restrict__ y, double* __restrict__ d,
                                                     const PackMap pm) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_own; i += stride) {
    const double d0 = inv_theta * dinv[i] * b[i];
    y[i] = d0;
    d[i] = d0;
    if (pm.ref) pack_row<1>(pm, i, &d0);
  }
}
// Iteration count of a warm-started solve, on the device, from the residual |t| of the guess (pass 1 has left its square in red[0]):
// the smallest m with 1 / T_m(sigma) <= tol / |t|, i.e. m = ceil(acosh(|t| / tol) / acosh(sigma)), at least m_min, at most m_max;
// m_done if the guess alone meets the tolerance.  Left in plan[0] for the launches and in the Krylov info slot for the host's
// statistics (travels with the next decision mail).
__global__ void k_cheb_plan(const double* __restrict__ red, double tol2, double inv_acosh_sigma, int m_min, int m_max,
                            int m_done, int* __restrict__ plan, double* __restrict__ info) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double rr = red[0];
  int m = m_max;
  if (isfinite(rr)) {
    // (m_done: the count when the tolerance is met already -- 0, or 2 for a solve whose first pass has run: the pass that adds
    //  the correction to the iterate is still to come)
    if (rr <= tol2) m = m_done;
    else {
      const double q = sqrt(rr / tol2);
      m = (int)ceil(log(q + sqrt(q * q - 1.0)) * inv_acosh_sigma);
      m = max(m_min, min(m_max, m));
    }
  }
  plan[0] = m;
  info[0] = (double)m;
  info[1] = rr;
}
// x -= delta (a Chebyshev solve taken back)
__global__ void k_sub_inplace(int64_t n, double* __restrict__ x, const double* __restrict__ d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] -= d[i];
}

// ===================================================================================================
// single-reduction PCG.  On entry: r = residual of the current x (owned rows), x = current iterate.
// ===================================================================================================
struct CgVecs {
  double *x, *r, *u, *w, *p, *s;
  const double* dinv;
  const double* vals;       // scalar operator plane; nullptr -> K_el blocks
  const uint8_t* fixed;
  int bs;
  bool k32 = false;         // block operator: stream the single-precision copy of K_el
  const float* vals32 = nullptr;   // scalar operator in single precision (optional; streamed instead of vals)
  MgHierarchy* mg = nullptr;   // u = V-cycle(r) of this hierarchy instead of the (block-)Jacobi scaling
  int mg_degree = 3;           // Chebyshev degree of its smoothers
  double* hist = nullptr;      // (alpha_k, beta_k) of the first GL_CG_HIST iterations are recorded here (device)
};

// w = A u with the fused dot product: one delta partial per SpMV block, interior launch first (slots [0, nbi)), then the
// boundary launch (slots [nbi, nbi + nbb))
static void apply_with_halo(glims_ctx* h, const CgVecs& v) {
  const DevPattern& p = h->pat;
  const bool split = h->world > 1 && h->n_peers > 0;
  if (!split) {
    if (v.vals) {
      hipEvent_t* ev = h->timing(glims_ctx::TK_SPMV) ? h->pair(glims_ctx::TK_SPMV) : nullptr;
      gl_launch_spmv(h, h->st, p.n_slices, nullptr, v.vals, v.u, v.w, v.fixed, nullptr, v.r, h->partials.p, 0,
                     h->done.p, v.vals32, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr);
    } else {
      const bool timed = h->timing(glims_ctx::TK_SPMVB);
      if (timed) h->tick(glims_ctx::TK_SPMVB);
      gl_launch_spmv_block(h, h->st, p.n_slices, nullptr, v.u, v.w, v.fixed, v.r, h->partials.p, 0, h->done.p, v.k32);
      if (timed) h->tick(glims_ctx::TK_SPMVB);
    }
    return;
  }
  // interior slices overlap with the xGMI transfer; boundary slices run once the ghosts have landed
  halo_start(h, v.u, v.bs, /*prepacked=*/v.mg == nullptr);
  const int nbi = p.n_interior > 0 ? gl_spmv_grid(p.n_interior) : 0;   // partial-sum slots of the interior launch
  if (v.vals) {
    // (glims_options.time_kernels: the interior launch carries the event pair -- the part of the operator pass that hides
    //  the exchange)
    hipEvent_t* ev = h->timing(glims_ctx::TK_SPMV) ? h->pair(glims_ctx::TK_SPMV) : nullptr;
    gl_launch_spmv(h, h->st, p.n_interior, p.interior_slices.p, v.vals, v.u, v.w, v.fixed, nullptr, v.r,
                   h->partials.p, 0, h->done.p, v.vals32, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr);
  } else
    gl_launch_spmv_block(h, h->st, p.n_interior, p.interior_slices.p, v.u, v.w, v.fixed, v.r, h->partials.p, 0,
                         h->done.p, v.k32);
  halo_finish(h);
  if (v.vals)
    gl_launch_spmv(h, h->st, p.n_boundary, p.boundary_slices.p, v.vals, v.u, v.w, v.fixed, nullptr, v.r,
                   h->partials.p, nbi, h->done.p, v.vals32);
  else
    gl_launch_spmv_block(h, h->st, p.n_boundary, p.boundary_slices.p, v.u, v.w, v.fixed, v.r, h->partials.p, nbi,
                         h->done.p, v.k32);
}

// `hint`: expected iteration count (0 = unknown).  The first batch is hint + 1 launches (the extra one only detects
// convergence), later batches are short; every launch after the device-side `done` is a ~2 us no-op.
// `defer` (needs a hint): enqueue hint + 2 iterations and return WITHOUT asking the device how it went
// (*its_out = -1); the caller reads {iterations, rr, done} together with its next decision (fetch(.., true)).
// Only valid where an unconverged linear solve is harmless, i.e. inside the inexact Newton iteration, whose own
// residual test decides.
static int cg_solve(glims_ctx* h, const CgVecs& v, double tol_abs, int maxit, int hint, int64_t* its_out,
                    double* res_out, bool defer = false) {
  const DevPattern& p = h->pat;
  const bool split = h->world > 1 && h->n_peers > 0;
  // delta partials: one per block of the SpMV launch(es)
  const int nblocks = split ? (p.n_interior > 0 ? gl_spmv_grid(p.n_interior) : 0) +
                                  (p.n_boundary > 0 ? gl_spmv_grid(p.n_boundary) : 0)
                            : gl_spmv_grid(p.n_slices);
  const int64_t n = h->n_own;
  // scal = [ping | pong | info{its, rr}]
  const unsigned g = grid_for(n);
  const double tol2 = tol_abs * tol_abs;
#define GL_VEC(K, ...)                                                                  \
  do {                                                                                  \
    if (v.bs == 1) hipLaunchKernelGGL(K<1>, dim3(g), dim3(256), 0, h->st, __VA_ARGS__); \
    else if (v.bs == 2) hipLaunchKernelGGL(K<2>, dim3(g), dim3(256), 0, h->st, __VA_ARGS__); \
    else hipLaunchKernelGGL(K<3>, dim3(g), dim3(256), 0, h->st, __VA_ARGS__);           \
    GL_HIP(hipGetLastError());                                                          \
  } while (0)
  PackMap pm;
  if (split && h->n_send > 0) {
    pm.ref = h->send_ref.p;
    pm.ptr = h->send_slot_ptr.p;
    pm.slot = h->send_slot.p;
    pm.sendbuf = h->sendbuf.p;
  }
  const int ext = v.mg ? 1 : 0;
  if (ext) pm = PackMap();   // the cycle's last kernel produces u: the halo payload is packed by k_pack
  // external preconditioner: the cycle's last kernel writes u and the (r.u, r.r) partial pairs of ITS blocks
  const int nvp = ext ? gl_mg_fine_blocks(h, gl_mg_split_level0(h, *v.mg)) : (int)g;
  auto precondition = [&]() {
    if (!ext) return;
    gl_mg_apply(h, *v.mg, v.mg_degree, v.r, v.u, h->done.p, h->partials_v.p);
  };
  GL_VEC(k_cg_init, n, v.r, v.dinv, v.u, v.p, v.s, h->partials_v.p, h->scal.p, h->done.p, pm, ext);
  precondition();
  int done = 0;
  double info[2] = {0.0, 0.0};
  const int batch = h->opt.check_every > 0 ? h->opt.check_every : 8;
  int enq = 0;
  double* info_dev = h->scal.p + 2 * SC_COUNT;
  // one Krylov iteration: operator, reduction, recurrence + vector update, [early convergence check], M^-1
  auto iteration = [&](int parity, bool early) {
    const double* prev = h->scal.p + parity * SC_COUNT;
    double* cur = h->scal.p + (parity ^ 1) * SC_COUNT;
    apply_with_halo(h, v);
    // gamma / rr partials: one pair per block of the previous vector kernel; delta: one per SpMV block
    hipLaunchKernelGGL(k_reduce_cg, dim3(1), dim3(1024), 0, h->st, nblocks, h->partials.p, nvp, h->partials_v.p,
                       h->red.p, h->done.p, h->nm);
    allreduce_sum(h, h->red.p, 3);
    const bool timed_upd = v.vals && h->timing(glims_ctx::TK_UPDATE);
    if (timed_upd) h->tick(glims_ctx::TK_UPDATE);
    // (with an external preconditioner the update's |r|^2 pairs go to a buffer of their own: partials_v is rewritten by
    //  the cycle's last kernel)
    GL_VEC(k_cg_update, n, h->red.p, prev, cur, info_dev, h->done.p, tol2, v.p, v.s, v.x, v.r, v.u, v.w, v.dinv,
           ext ? h->partials_rr.p : h->partials_v.p, /*nt=*/0, pm, ext, v.hist);
    if (timed_upd) h->tick(glims_ctx::TK_UPDATE);
    if (early) {
      const bool decide = h->world <= 1 || h->nm.slots != nullptr;
      hipLaunchKernelGGL(k_early_done, dim3(1), dim3(1024), 0, h->st, (int)g, ext ? h->partials_rr.p : h->partials_v.p,
                         tol2, h->red.p, info_dev, h->done.p, decide ? 1 : 0, h->nm);
      if (!decide) {
        allreduce_sum(h, h->red.p + 3, 1);
        hipLaunchKernelGGL(k_early_decide, dim3(1), dim3(1), 0, h->st, h->red.p, tol2, info_dev, h->done.p);
      }
      GL_HIP(hipGetLastError());
    }
    precondition();
  };
  while (enq < maxit + 1) {
    int want = batch;
    // First batch = the previous solve's count: convergence is noticed by the early check right after the last update,
    // not one iteration later.  With the Jacobi scaling one spare iteration (three no-op launches if unused) makes a miss
    // unlikely (two spares: +1.5 % step time at 1 M rows, same counts); with a V-cycle per iteration a spare one would cost
    // its ~25 no-op launches, so there is none -- a solve that comes one iteration short is polled on (elasticity) or is a
    // slightly weaker Newton step (deferred RD solves, whose own residual test decides)
    if (hint > 0) want = enq == 0 ? hint + (ext ? 0 : 1) : std::max(2, std::min(batch, hint / 4 + 1));
    const int nb = std::min(want, maxit + 1 - enq);
    // early check: after every update when an iteration carries a V-cycle or the count is unknown, otherwise from the
    // iteration before the expected last one on
    for (int j = 0; j < nb; ++j) iteration((enq + j) & 1, ext || hint <= 0 || enq + j + 2 >= hint);
    enq += nb;
    if (defer && hint > 0) {
      *its_out = -1;
      *res_out = 0.0;
      return GLIMS_OK;
    }
    poll(h, &done, info);
    if (done) break;
  }
#undef GL_VEC
  *its_out = (int64_t)info[0];
  *res_out = std::sqrt(info[1]);
  if (done == 1) return GLIMS_OK;
  if (done == 2) return GLIMS_NAN;
  return GLIMS_NOT_CONVERGED;
}

// ---- the dot-free solve -------------------------------------------------------------------------------------------------
// Safety factors on the chosen interval.  A lower end set too low only costs iterations (prototype: 0.5 x -> +50 %); an upper
// end set too low makes the iteration diverge on what lies above it, so that end gets more room.
static const double GL_CHEB_LO = 0.9, GL_CHEB_HI = 1.06;
static const double GL_CHEB_LOOSE = 3e-5;   // reductions down to this use the interval of the loose learning solves
#define GL_CHEB_HIST GL_CG_HIST
// Chebyshev recurrence on [a, b]: d_0 = z_0 / theta;  d_k = rho_k rho_(k-1) d_(k-1) + (2 rho_k / delta) z_k,  z = Dinv r
struct ChebRec {
  double theta, delta, sigma, rho;
  ChebRec(double a, double b) : theta(0.5 * (a + b)), delta(0.5 * (b - a)), sigma(theta / delta), rho(delta / theta) {}
  void next(double* c1, double* c2) {
    const double rn = 1.0 / (2.0 * sigma - rho);
    *c1 = rn * rho;
    *c2 = 2.0 * rn / delta;
    rho = rn;
  }
  double inv_acosh_sigma() const { return 1.0 / std::log(sigma + std::sqrt(sigma * sigma - 1.0)); }
  // operator-pass count for a residual reduction by `red` (< 1): smallest m with 1 / T_m(sigma) <= red
  int iterations(double red) const {
    if (!(red < 1.0)) return 0;
    const double q = 1.0 / std::max(red, 1e-300);
    return (int)std::ceil(std::log(q + std::sqrt(q * q - 1.0)) * inv_acosh_sigma());
  }
};

// The interval for the dot-free solves from one recorded PCG solve (alpha_k, beta_k; beta_0 = 0; m <= GL_CHEB_HIST iterations,
// residual reduction eps).  The Lanczos matrix T of the solve gives the spectral measure of ITS right-hand side in Dinv A:
// Ritz values theta_i with weights w_i = (first component of T's i-th eigenvector)^2 (Gauss quadrature: for any polynomial q
// of degree < 2m, |q(Dinv A) r|^2 / |r|^2 = sum_i w_i q(theta_i)^2).  The extreme Ritz values converge to the ends of the TRUE
// spectrum however little of the right-hand side lives there (brain-like mesh at 1 M nodes: [0.033, 3.63], kappa = 110, from
// components far below the tolerance; the bulk sits in [0.6, 2.1]) -- a Chebyshev iteration sized for that interval needs 4x
// PCG's passes.  So the interval is CHOSEN: starting from all Ritz values, the end points move inwards as long as the degree
// k that brings sqrt(sum_i w_i (T_k(x_i) / T_k(sigma))^2) below eps gets smaller -- Ritz values left outside are weighed with
// the polynomial's growth there, not ignored.
static double cheb_poly_ratio(int k, double x, double sigma) {   // |T_k(x)| / T_k(sigma), sigma > 1
  const double as = std::log(sigma + std::sqrt(sigma * sigma - 1.0));
  const double ax = std::fabs(x);
  if (ax <= 1.0) return std::fabs(std::cos(k * std::acos(ax))) / std::cosh(k * as);
  const double ao = std::log(ax + std::sqrt(ax * ax - 1.0));
  return std::exp(k * (ao - as)) * (1.0 + std::exp(-2.0 * k * ao)) / (1.0 + std::exp(-2.0 * k * as));
}
static int cheb_degree_for(const std::vector<double>& th, const std::vector<double>& w, double a, double b, double eps) {
  if (!(b > a) || !(a > 0.0)) return GL_CHEB_HIST * 4;
  const double theta = 0.5 * (a + b), delta = 0.5 * (b - a), sigma = theta / delta;
  for (int k = 1; k <= 4 * GL_CHEB_HIST; ++k) {
    double s2 = 0.0;
    for (size_t i = 0; i < th.size(); ++i) {
      const double q = cheb_poly_ratio(k, (theta - th[i]) / delta, sigma);
      s2 += w[i] * q * q;
    }
    if (std::sqrt(s2) <= eps) return k;
  }
  return 4 * GL_CHEB_HIST;
}
static bool ritz_interval(const double* hist, int m, double eps, double* lo, double* hi, double* full_lo, double* full_hi) {
  if (m < 2) return false;
  // T (dense, m <= 64) and its eigen-decomposition by cyclic Jacobi rotations
  std::vector<double> T((size_t)m * m, 0.0), V((size_t)m * m, 0.0);
  for (int k = 0; k < m; ++k) {
    const double al = hist[2 * k], be = hist[2 * k + 1];
    if (!(al > 0.0) || !std::isfinite(al) || !(be >= 0.0) || !std::isfinite(be)) return false;
    T[(size_t)k * m + k] = 1.0 / al + (k > 0 ? be / hist[2 * (k - 1)] : 0.0);
    if (k > 0) T[(size_t)k * m + k - 1] = T[(size_t)(k - 1) * m + k] = std::sqrt(be) / hist[2 * (k - 1)];
    V[(size_t)k * m + k] = 1.0;
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < m; ++p)
      for (int q = p + 1; q < m; ++q) off += T[(size_t)p * m + q] * T[(size_t)p * m + q];
    if (off < 1e-28) break;
    for (int p = 0; p < m; ++p)
      for (int q = p + 1; q < m; ++q) {
        const double apq = T[(size_t)p * m + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double app = T[(size_t)p * m + p], aqq = T[(size_t)q * m + q];
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = t * c;
        for (int k = 0; k < m; ++k) {
          const double akp = T[(size_t)k * m + p], akq = T[(size_t)k * m + q];
          T[(size_t)k * m + p] = c * akp - sn * akq;
          T[(size_t)k * m + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < m; ++k) {
          const double apk = T[(size_t)p * m + k], aqk = T[(size_t)q * m + k];
          T[(size_t)p * m + k] = c * apk - sn * aqk;
          T[(size_t)q * m + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < m; ++k) {
          const double vkp = V[(size_t)k * m + p], vkq = V[(size_t)k * m + q];
          V[(size_t)k * m + p] = c * vkp - sn * vkq;
          V[(size_t)k * m + q] = sn * vkp + c * vkq;
        }
      }
  }
  std::vector<int> ord((size_t)m);
  for (int i = 0; i < m; ++i) ord[(size_t)i] = i;
  std::sort(ord.begin(), ord.end(), [&](int x, int y) { return T[(size_t)x * m + x] < T[(size_t)y * m + y]; });
  std::vector<double> th((size_t)m), w((size_t)m);
  for (int i = 0; i < m; ++i) {
    th[(size_t)i] = T[(size_t)ord[(size_t)i] * m + ord[(size_t)i]];
    w[(size_t)i] = V[(size_t)0 * m + ord[(size_t)i]] * V[(size_t)0 * m + ord[(size_t)i]];   // first component of eigenvector i
  }
  if (!(th[0] > 0.0) || !std::isfinite(th[(size_t)m - 1])) return false;
  *full_lo = th[0];
  *full_hi = th[(size_t)m - 1];
  eps = std::min(0.5, std::max(1e-12, eps));
  int ia = 0, ib = m - 1;
  auto deg = [&](int a, int b) { return cheb_degree_for(th, w, GL_CHEB_LO * th[(size_t)a], GL_CHEB_HI * th[(size_t)b], eps); };
  int best = deg(ia, ib);
  for (bool moved = true; moved && ib - ia >= 2;) {
    moved = false;
    const int da = deg(ia + 1, ib);
    if (da < best) {
      best = da;
      ++ia;
      moved = true;
    }
    if (ib - ia < 2) break;
    const int db = deg(ia, ib - 1);
    if (db < best) {
      best = db;
      --ib;
      moved = true;
    }
  }
  *lo = th[(size_t)ia];
  *hi = th[(size_t)ib];
  return *lo > 0.0 && *hi > *lo;
}

// A finished PCG solve of `its` iterations (residual reduction eps) contributes its interval to the one being learnt
static void cheb_learn(glims_ctx* h, int64_t its, double eps) {
  const int m = (int)std::min<int64_t>(its, GL_CG_HIST);
  if (m < 2 || !h->cg_hist.p) return;
  double hist[2 * GL_CG_HIST];
  GL_HIP(hipMemcpyAsync(hist, h->cg_hist.p, (size_t)2 * m * sizeof(double), hipMemcpyDeviceToHost, h->st));
  GL_HIP(hipStreamSynchronize(h->st));
  double lo, hi, flo, fhi;
  if (!ritz_interval(hist, m, eps, &lo, &hi, &flo, &fhi)) return;
  if (getenv("GLIMS_VERBOSE"))
    fprintf(stderr, "glims dot-free solves: PCG solve of %lld iterations (reduction %.1e): Ritz values in [%.4f, %.4f], interval chosen "
            "[%.4f, %.4f]\n", (long long)its, eps, flo, fhi, lo, hi);
  glims_ctx::ChebState& cs = h->cheb;
  cs.acc_lmin = cs.learned ? std::min(cs.acc_lmin, lo) : lo;
  cs.acc_lmax = cs.learned ? std::max(cs.acc_lmax, hi) : hi;
  cs.learned++;
  if (eps >= GL_CHEB_LOOSE) {
    cs.acc_lmin0 = cs.learned0 ? std::min(cs.acc_lmin0, lo) : lo;
    cs.acc_lmax0 = cs.learned0 ? std::max(cs.acc_lmax0, hi) : hi;
    cs.learned0++;
  }
  h->stats.cheb_learn_solves++;
}
// the interval a solve that wants the reduction `red` uses
static void cheb_interval(const glims_ctx* h, double red, double* a, double* b) {
  const glims_ctx::ChebState& cs = h->cheb;
  const bool loose = red >= GL_CHEB_LOOSE && cs.lmax0 > 0.0;
  *a = GL_CHEB_LO * (loose ? cs.lmin0 : cs.lmin);
  *b = GL_CHEB_HI * (loose ? cs.lmax0 : cs.lmax) * h->cheb_test_hi;
}

// Cost of one Chebyshev pass relative to one PCG iteration (operator pass + reduction + vector update): algorithmic bytes of
// the two plus a per-launch overhead worth B0 bytes at the streaming rate (launch gap + write-back of the predecessor's dirty
// lines: ~5-8 us), from the rows and entries of the average rank -- the same number on every rank.
static double cheb_cost_ratio(glims_ctx* h) {
  double v[2] = {(double)h->n_own, (double)h->nnz};
  if (h->world > 1) {   // one collective per glims_setup (end of the first learning step)
    GL_HIP(hipMemcpyAsync(h->partials.p, v, sizeof(v), hipMemcpyHostToDevice, h->st));
    reduce_partials(h, 1, 2, nullptr);
    allreduce_sum(h, h->red.p, 2);
    read_red(h, 2, v);
    v[0] /= h->world;
    v[1] /= h->world;
  }
  const double B0 = 40e6;
  return (12.0 * v[1] + 60.0 * v[0] + B0) / (12.0 * v[1] + 136.0 * v[0] + 3.0 * B0);
}

#ifndef GL_D2_ORDER
#define GL_D2_ORDER 2                  // guess of a step's second solve: 1 = the previous step's correction, 2 = extrapolated from the last two
#endif
static const double GL_D2_KAPPA_MAX = 6.0;   // ... only where the solve's interval has lmax / lmin below this
static const int GL_CHEB_MAX = 96;    // launches of one solve at most
static const int GL_CHEB_LONG = 48;   // solves that would need more passes than this run PCG

struct ChebRun {
  int passes = 0;        // operator passes enqueued that can run (host-known count), or the upper bound when planned
  bool planned = false;  // the count is computed on the device; it arrives in the Krylov info of the next decision mail
};

// Solves A y = b (b = v.r, the Newton right-hand side at v.x) for the correction y of v.x: iterates in v.p / v.s (two buffers
// that change roles every pass), direction in cheb_dir; the last pass adds y to v.x and keeps it in ylast (take-back; next step's guess).
// v.r is left untouched unless want_res, in which case the final pass turns it into the residual b - A y.
// warm_u: the solve starts from this guess (the predicted increment, ghosts valid; may be ylast) instead of zero; the iteration count
// is then chosen on the device from |b - A u| (r_bound bounds the launches; hint_slot: which solve of the step), otherwise from r_norm = |b| here.
static ChebRun cheb_solve(glims_ctx* h, const CgVecs& v, double tol_abs, double r_norm, double r_bound, bool want_res,
                          const double* warm_u, double* ylast, int hint_slot) {
  const DevPattern& p = h->pat;
  const bool split = h->world > 1 && h->n_peers > 0;
  const int64_t n = h->n_own;
  h->cheb_dir.alloc((size_t)h->n_nodes);
  h->cheb_plan.alloc(1);
  double ia, ib;
  cheb_interval(h, tol_abs / std::max(r_norm > 0.0 ? r_norm : r_bound, tol_abs), &ia, &ib);
  ChebRec rec(ia, ib);
  PackMap pm;
  if (split && h->n_send > 0) {
    pm.ref = h->send_ref.p;
    pm.ptr = h->send_slot_ptr.p;
    pm.slot = h->send_slot.p;
    pm.sendbuf = h->sendbuf.p;
  }
  ChebRun run;
  run.planned = warm_u != nullptr;   // (then |r| on entry of the Chebyshev recurrence is only known to the device)
  const int m_min = want_res ? 1 : 2;
  int m = 0;
  if (run.planned) {
    // upper bound of the launches: from the residual before the warm start -- or, once known, what the device chose last time
    // + 2 (like the PCG solves' hints: the launches beyond the device's count return at once, but each still costs a dispatch
    // and, in a partitioned run, a halo exchange); a count clipped by the bound is a slightly weaker Newton step
    m = std::min(GL_CHEB_MAX, std::max(m_min, rec.iterations(tol_abs / std::max(r_bound, tol_abs)) + 3));
    if (h->cheb.m_hint[hint_slot] > 0) m = std::max(m_min, std::min(m, h->cheb.m_hint[hint_slot] + 2 + hint_slot));
  }
  else if (r_norm > tol_abs) m = std::min(GL_CHEB_MAX, std::max(m_min, rec.iterations(tol_abs / r_norm)));
  const unsigned g = grid_for(n);
  double* info_dev = h->scal.p + 2 * SC_COUNT;
  // Warm-started solve (warm_u = the predicted increment u, ghosts valid): u is the iterate of pass 1, which computes b - A u --
  // the product the warm start needs anyway --, takes the first Chebyshev direction from it and leaves the partial sums of its
  // square, from which the device chooses the count; no separate SpMV, no start kernel.
  const int shift = warm_u ? 1 : 0;
  double *d_in = v.p, *d_out = v.s;
  if (warm_u) {
    hipEvent_t* ev = h->timing(glims_ctx::TK_CHEB) ? h->pair(glims_ctx::TK_CHEB) : nullptr;
    // (one launch over all slices: the ghosts of u are current, nothing to exchange; the payload of d_1 is packed for pass 2)
    gl_launch_cheb(h, h->st, p.n_slices, nullptr, v.vals, v.vals32, warm_u, v.p, v.r, v.dinv, h->cheb_dir.p,
                   ylast, v.x, v.fixed, 0.0, 1.0 / rec.theta, 1, GL_CHEB_MAX + 8, nullptr, want_res ? 1 : 0, pm,
                   ev ? ev[0] : nullptr, ev ? ev[1] : nullptr, shift, h->partials.p);
    reduce_partials(h, gl_spmv_grid(p.n_slices), 1, nullptr);
    allreduce_sum(h, h->red.p, 1);
    hipLaunchKernelGGL(k_cheb_plan, dim3(1), dim3(1), 0, h->st, (const double*)h->red.p, tol_abs * tol_abs,
                       rec.inv_acosh_sigma(), 2, std::max(2, m), 2, h->cheb_plan.p, info_dev);
    GL_HIP(hipGetLastError());
    m = std::max(2, m);
  } else {
    // (zero guess: the first iterate y_1 = d_1 = Dinv b / theta needs no operator pass)
    hipLaunchKernelGGL(k_cheb_start, dim3(g), dim3(256), 0, h->st, n, (const double*)v.r, v.dinv, 1.0 / rec.theta, v.p,
                       h->cheb_dir.p, pm);
    GL_HIP(hipGetLastError());
  }
  const int last = (want_res ? m : m - 1) + shift;
  run.passes = std::max(0, last);
  for (int k = 1 + shift; k <= last; ++k) {
    double c1 = 0.0, c2 = 0.0;
    if (!want_res || k < m + shift) rec.next(&c1, &c2);
    const int* plan = run.planned ? h->cheb_plan.p : nullptr;
    hipEvent_t* ev = h->timing(glims_ctx::TK_CHEB) ? h->pair(glims_ctx::TK_CHEB) : nullptr;
    if (!split) {
      gl_launch_cheb(h, h->st, p.n_slices, nullptr, v.vals, v.vals32, d_in, d_out, v.r, v.dinv, h->cheb_dir.p,
                     ylast, v.x, v.fixed, c1, c2, k, m, plan, want_res ? 1 : 0, pm, ev ? ev[0] : nullptr,
                     ev ? ev[1] : nullptr, shift);
    } else {
      // the ghosts of d_in travel (payload packed by the kernel that produced it) while the slices without ghost columns run
      halo_start(h, d_in, 1, /*prepacked=*/true);
      gl_launch_cheb(h, h->st, p.n_interior, p.interior_slices.p, v.vals, v.vals32, d_in, d_out, v.r, v.dinv,
                     h->cheb_dir.p, ylast, v.x, v.fixed, c1, c2, k, m, plan, want_res ? 1 : 0, pm,
                     ev ? ev[0] : nullptr, ev ? ev[1] : nullptr, shift);
      halo_finish(h);
      gl_launch_cheb(h, h->st, p.n_boundary, p.boundary_slices.p, v.vals, v.vals32, d_in, d_out, v.r, v.dinv,
                     h->cheb_dir.p, ylast, v.x, v.fixed, c1, c2, k, m, plan, want_res ? 1 : 0, pm, nullptr, nullptr,
                     shift);
    }
    std::swap(d_in, d_out);
  }
  h->stats.cheb_solves++;
  return run;
}

// ===================================================================================================
// RD time stepping
// ===================================================================================================
// One sweep = Jacobian + right-hand side(s) + their norms.  With b2 the sweep also produces the first residual of
// the NEXT time step (same operator part 1/2 (A+S) c, different b), so that the convergence check of step n and
// the first assembly of step n+1 are one pass over the corner lists.
// `krylov`: also receives {iterations, rr, done} of a deferred linear solve enqueued before the sweep.
// `exchange_c`: the ghosts of c are stale (a linear solve has just updated the owned values): in a partitioned run
// the interior slices -- mass SpMV for b2 and sweep -- run while the halo is in flight, the boundary slices after it.
static void rd_sweep(glims_ctx* h, const double* b2, double* norms /*[2]*/, Mail* krylov = nullptr,
                     bool exchange_c = false, bool mass_for_b2 = false) {
  const DevPattern& p = h->pat;
  const double* load = h->have_load_rd ? h->load_rd.p : nullptr;
  const bool split = exchange_c && h->world > 1 && h->n_peers > 0;
  if (!split) {
    if (exchange_c) gl_halo_exchange(h, h->c.p, 1);
    if (mass_for_b2)
      gl_launch_spmv(h, h->st, p.n_slices, nullptr, h->vM.p, h->c.p, h->b2.p, nullptr, load, nullptr, nullptr, 0,
                     nullptr);
    const bool timed = h->timing(glims_ctx::TK_SWEEP);
    if (timed) h->tick(glims_ctx::TK_SWEEP);
    gl_rd_assemble(h, h->c.p, h->b.p, b2, h->cg_r.p, h->cg_r2.p, h->partials.p);
    if (timed) h->tick(glims_ctx::TK_SWEEP);
  } else {
    halo_start(h, h->c.p, 1);
    if (mass_for_b2)
      gl_launch_spmv(h, h->st, p.n_interior, p.interior_slices.p, h->vM.p, h->c.p, h->b2.p, nullptr, load, nullptr,
                     nullptr, 0, nullptr);
    gl_rd_assemble(h, h->c.p, h->b.p, b2, h->cg_r.p, h->cg_r2.p, h->partials.p, GL_PART_INTERIOR);
    halo_finish(h);
    if (mass_for_b2)
      gl_launch_spmv(h, h->st, p.n_boundary, p.boundary_slices.p, h->vM.p, h->c.p, h->b2.p, nullptr, load, nullptr,
                     nullptr, 0, nullptr);
    gl_rd_assemble(h, h->c.p, h->b.p, b2, h->cg_r.p, h->cg_r2.p, h->partials.p, GL_PART_BOUNDARY);
  }
  reduce_partials(h, gl_rd_grid(h), 2, nullptr);
  allreduce_sum(h, h->red.p, 2);
  const Mail m = fetch(h, 2, krylov != nullptr);
  norms[0] = std::sqrt(m.red[0]);
  norms[1] = std::sqrt(m.red[1]);
  if (krylov) *krylov = m;
  h->stats.rd_assemblies++;
}

// (a, delta) = (u, u)
__global__ void k_pair_of(int64_t n, const double* __restrict__ u, float2* __restrict__ ad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ad[i] = make_float2((float)u[i], (float)u[i]);
}
void gl_pair_of(glims_ctx* h, const double* u, float* ad) {
  hipLaunchKernelGGL(k_pair_of, dim3((unsigned)((h->n_nodes + 255) / 256)), dim3(256), 0, h->st, h->n_nodes, u, (float2*)ad);
  GL_HIP(hipGetLastError());
}
// a = c_new + c_k - 2 c_0 (= 2 (c_k - c_0) + delta),  delta = c_new - c_k   over all local nodes (ghosts included)
__global__ void k_quad_prep(int64_t i0, int64_t n, const double* __restrict__ cn, const double* __restrict__ ck,
                            const double* __restrict__ c0, float2* __restrict__ ad) {
  const int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = cn[i], y = ck[i], z = c0[i];
  ad[i] = make_float2((float)((x - z) + (y - z)), (float)(x - y));   // differences in double, then rounded
}

// The Newton residual after a solve with the step's first Jacobian, from the quadratic structure (k_rd_quad): the Krylov
// solver's final residual vector (in cg_r) becomes the next right-hand side, norms[0] its norm.  `krylov`: see rd_sweep.
static void rd_quad_update(glims_ctx* h, const double* ck, double* norms /*[2]*/, Mail* krylov) {
  // the ghosts of the new iterate travel while the slices without ghost columns are processed (those of c_k and c_0 came
  // with their copies), as in rd_sweep
  const bool split = h->world > 1 && h->n_peers > 0;
  auto prep = [&](int64_t i0, int64_t i1) {
    if (i1 > i0)
      hipLaunchKernelGGL(k_quad_prep, dim3(grid_exact(i1 - i0)), dim3(256), 0, h->st, i0, i1, h->c.p, ck, h->nq_c0.p,
                         (float2*)h->nq_ad.p);
  };
  if (!split) {
    gl_halo_exchange(h, h->c.p, 1);
    prep(0, h->n_nodes);
    const bool timed = h->timing(glims_ctx::TK_QUAD);
    if (timed) h->tick(glims_ctx::TK_QUAD);
    gl_rd_quad(h, h->nq_ad.p, h->cg_r.p, h->partials.p);
    if (timed) h->tick(glims_ctx::TK_QUAD);
  } else {
    halo_start(h, h->c.p, 1);
    prep(0, h->n_own);
    gl_rd_quad(h, h->nq_ad.p, h->cg_r.p, h->partials.p, GL_PART_INTERIOR);
    halo_finish(h);
    prep(h->n_own, h->n_nodes);
    gl_rd_quad(h, h->nq_ad.p, h->cg_r.p, h->partials.p, GL_PART_BOUNDARY);
  }
  reduce_partials(h, gl_rd_grid(h), 2, nullptr);
  allreduce_sum(h, h->red.p, 2);
  const Mail m = fetch(h, 2, krylov != nullptr);
  norms[0] = std::sqrt(m.red[0]);
  norms[1] = 0.0;
  if (krylov) *krylov = m;
  h->stats.rd_quad_updates++;
}

void glims_ctx::timing_begin() {
  if (opt.time_kernels && tev.empty()) {
    tev.resize(16384);
    tev_cat.assign(tev.size() / 2, 0);
    for (hipEvent_t& e : tev) GL_HIP(hipEventCreate(&e));
  }
  if (opt.time_kernels && world > 1 && cev.empty()) {
    cev.resize(8192);
    wev.resize(8192);
    for (hipEvent_t& e : cev) GL_HIP(hipEventCreate(&e));
    for (hipEvent_t& e : wev) GL_HIP(hipEventCreate(&e));
  }
  tev_used = 0;
  cev_used = wev_used = 0;
}

void glims_ctx::timing_collect() {
  if (cev_used >= 2 || wev_used >= 2) {   // partitioned run: exchange durations (communication stream) and exposed waits
    if (wev_used >= 2) GL_HIP(hipEventSynchronize(wev[wev_used - 1]));
    if (cev_used >= 2) GL_HIP(hipEventSynchronize(cev[cev_used - 1]));
    for (size_t q = 0; q + 1 < cev_used; q += 2) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, cev[q], cev[q + 1]) == hipSuccess) {
        stats.ms_exchange += t;
        stats.halo_exchanges_timed++;   // (the pool holds 4096 pairs per call; exchanges beyond it are counted, not timed)
      }
    }
    for (size_t q = 0; q + 1 < wev_used; q += 2) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, wev[q], wev[q + 1]) == hipSuccess) stats.ms_exchange_exposed += t;
    }
    cev_used = wev_used = 0;
  }
  if (tev_used < 2) return;
  // launches that the decision word turned into no-ops last a few microseconds: leave them out of sums and medians
  std::vector<float> d[TK_COUNT];
  for (size_t q = 0; q < tev_used / 2; ++q) {
    float t = 0.f;
    GL_HIP(hipEventElapsedTime(&t, tev[2 * q], tev[2 * q + 1]));
    d[std::min<int>(TK_COUNT - 1, tev_cat[q])].push_back(t);
  }
  double* sums[TK_COUNT] = {&stats.ms_spmv_steps, &stats.ms_sweep_steps, &stats.ms_update_steps, &stats.ms_mgfine_mech,
                            &stats.ms_spmvb_mech, &stats.ms_quad_steps, &stats.ms_cheb_steps};
  int64_t* cnts[TK_COUNT] = {&stats.n_spmv_steps, &stats.n_sweep_steps, &stats.n_update_steps, &stats.n_mgfine_mech,
                             &stats.n_spmvb_mech, &stats.n_quad_steps, &stats.n_cheb_steps};
  double* meds[TK_COUNT] = {&stats.us_spmv_median, &stats.us_sweep_median, &stats.us_update_median,
                            &stats.us_mgfine_median, &stats.us_spmvb_median, &stats.us_quad_median, &stats.us_cheb_median};
  for (int c = 0; c < TK_COUNT; ++c) {
    if (d[c].empty()) continue;
    // reference duration = the 90th percentile (no-op launches are the SHORT ones; a single pair that straddles a
    // preemption must not set the scale -- it once made every real launch look like a no-op): keep 0.2x .. 5x of it
    std::vector<float> sorted(d[c]);
    std::nth_element(sorted.begin(), sorted.begin() + (sorted.size() * 9) / 10, sorted.end());
    const float ref = sorted[(sorted.size() * 9) / 10];
    std::vector<float> real;
    for (float t : d[c])
      if (t > 0.2f * ref && t < 5.0f * ref) real.push_back(t);
    for (float t : real) {
      *sums[c] += t;
      ++*cnts[c];
    }
    if (!real.empty()) {
      std::nth_element(real.begin(), real.begin() + real.size() / 2, real.end());
      *meds[c] = 1e3 * real[real.size() / 2];
    }
  }
  tev_used = 0;
}

int gl_step(glims_ctx* h, int n_steps) {
  GL_REQUIRE(h->is_setup, "glims_step before glims_setup");
  GL_REQUIRE(h->have_state, "glims_step before glims_set_state");
  const DevPattern& p = h->pat;
  const int64_t n = h->n_own;
  const glims_options& o = h->opt;
  const bool extrapolate = (o.flags & GLIMS_FLAG_EXTRAPOLATE_GUESS) != 0;
  const double* load = h->have_load_rd ? h->load_rd.p : nullptr;
  int status = GLIMS_OK;
  h->timing_begin();
  if (h->rd_precond_active == 0) gl_rd_choose_precond(h);
  bool rd_mg = h->rd_precond_active == GLIMS_RD_PRECOND_MULTIGRID;
  if (rd_mg && !h->mg_rd.ready) gl_mg_setup_rd(h);
  const int64_t rd_cycles0 = h->mg_rd.cycles;
  GL_HIP(hipEventRecord(h->ev_a, h->st));
  for (int step = 0; step < n_steps && status == GLIMS_OK; ++step) {
    const int64_t newton0 = h->stats.newton_its, cg0 = h->stats.cg_its;
    double norms[2] = {0.0, 0.0};
    double nr;
    // Dot-free linear solves (glims_options.rd_linear): Jacobi-preconditioned RD solves run the Chebyshev iteration on the
    // interval measured by the PCG solves of a LEARNING step -- the first step of a run, every 32nd one after it, and the
    // step after a solve had to be taken back.
    glims_ctx::ChebState& cb = h->cheb;
    const bool cheb_allowed = o.rd_linear != GLIMS_RD_LINEAR_PCG && !rd_mg;
    if (cheb_allowed && cb.valid && ++cb.age >= 32) cb.valid = false;
    const bool cheb_learning = cheb_allowed && !cb.valid;
    if (cheb_learning) {
      cb.learned = cb.learned0 = 0;
      cb.pcg_best_its = 0;
      cb.m_hint[0] = cb.m_hint[1] = 0;
      h->cg_hist.alloc((size_t)2 * GL_CG_HIST);
    }
    if (h->pending) {
      // the sweep that verified the previous step already assembled A(c^n) and -R(c^n; c^n) for this one
      nr = h->pending_r0;
      h->pending = false;
    } else {
      // b = M c^n + load          ('u_previous1 * v1 * dx', simulation_tumor_growth.py:117)
      gl_launch_spmv(h, h->st, p.n_slices, nullptr, h->vM.p, h->c.p, h->b.p, nullptr, load, nullptr, nullptr, 0,
                     nullptr);
      // new Dirichlet data enter the ITERATE, after the old state went into b = M c^n: the reference's u_previous
      // keeps the previous step's boundary values while the DirichletBC constrains the unknown
      // (a rank lists its OWN constrained nodes; their ghost copies on the neighbours follow by a halo exchange that every
      //  rank takes part in -- also the ranks that own no constrained node: dirichlet_c_exchange)
      if (h->dirichlet_c_dirty) gl_apply_dirichlet_c(h);
      if (h->dirichlet_c_exchange) {
        gl_halo_exchange(h, h->c.p, 1);
        h->dirichlet_c_exchange = false;
        // new boundary values are a jump in the iterate: no warm start from the previous increments -- on EVERY rank (the
        // rank that wrote the values has dropped its own in gl_apply_dirichlet_c; a rank that kept it would run a differently
        // shaped first solve -- the warm-started dot-free solve reduces |r| first -- and the ranks' collectives would no longer pair up)
        h->have_c_old = false;
        h->have_d2 = false;
      }
      if (extrapolate) {
        hipLaunchKernelGGL(k_extrapolate, dim3(grid_exact(n)), dim3(256), 0, h->st, n, h->c.p, h->c_old.p,
                           h->have_fixed_c ? h->fixed_c.p : nullptr, h->stats.steps > 0 ? 1 : 0);
        gl_halo_exchange(h, h->c.p, 1);
      }
      rd_sweep(h, nullptr, norms);
      nr = norms[0];
    }
    const double r0 = nr;
    const double target = std::max(o.newton_atol, o.newton_rtol * r0);
    // Residuals from the quadratic structure (default; GLIMS_FLAG_FULL_NEWTON: a sweep after every solve).  Between two
    // sweeps the solves use A_0 = A(c_0), the Jacobian the last sweep assembled; after such a solve the residual is the
    // Krylov residual plus dt N(a) delta (rd_quad_update: 48 % of a sweep's bytes, no Jacobian written).  A sweep runs
    // after the step's first solve (see cheap_next), and
    // (a) where convergence is expected -- it returns the TRUE residual and assembles the next step's system, as before;
    // (b) when the cheap residual reports convergence without (a) having been predicted; (c) after an iteration whose
    // residual is more than 5 x the linear solve's tolerance (strong nonlinearity: dt rho |c - c_0| is no longer small) --
    // the sweep's Jacobian then becomes the new A_0.  Not combined with the extrapolated guess (no verifying sweep there).
    // (nor with the single-precision Jacobian: the Krylov residual then belongs to the rounded operator)
    // After a step in which (c) struck, the next eight steps run with sweeps only: where the nonlinearity is that strong a
    // cheap evaluation buys an extra Newton iteration (dt rho = 0.6: 46 against 38 iterations in 8 steps without this).
    if (h->nq_skip_steps > 0) --h->nq_skip_steps;
    const bool quad = (o.flags & GLIMS_FLAG_FULL_NEWTON) == 0 && !extrapolate && !h->jac32 && h->nq_skip_steps == 0;
    bool rebase = false;   // (c)
    // Copies of iterates are made only where a cheap evaluation follows (known before the solve): after a sweep the base
    // point c_0 IS the current iterate (`base_is_current`), so the first such copy serves as c_0 and as c_k (`ck_is_c0`);
    // a second cheap evaluation in a row copies c_k into a buffer of its own.  One 8 B / node copy per cheap evaluation.
    // Default forcing (round 4): the FIRST solve of a step decides whether two Newton iterations can be enough.  With the
    // second solve's Jacobian at c_1, two iterations leave q_2 (r_1 / r_0)^2 r_0, q_2 = what the quadratic term alone leaves of
    // a whole step (5e-4 late in config C4's run) -- below the Newton target 1e-10 r_0 only if r_1 <= 4e-4 r_0, i.e. with the
    // first solve at 0.3 cg_rtol instead of cg_rtol (+0-1 PCG iterations) AND a first residual not dominated by the quadratic
    // term.  Early in a run (small increments) the tolerance alone does it (mode 0: C4 steps 5-25 2.4 -> 2.1 iterations per
    // step, 9.50 -> 9.37 ms; brain-like mesh 2.74 -> 2.65; C3 1.50 -> 1.46); later the quadratic term has to go too, which is
    // what the midpoint correction of the first right-hand side does with the extrapolated increment (mode 1, one cheap pass:
    // C4 steps 120-160 three solves -> two, 8.89 -> 8.19 ms; steps 300-340 9.38 -> 8.71; C3 1.51 -> 1.29; brain-like mesh
    // 3.00 -> 2.70; profiles/r04_ab_midpoint.txt).  Either one alone is a loss there (tolerance alone: a third sweep, 9.67 ms;
    // correction alone: 9.39).  The mode follows the outcome: 0 until two steps within a few take three iterations (one step in ten
    // doing so is cheaper than a pass on every step), then 1; a step that
    // takes three WITH the correction sends the next 16 back to cg_rtol without it (mode 2: strong nonlinearity, where the
    // extra effort buys nothing); every 64th step in mode 1 tries mode 0 again; the first 8 steps of a run do not count (no
    // increments to extrapolate from yet).  GLIMS_FLAG_FIXED_FORCING: cg_rtol for every solve, no correction (mode 2 throughout).
    const bool fixed_forcing = (o.flags & GLIMS_FLAG_FIXED_FORCING) != 0;
    const int nw_mode = (quad && !fixed_forcing) ? h->nw_mode : 2;
    const bool midpoint = quad && nw_mode == 1;
    const double first_rtol = (quad && nw_mode != 2) ? 0.3 * o.cg_rtol : o.cg_rtol;
    bool base_is_current = true, ck_is_c0 = false;
    // (margin 3: with 1 the cheap pass reported convergence unpredicted -- pass + confirming sweep -- in 14-28 % of the steps
    //  of C4 / C3, with 3 in 2 %; with 10 the failed confirmations are back)
    const double spec_margin = 3.0;
    double ratio_est = h->nq_first_ratio;   // contraction of the previous Newton iteration (first one: of the last step's first)
    if (quad) {
      for (dvec<double>* v : {&h->nq_c0, &h->nq_ck}) v->alloc((size_t)h->n_nodes);
      h->nq_ad.alloc((size_t)2 * h->n_nodes);
    }
    double* last_ylast = nullptr;   // where the last dot-free solve left its correction
    int regime_now = -1;            // passes of this step's first solve when it was a dot-free one, and the forcing mode
    double r1_now = 0.0;            // the Newton residual this step's second solve started from
    bool pcg_rest = false;          // a dot-free solve of this step under-delivered: its remaining solves run PCG
    bool used_warm2 = false;        // this step's second solve started from the previous step's second correction
    bool warm2_miss = false;        // ... and left the residual above the target (the third iteration is the guess's doing)
    bool d2_pcg = false;            // ... the second solve was a PCG one
    if (h->d2_off > 0) {
      --h->d2_off;
      h->have_d2 = false;
    }
    bool d2_written = false;        // this step's second solve was a dot-free one (its correction is in cheb_delta2)
    for (int it = 0;; ++it) {
      if (!std::isfinite(nr)) {
        status = GLIMS_NAN;
        break;
      }
      if (nr <= target) break;
      if (it >= o.newton_maxit) {
        status = GLIMS_NOT_CONVERGED;
        break;
      }
      // A(c_k) delta = -R(c_k);  the update is accumulated straight into c (x0 = 0  <=>  x = c_k)
      // Forcing term.  The first solve of a step gets cg_rtol (1e-3): the quadratic term dt N(delta) delta that the step
      // leaves behind is of that size anyway.  From the second solve on the Jacobian is the one of c_1 and Newton converges
      // quadratically: the remainder after a solve from residual nr is ~ q nr^2 / r0, with q = the contraction the step's first
      // iteration was observed to achieve (R_1 / r_0: what the quadratic term alone leaves).  Solving to cg_rtol x nr again
      // would stop three decades short of that floor and spend a whole Newton iteration (evaluation, start-up of a solve) on
      // them: the linear tolerance follows the floor instead (Eisenstat & Walker's "eta_k = O(|R_k|)").  Where Jacobi-PCG
      // needs few iterations per decade (lattice configs: 3.1 Newton iterations per step either way) nothing changes; on the
      // unstructured brain-like mesh a step takes 2.25 Newton iterations instead of 4 and 30 PCG iterations instead of 39
      // (fewer restarts of the Krylov space): 3.86 -> 2.86 ms per step; C3 1.79 -> 1.65; C4 unchanged (10.7 vs 10.7-10.9).
      // Safety factor on the predicted remainder: 0.3 (with 1.0 more steps need a third iteration: 2.95 / 1.68 ms).
      // GLIMS_FLAG_FIXED_FORCING: cg_rtol always.
      const bool adaptive_forcing = (o.flags & GLIMS_FLAG_FIXED_FORCING) == 0 && it >= 1;
      const double floor_pred = std::min(0.5, std::max(1e-6, h->nq_first_ratio)) * nr * (nr / std::max(r0, 1e-300));
      const double tol_lin = std::max(std::max(o.cg_atol, 0.5 * target),
                                      adaptive_forcing ? std::min(o.cg_rtol * nr, 0.3 * floor_pred)
                                                       : (it == 0 ? first_rtol : o.cg_rtol) * nr);
      // what this iteration is expected to leave: the linear residual plus the quadratic remainder
      const double pred_next = adaptive_forcing ? tol_lin + floor_pred : nr * std::min(0.5, std::max(1e-6, ratio_est));
      // Newton converges quadratically here (the nonlinearity is exactly quadratic): once the residual before the
      // solve was below ~sqrt(rtol) of the initial one, the next sweep will almost surely only confirm convergence,
      // so let it also assemble the next step (costs one extra mass SpMV, saves a whole sweep per step).
      // Otherwise the evaluation after this solve is the cheap one -- both known before the solve.
      // With cheap evaluations a sweep that FAILS to confirm convergence is the expensive mistake (C4, steps 60-160 of the
      // 500: four iterations per step, the third evaluation a sweep that did not converge), so the prediction there is
      // "this iteration contracts like the previous one did": residual x last observed contraction <= target.
      const bool speculate_next =
          !extrapolate && (quad ? pred_next <= spec_margin * target
                                : nr <= 1e-4 * std::sqrt(o.newton_rtol / 1e-10) * r0);
      // (not after the step's FIRST solve, which takes the big step: its sweep moves A_0 to c_1, within ~1e-3 |delta_0| of
      //  the step's solution -- with A(c^n) kept instead every later iteration contracts by dt rho |c - c^n| ~ 3e-3 only,
      //  and the count per step rose from 3.35 to 3.65 at config C4)
      const bool cheap_next = quad && !speculate_next && !rebase && it >= 1;
      if (cheap_next) {   // c_k, the point the right-hand side belongs to (before a warm start moves c; ghosts are current)
        ck_is_c0 = base_is_current;
        base_is_current = false;
        GL_HIP(hipMemcpyAsync(ck_is_c0 ? h->nq_c0.p : h->nq_ck.p, h->c.p, (size_t)h->n_nodes * sizeof(double),
                              hipMemcpyDeviceToDevice, h->st));
      }
      bool use_cheb = cheb_allowed && cb.valid && !pcg_rest;
      if (use_cheb && tol_lin < nr) {
        double ia, ib;
        cheb_interval(h, tol_lin / nr, &ia, &ib);
        ChebRec rec(ia, ib);
        const int passes = rec.iterations(tol_lin / nr) - (cheap_next ? 0 : 1);
        if (passes > GL_CHEB_LONG) {
          // an ill-conditioned system (stiff step with the Jacobi preconditioner forced): the Chebyshev bound grows like
          // sqrt(kappa) per decade, PCG converges superlinearly there -- and a count cut off at GL_CHEB_MAX would be a weak solve
          use_cheb = false;
        } else if (it >= 1 && o.rd_linear == GLIMS_RD_LINEAR_AUTO && cb.cost_ratio > 0.0 && cb.pcg_its_per_decade > 0.0) {
          // a tight solve: PCG's iterations (from its rate in the last learning step) against the passes the Chebyshev bound
          // asks for, weighted by what each costs
          const double its_pcg = std::ceil(cb.pcg_its_per_decade * std::log10(nr / tol_lin)) + 1.0;
          if (its_pcg < 0.95 * cb.cost_ratio * passes) use_cheb = false;
        }
      }
      bool warm = false, ws_fused = false;
      if (it == 0 && (o.flags & GLIMS_FLAG_WARM_START) && !extrapolate) {   // both options own the c_old buffer
        // initial guess of the first linear solve = the increment predicted from the previous steps' (k_ws_delta): same linear
        // system, same solution, the Krylov iteration just starts closer.  One SpMV with the already assembled A(c^n).
        h->ws_du.alloc((size_t)h->n_nodes);
        // (dot-free solve: the guess u becomes direction 0 of the solve -- the product A u is then the solve's first operator
        //  pass and the correction accumulates from u: cheb_solve, warm_u)
        ws_fused = use_cheb && h->have_c_old;
        hipLaunchKernelGGL(k_ws_delta, dim3(grid_exact(h->n_nodes)), dim3(256), 0, h->st, h->n_nodes, h->c.p,
                           h->c_old.p, h->cg_u.p, h->ws_du.p, (h->have_c_old && h->ws_depth >= 1) ? 1 : 0);
        h->ws_depth = h->have_c_old ? 1 : 0;   // (ws_du holds a real increment from the second warm-started step on)
        if (h->have_c_old) {
          warm = true;
          if (!ws_fused) {
            gl_launch_spmv(h, h->st, p.n_slices, nullptr, h->vA.p, h->cg_u.p, h->cg_w.p,
                           h->have_fixed_c ? h->fixed_c.p : nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                           h->jac32 ? h->vA32.p : nullptr);
            hipLaunchKernelGGL(k_ws_apply, dim3(grid_exact(n)), dim3(256), 0, h->st, n, h->cg_r.p, h->cg_w.p, h->c.p,
                               h->cg_u.p);
          }
          // Midpoint correction of the step's first right-hand side.  For the exactly quadratic residual the whole step
          // delta* = c* - c_0 satisfies  A(c_0 + delta* / 2) delta* = -R(c_0)  -- the midpoint Jacobian solves the step in
          // ONE linear solve -- i.e.  A(c_0) delta* = -R(c_0) - dt N(delta*) delta*.  With delta* predicted by the previous
          // step's increment u (the warm-start vector) the right-hand side gets the term -dt N(u) u from one cheap pass:
          // what is left of the quadratic term after the first solve is dt rho |delta* - u| |delta*| instead of
          // dt rho |delta*|^2, so the first Newton iteration contracts as far as its linear tolerance lets it.
          if (midpoint) {
            h->nq_ad.alloc((size_t)2 * h->n_nodes);
            hipLaunchKernelGGL(k_pair_of, dim3(grid_exact(h->n_nodes)), dim3(256), 0, h->st, h->n_nodes, h->cg_u.p,
                               (float2*)h->nq_ad.p);
            gl_rd_quad(h, h->nq_ad.p, h->cg_r.p, h->partials.p);
            h->stats.rd_quad_updates++;
            h->stats.midpoint_steps++;
          }
        }
        h->have_c_old = true;
      }
      CgVecs v{h->c.p, h->cg_r.p, h->cg_u.p, h->cg_w.p, h->cg_p.p, h->cg_s.p,
               h->dinv.p, h->vA.p, h->have_fixed_c ? h->fixed_c.p : nullptr, 1};
      if (h->jac32) v.vals32 = h->vA32.p;
      if (rd_mg) {
        v.mg = &h->mg_rd;
        v.mg_degree = o.rd_mg_smooth > 0 ? o.rd_mg_smooth : (h->mg_rd.lattice ? 1 : 3);
      }
      int64_t its = 0;
      double res = 0.0;
      int slot = std::min(it, 6);   // (slot 7: second solves that start from the guess -- their counts say nothing about the ones from zero)
      int cs = GLIMS_OK;
      bool deferred = false;
      ChebRun crun;
      if (use_cheb) {
        // |r| on entry: known to the host unless the solve starts from the warm-start guess -- then the count is chosen on the
        // device from the norm its first pass measures
        // A step's SECOND solve starts from the previous step's second correction: what the first solve leaves behind -- the
        // quadratic remainder and the residual of a fixed polynomial applied to the extrapolation error -- changes by about a
        // per cent from one step to the next (|R_1| over steps 20-28 of config C4: 1.31, 1.32, 1.34, 1.36, 1.40e-5), so the
        // correction that removes it does too.
        const bool second = it == 1;
        h->cheb_delta.alloc((size_t)h->n_nodes);
        if (second) h->cheb_delta2.alloc((size_t)h->n_nodes);
        // (what the first solve leaves behind is a fixed polynomial of the operator applied to the extrapolation error: a first solve
        //  of another degree leaves something else -- config C4 / 8, step 24: 3 -> 2 passes, |R_1| 1.8e-5 -> 2.8e-5 -- and the
        //  second corrections before and after such a change do not continue each other)
        // The same for the forcing mode (a midpoint-corrected first right-hand side leaves a residual twenty times smaller), and
        // the check that needs no model: the correction is proportional to the residual it removes, so |R_1| has to continue too.
        if (second && h->have_d2 && (regime_now != h->d2_regime || !(nr > 0.7 * h->d2_r1 && nr < 1.43 * h->d2_r1)))
          h->have_d2 = false;
        if (!h->have_d2) h->d2_depth = 0;   // (dropped since the last step: a new state, new boundary values, a take-back)
        // Only on a narrow interval: the components that come back with the guess are the ones the solves' polynomial amplifies --
        // outside the interval, where it grows like exp(degree) -- and a wide interval means long solves (random-point mesh, 1 M
        // nodes, [0.18, 3.3]: 27-34 passes instead of 41-44 for eight steps, then a residual three times the target and a third
        // Newton iteration per step for the sixteen after: 2.9 -> 3.6 ms per step; not used there).
        bool narrow = false;
        if (second && h->have_d2) {
          double ia, ib;
          cheb_interval(h, tol_lin / nr, &ia, &ib);
          narrow = ib <= GL_D2_KAPPA_MAX * ia;
        }
        const bool warm2 = GL_D2_ORDER >= 1 && second && h->have_d2 && narrow && (o.flags & GLIMS_FLAG_WARM_START) && !extrapolate;
        if (warm2) {
          // (linearly extrapolated from the last two; the guess goes where the first solve's went: cg_u is free again)
          h->d2_prev.alloc((size_t)h->n_nodes);
          hipLaunchKernelGGL(k_d2_guess, dim3(grid_exact(n)), dim3(256), 0, h->st, n, (const double*)h->cheb_delta2.p,
                             h->d2_prev.p, h->cg_u.p, (h->d2_depth >= 2 && GL_D2_ORDER >= 2) ? 1 : 0);
          GL_HIP(hipGetLastError());
          if (h->world > 1) gl_halo_exchange(h, h->cg_u.p, 1);   // (corrections are kept by their row owners: ghosts)
        }
        last_ylast = second ? h->cheb_delta2.p : h->cheb_delta.p;
        if (second) {
          d2_written = true;
          r1_now = nr;
          used_warm2 = warm2;
        }
        crun = cheb_solve(h, v, tol_lin, nr, nr, cheap_next, (ws_fused || warm2) ? h->cg_u.p : (const double*)nullptr,
                          last_ylast, second ? 1 : 0);
        deferred = crun.planned;
        if (!deferred) {
          h->stats.cg_its += crun.passes;
          h->stats.cheb_its += crun.passes;
          h->stats.last_cg_res = tol_lin;
          if (it == 0) regime_now = 1000 + crun.passes + 100 * nw_mode;   // (a cold first solve: not the same thing as a warm one of that count)
        }
      } else {
        if (cheb_learning) v.hist = h->cg_hist.p;
        // The second solve's guess with PCG (the brain-like mesh's tight solves, multigrid-preconditioned stiff steps,
        // rd_linear = PCG): same guess, applied like the first solve's (r -= A u, c += u).  PCG accumulates into c, so the
        // correction is recovered as c after - c before.  No feedback problem here (PCG damps whatever the guess carries); the
        // learning steps' solves start from zero (their Lanczos coefficients are to describe the right-hand side itself).
        const bool second_pcg = it == 1 && !cheb_learning && (o.flags & GLIMS_FLAG_WARM_START) && !extrapolate;
        if (second_pcg) {
          h->cheb_delta2.alloc((size_t)h->n_nodes);
          const int regime_pcg = regime_now + 500;
          if (h->have_d2 && (regime_pcg != h->d2_regime || !(nr > 0.7 * h->d2_r1 && nr < 1.43 * h->d2_r1))) h->have_d2 = false;
          if (!h->have_d2) h->d2_depth = 0;
          const bool warm2 = GL_D2_ORDER >= 1 && h->have_d2;
          if (warm2) {
            h->d2_prev.alloc((size_t)h->n_nodes);
            hipLaunchKernelGGL(k_d2_guess, dim3(grid_exact(n)), dim3(256), 0, h->st, n, (const double*)h->cheb_delta2.p,
                               h->d2_prev.p, h->cg_u.p, (h->d2_depth >= 2 && GL_D2_ORDER >= 2) ? 1 : 0);
            if (h->world > 1) gl_halo_exchange(h, h->cg_u.p, 1);
          }
          GL_HIP(hipMemcpyAsync(h->cheb_delta2.p, h->c.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->st));
          if (warm2) {
            gl_launch_spmv(h, h->st, p.n_slices, nullptr, h->vA.p, h->cg_u.p, h->cg_w.p,
                           h->have_fixed_c ? h->fixed_c.p : nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                           h->jac32 ? h->vA32.p : nullptr);
            // (applied only if it lowers the residual -- decided on the device from two sums: on stiff steps, where the solution
            //  does not evolve smoothly from step to step, the extrapolated correction can be worse than none; 5.3 -> 5.6
            //  V-cycle-preconditioned iterations per solve at 1 M rows with the guess always applied)
            const unsigned gg = grid_for(n, 256, 1024);
            hipLaunchKernelGGL(k_guess_norms, dim3(gg), dim3(256), 0, h->st, n, (const double*)h->cg_r.p, (const double*)h->cg_w.p,
                               h->partials.p);
            reduce_partials(h, (int)gg, 2, nullptr);
            allreduce_sum(h, h->red.p, 2);
            hipLaunchKernelGGL(k_ws_apply_if, dim3(grid_exact(n)), dim3(256), 0, h->st, n, h->cg_r.p, (const double*)h->cg_w.p,
                               h->c.p, (const double*)h->cg_u.p, (const double*)h->red.p);
            GL_HIP(hipGetLastError());
          }
          d2_written = true;
          d2_pcg = true;
          r1_now = nr;
          used_warm2 = warm2;
          if (warm2) {
            slot = 7;
            if (h->cg_hint[7] <= 0) h->cg_hint[7] = h->cg_hint[1];   // (a first count to bound the launches: the solve from zero's)
          }
        }
        cs = cg_solve(h, v, tol_lin, o.cg_maxit, h->cg_hint[slot], &its, &res, /*defer=*/!cheb_learning);
        if (second_pcg) {
          hipLaunchKernelGGL(k_d2_from_state, dim3(grid_exact(n)), dim3(256), 0, h->st, n, (const double*)h->c.p,
                             h->cheb_delta2.p);
          GL_HIP(hipGetLastError());
        }
        deferred = its < 0;
        if (!deferred) {
          h->cg_hint[slot] = (int)its;
          h->stats.cg_its += its;
          h->stats.last_cg_res = res;
          if (cheb_learning && cs == GLIMS_OK) {
            // (reduction the solve achieved; a warm-started solve's initial residual is not known to the host: what was asked for)
            cheb_learn(h, its, (!warm && res > 0.0 && res < nr) ? res / nr : tol_lin / nr);
            // PCG's iterations per decade in the tightest solve whose initial residual the host knows (not a warm-started one)
            if (!warm && its > cb.pcg_best_its && res > 0.0 && res < nr) {
              cb.pcg_best_its = (int)its;
              cb.pcg_its_per_decade = (double)its / std::log10(nr / res);
            }
          }
        }
      }
      h->stats.newton_its++;
      if (cs != GLIMS_OK) {
        gl_halo_exchange(h, h->c.p, 1);   // ghosts of c current again
        status = cs;
        break;
      }
      const bool speculate = speculate_next, cheap = cheap_next;   // (decided before the solve, see there)
      Mail km;
      const double nr_before = nr;
      if (cheap) {
        rd_quad_update(h, ck_is_c0 ? h->nq_c0.p : h->nq_ck.p, norms, deferred ? &km : nullptr);
      } else {
        rd_sweep(h, speculate ? h->b2.p : nullptr, norms, deferred ? &km : nullptr, /*exchange_c=*/true,
                 /*mass_for_b2=*/speculate);
        base_is_current = true;   // a fresh Jacobian: A_0 = A(c) from here on
        rebase = false;
      }
      if (deferred && use_cheb) {   // the count the device chose for the warm-started solve
        const int64_t m_dev = (int64_t)km.info[0];
        cb.m_hint[it == 1 ? 1 : 0] = (int)std::max<int64_t>(1, m_dev);
        if (it == 0) regime_now = (int)m_dev + 100 * nw_mode;
        // (a count that used up its launches: the guess was further off than the last ones -- not a basis for the next step's)
        if (it == 1 && m_dev >= crun.passes) d2_written = false;
        const int64_t passes = std::max<int64_t>(0, std::min<int64_t>(crun.passes, (cheap ? m_dev : m_dev - 1) + 1));
        h->stats.cg_its += passes;
        h->stats.cheb_its += passes;
        h->stats.last_cg_res = tol_lin;
      } else if (deferred) {
        // the linear solve's outcome arrives with the sweep: a solve that used up its hint + 2 iterations simply
        // was a slightly weaker Newton step (the residual below decides); give it more room next time
        h->cg_hint[slot] = km.done == 1 ? (int)km.info[0] : (int)km.info[0] + 2;
        if (km.done != 1) h->stats_defer_miss++;
        h->stats.cg_its += (int64_t)km.info[0];
        h->stats.last_cg_res = std::sqrt(km.info[1]);
        if (km.done == 2) {
          status = GLIMS_NAN;
          break;
        }
        if (km.done == 3) h->cg_hint[slot] = 0;   // breakdown: next time take the polled path
      }
      nr = norms[0];
      if (use_cheb && getenv("GLIMS_VERBOSE_CHEB"))
        fprintf(stderr, "  cheb: step %lld it %d  %.3e -> %.3e  tol_lin %.3e target %.3e  %s passes %d (m_dev %g) %s\n",
                (long long)h->stats.steps, it, nr_before, nr, tol_lin, target, crun.planned ? "planned" : "host", crun.passes,
                crun.planned ? km.info[0] : -1.0, cheap ? "cheap" : (speculate ? "sweep+spec" : "sweep"));
      if (use_cheb && !(std::isfinite(nr) && (nr <= 0.5 * nr_before || nr <= target))) {
        // The Newton residual did not contract: part of the right-hand side lies outside the interval (the Chebyshev polynomial
        // grows there).  Take the correction back (it is still in cheb_delta), drop the interval -- this step's remaining
        // solves and the next step's run PCG and measure it again -- and repeat the iteration from a fresh sweep.
        if (getenv("GLIMS_VERBOSE"))
          fprintf(stderr, "glims dot-free solves: step %lld, Newton iteration %d: residual %.3e -> %.3e (target %.3e, linear tolerance "
                  "%.3e, %s count, %d passes enqueued) -- taken back, PCG from here\n", (long long)h->stats.steps, it, nr_before, nr,
                  target, tol_lin, crun.planned ? "device-side" : "host-side", crun.passes);
        hipLaunchKernelGGL(k_sub_inplace, dim3(grid_exact(n)), dim3(256), 0, h->st, n, h->c.p, (const double*)last_ylast);
        d2_written = false;
        h->have_d2 = false;
        if (it == 1 && used_warm2) {
          h->d2_off = h->d2_backoff;
          h->d2_backoff = std::min(256, 2 * h->d2_backoff);
          h->d2_good = 0;
        }
        GL_HIP(hipGetLastError());
        cb.valid = false;
        cb.lmax0 = cb.lmin0 = 0.0;   // (the loose interval is re-learnt from scratch)
        h->stats.cheb_fallbacks++;
        h->pending = false;
        rd_sweep(h, nullptr, norms, nullptr, /*exchange_c=*/true);
        nr = norms[0];
        base_is_current = true;
        rebase = false;
        continue;
      }
      // A second solve that started from the guess and left the residual above the target (by any margin): what it left is more
      // likely the guess's doing (components that the previous steps' solves amplified instead of damping come back with it) than
      // the quadratic remainder -- the third solve runs PCG, and the guess stays unused for a while (the solves from zero in
      // between start clean): 8 steps, doubling with every miss.
      if (it == 1 && used_warm2 && std::isfinite(nr)) {
        if (nr > target) {
          pcg_rest = true;
          d2_written = false;
          warm2_miss = true;
          h->d2_off = h->d2_backoff;
          h->d2_backoff = std::min(256, 2 * h->d2_backoff);
          h->d2_good = 0;
        } else if (++h->d2_good >= 32) {
          h->d2_backoff = 8;
        }
      }
      // A dot-free solve that contracts, but far less than it was sized for (10 x its tolerance plus the quadratic remainder),
      // has an interval that no longer fits what the right-hand sides excite: not a take-back -- the Newton iteration copes --
      // but two of them in a row make the next step a learning step instead of waiting for the 32nd.
      if (use_cheb && it >= 1 && std::isfinite(nr)) {
        const bool weak = nr > target && nr > 10.0 * (tol_lin + (adaptive_forcing ? floor_pred : 0.0));
        cb.weak = weak ? cb.weak + 1 : 0;
        if (cb.weak >= 2) {
          cb.age = 1 << 20;
          cb.weak = 0;
        }
      }
      if (std::isfinite(nr) && nr_before > 0.0) {
        ratio_est = nr / nr_before;
        if (it == 0) h->nq_first_ratio = ratio_est;
      }
      if (cheap && std::isfinite(nr)) {
        if (nr <= target) {
          // (b) converged by the cheap residual, unpredicted: the true residual and the next step's system from a sweep
          rd_sweep(h, h->b2.p, norms, nullptr, /*exchange_c=*/false, /*mass_for_b2=*/true);
          base_is_current = true;
          nr = norms[0];
          if (std::isfinite(nr) && nr <= target) {
            std::swap(h->b.p, h->b2.p);
            std::swap(h->cg_r.p, h->cg_r2.p);
            h->pending = true;
            h->pending_r0 = norms[1];
            break;
          }
        } else if (nr > 5.0 * (adaptive_forcing ? std::max(tol_lin, floor_pred) : o.cg_rtol * nr_before)) {
          // (c): the solve was asked for cg_rtol (cheap evaluations only happen where that bound, not the Newton target,
          // set its tolerance); a residual five times larger is the Jacobian's age showing
          rebase = true;
          h->nq_skip_steps = 9;
          h->stats.rebase_events++;
        }
      }
      if (speculate && std::isfinite(nr) && nr <= target) {
        std::swap(h->b.p, h->b2.p);
        std::swap(h->cg_r.p, h->cg_r2.p);
        h->pending = true;
        h->pending_r0 = norms[1];
        break;
      }
    }
    ++h->nw_steps;
    h->have_d2 = d2_written && status == GLIMS_OK;
    if (h->have_d2) {
      h->d2_regime = regime_now + (d2_pcg ? 500 : 0);
      h->d2_r1 = r1_now;
    }
    h->d2_depth = h->have_d2 ? std::min(2, h->d2_depth + 1) : 0;
    if (cheb_learning && status == GLIMS_OK && cb.learned > 0) {
      if (cb.cost_ratio == 0.0) cb.cost_ratio = cheb_cost_ratio(h);
      // (the interval forgets slowly: an upper end that one step's right-hand sides did not excite is not dropped at once)
      if (cb.lmax > 0.0) {
        cb.acc_lmax = std::max(cb.acc_lmax, 0.5 * (cb.acc_lmax + cb.lmax));
        cb.acc_lmin = std::min(cb.acc_lmin, 0.5 * (cb.acc_lmin + cb.lmin));
      }
      cb.lmin = cb.acc_lmin;
      cb.lmax = cb.acc_lmax;
      if (cb.learned0 > 0) {
        if (cb.lmax0 > 0.0) {
          cb.acc_lmax0 = std::max(cb.acc_lmax0, 0.5 * (cb.acc_lmax0 + cb.lmax0));
          cb.acc_lmin0 = std::min(cb.acc_lmin0, 0.5 * (cb.acc_lmin0 + cb.lmin0));
        }
        cb.lmin0 = cb.acc_lmin0;
        cb.lmax0 = cb.acc_lmax0;
      }
      cb.valid = true;
      cb.age = 0;
      h->stats.cheb_lmin = cb.lmin;
      h->stats.cheb_lmax = cb.lmax;
    }
    // (the first steps of a run have no increments to extrapolate from and take three iterations whatever the mode: they do
    //  not speak for it)
    if (quad && status == GLIMS_OK && !fixed_forcing && h->nw_steps > 8) {
      // (a third iteration that the second solve's guess caused says nothing about the forcing mode)
      const int64_t count = h->stats.newton_its - newton0 - (warm2_miss ? 1 : 0);
      if (h->nw_mode == 0) {
        // (one step in ten taking a third iteration is cheaper than the correction's pass on every step: two within a few)
        h->nw_hold = count >= 3 ? h->nw_hold + 4 : std::max(0, h->nw_hold - 1);
        if (h->nw_hold >= 6) {
          h->nw_mode = 1;
          h->nw_since = h->nw_hold = 0;
        }
      } else if (h->nw_mode == 1) {
        if (count >= 3) {
          h->nw_mode = 2;
          h->nw_hold = 16;
        } else if (++h->nw_since % 64 == 0) {
          h->nw_mode = 0;
          h->nw_hold = 0;
        }
      } else if (--h->nw_hold <= 0) {
        h->nw_mode = 1;
        h->nw_since = 0;
      }
    }
    h->stats.last_newton_res = nr;
    if (status == GLIMS_OK) h->stats.steps++;
    else h->stats.failed_steps++;
    // `auto` corrects its prediction by what the step just showed: Jacobi-PCG iterations per Newton solve above the
    // break-even -> the following steps use the hierarchy (the counts are global numbers: every rank switches together)
    if (status == GLIMS_OK && o.rd_precond == GLIMS_RD_PRECOND_AUTO && h->rd_precond_active == GLIMS_RD_PRECOND_JACOBI) {
      const int64_t dn = h->stats.newton_its - newton0, dc = h->stats.cg_its - cg0;
      if (dn > 0 && (double)dc / (double)dn > h->rd_break_even) {
        if (getenv("GLIMS_VERBOSE"))
          fprintf(stderr, "glims RD preconditioner: %.1f Jacobi-PCG iterations per Newton solve observed (break-even %.0f): "
                  "multigrid V-cycle from the next step on\n", (double)dc / (double)dn, h->rd_break_even);
        h->rd_precond_active = GLIMS_RD_PRECOND_MULTIGRID;
        h->stats.rd_precond_used = GLIMS_RD_PRECOND_MULTIGRID;
        rd_mg = true;
        if (!h->mg_rd.ready) gl_mg_setup_rd(h);
        for (int& hint : h->cg_hint) hint = 0;   // the counts of the Jacobi solves say nothing about the new ones
      }
    }
  }
  GL_HIP(hipEventRecord(h->ev_b, h->st));
  GL_HIP(hipEventSynchronize(h->ev_b));
  float ms = 0.f;
  GL_HIP(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
  h->stats.ms_steps += ms;
  h->stats.rd_mg_cycles += h->mg_rd.cycles - rd_cycles0;
  h->timing_collect();
  return status;
}

// ===================================================================================================
// L2 projection: M x = rhs (consistent P1 mass matrix), Jacobi-PCG with the same single-reduction recurrence
// ===================================================================================================
__global__ void k_mass_dinv(int64_t n_own, const int64_t* __restrict__ slice_ptr, const uint8_t* __restrict__ diag_k,
                            const double* __restrict__ vM, double* __restrict__ dinv) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_own) return;
  dinv[row] = 1.0 / vM[slice_ptr[row >> 6] + (int64_t)diag_k[row] * GL_WAVE + (row & 63)];
}

int gl_project(glims_ctx* h, double* rhs, double* x, double rtol) {
  GL_REQUIRE(h->is_setup, "glims_project before glims_setup");
  const int64_t n = h->n_own;
  h->pending = false;   // dinv and the PCG work vectors are shared with the time stepper
  hipLaunchKernelGGL(k_mass_dinv, dim3(grid_exact(n)), dim3(256), 0, h->st, n, h->pat.slice_ptr.p, h->pat.diag_k.p,
                     h->vM.p, h->dinv.p);
  const unsigned gd = grid_for(n, 256, 1024);
  hipLaunchKernelGGL(k_dot_partials, dim3(gd), dim3(256), 0, h->st, n, rhs, rhs, h->partials.p);
  reduce_partials(h, (int)gd, 1, nullptr);
  allreduce_sum(h, h->red.p, 1);
  const double nb = std::sqrt(read_red0(h));
  GL_HIP(hipMemsetAsync(x, 0, (size_t)h->n_nodes * sizeof(double), h->st));
  if (!(nb > 0.0)) return std::isfinite(nb) ? GLIMS_OK : GLIMS_NAN;
  CgVecs v{x, rhs, h->cg_u.p, h->cg_w.p, h->cg_p.p, h->cg_s.p, h->dinv.p, h->vM.p, nullptr, 1};
  int64_t its = 0;
  double res = 0.0;
  const int cs = cg_solve(h, v, rtol * nb, 10000, 0, &its, &res);
  gl_halo_exchange(h, x, 1);
  GL_HIP(hipStreamSynchronize(h->st));
  return cs;
}

// ===================================================================================================
// mechanics:  K_el u = G c + f,  Dirichlet dofs eliminated symmetrically (projected operator P K P)
// ===================================================================================================
int gl_solve_mechanics(glims_ctx* h, const double* c_dev) {
  GL_REQUIRE(h->is_setup && h->have_mech, "glims_solve_mechanics needs glims_setup(with_mechanics=1)");
  GL_REQUIRE(h->have_state, "glims_solve_mechanics before glims_set_state");
  const DevPattern& p = h->pat;
  const int bs = h->dim;
  const int64_t nd = h->n_own * bs;
  const uint8_t* fx = h->have_fixed_u ? h->fixed_u.p : nullptr;
  const int mh_depth = std::max(0, std::min((int)glims_ctx::MHIST, h->opt.mech_history));
  if (h->mh_next >= std::max(1, mh_depth)) h->mh_count = h->mh_next = 0;   // the depth option shrank
  const double t_wall0 = omp_get_wtime();
  if (h->opt.time_kernels) h->timing_begin();   // (kernel pairs: time_kernels = 3 only; exchange pairs of a partitioned run: any)
  // preconditioner of the constrained operator: one multigrid V-cycle (built on first use, K_el does not change in
  // time) or block-Jacobi
  const bool use_mg = h->opt.mech_precond == GLIMS_PRECOND_MULTIGRID;
  if (use_mg && !h->mg.ready) gl_mg_setup_mech(h);
  else if (!use_mg) gl_block_dinv(h);
  // rhs = G c + f - K u_D, zero on constrained dofs
  gl_apply_G(h, c_dev ? c_dev : h->c.p, h->m_rhs.p);
  if (fx) {
    gl_halo_exchange(h, h->m_uD.p, bs);
    gl_spmv_block(h, h->m_uD.p, h->m_w.p, false);
    hipLaunchKernelGGL(k_sub, dim3(grid_exact(nd)), dim3(256), 0, h->st, nd, h->m_rhs.p, h->m_rhs.p, h->m_w.p, fx);
  }
  // ||rhs||^2 for the relative tolerance and g_k = (rhs_k, rhs) against the stored right-hand sides of the solve
  // history: three products per pass over rhs and one host read per pass (one product per launch and read cost 27
  // round trips = 0.8 ms per solve at depth 6); the Gram matrix of the stored right-hand sides is kept on the host
  // and grows by the row g when this solve's right-hand side is stored
  const unsigned gd = grid_for(nd, 256, 1024);
  const int m_hist = (h->mh_count > 0 && mh_depth > 0) ? h->mh_count : 0;
  double g[glims_ctx::MHIST] = {0.0};
  double nb2 = 0.0;
  for (int q0 = -1; q0 < m_hist; q0 += 3) {   // value -1 = (rhs, rhs), values 0.. = (rhs_k, rhs)
    Dot3 dv;
    const int nq = std::min(3, m_hist - q0);
    for (int q = 0; q < 3; ++q) {
      const int k = q0 + q;
      dv.a[q] = (q < nq) ? (k < 0 ? h->m_rhs.p : h->mh_rhs[k].p) : h->m_rhs.p;
    }
    hipLaunchKernelGGL(k_dot3, dim3(gd), dim3(256), 0, h->st, nd, nq, dv, h->m_rhs.p, h->partials.p);
    GL_HIP(hipGetLastError());
    reduce_partials(h, (int)gd, nq, nullptr);
    allreduce_sum(h, h->red.p, nq);   // partitioned run: every rank gets the same sums, hence the same coefficients
    double out[3];
    read_red(h, nq, out);
    for (int q = 0; q < nq; ++q) {
      const int k = q0 + q;
      if (k < 0) nb2 = out[q];
      else g[k] = out[q];
    }
  }
  const double nb = std::sqrt(nb2);
  if (!std::isfinite(nb)) return GLIMS_NAN;
  const double tol = std::max(h->opt.mech_atol, h->opt.mech_rtol * nb);
  // Initial guess.  K_el is linear and does not change in time, so if rhs ~ sum_k a_k rhs_k for previously solved
  // right-hand sides then x ~ sum_k a_k x_k with residual rhs - sum_k a_k rhs_k: least-squares fit over the stored
  // solves; the concentration evolves smoothly, the fit removes several decades of the initial residual.
  // Falls back to the previous displacement (the fit with a = e_last) when that is better or the history is empty.
  if (m_hist > 0) {
    const int m = m_hist;
    double (*G)[glims_ctx::MHIST] = h->mh_G;
    // normal equations with a small ridge (the right-hand sides of consecutive steps are nearly parallel)
    double a[glims_ctx::MHIST] = {0.0};
    {
      double A[glims_ctx::MHIST][glims_ctx::MHIST + 1];
      double tr = 0.0;
      for (int k = 0; k < m; ++k) tr += G[k][k];
      for (int k = 0; k < m; ++k) {
        for (int l = 0; l < m; ++l) A[k][l] = G[k][l] + (k == l ? 1e-13 * tr : 0.0);
        A[k][m] = g[k];
      }
      bool ok = true;
      for (int c = 0; c < m && ok; ++c) {   // Gaussian elimination with partial pivoting on the m x m system
        int piv = c;
        for (int r = c + 1; r < m; ++r)
          if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
        if (!(std::fabs(A[piv][c]) > 0.0)) {
          ok = false;
          break;
        }
        for (int q = 0; q <= m; ++q) std::swap(A[c][q], A[piv][q]);
        for (int r = c + 1; r < m; ++r) {
          const double f = A[r][c] / A[c][c];
          for (int q = c; q <= m; ++q) A[r][q] -= f * A[c][q];
        }
      }
      if (ok)
        for (int c = m - 1; c >= 0; --c) {
          double v = A[c][m];
          for (int q = c + 1; q < m; ++q) v -= A[c][q] * a[q];
          a[c] = v / A[c][c];
        }
      // predicted squared residual of the fit against the plain warm start (most recent solution alone)
      auto res2 = [&](const double* coef) {
        double v = nb * nb;
        for (int k = 0; k < m; ++k) {
          v -= 2.0 * coef[k] * g[k];
          for (int l = 0; l < m; ++l) v += coef[k] * coef[l] * G[k][l];
        }
        return v;
      };
      double e_last[glims_ctx::MHIST] = {0.0};
      const int last = (h->mh_next + mh_depth - 1) % mh_depth;
      e_last[last] = 1.0;
      bool finite = ok;
      for (int k = 0; k < m; ++k) finite = finite && std::isfinite(a[k]);
      if (!finite || !(res2(a) <= res2(e_last)))
        for (int k = 0; k < m; ++k) a[k] = e_last[k];
    }
    LinComb lc;
    for (int k = 0; k < 16; ++k) {
      lc.x[k] = h->mh_x[k < m ? k : 0].p;
      lc.a[k] = k < m ? a[k] : 0.0;
    }
    hipLaunchKernelGGL(k_lincomb, dim3(grid_exact(nd)), dim3(256), 0, h->st, nd, m, lc, h->U.p);
    // K_el of that combination without an operator pass: K x_k = rhs_k - (final residual of solve k) was stored with x_k
    for (int k = 0; k < 16; ++k) lc.x[k] = h->mh_w[k < m ? k : 0].p;
    hipLaunchKernelGGL(k_lincomb, dim3(grid_exact(nd)), dim3(256), 0, h->st, nd, m, lc, h->m_w.p);
    GL_HIP(hipGetLastError());
  }
  // x = guess on the free dofs (previous displacement if there is no history), 0 on constrained ones
  if (fx) hipLaunchKernelGGL(k_mask_assign, dim3(grid_exact(nd)), dim3(256), 0, h->st, nd, h->U.p, fx,
                             (const double*)nullptr);
  gl_halo_exchange(h, h->U.p, bs);
  if (m_hist == 0) gl_launch_spmv_block(h, h->st, p.n_slices, nullptr, h->U.p, h->m_w.p, fx, nullptr, nullptr, 0, nullptr);
  hipLaunchKernelGGL(k_sub, dim3(grid_exact(nd)), dim3(256), 0, h->st, nd, h->m_r.p, h->m_rhs.p, h->m_w.p, fx);
  GL_HIP(hipGetLastError());
  CgVecs v{h->U.p, h->m_r.p, h->m_u.p, h->m_w.p, h->m_p.p, h->m_s.p, h->m_dinv.p, nullptr, fx, bs};
  v.mg = use_mg ? &h->mg : nullptr;
  v.mg_degree = h->opt.mg_smooth;
  const int64_t cycles0 = h->mg.cycles;
  int64_t its = 0;
  double res = 0.0;
  int cs = GLIMS_OK;
  // Mixed precision (mech_mixed: 0 off, 1 auto, 2 always).  Auto = with the block-Jacobi preconditioner, where an
  // iteration IS one operator pass, and only where the operator is streamed from HBM (below ~256 MB, the Infinity Cache,
  // an iteration is latency bound and the restarts of the refinement loop only add iterations).  Never with the multigrid
  // preconditioner: the operator pass is 7 % of an iteration there, and the restarts cost more than that -- C5 at 10 M
  // nodes 102 vs 107 ms per solve, Delaunay 1 M points 99 vs 117 ms, and with a stiff inclusion (stiffness jump 1e4) the
  // restarted iteration loses the superlinear phase of CG altogether: 11 iterations against 29-55.
  const bool big = (size_t)p.total_entries * bs * bs * sizeof(double) > ((size_t)256 << 20);
  const bool mixed = h->opt.mech_mixed == 2 || (h->opt.mech_mixed == 1 && big && !use_mg);
  if (mixed) gl_make_kel32(h);
  auto norm_of_m_r = [&]() {
    hipLaunchKernelGGL(k_dot_partials, dim3(gd), dim3(256), 0, h->st, nd, h->m_r.p, h->m_r.p, h->partials.p);
    reduce_partials(h, (int)gd, 1, nullptr);
    allreduce_sum(h, h->red.p, 1);
    return std::sqrt(read_red0(h));
  };
  if (!mixed) {
    cs = cg_solve(h, v, tol, h->opt.mech_maxit, h->mech_hint, &its, &res);
    h->mech_hint = (int)its;
    // Verification pass with the fp64 operator: m_w = K U, m_r = rhs - K U.  The recurrence residual the Krylov iteration
    // stops on drifts away from the true one by rounding, and the solve history below keeps K x_k for the next solves'
    // initial residuals -- it must be the product itself, not "rhs minus what the recurrence believes", or the gap of one
    // solve enters every later one through the (sign-alternating) least-squares coefficients and is never seen again.
    // One operator pass per solve (0.23 ms of ~10 at config C5).  If the true residual misses the tolerance by more than
    // the drift one expects, the iteration continues from it.
    for (int round = 0; cs == GLIMS_OK; ++round) {
      gl_halo_exchange(h, h->U.p, bs);
      gl_launch_spmv_block(h, h->st, p.n_slices, nullptr, h->U.p, h->m_w.p, fx, nullptr, nullptr, 0, nullptr);
      hipLaunchKernelGGL(k_sub, dim3(grid_exact(nd)), dim3(256), 0, h->st, nd, h->m_r.p, h->m_rhs.p, h->m_w.p, fx);
      GL_HIP(hipGetLastError());
      res = norm_of_m_r();
      if (!std::isfinite(res)) {
        cs = GLIMS_NAN;
        break;
      }
      if (getenv("GLIMS_VERBOSE"))
        fprintf(stderr, "glims elasticity solve: %lld iterations, recurrence residual <= %.3e, true residual %.3e (%.2f x the tolerance)\n",
                (long long)its, tol, res, res / tol);
      if (res <= 4.0 * tol || round >= 2) break;
      int64_t more = 0;
      double r2 = 0.0;
      cs = cg_solve(h, v, tol, h->opt.mech_maxit, 0, &more, &r2);
      its += more;
    }
  } else {
    // Mixed precision: the inner PCG streams a single-precision copy of K_el (40 instead of 76 bytes per block entry;
    // products and sums in fp64) and reduces the residual by 1e-3; the outer loop recomputes the true residual with
    // the fp64 operator and repeats.  The answer converges to the fp64 tolerance like the plain solver's (iterative
    // refinement); if a cycle gains less than a factor 2 the remaining ones use the fp64 operator.
    double nr = norm_of_m_r();
    v.k32 = true;
    for (int outer = 0; outer < 12; ++outer) {
      res = nr;
      if (!std::isfinite(nr)) {
        cs = GLIMS_NAN;
        break;
      }
      if (nr <= tol) break;
      if (its >= h->opt.mech_maxit || outer == 11) {
        cs = GLIMS_NOT_CONVERGED;
        break;
      }
      int64_t in_its = 0;
      double in_res = 0.0;
      // inner reduction 1e-6 (round 1: 1e-3): the single-precision operator is good for that, and with the
      // solve-history guess (initial residual ~1e-6 |rhs|) ONE inner solve then reaches the target -- the loop costs
      // two fp64 operator passes (initial residual, verification) instead of four to five
      const int ics = cg_solve(h, v, std::max(tol, 1e-6 * nr), (int)std::max<int64_t>(1, h->opt.mech_maxit - its), 0,
                               &in_its, &in_res);
      its += in_its;
      if (ics == GLIMS_NAN) {
        cs = ics;
        break;
      }
      gl_halo_exchange(h, h->U.p, bs);
      gl_launch_spmv_block(h, h->st, p.n_slices, nullptr, h->U.p, h->m_w.p, fx, nullptr, nullptr, 0, nullptr);
      hipLaunchKernelGGL(k_sub, dim3(grid_exact(nd)), dim3(256), 0, h->st, nd, h->m_r.p, h->m_rhs.p, h->m_w.p, fx);
      GL_HIP(hipGetLastError());
      const double nr_new = norm_of_m_r();
      if (!(nr_new < 0.5 * nr)) v.k32 = false;   // single precision has given what it can
      nr = nr_new;
    }
  }
  h->stats.mech_cg_its += its;
  h->stats.mg_cycles += h->mg.cycles - cycles0;
  h->stats.mech_solves++;
  h->stats.last_mech_res = res;
  if (cs == GLIMS_OK && mh_depth > 0) {   // remember (rhs, free-dof solution) for the next initial guess
    const int slot = h->mh_next;
    h->mh_rhs[slot].alloc((size_t)h->n_nodes * bs);
    h->mh_x[slot].alloc((size_t)h->n_nodes * bs);
    GL_HIP(hipMemcpyAsync(h->mh_rhs[slot].p, h->m_rhs.p, (size_t)nd * sizeof(double), hipMemcpyDeviceToDevice, h->st));
    GL_HIP(hipMemcpyAsync(h->mh_x[slot].p, h->U.p, (size_t)nd * sizeof(double), hipMemcpyDeviceToDevice, h->st));
    // m_w = K U on the free dofs: the product of the verification pass itself (see above), never "rhs - recurrence residual"
    h->mh_w[slot].alloc((size_t)h->n_nodes * bs);
    GL_HIP(hipMemcpyAsync(h->mh_w[slot].p, h->m_w.p, (size_t)nd * sizeof(double), hipMemcpyDeviceToDevice, h->st));
    for (int l = 0; l < m_hist; ++l) h->mh_G[slot][l] = h->mh_G[l][slot] = g[l];   // g against the slots that stay
    h->mh_G[slot][slot] = nb2;
    h->mh_next = (slot + 1) % mh_depth;
    h->mh_count = std::min(h->mh_count + 1, mh_depth);
  }
  if (fx) hipLaunchKernelGGL(k_mask_assign, dim3(grid_exact(nd)), dim3(256), 0, h->st, nd, h->U.p, fx,
                             (const double*)h->m_uD.p);
  gl_halo_exchange(h, h->U.p, bs);
  GL_HIP(hipStreamSynchronize(h->st));
  h->stats.ms_mech += 1e3 * (omp_get_wtime() - t_wall0);
  if (h->opt.time_kernels) h->timing_collect();
  return cs;
}


// One node-mailbox all-reduce of known values (rank + 1, 2 (rank + 1), 1) with a 5 s limit; every rank must call it.
int gl_mailbox_selftest(glims_ctx* h) {
  GL_REQUIRE(h->nm.slots, "glims_comm_mailbox_selftest without a mailbox");
  const double v[3] = {h->rank + 1.0, 2.0 * (h->rank + 1.0), 1.0};
  GL_HIP(hipMemcpyAsync(h->partials.p, v, sizeof(v), hipMemcpyHostToDevice, h->st));
  NodeMail nm = h->nm;
  nm.timeout_ticks = 5ll * 100000000ll;
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, 1, 3, h->partials.p, h->red.p, (const int*)nullptr, nm);
  GL_HIP(hipGetLastError());
  double out[3] = {0.0, 0.0, 0.0};
  int err = 0;
  GL_HIP(hipMemcpyAsync(out, h->red.p, sizeof(out), hipMemcpyDeviceToHost, h->st));
  GL_HIP(hipMemcpyAsync(&err, h->nm.err, sizeof(int), hipMemcpyDeviceToHost, h->st));
  GL_HIP(hipStreamSynchronize(h->st));
  const double w = h->world, tri = 0.5 * w * (w + 1.0);
  if (err || out[0] != tri || out[1] != 2.0 * tri || out[2] != w) {
    GL_HIP(hipMemsetAsync(h->nm.err, 0, sizeof(int), h->st));
    GL_HIP(hipStreamSynchronize(h->st));
    throw glims_error(GLIMS_E_RCCL, "node mailbox self-test failed (sum " + std::to_string(out[0]) + ", expected " +
                                        std::to_string(tri) + (err ? ", timed out)" : ")"));
  }
  return GLIMS_OK;
}

// ===================================================================================================
// RCCL self-test on a one-rank communicator (see glims_comm_selftest in the header)
// ===================================================================================================
__global__ void k_fill_iota(int64_t n, double* __restrict__ v, double scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = scale * (double)(i + 1);
}

int gl_comm_selftest(glims_ctx* h) {
  const int64_t n = 4096;
  const int bs = 3;
  ncclUniqueId ida, idb;
  GL_NCCL(ncclGetUniqueId(&ida));
  GL_NCCL(ncclGetUniqueId(&idb));
  ncclComm_t ca = nullptr, cb = nullptr;
  GL_NCCL(ncclCommInitRank(&ca, 1, ida, 0));
  GL_NCCL(ncclCommInitRank(&cb, 1, idb, 0));
  dvec<double> vec, sendbuf, red;
  dvec<int32_t> idx;
  vec.alloc_zero((size_t)2 * n * bs, h->st);          // [owned n | ghosts n], block size 3
  sendbuf.alloc_zero((size_t)n * bs, h->st);
  red.alloc_zero(4, h->st);
  std::vector<int32_t> hidx(n);
  for (int64_t k = 0; k < n; ++k) hidx[k] = (int32_t)(n - 1 - k);   // send the owned nodes in reverse order
  idx.upload(hidx, h->st);
  hipLaunchKernelGGL(k_fill_iota, dim3(grid_exact(n * bs)), dim3(256), 0, h->st, n * bs, vec.p, 0.5);
  // --- the halo sequence of halo_start()/halo_finish(), peer = self ---
  hipLaunchKernelGGL(k_pack, dim3(grid_exact(n * bs)), dim3(256), 0, h->st, n, bs, idx.p, vec.p, sendbuf.p);
  GL_HIP(hipEventRecord(h->ev_pack, h->st));
  GL_HIP(hipStreamWaitEvent(h->st_comm, h->ev_pack, 0));
  GL_NCCL(ncclGroupStart());
  GL_NCCL(ncclSend(sendbuf.p, (size_t)n * bs, ncclDouble, 0, ca, h->st_comm));
  GL_NCCL(ncclRecv(vec.p + n * bs, (size_t)n * bs, ncclDouble, 0, ca, h->st_comm));
  GL_NCCL(ncclGroupEnd());
  GL_HIP(hipEventRecord(h->ev_halo, h->st_comm));
  GL_HIP(hipStreamWaitEvent(h->st, h->ev_halo, 0));
  // --- the reduction sequence of allreduce_sum() ---
  hipLaunchKernelGGL(k_fill_iota, dim3(1), dim3(256), 0, h->st, (int64_t)3, red.p, 1.25);
  GL_NCCL(ncclAllReduce(red.p, red.p, 3, ncclDouble, ncclSum, cb, h->st));
  std::vector<double> out((size_t)2 * n * bs), r3(3);
  GL_HIP(hipMemcpyAsync(out.data(), vec.p, out.size() * sizeof(double), hipMemcpyDeviceToHost, h->st));
  GL_HIP(hipMemcpyAsync(r3.data(), red.p, 3 * sizeof(double), hipMemcpyDeviceToHost, h->st));
  GL_HIP(hipStreamSynchronize(h->st));
  GL_HIP(hipStreamSynchronize(h->st_comm));
  (void)ncclCommDestroy(ca);
  (void)ncclCommDestroy(cb);
  for (int64_t k = 0; k < n; ++k)
    for (int a = 0; a < bs; ++a) {
      const double want = 0.5 * (double)((n - 1 - k) * bs + a + 1);
      if (out[(size_t)(n + k) * bs + a] != want)
        throw glims_error(GLIMS_E_RCCL, "RCCL self-test: ghost value mismatch at " + std::to_string(k));
    }
  for (int q = 0; q < 3; ++q)
    if (r3[q] != 1.25 * (q + 1)) throw glims_error(GLIMS_E_RCCL, "RCCL self-test: all-reduce value mismatch");
  return GLIMS_OK;
}
