// HIP kernels for gfx950 (CDNA4, wave64).  No MFMA: nothing here is a dense contraction; every kernel is
// bound by HBM / L2 gather bandwidth, so the design goal is unit-stride streams + no atomics.
//
// Row ownership model: one lane = one matrix row, one wavefront = one SELL-64 slice.  Lane l of slice s reads
// entry k of its row at  base(s) + k*64 + l  -> every wave-level load of values / columns / corner records is a
// contiguous 512 B / 256 B segment.  Element ("corner") contributions are accumulated by the owning lane in its
// private LDS column acc[slot*64 + lane] (bank = lane -> conflict-free for any slot), so the segmented scatter-add
// of FEM assembly needs neither atomics nor colouring and is bitwise reproducible.
#include "glims_internal.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Epilogue of the dot-fused SpMV kernels (256-thread blocks): one partial per BLOCK, added over its 4 waves in a
// fixed order (a partial per wave meant 4x as many values for the reduction kernel that follows, and a first
// reduction stage at 10 M rows).
// (A "last block done" reduction fused into this epilogue was built and measured in round 2: with the partials in
//  uncached memory and two-level ticket counters the SpMV went from 36 to 62 us at 1 M rows -- more than the separate
//  reduction kernel and its dispatch gap cost together, 7 + 6 us; with device-scope fences it was 15x slower still.)
__device__ __forceinline__ void spmv_dot_partial(double pd, int b, double* __restrict__ partials, int partial_off) {
  __shared__ double smd[4];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  pd = wave_sum(pd);
  if (lane == 0) smd[wid] = pd;
  __syncthreads();
  if (threadIdx.x == 0) partials[partial_off + b] = (smd[0] + smd[1]) + (smd[2] + smd[3]);
}

// ---------------------------------------------------------------------------------------------------
// per-cell geometry: |T| and grad(lambda_a)
// ---------------------------------------------------------------------------------------------------
template <int D>
__global__ void k_egeo(int64_t n_cells, const double* __restrict__ xyz, const int32_t* __restrict__ cells,
                       double* __restrict__ egeo, double* __restrict__ evol,
                       unsigned long long* __restrict__ bad /*[2]: count, first (caller's) index*/,
                       const int32_t* __restrict__ cell_new2old /*internal -> caller's cell index, or null*/) {
  constexpr int NV = D + 1, GE = 1 + NV * D;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_cells) return;
  double X[NV][D];
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    const int64_t v = cells[e * NV + m];
#pragma unroll
    for (int a = 0; a < D; ++a) X[m][a] = xyz[v * D + a];
  }
  double* g = egeo + e * GE;
  if constexpr (D == 2) {
    const double a = X[1][0] - X[0][0], b = X[1][1] - X[0][1];
    const double c = X[2][0] - X[0][0], d = X[2][1] - X[0][1];
    const double det = a * d - b * c, inv = 1.0 / det;
    const double g1x = d * inv, g1y = -c * inv, g2x = -b * inv, g2y = a * inv;
    g[0] = fabs(det) * 0.5;
    g[1] = -(g1x + g2x);
    g[2] = -(g1y + g2y);
    g[3] = g1x;
    g[4] = g1y;
    g[5] = g2x;
    g[6] = g2y;
  } else {
    double r[3][3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int a = 0; a < 3; ++a) r[m][a] = X[m + 1][a] - X[0][a];
    double c12[3] = {r[1][1] * r[2][2] - r[1][2] * r[2][1], r[1][2] * r[2][0] - r[1][0] * r[2][2],
                     r[1][0] * r[2][1] - r[1][1] * r[2][0]};
    double c20[3] = {r[2][1] * r[0][2] - r[2][2] * r[0][1], r[2][2] * r[0][0] - r[2][0] * r[0][2],
                     r[2][0] * r[0][1] - r[2][1] * r[0][0]};
    double c01[3] = {r[0][1] * r[1][2] - r[0][2] * r[1][1], r[0][2] * r[1][0] - r[0][0] * r[1][2],
                     r[0][0] * r[1][1] - r[0][1] * r[1][0]};
    const double det = r[0][0] * c12[0] + r[0][1] * c12[1] + r[0][2] * c12[2];
    const double inv = 1.0 / det;
    g[0] = fabs(det) * (1.0 / 6.0);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double g1 = c12[a] * inv, g2 = c20[a] * inv, g3 = c01[a] * inv;
      g[1 + 0 * 3 + a] = -(g1 + g2 + g3);
      g[1 + 1 * 3 + a] = g1;
      g[1 + 2 * 3 + a] = g2;
      g[1 + 3 * 3 + a] = g3;
    }
  }
  evol[e] = g[0];   // the volumes once more, compact: what the per-incidence reaction weights read (k_corner_weights)
  if (!(g[0] > 0.0) || !isfinite(1.0 / g[0])) {   // degenerate (zero-volume) or non-finite cell
    atomicAdd(bad, 1ull);
    atomicMin(bad + 1, (unsigned long long)(cell_new2old ? cell_new2old[e] : e));
  }
}

// Per (row, cell) incidence: the reaction weight rho_T |T| d!/(d+3)! and the slot word RE-ORDERED for the hot kernels --
// byte 0 = the slot of the row's own node in its row (the diagonal slot), bytes 1.. = the other vertices of the cell in their
// cell order.  The sweep and the quadratic-term pass then need no "is this vertex the row's own" test: the own vertex's
// contribution (the diagonal formula) goes to a register, the others (the off-diagonal formula) to their slots.  (`cslots`
// keeps the cell's vertex order, which the static assembly needs to find the vertex's gradient.)  One wave per slice.
template <int D>
__global__ __launch_bounds__(GL_WAVE) void k_corner_weights(const int64_t* __restrict__ cslice_ptr,
                                                             const int32_t* __restrict__ celem,
                                                             const uint8_t* __restrict__ label,
                                                             const double* __restrict__ evol, const double* __restrict__ mat,
                                                             const uint8_t* __restrict__ diag_k,
                                                             const uint32_t* __restrict__ cslots, double* __restrict__ cw,
                                                             uint32_t* __restrict__ cs2, uint2* __restrict__ cq) {
  constexpr int NV = D + 1;
  constexpr double fact = D == 2 ? 1.0 / 60.0 : 1.0 / 120.0;
  // (neighbouring slices on ONE XCD: the four rows that share a cell then share an L2 -- with block b on XCD b % 8 each of them
  //  fetched the cell's data from memory on its own)
  const int s = xcd_chunk_remap(blockIdx.x, gridDim.x, GL_XCD_CHUNK), lane = threadIdx.x;
  const int64_t cbase = cslice_ptr[s];
  const int clen = (int)((cslice_ptr[s + 1] - cbase) >> 6);
  const uint32_t dk = diag_k[(int64_t)s * GL_WAVE + lane];
  for (int q = 0; q < clen; ++q) {
    const int64_t i = cbase + (int64_t)q * GL_WAVE + lane;
    const int32_t e = celem[i];
    // (|T| from the compact array: through the 104-byte geometry records this kernel fetched a whole line per incidence,
    //  41.8 GB at 10 M rows -- profiles/r04_a_pmc_c4.json)
    const double w = e < 0 ? 0.0 : mat[1 * GL_MAX_LABELS + label[e]] * evol[e] * fact;
    const uint32_t sl = cslots[i];
    uint32_t out = dk, pos = 1;
    if (e < 0) {
      out = 0u;   // padding: weight 0, every slot 0
    } else {
#pragma unroll
      for (int m = 0; m < NV; ++m) {
        const uint32_t k = (sl >> (8 * m)) & 255u;
        if (k != dk) {
          out |= k << (8 * pos);
          ++pos;
        }
      }
    }
    cw[i] = w;
    cs2[i] = out;
    // the same record for the quadratic-term pass: slot word + the weight in single precision, 8 B instead of 12 (the term it
    // feeds is itself a 1e-3 correction)
    cq[i] = make_uint2(out, __float_as_uint((float)w));
  }
}

// ---------------------------------------------------------------------------------------------------
// static assembly: one scalar plane of one operator per launch (setup path, not timed per step)
// ---------------------------------------------------------------------------------------------------
enum { MODE_M = 0, MODE_KEL = 2, MODE_G = 3 };   // MODE_M: the two scalar planes M and S together

// NP planes per launch, all from ONE walk over the row's (row, cell) incidences: {M, S}, the D blocks K_el(ca, 0..D-1) of
// one block row, or the D components of G.  (Until round 3 every plane was a launch of its own, 14 with mechanics in 3-D,
// and each re-read the incidence records and -- through cell ids in the caller's numbering, i.e. with poor locality -- the
// cells' 104-byte geometry records: 41-83 GB of fetches per plane at 10 M rows, profiles/r02_pmc_c4.json.)  Every plane
// accumulates the same terms in the same order as before: the operators are bit-identical.
struct PlaneOut {
  double* out[3];
  int stride[3], off[3];
};
template <int D, int NP>
__global__ __launch_bounds__(GL_WAVE) void k_assemble_static(
    int mode, int ca, int64_t n_own, const int64_t* __restrict__ slice_ptr,
    const int64_t* __restrict__ cslice_ptr, const uint32_t* __restrict__ cslots,
    const int32_t* __restrict__ celem, const uint8_t* __restrict__ diag_k, const double* __restrict__ egeo,
    const uint8_t* __restrict__ label, const double* __restrict__ mat, double dt, const PlaneOut po, int max_len) {
  constexpr int NV = D + 1, GE = 1 + NV * D;
  constexpr double mfac = 1.0 / ((D + 1) * (D + 2));
  extern __shared__ double acc[];   // [NP][max_len][64]
  const int s = xcd_chunk_remap(blockIdx.x, gridDim.x, GL_XCD_CHUNK), lane = threadIdx.x;   // (see k_corner_weights)
  const int64_t row = (int64_t)s * GL_WAVE + lane;
  const int64_t base = slice_ptr[s], cbase = cslice_ptr[s];
  const int len = (int)((slice_ptr[s + 1] - base) >> 6), clen = (int)((cslice_ptr[s + 1] - cbase) >> 6);
#pragma unroll
  for (int pl = 0; pl < NP; ++pl)
    for (int k = 0; k < len; ++k) acc[(pl * max_len + k) * GL_WAVE + lane] = 0.0;
  const int dk = diag_k[row];
  for (int q = 0; q < clen; ++q) {
    const int64_t ci = cbase + (int64_t)q * GL_WAVE + lane;
    const int32_t e = celem[ci];
    if (e < 0) continue;
    const uint32_t slots = cslots[ci];
    // (The record staged through LDS by the whole wave -- consecutive lanes reading consecutive doubles of a record, 5-6 lines per
    //  load instruction instead of 64 -- was built and measured in round 5: 42 instead of 48 GB of fabric reads, but 15.2 instead of
    //  10.3 ms at 10 M rows (two barriers, 13 cross-lane reads and 26 LDS accesses per incidence round): taken out again.)
    const double* g = egeo + (int64_t)e * GE;
    const double vol = g[0];
    const int lab = label[e];
    int li = 0;
#pragma unroll
    for (int m = 0; m < NV; ++m)
      if ((int)((slots >> (8 * m)) & 255u) == dk) li = m;
    double gi[D];
#pragma unroll
    for (int a = 0; a < D; ++a) gi[a] = g[1 + li * D + a];
    const double Dc = mat[0 * GL_MAX_LABELS + lab], rho = mat[1 * GL_MAX_LABELS + lab],
                 gam = mat[2 * GL_MAX_LABELS + lab], mu = mat[3 * GL_MAX_LABELS + lab],
                 lam = mat[4 * GL_MAX_LABELS + lab];
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int k = (int)((slots >> (8 * m)) & 255u);
      double gm[D], gg = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        gm[a] = g[1 + m * D + a];
        gg += gi[a] * gm[a];
      }
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        double v;
        if (mode == MODE_M)   // planes: 0 = M, 1 = S
          v = pl == 0 ? vol * mfac * (m == li ? 2.0 : 1.0)
                      : (1.0 - dt * rho) * vol * mfac * (m == li ? 2.0 : 1.0) + dt * Dc * vol * gg;
        else if (mode == MODE_KEL)   // plane pl = block entry (ca, cb = pl)
          v = vol * (lam * gi[ca] * gm[pl] + mu * gi[pl] * gm[ca] + (ca == pl ? mu * gg : 0.0));
        else                         // plane pl = component ca = pl of G
          v = gam * (2.0 * mu + D * lam) * vol * (1.0 / (D + 1)) * gi[pl];
        acc[(pl * max_len + k) * GL_WAVE + lane] += v;
      }
    }
  }
#pragma unroll
  for (int pl = 0; pl < NP; ++pl)
    for (int k = 0; k < len; ++k)
      po.out[pl][(base + (int64_t)k * GL_WAVE) * po.stride[pl] + po.off[pl] + lane] = acc[(pl * max_len + k) * GL_WAVE + lane];
}

// ---------------------------------------------------------------------------------------------------
// hot: RD Jacobian + Newton residual in one sweep
//   A(c) = S + 2 dt N(c),  N(c)_ij = sum_T w_T (c_i + c_j + s_T)  (i != j),  N_ii = sum_T w_T (4 c_i + 2 s_T),
//   s_T = sum of c over the cell's vertices, w_T = rho_T |T| d!/(d+3)!   [exact integral of rho c_h phi_i phi_j]
//   -R = b - 1/2 (A + S) c,   b = M c_prev + load
// ---------------------------------------------------------------------------------------------------
// column of one entry from its 16-bit code: window base by cross-lane read (lane w of `wb` holds base w), + offset
__device__ __forceinline__ int32_t decode_col(uint32_t code, int32_t wb) {
  return __builtin_amdgcn_ds_bpermute((int)((code >> GL_WIN_BITS) << 2), wb) + (int32_t)(code & ((1u << GL_WIN_BITS) - 1u));
}

// B (row, cell) incidence records of one lane: issue all loads, then apply them in order
template <int B, int NT, class F>
__device__ __forceinline__ void corner_batch(const double* __restrict__ wp, const uint32_t* __restrict__ sl, int q,
                                             F& corner) {
  double wb[B];
  uint32_t sb[B];
#pragma unroll
  for (int j = 0; j < B; ++j) {
    wb[j] = NT ? __builtin_nontemporal_load(wp + (int64_t)(q + j) * GL_WAVE) : wp[(int64_t)(q + j) * GL_WAVE];
    sb[j] = NT ? __builtin_nontemporal_load(sl + (int64_t)(q + j) * GL_WAVE) : sl[(int64_t)(q + j) * GL_WAVE];
  }
#pragma unroll
  for (int j = 0; j < B; ++j) corner(wb[j], sb[j]);
}

// this lane's LDS accumulator += v as ONE LDS instruction (ds_add_f64, no return value): the same IEEE addition in the same
// order as read - add - write (LDS instructions of a wave execute in order and every lane owns its column), but the
// incidence loop no longer waits for an accumulator to come back before the next record can start
__device__ __forceinline__ void lds_add(double* p, double v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int NV, int NT, int CU, int CIDX, class AT = double>
__global__ __launch_bounds__(GL_WAVE) void k_rd_assemble(
    const int32_t* __restrict__ slice_list, int64_t n_own, const int64_t* __restrict__ slice_ptr,
    const int32_t* __restrict__ cols, const uint16_t* __restrict__ cols16, const int32_t* __restrict__ win_base,
    const uint8_t* __restrict__ win_ok,
    const int64_t* __restrict__ cslice_ptr, const uint32_t* __restrict__ cslots, const double* __restrict__ cw,
    const uint8_t* __restrict__ diag_k, const double* __restrict__ vS, AT* __restrict__ vA,
    const double* __restrict__ c, const double* __restrict__ b, const double* __restrict__ b2,
    double* __restrict__ r_out, double* __restrict__ r2_out, double* __restrict__ dinv,
    const uint8_t* __restrict__ fixed, double two_dt, double* __restrict__ partials, int max_len, int remap) {
  // one block (= one wave) per slice of this launch's length class; LDS = 2 * max_len columns of 64 doubles
  extern __shared__ double lds[];
  double* acc = lds;
  double* cn = lds + (size_t)max_len * GL_WAVE;
  const int lane = threadIdx.x;
  double rr = 0.0, rr2 = 0.0;
  // same XCD-chunked mapping as the SpMV (one slice per block here, so 4x the chunk length)
  const int s = slice_list[remap > 1 ? xcd_chunk_remap(blockIdx.x, gridDim.x, 4 * remap) : blockIdx.x];
  {
    const int64_t row = (int64_t)s * GL_WAVE + lane;
    const int64_t base = slice_ptr[s], cbase = cslice_ptr[s];
    const int len = (int)((slice_ptr[s + 1] - base) >> 6), clen = (int)((cslice_ptr[s + 1] - cbase) >> 6);
    // phase 1: gather the row's neighbour values of c into the lane's LDS column (same pattern as the SpMV gather)
    auto gather = [&](auto load_col) {
      int k = 0;
      for (; k + 8 <= len; k += 8) {
        int32_t ci8[8];
        double x8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ci8[j] = load_col(k + j);
#pragma unroll
        for (int j = 0; j < 8; ++j) x8[j] = c[ci8[j]];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          cn[(k + j) * GL_WAVE + lane] = x8[j];
          acc[(k + j) * GL_WAVE + lane] = 0.0;
        }
      }
      if (k < len) {   // ragged tail as one more batch of 8 (slot index clamped), not as a serial loop
        int32_t ci8[8];
        double x8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ci8[j] = load_col(min(k + j, len - 1));
#pragma unroll
        for (int j = 0; j < 8; ++j) x8[j] = c[ci8[j]];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k + j < len) {
            cn[(k + j) * GL_WAVE + lane] = x8[j];
            acc[(k + j) * GL_WAVE + lane] = 0.0;
          }
      }
    };
    if (CIDX && win_ok[s]) {   // wave-uniform: 16-bit (window, offset) codes, see spmv_row
      const uint16_t* c16 = cols16 + base + lane;
      const int32_t wb = win_base[(int64_t)s * GL_N_WIN + (lane & (GL_N_WIN - 1))];
      gather([&](int k) {
        return decode_col(NT ? __builtin_nontemporal_load(c16 + (int64_t)k * GL_WAVE) : c16[(int64_t)k * GL_WAVE], wb);
      });
    } else {
      const int32_t* cc = cols + base + lane;
      gather([&](int k) {
        return NT ? __builtin_nontemporal_load(cc + (int64_t)k * GL_WAVE) : cc[(int64_t)k * GL_WAVE];
      });
    }
    const int dk = diag_k[row];
    const double ci = cn[dk * GL_WAVE + lane];
    const uint32_t* sl = cslots + cbase + lane;
    const double* wp = cw + cbase + lane;
    // phase 2: element contributions, CU incidence records in flight per lane (24 = a whole interior row of a
    // tetrahedral mesh; measured 4 -> 8 -> 24: 15.65 -> 15.3 -> 14.7 ms/step at C4)
    // (records carry the row's own slot in byte 0, k_corner_weights: the diagonal accumulator is a register, the others are
    //  bumped in LDS; the same expressions as in k_rd_assemble_s, so that both kernels produce the same bits)
    const double ci4 = 4.0 * ci;
    double acc_d = 0.0;
    auto corner = [&](double w, uint32_t slots) {
      int k[NV];
      double cv[NV], st = ci;
#pragma unroll
      for (int m = 1; m < NV; ++m) {
        k[m] = (int)((slots >> (8 * m)) & 255u);
        cv[m] = cn[k[m] * GL_WAVE + lane];
      }
#pragma unroll
      for (int m = 1; m < NV; ++m) st += cv[m];
      acc_d += w * (ci4 + 2.0 * st);
#pragma unroll
      for (int m = 1; m < NV; ++m) lds_add(&acc[k[m] * GL_WAVE + lane], w * (ci + cv[m] + st));
    };
    {
      int q = 0;
      for (; q + CU <= clen; q += CU) corner_batch<CU, NT>(wp, sl, q, corner);
      for (; q + 4 <= clen; q += 4) corner_batch<4, NT>(wp, sl, q, corner);     // ragged tail of longer slices
      for (; q < clen; ++q) corner_batch<1, NT>(wp, sl, q, corner);
    }
    // phase 3: A = S + 2 dt N(c), residual 1/2 (A + S) c, diagonal
    const double* sv = vS + base + lane;
    AT* av = vA + base + lane;
    double r = 0.0, d = 1.0;
    {
      int k = 0;
      for (; k + 8 <= len; k += 8) {
        double S8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          S8[j] = NT ? __builtin_nontemporal_load(sv + (int64_t)(k + j) * GL_WAVE) : sv[(int64_t)(k + j) * GL_WAVE];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const double a_k = (k + j == dk) ? acc_d : acc[(k + j) * GL_WAVE + lane];
          const double Av = S8[j] + two_dt * a_k;
          if (NT) __builtin_nontemporal_store((AT)Av, av + (int64_t)(k + j) * GL_WAVE);
          else av[(int64_t)(k + j) * GL_WAVE] = (AT)Av;
          r += 0.5 * (Av + S8[j]) * cn[(k + j) * GL_WAVE + lane];
          if (k + j == dk) d = Av;
        }
      }
      if (k < len) {   // ragged tail: one more batch, loads issued together
        double S8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kk = min(k + j, len - 1);
          S8[j] = NT ? __builtin_nontemporal_load(sv + (int64_t)kk * GL_WAVE) : sv[(int64_t)kk * GL_WAVE];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k + j < len) {
            const double a_k = (k + j == dk) ? acc_d : acc[(k + j) * GL_WAVE + lane];
            const double Av = S8[j] + two_dt * a_k;
            if (NT) __builtin_nontemporal_store((AT)Av, av + (int64_t)(k + j) * GL_WAVE);
            else av[(int64_t)(k + j) * GL_WAVE] = (AT)Av;
            r += 0.5 * (Av + S8[j]) * cn[(k + j) * GL_WAVE + lane];
            if (k + j == dk) d = Av;
          }
      }
    }
    if (row < n_own) {
      const bool fx = fixed && fixed[row];
      const double res = fx ? 0.0 : b[row] - r;
      r_out[row] = res;
      dinv[row] = fx ? 1.0 : 1.0 / d;
      rr += res * res;
      if (b2) {
        const double res2 = fx ? 0.0 : b2[row] - r;
        r2_out[row] = res2;
        rr2 += res2 * res2;
      }
    }
  }
  rr = wave_sum(rr);
  rr2 = wave_sum(rr2);
  if (lane == 0) {
    partials[(size_t)s * 2 + 0] = rr;
    partials[(size_t)s * 2 + 1] = rr2;
  }
}

// ---------------------------------------------------------------------------------------------------
// Newton residual WITHOUT a sweep.  The RD residual is exactly quadratic in c:
//     R(c + delta) = R(c) + A(c) delta + dt N(delta) delta,      A(c) = S + 2 dt N(c),  N linear in its argument,
// so after a linear solve  A0 delta = -R_k  with the Jacobian A0 = A(c_0) of the step's first iterate
//     R_{k+1} = (R_k + A0 delta) + dt N(a) delta,     a = 2 (c_k - c_0) + delta,
// and the bracket is minus the Krylov solver's final residual vector.  This kernel turns that vector (in r) into the next
// right-hand side -R_{k+1} = r - dt N(a) delta: phases 1 and 2 of k_rd_assemble with two staged vectors (a, delta) and a
// scalar accumulator per row -- the incidence records and the column codes are streamed, neither S nor A.
// partials: [slice][2] = (|r_new|^2 of the slice's rows, 0).
// ---------------------------------------------------------------------------------------------------
template <int NV, int CU, int CIDX>
__global__ __launch_bounds__(GL_WAVE) void k_rd_quad(
    const int32_t* __restrict__ slice_list, int64_t n_own, const int64_t* __restrict__ slice_ptr,
    const int32_t* __restrict__ cols, const uint16_t* __restrict__ cols16, const int32_t* __restrict__ win_base,
    const uint8_t* __restrict__ win_ok, const int64_t* __restrict__ cslice_ptr, const uint2* __restrict__ cq,
    const uint8_t* __restrict__ diag_k, const float2* __restrict__ ad,
    double* __restrict__ r, const uint8_t* __restrict__ fixed, double dt,
    double* __restrict__ partials, int max_len, int remap) {
  // (a, delta) pairs in single precision, one 8-byte LDS column per neighbour: the term is a correction of relative size
  // dt rho |delta| to the residual, its rounding error 1e-7 of THAT -- far below the Newton target -- and half the LDS of
  // two double columns means twice the waves per CU on a kernel that is bound by the incidence loop's LDS traffic
  extern __shared__ float2 ldsq[];
  const int lane = threadIdx.x;
  const int s = slice_list[remap > 1 ? xcd_chunk_remap(blockIdx.x, gridDim.x, 4 * remap) : blockIdx.x];
  const int64_t row = (int64_t)s * GL_WAVE + lane;
  const int64_t base = slice_ptr[s], cbase = cslice_ptr[s];
  const int len = (int)((slice_ptr[s + 1] - base) >> 6), clen = (int)((cslice_ptr[s + 1] - cbase) >> 6);
  auto gather = [&](auto load_col) {
    for (int k = 0; k < len; k += 8) {   // (ragged tail: slot index clamped, surplus stores skipped)
      int32_t ci8[8];
      float2 v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ci8[j] = load_col(min(k + j, len - 1));
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] = ad[ci8[j]];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k + j < len) ldsq[(k + j) * GL_WAVE + lane] = v8[j];
    }
  };
  if (CIDX && win_ok[s]) {
    const uint16_t* c16 = cols16 + base + lane;
    const int32_t wb = win_base[(int64_t)s * GL_N_WIN + (lane & (GL_N_WIN - 1))];
    gather([&](int k) { return decode_col(c16[(int64_t)k * GL_WAVE], wb); });
  } else {
    const int32_t* cc = cols + base + lane;
    gather([&](int k) { return cc[(int64_t)k * GL_WAVE]; });
  }
  const int dk = diag_k[row];
  const float2 own_v = ldsq[dk * GL_WAVE + lane];
  const float ai = own_v.x, di = own_v.y;
  const uint2* rec = cq + cbase + lane;
  float q = 0.0f;
  // (records carry the row's own slot in byte 0; same expressions as k_rd_quad_s)
  auto corner = [&](float w, uint32_t slots) {
    float2 v[NV];
    float sa = ai;
#pragma unroll
    for (int m = 1; m < NV; ++m) v[m] = ldsq[(int)((slots >> (8 * m)) & 255u) * GL_WAVE + lane];
#pragma unroll
    for (int m = 1; m < NV; ++m) sa += v[m].x;
    float t = (4.0f * ai + 2.0f * sa) * di;
#pragma unroll
    for (int m = 1; m < NV; ++m) t += (ai + v[m].x + sa) * v[m].y;
    q += w * t;
  };
  for (int qq = 0; qq < clen; qq += CU) {   // CU records in flight (index clamped past the end: padding has weight 0)
    uint2 rb[CU];
#pragma unroll
    for (int j = 0; j < CU; ++j) rb[j] = rec[(int64_t)min(qq + j, clen - 1) * GL_WAVE];
#pragma unroll
    for (int j = 0; j < CU; ++j)
      if (qq + j < clen) corner(__uint_as_float(rb[j].y), rb[j].x);
  }
  double rr = 0.0;
  if (row < n_own) {
    const double rn = (fixed && fixed[row]) ? 0.0 : r[row] - dt * (double)q;
    r[row] = rn;
    rr = rn * rn;
  }
  rr = wave_sum(rr);
  if (lane == 0) {
    partials[(size_t)s * 2 + 0] = rr;
    partials[(size_t)s * 2 + 1] = 0.0;
  }
}


// The same pass with the wave's memory round trips cut from ~13 to 4.  A wave of the kernel above walks a chain of dependent
// loads -- slice list, slice offsets, window flag, window bases, then per batch of 8 entries codes -> gather, incidence
// offsets, incidence records, r -- and lives ~20 us however little the memory system has to do; the kernel's rate is
// (resident waves) / (that lifetime), 2.4 TB/s on a general mesh.  Here: (1) one 32-byte scalar load of the slice's
// descriptor; (2) EVERY stream that depends on it alone is requested at once -- the first RB incidence records, r, the
// Dirichlet flag, the diagonal slot, window bases and all column codes of the row (CAP of them, compile-time: straight-line
// code, indices clamped instead of predicated so that no branch separates the loads); (3) all gathers of the row; (4) records
// beyond RB.  Records carry the row's own slot in byte 0 (k_corner_weights), so the own vertex's term needs neither an LDS
// read nor a test; the row sum is accumulated in single precision like its terms.
// Uniform base + lane offset: the compiler then addresses every stream as SGPR base + 32-bit VGPR offset.
#define GL_STREAM(ptr, base, k) ((ptr) + (base) + (int64_t)(k) * GL_WAVE)[lane_u]
template <int NV, int CAP, int RB, int CIDX>
__global__ __launch_bounds__(GL_WAVE) void k_rd_quad_s(
    const SliceDesc* __restrict__ desc, int64_t n_own, const int32_t* __restrict__ cols,
    const uint16_t* __restrict__ cols16, const int32_t* __restrict__ win_base, const uint2* __restrict__ cq,
    const uint8_t* __restrict__ diag_k, const float2* __restrict__ ad, double* __restrict__ r,
    const uint8_t* __restrict__ fixed, double dt, double* __restrict__ partials, int ldscap, int remap) {
  extern __shared__ float2 ldsq[];   // [ldscap][64], ldscap <= CAP = the longest slice of the launch
  const int lane = threadIdx.x;
  const uint32_t lane_u = threadIdx.x;
  const SliceDesc d = desc[remap > 1 ? xcd_chunk_remap(blockIdx.x, gridDim.x, 4 * remap) : blockIdx.x];
  const int s = d.s, len = d.len, clen = d.clen;
  const int64_t row = (int64_t)s * GL_WAVE + lane;
  // ---- round trip 2: everything that needs the descriptor only.  Program order = issue order: the column codes come
  // LAST, because the next round trip waits for them -- and with them, the counter being in order, for everything before
  uint2 rb[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) rb[j] = GL_STREAM(cq, d.cbase, min(j, clen - 1));
  const bool own = row < n_own;
  const int64_t rowc = own ? row : n_own - 1;   // (padding rows of the last slice read a valid address, nothing is stored)
  const double r_old = r[rowc];
  uint32_t fxv = 0;
  if (fixed) fxv = fixed[rowc];
  const int dk = diag_k[row];
  int32_t cu[CAP];
  const bool comp = CIDX && d.ok;   // wave-uniform
  int32_t wb = 0;
  if (comp) {
    wb = win_base[(int64_t)s * GL_N_WIN + (lane & (GL_N_WIN - 1))];
#pragma unroll
    for (int k = 0; k < CAP; ++k) cu[k] = (int32_t)GL_STREAM(cols16, d.base, min(k, len - 1));
  } else {
#pragma unroll
    for (int k = 0; k < CAP; ++k) cu[k] = GL_STREAM(cols, d.base, min(k, len - 1));
  }
  // ---- round trip 3: the row's neighbour values
  if (comp) {
#pragma unroll
    for (int k = 0; k < CAP; ++k) cu[k] = decode_col((uint32_t)cu[k], wb);
  }
  float2 v8[CAP];
#pragma unroll
  for (int k = 0; k < CAP; ++k) v8[k] = ad[cu[k]];
#pragma unroll
  for (int k = 0; k < CAP; ++k)
    if (k < ldscap) ldsq[k * GL_WAVE + lane] = v8[k];   // slots >= len hold a copy of the last entry, never read
  const float2 own_v = ldsq[dk * GL_WAVE + lane];
  const float ai = own_v.x, di = own_v.y;
  float q = 0.0f;
  // Row i of N(a) restricted to the cell, applied to delta: N_ii = w (4 a_i + 2 s), N_ij = w (a_i + a_j + s), s = sum of a
  // over the cell.  Padding records (shorter rows of the slice) have weight 0 and slots 0.
  auto corner = [&](float w, uint32_t slots) {
    float2 v[NV];
    float sa = ai;
#pragma unroll
    for (int m = 1; m < NV; ++m) v[m] = ldsq[(int)((slots >> (8 * m)) & 255u) * GL_WAVE + lane];
#pragma unroll
    for (int m = 1; m < NV; ++m) sa += v[m].x;
    float t = (4.0f * ai + 2.0f * sa) * di;
#pragma unroll
    for (int m = 1; m < NV; ++m) t += (ai + v[m].x + sa) * v[m].y;
    q += w * t;
  };
  for (int qq = 0; qq < clen; qq += RB) {
    if (qq > 0) {   // ---- round trip 4 (rows with more than RB incidences only)
#pragma unroll
      for (int j = 0; j < RB; ++j) rb[j] = GL_STREAM(cq, d.cbase, min(qq + j, clen - 1));
    }
#pragma unroll
    for (int j = 0; j < RB; ++j)
      if (qq + j < clen) corner(__uint_as_float(rb[j].y), rb[j].x);   // (wave-uniform)
  }
  double rr = 0.0;
  asm volatile("" : "+v"(fxv));   // the flag is looked at HERE, not where it was requested (a wait for it up there would
                                  // serialise the whole second round trip)
  if (own) {
    const double rn = fxv ? 0.0 : r_old - dt * (double)q;
    r[row] = rn;
    rr = rn * rn;
  }
  rr = wave_sum(rr);
  if (lane == 0) {
    partials[(size_t)s * 2 + 0] = rr;
    partials[(size_t)s * 2 + 1] = 0.0;
  }
}

// The assembly sweep in the same shape: one descriptor load; then the first RB incidence records, right-hand side(s), flags
// and column codes; then the row's gathers; then its entries of S (in flight while the records are worked through).  The
// diagonal accumulator lives in a register, the off-diagonal ones are bumped with ds_add_f64 (one LDS instruction per
// contribution, no read - add - write round trip): a record costs 3 LDS reads + 3 LDS adds (NV = 4) instead of 8 + 4.
// Every term is added to its accumulator in the same order as in k_rd_assemble.
template <int NV, int CAP, int RB, int CIDX, class AT>
__global__ __launch_bounds__(GL_WAVE) void k_rd_assemble_s(
    const SliceDesc* __restrict__ desc, int64_t n_own, const int32_t* __restrict__ cols,
    const uint16_t* __restrict__ cols16, const int32_t* __restrict__ win_base, const uint32_t* __restrict__ cs2,
    const double* __restrict__ cw, const uint8_t* __restrict__ diag_k, const double* __restrict__ vS,
    AT* __restrict__ vA, const double* __restrict__ c, const double* __restrict__ b, const double* __restrict__ b2,
    double* __restrict__ r_out, double* __restrict__ r2_out, double* __restrict__ dinv,
    const uint8_t* __restrict__ fixed, double two_dt, double* __restrict__ partials, int ldscap, int remap) {
  extern __shared__ double lds[];
  double* acc = lds;                             // [ldscap][64], ldscap <= CAP = the longest slice of the launch
  double* cn = lds + (size_t)ldscap * GL_WAVE;   // [ldscap][64]
  const int lane = threadIdx.x;
  const uint32_t lane_u = threadIdx.x;
  const SliceDesc d = desc[remap > 1 ? xcd_chunk_remap(blockIdx.x, gridDim.x, 4 * remap) : blockIdx.x];
  const int s = d.s, len = d.len, clen = d.clen;
  const int64_t row = (int64_t)s * GL_WAVE + lane;
  // ---- round trip 2 (program order = issue order; the column codes last, the next round trip waits for them)
  double wb_[RB];
  uint32_t sb_[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    wb_[j] = GL_STREAM(cw, d.cbase, min(j, clen - 1));
    sb_[j] = GL_STREAM(cs2, d.cbase, min(j, clen - 1));
  }
  const bool own = row < n_own;
  const int64_t rowc = own ? row : n_own - 1;
  const double b_row = b[rowc];
  double b2_row = 0.0;
  if (b2) b2_row = b2[rowc];
  uint32_t fxv = 0;
  if (fixed) fxv = fixed[rowc];
  const int dk = diag_k[row];
  int32_t cu[CAP];
  const bool comp = CIDX && d.ok;   // wave-uniform
  int32_t wb = 0;
  if (comp) {
    wb = win_base[(int64_t)s * GL_N_WIN + (lane & (GL_N_WIN - 1))];
#pragma unroll
    for (int k = 0; k < CAP; ++k) cu[k] = (int32_t)GL_STREAM(cols16, d.base, min(k, len - 1));
  } else {
#pragma unroll
    for (int k = 0; k < CAP; ++k) cu[k] = GL_STREAM(cols, d.base, min(k, len - 1));
  }
  // ---- round trip 3: the row's neighbour values of c
  if (comp) {
#pragma unroll
    for (int k = 0; k < CAP; ++k) cu[k] = decode_col((uint32_t)cu[k], wb);
  }
  {
    double x8[CAP];
#pragma unroll
    for (int k = 0; k < CAP; ++k) x8[k] = c[cu[k]];
#pragma unroll
    for (int k = 0; k < CAP; ++k)
      if (k < ldscap) {
        cn[k * GL_WAVE + lane] = x8[k];   // slots >= len hold a copy of the last entry, never read
        acc[k * GL_WAVE + lane] = 0.0;
      }
  }
  // ---- the row's entries of S: requested now, needed after the records
  double S8[CAP];
#pragma unroll
  for (int k = 0; k < CAP; ++k) S8[k] = GL_STREAM(vS, d.base, min(k, len - 1));
  const double ci = cn[dk * GL_WAVE + lane];
  const double ci4 = 4.0 * ci;
  double acc_d = 0.0;
  auto corner = [&](double w, uint32_t slots) {
    int k[NV];
    double cv[NV], st = ci;
#pragma unroll
    for (int m = 1; m < NV; ++m) {
      k[m] = (int)((slots >> (8 * m)) & 255u);
      cv[m] = cn[k[m] * GL_WAVE + lane];
    }
#pragma unroll
    for (int m = 1; m < NV; ++m) st += cv[m];
    acc_d += w * (ci4 + 2.0 * st);
#pragma unroll
    for (int m = 1; m < NV; ++m) lds_add(&acc[k[m] * GL_WAVE + lane], w * (ci + cv[m] + st));
  };
  for (int qq = 0; qq < clen; qq += RB) {
    if (qq > 0) {   // rows with more than RB incidences
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        wb_[j] = GL_STREAM(cw, d.cbase, min(qq + j, clen - 1));
        sb_[j] = GL_STREAM(cs2, d.cbase, min(qq + j, clen - 1));
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j)
      if (qq + j < clen) corner(wb_[j], sb_[j]);   // (wave-uniform; a padding record of a shorter row has weight 0 and slots 0)
  }
  // phase 3: A = S + 2 dt N(c), residual 1/2 (A + S) c, diagonal
  double r = 0.0, dg = 1.0;
#pragma unroll
  for (int k = 0; k < CAP; ++k)
    if (k < len) {
      const double a_k = (k == dk) ? acc_d : acc[k * GL_WAVE + lane];
      const double Av = S8[k] + two_dt * a_k;
      GL_STREAM(vA, d.base, k) = (AT)Av;
      r += 0.5 * (Av + S8[k]) * cn[k * GL_WAVE + lane];
      if (k == dk) dg = Av;
    }
  double rr = 0.0, rr2 = 0.0;
  asm volatile("" : "+v"(fxv));
  if (own) {
    const bool fx = fxv != 0;
    const double res = fx ? 0.0 : b_row - r;
    r_out[row] = res;
    dinv[row] = fx ? 1.0 : 1.0 / dg;
    rr = res * res;
    if (b2) {
      const double res2 = fx ? 0.0 : b2_row - r;
      r2_out[row] = res2;
      rr2 = res2 * res2;
    }
  }
  rr = wave_sum(rr);
  rr2 = wave_sum(rr2);
  if (lane == 0) {
    partials[(size_t)s * 2 + 0] = rr;
    partials[(size_t)s * 2 + 1] = rr2;
  }
}
#undef GL_STREAM

__global__ void k_make_desc(int n, const int32_t* __restrict__ list, const int64_t* __restrict__ slice_ptr,
                            const int64_t* __restrict__ cslice_ptr, const uint8_t* __restrict__ win_ok,
                            SliceDesc* __restrict__ desc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = list[i];
  SliceDesc d;
  d.base = slice_ptr[s];
  d.cbase = cslice_ptr[s];
  d.s = s;
  d.len = (int)((slice_ptr[s + 1] - d.base) >> 6);
  d.clen = (int)((cslice_ptr[s + 1] - d.cbase) >> 6);
  d.ok = win_ok[s];
  desc[i] = d;
}

// ---------------------------------------------------------------------------------------------------
// MATRIX-FREE variant of y = A(c) x, for the A/B that SURVEY 7.1 step 5 asks for (glims_apply which = 7; the solver
// does not use it): A(c) = S + 2 dt N(c) is never stored, N(c)'s row is rebuilt from the (row, cell) incidence lists
// -- exactly the sweep's phase 2 -- and applied to the gathered x on the fly.  Per row it streams the incidence
// records (12 B each, 23.7 per row on tetrahedra), S (8 B per entry) and the column codes, i.e. ~3x the bytes of the
// assembled product; measured (profiles/r02_matfree_ab.txt): 1.38 ms and 5.15 GB against 0.33 ms and 1.77 GB at 10 M rows.
// ---------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(GL_WAVE) void k_rd_matfree(
    const int32_t* __restrict__ slice_list, int64_t n_own, const int64_t* __restrict__ slice_ptr,
    const int32_t* __restrict__ cols, const int64_t* __restrict__ cslice_ptr, const uint32_t* __restrict__ cslots,
    const double* __restrict__ cw, const uint8_t* __restrict__ diag_k, const double* __restrict__ vS,
    const double* __restrict__ c, const double* __restrict__ x, double* __restrict__ y, double two_dt, int max_len) {
  extern __shared__ double lds[];
  double* acc = lds;
  double* cn = lds + (size_t)max_len * GL_WAVE;
  double* xn = lds + (size_t)2 * max_len * GL_WAVE;
  const int lane = threadIdx.x;
  const int s = slice_list[xcd_chunk_remap(blockIdx.x, gridDim.x, 4 * GL_XCD_CHUNK)];
  const int64_t row = (int64_t)s * GL_WAVE + lane;
  const int64_t base = slice_ptr[s], cbase = cslice_ptr[s];
  const int len = (int)((slice_ptr[s + 1] - base) >> 6), clen = (int)((cslice_ptr[s + 1] - cbase) >> 6);
  const int32_t* cc = cols + base + lane;
  for (int k = 0; k < len; k += 8) {
    int32_t ci8[8];
    double c8[8], x8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ci8[j] = cc[(int64_t)min(k + j, len - 1) * GL_WAVE];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c8[j] = c[ci8[j]];
      x8[j] = x[ci8[j]];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k + j < len) {
        cn[(k + j) * GL_WAVE + lane] = c8[j];
        xn[(k + j) * GL_WAVE + lane] = x8[j];
        acc[(k + j) * GL_WAVE + lane] = 0.0;
      }
  }
  const int dk = diag_k[row];
  const double ci = cn[dk * GL_WAVE + lane];
  const uint32_t* sl = cslots + cbase + lane;   // (the re-ordered slot words: the row's own slot in byte 0)
  const double* wp = cw + cbase + lane;
  const double ci4 = 4.0 * ci;
  double acc_d = 0.0;
  auto corner = [&](double w, uint32_t slots) {   // the sweep's expressions, so that the two products agree bit for bit
    int k[NV];
    double cv[NV], st = ci;
#pragma unroll
    for (int m = 1; m < NV; ++m) {
      k[m] = (int)((slots >> (8 * m)) & 255u);
      cv[m] = cn[k[m] * GL_WAVE + lane];
    }
#pragma unroll
    for (int m = 1; m < NV; ++m) st += cv[m];
    acc_d += w * (ci4 + 2.0 * st);
#pragma unroll
    for (int m = 1; m < NV; ++m) lds_add(&acc[k[m] * GL_WAVE + lane], w * (ci + cv[m] + st));
  };
  {
    int q = 0;
    for (; q + 24 <= clen; q += 24) corner_batch<24, 0>(wp, sl, q, corner);
    for (; q + 4 <= clen; q += 4) corner_batch<4, 0>(wp, sl, q, corner);
    for (; q < clen; ++q) corner_batch<1, 0>(wp, sl, q, corner);
  }
  const double* sv = vS + base + lane;
  double r = 0.0;
  for (int k = 0; k < len; k += 8) {
    double S8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) S8[j] = sv[(int64_t)min(k + j, len - 1) * GL_WAVE];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k + j < len) r += (S8[j] + two_dt * ((k + j == dk) ? acc_d : acc[(k + j) * GL_WAVE + lane])) * xn[(k + j) * GL_WAVE + lane];
  }
  if (row < n_own) y[row] = r;
}

// ---------------------------------------------------------------------------------------------------
// hot: SELL-64 SpMV  y = A x  (+ addv), optional Dirichlet row mask, optional fused dot products
//   DOTS: partials[b] = y.x over the rows of logical block b (the PCG's delta = w.u; r.u and r.r come from the
//   vector-update kernel, which has r and u in registers anyway, so this kernel never reads r)
// ---------------------------------------------------------------------------------------------------
// one lane's row of a slice: UNR independent (column, value, gather) triples in flight
// NT: values and columns are streamed exactly once per SpMV -> non-temporal, so that they do not displace the
// gathered x entries from L2 / Infinity Cache
// DK >= 0 (fused dot product): also returns x at the row's own column through xdiag (slot dk of the row), so that
// the dot y.x does not read x[row] a second time
// `rl` = the lane's own row length: slots [rl, len) are padding (value 0, column = the row itself) and are NOT read -- the
// rows of a slice are sorted by length, so the active lanes of a slot form a prefix and the memory system fetches only the
// lines they touch.  On an unstructured mesh the padding is +19.5 % of the entries (sigma = 256); its bytes used to be
// streamed like all others.
template <int COMP, int UNR, int NT, int WANT_DIAG, class VT>
__device__ __forceinline__ double spmv_row(const int32_t* __restrict__ cc, const uint16_t* __restrict__ c16,
                                            int32_t wb, const VT* __restrict__ v, const double* __restrict__ x,
                                            int len, int rl, int dk, double& xdiag) {
  constexpr bool NTC = NT == 1, NTV = NT != 0;   // NT = 2: only the 8-byte value stream is non-temporal
  // (the loops run to the SLICE's length on every lane -- the column decode reads window bases across lanes, which needs
  //  all of them active -- and the loads are predicated on the lane's own row length)
  double acc = 0.0;
  int k = 0;
  for (; k + UNR <= len; k += UNR) {
    int32_t cu[UNR];
    double vu[UNR], xu[UNR];
    if (COMP) {
      uint16_t qu[UNR];
#pragma unroll
      for (int j = 0; j < UNR; ++j)
        qu[j] = k + j < rl ? (NTC ? __builtin_nontemporal_load(c16 + (int64_t)(k + j) * GL_WAVE) : c16[(int64_t)(k + j) * GL_WAVE])
                           : (uint16_t)0;
#pragma unroll
      for (int j = 0; j < UNR; ++j)
        vu[j] = k + j < rl ? (double)(NTV ? __builtin_nontemporal_load(v + (int64_t)(k + j) * GL_WAVE) : v[(int64_t)(k + j) * GL_WAVE])
                           : 0.0;
#pragma unroll
      for (int j = 0; j < UNR; ++j) cu[j] = decode_col(qu[j], wb);
    } else {
#pragma unroll
      for (int j = 0; j < UNR; ++j)
        cu[j] = k + j < rl ? (NTC ? __builtin_nontemporal_load(cc + (int64_t)(k + j) * GL_WAVE) : cc[(int64_t)(k + j) * GL_WAVE])
                           : 0;
#pragma unroll
      for (int j = 0; j < UNR; ++j)
        vu[j] = k + j < rl ? (double)(NTV ? __builtin_nontemporal_load(v + (int64_t)(k + j) * GL_WAVE) : v[(int64_t)(k + j) * GL_WAVE])
                           : 0.0;
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) xu[j] = k + j < rl ? x[cu[j]] : 0.0;
#pragma unroll
    for (int j = 0; j < UNR; ++j) acc += vu[j] * xu[j];
    if (WANT_DIAG) {
#pragma unroll
      for (int j = 0; j < UNR; ++j) xdiag = (k + j == dk) ? xu[j] : xdiag;
    }
  }
  if (k < len) {
    // ragged tail (len is rarely a multiple of UNR: 15 on the structured 3-D meshes) as ONE more batch, so that its loads
    // are in flight together like all the others
    int32_t cu[UNR];
    double vu[UNR], xu[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
      const int kk = k + j;
      if (COMP)
        cu[j] = kk < rl ? (int32_t)(NTC ? __builtin_nontemporal_load(c16 + (int64_t)kk * GL_WAVE) : c16[(int64_t)kk * GL_WAVE]) : 0;
      else
        cu[j] = kk < rl ? (NTC ? __builtin_nontemporal_load(cc + (int64_t)kk * GL_WAVE) : cc[(int64_t)kk * GL_WAVE]) : 0;
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
      const int kk = k + j;
      vu[j] = kk < rl ? (double)(NTV ? __builtin_nontemporal_load(v + (int64_t)kk * GL_WAVE) : v[(int64_t)kk * GL_WAVE]) : 0.0;
    }
    if (COMP) {
#pragma unroll
      for (int j = 0; j < UNR; ++j) cu[j] = decode_col((uint32_t)cu[j], wb);
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) xu[j] = k + j < rl ? x[cu[j]] : 0.0;
#pragma unroll
    for (int j = 0; j < UNR; ++j) acc += vu[j] * xu[j];
    if (WANT_DIAG) {
#pragma unroll
      for (int j = 0; j < UNR; ++j) xdiag = (k + j == dk) ? xu[j] : xdiag;
    }
  }
  return acc;
}


// VT = double; float only for the optional single-precision copy of the Newton Jacobian (GLIMS_FLAG_FP32_JACOBIAN).
template <int DOTS, int UNR, int NT, int CIDX, class VT = double>
__global__ __launch_bounds__(256) void k_spmv(int n_launch, int chunk, const int32_t* __restrict__ slice_list,
                                               int64_t n_own, const int64_t* __restrict__ slice_ptr,
                                               const int32_t* __restrict__ cols, const uint16_t* __restrict__ cols16,
                                               const int32_t* __restrict__ win_base,
                                               const uint8_t* __restrict__ win_ok,
                                               const uint8_t* __restrict__ diag_k, const uint8_t* __restrict__ rlen,
                                               const VT* __restrict__ vals,
                                               const double* __restrict__ x, double* __restrict__ y,
                                               const uint8_t* __restrict__ fixed, const double* __restrict__ addv,
                                               const double* __restrict__ r, double* __restrict__ partials,
                                               int partial_off, const int* __restrict__ done, int remap) {
  if (done && *done) return;
  // block b owns the contiguous slice range [b*chunk, (b+1)*chunk); its 4 waves interleave inside that range
  const int b = remap > 1 ? xcd_chunk_remap(blockIdx.x, gridDim.x, remap) : blockIdx.x;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s_end = min(n_launch, (b + 1) * chunk);
  double pd = 0.0;
  for (int si = b * chunk + wid; si < s_end; si += 4) {
    const int s = slice_list ? slice_list[si] : si;
    const int64_t row = (int64_t)s * GL_WAVE + lane;
    const int64_t base = slice_ptr[s];
    const int len = (int)((slice_ptr[s + 1] - base) >> 6);
    const VT* v = vals + base + lane;
    double acc, xd = 0.0;
    const int dk = DOTS ? (int)diag_k[row] : -1;   // diag_k covers the padded rows of the last slice as well
    const int rl = (int)rlen[row];                 // ... and so does rlen (0 there)
    if (CIDX && win_ok[s]) {   // wave-uniform
      const int32_t wb = win_base[(int64_t)s * GL_N_WIN + (lane & (GL_N_WIN - 1))];
      acc = spmv_row<1, UNR, NT, DOTS, VT>(nullptr, cols16 + base + lane, wb, v, x, len, rl, dk, xd);
    } else {
      acc = spmv_row<0, UNR, NT, DOTS, VT>(cols + base + lane, nullptr, 0, v, x, len, rl, dk, xd);
    }
    if (row < n_own) {
      if (fixed && fixed[row]) acc = 0.0;
      if (addv) acc += addv[row];
      y[row] = acc;
      if (DOTS) pd += acc * xd;   // xd = x[row], picked up from the gather of the diagonal entry
    }
  }
  if (DOTS) spmv_dot_partial(pd, b, partials, partial_off);
}

// Dot-free Krylov iteration (round 5): ONE launch per iteration.  Step k of the Chebyshev semi-iteration for A y = b on the
// spectrum [lmin, lmax] of Dinv A, with the vector work in the operator pass's epilogue -- the row owner has (A y)_row in
// registers, and y_in[row] arrives with the gather of the diagonal entry:
//     t = b - A y_in;   d = c1 d + c2 Dinv t;   y_out = y_in + d                              (passes 1 .. last)
// The residual is recomputed from the iterate in every pass (no recurrence drift), the direction d is read and written by its
// row only, so the pass moves 40 B of vectors per row (b, Dinv, d twice, y_out) next to the operator.  No dot product, no
// reduction kernel, no all-reduce: the iteration count m follows from the interval and the wanted reduction (solver.hip,
// cheb_solve), either known to the host (m_host) or, for a step's first solve, computed on the device from the norm of the
// warm-started residual (*plan).  The LAST pass adds the correction y to the Newton iterate x and keeps a copy (ylast), so that
// a solve whose interval turns out wrong can be taken back (x -= ylast).
//   want_res = 1: one more pass (k = last) computes only t = b - A y and stores it in b -- the residual of the final iterate,
//                 which the quadratic-structure evaluation of the Newton residual builds on;
//   want_res = 0: a sweep re-evaluates the residual anyway; the last pass is a direction pass.
// shift = 1: the solve started from a non-zero guess u that is the iterate of pass 1 (y_in = u, c1 = 0): the product A u the
// warm start needs anyway IS this pass, no separate SpMV and no start kernel; every count moves by one.  nrm (pass 1 of such a
// solve): the partial sums of |t|^2 per block, from which the device chooses the iteration count.
// y_in is gathered (ghosts included), y_out written by the row owner: two buffers that change roles every launch.
template <int UNR, int NT, int CIDX, class VT>
__global__ __launch_bounds__(256) void k_cheb(int n_launch, int chunk, const int32_t* __restrict__ slice_list,
                                               int64_t n_own, const int64_t* __restrict__ slice_ptr,
                                               const int32_t* __restrict__ cols, const uint16_t* __restrict__ cols16,
                                               const int32_t* __restrict__ win_base, const uint8_t* __restrict__ win_ok,
                                               const uint8_t* __restrict__ diag_k, const uint8_t* __restrict__ rlen,
                                               const VT* __restrict__ vals, const double* __restrict__ y_in,
                                               double* __restrict__ y_out, double* __restrict__ b,
                                               const double* __restrict__ dinv, double* __restrict__ dvec,
                                               double* __restrict__ ylast, double* __restrict__ x,
                                               const uint8_t* __restrict__ fixed, double c1, double c2, int k, int m_host,
                                               const int* __restrict__ plan, int want_res, const PackMap pm, int remap,
                                               int shift, double* __restrict__ nrm) {
  const int m = plan ? *plan : m_host;
  const int last = (want_res ? m : m - 1) + shift;
  if (k > last) return;
  const bool direction = !want_res || k < m + shift, fin = k == last;
  double pn = 0.0;
  const int blk = remap > 1 ? xcd_chunk_remap(blockIdx.x, gridDim.x, remap) : blockIdx.x;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s_end = min(n_launch, (blk + 1) * chunk);
  for (int si = blk * chunk + wid; si < s_end; si += 4) {
    const int s = slice_list ? slice_list[si] : si;
    const int64_t row = (int64_t)s * GL_WAVE + lane;
    const int64_t base = slice_ptr[s];
    const int len = (int)((slice_ptr[s + 1] - base) >> 6);
    const VT* v = vals + base + lane;
    double acc, yn = 0.0;
    const int dk = (int)diag_k[row];
    const int rl = (int)rlen[row];
    if (CIDX && win_ok[s]) {   // wave-uniform
      const int32_t wb = win_base[(int64_t)s * GL_N_WIN + (lane & (GL_N_WIN - 1))];
      acc = spmv_row<1, UNR, NT, 1, VT>(nullptr, cols16 + base + lane, wb, v, y_in, len, rl, dk, yn);
    } else {
      acc = spmv_row<0, UNR, NT, 1, VT>(cols + base + lane, nullptr, 0, v, y_in, len, rl, dk, yn);
    }
    if (row >= n_own) continue;
    if (fixed && fixed[row]) acc = 0.0;   // constrained rows: b = 0 there, so every direction and the iterate stay 0
    const double t = b[row] - acc;
    pn += t * t;
    if (direction) {
      const double dn = (c1 != 0.0 ? c1 * dvec[row] : 0.0) + c2 * dinv[row] * t;
      yn += dn;
      if (!fin) {
        dvec[row] = dn;
        y_out[row] = yn;
        if (pm.ref) pack_row<1>(pm, row, &yn);
      }
    } else {
      b[row] = t;   // the residual of the final iterate (want_res)
    }
    if (fin) {
      ylast[row] = yn;
      x[row] += yn;
    }
  }
  if (nrm) spmv_dot_partial(pn, blk, nrm, 0);
}

// Pipelined block SpMV: KB block entries per batch -- all column loads, then all KB*BS*BS value loads (non-temporal:
// streamed once), then the gathers, then the FMAs; ragged tail as one clamped batch; blocks dealt to the XCDs in
// chunks like the scalar kernel (the contiguous-eighths mapping of the first version costs 15 % on the scalar SpMV).
// VT = double, or float for the single-precision copy of K_el that the inner solves of the mixed-precision elasticity
// solver stream (products and sums stay fp64).
template <int BS, int DOTS, int KB, class VT>
__global__ __launch_bounds__(256) void k_spmv_block2(int n_launch, int chunk, const int32_t* __restrict__ slice_list,
                                                      int64_t n_own, const int64_t* __restrict__ slice_ptr,
                                                      const int32_t* __restrict__ cols,
                                                      const VT* __restrict__ vals, const double* __restrict__ x,
                                                      double* __restrict__ y, const uint8_t* __restrict__ fixed,
                                                      const double* __restrict__ r, double* __restrict__ partials,
                                                      int partial_off, const int* __restrict__ done, int remap) {
  if (done && *done) return;
  constexpr int B2 = BS * BS;
  const int b = remap > 1 ? xcd_chunk_remap(blockIdx.x, gridDim.x, remap) : blockIdx.x;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s_end = min(n_launch, (b + 1) * chunk);
  double pd = 0.0;
  for (int si = b * chunk + wid; si < s_end; si += 4) {
    const int s = slice_list ? slice_list[si] : si;
    const int64_t row = (int64_t)s * GL_WAVE + lane;
    const int64_t base = slice_ptr[s];
    const int len = (int)((slice_ptr[s + 1] - base) >> 6);
    const int32_t* cc = cols + base + lane;
    const VT* vb = vals + base * B2 + lane;
    double acc[BS];
#pragma unroll
    for (int a = 0; a < BS; ++a) acc[a] = 0.0;
    for (int k = 0; k < len; k += KB) {
      int32_t cj[KB];
      double v[KB][B2], xj[KB][BS];
#pragma unroll
      for (int j = 0; j < KB; ++j) cj[j] = __builtin_nontemporal_load(cc + (int64_t)min(k + j, len - 1) * GL_WAVE);
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const VT* vk = vb + (int64_t)min(k + j, len - 1) * (GL_WAVE * B2);
#pragma unroll
        for (int e = 0; e < B2; ++e) v[j][e] = (double)__builtin_nontemporal_load(vk + e * GL_WAVE);
      }
#pragma unroll
      for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int bb = 0; bb < BS; ++bb) xj[j][bb] = (k + j < len) ? x[(int64_t)cj[j] * BS + bb] : 0.0;
#pragma unroll
      for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int a = 0; a < BS; ++a)
#pragma unroll
          for (int bb = 0; bb < BS; ++bb) acc[a] += v[j][a * BS + bb] * xj[j][bb];
    }
    if (row < n_own) {
#pragma unroll
      for (int a = 0; a < BS; ++a) {
        double v = acc[a];
        if (fixed && fixed[row * BS + a]) v = 0.0;
        y[row * BS + a] = v;
        if (DOTS) pd += v * x[row * BS + a];
      }
    }
  }
  if (DOTS) spmv_dot_partial(pd, b, partials, partial_off);
}

// Level-0 pass of the elasticity multigrid (mg.hip), in SYMMETRICALLY SCALED variables: K~ = S K S with
// S = diag(1 / sqrt(k_ii)), x~ = S^-1 x, r~ = S r.  |K~_ij| <= 1 with a unit diagonal, so the half-precision copy
// keeps every entry that matters whatever the range of cell sizes and stiffnesses (unscaled, with one global factor,
// the copy of a 1 M-point Delaunay mesh -- cell volumes over three decades, slivers -- lost most rows below the fp16
// range: 7 984 PCG iterations instead of 90).  Chebyshev on Dinv~ K~ is the same iteration as on Dinv K (similar
// matrices).  One sweep over the half- (or single-) precision copy of K~ with the
// smoother's vector work in the epilogue -- the row owner has (A x)_row in registers, so the residual, the Chebyshev
// direction and the new iterate cost no extra pass over the vectors.  Constrained dofs: rows masked here, columns see
// x = 0 there (the iterates are zero on constrained dofs by construction).  xout must not alias xin.
//   MODE 0: xout = r - A xin          MODE 1: d = c1 d + c2 Dinv (r - A xin), xout = xin + d          MODE 2: xout = Dinv A xin
template <int BS, int MODE, int KB, class VT, int CIDX, class XT>
__global__ __launch_bounds__(256) void k_mg_fine(int n_launch, const int32_t* __restrict__ slist, int pv_block0, int chunk,
                                                  int64_t n_own,
                                                  const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ cols,
                                                  const uint16_t* __restrict__ cols16,
                                                  const int32_t* __restrict__ win_base,
                                                  const VT* __restrict__ vals,
                                                  const float* __restrict__ dinv, const double* __restrict__ sc,
                                                  const uint8_t* __restrict__ fixed, const XT* __restrict__ xin,
                                                  const XT* __restrict__ r, XT* __restrict__ d,
                                                  XT* __restrict__ xout, double* __restrict__ uout, double c1,
                                                  double c2, int remap, const int* __restrict__ done,
                                                  const double* __restrict__ r_full, double* __restrict__ pv) {
  constexpr int B2 = BS * BS;
  using XN = XNode<BS, XT>;   // layout of the iterate (xin; xout of MODE 1 / 2); r, d and the MODE 0 result are packed
  if (done && *done) return;   // enqueued past the Krylov solver's convergence: nobody reads the result
  const int b = remap > 1 ? xcd_chunk_remap(blockIdx.x, gridDim.x, remap) : blockIdx.x;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s_end = min(n_launch, (b + 1) * chunk);
  double pg = 0.0, pr = 0.0;   // last pass of a cycle inside the Krylov solver: partials of (r, u) and (r, r)
  for (int q = b * chunk + wid; q < s_end; q += 4) {
    const int s = slist ? slist[q] : q;   // (partitioned runs: the interior slices while the halo travels, then the rest)
    const int64_t row = (int64_t)s * GL_WAVE + lane;
    const int64_t base = slice_ptr[s];
    const int len = (int)((slice_ptr[s + 1] - base) >> 6);
    const int32_t* cc = cols + base + lane;
    const uint16_t* c16 = cols16 + base + lane;
    const int32_t wb = CIDX ? win_base[(int64_t)s * GL_N_WIN + (lane & (GL_N_WIN - 1))] : 0;
    const VT* vb = vals + base * B2 + lane;
    double acc[BS];
#pragma unroll
    for (int a = 0; a < BS; ++a) acc[a] = 0.0;
    for (int k = 0; k < len; k += KB) {
      int32_t cj[KB];
      double v[KB][B2], xj[KB][BS];
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int64_t kk = (int64_t)min(k + j, len - 1) * GL_WAVE;
        if (CIDX) cj[j] = (int32_t)__builtin_nontemporal_load(c16 + kk);
        else cj[j] = __builtin_nontemporal_load(cc + kk);
      }
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const VT* vk = vb + (int64_t)min(k + j, len - 1) * (GL_WAVE * B2);
#pragma unroll
        for (int e = 0; e < B2; ++e) v[j][e] = (double)(float)__builtin_nontemporal_load(vk + e * GL_WAVE);
      }
      if (CIDX) {
#pragma unroll
        for (int j = 0; j < KB; ++j) cj[j] = decode_col((uint32_t)cj[j], wb);
      }
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        XN::load(xin, cj[j], xj[j]);   // slots past the row's end repeat its last entry: their products are dropped
        if (k + j >= len) {
#pragma unroll
          for (int bb = 0; bb < BS; ++bb) xj[j][bb] = 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int a = 0; a < BS; ++a)
#pragma unroll
          for (int bb = 0; bb < BS; ++bb) acc[a] += v[j][a * BS + bb] * xj[j][bb];
    }
    if (row >= n_own) continue;
    double t[BS];
#pragma unroll
    for (int a = 0; a < BS; ++a) {
      const bool fx = fixed && fixed[row * BS + a];
      if (MODE == 2) t[a] = fx ? 0.0 : acc[a];
      else t[a] = fx ? 0.0 : (double)r[row * BS + a] - acc[a];
    }
    if (MODE == 0) {   // the residual leaves the scaled variables: r - K x = S^-1 (r~ - K~ x~)
#pragma unroll
      for (int a = 0; a < BS; ++a) xout[row * BS + a] = (XT)(t[a] / sc[row * BS + a]);
    } else {
      double z[BS];
#pragma unroll
      for (int a = 0; a < BS; ++a) {
        z[a] = 0.0;
#pragma unroll
        for (int bb = 0; bb < BS; ++bb) z[a] += (double)dinv[row * B2 + a * BS + bb] * t[bb];
      }
      if (MODE == 2) {
        XN::store(xout, row, z);
      } else {
        double xo[BS], xn[BS];
        XN::load(xin, row, xo);
#pragma unroll
        for (int a = 0; a < BS; ++a) {
          const double dn = (c1 != 0.0 ? c1 * (double)d[row * BS + a] : 0.0) + c2 * z[a];
          xn[a] = xo[a] + dn;
          d[row * BS + a] = (XT)dn;
        }
        if (uout) {   // last step of the cycle: back to x = S x~, in double precision, where the Krylov solver wants it
#pragma unroll
          for (int a = 0; a < BS; ++a) {
            const double ua = sc[row * BS + a] * xn[a];
            uout[row * BS + a] = ua;
            if (pv) {
              const double ra = r_full[row * BS + a];
              pg += ra * ua;
              pr += ra * ra;
            }
          }
        } else {
          XN::store(xout, row, xn);
        }
      }
    }
  }
  if (MODE == 1 && pv) {   // (uniform over the launch) one pair per block, summed over its 4 waves in a fixed order
    __shared__ double smp[4][2];
    pg = wave_sum(pg);
    pr = wave_sum(pr);
    if (lane == 0) {
      smp[wid][0] = pg;
      smp[wid][1] = pr;
    }
    __syncthreads();
    if (threadIdx.x < 2)
      pv[(size_t)(pv_block0 + b) * 2 + threadIdx.x] = (smp[0][threadIdx.x] + smp[1][threadIdx.x]) + (smp[2][threadIdx.x] + smp[3][threadIdx.x]);
  }
}

// Symmetrically scaled copy of the block operator for the multigrid smoother: out[(i, a), (j, b)] = s_(i,a) K s_(j,b)
// in half or single precision, same block SELL-64 plane layout as the source (one wave per slice)
template <int BS, class OT>
__global__ __launch_bounds__(GL_WAVE) void k_scaled_copy(int n_slices, const int64_t* __restrict__ slice_ptr,
                                                           const int32_t* __restrict__ cols,
                                                           const double* __restrict__ vK, const double* __restrict__ sc,
                                                           OT* __restrict__ out) {
  constexpr int B2 = BS * BS;
  const int s = blockIdx.x, lane = threadIdx.x;
  const int64_t row = (int64_t)s * GL_WAVE + lane;
  const int64_t base = slice_ptr[s];
  const int len = (int)((slice_ptr[s + 1] - base) >> 6);
  double si[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) si[a] = sc[row * BS + a];   // sc covers the padded rows of the last slice (value 1)
  for (int k = 0; k < len; ++k) {
    const int64_t col = cols[base + (int64_t)k * GL_WAVE + lane];
    const int64_t e0 = (base + (int64_t)k * GL_WAVE) * B2 + lane;
#pragma unroll
    for (int a = 0; a < BS; ++a)
#pragma unroll
      for (int b = 0; b < BS; ++b)
        out[e0 + (a * BS + b) * GL_WAVE] = (OT)(float)(si[a] * vK[e0 + (a * BS + b) * GL_WAVE] * sc[col * BS + b]);
  }
}

// s = 1 / sqrt(diagonal of the constrained K_el) per dof (1 on constrained dofs and on padding), and the inverse
// diagonal blocks of the scaled operator in single precision: Dinv~ = S^-1 Dinv S^-1
template <int BS>
__global__ void k_mg_scaling(int64_t n_own, int64_t n_pad, const int64_t* __restrict__ slice_ptr,
                             const uint8_t* __restrict__ diag_k, const double* __restrict__ vKel,
                             const uint8_t* __restrict__ fixed, const double* __restrict__ dinv,
                             double* __restrict__ sc, float* __restrict__ dinv_s) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_pad) return;
  if (row >= n_own) {
#pragma unroll
    for (int a = 0; a < BS; ++a) sc[row * BS + a] = 1.0;
    return;
  }
  const int64_t sl = row >> 6;
  const int lane = (int)(row & 63);
  const double* v = vKel + (slice_ptr[sl] + (int64_t)diag_k[row] * GL_WAVE) * (BS * BS) + lane;
  double si[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) {
    const double kaa = v[(a * BS + a) * GL_WAVE];
    si[a] = (fixed && fixed[row * BS + a]) || !(kaa > 0.0) ? 1.0 : 1.0 / sqrt(kaa);
    sc[row * BS + a] = si[a];
  }
#pragma unroll
  for (int a = 0; a < BS; ++a)
#pragma unroll
    for (int b = 0; b < BS; ++b)
      dinv_s[row * BS * BS + a * BS + b] = (float)(dinv[row * BS * BS + a * BS + b] / (si[a] * si[b]));
}

// y[(row,a)] = sum_k G[(row,a),col_k] c[col_k]
template <int BS>
__global__ __launch_bounds__(256) void k_apply_G(int n_slices, int64_t n_own, const int64_t* __restrict__ slice_ptr,
                                                  const int32_t* __restrict__ cols, const double* __restrict__ vG,
                                                  const double* __restrict__ c, const double* __restrict__ addv,
                                                  double* __restrict__ y) {
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + wid;
  if (s >= n_slices) return;
  const int64_t row = (int64_t)s * GL_WAVE + lane;
  const int64_t base = slice_ptr[s];
  const int len = (int)((slice_ptr[s + 1] - base) >> 6);
  double acc[BS];
#pragma unroll
  for (int a = 0; a < BS; ++a) acc[a] = 0.0;
  for (int k = 0; k < len; ++k) {
    const double cj = c[cols[base + (int64_t)k * GL_WAVE + lane]];
    const double* v = vG + (base + (int64_t)k * GL_WAVE) * BS + lane;
#pragma unroll
    for (int a = 0; a < BS; ++a) acc[a] += v[a * GL_WAVE] * cj;
  }
  if (row < n_own)
#pragma unroll
    for (int a = 0; a < BS; ++a) y[row * BS + a] = acc[a] + (addv ? addv[row * BS + a] : 0.0);
}

}  // namespace

// ===================================================================================================
// launchers
// ===================================================================================================
template <class K>
static void set_lds(K kern, size_t bytes) {
  if (bytes > 48 * 1024)
    GL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

void gl_compute_egeo(glims_ctx* h, const double* d_xyz, const int32_t* d_cells) {
  const int GE = 1 + h->nv * h->dim;
  h->egeo.alloc((size_t)h->n_cells * GE);
  h->evol.alloc((size_t)h->n_cells);
  dvec<unsigned long long> bad;
  const unsigned long long init[2] = {0ull, ~0ull};
  bad.upload(init, 2, h->st);
  const int bs = 256;
  const unsigned grid = (unsigned)((h->n_cells + bs - 1) / bs);
  if (h->dim == 2)
    hipLaunchKernelGGL(k_egeo<2>, dim3(grid), dim3(bs), 0, h->st, h->n_cells, d_xyz, d_cells, h->egeo.p, h->evol.p, bad.p, h->cell_new2old.p);
  else
    hipLaunchKernelGGL(k_egeo<3>, dim3(grid), dim3(bs), 0, h->st, h->n_cells, d_xyz, d_cells, h->egeo.p, h->evol.p, bad.p, h->cell_new2old.p);
  GL_HIP(hipGetLastError());
  unsigned long long res[2];
  GL_HIP(hipMemcpyAsync(res, bad.p, sizeof(res), hipMemcpyDeviceToHost, h->st));
  GL_HIP(hipStreamSynchronize(h->st));
  if (res[0] != 0)
    throw glims_error(GLIMS_E_USAGE, std::to_string(res[0]) + " degenerate cell(s) (zero volume or non-finite geometry), "
                                         "the first one is cell " + std::to_string(res[1]));
}

__global__ void k_gather_u8(int64_t n, const int32_t* __restrict__ idx, const uint8_t* __restrict__ in, uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[i]];
}
__global__ void k_translate_cells(int64_t n, const int32_t* __restrict__ celem, const int32_t* __restrict__ new2old,
                                  int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = celem[i] < 0 ? -1 : new2old[celem[i]];
}
void gl_gather_u8(glims_ctx* h, int64_t n, const int32_t* idx, const uint8_t* in, uint8_t* out) {
  hipLaunchKernelGGL(k_gather_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, n, idx, in, out);
  GL_HIP(hipGetLastError());
}
void gl_translate_cells(glims_ctx* h, int64_t n, const int32_t* celem, const int32_t* new2old, int32_t* out) {
  hipLaunchKernelGGL(k_translate_cells, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, n, celem, new2old, out);
  GL_HIP(hipGetLastError());
}

__global__ void k_to_float(int64_t n, const double* __restrict__ a, float* __restrict__ b) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = (float)a[i];
}

template <int D, int NP>
static void assemble_planes(glims_ctx* h, int mode, int ca, const PlaneOut& po) {
  const DevPattern& p = h->pat;
  const size_t lds = (size_t)NP * p.max_len * GL_WAVE * sizeof(double);
  set_lds(k_assemble_static<D, NP>, lds);
  hipLaunchKernelGGL((k_assemble_static<D, NP>), dim3(p.n_slices), dim3(GL_WAVE), lds, h->st, mode, ca, h->n_own,
                     p.slice_ptr.p, p.cslice_ptr.p, p.cslots.p, p.celem.p, p.diag_k.p, h->egeo.p, h->label.p,
                     h->mat.p, h->opt.dt, po, p.max_len);
  GL_HIP(hipGetLastError());
}

template <int D>
static void assemble_static_t(glims_ctx* h, int with_mechanics) {
  DevPattern& p = h->pat;
  const size_t ne = (size_t)p.total_entries;
  h->vM.alloc(ne);
  h->vS.alloc(ne);
  h->vA.alloc(ne);
  assemble_planes<D, 2>(h, MODE_M, 0, PlaneOut{{h->vM.p, h->vS.p, nullptr}, {1, 1, 0}, {0, 0, 0}});
  GL_HIP(hipMemcpyAsync(h->vA.p, h->vS.p, ne * sizeof(double), hipMemcpyDeviceToDevice, h->st));
  if (h->jac32) {
    h->vA32.alloc(ne);
    hipLaunchKernelGGL(k_to_float, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, h->st, (int64_t)ne, h->vS.p,
                       h->vA32.p);
    GL_HIP(hipGetLastError());
  }
  p.cw.alloc((size_t)p.total_corners);
  p.cs2.alloc((size_t)p.total_corners);
  p.cq.alloc((size_t)2 * p.total_corners);
  hipLaunchKernelGGL(k_corner_weights<D>, dim3(p.n_slices), dim3(GL_WAVE), 0, h->st, p.cslice_ptr.p, p.celem.p, h->label.p,
                     h->evol.p, h->mat.p, p.diag_k.p, p.cslots.p, p.cw.p, p.cs2.p, (uint2*)p.cq.p);
  GL_HIP(hipGetLastError());
  if (with_mechanics) {
    h->vKel.alloc(ne * D * D);
    h->vG.alloc(ne * D);
    // one launch per block row of K_el (D planes), one for G: 1 + D + 1 launches instead of 2 + D * D + D
    for (int a = 0; a < D; ++a) {
      PlaneOut po{{h->vKel.p, h->vKel.p, h->vKel.p}, {D * D, D * D, D * D}, {0, 0, 0}};
      for (int b = 0; b < D; ++b) po.off[b] = (a * D + b) * GL_WAVE;
      assemble_planes<D, D>(h, MODE_KEL, a, po);
    }
    {
      PlaneOut po{{h->vG.p, h->vG.p, h->vG.p}, {D, D, D}, {0, 0, 0}};
      for (int a = 0; a < D; ++a) po.off[a] = a * GL_WAVE;
      assemble_planes<D, D>(h, MODE_G, 0, po);
    }
    h->vKel32.release();   // the single-precision copy (mixed-precision solver only) is rebuilt on demand, gl_make_kel32
  }
}

void gl_assemble_static(glims_ctx* h, int with_mechanics) {
  if (h->dim == 2)
    assemble_static_t<2>(h, with_mechanics);
  else
    assemble_static_t<3>(h, with_mechanics);
}

int gl_rd_grid(const glims_ctx* h) { return h->pat.n_slices; }

// ---- slice classes ------------------------------------------------------------------------------------------------------
// The incidence-list kernels are launched per class of slice length: their LDS footprint (16 B x length per row for the
// sweep) decides how many waves a CU holds, so the (few) long rows of an unstructured mesh must not size it for everybody;
// and the straight-line kernels unroll to a compile-time bound (16 / 20 / 24 / 32 entries; classes of longer slices take the
// looped kernels).  Built once per handle from the symbolic phase's length classes: a class of fewer than 1024 slices (less
// than half a round of waves: it costs a launch tail of one wave lifetime, ~14 us, for nothing) joins the next shorter one
// unless that would cost the shorter class a resident wave per CU.
static int sweep_waves_per_cu(int cap) { return std::max(1, std::min(16, 160 / std::max(1, cap))); }   // 160 KB LDS, cap KB per wave

static void ensure_classes(glims_ctx* h) {
  DevPattern& p = h->pat;
  if (!p.classes.empty() || p.bucket_cap.empty()) return;
  struct HostClass {
    int cap;
    std::vector<int32_t> interior, boundary;
  };
  std::vector<HostClass> hc;
  for (size_t bk = 0; bk < p.bucket_cap.size(); ++bk) {
    const int n = p.bucket_count[bk], ni = p.bucket_interior[bk];
    if (n <= 0) continue;
    std::vector<int32_t> all((size_t)n);
    GL_HIP(hipMemcpyAsync(all.data(), p.bucket_slices[bk]->p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, h->st));
    GL_HIP(hipStreamSynchronize(h->st));
    HostClass c;
    c.cap = p.bucket_cap[bk];
    c.interior.assign(all.begin(), all.begin() + ni);
    c.boundary.assign(all.begin() + ni, all.end());
    hc.push_back(std::move(c));
  }
  for (size_t i = hc.size(); i-- > 1;) {
    HostClass &hi = hc[i], &lo = hc[i - 1];
    const size_t n_hi = hi.interior.size() + hi.boundary.size(), n_lo = lo.interior.size() + lo.boundary.size();
    const bool same_kind = (hi.cap <= 32) == (lo.cap <= 32);
    const bool cheap = sweep_waves_per_cu(hi.cap) >= sweep_waves_per_cu(lo.cap) - (n_lo < 4096 ? 1 : 0);
    if (n_hi >= 1024 || !same_kind || !cheap) continue;
    lo.cap = std::max(lo.cap, hi.cap);
    lo.interior.insert(lo.interior.end(), hi.interior.begin(), hi.interior.end());
    lo.boundary.insert(lo.boundary.end(), hi.boundary.begin(), hi.boundary.end());
    hc.erase(hc.begin() + (long)i);
  }
  for (const HostClass& c : hc) {
    auto* sc = new SliceClass();
    sc->cap = c.cap;
    sc->n_interior = (int)c.interior.size();
    sc->n = (int)(c.interior.size() + c.boundary.size());
    std::vector<int32_t> all(c.interior);
    all.insert(all.end(), c.boundary.begin(), c.boundary.end());
    sc->list.upload(all, h->st);
    sc->desc.alloc((size_t)sc->n);
    hipLaunchKernelGGL(k_make_desc, dim3((unsigned)((sc->n + 255) / 256)), dim3(256), 0, h->st, sc->n, sc->list.p,
                       p.slice_ptr.p, p.cslice_ptr.p, p.win_ok.p, sc->desc.p);
    GL_HIP(hipGetLastError());
    GL_HIP(hipStreamSynchronize(h->st));   // `all` is a stack object
    p.classes.push_back(sc);
    if (getenv("GLIMS_VERBOSE"))
      fprintf(stderr, "glims slice class: %d slices (%d interior) of at most %d entries: %s kernels, %d waves per CU in the sweep\n",
              sc->n, sc->n_interior, sc->cap, sc->cap <= 32 ? "straight-line" : "looped", sweep_waves_per_cu(sc->cap));
  }
}

// what a launch of class `sc` covers for the given part of the mesh (all / interior / boundary slices)
struct ClassLaunch {
  int grid;
  const SliceDesc* desc;
  const int32_t* list;
};
static ClassLaunch class_launch(const SliceClass& sc, int part) {
  const int n_int = sc.n_interior;
  const int off = part == GL_PART_BOUNDARY ? n_int : 0;
  return {part == GL_PART_ALL ? sc.n : part == GL_PART_INTERIOR ? n_int : sc.n - n_int, sc.desc.p + off, sc.list.p + off};
}

// Assembles A(c) and the Newton right-hand side(s).  partials: [gl_rd_grid][2] = (|b - ..|^2, |b2 - ..|^2).
// Kernel configuration (measured): classes of at most 32 entries per row take the straight-line kernel
// (k_rd_assemble_s), longer ones the looped one with 24 incidence records in flight per lane; one slice per block dealt to
// the XCDs in chunks; cached (not non-temporal) streams.  AT: the Newton Jacobian is written in fp64 or
// (GLIMS_FLAG_FP32_JACOBIAN) fp32.
void gl_rd_assemble(glims_ctx* h, const double* c, const double* b, const double* b2, double* r_out, double* r2_out,
                    double* partials, int part) {
  ensure_classes(h);
  const DevPattern& p = h->pat;
  const uint8_t* fx = h->have_fixed_c ? h->fixed_c.p : nullptr;
#define GL_RD(NV, CIDX, AT, APTR)                                                                                   \
  do {                                                                                                             \
    set_lds(k_rd_assemble<NV, 0, 24, CIDX, AT>, lds);                                                              \
    hipLaunchKernelGGL((k_rd_assemble<NV, 0, 24, CIDX, AT>), dim3(cl.grid), dim3(GL_WAVE), lds, h->st, cl.list,     \
                       h->n_own, p.slice_ptr.p, p.cols.p, p.cols16.p, p.win_base.p, p.win_ok.p, p.cslice_ptr.p,     \
                       p.cs2.p, p.cw.p, p.diag_k.p, h->vS.p, APTR, c, b, b2, r_out, r2_out, h->dinv.p, fx,           \
                       2.0 * h->opt.dt, partials, cap, GL_XCD_CHUNK);                                              \
  } while (0)
#define GL_RDS3(NV, CAP, RB, CIDX, AT, APTR)                                                                        \
  do {                                                                                                             \
    set_lds(k_rd_assemble_s<NV, CAP, RB, CIDX, AT>, lds);                                                          \
    hipLaunchKernelGGL((k_rd_assemble_s<NV, CAP, RB, CIDX, AT>), dim3(cl.grid), dim3(GL_WAVE), lds, h->st, cl.desc,  \
                       h->n_own, p.cols.p, p.cols16.p, p.win_base.p, p.cs2.p, p.cw.p, p.diag_k.p, h->vS.p, APTR, c,  \
                       b, b2, r_out, r2_out, h->dinv.p, fx, 2.0 * h->opt.dt, partials, cap, GL_XCD_CHUNK);          \
  } while (0)
  // (incidence records of the first round trip: an interior row of a tetrahedral mesh with n entries has 2 n - 6 of them)
#define GL_RDS2(NV, CIDX, AT, APTR)                                                                                 \
  do {                                                                                                             \
    if (cap <= 16) GL_RDS3(NV, 16, 26, CIDX, AT, APTR);                                                            \
    else if (cap <= 20) GL_RDS3(NV, 20, 24, CIDX, AT, APTR);                                                       \
    else if (cap <= 24) GL_RDS3(NV, 24, 24, CIDX, AT, APTR);                                                       \
    else if (cap <= 32) GL_RDS3(NV, 32, 24, CIDX, AT, APTR);                                                       \
    else GL_RD(NV, CIDX, AT, APTR);                                                                                \
  } while (0)
#define GL_RDV(NV)                                                                                                  \
  do {                                                                                                             \
    if (h->jac32 && h->use_idx16) GL_RDS2(NV, 1, float, h->vA32.p);                                                \
    else if (h->jac32) GL_RDS2(NV, 0, float, h->vA32.p);                                                           \
    else if (h->use_idx16) GL_RDS2(NV, 1, double, h->vA.p);                                                        \
    else GL_RDS2(NV, 0, double, h->vA.p);                                                                          \
  } while (0)
  for (const SliceClass* sc : p.classes) {
    const ClassLaunch cl = class_launch(*sc, part);
    if (cl.grid <= 0) continue;
    const int cap = sc->cap;
    const size_t lds = (size_t)2 * cap * GL_WAVE * sizeof(double);
    if (h->nv == 3) GL_RDV(3); else GL_RDV(4);
  }
#undef GL_RDV
#undef GL_RDS2
#undef GL_RDS3
#undef GL_RD
  GL_HIP(hipGetLastError());
}

// r <- r - dt N(a) delta and the partial sums of |r|^2; same launch shape as the sweep
void gl_rd_quad(glims_ctx* h, const float* ad, double* r, double* partials, int part) {
  ensure_classes(h);
  const DevPattern& p = h->pat;
  const uint8_t* fx = h->have_fixed_c ? h->fixed_c.p : nullptr;
#define GL_RQS(NV, CAP, RB, CIDX)                                                                                    \
  hipLaunchKernelGGL((k_rd_quad_s<NV, CAP, RB, CIDX>), dim3(cl.grid), dim3(GL_WAVE), lds, h->st, cl.desc, h->n_own,   \
                     p.cols.p, p.cols16.p, p.win_base.p, (const uint2*)p.cq.p, p.diag_k.p, (const float2*)ad, r, fx,  \
                     h->opt.dt, partials, cap, GL_XCD_CHUNK)
#define GL_RQ(NV, CIDX)                                                                                             \
  do {                                                                                                             \
    if (cap <= 16) GL_RQS(NV, 16, 28, CIDX);                                                                       \
    else if (cap <= 20) GL_RQS(NV, 20, 32, CIDX);                                                                  \
    else if (cap <= 24) GL_RQS(NV, 24, 32, CIDX);                                                                  \
    else if (cap <= 32) GL_RQS(NV, 32, 32, CIDX);                                                                  \
    else {                                                                                                         \
      set_lds(k_rd_quad<NV, 24, CIDX>, lds);                                                                       \
      hipLaunchKernelGGL((k_rd_quad<NV, 24, CIDX>), dim3(cl.grid), dim3(GL_WAVE), lds, h->st, cl.list, h->n_own,     \
                         p.slice_ptr.p, p.cols.p, p.cols16.p, p.win_base.p, p.win_ok.p, p.cslice_ptr.p,            \
                         (const uint2*)p.cq.p, p.diag_k.p, (const float2*)ad, r, fx, h->opt.dt, partials, cap,      \
                         GL_XCD_CHUNK);                                                                            \
    }                                                                                                              \
  } while (0)
  for (const SliceClass* sc : p.classes) {
    const ClassLaunch cl = class_launch(*sc, part);
    if (cl.grid <= 0) continue;
    const int cap = sc->cap;
    const size_t lds = (size_t)cap * GL_WAVE * sizeof(float2);
    if (h->nv == 3) {
      if (h->use_idx16) GL_RQ(3, 1); else GL_RQ(3, 0);
    } else {
      if (h->use_idx16) GL_RQ(4, 1); else GL_RQ(4, 0);
    }
  }
#undef GL_RQ
#undef GL_RQS
  GL_HIP(hipGetLastError());
}

// Blocks per SpMV launch: every block owns a contiguous chunk of slices; with fused dots each block emits one
// partial sum, i.e. a launch fills gl_spmv_grid() slots.
// (one slice per wave: ~16x more blocks than fit on the chip, so the dispatcher balances the tail; equal-length
// persistent blocks measured 25 % slower because 2048 blocks do not fit a residency of 7 blocks/CU in one round)
int gl_spmv_grid(int n_launch) { return std::max(1, (n_launch + 3) / 4); }

// Generic entry used by the solver: slice subset + fused dot product (r != nullptr).
// Kernel configuration (measured best): 4 entries in flight per lane, non-temporal value / column
// streams, blocks dealt to the XCDs in chunks of 64, 16-bit column codes wherever a slice has them.
void gl_launch_spmv(glims_ctx* h, hipStream_t st, int n_launch, const int32_t* slice_list, const double* vals,
                    const double* x, double* y, const uint8_t* fixed, const double* addv, const double* r,
                    double* partials, int partial_off, const int* done, const float* vals32, hipEvent_t ev0,
                    hipEvent_t ev1) {
  if (n_launch <= 0) return;
  const DevPattern& p = h->pat;
  const int grid = gl_spmv_grid(n_launch);
  const int chunk = (n_launch + grid - 1) / grid;
  const int remap = slice_list ? 0 : GL_XCD_CHUNK;
  // ev0 / ev1 (glims_options.time_kernels): start / stop events attached to THIS dispatch (hipExtLaunchKernelGGL) --
  // the kernel's own timestamps, no extra packets in the queue.  hipEventRecord before and after the launch put a
  // 5-6 us idle gap on either side of every SpMV (profiles/r02_c3_timeline.txt), 14 % of the step at 1 M rows.
  // Entries in flight per lane (h->spmv_unroll): 8 on lattice meshes, 16 -- a whole typical row -- on general ones, where
  // the x gather misses the caches far more often (1 M-point Delaunay mesh: 58.5 / 52.5 / 48.7 us with 4 / 8 / 16; the
  // brain-extent box at 10 M rows: 332 / 323 / 323 us isolated, 366 / 356 / 379 inside the time steps).
  const int unr = h->spmv_unroll;
#define GL_SPMV5(DOTS, UNR, NT, CIDX, VT, VPTR)                                                                      \
  do {                                                                                                               \
    if (ev0 || ev1)                                                                                                  \
      hipExtLaunchKernelGGL((k_spmv<DOTS, UNR, NT, CIDX, VT>), dim3(grid), dim3(256), 0, st, ev0, ev1, 0, n_launch,   \
                            chunk, slice_list, h->n_own, p.slice_ptr.p, p.cols.p, p.cols16.p, p.win_base.p,          \
                            p.win_ok.p, p.diag_k.p, p.rlen.p, VPTR, x, y, fixed, addv, r, partials, partial_off,     \
                            done, remap);                                                                            \
    else                                                                                                             \
      hipLaunchKernelGGL((k_spmv<DOTS, UNR, NT, CIDX, VT>), dim3(grid), dim3(256), 0, st, n_launch, chunk, slice_list, \
                         h->n_own, p.slice_ptr.p, p.cols.p, p.cols16.p, p.win_base.p, p.win_ok.p, p.diag_k.p,        \
                         p.rlen.p, VPTR, x, y, fixed, addv, r, partials, partial_off, done, remap);                  \
  } while (0)
#define GL_SPMV4(DOTS, UNR, CIDX, VT, VPTR)                                                                          \
  do {                                                                                                               \
    if (h->stream_nt) GL_SPMV5(DOTS, UNR, 1, CIDX, VT, VPTR);                                                        \
    else GL_SPMV5(DOTS, UNR, 0, CIDX, VT, VPTR);                                                                     \
  } while (0)
#define GL_SPMV3(DOTS, CIDX)                                                                                         \
  do {                                                                                                               \
    if (vals32) GL_SPMV4(DOTS, 8, CIDX, float, vals32);                                                              \
    else if (unr >= 16) GL_SPMV4(DOTS, 16, CIDX, double, vals);                                                      \
    else GL_SPMV4(DOTS, 8, CIDX, double, vals);                                                                      \
  } while (0)
  if (r) {
    if (h->use_idx16) GL_SPMV3(1, 1); else GL_SPMV3(1, 0);
  } else {
    if (h->use_idx16) GL_SPMV3(0, 1); else GL_SPMV3(0, 0);
  }
#undef GL_SPMV3
#undef GL_SPMV4
#undef GL_SPMV5
  GL_HIP(hipGetLastError());
}

// One launch of the dot-free Krylov iteration (k_cheb) over a slice subset; same launch shape, stream policy and column-code
// choice as the SpMV it replaces.
void gl_launch_cheb(glims_ctx* h, hipStream_t st, int n_launch, const int32_t* slice_list, const double* vals,
                    const float* vals32, const double* y_in, double* y_out, double* b, const double* dinv, double* dvec,
                    double* ylast, double* x, const uint8_t* fixed, double c1, double c2, int k, int m_host, const int* plan,
                    int want_res, const PackMap& pm, hipEvent_t ev0, hipEvent_t ev1, int shift, double* nrm) {
  if (n_launch <= 0) return;
  const DevPattern& p = h->pat;
  const int grid = gl_spmv_grid(n_launch);
  const int chunk = (n_launch + grid - 1) / grid;
  const int remap = slice_list ? 0 : GL_XCD_CHUNK;
  const int unr = h->spmv_unroll;
#define GL_CH4(UNR, NT, CIDX, VT, VPTR)                                                                              \
  do {                                                                                                               \
    if (ev0 || ev1)                                                                                                  \
      hipExtLaunchKernelGGL((k_cheb<UNR, NT, CIDX, VT>), dim3(grid), dim3(256), 0, st, ev0, ev1, 0, n_launch, chunk,  \
                            slice_list, h->n_own, p.slice_ptr.p, p.cols.p, p.cols16.p, p.win_base.p, p.win_ok.p,     \
                            p.diag_k.p, p.rlen.p, VPTR, y_in, y_out, b, dinv, dvec, ylast, x, fixed, c1, c2, k,      \
                            m_host, plan, want_res, pm, remap, shift, nrm);                                          \
    else                                                                                                             \
      hipLaunchKernelGGL((k_cheb<UNR, NT, CIDX, VT>), dim3(grid), dim3(256), 0, st, n_launch, chunk, slice_list,      \
                         h->n_own, p.slice_ptr.p, p.cols.p, p.cols16.p, p.win_base.p, p.win_ok.p, p.diag_k.p,        \
                         p.rlen.p, VPTR, y_in, y_out, b, dinv, dvec, ylast, x, fixed, c1, c2, k, m_host, plan,       \
                         want_res, pm, remap, shift, nrm);                                                           \
  } while (0)
#define GL_CH3(UNR, CIDX, VT, VPTR)                                                                                  \
  do {                                                                                                               \
    if (h->stream_nt) GL_CH4(UNR, 1, CIDX, VT, VPTR);                                                                \
    else GL_CH4(UNR, 0, CIDX, VT, VPTR);                                                                             \
  } while (0)
#define GL_CH2(CIDX)                                                                                                 \
  do {                                                                                                               \
    if (vals32) GL_CH3(8, CIDX, float, vals32);                                                                      \
    else if (unr >= 16) GL_CH3(16, CIDX, double, vals);                                                              \
    else GL_CH3(8, CIDX, double, vals);                                                                              \
  } while (0)
  if (h->use_idx16) GL_CH2(1); else GL_CH2(0);
#undef GL_CH2
#undef GL_CH3
#undef GL_CH4
  GL_HIP(hipGetLastError());
}

void gl_launch_spmv_block(glims_ctx* h, hipStream_t st, int n_launch, const int32_t* slice_list, const double* x,
                          double* y, const uint8_t* fixed, const double* r, double* partials, int partial_off,
                          const int* done, bool single_precision_operator) {
  if (n_launch <= 0) return;
  const DevPattern& p = h->pat;
  const int grid = gl_spmv_grid(n_launch);
  const int chunk = (n_launch + grid - 1) / grid;
  const int remap = slice_list ? 0 : GL_XCD_CHUNK;
#define GL_BLK(BS, DOTS)                                                                                         \
  do {                                                                                                           \
    if (single_precision_operator)                                                                               \
      hipLaunchKernelGGL((k_spmv_block2<BS, DOTS, 2, float>), dim3(grid), dim3(256), 0, st, n_launch, chunk,      \
                         slice_list, h->n_own, p.slice_ptr.p, p.cols.p, h->vKel32.p, x, y, fixed, r, partials,    \
                         partial_off, done, remap);                                                          \
    else                                                                                                         \
      hipLaunchKernelGGL((k_spmv_block2<BS, DOTS, 2, double>), dim3(grid), dim3(256), 0, st, n_launch, chunk,     \
                         slice_list, h->n_own, p.slice_ptr.p, p.cols.p, h->vKel.p, x, y, fixed, r, partials,      \
                         partial_off, done, remap);                                                          \
  } while (0)
  if (h->dim == 2) {
    if (r) GL_BLK(2, 1); else GL_BLK(2, 0);
  } else {
    if (r) GL_BLK(3, 1); else GL_BLK(3, 0);
  }
#undef GL_BLK
  GL_HIP(hipGetLastError());
}

void gl_spmv_scalar(glims_ctx* h, const double* vals, const double* x, double* y, bool masked) {
  gl_launch_spmv(h, h->st, h->pat.n_slices, nullptr, vals, x, y,
                 masked && h->have_fixed_c ? h->fixed_c.p : nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                 vals == h->vA.p && h->jac32 ? h->vA32.p : nullptr);
}

void gl_spmv_block(glims_ctx* h, const double* x, double* y, bool masked) {
  gl_launch_spmv_block(h, h->st, h->pat.n_slices, nullptr, x, y,
                       masked && h->have_fixed_u ? h->fixed_u.p : nullptr, nullptr, nullptr, 0, nullptr, false);
}

// single-precision copy of K_el for the inner solves of the mixed-precision elasticity solver (built on first use)
void gl_make_kel32(glims_ctx* h) {
  if (h->vKel32.n == h->vKel.n && h->vKel32.p) return;
  const size_t nk = h->vKel.n;
  h->vKel32.alloc(nk);
  hipLaunchKernelGGL(k_to_float, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, h->st, (int64_t)nk, h->vKel.p,
                     h->vKel32.p);
  GL_HIP(hipGetLastError());
}

void gl_apply_G(glims_ctx* h, const double* c, double* y) {
  const DevPattern& p = h->pat;
  const unsigned grid = (unsigned)((p.n_slices + 3) / 4);
  const double* addv = h->have_mload ? h->mload.p : nullptr;
  if (h->dim == 2)
    hipLaunchKernelGGL(k_apply_G<2>, dim3(grid), dim3(256), 0, h->st, p.n_slices, h->n_own, p.slice_ptr.p, p.cols.p,
                       h->vG.p, c, addv, y);
  else
    hipLaunchKernelGGL(k_apply_G<3>, dim3(grid), dim3(256), 0, h->st, p.n_slices, h->n_own, p.slice_ptr.p, p.cols.p,
                       h->vG.p, c, addv, y);
  GL_HIP(hipGetLastError());
}

// (r, u) / (r, r) pairs the last level-0 pass of a cycle leaves for the Krylov solver: one per block of its launch(es)
int gl_mg_fine_blocks(glims_ctx* h, bool split) {
  const DevPattern& p = h->pat;
  if (!split) return gl_spmv_grid(p.n_slices);
  return (p.n_interior > 0 ? gl_spmv_grid(p.n_interior) : 0) + (p.n_boundary > 0 ? gl_spmv_grid(p.n_boundary) : 0);
}

// part: 0 = every slice; 1 = the interior slices (no ghost column: may run while the halo exchange of xin is in flight);
// 2 = the boundary slices
void gl_launch_mg_fine(glims_ctx* h, MgHierarchy& mg, int mode, const double* xin, const double* r, double* d,
                       double* xout, double c1, double c2, const int* done, double* uout, const double* r_full,
                       double* pv, int part) {
  const DevPattern& p = h->pat;
  const int n_launch = part == 0 ? p.n_slices : part == 1 ? p.n_interior : p.n_boundary;
  if (n_launch <= 0) return;
  const int32_t* slist = part == 0 ? nullptr : part == 1 ? p.interior_slices.p : p.boundary_slices.p;
  const int pv0 = part == 2 && p.n_interior > 0 ? gl_spmv_grid(p.n_interior) : 0;
  const int grid = gl_spmv_grid(n_launch);
  const int chunk = (n_launch + grid - 1) / grid;
  const uint8_t* fx = mg.op_fixed;
  const bool half = mg.half_smoother;
  const bool c16 = h->use_idx16 && h->stats.nnz_idx16 == h->stats.nnz_padded;   // every slice has 16-bit codes
  const bool x32 = mg.x32 && mode != 2;   // the power iteration of the set-up (mode 2) works on double vectors
  // entries in flight per lane: 2 blocks of 3 x 3 (2 x 2), or 8 scalars (an interior row of a tetrahedral mesh has 15)
#define GL_MGF4(BS, KB, MODE, VT, VPTR, CIDX, XT)                                                                    \
  hipLaunchKernelGGL((k_mg_fine<BS, MODE, KB, VT, CIDX, XT>), dim3(grid), dim3(256), 0, h->st, n_launch, slist, pv0,  \
                     chunk, h->n_own, p.slice_ptr.p, p.cols.p, p.cols16.p, p.win_base.p, VPTR, mg.dinv0.p, mg.sc.p,         \
                     fx, (const XT*)xin, (const XT*)r, (XT*)d, (XT*)xout, uout, c1, c2, GL_XCD_CHUNK, done, r_full, pv)
#define GL_MGF3(BS, KB, MODE, VT, VPTR, CIDX)                                                                        \
  do {                                                                                                               \
    if (MODE != 2 && x32) GL_MGF4(BS, KB, MODE, VT, VPTR, CIDX, float);                                              \
    else GL_MGF4(BS, KB, MODE, VT, VPTR, CIDX, double);                                                              \
  } while (0)
#define GL_MGF2(BS, KB, MODE, CIDX)                                                                                  \
  do {                                                                                                               \
    if (half) GL_MGF3(BS, KB, MODE, _Float16, (const _Float16*)mg.v16.p, CIDX);                                      \
    else GL_MGF3(BS, KB, MODE, float, mg.vK32s.p, CIDX);                                                             \
  } while (0)
#define GL_MGF(BS, KB, MODE)                                                                                         \
  do {                                                                                                               \
    if (c16) GL_MGF2(BS, KB, MODE, 1); else GL_MGF2(BS, KB, MODE, 0);                                                \
  } while (0)
#define GL_MGFM(BS, KB)                                                                                              \
  do {                                                                                                               \
    if (mode == 0) GL_MGF(BS, KB, 0); else if (mode == 1) GL_MGF(BS, KB, 1); else GL_MGF(BS, KB, 2);                 \
  } while (0)
  const bool timed = &mg == &h->mg && mode != 2 && part == 0 && h->timing(glims_ctx::TK_MGFINE);
  if (timed) h->tick(glims_ctx::TK_MGFINE);
  if (mg.bs == 1) GL_MGFM(1, 8);
  else if (mg.bs == 2) GL_MGFM(2, 2);
  else GL_MGFM(3, 2);
  if (timed) h->tick(glims_ctx::TK_MGFINE);
#undef GL_MGFM
#undef GL_MGF
#undef GL_MGF2
#undef GL_MGF3
#undef GL_MGF4
  GL_HIP(hipGetLastError());
}

// The smoother's operator of level 0: scaling vector, scaled inverse diagonal blocks, and the scaled copy of the
// hierarchy's operator (K_el, or S of the RD block) in half precision (default) or single precision
// (GLIMS_FLAG_MG_FP32_SMOOTHER).  Needs mg.op_dinv.
// exchange_scale: partitioned run whose level-0 passes see the ghosts -- their scale factors come from their owners.
void gl_make_smoother_copy(glims_ctx* h, MgHierarchy& mg, bool half, bool exchange_scale) {
  const DevPattern& p = h->pat;
  const int bs = mg.bs;
  const int64_t n_pad = std::max<int64_t>((int64_t)p.n_slices * GL_WAVE, h->n_nodes);
  mg.sc.alloc((size_t)n_pad * bs);
  mg.dinv0.alloc((size_t)h->n_own * bs * bs);
  const uint8_t* fx = mg.op_fixed;
  const unsigned g = (unsigned)((n_pad + 255) / 256);
#define GL_SCALING(BS)                                                                                               \
  hipLaunchKernelGGL(k_mg_scaling<BS>, dim3(g), dim3(256), 0, h->st, h->n_own, n_pad, p.slice_ptr.p, p.diag_k.p,       \
                     mg.op_vals, fx, mg.op_dinv, mg.sc.p, mg.dinv0.p)
  if (bs == 1) GL_SCALING(1); else if (bs == 2) GL_SCALING(2); else GL_SCALING(3);
#undef GL_SCALING
  GL_HIP(hipGetLastError());
  if (exchange_scale) gl_halo_exchange(h, mg.sc.p, bs);
  const size_t nk = (size_t)p.total_entries * bs * bs;
#define GL_COPY(BS, OT, OUT)                                                                                         \
  hipLaunchKernelGGL((k_scaled_copy<BS, OT>), dim3(p.n_slices), dim3(GL_WAVE), 0, h->st, p.n_slices, p.slice_ptr.p,   \
                     p.cols.p, mg.op_vals, mg.sc.p, OUT)
  if (half) {
    mg.v16.alloc(nk);
    mg.vK32s.release();
    _Float16* out = (_Float16*)mg.v16.p;
    if (bs == 1) GL_COPY(1, _Float16, out); else if (bs == 2) GL_COPY(2, _Float16, out); else GL_COPY(3, _Float16, out);
  } else {
    mg.vK32s.alloc(nk);
    mg.v16.release();
    if (bs == 1) GL_COPY(1, float, mg.vK32s.p); else if (bs == 2) GL_COPY(2, float, mg.vK32s.p); else GL_COPY(3, float, mg.vK32s.p);
  }
#undef GL_COPY
  GL_HIP(hipGetLastError());
}

// y = (S + 2 dt N(c)) x without the assembled Jacobian (measurement only, see k_rd_matfree)
void gl_rd_matfree(glims_ctx* h, const double* c, const double* x, double* y) {
  ensure_classes(h);
  const DevPattern& p = h->pat;
  for (const SliceClass* sc : p.classes) {
    if (sc->n <= 0) continue;
    const int cap = sc->cap;
    const size_t lds = (size_t)3 * cap * GL_WAVE * sizeof(double);
    if (h->nv == 3) {
      set_lds(k_rd_matfree<3>, lds);
      hipLaunchKernelGGL(k_rd_matfree<3>, dim3(sc->n), dim3(GL_WAVE), lds, h->st, sc->list.p, h->n_own, p.slice_ptr.p,
                         p.cols.p, p.cslice_ptr.p, p.cs2.p, p.cw.p, p.diag_k.p, h->vS.p, c, x, y, 2.0 * h->opt.dt, cap);
    } else {
      set_lds(k_rd_matfree<4>, lds);
      hipLaunchKernelGGL(k_rd_matfree<4>, dim3(sc->n), dim3(GL_WAVE), lds, h->st, sc->list.p, h->n_own, p.slice_ptr.p,
                         p.cols.p, p.cslice_ptr.p, p.cs2.p, p.cw.p, p.diag_k.p, h->vS.p, c, x, y, 2.0 * h->opt.dt, cap);
    }
  }
  GL_HIP(hipGetLastError());
}
