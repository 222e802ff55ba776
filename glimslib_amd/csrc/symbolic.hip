// Symbolic phase ON THE DEVICE: node renumbering, SELL-64 sparsity, (row, cell) incidence lists, 16-bit column codes,
// mesh metrics -- everything glims_create derives from the caller's coordinates and connectivity.
//
// What it replaces in the reference: nothing in-tree -- DOLFIN builds the dofmap and the PETSc AIJ sparsity when
// fenics.FunctionSpace / NonlinearVariationalProblem are constructed (glimslib/simulation_helpers/helper_classes.py:271-282,
// glimslib/simulation/simulation_tumor_growth.py:126).
//
// Until round 3 this ran on the host (setup_host.cpp, OpenMP): 2.4 s for the 10 M-node / 60 M-cell mesh of config C4 on
// the 16 cores a GPU job gets, next to 6.3 s for all 500 time steps.  The same algorithm as a handful of device passes:
//   1. Morton key per owned node, radix sort (rocPRIM; stable, so ties keep the caller's order)
//   2. (row, cell) incidences: one (Morton index of the vertex, cell) pair per cell vertex, radix sort by row -- cells
//      ascend inside every row because the sort is stable; row offsets by binary search
//   3. distinct neighbours per row (a thread per row keeps a small sorted list), then the SELL-C-sigma row sort: a block
//      per window of 256 rows ranks its rows by (length descending, Morton index ascending)
//   4. slice lengths -> offsets (scan of ~n / 64 numbers on the host), then a thread per row writes its sorted columns,
//      the slot of the diagonal and, per adjacent cell, the slots of the cell's vertices inside the row
//   5. 16-bit (window, offset) column codes: a wave per slice covers the slice's columns greedily with windows of 2048
//      -- every lane walks its (sorted) row with one pointer, the wave takes the minimum uncovered column as the next base
// The result is IDENTICAL, array by array, to what setup_host.cpp builds (tests/test_gpu_symbolic.py compares the two
// through the test hook GLIMS_HOST_SYMBOLIC); the host version stays in the library for exactly that purpose.
#include <cstring>

#include "glims_internal.h"

#include <rocprim/device/device_radix_sort.hpp>

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <numeric>

namespace {

constexpr int ROW_CAP = 256;   // distinct neighbours a row may have while it is being built (the product allows 150)

__device__ __forceinline__ uint64_t spread3(uint64_t x) {   // 21 bits -> every third bit
  x &= 0x1fffffULL;
  x = (x | x << 32) & 0x1f00000000ffffULL;
  x = (x | x << 16) & 0x1f0000ff0000ffULL;
  x = (x | x << 8) & 0x100f00f00f00f00fULL;
  x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
  x = (x | x << 2) & 0x1249249249249249ULL;
  return x;
}
__device__ __forceinline__ uint64_t spread2(uint64_t x) {   // 31 bits -> every second bit
  x &= 0x7fffffffULL;
  x = (x | x << 16) & 0x0000ffff0000ffffULL;
  x = (x | x << 8) & 0x00ff00ff00ff00ffULL;
  x = (x | x << 4) & 0x0f0f0f0f0f0f0f0fULL;
  x = (x | x << 2) & 0x3333333333333333ULL;
  x = (x | x << 1) & 0x5555555555555555ULL;
  return x;
}

inline unsigned gridn(int64_t n, int bs = 256) { return (unsigned)std::max<int64_t>(1, (n + bs - 1) / bs); }

// flags[0] cell vertex out of range, [1] non-finite coordinate, [2] orphaned owned node (+ its index in flags[3]),
// [4] a row with more than ROW_CAP neighbours
__global__ void k_check_cells(int64_t n, int64_t n_nodes, const int32_t* __restrict__ cells, int* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (cells[i] < 0 || cells[i] >= n_nodes)) flags[0] = 1;
}

// bounding boxes: out[0..2] / [3..5] = min / max over all nodes, [6..8] / [9..11] over the owned ones (block partials)
template <int D>
__global__ __launch_bounds__(256) void k_bbox(int64_t n_nodes, int64_t n_own, const double* __restrict__ xyz,
                                               double* __restrict__ part, int* __restrict__ flags) {
  __shared__ double sm[4][12];
  double v[12];
  for (int q = 0; q < 12; ++q) v[q] = (q / 3) % 2 == 0 ? 1e300 : -1e300;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += stride)
    for (int a = 0; a < D; ++a) {
      const double x = xyz[i * D + a];
      bad = bad || !isfinite(x);
      v[a] = fmin(v[a], x);
      v[3 + a] = fmax(v[3 + a], x);
      if (i < n_own) {
        v[6 + a] = fmin(v[6 + a], x);
        v[9 + a] = fmax(v[9 + a], x);
      }
    }
  if (bad) flags[1] = 1;
  for (int q = 0; q < 12; ++q) {
    const bool mn = (q / 3) % 2 == 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double t = __shfl_down(v[q], o, 64);
      v[q] = mn ? fmin(v[q], t) : fmax(v[q], t);
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const int q = threadIdx.x;
    const bool mn = (q / 3) % 2 == 0;
    double t = sm[0][q];
    for (int w = 1; w < 4; ++w) t = mn ? fmin(t, sm[w][q]) : fmax(t, sm[w][q]);
    part[(size_t)blockIdx.x * 12 + q] = t;
  }
}

template <int D>
__global__ void k_morton_keys(int64_t n_own, const double* __restrict__ xyz, double lo0, double lo1, double lo2,
                              double sc0, double sc1, double sc2, uint64_t* __restrict__ key, int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_own) return;
  const double lo[3] = {lo0, lo1, lo2}, sc[3] = {sc0, sc1, sc2};
  uint64_t q[3] = {0, 0, 0};
#pragma unroll
  for (int a = 0; a < D; ++a) q[a] = (uint64_t)((xyz[i * D + a] - lo[a]) * sc[a]);
  key[i] = D == 3 ? (spread3(q[0]) | spread3(q[1]) << 1 | spread3(q[2]) << 2) : (spread2(q[0]) | spread2(q[1]) << 1);
  idx[i] = (int32_t)i;
}

__global__ void k_invert(int64_t n, const int32_t* __restrict__ fwd, int32_t* __restrict__ inv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[fwd[i]] = (int32_t)i;
}

// Cells in the order of their first owner: key = smallest Morton index among the cell's owned vertices.  The set-up
// kernels reach the cells through the rows' incidence lists -- in the caller's cell order that is a scattered 16-byte read
// per (row, cell) pair plus a scattered look-up per vertex (343 + 130 GB of fabric fetches to build 3.5 GB of structures at
// 10 M nodes, profiles/r03_f_pmc_rdmg_10m.json); in this order the cells of neighbouring rows are neighbours in memory, and
// so are their geometry records and labels for the assembly of the static operators.
__global__ void k_cell_keys(int64_t n_cells, int nv, int64_t n_own, const int32_t* __restrict__ cells,
                            const int32_t* __restrict__ o2m, uint32_t* __restrict__ key, int32_t* __restrict__ val) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_cells) return;
  uint32_t k = (uint32_t)n_own;
  for (int m = 0; m < nv; ++m) {
    const int32_t v = cells[e * nv + m];
    if (v < n_own) k = min(k, (uint32_t)o2m[v]);
  }
  key[e] = k;
  val[e] = (int32_t)e;
}
__global__ void k_permute_cells(int64_t n_cells, int nv, const int32_t* __restrict__ new2old,
                                const int32_t* __restrict__ cells, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cells * nv) return;
  const int64_t e = i / nv;
  out[i] = cells[(int64_t)new2old[e] * nv + (i - e * nv)];
}

// one (row, cell) pair per cell vertex, generated in the CALLER's cell order (the stable sort by row then lists a row's
// cells in ascending caller index, as the host version does): row = Morton index of the vertex, n_own for vertices that are
// not owned rows; the cell is named by its internal index
__global__ void k_corner_pairs(int64_t n, int nv, int64_t n_own, const int32_t* __restrict__ cells,
                               const int32_t* __restrict__ o2m, const int32_t* __restrict__ c_old2new,
                               uint32_t* __restrict__ key, int32_t* __restrict__ val) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t v = cells[i];
  key[i] = v < n_own ? (uint32_t)o2m[v] : (uint32_t)n_own;
  val[i] = c_old2new[i / nv];
}

// adj_ptr[i] = first position of a key >= i in the sorted keys, i = 0 .. n_own
__global__ void k_row_offsets(int64_t n_own, int64_t n_pairs, const uint32_t* __restrict__ key, int64_t* __restrict__ adj_ptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_own) return;
  int64_t lo = 0, hi = n_pairs;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)key[mid] < i) lo = mid + 1;
    else hi = mid;
  }
  adj_ptr[i] = lo;
}

// Sorted list of a row's distinct neighbours while the row is being built.  The first LIST_LDS entries live in the
// thread's LDS column (entry k of lane l at [k][l]: conflict-free), the rest -- rows of more than LIST_LDS neighbours,
// which tetrahedral meshes do not have -- in private memory.  (Entirely in private memory, i.e. scratch, as until round 4,
// the two kernels below moved 343 + 130 GB through the fabric at 10 M nodes: every probe of the binary search and every
// shifted entry was a scratch access.)
constexpr int LIST_LDS = 48;
struct RowList {
  int32_t* lds;                       // this lane's column, stride GL_WAVE
  int32_t priv[ROW_CAP - LIST_LDS];
  __device__ __forceinline__ int32_t get(int k) const { return k < LIST_LDS ? lds[k * GL_WAVE] : priv[k - LIST_LDS]; }
  __device__ __forceinline__ void set(int k, int32_t v) {
    if (k < LIST_LDS) lds[k * GL_WAVE] = v;
    else priv[k - LIST_LDS] = v;
  }
  __device__ __forceinline__ int lower_bound(int n, int32_t v) const {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (get(mid) < v) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  }
  __device__ __forceinline__ int insert(int n, int32_t v) {   // returns the new length; n + 1 > ROW_CAP = overflow, list unchanged
    const int lo = lower_bound(n, v);
    if (lo < n && get(lo) == v) return n;
    if (n >= ROW_CAP) return n + 1;
    for (int k = n; k > lo; --k) set(k, get(k - 1));
    set(lo, v);
    return n + 1;
  }
};

// distinct neighbours (the row itself included) and adjacent cells of the row with Morton index i
__global__ __launch_bounds__(GL_WAVE) void k_row_lengths(int64_t n_own, int nv, const int64_t* __restrict__ adj_ptr,
                                                          const int32_t* __restrict__ adj, const int32_t* __restrict__ cells,
                                                          int32_t* __restrict__ len, int32_t* __restrict__ clen,
                                                          const int32_t* __restrict__ m2o, int* __restrict__ flags) {
  __shared__ int32_t sl[LIST_LDS * GL_WAVE];
  const int64_t i = (int64_t)xcd_chunk_remap(blockIdx.x, gridDim.x, GL_XCD_CHUNK) * blockDim.x + threadIdx.x;
  if (i >= n_own) return;
  RowList a;
  a.lds = sl + threadIdx.x;
  int n = 0;
  const int64_t q0 = adj_ptr[i], q1 = adj_ptr[i + 1];
  if (q1 == q0) {
    flags[2] = 1;
    atomicMin(flags + 3, m2o[i]);
  }
  for (int64_t q = q0; q < q1; ++q) {
    const int32_t* cv = cells + (int64_t)adj[q] * nv;
    for (int m = 0; m < nv; ++m) n = a.insert(min(n, ROW_CAP), cv[m]);
  }
  if (n > ROW_CAP) flags[4] = 1;
  len[i] = n;
  clen[i] = (int32_t)(q1 - q0);
}

// SELL-C-sigma row sort: inside each window of GL_SIGMA rows (Morton order) the rows are ordered by length, longest
// first, ties in Morton order (= what std::stable_sort does on the host).  new2old / old2new / row_m: final numbering.
__global__ __launch_bounds__(GL_SIGMA) void k_sigma_sort(int64_t n_own, const int32_t* __restrict__ len,
                                                          const int32_t* __restrict__ m2o, int32_t* __restrict__ new2old,
                                                          int32_t* __restrict__ old2new, int32_t* __restrict__ row_m) {
  __shared__ int32_t sl[GL_SIGMA];
  const int64_t a = (int64_t)blockIdx.x * GL_SIGMA;
  const int t = threadIdx.x;
  const int64_t i = a + t;
  const int cnt = (int)((n_own - a) < (int64_t)GL_SIGMA ? (n_own - a) : (int64_t)GL_SIGMA);
  sl[t] = t < cnt ? len[i] : -1;
  __syncthreads();
  if (t >= cnt) return;
  const int32_t me = sl[t];
  int rank = 0;
  for (int j = 0; j < cnt; ++j) rank += (sl[j] > me) || (sl[j] == me && j < t);
  const int32_t o = m2o[i];
  new2old[a + rank] = o;
  old2new[o] = (int32_t)(a + rank);
  row_m[a + rank] = (int32_t)i;
}

// per slice: longest row / longest incidence list (final numbering); sums for nnz
__global__ __launch_bounds__(256) void k_slice_lengths(int32_t n_slices, int64_t n_own, const int32_t* __restrict__ row_m,
                                                        const int32_t* __restrict__ len, const int32_t* __restrict__ clen,
                                                        int32_t* __restrict__ slen, int32_t* __restrict__ sclen,
                                                        unsigned long long* __restrict__ nnz) {
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= n_slices) return;
  const int64_t r = (int64_t)s * GL_WAVE + lane;
  int l = 0, c = 0;
  if (r < n_own) {
    l = len[row_m[r]];
    c = clen[row_m[r]];
  }
  unsigned long long sum = (unsigned long long)l;
  int lm = l, cm = c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lm = max(lm, __shfl_down(lm, o, 64));
    cm = max(cm, __shfl_down(cm, o, 64));
    sum += __shfl_down(sum, o, 64);
  }
  if (lane == 0) {
    slen[s] = lm;
    sclen[s] = cm;
    atomicAdd(nnz, sum);
  }
}

// out[i] = map[in[i]]
__global__ void k_translate_ids(int64_t n, const int32_t* __restrict__ in, const int32_t* __restrict__ map,
                                int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = map[in[i]];
}

// a thread per row of the final numbering (padding rows of the last slice included): columns, diagonal slot, incidences
// (the cells' vertex ids arrive translated to the final numbering -- ONE gather through old2new per vertex of a cell instead of
//  two per (row, cell, vertex): the per-incidence gathers through the caller's numbering were most of this kernel's 118 GB of
//  fabric reads at 10 M rows, profiles/r04_a_pmc_c4.json)
__global__ __launch_bounds__(GL_WAVE) void k_fill_pattern(int64_t n_own, int nv, const int64_t* __restrict__ slice_ptr,
                                                           const int64_t* __restrict__ cslice_ptr,
                                                           const int32_t* __restrict__ row_m,
                                                           const int64_t* __restrict__ adj_ptr,
                                                           const int32_t* __restrict__ adj,
                                                           const int32_t* __restrict__ cells /*vertex ids in the FINAL numbering*/,
                                                           int32_t* __restrict__ cols, uint8_t* __restrict__ diag_k,
                                                           uint8_t* __restrict__ rlen,
                                                           uint32_t* __restrict__ cslots, int32_t* __restrict__ celem,
                                                           uint8_t* __restrict__ is_boundary) {
  __shared__ int32_t sl[LIST_LDS * GL_WAVE];
  // (neighbouring slices on one XCD: rows that share cells share an L2)
  const int s = xcd_chunk_remap(blockIdx.x, gridDim.x, GL_XCD_CHUNK), l = threadIdx.x;
  const int64_t r = (int64_t)s * GL_WAVE + l;
  const int64_t base = slice_ptr[s], cbase = cslice_ptr[s];
  const int len = (int)((slice_ptr[s + 1] - base) >> 6), clen = (int)((cslice_ptr[s + 1] - cbase) >> 6);
  bool bnd = false;
  if (r >= n_own) {   // padding row: points at an existing column with zero value, no incidences
    for (int k = 0; k < len; ++k) cols[base + (int64_t)k * GL_WAVE + l] = 0;
    for (int q = 0; q < clen; ++q) {
      celem[cbase + (int64_t)q * GL_WAVE + l] = -1;
      cslots[cbase + (int64_t)q * GL_WAVE + l] = 0u;
    }
    diag_k[r] = 0;
    rlen[r] = 0;
  } else {
    RowList a;
    a.lds = sl + l;
    int n = 0;
    const int64_t i = row_m[r];
    const int64_t q0 = adj_ptr[i], q1 = adj_ptr[i + 1];
    for (int64_t q = q0; q < q1; ++q) {
      const int32_t* cv = cells + (int64_t)adj[q] * nv;
      for (int m = 0; m < nv; ++m) n = a.insert(n, cv[m]);
    }
    for (int k = 0; k < n; ++k) {
      const int32_t ak = a.get(k);
      cols[base + (int64_t)k * GL_WAVE + l] = ak;
      if (ak == (int32_t)r) diag_k[r] = (uint8_t)k;
      bnd = bnd || ak >= n_own;
    }
    for (int k = n; k < len; ++k) cols[base + (int64_t)k * GL_WAVE + l] = (int32_t)r;
    rlen[r] = (uint8_t)n;
    int q = 0;
    for (int64_t p = q0; p < q1; ++p, ++q) {
      const int32_t e = adj[p];
      uint32_t packed = 0;
      for (int m = 0; m < nv; ++m) {
        const int32_t vn = cells[(int64_t)e * nv + m];
        packed |= (uint32_t)a.lower_bound(n, vn) << (8 * m);
      }
      celem[cbase + (int64_t)q * GL_WAVE + l] = e;
      cslots[cbase + (int64_t)q * GL_WAVE + l] = packed;
    }
    for (; q < clen; ++q) {
      celem[cbase + (int64_t)q * GL_WAVE + l] = -1;
      cslots[cbase + (int64_t)q * GL_WAVE + l] = 0u;
    }
  }
  const unsigned long long any = __ballot(bnd);
  if (l == 0) is_boundary[s] = any != 0ull;
}

// 16-bit (window, offset) column codes, a wave per slice.  Greedy cover of the slice's sorted distinct columns by
// windows of 2^GL_WIN_BITS columns (optimal for fixed-length intervals): every lane's row is sorted, so the lane keeps
// one pointer to its first uncovered entry; a round takes the wave-wide minimum of those entries as the next window
// base and every lane advances past what the window covers.  More than `win_limit` windows: the slice keeps its
// 32-bit columns (win_ok = 0).
__global__ __launch_bounds__(GL_WAVE) void k_window_codes(const int64_t* __restrict__ slice_ptr,
                                                           const int32_t* __restrict__ cols, int win_limit,
                                                           uint16_t* __restrict__ cols16, int32_t* __restrict__ win_base,
                                                           uint8_t* __restrict__ win_ok) {
  const int s = blockIdx.x, l = threadIdx.x;
  const int64_t base = slice_ptr[s];
  const int len = (int)((slice_ptr[s + 1] - base) >> 6);
  const int32_t* row = cols + base + l;
  // the row as stored: sorted distinct columns, then padding entries equal to the row's own index (or 0 on padding
  // rows) -- padding is not sorted into the row, so it is covered separately: first the sorted part ...
  int n_sorted = len;
  // (padding entries repeat ONE value; the sorted prefix ends where an entry is not larger than its predecessor)
  for (int k = 1; k < len; ++k)
    if (row[(int64_t)k * GL_WAVE] <= row[(int64_t)(k - 1) * GL_WAVE]) {
      n_sorted = k;
      break;
    }
  const int32_t pad = n_sorted < len ? row[(int64_t)n_sorted * GL_WAVE] : -1;   // -1: none
  bool pad_open = pad >= 0;
  int k = 0;
  int32_t wb_mine = 0;    // lane w keeps base w
  int nw = 0;
  int32_t last = 0;
  for (;;) {
    int32_t cand = k < n_sorted ? row[(int64_t)k * GL_WAVE] : 0x7fffffff;
    if (pad_open) cand = min(cand, pad);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if (cand == 0x7fffffff) break;
    if (nw < GL_N_WIN && l == nw) wb_mine = cand;
    last = cand;
    ++nw;
    if (nw > GL_N_WIN) break;   // (the host stops counting there as well)
    const int64_t end = (int64_t)cand + (int64_t(1) << GL_WIN_BITS);
    while (k < n_sorted && (int64_t)row[(int64_t)k * GL_WAVE] < end) ++k;
    if (pad_open && (int64_t)pad < end) pad_open = false;
  }
  if (nw > win_limit || nw > GL_N_WIN) {
    if (l == 0) win_ok[s] = 0;
    if (l < GL_N_WIN) win_base[(int64_t)s * GL_N_WIN + l] = l < min(nw, GL_N_WIN) ? wb_mine : 0;
    for (int q = 0; q < len; ++q) cols16[base + (int64_t)q * GL_WAVE + l] = 0;
    return;
  }
  // bases of the unused windows repeat the last one; every lane gets all bases through shuffles
  const int32_t last_base = __shfl(wb_mine, nw - 1, 64);
  (void)last;
  if (l >= nw) wb_mine = last_base;
  if (l < GL_N_WIN) win_base[(int64_t)s * GL_N_WIN + l] = wb_mine;
  for (int q = 0; q < len; ++q) {
    const int32_t cj = row[(int64_t)q * GL_WAVE];
    // w = (number of bases <= cj among the first nw) - 1; the search runs the same six steps on every lane (cross-lane
    // reads need all lanes active)
    int cnt = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      const int t = cnt + step;
      const int32_t b = __shfl(wb_mine, min(t, GL_WAVE) - 1, 64);
      if (t <= nw && b <= cj) cnt = t;
    }
    const int w = max(cnt - 1, 0);
    cols16[base + (int64_t)q * GL_WAVE + l] = (uint16_t)((w << GL_WIN_BITS) | (cj - __shfl(wb_mine, w, 64)));
  }
  if (l == 0) win_ok[s] = 1;
}

// coordinates in the internal numbering (ghosts keep their place)
__global__ void k_permute_xyz(int64_t n_nodes, int d, const int32_t* __restrict__ new2old, const double* __restrict__ xyz,
                              double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const int64_t o = new2old[i];
  for (int a = 0; a < d; ++a) out[i * d + a] = xyz[o * d + a];
}

// edge statistics from the SELL pattern (owned columns, each edge once): per block the smallest positive coordinate
// difference per axis, the sum of the edge lengths and their number
template <int D>
__global__ __launch_bounds__(256) void k_edge_stats(int64_t n_own, const int64_t* __restrict__ slice_ptr,
                                                     const int32_t* __restrict__ cols, const double* __restrict__ x,
                                                     double e0, double e1, double e2, double* __restrict__ part) {
  __shared__ double sm[4][5];
  const double ext[3] = {e0, e1, e2};
  double hm[3] = {1e300, 1e300, 1e300}, es = 0.0, ec = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_own; i += stride) {
    const int64_t base = slice_ptr[i >> 6];
    const int len = (int)((slice_ptr[(i >> 6) + 1] - base) >> 6);
    for (int k = 0; k < len; ++k) {
      const int64_t j = cols[base + (int64_t)k * GL_WAVE + (i & 63)];
      if (j <= i || j >= n_own) continue;
      double e2s = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double dl = fabs(x[i * D + a] - x[j * D + a]);
        e2s += dl * dl;
        if (dl > 1e-9 * ext[a]) hm[a] = fmin(hm[a], dl);
      }
      es += sqrt(e2s);
      ec += 1.0;
    }
  }
  double v[5] = {hm[0], hm[1], hm[2], es, ec};
  for (int q = 0; q < 5; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double t = __shfl_down(v[q], o, 64);
      v[q] = q < 3 ? fmin(v[q], t) : v[q] + t;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int q = threadIdx.x;
    double t = sm[0][q];
    for (int w = 1; w < 4; ++w) t = q < 3 ? fmin(t, sm[w][q]) : t + sm[w][q];
    part[(size_t)blockIdx.x * 5 + q] = t;
  }
}

// lattice test: nodes whose coordinate on axis a is not an integer multiple of h (counted per axis)
template <int D>
__global__ void k_lattice_test(int64_t n_own, const double* __restrict__ x, double lo0, double lo1, double lo2, double h0,
                               double h1, double h2, unsigned long long* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_own) return;
  const double lo[3] = {lo0, lo1, lo2}, hh[3] = {h0, h1, h2};
#pragma unroll
  for (int a = 0; a < D; ++a) {
    if (!(hh[a] > 0.0)) continue;
    const double t = (x[i * D + a] - lo[a]) / hh[a];
    if (fabs(t - round(t)) > 1e-6) atomicAdd(bad + a, 1ull);
  }
}

template <class K, class V>
void sort_pairs(glims_ctx* h, dvec<K>& k_in, dvec<K>& k_out, dvec<V>& v_in, dvec<V>& v_out, size_t n, int end_bit) {
  size_t bytes = 0;
  GL_HIP(rocprim::radix_sort_pairs(nullptr, bytes, k_in.p, k_out.p, v_in.p, v_out.p, n, 0, end_bit, h->st));
  dvec<unsigned char> tmp;
  tmp.alloc(std::max<size_t>(bytes, 16));
  GL_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, k_in.p, k_out.p, v_in.p, v_out.p, n, 0, end_bit, h->st));
  GL_HIP(hipStreamSynchronize(h->st));   // tmp is released on return
}

}  // namespace

// Device-side glims_create: fills h->pat (device arrays), h->old2new / new2old, the counters and the mesh metrics from
// the caller's mesh already on the device (d_xyz [n_nodes][dim], d_cells [n_cells][dim + 1], caller's numbering).
void gl_build_pattern_device(glims_ctx* h, const double* d_xyz, const int32_t* d_cells, dvec<int32_t>& cells_p) {
  const bool verbose = getenv("GLIMS_VERBOSE") != nullptr;
  double t_last = omp_get_wtime();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    GL_HIP(hipStreamSynchronize(h->st));
    const double t = omp_get_wtime();
    fprintf(stderr, "glims setup (device): %-36s %7.3f s\n", what, t - t_last);
    t_last = t;
  };
  const int dim = h->dim, nv = h->nv;
  const int64_t n_nodes = h->n_nodes, n_own = h->n_own, n_cells = h->n_cells;
  GL_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  GL_REQUIRE(n_own > 0 && n_own <= n_nodes, "need 0 < n_own <= n_nodes");
  GL_REQUIRE(n_nodes < (int64_t(1) << 31) - 64, "too many nodes for 32-bit local indices");
  GL_REQUIRE(n_cells > 0 && n_cells * nv < (int64_t(1) << 31), "cell count out of range");
  DevPattern& p = h->pat;
  hipStream_t st = h->st;
  dvec<int> flags;
  {
    const int init[8] = {0, 0, 0, 0x7fffffff, 0, 0, 0, 0};
    flags.upload(init, 8, st);
  }
  // ---- validation, bounding boxes ----------------------------------------------------------------------------------
  hipLaunchKernelGGL(k_check_cells, dim3(gridn(n_cells * nv)), dim3(256), 0, st, n_cells * nv, n_nodes, d_cells, flags.p);
  const unsigned gb = (unsigned)std::min<int64_t>(1024, gridn(n_nodes));
  dvec<double> part;
  part.alloc((size_t)gb * 12);
  if (dim == 2) hipLaunchKernelGGL(k_bbox<2>, dim3(gb), dim3(256), 0, st, n_nodes, n_own, d_xyz, part.p, flags.p);
  else hipLaunchKernelGGL(k_bbox<3>, dim3(gb), dim3(256), 0, st, n_nodes, n_own, d_xyz, part.p, flags.p);
  GL_HIP(hipGetLastError());
  std::vector<double> hpart((size_t)gb * 12);
  int hflags[8];
  GL_HIP(hipMemcpyAsync(hpart.data(), part.p, hpart.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  GL_HIP(hipMemcpyAsync(hflags, flags.p, sizeof(hflags), hipMemcpyDeviceToHost, st));
  GL_HIP(hipStreamSynchronize(st));
  GL_REQUIRE(!hflags[0], "cell vertex index out of range");
  GL_REQUIRE(!hflags[1], "non-finite coordinate");
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  MeshMetrics& mm = h->mm;
  for (int a = 0; a < 3; ++a) {
    mm.lo[a] = 1e300;
    mm.hi[a] = -1e300;
  }
  for (unsigned b = 0; b < gb; ++b)
    for (int a = 0; a < dim; ++a) {
      lo[a] = std::min(lo[a], hpart[(size_t)b * 12 + a]);
      hi[a] = std::max(hi[a], hpart[(size_t)b * 12 + 3 + a]);
      mm.lo[a] = std::min(mm.lo[a], hpart[(size_t)b * 12 + 6 + a]);
      mm.hi[a] = std::max(mm.hi[a], hpart[(size_t)b * 12 + 9 + a]);
    }
  for (int a = dim; a < 3; ++a) mm.lo[a] = mm.hi[a] = 0.0;
  lap("validation, bounding boxes");

  // ---- 1. Morton order of the owned nodes ---------------------------------------------------------------------------
  const double qmax = dim == 3 ? 2097151.0 : 2147483647.0;
  double sc[3] = {0, 0, 0};
  for (int a = 0; a < dim; ++a) sc[a] = hi[a] > lo[a] ? qmax / (hi[a] - lo[a]) : 0.0;
  dvec<int32_t> m2o, o2m;   // Morton index <-> caller's index (owned nodes)
  {
    dvec<uint64_t> k_in, k_out;
    dvec<int32_t> v_in;
    k_in.alloc((size_t)n_own);
    k_out.alloc((size_t)n_own);
    v_in.alloc((size_t)n_own);
    m2o.alloc((size_t)n_own);
    if (dim == 2)
      hipLaunchKernelGGL(k_morton_keys<2>, dim3(gridn(n_own)), dim3(256), 0, st, n_own, d_xyz, lo[0], lo[1], lo[2], sc[0],
                         sc[1], sc[2], k_in.p, v_in.p);
    else
      hipLaunchKernelGGL(k_morton_keys<3>, dim3(gridn(n_own)), dim3(256), 0, st, n_own, d_xyz, lo[0], lo[1], lo[2], sc[0],
                         sc[1], sc[2], k_in.p, v_in.p);
    GL_HIP(hipGetLastError());
    sort_pairs(h, k_in, k_out, v_in, m2o, (size_t)n_own, dim == 3 ? 63 : 62);
  }
  o2m.alloc((size_t)n_own);
  hipLaunchKernelGGL(k_invert, dim3(gridn(n_own)), dim3(256), 0, st, n_own, m2o.p, o2m.p);
  GL_HIP(hipGetLastError());
  lap("Morton keys, sort");

  // ---- 1b. cells in the order of their first owner --------------------------------------------------------------------
  dvec<int32_t> c_old2new;
  {
    dvec<uint32_t> k_in, k_out;
    dvec<int32_t> v_in;
    k_in.alloc((size_t)n_cells);
    k_out.alloc((size_t)n_cells);
    v_in.alloc((size_t)n_cells);
    h->cell_new2old.alloc((size_t)n_cells);
    hipLaunchKernelGGL(k_cell_keys, dim3(gridn(n_cells)), dim3(256), 0, st, n_cells, nv, n_own, d_cells, o2m.p, k_in.p, v_in.p);
    GL_HIP(hipGetLastError());
    int cbits = 1;
    while ((int64_t(1) << cbits) <= n_own) ++cbits;
    sort_pairs(h, k_in, k_out, v_in, h->cell_new2old, (size_t)n_cells, cbits);   // stable: ties keep the caller's order
    c_old2new.alloc((size_t)n_cells);
    hipLaunchKernelGGL(k_invert, dim3(gridn(n_cells)), dim3(256), 0, st, n_cells, h->cell_new2old.p, c_old2new.p);
    cells_p.alloc((size_t)n_cells * nv);
    hipLaunchKernelGGL(k_permute_cells, dim3(gridn(n_cells * nv)), dim3(256), 0, st, n_cells, nv, h->cell_new2old.p, d_cells,
                       cells_p.p);
    GL_HIP(hipGetLastError());
  }
  lap("cells by first owner");

  // ---- 2. (row, cell) incidences sorted by row ------------------------------------------------------------------------
  const int64_t n_pairs = n_cells * nv;
  dvec<int32_t> adj;            // cells sorted by (Morton row, cell)
  dvec<int64_t> adj_ptr;        // [n_own + 1] (+ the rest: pairs of vertices that are not owned rows)
  {
    dvec<uint32_t> k_in, k_out;
    dvec<int32_t> v_in;
    k_in.alloc((size_t)n_pairs);
    k_out.alloc((size_t)n_pairs);
    v_in.alloc((size_t)n_pairs);
    adj.alloc((size_t)n_pairs);
    hipLaunchKernelGGL(k_corner_pairs, dim3(gridn(n_pairs)), dim3(256), 0, st, n_pairs, nv, n_own, d_cells, o2m.p,
                       c_old2new.p, k_in.p, v_in.p);
    GL_HIP(hipGetLastError());
    int bits = 1;
    while ((int64_t(1) << bits) <= n_own) ++bits;
    sort_pairs(h, k_in, k_out, v_in, adj, (size_t)n_pairs, bits);
    adj_ptr.alloc((size_t)n_own + 1);
    hipLaunchKernelGGL(k_row_offsets, dim3(gridn(n_own + 1)), dim3(256), 0, st, n_own, n_pairs, k_out.p, adj_ptr.p);
    GL_HIP(hipGetLastError());
    GL_HIP(hipStreamSynchronize(st));
  }
  lap("incidence pairs, sort, row offsets");

  // ---- 3. row lengths, sigma sort -> final numbering -----------------------------------------------------------------
  dvec<int32_t> len, clen, row_m, d_new2old;
  len.alloc((size_t)n_own);
  clen.alloc((size_t)n_own);
  hipLaunchKernelGGL(k_row_lengths, dim3((unsigned)((n_own + GL_WAVE - 1) / GL_WAVE)), dim3(GL_WAVE), 0, st, n_own, nv, adj_ptr.p, adj.p, cells_p.p, len.p,
                     clen.p, m2o.p, flags.p);
  GL_HIP(hipGetLastError());
  row_m.alloc((size_t)n_own);
  d_new2old.alloc((size_t)n_nodes);
  h->d_old2new.alloc((size_t)n_nodes);
  hipLaunchKernelGGL(k_sigma_sort, dim3((unsigned)((n_own + GL_SIGMA - 1) / GL_SIGMA)), dim3(GL_SIGMA), 0, st, n_own, len.p,
                     m2o.p, d_new2old.p, h->d_old2new.p, row_m.p);
  GL_HIP(hipGetLastError());
  h->new2old.resize((size_t)n_nodes);
  h->old2new.resize((size_t)n_nodes);
  GL_HIP(hipMemcpyAsync(hflags, flags.p, sizeof(hflags), hipMemcpyDeviceToHost, st));
  GL_HIP(hipMemcpyAsync(h->new2old.data(), d_new2old.p, (size_t)n_own * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GL_HIP(hipMemcpyAsync(h->old2new.data(), h->d_old2new.p, (size_t)n_own * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GL_HIP(hipStreamSynchronize(st));
  GL_REQUIRE(!hflags[2], "owned node " + std::to_string(hflags[3]) + " belongs to no cell (orphaned vertex)");
  GL_REQUIRE(!hflags[4], "a mesh node has more than " + std::to_string(ROW_CAP) + " neighbours");
  for (int64_t i = n_own; i < n_nodes; ++i) h->new2old[i] = h->old2new[i] = (int32_t)i;   // ghosts stay
  if (n_nodes > n_own) {
    GL_HIP(hipMemcpyAsync(h->d_old2new.p + n_own, h->old2new.data() + n_own, (size_t)(n_nodes - n_own) * sizeof(int32_t),
                          hipMemcpyHostToDevice, st));
    GL_HIP(hipMemcpyAsync(d_new2old.p + n_own, h->new2old.data() + n_own, (size_t)(n_nodes - n_own) * sizeof(int32_t),
                          hipMemcpyHostToDevice, st));
  }
  lap("row lengths, sigma sort, numbering to host");

  // ---- 4. slices: lengths -> offsets (host scan of n / 64 numbers), fill ------------------------------------------------
  const int32_t n_slices = (int32_t)((n_own + GL_WAVE - 1) / GL_WAVE);
  dvec<int32_t> slen, sclen;
  dvec<unsigned long long> d_nnz;
  slen.alloc((size_t)n_slices);
  sclen.alloc((size_t)n_slices);
  d_nnz.alloc_zero(1, st);
  hipLaunchKernelGGL(k_slice_lengths, dim3((unsigned)((n_slices + 3) / 4)), dim3(256), 0, st, n_slices, n_own, row_m.p,
                     len.p, clen.p, slen.p, sclen.p, d_nnz.p);
  GL_HIP(hipGetLastError());
  std::vector<int32_t> h_slen((size_t)n_slices), h_sclen((size_t)n_slices);
  unsigned long long h_nnz = 0;
  int64_t h_ncorners = 0;
  GL_HIP(hipMemcpyAsync(h_slen.data(), slen.p, h_slen.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GL_HIP(hipMemcpyAsync(h_sclen.data(), sclen.p, h_sclen.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GL_HIP(hipMemcpyAsync(&h_nnz, d_nnz.p, sizeof(h_nnz), hipMemcpyDeviceToHost, st));
  GL_HIP(hipMemcpyAsync(&h_ncorners, adj_ptr.p + n_own, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  GL_HIP(hipStreamSynchronize(st));
  std::vector<int64_t> slice_ptr((size_t)n_slices + 1, 0), cslice_ptr((size_t)n_slices + 1, 0);
  int max_len = 0, max_clen = 0;
  for (int32_t s = 0; s < n_slices; ++s) {
    slice_ptr[s + 1] = slice_ptr[s] + (int64_t)h_slen[s] * GL_WAVE;
    cslice_ptr[s + 1] = cslice_ptr[s] + (int64_t)h_sclen[s] * GL_WAVE;
    max_len = std::max(max_len, h_slen[s]);
    max_clen = std::max(max_clen, h_sclen[s]);
  }
  // slot indices are 8-bit, and the assembly sweep keeps 2 * len columns of 64 doubles in LDS (160 KB per CU)
  GL_REQUIRE(max_len <= 150, "a mesh node has " + std::to_string(max_len) +
                                 " neighbours; rows longer than 150 do not fit the LDS-resident assembly");
  GL_REQUIRE(slice_ptr[n_slices] * (int64_t)(dim * dim) < (int64_t(1) << 40), "operator too large");
  p.n_slices = n_slices;
  p.max_len = max_len;
  p.max_clen = max_clen;
  p.total_entries = slice_ptr[n_slices];
  p.total_corners = cslice_ptr[n_slices];
  h->nnz = (int64_t)h_nnz;
  h->n_corners = h_ncorners;
  p.slice_ptr.upload(slice_ptr, st);
  p.cslice_ptr.upload(cslice_ptr, st);
  p.cols.alloc((size_t)p.total_entries);
  p.diag_k.alloc((size_t)n_slices * GL_WAVE);
  p.rlen.alloc((size_t)n_slices * GL_WAVE);
  p.cslots.alloc((size_t)p.total_corners);
  p.celem.alloc((size_t)p.total_corners);
  dvec<uint8_t> is_boundary;
  is_boundary.alloc((size_t)n_slices);
  {
    dvec<int32_t> cells_n;   // the cells (internal order) with vertex ids in the final numbering
    cells_n.alloc((size_t)n_cells * nv);
    hipLaunchKernelGGL(k_translate_ids, dim3(gridn(n_cells * nv)), dim3(256), 0, st, n_cells * nv, cells_p.p,
                       h->d_old2new.p, cells_n.p);
    hipLaunchKernelGGL(k_fill_pattern, dim3(n_slices), dim3(GL_WAVE), 0, st, n_own, nv, p.slice_ptr.p, p.cslice_ptr.p, row_m.p,
                       adj_ptr.p, adj.p, cells_n.p, p.cols.p, p.diag_k.p, p.rlen.p, p.cslots.p, p.celem.p, is_boundary.p);
    GL_HIP(hipGetLastError());
    GL_HIP(hipStreamSynchronize(st));   // cells_n is released here
  }
  lap("slice offsets, columns + incidences");

  // ---- 5. 16-bit column codes ---------------------------------------------------------------------------------------
  int win_limit = GL_N_WIN;   // GLIMS_WIN_LIMIT < 32 (test hook, include/glims_hip.h) forces slices onto the 32-bit fallback
  if (const char* e = getenv("GLIMS_WIN_LIMIT")) win_limit = std::max(0, std::min(GL_N_WIN, atoi(e)));
  p.cols16.alloc((size_t)p.total_entries);
  p.win_base.alloc((size_t)n_slices * GL_N_WIN);
  p.win_ok.alloc((size_t)n_slices);
  hipLaunchKernelGGL(k_window_codes, dim3(n_slices), dim3(GL_WAVE), 0, st, p.slice_ptr.p, p.cols.p, win_limit, p.cols16.p,
                     p.win_base.p, p.win_ok.p);
  GL_HIP(hipGetLastError());
  std::vector<uint8_t> h_bnd((size_t)n_slices), h_ok((size_t)n_slices);
  GL_HIP(hipMemcpyAsync(h_bnd.data(), is_boundary.p, h_bnd.size(), hipMemcpyDeviceToHost, st));
  GL_HIP(hipMemcpyAsync(h_ok.data(), p.win_ok.p, h_ok.size(), hipMemcpyDeviceToHost, st));
  GL_HIP(hipStreamSynchronize(st));
  lap("16-bit column codes");

  // ---- slice lists (host, n / 64 entries): interior / boundary, length classes of the assembly sweep --------------------
  std::vector<int32_t> interior, boundary;
  for (int32_t s = 0; s < n_slices; ++s) (h_bnd[s] ? boundary : interior).push_back(s);
  p.interior_slices.upload(interior, st);
  p.boundary_slices.upload(boundary, st);
  p.n_interior = (int32_t)interior.size();
  p.n_boundary = (int32_t)boundary.size();
  h->nnz_idx16_avail = 0;
  for (int32_t s = 0; s < n_slices; ++s)
    if (h_ok[s]) h->nnz_idx16_avail += slice_ptr[s + 1] - slice_ptr[s];
  const std::vector<int> caps = {16, 20, 24, 32, 48, 64, 96, 128, 255};
  std::vector<std::vector<int32_t>> bucket(caps.size());
  std::vector<int32_t> bucket_interior(caps.size(), 0);
  for (int pass = 0; pass < 2; ++pass)
    for (int32_t s = 0; s < n_slices; ++s) {
      if ((int)h_bnd[s] != pass) continue;
      size_t b = 0;
      while (caps[b] < h_slen[s]) ++b;
      bucket[b].push_back(s);
      if (pass == 0) bucket_interior[b]++;
    }
  for (size_t b = 0; b < caps.size(); ++b) {
    if (bucket[b].empty()) continue;
    // LDS of the class = its actual longest slice, not the class bound (16 rows x 1 KB would be exactly 1/10 of the CU's
    // LDS and fit only 9 times; the structured meshes' 15 fits 10 times)
    int cap = 1;
    for (int32_t sl : bucket[b]) cap = std::max(cap, (int)h_slen[sl]);
    p.bucket_cap.push_back(cap);
    p.bucket_count.push_back((int32_t)bucket[b].size());
    p.bucket_interior.push_back(bucket_interior[b]);
    auto* dv = new dvec<int32_t>();
    dv->upload(bucket[b], st);
    p.bucket_slices.push_back(dv);
  }

  // ---- mesh metrics: coordinates in the internal numbering, edge statistics, lattice test -------------------------------
  h->xyz_new.alloc((size_t)n_nodes * dim);
  hipLaunchKernelGGL(k_permute_xyz, dim3(gridn(n_nodes)), dim3(256), 0, st, n_nodes, dim, d_new2old.p, d_xyz, h->xyz_new.p);
  GL_HIP(hipGetLastError());
  double ext[3];
  for (int a = 0; a < 3; ++a) ext[a] = std::max(mm.hi[a] - mm.lo[a], 1e-300);
  const unsigned ge = (unsigned)std::min<int64_t>(1024, gridn(n_own));
  dvec<double> epart;
  epart.alloc((size_t)ge * 5);
  if (dim == 2)
    hipLaunchKernelGGL(k_edge_stats<2>, dim3(ge), dim3(256), 0, st, n_own, p.slice_ptr.p, p.cols.p, h->xyz_new.p, ext[0], ext[1],
                       ext[2], epart.p);
  else
    hipLaunchKernelGGL(k_edge_stats<3>, dim3(ge), dim3(256), 0, st, n_own, p.slice_ptr.p, p.cols.p, h->xyz_new.p, ext[0], ext[1],
                       ext[2], epart.p);
  GL_HIP(hipGetLastError());
  std::vector<double> hep((size_t)ge * 5);
  GL_HIP(hipMemcpyAsync(hep.data(), epart.p, hep.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  GL_HIP(hipStreamSynchronize(st));
  double hmin[3] = {1e300, 1e300, 1e300}, esum = 0.0, ecnt = 0.0;
  for (unsigned b = 0; b < ge; ++b) {
    for (int a = 0; a < 3; ++a) hmin[a] = std::min(hmin[a], hep[(size_t)b * 5 + a]);
    esum += hep[(size_t)b * 5 + 3];
    ecnt += hep[(size_t)b * 5 + 4];
  }
  mm.mean_edge = ecnt > 0.0 ? esum / ecnt : ext[0];
  // lattice test: every coordinate an integer multiple of the axis' smallest edge component (box meshes)
  bool lat = true;
  double hl[3] = {0, 0, 0};
  for (int a = 0; a < dim; ++a) {
    if (!(hmin[a] < 1e299) || (mm.hi[a] - mm.lo[a]) / hmin[a] > 1e5) lat = false;
    hl[a] = hmin[a] < 1e299 ? hmin[a] : 0.0;
  }
  if (lat) {
    dvec<unsigned long long> bad;
    bad.alloc_zero(3, st);
    if (dim == 2)
      hipLaunchKernelGGL(k_lattice_test<2>, dim3(gridn(n_own)), dim3(256), 0, st, n_own, h->xyz_new.p, mm.lo[0], mm.lo[1],
                         mm.lo[2], hl[0], hl[1], hl[2], bad.p);
    else
      hipLaunchKernelGGL(k_lattice_test<3>, dim3(gridn(n_own)), dim3(256), 0, st, n_own, h->xyz_new.p, mm.lo[0], mm.lo[1],
                         mm.lo[2], hl[0], hl[1], hl[2], bad.p);
    GL_HIP(hipGetLastError());
    unsigned long long hb[3] = {0, 0, 0};
    GL_HIP(hipMemcpyAsync(hb, bad.p, sizeof(hb), hipMemcpyDeviceToHost, st));
    GL_HIP(hipStreamSynchronize(st));
    for (int a = 0; a < dim; ++a) lat = lat && hb[a] == 0;
  }
  // (the host version stops at the first axis that fails and leaves the later h_lattice entries unset; the values are
  //  only used when the mesh IS a lattice)
  for (int a = 0; a < 3; ++a) mm.h_lattice[a] = a < dim ? hl[a] : 0.0;
  mm.lattice = lat;
  GL_HIP(hipStreamSynchronize(st));
  lap("slice lists, mesh metrics");
}

// ---- helpers shared with the multigrid set-up (mg.hip): rocPRIM stays in this translation unit --------------------------
void gl_sort_pairs_u32(glims_ctx* h, uint32_t* k_in, uint32_t* k_out, int32_t* v_in, int32_t* v_out, size_t n, int end_bit) {
  size_t bytes = 0;
  GL_HIP(rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, n, 0, end_bit, h->st));
  dvec<unsigned char> tmp;
  tmp.alloc(std::max<size_t>(bytes, 16));
  GL_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, k_in, k_out, v_in, v_out, n, 0, end_bit, h->st));
  GL_HIP(hipStreamSynchronize(h->st));
}

namespace {
__global__ void k_offsets32(int64_t n_keys, int64_t n, const uint32_t* __restrict__ key, int32_t* __restrict__ ptr) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > n_keys) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)key[mid] < c) lo = mid + 1;
    else hi = mid;
  }
  ptr[c] = (int32_t)lo;
}
}  // namespace

// ptr[c] = first position of a key >= c in the sorted keys, c = 0 .. n_keys
void gl_offsets_of_sorted_keys(glims_ctx* h, const uint32_t* keys_sorted, int64_t n, int64_t n_keys, int32_t* ptr) {
  hipLaunchKernelGGL(k_offsets32, dim3(gridn(n_keys + 1)), dim3(256), 0, h->st, n_keys, n, keys_sorted, ptr);
  GL_HIP(hipGetLastError());
}
