// Symbolic phase (host, OpenMP): node renumbering, SELL-64 sparsity and (row, cell) incidence lists.
//
// What it replaces in the reference: nothing in-tree -- DOLFIN builds the dofmap and the PETSc AIJ sparsity
// when fenics.FunctionSpace / NonlinearVariationalProblem are constructed
// (glimslib/simulation_helpers/helper_classes.py:271-282, glimslib/simulation/simulation_tumor_growth.py:126).
//
// Layout decisions (MI355X):
//   * rows are renumbered along a Morton curve of their coordinates so that the x-gather of a 64-row slice
//     stays inside a compact neighbourhood (L1/L2 hits), then sorted by length inside windows of GL_SIGMA (256) rows
//     (SELL-C-sigma) so that a slice is padded only to the longest of 64 similar rows -- small windows on purpose:
//     a large window scatters its rows over its slices and destroys the gather locality (tools/sigma_sweep.py);
//   * a slice is 64 rows = one wavefront, entries stored slot-major ([slot][lane]) so that lane l of a wave
//     reads address base + slot*64 + l: every value/column stream is a unit-stride 512 B / 256 B access;
//   * the (row, cell) incidences ("corners") use the same slot-major layout.  A corner stores, for each vertex
//     of its cell, the slot of that vertex inside the row -- everything an element contribution needs to be
//     added to its row without atomics, and without reading the cell's connectivity again.
#include "glims_internal.h"

#include <omp.h>
#include <sched.h>
#include <cstdio>
#include <parallel/algorithm>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <cstdlib>

namespace {

inline uint64_t spread3(uint64_t x) {   // 21 bits -> every third bit
  x &= 0x1fffffULL;
  x = (x | x << 32) & 0x1f00000000ffffULL;
  x = (x | x << 16) & 0x1f0000ff0000ffULL;
  x = (x | x << 8) & 0x100f00f00f00f00fULL;
  x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
  x = (x | x << 2) & 0x1249249249249249ULL;
  return x;
}
inline uint64_t spread2(uint64_t x) {   // 31 bits -> every second bit
  x &= 0x7fffffffULL;
  x = (x | x << 16) & 0x0000ffff0000ffffULL;
  x = (x | x << 8) & 0x00ff00ff00ff00ffULL;
  x = (x | x << 4) & 0x0f0f0f0f0f0f0f0fULL;
  x = (x | x << 2) & 0x3333333333333333ULL;
  x = (x | x << 1) & 0x5555555555555555ULL;
  return x;
}

struct KeyIdx {
  uint64_t key;
  int32_t idx;
  bool operator<(const KeyIdx& o) const { return key < o.key || (key == o.key && idx < o.idx); }
};

}  // namespace

// OpenMP team for the symbolic phase: the hardware threads the process may really use.  A container often reports all
// threads of the host while its CPU quota is a fraction of them; oversubscribed teams are slower, not faster (C4 on a
// 16-core share of a 128-thread host: 3.4 s with 128 threads, 2.3 s with 16).  GLIMS_HOST_THREADS overrides.
int gl_host_threads() {
  if (const char* e = getenv("GLIMS_HOST_THREADS")) return std::max(1, atoi(e));
  long n = omp_get_num_procs();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<long>(n, std::max(1, CPU_COUNT(&set)));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
    char q[64];
    long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) {
      const long quota = atol(q);
      if (quota > 0) n = std::min(n, std::max(1l, (quota + period - 1) / period));
    }
    fclose(f);
  }
  // one process per GPU under torchrun / mpirun: the ranks of this host share the cores (torchrun's default
  // OMP_NUM_THREADS=1 is deliberately not taken as the limit here -- the symbolic phase is a one-off)
  for (const char* v : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS"})
    if (const char* e = getenv(v)) {
      n = std::max(1l, n / std::max(1, atoi(e)));
      break;
    }
  return (int)std::min(n, 64l);
}

void build_host_pattern(HostPattern& hp, int dim, int64_t n_nodes, int64_t n_own, int64_t n_cells,
                        const double* xyz, const int32_t* cells) {
  const bool verbose = getenv("GLIMS_VERBOSE") != nullptr;
  const int saved_threads = omp_get_max_threads();
  omp_set_num_threads(gl_host_threads());
  struct Restore {
    int n;
    ~Restore() { omp_set_num_threads(n); }
  } restore{saved_threads};
  if (verbose) fprintf(stderr, "glims setup: OpenMP team of %d (runtime default %d)\n", omp_get_max_threads(), saved_threads);
  double t_last = omp_get_wtime();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    const double t = omp_get_wtime();
    fprintf(stderr, "glims setup: %-44s %7.3f s\n", what, t - t_last);
    t_last = t;
  };
  GL_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  GL_REQUIRE(n_own > 0 && n_own <= n_nodes, "need 0 < n_own <= n_nodes");
  GL_REQUIRE(n_nodes < (int64_t(1) << 31) - 64, "too many nodes for 32-bit local indices");
  GL_REQUIRE(n_cells > 0 && n_cells * (dim + 1) < (int64_t(1) << 31), "cell count out of range");
  const int nv = dim + 1;
  hp.dim = dim;
  hp.nv = nv;
  hp.n_nodes = n_nodes;
  hp.n_own = n_own;
  hp.n_cells = n_cells;

  for (int64_t i = 0; i < n_cells * nv; ++i)
    GL_REQUIRE(cells[i] >= 0 && cells[i] < n_nodes, "cell vertex index out of range");

  lap("validate connectivity");
  // ---- 1. node -> cell adjacency of owned nodes (cells ascending inside each list) ------------------
  // (serial on purpose: 24 M scattered increments take 0.45 s at C4; the same loops with OpenMP atomics on the
  //  box's 128 hardware threads took 1.0 s)
  std::vector<int64_t> adj_ptr(n_own + 1, 0);
  for (int64_t e = 0; e < n_cells; ++e)
    for (int m = 0; m < nv; ++m) {
      const int32_t v = cells[e * nv + m];
      if (v < n_own) adj_ptr[v + 1]++;
    }
  for (int64_t i = 0; i < n_own; ++i) adj_ptr[i + 1] += adj_ptr[i];
  const int64_t n_corners = adj_ptr[n_own];
  std::vector<int32_t> adj(n_corners);
  {
    std::vector<int64_t> fill(adj_ptr.begin(), adj_ptr.end() - 1);
    for (int64_t e = 0; e < n_cells; ++e)
      for (int m = 0; m < nv; ++m) {
        const int32_t v = cells[e * nv + m];
        if (v < n_own) adj[fill[v]++] = (int32_t)e;
      }
  }
  for (int64_t i = 0; i < n_own; ++i)
    GL_REQUIRE(adj_ptr[i + 1] > adj_ptr[i], "owned node " + std::to_string(i) + " belongs to no cell (orphaned vertex)");

  lap("node -> cell adjacency");
  // ---- 2. neighbour lists (old numbering, sorted, diagonal included) ---------------------------------
  // one pass: every thread collects the lists of a contiguous range of rows into its own arena; the global
  // offsets follow from a prefix sum of the lengths, then the arenas are copied into place
  std::vector<int64_t> nbr_ptr(n_own + 1, 0);
  int64_t max_adj = 0;
  for (int64_t i = 0; i < n_own; ++i) max_adj = std::max(max_adj, adj_ptr[i + 1] - adj_ptr[i]);
  std::vector<int32_t> nbr;
  {
    const int nt = omp_get_max_threads();
    std::vector<std::vector<int32_t>> arena(nt);
    std::vector<int64_t> lo_row(nt + 1, 0);
#pragma omp parallel num_threads(nt)
    {
      const int t = omp_get_thread_num(), nth = omp_get_num_threads();
      const int64_t r0 = n_own * t / nth, r1 = n_own * (t + 1) / nth;
      lo_row[t] = r0;
      if (t == nth - 1)
        for (int q = nth; q <= nt; ++q) lo_row[q] = n_own;
      std::vector<int32_t>& out = arena[t];
      out.reserve((size_t)((adj_ptr[r1] - adj_ptr[r0]) * 3 / 4 + 64));
      std::vector<int32_t> buf(max_adj * nv + 4);
      for (int64_t i = r0; i < r1; ++i) {
        int cnt = 0;
        for (int64_t q = adj_ptr[i]; q < adj_ptr[i + 1]; ++q) {
          const int32_t* cv = cells + (int64_t)adj[q] * nv;
          for (int m = 0; m < nv; ++m) buf[cnt++] = cv[m];
        }
        std::sort(buf.begin(), buf.begin() + cnt);
        const int len = (int)(std::unique(buf.begin(), buf.begin() + cnt) - buf.begin());
        nbr_ptr[i + 1] = len;
        out.insert(out.end(), buf.begin(), buf.begin() + len);
      }
    }
    for (int64_t i = 0; i < n_own; ++i) nbr_ptr[i + 1] += nbr_ptr[i];
    nbr.resize(nbr_ptr[n_own]);
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < nt; ++t)
      if (!arena[t].empty())
        std::memcpy(nbr.data() + nbr_ptr[lo_row[t]], arena[t].data(), sizeof(int32_t) * arena[t].size());
  }
  hp.nnz = nbr_ptr[n_own];
  hp.n_corners = n_corners;

  lap("neighbour lists");
  // ---- 3. renumbering: Morton order, then length sort inside sigma windows ---------------------------
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = 0; i < n_nodes; ++i)
    for (int a = 0; a < dim; ++a) {
      double v = xyz[i * dim + a];
      GL_REQUIRE(std::isfinite(v), "non-finite coordinate");
      lo[a] = std::min(lo[a], v);
      hi[a] = std::max(hi[a], v);
    }
  const double qmax = dim == 3 ? 2097151.0 : 2147483647.0;
  double sc[3];
  for (int a = 0; a < dim; ++a) sc[a] = hi[a] > lo[a] ? qmax / (hi[a] - lo[a]) : 0.0;
  std::vector<KeyIdx> keys(n_own);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_own; ++i) {
    uint64_t q[3] = {0, 0, 0};
    for (int a = 0; a < dim; ++a) q[a] = (uint64_t)((xyz[i * dim + a] - lo[a]) * sc[a]);
    keys[i].key = dim == 3 ? (spread3(q[0]) | spread3(q[1]) << 1 | spread3(q[2]) << 2)
                           : (spread2(q[0]) | spread2(q[1]) << 1);
    keys[i].idx = (int32_t)i;
  }
  __gnu_parallel::sort(keys.begin(), keys.end());
  hp.new2old.resize(n_nodes);
  hp.old2new.resize(n_nodes);
  {
    // sigma = rows per sorting window (multiple of 64).  Large windows minimise padding but scatter a window's rows
    // over its slices by length, i.e. they trade gather locality for padding (measured on a 1 M-point Delaunay mesh,
    // sigma 64 ... 4096: SpMV 71 / 66 / 63 / 71 / 83 / 108 us; GL_SIGMA = 256).
    const int64_t sigma = GL_SIGMA;
    const int64_t nwin = (n_own + sigma - 1) / sigma;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t w = 0; w < nwin; ++w) {
      int64_t a = w * sigma, b = std::min<int64_t>(n_own, a + sigma);
      std::stable_sort(keys.begin() + a, keys.begin() + b, [&](const KeyIdx& x, const KeyIdx& y) {
        return (nbr_ptr[x.idx + 1] - nbr_ptr[x.idx]) > (nbr_ptr[y.idx + 1] - nbr_ptr[y.idx]);
      });
    }
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_own; ++i) {
    hp.new2old[i] = keys[i].idx;
    hp.old2new[keys[i].idx] = (int32_t)i;
  }
  for (int64_t i = n_own; i < n_nodes; ++i) hp.new2old[i] = hp.old2new[i] = (int32_t)i;   // ghosts stay
  keys.clear();
  keys.shrink_to_fit();

  lap("Morton + sigma sort");
  // ---- 4. SELL-64 matrix pattern + corner lists ------------------------------------------------------
  const int32_t n_slices = (int32_t)((n_own + GL_WAVE - 1) / GL_WAVE);
  hp.n_slices = n_slices;
  hp.slice_ptr.assign(n_slices + 1, 0);
  hp.cslice_ptr.assign(n_slices + 1, 0);
  int max_len = 0, max_clen = 0;
#pragma omp parallel for schedule(static) reduction(max : max_len, max_clen)
  for (int32_t s = 0; s < n_slices; ++s) {
    int64_t len = 0, clen = 0;
    for (int l = 0; l < GL_WAVE; ++l) {
      int64_t r = (int64_t)s * GL_WAVE + l;
      if (r >= n_own) break;
      int32_t o = hp.new2old[r];
      len = std::max(len, nbr_ptr[o + 1] - nbr_ptr[o]);
      clen = std::max(clen, adj_ptr[o + 1] - adj_ptr[o]);
    }
    hp.slice_ptr[s + 1] = len * GL_WAVE;
    hp.cslice_ptr[s + 1] = clen * GL_WAVE;
    max_len = std::max(max_len, (int)len);
    max_clen = std::max(max_clen, (int)clen);
  }
  // slot indices are 8-bit, and the assembly sweep keeps 2 * len columns of 64 doubles in LDS (160 KB per CU)
  GL_REQUIRE(max_len <= 150, "a mesh node has " + std::to_string(max_len) +
                                 " neighbours; rows longer than 150 do not fit the LDS-resident assembly");
  for (int32_t s = 0; s < n_slices; ++s) {
    hp.slice_ptr[s + 1] += hp.slice_ptr[s];
    hp.cslice_ptr[s + 1] += hp.cslice_ptr[s];
  }
  GL_REQUIRE(hp.slice_ptr[n_slices] * (int64_t)(dim * dim) < (int64_t(1) << 40), "operator too large");
  hp.max_len = max_len;
  hp.max_clen = max_clen;
  hp.cols.resize(hp.slice_ptr[n_slices]);
  hp.diag_k.assign((size_t)n_slices * GL_WAVE, 0);
  hp.rlen.assign((size_t)n_slices * GL_WAVE, 0);
  hp.cslots.assign(hp.cslice_ptr[n_slices], 0u);
  hp.celem.assign(hp.cslice_ptr[n_slices], -1);
  std::vector<uint8_t> is_boundary(n_slices, 0);

#pragma omp parallel
  {
    std::vector<int32_t> row(256);
#pragma omp for schedule(dynamic, 64)
    for (int32_t s = 0; s < n_slices; ++s) {
      const int64_t base = hp.slice_ptr[s], cbase = hp.cslice_ptr[s];
      const int len = (int)((hp.slice_ptr[s + 1] - base) / GL_WAVE);
      bool bnd = false;
      for (int l = 0; l < GL_WAVE; ++l) {
        const int64_t r = (int64_t)s * GL_WAVE + l;
        if (r >= n_own) {   // padding row: points at an existing column with zero value
          for (int k = 0; k < len; ++k) hp.cols[base + (int64_t)k * GL_WAVE + l] = 0;
          continue;
        }
        const int32_t o = hp.new2old[r];
        const int rl = (int)(nbr_ptr[o + 1] - nbr_ptr[o]);
        for (int k = 0; k < rl; ++k) row[k] = hp.old2new[nbr[nbr_ptr[o] + k]];
        std::sort(row.begin(), row.begin() + rl);
        for (int k = 0; k < rl; ++k) {
          hp.cols[base + (int64_t)k * GL_WAVE + l] = row[k];
          if (row[k] == (int32_t)r) hp.diag_k[r] = (uint8_t)k;
          if (row[k] >= n_own) bnd = true;
        }
        for (int k = rl; k < len; ++k) hp.cols[base + (int64_t)k * GL_WAVE + l] = (int32_t)r;
        hp.rlen[r] = (uint8_t)rl;
        int q = 0;
        for (int64_t a = adj_ptr[o]; a < adj_ptr[o + 1]; ++a, ++q) {
          const int32_t e = adj[a];
          uint32_t packed = 0;
          for (int m = 0; m < nv; ++m) {
            const int32_t vn = hp.old2new[cells[(int64_t)e * nv + m]];
            const int k = (int)(std::lower_bound(row.begin(), row.begin() + rl, vn) - row.begin());
            packed |= (uint32_t)k << (8 * m);
          }
          hp.celem[cbase + (int64_t)q * GL_WAVE + l] = e;
          hp.cslots[cbase + (int64_t)q * GL_WAVE + l] = packed;
        }
      }
      is_boundary[s] = bnd;
    }
  }
  for (int32_t s = 0; s < n_slices; ++s) (is_boundary[s] ? hp.boundary_slices : hp.interior_slices).push_back(s);
  lap("SELL-64 + corner packing");
  // 16-bit (window, offset) column codes: greedy cover of each slice's sorted distinct columns by windows of
  // 2^GL_WIN_BITS columns (optimal for fixed-length intervals); see glims_internal.h
  hp.cols16.assign(hp.cols.size(), 0);
  hp.win_base.assign((size_t)n_slices * GL_N_WIN, 0);
  hp.win_ok.assign(n_slices, 0);
  int64_t n_comp = 0;
  int win_limit = GL_N_WIN;   // GLIMS_WIN_LIMIT < 32 (test hook, include/glims_hip.h) forces slices onto the 32-bit fallback
  if (const char* e = getenv("GLIMS_WIN_LIMIT")) win_limit = std::max(0, std::min(GL_N_WIN, atoi(e)));
#pragma omp parallel reduction(+ : n_comp)
  {
    std::vector<int32_t> d;
#pragma omp for schedule(dynamic, 64)
    for (int32_t s = 0; s < n_slices; ++s) {
      const int64_t b = hp.slice_ptr[s], e = hp.slice_ptr[s + 1];
      d.assign(hp.cols.begin() + b, hp.cols.begin() + e);
      std::sort(d.begin(), d.end());
      d.erase(std::unique(d.begin(), d.end()), d.end());
      int32_t* wb = hp.win_base.data() + (size_t)s * GL_N_WIN;
      int nw = 0;
      size_t i = 0;
      while (i < d.size() && nw <= GL_N_WIN) {
        const int32_t start = d[i];
        if (nw < GL_N_WIN) wb[nw] = start;
        ++nw;
        while (i < d.size() && (int64_t)d[i] - start < (int64_t(1) << GL_WIN_BITS)) ++i;
      }
      if (nw > win_limit) continue;
      for (int w = nw; w < GL_N_WIN; ++w) wb[w] = wb[nw - 1];
      for (int64_t q = b; q < e; ++q) {
        const int32_t cj = hp.cols[q];
        const int w = (int)(std::upper_bound(wb, wb + nw, cj) - wb) - 1;
        hp.cols16[q] = (uint16_t)((w << GL_WIN_BITS) | (cj - wb[w]));
      }
      hp.win_ok[s] = 1;
      ++n_comp;
    }
  }
  hp.n_compressed = n_comp;
  lap("16-bit column codes");
  // length classes for the assembly sweep (its LDS footprint is 2 * cap * 64 * 8 B per wave)
  hp.bucket_cap = {16, 20, 24, 32, 48, 64, 96, 128, 255};
  hp.bucket_slices.assign(hp.bucket_cap.size(), {});
  // inside a class the interior slices (no ghost column) come first, so that a partitioned run can sweep them while
  // the halo of c is still in flight (bucket_interior[b] of them), then the rest
  hp.bucket_interior.assign(hp.bucket_cap.size(), 0);
  for (int pass = 0; pass < 2; ++pass)
    for (int32_t s = 0; s < n_slices; ++s) {
      if ((int)is_boundary[s] != pass) continue;
      const int len = (int)((hp.slice_ptr[s + 1] - hp.slice_ptr[s]) / GL_WAVE);
      size_t b = 0;
      while (hp.bucket_cap[b] < len) ++b;
      hp.bucket_slices[b].push_back(s);
      if (pass == 0) hp.bucket_interior[b]++;
    }
}
