// extern "C" surface of libglimship.so (see include/glims_hip.h for the contract of every entry point).
#include "glims_internal.h"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <omp.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

static void mailbox_close(glims_ctx* h);

namespace {

std::mutex g_err_mu;
std::string g_create_err;

// internal[new] <- staged[old]   /   staged[old] <- internal[new]
__global__ void k_perm_in(int64_t n, int bs, const int32_t* __restrict__ old2new, const double* __restrict__ ext,
                          double* __restrict__ in) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * bs) return;
  const int64_t o = i / bs;
  const int a = (int)(i - o * bs);
  in[(int64_t)old2new[o] * bs + a] = ext[i];
}
__global__ void k_perm_out(int64_t n, int64_t n_valid_new, int bs, const int32_t* __restrict__ old2new,
                           const double* __restrict__ in, double* __restrict__ ext) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * bs) return;
  const int64_t o = i / bs;
  const int a = (int)(i - o * bs);
  const int64_t nw = old2new[o];
  ext[i] = nw < n_valid_new ? in[nw * bs + a] : 0.0;
}

inline unsigned grid_exact(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

void to_device_perm(glims_ctx* h, const double* host, double* dst, int bs) {
  const int64_t n = h->n_nodes;
  GL_HIP(hipMemcpyAsync(h->stage.p, host, (size_t)n * bs * sizeof(double), hipMemcpyHostToDevice, h->st));
  hipLaunchKernelGGL(k_perm_in, dim3(grid_exact(n * bs)), dim3(256), 0, h->st, n, bs, h->d_old2new.p, h->stage.p,
                     dst);
  GL_HIP(hipGetLastError());
  GL_HIP(hipStreamSynchronize(h->st));
}
void from_device_perm(glims_ctx* h, const double* src, double* host, int bs, int64_t n_valid_new) {
  const int64_t n = h->n_nodes;
  hipLaunchKernelGGL(k_perm_out, dim3(grid_exact(n * bs)), dim3(256), 0, h->st, n, n_valid_new, bs,
                     h->d_old2new.p, src, h->stage.p);
  GL_HIP(hipGetLastError());
  GL_HIP(hipMemcpyAsync(host, h->stage.p, (size_t)n * bs * sizeof(double), hipMemcpyDeviceToHost, h->st));
  GL_HIP(hipStreamSynchronize(h->st));
}

template <class T>
uint64_t fnv_of(glims_ctx* h, const T* dev, size_t n) {
  std::vector<T> host(n);
  if (n) GL_HIP(hipMemcpyAsync(host.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost, h->st));
  GL_HIP(hipStreamSynchronize(h->st));
  uint64_t x = 1469598103934665603ull;
  const unsigned char* b = reinterpret_cast<const unsigned char*>(host.data());
  for (size_t i = 0; i < n * sizeof(T); ++i) x = (x ^ b[i]) * 1099511628211ull;
  return x;
}

template <class F>
int guarded(glims_ctx* h, F&& f) {
  if (!h) return GLIMS_E_USAGE;
  try {
    if (hipSetDevice(h->device) != hipSuccess) throw glims_error(GLIMS_E_HIP, "hipSetDevice failed");
    return f();
  } catch (const glims_error& e) {
    h->err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    h->err = e.what();
    return GLIMS_E_USAGE;
  }
}

}  // namespace

extern "C" {

int glims_abi_version(void) { return GLIMS_ABI_VERSION; }

int glims_options_default(glims_options* o) {
  if (!o) return GLIMS_E_USAGE;
  o->dt = 1.0;
  o->newton_rtol = 1e-10;
  o->newton_atol = 1e-13;
  o->newton_maxit = 50;
  o->cg_rtol = 1e-3;
  o->cg_atol = 0.0;
  o->cg_maxit = 5000;
  o->mech_rtol = 1e-10;
  o->mech_atol = 0.0;
  o->mech_maxit = 200000;
  o->check_every = 8;
  o->flags = GLIMS_FLAG_WARM_START;
  o->mech_precond = GLIMS_PRECOND_MULTIGRID;
  o->mech_mixed = 1;
  o->mech_history = 8;
  o->mg_smooth = 3;
  o->mg_coarse_nodes = 216;
  o->mg_h_factor = 0.0;
  o->mg_cheb_ratio = 0.0;
  o->time_kernels = 0;
  o->rd_precond = GLIMS_RD_PRECOND_AUTO;
  o->rd_mg_smooth = 0;
  o->rd_linear = GLIMS_RD_LINEAR_AUTO;
  o->stream_policy = GLIMS_STREAM_AUTO;
  return GLIMS_OK;
}

const char* glims_last_error(const glims_ctx* h) {
  if (h) return h->err.c_str();
  std::lock_guard<std::mutex> lk(g_err_mu);
  static thread_local std::string copy;
  copy = g_create_err;
  return copy.c_str();
}

int glims_create(glims_ctx** out, int dim, int64_t n_nodes, int64_t n_own, int64_t n_cells, const double* xyz,
                 const int32_t* cells, const int32_t* cell_label, int device) {
  if (!out) return GLIMS_E_USAGE;
  *out = nullptr;
  glims_ctx* h = nullptr;
  try {
    GL_REQUIRE(xyz && cells && cell_label, "null mesh arrays");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw glims_error(GLIMS_E_NO_DEVICE,
                        "no HIP device visible: libglimship has no CPU fallback (gfx950 / MI355X required)");
    GL_REQUIRE(device >= 0 && device < ndev, "device ordinal out of range");
    GL_HIP(hipSetDevice(device));
    h = new glims_ctx();
    h->device = device;
    h->dim = dim;
    h->nv = dim + 1;
    h->n_nodes = n_nodes;
    h->n_own = n_own;
    h->n_cells = n_cells;
    glims_options_default(&h->opt);
    std::memset(&h->stats, 0, sizeof(h->stats));
    if (const char* e = getenv("GLIMS_CHEB_TEST_SCALE_HI")) h->cheb_test_hi = std::max(0.05, atof(e));   // test hook, per handle
    GL_HIP(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
    GL_HIP(hipStreamCreateWithFlags(&h->st_comm, hipStreamNonBlocking));
    GL_HIP(hipEventCreate(&h->ev_a));
    GL_HIP(hipEventCreate(&h->ev_b));
    GL_HIP(hipEventCreateWithFlags(&h->ev_pack, hipEventDisableTiming));
    GL_HIP(hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming));
    GL_HIP(hipHostMalloc((void**)&h->h_pinned, 32 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h->h_pinned, 0, 32 * sizeof(double));
    GL_HIP(hipHostGetDevicePointer((void**)&h->mail_dev, h->h_pinned, 0));

    for (int64_t e = 0; e < n_cells; ++e)
      GL_REQUIRE(cell_label[e] >= 0 && cell_label[e] < GL_MAX_LABELS, "cell label outside [0, 256)");

    const bool verbose = getenv("GLIMS_VERBOSE") != nullptr;
    double t_last = omp_get_wtime();
    auto lap = [&](const char* what) {
      if (!verbose) return;
      const double t = omp_get_wtime();
      fprintf(stderr, "glims create: %-43s %7.3f s\n", what, t - t_last);
      t_last = t;
    };
    // caller's mesh to the device: the symbolic phase and the per-cell geometry read it there; coordinates and
    // connectivity in the caller's numbering are not kept
    dvec<double> d_xyz;
    dvec<int32_t> d_cells, cells_p;
    d_xyz.upload(xyz, (size_t)n_nodes * dim, h->st);
    d_cells.upload(cells, (size_t)n_cells * (dim + 1), h->st);
    DevPattern& p = h->pat;
    if (getenv("GLIMS_HOST_SYMBOLIC")) {   // TEST HOOK: the host implementation of the same phase (setup_host.cpp)
      HostPattern hp;
      build_host_pattern(hp, dim, n_nodes, n_own, n_cells, xyz, cells);
      lap("host pattern (total)");
      gl_mesh_metrics(h, hp, xyz);
      lap("mesh metrics (lattice test, edge lengths)");
      h->old2new = hp.old2new;
      h->new2old = hp.new2old;
      h->nnz = hp.nnz;
      h->n_corners = hp.n_corners;
      p.n_slices = hp.n_slices;
      p.max_len = hp.max_len;
      p.max_clen = hp.max_clen;
      p.total_entries = hp.slice_ptr[hp.n_slices];
      p.total_corners = hp.cslice_ptr[hp.n_slices];
      p.slice_ptr.upload(hp.slice_ptr, h->st);
      p.cols.upload(hp.cols, h->st);
      p.cols16.upload(hp.cols16, h->st);
      p.win_base.upload(hp.win_base, h->st);
      p.win_ok.upload(hp.win_ok, h->st);
      p.diag_k.upload(hp.diag_k, h->st);
      p.rlen.upload(hp.rlen, h->st);
      p.cslice_ptr.upload(hp.cslice_ptr, h->st);
      p.cslots.upload(hp.cslots, h->st);
      p.celem.upload(hp.celem, h->st);
      p.interior_slices.upload(hp.interior_slices, h->st);
      p.boundary_slices.upload(hp.boundary_slices, h->st);
      p.n_interior = (int32_t)hp.interior_slices.size();
      p.n_boundary = (int32_t)hp.boundary_slices.size();
      h->d_old2new.upload(hp.old2new, h->st);
      for (size_t b = 0; b < hp.bucket_cap.size(); ++b) {
        if (hp.bucket_slices[b].empty()) continue;
        // LDS of the class = its actual longest slice, not the class bound (16 rows x 1 KB would be exactly 1/10 of
        // the CU's LDS and fit only 9 times; the structured meshes' 15 fits 10 times)
        int cap = 1;
        for (int32_t sl : hp.bucket_slices[b])
          cap = std::max(cap, (int)((hp.slice_ptr[sl + 1] - hp.slice_ptr[sl]) / GL_WAVE));
        p.bucket_cap.push_back(cap);
        p.bucket_count.push_back((int32_t)hp.bucket_slices[b].size());
        p.bucket_interior.push_back(hp.bucket_interior[b]);
        auto* dv = new dvec<int32_t>();
        dv->upload(hp.bucket_slices[b], h->st);
        p.bucket_slices.push_back(dv);
      }
      for (int32_t sl = 0; sl < hp.n_slices; ++sl)
        if (hp.win_ok[sl]) h->nnz_idx16_avail += hp.slice_ptr[sl + 1] - hp.slice_ptr[sl];
      lap("pattern upload");
    } else {
      gl_build_pattern_device(h, d_xyz.p, d_cells.p, cells_p);
      lap("symbolic phase on the device (total)");
    }
    std::vector<uint8_t> lab(n_cells);
    for (int64_t e = 0; e < n_cells; ++e) lab[e] = (uint8_t)cell_label[e];
    if (h->cell_new2old.p) {   // labels and geometry in the internal cell order
      dvec<uint8_t> lab_caller;
      lab_caller.upload(lab, h->st);
      h->label.alloc((size_t)n_cells);
      gl_gather_u8(h, n_cells, h->cell_new2old.p, lab_caller.p, h->label.p);
      GL_HIP(hipStreamSynchronize(h->st));
    } else {
      h->label.upload(lab, h->st);
    }
    gl_compute_egeo(h, d_xyz.p, cells_p.p ? cells_p.p : d_cells.p);
    cells_p.release();
    GL_HIP(hipStreamSynchronize(h->st));
    d_xyz.release();
    d_cells.release();

    lap("labels, geometry");
    const size_t nn = (size_t)n_nodes, nd = (size_t)n_nodes * dim;
    h->c.alloc_zero(nn, h->st);
    h->c_old.alloc_zero(nn, h->st);
    h->b.alloc_zero(nn, h->st);
    h->dinv.alloc_zero(nn, h->st);
    h->cg_p.alloc_zero(nn, h->st);
    h->cg_s.alloc_zero(nn, h->st);
    h->cg_u.alloc_zero(nn, h->st);
    h->cg_w.alloc_zero(nn, h->st);
    h->cg_r.alloc_zero(nn, h->st);
    h->cg_r2.alloc_zero(nn, h->st);
    h->b2.alloc_zero(nn, h->st);
    h->stage.alloc_zero(nd, h->st);
    h->mat.alloc_zero(5 * GL_MAX_LABELS, h->st);
    h->partials.alloc_zero((size_t)p.n_slices * 3 + 4096, h->st);
    h->partials2.alloc_zero((size_t)(p.n_slices * 3 + 4096) / 1024 * 3 + 64, h->st);
    // block pairs of the vector kernels (grid_for caps at 4096 blocks) or of the multigrid cycle's last level-0 pass
    h->partials_v.alloc_zero((size_t)std::max(2 * 4096, 2 * (p.n_slices / 4 + 1)) + 64, h->st);
    h->partials_rr.alloc_zero(2 * 4096 + 64, h->st);
    h->red.alloc_zero(4, h->st);
    h->scal.alloc_zero(2 * SC_COUNT + 8, h->st);
    h->done.alloc_zero(1, h->st);
    GL_HIP(hipStreamSynchronize(h->st));

    h->spmv_unroll = h->mm.lattice ? 8 : 16;
    h->stats.n_rows = n_own;
    h->stats.nnz = h->nnz;
    h->stats.nnz_padded = p.total_entries;
    h->stats.n_corners = h->n_corners;
    h->stats.nnz_idx16 = h->nnz_idx16_avail;
    *out = h;
    return GLIMS_OK;
  } catch (const glims_error& e) {
    {
      std::lock_guard<std::mutex> lk(g_err_mu);
      g_create_err = e.what();
    }
    if (h) glims_destroy(h);
    return e.code;
  } catch (const std::exception& e) {
    {
      std::lock_guard<std::mutex> lk(g_err_mu);
      g_create_err = e.what();
    }
    if (h) glims_destroy(h);
    return GLIMS_E_USAGE;
  }
}

int glims_destroy(glims_ctx* h) {
  if (!h) return GLIMS_OK;
  (void)hipSetDevice(h->device);
  if (h->st) (void)hipStreamSynchronize(h->st);
  if (h->st_comm) (void)hipStreamSynchronize(h->st_comm);
  for (auto* d : h->snapshots) delete d;
  h->snapshots.clear();
  gl_comm_destroy(h);
  if (getenv("GLIMS_VERBOSE")) fprintf(stderr, "glims: deferred linear solves that ran out of iterations: %lld\n", (long long)h->stats_defer_miss);
  mailbox_close(h);
  for (hipEvent_t e : h->tev) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->cev) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->wev) (void)hipEventDestroy(e);
  if (h->h_pinned) (void)hipHostFree(h->h_pinned);
  if (h->ev_a) (void)hipEventDestroy(h->ev_a);
  if (h->ev_b) (void)hipEventDestroy(h->ev_b);
  if (h->ev_pack) (void)hipEventDestroy(h->ev_pack);
  if (h->ev_halo) (void)hipEventDestroy(h->ev_halo);
  if (h->st) (void)hipStreamDestroy(h->st);
  if (h->st_comm) (void)hipStreamDestroy(h->st_comm);
  delete h;   // dvec destructors release device memory
  return GLIMS_OK;
}

int glims_set_materials(glims_ctx* h, int n_labels, const double* D, const double* rho, const double* gamma,
                        const double* E, const double* nu) {
  return guarded(h, [&]() {
    h->mh_count = h->mh_next = 0;   // the elasticity solve history belongs to one operator
    h->mg.ready = h->mg_rd.ready = false;
    h->rd_precond_active = 0;
    GL_REQUIRE(n_labels > 0 && n_labels <= GL_MAX_LABELS, "n_labels out of range");
    GL_REQUIRE(D && rho && gamma && E && nu, "null material table");
    std::vector<double> m(5 * GL_MAX_LABELS, 0.0);
    for (int l = 0; l < n_labels; ++l) {
      GL_REQUIRE(std::isfinite(D[l]) && std::isfinite(rho[l]) && std::isfinite(gamma[l]) && std::isfinite(E[l]) &&
                     std::isfinite(nu[l]),
                 "non-finite material value");
      m[0 * GL_MAX_LABELS + l] = D[l];
      m[1 * GL_MAX_LABELS + l] = rho[l];
      m[2 * GL_MAX_LABELS + l] = gamma[l];
      // math_linear_elasticity.py:6-10
      m[3 * GL_MAX_LABELS + l] = E[l] / (2.0 * (1.0 + nu[l]));
      m[4 * GL_MAX_LABELS + l] = E[l] * nu[l] / ((1.0 + nu[l]) * (1.0 - 2.0 * nu[l]));
      GL_REQUIRE(std::isfinite(m[3 * GL_MAX_LABELS + l]) && std::isfinite(m[4 * GL_MAX_LABELS + l]),
                 "Poisson ratio " + std::to_string(nu[l]) + " of label " + std::to_string(l) +
                     " gives infinite Lame constants (need -1 < nu < 0.5)");
    }
    h->mat.upload(m, h->st);
    GL_HIP(hipStreamSynchronize(h->st));
    h->have_materials = true;
    h->is_setup = false;
    h->pending = false;
    return GLIMS_OK;
  });
}

int glims_set_options(glims_ctx* h, const glims_options* opt) {
  return guarded(h, [&]() {
    GL_REQUIRE(opt, "null options");
    GL_REQUIRE(opt->dt > 0.0 && std::isfinite(opt->dt), "dt must be positive");
    GL_REQUIRE(opt->newton_maxit >= 0 && opt->cg_maxit > 0 && opt->mech_maxit > 0, "bad iteration caps");
    GL_REQUIRE(opt->mech_precond == GLIMS_PRECOND_BLOCK_JACOBI || opt->mech_precond == GLIMS_PRECOND_MULTIGRID,
               "unknown mech_precond");
    GL_REQUIRE(opt->mech_mixed >= 0 && opt->mech_mixed <= 2 && opt->mech_history >= 0 && opt->mg_smooth >= 1 &&
                   opt->mg_coarse_nodes >= 1 && opt->mg_h_factor >= 0.0 && (opt->mg_cheb_ratio == 0.0 || opt->mg_cheb_ratio > 1.0),
               "bad elasticity solver options");
    GL_REQUIRE(opt->rd_precond >= GLIMS_RD_PRECOND_AUTO && opt->rd_precond <= GLIMS_RD_PRECOND_MULTIGRID &&
                   opt->rd_mg_smooth >= 0 && opt->rd_mg_smooth <= 8,
               "bad RD preconditioner options");
    GL_REQUIRE(opt->rd_linear >= GLIMS_RD_LINEAR_AUTO && opt->rd_linear <= GLIMS_RD_LINEAR_CHEBYSHEV &&
                   opt->stream_policy >= GLIMS_STREAM_AUTO && opt->stream_policy <= GLIMS_STREAM_CACHED,
               "bad rd_linear / stream_policy");
    // mg_smooth and mg_cheb_ratio are read by every cycle (no rebuild); the grids depend on the other two
    if (opt->mg_coarse_nodes != h->opt.mg_coarse_nodes || opt->mg_h_factor != h->opt.mg_h_factor)
      h->mg.ready = h->mg_rd.ready = false;
    // partitioned runs keep the first grid's operator for work boxes sized by the smoother degree: a new degree, a new set-up
    if (h->world > 1 && opt->mg_smooth != h->opt.mg_smooth) h->mg.ready = false;
    if (h->world > 1 && opt->rd_mg_smooth != h->opt.rd_mg_smooth) h->mg_rd.ready = false;
    if (opt->rd_precond != h->opt.rd_precond) {
      h->rd_precond_active = 0;                   // decided again by the next glims_step
      for (int& hint : h->cg_hint) hint = 0;      // iteration counts of the other preconditioner predict nothing
    }
    // the solve history is a ring of `mech_history` slots: a new depth starts an empty ring (a larger depth would
    // otherwise count never-allocated slots as stored solves)
    if (opt->mech_history != h->opt.mech_history) h->mh_count = h->mh_next = 0;
    if (opt->dt != h->opt.dt) h->is_setup = false;
    if ((opt->flags ^ h->opt.flags) & (GLIMS_FLAG_FP32_JACOBIAN | GLIMS_FLAG_INT32_COLUMNS)) h->is_setup = false;
    if (opt->stream_policy != h->opt.stream_policy) h->is_setup = false;
    if ((opt->flags ^ h->opt.flags) & (GLIMS_FLAG_MG_FP32_SMOOTHER | GLIMS_FLAG_MG_FP64_VECTORS | GLIMS_FLAG_MG_WHOLE_GRID | GLIMS_FLAG_MG_NO_LUMPING))
      h->mg.ready = h->mg_rd.ready = false;
    h->opt = *opt;
    h->pending = false;
    return GLIMS_OK;
  });
}

int glims_set_dirichlet_c(glims_ctx* h, int64_t n, const int64_t* node_ids, const double* values) {
  return guarded(h, [&]() {
    h->pending = false;
    h->dirichlet_c_exchange = h->world > 1;
    if (n <= 0) {
      if (h->have_fixed_c) h->mg_rd.ready = false;   // the RD hierarchy eliminates the constrained nodes
      h->have_fixed_c = false;
      h->fixed_c_host.clear();
      h->dirichlet_c_dirty = false;
      return GLIMS_OK;
    }
    GL_REQUIRE(node_ids && values, "null Dirichlet arrays");
    std::vector<uint8_t> fx(h->n_nodes, 0);
    std::vector<double> val(h->n_nodes, 0.0);
    for (int64_t k = 0; k < n; ++k) {
      GL_REQUIRE(node_ids[k] >= 0 && node_ids[k] < h->n_nodes, "Dirichlet node out of range");
      GL_REQUIRE(std::isfinite(values[k]), "non-finite Dirichlet value");
      fx[h->old2new[node_ids[k]]] = 1;
      val[h->old2new[node_ids[k]]] = values[k];
    }
    if (fx != h->fixed_c_host) h->mg_rd.ready = false;   // new VALUES every step are the normal case: no rebuild for those
    h->fixed_c_host = fx;
    h->fixed_c.upload(fx, h->st);
    h->fixed_c_val.upload(val, h->st);
    h->have_fixed_c = true;
    h->dirichlet_c_dirty = true;   // written into the iterate by the next step, after b = M c^n took the old values
    h->pending = false;
    GL_HIP(hipStreamSynchronize(h->st));
    return GLIMS_OK;
  });
}

int glims_set_dirichlet_u(glims_ctx* h, int64_t n, const int64_t* dof_ids, const double* values) {
  return guarded(h, [&]() {
    h->mh_count = h->mh_next = 0;   // the elasticity solve history belongs to one operator
    h->mg.ready = false;            // ... and so does the multigrid hierarchy (constrained dofs are eliminated in it)
    if (n <= 0) {
      h->have_fixed_u = false;
      return GLIMS_OK;
    }
    GL_REQUIRE(dof_ids && values, "null Dirichlet arrays");
    const int d = h->dim;
    std::vector<uint8_t> fx((size_t)h->n_nodes * d, 0);
    std::vector<double> val((size_t)h->n_nodes * d, 0.0);
    for (int64_t k = 0; k < n; ++k) {
      GL_REQUIRE(dof_ids[k] >= 0 && dof_ids[k] < h->n_nodes * d, "Dirichlet dof out of range");
      const int64_t node = dof_ids[k] / d;
      const int a = (int)(dof_ids[k] % d);
      const int64_t j = (int64_t)h->old2new[node] * d + a;
      fx[j] = 1;
      val[j] = values[k];
    }
    h->fixed_u.upload(fx, h->st);
    h->m_uD.upload(val, h->st);
    GL_HIP(hipStreamSynchronize(h->st));
    h->have_fixed_u = true;
    return GLIMS_OK;
  });
}

int glims_set_rd_load(glims_ctx* h, const double* f) {
  return guarded(h, [&]() {
    h->pending = false;
    if (!f) {
      h->have_load_rd = false;
      return GLIMS_OK;
    }
    h->load_rd.alloc((size_t)h->n_nodes);
    to_device_perm(h, f, h->load_rd.p, 1);
    h->have_load_rd = true;
    return GLIMS_OK;
  });
}

int glims_set_mech_load(glims_ctx* h, const double* f) {
  return guarded(h, [&]() {
    if (!f) {
      h->have_mload = false;
      return GLIMS_OK;
    }
    h->mload.alloc((size_t)h->n_nodes * h->dim);
    to_device_perm(h, f, h->mload.p, h->dim);
    h->have_mload = true;
    return GLIMS_OK;
  });
}

int glims_setup(glims_ctx* h, int with_mechanics) {
  return guarded(h, [&]() {
    h->mh_count = h->mh_next = 0;   // the elasticity solve history belongs to one operator
    h->mg.ready = h->mg_rd.ready = false;
    h->rd_precond_active = 0;       // decided by the first glims_step (a collective in partitioned runs)
    h->stats.rd_precond_used = 0;
    for (int& hint : h->cg_hint) hint = 0;
    h->mech_hint = 0;
    GL_REQUIRE(h->have_materials, "glims_setup before glims_set_materials");
    if (with_mechanics) {
      const size_t nd = (size_t)h->n_nodes * h->dim;
      if (!h->U.p) {
        h->U.alloc_zero(nd, h->st);
        h->m_rhs.alloc_zero(nd, h->st);
        h->m_p.alloc_zero(nd, h->st);
        h->m_s.alloc_zero(nd, h->st);
        h->m_u.alloc_zero(nd, h->st);
        h->m_w.alloc_zero(nd, h->st);
        h->m_r.alloc_zero(nd, h->st);
        h->m_dinv.alloc_zero((size_t)h->n_nodes * h->dim * h->dim, h->st);
      }
      if (!h->m_uD.p) h->m_uD.alloc_zero(nd, h->st);
    }
    h->pending = false;
    h->jac32 = (h->opt.flags & GLIMS_FLAG_FP32_JACOBIAN) != 0;
    h->use_idx16 = (h->opt.flags & GLIMS_FLAG_INT32_COLUMNS) == 0;
    h->stats.nnz_idx16 = h->use_idx16 ? h->nnz_idx16_avail : 0;
    h->cheb = glims_ctx::ChebState();   // the interval belongs to one operator
    h->have_d2 = false;
    {
      // Cache policy of the operator streams in the Krylov pass.  What one Krylov iteration touches: the stored entries
      // (values + column codes) and the iteration's vector traffic.  While that fits the 256 MiB Infinity Cache with room to
      // spare, default-policy loads keep the operator resident from pass to pass; beyond it non-temporal streams stop the
      // operator from displacing the gathered vector (measured on a 1/8 share of config C4 and on C4).
      const int64_t ent = h->pat.total_entries;
      const int64_t ws = ent * ((h->jac32 ? 4 : 8) + (h->use_idx16 ? 2 : 4)) + h->n_own * 64;
      h->stats.krylov_working_set = ws;
      h->stream_nt = h->opt.stream_policy == GLIMS_STREAM_NONTEMPORAL ? 1
                     : h->opt.stream_policy == GLIMS_STREAM_CACHED    ? 0
                                                                      : (ws > GL_STREAM_CACHED_LIMIT ? 1 : 0);
      h->stats.stream_nontemporal = h->stream_nt;
    }
    gl_assemble_static(h, with_mechanics);
    GL_HIP(hipStreamSynchronize(h->st));
    h->is_setup = true;
    h->have_mech = with_mechanics != 0;
    return GLIMS_OK;
  });
}

int glims_set_state(glims_ctx* h, const double* c, const double* u) {
  return guarded(h, [&]() {
    GL_REQUIRE(c, "null concentration");
    h->pending = false;
    h->have_c_old = false;
    h->have_d2 = false;
    h->d2_off = h->d2_good = 0;
    h->d2_backoff = 8;
    // A new state starts a new run (FenicsSimulation.run() may be called again on the same object, simulation_base.py:166-168,
    // run_for_adjoint does): what the Newton iteration has learnt from the previous run's steps is forgotten, so that the run
    // takes the iteration path -- and produces the bits -- of a fresh handle.  (What stays: the operators, both multigrid
    // hierarchies and the preconditioner `auto` has settled on; they depend on the mesh and the parameters, not on the run.)
    h->nw_mode = h->nw_hold = h->nw_since = h->nw_steps = 0;
    h->cheb = glims_ctx::ChebState();   // ... and the spectral interval of the dot-free solves (measured again by the first step)
    h->nq_first_ratio = 1e-3;
    h->nq_skip_steps = 0;
    for (int& hint : h->cg_hint) hint = 0;
    h->mech_hint = 0;
    h->mh_count = h->mh_next = 0;   // ... and the elasticity solver's history of right-hand sides
    to_device_perm(h, c, h->c.p, 1);
    h->have_state = true;
    h->dirichlet_c_dirty = h->have_fixed_c;
    h->dirichlet_c_exchange = h->world > 1;
    if (h->U.p) {
      if (u)
        to_device_perm(h, u, h->U.p, h->dim);
      else
        GL_HIP(hipMemsetAsync(h->U.p, 0, (size_t)h->n_nodes * h->dim * sizeof(double), h->st));
    }
    GL_HIP(hipStreamSynchronize(h->st));
    h->have_state = true;
    h->stats.steps = 0;
    return GLIMS_OK;
  });
}

int glims_get_state(glims_ctx* h, double* c, double* u) {
  return guarded(h, [&]() {
    GL_REQUIRE(h->have_state, "glims_get_state before glims_set_state");
    if (c) from_device_perm(h, h->c.p, c, 1, h->n_nodes);
    if (u) {
      GL_REQUIRE(h->U.p, "no displacement field: glims_setup(with_mechanics=1) first");
      from_device_perm(h, h->U.p, u, h->dim, h->n_nodes);
    }
    return GLIMS_OK;
  });
}

int glims_step(glims_ctx* h, int n_steps) {
  return guarded(h, [&]() { return gl_step(h, n_steps); });
}

int glims_solve_mechanics(glims_ctx* h) {
  return guarded(h, [&]() { return gl_solve_mechanics(h); });
}

int glims_get_stats(const glims_ctx* h, glims_stats* st) {
  if (!h || !st) return GLIMS_E_USAGE;
  *st = h->stats;
  return GLIMS_OK;
}

int glims_reset_stats(glims_ctx* h) {
  if (!h) return GLIMS_E_USAGE;
  glims_stats keep = h->stats;
  std::memset(&h->stats, 0, sizeof(h->stats));
  h->stats.n_rows = keep.n_rows;
  h->stats.nnz = keep.nnz;
  h->stats.nnz_padded = keep.nnz_padded;
  h->stats.n_corners = keep.n_corners;
  h->stats.nnz_idx16 = keep.nnz_idx16;
  h->stats.mg_levels = keep.mg_levels;
  h->stats.mg_complexity = keep.mg_complexity;
  h->stats.ms_mg_setup = keep.ms_mg_setup;
  h->stats.rd_precond_used = keep.rd_precond_used;
  h->stats.rd_stiffness_ratio = keep.rd_stiffness_ratio;
  h->stats.rd_mg_levels = keep.rd_mg_levels;
  h->stats.rd_mg_complexity = keep.rd_mg_complexity;
  h->stats.ms_rd_mg_setup = keep.ms_rd_mg_setup;
  h->stats.reduce_transport = keep.reduce_transport;
  h->stats.mg_grid1_bytes = keep.mg_grid1_bytes;
  h->stats.mg_box_fraction = keep.mg_box_fraction;
  h->stats.cheb_lmin = keep.cheb_lmin;
  h->stats.cheb_lmax = keep.cheb_lmax;
  h->stats.stream_nontemporal = keep.stream_nontemporal;
  h->stats.krylov_working_set = keep.krylov_working_set;
  h->tev_used = 0;
  h->stats.steps = keep.steps;   // step counter drives the extrapolated guess; keep it
  return GLIMS_OK;
}

int glims_apply(glims_ctx* h, int which, const double* x, double* y, int reps, double* ms_total) {
  return guarded(h, [&]() {
    GL_REQUIRE(h->is_setup, "glims_apply before glims_setup");
    GL_REQUIRE(x && y && reps >= 1, "bad arguments");
    GL_REQUIRE(which >= 0 && which <= 9 && which != 6, "unknown operator");
    if (which == 7) GL_REQUIRE(h->have_state, "the matrix-free product needs the state c (glims_set_state)");
    if (which >= 8) h->pending = false;   // the sweep rewrites A(c), dinv and the Krylov work vectors
    const int d = h->dim;
    const bool blk_in = which == 3, blk_out = which == 3 || which == 4;
    if (blk_out) GL_REQUIRE(h->have_mech, "mechanics operators not assembled");
    dvec<double> xin, yout;
    xin.alloc_zero((size_t)h->n_nodes * (blk_in ? d : 1), h->st);
    yout.alloc_zero((size_t)h->n_nodes * (blk_out ? d : 1), h->st);
    to_device_perm(h, x, xin.p, blk_in ? d : 1);
    const bool saved_mload = h->have_mload;
    h->have_mload = false;
    dvec<double> zero_b;
    if (which == 8) zero_b.alloc_zero((size_t)h->n_nodes, h->st);
    GL_HIP(hipEventRecord(h->ev_a, h->st));
    for (int r = 0; r < reps; ++r) {
      if (which == 7)   // matrix-free A(c) x from the incidence lists (measurement only)
        gl_rd_matfree(h, h->c.p, xin.p, yout.p);
      else if (which == 8)   // the assembly sweep at c = x with b = 0: y = -1/2 (A(x) + S) x; A(x) and its diagonal are left in place
        gl_rd_assemble(h, xin.p, zero_b.p, nullptr, yout.p, h->cg_r2.p, h->partials.p);
      else if (which == 9) {   // the quadratic-term pass with a = delta = x: y -= dt N(x) x per repetition
        h->nq_ad.alloc((size_t)2 * h->n_nodes);
        gl_pair_of(h, xin.p, h->nq_ad.p);
        gl_rd_quad(h, h->nq_ad.p, yout.p, h->partials.p);
      }
      else if (which == 5)   // A x with the fused dot product of the Krylov iteration (timing studies)
        gl_launch_spmv(h, h->st, h->pat.n_slices, nullptr, h->vA.p, xin.p, yout.p, nullptr, nullptr, xin.p,
                       h->partials.p, 0, nullptr, h->jac32 ? h->vA32.p : nullptr);
      else if (which <= 2)
        gl_spmv_scalar(h, which == 0 ? h->vA.p : which == 1 ? h->vS.p : h->vM.p, xin.p, yout.p, false);
      else if (which == 3)
        gl_spmv_block(h, xin.p, yout.p, false);
      else
        gl_apply_G(h, xin.p, yout.p);
    }
    GL_HIP(hipEventRecord(h->ev_b, h->st));
    GL_HIP(hipEventSynchronize(h->ev_b));
    h->have_mload = saved_mload;
    float ms = 0.f;
    GL_HIP(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    if (ms_total) *ms_total = ms;
    h->stats.ms_spmv += ms;
    from_device_perm(h, yout.p, y, blk_out ? d : 1, h->n_own);
    // which = 8 has left A(x) and its diagonal where A(c) belongs: put A(c) back (one more sweep, outside the timed region), so
    // that every later glims_apply(0 | 5), glims_project or glims_step sees the operator of the STATE, as after any other `which`
    if (which == 8 && h->have_state) gl_rd_assemble(h, h->c.p, zero_b.p, nullptr, xin.p, h->cg_r2.p, h->partials.p);
    return GLIMS_OK;
  });
}

int glims_rd_residual(glims_ctx* h, const double* c, const double* c_prev, double* R) {
  return guarded(h, [&]() {
    GL_REQUIRE(h->is_setup, "glims_rd_residual before glims_setup");
    GL_REQUIRE(c && c_prev, "null vectors");
    dvec<double> dc, dcp;
    dc.alloc_zero((size_t)h->n_nodes, h->st);
    dcp.alloc_zero((size_t)h->n_nodes, h->st);
    to_device_perm(h, c, dc.p, 1);
    to_device_perm(h, c_prev, dcp.p, 1);
    gl_launch_spmv(h, h->st, h->pat.n_slices, nullptr, h->vM.p, dcp.p, h->b.p, nullptr,
                   h->have_load_rd ? h->load_rd.p : nullptr, nullptr, nullptr, 0, nullptr);
    h->pending = false;
    gl_rd_assemble(h, dc.p, h->b.p, nullptr, h->cg_r.p, h->cg_r2.p, h->partials.p);
    GL_HIP(hipStreamSynchronize(h->st));
    if (R) {
      from_device_perm(h, h->cg_r.p, R, 1, h->n_own);
      for (int64_t i = 0; i < h->n_nodes; ++i) R[i] = -R[i];   // the kernel stores -R (the Newton right-hand side)
    }
    return GLIMS_OK;
  });
}

int glims_get_numbering(glims_ctx* h, int32_t* old2new) {
  return guarded(h, [&]() {
    GL_REQUIRE(old2new, "null output");
    std::memcpy(old2new, h->old2new.data(), (size_t)h->n_nodes * sizeof(int32_t));
    return GLIMS_OK;
  });
}

int glims_pattern_checksum(glims_ctx* h, uint64_t out[13]) {
  return guarded(h, [&]() {
    GL_REQUIRE(out, "null output");
    const DevPattern& p = h->pat;
    out[0] = fnv_of(h, p.slice_ptr.p, (size_t)p.n_slices + 1);
    out[1] = fnv_of(h, p.cols.p, (size_t)p.total_entries);
    out[2] = fnv_of(h, p.cols16.p, (size_t)p.total_entries);
    out[3] = fnv_of(h, p.win_base.p, (size_t)p.n_slices * GL_N_WIN);
    out[4] = fnv_of(h, p.win_ok.p, (size_t)p.n_slices);
    out[5] = fnv_of(h, p.diag_k.p, (size_t)p.n_slices * GL_WAVE);
    out[6] = fnv_of(h, p.cslice_ptr.p, (size_t)p.n_slices + 1);
    out[7] = fnv_of(h, p.cslots.p, (size_t)p.total_corners);
    if (h->cell_new2old.p) {   // the incidences' cells in the CALLER's numbering, as the host version stores them
      dvec<int32_t> tmp;
      tmp.alloc((size_t)p.total_corners);
      gl_translate_cells(h, (int64_t)p.total_corners, p.celem.p, h->cell_new2old.p, tmp.p);
      out[8] = fnv_of(h, tmp.p, (size_t)p.total_corners);
    } else {
      out[8] = fnv_of(h, p.celem.p, (size_t)p.total_corners);
    }
    out[9] = fnv_of(h, p.interior_slices.p, (size_t)p.n_interior);
    out[10] = fnv_of(h, p.boundary_slices.p, (size_t)p.n_boundary);
    out[11] = fnv_of(h, h->d_old2new.p, (size_t)h->n_nodes);
    out[12] = fnv_of(h, p.rlen.p, (size_t)p.n_slices * GL_WAVE);
    return GLIMS_OK;
  });
}

int glims_snapshot_save(glims_ctx* h, int64_t* id_out) {
  return guarded(h, [&]() {
    GL_REQUIRE(h->have_state && id_out, "glims_snapshot_save needs a state and an output id");
    auto* d = new dvec<double>();
    try {
      d->alloc((size_t)h->n_nodes);
    } catch (...) {
      delete d;
      throw;
    }
    GL_HIP(hipMemcpyAsync(d->p, h->c.p, (size_t)h->n_nodes * sizeof(double), hipMemcpyDeviceToDevice, h->st));
    GL_HIP(hipStreamSynchronize(h->st));
    h->snapshots.push_back(d);
    *id_out = (int64_t)h->snapshots.size() - 1;
    return GLIMS_OK;
  });
}

int glims_snapshot_load(glims_ctx* h, int64_t id, double* c) {
  return guarded(h, [&]() {
    GL_REQUIRE(c && id >= 0 && id < (int64_t)h->snapshots.size() && h->snapshots[id], "unknown snapshot id");
    from_device_perm(h, h->snapshots[id]->p, c, 1, h->n_nodes);
    return GLIMS_OK;
  });
}

int glims_snapshot_mechanics(glims_ctx* h, int64_t id, double* u) {
  return guarded(h, [&]() {
    GL_REQUIRE(u && id >= 0 && id < (int64_t)h->snapshots.size() && h->snapshots[id], "unknown snapshot id");
    const int st = gl_solve_mechanics(h, h->snapshots[id]->p);
    from_device_perm(h, h->U.p, u, h->dim, h->n_nodes);
    return st;
  });
}

int glims_snapshot_clear(glims_ctx* h) {
  return guarded(h, [&]() {
    for (auto* d : h->snapshots) delete d;
    h->snapshots.clear();
    return GLIMS_OK;
  });
}

int glims_project(glims_ctx* h, const double* rhs, double* x, int ncomp, double rtol) {
  return guarded(h, [&]() {
    GL_REQUIRE(rhs && x && ncomp >= 1 && rtol > 0.0, "bad arguments");
    const int64_t n = h->n_nodes;
    std::vector<double> col(n), out(n);
    dvec<double> d_rhs, d_x;
    d_rhs.alloc_zero((size_t)n, h->st);
    d_x.alloc_zero((size_t)n, h->st);
    int status = GLIMS_OK;
    for (int q = 0; q < ncomp; ++q) {
      for (int64_t i = 0; i < n; ++i) col[i] = rhs[i * ncomp + q];
      to_device_perm(h, col.data(), d_rhs.p, 1);
      const int st = gl_project(h, d_rhs.p, d_x.p, rtol);
      if (st != GLIMS_OK) status = st;
      from_device_perm(h, d_x.p, out.data(), 1, h->n_nodes);
      for (int64_t i = 0; i < n; ++i) x[i * ncomp + q] = out[i];
    }
    return status;
  });
}

int glims_comm_unique_id(char id[GLIMS_UNIQUE_ID_BYTES]) {
  static_assert(GLIMS_UNIQUE_ID_BYTES >= 2 * sizeof(ncclUniqueId), "unique id buffer too small");
  if (!id) return GLIMS_E_USAGE;
  ncclUniqueId a, b;
  if (ncclGetUniqueId(&a) != ncclSuccess || ncclGetUniqueId(&b) != ncclSuccess) return GLIMS_E_RCCL;
  std::memset(id, 0, GLIMS_UNIQUE_ID_BYTES);
  std::memcpy(id, &a, sizeof(a));
  std::memcpy(id + sizeof(a), &b, sizeof(b));
  return GLIMS_OK;
}

int glims_comm_init(glims_ctx* h, int rank, int world, const char id[GLIMS_UNIQUE_ID_BYTES]) {
  return guarded(h, [&]() {
    GL_REQUIRE(world >= 1 && rank >= 0 && rank < world && id, "bad communicator arguments");
    gl_comm_destroy(h);
    h->rank = rank;
    h->world = world;
    if (world > 1) {
      ncclUniqueId a, b;
      std::memcpy(&a, id, sizeof(a));
      std::memcpy(&b, id + sizeof(a), sizeof(b));
      // two communicators: halo send/recv run on the communication stream, the scalar all-reduce on the compute
      // stream; a communicator must not be driven from two streams at once
      GL_NCCL(ncclCommInitRank(&h->comm_halo, world, a, rank));
      GL_NCCL(ncclCommInitRank(&h->comm_red, world, b, rank));
    }
    return GLIMS_OK;
  });
}

static void mailbox_close(glims_ctx* h) {
  if (h->nm_host) {
    (void)hipHostUnregister(h->nm_host);
    (void)munmap(h->nm_host, h->nm_bytes);
  }
  h->nm_host = nullptr;
  h->nm_bytes = 0;
  h->nm = NodeMail();
}

int glims_comm_mailbox(glims_ctx* h, const char* shm_name) {
  return guarded(h, [&]() {
    GL_HIP(hipStreamSynchronize(h->st));
    mailbox_close(h);
    if (!shm_name || h->world <= 1) return GLIMS_OK;
    GL_REQUIRE(shm_name[0] == '/', "shm_name must start with '/'");
    GL_REQUIRE(h->world <= GL_WAVE, "node mailbox supports at most 64 ranks");
    const size_t bytes = ((size_t)2 * h->world * 64 + 4095) / 4096 * 4096;
    const int fd = shm_open(shm_name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) throw glims_error(GLIMS_E_USAGE, std::string("shm_open failed: ") + std::strerror(errno));
    // every rank sizes the object to the same length: the first one zero-fills it, the others change nothing
    if (ftruncate(fd, (off_t)bytes) != 0) {
      const std::string m = std::string("ftruncate failed: ") + std::strerror(errno);
      close(fd);
      throw glims_error(GLIMS_E_USAGE, m);
    }
    void* ptr = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (ptr == MAP_FAILED) throw glims_error(GLIMS_E_USAGE, std::string("mmap failed: ") + std::strerror(errno));
    h->nm_host = ptr;
    h->nm_bytes = bytes;
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) {
      (void)munmap(ptr, bytes);
      h->nm_host = nullptr;
      throw glims_error(GLIMS_E_HIP, std::string("hipHostRegister of the node mailbox failed: ") + hipGetErrorString(e));
    }
    void* dev = nullptr;
    GL_HIP(hipHostGetDevicePointer(&dev, ptr, 0));
    h->nm_seq.alloc_zero(1, h->st);
    h->nm_err.alloc_zero(1, h->st);
    GL_HIP(hipStreamSynchronize(h->st));
    h->nm.slots = static_cast<double*>(dev);
    h->nm.seq = h->nm_seq.p;
    h->nm.err = h->nm_err.p;
    h->nm.rank = h->rank;
    h->nm.world = h->world;
    return GLIMS_OK;
  });
}

int glims_comm_mailbox_selftest(glims_ctx* h) {
  return guarded(h, [&]() { return gl_mailbox_selftest(h); });
}

int glims_comm_selftest(glims_ctx* h) {
  return guarded(h, [&]() { return gl_comm_selftest(h); });
}

int glims_set_transport(glims_ctx* h, int rank, int world, glims_halo_fn halo, glims_allreduce_fn allreduce,
                        void* user) {
  return guarded(h, [&]() {
    GL_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    GL_REQUIRE(world == 1 || (halo && allreduce), "both transport callbacks are required");
    gl_comm_destroy(h);
    h->rank = rank;
    h->world = world;
    h->tr_halo = halo;
    h->tr_allreduce = allreduce;
    h->tr_user = user;
    return GLIMS_OK;
  });
}

int glims_set_mg_frame(glims_ctx* h, const double* lo, const double* hi) {
  return guarded(h, [&]() {
    h->mg.ready = h->mg_rd.ready = false;
    if (!lo || !hi) {
      h->mg_frame_set = false;
      return GLIMS_OK;
    }
    for (int a = 0; a < h->dim; ++a) {
      GL_REQUIRE(std::isfinite(lo[a]) && std::isfinite(hi[a]) && hi[a] > lo[a], "bad bounding box (need hi > lo on every axis)");
      h->mg_frame_lo[a] = lo[a];
      h->mg_frame_hi[a] = hi[a];
    }
    h->mg_frame_set = true;
    return GLIMS_OK;
  });
}

int glims_set_halo(glims_ctx* h, int n_peers, const int32_t* peer_rank, const int64_t* send_ptr,
                   const int32_t* send_idx, const int64_t* recv_count) {
  return guarded(h, [&]() {
    GL_REQUIRE(n_peers >= 0, "negative peer count");
    h->n_peers = n_peers;
    h->peer_rank.assign(peer_rank, peer_rank + n_peers);
    h->send_ptr.assign(send_ptr, send_ptr + n_peers + 1);
    h->recv_ptr.assign(n_peers + 1, 0);
    for (int p = 0; p < n_peers; ++p) {
      GL_REQUIRE(peer_rank[p] >= 0 && peer_rank[p] < h->world && peer_rank[p] != h->rank, "bad peer rank");
      GL_REQUIRE(recv_count[p] >= 0 && send_ptr[p + 1] >= send_ptr[p], "bad halo counts");
      h->recv_ptr[p + 1] = h->recv_ptr[p] + recv_count[p];
    }
    GL_REQUIRE(h->recv_ptr[n_peers] == h->n_nodes - h->n_own, "ghost count does not match the halo plan");
    h->n_send = send_ptr[n_peers];
    std::vector<int32_t> idx(h->n_send);
    for (int64_t k = 0; k < h->n_send; ++k) {
      GL_REQUIRE(send_idx[k] >= 0 && send_idx[k] < h->n_own, "send index is not an owned node");
      idx[k] = h->old2new[send_idx[k]];
    }
    h->send_idx.upload(idx, h->st);
    {   // inverse map: which send slots does an owned row feed (a corner node goes to several peers)
      std::vector<int32_t> cnt(h->n_own, 0);
      for (int64_t k = 0; k < h->n_send; ++k) cnt[idx[k]]++;
      std::vector<int32_t> ref(h->n_own, -1), ptr(1, 0);
      for (int64_t r = 0; r < h->n_own; ++r)
        if (cnt[r]) {
          ref[r] = (int32_t)ptr.size() - 1;
          ptr.push_back(ptr.back() + cnt[r]);
        }
      std::vector<int32_t> slot(std::max<int64_t>(h->n_send, 1), 0), fill(ptr.begin(), ptr.end());
      for (int64_t k = 0; k < h->n_send; ++k) slot[fill[ref[idx[k]]]++] = (int32_t)k;
      h->send_ref.upload(ref, h->st);
      h->send_slot_ptr.upload(ptr, h->st);
      h->send_slot.upload(slot, h->st);
    }
    h->sendbuf.alloc_zero((size_t)std::max<int64_t>(h->n_send, 1) * h->dim, h->st);
    GL_HIP(hipStreamSynchronize(h->st));
    return GLIMS_OK;
  });
}

}  // extern "C"
