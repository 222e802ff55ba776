// Internal declarations of libglimship.so (gfx950 only; no CPU fallback, no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <stdexcept>
#include "../../include/glims_hip.h"

#define GL_WAVE 64                 // wavefront width of CDNA4; SELL slice height
#define GL_SIGMA 256               // sigma window (rows) of the SELL-C-sigma row sort (tools/sigma_sweep.py: on an
                                   // unstructured mesh 256 beats 4096 by 1.7x in SpMV time despite +17 % padding)
#define GL_MAX_LABELS 256
// 16-bit column codes: the columns of one 64-row slice fall into a few compact ranges of the Morton numbering (the
// neighbouring 4x4x4 blocks), so a code = (window id, offset) against a per-slice table of GL_N_WIN window bases
// covers every slice of the BASELINE meshes and of 1 M-point Delaunay meshes (tools/pattern_stats.cpp); the SpMV then
// streams 2 B instead of 4 B of index per entry.  Slices that need more windows fall back to the 32-bit stream.
#define GL_WIN_BITS 11
#define GL_N_WIN 32
// multigrid, partitioned runs with glims_set_mg_frame: the auxiliary grids are replicated on every rank; the operator of
// the first one (stencil entries x block entries x 4 B per node: 27 x 9 x 4 B for elasticity on a lattice mesh = 4 M
// nodes, 125 x 9 x 4 B on a general mesh = 0.9 M nodes) may take at most this many bytes -- beyond it the grid spacing
// is widened until it fits
#define GL_MG_GLOBAL_BYTES (4ll << 30)
// Blocks are dealt to the 8 XCDs in chunks of this many consecutive logical blocks: neighbouring slices (overlapping x
// gathers) share one L2 while the XCDs together still walk the matrix front to back (measured: time of the plain
// mapping, fabric reads 2.21 -> 1.97 GB per SpMV at 10 M rows; contiguous eighths are 1-5 % slower).
// glims_options.stream_policy = AUTO: Krylov working sets up to this many bytes stream the operator with the default cache policy
// (measured, profiles/r05_ab_rank_sized.txt: at 269 MB -- C4 / 8 -- cached loads win by 6-7 %, at 2.2 GB -- C4 -- the two are equal)
#define GL_STREAM_CACHED_LIMIT (1024ll << 20)
// (chunks of 128-512 blocks change the in-step kernel times by <= 2 %: profiles/r04_ab_xcd_chunk.txt)
#define GL_XCD_CHUNK 64

struct glims_error : std::runtime_error {
  int code;
  glims_error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define GL_HIP(expr)                                                                            \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      throw glims_error(GLIMS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + \
                                         __FILE__ + ":" + std::to_string(__LINE__) + ")");      \
  } while (0)

#define GL_NCCL(expr)                                                                              \
  do {                                                                                             \
    ncclResult_t _r = (expr);                                                                      \
    if (_r != ncclSuccess)                                                                         \
      throw glims_error(GLIMS_E_RCCL, std::string(#expr) + ": " + ncclGetErrorString(_r) + " (" + \
                                          __FILE__ + ":" + std::to_string(__LINE__) + ")");        \
  } while (0)

#define GL_REQUIRE(cond, msg)                                  \
  do {                                                         \
    if (!(cond)) throw glims_error(GLIMS_E_USAGE, (msg));      \
  } while (0)

// Device array with explicit lifetime (no implicit copies).
template <class T>
struct dvec {
  T* p = nullptr;
  size_t n = 0;
  dvec() = default;
  dvec(const dvec&) = delete;
  dvec& operator=(const dvec&) = delete;
  ~dvec() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    if (count == n && p) return;
    release();
    if (count) GL_HIP(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
  }
  void alloc_zero(size_t count, hipStream_t s) {
    alloc(count);
    if (count) GL_HIP(hipMemsetAsync(p, 0, count * sizeof(T), s));
  }
  void upload(const T* h, size_t count, hipStream_t s) {
    alloc(count);
    if (count) GL_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void upload(const std::vector<T>& h, hipStream_t s) { upload(h.data(), h.size(), s); }
};

// Host-side result of the symbolic phase (setup_host.cpp).
struct HostPattern {
  int dim = 0, nv = 0;
  int64_t n_nodes = 0, n_own = 0, n_cells = 0;
  std::vector<int32_t> old2new, new2old;   // node renumbering (ghosts keep their place)
  int32_t n_slices = 0;                    // ceil(n_own / 64)
  std::vector<int64_t> slice_ptr;          // [n_slices+1] offsets into cols (multiples of 64)
  std::vector<int32_t> cols;               // SELL-64 column indices (new numbering), padded with the row itself
  // compressed column stream: cols16 = (window << GL_WIN_BITS) | offset, column = win_base[slice][window] + offset;
  // slices whose columns need more than GL_N_WIN windows keep win_ok = 0 and are read through `cols`
  std::vector<uint16_t> cols16;
  std::vector<int32_t> win_base;           // [n_slices * GL_N_WIN]
  std::vector<uint8_t> win_ok;             // [n_slices]
  int64_t n_compressed = 0;                // slices with win_ok
  std::vector<uint8_t> diag_k;             // [n_slices*64] slot of the diagonal in each row
  std::vector<uint8_t> rlen;               // [n_slices*64] stored entries of each row (0 on the padding rows of the last slice)
  std::vector<int64_t> cslice_ptr;         // [n_slices+1] offsets into the (row, cell) incidence arrays
  std::vector<uint32_t> cslots;            // 4 x uint8: slot (within the row) of each vertex of the cell
  std::vector<int32_t> celem;              // cell id, -1 = padding
  std::vector<int32_t> interior_slices, boundary_slices;   // boundary = references a ghost column
  // slices grouped by row-length class (LDS footprint of the assembly sweep): bucket b holds the slices with
  // bucket_cap[b-1] < len <= bucket_cap[b]
  std::vector<int> bucket_cap;
  std::vector<std::vector<int32_t>> bucket_slices;
  std::vector<int32_t> bucket_interior;      // leading entries of bucket_slices[b] that are interior slices
  int max_len = 0, max_clen = 0;
  int64_t nnz = 0, n_corners = 0;
};

struct glims_ctx;
// symbolic.hip: the same phase on the device (the default; the host version above stays as its cross-check)
// cells_p (out): the connectivity in the internal cell order (caller's vertex indices), for the per-cell geometry
void gl_build_pattern_device(glims_ctx* h, const double* d_xyz, const int32_t* d_cells, dvec<int32_t>& cells_p);
int gl_host_threads();   // OpenMP team the host phases may use (affinity mask, cgroup quota, ranks per host)
void build_host_pattern(HostPattern& hp, int dim, int64_t n_nodes, int64_t n_own, int64_t n_cells,
                        const double* xyz, const int32_t* cells);

// Everything a wave of the incidence-list kernels needs to know about its slice, in ONE 32-byte scalar load (the chain
// slice list -> slice offsets -> incidence offsets -> window flag was four dependent memory round trips before a wave could
// issue its first useful load)
struct SliceDesc {
  int64_t base, cbase;     // first entry / first incidence record of the slice
  int32_t s;               // slice
  int32_t len, clen;       // entries / incidence records per lane
  int32_t ok;              // the slice has 16-bit column codes
};

// Slices of similar length that the incidence-list kernels launch together (kernels.hip, ensure_classes)
struct SliceClass {
  int cap = 0;                 // longest slice of the class, entries per lane (= LDS columns of its launches)
  int n = 0, n_interior = 0;   // slices; the interior ones (no ghost column) come first
  dvec<SliceDesc> desc;        // ... as descriptors (straight-line kernels)
  dvec<int32_t> list;          // ... as plain slice ids (looped kernels, classes of more than 32 entries per row)
};

// Device-resident SELL-64 sparsity + incidence lists.
struct DevPattern {
  int32_t n_slices = 0;
  int max_len = 0, max_clen = 0;
  dvec<int64_t> slice_ptr;
  dvec<int32_t> cols;
  dvec<uint16_t> cols16;
  dvec<int32_t> win_base;
  dvec<uint8_t> win_ok;
  dvec<uint8_t> diag_k;
  dvec<uint8_t> rlen;                      // entries of each row: the SpMV's lanes stop there (slots beyond are padding)
  dvec<int64_t> cslice_ptr;
  dvec<uint32_t> cslots;
  dvec<int32_t> celem;
  dvec<double> cw;                         // per-incidence reaction weight rho_T |T| d!/(d+3)!
  dvec<uint32_t> cs2;                      // slot word re-ordered for the hot kernels: byte 0 = the row's own (diagonal) slot, then the cell's other vertices
  dvec<uint32_t> cq;                       // [incidence][2] = (re-ordered slot word, weight as float bits): the quadratic-term pass's 8-byte records
  dvec<int32_t> interior_slices, boundary_slices;
  int32_t n_interior = 0, n_boundary = 0;
  std::vector<int> bucket_cap;
  std::vector<int32_t> bucket_count;
  std::vector<int32_t> bucket_interior;
  std::vector<dvec<int32_t>*> bucket_slices;   // owned; released in ~DevPattern
  std::vector<SliceClass*> classes;            // launch classes of the incidence-list kernels (built on first use, kernels.hip)
  ~DevPattern() {
    for (auto* b : bucket_slices) delete b;
    for (auto* c : classes) delete c;
  }
  int64_t total_entries = 0, total_corners = 0;
};

// Node-local all-reduce through a mailbox in host shared memory that every rank's GPU maps (fine-grained): slot
// [parity][rank] = {seq, v[0..3]} (64 B).  Fused into the final block of the reduction kernels: no extra launch and
// a few microseconds of PCIe latency instead of an RCCL kernel per Krylov iteration.  `seq` lives on the device and
// only advances when the kernel really runs (kernels skipped by the `done` word must not consume a number).
struct NodeMail {
  double* slots = nullptr;             // device address of the mapping; nullptr = not in use
  unsigned long long* seq = nullptr;   // device counter
  int* err = nullptr;                  // device flag: a peer did not arrive in time
  int rank = 0, world = 1;
  long long timeout_ticks = 60ll * 100000000ll;   // wall_clock64 runs at 100 MHz
};

// ---- elasticity multigrid (mg.hip) --------------------------------------------------------------------------
// Auxiliary-grid geometric multigrid for K_el: level 0 is the mesh itself (SELL-64 blocks, single-precision copy),
// level 1 a Cartesian grid of width H ~ 2h laid over the mesh (d-linear interpolation onto the mesh nodes), levels
// 2.. its 2:1 coarsenings.  Coarse operators are Galerkin products stored as dense stencils: no column indices on
// any coarse level, every stream unit-stride over the grid nodes.
struct MgGrid {
  int n[3] = {1, 1, 1};                    // nodes per axis (n[2] = 1 in 2-D)
  int64_t nn = 1;
};
struct MgLevel {                           // one Cartesian level
  MgGrid g;
  int f[3] = {1, 1, 1};                    // coarsening factor per axis towards the next level (1 | 2)
  int o[3] = {0, 0, 0};                    // global index of this box's node 0 (partitioned runs; 0 otherwise)
  int ng[3] = {1, 1, 1};                   // nodes per axis of the level's GLOBAL grid
  bool global = false;                     // replicated on every rank (box = whole grid), residual all-reduced per cycle
  // storage box of the OPERATOR planes (A / A16): lo[3], n[3]; sb_nn = 0: the whole grid.  Box-limited partitioned runs keep
  // the first grid's operator rows for the rank's work box only (vectors, dinv and sc stay whole-grid arrays)
  int sb[6] = {0, 0, 0, 0, 0, 0};
  long long sb_nn = 0;
  dvec<float> A;                           // [S][bs*bs][rows] stencil-major planes (rows = sb_nn or nn)
  bool half = false;                       // smoothed in symmetrically scaled variables with the half-precision copy A16
  dvec<uint16_t> A16;                      // [S][bs*bs][nn] _Float16 planes of S A S, S = diag(1 / sqrt(a_ii))   (half only)
  dvec<double> sc;                         // [bs][nn] the scale factors s; dinv then holds (S A S)_ii^-1       (half only)
  dvec<double> dinv;                       // [bs*bs][nn] inverse diagonal blocks
  dvec<double> x, x2, r, d, res;           // [bs][nn] (component-major)
  double lam = 1.0;                        // estimate of lambda_max(Dinv A)
};
// Neighbour exchange that replaces the all-reduce of the first grid's restricted residual in a box-limited partitioned
// cycle: rank p needs the sum on its work box only, and rank q's partial sums are nonzero only on q's core, so p receives
// (box_p n core_q) from q and sends (box_q n core_p) -- boxes every rank derives from the all-gathered cores.
struct GridRegion {
  int lo[3], n[3];
  int rank;              // the peer this region comes from / goes to
  long long off;         // first node of the region in the message buffers (nodes, not doubles)
};
struct GridExchange {
  bool ready = false;
  int deg = 0;                               // smoother degree the boxes were sized for (a larger one falls back to the all-reduce)
  std::vector<int32_t> peers;                // ascending rank
  std::vector<int64_t> send_ptr, recv_ptr;   // per peer, in nodes
  int n_send_reg = 0, n_recv_reg = 0;
  long long n_send = 0, n_recv = 0;          // nodes
  dvec<GridRegion> send_reg, recv_reg;       // ascending peer rank
  dvec<double> sendbuf, recvbuf;             // [node][bs]
  void reset() {
    ready = false;
    peers.clear();
    send_ptr.clear();
    recv_ptr.clear();
    n_send_reg = n_recv_reg = 0;
    n_send = n_recv = 0;
  }
};
struct MgHierarchy {
  bool ready = false;
  // The operator this hierarchy preconditions (set by the caller of gl_mg_setup; borrowed pointers that must stay valid
  // while `ready`): K_el with bs = dim (elasticity), or the static part S of the RD Jacobian with bs = 1.
  int bs = 0;                              // dofs per mesh node
  const double* op_vals = nullptr;         // fp64 SELL-64 planes [entry][bs * bs][64]
  const uint8_t* op_fixed = nullptr;       // constrained dofs [n_nodes * bs], nullptr = none
  const double* op_dinv = nullptr;         // fp64 inverse diagonal blocks of the constrained operator [n_own][bs * bs]
  dvec<double> own_dinv;                   // ... kept here when nobody else owns them (RD hierarchy)
  dvec<uint16_t> v16;                      // level 0: half-precision copy of S A S (bit pattern of _Float16)
  bool lattice = false;
  int n_levels = 0;                        // incl. the mesh itself
  int64_t cycles = 0;
  double complexity = 0.0, ms_setup = 0.0;
  int R = 1, S = 27;                       // stencil radius / entries of the Cartesian levels
  double lo[3] = {0, 0, 0}, H[3] = {1, 1, 1};
  dvec<int32_t> cell0;                     // [n_own] lower-corner grid node of the level-1 cell that holds a mesh node
  dvec<double> wgt;                        // [n_own][dim] interpolation weight towards the upper node, per axis
  dvec<int32_t> cell_ptr, cell_nodes;      // mesh nodes sorted by level-1 cell (children lists of the grid nodes)
  // restriction mesh -> grid as an explicit operator, SELL-64 over the grid nodes: (child, weight) pairs
  dvec<int64_t> pt_ptr;                    // [ceil(nn / 64) + 1]
  dvec<int32_t> pt_idx;
  dvec<float> pt_w;
  dvec<double> x, x2, d, res;              // level-0 work vectors [n_nodes*bs] (ghost slots stay zero)
  double lam0 = 1.0;
  double cheb_ratio = 30.0;                // the smoothers' interval is [lambda_max / cheb_ratio, lambda_max]
  double default_ratio = 30.0;             // ... its default for this mesh (mg_setup_t), used when glims_options.mg_cheb_ratio = 0
  double dropped_fraction = 0.0;           // mesh edges whose parents lie more than two grid cells apart (not in the coarse operators)
  bool half_smoother = true;               // level-0 smoother streams the half-precision copy of K_el
  bool x32 = false;                        // level-0 cycle vectors in single precision (XNode<BS, float> layout)
  bool exact_level0 = false;               // partitioned run with a global frame: level-0 passes see the ghosts (halo exchange)
  // partitioned run with a global frame, replicated first grid: each rank smooths only its work box of that grid
  bool boxed = false;
  int core[6] = {0, 0, 0, 0, 0, 0};        // this rank's core on the first grid: lo[3], hi[3] (inclusive) of the nodes its own mesh nodes interpolate from
  dvec<int> cores;                         // every rank's core [world][6] (owner rule of the first-grid -> second-grid restriction)
  double box_fraction = 1.0;               // work box / first grid at the set-up's smoother degree
  GridExchange gx;                         // first-grid residual: neighbour exchange instead of the all-reduce (boxed only)
  dvec<double> sc;                         // level-0 scaling s = 1 / sqrt(diag K_el) per dof (1 on constrained dofs)
  dvec<float> dinv0;                       // inverse diagonal blocks of the scaled operator S K S, single precision
  dvec<float> vK32s;                       // single-precision scaled copy (GLIMS_FLAG_MG_FP32_SMOOTHER only)
  dvec<double> rs;                         // scaled residual r~ = S r of the current cycle
  std::vector<MgLevel*> lv;                // owned
  dvec<double> coarse_inv;                 // dense inverse of the coarsest operator [nc][nc], nc = lv.back()->g.nn * bs
  int64_t entries = 0;                     // stored operator entries of the coarse levels (scalars)
  void clear() {
    for (auto* l : lv) delete l;
    lv.clear();
    ready = false;
  }
  ~MgHierarchy() { clear(); }
};
// what glims_create keeps of the mesh geometry for the multigrid set-up
struct MeshMetrics {
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  double h_lattice[3] = {0, 0, 0};         // lattice constant per axis if the nodes form a lattice
  bool lattice = false;
  double mean_edge = 0.0;
};

// halo payload of one owned row, written by the kernel that has just produced the row's value(s)
struct PackMap {
  const int32_t* ref = nullptr;    // [n_own]: -1 or index into ptr; nullptr = no fused packing
  const int32_t* ptr = nullptr;
  const int32_t* slot = nullptr;
  double* sendbuf = nullptr;
};

#ifdef __HIPCC__
// Node-local all-reduce of red[0..nq) (see NodeMail): called by every thread of the (single) final reduction block.
// Sums in rank order on every rank -> the same bits everywhere, hence identical decisions.
static __device__ __forceinline__ void node_allreduce(double* __restrict__ red, int nq, const NodeMail nm) {
  __syncthreads();   // red[] written by threads < nq
  if (threadIdx.x >= GL_WAVE) return;
  const int lane = threadIdx.x;
  unsigned long long seq = 0;
  if (lane == 0) {
    seq = *nm.seq + 1ull;
    *nm.seq = seq;
  }
  seq = __shfl(seq, 0, GL_WAVE);
  double* bank = nm.slots + (size_t)(seq & 1ull) * nm.world * 8;
  if (lane == 0) {
    double* mine = bank + (size_t)nm.rank * 8;
    for (int q = 0; q < nq; ++q) __hip_atomic_store(mine + 1 + q, red[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  bool ok = true;
  if (lane < nm.world) {
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(bank + (size_t)lane * 8);
    const long long t0 = wall_clock64();   // 100 MHz
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > nm.timeout_ticks) {   // default: a peer that is a minute late is not coming
        ok = false;
        break;
      }
    }
  }
  ok = __all(ok);
  __threadfence_system();
  if (lane < nq) {
    double t = 0.0;
    if (ok)
      for (int r = 0; r < nm.world; ++r)
        t += __hip_atomic_load(bank + (size_t)r * 8 + 1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
      t = __builtin_nan("");
    red[lane] = t;
  }
  if (!ok && lane == 0) *nm.err = 1;
}

template <int BS>
static __device__ __forceinline__ void pack_row(const PackMap& pm, int64_t row, const double* vals /*[BS]*/) {
  const int32_t r = pm.ref[row];
  if (r < 0) return;
  for (int32_t q = pm.ptr[r]; q < pm.ptr[r + 1]; ++q) {
    const int64_t k = pm.slot[q];
#pragma unroll
    for (int a = 0; a < BS; ++a) pm.sendbuf[k * BS + a] = vals[a];
  }
}

static __device__ __forceinline__ int xcd_chunk_remap(int b, int nb, int G) {
  // Hardware deals block b to XCD b % 8.  Give every XCD chunks of G consecutive logical blocks, chunk after chunk
  // round-robin over the XCDs: neighbouring slices (overlapping x gathers) share one L2, while the eight XCDs
  // together still walk the matrix front to back (one shared, moving x window in the Infinity Cache; DRAM pages are
  // visited nearly sequentially).  The blocks beyond the last full group of 8*G form one more group with chunks of
  // (what is left) / 8; the last < 8 blocks keep their index -> bijective.
  const int full = (nb / (8 * G)) * (8 * G);
  if (b < full) {
    const int q = b >> 3, xcd = b & 7;
    return ((q / G) * 8 + xcd) * G + (q % G);
  }
  const int g2 = (nb - full) >> 3;
  const int t = b - full;
  if (t >= 8 * g2) return b;
  return full + (t & 7) * g2 + (t >> 3);
}

// Level-0 cycle vectors of the elasticity multigrid.  XT = double: [node][BS] doubles.  XT = float (default): the
// ITERATE lives in node records of 16 B (BS = 3: x, y, z, pad) / 8 B (BS = 2), so that a neighbour's value is ONE
// aligned load in the smoother's gather instead of three 8-byte ones; residual / direction vectors are packed
// [node * BS] floats.  In units of doubles a record is GL_XREC(BS) = 2 / 1 long, which is what the halo exchange of a
// partitioned run is told.
template <int BS, class XT> struct XNode;
template <int BS> struct XNode<BS, double> {
  static __device__ __forceinline__ void load(const double* __restrict__ x, int64_t j, double* o) {
#pragma unroll
    for (int a = 0; a < BS; ++a) o[a] = x[j * BS + a];
  }
  static __device__ __forceinline__ void store(double* __restrict__ x, int64_t j, const double* v) {
#pragma unroll
    for (int a = 0; a < BS; ++a) x[j * BS + a] = v[a];
  }
};
template <> struct XNode<3, float> {
  static __device__ __forceinline__ void load(const float* __restrict__ x, int64_t j, double* o) {
    const float4 v = reinterpret_cast<const float4*>(x)[j];
    o[0] = (double)v.x;
    o[1] = (double)v.y;
    o[2] = (double)v.z;
  }
  static __device__ __forceinline__ void store(float* __restrict__ x, int64_t j, const double* v) {
    reinterpret_cast<float4*>(x)[j] = make_float4((float)v[0], (float)v[1], (float)v[2], 0.0f);
  }
};
template <> struct XNode<1, float> {
  static __device__ __forceinline__ void load(const float* __restrict__ x, int64_t j, double* o) { o[0] = (double)x[j]; }
  static __device__ __forceinline__ void store(float* __restrict__ x, int64_t j, const double* v) { x[j] = (float)v[0]; }
};
template <> struct XNode<2, float> {
  static __device__ __forceinline__ void load(const float* __restrict__ x, int64_t j, double* o) {
    const float2 v = reinterpret_cast<const float2*>(x)[j];
    o[0] = (double)v.x;
    o[1] = (double)v.y;
  }
  static __device__ __forceinline__ void store(float* __restrict__ x, int64_t j, const double* v) {
    reinterpret_cast<float2*>(x)[j] = make_float2((float)v[0], (float)v[1]);
  }
};

#endif
// (bs = 1 with single-precision vectors has no whole-double record: such a hierarchy keeps double vectors whenever its
// level-0 passes exchange halos, see mg_setup_t)
inline int gl_xrec_doubles(int bs, bool x32) { return x32 ? (bs == 3 ? 2 : 1) : bs; }

// scalar slots of the Krylov recurrence (device array `scal`)
enum { SC_ALPHA = 0, SC_BETA, SC_GAMMA, SC_IT, SC_COUNT = 8 };
#define GL_CG_HIST 64   // PCG iterations whose recurrence coefficients are recorded (Ritz values: spectral interval of the dot-free solves)

struct glims_ctx {
  int dim = 0, nv = 0, device = 0;
  int64_t n_nodes = 0, n_own = 0, n_cells = 0;
  hipStream_t st = nullptr, st_comm = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_pack = nullptr, ev_halo = nullptr;

  std::vector<int32_t> old2new, new2old;
  dvec<int32_t> d_old2new;
  dvec<double> xyz_new;                     // coordinates in the internal numbering [n_nodes][dim] (device-side symbolic phase)
  DevPattern pat;
  int64_t nnz = 0, n_corners = 0;

  dvec<uint8_t> label;                     // per cell, INTERNAL cell order (= celem's indices)
  dvec<int32_t> cell_new2old;              // internal cell index -> caller's (device-side symbolic phase: cells sorted by first owner; empty = identity)
  dvec<double> egeo;                       // per cell: |T|, grad(lambda_a) [nv][dim]
  dvec<double> evol;                       // per cell: |T| once more, compact (the per-incidence weights read 8 B, not a 104-B record)
  dvec<double> mat;                        // [5][GL_MAX_LABELS]: D, rho, gamma, mu, lambda
  bool have_materials = false, is_setup = false, have_mech = false, have_state = false;

  glims_options opt;
  glims_stats stats;
  bool use_idx16 = true;                   // columns as 16-bit (window, offset) codes where a slice has them
                                           // (off with GLIMS_FLAG_INT32_COLUMNS: the int32 stream everywhere)
  int64_t stats_defer_miss = 0;
  int64_t nnz_idx16_avail = 0;             // stored entries of slices that have 16-bit codes
  // glims_options.time_kernels: event pairs around the hot kernels of glims_step (bench.py's in-step roofline figures)
  enum { TK_SPMV = 0, TK_SWEEP = 1, TK_UPDATE = 2, TK_MGFINE = 3, TK_SPMVB = 4, TK_QUAD = 5, TK_CHEB = 6, TK_COUNT = 7 };
  std::vector<hipEvent_t> tev;
  std::vector<uint8_t> tev_cat;             // category of pair q = events 2q, 2q+1
  size_t tev_used = 0;
  // time_kernels = 1: the Krylov SpMV only (two event records per launch cost ~2 us each -- too much for the other two
  // kernels inside a timed region at 1 M rows); 2: all three categories
  // 3: the two dominant kernels of the elasticity solve instead (level-0 multigrid pass, block SpMV; eager launches)
  bool timing(int cat) const {
    const bool on = (cat == TK_MGFINE || cat == TK_SPMVB)
                        ? opt.time_kernels == 3
                        : (opt.time_kernels == 2 || (opt.time_kernels == 1 && (cat == TK_SPMV || cat == TK_CHEB)));
    return on && tev_used + 2 <= tev.size();
  }
  void timing_begin();                      // allocates the event pool on first use
  void timing_collect();                    // elapsed times of the recorded pairs -> stats (sums, counts, medians)
  void tick(int cat) {                      // first call opens a pair of category `cat`, the second closes it
    if ((tev_used & 1) == 0) tev_cat[tev_used / 2] = (uint8_t)cat;
    (void)hipEventRecord(tev[tev_used++], st);
  }
  // partitioned runs with time_kernels != 0: event pairs around every halo exchange on the communication stream (cev) and
  // around the compute stream's wait for it (wev) -> stats.ms_exchange / ms_exchange_exposed
  std::vector<hipEvent_t> cev, wev;
  size_t cev_used = 0, wev_used = 0;
  hipEvent_t* comm_pair() {
    if (!opt.time_kernels || cev_used + 2 > cev.size()) return nullptr;
    cev_used += 2;
    return &cev[cev_used - 2];
  }
  hipEvent_t* wait_pair() {
    if (!opt.time_kernels || wev_used + 2 > wev.size()) return nullptr;
    wev_used += 2;
    return &wev[wev_used - 2];
  }
  hipEvent_t* pair(int cat) {               // a pair to be attached to one dispatch (hipExtLaunchKernelGGL)
    tev_cat[tev_used / 2] = (uint8_t)cat;
    tev_used += 2;
    return &tev[tev_used - 2];
  }

  // scalar operator planes (SELL-64 layout) and block planes
  dvec<double> vM, vS, vA, vKel, vG;
  dvec<float> vA32;                        // Newton Jacobian in single precision (GLIMS_FLAG_FP32_JACOBIAN only)
  bool jac32 = false;
  int spmv_unroll = 8;                     // entries in flight per lane of the scalar SpMV: 8 on lattice meshes, 16 on general ones
  int stream_nt = 1;                       // value / column-code streams of the Krylov operator pass non-temporal (1) or with the
                                           // default cache policy (0): glims_options.stream_policy, resolved at glims_setup
  // Dot-free RD linear solves (Chebyshev semi-iteration, solver.hip): interval of the spectrum of Dinv A(c) the right-hand sides
  // of this run excite, from the Lanczos coefficients of recorded PCG solves
  struct ChebState {
    bool valid = false;                    // [lmin, lmax] usable
    // Two intervals: [1] from all PCG solves of the last learning step, [0] from its LOOSE ones only (reduction >= GL_CHEB_LOOSE):
    // what a tight solve has to resolve (components of relative size 1e-7 at the ends of the spectrum) a solve to 3e-4 may
    // ignore -- brain-like mesh at 1 M nodes: [0.70, 1.96] against [0.033, 3.63]
    double lmin = 0.0, lmax = 0.0;         // [1]
    double lmin0 = 0.0, lmax0 = 0.0;       // [0]; lmax0 = 0: none (then [1] serves)
    double acc_lmin = 0.0, acc_lmax = 0.0; // ... being accumulated by the current learning step
    double acc_lmin0 = 0.0, acc_lmax0 = 0.0;
    int learned = 0, learned0 = 0;         // PCG solves that contributed to acc_* / acc_*0
    int age = 0;                           // steps since the interval was measured
    int weak = 0;                          // consecutive dot-free solves that contracted far less than they were sized for
    int m_hint[2] = {0, 0};                // passes the device chose for the last warm-started solve (bounds the next one's launches):
                                           // [0] a step's first solve, [1] its second
    // Which iteration a solve AFTER a step's first one uses (the first, loose one always takes the dot-free iteration): PCG
    // needs fewer operator passes for a tight solve (superlinear convergence: 8 iterations where the Chebyshev bound asks for
    // 13-15 at config C4), the dot-free iteration cheaper ones.  cost_ratio = cost of a Chebyshev pass / cost of a PCG
    // iteration from a byte model of the two (solver.hip, cheb_cost_ratio: 0.74 at 10 M rows, 0.66 at 1.26 M -- measured 0.74 /
    // 0.6), pcg_its_per_decade from the tightest PCG solve of the last learning step.  Deterministic (no timings): the
    // iteration path of a run stays reproducible bit for bit.  0 = unknown (then: Chebyshev).
    double cost_ratio = 0.0, pcg_its_per_decade = 0.0;
    int pcg_best_its = 0;                  // iterations of the solve pcg_its_per_decade comes from
  } cheb;
  double cheb_test_hi = 1.0;               // TEST HOOK GLIMS_CHEB_TEST_SCALE_HI (read by glims_create): factor on the measured upper end
  dvec<double> cg_hist;                    // [2 * GL_CG_HIST] (alpha_k, beta_k) of the running PCG solve
  dvec<double> cheb_delta;                 // the correction the last Chebyshev solve added to the iterate [n_nodes] (take-back)
  dvec<double> cheb_delta2;                // the same of a step's SECOND solve: kept across the step boundary, it is the next step's guess
  bool have_d2 = false;                    // ... cheb_delta2 holds the previous step's second correction (same run, no jump in the state)
  dvec<double> d2_prev;                    // the second correction of the step before that one (linear extrapolation of the guess)
  int d2_depth = 0;                        // consecutive steps whose second correction was kept (2: d2_prev is a real one)
  int d2_regime = -1;                      // the step cheb_delta2 comes from: passes of its FIRST solve and the forcing mode (what that solve
                                           // left behind depends on them) ...
  double d2_r1 = 0.0;                      // ... and the Newton residual its second solve started from
  int d2_off = 0, d2_backoff = 8, d2_good = 0;   // steps for which the guess stays unused after one that missed the target; the length
                                           // doubles with every miss (8 .. 256) and returns to 8 after 32 guesses that did not
  dvec<double> cheb_dir;                   // the running solve's direction d [n_nodes]
  dvec<int> cheb_plan;                     // [1] iteration count computed on the device (a step's first, warm-started solve)
  dvec<float> vKel32;                      // single-precision copy of K_el (inner solves of the elasticity solver)
  // vectors (internal numbering; length n_nodes unless noted)
  dvec<double> c, c_old, b, load_rd, dinv;
  dvec<double> cg_p, cg_s, cg_u, cg_w, cg_r, cg_r2, b2;     // scalar CG work vectors; r2/b2: speculative next step
  // Newton residuals from the quadratic structure (k_rd_quad): the step's first iterate c_0 (whose Jacobian the solves
  // use), the iterate before the last solve, and the two staged vectors a = c_new + c_k - 2 c_0, delta = c_new - c_k
  dvec<double> nq_c0, nq_ck;
  dvec<float> nq_ad;                                          // (a, delta) pairs, single precision (see k_rd_quad)
  double nq_first_ratio = 1e-3;                               // residual contraction of the last step's first Newton iteration
  // default forcing: how a step's FIRST solve is run (gl_step).  0: tolerance 0.3 cg_rtol; 1: the same + midpoint correction;
  // 2: cg_rtol, no correction (for nw_hold steps after a step that took three iterations even with the correction)
  int nw_mode = 0, nw_hold = 0, nw_since = 0, nw_steps = 0;   // nw_steps: steps since glims_set_state
  int nq_skip_steps = 0;                                      // steps left without cheap evaluations (after a poor contraction)
  int cg_hint[8] = {0, 0, 0, 0, 0, 0, 0, 0};                  // PCG iterations of the k-th Newton solve of the previous step
  int mech_hint = 0;
  // history of solved elasticity problems (right-hand side, free-dof solution): the operator is linear and time
  // independent, so the least-squares fit of a new right-hand side by the stored ones gives the initial guess
  static constexpr int MHIST = 16;  // upper bound of glims_options.mech_history
  dvec<double> mh_rhs[MHIST], mh_x[MHIST], mh_w[MHIST];   // solve history: right-hand sides, solutions, K_el x (= rhs - final residual)
  int mh_count = 0, mh_next = 0;           // depth: glims_options.mech_history
  double mh_G[MHIST][MHIST] = {{0.0}};     // Gram matrix (rhs_k, rhs_l) of the stored right-hand sides (host copy)
  dvec<double> ws_du;                                         // the increment of the step before the last (warm start, k_ws_delta)
  int ws_depth = 0;                                           // ... 1 once it holds a real increment
  bool have_c_old = false;                                    // c_old holds the state at the start of the previous step
  bool pending = false;                                      // cg_r / b / vA already hold the first assembly of the next step
  double pending_r0 = 0.0;
  dvec<double> U, mload, m_rhs, m_p, m_s, m_u, m_w, m_r, m_dinv, m_uD;   // mechanics, [n_nodes*dim]
  dvec<uint8_t> fixed_c, fixed_u;
  bool have_fixed_c = false, have_fixed_u = false, have_load_rd = false, have_mload = false;
  MgHierarchy mg;                           // elasticity: K_el, 3 x 3 (2 x 2) blocks
  MgHierarchy mg_rd;                        // RD Jacobian: built on its static part S = (1 - dt rho) M + dt K_D, scalar
  int rd_precond_active = 0;                // what the RD solves use: GLIMS_RD_PRECOND_JACOBI | _MULTIGRID (decided at glims_setup)
  double rd_stiffness_ratio = 0.0;          // mean over the rows of S_ii / M_ii (the quantity `auto` decides on)
  double rd_break_even = 1e300;             // Jacobi-PCG iterations per Newton solve above which the hierarchy pays
  std::vector<uint8_t> fixed_c_host;        // host copy of the concentration's Dirichlet mask (did the SET of nodes change?)
  bool mg_frame_set = false;               // glims_set_mg_frame: global bounding box of a partitioned mesh
  double mg_frame_lo[3] = {0, 0, 0}, mg_frame_hi[3] = {0, 0, 0};
  MeshMetrics mm;
  dvec<double> fixed_c_val;                 // Dirichlet values of the concentration [n_nodes] (internal numbering)
  bool dirichlet_c_dirty = false;           // values not yet written into the iterate (done by the next step)
  // partitioned runs: SOME rank may have new Dirichlet values on nodes this rank holds as ghosts -- set by every
  // glims_set_dirichlet_c / glims_set_state (collective calls: every rank makes them, with n = 0 where it owns no constrained
  // node), so that the next step's halo exchange of the iterate happens on all ranks or on none
  bool dirichlet_c_exchange = false;
  std::vector<dvec<double>*> snapshots;     // device-resident recorded concentrations (owned)
  dvec<double> stage;                      // staging for host<->device permuted transfers [n_nodes*dim]

  dvec<double> partials, partials2;        // per-block partial sums (stage 1 / stage 2 of the reduction)
  dvec<double> partials_v;                 // (r.u, r.r) pairs per block of the PCG vector kernels
  dvec<double> partials_rr;                // (-, r.r) pairs of the vector update when the preconditioner is a V-cycle
  dvec<double> red;                        // [4] reduced sums ([3]: the early convergence check's |r|^2)
  dvec<double> scal;                       // [SC_COUNT] recurrence scalars
  dvec<int> done;                          // [1] 0 running, 1 converged, 2 non-finite, 3 breakdown
  double* h_pinned = nullptr;              // pinned, device-mapped mailbox [seq | red[4] | info[2] | done] (32 doubles)
  double* mail_dev = nullptr;              // its device address
  unsigned long long mail_seq = 0;

  // multi-GPU
  int rank = 0, world = 1;
  ncclComm_t comm_halo = nullptr, comm_red = nullptr;
  int n_peers = 0;
  std::vector<int32_t> peer_rank;
  std::vector<int64_t> send_ptr, recv_ptr;
  dvec<int32_t> send_idx;
  // row -> send slots (the vector-update kernel packs the halo payload itself): send_ref[row] = -1 or r with the
  // slots send_slot[send_slot_ptr[r] .. send_slot_ptr[r+1])
  dvec<int32_t> send_ref, send_slot_ptr, send_slot;
  dvec<double> sendbuf;
  int64_t n_send = 0;
  NodeMail nm;                              // active when nm.slots != nullptr
  void* nm_host = nullptr;                  // the mmap'ed segment
  size_t nm_bytes = 0;
  dvec<unsigned long long> nm_seq;
  dvec<int> nm_err;
  glims_halo_fn tr_halo = nullptr;          // host-provided transport (tests / MPI hosts); RCCL when null
  glims_allreduce_fn tr_allreduce = nullptr;
  void* tr_user = nullptr;

  std::string err;
};

// kernels.hip ---------------------------------------------------------------------------------------
void gl_compute_egeo(glims_ctx* h, const double* d_xyz, const int32_t* d_cells);
void gl_gather_u8(glims_ctx* h, int64_t n, const int32_t* idx, const uint8_t* in, uint8_t* out);   // out[i] = in[idx[i]]
void gl_translate_cells(glims_ctx* h, int64_t n, const int32_t* celem, const int32_t* new2old, int32_t* out);
void gl_assemble_static(glims_ctx* h, int with_mechanics);
int gl_rd_grid(const glims_ctx* h);
void gl_rd_quad(glims_ctx* h, const float* ad /*[n_nodes][2] = (a, delta)*/, double* r, double* partials /*[gl_rd_grid][2]*/,
                int part = 0 /*GL_PART_ALL*/);
int gl_spmv_grid(int n_launch);
enum { GL_PART_ALL = 0, GL_PART_INTERIOR = 1, GL_PART_BOUNDARY = 2 };   // slices without / with ghost columns
void gl_rd_assemble(glims_ctx* h, const double* c, const double* b, const double* b2, double* r_out, double* r2_out,
                    double* partials /*[gl_rd_grid][2]*/, int part = GL_PART_ALL);
void gl_spmv_scalar(glims_ctx* h, const double* vals, const double* x, double* y, bool masked);
void gl_apply_G(glims_ctx* h, const double* c, double* y);
void gl_rd_matfree(glims_ctx* h, const double* c, const double* x, double* y);
void gl_spmv_block(glims_ctx* h, const double* x, double* y, bool masked);
void gl_launch_spmv(glims_ctx* h, hipStream_t st, int n_launch, const int32_t* slice_list, const double* vals,
                    const double* x, double* y, const uint8_t* fixed, const double* addv, const double* r,
                    double* partials, int partial_off, const int* done, const float* vals32 = nullptr,
                    hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
void gl_launch_cheb(glims_ctx* h, hipStream_t st, int n_launch, const int32_t* slice_list, const double* vals,
                    const float* vals32, const double* y_in, double* y_out, double* b, const double* dinv, double* dvec,
                    double* ylast, double* x, const uint8_t* fixed, double c1, double c2, int k, int m_host, const int* plan,
                    int want_res, const PackMap& pm, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int shift = 0,
                    double* nrm = nullptr);
void gl_launch_spmv_block(glims_ctx* h, hipStream_t st, int n_launch, const int32_t* slice_list, const double* x,
                          double* y, const uint8_t* fixed, const double* r, double* partials, int partial_off,
                          const int* done, bool single_precision_operator = false);

// solver.hip ----------------------------------------------------------------------------------------
int gl_step(glims_ctx* h, int n_steps);
int gl_solve_mechanics(glims_ctx* h, const double* c_dev = nullptr);   // c_dev: concentration to use (default: state)
void gl_halo_exchange(glims_ctx* h, double* vec, int bs);   // blocking w.r.t. h->st (no overlap)
void gl_comm_destroy(glims_ctx* h);
int gl_comm_selftest(glims_ctx* h);
int gl_mailbox_selftest(glims_ctx* h);
int gl_project(glims_ctx* h, double* rhs_dev /*[n_nodes], overwritten*/, double* x_dev /*[n_nodes]*/, double rtol);
double gl_dot(glims_ctx* h, const double* a, const double* b, int64_t n, bool global = true);   // deterministic; host value
void gl_allreduce_bulk(glims_ctx* h, double* dev, size_t n);   // in-place sum over ranks of a device vector (RCCL / transport)
void gl_pair_of(glims_ctx* h, const double* u, float* ad /*[n_nodes][2]*/);   // (a, delta) = (u, u)
void gl_apply_dirichlet_c(glims_ctx* h);                                     // c[fixed] = stored values (+ halo)
void gl_block_dinv(glims_ctx* h);                                            // m_dinv of the constrained K_el

// mg.hip --------------------------------------------------------------------------------------------
void gl_mesh_metrics(glims_ctx* h, const HostPattern& hp, const double* xyz_old);
void gl_mg_setup(glims_ctx* h, MgHierarchy& mg);   // mg.bs / op_vals / op_fixed / op_dinv set by the caller
void gl_make_smoother_copy(glims_ctx* h, MgHierarchy& mg, bool half, bool exchange_scale);
// u = V-cycle(r) with Chebyshev smoothers of the given degree; r zero on constrained dofs
// pv (optional): the cycle's last kernel also leaves the per-block partial sums of (r, u) and (r, r), gl_spmv_grid(n_slices)
// pairs -- what the single-reduction PCG wants next
void gl_mg_apply(glims_ctx* h, MgHierarchy& mg, int degree, const double* r, double* u, const int* done = nullptr,
                 double* pv = nullptr);
void gl_make_kel32(glims_ctx* h);
// level-0 operator pass of the multigrid (kernels.hip): mode 0 out = r - A x, 1 Chebyshev step, 2 out = Dinv A x
void gl_launch_mg_fine(glims_ctx* h, MgHierarchy& mg, int mode, const double* xin, const double* r, double* d,
                       double* xout, double c1, double c2, const int* done = nullptr, double* uout = nullptr,
                       const double* r_full = nullptr, double* pv = nullptr, int part = 0);
int gl_mg_fine_blocks(glims_ctx* h, bool split);
bool gl_mg_split_level0(const glims_ctx* h, const MgHierarchy& mg);   // level-0 passes in two launches around the halo exchange
// generic grouped exchange with the handle's transport: per peer p, sendbuf[send_ptr[p] .. send_ptr[p+1]) * bs doubles go
// to peers[p] and recvbuf[recv_ptr[p] .. recv_ptr[p+1]) * bs come from it (empty directions are skipped on both sides)
void gl_exchange(glims_ctx* h, const std::vector<int32_t>& peers, const std::vector<int64_t>& send_ptr,
                 const std::vector<int64_t>& recv_ptr, const double* sendbuf, double* recvbuf, int bs);
void gl_halo_start(glims_ctx* h, double* vec, int bs);
void gl_halo_finish(glims_ctx* h);
// symbolic.hip: device sort helpers (rocPRIM)
void gl_sort_pairs_u32(glims_ctx* h, uint32_t* k_in, uint32_t* k_out, int32_t* v_in, int32_t* v_out, size_t n, int end_bit);
void gl_offsets_of_sorted_keys(glims_ctx* h, const uint32_t* keys_sorted, int64_t n, int64_t n_keys, int32_t* ptr);
// solver.hip: the two hierarchies
void gl_mg_setup_mech(glims_ctx* h);
void gl_mg_setup_rd(glims_ctx* h);
void gl_rd_choose_precond(glims_ctx* h);    // glims_options.rd_precond -> rd_precond_active (collective in partitioned runs)
