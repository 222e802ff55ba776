/*
 * glims_hip.h -- C-ABI of libglimship.so, the MI355X (gfx950) backend of the forward time-stepping
 * path of GlimSLib's TumorGrowth / TumorGrowthBrain models.
 *
 * The reference has no FFI of its own: its seam is the Python protocol of FenicsSimulation
 * (glimslib/simulation/simulation_base.py:36-158), where run() (:236-317) only touches
 * `self.solver.solve()` (:285/:302), `self.solution` and `u_previous.assign(self.solution)` (:312).
 * This library replaces what sits behind `self.solver.solve()` -- DOLFIN assemble + PETSc SNES/LU,
 * set up in glimslib/simulation/simulation_tumor_growth.py:78-140 and
 * glimslib/simulation/simulation_tumor_growth_brain.py:24-125 -- and the surrounding n-step loop.
 * Each entry point below names the reference interface it stands in for.
 *
 * Conventions
 *   - plain C, opaque handle, caller-owned host buffers (row-major, contiguous), no global state;
 *   - every function returns a status: GLIMS_OK (0), GLIMS_NOT_CONVERGED (1), GLIMS_NAN (2),
 *     or a negative GLIMS_E_* code; glims_last_error() gives the message of the last failure;
 *   - one handle = one device + its own HIP streams; a handle is not thread-safe, distinct handles are;
 *   - node/cell indices are 0-based and local to the handle (rank-local in a partitioned run);
 *   - displacement dofs are node-major interleaved: dof = node*dim + component
 *     (the reference's mixed space W = [P1^dim, P1], simulation_tumor_growth.py:67-72);
 *   - all floating point is IEEE fp64.
 *
 * Environment variables read by the library (everything else is a glims_options field):
 *   GLIMS_VERBOSE        any value: timing lines of the set-up phases and the solver's one-off decisions (preconditioner,
 *                        spectral interval of the dot-free solves, solves taken back) on stderr
 *   GLIMS_VERBOSE_CHEB   any value: one line per dot-free RD solve (residual before / after, tolerance, passes)
 *   GLIMS_HOST_THREADS   OpenMP team of the host-side symbolic phase (default: the CPUs this process may use,
 *                        divided by the ranks on the host)
 *   GLIMS_HOST_SYMBOLIC  TEST HOOK: glims_create runs the host (OpenMP) implementation of the symbolic phase instead of
 *                        the device one -- same structures, array by array (tests/test_gpu_symbolic.py)
 *   GLIMS_WIN_LIMIT      TEST HOOK: at most this many (<= 32) column windows per 64-row slice before a slice falls back
 *                        to 4-byte column indices -- lets tests/ exercise the mixed 16-bit / 32-bit index path on
 *                        meshes whose slices would all be compressible
 *   GLIMS_CHEB_TEST_SCALE_HI  TEST HOOK (read once by glims_create, per handle): factor on the measured upper end of the
 *                        spectral interval of the dot-free RD solves -- 0.5 makes their polynomial diverge on part of the
 *                        right-hand side, so that tests/ can exercise the take-back / PCG fallback path
 *   GLIMS_MG_BOX_MIN_NODES  TEST HOOK: smallest replicated first grid (nodes, at least 6 001) on which a partitioned run
 *                        limits each rank's smoothing to its work box (see GLIMS_FLAG_MG_WHOLE_GRID), instead of the
 *                        library's estimate of what that saves -- lets tests/ exercise the path on small meshes
 */
#ifndef GLIMS_HIP_H
#define GLIMS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLIMS_ABI_VERSION 6

enum {
  GLIMS_OK = 0,
  GLIMS_NOT_CONVERGED = 1,   /* Newton or Krylov hit its iteration cap (reference: exception from solver.solve(),
                                swallowed at simulation_base.py:303-305 -> "warn, stop, return last solution") */
  GLIMS_NAN = 2,             /* non-finite residual */
  GLIMS_E_USAGE = -1,        /* bad argument / wrong call order */
  GLIMS_E_HIP = -2,          /* HIP runtime error */
  GLIMS_E_RCCL = -3,         /* RCCL error */
  GLIMS_E_NO_DEVICE = -4     /* no usable gfx950 device: the library has no CPU fallback */
};

typedef struct glims_ctx glims_ctx;

/* Solver options.  The reference hard-codes 'snes' with DOLFIN defaults (simulation_tumor_growth.py:126-130:
 * SNES rtol 1e-9, atol 1e-10, max_it 50, linear_solver 'default' = sparse LU); this backend replaces LU by
 * Jacobi-preconditioned CG and therefore converges *tighter* than those defaults. */
typedef struct glims_options {
  double dt;              /* params.sim_time_step (simulation_tumor_growth.py:108) */
  double newton_rtol;     /* ||R_k||_2 <= max(newton_atol, newton_rtol*||R_0||_2);            default 1e-10 */
  double newton_atol;     /*                                                               default 1e-13 */
  int    newton_maxit;    /*                                                               default 50    */
  double cg_rtol;         /* RD linear solve: ||r||_2 <= max(cg_atol, cg_rtol*||R_k||_2,
                             0.5*newton target)   (inexact Newton forcing term)            default 1e-3  */
  double cg_atol;         /*                                                               default 0     */
  int    cg_maxit;        /*                                                               default 5000  */
  double mech_rtol;       /* mechanics PCG: ||r||_2 <= max(mech_atol, mech_rtol*||b||_2)    default 1e-10 */
  double mech_atol;       /*                                                               default 0     */
  int    mech_maxit;      /*                                                               default 200000*/
  int    check_every;     /* Krylov iterations enqueued between host convergence polls when the
                             iteration count cannot be predicted from the previous solve    default 8     */
  int    flags;           /* GLIMS_FLAG_*                                                   default WARM_START */
  /* ---- ABI 2: elasticity solver.  The reference gets its robustness for nu -> 0.5 from a sparse LU of the monolithic
   * system (simulation_tumor_growth.py:126-130) and names AMG as the alternative
   * (simulation_tumor_growth_brain_quad.py:116-119); here: PCG preconditioned by one multigrid V-cycle. */
  int    mech_precond;    /* GLIMS_PRECOND_BLOCK_JACOBI | GLIMS_PRECOND_MULTIGRID                default MULTIGRID */
  int    mech_mixed;      /* inner PCG streams a single-precision copy of K_el under an fp64 iterative-refinement
                             loop: 0 off, 1 auto (block-Jacobi preconditioner and K_el larger than the Infinity
                             Cache; never with the multigrid preconditioner), 2 always              default 1     */
  int    mech_history;    /* right-hand sides / solutions of the last k solves kept for the least-squares initial
                             guess (K_el is linear and time independent), 0..16                     default 8     */
  int    mg_smooth;       /* Chebyshev degree of the pre- and of the post-smoother on every level
                             (1 = damped block-Jacobi)                                             default 3     */
  int    mg_coarse_nodes; /* coarsen until a grid has at most this many nodes; that level is solved with a dense
                             inverse computed once (Gauss-Jordan on the device)                     default 216   */
  double mg_h_factor;     /* spacing of the first auxiliary Cartesian grid in units of the mesh width (lattice meshes:
                             of the lattice constant per axis); 0 = 2 on one GPU, and in partitioned runs with a global
                             frame (glims_set_mg_frame), whose Cartesian levels are replicated on every rank, 2 / 3 / 4
                             for <= 2 / <= 6 / more ranks                                          default 0     */
  double mg_cheb_ratio;   /* the Chebyshev smoothers act on [lambda_max / ratio, lambda_max] of Dinv A;
                             0 = by mesh class: 30 on lattice meshes and on general meshes of bounded node spacing,
                             10 on meshes with long edges / slivers (measured)                       default 0     */
  int    time_kernels;    /* HIP-event pairs on the handle's stream around hot kernels of glims_step: 1 = the Krylov
                             SpMV, 2 = also the assembly sweep and the PCG vector update; 3 = instead the level-0
                             multigrid pass and the block SpMV of glims_solve_mechanics; results in glims_stats.*_steps / *_mech /
                             us_*_median (bench.py's roofline figures)                                default 0     */
  /* ---- ABI 3: preconditioner of the RD linear solves.  The reference's LU does not care how stiff a step is
   * (simulation_tumor_growth.py:126-130); Jacobi-PCG needs ~sqrt(2 dt D / h^2 ...) iterations per Newton solve, e.g.
   * 66 / 129 on the unit cube with D = 0.1, dt = 1 at n = 32 / 64 (BASELINE config C2 is such a case), 4-5 on the
   * mass-dominated brain-extent configs C3 / C4. */
  int    rd_precond;      /* GLIMS_RD_PRECOND_AUTO | _JACOBI | _MULTIGRID.  MULTIGRID: one V-cycle of the same auxiliary-grid
                             hierarchy as the elasticity solver's, with 1 x 1 blocks, built once per glims_setup on the
                             static part S = (1 - dt rho) M + dt K_D of the Jacobian (S^-1 A(c) has its spectrum in
                             [1, 1 + 2 dt rho c / (1 - dt rho)]).  AUTO decides at the first glims_step after glims_setup
                             from q = mean_i S_ii / M_ii: multigrid when the predicted Jacobi count sqrt(2 q) exceeds the
                             break-even (20 for >= 400 k rows, 45 for >= 50 k, 90 below); the choice and q are in
                             glims_stats.rd_precond_used / rd_stiffness_ratio.  General (non-lattice) meshes: four
                             times those break-even counts (the cycle is weaker and dearer there), and because the
                             prediction is poor on such meshes, AUTO also switches to the hierarchy after any step
                             whose OBSERVED Jacobi count per solve exceeds the break-even                 default AUTO  */
  int    rd_mg_smooth;    /* Chebyshev degree of the RD hierarchy's smoothers (1 = damped Jacobi); 0 = by mesh class: 1 on
                             lattice meshes, 3 on general ones.  Measured on the unit cube with D = 0.1, dt = 1 at 0.1 / 1 /
                             10 M rows: degree 1 on [lambda/10, lambda] 1.9 / 4.3 / 26.3 ms per step, degree 3 on
                             [lambda/30, lambda] 2.9 / 7.4 / 46.2 (Jacobi-PCG: 4.4 / 26.5 / 486); on a Delaunay mesh of
                             200 k random points degree 1 / 3: 77 / 40 iterations per solve.  mg_cheb_ratio = 0 means 10
                             for this hierarchy                                                           default 0     */
  /* ---- ABI 6: the Krylov iteration of the Jacobi-preconditioned RD solves, and how its operator streams are cached.  PETSc's
   * counterpart is the KSP type (-ksp_type cg | chebyshev) behind solver.parameters (simulation_tumor_growth.py:126-130). */
  int    rd_linear;       /* GLIMS_RD_LINEAR_AUTO | _PCG | _CHEBYSHEV.  CHEBYSHEV: the dot-free iteration -- ONE kernel per
                             iteration (operator pass with the Chebyshev recurrence in its epilogue; a step's first solve takes the
                             warm-start guess as its first iterate, so the product A u is that solve's first pass), no dot
                             product, no reduction kernel, no all-reduce in partitioned runs.  The iteration count follows from
                             the wanted reduction and an interval [lmin, lmax] of Dinv A(c) chosen from the spectral measure of the
                             run's own right-hand sides (Lanczos coefficients of PCG solves: all solves of the first step after
                             glims_set_state and of every 32nd step run PCG and (re)measure it; loose and tight solves keep
                             intervals of their own).  Every solve is followed by a Newton residual evaluation anyway: one that
                             did not contract is taken back, the interval dropped and the iteration repeated with PCG
                             (glims_stats.cheb_fallbacks); two that contract far less than sized for bring the next learning step
                             forward; a solve whose bound exceeds 48 passes runs PCG.  AUTO = CHEBYSHEV wherever the RD solves
                             are Jacobi-preconditioned, except that a tight solve goes to PCG when a byte model of the two
                             iterations and PCG's measured rate say it is cheaper; solves with the multigrid preconditioner
                             always use PCG.  The choice never depends on timings: runs are bitwise reproducible
                                                                                                          default AUTO  */
  int    stream_policy;   /* GLIMS_STREAM_AUTO | _NONTEMPORAL | _CACHED: cache policy of the operator's value / column-code
                             streams in the Krylov operator pass.  Non-temporal keeps the gathered vector in L2 when the
                             operator is far larger than the 256 MiB Infinity Cache (config C4: +3 %); default-policy loads
                             let a small operator (a 1/8 share of C4 per GPU) stay resident between the passes of a solve.
                             AUTO picks by the working set at glims_setup              default AUTO  */
} glims_options;

#define GLIMS_PRECOND_BLOCK_JACOBI 0
#define GLIMS_PRECOND_MULTIGRID 1
#define GLIMS_RD_PRECOND_AUTO 0
#define GLIMS_RD_PRECOND_JACOBI 1
#define GLIMS_RD_PRECOND_MULTIGRID 2
#define GLIMS_RD_LINEAR_AUTO 0
#define GLIMS_RD_LINEAR_PCG 1
#define GLIMS_RD_LINEAR_CHEBYSHEV 2
#define GLIMS_STREAM_AUTO 0
#define GLIMS_STREAM_NONTEMPORAL 1
#define GLIMS_STREAM_CACHED 2

#define GLIMS_FLAG_EXTRAPOLATE_GUESS 1  /* Newton guess c^n + (c^n - c^{n-1}) instead of c^n (reference: c^n) */
#define GLIMS_FLAG_FP32_JACOBIAN 4       /* OFF by default.  The Newton Jacobian A(c) is stored and streamed in single
                                           precision inside the Krylov solves (products, sums, all vectors and the Newton
                                           residual stay fp64, so the iteration still converges to the fp64 tolerance of
                                           the same fixed point); takes effect at glims_setup */
#define GLIMS_FLAG_MG_FP32_SMOOTHER 8     /* OFF by default.  The level-0 smoother of the elasticity multigrid streams a
                                           single-precision copy of K_el instead of the (scaled) half-precision one, and
                                           the first Cartesian grid keeps its single-precision stencils */
#define GLIMS_FLAG_MG_FP64_VECTORS 32    /* OFF by default.  The level-0 cycle vectors of the elasticity multigrid (iterate,
                                           direction, scaled residual) are kept in double instead of single precision; the
                                           preconditioned residual handed to the Krylov solver is double either way */
#define GLIMS_FLAG_FULL_NEWTON 128       /* OFF by default.  Every Newton iteration of the RD block re-assembles Jacobian and
                                           residual in a sweep.  Default: in the middle of a time step (after its second,
                                           third ... solve, unless convergence is expected) the residual follows from the
                                           Krylov solver's final residual plus the exactly quadratic term dt N(a) delta -- one
                                           pass over the incidence lists, the Jacobian of the previous sweep stays in use; the
                                           sweep after the first solve and the one that confirms convergence (true residual,
                                           next step's Jacobian) still run, and an iteration whose residual exceeds 5 x the
                                           linear tolerance falls back to sweeps */
#define GLIMS_FLAG_MG_WHOLE_GRID 64      /* OFF by default.  Partitioned runs with glims_set_mg_frame: every rank smooths the
                                           WHOLE replicated first grid instead of its work box (its part plus the smoothers'
                                           dependency margin) -- the same preconditioner up to rounding, more work per rank */
#define GLIMS_FLAG_INT32_COLUMNS 16       /* OFF by default.  Stream the 4-byte column indices everywhere instead of the 16-bit
                                           (window, offset) codes (same bits in every result; takes effect at glims_setup) */
#define GLIMS_FLAG_FIXED_FORCING 256      /* OFF by default.  Every linear solve of the RD Newton iteration is asked for cg_rtol x
                                           its right-hand side and no right-hand side gets the midpoint correction (steps of
                                           three to four Newton iterations: the textbook inexact Newton the tests compare
                                           with).  Default: a step's FIRST solve runs at 0.3 cg_rtol, with that correction once
                                           steps start taking a third iteration; from the second solve on the tolerance follows
                                           the quadratic remainder the iteration is about to leave (q |R_k|^2 / |R_0|, q observed
                                           in the step's first iteration) -- two Newton iterations per step */
#define GLIMS_FLAG_MG_NO_LUMPING 512      /* OFF by default.  Mesh -> first-grid Galerkin product of the multigrid hierarchies: mesh
                                           edges whose end points' parents lie more than the stencil radius apart lose their
                                           cross terms only.  Default: a far POSITIVE coupling is lumped onto the two rows' own
                                           diagonals (the edge's term taken out whole: positive semi-definite, row sums kept) --
                                           sliver meshes 51 / 45 / 35 -> 48 / 39 / 32 iterations; lattices and quality-controlled
                                           meshes have no such edges and are not affected (rebuilds the hierarchies) */
#define GLIMS_FLAG_WARM_START 2         /* first linear solve of a step starts from the increment extrapolated from the previous two steps',
                                           its second solve from the second correction extrapolated likewise (round 5; only where the
                                           previous steps' continue each other); iteration counts change, what a
                                           step returns does not -- the Newton tolerance decides that
                                           (ignored when GLIMS_FLAG_EXTRAPOLATE_GUESS is set) */

typedef struct glims_stats {
  int64_t steps;            /* implicit time steps taken */
  int64_t newton_its;       /* Newton linear solves (RD) */
  int64_t rd_assemblies;    /* Jacobian+residual assembly sweeps */
  int64_t cg_its;           /* RD Krylov iterations = SpMV applications of A(c) */
  int64_t mech_solves;
  int64_t mech_cg_its;
  double  last_newton_res;  /* ||R||_2 at exit of the last step */
  double  last_cg_res;
  double  last_mech_res;
  double  ms_steps;         /* device time of glims_step calls (HIP events on the handle's stream) */
  double  ms_spmv;          /* accumulated by glims_apply only */
  int64_t n_rows;           /* owned rows */
  int64_t nnz;              /* structural nonzeros of the scalar operator (unpadded) */
  int64_t nnz_padded;       /* stored SELL-64 entries */
  int64_t n_corners;        /* (row, cell) incidences */
  int64_t nnz_idx16;        /* stored entries whose column is streamed as a 16-bit (window, offset) code */
  double  ms_spmv_steps;    /* time_kernels = 1 only: HIP-event time of the Krylov SpMV launches inside glims_step */
  int64_t n_spmv_steps;     /*   ... and how many of them really ran (launches skipped behind the decision word excluded) */
  /* ---- ABI 2 */
  int64_t failed_steps;     /* time steps whose Newton / Krylov solve gave up (not counted in `steps`) */
  int64_t mg_levels;        /* levels of the elasticity multigrid hierarchy incl. the mesh itself (0 = not built) */
  int64_t mg_cycles;        /* V-cycles applied */
  double  mg_complexity;    /* stored operator entries of all levels / entries of K_el */
  double  ms_mg_setup;      /* wall time of the last hierarchy set-up (transfer lists, Galerkin products, eigenvalue
                               estimates, coarse inverse) */
  double  ms_mech;          /* wall time spent in elasticity solves */
  /* time_kernels = 1 only; launches skipped behind the decision word (a few microseconds) are left out */
  double  ms_sweep_steps;   /* assembly sweeps (k_rd_assemble) inside glims_step */
  int64_t n_sweep_steps;
  double  ms_update_steps;  /* PCG vector updates (k_cg_update) of the RD solves inside glims_step */
  int64_t n_update_steps;
  double  us_spmv_median;   /* medians over the last glims_step call */
  double  us_sweep_median;
  double  us_update_median;
  /* ---- ABI 3 */
  int64_t rd_precond_used;  /* GLIMS_RD_PRECOND_JACOBI | _MULTIGRID: what the RD solves really use (0 = not decided yet) */
  double  rd_stiffness_ratio; /* q = mean over the rows of S_ii / M_ii, the number `auto` decides on */
  int64_t rd_mg_levels;     /* levels of the RD hierarchy incl. the mesh (0 = not built) */
  int64_t rd_mg_cycles;     /* V-cycles applied inside glims_step */
  double  rd_mg_complexity; /* stored operator entries of all levels / entries of S */
  double  ms_rd_mg_setup;   /* wall time of the last set-up of the RD hierarchy */
  /* time_kernels = 3 only: the two dominant kernels of glims_solve_mechanics (HIP events, eager launches) */
  double  ms_mgfine_mech;   /* level-0 passes of the elasticity multigrid (k_mg_fine) */
  int64_t n_mgfine_mech;
  double  us_mgfine_median;
  double  ms_spmvb_mech;    /* block SpMV of the Krylov iteration (k_spmv_block2) */
  int64_t n_spmvb_mech;
  double  us_spmvb_median;
  /* ---- ABI 4: Newton residuals from the quadratic structure of the RD residual (see GLIMS_FLAG_FULL_NEWTON) */
  int64_t rd_quad_updates;  /* residual evaluations through the incidence lists only (no sweep, no Jacobian) */
  double  ms_quad_steps;    /* time_kernels = 2: those passes (k_rd_quad), HIP events */
  int64_t n_quad_steps;
  double  us_quad_median;
  /* ---- ABI 5: which paths of the Newton iteration ran (the re-entrancy / parity tests assert on these) */
  int64_t midpoint_steps;   /* time steps whose first right-hand side carried the midpoint correction -dt N(u) u */
  int64_t rebase_events;    /* Newton iterations after which a cheap residual exceeded 5 x the linear tolerance: fresh Jacobian,
                               sweeps only for the next steps */
  /* ---- ABI 5: partitioned runs, per rank (what the first multi-GPU run needs to say where its time went) */
  int64_t halo_exchanges;   /* neighbour exchanges started by glims_step / glims_solve_mechanics (mesh halos and first-grid boxes) */
  int64_t halo_bytes;       /* bytes this rank SENT in them */
  double  ms_exchange;      /* time_kernels != 0: duration of the exchanges on the communication stream (pack -> last receive),
                               HIP events; 0 with a host transport (glims_set_transport: the callback is synchronous) */
  double  ms_exchange_exposed; /* time_kernels != 0: part of it the compute stream really waited for (not hidden behind the
                               interior slices) */
  int64_t allreduces;       /* scalar all-reduces of the Krylov / Newton iterations */
  int64_t reduce_transport; /* how they travel: 0 single rank, 1 node mailbox (inside the reduction kernel), 2 ncclAllReduce,
                               3 host callback */
  int64_t mg_grid1_bytes;   /* elasticity multigrid: bytes of the first Cartesian grid's operator held by this rank */
  /* ---- ABI 6 */
  int64_t halo_exchanges_timed; /* exchanges that carried an event pair (ms_exchange / this = time per exchange; the event pools hold
                               4096 pairs per glims_step call, exchanges of the multigrid's first grid and of host transports
                               carry none) */
  int64_t cheb_solves;      /* RD linear solves by the dot-free Chebyshev iteration (glims_options.rd_linear) */
  int64_t cheb_its;         /* ... their operator passes (also counted in cg_its: "Krylov iterations") */
  int64_t cheb_fallbacks;   /* ... taken back because the Newton residual did not contract; repeated with PCG */
  int64_t cheb_learn_solves;/* PCG solves whose Lanczos coefficients (re)measured the interval */
  double  cheb_lmin;        /* the interval of Dinv A(c) in use (0 = none yet) */
  double  cheb_lmax;
  double  ms_cheb_steps;    /* time_kernels != 0: launches of the dot-free iteration's kernel (k_cheb) inside glims_step */
  int64_t n_cheb_steps;
  double  us_cheb_median;
  int64_t stream_nontemporal; /* what stream_policy resolved to: 1 non-temporal, 0 cached */
  int64_t krylov_working_set; /* bytes the operator pass + vector update of one Krylov iteration touch (what AUTO decides on) */
  double  mg_box_fraction;  /* elasticity multigrid, partitioned runs: this rank's work box / the replicated first grid (1 = the
                               whole grid: not box-limited); with parts that are boxes (recursive coordinate bisection) ~ 1 / ranks
                               + the smoothers' margin */
} glims_stats;

/* ---- lifetime -------------------------------------------------------------------------------------- */

/* Builds the device-resident discretisation for one mesh (rank-local sub-mesh in a partitioned run):
 * renumbering, SELL-64 sparsity, (row, cell) incidence lists, per-cell geometry.
 * Stands in for FenicsSimulation.__init__ + _setup_functionspace (simulation_base.py:91-109,
 * simulation_tumor_growth.py:67-72) and the cell-subdomain MeshFunction (helper_classes.py:402-444).
 *   dim        2 (triangles) or 3 (tetrahedra);  cells are [n_cells][dim+1] vertex indices
 *   n_own      nodes [0, n_own) are owned rows; nodes [n_own, n_nodes) are ghosts (single GPU: n_own == n_nodes)
 *   cell_label tissue id per cell, 0 <= label < 256
 *   device     HIP device ordinal */
int glims_create(glims_ctx** out, int dim, int64_t n_nodes, int64_t n_own, int64_t n_cells,
                 const double* xyz, const int32_t* cells, const int32_t* cell_label, int device);
int glims_destroy(glims_ctx* h);
const char* glims_last_error(const glims_ctx* h);   /* h may be NULL: error of the last failed glims_create */
int glims_abi_version(void);

/* ---- model data ------------------------------------------------------------------------------------ */

/* Per-label coefficient tables (what DiscontinuousScalar.eval_cell looks up per cell, helper_classes.py:47-58,
 * or TumorGrowthBrain's per-tissue constants, simulation_tumor_growth_brain.py:29-39,82-104).
 * All arrays have n_labels entries; label l of glims_create indexes them. */
int glims_set_materials(glims_ctx* h, int n_labels, const double* D, const double* rho,
                        const double* gamma, const double* E, const double* nu);
int glims_options_default(glims_options* opt);
int glims_set_options(glims_ctx* h, const glims_options* opt);

/* Dirichlet data (fenics.DirichletBC lists built at helper_classes.py:632-723).  n == 0 clears. */
int glims_set_dirichlet_u(glims_ctx* h, int64_t n, const int64_t* dof_ids, const double* values);
/* Concentration: the listed nodes are held at `values` from the next glims_step on.  As with fenics.DirichletBC the
 * data constrain the UNKNOWN of a step: the state that enters the step's 'u_previous' term keeps the values it has
 * (simulation_tumor_growth.py:115-117), the new values are written into the Newton iterate.  Time-dependent data: call
 * again before the next glims_step, as BoundaryConditions.time_update_bcs does for every step
 * (helper_classes.py:839-859).  Partitioned runs: a COLLECTIVE call -- every rank makes it, with its own constrained nodes
 * (n = 0 where it owns none): the next glims_step exchanges the halo of the iterate on all ranks. */
int glims_set_dirichlet_c(glims_ctx* h, int64_t n, const int64_t* node_ids, const double* values);

/* Load vectors already integrated by the host (NULL clears):
 *   rd_load[n_nodes]        = dt*( int s phi_i dx + oint g D phi_i ds )   (simulation_tumor_growth.py:119-120)
 *   mech_load[n_nodes*dim]  = int f.v dx + oint g.v ds                    (simulation_tumor_growth.py:112-113) */
int glims_set_rd_load(glims_ctx* h, const double* rd_load);
int glims_set_mech_load(glims_ctx* h, const double* mech_load);

/* (Re)assembles the time-independent operators for the current materials and dt:
 * M, S = (1 - dt rho) M + dt K_D, the per-incidence reaction weights, and -- if with_mechanics --
 * K_el (dim x dim blocks) and the coupling operator G.  Stands in for _setup_problem
 * (simulation_tumor_growth.py:78-140); cheap enough to call again after a parameter change
 * (run_for_adjoint, :142-155). */
int glims_setup(glims_ctx* h, int with_mechanics);

/* ---- state + time stepping --------------------------------------------------------------------------- */

/* c[n_nodes], u[n_nodes*dim] (u may be NULL = zero).  create_initial_value_function (helper_classes.py:983-986). */
int glims_set_state(glims_ctx* h, const double* c, const double* u);
int glims_get_state(glims_ctx* h, double* c, double* u);   /* either may be NULL */

/* `steps` counts converged steps only; a step whose solve gave up is counted in `failed_steps`, ends the call with
 * GLIMS_NOT_CONVERGED / GLIMS_NAN and leaves the iterate where the solver stopped (reference: simulation_base.py:301-305).
 * n_steps backward-Euler steps of the concentration equation, entirely on the device
 * (the `while` loop body solver.solve() + u_previous.assign, simulation_base.py:297-312, for the F_rd block). */
int glims_step(glims_ctx* h, int n_steps);

/* Solves K_el u = G c + f for the current concentration (the F_m block; it is linear in u and does not
 * feed back into c, simulation_tumor_growth.py:110-120, so it is only needed at recorded steps).
 * After the Krylov iteration the residual is recomputed with the fp64 operator (one pass) and the iteration continues
 * from it should it miss the tolerance; glims_stats.last_mech_res is that true residual. */
int glims_solve_mechanics(glims_ctx* h);

int glims_get_stats(const glims_ctx* h, glims_stats* st);
int glims_reset_stats(glims_ctx* h);

/* ---- operator hooks (tests, roofline) ---------------------------------------------------------------- */

/* y = Op x, repeated `reps` times on the handle's stream and timed with HIP events (ms_total out, may be NULL).
 * which: 0 = current RD Jacobian A(c), 1 = S, 2 = M (scalar, [n_nodes]);  3 = K_el, ([n_nodes*dim], unconstrained);
 *        4 = G (x [n_nodes] -> y [n_nodes*dim]);  5 = A(c) through the kernel variant with the Krylov iteration's
 *        fused dot product;  7 = the MATRIX-FREE product (S + 2 dt N(c)) x rebuilt from the (row, cell) incidence lists
 *        for the current state c (equals which = 0 once A(c) has been assembled for that state; not used by the solver:
 *        measured 4.2x slower than the assembled product at 10 M rows -- 1.38 ms and 5.15 GB against 0.33 ms and 1.77 GB --
 *        profiles/r02_matfree_ab.txt);  8 = the assembly sweep at c = x with a zero right-hand side, y = -1/2 (A(x) + S) x
 *        (timing hook, tools/ab_sweep.py; with a state set, A(c) and its diagonal are re-assembled before the call returns);
 *        9 = the quadratic-term pass with a = delta = x, y -= dt N(x) x per repetition (timing hook).
 *        Ghost rows of y are returned as 0. */
int glims_apply(glims_ctx* h, int which, const double* x, double* y, int reps, double* ms_total);

/* Assembles A(c) and the Newton residual R(c; c_prev) for host vectors (ghost rows 0):
 *   R = 1/2 (A(c) + S) c - M c_prev - rd_load.  R may be NULL. */
int glims_rd_residual(glims_ctx* h, const double* c, const double* c_prev, double* R);

/* Diagnostics of the symbolic phase (node renumbering, SELL-64 sparsity, incidence lists, column codes -- built on the
 * device by glims_create; DOLFIN's counterpart is the dofmap + AIJ preallocation behind fenics.FunctionSpace,
 * helper_classes.py:271-282).
 *   glims_get_numbering     old2new[n_nodes]: internal index of the caller's node i (ghosts keep their index)
 *   glims_pattern_checksum  64-bit FNV-1a hashes of the device arrays, in this order: slice offsets, columns, 16-bit
 *                           column codes, window bases, window flags, diagonal slots, incidence offsets, incidence slots,
 *                           incidence cells, interior slice list, boundary slice list, numbering, row lengths (13 values).  Two handles
 *                           with equal checksums hold identical discretisation structures (tests compare the device-built
 *                           structures with the host implementation kept behind the test hook GLIMS_HOST_SYMBOLIC). */
int glims_get_numbering(glims_ctx* h, int32_t* old2new);
int glims_pattern_checksum(glims_ctx* h, uint64_t out[13]);

/* ---- device-resident time series ----------------------------------------------------------------------------------
 * Results.add_to_results deep-copies the mixed solution on the host at every recorded step
 * (helper_classes.py:1128-1144, called from simulation_base.py:306-309).  With 288 GB of HBM the recorded
 * concentration fields can stay on the device instead (80 MB per step at 10 M nodes), and -- because the displacement of
 * a step depends only on that step's concentration (simulation_tumor_growth.py:110-120) -- the elastic solve of a
 * recorded step can be deferred until somebody asks for its displacement.
 *   glims_snapshot_save       copies the current concentration into a new device buffer, returns its id (>= 0)
 *   glims_snapshot_load       concentration of snapshot `id` -> host c[n_nodes]
 *   glims_snapshot_mechanics  solves K_el u = G c_id + f for that snapshot -> host u[n_nodes*dim]; the time-stepping
 *                             state is not touched (status as glims_solve_mechanics)
 *   glims_snapshot_clear      releases all snapshots */
int glims_snapshot_save(glims_ctx* h, int64_t* id_out);
int glims_snapshot_load(glims_ctx* h, int64_t id, double* c);
int glims_snapshot_mechanics(glims_ctx* h, int64_t id, double* u);
int glims_snapshot_clear(glims_ctx* h);

/* L2 projection onto the P1 space: solves M x = rhs for `ncomp` right-hand sides stored [n_nodes][ncomp]
 * (rhs_i = int f phi_i dx, integrated by the caller), Jacobi-PCG to ||r|| <= rtol*||rhs|| per component.
 * Stands in for fenics.project(expr, FunctionSpace(mesh, "Lagrange", 1)) as used by the PostProcess classes
 * (helper_classes.py:1566-1618, 1736-1786).  Ghost entries of rhs are ignored, ghosts of x are filled. */
int glims_project(glims_ctx* h, const double* rhs, double* x, int ncomp, double rtol);

/* ---- single-node multi-GPU (one process per GPU, RCCL over xGMI) ------------------------------------- */

#define GLIMS_UNIQUE_ID_BYTES 256   /* two RCCL unique ids: halo communicator + reduction communicator */
int glims_comm_unique_id(char id[GLIMS_UNIQUE_ID_BYTES]);           /* rank 0; broadcast by the host program */
int glims_comm_init(glims_ctx* h, int rank, int world, const char id[GLIMS_UNIQUE_ID_BYTES]);

/* Node-local all-reduce for the scalars of the Krylov / Newton iterations: all ranks of ONE host map the POSIX
 * shared-memory object `shm_name` ("/name", created on first use, >= world * 128 bytes) into their GPU's address
 * space; the final block of each reduction kernel publishes its partial sums there and adds up everybody's in rank
 * order.  No launch and no RCCL kernel per iteration (the all-reduce of three doubles is pure latency; the halo
 * exchange stays on RCCL / xGMI).  Call after glims_comm_init or glims_set_transport (they set rank / world), from
 * every rank with the same name; the caller unlinks the object once every rank has returned.  Without this call the
 * reductions use ncclAllReduce (or the transport's allreduce callback).  shm_name = NULL switches it off again.
 * New in this build; the reference's counterpart is PETSc's MPI_Allreduce inside KSP/SNES. */
int glims_comm_mailbox(glims_ctx* h, const char* shm_name);
/* Collective check of the mailbox: one all-reduce of known values with a 5 s limit; every rank calls it.  A host
 * program that gets an error from ANY rank switches the mailbox off on ALL ranks (glims_comm_mailbox(h, NULL)). */
int glims_comm_mailbox_selftest(glims_ctx* h);

/* Diagnostic: runs the library's RCCL call sequence (pack -> event -> grouped ncclSend/ncclRecv on the communication
 * stream -> event -> compute stream, then ncclAllReduce on the compute stream) on a fresh ONE-rank communicator pair,
 * sending to itself, and checks the data.  Proves that the RCCL library the process resolved is usable with this
 * build (ABI, datatypes, stream/event ordering); it cannot prove multi-rank semantics. */
int glims_comm_selftest(glims_ctx* h);

/* Halo plan: for peer p (rank peer_rank[p]) this rank sends the values of its owned nodes
 * send_idx[send_ptr[p] .. send_ptr[p+1]) and receives recv_count[p] values into consecutive ghost slots;
 * ghost slots are laid out peer after peer in the order of peer_rank[], starting at node n_own, and the
 * k-th ghost of a peer is the k-th entry of that peer's send list for this rank.  Must precede glims_setup. */
int glims_set_halo(glims_ctx* h, int n_peers, const int32_t* peer_rank,
                   const int64_t* send_ptr, const int32_t* send_idx, const int64_t* recv_count);

/* Partitioned runs with mechanics: the bounding box lo[dim], hi[dim] of the WHOLE mesh (the same numbers on every rank).
 * The elasticity multigrid then lays one global frame of auxiliary grids over the partitioned mesh and replicates its
 * coarse levels (their operators are summed over the ranks once, the restricted residual once per cycle), so that the
 * low-frequency part of the error is corrected globally -- the counterpart of what a parallel AMG does inside PETSc
 * under mpirun (simulation_tumor_growth_brain_quad.py:116-119, README.md:142-183).  Without this call each rank
 * preconditions its own rows only (still correct, about three times the iterations at 2 ranks).  NULL clears. */
int glims_set_mg_frame(glims_ctx* h, const double* lo, const double* hi);

/* Optional host-provided transport instead of RCCL (e.g. the reference's own MPI communicator, which DOLFIN hands
 * around as mesh.mpi_comm(), helper_classes.py:1249-1267; also used by the 2-rank tests on a single GPU, where RCCL
 * refuses two ranks per device).  Both callbacks are invoked from the calling thread with device pointers and the
 * handle's HIP stream; they must return only when the exchange is complete and visible to later work on that stream
 * (e.g. hipStreamSynchronize, staged copies, hipMemcpy back).  Return 0 on success.
 *   halo:      sendbuf[(send_ptr[p] .. send_ptr[p+1]) * bs] goes to peer p; the values received from peer p must be
 *              stored at ghosts[(recv_ptr[p] .. recv_ptr[p+1]) * bs]  (ghosts = first ghost slot of the vector).
 *   allreduce: in-place sum of n doubles over all ranks.
 * Replaces glims_comm_init (sets rank / world); glims_set_halo is still required. */
typedef int (*glims_halo_fn)(void* user, const double* sendbuf_dev, const int64_t* send_ptr, double* ghosts_dev,
                             const int64_t* recv_ptr, int n_peers, const int32_t* peer_rank, int bs, void* hip_stream);
typedef int (*glims_allreduce_fn)(void* user, double* values_dev, int n, void* hip_stream);
int glims_set_transport(glims_ctx* h, int rank, int world, glims_halo_fn halo, glims_allreduce_fn allreduce,
                        void* user);

#ifdef __cplusplus
}
#endif
#endif /* GLIMS_HIP_H */
