/*
 * CPU ORACLE (C / OpenMP restatement) -- TEST INFRASTRUCTURE ONLY, like oracle/glims_oracle.py.
 * Only tests/ and bench.py's cpu_baseline leg load it; the product (glimslib_amd) never does.
 * PARITY UNPINNED against FEniCS (see oracle/glims_oracle.py header for why and for what pins it instead).
 *
 * Restates the reaction-diffusion block of the reference's per-timestep solve
 *   F_rd  (glimslib/simulation/simulation_tumor_growth.py:115-120), logistic term math_reaction_diffusion.py:2-3,
 *   Jacobian = derivative(F) (:124), Newton to convergence (:126-130), backward Euler loop (simulation_base.py:277-312)
 * for P1 simplices with the consistent mass matrix and the cubic term integrated exactly, as a third, independent
 * implementation: CSR storage, element integrals from the generic monomial formula
 *   int_T prod lambda_a^alpha_a = |T| d! prod(alpha_a!) / (d + sum alpha)!,
 * row-parallel assembly (OpenMP), Newton + Jacobi-preconditioned CG (textbook two-reduction PCG).
 * It serves as (a) an independent check of the numpy oracle and of the HIP path at sizes numpy cannot reach and
 * (b) the multi-core "port" CPU baseline of bench.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int dim, nv;
  int64_t n, m;
  double dt;
  double *vol, *rho, *D, *grad; /* per cell: |T|, rho, D, grad(lambda) [nv][dim] */
  int32_t* cells;
  int64_t *adj_ptr; int32_t* adj;          /* node -> cells */
  int64_t* rowptr; int32_t* col;           /* CSR pattern (sorted columns) */
  double *M, *S, *A;                       /* CSR values */
  double Mref[4][4], Tref[4][4][4];
  double *r, *z, *p, *q, *dinv, *b, *dx;   /* work vectors */
  int64_t cg_its, newton_its, sweeps;
} oc_t;

static double factorial(int k) { double f = 1; for (int i = 2; i <= k; ++i) f *= i; return f; }

static double monomial(const int* alpha, int nv, int d) {
  double num = factorial(d); int s = 0;
  for (int a = 0; a < nv; ++a) { num *= factorial(alpha[a]); s += alpha[a]; }
  return num / factorial(d + s);
}

static int cmp_i32(const void* a, const void* b) { int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }

static int find_col(const oc_t* o, int64_t row, int32_t c) {
  int64_t lo = o->rowptr[row], hi = o->rowptr[row + 1] - 1;
  while (lo <= hi) { int64_t mid = (lo + hi) >> 1; int32_t v = o->col[mid]; if (v == c) return (int)(mid - o->rowptr[row]); if (v < c) lo = mid + 1; else hi = mid - 1; }
  return -1;
}

static void geometry(oc_t* o, const double* xyz) {
  const int d = o->dim, nv = o->nv;
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < o->m; ++e) {
    double X[4][3]; for (int a = 0; a < nv; ++a) for (int k = 0; k < d; ++k) X[a][k] = xyz[(int64_t)o->cells[e * nv + a] * d + k];
    double* g = o->grad + e * nv * d;
    if (d == 2) {
      double a = X[1][0] - X[0][0], b = X[1][1] - X[0][1], c = X[2][0] - X[0][0], dd = X[2][1] - X[0][1];
      double det = a * dd - b * c;
      o->vol[e] = fabs(det) / 2.0;
      g[2] = dd / det; g[3] = -c / det; g[4] = -b / det; g[5] = a / det;
      g[0] = -(g[2] + g[4]); g[1] = -(g[3] + g[5]);
    } else {
      double r[3][3]; for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) r[i][k] = X[i + 1][k] - X[0][k];
      double c0[3] = {r[1][1]*r[2][2]-r[1][2]*r[2][1], r[1][2]*r[2][0]-r[1][0]*r[2][2], r[1][0]*r[2][1]-r[1][1]*r[2][0]};
      double c1[3] = {r[2][1]*r[0][2]-r[2][2]*r[0][1], r[2][2]*r[0][0]-r[2][0]*r[0][2], r[2][0]*r[0][1]-r[2][1]*r[0][0]};
      double c2[3] = {r[0][1]*r[1][2]-r[0][2]*r[1][1], r[0][2]*r[1][0]-r[0][0]*r[1][2], r[0][0]*r[1][1]-r[0][1]*r[1][0]};
      double det = r[0][0]*c0[0] + r[0][1]*c0[1] + r[0][2]*c0[2];
      o->vol[e] = fabs(det) / 6.0;
      for (int k = 0; k < 3; ++k) { g[3+k] = c0[k]/det; g[6+k] = c1[k]/det; g[9+k] = c2[k]/det; g[k] = -(g[3+k]+g[6+k]+g[9+k]); }
    }
  }
}

void* oc_create(int dim, int64_t n, int64_t m, const double* xyz, const int32_t* cells, const double* D,
                const double* rho, double dt) {
  oc_t* o = (oc_t*)calloc(1, sizeof(oc_t));
  o->dim = dim; o->nv = dim + 1; o->n = n; o->m = m; o->dt = dt;
  const int nv = o->nv;
  o->cells = (int32_t*)malloc(sizeof(int32_t) * m * nv); memcpy(o->cells, cells, sizeof(int32_t) * m * nv);
  o->vol = (double*)malloc(sizeof(double) * m); o->grad = (double*)malloc(sizeof(double) * m * nv * dim);
  o->rho = (double*)malloc(sizeof(double) * m); o->D = (double*)malloc(sizeof(double) * m);
  memcpy(o->rho, rho, sizeof(double) * m); memcpy(o->D, D, sizeof(double) * m);
  geometry(o, xyz);
  for (int i = 0; i < nv; ++i) for (int j = 0; j < nv; ++j) {
    int al[4] = {0,0,0,0}; al[i]++; al[j]++; o->Mref[i][j] = monomial(al, nv, dim);
    for (int k = 0; k < nv; ++k) { int be[4] = {0,0,0,0}; be[i]++; be[j]++; be[k]++; o->Tref[i][j][k] = monomial(be, nv, dim); }
  }
  /* node -> cell adjacency */
  o->adj_ptr = (int64_t*)calloc(n + 1, sizeof(int64_t));
  for (int64_t e = 0; e < m; ++e) for (int a = 0; a < nv; ++a) o->adj_ptr[cells[e * nv + a] + 1]++;
  for (int64_t i = 0; i < n; ++i) o->adj_ptr[i + 1] += o->adj_ptr[i];
  o->adj = (int32_t*)malloc(sizeof(int32_t) * o->adj_ptr[n]);
  int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * n); memcpy(fill, o->adj_ptr, sizeof(int64_t) * n);
  for (int64_t e = 0; e < m; ++e) for (int a = 0; a < nv; ++a) o->adj[fill[cells[e * nv + a]]++] = (int32_t)e;
  free(fill);
  /* CSR pattern */
  o->rowptr = (int64_t*)calloc(n + 1, sizeof(int64_t));
  int64_t maxadj = 0; for (int64_t i = 0; i < n; ++i) if (o->adj_ptr[i+1]-o->adj_ptr[i] > maxadj) maxadj = o->adj_ptr[i+1]-o->adj_ptr[i];
#pragma omp parallel
  {
    int32_t* buf = (int32_t*)malloc(sizeof(int32_t) * (maxadj * nv + 4));
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      int cnt = 0;
      for (int64_t q = o->adj_ptr[i]; q < o->adj_ptr[i+1]; ++q) for (int a = 0; a < nv; ++a) buf[cnt++] = cells[(int64_t)o->adj[q] * nv + a];
      qsort(buf, cnt, sizeof(int32_t), cmp_i32);
      int u = 0; for (int k = 0; k < cnt; ++k) if (k == 0 || buf[k] != buf[k-1]) u++;
      o->rowptr[i + 1] = u;
    }
    free(buf);
  }
  for (int64_t i = 0; i < n; ++i) o->rowptr[i + 1] += o->rowptr[i];
  const int64_t nnz = o->rowptr[n];
  o->col = (int32_t*)malloc(sizeof(int32_t) * nnz);
#pragma omp parallel
  {
    int32_t* buf = (int32_t*)malloc(sizeof(int32_t) * (maxadj * nv + 4));
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      int cnt = 0;
      for (int64_t q = o->adj_ptr[i]; q < o->adj_ptr[i+1]; ++q) for (int a = 0; a < nv; ++a) buf[cnt++] = cells[(int64_t)o->adj[q] * nv + a];
      qsort(buf, cnt, sizeof(int32_t), cmp_i32);
      int64_t w = o->rowptr[i];
      for (int k = 0; k < cnt; ++k) if (k == 0 || buf[k] != buf[k-1]) o->col[w++] = buf[k];
    }
    free(buf);
  }
  o->M = (double*)calloc(nnz, sizeof(double)); o->S = (double*)calloc(nnz, sizeof(double)); o->A = (double*)calloc(nnz, sizeof(double));
  /* static matrices, row-parallel: M and S = (1 - dt rho) M + dt D K */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    for (int64_t q = o->adj_ptr[i]; q < o->adj_ptr[i+1]; ++q) {
      const int64_t e = o->adj[q]; const int32_t* cv = o->cells + e * nv; const double* g = o->grad + e * nv * dim;
      int li = 0; for (int a = 0; a < nv; ++a) if (cv[a] == i) li = a;
      for (int a = 0; a < nv; ++a) {
        double gg = 0; for (int k = 0; k < dim; ++k) gg += g[li * dim + k] * g[a * dim + k];
        const int s = find_col(o, i, cv[a]); const int64_t idx = o->rowptr[i] + s;
        const double mij = o->vol[e] * o->Mref[li][a];
        o->M[idx] += mij;
        o->S[idx] += (1.0 - dt * o->rho[e]) * mij + dt * o->D[e] * o->vol[e] * gg;
      }
    }
  double** vecs[] = {&o->r, &o->z, &o->p, &o->q, &o->dinv, &o->b, &o->dx};
  for (int k = 0; k < 7; ++k) *vecs[k] = (double*)calloc(n, sizeof(double));
  return o;
}

void oc_destroy(void* h) {
  oc_t* o = (oc_t*)h; if (!o) return;
  free(o->cells); free(o->vol); free(o->grad); free(o->rho); free(o->D); free(o->adj_ptr); free(o->adj); free(o->rowptr); free(o->col);
  free(o->M); free(o->S); free(o->A); free(o->r); free(o->z); free(o->p); free(o->q); free(o->dinv); free(o->b); free(o->dx); free(o);
}

int64_t oc_nnz(void* h) { return ((oc_t*)h)->rowptr[((oc_t*)h)->n]; }

static void spmv(const oc_t* o, const double* v, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < o->n; ++i) { double a = 0; for (int64_t k = o->rowptr[i]; k < o->rowptr[i+1]; ++k) a += v[k] * x[o->col[k]]; y[i] = a; }
}
void oc_apply(void* h, int which, const double* x, double* y) { oc_t* o = (oc_t*)h; spmv(o, which == 0 ? o->A : which == 1 ? o->S : o->M, x, y); }

/* A = S + 2 dt N(c), N_ij = sum_T rho_T |T| sum_k Tref_ijk c_k ;  returns -R = b - (S c + dt N(c) c) in r, ||R||_2 */
static double sweep(oc_t* o, const double* c) {
  const int nv = o->nv; double nrm = 0;
  o->sweeps++;
#pragma omp parallel for schedule(static) reduction(+ : nrm)
  for (int64_t i = 0; i < o->n; ++i) {
    for (int64_t k = o->rowptr[i]; k < o->rowptr[i+1]; ++k) o->A[k] = 0.0;
    for (int64_t q = o->adj_ptr[i]; q < o->adj_ptr[i+1]; ++q) {
      const int64_t e = o->adj[q]; const int32_t* cv = o->cells + e * nv;
      int li = 0; for (int a = 0; a < nv; ++a) if (cv[a] == i) li = a;
      const double w = o->rho[e] * o->vol[e];
      if (w == 0.0) continue;
      for (int a = 0; a < nv; ++a) {
        double t = 0; for (int k = 0; k < nv; ++k) t += o->Tref[li][a][k] * c[cv[k]];
        o->A[o->rowptr[i] + find_col(o, i, cv[a])] += w * t;
      }
    }
    double sc = 0, nc = 0, dg = 1;
    for (int64_t k = o->rowptr[i]; k < o->rowptr[i+1]; ++k) {
      const double Nk = o->A[k], cj = c[o->col[k]];
      sc += o->S[k] * cj; nc += Nk * cj;
      o->A[k] = o->S[k] + 2.0 * o->dt * Nk;
      if (o->col[k] == i) dg = o->A[k];
    }
    const double res = o->b[i] - (sc + o->dt * nc);
    o->r[i] = res; o->dinv[i] = 1.0 / dg; nrm += res * res;
  }
  return sqrt(nrm);
}

static double dot(const oc_t* o, const double* a, const double* b) {
  double s = 0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (int64_t i = 0; i < o->n; ++i) s += a[i] * b[i];
  return s;
}

/* Jacobi-PCG on A dx = r (r is overwritten), dx starts at 0; stops at ||r|| <= tol */
static int pcg(oc_t* o, double tol, int maxit) {
  const int64_t n = o->n; double *r = o->r, *z = o->z, *p = o->p, *q = o->q, *x = o->dx;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) { x[i] = 0; z[i] = o->dinv[i] * r[i]; p[i] = z[i]; }
  double rz = dot(o, r, z); int it = 0;
  while (it < maxit && sqrt(dot(o, r, r)) > tol) {
    spmv(o, o->A, p, q);
    const double alpha = rz / dot(o, p, q);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * q[i]; z[i] = o->dinv[i] * r[i]; }
    const double rz2 = dot(o, r, z), beta = rz2 / rz; rz = rz2;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    ++it;
  }
  o->cg_its += it;
  return it;
}

/* n_steps backward-Euler steps in place on c; returns 0 on success, 1 if Newton failed */
int oc_step(void* h, double* c, int n_steps, double rtol, double atol, double cg_rtol, const double* load) {
  oc_t* o = (oc_t*)h; const int64_t n = o->n;
  for (int s = 0; s < n_steps; ++s) {
    spmv(o, o->M, c, o->b);
    if (load) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) o->b[i] += load[i];
    }
    double nr = sweep(o, c), r0 = nr, target = fmax(atol, rtol * r0);
    int it = 0;
    while (nr > target) {
      if (it++ >= 50 || !isfinite(nr)) return 1;
      pcg(o, fmax(0.1 * target, cg_rtol * nr), 20000);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) c[i] += o->dx[i];
      o->newton_its++;
      nr = sweep(o, c);
    }
  }
  return 0;
}

void oc_stats(void* h, int64_t* out3) { oc_t* o = (oc_t*)h; out3[0] = o->newton_its; out3[1] = o->cg_its; out3[2] = o->sweeps; }
void oc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int oc_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
