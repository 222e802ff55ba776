/* Sanitizer driver for the C oracle (TEST INFRASTRUCTURE ONLY): a small 3-D box mesh, a few implicit steps, the
 * operator hooks -- built by `make -C oracle sanitize` with -fsanitize=address,undefined and run on the CPU.
 * (GPU AddressSanitizer is not available on this pool; the device code is covered by the parity tests instead.) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void* oc_create(int dim, int64_t n, int64_t m, const double* xyz, const int32_t* cells, const double* D,
                const double* rho, double dt);
void oc_destroy(void* h);
int64_t oc_nnz(void* h);
void oc_apply(void* h, int which, const double* x, double* y);
int oc_step(void* h, double* c, int n_steps, double rtol, double atol, double cg_rtol, const double* load);
void oc_stats(void* h, int64_t* out3);

int main(void) {
  const int nx = 5, ny = 4, nz = 3;
  const int64_t n = (int64_t)(nx + 1) * (ny + 1) * (nz + 1), m = (int64_t)nx * ny * nz * 6;
  double* xyz = malloc(sizeof(double) * 3 * n);
  int32_t* cells = malloc(sizeof(int32_t) * 4 * m);
  double *D = malloc(sizeof(double) * m), *rho = malloc(sizeof(double) * m), *c = malloc(sizeof(double) * n),
         *y = malloc(sizeof(double) * n);
  for (int k = 0; k <= nz; ++k)
    for (int j = 0; j <= ny; ++j)
      for (int i = 0; i <= nx; ++i) {
        const int64_t v = ((int64_t)k * (ny + 1) + j) * (nx + 1) + i;
        xyz[3 * v] = i * 0.7, xyz[3 * v + 1] = j * 0.9, xyz[3 * v + 2] = k * 1.1;
        c[v] = exp(-0.3 * ((i - 2.0) * (i - 2.0) + (j - 2.0) * (j - 2.0) + (k - 1.0) * (k - 1.0)));
      }
  /* six tetrahedra per cube around the main diagonal v0-v7 */
  static const int tet[6][4] = {{0, 1, 3, 7}, {0, 1, 7, 5}, {0, 5, 7, 4}, {0, 3, 2, 7}, {0, 6, 4, 7}, {0, 2, 6, 7}};
  int64_t e = 0;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        int64_t v[8];
        for (int q = 0; q < 8; ++q)
          v[q] = ((int64_t)(k + ((q >> 2) & 1)) * (ny + 1) + (j + ((q >> 1) & 1))) * (nx + 1) + (i + (q & 1));
        for (int t = 0; t < 6; ++t, ++e) {
          for (int a = 0; a < 4; ++a) cells[4 * e + a] = (int32_t)v[tet[t][a]];
          D[e] = i < nx / 2 ? 0.1 : 0.02;
          rho[e] = 0.05;
        }
      }
  void* h = oc_create(3, n, m, xyz, cells, D, rho, 1.0);
  if (!h) return 2;
  const int rc = oc_step(h, c, 4, 1e-10, 1e-13, 1e-3, NULL);
  oc_apply(h, 1, c, y);
  int64_t st[3];
  oc_stats(h, st);
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += y[i];
  printf("sanitize_main: status %d, nnz %lld, newton %lld, pcg %lld, checksum %.12e\n", rc, (long long)oc_nnz(h),
         (long long)st[0], (long long)st[1], s);
  oc_destroy(h);
  free(xyz); free(cells); free(D); free(rho); free(c); free(y);
  return rc == 0 && isfinite(s) ? 0 : 1;
}
