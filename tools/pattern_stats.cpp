// Host-only study tool: how compressible are the SELL-64 column indices of a mesh?
// usage: pattern_stats dim n_nodes n_cells pts.f64 cells.i32     (raw little-endian arrays)
// build: hipcc -O3 -std=c++17 -fopenmp -I include -I glimslib_amd/csrc tools/pattern_stats.cpp \
//        glimslib_amd/csrc/build/setup_host.cpp.o -o /tmp/pattern_stats
#include "glims_internal.h"
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

template <class T> static std::vector<T> slurp(const char* f, size_t n) {
  std::vector<T> v(n);
  FILE* fp = fopen(f, "rb");
  if (!fp || fread(v.data(), sizeof(T), n, fp) != n) { perror(f); exit(1); }
  fclose(fp);
  return v;
}

int main(int argc, char** argv) {
  const int dim = atoi(argv[1]);
  const int64_t nn = atoll(argv[2]), nc = atoll(argv[3]);
  auto pts = slurp<double>(argv[4], nn * dim);
  auto cells = slurp<int32_t>(argv[5], nc * (dim + 1));
  HostPattern hp;
  build_host_pattern(hp, dim, nn, nn, nc, pts.data(), cells.data());
  int64_t ent = 0, fit_minmax16 = 0, fit_row16 = 0, fit_minmax8 = 0;
  int64_t fit_multi[5] = {0, 0, 0, 0, 0};
  for (int s = 0; s < hp.n_slices; ++s) {
    const int64_t b = hp.slice_ptr[s], e = hp.slice_ptr[s + 1];
    int32_t lo = INT32_MAX, hi = INT32_MIN;
    std::vector<int32_t> win;
    for (int64_t i = b; i < e; ++i) {
      lo = std::min(lo, hp.cols[i]);
      hi = std::max(hi, hp.cols[i]);
      win.push_back(hp.cols[i] >> 14);   // 16 K-aligned windows
    }
    std::sort(win.begin(), win.end());
    win.erase(std::unique(win.begin(), win.end()), win.end());
    ent += e - b;
    if ((int64_t)hi - lo < 65536) fit_minmax16 += e - b;
    if ((int64_t)hi - lo < 256) fit_minmax8 += e - b;
    const int64_t r0 = (int64_t)s * 64;
    if (lo - r0 >= -32768 && hi - r0 <= 32767) fit_row16 += e - b;
    for (int m = 1; m <= 4; ++m)
      if ((int)win.size() <= m) fit_multi[m] += e - b;
  }
  // greedy covering of each slice's distinct columns with W windows of 2^B columns
  const int combos[][2] = {{2, 14}, {3, 13}, {4, 12}, {5, 11}, {6, 10}, {6, 9}, {6, 8}};
  for (auto& cb : combos) {
    const int W = 1 << cb[0];
    const int64_t R = int64_t(1) << cb[1];
    int64_t fit = 0, nd_sum = 0, nw_sum = 0;
    int nd_max = 0;
#pragma omp parallel for reduction(+ : fit, nd_sum, nw_sum) reduction(max : nd_max)
    for (int s = 0; s < hp.n_slices; ++s) {
      const int64_t b = hp.slice_ptr[s], e = hp.slice_ptr[s + 1];
      std::vector<int32_t> d(hp.cols.begin() + b, hp.cols.begin() + e);
      std::sort(d.begin(), d.end());
      d.erase(std::unique(d.begin(), d.end()), d.end());
      int nw = 0;
      size_t i = 0;
      while (i < d.size()) {
        const int64_t start = d[i];
        ++nw;
        while (i < d.size() && d[i] - start < R) ++i;
      }
      if (nw <= W) fit += e - b;
      nd_sum += (int64_t)d.size();
      nw_sum += nw;
      nd_max = std::max(nd_max, (int)d.size());
    }
    printf("%2d windows x %6lld cols: %.3f %% of entries fit; mean windows needed %.1f; distinct cols/slice mean %.0f max %d\n", W,
           (long long)R, 100.0 * fit / ent, (double)nw_sum / hp.n_slices, (double)nd_sum / hp.n_slices, nd_max);
  }
  printf("slices %d entries %lld (nnz %lld)\n", hp.n_slices, (long long)ent, (long long)hp.nnz);
  printf("u16 offset from slice min : %.2f %% of entries\n", 100.0 * fit_minmax16 / ent);
  printf("i16 offset from first row : %.2f %%\n", 100.0 * fit_row16 / ent);
  printf("u8  offset from slice min : %.2f %%\n", 100.0 * fit_minmax8 / ent);
  for (int m = 1; m <= 4; ++m) printf("<= %d 16K-aligned windows   : %.2f %%\n", m, 100.0 * fit_multi[m] / ent);
  return 0;
}
