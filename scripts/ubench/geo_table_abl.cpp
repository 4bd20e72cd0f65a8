// Timing harness for geo_embed_table_kernel ablations: hipcc -DGT_ABL=<n> ... (probe only; random tables, results not checked).
#include "../../unopose_amd/csrc/embed.hip"
#include <cstdarg>
#include <vector>
namespace unopose { void set_error(const char *fmt, ...) { va_list a; va_start(a, fmt); vprintf(fmt, a); va_end(a); puts(""); } }
#include <cstdlib>
int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16, n = 197, rows_d = 260, rows_a = 52;
  std::vector<float> pts((size_t)B * n * 3), td((size_t)rows_d * 256), ta((size_t)rows_a * 256), bias(256), wd(256 * 256), div(128);
  srand(1);
  for (auto &v : pts) v = (rand() / (float)RAND_MAX - 0.5f) * 1.2f;
  for (auto &v : td) v = rand() / (float)RAND_MAX;
  for (auto &v : ta) v = rand() / (float)RAND_MAX;
  for (int i = 0; i < 128; ++i) div[i] = powf(10000.f, -i / 128.f);
  float *dp, *dtd, *dta, *db, *dw, *dd; int32_t *knn; void *out;
  hipMalloc(&dp, pts.size() * 4); hipMalloc(&dtd, td.size() * 4); hipMalloc(&dta, ta.size() * 4); hipMalloc(&db, 1024);
  hipMalloc(&dw, wd.size() * 4); hipMalloc(&dd, 512); hipMalloc(&knn, (size_t)B * n * 12); hipMalloc(&out, (size_t)B * n * n * 512);
  hipMemcpy(dp, pts.data(), pts.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dtd, td.data(), td.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dta, ta.data(), ta.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, bias.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(dw, wd.data(), wd.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dd, div.data(), 512, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; ++it)
      if (unopose_geo_embedding_table(dp, B, n, dtd, rows_d, dta, rows_a, db, dw, dd, 4, 4, 0.2f, 3.8197186f, 0, 1, knn, out, 0)) return 1;
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("GT_ABL=%d B=%d: %.1f us per call (knn + table kernel)\n",
#ifdef GT_ABL
           GT_ABL,
#else
           0,
#endif
           B, ms * 50.f);
  }
  return 0;
}
