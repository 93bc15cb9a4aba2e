// micro-probe: how fast can n threads write K columns (SoA, 8 B per thread and column)?  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int K>
__global__ void k_soa(double** col, int n, double v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < K; k++) col[k][i] = v + k;
}
template <int K>
__global__ void k_soa_rw(double** col, int n, double v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x[K];
#pragma unroll
  for (int k = 0; k < K; k++) x[k] = col[k][i];
#pragma unroll
  for (int k = 0; k < K; k++) col[k][i] = x[k] * v + k;
}
template <int K>
__global__ void k_aos(double* rec, int n, double v) {      // K doubles per thread, contiguous per thread
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < K; k++) rec[(size_t)i * K + k] = v + k;
}
template <class F>
float timeit(F f, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int r = 0; r < 3; r++) f();
  hipEventRecord(a);
  for (int r = 0; r < reps; r++) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 100000;
  int tpb = argc > 2 ? atoi(argv[2]) : 256;
  constexpr int KM = 32;
  std::vector<double*> h(KM);
  char* arena; hipMalloc(&arena, (size_t)KM * n * 8 + 4096);
  for (int k = 0; k < KM; k++) h[k] = (double*)(arena + (size_t)k * n * 8);
  double** d; hipMalloc(&d, KM * sizeof(double*)); hipMemcpy(d, h.data(), KM * sizeof(double*), hipMemcpyHostToDevice);
  int nb = (n + tpb - 1) / tpb;
  printf("n=%d tpb=%d\n", n, tpb);
  printf("soa write  K=1  %7.1f us\n", timeit([&] { k_soa<1><<<nb, tpb>>>(d, n, 1.0); }));
  printf("soa write  K=8  %7.1f us\n", timeit([&] { k_soa<8><<<nb, tpb>>>(d, n, 1.0); }));
  printf("soa write  K=27 %7.1f us  (%.2f TB/s)\n", timeit([&] { k_soa<27><<<nb, tpb>>>(d, n, 1.0); }), 27.0 * n * 8 / timeit([&] { k_soa<27><<<nb, tpb>>>(d, n, 1.0); }) / 1e6);
  printf("soa r+w    K=27 %7.1f us\n", timeit([&] { k_soa_rw<27><<<nb, tpb>>>(d, n, 1.0); }));
  printf("aos write  K=27 %7.1f us\n", timeit([&] { k_aos<27><<<nb, tpb>>>((double*)arena, n, 1.0); }));
  printf("aos write  K=8  %7.1f us\n", timeit([&] { k_aos<8><<<nb, tpb>>>((double*)arena, n, 1.0); }));
  // the same with separate allocations per column
  for (int k = 0; k < KM; k++) hipMalloc(&h[k], (size_t)n * 8);
  hipMemcpy(d, h.data(), KM * sizeof(double*), hipMemcpyHostToDevice);
  printf("soa write  K=27 %7.1f us  (one hipMalloc per column)\n", timeit([&] { k_soa<27><<<nb, tpb>>>(d, n, 1.0); }));
  printf("soa r+w    K=27 %7.1f us  (one hipMalloc per column)\n", timeit([&] { k_soa_rw<27><<<nb, tpb>>>(d, n, 1.0); }));
  return 0;
}
