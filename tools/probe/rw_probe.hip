// micro-probe: n threads read KI columns, spin on fp64 arithmetic for `work` rounds, write KO columns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KI, int KO>
__global__ void __launch_bounds__(256) k_rw(double** col, int n, int work, double v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x[KI];
#pragma unroll
  for (int k = 0; k < KI; k++) x[k] = col[k][i];
  for (int r = 0; r < work; r++) {
#pragma unroll
    for (int k = 0; k < KI; k++) x[k] = x[k] * v + x[(k + 1) % KI] / (x[(k + 2) % KI] + 3.0);
  }
#pragma unroll
  for (int k = 0; k < KO; k++) col[k][i] = x[k];
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
  constexpr int KM = 40;
  std::vector<double*> h(KM);
  for (int k = 0; k < KM; k++) { hipMalloc(&h[k], (size_t)2 * n * 8); hipMemset(h[k], 0, (size_t)2 * n * 8); }
  double** d; hipMalloc(&d, KM * sizeof(double*)); hipMemcpy(d, h.data(), KM * sizeof(double*), hipMemcpyHostToDevice);
  int nb = (2 * n + 255) / 256;      // as the engine launches: grid over the capacity, half of it idle
  for (int work : {0, 1, 4, 16})
    printf("n=%d work=%2d  KO=27: %7.1f us   KO=1: %7.1f us\n", n, work,
           timeit([&] { k_rw<33, 27><<<nb, 256>>>(d, n, work, 0.999); }), timeit([&] { k_rw<33, 1><<<nb, 256>>>(d, n, work, 0.999); }));
  return 0;
}
