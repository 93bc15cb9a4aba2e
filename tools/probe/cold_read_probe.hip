// Probe (round 4): what does the FIRST batch of scattered loads of a launch cost, depending on who touched the data last?
// The narrow phase's staging (rings of two floes + two record lines per 8-lane item) is imitated on 10 000 "floes":
//   W  writes every ring and record (as GEO / the update of the step before do),
//   R  reads them item by item (2560 workgroups of 64 threads, one item per 8 lanes, partner floe random or adjacent),
//   T  streams through all of it once (a "touch" by some other launch in between).
// Printed: event time of R and the mean / max cycles its wavefronts wait for the batch, for the orders W R | W T R | R R.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cold_read_probe tools/probe/cold_read_probe.hip && /tmp/cold_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int RING_Q = 16;   // double2 per ring slot (256 bytes)
__global__ void kW(double2* rings, double2* rec, int n, double v) {
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < (size_t)n * RING_Q; q += (size_t)gridDim.x * blockDim.x) rings[q] = make_double2(v + q, v - q);
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < (size_t)n * 8; q += (size_t)gridDim.x * blockDim.x) rec[q] = make_double2(v, q);
}
__global__ void kT(const double2* rings, const double2* rec, int n, double* sink) {
  double s = 0;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < (size_t)n * RING_Q; q += (size_t)gridDim.x * blockDim.x) s += rings[q].x;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < (size_t)n * 8; q += (size_t)gridDim.x * blockDim.x) s += rec[q].x;
  if (s == 12345.678) sink[0] = s;
}
__global__ void __launch_bounds__(64) kR(const double2* rings, const double2* rec, const int* pi, const int* pj, int nitems, double* sink, long long* lat, int extra_lds) {
  __shared__ double lds[2048];
  const int gl = threadIdx.x & 7, gi = threadIdx.x >> 3;
  const int t = blockIdx.x * 8 + gi;
  double s = 0;
  long long t0 = clock64();
  if (t < nitems) {
    const int i = pi[t], j = pj[t];
    const long long t1 = clock64();
    const double2 a0 = rings[(size_t)i * RING_Q + gl], a1 = rings[(size_t)i * RING_Q + 8 + gl];
    const double2 b0 = rings[(size_t)j * RING_Q + gl], b1 = rings[(size_t)j * RING_Q + 8 + gl];
    const double2 r0 = rec[(size_t)(gl < 4 ? i : j) * 8 + (gl & 3)], r1 = rec[(size_t)(gl < 4 ? j : i) * 8 + 4 + (gl & 3)];
    s = a0.x + a1.y + b0.x + b1.y + r0.x + r1.y;
    lds[threadIdx.x] = s;
    __syncthreads();
    const long long t2 = clock64();
    if (threadIdx.x == 0) { lat[blockIdx.x * 2] = t1 - t0; lat[blockIdx.x * 2 + 1] = t2 - t1; }
  }
  if (s == 12345.678) sink[0] = s + lds[(threadIdx.x + 1) & 63];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 10000, nitems = argc > 2 ? atoi(argv[2]) : 17000;
  double2 *rings, *rec; double* sink; int *pi, *pj; long long* lat;
  CK(hipMalloc(&rings, (size_t)n * RING_Q * 16)); CK(hipMalloc(&rec, (size_t)n * 128)); CK(hipMalloc(&sink, 64));
  const int nb = (nitems + 7) / 8;
  CK(hipMalloc(&pi, nitems * 4)); CK(hipMalloc(&pj, nitems * 4)); CK(hipMalloc(&lat, nb * 16));
  std::vector<int> hi(nitems), hj(nitems);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<long long> hl(nb * 2);
  auto runR = [&](const char* what) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kR, dim3(nb), dim3(64), 0, 0, rings, rec, pi, pj, nitems, sink, lat, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(hl.data(), lat, nb * 16, hipMemcpyDeviceToHost));
    double m0 = 0, m1 = 0; long long x1 = 0; std::vector<long long> v;
    for (int b = 0; b < nb; b++) { m0 += hl[2 * b]; m1 += hl[2 * b + 1]; x1 = std::max(x1, hl[2 * b + 1]); v.push_back(hl[2 * b + 1]); }
    std::sort(v.begin(), v.end());
    printf("  %-46s R: %6.1f us   index loads %5.0f cyc   batch mean %6.0f  median %6lld  90%% %6lld  max %6lld cycles\n", what, ms * 1e3, m0 / nb, m1 / nb, v[nb / 2], v[nb * 9 / 10], x1);
  };
  for (int order = 0; order < 2; order++) {
    srand(1);
    for (int t = 0; t < nitems; t++) {
      const int i = (int)((long long)t * n / nitems);
      hi[t] = i; hj[t] = order == 0 ? (int)(((long long)rand() * 32768 + rand()) % n) : std::min(n - 1, i + 1 + t % 3);
    }
    CK(hipMemcpy(pi, hi.data(), nitems * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(pj, hj.data(), nitems * 4, hipMemcpyHostToDevice));
    printf("%d floes, %d items, partner floe %s\n", n, nitems, order == 0 ? "RANDOM" : "ADJACENT (i+1..i+3)");
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(kW, dim3(1024), dim3(256), 0, 0, rings, rec, n, 1.0 + rep);
      runR("after W (data written by the launch before)");
      runR("after R (data read by the launch before)");
      hipLaunchKernelGGL(kW, dim3(1024), dim3(256), 0, 0, rings, rec, n, 2.0 + rep);
      hipLaunchKernelGGL(kT, dim3(1024), dim3(256), 0, 0, rings, rec, n, sink);
      runR("after W, T (written, then streamed through once)");
    }
  }
  return 0;
}
