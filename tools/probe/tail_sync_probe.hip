// Probe: what does a producer -> consumer hand-over INSIDE one launch cost on this chip, against a launch boundary?
//   hipcc --offload-arch=gfx950 -O3 -o tail_sync_probe tools/probe/tail_sync_probe.hip && ./tail_sync_probe
// Producers (P workgroups of 64 threads, ~`work` us of busy time each, like the narrow phase's lane groups) write one 40-byte row per
// thread; consumers (C workgroups, handed out behind the producers) gather rows written by OTHER workgroups -- other XCDs -- and sum them.
//   mode 0: two launches (the boundary does the cache write-back / invalidate)
//   mode 1: one launch, consumers wait on a done-count; rows written with agent-scope (write-through) stores, read with agent-scope
//           loads, the count bumped by a relaxed agent-scope atomic after s_waitcnt vmcnt(0): NO cache-wide fence
//   mode 2: one launch, ordinary stores / loads with __threadfence() (release) in the producers and an acquire fence in the consumers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void busy(long long cycles) { const long long t0 = clock64(); while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(2); }

template <int MODE>
__device__ __forceinline__ void produce(double* rows, unsigned* done, int bid, long long cycles, int epoch) {
  // uneven work, as the narrow phase's rounds: most workgroups short, a few long
  const long long c = (bid % 16 == 0) ? cycles : cycles / 3;
  busy(c);
  const size_t t = (size_t)bid * 64 + threadIdx.x;
  for (int k = 0; k < 5; k++) {
    const double v = (double)(t * 5 + k + epoch);
    if (MODE == 1) __hip_atomic_store(rows + t * 5 + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else rows[t * 5 + k] = v;
  }
  if (MODE == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (MODE == 2) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <int MODE>
__device__ __forceinline__ void consume(const double* rows, unsigned* done, double* out, int cid, int P, int epoch, unsigned target) {
  if (MODE != 0) {
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(done, MODE == 2 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  // six rows of other workgroups (a floe's items were run by lane groups all over the chip)
  const size_t n = (size_t)P * 64;
  const size_t me = (size_t)cid * 64 + threadIdx.x;
  double s = 0.0;
  for (int q = 0; q < 6; q++) {
    const size_t t = (me * 2654435761ull + (size_t)q * 40503ull) % n;
    for (int k = 0; k < 5; k++) {
      double v;
      if (MODE == 1) v = __hip_atomic_load(rows + t * 5 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else v = rows[t * 5 + k];
      s += v - (double)(t * 5 + k + epoch);          // 0 when the row of THIS epoch was seen
    }
  }
  out[me] = s;
}
template <int MODE>
__global__ void __launch_bounds__(64) k_prod(double* rows, unsigned* done, long long cycles, int epoch) { produce<MODE>(rows, done, blockIdx.x, cycles, epoch); }
template <int MODE>
__global__ void __launch_bounds__(64) k_cons(const double* rows, unsigned* done, double* out, int P, int epoch) { consume<MODE>(rows, done, out, blockIdx.x, P, epoch, 0); }
template <int MODE>
__global__ void __launch_bounds__(64) k_fused(double* rows, unsigned* done, double* out, int P, long long cycles, int epoch, unsigned target) {
  if ((int)blockIdx.x < P) produce<MODE>(rows, done, blockIdx.x, cycles, epoch);
  else consume<MODE>(rows, done, out, (int)blockIdx.x - P, P, epoch, target);
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 2560, C = argc > 2 ? atoi(argv[2]) : 1300, iters = 200;
  const double work_us = argc > 3 ? atof(argv[3]) : 30.0;
  const long long cycles = (long long)(work_us * 2100.0);      // clock64: shader clock, ~2.1 GHz
  double *rows, *out; unsigned* done;
  CHK(hipMalloc(&rows, (size_t)P * 64 * 5 * sizeof(double))); CHK(hipMalloc(&out, (size_t)C * 64 * sizeof(double))); CHK(hipMalloc(&done, 256));
  CHK(hipMemset(done, 0, 256));
  hipStream_t st; CHK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  std::vector<double> h((size_t)C * 64);
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9f, sum = 0.f; double bad = 0.0;
    for (int rep = 0; rep < 5; rep++) {
      CHK(hipMemsetAsync(done, 0, 256, st));
      CHK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; it++) {
        const int epoch = rep * iters + it + 1;
        const unsigned target = (unsigned)(it + 1) * (unsigned)P;
        if (mode == 0) {
          hipLaunchKernelGGL(k_prod<0>, dim3(P), dim3(64), 0, st, rows, done, cycles, epoch);
          hipLaunchKernelGGL(k_cons<0>, dim3(C), dim3(64), 0, st, rows, done, out, P, epoch);
        } else if (mode == 1) hipLaunchKernelGGL(k_fused<1>, dim3(P + C), dim3(64), 0, st, rows, done, out, P, cycles, epoch, target);
        else hipLaunchKernelGGL(k_fused<2>, dim3(P + C), dim3(64), 0, st, rows, done, out, P, cycles, epoch, target);
      }
      CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best; sum += ms;
      CHK(hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (double v : h) bad += v != 0.0;
    }
    printf("mode %d (%s): %.2f us per iteration (best of 5: %.2f), stale or wrong sums: %.0f\n", mode,
           mode == 0 ? "two launches" : mode == 1 ? "one launch, write-through rows + relaxed count" : "one launch, fences", sum / 5 / iters * 1000, best / iters * 1000, bad);
  }
  return 0;
}
