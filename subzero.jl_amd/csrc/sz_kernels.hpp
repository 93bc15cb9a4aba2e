// sz_kernels.hpp — the HIP kernels of one Subzero timestep on gfx950 (MI355X).
//
// Pipeline (every launch reads its sizes from the device counter block, no host round trip):
//   ghosts    sz_k_ghost_flag_scan (look-back scan), sz_k_ghost_fill   add_ghosts!          collisions.jl:1060-1174
//   broad     sz_k_bounds, sz_k_cell_build, sz_k_neighbors,
//             sz_k_pscan_fill (look-back scan + pair fill)            pair loop + Dict     collisions.jl:745-775
//   domain    sz_k_elem_scan_fill (count, scan, fill)                       wall prefilters      collisions.jl:608-660
//   narrow    sz_k_narrow<G,CAP,...>                                       floe_floe_interaction! / floe_domain_element_interaction!
//   reduce    sz_k_inter_fill                                             mirror, ghost fold, torque, totals  collisions.jl:799-862
//   forcing   sz_k_forcing<TW>  (+ sz_twoway.hpp with two-way coupling)   timestep_coupling!   coupling.jl:1486-1738
//   integrate sz_k_integrate, sz_k_move_strain                            timestep_floe_properties! update_floe.jl:469-551
//   tiles     sz_k_halo_pack / sz_k_halo_unpack                           ghost-floe halo of a tiled (multi-GPU) run
// All of it is HBM/latency-bound integer + fp64 vector work: no MFMA anywhere.
#pragma once
#include <type_traits>
#include "sz_geom.hpp"
#include "sz_state.hpp"

namespace sz {
using namespace szg;

#define SZ_ACTIVE 1
#define SZ_REMOVE 2
#define SZ_FUSE 3

// Resident batches (sz_step) stop after the step in which a floe was tagged remove / fuse: the reference runs
// simplify_floes! after every step (simulation.jl:205-214), and that is host work.  The launches of the later steps
// are already enqueued; each of them asks this first and returns at once.  S.step is the 1-based step of the launch
// inside its batch (kernarg), cnt[C_STOP] the step that asked for the stop (0: none).
__device__ __forceinline__ bool stopped(const State& S) {
  const int s = S.cnt[C_STOP], r = S.cnt[C_RETRYSTOP];
  return (s != 0 && S.step > s) || (r != 0 && S.step > r);
}
// The other reason a batch pauses: the largest narrow-phase variant (sz_k_narrow<64, ..>) exists for items whose clip
// outgrows the small working set (IT_RETRY) -- none in most fields, yet its launch costs ~4 us + a launch boundary in
// every step.  sz_step therefore leaves it out (State::retry_stop) until such an item shows up: the small variant then
// raises cnt[C_RETRYSTOP], the kernels that FOLLOW the narrow phase in that step (reduce, update) and all later steps return
// at once, and the host enqueues the missing variant, the rest of the step and the remaining steps (the variant stays in
// from then on).  Kernels after the narrow phase ask stopped_late().
__device__ __forceinline__ bool stopped_late(const State& S) {
  const int s = S.cnt[C_STOP], r = S.cnt[C_RETRYSTOP];
  return (s != 0 && S.step > s) || (r != 0 && S.step >= r);
}
// The test costs a round trip to the counter block, and a kernel that tests FIRST spends it before it has asked for anything
// else.  The kernels on the step's critical path therefore ask for the counters (stop_load), then for their first batch of
// rows, and test afterwards (stop_test: nothing has been stored yet); loads_issued() keeps the compiler from sinking the
// loads below the test.
struct StopRegs { int s, r; };
__device__ __forceinline__ StopRegs stop_load(const State& S) { StopRegs q; q.s = S.cnt[C_STOP]; q.r = S.cnt[C_RETRYSTOP]; return q; }
__device__ __forceinline__ bool stop_test(const State& S, const StopRegs& q) { return (q.s != 0 && S.step > q.s) || (q.r != 0 && S.step > q.r); }
__device__ __forceinline__ bool stop_test_late(const State& S, const StopRegs& q) { return (q.s != 0 && S.step > q.s) || (q.r != 0 && S.step >= q.r); }
__device__ __forceinline__ void loads_issued() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void request_stop(const State& S) {
  if (S.step > 0 && S.stop_on_tags) S.cnt[C_STOP] = S.step;      // every requester of a step writes the same value
}
// a NEW tag (raised by the narrow phase or the forcings of the step): also ends the enqueued steps of a pipelined batch that runs through
// (State::restart_on_tags)
__device__ __forceinline__ void request_stop_new_tag(const State& S) {
  if (S.step > 0 && (S.stop_on_tags || S.restart_on_tags)) S.cnt[C_STOP] = S.step;
}
// A list of this step has outgrown its capacity (neighbours per floe, pair items, interaction rows per floe): the reference's lists
// grow as needed (collisions.jl:290-296), the engine's are carved per upload.  Besides the sticky error bit the step is paused like
// the step that needs the largest narrow variant (C_RETRYSTOP): the kernels that follow in it and all later steps return at once,
// nothing of the floes' state has changed, and the host carves larger lists and runs the step again (sz_api.hip: grow_lists).
__device__ __forceinline__ void capacity_stop(const State& S) {
  if (S.step > 0) S.cnt[C_RETRYSTOP] = S.step;
}

// ring offset / size of floe i: into vx, vy (CSR offsets voff) or, on body-frame rings (mixed precision), into ring32
__device__ __forceinline__ int ring_off(const State& S, int i) { return S.body_rings ? S.rb_off[i] : S.voff[i]; }
__device__ __forceinline__ int ring_n(const State& S, int i) { return S.body_rings ? S.rb_n[i] : S.voff[i + 1] - S.voff[i]; }
// margins of the fp32 broad-phase prefilter (metres): coordinates up to 2^23 m (8 388 km) round to <= 0.5 m in fp32, a
// difference of two of them is off by <= 1 m, a sum of two radii by far less; the box margin covers both roundings
constexpr float MIX_CIRCLE_MARGIN = 4.0f, MIX_BOX_MARGIN = 2.0f;
__device__ __forceinline__ void rec32_store(const State& S, int i, double cx, double cy, double r, double x0, double x1, double y0, double y1) {
  S.rec32[2 * (size_t)i] = make_float4((float)cx, (float)cy, (float)r, 0.0f);
  S.rec32[2 * (size_t)i + 1] = make_float4((float)x0, (float)x1, (float)y0, (float)y1);
}

// ---- collision records (State::crec): packing of the two integer quads, and the stores of whoever places a floe
constexpr long long CREC_KEYMASK = (1ll << 48) - 1;          // order keys stay below 2^48: (pass << 40) + 4 * (global index) + k
__device__ __forceinline__ double crec_okf(long long okey, int nv, int osign) {
  return __longlong_as_double(okey | ((long long)(nv & 255) << 48) | ((long long)(osign < 0 ? 1 : 0) << 56));
}
__device__ __forceinline__ double crec_vp(int voff, int parent, int ngh) {
  return __longlong_as_double((long long)(unsigned)voff | ((long long)parent << 32) | ((long long)ngh << 60));
}
__device__ __forceinline__ void crec_store_all(const State& S, int row, double cx, double cy, double rmax, long long id, long long okey, int nv, int osign,
                                               int voff, int parent, int ngh, double x0, double x1, double y0, double y1, double u, double v, double xi,
                                               double area, double h, long long ghost_id) {
  double2* r = S.crec + (size_t)row * 8;
  r[0] = make_double2(cx, cy); r[1] = make_double2(rmax, __longlong_as_double(id));
  r[2] = make_double2(crec_okf(okey, nv, osign), crec_vp(voff, parent, ngh));
  r[3] = make_double2(x0, x1); r[4] = make_double2(y0, y1); r[5] = make_double2(u, v); r[6] = make_double2(xi, area);
  r[7] = make_double2(h, __longlong_as_double(ghost_id));
}
// a floe that has only been translated (a parent that swapped with its ghost)
__device__ __forceinline__ void crec_store_place(const State& S, int row, double cx, double cy, double x0, double x1, double y0, double y1) {
  double2* r = S.crec + (size_t)row * 8;
  r[0] = make_double2(cx, cy); r[3] = make_double2(x0, x1); r[4] = make_double2(y0, y1);
}

// ============================================================================ scan (exclusive, int)
constexpr int SCAN_B = 1024;

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  __shared__ int wsum[SCAN_B / 64];
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
  for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    int w = lane < (int)(blockDim.x >> 6) ? wsum[lane] : 0, wi = w;
    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(wi, d); if (lane >= d) wi += t; }
    if (lane < (int)(blockDim.x >> 6)) wsum[lane] = wi - w;
    if (lane == 63) *total = wi;
  }
  __syncthreads();
  int res = inc - v + wsum[wid];
  __syncthreads();
  return res;
}

// n = cnt[ci] + add (ci >= 0) or add alone (ci < 0)
__global__ void __launch_bounds__(SCAN_B) sz_k_scan1(const int* in, int* out, int* blk, const int* cnt, int ci, int add) {
  __shared__ int tot;
  int n = (ci >= 0 ? cnt[ci] : 0) + add;
  int base = blockIdx.x * SCAN_B;
  if (base >= n && !(n == 0 && blockIdx.x == 0)) return;
  int i = base + threadIdx.x;
  int v = i < n ? in[i] : 0;
  int ex = block_exclusive_scan(v, &tot);
  if (i < n) out[i] = ex;
  if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(SCAN_B) sz_k_scan2(int* blk, const int* cnt, int ci, int add) {
  __shared__ int tot;
  int n = (ci >= 0 ? cnt[ci] : 0) + add;
  int nb = (n + SCAN_B - 1) / SCAN_B;
  if (nb < 1) nb = 1;
  int carry = 0;
  for (int base = 0; base < nb; base += SCAN_B) {
    int i = base + threadIdx.x;
    int v = i < nb ? blk[i] : 0;
    int ex = block_exclusive_scan(v, &tot);
    if (i < nb) blk[i] = ex + carry;
    carry += tot;
    __syncthreads();
  }
}
// adds block offsets, writes the grand total to out[n] and (co >= 0) to cnt[co]
__global__ void __launch_bounds__(SCAN_B) sz_k_scan3(const int* in, int* out, const int* blk, int* cnt, int ci, int add, int co) {
  int n = (ci >= 0 ? cnt[ci] : 0) + add;
  int i = blockIdx.x * SCAN_B + threadIdx.x;
  if (n == 0) { if (i == 0) { out[0] = 0; if (co >= 0) cnt[co] = 0; } return; }
  if (i >= n) return;
  int o = out[i] + blk[blockIdx.x];
  out[i] = o;
  if (i == n - 1) { int t = o + in[i]; out[n] = t; if (co >= 0) cnt[co] = t; }
}

// The whole scan in ONE workgroup (4 elements per thread and tile): below a few thousand elements three
// launches cost more than the work.  Same contract as scan1..3.
__global__ void __launch_bounds__(SCAN_B) sz_k_scan_one(const int* in, int* out, int* cnt, int ci, int add, int co) {
  __shared__ int tot;
  const int n = (ci >= 0 ? cnt[ci] : 0) + add;
  int carry = 0;
  for (int base = 0; base < n; base += 4 * SCAN_B) {
    const int i0 = base + 4 * (int)threadIdx.x;
    int v[4], sum = 0;
    for (int k = 0; k < 4; k++) { v[k] = i0 + k < n ? in[i0 + k] : 0; sum += v[k]; }
    int ex = block_exclusive_scan(sum, &tot) + carry;
    for (int k = 0; k < 4; k++) { if (i0 + k < n) out[i0 + k] = ex; ex += v[k]; }
    carry += tot;
  }
  if (threadIdx.x == 0) { out[n] = carry; if (co >= 0) cnt[co] = carry; }
}

// ============================================================================ small utilities
// several buffers filled in ONE launch (words of 4 bytes, a value each): a batch of resident steps starts and ends with a dozen small clears --
// counters, queue heads, cell counts, ghost links, the fixed-point totals -- and a hipMemsetAsync apiece is a launch apiece (~16 of them, 7 us
// apart, in front of every batch: a tenth of a 20-step batch)
constexpr int CLEAR_MAX = 12;
struct ClearList { unsigned* p[CLEAR_MAX]; unsigned long long words[CLEAR_MAX]; unsigned val[CLEAR_MAX]; int n; };
__global__ void __launch_bounds__(256) sz_k_clear_many(ClearList L) {
  for (int e = 0; e < L.n; e++) {
    unsigned* const p = L.p[e]; const unsigned long long n = L.words[e]; const unsigned v = L.val[e];
    for (unsigned long long q = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; q < n; q += (unsigned long long)gridDim.x * blockDim.x) p[q] = v;
  }
}
__global__ void sz_k_zero_int(int* p, const int* cnt, int ci, int add) {
  int n = (ci >= 0 ? cnt[ci] : 0) + add;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}

// ring orientation sign (sign of GO._signed_area), floes [first, M)
__global__ void sz_k_osign(State S, int first) {
  int M = S.cnt[C_M];
  for (int i = first + blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
    int o = S.voff[i], n = S.voff[i + 1] - o;
    double a = ring_signed_area((const double*)(S.vxy + o), (const double*)(S.vxy + o) + 1, n, 2);
    S.osign[i] = a >= 0.0 ? 1 : -1;
    double x0 = __builtin_inf(), x1 = -__builtin_inf(), y0 = __builtin_inf(), y1 = -__builtin_inf();
    for (int q = 0; q < n; q++) { const double2 p = S.vxy[o + q]; x0 = fmin(x0, p.x); x1 = fmax(x1, p.x); y0 = fmin(y0, p.y); y1 = fmax(y1, p.y); }
    S.bbx0[i] = x0; S.bbx1[i] = x1; S.bby0[i] = y0; S.bby1[i] = y1;
    double al = S.alpha[i];
    S.trig[2 * i] = cos(al); S.trig[2 * i + 1] = sin(al);
  }
}
__global__ void sz_k_elem_osign(State S) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < S.nelem; e += gridDim.x * blockDim.x) {
    int o = S.eoff[e], n = S.eoff[e + 1] - o;
    double a = ring_signed_area(S.ex + o, S.ey + o, n);
    S.eosign[e] = a >= 0.0 ? 1 : -1;
    double x0 = __builtin_inf(), x1 = -__builtin_inf(), y0 = __builtin_inf(), y1 = -__builtin_inf();
    for (int q = 0; q < n; q++) { x0 = fmin(x0, S.ex[o + q]); x1 = fmax(x1, S.ex[o + q]); y0 = fmin(y0, S.ey[o + q]); y1 = fmax(y1, S.ey[o + q]); }
    S.ebb[4 * e] = x0; S.ebb[4 * e + 1] = x1; S.ebb[4 * e + 2] = y0; S.ebb[4 * e + 3] = y1;
  }
}

// Grid geometry lives in bounds[0..7] = x0, y0, cell size x / y, cells x / y, wrap x / y.  It is either fitted to
// the centroids every call (sz_k_bounds: no wrap, every floe inside) or fixed by the host for the resident steps
// (static grid: the domain box, indices WRAP in a periodic direction -- a ghost then shares the cell of its
// parent's image -- and are CLAMPED in a non-periodic one; both maps are 1-Lipschitz in cell units, so floes
// closer than one cell size still land in adjacent cells).
struct GridGeo { double x0, y0, csx, csy; int ncx, ncy, wrapx, wrapy; };
__device__ __forceinline__ GridGeo grid_geo(const State& S) {
  GridGeo g;
  g.x0 = S.bounds[0]; g.y0 = S.bounds[1]; g.csx = S.bounds[2]; g.csy = S.bounds[3];
  g.ncx = (int)S.bounds[4]; g.ncy = (int)S.bounds[5]; g.wrapx = (int)S.bounds[6]; g.wrapy = (int)S.bounds[7];
  return g;
}
__device__ __forceinline__ int cell_fold(int i, int n, int wrap) {
  if (wrap) { i %= n; return i < 0 ? i + n : i; }
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
__device__ __forceinline__ void cell_of(const GridGeo& g, double x, double y, int& ix, int& iy) {
  ix = cell_fold((int)floor((x - g.x0) / g.csx), g.ncx, g.wrapx);
  iy = cell_fold((int)floor((y - g.y0) / g.csy), g.ncy, g.wrapy);
}
// one floe into its cell: the first CELL_K arrivals sit in the cell's bucket, later ones on its overflow chain (the order
// inside a cell is arbitrary, the consumers sort by order key)
__device__ __forceinline__ void cell_insert(const State& S, const GridGeo& g, int i, double x, double y) {
  int ix, iy; cell_of(g, x, y, ix, iy);
  const int c = iy * g.ncx + ix;
  const int s = atomicAdd(&S.cell_cnt[c], 1);
  if (s < CELL_K) S.cell_slots[(size_t)c * CELL_K + s] = i;
  else S.cell_items[i] = atomicExch(&S.cell_ovf[c], i + 1) - 1;
}

// ============================================================================ ghosts (A1)
// add_ghosts! (collisions.jl:1060-1174) runs an east/west pass and then a north/south pass over
// the parents.  Whether a parent gets a ghost in a pass depends on that parent alone (the x-swap of
// the first pass does not touch y), so both passes are planned by ONE flag kernel, ordered by ONE
// scan of int4 {E/W ghosts, E/W ring points, N/S ghosts, N/S ring points} and carried out by ONE
// fill kernel in which each thread does its parent's E/W step and then its N/S step (which copies
// the E/W ghost the same thread has just made).  Rows come out in the reference's order: all E/W
// ghosts by parent, then all N/S ghosts by parent with the ghost-of-ghost before the parent's copy.

// int4 exclusive scan (same three-kernel scheme as the int scan)
__device__ __forceinline__ int4 add4(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ int4 shfl_up4(int4 v, int d) { return make_int4(__shfl_up(v.x, d), __shfl_up(v.y, d), __shfl_up(v.z, d), __shfl_up(v.w, d)); }
__device__ __forceinline__ int4 block_exclusive_scan4(int4 v, int4* total) {
  __shared__ int4 wsum4[SCAN_B / 64];
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int4 inc = v;
  for (int d = 1; d < 64; d <<= 1) { int4 t = shfl_up4(inc, d); if (lane >= d) inc = add4(inc, t); }
  if (lane == 63) wsum4[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    int4 w = lane < (int)(blockDim.x >> 6) ? wsum4[lane] : make_int4(0, 0, 0, 0), wi = w;
    for (int d = 1; d < 64; d <<= 1) { int4 t = shfl_up4(wi, d); if (lane >= d) wi = add4(wi, t); }
    if (lane < (int)(blockDim.x >> 6)) wsum4[lane] = make_int4(wi.x - w.x, wi.y - w.y, wi.z - w.z, wi.w - w.w);
    if (lane == 63) *total = wi;
  }
  __syncthreads();
  int4 o = wsum4[wid];
  int4 res = make_int4(inc.x - v.x + o.x, inc.y - v.y + o.y, inc.z - v.z + o.z, inc.w - v.w + o.w);
  __syncthreads();
  return res;
}
// ---------------------------------------------------------------- single-pass scan (decoupled look-back)
// A scan over M elements in ONE launch: every workgroup scans its tile, publishes the tile total (AGG), adds up
// its predecessors' totals until it meets one that already knows its inclusive prefix (INC), then publishes
// its own.  The status words carry the launch number (`epoch`), so nothing has to be cleared between
// launches.  Workgroups are dispatched in index order and a workgroup only waits for lower indices; the wait
// is bounded (ERR_SCAN instead of a hang should the protocol ever be broken).
constexpr int ERR_SCAN = 8192;
__device__ __forceinline__ int4 ld4_agent(const int4* p) {
  int* q = (int*)p;
  return make_int4(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st4_agent(int4* p, int4 v) {
  int* q = (int*)p;
  __hip_atomic_store(q, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(q + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(q + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// called by every thread of the workgroup with the tile total; returns the sum of all earlier tiles
__device__ __forceinline__ int4 lookback_prefix4(State& S, int4 tot, unsigned epoch, int tile = -1) {
  __shared__ int4 s_prefix;
  const int b = tile >= 0 ? tile : (int)blockIdx.x;      // (tile: the scan's workgroups are the tail of a larger launch)
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (b == 0) {
      if (lane == 0) {
        st4_agent(&S.lb_inc[0], tot);
        __hip_atomic_store(&S.lb_flag[0], (epoch << 2) | 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        s_prefix = make_int4(0, 0, 0, 0);
      }
    } else {
      if (lane == 0) {
        st4_agent(&S.lb_agg[b], tot);
        __hip_atomic_store(&S.lb_flag[b], (epoch << 2) | 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      int4 running = make_int4(0, 0, 0, 0);
      int p = b - 1, spins = 0;
      for (;;) {
        const int q = p - lane;                       // lane 0 looks at the nearest predecessor
        unsigned f = 0;
        if (q >= 0) f = __hip_atomic_load(&S.lb_flag[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = q < 0 || (f >> 2) == epoch;
        const unsigned long long isinc = __ballot(q >= 0 && ready && (f & 3u) == 2u);
        const int firstinc = isinc ? __ffsll((long long)isinc) - 1 : 64;
        const unsigned long long need = firstinc >= 63 ? ~0ull : ((2ull << firstinc) - 1ull);
        if (__ballot(!ready) & need) {
          if (++spins > (1 << 22)) { if (lane == 0) atomicOr(&S.cnt[C_ERR], ERR_SCAN); break; }
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        int4 v = make_int4(0, 0, 0, 0);
        if (q >= 0 && lane <= firstinc) v = lane == firstinc ? ld4_agent(&S.lb_inc[q]) : ld4_agent(&S.lb_agg[q]);
        for (int d = 32; d >= 1; d >>= 1) v = add4(v, make_int4(__shfl_xor(v.x, d), __shfl_xor(v.y, d), __shfl_xor(v.z, d), __shfl_xor(v.w, d)));
        running = add4(running, v);
        if (firstinc < 64) break;
        p -= 64;
        if (p < 0) break;
      }
      if (lane == 0) {
        st4_agent(&S.lb_inc[b], add4(running, tot));
        __hip_atomic_store(&S.lb_flag[b], (epoch << 2) | 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        s_prefix = running;
      }
    }
  }
  __syncthreads();
  const int4 r = s_prefix;
  __syncthreads();
  return r;
}

// !isempty(intersect_polys(poly, boundary.poly)) of collisions.jl:889: the wall rectangle extends
// half a domain outward, so a positive-area overlap exists iff a vertex lies strictly beyond the wall
__device__ __forceinline__ int ghost_dir(const State& S, int i, int axis) {
  const int maxb = axis == 0 ? 2 : 0, minb = axis == 0 ? 3 : 1;
  double maxv = S.eval[maxb], minv = S.eval[minb];
  double c = axis == 0 ? S.cx[i] : S.cy[i], r = S.rmax[i];
  int dir = 0;
  if (c - r < minv) dir = 1; else if (c + r > maxv) dir = -1;
  if (dir != 0) {
    // "a vertex strictly beyond the wall" == the ring's box reaches strictly beyond it
    bool beyond = dir > 0 ? ((axis == 0 ? S.bbx0[i] : S.bby0[i]) < minv) : ((axis == 0 ? S.bbx1[i] : S.bby1[i]) > maxv);
    if (!beyond) dir = 0;
  }
  return dir;
}

// plan of both passes for every parent: gplan[i] = {E/W ghosts, E/W points, N/S ghosts, N/S points}
// `drop_old`: ghosts of the previous step are still attached (their removal, simulation.jl:138-144,
// was deferred because nothing after the collision kernels looks at them): detach them here.
__device__ __forceinline__ int4 ghost_plan(State& S, int i, int N, int drop_old, int commit) {
  if (drop_old) {
    S.ngh[i] = 0;
    for (int q = 0; q < MAX_GHOSTS; q++) S.gh[i * MAX_GHOSTS + q] = -1;
    if (i == 0 && !commit) { S.cnt[C_M] = N; S.cnt[C_NV] = S.voff[N]; }
  }
  int dx = 0, dy = 0;
  if (S.status[i] == SZ_ACTIVE && S.ghost_id[i] == 0) {
    if (S.any_periodic_ew) dx = ghost_dir(S, i, 0);
    if (S.any_periodic_ns) dy = ghost_dir(S, i, 1);
  }
  int nv = S.voff[i + 1] - S.voff[i];
  int gew = dx != 0 ? 1 : 0, gns = dy != 0 ? 1 + gew : 0;
  if (i == 0 && !commit) S.cnt[C_NGHOSTS] = 0;    // last step's count is kept until here for the stats
  S.gflag[i] = (dx + 1) | ((dy + 1) << 2);
  int4 plan = make_int4(gew, gew * nv, gns, gns * nv);
  S.gplan[i] = plan;
  return plan;
}
// flag kernel + the whole int4 scan of the plan in one launch (look-back scan).  commit (resident steps, always
// with drop_old): the thread that sees the totals also commits the new counts (ghost_commit below), so that no
// single-workgroup launch has to follow the fill kernel; the fill kernel then takes its base from C_N.
// nh: the number of parents when the host knows it (resident steps: it never changes) -- the first loads then do
// not wait for the counter block; < 0: read it
__global__ void __launch_bounds__(SCAN_B) sz_k_ghost_flag_scan(State S, int drop_old, int commit, unsigned epoch, int nh) {
  __shared__ int4 tot;
  if (stopped(S)) return;
  const int n = nh >= 0 ? nh : S.cnt[C_N];
  const int base = blockIdx.x * SCAN_B;
  if (base >= n && blockIdx.x != 0) return;          // tiles past the end: nobody waits for them
  const int i = base + threadIdx.x;
  const int4 v = i < n ? ghost_plan(S, i, n, drop_old, commit) : make_int4(0, 0, 0, 0);
  const int4 ex = block_exclusive_scan4(v, &tot);
  const int4 before = lookback_prefix4(S, tot, epoch);
  if (i < n) S.gscan4[i] = add4(ex, before);
  if (n == 0 ? i == 0 : i == n - 1) {
    const int4 T = n == 0 ? make_int4(0, 0, 0, 0) : add4(add4(ex, before), v);
    S.gtot4[0] = T;
    if (commit) {
      const int newg = T.x + T.z, newv = T.y + T.w, nv0 = S.voff[n];
      const bool fits = n + newg <= S.capM && nv0 + newv <= S.capV;       // else the fill kernel raises the error
      const int M = fits ? n + newg : n;
      S.cnt[C_M] = M; S.cnt[C_NV] = fits ? nv0 + newv : nv0; S.cnt[C_NGHOSTS] = fits ? newg : 0;
      if (fits) S.voff[M] = nv0 + newv;
    }
  }
}

__device__ __forceinline__ void copy_floe_row(State& S, int dst, int src) {
  S.cx[dst] = S.cx[src]; S.cy[dst] = S.cy[src]; S.rmax[dst] = S.rmax[src]; S.area[dst] = S.area[src];
  S.height[dst] = S.height[src]; S.mass[dst] = S.mass[src]; S.moment[dst] = S.moment[src];
  S.alpha[dst] = S.alpha[src]; S.u[dst] = S.u[src]; S.v[dst] = S.v[src]; S.xi[dst] = S.xi[src];
  S.overarea[dst] = S.overarea[src]; S.id[dst] = S.id[src]; S.status[dst] = S.status[src];
  S.osign[dst] = S.osign[src];
  S.bbx0[dst] = S.bbx0[src]; S.bbx1[dst] = S.bbx1[src]; S.bby0[dst] = S.bby0[src]; S.bby1[dst] = S.bby1[src];
  S.cfx[dst] = 0.0; S.cfy[dst] = 0.0; S.ctrq[dst] = 0.0;
  S.ngh[dst] = 0;
}

// Ghost rows are made by a whole wavefront per flagged parent.  The geometry that the two passes
// change (centroid, box, this lane's ring points) of the parent and of its up to three ghosts is kept
// in REGISTERS through both passes and stored once at the end, so the chain of one parent is two
// memory round trips instead of one per ghost and per swap.  The arithmetic is the reference's:
// a ghost is a copy translated by (+-L, 0) / (0, +-L) (ghosts_on_bounds!, collisions.jl:881-901), the
// swap translates parent and its own new ghost in opposite directions (:942-950, :994-1000).
// WIDE: rings of 65..128 points (a second point per lane: x1, y1); the candidate-list pass of the resident steps is compiled
// without it (rings of at most 64 points) to stay inside the register budget of the launch it rides in
// The wavefront-uniform part of a row -- centroid and ring box -- is spread over the lanes as well: lane q < 6 carries column q
// of {cx, cy, bbx0, bbx1, bby0, bby1} in `pv` (a row is then 3 doubles per lane, not 8: four rows are live in the corner case).
template <bool WIDE> struct RigT { double pv, x0, y0, x1, y1; };
template <> struct RigT<false> { double pv, x0, y0; };
__device__ __forceinline__ bool rig_lane_is_x(int lane) { return lane == 0 || lane == 2 || lane == 3; }
template <bool WIDE>
__device__ __forceinline__ RigT<WIDE> rig_shift(RigT<WIDE> r, int lane, double dx, double dy) {
  r.pv += rig_lane_is_x(lane) ? dx : dy;
  r.x0 += dx; r.y0 += dy;
  if constexpr (WIDE) { r.x1 += dx; r.y1 += dy; }
  return r;
}
__device__ __forceinline__ double* rig_col(const State& S, int lane) {
  return lane == 0 ? S.cx : lane == 1 ? S.cy : lane == 2 ? S.bbx0 : lane == 3 ? S.bbx1 : lane == 4 ? S.bby0 : S.bby1;
}
template <bool WIDE>
__device__ __forceinline__ void rig_store(State& S, int lane, int f, int vo, int n, const RigT<WIDE>& r) {
  if (lane < 6) rig_col(S, lane)[f] = r.pv;
  if (S.body_rings) return;                      // the ring is the parent's, in its body frame: only the pose moves
  if (lane < n) S.vxy[vo + lane] = make_double2(r.x0, r.y0);
  if constexpr (WIDE) { if (lane + 64 < n) S.vxy[vo + lane + 64] = make_double2(r.x1, r.y1); }
}

// One flagged parent, by a whole wavefront: both passes of add_ghosts! for parent i.  fl: its flags (dx+1 | (dy+1)<<2),
// sc: the exclusive prefix of {E/W ghosts, E/W points, N/S ghosts, N/S points} over the flagged parents before it
// (storage order), T: the totals, M0 / NV0: first free floe row / ring point; vo, n: the parent's ring.
// Memory order is the whole cost (every dependent batch is one HBM round trip): everything the parent's row holds is
// asked for in ONE batch -- the scalar columns one per lane, the ring one point per lane -- and stored afterwards.
// The up to three ghosts are named registers (no indexed arrays: those went to scratch memory): G0 is the E/W ghost, or the
// N/S ghost of a parent that has no E/W one; G1 the N/S ghost of G0 and G2 the parent's own N/S ghost (corner parents).
template <bool WIDE>
__device__ __forceinline__ void ghost_fill_parent(State& S, const GridGeo& geo, int lane, int i, int fl, int4 sc, int4 T, int M0, int NV0, int bin,
                                                  int vo, int n, const double* wall) {
  using R = RigT<WIDE>;
  const int dir0 = (fl & 3) - 1, dir1 = ((fl >> 2) & 3) - 1;
  // ---- loads
  const int ngh_old = S.ngh[i];
  R P;
  P.pv = lane < 6 ? rig_col(S, lane)[i] : 0.0;
  const bool body = S.body_rings != 0;
  { const double2 p = !body && lane < n ? S.vxy[vo + lane] : make_double2(0.0, 0.0); P.x0 = p.x; P.y0 = p.y; }
  if constexpr (WIDE) { const double2 p = !body && lane + 64 < n ? S.vxy[vo + lane + 64] : make_double2(0.0, 0.0); P.x1 = p.x; P.y1 = p.y; }
  const double tc = body ? S.trig[2 * i] : 0.0, ts = body ? S.trig[2 * i + 1] : 0.0;
  const double rmx = S.rec32 ? S.rmax[i] : 0.0;
  // the copied scalar columns (deepcopy of the parent, collisions.jl:893): lane q < 10 carries double column q
  double* const dcol = lane == 0 ? S.rmax : lane == 1 ? S.area : lane == 2 ? S.height : lane == 3 ? S.mass : lane == 4 ? S.moment :
                       lane == 5 ? S.alpha : lane == 6 ? S.u : lane == 7 ? S.v : lane == 8 ? S.xi : S.overarea;
  const double dval = lane < 10 ? dcol[i] : 0.0;
  const long long idv = S.id[i]; const int stv = S.status[i]; const signed char osv = S.osign[i];
  const long long okp = S.okey[i];
  const long long oki = S.tiled ? okp : 0;
  State::Fam* const F = S.fam + i;            // the family record of the Dict rule (pair_allowed_fam), as the inline ghost maker leaves it
  if (ngh_old != 0) { if (lane == 0) atomicOr(&S.cnt[C_ERR], ERR_GHOSTS_PER_PARENT); return; }
  if (n > (WIDE ? 128 : 64)) { if (lane == 0) atomicOr(&S.cnt[C_ERR], ERR_CAP_RING); return; }
  // ---- the two passes, in registers; a ghost is stored as soon as nothing is derived from it any more (at most two
  // ghosts and the parent are live at a time)
  auto put = [&](const R& G, int g, int vb, int gid, long long key) {
    if (lane < 10) dcol[g] = dval;
    if (lane == 10) { S.id[g] = idv; S.ghost_id[g] = (long long)gid; S.okey[g] = key; }
    if (lane == 11) { S.status[g] = stv; S.parent[g] = i; S.ngh[g] = 0; S.osign[g] = osv; }
    if (lane == 12) { S.cfx[g] = 0.0; S.cfy[g] = 0.0; S.ctrq[g] = 0.0; }
    rig_store<WIDE>(S, lane, g, vb, n, G);
    if (lane == 13) {
      if (body) { S.rb_off[g] = vo; S.rb_n[g] = n; S.trig[2 * g] = tc; S.trig[2 * g + 1] = ts; }
      else { S.voff[g] = vb; S.voff[g + 1] = vb + n; }   // neighbours write the same values: rings are packed back to back
    }
    const double gcx = __shfl(G.pv, 0), gcy = __shfl(G.pv, 1);
    if (S.rec32) {
      const double g0 = __shfl(G.pv, 2), g1 = __shfl(G.pv, 3), g2 = __shfl(G.pv, 4), g3 = __shfl(G.pv, 5);
      if (lane == 15) rec32_store(S, g, gcx, gcy, rmx, g0, g1, g2, g3);
    }
    if (bin && lane == 14) cell_insert(S, geo, g, gcx, gcy);
    if (lane == 16) { F->key[gid] = key; F->gid[gid] = (long long)gid; F->cx[gid] = gcx; F->cy[gid] = gcy; }
    if (lane < MAX_GHOSTS) S.gh[g * MAX_GHOSTS + lane] = -1;
    if (S.facc && lane >= 20 && lane < 36) S.facc[(size_t)g * FX_WORDS + (lane - 20)] = 0;      // (fixed-point totals of the new row)
  };
  int s0 = 0, s1 = 0, s2 = 0;
  int ng = 0; bool moved = false;
  R G0 = P;
  if (dir0 != 0) {                          // E/W: the parent's ghost
    const double maxv = wall[2], minv = wall[3], L = maxv - minv, t = dir0 > 0 ? L : -L;
    G0 = rig_shift<WIDE>(P, lane, t, 0.0);
    s0 = M0 + sc.x;
    // parent centroid outside the domain: swap roles with its own new ghost
    double sp = 0.0;
    { const double pc = __shfl(P.pv, 0); if (pc < minv) sp = L; else if (maxv < pc) sp = -L; }
    if (sp != 0.0) { P = rig_shift<WIDE>(P, lane, sp, 0.0); moved = true; G0 = rig_shift<WIDE>(G0, lane, -sp, -0.0); }
    ng = 1;
  }
  if (dir1 != 0) {                          // N/S: ghosts of the existing ghost first, then the parent's (every copy has the parent's ring size)
    const double maxv = wall[0], minv = wall[1], L = maxv - minv, t = dir1 > 0 ? L : -L;
    const int gbase = M0 + T.x + sc.z, vb = NV0 + T.y + sc.w;
    double sp = 0.0;
    { const double pc = __shfl(P.pv, 1); if (pc < minv) sp = L; else if (maxv < pc) sp = -L; }
    if (ng == 1) {
      {
        const R G1 = rig_shift<WIDE>(G0, lane, 0.0, t);
        put(G0, s0, NV0 + sc.y, 1, S.tiled ? ((long long)1 << 40) + oki * 4 : (long long)s0);
        s1 = gbase;
        put(G1, s1, vb, 2, S.tiled ? ((long long)2 << 40) + oki * 4 : (long long)gbase);
      }
      R G2 = rig_shift<WIDE>(P, lane, 0.0, t); s2 = gbase + 1;
      if (sp != 0.0) { P = rig_shift<WIDE>(P, lane, 0.0, sp); moved = true; G2 = rig_shift<WIDE>(G2, lane, -0.0, -sp); }
      put(G2, s2, vb + n, 3, S.tiled ? ((long long)2 << 40) + oki * 4 + 1 : (long long)(gbase + 1));
      ng = 3;
    } else {
      G0 = rig_shift<WIDE>(P, lane, 0.0, t); s0 = gbase;
      if (sp != 0.0) { P = rig_shift<WIDE>(P, lane, 0.0, sp); moved = true; G0 = rig_shift<WIDE>(G0, lane, -0.0, -sp); }
      put(G0, s0, vb, 1, S.tiled ? ((long long)2 << 40) + oki * 4 : (long long)gbase);
      ng = 1;
    }
  } else if (ng == 1) put(G0, s0, NV0 + sc.y, 1, S.tiled ? ((long long)1 << 40) + oki * 4 : (long long)s0);
  if (moved) {
    rig_store<WIDE>(S, lane, i, vo, n, P);
    if (S.rec32) {
      const double q0 = __shfl(P.pv, 0), q1 = __shfl(P.pv, 1), q2 = __shfl(P.pv, 2), q3 = __shfl(P.pv, 3), q4 = __shfl(P.pv, 4), q5 = __shfl(P.pv, 5);
      if (lane == 15) rec32_store(S, i, q0, q1, rmx, q2, q3, q4, q5);
    }
  }
  if (lane < MAX_GHOSTS) S.gh[i * MAX_GHOSTS + lane] = lane < ng ? (lane == 0 ? s0 : lane == 1 ? s1 : s2) : -1;
  if (lane == 0) S.ngh[i] = ng;
  {
    const double pcx = __shfl(P.pv, 0), pcy = __shfl(P.pv, 1);      // the parent where it now lies
    if (lane == 16) { F->key[0] = okp; F->gid[0] = 0; F->cx[0] = pcx; F->cy[0] = pcy; F->r = S.rmax[i]; F->n = ng + 1; }
  }
}

// gscan4 holds the exclusive int4 scan of gplan, gtot4[0] the totals
__global__ void __launch_bounds__(256) sz_k_ghost_fill(State S, int committed, int bin, int nh) {
  if (stopped(S)) return;
  const GridGeo geo = grid_geo(S);
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int N = nh >= 0 ? nh : S.cnt[C_N];
  // the first chunk's flags are asked for together with the totals (one round trip, not two)
  int flm0 = 5;
  {
    unsigned np2 = 1; while ((int)np2 < N) np2 <<= 1;
    const unsigned qv = (unsigned)(wave * 8 + lane);
    const int mine = (int)((qv * 0x9E3779B1u) & (np2 - 1));
    if (lane < 8 && qv < np2 && mine < N) flm0 = S.gflag[mine];
  }
  const int M0 = committed ? N : S.cnt[C_M], NV0 = committed ? S.voff[N] : S.cnt[C_NV];
  const double wall[4] = { S.eval[0], S.eval[1], S.eval[2], S.eval[3] };
  int4 T = S.gtot4[0];
  if (M0 + T.x + T.z > S.capM) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&S.cnt[C_ERR], ERR_CAP_FLOES); return; }
  if (NV0 + T.y + T.w > S.capV) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&S.cnt[C_ERR], ERR_CAP_VERTS); return; }
  // chunk c looks at parents c, c + nchunks, c + 2 nchunks, ..: floes near one wall have consecutive
  // indices, striding spreads them over the wavefronts
  // Only GH_CH parents per wavefront visit, taken in a scattered order (multiplicative permutation of
  // [0, 2^k)): flagged parents are a few per cent but sit in rows/columns of the index space, and a
  // wavefront works its flagged parents off one after the other.
  constexpr int GH_CH = 8;                 // (the prefetch above assumes 8)
  unsigned np2 = 1; while ((int)np2 < N) np2 <<= 1;
  const int nchunks = (int)(np2 / GH_CH) > 0 ? (int)(np2 / GH_CH) : 1;
  for (int chunk = wave; chunk < nchunks; chunk += nwaves) {
    unsigned qv = (unsigned)(chunk * GH_CH + lane);
    int mine = (int)((qv * 0x9E3779B1u) & (np2 - 1));
    int flm = chunk == wave ? flm0 : ((lane < GH_CH && qv < np2 && mine < N) ? S.gflag[mine] : 5);   // 5 = (0+1) | (0+1)<<2: no ghost
    unsigned long long todo = __ballot(flm != 5);
    while (todo) {
      int src_lane = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int i = __shfl(mine, src_lane);
      const int fl = __shfl(flm, src_lane);
      const int vo = S.voff[i];
      ghost_fill_parent<true>(S, geo, lane, i, fl, S.gscan4[i], T, M0, NV0, bin, vo, S.voff[i + 1] - vo, wall);
    }
  }
}

// ---------------------------------------------------------------- the ghost pass of the resident steps: ONE launch
// Whether a parent gets ghosts depends on its centroid, rmax, ring box and status alone (ghost_dir), i.e. on what the
// kernel that last placed the floe already had in registers.  In the resident steps that kernel (integrator, halo
// unpack; the seeding launch after an upload) appends the few parents near a periodic wall -- 2 % of a 10 k field -- to a
// candidate list {parent, flags, ring points}; this kernel gives every entry a wavefront, which finds its place in the
// reference's ghost order (all E/W ghosts by parent, then all N/S ghosts by parent: the prefix over the entries with a
// smaller parent index, read straight off the list), makes the parent's ghosts (ghost_fill_parent) and, in the first
// wavefront, commits the new counts.  It replaces flag + scan over ALL parents and the fill launch (8.6 + 14.6 us at
// 10 k floes for ~240 ghosts).  Two lists alternate: the integrator of a step fills the one the next step consumes, and
// the consumer clears the counter of the other one.  The host falls back to the two-launch path when the list is long
// (the prefix is quadratic in its length) or stale (process-mode calls).
__device__ __forceinline__ int ghost_flag_of(const double* ev, int per_ew, int per_ns, double cx, double cy, double r,
                                             double bx0, double bx1, double by0, double by1, bool active) {
  int dx = 0, dy = 0;
  if (active) {
    if (per_ew) { if (cx - r < ev[3]) dx = bx0 < ev[3] ? 1 : 0; else if (cx + r > ev[2]) dx = bx1 > ev[2] ? -1 : 0; }
    if (per_ns) { if (cy - r < ev[1]) dy = by0 < ev[1] ? 1 : 0; else if (cy + r > ev[0]) dy = by1 > ev[0] ? -1 : 0; }
  }
  return (dx + 1) | ((dy + 1) << 2);
}
// thread-per-floe kernels: one counter atomic per wavefront
__device__ __forceinline__ void ghost_candidate_wave(const State& S, int list, bool have, int i, int flag, int nv, int vo) {
  const unsigned long long mask = __ballot(have);
  if (!mask) return;
  const int lane = threadIdx.x & 63, first = __ffsll((long long)mask) - 1;
  int base = 0;
  if (lane == first) base = atomicAdd(&S.cnt[C_NGCAND + list], __popcll(mask));
  base = __shfl(base, first);
  if (have) S.gcand[(size_t)list * S.capM + base + __popcll(mask & ((1ull << lane) - 1ull))] = make_int4(i, flag, nv, vo);
}
__device__ __forceinline__ void ghost_candidate_one(const State& S, int list, int i, int flag, int nv, int vo) {
  S.gcand[(size_t)list * S.capM + atomicAdd(&S.cnt[C_NGCAND + list], 1)] = make_int4(i, flag, nv, vo);
}
__global__ void __launch_bounds__(256) sz_k_ghost_list(State S, int list, int bin, int nh) {
  const StopRegs stop = stop_load(S);
  const GridGeo geo = grid_geo(S);
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int n = S.cnt[C_NGCAND + list];
  const int N = nh >= 0 ? nh : S.cnt[C_N];
  const int NV0 = S.voff[N];
  const double wall[4] = { S.eval[0], S.eval[1], S.eval[2], S.eval[3] };
  const int4* L = S.gcand + (size_t)list * S.capM;
  // the wavefront's own entry and the head of the list are asked for together with its length (entries past the end exist -- the
  // list has capM slots -- and are ignored): the launch's chain is counters + list, parent's row, stores
  constexpr int SPEC = 4;
  int4 sp[SPEC];
#pragma unroll
  for (int c = 0; c < SPEC; c++) { const int q = lane + 64 * c; sp[c] = q < S.capM ? L[q] : make_int4(0, 5, 0, 0); }
  const int4 mine0 = wave < S.capM ? L[wave] : make_int4(0, 5, 0, 0);
  loads_issued();
  if (stop_test(S, stop)) return;
  for (int e = wave; e < (n > 0 ? n : 1); e += nwaves) {
    const int4 mine = e >= n ? make_int4(0x7fffffff, 5, 0, 0) : e == wave ? mine0 : L[e];
    int4 ex = make_int4(0, 0, 0, 0), T = make_int4(0, 0, 0, 0);
    auto take = [&](const int4& o) {
      const int gew = (o.y & 3) != 1 ? 1 : 0, gns = ((o.y >> 2) & 3) != 1 ? 1 + gew : 0;
      const int4 p = make_int4(gew, gew * o.z, gns, gns * o.z);
      T = add4(T, p);
      if (o.x < mine.x) ex = add4(ex, p);
    };
#pragma unroll
    for (int c = 0; c < SPEC; c++) if (lane + 64 * c < n) take(sp[c]);
    for (int q = lane + 64 * SPEC; q < n; q += 64) take(L[q]);
    for (int d = 32; d >= 1; d >>= 1) {
      ex = add4(ex, make_int4(__shfl_xor(ex.x, d), __shfl_xor(ex.y, d), __shfl_xor(ex.z, d), __shfl_xor(ex.w, d)));
      T = add4(T, make_int4(__shfl_xor(T.x, d), __shfl_xor(T.y, d), __shfl_xor(T.z, d), __shfl_xor(T.w, d)));
    }
    const int newg = T.x + T.z, newv = S.body_rings ? 0 : T.y + T.w;      // (ghosts on body-frame rings need no ring points)
    const bool fits = N + newg <= S.capM && NV0 + newv <= S.capV;
    if (e == 0 && lane == 0) {                       // commit (as sz_k_ghost_flag_scan does with commit = 1)
      const int M = fits ? N + newg : N;
      S.cnt[C_M] = M; S.cnt[C_NV] = fits ? NV0 + newv : NV0; S.cnt[C_NGHOSTS] = fits ? newg : 0;
      if (fits) { if (!S.body_rings) S.voff[M] = NV0 + newv; }
      else atomicOr(&S.cnt[C_ERR], N + newg > S.capM ? ERR_CAP_FLOES : ERR_CAP_VERTS);
      S.cnt[C_NGCAND + (1 - list)] = 0;              // the list this step's integrator fills
    }
    if (!fits || e >= n) break;
    ghost_fill_parent<false>(S, geo, lane, mine.x, mine.y, ex, T, N, NV0, bin, mine.w, mine.z, wall);
  }
}
// ---------------------------------------------------------------- "inline" ghosts: no ghost launch in the resident steps
// Which ghosts a parent gets, and where they lie, depends on that parent alone; only their PLACE in the reference's order (all
// E/W ghosts by parent, then all N/S ghosts by parent) needs the other parents.  The order matters in two ways: the serial order
// of the pair loop / Dict rule -- carried by the order keys (okey), which need not be storage positions: (pass << 40) + 4 *
// parent key + k orders the ghosts exactly as the reference does, as in the tiled runs -- and the numbers the host sees
// (partner column of the rows, pair list, fuse lists), which are translated on download from the keys of the step's ghosts
// (sz_api.hip ghost_ref_index).  So the thread that places a parent (the integrator at the end of step t, a seeding launch at
// the start of a batch) makes the parent's ghosts for step t + 1 itself: rows and ring points from ONE packed atomic
// {ghosts, points} -- rings stay packed back to back in allocation order -- copies of the parent's stored row translated by the
// reference's own sequence of additions (ghosts_on_bounds!, collisions.jl:881-1000: ghost = copy + (+-L, 0) / (0, +-L); a parent
// whose centroid has left the domain swaps with its new ghost).  The next neighbour search commits the counts.  The ghost
// launch (11.5 us + a launch boundary at 10 k floes for ~230 ghosts) is gone from the step.
constexpr int MV_RING = 20;      // ring points the one-launch integrator (sz_k_integrate<true>) holds in registers
#ifdef SZ_STAMPS
#ifndef SZ_ISTAMP_FLOE
#define SZ_ISTAMP_FLOE 99
#endif
#define GSTAMP(k) do { if (i == SZ_ISTAMP_FLOE) S.stamps[900 + (k)] = clock64(); } while (0)
#else
#define GSTAMP(k) do {} while (0)
#endif
// up to three translations applied one after the other; an unused place holds (-0.0, -0.0), which changes no bit of any double
struct Shift { double dx0, dy0, dx1, dy1, dx2, dy2; };
__device__ __forceinline__ Shift shift_none() { Shift c; c.dx0 = c.dy0 = c.dx1 = c.dy1 = c.dx2 = c.dy2 = -0.0; return c; }
__device__ __forceinline__ void shift_apply(const Shift& c, double& x, double& y) {
  x += c.dx0; y += c.dy0; x += c.dx1; y += c.dy1; x += c.dx2; y += c.dy2;
}
// the parent's row as the ghosts copy it (deepcopy of the parent, collisions.jl:893): from the registers of the kernel that has just
// placed it, or read back (ghost_row_load) -- every load before the first store, so that a thread pays one round trip, not one per point
struct GhostRow {
  double cx, cy, b0, b1, b2, b3, rmax, area, h, mass, mom, al, u, v, xi, over, tc, ts;
  long long id, oki; int st; signed char os;
  double rx[MV_RING], ry[MV_RING];          // the ring in world coordinates (unused on body-frame rings)
};
__device__ __forceinline__ void ghost_row_load(const State& S, int i, int n, int vo, GhostRow& r) {
  r.cx = S.cx[i]; r.cy = S.cy[i]; r.b0 = S.bbx0[i]; r.b1 = S.bbx1[i]; r.b2 = S.bby0[i]; r.b3 = S.bby1[i];
  r.rmax = S.rmax[i]; r.area = S.area[i]; r.h = S.height[i]; r.mass = S.mass[i]; r.mom = S.moment[i]; r.al = S.alpha[i];
  r.u = S.u[i]; r.v = S.v[i]; r.xi = S.xi[i]; r.over = S.overarea[i];
  r.id = S.id[i]; r.oki = S.okey[i]; r.st = S.status[i]; r.os = S.osign[i];
  const bool body = S.body_rings != 0;
  r.tc = body ? S.trig[2 * i] : 0.0; r.ts = body ? S.trig[2 * i + 1] : 0.0;
#pragma unroll
  for (int k = 0; k < MV_RING; k++) { const double2 p = !body && k < n ? S.vxy[vo + k] : make_double2(0.0, 0.0); r.rx[k] = p.x; r.ry[k] = p.y; }
}
// parent i with row R, flags fl != 5, ring (vo, n <= MV_RING); N parents, ring points of the parents NV0; slot: the allocator
// how many ghosts the flags ask for
__device__ __forceinline__ int ghost_count_of(int fl) { return fl == 5 ? 0 : ((fl & 3) != 1 && ((fl >> 2) & 3) != 1) ? 3 : 1; }
// given: the caller has already drawn the rows and ring points of the ghosts from the allocator ({rows << 32 | points} before them) -- the halo
// unpack of a tiled run allocates a received floe and its ghosts in one go
// RING: where the parent's (moved) ring comes from, `void get(int k, double& x, double& y)` -- the row's registers (RingRegs), or the old ring in
// memory put through the step's motion again (sz_pipeline.hpp RingMoved: the same expressions, the same bits)
struct RingRegs { const GhostRow& R; __device__ __forceinline__ void get(int k, double& x, double& y) const { x = R.rx[k]; y = R.ry[k]; } };
template <class ROW, class RING>
__device__ __forceinline__ void ghost_inline_make(State& S, const GridGeo& geo, const double* wall, int N, int NV0, int slot, int i, int fl, int n, int vo,
                                                  const ROW& R, const RING& ring, const unsigned long long* given = nullptr) {
  const int dir0 = (fl & 3) - 1, dir1 = ((fl >> 2) & 3) - 1;
  const int ng = dir0 != 0 && dir1 != 0 ? 3 : 1;
  const bool body = S.body_rings != 0;
  const int npts = body ? 0 : ng * n;                      // (ghosts on body-frame rings share the parent's ring)
  // one allocation per wavefront for all its parents that make ghosts (the callers are thread-per-parent kernels; same-address atomics
  // are worked off one at a time for the whole chip), handed on by a prefix over the lanes
  const unsigned long long mine = ((unsigned long long)ng << 32) | (unsigned)npts;
  const int lane = threadIdx.x & 63;
  int first = 0;
  unsigned long long pre = 0, tot = 0;          // (a walk over the active lanes: the others hold nothing to shuffle from)
  unsigned long long base = 0;
  if (!given) {
    const unsigned long long act = __ballot(1);
    first = __ffsll((long long)act) - 1;
    for (unsigned long long m = act; m; m &= m - 1) {
      const int l = __ffsll((long long)m) - 1;
      const unsigned long long v = __shfl(mine, l);
      if (l < lane) pre += v;
      tot += v;
    }
    if (lane == first) base = atomicAdd(&S.galloc[slot * 16], tot);      // (its answer is looked at below, after the cell counters have gone out too)
  }
  const double pcx = R.cx, pcy = R.cy, pb0 = R.b0, pb1 = R.b1, pb2 = R.b2, pb3 = R.b3;
  const double c_rmax = R.rmax, c_area = R.area, c_h = R.h;
  const double c_u = R.u, c_v = R.v, c_xi = R.xi, c_over = R.over, tc = R.tc, ts = R.ts;
  const long long idv = R.id, oki = R.oki; const int stv = R.st; const signed char osv = R.os;
  // ---- the two passes as sequences of translations (each applied to centroid, box and ring in turn: the reference's arithmetic)
  // cp: the parent (swaps), c0: the E/W ghost -- or the N/S ghost of a parent without one --, c1: the N/S ghost of the E/W ghost,
  // c2: the parent's own N/S ghost (corner parents)
  Shift cp = shift_none(), c0 = shift_none(), c1 = shift_none(), c2 = shift_none();
  long long k0 = 0, k1 = 0, k2 = 0;
  bool moved = false;
  if (dir0 != 0) {
    const double maxv = wall[2], minv = wall[3], L = maxv - minv, t = dir0 > 0 ? L : -L;
    c0.dx0 = t; c0.dy0 = 0.0;
    double sp = 0.0;
    if (pcx < minv) sp = L; else if (maxv < pcx) sp = -L;
    if (sp != 0.0) { cp.dx0 = sp; cp.dy0 = 0.0; c0.dx1 = -sp; c0.dy1 = -0.0; moved = true; }
    k0 = ((long long)1 << 40) + oki * 4;
  }
  if (dir1 != 0) {
    const double maxv = wall[0], minv = wall[1], L = maxv - minv, t = dir1 > 0 ? L : -L;
    double sp = 0.0;
    if (pcy < minv) sp = L; else if (maxv < pcy) sp = -L;       // (the first pass does not move the parent in y)
    if (dir0 != 0) {
      c1 = c0; c1.dx2 = 0.0; c1.dy2 = t;                      // the ghost of the E/W ghost
      c2.dx0 = cp.dx0; c2.dy0 = cp.dy0; c2.dx1 = 0.0; c2.dy1 = t;       // the parent's own (the parent as the first pass left it)
      if (sp != 0.0) { cp.dx1 = 0.0; cp.dy1 = sp; c2.dx2 = -0.0; c2.dy2 = -sp; moved = true; }
      k1 = ((long long)2 << 40) + oki * 4; k2 = ((long long)2 << 40) + oki * 4 + 1;
    } else {
      c0.dx0 = 0.0; c0.dy0 = t;
      if (sp != 0.0) { cp.dx1 = 0.0; cp.dy1 = sp; c0.dx1 = -0.0; c0.dy1 = -sp; moved = true; }
      k0 = ((long long)2 << 40) + oki * 4;
    }
  }
  // ---- the counters of the cells the ghosts fall into are drawn now, beside the allocation: one round trip for both
  auto cellof = [&](const Shift& c) { double x = pcx, y = pcy; shift_apply(c, x, y); int ix, iy; cell_of(geo, x, y, ix, iy); return iy * geo.ncx + ix; };
  const int cl0 = cellof(c0), cl1 = ng == 3 ? cellof(c1) : -1, cl2 = ng == 3 ? cellof(c2) : -1;
  GSTAMP(10);
  const int s0 = atomicAdd(&S.cell_cnt[cl0], 1);
  const int s1 = cl1 >= 0 ? atomicAdd(&S.cell_cnt[cl1], 1) : 0, s2 = cl2 >= 0 ? atomicAdd(&S.cell_cnt[cl2], 1) : 0;
  GSTAMP(11);
  if (!given) base = __shfl(base, first);
  GSTAMP(12);
  const unsigned long long old = given ? *given : base + pre;
  const int og = (int)(old >> 32), ov = (int)(old & 0xffffffffull);
  // (out of rows or ring points: the error bit, and the allocator is marked POISONED in the word beside it -- its count then says nothing
  //  about how many rows were really written, and the neighbour search that commits it runs the step on the parents alone instead of
  //  walking rows nobody made.  The count itself is left alone: later allocations must keep seeing a plain, ever larger offset.)
  if (og + ng > (S.gcap > 0 ? S.gcap : S.capM - N - S.goff)) { atomicOr(&S.cnt[C_ERR], ERR_CAP_FLOES); atomicOr(&S.galloc[slot * 16 + 1], 1ull); return; }
  const int NR = N + S.goff;          // first row of the step's allocated rows (pipelined steps: one of two regions)
  if (NV0 + ov + npts > S.capV) { atomicOr(&S.cnt[C_ERR], ERR_CAP_VERTS); atomicOr(&S.galloc[slot * 16 + 1], 1ull); return; }
  // ---- stores
  auto put = [&](const Shift& c, int w, long long key) {
    const int g = NR + og + w, vb = NV0 + ov + w * n;
    double gx = pcx, gy = pcy, x0 = pb0, y0 = pb2, x1 = pb1, y1 = pb3;
    shift_apply(c, gx, gy); shift_apply(c, x0, y0); shift_apply(c, x1, y1);
    // (only what the collision kernels read of a ghost: these rows live for one resident step and are never handed to the host -- mass,
    //  moment, alpha, the collision totals and the ghost's own ghost links are not among it; every store less is a slot of the 63 memory
    //  operations the wavefront can have in flight)
    S.rmax[g] = c_rmax; S.area[g] = c_area; S.height[g] = c_h;
    S.u[g] = c_u; S.v[g] = c_v; S.xi[g] = c_xi; S.overarea[g] = c_over;
    S.id[g] = idv; S.ghost_id[g] = (long long)(w + 1); S.okey[g] = key;
    S.status[g] = stv; S.parent[g] = i; S.osign[g] = osv;
    S.cx[g] = gx; S.cy[g] = gy; S.bbx0[g] = x0; S.bbx1[g] = x1; S.bby0[g] = y0; S.bby1[g] = y1;
    GSTAMP(16);
    if (body) { S.rb_off[g] = vo; S.rb_n[g] = n; S.trig[2 * g] = tc; S.trig[2 * g + 1] = ts; }
    else {
      S.voff[g] = vb; S.voff[g + 1] = vb + n;                // (the next allocation writes the same value: rings are packed back to back)
#pragma unroll
      for (int k = 0; k < MV_RING; k++) if (k < n) { double x, y; ring.get(k, x, y); shift_apply(c, x, y); S.vxy[vb + k] = make_double2(x, y); }
    }
    GSTAMP(17);
    if (S.rec32) rec32_store(S, g, gx, gy, c_rmax, x0, x1, y0, y1);
    if (S.crec) crec_store_all(S, g, gx, gy, c_rmax, idv, key, n, osv, body ? vo : vb, i, 0, x0, x1, y0, y1, c_u, c_v, c_xi, c_area, c_h, (long long)(w + 1));
    S.gkeys[(size_t)slot * S.capM + og + w] = key;
    // (the fixed-point totals of the new row are cleared by the neighbour search of its step, not here: the rows of step t + 1 are the rows
    //  of step t again, and while this thread makes a ghost another thread of the same launch may still be reading the totals of ITS
    //  step-t ghost in that row)
  };
  GSTAMP(13);
  {
    // the cell entries BEFORE the copies: both need answers (allocation, counters), and a wait for an answer after the ~70 stores of
    // a copy would be a wait for all of them
    auto place = [&](int cl, int sl, int g) {            // as cell_insert()
      if (sl < CELL_K) S.cell_slots[(size_t)cl * CELL_K + sl] = g;
      else S.cell_items[g] = atomicExch(&S.cell_ovf[cl], g + 1) - 1;
    };
    place(cl0, s0, NR + og);
    if (cl1 >= 0) { place(cl1, s1, NR + og + 1); place(cl2, s2, NR + og + 2); }
  }
  GSTAMP(14);
  put(c0, 0, k0);
  if (ng == 3) { put(c1, 1, k1); put(c2, 2, k2); }
  GSTAMP(15);
  if (moved) {                                               // the parent swapped with its ghost(s)
    double px = pcx, py = pcy, x0 = pb0, y0 = pb2, x1 = pb1, y1 = pb3;
    shift_apply(cp, px, py); shift_apply(cp, x0, y0); shift_apply(cp, x1, y1);
    // (pipelined steps: this runs beside the narrow phase and the forcings of the step BEFORE, which read the parent's columns -- they are
    //  brought up to date from the record by that step's update, sz_pipeline.hpp)
    if (!S.pipe) { S.cx[i] = px; S.cy[i] = py; S.bbx0[i] = x0; S.bbx1[i] = x1; S.bby0[i] = y0; S.bby1[i] = y1; }
    if (!body) {
#pragma unroll
      for (int k = 0; k < MV_RING; k++) if (k < n) { double x, y; ring.get(k, x, y); shift_apply(cp, x, y); S.vxy[vo + k] = make_double2(x, y); }
    }
    if (S.rec32) rec32_store(S, i, px, py, c_rmax, x0, x1, y0, y1);
    if (S.crec) crec_store_place(S, i, px, py, x0, x1, y0, y1);
  }
  for (int q = 0; q < MAX_GHOSTS; q++) S.gh[i * MAX_GHOSTS + q] = q < ng ? NR + og + q : -1;
  S.ngh[i] = ng | (S.pipe && moved ? 0x100 : 0);          // (0x100: the parent swapped with its ghost -- only pipelined steps look at it)
  if (S.crec) S.crec[(size_t)i * 8 + 2].y = crec_vp(vo, i, ng);
  {
    // the family record of this id for the Dict rule (pair_allowed_fam): the parent as it now lies, then its ghosts -- the values of the rows
    State::Fam* F = S.fam + i;
    double x = pcx, y = pcy; shift_apply(cp, x, y);
    F->key[0] = oki; F->gid[0] = 0; F->cx[0] = x; F->cy[0] = y;
    x = pcx; y = pcy; shift_apply(c0, x, y); F->key[1] = k0; F->gid[1] = 1; F->cx[1] = x; F->cy[1] = y;
    if (ng == 3) {
      x = pcx; y = pcy; shift_apply(c1, x, y); F->key[2] = k1; F->gid[2] = 2; F->cx[2] = x; F->cy[2] = y;
      x = pcx; y = pcy; shift_apply(c2, x, y); F->key[3] = k2; F->gid[3] = 3; F->cx[3] = x; F->cy[3] = y;
    }
    F->r = c_rmax; F->n = ng + 1;
  }
}
// start of a resident batch: the ghosts of its first step, from the parents as they lie (thread per parent)
__global__ void __launch_bounds__(256) sz_k_ghost_inline_seed(State S, int slot, int nh) {
  const GridGeo geo = grid_geo(S);
  const int N = nh >= 0 ? nh : S.cnt[C_N];
  const int NV0 = S.voff[N];
  const double wall[4] = { S.eval[0], S.eval[1], S.eval[2], S.eval[3] };
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int fl = ghost_flag_of(wall, S.any_periodic_ew, S.any_periodic_ns, S.cx[i], S.cy[i], S.rmax[i], S.bbx0[i], S.bbx1[i], S.bby0[i], S.bby1[i],
                                 S.status[i] == SZ_ACTIVE && S.ghost_id[i] == 0);
    if (fl != 5) {
      const int n = ring_n(S, i), vo = ring_off(S, i);
      if (n > MV_RING) { atomicOr(&S.cnt[C_ERR], ERR_CAP_RING); continue; }
      GhostRow R; ghost_row_load(S, i, n, vo, R);
      ghost_inline_make(S, geo, wall, N, NV0, slot, i, fl, n, vo, R, RingRegs{ R });
    }
  }
}

// mixed precision: the rings of the parents into their body frame (offsets from the centroid at alpha = 0, fp32) ...
__global__ void sz_k_body_rings(State S) {
  const int N = S.cnt[C_N];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int o = S.voff[i], n = S.voff[i + 1] - o;
    const double cx = S.cx[i], cy = S.cy[i], ca = S.trig[2 * i], sa = S.trig[2 * i + 1];
    for (int k = 0; k < n; k++) {
      const double2 p = S.vxy[o + k];
      const double dx = p.x - cx, dy = p.y - cy;
      S.ring32[o + k] = make_float2((float)(ca * dx + sa * dy), (float)(-sa * dx + ca * dy));
    }
    S.rb_off[i] = o; S.rb_n[i] = n;
  }
}
// ... and back: world rings (and their boxes) from the body rings and the pose, for everything outside the resident steps
__global__ void sz_k_world_rings(State S) {
  const int N = S.cnt[C_N];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int o = S.voff[i], n = S.voff[i + 1] - o;
    const double cx = S.cx[i], cy = S.cy[i], ca = S.trig[2 * i], sa = S.trig[2 * i + 1];
    for (int k = 0; k < n; k++) {
      const float2 b = S.ring32[o + k];
      S.vxy[o + k] = make_double2((ca * (double)b.x - sa * (double)b.y) + cx, (sa * (double)b.x + ca * (double)b.y) + cy);
    }
  }
}
// the fp32 broad-phase records of all floes as they lie
__global__ void sz_k_rec32_seed(State S) {
  const int M = S.cnt[C_M];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x)
    rec32_store(S, i, S.cx[i], S.cy[i], S.rmax[i], S.bbx0[i], S.bbx1[i], S.bby0[i], S.bby1[i]);
}
// the collision records of the rows [0, n) as the columns hold them (start of a resident batch; the kernels of the batch keep them current)
__global__ void __launch_bounds__(256) sz_k_crec_seed(State S, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    crec_store_all(S, i, S.cx[i], S.cy[i], S.rmax[i], S.id[i], S.okey[i], ring_n(S, i), S.osign[i], ring_off(S, i), S.parent[i], S.ngh[i],
                   S.bbx0[i], S.bbx1[i], S.bby0[i], S.bby1[i], S.u[i], S.v[i], S.xi[i], S.area[i], S.height[i], S.ghost_id[i]);
}
// test hook: how many of the rows [0, n) hold a collision record that differs from the columns it caches (bit for bit; the ghost count of
// a parent is compared with its ngh column) -- 0 after any resident batch, whoever placed the floes in it
__global__ void __launch_bounds__(256) sz_k_crec_check(State S, const double2* rec, int n, unsigned long long* bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double2* r = rec + (size_t)i * 8;
    auto ne = [](double a, double b) { return __double_as_longlong(a) != __double_as_longlong(b); };
    int m = 0;
    m += ne(r[0].x, S.cx[i]) || ne(r[0].y, S.cy[i]);
    m += ne(r[1].x, S.rmax[i]) || __double_as_longlong(r[1].y) != S.id[i];
    m += ne(r[2].x, crec_okf(S.okey[i], ring_n(S, i), S.osign[i])) || ne(r[2].y, crec_vp(ring_off(S, i), S.parent[i], S.ngh[i]));
    m += ne(r[3].x, S.bbx0[i]) || ne(r[3].y, S.bbx1[i]) || ne(r[4].x, S.bby0[i]) || ne(r[4].y, S.bby1[i]);
    m += ne(r[5].x, S.u[i]) || ne(r[5].y, S.v[i]) || ne(r[6].x, S.xi[i]) || ne(r[6].y, S.area[i]);
    m += ne(r[7].x, S.height[i]) || __double_as_longlong(r[7].y) != S.ghost_id[i];
    if (m) atomicAdd(bad, (unsigned long long)m);
  }
}
// seeds candidate list `list` from the parents as they lie (after an upload / a process-mode call), thread per parent
__global__ void sz_k_ghost_seed(State S, int list) {
  const int N = S.cnt[C_N];
  const double ev[4] = { S.eval[0], S.eval[1], S.eval[2], S.eval[3] };
  for (int i0 = blockIdx.x * blockDim.x; i0 < N; i0 += gridDim.x * blockDim.x) {
    const int i = i0 + threadIdx.x;
    int fl = 5, nv = 0, vo = 0;
    if (i < N) {
      fl = ghost_flag_of(ev, S.any_periodic_ew, S.any_periodic_ns, S.cx[i], S.cy[i], S.rmax[i], S.bbx0[i], S.bbx1[i], S.bby0[i], S.bby1[i],
                         S.status[i] == SZ_ACTIVE && S.ghost_id[i] == 0);
      vo = ring_off(S, i); nv = ring_n(S, i);
    }
    ghost_candidate_wave(S, list, fl != 5, i, fl, nv, vo);
  }
}
__device__ __forceinline__ void ghost_commit(State& S) {
  int4 T = S.gtot4[0];
  int newg = T.x + T.z, newv = T.y + T.w;
  if (S.cnt[C_M] + newg <= S.capM && S.cnt[C_NV] + newv <= S.capV) {
    int M = S.cnt[C_M] + newg;
    S.cnt[C_M] = M; S.cnt[C_NV] += newv; S.cnt[C_NGHOSTS] += newg;
    S.voff[M] = S.cnt[C_NV];
  }
  S.gtot4[0] = make_int4(0, 0, 0, 0);      // committed
}
__global__ void sz_k_ghost_commit(State S) {
  if (blockIdx.x == 0 && threadIdx.x == 0) ghost_commit(S);
}
// simulation.jl:138-144
// drop_halo: tiled runs also forget the halo parents (N := owned)
__global__ void sz_k_remove_ghosts(State S, int drop_halo) {
  if (S.retry_stop && S.cnt[C_RETRYSTOP] != 0) return;       // the batch is paused inside a step: its ghosts are still needed
  int N = drop_halo ? S.cnt[C_NOWN] : S.cnt[C_N];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    S.ngh_save[i] = S.ngh[i]; S.ngh[i] = 0;
    for (int q = 0; q < MAX_GHOSTS; q++) { S.gh_save[i * MAX_GHOSTS + q] = S.gh[i * MAX_GHOSTS + q]; S.gh[i * MAX_GHOSTS + q] = -1; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    S.cnt[C_M] = N; S.cnt[C_N] = N; S.cnt[C_NV] = S.voff[N];        // C_NGHOSTS keeps the last step's count
  }
}

// ============================================================================ broad phase (A2, A3)
// Single block.  Prologue: commit the ghosts planned by the ghost kernels (C_M, C_NV).  Then the
// bounding box of all centroids and the largest rmax -> uniform grid geometry; finally the cell
// heads are cleared and the per-step counters reset, so the step needs no separate launches for that.
__global__ void __launch_bounds__(1024) sz_k_bounds(State S, int commit_ghosts) {
  __shared__ double sh[5][16];
  __shared__ int s_ncells;
  if (stopped(S)) return;
  if (threadIdx.x < WARN_SLOTS * 4) S.warn[(threadIdx.x >> 2) * 32 + (threadIdx.x & 3)] = 0;   // guards of the coming update
  if (commit_ghosts) {
    if (threadIdx.x == 0) ghost_commit(S);
    __syncthreads();
  }
  int M = S.cnt[C_M];
  double x0 = __builtin_inf(), y0 = __builtin_inf(), x1 = -__builtin_inf(), y1 = -__builtin_inf(), rm = 0.0;
  // one workgroup reads three columns: eight floes per thread and trip, so that the loads of a trip are all
  // in flight together (a plain loop would pay one memory round trip per floe)
  for (int i0 = threadIdx.x; i0 < M; i0 += 8 * blockDim.x) {
    double ax[8], ay[8], ar[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = i0 + k * (int)blockDim.x; const bool in = i < M;
      ax[k] = in ? S.cx[i] : __builtin_nan(""); ay[k] = in ? S.cy[i] : __builtin_nan(""); ar[k] = in ? S.rmax[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {      // fmin / fmax ignore a NaN operand
      x0 = fmin(x0, ax[k]); x1 = fmax(x1, ax[k]); y0 = fmin(y0, ay[k]); y1 = fmax(y1, ay[k]); rm = fmax(rm, ar[k]);
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    x0 = fmin(x0, __shfl_xor(x0, d)); y0 = fmin(y0, __shfl_xor(y0, d));
    x1 = fmax(x1, __shfl_xor(x1, d)); y1 = fmax(y1, __shfl_xor(y1, d)); rm = fmax(rm, __shfl_xor(rm, d));
  }
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wid] = x0; sh[1][wid] = y0; sh[2][wid] = x1; sh[3][wid] = y1; sh[4][wid] = rm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
      x0 = fmin(x0, sh[0][w]); y0 = fmin(y0, sh[1][w]); x1 = fmax(x1, sh[2][w]); y1 = fmax(y1, sh[3][w]); rm = fmax(rm, sh[4][w]);
    }
    double cs = 2.0 * rm;
    if (!(cs > 0.0)) cs = 1.0;
    if (M == 0) { x0 = y0 = 0.0; x1 = y1 = 0.0; }
    long long ncx, ncy;
    for (;;) {
      ncx = (long long)floor((x1 - x0) / cs) + 1; ncy = (long long)floor((y1 - y0) / cs) + 1;
      if (ncx * ncy <= (long long)S.capCells) break;
      cs *= 2.0;
    }
    S.cnt[C_ITEMCLASS] = 0;                              // narrow-phase size classes of this step
    S.bounds[0] = x0; S.bounds[1] = y0; S.bounds[2] = cs; S.bounds[3] = cs; S.bounds[4] = (double)ncx; S.bounds[5] = (double)ncy;
    S.bounds[6] = 0.0; S.bounds[7] = 0.0;       // a grid fitted to the centroids neither wraps nor clamps
    S.cnt[C_NCELLS] = (int)(ncx * ncy);
    s_ncells = (int)(ncx * ncy);
  }
  __syncthreads();
  for (int q = threadIdx.x; q <= s_ncells; q += blockDim.x) { S.cell_cnt[q] = 0; S.cell_ovf[q] = 0; }
}
// uniform-grid binning as per-cell linked lists (atomic exchange on the cell heads): no counting pass, no scan.
// parents_only: the resident steps keep the lists current themselves (the kernels that place a floe -- integrator,
// ghost fill, halo unpack -- also bin it); this launch then only seeds them with the parents once.
__global__ void sz_k_cell_build(State S, int parents_only) {
  if (stopped(S)) return;
  int M = parents_only ? S.cnt[C_N] : S.cnt[C_M];
  const GridGeo g = grid_geo(S);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) cell_insert(S, g, i, S.cx[i], S.cy[i]);
}

// potential_interaction, collisions.jl:705-710 (bounding circles, strict <)
__device__ __forceinline__ bool circles_touch(const State& S, int a, int b) {
  double dx = S.cx[a] - S.cx[b], dy = S.cy[a] - S.cy[b], rr = S.rmax[a] + S.rmax[b];
  return (dx * dx + dy * dy) < rr * rr;
}

// The Dict rule of collisions.jl:751-775 evaluated without a Dict: the stored ghost-id pair of an
// id pair is the one of the lexicographically first (i, j) -- serial order i asc, j asc -- among
// all instances (parent + ghosts) of the two ids that pass the circle test.
__device__ bool pair_allowed(const State& S, int i, int j) {   // i < j, ids differ, circles touch
  int pi = S.parent[i], pj = S.parent[j];
  int ni = S.ngh[pi] & 0xff, nj = S.ngh[pj] & 0xff;
  if (ni == 0 && nj == 0) return true;
  int ba = 0x7fffffff, bb = 0x7fffffff;
  for (int x = -1; x < ni; x++) {
    int fx = x < 0 ? pi : S.gh[pi * MAX_GHOSTS + x];
    for (int y = -1; y < nj; y++) {
      int fy = y < 0 ? pj : S.gh[pj * MAX_GHOSTS + y];
      bool lt = S.okey[fx] < S.okey[fy];
      int a = lt ? fx : fy, b = lt ? fy : fx;
      if (ba != 0x7fffffff) {
        long long ka = S.okey[a], kb = S.okey[b], kba = S.okey[ba], kbb = S.okey[bb];
        if (ka > kba || (ka == kba && kb >= kbb)) continue;
      }
      if (circles_touch(S, a, b)) { ba = a; bb = b; }
    }
  }
  long long g1, g2, c1, c2;
  if (S.id[ba] > S.id[bb]) { g1 = S.ghost_id[ba]; g2 = S.ghost_id[bb]; } else { g1 = S.ghost_id[bb]; g2 = S.ghost_id[ba]; }
  if (S.id[i] > S.id[j]) { c1 = S.ghost_id[i]; c2 = S.ghost_id[j]; } else { c1 = S.ghost_id[j]; c2 = S.ghost_id[i]; }
  bool A = c1 == g1, B = c2 == g2;
  return (A && B) || (A != B);
}

// ... and on collision records (the lean search of large fields, where the family records cost a wavefront per SIMD): every value of an
// instance comes from its record -- in a pipelined step the parents' COLUMNS are being brought up to date beside this search -- and the
// ghost count is the links' low byte (pipelined steps keep a flag above it)
__device__ bool pair_allowed_rec(const State& S, int i, int j) {   // i before j in the serial order, ids differ, circles touch
  auto key_of = [&](int f) { return __double_as_longlong(S.crec[(size_t)f * 8 + 2].x) & CREC_KEYMASK; };
  auto touch = [&](int a, int b) {
    const double2 pa = S.crec[(size_t)a * 8], pb = S.crec[(size_t)b * 8];
    const double dx = pa.x - pb.x, dy = pa.y - pb.y, rr = S.crec[(size_t)a * 8 + 1].x + S.crec[(size_t)b * 8 + 1].x;
    return (dx * dx + dy * dy) < rr * rr;
  };
  const int pi = (int)(__double_as_longlong(S.crec[(size_t)i * 8 + 2].y) >> 32) & 0x0fffffff, pj = (int)(__double_as_longlong(S.crec[(size_t)j * 8 + 2].y) >> 32) & 0x0fffffff;
  const int ni = S.ngh[pi] & 0xff, nj = S.ngh[pj] & 0xff;
  if (ni == 0 && nj == 0) return true;
  int ba = 0x7fffffff, bb = 0x7fffffff;
  for (int x = -1; x < ni; x++) {
    const int fx = x < 0 ? pi : S.gh[pi * MAX_GHOSTS + x];
    for (int y = -1; y < nj; y++) {
      const int fy = y < 0 ? pj : S.gh[pj * MAX_GHOSTS + y];
      const bool lt = key_of(fx) < key_of(fy);
      const int a = lt ? fx : fy, b = lt ? fy : fx;
      if (ba != 0x7fffffff) {
        const long long ka = key_of(a), kb = key_of(b), kba = key_of(ba), kbb = key_of(bb);
        if (ka > kba || (ka == kba && kb >= kbb)) continue;
      }
      if (touch(a, b)) { ba = a; bb = b; }
    }
  }
  auto id_of = [&](int f) { return __double_as_longlong(S.crec[(size_t)f * 8 + 1].y); };
  auto gid_of = [&](int f) { return __double_as_longlong(S.crec[(size_t)f * 8 + 7].y); };
  long long g1, g2, c1, c2;
  if (id_of(ba) > id_of(bb)) { g1 = gid_of(ba); g2 = gid_of(bb); } else { g1 = gid_of(bb); g2 = gid_of(ba); }
  if (id_of(i) > id_of(j)) { c1 = gid_of(i); c2 = gid_of(j); } else { c1 = gid_of(j); c2 = gid_of(i); }
  const bool A = c1 == g1, B = c2 == g2;
  return (A && B) || (A != B);
}

// The same rule from the family records the inline ghost maker leaves (State::Fam): a side without ghosts is the floe itself (the
// caller has its row), a side with ghosts is ONE contiguous record -- one round trip instead of the chain parent -> links -> rows
// of every instance.  i before j in the serial order; (ki, gi, xi, yi, ri): key, ghost id, centroid, rmax of i, likewise j;
// pi / pj: their parents; fi / fj: the side has ghosts.
__device__ __forceinline__ bool pair_allowed_fam(const State& S, int pi, int pj, bool fi, bool fj,
                                                 long long ki, long long gi, double xi, double yi, double ri,
                                                 long long kj, long long gj, double xj, double yj, double rj) {
  long long kx[4] = { ki, 0, 0, 0 }, gx[4] = { gi, 0, 0, 0 }, ky[4] = { kj, 0, 0, 0 }, gy[4] = { gj, 0, 0, 0 };
  double ax[4] = { xi, 0, 0, 0 }, ay[4] = { yi, 0, 0, 0 }, bx[4] = { xj, 0, 0, 0 }, by[4] = { yj, 0, 0, 0 };
  double ar = ri, br = rj; int nx = 1, ny = 1;
  const State::Fam* FI = S.fam + pi; const State::Fam* FJ = S.fam + pj;
  if (fi) {
#pragma unroll
    for (int q = 0; q < 4; q++) { kx[q] = FI->key[q]; gx[q] = FI->gid[q]; ax[q] = FI->cx[q]; ay[q] = FI->cy[q]; }
    ar = FI->r; nx = FI->n;
  }
  if (fj) {
#pragma unroll
    for (int q = 0; q < 4; q++) { ky[q] = FJ->key[q]; gy[q] = FJ->gid[q]; bx[q] = FJ->cx[q]; by[q] = FJ->cy[q]; }
    br = FJ->r; ny = FJ->n;
  }
  long long bka = 0, bkb = 0, gbx = 0, gby = 0; bool have = false;
#pragma unroll
  for (int x = 0; x < 4; x++) {
#pragma unroll
    for (int y = 0; y < 4; y++) {
      if (x >= nx || y >= ny) continue;
      const bool lt = kx[x] < ky[y];
      const long long ka = lt ? kx[x] : ky[y], kb = lt ? ky[y] : kx[x];
      if (have && (ka > bka || (ka == bka && kb >= bkb))) continue;
      // circles_touch(first, second): differences are squared and the sum of the radii commutes -- the same bits either way round
      const double dx = lt ? ax[x] - bx[y] : bx[y] - ax[x], dy = lt ? ay[x] - by[y] : by[y] - ay[x], rr = lt ? ar + br : br + ar;
      if ((dx * dx + dy * dy) < rr * rr) { have = true; bka = ka; bkb = kb; gbx = gx[x]; gby = gy[y]; }
    }
  }
  return gi == gbx || gj == gby;
}

// Workgroups are dealt round robin over the 8 XCDs (blocks b and b + 8 share one, MI355X_MICROARCH.md), each XCD has its own
// L2, and every launch starts with cold L2s: a floe's columns and ring are fetched once per XCD that touches them.  Kernels in
// which a floe is read on behalf of its neighbours (neighbour search, narrow phase, reduce) therefore give the workgroups of one
// XCD a CONTIGUOUS range of floes -- neighbours in space are neighbours in index -- instead of every eighth group of 16:
// virtual workgroup id such that the ids of one XCD are consecutive (bijective for any grid size; speed only, never correctness).
// nact: the virtual ids that have work (the grid is sized from a capacity, the live count is only known on the device); when the
// grid covers them all, the nact ids are cut into 8 consecutive runs, one per XCD, and a workgroup beyond its XCD's run gets -1
// (nothing to do); otherwise (grid-stride launches) the whole grid is permuted.
__device__ __forceinline__ int xcd_contiguous(int bid, int nblk, int nact) {
  const int x = bid % 8, idx = bid / 8;
  if (nact <= nblk) {
    const int q = nact / 8, r = nact % 8;
    const int cnt = q + (x < r ? 1 : 0), start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return idx < cnt ? start + idx : -1;        // (an XCD always has at least as many workgroups as its run: nact <= nblk)
  }
  const int q = nblk / 8, r = nblk % 8;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
}

// the same for workgroups [first, first + nblk) of a launch (horizontal fusion: the forcing workgroups come after the neighbour
// search's): the XCD label is that of the PHYSICAL workgroup id, the rank counts the workgroups of the range on that XCD
__device__ __forceinline__ int xcd_contiguous_from(int phys, int first, int nblk, int nact) {
  const int x = phys % 8;
  const int f0 = first + ((x - first % 8) + 8) % 8;            // first workgroup of the range on XCD x
  const int idx = (phys - f0) / 8;
  // workgroups of the range per XCD: labels in the range's own order are (x - first) mod 8
  const int lx = ((x - first % 8) + 8) % 8;
  if (nact <= nblk) {
    const int q = nact / 8, r = nact % 8;
    const int cnt = q + (lx < r ? 1 : 0), start = lx < r ? lx * (q + 1) : r * (q + 1) + (lx - r) * q;
    return idx < cnt ? start + idx : -1;
  }
  const int q = nblk / 8, r = nblk % 8;
  return (lx < r ? lx * (q + 1) : r * (q + 1) + (lx - r) * q) + idx;
}

// Work list of the narrow phase: NSEG segments of capPairs / NSEG pair items, filled by the neighbour search (segment =
// workgroup index modulo NSEG, one tail counter per segment a cache line apart: same-address atomics serialise chip-wide).
// The element items of a step (el_floe / el_elem, compact) are dealt out to the segments round robin.  Item t of
// segment s: t < pairs(s): work[s * segcap + t]; else element (t - pairs(s)) * NSEG + s.
struct Seg { int s, np, ne, n; };
__device__ __forceinline__ int seg_cap(const State& S) { return S.capPairs / NSEG; }
__device__ __forceinline__ Seg seg_of(const State& S, int s) {
  Seg g; g.s = s; g.np = S.wq[s * 32 + 1];
  const int nel = S.cnt[C_NELEM];
  g.ne = nel > s ? (nel - s + NSEG - 1) / NSEG : 0;
  g.n = g.np + g.ne;
  return g;
}
// rows: index into it_rows (units of items), info: index into it_info; ao, na / bo, nb: ring offsets and sizes (a pair item
// carries them -- the neighbour search had them at hand -- so that staging an item is one round trip less)
struct Item { int i, j, e, rows, info, ao, na, bo, nb; bool is_pair; };
__device__ __forceinline__ Item item_of(const State& S, const Seg& g, int t) {
  Item it; it.is_pair = t < g.np;
  if (it.is_pair) {
    const int w = g.s * seg_cap(S) + t;
    const int4 k = S.work[2 * (size_t)w], r = S.work[2 * (size_t)w + 1];
    it.info = k.x; it.i = k.y; it.j = k.z; it.e = -1; it.rows = w; it.ao = k.w; it.na = r.x; it.bo = r.y; it.nb = r.z;
  } else {
    const int q = (t - g.np) * NSEG + g.s; it.i = S.el_floe[q]; it.e = S.el_elem[q]; it.j = -1; it.rows = S.capPairs + q; it.info = S.capM * S.maxnb + q;
    it.ao = ring_off(S, it.i); it.na = ring_n(S, it.i); it.bo = S.eoff[it.e]; it.nb = S.eoff[it.e + 1] - it.bo;
  }
  return it;
}

// 16 lanes per floe.  Nine of them fetch one cell of the 3x3 neighbourhood each -- count and bucket in one round trip --
// and pool the floes they find; then all 16 lanes take one candidate each: everything the tests need about it is asked
// for at once (second round trip).  Neighbours that come later in the serial order are the pairs this floe owns
// (outgoing), earlier ones the pairs mirrored onto it (incoming).  Candidates are rank-sorted by order key in LDS.  The
// owned pairs whose ring boxes overlap are appended to the narrow phase's work list here (one tail atomic per workgroup);
// the others get their (empty) result at once.
constexpr int NB_G = 16, NB_TPB = 128;
static_assert(NB_G == FX_WORDS, "the neighbour search clears a new row's fixed-point totals with one store per lane");
// FAM: the Dict rule may read family records (inline ghosts; State::Fam) -- its arrays cost ~50 registers, a wavefront per SIMD that a
// large field (throughput-bound search) misses: the host picks the instantiation by size
// NBC: neighbours a floe may have in either direction (= State::maxnb, the stride of the neighbour lists): 24 for fields of like-sized
// floes, 64 where the host's count at upload finds a size spectrum (Voronoi fields: a large cell has dozens of small neighbours)
// REC: the floe's own row and every candidate come from the collision records (State::crec: five 16-byte loads from ONE line instead of
// thirteen scattered columns -- the kernel is bound by the number of lines its loads touch)
template <int TPB, bool FAM = true, int NBC = MAXNB, bool REC = false>
__device__ __forceinline__ void neighbors_body(State& S, int bid, int nblk) {
  constexpr int GPB = TPB / NB_G;
  constexpr int NB_POOL = NBC <= 24 ? 96 : (NBC <= 64 ? 224 : 256);       // floes the 3 x 3 cells around a floe may hold (chunked pool: per chunk)
  constexpr int MW = (NBC + 63) / 64;                  // 64-bit words of the box mask of the owned pairs
  static_assert(NB_POOL >= NBC, "the pool is reused for the sorted list of the owned pairs");
  __shared__ int cand[GPB][2][NBC];
  __shared__ long long ckey[GPB][2][NBC];
  __shared__ int cnts[GPB][2];
  using mask_t = typename std::conditional<(NBC <= 32), unsigned, unsigned long long>::type;      // box mask of the owned pairs (NBC > 64: MW words)
  __shared__ mask_t wmask[MW > 1 ? 1 : GPB];
  __shared__ unsigned long long wmaskw[MW > 1 ? GPB : 1][MW];
  __shared__ int pool[GPB][NB_POOL];
  __shared__ int pool2[GPB][NBC > MAXNB ? NB_POOL : 1];      // (NBC = 64: the candidates of a chunk that passed the bounding-circle test)
  __shared__ int npool[GPB];
  __shared__ int wbase[GPB + 1];
  __shared__ int cvo[GPB][NBC], cnv[GPB][NBC], svo[GPB][NBC], snv[GPB][NBC];    // ring offset / size of the owned pairs' partners (unsorted, sorted)
  __shared__ double kbox[REC ? GPB : 1][4];      // (records: the floe's own ring box waits here for the AABB cull instead of in eight registers through the Dict rule)
  const int gl = threadIdx.x % NB_G, gi = threadIdx.x / NB_G;
  const StopRegs stop = stop_load(S);
  int M = S.cnt[C_M];
  int nfix = 0x7fffffff;        // rows from here on get their fixed-point totals cleared by this launch (the rows the step's inline makers allocated)
  // Pipelined steps (State::goff != 0): the rows the step's makers allocated lie at [N + goff, ..), a multiple of 16 (the host's choice), not
  // straight behind the parents.  The launch walks blocks of GPB rows; the blocks that hold parents end at nsplit = N rounded up to 16, the
  // blocks from there on are the allocated rows, kshift further on -- a SCALAR decision per block (the search sits on its register budget:
  // a per-lane mapping cost it a wavefront per SIMD), rows [N, nsplit) do not exist.  Without goff: nsplit is never reached.
  int nsplit = 0x7fffffff, kshift = 0, lim0 = M, lim1 = M;      // rows below nsplit exist while < lim0, the others while < lim1 (physical numbers)
  if (S.ginline) {            // inline ghosts: the step's ghosts were made by the kernel that placed their parents; the counts are committed here
    const unsigned long long a = S.galloc[S.gslot * 16], poisoned = S.galloc[S.gslot * 16 + 1];
    const int N = S.cnt[C_N]; int G = (int)(a >> 32), V = (int)(a & 0xffffffffull);
    if (REC && S.facc) nfix = N;          // (the instantiations on collision records -- what the resident steps run; without records the host clears the rows, sz_step)
    if (poisoned || N + S.goff + G > S.capM || (S.gcap > 0 && G > S.gcap)) { G = 0; V = 0; }          // a poisoned allocator (see ghost_inline_make): the parents alone, the error bit is up
    M = N + G; lim0 = M; lim1 = M;
    if (bid == 0 && threadIdx.x == 0 && !stop_test(S, stop)) { S.cnt[C_M] = M; S.cnt[C_NV] = S.voff[N] + V; S.cnt[C_NGHOSTS] = G; }
    if (REC && S.goff != 0) { nsplit = (N + 15) & ~15; kshift = N + S.goff - nsplit; lim0 = N; lim1 = N + S.goff + G; M = nsplit + G; }      // (M: the walk's end, in block numbers)
  }
  const GridGeo g = grid_geo(S);
  const int ncx = g.ncx, ncy = g.ncy;
  const int seg = bid % NSEG, segcap = seg_cap(S);
  // (measured: contiguous ranges per XCD pay in the reduce kernel -- 12.9 -> 11.3 us at 10 k floes, 51 -> 44 at 100 k -- but cost
  //  the neighbour search and, through the uneven segments of the work list, the narrow phase 20 % at 100 k floes: plain order
  //  here unless SZ_XCD=1)
  const int vb0 = S.xcd_neigh ? xcd_contiguous(bid, nblk, (M + GPB - 1) / GPB) : bid;
  // (the floe's own row is asked for before the count of floes is looked at -- rows up to capM exist, one past the end is read and
  //  ignored: the launch's first two round trips, counter block and row, become one)
  static_assert(16 % GPB == 0 || GPB % 16 == 0, "blocks of rows must not straddle nsplit");
  for (int kb = vb0 < 0 ? S.capM : vb0 * GPB; kb < S.capM; kb += nblk * GPB) {
    const bool hi = REC && kb >= nsplit;                        // (uniform)
    const int k = (hi ? kb + kshift : kb) + gi;
    const bool act = k < (hi ? lim1 : lim0);
    double ckx = 0, cky = 0, rk = 0, kx0 = 0, kx1 = 0, ky0 = 0, ky1 = 0;
    long long idk = 0, okk = 0, kgid = 0; bool kplain = true; int ix = 0, iy = 0, vok = 0, nvk = 0, kpar = 0;
    if (k < S.capM) {
      if constexpr (REC) {
        const double2* rr = S.crec + (size_t)k * 8;
        const double2 q0 = rr[0], q1 = rr[1], q2 = rr[2], q3 = rr[3], q4 = rr[4];
        const long long okf = __double_as_longlong(q2.x), vp = __double_as_longlong(q2.y);
        ckx = q0.x; cky = q0.y; rk = q1.x; idk = __double_as_longlong(q1.y);
        okk = okf & CREC_KEYMASK; nvk = (int)(okf >> 48) & 255; vok = (int)(unsigned)vp;
        kpar = (int)(vp >> 32) & 0x0fffffff; kplain = kpar == k && (vp >> 60) == 0;
        if (gl == 0) { kbox[gi][0] = q3.x; kbox[gi][1] = q3.y; kbox[gi][2] = q4.x; kbox[gi][3] = q4.y; }      // (before the barriers below; rewritten only after the loop's last one)
      } else {
        vok = ring_off(S, k); nvk = ring_n(S, k);
        ckx = S.cx[k]; cky = S.cy[k]; rk = S.rmax[k];
        kx0 = S.bbx0[k]; kx1 = S.bbx1[k]; ky0 = S.bby0[k]; ky1 = S.bby1[k];
        idk = S.id[k]; okk = S.okey[k];
        kpar = S.parent[k]; kplain = kpar == k && S.ngh[k] == 0; kgid = S.ghost_id[k];
      }
    }
    loads_issued();
    if (stop_test(S, stop) || kb >= M) break;
    if (bid == 0 && threadIdx.x == 0 && kb == vb0 * GPB) { S.cnt[C_ITEMCLASS] = 0; S.cnt[C_NFUSE] = 0; }     // per-step counters the narrow phase raises
    if constexpr (REC) { if (act && k >= nfix) S.facc[(size_t)k * FX_WORDS + gl] = 0; }         // (NB_G = FX_WORDS lanes: one line)
    __syncthreads();
    if (gl == 0) { cnts[gi][0] = 0; cnts[gi][1] = 0; npool[gi] = 0; if constexpr (MW == 1) wmask[gi] = (mask_t)0; }
    if constexpr (MW > 1) { if (gl < MW) wmaskw[gi][gl] = 0ull; }
    __syncthreads();
    if (act) cell_of(g, ckx, cky, ix, iy);
    bool ovf = false;
    auto candidates = [&](int np) {
      for (int e = gl; e < np; e += NB_G) {
        int o;
        if constexpr (NBC > MAXNB) o = pool2[gi][e]; else o = pool[gi][e];
        if (o == k) continue;      // (the chunked pool holds the floe itself; in the one-pass variant this test never fires, but taking it out
                                   //  moves the register allocation of the FAM instantiation from 168 to 173 -- one wavefront per SIMD less, 17 -> 23 us)
        if (S.rec32) {
          // mixed precision: the candidate's fp32 record first -- two 16-byte loads instead of seven scattered doubles; a
          // candidate that fails the bounding-circle test even with the margin that covers the fp32 roundings is gone, the
          // others are confirmed by the exact fp64 predicate below, so the pair list is the fp64 one bit for bit
          const float4 oc = S.rec32[2 * (size_t)o];
          const float fdx = (float)ckx - oc.x, fdy = (float)cky - oc.y;
          const float frr = (float)rk + oc.z + MIX_CIRCLE_MARGIN + 1e-6f * fmaxf(fabsf((float)ckx), fabsf((float)cky));    // (+ 1 m per 1000 km of coordinate: fp32 spacing)
          if (!(fdx * fdx + fdy * fdy < frr * frr)) continue;
        }
        // everything the tests below may need about o is requested at once
        double ocx, ocy, orm, ox0, ox1, oy0, oy1; long long oid, ko; int opar, voo, nvo; bool oplain;
        if constexpr (REC) {
          const double2* ro = S.crec + (size_t)o * 8;
          const double2 q0 = ro[0], q1 = ro[1], q2 = ro[2], q3 = ro[3], q4 = ro[4];
          const long long okf = __double_as_longlong(q2.x), vp = __double_as_longlong(q2.y);
          ocx = q0.x; ocy = q0.y; orm = q1.x; oid = __double_as_longlong(q1.y);
          ko = okf & CREC_KEYMASK; nvo = (int)(okf >> 48) & 255; voo = (int)(unsigned)vp;
          opar = (int)(vp >> 32) & 0x0fffffff; oplain = opar == o && (vp >> 60) == 0;
          ox0 = q3.x; ox1 = q3.y; oy0 = q4.x; oy1 = q4.y;
        } else {
          ocx = S.cx[o]; ocy = S.cy[o]; orm = S.rmax[o];
          ox0 = S.bbx0[o]; ox1 = S.bbx1[o]; oy0 = S.bby0[o]; oy1 = S.bby1[o];
          oid = S.id[o]; ko = S.okey[o];
          opar = S.parent[o];
          oplain = opar == o && S.ngh[o] == 0;     // a parent without ghosts
          voo = ring_off(S, o); nvo = ring_n(S, o);
        }
        // potential_interaction (collisions.jl:705-710), symmetric in its arguments
        double ddx = ckx - ocx, ddy = cky - ocy, rr = rk + orm;
        if (!((ddx * ddx + ddy * ddy) < rr * rr)) continue;
        if (oid == idk) continue;
        bool after = ko > okk;                   // o comes after k in the serial order
        // the Dict rule only bites when one of the two floes has periodic images
        if (!(kplain && oplain)) {
          bool ok;
          if (FAM && S.famrec) {           // (inline ghosts leave a record per family: a floe that is a ghost, or has ghosts, has one at its parent)
            const int kp = kpar, op = opar;
            // (on records the floe's own ghost number is asked for here, in the rare branch that needs it, instead of being held through the loop)
            const long long gk = REC ? __double_as_longlong(S.crec[(size_t)k * 8 + 7].y) : kgid, go = REC ? __double_as_longlong(S.crec[(size_t)o * 8 + 7].y) : S.ghost_id[o];
            ok = after ? pair_allowed_fam(S, kp, op, !kplain, !oplain, okk, gk, ckx, cky, rk, ko, go, ocx, ocy, orm)
                       : pair_allowed_fam(S, op, kp, !oplain, !kplain, ko, go, ocx, ocy, orm, okk, gk, ckx, cky, rk);
          } else if constexpr (REC) ok = pair_allowed_rec(S, after ? k : o, after ? o : k);
          else ok = pair_allowed(S, after ? k : o, after ? o : k);
          if (!ok) continue;
        }
        int w = after ? 0 : 1;
        int slot = atomicAdd(&cnts[gi][w], 1);
        // AABB cull of the pairs this floe owns: rings whose boxes are disjoint cannot overlap, the item would
        // end at the first test of the clip (sz_geom.hpp clip()) with no row and no flag -- it is not run at all
        int boxes = 1;
        if (after) {
          if constexpr (REC) boxes = !(kbox[gi][1] < ox0 || ox1 < kbox[gi][0] || kbox[gi][3] < oy0 || oy1 < kbox[gi][2]);
          else boxes = !(kx1 < ox0 || ox1 < kx0 || ky1 < oy0 || oy1 < ky0);
        }
        if (slot < NBC) { cand[gi][w][slot] = o | (boxes << 30); ckey[gi][w][slot] = ko; if (after) { cvo[gi][slot] = voo; cnv[gi][slot] = nvo; } } else ovf = true;
      }
    };
    // Two ways to pool the floes of the 3 x 3 cells.  Fields of like-sized floes (NBC = 24): one pass, offsets from an LDS atomic, a pool of 96 --
    // more is an error, and the host's count at upload rules it out (it picks NBC = 64 otherwise).  Fields with a size spectrum (NBC = 64):
    // the chunked pool below, which no crowding overflows; it costs ~15 % of this kernel's time, which is why the first way is kept.
    if constexpr (NBC <= MAXNB) {
      if (act && gl < 9) {
        const int oy = gl / 3 - 1, ox = gl % 3 - 1;
        int cy = iy + oy, cxi = ix + ox;
        bool visit = true;
        // a wrapped direction with fewer than three cells would meet the same cell twice
        if (g.wrapx) { cxi = cell_fold(cxi, ncx, 1); if ((ncx == 1 && ox != 0) || (ncx == 2 && ox > 0)) visit = false; }
        else if (cxi < 0 || cxi >= ncx) visit = false;
        if (g.wrapy) { cy = cell_fold(cy, ncy, 1); if ((ncy == 1 && oy != 0) || (ncy == 2 && oy > 0)) visit = false; }
        else if (cy < 0 || cy >= ncy) visit = false;
        if (visit) {
          const int c = cy * ncx + cxi;
          const int n = S.cell_cnt[c];
          const int4 s0 = *(const int4*)(S.cell_slots + (size_t)c * CELL_K), s1 = *(const int4*)(S.cell_slots + (size_t)c * CELL_K + 4);
          const int sl[CELL_K] = { s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w };
          const int nbk = n < CELL_K ? n : CELL_K;
          int take = 0;
#pragma unroll
          for (int q = 0; q < CELL_K; q++) take += (q < nbk && sl[q] != k) ? 1 : 0;
          int at = take ? atomicAdd(&npool[gi], take) : 0;
#pragma unroll
          for (int q = 0; q < CELL_K; q++) if (q < nbk && sl[q] != k) { if (at < NB_POOL) pool[gi][at] = sl[q]; else ovf = true; at++; }
          if (n > CELL_K) {            // a crowded cell: the floes beyond the bucket are on a chain
            for (int o = S.cell_ovf[c] - 1; o >= 0; o = S.cell_items[o]) {
              if (o == k) continue;
              const int a2 = atomicAdd(&npool[gi], 1);
              if (a2 < NB_POOL) pool[gi][a2] = o; else ovf = true;
            }
          }
        }
      }
      gsync();
      const int np = npool[gi] < NB_POOL ? npool[gi] : NB_POOL;
      candidates(np);
    } else {
      // The floes of the 3 x 3 cells: lanes 0..8 hold one cell each (count, bucket, chain); the cells' floes are numbered cell after cell
      // (prefix of the counts over the lanes), pooled and tested in chunks of NB_POOL -- one chunk unless the cells are crowded (a size
      // spectrum: the cell side follows the LARGEST floe, so a cell holds many small ones; the floe itself is in the pool and skipped).
      int cn = 0, ccell = -1;
      int4 b0 = make_int4(0, 0, 0, 0), b1 = b0;
      if (act && gl < 9) {
        const int oy = gl / 3 - 1, ox = gl % 3 - 1;
        int cy = iy + oy, cxi = ix + ox;
        bool visit = true;
        // a wrapped direction with fewer than three cells would meet the same cell twice
        if (g.wrapx) { cxi = cell_fold(cxi, ncx, 1); if ((ncx == 1 && ox != 0) || (ncx == 2 && ox > 0)) visit = false; }
        else if (cxi < 0 || cxi >= ncx) visit = false;
        if (g.wrapy) { cy = cell_fold(cy, ncy, 1); if ((ncy == 1 && oy != 0) || (ncy == 2 && oy > 0)) visit = false; }
        else if (cy < 0 || cy >= ncy) visit = false;
        if (visit) {
          ccell = cy * ncx + cxi;
          cn = S.cell_cnt[ccell];
          b0 = *(const int4*)(S.cell_slots + (size_t)ccell * CELL_K); b1 = *(const int4*)(S.cell_slots + (size_t)ccell * CELL_K + 4);     // (count and bucket in one round trip)
        }
      }
      int coff = cn;
#pragma unroll
      for (int d = 1; d < NB_G; d <<= 1) { const int t = __shfl_up(coff, d, NB_G); if (gl >= d) coff += t; }
      const int ntot = __shfl(coff, NB_G - 1, NB_G);
      coff -= cn;
      if (cn > 0 && coff < NB_POOL) {          // the buckets' part of the first chunk, from the registers
        const int sl[CELL_K] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
        for (int q = 0; q < CELL_K; q++) { const int idx = coff + q; if (q < cn && idx < NB_POOL) pool[gi][idx] = sl[q]; }
      }
      for (int cb = 0; cb < ntot; cb += NB_POOL) {
        if (cn > 0 && coff < cb + NB_POOL && coff + cn > cb) {
          if (cb > 0) {          // (later chunks ask for the bucket again: holding it across the chunks would cost the kernel a wavefront per SIMD)
            const int4 s0 = *(const int4*)(S.cell_slots + (size_t)ccell * CELL_K), s1 = *(const int4*)(S.cell_slots + (size_t)ccell * CELL_K + 4);
            const int sl[CELL_K] = { s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w };
#pragma unroll
            for (int q = 0; q < CELL_K; q++) { const int idx = coff + q - cb; if (q < cn && idx >= 0 && idx < NB_POOL) pool[gi][idx] = sl[q]; }
          }
          if (cn > CELL_K) {            // a crowded cell: the floes beyond the bucket are on a chain
            int idx = coff + CELL_K - cb;
            for (int o = S.cell_ovf[ccell] - 1; o >= 0 && idx < NB_POOL; o = S.cell_items[o], idx++) if (idx >= 0) pool[gi][idx] = o;
            for (const int end = coff + cn - cb < NB_POOL ? coff + cn - cb : NB_POOL; idx < end; idx++) if (idx >= 0) pool[gi][idx] = k;      // (never: count and chain agree)
          }
        }
        gsync();
        const int npc = ntot - cb < NB_POOL ? ntot - cb : NB_POOL;
        // A window of such a field holds a hundred floes and more, a dozen of which are neighbours: the bounding-circle test runs first, on
        // three columns, and only the floes that pass it get the full set of loads (the one-pass variant asks for everything at once:
        // most of ITS candidates pass).  potential_interaction (collisions.jl:705-710), the same expression as below.
        if (gl == 0) npool[gi] = 0;
        gsync();
        for (int e = gl; e < npc; e += NB_G) {
          const int o = pool[gi][e];
          if (o == k) continue;
          const double ocx = S.cx[o], ocy = S.cy[o], orm = S.rmax[o];
          const double ddx = ckx - ocx, ddy = cky - ocy, rr = rk + orm;
          if ((ddx * ddx + ddy * ddy) < rr * rr) pool2[gi][atomicAdd(&npool[gi], 1)] = o;
        }
        gsync();
        const int np = npool[gi];
        candidates(np);
        gsync();          // (the next chunk overwrites the pools)
      }
    }
    if (ovf) { atomicOr(&S.cnt[C_ERR], ERR_CAP_NEIGH); }
    gsync();
    for (int w = 0; w < 2; w++) {
      int n = cnts[gi][w] < NBC ? cnts[gi][w] : NBC;
      int* dst = (w == 0 ? S.nb_out : S.nb_in) + (size_t)k * NBC;
      for (int e = gl; e < n; e += NB_G) {
        long long ke = ckey[gi][w][e]; int r = 0;
        for (int f = 0; f < n; f++) r += ckey[gi][w][f] < ke;
        const int cv = cand[gi][w][e];
        dst[r] = cv & 0x3fffffff;
        if (w == 0) { pool[gi][r] = cv & 0x3fffffff; svo[gi][r] = cvo[gi][e]; snv[gi][r] = cnv[gi][e]; }   // (the pool is free by now: the sorted owned list, for the work items below)
        if (w == 0 && (cv >> 30)) { if constexpr (MW == 1) atomicOr(&wmask[gi], (mask_t)1 << r); else atomicOr(&wmaskw[gi][r >> 6], 1ull << (r & 63)); }
      }
      gsync();
      if (gl == 0 && act) {
        if (w == 0) S.n_out[k] = n;
        else S.n_in[k] = n;
      }
    }
    // ---- this workgroup's share of the work list: one tail atomic for all its floes
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int q = 0; q < GPB; q++) {
        wbase[q] = tot;
        if constexpr (MW == 1) tot += __popcll((unsigned long long)wmask[q]); else for (int w8 = 0; w8 < MW; w8++) tot += __popcll(wmaskw[q][w8]);
      }
      int base = tot ? atomicAdd(&S.wq[seg * 32 + 1], tot) : 0;
      if (base + tot > segcap) { atomicOr(&S.cnt[C_ERR], ERR_CAP_PAIRS); base = -1; }
      wbase[GPB] = base;
    }
    __syncthreads();
    if (act) {
      const int base = wbase[GPB], nk = cnts[gi][0] < NBC ? cnts[gi][0] : NBC;
      mask_t mask = (mask_t)0;
      if constexpr (MW == 1) mask = wmask[gi];
      for (int r = gl; r < nk; r += NB_G) {
        const int slot = k * NBC + r;
        bool on; int before;
        if constexpr (MW == 1) { on = (mask >> r & 1) != 0; before = __popcll((unsigned long long)(mask & (((mask_t)1 << r) - (mask_t)1))); }
        else {
          const unsigned long long mword = wmaskw[gi][r >> 6];
          on = (mword >> (r & 63) & 1) != 0; before = __popcll(mword & ((1ull << (r & 63)) - 1ull));
          for (int w8 = 0; w8 < (r >> 6); w8++) before += __popcll(wmaskw[gi][w8]);
        }
        if (on && base >= 0) {
          const int j = pool[gi][r];
          const size_t w2 = 2 * ((size_t)seg * segcap + base + wbase[gi] + before);
          S.work[w2] = make_int4(slot, k, j, vok); S.work[w2 + 1] = make_int4(nvk, svo[gi][r], snv[gi][r], 0);
        } else S.it_info[slot] = make_int2(0, -1);      // boxes disjoint: no region, no row, no flag (the clip's own first test)
      }
    }
  }
}
// NBC = 256 (a floe with more than 64 neighbours in one direction: a large floe among many small ones): a quarter of the lane groups per
// workgroup (48 KB of LDS for four floes) -- the capacity that keeps such a field running, not a fast path
template <int NBC> constexpr int nb_tpb() { return NBC > 64 ? 64 : NB_TPB; }
#ifndef SZ_NB_LEAN_WPE
#define SZ_NB_LEAN_WPE 1      // (measured: the lean instantiation on records compiled for five wavefronts per SIMD -- 96 registers, 32 B of scratch -- is slower,
                              //  79.2 -> 84.8 us at 100 k floes, even at 40 k: profiles/r03_runs/r3_h_*)
#endif
template <bool FAM, int NBC = MAXNB, bool REC = false>
__global__ void __launch_bounds__(nb_tpb<NBC>(), (NBC > 64 ? 1 : NBC > MAXNB ? (FAM ? 3 : 4) : (REC && !FAM ? SZ_NB_LEAN_WPE : 1))) sz_k_neighbors(State S) { neighbors_body<nb_tpb<NBC>(), FAM, NBC, REC>(S, blockIdx.x, gridDim.x); }

// The compact pair list in the reference's serial order (i asc, j asc) -- out_off, pair_i, pair_j -- is only made when
// the host asks for it (sz_download_pairs): fill after a scan of n_out.
__global__ void sz_k_pairs_fill(State S, int M) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
    const int o = S.out_off[i], nk = S.n_out[i];
    if (o + nk > S.capPairs) { atomicOr(&S.cnt[C_ERR], ERR_CAP_PAIRS); continue; }
    for (int t = 0; t < nk; t++) { S.pair_i[o + t] = i; S.pair_j[o + t] = S.nb_out[(size_t)i * S.maxnb + t]; }
  }
}
// explicit pair list (sz_collide_pairs; sorted by (i, j) on the host): out lists from the given pairs, no incoming
// lists, every pair is run as given (floe_floe_interaction! has no broad phase); item p goes to segment p % NSEG
__global__ void sz_k_pairs_explicit(State S, int np) {
  int M = S.cnt[C_M];
  const int segcap = seg_cap(S);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < M; k += gridDim.x * blockDim.x) {
    int lo = 0, hi = np;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (S.pair_i[mid] < k) lo = mid + 1; else hi = mid; }
    int e = lo; while (e < np && S.pair_i[e] == k) e++;
    int nk = e - lo;
    if (nk > S.maxnb) { atomicOr(&S.cnt[C_ERR], ERR_CAP_NEIGH); nk = S.maxnb; }
    S.n_out[k] = nk; S.n_in[k] = 0;
    for (int r = 0; r < nk; r++) {
      const int p = lo + r, j = S.pair_j[p];
      S.nb_out[(size_t)k * S.maxnb + r] = j;
      if (p / NSEG < segcap) {
        const size_t w2 = 2 * ((size_t)(p % NSEG) * segcap + p / NSEG);
        S.work[w2] = make_int4(k * S.maxnb + r, k, j, S.voff[k]); S.work[w2 + 1] = make_int4(S.voff[k + 1] - S.voff[k], S.voff[j], S.voff[j + 1] - S.voff[j], 0);
      } else atomicOr(&S.cnt[C_ERR], ERR_CAP_PAIRS);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < NSEG) {
    const int s = threadIdx.x;
    S.wq[s * 32] = 0; S.wq[s * 32 + 1] = np > s ? (np - s + NSEG - 1) / NSEG : 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { S.cnt[C_ITEMCLASS] = 0; S.cnt[C_NFUSE] = 0; }
}
// ============================================================================ domain element items (A10 prefilter)
// REC: centroid and rmax from the floe's collision record (pipelined steps: the columns are being brought up to date beside this scan)
template <bool REC = false, typename F>
__device__ __forceinline__ void elem_candidates(const State& S, int k, F&& emit) {
  double cx, cy, r;
  if constexpr (REC) { const double2 q0 = S.crec[(size_t)k * 8], q1 = S.crec[(size_t)k * 8 + 1]; cx = q0.x; cy = q0.y; r = q1.x; }
  else { cx = S.cx[k]; cy = S.cy[k]; r = S.rmax[k]; }
  if (cy + r > S.eval[0] && S.ekind[0] != 1) emit(0);
  if (cy - r < S.eval[1] && S.ekind[1] != 1) emit(1);
  if (cx + r > S.eval[2] && S.ekind[2] != 1) emit(2);
  if (cx - r < S.eval[3] && S.ekind[3] != 1) emit(3);
  for (int e = 4; e < S.nelem; e++) {
    double dx = S.ecx[e] - cx, dy = S.ecy[e] - cy, rr = S.ermax[e] + r;
    if (dx * dx + dy * dy < rr * rr) emit(e);
  }
}
// count, scan (look-back) and fill of the floe-element items in one launch; tile: this workgroup's number among the scan's workgroups
// mh >= 0 (pipelined steps of fields between walls: no ghosts, the host knows the count): the number of floes
template <bool REC = false>
__device__ __forceinline__ void elem_scan_fill_body(State& S, unsigned epoch, int tile, int mh = -1) {
  __shared__ int4 tot;
  if (stopped(S)) return;
  const int M = mh >= 0 ? mh : S.cnt[C_M];
  const int base = tile * (int)blockDim.x;
  if (base >= M && tile != 0) return;
  const int k = base + threadIdx.x;
  int c = 0;
  if (k < M) elem_candidates<REC>(S, k, [&](int) { c++; });
  const int4 ex = block_exclusive_scan4(make_int4(c, 0, 0, 0), &tot);
  const int4 before = lookback_prefix4(S, tot, epoch, tile);
  if (M == 0) { if (k == 0) { S.el_off[0] = 0; S.cnt[C_NELEM] = 0; } return; }
  if (k >= M) return;
  int o = ex.x + before.x;
  S.el_off[k] = o;
  if (k == M - 1) {
    int t = o + c; S.el_off[M] = t;
    if (t > S.capElem) { atomicOr(&S.cnt[C_ERR], ERR_CAP_ELEM); t = 0; }
    S.cnt[C_NELEM] = t;
  }
  if (o + c > S.capElem) return;
  elem_candidates<REC>(S, k, [&](int e) { S.el_floe[o] = k; S.el_elem[o] = e; o++; });
}
__global__ void __launch_bounds__(SCAN_B) sz_k_elem_scan_fill(State S, unsigned epoch) { elem_scan_fill_body(S, epoch, (int)blockIdx.x); }
// The element items ride in the tail of the neighbour search's launch (fields between walls: nothing in that launch changes the floe
// count the scan reads -- with periodic walls its first workgroup commits the step's ghosts): the scan's workgroups only wait for
// one another, they are handed out in index order behind the search's, and its own launch (7 us + a launch boundary) is gone.
template <bool FAM, bool REC = false>
__global__ void __launch_bounds__(NB_TPB) sz_k_neighbors_elem(State S, unsigned epoch, int nbn) {
  if ((int)blockIdx.x < nbn) neighbors_body<NB_TPB, FAM, MAXNB, REC>(S, blockIdx.x, nbn);
  else elem_scan_fill_body(S, epoch, (int)blockIdx.x - nbn);
}

// ============================================================================ narrow phase (A4-A10)
// items [0, P): floe-floe pairs; [P, P+Q): floe-element items.  SMALL kernels take the items
// whose rings both fit LO..CAP points.
// WPE: wavefronts per SIMD the kernel is compiled for (register budget 512 / WPE)
// FRC / nbf: the step's forcings ride in this launch -- the last nbf workgroups evaluate them (FRC 1: fp64, 2: mixed precision).
// Workgroups are handed out in index order: the narrow ones fill the chip first, the forcing ones move in as those finish.  The
// narrow phase of a small field is ONE round whose length is set by its slowest wavefront (~47 us at 10 k floes, most
// wavefronts are done after 20): the forcings (22 us as a launch of their own) run in that tail instead of beside the neighbour search.
template <bool TW>
__device__ __forceinline__ void forcing_body(State& S, const Params& P, int bid, int nblk, int pmax, int first);
__device__ __forceinline__ void forcing_mixed_body(State& S, const Params& P, int bid, int nblk, int first);
// GEO / PA / nbg / nn (pipelined steps, sz_pipeline.hpp): nbg more workgroups, straight behind the narrow ones, make the geometry of the NEXT step
// (thread per parent, nn parents) into the other parity's buffers
__device__ __forceinline__ void geo_body(State S, const PipeAlt& A, int dt, int bid, int nblk, int N);
template <int G, int CAP, int KC, int RC, int RM, int TPB, int LO, int CLS, int WPE = 1, int FRC = 0, int GEO = 0>
__global__ void __launch_bounds__(TPB, WPE) sz_k_narrow(State S, Params P, int dt, double ff_max_overlap, double fd_max_overlap, int dbg, int queue, int nbf,
                                                       PipeAlt PA, int nbg, int nn) {
  constexpr int GPB = TPB / G;
  // nbf < 0: -nbf forcing workgroups IN FRONT of the narrow ones (a few persistent wavefronts per CU that walk all floes, in the wave slots the
  // narrow workgroups -- which fill the LDS -- leave free) instead of behind them
  const int nf = nbf < 0 ? -nbf : nbf;
  const int nblk = (int)gridDim.x - (FRC != 0 ? nf : 0) - (GEO != 0 ? nbg : 0);       // the narrow workgroups
  // (the GEO workgroups come FIRST: a few wavefronts per CU that fit beside the narrow ones -- those fill the LDS, not the wave slots -- and
  //  are done long before the narrow round is; behind the narrow workgroups they would only start when those finish)
  if (GEO != 0 && (int)blockIdx.x < nbg) { geo_body(S, PA, dt, (int)blockIdx.x, nbg, nn); return; }
  int bidx = (int)blockIdx.x - (GEO != 0 ? nbg : 0);        // this workgroup's number among the narrow ones
  if (FRC != 0) {
    int fb = -1, first = 0;
    if (nbf < 0) { if (bidx < nf) { fb = bidx; first = GEO != 0 ? nbg : 0; } else bidx -= nf; }
    else if (bidx >= nblk) { fb = bidx - nblk; first = nblk + (GEO != 0 ? nbg : 0); }
    if (fb >= 0) {
      if (FRC == 1) forcing_body<false>(S, P, fb, nf, 0, first);
      else forcing_mixed_body(S, P, fb, nf, first);
      return;
    }
  }
  static_assert(4 * KC <= 2 * RC, "raw crossing slots alias reg[1]");
  Stamps st; STAMP_INIT(st);
#ifdef SZ_STAMPS
  st.on = (CLS == 0 && bidx == (dbg >> 8) && threadIdx.x == 0); st.log = S.stamps + 1;
#endif
  const int qk = bidx % NSEG;
  // the first round's item of this lane group is known without the segment's length: its work-list entry is asked for
  // together with the length (one dependent round trip less in the launch's chain; an entry past the end is read and ignored)
  const int t_first = (bidx / NSEG) * (TPB / G) + (int)(threadIdx.x / G);
  const bool pre_ok = !(G == 64 && TPB == 64) && t_first < seg_cap(S);
  int4 pre0 = make_int4(0, 0, 0, 0), pre1 = make_int4(0, 0, 0, 0);
  if (pre_ok) { const size_t w2 = 2 * ((size_t)qk * seg_cap(S) + t_first); pre0 = S.work[w2]; pre1 = S.work[w2 + 1]; }
  // ... and so are the segment's lengths and the stop counters: one round trip, then the tests
  const StopRegs stop = stop_load(S);
  const Seg sg = seg_of(S, qk);
  const int icls = CLS > 0 ? S.cnt[C_ITEMCLASS] : 0;
  loads_issued();
  if (CLS > 0 && icls < CLS) return;   // no item needs this (larger) variant this step
  if (stop_test(S, stop)) return;
  STAMP(st, 20);
  __shared__ GroupMem<CAP, KC, RC, RM> mem[GPB];
  const int gl = threadIdx.x % G, gi = threadIdx.x / G;
  GroupMem<CAP, KC, RC, RM>& m = mem[gi];
  // Items: segment `qk` of the work list (pair items appended by the neighbour search + this segment's share of the element
  // items).  The first round is static (lane group `gi` of the r-th workgroup of the segment takes item r * GPB + gi); the
  // rounds after it are handed out dynamically from the segment's queue head: a round of 8 items takes anything between a
  // few thousand cycles (no overlap) and 150 k, so with a static split the slowest workgroup sets the time of a deep launch
  // (narrow kernel 374 -> 242 us at 100 k floes; one head for the whole chip costs ~20 ns per ticket, serialised across the
  // XCDs: measured slower than the static split).  The results do not depend on who runs an item.
  const int nitems = sg.n;
  if (gl == 0) { m.err = 0; m.ierr = 0; m.ntracefail = 0; m.nchk = 0; m.nkeep = 0; m.acc[0] = m.acc[1] = 0; m.acc16[0] = m.acc16[1] = m.acc16[2] = m.acc16[3] = m.acc16[4] = 0; }   // acc: work done by this group (in LDS: no register held across the rounds)
  if (CLS == 0) {
    // housekeeping of the step (the neighbour search has consumed the cells; the integrator fills them again): cell counts
    // and overflow heads cleared, the guard counters of the coming update reset
    const int ncells = (int)S.bounds[4] * (int)S.bounds[5];
    for (int q = bidx * TPB + threadIdx.x; q <= ncells; q += nblk * TPB) { S.cell_cnt[q] = 0; S.cell_ovf[q] = 0; }
    if (bidx == 0) for (int q = threadIdx.x; q < WARN_SLOTS * 4; q += TPB) S.warn[(q >> 2) * 32 + (q & 3)] = 0;
    if (S.ginline && !GEO && bidx == 0 && threadIdx.x == 0) { S.galloc[(1 - S.gslot) * 16] = 0ull; S.galloc[(1 - S.gslot) * 16 + 1] = 0ull; }      // the allocator (and its poison mark) this step's integrator makes the next ghosts in (pipelined steps: cleared by the update of the step before -- GEO draws from it in this very launch)
    // a list the neighbour search outgrew (its error bits): the batch pauses in this step -- raised here, where the counter block is
    // at hand anyway, rather than in the search (which sits exactly on its register budget)
    if (bidx == 0 && threadIdx.x == 0 && (S.cnt[C_ERR] & (ERR_CAP_NEIGH | ERR_CAP_PAIRS))) capacity_stop(S);
  }
  STAMP(st, 21);
  // The one-item-per-wavefront variant mostly looks for the few items meant for it: its lanes test 64
  // items at a time and the wavefront then works the flagged ones off one by one.
  constexpr bool SCAN = (G == 64 && TPB == 64);
  constexpr int STRIDE = SCAN ? 64 : GPB;
  const bool useq = CLS == 0 && !SCAN && queue != 0;       // (the larger variants look at every item of their segment: static rounds)
  const int rb = bidx / NSEG, nbq = (nblk + NSEG - 1 - qk) / NSEG;    // this workgroup's rank in its segment, workgroups per segment
  const int limit = nitems;
  // (measured and dropped: spreading the only round of a small field over ALL resident workgroups -- 6 to 7 items per wavefront
  //  instead of 8 -- made the launch 18 % SLOWER, 47 -> 56 us at 10 k floes: a SIMD issues the instruction streams of its
  //  wavefronts one after the other, and a wavefront's stream is as long with 6 items as with 8)
  for (int t0 = rb * STRIDE; t0 < limit;) {
   unsigned long long todo = 1;
   int tk_next = 0;
   if (SCAN) {
     const int tt = t0 + (int)threadIdx.x;
     bool want = false;
     if (tt < nitems) {
       const Item it_ = item_of(S, sg, tt);
       const int nb_ = it_.nb, na_ = it_.na;
       want = (na_ > nb_ ? na_ : nb_) > LO || ((S.it_info[it_.info].x >> 8) & IT_RETRY);
     }
     todo = __ballot(want);
   }
   while (todo) {
    // ================= phase A: every lane group stages its own item and runs the contact clip
#ifdef SZ_STAMPS
    st.tmark = clock64();
    if (st.cP < 0) st.cP = st.tmark - st.t0w;          // (the wavefront's prologue: launch to its first item)
#endif
    int t;
    if (SCAN) { t = t0 + __ffsll((long long)todo) - 1; todo &= todo - 1; }
    else { todo = 0; t = t0 + gi; }
    bool have = t < limit;
    Item it; it.i = 0; it.j = -1; it.e = -1; it.rows = 0; it.info = 0; it.is_pair = true; it.ao = it.na = it.bo = it.nb = 0;
    if (have) {
      if (pre_ok && t == t_first && t < sg.np) {
        it.is_pair = true; it.info = pre0.x; it.i = pre0.y; it.j = pre0.z; it.e = -1; it.rows = qk * seg_cap(S) + t;
        it.ao = pre0.w; it.na = pre1.x; it.bo = pre1.y; it.nb = pre1.z;
      } else it = item_of(S, sg, t);
    }
    STAMP(st, 22);
    const bool is_pair = it.is_pair;
    const int i = it.i, j = it.j, e = it.e;
    // What phase C needs of the item -- its it_info index and the rows of its two floes (fixed-point totals) -- rides through the clips in ONE
    // register, a different word in each of the group's first three lanes (every lane of a group holds the same item, so the lanes can share the
    // work of remembering it).  Round 4, second half: phase C used to ask the work list for the two rows again (a hit in L2 -- but a vector LOAD
    // late in a wavefront's life, behind the forcing wavefronts that have moved into the CU by then and keep its texture path busy with scattered
    // lattice reads: the slowest wavefronts of a launch -- which set its length -- spent 52 k of their 180 k cycles in phase C, 4 k before the
    // fixed-point totals: profiles/r04_wave_records*.txt).  Lane shuffles do not go through the texture path.
    int carry = (gl == 0 ? it.info : gl == 1 ? i : j);
    asm volatile("" : "+v"(carry));          // (opaque: the three words are not kept alive on their own beside it)
    const int na = it.na, nb = it.nb, ao = it.ao, bo = it.bo;
    if (have) {
      const int big = na > nb ? na : nb;
      // an item belongs to the first variant whose ring capacity fits it; an item a smaller variant
      // gave up on (more crossings / region points / regions than its working set holds) is handed to
      // the largest one through IT_RETRY
      const bool retry = CLS == 2 && ((S.it_info[it.info].x >> 8) & IT_RETRY);
      if (big <= LO && !retry) have = false;
      else if (big > CAP) { if (CLS == 2 && gl == 0) atomicOr(&S.cnt[C_ERR], ERR_CAP_RING); have = false; }
    }
    STAMP(st, 23);
    ItemCtx ic;
    ic.E = P.E; ic.nu = P.nu; ic.mu = P.mu; ic.dt = dt; ic.dbg = dbg;
    ic.mode = ITEM_PAIR; ic.max_overlap = ff_max_overlap; ic.elem_dir = -1; ic.elem_val = 0.0; ic.rigid_j = 0;
    int flags = 0;
    int pna = 0, pnb = 0, poa = 1, pob = 1; Box pba{ 0, 0, 0, 0 }, pbb{ 0, 0, 0, 0 };      // operands of the pass's clip
    gsync();
    if (have) {
      // Staging: every load of the item -- rings, scalars, signs, boxes -- is asked for BEFORE the first LDS store (a loop of
      // load -> store pairs waits for each load in turn: seven dependent round trips of ~2.5 k cycles instead of one).
      constexpr int NIT = (CAP + G - 1) / G, NKI = (14 + G - 1) / G;
      double rax[NIT], ray[NIT], rbx[NIT], rby[NIT], kv[NKI];
      const int ekind = is_pair ? 0 : S.ekind[e];
      // The item's scalars, ring signs and ring boxes reach the lanes through LDS: every lane asks for a few values, stores them into the
      // item's memory (kin, box, roa / rob) and all lanes read what they need after the group barrier -- four doubles per lane are held
      // across the wait instead of two boxes and two signs in every lane.
      //  * a pair item on collision records (State::crec): two 16-byte loads per lane from the two floes' record lines instead of
      //    loads from 22 scattered column lines -- lanes 0..3 take quads {0, 5, 6, 7} of floe i and then the box quads of i and j,
      //    lanes 4..7 the same quads of floe j and then the two sign quads;
      //  * otherwise the columns: scalar q = lane (+ G), box value = lane (< 8), sign = lane (< 2).
      const bool urec = G == 8 && S.crec != nullptr && is_pair;
      constexpr int NBV = (8 + G - 1) / G;          // box values a lane asks for on the column path (one for lane groups of 8 and more)
      double bv0 = 0.0, bv1 = 0.0, bvx[NBV > 1 ? NBV : 1]; int osv = 1;
      for (int r = 0; r < (NBV > 1 ? NBV : 1); r++) bvx[r] = 0.0;
      // (which column / quad a lane asks for depends on the lane alone: the compiler would compute those pointers once, before the loop over
      //  the items, hold them in registers through the whole kernel and -- at this kernel's budget -- spill them; the lane number is
      //  therefore made opaque here, a handful of selects per item instead of scratch reloads)
      int glv = gl; asm volatile("" : "+v"(glv));
#pragma unroll
      for (int r = 0; r < NKI; r++) kv[r] = 0.0;
      if (urec) {
        const double2* ri = S.crec + (size_t)i * 8; const double2* rj = S.crec + (size_t)j * 8;
        const int qsel = (glv & 3) == 0 ? 0 : 4 + (glv & 3);
        const double2 qa = (glv < 4 ? ri : rj)[qsel];
        const double2 qb = glv < 2 ? ri[3 + glv] : glv < 4 ? rj[1 + glv] : glv == 4 ? ri[2] : glv == 5 ? rj[2] : glv == 6 ? ri[1] : rj[1];      // (6, 7: {rmax, id} -- the scale of the fixed-point torque / stress sums)
        kv[0] = qa.x; kv[NKI - 1] = qa.y; bv0 = qb.x; bv1 = qb.y;
      } else {
#pragma unroll
        for (int r = 0; r < NKI; r++) {             // the item's scalars, one lane each
          const int q = glv + r * G;
          const int c = q < 10 ? q % 5 : 5 + (q & 1);                  // cx cy u v xi | area height
          const bool side_j = q < 10 ? q >= 5 : q >= 12;
          const double* col = c == 0 ? S.cx : c == 1 ? S.cy : c == 2 ? S.u : c == 3 ? S.v : c == 4 ? S.xi : c == 5 ? S.area : S.height;
          double val = 0.0;
          if (q < 14) {
            if (!side_j) val = col[i];
            else if (is_pair) val = col[j];
            else val = (ekind == 3 && c == 2) ? S.eu[e] : (ekind == 3 && c == 3) ? S.ev[e] : 0.0;   // a wall / topography: rigid (u, v)
          }
          kv[r] = val;
        }
#pragma unroll
        for (int r = 0; r < NBV; r++) {
          const int qb = glv + r * G;
          if (qb < 8) {
            const int w = qb & 3;
            const double* bc = w == 0 ? S.bbx0 : w == 1 ? S.bbx1 : w == 2 ? S.bby0 : S.bby1;
            const double bvv = qb < 4 ? bc[i] : is_pair ? bc[j] : S.ebb[4 * e + w];
            if (NBV > 1) bvx[r] = bvv; else bv0 = bvv;
          }
        }
        if (glv < 2) osv = glv == 0 ? S.osign[i] : is_pair ? S.osign[j] : S.eosign[e];
        if (S.facc && glv >= 2 && glv < 4) bv1 = glv == 2 ? S.rmax[i] : is_pair ? S.rmax[j] : 1.0;      // (the scale of the fixed-point torque / stress sums)
      }
      if (!is_pair) {
        ic.mode = ekind == 0 ? ITEM_OPEN : ITEM_SOLID; ic.max_overlap = fd_max_overlap;
        ic.elem_dir = e < 4 ? e : -1; ic.elem_val = S.eval[e]; ic.rigid_j = 1;
      }
      if (S.body_rings) {
        // mixed precision: rings live once, in the body frame, as fp32; the world coordinates the predicates work on are
        // rebuilt here in fp64 from the fp64 pose -- the same expression the integrator's box and sz_k_world_rings use
        const double cxi = S.cx[i], cyi = S.cy[i], cai = S.trig[2 * i], sai = S.trig[2 * i + 1];
        const int jj = is_pair ? j : i;
        const double cxj = S.cx[jj], cyj = S.cy[jj], caj = S.trig[2 * jj], saj = S.trig[2 * jj + 1];
#pragma unroll
        for (int r = 0; r < NIT; r++) {
          const int q = gl + r * G;
          const float2 pa = q < na ? S.ring32[ao + q] : make_float2(0.f, 0.f);
          rax[r] = (cai * (double)pa.x - sai * (double)pa.y) + cxi; ray[r] = (sai * (double)pa.x + cai * (double)pa.y) + cyi;
          if (is_pair) {
            const float2 pb = q < nb ? S.ring32[bo + q] : make_float2(0.f, 0.f);
            rbx[r] = (caj * (double)pb.x - saj * (double)pb.y) + cxj; rby[r] = (saj * (double)pb.x + caj * (double)pb.y) + cyj;
          } else { rbx[r] = q < nb ? S.ex[bo + q] : 0.0; rby[r] = q < nb ? S.ey[bo + q] : 0.0; }
        }
      } else {
#pragma unroll
        for (int r = 0; r < NIT; r++) {
          const int q = gl + r * G;
          const double2 pa = q < na ? S.vxy[ao + q] : make_double2(0.0, 0.0);
          rax[r] = pa.x; ray[r] = pa.y;
          if (is_pair) { const double2 pb = q < nb ? S.vxy[bo + q] : make_double2(0.0, 0.0); rbx[r] = pb.x; rby[r] = pb.y; }
          else { rbx[r] = q < nb ? S.ex[bo + q] : 0.0; rby[r] = q < nb ? S.ey[bo + q] : 0.0; }
        }
      }
#pragma unroll
      for (int r = 0; r < NIT; r++) {
        const int q = gl + r * G;
        if (q < na) { m.ax[q] = rax[r]; m.ay[q] = ray[r]; }
        if (q < nb) { m.bx[q] = rbx[r]; m.by[q] = rby[r]; }
      }
      if (urec) {
        // kin: i: cx cy u v xi area h, j: the same -- the quads as they come;  box: a x0 x1 y0 y1, b x0 x1 y0 y1
        const int w = gl & 3, at = (gl < 4 ? 0 : 7) + 2 * w;
        m.kin[at] = kv[0];
        if (w != 3) m.kin[at + 1] = kv[NKI - 1];
        if (gl < 4) { m.box[2 * gl] = bv0; m.box[2 * gl + 1] = bv1; }
        else if (gl < 6) (&m.roa)[gl - 4] = (__double_as_longlong(bv0) >> 56) & 1 ? -1 : 1;
        else (&m.eri)[gl - 6] = (int8_t)fx_lever_exp(bv0);
      } else {
#pragma unroll
        for (int r = 0; r < NKI; r++) {
          const int q = gl + r * G;
          if (q < 14) m.kin[q < 10 ? (q >= 5 ? 7 : 0) + q % 5 : (q >= 12 ? 7 : 0) + 5 + (q & 1)] = kv[r];
        }
        if (NBV > 1) {
#pragma unroll
          for (int r = 0; r < NBV; r++) { const int qb = gl + r * G; if (qb < 8) m.box[qb] = bvx[r]; }
        } else if (gl < 8) m.box[gl] = bv0;
        if (gl < 2) (&m.roa)[gl] = (int8_t)osv;
        if (S.facc && gl >= 2 && gl < 4) (&m.eri)[gl - 2] = (int8_t)fx_lever_exp(bv1);
      }
      gsync();
      const int oa = m.roa, ob = m.rob;
      const Box ba{ m.box[0], m.box[1], m.box[2], m.box[3] }, bb{ m.box[4], m.box[5], m.box[6], m.box[7] };
      STAMP(st, 0);
#ifdef SZ_STAMPS
      { long long now = clock64(); st.cA1 += now - st.tmark; }
#endif
      pna = na; pnb = nb; poa = oa; pob = ob; pba = ba; pbb = bb;
      contact_pre(m, gl, na, oa, nb, ob, ba, bb);
    } else if (gl == 0) { m.nkeep = 0; m.nchk = 0; m.ierr = 0; m.lists_ok = 0; }
    // ================= phases A (pass 0: the contact clip of the own item) and B (passes 1..: the direction checks of ALL
    // items of the wavefront, one per lane group and pass -- an item with two or three contact regions no longer works its
    // checks off one after the other while the other groups idle: 14 % of the wavefronts hold such an item, and they used
    // to set the kernel's duration).  One loop, ONE call site of clip().
    {
      int total = 0;
#ifdef SZ_STAMPS
      long long tA0 = clock64(); (void)tA0;
#endif
      for (int pass = 0; pass == 0 || (pass - 1) * GPB < total; pass++) {
        bool run = false, cert = false; int g = gi, q = 0, buf = 0; double ox = 0.0, oy = 0.0;
        if (pass == 0) run = have && !(dbg & 4);
        else {
          const int idx = (pass - 1) * GPB + gi;
          if (idx < total) {
            int pre = 0; g = 0;
            while (pre + mem[g].nchk <= idx) { pre += mem[g].nchk; g++; }
            q = (int)mem[g].chk[idx - pre];
            const auto& o = mem[g];
            cert = o.chkc[idx - pre] != 0;
            run = true; buf = 1; ox = o.dxv[q]; oy = o.dyv[q];
            pna = o.rna; pnb = o.rnb; poa = o.roa; pob = o.rob;
            pba = Box{ o.box[0], o.box[1], o.box[2], o.box[3] }; pbb = Box{ o.box[4], o.box[5], o.box[6], o.box[7] };
          }
        }
#ifdef SZ_STAMPS
        long long tc0 = clock64();
#endif
        // A check of a lens-shaped region first tries to do without the second clip: only the crossing detection of the translated polygon
        // (clip(.., detect_only)), then certified_check(); the full clip + check_post follow when that cannot settle the sign.  One call
        // site of clip() for all of it (the turns of one loop: the routine exists once in the instruction stream).
        // The certified attempt (detect-only clip + certified_check) takes its candidate edges from the contact clip's lists in the OWNER's
        // memory, and is only made while they still stand there: no later clip has rebuilt them (lists_ok), and the owner group does not
        // rebuild them at this very turn of the loop (fullnow: a check that is not certifiable is a full clip in the owner group's own memory
        // at attempt 0, at the same instructions).  Everything else -- not certifiable, lists gone, attempt failed -- is the full clip, which
        // rebuilds the lists of the memory it works in; for a certifiable check that happens at attempt 1, when every reader of attempt 0 is done.
        bool try_cert = false;
        if (pass > 0) {
          if (gl == 0) m.fullnow = (run && !cert) ? 1 : 0;
          gsync();
          try_cert = run && cert && mem[g].lists_ok != 0 && mem[g].fullnow == 0;
        }
        bool settled = false;
        for (int att = 0;; att++) {
          const bool detect = try_cert && att == 0;
          const bool full = run && !detect && (att == 1 || !cert);
          if (detect || full) clip<G>(mem[g], m, gl, ox, oy, pna, poa, pnb, pob, buf, pba, pbb, st, detect, detect);
          if (full && gl == 0) m.lists_ok = pass == 0 ? 1 : 0;      // (this clip has rebuilt the candidate lists of the memory it worked in)
          if (detect) { settled = certified_check<G>(mem[g], m, gl, q); STAMP(st, 14); }
          if (att == 1 || !run || !cert || settled) break;
        }
        if (pass > 0 && run && gl == 0) { m.acc16[3]++; if (settled) m.acc16[4]++; }
#ifdef SZ_STAMPS
        if (pass == 0) st.cA2 += clock64() - tc0;
#endif
        if (pass == 0) {
          if (run) contact_post<G>(m, gl, na, nb, ic, flags, st);
          else if (have && gl == 0) { m.nkeep = 0; m.nchk = 0; m.ierr = 0; m.ff = 0.0; m.lists_ok = 0; }
          gsync();
          for (int k = 0; k < GPB; k++) total += mem[k].nchk;
#ifdef SZ_STAMPS
          { long long now = clock64(); st.cA += now - st.tmark; st.tmark = now; st.ntask += total; int nl = 0; for (int k = 0; k < GPB; k++) nl += mem[k].nx > 0 && mem[k].nkeep > 0; st.nlive += nl; }
#endif
        } else {
          if (run && !settled) check_post<G>(mem[g], m, gl, q, st);
          gsync();
#ifdef SZ_STAMPS
          st.npass++;
#endif
        }
      }
#ifdef SZ_STAMPS
      { long long now = clock64(); st.cB += now - st.tmark; st.tmark = now; }
#endif
    }
    // ================= phase C: friction and the rows of the own item, in region order
    // The ticket for the wavefront's NEXT round is drawn here, in front of phase C: every load of the round has long returned, so the answer is back
    // while the rows are formed -- behind phase C it waits, in order, for the round's row stores and the atomics of its fixed-point totals (and a
    // ticket drawn before the round's own loads makes THEM wait: measured in round 1).  SCAN rounds work several items off: the last one draws.
    if (useq && !todo && nbq * GPB < limit && threadIdx.x == 0) tk_next = atomicAdd(&S.wq[qk * 32], GPB);
    const int gbase = (int)(threadIdx.x & 63) - gl;          // first lane of this group within its wavefront
    const int c_info = __shfl(carry, gbase), c_i = __shfl(carry, gbase + 1), c_j = __shfl(carry, gbase + 2);
    if (have) {
      double* out = S.it_rows + (size_t)it.rows * ROWS_PER_ITEM * 5;
      // fixed-point totals (sz_geom.hpp): the rows are parked in region buffer 1 as well (every direction check of the wavefront is done: it
      // is free, and holds 2 RC >= 5 ROWS_PER_ITEM doubles in the variants whose items can have that many rows)
      const bool fxon = S.facc != nullptr;
      static_assert(2 * RC >= 5 * (RM < ROWS_PER_ITEM ? RM : ROWS_PER_ITEM), "the parked rows live in region buffer 1");
      double* const park = &m.reg[1][0][0];
      int nrows = finish_phase<G>(m, gl, ic, out, ROWS_PER_ITEM, st, fxon ? park : nullptr);
      gsync();
      if (CLS < 2 && (m.ierr & CAPBITS)) {                // working set too small: let the largest variant redo the item
        nrows = 0; flags = IT_RETRY;
        if (gl == 0) { atomicMax(&S.cnt[C_ITEMCLASS], 2); atomicAdd(&S.cnt[C_NRETRY], 1); if (S.retry_stop && S.step > 0) { S.cnt[C_RETRYSTOP] = S.step; S.cnt[C_PAUSED] = S.step; } }
      } else if (gl == 0 && m.ierr) m.err |= m.ierr;
      if (fxon && ((nrows > 0 && !(flags & IT_RETRY)) || (flags & (IT_FUSE | IT_REMOVE)))) {
        // the item's two rows come from the work list again (a hit in L2: the entry was read when the item started) rather than being
        // held in two registers through the clips -- the kernel sits on its register budget, and nothing waits for these atomics
        const int ri = c_i, rj = is_pair ? c_j : -1;
        int glw = gl; asm volatile("" : "+v"(glw));          // (opaque: nothing lane-dependent of this block is hoisted out of the item loop and spilled)
        if (nrows > 0 && !(flags & IT_RETRY) && glw < 7) {          // lane w < 7: word w of what the item adds to floe i and to floe j
          long long qi = 0, li = 0, qj = 0, lj = 0; int bad = 0;
          const int eFi = fx_force_exp(S.kexp, m.kin[KIN_AREA_I], m.kin[KIN_H_I]), eAi = fx_area_exp(m.kin[KIN_AREA_I]);
          const int eFj = rj >= 0 ? fx_force_exp(S.kexp, m.kin[KIN_AREA_J], m.kin[KIN_H_J]) : 0, eAj = rj >= 0 ? fx_area_exp(m.kin[KIN_AREA_J]) : 0;
          for (int r = 0; r < nrows; r++) {
            fx_word(glw, park + r * 5, 1.0, m.kin[KIN_I], m.kin[KIN_I + 1], eFi, eAi, m.eri, qi, li, bad);
            if (rj >= 0) fx_word(glw, park + r * 5, -1.0, m.kin[KIN_J], m.kin[KIN_J + 1], eFj, eAj, m.erj, qj, lj, bad);
          }
          unsigned long long* const ai = (unsigned long long*)(S.facc + (size_t)ri * FX_WORDS);
          if (qi) atomicAdd(ai + glw, (unsigned long long)qi);
          if (li) atomicAdd(ai + 8 + glw, (unsigned long long)li);
          if (rj >= 0) {
            unsigned long long* const aj = (unsigned long long*)(S.facc + (size_t)rj * FX_WORDS);
            if (qj) atomicAdd(aj + glw, (unsigned long long)qj);
            if (lj) atomicAdd(aj + 8 + glw, (unsigned long long)lj);
          }
          if (bad) atomicOr(&S.cnt[C_ERR], ERR_FIXED_RANGE);
        }
        // status tags (collisions.jl:367, 438, 525, 801-806) as bits beside the totals: 1 fuse as the pair's first floe, 2 remove (domain element),
        // 4 fuse as its second floe -- the integrator resolves them in the reference's order.  A tag on a floe this context integrates ends the
        // batch after this step (simplify_floes!, simulation.jl:205-214), and is raised HERE so that the step's integrator already knows
        if ((flags & (IT_FUSE | IT_REMOVE)) && gl == 0) {
          const int nown = S.cnt[C_NOWN];
          if (flags & IT_FUSE) { atomicOr((int*)(S.facc + (size_t)ri * FX_WORDS + 7), 1); if (rj >= 0) atomicOr((int*)(S.facc + (size_t)rj * FX_WORDS + 7), 4); }
          if (flags & IT_REMOVE) atomicOr((int*)(S.facc + (size_t)ri * FX_WORDS + 7), 2);
          if (ri < nown || (rj >= 0 && rj < nown && (flags & IT_FUSE))) request_stop_new_tag(S);
        }
      }
      if (gl == 0) {
        S.it_info[c_info] = make_int2(nrows | (flags << 8), it.rows);
        if (flags & IT_FUSE) atomicAdd(&S.cnt[C_NFUSE], 1);
        if (!(flags & IT_RETRY)) {                           // counted by the variant that finishes the item
          if (is_pair) { m.acc16[0]++; m.acc[0] += (unsigned)(na + nb); m.acc[1] += (unsigned)nrows; } else { m.acc16[1]++; m.acc16[2] += (uint16_t)nrows; }
        }
      }
      STAMP(st, 11);
#ifdef SZ_STAMPS
      if (st.on) { if (nrows > 0) { S.stamps[0] = st.n; if (!(dbg & 16) || st.pass) st.on = false; } else { st.n = 0; st.t0 = clock64(); } }
      st.maxrows = st.maxrows > nrows ? st.maxrows : nrows;
#endif
    }
   }
#ifdef SZ_STAMPS
   { long long now = clock64(); st.cC += now - st.tmark; st.tmark = now; }
   if ((dbg & 16) && !st.pass) { st.pass = 1; continue; }      // timing experiment: the same round again, now with a warm instruction cache
#endif
   // (a segment the first -- static -- round has covered hands out nothing more: no ticket.  The ticket is a RETURNING atomic, and it comes back
   //  in order, behind the round's stores and the atomics of its fixed-point totals, through a texture path the forcing wavefronts of the launch's
   //  tail keep busy: the slowest wavefronts of a one-round launch -- which set its length -- waited ~50 k cycles for a ticket that said "nothing")
   if (useq && nbq * GPB >= limit) break;
   if (useq) t0 = nbq * GPB + __shfl(tk_next, 0);
   else t0 += nbq * STRIDE;
  }
  gsync();
  if (gl == 0 && m.err) atomicOr(&S.cnt[C_ERR], m.err);
  if (gl == 0 && m.ntracefail) atomicAdd(&S.cnt[C_TRACE_FAIL], m.ntracefail);
  {
    // one set of atomics per wavefront, spread over ACC_SLOTS lines (same-address atomics serialise chip-wide)
    unsigned v[7];
    for (int k = 0; k < 7; k++) { v[k] = gl != 0 ? 0u : k == 0 ? m.acc16[0] : k == 1 ? m.acc[0] : k == 2 ? m.acc[1] : k == 3 ? m.acc16[1] : k == 4 ? m.acc16[2] : k == 5 ? m.acc16[3] : m.acc16[4]; for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d); }
    if ((threadIdx.x & 63) == 0 && !(dbg & 32)) {          // (SZ_DEBUG=32: without the work counters -- a timing experiment)
      unsigned long long* a = S.acc + (size_t)((bidx * (TPB / 64) + (threadIdx.x >> 6)) % ACC_SLOTS) * 8;
      for (int k = 0; k < 7; k++) if (v[k]) atomicAdd(a + 1 + k, (unsigned long long)v[k]);
      if (CLS == 0 && bidx == 0 && threadIdx.x == 0) atomicAdd(a, 1ull);
    }
  }
#ifdef SZ_STAMPS
  // lifetimes of the wavefronts (4096-cycle buckets: stamps[256 + bucket]) and their mean by the largest row
  // count among the wavefront's items (stamps[400 + 2r] cycles, [401 + 2r] wavefronts)
  if (CLS == 0) {
    int mr = st.maxrows;
    for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(mr, d); mr = mr > o ? mr : o; }
    if (threadIdx.x == 0) {
      long long el = clock64() - st.t0w; int bkt = (int)(el >> 12); if (bkt > 127) bkt = 127;
      atomicAdd((unsigned long long*)&S.stamps[256 + bkt], 1ull);
      atomicAdd((unsigned long long*)&S.stamps[400 + 2 * mr], (unsigned long long)el);
      atomicAdd((unsigned long long*)&S.stamps[401 + 2 * mr], 1ull);
      // one record per wavefront (stamps[512 ..]): lifetime | passes of the check loop | check tasks | items with a contact clip that
      // found crossings | cycles in phase A (staging + contact clip + contact_post), B (checks), C (rows)
      const unsigned long long slot = atomicAdd((unsigned long long*)&S.stamps[511], 1ull);
      if (slot < 8000) {
        long long* r = S.stamps + 512 + slot * 8;
        r[0] = el; r[1] = st.npass | ((st.cP > 0 ? st.cP >> 8 : 0) << 8); r[2] = st.ntask; r[3] = st.nlive; r[4] = st.cA; r[5] = st.cB; r[6] = st.cC; r[7] = mr | ((st.cA1 >> 8) << 8) | ((st.cA2 >> 8) << 36);
      }
    }
  }
#endif
}

constexpr int NARROW_CAP0 = 18, NARROW_CAP1 = 32, NARROW_CAP2 = 255;   // ring points per variant (18: rings of up to 17 vertices; the
                                                                       // working set of the first variant is cut to fit 2 KB of LDS per item; 255: ring
                                                                       // sizes and edge indices travel as bytes -- the reference's own test shapes,
                                                                       // test/inputs/floe_shapes.jld2, have rings of up to 203 points)
constexpr int NARROW_KC2 = 64, NARROW_RC2 = 640;                       // crossings / region points of the largest variant

// items not touched by any narrow variant would keep stale row counts: clear them first
__global__ void sz_k_items_clear(State S) {
  if (stopped(S)) return;
  for (int s = 0; s < NSEG; s++) {
    const Seg sg = seg_of(S, s);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < sg.n; t += gridDim.x * blockDim.x) {
      // size class of the item: which narrow-phase variant takes it
      const Item it = item_of(S, sg, t);
      const int nb = it.nb;
      S.it_info[it.info] = make_int2(0, it.rows);
      int na = it.na;
      int big = na > nb ? na : nb;
      int cls = big <= NARROW_CAP0 ? 0 : (big <= NARROW_CAP1 ? 1 : 2);
      if (cls > 0) atomicMax(&S.cnt[C_ITEMCLASS], cls);
    }
  }
}

// ============================================================================ reduce (A9, A11)
// Interaction rows live at a fixed stride (ROWCAP rows per floe): no offsets, no scan in the step;
// sz_download_interactions compacts them to CSR on demand.  A floe with more rows than ROWCAP
// raises ERR_CAP_INTER.
constexpr int ROWCAP = 32;      // (the default of State::rowcap)

// A group of IF_G lanes per floe.  Appends the rows of floe f -- own pairs (j asc), domain elements
// (N,S,E,W, topography), rows mirrored from partners that come earlier in the serial order (i asc,
// force negated) -- to dst at row c, points shifted by (-sx, -sy); every lane takes one source (an
// item), the row positions come from a prefix sum over the lanes, so the rows land in exactly the
// serial order.  Returns the new count (clamped to cap, overflow flagged).  When `st` is given it
// also resolves the status tag of f: tagA after the pair/domain phase (collisions.jl:367,438,525),
// st after the mirror pass (:801-806).
constexpr int IF_G = 8;
// pre (may be null): {n_out[f], el_off[f], el_off[f + 1], n_in[f]} already in registers
__device__ __forceinline__ int emit_rows(const State& S, int lane, int f, double* dst, int c, int cap, double sx, double sy,
                                         int mirror, bool& ovf, int* st, int* tagA, const int4* pre = nullptr) {
  const int nown = pre ? pre->x : S.n_out[f];
  const int e0 = pre ? pre->y : S.el_off[f], nel = (pre ? pre->z : S.el_off[f + 1]) - e0;
  const int nin = !mirror ? 0 : pre ? pre->w : S.n_in[f];
  const int T = nown + nel + nin;
  const int gshift = (int)(threadIdx.x & 63) / IF_G * IF_G;
  unsigned fuse_own = 0, rem_el = 0, fuse_in = 0;
  for (int base = 0; base < T; base += IF_G) {
    const int s = base + lane;
    int info = -1, n = 0, kind = -1; double idx = 0.0, sign = 1.0;
    if (s < nown) { info = f * S.maxnb + s; idx = (double)(S.okey[S.nb_out[info]] + 1); kind = 0; }
    else if (s < nown + nel) { int q = e0 + (s - nown); info = S.capM * S.maxnb + q; idx = -(double)(S.el_elem[q] + 1); kind = 1; }
    else if (s < T) {
      const int i = S.nb_in[(size_t)f * S.maxnb + (s - nown - nel)];
      const int ni = S.n_out[i];
      for (int q = 0; q < ni; q++) if (S.nb_out[(size_t)i * S.maxnb + q] == f) info = i * S.maxnb + q;     // pair (i, f); absent if the Dict rule dropped it
      idx = (double)(S.okey[i] + 1); sign = -1.0; kind = 2;
    }
    int fl = 0; int2 iv = make_int2(0, 0);
    if (info >= 0) { iv = S.it_info[info]; n = iv.x & 0xff; fl = iv.x >> 8; }
    fuse_own |= (unsigned)(__ballot(kind == 0 && (fl & IT_FUSE)) >> gshift) & 0xffu;
    rem_el |= (unsigned)(__ballot(kind == 1 && (fl & IT_REMOVE)) >> gshift) & 0xffu;
    fuse_in |= (unsigned)(__ballot(kind == 2 && (fl & IT_FUSE)) >> gshift) & 0xffu;
    int inc = n;
    for (int d = 1; d < IF_G; d <<= 1) { int t = __shfl_up(inc, d, IF_G); if (lane >= d) inc += t; }
    const int tot = __shfl(inc, IF_G - 1, IF_G), off = inc - n;
    const double* src = S.it_rows + (size_t)(n > 0 ? iv.y : 0) * ROWS_PER_ITEM * 5;
    for (int r = 0; r < n; r++) {
      const int pos = c + off + r;
      if (pos < cap) {
        double* d = dst + (size_t)pos * 7; const double* q = src + r * 5;
        d[0] = idx; d[1] = q[0] * sign; d[2] = q[1] * sign; d[3] = q[2] - sx; d[4] = q[3] - sy; d[5] = 0.0; d[6] = q[4];
      } else ovf = true;
    }
    c = c + tot < cap ? c + tot : cap;
  }
  if (st) {
    if (fuse_own) *st = SZ_FUSE;
    if (rem_el) *st = SZ_REMOVE;
    *tagA = *st;
    if (fuse_in) *st = SZ_FUSE;
  }
  return c;
}
// The rows of parent f and, behind them, the rows of its ng ghosts shifted into the parent's frame (the ghost fold of
// collisions.jl:830-850) in ONE pass: the sources of all of them form one list that the lanes work off together, so the chains of
// dependent loads behind a source (mirrored rows: neighbour list -> partner's list -> item -> rows) run side by side for the parent and
// its ghosts instead of one emit_rows() after the other (a parent next to a periodic wall used to take 1 + ng times as long as the
// others, and set the launch's duration).  Same rows at the same positions.  pre0: {n_out, el_off[f], el_off[f + 1], n_in} of f.
__device__ __forceinline__ int emit_rows_fold(const State& S, int lane, int f, double* dst, int cap, int mirror, bool& ovf, int* st, int* tagA,
                                              const int4& pre0, int ng, const int* gf, double cx, double cy) {
  int sf[MAX_GHOSTS + 1], so[MAX_GHOSTS + 1], se[MAX_GHOSTS + 1], sn[MAX_GHOSTS + 1], si[MAX_GHOSTS + 1], sT[MAX_GHOSTS + 2];
  double ssx[MAX_GHOSTS + 1], ssy[MAX_GHOSTS + 1];
  sf[0] = f; so[0] = pre0.x; se[0] = pre0.y; sn[0] = pre0.z - pre0.y; si[0] = mirror ? pre0.w : 0; ssx[0] = 0.0; ssy[0] = 0.0;
#pragma unroll
  for (int q = 1; q <= MAX_GHOSTS; q++) {
    const bool on = q <= ng; const int g = on ? gf[q - 1] : f;
    const int e0 = S.el_off[g];
    sf[q] = g; so[q] = on ? S.n_out[g] : 0; se[q] = e0; sn[q] = on ? S.el_off[g + 1] - e0 : 0; si[q] = on && mirror ? S.n_in[g] : 0;
    ssx[q] = on ? S.cx[g] - cx : 0.0; ssy[q] = on ? S.cy[g] - cy : 0.0;
  }
  sT[0] = 0;
#pragma unroll
  for (int q = 0; q <= MAX_GHOSTS; q++) sT[q + 1] = sT[q] + so[q] + sn[q] + si[q];
  const int T = sT[MAX_GHOSTS + 1];
  const int gshift = (int)(threadIdx.x & 63) / IF_G * IF_G;
  unsigned fuse_own = 0, rem_el = 0, fuse_in = 0;
  int c = 0;
  for (int base = 0; base < T; base += IF_G) {
    const int s = base + lane;
    int q = 0;
#pragma unroll
    for (int t = 1; t <= MAX_GHOSTS; t++) q += s >= sT[t] ? 1 : 0;
    int fq = sf[0], nown = so[0], e0 = se[0], nel = sn[0], s0 = sT[0]; double sx = ssx[0], sy = ssy[0];
#pragma unroll
    for (int t = 1; t <= MAX_GHOSTS; t++) if (q == t) { fq = sf[t]; nown = so[t]; e0 = se[t]; nel = sn[t]; s0 = sT[t]; sx = ssx[t]; sy = ssy[t]; }
    const int ls = s - s0;
    int info = -1, n = 0, kind = -1; double idx = 0.0, sign = 1.0;
    if (s < T) {
      if (ls < nown) { info = fq * S.maxnb + ls; idx = (double)(S.okey[S.nb_out[info]] + 1); kind = 0; }
      else if (ls < nown + nel) { int qq = e0 + (ls - nown); info = S.capM * S.maxnb + qq; idx = -(double)(S.el_elem[qq] + 1); kind = 1; }
      else {
        const int i = S.nb_in[(size_t)fq * S.maxnb + (ls - nown - nel)];
        const int ni = S.n_out[i];
        for (int qq = 0; qq < ni; qq++) if (S.nb_out[(size_t)i * S.maxnb + qq] == fq) info = i * S.maxnb + qq;     // pair (i, fq); absent if the Dict rule dropped it
        idx = (double)(S.okey[i] + 1); sign = -1.0; kind = 2;
      }
    }
    int fl = 0; int2 iv = make_int2(0, 0);
    if (info >= 0) { iv = S.it_info[info]; n = iv.x & 0xff; fl = iv.x >> 8; }
    // (the tags are the parent's own: its ghosts' items do not tag it here)
    fuse_own |= (unsigned)(__ballot(q == 0 && kind == 0 && (fl & IT_FUSE)) >> gshift) & 0xffu;
    rem_el |= (unsigned)(__ballot(q == 0 && kind == 1 && (fl & IT_REMOVE)) >> gshift) & 0xffu;
    fuse_in |= (unsigned)(__ballot(q == 0 && kind == 2 && (fl & IT_FUSE)) >> gshift) & 0xffu;
    int inc = n;
    for (int d = 1; d < IF_G; d <<= 1) { int t = __shfl_up(inc, d, IF_G); if (lane >= d) inc += t; }
    const int tot = __shfl(inc, IF_G - 1, IF_G), off = inc - n;
    const double* src = S.it_rows + (size_t)(n > 0 ? iv.y : 0) * ROWS_PER_ITEM * 5;
    for (int r = 0; r < n; r++) {
      const int pos = c + off + r;
      if (pos < cap) {
        double* d = dst + (size_t)pos * 7; const double* qr = src + r * 5;
        d[0] = idx; d[1] = qr[0] * sign; d[2] = qr[1] * sign; d[3] = qr[2] - sx; d[4] = qr[3] - sy; d[5] = 0.0; d[6] = qr[4];
      } else ovf = true;
    }
    c = c + tot < cap ? c + tot : cap;
  }
  if (fuse_own) *st = SZ_FUSE;
  if (rem_el) *st = SZ_REMOVE;
  *tagA = *st;
  if (fuse_in) *st = SZ_FUSE;
  return c;
}
// mirror pass, ghost fold, torque and totals (collisions.jl:799-862)
// m_hint (resident steps; 0: none): about how many floes there are, from the host -- the floe a group starts with then does not
// depend on the device's count, and its first loads go out together with the counter block (one round trip less in the launch's
// chain).  Only the mapping uses the hint: floes beyond it are picked up afterwards, groups beyond the real count idle.
// rows_only (resident batches, round 4): the floes' totals, the parents' status tags and floe.overarea are the integrator's, from the fixed-point
// words of the narrow phase (State::facc); this launch then only assembles floe.interactions (rows, torque column, counts) and the
// tags of the ghost rows -- once, BEHIND the batch, for the step that ended it (oldc: the parents have been moved since: their
// centroids of that step are in `mot`), or inside every step on the paths that keep it there
__global__ void __launch_bounds__(128) sz_k_inter_fill(State S, int mirror, int n_init_arg, int m_hint, int rows_only, int oldc) {
  const StopRegs stop = stop_load(S);
  // (behind a batch that is paused inside a step: that step is finished first; oldc - 1 = the 1-based step the launch was enqueued for: a
  //  batch that a tag ended BEFORE it has this launch again, for the step that did end it)
  if (oldc && (stop.r != 0 || (stop.s != 0 && stop.s < oldc - 1))) return;
  const int M = S.cnt[C_M];
  const int nparents = S.cnt[C_NOWN];
  const int n_init = n_init_arg >= 0 ? n_init_arg : S.cnt[C_N];   // < 0: every parent on the device
  const int lane = threadIdx.x % IF_G, gpb = blockDim.x / IF_G, grp = threadIdx.x / IF_G;
  const int mh = m_hint < S.capM ? m_hint : S.capM, nact_h = (mh + gpb - 1) / gpb;
  const bool hinted = mh > 0 && nact_h <= (int)gridDim.x;
  int k0 = -1;
  // (pipelined steps: the rows behind the parents lie goff further on -- the launch walks VIRTUAL rows [0, M) and maps them)
  const int nsplit = S.goff != 0 ? S.cnt[C_N] : 0x7fffffff;
  auto phys = [&](int kv) { return kv >= nsplit ? kv + S.goff : kv; };
  if (hinted) { const int vb = xcd_contiguous((int)blockIdx.x, (int)gridDim.x, nact_h); if (vb >= 0) k0 = vb * gpb + grp; }
  const int k0v = k0;
  if (k0 >= 0) k0 = phys(k0) < S.capM ? phys(k0) : -1;
  struct Pre { long long gid; double cx, cy; int st, par, ng; int4 cnts; } pre = { 0, 0.0, 0.0, 0, 0, 0, make_int4(0, 0, 0, 0) };
  const int nmoved = oldc ? S.cnt[C_NOWN] : 0;        // rows [0, nmoved): their centroid of the step is in mot
  auto cxy = [&](int k, double& x, double& y) { if (k < nmoved) { const double2 o = *(const double2*)(S.mot + (size_t)k * 4); x = o.x; y = o.y; } else { x = S.cx[k]; y = S.cy[k]; } };
  if (k0 >= 0) {
    pre.gid = S.ghost_id[k0]; cxy(k0, pre.cx, pre.cy); pre.st = S.status[k0]; pre.par = S.parent[k0]; pre.ng = S.ngh[k0] & 0xff;
    pre.cnts = make_int4(S.n_out[k0], S.el_off[k0], S.el_off[k0 + 1], S.n_in[k0]);
  }
  loads_issued();
  if (stop_test_late(S, stop)) return;
  // the narrow phase has consumed the work list: heads and lengths of its segments start from zero for the next step
  // (a launch whose grid is smaller than NSEG * 2 ints would be odd: threads 0 .. 15 of workgroup 0 do it)
  if (blockIdx.x == 0 && threadIdx.x < 2 * NSEG) S.wq[(threadIdx.x >> 1) * 32 + (threadIdx.x & 1)] = 0;
  auto reduce = [&](int k, const Pre* pr) {
    double* dst = S.inter_rows + (size_t)k * S.rowcap * 7;
    const bool is_ghost = (pr ? pr->gid : S.ghost_id[k]) != 0;
    double cx, cy;
    if (pr) { cx = pr->cx; cy = pr->cy; } else cxy(k, cx, cy);
    double sx = 0.0, sy = 0.0;
    const int par = pr ? pr->par : S.parent[k];
    if (mirror && is_ghost && par < n_init) { double pcx, pcy; cxy(par, pcx, pcy); sx = cx - pcx; sy = cy - pcy; }
    bool ovf = false;
    int st = pr ? pr->st : S.status[k], tagA = st;
    const bool totals = mirror && k < n_init;
    int c;
    if (totals) {                      // own rows + ghost fold (collisions.jl:830-850), one pass
      const int ng = pr ? pr->ng : (S.ngh[k] & 0xff);
      int gf[MAX_GHOSTS];
#pragma unroll
      for (int g = 0; g < MAX_GHOSTS; g++) gf[g] = S.gh[k * MAX_GHOSTS + g];
      const int4 cn = pr ? pr->cnts : make_int4(S.n_out[k], S.el_off[k], S.el_off[k + 1], S.n_in[k]);
      c = emit_rows_fold(S, lane, k, dst, S.rowcap, mirror, ovf, &st, &tagA, cn, ng, gf, cx, cy);
    } else c = emit_rows(S, lane, k, dst, 0, S.rowcap, sx, sy, mirror, ovf, &st, &tagA, pr ? &pr->cnts : nullptr);
    if (ovf) { atomicOr(&S.cnt[C_ERR], ERR_CAP_INTER); capacity_stop(S); }
    __threadfence_block();             // the rows were written by other lanes of this wavefront
    // torque per row over the lanes; totals (collisions.jl:747-749, 852-861) and the overlap sum in row
    // order; ghosts keep zero totals.  Without the mirror pass (floe_floe_interaction! /
    // floe_domain_interaction! entry points) the torque is filled for convenience.
    double fx = 0.0, fy = 0.0, tq = 0.0, over = 0.0;
    for (int base = 0; base < c; base += IF_G) {
      const int r = base + lane;
      double f1 = 0.0, f2 = 0.0, t = 0.0, ov = 0.0;
      if (r < c) {
        double* d = dst + (size_t)r * 7;
        f1 = d[1]; f2 = d[2]; ov = d[6];
        if (totals || !mirror) { double xp = d[3] - cx, yp = d[4] - cy; t = xp * f2 - yp * f1; d[5] = t; }
      }
      const int cnt = c - base < IF_G ? c - base : IF_G;
      for (int l = 0; l < cnt; l++) {
        fx += __shfl(f1, l, IF_G); fy += __shfl(f2, l, IF_G); tq += __shfl(t, l, IF_G); over += __shfl(ov, l, IF_G);
      }
    }
    if (lane == 0 && rows_only) {
      S.inter_cnt[k] = c;
      // (the tags of the parents are the integrator's, from the bits the narrow phase raised; a ghost's own tags are resolved here -- the fuse
      //  replay walks them, collisions.jl:801-806 -- and so is every row's status before the mirror pass: for a parent it is re-derived from the
      //  status the integrator has left, which differs from the original only where the replay does not look -- a floe fused as SECOND floe
      //  of its pairs has no fuse list of its own to propagate, one removed by the coupling is not `fuse`)
      S.tagA[k] = tagA;
      if (k >= nparents) S.status[k] = st;
    }
    if (lane == 0 && !rows_only) {
      S.status[k] = st; S.tagA[k] = tagA;
      if (st != SZ_ACTIVE && k < nparents) request_stop(S);        // simplify_floes! has work after this step
      S.inter_cnt[k] = c;
      S.cfx[k] = totals ? fx : 0.0; S.cfy[k] = totals ? fy : 0.0; S.ctrq[k] = totals ? tq : 0.0;
      // floe.overarea accumulates (collisions.jl:304).  A call that is run AGAIN after its lists were carved larger (grow_lists) must
      // add its overlap once, and the complete one: the column's value from before the call is kept beside the number of the call
      double base = S.overarea[k];
      if (S.over_stamp[k] != S.callid) { S.over_base[k] = base; S.over_stamp[k] = S.callid; } else base = S.over_base[k];
      S.overarea[k] = base + over;
    }
  };
  if (hinted) {
    if (k0 >= 0 && k0v < M) reduce(k0, &pre);
    for (int k = nact_h * gpb + (int)blockIdx.x * gpb + grp; k < M; k += gridDim.x * gpb) reduce(phys(k), nullptr);      // beyond the hint
  } else {
    const int vb0 = xcd_contiguous((int)blockIdx.x, (int)gridDim.x, (M + gpb - 1) / gpb);
    for (int k = vb0 < 0 ? M : vb0 * gpb + grp; k < M; k += gridDim.x * gpb) reduce(phys(k), nullptr);
  }
}
// CSR compaction of the fixed-stride rows (sz_download_interactions only)
__global__ void sz_k_inter_compact(State S, double* dst) {
  int M = S.cnt[C_M];
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < M; k += gridDim.x * blockDim.x) {
    int n = S.inter_cnt[k]; const double* src = S.inter_rows + (size_t)k * S.rowcap * 7;
    double* d = dst + (size_t)S.inter_off[k] * 7;
    for (int q = 0; q < n * 7; q++) d[q] = src[q];
  }
}
// update_boundaries!, collisions.jl:565-571, boundaries.jl:526-568 (MovingBoundary only)
__global__ void sz_k_update_boundaries(State S, int dt) {
  int e = threadIdx.x;
  if (blockIdx.x != 0 || e >= 4 || S.ekind[e] != 3 || stopped_late(S)) return;
  double* rc = S.erect + e * 4;   // xmin, xmax, ymin, ymax
  if (e < 2) { double dy = S.ev[e] * dt; rc[2] += dy; rc[3] += dy; S.eval[e] += dy; }
  else { double dx = S.eu[e] * dt; rc[0] += dx; rc[1] += dx; S.eval[e] += dx; }
  int o = S.eoff[e];
  S.ex[o] = rc[0]; S.ey[o] = rc[2]; S.ex[o + 1] = rc[0]; S.ey[o + 1] = rc[3]; S.ex[o + 2] = rc[1]; S.ey[o + 2] = rc[3];
  S.ex[o + 3] = rc[1]; S.ey[o + 3] = rc[2]; S.ex[o + 4] = rc[0]; S.ey[o + 4] = rc[2];
  S.ebb[4 * e] = rc[0]; S.ebb[4 * e + 1] = rc[1]; S.ebb[4 * e + 2] = rc[2]; S.ebb[4 * e + 3] = rc[3];
}

// ============================================================================ forcings (A13)
// bilinear sample on the grid-line lattice (Interpolations.linear_interpolation over the knot window
// of mc_interpolation, coupling.jl:845-902, incl. the periodic wrap of find_interp_knots :702-744).
// Cell and weights depend only on the point, so they are computed once for all five fields.
struct LatticeCell { int o00, o01, o10, o11; double tx, ty; };
__device__ __forceinline__ int wrap_index(int i, int n) {       // i mod n for i within a few periods of [0, n)
  if (i < 0) { i += n; if (i < 0) i = ((i % n) + n) % n; }
  else if (i >= n) { i -= n; if (i >= n) i %= n; }
  return i;
}
__device__ __forceinline__ LatticeCell lattice_cell(const State& S, double x, double y, int per_x, int per_y) {
  int Nx = S.Nx, Ny = S.Ny;
  // cell index and weights with the reciprocal spacing: a point on a grid line may land in the
  // neighbouring cell, where bilinear interpolation gives the same value
  int ix = (int)floor((x - S.gx0) * S.rdx), iy = (int)floor((y - S.gy0) * S.rdy);
  if (!per_x) { if (ix < 0) ix = 0; if (ix > Nx - 1) ix = Nx - 1; }
  if (!per_y) { if (iy < 0) iy = 0; if (iy > Ny - 1) iy = Ny - 1; }
  double xk = S.gx0 + (double)ix * S.gdx, yk = S.gy0 + (double)iy * S.gdy;
  LatticeCell c;
  c.tx = (x - xk) * S.rdx; c.ty = (y - yk) * S.rdy;
  int i0, i1, j0, j1;
  if (per_x) { i0 = wrap_index(ix, Nx); i1 = i0 + 1 == Nx ? 0 : i0 + 1; } else { i0 = ix; i1 = ix + 1; }
  if (per_y) { j0 = wrap_index(iy, Ny); j1 = j0 + 1 == Ny ? 0 : j0 + 1; } else { j0 = iy; j1 = iy + 1; }
  int s = Ny + 1;
  c.o00 = i0 * s + j0; c.o01 = i0 * s + j1; c.o10 = i1 * s + j0; c.o11 = i1 * s + j1;
  return c;
}
// The five lattices are interleaved per node (8 doubles = one 64-byte line: uo, vo, hf, ua, va, pad):
// a point touches 4 lines instead of 10.  Field f of the four corner nodes -> bilinear value.
// in_bounds, coupling.jl:494-597 (four methods dispatched on the boundary kinds): a direction with a periodic pair admits every coordinate
__device__ __forceinline__ bool point_in_bounds(const State& S, double x, double y, int per_x, int per_y) {
  return (per_x || (S.gx0 <= x && x <= S.gxf)) && (per_y || (S.gy0 <= y && y <= S.gyf));
}
__device__ __forceinline__ double sample_field(const double* nodes, int f, const LatticeCell& c) {
  double c0 = (1.0 - c.ty) * nodes[(size_t)c.o00 * 8 + f] + c.ty * nodes[(size_t)c.o01 * 8 + f];
  double c1 = (1.0 - c.ty) * nodes[(size_t)c.o10 * 8 + f] + c.ty * nodes[(size_t)c.o11 * 8 + f];
  return (1.0 - c.tx) * c0 + c.tx * c1;
}
// test hook (sz_debug_sample_fields): the in-bounds test and the lattice sample of the forcing kernels at given points -- out[12 k ..] = in_bounds,
// uocn, vocn, hflx, uatm, vatm, the 1-based grid lines west / east / south / north the blend reads, its weights tx, ty
__global__ void sz_k_debug_sample(State S, int n, const double* x, const double* y, double* out) {
  const int per_x = S.ekind[2] == 1, per_y = S.ekind[0] == 1, s = S.Ny + 1;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    double* o = out + (size_t)k * 12;
    o[0] = point_in_bounds(S, x[k], y[k], per_x, per_y) ? 1.0 : 0.0;
    const LatticeCell c = lattice_cell(S, x[k], y[k], per_x, per_y);
    for (int f = 0; f < 5; f++) o[1 + f] = sample_field(S.nodes, f, c);
    o[6] = (double)(c.o00 / s + 1); o[7] = (double)(c.o10 / s + 1); o[8] = (double)(c.o00 % s + 1); o[9] = (double)(c.o01 % s + 1);
    o[10] = c.tx; o[11] = c.ty;
  }
}
__global__ void sz_k_interleave_fields(State S) {
  int n = (S.Nx + 1) * (S.Ny + 1);
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
    double* d = S.nodes + (size_t)q * 8;
    d[0] = S.uo[q]; d[1] = S.vo[q]; d[2] = S.hf[q]; d[3] = S.ua[q]; d[4] = S.va[q]; d[5] = 0.0; d[6] = 0.0; d[7] = 0.0;
  }
}

// apply the removal flags of the forcing kernel (standalone timestep_coupling! call)
__global__ void sz_k_apply_frc(State S) {
  int N = S.cnt[C_NOWN];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
    if (S.frc_remove[i]) S.status[i] = SZ_REMOVE;
}
// Tiled steps evaluate the forcings beside the halo exchange, BEFORE the step's ghost pass; the reference couples after it, i.e. a
// parent whose centroid has left the domain through a periodic wall has swapped with its ghost by then.  The forcing kernels of a tiled
// context therefore apply the swap's translation themselves (the ghost pass's own test and addition: same bits as the row it will store).
__device__ __forceinline__ void forcing_wrap(const State& S, int i, double& cx, double& cy) {
  if (!S.tiled || S.status[i] != SZ_ACTIVE) return;
  if (S.any_periodic_ew) { const double maxv = S.eval[2], minv = S.eval[3], L = maxv - minv; if (cx < minv) cx = cx + L; else if (maxv < cx) cx = cx + (-L); }
  if (S.any_periodic_ns) { const double maxv = S.eval[0], minv = S.eval[1], L = maxv - minv; if (cy < minv) cy = cy + L; else if (maxv < cy) cy = cy + (-L); }
}
constexpr int FRC_G = 32;      // lanes per floe of the two-way variant
#ifndef FRC_PLAIN_LANES
#define FRC_PLAIN_LANES 16
#endif
constexpr int FRC_PLAIN = FRC_PLAIN_LANES;   // lanes per floe of the one-way kernels.  Round 4, lean loop (tools/probe/r4_frc_lanes.sh; ms/step at 10 k | 100 k floes):
                                             // 64 lanes 0.0846 | 0.5173, 32 lanes 0.0772 | 0.4894, 16 lanes 0.0765 | 0.4773, 8 lanes 0.0774 | 0.5038 -- a floe's ~100 points
                                             // fill 7 trips of 16 lanes to 90 %, 4 trips of 32 to 78 %; with 8 lanes the floes' scalar loads take over
// Two-way coupling (TW): the kernel also fills the floe's part of grid.floe_locations / ocean.scells
// (floe_to_grid_info!, coupling.jl:1417-1454): per distinct centre cell its sub-floe points fall into, the
// periodic shift of the first such point, the sum of minus the ocean stress over the points IN POINT ORDER, and
// their number.  Points (cell, stress) are parked in LDS; slots are opened in order of first appearance; every
// lane then owns two slots and walks the points in order.
constexpr int FC_CAP = 64;      // distinct centre cells per floe
constexpr int TW_PMAX = 512;    // sub-floe points per floe with two-way coupling on
constexpr int TW_FPB = 4;       // floes per workgroup of the two-way variant (128 threads)
// pmax (two-way only): sub-floe points per floe the launch provides LDS for (dynamic: 21 bytes per point and floe;
// the host sizes it from the largest floe, so that as many floes as possible are in flight per CU)
// bid / nblk: rank and number of the forcing workgroups; first: physical id of the first of them in the launch.  The floes of one
// XCD's workgroups are a contiguous index range (= a region in space): a lattice node is then fetched by one or two XCDs' L2
// instead of all eight
// ---- the one-way forcings with lean per-point arithmetic (round 4; SZ_FRC_LEAN=0 compiles the plain loop back in for A/B).
// The contract on fxOA / fyOA / trqOA / hflx is a tolerance against the reference (1e-10 relative; the reference's own tests hold them to
// 1e-3), not bit-equality with the straightforward evaluation -- and the kernel is bound by the fp64 instructions it issues (L1 launch at
// 10 k floes: 63 % of all issue cycles busy, a third of its vector instructions are this loop).  What stays EXACTLY as before: the
// point's world coordinates, the in-bounds test and the lattice cell (the discrete decisions).  What changes, a few ulps per point:
//   * the bilinear blend as four weights (one product + three fused multiply-adds per field instead of six operations),
//   * square roots from the hardware reciprocal-square-root estimate + two coupled Newton steps (~1 ulp; the arguments are squared
//     relative speeds: zero or far above the denormal range),
//   * the stress sums contracted into fused multiply-adds,
//   * the next point's body coordinates are asked for before the current point is worked on.
#ifndef SZ_FRC_LEAN
#define SZ_FRC_LEAN 1
#endif
__device__ __forceinline__ double sqrt_fast(double s) {
  const double y = __builtin_amdgcn_rsq(s);
  double g = s * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  const double d = fma(-g, g, s);
  g = fma(d, h, g);
  return s > 0.0 ? g : 0.0;
}
template <int FG>
__device__ __forceinline__ void forcing_lean_body(State& S, const Params& P, int bid, int nblk, int first);

template <bool TW>
__device__ __forceinline__ void forcing_body(State& S, const Params& P, int bid, int nblk, int pmax, int first) {
  if constexpr (!TW && SZ_FRC_LEAN != 0) { forcing_lean_body<FRC_PLAIN>(S, P, bid, nblk, first); return; }
  extern __shared__ double tw_lds[];
  constexpr int FG = TW ? FRC_G : FRC_PLAIN;      // lanes per floe
  int lane = threadIdx.x % FG, wpb = blockDim.x / FG, wid = threadIdx.x / FG;
  // per floe of the workgroup: ptx[pmax], pty[pmax] | pkey[pmax], skey[FC_CAP] | pcode[pmax], scode[FC_CAP]
  double* const ptx_w = TW ? tw_lds + (size_t)wid * 2 * pmax : nullptr;
  double* const pty_w = TW ? ptx_w + pmax : nullptr;
  int* const pkey_w = TW ? (int*)(tw_lds + (size_t)TW_FPB * 2 * pmax) + (size_t)wid * (pmax + FC_CAP) : nullptr;
  int* const skey_w = TW ? pkey_w + pmax : nullptr;
  signed char* const pcode_w = TW ? (signed char*)((int*)(tw_lds + (size_t)TW_FPB * 2 * pmax) + (size_t)TW_FPB * (pmax + FC_CAP)) + (size_t)wid * (pmax + FC_CAP) : nullptr;
  signed char* const scode_w = TW ? pcode_w + pmax : nullptr;
  if (stopped(S)) return;
  int N = S.cnt[C_NOWN];
  int per_x = S.ekind[2] == 1, per_y = S.ekind[0] == 1;
  double cturn = cos(P.turn), sturn = sin(P.turn);
  const int vb0 = S.xcd_forcing ? xcd_contiguous_from(first + bid, first, nblk, (N + wpb - 1) / wpb) : bid;
  for (int i = vb0 < 0 ? N : vb0 * wpb + wid; i < N; i += nblk * wpb) {
    double cxf = S.cx[i], cyf = S.cy[i], u = S.u[i], v = S.v[i], xi = S.xi[i];
    forcing_wrap(S, i, cxf, cyf);
    double ca = S.trig[2 * i], sa = S.trig[2 * i + 1];   // cos(alpha), sin(alpha)
    double ma_ratio = S.mass[i] / S.area[i];
    int o = S.soff[i], ns = S.soff[i + 1] - o;
    if (TW) {
      if (ns > pmax) { if (lane == 0) atomicOr(&S.cnt[C_ERR], ERR_CAP_CELLS); ns = pmax; }
      gsync();
      for (int k = lane; k < ns; k += FG) pkey_w[k] = -1;
      gsync();
    }
    double tx = 0, ty = 0, ttrq = 0, th = 0; int np = 0;
    for (int k = lane; k < ns; k += FG) {
      double sxk = S.sx[o + k], syk = S.sy[o + k];
      double x = (ca * sxk - sa * syk) + cxf;
      double y = (sa * sxk + ca * syk) + cyf;
      bool inb = point_in_bounds(S, x, y, per_x, per_y);
      if (!inb) continue;
      np++;
      double xc = x - cxf, yc = y - cyf;
      // coupling.jl:1530-1537 forms rad * sin(atan(yc, xc)) and rad * cos(atan(yc, xc)) with rad = hypot(xc, yc): these ARE yc and xc
      // (to a couple of ulps, the reference's own round-off; atan(0, 0) = 0 gives 0 and rad = 0 likewise).  Using them directly
      // saves a square root, a division and eight multiplications per point -- a sixth of the kernel's instructions, and the
      // kernel is bound by them (SQ counters at 100 k floes: 59 M VALU instructions per launch = 102 of its 153 us)
      double up = u - xi * yc, vp = v + xi * xc;
      LatticeCell lc = lattice_cell(S, x, y, per_x, per_y);
      // the four corner nodes, two wide loads each (uo vo hf ua | va) instead of five 8-byte ones: the kernel is bound by
      // the number of scattered load instructions, not by bytes
      double n4[4][5];
      {
        const int oo[4] = { lc.o00, lc.o01, lc.o10, lc.o11 };
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const double4 a = *(const double4*)(S.nodes + (size_t)oo[q] * 8);
          n4[q][0] = a.x; n4[q][1] = a.y; n4[q][2] = a.z; n4[q][3] = a.w; n4[q][4] = S.nodes[(size_t)oo[q] * 8 + 4];
        }
      }
      const double omty = 1.0 - lc.ty, omtx = 1.0 - lc.tx;
      auto sample = [&](int f) {          // sample_field() on the values just read, the second product of each blend fused into the sum
        double c0 = fma(lc.ty, n4[1][f], omty * n4[0][f]);
        double c1 = fma(lc.ty, n4[3][f], omty * n4[2][f]);
        return fma(lc.tx, c1, omtx * c0);
      };
      double uatm = sample(3), vatm = sample(4);
      double du = uatm - up, dv = vatm - vp;
      double nrm = sqrt(du * du + dv * dv);
      double tax = P.rho_a * P.Cd_ia * nrm * du, tay = P.rho_a * P.Cd_ia * nrm * dv;
      double uocn = sample(0), vocn = sample(1);
      double hfl = sample(2);
      double duo = uocn - up, dvo = vocn - vp;
      double nrmo = sqrt(duo * duo + dvo * dvo);
      double tox = P.rho_o * P.Cd_io * nrmo * (cturn * duo - sturn * dvo);
      double toy = P.rho_o * P.Cd_io * nrmo * (sturn * duo + cturn * dvo);
      double tpx = -ma_ratio * P.fcor * vocn, tpy = ma_ratio * P.fcor * uocn;
      double fx = tax + tpx + tox, fy = tay + tpy + toy;
      tx += fx; ty += fy; ttrq += (-fx * yc + fy * xc); th += hfl;
      if (TW) {
        // find_center_cell_index (coupling.jl:466-470, 1-based) and shift_cell_idx (:1154-1178)
        int xidx = (int)floor((x - S.gx0) / S.gdx + 0.5) + 1, yidx = (int)floor((y - S.gy0) / S.gdy + 0.5) + 1;
        int sxi = xidx, syi = yidx;
        if (per_x) sxi = xidx < 1 ? xidx + S.Nx : (S.Nx < xidx ? xidx - S.Nx : xidx);
        if (per_y) syi = yidx < 1 ? yidx + S.Ny : (S.Ny < yidx ? yidx - S.Ny : yidx);
        if (sxi >= 1 && sxi <= S.Nx + 1 && syi >= 1 && syi <= S.Ny + 1) {
          pkey_w[k] = (sxi - 1) * (S.Ny + 1) + (syi - 1);
          ptx_w[k] = -tox; pty_w[k] = -toy;
          int cx3 = sxi == xidx ? 1 : (sxi > xidx ? 2 : 0), cy3 = syi == yidx ? 1 : (syi > yidx ? 2 : 0);
          pcode_w[k] = (signed char)(cx3 + 3 * cy3);
        }
      }
    }
    for (int d = FG / 2; d >= 1; d >>= 1) {
      tx += __shfl_xor(tx, d, FG); ty += __shfl_xor(ty, d, FG); ttrq += __shfl_xor(ttrq, d, FG); th += __shfl_xor(th, d, FG);
    }
    int npt = np;
    for (int d = FG / 2; d >= 1; d >>= 1) npt += __shfl_xor(npt, d, FG);
    if (lane == 0) {
      // no in-bounds point: the floe is marked for removal (coupling.jl:1507-1508).  The tag itself is
      // written by the integrate kernel: this kernel may run beside the collision kernels, which also
      // write status, and the reference applies the coupling result after them
      S.frc_remove[i] = npt == 0 ? 1 : 0;
      if (npt == 0) { if (S.step > 0 && (S.stop_on_tags || S.restart_on_tags)) S.cnt[C_FRCSTOP] = S.step; }      // (this step's integrator will tag the floe and end the batch: see C_FRCSTOP)
      else {
        double xcor = ma_ratio * P.fcor * v, ycor = ma_ratio * P.fcor * u;
        double totx = npt * xcor + tx, toty = -npt * ycor + ty;
        double area = S.area[i];
        S.fxOA[i] = totx / npt * area; S.fyOA[i] = toty / npt * area;
        S.trqOA[i] = ttrq / npt * area; S.hflx[i] = th / npt;
      }
    }
    if (TW) {
      gsync();
      // slots in order of first appearance (add_point!, coupling.jl:1336-1360, keeps one entry per floe and cell)
      int nslots = 0;
      const unsigned long long half = 0xffffffffull << (32 * ((threadIdx.x >> 5) & 1));
      for (int base = 0; base < ns; base += FG) {
        const int k = base + lane;
        bool first = false;
        if (k < ns && pkey_w[k] >= 0) {
          const int key = pkey_w[k];
          first = true;
          for (int s = 0; s < nslots && s < FC_CAP; s++) if (skey_w[s] == key) { first = false; break; }
          for (int j = base; first && j < k; j++) if (pkey_w[j] == key) first = false;
        }
        const unsigned long long mask = __ballot(first) & half;
        if (first) {
          const int slot = nslots + __popcll(mask & ((1ull << (threadIdx.x & 63)) - 1));
          if (slot < FC_CAP) { skey_w[slot] = pkey_w[k]; scode_w[slot] = pcode_w[k]; }
        }
        nslots += __popcll(mask);
        gsync();
      }
      if (nslots > FC_CAP) { if (lane == 0) atomicOr(&S.cnt[C_ERR], ERR_CAP_CELLS); nslots = FC_CAP; }
      // ordered sums: lane owns slots lane and lane + 32
      const int k0 = lane < nslots ? skey_w[lane] : -2, k1 = lane + 32 < nslots ? skey_w[lane + 32] : -2;
      double ax0 = 0, ay0 = 0, ax1 = 0, ay1 = 0; int n0 = 0, n1 = 0;
      bool f0 = true, f1 = true;
      for (int j = 0; j < ns; j++) {
        const int key = pkey_w[j];
        if (key == k0) { if (f0) { ax0 = ptx_w[j]; ay0 = pty_w[j]; f0 = false; } else { ax0 += ptx_w[j]; ay0 += pty_w[j]; } n0++; }
        if (key == k1) { if (f1) { ax1 = ptx_w[j]; ay1 = pty_w[j]; f1 = false; } else { ax1 += ptx_w[j]; ay1 += pty_w[j]; } n1++; }
      }
      const size_t fb = (size_t)i * FC_CAP;
      if (lane < nslots) { S.fc_key[fb + lane] = k0; S.fc_code[fb + lane] = scode_w[lane]; S.fc_tx[fb + lane] = ax0; S.fc_ty[fb + lane] = ay0; S.fc_n[fb + lane] = n0; }
      if (lane + 32 < nslots) { S.fc_key[fb + lane + 32] = k1; S.fc_code[fb + lane + 32] = scode_w[lane + 32]; S.fc_tx[fb + lane + 32] = ax1; S.fc_ty[fb + lane + 32] = ay1; S.fc_n[fb + lane + 32] = n1; }
      if (lane == 0) S.fc_cnt[i] = npt == 0 ? 0 : nslots;
      gsync();
    }
  }
}

// the constants of the loop (hoisted out of the floe loop by whoever runs it)
struct FrcConsts { int per_x, per_y; double cturn, sturn, ka, ko; };
__device__ __forceinline__ FrcConsts frc_consts(const State& S, const Params& P) {
  FrcConsts C; C.per_x = S.ekind[2] == 1; C.per_y = S.ekind[0] == 1; C.cturn = cos(P.turn); C.sturn = sin(P.turn);
  C.ka = P.rho_a * P.Cd_ia; C.ko = P.rho_o * P.Cd_io;
  return C;
}
// the forcings of ONE floe by the FG lanes of a group (calc_one_way_coupling!, coupling.jl:1486-1589): on return every lane holds npt, the number of
// in-bounds points, and -- if npt > 0 -- the floe's fxOA, fyOA, trqOA and hflx_factor.  Nothing is stored.
template <int FG>
__device__ __forceinline__ void forcing_lean_floe(const State& S, const Params& P, const FrcConsts& C, int i, int lane, int& npt, double& o_fx, double& o_fy, double& o_trq, double& o_hflx) {
  double cxf = S.cx[i], cyf = S.cy[i]; const double u = S.u[i], v = S.v[i], xi = S.xi[i];
  forcing_wrap(S, i, cxf, cyf);
  const double ca = S.trig[2 * i], sa = S.trig[2 * i + 1];   // cos(alpha), sin(alpha)
  const double area = S.area[i];
  const double ma_ratio = S.mass[i] / area;
  const double mf = ma_ratio * P.fcor;
  const int o = S.soff[i], ns = S.soff[i + 1] - o;
  double tx = 0, ty = 0, ttrq = 0, th = 0; int np = 0;
  const bool blocked = S.sxy != nullptr;      // (the blocked copy when it has been made: State::sxy)
  auto point = [&](int k) { return k >= ns ? make_double2(0.0, 0.0) : blocked ? S.sxy[o + k] : make_double2(S.sx[o + k], S.sy[o + k]); };
  double2 nxt = point(lane);
  for (int k = lane; k < ns; k += FG) {
    const double sxk = nxt.x, syk = nxt.y;
    nxt = point(k + FG);
    // (the point's coordinates, the in-bounds test and the cell: the expressions of the plain loop, bit for bit)
    const double x = (ca * sxk - sa * syk) + cxf;
    const double y = (sa * sxk + ca * syk) + cyf;
    if (!point_in_bounds(S, x, y, C.per_x, C.per_y)) continue;
    np++;
    const double xc = x - cxf, yc = y - cyf;      // (rad sin / rad cos of coupling.jl:1530-1537: see the plain loop)
    const double up = fma(-xi, yc, u), vp = fma(xi, xc, v);
    const LatticeCell lc = lattice_cell(S, x, y, C.per_x, C.per_y);
    double n4[4][5];
    {
      const int oo[4] = { lc.o00, lc.o01, lc.o10, lc.o11 };
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double4 a = *(const double4*)(S.nodes + (size_t)oo[q] * 8);
        n4[q][0] = a.x; n4[q][1] = a.y; n4[q][2] = a.z; n4[q][3] = a.w; n4[q][4] = S.nodes[(size_t)oo[q] * 8 + 4];
      }
    }
    const double omty = 1.0 - lc.ty, omtx = 1.0 - lc.tx;
    const double w00 = omtx * omty, w01 = omtx * lc.ty, w10 = lc.tx * omty, w11 = lc.tx * lc.ty;
    auto sample = [&](int f) { return fma(w11, n4[3][f], fma(w10, n4[2][f], fma(w01, n4[1][f], w00 * n4[0][f]))); };
    const double uatm = sample(3), vatm = sample(4), uocn = sample(0), vocn = sample(1), hfl = sample(2);
    const double du = uatm - up, dv = vatm - vp, duo = uocn - up, dvo = vocn - vp;
    const double qa = C.ka * sqrt_fast(fma(du, du, dv * dv)), qo = C.ko * sqrt_fast(fma(duo, duo, dvo * dvo));
    const double fx = fma(qa, du, fma(qo, fma(C.cturn, duo, -(C.sturn * dvo)), -(mf * vocn)));
    const double fy = fma(qa, dv, fma(qo, fma(C.sturn, duo, C.cturn * dvo), mf * uocn));
    tx += fx; ty += fy; th += hfl;
    ttrq += fma(fy, xc, -(fx * yc));
  }
  for (int d = FG / 2; d >= 1; d >>= 1) {
    tx += __shfl_xor(tx, d, FG); ty += __shfl_xor(ty, d, FG); ttrq += __shfl_xor(ttrq, d, FG); th += __shfl_xor(th, d, FG);
  }
  npt = np;
  for (int d = FG / 2; d >= 1; d >>= 1) npt += __shfl_xor(npt, d, FG);
  o_fx = o_fy = o_trq = o_hflx = 0.0;
  if (npt != 0) {
    const double xcor = ma_ratio * P.fcor * v, ycor = ma_ratio * P.fcor * u;
    const double totx = npt * xcor + tx, toty = -npt * ycor + ty;
    o_fx = totx / npt * area; o_fy = toty / npt * area; o_trq = ttrq / npt * area; o_hflx = th / npt;
  }
}
template <int FG>
__device__ __forceinline__ void forcing_lean_body(State& S, const Params& P, int bid, int nblk, int first) {
  const int lane = threadIdx.x % FG, wpb = blockDim.x / FG, wid = threadIdx.x / FG;
  if (stopped(S)) return;
  const int N = S.cnt[C_NOWN];
  const FrcConsts C = frc_consts(S, P);
  const int vb0 = S.xcd_forcing ? xcd_contiguous_from(first + bid, first, nblk, (N + wpb - 1) / wpb) : bid;
  for (int i = vb0 < 0 ? N : vb0 * wpb + wid; i < N; i += nblk * wpb) {
    int npt; double fx, fy, trq, hf;
    forcing_lean_floe<FG>(S, P, C, i, lane, npt, fx, fy, trq, hf);
    if (lane == 0) {
      // (no in-bounds point: marked for removal, coupling.jl:1507-1508 -- see the plain loop)
      S.frc_remove[i] = npt == 0 ? 1 : 0;
      if (npt == 0) { if (S.step > 0 && (S.stop_on_tags || S.restart_on_tags)) S.cnt[C_FRCSTOP] = S.step; }
      else { S.fxOA[i] = fx; S.fyOA[i] = fy; S.trqOA[i] = trq; S.hflx[i] = hf; }
    }
  }
}

template <bool TW>
__global__ void __launch_bounds__(256) sz_k_forcing(State S, Params P, int pmax) { forcing_body<TW>(S, P, blockIdx.x, gridDim.x, pmax, 0); }

// dynamic LDS of sz_k_forcing<true> for pmax points per floe
inline size_t tw_forcing_lds(int pmax) { return (size_t)TW_FPB * ((size_t)2 * pmax * sizeof(double) + (size_t)(pmax + FC_CAP) * (sizeof(int) + 1)); }

// ---------------------------------------------------------------- mixed precision (BASELINE configs[4])
// The forcings are sums of ~100 smooth per-point terms per floe: they do not need fp64 per point.  In mixed mode
// the point's offset from the centroid, its velocity, the bilinear interpolation and the stresses are fp32 (the
// sub-floe points and the lattice are kept as fp32 copies: half the bytes, and sqrt / reciprocal are a handful of
// instructions instead of dozens); the absolute position -- needed for the in-bounds test and the lattice cell
// -- and the per-floe totals stay fp64.  Everything else (contacts, integrator) is unchanged fp64.  The
// reference has no Float32 answers to match (documentation.md:25: only Float64 is tested and supported): the
// mixed path is held to the fp64 path with a stated tolerance (tests/test_hip_parity.py::test_mixed_precision).
// blocked copy of the sub-floe points (State::sxy): one wavefront per floe ranks the floe's points by the Morton key of their body-frame
// coordinates (quantum q: a quarter of the lattice spacing) -- a counting rank, ties by index: a permutation whatever the keys are
constexpr int BLK_CAP = 2048;
__device__ __forceinline__ unsigned morton16(unsigned x, unsigned y) {
  auto spread = [](unsigned v) { v &= 0xffffu; v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u; return v; };
  return spread(x) | (spread(y) << 1);
}
__global__ void __launch_bounds__(64) sz_k_block_points(State S, int N, double rq) {
  __shared__ unsigned key[BLK_CAP];
  for (int i = blockIdx.x; i < N; i += gridDim.x) {
    const int o = S.soff[i], ns = S.soff[i + 1] - o;
    if (ns > BLK_CAP) {          // (a floe with more points than the sort holds keeps its order)
      for (int k = threadIdx.x; k < ns; k += 64) S.sxy[o + k] = make_double2(S.sx[o + k], S.sy[o + k]);
      continue;
    }
    for (int k = threadIdx.x; k < ns; k += 64) {
      const double fx = floor(S.sx[o + k] * rq) + 32768.0, fy = floor(S.sy[o + k] * rq) + 32768.0;
      const unsigned qx = (unsigned)fmin(fmax(fx, 0.0), 65535.0), qy = (unsigned)fmin(fmax(fy, 0.0), 65535.0);
      key[k] = morton16(qx, qy);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ns; k += 64) {
      const unsigned mine = key[k];
      int r = 0;
      for (int j = 0; j < ns; j++) { const unsigned kj = key[j]; r += (kj < mine || (kj == mine && j < k)) ? 1 : 0; }
      S.sxy[o + r] = make_double2(S.sx[o + k], S.sy[o + k]);
    }
    __syncthreads();
  }
}
__global__ void sz_k_to_f32_points(State S, int n) {
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) S.s32[q] = make_float2((float)S.sx[q], (float)S.sy[q]);
}
__global__ void sz_k_to_f32_nodes(State S) {
  int n = (S.Nx + 1) * (S.Ny + 1) * 8;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) S.nodes32[q] = (float)S.nodes[q];
}
__device__ __forceinline__ float sample_field32(const float* nodes, int f, const LatticeCell& c, float tx, float ty) {
  float c0 = (1.0f - ty) * nodes[(size_t)c.o00 * 8 + f] + ty * nodes[(size_t)c.o01 * 8 + f];
  float c1 = (1.0f - ty) * nodes[(size_t)c.o10 * 8 + f] + ty * nodes[(size_t)c.o11 * 8 + f];
  return (1.0f - tx) * c0 + tx * c1;
}
__device__ __forceinline__ void forcing_mixed_body(State& S, const Params& P, int bid, int nblk, int first) {
  if (stopped(S)) return;
  int N = S.cnt[C_NOWN];
  int lane = threadIdx.x % FRC_PLAIN, wpb = blockDim.x / FRC_PLAIN, wid = threadIdx.x / FRC_PLAIN;
  int per_x = S.ekind[2] == 1, per_y = S.ekind[0] == 1;
  const float cturn = (float)cos(P.turn), sturn = (float)sin(P.turn);
  const float ka = (float)(P.rho_a * P.Cd_ia), ko = (float)(P.rho_o * P.Cd_io);
  const int vb0 = S.xcd_forcing ? xcd_contiguous_from(first + bid, first, nblk, (N + wpb - 1) / wpb) : bid;
  for (int i = vb0 < 0 ? N : vb0 * wpb + wid; i < N; i += nblk * wpb) {
    double cxf = S.cx[i], cyf = S.cy[i]; const double u = S.u[i], v = S.v[i];
    forcing_wrap(S, i, cxf, cyf);
    const float uf = (float)u, vf = (float)v, xif = (float)S.xi[i];
    const float ca = (float)S.trig[2 * i], sa = (float)S.trig[2 * i + 1];
    const double ma_ratio = S.mass[i] / S.area[i];
    const float mf = (float)(ma_ratio * P.fcor);
    int o = S.soff[i], ns = S.soff[i + 1] - o;
    float tx = 0, ty = 0, ttrq = 0, th = 0; int np = 0;
    for (int k = lane; k < ns; k += FRC_PLAIN) {
      const float2 sp = S.s32[o + k];
      const float px = ca * sp.x - sa * sp.y, py = sa * sp.x + ca * sp.y;      // offset from the centroid
      const double x = cxf + (double)px, y = cyf + (double)py;
      bool inb = point_in_bounds(S, x, y, per_x, per_y);
      if (!inb) continue;
      np++;
      const float up = uf - xif * py, vp = vf + xif * px;            // (rad * sin / cos of the point's angle are py and px: see the fp64 kernel)
      const LatticeCell lc = lattice_cell(S, x, y, per_x, per_y);
      const float wx = (float)lc.tx, wy = (float)lc.ty;
      const float uatm = sample_field32(S.nodes32, 3, lc, wx, wy), vatm = sample_field32(S.nodes32, 4, lc, wx, wy);
      const float du = uatm - up, dv = vatm - vp;
      const float nrm = sqrtf(du * du + dv * dv);
      const float tax = ka * nrm * du, tay = ka * nrm * dv;
      const float uocn = sample_field32(S.nodes32, 0, lc, wx, wy), vocn = sample_field32(S.nodes32, 1, lc, wx, wy);
      const float hfl = sample_field32(S.nodes32, 2, lc, wx, wy);
      const float duo = uocn - up, dvo = vocn - vp;
      const float nrmo = sqrtf(duo * duo + dvo * dvo);
      const float tox = ko * nrmo * (cturn * duo - sturn * dvo), toy = ko * nrmo * (sturn * duo + cturn * dvo);
      const float tpx = -mf * vocn, tpy = mf * uocn;
      const float fx = tax + tpx + tox, fy = tay + tpy + toy;
      tx += fx; ty += fy; ttrq += (-fx * py + fy * px); th += hfl;
    }
    double dtx = tx, dty = ty, dtq = ttrq, dth = th;
    for (int d = FRC_PLAIN / 2; d >= 1; d >>= 1) {
      dtx += __shfl_xor(dtx, d, FRC_PLAIN); dty += __shfl_xor(dty, d, FRC_PLAIN); dtq += __shfl_xor(dtq, d, FRC_PLAIN); dth += __shfl_xor(dth, d, FRC_PLAIN);
    }
    int npt = np;
    for (int d = FRC_PLAIN / 2; d >= 1; d >>= 1) npt += __shfl_xor(npt, d, FRC_PLAIN);
    if (lane == 0) {
      S.frc_remove[i] = npt == 0 ? 1 : 0;
      if (npt == 0 && S.step > 0 && (S.stop_on_tags || S.restart_on_tags)) S.cnt[C_FRCSTOP] = S.step;
      if (npt != 0) {
        double xcor = ma_ratio * P.fcor * v, ycor = ma_ratio * P.fcor * u;
        double totx = npt * xcor + dtx, toty = -npt * ycor + dty;
        double area = S.area[i];
        S.fxOA[i] = totx / npt * area; S.fyOA[i] = toty / npt * area;
        S.trqOA[i] = dtq / npt * area; S.hflx[i] = dth / npt;
      }
    }
  }
}

__global__ void __launch_bounds__(256, 6) sz_k_forcing_mixed(State S, Params P) { forcing_mixed_body(S, P, blockIdx.x, gridDim.x, 0); }

// Horizontal fusion: the neighbour search and the forcings are independent of each other (the forcings only need
// the state the previous step left) and both are latency-bound per-floe kernels of ~20 us that leave most of the
// chip idle; a second stream would cost ~10 us of fork/join.  One launch: workgroups [0, nb_neigh) search neighbours,
// the rest evaluate the forcings (FRC 1: fp64, 2: mixed precision).  Measured at 10 k floes: 0.180 -> 0.168 ms/step.
template <int FRC, bool REC = false>
__global__ void __launch_bounds__(256) sz_k_neighbors_forcing(State S, Params P, int nb_neigh) {
  if ((int)blockIdx.x < nb_neigh) neighbors_body<256, false, MAXNB, REC>(S, blockIdx.x, nb_neigh);      // (fields of 30 k floes and more: the lean instantiation)
  else if (FRC == 1) forcing_body<false>(S, P, (int)blockIdx.x - nb_neigh, (int)gridDim.x - nb_neigh, 0, nb_neigh);
  else forcing_mixed_body(S, P, (int)blockIdx.x - nb_neigh, (int)gridDim.x - nb_neigh, nb_neigh);
}

// ============================================================================ rigid-body update (A12)
__device__ __forceinline__ double sgn(double x) { return (double)((x > 0) - (x < 0)); }

// calc_stress!, update_floe.jl:392-414, and _update_stress_accum!, stress_calculators.jl:118-122
__device__ __forceinline__ void floe_stress(State& S, const Params& P, int i, double cx, double cy) {
  double s11 = 0, s12 = 0, s21 = 0, s22 = 0;
  int rn = S.inter_cnt[i];
  if (rn > 0) {
    for (int k = 0; k < rn; k++) {
      const double* r = S.inter_rows + ((size_t)i * S.rowcap + k) * 7;
      s11 += (r[3] - cx) * r[1];
      s12 += (r[4] - cy) * r[1] + (r[3] - cx) * r[2];
      s22 += (r[4] - cy) * r[2];
    }
    s12 *= 0.5; s21 = s12;
    double sc = 1 / (S.area[i] * S.height[i]);
    s11 *= sc; s12 *= sc; s21 *= sc; s22 *= sc;
  }
  double l = P.lambda, s[4] = { s11, s12, s21, s22 };
  for (int k = 0; k < 4; k++) { S.sa[i * 4 + k] = (1 - l) * S.sa[i * 4 + k] + l * s[k]; S.si[i * 4 + k] = s[k]; }
}
// calc_stress! on its own (the reference's tests call it on hand-made interaction matrices)
__global__ void sz_k_calc_stress(State S, Params P) {
  int N = S.cnt[C_NOWN];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) floe_stress(S, P, i, S.cx[i], S.cy[i]);
}

// ---- halo records of tiled runs (multi-GPU; the exchange itself: "halo exchange" below).  One record per floe sent to another
// rank: the columns the collision path reads + the ring.
constexpr int HALO_RING = 32;                  // the smallest ring capacity of a record (State::halo_ring: what this context's records have room for)
constexpr int HALO_REC = 12 + 2 * HALO_RING;   // doubles, at that smallest capacity
__device__ __host__ __forceinline__ int halo_ring_of(const State& S) { return S.halo_ring > HALO_RING ? S.halo_ring : HALO_RING; }
__device__ __host__ __forceinline__ int halo_rec(const State& S) { return 12 + 2 * halo_ring_of(S); }
constexpr int ERR_HALO_DRIFT = 16384;
// the ranks whose (already expanded) box holds the centroid or one of its periodic images, as a bit set
__device__ __forceinline__ unsigned long long halo_hits(const double* boxes, int nranks, int me, double Lx, double Ly, int per_x, int per_y, double cx, double cy) {
  unsigned long long hits = 0;
  for (int d = 0; d < nranks; d++) {
    if (d == me) continue;
    const double* b = boxes + 4 * d;
    bool hit = false;
    for (int kx = (per_x ? -1 : 0); kx <= (per_x ? 1 : 0) && !hit; kx++)
      for (int ky = (per_y ? -1 : 0); ky <= (per_y ? 1 : 0) && !hit; ky++) {
        double x = cx + kx * Lx, y = cy + ky * Ly;
        hit = (b[0] <= x && x <= b[1] && b[2] <= y && y <= b[3]);
      }
    if (hit) hits |= 1ull << d;
  }
  return hits;
}
// the workgroup that finishes a pack last (threads 0 .. nranks - 1 and thread 0): the header record of every region, the totals, the scratch
// words (counts[64 + d]: running totals, [128]: ticket, [129]: drift) zero again for the next pack
__device__ __forceinline__ void halo_headers(const State& S, int nranks, double* send, int cap, int* counts, const int* dcap, bool drift) {
  int* run = counts + 64;
  if ((int)threadIdx.x < nranks) {
    int d = threadIdx.x;
    const int tot = atomicAdd(&run[d], 0);
    run[d] = 0;
    counts[d] = tot;
    const int room = dcap ? dcap[d] : cap;
    if (send) {
      // header record: [0] the count, [1] this rank's stop request (resident batches end after the first step that tags a floe,
      // on EVERY rank: sz_k_halo_unpack_inline reads the flags of all ranks before the next step does anything)
      // (the stop word may have been raised inside the launch that packs -- the integrator, on another XCD: read where it was written, past
      //  this XCD's L2)
      double* hdr = send + (size_t)d * (cap + 1) * halo_rec(S);
      hdr[0] = (double)(tot < room ? tot : room); hdr[1] = (double)__hip_atomic_load(&S.cnt[C_STOP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      hdr[2] = (double)S.cnt[C_RETRYSTOP];      // ... and its pause (the step whose narrow phase met an item for the variant that was left out, or outgrew a list)
    }
  }
  if (threadIdx.x == 0) {
    if (drift) S.cnt[C_DRIFT] = atomicAdd(&run[65], 0);          // what the host sizes the next gather interval with
    run[65] = 0; run[64] = 0;
  }
}
// what sz_k_halo_pack is given, as one kernel argument: the integrator of a tiled step writes the halo records of the NEXT step itself
// (sz_k_integrate<true, true>) -- the thread that has just placed a floe holds everything a record carries
struct PackInl {
  double* send; const double* boxes; const int* dcap; const double* ref; int* counts;
  double Lx, Ly, margin;
  int nranks, me, cap, per_x, per_y;
};

// one thread per floe: stress, guards, thermodynamics, AB2 velocity update; stores the motion.
// MOVE (resident steps, rings of at most MV_RING points): the same thread also moves the ring, refreshes its
// box, evaluates the strain and bins the floe -- what sz_k_move_strain does with 16 lanes per floe in a
// second launch.  The ring is read into registers in one go (one memory round trip), the per-edge strain terms
// are the same expressions summed in the same order.
template <bool MOVE, bool PACK = false>
// gl_fill (resident steps): the ghost-candidate list to append to (sz_k_ghost_list), -1: none
// ginl: the allocator to make the next step's ghosts in (inline ghosts), -1: none
// PACK (tiled steps, with MOVE): the thread also writes the floe's halo records for the next step (PK: what sz_k_halo_pack is given) -- the
// pack launch of that step and its place in the chain of dependent launches go away
#ifdef SZ_STAMPS
// diagnostic build: clock of the thread that updates floe SZ_ISTAMP_FLOE at a few points of the integrator (stamps[900 + k]);
// tools/integrate_stamps.py prints them
#ifndef SZ_ISTAMP_FLOE
#define SZ_ISTAMP_FLOE 99
#endif
#define ISTAMP(k) do { if (MOVE && i == SZ_ISTAMP_FLOE) S.stamps[900 + (k)] = clock64(); } while (0)
#else
#define ISTAMP(k) do {} while (0)
#endif
// acc_mode (resident batches): bit 0: the collision totals, the stress sums and the status tags come from the fixed-point words the narrow
// phase has accumulated (State::facc) -- no reduce launch ran in this step; bit 1: the host knows this is the batch's last step
__global__ void __launch_bounds__(MOVE ? 128 : 256) sz_k_integrate(State S, Params P, int dt, int apply_frc, int bin, int nh, int gl_fill, int ginl, PackInl PK, int acc_mode) {
  const GridGeo geo = grid_geo(S);
  const StopRegs stop = stop_load(S);
  const bool use_acc = (acc_mode & 1) != 0;
  const int frcstop = use_acc ? S.cnt[C_FRCSTOP] : 0;
  const int N = nh >= 0 ? nh : S.cnt[C_NOWN];     // nh: see sz_k_ghost_flag_scan
  const int nv0 = MOVE && ginl >= 0 ? S.voff[N] : 0;       // ring points of the parents (inline ghosts are laid out behind them)
  int wh = 0, wf = 0, wv = 0, wx = 0;
  const double wall[4] = { S.eval[0], S.eval[1], S.eval[2], S.eval[3] };
  bool tested = false;
  double pk_drift = 0.0;
  // the batch's last step -- the host says so, or a tag of this step has already asked for the stop (narrow phase: C_STOP; forcings: C_FRCSTOP):
  // the step's ghosts stay attached and none are made for a next step, so that the rows of THIS step can still be assembled afterwards
  // (sz_k_inter_fill runs once, behind the batch), and the parents' old centroids are kept for it in `mot`
  const bool last_step = use_acc && ((acc_mode & 2) != 0 || (S.step > 0 && S.stop_on_tags && (stop.s == S.step || frcstop == S.step)));
  if (last_step) { gl_fill = -1; ginl = -1; }
  if (use_acc && blockIdx.x == 0 && threadIdx.x < 2 * NSEG && !stop_test_late(S, stop)) S.wq[(threadIdx.x >> 1) * 32 + (threadIdx.x & 1)] = 0;      // the narrow phase has consumed the work list (as sz_k_inter_fill does)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    ISTAMP(0);
    const bool gl_any = gl_fill >= 0 || ginl >= 0;
    const int st0 = gl_any || PACK || use_acc ? S.status[i] : SZ_ACTIVE, ngh0 = gl_any || use_acc ? S.ngh[i] : 0;
    const double rmx = (MOVE && (gl_any || S.rec32 || PACK)) || use_acc ? S.rmax[i] : 0.0;
    longlong4 fa0 = make_longlong4(0, 0, 0, 0), fa1 = fa0, fa2 = fa0, fa3 = fa0;
    if (use_acc) { const longlong4* a = (const longlong4*)(S.facc + (size_t)i * FX_WORDS); fa0 = a[0]; fa1 = a[1]; fa2 = a[2]; fa3 = a[3]; }
    const double pk_rx = PACK && PK.ref ? PK.ref[i] : 0.0, pk_ry = PACK && PK.ref ? PK.ref[S.capM + i] : 0.0;      // where the floe lay when the boxes were gathered
    // what a ghost copies of its parent beyond the update's own operands (inline ghosts): asked for HERE, with the first batch -- inside the
    // ghost branch these four loads were a dependent round trip behind the thread's ~60 stores (6.7 k cycles of the 21 k the ghost cost)
    const bool ghost_ops = MOVE && (ginl >= 0 || PACK);
    const long long g_id = ghost_ops ? S.id[i] : 0, g_oki = ghost_ops ? S.okey[i] : 0;
    double g_over = ghost_ops || use_acc ? S.overarea[i] : 0.0;
    const signed char g_os = ghost_ops ? S.osign[i] : (signed char)1;
    // Memory order is the whole cost of this kernel (a store in between keeps the compiler from hoisting the
    // loads behind it, and every batch of loads is one HBM round trip): everything is read first -- the
    // columns, then what their values address (contact rows, ring) -- then computed, then stored.
    const int frc_rm = apply_frc ? S.frc_remove[i] : 0;
    double cfx = use_acc ? 0.0 : S.cfx[i], cfy = use_acc ? 0.0 : S.cfy[i], ctrq = use_acc ? 0.0 : S.ctrq[i];
    const double cx = S.cx[i], cy = S.cy[i];
    const int rn = use_acc ? 0 : S.inter_cnt[i];
    const double area = S.area[i], height0 = S.height[i], mass0 = S.mass[i], moment0 = S.moment[i], hflx = S.hflx[i];
    const double u = S.u[i], v = S.v[i], xi = S.xi[i], alpha0 = S.alpha[i];
    const double p_dxdt = S.p_dxdt[i], p_dydt = S.p_dydt[i], p_dalphadt = S.p_dalphadt[i];
    const double p_dudt = S.p_dudt[i], p_dvdt = S.p_dvdt[i], p_dxidt = S.p_dxidt[i];
    const double fxOA = S.fxOA[i], fyOA = S.fyOA[i], trqOA = S.trqOA[i];
    const double4 sa4 = *(const double4*)(S.sa + (size_t)i * 4);
    const double sa0[4] = { sa4.x, sa4.y, sa4.z, sa4.w };
    const int o = MOVE ? ring_off(S, i) : 0, n = MOVE ? ring_n(S, i) : 0;
    if (!tested) {             // (the stop test after the first batch of loads has gone out, before anything is stored)
      loads_issued();
      if (stop_test_late(S, stop)) break;
      tested = true;
    }
    ISTAMP(1);
    const bool body = MOVE && S.body_rings;           // the ring in its body frame (fp32), pose in fp64: nothing to rewrite
    double px[MOVE ? MV_RING : 1], py[MOVE ? MV_RING : 1];
    if (MOVE) {
      if (body) {
#pragma unroll
        for (int k = 0; k < MV_RING; k++) { const float2 b = k < n ? S.ring32[o + k] : make_float2(0.f, 0.f); px[k] = (double)b.x; py[k] = (double)b.y; }
      } else {
#pragma unroll
        for (int k = 0; k < MV_RING; k++) { const double2 p = k < n ? S.vxy[o + k] : make_double2(0.0, 0.0); px[k] = p.x; py[k] = p.y; }
      }
    }
    // where the centroid goes depends on the old velocities alone (AB2): the counter of the cell it lands in is drawn now, and its
    // answer -- a round trip -- is used with the last stores instead of being waited for there
    const double dx = 1.5 * dt * u - 0.5 * dt * p_dxdt;
    const double dy = 1.5 * dt * v - 0.5 * dt * p_dydt;
    int cell_c = 0, cell_s = 0;
    if (MOVE && bin) { int ix, iy; cell_of(geo, cx + dx, cy + dy, ix, iy); cell_c = iy * geo.ncx + ix; cell_s = atomicAdd(&S.cell_cnt[cell_c], 1); }
    // calc_stress! (update_floe.jl:392-414): as floe_stress(), on the values read above
    double s11 = 0, s12 = 0, s21 = 0, s22 = 0;
    int st_new = st0; bool st_dirty = false, over_dirty = false;
    if (use_acc) {
      // the totals of this floe's rows -- and of its ghosts' rows: the ghost fold of collisions.jl:830-850 is an integer addition here,
      // a ghost's levers were taken about ITS centroid, which is the parent's shifted by the same vector as the contact points --
      // as the narrow phase has accumulated them (fx_row): collision_force / collision_trq (:747-749, 852-861), overarea (:304), the
      // stress sums, and the status tags in the reference's order (:367, 438, 525, then the mirror pass :801-806)
      long long q[7] = { fa0.x, fa0.y, fa0.z, fa0.w, fa1.x, fa1.y, fa1.z }, ql[7] = { fa2.x, fa2.y, fa2.z, fa2.w, fa3.x, fa3.y, fa3.z };      // ql: the low words
      const int tagb = (int)(fa1.w & 0xffffffffll);
      if (ngh0 != 0) {
        for (int g3 = 0; g3 < MAX_GHOSTS; g3++) {
          const int g = S.gh[i * MAX_GHOSTS + g3];
          if (g >= 0) {
            const longlong4* a = (const longlong4*)(S.facc + (size_t)g * FX_WORDS); const longlong4 b0 = a[0], b1 = a[1], b2 = a[2], b3 = a[3];
            q[0] += b0.x; q[1] += b0.y; q[2] += b0.z; q[3] += b0.w; q[4] += b1.x; q[5] += b1.y; q[6] += b1.z;
            ql[0] += b2.x; ql[1] += b2.y; ql[2] += b2.z; ql[3] += b2.w; ql[4] += b3.x; ql[5] += b3.y; ql[6] += b3.z;
          }
        }
      }
      const int eF = fx_force_exp(S.kexp, area, height0), eT = eF + fx_lever_exp(rmx);
      cfx = fx_join(q[0], ql[0], eF); cfy = fx_join(q[1], ql[1], eF); ctrq = fx_join(q[4] - q[3], ql[4] - ql[3], eT);
      S.cfx[i] = cfx; S.cfy[i] = cfy; S.ctrq[i] = ctrq;          // (the totals as summed: the force guard below works on copies, as before)
      if ((q[0] | q[1] | q[2] | q[3] | q[4] | q[5] | ql[0] | ql[1] | ql[2] | ql[3] | ql[4] | ql[5]) != 0) {
        const double sc = 1 / (area * height0);
        s11 = fx_join(q[2], ql[2], eT) * sc; s12 = fx_join(q[3] + q[4], ql[3] + ql[4], eT) * 0.5 * sc; s21 = s12; s22 = fx_join(q[5], ql[5], eT) * sc;
      }
      g_over = g_over + fx_join(q[6], ql[6], fx_area_exp(area)); over_dirty = (q[6] | ql[6]) != 0;
      if (tagb & 1) st_new = SZ_FUSE;
      if (tagb & 2) st_new = SZ_REMOVE;
      if (tagb & 4) st_new = SZ_FUSE;
      st_dirty = tagb != 0 || st0 != SZ_ACTIVE;
    } else
    if (rn > 0) {
      for (int k = 0; k < rn; k++) {
        const double* r = S.inter_rows + ((size_t)i * S.rowcap + k) * 7;
        s11 += (r[3] - cx) * r[1];
        s12 += (r[4] - cy) * r[1] + (r[3] - cx) * r[2];
        s22 += (r[4] - cy) * r[2];
      }
      s12 *= 0.5; s21 = s12;
      const double sc = 1 / (area * height0);
      s11 *= sc; s12 *= sc; s21 *= sc; s22 *= sc;
    }
    const double l = P.lambda, sv[4] = { s11, s12, s21, s22 };
    ISTAMP(2);
    double hh = height0;
    if (hh > P.max_h) { hh = P.max_h; wh++; }
    double mass = mass0;
    for (int it = 0; it < 400 && fmax(fabs(cfx), fabs(cfy)) > mass / (5 * dt); it++) { cfx = cfx / 10; cfy = cfy / 10; ctrq = ctrq / 10; wf++; }
    double h = hh;
    double dh = hflx / h;
    double hfrac = (h + dh) / h;
    mass *= hfrac; double moment = moment0 * hfrac; h -= dh;
    double da = 1.5 * dt * xi - 0.5 * dt * p_dalphadt;
    const double al = alpha0 + da;
    double cal, sal, cda, sda;                              // cos / sin alpha for the forcing kernel (32 lanes per floe: not the place for it)
    sincos(al, &sal, &cal); sincos(da, &sda, &cda);         // (one argument reduction for each pair)
    double dudt = (fxOA + cfx) / mass, dvdt = (fyOA + cfy) / mass;
    double frac = 1.0, au = fabs(dt * dudt), av = fabs(dt * dvdt), h2 = h / 2;
    if (au > h2 && av > h2) {
      double f1 = (sgn(dudt) * h / (2 * dt)) / dudt, f2 = (sgn(dvdt) * h / (2 * dt)) / dvdt;
      frac = f1 < f2 ? f1 : f2;
    } else if (au > h2 && av < h2) frac = (sgn(dudt) * h / (2 * dt)) / dudt;
    else if (au < h2 && av > h2) frac = (sgn(dvdt) * h / (2 * dt)) / dvdt;
    if (frac != 1) { dudt = frac * dudt; dvdt = frac * dvdt; wv++; }
    const double nu = u + (1.5 * dt * dudt - 0.5 * dt * p_dudt);
    const double nv = v + (1.5 * dt * dvdt - 0.5 * dt * p_dvdt);
    double dxidt = (trqOA + ctrq) / moment;
    dxidt = frac * dxidt;
    double nxi = xi + 1.5 * dt * dxidt - 0.5 * dt * p_dxidt;
    if (fabs(nxi) > P.max_xi) { nxi = sgn(nxi) * P.max_xi; wx++; }
    ISTAMP(3);
    // ---- stores.  First the floe's cell entry: its counter was drawn long ago, and a wait for an answer AFTER the stores below would
    // be a wait for every one of them (loads and stores return in order only among their own kind: the counter says "all done")
    if (MOVE && bin) {               // as cell_insert()
      if (cell_s < CELL_K) S.cell_slots[(size_t)cell_c * CELL_K + cell_s] = i;
      else S.cell_items[i] = atomicExch(&S.cell_ovf[cell_c], i + 1) - 1;
    }
    if (use_acc) {
      if (over_dirty) S.overarea[i] = g_over;
      if ((fa0.x | fa0.y | fa0.z | fa0.w | fa1.x | fa1.y | fa1.z | fa1.w | fa2.x | fa2.y | fa2.z | fa2.w | fa3.x | fa3.y | fa3.z) != 0) {
        longlong4* a = (longlong4*)(S.facc + (size_t)i * FX_WORDS); a[0] = make_longlong4(0, 0, 0, 0); a[1] = make_longlong4(0, 0, 0, 0); a[2] = make_longlong4(0, 0, 0, 0); a[3] = make_longlong4(0, 0, 0, 0);
      }
      if (st_dirty) S.status[i] = st_new;          // (tagA -- the status before the mirror pass -- is written for every row by the launch that assembles the rows)
      if (MOVE && last_step) *(double2*)(S.mot + (size_t)i * 4) = make_double2(cx, cy);      // (sz_k_inter_fill behind the batch: levers and ghost shifts of this step)
    }
    if (frc_rm || (use_acc && st_new != SZ_ACTIVE)) {
      if (frc_rm) { S.status[i] = SZ_REMOVE; st_new = SZ_REMOVE; }
      if (PACK) {            // the workgroup that writes the halo headers at the end of THIS launch must see the request: performed at memory and waited for
        if (S.step > 0 && S.stop_on_tags) { const int was = atomicMax(&S.cnt[C_STOP], S.step); asm volatile("" :: "v"(was)); }
      } else request_stop(S);
    }
    // the ghosts of this step are detached here (nothing after the reduce looks at them): the next step's ghost pass
    // then only visits the parents that get new ones
    if (ngh0 != 0 && !last_step) { S.ngh[i] = 0; for (int q = 0; q < MAX_GHOSTS; q++) S.gh[i * MAX_GHOSTS + q] = -1; }
    // (a wavefront holds at most 63 memory operations in flight, and this thread issues ~70 stores: the four-component columns go out
    //  as one 32-byte store each)
    *(double4*)(S.sa + (size_t)i * 4) = make_double4((1 - l) * sa0[0] + l * sv[0], (1 - l) * sa0[1] + l * sv[1], (1 - l) * sa0[2] + l * sv[2], (1 - l) * sa0[3] + l * sv[3]);
    *(double4*)(S.si + (size_t)i * 4) = make_double4(sv[0], sv[1], sv[2], sv[3]);
    S.mass[i] = mass; S.moment[i] = moment; S.height[i] = h;
    S.alpha[i] = al;
    *(double2*)(S.trig + (size_t)i * 2) = make_double2(cal, sal);
    if (!MOVE) *(double4*)(S.mot + (size_t)i * 4) = make_double4(dx, dy, cda, sda);
    S.p_dxdt[i] = u; S.p_dydt[i] = v; S.p_dalphadt[i] = xi;
    S.u[i] = nu; S.v[i] = nv;
    S.p_dudt[i] = dudt; S.p_dvdt[i] = dvdt;
    S.xi[i] = nxi; S.p_dxidt[i] = dxidt;
    if (MOVE) {
      // _move_floe! (floe_utils.jl:82-93) and calc_strain! (update_floe.jl:425-453) with the new velocities
      const double ncx = cx + dx, ncy = cy + dy;
      double e11 = 0, e12 = 0, e22 = 0;
      double bx0 = __builtin_inf(), bx1 = -__builtin_inf(), by0 = __builtin_inf(), by1 = -__builtin_inf();
      double ax = 0.0, ay = 0.0;
#pragma unroll
      for (int k = 0; k < MV_RING; k++) {
        if (k < n) {
          // world position of the vertex after the move: the reference's rotate-about-the-centroid-then-translate on the
          // old world ring, or -- body-frame rings -- the new pose applied to the constant body offsets
          const double x = body ? px[k] : px[k] + (-cx), y = body ? py[k] : py[k] + (-cy);
          const double rc = body ? cal : cda, rs = body ? sal : sda;
          const double xr = rc * x - rs * y, yr = rs * x + rc * y;
          const double mx = xr + (cx + dx), my = yr + (cy + dy);
          if (k > 0) {
            const double x1 = ax + (-ncx), y1 = ay + (-ncy), x2 = mx + (-ncx), y2 = my + (-ncy);
            const double xd = x2 - x1, yd = y2 - y1;
            const double u1 = nu - nxi * y1, u2 = nu - nxi * y2;
            const double v1 = nu + nxi * x1, v2 = nu + nxi * x2;
            const double ud = u2 - u1, vd = v2 - v1;
            e11 += ud * yd; e12 += ud * xd + vd * yd; e22 += vd * xd;
          }
          if (!body) S.vxy[o + k] = make_double2(mx, my);
          bx0 = fmin(bx0, mx); bx1 = fmax(bx1, mx); by0 = fmin(by0, my); by1 = fmax(by1, my);
          ax = mx; ay = my;
        }
      }
      S.bbx0[i] = bx0; S.bbx1[i] = bx1; S.bby0[i] = by0; S.bby1[i] = by1;
      e12 *= 0.5;
      const double d = 2 * area;
      *(double4*)(S.strain + (size_t)i * 4) = make_double4(e11 / d, e12 / d, e12 / d, e22 / d);
      S.cx[i] = ncx; S.cy[i] = ncy;
      if (S.rec32) rec32_store(S, i, ncx, ncy, rmx, bx0, bx1, by0, by1);
      if (S.crec) {            // what has changed of the floe's collision record (rmax, id, order key, ring size and sign stay)
        double2* r = S.crec + (size_t)i * 8;
        r[0] = make_double2(ncx, ncy); r[2].y = crec_vp(o, i, 0); r[3] = make_double2(bx0, bx1); r[4] = make_double2(by0, by1);
        r[5] = make_double2(nu, nv); r[6] = make_double2(nxi, area); r[7].x = h;
      }
      ISTAMP(4);
      if (gl_any) {
        const int gf = ghost_flag_of(wall, S.any_periodic_ew, S.any_periodic_ns, ncx, ncy, rmx, bx0, bx1, by0, by1, st_new == SZ_ACTIVE);
        if (ginl >= 0) {
          if (gf != 5) {             // the next step's ghosts of this parent, from what has just been computed (a few per cent of the threads)
            GhostRow R;
            R.cx = ncx; R.cy = ncy; R.b0 = bx0; R.b1 = bx1; R.b2 = by0; R.b3 = by1;
            R.rmax = rmx; R.area = area; R.h = h; R.mass = mass; R.mom = moment; R.al = al; R.u = nu; R.v = nv; R.xi = nxi;
            R.over = g_over; R.id = g_id; R.oki = g_oki; R.os = g_os; R.st = SZ_ACTIVE; R.tc = cal; R.ts = sal;
#pragma unroll
            for (int k = 0; k < MV_RING; k++) {        // the moved ring again (the expressions of the stores above: the same bits)
              const double x = body ? 0.0 : px[k] + (-cx), y = body ? 0.0 : py[k] + (-cy);
              const double xr = cda * x - sda * y, yr = sda * x + cda * y;
              R.rx[k] = xr + (cx + dx); R.ry[k] = yr + (cy + dy);
            }
            ISTAMP(5);
            ghost_inline_make(S, geo, wall, N, nv0, ginl, i, gf, n, o, R, RingRegs{ R });
            ISTAMP(6);
          }
        }
        else ghost_candidate_wave(S, gl_fill, gf != 5, i, gf, n, o);
      }
      if (PACK) {
        // ---- the floe's halo records for the next step: the drift test against the positions the boxes were gathered at, the ranks whose
        // box holds the centroid (or an image of it), a slot in each of their regions -- ONE atomic per wavefront and destination
        // (same-address atomics are worked off one at a time for the whole chip), handed on by the lanes' rank among the wavefront's
        // hits -- and the record.  The record carries the floe as the update has left it, BEFORE a swap with its ghost (a parent that
        // left the domain): the receiving rank makes the ghosts from the record with the very routine the owner has just run on the same
        // values (ghost_inline_make, which also swaps the received parent), so the instances of a floe are the same bits -- and the same
        // ghost numbers at the same places -- on every rank.  Ghosts made from a parent AFTER its swap are the same four places with two
        // numbers exchanged and one coordinate rounded once more: the Dict rule then keeps another instance pair of the same contact.
        const double fx = ncx, fy = ncy;
        if (PK.ref) {
          double ddx = fabs(fx - pk_rx), ddy = fabs(fy - pk_ry);
          if (PK.per_x && ddx > 0.5 * PK.Lx) ddx = fabs(ddx - PK.Lx);          // a parent the ghost pass wrapped around the domain
          if (PK.per_y && ddy > 0.5 * PK.Ly) ddy = fabs(ddy - PK.Ly);
          if (2.0 * fmax(ddx, ddy) > PK.margin) atomicOr(&S.cnt[C_ERR], ERR_HALO_DRIFT);
          pk_drift = fmax(pk_drift, fmax(ddx, ddy));
        }
        const unsigned long long hits = halo_hits(PK.boxes, PK.nranks, PK.me, PK.Lx, PK.Ly, PK.per_x, PK.per_y, fx, fy);
        const int lane = threadIdx.x & 63;
        int* run = PK.counts + 64;
        // Lane d draws the wavefront's slots of destination d: ALL destinations with ONE atomic instruction and one wait (nranks <= 64 lanes) --
        // a loop of returning atomics, one per destination the wavefront's floes reach, was as many dependent round trips (a corner tile's
        // wavefronts reach three to eight ranks).  Which lane asks changes nothing about the slots: they are handed out by the running totals.
        // (a wavefront with idle lanes -- the last one of the launch -- asks destination by destination, through its first lane that has a hit)
        const bool full = __ballot(true) == ~0ull;
        int mybase = 0;
        if (full) {
          int mycount = 0;
          for (int d = 0; d < PK.nranks; d++) { const int cnt_d = __popcll(__ballot((hits >> d) & 1ull)); if (lane == d) mycount = cnt_d; }
          if (mycount > 0) mybase = atomicAdd(&run[lane], mycount);
        }
        for (int d = 0; d < PK.nranks; d++) {
          const bool mine = (hits >> d) & 1ull;
          const unsigned long long m = __ballot(mine);
          if (!m) continue;
          int base;
          if (full) base = __shfl(mybase, d);
          else {
            const int leader = __ffsll((long long)m) - 1;
            base = 0;
            if (lane == leader) base = atomicAdd(&run[d], __popcll(m));
            base = __shfl(base, leader);
          }
          if (mine) {
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            if (slot >= (PK.dcap ? PK.dcap[d] : PK.cap) || n > halo_ring_of(S)) atomicOr(&S.cnt[C_ERR], n > halo_ring_of(S) ? ERR_CAP_RING : ERR_CAP_FLOES);
            else {
              double* r = PK.send + ((size_t)d * (PK.cap + 1) + 1 + slot) * halo_rec(S);
              r[0] = (double)g_oki; r[1] = (double)st_new; r[2] = (double)n; r[3] = fx; r[4] = fy; r[5] = rmx;
              r[6] = area; r[7] = h; r[8] = nu; r[9] = nv; r[10] = nxi; r[11] = (double)g_id;
#pragma unroll
              for (int k = 0; k < MV_RING; k++) {      // the moved ring once more (the expressions of the stores above: the same bits)
                if (k < n) {
                  const double x = px[k] + (-cx), y = py[k] + (-cy);
                  const double xr = cda * x - sda * y, yr = sda * x + cda * y;
                  r[12 + k] = xr + (cx + dx); r[12 + halo_ring_of(S) + k] = yr + (cy + dy);
                }
              }
            }
          }
        }
      }
    }
  }
  if (PACK) {
    // the largest displacement since the gather (one atomic per wavefront), then -- unless this launch belongs to a step the batch has
    // stopped or paused before -- the workgroup that finishes last writes the header records (halo_headers)
    __shared__ int pk_last;
    float dm = (float)pk_drift * 1.0001f;
    for (int d = 32; d >= 1; d >>= 1) dm = fmaxf(dm, __shfl_xor(dm, d));
    if ((threadIdx.x & 63) == 0 && PK.ref && dm > 0.f) {      // (non-negative floats order like their bits; waited for: the ticket below follows it)
      const int was = atomicMax(&PK.counts[64 + 65], __float_as_int(dm)); asm volatile("" :: "v"(was));
    }
    if (stop_test_late(S, stop)) {
      // nothing was integrated and nothing packed: the records of the last pack stay, but the peers must hear of the stop / the pause in the
      // header words of the next exchange (the pack launch this replaces ran regardless)
      if (blockIdx.x == 0 && (int)threadIdx.x < PK.nranks) {
        double* hdr = PK.send + (size_t)threadIdx.x * (PK.cap + 1) * halo_rec(S);
        hdr[1] = (double)S.cnt[C_STOP]; hdr[2] = (double)S.cnt[C_RETRYSTOP];
      }
    } else {
      // (no fence before the ticket: what the last workgroup reads of the others -- slot totals, drift, the stop word -- was written by
      //  atomics that have returned; the records themselves are read after the launch)
      __syncthreads();
      if (threadIdx.x == 0) pk_last = atomicAdd(&PK.counts[64 + 64], 1) == (int)gridDim.x - 1;
      __syncthreads();
      if (pk_last) halo_headers(S, PK.nranks, PK.send, PK.cap, PK.counts, PK.dcap, PK.ref != nullptr);
    }
  }
  // one atomic per wavefront and counter, spread over WARN_SLOTS lines: the guards fire for most floes of a stiff
  // field, and atomics on one address are worked off one at a time for the whole chip (~8 ns each: with a single
  // counter word this kernel took 56 us at 100k floes instead of 16)
  for (int d = 32; d >= 1; d >>= 1) { wh += __shfl_xor(wh, d); wf += __shfl_xor(wf, d); wv += __shfl_xor(wv, d); wx += __shfl_xor(wx, d); }
  if ((threadIdx.x & 63) == 0) {
    int* w = S.warn + (((blockIdx.x * blockDim.x + threadIdx.x) >> 6) % WARN_SLOTS) * 32;
    if (wh) atomicAdd(w + 0, wh);
    if (wf) atomicAdd(w + 1, wf);
    if (wv) atomicAdd(w + 2, wv);
    if (wx) atomicAdd(w + 3, wx);
  }
}

// 16 lanes per floe: _move_floe! (floe_utils.jl:82-93) on the ring and calc_strain!
// (update_floe.jl:425-453) with the new velocities.  Strain terms are evaluated from the moved
// coordinates recomputed in registers (same expression as the store, hence the same bits), the
// ring is overwritten afterwards, and the per-edge terms are summed in ring order.
// strain_only: calc_strain! on its own -- the ring stays where it is and nothing but the strain is written
__global__ void __launch_bounds__(256) sz_k_move_strain(State S, int strain_only, int bin, int gl_fill) {
  const GridGeo geo = grid_geo(S);
  constexpr int G = 16;
  __shared__ double t11[256 / G][64], t12[256 / G][64], t22[256 / G][64];
  if (stopped_late(S)) return;
  int N = S.cnt[C_NOWN];
  int gl = threadIdx.x % G, gi = threadIdx.x / G, gpb = blockDim.x / G;
  for (int i = blockIdx.x * gpb + gi; i < N; i += gridDim.x * gpb) {
    double cx = S.cx[i], cy = S.cy[i];
    double dx = 0.0, dy = 0.0, c = 1.0, s = 0.0;
    if (!strain_only) { dx = S.mot[i * 4]; dy = S.mot[i * 4 + 1]; c = S.mot[i * 4 + 2]; s = S.mot[i * 4 + 3]; }
    int o = S.voff[i], n = S.voff[i + 1] - o;
    double ncx = cx + dx, ncy = cy + dy;
    double u = S.u[i], xi = S.xi[i];
    auto moved = [&](int k, double& mx, double& my) {
      const double2 p = S.vxy[o + k];
      if (strain_only) { mx = p.x; my = p.y; return; }
      double x = p.x + (-cx), y = p.y + (-cy);
      double xr = c * x - s * y, yr = s * x + c * y;
      mx = xr + (cx + dx); my = yr + (cy + dy);
    };
    double e11 = 0, e12 = 0, e22 = 0;
    for (int base = 1; base < n; base += 64) {
      for (int k = base + gl; k < n && k < base + 64; k += G) {
        double ax, ay, bx, by;
        moved(k - 1, ax, ay); moved(k, bx, by);
        double x1 = ax + (-ncx), y1 = ay + (-ncy), x2 = bx + (-ncx), y2 = by + (-ncy);
        double xd = x2 - x1, yd = y2 - y1;
        // rad*sin(atan(y, x)) == y and rad*cos(atan(y, x)) == x up to the last bit (update_floe.jl:437-442
        // evaluates them through atan/sin/cos); the vertex velocities use floe.u for v1, v2: literal
        double u1 = u - xi * y1, u2 = u - xi * y2;
        double v1 = u + xi * x1, v2 = u + xi * x2;
        double ud = u2 - u1, vd = v2 - v1;
        t11[gi][k - base] = ud * yd; t12[gi][k - base] = ud * xd + vd * yd; t22[gi][k - base] = vd * xd;
      }
      gsync();
      int lim = n - base < 64 ? n - base : 64;
      for (int k = 0; k < lim; k++) { e11 += t11[gi][k]; e12 += t12[gi][k]; e22 += t22[gi][k]; }
      gsync();
    }
    // all reads of the old ring are done (each lane's loads complete before its dependent LDS
    // stores, and the sums above consumed every LDS store of the group)
    double bx0 = __builtin_inf(), bx1 = -__builtin_inf(), by0 = __builtin_inf(), by1 = -__builtin_inf();
    for (int k = gl; k < n && !strain_only; k += G) {
      double mx, my; moved(k, mx, my);
      S.vxy[o + k] = make_double2(mx, my);
      bx0 = fmin(bx0, mx); bx1 = fmax(bx1, mx); by0 = fmin(by0, my); by1 = fmax(by1, my);
    }
    bx0 = gmin<G>(bx0); bx1 = gmax<G>(bx1); by0 = gmin<G>(by0); by1 = gmax<G>(by1);
    if (gl == 0 && !strain_only) {
      S.bbx0[i] = bx0; S.bbx1[i] = bx1; S.bby0[i] = by0; S.bby1[i] = by1;
      if (S.rec32) rec32_store(S, i, ncx, ncy, S.rmax[i], bx0, bx1, by0, by1);
    }
    if (gl == 0) {
      e12 *= 0.5;
      double d = 2 * S.area[i];
      S.strain[i * 4 + 0] = e11 / d; S.strain[i * 4 + 1] = e12 / d; S.strain[i * 4 + 2] = e12 / d; S.strain[i * 4 + 3] = e22 / d;
      if (!strain_only) { S.cx[i] = ncx; S.cy[i] = ncy; }
      if (bin) cell_insert(S, geo, i, ncx, ncy);          // for the next step's neighbour search
      if (gl_fill >= 0) {                                 // ... and its ghost pass
        const double wall[4] = { S.eval[0], S.eval[1], S.eval[2], S.eval[3] };
        const int gf = ghost_flag_of(wall, S.any_periodic_ew, S.any_periodic_ns, ncx, ncy, S.rmax[i], bx0, bx1, by0, by1, S.status[i] == SZ_ACTIVE);
        if (gf != 5) ghost_candidate_one(S, gl_fill, i, gf, n, o);
      }
    }
  }
}

// ============================================================================ halo exchange (multi-GPU)
// One record per floe sent to another rank: the columns the collision path reads + the ring (HALO_REC, defined before the integrator, which
// writes the records of the next step itself in tiled runs).

// Every owned floe whose centroid -- or one of its periodic images -- lies inside another rank's
// (already expanded) box is written to that rank's region of the send buffer.  One block: the
// per-destination counters live in LDS and the header records are written by the same launch.
// With send == nullptr only the counts are produced (sizing pass).
// dcap (may be null): slots the region of every destination really has (<= cap, the stride of the regions; 0: not a
// neighbour -- a floe that should go there all the same means the boxes are out of date).  ref (may be null): the owned
// centroids when the boxes were gathered, 2 x capM doubles: the halo selection is only valid while no floe has moved
// further than half the drift margin built into the boxes -- beyond that a neighbour across a tile edge could be missed
// silently, so it is an error (ERR_HALO_DRIFT), raised one step early enough.
constexpr int PACK_TPB = 256;
// Workgroups of 256 owned floes: a workgroup counts its records per destination in LDS, reserves that many slots of every region
// with one atomic per destination (counts[64 + d]: running totals), then writes its records; the workgroup that finishes last
// (ticket in counts[128]) writes the header records and the totals and leaves the scratch words zero for the next launch.  (The
// order of the records inside a region is arbitrary, as it was with one workgroup: order-dependent rules use the global index.)
__global__ void __launch_bounds__(PACK_TPB) sz_k_halo_pack(State S, int nranks, int me, const double* boxes, double Lx, double Ly,
                                                           int per_x, int per_y, double* send, int cap, int* counts,
                                                           const int* dcap, const double* ref, double margin) {
  __shared__ int lc[64], lc2[64], base[64];
  __shared__ int ldrift, last;
  int* run = counts + 64;
  if (threadIdx.x < 64) { lc[threadIdx.x] = 0; lc2[threadIdx.x] = 0; }
  if (threadIdx.x == 0) ldrift = 0;
  __syncthreads();
  const int n = S.cnt[C_NOWN];
  auto hits_of = [&](double cx, double cy) { return halo_hits(boxes, nranks, me, Lx, Ly, per_x, per_y, cx, cy); };
  double dmax = 0.0;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
    double cx = S.cx[q], cy = S.cy[q];
    if (ref) {
      double ddx = fabs(cx - ref[q]), ddy = fabs(cy - ref[S.capM + q]);
      if (per_x && ddx > 0.5 * Lx) ddx = fabs(ddx - Lx);          // a parent the ghost pass wrapped around the domain
      if (per_y && ddy > 0.5 * Ly) ddy = fabs(ddy - Ly);
      if (2.0 * fmax(ddx, ddy) > margin) atomicOr(&S.cnt[C_ERR], ERR_HALO_DRIFT);
      dmax = fmax(dmax, fmax(ddx, ddy));
    }
    unsigned long long hits = hits_of(cx, cy);
    while (hits) { int d = __ffsll((long long)hits) - 1; hits &= hits - 1; atomicAdd(&lc[d], 1); }
  }
  if (ref && dmax > 0.0) atomicMax(&ldrift, __float_as_int((float)dmax * 1.0001f));      // (non-negative floats order like their bits)
  __syncthreads();
  if ((int)threadIdx.x < nranks) base[threadIdx.x] = lc[threadIdx.x] ? atomicAdd(&run[threadIdx.x], lc[threadIdx.x]) : 0;
  __syncthreads();
  if (send)
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
      double cx = S.cx[q], cy = S.cy[q];
      unsigned long long hits = hits_of(cx, cy);
      int o = S.voff[q], nv = S.voff[q + 1] - o;
      while (hits) {
        int d = __ffsll((long long)hits) - 1; hits &= hits - 1;
        int slot = base[d] + atomicAdd(&lc2[d], 1);
        if (slot >= (dcap ? dcap[d] : cap) || nv > halo_ring_of(S)) { atomicOr(&S.cnt[C_ERR], nv > halo_ring_of(S) ? ERR_CAP_RING : ERR_CAP_FLOES); continue; }
        double* r = send + ((size_t)d * (cap + 1) + 1 + slot) * halo_rec(S);
        r[0] = (double)S.okey[q]; r[1] = (double)S.status[q]; r[2] = (double)nv; r[3] = cx; r[4] = cy; r[5] = S.rmax[q];
        r[6] = S.area[q]; r[7] = S.height[q]; r[8] = S.u[q]; r[9] = S.v[q]; r[10] = S.xi[q]; r[11] = (double)S.id[q];
        for (int k = 0; k < nv; k++) { const double2 p = S.vxy[o + k]; r[12 + k] = p.x; r[12 + halo_ring_of(S) + k] = p.y; }
      }
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (ref && ldrift) atomicMax(&run[65], ldrift);
    __threadfence();
    last = atomicAdd(&run[64], 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  halo_headers(S, nranks, send, cap, counts, dcap, ref != nullptr);
}
// ---- fixed-layout exchange buffers: region of peer r = 1 header record (count in [0]) followed by
// `cap` record slots.  The host never needs the counts, so a whole step is enqueued without a sync.
// One block appends the received floes as extra parents [nown, nown + nrec): ring offsets by an
// in-kernel scan of the ring sizes, then the copy.
__global__ void __launch_bounds__(1024) sz_k_halo_unpack(State S, const double* recv, int nranks, int cap, int bin, int gl_fill) {
  const GridGeo geo = grid_geo(S);
  __shared__ int before[65];
  __shared__ int tot, carry_s;
  if (stopped(S)) return;
  if (S.step > 0 && S.stop_on_tags) {          // sz_tile_run's list-based steps: a peer tagged a floe in the step before (its header's stop word; this rank's own region is zero)
    int stop = 0;
    for (int r = 0; r < nranks; r++) { const int f = (int)recv[(size_t)r * (cap + 1) * halo_rec(S) + 1]; if (f > 0) stop = f; }
    if (stop > 0) { if (threadIdx.x == 0) S.cnt[C_STOP] = stop; return; }
  }
  if (threadIdx.x == 0) {
    int acc = 0; bool bad = false;
    for (int r = 0; r < nranks; r++) { int cnt = (int)recv[(size_t)r * (cap + 1) * halo_rec(S)]; if (cnt > cap) { bad = true; cnt = cap; } before[r] = acc; acc += cnt; }
    before[nranks] = acc;
    if (bad) atomicOr(&S.cnt[C_ERR], ERR_CAP_FLOES);
    carry_s = 0;
  }
  __syncthreads();
  const int nrec = before[nranks];
  const int nown = S.cnt[C_NOWN];
  const int vbase = S.voff[nown];
  if (nown + nrec > S.capM) { if (threadIdx.x == 0) atomicOr(&S.cnt[C_ERR], ERR_CAP_FLOES); return; }
  // pass 1: ring offsets (gvscan[q] = exclusive scan of the ring sizes in record order)
  for (int base = 0; base < nrec; base += blockDim.x) {
    int q = base + threadIdx.x, nv = 0;
    if (q < nrec) {
      int src = 0; while (q >= before[src + 1]) src++;
      nv = (int)recv[((size_t)src * (cap + 1) + 1 + (q - before[src])) * halo_rec(S) + 2];
    }
    int ex = block_exclusive_scan(nv, &tot);
    if (q < nrec) S.gvscan[q] = carry_s + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  const int totv = carry_s;
  if (vbase + totv > S.capV) { if (threadIdx.x == 0) atomicOr(&S.cnt[C_ERR], ERR_CAP_VERTS); return; }
  // pass 2: copy
  for (int q = threadIdx.x; q < nrec; q += blockDim.x) {
    int src = 0; while (q >= before[src + 1]) src++;
    const double* r = recv + ((size_t)src * (cap + 1) + 1 + (q - before[src])) * halo_rec(S);
    int g = nown + q, nv = (int)r[2], vb = vbase + S.gvscan[q];
    S.okey[g] = (long long)r[0]; S.status[g] = (int)r[1]; S.cx[g] = r[3]; S.cy[g] = r[4]; S.rmax[g] = r[5];
    S.area[g] = r[6]; S.height[g] = r[7]; S.u[g] = r[8]; S.v[g] = r[9]; S.xi[g] = r[10]; S.id[g] = (long long)r[11];
    S.ghost_id[g] = 0; S.parent[g] = g; S.ngh[g] = 0; S.overarea[g] = 0.0;
    S.mass[g] = 0.0; S.moment[g] = 0.0; S.alpha[g] = 0.0;
    for (int k = 0; k < MAX_GHOSTS; k++) S.gh[g * MAX_GHOSTS + k] = -1;
    S.voff[g] = vb; S.voff[g + 1] = vb + nv;
    double x0 = __builtin_inf(), x1 = -__builtin_inf(), y0 = __builtin_inf(), y1 = -__builtin_inf();
    for (int k = 0; k < nv; k++) {
      double x = r[12 + k], y = r[12 + halo_ring_of(S) + k];
      S.vxy[vb + k] = make_double2(x, y);
      x0 = fmin(x0, x); x1 = fmax(x1, x); y0 = fmin(y0, y); y1 = fmax(y1, y);
    }
    S.osign[g] = ring_signed_area(r + 12, r + 12 + halo_ring_of(S), nv) >= 0.0 ? 1 : -1;
    S.bbx0[g] = x0; S.bbx1[g] = x1; S.bby0[g] = y0; S.bby1[g] = y1;
    if (S.rec32) rec32_store(S, g, r[3], r[4], r[5], x0, x1, y0, y1);
    if (bin) cell_insert(S, geo, g, r[3], r[4]);
    if (gl_fill >= 0) {
      const double wall[4] = { S.eval[0], S.eval[1], S.eval[2], S.eval[3] };
      const int gf = ghost_flag_of(wall, S.any_periodic_ew, S.any_periodic_ns, r[3], r[4], r[5], x0, x1, y0, y1, (int)r[1] == SZ_ACTIVE);
      if (gf != 5) ghost_candidate_one(S, gl_fill, g, gf, nv, vb);
    }
  }
  if (threadIdx.x == 0) {
    S.cnt[C_M] = nown + nrec; S.cnt[C_N] = nown + nrec; S.cnt[C_NV] = vbase + totv; S.cnt[C_NHALO] = nrec;
    if (nrec == 0) S.voff[nown] = vbase;
  }
}
// ---- the halo of a tiled step whose ghosts are made "inline" (sz_tile_run): THREAD per received record, as many workgroups as the
// regions hold slots.  A received floe and the ghosts it needs on this rank are rows of the SAME bump allocator the owned parents'
// ghosts come from (State::galloc: {rows << 32 | ring points} in one atomic per wavefront, rings packed back to back in allocation
// order), behind the owned floes: [owned | ghosts of owned, halo floes, ghosts of halo floes in allocation order].  Nothing in a
// step needs the parents to be contiguous -- order-dependent rules use the order keys (global indices), totals are only kept for
// owned floes -- so the count of "parents" stays the owned count and no row offset has to be scanned.  The thread bins the floe,
// makes its periodic ghosts from the record it holds (ghost_inline_make) and leaves its order key in the step's key table.
// Stop agreement: the header record of every rank carries that rank's stop request (a floe was tagged in the step before:
// simplify_floes!, simulation.jl:205-214, is the host's); any request ends the batch on THIS rank too, before this step has touched
// anything -- all ranks return the state after the same step.
constexpr int UNPACK_TPB = 256;
__global__ void __launch_bounds__(UNPACK_TPB) sz_k_halo_unpack_inline(State S, const double* recv, int nranks, int me, int cap, int slot, int nown) {
  if (stopped(S)) return;
  {
    int stop = 0, pause = 0;
    for (int r = 0; r < nranks; r++) {
      if (r == me) continue;
      const double* hdr = recv + (size_t)r * (cap + 1) * halo_rec(S);
      const int f = (int)hdr[1], pz = (int)hdr[2];
      if (f > 0) stop = f;
      if (pz > 0) pause = pz;
    }
    // a peer paused inside the step before (its narrow phase needs the variant that is left out of the steps): this rank has finished that
    // step and waits here -- nothing of this step has touched anything -- until the host has finished it over there and starts the rest again
    if (pause > 0) { if (blockIdx.x == 0 && threadIdx.x == 0 && S.cnt[C_RETRYSTOP] == 0) S.cnt[C_RETRYSTOP] = pause; }
    if (stop > 0 && S.stop_on_tags) { if (blockIdx.x == 0 && threadIdx.x == 0) S.cnt[C_STOP] = stop; }      // (every requester of a batch names the same step)
    if (pause > 0 || (stop > 0 && S.stop_on_tags)) return;
  }
  const GridGeo geo = grid_geo(S);
  const int NV0 = S.voff[nown];
  const double wall[4] = { S.eval[0], S.eval[1], S.eval[2], S.eval[3] };
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = (int)(t / cap), q = (int)(t % cap);
  bool act = r < nranks && r != me;
  if (act) {
    int cnt_r = (int)recv[(size_t)r * (cap + 1) * halo_rec(S)];
    if (cnt_r > cap) { if (q == 0) atomicOr(&S.cnt[C_ERR], ERR_CAP_FLOES); cnt_r = cap; }
    act = q < cnt_r;
  }
  if (t == 0) {          // (statistics only)
    int tot = 0;
    for (int rr = 0; rr < nranks; rr++) if (rr != me) { const int c0 = (int)recv[(size_t)rr * (cap + 1) * halo_rec(S)]; tot += c0 < cap ? c0 : cap; }
    S.cnt[C_NHALO] = tot;
  }
  // ---- the record, all of it before the first store
  const double* rec = recv + ((size_t)(act ? r : 0) * (cap + 1) + 1 + (act ? q : 0)) * halo_rec(S);
  GhostRow R;
  int nv = 0;
  if (act) {
    nv = (int)rec[2];
    if (nv > MV_RING) { atomicOr(&S.cnt[C_ERR], ERR_CAP_RING); act = false; nv = 0; }
  }
  R.oki = act ? (long long)rec[0] : 0; R.st = act ? (int)rec[1] : SZ_ACTIVE; R.cx = act ? rec[3] : 0.0; R.cy = act ? rec[4] : 0.0;
  R.rmax = act ? rec[5] : 0.0; R.area = act ? rec[6] : 0.0; R.h = act ? rec[7] : 0.0; R.u = act ? rec[8] : 0.0; R.v = act ? rec[9] : 0.0;
  R.xi = act ? rec[10] : 0.0; R.id = act ? (long long)rec[11] : 0; R.mass = 0.0; R.mom = 0.0; R.al = 0.0; R.over = 0.0; R.tc = 1.0; R.ts = 0.0;
#pragma unroll
  for (int k = 0; k < MV_RING; k++) { R.rx[k] = k < nv ? rec[12 + k] : 0.0; R.ry[k] = k < nv ? rec[12 + halo_ring_of(S) + k] : 0.0; }
  double x0 = __builtin_inf(), x1 = -__builtin_inf(), y0 = __builtin_inf(), y1 = -__builtin_inf();
#pragma unroll
  for (int k = 0; k < MV_RING; k++) if (k < nv) { x0 = fmin(x0, R.rx[k]); x1 = fmax(x1, R.rx[k]); y0 = fmin(y0, R.ry[k]); y1 = fmax(y1, R.ry[k]); }
  R.b0 = x0; R.b1 = x1; R.b2 = y0; R.b3 = y1;
  double sarea = 0.0;          // ring_signed_area() on the registers: the same sums in the same order
  if (nv > 0) {
    double p1x = R.rx[0], p1y = R.ry[0];
#pragma unroll
    for (int k = 1; k < MV_RING; k++) if (k < nv) { const double p2x = R.rx[k], p2y = R.ry[k]; sarea += p1x * p2y - p1y * p2x; p1x = p2x; p1y = p2y; }
    sarea += p1x * R.ry[0] - p1y * R.rx[0];
    sarea = sarea / 2.0;
  }
  R.os = sarea >= 0.0 ? 1 : -1;
  const int gf = act ? ghost_flag_of(wall, S.any_periodic_ew, S.any_periodic_ns, R.cx, R.cy, R.rmax, x0, x1, y0, y1, R.st == SZ_ACTIVE) : 5;
  const int ng = ghost_count_of(gf);
  // ---- rows and ring points of the floe and its ghosts: one allocation per wavefront
  const unsigned long long mine = act ? (((unsigned long long)(1 + ng) << 32) | (unsigned)(nv * (1 + ng))) : 0ull;
  unsigned long long inc = mine;
  const int lane = threadIdx.x & 63;
  for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  const unsigned long long tot = __shfl(inc, 63);
  unsigned long long base = 0;
  if (lane == 0 && tot) base = atomicAdd(&S.galloc[slot * 16], tot);
  base = __shfl(base, 0);
  if (!act) return;
  const unsigned long long old = base + inc - mine;
  const int og = (int)(old >> 32), ov = (int)(old & 0xffffffffull);
  if (nown + og + 1 + ng > S.capM) { atomicOr(&S.cnt[C_ERR], ERR_CAP_FLOES); atomicOr(&S.galloc[slot * 16 + 1], 1ull); return; }
  if (NV0 + ov + nv * (1 + ng) > S.capV) { atomicOr(&S.cnt[C_ERR], ERR_CAP_VERTS); atomicOr(&S.galloc[slot * 16 + 1], 1ull); return; }
  const int g = nown + og, vb = NV0 + ov;
  int cix, ciy; cell_of(geo, R.cx, R.cy, cix, ciy);
  const int cell_c = ciy * geo.ncx + cix, cell_s = atomicAdd(&S.cell_cnt[cell_c], 1);
  if (cell_s < CELL_K) S.cell_slots[(size_t)cell_c * CELL_K + cell_s] = g;
  else S.cell_items[g] = atomicExch(&S.cell_ovf[cell_c], g + 1) - 1;
  S.okey[g] = R.oki; S.status[g] = R.st; S.cx[g] = R.cx; S.cy[g] = R.cy; S.rmax[g] = R.rmax;
  S.area[g] = R.area; S.height[g] = R.h; S.u[g] = R.u; S.v[g] = R.v; S.xi[g] = R.xi; S.id[g] = R.id;
  S.ghost_id[g] = 0; S.parent[g] = g; S.ngh[g] = 0; S.overarea[g] = 0.0;
  for (int k = 0; k < MAX_GHOSTS; k++) S.gh[g * MAX_GHOSTS + k] = -1;
  S.voff[g] = vb; S.voff[g + 1] = vb + nv;
#pragma unroll
  for (int k = 0; k < MV_RING; k++) if (k < nv) S.vxy[vb + k] = make_double2(R.rx[k], R.ry[k]);
  S.osign[g] = R.os;
  S.bbx0[g] = x0; S.bbx1[g] = x1; S.bby0[g] = y0; S.bby1[g] = y1;
  if (S.rec32) rec32_store(S, g, R.cx, R.cy, R.rmax, x0, x1, y0, y1);
  if (S.crec) crec_store_all(S, g, R.cx, R.cy, R.rmax, R.id, R.oki, nv, R.os, vb, g, 0, x0, x1, y0, y1, R.u, R.v, R.xi, R.area, R.h, 0ll);
  S.gkeys[(size_t)slot * S.capM + og] = R.oki;
  if (gf != 5) {
    const unsigned long long given = ((unsigned long long)(og + 1) << 32) | (unsigned)(ov + nv);
    ghost_inline_make(S, geo, wall, nown, NV0, slot, g, gf, nv, vb, R, RingRegs{ R }, &given);
  }
}
// diagnosis (sz_debug_find_key): the row that held order key `key` in the last resident step that used allocator `slot` -- ghosts and halo floes stay
// in memory behind the parents until the next step overwrites them.  out[0] = row (-1: not found), then cx, cy, u, v, xi, rmax, area, height,
// box x0 x1 y0 y1, ring points, parent, status; the ring x.. at out[16], y.. at out[16 + MV_RING]
__global__ void sz_k_debug_find_key(State S, int slot, long long key, int nown, double* out) {
  if (blockIdx.x || threadIdx.x) return;
  const int rows = (int)(S.galloc[slot * 16] >> 32);
  out[0] = -1.0;
  for (int og = 0; og < rows && og < S.capM - nown; og++) {
    if (S.gkeys[(size_t)slot * S.capM + og] != key) continue;
    const int g = nown + og, o = S.voff[g], n = S.voff[g + 1] - o;
    out[0] = g; out[1] = S.cx[g]; out[2] = S.cy[g]; out[3] = S.u[g]; out[4] = S.v[g]; out[5] = S.xi[g]; out[6] = S.rmax[g]; out[7] = S.area[g]; out[8] = S.height[g];
    out[9] = S.bbx0[g]; out[10] = S.bbx1[g]; out[11] = S.bby0[g]; out[12] = S.bby1[g]; out[13] = n; out[14] = S.parent[g]; out[15] = S.status[g];
    for (int k = 0; k < MV_RING; k++) { const double2 p = k < n ? S.vxy[o + k] : make_double2(0.0, 0.0); out[16 + k] = p.x; out[16 + MV_RING + k] = p.y; }
    return;
  }
}
// diagnosis (sz_debug_pairs_of_ids): the pair items of the last resident step between instances (parent, ghosts) of two floe ids, by order key:
// out[0] = entries, then per entry {key of the owner row, key of the partner row, contact rows of the item, owner row, partner row}
__global__ void sz_k_debug_pairs_of_ids(State S, int slot, long long ida, long long idb, int nown, double* out, int cap) {
  if (blockIdx.x || threadIdx.x) return;
  const int rows = nown + (int)(S.galloc[slot * 16] >> 32);
  int n = 0;
  for (int k = 0; k < rows && k < S.capM; k++) {
    const long long idk = S.id[k];
    if (idk != ida && idk != idb) continue;
    const int no = S.n_out[k];
    for (int r = 0; r < no && r < S.maxnb; r++) {
      const int p = S.nb_out[(size_t)k * S.maxnb + r];
      if (p < 0 || p >= S.capM) continue;
      const long long idp = S.id[p];
      if (!((idk == ida && idp == idb) || (idk == idb && idp == ida))) continue;
      if (n < cap) {
        const int2 info = S.it_info[(size_t)k * S.maxnb + r];
        out[1 + 5 * n] = (double)S.okey[k]; out[2 + 5 * n] = (double)S.okey[p]; out[3 + 5 * n] = (double)(info.x & 255); out[4 + 5 * n] = k; out[5 + 5 * n] = p;
      }
      n++;
    }
  }
  out[0] = n;
}
// bounding box of the owned centroids and the largest rmax: out[0..4] = xmin, xmax, ymin, ymax, rmax; out[5] = the largest
// displacement since the last box gather as the last pack kernel measured it (C_DRIFT); out[6] = the largest |u|, |v| now
// ctr (may be null): the centre of the box at the last gather -- in a periodic direction every centroid is taken at its image nearest to
// that centre, so that a parent the ghost pass has wrapped to the other side of the domain does not stretch its owner's box (and with it
// the owner's halo) across the whole domain; the box may then reach beyond the walls, which the senders' image test handles
__global__ void __launch_bounds__(1024) sz_k_owned_box(State S, double* out, const double* ctr, double Lx, double Ly, int per_x, int per_y) {
  __shared__ double sh[6][16];
  int n = S.cnt[C_NOWN];
  double x0 = __builtin_inf(), y0 = __builtin_inf(), x1 = -__builtin_inf(), y1 = -__builtin_inf(), rm = 0.0, vm = 0.0;
  const double ccx = ctr ? ctr[0] : 0.0, ccy = ctr ? ctr[1] : 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double xi = S.cx[i], yi = S.cy[i];
    if (ctr && per_x) { if (xi - ccx > 0.5 * Lx) xi -= Lx; else if (ccx - xi > 0.5 * Lx) xi += Lx; }
    if (ctr && per_y) { if (yi - ccy > 0.5 * Ly) yi -= Ly; else if (ccy - yi > 0.5 * Ly) yi += Ly; }
    x0 = fmin(x0, xi); x1 = fmax(x1, xi); y0 = fmin(y0, yi); y1 = fmax(y1, yi); rm = fmax(rm, S.rmax[i]);
    vm = fmax(vm, fmax(fabs(S.u[i]), fabs(S.v[i])));
  }
  for (int d = 32; d >= 1; d >>= 1) {
    x0 = fmin(x0, __shfl_xor(x0, d)); y0 = fmin(y0, __shfl_xor(y0, d));
    x1 = fmax(x1, __shfl_xor(x1, d)); y1 = fmax(y1, __shfl_xor(y1, d)); rm = fmax(rm, __shfl_xor(rm, d)); vm = fmax(vm, __shfl_xor(vm, d));
  }
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wid] = x0; sh[1][wid] = y0; sh[2][wid] = x1; sh[3][wid] = y1; sh[4][wid] = rm; sh[5][wid] = vm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
      x0 = fmin(x0, sh[0][w]); y0 = fmin(y0, sh[1][w]); x1 = fmax(x1, sh[2][w]); y1 = fmax(y1, sh[3][w]); rm = fmax(rm, sh[4][w]); vm = fmax(vm, sh[5][w]);
    }
    out[0] = x0; out[1] = x1; out[2] = y0; out[3] = y1; out[4] = rm; out[5] = (double)__int_as_float(S.cnt[C_DRIFT]); out[6] = vm;
  }
}

// which_vertices_match_points (floe_utils.jl:331-352) on its own, for the reference's test vectors: the narrow phase's
// match_vertices() -- the crossing points of an item against one contact region -- fed with given points and a given ring.
// One wavefront; out[0] = count, out[1..] = sorted 0-based vertex indices.
__global__ void __launch_bounds__(64) sz_k_debug_match_vertices(int npts, const double* px, const double* py, int nr, const double* rx,
                                                               const double* ry, int* out) {
  __shared__ GroupMem<NARROW_CAP2, NARROW_KC2, NARROW_RC2, 16> m;
  const int gl = threadIdx.x;
  for (int k = gl; k < npts; k += 64) { m.cx[k] = px[k]; m.cy[k] = py[k]; m.uniq[k] = 1; }
  for (int k = gl; k < nr; k += 64) { m.reg[0][0][k] = rx[k]; m.reg[0][1][k] = ry[k]; }
  if (gl == 0) m.nx = npts;
  __syncthreads();
  const int n = match_vertices<64>(m, gl, npts, m.reg[0][0], m.reg[0][1], nr);
  if (gl == 0) { out[0] = n; for (int k = 0; k < n; k++) out[1 + k] = m.midx[k]; }
}

// ============================================================================ stats
__global__ void sz_k_stats(State S, long long* out) {
  // out[0] = sum ring points over the pairs the narrow phase ran, out[1] = pair rows, out[2] = elem rows,
  // out[3] = interaction rows, out[4], out[5]: floes tagged remove / fuse (what simplify_floes!, simulation.jl:206,
  // has to act on), out[6..9]: the guard counters of the last timestep_floe_properties! (sum over the slots)
  const int nel = S.cnt[C_NELEM];
  long long v[12] = { 0 };
  // pairs of the last step: per floe its owned pairs (all / those with overlapping ring boxes = the items run)
  const int mlast = S.cnt[C_M] > S.cnt[C_N] ? S.cnt[C_M] : S.cnt[C_N] + S.cnt[C_NGHOSTS];      // the ghosts of the last step own pairs too
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < mlast; k += gridDim.x * blockDim.x) {
    const int nk = S.n_out[k];
    v[10] += nk;
    const int nvk = ring_n(S, k);
    for (int r = 0; r < nk; r++) {
      const int2 info = S.it_info[(size_t)k * S.maxnb + r];
      if (info.y < 0) continue;          // ring boxes disjoint: the pair was not an item of the narrow phase (it_info = {0, -1})
      v[11]++;
      const int j = S.nb_out[(size_t)k * S.maxnb + r];
      v[0] += nvk + ring_n(S, j);
      v[1] += info.x & 0xff;
    }
  }
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nel; q += gridDim.x * blockDim.x) v[2] += S.it_info[S.capM * S.maxnb + q].x & 0xff;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < S.cnt[C_M]; k += gridDim.x * blockDim.x) {
    v[3] += S.inter_cnt[k];
    int st = S.status[k];
    v[4] += st == SZ_REMOVE; v[5] += st == SZ_FUSE;
  }
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < WARN_SLOTS * 4; q += gridDim.x * blockDim.x) v[6 + (q & 3)] += S.warn[(q >> 2) * 32 + (q & 3)];
  for (int k = 0; k < 12; k++) {
    long long x = v[k];
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd((unsigned long long*)&out[k < 10 ? k : k + 6], (unsigned long long)x);     // pairs, pairs run: out[16], out[17]
  }
  // out[10..15], out[18..19]: the cumulative narrow-phase work counters (sum over the slots)
  if (blockIdx.x == 0 && threadIdx.x < 8) {
    unsigned long long t = 0;
    for (int q = 0; q < ACC_SLOTS; q++) t += S.acc[(size_t)q * 8 + threadIdx.x];
    out[threadIdx.x < 6 ? 10 + threadIdx.x : 12 + threadIdx.x] = (long long)t;
  }
}

}  // namespace sz
