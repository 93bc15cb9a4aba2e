// sz_output.hpp — the output path that reads the resident floe state:
//   calc_eulerian_data! (output.jl:793-914): floe data averaged on the GridOutputWriter's grid, and the
//   "nothing to simplify" test of simplify_floes! (simplification.jl:66, 287-290).
// The Eulerian averages are the third floe ∩ rectangle clip workload after the contact path and the two-way
// coupling, and reuse the same group-cooperative clipper:
//   sz_k_eul_entries   per floe: the output cells its ring box reaches -> entries keyed  cell * M + floe.  The
//                      reference's candidate test (cell centre within rmax + cell diagonal of the centroid,
//                      :808-819) only prefilters: entries without area are dropped afterwards (:851-852), and
//                      every floe with area in a cell passes both tests, so the box test gives the same lists.
//   (radix sort)       entries by key = per cell, floes ascending: the order the serial reference sums in
//   sz_k_eul_cell_area per cell: area of the cell minus topography (the denominator of si_frac, :829-837)
//   sz_k_eul_area      per entry: area of floe ∩ cell, minus what topography covers of it
//   sz_k_eul_reduce    per cell: the 18 averages (:855-905)
// Topography: the reference subtracts it from the cell polygon (diff_polys) before clipping; here the same areas
// come from area(f ∩ (c \ t)) = area(f ∩ c) - sum_t area((f ∩ c) ∩ t) for elements that do not overlap one another.
#pragma once
#include "sz_kernels.hpp"
#include "sz_twoway.hpp"      // RectClip: the rectangle pipeline

namespace sz {

constexpr int EUL_COUNT = 18;   // SZ_EUL_* of subzero_hip.h

struct EulGrid {
  int nx, ny, M, cap;
  const double *xg, *yg;            // grid lines (nx + 1, ny + 1)
  unsigned long long *keys;         // entries: cell * M + floe (sorted before the area kernel runs)
  double* pic;                      // per entry: area of the floe's part in the cell
  double* cell_area;                // per cell: cell minus topography
  double* data;                     // [EUL_COUNT][nx][ny]
  int* count;
};

__device__ __forceinline__ int eul_floor_clamped(double v, int lo, int hi) {
  if (!(v > (double)lo)) return lo;
  if (!(v < (double)hi)) return hi;
  return (int)floor(v);
}
// range of grid intervals [g[k], g[k+1]] that the closed interval [b0, b1] touches (the lines are evenly spaced; the
// estimate is widened by one and then checked against the actual lines)
__device__ __forceinline__ void eul_range(const double* g, int n, double b0, double b1, int& lo, int& hi) {
  const double rd = 1.0 / (g[1] - g[0]);
  lo = eul_floor_clamped((b0 - g[0]) * rd - 1.0, 0, n - 1);
  hi = eul_floor_clamped((b1 - g[0]) * rd + 1.0, 0, n - 1);
  while (lo <= hi && g[lo + 1] < b0) lo++;
  while (hi >= lo && b1 < g[hi]) hi--;
}
__global__ void sz_k_eul_entries(State S, EulGrid E, int fill) {
  const int M = S.cnt[C_M];
  const int lane = threadIdx.x & 63;
  // whole wavefronts go round together: one atomic per wavefront reserves the entries of its 64 floes (atomics on
  // one address are worked off one at a time chip-wide)
  for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63; i0 < M; i0 += gridDim.x * blockDim.x) {
    const int i = i0 + lane;
    int lx = 0, hx = -1, ly = 0, hy = -1;
    if (i < M) {
      eul_range(E.xg, E.nx, S.bbx0[i], S.bbx1[i], lx, hx);
      eul_range(E.yg, E.ny, S.bby0[i], S.bby1[i], ly, hy);
    }
    const int n = (lx > hx || ly > hy) ? 0 : (hx - lx + 1) * (hy - ly + 1);
    int inc = n;
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    const int tot = __shfl(inc, 63);
    int base = 0;
    if (lane == 0 && tot) base = atomicAdd(E.count, tot);
    base = __shfl(base, 0);
    if (!fill || n == 0 || base + inc > E.cap) continue;
    int k = base + inc - n;
    for (int ix = lx; ix <= hx; ix++)
      for (int iy = ly; iy <= hy; iy++) E.keys[k++] = (unsigned long long)(ix * E.ny + iy) * (unsigned long long)E.M + (unsigned long long)i;
  }
}

constexpr int EU_G = 16, EU_CAP = 64, EU_KC = 32, EU_RC = 128, EU_RM = 8;
struct EulMem {
  GroupMem<EU_CAP, EU_KC, EU_RC, EU_RM> g;
  double sx[EU_RC], sy[EU_RC];      // regions of floe ∩ cell while they are clipped against topography
  int soff[EU_RM + 1];
};
// _make_bounding_box_polygon: (xmin,ymin) (xmin,ymax) (xmax,ymax) (xmax,ymin) (xmin,ymin)
template <class MEM>
__device__ __forceinline__ void eul_cell_ring(MEM& m, int gl, double xmin, double xmax, double ymin, double ymax) {
  if (gl < 5) { m.ax[gl] = (gl == 2 || gl == 3) ? xmax : xmin; m.ay[gl] = (gl == 1 || gl == 2) ? ymax : ymin; }
}
// area of (ring a of m, n points, box ba) ∩ topography element te; 0 if the boxes miss
template <class MEM>
__device__ __forceinline__ double eul_clip_topo(const State& S, MEM& m, int gl, int n, const Box& ba, int te, Stamps& st, bool& cap_err) {
  const int e = 4 + te;
  const Box bt{ S.ebb[4 * e], S.ebb[4 * e + 1], S.ebb[4 * e + 2], S.ebb[4 * e + 3] };
  if (ba.x1 < bt.x0 || bt.x1 < ba.x0 || ba.y1 < bt.y0 || bt.y1 < ba.y0) return 0.0;
  const int eo = S.eoff[e], ne = S.eoff[e + 1] - eo;
  if (ne > EU_CAP) { cap_err = true; return 0.0; }
  gsync();
  for (int k = gl; k < ne; k += EU_G) { m.bx[k] = S.ex[eo + k]; m.by[k] = S.ey[eo + k]; }
  gsync();
  const int oa = ring_signed_area(m.ax, m.ay, n) >= 0.0 ? 1 : -1;
  clip<EU_G>(m, gl, 0.0, 0.0, n, oa, ne, (int)S.eosign[e], 0, ba, bt, st);
  gsync();
  double a = 0.0;
  const int nreg = m.nreg[0];
  for (int r = 0; r < nreg; r++) a += m.rarea[0][r];
  return a;
}
__global__ void __launch_bounds__(64) sz_k_eul_cell_area(State S, EulGrid E) {
  constexpr int GPB = 64 / EU_G;
  __shared__ EulMem mem[GPB];
  const int gl = threadIdx.x % EU_G, gi = threadIdx.x / EU_G;
  auto& m = mem[gi].g;
  const int ncell = E.nx * E.ny, ntopo = S.nelem - 4;
  if (gl == 0) { m.err = 0; m.ntracefail = 0; }
  Stamps st; STAMP_INIT(st);
  bool cap_err = false;
  for (int q0 = blockIdx.x * GPB; q0 < ncell; q0 += gridDim.x * GPB) {
    const int q = q0 + gi;
    if (q >= ncell) continue;
    const int ix = q / E.ny, iy = q % E.ny;
    const double xmin = E.xg[ix], xmax = E.xg[ix + 1], ymin = E.yg[iy], ymax = E.yg[iy + 1];
    const Box bc{ xmin, xmax, ymin, ymax };
    gsync();
    eul_cell_ring(m, gl, xmin, xmax, ymin, ymax);
    gsync();
    double a = fabs(ring_signed_area(m.ax, m.ay, 5));
    for (int te = 0; te < ntopo; te++) a -= eul_clip_topo(S, m, gl, 5, bc, te, st, cap_err);
    if (gl == 0) E.cell_area[q] = a;
  }
  gsync();
  if (gl == 0 && (cap_err || m.err)) atomicOr(&S.cnt[C_ERR], (cap_err ? ERR_CAP_RING : 0) | (m.err & (ERR_CAP_XING | ERR_CAP_REGION)));
  if (gl == 0 && m.ntracefail) atomicAdd(&S.cnt[C_TRACE_FAIL], m.ntracefail);
}
__global__ void __launch_bounds__(64) sz_k_eul_area(State S, EulGrid E, int nent) {
  constexpr int GPB = 64 / EU_G;
  __shared__ EulMem mem[GPB];
  const int gl = threadIdx.x % EU_G, gi = threadIdx.x / EU_G;
  auto& mm = mem[gi];
  auto& m = mm.g;
  const int ntopo = S.nelem - 4;
  if (gl == 0) { m.err = 0; m.ntracefail = 0; }
  Stamps st; STAMP_INIT(st);
  bool cap_err = false;
  for (int t0 = blockIdx.x * GPB; t0 < nent; t0 += gridDim.x * GPB) {
    const int t = t0 + gi;
    if (t >= nent) continue;
    const unsigned long long key = E.keys[t];
    const int i = (int)(key % (unsigned long long)E.M), q = (int)(key / (unsigned long long)E.M);
    const int ix = q / E.ny, iy = q % E.ny;
    const double xmin = E.xg[ix], xmax = E.xg[ix + 1], ymin = E.yg[iy], ymax = E.yg[iy + 1];
    const int bo = S.voff[i], nb = S.voff[i + 1] - bo;
    gsync();
    if (nb > EU_CAP) { cap_err = true; if (gl == 0) E.pic[t] = 0.0; continue; }
    eul_cell_ring(m, gl, xmin, xmax, ymin, ymax);
    for (int k = gl; k < nb; k += EU_G) { const double2 p = S.vxy[bo + k]; m.bx[k] = p.x; m.by[k] = p.y; }
    gsync();
    const Box bc{ xmin, xmax, ymin, ymax };
    const Box bf{ S.bbx0[i], S.bbx1[i], S.bby0[i], S.bby1[i] };
    const int oc = ring_signed_area(m.ax, m.ay, 5) >= 0.0 ? 1 : -1;
    // regions into buffer 1: it keeps the ring of a contained floe / cell too
    clip<EU_G>(m, gl, 0.0, 0.0, 5, oc, nb, (int)S.osign[i], 1, bc, bf, st);
    gsync();
    double a = 0.0;
    const int nreg = m.nreg[1];
    for (int r = 0; r < nreg; r++) a += m.rarea[1][r];
    if (ntopo > 0 && a > 0) {
      const Box bq{ fmax(xmin, bf.x0), fmin(xmax, bf.x1), fmax(ymin, bf.y0), fmin(ymax, bf.y1) };
      bool any = false;
      for (int te = 0; te < ntopo; te++) {
        const int e = 4 + te;
        any |= !(bq.x1 < S.ebb[4 * e] || S.ebb[4 * e + 1] < bq.x0 || bq.y1 < S.ebb[4 * e + 2] || S.ebb[4 * e + 3] < bq.y0);
      }
      if (any) {
        // the second clip reuses the region buffers: park the regions first
        const int tot = m.roff[1][nreg];
        for (int k = gl; k < tot; k += EU_G) { mm.sx[k] = m.reg[1][0][k]; mm.sy[k] = m.reg[1][1][k]; }
        if (gl <= nreg) mm.soff[gl] = m.roff[1][gl];
        gsync();
        for (int r = 0; r < nreg; r++) {
          const int s0 = mm.soff[r], n = mm.soff[r + 1] - s0;
          if (n > EU_CAP) { cap_err = true; continue; }
          double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
          for (int k = gl; k < n; k += EU_G) {
            const double x = mm.sx[s0 + k], y = mm.sy[s0 + k];
            x0 = fmin(x0, x); x1 = fmax(x1, x); y0 = fmin(y0, y); y1 = fmax(y1, y);
          }
          const Box br{ gmin<EU_G>(x0), gmax<EU_G>(x1), gmin<EU_G>(y0), gmax<EU_G>(y1) };
          for (int te = 0; te < ntopo; te++) {
            gsync();
            for (int k = gl; k < n; k += EU_G) { m.ax[k] = mm.sx[s0 + k]; m.ay[k] = mm.sy[s0 + k]; }
            gsync();
            a -= eul_clip_topo(S, m, gl, n, br, te, st, cap_err);
          }
        }
      }
    }
    if (gl == 0) E.pic[t] = a;
  }
  gsync();
  if (gl == 0 && (cap_err || m.err)) atomicOr(&S.cnt[C_ERR], (cap_err ? ERR_CAP_RING : 0) | (m.err & (ERR_CAP_XING | ERR_CAP_REGION)));
  if (gl == 0 && m.ntracefail) atomicAdd(&S.cnt[C_TRACE_FAIL], m.ntracefail);
}

// without topography the window is a plain rectangle: one thread per entry, the rectangle pipeline of the two-way
// coupling (sz_twoway.hpp)
__global__ void __launch_bounds__(256) sz_k_eul_area_rect(State S, EulGrid E, int nent) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nent; t += gridDim.x * blockDim.x) {
    const unsigned long long key = E.keys[t];
    const int i = (int)(key % (unsigned long long)E.M), q = (int)(key / (unsigned long long)E.M);
    const int ix = q / E.ny, iy = q % E.ny;
    const double xmin = E.xg[ix], ymin = E.yg[iy];
    const int bo = S.voff[i], nb = S.voff[i + 1] - bo;
    RectClip rc;
    rc.w = E.xg[ix + 1] - xmin; rc.h = E.yg[iy + 1] - ymin; rc.acc = 0.0; rc.hout = false;
    rc.ox0 = rc.oy0 = rc.oxp = rc.oyp = 0.0;
    for (int k = 0; k < 4; k++) { rc.have[k] = false; rc.fx[k] = rc.fy[k] = rc.px[k] = rc.py[k] = 0.0; }
    for (int k = 0; k + 1 < nb; k++) { const double2 p = S.vxy[bo + k]; rc.feed<0>(p.x - xmin, p.y - ymin); }
    E.pic[t] = rc.finish();
  }
}

__device__ __forceinline__ int eul_lower_bound(const unsigned long long* k, int n, unsigned long long v) {
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (k[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}
// the averages of one cell, floes in ascending order (output.jl:839-905).  Ghost rows only carry the columns the
// collision path needs; the rest is read from the parent row (a ghost is a deep copy, collisions.jl:881-901).
__global__ void sz_k_eul_reduce(State S, EulGrid E, int nent) {
  const int ncell = E.nx * E.ny, N = S.cnt[C_N];
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < ncell; q += gridDim.x * blockDim.x) {
    double s[EUL_COUNT];
    for (int k = 0; k < EUL_COUNT; k++) s[k] = 0.0;
    const double cell_area = E.cell_area[q];
    const unsigned long long base = (unsigned long long)q * (unsigned long long)E.M;
    const int lo = eul_lower_bound(E.keys, nent, base), hi = eul_lower_bound(E.keys, nent, base + (unsigned long long)E.M);
    double area_tot = 0.0, mass_tot = 0.0; int m = 0;
    if (cell_area > 0) {
      for (int t = lo; t < hi; t++) {
        const double a = E.pic[t];
        if (!(a > 0)) continue;
        const int i = (int)(E.keys[t] - base);
        area_tot += a; mass_tot += S.mass[i] * (a / S.area[i]); m++;
      }
    }
    if (mass_tot > 0) {
      double over = 0.0;
      for (int t = lo; t < hi; t++) {
        const double a = E.pic[t];
        if (!(a > 0)) continue;
        const int i = (int)(E.keys[t] - base);
        const int p = i < N ? i : S.parent[i];
        const double r = (a / S.area[i]) * (S.mass[i] / mass_tot);      // ma_ratios
        s[0] += S.u[i] * r; s[1] += S.v[i] * r; s[2] += S.p_dudt[p] * r; s[3] += S.p_dvdt[p] * r;
        s[7] += S.height[i] * r;
        s[9] += S.sa[4 * p] * r; s[10] += S.sa[4 * p + 1] * r; s[11] += S.sa[4 * p + 2] * r; s[12] += S.sa[4 * p + 3] * r;
        s[14] += S.strain[4 * p] * r; s[15] += S.strain[4 * p + 1] * r; s[16] += S.strain[4 * p + 2] * r; s[17] += S.strain[4 * p + 3] * r;
        over += S.overarea[i];
      }
      s[4] = over / m; s[5] = mass_tot; s[6] = area_tot; s[8] = area_tot / cell_area;
      // maximum(eigvals([xx yx; xy yy])) of the symmetric 2 x 2, zeroed beyond 1e8 (:882-891)
      const double hm = 0.5 * (s[9] + s[12]), hd = 0.5 * (s[9] - s[12]);
      double e = hm + sqrt(hd * hd + s[11] * s[10]);
      if (fabs(e) > 1e8) e = 0.0;
      s[13] = e;
    }
    for (int k = 0; k < EUL_COUNT; k++) E.data[(size_t)k * ncell + q] = s[k];
  }
}

// tiled runs: a floe (and its ghosts) lives on the rank that owns it, so every rank sums what its floes put into
// each cell -- area, mass, entry count, overarea and the 13 weighted sums, EUL_PARTIAL fields per cell, field-major
// -- the host adds the buffers up across the ranks (all-reduce) and every rank finishes the cells.  The weights are
// the reference's ma_ratios with the division by the cell's mass moved to the end (the same numbers up to round-off).
constexpr int EUL_PARTIAL = 17;
__global__ void sz_k_eul_partial(State S, EulGrid E, int nent, double* partial) {
  const int ncell = E.nx * E.ny, N = S.cnt[C_N];
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < ncell; q += gridDim.x * blockDim.x) {
    double s[EUL_PARTIAL];
    for (int k = 0; k < EUL_PARTIAL; k++) s[k] = 0.0;
    const unsigned long long base = (unsigned long long)q * (unsigned long long)E.M;
    const int lo = eul_lower_bound(E.keys, nent, base), hi = eul_lower_bound(E.keys, nent, base + (unsigned long long)E.M);
    for (int t = lo; t < hi; t++) {
      const double a = E.pic[t];
      if (!(a > 0)) continue;
      const int i = (int)(E.keys[t] - base);
      const int p = i < N ? i : S.parent[i];
      const double r = S.mass[i] * (a / S.area[i]);
      s[0] += a; s[1] += r; s[2] += 1.0; s[3] += S.overarea[i];
      s[4] += S.u[i] * r; s[5] += S.v[i] * r; s[6] += S.p_dudt[p] * r; s[7] += S.p_dvdt[p] * r; s[8] += S.height[i] * r;
      for (int c = 0; c < 4; c++) { s[9 + c] += S.sa[4 * p + c] * r; s[13 + c] += S.strain[4 * p + c] * r; }
    }
    for (int k = 0; k < EUL_PARTIAL; k++) partial[(size_t)k * ncell + q] = s[k];
  }
}
__global__ void sz_k_eul_finish(EulGrid E, const double* partial) {
  const int ncell = E.nx * E.ny;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < ncell; q += gridDim.x * blockDim.x) {
    double s[EUL_COUNT];
    for (int k = 0; k < EUL_COUNT; k++) s[k] = 0.0;
    const double cell_area = E.cell_area[q];
    const double area_tot = partial[q], mass_tot = partial[(size_t)ncell + q], cnt = partial[2 * (size_t)ncell + q];
    if (cell_area > 0 && mass_tot > 0) {
      auto w = [&](int f) { return partial[(size_t)f * ncell + q] / mass_tot; };
      s[0] = w(4); s[1] = w(5); s[2] = w(6); s[3] = w(7); s[7] = w(8);
      s[9] = w(9); s[10] = w(10); s[11] = w(11); s[12] = w(12);
      s[14] = w(13); s[15] = w(14); s[16] = w(15); s[17] = w(16);
      s[4] = partial[3 * (size_t)ncell + q] / cnt; s[5] = mass_tot; s[6] = area_tot; s[8] = area_tot / cell_area;
      const double hm = 0.5 * (s[9] + s[12]), hd = 0.5 * (s[9] - s[12]);
      double e = hm + sqrt(hd * hd + s[11] * s[10]);
      if (fabs(e) > 1e8) e = 0.0;
      s[13] = e;
    }
    for (int k = 0; k < EUL_COUNT; k++) E.data[(size_t)k * ncell + q] = s[k];
  }
}

// simplify_floes! has work if any of these is non-zero: floes tagged remove, tagged fuse, rings with more than
// max_vertices points (GI.npoint counts the closing point, simplification.jl:66), floes not tagged remove under
// the minimum area / height (:287-290)
__global__ void sz_k_simplify_check(State S, int max_vertices, double min_area, double min_height, unsigned long long* out4) {
  const int M = S.cnt[C_M];
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
    const int st = S.status[i];
    c0 += st == SZ_REMOVE; c1 += st == SZ_FUSE;
    c2 += (S.voff[i + 1] - S.voff[i]) > max_vertices;
    c3 += st != SZ_REMOVE && (S.area[i] < min_area || S.height[i] < min_height);
  }
  for (int d = 32; d >= 1; d >>= 1) { c0 += __shfl_xor(c0, d); c1 += __shfl_xor(c1, d); c2 += __shfl_xor(c2, d); c3 += __shfl_xor(c3, d); }
  if ((threadIdx.x & 63) == 0) {
    if (c0) atomicAdd(&out4[0], (unsigned long long)c0);
    if (c1) atomicAdd(&out4[1], (unsigned long long)c1);
    if (c2) atomicAdd(&out4[2], (unsigned long long)c2);
    if (c3) atomicAdd(&out4[3], (unsigned long long)c3);
  }
}

}  // namespace sz
