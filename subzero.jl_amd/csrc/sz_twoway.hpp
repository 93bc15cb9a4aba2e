// sz_twoway.hpp — ice-on-ocean stress per centre cell: calc_two_way_coupling! (coupling.jl:1617-1680).
// The forcing kernel (sz_k_forcing<true>) has left, per floe, the centre cells its sub-floe points fell into
// with the summed ocean stress (floe_to_grid_info!).  Here: the per-cell lists (a counting sort over the
// (Nx+1) x (Ny+1) centre cells, entries ordered by floe index = the order the serial reference pushes them),
// the area of floe ∩ cell for every entry (the same group-cooperative clip as the contact path, the cell
// rectangle as ring a, the translated floe ring as ring b: intersect_polys(cell_poly, floe_poly)), and the
// per-cell reduction incl. the atmosphere-on-ocean stress and the heat-flux factor.
#pragma once
#include "sz_kernels.hpp"

namespace sz {

__global__ void sz_k_tw_count(State S) {
  if (stopped(S)) return;          // (a step behind the one that ended the batch: its launches are enqueued and return at once, like the forcing kernel's)
  if (stopped(S)) return;
  int N = S.cnt[C_NOWN];
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)N * FC_CAP; t += (long long)gridDim.x * blockDim.x) {
    int i = (int)(t / FC_CAP), s = (int)(t % FC_CAP);
    if (s < S.fc_cnt[i]) atomicAdd(&S.cl_cnt[S.fc_key[t]], 1);
  }
}
__global__ void sz_k_tw_fill(State S) {
  if (stopped(S)) return;          // (a step behind the one that ended the batch: its launches are enqueued and return at once, like the forcing kernel's)
  if (stopped(S)) return;
  int N = S.cnt[C_NOWN];
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)N * FC_CAP; t += (long long)gridDim.x * blockDim.x) {
    int i = (int)(t / FC_CAP), s = (int)(t % FC_CAP);
    if (s < S.fc_cnt[i]) { int q = S.fc_key[t]; S.cl_ent[S.cl_off[q] + atomicAdd(&S.cl_cur[q], 1)] = (int)t; }
  }
}
// every cell's entries in the order the serial reference meets them: floe index ascending, then slot
// (= first appearance among the floe's points); the lists are a handful of entries long
__global__ void sz_k_tw_sort(State S, int ncell) {
  if (stopped(S)) return;          // (a step behind the one that ended the batch: its launches are enqueued and return at once, like the forcing kernel's)
  if (stopped(S)) return;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < ncell; q += gridDim.x * blockDim.x) {
    int lo = S.cl_off[q], hi = S.cl_off[q + 1];
    for (int a = lo + 1; a < hi; a++) {
      int v = S.cl_ent[a], b = a - 1;
      while (b >= lo && S.cl_ent[b] > v) { S.cl_ent[b + 1] = S.cl_ent[b]; b--; }
      S.cl_ent[b + 1] = v;
    }
    S.cl_cnt[q] = 0; S.cl_cur[q] = 0;          // ready for the next coupling step
  }
}
// centre cell rectangle: center_cell_coords (coupling.jl:1116-1140) with check_cell_bounds (:931-1087)
__device__ __forceinline__ void center_cell(const State& S, int ix0, int iy0, int per_x, int per_y, double& xmin, double& xmax,
                                            double& ymin, double& ymax) {
  xmin = ((ix0 + 1) - 1.5) * S.gdx + S.gx0; xmax = xmin + S.gdx;
  ymin = ((iy0 + 1) - 1.5) * S.gdy + S.gy0; ymax = ymin + S.gdy;
  if (!per_x) {
    xmin = xmin < S.gx0 ? S.gx0 : (xmin > S.gxf ? S.gxf : xmin);
    xmax = xmax > S.gxf ? S.gxf : (xmax < S.gx0 ? S.gx0 : xmax);
  }
  if (!per_y) {
    ymin = ymin < S.gy0 ? S.gy0 : (ymin > S.gyf ? S.gyf : ymin);
    ymax = ymax > S.gyf ? S.gyf : (ymax < S.gy0 ? S.gy0 : ymax);
  }
}
// floe_area_in_cell of every (floe, cell) entry: G lanes per entry
constexpr int TW_G = 8, TW_CAP = 32, TW_KC = 16, TW_RC = 64, TW_RM = 6;
__global__ void __launch_bounds__(64) sz_k_tw_area(State S) {
  if (stopped(S)) return;          // (a step behind the one that ended the batch: its launches are enqueued and return at once, like the forcing kernel's)
  if (stopped(S)) return;
  constexpr int GPB = 64 / TW_G;
  __shared__ GroupMem<TW_CAP, TW_KC, TW_RC, TW_RM> mem[GPB];
  const int gl = threadIdx.x % TW_G, gi = threadIdx.x / TW_G;
  auto& m = mem[gi];
  const int nent = S.cnt[C_NENT];
  const int per_x = S.ekind[2] == 1, per_y = S.ekind[0] == 1;
  if (gl == 0) { m.err = 0; m.ntracefail = 0; }
  Stamps st; STAMP_INIT(st);
  for (int t0 = blockIdx.x * GPB; t0 < nent; t0 += gridDim.x * GPB) {
    const int t = t0 + gi;
    if (t >= nent) continue;
    const int ent = S.cl_ent[t];
    const int i = ent / FC_CAP, q = S.fc_key[ent], code = S.fc_code[ent];
    const int ix0 = q / (S.Ny + 1), iy0 = q % (S.Ny + 1);
    // (shifted_idx - idx) * grid.Δ, coupling.jl:1432-1433: the shift is one grid length or none
    const double dx = (double)((code % 3 - 1) * S.Nx) * S.gdx, dy = (double)((code / 3 - 1) * S.Ny) * S.gdy;
    double xmin, xmax, ymin, ymax;
    center_cell(S, ix0, iy0, per_x, per_y, xmin, xmax, ymin, ymax);
    const int bo = S.voff[i], nb = S.voff[i + 1] - bo;
    gsync();
    if (nb > TW_CAP) { if (gl == 0) { atomicOr(&S.cnt[C_ERR], ERR_CAP_RING); S.fc_area[ent] = 0.0; } continue; }
    // _make_bounding_box_polygon: (xmin,ymin) (xmin,ymax) (xmax,ymax) (xmax,ymin) (xmin,ymin)
    if (gl < 5) { m.ax[gl] = (gl == 2 || gl == 3) ? xmax : xmin; m.ay[gl] = (gl == 1 || gl == 2) ? ymax : ymin; }
    for (int k = gl; k < nb; k += TW_G) { const double2 p = S.vxy[bo + k]; m.bx[k] = p.x + dx; m.by[k] = p.y + dy; }   // _translate_poly
    gsync();
    const Box ba{ xmin, xmax, ymin, ymax };
    const Box bb{ S.bbx0[i] + dx, S.bbx1[i] + dx, S.bby0[i] + dy, S.bby1[i] + dy };
    const int oa = ring_signed_area(m.ax, m.ay, 5) >= 0.0 ? 1 : -1;   // as sz_k_osign does for floe rings (listed clockwise: -1)
    clip<TW_G>(m, gl, 0.0, 0.0, 5, oa, nb, (int)S.osign[i], 0, ba, bb, st);
    gsync();
    double a = 0.0;
    const int nreg = m.nreg[0];
    for (int r = 0; r < nreg; r++) a += m.rarea[0][r];
    if (gl == 0) S.fc_area[ent] = a;
  }
  gsync();
  if (gl == 0 && (m.err & (ERR_CAP_XING | ERR_CAP_REGION))) atomicOr(&S.cnt[C_ERR], m.err);
  if (gl == 0 && m.ntracefail) atomicAdd(&S.cnt[C_TRACE_FAIL], m.ntracefail);
}
// ---- the same areas, one thread per entry.  The clip window is an axis-parallel rectangle, so the intersection
// is what the re-entrant Sutherland-Hodgman pipeline produces: every ring vertex is pushed through the four
// half-plane stages (left, right, bottom, top), each stage remembers its first and previous point only, and what
// leaves the last stage is summed into the shoelace area on the fly -- no polygon is stored, no LDS, no lane sits
// idle behind a neighbour's different path.  For a floe that leaves the window in several pieces the pipeline
// emits one ring whose connecting edges run along the window's boundary back and forth and cancel in the area,
// so the sum equals the total of the pieces (what the reference adds up over intersect_polys' regions).
// Coordinates are taken relative to the window's corner: the result carries less round-off than the general
// clipper's (differences of a few 1e-11 relative at 2000 km; the tests state 1e-9).
struct RectClip {
  double w, h;                  // window [0, w] x [0, h]
  double fx[4], fy[4], px[4], py[4];
  bool have[4];
  double ox0, oy0, oxp, oyp, acc; bool hout;
  __device__ __forceinline__ bool inside(int s, double x, double y) const {
    return s == 0 ? x >= 0.0 : s == 1 ? x <= w : s == 2 ? y >= 0.0 : y <= h;
  }
  __device__ __forceinline__ void cross(int s, double ax, double ay, double bx, double by, double& ix, double& iy) const {
    if (s < 2) { const double c = s == 0 ? 0.0 : w; const double t = (c - ax) / (bx - ax); ix = c; iy = ay + t * (by - ay); }
    else { const double c = s == 2 ? 0.0 : h; const double t = (c - ay) / (by - ay); iy = c; ix = ax + t * (bx - ax); }
  }
  __device__ __forceinline__ void out(double x, double y) {
    if (!hout) { ox0 = x; oy0 = y; hout = true; } else acc += oxp * y - x * oyp;
    oxp = x; oyp = y;
  }
  template <int SS>
  __device__ __forceinline__ void feed(double x, double y) {
    if constexpr (SS == 4) out(x, y);
    else {
      if (!have[SS]) { fx[SS] = x; fy[SS] = y; have[SS] = true; }
      else if (inside(SS, px[SS], py[SS]) != inside(SS, x, y)) { double ix, iy; cross(SS, px[SS], py[SS], x, y, ix, iy); feed<SS + 1>(ix, iy); }
      px[SS] = x; py[SS] = y;
      if (inside(SS, x, y)) feed<SS + 1>(x, y);
    }
  }
  template <int SS>
  __device__ __forceinline__ void close_from() {
    if constexpr (SS < 4) {
      if (have[SS] && inside(SS, px[SS], py[SS]) != inside(SS, fx[SS], fy[SS])) {
        double ix, iy; cross(SS, px[SS], py[SS], fx[SS], fy[SS], ix, iy); feed<SS + 1>(ix, iy);
      }
      close_from<SS + 1>();
    }
  }
  __device__ __forceinline__ double finish() {
    close_from<0>();
    if (!hout) return 0.0;
    acc += oxp * oy0 - ox0 * oyp;
    return fabs(acc) * 0.5;
  }
};
__global__ void __launch_bounds__(256) sz_k_tw_area_rect(State S) {
  if (stopped(S)) return;          // (a step behind the one that ended the batch: its launches are enqueued and return at once, like the forcing kernel's)
  if (stopped(S)) return;
  const int nent = S.cnt[C_NENT];
  const int per_x = S.ekind[2] == 1, per_y = S.ekind[0] == 1;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nent; t += gridDim.x * blockDim.x) {
    const int ent = S.cl_ent[t];
    const int i = ent / FC_CAP, q = S.fc_key[ent], code = S.fc_code[ent];
    const int ix0 = q / (S.Ny + 1), iy0 = q % (S.Ny + 1);
    const double dx = (double)((code % 3 - 1) * S.Nx) * S.gdx, dy = (double)((code / 3 - 1) * S.Ny) * S.gdy;
    double xmin, xmax, ymin, ymax;
    center_cell(S, ix0, iy0, per_x, per_y, xmin, xmax, ymin, ymax);
    const int bo = S.voff[i], nb = S.voff[i + 1] - bo;
    RectClip rc;
    rc.w = xmax - xmin; rc.h = ymax - ymin; rc.acc = 0.0; rc.hout = false;
    rc.ox0 = rc.oy0 = rc.oxp = rc.oyp = 0.0;
    for (int k = 0; k < 4; k++) { rc.have[k] = false; rc.fx[k] = rc.fy[k] = rc.px[k] = rc.py[k] = 0.0; }
    double a = 0.0;
    // the ring box decides the two cheap cases: no overlap, and (for a window inside the box nothing is known)
    const double b0 = S.bbx0[i] + dx, b1 = S.bbx1[i] + dx, b2 = S.bby0[i] + dy, b3 = S.bby1[i] + dy;
    if (!(b1 < xmin || xmax < b0 || b3 < ymin || ymax < b2) && rc.w > 0.0 && rc.h > 0.0) {
      for (int k = 0; k + 1 < nb; k++) { const double2 p = S.vxy[bo + k]; rc.feed<0>((p.x + dx) - xmin, (p.y + dy) - ymin); }   // _translate_poly, then window coordinates
      a = rc.finish();
    }
    S.fc_area[ent] = a;
  }
}

// per centre cell, first half: sums over the floes in it, in floe order -- the ice stress weighted by the area of
// floe in cell (numerators) and that area (coupling.jl:1631-1662)
__device__ __forceinline__ void tw_cell_sums(const State& S, int q, double& tx, double& ty, double& si) {
  tx = 0.0; ty = 0.0; si = 0.0;
  const int lo = S.cl_off[q], hi = S.cl_off[q + 1];
  for (int a = lo; a < hi;) {
    // one entry per floe and cell: a floe that reached this cell through two different unshifted cells was
    // merged by add_point! under its first shift (coupling.jl:1345)
    const int e0 = S.cl_ent[a], i = e0 / FC_CAP;
    double ex = S.fc_tx[e0], ey = S.fc_ty[e0]; int n = S.fc_n[e0];
    int b = a + 1;
    while (b < hi && S.cl_ent[b] / FC_CAP == i) { int e = S.cl_ent[b]; ex += S.fc_tx[e]; ey += S.fc_ty[e]; n += S.fc_n[e]; b++; }
    const double area = S.fc_area[e0];
    if (area > 0) { tx += (ex / n) * area; ty += (ey / n) * area; si += area; }
    a = b;
  }
}
// second half: area-weighted mean, sea-ice fraction, atmosphere-on-ocean stress on the open part, heat-flux factor
// (coupling.jl:1663-1677)
__device__ __forceinline__ void tw_cell_finish(State& S, const Params& P, int q, double tx, double ty, double si, int dt) {
  const double cell_area = S.gdx * S.gdy;
  if (si > 0) { tx /= si; ty /= si; si /= cell_area; }
  const double du = S.ua[q] - S.uo[q], dv = S.va[q] - S.vo[q];
  const double ocn_frac = 1 - si;
  const double nrm = sqrt(du * du + dv * dv);
  tx += P.rho_a * P.Cd_ao * ocn_frac * nrm * du;
  ty += P.rho_a * P.Cd_ao * ocn_frac * nrm * dv;
  S.tau_x[q] = tx; S.tau_y[q] = ty; S.si_frac[q] = si;
  const double hf = dt * P.k_ice / (P.rho_i * P.L_ice) * (S.t_ocn[q] - S.t_atm[q]);
  S.hf[q] = hf; S.nodes[(size_t)q * 8 + 2] = hf;
}
__global__ void sz_k_tw_reduce(State S, Params P, int ncell, int dt) {
  if (stopped(S)) return;          // (a step behind the one that ended the batch: its launches are enqueued and return at once, like the forcing kernel's)
  if (stopped(S)) return;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < ncell; q += gridDim.x * blockDim.x) {
    double tx, ty, si;
    tw_cell_sums(S, q, tx, ty, si);
    tw_cell_finish(S, P, q, tx, ty, si, dt);
  }
}
// tiled runs: every rank sums over the floes it owns (partial[q], [ncell + q], [2 ncell + q]), the host adds the
// partial fields up across the ranks (all-reduce), then every rank finishes the cells
__global__ void sz_k_tw_partial(State S, int ncell, double* partial) {
  if (stopped(S)) return;          // (a step behind the one that ended the batch: its launches are enqueued and return at once, like the forcing kernel's)
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < ncell; q += gridDim.x * blockDim.x) {
    double tx, ty, si;
    tw_cell_sums(S, q, tx, ty, si);
    partial[q] = tx; partial[ncell + q] = ty; partial[2 * (size_t)ncell + q] = si;
  }
}
__global__ void sz_k_tw_finish(State S, Params P, int ncell, int dt, const double* partial) {
  if (stopped(S)) return;          // (a step behind the one that ended the batch: its launches are enqueued and return at once, like the forcing kernel's)
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < ncell; q += gridDim.x * blockDim.x)
    tw_cell_finish(S, P, q, partial[q], partial[ncell + q], partial[2 * (size_t)ncell + q], dt);
}

}  // namespace sz
