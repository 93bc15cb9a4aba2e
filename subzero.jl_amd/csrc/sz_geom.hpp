// sz_geom.hpp — group-cooperative polygon clipping and contact-force evaluation for gfx950.
//
// One "group" of G lanes (G = 16 or 64, a power of two <= the 64-wide wavefront) works on one
// work item (a floe-floe pair or a floe-domain-element pair).  Both vertex rings are staged in
// LDS; edge-edge crossing detection, nearest-vertex matching, edge classification and the
// region-region `intersects` test are partitioned over the lanes, while the short serial parts
// (boundary tracing, shoelace sums in ring order) are evaluated redundantly by every lane with
// only lane 0 storing, so that no value has to be broadcast and all control flow is uniform
// inside a group.  Lanes of one wavefront execute in lockstep and a wave's LDS operations
// complete in order, so groups only need a compiler-level fence between LDS phases (gsync()).
//
// The arithmetic is fp64 without FMA contraction (-ffp-contract=off) and follows, expression
// by expression, the semantics of the reference functions:
//   intersect_polys / GO.intersection      src/floe_utils.jl:55          -> clip()
//   GO.intersection_points                 collisions.jl:156             -> crossings (unique)
//   GO.area / GO.centroid                  collisions.jl:360,178         -> ring_area(), ring_centroid()
//   which_vertices_match_points            src/floe_utils.jl:331-352     -> match_vertices()
//   calc_normal_force                      collisions.jl:30-70           -> normal_force()
//   _many_intersect_normal_force!          collisions.jl:78-119          -> many_intersect()
//   calc_elastic_forces                    collisions.jl:149-188         -> collide()
//   calc_friction_forces                   collisions.jl:243-283         -> friction()
// Degenerate contacts use the symbolic perturbation "ring b translated by eps*(1, delta)"
// (DESIGN.md §3.2); it reproduces the reference's known answers for its degenerate tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace szg {

#define SZ_DEV __device__ __forceinline__

SZ_DEV void gsync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

constexpr int RMAX = 6;   // regions kept per clip

// LDS working set of one group.  CAP: ring points per polygon, KC: crossings, RC: region points.
template <int CAP, int KC, int RC>
struct GroupMem {
  double ax[CAP], ay[CAP], bx[CAP], by[CAP];
  double tx[CAP], ty[CAP];                 // p1 translated by the force direction
  // crossings in canonical (ia, ib) order
  double cta[KC], ctb[KC], cx[KC], cy[KC];
  // raw detection slots
  double rta[KC], rtb[KC], rcx[KC], rcy[KC];
  double earr[RC], farr[RC], garr[RC];     // per-edge scratch (many-intersect)
  double rx[2][RC], ry[2][RC];             // region rings of clip 0 (contact) and clip 1 (check)
  double rarea[2][RMAX];
  double red[64 * 4];                      // per-lane partials (bbox)
  int16_t cia[KC], cib[KC], ria[KC], rib[KC];
  int16_t ordA[KC], ordB[KC], rnkA[KC], rnkB[KC];
  int16_t midx[KC];                        // matched region-vertex index per ipoint
  int16_t roff[2][RMAX + 2];
  uint8_t cfl[KC], rfl[KC], uniq[KC];
  int nraw, nx, nreg[2], flag, err, ntracefail;
};

enum { ERR_CAP_RING = 1, ERR_CAP_XING = 2, ERR_CAP_REGION = 4, ERR_CAP_ROWS = 8, ERR_TRACE = 16,
       ERR_CAP_NEIGH = 32, ERR_CAP_PAIRS = 64, ERR_CAP_ELEM = 128, ERR_CAP_INTER = 256,
       ERR_CAP_FLOES = 512, ERR_CAP_VERTS = 1024, ERR_CAP_CELLS = 2048, ERR_GHOSTS_PER_PARENT = 4096 };

SZ_DEV double orient(double ax, double ay, double bx, double by, double cx, double cy) {
  return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}
// a-point p against b-edge r->s; b is the symbolically translated ring
SZ_DEV int side_a_vs_b(double rx, double ry, double sx, double sy, double px, double py) {
  double o = orient(rx, ry, sx, sy, px, py);
  if (o > 0.0) return 1;
  if (o < 0.0) return -1;
  double dx = sx - rx, dy = sy - ry;
  if (dy != 0.0) return dy > 0.0 ? 1 : -1;
  return dx > 0.0 ? -1 : 1;
}
// b-point r against a-edge p->q
SZ_DEV int side_b_vs_a(double px, double py, double qx, double qy, double rx, double ry) {
  double o = orient(px, py, qx, qy, rx, ry);
  if (o > 0.0) return 1;
  if (o < 0.0) return -1;
  double dx = qx - px, dy = qy - py;
  if (dy != 0.0) return dy > 0.0 ? -1 : 1;
  return dx > 0.0 ? 1 : -1;
}
SZ_DEV bool on_segment(double ax, double ay, double bx, double by, double cx, double cy) {
  return fmin(ax, bx) <= cx && cx <= fmax(ax, bx) && fmin(ay, by) <= cy && cy <= fmax(ay, by);
}
// 1 inside, 0 outside, -1 on the boundary (ring closed, n points)
SZ_DEV int point_in_ring3(double x, double y, const double* rx, const double* ry, int n) {
  int inside = 0;
  for (int i = 0; i + 1 < n; i++) {
    double ax = rx[i], ay = ry[i], bx = rx[i + 1], by = ry[i + 1];
    double o = orient(ax, ay, bx, by, x, y);
    if (o == 0.0 && on_segment(ax, ay, bx, by, x, y)) return -1;
    if ((ay > y) != (by > y)) {
      if (by > ay) { if (o > 0.0) inside = !inside; }
      else         { if (o < 0.0) inside = !inside; }
    }
  }
  return inside;
}
SZ_DEV bool coveredby(double x, double y, const double* rx, const double* ry, int n) {
  return point_in_ring3(x, y, rx, ry, n) != 0;
}
SZ_DEV bool ring_inside(const double* ax, const double* ay, int na, const double* bx, const double* by, int nb) {
  for (int i = 0; i + 1 < na; i++) {
    int c = point_in_ring3(ax[i], ay[i], bx, by, nb);
    if (c >= 0) return c != 0;
  }
  return true;
}
SZ_DEV double dist_pt_seg(double x0, double y0, double ax, double ay, double bx, double by) {
  double vx = bx - ax, vy = by - ay;
  double wx = x0 - ax, wy = y0 - ay;
  double c1 = wx * vx + wy * vy;
  if (c1 <= 0.0) return sqrt((x0 - ax) * (x0 - ax) + (y0 - ay) * (y0 - ay));
  double c2 = vx * vx + vy * vy;
  if (c2 <= c1) return sqrt((x0 - bx) * (x0 - bx) + (y0 - by) * (y0 - by));
  double b2 = c1 / c2;
  double px = ax + b2 * vx, py = ay + b2 * vy;
  return sqrt((x0 - px) * (x0 - px) + (y0 - py) * (y0 - py));
}
SZ_DEV double dist_to_ring(double x, double y, const double* rx, const double* ry, int n) {
  double md = __builtin_inf();
  for (int i = 0; i + 1 < n; i++) {
    double d = dist_pt_seg(x, y, rx[i], ry[i], rx[i + 1], ry[i + 1]);
    if (d < md) md = d;
  }
  return md;
}
SZ_DEV bool seg_seg_touch(double px, double py, double qx, double qy, double rx, double ry, double sx, double sy) {
  double o1 = orient(px, py, qx, qy, rx, ry), o2 = orient(px, py, qx, qy, sx, sy);
  double o3 = orient(rx, ry, sx, sy, px, py), o4 = orient(rx, ry, sx, sy, qx, qy);
  if (((o1 > 0 && o2 < 0) || (o1 < 0 && o2 > 0)) && ((o3 > 0 && o4 < 0) || (o3 < 0 && o4 > 0))) return true;
  if (o1 == 0 && on_segment(px, py, qx, qy, rx, ry)) return true;
  if (o2 == 0 && on_segment(px, py, qx, qy, sx, sy)) return true;
  if (o3 == 0 && on_segment(rx, ry, sx, sy, px, py)) return true;
  if (o4 == 0 && on_segment(rx, ry, sx, sy, qx, qy)) return true;
  return false;
}
// GO._signed_area on a closed ring
SZ_DEV double ring_signed_area(const double* x, const double* y, int n) {
  if (n == 0) return 0.0;
  double area = 0.0, p1x = x[0], p1y = y[0];
  for (int i = 1; i < n; i++) {
    double p2x = x[i], p2y = y[i];
    area += p1x * p2y - p1y * p2x;
    p1x = p2x; p1y = p2y;
  }
  area += p1x * y[0] - p1y * x[0];
  return area / 2.0;
}
SZ_DEV void ring_centroid(const double* x, const double* y, int n, double& cx, double& cy) {
  double xc = 0.0, yc = 0.0, area = 0.0, p1x = x[0], p1y = y[0];
  for (int i = 1; i < n; i++) {
    double p2x = x[i], p2y = y[i];
    double ac = p1x * p2y - p2x * p1y;
    area += ac;
    xc += (p1x + p2x) * ac;
    yc += (p1y + p2y) * ac;
    p1x = p2x; p1y = p2y;
  }
  area /= 2.0;
  cx = xc / (6.0 * area);
  cy = yc / (6.0 * area);
}

// ---------------------------------------------------------------------------------------------
// clip(): regions of a ∩ b into buffer `buf` of the group memory.  a = (pax, pay, na) and
// b = (m.bx, m.by, nb) are closed rings in LDS; oa/ob are their orientation signs.
// Group-uniform: every lane must call it with the same arguments.
template <int G, int CAP, int KC, int RC>
SZ_DEV void clip(GroupMem<CAP, KC, RC>& m, int gl, const double* pax, const double* pay, int na,
                 int oa, int nb, int ob, int buf) {
  const double* pbx = m.bx; const double* pby = m.by;
  if (gl == 0) { m.nraw = 0; m.nreg[buf] = 0; m.roff[buf][0] = 0; }
  // ---- bounding boxes (lane partials, then every lane reduces the G partials)
  {
    double x0 = __builtin_inf(), x1 = -__builtin_inf(), y0 = __builtin_inf(), y1 = -__builtin_inf();
    double u0 = x0, u1 = x1, v0 = y0, v1 = y1;
    for (int i = gl; i < na; i += G) { x0 = fmin(x0, pax[i]); x1 = fmax(x1, pax[i]); y0 = fmin(y0, pay[i]); y1 = fmax(y1, pay[i]); }
    for (int i = gl; i < nb; i += G) { u0 = fmin(u0, pbx[i]); u1 = fmax(u1, pbx[i]); v0 = fmin(v0, pby[i]); v1 = fmax(v1, pby[i]); }
    // overlap test needs min/max over the whole group: go through LDS
    m.red[gl * 4 + 0] = x0; m.red[gl * 4 + 1] = x1; m.red[gl * 4 + 2] = y0; m.red[gl * 4 + 3] = y1;
    gsync();
    for (int l = 0; l < G; l++) { x0 = fmin(x0, m.red[l * 4]); x1 = fmax(x1, m.red[l * 4 + 1]); y0 = fmin(y0, m.red[l * 4 + 2]); y1 = fmax(y1, m.red[l * 4 + 3]); }
    gsync();
    m.red[gl * 4 + 0] = u0; m.red[gl * 4 + 1] = u1; m.red[gl * 4 + 2] = v0; m.red[gl * 4 + 3] = v1;
    gsync();
    for (int l = 0; l < G; l++) { u0 = fmin(u0, m.red[l * 4]); u1 = fmax(u1, m.red[l * 4 + 1]); v0 = fmin(v0, m.red[l * 4 + 2]); v1 = fmax(v1, m.red[l * 4 + 3]); }
    gsync();
    if (na < 4 || nb < 4 || x1 < u0 || u1 < x0 || y1 < v0 || v1 < y0) { m.nx = 0; gsync(); return; }
  }
  // ---- crossing detection: a-edges over lanes, all b-edges per lane
  for (int ia = gl; ia + 1 < na; ia += G) {
    double px = pax[ia], py = pay[ia], qx = pax[ia + 1], qy = pay[ia + 1];
    for (int ib = 0; ib + 1 < nb; ib++) {
      double rx = pbx[ib], ry = pby[ib], sx = pbx[ib + 1], sy = pby[ib + 1];
      int sp = side_a_vs_b(rx, ry, sx, sy, px, py), sq = side_a_vs_b(rx, ry, sx, sy, qx, qy);
      if (sp == sq) continue;
      int sr = side_b_vs_a(px, py, qx, qy, rx, ry), ss = side_b_vs_a(px, py, qx, qy, sx, sy);
      if (sr == ss) continue;
      double ex = qx - px, ey = qy - py, fx = sx - rx, fy = sy - ry;
      double denom = ex * fy - ey * fx;
      if (denom == 0.0) continue;
      double wx = rx - px, wy = ry - py;
      double ta = (wx * fy - wy * fx) / denom;
      double tb = (wx * ey - wy * ex) / denom;
      ta = ta < 0.0 ? 0.0 : (ta > 1.0 ? 1.0 : ta);
      tb = tb < 0.0 ? 0.0 : (tb > 1.0 ? 1.0 : tb);
      int slot = atomicAdd(&m.nraw, 1);
      if (slot < KC) {
        m.ria[slot] = (int16_t)ia; m.rib[slot] = (int16_t)ib; m.rta[slot] = ta; m.rtb[slot] = tb;
        m.rcx[slot] = px + ta * ex; m.rcy[slot] = py + ta * ey;
        m.rfl[slot] = (uint8_t)((((sp * ob) < 0) ? 1 : 0) | (((sr * oa) < 0) ? 2 : 0));
      }
    }
  }
  gsync();
  int K = m.nraw;
  if (K > KC) { if (gl == 0) { m.err |= ERR_CAP_XING; m.nx = 0; } gsync(); return; }
  // ---- canonical (ia, ib) order = the serial discovery order
  for (int s = gl; s < K; s += G) {
    int key = (int)m.ria[s] * 65536 + (int)m.rib[s], r = 0;
    for (int t = 0; t < K; t++) r += ((int)m.ria[t] * 65536 + (int)m.rib[t]) < key;
    m.cia[r] = m.ria[s]; m.cib[r] = m.rib[s]; m.cta[r] = m.rta[s]; m.ctb[r] = m.rtb[s];
    m.cx[r] = m.rcx[s]; m.cy[r] = m.rcy[s]; m.cfl[r] = m.rfl[s];
  }
  if (gl == 0) m.nx = K;
  gsync();
  if (K == 0) {
    // boundaries do not cross: containment (lane 0 stores)
    bool a_in_b = ring_inside(pax, pay, na, pbx, pby, nb);
    bool b_in_a = a_in_b ? false : ring_inside(pbx, pby, nb, pax, pay, na);
    if (a_in_b || b_in_a) {
      const double* sx = a_in_b ? pax : pbx; const double* sy = a_in_b ? pay : pby;
      int n = a_in_b ? na : nb;
      if (n > RC) { if (gl == 0) m.err |= ERR_CAP_REGION; }
      else {
        for (int i = gl; i < n; i += G) { m.rx[buf][i] = sx[i]; m.ry[buf][i] = sy[i]; }
        gsync();
        double sa = ring_signed_area(m.rx[buf], m.ry[buf], n);
        if (gl == 0) { m.nreg[buf] = 1; m.roff[buf][1] = (int16_t)n; m.rarea[buf][0] = fabs(sa); }
      }
    }
    gsync();
    return;
  }
  // ---- order along a: key (ia, ta, ib); along b: key (ib, tb, ia)
  for (int k = gl; k < K; k += G) {
    int ra = 0, rb = 0;
    int ia = m.cia[k], ib = m.cib[k]; double ta = m.cta[k], tb = m.ctb[k];
    for (int l = 0; l < K; l++) {
      int ja = m.cia[l], jb = m.cib[l]; double ua = m.cta[l], ub = m.ctb[l];
      ra += (ja < ia) || (ja == ia && (ua < ta || (ua == ta && jb < ib)));
      rb += (jb < ib) || (jb == ib && (ub < tb || (ub == tb && ja < ia)));
    }
    m.rnkA[k] = (int16_t)ra; m.ordA[ra] = (int16_t)k;
    m.rnkB[k] = (int16_t)rb; m.ordB[rb] = (int16_t)k;
  }
  gsync();
  // ---- trace (every lane walks, lane 0 stores)
  const int nea = na - 1, neb = nb - 1;
  uint64_t visited = 0;      // KC <= 64
  int nreg = 0, off = 0;
  const int guard_max = 2 * (na + nb + 2 * K) + 8;
  int nfail = 0;
  for (int s = 0; s < K; s++) {
    bool failed = false;
    int c0 = m.ordA[s];
    if ((visited >> c0) & 1) continue;
    int start = off, cnt = 0, guard = 0;
    auto emit = [&](double x, double y) {
      if (off + cnt < RC) { if (gl == 0) { m.rx[buf][off + cnt] = x; m.ry[buf][off + cnt] = y; } }
      cnt++;
    };
    emit(m.cx[c0], m.cy[c0]);
    visited |= (1ull << c0);
    int cur = c0; bool on_a = true;
    do {
      int fl = m.cfl[cur];
      bool fwd = on_a ? (fl & 1) : ((fl >> 1) & 1);
      int ne = on_a ? nea : neb;
      const double* vx = on_a ? pax : pbx; const double* vy = on_a ? pay : pby;
      int r = on_a ? m.rnkA[cur] : m.rnkB[cur];
      int e0 = on_a ? m.cia[cur] : m.cib[cur];
      int rn, nxt, e1, nv, v;
      if (fwd) {
        rn = r + 1; if (rn == K) rn = 0;
        nxt = on_a ? m.ordA[rn] : m.ordB[rn];
        e1 = on_a ? m.cia[nxt] : m.cib[nxt];
        nv = (rn > r) ? (e1 - e0) : (ne - e0 + e1);
        v = e0 + 1;
        for (int t = 0; t < nv; t++) { if (v >= ne) v -= ne; emit(vx[v], vy[v]); v++; }
      } else {
        rn = r - 1; if (rn < 0) rn = K - 1;
        nxt = on_a ? m.ordA[rn] : m.ordB[rn];
        e1 = on_a ? m.cia[nxt] : m.cib[nxt];
        nv = (rn < r) ? (e0 - e1) : (ne + e0 - e1);
        v = e0;
        for (int t = 0; t < nv; t++) { if (v < 0) v += ne; emit(vx[v], vy[v]); v--; }
      }
      emit(m.cx[nxt], m.cy[nxt]);
      guard += nv + 1;
      if (guard > guard_max) { failed = true; break; }
      cur = nxt;
      visited |= (1ull << cur);
      on_a = !on_a;
    } while (cur != c0);
    // a trace abandoned by the guard (inconsistent crossing flags: self-intersecting input or
    // round-off) yields no region; tracing goes on with the next unvisited crossing
    if (failed) { nfail++; gsync(); continue; }
    if (off + cnt > RC) { if (gl == 0) m.err |= ERR_CAP_REGION; break; }
    gsync();
    double sa = (cnt >= 4) ? ring_signed_area(&m.rx[buf][start], &m.ry[buf][start], cnt) : 0.0;
    if (cnt >= 4 && sa != 0.0) {
      if (nreg < RMAX) {
        if (gl == 0) { m.rarea[buf][nreg] = fabs(sa); m.roff[buf][nreg + 1] = (int16_t)(off + cnt); }
        nreg++; off += cnt;
      } else { if (gl == 0) m.err |= ERR_CAP_REGION; break; }
    }
    gsync();
  }
  if (nfail && gl == 0) m.ntracefail += nfail;
  if (gl == 0) m.nreg[buf] = nreg;
  gsync();
}

// ---------------------------------------------------------------------------------------------
// which_vertices_match_points (floe_utils.jl:331-352) of the unique crossing points against one
// region ring; fills m.midx[0..mcount) sorted ascending and returns mcount.
template <int G, int CAP, int KC, int RC>
SZ_DEV int match_vertices(GroupMem<CAP, KC, RC>& m, int gl, int nuniq, const double* rx, const double* ry, int nr) {
  // unique crossing points are listed (in canonical order) in m.ordB reused as index list? no:
  // m.uniq[k] flags them; the quirk `points[1] == points[end]` drops the last one.
  int K = m.nx;
  int first = -1, last = -1;
  for (int k = 0; k < K; k++) if (m.uniq[k]) { if (first < 0) first = k; last = k; }
  int np = nuniq;
  bool drop_last = (np > 0 && m.cx[first] == m.cx[last] && m.cy[first] == m.cy[last]);
  for (int k = gl; k < K; k += G) {
    int res = -1;
    if (m.uniq[k] && !(drop_last && k == last)) {
      double md = __builtin_inf(); int mv = 0;
      for (int j = 0; j < nr; j++) {
        double dx = rx[j] - m.cx[k], dy = ry[j] - m.cy[k];
        double d = sqrt(sqrt(dx * dx + dy * dy));
        if (d < md) { md = d; mv = j; }
      }
      if (md < 1.0) res = mv;
    }
    m.rnkB[k] = (int16_t)res;     // rnkB is free after the trace
  }
  gsync();
  // gather + sort ascending (tiny; every lane computes, lane 0 stores)
  int cnt = 0;
  for (int k = 0; k < K; k++) if (m.rnkB[k] >= 0) cnt++;
  if (gl == 0) {
    int c = 0;
    for (int k = 0; k < K; k++) {
      int v = m.rnkB[k];
      if (v < 0) continue;
      int u = c - 1;
      while (u >= 0 && m.midx[u] > v) { m.midx[u + 1] = m.midx[u]; u--; }
      m.midx[u + 1] = (int16_t)v; c++;
    }
  }
  gsync();
  return cnt;
}

// _many_intersect_normal_force! (collisions.jl:78-119); returns Δl, updates dir
template <int G, int CAP, int KC, int RC>
SZ_DEV double many_intersect(GroupMem<CAP, KC, RC>& m, int gl, const double* rx, const double* ry, int nr,
                             int na, double force_factor, double& dirx, double& diry) {
  // per-edge classification in parallel: earr = mag (or -1 if the edge is not on p1), farr/garr = Fn
  for (int i = 1 + gl; i < nr; i += G) {
    double x1 = rx[i - 1], y1 = ry[i - 1], x2 = rx[i], y2 = ry[i];
    double xmid = 0.5 * (x2 + x1), ymid = 0.5 * (y2 + y1);
    double dist = dist_to_ring(xmid, ymid, m.ax, m.ay, na);
    double mag = -1.0, fnx = 0.0, fny = 0.0;
    if (dist < 1e-8) {
      double dx = x2 - x1, dy = y2 - y1;
      mag = sqrt(dx * dx + dy * dy);
      double xt = xmid + (-dy / (100 * mag));
      double yt = ymid + (dx / (100 * mag));
      bool in_region = coveredby(xt, yt, rx, ry, nr);
      double f_sign = in_region ? 1.0 : -1.0;
      fnx = (f_sign * force_factor) * (-dy); fny = (f_sign * force_factor) * dx;
    }
    m.earr[i] = mag; m.farr[i] = fnx; m.garr[i] = fny;
  }
  gsync();
  double dl = 0.0, fx = 0.0, fy = 0.0; int n_pts = 0;
  for (int i = 1; i < nr; i++) {
    double mag = m.earr[i];
    if (mag >= 0.0) { dl += mag; n_pts += 1; fx += m.farr[i]; fy += m.garr[i]; }
  }
  gsync();
  if (0 < n_pts && n_pts < nr - 1) {
    dl /= n_pts;
    if (dl > 0.1) {
      double nrm = sqrt(fx * fx + fy * fy);
      dirx = fx / nrm; diry = fy / nrm;
    }
  }
  return dl;
}

// GO.intersects(ring1, ring2) with both rings in LDS
template <int G, int CAP, int KC, int RC>
SZ_DEV bool rings_intersect(GroupMem<CAP, KC, RC>& m, int gl, const double* x1, const double* y1, int n1,
                            const double* x2, const double* y2, int n2) {
  if (n1 < 2 || n2 < 2) return false;
  if (gl == 0) m.flag = 0;
  gsync();
  int ne1 = n1 - 1, ne2 = n2 - 1, tot = ne1 * ne2;
  bool hit = false;
  for (int t = gl; t < tot && !hit; t += G) {
    int i = t / ne2, j = t - i * ne2;
    if (seg_seg_touch(x1[i], y1[i], x1[i + 1], y1[i + 1], x2[j], y2[j], x2[j + 1], y2[j + 1])) hit = true;
  }
  if (hit) m.flag = 1;
  gsync();
  bool res = m.flag != 0;
  gsync();
  if (res) return true;
  if (coveredby(x1[0], y1[0], x2, y2, n2)) return true;
  if (coveredby(x2[0], y2[0], x1, y1, n1)) return true;
  return false;
}

struct Body {          // kinematics of one side of a contact
  double cx, cy, u, v, xi;
  int rigid_uv;        // 1: boundary/topography: velocity is (u, v) everywhere
};

struct ContactParams {
  double E, nu, mu; int dt;
  double force_factor;
  int elem_dir;        // -1 floe-floe / topography; else SZ_NORTH.. for _normal_direction_correct!
  double elem_val;
};

// calc_elastic_forces + calc_friction_forces on the regions of clip buffer 0.
// rows: out[k*5 + {fx, fy, px, py, overlap}], returns number of rows written (zero-force rows
// are dropped exactly like add_interactions!, collisions.jl:288).
template <int G, int CAP, int KC, int RC>
SZ_DEV int contact_rows(GroupMem<CAP, KC, RC>& m, int gl, int na, int oa, int nb, int ob, const Body& bi,
                        const Body& bj, const ContactParams& cp, double* out, int max_rows) {
  int K = m.nx;
  // unique crossing points (GO.intersection_points): first occurrences in canonical order
  for (int k = gl; k < K; k += G) {
    bool dup = false;
    for (int l = 0; l < k; l++) if (m.cx[l] == m.cx[k] && m.cy[l] == m.cy[k]) { dup = true; break; }
    m.uniq[k] = dup ? 0 : 1;
  }
  gsync();
  int nip = 0;
  for (int k = 0; k < K; k++) nip += m.uniq[k];
  int nreg = m.nreg[0];
  // region list after the min-area filter (collisions.jl:158-170): keep[] indexes into buffer 0
  int keep[RMAX]; int nkeep = 0;
  if (nip >= 2) {
    int n1 = na - 1, n2 = nb - 1;
    double min_area = (double)((n1 < n2 ? n1 : n2) * 100) / 1.75;
    for (int r = 0; r < nreg; r++) if (!(m.rarea[0][r] < min_area)) keep[nkeep++] = r;
  }
  int nrows = 0;
  // save crossings needed later? the direction-check clip overwrites the crossing arrays, so the
  // matching for ALL kept regions is done first.
  double dlv[RMAX], dxv[RMAX], dyv[RMAX];
  for (int q = 0; q < nkeep; q++) {
    int r = keep[q];
    const double* rx = &m.rx[0][m.roff[0][r]]; const double* ry = &m.ry[0][m.roff[0][r]];
    int nr = m.roff[0][r + 1] - m.roff[0][r];
    double dirx = 0.0, diry = 0.0, dl = 0.0;
    if (m.rarea[0][r] != 0) {
      int mc = match_vertices<G>(m, gl, nip, rx, ry, nr);
      if (mc == 2) {
        int i1 = m.midx[0], i2 = m.midx[1];
        double dx = rx[i2] - rx[i1], dy = ry[i2] - ry[i1];
        dl = sqrt(dx * dx + dy * dy);
        if (dl > 0.1) { dirx = -dy / dl; diry = dx / dl; }
      } else if (mc != 0) {
        dl = many_intersect<G>(m, gl, rx, ry, nr, na, cp.force_factor, dirx, diry);
      }
      gsync();
    }
    dlv[q] = dl; dxv[q] = dirx; dyv[q] = diry;
  }
  for (int q = 0; q < nkeep; q++) {
    int r = keep[q];
    const double* rx = &m.rx[0][m.roff[0][r]]; const double* ry = &m.ry[0][m.roff[0][r]];
    int nr = m.roff[0][r + 1] - m.roff[0][r];
    double area = m.rarea[0][r];
    double fxn = 0.0, fyn = 0.0, px = 0.0, py = 0.0, dl = dlv[q];
    if (area != 0) {
      ring_centroid(rx, ry, nr, px, py);
      double dirx = dxv[q], diry = dyv[q];
      if (dl > 0.1) {
        // direction check (collisions.jl:58-68): move p1 by the unit direction and re-clip
        for (int i = gl; i < na; i += G) { m.tx[i] = m.ax[i] + dirx; m.ty[i] = m.ay[i] + diry; }
        gsync();
        clip<G>(m, gl, m.tx, m.ty, na, oa, nb, ob, 1);
        int nn = m.nreg[1];
        for (int t = 0; t < nn; t++) {
          const double* nx_ = &m.rx[1][m.roff[1][t]]; const double* ny_ = &m.ry[1][m.roff[1][t]];
          int nnr = m.roff[1][t + 1] - m.roff[1][t];
          bool ints = rings_intersect<G>(m, gl, nx_, ny_, nnr, rx, ry, nr);
          if (ints && m.rarea[1][t] / area > 1) { dirx *= -1; diry *= -1; }
        }
      }
      fxn = dirx * area * cp.force_factor;
      fyn = diry * area * cp.force_factor;
    }
    // _normal_direction_correct! (boundaries.jl:37,73,110,147)
    if (cp.elem_dir == 0 && py >= cp.elem_val) fxn = 0.0;
    if (cp.elem_dir == 1 && py <= cp.elem_val) fxn = 0.0;
    if (cp.elem_dir == 2 && px >= cp.elem_val) fyn = 0.0;
    if (cp.elem_dir == 3 && px <= cp.elem_val) fyn = 0.0;
    // calc_friction_forces (collisions.jl:243-283)
    double G_ = cp.E / (2 * (1 + cp.nu));
    double nnorm = sqrt(fxn * fxn + fyn * fyn);
    double iu = bi.u + bi.xi * (px - bi.cx), iv = bi.v + bi.xi * (py - bi.cy);
    double ju = bj.rigid_uv ? bj.u : bj.u + bj.xi * (px - bj.cx);
    double jv = bj.rigid_uv ? bj.v : bj.v + bj.xi * (py - bj.cy);
    double udiff = iu - ju, vdiff = iv - jv;
    double vnorm = sqrt(udiff * udiff + vdiff * vdiff);
    double xdir = 0.0, ydir = 0.0;
    if (udiff != 0 || vdiff != 0) { xdir = udiff / vnorm; ydir = vdiff / vnorm; }
    double dot_dir = xdir * udiff + ydir * vdiff;
    double xf = G_ * dl * cp.dt * nnorm * xdir * -dot_dir;
    double yf = G_ * dl * cp.dt * nnorm * ydir * -dot_dir;
    double norm_fric = sqrt(xf * xf + yf * yf);
    if (norm_fric > cp.mu * nnorm) { xf = -cp.mu * nnorm * xdir; yf = -cp.mu * nnorm * ydir; }
    double fx = fxn + xf, fy = fyn + yf;
    if (fx != 0 || fy != 0) {
      if (nrows < max_rows) {
        if (gl == 0) { double* o = out + nrows * 5; o[0] = fx; o[1] = fy; o[2] = px; o[3] = py; o[4] = area; }
      } else if (gl == 0) m.err |= ERR_CAP_ROWS;
      nrows++;
    }
    gsync();
  }
  return nrows < max_rows ? nrows : max_rows;
}

}  // namespace szg
