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
#include <type_traits>

namespace szg {

#define SZ_DEV __device__ __forceinline__

// In-kernel phase stamps for the diagnostic build only (-DSZ_STAMPS): cycles per phase summed
// over all groups.  The production build compiles them out.
#ifdef SZ_STAMPS
struct Stamps { long long t0, t0w, tmark, cA, cB, cC, cA1, cA2, cP; long long* log; int n; bool on; int maxrows; int pass; int npass, ntask, nlive; };
#define STAMP_INIT(st) do { (st).t0 = clock64(); (st).t0w = (st).t0; (st).maxrows = 0; (st).pass = 0; (st).npass = 0; (st).ntask = 0; (st).nlive = 0; (st).cA = (st).cB = (st).cC = (st).cA1 = (st).cA2 = 0; (st).cP = -1; (st).tmark = (st).t0; (st).n = 0; (st).on = false; (st).log = nullptr; } while (0)
#define STAMP(st, k) do { if ((st).on && (st).n < 500) { (st).log[(st).n++] = ((long long)(k) << 48) | (clock64() - (st).t0); } } while (0)
#else
struct Stamps {};
#define STAMP_INIT(st) do {} while (0)
#define STAMP(st, k) do {} while (0)
#endif

SZ_DEV void gsync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// LDS working set of one group.  CAP: ring points per polygon, KC: crossings, RC: region points
// per clip, RM: regions per clip.  The set is kept small on purpose: the number of pairs in
// flight per CU is 160 KB / sizeof(GroupMem), and the narrow phase is latency-bound.
template <int CAP, int KC, int RC, int RM>
struct GroupMem {
  static constexpr int RMAXV = RM;
  // (members ordered by alignment -- doubles, 4-byte, 2-byte, bytes -- so that the struct has no padding: eight of
  //  them must stay within 16 KB for ten workgroups per CU)
  using roff_t = typename std::conditional<(RC < 256), uint8_t, int16_t>::type;     // a position in a region buffer
  double ax[CAP], ay[CAP], bx[CAP], by[CAP];
  double cta[KC], ctb[KC], cx[KC], cy[KC];   // crossings in canonical (ia, ib) order
  double reg[2][2][RC];                      // region rings [clip][x|y][point]; reg[1] doubles as the
                                             // raw crossing slots during detection (4*KC <= 2*RC)
  double rarea[2][RM];
  double rcx[RM], rcy[RM];                   // centroids of the regions of clip 0
  // per-item scalars live here rather than in registers across the clips (the kernel's register
  // budget decides how many items the chip holds in flight):
  double kin[14];                            // i: cx cy u v xi area h, j: the same (the order of the collision record's quads {cx, cy} {u, v} {xi, area} {h, -})
  double dlv[RM], dxv[RM], dyv[RM];          // per kept region: contact length and force direction
  double box[8];                             // ring boxes of the item: a x0 x1 y0 y1, b x0 x1 y0 y1 (a direction check of this item may be
  double ff;                                 // run by another lane group of the wavefront, which finds them here); force factor
  uint32_t cinfo[KC];                        // per crossing: ia | ib<<8 | rankA<<16 | rankB<<22 | flags<<28 (edges < 256, ranks < 64, two flag bits)
  int nraw, err, ierr, nea, neb;             // err: bits raised while this group's memory was the scratch; ierr: bits of ITS item (LDS atomics: 32-bit)
  unsigned acc[2];                           // work counters of the group: ring points of its pair items, pair rows
  uint16_t acc16[5];                         // ... pair items, element items, element rows, direction checks, of which certified (a group runs a few dozen items per launch)
  int16_t nx, nreg[2], flag, ntracefail;
  int16_t ria[KC];                           // raw crossing slots: a-edge (doubles as the scratch of match_vertices: the raw slots are dead after the clip)
  roff_t midx[KC];                           // matched region-vertex index per ipoint
  roff_t roff[2][RM + 2];
  roff_t cpos[KC];                           // contact clip: where crossing k sits in its region's ring (position in the region buffer) ...
  uint8_t creg[KC];                          // ... and which region that is (0xff: none) -- kept for the certified direction check
  uint8_t sga[KC], sgb[KC], sgf[KC];         // the contact clip's crossing set: a-edge, b-edge, flags (a copy no later clip in this memory overwrites)
  uint8_t cia[KC], cib[KC], rib[KC];
  uint8_t ordA[KC], ordB[KC];                // crossing id at rank r along a / b
  uint8_t cfl[KC], rfl[KC], uniq[KC];
  int8_t ecode[RC];                          // many-intersect per-edge class
  uint8_t ea[CAP], eb[CAP];                  // edges of a / b that reach into the overlap box
  int8_t keep[RM];
  int8_t chk[RM];                            // the kept regions (positions in keep) that need the direction check
  uint8_t chkc[RM];                          // ... 1: the region is a lens (two crossings): its check may be certified without a second clip
  uint8_t rna, rnb; int8_t roa, rob;         // ring sizes and orientation signs of the item
  int8_t nchk, nkeep;
  uint8_t nsig;                              // crossings of the contact clip (entries of sga / sgb / sgf)
  int8_t eri, erj;                           // fx_lever_exp(rmax) of the item's two floes (fixed-point totals: the scale of the torque / stress sums)
  uint8_t lists_ok, fullnow;                 // ea / eb / nea / neb are still the CONTACT clip's (no later clip in this memory has rebuilt them); this group
                                             // runs a clip that rebuilds them in the current pass -- see clip(.., reuse) and the narrow kernel's pass loop
};
static_assert(sizeof(GroupMem<18, 8, 16, 4>) <= 2048, "eight groups of the first narrow variant must fit 16 KB");

enum { ERR_CAP_RING = 1, ERR_CAP_XING = 2, ERR_CAP_REGION = 4, ERR_CAP_ROWS = 8, ERR_TRACE = 16,
       ERR_CAP_NEIGH = 32, ERR_CAP_PAIRS = 64, ERR_CAP_ELEM = 128, ERR_CAP_INTER = 256,
       ERR_CAP_FLOES = 512, ERR_CAP_VERTS = 1024, ERR_CAP_CELLS = 2048, ERR_GHOSTS_PER_PARENT = 4096,
       ERR_FIXED_RANGE = 32768 };       // (8192: ERR_SCAN, 16384: ERR_HALO_DRIFT -- sz_kernels.hpp)

// ---------------------------------------------------------------------------------------------
// Fixed-point totals (round 4).  collision_force, collision_trq, overarea and the row sums of calc_stress! (collisions.jl:747-749,
// 852-861; update_floe.jl:392-414) are sums over a floe's interaction rows.  The resident steps form them WITHOUT a reduce launch: the
// lane group that finishes an item adds its rows to both floes' totals with integer atomics.  Integer addition is associative, so the
// result does not depend on who adds when -- a tile, the single context and every repetition of a run give the same bits.
//   * a row's force on floe k is bounded by (1 + mu) E h_k sqrt(area_k) (force_factor <= E h_k / sqrt(area_k) for either side, overlap
//     <= min(area); friction <= mu |normal force|): with e_F = kexp + ilogb(h_k) + 1 + ceil((ilogb(area_k) + 1) / 2) every |f| < 2^e_F,
//     and f 2^(52 - e_F) has 10 bits of headroom for the sum of up to 1024 rows in an int64.  The exponents are read off the doubles'
//     exponent fields (area, height, rmax of the floe: the same bits in every instance of it, ghost or halo copy), so every
//     contributor uses the same grid;
//   * lever arms |x - cx_k| <= rmax_k (the contact point is the centroid of a region inside the floe): torque / stress products get
//     e_T = e_F + ilogb(rmax_k) + 1;  overlap areas e_O = ilogb(area_k) + 1;
//   * every word is kept to 40 more bits in a second word (hi = rint(t), lo = rint((t - hi) 2^40), t - hi exact): the sum is then the
//     EXACT sum of the rows' doubles down to 2^-92 of the bound, rounded once when it is read -- closer to the true sum than the serial
//     double sum of the reference (one word alone, 2^-52 of the BOUND, was measurably coarser than the reference's own round-off: the
//     suite's 10-step trajectories left the oracle's by 1e-8), and equal and opposite forces stay exactly so.
// A value beyond its bound by more than the headroom raises ERR_FIXED_RANGE (never seen: the bounds are theorems about the formulas).
constexpr int FX_WORDS = 16;        // int64 words per floe (one 128-byte line): 0 fx, 1 fy, 2 (x-cx)fx, 3 (y-cy)fx, 4 (x-cx)fy, 5 (y-cy)fy, 6 overlap, 7 tag bits,
                                    // 8..14 the low words of 0..6
SZ_DEV int fx_ilogb(double x, int lo, int hi) {
  int e = lo;
  if (x > 0.0 && x < __builtin_inf()) { e = ilogb(x); e = e < lo ? lo : (e > hi ? hi : e); }
  return e;
}
SZ_DEV int fx_force_exp(int kexp, double area, double h) { return kexp + fx_ilogb(h, -40, 40) + 1 + ((fx_ilogb(area, -80, 120) + 2) >> 1); }
SZ_DEV int fx_lever_exp(double rmax) { return fx_ilogb(rmax, -40, 60) + 1; }
SZ_DEV int fx_area_exp(double area) { return fx_ilogb(area, -80, 120) + 1; }
SZ_DEV void fx_split(double v, int e, long long& hi, long long& lo, int& bad) {
  const double t = ldexp(v, 52 - e);
  if (!(fabs(t) < 0x1p62)) { bad = 1; hi = 0; lo = 0; return; }
  const double r = rint(t);
  hi = (long long)r; lo = __double2ll_rn(ldexp(t - r, 40));        // (|t| >= 2^52: t is an integer, the remainder 0)
}
SZ_DEV double fx_join(long long hi, long long lo, int e) {
  const long long H = hi + (lo >> 40), L = lo & ((1ll << 40) - 1);      // (floor and a non-negative remainder: lo may be a negative sum)
  return ldexp((double)H + ldexp((double)L, -40), e - 52);
}
// word w (0..6) of what one interaction row {fx, fy, px, py, overlap} adds to the totals of a floe with centroid (cx, cy); sign: +1 for the pair's
// first floe, -1 for the second (the mirrored row, collisions.jl:820-828).  eF, eA, eL: the floe's force, area and lever exponents (fx_force_exp,
// fx_area_exp, fx_lever_exp -- formed once per item and side, not per row).  The seven words take ONE path: value and exponent are selected, the
// split is evaluated once (the lanes of a group hold a word each; three branches here were three passes for every row and side).
SZ_DEV void fx_word(int w, const double* row, double sign, double cx, double cy, int eF, int eA, int eL, long long& hi, long long& lo, int& bad) {
  const bool lx = w == 2 || w == 4, ffx = w == 0 || w == 2 || w == 3;       // 2: (x - cx) fx   3: (y - cy) fx   4: (x - cx) fy   5: (y - cy) fy
  const double f = (ffx ? row[0] : row[1]) * sign;
  const double lever = (lx ? row[2] : row[3]) - (lx ? cx : cy);
  const double v = w == 6 ? row[4] : w < 2 ? f : lever * f;
  const int e = w == 6 ? eA : w < 2 ? eF : eF + eL;
  long long a = 0, b = 0;
  fx_split(v, e, a, b, bad);
  hi += a; lo += b;
}


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
#pragma unroll 4
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
// is ring a (+ offset) inside ring b, given that their boundaries do not cross
SZ_DEV bool ring_inside(const double* ax, const double* ay, int na, double ox, double oy, const double* bx, const double* by, int nb) {
  for (int i = 0; i + 1 < na; i++) {
    int c = point_in_ring3(ax[i] + ox, ay[i] + oy, bx, by, nb);
    if (c >= 0) return c != 0;
  }
  return true;
}
// 1 inside, 0 outside, -1 on the boundary of ring (rx + ox, ry + oy)
SZ_DEV int point_in_ring3_off(double x, double y, const double* rx, const double* ry, int n, double ox, double oy) {
  int inside = 0;
#pragma unroll 4
  for (int i = 0; i + 1 < n; i++) {
    double ax = rx[i] + ox, ay = ry[i] + oy, bx = rx[i + 1] + ox, by = ry[i + 1] + oy;
    double o = orient(ax, ay, bx, by, x, y);
    if (o == 0.0 && on_segment(ax, ay, bx, by, x, y)) return -1;
    if ((ay > y) != (by > y)) {
      if (by > ay) { if (o > 0.0) inside = !inside; }
      else         { if (o < 0.0) inside = !inside; }
    }
  }
  return inside;
}
SZ_DEV bool ring_inside_off(const double* bx, const double* by, int nb, const double* ax, const double* ay, int na, double ox, double oy) {
  for (int i = 0; i + 1 < nb; i++) {
    int c = point_in_ring3_off(bx[i], by[i], ax, ay, na, ox, oy);
    if (c >= 0) return c != 0;
  }
  return true;
}
// squared GO._euclid_distance(point, segment); sqrt is monotone and correctly rounded, so
// min over edges of sqrt(d2) == sqrt(min over edges of d2) bit for bit
SZ_DEV double dist2_pt_seg(double x0, double y0, double ax, double ay, double bx, double by) {
  double vx = bx - ax, vy = by - ay;
  double wx = x0 - ax, wy = y0 - ay;
  double c1 = wx * vx + wy * vy;
  if (c1 <= 0.0) return (x0 - ax) * (x0 - ax) + (y0 - ay) * (y0 - ay);
  double c2 = vx * vx + vy * vy;
  if (c2 <= c1) return (x0 - bx) * (x0 - bx) + (y0 - by) * (y0 - by);
  double b2 = c1 / c2;
  double px = ax + b2 * vx, py = ay + b2 * vy;
  return (x0 - px) * (x0 - px) + (y0 - py) * (y0 - py);
}
SZ_DEV double dist_to_ring(double x, double y, const double* rx, const double* ry, int n) {
  double md = __builtin_inf();
#pragma unroll 4
  for (int i = 0; i + 1 < n; i++) {
    double d = dist2_pt_seg(x, y, rx[i], ry[i], rx[i + 1], ry[i + 1]);
    if (d < md) md = d;
  }
  return sqrt(md);
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
// one pass over a closed ring: GO._signed_area (sa) and the sums of GO.centroid (cx, cy); the
// area terms of both formulas are the same products, so one accumulator serves both
SZ_DEV void ring_area_centroid(const double* x, const double* y, int n, double& sa, double& cx, double& cy) {
  double xc = 0.0, yc = 0.0, a2 = 0.0, p1x = x[0], p1y = y[0];
#pragma unroll 4
  for (int i = 1; i < n; i++) {
    double p2x = x[i], p2y = y[i];
    double ac = p1x * p2y - p2x * p1y;
    a2 += ac;
    xc += (p1x + p2x) * ac;
    yc += (p1y + p2y) * ac;
    p1x = p2x; p1y = p2y;
  }
  double ah = a2 / 2.0;
  cx = xc / (6.0 * ah); cy = yc / (6.0 * ah);
  sa = (a2 + (p1x * y[0] - p1y * x[0])) / 2.0;
}
// GO._signed_area on a closed ring
// (st: stride of the coordinate arrays in doubles -- 2 for the interleaved floe rings)
SZ_DEV double ring_signed_area(const double* x, const double* y, int n, int st = 1) {
  if (n == 0) return 0.0;
  double area = 0.0, p1x = x[0], p1y = y[0];
  for (int i = 1; i < n; i++) {
    double p2x = x[(size_t)i * st], p2y = y[(size_t)i * st];
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

struct Box { double x0, x1, y0, y1; };   // closed bounding box of a ring

template <int G>
SZ_DEV double gmin(double v) { for (int d = G / 2; d >= 1; d >>= 1) v = fmin(v, __shfl_xor(v, d, G)); return v; }
template <int G>
SZ_DEV double gmax(double v) { for (int d = G / 2; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, G)); return v; }

// ---------------------------------------------------------------------------------------------
// clip(): regions of (a + (ox, oy)) ∩ b into buffer `buf`.  a = (m.ax, m.ay, na) and
// b = (m.bx, m.by, nb) are closed rings in LDS; oa/ob their orientation signs.  The translation
// is applied on the fly (x + 0.0 == x, so the untranslated clip is unchanged).
// Group-uniform: every lane of the group calls it with the same arguments.
// `ring` holds the two rings, `m` is the working set the clip writes (crossings, regions of buffer `buf`): the same
// GroupMem for an item's own clips; the direction check of an item may be run by ANOTHER lane group of the wavefront,
// which then reads the rings from the owner's memory and works in its own (narrow phase, sz_kernels.hpp).
// detect_only: stop after the crossing detection (m.nraw raw crossings in m.ria / m.rib / m.rfl, unordered) -- what the certified
// direction check needs of the translated polygon; nothing but the raw slots and the candidate-edge lists of `m` is written
template <int G, class MEM>
// reuse (detect-only clips of a direction check): the candidate edges are the contact clip's, still standing in the OWNER's memory (`ring`) --
// the contact clip selects them with the overlap box grown by 2 m, which covers every translation of ring a by a unit vector (the box
// moves by at most 1 m, the edges of a by at most 1 m), and a superset of candidates changes no result: the crossing test is exact
SZ_DEV void clip(const MEM& ring, MEM& m, int gl, double ox, double oy, int na, int oa, int nb, int ob, int buf, const Box& ba, const Box& bb,
                 Stamps& st, bool detect_only = false, bool reuse = false) {
  using roff_t = typename MEM::roff_t;
  constexpr int KC = sizeof(m.cta) / sizeof(double);
  constexpr int RC = sizeof(m.ecode);
  constexpr int RM = MEM::RMAXV;
  static_assert(KC <= 64 && sizeof(m.ea) <= 255, "crossing ranks travel in 6 bits, edge indices and ring sizes in 8");
  const double* pax = ring.ax; const double* pay = ring.ay; const double* pbx = ring.bx; const double* pby = ring.by;
  double* rgx = m.reg[buf][0]; double* rgy = m.reg[buf][1];
  if (gl == 0) { m.nraw = 0; if (!reuse) { m.nea = 0; m.neb = 0; } if (!detect_only) { m.nreg[buf] = 0; m.roff[buf][0] = 0; } }
  if (na < 4 || nb < 4) { if (gl == 0 && !detect_only) m.nx = 0; gsync(); return; }
  // bounding boxes are kept per floe (min/max commute with the rounding of `+ ox`, so the box of
  // the translated ring is the translated box, bit for bit)
  const double ax0 = ba.x0 + ox, ax1 = ba.x1 + ox, ay0 = ba.y0 + oy, ay1 = ba.y1 + oy;
  if (ax1 < bb.x0 || bb.x1 < ax0 || ay1 < bb.y0 || bb.y1 < ay0) { if (gl == 0 && !detect_only) m.nx = 0; gsync(); return; }
  // overlap box: a crossing point lies in both rings' boxes, so only edges that reach into it can cross
  const double grow = buf == 0 ? 2.0 : 0.0;          // (the contact clip's candidates also serve the direction checks: see `reuse`)
  const double qx0 = fmax(ax0, bb.x0) - grow, qx1 = fmin(ax1, bb.x1) + grow, qy0 = fmax(ay0, bb.y0) - grow, qy1 = fmin(ay1, bb.y1) + grow;
  gsync();
  for (int ia = gl; ia + 1 < na && !reuse; ia += G) {
    double px = pax[ia] + ox, py = pay[ia] + oy, rx = pax[ia + 1] + ox, ry = pay[ia + 1] + oy;
    if (fmax(px, rx) >= qx0 && fmin(px, rx) <= qx1 && fmax(py, ry) >= qy0 && fmin(py, ry) <= qy1) m.ea[atomicAdd(&m.nea, 1)] = (uint8_t)ia;
  }
  for (int ib = gl; ib + 1 < nb && !reuse; ib += G) {
    double px = pbx[ib], py = pby[ib], rx = pbx[ib + 1], ry = pby[ib + 1];
    if (fmax(px, rx) >= qx0 && fmin(px, rx) <= qx1 && fmax(py, ry) >= qy0 && fmin(py, ry) <= qy1) m.eb[atomicAdd(&m.neb, 1)] = (uint8_t)ib;
  }
  gsync();
  STAMP(st, 1);
  // ---- crossing detection, phase 1: which candidate (a-edge, b-edge) pairs cross -- orientation
  // signs only, candidate pairs over lanes.  The divisions of the crossing parameters are done
  // afterwards, once per crossing (phase 2).
  double* raw = m.reg[1][0];
  {
    const uint8_t* lea = reuse ? ring.ea : m.ea; const uint8_t* leb = reuse ? ring.eb : m.eb;
    const int ca = reuse ? ring.nea : m.nea, cb = reuse ? ring.neb : m.neb, tot = ca * cb;
    for (int t = gl; t < tot; t += G) {
      int ua = t / cb, ub = t - ua * cb;
      int ia = lea[ua], ib = leb[ub];
      double px = pax[ia] + ox, py = pay[ia] + oy, qx = pax[ia + 1] + ox, qy = pay[ia + 1] + oy;
      double rx = pbx[ib], ry = pby[ib], sx = pbx[ib + 1], sy = pby[ib + 1];
      int sp = side_a_vs_b(rx, ry, sx, sy, px, py), sq = side_a_vs_b(rx, ry, sx, sy, qx, qy);
      if (sp != sq) {
        int sr = side_b_vs_a(px, py, qx, qy, rx, ry), ss = side_b_vs_a(px, py, qx, qy, sx, sy);
        if (sr != ss) {
          int slot = atomicAdd(&m.nraw, 1);
          if (slot < KC) {
            m.ria[slot] = (int16_t)ia; m.rib[slot] = (uint8_t)ib;
            m.rfl[slot] = (uint8_t)((((sp * ob) < 0) ? 1 : 0) | (((sr * oa) < 0) ? 2 : 0));
          }
        }
      }
    }
  }
  gsync();
  if (detect_only) return;
  int K = m.nraw;
  if (K > KC) { if (gl == 0) { m.err |= ERR_CAP_XING; m.nx = 0; } gsync(); return; }
  // ---- phase 2: parameters and point of every crossing (one lane per crossing)
  for (int s2 = gl; s2 < K; s2 += G) {
    int ia = m.ria[s2], ib = m.rib[s2];
    double px = pax[ia] + ox, py = pay[ia] + oy, qx = pax[ia + 1] + ox, qy = pay[ia + 1] + oy;
    double rx = pbx[ib], ry = pby[ib], sx = pbx[ib + 1], sy = pby[ib + 1];
    double ex = qx - px, ey = qy - py, fx = sx - rx, fy = sy - ry;
    double denom = ex * fy - ey * fx;
    double wx = rx - px, wy = ry - py;
    double ta = (wx * fy - wy * fx) / denom;
    double tb = (wx * ey - wy * ex) / denom;
    ta = ta < 0.0 ? 0.0 : (ta > 1.0 ? 1.0 : ta);
    tb = tb < 0.0 ? 0.0 : (tb > 1.0 ? 1.0 : tb);
    raw[s2] = ta; raw[KC + s2] = tb; raw[2 * KC + s2] = px + ta * ex; raw[3 * KC + s2] = py + ta * ey;
    if (denom == 0.0) m.rfl[s2] |= 128;     // cannot happen for straddling edges; kept as a marker
  }
  gsync();
  STAMP(st, 3);
  // ---- canonical (ia, ib) order = the serial discovery order
  for (int s = gl; s < K; s += G) {
    int key = (int)m.ria[s] * 65536 + (int)m.rib[s], r = 0;
    for (int t = 0; t < K; t++) r += ((int)m.ria[t] * 65536 + (int)m.rib[t]) < key;
    m.cia[r] = (uint8_t)m.ria[s]; m.cib[r] = m.rib[s]; m.cta[r] = raw[s]; m.ctb[r] = raw[KC + s];
    m.cx[r] = raw[2 * KC + s]; m.cy[r] = raw[3 * KC + s]; m.cfl[r] = m.rfl[s];
    if (buf == 0) { m.sga[r] = (uint8_t)m.ria[s]; m.sgb[r] = m.rib[s]; m.sgf[r] = m.rfl[s] & 3; m.creg[r] = 0xff; }      // (the contact clip's crossing set, kept)
  }
  if (gl == 0) { m.nx = K; if (buf == 0) m.nsig = (uint8_t)K; }
  gsync();
  STAMP(st, 4);
  if (K == 0) {
    // boundaries do not cross: containment
    // a ring inside another has its box inside the other's box (closed): skip the walk otherwise
    bool a_in_b = (bb.x0 <= ax0 && ax1 <= bb.x1 && bb.y0 <= ay0 && ay1 <= bb.y1) && ring_inside(pax, pay, na, ox, oy, pbx, pby, nb);
    bool b_in_a = a_in_b ? false
                         : (ax0 <= bb.x0 && bb.x1 <= ax1 && ay0 <= bb.y0 && bb.y1 <= ay1) && ring_inside_off(pbx, pby, nb, pax, pay, na, ox, oy);
    if (a_in_b || b_in_a) {
      int n = a_in_b ? na : nb;
      double sa, ccx, ccy;
      if (buf == 0) {
        // contact clip (ox = oy = 0): the region IS the contained ring.  Without crossing points it
        // can only contribute its area (fuse / remove tests), never a force row (collisions.jl:156-170),
        // so it is measured where it lies instead of being copied into the region buffer.
        ring_area_centroid(a_in_b ? pax : pbx, a_in_b ? pay : pby, n, sa, ccx, ccy);
        if (gl == 0) { m.nreg[0] = 1; m.roff[0][1] = 0; m.rarea[0][0] = fabs(sa); m.rcx[0] = ccx; m.rcy[0] = ccy; }
      } else if (n > RC) { if (gl == 0) m.err |= ERR_CAP_REGION; }
      else {
        for (int i = gl; i < n; i += G) {
          rgx[i] = a_in_b ? pax[i] + ox : pbx[i]; rgy[i] = a_in_b ? pay[i] + oy : pby[i];
        }
        gsync();
        ring_area_centroid(rgx, rgy, n, sa, ccx, ccy);
        if (gl == 0) { m.nreg[buf] = 1; m.roff[buf][1] = (roff_t)n; m.rarea[buf][0] = fabs(sa); }
      }
    }
    gsync();
    STAMP(st, 5);
    return;
  }
  // ---- order along a: key (ia, ta, ib); along b: key (ib, tb, ia); everything the walk needs
  // about a crossing is packed into one word so that a step costs two dependent LDS reads
  for (int k = gl; k < K; k += G) {
    int ra = 0, rb = 0;
    int ia = m.cia[k], ib = m.cib[k]; double ta = m.cta[k], tb = m.ctb[k];
#pragma unroll 2
    for (int l = 0; l < K; l++) {
      int ja = m.cia[l], jb = m.cib[l]; double ua = m.cta[l], ub = m.ctb[l];
      ra += (ja < ia) || (ja == ia && (ua < ta || (ua == ta && jb < ib)));
      rb += (jb < ib) || (jb == ib && (ub < tb || (ub == tb && ja < ia)));
    }
    m.ordA[ra] = (uint8_t)k; m.ordB[rb] = (uint8_t)k;
    m.cinfo[k] = (uint32_t)ia | ((uint32_t)ib << 8) | ((uint32_t)ra << 16) | ((uint32_t)rb << 22) | ((uint32_t)(m.cfl[k] & 3) << 28);
  }
  gsync();
  STAMP(st, 6);
  // ---- trace (every lane walks, lane 0 stores)
  const int nea = na - 1, neb = nb - 1;
  uint64_t visited = 0;      // KC <= 64
  int nreg = 0, off = 0, nfail = 0;
  const int guard_max = 2 * (na + nb + 2 * K) + 8;
  for (int s = 0; s < K; s++) {
    bool failed = false;
    int c0 = m.ordA[s];
    if ((visited >> c0) & 1) continue;
    int start = off, cnt = 0, guard = 0;
    // points are appended at off + cnt; every lane tracks cnt, the stores are spread over the lanes
    auto emit1 = [&](double x, double y) {          // one point (a crossing): lane 0 stores
      if (off + cnt < RC) { if (gl == 0) { rgx[off + cnt] = x; rgy[off + cnt] = y; } }
      cnt++;
    };
    const uint64_t visited0 = visited;
    if (buf == 0 && gl == 0) { m.cpos[c0] = (roff_t)(off + cnt); m.creg[c0] = (uint8_t)nreg; }
    emit1(m.cx[c0], m.cy[c0]);
    visited |= (1ull << c0);
    int cur = c0; bool on_a = true;
    uint32_t ci = m.cinfo[cur];
    do {
      int fl = (int)(ci >> 28);
      bool fwd = on_a ? (fl & 1) : ((fl >> 1) & 1);
      int ne = on_a ? nea : neb;
      const double* vx = on_a ? pax : pbx; const double* vy = on_a ? pay : pby;
      double sx = on_a ? ox : 0.0, sy = on_a ? oy : 0.0;
      int r = on_a ? (int)((ci >> 16) & 63) : (int)((ci >> 22) & 63);
      int e0 = on_a ? (int)(ci & 255) : (int)((ci >> 8) & 255);
      int rn, nxt, e1, nv;
      uint32_t cn;
      if (fwd) { rn = r + 1; if (rn == K) rn = 0; } else { rn = r - 1; if (rn < 0) rn = K - 1; }
      nxt = on_a ? m.ordA[rn] : m.ordB[rn];
      cn = m.cinfo[nxt];
      e1 = on_a ? (int)(cn & 255) : (int)((cn >> 8) & 255);
      if (fwd) nv = (rn > r) ? (e1 - e0) : (ne - e0 + e1);
      else     nv = (rn < r) ? (e0 - e1) : (ne + e0 - e1);
      // the run of ring vertices between the two crossings: vertex t of the run is
      // e0 + 1 + t (forward) or e0 - t (backward), modulo the ring; lanes copy it in parallel
      for (int t = gl; t < nv; t += G) {
        int v = fwd ? e0 + 1 + t : e0 - t;
        if (v >= ne) v -= ne;
        if (v < 0) v += ne;
        if (off + cnt + t < RC) { rgx[off + cnt + t] = vx[v] + sx; rgy[off + cnt + t] = vy[v] + sy; }
      }
      cnt += nv;
      if (buf == 0 && gl == 0 && nxt != c0) { m.cpos[nxt] = (roff_t)(off + cnt); m.creg[nxt] = (uint8_t)nreg; }
      emit1(m.cx[nxt], m.cy[nxt]);
      guard += nv + 1;
      if (guard > guard_max) { failed = true; break; }
      cur = nxt; ci = cn;
      visited |= (1ull << cur);
      on_a = !on_a;
    } while (cur != c0);
    // a trace abandoned by the guard (inconsistent crossing flags: self-intersecting input or
    // round-off) yields no region; tracing goes on with the next unvisited crossing
    // (a trace that yields no region: its crossings belong to none)
    auto disown = [&]() { if (buf == 0 && gl == 0) for (uint64_t q = visited ^ visited0; q; q &= q - 1) m.creg[__ffsll((long long)q) - 1] = 0xff; };
    if (failed) { nfail++; disown(); gsync(); continue; }
    if (off + cnt > RC) { if (gl == 0) m.err |= ERR_CAP_REGION; break; }
    gsync();
    double sa = 0.0, ccx = 0.0, ccy = 0.0;
    if (cnt >= 4) ring_area_centroid(&rgx[start], &rgy[start], cnt, sa, ccx, ccy);
    if (cnt >= 4 && sa != 0.0) {
      if (nreg < RM) {
        if (gl == 0) {
          m.rarea[buf][nreg] = fabs(sa); m.roff[buf][nreg + 1] = (roff_t)(off + cnt);
          if (buf == 0) { m.rcx[nreg] = ccx; m.rcy[nreg] = ccy; }
        }
        nreg++; off += cnt;
      } else { if (gl == 0) m.err |= ERR_CAP_REGION; break; }
    } else disown();
    gsync();
  }
  if (nfail && gl == 0) m.ntracefail += nfail;
  if (gl == 0) m.nreg[buf] = nreg;
  gsync();
  STAMP(st, 7);
}

template <int G, class MEM>
SZ_DEV void clip(MEM& m, int gl, double ox, double oy, int na, int oa, int nb, int ob, int buf, const Box& ba, const Box& bb, Stamps& st) {
  clip<G>(m, m, gl, ox, oy, na, oa, nb, ob, buf, ba, bb, st, false);
}

// ---------------------------------------------------------------------------------------------
// which_vertices_match_points (floe_utils.jl:331-352) of the unique crossing points against one
// region ring; fills m.midx[0..mcount) sorted ascending and returns mcount.  The nearest vertex
// is found on squared distances (sqrt is monotone; a tie is only possible between coincident
// vertices, where the first wins either way) and sqrt(sqrt(.)) is applied once, to the minimum.
template <int G, class MEM>
SZ_DEV int match_vertices(MEM& m, int gl, int nuniq, const double* rx, const double* ry, int nr) {
  int K = m.nx;
  int first = -1, last = -1;
  for (int k = 0; k < K; k++) if (m.uniq[k]) { if (first < 0) first = k; last = k; }
  bool drop_last = (nuniq > 0 && m.cx[first] == m.cx[last] && m.cy[first] == m.cy[last]);
  for (int k = gl; k < K; k += G) {
    int res = -1;
    if (m.uniq[k] && !(drop_last && k == last)) {
      double md = __builtin_inf(); int mv = 0;
      double px = m.cx[k], py = m.cy[k];
#pragma unroll 4
      for (int j = 0; j < nr; j++) {
        double dx = rx[j] - px, dy = ry[j] - py;
        double d = dx * dx + dy * dy;
        if (d < md) { md = d; mv = j; }
      }
      if (sqrt(sqrt(md)) < 1.0) res = mv;
    }
    m.ria[k] = (int16_t)res;      // the raw crossing slots are free after the clip
  }
  gsync();
  int cnt = 0;
  for (int k = 0; k < K; k++) if (m.ria[k] >= 0) cnt++;
  if (gl == 0) {
    int c = 0;
    for (int k = 0; k < K; k++) {
      int v = m.ria[k];
      if (v < 0) continue;
      int u = c - 1;
      while (u >= 0 && m.midx[u] > v) { m.midx[u + 1] = m.midx[u]; u--; }
      m.midx[u + 1] = (typename MEM::roff_t)v; c++;
    }
  }
  gsync();
  return cnt;
}

// _many_intersect_normal_force! (collisions.jl:78-119); returns Δl, updates dir.  The per-edge
// classification (is the edge midpoint on p1?  which side is inside the region?) runs over the
// lanes; the sums run in ring order with the edge terms recomputed from the same expressions.
template <int G, class MEM>
SZ_DEV double many_intersect(MEM& m, int gl, const double* rx, const double* ry, int nr, int na, double force_factor,
                             double& dirx, double& diry) {
  for (int i = 1 + gl; i < nr; i += G) {
    double x1 = rx[i - 1], y1 = ry[i - 1], x2 = rx[i], y2 = ry[i];
    double xmid = 0.5 * (x2 + x1), ymid = 0.5 * (y2 + y1);
    double dist = dist_to_ring(xmid, ymid, m.ax, m.ay, na);
    int code = 0;
    if (dist < 1e-8) {
      double dx = x2 - x1, dy = y2 - y1;
      double mag = sqrt(dx * dx + dy * dy);
      double xt = xmid + (-dy / (100 * mag));
      double yt = ymid + (dx / (100 * mag));
      code = coveredby(xt, yt, rx, ry, nr) ? 1 : -1;
    }
    m.ecode[i] = (int8_t)code;
  }
  gsync();
  double dl = 0.0, fx = 0.0, fy = 0.0; int n_pts = 0;
  for (int i = 1; i < nr; i++) {
    int code = m.ecode[i];
    if (code != 0) {
      double dx = rx[i] - rx[i - 1], dy = ry[i] - ry[i - 1];
      double mag = sqrt(dx * dx + dy * dy);
      double f_sign = code > 0 ? 1.0 : -1.0;
      dl += mag; n_pts += 1; fx += (f_sign * force_factor) * (-dy); fy += (f_sign * force_factor) * dx;
    }
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
template <int G, class MEM>
SZ_DEV bool rings_intersect(MEM& m, int gl, const double* x1, const double* y1, int n1, const double* x2,
                            const double* y2, int n2) {
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

enum { ITEM_PAIR = 0, ITEM_OPEN = 1, ITEM_SOLID = 2 };
enum { IT_FUSE = 1, IT_REMOVE = 2, IT_RETRY = 4 };

enum { KIN_I = 0, KIN_J = 7, KIN_AREA_I = 5, KIN_H_I = 6, KIN_AREA_J = 12, KIN_H_J = 13 };   // GroupMem::kin

struct ItemCtx {
  int mode;            // ITEM_PAIR: floe-floe; ITEM_OPEN: open boundary; ITEM_SOLID: collision/moving boundary, topography
  double E, nu, mu; int dt; int dbg;
  double max_overlap;  // floe_floe_max_overlap (pairs) or floe_domain_max_overlap (elements)
  int elem_dir;        // -1 floe-floe / topography; else SZ_NORTH.. for _normal_direction_correct!
  double elem_val;
  int rigid_j;         // 1: boundary/topography: the velocity of side j is (u, v) everywhere
};

// One work item = floe_floe_interaction! (collisions.jl:347-408) or floe_domain_element_interaction! (:427-557) on the
// rings staged in m.ax/ay, m.bx/by, in three phases so that the expensive part -- the reference's direction check, a
// second clip of the translated polygon per contact region (collisions.jl:58-68) -- can be handed to whichever lane
// group of the wavefront is free (the groups of a wavefront share LDS):
//   contact clip + contact_post  overlap tests (fuse / remove), force factor, min-area filter, and per kept region the normal
//                  direction up to its sign (m.keep / dlv / dxv / dyv; m.chk = the regions that need the check)
//   check clip + check_post      ONE direction check: region q of the item in `own`, working set `scr` (any group's)
//   finish_phase   friction and the rows, in region order (zero-force rows dropped like add_interactions!, :288)
// rows: out[k*5 + {fx, fy, px, py, overlap}].  clip() -- by far the largest routine -- is called from ONE place in the
// kernel (the passes of one loop: pass 0 = the contact clips, passes 1.. = the check clips), so that the kernel fits the
// instruction cache: with two call sites it grew from 34 to 47 KB and lost 25 % at 100 k floes, where ten wavefronts
// per CU are in different phases at any time.
constexpr int CAPBITS = ERR_CAP_XING | ERR_CAP_REGION | ERR_CAP_ROWS;

// before the contact clip: what another lane group needs to run a direction check of this item
template <class MEM>
SZ_DEV void contact_pre(MEM& m, int gl, int na, int oa, int nb, int ob, const Box& ba, const Box& bb) {
  if (gl == 0) {
    m.box[0] = ba.x0; m.box[1] = ba.x1; m.box[2] = ba.y0; m.box[3] = ba.y1; m.box[4] = bb.x0; m.box[5] = bb.x1; m.box[6] = bb.y0; m.box[7] = bb.y1;
    m.rna = (uint8_t)na; m.rnb = (uint8_t)nb; m.roa = (int8_t)oa; m.rob = (int8_t)ob; m.nkeep = 0; m.nchk = 0;
  }
}
// after the contact clip (clip<G>(m, m, gl, 0, 0, .., buffer 0, ..))
template <int G, class MEM>
SZ_DEV void contact_post(MEM& m, int gl, int na, int nb, const ItemCtx& cx_, int& flags, Stamps& st) {
  int nkeep = 0, nchk = 0;
  double force_factor = 0.0;
  flags = 0;
  do {
    // ---------------- after the contact clip: overlap tests, force factor, per-region direction
    const int nreg = m.nreg[0];
    double total = 0.0, amax = 0.0;
    for (int t = 0; t < nreg; t++) { double a = m.rarea[0][t]; total += a; if (a > amax) amax = a; }
    const double area_i = m.kin[KIN_AREA_I], h_i = m.kin[KIN_H_I];
    if (cx_.mode == ITEM_PAIR) {
      if (!(total > 0)) break;
      const double area_j = m.kin[KIN_AREA_J], h_j = m.kin[KIN_H_J];
      double r1 = total / area_i, r2 = total / area_j;
      if ((r1 > r2 ? r1 : r2) > cx_.max_overlap) { flags |= IT_FUSE; break; }
      double ih = h_i, ir = sqrt(area_i), jh = h_j, jr = sqrt(area_j);
      if (ir > 1e5 || jr > 1e5) force_factor = cx_.E * (ih < jh ? ih : jh) / (ir < jr ? ir : jr);
      else force_factor = cx_.E * (ih * jh) / (ih * jr + jh * ir);
    } else if (cx_.mode == ITEM_OPEN) {
      if (total > 0) flags |= IT_REMOVE;
      break;
    } else {
      if (!(amax > 0)) break;
      if (amax / area_i > cx_.max_overlap) { flags |= IT_REMOVE; break; }
      force_factor = cx_.E * h_i / sqrt(area_i);
    }
    if (cx_.dbg & 1) break;
    STAMP(st, 12);
    // unique crossing points (GO.intersection_points): first occurrences in canonical order
    const int K = m.nx;
    int nip = 0;
    // The common contact -- two crossings, one lens-shaped region -- needs none of the general matching: both crossing points are vertices
    // of the region (at distance 0, so which_vertices_match_points, floe_utils.jl:331-352, matches each to the FIRST region vertex with
    // exactly its coordinates), two matches is the m == 2 branch of calc_normal_force.  Same indices, same expressions, same bits.
    const bool two = K == 2 && nreg == 1 && m.creg[0] == 0 && m.creg[1] == 0;
    if (two) nip = (m.cx[0] == m.cx[1] && m.cy[0] == m.cy[1]) ? 1 : 2;
    else {
      for (int k = gl; k < K; k += G) {
        bool dup = false;
        for (int l = 0; l < k; l++) if (m.cx[l] == m.cx[k] && m.cy[l] == m.cy[k]) { dup = true; break; }
        m.uniq[k] = dup ? 0 : 1;
      }
      gsync();
      for (int k = 0; k < K; k++) nip += m.uniq[k];
    }
    // min-area filter (collisions.jl:158-170)
    if (nip >= 2) {
      int n1 = na - 1, n2 = nb - 1;
      double min_area = (double)((n1 < n2 ? n1 : n2) * 100) / 1.75;
      for (int t = 0; t < nreg; t++) if (!(m.rarea[0][t] < min_area)) { if (gl == 0) m.keep[nkeep] = (int8_t)t; nkeep++; }
    }
    gsync();
    // calc_normal_force up to the direction check, for every kept region (the check clips overwrite the crossing
    // arrays of whichever group runs them, so all matching happens here)
    for (int w = 0; w < nkeep; w++) {
      int rr = m.keep[w];
      const double* rx = &m.reg[0][0][m.roff[0][rr]]; const double* ry = &m.reg[0][1][m.roff[0][rr]];
      int nr = m.roff[0][rr + 1] - m.roff[0][rr];
      double ddx = 0.0, ddy = 0.0, ddl = 0.0;
      if (m.rarea[0][rr] != 0) {
        if (two) {
          // first region vertex with the coordinates of crossing 0 / 1 (the crossing's own position unless an earlier vertex coincides with it)
          int f0 = nr, f1 = nr;
          const double c0x = m.cx[0], c0y = m.cy[0], c1x = m.cx[1], c1y = m.cy[1];
          for (int j = gl; j < nr; j += G) {
            if (rx[j] == c0x && ry[j] == c0y && j < f0) f0 = j;
            if (rx[j] == c1x && ry[j] == c1y && j < f1) f1 = j;
          }
          for (int d = G / 2; d >= 1; d >>= 1) { const int o0 = __shfl_xor(f0, d, G), o1 = __shfl_xor(f1, d, G); f0 = o0 < f0 ? o0 : f0; f1 = o1 < f1 ? o1 : f1; }
          const int i1 = f0 < f1 ? f0 : f1, i2 = f0 < f1 ? f1 : f0;
          double ex = rx[i2] - rx[i1], ey = ry[i2] - ry[i1];
          ddl = sqrt(ex * ex + ey * ey);
          if (ddl > 0.1) { ddx = -ey / ddl; ddy = ex / ddl; }
        } else {
          int mc = match_vertices<G>(m, gl, nip, rx, ry, nr);
          if (mc == 2) {
            int i1 = m.midx[0], i2 = m.midx[1];
            double ex = rx[i2] - rx[i1], ey = ry[i2] - ry[i1];
            ddl = sqrt(ex * ex + ey * ey);
            if (ddl > 0.1) { ddx = -ey / ddl; ddy = ex / ddl; }
          } else if (mc != 0) {
            ddl = many_intersect<G>(m, gl, rx, ry, nr, na, force_factor, ddx, ddy);
          }
          gsync();
        }
      }
      // the direction check runs for a region with area and a contact length (collisions.jl:58)
      const bool check = m.rarea[0][rr] != 0 && ddl > 0.1 && !(cx_.dbg & 2);
      // a region with exactly two crossings on its boundary is a lens: its check can be certified from the crossing detection of the
      // translated polygon alone (certified_check below)
      int ncr = 0;
      for (int k = 0; k < K; k++) ncr += m.creg[k] == rr;
      if (gl == 0) { m.dlv[w] = ddl; m.dxv[w] = ddx; m.dyv[w] = ddy; if (check) { m.chk[nchk] = (int8_t)w; m.chkc[nchk] = (ncr == 2 && !(cx_.dbg & 64)) ? 1 : 0; } }
      if (check) nchk++;
    }
    STAMP(st, 8);
  } while (false);
  if (gl == 0) { m.nkeep = (int8_t)nkeep; m.nchk = (int8_t)nchk; m.ff = force_factor; m.ierr = m.err & CAPBITS; m.err &= ~CAPBITS; }
  gsync();
}

// ---------------------------------------------------------------------------------------------
// The direction check of calc_normal_force (collisions.jl:58-68) WITHOUT the second clip, for a contact region that is a lens
// (two crossings c_in, c_out: p1's ring enters p2 at c_in and leaves it at c_out).  The reference translates p1 by d (the unit
// normal), clips again and flips the sign for every new region that intersects the old one and is larger.  If the translated p1
// crosses p2 on exactly the same edge pairs with the same entry / exit flags as before -- checked here against the crossing
// detection of the translated polygon, run by clip(.., detect_only) into `scr` -- the new clip has the same regions with the same
// vertex sequences, only the crossing points have slid along their (straight) edges, LINEARLY in the translation.  The area of
// the lens then changes by EXACTLY
//     dA = oa * [ d x (c_out - c_in) + 1/2 d x (w_out - w_in) ],      w = slide of a crossing = e_b * (d x e_a) / (e_b x e_a)
// (the flux of d through the part of the lens boundary that belongs to p1, integrated over the translation; oa = orientation sign
// of p1; e_a, e_b = the crossing's edge vectors), and "new region larger" is dA > 0.  The decision is taken only when |dA| is far
// above the round-off of the reference's own area sums (shoelace over absolute coordinates: ~ n eps Lc^2), when the new lens
// certainly still intersects the old one (they share a vertex of p2, or their intervals on the one p2 edge overlap), and when no
// OTHER region of the item can come within reach of this one (bounding boxes more than the translation + slides apart).  Otherwise:
// false, and the caller runs the full check.  Returns true when the sign has been settled (and flipped if need be).
template <int G, class MEM>
SZ_DEV bool certified_check(MEM& own, const MEM& scr, int gl, int q) {
  const int K = own.nsig, K1 = scr.nraw;
  if (K1 != K || K < 2) return false;
  // every crossing of the translated polygon is one of the contact clip's (same edges, same flags); K1 == K and distinct pairs: a bijection
  {
    bool mine = true;
    for (int s2 = gl; s2 < K1; s2 += G) {
      const int ia = scr.ria[s2], ib = scr.rib[s2], fl = scr.rfl[s2] & 3;
      bool found = false;
      for (int k = 0; k < K; k++) found |= own.sga[k] == ia && own.sgb[k] == ib && own.sgf[k] == fl;
      mine &= found;
    }
    const int gshift = (int)(threadIdx.x & 63) / G * G;
    const unsigned long long bad = (__ballot(!mine) >> gshift) & (G >= 64 ? ~0ull : ((1ull << G) - 1ull));
    if (bad) return false;
  }
  const int r = own.keep[q];
  int kin = -1, kout = -1, ncr = 0;
  for (int k = 0; k < K; k++) {
    const int cr = own.creg[k];
    if (cr == 0xff) return false;            // a crossing of no region (a sliver of zero area): the translation may give it one
    if (cr == r) { ncr++; if (own.sgf[k] & 1) kin = k; else kout = k; }
  }
  if (ncr != 2 || kin < 0 || kout < 0) return false;
  const double dx = own.dxv[q], dy = own.dyv[q];
  const double* rgx = own.reg[0][0]; const double* rgy = own.reg[0][1];
  // slide of crossing k along its b-edge for the unit translation d, as a parameter step of that edge: t = (d x e_a) / (e_b x e_a)
  auto slide = [&](int k, double& ebx, double& eby) {
    const int ia = own.sga[k], ib = own.sgb[k];
    const double eax = own.ax[ia + 1] - own.ax[ia], eay = own.ay[ia + 1] - own.ay[ia];
    ebx = own.bx[ib + 1] - own.bx[ib]; eby = own.by[ib + 1] - own.by[ib];
    return (dx * eay - dy * eax) / (ebx * eay - eby * eax);          // (den == 0: inf / nan, refused by the tolerance test below)
  };
  // other regions of the item stay within (1 m + their slides) of where they were: their boxes, grown by that reach, must miss this one's
  const int nreg = own.nreg[0];
  if (nreg > 1) {
    double bx0 = __builtin_inf(), bx1 = -__builtin_inf(), by0 = __builtin_inf(), by1 = -__builtin_inf();
    for (int j = own.roff[0][r]; j < own.roff[0][r + 1]; j++) { bx0 = fmin(bx0, rgx[j]); bx1 = fmax(bx1, rgx[j]); by0 = fmin(by0, rgy[j]); by1 = fmax(by1, rgy[j]); }
    for (int t = 0; t < nreg; t++) {
      if (t == r) continue;
      double reach = 0.0;
      for (int k = 0; k < K; k++) if (own.creg[k] == t) { double ebx, eby; const double tt = slide(k, ebx, eby); reach = fmax(reach, fabs(tt) * sqrt(ebx * ebx + eby * eby)); }
      reach += 1.001;
      bool apart = false;
      { double lo = __builtin_inf(), hi = -__builtin_inf(); for (int j = own.roff[0][t]; j < own.roff[0][t + 1]; j++) { lo = fmin(lo, rgx[j]); hi = fmax(hi, rgx[j]); } apart |= hi + reach < bx0 || bx1 < lo - reach; }
      { double lo = __builtin_inf(), hi = -__builtin_inf(); for (int j = own.roff[0][t]; j < own.roff[0][t + 1]; j++) { lo = fmin(lo, rgy[j]); hi = fmax(hi, rgy[j]); } apart |= hi + reach < by0 || by1 < lo - reach; }
      if (!apart) return false;            // (reach = nan compares false everywhere: refused)
    }
  }
  const double cix = rgx[own.cpos[kin]], ciy = rgy[own.cpos[kin]], cox = rgx[own.cpos[kout]], coy = rgy[own.cpos[kout]];
  double ebx, eby;
  const double tin = slide(kin, ebx, eby);
  double second = -(dx * (tin * eby) - dy * (tin * ebx));
  const double tout = slide(kout, ebx, eby);
  second += dx * (tout * eby) - dy * (tout * ebx);
  const double dA = (double)own.roa * ((dx * (coy - ciy) - dy * (cox - cix)) + 0.5 * second);
  // round-off floor of the reference's two area sums and of the crossing points: n eps Lc^2 with a wide margin
  const int nr = own.roff[0][r + 1] - own.roff[0][r];
  const double Lc = fmax(fmax(fabs(cix), fabs(ciy)), fmax(fabs(cox), fabs(coy)));
  const double tol = 256.0 * (double)(nr + 8) * 2.220446049250313e-16 * Lc * Lc + 1e-9 * own.rarea[0][r];
  if (!(fabs(dA) > tol)) return false;
  // the new lens intersects the old one: a p2 vertex lies on both boundaries unless both crossings sit on ONE edge of p2 -- then the
  // old and the new interval of that edge must overlap (parameters along the edge; (ebx, eby) is that edge's vector after the call above)
  if (own.sgb[kin] == own.sgb[kout]) {
    const int ib = own.sgb[kin];
    const double e2 = ebx * ebx + eby * eby;
    const double t0 = ((cix - own.bx[ib]) * ebx + (ciy - own.by[ib]) * eby) / e2, t1 = ((cox - own.bx[ib]) * ebx + (coy - own.by[ib]) * eby) / e2;
    const double lo = fmin(t0, t1), hi = fmax(t0, t1), lo1 = fmin(t0 + tin, t1 + tout), hi1 = fmax(t0 + tin, t1 + tout);
    if (!(fmax(lo, lo1) < fmin(hi, hi1))) return false;
  }
  if (gl == 0 && dA > 0.0) { own.dxv[q] = dx * -1; own.dyv[q] = dy * -1; }
  return true;
}

// The direction check of calc_normal_force (collisions.jl:58-68) for kept region q of the item whose rings and contact
// regions are in `own`: p1 translated by the unit normal, clipped against p2 again; every new region that intersects
// the old one and is larger flips the sign.  Works in `scr` (crossing arrays, region buffer 1), touches nothing of `own`
// but the sign of (dxv, dyv)[q] and its error word.
// after the check clip (clip<G>(own, scr, gl, dxv[q], dyv[q], .., buffer 1, ..))
template <int G, class MEM>
SZ_DEV void check_post(MEM& own, MEM& scr, int gl, int q, Stamps& st) {
  const int r = own.keep[q];
  const double area = own.rarea[0][r];
  const double dirx = own.dxv[q], diry = own.dyv[q];
  const double* rx = &own.reg[0][0][own.roff[0][r]]; const double* ry = &own.reg[0][1][own.roff[0][r]];
  const int nr = own.roff[0][r + 1] - own.roff[0][r];
  const int nn = scr.nreg[1];
  bool flip = false;
  for (int t = 0; t < nn; t++) {
    const double* nx_ = &scr.reg[1][0][scr.roff[1][t]]; const double* ny_ = &scr.reg[1][1][scr.roff[1][t]];
    int nnr = scr.roff[1][t + 1] - scr.roff[1][t];
    // area ratio first: `intersects && ratio > 1` needs the (costlier) predicate only then
    if (scr.rarea[1][t] / area > 1 && rings_intersect<G>(scr, gl, nx_, ny_, nnr, rx, ry, nr)) flip = !flip;
  }
  gsync();
  if (gl == 0) {
    if (flip) { own.dxv[q] = dirx * -1; own.dyv[q] = diry * -1; }
    const int e = scr.err & CAPBITS;           // a working set that was too small is the ITEM's problem (it is retried)
    if (e) { atomicOr(&own.ierr, e); scr.err &= ~CAPBITS; }
  }
  STAMP(st, 10);
}

// park (may be null; LDS): the rows are left there as well, for the lanes that add them to the floes' fixed-point totals (fx_word)
template <int G, class MEM>
SZ_DEV int finish_phase(MEM& m, int gl, const ItemCtx& cx_, double* out, int max_rows, Stamps& st, double* park = nullptr) {
  const int nkeep = m.nkeep;
  const double force_factor = m.ff;
  int nrows = 0;
  for (int q = 0; q < nkeep; q++) {
    const int r = m.keep[q];
    const double area = m.rarea[0][r], dl = m.dlv[q];
    const double dirx = m.dxv[q], diry = m.dyv[q];
    double fxn = 0.0, fyn = 0.0, px = 0.0, py = 0.0;
    if (area != 0) {
      px = m.rcx[r]; py = m.rcy[r];
      fxn = dirx * area * force_factor;
      fyn = diry * area * force_factor;
    }
    STAMP(st, 13);
    // _normal_direction_correct! (boundaries.jl:37,73,110,147)
    if (cx_.elem_dir == 0 && py >= cx_.elem_val) fxn = 0.0;
    if (cx_.elem_dir == 1 && py <= cx_.elem_val) fxn = 0.0;
    if (cx_.elem_dir == 2 && px >= cx_.elem_val) fyn = 0.0;
    if (cx_.elem_dir == 3 && px <= cx_.elem_val) fyn = 0.0;
    // calc_friction_forces (collisions.jl:243-283)
    const Body bi{ m.kin[KIN_I], m.kin[KIN_I + 1], m.kin[KIN_I + 2], m.kin[KIN_I + 3], m.kin[KIN_I + 4], 0 };
    const Body bj{ m.kin[KIN_J], m.kin[KIN_J + 1], m.kin[KIN_J + 2], m.kin[KIN_J + 3], m.kin[KIN_J + 4], cx_.rigid_j };
    double G_ = cx_.E / (2 * (1 + cx_.nu));
    double nnorm = sqrt(fxn * fxn + fyn * fyn);
    double iu = bi.u + bi.xi * (px - bi.cx), iv = bi.v + bi.xi * (py - bi.cy);
    double ju = bj.rigid_uv ? bj.u : bj.u + bj.xi * (px - bj.cx);
    double jv = bj.rigid_uv ? bj.v : bj.v + bj.xi * (py - bj.cy);
    double udiff = iu - ju, vdiff = iv - jv;
    double vnorm = sqrt(udiff * udiff + vdiff * vdiff);
    double xdir = 0.0, ydir = 0.0;
    if (udiff != 0 || vdiff != 0) { xdir = udiff / vnorm; ydir = vdiff / vnorm; }
    double dot_dir = xdir * udiff + ydir * vdiff;
    double xf = G_ * dl * cx_.dt * nnorm * xdir * -dot_dir;
    double yf = G_ * dl * cx_.dt * nnorm * ydir * -dot_dir;
    double norm_fric = sqrt(xf * xf + yf * yf);
    if (norm_fric > cx_.mu * nnorm) { xf = -cx_.mu * nnorm * xdir; yf = -cx_.mu * nnorm * ydir; }
    double fx = fxn + xf, fy = fyn + yf;
    if (fx != 0 || fy != 0) {
      if (nrows < max_rows) {
        if (gl == 0) {
          double* o = out + nrows * 5; o[0] = fx; o[1] = fy; o[2] = px; o[3] = py; o[4] = area;
          if (park) { double* q = park + nrows * 5; q[0] = fx; q[1] = fy; q[2] = px; q[3] = py; q[4] = area; }
        }
      } else if (gl == 0) m.ierr |= ERR_CAP_ROWS;
      nrows++;
    }
  }
  return nrows < max_rows ? nrows : max_rows;
}

}  // namespace szg
