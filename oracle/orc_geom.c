/*
 * orc_geom.c — CPU ORACLE (test infrastructure, see orc.h): polygon arithmetic.
 *
 * Restates the GeometryOps.jl 0.1.x operations the reference's hot path calls
 * (GeometryOps is a Project.toml:38 dependency, source absent from /root/reference):
 *   GO.intersection(p1, p2; target = PolygonTrait())   src/floe_utils.jl:55
 *   GO.intersection_points(p1, p2)                      src/physical_processes/collisions.jl:156
 *   GO.area / GO.centroid                               collisions.jl:360,178
 *   GO.signed_distance(point, poly)                     collisions.jl:91
 *   GO.coveredby(point, poly)                           collisions.jl:99
 *   GO.intersects(poly, poly)                           collisions.jl:64
 *
 * Clipping follows the Greiner-Hormann boundary trace that GeometryOps' Foster-Hormann
 * implementation reduces to for transversal crossings: both rings are turned into node lists
 * with the crossing points inserted in edge order, every crossing is flagged entry/exit, and
 * regions are traced starting from the first unprocessed crossing along p1 (so region order =
 * order of first crossing along p1's ring, as the reference's tests pin,
 * test_collisions.jl:68-77).  Degenerate contacts (vertex on edge, collinear edges) are
 * resolved by a symbolic perturbation (p2 translated by an infinitesimal generic vector, see
 * side_a_vs_b / side_b_vs_a).  This reproduces the reference's
 * known answers for its degenerate test inputs (test_collisions.jl:83-102,125-133).
 */
#define _GNU_SOURCE
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ containers */
void orc_ring_init(orc_ring *r) { r->n = 0; r->cap = 0; r->p = NULL; }
void orc_ring_free(orc_ring *r) { free(r->p); r->p = NULL; r->n = r->cap = 0; }
void orc_ring_push(orc_ring *r, double x, double y) {
  if (r->n == r->cap) {
    r->cap = r->cap ? 2 * r->cap : 8;
    r->p = (orc_pt *)realloc(r->p, (size_t)r->cap * sizeof(orc_pt));
  }
  r->p[r->n].x = x; r->p[r->n].y = y; r->n++;
}
void orc_ring_copy(orc_ring *dst, const orc_ring *src) {
  dst->n = 0;
  for (int i = 0; i < src->n; i++) orc_ring_push(dst, src->p[i].x, src->p[i].y);
}
void orc_ring_from_xy(orc_ring *r, int n, const double *x, const double *y) {
  r->n = 0;
  for (int i = 0; i < n; i++) orc_ring_push(r, x[i], y[i]);
}
void orc_regions_init(orc_regions *rg) { rg->n = 0; rg->cap = 0; rg->r = NULL; }
void orc_regions_free(orc_regions *rg) {
  for (int i = 0; i < rg->n; i++) orc_ring_free(&rg->r[i]);
  free(rg->r); rg->r = NULL; rg->n = rg->cap = 0;
}
static orc_ring *regions_new(orc_regions *rg) {
  if (rg->n == rg->cap) {
    rg->cap = rg->cap ? 2 * rg->cap : 4;
    rg->r = (orc_ring *)realloc(rg->r, (size_t)rg->cap * sizeof(orc_ring));
  }
  orc_ring_init(&rg->r[rg->n]);
  return &rg->r[rg->n++];
}

/* ------------------------------------------------------------------ area / centroid */
/* GO._signed_area: shoelace on raw coordinates, ring order, then the closing edge
   (zero when the ring is explicitly closed). */
double orc_signed_area(const orc_ring *r) {
  if (r->n == 0) return 0.0;
  double area = 0.0;
  orc_pt p1 = r->p[0];
  for (int i = 1; i < r->n; i++) {
    orc_pt p2 = r->p[i];
    area += p1.x * p2.y - p1.y * p2.x;
    p1 = p2;
  }
  orc_pt p2 = r->p[0];
  area += p1.x * p2.y - p1.y * p2.x;
  return area / 2.0;
}
double orc_area(const orc_ring *r) { return fabs(orc_signed_area(r)); }

/* GO.centroid_and_area for a linear ring: area-weighted, raw coordinates */
void orc_centroid(const orc_ring *r, double *cx, double *cy) {
  double xc = 0.0, yc = 0.0, area = 0.0;
  orc_pt p1 = r->p[0];
  for (int i = 1; i < r->n; i++) {
    orc_pt p2 = r->p[i];
    double ac = p1.x * p2.y - p2.x * p1.y;
    area += ac;
    xc += (p1.x + p2.x) * ac;
    yc += (p1.y + p2.y) * ac;
    p1 = p2;
  }
  area /= 2.0;
  *cx = xc / (6.0 * area);
  *cy = yc / (6.0 * area);
}

/* ------------------------------------------------------------------ predicates */
static inline double orient(orc_pt a, orc_pt b, orc_pt c) {
  return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
}
/* Tie rule = symbolic perturbation: ring b is thought of as translated by eps*(1, delta),
   0 < delta << 1, eps -> 0+.  A point of a that lies exactly on the line of a b-edge (r->s),
   or a point of b exactly on the line of an a-edge, then has the side it would have after that
   translation.  Because it is one rigid motion of b, every tie is broken consistently with an
   actual geometric configuration, so the traced topology is always valid (collinear edges,
   vertex-on-edge and vertex-on-vertex contacts included). */
static inline int side_a_vs_b(orc_pt r, orc_pt s, orc_pt p) {   /* p of a against edge r->s of b */
  double o = orient(r, s, p);
  if (o > 0.0) return 1;
  if (o < 0.0) return -1;
  double dx = s.x - r.x, dy = s.y - r.y;
  if (dy != 0.0) return dy > 0.0 ? 1 : -1;
  return dx > 0.0 ? -1 : 1;
}
static inline int side_b_vs_a(orc_pt p, orc_pt q, orc_pt r) {   /* r of b against edge p->q of a */
  double o = orient(p, q, r);
  if (o > 0.0) return 1;
  if (o < 0.0) return -1;
  double dx = q.x - p.x, dy = q.y - p.y;
  if (dy != 0.0) return dy > 0.0 ? -1 : 1;
  return dx > 0.0 ? 1 : -1;
}

static inline int on_segment(orc_pt a, orc_pt b, orc_pt c) {
  /* c collinear with ab assumed; is it within the segment's box? */
  return fmin(a.x, b.x) <= c.x && c.x <= fmax(a.x, b.x) && fmin(a.y, b.y) <= c.y &&
         c.y <= fmax(a.y, b.y);
}

/* GO.coveredby(point, polygon): inside or on the boundary */
int orc_coveredby(double x, double y, const orc_ring *r) {
  orc_pt pt = { x, y };
  int inside = 0;
  for (int i = 0; i + 1 < r->n; i++) {
    orc_pt a = r->p[i], b = r->p[i + 1];
    double o = orient(a, b, pt);
    if (o == 0.0 && on_segment(a, b, pt)) return 1;
    if ((a.y > y) != (b.y > y)) {
      /* edge straddles the horizontal line through pt: is the crossing to the right? */
      if (b.y > a.y) { if (o > 0.0) inside = !inside; }
      else           { if (o < 0.0) inside = !inside; }
    }
  }
  return inside;
}

/* GO._euclid_distance(point, segment) */
static double dist_pt_seg(double x0, double y0, orc_pt a, orc_pt b) {
  double vx = b.x - a.x, vy = b.y - a.y;
  double wx = x0 - a.x, wy = y0 - a.y;
  double c1 = wx * vx + wy * vy;
  if (c1 <= 0.0) return sqrt((x0 - a.x) * (x0 - a.x) + (y0 - a.y) * (y0 - a.y));
  double c2 = vx * vx + vy * vy;
  if (c2 <= c1) return sqrt((x0 - b.x) * (x0 - b.x) + (y0 - b.y) * (y0 - b.y));
  double b2 = c1 / c2;
  double px = a.x + b2 * vx, py = a.y + b2 * vy;
  return sqrt((x0 - px) * (x0 - px) + (y0 - py) * (y0 - py));
}

/* |GO.signed_distance(point, polygon)| = distance to the exterior ring */
double orc_dist_to_ring(double x, double y, const orc_ring *r) {
  double md = INFINITY;
  for (int i = 0; i + 1 < r->n; i++) {
    double d = dist_pt_seg(x, y, r->p[i], r->p[i + 1]);
    if (d < md) md = d;
  }
  return md;
}

static int seg_seg_touch(orc_pt p, orc_pt q, orc_pt r, orc_pt s) {
  double o1 = orient(p, q, r), o2 = orient(p, q, s), o3 = orient(r, s, p), o4 = orient(r, s, q);
  if (((o1 > 0 && o2 < 0) || (o1 < 0 && o2 > 0)) && ((o3 > 0 && o4 < 0) || (o3 < 0 && o4 > 0)))
    return 1;
  if (o1 == 0 && on_segment(p, q, r)) return 1;
  if (o2 == 0 && on_segment(p, q, s)) return 1;
  if (o3 == 0 && on_segment(r, s, p)) return 1;
  if (o4 == 0 && on_segment(r, s, q)) return 1;
  return 0;
}

/* GO.intersects(poly, poly) = !disjoint: closed sets share at least one point */
int orc_intersects(const orc_ring *a, const orc_ring *b) {
  if (a->n < 2 || b->n < 2) return 0;
  for (int i = 0; i + 1 < a->n; i++)
    for (int j = 0; j + 1 < b->n; j++)
      if (seg_seg_touch(a->p[i], a->p[i + 1], b->p[j], b->p[j + 1])) return 1;
  if (orc_coveredby(a->p[0].x, a->p[0].y, b)) return 1;
  if (orc_coveredby(b->p[0].x, b->p[0].y, a)) return 1;
  return 0;
}

/* ------------------------------------------------------------------ crossings */
typedef struct {
  int ia, ib;        /* edge indices on a and b                               */
  double ta, tb;     /* parameters along those edges                          */
  double x, y;       /* the crossing point (computed once, from a's edge)     */
  int ent_a, ent_b;  /* walking FORWARD along a (b) enters b (a) here         */
  int pos_a, pos_b;  /* node positions in the two lists                       */
  int visited;
} xing_t;

static void extent(const orc_ring *r, double *x0, double *x1, double *y0, double *y1) {
  *x0 = *y0 = INFINITY; *x1 = *y1 = -INFINITY;
  for (int i = 0; i < r->n; i++) {
    if (r->p[i].x < *x0) *x0 = r->p[i].x;
    if (r->p[i].x > *x1) *x1 = r->p[i].x;
    if (r->p[i].y < *y0) *y0 = r->p[i].y;
    if (r->p[i].y > *y1) *y1 = r->p[i].y;
  }
}

static int find_crossings(const orc_ring *a, const orc_ring *b, xing_t **out) {
  *out = NULL;
  if (a->n < 4 || b->n < 4) return 0;
  double ax0, ax1, ay0, ay1, bx0, bx1, by0, by1;
  extent(a, &ax0, &ax1, &ay0, &ay1);
  extent(b, &bx0, &bx1, &by0, &by1);
  if (ax1 < bx0 || bx1 < ax0 || ay1 < by0 || by1 < ay0) return 0;
  int oa = orc_signed_area(a) >= 0.0 ? 1 : -1;
  int ob = orc_signed_area(b) >= 0.0 ? 1 : -1;
  int n = 0, cap = 0;
  xing_t *xs = NULL;
  for (int ia = 0; ia + 1 < a->n; ia++) {
    orc_pt p = a->p[ia], q = a->p[ia + 1];
    for (int ib = 0; ib + 1 < b->n; ib++) {
      orc_pt r = b->p[ib], s = b->p[ib + 1];
      int sp = side_a_vs_b(r, s, p), sq = side_a_vs_b(r, s, q);
      if (sp == sq) continue;
      int sr = side_b_vs_a(p, q, r), ss = side_b_vs_a(p, q, s);
      if (sr == ss) continue;
      double rx = q.x - p.x, ry = q.y - p.y;   /* a edge vector */
      double sx = s.x - r.x, sy = s.y - r.y;   /* b edge vector */
      double denom = rx * sy - ry * sx;
      if (denom == 0.0) continue;
      double wx = r.x - p.x, wy = r.y - p.y;
      double ta = (wx * sy - wy * sx) / denom;
      double tb = (wx * ry - wy * rx) / denom;
      if (ta < 0.0) ta = 0.0;
      if (ta > 1.0) ta = 1.0;
      if (tb < 0.0) tb = 0.0;
      if (tb > 1.0) tb = 1.0;
      if (n == cap) { cap = cap ? 2 * cap : 8; xs = (xing_t *)realloc(xs, (size_t)cap * sizeof(xing_t)); }
      xing_t *c = &xs[n++];
      c->ia = ia; c->ib = ib; c->ta = ta; c->tb = tb;
      c->x = p.x + ta * rx; c->y = p.y + ta * ry;
      c->ent_a = (sp * ob) < 0;   /* a's edge start is outside b => forward along a enters b */
      c->ent_b = (sr * oa) < 0;
      c->pos_a = c->pos_b = -1; c->visited = 0;
    }
  }
  *out = xs;
  return n;
}

/* GO.intersection_points: every edge-edge crossing, exact duplicates removed */
int orc_intersection_points(const orc_ring *a, const orc_ring *b, orc_pt **pts) {
  xing_t *xs; int n = find_crossings(a, b, &xs);
  orc_pt *o = (orc_pt *)malloc((size_t)(n > 0 ? n : 1) * sizeof(orc_pt));
  int m = 0;
  for (int k = 0; k < n; k++) {
    int dup = 0;
    for (int l = 0; l < m; l++) if (o[l].x == xs[k].x && o[l].y == xs[k].y) { dup = 1; break; }
    if (!dup) { o[m].x = xs[k].x; o[m].y = xs[k].y; m++; }
  }
  free(xs);
  *pts = o;
  return m;
}

/* ------------------------------------------------------------------ intersection */
typedef struct { double x, y; int xid; } node_t;   /* xid >= 0: crossing index */

/* build the node list of `r`: vertex i, then the crossings on edge i in parameter order */
static int build_list(const orc_ring *r, xing_t *xs, int nx, int on_a, node_t *list) {
  int n = 0;
  int *tmp = (int *)malloc((size_t)(nx > 0 ? nx : 1) * sizeof(int));
  for (int e = 0; e + 1 < r->n; e++) {
    list[n].x = r->p[e].x; list[n].y = r->p[e].y; list[n].xid = -1; n++;
    int m = 0;
    for (int k = 0; k < nx; k++) if ((on_a ? xs[k].ia : xs[k].ib) == e) tmp[m++] = k;
    /* insertion sort by parameter, stable in discovery order */
    for (int u = 1; u < m; u++) {
      int k = tmp[u]; double t = on_a ? xs[k].ta : xs[k].tb; int v = u - 1;
      while (v >= 0 && (on_a ? xs[tmp[v]].ta : xs[tmp[v]].tb) > t) { tmp[v + 1] = tmp[v]; v--; }
      tmp[v + 1] = k;
    }
    for (int u = 0; u < m; u++) {
      int k = tmp[u];
      list[n].x = xs[k].x; list[n].y = xs[k].y; list[n].xid = k;
      if (on_a) xs[k].pos_a = n; else xs[k].pos_b = n;
      n++;
    }
  }
  free(tmp);
  return n;
}

/* strict point-in-ring: 1 inside, 0 outside, -1 exactly on the boundary */
static int point_in_ring3(double x, double y, const orc_ring *r) {
  orc_pt pt = { x, y };
  int inside = 0;
  for (int i = 0; i + 1 < r->n; i++) {
    orc_pt a = r->p[i], b = r->p[i + 1];
    double o = orient(a, b, pt);
    if (o == 0.0 && on_segment(a, b, pt)) return -1;
    if ((a.y > y) != (b.y > y)) {
      if (b.y > a.y) { if (o > 0.0) inside = !inside; }
      else           { if (o < 0.0) inside = !inside; }
    }
  }
  return inside;
}
/* the boundaries do not cross: a lies inside b iff its first vertex that is not exactly on
   b's boundary does (all on the boundary: identical rings, counted as inside) */
static int ring_inside(const orc_ring *a, const orc_ring *b) {
  for (int i = 0; i + 1 < a->n; i++) {
    int c = point_in_ring3(a->p[i].x, a->p[i].y, b);
    if (c >= 0) return c;
  }
  return 1;
}

/* diagnostics: traces abandoned by the guard (inconsistent crossing flags from round-off) */
static long orc_trace_failures_ = 0;
long orc_trace_failures(void) { return orc_trace_failures_; }

void orc_intersection(const orc_ring *a, const orc_ring *b, orc_regions *out) {
  out->n = 0;
  if (a->n < 4 || b->n < 4) return;
  xing_t *xs; int nx = find_crossings(a, b, &xs);
  if (nx == 0) {
    free(xs);
    /* no boundary crossing: one ring may lie inside the other */
    double ax0, ax1, ay0, ay1, bx0, bx1, by0, by1;
    extent(a, &ax0, &ax1, &ay0, &ay1);
    extent(b, &bx0, &bx1, &by0, &by1);
    if (ax1 < bx0 || bx1 < ax0 || ay1 < by0 || by1 < ay0) return;
    if (ring_inside(a, b)) orc_ring_copy(regions_new(out), a);
    else if (ring_inside(b, a)) orc_ring_copy(regions_new(out), b);
    return;
  }
  node_t *la = (node_t *)malloc((size_t)(a->n + nx) * sizeof(node_t));
  node_t *lb = (node_t *)malloc((size_t)(b->n + nx) * sizeof(node_t));
  int na = build_list(a, xs, nx, 1, la);
  int nb = build_list(b, xs, nx, 0, lb);
  int guard_max = 2 * (na + nb) + 8;
  /* start regions at unprocessed crossings in order along a */
  for (int s = 0; s < na; s++) {
    if (la[s].xid < 0 || xs[la[s].xid].visited) continue;
    int c0 = la[s].xid;
    orc_ring reg; orc_ring_init(&reg);
    orc_ring_push(&reg, xs[c0].x, xs[c0].y);
    xs[c0].visited = 1;
    int cur = c0, on_a = 1, guard = 0, ok = 1;
    do {
      node_t *L = on_a ? la : lb;
      int nL = on_a ? na : nb;
      int idx = on_a ? xs[cur].pos_a : xs[cur].pos_b;
      int dir = (on_a ? xs[cur].ent_a : xs[cur].ent_b) ? 1 : -1;
      for (;;) {
        idx += dir;
        if (idx >= nL) idx = 0;
        if (idx < 0) idx = nL - 1;
        orc_ring_push(&reg, L[idx].x, L[idx].y);
        if (++guard > guard_max) { ok = 0; break; }
        if (L[idx].xid >= 0) { cur = L[idx].xid; break; }
      }
      if (!ok) break;
      xs[cur].visited = 1;
      on_a = !on_a;
    } while (cur != c0);
    /* rings with fewer than 3 distinct points or with exactly zero area (slivers that only exist
       because of the symbolic perturbation: touching edges) are not regions */
    if (ok && reg.n >= 4 && orc_signed_area(&reg) != 0.0) {
      orc_ring *dst = regions_new(out);
      *dst = reg;
    } else {
      if (!ok) {
#ifdef _OPENMP
#pragma omp atomic
#endif
        orc_trace_failures_++;
      }
      orc_ring_free(&reg);
    }
  }
  free(la); free(lb); free(xs);
}

/* ------------------------------------------------------------------ flat wrappers */
int orc_clip_flat(int na, const double *ax, const double *ay, int nb, const double *bx,
                  const double *by, int max_regions, int max_pts, int *reg_off, double *rx,
                  double *ry) {
  orc_ring a, b; orc_ring_init(&a); orc_ring_init(&b);
  orc_ring_from_xy(&a, na, ax, ay); orc_ring_from_xy(&b, nb, bx, by);
  orc_regions rg; orc_regions_init(&rg);
  orc_intersection(&a, &b, &rg);
  int nr = rg.n, tot = 0, ret = nr;
  reg_off[0] = 0;
  for (int k = 0; k < nr && k < max_regions; k++) {
    if (tot + rg.r[k].n > max_pts) { ret = -1; break; }
    for (int i = 0; i < rg.r[k].n; i++) { rx[tot] = rg.r[k].p[i].x; ry[tot] = rg.r[k].p[i].y; tot++; }
    reg_off[k + 1] = tot;
  }
  if (nr > max_regions) ret = -1;
  orc_regions_free(&rg); orc_ring_free(&a); orc_ring_free(&b);
  return ret;
}

int orc_ipoints_flat(int na, const double *ax, const double *ay, int nb, const double *bx,
                     const double *by, int max_pts, double *px, double *py) {
  orc_ring a, b; orc_ring_init(&a); orc_ring_init(&b);
  orc_ring_from_xy(&a, na, ax, ay); orc_ring_from_xy(&b, nb, bx, by);
  orc_pt *pts; int n = orc_intersection_points(&a, &b, &pts);
  for (int i = 0; i < n && i < max_pts; i++) { px[i] = pts[i].x; py[i] = pts[i].y; }
  free(pts); orc_ring_free(&a); orc_ring_free(&b);
  return n;
}
