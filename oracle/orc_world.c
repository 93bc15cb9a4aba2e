/*
 * orc_world.c — CPU ORACLE (test infrastructure, see orc.h): floe state and the
 * reference's process functions, restated function by function.
 *
 * Floe indices stored in the `floeidx` interaction column are 1-BASED (as the reference
 * stores them, collisions.jl:297) and negative for domain elements (-1 N, -2 S, -3 E, -4 W,
 * -(4+k) topography k, collisions.jl:608-660).  Everything else in the C API is 0-based.
 */
#define _GNU_SOURCE
#define _POSIX_C_SOURCE 200809L
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* interaction row layout, floe.jl:102-110 */
enum { I_IDX = 0, I_FX, I_FY, I_PX, I_PY, I_TRQ, I_OVER, I_NCOL };

typedef struct {
  orc_ring poly;
  double cx, cy, rmax, area, height, mass, moment, alpha, u, v, xi;
  double p_dxdt, p_dydt, p_dalphadt, p_dudt, p_dvdt, p_dxidt;
  double fxOA, fyOA, trqOA, hflx, overarea;
  double cfx, cfy, ctrq;
  double sa[4], si[4], strain[4];   /* order: 11, 12, 21, 22 */
  int status;
  int *fuse_idx; int nfuse, capfuse;
  int64_t id, ghost_id;
  int *ghosts; int nghosts, capghosts;
  double *inter; int ninter, capinter;
  double *sx, *sy; int nsub;
} floe_t;

typedef struct {
  int kind; double val, u, v;
  double x0, x1, y0, y1;   /* rectangle extent */
  orc_ring poly;
} bound_t;

typedef struct { orc_ring poly; double cx, cy, rmax; } topo_t;

typedef struct { int floe; double dx, dy, tx, ty; int n; } cell_ent;
struct cell_list { cell_ent *e; int n, cap; };

struct orc_world {
  floe_t *f; int M, cap;
  bound_t b[4];
  topo_t *topo; int ntopo;
  /* constants */
  double E, nu, mu, rho_o, rho_a, Cd_io, Cd_ia, fcor, turn;
  /* settings */
  double ff_max_overlap, fd_max_overlap, rho_i, max_h, max_xi, lambda; int dd;
  /* grid */
  int Nx, Ny; double gx0, gxf, gy0, gyf, gdx, gdy;
  double *uo, *vo, *hf, *ua, *va;
  /* two-way coupling (coupling.jl:1617-1680): per centre cell (cells centred on the grid lines) the floes whose
     sub-floe points fell into it, CellFloes / CellStresses of grids.jl:4-8 and oceans.jl:4-8 */
  int two_way; double Cd_ao, k_ice, L_ice; int dt_couple;
  double *t_ocn, *t_atm, *tau_x, *tau_y, *si_frac;
  struct cell_list *cells;
  /* last broad-phase result */
  int32_t *pi, *pj; int npairs, cappairs;
  int64_t warn[4];
  int nthreads;
  /* wall seconds per phase since the last orc_reset_phase_times (bench.py prints where the CPU path spends its time):
     0 add_ghosts!, 1 pair loop (all-pairs bounding circles, threaded over i), 2 Dict pass (serial), 3 floe-floe / floe-domain
     interactions (threaded over i), 4 mirror + ghost fold + totals (serial, as in the reference), 5 timestep_coupling! (serial
     loop over floes, as in the reference), 6 timestep_floe_properties! (threaded) */
  double tphase[8];
};
#include <time.h>
static double wall_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
void orc_get_phase_times(const orc_world *w, double *out8) { for (int k = 0; k < 8; k++) out8[k] = w->tphase[k]; }
void orc_reset_phase_times(orc_world *w) { for (int k = 0; k < 8; k++) w->tphase[k] = 0.0; }

/* ------------------------------------------------------------------ helpers */
static void floe_init(floe_t *f) { memset(f, 0, sizeof(*f)); orc_ring_init(&f->poly); f->status = ORC_ACTIVE; }
static void floe_free(floe_t *f) {
  orc_ring_free(&f->poly); free(f->fuse_idx); free(f->ghosts); free(f->inter); free(f->sx); free(f->sy);
}
static void push_int(int **a, int *n, int *cap, int v) {
  if (*n == *cap) { *cap = *cap ? 2 * *cap : 4; *a = (int *)realloc(*a, (size_t)*cap * sizeof(int)); }
  (*a)[(*n)++] = v;
}
/* deepcopy_floe, floe_utils.jl:120-161 */
static void floe_copy(floe_t *d, const floe_t *s) {
  *d = *s;
  orc_ring_init(&d->poly); orc_ring_copy(&d->poly, &s->poly);
  d->fuse_idx = NULL; d->capfuse = 0; d->nfuse = 0;
  for (int i = 0; i < s->nfuse; i++) push_int(&d->fuse_idx, &d->nfuse, &d->capfuse, s->fuse_idx[i]);
  d->ghosts = NULL; d->capghosts = 0; d->nghosts = 0;
  for (int i = 0; i < s->nghosts; i++) push_int(&d->ghosts, &d->nghosts, &d->capghosts, s->ghosts[i]);
  /* the Floe constructor gives the copy a fresh (empty) interactions matrix and num_inters = 0 */
  d->inter = NULL; d->ninter = 0; d->capinter = 0;
  d->sx = d->sy = NULL;
  if (s->nsub > 0) {
    d->sx = (double *)malloc((size_t)s->nsub * sizeof(double));
    d->sy = (double *)malloc((size_t)s->nsub * sizeof(double));
    memcpy(d->sx, s->sx, (size_t)s->nsub * sizeof(double));
    memcpy(d->sy, s->sy, (size_t)s->nsub * sizeof(double));
  }
}
static floe_t *world_push(orc_world *w) {
  if (w->M == w->cap) { w->cap = w->cap ? 2 * w->cap : 64; w->f = (floe_t *)realloc(w->f, (size_t)w->cap * sizeof(floe_t)); }
  floe_init(&w->f[w->M]);
  return &w->f[w->M++];
}
/* _make_bounding_box_polygon, floe_utils.jl:104-108 */
static void bbox_poly(orc_ring *r, double xmin, double xmax, double ymin, double ymax) {
  r->n = 0;
  orc_ring_push(r, xmin, ymin); orc_ring_push(r, xmin, ymax); orc_ring_push(r, xmax, ymax);
  orc_ring_push(r, xmax, ymin); orc_ring_push(r, xmin, ymin);
}
/* _translate_poly, floe_utils.jl:60-64 */
static void translate_ring(orc_ring *dst, const orc_ring *src, double dx, double dy) {
  dst->n = 0;
  for (int i = 0; i < src->n; i++) orc_ring_push(dst, src->p[i].x + dx, src->p[i].y + dy);
}
/* _translate_floe!, floe_utils.jl:66-72 */
static void translate_floe(floe_t *f, double dx, double dy) {
  f->cx += dx; f->cy += dy;
  for (int i = 0; i < f->poly.n; i++) { f->poly.p[i].x += dx; f->poly.p[i].y += dy; }
}
/* calc_max_radius, floe_utils.jl:301-313 */
static double calc_max_radius(const orc_ring *r, double cx, double cy) {
  double m = 0.0;
  for (int i = 0; i < r->n; i++) {
    double x = r->p[i].x - cx, y = r->p[i].y - cy;
    double rs = x * x + y * y;
    if (rs > m) m = rs;
  }
  return sqrt(m);
}
/* _calc_moment_inertia, floe_utils.jl:273-298 (the centroid is subtracted twice in wi: literal) */
static double calc_moment_inertia(const orc_ring *r, double xc, double yc, double height, double rho_i) {
  double Ixx = 0.0, Iyy = 0.0, x1 = 0.0, y1 = 0.0;
  for (int i = 0; i < r->n; i++) {
    double x2 = r->p[i].x - xc, y2 = r->p[i].y - yc;
    if (i == 0) { x1 = x2; y1 = y2; continue; }
    double wi = (x1 - xc) * (y2 - yc) - (x2 - xc) * (y1 - yc);
    Ixx += wi * (y1 * y1 + y1 * y2 + y2 * y2);
    Iyy += wi * (x1 * x1 + x1 * x2 + x2 * x2);
    x1 = x2; y1 = y2;
  }
  Ixx *= 1.0 / 12.0; Iyy *= 1.0 / 12.0;
  return fabs(Ixx + Iyy) * height * rho_i;
}

/* ------------------------------------------------------------------ world setup */
orc_world *orc_create(void) {
  orc_world *w = (orc_world *)calloc(1, sizeof(orc_world));
  /* Constants(), simulation.jl:5-18 */
  w->E = 6e6; w->nu = 0.3; w->mu = 0.2; w->rho_o = 1027.0; w->rho_a = 1.2;
  w->Cd_io = 3e-3; w->Cd_ia = 1e-3; w->fcor = 1.4e-4; w->turn = 15.0 * M_PI / 180.0;
  w->ff_max_overlap = 0.55; w->fd_max_overlap = 0.75; w->rho_i = 920.0; w->max_h = 10.0;
  w->max_xi = 1e-5; w->lambda = 0.2; w->dd = 1;
  for (int k = 0; k < 4; k++) { w->b[k].kind = ORC_OPEN; orc_ring_init(&w->b[k].poly); }
  w->nthreads = 1;
  return w;
}
void orc_destroy(orc_world *w) {
  if (!w) return;
  for (int i = 0; i < w->M; i++) floe_free(&w->f[i]);
  free(w->f);
  for (int k = 0; k < 4; k++) orc_ring_free(&w->b[k].poly);
  for (int k = 0; k < w->ntopo; k++) orc_ring_free(&w->topo[k].poly);
  free(w->topo); free(w->uo); free(w->vo); free(w->hf); free(w->ua); free(w->va);
  free(w->t_ocn); free(w->t_atm); free(w->tau_x); free(w->tau_y); free(w->si_frac);
  if (w->cells) { size_t nc = (size_t)(w->Nx + 1) * (size_t)(w->Ny + 1); for (size_t q = 0; q < nc; q++) free(w->cells[q].e); free(w->cells); }
  free(w->pi); free(w->pj);
  free(w);
}
void orc_set_threads(orc_world *w, int n) { w->nthreads = n < 1 ? 1 : n; }
void orc_set_consts(orc_world *w, double E, double nu, double mu, double rho_o, double rho_a,
                    double Cd_io, double Cd_ia, double f, double turn_theta) {
  w->E = E; w->nu = nu; w->mu = mu; w->rho_o = rho_o; w->rho_a = rho_a;
  w->Cd_io = Cd_io; w->Cd_ia = Cd_ia; w->fcor = f; w->turn = turn_theta;
}
void orc_set_settings(orc_world *w, double ffmo, double fdmo, double rho_i, double max_h,
                      double max_xi, double lambda, int dd) {
  w->ff_max_overlap = ffmo; w->fd_max_overlap = fdmo; w->rho_i = rho_i; w->max_h = max_h;
  w->max_xi = max_xi; w->lambda = lambda; w->dd = dd;
}
/* _boundary_info_from_extent, boundaries.jl:29,65,102,139 */
void orc_set_domain_extent(orc_world *w, const int *kinds, double x0, double xf, double y0,
                           double yf, const double *bu, const double *bv) {
  double dx = (xf - x0) / 2, dy = (yf - y0) / 2;
  for (int k = 0; k < 4; k++) {
    bound_t *b = &w->b[k];
    b->kind = kinds[k]; b->u = bu ? bu[k] : 0.0; b->v = bv ? bv[k] : 0.0;
    switch (k) {
      case ORC_NORTH: b->x0 = x0 - dx; b->x1 = xf + dx; b->y0 = yf; b->y1 = yf + dy; b->val = yf; break;
      case ORC_SOUTH: b->x0 = x0 - dx; b->x1 = xf + dx; b->y0 = y0 - dy; b->y1 = y0; b->val = y0; break;
      case ORC_EAST:  b->x0 = xf; b->x1 = xf + dx; b->y0 = y0 - dy; b->y1 = yf + dy; b->val = xf; break;
      default:        b->x0 = x0 - dx; b->x1 = x0; b->y0 = y0 - dy; b->y1 = yf + dy; b->val = x0; break;
    }
    bbox_poly(&b->poly, b->x0, b->x1, b->y0, b->y1);
  }
}
void orc_get_boundary_vals(const orc_world *w, double *vals4) { for (int k = 0; k < 4; k++) vals4[k] = w->b[k].val; }
/* the boundary polygons as they stand (5 points each, x then y: 10 doubles per boundary, order N, S, E, W) */
void orc_get_boundary_polys(const orc_world *w, double *xy40) {
  for (int k = 0; k < 4; k++) for (int q = 0; q < 5; q++) { xy40[k * 10 + q] = w->b[k].poly.p[q].x; xy40[k * 10 + 5 + q] = w->b[k].poly.p[q].y; }
}
void orc_set_topography(orc_world *w, int ntopo, const int *off, const double *x, const double *y) {
  for (int k = 0; k < w->ntopo; k++) orc_ring_free(&w->topo[k].poly);
  free(w->topo);
  w->topo = (topo_t *)calloc((size_t)(ntopo > 0 ? ntopo : 1), sizeof(topo_t));
  w->ntopo = ntopo;
  for (int k = 0; k < ntopo; k++) {
    topo_t *t = &w->topo[k];
    orc_ring_init(&t->poly);
    orc_ring_from_xy(&t->poly, off[k + 1] - off[k], x + off[k], y + off[k]);
    orc_centroid(&t->poly, &t->cx, &t->cy);
    t->rmax = calc_max_radius(&t->poly, t->cx, t->cy);
  }
}
static double *dupd(const double *s, size_t n) { double *d = (double *)malloc(n * sizeof(double)); memcpy(d, s, n * sizeof(double)); return d; }
void orc_set_grid_fields(orc_world *w, int Nx, int Ny, double x0, double xf, double y0, double yf,
                         const double *uo, const double *vo, const double *hflx, const double *ua,
                         const double *va) {
  free(w->uo); free(w->vo); free(w->hf); free(w->ua); free(w->va);
  if (w->cells) { size_t nc = (size_t)(w->Nx + 1) * (size_t)(w->Ny + 1); for (size_t q = 0; q < nc; q++) free(w->cells[q].e); free(w->cells); }
  free(w->t_ocn); free(w->t_atm); free(w->tau_x); free(w->tau_y); free(w->si_frac);
  w->Nx = Nx; w->Ny = Ny; w->gx0 = x0; w->gxf = xf; w->gy0 = y0; w->gyf = yf;
  w->gdx = (xf - x0) / Nx; w->gdy = (yf - y0) / Ny;
  size_t n = (size_t)(Nx + 1) * (size_t)(Ny + 1);
  w->uo = dupd(uo, n); w->vo = dupd(vo, n); w->hf = dupd(hflx, n); w->ua = dupd(ua, n); w->va = dupd(va, n);
  w->t_ocn = (double *)calloc(n, sizeof(double)); w->t_atm = (double *)calloc(n, sizeof(double));
  w->tau_x = (double *)calloc(n, sizeof(double)); w->tau_y = (double *)calloc(n, sizeof(double));
  w->si_frac = (double *)calloc(n, sizeof(double));
  w->cells = (struct cell_list *)calloc(n, sizeof(struct cell_list));
}

/* Floe{FT}(poly, hmean, 0), floe.jl:144-200 */
int orc_add_floe(orc_world *w, int n, const double *x, const double *y, double height) {
  floe_t *f = world_push(w);
  /* valid_ringvec!, floe_utils.jl:10-17: drop adjacent duplicates, close the ring */
  for (int i = 0; i < n; i++) {
    if (i + 1 < n && x[i] == x[i + 1] && y[i] == y[i + 1]) continue;
    orc_ring_push(&f->poly, x[i], y[i]);
  }
  if (f->poly.p[0].x != f->poly.p[f->poly.n - 1].x || f->poly.p[0].y != f->poly.p[f->poly.n - 1].y)
    orc_ring_push(&f->poly, f->poly.p[0].x, f->poly.p[0].y);
  orc_centroid(&f->poly, &f->cx, &f->cy);
  f->height = height;
  f->area = orc_area(&f->poly);
  f->mass = f->area * height * w->rho_i;
  f->moment = calc_moment_inertia(&f->poly, f->cx, f->cy, height, w->rho_i);
  f->rmax = calc_max_radius(&f->poly, f->cx, f->cy);
  f->id = w->M;   /* initialize_floe_field sets id = index (1-based), floe.jl */
  return w->M - 1;
}
void orc_set_subpoints(orc_world *w, int i, int n, const double *sx, const double *sy) {
  floe_t *f = &w->f[i];
  free(f->sx); free(f->sy);
  f->sx = dupd(sx, (size_t)(n > 0 ? n : 1)); f->sy = dupd(sy, (size_t)(n > 0 ? n : 1)); f->nsub = n;
}
int orc_num_floes(const orc_world *w) { return w->M; }

static double *field_ptr(floe_t *f, int field) {
  switch (field) {
    case ORC_F_CX: return &f->cx; case ORC_F_CY: return &f->cy; case ORC_F_RMAX: return &f->rmax;
    case ORC_F_AREA: return &f->area; case ORC_F_HEIGHT: return &f->height; case ORC_F_MASS: return &f->mass;
    case ORC_F_MOMENT: return &f->moment; case ORC_F_ALPHA: return &f->alpha; case ORC_F_U: return &f->u;
    case ORC_F_V: return &f->v; case ORC_F_XI: return &f->xi;
    case ORC_F_P_DXDT: return &f->p_dxdt; case ORC_F_P_DYDT: return &f->p_dydt;
    case ORC_F_P_DALPHADT: return &f->p_dalphadt; case ORC_F_P_DUDT: return &f->p_dudt;
    case ORC_F_P_DVDT: return &f->p_dvdt; case ORC_F_P_DXIDT: return &f->p_dxidt;
    case ORC_F_FXOA: return &f->fxOA; case ORC_F_FYOA: return &f->fyOA; case ORC_F_TRQOA: return &f->trqOA;
    case ORC_F_HFLX: return &f->hflx; case ORC_F_OVERAREA: return &f->overarea;
    case ORC_F_COLL_FX: return &f->cfx; case ORC_F_COLL_FY: return &f->cfy; case ORC_F_COLL_TRQ: return &f->ctrq;
    default:
      if (field >= ORC_F_SA11 && field <= ORC_F_SA22) return &f->sa[field - ORC_F_SA11];
      if (field >= ORC_F_SI11 && field <= ORC_F_SI22) return &f->si[field - ORC_F_SI11];
      if (field >= ORC_F_E11 && field <= ORC_F_E22) return &f->strain[field - ORC_F_E11];
  }
  return NULL;
}
void orc_get_field(const orc_world *w, int field, double *out) {
  for (int i = 0; i < w->M; i++) out[i] = *field_ptr((floe_t *)&w->f[i], field);
}
void orc_set_field(orc_world *w, int field, const double *in) {
  for (int i = 0; i < w->M; i++) *field_ptr(&w->f[i], field) = in[i];
}
void orc_get_ids(const orc_world *w, int64_t *id, int64_t *ghost_id, int32_t *status) {
  for (int i = 0; i < w->M; i++) {
    if (id) id[i] = w->f[i].id;
    if (ghost_id) ghost_id[i] = w->f[i].ghost_id;
    if (status) status[i] = w->f[i].status;
  }
}
void orc_set_ids(orc_world *w, const int64_t *id) { for (int i = 0; i < w->M; i++) w->f[i].id = id[i]; }
void orc_set_status(orc_world *w, const int32_t *s) { for (int i = 0; i < w->M; i++) w->f[i].status = s[i]; }
int orc_total_ring_points(const orc_world *w) { int t = 0; for (int i = 0; i < w->M; i++) t += w->f[i].poly.n; return t; }
void orc_get_rings(const orc_world *w, int32_t *off, double *x, double *y) {
  int t = 0; off[0] = 0;
  for (int i = 0; i < w->M; i++) {
    for (int k = 0; k < w->f[i].poly.n; k++) { x[t] = w->f[i].poly.p[k].x; y[t] = w->f[i].poly.p[k].y; t++; }
    off[i + 1] = t;
  }
}
int orc_total_interactions(const orc_world *w) { int t = 0; for (int i = 0; i < w->M; i++) t += w->f[i].ninter; return t; }
void orc_get_interactions(const orc_world *w, int32_t *off, double *rows) {
  int t = 0; off[0] = 0;
  for (int i = 0; i < w->M; i++) {
    memcpy(rows + (size_t)t * I_NCOL, w->f[i].inter, (size_t)w->f[i].ninter * I_NCOL * sizeof(double));
    t += w->f[i].ninter; off[i + 1] = t;
  }
}
int orc_total_ghost_links(const orc_world *w) { int t = 0; for (int i = 0; i < w->M; i++) t += w->f[i].nghosts; return t; }
void orc_get_ghosts(const orc_world *w, int32_t *off, int32_t *idx) {
  int t = 0; off[0] = 0;
  for (int i = 0; i < w->M; i++) { for (int k = 0; k < w->f[i].nghosts; k++) idx[t++] = w->f[i].ghosts[k]; off[i + 1] = t; }
}
int orc_total_fuse(const orc_world *w) { int t = 0; for (int i = 0; i < w->M; i++) t += w->f[i].nfuse; return t; }
void orc_get_fuse(const orc_world *w, int32_t *off, int32_t *idx) {
  int t = 0; off[0] = 0;
  for (int i = 0; i < w->M; i++) { for (int k = 0; k < w->f[i].nfuse; k++) idx[t++] = w->f[i].fuse_idx[k]; off[i + 1] = t; }
}
int orc_num_pairs(const orc_world *w) { return w->npairs; }
void orc_get_pairs(const orc_world *w, int32_t *pi, int32_t *pj) {
  memcpy(pi, w->pi, (size_t)w->npairs * sizeof(int32_t)); memcpy(pj, w->pj, (size_t)w->npairs * sizeof(int32_t));
}
void orc_get_warn_counts(const orc_world *w, int64_t *out4) { memcpy(out4, w->warn, sizeof(w->warn)); }

/* ------------------------------------------------------------------ collisions: forces */

/* which_vertices_match_points, floe_utils.jl:331-352 (atol = 1; note the sqrt of a distance) */
static int which_vertices_match_points(const orc_pt *pts, int npts, const orc_ring *region, int *idx) {
  int m = 0, np = npts;
  if (np > 0 && pts[0].x == pts[np - 1].x && pts[0].y == pts[np - 1].y) np -= 1;
  for (int i = 0; i < np; i++) {
    double min_dist = INFINITY; int min_vert = 0;
    for (int j = 0; j < region->n; j++) {
      double dx = region->p[j].x - pts[i].x, dy = region->p[j].y - pts[i].y;
      double dist = sqrt(sqrt(dx * dx + dy * dy));
      if (dist < min_dist) { min_dist = dist; min_vert = j; }
    }
    if (min_dist < 1.0) idx[m++] = min_vert;
  }
  for (int u = 1; u < m; u++) { int k = idx[u], v = u - 1; while (v >= 0 && idx[v] > k) { idx[v + 1] = idx[v]; v--; } idx[v + 1] = k; }
  return m;
}

/* which_vertices_match_points on plain arrays (tests: test_floe_utils.jl:74-137): 0-based indices out, returns the count */
int orc_which_vertices_match_points(int npts, const double *px, const double *py, int nr, const double *rx, const double *ry, int32_t *idx) {
  orc_pt *pts = (orc_pt *)malloc(sizeof(orc_pt) * (size_t)(npts > 0 ? npts : 1));
  orc_ring region; region.n = nr; region.cap = nr; region.p = (orc_pt *)malloc(sizeof(orc_pt) * (size_t)(nr > 0 ? nr : 1));
  for (int i = 0; i < npts; i++) { pts[i].x = px[i]; pts[i].y = py[i]; }
  for (int j = 0; j < nr; j++) { region.p[j].x = rx[j]; region.p[j].y = ry[j]; }
  int *tmp = (int *)malloc(sizeof(int) * (size_t)(npts > 0 ? npts : 1));
  int m = which_vertices_match_points(pts, npts, &region, tmp);
  for (int k = 0; k < m; k++) idx[k] = tmp[k];
  free(tmp); free(pts); free(region.p);
  return m;
}

/* _many_intersect_normal_force!, collisions.jl:78-119 */
static double many_intersect_normal_force(double *force_dir, const orc_ring *region, const orc_ring *poly, double force_factor) {
  double x1 = 0, y1 = 0, dl = 0, fx = 0, fy = 0; int n_pts = 0;
  for (int i = 0; i < region->n; i++) {
    double x2 = region->p[i].x, y2 = region->p[i].y;
    if (i == 0) { x1 = x2; y1 = y2; continue; }
    double xmid = 0.5 * (x2 + x1), ymid = 0.5 * (y2 + y1);
    double dist = orc_dist_to_ring(xmid, ymid, poly);
    if (dist < 1e-8) {
      double dx = x2 - x1, dy = y2 - y1;
      double mag = sqrt(dx * dx + dy * dy);
      double xt = xmid + (-dy / (100 * mag));
      double yt = ymid + (dx / (100 * mag));
      int in_region = orc_coveredby(xt, yt, region);
      double f_sign = in_region ? 1.0 : -1.0;
      double Fnx = (f_sign * force_factor) * (-dy), Fny = (f_sign * force_factor) * dx;
      dl += mag; n_pts += 1; fx += Fnx; fy += Fny;
    }
    x1 = x2; y1 = y2;
  }
  if (0 < n_pts && n_pts < region->n - 1) {
    dl /= n_pts;
    if (dl > 0.1) {
      double nrm = sqrt(fx * fx + fy * fy);
      force_dir[0] = fx / nrm; force_dir[1] = fy / nrm;
    }
  }
  return dl;
}

/* calc_normal_force, collisions.jl:30-70 */
static double calc_normal_force(const orc_ring *p1, const orc_ring *p2, const orc_ring *region, double area,
                                const orc_pt *ipts, int nip, double force_factor, double *force) {
  double dir[2] = { 0.0, 0.0 };
  int *p = (int *)malloc((size_t)(nip > 0 ? nip : 1) * sizeof(int));
  int m = which_vertices_match_points(ipts, nip, region, p);
  double dl = 0.0;
  if (m == 2) {
    orc_pt pt1 = region->p[p[0]], pt2 = region->p[p[1]];
    double dx = pt2.x - pt1.x, dy = pt2.y - pt1.y;
    dl = sqrt(dx * dx + dy * dy);
    if (dl > 0.1) { dir[0] = -dy / dl; dir[1] = dx / dl; }
  } else if (m != 0) {
    dl = many_intersect_normal_force(dir, region, p1, force_factor);
  }
  free(p);
  if (dl > 0.1) {
    orc_ring p1new; orc_ring_init(&p1new);
    translate_ring(&p1new, p1, dir[0], dir[1]);
    orc_regions nr; orc_regions_init(&nr);
    orc_intersection(&p1new, p2, &nr);
    for (int k = 0; k < nr.n; k++) {
      if (orc_intersects(&nr.r[k], region) && orc_area(&nr.r[k]) / area > 1) { dir[0] *= -1; dir[1] *= -1; }
    }
    orc_regions_free(&nr); orc_ring_free(&p1new);
  }
  force[0] = dir[0] * area * force_factor;
  force[1] = dir[1] * area * force_factor;
  return dl;
}

/* calc_elastic_forces, collisions.jl:149-188. regions/areas are edited in place (small
   regions deleted); returns ncontact and fills force/fpoint/dl (ncontact entries). */
static int calc_elastic_forces(const orc_ring *p1, const orc_ring *p2, orc_regions *regions, double *areas,
                               double force_factor, double *force, double *fpoint, double *dl) {
  orc_pt *ipts; int nip = orc_intersection_points(p1, p2, &ipts);
  int ncontact = 0;
  if (nip >= 2) {
    int n1 = p1->n - 1, n2 = p2->n - 1;
    double min_area = (double)((n1 < n2 ? n1 : n2) * 100) / 1.75;
    for (int i = regions->n - 1; i >= 0; i--) {
      if (areas[i] < min_area) {
        orc_ring_free(&regions->r[i]);
        for (int k = i; k + 1 < regions->n; k++) { regions->r[k] = regions->r[k + 1]; areas[k] = areas[k + 1]; }
        regions->n--;
      } else ncontact++;
    }
  }
  for (int k = 0; k < ncontact; k++) {
    force[2 * k] = force[2 * k + 1] = 0.0; fpoint[2 * k] = fpoint[2 * k + 1] = 0.0; dl[k] = 0.0;
    if (areas[k] != 0) {
      orc_centroid(&regions->r[k], &fpoint[2 * k], &fpoint[2 * k + 1]);
      dl[k] = calc_normal_force(p1, p2, &regions->r[k], areas[k], ipts, nip, force_factor, &force[2 * k]);
    }
  }
  free(ipts);
  return ncontact;
}

/* calc_friction_forces, collisions.jl:243-283; (ju, jv) is the velocity of the other body at
   the force point (floe: _get_velocity :206-214; boundary/topography: boundaries.jl:522,565,
   topography.jl:76) supplied through the callback-free form below */
static void friction_one(const orc_world *w, const floe_t *fi, double ju, double jv, double px, double py,
                         const double *normal, double dl, int dt, double *out) {
  double G = w->E / (2 * (1 + w->nu));
  double nnorm = sqrt(normal[0] * normal[0] + normal[1] * normal[1]);
  double iu = fi->u + fi->xi * (px - fi->cx);
  double iv = fi->v + fi->xi * (py - fi->cy);
  double udiff = iu - ju, vdiff = iv - jv;
  double vnorm = sqrt(udiff * udiff + vdiff * vdiff);
  double xdir = 0.0, ydir = 0.0;
  if (udiff != 0 || vdiff != 0) { xdir = udiff / vnorm; ydir = vdiff / vnorm; }
  double dot_dir = xdir * udiff + ydir * vdiff;
  double xf = G * dl * dt * nnorm * xdir * -dot_dir;
  double yf = G * dl * dt * nnorm * ydir * -dot_dir;
  double norm_fric = sqrt(xf * xf + yf * yf);
  if (norm_fric > w->mu * nnorm) { xf = -w->mu * nnorm * xdir; yf = -w->mu * nnorm * ydir; }
  out[0] = xf; out[1] = yf;
}

/* add_interactions!, collisions.jl:285-309 */
static void add_interaction_row(floe_t *f, double idx, double fx, double fy, double px, double py, double over) {
  if (fx != 0 || fy != 0) {
    if (f->ninter == f->capinter) {
      f->capinter = f->capinter ? 2 * f->capinter : 4;
      f->inter = (double *)realloc(f->inter, (size_t)f->capinter * I_NCOL * sizeof(double));
    }
    double *r = f->inter + (size_t)f->ninter * I_NCOL;
    r[I_IDX] = idx; r[I_FX] = fx; r[I_FY] = fy; r[I_PX] = px; r[I_PY] = py; r[I_TRQ] = 0.0; r[I_OVER] = over;
    f->ninter++;
    f->overarea += over;
  }
}

/* floe_floe_interaction!, collisions.jl:347-408 (i, j 0-based here) */
void orc_floe_floe_interaction(orc_world *w, int i, int j, int dt, double max_overlap) {
  floe_t *fi = &w->f[i], *fj = &w->f[j];
  orc_regions rg; orc_regions_init(&rg);
  orc_intersection(&fi->poly, &fj->poly, &rg);
  int nr = rg.n;
  double *areas = (double *)malloc((size_t)(nr > 0 ? nr : 1) * sizeof(double));
  double total = 0.0;
  for (int k = 0; k < nr; k++) { areas[k] = orc_area(&rg.r[k]); total += areas[k]; }
  if (total > 0) {
    double r1 = total / fi->area, r2 = total / fj->area;
    if ((r1 > r2 ? r1 : r2) > max_overlap) {
      fi->status = ORC_FUSE;
      push_int(&fi->fuse_idx, &fi->nfuse, &fi->capfuse, j);
    } else {
      double ih = fi->height, ir = sqrt(fi->area), jh = fj->height, jr = sqrt(fj->area);
      double ff;
      if (ir > 1e5 || jr > 1e5) ff = w->E * (ih < jh ? ih : jh) / (ir < jr ? ir : jr);
      else ff = w->E * (ih * jh) / (ih * jr + jh * ir);
      double *force = (double *)malloc((size_t)nr * 2 * sizeof(double));
      double *fpoint = (double *)malloc((size_t)nr * 2 * sizeof(double));
      double *dl = (double *)malloc((size_t)nr * sizeof(double));
      int np = calc_elastic_forces(&fi->poly, &fj->poly, &rg, areas, ff, force, fpoint, dl);
      for (int k = 0; k < np; k++) {
        double px = fpoint[2 * k], py = fpoint[2 * k + 1], fr[2];
        double ju = fj->u + fj->xi * (px - fj->cx), jv = fj->v + fj->xi * (py - fj->cy);
        friction_one(w, fi, ju, jv, px, py, &force[2 * k], dl[k], dt, fr);
        add_interaction_row(fi, (double)(j + 1), force[2 * k] + fr[0], force[2 * k + 1] + fr[1], px, py, areas[k]);
      }
      free(force); free(fpoint); free(dl);
    }
  }
  free(areas);
  orc_regions_free(&rg);
}

/* floe_domain_element_interaction!, collisions.jl:427-557.
   elem_kind: boundary kind, or -1 for topography; dir: ORC_NORTH.. for boundaries */
static void floe_domain_element_interaction(orc_world *w, floe_t *f, const orc_ring *epoly, int elem_kind,
                                            int dir, double val, double eu, double ev, double elem_idx,
                                            int dt, double max_overlap) {
  if (elem_kind == ORC_PERIODIC) return;
  orc_regions rg; orc_regions_init(&rg);
  orc_intersection(&f->poly, epoly, &rg);
  int nr = rg.n;
  if (elem_kind == ORC_OPEN) {
    double a = 0.0;
    for (int k = 0; k < nr; k++) a += orc_area(&rg.r[k]);
    if (a > 0) f->status = ORC_REMOVE;
    orc_regions_free(&rg);
    return;
  }
  double *areas = (double *)malloc((size_t)(nr > 0 ? nr : 1) * sizeof(double));
  double max_area = 0.0;
  for (int k = 0; k < nr; k++) { areas[k] = orc_area(&rg.r[k]); if (areas[k] > max_area) max_area = areas[k]; }
  if (max_area > 0) {
    if (max_area / f->area > max_overlap) {
      f->status = ORC_REMOVE;
    } else {
      double ff = w->E * f->height / sqrt(f->area);
      double *force = (double *)malloc((size_t)nr * 2 * sizeof(double));
      double *fpoint = (double *)malloc((size_t)nr * 2 * sizeof(double));
      double *dl = (double *)malloc((size_t)nr * sizeof(double));
      int np = calc_elastic_forces(&f->poly, epoly, &rg, areas, ff, force, fpoint, dl);
      /* _normal_direction_correct!, boundaries.jl:37,73,110,147 (no-op for topography) */
      if (elem_kind >= 0) {
        for (int k = 0; k < np; k++) {
          double px = fpoint[2 * k], py = fpoint[2 * k + 1];
          if (dir == ORC_NORTH && py >= val) force[2 * k] = 0.0;
          if (dir == ORC_SOUTH && py <= val) force[2 * k] = 0.0;
          if (dir == ORC_EAST && px >= val) force[2 * k + 1] = 0.0;
          if (dir == ORC_WEST && px <= val) force[2 * k + 1] = 0.0;
        }
      }
      for (int k = 0; k < np; k++) {
        double px = fpoint[2 * k], py = fpoint[2 * k + 1], fr[2];
        friction_one(w, f, eu, ev, px, py, &force[2 * k], dl[k], dt, fr);
        add_interaction_row(f, elem_idx, force[2 * k] + fr[0], force[2 * k + 1] + fr[1], px, py, areas[k]);
      }
      free(force); free(fpoint); free(dl);
    }
  }
  free(areas);
  orc_regions_free(&rg);
}

/* floe_domain_interaction!, collisions.jl:594-662 */
void orc_floe_domain_interaction(orc_world *w, int i, int dt, double max_overlap) {
  floe_t *f = &w->f[i];
  bound_t *nb = &w->b[ORC_NORTH], *sb = &w->b[ORC_SOUTH], *eb = &w->b[ORC_EAST], *wb = &w->b[ORC_WEST];
#define BVEL(b) ((b)->kind == ORC_MOVING ? (b)->u : 0.0), ((b)->kind == ORC_MOVING ? (b)->v : 0.0)
  if (f->cy + f->rmax > nb->val)
    floe_domain_element_interaction(w, f, &nb->poly, nb->kind, ORC_NORTH, nb->val, BVEL(nb), -1.0, dt, max_overlap);
  if (f->cy - f->rmax < sb->val)
    floe_domain_element_interaction(w, f, &sb->poly, sb->kind, ORC_SOUTH, sb->val, BVEL(sb), -2.0, dt, max_overlap);
  if (f->cx + f->rmax > eb->val)
    floe_domain_element_interaction(w, f, &eb->poly, eb->kind, ORC_EAST, eb->val, BVEL(eb), -3.0, dt, max_overlap);
  if (f->cx - f->rmax < wb->val)
    floe_domain_element_interaction(w, f, &wb->poly, wb->kind, ORC_WEST, wb->val, BVEL(wb), -4.0, dt, max_overlap);
#undef BVEL
  for (int k = 0; k < w->ntopo; k++) {
    topo_t *t = &w->topo[k];
    double dx = t->cx - f->cx, dy = t->cy - f->cy, rr = t->rmax + f->rmax;
    if (dx * dx + dy * dy < rr * rr)
      floe_domain_element_interaction(w, f, &t->poly, -1, -1, 0.0, 0.0, 0.0, -(double)(4 + k + 1), dt, max_overlap);
  }
}

/* calc_torque!, collisions.jl:673-686 */
void orc_calc_torque(orc_world *w, int i) {
  floe_t *f = &w->f[i];
  for (int k = 0; k < f->ninter; k++) {
    double *r = f->inter + (size_t)k * I_NCOL;
    double xp = r[I_PX] - f->cx, yp = r[I_PY] - f->cy;
    r[I_TRQ] = xp * r[I_FY] - yp * r[I_FX];
  }
}

/* update_boundaries!, collisions.jl:565-571 and _update_boundary!, boundaries.jl:526-568 */
static void update_boundaries(orc_world *w, int dt) {
  for (int k = 0; k < 4; k++) {
    bound_t *b = &w->b[k];
    if (b->kind != ORC_MOVING) continue;
    if (k == ORC_NORTH || k == ORC_SOUTH) {
      double dy = b->v * dt; b->y0 += dy; b->y1 += dy; b->val += dy;
    } else {
      double dx = b->u * dt; b->x0 += dx; b->x1 += dx; b->val += dx;
    }
    bbox_poly(&b->poly, b->x0, b->x1, b->y0, b->y1);
  }
}

/* ---- Dict{Tuple{Int,Int},Tuple{Int,Int}} of collisions.jl:743 */
typedef struct { int64_t k1, k2, g1, g2; int used; } dent_t;
typedef struct { dent_t *e; size_t cap, n; } dict_t;
static size_t dhash(int64_t a, int64_t b, size_t cap) {
  uint64_t h = (uint64_t)a * 0x9E3779B97F4A7C15ull ^ ((uint64_t)b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
  h ^= h >> 29;
  return (size_t)(h & (cap - 1));
}
static void dict_init(dict_t *d, size_t cap) { d->cap = cap; d->n = 0; d->e = (dent_t *)calloc(cap, sizeof(dent_t)); }
static dent_t *dict_get_or_insert(dict_t *d, int64_t k1, int64_t k2, int64_t g1, int64_t g2) {
  if (2 * (d->n + 1) > d->cap) {
    dict_t nd; dict_init(&nd, d->cap * 2);
    for (size_t i = 0; i < d->cap; i++) if (d->e[i].used) dict_get_or_insert(&nd, d->e[i].k1, d->e[i].k2, d->e[i].g1, d->e[i].g2);
    free(d->e); *d = nd;
  }
  size_t h = dhash(k1, k2, d->cap);
  while (d->e[h].used) {
    if (d->e[h].k1 == k1 && d->e[h].k2 == k2) return &d->e[h];
    h = (h + 1) & (d->cap - 1);
  }
  d->e[h].used = 1; d->e[h].k1 = k1; d->e[h].k2 = k2; d->e[h].g1 = g1; d->e[h].g2 = g2; d->n++;
  return &d->e[h];
}

/* timestep_collisions!, collisions.jl:734-864.  The reference's threaded loop shares one Dict
   behind a SpinLock (first writer wins, thread-schedule dependent); this restatement reproduces
   the SINGLE-THREADED order: candidates are found per i (parallel-safe), the Dict is consulted
   serially in (i asc, j asc) order, and the interactions are then evaluated per i. */
void orc_timestep_collisions(orc_world *w, int n_init, int dt) {
  int M = w->M;
  floe_t *F = w->f;
  double t0 = wall_now(), t1;
  int **cand = (int **)calloc((size_t)(M > 0 ? M : 1), sizeof(int *));
  int *ncand = (int *)calloc((size_t)(M > 0 ? M : 1), sizeof(int));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(w->nthreads)
#endif
  for (int i = 0; i < M; i++) {
    F[i].cfx = F[i].cfy = 0.0; F[i].ctrq = 0.0; F[i].ninter = 0;
    int cap = 0, n = 0; int *c = NULL;
    for (int j = i + 1; j < M; j++) {
      if (F[i].id == F[j].id) continue;
      /* potential_interaction, collisions.jl:705-710 */
      double dx = F[i].cx - F[j].cx, dy = F[i].cy - F[j].cy, rr = F[i].rmax + F[j].rmax;
      if ((dx * dx + dy * dy) < rr * rr) {
        if (n == cap) { cap = cap ? 2 * cap : 8; c = (int *)realloc(c, (size_t)cap * sizeof(int)); }
        c[n++] = j;
      }
    }
    cand[i] = c; ncand[i] = n;
  }
  t1 = wall_now(); w->tphase[1] += t1 - t0; t0 = t1;
  /* serial Dict pass, collisions.jl:751-775 */
  dict_t d; dict_init(&d, 1024);
  w->npairs = 0;
  for (int i = 0; i < M; i++) {
    int keep = 0;
    for (int u = 0; u < ncand[i]; u++) {
      int j = cand[i][u];
      int64_t k1, k2, gp1, gp2;
      if (F[i].id > F[j].id) { k1 = F[i].id; k2 = F[j].id; gp1 = F[i].ghost_id; gp2 = F[j].ghost_id; }
      else { k1 = F[j].id; k2 = F[i].id; gp1 = F[j].ghost_id; gp2 = F[i].ghost_id; }
      dent_t *e = dict_get_or_insert(&d, k1, k2, gp1, gp2);
      int a = gp1 == e->g1, b = gp2 == e->g2;
      if ((a && b) || (a != b)) {
        cand[i][keep++] = j;
        if (w->npairs == w->cappairs) {
          w->cappairs = w->cappairs ? 2 * w->cappairs : 1024;
          w->pi = (int32_t *)realloc(w->pi, (size_t)w->cappairs * sizeof(int32_t));
          w->pj = (int32_t *)realloc(w->pj, (size_t)w->cappairs * sizeof(int32_t));
        }
        w->pi[w->npairs] = i; w->pj[w->npairs] = j; w->npairs++;
      }
    }
    ncand[i] = keep;
  }
  free(d.e);
  t1 = wall_now(); w->tphase[2] += t1 - t0; t0 = t1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(w->nthreads)
#endif
  for (int i = 0; i < M; i++) {
    for (int u = 0; u < ncand[i]; u++) orc_floe_floe_interaction(w, i, cand[i][u], dt, w->ff_max_overlap);
    orc_floe_domain_interaction(w, i, dt, w->fd_max_overlap);
    free(cand[i]);
  }
  free(cand); free(ncand);
  t1 = wall_now(); w->tphase[3] += t1 - t0; t0 = t1;
  update_boundaries(w, dt);
  /* mirror pass, collisions.jl:799-828 */
  for (int i = 0; i < M; i++) {
    if (F[i].status == ORC_FUSE) {
      int nf = F[i].nfuse;   /* iterate the list as it stands (pushes go to other floes) */
      for (int k = 0; k < nf; k++) {
        int idx = F[i].fuse_idx[k];
        F[idx].status = ORC_FUSE;
        push_int(&F[idx].fuse_idx, &F[idx].nfuse, &F[idx].capfuse, i);
      }
    }
    int ni = F[i].ninter;
    for (int k = 0; k < ni; k++) {
      double *r = F[i].inter + (size_t)k * I_NCOL;
      double j = r[I_IDX];
      if (j <= (double)M && j > (double)(i + 1)) {
        int jidx = (int)j - 1;
        double fx = r[I_FX], fy = r[I_FY], px = r[I_PX], py = r[I_PY], ov = r[I_OVER];
        add_interaction_row(&F[jidx], (double)(i + 1), fx, fy, px, py, ov);
        /* equal and opposite forces on the row that is now last in j */
        if (F[jidx].ninter > 0) {
          double *rj = F[jidx].inter + (size_t)(F[jidx].ninter - 1) * I_NCOL;
          rj[I_FX] *= -1; rj[I_FY] *= -1;
        }
      }
    }
  }
  /* ghost fold + totals, collisions.jl:830-862 */
  for (int i = 0; i < n_init && i < M; i++) {
    for (int gk = 0; gk < F[i].nghosts; gk++) {
      int g = F[i].ghosts[gk];
      int gnp = F[g].ninter;
      double sx = F[g].cx - F[i].cx, sy = F[g].cy - F[i].cy;
      for (int k = 0; k < gnp; k++) { double *r = F[g].inter + (size_t)k * I_NCOL; r[I_PX] -= sx; r[I_PY] -= sy; }
      for (int k = 0; k < gnp; k++) {
        double *r = F[g].inter + (size_t)k * I_NCOL;
        add_interaction_row(&F[i], (double)(i + 1), r[I_FX], r[I_FY], r[I_PX], r[I_PY], r[I_OVER]);
      }
      int inp = F[i].ninter;
      for (int k = 0; k < gnp; k++) F[i].inter[(size_t)(inp - gnp + k) * I_NCOL + I_IDX] = F[g].inter[(size_t)k * I_NCOL + I_IDX];
    }
    orc_calc_torque(w, i);
    double sfx = 0, sfy = 0, st = 0;
    for (int k = 0; k < F[i].ninter; k++) {
      double *r = F[i].inter + (size_t)k * I_NCOL;
      sfx += r[I_FX]; sfy += r[I_FY]; st += r[I_TRQ];
    }
    F[i].cfx += sfx; F[i].cfy += sfy; F[i].ctrq += st;
  }
  w->tphase[4] += wall_now() - t0;
}

/* ------------------------------------------------------------------ ghosts */
/* ghosts_on_bounds!, collisions.jl:881-901 */
static void ghosts_on_bounds(orc_world *w, int elem, const bound_t *b, double tx, double ty) {
  int nfloes = w->M, nghosts = 1;
  orc_regions rg; orc_regions_init(&rg);
  orc_intersection(&w->f[elem].poly, &b->poly, &rg);
  int nonempty = rg.n > 0;
  orc_regions_free(&rg);
  /* A floe that only TOUCHES the wall has no ghost: the reference's own case is the triangle whose edges lie on the west and
     south walls (test_collisions.jl:282-283, "no ghosts").  The boundary polygon is the half plane beyond the wall as far as a
     floe is concerned, so a region of positive area needs a vertex strictly beyond the wall; without that test the restated
     clipper can return a sliver of round-off area for collinear edges at a wall coordinate like 1.5e5 (Voronoi cells cut at the
     walls: tests/test_hip_parity.py::test_voronoi_field_touching_cells). */
  if (nonempty) {
    const orc_ring *r = &w->f[elem].poly;
    int beyond = 0;
    for (int k = 0; k < r->n && !beyond; k++)
      beyond = tx > 0 ? r->p[k].x < b->val : tx < 0 ? r->p[k].x > b->val : ty > 0 ? r->p[k].y < b->val : r->p[k].y > b->val;
    nonempty = beyond;
  }
  if (!nonempty) return;
  int ng = w->f[elem].nghosts;
  for (int k = 0; k < ng; k++) {
    int gi = w->f[elem].ghosts[k];
    floe_t *nf = world_push(w);
    floe_t tmp; floe_copy(&tmp, &w->f[gi]); *nf = tmp;
    nghosts++;
  }
  { floe_t *nf = world_push(w); floe_t tmp; floe_copy(&tmp, &w->f[elem]); *nf = tmp; }
  for (int i = nfloes; i < nfloes + nghosts; i++) translate_floe(&w->f[i], tx, ty);
}
/* find_ghosts!, collisions.jl:925-1003; axis 0 = east/west, 1 = north/south */
static void find_ghosts(orc_world *w, int elem, int axis) {
  const bound_t *maxb = &w->b[axis == 0 ? ORC_EAST : ORC_NORTH];
  const bound_t *minb = &w->b[axis == 0 ? ORC_WEST : ORC_SOUTH];
  double L = maxb->val - minb->val;
  int nfloes = w->M;
  double c = axis == 0 ? w->f[elem].cx : w->f[elem].cy, r = w->f[elem].rmax;
  if (c - r < minb->val) ghosts_on_bounds(w, elem, minb, axis == 0 ? L : 0.0, axis == 0 ? 0.0 : L);
  else if (c + r > maxb->val) ghosts_on_bounds(w, elem, maxb, axis == 0 ? -L : 0.0, axis == 0 ? 0.0 : -L);
  int new_n = w->M;
  if (new_n > nfloes) {
    c = axis == 0 ? w->f[elem].cx : w->f[elem].cy;
    if (c < minb->val) {
      translate_floe(&w->f[elem], axis == 0 ? L : 0.0, axis == 0 ? 0.0 : L);
      translate_floe(&w->f[new_n - 1], axis == 0 ? -L : 0.0, axis == 0 ? 0.0 : -L);
    } else if (maxb->val < c) {
      translate_floe(&w->f[elem], axis == 0 ? -L : 0.0, axis == 0 ? 0.0 : -L);
      translate_floe(&w->f[new_n - 1], axis == 0 ? L : 0.0, axis == 0 ? 0.0 : L);
    }
  }
}
/* add_floe_ghosts!, collisions.jl:1017-1047 */
static void add_floe_ghosts(orc_world *w, int axis) {
  int nfloes = w->M, n0 = w->M;
  for (int i = 0; i < n0; i++) {
    if (w->f[i].status == ORC_ACTIVE && w->f[i].ghost_id == 0) {
      find_ghosts(w, i, axis);
      int new_n = w->M;
      if (new_n > nfloes) {
        int ng = new_n - nfloes, base = w->f[i].nghosts;
        for (int k = 0; k < ng; k++) {
          w->f[nfloes + k].ghost_id = (k + 1) + base;
          w->f[nfloes + k].nghosts = 0;
        }
        for (int k = 0; k < ng; k++) push_int(&w->f[i].ghosts, &w->f[i].nghosts, &w->f[i].capghosts, nfloes + k);
        nfloes += ng;
      }
    }
  }
}
/* add_ghosts!, collisions.jl:1060-1174 */
void orc_add_ghosts(orc_world *w) {
  int ew = w->b[ORC_EAST].kind == ORC_PERIODIC && w->b[ORC_WEST].kind == ORC_PERIODIC;
  int ns = w->b[ORC_NORTH].kind == ORC_PERIODIC && w->b[ORC_SOUTH].kind == ORC_PERIODIC;
  if (ew) add_floe_ghosts(w, 0);
  if (ns) add_floe_ghosts(w, 1);
}
/* simulation.jl:138-144 */
void orc_remove_ghosts(orc_world *w, int n_init) {
  for (int i = n_init; i < w->M; i++) floe_free(&w->f[i]);
  if (w->M > n_init) w->M = n_init;
  for (int i = 0; i < w->M; i++) w->f[i].nghosts = 0;
}

/* ------------------------------------------------------------------ forcings (one-way coupling) */
/* bilinear sample on the grid-line lattice; equals Interpolations.linear_interpolation over the
   knot window of mc_interpolation (coupling.jl:845-902) incl. the periodic wrap of
   find_interp_knots (:702-744): line Nx+1 == line 1 and is not repeated */
static double sample(const orc_world *w, const double *A, double x, double y, int per_x, int per_y) {
  int Nx = w->Nx, Ny = w->Ny;
  double fxi = floor((x - w->gx0) / w->gdx), fyi = floor((y - w->gy0) / w->gdy);
  long ix = (long)fxi, iy = (long)fyi;
  if (!per_x) { if (ix < 0) ix = 0; if (ix > Nx - 1) ix = Nx - 1; }
  if (!per_y) { if (iy < 0) iy = 0; if (iy > Ny - 1) iy = Ny - 1; }
  double xk = w->gx0 + (double)ix * w->gdx, yk = w->gy0 + (double)iy * w->gdy;
  double tx = (x - xk) / w->gdx, ty = (y - yk) / w->gdy;
  long i0, i1, j0, j1;
  if (per_x) { i0 = ((ix % Nx) + Nx) % Nx; i1 = (((ix + 1) % Nx) + Nx) % Nx; } else { i0 = ix; i1 = ix + 1; }
  if (per_y) { j0 = ((iy % Ny) + Ny) % Ny; j1 = (((iy + 1) % Ny) + Ny) % Ny; } else { j0 = iy; j1 = iy + 1; }
  size_t s = (size_t)(Ny + 1);
  double a00 = A[(size_t)i0 * s + (size_t)j0], a01 = A[(size_t)i0 * s + (size_t)j1];
  double a10 = A[(size_t)i1 * s + (size_t)j0], a11 = A[(size_t)i1 * s + (size_t)j1];
  double c0 = (1.0 - ty) * a00 + ty * a01;
  double c1 = (1.0 - ty) * a10 + ty * a11;
  return (1.0 - tx) * c0 + tx * c1;
}

/* in_bounds, coupling.jl:494-597: four methods dispatched on the (north/south, east/west) boundary kinds -- a direction with a
   periodic pair admits every coordinate.  The expression orc_timestep_coupling evaluates per sub-floe point. */
int orc_in_bounds(const orc_world *w, double x, double y, int per_x, int per_y) {
  return (per_x || (w->gx0 <= x && x <= w->gxf)) && (per_y || (w->gy0 <= y && y <= w->gyf));
}
/* find_interp_knots, coupling.jl:702-744 (periodic) and :776-797 (non-periodic): the grid lines (1-based numbers in knot_idx, values in
   knots) around the points nearest to lines point_idx[0..npts), with a buffer of dd + 1 lines either side.  Periodic: line ncells + 1 IS
   line 1 and is not repeated; lines beyond an edge are the far side's, their values shifted by the grid length L.  Returns the number of
   knots (at most cap). */
int orc_find_interp_knots(int npts, const int *point_idx, int ncells, double g0, double dg, double L, int dd, int periodic,
                          int cap, double *knots, int *knot_idx) {
  int min_line = point_idx[0], max_line = point_idx[0], n = 0;
  for (int k = 1; k < npts; k++) { if (point_idx[k] < min_line) min_line = point_idx[k]; if (point_idx[k] > max_line) max_line = point_idx[k]; }
  min_line -= dd + 1; max_line += dd + 1;
#define ORC_KNOT(line, shift) do { if (n < cap) { knot_idx[n] = (line); knots[n] = g0 + ((line) - 1) * dg + (shift); } n++; } while (0)
  if (!periodic) {
    const int nlines = ncells + 1;
    if (min_line < 1) min_line = 1;
    if (max_line > nlines) max_line = nlines;
    for (int l = min_line; l <= max_line; l++) ORC_KNOT(l, 0.0);
    return n;
  }
  int lo0 = 1, lo1 = 0, hi0 = 1, hi1 = 0, in0, in1;          /* empty ranges */
  if (min_line < 1 && max_line > ncells) { lo0 = min_line + ncells; lo1 = ncells; hi0 = 1; hi1 = max_line - ncells; in0 = 1; in1 = ncells; }
  else if (min_line < 1) { lo0 = min_line + ncells; lo1 = ncells; in0 = 1; in1 = max_line; }
  else if (max_line > ncells) { hi0 = 1; hi1 = max_line - ncells; in0 = min_line; in1 = ncells; }
  else { in0 = min_line; in1 = max_line; }
  for (int l = lo0; l <= lo1; l++) ORC_KNOT(l, -L);
  for (int l = in0; l <= in1; l++) ORC_KNOT(l, 0.0);
  for (int l = hi0; l <= hi1; l++) ORC_KNOT(l, L);
#undef ORC_KNOT
  return n;
}
/* the grid lines sample() blends at (x, y) -- 1-based numbers west, east, south, north -- and its weights: what ties sample()'s wrap to
   find_interp_knots' knot_idx (tests/test_oracle_golden.py) */
void orc_sample_lines(const orc_world *w, double x, double y, int per_x, int per_y, int *lines4, double *t2) {
  int Nx = w->Nx, Ny = w->Ny;
  long ix = (long)floor((x - w->gx0) / w->gdx), iy = (long)floor((y - w->gy0) / w->gdy);
  if (!per_x) { if (ix < 0) ix = 0; if (ix > Nx - 1) ix = Nx - 1; }
  if (!per_y) { if (iy < 0) iy = 0; if (iy > Ny - 1) iy = Ny - 1; }
  t2[0] = (x - (w->gx0 + (double)ix * w->gdx)) / w->gdx; t2[1] = (y - (w->gy0 + (double)iy * w->gdy)) / w->gdy;
  long i0, i1, j0, j1;
  if (per_x) { i0 = ((ix % Nx) + Nx) % Nx; i1 = (((ix + 1) % Nx) + Nx) % Nx; } else { i0 = ix; i1 = ix + 1; }
  if (per_y) { j0 = ((iy % Ny) + Ny) % Ny; j1 = (((iy + 1) % Ny) + Ny) % Ny; } else { j0 = iy; j1 = iy + 1; }
  lines4[0] = (int)i0 + 1; lines4[1] = (int)i1 + 1; lines4[2] = (int)j0 + 1; lines4[3] = (int)j1 + 1;
}
/* the five lattices at (x, y) as the coupling samples them: uocn, vocn, hflx, uatm, vatm */
void orc_sample_fields(const orc_world *w, double x, double y, int per_x, int per_y, double *out5) {
  out5[0] = sample(w, w->uo, x, y, per_x, per_y); out5[1] = sample(w, w->vo, x, y, per_x, per_y); out5[2] = sample(w, w->hf, x, y, per_x, per_y);
  out5[3] = sample(w, w->ua, x, y, per_x, per_y); out5[4] = sample(w, w->va, x, y, per_x, per_y);
}

/* ------------------------------------------------------------------ two-way coupling */
void orc_set_two_way(orc_world *w, int on, double Cd_ao, double k, double L, int dt) {
  w->two_way = on; w->Cd_ao = Cd_ao; w->k_ice = k; w->L_ice = L; w->dt_couple = dt;
}
void orc_set_temps(orc_world *w, const double *t_ocn, const double *t_atm) {
  size_t n = (size_t)(w->Nx + 1) * (size_t)(w->Ny + 1);
  memcpy(w->t_ocn, t_ocn, n * sizeof(double)); memcpy(w->t_atm, t_atm, n * sizeof(double));
}
void orc_get_ocean_stress(const orc_world *w, double *tau_x, double *tau_y, double *si_frac, double *hflx) {
  size_t n = (size_t)(w->Nx + 1) * (size_t)(w->Ny + 1);
  memcpy(tau_x, w->tau_x, n * sizeof(double)); memcpy(tau_y, w->tau_y, n * sizeof(double));
  memcpy(si_frac, w->si_frac, n * sizeof(double)); memcpy(hflx, w->hf, n * sizeof(double));
}
void orc_clear_cells(orc_world *w) {
  size_t n = (size_t)(w->Nx + 1) * (size_t)(w->Ny + 1);
  for (size_t q = 0; q < n; q++) w->cells[q].n = 0;
}
/* shift_cell_idx, coupling.jl:1154-1178 (1-based grid-line index) */
int orc_shift_cell_idx(int idx, int nlines, int periodic) {
  if (!periodic) return idx;
  int ncells = nlines - 1;
  return idx < 1 ? (idx + ncells) : (ncells < idx ? (idx - ncells) : idx);
}
/* center_cell_coords, coupling.jl:1116-1140, with check_cell_bounds :931-1087; out = xmin, xmax, ymin, ymax */
void orc_center_cell_coords(const orc_world *w, int xidx, int yidx, int per_y, int per_x, double *out) {
  double xmin = (xidx - 1.5) * w->gdx + w->gx0, xmax = xmin + w->gdx;
  double ymin = (yidx - 1.5) * w->gdy + w->gy0, ymax = ymin + w->gdy;
  if (!per_x) {
    xmin = xmin < w->gx0 ? w->gx0 : (xmin > w->gxf ? w->gxf : xmin);
    xmax = xmax > w->gxf ? w->gxf : (xmax < w->gx0 ? w->gx0 : xmax);
  }
  if (!per_y) {
    ymin = ymin < w->gy0 ? w->gy0 : (ymin > w->gyf ? w->gyf : ymin);
    ymax = ymax > w->gyf ? w->gyf : (ymax < w->gy0 ? w->gy0 : ymax);
  }
  out[0] = xmin; out[1] = xmax; out[2] = ymin; out[3] = ymax;
}
/* floe_to_grid_info!, coupling.jl:1417-1454, with add_point! :1336-1360: one sub-floe point of floe `floeidx`
   (0-based here) in centre cell (xidx, yidx) (1-based, unshifted) with ocean stress (tx_ocn, ty_ocn) */
void orc_floe_to_grid_info(orc_world *w, int floeidx, int xidx, int yidx, double tx_ocn, double ty_ocn) {
  int per_x = w->b[ORC_EAST].kind == ORC_PERIODIC, per_y = w->b[ORC_NORTH].kind == ORC_PERIODIC;
  int sx = orc_shift_cell_idx(xidx, w->Nx + 1, per_x), sy = orc_shift_cell_idx(yidx, w->Ny + 1, per_y);
  double dx = (sx - xidx) * w->gdx, dy = (sy - yidx) * w->gdy;
  if (sx < 1 || sx > w->Nx + 1 || sy < 1 || sy > w->Ny + 1) return;   /* the reference would throw a BoundsError */
  struct cell_list *c = &w->cells[(size_t)(sx - 1) * (size_t)(w->Ny + 1) + (size_t)(sy - 1)];
  if (c->n == 0 || c->e[c->n - 1].floe != floeidx) {
    if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 4; c->e = (cell_ent *)realloc(c->e, (size_t)c->cap * sizeof(cell_ent)); }
    cell_ent *e = &c->e[c->n++];
    e->floe = floeidx; e->dx = dx; e->dy = dy; e->tx = -tx_ocn; e->ty = -ty_ocn; e->n = 1;
  } else {
    cell_ent *e = &c->e[c->n - 1];
    e->tx += -tx_ocn; e->ty += -ty_ocn; e->n += 1;
  }
}
int orc_cell_count(const orc_world *w, int xidx, int yidx) { return w->cells[(size_t)(xidx - 1) * (size_t)(w->Ny + 1) + (size_t)(yidx - 1)].n; }
/* entry k of centre cell (xidx, yidx): out = floeidx, dx, dy, tx, ty, npoints */
void orc_cell_entry(const orc_world *w, int xidx, int yidx, int k, double *out6) {
  const cell_ent *e = &w->cells[(size_t)(xidx - 1) * (size_t)(w->Ny + 1) + (size_t)(yidx - 1)].e[k];
  out6[0] = e->floe; out6[1] = e->dx; out6[2] = e->dy; out6[3] = e->tx; out6[4] = e->ty; out6[5] = e->n;
}
/* calc_two_way_coupling!, coupling.jl:1617-1680 */
void orc_calc_two_way_coupling(orc_world *w) {
  int per_x = w->b[ORC_EAST].kind == ORC_PERIODIC, per_y = w->b[ORC_NORTH].kind == ORC_PERIODIC;
  double cell_area = w->gdx * w->gdy;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(w->nthreads)
#endif
  for (int ix = 1; ix <= w->Nx + 1; ix++) {
    orc_ring cell, moved; orc_regions rg;
    orc_ring_init(&cell); orc_ring_init(&moved); orc_regions_init(&rg);
    for (int iy = 1; iy <= w->Ny + 1; iy++) {
      size_t q = (size_t)(ix - 1) * (size_t)(w->Ny + 1) + (size_t)(iy - 1);
      double tx = 0.0, ty = 0.0, si = 0.0;
      const struct cell_list *c = &w->cells[q];
      if (c->n > 0) {
        double bb[4]; orc_center_cell_coords(w, ix, iy, per_y, per_x, bb);
        /* _make_bounding_box_polygon: (xmin,ymin) (xmin,ymax) (xmax,ymax) (xmax,ymin) (xmin,ymin) */
        cell.n = 0;
        orc_ring_push(&cell, bb[0], bb[2]); orc_ring_push(&cell, bb[0], bb[3]); orc_ring_push(&cell, bb[1], bb[3]);
        orc_ring_push(&cell, bb[1], bb[2]); orc_ring_push(&cell, bb[0], bb[2]);
        for (int k = 0; k < c->n; k++) {
          const cell_ent *e = &c->e[k];
          const floe_t *f = &w->f[e->floe];
          moved.n = 0;
          for (int v = 0; v < f->poly.n; v++) orc_ring_push(&moved, f->poly.p[v].x + e->dx, f->poly.p[v].y + e->dy);
          orc_intersection(&cell, &moved, &rg);
          double a = 0.0;
          for (int r = 0; r < rg.n; r++) a += orc_area(&rg.r[r]);
          if (a > 0) {
            tx += (e->tx / e->n) * a; ty += (e->ty / e->n) * a; si += a;
          }
        }
        if (si > 0) { tx /= si; ty /= si; si /= cell_area; }
      }
      double du = w->ua[q] - w->uo[q], dv = w->va[q] - w->vo[q];
      double ocn_frac = 1 - si;
      double nrm = sqrt(du * du + dv * dv);
      tx += w->rho_a * w->Cd_ao * ocn_frac * nrm * du;
      ty += w->rho_a * w->Cd_ao * ocn_frac * nrm * dv;
      w->tau_x[q] = tx; w->tau_y[q] = ty; w->si_frac[q] = si;
      w->hf[q] = w->dt_couple * w->k_ice / (w->rho_i * w->L_ice) * (w->t_ocn[q] - w->t_atm[q]);
    }
    orc_ring_free(&cell); orc_ring_free(&moved); orc_regions_free(&rg);
  }
}

/* ------------------------------------------------------------------ output path */
/* calc_eulerian_data!, output.jl:793-914: floe data averaged on the writer's grid (lines xg: nx+1, yg: ny+1).
   data[k][ix][iy] at (k * nx + ix) * ny + iy, k in the order of ORC_EUL_* (orc.h); cell (ix, iy) is
   [xg[ix], xg[ix+1]] x [yg[iy], yg[iy+1]] (writer.data[j, i, k] with j = ix + 1, i = iy + 1).
   Topography (the cell polygon minus the topography, :829-832) is restated through the set identity
   area(floe ∩ (cell \ topo)) = area(floe ∩ cell) - sum_t area((floe ∩ cell) ∩ topo_t), valid for topography
   elements that do not overlap one another (the reference's diff_polys lives in GeometryOps). */
static double regions_area(const orc_regions *rg) {
  double a = 0.0;
  for (int r = 0; r < rg->n; r++) a += orc_area(&rg->r[r]);
  return a;
}
void orc_calc_eulerian_data(const orc_world *w, int nx, int ny, const double *xg, const double *yg, double *data) {
  const double dx = xg[1] - xg[0], dy = yg[1] - yg[0];
  const double cell_rmax = sqrt(dx * dx + dy * dy);
  const int M = w->M;
  int *idx = (int *)malloc((size_t)(M > 0 ? M : 1) * sizeof(int));
  double *pic = (double *)malloc((size_t)(M > 0 ? M : 1) * sizeof(double));
  orc_ring cell; orc_regions rg, rg2;
  orc_ring_init(&cell); orc_regions_init(&rg); orc_regions_init(&rg2);
  for (int ix = 0; ix < nx; ix++) for (int iy = 0; iy < ny; iy++) {
    double *out[ORC_EUL_COUNT];
    for (int k = 0; k < ORC_EUL_COUNT; k++) { out[k] = &data[((size_t)k * nx + ix) * ny + iy]; *out[k] = 0.0; }
    /* potential interactions (:808-819): cell centre within rmax + cell_rmax of the centroid */
    const double xc = xg[ix] + 0.5 * dx, yc = yg[iy] + 0.5 * dy;
    int n = 0;
    for (int i = 0; i < M; i++) {
      const floe_t *f = &w->f[i];
      double d = sqrt((xc - f->cx) * (xc - f->cx) + (yc - f->cy) * (yc - f->cy)) - (f->rmax + cell_rmax);
      if (d < 0) idx[n++] = i;
    }
    if (n == 0) continue;
    cell.n = 0;
    orc_ring_push(&cell, xg[ix], yg[iy]); orc_ring_push(&cell, xg[ix], yg[iy + 1]); orc_ring_push(&cell, xg[ix + 1], yg[iy + 1]);
    orc_ring_push(&cell, xg[ix + 1], yg[iy]); orc_ring_push(&cell, xg[ix], yg[iy]);
    double cell_area = orc_area(&cell);
    for (int t = 0; t < w->ntopo; t++) {
      orc_intersection(&cell, &w->topo[t].poly, &rg);
      cell_area -= regions_area(&rg);
    }
    if (!(cell_area > 0)) continue;          /* length(cell_poly_list) == 0, :834-837 */
    int m = 0;
    for (int k = 0; k < n; k++) {
      const floe_t *f = &w->f[idx[k]];
      orc_intersection(&f->poly, &cell, &rg);
      double a = regions_area(&rg);
      for (int t = 0; t < w->ntopo && a > 0; t++)
        for (int r = 0; r < rg.n; r++) { orc_intersection(&rg.r[r], &w->topo[t].poly, &rg2); a -= regions_area(&rg2); }
      if (a > 0) { idx[m] = idx[k]; pic[m] = a; m++; }
    }
    double area_tot = 0.0, mass_tot = 0.0;
    for (int k = 0; k < m; k++) { const floe_t *f = &w->f[idx[k]]; area_tot += pic[k]; mass_tot += f->mass * (pic[k] / f->area); }
    if (!(mass_tot > 0)) continue;
    double s[ORC_EUL_COUNT] = { 0 }, over = 0.0;
    for (int k = 0; k < m; k++) {
      const floe_t *f = &w->f[idx[k]];
      const double r = (pic[k] / f->area) * (f->mass / mass_tot);     /* ma_ratios */
      s[ORC_EUL_U] += f->u * r; s[ORC_EUL_V] += f->v * r; s[ORC_EUL_DUDT] += f->p_dudt * r; s[ORC_EUL_DVDT] += f->p_dvdt * r;
      s[ORC_EUL_HEIGHT] += f->height * r;
      s[ORC_EUL_STRESS_XX] += f->sa[0] * r; s[ORC_EUL_STRESS_YX] += f->sa[1] * r;
      s[ORC_EUL_STRESS_XY] += f->sa[2] * r; s[ORC_EUL_STRESS_YY] += f->sa[3] * r;
      s[ORC_EUL_STRAIN_UX] += f->strain[0] * r; s[ORC_EUL_STRAIN_VX] += f->strain[1] * r;
      s[ORC_EUL_STRAIN_UY] += f->strain[2] * r; s[ORC_EUL_STRAIN_VY] += f->strain[3] * r;
      over += f->overarea;
    }
    s[ORC_EUL_OVERAREA] = over / m; s[ORC_EUL_MASS] = mass_tot; s[ORC_EUL_AREA] = area_tot;
    s[ORC_EUL_SI_FRAC] = area_tot / cell_area;
    {   /* maximum(eigvals([xx yx; xy yy])), zeroed beyond 1e8 (:882-891) */
      double xx = s[ORC_EUL_STRESS_XX], yx = s[ORC_EUL_STRESS_YX], xy = s[ORC_EUL_STRESS_XY], yy = s[ORC_EUL_STRESS_YY];
      double hm = 0.5 * (xx + yy), hd = 0.5 * (xx - yy);
      double e = hm + sqrt(hd * hd + xy * yx);
      if (fabs(e) > 1e8) e = 0.0;
      s[ORC_EUL_STRESS_EIG] = e;
    }
    for (int k = 0; k < ORC_EUL_COUNT; k++) *out[k] = s[k];
  }
  orc_ring_free(&cell); orc_regions_free(&rg); orc_regions_free(&rg2); free(idx); free(pic);
}
/* what simplify_floes! (simplification.jl:339-378) would have to do: out4 = floes tagged remove, floes tagged
   fuse, floes with more than max_vertices ring points (smooth_floes!, :66: GI.npoint counts the closing point),
   floes not tagged remove under min_floe_area / min_floe_height (remove_floes!, :287-290) */
void orc_simplify_check(const orc_world *w, int max_vertices, double min_floe_area, double min_floe_height, int64_t *out4) {
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  for (int i = 0; i < w->M; i++) {
    const floe_t *f = &w->f[i];
    if (f->status == ORC_REMOVE) out4[0]++;
    if (f->status == ORC_FUSE) out4[1]++;
    if (f->poly.n > max_vertices) out4[2]++;
    if (f->status != ORC_REMOVE && (f->area < min_floe_area || f->height < min_floe_height)) out4[3]++;
  }
}

/* calc_one_way_coupling!, coupling.jl:1486-1589 (with calc_subfloe_values! :627-657,
   in_bounds :494-597, calc_atmosphere_forcing :1212-1232, calc_ocean_forcing! :1277-1299) */
void orc_timestep_coupling(orc_world *w) {
  int per_x = w->b[ORC_EAST].kind == ORC_PERIODIC;
  int per_y = w->b[ORC_NORTH].kind == ORC_PERIODIC;
  if (w->two_way) orc_clear_cells(w);            /* empty!.(grid.floe_locations), empty!.(ocean.scells): :1721-1724 */
  for (int i = 0; i < w->M; i++) {
    floe_t *f = &w->f[i];
    double ca = cos(f->alpha), sa = sin(f->alpha);
    double ma_ratio = f->mass / f->area;
    double xcor = (f->mass / f->area) * w->fcor * f->v;
    double ycor = (f->mass / f->area) * w->fcor * f->u;
    double tot_x = 0, tot_y = 0, tot_trq = 0, tot_h = 0; int npoints = 0;
    /* first pass counts in-bounds points (npoints multiplies the coriolis term before the loop) */
    for (int k = 0; k < f->nsub; k++) {
      double x = (ca * f->sx[k] - sa * f->sy[k]) + f->cx;
      double y = (sa * f->sx[k] + ca * f->sy[k]) + f->cy;
      int inb = orc_in_bounds(w, x, y, per_x, per_y);
      if (inb) npoints++;
    }
    if (npoints == 0) { f->status = ORC_REMOVE; continue; }
    tot_x = npoints * xcor; tot_y = -npoints * ycor;
    for (int k = 0; k < f->nsub; k++) {
      double x = (ca * f->sx[k] - sa * f->sy[k]) + f->cx;
      double y = (sa * f->sx[k] + ca * f->sy[k]) + f->cy;
      int inb = orc_in_bounds(w, x, y, per_x, per_y);
      if (!inb) continue;
      double xc = x - f->cx, yc = y - f->cy;
      double th = atan2(yc, xc), rad = sqrt(xc * xc + yc * yc);
      double st = sin(th), ct = cos(th);
      double up = f->u - f->xi * rad * st, vp = f->v + f->xi * rad * ct;
      double uatm = sample(w, w->ua, x, y, per_x, per_y), vatm = sample(w, w->va, x, y, per_x, per_y);
      double du = uatm - up, dv = vatm - vp;
      double nrm = sqrt(du * du + dv * dv);
      double tax = w->rho_a * w->Cd_ia * nrm * du, tay = w->rho_a * w->Cd_ia * nrm * dv;
      double uocn = sample(w, w->uo, x, y, per_x, per_y), vocn = sample(w, w->vo, x, y, per_x, per_y);
      double hfl = sample(w, w->hf, x, y, per_x, per_y);
      double duo = uocn - up, dvo = vocn - vp;
      double nrmo = sqrt(duo * duo + dvo * dvo);
      double tox = w->rho_o * w->Cd_io * nrmo * (cos(w->turn) * duo - sin(w->turn) * dvo);
      double toy = w->rho_o * w->Cd_io * nrmo * (sin(w->turn) * duo + cos(w->turn) * dvo);
      double tpx = -ma_ratio * w->fcor * vocn, tpy = ma_ratio * w->fcor * uocn;
      double tx = tax + tpx + tox, ty = tay + tpy + toy;
      double trq = (-tx * st + ty * ct) * rad;
      tot_x += tx; tot_y += ty; tot_trq += trq; tot_h += hfl;
      if (w->two_way) {
        /* find_center_cell_index, coupling.jl:466-470 (1-based), then floe_to_grid_info! :1417-1454 */
        int xidx = (int)floor((x - w->gx0) / w->gdx + 0.5) + 1, yidx = (int)floor((y - w->gy0) / w->gdy + 0.5) + 1;
        orc_floe_to_grid_info(w, i, xidx, yidx, tox, toy);
      }
    }
    f->fxOA = tot_x / npoints * f->area;
    f->fyOA = tot_y / npoints * f->area;
    f->trqOA = tot_trq / npoints * f->area;
    f->hflx = tot_h / npoints;
  }
  if (w->two_way) orc_calc_two_way_coupling(w);
}

/* ------------------------------------------------------------------ rigid-body update */
/* calc_stress!, update_floe.jl:392-414 + _update_stress_accum!, stress_calculators.jl:118-122 */
static void calc_stress(orc_world *w, floe_t *f) {
  double s11 = 0, s12 = 0, s21 = 0, s22 = 0;
  if (f->ninter > 0) {
    for (int k = 0; k < f->ninter; k++) {
      double *r = f->inter + (size_t)k * I_NCOL;
      s11 += (r[I_PX] - f->cx) * r[I_FX];
      s12 += (r[I_PY] - f->cy) * r[I_FX] + (r[I_PX] - f->cx) * r[I_FY];
      s22 += (r[I_PY] - f->cy) * r[I_FY];
    }
    s12 *= 0.5; s21 = s12;
    double sc = 1 / (f->area * f->height);
    s11 *= sc; s12 *= sc; s21 *= sc; s22 *= sc;
  }
  double l = w->lambda, s[4] = { s11, s12, s21, s22 };
  for (int k = 0; k < 4; k++) f->sa[k] = (1 - l) * f->sa[k] + l * s[k];
  for (int k = 0; k < 4; k++) f->si[k] = s[k];
}
/* calc_strain!, update_floe.jl:425-453 (v1, v2 use floe.u: literal) */
static void calc_strain(floe_t *f) {
  double e11 = 0, e12 = 0, e22 = 0, x1 = 0, y1 = 0;
  for (int i = 0; i < f->poly.n; i++) {
    double x2 = f->poly.p[i].x + (-f->cx), y2 = f->poly.p[i].y + (-f->cy);
    if (i == 0) { x1 = x2; y1 = y2; continue; }
    double xd = x2 - x1, yd = y2 - y1;
    double rad1 = sqrt(x1 * x1 + y1 * y1), rad2 = sqrt(x2 * x2 + y2 * y2);
    double t1 = atan2(y1, x1), t2 = atan2(y2, x2);
    double u1 = f->u - f->xi * rad1 * sin(t1), u2 = f->u - f->xi * rad2 * sin(t2);
    double v1 = f->u + f->xi * rad1 * cos(t1), v2 = f->u + f->xi * rad2 * cos(t2);
    double ud = u2 - u1, vd = v2 - v1;
    e11 += ud * yd; e12 += ud * xd + vd * yd; e22 += vd * xd;
    x1 = x2; y1 = y2;
  }
  e12 *= 0.5;
  double d = 2 * f->area;
  f->strain[0] = e11 / d; f->strain[1] = e12 / d; f->strain[2] = e12 / d; f->strain[3] = e22 / d;
}
static double sgn(double x) { return (x > 0) - (x < 0); }

/* the reference's own tests call calc_stress! / calc_strain! on floes whose interaction matrix was set by hand
   (test_update_floe.jl:10-39): the three entry points below are what those tests need */
void orc_set_interactions(orc_world *w, int i, int k, const double *rows) {
  floe_t *f = &w->f[i];
  if (k > f->capinter) { f->capinter = k; f->inter = (double *)realloc(f->inter, (size_t)k * I_NCOL * sizeof(double)); }
  memcpy(f->inter, rows, (size_t)k * I_NCOL * sizeof(double));
  f->ninter = k;
}
void orc_calc_stress(orc_world *w, int i) { calc_stress(w, &w->f[i]); }
void orc_calc_strain(orc_world *w, int i) { calc_strain(&w->f[i]); }

/* timestep_floe_properties!, update_floe.jl:469-551 */
void orc_timestep_floe_properties(orc_world *w, int dt) {
  int64_t wh = 0, wf = 0, wv = 0, wx = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(w->nthreads) reduction(+ : wh, wf, wv, wx)
#endif
  for (int i = 0; i < w->M; i++) {
    floe_t *f = &w->f[i];
    double cfx = f->cfx, cfy = f->cfy, ctrq = f->ctrq;
    calc_stress(w, f);
    if (f->height > w->max_h) { f->height = w->max_h; wh++; }
    for (int it = 0; it < 400 && fmax(fabs(cfx), fabs(cfy)) > f->mass / (5 * dt); it++) { cfx = cfx / 10; cfy = cfy / 10; ctrq = ctrq / 10; wf++; }
    double h = f->height;
    double dh = f->hflx / h;
    double hfrac = (h + dh) / h;
    f->mass *= hfrac; f->moment *= hfrac; f->height -= dh; h = f->height;
    double dx = 1.5 * dt * f->u - 0.5 * dt * f->p_dxdt;
    double dy = 1.5 * dt * f->v - 0.5 * dt * f->p_dydt;
    double da = 1.5 * dt * f->xi - 0.5 * dt * f->p_dalphadt;
    f->alpha += da;
    /* _move_floe!, floe_utils.jl:82-93: rotate about the centroid, then translate */
    {
      double cx = f->cx, cy = f->cy, c = cos(da), s = sin(da);
      for (int k = 0; k < f->poly.n; k++) {
        double x = f->poly.p[k].x + (-cx), y = f->poly.p[k].y + (-cy);
        double xr = c * x - s * y, yr = s * x + c * y;
        f->poly.p[k].x = xr + (cx + dx); f->poly.p[k].y = yr + (cy + dy);
      }
      f->cx += dx; f->cy += dy;
    }
    f->p_dxdt = f->u; f->p_dydt = f->v; f->p_dalphadt = f->xi;
    double dudt = (f->fxOA + cfx) / f->mass, dvdt = (f->fyOA + cfy) / f->mass;
    double frac = 1.0, au = fabs(dt * dudt), av = fabs(dt * dvdt), h2 = h / 2;
    if (au > h2 && av > h2) {
      double f1 = (sgn(dudt) * h / (2 * dt)) / dudt, f2 = (sgn(dvdt) * h / (2 * dt)) / dvdt;
      frac = f1 < f2 ? f1 : f2;
    } else if (au > h2 && av < h2) frac = (sgn(dudt) * h / (2 * dt)) / dudt;
    else if (au < h2 && av > h2) frac = (sgn(dvdt) * h / (2 * dt)) / dvdt;
    if (frac != 1) { dudt = frac * dudt; dvdt = frac * dvdt; wv++; }
    f->u += 1.5 * dt * dudt - 0.5 * dt * f->p_dudt;
    f->v += 1.5 * dt * dvdt - 0.5 * dt * f->p_dvdt;
    f->p_dudt = dudt; f->p_dvdt = dvdt;
    double dxidt = (f->trqOA + ctrq) / f->moment;
    dxidt = frac * dxidt;
    double xi = f->xi + 1.5 * dt * dxidt - 0.5 * dt * f->p_dxidt;
    if (fabs(xi) > w->max_xi) { xi = sgn(xi) * w->max_xi; wx++; }
    f->xi = xi; f->p_dxidt = dxidt;
    calc_strain(f);
  }
  w->warn[0] = wh; w->warn[1] = wf; w->warn[2] = wv; w->warn[3] = wx;
}

/* timestep_sim!, simulation.jl:94-170, hot-path processes only */
void orc_timestep_sim(orc_world *w, int tstep, int dt, int coupling_dt, int collisions_on, int coupling_on) {
  if (w->M == 0) return;
  int n_init = w->M;
  double t0 = wall_now(), t1;
  orc_add_ghosts(w);
  t1 = wall_now(); w->tphase[0] += t1 - t0; t0 = t1;
  if (collisions_on) orc_timestep_collisions(w, n_init, dt);
  t0 = wall_now();
  orc_remove_ghosts(w, n_init);
  t1 = wall_now(); w->tphase[0] += t1 - t0; t0 = t1;
  if (coupling_on && coupling_dt > 0 && (tstep % coupling_dt) == 0) orc_timestep_coupling(w);
  t1 = wall_now(); w->tphase[5] += t1 - t0; t0 = t1;
  orc_timestep_floe_properties(w, dt);
  w->tphase[6] += wall_now() - t0;
}
