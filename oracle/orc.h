/*
 * orc.h — CPU ORACLE for the Subzero.jl per-timestep collision / forcing / rigid-body path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C (C11, fp64) restatement of the reference's
 * algorithm, used as the parity checker by tests/, by __graft_entry__.smoke() and by the
 * `cpu_baseline` leg of bench.py.  Nothing under subzero.jl_amd/ (the product) may include,
 * link, import or execute anything in this directory.
 *
 * Parity pinning:
 *   - collisions (floe-floe, floe-boundary, topography, ghosts): PINNED by the literal-input
 *     known-answer tests of the reference, transcribed into tests/golden/collisions.json
 *     (reference: test/test_physical_processes/test_collisions.jl:43-362).
 *   - moment of inertia: PINNED (test/test_floe_utils.jl:66-71).
 *   - ocean/atmosphere forcings: PINNED by test/test_physical_processes/test_coupling.jl:464-639
 *     with the sub-floe points decoded from test/inputs/test_mc_points.jld2
 *     (tests/golden/make_golden.py).
 *   - calc_stress! / calc_strain!: PINNED by test/test_physical_processes/test_update_floe.jl:10-39 with
 *     the two floes of test/inputs/stress_strain.jld2 decoded into tests/golden/update_floe.json.
 *   - two-way coupling bookkeeping (find_center_cell_index, center_cell_coords, shift_cell_idx,
 *     floe_to_grid_info!): PINNED by test/test_physical_processes/test_coupling.jl:165-180, 276-460
 *     (tests/golden/coupling_grid.json); calc_two_way_coupling! itself has no reference fixture and is
 *     checked on an analytic case.
 *   - calc_eulerian_data! (output.jl:793-914, the GridOutputWriter averages): no numeric fixture in the reference
 *     (test/test_output.jl checks only names and shapes) => "parity unpinned"; checked on analytic cases.
 *   - the rest of timestep_floe_properties! (guards, thermodynamics, AB2 update): formula-level
 *     restatement, no reference fixture exists => "parity unpinned" for those lines beyond the
 *     conservation properties checked in tests/.
 *
 * The polygon arithmetic of the reference lives in GeometryOps.jl 0.1.x (Project.toml:38),
 * which is NOT under /root/reference.  orc_geom.c restates its published algorithms
 * (Greiner-Hormann / Foster-Hormann boundary tracing, shoelace area, area-weighted centroid,
 * point-segment distance, Hao-Sun point-in-polygon) and is anchored on the reference's call
 * sites (src/floe_utils.jl:55, src/physical_processes/collisions.jl:64,91,99,156,178,360).
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ geometry */
typedef struct { double x, y; } orc_pt;

/* closed ring: p[n-1] == p[0]; n = GI.npoint(ring) */
typedef struct { int n; int cap; orc_pt *p; } orc_ring;

typedef struct { int n; int cap; orc_ring *r; } orc_regions;

void   orc_ring_init(orc_ring *r);
void   orc_ring_free(orc_ring *r);
void   orc_ring_push(orc_ring *r, double x, double y);
void   orc_ring_copy(orc_ring *dst, const orc_ring *src);
void   orc_ring_from_xy(orc_ring *r, int n, const double *x, const double *y);
void   orc_regions_init(orc_regions *rg);
void   orc_regions_free(orc_regions *rg);

double orc_signed_area(const orc_ring *r);             /* GO._signed_area */
double orc_area(const orc_ring *r);                    /* GO.area (unsigned) */
void   orc_centroid(const orc_ring *r, double *cx, double *cy); /* GO.centroid */
int    orc_coveredby(double x, double y, const orc_ring *r);    /* GO.coveredby(point, poly) */
double orc_dist_to_ring(double x, double y, const orc_ring *r); /* |GO.signed_distance| */
int    orc_intersects(const orc_ring *a, const orc_ring *b);    /* GO.intersects(poly, poly) */
void   orc_intersection(const orc_ring *a, const orc_ring *b, orc_regions *out); /* intersect_polys */
/* GO.intersection_points: returns count, fills pts (malloc'd, caller frees) */
int    orc_intersection_points(const orc_ring *a, const orc_ring *b, orc_pt **pts);

long   orc_trace_failures(void);   /* diagnostics */

/* ctypes-friendly wrappers on flat arrays (tests call these) */
int    orc_clip_flat(int na, const double *ax, const double *ay,
                     int nb, const double *bx, const double *by,
                     int max_regions, int max_pts,
                     int *reg_off /* max_regions+1 */, double *rx, double *ry);
int    orc_ipoints_flat(int na, const double *ax, const double *ay,
                        int nb, const double *bx, const double *by,
                        int max_pts, double *px, double *py);

/* ------------------------------------------------------------------ world */
typedef struct orc_world orc_world;

enum { ORC_OPEN = 0, ORC_PERIODIC = 1, ORC_COLLISION = 2, ORC_MOVING = 3 };
enum { ORC_NORTH = 0, ORC_SOUTH = 1, ORC_EAST = 2, ORC_WEST = 3 };
enum { ORC_ACTIVE = 1, ORC_REMOVE = 2, ORC_FUSE = 3 };   /* floe.jl:8-12 */

/* per-floe scalar field ids for orc_get_field / orc_set_field */
enum {
  ORC_F_CX = 0, ORC_F_CY, ORC_F_RMAX, ORC_F_AREA, ORC_F_HEIGHT, ORC_F_MASS, ORC_F_MOMENT,
  ORC_F_ALPHA, ORC_F_U, ORC_F_V, ORC_F_XI,
  ORC_F_P_DXDT, ORC_F_P_DYDT, ORC_F_P_DALPHADT, ORC_F_P_DUDT, ORC_F_P_DVDT, ORC_F_P_DXIDT,
  ORC_F_FXOA, ORC_F_FYOA, ORC_F_TRQOA, ORC_F_HFLX, ORC_F_OVERAREA,
  ORC_F_COLL_FX, ORC_F_COLL_FY, ORC_F_COLL_TRQ,
  ORC_F_SA11, ORC_F_SA12, ORC_F_SA21, ORC_F_SA22,     /* stress_accum   */
  ORC_F_SI11, ORC_F_SI12, ORC_F_SI21, ORC_F_SI22,     /* stress_instant */
  ORC_F_E11, ORC_F_E12, ORC_F_E21, ORC_F_E22,         /* strain         */
  ORC_F_COUNT
};

orc_world *orc_create(void);
void orc_destroy(orc_world *w);

/* simulation.jl:5-18 Constants; only the hot-path members */
void orc_set_consts(orc_world *w, double E, double nu, double mu, double rho_o, double rho_a,
                    double Cd_io, double Cd_ia, double f, double turn_theta);
/* process_settings.jl:183-187, :25-32, stress_calculators.jl:82, process_settings.jl:133-137 */
void orc_set_settings(orc_world *w, double floe_floe_max_overlap, double floe_domain_max_overlap,
                      double rho_i, double max_floe_height, double max_xi, double lambda,
                      int coupling_dd);
/* kinds[4] in order N,S,E,W; polygons/vals from _boundary_info_from_extent (boundaries.jl:29-150) */
void orc_set_domain_extent(orc_world *w, const int *kinds, double x0, double xf, double y0,
                           double yf, const double *bu, const double *bv);
/* topography CSR; centroid+rmax computed like TopographyElement (topography.jl:66-72) */
void orc_set_topography(orc_world *w, int ntopo, const int *off, const double *x, const double *y);
/* grids.jl:106, oceans.jl:74, atmos.jl:4: fields are (Nx+1) x (Ny+1), element [ix][iy] at ix*(Ny+1)+iy */
void orc_set_grid_fields(orc_world *w, int Nx, int Ny, double x0, double xf, double y0, double yf,
                         const double *uo, const double *vo, const double *hflx,
                         const double *ua, const double *va);

/* Floe(coords, hmean, 0) (floe.jl:144-242): derives centroid, area, mass, moment, rmax. Returns index. */
int  orc_add_floe(orc_world *w, int n, const double *x, const double *y, double height);
void orc_set_subpoints(orc_world *w, int i, int n, const double *sx, const double *sy);
int  orc_num_floes(const orc_world *w);
void orc_get_field(const orc_world *w, int field, double *out);
void orc_set_field(orc_world *w, int field, const double *in);
void orc_get_ids(const orc_world *w, int64_t *id, int64_t *ghost_id, int32_t *status);
void orc_set_ids(orc_world *w, const int64_t *id);
void orc_set_status(orc_world *w, const int32_t *status);
int  orc_total_ring_points(const orc_world *w);
void orc_get_rings(const orc_world *w, int32_t *off /*M+1*/, double *x, double *y);
int  orc_total_interactions(const orc_world *w);
void orc_get_interactions(const orc_world *w, int32_t *off /*M+1*/, double *rows /* k x 7 row-major */);
int  orc_total_ghost_links(const orc_world *w);
void orc_get_ghosts(const orc_world *w, int32_t *off /*M+1*/, int32_t *idx /* 0-based */);
int  orc_total_fuse(const orc_world *w);
void orc_get_fuse(const orc_world *w, int32_t *off /*M+1*/, int32_t *idx /* 0-based */);
void orc_get_boundary_vals(const orc_world *w, double *vals4);
void orc_get_boundary_polys(const orc_world *w, double *xy40);
int orc_which_vertices_match_points(int npts, const double *px, const double *py, int nr, const double *rx, const double *ry, int32_t *idx);

/* the reference's process API */
void orc_add_ghosts(orc_world *w);                                   /* collisions.jl:1060-1174 */
void orc_remove_ghosts(orc_world *w, int n_init);                    /* simulation.jl:138-144 */
void orc_timestep_collisions(orc_world *w, int n_init, int dt);      /* collisions.jl:734-864 */
void orc_floe_floe_interaction(orc_world *w, int i, int j, int dt, double max_overlap); /* :347 */
void orc_floe_domain_interaction(orc_world *w, int i, int dt, double max_overlap);      /* :594 */
void orc_calc_torque(orc_world *w, int i);                           /* collisions.jl:673-686 */
void orc_timestep_coupling(orc_world *w);                            /* coupling.jl:1705-1738 */
int  orc_in_bounds(const orc_world *w, double x, double y, int per_x, int per_y);      /* coupling.jl:494-597 */
int  orc_find_interp_knots(int npts, const int *point_idx, int ncells, double g0, double dg, double L, int dd, int periodic,
                           int cap, double *knots, int *knot_idx);                      /* coupling.jl:702-797 */
void orc_sample_lines(const orc_world *w, double x, double y, int per_x, int per_y, int *lines4, double *t2);
void orc_sample_fields(const orc_world *w, double x, double y, int per_x, int per_y, double *out5);   /* coupling.jl:845-902 */
/* two-way coupling (off by default): ice-on-ocean stress per centre cell, coupling.jl:1617-1680 */
void orc_set_two_way(orc_world *w, int on, double Cd_ao, double k, double L, int dt);
void orc_set_temps(orc_world *w, const double *t_ocn, const double *t_atm);       /* (Nx+1) x (Ny+1), [ix][iy] */
void orc_get_ocean_stress(const orc_world *w, double *tau_x, double *tau_y, double *si_frac, double *hflx);
void orc_clear_cells(orc_world *w);
int  orc_shift_cell_idx(int idx, int nlines, int periodic);              /* coupling.jl:1154-1178 */
void orc_center_cell_coords(const orc_world *w, int xidx, int yidx, int ns_periodic, int ew_periodic, double *out4); /* :1116 */
void orc_floe_to_grid_info(orc_world *w, int floeidx, int xidx, int yidx, double tx_ocn, double ty_ocn); /* :1417 */
int  orc_cell_count(const orc_world *w, int xidx, int yidx);
void orc_cell_entry(const orc_world *w, int xidx, int yidx, int k, double *out6);
void orc_calc_two_way_coupling(orc_world *w);                           /* coupling.jl:1617-1680 */
void orc_timestep_floe_properties(orc_world *w, int dt);             /* update_floe.jl:469-551 */
void orc_set_interactions(orc_world *w, int i, int k, const double *rows);   /* floe.interactions = k x 7 matrix, row-major */
void orc_calc_stress(orc_world *w, int i);                            /* calc_stress!, update_floe.jl:392-414 */
void orc_calc_strain(orc_world *w, int i);                            /* calc_strain!, update_floe.jl:425-453 */
/* ---- output path */
/* grid outputs of calc_eulerian_data! (output.jl:855-905) */
enum {
  ORC_EUL_U = 0, ORC_EUL_V, ORC_EUL_DUDT, ORC_EUL_DVDT, ORC_EUL_OVERAREA, ORC_EUL_MASS, ORC_EUL_AREA, ORC_EUL_HEIGHT,
  ORC_EUL_SI_FRAC, ORC_EUL_STRESS_XX, ORC_EUL_STRESS_YX, ORC_EUL_STRESS_XY, ORC_EUL_STRESS_YY, ORC_EUL_STRESS_EIG,
  ORC_EUL_STRAIN_UX, ORC_EUL_STRAIN_VX, ORC_EUL_STRAIN_UY, ORC_EUL_STRAIN_VY, ORC_EUL_COUNT
};
void orc_calc_eulerian_data(const orc_world *w, int nx, int ny, const double *xg, const double *yg, double *data); /* output.jl:793-914 */
void orc_simplify_check(const orc_world *w, int max_vertices, double min_floe_area, double min_floe_height, int64_t *out4); /* simplification.jl:66,287-290 */
/* timestep_sim! (simulation.jl:94-170) restricted to the hot path */
void orc_timestep_sim(orc_world *w, int tstep, int dt, int coupling_dt, int collisions_on, int coupling_on);

/* the broad-phase result of the last orc_timestep_collisions call: the pairs that reached
   floe_floe_interaction! (collisions.jl:776), in serial (i asc, j asc) order, 0-based */
int  orc_num_pairs(const orc_world *w);
void orc_get_pairs(const orc_world *w, int32_t *pi, int32_t *pj);
/* warn counters of the last orc_timestep_floe_properties call: height clamps, force
   down-scalings, velocity fracs, xi clamps (update_floe.jl:482-491,516-531,540-543) */
void orc_get_warn_counts(const orc_world *w, int64_t *out4);
void orc_set_threads(orc_world *w, int nthreads);
/* wall seconds per phase of the steps run since the last reset (8 doubles: add_ghosts!, pair loop, Dict pass, interactions,
   mirror + ghost fold + totals, timestep_coupling!, timestep_floe_properties!, -): where the CPU path spends its time */
void orc_get_phase_times(const orc_world *w, double *out8);
void orc_reset_phase_times(orc_world *w);

#ifdef __cplusplus
}
#endif
#endif
