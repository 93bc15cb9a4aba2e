/*
 * subzero_hip.h — C-ABI of libsubzero_hip.so, the MI355X (gfx950) engine for Subzero.jl's
 * per-timestep collision / forcing / rigid-body-update path.
 *
 * This is the drop-in boundary: a Julia shim or any other host
 * binds exactly these symbols with @ccall / ctypes / cgo.  Plain pointers and sizes only; no
 * exceptions cross the boundary.  Every int-returning call returns 0 on success and a negative
 * SZ_E_* code on failure (text via sz_last_error).  All calls are synchronous at return and
 * must come from one host thread per context.  The caller owns every host buffer; the library
 * keeps no host pointer after a call returns.
 *
 * Reference interfaces replaced (paths relative to the Subzero.jl repository):
 *   sz_add_ghosts                 <- add_ghosts!(floes, domain)            src/physical_processes/collisions.jl:1060-1174
 *   sz_timestep_collisions        <- timestep_collisions!(floes, n_init_floes, domain, consts, Δt,
 *                                    collision_settings, spinlock)          collisions.jl:734-864
 *   sz_collide_pairs              <- floe_floe_interaction!(ifloe, i, jfloe, j, consts, Δt,
 *                                    max_overlap)                           collisions.jl:347-408
 *   sz_collide_domain             <- floe_domain_interaction!(floe, domain, consts, Δt,
 *                                    max_overlap)                           collisions.jl:594-662
 *   sz_remove_ghosts              <- ghost-row deletion in timestep_sim!    src/simulation_components/simulation.jl:138-144
 *   sz_timestep_coupling          <- timestep_coupling!(model, Δt, consts, coupling_settings,
 *                                    floe_settings), one-way part           src/physical_processes/coupling.jl:1705-1738
 *   sz_timestep_floe_properties   <- timestep_floe_properties!(floes, tstep, Δt, floe_settings)
 *                                                                           src/physical_processes/update_floe.jl:469-551
 *   sz_step                       <- timestep_sim!(sim, tstep), hot-path processes only,
 *                                    state resident in HBM between steps   simulation.jl:94-170
 *   sz_params                     <- Constants (simulation.jl:5-18), CollisionSettings
 *                                    (process_settings.jl:183-187), FloeSettings (:25-32),
 *                                    DecayAreaScaledCalculator.λ (stress_calculators.jl:82),
 *                                    CouplingSettings.Δd (process_settings.jl:133-137)
 *   sz_floe_columns               <- the hot columns of StructArray{Floe{Float64}} (src/simulation_components/floe.jl:24-77)
 *
 * Index conventions: all indices in this API are 0-based, EXCEPT the `floeidx` column (column 0)
 * of interaction rows, which holds the partner exactly as the reference stores it
 * (collisions.jl:297): 1-based floe index as a double, -1/-2/-3/-4 for the N/S/E/W boundary,
 * -(4+k) for topography element k (1-based).
 * Interaction rows are 7 doubles, row-major: floeidx, xforce, yforce, xpoint, ypoint, torque,
 * overlap (floe.jl:102-110).
 * 2x2 tensors are 4 doubles per floe in the order 11, 12, 21, 22.
 */
#ifndef SUBZERO_HIP_H
#define SUBZERO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sz_ctx sz_ctx;

enum { SZ_OK = 0, SZ_E_HIP = -1, SZ_E_ARG = -2, SZ_E_CAPACITY = -3, SZ_E_STATE = -4, SZ_E_NODEVICE = -5 };
enum { SZ_OPEN = 0, SZ_PERIODIC = 1, SZ_COLLISION = 2, SZ_MOVING = 3 };       /* boundary kinds  */
enum { SZ_NORTH = 0, SZ_SOUTH = 1, SZ_EAST = 2, SZ_WEST = 3 };                /* boundary order  */
enum { SZ_ACTIVE = 1, SZ_REMOVE = 2, SZ_FUSE = 3 };                           /* floe.jl:8-12    */
/* process switches for sz_step (CollisionSettings.collisions_on, CouplingSettings.coupling_on) */
enum { SZ_COLLISIONS_ON = 1, SZ_COUPLING_ON = 2,
       SZ_NO_STOP = 4 };   /* sz_step: run all nsteps even when a floe gets tagged remove / fuse (departs from the reference,
                              which runs simplify_floes! after every step: measurement and soak runs only) */

typedef struct {
  double E, nu, mu, rho_o, rho_a, Cd_io, Cd_ia, f, turn_theta;
  double floe_floe_max_overlap, floe_domain_max_overlap;
  double rho_i, max_floe_height, maximum_xi, lambda;
  int32_t coupling_dd;
  int32_t _pad;
} sz_params;

/* Host-side view of the floe columns.  Scalar columns have one entry per floe (M entries);
   ragged columns are CSR.  Rings are closed (last point == first).  NULL columns are read as
   zeros on upload and skipped on download. */
typedef struct {
  double *cx, *cy, *rmax, *area, *height, *mass, *moment, *alpha, *u, *v, *xi;
  double *p_dxdt, *p_dydt, *p_dalphadt, *p_dudt, *p_dvdt, *p_dxidt;
  double *fxOA, *fyOA, *trqOA, *hflx_factor, *overarea;
  double *coll_fx, *coll_fy, *coll_trq;
  double *stress_accum, *stress_instant, *strain;        /* 4 per floe */
  int64_t *id, *ghost_id;
  int32_t *status;
  int32_t *vert_off;  double *vx, *vy;                   /* M+1 offsets, closed rings   */
  int32_t *sub_off;   double *sx, *sy;                   /* sub-floe points, centred    */
  int32_t *ghost_off; int32_t *ghost_idx;                /* parent -> ghost rows        */
} sz_floe_columns;

typedef struct {
  int64_t M, N;                 /* floes incl. ghosts / parents                              */
  int64_t n_ring_points;        /* V_M                                                       */
  int64_t n_sub_points;         /* S                                                         */
  int64_t n_pairs;              /* P: pairs that reached the narrow phase in the last step   */
  int64_t n_pair_ring_points;   /* sum over the pairs the narrow phase ran of both rings' point counts */
  int64_t n_pair_rows;          /* C: floe-floe contact rows before mirroring                */
  int64_t n_elem_items;         /* floe-boundary / floe-topography clips                     */
  int64_t n_elem_rows;          /* C_b                                                       */
  int64_t n_inter_rows;         /* total interaction rows after mirror + ghost fold          */
  int64_t n_ghosts;             /* G                                                         */
  int64_t warn_height, warn_force, warn_vel, warn_xi;   /* update_floe.jl guards             */
  int64_t n_trace_fail;         /* clip traces abandoned (self-intersecting input / round-off), cumulative */
  int64_t n_halo;               /* halo floes received in the last tiled step */
  int64_t n_pairs_clipped;      /* pairs with overlapping ring boxes: the pair items the narrow phase ran */
  int64_t n_status_remove;      /* floes tagged remove / fuse: when both are zero the host-side simplify_floes!  */
  int64_t n_status_fuse;        /* (simulation.jl:206) has no removal or fusion to do and needs no download       */
  int64_t n_retry;              /* narrow-phase items redone by the largest kernel variant (working set overflow), cumulative */
  /* cumulative since the last sz_profile_reset: what the narrow phase did over a window of steps (bench.py prices the
     dominant kernel's algorithmic bytes with the counts of the very launches whose time it averages) */
  int64_t acc_narrow_launches;  /* launches of the first narrow variant (= collision steps)                    */
  int64_t acc_pair_items;       /* pair items run (pairs whose ring boxes overlap)                              */
  int64_t acc_pair_ring_points; /* sum over those items of both rings' point counts                              */
  int64_t acc_pair_rows;        /* floe-floe contact rows before mirroring                                       */
  int64_t acc_elem_items;       /* floe-boundary / floe-topography items run                                     */
  int64_t acc_elem_rows;
  int64_t acc_dir_checks;       /* direction checks of calc_normal_force (collisions.jl:58-68) run ...                          */
  int64_t acc_dir_checks_certified; /* ... of which settled from the crossing detection of the translated polygon alone (no second clip) */
} sz_stats;

/* kernel classes for sz_kernel_time_ms */
enum { SZ_K_GHOSTS = 0, SZ_K_BROAD = 1, SZ_K_NARROW = 2, SZ_K_REDUCE = 3, SZ_K_FORCING = 4,
       SZ_K_INTEGRATE = 5, SZ_K_COUNT = 6 };

/* ---- lifetime */
sz_ctx     *sz_create(int device_id);            /* NULL if no HIP device / allocation failure */
void        sz_destroy(sz_ctx *ctx);
const char *sz_last_error(const sz_ctx *ctx);
const char *sz_version(void);

/* ---- static inputs */
int sz_set_params(sz_ctx *ctx, const sz_params *p);
/* boundaries in the order N, S, E, W: kind, wall coordinate `val`, rectangle
   {xmin, xmax, ymin, ymax} (boundaries.jl:29-150) and velocity (MovingBoundary only) */
int sz_set_domain(sz_ctx *ctx, const int32_t *kinds, const double *vals, const double *rects,
                  const double *bu, const double *bv);
/* topography elements: closed rings (CSR) with centroid and rmax (topography.jl:5-9) */
int sz_set_topography(sz_ctx *ctx, int32_t ntopo, const int32_t *off, const double *x,
                      const double *y, const double *cx, const double *cy, const double *rmax);
/* grid (grids.jl:106) and the (Nx+1)x(Ny+1) ocean/atmosphere lattices, element [ix][iy] at
   ix*(Ny+1)+iy (oceans.jl:74, atmos.jl:4) */
int sz_set_fields(sz_ctx *ctx, int32_t Nx, int32_t Ny, double x0, double xf, double y0, double yf,
                  const double *uocn, const double *vocn, const double *hflx_factor,
                  const double *uatm, const double *vatm);

/* ---- floe state */
/* M rows, of which the first N are parents; M > N only when the caller ran add_ghosts!
   itself, in which case ghost_off/ghost_idx must be given.
   floe.interactions is ragged and not part of the column struct: the rows the last collision call left on the
   device stay valid across an upload of the same M (the shim's re-upload between timestep_collisions! and
   timestep_floe_properties!, whose calc_stress! reads them); an upload of another size drops them, and
   sz_timestep_floe_properties / sz_calc_stress then fail with SZ_E_STATE until sz_upload_interactions (the
   matrices the host holds, possibly empty) or a collision call provides rows again.  A context that is given
   a DIFFERENT field of the same size must be told so with sz_upload_interactions. */
int sz_upload_floes(sz_ctx *ctx, int64_t M, int64_t N, const sz_floe_columns *cols);
int sz_get_stats(sz_ctx *ctx, sz_stats *out);
/* copies every non-NULL column (sized from sz_get_stats) */
int sz_download_floes(sz_ctx *ctx, sz_floe_columns *cols);
/* interactions of all M floes: off has M+1 entries, rows has n_inter_rows*7 doubles */
int sz_download_interactions(sz_ctx *ctx, int32_t *off, double *rows);
/* ---- the same boundary for hosts whose floes are Floe{Float32} (floe.jl:24: the struct is generic in its float type FT; SURVEY section 8b
   "two instantiations: _f64, _f32").  documentation.md:25 supports and tests Float64 only, so there are no Float32 answers to reproduce: the
   engine computes in double (or mixed precision, sz_set_precision) whatever the host's FT -- these entry points widen the columns on the
   way in and round them on the way out (a run from Float32 columns therefore equals the run from the same values held as doubles, rounded
   once at the end).  Integer columns and the CSR offsets are the same types as above. */
typedef struct {
  float *cx, *cy, *rmax, *area, *height, *mass, *moment, *alpha, *u, *v, *xi;
  float *p_dxdt, *p_dydt, *p_dalphadt, *p_dudt, *p_dvdt, *p_dxidt;
  float *fxOA, *fyOA, *trqOA, *hflx_factor, *overarea;
  float *coll_fx, *coll_fy, *coll_trq;
  float *stress_accum, *stress_instant, *strain;         /* 4 per floe */
  int64_t *id, *ghost_id;
  int32_t *status;
  int32_t *vert_off;  float *vx, *vy;
  int32_t *sub_off;   float *sx, *sy;
  int32_t *ghost_off; int32_t *ghost_idx;
} sz_floe_columns_f32;
int sz_upload_floes_f32(sz_ctx *ctx, int64_t M, int64_t N, const sz_floe_columns_f32 *cols);
int sz_download_floes_f32(sz_ctx *ctx, sz_floe_columns_f32 *cols);
int sz_set_fields_f32(sz_ctx *ctx, int32_t Nx, int32_t Ny, double x0, double xf, double y0, double yf,
                      const float *uocn, const float *vocn, const float *hflx_factor, const float *uatm, const float *vatm);
int sz_download_interactions_f32(sz_ctx *ctx, int32_t *off, float *rows);
/* the pairs that reached the narrow phase, in the reference's serial (i asc, j asc) order */
int sz_download_pairs(sz_ctx *ctx, int32_t *pi, int32_t *pj);
/* status.fuse_idx after the mirror pass (collisions.jl:801-806): off M+1, idx; call with
   idx == NULL to get only the offsets (off[M] = total) */
int sz_download_fuse(sz_ctx *ctx, int32_t *off, int32_t *idx);
int sz_get_boundary_vals(sz_ctx *ctx, double *vals4);
/* the boundary rectangles as they stand, {xmin, xmax, ymin, ymax} for N, S, E, W (MovingBoundary walls move with
   update_boundaries!, collisions.jl:565-571) */
int sz_get_boundary_rects(sz_ctx *ctx, double *rects16);

/* ---- processes (each mirrors one reference function, see the header comment) */
int sz_add_ghosts(sz_ctx *ctx);
int sz_remove_ghosts(sz_ctx *ctx);
int sz_timestep_collisions(sz_ctx *ctx, int64_t n_init, int32_t dt);
/* floe_floe_interaction! on explicit (i, j) pairs: only floe i of each pair is updated */
int sz_collide_pairs(sz_ctx *ctx, int64_t npairs, const int32_t *pi, const int32_t *pj, int32_t dt,
                     double max_overlap);
/* floe_domain_interaction! for every floe */
int sz_collide_domain(sz_ctx *ctx, int32_t dt, double max_overlap);
int sz_timestep_coupling(sz_ctx *ctx);
int sz_timestep_floe_properties(sz_ctx *ctx, int32_t dt);
/* ---- precision (BASELINE configs[4]): 0 = fp64 (default); 1 = mixed.  Mixed: the per-point arithmetic of
   calc_one_way_coupling! in fp32 on fp32 copies of the sub-floe points and the lattice; the broad phase on 32-byte fp32
   records with fp64 confirmation of the bounding-circle test (pair list bit-exact); in resident batches of single-context
   runs the rings as fp32 offsets in the floe's body frame + fp64 pose, world coordinates rebuilt in fp64 for the
   narrow-phase predicates.  Absolute positions, predicates, contact rows and per-floe totals stay
   fp64.  The reference has no Float32 answers (documentation.md:25); the acceptance criterion against the fp64 path is
   stated in tests/test_hip_parity.py.  Two-way coupling keeps its fp64 forcing kernel. */
int sz_set_precision(sz_ctx *ctx, int32_t mode);
/* ---- two-way coupling: calc_two_way_coupling! (coupling.jl:1617-1680) with floe_to_grid_info! (:1417-1454),
   center_cell_coords (:1116-1140) and shift_cell_idx (:1154-1178).  Off by default like CouplingSettings().
   With it on, every coupling step (sz_timestep_coupling, sz_step) also produces the ice+atmosphere stress on the
   ocean, the sea-ice fraction and the heat-flux factor per centre cell ((Nx+1) x (Ny+1) values, element
   [ix][iy] at ix*(Ny+1)+iy); the heat-flux factor replaces the hflx lattice given to sz_set_fields, as
   ocean.hflx_factor is overwritten in the reference.  Cd_ao, k, L: Constants() (simulation.jl:10-14); dt is the
   timestep used for the heat-flux factor by sz_timestep_coupling (sz_step uses its own).
   Tiled (multi-GPU) runs: a floe only contributes on the rank that owns it, so after a tiled coupling step every
   rank calls sz_two_way_partial (its per-cell sums -> a device buffer of 3 (Nx+1)(Ny+1) doubles: stress numerators
   x, y and ice area), the host adds the buffers up across the ranks (all-reduce) and sz_two_way_finish turns the
   sums into the ocean fields on every rank (both ASYNC on the context's stream). */
int sz_set_two_way(sz_ctx *ctx, int32_t on, double Cd_ao, double k, double L, int32_t dt);
int sz_set_temps(sz_ctx *ctx, const double *t_ocn, const double *t_atm);
int sz_download_ocean_stress(sz_ctx *ctx, double *tau_x, double *tau_y, double *si_frac, double *hflx);
int sz_two_way_partial(sz_ctx *ctx, void *d_partial);
int sz_two_way_finish(sz_ctx *ctx, const void *d_partial, int32_t dt);
/* calc_stress! (update_floe.jl:392-414, with _update_stress_accum!, stress_calculators.jl:118-122) and
   calc_strain! (update_floe.jl:425-453) on their own, for every floe, as the reference's tests call them
   (test_update_floe.jl:10-39).  sz_upload_interactions replaces floe.interactions of every floe by hand-made
   matrices first (CSR offsets, rows of 7: floeidx, xforce, yforce, xpoint, ypoint, torque, overlap). */
int sz_upload_interactions(sz_ctx *ctx, const int32_t *inter_off, const double *rows);
int sz_calc_stress(sz_ctx *ctx);
int sz_calc_strain(sz_ctx *ctx);
/* nsteps x timestep_sim! with the state resident in HBM; tstep counts from tstep0.
   The reference runs simplify_floes! (host work: fuse, remove, smooth, simulation.jl:205-214) at the end of EVERY
   step.  The batch therefore ends after the first step that leaves a parent tagged remove or fuse: the launches of
   the remaining steps are already enqueued and return at once.  *steps_done (may be NULL) = steps actually run
   (== nsteps when nothing was tagged); the state is that of the reference after steps_done steps, status.fuse_idx
   (sz_download_fuse) included, and the host resumes with sz_step(nsteps - steps_done, tstep0 + steps_done, ...) after
   its simplify_floes! (and the upload that follows it).  Floes already tagged at entry end the batch after one
   step.  SZ_NO_STOP in flags runs all steps regardless. */
int sz_step(sz_ctx *ctx, int32_t nsteps, int32_t tstep0, int32_t dt, int32_t coupling_dt,
            int32_t flags, int32_t *steps_done);

/* ---- measurement: HIP-event time per kernel class, accumulated since the last reset, on the
   stream the kernels are launched on; launches = number of timed launches of that class.
   on = 0: off; 1: every class; otherwise a mask, bit (k+1) = class k (e.g. 2 << SZ_K_NARROW: only
   the dominant kernel -- every event pair costs a few microseconds of a ~250 us step) */
int sz_profile_enable(sz_ctx *ctx, int32_t on);
int sz_profile_reset(sz_ctx *ctx);
int sz_kernel_time_ms(sz_ctx *ctx, int32_t kclass, double *ms, int64_t *launches);
/* which launch evaluated the forcings (timestep_coupling!, coupling.jl:1705) in the steps of the last sz_step: 0 their own,
   1 the neighbour search's, 2 the narrow phase's first variant (class SZ_K_NARROW then times both: a small field's narrow
   phase is one round with a long tail, and the forcings run in that tail), -1 no coupling step yet */
int sz_forcing_launch(sz_ctx *ctx, int32_t *where);
/* name of the dominant kernel's instantiation as a kernel trace (rocprofv3) prints it, for the last batch: buf gets at most n bytes */
int sz_narrow_kernel_name(sz_ctx *ctx, char *buf, int32_t n);

/* ---- multi-GPU halo support (no counterpart in the single-process reference:
   its periodic ghost floes, collisions.jl:881-1047, are the same pattern inside one address
   space).  One context per rank owns a fixed subset of the floes; every step the ranks trade
   ghost-floe records (sz_halo_record_doubles() doubles each) through buffers in DEVICE memory
   that the host hands to RCCL (torch.distributed all_to_all_single).
     sz_tile_enable   after sz_upload_floes of the owned floes: gidx[i] = global index of owned
                      floe i; all order-dependent rules then use global indices.  max_ring = largest
                      ring (points) over the floes of ALL ranks, 0 if unknown; max_rmax = largest rmax
                      over the floes of ALL ranks (the resident steps' fixed broad-phase grid must hold
                      for halo floes too), 0 if unknown (the grid is then fitted every step)
     sz_owned_box     bounding box of the owned centroids + largest rmax: xmin,xmax,ymin,ymax,rmax
     sz_halo_set_boxes  nranks x {xmin,xmax,ymin,ymax}, already expanded by the interaction range
     sz_halo_pack     ASYNC. Exchange buffers have one region per peer: 1 header record (count in
                      double [0]) + cap record slots.  Fills d_send with the owned floes whose
                      centroid, or a periodic image of it, lies in the peer's box
     sz_tile_forcing  ASYNC, optional.  The forcings of step tstep (they need nothing from the halo); called between
                      sz_halo_pack and the collective they run beside the exchange, and sz_tile_step(tstep)
                      skips them
     sz_tile_step     ASYNC. Appends the records of d_recv (same layout, region r = from rank r) as
                      halo floes, runs one timestep_sim! on owned + halo floes; only owned floes are
                      integrated, the halo is dropped at the end.  d_recv may be NULL (no peers)
     sz_sync          waits for everything enqueued, reports sticky device errors
     sz_set_stream    enqueue on the caller's HIP stream (torch.cuda.current_stream().cuda_stream) so
                      that the framework's collectives order with the kernels without host syncs */
int sz_tile_enable(sz_ctx *ctx, const int64_t *gidx, double max_ring, double max_rmax);
int sz_owned_box(sz_ctx *ctx, double *out5);
int sz_halo_record_doubles(void);
/* ... of THIS context after sz_tile_enable: the records have room for the largest ring of all ranks' floes (12 + 2 x ring capacity, at
   least the value above; rings of up to 255 points, as in a single context -- Floe rings are unbounded, floe.jl:24-77) */
int sz_halo_record_doubles_ctx(sz_ctx *ctx);
int sz_halo_set_boxes(sz_ctx *ctx, int32_t nranks, const double *boxes);
int sz_halo_pack(sz_ctx *ctx, int32_t nranks, int32_t my_rank, double Lx, double Ly, int32_t periodic_x,
                 int32_t periodic_y, void *d_send, int32_t cap);
int sz_halo_counts(sz_ctx *ctx, int32_t nranks, int32_t *counts_out);   /* of the last pack; d_send NULL = count only */
int sz_tile_forcing(sz_ctx *ctx, int32_t tstep, int32_t coupling_dt, int32_t flags);
int sz_tile_step(sz_ctx *ctx, const void *d_recv, int32_t nranks, int32_t cap, int32_t tstep, int32_t dt,
                 int32_t coupling_dt, int32_t flags);
int sz_sync(sz_ctx *ctx);
int sz_set_stream(sz_ctx *ctx, void *hip_stream);

/* ---- the same halo exchange INSIDE the library: RCCL over xGMI, one process per GPU, no Python / torch involved.
   (The single-process reference has no counterpart: it keeps all floes in one address space, and its periodic ghost floes
   -- collisions.jl:881-1047, folded back at :830-850 -- are the pattern this generalises to tiles.)
     sz_comm_unique_id   on ONE rank: 128 bytes (an ncclUniqueId) that the host passes to every rank by its own channel
                         (MPI broadcast, a file, ...)
     sz_comm_init        on every rank, collectively: the communicator of this context (nranks == 1: no RCCL needed)
     sz_tile_setup       after sz_upload_floes + sz_tile_enable: domain lengths and periodicity, the drift margin (metres a
                         floe may move between two box gathers, on top of the interaction range) and the LONGEST gather
                         interval: the library starts with 8 steps and sizes every following interval from the displacement
                         it measured and the largest velocity now (30 % of the margin), at most doubling it.  rebox_every < 0:
                         exactly every |rebox_every| steps, no adaptation (tests of the drift error)
     sz_tile_run         nsteps x timestep_sim! of the tiled run, collectively, same arguments on every rank.  Per step:
                         grouped ncclSend / ncclRecv (whole regions with the neighbouring tiles, real counts in the header
                         records, slots per pair sized at the last gather; the header record with every rank), forcings of
                         the owned floes beside the exchange, unpack + step; the halo records of the next step are written by
                         the step's integrator (a pack launch starts a batch), each floe as the update left it, before a swap
                         with its ghost -- the receiving rank swaps it with the owner's routine, so a floe's instances are the
                         same bits with the same ghost numbers on every rank.  Two-way coupling partial sums are all-reduced
                         (those batches run to their end: no tag stop).  A floe that moves further than half
                         the margin between two gathers is an error (halo-drift bit), never a silently missed contact.
                         As with sz_step the batch ends after the first step that leaves a floe tagged remove / fuse ON ANY RANK
                         (simplify_floes!, simulation.jl:205-214, runs after every step and is the host's): every rank's header
                         record carries its stop request to every other rank with the next exchange, whose unpack kernel ends the
                         batch there before that step has touched anything.  *steps_done (may be NULL) is the same number on every
                         rank; status.fuse_idx (sz_download_fuse) of a tiled context names partners by GLOBAL floe index.
                         SZ_NO_STOP runs all steps.
                         Device errors are per rank; the ranks agree on them at every box gather and at the end of the call,
                         so that EVERY rank returns the same code at the same step (a rank leaving on its own would hang the
                         others: RCCL has no timeout).  A new sz_upload_floes invalidates sz_tile_enable / sz_tile_setup.
     sz_comm_allreduce   sum of n doubles in device memory over the ranks, in place (sz_eulerian_partial / sz_two_way_partial) */
int sz_comm_available(void);   /* SZ_OK when RCCL can be bound in this process; ask on every rank and agree before sz_comm_init (collective) */
int sz_comm_unique_id(void *id128);
int sz_comm_init(sz_ctx *ctx, int32_t nranks, int32_t rank, const void *id128);
/* The same tiled run over the HOST's own channel between its ranks, for hosts whose ranks cannot open an RCCL communicator
   (an MPI.jl build that is not device-aware, several ranks sharing one GPU -- RCCL refuses that, so this is also how the
   library's multi-rank exchange is rehearsed on a one-GPU box): sz_comm_init_host instead of sz_comm_init, everything else
   as above.  The library stages the regions through host memory and calls the three collectives below with HOST pointers;
   they block, return 0 on success, and are called by every rank in the same order.
     allgather          `bytes` from every rank into recv (nranks x bytes, in rank order)
     sendrecv           npeers point-to-point transfers in each direction, all of them in flight together:
                        send[k] / send_bytes[k] to rank peer[k], recv[k] / recv_bytes[k] from rank peer[k]
                        (a peer's send_bytes to this rank equals this rank's recv_bytes for it; either may be 0)
     allreduce_sum_f64  element-wise sum of n doubles over the ranks, in place */
typedef struct sz_host_transport {
  void *user;
  int (*allgather)(void *user, const void *send, void *recv, int64_t bytes);
  int (*sendrecv)(void *user, int32_t npeers, const int32_t *peer, const void *const *send, const int64_t *send_bytes,
                  void *const *recv, const int64_t *recv_bytes);
  int (*allreduce_sum_f64)(void *user, double *buf, int64_t n);
} sz_host_transport;
int sz_comm_init_host(sz_ctx *ctx, int32_t nranks, int32_t rank, const sz_host_transport *transport);
int sz_comm_destroy(sz_ctx *ctx);
/* one-rank self test of the RCCL binding (run-time loading, communicator of one rank, all-gather, all-reduce, grouped
   send / receive to self with the stream hand-shake of sz_tile_run): what of the exchange can run on a one-GPU box */
int sz_comm_selftest(sz_ctx *ctx);
int sz_comm_allreduce(sz_ctx *ctx, void *d_buf, int64_t n);
int sz_tile_setup(sz_ctx *ctx, double Lx, double Ly, int32_t periodic_x, int32_t periodic_y, double drift_margin,
                  int32_t rebox_every);
/* optional, after sz_tile_setup: the centre of this rank's tile.  The bounding box of the owned floes -- what the peers select this rank's
   halo with -- takes every centroid at its periodic image nearest to that point at the first gather (later gathers use the centre of the
   box before), so that a parent the ghost pass has wrapped to the far side of the domain does not stretch the box across the domain */
int sz_tile_set_center(sz_ctx *ctx, double x, double y);
int sz_tile_run(sz_ctx *ctx, int32_t nsteps, int32_t tstep0, int32_t dt, int32_t coupling_dt, int32_t flags, int32_t *steps_done);
/* Migration (collective, between two sz_tile_run calls): every owned floe is re-assigned to the tile that holds its centroid -- px x py
   tiles over the domain of sz_set_domain, tile (ix, iy) = rank iy * px + ix; owner_override (may be NULL; one entry per owned floe, in
   the order of the last upload) names the new owner rank instead.  Floes that change tile travel with their complete state (all columns,
   tensors, status, ring, sub-floe points) over the library's channel (RCCL or the host transport); the context is then rebuilt from the
   kept and the received floes, ordered by global index, exactly as an upload + sz_tile_enable + sz_tile_setup (same parameters) would.
   The movers are packed on the device, one stream per destination, travel device to device (RCCL grouped send / receive; a host
   transport stages those streams only) and the rows are gathered into their new order on the device: the host sees the owner and
   offset columns and a directory of what arrived, no floe data (csrc/sz_migrate.hpp).  The capacities of the last upload stay; a
   tile that would crowd them (more than 1/8 above what that upload held, on any rank) is rebuilt through a download and an upload
   instead -- the host-staged path, on every rank (SZ_MIGRATE_HOST=1 forces it; sz_debug_migrate_path: 1 device, 2 host-staged).
   *n_sent = floes this rank gave away, *n_owned = floes it owns now (sz_tile_owned_gidx names them; the host's own copies of the
   columns are stale).  floe.interactions do not travel.
   sz_download_subpoints: the sub-floe points of the floes the context holds (off: N + 1 entries; sx == NULL: offsets only). */
int sz_tile_migrate(sz_ctx *ctx, int32_t px, int32_t py, const int32_t *owner_override, int64_t *n_sent, int64_t *n_owned);
int sz_tile_owned_gidx(sz_ctx *ctx, int64_t *gidx, int64_t n_cap);
int sz_debug_migrate_path(sz_ctx *ctx);
/* diagnosis: the ghost / halo row that carried order key `key` in the last resident step that used ghost allocator `slot` (0-based step & 1), as the
   collision kernels saw it -- out56: row (-1: none), cx, cy, u, v, xi, rmax, area, height, box x0 x1 y0 y1, ring points, parent, status, ring x[20], y[20] */
int sz_debug_find_key(sz_ctx *ctx, int32_t slot, int64_t key, double *out56);
/* diagnosis: pair items of the last resident step between the instances (parent, ghosts) of two floe ids -- out61: entries, then {owner key, partner
   key, contact rows, owner row, partner row} for up to 12 of them */
int sz_debug_pairs_of_ids(sz_ctx *ctx, int32_t slot, int64_t id_a, int64_t id_b, double *out61);
int sz_download_subpoints(sz_ctx *ctx, int32_t *off, double *sx, double *sy);

/* ---- output path on the resident state
   sz_eulerian_data: calc_eulerian_data! (output.jl:793-914), the GridOutputWriter averages, over the rows the
   context holds -- write_data! runs after add_ghosts! (simulation.jl:102-105), so callers that step with sz_step
   bracket it with sz_add_ghosts / sz_remove_ghosts.  xg (nx + 1) and yg (ny + 1) are the writer's grid lines (evenly
   spaced, as GridOutputWriter builds them, output.jl:352-353); outputs lists nout of the SZ_EUL_* codes;
   data[k][ix][iy] at (k * nx + ix) * ny + iy is writer.data[ix + 1, iy + 1, k + 1].  Topography is taken out of
   the cells as the reference does (elements must not overlap one another).
   The FloeOutputWriter (write_floe_data!, output.jl:558-574) writes whole columns: sz_download_floes with only
   the wanted columns non-NULL is its device-side packing.
   sz_simplify_check: what simplify_floes! (simplification.jl:339-378) would find to do -- out4 = floes tagged remove,
   tagged fuse, rings with more than max_vertices points (smooth_floes!, :66) and floes under min_floe_area /
   min_floe_height that are not tagged remove (remove_floes!, :287-290).  All four zero: the pass changes nothing and
   no geometry needs to leave the device. */
enum { SZ_EUL_U = 0, SZ_EUL_V, SZ_EUL_DUDT, SZ_EUL_DVDT, SZ_EUL_OVERAREA, SZ_EUL_MASS, SZ_EUL_AREA, SZ_EUL_HEIGHT,
       SZ_EUL_SI_FRAC, SZ_EUL_STRESS_XX, SZ_EUL_STRESS_YX, SZ_EUL_STRESS_XY, SZ_EUL_STRESS_YY, SZ_EUL_STRESS_EIG,
       SZ_EUL_STRAIN_UX, SZ_EUL_STRAIN_VX, SZ_EUL_STRAIN_UY, SZ_EUL_STRAIN_VY, SZ_EUL_COUNT };
int sz_eulerian_data(sz_ctx *ctx, int32_t nx, int32_t ny, const double *xg, const double *yg, int32_t nout,
                     const int32_t *outputs, double *data);
/* tiled (multi-GPU) runs: a floe and its ghosts count on the rank that owns it.  Every rank calls
   sz_eulerian_partial (its per-cell sums -> SZ_EUL_PARTIAL * nx * ny doubles in device memory), the host adds the
   buffers up across the ranks (all-reduce), sz_eulerian_finish turns the sums into the averages on every rank. */
enum { SZ_EUL_PARTIAL = 17 };
int sz_eulerian_partial(sz_ctx *ctx, int32_t nx, int32_t ny, const double *xg, const double *yg, void *d_partial);
int sz_eulerian_finish(sz_ctx *ctx, int32_t nx, int32_t ny, const double *xg, const double *yg, const void *d_partial,
                       int32_t nout, const int32_t *outputs, double *data);
int sz_simplify_check(sz_ctx *ctx, int32_t max_vertices, double min_floe_area, double min_floe_height, int64_t *out4);

/* test hook: which_vertices_match_points(points, region) (floe_utils.jl:331-352) as the narrow phase evaluates it, on
   given points (<= 64) and a given closed region ring; idx = sorted 0-based vertex indices */
int sz_debug_match_vertices(sz_ctx *ctx, int32_t npts, const double *px, const double *py, int32_t nr, const double *rx,
                            const double *ry, int32_t *idx, int32_t *n_out);

/* test hook: the in-bounds test of calc_subfloe_values! (in_bounds, coupling.jl:494-597) and the lattice sample of mc_interpolation
   (find_interp_knots + linear_interpolation, coupling.jl:702-902) as the forcing kernels evaluate them, at n given points.
   out12[12 k ..] = in_bounds (0 / 1), uocn, vocn, hflx, uatm, vatm, the 1-based grid lines west, east, south, north the bilinear
   blend reads, its weights tx, ty.  Needs sz_set_domain and sz_set_fields. */
int sz_debug_sample_fields(sz_ctx *ctx, int32_t n, const double *x, const double *y, double *out12);

/* test hook: 1 when the last sz_step batch ran as pipelined steps (two launches per timestep: csrc/sz_pipeline.hpp), 0 otherwise */
int sz_debug_pipelined(sz_ctx *ctx);
/* diagnostic build (-DSZ_STAMPS) only: stamp log of one lane group of the narrow phase
   (out512[0] = entries, then (stage << 48) | cycles since the wave started) */
int sz_debug_stamps(sz_ctx *ctx, long long *out512);
/* test hook: parts of the collision records (the per-floe 128-byte cache of the columns the neighbour search and the narrow phase read;
   DESIGN.md section 3.00) of the owned parents that differ from the columns after the last resident batch -- 0 expected; -1: that batch
   did not run on records */
int sz_debug_crec_mismatches(sz_ctx *ctx, int64_t *n_bad);

#ifdef __cplusplus
}
#endif
#endif
