// sz_state.hpp — HBM layout of the engine: struct-of-arrays floe columns,
// CSR vertex rings, the domain-element table, broad-phase and contact-row workspaces.
// The struct is passed BY VALUE to every kernel (kernarg), so it only holds raw device
// pointers and host-known capacities; all per-step counts live in the device counter block
// `cnt` so that a whole timestep is enqueued without a host round trip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sz {

// device counter block
enum {
  C_M = 0,        // floes incl. ghosts
  C_N,            // parents
  C_NV,           // ring points in use (vx/vy)
  C_NPAIRS,       // narrow-phase pairs
  C_NELEM,        // floe-domain-element items
  C_NINTER,       // interaction rows (all floes)
  C_ERR,          // sticky error bits (szg::ERR_*)
  C_NPAIR_PTS,    // sum of ring points over pairs (stats)
  C_NPAIR_ROWS,   // contact rows before mirroring (stats)
  C_NELEM_ROWS,
  C_NGCAND, C_NGCAND1,   // entries of the two ghost-candidate lists (State::gcand), used alternately step by step
  C_NFUSE,        // pair items that asked for a fuse in the last collision call (the host replays the fuse lists only then)
  C_RETRYSTOP,    // resident batches: 0, or 1 + the batch-relative step whose narrow phase met an item for the largest variant while that
                  // variant was not enqueued (State::retry_stop): the batch pauses after that step's narrow launch and the host finishes it
                  // (the guard counters live in State::warn: one word each serialised the chip)
  C_NG_NEW,       // ghosts created by the current pass
  C_NGHOSTS,
  C_NCELLS,
  C_SCRATCH0, C_SCRATCH1,
  C_TRACE_FAIL,
  C_NOWN,         // floes this context integrates (== C_N unless tiled: owned floes come first)
  C_NHALO,
  C_ITEMCLASS,    // largest narrow-phase size class among this step's items
  C_NRETRY,       // items handed on to the largest narrow variant (diagnostic, cumulative)
  C_NWORK,        // pairs whose ring boxes overlap: the pair items the narrow phase runs
  C_NENT,         // two-way coupling: (floe, centre cell) entries of the current coupling step
  C_STOP,         // resident batches (sz_step): 0, or 1 + the batch-relative step after which the batch stops -- a floe was tagged
                  // remove / fuse (or fell under the dissolve thresholds), so the host's simplify_floes! (simulation.jl:205-214) has work
  C_DRIFT,        // tiled runs: largest displacement of an owned floe since the last box gather, metres as float bits (pack kernel)
  C_PAUSED,       // the step C_RETRYSTOP names was paused by THIS context's narrow phase (tiled runs: a peer's pause arrives in the halo headers
                  // and sets C_RETRYSTOP alone)
  C_FRCSTOP,      // resident batches: 0, or the batch-relative step whose forcing kernel found a floe without an in-bounds sub-floe point (it will be
                  // tagged `remove` by that step's integrator, which ends the batch there): a HINT for that integrator -- "this step is the batch's
                  // last: keep the ghost links, make no ghosts for a next step" -- raised a launch earlier than the tag itself
  C_COUNT = 32
};

constexpr int MAXNB = 24;       // broad-phase neighbours kept per floe and direction: the default of State::maxnb (fields with a size spectrum get 64)
constexpr int CELL_K = 8;       // floes a broad-phase cell holds in its bucket (more: overflow chain)
constexpr int NSEG = 8;         // segments of the narrow phase's work list (one tail counter and one queue head each, a cache line apart)
constexpr int ROWS_PER_ITEM = 16; // contact rows kept per pair / element item (the 8-lane kernels hold 4 regions and hand larger items on)
constexpr int WARN_SLOTS = 256;
constexpr int ACC_SLOTS = 256;
constexpr int MAX_GHOSTS = 3;   // ghosts per parent (doubly periodic corner floe)

struct Params {
  double E, nu, mu, rho_o, rho_a, Cd_io, Cd_ia, fcor, turn;
  double ff_max_overlap, fd_max_overlap;
  double rho_i, max_h, max_xi, lambda;
  int dd;
  double Cd_ao, k_ice, L_ice;   // two-way coupling: atmosphere-ocean drag, conductivity and latent heat of ice (Constants())
};

struct State {
  // ---- capacities (host constants)
  int capM, capV, capPairs, capElem, capRows, capCells, capS;
  int nelem;                 // 4 boundaries + topography elements
  int any_periodic_ew, any_periodic_ns, any_domain_work;
  int tiled;                 // halo mode: order keys are global indices
  int step;                  // 1-based index of the step inside the running sz_step batch (0: process-mode call): kernels of
                             // steps after a stop request return at once (stopped(), sz_kernels.hpp)
  int retry_stop;            // sz_step: the largest narrow variant is not enqueued; an item that needs it raises C_RETRYSTOP
  int stop_on_tags;          // sz_step: raise C_STOP when a parent is tagged (off with SZ_NO_STOP)
  int goff;                  // pipelined resident steps (sz_pipeline.hpp): the rows a step's inline makers allocate lie at [N + goff, ..) -- two regions, used
                             // alternately step by step, so that the ghosts of step t + 1 can be made while those of step t are still read; 0 elsewhere
  int halo_ring;             // tiled runs: ring points a halo record has room for (>= 32; sz_tile_enable sizes it from the largest ring of ALL ranks, <= 255);
                             // a record is 12 + 2 * halo_ring doubles (sz_kernels.hpp halo_rec)
  int gcap;                  // rows the step's inline makers may allocate (pipelined steps: the size of the step's region; 0: all rows behind the parents)
  int pipe;                  // this launch belongs to a pipelined step (the ghost maker leaves the parent's COLUMNS alone and marks a swapped parent)
  int restart_on_tags;       // pipelined batches that run through (SZ_NO_STOP): a NEW tag still ends the enqueued steps -- the host starts the rest again,
                             // so that the next step's ghosts are made knowing the tag (the steps' ghosts are made one launch ahead of the tags)
  int xcd_forcing;           // SZ_XCD_FORCING=1: XCD-contiguous floe ranges in the forcing kernels (A/B switch; default off: slower at 100 k)
  int xcd_neigh;             // SZ_XCD=1: XCD-contiguous floe ranges in the neighbour search too (A/B switch; default off)
  // ---- counters
  int* cnt;
  int* warn;                 // guard counters of timestep_floe_properties!: WARN_SLOTS slots of 32 ints (one 128-byte line each; words 0..3 =
                             // height, force, velocity, xi) -- a single counter word serialises the whole chip's atomics (8 ns each)
  // ---- floe columns
  double *cx, *cy, *rmax, *area, *height, *mass, *moment, *alpha, *u, *v, *xi;
  double *p_dxdt, *p_dydt, *p_dalphadt, *p_dudt, *p_dvdt, *p_dxidt;
  double *fxOA, *fyOA, *trqOA, *hflx, *overarea, *cfx, *cfy, *ctrq;
  double *sa, *si, *strain;   // 4 per floe
  long long *id, *ghost_id, *okey;   // okey: position in the reference's serial order
  int *status, *parent, *gh, *ngh, *frc_remove;   // gh: MAX_GHOSTS per floe
  int *gh_save, *ngh_save;                         // the ghost links sz_k_remove_ghosts has just cleared (the rows of a batch's last step may be assembled once more: sz_step)
  signed char* osign;                // ring orientation sign
  double *bbx0, *bbx1, *bby0, *bby1; // ring bounding boxes (kept current by every kernel that moves a ring)
  int* voff; double2* vxy;           // rings, CSR: point k of floe i at vxy[voff[i] + k] = {x, y} (closed: the first point repeated) -- interleaved,
                                     // so that whoever moves, copies or stages a ring issues one 16-byte access per point instead of two 8-byte ones
  int* soff; double *sx, *sy;
  // the sub-floe points once more, {x, y} interleaved and every floe's points in BLOCKED order (Morton order of the body-frame coordinates on a quarter
  // of the lattice spacing: sz_k_block_points).  The one-way forcing loop reads these: 16 consecutive points then lie in one or two lattice cells
  // instead of along a row of the floe's sub-grid, and the lattice loads of a wavefront touch a third as many cache lines -- the loop was bound by
  // the lines its loads touch (round 4: 79.6 M line accesses per launch at 100 k floes = 311 k cycles per CU of a 327 k-cycle kernel).  A derived
  // copy like s32 (null until made; made again after an upload, a migration or new fields); the per-floe sums of the one-way path are held to a
  // tolerance, not to the point order -- the two-way kernel, whose per-cell sums ARE ordered, keeps reading sx / sy.
  double2* sxy;
  // ---- domain elements: 0..3 = N,S,E,W boundaries, 4.. = topography
  int* eoff; double *ex, *ey;
  int *ekind, *edir; double *eval, *eu, *ev, *ecx, *ecy, *ermax, *erect;  // erect: 4 per boundary
  signed char* eosign; double* ebb;   // element orientation signs and boxes (4 per element)
  // ---- grid fields
  int Nx, Ny; double gx0, gxf, gy0, gyf, gdx, gdy, rdx, rdy;   // rdx = 1/gdx
  double *uo, *vo, *hf, *ua, *va;
  double* nodes;             // the five lattices interleaved per node (8 doubles each)
  // ---- mixed precision (sz_set_precision): fp32 copies the forcing kernel reads -- sub-floe points (x, y) and
  // the interleaved lattice (8 floats per node: uo, vo, hf, ua, va, -, -, -)
  float2* s32; float* nodes32;
  // ---- mixed precision, geometry (sz_set_precision(1), resident steps of single-context runs with small rings): the ring of a
  // floe is kept ONCE, in its body frame (offsets from the centroid at alpha = 0) as fp32, together with the fp64 pose (centroid,
  // cos / sin alpha in `trig`); world coordinates are reconstructed in fp64 where a kernel needs them (narrow-phase staging, the
  // integrator's box and strain), so the integrator does not rewrite rings and ghosts share their parent's ring.  rb_off / rb_n:
  // ring offset into ring32 and ring size per floe (a ghost: its parent's).  body_rings: this launch works on them (kernarg);
  // the world rings vx / vy are then stale until sz_k_world_rings rebuilds them (any call that needs them does that first).
  float2* ring32; int *rb_off, *rb_n; int body_rings;
  // fp32 broad-phase record per floe {cx, cy, rmax, -} {box x0, x1, y0, y1}: the neighbour search of mixed mode tests candidates
  // on it with a conservative margin and confirms the survivors with the exact fp64 predicate (the pair list stays bit-exact)
  float4* rec32;
  // ---- collision record (round 3): what the neighbour search needs of a CANDIDATE and the narrow phase's staging of an item's two floes,
  // in ONE 128-byte line per row instead of ~13 scattered 8-byte columns (the texture path works off one line per lane and load
  // instruction: the candidate loads were half of the neighbour search's time).  Eight 16-byte quads:
  //   0 {cx, cy}  1 {rmax, id}  2 {order key | ring points << 48 | (ring orientation < 0) << 56,  ring offset | parent << 32 | ghosts << 60}
  //   3 {box x0, x1}  4 {box y0, y1}  5 {u, v}  6 {xi, area}  7 {height, ghost_id}
  // A CACHE of the columns, not their replacement: non-null only in the launches of resident batches whose kernels keep it current
  // (a seeding launch at the start of the batch, then whoever places a floe: the one-launch integrator, the inline ghost maker, the
  // inline halo unpack); everything else reads and writes the columns as before.
  double2* crec;
  // ---- two-way coupling (allocated by sz_set_two_way): per floe the centre cells its sub-floe points fall into
  // (FC_CAP slots per floe: cell id, shift code, sum of -tau_ocn, points), per cell the entries sorted by floe
  int *fc_key, *fc_n, *fc_cnt; signed char* fc_code; double *fc_tx, *fc_ty, *fc_area;
  int *cl_cnt, *cl_off, *cl_cur, *cl_ent;
  double *t_ocn, *t_atm, *tau_x, *tau_y, *si_frac;
  // ---- ghosts workspace
  int *gflag, *gvscan;        // gvscan: ring offsets of the halo records being unpacked
  int4 *gplan, *gscan4, *gtot4;
  // ghosts made by the kernel that places the parent (resident steps, "inline" ghosts: sz_kernels.hpp ghost_inline_make): two
  // allocators {ghosts << 32 | ring points} a cache line apart (galloc[0], galloc[16]), used alternately step by step, and per
  // allocator the order key of the ghost at every storage offset (gkeys[slot * capM + offset])
  // ... and, per parent that has ghosts, a record of all instances of its id (itself + its ghosts): what the Dict rule of the pair
  // loop needs about a floe's "family", in one contiguous read instead of a chain through parent -> ghost links -> rows
  struct Fam { long long key[4], gid[4]; double cx[4], cy[4], r; int n, pad; };
  Fam* fam;
  unsigned long long* galloc; long long* gkeys;
  int ginline, gslot;         // inline ghosts on; the allocator holding the ghosts of the step being launched
  int famrec;                 // every ghost present was made inside a step by a maker that leaves the family records (State::Fam): the Dict rule reads them
  int4* gcand;                // two lists of capM entries {parent, ghost flags, ring points, -}: the parents that get ghosts in the next
                              // step, appended by whoever places a floe (integrator, halo unpack) -- see sz_k_ghost_list
  int4 *lb_agg, *lb_inc; unsigned* lb_flag;   // decoupled look-back scan: per workgroup aggregate, inclusive prefix, (epoch << 2 | status)
  // ---- broad phase
  double* bounds;            // xmin, ymin, cell size, (ncx, ncy as doubles)
  // cells: a bucket of CELL_K floes per cell (count in cell_cnt; readers fetch count and bucket in ONE round trip instead of
  // walking a list node by node), floes beyond that on an overflow chain (cell_ovf: head + 1, cell_items: next links)
  int *cell_cnt, *cell_slots, *cell_ovf, *cell_items;
  // per floe k: the neighbours that come later in the serial order (nb_out: the pairs k owns, sorted by order key; pair SLOT
  // = k * MAXNB + rank) and earlier (nb_in: the pairs mirrored onto it)
  int *nb_out, *nb_in, *n_out, *n_in;
  double* over_base; int* over_stamp; int callid;       // floe.overarea before the collision call `over_stamp` (sz_k_inter_fill); collision call that last added its overlap to floe.overarea (sz_k_inter_fill): a call run again after its lists grew adds nothing twice
  int maxnb, rowcap;          // neighbours kept per floe and direction (stride of nb_out / nb_in / the pair slots: 24, 64 or 256 -- chosen at upload, grown on demand), interaction rows per floe
  int *out_off, *pair_i, *pair_j;      // the compact pair list in serial order: made on demand (sz_download_pairs) / given (sz_collide_pairs)
  // work list of the narrow phase: the pair items to run, two int4 each {slot, i, j, ring offset i} {ring size i, ring offset j, ring size j, -}, appended by the neighbour search in NSEG segments of
  // capPairs / NSEG items; wq[s * 32] = queue head of segment s (rounds after the first), wq[s * 32 + 1] = its length
  int4* work;
  int* wq;
  // ---- element items
  int *el_off, *el_floe, *el_elem;
  // ---- contact rows per item (pairs first, then element items at capPairs + e)
  // results per item: rows at work index w (pairs: segment * (capPairs / NSEG) + position; elements: capPairs + e);
  // it_info[pair slot] / it_info[capM * MAXNB + e] = {rows | flags << 8, w}
  double* it_rows; int2* it_info;
  // ---- per-floe interaction lists
  int *inter_cnt, *inter_off, *tagA; double* inter_rows;   // ROWCAP rows per floe; inter_off: download scratch
  // ---- scan scratch
  int* blk;
  // ---- motion scratch (integrator)
  double* mot;               // 4 per floe: dx, dy, cos, sin
  double* mot2;              // 2 per floe: cos / sin of the step's rotation (pipelined batches: what the un-swap behind a batch needs beside mot = {old cx, cy, dx, dy})
  double* trig;              // 2 per floe: cos(alpha), sin(alpha), kept current by the upload and the integrator
  long long* stamps;         // diagnostic build (-DSZ_STAMPS) only
  // ---- per-floe collision totals as order-independent fixed-point sums (round 4; sz_geom.hpp "fixed-point totals"): FX_WORDS = 16 int64 words per row
  //   0 sum fx   1 sum fy   2 sum (x - cx) fx   3 sum (y - cy) fx   4 sum (x - cx) fy   5 sum (y - cy) fy   6 sum overlap area   7 tag bits (low word)
  //   8 .. 11 the low words (40 more bits) of 0, 1, 3, 4
  // accumulated by the narrow phase with one atomic per word and row side (no reduce launch in the resident steps: the integrator reads
  // the eight words, forms collision_force / collision_trq / the stress sums of calc_stress! from them and clears them).  Integer sums do
  // not depend on the order of the additions, so a tile and the single context -- and any two runs -- give the same bits.  Non-null only in
  // the launches of resident batches; process-mode calls keep the serial double sums of sz_k_inter_fill.
  long long* facc; int kexp;   // kexp: ilogb of (1 + mu) E, + 1 (host, sz_set_params): the force scale of a floe is 2^(kexp + ...), see fx_force_exp
  unsigned long long* acc;   // cumulative work counters of the narrow phase (ACC_SLOTS lines of 8 words: launches, pair items run,
                             // their ring points, pair rows, element items, element rows): what a launch averaged over a window of
                             // steps really did, for the roofline of bench.py (sz_get_stats acc_*, cleared by sz_profile_reset)
};

// what a launch of a pipelined step (sz_pipeline.hpp) needs of the OTHER step parity -- the State it is given holds one parity's pointers
struct PipeAlt {
  double2 *crec, *vxy;
  int *cell_cnt, *cell_slots, *cell_ovf, *cell_items;
  int4* work; int* wq;
  int *gh, *ngh;
  int goff, gslot;
  int make_ghosts;        // GEO: 0 in a step the host knows to be the batch's last (no ghosts, no swap: there is no step to make them for)
};

using szg_flags_note = int;   // IT_FUSE / IT_REMOVE are defined in sz_geom.hpp

}  // namespace sz
