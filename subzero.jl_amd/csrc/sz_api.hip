// sz_api.hip — context, HBM allocation, launch orchestration and the extern "C" boundary
// declared in include/subzero_hip.h.  Host code here is plumbing; the arithmetic is in
// sz_kernels.hpp / sz_geom.hpp.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/subzero_hip.h"
#include "sz_kernels.hpp"
#include "sz_pipeline.hpp"
#include "sz_twoway.hpp"
#include "sz_output.hpp"
#include "sz_migrate.hpp"
#include <rocprim/rocprim.hpp>      // device radix sort of the output-grid entries (sz_eulerian_data)

using namespace sz;

namespace {

constexpr int NK = SZ_K_COUNT + 2;   // + large narrow variant, + the halo exchange of a tiled step (events on the communication stream)
constexpr int K_NARROW_LARGE = SZ_K_COUNT, K_EXCHANGE = SZ_K_COUNT + 1;
#ifndef NARROW_G
#define NARROW_G 8
#endif
#ifndef NARROW_KC0          // working set of the first narrow variant: crossings, region points
#define NARROW_KC0 8
#define NARROW_RC0 16
#endif

struct EvPair { int k; hipEvent_t a, b; };

// Device allocations of one lifetime.  The ~130 columns and work arrays are carved out of a few large chunks
// instead of one hipMalloc each: the chunks are mapped with 2 MB fragments, so a kernel that walks 60 columns
// needs a handful of TLB entries instead of several per column.
struct Pool {
  std::vector<void*> chunks; std::vector<size_t> sizes;
  size_t ci = 0;                 // chunk being carved
  char* cur = nullptr; size_t left = 0, next = 8u << 20;
  bool empty() const { return chunks.empty(); }
  void push_back(void* q) { chunks.push_back(q); sizes.push_back(0); }      // a stand-alone allocation handed to the pool
};

}  // namespace

struct sz_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  hipStream_t stream2 = nullptr;        // forcings beside the collision kernels
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  State S{};
  Params P{};
  std::string err;
  Pool allocs;        // per-upload allocations
  Pool list_allocs;   // the lists whose capacity follows the field and GROWS on demand (grow_lists): neighbour lists, pair items, item rows
  int callid = 0;     // collision calls so far (State::callid: a call run again after its lists grew adds its overlap to floe.overarea once)
  Pool inter_allocs;  // floe.interactions (inter_cnt, inter_rows): survive an upload of the same size -- a shim uploads between
                      // timestep_collisions! and timestep_floe_properties!, and calc_stress! reads the rows of the collisions
  int inter_capM = 0, inter_rowcap = 0; bool inter_any = false, inter_lost = false;
  int nb_count_max = 0;             // bounding-circle neighbours of the most crowded floe at upload (inflated circles): sizes State::maxnb
  Pool static_allocs; // domain element table
  Pool field_allocs;  // ocean / atmosphere lattices
  bool have_floes = false, have_domain = false, have_fields = false;
  int hostM = 0, hostN = 0;
  // element table (host copy, rebuilt on set_domain / set_topography)
  int h_kinds[4] = { 0, 0, 0, 0 };
  double h_vals[4] = { 0, 0, 0, 0 }, h_rects[16] = { 0 }, h_bu[4] = { 0 }, h_bv[4] = { 0 };
  std::vector<int> h_toff; std::vector<double> h_tx, h_ty, h_tcx, h_tcy, h_trmax;
  // profiling
  unsigned pmask = 0;               // bit k: kernel class k is event-timed
  std::vector<EvPair> evs; size_t ev_used = 0;
  double kms[NK] = { 0 }; long long kl[NK] = { 0 };
  // fuse bookkeeping (status.fuse_idx lives on the host: it only changes on rare fuse events)
  std::vector<std::vector<int>> fuse_lists;
  long long* d_stats = nullptr;
  int last_dt = 0;
  bool any_moving = false;
  int overlap_forcing = -1;       // -1: by size (fp64 fields above 65 536 floes, where the forcings have a launch of their own: 0.585 -> 0.567 ms/step at 100 k; not
                                  // in mixed precision: 0.181 -> 0.187 on configs[4]); SZ_OVERLAP=0|1 forces it.  SZ_OVERLAP=1: forcings on a second stream beside the broad / narrow / reduce kernels.  The fork/join
                                  // costs ~10 us; riding in the neighbour launch (fuse_forcing) is as good or better at every size
  int max_sub = 0;                  // most sub-floe points of one floe (sizes the LDS of the two-way forcing kernel)
  int max_ring = 0, max_elem_ring = 5, max_ring_tiled = 0;   // largest ring sizes (host knowledge: which narrow variants can be needed)
  int narrow_grid0 = 0;
  // mixed precision (sz_set_precision): fp32 copies for the forcing kernel, rebuilt when their sources change
  int precision = 0; bool mixed_pts_ok = false, mixed_nodes_ok = false; Pool mixed_pt_allocs, mixed_node_allocs;
  bool blk_pts_ok = false, no_block_points = false; Pool blk_pt_allocs; int pts_N = 0;      // State::sxy (ensure_block_points); pts_N: floes whose soff entries are set (upload, migration)
  // mixed precision, geometry: fp32 broad-phase records and body-frame rings (sz_state.hpp); rings_stale: resident steps ran on the
  // body rings, the world rings vx / vy are behind (rebuilt by world_rings() before anything else looks at them)
  bool mixed_geom_ok = false, rings_stale = false, no_body_rings = false; Pool mixed_geom_allocs;
  // two-way coupling (off by default, like CouplingSettings())
  bool tw_general_clip = false;   // SZ_TW_GENERAL_CLIP=1: floe-in-cell areas by the general clipper (8 lanes per entry) instead of the rectangle pipeline
  bool two_way = false; int tw_dt = 10; int tw_capM = 0; size_t tw_ncell = 0; bool temps_set = false;
  Pool tw_allocs, tw_field_allocs;
  // static broad-phase grid of the resident steps (fixed by the host: no bounds reduction per step)
  // inline ghosts (sz_kernels.hpp ghost_inline_make): the resident steps make a step's ghosts in the kernel that places their parents,
  // in allocation order; the reference's ghost numbers are recovered from the order keys of the last step that ran
  bool ghost_inline = true;         // SZ_GHOST_INLINE=0: the candidate-list launch instead
  bool gi_valid = false;            // the interaction rows / pair lists on the device carry order keys of inline ghosts
  std::vector<long long> gi_keys;   // order key of the ghost at storage offset k (floe N + k) in the last step that ran
  bool gi_pending = false; int gi_pending_n = 0, gi_pending_slot = 0;      // ... still to be fetched from the device (gi_fetch)
  std::vector<int> gi_ref;          // ... and its number among the ghosts in the reference's order (ghost N + gi_ref[k])
  bool retry_seen = false;          // an item has needed the largest narrow variant: sz_step enqueues it in every step from now on
  bool no_elems_ride = false;       // SZ_ELEMS_RIDE=0: the element items always get their own launch
  bool no_lean_narrow = false;      // SZ_LEAN_NARROW=0: always enqueue it
  bool no_crec = false;             // SZ_CREC=0: no collision records (State::crec) in the resident steps
  double* frc_alt[4] = { nullptr, nullptr, nullptr, nullptr };      // second set of the forcing outputs fxOA, fyOA, trqOA, hflx (tiled steps with peers, see sz_tile_run)
  bool tile_forcing_in_tail = false;      // SZ_TILE_FORCING_TAIL=1: tiled steps with peers keep the forcings in the narrow launch's tail (A/B switch)
  double2* crec_buf = nullptr;      // the records' memory (State::crec points at it only inside the batches that keep it current)
  bool crec_was_live = false;       // the last resident batch ran on records (sz_debug_crec_mismatches)
  int forcing_where = -1;           // sz_forcing_launch
  int fuse_forcing_mode = 0;        // ... 1: in the neighbour launch, 2: in the narrow launch (its tail), 0: by size -- the narrow launch while the narrow phase is one
                                    // round with a long tail (measured better up to 20 k floes, even at 40 k, worse at 65 k); SZ_FUSE_FORCING=1|2 forces one
  bool fuse_forcing = true;         // forcings inside the neighbour launch (sz_k_neighbors_forcing); SZ_FUSE_FORCING=0: own launch
  bool fused_move = true;           // integrate + move/strain in one thread-per-floe launch when rings are small (-2 us at 10k); SZ_FUSED_MOVE=0: two launches
  bool no_queue = false;            // SZ_NARROW_QUEUE=0: static split of the narrow items over the workgroups
  bool no_static_grid = false;      // SZ_STATIC_GRID=0: fit the grid to the centroids every step (sz_k_bounds), as process mode does
  double rmax_max = 0.0, rmax_hint = 0.0; bool grid_ok = false, grid_live = false; double h_grid[8] = { 0 };
  unsigned scan_epoch = 0;      // launch counter of the look-back scans (their flags carry it: no reset pass)
  // ghost-candidate lists of the resident steps (sz_k_ghost_list): gl_cur = the list the next step consumes, gl_valid = it is
  // current (kept so by the integrator / halo unpack; any process-mode call or upload makes it stale: it is then seeded again),
  // gl_est = how long it is (host estimate at upload, device count after every batch): long lists take the two-launch path
  int gl_cur = 0; bool gl_valid = false; int gl_est = 0; bool no_ghost_list = false; int gl_max = 2048;
  // tiled runs with the exchange inside the library (sz_comm_init / sz_tile_setup / sz_tile_run): the RCCL communicator, a second
  // stream for the sends / receives (the forcings of the owned floes run beside them), the exchange buffers and their layout
  void* comm = nullptr; int comm_n = 0, comm_rank = 0;
  sz_host_transport host_tr = { nullptr, nullptr, nullptr, nullptr }; bool host_transport = false;   // sz_comm_init_host: the collectives are the host's, staged through h_send / h_recv
  std::vector<double> h_send, h_recv;
  hipStream_t comm_stream = nullptr; hipEvent_t ev_packed = nullptr, ev_recv = nullptr;
  Pool comm_allocs; double *d_send = nullptr, *d_recv = nullptr, *d_ref = nullptr, *d_gather = nullptr; int* d_dcap = nullptr;
  int halo_cap = 0; std::vector<int> cap_send, cap_recv;      // slots per peer region (stride) and what is really sent to / received from each peer
  double tile_Lx = 0, tile_Ly = 0, tile_margin = 0; int tile_per_x = 0, tile_per_y = 0, tile_rebox_every = 50, tile_since_box = -1, tile_rebox_cur = 8, tile_dt = 0; bool tile_rebox_fixed = false;    // rebox_cur: the gather interval in use (<= rebox_every, from the measured drift)
  Pool tw_part_allocs; double* d_tw_partial = nullptr;
  Pool mig_allocs;                  // scratch of sz_tile_migrate (streams, directory, the gathered rows): kept between migrations
  Pool sub_allocs;                  // sub-floe points of a tile that outgrew State::capS in a migration (sz_tile_migrate): until the next upload
  int upload_M = 0, upload_V = 0;   // floes and ring points of the last sz_upload_floes (what its capacities were carved for)
  int migrate_path = 0;             // how the last sz_tile_migrate ran: 1 packed on the device, 2 staged through the host (sz_debug_migrate_path)
  std::vector<long long> tile_gidx; // global index of every owned floe (sz_tile_enable): status.fuse_idx of a tiled context is reported in global numbers
  bool tile_hdr_neighbours = false; // SZ_TILE_HEADERS=neighbours (measurement only, batches that run through): the inline steps trade with the neighbouring tiles only --
                                    // no header record to the others, hence no tag stop and no pause agreement in that arm (the largest narrow variant stays in)
  bool tile_inline_off = false;     // SZ_TILE_INLINE=0: the tiled steps of sz_tile_run keep the list-based ghost pass, their own forcing launch and the one-workgroup unpack (A/B)
  double tile_box_ctr[2] = { 0, 0 }; bool tile_box_valid = false;   // centre of this rank's owned box at the last gather (sz_k_owned_box: periodic images)
  int tile_forcing_tstep = -1;      // timestep whose forcings sz_tile_forcing has already enqueued
  bool tile_dirty = false;      // ghosts / halo floes of the last sz_tile_step still appended
  // fixed-point totals (State::facc): resident batches only.  acc_mode: what the integrator is told (bit 0: totals / stress sums / tags from facc,
  // bit 1: the batch's last step); reduce_mode: 0 sz_k_inter_fill does everything inside the step (process mode), 1 it only assembles rows inside
  // the step, 2 it is left out of the steps and runs once behind the batch (the reduce-free steps)
  long long* facc_buf = nullptr; int acc_mode = 0, reduce_mode = 0;
  // pipelined resident steps (sz_pipeline.hpp): the second set of what is double-buffered by step parity.  pb[0] is what the upload carved
  // (State::vxy, crec_buf, the cell lists, the work list, the ghost links), pb[1] its twin; gpar: the set that holds the context's state.
  struct PipeBuf { double2 *vxy = nullptr, *crec = nullptr; int *cell_cnt = nullptr, *cell_slots = nullptr, *cell_ovf = nullptr, *cell_items = nullptr;
                   int4* work = nullptr; int* wq = nullptr; int *gh = nullptr, *ngh = nullptr; } pb[2];
  int gpar = 0;
  bool no_pipeline = false;         // SZ_PIPELINE=0: the three-launch steps (A/B)
  int pipe_min_steps = 4;           // batches shorter than this take the three-launch steps (a pipelined batch has a prologue and an epilogue)
  int frc_first = 0;                // SZ_FRC_FIRST=n: the forcing tail of the narrow launch as n persistent workgroups in FRONT of the narrow ones (0: behind them)
  int pipe_max_floes = 60000;       // larger fields keep the three-launch steps: they are throughput-bound, nothing idles beside the narrow phase (measured at 100 k: 0.486 against 0.476 ms; SZ_PIPE_MAX_FLOES)
  int last_pipelined = 0;           // the last sz_step batch ran pipelined (sz_debug_pipelined)
  bool crec_current = false;        // the collision records of set gpar hold the parents as they lie (a pipelined batch left them so; any call that moves or
                                    // re-uploads floes outside such a batch clears it) and the twin set has the static quads: the next batch seeds neither
  bool no_reduce_free = false;      // SZ_REDUCE_FREE=0: keep the (rows-only) reduce launch inside every step (A/B)
  bool maybe_tagged = false;        // a parent may be non-active on the device (an upload said so, a batch ended on a tag, a process-mode call ran):
                                    // the next batch then runs its first step on its own (see sz_step)
  int last_err_bits = 0;   // device error bits the last sync_and_check found (tiled runs agree on them between the ranks)
  int dbg = 0;   // SZ_DEBUG bits: timing experiments only (1 skip contact rows, 2 skip direction check, 4 skip clip)
};

namespace {

#define HIPCHK(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
      return SZ_E_HIP;                                                                 \
    }                                                                                  \
  } while (0)

template <typename T>
int dalloc(sz_ctx* c, T** p, size_t n, Pool& pool) {
  const size_t bytes = (((n ? n : 1) * sizeof(T)) + 255) & ~(size_t)255;
  if (bytes > pool.left) {
    // a chunk kept from before the last reset_pool() that is large enough comes first (an upload of the same sizes
    // as the previous one then allocates nothing)
    size_t k = pool.cur ? pool.ci + 1 : 0;
    while (k < pool.chunks.size() && pool.sizes[k] < bytes) {      // too small now: will not fit later either
      (void)hipFree(pool.chunks[k]); pool.chunks.erase(pool.chunks.begin() + k); pool.sizes.erase(pool.sizes.begin() + k);
    }
    if (k < pool.chunks.size()) { pool.ci = k; pool.cur = (char*)pool.chunks[k]; pool.left = pool.sizes[k]; }
    else {
      const size_t chunk = std::max(bytes, pool.next);
      void* q = nullptr;
      HIPCHK(c, hipMalloc(&q, chunk));
      pool.chunks.push_back(q); pool.sizes.push_back(chunk); pool.ci = pool.chunks.size() - 1; pool.cur = (char*)q; pool.left = chunk;
      if (pool.next < ((size_t)256 << 20)) pool.next *= 2;
    }
    // allocations are handed out zeroed: one fill per chunk instead of one per array (~130 launches per upload)
    HIPCHK(c, hipMemsetAsync(pool.cur, 0, pool.left, c->stream));
  }
  void* q = pool.cur; pool.cur += bytes; pool.left -= bytes;
  *p = (T*)q;
  return SZ_OK;
}
void free_pool(Pool& pool) { for (void* p : pool.chunks) (void)hipFree(p); pool.chunks.clear(); pool.sizes.clear(); pool.ci = 0; pool.cur = nullptr; pool.left = 0; pool.next = 8u << 20; }
// forget the allocations, keep the memory for the next round of dalloc()s
void reset_pool(Pool& pool) { pool.ci = 0; pool.cur = nullptr; pool.left = 0; }
// after a round: chunks the round did not reach go back to the driver
void trim_pool(Pool& pool) {
  const size_t keep = pool.cur ? pool.ci + 1 : 0;
  for (size_t k = keep; k < pool.chunks.size(); k++) (void)hipFree(pool.chunks[k]);
  pool.chunks.resize(keep); pool.sizes.resize(keep);
}

struct PoolGuard { Pool v; PoolGuard() { v.next = 1u << 16; } ~PoolGuard() { free_pool(v); } };

inline int grid_for(long long n, int tpb, int maxb = 4096) {
  long long b = (n + tpb - 1) / tpb;
  if (b < 1) b = 1;
  if (b > maxb) b = maxb;
  return (int)b;
}

struct Timed {   // RAII-free helper: begin/end a timed kernel class
  sz_ctx* c; int k; size_t idx = (size_t)-1; hipStream_t st;
  Timed(sz_ctx* c_, int k_, hipStream_t st_ = nullptr) : c(c_), k(k_), st(st_ ? st_ : c_->stream) {
    if (!(c->pmask >> k & 1u)) return;
    if (c->ev_used == c->evs.size()) {
      EvPair e; e.k = k; (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b); c->evs.push_back(e);
    }
    idx = c->ev_used++;
    c->evs[idx].k = k;
    (void)hipEventRecord(c->evs[idx].a, st);
  }
  void end() { if (idx != (size_t)-1) (void)hipEventRecord(c->evs[idx].b, st); }
};
void resolve_events(sz_ctx* c) {
  for (size_t i = 0; i < c->ev_used; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->evs[i].a, c->evs[i].b) == hipSuccess) { c->kms[c->evs[i].k] += ms; c->kl[c->evs[i].k] += 1; }
  }
  c->ev_used = 0;
}

// exclusive scan of in[0..n) into out[0..n], n = cnt[ci] + add; total also to cnt[co]
constexpr int SCAN_ONE_MAX = 1 << 13;     // up to here a scan is one single-workgroup launch instead of three (measured: wins below ~5k floes)
void scan(sz_ctx* c, const int* in, int* out, int cap, int ci, int add, int co) {
  if (cap <= SCAN_ONE_MAX) { hipLaunchKernelGGL(sz_k_scan_one, dim3(1), dim3(SCAN_B), 0, c->stream, in, out, c->S.cnt, ci, add, co); return; }
  int nb = grid_for(cap, SCAN_B, 1 << 20);
  hipLaunchKernelGGL(sz_k_scan1, dim3(nb), dim3(SCAN_B), 0, c->stream, in, out, c->S.blk, c->S.cnt, ci, add);
  hipLaunchKernelGGL(sz_k_scan2, dim3(1), dim3(SCAN_B), 0, c->stream, c->S.blk, c->S.cnt, ci, add);
  hipLaunchKernelGGL(sz_k_scan3, dim3(nb), dim3(SCAN_B), 0, c->stream, in, out, c->S.blk, c->S.cnt, ci, add, co);
}

// after tiled steps: forget the ghosts and halo floes of the last one (simulation.jl:138-144; N := owned)
void tile_cleanup(sz_ctx* c) {
  if (!c->tile_dirty) return;
  hipLaunchKernelGGL(sz_k_remove_ghosts, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, 1);
  c->tile_dirty = false;
}
int sync_and_check(sz_ctx* c, int* cnt_out = nullptr) {
  tile_cleanup(c);
  int h[C_COUNT];
  HIPCHK(c, hipMemcpyAsync(h, c->S.cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  if (c->pmask) resolve_events(c);
  c->hostM = h[C_M]; c->hostN = h[C_N];
  if (cnt_out) memcpy(cnt_out, h, sizeof(h));
  c->last_err_bits = h[C_ERR];
  if (h[C_ERR]) {
    char buf[400];
    snprintf(buf, sizeof(buf),
             "device capacity/consistency error bits 0x%x (ring=1 crossings=2 regions=4 rows=8 trace=16 neighbours=32 "
             "pairs=64 elems=128 inter=256 floes=512 verts=1024 cells=2048 ghosts/parent=4096 scan=8192 halo-drift=16384 fixed-point-range=32768)", h[C_ERR]);
    c->err = buf;
    int z = 0;
    (void)hipMemcpy(c->S.cnt + C_ERR, &z, sizeof(int), hipMemcpyHostToDevice);
    return SZ_E_CAPACITY;
  }
  return SZ_OK;
}

// ---------------------------------------------------------------- lists that follow the field
// floe.interactions (rows at a stride of State::rowcap per floe): kept across an upload of the same size (see sz_upload_floes)
int carve_interactions(sz_ctx* c) {
  State& S = c->S;
  if (c->inter_capM != S.capM || c->inter_rowcap != S.rowcap || c->inter_allocs.empty()) {
    free_pool(c->inter_allocs);
    int rc;
    if ((rc = dalloc(c, &S.inter_cnt, (size_t)S.capM + 1, c->inter_allocs))) return rc;
    if ((rc = dalloc(c, &S.inter_rows, (size_t)S.capM * S.rowcap * 7, c->inter_allocs))) return rc;
    HIPCHK(c, hipMemsetAsync(S.inter_cnt, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));
    c->inter_capM = S.capM; c->inter_rowcap = S.rowcap; c->inter_lost = c->inter_any; c->inter_any = false;
  }
  return SZ_OK;
}
// neighbour lists (stride State::maxnb), the narrow phase's work list and the rows of its items (State::capPairs)
int carve_lists(sz_ctx* c) {
  State& S = c->S;
  reset_pool(c->list_allocs);
  int rc;
#define DL(field, n) if ((rc = dalloc(c, &S.field, (size_t)(n), c->list_allocs))) return rc
  DL(nb_out, (size_t)S.capM * S.maxnb); DL(nb_in, (size_t)S.capM * S.maxnb);
  DL(work, 2 * ((size_t)S.capPairs + NSEG)); DL(wq, NSEG * 32); DL(pair_i, S.capPairs); DL(pair_j, S.capPairs);
  if ((rc = dalloc(c, &c->pb[1].work, 2 * ((size_t)S.capPairs + NSEG), c->list_allocs)) || (rc = dalloc(c, &c->pb[1].wq, (size_t)NSEG * 32, c->list_allocs))) return rc;
  c->pb[0].work = S.work; c->pb[0].wq = S.wq;
  DL(it_rows, ((size_t)S.capPairs + S.capElem) * ROWS_PER_ITEM * 5); DL(it_info, (size_t)S.capM * S.maxnb + S.capElem + 1);
#undef DL
  trim_pool(c->list_allocs);
  HIPCHK(c, hipMemsetAsync(S.wq, 0, NSEG * 32 * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(c->pb[1].wq, 0, NSEG * 32 * sizeof(int), c->stream));
  return SZ_OK;
}
// The reference's lists grow as needed (collisions.jl:290-296: vcat; the Dict of the pair loop).  A call / step that outgrew a list has
// raised the matching error bit (and, inside a resident batch, paused the batch before anything of the floes' state changed): the lists
// are carved again with the next capacity and the caller runs the call / step again.  growable: nothing but list capacities overflowed.
constexpr int GROW_BITS = ERR_CAP_NEIGH | ERR_CAP_PAIRS | ERR_CAP_INTER;
bool growable(int bits) { return bits != 0 && (bits & ~GROW_BITS) == 0; }
int grow_lists(sz_ctx* c, int bits) {
  State& S = c->S;
  int maxnb = S.maxnb, rowcap = S.rowcap; long long capPairs = S.capPairs;
  if (bits & ERR_CAP_NEIGH) {
    if (maxnb >= 256) { c->err = "a floe has more than 256 bounding-circle neighbours in one direction: beyond the engine's largest neighbour capacity"; return SZ_E_CAPACITY; }
    maxnb = maxnb <= MAXNB ? 64 : 256;
    rowcap = std::max(rowcap, maxnb <= 64 ? 128 : 512);
    capPairs = std::max(capPairs, (long long)S.capM * 16);
  }
  if (bits & ERR_CAP_INTER) rowcap *= 4;
  if (bits & ERR_CAP_PAIRS) capPairs *= 2;
  const double bytes = (double)S.capM * maxnb * 16.0 + (double)capPairs * (16.0 + 8.0 + ROWS_PER_ITEM * 40.0) + (double)S.capM * rowcap * 56.0;
  if (rowcap > 8192 || capPairs > (1LL << 30) || bytes > 64e9) { c->err = "the lists a step needs have outgrown 64 GB (neighbours / pair items / interaction rows per floe)"; return SZ_E_CAPACITY; }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (getenv("SZ_VERBOSE")) fprintf(stderr, "[subzero-hip] lists grow (bits 0x%x): neighbours %d -> %d, rows per floe %d -> %d, pair items %d -> %lld\n", bits, S.maxnb, maxnb, S.rowcap, rowcap, S.capPairs, capPairs);
  S.maxnb = maxnb; S.rowcap = rowcap; S.capPairs = (int)capPairs;
  int rc = carve_lists(c);
  if (!rc) rc = carve_interactions(c);
  if (!rc) HIPCHK(c, hipMemsetAsync(S.inter_cnt, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));
  c->inter_lost = false;             // (the call that is run again provides the rows)
  return rc;
}

// ---------------------------------------------------------------- element table upload
int upload_elements(sz_ctx* c) {
  free_pool(c->static_allocs);
  State& S = c->S;
  int ntopo = (int)c->h_toff.size() > 0 ? (int)c->h_toff.size() - 1 : 0;
  int ne = 4 + ntopo;
  std::vector<int> eoff(ne + 1), ekind(ne), edir(ne);
  std::vector<double> ex, ey, eval(ne), eu(ne), ev(ne), ecx(ne), ecy(ne), erm(ne), erect(16);
  eoff[0] = 0;
  for (int k = 0; k < 4; k++) {
    const double* r = c->h_rects + 4 * k;   // xmin, xmax, ymin, ymax
    // _make_bounding_box_polygon, floe_utils.jl:104-108
    double px[5] = { r[0], r[0], r[1], r[1], r[0] }, py[5] = { r[2], r[3], r[3], r[2], r[2] };
    for (int q = 0; q < 5; q++) { ex.push_back(px[q]); ey.push_back(py[q]); }
    eoff[k + 1] = (int)ex.size();
    ekind[k] = c->h_kinds[k]; edir[k] = k; eval[k] = c->h_vals[k]; eu[k] = c->h_bu[k]; ev[k] = c->h_bv[k];
    ecx[k] = ecy[k] = erm[k] = 0.0;
    for (int q = 0; q < 4; q++) erect[4 * k + q] = r[q];
  }
  for (int t = 0; t < ntopo; t++) {
    for (int q = c->h_toff[t]; q < c->h_toff[t + 1]; q++) { ex.push_back(c->h_tx[q]); ey.push_back(c->h_ty[q]); }
    int e = 4 + t;
    eoff[e + 1] = (int)ex.size();
    ekind[e] = SZ_COLLISION; edir[e] = -1; eval[e] = 0.0; eu[e] = ev[e] = 0.0;
    ecx[e] = c->h_tcx[t]; ecy[e] = c->h_tcy[t]; erm[e] = c->h_trmax[t];
  }
  S.nelem = ne;
  c->max_elem_ring = 5;
  for (int e = 0; e < ne; e++) c->max_elem_ring = std::max(c->max_elem_ring, eoff[e + 1] - eoff[e]);
  int rc;
  if ((rc = dalloc(c, &S.eoff, ne + 1, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.ex, ex.size(), c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.ey, ey.size(), c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.ekind, ne, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.edir, ne, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.eval, ne, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.eu, ne, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.ev, ne, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.ecx, ne, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.ecy, ne, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.ermax, ne, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.erect, 16, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.eosign, ne, c->static_allocs))) return rc;
  if ((rc = dalloc(c, &S.ebb, 4 * ne, c->static_allocs))) return rc;
#define H2D(dst, src, n, T) HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)(n) * sizeof(T), hipMemcpyHostToDevice, c->stream))
  H2D(S.eoff, eoff.data(), ne + 1, int); H2D(S.ex, ex.data(), ex.size(), double); H2D(S.ey, ey.data(), ey.size(), double);
  H2D(S.ekind, ekind.data(), ne, int); H2D(S.edir, edir.data(), ne, int); H2D(S.eval, eval.data(), ne, double);
  H2D(S.eu, eu.data(), ne, double); H2D(S.ev, ev.data(), ne, double); H2D(S.ecx, ecx.data(), ne, double);
  H2D(S.ecy, ecy.data(), ne, double); H2D(S.ermax, erm.data(), ne, double); H2D(S.erect, erect.data(), 16, double);
  hipLaunchKernelGGL(sz_k_elem_osign, dim3(1), dim3(256), 0, c->stream, S);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  S.any_periodic_ew = c->h_kinds[SZ_EAST] == SZ_PERIODIC && c->h_kinds[SZ_WEST] == SZ_PERIODIC;
  S.any_periodic_ns = c->h_kinds[SZ_NORTH] == SZ_PERIODIC && c->h_kinds[SZ_SOUTH] == SZ_PERIODIC;
  S.any_domain_work = ntopo > 0;
  c->any_moving = false;
  for (int k = 0; k < 4; k++) { if (c->h_kinds[k] != SZ_PERIODIC) S.any_domain_work = 1; if (c->h_kinds[k] == SZ_MOVING) c->any_moving = true; }
  c->have_domain = true;
  return SZ_OK;
}

// launch number of the look-back scans; when the counter would no longer fit beside the status bits the flags
// are cleared once and it starts over
unsigned next_epoch(sz_ctx* c) {
  if (++c->scan_epoch >= (1u << 30)) {
    (void)hipMemsetAsync(c->S.lb_flag, 0, ((size_t)c->S.capM / SCAN_B + 8) * sizeof(unsigned), c->stream);
    c->scan_epoch = 1;
  }
  return c->scan_epoch;
}

// Grid for the resident steps: the domain box cut into cells of at least 2 max(rmax) (so that touching circles
// are in adjacent cells), indices wrapped in a periodic direction and clamped otherwise (sz_kernels.hpp, GridGeo).
void setup_grid(sz_ctx* c) {
  c->grid_ok = false; c->grid_live = false;
  const double rm = std::max(c->rmax_max, c->rmax_hint);
  if (!c->have_domain || !c->have_floes || !(rm > 0.0)) return;
  const double x0 = c->h_vals[3], xf = c->h_vals[2], y0 = c->h_vals[1], yf = c->h_vals[0];     // W, E, S, N
  if (!(xf > x0) || !(yf > y0)) return;
  // cells a hair wider than 2 max(rmax): a floe binned one cell off by round-off at a cell edge (e.g. a parent
  // wrapped by exactly one domain length after it was binned) still meets every floe whose circle touches its own
  const double cmin = 2.0 * rm * (1.0 + 1e-9);
  long long ncx = std::max(1LL, (long long)std::floor((xf - x0) / cmin));
  long long ncy = std::max(1LL, (long long)std::floor((yf - y0) / cmin));
  while (ncx * ncy > (long long)c->S.capCells) { ncx = std::max(1LL, ncx / 2); ncy = std::max(1LL, ncy / 2); }
  double* g = c->h_grid;
  g[0] = x0; g[1] = y0; g[2] = (xf - x0) / (double)ncx; g[3] = (yf - y0) / (double)ncy; g[4] = (double)ncx; g[5] = (double)ncy;
  g[6] = c->S.any_periodic_ew ? 1.0 : 0.0; g[7] = c->S.any_periodic_ns ? 1.0 : 0.0;
  c->grid_ok = true;
}
// make the static grid the live one (a process-mode call may have fitted a grid to the centroids meanwhile)
void use_static_grid(sz_ctx* c) {
  if (c->grid_live) return;
  (void)hipMemcpyAsync(c->S.bounds, c->h_grid, 8 * sizeof(double), hipMemcpyHostToDevice, c->stream);
  (void)hipMemsetAsync(c->S.cell_cnt, 0, ((size_t)c->S.capCells + 1) * sizeof(int), c->stream);
  (void)hipMemsetAsync(c->S.cell_ovf, 0, ((size_t)c->S.capCells + 1) * sizeof(int), c->stream);
  hipLaunchKernelGGL(sz_k_cell_build, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, 1);
  c->grid_live = true;
}

// resident steps of mixed precision run on body-frame rings: the world rings are rebuilt before anything else looks at them
void world_rings(sz_ctx* c) {
  if (!c->rings_stale) return;
  hipLaunchKernelGGL(sz_k_world_rings, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S);
  c->rings_stale = false;
}
// every call outside the resident steps: the candidate list they keep goes stale, the world rings must be current
void leave_resident(sz_ctx* c) { c->gl_valid = false; c->S.famrec = 0; c->crec_current = false; world_rings(c); }

// the candidate list of the coming step, seeded from the parents as they lie
void use_ghost_list(sz_ctx* c) {
  if (c->gl_valid) return;
  (void)hipMemsetAsync(c->S.cnt + C_NGCAND, 0, 2 * sizeof(int), c->stream);
  c->gl_cur = 0;
  hipLaunchKernelGGL(sz_k_ghost_seed, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, 0);
  c->gl_valid = true;
}
bool ghost_list_wanted(const sz_ctx* c, bool sg) {
  // (the list pass gives a parent one wavefront lane per ring point: rings of up to 64 points)
  return sg && !c->no_ghost_list && (c->S.any_periodic_ew || c->S.any_periodic_ns) && c->gl_est <= c->gl_max && c->max_ring <= 64;
}

// ---------------------------------------------------------------- pipeline stages
// in_step: the previous step's ghosts are dropped by the flag kernel and the commit is done by
// the bounds kernel of the broad phase (which always follows inside a step)
// commit: the flag/scan kernel commits the new counts itself (resident steps with the static grid, where no
// bounds kernel follows); otherwise the bounds kernel (in_step) or a commit launch does
// use_list: the candidate-list pass (one launch) instead of flag/scan + fill
void stage_ghosts(sz_ctx* c, bool in_step = false, bool commit = false, bool use_list = false) {
  State& S = c->S;
  // the parents' count is the host's in resident single-context steps (nothing creates or removes floes there)
  const int nh = in_step && !S.tiled ? c->hostN : -1;
  if (!S.any_periodic_ew && !S.any_periodic_ns) return;
  S.famrec = in_step ? 1 : 0;          // (process mode: ghosts the host uploaded may be present, which have no records)
  Timed t(c, SZ_K_GHOSTS);
  if (use_list) {
    const int waves = std::min(std::max(2 * c->gl_est + 64, 256), 8192);
    hipLaunchKernelGGL(sz_k_ghost_list, dim3((waves + 3) / 4), dim3(256), 0, c->stream, S, c->gl_cur, 1, nh);
    t.end();
    return;
  }
  const int nb = grid_for(S.capM, SCAN_B, 1 << 20);
  hipLaunchKernelGGL(sz_k_ghost_flag_scan, dim3(nb), dim3(SCAN_B), 0, c->stream, S, in_step ? 1 : 0, commit ? 1 : 0, next_epoch(c), nh);
  hipLaunchKernelGGL(sz_k_ghost_fill, dim3(grid_for(S.capM, 32, 2048)), dim3(256), 0, c->stream, S, commit ? 1 : 0, commit ? 1 : 0, nh);
  if (!in_step) hipLaunchKernelGGL(sz_k_ghost_commit, dim3(1), dim3(64), 0, c->stream, S);
  t.end();
}

// static_grid: the geometry in S.bounds is the host's (use_static_grid), no bounds kernel; the pair kernel does
// the housekeeping the bounds kernel would have done
void stage_broad(sz_ctx* c, bool commit_ghosts = false, bool static_grid = false, bool fuse_forcing = false, bool with_elems = false) {
  State& S = c->S;
  Timed t(c, SZ_K_BROAD);
  int gM = grid_for(S.capM, 256);
  if (!static_grid) {          // with the static grid the cells are already current (see sz_k_cell_build)
    hipLaunchKernelGGL(sz_k_bounds, dim3(1), dim3(1024), 0, c->stream, S, commit_ghosts ? 1 : 0);
    hipLaunchKernelGGL(sz_k_cell_build, dim3(gM), dim3(256), 0, c->stream, S, 0);
    c->grid_live = false;
  }
  // the neighbour search appends the pair items to the narrow phase's work list itself: no scan, no pair-list launch
  const bool rec = S.crec != nullptr;      // collision records are current in this batch: the instantiations that read them
  if (fuse_forcing) {          // the step's forcings ride in the neighbour launch (sz_k_neighbors_forcing)
    const int nbn = grid_for(S.capM, 256 / NB_G, 8192), nbf = grid_for(S.capM, 256 / FRC_PLAIN, 8192);
    if (c->precision == 1) { if (rec) hipLaunchKernelGGL((sz_k_neighbors_forcing<2, true>), dim3(nbn + nbf), dim3(256), 0, c->stream, S, c->P, nbn); else hipLaunchKernelGGL((sz_k_neighbors_forcing<2, false>), dim3(nbn + nbf), dim3(256), 0, c->stream, S, c->P, nbn); }
    else { if (rec) hipLaunchKernelGGL((sz_k_neighbors_forcing<1, true>), dim3(nbn + nbf), dim3(256), 0, c->stream, S, c->P, nbn); else hipLaunchKernelGGL((sz_k_neighbors_forcing<1, false>), dim3(nbn + nbf), dim3(256), 0, c->stream, S, c->P, nbn); }
  } else
  {
    const dim3 gr(grid_for(S.capM, NB_TPB / NB_G, 8192)), bl(NB_TPB);
    const bool fam = c->hostN <= 40000 && (S.any_periodic_ew || S.any_periodic_ns);      // (no periodic wall: no ghosts, no Dict rule)
    if (with_elems) {          // the element items in the launch's tail (elems_ride)
      // (no periodic wall, no ghosts, no Dict rule: the search's lean instantiation)
      const int nbn = (int)gr.x, nbe = grid_for(S.capM, NB_TPB, 1 << 20);
      if (rec) hipLaunchKernelGGL((sz_k_neighbors_elem<false, true>), dim3(nbn + nbe), bl, 0, c->stream, S, next_epoch(c), nbn);
      else hipLaunchKernelGGL((sz_k_neighbors_elem<false, false>), dim3(nbn + nbe), bl, 0, c->stream, S, next_epoch(c), nbn);
    } else if (S.maxnb <= MAXNB) {
      if (fam) { if (rec) hipLaunchKernelGGL((sz_k_neighbors<true, MAXNB, true>), gr, bl, 0, c->stream, S); else hipLaunchKernelGGL((sz_k_neighbors<true, MAXNB>), gr, bl, 0, c->stream, S); }
      else { if (rec) hipLaunchKernelGGL((sz_k_neighbors<false, MAXNB, true>), gr, bl, 0, c->stream, S); else hipLaunchKernelGGL((sz_k_neighbors<false, MAXNB>), gr, bl, 0, c->stream, S); }
    } else if (S.maxnb <= 64) {
      if (fam) hipLaunchKernelGGL((sz_k_neighbors<true, 64>), gr, bl, 0, c->stream, S);
      else hipLaunchKernelGGL((sz_k_neighbors<false, 64>), gr, bl, 0, c->stream, S);
    } else {          // (a floe with more than 64 neighbours: the capacity that keeps such a field running)
      hipLaunchKernelGGL((sz_k_neighbors<false, 256>), dim3(grid_for(S.capM, 64 / NB_G, 16384)), dim3(64), 0, c->stream, S);
    }
  }
  t.end();
}

void stage_elems(sz_ctx* c, bool enabled) {
  State& S = c->S;
  if (!enabled || !S.any_domain_work) {
    // el_off stays all-zero (allocated zeroed and never written in this mode)
    if (!S.any_domain_work) return;              // C_NELEM is 0 since the upload and nothing ever changes it
    hipLaunchKernelGGL(sz_k_zero_int, dim3(1), dim3(64), 0, c->stream, S.cnt + C_NELEM, S.cnt, -1, 1);
    if (S.any_domain_work)
      hipLaunchKernelGGL(sz_k_zero_int, dim3(grid_for(S.capM + 1, 256)), dim3(256), 0, c->stream, S.el_off, S.cnt, C_M, 1);
    return;
  }
  Timed t(c, SZ_K_BROAD);
  hipLaunchKernelGGL(sz_k_elem_scan_fill, dim3(grid_for(S.capM, SCAN_B, 1 << 20)), dim3(SCAN_B), 0, c->stream, S, next_epoch(c));
  t.end();
}

// frc: the step's forcings ride in the launch of the first variant (0: no, 1: fp64, 2: mixed precision)
// rings above the first narrow variant's capacity exist (rings never change size inside the hot path, so the host knows; halo floes of a
// tiled run arrive unseen: their bound counts)
bool larger_rings(const sz_ctx* c) {
  return std::max(std::max(c->max_ring, c->max_elem_ring), c->S.tiled ? c->max_ring_tiled : 0) > NARROW_CAP0;
}
// parts: 0 everything (the largest variant is always enqueued: it takes the items the others hand on), 1 without the largest variant unless
// rings that need it exist (sz_step's retry_stop mode), 2 only the larger variants (the rest of a paused step)
void stage_narrow(sz_ctx* c, int dt, double ffmo, double fdmo, bool housekept = false, int frc = 0, int parts = 0) {
  State& S = c->S;
  // dynamic rounds (see sz_k_narrow) where the queue heads were just cleared (static-grid steps); SZ_NARROW_QUEUE=0: off
  const int queue = c->no_queue ? 0 : 1;      // (the reduce kernel resets the queue heads after every narrow phase)
  (void)housekept;
  long long capItems = (long long)S.capPairs + S.capElem;
  // Rings never change size inside the hot path, so the host knows whether any item can need a
  // larger variant (halo floes of a tiled run arrive unseen: then always check on the device).
  const bool larger = larger_rings(c);
  if (larger && parts != 2) hipLaunchKernelGGL(sz_k_items_clear, dim3(grid_for(capItems, 256)), dim3(256), 0, c->stream, S);
  if (parts != 2) {
    Timed t(c, SZ_K_NARROW);
    constexpr int G = NARROW_G, TPB = 64;
    // 160 VGPRs (3 wavefronts per SIMD) and 16 KB of LDS per workgroup: 10 workgroups = 80 items in flight per CU
    auto kern = sz_k_narrow<G, NARROW_CAP0, NARROW_KC0, NARROW_RC0, 4, TPB, 0, 0, 3, 0>;
    int& grid = c->narrow_grid0;
    if (grid == 0) {       // as many workgroups as the chip holds at once, so that every one of them runs the same number of rounds
      int per_cu = 0, cus = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, TPB, 0);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
      grid = per_cu > 0 && cus > 0 ? per_cu * cus : 2048;
      if (const char* e = getenv("SZ_NARROW_GRID")) { int v = atoi(e); if (v > 0) grid = v; }
      if (getenv("SZ_VERBOSE")) fprintf(stderr, "[subzero-hip] narrow: %d workgroups per CU x %d CUs\n", per_cu, cus);
    }
    const int nbn = grid_for(capItems, TPB / G, grid);
    int nbf = frc ? grid_for(S.capM, TPB / FRC_PLAIN, 32768) : 0;
    const int nbfg = c->frc_first > 0 && nbf > 0 ? std::min(c->frc_first, nbf) : nbf;      // (workgroups in the grid)
    if (c->frc_first > 0 && nbf > 0) nbf = -nbfg;
    if (frc == 1) hipLaunchKernelGGL((sz_k_narrow<G, NARROW_CAP0, NARROW_KC0, NARROW_RC0, 4, TPB, 0, 0, 3, 1>), dim3(nbn + nbfg), dim3(TPB), 0, c->stream, S, c->P, dt, ffmo, fdmo, c->dbg, queue, nbf, PipeAlt{}, 0, 0);
    else if (frc == 2) hipLaunchKernelGGL((sz_k_narrow<G, NARROW_CAP0, NARROW_KC0, NARROW_RC0, 4, TPB, 0, 0, 3, 2>), dim3(nbn + nbfg), dim3(TPB), 0, c->stream, S, c->P, dt, ffmo, fdmo, c->dbg, queue, nbf, PipeAlt{}, 0, 0);
    else hipLaunchKernelGGL(kern, dim3(nbn), dim3(TPB), 0, c->stream, S, c->P, dt, ffmo, fdmo, c->dbg, queue, 0, PipeAlt{}, 0, 0);
    if (c->dbg & 8)       // timing experiment: the same launch again (same results) -- how much of a launch is a cold instruction cache?
      hipLaunchKernelGGL(kern, dim3(nbn), dim3(TPB), 0, c->stream, S, c->P, dt, ffmo, fdmo, c->dbg, queue, 0, PipeAlt{}, 0, 0);
    t.end();
  }
  {
    // larger working sets: items with larger rings (only if such rings can exist) and items the
    // smaller variant handed on; both kernels return at once when the step has no such item
    Timed t(c, K_NARROW_LARGE);
    if (parts == 1 && !larger) { t.end(); return; }
    if (larger)
      hipLaunchKernelGGL((sz_k_narrow<16, NARROW_CAP1, 16, 80, 6, 64, NARROW_CAP0, 1>), dim3(grid_for(capItems, 4, 2048)), dim3(64), 0,
                         c->stream, S, c->P, dt, ffmo, fdmo, c->dbg, queue, 0, PipeAlt{}, 0, 0);
    hipLaunchKernelGGL((sz_k_narrow<64, NARROW_CAP2, NARROW_KC2, NARROW_RC2, 16, 64, NARROW_CAP1, 2>), dim3(grid_for(capItems, 1, larger ? 2048 : 256)), dim3(64), 0,
                       c->stream, S, c->P, dt, ffmo, fdmo, c->dbg, queue, 0, PipeAlt{}, 0, 0);
    t.end();
  }
}

// m_hint: see sz_k_inter_fill (resident steps: the parents + the ghosts the last look at the device showed, and some)
// the force scale of the fixed-point totals (sz_geom.hpp fx_force_exp): |row force| <= (1 + mu) E h sqrt(area)
int force_scale_exp(const sz_ctx* c) { const double b = (1.0 + std::max(c->P.mu, 0.0)) * c->P.E; return (b > 0.0 && b < 1e300 ? std::ilogb(b) : 0) + 1; }
// behind: the launch that assembles the rows of a reduce-free batch's last step (parents' centroids of that step from `mot`)
void stage_reduce(sz_ctx* c, int mirror, int n_init, int dt, int m_hint = 0, bool behind = false) {
  State& S = c->S;
  Timed t(c, SZ_K_REDUCE);
  if (behind || c->reduce_mode != 2)
    hipLaunchKernelGGL(sz_k_inter_fill, dim3(grid_for(S.capM, 128 / IF_G, 16384)), dim3(128), 0, c->stream, S, mirror, n_init, m_hint, behind || c->reduce_mode == 1 ? 1 : 0, behind ? 1 : 0);
  if (!behind && mirror && c->any_moving) hipLaunchKernelGGL(sz_k_update_boundaries, dim3(1), dim3(64), 0, c->stream, S, dt);
  t.end();
}

// a step of sz_step: `resume` = the rest of a step that paused after its narrow launch (see stopped_late())
void collisions_step(sz_ctx* c, int n_init, int dt, bool commit_ghosts, bool static_grid, int fuse_forcing, bool lean, bool resume) {
  if (!resume) {
    // resident steps of a field between walls: the element items are made in the tail of the neighbour search's launch
    const bool ride = static_grid && fuse_forcing != 1 && c->S.any_domain_work && !c->S.any_periodic_ew && !c->S.any_periodic_ns &&
                      c->S.maxnb <= MAXNB && !c->no_elems_ride && !c->S.tiled;      // (a tile's neighbour launch commits the halo rows' count: the scan
                                                                                        //  would read it while it changes)
    stage_broad(c, commit_ghosts, static_grid, fuse_forcing == 1, ride);
    if (!ride) stage_elems(c, true);
  }
  stage_narrow(c, dt, c->P.ff_max_overlap, c->P.fd_max_overlap, static_grid, fuse_forcing == 2 ? (c->precision == 1 ? 2 : 1) : 0, resume ? 2 : lean ? 1 : 0);
  // (rows the step can hold at most, for the reduce's first batch of loads: a tile's halo floes are bounded by the slots of its receive regions,
  //  and any of them may bring up to three ghosts)
  int halo = 0;
  if (c->S.tiled) for (size_t r = 0; r < c->cap_recv.size(); r++) halo += std::max(c->cap_recv[r], 0);
  stage_reduce(c, 1, n_init, dt, c->hostN + halo + 3 * (c->gl_est + halo) + c->hostN / 64 + 32);
}
// fuse_forcing: the step's forcings ride in another launch: 1 the neighbour search's, 2 the narrow phase's
void collisions(sz_ctx* c, int n_init, int dt, bool commit_ghosts = false, bool static_grid = false, int fuse_forcing = 0) {
  stage_broad(c, commit_ghosts, static_grid, fuse_forcing == 1);
  stage_elems(c, true);
  stage_narrow(c, dt, c->P.ff_max_overlap, c->P.fd_max_overlap, static_grid, fuse_forcing == 2 ? (c->precision == 1 ? 2 : 1) : 0);
  stage_reduce(c, 1, n_init, dt);
}

// The forcings only read the floes' state at the start of the step and write fxOA/fyOA/trqOA/
// hflx_factor, which nothing but the integrator reads: inside a step they run on a second stream
// BESIDE the ghost / broad / narrow / reduce kernels (all of them latency-bound, the chip is far
// from full) and join before the integrator.
void stage_forcing_fork(sz_ctx* c, const State* Sp = nullptr) {
  const State& S = Sp ? *Sp : c->S;
  (void)hipEventRecord(c->ev_fork, c->stream);
  (void)hipStreamWaitEvent(c->stream2, c->ev_fork, 0);
  Timed t(c, SZ_K_FORCING, c->stream2);
  if (c->precision == 1) hipLaunchKernelGGL(sz_k_forcing_mixed, dim3(grid_for(c->S.capM, 256 / FRC_PLAIN, 8192)), dim3(256), 0, c->stream2, S, c->P);
  else hipLaunchKernelGGL(sz_k_forcing<false>, dim3(grid_for(c->S.capM, 256 / FRC_PLAIN, 8192)), dim3(256), 0, c->stream2, S, c->P, 0);
  t.end();
  (void)hipEventRecord(c->ev_join, c->stream2);
}
void stage_forcing_join(sz_ctx* c) { (void)hipStreamWaitEvent(c->stream, c->ev_join, 0); }
// the blocked copy of the sub-floe points the one-way fp64 forcing loop reads (State::sxy): made when the points or the lattice spacing changed
// (SZ_BLOCK_POINTS=0: never -- the loop then reads sx / sy in the caller's order, A/B switch)
int ensure_block_points(sz_ctx* c) {
  State& S = c->S;
  if (c->blk_pts_ok || c->no_block_points || !c->have_fields) return SZ_OK;
  int rc;
  free_pool(c->blk_pt_allocs);
  S.sxy = nullptr;
  if ((rc = dalloc(c, &S.sxy, (size_t)std::max(S.capS, 1), c->blk_pt_allocs))) return rc;
  const double q = std::min(S.gdx, S.gdy) / 4.0;
  if (c->pts_N > 0) hipLaunchKernelGGL(sz_k_block_points, dim3(grid_for(c->pts_N, 1, 1 << 16)), dim3(64), 0, c->stream, S, c->pts_N, q > 0 ? 1.0 / q : 1.0);
  c->blk_pts_ok = true;
  return SZ_OK;
}
// fp32 copies of the sub-floe points and of the lattice for the mixed-precision forcing kernel
int ensure_mixed(sz_ctx* c) {
  State& S = c->S;
  int rc;
  if (!c->mixed_pts_ok) {
    free_pool(c->mixed_pt_allocs);
    if ((rc = dalloc(c, &S.s32, (size_t)std::max(S.capS, 1), c->mixed_pt_allocs))) return rc;
    hipLaunchKernelGGL(sz_k_to_f32_points, dim3(grid_for(S.capS, 256)), dim3(256), 0, c->stream, S, S.capS);
    c->mixed_pts_ok = true;
  }
  if (!c->mixed_geom_ok) {
    world_rings(c);                 // (the body rings are made from the world rings)
    free_pool(c->mixed_geom_allocs);
    if ((rc = dalloc(c, &S.rec32, (size_t)2 * S.capM, c->mixed_geom_allocs)) || (rc = dalloc(c, &S.ring32, (size_t)std::max(S.capV, 1), c->mixed_geom_allocs)) ||
        (rc = dalloc(c, &S.rb_off, (size_t)S.capM, c->mixed_geom_allocs)) || (rc = dalloc(c, &S.rb_n, (size_t)S.capM, c->mixed_geom_allocs))) return rc;
    hipLaunchKernelGGL(sz_k_rec32_seed, dim3(grid_for(S.capM, 256)), dim3(256), 0, c->stream, S);
    hipLaunchKernelGGL(sz_k_body_rings, dim3(grid_for(S.capM, 256)), dim3(256), 0, c->stream, S);
    c->mixed_geom_ok = true;
  }
  if (!c->mixed_nodes_ok) {
    free_pool(c->mixed_node_allocs);
    const size_t n = (size_t)(S.Nx + 1) * (S.Ny + 1) * 8;
    if ((rc = dalloc(c, &S.nodes32, n, c->mixed_node_allocs))) return rc;
    hipLaunchKernelGGL(sz_k_to_f32_nodes, dim3(grid_for((long long)n, 256)), dim3(256), 0, c->stream, S);
    c->mixed_nodes_ok = true;
  }
  return SZ_OK;
}
// buffers of the two-way coupling: per-floe cell slots follow the floe capacity, per-cell arrays the lattice
int ensure_two_way(sz_ctx* c) {
  State& S = c->S;
  const size_t ncell = (size_t)(S.Nx + 1) * (S.Ny + 1);
  int rc;
  if (c->tw_field_allocs.empty() || c->tw_ncell != ncell) {
    free_pool(c->tw_field_allocs);
    if ((rc = dalloc(c, &S.t_ocn, ncell, c->tw_field_allocs)) || (rc = dalloc(c, &S.t_atm, ncell, c->tw_field_allocs)) ||
        (rc = dalloc(c, &S.tau_x, ncell, c->tw_field_allocs)) || (rc = dalloc(c, &S.tau_y, ncell, c->tw_field_allocs)) ||
        (rc = dalloc(c, &S.si_frac, ncell, c->tw_field_allocs)) || (rc = dalloc(c, &S.cl_cnt, ncell + 1, c->tw_field_allocs)) ||
        (rc = dalloc(c, &S.cl_off, ncell + 2, c->tw_field_allocs)) || (rc = dalloc(c, &S.cl_cur, ncell + 1, c->tw_field_allocs)))
      return rc;
    c->tw_ncell = ncell;
  }
  if (c->have_floes && (c->tw_allocs.empty() || c->tw_capM != S.capM)) {
    free_pool(c->tw_allocs);
    const size_t ne = (size_t)S.capM * FC_CAP;
    if ((rc = dalloc(c, &S.fc_key, ne, c->tw_allocs)) || (rc = dalloc(c, &S.fc_n, ne, c->tw_allocs)) ||
        (rc = dalloc(c, &S.fc_cnt, (size_t)S.capM, c->tw_allocs)) || (rc = dalloc(c, &S.fc_code, ne, c->tw_allocs)) ||
        (rc = dalloc(c, &S.fc_tx, ne, c->tw_allocs)) || (rc = dalloc(c, &S.fc_ty, ne, c->tw_allocs)) ||
        (rc = dalloc(c, &S.fc_area, ne, c->tw_allocs)) || (rc = dalloc(c, &S.cl_ent, ne, c->tw_allocs)))
      return rc;
    c->tw_capM = S.capM;
  }
  return SZ_OK;
}
void stage_forcing(sz_ctx* c, int dt = -1) {      // in-order variant (process mode, profiling)
  Timed t(c, SZ_K_FORCING);
  if (!c->two_way && c->precision == 1) {
    hipLaunchKernelGGL(sz_k_forcing_mixed, dim3(grid_for(c->S.capM, 256 / FRC_PLAIN, 8192)), dim3(256), 0, c->stream, c->S, c->P);
  } else if (!c->two_way) {
    hipLaunchKernelGGL(sz_k_forcing<false>, dim3(grid_for(c->S.capM, 256 / FRC_PLAIN, 8192)), dim3(256), 0, c->stream, c->S, c->P, 0);
  } else {
    // timestep_coupling! with two_way_coupling_on (coupling.jl:1705-1738): one-way forcings + per-floe cell slots,
    // then calc_two_way_coupling! (:1617-1680) as a counting sort by cell, one clip per (floe, cell) entry, a reduction
    State& S = c->S;
    const int ncell = (int)c->tw_ncell;
    // LDS for the largest floe's points only (capacity TW_PMAX): the kernel is bound by the floes in flight per CU
    const int pmax = std::min(TW_PMAX, std::max(32, (c->max_sub + 31) / 32 * 32));
    hipLaunchKernelGGL(sz_k_forcing<true>, dim3(grid_for(S.capM, TW_FPB, 16384)), dim3(TW_FPB * FRC_G), tw_forcing_lds(pmax), c->stream, S, c->P, pmax);
    const int ge = grid_for((long long)S.capM * FC_CAP, 256, 8192);
    hipLaunchKernelGGL(sz_k_tw_count, dim3(ge), dim3(256), 0, c->stream, S);
    scan(c, S.cl_cnt, S.cl_off, ncell, -1, ncell, C_NENT);
    hipLaunchKernelGGL(sz_k_tw_fill, dim3(ge), dim3(256), 0, c->stream, S);
    hipLaunchKernelGGL(sz_k_tw_sort, dim3(grid_for(ncell, 256)), dim3(256), 0, c->stream, S, ncell);
    if (c->tw_general_clip) hipLaunchKernelGGL(sz_k_tw_area, dim3(grid_for((long long)S.capM * FC_CAP, 64 / TW_G, 4096)), dim3(64), 0, c->stream, S);
    else hipLaunchKernelGGL(sz_k_tw_area_rect, dim3(grid_for((long long)S.capM * 16, 256, 8192)), dim3(256), 0, c->stream, S);
    // tiled runs finish the cells after the partial sums of all ranks have been added up (sz_two_way_partial / _finish)
    if (!S.tiled) hipLaunchKernelGGL(sz_k_tw_reduce, dim3(grid_for(ncell, 256)), dim3(256), 0, c->stream, S, c->P, ncell, dt >= 0 ? dt : c->tw_dt);
  }
  t.end();
}
// gl_fill: ghost-candidate list the integrator appends to (resident steps on the list path), -1: none
void stage_integrate(sz_ctx* c, int dt, bool reset_guards, bool apply_frc, bool bin = false, int gl_fill = -1, int ginl = -1, const PackInl* pack = nullptr) {
  const int nh = bin && !c->S.tiled ? c->hostN : -1;     // resident single-context steps: the host knows the count
  // the guard counters describe the last timestep_floe_properties! call (inside a step the
  // ghost-removal kernel has already cleared them)
  if (reset_guards) (void)hipMemsetAsync(c->S.warn, 0, (size_t)WARN_SLOTS * 32 * sizeof(int), c->stream);
  Timed t(c, SZ_K_INTEGRATE);
  // resident steps with small rings: one launch (thread per floe) integrates, moves the ring and bins the floe
  if (bin && c->max_ring <= MV_RING && c->fused_move) {
    // (tiled steps: the same thread also writes the floe's halo records for the next step -- sz_k_integrate<true, true>)
    if (pack) hipLaunchKernelGGL((sz_k_integrate<true, true>), dim3(grid_for(c->S.capM, 128)), dim3(128), 0, c->stream, c->S, c->P, dt, apply_frc ? 1 : 0, 1, nh, gl_fill, ginl, *pack, c->acc_mode);
    else hipLaunchKernelGGL(sz_k_integrate<true>, dim3(grid_for(c->S.capM, 128)), dim3(128), 0, c->stream, c->S, c->P, dt, apply_frc ? 1 : 0, 1, nh, gl_fill, ginl, PackInl{}, c->acc_mode);
  } else {
    hipLaunchKernelGGL(sz_k_integrate<false>, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, c->P, dt, apply_frc ? 1 : 0, 0, nh, gl_fill, -1, PackInl{}, c->acc_mode);
    hipLaunchKernelGGL(sz_k_move_strain, dim3(grid_for(c->S.capM, 16, 8192)), dim3(256), 0, c->stream, c->S, 0, bin ? 1 : 0, gl_fill);
  }
  if (!bin) c->grid_live = false;          // floes moved without re-binning: the resident steps' cell lists are stale
  t.end();
}

// inline ghosts: the keys of the last step's ghosts from the device, and each ghost's number in the reference's order
int gi_fetch(sz_ctx* c) {
  if (!c->gi_pending) return SZ_OK;
  const int G = c->gi_pending_n;
  c->gi_keys.assign(G, 0);
  if (G > 0) HIPCHK(c, hipMemcpy(c->gi_keys.data(), c->S.gkeys + (size_t)c->gi_pending_slot * c->S.capM, (size_t)G * sizeof(long long), hipMemcpyDeviceToHost));
  std::vector<int> ord(G);
  for (int k = 0; k < G; k++) ord[k] = k;
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return c->gi_keys[a] < c->gi_keys[b]; });
  c->gi_ref.assign(G, 0);
  for (int r = 0; r < G; r++) c->gi_ref[ord[r]] = r;
  c->gi_pending = false;
  return SZ_OK;
}
// exact host replay of the fuse bookkeeping (collisions.jl:367-368 and :801-806) for the rare
// steps in which a pair exceeded max_overlap
// after_step (sz_step): the ghosts of the step have already been detached (C_M == N; their rows and the pair arrays are
// still in place) and the integrator has run since: a floe the coupling marked for removal stays `remove`
// (timestep_coupling! follows timestep_collisions! in timestep_sim!, simulation.jl:109-161)
// tkeys (tiled contexts): the order key of every local row -- owned floes, then the step's halo floes and ghosts in allocation order; the
// replay then walks the rows in the order of their keys (= the single context's floe order) and the lists come back indexed by STORAGE row
// with partners named by storage row (tile_fuse_global turns those into global floe numbers)
int host_fuse_fixup(sz_ctx* c, const int* h, bool mirror, bool after_step = false, bool coupled = false, const std::vector<long long>* tkeys = nullptr) {
  State& S = c->S;
  int M = after_step ? h[C_N] + h[C_NGHOSTS] : h[C_M];
  // A resident batch's lists describe the step that ended it, and only that step: whatever earlier batches left in them is dropped.  (The
  // reference's simplify_floes! consumes status.fuse_idx after every step; batches that run on past a fuse -- SZ_NO_STOP, measurement runs --
  // would otherwise replay lists that name the ghost numbers of older steps.)
  if (tkeys || after_step) c->fuse_lists.assign(M, {});
  if ((int)c->fuse_lists.size() < M) c->fuse_lists.resize(M);
  if (h[C_NFUSE] == 0 || M == 0) return SZ_OK;          // no pair asked for a fuse (the narrow phase counts them)
  // the pairs in the reference's serial order (i asc, j asc): per floe its sorted list of owned pairs
  const int MAXNB = S.maxnb;
  std::vector<int> nout(M), nbo((size_t)M * MAXNB); std::vector<int2> info((size_t)M * MAXNB);
  HIPCHK(c, hipMemcpy(nout.data(), S.n_out, (size_t)M * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(nbo.data(), S.nb_out, (size_t)M * MAXNB * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(info.data(), S.it_info, (size_t)M * MAXNB * sizeof(int2), hipMemcpyDeviceToHost));
  std::vector<int> tag(M);
  HIPCHK(c, hipMemcpy(tag.data(), c->S.tagA, (size_t)M * sizeof(int), hipMemcpyDeviceToHost));
  // inline ghosts lie in allocation order: the replay walks the floes in the reference's order and the lists hold its numbers
  // (ref[storage index] / sto[reference number]; the identity otherwise)
  const int Np = h[C_N];
  if (after_step && c->gi_valid) { int rc = gi_fetch(c); if (rc) return rc; }
  const bool renum = after_step && c->gi_valid && (int)c->gi_ref.size() == M - Np;
  std::vector<int> ref(M), sto(M);
  for (int i = 0; i < M; i++) { ref[i] = i < Np || !renum ? i : Np + c->gi_ref[i - Np]; sto[ref[i]] = i; }
  if (tkeys && (int)tkeys->size() == M) {
    for (int i = 0; i < M; i++) sto[i] = i;
    std::sort(sto.begin(), sto.end(), [&](int a, int b) { return (*tkeys)[a] < (*tkeys)[b]; });
    for (int r = 0; r < M; r++) ref[sto[r]] = r;
  }
  for (int ii = 0; ii < M; ii++) {
    const int i = sto[ii];
    for (int r = 0; r < nout[i]; r++)
      if ((info[(size_t)i * MAXNB + r].x >> 8) & IT_FUSE) { const int pj = nbo[(size_t)i * MAXNB + r]; if (pj >= 0 && pj < M) c->fuse_lists[ii].push_back(ref[pj]); }
  }
  if (mirror) {
    for (int ii = 0; ii < M; ii++) {
      if (tag[sto[ii]] != SZ_FUSE) continue;
      size_t n = c->fuse_lists[ii].size();
      for (size_t k = 0; k < n; k++) {
        const int idx = c->fuse_lists[ii][k];
        if (idx < 0 || idx >= M) continue;          // (a partner number of a step whose ghosts are gone: process-mode lists accumulate like the reference's)
        tag[sto[idx]] = SZ_FUSE; c->fuse_lists[idx].push_back(ii);
      }
    }
  }
  if (after_step && coupled) {
    std::vector<int> rm(h[C_N]);
    HIPCHK(c, hipMemcpy(rm.data(), S.frc_remove, (size_t)h[C_N] * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < h[C_N]; i++) if (rm[i]) tag[i] = SZ_REMOVE;
  }
  HIPCHK(c, hipMemcpy(S.status, tag.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice));
  if (tkeys && (int)tkeys->size() == M) {          // back to storage rows (index and values)
    std::vector<std::vector<int>> byrow(M);
    for (int r = 0; r < M; r++) { byrow[sto[r]] = c->fuse_lists[r]; for (int& v : byrow[sto[r]]) v = v >= 0 && v < M ? sto[v] : v; }
    c->fuse_lists.swap(byrow);
  }
  return SZ_OK;
}

}  // namespace

// =====================================================================================================
extern "C" {

const char* sz_version(void) { return "subzero-hip 0.1 (gfx950)"; }

// SZ_BACKTRACE=1 (diagnosis of a host-side crash on a box without a debugger): SIGSEGV / SIGABRT print the C frames (module + offset: resolve with
// addr2line -e libsubzero_hip.so <offset>) before the process dies
static void sz_crash_handler(int sig) {
  signal(SIGALRM, SIG_DFL); alarm(2);          // (a handler that gets stuck -- the heap's lock may be held -- still ends the process)
  void* fr[64];
  const int n = backtrace(fr, 64);
  const char msg[] = "[subzero-hip] fatal signal, C frames:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(fr, n, 2);
  signal(sig, SIG_DFL); raise(sig);
}
sz_ctx* sz_create(int device_id) {
  if (getenv("SZ_BACKTRACE")) { void* warm[4]; (void)backtrace(warm, 4); signal(SIGSEGV, sz_crash_handler); signal(SIGABRT, sz_crash_handler); }      // (the first backtrace() loads its library: not inside a handler)
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id >= n) return nullptr;
  if (hipSetDevice(device_id) != hipSuccess) return nullptr;
  sz_ctx* c = new sz_ctx();
  c->device = device_id;
  if (const char* e = getenv("SZ_DEBUG")) c->dbg = atoi(e);
  if (const char* e = getenv("SZ_OVERLAP")) c->overlap_forcing = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("SZ_NARROW_QUEUE")) c->no_queue = atoi(e) == 0;
  if (const char* e = getenv("SZ_FUSED_MOVE")) c->fused_move = atoi(e) != 0;
  if (const char* e = getenv("SZ_TW_GENERAL_CLIP")) c->tw_general_clip = atoi(e) != 0;
  if (const char* e = getenv("SZ_LEAN_NARROW")) c->no_lean_narrow = atoi(e) == 0;
  if (const char* e = getenv("SZ_CREC")) c->no_crec = atoi(e) == 0;
  if (const char* e = getenv("SZ_REDUCE_FREE")) c->no_reduce_free = atoi(e) == 0;
  if (const char* e = getenv("SZ_PIPELINE")) c->no_pipeline = atoi(e) == 0;
  if (const char* e = getenv("SZ_TILE_HEADERS")) c->tile_hdr_neighbours = strcmp(e, "neighbours") == 0;
  if (const char* e = getenv("SZ_PIPE_MIN_STEPS")) c->pipe_min_steps = std::max(2, atoi(e));
  if (const char* e = getenv("SZ_PIPE_MAX_FLOES")) c->pipe_max_floes = atoi(e);
  if (const char* e = getenv("SZ_FRC_FIRST")) c->frc_first = std::max(0, atoi(e));
  if (const char* e = getenv("SZ_BLOCK_POINTS")) c->no_block_points = atoi(e) == 0;
  if (const char* e = getenv("SZ_TILE_FORCING_TAIL")) c->tile_forcing_in_tail = atoi(e) != 0;
  if (const char* e = getenv("SZ_GHOST_INLINE")) c->ghost_inline = atoi(e) != 0;
  if (const char* e = getenv("SZ_TILE_INLINE")) c->tile_inline_off = atoi(e) == 0;
  if (const char* e = getenv("SZ_ELEMS_RIDE")) c->no_elems_ride = atoi(e) == 0;
  if (const char* e = getenv("SZ_FUSE_FORCING")) { c->fuse_forcing = atoi(e) != 0; if (atoi(e) > 0) c->fuse_forcing_mode = atoi(e) >= 2 ? 2 : 1; }
  if (const char* e = getenv("SZ_STATIC_GRID")) c->no_static_grid = atoi(e) == 0;
  if (const char* e = getenv("SZ_BODY_RINGS")) c->no_body_rings = atoi(e) == 0;
  if (const char* e = getenv("SZ_XCD")) c->S.xcd_neigh = atoi(e) != 0 ? 1 : 0;
  c->S.xcd_forcing = 0;          // measured: no change at 10 k floes, 159 -> 203 us at 100 k
  if (const char* e = getenv("SZ_XCD_FORCING")) c->S.xcd_forcing = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("SZ_GHOST_LIST")) { c->no_ghost_list = atoi(e) == 0; if (atoi(e) > 1) c->gl_max = atoi(e); }
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);     // lo = least urgent, hi = most urgent
  if (hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prio_hi) != hipSuccess) { delete c; return nullptr; }
  if (hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_lo) != hipSuccess) { delete c; return nullptr; }
  (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming); (void)hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
  // Constants() and default settings of the reference
  Params& P = c->P;
  P.E = 6e6; P.nu = 0.3; P.mu = 0.2; P.rho_o = 1027.0; P.rho_a = 1.2; P.Cd_io = 3e-3; P.Cd_ia = 1e-3;
  P.fcor = 1.4e-4; P.turn = 15.0 * 3.14159265358979323846 / 180.0; P.ff_max_overlap = 0.55; P.fd_max_overlap = 0.75;
  P.rho_i = 920.0; P.max_h = 10.0; P.max_xi = 1e-5; P.lambda = 0.2; P.dd = 1;
  P.Cd_ao = 1.25e-3; P.k_ice = 2.14; P.L_ice = 2.93e5;
  if (hipMalloc((void**)&c->d_stats, 20 * sizeof(long long)) != hipSuccess) { delete c; return nullptr; }
  if (hipMalloc((void**)&c->S.acc, (size_t)ACC_SLOTS * 8 * sizeof(unsigned long long)) != hipSuccess) { delete c; return nullptr; }
  (void)hipMemset(c->S.acc, 0, (size_t)ACC_SLOTS * 8 * sizeof(unsigned long long));
  return c;
}

void sz_destroy(sz_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  free_pool(c->allocs); free_pool(c->list_allocs); free_pool(c->inter_allocs); free_pool(c->static_allocs); free_pool(c->field_allocs); free_pool(c->tw_allocs); free_pool(c->tw_field_allocs);
  free_pool(c->blk_pt_allocs); free_pool(c->mixed_pt_allocs); free_pool(c->mixed_node_allocs); free_pool(c->mixed_geom_allocs); free_pool(c->comm_allocs); free_pool(c->tw_part_allocs); free_pool(c->sub_allocs); free_pool(c->mig_allocs);
  (void)sz_comm_destroy(c);
  for (auto& e : c->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  (void)hipFree(c->d_stats); (void)hipFree(c->S.acc);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2);
  (void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_join);
  delete c;
}

const char* sz_last_error(const sz_ctx* c) { return c ? c->err.c_str() : "no context (no HIP device?)"; }

int sz_set_params(sz_ctx* c, const sz_params* p) {
  if (!c || !p) return SZ_E_ARG;
  Params& P = c->P;
  P.E = p->E; P.nu = p->nu; P.mu = p->mu; P.rho_o = p->rho_o; P.rho_a = p->rho_a; P.Cd_io = p->Cd_io; P.Cd_ia = p->Cd_ia;
  P.fcor = p->f; P.turn = p->turn_theta; P.ff_max_overlap = p->floe_floe_max_overlap; P.fd_max_overlap = p->floe_domain_max_overlap;
  P.rho_i = p->rho_i; P.max_h = p->max_floe_height; P.max_xi = p->maximum_xi; P.lambda = p->lambda; P.dd = p->coupling_dd;
  return SZ_OK;
}

int sz_set_domain(sz_ctx* c, const int32_t* kinds, const double* vals, const double* rects, const double* bu, const double* bv) {
  if (!c || !kinds || !vals || !rects) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  for (int k = 0; k < 4; k++) { c->h_kinds[k] = kinds[k]; c->h_vals[k] = vals[k]; c->h_bu[k] = bu ? bu[k] : 0.0; c->h_bv[k] = bv ? bv[k] : 0.0; }
  memcpy(c->h_rects, rects, 16 * sizeof(double));
  // keep the grid fields alive across the element re-upload
  int rc = upload_elements(c);
  if (!rc) setup_grid(c);
  return rc;
}

int sz_set_topography(sz_ctx* c, int32_t ntopo, const int32_t* off, const double* x, const double* y, const double* cx,
                      const double* cy, const double* rmax) {
  if (!c || ntopo < 0) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  c->h_toff.clear(); c->h_tx.clear(); c->h_ty.clear(); c->h_tcx.clear(); c->h_tcy.clear(); c->h_trmax.clear();
  if (ntopo > 0) {
    c->h_toff.assign(off, off + ntopo + 1);
    c->h_tx.assign(x, x + off[ntopo]); c->h_ty.assign(y, y + off[ntopo]);
    c->h_tcx.assign(cx, cx + ntopo); c->h_tcy.assign(cy, cy + ntopo); c->h_trmax.assign(rmax, rmax + ntopo);
  }
  return upload_elements(c);
}

int sz_set_fields(sz_ctx* c, int32_t Nx, int32_t Ny, double x0, double xf, double y0, double yf, const double* uocn,
                  const double* vocn, const double* hflx, const double* uatm, const double* vatm) {
  if (!c || Nx < 1 || Ny < 1 || !uocn || !vocn || !hflx || !uatm || !vatm) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  State& S = c->S;
  size_t n = (size_t)(Nx + 1) * (Ny + 1);
  free_pool(c->field_allocs);
  // the ocean / atmosphere temperatures (sz_set_temps) and the stress fields stay when the lattice keeps its shape:
  // re-uploading changing currents into one context must not reset them to zero
  if (c->tw_ncell != n) { free_pool(c->tw_field_allocs); c->tw_ncell = 0; c->temps_set = false; }
  double** dst[5] = { &S.uo, &S.vo, &S.hf, &S.ua, &S.va };
  const double* src[5] = { uocn, vocn, hflx, uatm, vatm };
  for (int k = 0; k < 5; k++) {
    void* q = nullptr;
    HIPCHK(c, hipMalloc(&q, n * sizeof(double)));
    HIPCHK(c, hipMemcpy(q, src[k], n * sizeof(double), hipMemcpyHostToDevice));
    c->field_allocs.push_back(q);
    *dst[k] = (double*)q;
  }
  {
    void* q = nullptr;
    HIPCHK(c, hipMalloc(&q, n * 8 * sizeof(double)));
    c->field_allocs.push_back(q);
    S.nodes = (double*)q;
  }
  S.Nx = Nx; S.Ny = Ny; S.gx0 = x0; S.gxf = xf; S.gy0 = y0; S.gyf = yf; S.gdx = (xf - x0) / Nx; S.gdy = (yf - y0) / Ny; S.rdx = 1.0 / S.gdx; S.rdy = 1.0 / S.gdy;
  hipLaunchKernelGGL(sz_k_interleave_fields, dim3(grid_for((long long)n, 256)), dim3(256), 0, c->stream, S);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_fields = true; c->mixed_nodes_ok = false; c->blk_pts_ok = false; c->S.sxy = nullptr;
  return SZ_OK;
}

namespace {
// The largest number of bounding-circle neighbours a floe of the uploaded field has (periodic images included), with the circles
// inflated by 25 % so that the count still holds after the field has compacted somewhat: the neighbour lists of the device are sized from
// it (State::maxnb).  window_max: the most floes any 3 x 3 cell window holds (cells a little larger than the device's): the pool of the
// neighbour search's fast variant holds 96.  Fields of like-sized floes stay below 24; a field with a size spectrum -- the reference's Voronoi fields: a large
// cell among small ones -- does not, and the reference's own lists grow as needed (collisions.jl:290-296).  O(M) with a uniform grid.
int host_max_neighbours(const sz_ctx* c, const sz_floe_columns* f, int M, int* window_max) {
  *window_max = 0;
  if (!f->rmax || M < 2) return 0;
  double rm = 0.0;
  for (int i = 0; i < M; i++) rm = std::max(rm, f->rmax[i]);
  if (!(rm > 0.0)) return 0;
  const bool px = c->h_kinds[SZ_EAST] == SZ_PERIODIC && c->h_kinds[SZ_WEST] == SZ_PERIODIC;
  const bool py = c->h_kinds[SZ_NORTH] == SZ_PERIODIC && c->h_kinds[SZ_SOUTH] == SZ_PERIODIC;
  double x0 = c->h_vals[3], xf = c->h_vals[2], y0 = c->h_vals[1], yf = c->h_vals[0];     // W, E, S, N
  double bx0 = f->cx[0], bx1 = f->cx[0], by0 = f->cy[0], by1 = f->cy[0];
  for (int i = 1; i < M; i++) { bx0 = std::min(bx0, f->cx[i]); bx1 = std::max(bx1, f->cx[i]); by0 = std::min(by0, f->cy[i]); by1 = std::max(by1, f->cy[i]); }
  if (!px || !(xf > x0)) { x0 = bx0; xf = bx1 + 1e-9 * std::max(1.0, std::fabs(bx1)); }
  if (!py || !(yf > y0)) { y0 = by0; yf = by1 + 1e-9 * std::max(1.0, std::fabs(by1)); }
  const double Lx = std::max(xf - x0, 1e-300), Ly = std::max(yf - y0, 1e-300), cs = 2.5 * rm;
  const int nx = (int)std::max(1.0, std::min(2048.0, std::floor(Lx / cs))), ny = (int)std::max(1.0, std::min(2048.0, std::floor(Ly / cs)));
  auto wrap = [](double v, double lo, double L) { v = std::fmod(v - lo, L); return v < 0 ? v + L : v; };
  auto cell1 = [&](double v, double lo, double L, int n, bool per) { const double t = per ? wrap(v, lo, L) : v - lo; return std::max(0, std::min(n - 1, (int)(t / L * n))); };
  std::vector<int> head((size_t)nx * ny, -1), next(M), cxi(M), cyi(M);
  for (int i = 0; i < M; i++) {
    cxi[i] = cell1(f->cx[i], x0, Lx, nx, px); cyi[i] = cell1(f->cy[i], y0, Ly, ny, py);
    const size_t cidx = (size_t)cyi[i] * nx + cxi[i]; next[i] = head[cidx]; head[cidx] = i;
  }
  int best = 0;
  for (int i = 0; i < M; i++) {
    int cnt = 0, seen[9], ns = 0, win = 0;
    for (int oy = -1; oy <= 1; oy++)
      for (int ox = -1; ox <= 1; ox++) {
        int ax = cxi[i] + ox, ay = cyi[i] + oy;
        if (px) ax = (ax % nx + nx) % nx; else if (ax < 0 || ax >= nx) continue;
        if (py) ay = (ay % ny + ny) % ny; else if (ay < 0 || ay >= ny) continue;
        const int cidx = ay * nx + ax;
        bool dup = false;
        for (int q = 0; q < ns; q++) dup |= seen[q] == cidx;
        if (dup) continue;
        seen[ns++] = cidx;
        for (int j = head[cidx]; j >= 0; j = next[j]) {
          win++;
          if (j == i) continue;
          double dx = f->cx[i] - f->cx[j], dy = f->cy[i] - f->cy[j];
          if (px) dx -= Lx * std::nearbyint(dx / Lx);
          if (py) dy -= Ly * std::nearbyint(dy / Ly);
          const double rr = 1.25 * (f->rmax[i] + f->rmax[j]);
          cnt += (dx * dx + dy * dy) < rr * rr;
        }
      }
    best = std::max(best, cnt); *window_max = std::max(*window_max, win);
  }
  return best;
}
}  // namespace

int sz_upload_floes(sz_ctx* c, int64_t M64, int64_t N64, const sz_floe_columns* f) {
  if (!c || !f || M64 < 0 || N64 < 0 || N64 > M64 || !f->vert_off || !f->vx || !f->vy || !f->cx || !f->cy) return SZ_E_ARG;
  if (!c->have_domain) { c->err = "sz_set_domain must be called before sz_upload_floes"; return SZ_E_STATE; }
  if (M64 > N64 && (!f->ghost_off || !f->ghost_idx || !f->ghost_id)) { c->err = "M > N needs ghost_off/ghost_idx/ghost_id"; return SZ_E_ARG; }
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->gi_pending && c->gi_valid) { int rc0 = gi_fetch(c); if (rc0) return rc0; }      // (the key tables go with the pool; the rows may stay)
  c->gi_pending = false;
  free_pool(c->sub_allocs);
  reset_pool(c->allocs);       // the chunks of the previous upload are carved again (a shim uploads before every replaced call)
  State& S = c->S;
  const int M = (int)M64, N = (int)N64;
  const int V = f->vert_off[M];
  c->max_ring = 0;
  for (int i = 0; i < M; i++) c->max_ring = std::max(c->max_ring, f->vert_off[i + 1] - f->vert_off[i]);
  const int NS = f->sub_off ? f->sub_off[N] : 0;
  // neighbour and row capacities from the field itself: like-sized floes 24 / 32, a size spectrum 64 / 128 (SZ_MAXNB=24|64 overrides)
  int window_max = 0;
  c->nb_count_max = host_max_neighbours(c, f, M, &window_max);
  S.maxnb = c->nb_count_max + 2 <= MAXNB && window_max <= 80 ? MAXNB : c->nb_count_max + 2 <= 64 ? 64 : 256;
  if (const char* e = getenv("SZ_MAXNB")) S.maxnb = atoi(e) > 64 ? 256 : atoi(e) > MAXNB ? 64 : MAXNB;
  S.rowcap = S.maxnb <= MAXNB ? ROWCAP : S.maxnb <= 64 ? 128 : 512;
  if (const char* e = getenv("SZ_ROWCAP")) S.rowcap = std::max(1, atoi(e));          // (tests of the growth path)
  S.capM = std::max(2 * M + 64, M + 2048);      // (rows for the floes, their ghosts and -- a tile -- its halo: small tiles of fast floes hold more halo floes than owned ones)
  S.capV = std::max(2 * V + 4096, V + 32768); S.capPairs = S.capM * (S.maxnb <= MAXNB ? 8 : 16); S.capElem = S.capM * 4;
  S.capRows = S.capPairs * 3 + S.capElem * 2; S.capCells = 4 * S.capM + 64; S.capS = NS;
  int rc;
#define DA(field, n) if ((rc = dalloc(c, &S.field, (size_t)(n), c->allocs))) return rc
  DA(cnt, C_COUNT + 64 + 72); DA(warn, WARN_SLOTS * 32);      // counters | 64 per-rank counts of the pack kernel | its 66 scratch words
  double** dcols[] = { &S.cx, &S.cy, &S.rmax, &S.area, &S.height, &S.mass, &S.moment, &S.alpha, &S.u, &S.v, &S.xi,
                       &S.p_dxdt, &S.p_dydt, &S.p_dalphadt, &S.p_dudt, &S.p_dvdt, &S.p_dxidt, &S.fxOA, &S.fyOA, &S.trqOA,
                       &S.hflx, &S.overarea, &S.cfx, &S.cfy, &S.ctrq };
  double* const hcols[] = { f->cx, f->cy, f->rmax, f->area, f->height, f->mass, f->moment, f->alpha, f->u, f->v, f->xi,
                            f->p_dxdt, f->p_dydt, f->p_dalphadt, f->p_dudt, f->p_dvdt, f->p_dxidt, f->fxOA, f->fyOA, f->trqOA,
                            f->hflx_factor, f->overarea, f->coll_fx, f->coll_fy, f->coll_trq };
  for (size_t k = 0; k < sizeof(dcols) / sizeof(dcols[0]); k++) {
    if ((rc = dalloc(c, dcols[k], S.capM, c->allocs))) return rc;
    if (hcols[k]) H2D(*dcols[k], hcols[k], M, double);
  }
  DA(sa, 4 * S.capM); DA(si, 4 * S.capM); DA(strain, 4 * S.capM); DA(mot, 4 * S.capM); DA(mot2, 2 * S.capM); DA(trig, 2 * S.capM);
  if (f->stress_accum) H2D(S.sa, f->stress_accum, 4 * M, double);
  if (f->stress_instant) H2D(S.si, f->stress_instant, 4 * M, double);
  if (f->strain) H2D(S.strain, f->strain, 4 * M, double);
  DA(id, S.capM); DA(ghost_id, S.capM); DA(okey, S.capM); DA(status, S.capM); DA(parent, S.capM); DA(gh, MAX_GHOSTS * S.capM); DA(ngh, S.capM);
  DA(gh_save, MAX_GHOSTS * S.capM); DA(ngh_save, S.capM);
  std::vector<int> tag0;          // (tagA starts as the status column: the integrator of a resident step only rewrites it where a tag was raised)
  DA(frc_remove, S.capM); DA(osign, S.capM); DA(bbx0, S.capM); DA(bbx1, S.capM); DA(bby0, S.capM); DA(bby1, S.capM);
  {
    std::vector<long long> id(M), gid(M, 0);
    std::vector<int> st(M, SZ_ACTIVE), parent(M), gh((size_t)MAX_GHOSTS * M, -1), ngh(M, 0);
    for (int i = 0; i < M; i++) { id[i] = f->id ? f->id[i] : i + 1; if (f->ghost_id) gid[i] = f->ghost_id[i]; if (f->status) st[i] = f->status[i]; parent[i] = i; }
    if (M > N) {
      for (int i = 0; i < N; i++) {
        int n = f->ghost_off[i + 1] - f->ghost_off[i];
        if (n > MAX_GHOSTS) { c->err = "more than 3 ghosts for one parent"; return SZ_E_ARG; }
        ngh[i] = n;
        for (int k = 0; k < n; k++) { int g = f->ghost_idx[f->ghost_off[i] + k]; if (g < N || g >= M) { c->err = "ghost index out of range"; return SZ_E_ARG; } gh[(size_t)i * MAX_GHOSTS + k] = g; parent[g] = i; }
      }
    }
    { std::vector<long long> ok(S.capM); for (int i = 0; i < S.capM; i++) ok[i] = i; H2D(S.okey, ok.data(), S.capM, long long); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    H2D(S.id, id.data(), M, long long); H2D(S.ghost_id, gid.data(), M, long long); H2D(S.status, st.data(), M, int);
    tag0 = st;
    H2D(S.parent, parent.data(), M, int); H2D(S.gh, gh.data(), (size_t)MAX_GHOSTS * M, int); H2D(S.ngh, ngh.data(), M, int);
    HIPCHK(c, hipStreamSynchronize(c->stream));   // host vectors go out of scope
  }
  DA(voff, S.capM + 1); DA(vxy, S.capV);
  H2D(S.voff, f->vert_off, M + 1, int);
  {          // the rings, interleaved {x, y} on the device (State::vxy)
    std::vector<double> xy((size_t)2 * std::max(V, 1));
    for (int k = 0; k < V; k++) { xy[(size_t)2 * k] = f->vx[k]; xy[(size_t)2 * k + 1] = f->vy[k]; }
    H2D(S.vxy, xy.data(), (size_t)2 * V, double); HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  DA(soff, S.capM + 1); DA(sx, NS); DA(sy, NS);
  if (f->sub_off) { H2D(S.soff, f->sub_off, N + 1, int); H2D(S.sx, f->sx, NS, double); H2D(S.sy, f->sy, NS, double); }
  c->max_sub = 0;
  if (f->sub_off) for (int i = 0; i < N; i++) c->max_sub = std::max(c->max_sub, f->sub_off[i + 1] - f->sub_off[i]);
  DA(gplan, S.capM + 1); DA(gscan4, S.capM + 1); DA(gtot4, 4);
  DA(lb_agg, S.capM / 128 + 8); DA(lb_inc, S.capM / 128 + 8); DA(lb_flag, S.capM / 128 + 8); c->scan_epoch = 0;      // (tiles of 128 .. SCAN_B elements)
  DA(gflag, S.capM + 1); DA(gvscan, S.capM + 2); DA(gcand, (size_t)2 * S.capM); DA(galloc, 32); DA(gkeys, (size_t)2 * S.capM); DA(fam, S.capM);
  DA(crec, (size_t)8 * S.capM); c->crec_buf = S.crec; S.crec = nullptr;
  {          // the twin set of the pipelined steps' double buffers (sz_pipeline.hpp)
    sz_ctx::PipeBuf& B = c->pb[1];
    if ((rc = dalloc(c, &B.vxy, (size_t)S.capV, c->allocs)) || (rc = dalloc(c, &B.crec, (size_t)8 * S.capM, c->allocs)) ||
        (rc = dalloc(c, &B.gh, (size_t)MAX_GHOSTS * S.capM, c->allocs)) || (rc = dalloc(c, &B.ngh, (size_t)S.capM, c->allocs))) return rc;
    HIPCHK(c, hipMemsetAsync(B.gh, 0xff, (size_t)MAX_GHOSTS * S.capM * sizeof(int), c->stream));
    c->gpar = 0;
  }
  DA(facc, (size_t)FX_WORDS * S.capM); c->facc_buf = S.facc; S.facc = nullptr;
  c->maybe_tagged = false; c->crec_current = false;
  if (f->status) for (int i = 0; i < N; i++) if (f->status[i] != SZ_ACTIVE) { c->maybe_tagged = true; break; }
  for (int k = 0; k < 4; k++) if ((rc = dalloc(c, &c->frc_alt[k], (size_t)S.capM, c->allocs))) return rc;
  DA(bounds, 16 + 64 * 4); DA(cell_cnt, S.capCells + 1); DA(cell_ovf, S.capCells + 1); DA(cell_slots, (size_t)S.capCells * CELL_K + 8);
  DA(cell_items, S.capM);
  {
    sz_ctx::PipeBuf& B = c->pb[1];
    if ((rc = dalloc(c, &B.cell_cnt, (size_t)S.capCells + 1, c->allocs)) || (rc = dalloc(c, &B.cell_ovf, (size_t)S.capCells + 1, c->allocs)) ||
        (rc = dalloc(c, &B.cell_slots, (size_t)S.capCells * CELL_K + 8, c->allocs)) || (rc = dalloc(c, &B.cell_items, (size_t)S.capM, c->allocs))) return rc;
  }
  DA(n_out, S.capM + 1); DA(n_in, S.capM + 1); DA(over_stamp, S.capM + 1); DA(over_base, S.capM + 1);
  DA(out_off, S.capM + 2);
  DA(el_off, S.capM + 2); DA(el_floe, S.capElem); DA(el_elem, S.capElem);
  DA(inter_off, S.capM + 2);
  HIPCHK(c, hipMemsetAsync(S.over_stamp, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));
  // (a re-upload carves the chunks of the previous one again: per-floe COUNTS that a kernel may read before a collision call has written them
  //  -- sz_k_stats walks n_out -- must not hold what some other array left there)
  HIPCHK(c, hipMemsetAsync(S.n_out, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.n_in, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.el_off, 0, ((size_t)S.capM + 2) * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.warn, 0, (size_t)WARN_SLOTS * 32 * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.lb_flag, 0, ((size_t)S.capM / 128 + 8) * sizeof(unsigned), c->stream));      // (the look-back scans' epochs start over with scan_epoch = 0)
  if (!f->sub_off) HIPCHK(c, hipMemsetAsync(S.soff, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));
  // neighbour lists, pair items and their rows: in a pool of their own, carved again (larger) when a step outgrows them
  if ((rc = carve_lists(c))) return rc;
  // floe.interactions is part of the floe state, but not of sz_floe_columns (it is ragged): the rows the last
  // collision call left stay valid across an upload of the same size; after an upload of another size they are
  // gone, and sz_timestep_floe_properties / sz_calc_stress refuse to run on nothing (SZ_E_STATE) until
  // sz_upload_interactions or a collision call provides them again
  if ((rc = carve_interactions(c))) return rc;
  DA(blk, std::max(S.capCells, std::max(S.capM, 1024)) / SCAN_B + 1024);
  { sz_ctx::PipeBuf& B = c->pb[0]; B.vxy = S.vxy; B.crec = c->crec_buf; B.cell_cnt = S.cell_cnt; B.cell_slots = S.cell_slots; B.cell_ovf = S.cell_ovf; B.cell_items = S.cell_items; B.gh = S.gh; B.ngh = S.ngh; }
  DA(tagA, S.capM + 1);
  if (!tag0.empty()) { H2D(S.tagA, tag0.data(), M, int); HIPCHK(c, hipStreamSynchronize(c->stream)); }
  DA(stamps, 512 + 8 * 8000);
  trim_pool(c->allocs);
  int h[C_COUNT + 64 + 72] = { 0 };
  h[C_M] = M; h[C_N] = N; h[C_NV] = V; h[C_NGHOSTS] = M - N; h[C_NOWN] = N;
  S.tiled = 0; S.famrec = 0;
  // a new field: whatever sz_tile_enable / sz_tile_setup established belongs to the old one (exchange buffers sized for its capM, the
  // drift reference, the gather interval): sz_tile_run refuses to run until both have been called again
  c->tile_margin = 0.0; c->tile_since_box = -1; c->halo_cap = 0; c->d_send = c->d_recv = c->d_ref = nullptr; c->d_dcap = nullptr;
  free_pool(c->comm_allocs);
  H2D(S.cnt, h, C_COUNT + 64 + 72, int);
  hipLaunchKernelGGL(sz_k_osign, dim3(grid_for(S.capM, 256)), dim3(256), 0, c->stream, S, 0);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->hostM = M; c->hostN = N; c->have_floes = true; c->tile_dirty = false; c->mixed_pts_ok = false; c->blk_pts_ok = false; S.sxy = nullptr; c->pts_N = f->sub_off ? N : 0; c->upload_M = M; c->upload_V = V;
  c->gl_valid = false; c->gl_est = 0;
  c->mixed_geom_ok = false; c->rings_stale = false; S.rec32 = nullptr; S.ring32 = nullptr; S.body_rings = 0;
  if (f->rmax) {          // parents near a periodic wall: how long the ghost-candidate list will be (a superset of it)
    const double x0 = c->h_vals[3], xf = c->h_vals[2], y0 = c->h_vals[1], yf = c->h_vals[0];
    const bool pew = c->h_kinds[SZ_EAST] == SZ_PERIODIC && c->h_kinds[SZ_WEST] == SZ_PERIODIC;
    const bool pns = c->h_kinds[SZ_NORTH] == SZ_PERIODIC && c->h_kinds[SZ_SOUTH] == SZ_PERIODIC;
    for (int i = 0; i < N; i++) {
      const double r = f->rmax[i];
      if ((pew && (f->cx[i] - r < x0 || f->cx[i] + r > xf)) || (pns && (f->cy[i] - r < y0 || f->cy[i] + r > yf))) c->gl_est++;
    }
  } else c->gl_est = N;
  c->rmax_max = 0.0; c->rmax_hint = 0.0;
  if (f->rmax) for (int i = 0; i < M; i++) c->rmax_max = std::max(c->rmax_max, f->rmax[i]);
  setup_grid(c);
  c->fuse_lists.assign(M, {});
  return SZ_OK;
}

int sz_get_stats(sz_ctx* c, sz_stats* out) {
  if (!c || !out || !c->have_floes) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  tile_cleanup(c);
  State& S = c->S;
  HIPCHK(c, hipMemsetAsync(c->d_stats, 0, 20 * sizeof(long long), c->stream));
  hipLaunchKernelGGL(sz_k_stats, dim3(grid_for((long long)S.capPairs + S.capElem, 256, 1024)), dim3(256), 0, c->stream, S, c->d_stats);
  int h[C_COUNT]; long long st[20];
  HIPCHK(c, hipMemcpyAsync(st, c->d_stats, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h, S.cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int soffN = 0;
  HIPCHK(c, hipMemcpy(&soffN, S.soff + h[C_N], sizeof(int), hipMemcpyDeviceToHost));
  out->M = h[C_M]; out->N = h[C_N]; out->n_ring_points = h[C_NV]; out->n_sub_points = soffN;
  out->n_pairs = st[16]; out->n_pair_ring_points = st[0]; out->n_pair_rows = st[1];
  out->n_elem_items = h[C_NELEM]; out->n_elem_rows = st[2]; out->n_inter_rows = st[3]; out->n_ghosts = h[C_NGHOSTS];
  out->warn_height = st[6]; out->warn_force = st[7]; out->warn_vel = st[8]; out->warn_xi = st[9];
  out->n_trace_fail = h[C_TRACE_FAIL];
  out->n_halo = h[C_NHALO];
  out->n_pairs_clipped = st[17];
  out->n_status_remove = st[4]; out->n_status_fuse = st[5];
  out->n_retry = h[C_NRETRY];
  out->acc_narrow_launches = st[10]; out->acc_pair_items = st[11]; out->acc_pair_ring_points = st[12];
  out->acc_pair_rows = st[13]; out->acc_elem_items = st[14]; out->acc_elem_rows = st[15];
  out->acc_dir_checks = st[18]; out->acc_dir_checks_certified = st[19];
  return SZ_OK;
}

int sz_download_floes(sz_ctx* c, sz_floe_columns* f) {
  if (!c || !f || !c->have_floes) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  tile_cleanup(c);
  world_rings(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  State& S = c->S;
  int h[C_COUNT];
  HIPCHK(c, hipMemcpy(h, S.cnt, sizeof(h), hipMemcpyDeviceToHost));
  int M = h[C_M], N = h[C_N], V = h[C_NV];
#define D2H(dst, src, n, T) if (dst) HIPCHK(c, hipMemcpy(dst, src, (size_t)(n) * sizeof(T), hipMemcpyDeviceToHost))
  D2H(f->cx, S.cx, M, double); D2H(f->cy, S.cy, M, double); D2H(f->rmax, S.rmax, M, double); D2H(f->area, S.area, M, double);
  D2H(f->height, S.height, M, double); D2H(f->mass, S.mass, M, double); D2H(f->moment, S.moment, M, double);
  D2H(f->alpha, S.alpha, M, double); D2H(f->u, S.u, M, double); D2H(f->v, S.v, M, double); D2H(f->xi, S.xi, M, double);
  D2H(f->p_dxdt, S.p_dxdt, M, double); D2H(f->p_dydt, S.p_dydt, M, double); D2H(f->p_dalphadt, S.p_dalphadt, M, double);
  D2H(f->p_dudt, S.p_dudt, M, double); D2H(f->p_dvdt, S.p_dvdt, M, double); D2H(f->p_dxidt, S.p_dxidt, M, double);
  D2H(f->fxOA, S.fxOA, M, double); D2H(f->fyOA, S.fyOA, M, double); D2H(f->trqOA, S.trqOA, M, double);
  D2H(f->hflx_factor, S.hflx, M, double); D2H(f->overarea, S.overarea, M, double);
  D2H(f->coll_fx, S.cfx, M, double); D2H(f->coll_fy, S.cfy, M, double); D2H(f->coll_trq, S.ctrq, M, double);
  D2H(f->stress_accum, S.sa, 4 * M, double); D2H(f->stress_instant, S.si, 4 * M, double); D2H(f->strain, S.strain, 4 * M, double);
  D2H(f->id, S.id, M, long long); D2H(f->ghost_id, S.ghost_id, M, long long); D2H(f->status, S.status, M, int);
  D2H(f->vert_off, S.voff, M + 1, int);
  if ((f->vx || f->vy) && V > 0) {
    std::vector<double> xy((size_t)2 * V);
    HIPCHK(c, hipMemcpy(xy.data(), S.vxy, (size_t)2 * V * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < V; k++) { if (f->vx) f->vx[k] = xy[(size_t)2 * k]; if (f->vy) f->vy[k] = xy[(size_t)2 * k + 1]; }
  }
  if (f->ghost_off) {
    std::vector<int> gh((size_t)MAX_GHOSTS * M), ngh(M);
    HIPCHK(c, hipMemcpy(gh.data(), S.gh, gh.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ngh.data(), S.ngh, (size_t)M * sizeof(int), hipMemcpyDeviceToHost));
    int t = 0; f->ghost_off[0] = 0;
    for (int i = 0; i < M; i++) {
      int n = i < N ? ngh[i] : 0;
      for (int k = 0; k < n; k++) { if (f->ghost_idx) f->ghost_idx[t] = gh[(size_t)i * MAX_GHOSTS + k]; t++; }
      f->ghost_off[i + 1] = t;
    }
  }
  return SZ_OK;
}

// ---------------------------------------------------------------- Float32 hosts (include/subzero_hip.h: sz_floe_columns_f32)
#define SZ_F32_SCALARS(X) X(cx) X(cy) X(rmax) X(area) X(height) X(mass) X(moment) X(alpha) X(u) X(v) X(xi) X(p_dxdt) X(p_dydt) X(p_dalphadt) \
  X(p_dudt) X(p_dvdt) X(p_dxidt) X(fxOA) X(fyOA) X(trqOA) X(hflx_factor) X(overarea) X(coll_fx) X(coll_fy) X(coll_trq)
#define SZ_F32_TENSORS(X) X(stress_accum) X(stress_instant) X(strain)
int sz_upload_floes_f32(sz_ctx* c, int64_t M, int64_t N, const sz_floe_columns_f32* f) {
  if (!c || !f || M < 0 || N < 0 || N > M) return SZ_E_ARG;
  if (M > 0 && !f->vert_off) { c->err = "sz_upload_floes_f32: vert_off is required"; return SZ_E_ARG; }
  const size_t V = M > 0 ? (size_t)f->vert_off[M] : 0, NS = (f->sub_off && N > 0) ? (size_t)f->sub_off[N] : 0;
  std::vector<std::vector<double>> keep;
  auto widen = [&](const float* src, size_t n) -> double* {
    if (!src) return nullptr;
    keep.emplace_back(n ? n : 1);
    for (size_t k = 0; k < n; k++) keep.back()[k] = (double)src[k];
    return keep.back().data();
  };
  sz_floe_columns d; memset(&d, 0, sizeof(d));
#define X(name) d.name = widen(f->name, (size_t)M);
  SZ_F32_SCALARS(X)
#undef X
#define X(name) d.name = widen(f->name, (size_t)4 * M);
  SZ_F32_TENSORS(X)
#undef X
  d.id = f->id; d.ghost_id = f->ghost_id; d.status = f->status; d.vert_off = f->vert_off; d.sub_off = f->sub_off; d.ghost_off = f->ghost_off; d.ghost_idx = f->ghost_idx;
  d.vx = widen(f->vx, V); d.vy = widen(f->vy, V); d.sx = widen(f->sx, NS); d.sy = widen(f->sy, NS);
  return sz_upload_floes(c, M, N, &d);
}
int sz_download_floes_f32(sz_ctx* c, sz_floe_columns_f32* f) {
  if (!c || !f) return SZ_E_ARG;
  sz_stats st;
  int rc = sz_get_stats(c, &st); if (rc) return rc;
  const size_t M = (size_t)st.M, V = (size_t)st.n_ring_points, NS = (size_t)st.n_sub_points;
  std::vector<std::vector<double>> keep;
  struct Back { float* dst; double* src; size_t n; }; std::vector<Back> back;
  auto room = [&](float* dst, size_t n) -> double* {
    if (!dst) return nullptr;
    keep.emplace_back(n ? n : 1);
    back.push_back({ dst, keep.back().data(), n });
    return keep.back().data();
  };
  sz_floe_columns d; memset(&d, 0, sizeof(d));
#define X(name) d.name = room(f->name, M);
  SZ_F32_SCALARS(X)
#undef X
#define X(name) d.name = room(f->name, 4 * M);
  SZ_F32_TENSORS(X)
#undef X
  d.id = f->id; d.ghost_id = f->ghost_id; d.status = f->status; d.vert_off = f->vert_off; d.sub_off = f->sub_off; d.ghost_off = f->ghost_off; d.ghost_idx = f->ghost_idx;
  // (sx / sy are not written by a download -- sz_download_subpoints is their way back --, so the caller's buffers are left alone: staging them
  //  here used to hand zeros back)
  d.vx = room(f->vx, V); d.vy = room(f->vy, V); d.sx = nullptr; d.sy = nullptr; (void)NS;
  rc = sz_download_floes(c, &d); if (rc) return rc;
  for (const Back& b : back) for (size_t k = 0; k < b.n; k++) b.dst[k] = (float)b.src[k];
  return SZ_OK;
}
int sz_set_fields_f32(sz_ctx* c, int32_t Nx, int32_t Ny, double x0, double xf, double y0, double yf, const float* uocn, const float* vocn,
                      const float* hflx, const float* uatm, const float* vatm) {
  if (!c || Nx < 1 || Ny < 1) return SZ_E_ARG;
  const size_t n = (size_t)(Nx + 1) * (Ny + 1);
  std::vector<double> a[5]; const float* src[5] = { uocn, vocn, hflx, uatm, vatm };
  for (int k = 0; k < 5; k++) if (src[k]) { a[k].resize(n); for (size_t q = 0; q < n; q++) a[k][q] = (double)src[k][q]; }
  auto p = [&](int k) { return src[k] ? a[k].data() : (const double*)nullptr; };
  return sz_set_fields(c, Nx, Ny, x0, xf, y0, yf, p(0), p(1), p(2), p(3), p(4));
}
int sz_download_interactions_f32(sz_ctx* c, int32_t* off, float* rows) {
  if (!c || !off) return SZ_E_ARG;
  sz_stats st;
  int rc = sz_get_stats(c, &st); if (rc) return rc;
  rc = sz_download_interactions(c, off, nullptr); if (rc) return rc;          // the offsets first: off[M] = rows in all
  const size_t total = (size_t)off[st.M];
  if (!rows || total == 0) return SZ_OK;
  std::vector<double> r(total * 7);
  rc = sz_download_interactions(c, off, r.data()); if (rc) return rc;
  for (size_t k = 0; k < total * 7; k++) rows[k] = (float)r[k];
  return SZ_OK;
}

int sz_download_interactions(sz_ctx* c, int32_t* off, double* rows) {
  if (!c || !off || !c->have_floes) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  tile_cleanup(c);
  State& S = c->S;
  // rows are kept at a fixed stride on the device: compact to CSR here
  scan(c, S.inter_cnt, S.inter_off, S.capM, C_M, 0, C_NINTER);
  int h[C_COUNT];
  HIPCHK(c, hipMemcpyAsync(h, S.cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(off, S.inter_off, (size_t)(h[C_M] + 1) * sizeof(int), hipMemcpyDeviceToHost));
  int total = h[C_NINTER];
  if (rows && total > 0) {
    double* tmp = nullptr;
    HIPCHK(c, hipMalloc((void**)&tmp, (size_t)total * 7 * sizeof(double)));
    hipLaunchKernelGGL(sz_k_inter_compact, dim3(grid_for(S.capM, 128)), dim3(128), 0, c->stream, S, tmp);
    hipError_t e = hipMemcpyAsync(rows, tmp, (size_t)total * 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) { c->err = hipGetErrorString(e); return SZ_E_HIP; }
    if (c->gi_valid) { int rc2 = gi_fetch(c); if (rc2) return rc2; }
    if (c->gi_valid) {            // partners that were inline ghosts: order key -> the reference's floe number
      std::vector<long long> sorted(c->gi_keys);
      std::sort(sorted.begin(), sorted.end());
      const double lim = (double)((long long)1 << 40);
      for (int r = 0; r < total; r++) {
        const double idx = rows[(size_t)r * 7];
        if (idx <= lim) continue;
        const long long key = (long long)idx - 1;
        const auto it = std::lower_bound(sorted.begin(), sorted.end(), key);
        if (it == sorted.end() || *it != key) { c->err = "interaction row names a ghost that is not among the last step's"; return SZ_E_STATE; }
        rows[(size_t)r * 7] = (double)(c->hostN + (int)(it - sorted.begin()) + 1);
      }
    }
  }
  return SZ_OK;
}

int sz_download_pairs(sz_ctx* c, int32_t* pi, int32_t* pj) {
  if (!c || !pi || !pj || !c->have_floes) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  tile_cleanup(c);
  State& S = c->S;
  // the compact list is made here, on demand: the steps themselves only keep the per-floe sorted lists (the ghosts of the
  // last resident step own pairs too: their rows outlive their removal)
  int h[C_COUNT];
  int rc = sync_and_check(c, h);
  if (rc) return rc;
  const int mlast = std::max(h[C_M], h[C_N] + h[C_NGHOSTS]);
  scan(c, S.n_out, S.out_off, S.capM, -1, mlast, C_NPAIRS);
  hipLaunchKernelGGL(sz_k_pairs_fill, dim3(grid_for(S.capM, 256)), dim3(256), 0, c->stream, S, mlast);
  rc = sync_and_check(c, h);
  if (rc) return rc;
  if (h[C_NPAIRS] > 0) {
    HIPCHK(c, hipMemcpy(pi, c->S.pair_i, (size_t)h[C_NPAIRS] * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(pj, c->S.pair_j, (size_t)h[C_NPAIRS] * sizeof(int), hipMemcpyDeviceToHost));
    if (c->gi_valid) { int rc2 = gi_fetch(c); if (rc2) return rc2; }
    if (c->gi_valid && (int)c->gi_ref.size() == mlast - h[C_N]) {       // inline ghosts: storage index -> the reference's number, then its serial order
      const int Np = h[C_N], np = h[C_NPAIRS];
      std::vector<std::pair<int, int>> ps(np);
      for (int k = 0; k < np; k++)
        ps[k] = { pi[k] < Np ? pi[k] : Np + c->gi_ref[pi[k] - Np], pj[k] < Np ? pj[k] : Np + c->gi_ref[pj[k] - Np] };
      std::sort(ps.begin(), ps.end());
      for (int k = 0; k < np; k++) { pi[k] = ps[k].first; pj[k] = ps[k].second; }
    }
  }
  return SZ_OK;
}

int sz_download_fuse(sz_ctx* c, int32_t* off, int32_t* idx) {
  if (!c || !off || !c->have_floes) return SZ_E_ARG;
  int M = c->hostM, t = 0;
  off[0] = 0;
  for (int i = 0; i < M; i++) {
    if (i < (int)c->fuse_lists.size())
      for (int v : c->fuse_lists[i]) { if (idx) idx[t] = v; t++; }
    off[i + 1] = t;
  }
  return SZ_OK;
}

int sz_get_boundary_vals(sz_ctx* c, double* vals4) {
  if (!c || !vals4 || !c->have_domain) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  tile_cleanup(c);
  HIPCHK(c, hipMemcpy(vals4, c->S.eval, 4 * sizeof(double), hipMemcpyDeviceToHost));
  return SZ_OK;
}

// the four boundary rectangles as they stand: {xmin, xmax, ymin, ymax} each, order N, S, E, W (MovingBoundary walls move)
int sz_get_boundary_rects(sz_ctx* c, double* rects16) {
  if (!c || !rects16 || !c->have_domain) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  tile_cleanup(c);
  HIPCHK(c, hipMemcpy(rects16, c->S.erect, 16 * sizeof(double), hipMemcpyDeviceToHost));
  return SZ_OK;
}

// which_vertices_match_points on given points and a given region ring (the reference's test vectors for it)
int sz_debug_match_vertices(sz_ctx* c, int32_t npts, const double* px, const double* py, int32_t nr, const double* rx, const double* ry,
                            int32_t* idx, int32_t* n_out) {
  if (!c || npts < 0 || npts > NARROW_KC2 || nr < 1 || nr > NARROW_RC2 || !idx || !n_out) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  PoolGuard pool;
  double *dpx, *dpy, *drx, *dry; int* dout;
  int rc;
  if ((rc = dalloc(c, &dpx, npts, pool.v)) || (rc = dalloc(c, &dpy, npts, pool.v)) || (rc = dalloc(c, &drx, nr, pool.v)) ||
      (rc = dalloc(c, &dry, nr, pool.v)) || (rc = dalloc(c, &dout, npts + 1, pool.v))) return rc;
  H2D(dpx, px, npts, double); H2D(dpy, py, npts, double); H2D(drx, rx, nr, double); H2D(dry, ry, nr, double);
  hipLaunchKernelGGL(sz_k_debug_match_vertices, dim3(1), dim3(64), 0, c->stream, npts, dpx, dpy, nr, drx, dry, dout);
  std::vector<int> h(npts + 1);
  HIPCHK(c, hipMemcpyAsync(h.data(), dout, (size_t)(npts + 1) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_out = h[0];
  for (int k = 0; k < h[0]; k++) idx[k] = h[1 + k];
  return SZ_OK;
}

// the forcing kernels' in-bounds test and lattice sample at given points (the reference's vectors for in_bounds / find_interp_knots)
int sz_debug_sample_fields(sz_ctx* c, int32_t n, const double* x, const double* y, double* out12) {
  if (!c || n < 0 || (n > 0 && (!x || !y || !out12))) return SZ_E_ARG;
  if (!c->have_fields || !c->have_domain) { c->err = "sz_debug_sample_fields needs sz_set_domain and sz_set_fields"; return SZ_E_STATE; }
  if (n == 0) return SZ_OK;
  (void)hipSetDevice(c->device);
  PoolGuard pool;
  double *dx, *dy, *dout;
  int rc;
  if ((rc = dalloc(c, &dx, n, pool.v)) || (rc = dalloc(c, &dy, n, pool.v)) || (rc = dalloc(c, &dout, (size_t)12 * n, pool.v))) return rc;
  H2D(dx, x, n, double); H2D(dy, y, n, double);
  hipLaunchKernelGGL(sz_k_debug_sample, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->S, n, dx, dy, dout);
  HIPCHK(c, hipMemcpyAsync(out12, dout, (size_t)12 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SZ_OK;
}

// ---------------------------------------------------------------- processes
int sz_add_ghosts(sz_ctx* c) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  c->gi_valid = false;         // (what follows numbers its ghosts by storage position)
  leave_resident(c);            // a process-mode call: the resident steps' ghost-candidate list is stale, the world rings must be current
  int oldM = c->hostM;
  stage_ghosts(c);
  int rc = sync_and_check(c);
  if (rc) return rc;
  // deepcopy_floe copies status.fuse_idx (floe_utils.jl:138)
  if (c->hostM > oldM) {
    std::vector<int> parent(c->hostM);
    HIPCHK(c, hipMemcpy(parent.data(), c->S.parent, (size_t)c->hostM * sizeof(int), hipMemcpyDeviceToHost));
    c->fuse_lists.resize(c->hostM);
    for (int g = oldM; g < c->hostM; g++) c->fuse_lists[g] = c->fuse_lists[parent[g]];
  }
  return SZ_OK;
}

int sz_remove_ghosts(sz_ctx* c) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  leave_resident(c);            // a process-mode call: the resident steps' ghost-candidate list is stale, the world rings must be current
  hipLaunchKernelGGL(sz_k_remove_ghosts, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, 0);
  int rc = sync_and_check(c);
  if (rc) return rc;
  c->fuse_lists.resize(c->hostM);
  return SZ_OK;
}

int sz_timestep_collisions(sz_ctx* c, int64_t n_init, int32_t dt) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  c->maybe_tagged = true;
  (void)hipSetDevice(c->device);
  c->gi_valid = false;         // (what follows numbers its ghosts by storage position)
  leave_resident(c);            // a process-mode call: the resident steps' ghost-candidate list is stale, the world rings must be current
  c->S.callid = ++c->callid;
  int h[C_COUNT];
  for (;;) {
    collisions(c, (int)n_init, dt);
    c->inter_any = true; c->inter_lost = false;
    int rc = sync_and_check(c, h);
    // a list outgrown (neighbours per floe, pair items, rows per floe): larger lists, the call again -- the reference's lists grow (collisions.jl:290-296)
    if (rc == SZ_E_CAPACITY && growable(c->last_err_bits)) { if ((rc = grow_lists(c, c->last_err_bits))) return rc; continue; }
    if (rc) return rc;
    break;
  }
  return host_fuse_fixup(c, h, true);
}

int sz_collide_pairs(sz_ctx* c, int64_t np, const int32_t* pi, const int32_t* pj, int32_t dt, double max_overlap) {
  if (!c || !c->have_floes || np < 0 || (np > 0 && (!pi || !pj))) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  c->gi_valid = false;         // (what follows numbers its ghosts by storage position)
  leave_resident(c);
  State& S = c->S;
  if (np > S.capPairs) { c->err = "too many explicit pairs"; return SZ_E_CAPACITY; }
  c->maybe_tagged = true;
  std::vector<std::pair<int, int>> ps(np);
  for (int64_t k = 0; k < np; k++) {
    if (pi[k] < 0 || pj[k] < 0 || pi[k] >= c->hostM || pj[k] >= c->hostM || pi[k] == pj[k]) { c->err = "pair index out of range"; return SZ_E_ARG; }
    ps[k] = { pi[k], pj[k] };
  }
  std::sort(ps.begin(), ps.end());
  std::vector<int> hi(np), hj(np);
  for (int64_t k = 0; k < np; k++) { hi[k] = ps[k].first; hj[k] = ps[k].second; }
  if (np) { H2D(S.pair_i, hi.data(), np, int); H2D(S.pair_j, hj.data(), np, int); }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  S.callid = ++c->callid;
  hipLaunchKernelGGL(sz_k_pairs_explicit, dim3(grid_for(S.capM + 1, 256)), dim3(256), 0, c->stream, S, (int)np);
  stage_elems(c, false);
  stage_narrow(c, dt, max_overlap, c->P.fd_max_overlap);
  stage_reduce(c, 0, c->hostN, dt);
  c->inter_any = true; c->inter_lost = false;
  int h[C_COUNT];
  int rc = sync_and_check(c, h);
  if (rc) return rc;
  return host_fuse_fixup(c, h, false);
}

int sz_collide_domain(sz_ctx* c, int32_t dt, double max_overlap) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  c->maybe_tagged = true;
  (void)hipSetDevice(c->device);
  c->gi_valid = false;         // (what follows numbers its ghosts by storage position)
  leave_resident(c);            // a process-mode call: the resident steps' ghost-candidate list is stale, the world rings must be current
  State& S = c->S;
  S.callid = ++c->callid;
  hipLaunchKernelGGL(sz_k_pairs_explicit, dim3(grid_for(S.capM + 1, 256)), dim3(256), 0, c->stream, S, 0);
  stage_elems(c, true);
  stage_narrow(c, dt, c->P.ff_max_overlap, max_overlap);
  stage_reduce(c, 0, c->hostN, dt);
  c->inter_any = true; c->inter_lost = false;
  return sync_and_check(c);
}

int sz_timestep_coupling(sz_ctx* c) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  if (!c->have_fields) { c->err = "sz_set_fields must be called before sz_timestep_coupling"; return SZ_E_STATE; }
  c->maybe_tagged = true;
  (void)hipSetDevice(c->device);
  leave_resident(c);            // a process-mode call: the resident steps' ghost-candidate list is stale, the world rings must be current
  if (c->two_way) { if (c->S.tiled) { c->err = "tiled contexts couple through sz_tile_step + sz_two_way_partial / sz_two_way_finish"; return SZ_E_STATE; } int rc = ensure_two_way(c); if (rc) return rc; }
  if (c->precision == 1 && !c->two_way) { int rc = ensure_mixed(c); if (rc) return rc; }
  if (c->precision == 0 && !c->two_way) { int rc = ensure_block_points(c); if (rc) return rc; }
  stage_forcing(c);
  hipLaunchKernelGGL(sz_k_apply_frc, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S);
  return sync_and_check(c);
}

// ---- precision of the forcings: 0 = fp64 (default), 1 = mixed (per-point arithmetic in fp32, sz_kernels.hpp)
int sz_set_precision(sz_ctx* c, int32_t mode) {
  if (!c || mode < 0 || mode > 1) return SZ_E_ARG;
  if (mode == 0 && c->precision == 1 && c->have_floes) {        // back to fp64: the world rings are the state again
    (void)hipSetDevice(c->device);
    world_rings(c);
    c->S.rec32 = nullptr; c->S.body_rings = 0; c->mixed_geom_ok = false;
  }
  c->precision = mode;
  return SZ_OK;
}
// ---- two-way coupling (coupling.jl:1617-1680; CouplingSettings(two_way_coupling_on = true))
int sz_set_two_way(sz_ctx* c, int32_t on, double Cd_ao, double k, double L, int32_t dt) {
  if (!c) return SZ_E_ARG;
  c->two_way = on != 0; c->P.Cd_ao = Cd_ao; c->P.k_ice = k; c->P.L_ice = L; c->tw_dt = dt;
  return SZ_OK;
}
int sz_set_temps(sz_ctx* c, const double* t_ocn, const double* t_atm) {
  if (!c || !t_ocn || !t_atm) return SZ_E_ARG;
  if (!c->have_fields) { c->err = "sz_set_fields must be called before sz_set_temps"; return SZ_E_STATE; }
  (void)hipSetDevice(c->device);
  int rc = ensure_two_way(c); if (rc) return rc;
  H2D(c->S.t_ocn, t_ocn, c->tw_ncell, double); H2D(c->S.t_atm, t_atm, c->tw_ncell, double);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->temps_set = true;
  return SZ_OK;
}
int sz_download_ocean_stress(sz_ctx* c, double* tau_x, double* tau_y, double* si_frac, double* hflx) {
  if (!c || !c->have_fields) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  int rc = ensure_two_way(c); if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t nb = c->tw_ncell * sizeof(double);
  if (tau_x) HIPCHK(c, hipMemcpy(tau_x, c->S.tau_x, nb, hipMemcpyDeviceToHost));
  if (tau_y) HIPCHK(c, hipMemcpy(tau_y, c->S.tau_y, nb, hipMemcpyDeviceToHost));
  if (si_frac) HIPCHK(c, hipMemcpy(si_frac, c->S.si_frac, nb, hipMemcpyDeviceToHost));
  if (hflx) HIPCHK(c, hipMemcpy(hflx, c->S.hf, nb, hipMemcpyDeviceToHost));
  return SZ_OK;
}

namespace {
int need_interactions(sz_ctx* c) {
  if (!c->inter_lost) return SZ_OK;
  c->err = "the interaction rows of the last collision call were dropped by an upload of another size: "
           "sz_upload_interactions (floe.interactions of every floe, possibly empty) or a collision call must come first";
  return SZ_E_STATE;
}
}  // namespace
int sz_timestep_floe_properties(sz_ctx* c, int32_t dt) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  if (int rc = need_interactions(c)) return rc;
  (void)hipSetDevice(c->device);
  leave_resident(c);            // a process-mode call: the resident steps' ghost-candidate list is stale, the world rings must be current
  stage_integrate(c, dt, true, false);
  return sync_and_check(c);
}

// floe.interactions of every floe replaced by hand (CSR, rows k x 7: floeidx, xforce, yforce, xpoint, ypoint,
// torque, overlap): what the reference's calc_stress! test does before calling it (test_update_floe.jl:27-30)
int sz_upload_interactions(sz_ctx* c, const int32_t* off, const double* rows) {
  if (!c || !c->have_floes || !off) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  tile_cleanup(c);
  State& S = c->S;
  const int M = c->hostM;
  if (off[M] == 0) {           // no floe has interactions (fresh floes): the counts are all there is to say
    HIPCHK(c, hipMemsetAsync(S.inter_cnt, 0, (size_t)M * sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->inter_any = true; c->inter_lost = false;
    return SZ_OK;
  }
  if (!rows) return SZ_E_ARG;
  const int ROWCAP = S.rowcap;
  std::vector<int> cnt(M); std::vector<double> buf((size_t)M * ROWCAP * 7, 0.0);
  for (int i = 0; i < M; i++) {
    int k = off[i + 1] - off[i];
    if (k < 0 || k > ROWCAP) { c->err = "more interaction rows per floe than the fixed stride holds"; return SZ_E_CAPACITY; }
    cnt[i] = k;
    if (k) memcpy(&buf[(size_t)i * ROWCAP * 7], rows + (size_t)off[i] * 7, (size_t)k * 7 * sizeof(double));
  }
  H2D(S.inter_cnt, cnt.data(), M, int);
  H2D(S.inter_rows, buf.data(), (size_t)M * ROWCAP * 7, double);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->inter_any = true; c->inter_lost = false;
  return SZ_OK;
}
// calc_stress! (update_floe.jl:392-414) and calc_strain! (:425-453) on their own, for every floe
int sz_calc_stress(sz_ctx* c) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  if (int rc = need_interactions(c)) return rc;
  (void)hipSetDevice(c->device);
  hipLaunchKernelGGL(sz_k_calc_stress, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, c->P);
  return sync_and_check(c);
}
int sz_calc_strain(sz_ctx* c) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  hipLaunchKernelGGL(sz_k_move_strain, dim3(grid_for(c->S.capM, 16, 8192)), dim3(256), 0, c->stream, c->S, 1, 0, -1);
  return sync_and_check(c);
}

namespace {
// a list of buffers to fill with one launch (sz_k_clear_many); bytes must be a multiple of 4, the value is a byte value as for memset
struct Clears {
  ClearList L{}; unsigned long long maxw = 0; sz_ctx* ctx;
  explicit Clears(sz_ctx* c) : ctx(c) {}
  void add(void* p, size_t bytes, int byte_val = 0) {
    if (L.n == CLEAR_MAX) launch(ctx);          // (a full list goes out; the order of the clears among themselves does not matter)
    const unsigned b = (unsigned)(byte_val & 0xff);
    L.p[L.n] = (unsigned*)p; L.words[L.n] = bytes / 4; L.val[L.n] = b | (b << 8) | (b << 16) | (b << 24);
    maxw = std::max(maxw, L.words[L.n]); L.n++;
  }
  void launch(sz_ctx* c) {
    if (!L.n) return;
    hipLaunchKernelGGL(sz_k_clear_many, dim3(grid_for((long long)maxw, 256, 2048)), dim3(256), 0, c->stream, L);
    L.n = 0; maxw = 0;
  }
};
// ---------------------------------------------------------------- pipelined batches (sz_pipeline.hpp)
// the State of step parity q: everything that is double-buffered points at set q; rows of the step's makers at region q
State pipe_state(sz_ctx* c, int q) {
  State S = c->S;
  const sz_ctx::PipeBuf& B = c->pb[q];
  S.vxy = B.vxy; S.crec = B.crec; S.cell_cnt = B.cell_cnt; S.cell_slots = B.cell_slots; S.cell_ovf = B.cell_ovf; S.cell_items = B.cell_items;
  S.work = B.work; S.wq = B.wq; S.gh = B.gh; S.ngh = B.ngh;
  // rows of a step's makers: set 0 straight behind the parents (as everywhere else), set 1 from a multiple of 16 half way through the spare rows
  const int nr1 = (c->hostN + (S.capM - c->hostN) / 2 + 15) & ~15;
  S.goff = q ? nr1 - c->hostN : 0; S.gcap = q ? S.capM - nr1 : nr1 - c->hostN; S.gslot = q;
  return S;
}
PipeAlt pipe_alt(sz_ctx* c, int q, int make_ghosts) {
  const State T = pipe_state(c, q);
  PipeAlt A;
  A.crec = T.crec; A.vxy = T.vxy; A.cell_cnt = T.cell_cnt; A.cell_slots = T.cell_slots; A.cell_ovf = T.cell_ovf; A.cell_items = T.cell_items;
  A.work = T.work; A.wq = T.wq; A.gh = T.gh; A.ngh = T.ngh; A.goff = T.goff; A.gslot = T.gslot; A.make_ghosts = make_ghosts;
  return A;
}
// make set q the context's own (c->S, crec_buf): where the state lies after a pipelined batch
void pipe_adopt(sz_ctx* c, int q) {
  const State T = pipe_state(c, q);
  State& S = c->S;
  S.vxy = T.vxy; S.cell_cnt = T.cell_cnt; S.cell_slots = T.cell_slots; S.cell_ovf = T.cell_ovf; S.cell_items = T.cell_items;
  S.work = T.work; S.wq = T.wq; S.gh = T.gh; S.ngh = T.ngh;
  c->crec_buf = T.crec; c->gpar = q;
}
bool pipeline_eligible(const sz_ctx* c, int nsteps, bool coll, bool sg, bool gi, bool periodic, bool cr, bool rfree, int flags) {
  return rfree && !c->no_pipeline && coll && sg && (gi || !periodic) && cr && nsteps >= c->pipe_min_steps && c->hostN <= c->pipe_max_floes && c->precision == 0 && !c->two_way &&
         (!c->S.any_domain_work || (!periodic && !c->any_moving)) && c->S.maxnb <= MAXNB && !larger_rings(c) && c->pb[1].vxy && c->pb[1].work &&
         (c->S.capM - c->hostN) / 2 > 64 && !(c->dbg & 8) && (flags & SZ_COLLISIONS_ON);
}

// A batch of pipelined steps: L1(s) = narrow(s) | GEO(s) | forcings(s), L2(s) = VEL(s) | search(s + 1).  Same contract as the loop of sz_step
// it replaces: h = the counter block after the batch, *done = the steps that ran; the context's state is complete when it returns (rows of
// the last step assembled, strain evaluated, ghosts detached).
int step_batch_pipelined(sz_ctx* c, int nsteps, int tstep0, int dt, int coupling_dt, int flags, bool periodic, bool gi, int* h, int* done_out) {
  State& S0 = c->S;
  const int N = c->hostN;
  const bool user_stop = !(flags & SZ_NO_STOP);
  const bool fam = N <= 40000 && periodic;
  const bool elems = S0.any_domain_work != 0;          // (eligible only without a periodic pair: no ghosts, the floe count is the host's)
  const int q0 = c->gpar;
  auto par = [&](int s) { return (q0 + s) & 1; };
  S0.stop_on_tags = user_stop ? 1 : 0; S0.restart_on_tags = user_stop ? 0 : 1;
  S0.ginline = gi ? 1 : 0; S0.famrec = gi ? 1 : 0; S0.pipe = 0;
  S0.facc = c->facc_buf; S0.kexp = force_scale_exp(c);
  bool lean = !c->retry_seen && !c->no_lean_narrow;
  auto leave = [&](int rc) { S0.retry_stop = 0; S0.ginline = 0; S0.famrec = 0; S0.step = 0; S0.crec = nullptr; S0.facc = nullptr; S0.goff = 0; S0.gcap = 0; S0.pipe = 0; S0.restart_on_tags = 0; c->acc_mode = 0; c->reduce_mode = 0; return rc; };
  Clears clr(c);          // (the batch's clears go out with the first prologue's, in one launch)
  clr.add(c->facc_buf, (size_t)FX_WORDS * S0.capM * sizeof(long long));
  clr.add(S0.cnt + C_FRCSTOP, sizeof(int));
  const int callid0 = c->callid; c->callid += nsteps;
  // ---- the prologue of a (sub-)batch that starts at step s: cells, records, ghosts and the neighbour search of that step, from the floes as they lie
  bool first_start = true;
  auto prologue = [&](int s) -> int {
    const int q = par(s);
    pipe_adopt(c, q);                                   // (the geometry of step s is in set q: the context's own from here on)
    S0.step = 0; S0.goff = 0; S0.gcap = 0;
    if (!first_start) c->grid_live = false;             // (a restart: the cells hold ghosts of a step that is started afresh)
    use_static_grid(c);                                 // cells[q] <- the parents (unless they are: the last batch's update binned them)
    const sz_ctx::PipeBuf& O = c->pb[1 - q];
    clr.add(O.cell_cnt, ((size_t)S0.capCells + 1) * sizeof(int));
    clr.add(O.cell_ovf, ((size_t)S0.capCells + 1) * sizeof(int));
    clr.add(c->pb[0].wq, NSEG * 32 * sizeof(int)); clr.add(c->pb[1].wq, NSEG * 32 * sizeof(int));
    clr.add(c->pb[0].ngh, (size_t)S0.capM * sizeof(int));          // (no links: a restart after a tag comes with those GEO made for a step that is now started afresh)
    clr.add(c->pb[1].ngh, (size_t)S0.capM * sizeof(int));
    clr.add(S0.galloc, 32 * sizeof(unsigned long long));
    clr.launch(c);
    // the records of both sets from the columns (the static quads of the twin; its geometry quads are GEO's) -- unless the last batch left them current
    if (!(first_start && c->crec_current))
      for (int b = 0; b < 2; b++) { State T = pipe_state(c, b); T.step = 0; hipLaunchKernelGGL(sz_k_crec_seed, dim3(grid_for(N, 256)), dim3(256), 0, c->stream, T, N); }
    first_start = false;
    State T = pipe_state(c, q); T.step = s + 1; T.callid = callid0 + s + 1; T.retry_stop = lean ? 1 : 0;
    if (gi) hipLaunchKernelGGL(sz_k_ghost_inline_seed, dim3(grid_for(S0.capM, 256)), dim3(256), 0, c->stream, T, q, N);
    const dim3 gr(grid_for(S0.capM, NB_TPB / NB_G, 8192)), bl(NB_TPB);
    if (elems) {          // (between walls: the element items of the step in the tail of its search, as the three-launch steps do)
      const int nbe0 = grid_for(S0.capM, NB_TPB, 1 << 20);
      hipLaunchKernelGGL((sz_k_neighbors_elem<false, true>), dim3(gr.x + nbe0), bl, 0, c->stream, T, next_epoch(c), (int)gr.x);
    } else if (fam) hipLaunchKernelGGL((sz_k_neighbors<true, MAXNB, true>), gr, bl, 0, c->stream, T);
    else hipLaunchKernelGGL((sz_k_neighbors<false, MAXNB, true>), gr, bl, 0, c->stream, T);
    return SZ_OK;
  };
  auto coupling_at = [&](int s) { return (flags & SZ_COUPLING_ON) && coupling_dt > 0 && ((tstep0 + s) % coupling_dt) == 0; };
  // grid of the narrow launch (as stage_narrow)
  constexpr int TPB = 64;
  if (c->narrow_grid0 == 0) {
    int per_cu = 0, cus = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sz_k_narrow<NARROW_G, NARROW_CAP0, NARROW_KC0, NARROW_RC0, 4, TPB, 0, 0, 3, 0>, TPB, 0);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
    c->narrow_grid0 = per_cu > 0 && cus > 0 ? per_cu * cus : 2048;
    if (const char* e = getenv("SZ_NARROW_GRID")) { int v = atoi(e); if (v > 0) c->narrow_grid0 = v; }
  }
  const long long capItems = (long long)S0.capPairs + S0.capElem;
  const int nbn = grid_for(capItems, TPB / NARROW_G, c->narrow_grid0), nbg = grid_for(N, TPB, 1 << 20);
  const int queue = c->no_queue ? 0 : 1;
  auto launch_L1 = [&](int s, bool make_ghosts) {
    State T = pipe_state(c, par(s)); T.step = s + 1; T.callid = callid0 + s + 1; T.retry_stop = lean ? 1 : 0;
    const PipeAlt A = pipe_alt(c, par(s + 1), make_ghosts ? 1 : 0);
    const bool coupling = coupling_at(s);
    const bool overlap = coupling && (c->overlap_forcing >= 0 ? c->overlap_forcing != 0 : N > 65536);
    if (overlap) stage_forcing_fork(c, &T);
    int nbf = coupling && !overlap ? grid_for(S0.capM, TPB / FRC_PLAIN, 32768) : 0;
    const int nbfg = c->frc_first > 0 && nbf > 0 ? std::min(c->frc_first, nbf) : nbf;
    if (c->frc_first > 0 && nbf > 0) nbf = -nbfg;
    if (coupling) c->forcing_where = overlap ? 0 : 2;
    Timed tm(c, SZ_K_NARROW);          // (event-timed classes of a pipelined step: "narrow" = L1, "integrate" = L2)
    if (nbf) hipLaunchKernelGGL((sz_k_narrow<NARROW_G, NARROW_CAP0, NARROW_KC0, NARROW_RC0, 4, TPB, 0, 0, 3, 1, 1>), dim3(nbn + nbg + nbfg), dim3(TPB), 0, c->stream,
                                T, c->P, dt, c->P.ff_max_overlap, c->P.fd_max_overlap, c->dbg, queue, nbf, A, nbg, N);
    else hipLaunchKernelGGL((sz_k_narrow<NARROW_G, NARROW_CAP0, NARROW_KC0, NARROW_RC0, 4, TPB, 0, 0, 3, 0, 1>), dim3(nbn + nbg), dim3(TPB), 0, c->stream,
                            T, c->P, dt, c->P.ff_max_overlap, c->P.fd_max_overlap, c->dbg, queue, 0, A, nbg, N);
    tm.end();
    if (!lean) {          // the largest variant takes what the small one hands on (see stage_narrow)
      hipLaunchKernelGGL((sz_k_narrow<64, NARROW_CAP2, NARROW_KC2, NARROW_RC2, 16, 64, NARROW_CAP1, 2>), dim3(grid_for(capItems, 1, 256)), dim3(64), 0,
                         c->stream, T, c->P, dt, c->P.ff_max_overlap, c->P.fd_max_overlap, c->dbg, queue, 0, PipeAlt{}, 0, 0);
    }
    return overlap;
  };
  auto launch_L2 = [&](int s, bool with_search, bool host_last, bool joined) {
    if (joined) stage_forcing_join(c);
    State T = pipe_state(c, par(s + 1)); T.step = s + 2; T.callid = callid0 + s + 2; T.retry_stop = lean ? 1 : 0;
    const PipeAlt A = pipe_alt(c, par(s), 0);
    const int nbv = grid_for(N, NB_TPB, 1 << 20), nbs = with_search ? grid_for(S0.capM, NB_TPB / NB_G, 8192) : 0;
    const int nbe = with_search && elems ? grid_for(N, NB_TPB, 1 << 20) : 0;
    const unsigned ep = nbe ? next_epoch(c) : 0u;
    const int am = 1 | 4 | (host_last ? 2 : 0);
    Timed tm(c, SZ_K_INTEGRATE);
    if (fam) hipLaunchKernelGGL((sz_k_vel_search<true>), dim3(nbv + nbs + nbe), dim3(NB_TPB), 0, c->stream, T, c->P, A, dt, coupling_at(s) ? 1 : 0, nbv, N, am, nbe, ep);
    else hipLaunchKernelGGL((sz_k_vel_search<false>), dim3(nbv + nbs + nbe), dim3(NB_TPB), 0, c->stream, T, c->P, A, dt, coupling_at(s) ? 1 : 0, nbv, N, am, nbe, ep);
    tm.end();
  };
  // what lies behind the last step `last` (0-based) of the batch: parents un-swapped after a tag stop, strain, the step's rows, rows home, ghosts off
  auto epilogue = [&](int last, bool after_device_stop) -> int {
    const int q = par(last + 1);                          // the geometry of the state that is handed back
    pipe_adopt(c, q);
    if (after_device_stop) {
      State T = pipe_state(c, q); T.step = 0;
      hipLaunchKernelGGL(sz_k_unswap, dim3(grid_for(N, 128)), dim3(128), 0, c->stream, T, pipe_alt(c, 1 - q, 0), N);
      c->grid_live = false;
    }
    State T = pipe_state(c, q); T.step = 0; T.goff = 0;
    hipLaunchKernelGGL(sz_k_move_strain, dim3(grid_for(S0.capM, 16, 8192)), dim3(256), 0, c->stream, T, 1, 0, -1);      // calc_strain! of the state handed back
    // the rows of step `last`: its links and row region are parity par(last)'s; the parents' centroids of that step are in mot
    State R = pipe_state(c, par(last)); R.step = 0; R.vxy = T.vxy;
    hipLaunchKernelGGL(sz_k_inter_fill, dim3(grid_for(S0.capM, 128 / IF_G, 16384)), dim3(128), 0, c->stream, R, 1, N, 0, 1, 2 + last);      // (2 + last: behind, for 1-based step last + 1)
    return SZ_OK;
  };
  int s_end = c->maybe_tagged && user_stop && nsteps > 1 ? 1 : nsteps;
  int s0 = 0, done = 0; bool need_prologue = true;
  for (;;) {
    S0.retry_stop = lean ? 1 : 0;
    if (need_prologue) { int rc = prologue(s0); if (rc) return leave(rc); need_prologue = false; }
    for (int s = s0; s < s_end; s++) {
      const bool host_last = s + 1 == s_end;
      const bool joined = launch_L1(s, !host_last);
      launch_L2(s, !host_last, host_last, joined);
    }
    // the epilogue of the case "all steps ran" goes out with the steps: its launches look at the counters and return when the batch ended early
    // (sz_k_inter_fill: behind-mode guard; the strain launch is harmless either way and is repeated below)
    S0.step = 0;
    { int rc = epilogue(s_end - 1, false); if (rc) return leave(rc); }
    HIPCHK(c, hipMemcpyAsync(h, S0.cnt, C_COUNT * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2));
    if (h[C_ERR] & ~ERR_CAP_INTER) {          // (the rows' stride, ERR_CAP_INTER, is dealt with behind the loop: the bit stays up until then)
      const int bits = h[C_ERR] & ~ERR_CAP_INTER;
      if (growable(bits) && h[C_RETRYSTOP] > 0) {
        // a list of step sr outgrown (neighbours, pair items): the batch paused there before the step changed anything -- larger lists, then that
        // step and the rest again (as sz_step)
        int z = 0; (void)hipMemcpy(S0.cnt + C_ERR, &z, sizeof(int), hipMemcpyHostToDevice);
        const int sr = h[C_RETRYSTOP] - 1;
        pipe_adopt(c, par(sr));
        int rc = grow_lists(c, bits & (ERR_CAP_NEIGH | ERR_CAP_PAIRS)); if (rc) return leave(rc);
        (void)hipMemsetAsync(S0.cnt + C_RETRYSTOP, 0, sizeof(int), c->stream); (void)hipMemsetAsync(S0.cnt + C_STOP, 0, sizeof(int), c->stream);
        (void)hipMemsetAsync(S0.cnt + C_FRCSTOP, 0, sizeof(int), c->stream);
        (void)hipMemsetAsync(c->facc_buf, 0, (size_t)FX_WORDS * S0.capM * sizeof(long long), c->stream);
        s0 = sr; need_prologue = true;
        continue;
      }
      (void)sync_and_check(c, h);          // (sets the error text, clears the word)
      return leave(SZ_E_CAPACITY);
    }
    if (lean && h[C_RETRYSTOP] > 0) {
      // paused inside step sr: an item for the largest narrow variant.  That variant on the step's own State, the step's second launch again, on
      // with the steps behind it (the variant stays in from now on)
      const int sr = h[C_RETRYSTOP] - 1;
      c->retry_seen = true; lean = false;
      (void)hipMemsetAsync(S0.cnt + C_RETRYSTOP, 0, sizeof(int), c->stream); (void)hipMemsetAsync(S0.cnt + C_PAUSED, 0, sizeof(int), c->stream);
      State T = pipe_state(c, par(sr)); T.step = sr + 1; T.callid = callid0 + sr + 1; T.retry_stop = 0;
      hipLaunchKernelGGL((sz_k_narrow<64, NARROW_CAP2, NARROW_KC2, NARROW_RC2, 16, 64, NARROW_CAP1, 2>), dim3(grid_for(capItems, 1, 256)), dim3(64), 0,
                         c->stream, T, c->P, dt, c->P.ff_max_overlap, c->P.fd_max_overlap, c->dbg, queue, 0, PipeAlt{}, 0, 0);
      const bool host_last = sr + 1 == s_end;
      S0.retry_stop = 0;
      launch_L2(sr, !host_last, host_last, false);
      s0 = sr + 1;
      if (s0 >= s_end) {          // it was the last step: only the epilogue is left
        S0.step = 0;
        { int rc = epilogue(s_end - 1, false); if (rc) return leave(rc); }
        HIPCHK(c, hipMemcpyAsync(h, S0.cnt, C_COUNT * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (h[C_ERR] & ~ERR_CAP_INTER) { (void)sync_and_check(c, h); return leave(SZ_E_CAPACITY); }
        done = s_end;
        break;
      }
      continue;
    }
    if (h[C_STOP] > 0 && h[C_STOP] < s_end) {
      // a tag ended the enqueued steps after step k: the state behind it (GEO(k) has run ahead: parents un-swapped), then either the end of
      // the batch (the caller's stop) or -- a batch that runs through -- the rest of it, started like a batch (the ghosts know the tag now)
      const int k = h[C_STOP] - 1;
      if (user_stop) { int rc = epilogue(k, true); if (rc) return leave(rc); done = k + 1; HIPCHK(c, hipMemcpyAsync(h, S0.cnt, C_COUNT * sizeof(int), hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); break; }
      pipe_adopt(c, par(k + 1));
      { State T = pipe_state(c, par(k + 1)); T.step = 0; hipLaunchKernelGGL(sz_k_unswap, dim3(grid_for(N, 128)), dim3(128), 0, c->stream, T, pipe_alt(c, par(k), 0), N); }
      (void)hipMemsetAsync(S0.cnt + C_STOP, 0, sizeof(int), c->stream); (void)hipMemsetAsync(S0.cnt + C_FRCSTOP, 0, sizeof(int), c->stream);
      s0 = k + 1; need_prologue = true;
      continue;
    }
    if (s_end < nsteps && h[C_STOP] == 0) {          // the first step ran on its own (a parent might have been tagged before the batch): the rest
      s0 = s_end; s_end = nsteps; need_prologue = true;
      continue;
    }
    done = h[C_STOP] > 0 ? std::min(h[C_STOP], nsteps) : s_end;
    break;
  }
  // ---- the batch is over: state in set par(done); rows of step done - 1 assembled (region par(done - 1))
  const int qlast = par(done - 1);
  if (h[C_ERR] & ERR_CAP_INTER) {          // a floe of the last step has more rows than the stride holds: more room, the launch again (sz_step does the same)
    int z = 0; (void)hipMemcpy(S0.cnt + C_ERR, &z, sizeof(int), hipMemcpyHostToDevice);
    for (int tries = 0; tries < 6; tries++) {
      S0.rowcap *= 4;
      if (S0.rowcap > 8192) { c->err = "a floe has more than 8192 interaction rows"; return leave(SZ_E_CAPACITY); }
      int rc = carve_interactions(c); if (rc) return leave(rc);
      c->inter_lost = false;
      State R = pipe_state(c, qlast); R.step = 0; R.vxy = S0.vxy;
      hipLaunchKernelGGL(sz_k_inter_fill, dim3(grid_for(S0.capM, 128 / IF_G, 16384)), dim3(128), 0, c->stream, R, 1, N, 0, 1, 1);
      HIPCHK(c, hipMemcpyAsync(h, S0.cnt, C_COUNT * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (!(h[C_ERR] & ERR_CAP_INTER)) break;
      (void)hipMemcpy(S0.cnt + C_ERR, &z, sizeof(int), hipMemcpyHostToDevice);
    }
    if (h[C_ERR]) { (void)sync_and_check(c, h); return leave(SZ_E_CAPACITY); }
  }
  {          // the per-row results of the last step to the rows straight behind the parents; its links become the context's; ghosts off
    const State R = pipe_state(c, qlast);
    const int G = h[C_NGHOSTS];
    if (R.goff != 0 && G > 0) {
      State T = c->S; T.gh = R.gh; T.ngh = R.ngh; T.step = 0;
      hipLaunchKernelGGL(sz_k_rows_home, dim3(grid_for((long long)G * S0.maxnb, 256)), dim3(256), 0, c->stream, T, N, G, R.goff);
      hipLaunchKernelGGL(sz_k_rows_rename, dim3(grid_for((long long)(N + G) * S0.maxnb, 256, 8192)), dim3(256), 0, c->stream, T, N, G, R.goff);
    }
    // (the links of the last step are set qlast's; the context's own set is par(done)'s: sz_k_remove_ghosts saves and clears what it is given)
    State T = c->S; T.gh = R.gh; T.ngh = R.ngh; T.step = 0; T.retry_stop = 0;
    hipLaunchKernelGGL(sz_k_remove_ghosts, dim3(grid_for(S0.capM, 256)), dim3(256), 0, c->stream, T, 0);
    Clears tail(c);
    tail.add(c->pb[1 - qlast].ngh, (size_t)S0.capM * sizeof(int));
    tail.add(c->pb[1 - qlast].gh, (size_t)MAX_GHOSTS * S0.capM * sizeof(int), 0xff);
    tail.launch(c);
  }
  {
    const int keepG = h[C_NGHOSTS];
    int rc = sync_and_check(c, h);
    h[C_NGHOSTS] = keepG;
    if (rc) return leave(rc);
  }
  if (done < nsteps) c->grid_live = false;
  *done_out = done;
  c->last_pipelined = 1; c->gi_pending_slot = qlast;
  c->crec_current = done == nsteps;          // (the records of the adopted set follow the columns; after a tag stop the un-swap rewrote a few: seeded again next time)
  // (h[C_STOP]: the caller's view -- a batch that ran through ended at nsteps)
  if (!user_stop) h[C_STOP] = 0;
  return leave(SZ_OK);
}
}  // namespace

int sz_debug_pipelined(sz_ctx* c) { return c ? c->last_pipelined : 0; }

int sz_step(sz_ctx* c, int32_t nsteps, int32_t tstep0, int32_t dt, int32_t coupling_dt, int32_t flags, int32_t* steps_done) {
  if (steps_done) *steps_done = 0;
  if (!c || !c->have_floes) return SZ_E_STATE;
  if (nsteps < 0) return SZ_E_ARG;
  if ((flags & SZ_COUPLING_ON) && !c->have_fields) { c->err = "sz_set_fields must be called before coupling"; return SZ_E_STATE; }
  (void)hipSetDevice(c->device);
  const bool periodic = c->S.any_periodic_ew || c->S.any_periodic_ns;
  const bool coll = (flags & SZ_COLLISIONS_ON) != 0;
  if (c->two_way && (flags & SZ_COUPLING_ON)) {
    if (c->S.tiled) { c->err = "tiled contexts couple through sz_tile_step + sz_two_way_partial / sz_two_way_finish"; return SZ_E_STATE; }
    if (!c->temps_set) { c->err = "two-way coupling needs sz_set_temps (after the sz_set_fields that fixed the lattice shape)"; return SZ_E_STATE; }
    int rc = ensure_two_way(c); if (rc) return rc;
  }
  if (!coll) { if (int rc = need_interactions(c)) return rc; }
  // the batch ends after the first step that leaves a parent tagged remove / fuse (simplify_floes!, simulation.jl:205-214,
  // is the host's): the launches of the later steps are enqueued all the same and return at once (stopped())
  c->S.stop_on_tags = (flags & SZ_NO_STOP) ? 0 : 1;
  HIPCHK(c, hipMemsetAsync(c->S.cnt + C_STOP, 0, sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(c->S.cnt + C_RETRYSTOP, 0, sizeof(int), c->stream));
  bool last_coupled = false;
  const bool sg = coll && c->grid_ok && !c->no_static_grid;
  if (sg) use_static_grid(c);
  const bool gl = ghost_list_wanted(c, sg);
  // (ghosts a process-mode sz_add_ghosts left attached are dropped first: the list pass only visits the parents that get new ones)
  if (gl && periodic && c->hostM != c->hostN) hipLaunchKernelGGL(sz_k_remove_ghosts, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, 0);
  // inline ghosts: no ghost launch in the steps at all (the integrator makes the next step's ghosts; needs the one-launch integrator)
  const bool gi = gl && c->ghost_inline && !c->S.tiled && c->fused_move && c->max_ring <= MV_RING;
  if (gl && !gi) use_ghost_list(c); else c->gl_valid = false;
  const int gl0 = c->gl_cur;
  c->S.ginline = gi ? 1 : 0;
  if (gi) c->S.famrec = 1;
  if (c->gi_pending && c->gi_valid && !coll) { int rc = gi_fetch(c); if (rc) return rc; }      // (the key tables are about to be reused)
  c->gi_pending = false;
  if (coll) c->gi_valid = false;    // (this batch's rows replace the old ones; set again below if they carry order keys of inline ghosts)
  const bool mixed = c->precision == 1 && !c->two_way;
  if (mixed) { int rc = ensure_mixed(c); if (rc) return rc; }
  if (c->precision == 0 && !c->two_way && c->have_fields) { int rc = ensure_block_points(c); if (rc) return rc; }
  // mixed precision: the steps run on body-frame rings (the integrator moves poses, not rings) when nothing else in the batch
  // needs world rings -- single context, the list path for the ghosts, rings small enough for the fused integrator
  const bool body = mixed && coll && sg && (gl || !periodic) && !c->S.tiled && c->fused_move && c->max_ring <= MV_RING && !c->no_body_rings;
  if (!body) world_rings(c);
  c->S.body_rings = body ? 1 : 0;
  // collision records (State::crec): in batches whose kernels keep them current -- the one-launch integrator, and for periodic walls the
  // inline ghost maker; seeded here from the columns (before the ghost seed: the maker updates the records of the parents it visits)
  const bool cr = coll && sg && !c->no_crec && c->crec_buf && c->fused_move && c->max_ring <= MV_RING && (gi || !periodic) && nsteps > 0;
  c->S.crec = cr ? c->crec_buf : nullptr; c->crec_was_live = cr;
  // pipelined batches (sz_pipeline.hpp: two launches per step) run their own prologue -- records, first ghosts, first neighbour search
  c->last_pipelined = 0;
  const bool pipe = pipeline_eligible(c, nsteps, coll, sg, gi, periodic, cr,
                                      coll && c->facc_buf != nullptr && !c->no_reduce_free && sg && (gi || !periodic) && c->fused_move && c->max_ring <= MV_RING && !c->any_moving, flags);
  if (!pipe) c->crec_current = false;          // (the three-launch steps seed the records they use; they may not keep the twin set's)
  if (cr && !pipe) hipLaunchKernelGGL(sz_k_crec_seed, dim3(grid_for(c->hostN, 256)), dim3(256), 0, c->stream, c->S, c->hostN);
  if (gi && !pipe) {               // the ghosts of the first step, from the parents as they lie (after the rings are in the batch's form)
    HIPCHK(c, hipMemsetAsync(c->S.galloc, 0, 32 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(sz_k_ghost_inline_seed, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, 0, c->hostN);
  }
  // the largest narrow variant only takes items the small one hands on (none in most fields): it is left out of the steps until one
  // shows up -- the batch then pauses inside that step (stopped_late()) and is finished below
  bool lean = coll && !c->retry_seen && !c->no_lean_narrow && !c->S.tiled && !larger_rings(c);
  // Fixed-point totals (State::facc; sz_geom.hpp): the narrow phase adds every row to both floes' totals, the integrator reads them -- no reduce
  // launch inside the steps.  floe.interactions of the step that ended the batch is assembled once, behind the batch (stage_reduce(.., behind)):
  // that needs the ghosts of that step still in their rows, i.e. the one-launch integrator with inline ghosts (or no periodic wall), which
  // knows when it runs a batch's last step (sz_k_integrate: last_step).  The other paths keep the launch inside the step, rows only.
  const bool facc_on = coll && c->facc_buf != nullptr;
  const bool rfree = facc_on && !c->no_reduce_free && sg && (gi || !periodic) && c->fused_move && c->max_ring <= MV_RING && !c->any_moving;
  c->S.facc = facc_on ? c->facc_buf : nullptr; c->S.kexp = force_scale_exp(c);
  c->reduce_mode = !facc_on ? 0 : rfree ? 2 : 1;
  if (facc_on && !pipe) {          // (a pipelined batch clears them with the rest of its prologue: one launch)
    HIPCHK(c, hipMemsetAsync(c->facc_buf, 0, (size_t)FX_WORDS * c->S.capM * sizeof(long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->S.cnt + C_FRCSTOP, 0, sizeof(int), c->stream));
  }
  // a parent that is already tagged ends the batch after its first step, and the integrator only finds out while it runs (the tags of a
  // step itself are raised by its narrow phase / forcings, a launch earlier): that first step is then enqueued on its own, as a last step
  int s_end = rfree && c->maybe_tagged && c->S.stop_on_tags && nsteps > 1 ? 1 : nsteps;
  auto leave = [&]() { c->S.retry_stop = 0; c->S.body_rings = 0; c->S.ginline = 0; c->S.famrec = 0; c->S.step = 0; c->S.crec = nullptr; c->S.facc = nullptr; c->acc_mode = 0; c->reduce_mode = 0; };
  int h[C_COUNT];
  int pipe_done = -1;
  if (pipe) {
    c->S.stop_on_tags = (flags & SZ_NO_STOP) ? 0 : 1;
    int rcp = step_batch_pipelined(c, nsteps, tstep0, dt, coupling_dt, flags, periodic, gi, h, &pipe_done);
    c->S.stop_on_tags = (flags & SZ_NO_STOP) ? 0 : 1;
    if (rcp) { leave(); return rcp; }
  }
  const int callid0 = c->callid; if (!pipe) c->callid += nsteps;          // (step s of this batch is collision call callid0 + s + 1, also when it is run again)
  for (int s0 = 0, mid = 0; !pipe;) {
    c->S.retry_stop = lean ? 1 : 0;
    for (int s = s0; s < s_end; s++) {
      int tstep = tstep0 + s;
      c->S.step = s + 1; c->S.callid = callid0 + s + 1;
      const bool resume = mid && s == s0;
      const bool coupling = (flags & SZ_COUPLING_ON) && coupling_dt > 0 && (tstep % coupling_dt) == 0;
      const bool overlap = coupling && !c->two_way && (c->overlap_forcing >= 0 ? c->overlap_forcing != 0 : (c->hostN > 65536 && c->precision == 0 && coll));
      // with collisions on, the ghosts of step s are detached by the ghost kernels of step s+1 (nothing
      // in between looks past the parents) and committed by the bounds kernel: two launches less
      // The forcings only read the floes' state at the start of the step (after the ghost pass has wrapped the parents that left the
      // domain) and write columns nothing reads before the update: they can run beside the collision kernels (second stream, fork
      // after the ghost pass) or inside one of their launches.
      // (riding in the neighbour launch pays while both kernels leave the chip idle: measured better up to 40 k floes,
      // neutral at 100 k dense, worse at 100 k sparse -- there the forcings get their own launch)
      const bool fuse = coupling && !overlap && coll && !c->two_way && !(c->pmask >> SZ_K_FORCING & 1u) && c->fuse_forcing && c->hostN <= 65536;
      int fmode = !fuse ? 0 : c->fuse_forcing_mode ? c->fuse_forcing_mode : (c->hostN <= 30000 ? 2 : 1);
      if (fmode == 1 && c->S.maxnb > MAXNB) fmode = 2;      // (the neighbour + forcing launch exists for the default neighbour capacity only)
      if (!resume) {          // (a paused step has all of this behind it)
        if (coll && !gi) stage_ghosts(c, true, sg, gl);
        // (after the ghost pass, like the reference's timestep_coupling!: a parent that has just swapped with its ghost is sampled where it
        //  now lies -- the same lattice values as at its image, but the interpolation weights come from other coordinates)
        if (coupling && !overlap && !fuse) stage_forcing(c, dt);
        if (overlap) stage_forcing_fork(c);
        if (coupling) c->forcing_where = fmode;
      }
      c->S.gslot = s & 1;
      // (the totals of the rows the inline makers allocated for this step are cleared by its neighbour search when it runs on collision records;
      //  without them -- SZ_CREC=0 -- here)
      if (facc_on && gi && !(cr && c->S.maxnb <= MAXNB) && !resume) (void)hipMemsetAsync(c->facc_buf + (size_t)FX_WORDS * c->hostN, 0, (size_t)FX_WORDS * (c->S.capM - c->hostN) * sizeof(long long), c->stream);
      if (coll) collisions_step(c, c->hostN, dt, periodic && !sg, sg, resume ? 0 : fmode, lean, resume);
      if (overlap && !resume) stage_forcing_join(c);
      // (inline ghosts: the last step of the batch makes none -- there is no step to make them for, and the cell lists stay the parents')
      c->acc_mode = !facc_on ? 0 : 1 | (rfree ? 4 | (s + 1 == s_end ? 2 : 0) : 0);
      stage_integrate(c, dt, !coll, coupling, sg, gl && !gi ? 1 - c->gl_cur : -1, gi && s + 1 < s_end ? 1 - (s & 1) : -1);
      if (gl && !gi) c->gl_cur ^= 1;
    }
    c->S.step = 0;
    // floe.interactions of the step that ended the batch (reduce-free steps): one launch for the whole batch
    if (rfree) stage_reduce(c, 1, c->hostN, dt, c->hostN + 3 * c->gl_est + c->hostN / 64 + 32, true);
    if (coll && periodic) hipLaunchKernelGGL(sz_k_remove_ghosts, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, 0);
    int rc = sync_and_check(c, h);
    if (rfree && rc == SZ_E_CAPACITY && c->last_err_bits == ERR_CAP_INTER && h[C_RETRYSTOP] == 0) {
      // a floe of that last step has more rows than the stride holds: only the rows' memory grows (collisions.jl:290-296), and the launch runs
      // again on the parents with the ghost links sz_k_remove_ghosts has put aside
      rc = SZ_OK;
      for (int tries = 0; tries < 6; tries++) {
        c->S.rowcap *= 4;
        if (c->S.rowcap > 8192) { c->err = "a floe has more than 8192 interaction rows"; leave(); return SZ_E_CAPACITY; }
        if ((rc = carve_interactions(c))) { leave(); return rc; }
        c->inter_lost = false;
        State S2 = c->S; S2.ngh = c->S.ngh_save; S2.gh = c->S.gh_save;
        hipLaunchKernelGGL(sz_k_inter_fill, dim3(grid_for(S2.capM, 128 / IF_G, 16384)), dim3(128), 0, c->stream, S2, 1, c->hostN, 0, 1, 1);
        int h2[C_COUNT];
        rc = sync_and_check(c, h2);
        if (!(rc == SZ_E_CAPACITY && c->last_err_bits == ERR_CAP_INTER)) break;
      }
    }
    if (rc == SZ_E_CAPACITY && growable(c->last_err_bits) && h[C_RETRYSTOP] > 0 && coll) {
      // A list outgrown inside step h[C_RETRYSTOP] (neighbours per floe, pair items, rows per floe): the batch paused there before anything
      // of the floes' state changed (capacity_stop()).  Larger lists, then that step and the rest of the batch again, from the parents as
      // they lie -- exactly as a batch that starts at that step would (cells, the step's ghosts): the reference's lists grow (collisions.jl:290-296).
      if ((rc = grow_lists(c, c->last_err_bits))) { leave(); return rc; }
      s0 = h[C_RETRYSTOP] - 1; mid = 0;
      (void)hipMemsetAsync(c->S.cnt + C_RETRYSTOP, 0, sizeof(int), c->stream);
      (void)hipMemsetAsync(c->S.cnt + C_STOP, 0, sizeof(int), c->stream);
      if (facc_on) {          // (the narrow phase of the step that is run again adds its rows again)
        (void)hipMemsetAsync(c->facc_buf, 0, (size_t)FX_WORDS * c->S.capM * sizeof(long long), c->stream);
        (void)hipMemsetAsync(c->S.cnt + C_FRCSTOP, 0, sizeof(int), c->stream);
      }
      c->grid_live = false; use_static_grid(c);
      if (cr) hipLaunchKernelGGL(sz_k_crec_seed, dim3(grid_for(c->hostN, 256)), dim3(256), 0, c->stream, c->S, c->hostN);
      if (gi) {
        (void)hipMemsetAsync(c->S.galloc, 0, 32 * sizeof(unsigned long long), c->stream);
        c->S.gslot = s0 & 1;
        hipLaunchKernelGGL(sz_k_ghost_inline_seed, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, s0 & 1, c->hostN);
      } else if (gl) { c->gl_valid = false; use_ghost_list(c); }
      continue;
    }
    if (rc) { leave(); return rc; }
    if ((!lean || h[C_RETRYSTOP] == 0) && s_end < nsteps && h[C_STOP] == 0) {
      // the first step ran on its own (a parent might have been tagged already) and nothing ended the batch: the rest of it, from the floes
      // as they lie -- the cells hold the parents (the step made no ghosts), the ghosts of the next step are seeded as at a batch's start
      s0 = s_end; s_end = nsteps; mid = 0;
      if (gi) {
        (void)hipMemsetAsync(c->S.galloc, 0, 32 * sizeof(unsigned long long), c->stream);
        c->S.gslot = s0 & 1;
        hipLaunchKernelGGL(sz_k_ghost_inline_seed, dim3(grid_for(c->S.capM, 256)), dim3(256), 0, c->stream, c->S, s0 & 1, c->hostN);
      }
      continue;
    }
    if (!lean || h[C_RETRYSTOP] == 0) break;
    // paused after the narrow launch of step h[C_RETRYSTOP]: that variant is in from now on
    c->retry_seen = true; lean = false;
    s0 = h[C_RETRYSTOP] - 1; mid = 1;
    if (gl && !gi) c->gl_cur = (gl0 + s0) & 1;
    (void)hipMemsetAsync(c->S.cnt + C_RETRYSTOP, 0, sizeof(int), c->stream);
  }
  c->S.retry_stop = 0;
  c->S.ginline = 0; c->S.famrec = 0; c->S.crec = nullptr; c->S.facc = nullptr; c->acc_mode = 0; c->reduce_mode = 0;
  if (coll && (h[C_STOP] > 0 || (flags & SZ_NO_STOP))) c->maybe_tagged = true;
  if (body && nsteps > 0) c->rings_stale = true;
  c->S.body_rings = 0;
  if (coll) { c->inter_any = true; c->inter_lost = false; }
  int rc = SZ_OK;
  const int done = pipe ? pipe_done : h[C_STOP] > 0 ? std::min(h[C_STOP], (int)nsteps) : nsteps;
  if (steps_done) *steps_done = done;
  if (gl && !gi) {        // the list the last step that RAN has filled, and how long it is
    c->gl_cur = (gl0 + done) & 1;
    c->gl_est = h[C_NGCAND + c->gl_cur];
  }
  if (gi) {               // the order keys of the last step's ghosts are what the host needs to number them as the reference does:
    // they are fetched when somebody asks for numbers (gi_fetch: downloads, the fuse replay) -- most batches end without
    c->gi_pending_n = done > 0 ? h[C_NGHOSTS] : 0; if (!pipe) c->gi_pending_slot = (done - 1) & 1; c->gi_pending = true;
    if (coll) c->gi_valid = done > 0;
    if (done < nsteps) c->grid_live = false;       // stopped early: the step that ended the batch has binned ghosts for a step that did not come
    c->gl_est = std::max(c->gl_est, c->gi_pending_n);        // (sizes the list pass should the next batch use it)
  }
  // status.fuse_idx of the step that ended the batch: the reference's serial propagation, replayed on the host as
  // sz_timestep_collisions does (only that step can have produced fuse pairs: the batch stops on the first tag)
  if (coll && done > 0) {
    const int tlast = tstep0 + done - 1;
    last_coupled = (flags & SZ_COUPLING_ON) && coupling_dt > 0 && (tlast % coupling_dt) == 0;
    if (h[C_STOP] > 0 || (flags & SZ_NO_STOP)) {
      rc = host_fuse_fixup(c, h, true, true, last_coupled);
      c->fuse_lists.resize(c->hostM);
      c->gl_valid = false;          // the replay may have changed status tags
    }
  }
  return rc;
}

int sz_profile_enable(sz_ctx* c, int32_t on) {
  if (!c) return SZ_E_ARG;
  c->pmask = on == 1 ? ~0u : on > 1 ? (unsigned)on >> 1 : 0u;
  return SZ_OK;
}
// the instantiation of the dominant kernel as the kernel trace names it (the first narrow variant; its last template argument says whether
// the step's forcings rode in the launch in the last batch): what a profile reader has to look for, derived from the code that ran
int sz_narrow_kernel_name(sz_ctx* c, char* buf, int32_t n) {
  if (!c || !buf || n < 8) return SZ_E_ARG;
  const int frc = c->forcing_where == 2 ? (c->precision == 1 ? 2 : 1) : 0;
  snprintf(buf, (size_t)n, "sz_k_narrow<%d,%d,%d,%d,4,64,0,0,3,%d,%d>", NARROW_G, NARROW_CAP0, NARROW_KC0, NARROW_RC0, frc, c->last_pipelined ? 1 : 0);
  return SZ_OK;
}
int sz_forcing_launch(sz_ctx* c, int32_t* where) {
  if (!c || !where) return SZ_E_ARG;
  *where = c->forcing_where;
  return SZ_OK;
}
int sz_profile_reset(sz_ctx* c) {
  if (!c) return SZ_E_ARG;
  for (int k = 0; k < NK; k++) { c->kms[k] = 0; c->kl[k] = 0; }
  c->ev_used = 0;
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipMemsetAsync(c->S.acc, 0, (size_t)ACC_SLOTS * 8 * sizeof(unsigned long long), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SZ_OK;
}
int sz_kernel_time_ms(sz_ctx* c, int32_t k, double* ms, int64_t* launches) {
  if (!c || k < 0 || k >= NK) return SZ_E_ARG;
  if (ms) *ms = c->kms[k];
  if (launches) *launches = c->kl[k];
  return SZ_OK;
}


// ---------------------------------------------------------------- multi-GPU halo API
int sz_tile_enable(sz_ctx* c, const int64_t* gidx, double halo_capacity_factor, double max_rmax) {
  if (!c || !c->have_floes || !gidx) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  leave_resident(c);
  State& S = c->S;
  if (c->hostM != c->hostN) { c->err = "sz_tile_enable needs a ghost-free upload"; return SZ_E_STATE; }
  std::vector<long long>& ok = c->tile_gidx;
  ok.assign(c->hostN, 0);
  for (int i = 0; i < c->hostN; i++) ok[i] = gidx[i];
  H2D(S.okey, ok.data(), c->hostN, long long);
  if (c->facc_buf) HIPCHK(c, hipMemsetAsync(c->facc_buf, 0, (size_t)FX_WORDS * S.capM * sizeof(long long), c->stream));      // (rows change hands in a migration: no stale totals)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  S.tiled = 1;
  // largest ring among ALL ranks' floes (halo floes arrive unseen): decides which narrow variants can be needed
  c->max_ring_tiled = halo_capacity_factor > 0 ? (int)halo_capacity_factor : HALO_RING;
  // the halo records have room for the largest ring of ANY rank's floes (Floe rings are unbounded, floe.jl:24-77; the engine's narrow phase
  // takes 255 points, and so do the tiles): 12 + 2 * halo_ring doubles per record
  if (c->max_ring_tiled > NARROW_CAP2) { c->err = "a ring has more than 255 points: beyond the narrow phase's largest variant"; return SZ_E_CAPACITY; }
  S.halo_ring = std::max(HALO_RING, (std::max(c->max_ring_tiled, c->max_ring) + 3) & ~3);
  // largest rmax among ALL ranks' floes: the static broad-phase grid must hold for halo floes too (0: unknown ->
  // the grid is fitted to the centroids every step instead)
  c->rmax_hint = max_rmax; c->rmax_max = max_rmax > 0 ? c->rmax_max : 0.0;
  setup_grid(c);
  return SZ_OK;
}

int sz_owned_box(sz_ctx* c, double* out5) {
  if (!c || !c->have_floes || !out5) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  hipLaunchKernelGGL(sz_k_owned_box, dim3(1), dim3(1024), 0, c->stream, c->S, c->S.bounds + 8, (const double*)nullptr, 0.0, 0.0, 0, 0);
  HIPCHK(c, hipMemcpyAsync(out5, c->S.bounds + 8, 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SZ_OK;
}

int sz_halo_record_doubles(void) { return HALO_REC; }
int sz_halo_record_doubles_ctx(sz_ctx* c) { return c ? halo_rec(c->S) : HALO_REC; }

// diagnostic build only: cycles per narrow-phase stage, summed over groups (zeros otherwise)
// test hook: quads of the collision records (State::crec) of the owned parents that differ from the columns they cache; *n_bad = -1 when the last
// resident batch did not run on records
int sz_debug_crec_mismatches(sz_ctx* c, int64_t* n_bad) {
  if (!c || !c->have_floes || !n_bad) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  if (!c->crec_buf || !c->crec_was_live) { *n_bad = -1; return SZ_OK; }
  int h[C_COUNT];
  int rc = sync_and_check(c, h); if (rc) return rc;
  unsigned long long* d = (unsigned long long*)(c->S.cnt + C_COUNT + 64 + 68);      // (two spare words of the counter block)
  HIPCHK(c, hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(sz_k_crec_check, dim3(grid_for(c->hostN, 256)), dim3(256), 0, c->stream, c->S, (const double2*)c->crec_buf, c->hostN, d);
  unsigned long long v = 0;
  HIPCHK(c, hipMemcpyAsync(&v, d, sizeof(v), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_bad = (int64_t)v;
  return SZ_OK;
}
// diagnosis: the ghost / halo row that carried order key `key` in the last resident step that used ghost allocator `slot` (step s of a batch,
// 0-based: s & 1), as the collision kernels saw it: out[0] = row (-1: none), cx, cy, u, v, xi, rmax, area, height, box, ring points, parent,
// status, ring x (20) and y (20) -- 56 doubles
int sz_debug_find_key(sz_ctx* c, int32_t slot, int64_t key, double* out56) {
  if (!c || !c->have_floes || !out56 || slot < 0 || slot > 1) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  double* d = nullptr;
  HIPCHK(c, hipMalloc((void**)&d, 56 * sizeof(double)));
  hipLaunchKernelGGL(sz_k_debug_find_key, dim3(1), dim3(64), 0, c->stream, c->S, slot, (long long)key, c->hostN, d);
  HIPCHK(c, hipMemcpyAsync(out56, d, 56 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  return SZ_OK;
}
// diagnosis: the pair items of the last resident step (ghost allocator `slot`) between the instances of two floe ids: out[0] = entries (at most 12
// returned), then {owner key, partner key, contact rows, owner row, partner row} each -- 61 doubles
int sz_debug_pairs_of_ids(sz_ctx* c, int32_t slot, int64_t id_a, int64_t id_b, double* out61) {
  if (!c || !c->have_floes || !out61 || slot < 0 || slot > 1) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  double* d = nullptr;
  HIPCHK(c, hipMalloc((void**)&d, 61 * sizeof(double)));
  HIPCHK(c, hipMemsetAsync(d, 0, 61 * sizeof(double), c->stream));
  hipLaunchKernelGGL(sz_k_debug_pairs_of_ids, dim3(1), dim3(64), 0, c->stream, c->S, slot, (long long)id_a, (long long)id_b, c->hostN, d, 12);
  HIPCHK(c, hipMemcpyAsync(out61, d, 61 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  return SZ_OK;
}
int sz_debug_stamps(sz_ctx* c, long long* out16) {
  if (!c || !c->have_floes || !out16) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipMemcpy(out16, c->S.stamps, (512 + 8 * 8000) * sizeof(long long), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemset(c->S.stamps, 0, (512 + 8 * 8000) * sizeof(long long)));
  return SZ_OK;
}

// boxes: nranks x {xmin, xmax, ymin, ymax}, already expanded by the interaction range (rarely changes)
int sz_halo_set_boxes(sz_ctx* c, int32_t nranks, const double* boxes) {
  if (!c || !c->have_floes || nranks < 1 || nranks > 64 || !boxes) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipMemcpyAsync(c->S.bounds + 16, boxes, (size_t)nranks * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SZ_OK;
}

// asynchronous: fills d_send (nranks regions of (cap + 1) records, record 0 = header with the count)
int sz_halo_pack(sz_ctx* c, int32_t nranks, int32_t me, double Lx, double Ly, int32_t per_x, int32_t per_y, void* d_send,
                 int32_t cap) {
  if (!c || !c->have_floes || nranks < 1 || nranks > 64 || cap < 1) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  State& S = c->S;
  int* dcnt = S.cnt + C_COUNT;                  // 64 ints reserved behind the counter block
  hipLaunchKernelGGL(sz_k_halo_pack, dim3(grid_for(std::max(c->hostN, 1), PACK_TPB)), dim3(PACK_TPB), 0, c->stream, S, nranks, me, S.bounds + 16, Lx, Ly,
                     per_x, per_y, (double*)d_send, cap, dcnt, (const int*)nullptr, (const double*)nullptr, 0.0);
  return SZ_OK;
}

// asynchronous: unpack d_recv (same layout, region r = records from rank r) and run one timestep_sim!
// on owned + halo floes; only owned floes are integrated, the halo is dropped afterwards
namespace {
int tile_forcing(sz_ctx* c) {
  if (!c->have_fields) { c->err = "sz_set_fields must be called before coupling"; return SZ_E_STATE; }
  if (c->two_way) { int rc = ensure_two_way(c); if (rc) return rc; }
  else if (c->precision == 1) { int rc = ensure_mixed(c); if (rc) return rc; }
  else { int rc = ensure_block_points(c); if (rc) return rc; }
  stage_forcing(c);
  return SZ_OK;
}
}  // namespace

int sz_tile_step(sz_ctx* c, const void* d_recv, int32_t nranks, int32_t cap, int32_t tstep, int32_t dt, int32_t coupling_dt,
                 int32_t flags) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  State& S = c->S;
  const bool coll = (flags & SZ_COLLISIONS_ON) != 0;
  const bool sg = coll && c->grid_ok && !c->no_static_grid;
  if (sg) use_static_grid(c);
  const bool gl = ghost_list_wanted(c, sg);
  if (gl) use_ghost_list(c); else c->gl_valid = false;
  if (d_recv && nranks > 0) {
    // halo floes join the candidate list of THIS step (the owned floes were appended by the last integrator)
    hipLaunchKernelGGL(sz_k_halo_unpack, dim3(1), dim3(1024), 0, c->stream, S, (const double*)d_recv, nranks, cap, sg ? 1 : 0, gl ? c->gl_cur : -1);
  }
  const bool coupling = (flags & SZ_COUPLING_ON) && coupling_dt > 0 && (tstep % coupling_dt) == 0;
  const bool periodic = S.any_periodic_ew || S.any_periodic_ns;
  // the forcings of this step: already enqueued by sz_tile_forcing (beside the exchange), else now -- in either
  // case before the ghost pass, like sz_step
  if (coupling && c->tile_forcing_tstep != tstep) { int rc = tile_forcing(c); if (rc) return rc; }
  c->tile_forcing_tstep = -1;
  // As in sz_step, the ghosts of the previous step are detached by this step's flag kernel and the new ones
  // committed by the flag/scan kernel; the halo of the previous step was overwritten by the unpack kernel.  Nothing
  // between two steps looks past the owned floes, so no clean-up launch is needed per step: the ghosts and halo
  // floes of the LAST step are dropped when the host next looks at the state (tile_cleanup).
  // n_init = every local parent (owned + halo): totals of halo floes are computed and then ignored
  S.callid = ++c->callid;
  // (fixed-point totals as in the resident steps of sz_step, so that a tile and the single context give the same bits; the reduce launch stays
  //  inside the step here, assembling rows only)
  const bool facc_on = coll && c->facc_buf != nullptr;
  S.facc = facc_on ? c->facc_buf : nullptr; S.kexp = force_scale_exp(c); c->reduce_mode = facc_on ? 1 : 0; c->acc_mode = facc_on ? 1 : 0;
  if (coll) stage_ghosts(c, true, sg, gl);
  if (coll) collisions(c, -1, dt, periodic && !sg, sg);
  stage_integrate(c, dt, false, coupling, sg, gl ? 1 - c->gl_cur : -1);
  S.facc = nullptr; c->reduce_mode = 0; c->acc_mode = 0;
  if (gl) { c->gl_cur ^= 1; c->gl_est = std::max(c->gl_est, 64); }
  c->tile_dirty = true;
  return SZ_OK;
}

// Two-way coupling across tiles.  After a tiled coupling step: sz_two_way_partial writes this rank's per-cell sums
// (3 x (Nx+1)(Ny+1) doubles: stress numerators x / y, ice area) to a DEVICE buffer of the caller, the caller adds
// the buffers of all ranks up (all-reduce), sz_two_way_finish turns the sums into the ocean fields on every rank.
int sz_two_way_partial(sz_ctx* c, void* d_partial) {
  if (!c || !c->have_floes || !c->two_way || !d_partial) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  const int ncell = (int)c->tw_ncell;
  hipLaunchKernelGGL(sz_k_tw_partial, dim3(grid_for(ncell, 256)), dim3(256), 0, c->stream, c->S, ncell, (double*)d_partial);
  return SZ_OK;
}
int sz_two_way_finish(sz_ctx* c, const void* d_partial, int32_t dt) {
  if (!c || !c->have_floes || !c->two_way || !d_partial) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  const int ncell = (int)c->tw_ncell;
  hipLaunchKernelGGL(sz_k_tw_finish, dim3(grid_for(ncell, 256)), dim3(256), 0, c->stream, c->S, c->P, ncell, dt, (const double*)d_partial);
  return SZ_OK;
}

// ---------------------------------------------------------------- output path

// shared front of the grid-output calls: argument checks, grid lines to the device, cell areas
int eul_grid(sz_ctx* c, int32_t nx, int32_t ny, const double* xg, const double* yg, PoolGuard& pool, EulGrid& E) {
  if (nx < 1 || ny < 1 || !xg || !yg) return SZ_E_ARG;
  const double dx = xg[1] - xg[0], dy = yg[1] - yg[0];
  if (!(dx > 0) || !(dy > 0)) { c->err = "grid lines must ascend"; return SZ_E_ARG; }
  for (int k = 0; k <= nx; k++) if (fabs(xg[k] - (xg[0] + k * dx)) > 1e-6 * dx) { c->err = "x grid lines must be evenly spaced"; return SZ_E_ARG; }
  for (int k = 0; k <= ny; k++) if (fabs(yg[k] - (yg[0] + k * dy)) > 1e-6 * dy) { c->err = "y grid lines must be evenly spaced"; return SZ_E_ARG; }
  (void)hipSetDevice(c->device);
  world_rings(c);
  int rc = sync_and_check(c);          // hostM current, nothing pending
  if (rc) return rc;
  const int ncell = nx * ny;
  E = EulGrid{};
  E.nx = nx; E.ny = ny; E.M = c->hostM > 0 ? c->hostM : 1;
  double *d_xg, *d_yg;
  if ((rc = dalloc(c, &d_xg, nx + 1, pool.v)) || (rc = dalloc(c, &d_yg, ny + 1, pool.v)) || (rc = dalloc(c, &E.count, 1, pool.v)) ||
      (rc = dalloc(c, &E.cell_area, ncell, pool.v)) || (rc = dalloc(c, &E.data, (size_t)EUL_COUNT * ncell, pool.v))) return rc;
  H2D(d_xg, xg, nx + 1, double); H2D(d_yg, yg, ny + 1, double);
  E.xg = d_xg; E.yg = d_yg;
  hipLaunchKernelGGL(sz_k_eul_cell_area, dim3(grid_for(ncell, 64 / EU_G, 8192)), dim3(64), 0, c->stream, c->S, E);
  return SZ_OK;
}
// entries (count, size, fill, sort) and the area of every entry
int eul_entries(sz_ctx* c, PoolGuard& pool, EulGrid& E, int& nent) {
  State& S = c->S;
  int rc;
  hipLaunchKernelGGL(sz_k_eul_entries, dim3(grid_for(S.capM, 256)), dim3(256), 0, c->stream, S, E, 0);
  nent = 0;
  HIPCHK(c, hipMemcpyAsync(&nent, E.count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (nent < 0) { c->err = "output grid: entry count overflow"; return SZ_E_CAPACITY; }
  unsigned long long* keys_in = nullptr;
  E.cap = nent;
  if ((rc = dalloc(c, &keys_in, nent, pool.v)) || (rc = dalloc(c, &E.keys, nent, pool.v)) || (rc = dalloc(c, &E.pic, nent, pool.v))) return rc;
  if (nent == 0) return SZ_OK;
  HIPCHK(c, hipMemsetAsync(E.count, 0, sizeof(int), c->stream));
  EulGrid Ein = E; Ein.keys = keys_in;
  hipLaunchKernelGGL(sz_k_eul_entries, dim3(grid_for(S.capM, 256)), dim3(256), 0, c->stream, S, Ein, 1);
  unsigned long long top = (unsigned long long)E.nx * E.ny * (unsigned long long)E.M;
  unsigned bits = 1; while (bits < 64 && (top >> bits)) bits++;
  size_t tmp_bytes = 0;
  HIPCHK(c, rocprim::radix_sort_keys(nullptr, tmp_bytes, keys_in, E.keys, (size_t)nent, 0u, bits, c->stream));
  char* tmp = nullptr;
  if ((rc = dalloc(c, &tmp, tmp_bytes, pool.v))) return rc;
  HIPCHK(c, rocprim::radix_sort_keys((void*)tmp, tmp_bytes, keys_in, E.keys, (size_t)nent, 0u, bits, c->stream));
  if (S.nelem > 4) hipLaunchKernelGGL(sz_k_eul_area, dim3(grid_for(nent, 64 / EU_G, 1 << 16)), dim3(64), 0, c->stream, S, E, nent);
  else hipLaunchKernelGGL(sz_k_eul_area_rect, dim3(grid_for(nent, 256, 1 << 16)), dim3(256), 0, c->stream, S, E, nent);
  return SZ_OK;
}
int eul_check_outputs(sz_ctx* c, int32_t nout, const int32_t* outputs, const double* data) {
  if (nout < 0 || (nout > 0 && (!outputs || !data))) return SZ_E_ARG;
  for (int k = 0; k < nout; k++) if (outputs[k] < 0 || outputs[k] >= EUL_COUNT) { c->err = "unknown grid output"; return SZ_E_ARG; }
  return SZ_OK;
}
int eul_copy_out(sz_ctx* c, const EulGrid& E, int32_t nout, const int32_t* outputs, double* data) {
  const size_t ncell = (size_t)E.nx * E.ny;
  for (int k = 0; k < nout; k++)
    HIPCHK(c, hipMemcpyAsync(data + (size_t)k * ncell, E.data + (size_t)outputs[k] * ncell, ncell * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return sync_and_check(c);
}

// calc_eulerian_data! (output.jl:793-914) on the rows the context holds (parents and, if the caller ran
// sz_add_ghosts, ghosts -- write_data! runs after add_ghosts!, simulation.jl:102-105)
int sz_eulerian_data(sz_ctx* c, int32_t nx, int32_t ny, const double* xg, const double* yg, int32_t nout,
                     const int32_t* outputs, double* data) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  int rc = eul_check_outputs(c, nout, outputs, data);
  if (rc) return rc;
  if (c->S.tiled) { c->err = "tiled contexts: sz_eulerian_partial, all-reduce, sz_eulerian_finish"; return SZ_E_STATE; }
  PoolGuard pool; EulGrid E; int nent = 0;
  if ((rc = eul_grid(c, nx, ny, xg, yg, pool, E)) || (rc = eul_entries(c, pool, E, nent))) return rc;
  hipLaunchKernelGGL(sz_k_eul_reduce, dim3(grid_for(nx * ny, 64)), dim3(64), 0, c->stream, c->S, E, nent);
  return eul_copy_out(c, E, nout, outputs, data);
}
// tiled runs, first half: this rank's per-cell sums into d_partial (SZ_EUL_PARTIAL * nx * ny doubles on the device)
int sz_eulerian_partial(sz_ctx* c, int32_t nx, int32_t ny, const double* xg, const double* yg, void* d_partial) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  if (!d_partial) return SZ_E_ARG;
  PoolGuard pool; EulGrid E; int nent = 0, rc;
  if ((rc = eul_grid(c, nx, ny, xg, yg, pool, E)) || (rc = eul_entries(c, pool, E, nent))) return rc;
  hipLaunchKernelGGL(sz_k_eul_partial, dim3(grid_for(nx * ny, 64)), dim3(64), 0, c->stream, c->S, E, nent, (double*)d_partial);
  return sync_and_check(c);
}
// second half: the summed buffer -> the averages
int sz_eulerian_finish(sz_ctx* c, int32_t nx, int32_t ny, const double* xg, const double* yg, const void* d_partial, int32_t nout,
                       const int32_t* outputs, double* data) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  if (!d_partial) return SZ_E_ARG;
  int rc = eul_check_outputs(c, nout, outputs, data);
  if (rc) return rc;
  PoolGuard pool; EulGrid E;
  if ((rc = eul_grid(c, nx, ny, xg, yg, pool, E))) return rc;
  hipLaunchKernelGGL(sz_k_eul_finish, dim3(grid_for(nx * ny, 64)), dim3(64), 0, c->stream, E, (const double*)d_partial);
  return eul_copy_out(c, E, nout, outputs, data);
}

// what simplify_floes! (simplification.jl:339-378) would find to do, without downloading a floe
int sz_simplify_check(sz_ctx* c, int32_t max_vertices, double min_floe_area, double min_floe_height, int64_t* out4) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  if (!out4) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  PoolGuard pool;
  unsigned long long* d = nullptr;
  int rc = dalloc(c, &d, 4, pool.v);
  if (rc) return rc;
  hipLaunchKernelGGL(sz_k_simplify_check, dim3(grid_for(c->S.capM, 256, 1024)), dim3(256), 0, c->stream, c->S, max_vertices,
                     min_floe_area, min_floe_height, d);
  unsigned long long h[4];
  HIPCHK(c, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  rc = sync_and_check(c);
  if (rc) return rc;
  for (int k = 0; k < 4; k++) out4[k] = (int64_t)h[k];
  return SZ_OK;
}

// ASYNC: the forcings of step `tstep` (owned floes only; they need nothing from the halo), to be enqueued between
// sz_halo_pack and the collective so that they run beside the exchange; sz_tile_step(tstep) then skips them
int sz_tile_forcing(sz_ctx* c, int32_t tstep, int32_t coupling_dt, int32_t flags) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  const bool coupling = (flags & SZ_COUPLING_ON) && coupling_dt > 0 && (tstep % coupling_dt) == 0;
  if (!coupling) return SZ_OK;
  int rc = tile_forcing(c); if (rc) return rc;
  c->tile_forcing_tstep = tstep;
  return SZ_OK;
}

// ---------------------------------------------------------------- the halo exchange inside the library (RCCL over xGMI)
// "library owns device buffers, streams, RCCL communicators inside the opaque sz_ctx".  A host that is not
// Python (the reference's is Julia: one process per GPU, e.g. under MPI.jl) drives a tiled run with
//     sz_comm_unique_id (rank 0)  ->  the 128 bytes to every rank by any host channel  ->  sz_comm_init
//     sz_upload_floes (the owned floes) / sz_tile_enable  ->  sz_tile_setup  ->  sz_tile_run(nsteps) on every rank.
// Per step: pack kernel -> grouped ncclSend / ncclRecv with the NEIGHBOUR tiles only (the all-to-all-v of the halo records;
// a peer's region carries its real count in the header record and is sized per pair from the counts at the last box gather)
// on a second stream, beside the forcings of the owned floes -> unpack + the ordinary step.  The boxes are gathered again
// (ncclAllGather) every `rebox_every` steps; a floe that out-runs the drift margin in between raises ERR_HALO_DRIFT.
// RCCL is bound at run time (dlopen: the library has no link-time dependency on it, and a process that already holds an
// RCCL -- torch's -- shares that copy).
namespace {
struct UId { char b[128]; };
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(UId*) = nullptr;
  int (*CommInitRank)(void**, int, UId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
constexpr int NCCL_INT32 = 2, NCCL_FLOAT64 = 8, NCCL_SUM = 0;
bool rccl_load(std::string& err) {
  if (g_rccl.h) return true;
  if (getenv("SZ_RCCL_DISABLE")) { err = "RCCL binding switched off (SZ_RCCL_DISABLE)"; return false; }     // (to rehearse the callers' fallback)
  // an RCCL the process already holds comes first (a host framework's: two RCCL builds in one process each bring their own
  // runtime threads), then the system's
  const char* names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1" };
  for (const char* n : names) if ((g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!g_rccl.h) for (const char* n : names) if ((g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!g_rccl.h) { err = std::string("RCCL not found: ") + dlerror(); return false; }
#define RSYM(field, name) g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.h, name); if (!g_rccl.field) { err = std::string("RCCL symbol missing: ") + name; g_rccl.h = nullptr; return false; }
  RSYM(GetUniqueId, "ncclGetUniqueId") RSYM(CommInitRank, "ncclCommInitRank") RSYM(CommDestroy, "ncclCommDestroy")
  RSYM(Send, "ncclSend") RSYM(Recv, "ncclRecv") RSYM(AllGather, "ncclAllGather") RSYM(AllReduce, "ncclAllReduce")
  RSYM(GroupStart, "ncclGroupStart") RSYM(GroupEnd, "ncclGroupEnd") RSYM(GetErrorString, "ncclGetErrorString")
#undef RSYM
  return true;
}
#define NCCLCHK(ctx, call)                                                                             \
  do {                                                                                                 \
    int r_ = (call);                                                                                   \
    if (r_ != 0) { (ctx)->err = std::string(#call) + ": " + g_rccl.GetErrorString(r_); return SZ_E_HIP; } \
  } while (0)

#define HOSTCHK(ctx, call, what)                                                                        \
  do {                                                                                                 \
    int r_ = (call);                                                                                   \
    if (r_ != 0) { (ctx)->err = std::string("host transport: ") + what + " returned " + std::to_string(r_); return SZ_E_HIP; } \
  } while (0)

// all-gather of `bytes` per rank between device buffers on the context's stream (host transport: through the host, synchronous)
int comm_allgather(sz_ctx* c, const void* d_src, void* d_dst, size_t count, int nccl_type, size_t elem) {
  const int n = c->comm_n;
  if (n == 1) { HIPCHK(c, hipMemcpyAsync(d_dst, d_src, count * elem, hipMemcpyDeviceToDevice, c->stream)); return SZ_OK; }
  if (!c->host_transport) { NCCLCHK(c, g_rccl.AllGather(d_src, d_dst, count, nccl_type, c->comm, c->stream)); return SZ_OK; }
  std::vector<char> hs(count * elem), hr(count * elem * n);
  HIPCHK(c, hipMemcpyAsync(hs.data(), d_src, hs.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HOSTCHK(c, c->host_tr.allgather(c->host_tr.user, hs.data(), hr.data(), (int64_t)hs.size()), "allgather");
  HIPCHK(c, hipMemcpyAsync(d_dst, hr.data(), hr.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SZ_OK;
}

// Collective: the OR of a word over the ranks.  Device errors (capacity bits, halo drift) are per rank and sticky; a rank that returned
// on its own while its peers went on into the next collective would leave them waiting forever (RCCL has no timeout).  Every point
// at which sz_tile_run looks at the error word therefore agrees on it first: all ranks return the same code at the same step.
int comm_agree_bits(sz_ctx* c, int local, int* all) {
  const int n = c->comm_n;
  *all = local;
  if (n == 1) return SZ_OK;
  int* d = (int*)(c->d_gather + 8 + 8 * 64 + 64 * 64 / 2);       // the 64 spare doubles behind the count matrix: word | words of all ranks
  HIPCHK(c, hipMemcpyAsync(d, &local, sizeof(int), hipMemcpyHostToDevice, c->stream));
  int rc = comm_allgather(c, d, d + 32, 1, NCCL_INT32, sizeof(int));
  if (rc) return rc;
  int h[64];
  HIPCHK(c, hipMemcpyAsync(h, d + 32, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int bits = 0, who = -1;
  for (int r = 0; r < n; r++) { if (h[r] && who < 0) who = r; bits |= h[r]; }
  *all = bits;
  if (bits && !local) {
    char buf[200];
    snprintf(buf, sizeof(buf), "rank %d of the tiled run reported device error bits 0x%x (this rank is clean; all ranks stop together)", who, bits);
    c->err = buf;
  }
  return SZ_OK;
}
// sync + sticky device errors of THIS rank + agreement: SZ_OK on every rank or the same error code on every rank
// one int of every rank (n <= 64), on every rank
int comm_gather_int(sz_ctx* c, int local, int* all64) {
  const int n = c->comm_n;
  all64[0] = local;
  if (n == 1) return SZ_OK;
  int* d = (int*)(c->d_gather + 8 + 8 * 64 + 64 * 64 / 2);
  HIPCHK(c, hipMemcpyAsync(d, &local, sizeof(int), hipMemcpyHostToDevice, c->stream));
  int rc = comm_allgather(c, d, d + 32, 1, NCCL_INT32, sizeof(int));
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(all64, d + 32, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SZ_OK;
}
// The two words that decide how a tiled batch goes on -- the step a tag ended it at (C_STOP) and the step that paused for the largest narrow
// variant or a list that outgrew its capacity (C_RETRYSTOP) -- as ALL ranks must see them before anyone branches: the smallest non-zero
// value of each.  A rank's own counters are not enough: a pause on one rank and a tag on another in the SAME step are not heard by either
// (the unpack kernels of the next step return at their stop test before they read the peers' headers), and ranks that then take
// different branches wait for each other in different collectives.
int comm_agree_steps(sz_ctx* c, int stop_local, int pause_local, int* stop_all, int* pause_all) {
  int a[64], b[64];
  int rc = comm_gather_int(c, stop_local, a); if (rc) return rc;
  rc = comm_gather_int(c, pause_local, b); if (rc) return rc;
  int s = 0, p = 0;
  for (int r = 0; r < c->comm_n; r++) { if (a[r] > 0 && (s == 0 || a[r] < s)) s = a[r]; if (b[r] > 0 && (p == 0 || b[r] < p)) p = b[r]; }
  *stop_all = s; *pause_all = p;
  return SZ_OK;
}
int tile_sync_agree(sz_ctx* c, int* cnt_out = nullptr) {
  int rc = sync_and_check(c, cnt_out);
  if (rc == SZ_E_HIP) return rc;              // (the runtime itself failed: nothing to agree on)
  int all = 0;
  const int rc2 = comm_agree_bits(c, rc ? (c->last_err_bits ? c->last_err_bits : 1) : 0, &all);
  if (rc2) return rc2;
  return all ? SZ_E_CAPACITY : SZ_OK;
}

// do the expanded box of rank d and the (margin-expanded) owned box of rank s meet, periodic images included?
bool tiles_adjacent(const double* owned_s, const double* expanded_d, double margin, double Lx, double Ly, int per_x, int per_y) {
  for (int kx = (per_x ? -1 : 0); kx <= (per_x ? 1 : 0); kx++)
    for (int ky = (per_y ? -1 : 0); ky <= (per_y ? 1 : 0); ky++) {
      const double x0 = owned_s[0] - margin + kx * Lx, x1 = owned_s[1] + margin + kx * Lx;
      const double y0 = owned_s[2] - margin + ky * Ly, y1 = owned_s[3] + margin + ky * Ly;
      if (!(x1 < expanded_d[0] || expanded_d[1] < x0 || y1 < expanded_d[2] || expanded_d[3] < y0)) return true;
    }
  return false;
}

// collective: owned boxes of all ranks -> expanded boxes on the device, neighbour relation, per-pair slot counts, buffers,
// reference positions of the drift check.  Synchronises (it runs once per rebox_every steps).
int tile_rebox(sz_ctx* c) {
  State& S = c->S;
  const int n = c->comm_n, me = c->comm_rank;
  int rc = tile_sync_agree(c); if (rc) return rc;
  // (every gather after the first: centroids at their periodic image nearest to the centre of the last box -- d_gather[8 + 8 me ..] still holds it)
  double* d_ctr = c->d_gather + 8 + 8 * 64 + 64 * 64 / 2 + 48;          // two of the spare doubles
  if (c->tile_box_valid) {
    const double ctr[2] = { c->tile_box_ctr[0], c->tile_box_ctr[1] };
    HIPCHK(c, hipMemcpyAsync(d_ctr, ctr, sizeof(ctr), hipMemcpyHostToDevice, c->stream));
  }
  hipLaunchKernelGGL(sz_k_owned_box, dim3(1), dim3(1024), 0, c->stream, S, c->d_gather, c->tile_box_valid ? (const double*)d_ctr : (const double*)nullptr,
                     c->tile_Lx, c->tile_Ly, c->tile_per_x, c->tile_per_y);
  constexpr int GB = 8;      // doubles per rank in the gather: box, rmax, drift, speed, (spare)
  std::vector<double> all((size_t)GB * n);
  if ((rc = comm_allgather(c, c->d_gather, c->d_gather + 8, GB, NCCL_FLOAT64, sizeof(double)))) return rc;
  HIPCHK(c, hipMemcpyAsync(all.data(), c->d_gather + 8, all.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->tile_box_ctr[0] = 0.5 * (all[GB * me] + all[GB * me + 1]); c->tile_box_ctr[1] = 0.5 * (all[GB * me + 2] + all[GB * me + 3]); c->tile_box_valid = true;
  double rmax = 0.0, drift = 0.0, speed = 0.0;
  for (int r = 0; r < n; r++) { rmax = std::max(rmax, all[GB * r + 4]); drift = std::max(drift, all[GB * r + 5]); speed = std::max(speed, all[GB * r + 6]); }
  // the gather interval follows the floes (every rank computes the same number): at the faster of the measured displacement per step since
  // the last gather and the largest velocity component now, they may use 30 % of the margin before the next gather (half of it is the
  // error threshold); the interval at most doubles from one gather to the next, and a new setup starts with a short one -- floes that
  // start from rest are slower in their first steps than later
  {
    const double per_step = std::max(c->tile_since_box > 0 ? drift / c->tile_since_box : 0.0, speed * std::fabs((double)c->tile_dt));
    int want = per_step > 0.0 ? (int)std::max(1.0, std::min((double)c->tile_rebox_every, 0.3 * c->tile_margin / per_step)) : c->tile_rebox_every;
    if (c->tile_since_box > 0) want = std::min(want, 2 * c->tile_rebox_cur);
    else want = std::min(want, c->tile_rebox_cur);
    c->tile_rebox_cur = c->tile_rebox_fixed ? c->tile_rebox_every : std::max(1, want);
  }
  const double reach = 2.0 * rmax + c->tile_margin;
  std::vector<double> boxes((size_t)4 * n);
  for (int r = 0; r < n; r++) { boxes[4 * r] = all[GB * r] - reach; boxes[4 * r + 1] = all[GB * r + 1] + reach; boxes[4 * r + 2] = all[GB * r + 2] - reach; boxes[4 * r + 3] = all[GB * r + 3] + reach; }
  HIPCHK(c, hipMemcpyAsync(S.bounds + 16, boxes.data(), boxes.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  // counting pass, then the count matrix of all ranks (row s: what s sends to every d)
  int* dcnt = S.cnt + C_COUNT;
  hipLaunchKernelGGL(sz_k_halo_pack, dim3(grid_for(std::max(c->hostN, 1), PACK_TPB)), dim3(PACK_TPB), 0, c->stream, S, n, me, S.bounds + 16, c->tile_Lx, c->tile_Ly, c->tile_per_x,
                     c->tile_per_y, (double*)nullptr, 1, dcnt, (const int*)nullptr, (const double*)nullptr, 0.0);
  int* d_mat = (int*)(c->d_gather + 8 + GB * 64);
  std::vector<int> mat((size_t)n * n);
  if ((rc = comm_allgather(c, dcnt, d_mat, (size_t)n, NCCL_INT32, sizeof(int)))) return rc;
  HIPCHK(c, hipMemcpyAsync(mat.data(), d_mat, mat.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // slots per ordered pair: neighbours get 1.5 x the present count + 32, the others nothing (every rank computes the same table)
  c->cap_send.assign(n, 0); c->cap_recv.assign(n, 0);
  int cap = 32;
  for (int s = 0; s < n; s++)
    for (int d = 0; d < n; d++) {
      if (s == d) continue;
      const bool adj = tiles_adjacent(&all[GB * s], &boxes[4 * d], c->tile_margin, c->tile_Lx, c->tile_Ly, c->tile_per_x, c->tile_per_y);
      const int k = adj || mat[(size_t)s * n + d] > 0 ? mat[(size_t)s * n + d] * 3 / 2 + 32 : 0;
      if (s == me) c->cap_send[d] = k;
      if (d == me) c->cap_recv[s] = k;
      cap = std::max(cap, k);
    }
  if (cap > c->halo_cap || !c->d_send) {
    reset_pool(c->comm_allocs);          // (chunks that are large enough are carved again: a set-up after a migration allocates nothing)
    c->halo_cap = cap;
    const size_t nd = (size_t)n * (cap + 1) * halo_rec(c->S);
    if ((rc = dalloc(c, &c->d_send, nd, c->comm_allocs)) || (rc = dalloc(c, &c->d_recv, nd, c->comm_allocs)) ||
        (rc = dalloc(c, &c->d_ref, (size_t)2 * S.capM, c->comm_allocs)) || (rc = dalloc(c, &c->d_dcap, 64, c->comm_allocs))) return rc;
    trim_pool(c->comm_allocs);
  }
  // regions of ranks that send nothing keep a zero count in their header record
  HIPCHK(c, hipMemsetAsync(c->d_recv, 0, (size_t)n * (c->halo_cap + 1) * halo_rec(c->S) * sizeof(double), c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_dcap, c->cap_send.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_ref, S.cx, (size_t)c->hostN * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_ref + S.capM, S.cy, (size_t)c->hostN * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->tile_since_box = 0;
  return SZ_OK;
}
}  // namespace

// can the RCCL binding be made in this process (run-time loading)?  Hosts ask on EVERY rank and agree on the answer over their own
// channel before the collective sz_comm_init: a rank that cannot bind would leave the others waiting inside ncclCommInitRank
int sz_comm_available(void) {
  std::string err;
  return rccl_load(err) ? SZ_OK : SZ_E_STATE;
}
int sz_comm_unique_id(void* id128) {
  std::string err;
  if (!id128 || !rccl_load(err)) return SZ_E_STATE;
  return g_rccl.GetUniqueId((UId*)id128) == 0 ? SZ_OK : SZ_E_HIP;
}
int sz_comm_init(sz_ctx* c, int32_t nranks, int32_t rank, const void* id128) {
  if (!c || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || (nranks > 1 && !id128)) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  (void)sz_comm_destroy(c);
  if (nranks > 1) {
    if (!rccl_load(c->err)) return SZ_E_STATE;
    UId id; memcpy(&id, id128, sizeof(id));
    NCCLCHK(c, g_rccl.CommInitRank(&c->comm, nranks, id, rank));
  }
  c->comm_n = nranks; c->comm_rank = rank;
  HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_recv, hipEventDisableTiming));
  return SZ_OK;
}
// the host's own channel instead of RCCL (include/subzero_hip.h: sz_host_transport)
int sz_comm_init_host(sz_ctx* c, int32_t nranks, int32_t rank, const sz_host_transport* t) {
  if (!c || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return SZ_E_ARG;
  if (nranks > 1 && (!t || !t->allgather || !t->sendrecv || !t->allreduce_sum_f64)) { c->err = "sz_comm_init_host: the transport needs all three collectives"; return SZ_E_ARG; }
  (void)hipSetDevice(c->device);
  (void)sz_comm_destroy(c);
  if (nranks > 1) { c->host_tr = *t; c->host_transport = true; }
  c->comm_n = nranks; c->comm_rank = rank;
  HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_recv, hipEventDisableTiming));
  return SZ_OK;
}
// One-rank self test of the RCCL binding (the build box has one GPU, so the multi-rank exchange cannot run there): the
// run-time binding, ncclGetUniqueId / ncclCommInitRank with the id passed by value, an all-gather, an all-reduce and a
// grouped send / receive to self on the communication stream with the event hand-shake sz_tile_run uses.  Returns SZ_OK
// when every buffer holds what it should.
int sz_comm_selftest(sz_ctx* c) {
  if (!c) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  if (!rccl_load(c->err)) return SZ_E_STATE;
  UId id;
  NCCLCHK(c, g_rccl.GetUniqueId(&id));
  void* comm = nullptr;
  NCCLCHK(c, g_rccl.CommInitRank(&comm, 1, id, 0));
  hipStream_t cs = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
  HIPCHK(c, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  HIPCHK(c, hipEventCreateWithFlags(&e0, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  const int n = 4096;
  double* d = nullptr;
  HIPCHK(c, hipMalloc((void**)&d, (size_t)4 * n * sizeof(double)));
  std::vector<double> h((size_t)4 * n, 0.0);
  for (int k = 0; k < n; k++) h[k] = 1.0 + k;
  HIPCHK(c, hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(e0, c->stream));
  HIPCHK(c, hipStreamWaitEvent(cs, e0, 0));
  int rc = SZ_OK;
  NCCLCHK(c, g_rccl.GroupStart());
  NCCLCHK(c, g_rccl.Send(d, (size_t)n, NCCL_FLOAT64, 0, comm, cs));
  NCCLCHK(c, g_rccl.Recv(d + n, (size_t)n, NCCL_FLOAT64, 0, comm, cs));
  NCCLCHK(c, g_rccl.GroupEnd());
  HIPCHK(c, hipEventRecord(e1, cs));
  HIPCHK(c, hipStreamWaitEvent(c->stream, e1, 0));
  NCCLCHK(c, g_rccl.AllGather(d + n, d + 2 * n, (size_t)n, NCCL_FLOAT64, comm, c->stream));
  NCCLCHK(c, g_rccl.AllReduce(d + 2 * n, d + 3 * n, (size_t)n, NCCL_FLOAT64, NCCL_SUM, comm, c->stream));
  HIPCHK(c, hipMemcpyAsync(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < n && rc == SZ_OK; k++)
    if (h[n + k] != 1.0 + k || h[2 * n + k] != 1.0 + k || h[3 * n + k] != 1.0 + k) { c->err = "RCCL self test: wrong data"; rc = SZ_E_HIP; }
  (void)hipFree(d); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(cs);
  (void)g_rccl.CommDestroy(comm);
  return rc;
}
int sz_comm_destroy(sz_ctx* c) {
  if (!c) return SZ_E_ARG;
  if (c->comm) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
  if (c->comm_stream) { (void)hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
  if (c->ev_packed) { (void)hipEventDestroy(c->ev_packed); c->ev_packed = nullptr; }
  if (c->ev_recv) { (void)hipEventDestroy(c->ev_recv); c->ev_recv = nullptr; }
  c->host_transport = false; c->host_tr = sz_host_transport{ nullptr, nullptr, nullptr, nullptr };
  c->comm_n = 0; c->d_send = c->d_recv = c->d_ref = nullptr; c->d_dcap = nullptr; c->halo_cap = 0; c->tile_since_box = -1;
  if (c->d_gather) { (void)hipFree(c->d_gather); c->d_gather = nullptr; }
  free_pool(c->comm_allocs);
  return SZ_OK;
}
// sum of n doubles in device memory over all ranks, in place, on the context's stream (per-cell partial sums of the
// two-way coupling and of the grid output)
int sz_comm_allreduce(sz_ctx* c, void* d_buf, int64_t n) {
  if (!c || !d_buf || n < 0 || c->comm_n < 1) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  if (c->comm_n > 1 && c->host_transport) {
    std::vector<double> h((size_t)n);
    HIPCHK(c, hipMemcpyAsync(h.data(), d_buf, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HOSTCHK(c, c->host_tr.allreduce_sum_f64(c->host_tr.user, h.data(), n), "allreduce_sum_f64");
    HIPCHK(c, hipMemcpyAsync(d_buf, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  } else if (c->comm_n > 1) NCCLCHK(c, g_rccl.AllReduce(d_buf, d_buf, (size_t)n, NCCL_FLOAT64, NCCL_SUM, c->comm, c->stream));
  return SZ_OK;
}
int sz_tile_setup(sz_ctx* c, double Lx, double Ly, int32_t per_x, int32_t per_y, double drift_margin, int32_t rebox_every) {
  if (!c || !c->have_floes || !c->S.tiled || c->comm_n < 1 || !(drift_margin > 0) || rebox_every == 0) {
    if (c) c->err = "sz_tile_setup needs sz_upload_floes, sz_tile_enable and sz_comm_init first, a positive drift margin and rebox interval";
    return SZ_E_STATE;
  }
  (void)hipSetDevice(c->device);
  c->tile_Lx = Lx; c->tile_Ly = Ly; c->tile_per_x = per_x; c->tile_per_y = per_y; c->tile_margin = drift_margin; c->tile_rebox_every = std::abs(rebox_every); c->tile_rebox_fixed = rebox_every < 0;
  c->tile_since_box = -1; c->halo_cap = 0; c->d_send = nullptr; c->tile_rebox_cur = rebox_every < 0 ? -rebox_every : std::min(rebox_every, 8);
  c->tile_box_valid = false;
  if (!c->d_gather) {          // own box | all boxes | count matrix (ints): lives as long as the communicator
    HIPCHK(c, hipMalloc((void**)&c->d_gather, (8 + 8 * 64 + 64 * 64 / 2 + 64) * sizeof(double)));
  }
  return SZ_OK;
}
// ---------------------------------------------------------------- migration
// Floes drift; ownership follows the tile that holds the centroid.  sz_tile_migrate re-assigns every owned floe (collective): the floes
// that changed tile travel with their COMPLETE state -- every column incl. the previous-step tendencies and the stress / strain tensors,
// status, ring, sub-floe points -- over the library's own channel (RCCL send / receive between device buffers, or the host's transport),
// and every rank's context is rebuilt from the floes it keeps and the ones it received, ordered by global index, through the same path an
// upload takes (capacities, neighbour counts, grid, ghost-candidate estimate are all re-derived).  The re-assignment is host-staged inside
// the library -- a rare operation (floes move metres per step against tiles of hundreds of km) whose cost is a download and an upload of
// the tile; the in-reference analogue is the parent / ghost swap of collisions.jl:942-950.  floe.interactions of the last collision call do
// not travel (the next step's collision call rebuilds them before anything reads them).
namespace {
// variable-size all-to-all of doubles between the ranks: sendv[d] to rank d, recvv[s] (sized here) from rank s
int comm_alltoallv(sz_ctx* c, const std::vector<std::vector<double>>& sendv, std::vector<std::vector<double>>& recvv) {
  const int n = c->comm_n, me = c->comm_rank;
  recvv.assign(n, {});
  if (n == 1) return SZ_OK;
  // sizes first: every rank's row of the size matrix
  std::vector<int> mine(n), all((size_t)n * n);
  for (int d = 0; d < n; d++) mine[d] = (int)sendv[d].size();
  int* d_row = (int*)(c->d_gather + 8 + 8 * 64);          // (the count-matrix area of the box gather: free between gathers)
  HIPCHK(c, hipMemcpyAsync(d_row, mine.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  int* d_all = d_row + 64;
  int rc = comm_allgather(c, d_row, d_all, (size_t)n, NCCL_INT32, sizeof(int));
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(all.data(), d_all, all.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int s2 = 0; s2 < n; s2++) if (s2 != me) recvv[s2].assign((size_t)all[(size_t)s2 * n + me], 0.0);
  if (c->host_transport) {
    std::vector<int32_t> peer; std::vector<const void*> sp; std::vector<void*> rp; std::vector<int64_t> sb, rb;
    for (int d = 0; d < n; d++) {
      if (d == me) continue;
      peer.push_back(d); sp.push_back(sendv[d].data()); sb.push_back((int64_t)(sendv[d].size() * sizeof(double)));
      rp.push_back(recvv[d].data()); rb.push_back((int64_t)(recvv[d].size() * sizeof(double)));
    }
    HOSTCHK(c, c->host_tr.sendrecv(c->host_tr.user, (int32_t)peer.size(), peer.data(), sp.data(), sb.data(), rp.data(), rb.data()), "sendrecv");
    return SZ_OK;
  }
  // RCCL: device staging buffers, one grouped send / receive
  size_t ts = 0, tr = 0;
  for (int d = 0; d < n; d++) { if (d == me) continue; ts += sendv[d].size(); tr += recvv[d].size(); }
  PoolGuard pool; double *ds = nullptr, *dr = nullptr;
  if ((rc = dalloc(c, &ds, ts, pool.v)) || (rc = dalloc(c, &dr, tr, pool.v))) return rc;
  size_t os = 0;
  for (int d = 0; d < n; d++) { if (d == me || sendv[d].empty()) continue; HIPCHK(c, hipMemcpyAsync(ds + os, sendv[d].data(), sendv[d].size() * sizeof(double), hipMemcpyHostToDevice, c->stream)); os += sendv[d].size(); }
  NCCLCHK(c, g_rccl.GroupStart());
  os = 0; size_t orr = 0;
  for (int d = 0; d < n; d++) {
    if (d == me) continue;
    if (!sendv[d].empty()) { NCCLCHK(c, g_rccl.Send(ds + os, sendv[d].size(), NCCL_FLOAT64, d, c->comm, c->stream)); os += sendv[d].size(); }
    if (!recvv[d].empty()) { NCCLCHK(c, g_rccl.Recv(dr + orr, recvv[d].size(), NCCL_FLOAT64, d, c->comm, c->stream)); orr += recvv[d].size(); }
  }
  NCCLCHK(c, g_rccl.GroupEnd());
  orr = 0;
  for (int d = 0; d < n; d++) { if (d == me || recvv[d].empty()) continue; HIPCHK(c, hipMemcpyAsync(recvv[d].data(), dr + orr, recvv[d].size() * sizeof(double), hipMemcpyDeviceToHost, c->stream)); orr += recvv[d].size(); }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SZ_OK;
}
// sizes of a variable-size all-to-all: mine[d] doubles go to rank d; all[s * n + d] = what rank s sends to rank d
int comm_sizes(sz_ctx* c, const std::vector<int>& mine, std::vector<int>& all) {
  const int n = c->comm_n;
  all.assign((size_t)n * n, 0);
  if (n == 1) { all[0] = mine[0]; return SZ_OK; }
  int* d_row = (int*)(c->d_gather + 8 + 8 * 64);          // (the count-matrix area of the box gather: free between gathers)
  HIPCHK(c, hipMemcpyAsync(d_row, mine.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  int* d_all = d_row + 64;
  int rc = comm_allgather(c, d_row, d_all, (size_t)n, NCCL_INT32, sizeof(int));
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(all.data(), d_all, all.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SZ_OK;
}

// sz_tile_migrate with the movers packed on the device (sz_migrate.hpp): owners, pack, exchange device to device, merge of the directories, the
// rows gathered into the new order, then what sz_upload_floes does behind its copies (counters, ring signs and boxes, cleared per-floe counts).
// The capacities the context was carved with stay; *fell_back = 1 (and nothing has changed) when some rank's new tile would crowd them --
// every rank then takes the host-staged path below, which carves anew.  The host reads the owner and offset columns (ints) and the merged
// directory; no floe column, ring or sub-floe point crosses to the host (a host transport stages the movers' streams).
int tile_migrate_device(sz_ctx* c, int px, int py, const int32_t* owner_override, int64_t* n_sent, int64_t* n_owned, int* fell_back) {
  State& S = c->S;
  const int n = c->comm_n, me = c->comm_rank, N = c->hostN;
  *fell_back = 0;
  int rc;
  if (c->gi_pending && c->gi_valid) { if ((rc = gi_fetch(c))) return rc; }
  c->gi_pending = false;
  struct Scratch { Pool& v; } pool{ c->mig_allocs };      // (the scratch of the last migration is carved again: no allocation in the common case)
  reset_pool(pool.v);
  // ---- owners, and how much goes where
  int *d_owner = nullptr, *d_override = nullptr, *d_bad = nullptr, *d_cntd = nullptr;
  unsigned long long *d_tally = nullptr, *d_cur = nullptr; long long *d_base = nullptr, *d_rbase = nullptr, *d_rsize = nullptr; double** d_cols = nullptr;
  if ((rc = dalloc(c, &d_owner, (size_t)N + 1, pool.v)) || (rc = dalloc(c, &d_tally, 128, pool.v)) || (rc = dalloc(c, &d_cur, 128, pool.v)) ||
      (rc = dalloc(c, &d_base, 64, pool.v)) || (rc = dalloc(c, &d_rbase, 64, pool.v)) || (rc = dalloc(c, &d_rsize, 64, pool.v)) ||
      (rc = dalloc(c, &d_cntd, 64, pool.v)) || (rc = dalloc(c, &d_bad, 1, pool.v)) || (rc = dalloc(c, &d_cols, 32, pool.v))) return rc;
  if (owner_override) {
    if ((rc = dalloc(c, &d_override, (size_t)N + 1, pool.v))) return rc;
    if (N) HIPCHK(c, hipMemcpyAsync(d_override, owner_override, (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
  }
  const double x0 = c->h_vals[3], y0 = c->h_vals[1], Lx = c->h_vals[2] - c->h_vals[3], Ly = c->h_vals[0] - c->h_vals[1];
  hipLaunchKernelGGL(sz_k_mig_owner, dim3(grid_for(std::max(N, 1), 256)), dim3(256), 0, c->stream, S, N, (const int*)d_override, x0, y0, Lx, Ly, px, py,
                     c->tile_per_x, c->tile_per_y, me, n, d_owner, d_tally, d_bad);
  unsigned long long tally[128]; int bad = 0;
  std::vector<int> owner((size_t)N + 1), voff((size_t)N + 1), soff((size_t)N + 1);
  HIPCHK(c, hipMemcpyAsync(tally, d_tally, sizeof(tally), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  if (N) HIPCHK(c, hipMemcpyAsync(owner.data(), d_owner, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(voff.data(), S.voff, ((size_t)N + 1) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(soff.data(), S.soff, ((size_t)N + 1) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int any_bad = 0;
  if ((rc = comm_agree_bits(c, bad ? 1 : 0, &any_bad))) return rc;
  if (any_bad) { c->err = "sz_tile_migrate: owner out of range"; return SZ_E_ARG; }
  // ---- the movers' records, one stream per destination
  std::vector<int> mine(n, 0), all, cntd(64, 0); std::vector<long long> base(64, 0);
  size_t ts = 0; int nmove = 0; bool too_long = false;
  for (int d = 0; d < n; d++) {
    const unsigned long long cnt = tally[2 * d];
    if (d == me || !cnt) continue;
    const unsigned long long sz = 1 + (unsigned long long)MIG_DIR * cnt + tally[2 * d + 1];
    if (sz > 0x7fffffffull) { too_long = true; break; }
    mine[d] = (int)sz; base[d] = (long long)ts; ts += (size_t)sz; cntd[d] = (int)cnt; nmove += (int)cnt;
  }
  if (too_long) { for (int d = 0; d < n; d++) { mine[d] = 0; cntd[d] = 0; } ts = 0; }      // (says so in the agreement below; nothing is sent)
  double* d_sendb = nullptr;
  if ((rc = dalloc(c, &d_sendb, ts, pool.v))) return rc;
  double* const hcols[MIG_NSC + 3] = { S.cx, S.cy, S.rmax, S.area, S.height, S.mass, S.moment, S.alpha, S.u, S.v, S.xi, S.p_dxdt, S.p_dydt, S.p_dalphadt,
                                       S.p_dudt, S.p_dvdt, S.p_dxidt, S.fxOA, S.fyOA, S.trqOA, S.hflx, S.overarea, S.cfx, S.cfy, S.ctrq, S.sa, S.si, S.strain };
  HIPCHK(c, hipMemcpyAsync(d_cols, hcols, sizeof(hcols), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_base, base.data(), 64 * sizeof(long long), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_cntd, cntd.data(), 64 * sizeof(int), hipMemcpyHostToDevice, c->stream));
  if (ts) hipLaunchKernelGGL(sz_k_mig_pack, dim3(grid_for((long long)N * 64, 256, 2048)), dim3(256), 0, c->stream, S, N, (const int*)d_owner, me, d_sendb,
                             (const long long*)d_base, (const int*)d_cntd, d_cur, (double* const*)d_cols);
  // ---- sizes, then the streams: device to device (a host transport: staged, the movers only)
  if ((rc = comm_sizes(c, mine, all))) return rc;
  std::vector<long long> rbase(64, 0), rsize(64, 0);
  size_t tr = 0;
  for (int s2 = 0; s2 < n; s2++) { if (s2 == me) continue; rbase[s2] = (long long)tr; rsize[s2] = all[(size_t)s2 * n + me]; tr += (size_t)rsize[s2]; }
  double* d_recvb = nullptr;
  if ((rc = dalloc(c, &d_recvb, tr, pool.v))) return rc;
  if (n > 1 && c->host_transport) {
    std::vector<double> hs(std::max<size_t>(ts, 1)), hr(std::max<size_t>(tr, 1));
    if (ts) HIPCHK(c, hipMemcpyAsync(hs.data(), d_sendb, ts * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<int32_t> peer; std::vector<const void*> sp; std::vector<void*> rp; std::vector<int64_t> sb, rb;
    for (int d = 0; d < n; d++) {
      if (d == me) continue;
      peer.push_back(d); sp.push_back(hs.data() + base[d]); sb.push_back((int64_t)mine[d] * (int64_t)sizeof(double));
      rp.push_back(hr.data() + rbase[d]); rb.push_back((int64_t)rsize[d] * (int64_t)sizeof(double));
    }
    HOSTCHK(c, c->host_tr.sendrecv(c->host_tr.user, (int32_t)peer.size(), peer.data(), sp.data(), sb.data(), rp.data(), rb.data()), "sendrecv");
    if (tr) HIPCHK(c, hipMemcpyAsync(d_recvb, hr.data(), tr * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  } else if (n > 1) {
    NCCLCHK(c, g_rccl.GroupStart());
    for (int d = 0; d < n; d++) {
      if (d == me) continue;
      if (mine[d]) NCCLCHK(c, g_rccl.Send(d_sendb + base[d], (size_t)mine[d], NCCL_FLOAT64, d, c->comm, c->stream));
      if (rsize[d]) NCCLCHK(c, g_rccl.Recv(d_recvb + rbase[d], (size_t)rsize[d], NCCL_FLOAT64, d, c->comm, c->stream));
    }
    NCCLCHK(c, g_rccl.GroupEnd());
  }
  // ---- what arrived: the merged directory is all the host reads of it
  const int dcap = (int)(tr / (size_t)(MIG_DIR + MIG_NCOL)) + 1;
  double* d_dirs = nullptr;
  if ((rc = dalloc(c, &d_dirs, (size_t)MIG_DIR * (1 + (size_t)dcap), pool.v))) return rc;
  HIPCHK(c, hipMemcpyAsync(d_rbase, rbase.data(), 64 * sizeof(long long), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_rsize, rsize.data(), 64 * sizeof(long long), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(sz_k_mig_dirs, dim3(1), dim3(256), 0, c->stream, (const double*)d_recvb, (const long long*)d_rbase, (const long long*)d_rsize, n, d_dirs, dcap);
  std::vector<double> dirs((size_t)MIG_DIR * (1 + (size_t)dcap));
  HIPCHK(c, hipMemcpyAsync(dirs.data(), d_dirs, dirs.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int R = (int)dirs[0];
  // ---- the new tile: kept floes + received ones, ordered by global index
  struct Src { long long g; int s; };
  std::vector<Src> src;
  for (int i = 0; i < N; i++) if (owner[i] == me) src.push_back({ c->tile_gidx[i], i });
  for (int e = 0; e < R; e++) src.push_back({ (long long)dirs[(size_t)MIG_DIR * (1 + e)], -(e + 1) });
  std::sort(src.begin(), src.end(), [](const Src& a, const Src& b2) { return a.g < b2.g; });
  const int Nn = (int)src.size();
  std::vector<int> hsrc((size_t)Nn + 1), nvoff((size_t)Nn + 1, 0), nsoff((size_t)Nn + 1, 0); std::vector<long long> ngid((size_t)Nn + 1);
  int ring_in = 0, sub_in = 0; double rmax_in = 0.0;
  for (int r = 0; r < Nn; r++) {
    const int s2 = src[r].s;
    int nv, ns;
    if (s2 >= 0) { nv = voff[s2 + 1] - voff[s2]; ns = soff[s2 + 1] - soff[s2]; }
    else {
      const double* e = dirs.data() + (size_t)MIG_DIR * (size_t)(-s2);
      nv = (int)e[1]; ns = (int)e[2]; ring_in = std::max(ring_in, nv); sub_in = std::max(sub_in, ns); rmax_in = std::max(rmax_in, e[4]);
    }
    hsrc[r] = s2; ngid[r] = src[r].g; nvoff[r + 1] = nvoff[r] + nv; nsoff[r + 1] = nsoff[r] + ns;
  }
  const int Vn = nvoff[Nn], NSn = nsoff[Nn];
  // does it fit what the context was carved for (sz_upload_floes: at least 2 M + 64 rows and 2 V + 4096 ring points, for the owned floes, their
  // ghosts and the halo)?  An eighth more than the upload held is let in.
  const bool fits = Nn <= c->upload_M + c->upload_M / 8 && Vn <= c->upload_V + c->upload_V / 8 && !too_long;
  int bits = (nmove > 0 ? 1 : 0) | (fits ? 0 : 2) | (Nn == 0 ? 4 : 0) | (R < 0 ? 8 : 0), allb = 0;
  if ((rc = comm_agree_bits(c, bits, &allb))) return rc;
  if (allb & 8) { c->err = "sz_tile_migrate: a stream of movers arrived inconsistent"; return SZ_E_STATE; }
  if (n_sent) *n_sent = nmove;
  if (!(allb & 1)) { c->err.clear(); if (n_owned) *n_owned = N; return SZ_OK; }
  if (allb & 4) { c->err = "sz_tile_migrate: a tile without floes (every rank must own at least one)"; return SZ_E_STATE; }
  if (allb & 2) { c->err.clear(); *fell_back = 1; return SZ_OK; }
  c->err.clear();
  // ---- the rows into their new order: gathered beside the old ones first (a row's source may lie on either side of it)
  int *d_src = nullptr, *d_nvoff = nullptr, *d_nsoff = nullptr; double *d_tmp = nullptr, *d_tsx = nullptr, *d_tsy = nullptr; double2* d_tv = nullptr;
  if ((rc = dalloc(c, &d_src, (size_t)Nn + 1, pool.v)) || (rc = dalloc(c, &d_nvoff, (size_t)Nn + 1, pool.v)) || (rc = dalloc(c, &d_nsoff, (size_t)Nn + 1, pool.v)) ||
      (rc = dalloc(c, &d_tmp, (size_t)39 * Nn, pool.v)) || (rc = dalloc(c, &d_tv, (size_t)Vn, pool.v)) ||
      (rc = dalloc(c, &d_tsx, (size_t)NSn, pool.v)) || (rc = dalloc(c, &d_tsy, (size_t)NSn, pool.v))) return rc;
  HIPCHK(c, hipMemcpyAsync(d_src, hsrc.data(), (size_t)Nn * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_nvoff, nvoff.data(), ((size_t)Nn + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_nsoff, nsoff.data(), ((size_t)Nn + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(sz_k_mig_gather, dim3(grid_for(Nn, 256)), dim3(256), 0, c->stream, S, Nn, (const int*)d_src, (const double*)d_dirs, (const double*)d_recvb,
                     (double* const*)d_cols, d_tmp);
  hipLaunchKernelGGL(sz_k_mig_points, dim3(grid_for((long long)Nn * 64, 256, 4096)), dim3(256), 0, c->stream, S, Nn, (const int*)d_src, (const double*)d_dirs,
                     (const double*)d_recvb, (const int*)d_nvoff, (const int*)d_nsoff, d_tv, d_tsx, d_tsy);
  hipLaunchKernelGGL(sz_k_mig_scatter, dim3(grid_for(Nn, 256)), dim3(256), 0, c->stream, S, Nn, (double* const*)d_cols, (const double*)d_tmp);
  if (Vn) HIPCHK(c, hipMemcpyAsync(S.vxy, d_tv, (size_t)Vn * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(S.voff, d_nvoff, ((size_t)Nn + 1) * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
  Pool old_sub;
  if (NSn > S.capS) {          // the sub-floe points have no slack at upload (they are the largest array of a field): a tile that gained points gets a new pair
    old_sub = c->sub_allocs; c->sub_allocs = Pool();
    S.capS = NSn + NSn / 4;
    if ((rc = dalloc(c, &S.sx, (size_t)S.capS, c->sub_allocs)) || (rc = dalloc(c, &S.sy, (size_t)S.capS, c->sub_allocs))) return rc;
  }
  if (NSn) {
    HIPCHK(c, hipMemcpyAsync(S.sx, d_tsx, (size_t)NSn * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(S.sy, d_tsy, (size_t)NSn * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(c, hipMemcpyAsync(S.soff, d_nsoff, ((size_t)Nn + 1) * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
  // ---- as behind the copies of sz_upload_floes: counters, per-floe counts a kernel may read before a collision call writes them, ring signs / boxes / trig
  int h[C_COUNT + 64 + 72] = { 0 };
  h[C_M] = Nn; h[C_N] = Nn; h[C_NV] = Vn; h[C_NGHOSTS] = 0; h[C_NOWN] = Nn;
  HIPCHK(c, hipMemcpyAsync(S.cnt, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(S.over_stamp, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.n_out, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.n_in, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.el_off, 0, ((size_t)S.capM + 2) * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.warn, 0, (size_t)WARN_SLOTS * 32 * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.lb_flag, 0, ((size_t)S.capM / 128 + 8) * sizeof(unsigned), c->stream)); c->scan_epoch = 0;
  HIPCHK(c, hipMemsetAsync(S.inter_cnt, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));          // (no rows until the next collision call)
  hipLaunchKernelGGL(sz_k_osign, dim3(grid_for(S.capM, 256)), dim3(256), 0, c->stream, S, 0);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  free_pool(old_sub);
  const double Lx0 = c->tile_Lx, Ly0 = c->tile_Ly, margin = c->tile_margin; const int perx = c->tile_per_x, pery = c->tile_per_y;
  const int rebox = c->tile_rebox_fixed ? -c->tile_rebox_every : c->tile_rebox_every;
  const double ring_hint = (double)c->max_ring_tiled, rmax_hint = c->rmax_hint;
  const int prec = c->precision;
  S.tiled = 0; S.famrec = 0;
  c->tile_margin = 0.0; c->tile_since_box = -1; c->halo_cap = 0; c->d_send = c->d_recv = c->d_ref = nullptr; c->d_dcap = nullptr;
  c->hostM = Nn; c->hostN = Nn; c->tile_dirty = false; c->mixed_pts_ok = false; c->blk_pts_ok = false; S.sxy = nullptr; c->pts_N = Nn;
  c->gl_valid = false; c->gl_est = std::min(Nn, c->gl_est + R);          // (the ghost-candidate estimate is an upper bound)
  c->mixed_geom_ok = false; c->rings_stale = false; S.rec32 = nullptr; S.ring32 = nullptr; S.body_rings = 0;
  c->max_ring = std::max(c->max_ring, ring_in); c->max_sub = std::max(c->max_sub, sub_in); c->rmax_max = std::max(c->rmax_max, rmax_in);
  setup_grid(c);
  c->fuse_lists.assign(Nn, {});
  c->inter_lost = false; c->inter_any = true;
  if ((rc = sz_tile_enable(c, (const int64_t*)ngid.data(), ring_hint, rmax_hint))) return rc;
  if ((rc = sz_tile_setup(c, Lx0, Ly0, perx, pery, margin, rebox))) return rc;
  if (!owner_override) (void)sz_tile_set_center(c, x0 + ((me % px) + 0.5) * Lx / px, y0 + ((me / px) + 0.5) * Ly / py);
  c->precision = prec;
  trim_pool(pool.v);
  if (n_owned) *n_owned = Nn;
  return SZ_OK;
}
int tile_migrate_host(sz_ctx* c, int32_t px, int32_t py, const int32_t* owner_override, int64_t* n_sent, int64_t* n_owned);
}  // namespace

int sz_tile_migrate(sz_ctx* c, int32_t px, int32_t py, const int32_t* owner_override, int64_t* n_sent, int64_t* n_owned) {
  if (n_sent) *n_sent = 0;
  if (!c || !c->have_floes || !c->S.tiled || c->comm_n < 1 || c->tile_margin <= 0 || px < 1 || py < 1 || (!owner_override && px * py != c->comm_n)) {
    if (c) c->err = "sz_tile_migrate needs a tiled context after sz_tile_setup, and px * py == the number of ranks";
    return SZ_E_STATE;
  }
  (void)hipSetDevice(c->device);
  int rc = tile_sync_agree(c); if (rc) return rc;             // (ghosts and halo floes of the last step are dropped: the state is the owned floes)
  world_rings(c);
  c->migrate_path = 0;
  const char* e = getenv("SZ_MIGRATE_HOST");                  // (A/B switch, the same on every rank: the host-staged path only)
  if (!(e && atoi(e) != 0)) {
    int fell_back = 0;
    rc = tile_migrate_device(c, px, py, owner_override, n_sent, n_owned, &fell_back);
    if (rc) return rc;
    if (!fell_back) { c->migrate_path = 1; return SZ_OK; }
  }
  rc = tile_migrate_host(c, px, py, owner_override, n_sent, n_owned);
  if (rc == SZ_OK) c->migrate_path = 2;
  return rc;
}
// how the last sz_tile_migrate ran: 1 = movers packed on the device, 2 = staged through the host (0: it did not get that far)
int sz_debug_migrate_path(sz_ctx* c) { return c ? c->migrate_path : 0; }
// global indices of the owned floes (sz_tile_enable; after sz_tile_migrate: of the new tile), n_cap >= the number of owned floes
int sz_tile_owned_gidx(sz_ctx* c, int64_t* out, int64_t n_cap) {
  if (!c || !c->have_floes || !c->S.tiled || !out || n_cap < (int64_t)c->tile_gidx.size()) return SZ_E_ARG;
  for (size_t i = 0; i < c->tile_gidx.size(); i++) out[i] = c->tile_gidx[i];
  return SZ_OK;
}

namespace {
int tile_migrate_host(sz_ctx* c, int32_t px, int32_t py, const int32_t* owner_override, int64_t* n_sent, int64_t* n_owned) {
  State& S = c->S;
  const int n = c->comm_n, me = c->comm_rank;
  int rc;
  const int N = c->hostN;
  // ---- the tile's state on the host
  double* const dcol[25] = { S.cx, S.cy, S.rmax, S.area, S.height, S.mass, S.moment, S.alpha, S.u, S.v, S.xi, S.p_dxdt, S.p_dydt, S.p_dalphadt, S.p_dudt, S.p_dvdt, S.p_dxidt,
                             S.fxOA, S.fyOA, S.trqOA, S.hflx, S.overarea, S.cfx, S.cfy, S.ctrq };
  std::vector<std::vector<double>> col(25, std::vector<double>((size_t)N));
  std::vector<double> ten[3] = { std::vector<double>((size_t)4 * N), std::vector<double>((size_t)4 * N), std::vector<double>((size_t)4 * N) };
  std::vector<long long> id((size_t)N); std::vector<int> status((size_t)N), voff((size_t)N + 1), soff((size_t)N + 1);
  for (int k = 0; k < 25; k++) if (N) HIPCHK(c, hipMemcpy(col[k].data(), dcol[k], (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
  double* const dten[3] = { S.sa, S.si, S.strain };
  for (int k = 0; k < 3; k++) if (N) HIPCHK(c, hipMemcpy(ten[k].data(), dten[k], (size_t)4 * N * sizeof(double), hipMemcpyDeviceToHost));
  if (N) { HIPCHK(c, hipMemcpy(id.data(), S.id, (size_t)N * sizeof(long long), hipMemcpyDeviceToHost)); HIPCHK(c, hipMemcpy(status.data(), S.status, (size_t)N * sizeof(int), hipMemcpyDeviceToHost)); }
  HIPCHK(c, hipMemcpy(voff.data(), S.voff, ((size_t)N + 1) * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(soff.data(), S.soff, ((size_t)N + 1) * sizeof(int), hipMemcpyDeviceToHost));
  const int V = voff[N], NS = soff[N];
  std::vector<double> vx((size_t)std::max(V, 1)), vy((size_t)std::max(V, 1)), sx((size_t)std::max(NS, 1)), sy((size_t)std::max(NS, 1));
  if (V) {
    std::vector<double> xy((size_t)2 * V);
    HIPCHK(c, hipMemcpy(xy.data(), S.vxy, (size_t)2 * V * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < V; k++) { vx[k] = xy[(size_t)2 * k]; vy[k] = xy[(size_t)2 * k + 1]; }
  }
  if (NS) { HIPCHK(c, hipMemcpy(sx.data(), S.sx, (size_t)NS * sizeof(double), hipMemcpyDeviceToHost)); HIPCHK(c, hipMemcpy(sy.data(), S.sy, (size_t)NS * sizeof(double), hipMemcpyDeviceToHost)); }
  // ---- who owns what now: the tile that holds the centroid (periodic: of its image inside the domain), px x py tiles over the domain
  const double x0 = c->h_vals[3], y0 = c->h_vals[1], Lx = c->h_vals[2] - c->h_vals[3], Ly = c->h_vals[0] - c->h_vals[1];
  std::vector<int> owner((size_t)N);
  for (int i = 0; i < N; i++) {
    if (owner_override) { owner[i] = owner_override[i]; if (owner[i] < 0 || owner[i] >= n) { c->err = "sz_tile_migrate: owner out of range"; return SZ_E_ARG; } continue; }
    double x = col[0][i] - x0, y = col[1][i] - y0;
    if (c->tile_per_x) { x = std::fmod(x, Lx); if (x < 0) x += Lx; }
    if (c->tile_per_y) { y = std::fmod(y, Ly); if (y < 0) y += Ly; }
    const int ix = std::max(0, std::min(px - 1, (int)(x / Lx * px))), iy = std::max(0, std::min(py - 1, (int)(y / Ly * py)));
    owner[i] = iy * px + ix;
  }
  // ---- movers, one stream of doubles per destination: MIG_NCOL scalars, then ring x / y, then sub-floe points x / y of every floe
  std::vector<std::vector<double>> sendv(n), recvv;
  int nmove = 0;
  for (int i = 0; i < N; i++) {
    if (owner[i] == me) continue;
    nmove++;
    std::vector<double>& b = sendv[owner[i]];
    for (int k = 0; k < 25; k++) b.push_back(col[k][i]);
    for (int k = 0; k < 3; k++) for (int q = 0; q < 4; q++) b.push_back(ten[k][(size_t)4 * i + q]);
    const int nv = voff[i + 1] - voff[i], ns = soff[i + 1] - soff[i];
    b.push_back((double)id[i]); b.push_back((double)status[i]); b.push_back((double)c->tile_gidx[i]); b.push_back((double)nv); b.push_back((double)ns);
    b.insert(b.end(), vx.begin() + voff[i], vx.begin() + voff[i + 1]); b.insert(b.end(), vy.begin() + voff[i], vy.begin() + voff[i + 1]);
    b.insert(b.end(), sx.begin() + soff[i], sx.begin() + soff[i + 1]); b.insert(b.end(), sy.begin() + soff[i], sy.begin() + soff[i + 1]);
  }
  if ((rc = comm_alltoallv(c, sendv, recvv))) return rc;
  // did anything move anywhere?  (every rank must take the same branch: the rebuild below ends in collective set-up calls)
  int moved_all = 0;
  if ((rc = comm_agree_bits(c, nmove > 0 ? 1 : 0, &moved_all))) return rc;
  if (n_sent) *n_sent = nmove;
  if (!moved_all) { if (n_owned) *n_owned = N; return SZ_OK; }
  // ---- the new tile: kept floes + received ones, ordered by global index
  struct Src { long long g; int from; size_t at; };          // from < 0: local row `at`; else stream `from`, offset `at`
  std::vector<Src> src;
  for (int i = 0; i < N; i++) if (owner[i] == me) src.push_back({ c->tile_gidx[i], -1, (size_t)i });
  for (int s2 = 0; s2 < n; s2++) {
    const std::vector<double>& b = recvv[s2];
    for (size_t at = 0; at < b.size();) {
      if (at + MIG_NCOL > b.size()) { c->err = "sz_tile_migrate: truncated record"; return SZ_E_STATE; }
      const int nv = (int)b[at + 40], ns = (int)b[at + 41];
      src.push_back({ (long long)b[at + 39], s2, at });
      at += (size_t)MIG_NCOL + 2 * (size_t)nv + 2 * (size_t)ns;
    }
  }
  std::sort(src.begin(), src.end(), [](const Src& a, const Src& b2) { return a.g < b2.g; });
  const int Nn = (int)src.size();
  if (Nn == 0) { c->err = "sz_tile_migrate: a tile without floes (every rank must own at least one)"; return SZ_E_STATE; }
  std::vector<std::vector<double>> ncol(25, std::vector<double>((size_t)Nn));
  std::vector<double> nten[3] = { std::vector<double>((size_t)4 * Nn), std::vector<double>((size_t)4 * Nn), std::vector<double>((size_t)4 * Nn) };
  std::vector<long long> nid((size_t)Nn), ngid((size_t)Nn); std::vector<int> nstatus((size_t)Nn), nvoff((size_t)Nn + 1, 0), nsoff((size_t)Nn + 1, 0);
  std::vector<double> nvx, nvy, nsx, nsy;
  for (int r = 0; r < Nn; r++) {
    const Src& q = src[r];
    ngid[r] = q.g;
    if (q.from < 0) {
      const size_t i = q.at;
      for (int k = 0; k < 25; k++) ncol[k][r] = col[k][i];
      for (int k = 0; k < 3; k++) for (int t = 0; t < 4; t++) nten[k][(size_t)4 * r + t] = ten[k][4 * i + t];
      nid[r] = id[i]; nstatus[r] = status[i];
      nvx.insert(nvx.end(), vx.begin() + voff[i], vx.begin() + voff[i + 1]); nvy.insert(nvy.end(), vy.begin() + voff[i], vy.begin() + voff[i + 1]);
      nsx.insert(nsx.end(), sx.begin() + soff[i], sx.begin() + soff[i + 1]); nsy.insert(nsy.end(), sy.begin() + soff[i], sy.begin() + soff[i + 1]);
    } else {
      const double* b = recvv[q.from].data() + q.at;
      for (int k = 0; k < 25; k++) ncol[k][r] = b[k];
      for (int k = 0; k < 3; k++) for (int t = 0; t < 4; t++) nten[k][(size_t)4 * r + t] = b[25 + 4 * k + t];
      nid[r] = (long long)b[37]; nstatus[r] = (int)b[38];
      const int nv = (int)b[40], ns = (int)b[41];
      const double* p = b + MIG_NCOL;
      nvx.insert(nvx.end(), p, p + nv); nvy.insert(nvy.end(), p + nv, p + 2 * nv);
      p += 2 * (size_t)nv;
      nsx.insert(nsx.end(), p, p + ns); nsy.insert(nsy.end(), p + ns, p + 2 * ns);
    }
    nvoff[r + 1] = (int)nvx.size(); nsoff[r + 1] = (int)nsx.size();
  }
  if (nvx.empty()) { nvx.push_back(0.0); nvy.push_back(0.0); }
  if (nsx.empty()) { nsx.push_back(0.0); nsy.push_back(0.0); }
  // ---- rebuild through the upload path, then the tile set-up again (collective, same parameters as before)
  const double Lx0 = c->tile_Lx, Ly0 = c->tile_Ly, margin = c->tile_margin; const int perx = c->tile_per_x, pery = c->tile_per_y;
  const int rebox = c->tile_rebox_fixed ? -c->tile_rebox_every : c->tile_rebox_every;
  const double ring_hint = (double)c->max_ring_tiled, rmax_hint = c->rmax_hint;
  sz_floe_columns f; memset(&f, 0, sizeof(f));
  f.cx = ncol[0].data(); f.cy = ncol[1].data(); f.rmax = ncol[2].data(); f.area = ncol[3].data(); f.height = ncol[4].data(); f.mass = ncol[5].data(); f.moment = ncol[6].data();
  f.alpha = ncol[7].data(); f.u = ncol[8].data(); f.v = ncol[9].data(); f.xi = ncol[10].data(); f.p_dxdt = ncol[11].data(); f.p_dydt = ncol[12].data(); f.p_dalphadt = ncol[13].data();
  f.p_dudt = ncol[14].data(); f.p_dvdt = ncol[15].data(); f.p_dxidt = ncol[16].data(); f.fxOA = ncol[17].data(); f.fyOA = ncol[18].data(); f.trqOA = ncol[19].data();
  f.hflx_factor = ncol[20].data(); f.overarea = ncol[21].data(); f.coll_fx = ncol[22].data(); f.coll_fy = ncol[23].data(); f.coll_trq = ncol[24].data();
  f.stress_accum = nten[0].data(); f.stress_instant = nten[1].data(); f.strain = nten[2].data();
  f.id = (int64_t*)nid.data(); f.status = nstatus.data(); f.vert_off = nvoff.data(); f.vx = nvx.data(); f.vy = nvy.data(); f.sub_off = nsoff.data(); f.sx = nsx.data(); f.sy = nsy.data();
  const int prec = c->precision;
  if ((rc = sz_upload_floes(c, Nn, Nn, &f))) return rc;
  c->inter_lost = false; c->inter_any = true;
  HIPCHK(c, hipMemsetAsync(S.inter_cnt, 0, ((size_t)S.capM + 1) * sizeof(int), c->stream));          // (no rows until the next collision call)
  if ((rc = sz_tile_enable(c, (const int64_t*)ngid.data(), ring_hint, rmax_hint))) return rc;
  if ((rc = sz_tile_setup(c, Lx0, Ly0, perx, pery, margin, rebox))) return rc;
  if (!owner_override) (void)sz_tile_set_center(c, x0 + ((me % px) + 0.5) * Lx / px, y0 + ((me / px) + 0.5) * Ly / py);
  c->precision = prec;
  if (n_owned) *n_owned = Nn;
  return SZ_OK;
}
}  // namespace
// sub-floe points of the floes the context holds (CSR: off has N + 1 entries; call with sx == NULL for the offsets alone): after a
// migration the host's copy of these is the library's
int sz_download_subpoints(sz_ctx* c, int32_t* off, double* sx, double* sy) {
  if (!c || !c->have_floes || !off) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  tile_cleanup(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int N = c->hostN;
  HIPCHK(c, hipMemcpy(off, c->S.soff, ((size_t)N + 1) * sizeof(int), hipMemcpyDeviceToHost));
  if (sx && sy && off[N] > 0) {
    HIPCHK(c, hipMemcpy(sx, c->S.sx, (size_t)off[N] * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(sy, c->S.sy, (size_t)off[N] * sizeof(double), hipMemcpyDeviceToHost));
  }
  return SZ_OK;
}

// The exchange of one step: the regions of d_send to the peers, theirs into d_recv, on the communication stream behind the pack kernel
// (ev_packed) -- ev_recv is recorded when the halo is in.  all_ranks: every rank gets at least the header record (the stop agreement of the
// inline steps reads the flags of ALL ranks); otherwise only the neighbouring tiles take part.
namespace {
int tile_exchange(sz_ctx* c, bool all_ranks) {
  const int n = c->comm_n, me = c->comm_rank;
  if (n <= 1) return SZ_OK;
  const int HREC = halo_rec(c->S);
  const size_t stride = (size_t)(c->halo_cap + 1) * HREC;
  HIPCHK(c, hipEventRecord(c->ev_packed, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_packed, 0));
  if (c->host_transport) {
    Timed tx(c, K_EXCHANGE, c->comm_stream);
    c->h_send.resize((size_t)n * stride); c->h_recv.resize((size_t)n * stride);
    std::vector<int32_t> peer; std::vector<const void*> sp; std::vector<void*> rp; std::vector<int64_t> sb, rb;
    for (int d = 0; d < n; d++) {
      if (d == me || (!all_ranks && c->cap_send[d] <= 0 && c->cap_recv[d] <= 0)) continue;
      const size_t ns = (all_ranks || c->cap_send[d] > 0) ? (size_t)(c->cap_send[d] + 1) * HREC : 0;
      const size_t nr = (all_ranks || c->cap_recv[d] > 0) ? (size_t)(c->cap_recv[d] + 1) * HREC : 0;
      if (ns) HIPCHK(c, hipMemcpyAsync(c->h_send.data() + d * stride, c->d_send + d * stride, ns * sizeof(double), hipMemcpyDeviceToHost, c->comm_stream));
      peer.push_back(d); sp.push_back(c->h_send.data() + d * stride); sb.push_back((int64_t)(ns * sizeof(double)));
      rp.push_back(c->h_recv.data() + d * stride); rb.push_back((int64_t)(nr * sizeof(double)));
    }
    HIPCHK(c, hipStreamSynchronize(c->comm_stream));
    HOSTCHK(c, c->host_tr.sendrecv(c->host_tr.user, (int32_t)peer.size(), peer.data(), sp.data(), sb.data(), rp.data(), rb.data()), "sendrecv");
    for (size_t k = 0; k < peer.size(); k++)
      if (rb[k]) HIPCHK(c, hipMemcpyAsync(c->d_recv + peer[k] * stride, rp[k], (size_t)rb[k], hipMemcpyHostToDevice, c->comm_stream));
    HIPCHK(c, hipStreamSynchronize(c->comm_stream));       // (h_recv is reused by the next step)
    tx.end();
  } else {
    Timed tx(c, K_EXCHANGE, c->comm_stream);          // (class "exchange" of sz_kernel_time_ms: the grouped send / receive on the communication stream)
    NCCLCHK(c, g_rccl.GroupStart());
    for (int d = 0; d < n; d++) {
      if (d == me) continue;
      if (all_ranks || c->cap_send[d] > 0) NCCLCHK(c, g_rccl.Send(c->d_send + d * stride, (size_t)(c->cap_send[d] + 1) * HREC, NCCL_FLOAT64, d, c->comm, c->comm_stream));
      if (all_ranks || c->cap_recv[d] > 0) NCCLCHK(c, g_rccl.Recv(c->d_recv + d * stride, (size_t)(c->cap_recv[d] + 1) * HREC, NCCL_FLOAT64, d, c->comm, c->comm_stream));
    }
    NCCLCHK(c, g_rccl.GroupEnd());
    tx.end();
  }
  HIPCHK(c, hipEventRecord(c->ev_recv, c->comm_stream));
  return SZ_OK;
}
// what a pack is given (sz_k_halo_pack; the integrator of a tiled step that packs for the next one: sz_k_integrate<true, true>)
PackInl tile_pack_args(sz_ctx* c) {
  State& S = c->S;
  PackInl a;
  a.send = c->d_send; a.boxes = S.bounds + 16; a.dcap = c->d_dcap; a.ref = c->d_ref; a.counts = S.cnt + C_COUNT;
  a.Lx = c->tile_Lx; a.Ly = c->tile_Ly; a.margin = c->tile_margin;
  a.nranks = c->comm_n; a.me = c->comm_rank; a.cap = c->halo_cap; a.per_x = c->tile_per_x; a.per_y = c->tile_per_y;
  return a;
}
void tile_pack(sz_ctx* c) {
  State& S = c->S;
  hipLaunchKernelGGL(sz_k_halo_pack, dim3(grid_for(std::max(c->hostN, 1), PACK_TPB)), dim3(PACK_TPB), 0, c->stream, S, c->comm_n, c->comm_rank, S.bounds + 16,
                     c->tile_Lx, c->tile_Ly, c->tile_per_x, c->tile_per_y, c->d_send, c->halo_cap, S.cnt + C_COUNT, (const int*)c->d_dcap, (const double*)c->d_ref, c->tile_margin);
}
// status.fuse_idx of a tiled context, in GLOBAL floe numbers: the order keys of the local rows of the step that ended the batch (owned
// floes: their global index; halo floes and ghosts: the step's key table), the replay of host_fuse_fixup in key order, and the partners
// renamed -- a ghost by its parent
int tile_fuse_replay(sz_ctx* c, const int* h, bool last_coupled) {
  int rc = gi_fetch(c); if (rc) return rc;
  const int M = h[C_N] + h[C_NGHOSTS];
  std::vector<long long> keys(M, 0);
  for (int i = 0; i < M; i++) keys[i] = i < c->hostN ? c->tile_gidx[i] : (i - c->hostN < (int)c->gi_keys.size() ? c->gi_keys[i - c->hostN] : ((long long)3 << 40) + i);
  rc = host_fuse_fixup(c, h, true, true, last_coupled, &keys);
  if (rc) return rc;
  const long long lim = (long long)1 << 40;
  for (int i = 0; i < c->hostN && i < (int)c->fuse_lists.size(); i++)
    for (int& v : c->fuse_lists[i]) { if (v < 0 || v >= (int)keys.size()) continue; long long key = keys[v]; if (key >= lim) key = (key & (lim - 1)) >> 2; v = (int)key; }
  return SZ_OK;
}
}  // namespace

// the centre of this rank's tile (optional, after sz_tile_setup): in a periodic direction the owned box of the FIRST gather then takes every
// centroid at its image nearest to it, as the later gathers do with the centre of the box before (sz_k_owned_box)
int sz_tile_set_center(sz_ctx* c, double x, double y) {
  if (!c || !c->S.tiled || c->tile_margin <= 0) { if (c) c->err = "sz_tile_set_center needs sz_tile_setup"; return SZ_E_STATE; }
  c->tile_box_ctr[0] = x; c->tile_box_ctr[1] = y; c->tile_box_valid = true;
  return SZ_OK;
}
// nsteps x timestep_sim! of a tiled run, collectively on every rank (same arguments everywhere)
int sz_tile_run(sz_ctx* c, int32_t nsteps, int32_t tstep0, int32_t dt, int32_t coupling_dt, int32_t flags, int32_t* steps_done) {
  if (steps_done) *steps_done = 0;
  if (!c || !c->have_floes || !c->S.tiled || c->comm_n < 1 || c->tile_margin <= 0) {
    if (c) c->err = "sz_tile_run needs sz_tile_enable and sz_tile_setup after the last sz_upload_floes";
    return SZ_E_STATE;
  }
  if (nsteps < 0) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  State& S = c->S;
  const int n = c->comm_n, me = c->comm_rank;
  const bool coll = (flags & SZ_COLLISIONS_ON) != 0;
  // The steps of a tile are the single context's (sz_step): ghosts made by whoever places the parent (integrator: owned floes, unpack:
  // halo floes), forcings in the tail of the narrow launch, no ghost launch -- plus the pack and unpack kernels and the exchange.
  // Needs what the inline ghost maker needs (rings that fit the one-launch integrator, the static grid).  Otherwise (and with
  // SZ_TILE_INLINE=0): the list-based steps of sz_tile_step.
  S.crec = nullptr;
  const bool inl = coll && !c->tile_inline_off && c->ghost_inline && c->fused_move && c->grid_ok && !c->no_static_grid && !c->two_way &&
                   std::max(c->max_ring, c->max_ring_tiled) <= MV_RING && ((flags & SZ_COUPLING_ON) == 0 || c->have_fields);
  if (!inl) {
    // The tag stop of these steps (one-way coupling): the pack kernel's header records carry every rank's stop word to EVERY rank, the unpack
    // kernel of the next step reads them before that step has touched anything and ends the batch there (sz_k_halo_unpack), as in the inline
    // steps.  The forcings then run behind the unpack instead of beside the exchange: a rank must not compute the forcings of a step its
    // peers have already called off.  Two-way coupling across tiles (round 4): the same stop -- the steps behind it are enqueued and return at
    // once; their all-reduces of the per-cell sums still run on every rank (collectives must), on the sums of the step that ended the
    // batch, and sz_two_way_finish writes the ocean fields of that step once more: the same values.
    const bool stopping = !(flags & SZ_NO_STOP);
    S.stop_on_tags = stopping ? 1 : 0; S.retry_stop = 0;
    HIPCHK(c, hipMemsetAsync(S.cnt + C_STOP, 0, sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(S.cnt + C_RETRYSTOP, 0, sizeof(int), c->stream));
    // (every way out of the loop leaves the context as a batch that has ended: no step number in the State, or the next process-mode call's
    //  kernels would take themselves for launches behind a stop)
    auto out = [&](int rc) { S.step = 0; S.stop_on_tags = 0; return rc; };
    for (int s = 0; s < nsteps; s++) {
      const int tstep = tstep0 + s;
      c->tile_dt = dt;
      S.step = stopping ? s + 1 : 0;
      if (c->tile_since_box < 0 || c->tile_since_box >= c->tile_rebox_cur) { int rc = tile_rebox(c); if (rc) return out(rc); }
      c->tile_since_box++;
      const bool coupling = (flags & SZ_COUPLING_ON) && coupling_dt > 0 && (tstep % coupling_dt) == 0;
      tile_pack(c);
      // (the host's channel blocks: the forcings go to the device first and run while the host trades the regions)
      if (coupling && !stopping && n > 1 && c->host_transport) { int rc = sz_tile_forcing(c, tstep, coupling_dt, flags); if (rc) return out(rc); }
      { int rc = tile_exchange(c, stopping); if (rc) return out(rc); }
      // the forcings of the owned floes need nothing from the halo: they run beside the exchange
      if (coupling && !stopping && !(n > 1 && c->host_transport)) { int rc = sz_tile_forcing(c, tstep, coupling_dt, flags); if (rc) return out(rc); }
      if (n > 1 && hipStreamWaitEvent(c->stream, c->ev_recv, 0) != hipSuccess) { c->err = "hipStreamWaitEvent (halo exchange)"; return out(SZ_E_HIP); }
      int rc = sz_tile_step(c, c->d_recv, n, c->halo_cap, tstep, dt, coupling_dt, flags);
      if (rc) return out(rc);
      if (c->two_way && coupling) {       // ice-on-ocean stress: per-cell partial sums, summed over the ranks, finished on every rank
        const size_t nc = 3 * c->tw_ncell;
        if (!c->d_tw_partial) { int r2 = dalloc(c, &c->d_tw_partial, nc, c->tw_part_allocs); if (r2) return r2; }
        if ((rc = sz_two_way_partial(c, c->d_tw_partial)) || (rc = sz_comm_allreduce(c, c->d_tw_partial, (int64_t)nc)) ||
            (rc = sz_two_way_finish(c, c->d_tw_partial, dt))) return out(rc);
      }
    }
    // (the ranks agree on the error word: a rank with a device error and a clean one return the same code)
    S.step = 0;
    int hl[C_COUNT] = { 0 };
    const int rce = tile_sync_agree(c, hl);
    c->fuse_lists.resize(c->hostM);
    const int done = stopping && hl[C_STOP] > 0 ? std::min(hl[C_STOP], (int)nsteps) : nsteps;
    if (done < nsteps) { c->grid_live = false; c->gl_valid = false; }          // stopped early: cells and ghost-candidate lists belong to steps that did not come
    if (steps_done) *steps_done = done;
    return rce;
  }
  // ---------------- inline steps
  const bool periodic = S.any_periodic_ew || S.any_periodic_ns;
  tile_cleanup(c);                                  // (the halo floes / ghosts a list-based call may have left attached)
  if (c->gi_pending) c->gi_pending = false;
  c->gi_valid = false;
  c->gl_valid = false;
  world_rings(c);
  S.stop_on_tags = (flags & SZ_NO_STOP) ? 0 : 1;
  HIPCHK(c, hipMemsetAsync(S.cnt + C_STOP, 0, sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.cnt + C_RETRYSTOP, 0, sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.cnt + C_PAUSED, 0, sizeof(int), c->stream));
  use_static_grid(c);
  if (c->precision == 1) { int rc = ensure_mixed(c); if (rc) return rc; }
  else if (!c->two_way && c->have_fields) { int rc = ensure_block_points(c); if (rc) return rc; }
  S.ginline = 1; S.famrec = 1; S.retry_stop = 0; S.body_rings = 0;
  HIPCHK(c, hipMemsetAsync(S.galloc, 0, 32 * sizeof(unsigned long long), c->stream));
  // the periodic ghosts of the owned floes for the first step (and the swap of parents that lie outside the domain), BEFORE the first pack
  // collision records of the owned floes (the halo floes get theirs from the unpack kernel, ghosts from their maker; see sz_step)
  S.crec = (!c->no_crec && c->crec_buf && nsteps > 0) ? c->crec_buf : nullptr; c->crec_was_live = S.crec != nullptr;
  if (S.crec) hipLaunchKernelGGL(sz_k_crec_seed, dim3(grid_for(c->hostN, 256)), dim3(256), 0, c->stream, S, c->hostN);
  // (the periodic ghosts of the owned floes for the first step, and the swap of parents that lie outside the domain: in the loop, BEHIND the
  //  first pack -- see there)
  // fixed-point totals and reduce-free steps as in sz_step.  With peers and the tag stop a rank learns that a step was the batch's last only
  // in the unpack of the NEXT one -- after its integrator has made that step's ghosts over the rows of this one -- so the rows are then
  // assembled inside every step (rows only); alone, or in batches that run through (SZ_NO_STOP), once behind the batch.
  const bool facc_on = c->facc_buf != nullptr;
  const int rmode = !facc_on ? 0 : (c->no_reduce_free || (n > 1 && S.stop_on_tags)) ? 1 : 2;
  S.facc = facc_on ? c->facc_buf : nullptr; S.kexp = force_scale_exp(c); c->reduce_mode = rmode;
  if (facc_on) {
    HIPCHK(c, hipMemsetAsync(c->facc_buf, 0, (size_t)FX_WORDS * S.capM * sizeof(long long), c->stream));
    HIPCHK(c, hipMemsetAsync(S.cnt + C_FRCSTOP, 0, sizeof(int), c->stream));
  }
  auto accm = [&](bool last) { return !facc_on ? 0 : 1 | (rmode == 2 ? 4 | (last ? 2 : 0) : 0); };
  int cur_set = 0;          // (the forcing output set in use: see `beside` below)
  auto swap_frc = [&]() { std::swap(S.fxOA, c->frc_alt[0]); std::swap(S.fyOA, c->frc_alt[1]); std::swap(S.trqOA, c->frc_alt[2]); std::swap(S.hflx, c->frc_alt[3]); cur_set ^= 1; };
  auto fail = [&](int rc) { if (cur_set) swap_frc(); S.ginline = 0; S.famrec = 0; S.step = 0; S.crec = nullptr; S.retry_stop = 0; S.facc = nullptr; c->acc_mode = 0; c->reduce_mode = 0; return rc; };
  // SZ_SYNC_DEBUG=1 (diagnosis of a faulting kernel): wait after every stage of every step and say so on stderr -- the last line names the stage
  const bool dbgsync = getenv("SZ_SYNC_DEBUG") != nullptr;
  auto stage_done = [&](int s, const char* what) {
    if (!dbgsync) return;
    const hipError_t e = hipStreamSynchronize(c->stream);
    int h4[C_COUNT]; (void)hipMemcpy(h4, S.cnt, sizeof(h4), hipMemcpyDeviceToHost);
    fprintf(stderr, "[sz rank %d] step %d: %s done (%s) M=%d N=%d own=%d halo=%d ghosts=%d err=0x%x\n", me, s, what, hipGetErrorString(e), h4[C_M], h4[C_N], h4[C_NOWN], h4[C_NHALO], h4[C_NGHOSTS], h4[C_ERR]);
    fflush(stderr);
  };
  stage_done(-1, "seed");
  std::vector<signed char> fset((size_t)std::max(nsteps, 1), (signed char)-1);      // the output set the forcings of step s wrote (-1: none of this kind)
  // The largest narrow variant is left out of the steps until an item needs it, as in sz_step (-4 us and a launch boundary per step).  A
  // rank whose narrow phase meets such an item pauses inside that step (C_RETRYSTOP); its pause rides in the header records of the next
  // exchange (sz_k_halo_pack hdr[2]), whose unpack kernel stops every other rank before that step has touched anything.  After the sync
  // all ranks know the step: the rank that paused finishes it (the variant, the reduce, the integrator), and everybody runs the rest of
  // the batch again from the step after it, as a batch that starts there (cells, ghosts, records seeded anew) with the variant in.
  const bool hdr_all = !(c->tile_hdr_neighbours && (flags & SZ_NO_STOP));      // (the A/B arm without the all-pairs headers: see tile_hdr_neighbours)
  bool lean = !c->retry_seen && !c->no_lean_narrow && !larger_rings(c) && !dbgsync && hdr_all;
  std::vector<int> callid_of((size_t)std::max(nsteps, 1), 0);
  int h[C_COUNT]; int rc = SZ_OK;
  // The halo records of step s + 1 are written by the integrator of step s: the thread that has just placed the floe holds all a record
  // carries, and it packs the floe as the update left it, BEFORE the swap of a parent that left the domain -- the receiving rank makes
  // the ghosts with the routine the owner runs on the same values (see sz_k_integrate<true, true>).  A pack launch remains for the first
  // step of a batch (and the step a paused batch is taken up again at): it runs BEFORE the launch that makes the owned floes' first
  // ghosts and swaps the parents that lie outside, for the same reason.  New boxes for step s + 1 are therefore gathered before the
  // integrator of step s (from the positions that step started with; the drift margin covers the step in between).
  // fresh: this rank starts the (sub-)batch from its parents as they lie -- boxes, the first pack, the first ghosts.  After a pause only the
  // rank that paused does: it finished its step without making anything for the next one.  The others have that step behind them as any other
  // -- their integrator has made the next step's ghosts (from the parents BEFORE their swap, like the single context and the reference:
  // collisions.jl:942-950) and packed the halo records -- and simply take up the steps where the pause stopped them.
  bool fresh = true;
  for (int s_begin = 0;;) {
  S.retry_stop = lean ? 1 : 0;
  for (int s = s_begin; s < nsteps; s++) {
    const int tstep = tstep0 + s;
    S.step = s + 1; S.gslot = s & 1;
    c->tile_dt = dt;
    if (s == s_begin && s_begin == 0) {
      if (c->tile_since_box < 0 || c->tile_since_box >= c->tile_rebox_cur) { int rc = tile_rebox(c); if (rc) return fail(rc); }
      stage_done(s, "rebox");
    }
    if (s == s_begin && fresh) {
      tile_pack(c);
      if (periodic) hipLaunchKernelGGL(sz_k_ghost_inline_seed, dim3(grid_for(S.capM, 256)), dim3(256), 0, c->stream, S, s & 1, c->hostN);
    }
    c->tile_since_box++;
    const bool coupling = (flags & SZ_COUPLING_ON) && coupling_dt > 0 && (tstep % coupling_dt) == 0;
    stage_done(s, "pack");
    // With peers the forcings of the owned floes (they need nothing from the halo) run BESIDE the exchange -- on the main stream while the
    // communication stream trades the regions, before the host's channel blocks -- and the narrow launch carries no forcing tail; without
    // peers there is nothing to hide them behind and they ride in the narrow launch's tail as in sz_step.
    const bool beside = coupling && n > 1 && !(c->pmask >> SZ_K_FORCING & 1u) && !c->tile_forcing_in_tail;
    // (these forcings run before this rank knows whether a peer has asked for the batch to end at the previous step -- the unpack kernel
    //  below finds out.  They therefore write a SECOND set of the four output columns, alternating step by step, and the set the last
    //  step that really ran has written is made the context's at the end of the call: a batch that ends early leaves fxOA .. hflx of the
    //  step it ended with, as sz_step does.)
    if (beside) { swap_frc(); fset[s] = (signed char)cur_set; }
    if (beside && c->host_transport) stage_forcing(c, dt);
    { int rc = tile_exchange(c, hdr_all); if (rc) return fail(rc); }
    if (beside && !c->host_transport) stage_forcing(c, dt);
    if (n > 1) {
      if (hipStreamWaitEvent(c->stream, c->ev_recv, 0) != hipSuccess) { c->err = "hipStreamWaitEvent (halo exchange)"; return fail(SZ_E_HIP); }
      const long long slots = (long long)n * c->halo_cap;
      hipLaunchKernelGGL(sz_k_halo_unpack_inline, dim3(grid_for(slots, UNPACK_TPB, 1 << 20)), dim3(UNPACK_TPB), 0, c->stream, S, (const double*)c->d_recv, n, me, c->halo_cap,
                         S.gslot, c->hostN);
    }
    stage_done(s, "exchange + unpack");
    // the forcings: where sz_step puts them (the tail of the narrow launch for tiles of up to 30 k owned floes, the neighbour launch up to 65 k)
    const bool fuse = coupling && !beside && !(c->pmask >> SZ_K_FORCING & 1u) && c->fuse_forcing && c->hostN <= 65536;
    int fmode = !fuse ? 0 : c->fuse_forcing_mode ? c->fuse_forcing_mode : (c->hostN <= 30000 ? 2 : 1);
    if (fmode == 1 && S.maxnb > MAXNB) fmode = 2;
    if (coupling && !fuse && !beside) stage_forcing(c, dt);
    if (coupling) c->forcing_where = fmode;
    S.callid = ++c->callid; callid_of[s] = S.callid;
    if (facc_on && !(S.crec && S.maxnb <= MAXNB)) (void)hipMemsetAsync(c->facc_buf + (size_t)FX_WORDS * c->hostN, 0, (size_t)FX_WORDS * (S.capM - c->hostN) * sizeof(long long), c->stream);      // (see sz_step)
    if (dbgsync) {          // (the stages of collisions_step one by one)
      stage_broad(c, false, true, fmode == 1, false); stage_done(s, "neighbour search");
      stage_elems(c, true); stage_done(s, "element items");
      stage_narrow(c, dt, c->P.ff_max_overlap, c->P.fd_max_overlap, true, fmode == 2 ? (c->precision == 1 ? 2 : 1) : 0, 0); stage_done(s, "narrow phase");
      stage_reduce(c, 1, -1, dt, 0); stage_done(s, "reduce");
    } else collisions_step(c, -1, dt, false, true, fmode, lean, false);
    const bool pack_next = s + 1 < nsteps;
    if (pack_next && c->tile_since_box >= c->tile_rebox_cur) { int rc = tile_rebox(c); if (rc) return fail(rc); }      // (synchronises: once per gather interval)
    const PackInl pk = tile_pack_args(c);
    c->acc_mode = accm(s + 1 == nsteps);
    stage_integrate(c, dt, false, coupling, true, -1, periodic && s + 1 < nsteps ? 1 - (s & 1) : -1, pack_next ? &pk : nullptr);
    stage_done(s, "integrate");
  }
  S.step = 0;
  if (rmode == 2 && nsteps > 0) stage_reduce(c, 1, -1, dt, 0, true);          // floe.interactions of the step that ended the batch
  c->tile_dirty = nsteps > 0;
  rc = sync_and_check(c, h);                     // (drops the halo floes and ghosts of the last step: tile_cleanup -- unless a step is paused)
  if (rc == SZ_E_HIP) return fail(rc);
  {          // (the ranks agree on the error word BEFORE anybody decides to run steps again: a rank leaving on its own would hang the others)
    int all = 0;
    const int rc2 = comm_agree_bits(c, rc ? (c->last_err_bits ? c->last_err_bits : 1) : 0, &all);
    if (rc2) return fail(rc2);
    if (all) return fail(SZ_E_CAPACITY);
  }
  // the step that was paused (here or on a peer) and the step a tag ended the batch at, as every rank sees them (comm_agree_steps)
  int sp = h[C_RETRYSTOP], st_all = h[C_STOP];
  { const int rc3 = comm_agree_steps(c, h[C_STOP], h[C_RETRYSTOP], &st_all, &sp); if (rc3) return fail(rc3); }
  if (st_all > 0) h[C_STOP] = st_all;
  if (!lean || sp <= 0 || (st_all > 0 && st_all < sp)) break;
  // ---- a pause for the largest narrow variant in step sp
  const int tsp = tstep0 + sp - 1;
  const bool coupling_sp = (flags & SZ_COUPLING_ON) && coupling_dt > 0 && (tsp % coupling_dt) == 0;
  HIPCHK(c, hipMemsetAsync(S.cnt + C_RETRYSTOP, 0, sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(S.cnt + C_PAUSED, 0, sizeof(int), c->stream));
  S.retry_stop = 0; c->retry_seen = true; lean = false;
  {          // the forcing outputs as of step sp (the steps after it are run again).  BEFORE that step is finished: its integrator reads them, and
             // the steps enqueued behind it have gone on alternating the sets on the host while their forcing kernels returned at once
    int want = 0;
    for (int s2 = 0; s2 < sp; s2++) if (fset[s2] >= 0) want = fset[s2];
    if (want != cur_set) swap_frc();
    for (int s2 = sp; s2 < nsteps; s2++) fset[s2] = -1;
  }
  if (h[C_PAUSED] == sp) {          // this rank's step: the variant, then what the pause held back
    S.step = sp; S.gslot = (sp - 1) & 1; S.callid = callid_of[sp - 1];
    stage_narrow(c, dt, c->P.ff_max_overlap, c->P.fd_max_overlap, true, 0, 2);
    stage_reduce(c, 1, -1, dt, 0);
    c->acc_mode = accm(sp >= nsteps);
    stage_integrate(c, dt, false, coupling_sp, true, -1, -1);
    S.step = 0;
    if (rmode == 2 && sp >= nsteps) stage_reduce(c, 1, -1, dt, 0, true);
  }
  c->tile_dirty = true;
  if (sp >= nsteps || (st_all > 0 && st_all <= sp)) {               // (the last step of the batch -- or a peer tagged a floe in this very step: the batch ends with it -- nothing is run again)
    rc = sync_and_check(c, h);
    if (rc == SZ_E_HIP) return fail(rc);
    if (st_all > 0) h[C_STOP] = h[C_STOP] > 0 ? std::min(h[C_STOP], st_all) : st_all;
    break;
  }
  // the rest of the batch again.  The rank that paused: from its floes as they lie after step sp (sz_step's capacity restart does the same);
  // the others: on from where the pause stopped them (see `fresh`)
  fresh = h[C_PAUSED] == sp;
  if (fresh) {
    tile_cleanup(c);
    c->grid_live = false; use_static_grid(c);
    HIPCHK(c, hipMemsetAsync(S.galloc, 0, 32 * sizeof(unsigned long long), c->stream));
    if (S.crec) hipLaunchKernelGGL(sz_k_crec_seed, dim3(grid_for(c->hostN, 256)), dim3(256), 0, c->stream, S, c->hostN);
  } else {
    c->tile_dirty = false;          // (nothing to drop: the rows behind the owned floes are the NEXT step's ghosts)
    // the header records this rank's last pack left say "paused" (the launches enqueued behind the pause wrote the word there): not any more
    const size_t hstride = (size_t)(c->halo_cap + 1) * halo_rec(S);
    for (int d = 0; d < n; d++) HIPCHK(c, hipMemsetAsync(c->d_send + (size_t)d * hstride + 2, 0, sizeof(double), c->stream));
  }
  s_begin = sp;          // (the ghosts of that step: behind its pack, at the top of the loop)
  }
  S.step = 0; S.ginline = 0; S.famrec = 0; S.crec = nullptr; S.retry_stop = 0; S.facc = nullptr; c->acc_mode = 0; c->reduce_mode = 0;
  {
    int all = 0;
    const int rc2 = comm_agree_bits(c, rc ? (c->last_err_bits ? c->last_err_bits : 1) : 0, &all);
    if (rc2) return rc2;
    if (all) return SZ_E_CAPACITY;
  }
  const int done = h[C_STOP] > 0 ? std::min(h[C_STOP], (int)nsteps) : nsteps;
  if (steps_done) *steps_done = done;
  {          // the forcing outputs of the last step that ran (see `beside` above)
    int want = 0;                                                        // (set 0: the columns as they were at entry)
    for (int s2 = 0; s2 < done; s2++) if (fset[s2] >= 0) want = fset[s2];
    if (want != cur_set) swap_frc();
  }
  if (done < nsteps) c->grid_live = false;          // stopped early: cells hold floes of a step that did not come
  c->inter_any = true; c->inter_lost = false;
  // status.fuse_idx of the step that ended the batch (as sz_step: only that step can have produced fuse pairs)
  if (done > 0 && (h[C_STOP] > 0 || (flags & SZ_NO_STOP))) {
    const int tlast = tstep0 + done - 1;
    const bool last_coupled = (flags & SZ_COUPLING_ON) && coupling_dt > 0 && (tlast % coupling_dt) == 0;
    c->gi_pending_n = h[C_NGHOSTS]; c->gi_pending_slot = (done - 1) & 1; c->gi_pending = true;
    rc = tile_fuse_replay(c, h, last_coupled);
    c->gi_pending = false;
  }
  c->fuse_lists.resize(c->hostM);
  return rc;
}

// counts of the last sz_halo_pack per destination rank (synchronises); used to size the exchange buffers
int sz_halo_counts(sz_ctx* c, int32_t nranks, int32_t* counts_out) {
  if (!c || !c->have_floes || nranks < 1 || nranks > 64 || !counts_out) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipMemcpyAsync(counts_out, c->S.cnt + C_COUNT, (size_t)nranks * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SZ_OK;
}

// waits for everything enqueued so far and reports sticky device errors
int sz_sync(sz_ctx* c) {
  if (!c || !c->have_floes) return SZ_E_STATE;
  (void)hipSetDevice(c->device);
  int h[C_COUNT];
  int rc = sync_and_check(c, h);
  if (rc) return rc;
  c->fuse_lists.resize(c->hostM);
  return SZ_OK;
}

// run on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream) instead of the
// library's own, so that collectives enqueued by the host framework order with the kernels
int sz_set_stream(sz_ctx* c, void* hip_stream) {
  if (!c) return SZ_E_ARG;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
  c->stream = (hipStream_t)hip_stream;
  return SZ_OK;
}

}  // extern "C"
