// sz_pipeline.hpp — pipelined resident steps (round 4): two launches per timestep instead of three.
//
// The AB2 update of timestep_floe_properties! (update_floe.jl:502-545) moves a floe with the velocities it HAD at the start of the step:
//     centroid += 1.5 dt (u, v) - 0.5 dt (p_dxdt, p_dydt)         alpha += 1.5 dt xi - 0.5 dt p_dalphadt
// and only the velocities take the step's forces.  So the geometry of step t + 1 -- rings, boxes, cells, the periodic ghosts and
// their order -- does not depend on the contact forces of step t, and neither does the neighbour search of step t + 1.  A step is
//     L1(t) = narrow phase(t)  |  forcings(t)  |  GEO(t): rings, boxes, records, cells and ghost GEOMETRY of step t + 1
//     L2(t) = VEL(t): totals -> velocities, stress, thermodynamics  |  neighbour search(t + 1)
// with the pieces of a launch side by side in one grid (horizontal fusion: the narrow phase of a small field is one round whose length
// is set by its slowest wavefront; the chip is mostly idle beside it).  What one piece writes while another still reads the old
// version is double-buffered by step parity: rings (vxy) and collision records (crec) -- narrow(t) reads parity p while GEO(t) writes
// 1 - p --, the cell lists, the work list and its counters, the ghost links, and the ROWS of ghosts (State::goff: two regions behind the
// parents).  The floe columns stay single: cx, cy and the ring boxes are brought up to date from the new record by VEL(t), after
// everything of L1(t) that reads them (forcings, element items) is done.
// The results are bit for bit those of the three-launch steps (same expressions on the same values): the GPU suite compares tiles --
// which keep the three-launch steps -- with the single context.
// Reference semantics: simulation.jl:94-170 (order of the processes), collisions.jl:881-1174 (ghosts), update_floe.jl:469-551.
#pragma once
#include "sz_kernels.hpp"

namespace sz {

// the State of the other parity
__device__ __forceinline__ State pipe_other(State S, const PipeAlt& A) {
  S.crec = A.crec; S.vxy = A.vxy; S.cell_cnt = A.cell_cnt; S.cell_slots = A.cell_slots; S.cell_ovf = A.cell_ovf; S.cell_items = A.cell_items;
  S.work = A.work; S.wq = A.wq; S.gh = A.gh; S.ngh = A.ngh; S.goff = A.goff; S.gslot = A.gslot;
  return S;
}

// ---------------------------------------------------------------- GEO(t): thread per parent
// the parent's row as its ghosts copy it, without the ring (GhostRow's scalars)
struct GhostRowLite { double cx, cy, b0, b1, b2, b3, rmax, area, h, mass, mom, al, u, v, xi, over, tc, ts; long long id, oki; int st; signed char os; };
// the moved ring of a parent, point by point: the old ring from memory through the step's motion -- the expressions of the stores in geo_body
struct RingMoved {
  const double2* src; double cx, cy, dx, dy, cda, sda;
  __device__ __forceinline__ void get(int k, double& mx, double& my) const {
    const double2 p = src[k];
    const double x = p.x + (-cx), y = p.y + (-cy);
    const double xr = cda * x - sda * y, yr = sda * x + cda * y;
    mx = xr + (cx + dx); my = yr + (cy + dy);
  }
};
// S: the step's own parity (what narrow(t) reads); A: where the geometry of step t + 1 goes.  The move is _move_floe! (floe_utils.jl:82-93)
// with the expressions of sz_k_integrate<true> (same bits); the ghosts are made by ghost_inline_make on the moved ring, with the
// kinematic columns of the parent as they are NOW -- VEL(t) overwrites them with the step's new values.  The launch this rides in is
// compiled for the narrow phase's register budget: the ring is held once, for the move, and a parent that gets ghosts (a few per cent)
// reads it again point by point.
__device__ __forceinline__ void geo_body(State S, const PipeAlt& A, int dt, int bid, int nblk, int N) {
  const StopRegs stop = stop_load(S);
  State So = pipe_other(S, A);
  So.pipe = 1;
  const GridGeo geo = grid_geo(S);
  const int nv0 = S.voff[N];
  const double wall[4] = { S.eval[0], S.eval[1], S.eval[2], S.eval[3] };
  bool tested = false;
  for (int i = bid * (int)blockDim.x + (int)threadIdx.x; i < N; i += nblk * (int)blockDim.x) {
    const double cx = S.cx[i], cy = S.cy[i];
    const double u = S.u[i], v = S.v[i], xi = S.xi[i];
    const double p_dxdt = S.p_dxdt[i], p_dydt = S.p_dydt[i], p_dalphadt = S.p_dalphadt[i];
    const double rmx = S.rmax[i];
    const int st0 = S.status[i];
    const int o = S.voff[i], n = S.voff[i + 1] - o;
    if (!tested) {
      loads_issued();
      if (stop_test(S, stop)) break;
      tested = true;
    }
    const double dx = 1.5 * dt * u - 0.5 * dt * p_dxdt;
    const double dy = 1.5 * dt * v - 0.5 * dt * p_dydt;
    int cell_c, cell_s;
    { int ix, iy; cell_of(geo, cx + dx, cy + dy, ix, iy); cell_c = iy * geo.ncx + ix; cell_s = atomicAdd(&So.cell_cnt[cell_c], 1); }
    const double da = 1.5 * dt * xi - 0.5 * dt * p_dalphadt;
    double cda, sda;
    sincos(da, &sda, &cda);
    const double ncx = cx + dx, ncy = cy + dy;
    double bx0 = __builtin_inf(), bx1 = -__builtin_inf(), by0 = __builtin_inf(), by1 = -__builtin_inf();
    {
      double2 p[MV_RING];
#pragma unroll
      for (int k = 0; k < MV_RING; k++) p[k] = k < n ? S.vxy[o + k] : make_double2(0.0, 0.0);
      // the floe's cell entry first (its counter was drawn above; see sz_k_integrate)
      if (cell_s < CELL_K) So.cell_slots[(size_t)cell_c * CELL_K + cell_s] = i;
      else So.cell_items[i] = atomicExch(&So.cell_ovf[cell_c], i + 1) - 1;
#pragma unroll
      for (int k = 0; k < MV_RING; k++) {
        if (k < n) {
          const double x = p[k].x + (-cx), y = p[k].y + (-cy);
          const double xr = cda * x - sda * y, yr = sda * x + cda * y;
          const double mx = xr + (cx + dx), my = yr + (cy + dy);
          So.vxy[o + k] = make_double2(mx, my);
          bx0 = fmin(bx0, mx); bx1 = fmax(bx1, mx); by0 = fmin(by0, my); by1 = fmax(by1, my);
        }
      }
    }
    {          // the geometry quads of the floe's next record (rmax, id, order key, ring size and sign: seeded; u, v, xi, area, height: VEL)
      double2* r = So.crec + (size_t)i * 8;
      r[0] = make_double2(ncx, ncy); r[2].y = crec_vp(o, i, 0); r[3] = make_double2(bx0, bx1); r[4] = make_double2(by0, by1);
    }
    const int gf = A.make_ghosts ? ghost_flag_of(wall, S.any_periodic_ew, S.any_periodic_ns, ncx, ncy, rmx, bx0, bx1, by0, by1, st0 == SZ_ACTIVE) : 5;
    if (gf != 5) {
      GhostRowLite R;
      R.cx = ncx; R.cy = ncy; R.b0 = bx0; R.b1 = bx1; R.b2 = by0; R.b3 = by1;
      R.rmax = rmx; R.area = S.area[i]; R.h = S.height[i]; R.mass = 0.0; R.mom = 0.0; R.al = 0.0; R.u = u; R.v = v; R.xi = xi;
      R.over = S.overarea[i]; R.id = S.id[i]; R.oki = S.okey[i]; R.os = S.osign[i]; R.st = SZ_ACTIVE; R.tc = 1.0; R.ts = 0.0;
      const RingMoved ring{ S.vxy + o, cx, cy, dx, dy, cda, sda };
      ghost_inline_make(So, geo, wall, N, nv0, A.gslot, i, gf, n, o, R, ring);
    } else {
      So.ngh[i] = 0;          // (the links of this parity: no ghosts in the coming step)
    }
  }
}

// ---------------------------------------------------------------- VEL(t): thread per parent
// S: the parity of step t + 1 (its record gets the new u, v, xi, height; the geometry GEO(t) left in it goes into the columns); A: step t's
// (ghost links for the fold of the ghosts' totals, the work-list counters to clear).  The arithmetic is sz_k_integrate's, expression
// for expression (update_floe.jl:469-551); the ring is not touched: calc_strain! is evaluated once, behind the batch.
// acc_mode: bit 1 = the host knows this is the batch's last step
__device__ __forceinline__ void vel_body(State S, const Params& P, const PipeAlt& A, int dt, int apply_frc, int bid, int nblk, int N, int acc_mode) {
  const int step = S.step - 1;                 // (the State carries the step number of the neighbour search beside this: t + 1)
  const int cs = S.cnt[C_STOP], cr = S.cnt[C_RETRYSTOP], frcstop = S.cnt[C_FRCSTOP];
  const bool halted = (cs != 0 && step > cs) || (cr != 0 && step >= cr);        // stop_test_late for this step
  const bool last_step = (acc_mode & 2) != 0 || ((S.stop_on_tags || S.restart_on_tags) && (cs == step || frcstop == step));
  if (bid == 0 && !halted) {
    if (threadIdx.x < 2 * NSEG) A.wq[(threadIdx.x >> 1) * 32 + (threadIdx.x & 1)] = 0;        // narrow(t) has consumed its work list
    if (threadIdx.x == 0) { S.galloc[(1 - S.gslot) * 16] = 0ull; S.galloc[(1 - S.gslot) * 16 + 1] = 0ull; }      // the allocator GEO(t + 1) draws from
  }
  int wh = 0, wf = 0, wv = 0, wx = 0;
  bool tested = false;
  // Three batches of loads instead of one: the launch is compiled for the search's register budget (three wavefronts per SIMD), and the
  // update is not what the launch waits for -- the search beside it takes longer.  (The stores of a phase keep the compiler from
  // hoisting the next phase's loads above them.)
  for (int i = bid * (int)blockDim.x + (int)threadIdx.x; i < N; i += nblk * (int)blockDim.x) {
    // ---- phase 1: the step's totals (fixed-point words of the floe and of its ghosts), tags
    const int st0 = S.status[i], ngh0 = A.ngh[i] & 0xff;
    const double rmx = S.rmax[i], area = S.area[i], height0 = S.height[i];
    longlong4 fa0, fa1, fa2, fa3;
    { const longlong4* a = (const longlong4*)(S.facc + (size_t)i * FX_WORDS); fa0 = a[0]; fa1 = a[1]; fa2 = a[2]; fa3 = a[3]; }
    double g_over = S.overarea[i];
    const int frc_rm = apply_frc ? S.frc_remove[i] : 0;
    if (!tested) {
      loads_issued();
      if (halted) break;
      tested = true;
    }
    double cfx, cfy, ctrq, s11 = 0, s12 = 0, s21 = 0, s22 = 0;
    int st_new = st0; bool st_dirty = false;
    {
      long long q[7] = { fa0.x, fa0.y, fa0.z, fa0.w, fa1.x, fa1.y, fa1.z }, ql[7] = { fa2.x, fa2.y, fa2.z, fa2.w, fa3.x, fa3.y, fa3.z };
      const int tagb = (int)(fa1.w & 0xffffffffll);
      const bool dirty = (fa0.x | fa0.y | fa0.z | fa0.w | fa1.x | fa1.y | fa1.z | fa1.w | fa2.x | fa2.y | fa2.z | fa2.w | fa3.x | fa3.y | fa3.z) != 0;
      if (ngh0 != 0) {
        for (int g3i = 0; g3i < MAX_GHOSTS; g3i++) {
          const int g = A.gh[i * MAX_GHOSTS + g3i];
          if (g >= 0) {
            const longlong4* a = (const longlong4*)(S.facc + (size_t)g * FX_WORDS); const longlong4 b0 = a[0], b1 = a[1], b2 = a[2], b3 = a[3];
            q[0] += b0.x; q[1] += b0.y; q[2] += b0.z; q[3] += b0.w; q[4] += b1.x; q[5] += b1.y; q[6] += b1.z;
            ql[0] += b2.x; ql[1] += b2.y; ql[2] += b2.z; ql[3] += b2.w; ql[4] += b3.x; ql[5] += b3.y; ql[6] += b3.z;
          }
        }
      }
      const int eF = fx_force_exp(S.kexp, area, height0), eT = eF + fx_lever_exp(rmx);
      cfx = fx_join(q[0], ql[0], eF); cfy = fx_join(q[1], ql[1], eF); ctrq = fx_join(q[4] - q[3], ql[4] - ql[3], eT);
      if ((q[0] | q[1] | q[2] | q[3] | q[4] | q[5] | ql[0] | ql[1] | ql[2] | ql[3] | ql[4] | ql[5]) != 0) {
        const double sc = 1 / (area * height0);
        s11 = fx_join(q[2], ql[2], eT) * sc; s12 = fx_join(q[3] + q[4], ql[3] + ql[4], eT) * 0.5 * sc; s21 = s12; s22 = fx_join(q[5], ql[5], eT) * sc;
      }
      if ((q[6] | ql[6]) != 0) { g_over = g_over + fx_join(q[6], ql[6], fx_area_exp(area)); S.overarea[i] = g_over; }
      if (tagb & 1) st_new = SZ_FUSE;
      if (tagb & 2) st_new = SZ_REMOVE;
      if (tagb & 4) st_new = SZ_FUSE;
      st_dirty = tagb != 0 || st0 != SZ_ACTIVE;
      S.cfx[i] = cfx; S.cfy[i] = cfy; S.ctrq[i] = ctrq;
      if (dirty) { longlong4* a = (longlong4*)(S.facc + (size_t)i * FX_WORDS); a[0] = make_longlong4(0, 0, 0, 0); a[1] = make_longlong4(0, 0, 0, 0); a[2] = make_longlong4(0, 0, 0, 0); a[3] = make_longlong4(0, 0, 0, 0); }
    }
    if (st_dirty) S.status[i] = st_new;
    if (frc_rm || st_new != SZ_ACTIVE) {
      if (frc_rm) { S.status[i] = SZ_REMOVE; st_new = SZ_REMOVE; }
      if (step > 0 && S.stop_on_tags) S.cnt[C_STOP] = step;          // (request_stop for THIS step: the State carries the search's number)
    }
    {          // calc_stress! / _update_stress_accum! (update_floe.jl:392-414, stress_calculators.jl:118-122)
      const double l = P.lambda;
      const double4 sa4 = *(const double4*)(S.sa + (size_t)i * 4);
      *(double4*)(S.sa + (size_t)i * 4) = make_double4((1 - l) * sa4.x + l * s11, (1 - l) * sa4.y + l * s12, (1 - l) * sa4.z + l * s21, (1 - l) * sa4.w + l * s22);
      *(double4*)(S.si + (size_t)i * 4) = make_double4(s11, s12, s21, s22);
    }
    // ---- phase 2: the update proper (update_floe.jl:482-545), sz_k_integrate's expressions
    const double mass0 = S.mass[i], moment0 = S.moment[i], hflx = S.hflx[i];
    const double u = S.u[i], v = S.v[i], xi = S.xi[i], alpha0 = S.alpha[i];
    const double p_dxdt = S.p_dxdt[i], p_dydt = S.p_dydt[i], p_dalphadt = S.p_dalphadt[i];
    const double p_dudt = S.p_dudt[i], p_dvdt = S.p_dvdt[i], p_dxidt = S.p_dxidt[i];
    const double fxOA = S.fxOA[i], fyOA = S.fyOA[i], trqOA = S.trqOA[i];
    double hh = height0;
    if (hh > P.max_h) { hh = P.max_h; wh++; }
    double mass = mass0;
    for (int it = 0; it < 400 && fmax(fabs(cfx), fabs(cfy)) > mass / (5 * dt); it++) { cfx = cfx / 10; cfy = cfy / 10; ctrq = ctrq / 10; wf++; }
    double h = hh;
    double dh = hflx / h;
    double hfrac = (h + dh) / h;
    mass *= hfrac; double moment = moment0 * hfrac; h -= dh;
    double da = 1.5 * dt * xi - 0.5 * dt * p_dalphadt;
    const double al = alpha0 + da;
    double cal, sal;
    sincos(al, &sal, &cal);
    double dudt = (fxOA + cfx) / mass, dvdt = (fyOA + cfy) / mass;
    double frac = 1.0, au = fabs(dt * dudt), av = fabs(dt * dvdt), h2 = h / 2;
    if (au > h2 && av > h2) {
      double f1 = (sgn(dudt) * h / (2 * dt)) / dudt, f2 = (sgn(dvdt) * h / (2 * dt)) / dvdt;
      frac = f1 < f2 ? f1 : f2;
    } else if (au > h2 && av < h2) frac = (sgn(dudt) * h / (2 * dt)) / dudt;
    else if (au < h2 && av > h2) frac = (sgn(dvdt) * h / (2 * dt)) / dvdt;
    if (frac != 1) { dudt = frac * dudt; dvdt = frac * dvdt; wv++; }
    const double nu = u + (1.5 * dt * dudt - 0.5 * dt * p_dudt);
    const double nv = v + (1.5 * dt * dvdt - 0.5 * dt * p_dvdt);
    double dxidt = (trqOA + ctrq) / moment;
    dxidt = frac * dxidt;
    double nxi = xi + 1.5 * dt * dxidt - 0.5 * dt * p_dxidt;
    if (fabs(nxi) > P.max_xi) { nxi = sgn(nxi) * P.max_xi; wx++; }
    S.mass[i] = mass; S.moment[i] = moment; S.height[i] = h;
    S.alpha[i] = al;
    *(double2*)(S.trig + (size_t)i * 2) = make_double2(cal, sal);
    S.p_dxdt[i] = u; S.p_dydt[i] = v; S.p_dalphadt[i] = xi;
    S.u[i] = nu; S.v[i] = nv;
    S.p_dudt[i] = dudt; S.p_dvdt[i] = dvdt;
    S.xi[i] = nxi; S.p_dxidt[i] = dxidt;
    // ---- phase 3: the columns follow the record GEO(t) wrote (for a parent that swapped with its ghost: the swapped place); the record
    // and the ghosts GEO(t) made of this parent take the new kinematic columns (deepcopy of the parent, collisions.jl:893)
    const int ngh1 = S.ngh[i];
    const double cx = S.cx[i], cy = S.cy[i];
    double2* r = S.crec + (size_t)i * 8;
    const double2 g0 = r[0], g3 = r[3], g4 = r[4];
    if (last_step) {          // behind the batch: the rows of this step (levers about the old centroid) and, after a stop, the un-swap of the parents
      double cda, sda;
      sincos(da, &sda, &cda);
      *(double4*)(S.mot + (size_t)i * 4) = make_double4(cx, cy, 1.5 * dt * u - 0.5 * dt * p_dxdt, 1.5 * dt * v - 0.5 * dt * p_dydt);
      *(double2*)(S.mot2 + (size_t)i * 2) = make_double2(cda, sda);
    }
    S.cx[i] = g0.x; S.cy[i] = g0.y; S.bbx0[i] = g3.x; S.bbx1[i] = g3.y; S.bby0[i] = g4.x; S.bby1[i] = g4.y;
    r[5] = make_double2(nu, nv); r[6] = make_double2(nxi, area); r[7].x = h;
    for (int q = 0; q < (ngh1 & 0xff); q++) {
      const int g = S.gh[i * MAX_GHOSTS + q];
      if (g < 0) continue;
      double2* rg = S.crec + (size_t)g * 8;
      rg[5] = make_double2(nu, nv); rg[6] = make_double2(nxi, area); rg[7].x = h;
      S.u[g] = nu; S.v[g] = nv; S.xi[g] = nxi; S.height[g] = h; S.overarea[g] = g_over;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) { wh += __shfl_xor(wh, d); wf += __shfl_xor(wf, d); wv += __shfl_xor(wv, d); wx += __shfl_xor(wx, d); }
  if ((threadIdx.x & 63) == 0) {
    int* w = S.warn + (((bid * blockDim.x + threadIdx.x) >> 6) % WARN_SLOTS) * 32;
    if (wh) atomicAdd(w + 0, wh);
    if (wf) atomicAdd(w + 1, wf);
    if (wv) atomicAdd(w + 2, wv);
    if (wx) atomicAdd(w + 3, wx);
  }
}

// L2(t): workgroups [0, nbv) update the parents, the others search the neighbours of step t + 1 (the search's REC instantiations)
// nbe / epoch (fields between walls: no periodic pair, no ghosts): nbe more workgroups behind the search's count, scan and fill the floe-wall /
// floe-topography items of step t + 1 (sz_k_elem_scan_fill, from the collision records)
template <bool FAM>
__global__ void __launch_bounds__(NB_TPB, 3) sz_k_vel_search(State S, Params P, PipeAlt A, int dt, int apply_frc, int nbv, int N, int acc_mode, int nbe, unsigned epoch) {
  if ((int)blockIdx.x < nbv) { vel_body(S, P, A, dt, apply_frc, (int)blockIdx.x, nbv, N, acc_mode); return; }
  // (a floe the forcings of step t leave without an in-bounds point is tagged by VEL(t), in this very launch: the hint says so a launch earlier)
  if ((S.stop_on_tags || S.restart_on_tags) && S.cnt[C_FRCSTOP] == S.step - 1 && S.step > 1) return;
  const int nbs = (int)gridDim.x - nbv - nbe;
  if ((int)blockIdx.x < nbv + nbs) neighbors_body<NB_TPB, FAM, MAXNB, true>(S, (int)blockIdx.x - nbv, nbs);
  else elem_scan_fill_body<true>(S, epoch, (int)blockIdx.x - nbv - nbs, N);
}

// ---------------------------------------------------------------- behind a pipelined batch
// A batch that a tag ended at step t (known only while L2(t) ran) has the geometry of step t + 1 as GEO(t) made it: parents that left the
// domain have ALREADY swapped with their ghosts -- in the reference that happens in the add_ghosts! of the next step.  The state handed
// back is the unswapped one: the move is evaluated once more from the ring of step t (the other parity: intact) with the motion VEL(t)
// left in mot / mot2 -- the expressions of geo_body, the same bits -- for the parents GEO marked (ngh & 0x100).
// S: parity of step t + 1 (what becomes the context's state); A: parity of step t.
__global__ void __launch_bounds__(128) sz_k_unswap(State S, PipeAlt A, int N) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    if (!(S.ngh[i] & 0x100)) continue;
    const double4 m = *(const double4*)(S.mot + (size_t)i * 4);
    const double2 t = *(const double2*)(S.mot2 + (size_t)i * 2);
    const double cx = m.x, cy = m.y, dx = m.z, dy = m.w, cda = t.x, sda = t.y;
    const int o = S.voff[i], n = S.voff[i + 1] - o;
    double bx0 = __builtin_inf(), bx1 = -__builtin_inf(), by0 = __builtin_inf(), by1 = -__builtin_inf();
    for (int k = 0; k < n; k++) {
      const double2 p = A.vxy[o + k];
      const double x = p.x + (-cx), y = p.y + (-cy);
      const double xr = cda * x - sda * y, yr = sda * x + cda * y;
      const double mx = xr + (cx + dx), my = yr + (cy + dy);
      S.vxy[o + k] = make_double2(mx, my);
      bx0 = fmin(bx0, mx); bx1 = fmax(bx1, mx); by0 = fmin(by0, my); by1 = fmax(by1, my);
    }
    const double ncx = cx + dx, ncy = cy + dy;
    S.cx[i] = ncx; S.cy[i] = ncy; S.bbx0[i] = bx0; S.bbx1[i] = bx1; S.bby0[i] = by0; S.bby1[i] = by1;
    double2* r = S.crec + (size_t)i * 8;
    r[0] = make_double2(ncx, ncy); r[3] = make_double2(bx0, bx1); r[4] = make_double2(by0, by1);
  }
}

// The per-row results of a batch's last step -- neighbour lists, item infos, tags, row counts of the ghosts -- lie in the step's row
// region; everything that looks at them after the batch (fuse replay, sz_download_pairs, statistics) expects the ghosts straight behind
// the parents.  Region 1 is moved down and the row numbers inside the lists renamed.  One launch per batch, only when the last step used
// region 1.  G: ghosts of that step.
__global__ void __launch_bounds__(256) sz_k_rows_home(State S, int N, int G, int goff) {
  const int nb = S.maxnb, M = N + G;
  // pass 1: the ghost rows' own data (a thread per (row, entry))
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)G * nb; t += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(t / nb), e = (int)(t % nb);
    const size_t src = (size_t)(N + goff + g) * nb + e, dst = (size_t)(N + g) * nb + e;
    S.nb_out[dst] = S.nb_out[src]; S.nb_in[dst] = S.nb_in[src]; S.it_info[dst] = S.it_info[src];
    if (e == 0) {
      const int a = N + goff + g, b = N + g;
      S.n_out[b] = S.n_out[a]; S.n_in[b] = S.n_in[a]; S.tagA[b] = S.tagA[a]; S.status[b] = S.status[a]; S.inter_cnt[b] = S.inter_cnt[a];
      S.okey[b] = S.okey[a]; S.ghost_id[b] = S.ghost_id[a]; S.parent[b] = S.parent[a]; S.id[b] = S.id[a];
      S.voff[b] = S.voff[a]; if (g == G - 1) S.voff[b + 1] = S.voff[a + 1];
      S.cx[b] = S.cx[a]; S.cy[b] = S.cy[a];
    }
  }
  (void)M;
}
__global__ void __launch_bounds__(256) sz_k_rows_rename(State S, int N, int G, int goff) {
  const int nb = S.maxnb, M = N + G;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)M * nb; t += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(t / nb), e = (int)(t % nb);
    if (e < S.n_out[k]) { const int j = S.nb_out[t]; if (j >= N + goff) S.nb_out[t] = j - goff; }
    if (e < S.n_in[k]) { const int j = S.nb_in[t]; if (j >= N + goff) S.nb_in[t] = j - goff; }
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
    for (int q = 0; q < MAX_GHOSTS; q++) { const int g = S.gh[i * MAX_GHOSTS + q]; if (g >= N + goff) S.gh[i * MAX_GHOSTS + q] = g - goff; }
}

}  // namespace sz
