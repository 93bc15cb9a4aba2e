// sz_migrate.hpp — device side of sz_tile_migrate (floes that left their tile go to the tile that holds
// their centroid now; the in-reference analogue is the parent / ghost swap of collisions.jl:942-950).
//
// The movers are packed ON THE DEVICE, one stream of doubles per destination rank, and travel device to device (RCCL grouped
// send / receive; a host transport stages the mover bytes only).  A stream is
//     [0]                    number of records R
//     [1 .. 1 + MIG_DIR R)   directory: per record {global index, ring points, sub-floe points, offset of the record in the stream,
//                                                   rmax, cx, cy, -}
//     records                MIG_NCOL doubles (25 scalar columns, stress_accum / stress_instant / strain, id, status, global index,
//                            ring points, sub-floe points), the ring as {x, y} pairs, the sub-floe points x.. then y..
// The receiver merges the directories of all streams (sz_k_mig_dirs) -- the only part the host reads -- builds the new row order
// (kept and received floes by global index, as the single context holds them), and the rows are gathered on the device into
// that order (sz_k_mig_gather, sz_k_mig_points), kept floes from the old rows and received ones from the streams.
#pragma once
#include "sz_kernels.hpp"

namespace sz {

constexpr int MIG_NSC = 25;                 // scalar columns of a record, in the order of mig_column_table()
constexpr int MIG_NCOL = MIG_NSC + 12 + 5;  // + three 2 x 2 tensors, id, status, global index, ring points, sub-floe points
constexpr int MIG_DIR = 8;

// who owns floe i now: the tile that holds the centroid (periodic: of its image inside the domain), px x py tiles over the domain.
// tally[2 d], tally[2 d + 1]: movers to rank d and the doubles of their records
__global__ void sz_k_mig_owner(State S, int N, const int* owner_override, double x0, double y0, double Lx, double Ly, int px, int py, int perx, int pery,
                               int me, int nranks, int* owner, unsigned long long* tally, int* bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    int o;
    if (owner_override) {
      o = owner_override[i];
      if (o < 0 || o >= nranks) { atomicOr(bad, 1); o = me; }
    } else {
      double x = S.cx[i] - x0, y = S.cy[i] - y0;
      if (perx) { x = fmod(x, Lx); if (x < 0) x += Lx; }
      if (pery) { y = fmod(y, Ly); if (y < 0) y += Ly; }
      const int ix = max(0, min(px - 1, (int)(x / Lx * px))), iy = max(0, min(py - 1, (int)(y / Ly * py)));
      o = iy * px + ix;
    }
    owner[i] = o;
    if (o != me) {
      const int nv = S.voff[i + 1] - S.voff[i], ns = S.soff[i + 1] - S.soff[i];
      atomicAdd(&tally[2 * o], 1ull);
      atomicAdd(&tally[2 * o + 1], (unsigned long long)(MIG_NCOL + 2 * nv + 2 * ns));
    }
  }
}

// one wavefront per mover: its record into the stream of its destination.  base[d]: start of stream d in `send`; cntd[d]: records of
// stream d (the directory's length); cur[2 d], cur[2 d + 1]: next directory entry, next record offset behind the directory.
// cols: the MIG_NSC scalar columns, then stress_accum, stress_instant, strain.
__global__ void sz_k_mig_pack(State S, int N, const int* owner, int me, double* send, const long long* base, const int* cntd, unsigned long long* cur,
                              double* const* cols) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < N; i += nw) {
    const int d = owner[i];
    if (d == me) continue;
    const int vo = S.voff[i], nv = S.voff[i + 1] - vo, so = S.soff[i], ns = S.soff[i + 1] - so;
    unsigned e = 0, olo = 0, ohi = 0;
    if (lane == 0) {
      e = (unsigned)atomicAdd(&cur[2 * d], 1ull);
      const unsigned long long o = atomicAdd(&cur[2 * d + 1], (unsigned long long)(MIG_NCOL + 2 * nv + 2 * ns));
      olo = (unsigned)o; ohi = (unsigned)(o >> 32);
    }
    e = __shfl(e, 0); olo = __shfl(olo, 0); ohi = __shfl(ohi, 0);
    double* st = send + base[d];
    const size_t off = 1 + (size_t)MIG_DIR * cntd[d] + (((size_t)ohi << 32) | olo);
    double* rec = st + off;
    if (lane < MIG_NSC) rec[lane] = cols[lane][i];
    else if (lane < MIG_NSC + 12) { const int t = (lane - MIG_NSC) >> 2, q = (lane - MIG_NSC) & 3; rec[lane] = cols[MIG_NSC + t][(size_t)4 * i + q]; }
    else if (lane == 37) rec[37] = (double)S.id[i];
    else if (lane == 38) rec[38] = (double)S.status[i];
    else if (lane == 39) rec[39] = (double)S.okey[i];
    else if (lane == 40) rec[40] = (double)nv;
    else if (lane == 41) rec[41] = (double)ns;
    const double* ring = (const double*)(S.vxy + vo);
    for (int k = lane; k < 2 * nv; k += 64) rec[MIG_NCOL + k] = ring[k];
    double* sub = rec + MIG_NCOL + 2 * nv;
    for (int k = lane; k < ns; k += 64) { sub[k] = S.sx[so + k]; sub[ns + k] = S.sy[so + k]; }
    if (lane == 0) {
      double* dir = st + 1 + (size_t)MIG_DIR * e;
      dir[0] = (double)S.okey[i]; dir[1] = (double)nv; dir[2] = (double)ns; dir[3] = (double)off;
      dir[4] = S.rmax[i]; dir[5] = S.cx[i]; dir[6] = S.cy[i]; dir[7] = 0.0;
      st[0] = (double)cntd[d];
    }
  }
}

// receiver, one workgroup: the directories of all streams in one table -- out[0] = records R (or -1: a stream is inconsistent), entry e
// at out[MIG_DIR (1 + e)] with its offset made absolute in `recv`.  rbase / rsize: start and doubles of the stream of rank s.
__global__ void __launch_bounds__(256) sz_k_mig_dirs(const double* recv, const long long* rbase, const long long* rsize, int nranks, double* out, int cap) {
  __shared__ int pre[65];
  __shared__ int ok;
  if (threadIdx.x == 0) {
    int t = 0; ok = 1;
    for (int s = 0; s < nranks; s++) {
      pre[s] = t;
      if (rsize[s] > 0) {
        const double r = recv[rbase[s]];
        if (!(r >= 0.0) || 1.0 + (double)(MIG_DIR + MIG_NCOL) * r > (double)rsize[s]) ok = 0;
        else t += (int)r;
      }
    }
    pre[nranks] = t;
    if (t > cap) ok = 0;
    out[0] = ok ? (double)t : -1.0;
  }
  __syncthreads();
  if (!ok) return;
  for (int s = 0; s < nranks; s++) {
    const int cnt = pre[s + 1] - pre[s];
    const double* dir = recv + rbase[s] + 1;
    for (int k = threadIdx.x; k < cnt * MIG_DIR; k += blockDim.x) {
      double v = dir[k];
      if ((k & (MIG_DIR - 1)) == 3) v += (double)rbase[s];
      out[(size_t)MIG_DIR * (1 + pre[s]) + k] = v;
    }
  }
}

// new row r <- old row src[r] (>= 0) or received record -src[r] - 1 of the directory table: the columns into tmp[k * Nn + r]
// (k < 39: the record's first 39 doubles)
__global__ void sz_k_mig_gather(State S, int Nn, const int* src, const double* dirs, const double* recv, double* const* cols, double* tmp) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Nn; r += gridDim.x * blockDim.x) {
    const int s = src[r];
    if (s >= 0) {
      for (int k = 0; k < MIG_NSC; k++) tmp[(size_t)k * Nn + r] = cols[k][s];
      for (int t = 0; t < 3; t++) for (int q = 0; q < 4; q++) tmp[(size_t)(MIG_NSC + 4 * t + q) * Nn + r] = cols[MIG_NSC + t][(size_t)4 * s + q];
      tmp[(size_t)37 * Nn + r] = (double)S.id[s]; tmp[(size_t)38 * Nn + r] = (double)S.status[s];
    } else {
      const double* b = recv + (size_t)dirs[(size_t)MIG_DIR * (size_t)(-s) + 3];         // entry e = -s - 1 lives at MIG_DIR (1 + e)
      for (int k = 0; k < 39; k++) tmp[(size_t)k * Nn + r] = b[k];
    }
  }
}
// ... and back into the columns, with the links of a freshly placed ghost-free field (as sz_upload_floes leaves them)
__global__ void sz_k_mig_scatter(State S, int Nn, double* const* cols, const double* tmp) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Nn; r += gridDim.x * blockDim.x) {
    for (int k = 0; k < MIG_NSC; k++) cols[k][r] = tmp[(size_t)k * Nn + r];
    for (int t = 0; t < 3; t++) for (int q = 0; q < 4; q++) cols[MIG_NSC + t][(size_t)4 * r + q] = tmp[(size_t)(MIG_NSC + 4 * t + q) * Nn + r];
    S.id[r] = (long long)tmp[(size_t)37 * Nn + r]; S.status[r] = (int)tmp[(size_t)38 * Nn + r];
    S.ghost_id[r] = 0; S.parent[r] = r; S.ngh[r] = 0; S.frc_remove[r] = 0;
    for (int q = 0; q < MAX_GHOSTS; q++) S.gh[r * MAX_GHOSTS + q] = -1;
  }
}
// rings and sub-floe points of the new rows, one wavefront per row, into tv / tsx / tsy at the new offsets nvoff / nsoff
__global__ void sz_k_mig_points(State S, int Nn, const int* src, const double* dirs, const double* recv, const int* nvoff, const int* nsoff,
                                double2* tv, double* tsx, double* tsy) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < Nn; r += nw) {
    const int s = src[r];
    const int o = nvoff[r], nv = nvoff[r + 1] - o, so = nsoff[r], ns = nsoff[r + 1] - so;
    if (s >= 0) {
      const int vo = S.voff[s], po = S.soff[s];
      for (int k = lane; k < nv; k += 64) tv[o + k] = S.vxy[vo + k];
      for (int k = lane; k < ns; k += 64) { tsx[so + k] = S.sx[po + k]; tsy[so + k] = S.sy[po + k]; }
    } else {
      const double* b = recv + (size_t)dirs[(size_t)MIG_DIR * (size_t)(-s) + 3] + MIG_NCOL;
      for (int k = lane; k < nv; k += 64) tv[o + k] = make_double2(b[2 * k], b[2 * k + 1]);
      b += 2 * nv;
      for (int k = lane; k < ns; k += 64) { tsx[so + k] = b[k]; tsy[so + k] = b[ns + k]; }
    }
  }
}

}  // namespace sz
