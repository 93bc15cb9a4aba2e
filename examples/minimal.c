/* minimal.c -- the C-ABI from plain C: two overlapping floes in a periodic box, one timestep_collisions! call,
 * ten resident timesteps, the simplify check and a grid output.  Build (on a box with an MI355X):
 *   gcc -std=c11 -Iinclude examples/minimal.c -Lsubzero.jl_amd -lsubzero_hip -Wl,-rpath,$PWD/subzero.jl_amd -lm -o minimal
 * The host computes the derived floe columns (centroid, area, mass, moment, rmax) exactly as Floe(...) does in the
 * reference (floe.jl:144-200); here by the textbook formulas for the two squares used. */
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "subzero_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != SZ_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sz_last_error(ctx)); return 1; } } while (0)

int main(void) {
  sz_ctx *ctx = sz_create(0);
  if (!ctx) { fprintf(stderr, "no HIP device\n"); return 1; }
  /* domain: 100 km box, all four walls periodic (kinds N,S,E,W; 1 = periodic) */
  const double L = 1e5;
  int32_t kinds[4] = { 1, 1, 1, 1 };
  double vals[4] = { L, 0.0, L, 0.0 };                       /* N, S, E, W coordinates */
  double rects[16] = { 0, L, L, 1.5 * L,   0, L, -0.5 * L, 0,   L, 1.5 * L, 0, L,   -0.5 * L, 0, 0, L };  /* x0,xf,y0,yf each */
  CHECK(sz_set_domain(ctx, kinds, vals, rects, NULL, NULL));
  /* two 10 km squares overlapping by 500 m, closed rings, clockwise like the reference's examples */
  const double s = 1e4, h = 0.5, rho_i = 920.0;
  double vx[10], vy[10];
  double ox[2] = { 4.0e4, 4.95e4 }, oy[2] = { 4.5e4, 4.6e4 };
  for (int f = 0; f < 2; f++) {
    double px[5] = { 0, 0, s, s, 0 }, py[5] = { 0, s, s, 0, 0 };
    for (int k = 0; k < 5; k++) { vx[5 * f + k] = ox[f] + px[k]; vy[5 * f + k] = oy[f] + py[k]; }
  }
  int32_t vert_off[3] = { 0, 5, 10 }, sub_off[3] = { 0, 1, 2 };
  double cx[2], cy[2], rmax[2], area[2], height[2], mass[2], moment[2], u[2] = { 0.1, -0.1 }, v[2] = { 0, 0 }, xi[2] = { 0, 0 };
  double sx[2] = { 0, 0 }, sy[2] = { 0, 0 };
  int64_t id[2] = { 1, 2 };
  for (int f = 0; f < 2; f++) {
    cx[f] = ox[f] + s / 2; cy[f] = oy[f] + s / 2; area[f] = s * s; height[f] = h; rmax[f] = s / sqrt(2.0);
    mass[f] = area[f] * h * rho_i; moment[f] = mass[f] * (s * s + s * s) / 12.0;
  }
  sz_floe_columns c;
  memset(&c, 0, sizeof(c));
  c.cx = cx; c.cy = cy; c.rmax = rmax; c.area = area; c.height = height; c.mass = mass; c.moment = moment;
  c.u = u; c.v = v; c.xi = xi; c.id = id; c.vert_off = vert_off; c.vx = vx; c.vy = vy; c.sub_off = sub_off; c.sx = sx; c.sy = sy;
  CHECK(sz_upload_floes(ctx, 2, 2, &c));
  CHECK(sz_add_ghosts(ctx));
  CHECK(sz_timestep_collisions(ctx, 2, 10));
  sz_stats st;
  CHECK(sz_get_stats(ctx, &st));
  printf("pairs %lld, contact rows %lld\n", (long long)st.n_pairs, (long long)st.n_inter_rows);
  CHECK(sz_remove_ghosts(ctx));
  int32_t done = 0;
  CHECK(sz_step(ctx, 10, 0, 10, 10, SZ_COLLISIONS_ON, &done));      /* ten timesteps, state resident on the device */
  if (done != 10) { fprintf(stderr, "the batch stopped after %d steps: a floe was tagged\n", done); return 1; }
  double ucol[2];
  memset(&c, 0, sizeof(c)); c.u = ucol;
  CHECK(sz_download_floes(ctx, &c));
  printf("u after 10 steps: %.6f %.6f (the floes push each other apart)\n", ucol[0], ucol[1]);
  /* output step: does simplify_floes! have anything to do, and the GridOutputWriter averages on a 4 x 4 grid
     (write_data! sees the ghosts: add them for the call) */
  int64_t todo[4];
  CHECK(sz_simplify_check(ctx, 30, 1e6, 0.1, todo));
  printf("simplify: %lld remove, %lld fuse, %lld over max_vertices, %lld to dissolve\n", (long long)todo[0], (long long)todo[1],
         (long long)todo[2], (long long)todo[3]);
  double xg[5], yg[5], grid[2 * 4 * 4];
  for (int k = 0; k <= 4; k++) { xg[k] = k * L / 4; yg[k] = k * L / 4; }
  int32_t outs[2] = { SZ_EUL_SI_FRAC, SZ_EUL_U };
  CHECK(sz_add_ghosts(ctx));
  CHECK(sz_eulerian_data(ctx, 4, 4, xg, yg, 2, outs, grid));
  CHECK(sz_remove_ghosts(ctx));
  double ice = 0.0;
  for (int q = 0; q < 16; q++) ice += grid[q] * (L / 4) * (L / 4);
  printf("ice area from the grid output: %.4e m^2 (two floes of %.1e m^2)\n", ice, s * s);
  sz_destroy(ctx);
  return 0;
}
