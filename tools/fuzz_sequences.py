"""Random SEQUENCES of calls on small periodic fields, HIP path against the CPU oracle at every checkpoint (pairs and partner
numbers bit-exact, rows / state within 1e-9): resident batches of 1..6 steps, timestep_sim! step by step, the process-mode
sequence (add_ghosts / collisions / remove_ghosts / coupling / update), host edits that force an upload, and the downloads in
between -- what one call leaves behind (cell lists, ghost bookkeeping, order keys, lists) must not leak into the next.

    python tools/fuzz_sequences.py [nseeds] [nops]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity
import subzero_jl_amd
from subzero_jl_amd import fields, floe as floe_mod
from oracle import orc



def run(nseeds=8, nops=10, rtol=1e-9, device=0, verbose=True, walls=False, seed0=0):
  """returns the number of sequences that disagree (seeds seed0 .. seed0 + nseeds - 1)"""
  bad = 0; t00 = time.time()
  for seed in range(seed0, seed0 + nseeds):
      rng = np.random.default_rng(9000 + seed)
      n = int(rng.integers(150, 500))
      walled = walls and seed % 2 == 1          # every other field between collision walls with the strait's topography
      # (walled fields: 1e-8 -- floes pressed against walls and coasts for dozens of steps carry the round-off of the two codes' different
      #  summation orders to 1.1 .. 1.4e-9 in 3 of 60 sequences; with and without the fused element launch the numbers are the same bits)
      rtol_f = max(rtol, 1e-8) if walled else rtol
      cfg = fields.make_config(n_floes=n, seed=300 + seed, concentration=float(rng.choice([0.5, 0.8, 0.95])),
                               ocean=str(rng.choice(["uniform", "shear", "converge_diverge"])), walls=walled, topography=walled)
      if walled:
          # floes pressed against the walls and the coasts (element items every step: they ride in the neighbour launch in resident steps)
          cfg["u"] = cfg["u"] * 10.0 + rng.uniform(-1, 1); cfg["v"] = cfg["v"] * 10.0 + rng.uniform(-1, 1)
      else:
          # fast floes, field shifted so that parents straddle and cross the walls (swaps with their ghosts)
          cfg["u"] = cfg["u"] * 40.0 + rng.uniform(-3, 3); cfg["v"] = cfg["v"] * 40.0 + rng.uniform(-3, 3)
          cfg["vx"] = cfg["vx"] + rng.uniform(0, 2e4); cfg["vy"] = cfg["vy"] + rng.uniform(0, 2e4)
          cfg["derived"] = floe_mod.derive(cfg["vert_off"], cfg["vx"], cfg["vy"], cfg["height"])
      rng.integers(0, 4)          # (keeps the random streams of the seeds as they were when the 300-seed sweep was run)
      hw = fields.build_world(subzero_jl_amd.World(device), cfg); ow = fields.build_world(orc.World(), cfg); ow.set_threads(8)
      dt = cfg["dt"]; t = 0; log = []; fresh = False
      try:
          for op in range(nops):
              kind = int(rng.integers(0, 6))
              cdt = int(rng.choice([1, 2]))
              if kind <= 1:
                  k = int(rng.integers(1, 7)); log.append(f"run{k}")
                  assert hw.run(k, t, dt, coupling_dt=cdt, stop_on_tags=False) == k
                  for q in range(k):
                      ow.timestep_sim(t + q, dt, coupling_dt=cdt)
                  t += k
              elif kind == 2:
                  k = int(rng.integers(1, 4)); log.append(f"sim{k}")
                  for q in range(k):
                      hw.timestep_sim(t, dt, coupling_dt=cdt); ow.timestep_sim(t, dt, coupling_dt=cdt); t += 1
              elif kind == 3:
                  log.append("process")
                  for w in (hw, ow):
                      m = w.M
                      w.add_ghosts(); w.timestep_collisions(m, dt); w.remove_ghosts(m)
                      w.timestep_coupling(); w.timestep_floe_properties(dt)
                  t += 1
              elif kind == 4:
                  log.append("edit")
                  du = rng.uniform(-0.5, 0.5, n)
                  for w in (hw, ow):
                      u = w.get("u"); u += du; w.set("u", u)
              else:
                  if not fresh:          # pair lists and rows describe the last collision call: nothing to compare before one / after an edit
                      continue
                  log.append("downloads")
                  parity.compare_pairs(hw, ow); parity.compare_interactions(hw, ow, rtol_f)
                  continue
              fresh = kind != 4
              parity.compare_worlds(hw, ow, rtol=rtol_f, check_pairs=fresh, check_inter=fresh)
          st = hw.stats()
          if verbose: print(f"ok   seed {seed} n {n} {' '.join(log)}  steps {t} ghosts {st['n_ghosts']} retry {st['n_retry']}", flush=True)
      except (AssertionError, RuntimeError) as e:
          bad += 1
          print(f"FAIL seed {seed} n {n} after {' '.join(log)}: {str(e)[:300]}", flush=True)
  if verbose:
    print(f"{nseeds - bad}/{nseeds} sequences agree ({time.time() - t00:.0f} s)")
  return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 10, walls=len(sys.argv) > 3 and sys.argv[3] == "walls", seed0=int(sys.argv[4]) if len(sys.argv) > 4 else 0) else 0)
