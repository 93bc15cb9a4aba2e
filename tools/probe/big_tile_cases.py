"""a few large tiled cases with re-tiling against the single context (tests.test_tiles_gpu.migration_case): python tools/probe/big_tile_cases.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
  from tests import test_tiles_gpu as T
  for world, n, seed, steps, every, shape, fast, stop in [(4, 20000, 71, 30, 10, "star", True, False), (4, 12000, 72, 24, 8, "star", True, False),
                                                            (2, 16000, 73, 24, 12, "star", False, False), (4, 14400, 74, 16, 8, "voronoi", True, False),
                                                            (4, 16000, 75, 30, 10, "walls-topo", True, False), (3, 9000, 76, 40, 7, "star", True, True)]:
      t = time.time()
      try:
          moved = T.migration_case(world, n, seed, steps, every, verbose=False, shape=shape, fast=fast, stop=stop)
          print(f"world {world} n {n} {shape} fast={fast} stop={stop} steps {steps} re-tile every {every}: bit-equal, {moved} floes changed tile ({time.time() - t:.1f} s)", flush=True)
      except AssertionError as e:
          print(f"world {world} n {n} {shape} fast={fast} stop={stop} steps {steps} re-tile every {every}: FAILED {str(e)[:1200]}", flush=True)


if __name__ == "__main__":
    main()
