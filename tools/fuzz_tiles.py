"""Tiled runs with re-tiling against the single context, over random cases: 2 or 4 ranks sharing the GPU (library exchange over the host
transport), fast floes crossing tile edges and periodic walls, a re-tile (sz_tile_migrate, device path) every few steps; every owned column
must be bit-equal to the single context's.      python tools/fuzz_tiles.py [cases] [seed0] [mixed | walls]"""
import os, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def case_params(seed, mixed=False, walls=False):
    """the case a seed stands for: (world, n, steps, every, kind, shape, fast, stop)"""
    rnd = random.Random(seed)
    world = rnd.choice([2, 2, 4])
    n = rnd.randrange(500, 1600) if world == 2 else rnd.randrange(1000, 2400)
    every = rnd.randrange(4, 16)
    steps = every * rnd.randrange(2, 5) + rnd.randrange(0, every)
    # (the variants are drawn after the parameters above, so that a seed means the same fast star field as in the first version of this tool)
    kinds = ["fast", "fast", "slow", "voronoi", "voronoi-fast", "fast-stop", "voronoi-stop"]
    if walls:
        kinds = ["walls", "walls-fast", "walls-topo", "walls-topo-fast", "walls-fast-stop"]
    kind = rnd.choice(kinds) if mixed or walls else "fast"
    shape = "voronoi" if kind.startswith("voronoi") else "walls-topo" if kind.startswith("walls-topo") else "walls" if kind.startswith("walls") else "star"
    fast = "fast" in kind
    stop = kind.endswith("stop")
    if shape == "voronoi":
        n = max(n, 900 if world == 2 else 3600)          # (a tile holds at most as many halo floes as owned ones)
        steps = min(steps, 24)
    return world, n, steps, every, kind, shape, fast, stop


def main():
    from tests import test_tiles_gpu as T
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    walls = len(sys.argv) > 3 and sys.argv[3] == "walls"          # fields between collision walls (no ghosts; floe - wall and floe - topography items)
    mixed = len(sys.argv) > 3 and sys.argv[3] == "mixed"          # also slow fields, Voronoi fields (touching cells, a size spectrum), batches that end on a tag
    ok = 0
    for c in range(cases):
        world, n, steps, every, kind, shape, fast, stop = case_params(seed0 + c, mixed, walls)
        t = time.time()
        try:
            moved = T.migration_case(world, n, seed0 + c, steps, every, verbose=False, shape=shape, fast=fast, stop=stop)
            ok += 1
            print(f"case {c} [{kind}]: world {world} n {n} steps {steps} re-tile every {every}: bit-equal, {moved} floes changed tile ({time.time() - t:.1f} s)", flush=True)
        except AssertionError as e:
            print(f"case {c} [{kind}]: world {world} n {n} steps {steps} re-tile every {every}: FAILED {str(e)[:1500]}", flush=True)
    print(f"{ok} / {cases} cases bit-equal")
    sys.exit(0 if ok == cases else 1)


if __name__ == "__main__":
    main()
