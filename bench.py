#!/usr/bin/env python3
"""bench.py — floe-steps/s of the HIP collision / forcing / rigid-body engine on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one timestep_sim! of the hot path (add_ghosts! -> timestep_collisions! -> ghost
removal -> timestep_coupling! -> timestep_floe_properties!) over the whole synthetic floe field,
state resident in HBM.  At N=1 the workload is BASELINE.json configs[1]: 10 000 random-polygon
floes (8-16 vertices), doubly periodic box, uniform_flow ocean forcing, fp64.  For N>1 (one
process per GPU under torch.distributed / RCCL) the default is the metric's multi-GPU workload,
BASELINE.json configs[2]: 100 000 floes, converge/diverge flow, sharded by spatial tile with a
ghost-floe halo traded every step (subzero_jl_amd.tiles) -- the job size is fixed, i.e. STRONG
scaling; `--floes K` runs N x K floes of the configs[1] field instead (weak scaling).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (the narrow phase) with
the algorithmic byte count of SURVEY.md §8(d) against HBM peak; `cpu_baseline` is the CPU oracle
(a C/OpenMP port of the reference's algorithm) timed on the host cores of the same box.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
W = 8                        # fp64 bytes


WATCHDOG_AFTER_SETUP_S = 120.0


def log(msg):
    """progress line on stderr: what the self-launching parent's watchdog (and a human) sees of a running rank"""
    sys.stderr.write(f"[bench rank {os.environ.get('RANK', '0')}] {msg}\n"); sys.stderr.flush()


def launch_ranks(n, child_cmd, watchdog_s=900.0, shared_gpu=False, extra_env=None, out=None, err=None, poll_s=0.2):
    """`python3 bench.py --gpus N` without an external launcher: N fresh child processes, one per GPU, started BEFORE this
    process has imported torch or touched HIP (it never does: the parent only waits).  Every child gets RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run would set them; rank 0's stdout (the ONE JSON line) is
    passed through, every child's stderr is passed through with its rank in front.  Watchdog: when NO child has written
    anything for `watchdog_s` seconds, or one has exited non-zero, the children still running are killed (their exact PIDs)
    and the parent exits non-zero with the stderr tails.  Returns the exit code."""
    import collections
    import socket
    import subprocess
    import threading
    out = out or sys.stdout; err = err or sys.stderr
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs, tails, threads = [], [], []
    last = [time.monotonic()]
    lines0 = []
    set_up = set()          # ranks that have said "tile set up": from then on every phase of every rank writes a line, and the watchdog is short

    def pump(stream, rank, is_out):
        for raw in iter(stream.readline, b""):
            last[0] = time.monotonic()
            line = raw.decode(errors="replace")
            if is_out:
                lines0.append(line)
            else:
                tails[rank].append(line)
                if "tile set up" in line:
                    set_up.add(rank)
                err.write(f"[rank {rank}] {line}"); err.flush()
        stream.close()

    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(0 if shared_gpu else r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        p = subprocess.Popen(list(child_cmd), env=env, stdin=subprocess.DEVNULL, stderr=subprocess.PIPE,
                             stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL)
        procs.append(p); tails.append(collections.deque(maxlen=40))
        t = threading.Thread(target=pump, args=(p.stderr, r, False), daemon=True); t.start(); threads.append(t)
        if r == 0:
            t = threading.Thread(target=pump, args=(p.stdout, 0, True), daemon=True); t.start(); threads.append(t)

    def kill_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill(); p.wait()

    rc, why = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = 1, f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            break
        # (building the 100 k-floe field and the first RCCL communicator are silent for tens of seconds; once every rank has its tile, a rank
        #  reports each phase and a step takes a fraction of a millisecond: silence then means a hung exchange, and a second attempt on
        #  the torch exchange must still fit the driver's time limit)
        wd = min(watchdog_s, WATCHDOG_AFTER_SETUP_S) if len(set_up) == n else watchdog_s
        if time.monotonic() - last[0] > wd:
            rc, why = 3, f"no rank wrote anything for {wd:.0f} s" + (" after the tiles were set up" if len(set_up) == n else "")
            break
        time.sleep(poll_s)
    if rc:
        kill_all()
    for t in threads:
        t.join(timeout=5.0)
    if rc:
        err.write(f"[bench launcher] {why}; children killed.  stderr tails:\n")
        for r, tl in enumerate(tails):
            err.write(f"--- rank {r} ---\n" + "".join(tl))
        err.flush()
        try:          # the evidence of a failed attempt is kept as a file as well (the line of a second attempt names it)
            d = os.path.join(ROOT, "gpurun_out"); os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "bench_failed_attempt_stderr.txt"), "w") as f:
                f.write(f"{why}\n")
                for r, tl in enumerate(tails):
                    f.write(f"--- rank {r} ---\n" + "".join(tl))
        except OSError:
            pass
        return rc
    js = [l for l in lines0 if l.lstrip().startswith("{")]
    if len(js) != 1:
        err.write(f"[bench launcher] rank 0 printed {len(js)} JSON lines, expected 1\n"); err.flush()
        return 4
    out.write(js[0]); out.flush()
    return 0


def narrow_algorithmic_bytes(st):
    """B_narrow of SURVEY.md §8(d): both rings of every pair executed (2 coordinates x w per point)
    plus 7 scalars of both floes, and every contact row written (floe-floe rows twice: i and mirrored
    j; boundary rows once).  "Executed" = the pairs the narrow kernel is launched on: the broad phase
    culls the bounding-circle pairs whose ring boxes are disjoint (about a third), and those cost the
    narrow phase no byte."""
    return (2 * W * st["n_pair_ring_points"] + 2 * 7 * W * st["n_pairs_clipped"]
            + 7 * W * (2 * st["n_pair_rows"] + st["n_elem_rows"]))


def step_algorithmic_bytes(st):
    """B = B_broad + B_narrow + B_reduce + B_force + B_integ of SURVEY.md §8(d)."""
    M, N = st["M"] + st["n_ghosts"], st["N"]
    idx = 4
    b_broad = M * (3 * W + 2 * idx)
    b_reduce = N * (3 * W + idx)
    b_force = forcing_algorithmic_bytes(st)
    b_integ = N * (22 * W + 14 * W + 24 * W) + 4 * W * st["n_ring_points"]
    return b_broad + narrow_algorithmic_bytes(st) + b_reduce + b_force + b_integ


def measured_hbm_ceiling(torch, device):
    """On-device copy ceiling (SURVEY.md §8d asks for the fraction against a measured ceiling as well as the
    nominal 8 TB/s): a 1 GiB device-to-device copy timed with events, read + write bytes over the best of 5."""
    n = 1 << 27
    a = torch.empty(n, dtype=torch.float64, device=device); b = torch.empty_like(a)
    a.fill_(1.0); b.copy_(a); torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        best = max(best, 2 * 8 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def reference_cpu(cfg, cores, budget_s=60.0):
    """The reference package itself, if this machine has Julia with Subzero.jl installed (preferred over the
    port; the GPU box is not expected to have it).  bench/subzero_cpu.jl builds the same field through the package's
    public constructors and times timestep_sim!.  Returns a cpu_baseline dict of kind "reference" or None."""
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    julia = shutil.which("julia")
    uo_field = np.asarray(cfg["uo"], float)
    if (not julia or any(k != "periodic" for k in cfg["kinds"]) or cfg.get("topography") or np.ptp(uo_field) > 0
            or np.ptp(np.asarray(cfg["vo"], float)) > 0 or np.any(np.asarray(cfg["vo"], float) != 0)):
        return None          # the script covers the periodic, uniform-flow workload (BASELINE configs[1])
    try:
        probe = subprocess.run([julia, "-e", "using Subzero"], capture_output=True, timeout=600)
        if probe.returncode != 0:
            return None
        off, vx, vy = cfg["vert_off"], cfg["vx"], cfg["vy"]
        steps = 5
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            uo = float(uo_field.ravel()[0]); dgrid = cfg["L"] / cfg["Nx"]
            hmean = float(np.asarray(cfg["height"], float).ravel()[0])
            npc = max(1, int(round(dgrid / cfg["dg"])))          # SubGridPointsGenerator(grid, npc): sub-grid spacing dgrid / npc
            f.write(f"{cfg['n_floes']} {cfg['L']!r} {cfg['dt']} {dgrid!r} {hmean!r} {float(cfg['E'])!r} {uo!r} {npc}\n")
            for i in range(cfg["n_floes"]):
                o0, o1 = off[i], off[i + 1]
                pts = " ".join(f"{x!r} {y!r}" for x, y in zip(vx[o0:o1].tolist(), vy[o0:o1].tolist()))
                f.write(f"{o1 - o0} {float(cfg['u'][i])!r} {float(cfg['v'][i])!r} {float(cfg['xi'][i])!r} {pts}\n")
            path = f.name
        here = os.path.dirname(os.path.abspath(__file__))
        out = subprocess.run([julia, "-t", str(cores), os.path.join(here, "bench", "subzero_cpu.jl"), path, str(steps)],
                             capture_output=True, text=True, timeout=max(600.0, 10 * budget_s))
        os.unlink(path)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        if out.returncode != 0 or not line:
            return None
        kv = dict(t.split("=") for t in line[0].split()[1:])
        return {"value": float(kv["floe_steps_per_sec"]), "unit": "floe-steps/s", "cores": int(kv["threads"]), "kind": "reference",
                "sample": f"Subzero.jl (julia {kv['julia']}) timestep_sim! x {kv['steps']} on the same {cfg['n_floes']}-floe field, "
                          f"collisions + coupling every step, other processes off"}
    except Exception:
        return None


class RelaxationParity:
    """The HIP engine against the oracle THROUGH the relaxation -- up to the state the timed window starts from, not only on the unrelaxed field:
    a fresh context takes the same timesteps the oracle walks through (cpu_baseline) and the two are compared at checkpoints: pair list of
    the step (bit-exact), guard counters, and the state columns on their own scale.  The per-step agreement is what the suite holds (3 .. 10
    steps: 1e-9); a stiff contact network then amplifies the last-bit differences of a step (libm trig; the exactly summed totals against the
    reference's serial sums) -- the error curve `err_by_step` shows that growth from round-off level, which is what separates it from a defect
    (that would show at step 1).  Stated tolerance at the end of the 50 steps: 1e-4 of each column's scale, with the pair lists equal."""
    CHECK = (1, 2, 5, 10, 20, 30, 40, 50)
    COLS = ("cx", "cy", "alpha", "u", "v", "xi", "coll_fx", "coll_fy", "coll_trq", "fxOA", "fyOA", "trqOA", "overarea")

    def __init__(self, cfg, device):
        import subzero_jl_amd
        from subzero_jl_amd import fields
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        self.cfg = cfg; self.t = 0; self.out = {"tolerance": 1e-4, "err_by_step": {}, "pairs_equal_at": {}}
        try:
            self.hw = fields.build_world(subzero_jl_amd.World(device), cfg)
        except Exception as e:      # noqa: BLE001
            self.hw = None; self.out["error"] = str(e)[:300]

    def after_oracle_step(self, ow, k):          # the oracle has just finished step k - 1 (k steps in all)
        import numpy as np
        import parity
        if self.hw is None or k not in self.CHECK:
            return
        try:
            self.t += self.hw.run(k - self.t, self.t, self.cfg["dt"], coupling_dt=1)
            errs = {f: parity.relerr(self.hw.get(f), ow.get(f)) for f in self.COLS}
            hi, hj = self.hw.pairs(); oi, oj = ow.pairs()
            self.out["err_by_step"][str(k)] = max(errs.values())
            self.out["pairs_equal_at"][str(k)] = bool(len(hi) == len(oi) and np.array_equal(hi, oi) and np.array_equal(hj, oj))
            self.out.update(steps=k, steps_run=self.t, n_pairs=int(len(oi)), max_rel_state_err=max(errs.values()), worst_column=max(errs, key=errs.get),
                            guards_equal=bool(np.array_equal(self.hw.warn_counts(), ow.warn_counts())))
        except Exception as e:      # noqa: BLE001
            self.out["error"] = str(e)[:300]; self.hw = None

    def result(self):
        o = self.out
        o["ok"] = bool("error" not in o and o.get("steps_run") == o.get("steps") and all(o["pairs_equal_at"].values()) and o.get("max_rel_state_err", 1.0) <= o["tolerance"])
        return o


def cpu_baseline(cfg, budget_s=20.0, relax_steps=0, device=None):
    """The oracle (kind: port) on a bounded sample of the same workload: the same 10k-floe field,
    as many whole timesteps as fit the budget (at least 2), all host cores of this process."""
    from oracle import orc
    from subzero_jl_amd import fields
    cores = len(os.sched_getaffinity(0))
    try:     # the box gives one GPU's share of the host: honour the cgroup CPU quota
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(quota) // int(period)))
    except Exception:
        pass
    ref = reference_cpu(cfg, cores)
    if ref is not None:
        return ref
    w = fields.build_world(orc.World(), cfg)
    w.set_threads(cores)
    # the same relaxation as the measured run (untimed; it also pages everything in), so that both time the same kind of step
    rp = RelaxationParity(cfg, device) if device is not None else None
    tr = time.perf_counter(); tp = 0.0
    for k in range(max(relax_steps, 1)):
        w.timestep_sim(k, cfg["dt"], coupling_dt=1)
        if rp is not None:
            t2 = time.perf_counter(); rp.after_oracle_step(w, k + 1); tp += time.perf_counter() - t2
    tr = time.perf_counter() - tr - tp
    base = max(relax_steps, 1)
    relaxed_parity = rp.result() if rp is not None else None
    del rp
    w.phase_times()
    t0 = time.perf_counter(); steps = 0
    while steps < 2 or (time.perf_counter() - t0 < 0.75 * budget_s and steps < 300):
        w.timestep_sim(base + steps, cfg["dt"], coupling_dt=1); steps += 1
    el = time.perf_counter() - t0
    ph = w.phase_times()
    # the same on one core (the reference's default when Julia is started without -t)
    w.set_threads(1)
    t1 = time.perf_counter(); s1 = 0
    while s1 < 2 or (time.perf_counter() - t1 < 0.25 * budget_s and s1 < 100):
        w.timestep_sim(base + steps + s1, cfg["dt"], coupling_dt=1); s1 += 1
    el1 = time.perf_counter() - t1
    ph1 = w.phase_times()
    return {"value": cfg["n_floes"] * steps / el, "unit": "floe-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timesteps of the same {cfg['n_floes']}-floe field ({el:.1f} s) after {base} untimed relaxation steps ({tr:.1f} s), "
                      f"OpenMP over floes like the reference's Threads.@threads loops; then {s1} steps on one core ({el1:.1f} s)",
            "value_1core": cfg["n_floes"] * s1 / el1,
            "hip_against_this_oracle_after_the_relaxation": relaxed_parity,
            # where the port spends its time (ms per step): the phases the reference runs serially (Dict pass, mirror / ghost fold,
            # the forcing loop: collisions.jl:799-862, coupling.jl:1498) are why more cores buy little
            "ms_per_step_by_phase": {k: 1e3 * v / steps for k, v in ph.items()},
            "ms_per_step_by_phase_1core": {k: 1e3 * v / s1 for k, v in ph1.items()}}


def pmc_traffic(workload, n_floes, kernel):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (separate rocprofv3 --pmc runs of
    this same command, summarised by tools/pmc_summary.py into profiles/pmc_traffic.json with the gfx950 correction
    2 x FETCH_SIZE + WRITE_SIZE).  None -- with the reason -- when no committed pass covers this workload: the counters
    cannot be collected inside a timed run."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        table = json.load(open(path))
    except Exception:
        return None, "profiles/pmc_traffic.json not found"
    grp = table.get(f"{workload}:{n_floes}", {})
    ent = grp.get(kernel)
    if ent is None:
        return None, f"no committed PMC pass for {kernel} of {workload} at {n_floes} floes (tools/profile_round.sh makes one)"
    sha = (grp.get("_meta") or {}).get("kernel_source_sha16")
    if sha != kernel_source_sha16():
        return None, (f"the committed PMC pass was made on other kernel sources (sha {sha}, now {kernel_source_sha16()}): "
                      f"not quoted; tools/profile_round.sh makes a new one")
    return float(ent["hbm_bytes_per_launch"]), ent.get("source", path)


def step_traffic(workload, n_floes, narrow_kernel):
    """counter traffic of the step's launches from the same committed PMC passes: {launch: bytes} and their sum (None without a pass)"""
    try:
        ent = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(f"{workload}:{n_floes}", {})
    except Exception:
        return None
    if (ent.get("_meta") or {}).get("kernel_source_sha16") != kernel_source_sha16():
        return None          # (measured on other kernel sources: see roofline.traffic_source)
    pick = {}
    for k, v in ent.items():
        if k == "_meta":
            continue
        if k == narrow_kernel or k.startswith(("sz_k_neighbors", "sz_k_inter_fill", "sz_k_integrate<true", "sz_k_vel_search", "sz_k_elem_scan_fill", "sz_k_ghost_list")):
            pick[k] = float(v["hbm_bytes_per_launch"])
    return {"per_launch_bytes": pick, "bytes": sum(pick.values())} if pick else None


def kernel_source_sha16():
    """sha256 (first 16 hex digits) of the kernel sources: the committed PMC table is only quoted for the code it was measured on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "subzero.jl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hpp", ".hip")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def forcing_algorithmic_bytes(st):
    """B_force of SURVEY.md §8(d): two coordinates per sub-floe point, 8 columns read and 4 written per floe"""
    return 2 * W * st["n_sub_points"] + st["N"] * (8 * W + 4 * W)


PARITY_COLS = ["cx", "cy", "alpha", "u", "v", "xi", "height", "mass", "moment", "p_dxdt", "p_dydt", "p_dalphadt", "p_dudt", "p_dvdt",
               "p_dxidt", "fxOA", "fyOA", "trqOA", "overarea", "coll_fx", "coll_fy", "coll_trq", "stress_accum", "stress_instant", "strain"]


def oracle_parity(cfg, device, steps=2, coupling_dt=1):
    """Parity evidence for the very field the bench times (N = 1 line): add_ghosts! + timestep_collisions! of the HIP engine against the
    CPU oracle -- pair list bit-exact?, worst interaction-row element relative to tests/parity.py's per-element tolerance (rtol 1e-10 +
    the round-off floor of the reference's own shoelace sums) -- then `steps` resident timesteps against the oracle's timestep_sim!
    (largest max-norm relative error over the state columns).  The oracle is the checker here, never the thing measured."""
    import numpy as np
    import subzero_jl_amd
    from subzero_jl_amd import fields
    from oracle import orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    n = cfg["n_floes"]
    out = {"steps": steps, "n_floes": n}
    try:
        hw = fields.build_world(subzero_jl_amd.World(device), cfg); ow = fields.build_world(orc.World(), cfg)
        ow.set_threads(len(os.sched_getaffinity(0)))
        hw.add_ghosts(); ow.add_ghosts()
        out["ghosts_equal"] = bool(hw.M == ow.M and hw.ghosts() == ow.ghosts())
        hw.timestep_collisions(n, cfg["dt"]); ow.timestep_collisions(n, cfg["dt"])
        hi, hj = hw.pairs(); oi, oj = ow.pairs()
        out["n_pairs"] = int(len(oi))
        out["pairs_equal"] = bool(len(hi) == len(oi) and np.array_equal(hi, oi) and np.array_equal(hj, oj))
        hoff, hrows = hw.interactions(); ooff, orows = ow.interactions()
        out["n_rows"] = int(len(orows))
        if np.array_equal(hoff, ooff) and np.array_equal(hrows[:, 0], orows[:, 0]):
            fl = parity.force_floors(orows, float(np.max(ow.get("rmax"))))
            worst = 0.0; rel = 0.0
            for c, floor in ((1, fl["force"]), (2, fl["force"]), (5, fl["torque"]), (6, fl["area"]), (3, 1e-10 * fl["Lc"]), (4, 1e-10 * fl["Lc"])):
                r, _ = parity.worst_element(hrows[:, c], orows[:, c], 1e-10, floor); worst = max(worst, r)
                with np.errstate(divide="ignore", invalid="ignore"):
                    e = np.abs(hrows[:, c] - orows[:, c]) / np.maximum(np.abs(orows[:, c]), floor if floor > 0 else 1e-300)
                rel = max(rel, float(np.nanmax(e)) if len(e) else 0.0)
            out["rows_equal_structure"] = True
            out["max_rel_row_err"] = rel                      # |a - b| / max(|b|, floor) over all row elements
            out["rows_within_1e-10"] = bool(worst <= 1.0)     # the suite's criterion: |a - b| <= 1e-10 |b| + floor
        else:
            out["rows_equal_structure"] = False
        del hw, ow
        hw = fields.build_world(subzero_jl_amd.World(device), cfg); ow = fields.build_world(orc.World(), cfg)
        ow.set_threads(len(os.sched_getaffinity(0)))
        hw.run(steps, 0, cfg["dt"], coupling_dt=coupling_dt)
        for t in range(steps):
            ow.timestep_sim(t, cfg["dt"], coupling_dt=coupling_dt)
        errs = {f: parity.relerr(hw.get(f), ow.get(f)) for f in ("cx", "cy", "alpha", "u", "v", "xi", "coll_fx", "coll_fy", "coll_trq", "fxOA", "fyOA", "trqOA", "sa11", "sa22", "e11")}
        out["max_rel_state_err_after_steps"] = max(errs.values())
        out["state_within_1e-9"] = bool(max(errs.values()) <= 1e-9)
        out["guards_equal"] = bool(np.array_equal(hw.warn_counts(), ow.warn_counts()))
        out["ok"] = bool(out["ghosts_equal"] and out["pairs_equal"] and out.get("rows_within_1e-10", False) and out["state_within_1e-9"])
    except Exception as e:      # noqa: BLE001  (a line with the reason beats no line)
        out["ok"] = False; out["error"] = str(e)[:300]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=10, help="the block of --steps steps is timed this many times; value and ms_per_step "
                                                            "are the median block (min / max reported beside it)")
    ap.add_argument("--floes", type=int, default=0, help="floes per GPU (weak scaling of the configs[1] field); default: 10000 on one "
                                                         "GPU, and for N > 1 the metric's strong-scaling workload instead")
    ap.add_argument("--total-floes", type=int, default=0, help="fix the job size (strong scaling)")
    ap.add_argument("--workload", default=None, choices=["configs1", "configs2", "configs3", "configs4", "walls", "voronoi"],
                    help="BASELINE.json configs[k]: 1 = periodic box + uniform flow (the metric's 1-GPU config, default for N = 1); 2 = "
                         "100k floes converge/diverge flow (the metric's multi-GPU config, default for N > 1); 3 = four collision "
                         "walls + topography, strait flow; 4 = 25 %% concentration")
    ap.add_argument("--two-way", action="store_true", help="two-way coupling on (ice-on-ocean stress per centre cell; "
                                                           "off in the metric's config, as in CouplingSettings())")
    ap.add_argument("--precision", default="f64", choices=["f64", "mixed"],
                    help="mixed: per-point forcing arithmetic in fp32 (BASELINE configs[4]); the metric's config is f64")
    ap.add_argument("--coupling-dt", type=int, default=1, help="couple every k-th step (reference default: 10)")
    ap.add_argument("--relax-steps", type=int, default=50, help="untimed steps before the warm-up: the fields are generated on a jittered lattice "
                    "with overlapping neighbours and then relaxed, as SURVEY.md §8(d) specifies for the synthetic configurations (50 steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (cpu_baseline and the oracle parity of the timed field)")
    ap.add_argument("--rehearse-shared-gpu", action="store_true", help="N > 1 ranks that all use GPU 0, with gloo between them and the library's exchange "
                    "over the host channel (sz_comm_init_host): runs the whole multi-rank flow of this script on a one-GPU box; the "
                    "line it prints says so and is NOT a measurement")
    ap.add_argument("--force-tiled", action="store_true", help="run the halo/RCCL path even with one rank")
    ap.add_argument("--watchdog", type=float, default=300.0, help="self-launched N > 1 runs: seconds without a line from ANY rank after which the "
                    "parent kills the ranks and exits non-zero")
    ap.add_argument("--no-strong-reference", action="store_true", help="N = 1: skip the extra leg that times the multi-GPU workload (configs[2], 100 000 floes) "
                    "in one context on this GPU -- the denominator of the strong-scaling curve, which the N > 1 lines are measured on")
    ap.add_argument("--parity-steps", type=int, default=3, help="tiled runs: steps after which the owned columns of all ranks are compared with a "
                    "single-context run of the same field on rank 0 (tiled_parity in the line); 0: skip")
    args = ap.parse_args()

    # `python3 bench.py --gpus N` with no launcher around it: this process becomes the launcher -- N fresh children, one per GPU, before
    # torch is imported or HIP touched (the children see WORLD_SIZE and run the code below; `python -m torch.distributed.run ... bench.py`
    # sets it too, so the wrapped form takes the same path)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        child = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
        rc = launch_ranks(args.gpus, child, watchdog_s=args.watchdog, shared_gpu=args.rehearse_shared_gpu)
        if rc != 0 and "SZ_TILES_BACKEND" not in os.environ and not args.rehearse_shared_gpu:
            # The library's own RCCL exchange with real peers could not be exercised on the one-GPU build boxes.  Should it fail or
            # hang on the first multi-GPU run, ONE more attempt is made with fresh processes and the torch.distributed exchange
            # (all_to_all_single on the buffers the library packs); the line then says so.
            sys.stderr.write(f"[bench launcher] first attempt ended with code {rc}; one more with SZ_TILES_BACKEND=torch\n"); sys.stderr.flush()
            rc = launch_ranks(args.gpus, child, watchdog_s=args.watchdog, shared_gpu=False,
                              extra_env={"SZ_TILES_BACKEND": "torch", "SZ_BENCH_PRIMARY_RC": str(rc), "SZ_BENCH_NOTE": f"the run with the library's RCCL exchange ended with launcher code {rc}; "
                                                                                        "this line is the second attempt (torch.distributed all_to_all_single)"})
        sys.exit(rc)

    # RCCL prints a version banner on stdout; the contract is ONE JSON line there.  Everything the
    # libraries write to fd 1 goes to stderr, the JSON line is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import subzero_jl_amd
    from subzero_jl_amd import fields

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    if args.rehearse_shared_gpu:
        if world > 5:
            raise SystemExit("--rehearse-shared-gpu: at most 5 ranks (the build boxes allow 6 processes on a GPU, and the launcher is one of them)")
        local = 0
    cdev = "cpu" if args.rehearse_shared_gpu else "cuda"          # where the script's own small collectives live
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_tiled:
        import datetime
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29533"
        # every rank builds the 100 k-floe field on the host first (tens of seconds), rank 0 also runs the one-GPU reference: generous timeouts
        if args.rehearse_shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=datetime.timedelta(minutes=30))
    log(f"start: world {world}, device {local}")

    # BASELINE.json: "floe-steps/sec at 10k/100k floes, 1/2/4/8 MI355X" -- one GPU: configs[1] (10 000 floes, uniform flow);
    # several GPUs: the strong-scaling workload, configs[2] (100 000 floes, converge/diverge flow) cut into spatial tiles.
    # --floes asks for weak scaling of the configs[1] field instead.
    if args.total_floes > 0:
        n_total, scaling = args.total_floes, "strong"
    elif args.floes > 0:
        n_total, scaling = args.floes * world, "weak"
    elif world > 1:
        n_total, scaling = 100000, "strong"
    else:
        n_total, scaling = 10000, "strong"          # (one GPU, one fixed job: the N = 1 point of either curve)
    workload = args.workload or ("configs2" if (world > 1 and args.floes == 0) else "configs1")
    wl = {"configs1": dict(seed=12345), "configs2": dict(seed=12346, ocean="converge_diverge"),
          "configs3": dict(seed=12345, walls=True, topography=True, ocean="strait"),
          "configs4": dict(seed=12347, concentration=0.25), "walls": dict(seed=12345, walls=True),
          # the reference's own kind of field (initialize_floe_field): touching Voronoi cells, a size spectrum -> the larger neighbour capacity
          "voronoi": dict(seed=12345, shape="voronoi", ocean="shear")}[workload]
    cfg = fields.make_config(n_floes=n_total, **wl)
    log(f"field built: {workload}, {n_total} floes")
    coupling_dt = args.coupling_dt
    tiled = not (world == 1 and not args.force_tiled)
    tiled_parity = None
    ref = None                     # rank 0 of a tiled run: the same field in ONE context (parity check, then the one-GPU reference point)
    ref_steps = 0                  # steps that context has behind it
    steps_run = 0                  # steps every context has behind it when the warm-up starts
    if not tiled:
        hw = fields.build_world(subzero_jl_amd.World(local), cfg)
        if args.two_way:
            hw.set_two_way(True, dt=cfg["dt"]); hw.set_temps(0.0, -10.0)
        hw.set_precision(args.precision)
        # the batches run through even if a floe gets tagged (the host's simplify_floes! is not part of the timed path);
        # the tag counts at the end of the timed window are printed, so the reader sees whether that ever mattered
        runner = lambda n, t0: hw.run(n, t0, cfg["dt"], coupling_dt=coupling_dt, stop_on_tags=False)
    else:
        from subzero_jl_amd import tiles
        # the halo exchange runs inside the library (RCCL bound by libsubzero_hip.so: grouped send / receive with the
        # neighbouring tiles); SZ_TILES_BACKEND=torch: one torch.distributed all_to_all_single per step instead
        backend = os.environ.get("SZ_TILES_BACKEND", "library-host" if args.rehearse_shared_gpu else "library")
        backend_note = None

        def make_tiles(be):
            t = tiles.TiledWorld(cfg, rank, world, local, dist, always_exchange=args.force_tiled, backend=be, host_staging=args.rehearse_shared_gpu, rebox_every=150)
            t.repartition_every = 10 ** 9     # floes drift metres per step against tiles of hundreds of km: no re-tiling inside a bench run
            return t
        # The library-side exchange binds RCCL at run time.  TiledWorld agrees on a failed SET-UP between the ranks before any
        # mismatched collective (tiles.py: every rank raises TileSetupError together); then ALL ranks fall back to the
        # torch.distributed exchange -- a line with a note beats no line.  (A failure inside a RUNNING exchange cannot be agreed
        # on: sz_tile_run returns the same error code on every rank at the end of the batch, and that is fatal here.)
        try:
            tw = make_tiles(backend)
        except tiles.TileSetupError as e:
            if backend != "library":
                raise
            backend_note = f"library exchange unavailable ({str(e)[:200]}); torch.distributed all_to_all_single instead"
            log(backend_note)
            backend = "torch"
            tw = make_tiles(backend)
        hw = tw.world
        if args.precision == "mixed":
            hw.set_precision("mixed")
        runner = lambda n, t0: tw.run(n, t0, cfg["dt"], coupling_dt=coupling_dt)
        log(f"tile set up: {len(tw.gidx)} owned floes, exchange = {backend}")
        # ---- tiled result against the single context, at config scale: k steps of the timed field, owned columns of every rank
        # gathered on rank 0 and compared bit for bit with ONE context stepping the whole field there
        k = max(0, min(args.parity_steps, args.relax_steps if args.relax_steps > 0 else args.parity_steps))
        if k > 0 and dist is not None:
            runner(k, 0); steps_run = k
            mine = {"gidx": tw.gidx.copy()}
            for f in PARITY_COLS:
                mine[f] = hw.get(f)[:len(tw.gidx)]
            parts = [None] * world if rank == 0 else None
            dist.gather_object(mine, parts, dst=0)
            if rank == 0:
                tiled_parity = {"steps": k, "ranks": world, "columns": len(PARITY_COLS)}
                try:
                    ref = fields.build_world(subzero_jl_amd.World(local), cfg)
                    ref.set_precision(args.precision)
                    ref.run(k, 0, cfg["dt"], coupling_dt=coupling_dt, stop_on_tags=False); ref_steps = k
                    g = np.concatenate([p_["gidx"] for p_ in parts])
                    bad = {}
                    worst = 0.0
                    for f in PARITY_COLS:
                        a = np.concatenate([p_[f] for p_ in parts]); b = ref.get(f)[g]
                        if not np.array_equal(a, b):
                            sc = max(float(np.max(np.abs(b))), 1e-300)
                            bad[f] = float(np.max(np.abs(a - b)) / sc); worst = max(worst, bad[f])
                    tiled_parity.update(floes_compared=int(len(g)), all_floes_covered=bool(len(g) == cfg["n_floes"] and len(np.unique(g)) == len(g)),
                                        bit_equal=not bad, columns_differing=bad, max_rel_diff=worst,
                                        ok=bool(not bad and len(g) == cfg["n_floes"]))
                except Exception as e:      # noqa: BLE001
                    tiled_parity.update(ok=False, error=str(e)[:300])
                log(f"tiled parity after {k} steps: {tiled_parity}")
            del mine, parts

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- the oracle's word on the very field this run times (N = 1): before the timing, on fresh contexts
    parity_line = None
    if not tiled and not args.no_cpu_baseline and rank == 0 and not args.two_way and args.precision == "f64":
        parity_line = oracle_parity(cfg, local, steps=2, coupling_dt=coupling_dt)
        log(f"oracle parity: {parity_line}")

    if args.relax_steps > steps_run:
        runner(args.relax_steps - steps_run, steps_run)          # part of the workload's definition, not of the measurement
    t_relaxed = max(args.relax_steps, steps_run)
    runner(args.warmup, t_relaxed)
    log("relaxed and warm")
    # Inside the timed region the dominant kernel is bracketed by HIP events (one pair per step, on the stream it is launched on) --
    # in ONE of the timed blocks, the middle one (its launches are typical of the window: the narrow phase gets cheaper as the contact network
    # relaxes): an event pair makes the kernel before and the kernel after it wait ~6 us each (kernel
    # trace: consecutive launches are otherwise back to back), 10 % of a 0.12 ms step, and that is the instrument's cost, not the
    # workload's.  `value` is the median block; the event-timed block is printed beside it (`ms_per_step_event_timed_block`).
    nrep = max(1, args.repeats)
    kt = st_ev = forcing_where = None
    blocks = []
    tstep = t_relaxed + args.warmup
    for rep in range(nrep):
        if rep == nrep // 2:
            hw.profile(True, only="narrow")          # also clears the cumulative narrow-phase work counters: they count the launches the events time
        elif rep == nrep // 2 + 1:
            kt = hw.kernel_times(); st_ev = hw.stats(); forcing_where = hw.forcing_launch()
            hw.profile(False)
        barrier()
        t0 = time.perf_counter()
        runner(args.steps, tstep)
        barrier()
        el = time.perf_counter() - t0
        tstep += args.steps
        if dist is not None:
            t = torch.tensor([el], device=cdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        blocks.append(el)
        log(f"block {rep}: {1e3 * el / args.steps:.4f} ms/step")
    el = float(np.median(blocks))
    if kt is None:                            # (the event-timed block was the last one)
        kt = hw.kernel_times(); st_ev = hw.stats(); forcing_where = hw.forcing_launch()
    st = hw.stats()                           # counts of the last step (st_ev: the cumulative work counters of the event-timed block)
    kernel_name = hw.narrow_kernel_name()
    ceiling = measured_hbm_ceiling(torch, torch.device("cuda", local)) if rank == 0 else 0.0
    # per-class breakdown from a second, untimed pass with every class event-timed
    nb = max(1, min(args.steps, 50))
    hw.profile(True)
    runner(nb, tstep)
    torch.cuda.synchronize()
    kt_all = hw.kernel_times()
    hw.profile(False)
    # strong scaling: the same workload on ONE GPU (rank 0, the context of the parity check beside its tile) through the SAME regime --
    # the same relaxation and warm-up, the same step window, blocks of the same length, median -- so that the line carries its own
    # reference point; the other ranks wait at the barrier
    one_gpu = None
    if tiled and world > 1:
        if rank == 0:
            try:
                if ref is None:
                    ref = fields.build_world(subzero_jl_amd.World(local), cfg); ref.set_precision(args.precision); ref_steps = 0
                rr = lambda n, t0: ref.run(n, t0, cfg["dt"], coupling_dt=coupling_dt, stop_on_tags=False)
                if t_relaxed > ref_steps:
                    rr(t_relaxed - ref_steps, ref_steps)
                rr(args.warmup, t_relaxed)
                rb = []; ts = t_relaxed + args.warmup
                for rep in range(nrep):
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    rr(args.steps, ts)
                    torch.cuda.synchronize(); rb.append(time.perf_counter() - t1); ts += args.steps
                one_gpu = {"ms_per_step": 1e3 * float(np.median(rb)) / args.steps, "ms_per_step_min": 1e3 * min(rb) / args.steps,
                           "ms_per_step_max": 1e3 * max(rb) / args.steps, "steps": args.steps, "repeats": nrep,
                           "note": f"same field in one context on rank 0's GPU, same {t_relaxed} relaxation + {args.warmup} warm-up steps, the same step window "
                                   f"({t_relaxed + args.warmup}..{ts}), median of {nrep} blocks"}
                del ref
            except Exception as e:      # noqa: BLE001
                one_gpu = {"error": str(e)[:200]}
            log(f"one-GPU reference: {one_gpu}")
        barrier()

    if rank == 0:
        n_ms, n_launch = kt["narrow"]
        narrow_ms = n_ms / max(n_launch, 1)
        # algorithmic bytes of the SAME launches the event time averages: cumulative device counters over the window
        nl = max(st_ev["acc_narrow_launches"], 1)
        win = {"n_pair_ring_points": st_ev["acc_pair_ring_points"] / nl, "n_pairs_clipped": st_ev["acc_pair_items"] / nl,
               "n_pair_rows": st_ev["acc_pair_rows"] / nl, "n_elem_rows": st_ev["acc_elem_rows"] / nl}
        dirchk = {"per_launch": st_ev["acc_dir_checks"] / nl, "certified_fraction": st_ev["acc_dir_checks_certified"] / max(st_ev["acc_dir_checks"], 1),
                  "note": "direction checks of calc_normal_force (collisions.jl:58-68); certified = settled from the crossing detection of the translated polygon alone, no second clip"}
        b_narrow = narrow_algorithmic_bytes(win)
        # small fields: the step's forcings ride in the narrow launch (its tail) -- the launch the events bracket then does both
        rides = forcing_where == 2 and coupling_dt == 1
        # pipelined steps (two launches per timestep, csrc/sz_pipeline.hpp): the launch also moves the rings for the NEXT step (GEO) -- its share
        # of B_integ of SURVEY.md section 8(d): every ring point read and written (4 w per point), 8 columns read and the 6 geometry words of
        # the floe's collision record written per floe
        pipelined = bool(hw.pipelined()) if hasattr(hw, "pipelined") else False
        b_geo = (4 * W * st["n_ring_points"] + st["N"] * 14 * W) if pipelined else 0
        b_launch = b_narrow + (forcing_algorithmic_bytes(st) if rides else 0) + b_geo
        achieved = b_launch / (narrow_ms * 1e-3) / 1e9 if narrow_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(workload, cfg["n_floes"], kernel_name) if not tiled else (None, "tiled run")
        step_bytes = step_algorithmic_bytes({**st, **win})
        out = {
            "metric": "floe_steps_per_sec", "value": cfg["n_floes"] * args.steps / el, "unit": "floe-steps/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64" if args.precision == "f64" else "f64 (forcings: fp32 per point)", "data": "synthetic",
            "repeats": len(blocks), "ms_per_step_min": 1e3 * min(blocks) / args.steps, "ms_per_step_max": 1e3 * max(blocks) / args.steps,
            "ms_per_step_event_timed_block": 1e3 * blocks[len(blocks) // 2] / args.steps,
            "value_note": f"median of {len(blocks)} timed blocks of {args.steps} steps each (barrier + device sync on both sides of every block)",
            "config": {"workload": (f"configs[1]: {cfg['n_floes']} random-polygon floes (8-16 verts), doubly periodic box {cfg['L'] / 1e3:.0f} km, "
                                    f"uniform_flow ocean 0.1 m/s" if workload == "configs1" else
                                    f"configs[2]: {cfg['n_floes']} random-polygon floes, doubly periodic box {cfg['L'] / 1e3:.0f} km, converge_diverge_flow ocean"
                                    if workload == "configs2" else
                                    f"{'configs[' + workload[-1] + ']-style' if workload[-1].isdigit() else workload} field ({wl}): {cfg['n_floes']} floes, box {cfg['L'] / 1e3:.0f} km, boundaries {cfg['kinds'][0]}, "
                                    f"{len(cfg['topography'])} topography elements") +
                                   f", {t_relaxed} relaxation steps after generation (since round 2; round 1 timed the unrelaxed field)" +
                                   f"; collisions + one-way coupling every {coupling_dt} step(s) + rigid-body update, dt={cfg['dt']} s" +
                                   (f"; {world} spatial tiles, one-deep ghost-floe halo per step" if tiled else ""),
                       "n_floes": cfg["n_floes"], "seed": cfg["seed"], "coupling_dt": coupling_dt, "two_way_coupling": bool(args.two_way),
                       "tiles": world if tiled else 1, "timed_step_window": [t_relaxed + args.warmup, tstep]},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved,
                         "kernel_note": (("narrow phase + the step's forcings in one launch (the forcings run in the tail of the narrow phase's single round): "
                                          "algorithmic bytes = B_narrow + B_force") if rides else "narrow phase") +
                                        (" + the ring move, boxes, records, cells and periodic ghosts of the NEXT step (GEO, pipelined steps: + B_geo = 4w per ring point + 14w per floe)" if pipelined else ""),
                         "pipelined_steps": pipelined,
                         "launches_per_step": 2 if pipelined else 3,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "measured_copy_ceiling": ceiling, "frac_of_measured_ceiling": achieved / ceiling if ceiling > 0 else None,
                         "kernel_ms": narrow_ms, "kernel_launches_timed": n_launch, "algorithmic_bytes_per_launch": b_launch,
                         "algorithmic_bytes_narrow_only": b_narrow, "algorithmic_bytes_geo": b_geo,
                         "counts_per_launch": win, "direction_checks": dirchk, "counts_note": "device counters accumulated over the launches the event time averages (the middle timed block)",
                         "step_algorithmic_bytes": step_bytes,
                         "step_frac": step_bytes / (el / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "step_traffic": step_traffic(workload, cfg["n_floes"], kernel_name) if not tiled else None},
            "kernel_ms_per_step": {k: (v[0] / nb) for k, v in kt_all.items()},
            "kernel_ms_per_step_note": f"separate untimed pass of {nb} steps with all kernel classes event-timed",
            "counts": {k: st[k] for k in ("M", "N", "n_ghosts", "n_pairs", "n_pairs_clipped", "n_pair_rows", "n_elem_rows",
                                           "n_inter_rows", "n_ring_points", "n_sub_points", "n_trace_fail", "n_retry")},
            "tags_at_end_of_timed_window": {"n_status_remove": st["n_status_remove"], "n_status_fuse": st["n_status_fuse"],
                                            "note": "floes the reference's simplify_floes! would have acted on during the run (0 = the timed steps are the reference's trajectory)"},
        }
        if parity_line is not None:
            out["parity"] = parity_line
        if tiled_parity is not None:
            out["tiled_parity"] = tiled_parity
        if one_gpu is not None:
            out["one_gpu_same_workload"] = one_gpu
            if "ms_per_step" in one_gpu:
                out["speedup_vs_one_gpu_same_workload"] = one_gpu["ms_per_step"] / (1e3 * el / args.steps)
        if tiled:
            out["config"]["halo_exchange"] = backend if world > 1 or args.force_tiled else None
            if backend_note:
                out["config"]["halo_exchange_note"] = backend_note
        if os.environ.get("SZ_BENCH_NOTE"):
            out["note"] = os.environ["SZ_BENCH_NOTE"]
        if os.environ.get("SZ_BENCH_PRIMARY_RC"):
            # a degraded line must say so to tools, not only to readers: the library's own RCCL exchange -- the default multi-GPU path -- failed
            # or hung (launcher code: 1 a rank failed, 3 watchdog), and this line was measured on the torch.distributed exchange instead
            out["primary_exchange_failed"] = int(os.environ["SZ_BENCH_PRIMARY_RC"])
            out["primary_exchange_stderr"] = "gpurun_out/bench_failed_attempt_stderr.txt"
        if tiled:
            out["exchange"] = {"backend": backend, "headers": "neighbouring tiles only (SZ_TILE_HEADERS=neighbours: measurement arm, no tag stop)" if os.environ.get("SZ_TILE_HEADERS") == "neighbours" else "every ordered pair of ranks (stop / pause words)",
                               "ms_per_step_event_timed": (kt_all.get("exchange", (0.0, 0))[0] / nb) if "exchange" in kt_all else None,
                               "note": "HIP events around the grouped send / receive on the library's communication stream, in the separate pass that event-times every class"}
        if args.rehearse_shared_gpu:
            out["data"] = "REHEARSAL: all ranks share GPU 0, transfers over gloo through the host -- not a measurement"
        if world == 1 and not tiled and workload == "configs1" and coupling_dt == 1 and not args.two_way and not args.no_strong_reference:
            # The line's `value` couples the floes to ocean and atmosphere in EVERY step (coupling_dt 1: the heaviest setting, and what rounds 1-3
            # timed).  The reference's own default is CouplingSettings(Δt = 10) (process_settings.jl:134-135): the same field, the same steps, with
            # the forcings evaluated every tenth step -- what a user who switches over with default settings would see.  Reported beside `value`,
            # never instead of it.
            try:
                log("secondary leg: the reference's default coupling interval (every 10th step) ...")
                w10 = fields.build_world(subzero_jl_amd.World(local), cfg); w10.set_precision(args.precision)
                r10 = lambda n, t0: w10.run(n, t0, cfg["dt"], coupling_dt=10, stop_on_tags=False)
                r10(args.relax_steps, 0); r10(args.warmup, args.relax_steps)
                b10 = []; ts = args.relax_steps + args.warmup
                for rep in range(nrep):
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    r10(args.steps, ts)
                    torch.cuda.synchronize(); b10.append(time.perf_counter() - t1); ts += args.steps
                m10 = float(np.median(b10))
                out["reference_default_coupling_interval"] = {
                    "coupling_dt": 10, "value": cfg["n_floes"] * args.steps / m10, "unit": "floe-steps/s", "ms_per_step": 1e3 * m10 / args.steps,
                    "ms_per_step_min": 1e3 * min(b10) / args.steps, "steps": args.steps, "repeats": nrep,
                    "note": "same field and steps as `value`, forcings every 10th step (the reference's CouplingSettings default, process_settings.jl:134-135); `value` couples every step"}
                del w10
            except Exception as e:      # noqa: BLE001
                out["reference_default_coupling_interval"] = {"error": str(e)[:200]}
        if world == 1 and not tiled and workload == "configs1" and args.floes == 0 and args.total_floes == 0 and not args.no_strong_reference:
            # the N > 1 lines of this script time configs[2] (100 000 floes, strong scaling): the same field in ONE context on this GPU, through
            # the same regime, is the N = 1 point of THAT curve (this line's `value` is the metric's one-GPU configuration, configs[1])
            try:
                log("strong-scaling reference: configs[2] on this GPU ...")
                del hw
                cfg2 = fields.make_config(n_floes=100000, seed=12346, ocean="converge_diverge")
                w2 = fields.build_world(subzero_jl_amd.World(local), cfg2); w2.set_precision(args.precision)
                r2 = lambda n, t0: w2.run(n, t0, cfg2["dt"], coupling_dt=coupling_dt, stop_on_tags=False)
                r2(args.relax_steps, 0); r2(args.warmup, args.relax_steps)
                b2 = []; ts = args.relax_steps + args.warmup
                for rep in range(nrep):
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    r2(args.steps, ts)
                    torch.cuda.synchronize(); b2.append(time.perf_counter() - t1); ts += args.steps
                m2 = float(np.median(b2))
                out["strong_scaling_workload_on_one_gpu"] = {
                    "workload": f"configs[2]: 100000 random-polygon floes, doubly periodic box {cfg2['L'] / 1e3:.0f} km, converge_diverge_flow ocean (what --gpus N > 1 times, cut into tiles)",
                    "value": 100000 * args.steps / m2, "unit": "floe-steps/s", "ms_per_step": 1e3 * m2 / args.steps,
                    "ms_per_step_min": 1e3 * min(b2) / args.steps, "ms_per_step_max": 1e3 * max(b2) / args.steps, "steps": args.steps, "repeats": nrep,
                    "note": "the denominator for strong scaling 1 -> N GPUs at 100k floes; not this line's `value`"}
                del w2, cfg2
            except Exception as e:      # noqa: BLE001
                out["strong_scaling_workload_on_one_gpu"] = {"error": str(e)[:200]}
        if not args.no_cpu_baseline and world == 1:
            log("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline(cfg, relax_steps=args.relax_steps, device=local)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()

if __name__ == "__main__":
    main()
