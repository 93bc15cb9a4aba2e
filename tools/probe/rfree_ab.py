"""Pipelined resident steps (default) against the three-launch steps (SZ_PIPELINE=0; SZ_REDUCE_FREE=0 in the environment adds the in-step reduce), bit for bit:
state columns, interaction rows, ghost statistics.  Run on the GPU box:  python tools/probe/rfree_ab.py [n] [fast]"""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def child(n, fast, batches):
    import subzero_jl_amd
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=n, seed=21, concentration=0.8)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    if fast:
        rng = np.random.default_rng(3)
        hw.set("u", rng.uniform(-40.0, 40.0, cfg["n_floes"])); hw.set("v", rng.uniform(-40.0, 40.0, cfg["n_floes"]))
    t = 0; out = {}
    for k in batches:
        done = hw.run(k, t, cfg["dt"], coupling_dt=1, stop_on_tags=False); t += done
        out.setdefault("pipelined", []).append(int(hw.pipelined()))
    out["stats"] = {k: int(v) for k, v in hw.stats().items() if k in ("n_ghosts", "n_status_fuse", "n_inter_rows", "M", "N")}
    off, rows = hw.interactions()
    np.savez(os.environ["RF_OUT"], off=off, rows=rows, **{f: hw.get(f) for f in ("cx", "cy", "u", "v", "xi", "alpha", "coll_fx", "coll_trq", "overarea", "si11", "si12", "e11", "e22", "fxOA", "height")}, vx=hw.rings()[1], vy=hw.rings()[2])
    print(json.dumps(out))

if __name__ == "__main__":
    if os.environ.get("RF_CHILD"):
        child(int(sys.argv[1]), sys.argv[2] == "1", [int(x) for x in sys.argv[3].split(",")])
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    fast = "1" if len(sys.argv) > 2 and sys.argv[2] == "fast" else "0"
    batches = sys.argv[3] if len(sys.argv) > 3 else "1,7,3,1,12"
    res = {}
    for tag, env in (("rfree", {}), ("instep", {"SZ_PIPELINE": "0"})):
        e = dict(os.environ, RF_CHILD="1", RF_OUT=f"/tmp/rf_{tag}.npz", **env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), fast, batches], env=e, capture_output=True, text=True)
        print(tag, p.stdout.strip()[-300:], p.stderr.strip()[-500:])
        res[tag] = np.load(f"/tmp/rf_{tag}.npz")
    a, b = res["rfree"], res["instep"]
    for f in a.files:
        if f in ("off", "rows"): continue
        d = np.max(np.abs(a[f] - b[f])) if a[f].shape == b[f].shape else "shape"
        print(f"{f:10s} max |rfree - instep| = {d}")
    print("offsets equal:", np.array_equal(a["off"], b["off"]), "rows", a["rows"].shape, b["rows"].shape)
    if a["rows"].shape == b["rows"].shape:
        d = np.abs(a["rows"] - b["rows"]); k = np.unravel_index(np.argmax(d), d.shape)
        print("rows max diff", d.max(), "at", k, a["rows"][k[0]], b["rows"][k[0]])
        print("per column:", d.max(axis=0))
