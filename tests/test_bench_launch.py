"""bench.py's self-launcher (`python3 bench.py --gpus N` without torch.distributed.run around it): N fresh children with the rank
environment, ONE JSON line from rank 0 passed through, a failing rank ends the job with a non-zero code, a silent job is killed
by the watchdog.  The children here are stand-in scripts: no GPU, no torch."""
import io
import json
import os
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _child(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_children_get_the_rank_environment_and_rank0_prints_the_line(tmp_path):
    cmd = _child(tmp_path, """
        import json, os, sys
        r, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        sys.stderr.write(f"hello from {r}\\n")
        open(os.path.join(os.environ["OUT_DIR"], f"env{r}.json"), "w").write(json.dumps({k: os.environ.get(k) for k in
            ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}))
        print("RCCL banner that must not reach stdout of the job")
        print(json.dumps({"metric": "m", "rank": r, "world": n}))
    """)
    out, err = io.StringIO(), io.StringIO()
    rc = bench.launch_ranks(3, cmd, watchdog_s=30, extra_env={"OUT_DIR": str(tmp_path)}, out=out, err=err)
    assert rc == 0, err.getvalue()
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "rank": 0, "world": 3}      # rank 0's line, only the JSON one
    envs = [json.load(open(tmp_path / f"env{r}.json")) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) > 0
    assert "[rank 2] hello from 2" in err.getvalue()
    # ranks sharing GPU 0 (--rehearse-shared-gpu): every LOCAL_RANK is 0
    rc = bench.launch_ranks(2, cmd, watchdog_s=30, shared_gpu=True, extra_env={"OUT_DIR": str(tmp_path)}, out=io.StringIO(), err=io.StringIO())
    assert rc == 0 and [json.load(open(tmp_path / f"env{r}.json"))["LOCAL_RANK"] for r in range(2)] == ["0", "0"]


def test_a_failing_rank_ends_the_job(tmp_path):
    cmd = _child(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.stderr.write("rank 1 gives up\\n"); sys.exit(7)
        for _ in range(600):
            sys.stderr.write("waiting for my peer\\n"); sys.stderr.flush(); time.sleep(0.1)
    """)
    out, err = io.StringIO(), io.StringIO()
    rc = bench.launch_ranks(2, cmd, watchdog_s=60, out=out, err=err)
    assert rc == 1 and out.getvalue() == ""
    assert "rank 1 exited with code 7" in err.getvalue() and "rank 1 gives up" in err.getvalue()


def test_the_watchdog_kills_a_silent_job(tmp_path):
    cmd = _child(tmp_path, """
        import time
        time.sleep(600)
    """)
    out, err = io.StringIO(), io.StringIO()
    rc = bench.launch_ranks(2, cmd, watchdog_s=1.5, out=out, err=err)
    assert rc == 3 and out.getvalue() == "" and "no rank wrote anything" in err.getvalue()


def test_no_line_is_an_error(tmp_path):
    cmd = _child(tmp_path, "print('no json here')\n")
    assert bench.launch_ranks(2, cmd, watchdog_s=30, out=io.StringIO(), err=io.StringIO()) == 4
