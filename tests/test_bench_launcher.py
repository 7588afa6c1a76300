"""bench.py --gpus N started as a plain `python bench.py` (no WORLD_SIZE) must start N ranks itself -- as a fresh child under
torch.distributed.run, before anything touched a GPU -- relay rank 0's line and leave with the child's exit code; a rank that never joins a
collective must end the run with a non-zero exit code (watchdog), not hang it.  Rehearsed on the CPU over gloo (--launch-only)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    return p, time.time() - t0


def test_launcher_spawns_n_ranks():
    p, _ = run(["--gpus", "2", "--launch-only"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, p.stdout                                  # ONE line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_env"] == 2 and out["launched"] is True


def test_launcher_three_ranks_and_n1_via_launcher():
    p, _ = run(["--gpus", "3", "--launch-only"])
    assert p.returncode == 0 and json.loads(p.stdout.strip())["n_gpus"] == 3, p.stderr[-2000:]
    p, _ = run(["--gpus", "1", "--launch-only", "--via-launcher"])
    out = json.loads(p.stdout.strip())
    assert p.returncode == 0 and out["n_gpus"] == 1 and out["launched"] is True, p.stderr[-2000:]


def test_hung_rank_ends_in_nonzero_exit():
    p, dt = run(["--gpus", "2", "--launch-only", "--hang-rank", "1", "--watchdog", "5"], timeout=240)
    assert p.returncode != 0, (p.stdout, p.stderr[-2000:])
    assert "watchdog" in p.stderr
    assert p.stdout.strip() == ""                                     # no result line from a failed run
    assert dt < 200


def test_launch_timeout_kills_the_group():
    p, dt = run(["--gpus", "2", "--launch-only", "--hang-rank", "0", "--watchdog", "1000", "--launch-timeout", "20"], timeout=240)
    assert p.returncode == 124, (p.returncode, p.stderr[-2000:])
    assert dt < 120


def test_world_size_mismatch_is_refused():
    p, _ = run(["--gpus", "2", "--launch-only"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_sigterm_on_the_launcher_ends_the_ranks():
    """a harness that gives up on `python bench.py --gpus N` sends SIGTERM to the launcher: the ranks sit in sessions of their own and must not
    outlive it"""
    import signal
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-only", "--hang-rank", "0", "--watchdog", "1000", "--launch-timeout", "600"],
                         env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    def descendants(pid):
        kids = {}
        for d in os.listdir("/proc"):
            if d.isdigit():
                try:
                    with open("/proc/%s/stat" % d) as f:
                        kids.setdefault(int(f.read().rsplit(")", 1)[1].split()[1]), []).append(int(d))
                except (OSError, ValueError, IndexError):
                    pass
        out, todo = [], [pid]
        while todo:
            for k in kids.get(todo.pop(), []):
                out.append(k); todo.append(k)
        return out
    deadline = time.time() + 120
    tree = []
    while time.time() < deadline:                                    # until the launcher's child and its two ranks exist
        tree = descendants(p.pid)
        if len(tree) >= 3:
            break
        time.sleep(0.5)
    assert len(tree) >= 3, tree
    time.sleep(3)
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=120)
    assert rc == 128 + signal.SIGTERM, rc
    time.sleep(1)
    alive = [k for k in tree if os.path.exists("/proc/%d" % k) and open("/proc/%d/stat" % k).read().rsplit(")", 1)[1].split()[0] != "Z"]
    assert not alive, alive
