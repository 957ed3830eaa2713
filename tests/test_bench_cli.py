"""bench.py's launch contract on CPU: `--gpus N` must really run N ranks (the topology of
hydrainfer/cluster/cluster.py:63-79: one node per GPU), and the timed steps must cover the workload
the metric names whatever --steps is."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_flag_starts_that_many_ranks():
    """No WORLD_SIZE in the environment: bench.py is its own launcher (fresh children through
    torch.distributed.run) and exits with their code; the line reports the ranks that actually ran."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=240, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True
    assert d["roles"] == ["EP", "D"]


def test_gpus_flag_must_match_world_size():
    """Under torch.distributed.run the flag and WORLD_SIZE must agree: a line for another rank count is refused."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120,
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=_env())
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_timed_contexts_cover_the_generation():
    sys.path.insert(0, ROOT)
    import bench
    full = bench.timed_contexts(704, 256, 255)
    assert full == list(range(705, 960))                       # the 255 decode steps of a 256-token generation
    assert bench.timed_contexts(704, 256, 400) == full
    for k in (2, 5, 20, 40, 64, 128):
        c = bench.timed_contexts(704, 256, k)
        assert len(c) == k and 705 <= c[0] and c[-1] <= 959
        assert len({b - a for a, b in zip(c, c[1:])}) == 1 and c[1] > c[0]     # one fixed spacing: the step's own advance
        assert abs(sum(c) / k - 832) <= 1.0                     # same mean context as the whole generation
        assert c[-1] - c[0] >= 228                              # ... and at least 90 % of its range
    assert bench.timed_contexts(704, 256, 1) == [832]
