"""-m gpu: `bench.py --gpus 3` end to end on ONE GPU (HX_SINGLE_DEVICE=1: every rank uses cuda:0; HX_DIST_BACKEND=gloo:
RCCL cannot run several ranks on one device) — the launch contract of the driver's multi-GPU runs, with the tiny
model: three ranks really start, take the roles E, P, D (hydrainfer/cluster/cluster.py:63-79 one node per GPU;
BASELINE configs[3]), requests enter at E, image blocks are pulled E->P and KV blocks P->D with hx_migrate_blocks over
IPC-mapped pools, the Poisson trace of BASELINE configs[4] (benchmark/timestamp.py:9-16) is replayed, and the ONE JSON
line carries the `disaggregated` and `migration` objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_three_ranks_epd_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HX_DIST_BACKEND="gloo", HX_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--model", "tiny", "--batch", "8",
                        "--steps", "8", "--warmup", "2", "--rate", "40", "--no-cpu-baseline", "--no-13b"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["steps"] == 8 and d["value"] > 0
    dg = d["disaggregated"]
    assert dg is not None and "error" not in dg, dg
    assert dg["roles"] == ["E", "P", "D"] and dg["n_ranks"] == 3
    assert dg["requests"] == 8 and dg["output_tokens"] == 8 * 256
    assert "Poisson" in dg["arrivals"] and dg["rate_req_s"] == 40
    assert dg["ep_pull_p50_ms"] is not None and dg["pd_pull_p50_ms"] is not None and dg["pd_pull_GBps"] > 0
    assert dg["burst_at_t0"]["requests"] == 8 and dg["burst_at_t0"]["output_tokens"] == 8 * 256
    mg = d["migration"]
    assert mg is not None and "error" not in mg, mg
    assert d["legs_failed"] == [] and d["wedged_ranks"] == [], (d["legs_failed"], d["wedged_ranks"])


def test_bench_eight_ranks_hybrid_pool_on_one_gpu():
    """BASELINE configs[4] (hydrainfer/config/cluster/hybrid.yaml: 2 encode + 2 prefill + 4 decode nodes) as EIGHT
    processes on one device: what the first real 8-GPU run exercises first — two E ranks feeding two P ranks feeding
    four D ranks, every hop a round robin over all its downstream nodes (cluster/epdnode.py:56-75,419-420), a D rank
    pulling KV out of two different P pools, the front door alternating over both E ranks (cluster.py:178-184) — with the
    real kernels, real IPC pulls and the Poisson trace, through the driver's launch contract."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HX_DIST_BACKEND="gloo", HX_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--model", "tiny", "--batch", "8",
                        "--steps", "8", "--warmup", "2", "--rate", "40", "--no-cpu-baseline", "--no-13b"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 8 and d["value"] > 0
    dg = d["disaggregated"]
    assert dg is not None and "error" not in dg, dg
    roles = ["E", "E", "P", "P", "D", "D", "D", "D"]
    assert dg["roles"] == roles and dg["n_ranks"] == 8
    n = 8 * 4                                   # --batch requests per D rank
    assert dg["requests"] == n and dg["output_tokens"] == n * 256
    assert "Poisson" in dg["arrivals"] and dg["rate_req_s"] == 40 * 4
    assert dg["burst_at_t0"]["requests"] == n and dg["burst_at_t0"]["output_tokens"] == n * 256
    assert dg["ep_pull_p50_ms"] is not None and dg["pd_pull_p50_ms"] is not None and dg["pd_pull_GBps"] > 0
    for leg in (dg, dg["burst_at_t0"]):
        pairs = leg["pulls_per_pair"]
        for e in (0, 1):                        # every E rank handed image blocks to both P ranks, evenly
            c = [pairs.get(f"{e}->{p}", 0) for p in (2, 3)]
            assert sum(c) == n // 2 and abs(c[0] - c[1]) <= 1, pairs
        for p in (2, 3):                        # every P rank handed KV to all four D ranks, evenly
            c = [pairs.get(f"{p}->{dd}", 0) for dd in (4, 5, 6, 7)]
            assert sum(c) == n // 2 and max(c) - min(c) <= 1 and min(c) > 0, pairs
        assert sum(pairs.values()) == 2 * n     # two hops per request, nothing else
    assert d["legs_failed"] == [] and d["wedged_ranks"] == [], (d["legs_failed"], d["wedged_ranks"])
