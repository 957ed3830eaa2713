"""A hand-over that fails ends THAT request, never the node (hydrainfer/cluster/epdnode.py:428-442: two attempts, then the
request's blocks are freed and `(request_id, None)` goes to its stream).  CPU: E, P and D nodes of one LocalCluster over
the closed-form stand-in model of tests/test_distributed_engine_cpu.py, pools whose pull can be made to fail."""
from types import SimpleNamespace as NS

import pytest
import torch

from tests.golden import cases as C
from tests.test_distributed_engine_cpu import BS, IMAGE_TOKEN, N_IMG, _requests, expected_tokens


class FlakyPool:
    """tests.engine_util.CpuPoolManager whose receiver-side pull raises for the requests in `fail_plan`
    ({n_cache_tokens of the source cache: times to fail})."""

    def __new__(cls, *a, fail_plan=None, **k):
        from tests.engine_util import CpuPoolManager

        class Pool(CpuPoolManager):
            def migrate_blocks(self, src, dst, is_send=False):
                if not is_send and self.fail_plan.get(src.n_cache_tokens, 0) > 0:
                    self.fail_plan[src.n_cache_tokens] -= 1
                    self.n_failed += 1
                    raise RuntimeError("injected: the peer's pool could not be read")
                super().migrate_blocks(src, dst, is_send)
        pool = Pool(*a, **k)
        pool.fail_plan, pool.n_failed = dict(fail_plan or {}), 0
        return pool


def _cluster(fail_plan):
    from hydrainfer_amd.engine import BatchSchedulerConfig
    from hydrainfer_amd.engine.node import LocalCluster
    from tests.engine_util import make_node
    shape = NS(num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1, head_dim=8)

    class LM:
        image_token_id = IMAGE_TOKEN
        language_model = NS(shape=shape)

        def forward(self, ids, feats, pos, params):
            i, p = ids.tolist(), pos.tolist()
            return torch.tensor([C.engine_trace_sample(i[j], p[j]) for j in params.selected_token_ids.tolist()])

    class Vision:
        def forward(self, px):
            return torch.zeros(px.shape[0], N_IMG, 8)

    cfg = BatchSchedulerConfig(max_running_requests=4, token_budgets=64, image_budgets=2)
    nodes, pools = [], []
    for name, role in (("E0", "E"), ("P1", "P"), ("D2", "D")):
        kv = FlakyPool(1, 2, 64, BS, 1, 8, fail_plan=fail_plan if role == "D" else None)
        img = FlakyPool(1, 1, 10, N_IMG, 1, 8)
        pools.append((kv, img))
        nodes.append(make_node(name, role, LM(), Vision(), kv, img, shape, torch.float32, torch.device("cpu"), cfg))
    return LocalCluster(nodes), nodes, pools


def _run(fail_plan):
    from hydrainfer_amd.engine import InstructionCreator
    from hydrainfer_amd.engine.rcb import LogOutputTokenProcessor
    from tests.engine_util import run_trace
    cluster, nodes, pools = _cluster(fail_plan)
    reqs = _requests(10)
    creator = InstructionCreator(IMAGE_TOKEN, N_IMG, BS)
    logs = {}
    real = creator.process

    def process(r):
        rcb = real(r)
        logs[r.request_id] = LogOutputTokenProcessor()
        rcb.register_output_token_processor(logs[r.request_id])
        return rcb
    creator.process = process
    rcbs = run_trace(cluster, creator, [(i, r) for i, r in enumerate(reqs)])
    return reqs, rcbs, logs, nodes, pools


def _prompt_tokens(req):
    return sum(N_IMG if t == IMAGE_TOKEN else 1 for t in req.token_ids)


def test_a_pull_that_fails_twice_ends_that_request_only():
    reqs = _requests(10)
    victim = 3
    n_tok = _prompt_tokens(reqs[victim])
    assert sum(_prompt_tokens(r) == n_tok for r in reqs) == 1, "the plan must single out one request"
    reqs, rcbs, logs, nodes, pools = _run({n_tok: 2})
    d = nodes[2]
    assert pools[2][0].n_failed == 2
    # the request ended with the reference's None token; what it had sampled before (its prefill's token) stays
    assert [r.request_id for r in d.failed] == [victim] and "failed 2 times" in rcbs[victim].failed
    assert logs[victim].token_ids[-1] is None and logs[victim].token_ids[:-1] == expected_tokens(reqs[victim])[:1]
    # every other request is untouched
    for i, r in enumerate(reqs):
        if i != victim:
            assert rcbs[i].output_token_ids == expected_tokens(r) and logs[i].token_ids == expected_tokens(r), i
    assert sorted(r.request_id for r in d.finished) == [i for i in range(len(reqs)) if i != victim]
    # blocks: all back, on the sender (told to free by the receiver) and on the receiver (took some, returned them)
    for node in nodes:
        for m in (node.kv_cache_block_manager, node.image_cache_block_manager):
            if m is not None:
                assert len(m.shared_cache.to_be_evicted) == m.n_blocks, node.name
        assert node.batch_scheduler.migrating_cnt == 0 and node.idle()


def test_a_pull_that_fails_once_is_retried():
    reqs = _requests(10)
    n_tok = _prompt_tokens(reqs[3])
    reqs, rcbs, logs, nodes, pools = _run({n_tok: 1})
    assert pools[2][0].n_failed == 1 and not nodes[2].failed
    for i, r in enumerate(reqs):
        assert rcbs[i].output_token_ids == expected_tokens(r) and logs[i].token_ids == expected_tokens(r), i
    for node in nodes:
        for m in (node.kv_cache_block_manager, node.image_cache_block_manager):
            if m is not None:
                assert len(m.shared_cache.to_be_evicted) == m.n_blocks, node.name


def test_no_live_downstream_node_ends_the_request_at_the_sender():
    """Every D node of the hop is gone (RoundRobin.remove_worker, what RankEngine does on a dead peer): a request that
    reaches its hand-over is terminated where it is, its blocks freed."""
    from hydrainfer_amd.engine import InstructionCreator
    from hydrainfer_amd.engine.rcb import LogOutputTokenProcessor
    cluster, nodes, pools = _cluster({})
    nodes[1].pd_loadbalancer.remove_worker(nodes[2])
    creator = InstructionCreator(IMAGE_TOKEN, N_IMG, BS)
    req = _requests(1)[0]
    rcb = creator.process(req)
    log = LogOutputTokenProcessor()
    rcb.register_output_token_processor(log)
    cluster.add_request(rcb)
    cluster.run_until_idle(200)
    assert [r.request_id for r in nodes[1].failed] == [req.request_id] and "no live decode node" in rcb.failed
    assert log.token_ids[-1] is None and not nodes[2].finished
    for node in nodes:
        for m in (node.kv_cache_block_manager, node.image_cache_block_manager):
            if m is not None:
                assert len(m.shared_cache.to_be_evicted) == m.n_blocks, node.name
