"""Randomised engine runs on CPU: random arrival traces, budgets, priorities, topologies, prefix
sharing, end-of-sequence ids and decode look-ahead.  The stand-in model samples a closed-form
function of (input id, position), so every request's tokens are known independently of scheduling;
invariants: tokens match, every block of every pool is unpinned at the end, nothing is left
waiting for a FREE."""
import random
from types import SimpleNamespace as NS

import pytest
import torch

from hydrainfer_amd.engine import BatchSchedulerConfig, InstructionCreator, SamplingParameters, TokenRequest
from hydrainfer_amd.engine.node import LocalCluster
from tests.engine_util import CpuPoolManager, make_node
from tests.golden import cases as C
from tests.test_engine_trace import FakeGraphDecoder

N_IMG, BS, IMAGE_TOKEN = 24, 16, 32000


def closed_form(req, eos):
    ids = []
    for t in req.token_ids:
        ids += [IMAGE_TOKEN] * N_IMG if t == IMAGE_TOKEN else [t]
    out, last, pos = [], ids[-1], len(ids) - 1
    for _ in range(req.sampling_params.max_tokens):
        last = C.engine_trace_sample(last, pos)
        out.append(last)
        pos += 1
        if last == eos:
            break
    return out


class LM:
    image_token_id = IMAGE_TOKEN
    language_model = NS(shape=NS(num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1, head_dim=8))

    def forward(self, ids, feats, pos, params):
        i, p = ids.tolist(), pos.tolist()
        ap = params.attention_params[0]
        # what the kernels rely on: consistent lengths / slots / tables for every sequence
        q_cu, kv_cu, cu_b = ap.q_cu_seq_lens.tolist(), ap.kv_cu_seq_lens.tolist(), ap.cu_blocks_lens.tolist()
        slots, tables = ap.new_cache_slots.tolist(), ap.block_tables.tolist()
        for s in range(len(q_cu) - 1):
            kv_len = kv_cu[s + 1] - kv_cu[s]
            table = tables[cu_b[s]:cu_b[s + 1]]
            assert len(table) * BS >= kv_len
            for j in range(q_cu[s], q_cu[s + 1]):
                assert slots[j] == table[p[j] // BS] * BS + p[j] % BS
            assert p[q_cu[s + 1] - 1] == kv_len - 1
        if feats is not None:
            assert feats.shape[0] == sum(t == IMAGE_TOKEN for t in i)
        return torch.tensor([C.engine_trace_sample(i[j], p[j]) for j in params.selected_token_ids.tolist()])


class FakeStore:
    """In-memory stand-in for the torch.distributed store the mailbox runs on."""

    def __init__(self):
        self.kv = {}

    def add(self, key, n):
        self.kv[key] = self.kv.get(key, 0) + n
        return self.kv[key]

    def set(self, key, value):
        self.kv[key] = value

    def get(self, key):
        return self.kv[key]

    def check(self, keys):
        return all(k in self.kv for k in keys)


class WireCluster:
    """The nodes of a topology as RankEngines that talk only through pickled mailbox messages
    (engine/distributed.py) — stepped round-robin in one process."""

    def __init__(self, nodes, roles):
        from hydrainfer_amd.engine.distributed import RankEngine, StoreMailbox
        store = FakeStore()
        self.engines = [RankEngine(r, roles, n) for r, n in enumerate(nodes)]
        for e in self.engines:
            e.mailbox = StoreMailbox(store, e.rank, "fuzz")
        self.roles = roles
        self.n_added = [0, 0]

    def add_request(self, rcb):
        from hydrainfer_amd.engine.distributed import entry_rank
        from hydrainfer_amd.engine.isa import ImageEmbed
        has_image = isinstance(rcb.current_instruction(), ImageEmbed)
        r = entry_rank(self.n_added[has_image], self.roles, has_image)
        self.n_added[has_image] += 1
        self.engines[r].node.add_request(rcb)

    def step(self):
        for e in self.engines:
            e.step()

    def idle(self):
        return all(e.node.idle() and not e.held and not e.outbox for e in self.engines) and \
            all(not e.mailbox.store.check([f"fuzz/m/{e.rank}/{e.mailbox.next_slot}"]) for e in self.engines)


class Vision:
    def forward(self, px):
        return torch.zeros(px.shape[0], N_IMG, 8)


@pytest.mark.parametrize("seed", list(range(24)) + [40, 64, 83, 90, 106, 149])   # the last six: fully cached prompts
def test_engine_fuzz_closed_form(seed):
    rng = random.Random(seed)
    topology = rng.choice([["EPD"], ["EPD"], ["EP", "D"], ["E", "P", "D"], ["E", "PD"], ["ED", "P"], ["E", "P", "D", "D"]])
    max_running = rng.choice([2, 3, 5, 8])
    cfg = BatchSchedulerConfig(priority=rng.choice(["prefill", "decode"]), max_running_requests=max_running,
                               chunked_prefill=rng.random() < 0.7, token_budgets=rng.choice([24, 40, 64, 200]),
                               image_budgets=rng.choice([1, 2, 4]))
    lookahead = rng.random() < 0.5
    eager_migrate = rng.random() < 0.5
    n_req = rng.randint(4, 14)
    g = torch.Generator().manual_seed(seed)
    reqs, arrivals = [], []
    shared_text = torch.randint(1000, 31999, (rng.randint(20, 60),), generator=g).tolist()
    for i in range(n_req):
        has_image = rng.random() < 0.7
        if rng.random() < 0.3:
            text = list(shared_text)                       # prefix sharing (hits only with equal images)
        else:
            text = torch.randint(1000, 31999, (rng.randint(1, 70),), generator=g).tolist()
        reqs.append(TokenRequest(i, ([IMAGE_TOKEN] if has_image else []) + text,
                                 torch.zeros(1, 3, 2, 2) if has_image else None, (8, 8),
                                 777 if rng.random() < 0.5 else 1000 + i,
                                 SamplingParameters(max_tokens=rng.randint(1, 12))))
        arrivals.append(rng.randint(0, 25))
    eos = None
    if rng.random() < 0.4:     # an id that really occurs somewhere mid-stream
        cand = [t for r in reqs for t in closed_form(r, None)[:-1]]
        eos = rng.choice(cand) if cand else None
    # pools: enough for twice max_running requests of the longest kind
    worst = (N_IMG + 70 + 12 + BS - 1) // BS + 1
    nodes = []
    for k, t in enumerate(topology):
        kv = CpuPoolManager(1, 2, worst * (2 * max_running + 2), BS, 1, 8, seed=k)
        img = CpuPoolManager(1, 1, 2 * max_running + 2, N_IMG, 1, 8, seed=100 + k)
        node = make_node(f"{t}{k}", t, LM(), Vision(), kv, img, LM.language_model.shape, torch.float32,
                         torch.device("cpu"), BatchSchedulerConfig(**vars(cfg)), eager_migrate=eager_migrate)
        if lookahead and node.executor.fill_executor is not None and node.node_type.enable_decode:
            node.executor.fill_executor.graph_decoder = FakeGraphDecoder()
        nodes.append(node)
    wire = len(topology) > 1 and rng.random() < 0.5
    cluster = WireCluster(nodes, topology) if wire else LocalCluster(nodes)
    creator = InstructionCreator(IMAGE_TOKEN, N_IMG, BS, ignore_eos=eos is None, eos_token_id=eos if eos else 2)
    rcbs, step = [None] * n_req, 0
    while step <= max(arrivals) or not cluster.idle():
        for i in range(n_req):
            if arrivals[i] == step:
                rcbs[i] = creator.process(reqs[i])
                cluster.add_request(rcbs[i])
        cluster.step()
        step += 1
        assert step < 3000, "engine did not drain"
    done = {r.request_id: r for n in nodes for r in n.finished}      # (over the wire a request is re-created)
    for i, r in enumerate(reqs):
        assert done[i].output_token_ids == closed_form(r, eos), \
            f"request {i} ({topology}, lookahead={lookahead}, wire={wire})"
        assert len(done[i].metric.token_times) == len(done[i].output_token_ids)
    for node in nodes:
        for m in (node.kv_cache_block_manager, node.image_cache_block_manager):
            if m is not None:
                assert len(m.shared_cache.to_be_evicted) == m.n_blocks, f"{node.name}: blocks still pinned"
        assert node.batch_scheduler.migrating_cnt == 0
    assert sum(len(n.finished) for n in nodes) == n_req


def test_request_that_does_not_fit_waits_for_blocks():
    """A pool with room for one request at a time: the second request is deferred (not crashed on,
    as the reference would) until the first has finished and freed its blocks."""
    kv = CpuPoolManager(1, 2, 6, BS, 1, 8)                    # 6 blocks = 96 tokens
    img = CpuPoolManager(1, 1, 2, N_IMG, 1, 8)
    cfg = BatchSchedulerConfig(max_running_requests=2, token_budgets=200, image_budgets=2)
    node = make_node("EPD", "EPD", LM(), Vision(), kv, img, LM.language_model.shape, torch.float32,
                     torch.device("cpu"), cfg)
    cluster = LocalCluster([node])
    creator = InstructionCreator(IMAGE_TOKEN, N_IMG, BS)
    g = torch.Generator().manual_seed(1)
    reqs = [TokenRequest(i, torch.randint(1000, 31999, (60,), generator=g).tolist(), None, (0, 0), 0,
                         SamplingParameters(max_tokens=20)) for i in range(3)]      # 80 tokens = 5 blocks each
    rcbs = [creator.process(r) for r in reqs]
    for rcb in rcbs:
        cluster.add_request(rcb)
    steps = 0
    while not cluster.idle():
        cluster.step()
        steps += 1
        assert steps < 500
    for r, rcb in zip(reqs, rcbs):
        assert rcb.output_token_ids == closed_form(r, None)
    assert len(kv.shared_cache.to_be_evicted) == kv.n_blocks
    # they ran one after the other: each needs 20 decode steps
    assert steps >= 3 * 20
