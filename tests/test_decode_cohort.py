"""CPU: the steady-state decode cohort (engine/executor.py::DecodeCohort) is invisible.  The same arrival traces run with
the cohort on and off through the node's real scheduler / step loop / executor / stager, in front of a stand-in for the
device (a closed-form sampler that reads exactly what hx_stage_decode would have put into the resident buffer): every
request's tokens, the number of its time stamps, its final instruction-chain position, every pool's free blocks and the
device-side inputs of EVERY launch (ids resolved through the look-ahead feed, positions, slots, lengths, the block table
each row's offset points at) must be the same."""
import random
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from hydrainfer_amd.engine import BatchSchedulerConfig, InstructionCreator, SamplingParameters, TokenRequest
from hydrainfer_amd.engine.graph_decode import DecodeStager
from hydrainfer_amd.engine.node import LocalCluster
from tests.engine_util import CpuPoolManager, make_node
from tests.golden import cases as C
from tests.test_decode_stager import apply_staging

BS, CAP, IMAGE_TOKEN = 16, 24, 32000


class ResidentDecoder:
    """GraphedDecoder's interface over a numpy 'device': stage -> apply -> read the rows back OUT OF THE RESIDENT BUFFER and
    sample in closed form (tests.golden.cases.engine_trace_sample) — a wrong offset, a missed block append or a stale
    position changes the tokens."""

    def __init__(self, max_batch):
        self.pad_to, self.max_batch, self.cap = 4, max_batch, CAP
        self.stager = DecodeStager(max_batch, CAP, BS, 4095, 4096, 1 << 30)
        self.dev = np.full(self.stager.total_words, -7, dtype=np.int32)
        self.st = [np.zeros(self.stager.staging_words, dtype=np.int32) for _ in range(2)]
        self.launches, self.tokens, self.inputs = 0, {}, []
        self.n_cohort = 0

    def fits(self, n, max_blocks):
        return n <= self.max_batch and max_blocks <= self.cap

    def _run(self, n, B, st):
        apply_staging(self.dev, st, self.stager.head_words)
        o, d = self.stager.off, self.dev
        prev = self.tokens.get(self.launches, [])
        self.launches += 1
        out, seen = [], []
        for r in range(n):
            src = int(d[o["src"] + r])
            tok = prev[src] if src >= 0 else int(d[o["ids"] + r])
            pos, slot = int(d[o["pos"] + r]), int(d[o["slots"] + r])
            kv = int(d[o["kv_cu"] + r + 1] - d[o["kv_cu"] + r])
            start = self.stager.tables_off + int(d[o["cu_blocks"] + r])
            table = d[start:start + (kv + BS - 1) // BS].tolist()
            assert kv == pos + 1 and slot == table[pos // BS] * BS + pos % BS, (r, pos, slot, table)
            seen.append((tok, pos, slot, kv, tuple(table)))
            out.append(C.engine_trace_sample(tok, pos))
        from hydrainfer_amd.layer.causal_attention import decode_rank_descriptor
        kv_all = [int(d[o["kv_cu"] + r + 1] - d[o["kv_cu"] + r]) for r in range(B)]
        assert d[o["rank"]:o["rank"] + B + 1].tolist() == decode_rank_descriptor(kv_all)      # both stagers write the same descriptor
        self.inputs.append(seen)
        self.tokens[self.launches] = out
        return self.launches

    def launch(self, rows):
        B = (len(rows) + 3) // 4 * 4
        st = self.st[self.launches % 2]
        self.stager.stage(st, rows, B)
        return self._run(len(rows), B, st)

    def launch_cohort(self, n, pos, slots, starts, grown):
        B = (n + 3) // 4 * 4
        st = self.st[self.launches % 2]
        self.stager.stage_cohort(st, n, B, pos, slots, starts, grown)
        self.n_cohort += 1
        return self._run(n, B, st)

    def fetch(self, launch_id):
        return self.tokens[launch_id]

    def warmup(self, *a, **k):
        pass


def _run_trace(seed, cohort, with_streams=False, with_eos=False):
    rnd = random.Random(seed)
    shape = NS(num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1, head_dim=8, max_position_embeddings=4096)

    class LM:
        image_token_id = IMAGE_TOKEN
        language_model = NS(shape=shape)

        def forward(self, ids, feats, pos, params):
            i, p = ids.tolist(), pos.tolist()
            return torch.tensor([C.engine_trace_sample(i[j], p[j]) for j in params.selected_token_ids.tolist()])

    max_running = rnd.choice((4, 8, 12))
    kv = CpuPoolManager(1, 2, 40 * CAP, BS, 1, 8)
    cfg = BatchSchedulerConfig(priority=rnd.choice(("prefill", "decode")), max_running_requests=max_running, chunked_prefill=True,
                               token_budgets=rnd.choice((48, 96, 256)), image_budgets=2)
    node = make_node("EPD0", "EPD", LM(), None, kv, None, shape, torch.float32, torch.device("cpu"), cfg)
    fe = node.executor.fill_executor
    fe.graph_decoder = ResidentDecoder(16)
    fe.cohort_enabled = cohort
    cluster = LocalCluster([node])
    creator = InstructionCreator(IMAGE_TOKEN, 576, BS)
    n_req = rnd.randint(3, 14)
    reqs, arrive, logs = [], [], {}
    for i in range(n_req):
        # a third of the requests name end-of-sequence ids — a tenth of the sampler's range, so some end early
        eos = [t for t in range(100, 30100) if t % 10 == i % 10] if with_eos and i % 3 == 0 else []
        reqs.append(TokenRequest(i, [rnd.randint(1000, 31000) for _ in range(rnd.randint(3, 70))], None, (8, 8), 100 + i,
                                 SamplingParameters(max_tokens=rnd.randint(1, 90), eos_token_ids=eos)))
        arrive.append(rnd.choice((0, 0, 0, rnd.randint(1, 40), rnd.randint(40, 160))))
    rcbs = [None] * n_req
    step = 0
    while step <= max(arrive) or not cluster.idle():
        for i, a in enumerate(arrive):
            if a == step:
                rcbs[i] = creator.process(reqs[i])
                if with_streams and i % 2 == 0:
                    from hydrainfer_amd.engine.rcb import LogOutputTokenProcessor
                    logs[i] = LogOutputTokenProcessor()
                    rcbs[i].register_output_token_processor(logs[i])
                cluster.add_request(rcbs[i])
        cluster.step()
        step += 1
        assert step < 5000
    fe.resolve_pending()
    assert len(kv.shared_cache.to_be_evicted) == kv.n_blocks
    return {"tokens": [r.output_token_ids for r in rcbs], "stamps": [len(r.metric.token_times) for r in rcbs],
            "inputs": fe.graph_decoder.inputs, "streams": {i: l.token_ids for i, l in logs.items()},
            "n_cohort": fe.graph_decoder.n_cohort, "n_launches": fe.graph_decoder.launches, "steps": step}


@pytest.mark.parametrize("seed", range(40))
def test_the_cohort_changes_nothing(seed):
    kw = dict(with_streams=seed % 3 == 0, with_eos=seed % 2 == 1)
    on, off = _run_trace(seed, True, **kw), _run_trace(seed, False, **kw)
    assert off["n_cohort"] == 0
    for k in ("tokens", "stamps", "streams", "n_launches", "steps"):
        assert on[k] == off[k], k
    assert on["inputs"] == off["inputs"]              # every launch saw the same rows, tables and tokens on the device
    for toks in on["tokens"]:
        assert all(isinstance(t, int) for t in toks)
    if kw["with_eos"]:
        on_early = _run_trace(seed, True, **kw)
        assert on_early["tokens"] == on["tokens"]          # (deterministic)


def test_the_cohort_is_actually_used():
    used = sum(_run_trace(seed, True)["n_cohort"] for seed in range(10))
    total = sum(_run_trace(seed, True)["n_launches"] for seed in range(10))
    assert used > 0.5 * total, (used, total)
