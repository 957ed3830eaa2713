"""G11: the engine's host path (InstructionCreator, BatchScheduler, LanguageModelParametersBuilder,
both executors' bookkeeping, EPDNode.step) against a trace recorded from the reference's own
classes (tests/golden/generate_goldens.py::gen_engine_trace).  Integer-exact, CPU only: the two
models are replaced by the same stand-ins the generator used, device memory by CPU tensors."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from hydrainfer_amd.engine import (BatchScheduler, BatchSchedulerConfig, BatchSchedulerContext, Fill,
                                   InstructionCreator, SamplingParameters, TokenRequest)
from hydrainfer_amd.engine.executor import BatchFillExecutor, BatchImageEmbedExecutor, InstructionExecutor
from hydrainfer_amd.engine.node import EPDNode, LocalCluster, NodeType
from hydrainfer_amd.memory import compute_image_hash
from tests.engine_util import CpuPoolManager
from tests.golden import cases as C

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g11_engine_trace.npz")
INST_CODES = {"EM": 0, "TF": 1, "EF": 2, "IE": 3, "EPMR": 4, "PDMR": 5, "PR": 6}


class Ragged:
    def __init__(self, z, prefix):
        self.z, self.prefix, self.cursor = z, prefix, {}

    def rows(self, name):
        flat, off = self.z[f"{self.prefix}_{name}_flat"], self.z[f"{self.prefix}_{name}_off"]
        return [flat[off[i]:off[i + 1]] for i in range(len(off) - 1)]

    def next(self, name):
        i = self.cursor.get(name, 0)
        self.cursor[name] = i + 1
        return self.rows(name)[i]


def build_node(cfg, gold, log):
    kv = CpuPoolManager(cfg.n_layers, 2, cfg.kv_blocks, cfg.block_size, cfg.n_heads, cfg.head_dim)
    img = CpuPoolManager(1, 1, cfg.image_blocks, cfg.n_image_tokens, cfg.n_heads, cfg.head_dim)
    shape = NS(num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
               num_key_value_heads=cfg.n_heads, head_dim=cfg.head_dim)

    class LM:                       # same stand-in as the generator's Worker
        image_token_id = cfg.image_token_id
        language_model = NS(shape=shape)

        def forward(self, input_ids, image_features, position_ids, params):
            ap = params.attention_params[0]
            sel = params.selected_token_ids.tolist()
            log.append(dict(input_ids=input_ids, position_ids=position_ids, selected=sel,
                            image_rows=[] if image_features is None else image_features[:, 0].round().long(),
                            q_cu=ap.q_cu_seq_lens, kv_cu=ap.kv_cu_seq_lens, new_cache_slots=ap.new_cache_slots,
                            block_tables=ap.block_tables, cu_blocks_lens=ap.cu_blocks_lens,
                            fill_scalars=[ap.num_sequences, int(ap.all_sequences_decode), ap.q_max_seq_len]))
            ids, pos = input_ids.tolist(), position_ids.tolist()
            return torch.tensor([C.engine_trace_sample(ids[j], pos[j]) for j in sel], dtype=torch.int32)

    class Vision:
        def forward(self, pixels):
            f = torch.zeros(pixels.shape[0], cfg.n_image_tokens, cfg.n_heads * cfg.head_dim)
            for k in range(pixels.shape[0]):
                f[k, :, 0] = int(pixels[k].flatten()[0].item()) * 1000 + torch.arange(cfg.n_image_tokens)
            log.append(dict(encode_requests=[int(pixels[k].flatten()[0].item()) for k in range(pixels.shape[0])]))
            return f

    dev = torch.device("cpu")
    executor = InstructionExecutor(
        BatchFillExecutor(LM(), kv, img, torch.float32, dev),
        BatchImageEmbedExecutor(Vision(), img, cfg.n_heads, cfg.head_dim, torch.float32, dev))
    sched = BatchScheduler(
        BatchSchedulerConfig(priority=cfg.priority, max_running_requests=cfg.max_running_requests,
                             chunked_prefill=cfg.chunked_prefill, token_budgets=cfg.token_budgets,
                             image_budgets=cfg.image_budgets),
        BatchSchedulerContext(kv_cache_block_manager=kv, image_cache_block_manager=img))
    node = EPDNode("EPD", NodeType("EPD"), sched, executor, kv, img)
    return node, sched, kv, img


@pytest.mark.parametrize("cfg", C.ENGINE_TRACES, ids=lambda c: c.tag)
def test_engine_trace_matches_reference(cfg):
    z = np.load(GOLD)
    gold = Ragged(z, f"engine{cfg.tag}")
    reqs = C.engine_trace_requests(cfg)
    log = []
    node, sched, kv, img = build_node(cfg, gold, log)
    cluster = LocalCluster([node])
    creator = InstructionCreator(image_token_id=cfg.image_token_id, n_image_tokens_per_image=cfg.n_image_tokens,
                                 block_size=cfg.block_size, ignore_eos=True)
    batches = []
    real_step = sched.step

    def recording_step():
        b = real_step()
        batches.append([(rcb.sid, INST_CODES[repr(inst)], len(inst.token_ids) if isinstance(inst, Fill) else 0)
                        for rcb, inst in b])
        return b
    sched.step = recording_step

    rcbs, free_rows = [], []
    n_steps = int(z[f"engine{cfg.tag}_n_steps"][0])
    for step in range(n_steps):
        for i, r in enumerate(reqs):
            if r.arrival_step != step:
                continue
            pixels, image_hash = None, 0
            if r.image_seed >= 0:
                pixels = torch.full((1, 3, 2, 2), float(i))
                image_hash = compute_image_hash(C.engine_trace_image(r.image_seed))   # pinned by G11's prefix hashes
            rcb = creator.process(TokenRequest(i, r.token_ids, pixels, (8, 8), image_hash,
                                               SamplingParameters(max_tokens=r.max_tokens)))
            first = rcb.instructions.head.next
            fill = first if isinstance(first, Fill) else first.next.next.next
            want = gold.next("prefix_hashes").view(np.uint64)
            assert np.array_equal(np.array(fill.hashes, dtype=np.uint64), want), f"prefix hashes of request {i}"
            rcbs.append(rcb)
            cluster.add_request(rcb)
        cluster.step()
        free_rows.append([kv.get_num_avaiable_blocks(), img.get_num_avaiable_blocks(),
                          len(kv.block_allocator.free_blocks), len(sched.running), len(sched.waiting)])
    assert cluster.idle()

    # 1. batch composition of every scheduler step
    for s, got in enumerate(batches):
        assert [g[0] for g in got] == gold.rows("batch_sid")[s].tolist(), f"step {s}: sids"
        assert [g[1] for g in got] == gold.rows("batch_inst")[s].tolist(), f"step {s}: instructions"
        assert [g[2] for g in got] == gold.rows("batch_ntok")[s].tolist(), f"step {s}: token counts"
    assert len(batches) == len(gold.rows("batch_sid"))

    # 2. model inputs of every fill batch / encode batch
    fills = [e for e in log if "input_ids" in e]
    encodes = [e for e in log if "encode_requests" in e]
    assert len(fills) == len(gold.rows("input_ids"))
    n_chunk_heads = 0
    for k, e in enumerate(fills):
        for name in ("input_ids", "position_ids", "selected", "image_rows", "q_cu", "new_cache_slots",
                     "block_tables", "cu_blocks_lens", "fill_scalars"):
            got = np.asarray(torch.as_tensor(e[name]).numpy() if not isinstance(e[name], list) else e[name])
            assert np.array_equal(got.astype(np.int64), gold.rows(name)[k]), f"fill batch {k}: {name}"
        # kv lengths: equal to the reference except on the head chunk of a budget-chunked prefill,
        # where the engine uses the tokens actually present (parameters_builder.py docstring)
        q_cu, pos = e["q_cu"].tolist(), e["position_ids"].tolist()
        kv_len = np.diff(e["kv_cu"].numpy())
        ref_len = np.diff(gold.rows("kv_cu_reference")[k])
        for j in range(len(kv_len)):
            assert kv_len[j] == pos[q_cu[j + 1] - 1] + 1
            if kv_len[j] != ref_len[j]:
                assert kv_len[j] < ref_len[j]
                n_chunk_heads += 1
    assert (n_chunk_heads > 0) == cfg.chunked_prefill
    assert [e["encode_requests"] for e in encodes] == [r.tolist() for r in gold.rows("encode_requests")]

    # 3. pool accounting after every step and the generated tokens
    assert free_rows == [r.tolist() for r in gold.rows("free_blocks")]
    for i, rcb in enumerate(rcbs):
        assert rcb.output_token_ids == gold.rows("output_token_ids")[i].tolist(), f"tokens of request {i}"
        assert len(rcb.output_token_ids) == reqs[i].max_tokens


class FakeGraphDecoder:
    """CPU stand-in with GraphedDecoder's launch/fetch contract (tokens stay 'on the device' until
    fetched; a negative token means 'row -(t+1) of the previous launch')."""

    def __init__(self):
        self.launches, self.tokens, self.n_launch_ahead = 0, {}, 0

    def fits(self, n_seqs, n_blocks):
        return True

    def launch(self, rows):
        prev = self.tokens.get(self.launches, [])
        out = []
        for token, pos, slot, kv_len, table, _sid in rows:
            if token < 0:
                token = prev[-(token + 1)]
                self.n_launch_ahead += 1
            assert kv_len == pos + 1 and slot == table[pos // 16] * 16 + pos % 16
            out.append(C.engine_trace_sample(token, pos))
        self.launches += 1
        self.tokens[self.launches] = out
        return self.launches

    def fetch(self, launch_id):
        assert launch_id > self.launches - 2
        return list(self.tokens[launch_id])


def _run_plain(cfg, eos, decoder):
    reqs = C.engine_trace_requests(cfg)
    node, sched, kv, img = build_node(cfg, None, [])
    node.executor.fill_executor.graph_decoder = decoder
    cluster = LocalCluster([node])
    creator = InstructionCreator(image_token_id=cfg.image_token_id, n_image_tokens_per_image=cfg.n_image_tokens,
                                 block_size=cfg.block_size, ignore_eos=eos is None,
                                 eos_token_id=eos if eos is not None else 2)
    rcbs, step = [None] * len(reqs), 0
    while step <= max(r.arrival_step for r in reqs) or not cluster.idle():
        for i, r in enumerate(reqs):
            if r.arrival_step == step:
                px = torch.full((1, 3, 2, 2), float(i)) if r.image_seed >= 0 else None
                rcbs[i] = creator.process(TokenRequest(i, r.token_ids, px, (8, 8), 100 + r.image_seed,
                                                       SamplingParameters(max_tokens=r.max_tokens)))
                cluster.add_request(rcbs[i])
        cluster.step()
        step += 1
        assert step < 500
    for m in (kv, img):
        assert len(m.shared_cache.to_be_evicted) == m.n_blocks
    return [r.output_token_ids for r in rcbs], [len(r.metric.token_times) for r in rcbs]


@pytest.mark.parametrize("cfg", C.ENGINE_TRACES, ids=lambda c: c.tag)
def test_decode_lookahead_is_invisible(cfg):
    """Launching decode step N+1 before reading step N's tokens changes no request's output,
    with and without an end-of-sequence token cutting requests short."""
    plain, _ = _run_plain(cfg, None, None)
    dec = FakeGraphDecoder()
    ahead, n_times = _run_plain(cfg, None, dec)
    assert ahead == plain and dec.n_launch_ahead >= 5
    assert n_times == [len(t) for t in plain]
    # pick an end-of-sequence id that really occurs mid-request
    eos = next(t[2] for t in plain if len(t) >= 6)
    plain_eos, _ = _run_plain(cfg, eos, None)
    ahead_eos, n_times = _run_plain(cfg, eos, FakeGraphDecoder())
    assert plain_eos == ahead_eos and n_times == [len(t) for t in plain_eos]
    assert any(len(a) < len(b) and a[-1] == eos for a, b in zip(plain_eos, plain))


def test_profiler_budget_search_matches_reference():
    from hydrainfer_amd.engine.profiler import binary_search_max_batch_size
    z = np.load(GOLD)
    for hi in (8, 2048):
        want = z[f"profiler_search_{hi}"]
        got = [binary_search_max_batch_size(1, hi, lambda n, T=T: n <= T) for T in range(hi + 3)]
        got.append(binary_search_max_batch_size(1, hi, C.profiler_weird_criterion))
        assert got == want.tolist()


def test_profiler_runs_the_executors_and_frees_its_blocks():
    """Budgets come out of timing real executor calls on synthetic batches (CPU stand-ins here)."""
    from hydrainfer_amd.engine.profiler import BatchSchedulerProfiler, BatchSchedulerProfilerConfig
    cfg = C.ENGINE_TRACES[0]
    log = []
    node, sched, kv, img = build_node(cfg, None, log)
    prof = BatchSchedulerProfiler(BatchSchedulerProfilerConfig(tpot_slo=10.0, n_warmup_iter=1, n_profile_iter=1),
                                  node.executor, kv, img, pixel_values=torch.zeros(1, 3, 2, 2),
                                  n_image_tokens=cfg.n_image_tokens)
    assert prof.profile_image_budgets() == 8          # everything meets a 10 s SLO -> the upper bound
    n_fill_calls = len([e for e in log if "input_ids" in e])
    assert prof.profile_token_budgets() >= 16 and len([e for e in log if "input_ids" in e]) > n_fill_calls
    for m in (kv, img):
        assert len(m.shared_cache.to_be_evicted) == m.n_blocks
