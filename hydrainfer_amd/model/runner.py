"""Continuous-batching decode runner: drives LlamaForCausalLM through the reference's
execute_fill loop (hydrainfer/engine/executor.py:105-193) for a fixed batch of requests —
prefill (chunked under a token budget like scheduler.py:166-184), then greedy decode.

Reference behaviour kept: block ids come from the LIFO BlockAllocator in the order the
scheduler's per-step `realloc` would draw them (engine/scheduler.py:125-142 ->
memory/token_cache_manger.py:149-153); slots via v2p; per-layer KVCache views of one 6-D pool;
greedy argmax feeds the next step.

MI355X-first differences: the per-step Python metadata build + 6 H2D copies
+ `.tolist()` sync are replaced by device-resident metadata advanced by one tiny kernel
(hx_decode_advance), and the whole decode step (193 launches for 7B) is captured once into a
hipGraph and replayed; sampled tokens stay on the device until the end."""
import os
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import Tensor

from hydrainfer_amd import _lib, launch_plan
from hydrainfer_amd._C.kernel.norm import StepHead
from hydrainfer_amd.layer.causal_attention import AttentionParameters, AttentionParametersBuilder
from hydrainfer_amd.memory.block_allocator import BlockAllocator
from hydrainfer_amd.memory.kv_cache import KVCache
import hydrainfer_amd.memory.kv_pool as kv_pool
from hydrainfer_amd.model.llama import LanguageModelParameters, LlamaForCausalLM


def plan_block_tables(n_requests: int, prompt_len: int, n_generate: int, block_size: int,
                      n_blocks: int) -> List[List[int]]:
    """Block ids each request ends up with, drawn exactly as the reference scheduler would:
    every step, requests in order call realloc(current_len + new_tokens) which pops
    ceil(n/bs) - len(table) ids from the LIFO free list."""
    alloc = BlockAllocator(n_blocks)
    tables: List[List[int]] = [[] for _ in range(n_requests)]
    lens = [0] * n_requests

    def step(new_tokens: int) -> None:
        for r in range(n_requests):
            lens[r] += new_tokens
            need = (lens[r] + block_size - 1) // block_size - len(tables[r])
            got = alloc.allocate(need)
            assert len(got) == need, "KV pool too small for the planned run"
            tables[r] += got

    step(prompt_len)
    for _ in range(n_generate - 1):   # max_tokens - 1 decode steps (request_processor.py:152-166)
        step(1)
    return tables


def ragged_contexts(kind, B=32, seed=0):
    """The two ragged decode batches bench.py's `whole_step_ragged` times (BASELINE configs[2] "mixed image+text"):
    'uniform' = lengths drawn uniformly from 64..959; 'bimodal' = half text-only requests (~130 keys) and half image
    requests (~830 keys), in arrival (shuffled) order.  One definition for the benchmark and the tests."""
    g = torch.Generator().manual_seed(seed)
    if kind == "uniform":
        return torch.randint(64, 960, (B,), generator=g).tolist()
    assert kind == "bimodal"
    lens = [130 + int(j) for j in torch.randint(-8, 9, (B // 2,), generator=g)] + \
           [830 + int(j) for j in torch.randint(-8, 9, (B - B // 2,), generator=g)]
    return [lens[i] for i in torch.randperm(B, generator=g).tolist()]


@dataclass
class RunnerConfig:
    batch: int = 32
    prompt_len: int = 704
    n_generate: int = 256
    block_size: int = 16
    prefill_token_budget: int = 4096
    use_graph: bool = True
    # how a decode step is replayed when use_graph: "graph" = one captured hipGraph; "plan" = a launch plan
    # (hydrainfer_amd/launch_plan.py): the same launches, in the same stream order, issued by a native loop
    executor: str = "plan"
    advance_stride: int = 1      # tokens a decode step moves the contexts forward (1 = a real generation; bench.py
                                 # samples the generation's contexts at a fixed spacing when it times fewer steps)


class DecodeRunner:
    def __init__(self, model: LlamaForCausalLM, cfg: RunnerConfig, seed: int = 0):
        self.model, self.cfg = model, cfg
        sh, dev, dt = model.shape, model.device, model.dtype
        model.prepare_decode(max_rows=cfg.batch)      # packed decode layouts now, not inside the first step
        self.dev = dev
        B, bs = cfg.batch, cfg.block_size
        self.max_len = cfg.prompt_len + cfg.n_generate
        self.blocks_per_seq = (self.max_len - 1 + bs - 1) // bs  # last sampled token is never cached
        n_blocks = B * self.blocks_per_seq
        self.tables = plan_block_tables(B, cfg.prompt_len, cfg.n_generate, bs, n_blocks)
        # the pool may be mapped by a neighbour process (migration): keep its size out of the
        # window in which hipIpcOpenMemHandle was seen to hang (token_cache_manger.ipc_safe_n_blocks)
        from hydrainfer_amd.memory.token_cache_manger import ipc_safe_n_blocks
        n_blocks = ipc_safe_n_blocks(n_blocks, sh.num_hidden_layers * 2 * bs * sh.num_key_value_heads *
                                     sh.head_dim * torch.empty((), dtype=dt).element_size(),
                                     extra_bytes=sh.num_hidden_layers * 2 * kv_pool.KV_POOL_SKEW_BYTES)
        # the reference's 6-D pool (token_cache_manger.py:65) with its planes a few hundred bytes apart
        # (memory/kv_pool.py); randn = "garbage but finite", layer by layer: bounded fp32 temporaries
        g = torch.Generator(device=dev).manual_seed(seed + 1)
        self.pool = kv_pool.allocate_kv_pool((sh.num_hidden_layers, 2, n_blocks, bs, sh.num_key_value_heads, sh.head_dim),
                                             dt, dev, fill="randn", generator=g)
        self.kv_caches = [KVCache(self.pool[l, 0], self.pool[l, 1]) for l in range(sh.num_hidden_layers)]

        # device-resident decode metadata (capacity layout: each sequence owns a fixed slice of
        # the flat block table, so appending a block never moves another sequence's entries)
        i32 = dict(dtype=torch.int32, device=dev)
        flat = [b for t in self.tables for b in (t + [0] * (self.blocks_per_seq - len(t)))]
        self.block_table = torch.tensor(flat, **i32)
        self.cu_block_lens = torch.arange(0, (B + 1) * self.blocks_per_seq, self.blocks_per_seq, **i32)
        self.q_cu = torch.arange(0, B + 1, **i32)
        self.positions = torch.zeros(B, **i32)
        self.kv_lens = torch.zeros(B, **i32)
        self.cu_k = torch.zeros(B + 1, **i32)
        self.slots = torch.zeros(B, **i32)
        self.input_ids = torch.zeros(B, dtype=torch.int64, device=dev)
        # the batch's rank descriptor (attn_decode.hip, RANKED): rewritten by every step's metadata advance
        self.rank_desc = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        self.rank_desc[1:] = torch.arange(B, dtype=torch.int32, device=dev)
        self.decode_params = LanguageModelParameters(
            attention_params=[AttentionParameters(
                kv_cache=kc, q_cu_seq_lens=self.q_cu, kv_cu_seq_lens=self.cu_k,
                new_cache_slots=self.slots, block_tables=self.block_table,
                cu_blocks_lens=self.cu_block_lens, num_sequences=B, all_sequences_decode=True,
                q_max_seq_len=1, kv_max_seq_len=self.max_len, decode_rank=self.rank_desc) for kc in self.kv_caches],
            all_sequences_decode=True)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.executor_used = cfg.executor          # "plan" falls back to "graph" when the step is not recordable
        self.tokens: List[Tensor] = []

    # ------------------------------------------------------------------ prefill
    def prefill(self, prompt_ids: Tensor, image_features: Optional[Tensor] = None,
                image_token_id: int = 32000, requests: Optional[List[int]] = None) -> Tensor:
        """prompt_ids int64 [B, prompt_len] on the device.  Requests are packed into batches
        under the token budget; returns the first sampled token of every request [B].
        `requests` restricts the pass to a subset (decode state is then left untouched)."""
        cfg, sh, bs = self.cfg, self.model.shape, self.cfg.block_size
        B, P = cfg.batch, cfg.prompt_len
        per_batch = max(1, cfg.prefill_token_budget // P)
        first = torch.empty(B, dtype=torch.int64, device=self.dev)
        n_prompt_blocks = (P + bs - 1) // bs
        todo = list(range(B)) if requests is None else list(requests)
        for r0 in range(0, len(todo), per_batch):
            rs = todo[r0: r0 + per_batch]
            b = AttentionParametersBuilder(sh.num_attention_heads, sh.num_key_value_heads,
                                           sh.head_dim, bs, self.dev)
            for r in rs:
                t = self.tables[r][:n_prompt_blocks]
                slots = [t[p // bs] * bs + p % bs for p in range(P)]   # v2p
                b.add_request(P, P, slots, t)
            for kc in self.kv_caches:
                b.add_kv_cache(kc)
            ap = b.build_attention_parameters()
            ids = prompt_ids[rs].reshape(-1)
            embeds = self.model.embed(ids)
            if image_features is not None:   # llava.py:132-135: overwrite image-token rows
                mask = ids == image_token_id
                embeds[mask] = image_features[rs].reshape(-1, embeds.shape[-1]).to(embeds.dtype)
            pos = torch.arange(P, dtype=torch.int32, device=self.dev).repeat(len(rs))
            sel = torch.arange(P - 1, len(rs) * P, P, device=self.dev)
            params = LanguageModelParameters(attention_params=ap, all_sequences_decode=False,
                                             selected_token_ids=sel)
            first[rs] = self.model(embeds, pos, params)
        if requests is not None:
            return first
        self.positions.fill_(P - 1)
        self.kv_lens.fill_(P)
        self.input_ids.copy_(first)
        self.tokens = [first.clone()]
        return first

    def capture_prefill(self, request: int, image_token_id: int = 32000):
        """hipGraph of the prefill of ONE request (fixed prompt length, this request's blocks):
        the ~290 eager launches of a 704-token prefill cost several ms of host time, which is
        what bounds TTFT on an idle replica.  Returns (graph, static_ids, static_features,
        static_first_token)."""
        cfg, sh, bs = self.cfg, self.model.shape, self.cfg.block_size
        P = cfg.prompt_len
        n_prompt_blocks = (P + bs - 1) // bs
        b = AttentionParametersBuilder(sh.num_attention_heads, sh.num_key_value_heads, sh.head_dim, bs, self.dev)
        t = self.tables[request][:n_prompt_blocks]
        b.add_request(P, P, [t[p // bs] * bs + p % bs for p in range(P)], t)
        for kc in self.kv_caches:
            b.add_kv_cache(kc)
        params = LanguageModelParameters(attention_params=b.build_attention_parameters(),
                                         all_sequences_decode=False,
                                         selected_token_ids=torch.tensor([P - 1], device=self.dev))
        ids = torch.zeros(P, dtype=torch.int64, device=self.dev)
        feats = torch.zeros((576, sh.hidden_size), dtype=self.model.dtype, device=self.dev)
        pos = torch.arange(P, dtype=torch.int32, device=self.dev)
        first = torch.zeros(1, dtype=torch.int64, device=self.dev)
        n_img = 576

        def body():
            embeds = self.model.embed(ids)
            # image-token rows are the first 576 of the synthetic prompt (llava.py:132-135 with a
            # static mask, so the graph has no data-dependent shapes)
            embeds[:n_img] = feats
            first.copy_(self.model(embeds, pos, params))

        torch.cuda.synchronize(self.dev)
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            body()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        return graph, ids, feats, first

    def set_state(self, kv_len: int, input_ids: Optional[Tensor] = None) -> None:
        """Position the decode state at `kv_len` cached tokens per sequence without running a
        prefill (the cache then holds its randn fill — used by kernel-only measurements)."""
        self.positions.fill_(kv_len - 1)
        self.kv_lens.fill_(kv_len)
        if input_ids is not None:
            self.input_ids.copy_(input_ids)
        self.tokens = []

    def set_state_lens(self, kv_lens: List[int], input_ids: Optional[Tensor] = None) -> None:
        """set_state with one length per sequence (a ragged decode batch)."""
        assert len(kv_lens) == self.cfg.batch and max(kv_lens) + 1 <= self.max_len and min(kv_lens) >= 1
        lens = torch.tensor(kv_lens, dtype=torch.int32, device=self.dev)
        self.positions.copy_(lens - 1)
        self.kv_lens.copy_(lens)
        if input_ids is not None:
            self.input_ids.copy_(input_ids)
        self.tokens = []

    # ------------------------------------------------------------------ decode
    def _advance(self) -> None:
        _lib.check(_lib.lib().hx_decode_advance_ranked(
            self.positions.data_ptr(), self.kv_lens.data_ptr(), self.cu_k.data_ptr(),
            self.slots.data_ptr(), self.block_table.data_ptr(), self.cu_block_lens.data_ptr(),
            self.cfg.batch, self.cfg.block_size, self.cfg.advance_stride, self.rank_desc.data_ptr(),
            _lib.current_stream()), "decode_advance")

    def _step_body(self) -> None:
        # the metadata advance rides in the step's first launch (hx_decode_step_head) when the model can take it
        if self.model.step_head_supported(self.cfg.batch):
            self.decode_params.step_head = StepHead(
                positions=self.positions, kv_lens=self.kv_lens, cu_seqlens_k=self.cu_k, new_cache_slots=self.slots,
                block_table=self.block_table, cu_block_lens=self.cu_block_lens, batch=self.cfg.batch,
                block_size=self.cfg.block_size, stride=self.cfg.advance_stride, rank_desc=self.rank_desc)
        else:
            self.decode_params.step_head = None
            self._advance()
        self.model.sample_out = self.input_ids     # the sampled ids are the next step's input ids: no copy launch
        try:
            nxt = self.model(self.input_ids, self.positions, self.decode_params)
        finally:
            self.model.sample_out = None
            # consumed by the step's first launch (and recorded in the graph / plan): a later direct model(...) call with
            # these shared params must not advance positions / kv_lens / slots a second time (round-4 ADVICE)
            self.decode_params.step_head = None
        if nxt.data_ptr() != self.input_ids.data_ptr():
            launch_plan.host_op(lambda: self.input_ids.copy_(nxt))

    def capture(self) -> None:
        """Warm up on a side stream, then capture one decode step into a hipGraph.  Decode
        state is restored afterwards (warm-up steps only touch cache slots that the real run
        rewrites before reading)."""
        saved = (self.positions.clone(), self.kv_lens.clone(), self.input_ids.clone())
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                self._step_body()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        self.positions.copy_(saved[0]); self.kv_lens.copy_(saved[1]); self.input_ids.copy_(saved[2])
        self.executor_used = self.cfg.executor
        if self.cfg.executor == "plan":
            plan = launch_plan.LaunchPlan(self.dev)
            try:
                plan.capture(self._step_body)       # records, runs nothing: the decode state is untouched
                self.graph = plan
            except launch_plan.PlanNotRecordable:
                # the step is not all-hx (library GEMMs: use_hip_gemm = False, a shape off the fast path, fp32 ...):
                # a plan would drop those torch ops on replay — the captured hipGraph records them
                self.executor_used = "graph"
        if self.graph is None:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._step_body()
            self.graph = graph
            self.positions.copy_(saved[0]); self.kv_lens.copy_(saved[1]); self.input_ids.copy_(saved[2])
        torch.cuda.synchronize(self.dev)

    def step(self, record: bool = True) -> None:
        if self.cfg.use_graph:
            if self.graph is None:
                self.capture()
            self.graph.replay()
        else:
            self._step_body()
        if record:
            self.tokens.append(self.input_ids.clone())

    def generated(self) -> Tensor:
        """[n_steps_so_far, B] sampled tokens (one D2H sync, at the end).  Raises if an in-kernel hand-over of the
        last step gave up (its tokens would be garbage)."""
        if self.model.handover_failed():
            self.model.fuse_norm = False
            self.graph = None
            raise _lib.HydraHipError("a norm-fused GEMM launch gave up waiting for its producer workgroups: the tokens "
                                     "of this run are invalid (norm fusion now disabled for this model)")
        return torch.stack(self.tokens).cpu()

    # ------------------------------------------------------------------ accounting
    def step_bytes(self, ctx_total: int) -> int:
        """Algorithmic HBM bytes of one decode step:
        W + e*2*L*HK*D*(sum ctx + B) + activations."""
        sh, e = self.model.shape, self.pool.element_size()
        per_tok = 2 * sh.num_hidden_layers * sh.num_key_value_heads * sh.head_dim * e
        act = e * self.cfg.batch * sh.hidden_size * 12 * sh.num_hidden_layers
        return self.model.weight_bytes() + per_tok * (ctx_total + self.cfg.batch) + act

    def attention_bytes(self, ctxs: List[int]) -> int:
        """Algorithmic bytes of ONE decode-attention launch (one layer), SURVEY.md §8d:
        e*[2*HK*D*sum ctx + 2*B*H*D] + 4*sum ceil(ctx/16)."""
        sh, e, bs = self.model.shape, self.pool.element_size(), self.cfg.block_size
        return (e * (2 * sh.num_key_value_heads * sh.head_dim * sum(ctxs) +
                     2 * len(ctxs) * sh.num_attention_heads * sh.head_dim) +
                4 * sum((c + bs - 1) // bs for c in ctxs))
