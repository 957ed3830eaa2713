"""hipGraph replay of decode-only fill batches inside the engine (the
reference's unfinished attempt is hydrainfer/model_runner/cuda_graph_model_runner.py:1-72).

A decode step of a 7B model is ~260 launches of 5-70 us kernels; issued eagerly the host cannot
keep ahead of the GPU.  Here the step's integer inputs live in ONE static device buffer

    [ input_ids | positions | new_cache_slots | kv_cu | cu_blocks_lens | block_tables ... ]

filled by one pinned H2D copy per step, and the forward over views of that buffer is captured
once per padded batch size.  Batches are padded to a multiple of `pad_to` rows; padding rows are
1-token sequences that write into a scratch block reserved from the pool, so they never touch a
live request.  The attention kernel reads each sequence's true length from kv_cu, so one graph
serves every context length (the captured max length only seeds the split heuristic).

One step of look-ahead: the token a request sampled in launch N is its input in launch N+1, and
only the DEVICE needs it for that.  The graph therefore starts by taking each row's input id either
from the host-written id or from `prev_tokens[src_row]` (the previous launch's samples, kept in a
static buffer the graph itself updates at its end), so the host can build and enqueue launch N+1
while launch N is still running and read N's tokens afterwards (`launch` / `fetch`).  The host work
of a step (scheduler, tables, staging) no longer leaves the GPU idle."""
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

from hydrainfer_amd import _lib, launch_plan
from hydrainfer_amd._lib import HydraHipError
from hydrainfer_amd._C.kernel.norm import StepHead
from hydrainfer_amd.layer.causal_attention import AttentionParameters, decode_rank_descriptor
from hydrainfer_amd.memory.kv_cache import KVCache
from hydrainfer_amd.model.llama import LanguageModelParameters


class GraphedDecoder:
    def __init__(self, language_model, kv_cache_block_manager, max_batch: int = 64,
                 max_blocks_per_seq: int = 256, pad_to: int = 4, executor: str = None):
        # how a captured step is replayed: "graph" = a hipGraph (default here), "plan" = a launch plan
        # (hydrainfer_amd/launch_plan.py: the step's launches issued in stream order by a native loop).  The plan costs
        # ~0.65 ms of HOST time per 7B step (170 launches x 3.8 us) where a graph launch costs ~0.1: irrelevant for a
        # GPU-bound decode loop (model/runner.py, where the plan is 0.5-1 % faster), but the engine's one thread also
        # schedules, stages and runs eager prefill steps — at 16 req/s Poisson the plan measured TPOT p50 7.7 ms and
        # TTFT p50 84 ms against 6.5 / 51 for the graph (tools/bench_engine.py, round 3)
        self.executor = executor or os.environ.get("HX_ENGINE_EXECUTOR", "graph")
        self.kernel_copies = os.environ.get("HX_ENGINE_KERNEL_COPIES", "1") == "1"      # (0: hipMemcpyAsync, for A/B runs)
        self.lm = language_model                       # LlavaLanguageModel
        self.model = language_model.language_model     # LlamaForCausalLM
        self.kv = kv_cache_block_manager
        self.dev = self.kv.device
        self.pad_to = pad_to
        self.max_batch = (max_batch + pad_to - 1) // pad_to * pad_to
        self.table_cap = self.max_batch * max_blocks_per_seq
        B = self.max_batch
        # ("rank": the step's rank descriptor, [ragged?] + the rows by decreasing context — the fused decode attention lays a
        # big ragged batch over the CUs in that order, csrc/attn_decode.hip RANKED; the host has the lengths, so it ranks)
        self.off = {"ids": 0, "pos": B, "slots": 2 * B, "src": 3 * B, "kv_cu": 4 * B, "cu_blocks": 5 * B + 1,
                    "rank": 6 * B + 2, "tables": 7 * B + 3}
        total = self.off["tables"] + self.table_cap
        self.static = torch.zeros(total, dtype=torch.int32, device=self.dev)
        # two pinned staging buffers used alternately, each with an event recorded behind its H2D copy:
        # with one step of look-ahead the host fills launch N+1 while launch N's copy may still be
        # queued (a vision encode or a prefill chunk ahead of it on the stream) — a buffer is
        # refilled only after the copy that last read it has run
        self.staging = [torch.zeros(total, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.stage = [t.numpy() for t in self.staging]
        self.copy_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.fills = 0
        self.q_cu = torch.arange(0, B + 1, dtype=torch.int32, device=self.dev)
        self.prev_tokens = torch.zeros(B, dtype=torch.int64, device=self.dev)   # the previous launch's samples
        self.host_tokens = [torch.zeros(B, dtype=torch.int64).pin_memory() for _ in range(2)]
        # word that is nonzero iff an in-kernel hand-over of that launch gave up waiting (csrc/gemm_xreg.hip): it
        # travels to the host with the launch's tokens and is checked in fetch()
        self.host_err = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.launch_has_err = {}
        self.events = [torch.cuda.Event(), torch.cuda.Event()]
        self.launches = 0                  # id of the most recent launch (1-based)
        self.poisoned_until = 0            # launches <= this id were enqueued behind a failed hand-over: fetch raises
        self.launch_rows = {}              # launch id -> number of live rows
        n_layers = self.model.shape.num_hidden_layers
        self.kv_caches = [KVCache.from_token_cache(self.kv.get_layer_cache(l)) for l in range(n_layers)]
        # scratch block for padding rows
        self.pad_cache = self.kv.allocate_virtual_cache()
        self.kv.realloc(self.pad_cache, 1)
        self.pad_block = self.pad_cache.block_table[0]
        self.graphs: Dict[Tuple[int, int], tuple] = {}

    def fits(self, n_seqs: int, n_blocks: int) -> bool:
        padded = (n_seqs + self.pad_to - 1) // self.pad_to * self.pad_to
        return padded <= self.max_batch and n_blocks + (padded - n_seqs) <= self.table_cap

    def _views(self, B: int):
        o, s = self.off, self.static
        return (s[o["ids"]:o["ids"] + B], s[o["pos"]:o["pos"] + B], s[o["slots"]:o["slots"] + B],
                s[o["kv_cu"]:o["kv_cu"] + B + 1], s[o["cu_blocks"]:o["cu_blocks"] + B + 1],
                s[o["tables"]:], s[o["src"]:o["src"] + B], s[o["rank"]:o["rank"] + B + 1])

    def _kv_bucket(self, B: int, kv_max: int) -> int:
        if B * self.model.shape.num_attention_heads >= 768:
            return 0                       # one split whatever the length (attn_decode.hip heuristic)
        b = 256
        while b < kv_max:
            b *= 2
        return b

    def _params(self, B: int, kv_max: int) -> LanguageModelParameters:
        ids, pos, slots, kv_cu, cu_blocks, tables, _, rank = self._views(B)
        attn = [AttentionParameters(kv_cache=kc, q_cu_seq_lens=self.q_cu[:B + 1], kv_cu_seq_lens=kv_cu,
                                    new_cache_slots=slots, block_tables=tables, cu_blocks_lens=cu_blocks,
                                    num_sequences=B, all_sequences_decode=True, q_max_seq_len=1,
                                    kv_max_seq_len=kv_max, decode_rank=rank) for kc in self.kv_caches]
        return LanguageModelParameters(attention_params=attn, all_sequences_decode=True)

    def _body(self, B: int, params):
        """One decode step as library launches only (+ the lm_head GEMM): recordable in a launch plan."""
        ids, pos = self._views(B)[:2]
        src = self._views(B)[6]
        lib = _lib.lib()
        if self.model.step_head_supported(B):
            # the look-ahead feed rides in the step's first launch (hx_decode_step_head: id = sample of row src[r] of the
            # previous launch when src[r] >= 0, else the host-written id) — one launch less per step
            params.step_head = StepHead(feed_src=src, feed_prev=self.prev_tokens)
            fed = ids
        else:
            params.step_head = None
            fed = torch.empty(B, dtype=torch.int64, device=self.dev)
            _lib.check(lib.hx_decode_feed_ids(fed.data_ptr(), ids.data_ptr(), src.data_ptr(), self.prev_tokens.data_ptr(), B,
                                              _lib.current_stream()), "decode_feed_ids")
        self.model.xreg_sync = None
        # the greedy sampler writes straight into prev_tokens: this launch's feed (above) has read it, the next
        # launch's feed and the D2H copy of the tokens read it after this launch, in stream order
        out = self.prev_tokens[:B]
        self.model.sample_out = out
        try:
            res = self.model(fed, pos, params)
        finally:
            self.model.sample_out = None
            params.step_head = None      # consumed: a later direct model(...) call with these params must not feed again
        if res.data_ptr() != out.data_ptr():
            launch_plan.host_op(lambda: out.copy_(res))
        # give-up words of the step's in-kernel hand-overs (norm-fused launches) -> one word
        err = None
        sync = self.model.xreg_sync
        if sync is not None:
            err = torch.empty(1, dtype=torch.int32, device=self.dev)
            _lib.check(lib.hx_collect_errors(err.data_ptr(), sync.data_ptr(), sync.numel() // sync.shape[-1],
                                             sync.shape[-1], 1, None, _lib.current_stream()), "collect_errors")
        return out, err

    def _capture(self, B: int, bucket: int):
        params = self._params(B, bucket if bucket else 4096)
        saved = self.prev_tokens.clone()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(2):             # warm-up outside capture: workspace growth, lazy init
                self.prev_tokens.copy_(saved)
                self._body(B, params)
            self.prev_tokens.copy_(saved)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        graph = None
        if self.executor == "plan":
            graph = launch_plan.LaunchPlan(self.dev)
            try:
                out, err = graph.capture(lambda: self._body(B, params))
            except launch_plan.PlanNotRecordable:
                graph = None       # a step with torch ops in it (library GEMMs, a shape off the hx fast path): hipGraph
                self.executor = "graph"
        if graph is None:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out, err = self._body(B, params)
        return graph, out, err

    def _fill(self, rows: List[Tuple[int, int, int, int, List[int]]], B: int) -> int:
        """rows: (token, position, slot, kv_len, block_table) per live sequence; a token < 0 means
        "the sample of row -(token + 1) of the previous launch"."""
        which = self.fills % 2
        self.fills += 1
        self.copy_done[which].synchronize()      # no-op until the buffer has been used once
        o, st = self.off, self.stage[which]
        n = len(rows)
        bs = self.kv.block_size
        pad = (0, 0, self.pad_block * bs, 1, [self.pad_block])
        max_pos = self.model.shape.max_position_embeddings
        if not all(0 <= r[1] < max_pos for r in rows):    # the RoPE / attention kernels index cos_sin unchecked
            raise HydraHipError(f"decode position outside the rotary table (max_position_embeddings = {max_pos})")
        vocab = self.model.shape.vocab_size
        if not all(r[0] < vocab for r in rows):           # hx_embed_rms_norm would clamp, torch.embedding raises
            raise HydraHipError(f"token id outside the vocabulary ({vocab})")
        rows = rows + [pad] * (B - n)
        st[o["ids"]:o["ids"] + B] = [max(r[0], 0) for r in rows]
        st[o["src"]:o["src"] + B] = [-(r[0] + 1) if r[0] < 0 else -1 for r in rows]
        st[o["pos"]:o["pos"] + B] = [r[1] for r in rows]
        st[o["slots"]:o["slots"] + B] = [r[2] for r in rows]
        st[o["kv_cu"]] = 0
        st[o["kv_cu"] + 1:o["kv_cu"] + B + 1] = np.cumsum([r[3] for r in rows])
        st[o["rank"]:o["rank"] + B + 1] = decode_rank_descriptor([r[3] for r in rows])
        lens = [len(r[4]) for r in rows]
        st[o["cu_blocks"]] = 0
        st[o["cu_blocks"] + 1:o["cu_blocks"] + B + 1] = np.cumsum(lens)
        flat = [b for r in rows for b in r[4]]
        st[o["tables"]:o["tables"] + len(flat)] = flat
        used = o["tables"] + len(flat)
        if self.kernel_copies:
            # the step's integers go in by a KERNEL that reads the pinned buffer (hx_copy_words2, include/hydra_hip.h): a
            # memcpy between two graph launches left the stream idle for ~0.1 ms per step
            _lib.check(_lib.lib().hx_copy_words2(self.static.data_ptr(), self.staging[which].data_ptr(), used, None, None, 0,
                                                 _lib.current_stream()), "copy_words2")
        else:
            self.static[:used].copy_(self.staging[which][:used], non_blocking=True)
        self.copy_done[which].record()
        return max(r[3] for r in rows)

    def warmup(self, batch_sizes: List[int], kv_max: int = 1024) -> None:
        rows = [(1, 0, self.pad_block * self.kv.block_size, 1, [self.pad_block])]
        for n in batch_sizes:
            B = (n + self.pad_to - 1) // self.pad_to * self.pad_to
            key = (B, self._kv_bucket(B, kv_max))
            if key not in self.graphs:
                self._fill(rows, B)
                self.graphs[key] = self._capture(*key)
        torch.cuda.synchronize(self.dev)

    def launch(self, rows: List[Tuple[int, int, int, int, List[int]]]) -> int:
        """Enqueue one decode step; returns a launch id for `fetch`.  Does not wait for the GPU."""
        n = len(rows)
        B = (n + self.pad_to - 1) // self.pad_to * self.pad_to
        kv_max = self._fill(rows, B)
        key = (B, self._kv_bucket(B, kv_max))
        if key not in self.graphs:
            self.graphs[key] = self._capture(*key)
        graph, out, err = self.graphs[key]
        graph.replay()
        self.launches += 1
        slot = self.launches % 2
        if self.kernel_copies:      # tokens (int64: 2 words each) and the give-up word leave by one launch, straight into pinned memory
            _lib.check(_lib.lib().hx_copy_words2(self.host_tokens[slot].data_ptr(), out.data_ptr(), 2 * n,
                                                 self.host_err[slot].data_ptr() if err is not None else None,
                                                 err.data_ptr() if err is not None else None, 1 if err is not None else 0,
                                                 _lib.current_stream()), "copy_words2")
        else:
            self.host_tokens[slot][:n].copy_(out[:n], non_blocking=True)
            if err is not None:
                self.host_err[slot].copy_(err.view(1), non_blocking=True)
        self.launch_has_err[self.launches] = err is not None
        self.events[slot].record()
        self.launch_rows[self.launches] = n
        return self.launches

    def fetch(self, launch_id: int) -> List[int]:
        """Tokens sampled by a launch (waits for it).  Only the two most recent launches are kept."""
        assert launch_id > self.launches - 2, "tokens of an older launch have been overwritten"
        slot = launch_id % 2
        self.events[slot].synchronize()
        n = self.launch_rows.pop(launch_id)
        if launch_id <= self.poisoned_until:
            self.launch_has_err.pop(launch_id, None)
            raise HydraHipError(f"decode launch {launch_id} was enqueued behind a launch whose in-kernel hand-over gave "
                                "up: its input tokens were invalid, so are its samples")
        if self.launch_has_err.pop(launch_id, False) and int(self.host_err[slot][0]) != 0:
            # a norm-fused launch consumed activations nobody had produced: this step's tokens are garbage.
            # Later steps run with the add+RMSNorm as separate launches (no in-kernel hand-over).
            self.model.fuse_norm = False
            # launch N+1 may already be running out of the same graph / plan, fed from this launch's garbage tokens:
            # let it finish before its buffers (the plan's private pool) go, and refuse its tokens as well
            torch.cuda.synchronize(self.dev)
            self.graphs.clear()
            self.poisoned_until = self.launches
            raise HydraHipError(f"decode launch {launch_id}: an in-kernel hand-over (norm-fused GEMM launch) gave up "
                                "waiting for its producer workgroups; the step's tokens are invalid. Later steps run "
                                "with the add+RMSNorm as separate launches (fuse_norm = False)")
        return self.host_tokens[slot][:n].tolist()

    def run(self, rows: List[Tuple[int, int, int, int, List[int]]]) -> List[int]:
        return self.fetch(self.launch(rows))
