"""hipGraph replay of decode-only fill batches inside the engine (the
reference's unfinished attempt is hydrainfer/model_runner/cuda_graph_model_runner.py:1-72).

A decode step of a 7B model is ~260 launches of 5-70 us kernels; issued eagerly the host cannot
keep ahead of the GPU.  Here the step's integer inputs live in ONE static device buffer

    [ input_ids | positions | new_cache_slots | kv_cu | cu_blocks_lens | block_tables ... ]

and the forward over views of that buffer is captured once per padded batch size.  The block tables
are RESIDENT (SURVEY.md §8(f) rank 1 as written: "a persistent device-side block-table buffer with
incremental append"): every live sequence owns a fixed slice of the table region, a step stages its
head (ids, positions, slots, cumulative lengths, table offsets, rank descriptor — 7 words a row) and
only the block ids that are NEW since the sequence's last step (DecodeStager), and one kernel
(hx_stage_decode) applies both from pinned memory.  The reference rebuilds every table and copies
six tensors per step (hydrainfer/engine/parameters_builder.py:46-97, layer/causal_attention.py:147-168).  Batches are padded to a multiple of `pad_to` rows; padding rows are
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
from hydrainfer_amd.layer.causal_attention import RANKED_THR, AttentionParameters, decode_rank_descriptor
from hydrainfer_amd.memory.kv_cache import KVCache
from hydrainfer_amd.model.llama import LanguageModelParameters


class DecodeStager:
    """The host side of a decode step's integer inputs — numpy only, no device (tests/test_decode_stager.py drives it on
    the CPU).  Layout of the static device buffer it fills (word offsets; B = max_batch, cap = blocks per sequence):

        [ ids B | pos B | slots B | src B | kv_cu B+1 | cu_blocks B+1 | rank B+1 | tables (2B + 1) x cap ]

    cu_blocks[r] = the word offset of row r's table INSIDE the tables region (the kernels only ever use it as a start
    offset).  Table slot 0 is the padding rows' ([pad_block]); a sequence (keyed by the scheduler's sid) keeps its slot
    from its first decode step until it has not been seen while the slots ran out (least recently seen goes first).
    stage() writes one staging buffer: the head, then [n_runs] and the runs ([dst word offset] [count] [values]) that bring
    the resident tables up to date — hx_stage_decode's input."""

    def __init__(self, max_batch: int, cap: int, block_size: int, pad_block: int, max_pos: int, vocab: int):
        B = self.max_batch = max_batch
        self.cap, self.block_size, self.pad_block, self.max_pos, self.vocab = cap, block_size, pad_block, max_pos, vocab
        self.off = {"ids": 0, "pos": B, "slots": 2 * B, "src": 3 * B, "kv_cu": 4 * B, "cu_blocks": 5 * B + 1, "rank": 6 * B + 2}
        self.head_words = 7 * B + 3
        self.n_table_slots = 2 * B + 1
        self.tables_off = self.head_words
        self.total_words = self.head_words + self.n_table_slots * cap
        self.staging_words = self.head_words + 1 + (B + 1) * (2 + cap)
        self.slot_of: Dict[int, list] = {}        # sid -> [table slot, blocks written, last block id written, last step seen]
        self.free = list(range(self.n_table_slots - 1, 0, -1))
        self.pad_written = False
        self.step_no = 0
        self._arange = np.arange(max_batch, dtype=np.int32)

    def _alloc(self) -> int:
        if self.free:
            return self.free.pop()
        victim = min((sid for sid, e in self.slot_of.items() if e[3] != self.step_no), key=lambda sid: self.slot_of[sid][3])
        return self.slot_of.pop(victim)[0]

    def stage(self, st: "np.ndarray", rows: List[tuple], B: int) -> int:
        """rows: (token, position, slot, kv_len, block_table, sid) per live sequence (sid None: a padding / warm-up row);
        a token < 0 means "the sample of row -(token + 1) of the previous launch".  Pads to B rows.  Returns the largest
        kv_len."""
        self.step_no += 1
        o, cap, n = self.off, self.cap, len(rows)
        if not all(0 <= r[1] < self.max_pos for r in rows):    # the RoPE / attention kernels index cos_sin unchecked
            raise HydraHipError(f"decode position outside the rotary table (max_position_embeddings = {self.max_pos})")
        if not all(r[0] < self.vocab for r in rows):           # hx_embed_rms_norm would clamp, torch.embedding raises
            raise HydraHipError(f"token id outside the vocabulary ({self.vocab})")
        pad = B - n
        toks = [r[0] for r in rows]
        st[o["ids"]:o["ids"] + B] = [t if t > 0 else 0 for t in toks] + [0] * pad
        st[o["src"]:o["src"] + B] = [-(t + 1) if t < 0 else -1 for t in toks] + [-1] * pad
        st[o["pos"]:o["pos"] + B] = [r[1] for r in rows] + [0] * pad
        st[o["slots"]:o["slots"] + B] = [r[2] for r in rows] + [self.pad_block * self.block_size] * pad
        kv = [r[3] for r in rows] + [1] * pad
        st[o["kv_cu"]] = 0
        np.cumsum(kv, out=st[o["kv_cu"] + 1:o["kv_cu"] + B + 1])
        st[o["rank"]:o["rank"] + B + 1] = decode_rank_descriptor(kv)
        at = self.head_words + 1
        n_runs = 0
        if not self.pad_written:
            st[at:at + 3] = (self.tables_off, 1, self.pad_block)
            at += 3
            n_runs += 1
            self.pad_written = True
        starts = []
        for i, r in enumerate(rows):
            tbl = r[4]
            if len(r) > 5:
                sid, anonymous = r[5], False
                if sid is None:
                    starts.append(0)
                    continue
            else:                              # a caller that does not name its sequences: row i's table is written in full
                sid, anonymous = ("row", i), True
            nb = len(tbl)
            if nb > cap:
                raise HydraHipError(f"a sequence of {nb} blocks in a decoder built for {cap}")
            e = self.slot_of.get(sid)
            if e is None:
                e = self.slot_of[sid] = [self._alloc(), 0, -1, self.step_no]
            e[3] = self.step_no
            if anonymous or nb < e[1] or (e[1] and tbl[e[1] - 1] != e[2]):
                e[1] = 0                       # not the table this slot holds a prefix of: written again in full
            if nb > e[1]:
                new = tbl[e[1]:]
                st[at] = self.tables_off + e[0] * cap + e[1]
                st[at + 1] = len(new)
                st[at + 2:at + 2 + len(new)] = new
                at += 2 + len(new)
                n_runs += 1
                e[1], e[2] = nb, tbl[-1]
            starts.append(e[0] * cap)
        st[o["cu_blocks"]:o["cu_blocks"] + B] = starts + [0] * pad
        st[o["cu_blocks"] + B] = 0
        st[self.head_words] = n_runs
        return max(kv)


    def stage_cohort(self, st: "np.ndarray", n: int, B: int, pos: "np.ndarray", slots: "np.ndarray", starts: "np.ndarray",
                     grown: List[Tuple[int, int, int]]) -> int:
        """The steady-state form of stage(): the SAME n sequences as the previous launch, in the same order, one token
        further — every row's input id is the previous launch's sample of the same row (src[r] = r), positions / slots /
        table offsets come as arrays, and `grown` lists the rows that entered a new block: (sid, blocks before, new
        block id).  No per-row Python except for those."""
        self.step_no += 1
        o, pad = self.off, B - n
        if int(pos[:n].max()) >= self.max_pos:
            raise HydraHipError(f"decode position outside the rotary table (max_position_embeddings = {self.max_pos})")
        st[o["ids"]:o["ids"] + B] = 0
        st[o["src"]:o["src"] + n] = self._arange[:n]
        st[o["pos"]:o["pos"] + n] = pos[:n]
        st[o["slots"]:o["slots"] + n] = slots[:n]
        st[o["cu_blocks"]:o["cu_blocks"] + n] = starts[:n]
        if pad:
            st[o["src"] + n:o["src"] + B] = -1
            st[o["pos"] + n:o["pos"] + B] = 0
            st[o["slots"] + n:o["slots"] + B] = self.pad_block * self.block_size
            st[o["cu_blocks"] + n:o["cu_blocks"] + B] = 0
        st[o["cu_blocks"] + B] = 0
        kv = st[o["kv_cu"] + 1:o["kv_cu"] + B + 1]
        kv[:n] = pos[:n]
        kv[:n] += 1
        kv[n:] = 1
        # the rank descriptor (layer/causal_attention.py::decode_rank_descriptor, the same words) without a Python loop
        kv_max = int(kv.max())
        if B <= 256:
            st[o["rank"]] = 1 if float(kv_max) > float(kv.sum()) / B * RANKED_THR + 16.0 else 0
            st[o["rank"] + 1:o["rank"] + B + 1] = np.argsort(-kv, kind="stable")
        else:
            st[o["rank"]] = 0
            st[o["rank"] + 1:o["rank"] + B + 1] = self._arange_b(B)
        st[o["kv_cu"]] = 0
        np.cumsum(kv, out=kv)
        at = self.head_words + 1
        for sid, before, block in grown:
            e = self.slot_of[sid]
            st[at:at + 3] = (self.tables_off + e[0] * self.cap + before, 1, block)
            at += 3
            e[1], e[2] = before + 1, block
        st[self.head_words] = len(grown)
        return kv_max

    def _arange_b(self, B: int):
        return np.arange(B, dtype=np.int32)

    def touch(self, sids) -> None:
        """The sequences of a cohort episode were seen until now (least-recently-seen bookkeeping)."""
        for sid in sids:
            e = self.slot_of.get(sid)
            if e is not None:
                e[3] = self.step_no


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
        self.lm = language_model                       # LlavaLanguageModel
        self.model = language_model.language_model     # LlamaForCausalLM
        self.kv = kv_cache_block_manager
        self.dev = self.kv.device
        self.pad_to = pad_to
        self.max_batch = (max_batch + pad_to - 1) // pad_to * pad_to
        self.cap = max_blocks_per_seq
        B = self.max_batch
        # scratch block for padding rows
        self.pad_cache = self.kv.allocate_virtual_cache()
        self.kv.realloc(self.pad_cache, 1)
        self.pad_block = self.pad_cache.block_table[0]
        self.stager = DecodeStager(B, self.cap, self.kv.block_size, self.pad_block, self.model.shape.max_position_embeddings,
                                   self.model.shape.vocab_size)
        self.off = self.stager.off
        self.static = torch.zeros(self.stager.total_words, dtype=torch.int32, device=self.dev)
        # two pinned staging buffers used alternately, each with an event recorded behind the kernel that reads it:
        # with one step of look-ahead the host fills launch N+1 while launch N's staging may still be
        # queued (a vision encode or a prefill chunk ahead of it on the stream) — a buffer is
        # refilled only after the launch that last read it has run
        self.staging = [torch.zeros(self.stager.staging_words, dtype=torch.int32).pin_memory() for _ in range(2)]
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
        self.graphs: Dict[Tuple[int, int], tuple] = {}

    def fits(self, n_seqs: int, max_blocks: int) -> bool:
        """n_seqs rows whose longest block table has max_blocks entries."""
        padded = (n_seqs + self.pad_to - 1) // self.pad_to * self.pad_to
        return padded <= self.max_batch and max_blocks <= self.cap

    def _views(self, B: int):
        o, s = self.off, self.static
        return (s[o["ids"]:o["ids"] + B], s[o["pos"]:o["pos"] + B], s[o["slots"]:o["slots"] + B],
                s[o["kv_cu"]:o["kv_cu"] + B + 1], s[o["cu_blocks"]:o["cu_blocks"] + B + 1],
                s[self.stager.tables_off:], s[o["src"]:o["src"] + B], s[o["rank"]:o["rank"] + B + 1])

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

    def _fill(self, rows: List[tuple], B: int) -> int:
        """Stage the step (DecodeStager.stage) and enqueue the kernel that applies it to the resident buffer."""
        which = self.fills % 2
        self.fills += 1
        self.copy_done[which].synchronize()      # no-op until the buffer has been used once
        kv_max = self.stager.stage(self.stage[which], rows, B)
        # (a kernel reading the pinned buffer, not a memcpy: a memcpy between two graph launches left the stream idle
        # for ~0.1 ms per step)
        _lib.check(_lib.lib().hx_stage_decode(self.static.data_ptr(), self.static.numel(), self.staging[which].data_ptr(),
                                              self.stager.head_words, _lib.current_stream()), "stage_decode")
        self.copy_done[which].record()
        return kv_max

    def warmup(self, batch_sizes: List[int], kv_max: int = 1024) -> None:
        rows = [(1, 0, self.pad_block * self.kv.block_size, 1, [self.pad_block], None)]
        for n in batch_sizes:
            B = (n + self.pad_to - 1) // self.pad_to * self.pad_to
            key = (B, self._kv_bucket(B, kv_max))
            if key not in self.graphs:
                self._fill(rows, B)
                self.graphs[key] = self._capture(*key)
        torch.cuda.synchronize(self.dev)

    def launch_cohort(self, n: int, pos, slots, starts, grown) -> int:
        """launch() for the same n sequences as the previous launch, one token further (DecodeStager.stage_cohort)."""
        B = (n + self.pad_to - 1) // self.pad_to * self.pad_to
        which = self.fills % 2
        self.fills += 1
        self.copy_done[which].synchronize()
        kv_max = self.stager.stage_cohort(self.stage[which], n, B, pos, slots, starts, grown)
        _lib.check(_lib.lib().hx_stage_decode(self.static.data_ptr(), self.static.numel(), self.staging[which].data_ptr(),
                                              self.stager.head_words, _lib.current_stream()), "stage_decode")
        self.copy_done[which].record()
        return self._replay(n, B, kv_max)

    def launch(self, rows: List[tuple]) -> int:
        """Enqueue one decode step; returns a launch id for `fetch`.  Does not wait for the GPU."""
        n = len(rows)
        B = (n + self.pad_to - 1) // self.pad_to * self.pad_to
        return self._replay(n, B, self._fill(rows, B))

    def _replay(self, n: int, B: int, kv_max: int) -> int:
        key = (B, self._kv_bucket(B, kv_max))
        if key not in self.graphs:
            self.graphs[key] = self._capture(*key)
        graph, out, err = self.graphs[key]
        graph.replay()
        self.launches += 1
        slot = self.launches % 2
        # tokens (int64: 2 words each) and the give-up word leave by one launch, straight into pinned memory
        _lib.check(_lib.lib().hx_copy_words2(self.host_tokens[slot].data_ptr(), out.data_ptr(), 2 * n,
                                             self.host_err[slot].data_ptr() if err is not None else None,
                                             err.data_ptr() if err is not None else None, 1 if err is not None else 0,
                                             _lib.current_stream()), "copy_words2")
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
