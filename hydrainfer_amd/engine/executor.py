"""Instruction executors — mirror of hydrainfer/engine/executor.py:82-299.

BatchFillExecutor: publish finished blocks to the prefix cache, build the step's inputs, run the
language model on the HIP path, hand each sampled token to its request.
BatchImageEmbedExecutor: run the vision tower + projector and scatter the embeddings into the
image cache.  Both run on the current stream; `InstructionExecutor` can put the vision side on
its own stream (executor.py:247-249)."""
import os
import time
from typing import List, Optional

import numpy as np
import torch

from hydrainfer_amd._lib import HydraHipError

from hydrainfer_amd.engine import rcb as rcb_module
from hydrainfer_amd.engine.isa import Fill, TextFill
from hydrainfer_amd.engine.parameters_builder import LanguageModelParametersBuilder
from hydrainfer_amd.engine.rcb import BatchRequest
from hydrainfer_amd.model.llama import LanguageModelParameters


class PendingToken:
    """Placeholder for a token that has been sampled on the device but not read back yet
    (graph decode look-ahead): row `row` of decode launch `launch`."""
    __slots__ = ("launch", "row")

    def __init__(self, launch: int, row: int):
        self.launch, self.row = launch, row

    def __repr__(self):
        return f"<pending {self.launch}:{self.row}>"


class DecodeCohort:
    """The steady state of a decode batch — the SAME sequences step after step, each one token further — kept as arrays
    instead of being rediscovered from every request's instruction chain every step (the
    reference's per-step Python over every running request, hydrainfer/engine/scheduler.py:99-194 +
    parameters_builder.py:46-97, is what bounds a decode loop once the kernels are fast).  While the cohort lasts a step
    touches a request object only when it enters a new block or streams a token to a client; when it ends (anything
    else wants to happen: an arrival, a request about to finish, a cancelled stream) the requests are brought up to date
    in one pass and the general path goes on as if it had run every step itself."""

    def __init__(self, rcbs, running_ref, bs):
        n = self.n = len(rcbs)
        self.rcbs, self.running_ref, self.bs = rcbs, running_ref, bs
        self.pos = np.zeros(n, dtype=np.int32)          # rotary position of the NEXT launch's token, per row
        self.cid = np.zeros(n, dtype=np.int32)          # its virtual cache id
        self.left = np.zeros(n, dtype=np.int32)         # tokens the row has still to sample, the next launch's included
        self.nblk = np.zeros(n, dtype=np.int32)         # blocks in the row's table
        self.last_block = np.zeros(n, dtype=np.int32)   # the block the next cache id falls into (valid when cid // bs < nblk)
        self.starts = np.zeros(n, dtype=np.int32)       # the row's table offset in the decoder's resident buffer
        self.sids = [r.sid for r in rcbs]
        self.streams = [(i, r.output_token_processors) for i, r in enumerate(rcbs) if r.output_token_processors]
        self.first_inst = [r.current_instruction() for r in rcbs]
        self.eos_rows: List[int] = []                   # rows whose request names end-of-sequence ids
        self.k = 0                                      # cohort launches so far
        self.launch = None                              # the one in flight
        self.tok_log: List[List[int]] = []              # tokens of cohort launches 1 .. k - 1 (resolved)
        self.t_log: List[float] = []
        self.epoch = rcb_module.MUTATIONS[0]


class BatchFillExecutor:
    def __init__(self, language_model, kv_cache_block_manager, image_cache_block_manager,
                 dtype: torch.dtype, device: torch.device, graph_decoder=None):
        self.language_model = language_model            # LlavaLanguageModel
        lm = language_model.language_model
        self.shape = lm.shape
        self.kv_manager, self.image_manager = kv_cache_block_manager, image_cache_block_manager
        self.dtype, self.device = dtype, device
        self.graph_decoder = graph_decoder              # engine.graph_decode.GraphedDecoder or None
        self.pending = None        # (launch id, [(rcb, inst, index into rcb.output_token_ids)])
        self.cohort: Optional[DecodeCohort] = None
        self.cohort_enabled = os.environ.get("HX_DECODE_COHORT", "1") == "1"
        self.cohort_refused = None # the pending launch a cohort could not be formed behind (not tried again until the next one)
        self.n_cohort_steps = 0
        self.ended_rows: Optional[List] = None   # the rows of a cohort that ended inside its own step (an end-of-sequence id came back)

    def _decode_rows(self, batch: BatchRequest):
        """(token, position, slot, kv_len, block_table) per request if the whole batch is decode.
        A token still on the device is encoded as -(row + 1) of the pending launch."""
        rows, max_blocks = [], 0
        bs = self.kv_manager.block_size
        pending_launch = self.pending[0] if self.pending is not None else None
        for rcb, inst in batch:
            if len(inst.token_ids) != 1 or not inst.sample:
                return None
            token = inst.token_ids[0]
            if isinstance(token, PendingToken):
                if token.launch != pending_launch:
                    return None
                token = -(token.row + 1)
            bt = rcb.virtual_kv_cache.block_table
            c = inst.cache_ids[0]
            rows.append((token, inst.position_ids[0], bt[c // bs] * bs + c % bs, c + 1, bt, rcb.sid))      # (v2p inline)
            if len(bt) > max_blocks:
                max_blocks = len(bt)
        return rows if self.graph_decoder.fits(len(rows), max_blocks) else None

    def _launch_decode(self, batch: BatchRequest, rows) -> None:
        """Enqueue the step, hand every request a placeholder for its new token, THEN read the
        previous step's tokens: the GPU already has the next step queued while the host catches up."""
        launch = self.graph_decoder.launch(rows)
        entries = []
        for row, (rcb, inst) in enumerate(batch):
            placeholder = PendingToken(launch, row)
            entries.append((rcb, inst, len(rcb.output_token_ids), placeholder))
            if rcb.eos_hit:
                continue                       # ended by a late-read token; this row's sample is dropped
            rcb.output_token_ids.append(placeholder)
            if inst.sample_dst is not None:
                inst.sample_dst.token_ids = [placeholder]
        batch.step()
        previous, self.pending = self.pending, (launch, entries)
        if previous is not None:
            self._resolve(previous)

    def _resolve(self, pending) -> None:
        launch, entries = pending
        tokens = self.graph_decoder.fetch(launch)
        now = time.perf_counter()
        for (rcb, inst, index, placeholder), token in zip(entries, tokens):
            if rcb.eos_hit:
                continue                       # a step that ran past an end-of-sequence token
            rcb.output_token_ids[index] = token
            dst = inst.sample_dst
            if dst is not None and dst.token_ids and dst.token_ids[0] is placeholder:
                dst.token_ids = [token]
            rcb.metric.token_times.append(now)
            if token in rcb.sampling_params.eos_token_ids:
                del rcb.output_token_ids[index + 1:]   # whatever ran past it is discarded
                rcb.eos_hit = True
            last = rcb.eos_hit or index + 1 == rcb.sampling_params.max_tokens
            for p in rcb.output_token_processors:
                p.append_token_id(token, last)

    # ------------------------------------------------------------------ the steady-state cohort
    def cohort_step(self, scheduler) -> int:
        """Called by EPDNode.step() in front of the scheduler.  Launches the next decode step of the cohort and returns
        its row count — or ends / declines the cohort and returns 0: the general path then runs the step."""
        dec = self.graph_decoder
        if not self.cohort_enabled or dec is None or not hasattr(dec, "launch_cohort"):
            return 0
        co = self.cohort
        if co is None:
            if self.pending is None or self.pending[0] == self.cohort_refused:
                return 0
            co = self._cohort_begin(scheduler)
            if co is None:
                self.cohort_refused = self.pending[0]
                return 0
        elif (scheduler.waiting or scheduler.running is not co.running_ref or len(scheduler.running) != co.n
              or co.epoch != rcb_module.MUTATIONS[0] or int(co.left.min()) < 2 or co.n > scheduler.token_budgets):
            self._cohort_end()
            return 0
        bs = co.bs
        # rows whose next token opens a new block: the only per-row work of a step (one row in sixteen)
        grown = []
        need = np.nonzero(co.cid // bs >= co.nblk)[0]
        if len(need):
            if len(need) > len(self.kv_manager.shared_cache.to_be_evicted):
                self._cohort_end()              # the pool cannot give every row its block now: the scheduler's business
                return 0
            for r in need.tolist():
                vc = co.rcbs[r].virtual_kv_cache
                self.kv_manager.realloc(vc, int(co.cid[r]) + 1)
                grown.append((co.sids[r], int(co.nblk[r]), vc.block_table[-1]))
                co.nblk[r] += 1
                co.last_block[r] = vc.block_table[-1]
        slots = co.last_block * bs + co.cid % bs
        launch = dec.launch_cohort(co.n, co.pos, slots, co.starts, grown)
        previous, co.launch = co.launch, launch
        co.k += 1
        co.pos += 1
        co.cid += 1
        co.left -= 1
        self.n_cohort_steps += 1
        if previous is None:                    # the launch the cohort was formed behind: the general bookkeeping
            pending, self.pending = self.pending, None
            self._resolve(pending)
            if co.eos_rows and any(co.rcbs[r].eos_hit for r in co.eos_rows):
                self._cohort_end()              # one of them has just ended: over before it began
        else:
            tokens = dec.fetch(previous)
            co.tok_log.append(tokens)
            co.t_log.append(time.perf_counter())
            ended = [r for r in co.eos_rows if tokens[r] in co.rcbs[r].sampling_params.eos_token_ids] if co.eos_rows else ()
            for r, processors in co.streams:    # tokens go out as they come (a request's last one only as an end-of-sequence id)
                for p in processors:
                    p.append_token_id(tokens[r], r in ended)
            if ended:
                # an end-of-sequence id, read one step late like every token: the launch just made ran past it for those
                # rows — their extra sample is dropped, the general path frees them at its next step (as it would have)
                self._cohort_end()
                for r in ended:
                    rcb = co.rcbs[r]
                    del rcb.output_token_ids[-1:]          # the placeholder of the launch that ran past the end
                    rcb.eos_hit = True
        if self.cohort is None:                 # it ended inside this step: the node looks at every row the way it does after any step
            self.ended_rows = co.rcbs
        return co.n

    def _cohort_begin(self, scheduler) -> Optional[DecodeCohort]:
        launch, entries = self.pending
        rcbs = [e[0] for e in entries]
        if scheduler.waiting or scheduler.running != rcbs or len(rcbs) > scheduler.token_budgets:
            return None
        bs = self.kv_manager.block_size
        co = DecodeCohort(rcbs, scheduler.running, bs)
        slot_of, cap = self.graph_decoder.stager.slot_of, self.graph_decoder.stager.cap
        for r, (rcb, inst) in enumerate(zip(rcbs, co.first_inst)):
            tok = inst.token_ids[0] if isinstance(inst, TextFill) and inst.token_ids and len(inst.token_ids) == 1 else None
            if (not isinstance(tok, PendingToken) or tok.launch != launch or tok.row != r or not inst.sample or rcb.eos_hit
                    or rcb.sid not in slot_of):
                return None
            if rcb.sampling_params.eos_token_ids:
                co.eos_rows.append(r)
            vc = rcb.virtual_kv_cache
            c = inst.cache_ids[0]
            co.pos[r], co.cid[r] = inst.position_ids[0], c
            co.left[r] = rcb.sampling_params.max_tokens - len(rcb.output_token_ids)
            co.nblk[r] = len(vc.block_table)
            co.last_block[r] = vc.block_table[c // bs] if c // bs < len(vc.block_table) else -1
            co.starts[r] = slot_of[rcb.sid][0] * cap
        if int(co.left.min()) < 2:
            return None
        self.cohort = co
        return co

    def _cohort_end(self) -> None:
        """Bring every request of the cohort up to date: tokens, stamps, instruction chain, cache size — and leave the
        last launch pending the way the general path would have."""
        co, self.cohort = self.cohort, None
        if co is None or co.k == 0:
            return
        entries = []
        for r, rcb in enumerate(co.rcbs):
            out = rcb.output_token_ids
            placeholder = PendingToken(co.launch, r)
            if not rcb.eos_hit:                 # (a request that ended at the cohort's first resolve keeps what it has)
                out.extend(t[r] for t in co.tok_log)
                out.append(placeholder)
                rcb.metric.token_times.extend(co.t_log)
            inst = co.first_inst[r]
            for _ in range(co.k - 1):
                inst = inst.next                # the instructions of the launches that have been resolved
            if inst.sample_dst is not None:
                inst.sample_dst.token_ids = [placeholder]
            rcb.instructions.curr = inst.next
            vc = rcb.virtual_kv_cache
            if int(co.cid[r]) > vc.n_cache_tokens:
                vc.n_cache_tokens = int(co.cid[r])
            entries.append((rcb, inst, len(out) - 1, placeholder))
        self.pending = (co.launch, entries)
        self.graph_decoder.stager.touch(co.sids)

    def resolve_pending(self) -> None:
        """Read back the tokens of the decode step that is still in flight (if any)."""
        if self.cohort is not None:
            self._cohort_end()
        if self.pending is not None:
            pending, self.pending = self.pending, None
            self._resolve(pending)

    def _deliver(self, batch: BatchRequest, sampled: List[int]) -> None:
        now = time.perf_counter()
        i = 0
        for rcb, inst in batch:
            if not isinstance(inst, Fill) or not inst.sample:
                continue
            token = sampled[i]
            i += 1
            if rcb.eos_hit:
                continue                       # already ended by a token that was read back late
            if not inst.is_chunked:
                rcb.metric.token_times.append(now)
                rcb.output_token_ids.append(token)
            if inst.sample_dst is not None:
                inst.sample_dst.token_ids = [token]
            if not inst.is_chunked:
                last = rcb.is_finished()
                for p in rcb.output_token_processors:
                    p.append_token_id(token, last)
        batch.step()

    def _publish_prefix_blocks(self, batch: BatchRequest) -> None:
        """A block's hash enters the prefix cache in the step that computes its last token
        (executor.py:111-127); decode tokens are never published."""
        bs = self.kv_manager.block_size
        for rcb, inst in batch:
            if inst.hashes is None:
                continue
            vblocks = [c // bs for c in inst.cache_ids if c % bs == bs - 1 and c // bs < len(inst.hashes)]
            self.kv_manager.set_blocks(rcb.virtual_kv_cache, vblocks, [inst.hashes[v] for v in vblocks])

    def execute(self, batch: BatchRequest) -> None:
        if len(batch) == 0:
            return
        self._publish_prefix_blocks(batch)
        if self.graph_decoder is not None:
            rows = self._decode_rows(batch)
            if rows is not None:
                self._launch_decode(batch, rows)
                return
            self.resolve_pending()             # the eager path needs every token on the host
        sh = self.shape
        builder = LanguageModelParametersBuilder(
            self.image_manager, self.kv_manager, sh.num_hidden_layers, sh.num_attention_heads,
            sh.num_key_value_heads, sh.head_dim, self.language_model.image_token_id, self.dtype, self.device)
        builder.add_batch(batch)
        max_pos = getattr(sh, "max_position_embeddings", None)
        if max_pos is not None and builder.position_ids and max(builder.position_ids) >= max_pos:
            # the RoPE / attention kernels index the cos_sin table unchecked (admission normally rejects this:
            # request_processor.InstructionCreator.max_position_embeddings)
            raise HydraHipError(f"position {max(builder.position_ids)} outside the rotary table "
                                f"(max_position_embeddings = {max_pos})")
        inputs = builder.build_language_model_parameters()
        params = LanguageModelParameters(inputs.attention_params, inputs.all_sequences_decode,
                                         inputs.selected_token_ids_tensor, inputs.image_row_index)
        if not inputs.selected_token_ids:
            # nothing samples: the forward still has to run for its cache writes
            self.language_model.language_model.forward_hidden(
                self.language_model.embed(inputs.input_ids, inputs.image_features, inputs.image_row_index),
                inputs.position_ids, params)
            batch.step()
            return
        sampled = self.language_model.forward(inputs.input_ids, inputs.image_features, inputs.position_ids,
                                              params)
        if inputs.all_sequences_decode and sampled.numel() != len(inputs.selected_token_ids):
            sampled = sampled[inputs.selected_token_ids_tensor]
        sampled = sampled.tolist()                      # the step's only device sync

        self._deliver(batch, sampled)


class BatchImageEmbedExecutor:
    """use_graphs: the vision tower's shapes depend only on the number of images in the batch, so
    one hipGraph per image count (captured on first use) replaces ~280 eager launches — 5.5 ms of
    host-paced work becomes 3.2 ms for a single image."""

    def __init__(self, vision_model, image_cache_block_manager, n_qo_heads: int, head_dim: int,
                 dtype: torch.dtype, device: torch.device, use_graphs: bool = False):
        self.vision_model = vision_model
        self.manager = image_cache_block_manager
        self.n_qo_heads, self.head_dim = n_qo_heads, head_dim
        self.dtype, self.device = dtype, device
        self.use_graphs = use_graphs and device.type == "cuda"
        self.graphs = {}       # n_images -> (graph, static pixel buffer, static output)
        self._stages = {}      # (n_images, shape, dtype) -> [(pinned staging buffer, event behind its last copy), ...]

    def _encode(self, pixels: torch.Tensor) -> torch.Tensor:
        if not self.use_graphs or torch.cuda.is_current_stream_capturing():
            return self.vision_model.forward(pixels)
        n = pixels.shape[0]
        entry = self.graphs.get(n)
        if entry is None:
            static_in = pixels.clone()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self.vision_model.forward(static_in)          # warm-up outside capture
            torch.cuda.current_stream(self.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.vision_model.forward(static_in)
            entry = self.graphs[n] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        static_in.copy_(pixels)
        graph.replay()
        return static_out          # consumed (scattered into the image cache) before the next replay: same stream

    def _upload(self, pixels: List[torch.Tensor]) -> torch.Tensor:
        """The step's images as ONE device tensor of the model's dtype.  Host images go through a pinned staging buffer and
        ONE asynchronous copy instead of a synchronous pageable `.to(device)` per image (the runtime locks the pages of
        each first): the burst's 8-image steps are 6 ms shorter (first chunk done at 38 instead of 45 ms,
        tools/burst_timeline.py)."""
        if self.device.type != "cuda" or any(p.is_cuda for p in pixels):
            return torch.cat([p.to(device=self.device, dtype=self.dtype) for p in pixels], dim=0)
        n = sum(p.shape[0] for p in pixels)
        # kept pinned buffers, a few per (image count, shape): one whose last copy has finished is taken, another one is
        # made when all are still in flight — never a wait for the stream (an event wait made the host wait for the decode
        # step queued in front of the copy: Poisson TPOT p50 at 16 req/s 4.4 -> 5.6 ms) and, after the first steps, never
        # an allocation (a fresh pinned block per step is a 10-40 ms hipHostMalloc every now and then)
        key = (n, tuple(pixels[0].shape[1:]), pixels[0].dtype)
        ring = self._stages.setdefault(key, [])
        slot = next((e for e in ring if e[1].query()), None)
        if slot is None:
            slot = (torch.empty((n,) + key[1], dtype=key[2]).pin_memory(), torch.cuda.Event())
            if len(ring) < 8:
                ring.append(slot)
        stage, free = slot
        i = 0
        for p in pixels:
            stage[i:i + p.shape[0]].copy_(p)
            i += p.shape[0]
        dev_px = stage.to(self.device, non_blocking=True)
        free.record(torch.cuda.current_stream(self.device))
        return dev_px.to(self.dtype)

    def warmup(self, pixel_values: torch.Tensor, max_images: int) -> None:
        """Capture the graphs for 1 .. max_images images ahead of serving."""
        px = pixel_values.to(device=self.device, dtype=self.dtype)
        for n in range(1, max_images + 1):
            self._encode(px.expand(n, -1, -1, -1).contiguous())
            if not pixel_values.is_cuda:
                self._upload([pixel_values] * n)         # and the pinned staging buffer of that image count
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def execute(self, batch: BatchRequest) -> None:
        if len(batch) == 0:
            return
        slots: List[int] = []
        pixels = []
        for rcb, inst in batch:
            pixels.append(inst.pixel_values)
            inst.pixel_values = None
            slots += self.manager.v2p(rcb.virtual_image_cache, inst.cache_ids)
        feats = self._encode(self._upload(pixels))                      # (n_img, 576, hidden)
        tokens = feats.reshape(-1, self.n_qo_heads, self.head_dim)
        slot_t = torch.tensor(slots, dtype=torch.int32)
        if self.device.type == "cuda":
            slot_t = slot_t.pin_memory().to(self.device, non_blocking=True)
        self.manager.get_layer_cache(layer_id=0).set_caches(slot_t, [tokens])
        batch.step()


class InstructionExecutor:
    def __init__(self, fill_executor: Optional[BatchFillExecutor],
                 image_embed_executor: Optional[BatchImageEmbedExecutor], multi_streams_forward: bool = False):
        self.fill_executor = fill_executor
        self.image_embed_executor = image_embed_executor
        self.vision_stream = torch.cuda.Stream() if multi_streams_forward else None

    def execute_fill(self, batch: BatchRequest) -> None:
        if len(batch):
            self.fill_executor.execute(batch)

    def execute_image_embed(self, batch: BatchRequest) -> None:
        if len(batch) == 0:
            return
        if self.vision_stream is None:
            self.image_embed_executor.execute(batch)
            return
        self.vision_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.vision_stream):
            self.image_embed_executor.execute(batch)
        # the image cache is read by a later step's prefill on the main stream
        torch.cuda.current_stream().wait_stream(self.vision_stream)

    def execute_empty(self, batch: BatchRequest) -> None:
        batch.step()

    def resolve_pending(self) -> None:
        if self.fill_executor is not None:
            self.fill_executor.resolve_pending()
