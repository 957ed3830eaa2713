"""Offline inference on one GPU: a checkpoint directory (or two ready models) in, generated token
ids and their timing out — the collocated EPD node of engine/node.py behind a three-line API.
Mirrors what the reference's offline path hands back (hydrainfer/request/offline_inference_output.py:5-12);
prompts are token ids (one image placeholder id where the image goes): there is no tokenizer offline."""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

from hydrainfer_amd.engine.node import LocalCluster
from hydrainfer_amd.engine.rcb import SamplingParameters
from hydrainfer_amd.engine.request_processor import InstructionCreator, TokenRequest
from hydrainfer_amd.engine.scheduler import BatchSchedulerConfig
from hydrainfer_amd.engine.serve import build_node, quiet_gc, warm_library_gemms
from hydrainfer_amd.memory.shared_cache import compute_image_hash
from hydrainfer_amd.model.processor import ClipImageProcessor


@dataclass
class OfflineInferenceOutput:
    output_token_ids: List[int] = field(default_factory=list)
    arrival_time: float = -1.0
    finished_time: float = -1.0
    token_times: List[float] = field(default_factory=list)
    ttft: float = -1.0
    tpot: List[float] = field(default_factory=list)


@dataclass
class OfflineRequest:
    token_ids: List[int]
    image: object = None          # PIL image or None
    max_tokens: int = 50
    eos_token_ids: Sequence[int] = ()


class OfflineInferenceEngine:
    def __init__(self, language_model, vision_model, dtype: torch.dtype, device: str = "cuda:0",
                 max_running_requests: int = 32, token_budgets: int = 2048, image_budgets: int = 8,
                 max_context: int = 4096, kv_cache_gib: Optional[float] = None, warm_up: bool = True):
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        shape = language_model.language_model.shape
        self.n_image_tokens = (vision_model.shape.image_size // vision_model.shape.patch_size) ** 2
        blocks_per_seq = (max_context + 15) // 16
        per_block = shape.num_hidden_layers * 2 * 16 * shape.num_key_value_heads * shape.head_dim * 2
        n_blocks = int(kv_cache_gib * (1 << 30) // per_block) if kv_cache_gib else blocks_per_seq * (max_running_requests + 2)
        sched = BatchSchedulerConfig(priority="prefill", max_running_requests=max_running_requests,
                                     chunked_prefill=True, token_budgets=token_budgets, image_budgets=image_budgets)
        self.node = build_node("EPD0", "EPD", language_model, vision_model, shape, dtype, self.device, n_blocks,
                               2 * max_running_requests + 2, self.n_image_tokens, sched,
                               max_blocks_per_seq=blocks_per_seq)
        self.cluster = LocalCluster([self.node])
        self.creator = InstructionCreator(image_token_id=language_model.image_token_id,
                                          n_image_tokens_per_image=self.n_image_tokens, block_size=16, ignore_eos=True,
                                          max_position_embeddings=shape.max_position_embeddings)
        self.processor = ClipImageProcessor(size=vision_model.shape.image_size)
        if warm_up:
            warm_library_gemms(language_model, token_budgets, max_running_requests)
            size = vision_model.shape.image_size
            self.node.executor.image_embed_executor.warmup(torch.zeros(1, 3, size, size), image_budgets)
            self.node.executor.fill_executor.graph_decoder.warmup(list(range(4, max_running_requests + 1, 4)))

    @classmethod
    def from_checkpoint(cls, model_path: str, dtype: torch.dtype = torch.float16, device: str = "cuda:0", **kw):
        from hydrainfer_amd.model.loader import load_llava
        lm, vm = load_llava(model_path, dtype, device)
        return cls(lm, vm, dtype, device, **kw)

    def generate(self, requests: List[OfflineRequest]) -> List[OfflineInferenceOutput]:
        import time
        rcbs = []
        with quiet_gc():
            for i, r in enumerate(requests):
                pixels = self.processor.process(r.image) if r.image is not None else None
                req = TokenRequest(request_id=i, token_ids=list(r.token_ids), pixel_values=pixels,
                                   image_size=(r.image.size[1], r.image.size[0]) if r.image is not None else (0, 0),
                                   image_hash=compute_image_hash(r.image) if r.image is not None else 0,
                                   sampling_params=SamplingParameters(r.max_tokens, list(r.eos_token_ids)))
                rcb = self.creator.process(req)
                rcb.metric.arrival_time = time.perf_counter()
                rcbs.append(rcb)
                self.cluster.add_request(rcb)
                if (i + 1) % 8 == 0:
                    self.cluster.step()          # the GPU starts on the first ones while the rest is prepared
            while not self.cluster.idle():
                self.cluster.step()
        outs = []
        for rcb in rcbs:
            t = rcb.metric.token_times
            outs.append(OfflineInferenceOutput(
                output_token_ids=list(rcb.output_token_ids), arrival_time=rcb.metric.arrival_time,
                finished_time=rcb.metric.finished_time, token_times=list(t),
                ttft=t[0] - rcb.metric.arrival_time if t else -1.0,
                tpot=[b - a for a, b in zip(t, t[1:])]))
        return outs
