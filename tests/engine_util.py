"""Shared helpers for the engine tests: CPU stand-ins for device pools, oracle-backed models,
and a driver that feeds a scripted arrival trace to a LocalCluster."""
from types import SimpleNamespace as NS
from typing import List

import torch

from hydrainfer_amd.engine import BatchScheduler, BatchSchedulerConfig, BatchSchedulerContext, InstructionCreator
from hydrainfer_amd.engine.executor import BatchFillExecutor, BatchImageEmbedExecutor, InstructionExecutor
from hydrainfer_amd.engine.node import EPDNode, LocalCluster, NodeType
from hydrainfer_amd.memory.token_cache_manger import BlockTableManager


class CpuTokenCache:
    def __init__(self, caches):
        self.caches = caches

    def get_caches(self):
        return self.caches

    def set_caches(self, slot_ids, values):
        for cache, value in zip(self.caches, values):
            cache.view(-1, *cache.shape[2:])[slot_ids.long()] = value.to(cache.dtype)


class CpuPoolManager(BlockTableManager):
    """BlockTableManager + a CPU tensor standing in for the HBM pool (tests only)."""
    registry = {}

    def __init__(self, n_layers, n_tokens, n_blocks, block_size, heads, head_dim, dtype=torch.float32, seed=0):
        key = len(CpuPoolManager.registry) + 1
        super().__init__(n_blocks, block_size, rank=0, memory_handle=[key])
        CpuPoolManager.registry[key] = self
        self.device = torch.device("cpu")
        g = torch.Generator().manual_seed(seed)
        self.cache_tensor = torch.randn((n_layers, n_tokens, n_blocks, block_size, heads, head_dim),
                                        generator=g).to(dtype)

    def get_layer_cache(self, layer_id):
        return CpuTokenCache([self.cache_tensor[layer_id, t] for t in range(self.cache_tensor.shape[1])])

    def migrate_blocks(self, src, dst, is_send=False):
        if is_send:
            return
        peer = CpuPoolManager.registry[src.memory_handle[0]]
        n = len(src.block_table)
        assert len(dst.block_table) == n
        self.cache_tensor[:, :, dst.block_table] = peer.cache_tensor[:, :, src.block_table]


class OracleLM:
    """LlavaLanguageModel interface over the CPU oracle (tests only)."""

    def __init__(self, shape, sd, dtype, image_token_id):
        from oracle.model import OracleLlama
        self.language_model = NS(shape=shape)
        self.image_token_id = image_token_id
        self.sd, self.model = sd, OracleLlama(shape, sd, dtype)
        self.logits: List[torch.Tensor] = []

    def forward(self, input_ids, image_features, position_ids, params):
        from oracle.model import OracleAttnMeta
        ap = params.attention_params[0]
        meta = OracleAttnMeta(ap.q_cu_seq_lens, ap.kv_cu_seq_lens, ap.new_cache_slots, ap.block_tables,
                              ap.cu_blocks_lens)
        emb = torch.nn.functional.embedding(input_ids.long(), self.sd["model.embed_tokens.weight"])
        if image_features is not None:
            emb[input_ids == self.image_token_id] = image_features.reshape(-1, emb.shape[-1]).to(emb.dtype)
        caches = [p.kv_cache.get_kv_cache() for p in params.attention_params]
        logits = self.model.forward_logits(emb, position_ids, meta, caches, params.selected_token_ids).float()
        self.logits.append(logits)
        return logits.argmax(-1)


class OracleVision:
    def __init__(self, shape, sd):
        self.shape, self.sd = shape, sd

    def forward(self, pixels):
        from oracle.vision import vision_forward
        return vision_forward(self.shape, self.sd, pixels)


class LogitsTap:
    """Wraps a LlavaLanguageModel and keeps the logits of every fill batch."""

    def __init__(self, lm):
        self.lm, self.language_model, self.image_token_id = lm, lm.language_model, lm.image_token_id
        self.logits: List[torch.Tensor] = []

    def embed(self, *a):
        return self.lm.embed(*a)

    def forward(self, input_ids, image_features, position_ids, params):
        logits = self.lm.forward_logits(input_ids, image_features, position_ids, params)
        if logits.shape[0] != params.selected_token_ids.numel():
            logits = logits[params.selected_token_ids]
        self.logits.append(logits.float().cpu())
        return logits.argmax(-1)


def make_node(name, node_type, lm, vision, kv, img, lm_shape, dtype, device, sched_cfg: BatchSchedulerConfig,
              batch_log=None, graph_decode=False, eager_migrate=False):
    nt = NodeType(node_type)
    decoder = None
    if graph_decode and nt.enable_decode:
        from hydrainfer_amd.engine.graph_decode import GraphedDecoder
        decoder = GraphedDecoder(getattr(lm, "lm", lm), kv, max_batch=sched_cfg.max_running_requests,
                                 max_blocks_per_seq=8)
    fill = BatchFillExecutor(lm, kv, img, dtype, device, graph_decoder=decoder) if nt.has_language_model else None
    emb = BatchImageEmbedExecutor(vision, img, lm_shape.num_attention_heads, lm_shape.head_dim, dtype,
                                  device, use_graphs=graph_decode) if nt.has_vision_model else None
    if fill is not None and batch_log is not None:
        real = fill.execute

        def execute(batch):
            batch_log.append([rcb.request_id for rcb, inst in batch if inst.sample])
            real(batch)
        fill.execute = execute
    sched = BatchScheduler(sched_cfg, BatchSchedulerContext(kv if nt.has_kv_cache else None,
                                                            img if nt.has_image_cache else None))
    return EPDNode(name, nt, sched, InstructionExecutor(fill, emb), kv if nt.has_kv_cache else None,
                   img if nt.has_image_cache else None, eager_migrate=eager_migrate)


def run_trace(cluster: LocalCluster, creator: InstructionCreator, requests, max_steps=4000):
    """requests: list of (arrival_step, TokenRequest).  Returns the rcbs in request order."""
    rcbs = [None] * len(requests)
    last = max(a for a, _ in requests)
    step = 0
    while step <= last or not cluster.idle():
        for i, (a, r) in enumerate(requests):
            if a == step:
                rcbs[i] = creator.process(r)
                cluster.add_request(rcbs[i])
        cluster.step()
        step += 1
        assert step < max_steps, "engine did not drain"
    return rcbs
