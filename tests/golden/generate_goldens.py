"""Generates tests/golden/*.npz by running the REFERENCE's own Python handlers
(/root/reference, imported read-only) on the seeded inputs of tests/golden/cases.py.

Run only in the build container:  python tests/golden/generate_goldens.py
The reference never travels to the GPU box; only the .npz outputs (data) are committed.
Third-party modules the reference imports but this image lacks are replaced by empty
stubs (they are not on the code paths exercised here)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root
REFERENCE = os.environ.get("HYDRA_REFERENCE", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    for name in ("seaborn", "dacite", "ray", "ray.actor", "zmq", "zmq.asyncio", "zmq.sugar",
                 "zmq.sugar.socket", "hydra", "omegaconf", "shortuuid"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["dacite"].from_dict = lambda *a, **k: None
    sys.modules["dacite"].Config = object
    sys.modules["omegaconf"].OmegaConf = object
    sys.modules["omegaconf"].DictConfig = object
    sys.modules["ray"].remote = lambda *a, **k: (lambda f: f)
    sys.modules["ray.actor"].ActorHandle = object
    sys.modules["zmq.sugar.socket"].Socket = object
    sys.modules["zmq.asyncio"].Socket = object
    sys.modules["zmq"].sugar = sys.modules["zmq.sugar"]
    sys.modules["zmq"].asyncio = sys.modules["zmq.asyncio"]
    sys.modules["zmq.sugar"].socket = sys.modules["zmq.sugar.socket"]
    sys.modules["ray"].actor = sys.modules["ray.actor"]
    sys.path.insert(0, REFERENCE)
    import hydrainfer  # noqa: F401
    return hydrainfer


from tests.golden import cases as C  # noqa: E402


def gen_kv_cache(out):
    from hydrainfer.memory.kv_cache import KVCache
    from hydrainfer.memory.token_cache import TokenCache
    for i, case in enumerate(C.kv_cache_cases()):
        slot_ids, keys, values, kc, vc = C.kv_cache_inputs(case, seed=i)
        chk = C.checksum(slot_ids, keys.contiguous(), values.contiguous(), kc, vc)
        KVCache(kc, vc).set_kv_cache(slot_ids, keys, values)  # CPU loop kv_cache.py:44-50
        n = C.case_name("kv", i)
        # bit-exact op: the fixture is the checksum of the reference's resulting caches
        out[n + "_key_cache_chk"] = np.array(C.checksum(kc))
        out[n + "_value_cache_chk"] = np.array(C.checksum(vc))
        out[n + "_chk"] = np.array(chk)
        # image cache: same inputs, single tensor (token_cache.py:53-56)
        slot_ids, keys, values, kc, vc = C.kv_cache_inputs(case, seed=i)
        TokenCache([kc]).set_caches(slot_ids, [keys])
        out[n + "_image_cache_chk"] = np.array(C.checksum(kc))


def gen_paged_attention(out):
    from hydrainfer.layer.causal_attention import (AttentionParametersBuilder,
                                                   CausalGroupedQueryPageAttention,
                                                   CausalGroupedQueryPageAttentionConfig)
    from hydrainfer.memory.kv_cache import KVCache
    for i, case in enumerate(C.paged_attention_cases()):
        q, k, v, kc, vc, reqs = C.paged_attention_inputs(case, seed=i)
        chk = C.checksum(q, k, v, kc, vc)
        b = AttentionParametersBuilder(case["n_heads"], case["n_kv_heads"], case["head_dim"],
                                       case["block_size"], torch.device("cpu"))
        for r in reqs:
            b.add_request(r["q_len"], r["kv_len"], r["new_cache_slots"], r["block_table"])
        b.add_kv_cache(KVCache(kc, vc))
        params = b.build_attention_parameters()[0]
        attn = CausalGroupedQueryPageAttention(CausalGroupedQueryPageAttentionConfig(
            case["n_heads"], case["n_kv_heads"], case["head_dim"]))
        o = attn(q, k, v, params).o
        n = C.case_name("pattn", i)
        out[n + "_o"] = C.to_np(o)
        out[n + "_chk"] = np.array(chk)
        for f in ("q_cu_seq_lens", "kv_cu_seq_lens", "paged_kv_last_page_len", "new_cache_slots",
                  "block_tables", "cu_blocks_lens"):
            out[n + "_" + f] = getattr(params, f).numpy()
        out[n + "_scalars"] = np.array([params.num_sequences, int(params.all_sequences_decode),
                                        params.q_max_seq_len, params.kv_max_seq_len])
        # the cache after set_kv_cache is checked through a checksum (keeps the fixture small)
        out[n + "_cache_chk"] = np.array(C.checksum(kc, vc))


def gen_dense_attention(out):
    from hydrainfer.layer.multihead_attention import (MultiHeadAttentionConfig,
                                                      MultiHeadAttentionParameters,
                                                      TorchMultiHeadAttentionHandler)
    for i, case in enumerate(C.dense_attention_cases()):
        q, k, v = C.dense_attention_inputs(case, seed=i)
        h = TorchMultiHeadAttentionHandler(MultiHeadAttentionConfig(case["n_heads"], case["head_dim"]))
        o = h(q, k, v, MultiHeadAttentionParameters()).o
        n = C.case_name("dattn", i)
        out[n + "_o"] = C.to_np(o)
        out[n + "_chk"] = np.array(C.checksum(q, k, v))


def gen_rms_norm(out):
    from hydrainfer.layer.norm import rmsnorm
    for i, case in enumerate(C.rms_norm_cases()):
        x, w = C.rms_norm_inputs(case, seed=i)
        o = rmsnorm(x, w, case["eps"])  # CPU -> torch branch, norm.py:18-23
        n = C.case_name("rms", i)
        out[n + "_o"] = C.to_np(o)
        out[n + "_chk"] = np.array(C.checksum(x, w))


def gen_rope(out):
    from hydrainfer.layer.rotary_embedding import (FusedKernelRotaryEmbeddingHandler,
                                                   TorchRotaryEmbeddingHandler,
                                                   compute_default_inv_freq)
    for i, case in enumerate(C.rope_cases()):
        q, k, pos = C.rope_inputs(case, seed=i)
        dt = C.DTYPES[case["dtype"]]
        inv = compute_default_inv_freq(case["rotary_dim"], case["theta"])
        h = TorchRotaryEmbeddingHandler(case["rotary_dim"], case["max_pos"], inv, case["interleaved"])
        h = h.to(dt)  # model.to(dtype) casts the registered cos/sin buffer (llava.py:125-126)
        qo, ko = h(q, k, pos)
        fused = FusedKernelRotaryEmbeddingHandler(case["rotary_dim"], case["max_pos"], inv,
                                                  case["interleaved"]).to(dt)
        n = C.case_name("rope", i)
        out[n + "_q"] = C.to_np(qo)
        out[n + "_k"] = C.to_np(ko)
        # kernel-layout cache [max_pos, 2, rot/2]: only its checksum (rebuilt by the oracle)
        out[n + "_cos_sin_chk"] = np.array(C.checksum(fused.cos_sin_cache))
        out[n + "_chk"] = np.array(C.checksum(q, k, pos))


def gen_silu(out):
    from hydrainfer.layer.activation import silu
    for i, case in enumerate(C.silu_cases()):
        x = C.silu_inputs(case, seed=i)
        n = C.case_name("silu", i)
        out[n + "_o"] = C.to_np(silu(x))
        out[n + "_chk"] = np.array(C.checksum(x))


def gen_trace(out):
    """G7: integer metadata of a scripted continuous-batching run — block allocator order,
    v2p slots, prefix hashes and the per-step AttentionParameters tensors."""
    from hydrainfer.layer.causal_attention import AttentionParametersBuilder
    from hydrainfer.memory.block_allocator import BlockAllocator
    from hydrainfer.memory.shared_cache import compute_hash, SharedCache, SharedCacheConfig
    cfg = C.TraceConfig()
    bs = cfg.block_size
    alloc = BlockAllocator(cfg.n_blocks)
    shared = SharedCache(SharedCacheConfig(n_blocks=cfg.n_blocks))
    tables, lens = [], []

    def realloc(table, n_tokens):  # TokenCacheBlockManager.realloc growth branch, token_cache_manger.py:149-153
        need = (n_tokens + bs - 1) // bs - len(table)
        blocks = alloc.allocate(need)
        if len(blocks) < need:
            blocks += shared.allocate(need)
        shared.pin(blocks)
        table += blocks

    def v2p(table, ids):  # token_cache_manger.py:126-133
        return [table[i // bs] * bs + i % bs for i in ids]

    hashes = []
    for r in range(cfg.n_requests):
        ids = C.trace_token_ids(cfg, r)
        hashes.append(np.array(compute_hash(ids, bs, -1), dtype=np.uint64))
        tables.append([])
        lens.append(0)
    out["trace_hashes"] = np.stack(hashes)

    def step(q_lens, tag):
        b = AttentionParametersBuilder(32, 32, 128, bs, torch.device("cpu"))
        for r, q_len in enumerate(q_lens):
            realloc(tables[r], lens[r] + q_len)
            slots = v2p(tables[r], list(range(lens[r], lens[r] + q_len)))
            lens[r] += q_len
            b.add_request(q_len, lens[r], slots, tables[r])
        b.add_kv_cache(None)
        p = b.build_attention_parameters()[0]
        for f in ("q_cu_seq_lens", "kv_cu_seq_lens", "paged_kv_last_page_len", "new_cache_slots",
                  "block_tables", "cu_blocks_lens"):
            out[f"trace_{tag}_{f}"] = getattr(p, f).numpy()
        out[f"trace_{tag}_scalars"] = np.array([p.num_sequences, int(p.all_sequences_decode),
                                               p.q_max_seq_len, p.kv_max_seq_len])

    step([cfg.prompt_len] * cfg.n_requests, "prefill")
    for d in range(cfg.n_decode):
        step([1] * cfg.n_requests, f"decode{d}")
    # free two requests, admit again: LIFO reuse order (block_allocator.py:25-36)
    for r in (3, 17):
        shared.unpin(tables[r])
        alloc.free(tables[r])
        tables[r], lens[r] = [], 0
    realloc(tables[3], 40)
    realloc(tables[17], 700)
    out["trace_realloc_3"] = np.array(tables[3], dtype=np.int32)
    out["trace_realloc_17"] = np.array(tables[17], dtype=np.int32)
    out["trace_free_blocks_tail"] = np.array(alloc.free_blocks[-8:], dtype=np.int32)


def gen_tiny_llama(out):
    """G8: the reference's LlamaForCausalLM (hydrainfer/model/llama.py) on CPU — prefill of two
    requests then greedy decode; logits captured by hooking lm_head (the model returns ids only)."""
    from transformers import LlamaConfig
    from hydrainfer.layer.causal_attention import AttentionParametersBuilder
    from hydrainfer.memory.kv_cache import KVCache
    from hydrainfer.model.llama import LlamaForCausalLM
    from hydrainfer.model.parameters import LanguageModelParameters
    t = C.TINY_LLAMA
    bs = C.TINY_BLOCK_SIZE
    for dname in ("fp16", "bf16"):
        dt = C.DTYPES[dname]
        cfg = LlamaConfig(hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"],
                          num_hidden_layers=t["num_hidden_layers"],
                          num_attention_heads=t["num_attention_heads"],
                          num_key_value_heads=t["num_key_value_heads"], vocab_size=t["vocab_size"],
                          rms_norm_eps=t["rms_norm_eps"],
                          max_position_embeddings=t["max_position_embeddings"])
        cfg.head_dim = t["head_dim"]
        cfg.rope_theta = t["rope_theta"]
        model = LlamaForCausalLM(cfg)
        sd = C.tiny_llama_state_dict(dt)
        missing = model.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys and all("rotary" in k or "cos_sin" in k for k in missing.missing_keys), missing
        model.to(dt).eval()
        logits_log = []
        model.lm_head.register_forward_hook(lambda m, i, o: logits_log.append(o.detach().float().clone()))
        L, HK, D = t["num_hidden_layers"], t["num_key_value_heads"], t["head_dim"]
        gen = torch.Generator().manual_seed(77)
        pool = torch.randn((L, 2, C.TINY_BLOCKS, bs, HK, D), generator=gen).to(dt)
        out[f"tiny_{dname}_pool_chk"] = np.array(C.checksum(pool))  # before any KV write
        tables = C.tiny_block_tables()
        lens = [0, 0]
        tokens = []

        def run(ids_per_req):
            b = AttentionParametersBuilder(t["num_attention_heads"], HK, D, bs, torch.device("cpu"))
            ids, pos, sel, n = [], [], [], 0
            for r, new in enumerate(ids_per_req):
                slots = [tables[r][p // bs] * bs + p % bs for p in range(lens[r], lens[r] + len(new))]
                pos += list(range(lens[r], lens[r] + len(new)))
                lens[r] += len(new)
                b.add_request(len(new), lens[r], slots, tables[r][: (lens[r] + bs - 1) // bs])
                ids += new
                n += len(new)
                sel.append(n - 1)
            for l in range(L):
                b.add_kv_cache(KVCache(pool[l, 0], pool[l, 1]))
            ap = b.build_attention_parameters()
            params = LanguageModelParameters(
                input_ids_or_input_embeds=None, position_ids=None, image_features=None,
                image_overwrite_mask=None, attention_params=ap,
                all_sequences_decode=all(len(x) == 1 for x in ids_per_req), selected_token_ids=sel)
            with torch.inference_mode():
                return model(torch.tensor(ids, dtype=torch.int), torch.tensor(pos, dtype=torch.int), params)

        nxt = run([C.tiny_prompt_ids(0), C.tiny_prompt_ids(1)])
        tokens.append(nxt.tolist())
        for _ in range(C.TINY_DECODE_STEPS - 1):
            nxt = run([[int(nxt[0])], [int(nxt[1])]])
            tokens.append(nxt.tolist())
        out[f"tiny_{dname}_tokens"] = np.array(tokens, dtype=np.int64)
        out[f"tiny_{dname}_logits"] = torch.stack(logits_log).numpy()  # [steps, 2, vocab] fp32
        out[f"tiny_{dname}_pool_end_chk"] = np.array(C.checksum(pool))  # after all KV writes


def gen_tiny_clip(out):
    """G9: the reference's CLIPVisionModel + LlavaMultiModalProjector on CPU (tiny config)."""
    from transformers import CLIPVisionConfig
    from hydrainfer.model.clip import CLIPVisionModel
    from hydrainfer.model.llava import LlavaMultiModalProjector
    from hydrainfer.model.parameters import VisionModelParameters
    from hydrainfer_amd.model.clip import ClipShape, random_state_dict
    t = C.TINY_CLIP
    shape = ClipShape(**t)
    vcfg = CLIPVisionConfig(hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"],
                            num_hidden_layers=t["num_hidden_layers"],
                            num_attention_heads=t["num_attention_heads"], image_size=t["image_size"],
                            patch_size=t["patch_size"], layer_norm_eps=t["layer_norm_eps"])

    class _Cfg:  # the two attributes LlavaMultiModalProjector reads (llava.py:33-34)
        vision_config = vcfg
        text_config = type("T", (), {"hidden_size": t["projector_hidden_size"]})()

    sd32 = random_state_dict(shape, seed=3, std=0.05)
    pixels = C.tiny_clip_pixels()
    for dname in ("fp16", "bf16"):
        dt = C.DTYPES[dname]
        tower, proj = CLIPVisionModel(vcfg), LlavaMultiModalProjector(_Cfg)
        tower.load_state_dict({k[len("vision_tower."):]: v for k, v in sd32.items()
                               if k.startswith("vision_tower.")}, strict=False)
        proj.load_state_dict({k[len("multi_modal_projector."):]: v for k, v in sd32.items()
                              if k.startswith("multi_modal_projector.")})
        tower.to(dt).eval(); proj.to(dt).eval()
        with torch.inference_mode():
            h, _ = tower(pixels, t["vision_feature_layer"], VisionModelParameters())
            feat = proj(h[:, 1:])
        out[f"clip_{dname}_features"] = C.to_np(feat)
    out["clip_pixels_chk"] = np.array(C.checksum(pixels))


def gen_tiny_llava(out):
    """G10: image -> reference CLIP tower + projector -> features overwrite the image-token rows of
    the reference Llama's input embeddings (the four lines of hydrainfer/model/llava.py:132-136,
    whose class needs a checkpoint directory) -> reference LlamaForCausalLM prefill + greedy decode."""
    from transformers import CLIPVisionConfig, LlamaConfig
    from hydrainfer.layer.causal_attention import AttentionParametersBuilder
    from hydrainfer.memory.kv_cache import KVCache
    from hydrainfer.model.clip import CLIPVisionModel
    from hydrainfer.model.llama import LlamaForCausalLM
    from hydrainfer.model.llava import LlavaMultiModalProjector
    from hydrainfer.model.parameters import LanguageModelParameters, VisionModelParameters
    from hydrainfer_amd.model.clip import ClipShape, random_state_dict
    t, tc, bs = C.TINY_LLAMA, C.TINY_CLIP, C.TINY_BLOCK_SIZE
    vcfg = CLIPVisionConfig(hidden_size=tc["hidden_size"], intermediate_size=tc["intermediate_size"],
                            num_hidden_layers=tc["num_hidden_layers"], num_attention_heads=tc["num_attention_heads"],
                            image_size=tc["image_size"], patch_size=tc["patch_size"], layer_norm_eps=tc["layer_norm_eps"])

    class _Cfg:
        vision_config = vcfg
        text_config = type("T", (), {"hidden_size": tc["projector_hidden_size"]})()
    sd32 = random_state_dict(ClipShape(**tc), seed=3, std=0.05)
    pixels = C.tiny_clip_pixels(2)
    for dname in ("fp16", "bf16"):
        dt = C.DTYPES[dname]
        tower, proj = CLIPVisionModel(vcfg), LlavaMultiModalProjector(_Cfg)
        tower.load_state_dict({k[len("vision_tower."):]: v for k, v in sd32.items() if k.startswith("vision_tower.")}, strict=False)
        proj.load_state_dict({k[len("multi_modal_projector."):]: v for k, v in sd32.items() if k.startswith("multi_modal_projector.")})
        tower.to(dt).eval(); proj.to(dt).eval()
        cfg = LlamaConfig(hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"],
                          num_hidden_layers=t["num_hidden_layers"], num_attention_heads=t["num_attention_heads"],
                          num_key_value_heads=t["num_key_value_heads"], vocab_size=t["vocab_size"],
                          rms_norm_eps=t["rms_norm_eps"], max_position_embeddings=t["max_position_embeddings"])
        cfg.head_dim, cfg.rope_theta = t["head_dim"], t["rope_theta"]
        lm = LlamaForCausalLM(cfg)
        lm.load_state_dict(C.tiny_llama_state_dict(dt), strict=False)
        lm.to(dt).eval()
        logits_log = []
        lm.lm_head.register_forward_hook(lambda m, i, o: logits_log.append(o.detach().float().clone()))
        L, HK, D = t["num_hidden_layers"], t["num_key_value_heads"], t["head_dim"]
        pool = torch.randn((L, 2, C.TINY_BLOCKS, bs, HK, D), generator=torch.Generator().manual_seed(77)).to(dt)
        tables, lens, tokens = C.tiny_llava_block_tables(), [0, 0], []
        with torch.inference_mode():
            h, _ = tower(pixels, tc["vision_feature_layer"], VisionModelParameters())
            feats = proj(h[:, 1:])                                     # [2, 16, 256]

        def run(ids_per_req, image_features):
            b = AttentionParametersBuilder(t["num_attention_heads"], HK, D, bs, torch.device("cpu"))
            ids, pos, sel, n = [], [], [], 0
            for r, new in enumerate(ids_per_req):
                slots = [tables[r][p // bs] * bs + p % bs for p in range(lens[r], lens[r] + len(new))]
                pos += list(range(lens[r], lens[r] + len(new)))
                lens[r] += len(new)
                b.add_request(len(new), lens[r], slots, tables[r][: (lens[r] + bs - 1) // bs])
                ids += new
                n += len(new)
                sel.append(n - 1)
            for l in range(L):
                b.add_kv_cache(KVCache(pool[l, 0], pool[l, 1]))
            params = LanguageModelParameters(
                input_ids_or_input_embeds=None, position_ids=None, image_features=None, image_overwrite_mask=None,
                attention_params=b.build_attention_parameters(),
                all_sequences_decode=all(len(x) == 1 for x in ids_per_req), selected_token_ids=sel)
            with torch.inference_mode():
                input_ids = torch.tensor(ids, dtype=torch.int)
                embeds = lm.model.embed_tokens(input_ids)
                if image_features is not None:
                    embeds[input_ids == C.TINY_IMAGE_TOKEN_ID, :] = image_features.view(-1, embeds.shape[-1])
                return lm(embeds, torch.tensor(pos, dtype=torch.int), params)

        nxt = run([C.tiny_llava_prompt(0), C.tiny_llava_prompt(1)], feats)
        tokens.append(nxt.tolist())
        for _ in range(C.TINY_DECODE_STEPS - 1):
            nxt = run([[int(nxt[0])], [int(nxt[1])]], None)
            tokens.append(nxt.tolist())
        out[f"llava_{dname}_tokens"] = np.array(tokens, dtype=np.int64)
        out[f"llava_{dname}_logits"] = torch.stack(logits_log).numpy()


class Ragged:
    """Accumulates variable-length integer rows; saved as <name>_flat + <name>_off."""

    def __init__(self):
        self.rows = {}

    def add(self, name, values):
        self.rows.setdefault(name, []).append(np.asarray(values, dtype=np.int64).reshape(-1))

    def save(self, out, prefix):
        for name, rows in self.rows.items():
            out[f"{prefix}_{name}_flat"] = np.concatenate(rows) if rows else np.zeros(0, np.int64)
            out[f"{prefix}_{name}_off"] = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)


INST_CODES = {"EM": 0, "TF": 1, "EF": 2, "IE": 3, "EPMR": 4, "PDMR": 5, "PR": 6}


def gen_engine_trace(out):
    """G11: the reference's own InstructionCreator.process, BatchScheduler.step, BatchFillExecutor /
    BatchImageEmbedExecutor.execute (hence LanguageModelParametersBuilder and
    AttentionParametersBuilder) and AsyncEPDNode.step, run unmodified on CPU for one collocated EPD
    node over a scripted arrival trace.  Only the things that cannot exist here are stood in for:
    tokenizer / image processor / model factory (token ids and pixel tensors are given), the two
    models (the sampled token is tests.golden.cases.engine_trace_sample of the row; the image
    embedding rows carry request*1000 + index), ray actor identity, IPC handle and CUDA stream."""
    import asyncio
    from PIL import Image
    import hydrainfer.memory.token_cache_manger as tcm
    from hydrainfer.memory import TokenCacheBlockManager, TokenCacheBlockManagerConfig, TokenCacheBlockManagerContext
    from hydrainfer.engine import (BatchScheduler, RequestControlBlock, RequestProcessParameters, Fill)
    from hydrainfer.engine.scheduler import BatchSchedulerConfig, BatchSchedulerContext
    from hydrainfer.engine.request_processor import InstructionCreator
    from hydrainfer.engine.executor import (BatchFillExecutor, BatchImageEmbedExecutor, InstructionExecutor,
                                            ExecutorContext)
    from hydrainfer.engine.output_token_processor import OutputTokenParams
    from hydrainfer.engine.scenario import ScenarioClassifier
    from hydrainfer.model.parameters import LanguageModelOutput, VisionModelOutput
    from hydrainfer.request import Request, SamplingParameters
    from hydrainfer.cluster.epdnode import AsyncEPDNode, NodeContext
    from hydrainfer.cluster import MigrateGraph, MigrateNode, NodeType
    from types import SimpleNamespace as NS

    tcm.get_ipc_mem_handle = lambda t: [0] * 64
    tcm.CommunicationBackendManager = lambda *a, **k: None
    real_stream = torch.cuda.Stream
    torch.cuda.Stream = lambda *a, **k: NS(synchronize=lambda: None)
    try:
        for cfg in C.ENGINE_TRACES:
            _engine_trace_one(cfg, out, locals())
    finally:
        torch.cuda.Stream = real_stream
    # the budget search of BatchSchedulerProfiler (profiler.py:118-133) on threshold criteria and on
    # the non-monotonic / raising criterion of tests.golden.cases.profiler_weird_criterion
    from hydrainfer.engine.profiler import BatchSchedulerProfiler
    prof = object.__new__(BatchSchedulerProfiler)
    prof.config = NS(debug=False)
    for hi in (8, 2048):
        out[f"profiler_search_{hi}"] = np.array(
            [prof._binary_search_max_batch_size(1, hi, lambda n, T=T: n <= T) for T in range(hi + 3)] +
            [prof._binary_search_max_batch_size(1, hi, C.profiler_weird_criterion)])


def _engine_trace_one(cfg, out, L):
    import asyncio
    from PIL import Image
    from types import SimpleNamespace as NS
    reqs = C.engine_trace_requests(cfg)
    rag = Ragged()
    dev = torch.device("cpu")

    def manager(n_layers, n_tokens, n_blocks, block_size, heads):
        c = L["TokenCacheBlockManagerConfig"](communication_backend_manager_config=None, n_layers=n_layers,
                                              n_tokens=n_tokens, n_blocks=n_blocks, block_size=block_size,
                                              n_heads=heads, head_size=cfg.head_dim, dtype="fp32", device="cpu")
        return L["TokenCacheBlockManager"](c, L["TokenCacheBlockManagerContext"](rank=0, rank2host={0: "h"}))

    kv = manager(cfg.n_layers, 2, cfg.kv_blocks, cfg.block_size, cfg.n_heads)
    img = manager(1, 1, cfg.image_blocks, cfg.n_image_tokens, cfg.n_heads)
    lm_cfg = NS(n_layers=cfg.n_layers, n_qo_heads=cfg.n_heads, n_kv_heads=cfg.n_heads, head_dim=cfg.head_dim)
    vis_cfg = NS(image_token_id=cfg.image_token_id)

    class Worker:   # stands in for both models
        def execute_language_model(self, input_ids, image_features, position_ids, p):
            ap = p.attention_params[0]
            rag.add("input_ids", input_ids)
            rag.add("position_ids", position_ids)
            rag.add("selected", p.selected_token_ids)
            rag.add("image_rows", [] if image_features is None else image_features[:, 0].round().long())
            rag.add("q_cu", ap.q_cu_seq_lens)
            rag.add("kv_cu_reference", ap.kv_cu_seq_lens)
            rag.add("new_cache_slots", ap.new_cache_slots)
            rag.add("block_tables", ap.block_tables)
            rag.add("cu_blocks_lens", ap.cu_blocks_lens)
            rag.add("fill_scalars", [ap.num_sequences, int(ap.all_sequences_decode), ap.q_max_seq_len])
            ids, pos = input_ids.tolist(), position_ids.tolist()
            toks = [C.engine_trace_sample(ids[j], pos[j]) for j in p.selected_token_ids]
            return L["LanguageModelOutput"](sample_token_ids=torch.tensor(toks, dtype=torch.int))

        def execute_vision_model(self, pixel_values, params):
            feats = []
            for pv in pixel_values:     # pv[0,0,0,0] carries the request index
                r = int(pv.flatten()[0].item())
                f = torch.zeros(1, cfg.n_image_tokens, cfg.n_heads * cfg.head_dim)
                f[0, :, 0] = r * 1000 + torch.arange(cfg.n_image_tokens)
                feats.append(f)
            rag.add("encode_requests", [int(pv.flatten()[0].item()) for pv in pixel_values])
            return L["VisionModelOutput"](image_features=torch.cat(feats, dim=0))

    worker = Worker()
    fill = object.__new__(L["BatchFillExecutor"])
    fill.config = NS(use_flash_infer=False)
    fill.context = NS(kv_cache_block_manager=kv, image_cache_block_manager=img, worker=worker, zmq_send=None)
    fill.worker, fill.vision_model_config, fill.language_model_config = worker, vis_cfg, lm_cfg
    fill.tokenizer, fill.dtype, fill.device = None, torch.float32, dev
    fill.block_mangaer, fill.image_block_manager = kv, img
    fill.batch_prefill_with_paged_kvcache_wrapper = fill.batch_decode_with_paged_kvcache_wrapper = None
    fill.print_text_output_token_processor = None
    emb = object.__new__(L["BatchImageEmbedExecutor"])
    emb.worker, emb.block_manager, emb.language_model_config = worker, img, lm_cfg
    emb.n_qo_heads, emb.head_dim, emb.dtype, emb.device = cfg.n_heads, cfg.head_dim, torch.float32, dev
    executor = object.__new__(L["InstructionExecutor"])
    executor.image_embed_executor, executor.fill_executor = emb, fill

    profiler = NS(profile_image_budgets=lambda: cfg.image_budgets, profile_token_budgets=lambda: cfg.token_budgets)
    sched = L["BatchScheduler"](
        L["BatchSchedulerConfig"](priority=cfg.priority, max_running_requests=cfg.max_running_requests,
                                  chunked_prefill=cfg.chunked_prefill),
        L["BatchSchedulerContext"](profiler=profiler, kv_cache_block_manager=kv, image_cache_block_manager=img))
    real_step = sched.step

    def recording_step():
        batch = real_step()
        rows = [(rcb.sid, INST_CODES[repr(inst)], len(inst.token_ids) if isinstance(inst, L["Fill"]) else 0)
                for rcb, inst in batch] if len(batch) else []
        rag.add("batch_sid", [r[0] for r in rows])
        rag.add("batch_inst", [r[1] for r in rows])
        rag.add("batch_ntok", [r[2] for r in rows])
        return batch
    sched.step = recording_step

    creator = object.__new__(L["InstructionCreator"])
    creator.config = NS(debug=False)
    creator.image_token_id, creator.block_size = cfg.image_token_id, cfg.block_size
    creator.image_token_caculator = NS(get_num_image_tokens=lambda image_size: cfg.n_image_tokens)
    prompts = {}
    creator.tokenizer = NS(encode=lambda prompt: list(prompts[prompt]))
    creator.processor = NS(process=lambda image: torch.full((1, 3, 2, 2), float(image.info["request"])))

    node = object.__new__(L["AsyncEPDNode"])
    node.actor_id, node.actor_handle, node.name = "self", None, "EPD"
    node.config = NS(log_latency_breakdown=False)
    me = L["MigrateNode"](id="self", tpot_slo=0.4, actor=None)
    node.context = L["NodeContext"](rank=0, world_size=1, node_type=L["NodeType"]("EPD"),
                                    migrate_graph=L["MigrateGraph"](ep_table={"self": [me]}, pd_table={"self": [me]}))
    node._update_migrate_graph(node.context)
    node.batch_scheduler, node.executor = sched, executor
    node.kv_cache_block_manager, node.image_cache_block_manager, node.zmq_send = kv, img, None

    rcbs = []
    classifier = L["ScenarioClassifier"]()

    def admit(i, r):
        prompts[f"p{i}"] = r.token_ids
        image = None
        if r.image_seed >= 0:
            image = Image.fromarray(C.engine_trace_image(r.image_seed))
            image.info["request"] = i
        request = L["Request"](request_id=i, prompt=f"p{i}", image=image,
                               sampling_params=L["SamplingParameters"](max_tokens=r.max_tokens))
        rcb = L["RequestControlBlock"]()
        rcb.sampling_params = request.sampling_params          # SamplingParamsProcess, ignore_eos=True
        rcb = creator.process(request, rcb, L["RequestProcessParameters"]())
        rcb.scenario_type = classifier.classify(n_text_tokens=rcb.request_metadata.n_text_tokens,
                                                n_output_tokens=r.max_tokens)
        rcb.request_id = i
        rcb.output_token_params = L["OutputTokenParams"](print_output_text=False, zmq_output=False)
        first = rcb.instructions.head.next
        if first.hashes is not None and r.image_seed >= 0:
            rag.add("image_hash", np.array(first.hashes[:1], dtype=np.uint64).view(np.int64))
        fillinst = first if isinstance(first, L["Fill"]) else first.next.next.next
        rag.add("prefix_hashes", np.array(fillinst.hashes, dtype=np.uint64).view(np.int64))
        rcbs.append(rcb)
        sched.schedule_new(rcb)

    async def drive():
        step = 0
        while True:
            for i, r in enumerate(reqs):
                if r.arrival_step == step:
                    admit(i, r)
            await node.step()
            await asyncio.sleep(0.001)       # lets the migrate tasks of this step run (epdnode.py:347)
            rag.add("free_blocks", [kv.get_num_avaiable_blocks(), img.get_num_avaiable_blocks(),
                                    len(kv.block_allocator.free_blocks), len(sched.running), len(sched.waiting)])
            step += 1
            if step > max(r.arrival_step for r in reqs) and not sched.running and not sched.waiting:
                return step
            assert step < 2000

    n_steps = asyncio.run(drive())
    for rcb in rcbs:
        rag.add("output_token_ids", rcb.output_token_ids)
    rag.save(out, f"engine{cfg.tag}")
    out[f"engine{cfg.tag}_n_steps"] = np.array([n_steps])
    print(f"engine trace {cfg.tag}: {n_steps} steps, {len(rag.rows['input_ids'])} fill batches, "
          f"kv hit rate {kv.get_metrics().cache_hit_rate:.3f}, allocator left {len(kv.block_allocator.free_blocks)}")


def gen_moe(out):
    """MoE: the reference has no Python implementation of these ops, only CUDA kernels
    and, in tests/kernel/test_moe.py, torch references + assertions.  Here the reference's OWN test
    functions run on CPU — their grids, their torch refs, their assertions (indices exact, weights
    torch.allclose, unpermute 1e-2) — with `hydrainfer._C.kernel.moe` standing in as oracle/moe.py
    (kernel-side call signatures of hydrainfer/_C/kernel/moe/__init__.pyi).  A case is recorded only
    after the reference's assertions passed on it: the fixture holds (inputs, outputs) that the
    reference's own test oracle accepted, and tests/test_oracle_golden.py + tests/test_gpu_moe.py
    hold oracle/moe.py and the HIP kernels to them."""
    import importlib.util
    import itertools
    from oracle import moe as O

    rec = []

    def topk_softmax(gating_logit, weights, indices):
        w, i = O.topk_softmax(gating_logit.float(), weights.shape[1])
        weights.copy_(w); indices.copy_(i)
        rec.append(("topk_softmax", dict(logits=gating_logit.clone()), dict(weights=w.clone(), indices=i.clone())))

    def permute_with_index_map(tokens, topk_ids):
        permuted, _, row_id_map = O.permute_index(tokens, topk_ids)
        rec.append(("permute_index", dict(tokens=tokens.clone(), topk_ids=topk_ids.clone()),
                    dict(permuted=permuted.clone(), row_id_map=row_id_map.clone())))
        return permuted, row_id_map

    def unpermute_with_index_map(permuted, row_id_map, probs):
        o = O.unpermute_rows(permuted, row_id_map, probs)
        rec.append(("unpermute_index", dict(permuted=permuted.clone(), row_id_map=row_id_map.clone(), probs=probs.clone()),
                    dict(out=o.clone())))
        return o

    def permute_with_mask_map(tokens, routing_map, topk):
        permuted, _, row_id_map = O.permute_mask(tokens, routing_map)
        rec.append(("permute_mask", dict(tokens=tokens.clone(), routing_map=routing_map.clone()),
                    dict(permuted=permuted.clone(), row_id_map=row_id_map.clone())))
        return permuted, row_id_map

    def unpermute_with_mask_map(permuted, row_id_map, probs):
        o = O.unpermute_rows(permuted, row_id_map, probs)
        rec.append(("unpermute_mask", dict(permuted=permuted.clone(), row_id_map=row_id_map.clone(), probs=probs.clone()),
                    dict(out=o.clone())))
        return o

    _stub("hydrainfer._C")
    _stub("hydrainfer._C.kernel")
    _stub("hydrainfer._C.kernel.moe", topk_softmax=topk_softmax, permute_with_index_map=permute_with_index_map,
          unpermute_with_index_map=unpermute_with_index_map, permute_with_mask_map=permute_with_mask_map,
          unpermute_with_mask_map=unpermute_with_mask_map)
    spec = importlib.util.spec_from_file_location("ref_test_moe", os.path.join(REFERENCE, "tests", "kernel", "test_moe.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    class TorchOnCpu:                      # the reference test hard-codes torch.device('cuda:0')
        def __getattr__(self, name):
            return getattr(torch, name)

        def device(self, *a, **k):
            return torch.device("cpu")
    ref.torch = TorchOnCpu()
    cpu = torch.device("cpu")
    kept = []

    def run(fn, keep, seed, **kw):
        rec.clear()
        torch.manual_seed(seed)
        fn(**kw)                            # the reference's assertions
        if keep:
            kept.extend((op, dict(kw), i_, o_) for op, i_, o_ in rec)

    n = 0
    # grids of tests/kernel/test_moe.py:7-10, :40-45, :103-108 (all run; the small ones are kept as data)
    for n_tokens, n_experts, topk in itertools.product([1, 10, 16, 128, 1024], [4, 8, 16, 32, 64, 128, 256], [1, 2, 4]):
        run(ref.test_topk_softmax, n_tokens <= 128, 1000 + n, n_tokens=n_tokens, n_experts=n_experts, topk=topk,
            dtype=torch.float)
        n += 1
    for n_tokens, dim, n_experts, topk, dtype in itertools.product([1, 2, 16], [16, 64], [4, 8, 16], [1, 2, 4],
                                                                   [torch.float, torch.half, torch.bfloat16]):
        run(ref.test_permute_index, True, 2000 + n, n_tokens=n_tokens, dim=dim, n_experts=n_experts, topk=topk,
            dtype=dtype, device=cpu)
        run(ref.test_permute_mask, True, 3000 + n, n_tokens=n_tokens, dim=dim, n_experts=n_experts, topk=topk,
            dtype=dtype, device=cpu)
        n += 1
    names = {torch.float32: "f32", torch.float16: "f16", torch.bfloat16: "bf16", torch.int32: "i32", torch.bool: "b",
             torch.int64: "i64"}

    def put(key, t):
        out[key + "." + names[t.dtype]] = C.to_np(t) if hasattr(C, "to_np") else (
            t.view(torch.int16).numpy() if t.dtype == torch.bfloat16 else t.numpy())
    out["n_cases"] = np.array(len(kept))
    for k, (op, kw, ins, outs) in enumerate(kept):
        out[f"c{k}.op"] = np.array(op)
        out[f"c{k}.topk"] = np.array(kw["topk"])
        for name, t in ins.items():
            put(f"c{k}.in.{name}", t)
        for name, t in outs.items():
            put(f"c{k}.out.{name}", t)
    print(f"moe: {n} reference test invocations passed their own assertions; {len(kept)} op calls recorded")


def gen_api_protocol(out):
    """G13: the wire contract of the chat-completions endpoint from the reference's own classes:
    the stream chunks its pydantic models emit (`model_dump_json(exclude_unset=True)`, api_server.py:119-146), what
    APIServer._parse_content makes of a message with an image (api_server.py:62-79), and the prompt its LLaVA chat template
    renders (model/chat_template/template_llava.jinja through LlavaTokenizer.apply_chat_template, llava.py:168-175)."""
    from jinja2 import Template
    sys.modules["shortuuid"].random = lambda: "0" * 22
    sys.modules["hydra"].main = lambda *a, **k: (lambda f: f)        # entrypoint/__init__.py decorates its main with it
    from hydrainfer.entrypoint.api_protocol import (ChatCompletionContent, ChatCompletionImageURL, ChatCompletionMessage,
                                                    ChatCompletionResponseStreamChoice, ChatCompletionStreamResponse,
                                                    DeltaMessage)
    rid, created, model = C.API_CASE["id"], C.API_CASE["created"], C.API_CASE["model"]
    first = ChatCompletionStreamResponse(id=rid, object="chat.completion.chunk", created=created, model=model,
                                         choices=[ChatCompletionResponseStreamChoice(index=0, delta=DeltaMessage(role="assistant", content=""))])
    out["first_chunk"] = np.array(f"data: {first.model_dump_json(exclude_unset=True)}\n\n")
    chunks = []
    for piece in C.API_CASE["pieces"]:
        r = ChatCompletionStreamResponse(id=rid, object="chat.completion.chunk", created=created, model=model,
                                         choices=[ChatCompletionResponseStreamChoice(index=0, delta=DeltaMessage(content=piece))])
        chunks.append(f"data: {r.model_dump_json(exclude_unset=True)}\n\n")
    out["content_chunks"] = np.array(chunks)
    # _parse_content on the client's message (benchmark/backend.py:17-28: text first, then the image)
    from types import SimpleNamespace as NS
    try:
        from hydrainfer.entrypoint.api_server import APIServer
        parse = APIServer._parse_content
    except Exception as e:       # the module pulls in the whole serving stack: fall back to nothing, loudly
        raise RuntimeError(f"cannot import the reference's api_server: {e!r}")
    for name, msg in C.api_messages().items():
        m = ChatCompletionMessage(role=msg["role"], content=[
            ChatCompletionContent(type=c["type"], text=c.get("text"),
                                  image_url=ChatCompletionImageURL(url=c["image_url"]["url"]) if "image_url" in c else None)
            for c in msg["content"]])
        images = parse(NS(vision_config=NS(image_token="<image>")), [m])
        out[f"parsed_{name}_content"] = np.array(m.content)
        out[f"parsed_{name}_n_images"] = np.array(len(images))
        tpl = Template(open(os.path.join(REFERENCE, "hydrainfer", "model", "chat_template", "template_llava.jinja"), encoding="utf-8").read())
        out[f"prompt_{name}"] = np.array(tpl.render(messages=[{"role": m.role, "content": m.content}], bos_token="<s>",
                                                    eos_token="</s>", add_generation_prompt=True))


def main():
    import_reference()
    torch.manual_seed(0)
    sets = {
        "g1_cache_scatter": gen_kv_cache,
        "g2_paged_attention": gen_paged_attention,
        "g3_dense_attention": gen_dense_attention,
        "g4_rms_norm": gen_rms_norm,
        "g5_rope": gen_rope,
        "g6_silu": gen_silu,
        "g7_trace": gen_trace,
        "g8_tiny_llama": gen_tiny_llama,
        "g9_tiny_clip": gen_tiny_clip,
        "g10_tiny_llava": gen_tiny_llava,
        "g11_engine_trace": gen_engine_trace,
        "g12_moe": gen_moe,
        "g13_api_protocol": gen_api_protocol,
    }
    only = sys.argv[1:]
    for name, fn in sets.items():
        if only and name not in only:
            continue
        out = {}
        fn(out)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
