"""Seeded input builders shared by tests/golden/generate_goldens.py (which runs the
REFERENCE on them, in the build container only) and by the tests (which run the oracle
and the HIP path on the very same inputs).  Inputs are regenerated from seeds with
torch's CPU generator — identical for the same torch build, and every fixture stores an
input checksum so a drift would be caught rather than silently compared."""
import hashlib
import math
from dataclasses import dataclass
from typing import Dict, List

import numpy as np
import torch

DTYPES = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}


def checksum(*tensors: torch.Tensor) -> str:
    h = hashlib.sha256()
    for t in tensors:
        t = t.detach().contiguous()
        if t.dtype == torch.bfloat16:
            t = t.view(torch.int16)
        h.update(t.numpy().tobytes())
    return h.hexdigest()[:16]


def to_np(t: torch.Tensor) -> np.ndarray:
    t = t.detach().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy()
    return t.numpy()


def from_np(a: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype == torch.bfloat16:
        return t.view(torch.bfloat16)
    return t


# ------------------------------------------------------------------ G1 cache scatter
def kv_cache_cases() -> List[Dict]:
    # grid of the reference's tests/memory/test_kv_cache.py:6-12
    cases = []
    for block_size in (4, 8, 16):
        for n_heads, head_dim, n_tokens in ((8, 64, 1), (4, 128, 15), (2, 256, 64), (1, 128, 100)):
            for dt in ("fp16", "bf16", "fp32"):
                cases.append(dict(n_blocks=100, block_size=block_size, n_heads=n_heads,
                                  head_dim=head_dim, n_tokens=n_tokens, dtype=dt))
    return cases


def kv_cache_inputs(case: Dict, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    dt = DTYPES[case["dtype"]]
    shape = (case["n_blocks"], case["block_size"], case["n_heads"], case["head_dim"])
    key_cache = torch.randn(shape, generator=g).to(dt)
    value_cache = torch.randn(shape, generator=g).to(dt)
    n_slots = case["n_blocks"] * case["block_size"]
    slot_ids = torch.randperm(n_slots, generator=g)[: case["n_tokens"]].to(torch.int32)
    # keys/values are strided views of a fused qkv projection, as in model_forward.py:66-75
    qkv = torch.randn((case["n_tokens"], 3 * case["n_heads"] * case["head_dim"]), generator=g).to(dt)
    hd = case["n_heads"] * case["head_dim"]
    keys = qkv[:, hd: 2 * hd].view(-1, case["n_heads"], case["head_dim"])
    values = qkv[:, 2 * hd:].view(-1, case["n_heads"], case["head_dim"])
    return slot_ids, keys, values, key_cache, value_cache


# ------------------------------------------------------------------ G2 paged attention
def paged_attention_cases() -> List[Dict]:
    cases = []
    # tests/layer/test_attention.py:42-49 seq pairs; both (1,100)+(15,15) style batches
    seq_sets = [
        [(1, 100), (15, 15), (111, 234), (1, 1024)],
        [(1, 1), (1, 17), (1, 16), (1, 255)],          # all-decode batch (decode kernel)
        [(33, 33), (64, 64), (7, 71)],
    ]
    # Fixture size: the reference grid uses 8 query heads (tests/layer/test_attention.py:46);
    # heads are independent, so the two multi-hundred-token sets use 4 query heads and the
    # tiny all-decode set keeps 8.
    for si, seqs in enumerate(seq_sets):
        hq = 8 if si == 1 else 4
        for n_kv_heads in (hq, 2, 1):
            for head_dim in (64, 128, 256):
                for dt in ("fp16", "bf16"):
                    if head_dim == 256 and si == 0 and not (n_kv_heads == hq and dt == "fp16"):
                        continue
                    if si == 2 and (head_dim == 256 or n_kv_heads == 1):
                        continue
                    cases.append(dict(seqs=seqs, n_heads=hq, n_kv_heads=n_kv_heads,
                                      head_dim=head_dim, dtype=dt, block_size=16, n_blocks=128))
    cases.append(dict(seqs=[(40, 40), (1, 90)], n_heads=4, n_kv_heads=4, head_dim=128, dtype="fp16",
                      block_size=32, n_blocks=16))
    return cases


def paged_attention_inputs(case: Dict, seed: int = 0):
    """Returns q,k,v (new tokens), caches (pre-filled with history), and the metadata lists."""
    g = torch.Generator().manual_seed(seed)
    dt = DTYPES[case["dtype"]]
    bs, nb = case["block_size"], case["n_blocks"]
    H, HK, D = case["n_heads"], case["n_kv_heads"], case["head_dim"]
    key_cache = torch.randn((nb, bs, HK, D), generator=g).to(dt)
    value_cache = torch.randn((nb, bs, HK, D), generator=g).to(dt)
    perm = torch.randperm(nb, generator=g).tolist()
    n_tokens = sum(q for q, _ in case["seqs"])
    q = torch.randn((n_tokens, H * D), generator=g).to(dt)
    k = torch.randn((n_tokens, HK * D), generator=g).to(dt)
    v = torch.randn((n_tokens, HK * D), generator=g).to(dt)
    reqs = []
    used = 0
    for q_len, kv_len in case["seqs"]:
        n_need = (kv_len + bs - 1) // bs
        table = perm[used: used + n_need]
        used += n_need
        slots = [table[p // bs] * bs + p % bs for p in range(kv_len - q_len, kv_len)]
        reqs.append(dict(q_len=q_len, kv_len=kv_len, block_table=table, new_cache_slots=slots))
    assert used <= nb
    return q, k, v, key_cache, value_cache, reqs


# ------------------------------------------------------------------ G3 dense attention
def dense_attention_cases() -> List[Dict]:
    return [
        # CLIP ViT-L/14-336 tile: 577 tokens, D=64 (16 heads in the model; heads are independent)
        dict(batch=1, seq_len=577, n_heads=8, head_dim=64, dtype="fp16"),
        dict(batch=2, seq_len=577, n_heads=4, head_dim=64, dtype="bf16"),
        dict(batch=3, seq_len=50, n_heads=4, head_dim=128, dtype="fp16"),
        dict(batch=2, seq_len=33, n_heads=2, head_dim=32, dtype="bf16"),
        dict(batch=2, seq_len=70, n_heads=2, head_dim=96, dtype="fp16"),
    ]


def dense_attention_inputs(case: Dict, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    dt = DTYPES[case["dtype"]]
    hidden = case["n_heads"] * case["head_dim"]
    shape = (case["batch"], case["seq_len"], hidden)
    q = torch.randn(shape, generator=g).to(dt)
    k = torch.randn(shape, generator=g).to(dt)
    v = torch.randn(shape, generator=g).to(dt)
    return q, k, v


# ------------------------------------------------------------------ G4 rms_norm
def rms_norm_cases() -> List[Dict]:
    cases = []
    for hidden in (1, 2, 32, 333, 334, 1024, 4096, 5120):  # test_rms_norm_kernel.py:6-10 + LLaVA
        for dt in ("fp32", "fp16", "bf16"):
            cases.append(dict(rows=7 if hidden >= 1024 else 33, hidden=hidden, dtype=dt, eps=1e-5))
    return cases


def rms_norm_inputs(case: Dict, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    dt = DTYPES[case["dtype"]]
    x = torch.randn((case["rows"], case["hidden"]), generator=g).to(dt)
    w = (1.0 + 0.1 * torch.randn((case["hidden"],), generator=g)).to(dt)
    return x, w


# ------------------------------------------------------------------ G5 rope
def rope_cases() -> List[Dict]:
    cases = []
    for n_kv_heads in (32, 8, 1):  # tests/layer/test_rotary_embedding.py:111-120
        for theta in (1e4, 5e5):
            for interleaved in (False, True):
                for dt in ("fp16", "bf16", "fp32"):
                    cases.append(dict(n_tokens=4, n_heads=32, n_kv_heads=n_kv_heads, head_dim=128,
                                      rotary_dim=128, theta=theta, interleaved=interleaved,
                                      max_pos=4096, dtype=dt))
    cases.append(dict(n_tokens=5, n_heads=4, n_kv_heads=2, head_dim=64, rotary_dim=32, theta=1e4,
                      interleaved=False, max_pos=128, dtype="fp16"))
    cases.append(dict(n_tokens=5, n_heads=4, n_kv_heads=2, head_dim=64, rotary_dim=32, theta=1e4,
                      interleaved=True, max_pos=128, dtype="bf16"))
    return cases


def rope_inputs(case: Dict, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    dt = DTYPES[case["dtype"]]
    q = torch.randn((case["n_tokens"], case["n_heads"], case["head_dim"]), generator=g).to(dt)
    k = torch.randn((case["n_tokens"], case["n_kv_heads"], case["head_dim"]), generator=g).to(dt)
    pos = torch.randint(0, case["max_pos"], (case["n_tokens"],), generator=g).to(torch.int32)
    return q, k, pos


# ------------------------------------------------------------------ G6 silu
def silu_cases() -> List[Dict]:
    cases = []
    for n in (1, 2, 32, 333, 334, 1024, 4096, 11008):  # tests/kernel/test_activation.py:6-9 + LLaVA
        for dt in ("fp32", "fp16", "bf16"):
            cases.append(dict(rows=5, n=n, dtype=dt))
    return cases


def silu_inputs(case: Dict, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    x = (3.0 * torch.randn((case["rows"], case["n"]), generator=g)).to(DTYPES[case["dtype"]])
    return x


# ------------------------------------------------------------------ G7 integer trace
@dataclass
class TraceConfig:
    n_requests: int = 32
    prompt_len: int = 704      # 576 image + 128 text
    n_decode: int = 6          # decode steps recorded (full run = 255)
    block_size: int = 16
    n_blocks: int = 2048


def trace_token_ids(cfg: TraceConfig, req: int) -> List[int]:
    g = torch.Generator().manual_seed(1000 + req)
    text = torch.randint(1000, 31999, (cfg.prompt_len - 576,), generator=g).tolist()
    return [32000] * 576 + text


def case_name(prefix: str, i: int) -> str:
    return f"{prefix}_{i:03d}"


# ------------------------------------------------------------------ G8 tiny Llama end-to-end
TINY_LLAMA = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                  num_key_value_heads=2, head_dim=128, vocab_size=512, rms_norm_eps=1e-5,
                  rope_theta=10000.0, max_position_embeddings=4096)
TINY_PROMPTS = (40, 23)   # two requests prefilled together, then decoded together
TINY_DECODE_STEPS = 6
TINY_BLOCKS, TINY_BLOCK_SIZE = 16, 16


def tiny_llama_state_dict(dtype: torch.dtype, seed: int = 0, std: float = 0.08) -> Dict[str, torch.Tensor]:
    """Reference-named (HF Llama) random weights; std chosen so logits have usable margins."""
    g = torch.Generator().manual_seed(seed)
    t = TINY_LLAMA
    h, i, L = t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"]
    q = t["num_attention_heads"] * t["head_dim"]
    kv = t["num_key_value_heads"] * t["head_dim"]

    def w(*size):
        return (torch.randn(size, generator=g) * std).to(dtype)

    sd = {"model.embed_tokens.weight": (torch.randn((t["vocab_size"], h), generator=g)).to(dtype),
          "lm_head.weight": w(t["vocab_size"], h),
          "model.norm.weight": (1 + 0.1 * torch.randn(h, generator=g)).to(dtype)}
    for l in range(L):
        p = f"model.layers.{l}."
        sd[p + "self_attn.q_proj.weight"] = w(q, h)
        sd[p + "self_attn.k_proj.weight"] = w(kv, h)
        sd[p + "self_attn.v_proj.weight"] = w(kv, h)
        sd[p + "self_attn.o_proj.weight"] = w(h, q)
        sd[p + "mlp.gate_proj.weight"] = w(i, h)
        sd[p + "mlp.up_proj.weight"] = w(i, h)
        sd[p + "mlp.down_proj.weight"] = w(h, i)
        sd[p + "input_layernorm.weight"] = (1 + 0.1 * torch.randn(h, generator=g)).to(dtype)
        sd[p + "post_attention_layernorm.weight"] = (1 + 0.1 * torch.randn(h, generator=g)).to(dtype)
    return sd


def tiny_prompt_ids(req: int) -> List[int]:
    g = torch.Generator().manual_seed(500 + req)
    return torch.randint(0, TINY_LLAMA["vocab_size"], (TINY_PROMPTS[req],), generator=g).tolist()


def tiny_block_tables() -> List[List[int]]:
    """LIFO allocator order for prompt+decode tokens (all blocks of a request taken at once)."""
    tables, nxt = [], TINY_BLOCKS - 1
    for p in TINY_PROMPTS:
        n = (p + TINY_DECODE_STEPS + TINY_BLOCK_SIZE - 1) // TINY_BLOCK_SIZE
        tables.append([nxt - j for j in range(n)][::-1])
        nxt -= n
    return tables


# ------------------------------------------------------------------ G9 tiny CLIP + projector
TINY_CLIP = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                 image_size=56, patch_size=14, num_channels=3, layer_norm_eps=1e-5,
                 vision_feature_layer=-2, projector_hidden_size=256)


def tiny_clip_pixels(n_images: int = 2) -> torch.Tensor:
    # the reference's synthetic image generator (hydrainfer/utils/image_utils.py:4-7), CLIP-normalised
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (n_images, TINY_CLIP["image_size"], TINY_CLIP["image_size"], 3)).astype(np.float32)
    mean = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
    std = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)
    x = (img / 255.0 - mean) / std
    return torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------ G10 tiny LLaVA end-to-end
TINY_IMAGE_TOKEN_ID = TINY_LLAMA["vocab_size"] - 1
TINY_LLAVA_TEXT = (10, 7)      # text tokens after the 16 image tokens of each request


def tiny_llava_prompt(req: int) -> List[int]:
    n_img = (TINY_CLIP["image_size"] // TINY_CLIP["patch_size"]) ** 2
    g = torch.Generator().manual_seed(900 + req)
    text = torch.randint(0, TINY_IMAGE_TOKEN_ID, (TINY_LLAVA_TEXT[req],), generator=g).tolist()
    return [TINY_IMAGE_TOKEN_ID] * n_img + text


def tiny_llava_block_tables() -> List[List[int]]:
    tables, nxt = [], TINY_BLOCKS - 1
    for r in range(2):
        n = (len(tiny_llava_prompt(r)) + TINY_DECODE_STEPS + TINY_BLOCK_SIZE - 1) // TINY_BLOCK_SIZE
        tables.append([nxt - j for j in range(n)][::-1])
        nxt -= n
    return tables


# ---- G11: engine trace (scheduler + parameter builder + executor bookkeeping) -----------------
@dataclass
class EngineTraceConfig:
    tag: str
    priority: str
    chunked_prefill: bool
    token_budgets: int
    image_budgets: int
    max_running_requests: int
    kv_blocks: int
    image_blocks: int
    n_image_tokens: int = 40     # per image; also the image cache block size
    block_size: int = 16
    n_layers: int = 2
    n_heads: int = 1
    head_dim: int = 8
    image_token_id: int = 32000


@dataclass
class EngineTraceRequest:
    arrival_step: int
    token_ids: List[int]         # one image_token_id per image
    image_seed: int              # -1: text only
    max_tokens: int


ENGINE_TRACES = [
    EngineTraceConfig("A", "prefill", True, 48, 2, 5, 36, 8),
    EngineTraceConfig("B", "decode", False, 80, 1, 6, 31, 6),
    EngineTraceConfig("C", "decode", True, 56, 1, 4, 40, 6),     # decode priority WITH chunked prefill
]


def engine_trace_requests(cfg: EngineTraceConfig) -> List[EngineTraceRequest]:
    g = torch.Generator().manual_seed(77 + ord(cfg.tag))
    reqs: List[EngineTraceRequest] = []

    def text(n):
        return torch.randint(1000, 31999, (n,), generator=g).tolist()

    if cfg.tag == "C":   # image requests only (the reference's TextFill.chunk_prefill raises), staggered
        lens = [25, 9, 31, 14, 22, 6, 28, 17, 11, 20]
        arrive = [0, 0, 2, 2, 3, 8, 8, 9, 15, 15]
        for i, (n, a) in enumerate(zip(lens, arrive)):
            reqs.append(EngineTraceRequest(a, [1] + [cfg.image_token_id] + text(n), 900 + i, 2 + (i * 5) % 9))
        return reqs
    if cfg.tag == "A":   # image requests only; 4, 9 and 12 repeat request 0 -> prefix-cache hits
        lens = [21, 7, 30, 12, 21, 5, 26, 18, 9, 21, 14, 28, 21, 11]
        arrive = [0, 0, 0, 1, 9, 9, 10, 14, 14, 22, 22, 23, 40, 40]
        first = None
        for i, (n, a) in enumerate(zip(lens, arrive)):
            ids = [1] + [cfg.image_token_id] + text(n)
            seed = 500 + i
            if i == 0:
                first = ids
            if i in (4, 9, 12):
                ids, seed = list(first), 500
            reqs.append(EngineTraceRequest(a, ids, seed, 3 + (i * 5) % 7))
    else:                # text-only and image requests mixed, distinct prompts
        lens = [33, 12, 50, 8, 27, 41, 16, 22, 10, 36, 19, 45]
        arrive = [0, 0, 1, 1, 2, 6, 6, 11, 11, 12, 20, 20]
        for i, (n, a) in enumerate(zip(lens, arrive)):
            img = i % 3 == 1
            ids = [1] + ([cfg.image_token_id] if img else []) + text(n)
            reqs.append(EngineTraceRequest(a, ids, 700 + i if img else -1, 2 + (i * 3) % 8))
    return reqs


def engine_trace_image(seed: int):
    """8x8 RGB uint8 image of request `seed` (content only matters through its xxh64)."""
    rng = np.random.RandomState(seed)
    return rng.randint(0, 256, (8, 8, 3), dtype=np.uint8)


def engine_trace_sample(input_id: int, position_id: int) -> int:
    """Stand-in for the language model in the engine trace: the token sampled at a row is a fixed
    function of that row's input id and position."""
    return (position_id * 131 + input_id * 7 + 13) % 30000 + 100


def profiler_weird_criterion(n: int) -> bool:
    """Non-monotonic and sometimes failing, like a real latency measurement near an OOM."""
    if n % 7 == 0:
        raise RuntimeError("out of memory")
    return (n * 37) % 11 < 6


_MOE_DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16, "i32": torch.int32,
           "b": torch.bool, "i64": torch.int64}


def moe_golden_cases(g):
    """Yields (op, topk, inputs, outputs) of tests/golden/g12_moe.npz: calls that passed the
    assertions of the reference's tests/kernel/test_moe.py (generate_goldens.py::gen_moe)."""
    fields = {}
    for key in g.files:
        if key == "n_cases":
            continue
        case, rest = key.split(".", 1)
        fields.setdefault(case, []).append(rest)
    for k in range(int(g["n_cases"])):
        c = f"c{k}"
        ins, outs = {}, {}
        for rest in fields[c]:
            if rest in ("op", "topk"):
                continue
            side, name, dt = rest.split(".")
            (ins if side == "in" else outs)[name] = from_np(g[f"{c}.{rest}"], _MOE_DT[dt])
        yield str(g[f"{c}.op"]), int(g[f"{c}.topk"]), ins, outs


# ---- G13: chat-completions wire contract (hydrainfer/entrypoint/api_server.py:89-152) -------------------------
API_CASE = {"id": "chatcmpl-0000000000000000000000", "created": 1700000000, "model": "llava-hf/llava-1.5-7b-hf",
            "pieces": [" Hello", "世界", ' "quoted"\n', " <123>", "a/b\\c"]}
_PNG_1x1 = ("iVBORw0KGgoAAAANSUhEUgAAAAEAAAABCAIAAACQd1PeAAAADElEQVR4nGP4z8AAAAMBAQDJ/pLvAAAAAElFTkSuQmCC")


def api_messages():
    """Messages as the reference's client builds them (benchmark/backend.py:17-38): the text first, then the images."""
    url = {"url": "data:image/png;base64," + _PNG_1x1}
    return {"image_text": {"role": "user", "content": [{"type": "text", "text": "What is shown in this image?"},
                                                       {"type": "image_url", "image_url": url}]},
            "text_only": {"role": "user", "content": [{"type": "text", "text": "Describe  the weather.\nBriefly."}]}}
