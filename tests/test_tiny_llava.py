"""G10: tiny LLaVA end-to-end (image -> CLIP tower -> projector -> image cache -> embeddings
overwritten at image-token rows -> prefill -> greedy decode) against tokens/logits produced by
the reference's own modules (tests/golden/generate_goldens.py::gen_tiny_llava)."""
import numpy as np
import pytest
import torch

from tests.golden import cases as C
from tests.util import load_golden

BS = C.TINY_BLOCK_SIZE


def _steps(step_fn):
    tables = C.tiny_llava_block_tables()
    lens, toks, logs = [0, 0], [], []
    new = [C.tiny_llava_prompt(0), C.tiny_llava_prompt(1)]
    for s in range(C.TINY_DECODE_STEPS):
        ids, pos, sel, slots, q_lens, cur_tables = [], [], [], [], [], []
        n = 0
        for r, x in enumerate(new):
            slots += [tables[r][p // BS] * BS + p % BS for p in range(lens[r], lens[r] + len(x))]
            pos += list(range(lens[r], lens[r] + len(x)))
            lens[r] += len(x)
            ids += x
            n += len(x)
            sel.append(n - 1)
            q_lens.append(len(x))
            cur_tables.append(tables[r][: (lens[r] + BS - 1) // BS])
        logits = step_fn(ids, pos, dict(slots=slots, q_lens=q_lens, kv_lens=list(lens), tables=cur_tables), sel,
                         first=(s == 0))
        nxt = logits.argmax(-1).tolist()
        toks.append(nxt)
        logs.append(logits)
        new = [[nxt[0]], [nxt[1]]]
    return np.array(toks), torch.stack(logs).numpy()


def _check(toks, logs, g, dname, tol):
    ref_logits, ref_toks = g[f"llava_{dname}_logits"], g[f"llava_{dname}_tokens"]
    same = (toks == ref_toks).all(axis=1).cumprod() == 1
    n_same = int(same.sum())
    assert n_same >= 1
    err = np.abs(logs[:n_same] - ref_logits[:n_same]).max()
    assert err <= tol, f"logits max abs err {err} > {tol}"
    srt = np.sort(ref_logits, axis=-1)
    ok = (srt[..., -1] - srt[..., -2]) > 2 * tol
    for s in range(C.TINY_DECODE_STEPS):
        if not ok[s].all():
            return
        assert (toks[s] == ref_toks[s]).all(), f"greedy token mismatch at step {s}"


@pytest.mark.parametrize("dname", ["fp16", "bf16"])
def test_oracle_llava_matches_reference(dname):
    from hydrainfer_amd.model.clip import ClipShape, random_state_dict
    from hydrainfer_amd.model.llama import LlamaShape
    from oracle.model import OracleAttnMeta, OracleLlama
    from oracle.vision import vision_forward
    g = load_golden("g10_tiny_llava")
    dt = C.DTYPES[dname]
    cshape = ClipShape(**C.TINY_CLIP)
    csd = {k: v.to(dt) for k, v in random_state_dict(cshape, seed=3, std=0.05).items()}
    feats = vision_forward(cshape, csd, C.tiny_clip_pixels(2))
    sd = C.tiny_llama_state_dict(dt)
    model = OracleLlama(LlamaShape(**C.TINY_LLAMA), sd, dt)
    t = C.TINY_LLAMA
    pool = torch.randn((t["num_hidden_layers"], 2, C.TINY_BLOCKS, BS, t["num_key_value_heads"], t["head_dim"]),
                       generator=torch.Generator().manual_seed(77)).to(dt)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)

    def step(ids, pos, m, sel, first):
        cu_q = [0] + list(np.cumsum(m["q_lens"]))
        cu_k = [0] + list(np.cumsum(m["kv_lens"]))
        bt = [b for tb in m["tables"] for b in tb]
        cu_b = [0] + list(np.cumsum([len(tb) for tb in m["tables"]]))
        meta = OracleAttnMeta(i32(cu_q), i32(cu_k), i32(m["slots"]), i32(bt), i32(cu_b))
        ids_t = torch.tensor(ids, dtype=torch.int64)
        emb = torch.nn.functional.embedding(ids_t, sd["model.embed_tokens.weight"])
        if first:
            emb[ids_t == C.TINY_IMAGE_TOKEN_ID] = feats.reshape(-1, emb.shape[-1])
        caches = [(pool[l, 0], pool[l, 1]) for l in range(pool.shape[0])]
        return model.forward_logits(emb, i32(pos), meta, caches, torch.tensor(sel) if first else None).float()

    toks, logs = _steps(step)
    np.testing.assert_array_equal(toks, g[f"llava_{dname}_tokens"])
    np.testing.assert_allclose(logs, g[f"llava_{dname}_logits"], atol=2e-2 if dname == "bf16" else 4e-3, rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("dname", ["fp16", "bf16"])
def test_hip_llava_end_to_end(dname):
    from hydrainfer_amd.layer.causal_attention import AttentionParametersBuilder
    from hydrainfer_amd.memory.kv_cache import KVCache
    from hydrainfer_amd.memory.token_cache import TokenCache
    from hydrainfer_amd.model.clip import ClipShape, LlavaVisionModel, random_state_dict
    from hydrainfer_amd.model.llama import LanguageModelParameters, LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.llava import LlavaLanguageModel
    g = load_golden("g10_tiny_llava")
    dt, dev = C.DTYPES[dname], torch.device("cuda:0")
    cshape = ClipShape(**C.TINY_CLIP)
    vision = LlavaVisionModel(cshape, dt, dev, {k: v.to(dt).to(dev) for k, v in
                                                  random_state_dict(cshape, seed=3, std=0.05).items()})
    shape = LlamaShape(**C.TINY_LLAMA)
    lm = LlavaLanguageModel(LlamaForCausalLM.from_reference_state_dict(shape, C.tiny_llama_state_dict(dt), dt, dev),
                            image_token_id=C.TINY_IMAGE_TOKEN_ID)
    # encode -> image cache (one block per image, executor.py:228-231) -> read back for prefill
    feats = vision(C.tiny_clip_pixels(2).to(dev))                         # [2, 16, 256]
    n_img, n_tok, hid = feats.shape
    H, D = shape.num_attention_heads, shape.head_dim
    image_cache = torch.zeros((3, n_tok, H, D), dtype=dt, device=dev)
    img_slots = torch.cat([torch.arange(n_tok) + 2 * n_tok, torch.arange(n_tok)]).to(torch.int32).to(dev)
    TokenCache([image_cache]).set_caches(img_slots, [feats.reshape(n_img * n_tok, H, D)])
    cached_feats = image_cache.view(-1, hid)[img_slots.long()]           # parameters_builder gather
    pool = torch.randn((shape.num_hidden_layers, 2, C.TINY_BLOCKS, BS, shape.num_key_value_heads, D),
                       generator=torch.Generator().manual_seed(77)).to(dt).to(dev)

    def step(ids, pos, m, sel, first):
        b = AttentionParametersBuilder(H, shape.num_key_value_heads, D, BS, dev)
        off = 0
        for r in range(2):
            ql = m["q_lens"][r]
            b.add_request(ql, m["kv_lens"][r], m["slots"][off: off + ql], m["tables"][r])
            off += ql
        for l in range(shape.num_hidden_layers):
            b.add_kv_cache(KVCache(pool[l, 0], pool[l, 1]))
        params = LanguageModelParameters(attention_params=b.build_attention_parameters(),
                                         all_sequences_decode=not first,
                                         selected_token_ids=torch.tensor(sel, device=dev) if first else None)
        return lm.forward_logits(torch.tensor(ids, dtype=torch.int64, device=dev),
                                 cached_feats if first else None,
                                 torch.tensor(pos, dtype=torch.int32, device=dev), params).float().cpu()

    toks, logs = _steps(step)
    _check(toks, logs, g, dname, 1.5e-1 if dname == "bf16" else 2e-2)
