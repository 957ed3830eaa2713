"""-m gpu: the BENCHMARKED configuration held to the oracle directly: bf16, 32 rows, the whole
decode step replayed from a hipGraph, default flags — packed weights, activations-in-registers GEMMs, add+RMSNorm
folded into the gate|up / qkv launches (in-kernel hand-over), fused slab-reduce + RoPE + append + attention, embedding
+ norm and argmax step edges — on a 2-layer model of LLaVA-1.5-7B's width (hidden 4096, 32 heads x 128, inter 11008,
vocab 32064) against oracle/model.py (the reference's eager torch path restated,
hydrainfer/model/model_forward.py:66-105, llama.py:88-104) on the same weights, prompts and block tables.
Also: a hand-over that gives up must be LOUD in the product path (runner and engine), never silent garbage."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

# stated tolerance, bf16: logits (|logit| ~ 3) within 1.5e-1 of the fp32-accumulating oracle, greedy tokens
# identical wherever the oracle's top-1 margin exceeds 2 x that; KV pool within two bf16 ulps of the oracle's
LOGIT_TOL = 1.5e-1
KV_RTOL = 2.0 ** -6      # of the pool's largest magnitude: two bf16 ulps up there (one from the projection's accumulation
                         # order, one from RoPE's T arithmetic on it; RoPE's x*c - y*s cancels, so no per-element bound)


def _build(batch=32, prompt_len=40, n_generate=12, layers=2, seed=3, width=(4096, 11008, 32), executor="plan"):
    from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
    hidden, inter, heads = width
    shape = LlamaShape(hidden, inter, layers, heads, heads, 128, 32064)
    model = LlamaForCausalLM.random_init(shape, torch.bfloat16, DEV, seed=seed)
    runner = DecodeRunner(model, RunnerConfig(batch=batch, prompt_len=prompt_len, n_generate=n_generate, use_graph=True,
                                              executor=executor), seed=seed + 1)
    return shape, model, runner


# both widths bench.py reports (LLaVA-1.5-7B: the headline; LLaVA-1.5-13B: the `llava_13b` leg, BASELINE configs[2]) and
# both step executors (the launch plan bench.py / DecodeRunner replay by default, the hipGraph the engine replays)
@pytest.mark.parametrize("executor", ["plan", "graph"])
@pytest.mark.parametrize("width", [(4096, 11008, 32), (5120, 13824, 40)], ids=["7b-width", "13b-width"])
def test_benchmarked_decode_configuration_matches_oracle(width, executor):
    from hydrainfer_amd import launch_plan
    from oracle.model import OracleAttnMeta, OracleLlama
    B, P, steps, bs = 32, 40, 8, 16
    shape, model, runner = _build(B, P, steps + 4, width=width, executor=executor)
    # the flags bench.py runs with, and the layouts they imply
    assert model.use_hip_gemm and model.use_packed and model.use_xreg and model.xreg_qkv and model.fuse_norm
    assert model.fuse_decode_attention
    assert "l1.wqkv" in model.packed_x and "l0.wgu" in model.packed_x and "l0.wo" in model.packed
    oracle = OracleLlama(shape, model.to_reference_state_dict(), torch.bfloat16)
    pool0 = runner.pool.cpu().clone()
    g = torch.Generator().manual_seed(11)
    prompts = torch.randint(5, 32000, (B, P), generator=g)

    # capture the logits of every replay: forward_logits' result is a static buffer of the graph
    stash = {}
    orig = model.forward_logits

    def spy(*a, **k):
        stash["logits"] = orig(*a, **k)
        return stash["logits"]
    model.forward_logits = spy
    first = runner.prefill(prompts.to(DEV))
    hip_logits, hip_tokens = [], [first.cpu()]
    for _ in range(steps):
        runner.step()
        torch.cuda.synchronize()
        hip_logits.append(stash["logits"].float().cpu().clone())
        hip_tokens.append(runner.input_ids.cpu().clone())
    assert runner.graph is not None and runner.executor_used == executor     # the steps were replays, by the executor asked for
    assert isinstance(runner.graph, launch_plan.LaunchPlan) == (executor == "plan")
    assert model.xreg_sync is not None and not model.handover_failed()   # 5-launch layers ran, no hand-over gave up
    generated = runner.generated()                        # raises on a failed hand-over

    # ---- oracle on the same block tables, teacher-forced with the HIP path's tokens
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)
    caches = [(pool0[l, 0], pool0[l, 1]) for l in range(shape.num_hidden_layers)]
    tables = runner.tables
    n_pb = (P + bs - 1) // bs
    slots = [tables[r][p // bs] * bs + p % bs for r in range(B) for p in range(P)]
    meta = OracleAttnMeta(i32([P * r for r in range(B + 1)]), i32([P * r for r in range(B + 1)]), i32(slots),
                          i32([b for r in range(B) for b in tables[r][:n_pb]]), i32([n_pb * r for r in range(B + 1)]))
    sel = torch.arange(P - 1, B * P, P)
    with torch.inference_mode():
        ref = oracle.forward_logits(prompts.reshape(-1), i32(list(range(P)) * B), meta, caches, sel).float()
    srt = ref.sort(dim=-1).values
    clear = (srt[:, -1] - srt[:, -2]) > 2 * LOGIT_TOL
    assert (hip_tokens[0][clear] == ref.argmax(-1)[clear]).all(), "prefill: greedy token differs despite a clear margin"
    n_checked, worst = int(clear.sum()), 0.0
    for s in range(steps):
        ctx = P + s + 1
        pos = ctx - 1
        nb = (ctx + bs - 1) // bs
        meta = OracleAttnMeta(i32(list(range(B + 1))), i32([ctx * r for r in range(B + 1)]),
                              i32([tables[r][pos // bs] * bs + pos % bs for r in range(B)]),
                              i32([b for r in range(B) for b in tables[r][:nb]]), i32([nb * r for r in range(B + 1)]))
        with torch.inference_mode():
            ref = oracle.forward_logits(hip_tokens[s], i32([pos] * B), meta, caches).float()
        err = (hip_logits[s] - ref).abs().max().item()
        worst = max(worst, err)
        assert err <= LOGIT_TOL, f"decode step {s}: logits max abs err {err} > {LOGIT_TOL}"
        srt = ref.sort(dim=-1).values
        clear = (srt[:, -1] - srt[:, -2]) > 2 * LOGIT_TOL
        assert (hip_tokens[s + 1][clear] == ref.argmax(-1)[clear]).all(), \
            f"decode step {s}: greedy token differs despite a clear margin"
        n_checked += int(clear.sum())
    assert n_checked >= 8 * steps, "too few rows with a clear top-1 margin for the token check to mean anything"
    # the KV pool the graph steps appended to == the oracle's up to bf16 round-off (untouched blocks bit-equal)
    pool_h = runner.pool.cpu()
    pool_o = torch.stack([torch.stack(c) for c in caches]).float()
    assert (pool_h.float() - pool_o).abs().max().item() <= KV_RTOL * pool_o.abs().max().item()
    assert generated.shape == (steps + 1, B)


@pytest.mark.parametrize("B", [33, 64])
def test_wide_decode_layer_matches_oracle(B):
    """33 .. 64 rows (the reference's layers have no batch limit: hydrainfer/model/llama.py:24-27,48-50,
    model_forward.py:29-37): 6 launches per layer on the activations-in-registers layout (norm + gate|up to two slabs,
    silu*mul, down, norm + qkv; no LDS-slice copies of those weights) — held to oracle/model.py like the 32-row
    configuration above: logits within the tolerance, greedy tokens identical where the margin is clear, KV pool within
    two bf16 ulps.  2 layers of 7B width, hipGraph replay."""
    from oracle.model import OracleAttnMeta, OracleLlama
    P, steps, bs = 24, 4, 16
    shape, model, runner = _build(B, P, steps + 4, executor="graph")
    assert model._wide_ok(B) and "l0.wgu" not in model.packed and "l1.wqkv" not in model.packed      # no LDS-slice copies
    dp = model._decode_plan(B, torch.bfloat16)
    assert dp["wide"] and dp["nf_gu"] and dp["nf_qkv"] and not dp["fused"]
    oracle = OracleLlama(shape, model.to_reference_state_dict(), torch.bfloat16)
    pool0 = runner.pool.cpu().clone()
    g = torch.Generator().manual_seed(12)
    prompts = torch.randint(5, 32000, (B, P), generator=g)
    stash = {}
    orig = model.forward_logits

    def spy(*a, **k):
        stash["logits"] = orig(*a, **k)
        return stash["logits"]
    model.forward_logits = spy
    first = runner.prefill(prompts.to(DEV))
    hip_logits, hip_tokens = [], [first.cpu()]
    for _ in range(steps):
        runner.step()
        torch.cuda.synchronize()
        hip_logits.append(stash["logits"].float().cpu().clone())
        hip_tokens.append(runner.input_ids.cpu().clone())
    assert model.xreg_sync is not None and not model.handover_failed()
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)
    caches = [(pool0[l, 0], pool0[l, 1]) for l in range(shape.num_hidden_layers)]
    tables = runner.tables
    n_pb = (P + bs - 1) // bs
    slots = [tables[r][p // bs] * bs + p % bs for r in range(B) for p in range(P)]
    meta = OracleAttnMeta(i32([P * r for r in range(B + 1)]), i32([P * r for r in range(B + 1)]), i32(slots),
                          i32([b for r in range(B) for b in tables[r][:n_pb]]), i32([n_pb * r for r in range(B + 1)]))
    with torch.inference_mode():
        oracle.forward_logits(prompts.reshape(-1), i32(list(range(P)) * B), meta, caches, torch.arange(P - 1, B * P, P))
    n_checked = 0
    for s in range(steps):
        ctx = P + s + 1
        pos = ctx - 1
        nb = (ctx + bs - 1) // bs
        meta = OracleAttnMeta(i32(list(range(B + 1))), i32([ctx * r for r in range(B + 1)]),
                              i32([tables[r][pos // bs] * bs + pos % bs for r in range(B)]),
                              i32([b for r in range(B) for b in tables[r][:nb]]), i32([nb * r for r in range(B + 1)]))
        with torch.inference_mode():
            ref = oracle.forward_logits(hip_tokens[s], i32([pos] * B), meta, caches).float()
        err = (hip_logits[s] - ref).abs().max().item()
        assert err <= LOGIT_TOL, f"decode step {s}: logits max abs err {err} > {LOGIT_TOL}"
        srt = ref.sort(dim=-1).values
        clear = (srt[:, -1] - srt[:, -2]) > 2 * LOGIT_TOL
        assert (hip_tokens[s + 1][clear] == ref.argmax(-1)[clear]).all(), f"decode step {s}: greedy token differs"
        n_checked += int(clear.sum())
    assert n_checked >= 4 * steps
    pool_o = torch.stack([torch.stack(c) for c in caches]).float()
    assert (runner.pool.cpu().float() - pool_o).abs().max().item() <= KV_RTOL * pool_o.abs().max().item()


def test_handover_give_up_is_loud_in_runner_and_engine():
    """Test hook xreg_no_producers = 2: the norm-fused launches get no producers and no rescue, so every one of them
    gives up (2 ms bound under the hook) and leaves its error word.  (The hook is a launch argument: it has to be set
    when the step is captured / recorded.)  The runner must refuse to hand out that run's tokens, the engine's
    decoder must raise from fetch(), switch the model to separate norm launches and keep serving."""
    from hydrainfer_amd import _lib
    lib = _lib.lib()
    for executor in ("graph", "plan"):
        from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
        from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
        shape = LlamaShape(4096, 11008, 2, 32, 32, 128, 32064)
        model = LlamaForCausalLM.random_init(shape, torch.bfloat16, DEV, seed=5)
        runner = DecodeRunner(model, RunnerConfig(batch=8, prompt_len=24, n_generate=8, use_graph=True, executor=executor), seed=6)
        g = torch.Generator().manual_seed(1)
        runner.prefill(torch.randint(5, 32000, (8, 24), generator=g).to(DEV))
        try:
            assert lib.hx_debug_set_option(b"xreg_no_producers", 2) == 0
            runner.step(); torch.cuda.synchronize()        # captures / records under the hook, then replays once
        finally:
            lib.hx_debug_set_option(b"xreg_no_producers", 0)
        assert model.handover_failed()
        with pytest.raises(_lib.HydraHipError, match="gave up"):
            runner.generated()
        assert model.fuse_norm is False
        del runner, model

    # ---- engine path: GraphedDecoder.fetch (its default executor)
    from hydrainfer_amd.engine.graph_decode import GraphedDecoder
    from hydrainfer_amd.memory.token_cache_manger import (TokenCacheBlockManager, TokenCacheBlockManagerConfig,
                                                          TokenCacheBlockManagerContext)
    from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.llava import LlavaLanguageModel
    shape = LlamaShape(4096, 11008, 2, 32, 32, 128, 32064)
    model = LlamaForCausalLM.random_init(shape, torch.bfloat16, DEV, seed=5)
    kv = TokenCacheBlockManager(TokenCacheBlockManagerConfig(
        n_layers=shape.num_hidden_layers, n_tokens=2, n_blocks=64, block_size=16, n_heads=shape.num_key_value_heads,
        head_size=shape.head_dim, dtype="bf16", device=str(DEV)), TokenCacheBlockManagerContext(rank=0, rank2host={0: "localhost"}))
    dec = GraphedDecoder(LlavaLanguageModel(model, image_token_id=32000), kv, max_batch=8, max_blocks_per_seq=4, executor="plan")
    vc = kv.allocate_virtual_cache()
    kv.realloc(vc, 3)
    row0 = (17, 0, vc.block_table[0] * 16, 1, list(vc.block_table))
    row1 = (23, 1, vc.block_table[0] * 16 + 1, 2, list(vc.block_table))
    try:
        assert lib.hx_debug_set_option(b"xreg_no_producers", 2) == 0
        lid = dec.launch([row0])                      # first launch of this batch size: captured under the hook
        torch.cuda.synchronize()
    finally:
        lib.hx_debug_set_option(b"xreg_no_producers", 0)
    with pytest.raises(_lib.HydraHipError, match="hand-over"):
        dec.fetch(lid)
    assert model.fuse_norm is False and not dec.graphs
    assert len(dec.run([row0])) == 1 and len(dec.run([row1])) == 1      # recaptured without fusion: serving goes on
