"""-m gpu: launch plans (hydrainfer_amd/launch_plan.py, csrc/launch_plan.hip).  A decode step replayed from a plan — the
step's launches recorded once and issued by a native loop, the lm_head GEMM as a host-side step in between — must
produce exactly the tokens and the KV pool of the same step replayed from a captured hipGraph."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _run(executor, dt, shape, batch, steps, seed=3, prompt=40):
    from hydrainfer_amd.model.llama import LlamaForCausalLM
    from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
    model = LlamaForCausalLM.random_init(shape, dt, DEV, seed=seed)
    r = DecodeRunner(model, RunnerConfig(batch=batch, prompt_len=prompt, n_generate=steps + 8, use_graph=True,
                                         executor=executor), seed=seed + 1)
    g = torch.Generator().manual_seed(0)
    r.prefill(torch.randint(5, shape.vocab_size - 1, (batch, prompt), generator=g).to(DEV))
    for _ in range(steps):
        r.step()
    torch.cuda.synchronize()
    assert r.executor_used == executor, "an all-hx decode step must be recordable (no silent fall-back to the hipGraph)"
    return r, r.generated(), r.pool.clone()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("batch", [32, 5])
def test_plan_equals_graph(dt, batch):
    """3 layers of 7B width (the benchmark's launch shapes), 24 steps over the same buffers."""
    from hydrainfer_amd import launch_plan
    from hydrainfer_amd.model.llama import LlamaShape
    sh = LlamaShape(4096, 11008, 3, 32, 32, 128, 32064)
    outs = {}
    for ex in ("graph", "plan"):
        r, toks, pool = _run(ex, dt, sh, batch, 24)
        outs[ex] = (toks, pool)
        if ex == "plan":
            plan = r.graph
            assert isinstance(plan, launch_plan.LaunchPlan)
            # step head (advance + embed + norm + zeroing of the hand-over areas), layer 0's qkv, 5 per layer (attention,
            # o, norm + gate|up, down, norm + next qkv resp. the final norm), argmax; the lm_head GEMM is a host-side step
            assert plan.n_launches == 5 * sh.num_hidden_layers + 3
            assert sum(1 for it in plan.items if callable(it)) == 1
        del r
    assert torch.equal(outs["graph"][0], outs["plan"][0]), "sampled tokens differ from the hipGraph run"
    assert torch.equal(outs["graph"][1], outs["plan"][1]), "KV pool differs from the hipGraph run"


def test_plan_small_and_13b_widths():
    from hydrainfer_amd.model.llama import LlamaShape
    for sh, batch in ((LlamaShape(1024, 2816, 3, 8, 8, 128, 2048), 7), (LlamaShape(5120, 13824, 2, 40, 40, 128, 32064), 32),
                      (LlamaShape(1024, 2816, 2, 8, 8, 128, 2048), 40)):      # 40 rows: the 8-launch LDS-slice layer
        a = _run("graph", torch.bfloat16, sh, batch, 12)
        b = _run("plan", torch.bfloat16, sh, batch, 12)
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


def test_step_with_torch_ops_is_refused_by_the_plan_and_falls_back_to_the_graph():
    """Round-3 ADVICE: a torch op inside a recording ran once and was dropped on replay, silently — `bench.py
    --lib-gemm` (use_hip_gemm = False: every decoder GEMM is torch.matmul) replayed a step without its GEMMs.  The
    recording now raises PlanNotRecordable, DecodeRunner falls back to the captured hipGraph and says so; tokens and
    KV pool equal the eager (launch by launch) run of the same model."""
    from hydrainfer_amd import launch_plan
    from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
    sh = LlamaShape(1024, 2816, 2, 8, 8, 128, 2048)
    outs = {}
    for name, use_graph in (("eager", False), ("plan", True)):
        model = LlamaForCausalLM.random_init(sh, torch.bfloat16, DEV, seed=3)
        model.use_hip_gemm = False
        r = DecodeRunner(model, RunnerConfig(batch=8, prompt_len=40, n_generate=18, use_graph=use_graph, executor="plan"), seed=4)
        g = torch.Generator().manual_seed(0)
        r.prefill(torch.randint(5, sh.vocab_size - 1, (8, 40), generator=g).to(DEV))
        for _ in range(10):
            r.step()
        torch.cuda.synchronize()
        if use_graph:
            assert r.executor_used == "graph" and isinstance(r.graph, torch.cuda.CUDAGraph)
        outs[name] = (r.generated(), r.pool.clone())
    assert torch.equal(outs["eager"][0], outs["plan"][0]) and torch.equal(outs["eager"][1], outs["plan"][1])
    # and the guard itself: a kernel-launching torch op inside a recording raises, allocations and views do not
    plan = launch_plan.LaunchPlan(DEV)
    x = torch.ones(64, device=DEV)
    with pytest.raises(launch_plan.PlanNotRecordable, match="aten::"):
        plan.capture(lambda: x.add_(1))
    assert float(x.sum()) == 64.0 or float(x.sum()) == 128.0        # (the op may or may not have run: it never reaches a replay)
    plan2 = launch_plan.LaunchPlan(DEV)
    plan2.capture(lambda: torch.empty(16, device=DEV).view(4, 4)[1:, :2])


def test_plan_buffers_survive_other_allocations():
    """The plan's launches hold raw pointers into its private memory pool: allocating and freeing around replays
    must not disturb them."""
    from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
    sh = LlamaShape(1024, 2816, 2, 8, 8, 128, 2048)
    ref = _run("graph", torch.float16, sh, 8, 10)
    model = LlamaForCausalLM.random_init(sh, torch.float16, DEV, seed=3)
    r = DecodeRunner(model, RunnerConfig(batch=8, prompt_len=40, n_generate=18, use_graph=True, executor="plan"), seed=4)
    g = torch.Generator().manual_seed(0)
    r.prefill(torch.randint(5, sh.vocab_size - 1, (8, 40), generator=g).to(DEV))
    for i in range(10):
        junk = [torch.full((1 << 20,), float(i), device=DEV) for _ in range(8)]     # churn the caching allocator
        r.step()
        del junk
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    assert torch.equal(r.generated(), ref[1]) and torch.equal(r.pool, ref[2])


def test_recording_is_per_thread_and_exclusive():
    from hydrainfer_amd import _lib, launch_plan
    plan = launch_plan.LaunchPlan(DEV)
    x = torch.ones(64, dtype=torch.int32, device=DEV)

    def body():
        import ctypes
        h = ctypes.c_void_p()
        assert _lib.lib().hx_plan_begin(ctypes.byref(h)) == -5     # HX_ERR_UNSUPPORTED: one recording per thread
        assert launch_plan.current() is plan
        _lib.memset_zero(x)
    plan.capture(body)
    assert int(x.sum()) == 64                                      # recorded, not executed
    plan.replay(); torch.cuda.synchronize()
    assert int(x.sum()) == 0 and plan.n_launches == 1


def test_engine_decoder_plan_equals_graph_with_lookahead():
    """engine/graph_decode.GraphedDecoder with either executor: the same launches (look-ahead input ids taken from the
    previous launch's samples on the device, padded batch, error word) give the same tokens, step after step."""
    from hydrainfer_amd.engine.graph_decode import GraphedDecoder
    from hydrainfer_amd.memory.token_cache_manger import (TokenCacheBlockManager, TokenCacheBlockManagerConfig,
                                                          TokenCacheBlockManagerContext)
    from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.llava import LlavaLanguageModel
    shape = LlamaShape(1024, 2816, 2, 8, 8, 128, 2048)
    outs = {}
    for ex in ("graph", "plan"):
        model = LlamaForCausalLM.random_init(shape, torch.bfloat16, DEV, seed=7)
        kv = TokenCacheBlockManager(TokenCacheBlockManagerConfig(
            n_layers=shape.num_hidden_layers, n_tokens=2, n_blocks=64, block_size=16, n_heads=shape.num_key_value_heads,
            head_size=shape.head_dim, dtype="bf16", device=str(DEV)), TokenCacheBlockManagerContext(rank=0, rank2host={0: "localhost"}))
        dec = GraphedDecoder(LlavaLanguageModel(model, image_token_id=2047), kv, max_batch=8, max_blocks_per_seq=4, executor=ex)
        caches = []
        for _ in range(5):
            vc = kv.allocate_virtual_cache()
            kv.realloc(vc, 40)
            caches.append(vc)
        toks = []
        first = [11, 22, 33, 44, 55]
        pending = None
        for step in range(12):
            rows = []
            for r, vc in enumerate(caches):
                tok = first[r] if step == 0 else -(r + 1)          # step > 0: "the sample of row r of the previous launch"
                rows.append((tok, step, vc.block_table[step // 16] * 16 + step % 16, step + 1, list(vc.block_table)))
            lid = dec.launch(rows)                                  # enqueued before the previous launch's tokens are read
            if pending is not None:
                toks.append(dec.fetch(pending))
            pending = lid
        toks.append(dec.fetch(pending))
        outs[ex] = toks
        del dec, kv, model
    assert outs["graph"] == outs["plan"]
    assert len(outs["plan"]) == 12 and all(len(t) == 5 for t in outs["plan"])
