"""-m gpu: the in-kernel hand-over of the norm-fused GEMM launches (csrc/gemm_xreg.hip: write-through stores, drained,
relaxed agent-scope counter, per-XCD flag lines, sc1 loads) exercised the way the engine runs it — BESIDE A SECOND
STREAM (the reference runs the vision encoder on its own stream next to the language model,
hydrainfer/engine/executor.py:247-249; here engine/executor.py).  Round-3 VERDICT: the protocol rests on cache-policy
bits, not on fences the compiler knows, and no test ran it with other work occupying CUs and saturating HBM."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _runner(fuse_norm, executor, steps, seed=3, batch=32, prompt=40):
    from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
    shape = LlamaShape(4096, 11008, 2, 32, 32, 128, 32064)
    model = LlamaForCausalLM.random_init(shape, torch.bfloat16, DEV, seed=seed)
    model.fuse_norm = fuse_norm
    r = DecodeRunner(model, RunnerConfig(batch=batch, prompt_len=prompt, n_generate=steps + 4, use_graph=True,
                                         executor=executor), seed=seed + 1)
    g = torch.Generator().manual_seed(0)
    r.prefill(torch.randint(5, 32000, (batch, prompt), generator=g).to(DEV))
    return model, r


@pytest.mark.parametrize("executor", ["plan", "graph"])
def test_norm_fused_layer_beside_a_second_stream_is_bit_identical(executor):
    """200 replays of a 7B-width decode step with the add+RMSNorm inside the gate|up / qkv launches, each step's inputs
    being the previous step's samples, while a second stream keeps the GPU busy with an HBM-saturating 1 GiB copy and a
    4 x 704-token prefill attention launch per step (MFMA workgroups holding CUs the producers / waiters want).  Tokens
    and KV pool must equal, bit for bit, the run with the norms as separate launches and nothing beside it; no
    hand-over may give up."""
    from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd
    steps = 200
    _, ref = _runner(False, executor, steps)
    for _ in range(steps):
        ref.step()
    torch.cuda.synchronize()
    want_tokens, want_pool = ref.generated(), ref.pool.clone()
    del ref

    model, r = _runner(True, executor, steps)
    assert model.fuse_norm
    # the neighbour's work: a 1 GiB device copy (~0.4 ms: longer than the 2-layer step) + one prefill attention launch
    src = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
    dst = torch.empty_like(src)
    H, D, n_seq, S = 32, 128, 4, 704
    g = torch.Generator(device=DEV).manual_seed(1)
    q = torch.randn((n_seq * S, H, D), generator=g, device=DEV, dtype=torch.bfloat16)
    kc = torch.randn((n_seq * S // 16, 16, H, D), generator=g, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn_like(kc)
    o = torch.empty_like(q)
    cu = torch.arange(0, (n_seq + 1) * S, S, dtype=torch.int32, device=DEV)
    tables = torch.arange(n_seq * S // 16, dtype=torch.int32, device=DEV)
    cu_blocks = torch.arange(0, (n_seq + 1) * (S // 16), S // 16, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream(device=DEV)
    r.step(); torch.cuda.synchronize()             # capture / record outside the contended phase
    side.wait_stream(torch.cuda.current_stream(DEV))
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    with torch.cuda.stream(side):
        s0.record()
    for _ in range(steps - 1):
        with torch.cuda.stream(side):
            dst.copy_(src, non_blocking=True)
            mha_varlen_fwd(o, q, kc, vc, cu, cu, tables, cu_blocks, None, S, S, D ** -0.5, 0, -1, 0, 0)
        r.step()
    t1.record()
    with torch.cuda.stream(side):
        s1.record()
    torch.cuda.synchronize()
    # the neighbour really was there for the whole run: its work alone is >= 0.35 ms per step, and the decode steps
    # (0.35 ms each when alone) were stretched by sharing the GPU with it
    main_ms, side_ms = t0.elapsed_time(t1), s0.elapsed_time(s1)
    print(f"[stress/{executor}] {steps - 1} steps beside the second stream: decode stream {main_ms:.1f} ms, neighbour {side_ms:.1f} ms")
    assert side_ms > 0.25 * (steps - 1), (main_ms, side_ms)
    assert model.xreg_sync is not None and not model.handover_failed()
    got = r.generated()
    assert torch.equal(got, want_tokens), "tokens differ from the separate-launch run"
    assert torch.equal(r.pool, want_pool), "KV pool differs from the separate-launch run"
