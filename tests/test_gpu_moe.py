"""GPU: MoE ops against the reference's own test oracle — fixtures that passed the assertions of the
reference's tests/kernel/test_moe.py (tests/golden/g12_moe.npz) and oracle/moe.py, which those fixtures pin — on the
grids of the reference's tests/kernel/test_moe.py:7-158."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n_tokens", [1, 10, 16, 128, 1024])
@pytest.mark.parametrize("n_experts", [4, 8, 16, 32, 64, 128, 256, 60])
@pytest.mark.parametrize("topk", [1, 2, 4])
def test_topk_softmax(n_tokens, n_experts, topk):
    from hydrainfer_amd._C.kernel.moe import topk_softmax
    from oracle import moe
    g = torch.Generator().manual_seed(n_tokens * 1000 + n_experts + topk)
    logits = torch.randn((n_tokens, n_experts), generator=g)
    w_ref, i_ref = moe.topk_softmax(logits, topk)
    w = torch.empty((n_tokens, topk), device=DEV)
    i = torch.empty((n_tokens, topk), dtype=torch.int32, device=DEV)
    topk_softmax(logits.to(DEV), w, i)
    assert torch.equal(i.cpu(), i_ref)                      # indices exact (test_moe.py:31-32)
    assert torch.allclose(w.cpu(), w_ref, rtol=1e-5, atol=1e-7)


def test_topk_softmax_ties_pick_lower_index():
    from hydrainfer_amd._C.kernel.moe import topk_softmax
    logits = torch.zeros((3, 128), device=DEV)
    logits[1, 70] = 1.0
    logits[1, 5] = 1.0
    w = torch.empty((3, 4), device=DEV)
    i = torch.empty((3, 4), dtype=torch.int32, device=DEV)
    topk_softmax(logits, w, i)
    assert i[0].tolist() == [0, 1, 2, 3]
    assert i[1].tolist() == [5, 70, 0, 1]


@pytest.mark.parametrize("cfg", [(128, 4, 2, 8), (128, 8, 4, 8), (256, 8, 4, 8), (256, 16, 4, 6), (64, 8, 3, 4)])
def test_grouped_topk_sigmoid(cfg):
    from hydrainfer_amd._C.kernel.moe import grouped_topk_sigmoid
    from oracle import moe
    n_experts, n_groups, topk_group, topk = cfg
    g = torch.Generator().manual_seed(n_experts + n_groups)
    logits = torch.randn((37, n_experts), generator=g)
    bias = 0.1 * torch.randn((n_experts,), generator=g)
    w_ref, i_ref = moe.grouped_topk_sigmoid(logits, bias, n_groups, topk_group, topk)
    w = torch.empty((37, topk), device=DEV)
    i = torch.empty((37, topk), dtype=torch.int32, device=DEV)
    grouped_topk_sigmoid(logits.to(DEV), bias.to(DEV), n_groups, topk_group, topk, 2.5, w, i)
    assert torch.equal(i.cpu(), i_ref)
    assert torch.allclose(w.cpu(), w_ref, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n_experts,n_groups", [(128, 4), (128, 8), (256, 8), (256, 16)])
@pytest.mark.parametrize("topk_group,topk", [(1, 2), (2, 4), (4, 8)])
def test_grouped_topk_sigmoid_equals_the_lane_level_emulation(n_experts, n_groups, topk_group, topk):
    """hx_grouped_topk_sigmoid against the lane-level emulator of the reference kernel's butterflies and tie rules
    (tests/moe_lane_emulator.py; grouped_topk_sigmoid_kernel.cu:64-180) on tie-heavy inputs, for the four (experts, groups)
    pairs the reference instantiates: indices exact, weights = the raw sigmoid.  (The oracle is held to the same emulator
    on the CPU: tests/test_moe_lane_emulator.py.  Still restatements on both sides: the row stays "parity unpinned".)"""
    import numpy as np
    from hydrainfer_amd._C.kernel.moe import grouped_topk_sigmoid
    from tests.moe_lane_emulator import grouped_topk_sigmoid_lanes, tie_heavy_inputs
    if topk > topk_group * (n_experts // n_groups):
        pytest.skip("more experts asked for than the kept groups hold")
    for seed in (1, 2, 3):
        logits, bias = tie_heavy_inputs(n_experts, 40, seed + n_experts + n_groups)
        lt = torch.from_numpy(logits)
        scores = (1.0 / (1.0 + torch.exp(-lt))).numpy()
        w_emu, i_emu = grouped_topk_sigmoid_lanes(scores, bias, n_groups, topk_group, topk)
        w = torch.empty((40, topk), device=DEV)
        i = torch.empty((40, topk), dtype=torch.int32, device=DEV)
        grouped_topk_sigmoid(lt.to(DEV), torch.from_numpy(bias).to(DEV), n_groups, topk_group, topk, 2.5, w, i)
        assert np.array_equal(i.cpu().numpy(), i_emu), seed
        assert np.allclose(w.cpu().numpy(), w_emu, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("n_tokens", [1, 2, 16, 300])
@pytest.mark.parametrize("dim", [16, 64, 7])
@pytest.mark.parametrize("n_experts", [4, 8, 16])
@pytest.mark.parametrize("topk", [1, 2, 4])
@pytest.mark.parametrize("dtype", [torch.float, torch.half, torch.bfloat16])
def test_permute_index(n_tokens, dim, n_experts, topk, dtype):
    from hydrainfer_amd._C.kernel.moe import permute_with_index_map, unpermute_with_index_map
    from oracle import moe
    g = torch.Generator().manual_seed(n_tokens + dim + n_experts + topk)
    tokens = torch.randn((n_tokens, dim), generator=g).to(dtype)
    gating = torch.randn((n_tokens, n_experts), generator=g).to(dtype)
    weights, indices = gating.float().topk(topk, dim=-1)
    probs = weights.softmax(dim=-1).to(dtype)
    indices = indices.to(torch.int32)
    p_ref, sorted_ref, map_ref = moe.permute_index(tokens, indices)
    p, rmap = permute_with_index_map(tokens.to(DEV), indices.to(DEV))
    assert torch.equal(rmap.cpu(), map_ref)            # stable sort => identical map
    assert torch.equal(p.cpu(), p_ref)                 # pure copy: bit-exact
    out = unpermute_with_index_map(p, rmap, probs.to(DEV))
    out_ref = moe.unpermute_index(p_ref, sorted_ref, probs, n_tokens, topk)
    assert torch.allclose(out.cpu().float(), out_ref.float(), atol=1e-2, rtol=1e-2)
    assert torch.allclose(tokens.float(), out.cpu().float(), atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("n_tokens", [1, 2, 16, 700])
@pytest.mark.parametrize("dim", [16, 64])
@pytest.mark.parametrize("n_experts", [4, 8, 16])
@pytest.mark.parametrize("topk", [1, 2, 4])
@pytest.mark.parametrize("dtype", [torch.float, torch.half, torch.bfloat16])
def test_permute_mask(n_tokens, dim, n_experts, topk, dtype):
    from hydrainfer_amd._C.kernel.moe import permute_with_mask_map, unpermute_with_mask_map
    from oracle import moe
    g = torch.Generator().manual_seed(n_tokens + dim + n_experts + topk)
    tokens = torch.randn((n_tokens, dim), generator=g).to(dtype)
    gating = torch.randn((n_tokens, n_experts), generator=g)
    _, indices = gating.topk(topk, dim=-1)
    probs = torch.zeros_like(gating).scatter(1, indices, 1 / topk).to(dtype)
    routing = torch.zeros_like(gating, dtype=torch.int).scatter(1, indices, 1).to(torch.bool)
    p_ref, sorted_ref, map_ref = moe.permute_mask(tokens, routing)
    p, rmap = permute_with_mask_map(tokens.to(DEV), routing.to(DEV), topk)
    assert torch.equal(rmap.cpu(), map_ref)
    assert torch.equal(p.cpu(), p_ref)
    out = unpermute_with_mask_map(p, rmap, probs.to(DEV))
    permuted_probs = probs.t().masked_select(routing.t())
    out_ref = moe.unpermute_mask(p_ref, permuted_probs, sorted_ref, n_tokens)
    assert torch.allclose(out.cpu().float(), out_ref.float(), atol=1e-2, rtol=1e-2)
    assert torch.allclose(tokens.float(), out.cpu().float(), atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("topk", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("dtype", [torch.float, torch.half, torch.bfloat16])
def test_sum_out(topk, dtype):
    from hydrainfer_amd._C.kernel.moe import sum_out
    from oracle import moe
    for dim, scale in ((36, 1.0), (72, 1.0), (264, 50.0)):      # 36: the scalar form (not a multiple of 8); the others the 16-byte form
        x = (torch.randn((33, topk, dim)) * scale).to(dtype)
        out = torch.empty((33, dim), dtype=dtype, device=DEV)
        sum_out(x.to(DEV), out)
        ref = x.float().sum(dim=1).to(dtype)
        loose = 6e-2 if dtype == torch.bfloat16 else 1e-2      # (a bf16 running sum of 8 terms is up to ~4 ulp of bf16 from the fp32 sum)
        assert torch.allclose(out.cpu().float(), ref.float(), atol=loose * scale, rtol=loose)
        # the reference kernel's own arithmetic (oracle.moe.sum_out: scalar_t running sum for topk in {2,3,4,8}): bit for bit
        want = moe.sum_out(x)
        if topk in (2, 3, 4, 8) or dtype == torch.float:
            if dtype == torch.float and topk not in (2, 3, 4, 8):
                assert torch.allclose(out.cpu(), want, atol=1e-5 * scale, rtol=1e-5)      # torch's fp32 reduction order is its own
            else:
                assert torch.equal(out.cpu(), want), f"{(out.cpu() != want).sum().item()} elements differ"
        else:
            from tests.util import assert_ulp_close
            assert_ulp_close(out.cpu(), want, max_ulp=1, what=f"topk {topk} {dtype}")


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_unpermute_and_sum_out_streaming_forms_equal_the_scalar_forms(dtype):
    """The 16-byte streaming forms of unpermute / sum_out (valid rows compacted once in map order) against the scalar
    forms on the same values — selected by handing over a 2-byte-misaligned copy of the permuted rows — bit for bit, at
    production sizes: 33 tokens x 7168, top 8 of 256 experts, the index map (8 rows) and the mask map (256 rows)."""
    from hydrainfer_amd._C.kernel.moe import (permute_with_index_map, permute_with_mask_map, sum_out,
                                              unpermute_with_index_map, unpermute_with_mask_map)
    n, dim, n_exp, topk = 33, 7168, 256, 8
    g = torch.Generator().manual_seed(5)
    tokens = torch.randn((n, dim), generator=g).to(dtype).to(DEV)
    gating = torch.randn((n, n_exp), generator=g)
    weights, indices = gating.topk(topk, dim=-1)
    probs = weights.softmax(dim=-1).to(dtype).to(DEV)

    def misaligned(t):
        buf = torch.empty(t.numel() + 8, dtype=t.dtype, device=t.device)
        v = buf[1:1 + t.numel()].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 != 0 and v.is_contiguous()
        return v

    p, rmap = permute_with_index_map(tokens, indices.to(torch.int32).to(DEV))
    fast = unpermute_with_index_map(p, rmap, probs)
    slow = unpermute_with_index_map(misaligned(p), rmap, probs)
    assert torch.equal(fast, slow)
    assert torch.allclose(fast.float(), tokens.float(), atol=2e-2, rtol=2e-2)      # the probabilities sum to 1
    routing = torch.zeros((n, n_exp), dtype=torch.bool).scatter(1, indices, True).to(DEV)
    mprobs = torch.zeros((n, n_exp)).scatter(1, indices, weights.softmax(dim=-1)).to(dtype).to(DEV)
    pm, mmap = permute_with_mask_map(tokens, routing, topk)
    fast = unpermute_with_mask_map(pm, mmap, mprobs)
    slow = unpermute_with_mask_map(misaligned(pm), mmap, mprobs)
    assert torch.equal(fast, slow)
    assert torch.allclose(fast.float(), tokens.float(), atol=2e-2, rtol=2e-2)
    x = p.view(n, topk, dim)          # any [n, topk, dim] values
    o_fast, o_slow = torch.empty((n, dim), dtype=dtype, device=DEV), torch.empty((n, dim), dtype=dtype, device=DEV)
    sum_out(x, o_fast)
    sum_out(misaligned(x), o_slow)
    assert torch.equal(o_fast, o_slow)


def test_hip_moe_ops_against_the_reference_test_oracle_fixtures():
    """tests/golden/g12_moe.npz: inputs and outputs that passed the assertions of the reference's
    own tests/kernel/test_moe.py (its torch references, its tolerances) — the HIP kernels are held
    to the same bars on the same data: top-k indices exact and weights allclose (test_moe.py:31-32),
    permutation a bit-exact copy with the identical row map, unpermute within 1e-2 (:98-99, :156-157)."""
    from hydrainfer_amd._C.kernel import moe as K
    from tests.golden import cases as C
    from tests.util import load_golden
    counts = {}
    for op, topk, ins, outs in C.moe_golden_cases(load_golden("g12_moe")):
        d = {k: v.to(DEV) for k, v in ins.items()}
        if op == "topk_softmax":
            w = torch.empty(outs["weights"].shape, device=DEV)
            i = torch.empty(outs["indices"].shape, dtype=torch.int32, device=DEV)
            K.topk_softmax(d["logits"], w, i)
            assert torch.equal(i.cpu(), outs["indices"])
            assert torch.allclose(w.cpu(), outs["weights"])
        elif op == "permute_index":
            p, m = K.permute_with_index_map(d["tokens"], d["topk_ids"])
            assert torch.equal(p.cpu(), outs["permuted"]) and torch.equal(m.cpu(), outs["row_id_map"])
        elif op == "permute_mask":
            p, m = K.permute_with_mask_map(d["tokens"], d["routing_map"], topk)
            assert torch.equal(p.cpu(), outs["permuted"]) and torch.equal(m.cpu(), outs["row_id_map"])
        elif op == "unpermute_index":
            o = K.unpermute_with_index_map(d["permuted"], d["row_id_map"], d["probs"])
            assert torch.allclose(o.cpu().float(), outs["out"].float(), atol=1e-2, rtol=1e-2)
        else:
            o = K.unpermute_with_mask_map(d["permuted"], d["row_id_map"], d["probs"])
            assert torch.allclose(o.cpu().float(), outs["out"].float(), atol=1e-2, rtol=1e-2)
        counts[op] = counts.get(op, 0) + 1
    assert len(counts) == 5 and sum(counts.values()) >= 700, counts
