"""CPU: oracle/moe.py::grouped_topk_sigmoid held to the lane-level emulator of the reference kernel
(tests/moe_lane_emulator.py) on tie-heavy and random inputs, for the (experts, groups) pairs the reference instantiates.
The GPU twin (hx_grouped_topk_sigmoid against the same emulator): tests/test_gpu_moe.py."""
import numpy as np
import pytest
import torch

from tests.moe_lane_emulator import INSTANTIATED, grouped_topk_sigmoid_lanes, tie_heavy_inputs


@pytest.mark.parametrize("n_experts,n_groups", INSTANTIATED)
@pytest.mark.parametrize("topk_group,topk", [(1, 2), (2, 4), (3, 8), (4, 8)])
def test_oracle_equals_the_lane_level_emulation(n_experts, n_groups, topk_group, topk):
    from oracle import moe
    if topk_group > n_groups or topk > topk_group * (n_experts // n_groups):
        pytest.skip("more experts asked for than the kept groups hold")
    for seed, ties in ((1, True), (2, True), (3, False)):
        if ties:
            logits, bias = tie_heavy_inputs(n_experts, 12, seed + n_experts + n_groups)
        else:
            rng = np.random.RandomState(seed)
            logits, bias = rng.randn(12, n_experts).astype(np.float32), (0.1 * rng.randn(n_experts)).astype(np.float32)
        lt, bt = torch.from_numpy(logits), torch.from_numpy(bias)
        scores = (1.0 / (1.0 + torch.exp(-lt))).numpy()                   # the oracle's own sigmoid
        w_emu, i_emu = grouped_topk_sigmoid_lanes(scores, bias, n_groups, topk_group, topk)
        w_ref, i_ref = moe.grouped_topk_sigmoid(lt, bt, n_groups, topk_group, topk)
        assert np.array_equal(i_ref.numpy(), i_emu), (seed, ties)
        assert np.array_equal(w_ref.numpy(), w_emu)


def test_the_tie_rules_are_exercised():
    """All scores equal: the dropped groups are the HIGHEST-numbered ones (higher index wins the min), the experts kept are
    the LOWEST-numbered ones of what is left (lower index wins the max)."""
    n_experts, n_groups = 128, 8
    w, i = grouped_topk_sigmoid_lanes(np.full((1, n_experts), 0.5, np.float32), np.zeros(n_experts, np.float32), n_groups, 3, 6)
    assert i[0].tolist() == [0, 1, 2, 3, 4, 5]
    scores = np.full((1, n_experts), 0.5, np.float32)
    scores[0, :16] = 0.25                   # group 0 is the weakest: dropped first, then groups 7, 6, 5, 4
    w, i = grouped_topk_sigmoid_lanes(scores, np.zeros(n_experts, np.float32), n_groups, 3, 4)
    assert i[0].tolist() == [16, 17, 18, 19]


def test_sum_out_oracle_follows_the_kernels_scalar_t_running_sum():
    """oracle.moe.sum_out against the literal loop of topk_sum_kernel (align_block_kernel.cu:180-187), element by element:
    a scalar_t running sum (rounded after every add) for topk in {2, 3, 4, 8}; for other topk torch::sum_out.  On inputs
    where the two arithmetics differ (large cancelling terms) the oracle must be on the kernel's side."""
    import torch
    from oracle import moe
    g = torch.Generator().manual_seed(7)
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        for topk in (2, 3, 4, 8):
            x = (torch.randn((5, topk, 24), generator=g) * torch.tensor([1e3, 1.0, 1e-2] * 8)).to(dt)
            want = torch.empty((5, 24), dtype=dt)
            for t in range(5):
                for i in range(24):
                    s_ = torch.zeros((), dtype=dt)
                    for k in range(topk):
                        s_ = (s_.float() + x[t, k, i].float()).to(dt)       # `sum += input[...]` in scalar_t
                    want[t, i] = s_
            got = moe.sum_out(x)
            assert torch.equal(got, want), (dt, topk)
            if dt != torch.float32 and topk == 8:      # the two arithmetics really differ on such inputs
                assert not torch.equal(got, x.float().sum(dim=1).to(dt))
        x = torch.randn((4, 5, 16), generator=g).to(dt)
        assert torch.equal(moe.sum_out(x), torch.sum(x, dim=1))
