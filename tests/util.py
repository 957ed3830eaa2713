"""Comparison helpers shared by the CPU and GPU tests."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def _ordered_bits(t: torch.Tensor) -> torch.Tensor:
    """Map float bit patterns to integers that are monotone in the float value."""
    if t.dtype == torch.float32:
        b = t.contiguous().view(torch.int32).to(torch.int64)
        return torch.where(b < 0, -(b & 0x7FFFFFFF), b)
    b = t.contiguous().view(torch.int16).to(torch.int64)
    return torch.where(b < 0, -(b & 0x7FFF), b)


def assert_ulp_close(actual: torch.Tensor, expected: torch.Tensor, max_ulp: int = 1,
                     min_exact_frac: float = 0.0, what: str = ""):
    """Same dtype/shape; every element within max_ulp units in the last place of that dtype,
    and at least min_exact_frac of the elements bit-identical."""
    assert actual.dtype == expected.dtype, (actual.dtype, expected.dtype)
    assert actual.shape == expected.shape, (actual.shape, expected.shape)
    a, e = actual.detach().cpu(), expected.detach().cpu()
    assert not torch.isnan(a.float()).any(), f"{what}: NaN in result"
    d = (_ordered_bits(a) - _ordered_bits(e)).abs()
    worst = int(d.max()) if d.numel() else 0
    assert worst <= max_ulp, f"{what}: max ulp distance {worst} > {max_ulp}"
    if d.numel() >= 256:   # a fraction is meaningless for a handful of elements
        frac = float((d == 0).float().mean())
        assert frac >= min_exact_frac, f"{what}: only {frac:.4f} bit-exact (< {min_exact_frac})"


def assert_close_t(actual: torch.Tensor, expected: torch.Tensor, atol: float, rtol: float,
                   what: str = ""):
    a, e = actual.detach().cpu().float(), expected.detach().cpu().float()
    assert a.shape == e.shape, (a.shape, e.shape)
    assert not torch.isnan(a).any(), f"{what}: NaN in result"
    err = (a - e).abs()
    bound = atol + rtol * e.abs()
    bad = err > bound
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; "
                           f"max abs err {float(err.max()):.3e}")


# tolerances stated per dtype (reference bars: attention 1e-3 fp16 tests/kernel/
# test_attention_kernel.py:176, layer 1e-2 tests/layer/test_attention.py:102-106;
# bf16 is the extension tier, SURVEY.md §7 "Hard parts")
ATTN_TOL = {torch.float16: (1e-3, 1e-3), torch.bfloat16: (1e-2, 1e-2)}
