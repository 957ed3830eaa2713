"""CPU: the oracle (oracle/ops.py) against fixtures produced by the reference itself
(tests/golden/generate_goldens.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import ops
from tests.golden import cases as C
from tests.util import assert_ulp_close, load_golden


def _chk(g, key):
    return str(g[key])


def test_g1_cache_scatter_bit_exact():
    g = load_golden("g1_cache_scatter")
    for i, case in enumerate(C.kv_cache_cases()):
        n = C.case_name("kv", i)
        slot_ids, keys, values, kc, vc = C.kv_cache_inputs(case, seed=i)
        assert C.checksum(slot_ids, keys.contiguous(), values.contiguous(), kc, vc) == _chk(g, n + "_chk")
        ops.set_kv_cache(slot_ids, keys, values, kc, vc)
        assert C.checksum(kc) == _chk(g, n + "_key_cache_chk"), case
        assert C.checksum(vc) == _chk(g, n + "_value_cache_chk"), case
        slot_ids, keys, values, kc, vc = C.kv_cache_inputs(case, seed=i)
        ops.set_image_cache(slot_ids, keys, kc)
        assert C.checksum(kc) == _chk(g, n + "_image_cache_chk"), case


def test_g2_paged_attention():
    g = load_golden("g2_paged_attention")
    for i, case in enumerate(C.paged_attention_cases()):
        n = C.case_name("pattn", i)
        dt = C.DTYPES[case["dtype"]]
        q, k, v, kc, vc, reqs = C.paged_attention_inputs(case, seed=i)
        assert C.checksum(q, k, v, kc, vc) == _chk(g, n + "_chk")
        H, HK, D = case["n_heads"], case["n_kv_heads"], case["head_dim"]
        slots = torch.tensor([s for r in reqs for s in r["new_cache_slots"]], dtype=torch.int32)
        ops.set_kv_cache(slots, k.view(-1, HK, D), v.view(-1, HK, D), kc, vc)
        assert C.checksum(kc, vc) == _chk(g, n + "_cache_chk")
        o = ops.paged_attention(q.view(-1, H, D), kc, vc,
                                torch.from_numpy(g[n + "_q_cu_seq_lens"]),
                                torch.from_numpy(g[n + "_kv_cu_seq_lens"]),
                                torch.from_numpy(g[n + "_block_tables"]),
                                torch.from_numpy(g[n + "_cu_blocks_lens"]))
        ref = C.from_np(g[n + "_o"], dt)
        # same fp32 recipe; summation order inside torch kernels may differ by machine
        assert_ulp_close(o.reshape(ref.shape), ref, max_ulp=1, min_exact_frac=0.99, what=str(case))


def test_g3_dense_attention():
    g = load_golden("g3_dense_attention")
    for i, case in enumerate(C.dense_attention_cases()):
        n = C.case_name("dattn", i)
        dt = C.DTYPES[case["dtype"]]
        q, k, v = C.dense_attention_inputs(case, seed=i)
        assert C.checksum(q, k, v) == _chk(g, n + "_chk")
        B, S, H, D = case["batch"], case["seq_len"], case["n_heads"], case["head_dim"]
        cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32)
        o = ops.varlen_attention(q.view(B * S, H, D), k.view(B * S, H, D), v.view(B * S, H, D),
                                 cu, cu, causal=False)
        ref = C.from_np(g[n + "_o"], dt)
        assert_ulp_close(o.reshape(ref.shape), ref, max_ulp=1, min_exact_frac=0.99, what=str(case))


def test_g4_rms_norm_torch_variant():
    g = load_golden("g4_rms_norm")
    for i, case in enumerate(C.rms_norm_cases()):
        n = C.case_name("rms", i)
        x, w = C.rms_norm_inputs(case, seed=i)
        assert C.checksum(x, w) == _chk(g, n + "_chk")
        ref = C.from_np(g[n + "_o"], C.DTYPES[case["dtype"]])
        assert_ulp_close(ops.rms_norm_torch(x, w, case["eps"]), ref, max_ulp=1,
                         min_exact_frac=0.999, what=str(case))
        # the CUDA-kernel rounding variant differs from the torch path by at most the extra
        # T rounding before the weight multiply: <= 1 ulp of T (2 for fp32 rsqrt vs sqrt/div)
        assert_ulp_close(ops.rms_norm_kernel(x, w, case["eps"]), ref,
                         max_ulp=4 if case["dtype"] == "fp32" else 1, what="kernel-variant " + str(case))


def test_g5_rope_bit_exact():
    g = load_golden("g5_rope")
    for i, case in enumerate(C.rope_cases()):
        n = C.case_name("rope", i)
        dt = C.DTYPES[case["dtype"]]
        q, k, pos = C.rope_inputs(case, seed=i)
        assert C.checksum(q, k, pos) == _chk(g, n + "_chk")
        cs = ops.build_cos_sin_cache(case["rotary_dim"], case["max_pos"], case["theta"], dt)
        assert C.checksum(cs) == _chk(g, n + "_cos_sin_chk"), "cos/sin cache differs from reference"
        qo, ko = ops.apply_rotary_pos_emb(q, k, pos, cs, case["rotary_dim"], case["interleaved"])
        if case["dtype"] == "fp32":
            # reference fp32 path is the same four multiplies and two adds; a*b - c*d vs
            # a*b + (-c)*d is identical in IEEE arithmetic
            assert_ulp_close(qo, C.from_np(g[n + "_q"], dt), max_ulp=0, what=str(case))
            assert_ulp_close(ko, C.from_np(g[n + "_k"], dt), max_ulp=0, what=str(case))
        else:
            assert_ulp_close(qo, C.from_np(g[n + "_q"], dt), max_ulp=0, what=str(case))
            assert_ulp_close(ko, C.from_np(g[n + "_k"], dt), max_ulp=0, what=str(case))


def test_g6_silu():
    g = load_golden("g6_silu")
    for i, case in enumerate(C.silu_cases()):
        n = C.case_name("silu", i)
        x = C.silu_inputs(case, seed=i)
        assert C.checksum(x) == _chk(g, n + "_chk")
        ref = C.from_np(g[n + "_o"], C.DTYPES[case["dtype"]])
        assert_ulp_close(ops.silu(x), ref, max_ulp=0, what=str(case))
        # CUDA-kernel formula: same value up to exp() round-off
        assert_ulp_close(ops.silu_kernel(x), ref, max_ulp=4 if case["dtype"] == "fp32" else 1,
                         what="kernel-variant " + str(case))


def test_g9_migrate_blocks_semantics():
    # restated from csrc/data_transfer/block_migration.cpp:222-244 (CUDA-only; not executable here)
    g = torch.Generator().manual_seed(0)
    src = torch.randn((3, 2, 10, 4, 2, 8), generator=g)
    dst = torch.randn((3, 2, 7, 4, 2, 8), generator=g)
    before = dst.clone()
    ops.migrate_blocks([9, 0, 4], [1, 6, 2], src, dst)
    for s, d in zip([9, 0, 4], [1, 6, 2]):
        assert torch.equal(dst[:, :, d], src[:, :, s])
    for d in (0, 3, 4, 5):
        assert torch.equal(dst[:, :, d], before[:, :, d])


def test_g12_moe_oracle_equals_what_the_reference_tests_accepted():
    """g12: every (inputs, outputs) pair passed the assertions of the reference's own
    tests/kernel/test_moe.py against its torch references when it was generated; oracle/moe.py
    must still produce exactly those outputs."""
    from oracle import moe as O
    g = load_golden("g12_moe")
    n = 0
    for op, topk, ins, outs in C.moe_golden_cases(g):
        if op == "topk_softmax":
            w, i = O.topk_softmax(ins["logits"].float(), topk)
            assert torch.equal(i, outs["indices"]) and torch.equal(w, outs["weights"])
        elif op == "permute_index":
            p, _, m = O.permute_index(ins["tokens"], ins["topk_ids"])
            assert torch.equal(p, outs["permuted"]) and torch.equal(m, outs["row_id_map"])
        elif op == "permute_mask":
            p, _, m = O.permute_mask(ins["tokens"], ins["routing_map"])
            assert torch.equal(p, outs["permuted"]) and torch.equal(m, outs["row_id_map"])
        else:
            o = O.unpermute_rows(ins["permuted"], ins["row_id_map"], ins["probs"])
            assert torch.equal(o, outs["out"]), op
        n += 1
    assert n >= 700
