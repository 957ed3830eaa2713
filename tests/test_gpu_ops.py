"""GPU parity tests (through the C ABI via the hydrainfer._C-shaped shims): cache scatter,
rms_norm, rope, silu against the reference-generated fixtures and the oracle."""
import pytest
import torch

from tests.golden import cases as C
from tests.util import assert_close_t, assert_ulp_close, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from hydrainfer_amd._C.kernel import (activation, cache_kernels, kv_cache_kernels, norm,
                                          position_embedding)
    return activation, cache_kernels, kv_cache_kernels, norm, position_embedding


def test_native_library_is_loaded():
    from hydrainfer_amd import _lib
    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert "libhydra_hip.so" in maps


def test_set_kv_cache_and_image_cache_bit_exact():
    _, cache_kernels, kv_cache_kernels, _, _ = _ops()
    g = load_golden("g1_cache_scatter")
    for i, case in enumerate(C.kv_cache_cases()):
        n = C.case_name("kv", i)
        slot_ids, keys, values, kc, vc = C.kv_cache_inputs(case, seed=i)
        qkv_dev = keys._base.to(DEV) if keys._base is not None else None
        # rebuild the strided views on the device from the same fused buffer
        hd = case["n_heads"] * case["head_dim"]
        kd = qkv_dev[:, hd:2 * hd].view(-1, case["n_heads"], case["head_dim"])
        vd = qkv_dev[:, 2 * hd:].view(-1, case["n_heads"], case["head_dim"])
        kcd, vcd = kc.to(DEV), vc.to(DEV)
        kv_cache_kernels.set_kv_cache(slot_ids.to(DEV), kd, vd, kcd, vcd)
        assert C.checksum(kcd.cpu()) == str(g[n + "_key_cache_chk"]), case
        assert C.checksum(vcd.cpu()) == str(g[n + "_value_cache_chk"]), case
        icd = kc.to(DEV)
        cache_kernels.set_image_cache(slot_ids.to(DEV), kd, icd)
        assert C.checksum(icd.cpu()) == str(g[n + "_image_cache_chk"]), case


def test_set_kv_cache_edges():
    _, _, kv_cache_kernels, _, _ = _ops()
    kc = torch.zeros((4, 16, 2, 64), dtype=torch.float16, device=DEV)
    vc = torch.zeros_like(kc)
    # empty input
    kv_cache_kernels.set_kv_cache(torch.empty(0, dtype=torch.int32, device=DEV),
                                  torch.empty((0, 2, 64), dtype=torch.float16, device=DEV),
                                  torch.empty((0, 2, 64), dtype=torch.float16, device=DEV), kc, vc)
    assert float(kc.abs().sum()) == 0
    # last slot of the last block, and a non-contiguous per-layer view of a 6-D pool
    pool = torch.zeros((2, 2, 4, 16, 2, 64), dtype=torch.bfloat16, device=DEV)
    k = torch.randn((3, 2, 64), device=DEV).to(torch.bfloat16)
    v = torch.randn((3, 2, 64), device=DEV).to(torch.bfloat16)
    slots = torch.tensor([63, 0, 17], dtype=torch.int32, device=DEV)
    kv_cache_kernels.set_kv_cache(slots, k, v, pool[1, 0], pool[1, 1])
    assert torch.equal(pool[1, 0].view(-1, 2, 64)[slots.long()], k)
    assert torch.equal(pool[1, 1].view(-1, 2, 64)[slots.long()], v)
    assert float(pool[0].float().abs().sum()) == 0
    # wrong dtype for slots is an error, not a silent reinterpretation
    from hydrainfer_amd._lib import HydraHipError
    with pytest.raises(HydraHipError):
        kv_cache_kernels.set_kv_cache(slots.long(), k, v, pool[1, 0], pool[1, 1])


def test_rms_norm():
    from oracle import ops
    _, _, _, norm, _ = _ops()
    g = load_golden("g4_rms_norm")
    for i, case in enumerate(C.rms_norm_cases()):
        dt = C.DTYPES[case["dtype"]]
        x, w = C.rms_norm_inputs(case, seed=i)
        out = torch.empty_like(x, device=DEV)
        norm.rms_norm(out, x.to(DEV), w.to(DEV), case["eps"])
        # rounding points of the CUDA kernel (rms_norm.cu:39): bit-equal to the kernel-variant
        # oracle up to the fp32 reduction order (<= 1 ulp of T, 4 ulp in fp32)
        assert_ulp_close(out.cpu(), ops.rms_norm_kernel(x, w, case["eps"]),
                         max_ulp=4 if dt == torch.float32 else 1,
                         min_exact_frac=0.0 if dt == torch.float32 else 0.98, what=str(case))
        # reference's own bar vs its torch path: 1e-3 (tests/kernel/test_rms_norm_kernel.py:24);
        # bf16 (extension tier) 1e-2
        ref = C.from_np(g[C.case_name("rms", i) + "_o"], dt)
        tol = 1e-2 if dt == torch.bfloat16 else 1e-3
        assert_close_t(out, ref, tol, tol, what=str(case))


def test_add_rms_norm_equals_unfused():
    _, _, _, norm, _ = _ops()
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        for hidden in (4096, 5120, 334):
            x = torch.randn((9, hidden), device=DEV).to(dt)
            r = torch.randn((9, hidden), device=DEV).to(dt)
            w = (1 + 0.1 * torch.randn(hidden, device=DEV)).to(dt)
            h = r + x
            want = torch.empty_like(h)
            norm.rms_norm(want, h, w, 1e-5)
            got, res = torch.empty_like(x), r.clone()
            norm.add_rms_norm(got, res, x, w, 1e-5)
            assert torch.equal(res, h)
            assert torch.equal(got, want)


def test_rope_bit_exact():
    from oracle import ops
    _, _, _, _, pe = _ops()
    g = load_golden("g5_rope")
    for i, case in enumerate(C.rope_cases()):
        dt = C.DTYPES[case["dtype"]]
        n = C.case_name("rope", i)
        q, k, pos = C.rope_inputs(case, seed=i)
        cs = ops.build_cos_sin_cache(case["rotary_dim"], case["max_pos"], case["theta"], dt)
        qd, kd = q.to(DEV), k.to(DEV)
        pe.apply_rotary_pos_emb(qd, kd, pos.to(DEV), cs.to(DEV), case["rotary_dim"], case["interleaved"])
        # T arithmetic with the same rounding points as the reference: bit-exact
        assert_ulp_close(qd.cpu(), C.from_np(g[n + "_q"], dt), max_ulp=0, what=str(case))
        assert_ulp_close(kd.cpu(), C.from_np(g[n + "_k"], dt), max_ulp=0, what=str(case))


def test_rope_strided_qkv_views():
    from oracle import ops
    _, _, _, _, pe = _ops()
    H, HK, D, n = 32, 32, 128, 17
    qkv = torch.randn((n, (H + 2 * HK) * D)).to(torch.float16)
    pos = torch.arange(100, 100 + n, dtype=torch.int32)
    cs = ops.build_cos_sin_cache(D, 4096, 1e4, torch.float16)
    q_ref, k_ref = ops.apply_rotary_pos_emb(qkv[:, :H * D].view(n, H, D),
                                            qkv[:, H * D:(H + HK) * D].view(n, HK, D), pos, cs, D, False)
    d = qkv.to(DEV)
    pe.apply_rotary_pos_emb(d[:, :H * D].view(n, H, D), d[:, H * D:(H + HK) * D].view(n, HK, D),
                            pos.to(DEV), cs.to(DEV), D, False)
    assert torch.equal(d[:, :H * D].view(n, H, D).cpu(), q_ref)
    assert torch.equal(d[:, H * D:(H + HK) * D].view(n, HK, D).cpu(), k_ref)
    assert torch.equal(d[:, (H + HK) * D:].cpu(), qkv[:, (H + HK) * D:])  # v untouched


def test_silu():
    from oracle import ops
    act, *_ = _ops()
    g = load_golden("g6_silu")
    for i, case in enumerate(C.silu_cases()):
        dt = C.DTYPES[case["dtype"]]
        x = C.silu_inputs(case, seed=i)
        out = act.silu(x.to(DEV))
        assert out.is_contiguous() and out.shape == x.shape
        ref = C.from_np(g[C.case_name("silu", i) + "_o"], dt)
        # fp32: the kernel uses the fast exp of the CUDA original (__expf, activation.cu:18);
        # 16-bit types: at most the final rounding differs
        assert_ulp_close(out.cpu(), ref, max_ulp=16 if dt == torch.float32 else 1,
                         min_exact_frac=0.0 if dt == torch.float32 else 0.97, what=str(case))
        assert_close_t(out, ref, 1e-3, 1e-3 if dt != torch.bfloat16 else 8e-3, what=str(case))
    # row-strided input (gate half of a fused gate|up projection), activation.cu:36-37
    x = (3 * torch.randn((6, 2 * 1024))).to(torch.float16)
    out = act.silu(x.to(DEV)[:, :1024])
    assert_ulp_close(out.cpu(), ops.silu(x[:, :1024].contiguous()), max_ulp=1, what="strided")


def test_silu_and_mul_equals_unfused():
    act, *_ = _ops()
    for dt in (torch.float16, torch.bfloat16):
        gu = (2 * torch.randn((7, 2 * 11008), device=DEV)).to(dt)
        g_, u_ = gu[:, :11008], gu[:, 11008:]
        want = act.silu(g_) * u_
        got = act.silu_and_mul(g_, u_)
        assert torch.equal(got, want)


def test_quick_gelu_equals_the_three_torch_ops():
    """hx_quick_gelu against the reference's `x * torch.sigmoid(1.702 * x)` (activation.py:17-22) run op by op: the same
    three roundings — bit-identical for fp16 / bf16."""
    act, *_ = _ops()
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        for shape in ((577, 4096), (3, 577, 256), (5, 64)):
            x = (3 * torch.randn(shape, device=DEV)).to(dt)
            want = x * torch.sigmoid(1.702 * x)
            got = act.quick_gelu(x)
            assert got.shape == x.shape and got.dtype == dt
            if dt == torch.float32:
                assert_ulp_close(got.cpu(), want.cpu(), max_ulp=8, what=f"{dt} {shape}")
            else:                         # the same three roundings, each product through fp32 first: bit for bit
                assert torch.equal(got, want), f"{dt} {shape}: {(got != want).sum().item()} elements differ"
    x = (3 * torch.randn((9, 2 * 512), device=DEV)).to(torch.bfloat16)            # row-strided input
    assert torch.equal(act.quick_gelu(x[:, :512]), x[:, :512] * torch.sigmoid(1.702 * x[:, :512]))
    with pytest.raises(RuntimeError):
        act.quick_gelu(torch.zeros((4, 12), dtype=torch.bfloat16, device=DEV))    # n % 8


def test_add_layer_norm_equals_add_then_layer_norm():
    """hx_add_layer_norm against `h = h + y; F.layer_norm(h)`: the residual bit-exact, the normalised rows to the last
    place of T (fp32 moments on both sides, different summation orders) and against an fp64 reference of the same h."""
    import torch.nn.functional as F
    from hydrainfer_amd._C.kernel import norm
    for dt, tol in ((torch.float16, 2e-3), (torch.bfloat16, 1.6e-2), (torch.float32, 2e-6)):
        for rows, hidden in ((577, 1024), (33, 128), (5, 4096), (2, 8192 if dt != torch.float32 else 4096)):
            h = torch.randn((rows, hidden), device=DEV).to(dt)
            y = (0.5 * torch.randn((rows, hidden), device=DEV)).to(dt)
            w = (1 + 0.1 * torch.randn(hidden, device=DEV)).to(dt)
            b = (0.1 * torch.randn(hidden, device=DEV)).to(dt)
            h_ref = h + y
            want = F.layer_norm(h_ref, (hidden,), w, b, 1e-5)
            res, out = h.clone(), torch.empty_like(h)
            norm.add_layer_norm(out, res, y, w, b, 1e-5)
            assert torch.equal(res, h_ref)
            exact = F.layer_norm(h_ref.double(), (hidden,), w.double(), b.double(), 1e-5)
            assert (out.double() - exact).abs().max().item() <= tol * 4, (dt, rows, hidden)
            if dt != torch.float32:      # (fp32: an output next to zero is a cancellation — ulps mean nothing there)
                assert_ulp_close(out.cpu(), want.cpu(), max_ulp=2, min_exact_frac=0.98, what=f"{dt} {rows}x{hidden}")
            else:
                assert (out - want).abs().max().item() <= 4e-6
            res2, out2 = h_ref.clone(), torch.empty_like(h)
            norm.add_layer_norm(out2, res2, None, w, b, 1e-5)                 # plain layer norm: residual untouched
            assert torch.equal(res2, h_ref) and torch.equal(out2, out)
    with pytest.raises(RuntimeError):
        z = torch.zeros((2, 20), dtype=torch.bfloat16, device=DEV)
        norm.add_layer_norm(z.clone(), z, None, z[0], z[0], 1e-5)             # hidden % 8


def test_rope_set_kv_cache_equals_two_ops():
    from oracle import ops
    from hydrainfer_amd._C.kernel import kv_cache_kernels, position_embedding as pe
    for dt in (torch.float16, torch.bfloat16):
        H, HK, D, n, bs = 32, 32, 128, 37, 16
        qkv = torch.randn((n, (H + 2 * HK) * D)).to(dt).to(DEV)
        pos = torch.randint(0, 4096, (n,), dtype=torch.int32).to(DEV)
        cs = ops.build_cos_sin_cache(D, 4096, 1e4, dt).to(DEV)
        slots = torch.randperm(20 * bs)[:n].to(torch.int32).to(DEV)
        kc = torch.randn((20, bs, HK, D)).to(dt).to(DEV)
        vc = torch.randn((20, bs, HK, D)).to(dt).to(DEV)
        a, b = qkv.clone(), qkv.clone()
        kc2, vc2 = kc.clone(), vc.clone()
        view = lambda t: (t[:, :H * D].view(n, H, D), t[:, H * D:(H + HK) * D].view(n, HK, D),
                          t[:, (H + HK) * D:].view(n, HK, D))
        qa, ka, va = view(a)
        pe.apply_rotary_pos_emb(qa, ka, pos, cs, D, False)
        kv_cache_kernels.set_kv_cache(slots, ka, va, kc, vc)
        qb, kb, vb = view(b)
        pe.rope_set_kv_cache(qb, kb, vb, pos, cs, D, slots, kc2, vc2)
        assert torch.equal(a, b) and torch.equal(kc, kc2) and torch.equal(vc, vc2)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 7, 32, 33, 64])
def test_linear_decode_matches_fp32_reference(dt, M):
    """Weight-streaming decode GEMM vs an fp32 matmul of the same T inputs (fp32 accumulation in
    both; only the summation order differs): within 1 ulp of T almost everywhere."""
    from hydrainfer_amd._C.kernel.gemm import linear_decode
    # 7B: qkv / o / gate|up / down; 13B (configs[2]): qkv 15360x5120, o 5120x5120, gate|up 27648x5120,
    # down 5120x13824
    for (N, K) in ((4096, 4096), (12288, 4096), (4096, 11008), (22016, 4096), (48, 256), (5120 * 3, 5120),
                   (5120, 5120), (27648, 5120), (5120, 13824)):
        if N * K > 100e6 and M not in (1, 32, 64):
            continue
        g = torch.Generator().manual_seed(N + K + M)
        x = torch.randn((M, K), generator=g).to(dt)
        w = (torch.randn((N, K), generator=g) * 0.02).to(dt)
        got = linear_decode(x.to(DEV), w.to(DEV)).cpu()
        ref = (x.double() @ w.double().t())
        err = (got.double() - ref).abs()
        tol = (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) * ref.abs().clamp_min(0.05)
        assert (err <= tol).all(), f"N={N} K={K} M={M} {dt}: max err {err.max()}"


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_slab_consumers_equal_reduce_then_op(dt):
    """add_rms_norm_slabs / silu_and_mul_slabs == (linear_decode -> T) followed by the plain op."""
    from hydrainfer_amd._C.kernel import activation, gemm, norm
    M, hid, inter = 32, 4096, 11008
    g = torch.Generator().manual_seed(3)
    x = torch.randn((M, hid), generator=g).to(dt).to(DEV)
    res = torch.randn((M, hid), generator=g).to(dt).to(DEV)
    w_o = (torch.randn((hid, hid), generator=g) * 0.02).to(dt).to(DEV)
    w_gu = (torch.randn((2 * inter, hid), generator=g) * 0.02).to(dt).to(DEV)
    w_n = (1 + 0.1 * torch.randn(hid, generator=g)).to(dt).to(DEV)
    ws = torch.empty(gemm.workspace_floats(M, 2 * inter, hid), dtype=torch.float32, device=DEV)
    # o-projection -> residual add + norm
    a = gemm.linear_decode(x, w_o)
    want_res = res.clone(); want = torch.empty_like(a)
    norm.add_rms_norm(want, want_res, a, w_n, 1e-5)
    s = gemm.linear_decode_partial(x, w_o, ws)
    got_res = res.clone(); got = torch.empty_like(a)
    norm.add_rms_norm_slabs(got, got_res, ws, s, w_n, 1e-5)
    assert torch.equal(got_res, want_res)          # residual stream: bit-identical
    # the row statistic is reduced by 512 threads instead of 256 (different fp32 summation
    # order): the normalised output may differ in the last place of T on a few elements
    assert_ulp_close(got.cpu(), want.cpu(), max_ulp=1, min_exact_frac=0.99, what="add_rms_norm_slabs")
    # gate|up projection -> silu * mul
    gu = gemm.linear_decode(x, w_gu)
    want = activation.silu_and_mul(gu[:, :inter], gu[:, inter:])
    s = gemm.linear_decode_partial(x, w_gu, ws)
    got = activation.silu_and_mul_slabs(ws, s, M, inter, dt)
    assert torch.equal(got, want)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 32, 64])
def test_packed_weight_gemm_is_bit_identical(dt, M):
    """gemm_packed_kernel (fragment-order weights, the decode path's GEMM) == gemm_skinny_kernel
    (row-major weights): same k order, same accumulation chains -> identical fp32 slabs, on the 7B
    and 13B projection shapes and on a K with a short last split."""
    from hydrainfer_amd._C.kernel import gemm
    for (N, K) in ((4096, 4096), (12288, 4096), (4096, 11008), (22016, 4096), (48, 256), (15360, 5120),
                   (5120, 5120), (27648, 5120), (5120, 13824), (64, 2816)):
        g = torch.Generator().manual_seed(N + K + M)
        x = torch.randn((M, K), generator=g).to(dt).to(DEV)
        w = (torch.randn((N, K), generator=g) * 0.02).to(dt).to(DEV)
        a = torch.zeros(gemm.workspace_floats(M, N, K), dtype=torch.float32, device=DEV)
        b = torch.zeros_like(a)
        sa = gemm.linear_decode_partial(x, w, a)
        sb = gemm.linear_decode_partial_packed(x, gemm.pack_weight(w), N, b)
        assert sa == sb and torch.equal(a, b), f"N={N} K={K} M={M} {dt}"


# ---- activations-in-registers GEMM (csrc/gemm_xreg.hip) -----------------------------------------
XREG_SHAPES = ((4096, 4096), (22016, 4096), (4096, 11008), (27648, 5120), (5120, 13824), (48, 64),
               (64, 2816), (1024, 96), (32, 4128))   # 7B, 13B, tiny, ragged last split / padded waves


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 7, 16, 17, 32])
def test_xreg_gemm_matches_fp32_product_and_is_repeatable(dt, M):
    """gemm_xreg_kernel against the fp32 product of the same T inputs (the GEMM accumulates in fp32:
    only the summation order differs — rel. 1e-5 of the largest output), row-major x and
    fragment-major x bit-identical to each other, two runs bit-identical (deterministic order)."""
    from hydrainfer_amd._C.kernel import gemm
    for (N, K) in XREG_SHAPES:
        assert gemm.xreg_supported(M, N, K, dt)
        g = torch.Generator().manual_seed(N + K + M)
        x = torch.randn((M, K), generator=g).to(dt).to(DEV)
        w = (torch.randn((N, K), generator=g) * 0.02).to(dt).to(DEV)
        pk = gemm.pack_weight_xreg(w)
        a = torch.zeros(gemm.xreg_workspace_floats(M, N, K), dtype=torch.float32, device=DEV)
        b, c = torch.zeros_like(a), torch.zeros_like(a)
        s = gemm.linear_decode_partial_xreg(x, pk, N, a)
        assert s == a.numel() // (M * N) and 1 <= s <= (K + 4095) // 4096 + 1
        assert gemm.linear_decode_partial_xreg(x, pk, N, b) == s
        assert gemm.linear_decode_partial_xreg(gemm.to_fragment_major(x), pk, N, c, frag_shape=(M, K)) == s
        assert torch.equal(a, b), f"not repeatable N={N} K={K}"
        assert torch.equal(a, c), f"fragment-major x differs N={N} K={K}"
        ref = x.float() @ w.float().t()
        got = a.view(s, M, N).sum(0)
        assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-6, f"N={N} K={K} M={M} {dt}"
    assert not gemm.xreg_supported(65, 4096, 4096, dt)         # 33 .. 64 rows: the wide form (tests below)
    assert not gemm.xreg_supported(8, 4096, 4100, dt)
    assert not gemm.xreg_supported(8, 4090, 4096, dt)


def test_xreg_gemm_share_past_the_lds_tile_budget_grows_the_grid():
    """A workgroup keeps one LDS tile set per row group of its share and the share is derived from gridDim.x inside
    the kernel: when ceil(row groups / workgroups) exceeds the 16 tile sets the launcher sizes LDS for, the grid must
    grow (round-3 ADVICE: it did not, and the kernel indexed LDS past its allocation).  Reached with a very wide N, a
    wide gate|up with two K splits' worth of CUs, and the xreg_wgs tuning option."""
    from hydrainfer_amd import _lib
    from hydrainfer_amd._C.kernel import gemm
    lib, dt, M = _lib.lib(), torch.bfloat16, 32

    def check(N, K, gate_up=False):
        g = torch.Generator().manual_seed(N + K)
        x = torch.randn((M, K), generator=g).to(dt).to(DEV)
        w = (torch.randn((N, K), generator=g) * 0.02).to(dt).to(DEV)
        ref = x.float() @ w.float().t()
        a = torch.zeros(gemm.xreg_workspace_floats(M, N, K), dtype=torch.float32, device=DEV)
        s = gemm.linear_decode_partial_xreg(x, gemm.pack_weight_xreg(w), N, a)
        got = a.view(s, M, N).sum(0)
        assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-6, f"N={N} K={K}"
        if gate_up:
            inter = N // 2
            act = torch.zeros(gemm.fragment_major_elems(M, inter), dtype=dt, device=DEV)
            gemm.gate_up_silu_xreg(gemm.to_fragment_major(x), gemm.pack_weight_xreg(w, interleave_halves=True), inter, act,
                                   frag_shape=(M, K))
            want = torch.nn.functional.silu(ref[:, :inter]) * ref[:, inter:]
            got = gemm.from_fragment_major(act, M, inter).float()
            assert ((got - want).abs() <= 1.6e-2 * want.abs() + 1e-3).all(), f"gate|up N={N} K={K}"

    check(131072, 64)             # 8192 row groups over 256 workgroups: 32 per workgroup without the fix
    check(66560, 64, gate_up=True)   # 2080 gate/up pairs over 256 workgroups: 9 pairs = 18 tile sets without the fix
    assert lib.hx_debug_set_option(b"xreg_wgs", 16) == 0
    try:
        check(12288, 4096)        # 768 row groups over 16 workgroups
        check(22016, 4096, gate_up=True)
    finally:
        lib.hx_debug_set_option(b"xreg_wgs", 0)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 5, 16, 31, 32])
def test_fragment_major_producers_are_bit_identical(dt, M):
    """hx_add_rms_norm_slabs_ex / hx_silu_and_mul_slabs_ex with fragment-major output == the row-major
    forms (which are oracle-checked above), element for element."""
    from hydrainfer_amd._C.kernel import gemm
    from hydrainfer_amd._C.kernel.activation import silu_and_mul_slabs
    from hydrainfer_amd._C.kernel.norm import add_rms_norm_slabs
    for hid, inter, S in ((4096, 11008, 3), (5120, 13824, 4), (64, 96, 1), (1024, 2816, 2)):
        g = torch.Generator().manual_seed(hid + M)
        slabs = torch.randn((S, M, max(hid, 2 * inter)), generator=g).to(DEV)
        h = torch.randn((M, hid), generator=g).to(dt).to(DEV)
        h2, wn = h.clone(), torch.randn(hid, generator=g).to(dt).to(DEV)
        ps = slabs[:, :, :hid].contiguous()
        o1 = torch.empty_like(h)
        o2 = torch.zeros(gemm.fragment_major_elems(M, hid), dtype=dt, device=DEV)
        add_rms_norm_slabs(o1, h, ps, S, wn, 1e-5)
        add_rms_norm_slabs(o2, h2, ps, S, wn, 1e-5, fragment_major=True)
        assert torch.equal(h, h2) and torch.equal(gemm.from_fragment_major(o2, M, hid), o1)
        pg = slabs[:, :, :2 * inter].contiguous()
        a1 = silu_and_mul_slabs(pg, S, M, inter, dt)
        a2 = silu_and_mul_slabs(pg, S, M, inter, dt, fragment_major=True)
        assert torch.equal(gemm.from_fragment_major(a2, M, inter), a1)
    x = torch.randn((M, 128)).to(dt).to(DEV)
    assert torch.equal(gemm.from_fragment_major(gemm.to_fragment_major(x), M, 128), x)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 9, 16, 32])
def test_fused_gate_up_silu_equals_gemm_then_silu(dt, M):
    """hx_gate_up_silu_xreg (gate|up GEMM + silu*mul, one launch, no K split) == the xreg GEMM followed
    by hx_silu_and_mul_slabs: bit-identical with the k rotation off (the rotation start depends on the
    workgroup a row group lands on, which differs between the two launches), within a few ulp of T with
    it on; and within the silu tolerance of the fp32 formula either way."""
    from hydrainfer_amd import _lib
    from hydrainfer_amd._C.kernel import gemm
    from hydrainfer_amd._C.kernel.activation import silu_and_mul_slabs
    lib = _lib.lib()
    try:
        for stagger in (0, 1):
            assert lib.hx_debug_set_option(b"xreg_stagger", stagger) == 0
            for (inter, K) in ((11008, 4096), (4096, 2048), (96, 64), (13824, 5120)):   # shapes whose plain plan is one split too
                assert gemm.gate_up_silu_supported(M, inter, K, dt)
                g = torch.Generator().manual_seed(inter + M)
                x = torch.randn((M, K), generator=g).to(dt).to(DEV)
                w = (torch.randn((2 * inter, K), generator=g) * 0.03).to(dt).to(DEV)
                ws = torch.zeros(gemm.xreg_workspace_floats(M, 2 * inter, K), dtype=torch.float32, device=DEV)
                s = gemm.linear_decode_partial_xreg(x, gemm.pack_weight_xreg(w), 2 * inter, ws)
                assert s == 1
                want = silu_and_mul_slabs(ws, s, M, inter, dt)
                act = torch.zeros(gemm.fragment_major_elems(M, inter), dtype=dt, device=DEV)
                gemm.gate_up_silu_xreg(gemm.to_fragment_major(x), gemm.pack_weight_xreg(w, interleave_halves=True),
                                       inter, act, frag_shape=(M, K))
                got = gemm.from_fragment_major(act, M, inter)
                if stagger == 0:
                    assert torch.equal(got, want), f"inter={inter} K={K} M={M} {dt}"
                else:
                    ulp = 2.0 ** (-10 if dt == torch.float16 else -7)
                    # gate and up each move by <= 1 ulp of T (fp32 summation order), silu'(g) <= 1.1
                    assert ((got.float() - want.float()).abs() <= 4 * ulp * want.float().abs() + 0.25 * ulp).all()
                gu = x.float() @ w.float().t()
                ref = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
                tol = 2e-3 if dt == torch.float16 else 1.6e-2
                assert ((got.float() - ref).abs() <= tol * ref.abs() + tol * 0.05).all()
        assert gemm.gate_up_silu_supported(8, 13824, 5120, dt)       # 13B: 40 k-steps per wave, still one split
        assert not gemm.gate_up_silu_supported(8, 13824, 8192, dt)   # K = 8192 needs two splits: unfused path
    finally:
        lib.hx_debug_set_option(b"xreg_stagger", 1)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_norm_fused_in_front_of_xreg_product_is_bit_identical(dt):
    """hx_norm_linear_decode_xreg / hx_norm_gate_up_silu_xreg (add + RMSNorm computed by the first M
    workgroups of the GEMM grid, in-kernel hand-over) == hx_add_rms_norm_slabs_ex followed by the product:
    residual, x and the outputs bit for bit; the hand-over areas report no give-up.  Includes shapes with
    fewer workgroups than rows (every producer workgroup takes several rows) and many back-to-back launches
    on the SAME buffers with changing inputs; between the fused launches an ordinary launch re-reads the same x
    buffer from every CU, so that every L2 (and L1) holds the OLD x when the next fused launch rewrites it — a
    consumer served by a stale line would show."""
    from hydrainfer_amd._C.kernel import gemm
    from hydrainfer_amd._C.kernel.norm import add_rms_norm_slabs
    for (M, hid, inter, S_in) in ((32, 4096, 11008, 4), (7, 4096, 11008, 3), (17, 5120, 13824, 4), (32, 256, 512, 1),
                                  (3, 64, 128, 2), (32, 64, 32, 1)):
        g = torch.Generator().manual_seed(hid + M)
        nw = torch.randn(hid, generator=g).to(dt).to(DEV)
        wq = (torch.randn((3 * hid, hid), generator=g) * 0.03).to(dt).to(DEV)
        wgu = (torch.randn((2 * inter, hid), generator=g) * 0.03).to(dt).to(DEV)
        assert gemm.norm_xreg_supported(M, 3 * hid, hid, dt) and gemm.norm_xreg_supported(M, 2 * inter, hid, dt, gate_up=True)
        pq, pg = gemm.pack_weight_xreg(wq), gemm.pack_weight_xreg(wgu, interleave_halves=True)
        xf, xf2 = (torch.zeros(gemm.fragment_major_elems(M, hid), dtype=dt, device=DEV) for _ in range(2))
        a = torch.zeros(gemm.xreg_workspace_floats(M, 3 * hid, hid), dtype=torch.float32, device=DEV)
        b, c = torch.zeros_like(a), torch.zeros_like(a)
        act1, act2, act3 = (torch.zeros(gemm.fragment_major_elems(M, inter), dtype=dt, device=DEV) for _ in range(3))
        n_it = 40 if hid == 4096 and M == 32 else 3
        sync = torch.zeros((2 * n_it, gemm.XREG_SYNC_WORDS), dtype=torch.int32, device=DEV)
        for it in range(n_it):
            slabs = torch.randn((S_in, M, hid), generator=g).to(DEV)
            h = torch.randn((M, hid), generator=g).to(dt).to(DEV)
            h1, h2, h3, h4 = h.clone(), h.clone(), h.clone(), h.clone()
            add_rms_norm_slabs(xf, h1, slabs, S_in, nw, 1e-5, fragment_major=True)
            s = gemm.linear_decode_partial_xreg(xf, pq, 3 * hid, a, frag_shape=(M, hid))
            s2 = gemm.norm_linear_decode_xreg(h2, slabs, S_in, nw, 1e-5, xf2, pq, 3 * hid, b, sync[2 * it])
            assert s == s2 and torch.equal(h1, h2) and torch.equal(a, b), f"qkv it={it} M={M} hid={hid}"
            assert torch.equal(gemm.from_fragment_major(xf, M, hid), gemm.from_fragment_major(xf2, M, hid))
            gemm.linear_decode_partial_xreg(xf2, pq, 3 * hid, c, frag_shape=(M, hid))     # every CU caches this x
            assert torch.equal(a, c)
            add_rms_norm_slabs(xf, h3, slabs, S_in, nw, 1e-5, fragment_major=True)
            gemm.gate_up_silu_xreg(xf, pg, inter, act1, frag_shape=(M, hid))
            gemm.norm_gate_up_silu_xreg(h4, slabs, S_in, nw, 1e-5, xf2, pg, inter, act2, sync[2 * it + 1])
            assert torch.equal(h3, h4), f"gate|up residual it={it}"
            gemm.gate_up_silu_xreg(xf2, pg, inter, act3, frag_shape=(M, hid))                 # ... and this one
            assert torch.equal(act1, act3)
            assert torch.equal(gemm.from_fragment_major(act1, M, inter), gemm.from_fragment_major(act2, M, inter)), f"act it={it}"
        assert int(sync[:, 1].abs().sum()) == 0          # no workgroup gave up waiting
        assert (sync[:, 0] == M).all()                   # every launch counted each of its M rows in exactly once
    assert not gemm.norm_xreg_supported(32, 4096, 11008, dt)   # K = 11008 takes several splits
    assert gemm.norm_xreg_supported(33, 4096, 4096, dt) and not gemm.norm_xreg_supported(65, 4096, 4096, dt)   # 33 .. 64 rows: the wide form
    assert not gemm.norm_xreg_supported(33, 22016, 4096, dt, gate_up=True)      # ... which has no fused silu*mul


def test_norm_fused_launch_rescues_rows_nobody_produced():
    """The hand-over must not depend on WHICH workgroups are resident (two processes on one GPU interleave
    their workgroups per XCD: a producer that is not dispatched yet may sit behind the other launch's
    waiters).  Test hook xreg_no_producers: no workgroup computes its row up front — after 30 us the
    waiting workgroups claim the rows themselves.  Same results bit for bit, no give-up."""
    from hydrainfer_amd import _lib
    from hydrainfer_amd._C.kernel import gemm
    from hydrainfer_amd._C.kernel.norm import add_rms_norm_slabs
    dt, M, hid, S_in = torch.bfloat16, 32, 4096, 4
    g = torch.Generator().manual_seed(11)
    nw = torch.randn(hid, generator=g).to(dt).to(DEV)
    wq = (torch.randn((3 * hid, hid), generator=g) * 0.03).to(dt).to(DEV)
    pq = gemm.pack_weight_xreg(wq)
    slabs = torch.randn((S_in, M, hid), generator=g).to(DEV)
    h = torch.randn((M, hid), generator=g).to(dt).to(DEV)
    h1, h2 = h.clone(), h.clone()
    xf, xf2 = (torch.zeros(gemm.fragment_major_elems(M, hid), dtype=dt, device=DEV) for _ in range(2))
    a = torch.zeros(gemm.xreg_workspace_floats(M, 3 * hid, hid), dtype=torch.float32, device=DEV)
    b = torch.zeros_like(a)
    add_rms_norm_slabs(xf, h1, slabs, S_in, nw, 1e-5, fragment_major=True)
    gemm.linear_decode_partial_xreg(xf, pq, 3 * hid, a, frag_shape=(M, hid))
    sync = torch.zeros(gemm.XREG_SYNC_WORDS, dtype=torch.int32, device=DEV)
    lib = _lib.lib()
    assert lib.hx_debug_set_option(b"xreg_no_producers", 1) == 0
    try:
        gemm.norm_linear_decode_xreg(h2, slabs, S_in, nw, 1e-5, xf2, pq, 3 * hid, b, sync)
        torch.cuda.synchronize()
    finally:
        lib.hx_debug_set_option(b"xreg_no_producers", 0)
    assert int(sync[1]) == 0 and int(sync[0]) == M
    assert torch.equal(h1, h2) and torch.equal(a, b)
    assert torch.equal(gemm.from_fragment_major(xf, M, hid), gemm.from_fragment_major(xf2, M, hid))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_embed_rms_norm_and_argmax_rows_equal_the_torch_ops(dt):
    """The two step-edge fusions of the decode loop: hx_embed_rms_norm == embedding + rms_norm (which is
    oracle-checked above) bit for bit; hx_argmax_rows == torch.argmax including ties (smallest index) and NaN."""
    from hydrainfer_amd._C.kernel.norm import argmax_rows, embed_rms_norm, rms_norm
    g = torch.Generator().manual_seed(5)
    for rows, vocab, hidden, idt in ((32, 32064, 4096, torch.int64), (5, 100, 5120, torch.int32), (1, 7, 64, torch.int64),
                                     (64, 1000, 8192, torch.int32)):
        table = torch.randn((vocab, hidden), generator=g).to(dt).to(DEV)
        w = torch.randn(hidden, generator=g).to(dt).to(DEV)
        ids = torch.randint(0, vocab, (rows,), generator=g).to(idt).to(DEV)
        h, x = embed_rms_norm(ids, table, w, 1e-5)
        h_ref = torch.nn.functional.embedding(ids.long(), table)
        x_ref = torch.empty_like(h_ref)
        rms_norm(x_ref, h_ref, w, 1e-5)
        assert torch.equal(h, h_ref) and torch.equal(x, x_ref)
    for rows, n in ((32, 32064), (3, 1001), (1, 8), (17, 50000)):
        logits = torch.randn((rows, n), generator=g).to(dt).to(DEV)
        assert torch.equal(argmax_rows(logits), torch.argmax(logits, dim=-1))
        logits[0, n // 2] = logits[0].max()          # a tie: bf16 / fp16 values repeat anyway, force one more
        logits[-1, 3] = 1e4
        logits[-1, n - 1] = 1e4
        assert torch.equal(argmax_rows(logits), torch.argmax(logits, dim=-1))
        logits[0, 5] = float("nan")
        assert torch.equal(argmax_rows(logits), torch.argmax(logits, dim=-1))
        view = logits[:, : n - 3]                     # row stride != n, unaligned tail
        assert torch.equal(argmax_rows(view), torch.argmax(view, dim=-1))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_single_entry_decode_linear_dispatches_to_the_layout_kernels(dt):
    """hx_decode_weight_plan / _pack / hx_linear_decode_ex (the one entry a maintainer binds) pick the layout for the
    announced batch size and give the layout-specific kernels' results bit for bit."""
    from hydrainfer_amd import _lib
    from hydrainfer_amd._C.kernel import gemm
    g = torch.Generator().manual_seed(3)
    for (N, K, M) in ((4096, 4096, 32), (4096, 11008, 7), (768, 256, 20)):
        w = (torch.randn((N, K), generator=g) * 0.05).to(dt).to(DEV)
        x = torch.randn((M, K), generator=g).to(dt).to(DEV)
        # <= 32 rows: activations in registers
        dw = gemm.DecodeWeight(w, max_rows=32)
        assert dw.layout == "xreg" and not dw.interleaved and torch.equal(dw.packed, gemm.pack_weight_xreg(w))
        a = torch.zeros(dw.workspace_floats(M), dtype=torch.float32, device=DEV)
        b = torch.zeros_like(a)
        assert gemm.linear_decode_ex(x, dw, a) == gemm.linear_decode_partial_xreg(x, dw.packed, N, b) and torch.equal(a, b)
        xf = gemm.to_fragment_major(x)
        a.zero_()
        gemm.linear_decode_ex(xf, dw, a, frag_shape=(M, K))
        assert torch.equal(a, b)
        # up to 64 rows: the same layout and packing where the wide kernel can read it (round 4: all three shapes here) ...
        dl = gemm.DecodeWeight(w, max_rows=64)
        assert dl.layout == "xreg" and torch.equal(dl.packed, dw.packed)
        # ... and the LDS-slice layout on request (a projection whose input arrives row-major) or where it cannot
        dl = gemm.DecodeWeight(w, max_rows=64, lds_slice=True)
        assert dl.layout == "lds_slice" and torch.equal(dl.packed, gemm.pack_weight(w))
        a2 = torch.zeros(dl.workspace_floats(M), dtype=torch.float32, device=DEV)
        b2 = torch.zeros_like(a2)
        assert gemm.linear_decode_ex(x, dl, a2) == gemm.linear_decode_partial_packed(x, dl.packed, N, b2) and torch.equal(a2, b2)
        with pytest.raises(_lib.HydraHipError):
            gemm.linear_decode_ex(xf, dl, a2, frag_shape=(M, K))        # fragment-major x only on the XREG layout
    w13 = (torch.randn((15360, 5120), generator=g) * 0.05).to(dt).to(DEV)      # 13B qkv: 40 k-steps per wave, halved to 20 (round 5)
    assert gemm.DecodeWeight(w13, max_rows=64).layout == "xreg" and gemm.DecodeWeight(w13, max_rows=32).layout == "xreg"
    wodd = (torch.randn((1024, 3584), generator=g) * 0.05).to(dt).to(DEV)      # 29 k-steps per wave: no built half (15)
    assert gemm.DecodeWeight(wodd, max_rows=64).layout == "lds_slice" and gemm.DecodeWeight(wodd, max_rows=32).layout == "xreg"
    # a gate|up weight: interleaved for the fused epilogue; the plain product over it un-interleaves its slab columns
    wgu = (torch.randn((2 * 11008, 4096), generator=g) * 0.05).to(dt).to(DEV)
    dgu = gemm.DecodeWeight(wgu, max_rows=32, gate_up=True)
    assert dgu.layout == "xreg" and dgu.interleaved and torch.equal(dgu.packed, gemm.pack_weight_xreg(wgu, interleave_halves=True))
    x4 = torch.randn((4, 4096), generator=g).to(dt).to(DEV)
    ws4 = torch.zeros(dgu.workspace_floats(4), dtype=torch.float32, device=DEV)
    s4 = gemm.linear_decode_ex(x4, dgu, ws4)
    ref4 = x4.float() @ wgu.float().t()
    assert (ws4.view(s4, 4, 22016).sum(0) - ref4).abs().max().item() <= 1e-5 * ref4.abs().max().item() + 1e-6
    d64 = gemm.DecodeWeight(wgu, max_rows=64, gate_up=True)
    assert d64.layout == "xreg" and d64.interleaved and torch.equal(d64.packed, dgu.packed)
    with pytest.raises(_lib.HydraHipError):
        gemm.DecodeWeight(wgu, max_rows=65)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_decode_step_head_equals_the_launches_it_replaces(dt):
    """hx_decode_step_head (embedding gather + first RMSNorm, look-ahead feed, zeroing of the hand-over areas, metadata
    advance: ONE launch) against hx_decode_feed_ids -> hx_embed_rms_norm, hx_memset_zero and hx_decode_advance run one
    after the other: every output bit for bit, for all combinations of the optional parts, 7B / 13B / small widths,
    batches past one scan block."""
    from hydrainfer_amd import _lib
    from hydrainfer_amd._C.kernel.norm import StepHead, decode_step_head, embed_rms_norm
    lib = _lib.lib()
    i32 = dict(dtype=torch.int32, device=DEV)
    for (rows, hidden, vocab, batch, bs) in ((32, 4096, 32064, 32, 16), (7, 5120, 32064, 7, 16), (64, 256, 999, 300, 32), (1, 64, 50, 1, 16)):
        g = torch.Generator().manual_seed(rows + hidden)
        table = torch.randn((vocab, hidden), generator=g).to(dt).to(DEV)
        w = torch.randn(hidden, generator=g).to(dt).to(DEV)
        for ids_dt in (torch.int64, torch.int32):
            for with_feed in (False, True):
                for with_zero in (False, True):
                    for with_adv in (False, True):
                        ids = torch.randint(0, vocab, (rows,), generator=g).to(ids_dt).to(DEV)
                        prev = torch.randint(0, vocab, (rows,), generator=g).to(DEV)
                        src = torch.randint(-1, rows, (rows,), generator=g).to(torch.int32).to(DEV)
                        # --- the separate launches
                        if with_feed:
                            fed = torch.empty(rows, dtype=torch.int64, device=DEV)
                            ids32 = ids.to(torch.int32)
                            _lib.check(lib.hx_decode_feed_ids(fed.data_ptr(), ids32.data_ptr(), src.data_ptr(), prev.data_ptr(), rows,
                                                              _lib.current_stream()), "feed")
                            h1, x1 = embed_rms_norm(fed, table, w, 1e-5)
                        else:
                            h1, x1 = embed_rms_norm(ids, table, w, 1e-5)
                        blocks_per = 8
                        pos = torch.randint(0, blocks_per * bs - 3, (batch,), generator=g).to(torch.int32).to(DEV)
                        kvl = (pos + 1).clone()
                        tbl = torch.randperm(batch * blocks_per, generator=g).to(torch.int32).to(DEV)
                        cub = torch.arange(0, (batch + 1) * blocks_per, blocks_per, **i32)
                        adv1 = [pos.clone(), kvl.clone(), torch.zeros(batch + 1, **i32), torch.zeros(batch, **i32)]
                        adv2 = [pos.clone(), kvl.clone(), torch.zeros(batch + 1, **i32), torch.zeros(batch, **i32)]
                        _lib.check(lib.hx_decode_advance(adv1[0].data_ptr(), adv1[1].data_ptr(), adv1[2].data_ptr(), adv1[3].data_ptr(),
                                                         tbl.data_ptr(), cub.data_ptr(), batch, bs, 2, _lib.current_stream()), "adv")
                        area = torch.full((4, 2, 512 + 3), 7, **i32)
                        # --- one launch
                        head = StepHead()
                        if with_adv:
                            head = StepHead(positions=adv2[0], kv_lens=adv2[1], cu_seqlens_k=adv2[2], new_cache_slots=adv2[3],
                                            block_table=tbl, cu_block_lens=cub, batch=batch, block_size=bs, stride=2)
                        if with_feed:
                            head.feed_src, head.feed_prev = src, prev
                        ids_in = ids.to(torch.int32) if with_feed else ids
                        h2, x2 = decode_step_head(ids_in, table, w, 1e-5, zero=area if with_zero else None, head=head)
                        torch.cuda.synchronize()
                        assert torch.equal(h1, h2) and torch.equal(x1, x2), (rows, hidden, ids_dt, with_feed)
                        assert int(area.abs().sum()) == (0 if with_zero else 7 * area.numel())
                        if with_adv:
                            assert all(torch.equal(a, b) for a, b in zip(adv1, adv2))
                        else:
                            assert torch.equal(adv2[0], pos) and int(adv2[2].abs().sum()) == 0
    # argument checks
    with pytest.raises(_lib.HydraHipError):
        decode_step_head(ids.float(), table, w, 1e-5)
    with pytest.raises(_lib.HydraHipError):
        decode_step_head(ids, table, w, 1e-5, head=StepHead(feed_src=src))


# ---- 33 .. 64 rows: the wide form of the activations-in-registers kernel ------------------------------------
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [33, 48, 64])
def test_xreg_wide_product_matches_fp32_and_is_repeatable(dt, M):
    """hx_linear_decode_partial_xreg / hx_gate_up_xreg at 33 .. 64 rows (gemm_xreg_wide_kernel: half a packed K split
    per workgroup over the SAME packing as the <= 32-row kernel, pairs of row groups, per-pair reduction through LDS)
    against the fp32 product of the same T inputs (rel. 1e-5 of the largest output: only the summation order differs),
    row-major x and fragment-major x bit-identical, two runs bit-identical, slab counts as planned; the packing is the
    one the 32-row kernel reads (checked by running both kernels on it)."""
    from hydrainfer_amd._C.kernel import gemm
    from hydrainfer_amd._C.kernel.activation import silu_and_mul_slabs
    # (15360, 5120) / (5120, 13824): LLaVA-1.5-13B's qkv and down (round 5: 20 k-steps per wave with one row group per
    # unit; 27 k-steps per wave of the packing halved unevenly, 14 + 13)
    for (N, K) in ((12288, 4096), (4096, 11008), (4096, 4096), (64, 256), (3072, 1024), (1024, 2816), (96, 128),
                   (15360, 5120), (5120, 13824), (5120, 5120)):
        assert gemm.xreg_supported(M, N, K, dt), (N, K)
        g = torch.Generator().manual_seed(N + K + M)
        x = torch.randn((M, K), generator=g).to(dt).to(DEV)
        w = (torch.randn((N, K), generator=g) * 0.02).to(dt).to(DEV)
        pk = gemm.pack_weight_xreg(w)
        a = torch.zeros(gemm.xreg_workspace_floats(M, N, K), dtype=torch.float32, device=DEV)
        b, c = torch.zeros_like(a), torch.zeros_like(a)
        s = gemm.linear_decode_partial_xreg(x, pk, N, a)
        assert s == a.numel() // (M * N) and s == {(12288, 4096): 2, (4096, 4096): 2, (4096, 11008): 8, (15360, 5120): 2, (5120, 13824): 8}.get((N, K), s)      # twice the slabs of the <= 32-row launch
        assert gemm.linear_decode_partial_xreg(x, pk, N, b) == s
        assert gemm.linear_decode_partial_xreg(gemm.to_fragment_major(x), pk, N, c, frag_shape=(M, K)) == s
        assert torch.equal(a, b) and torch.equal(a, c), f"N={N} K={K}"
        ref = x.float() @ w.float().t()
        got = a.view(s, M, N).sum(0)
        assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-6, f"N={N} K={K} M={M} {dt}"
        # the same packing under the 32-row kernel (first 32 rows)
        a32 = torch.zeros(gemm.xreg_workspace_floats(32, N, K), dtype=torch.float32, device=DEV)
        s32 = gemm.linear_decode_partial_xreg(x[:32].contiguous(), pk, N, a32)
        got32 = a32.view(s32, 32, N).sum(0)
        assert (got32 - ref[:32]).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-6
    # gate|up over the INTERLEAVED packing: slabs in [gate | up] order, then silu*mul == the fp32 formula
    for (inter, K) in ((11008, 4096), (2816, 1024), (96, 64), (13824, 5120)):
        assert gemm.gate_up_xreg_supported(M, inter, K, dt)
        g = torch.Generator().manual_seed(inter + M)
        x = torch.randn((M, K), generator=g).to(dt).to(DEV)
        w = (torch.randn((2 * inter, K), generator=g) * 0.03).to(dt).to(DEV)
        pg = gemm.pack_weight_xreg(w, interleave_halves=True)
        ws = torch.zeros(gemm.gate_up_xreg_workspace_floats(M, inter, K), dtype=torch.float32, device=DEV)
        s = gemm.gate_up_xreg(gemm.to_fragment_major(x), pg, inter, ws, frag_shape=(M, K))
        gu = x.float() @ w.float().t()
        got = ws.view(s, M, 2 * inter).sum(0)
        assert (got - gu).abs().max().item() <= 1e-5 * gu.abs().max().item() + 1e-6, f"gate|up inter={inter} K={K}"
        act = gemm.from_fragment_major(silu_and_mul_slabs(ws, s, M, inter, dt, fragment_major=True), M, inter)
        want = silu_and_mul_slabs(ws, s, M, inter, dt)
        assert torch.equal(act, want)
        if M <= 32 + 16:       # the same packing through the fused <= 32-row launch gives the same activations up to T round-off
            a32 = torch.zeros(gemm.fragment_major_elems(32, inter), dtype=dt, device=DEV)
            gemm.gate_up_silu_xreg(gemm.to_fragment_major(x[:32].contiguous()), pg, inter, a32, frag_shape=(32, K))
            ulp = 2.0 ** (-10 if dt == torch.float16 else -7)
            d = (gemm.from_fragment_major(a32, 32, inter).float() - want[:32].float()).abs()
            assert (d <= 4 * ulp * want[:32].float().abs() + 0.25 * ulp).all()
    assert not gemm.xreg_supported(65, 4096, 4096, dt)
    assert not gemm.gate_up_silu_supported(33, 11008, 4096, dt)          # the fused epilogue stays a <= 32-row launch
    # the single dispatching entry, planned for 64 rows: the activations-in-registers layout (one packing for 1 .. 64
    # rows), a gate|up weight keeps its halves interleaved and hx_linear_decode_ex un-interleaves the slab columns;
    # few rows through the same entry (the wide kernel with one or two 16-row blocks of x)
    g = torch.Generator().manual_seed(7 + M)
    w = (torch.randn((2 * 2816, 1024), generator=g) * 0.03).to(dt).to(DEV)
    for gate_up in (False, True):
        dw = gemm.DecodeWeight(w, max_rows=64, gate_up=gate_up)
        assert dw.layout == "xreg" and dw.interleaved == gate_up
        for rows in (M, 8):
            x = torch.randn((rows, 1024), generator=g).to(dt).to(DEV)
            ws = torch.zeros(dw.workspace_floats(rows), dtype=torch.float32, device=DEV)
            s = gemm.linear_decode_ex(x, dw, ws)
            ref = x.float() @ w.float().t()
            got = ws.view(s, rows, 2 * 2816).sum(0)
            assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-6, (gate_up, rows)
    assert gemm.DecodeWeight(w, max_rows=64, lds_slice=True).layout == "lds_slice"


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_norm_fused_wide_launches_are_bit_identical(dt):
    """hx_norm_linear_decode_xreg / hx_norm_gate_up_xreg at 33 .. 64 rows (up to 64 producer workgroups) ==
    hx_add_rms_norm_slabs_ex followed by the wide product: residual, x and the slabs bit for bit; 30 launches on the
    same buffers with changing inputs, every launch's hand-over area clean; also with the rescue path alone."""
    from hydrainfer_amd import _lib
    from hydrainfer_amd._C.kernel import gemm
    from hydrainfer_amd._C.kernel.norm import add_rms_norm_slabs
    lib = _lib.lib()
    for (M, hid, inter, S_in) in ((64, 4096, 11008, 8), (33, 4096, 11008, 4), (48, 1024, 2816, 2), (64, 256, 512, 1),
                                  (64, 5120, 13824, 8), (33, 5120, 13824, 2)):      # 13B width: 20 k-steps per wave, one row group per unit
        g = torch.Generator().manual_seed(hid + M)
        nw = torch.randn(hid, generator=g).to(dt).to(DEV)
        wq = (torch.randn((3 * hid, hid), generator=g) * 0.03).to(dt).to(DEV)
        wgu = (torch.randn((2 * inter, hid), generator=g) * 0.03).to(dt).to(DEV)
        assert gemm.norm_xreg_supported(M, 3 * hid, hid, dt) and gemm.gate_up_xreg_supported(M, inter, hid, dt, with_norm=True)
        pq, pg = gemm.pack_weight_xreg(wq), gemm.pack_weight_xreg(wgu, interleave_halves=True)
        xf, xf2 = (torch.zeros(gemm.fragment_major_elems(M, hid), dtype=dt, device=DEV) for _ in range(2))
        a = torch.zeros(gemm.xreg_workspace_floats(M, 3 * hid, hid), dtype=torch.float32, device=DEV)
        b = torch.zeros_like(a)
        ga = torch.zeros(gemm.gate_up_xreg_workspace_floats(M, inter, hid), dtype=torch.float32, device=DEV)
        gb = torch.zeros_like(ga)
        n_it = 30 if hid == 4096 and M == 64 else (10 if hid == 5120 and M == 64 else 3)
        sync = torch.zeros((2 * n_it + 2, gemm.XREG_SYNC_WORDS), dtype=torch.int32, device=DEV)
        for it in range(n_it):
            slabs = torch.randn((S_in, M, hid), generator=g).to(DEV)
            h = torch.randn((M, hid), generator=g).to(dt).to(DEV)
            h1, h2, h3, h4 = h.clone(), h.clone(), h.clone(), h.clone()
            add_rms_norm_slabs(xf, h1, slabs, S_in, nw, 1e-5, fragment_major=True)
            s = gemm.linear_decode_partial_xreg(xf, pq, 3 * hid, a, frag_shape=(M, hid))
            s2 = gemm.norm_linear_decode_xreg(h2, slabs, S_in, nw, 1e-5, xf2, pq, 3 * hid, b, sync[2 * it])
            assert s == s2 and torch.equal(h1, h2) and torch.equal(a, b), f"qkv it={it} M={M} hid={hid}"
            assert torch.equal(gemm.from_fragment_major(xf, M, hid), gemm.from_fragment_major(xf2, M, hid))
            add_rms_norm_slabs(xf, h3, slabs, S_in, nw, 1e-5, fragment_major=True)
            sg = gemm.gate_up_xreg(xf, pg, inter, ga, frag_shape=(M, hid))
            sg2 = gemm.norm_gate_up_xreg(h4, slabs, S_in, nw, 1e-5, xf2, pg, inter, gb, sync[2 * it + 1])
            assert sg == sg2 and torch.equal(h3, h4) and torch.equal(ga, gb), f"gate|up it={it} M={M} hid={hid}"
        # every row produced by the rescue path only
        assert lib.hx_debug_set_option(b"xreg_no_producers", 1) == 0
        try:
            h5 = h.clone()
            gemm.norm_linear_decode_xreg(h5, slabs, S_in, nw, 1e-5, xf2, pq, 3 * hid, b, sync[2 * n_it])
            torch.cuda.synchronize()
        finally:
            lib.hx_debug_set_option(b"xreg_no_producers", 0)
        assert torch.equal(h5, h1) and torch.equal(a, b)
        assert int(sync[:, 1].abs().sum()) == 0
        assert (sync[:2 * n_it + 1, 0] == M).all()


def test_copy_words2_moves_words_between_device_and_pinned_memory():
    """hx_copy_words2 (the engine decode loop's way in and out, engine/graph_decode.py): pinned -> device, device ->
    pinned, two pairs in one launch, visible to the host behind an event; odd counts, one empty pair, misalignment refused."""
    from hydrainfer_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    host_in = torch.arange(1000, dtype=torch.int32).pin_memory()
    d = torch.zeros(1000, dtype=torch.int32, device=dev)
    _lib.check(lib.hx_copy_words2(d.data_ptr(), host_in.data_ptr(), 777, None, None, 0, _lib.current_stream()), "copy")
    torch.cuda.synchronize()
    assert torch.equal(d[:777].cpu(), host_in[:777]) and int(d[777:].abs().sum()) == 0
    toks = torch.randint(0, 1 << 40, (64,), dtype=torch.int64, device=dev)
    err = torch.tensor([5], dtype=torch.int32, device=dev)
    h_tok, h_err = torch.zeros(64, dtype=torch.int64).pin_memory(), torch.zeros(1, dtype=torch.int32).pin_memory()
    ev = torch.cuda.Event()
    _lib.check(lib.hx_copy_words2(h_tok.data_ptr(), toks.data_ptr(), 2 * 37, h_err.data_ptr(), err.data_ptr(), 1,
                                  _lib.current_stream()), "copy")
    ev.record()
    ev.synchronize()
    assert h_tok[:37].tolist() == toks[:37].cpu().tolist() and int(h_tok[37:].abs().sum()) == 0 and int(h_err[0]) == 5
    assert lib.hx_copy_words2(None, None, 0, None, None, 0, _lib.current_stream()) == 0
    assert lib.hx_copy_words2(d.data_ptr() + 2, host_in.data_ptr(), 4, None, None, 0, _lib.current_stream()) != 0
    assert lib.hx_copy_words2(None, host_in.data_ptr(), 4, None, None, 0, _lib.current_stream()) != 0


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_wide_gate_up_with_silu_in_the_launch_is_bit_identical(dt):
    """hx_norm_gate_up_silu_wide_xreg (33 .. 64 rows: a workgroup does both K halves of its (gate, up) pairs and writes
    silu(gate) * up itself — no slabs) == hx_norm_gate_up_xreg + hx_silu_and_mul_slabs_ex: residual, x and act bit for bit,
    repeated launches with changing inputs, also with the rescue path alone; rows past M of the last 16-row block are
    never written."""
    from hydrainfer_amd import _lib
    from hydrainfer_amd._C.kernel import gemm
    from hydrainfer_amd._C.kernel.activation import silu_and_mul_slabs
    lib = _lib.lib()
    hid, inter = 4096, 11008
    assert gemm.gate_up_silu_wide_supported(64, inter, hid, dt) and gemm.gate_up_silu_wide_supported(33, inter, hid, dt)
    assert not gemm.gate_up_silu_wide_supported(32, inter, hid, dt)        # <= 32 rows: the one-split kernel's own epilogue
    assert not gemm.gate_up_silu_wide_supported(64, 13824, 5120, dt)      # 13B: 20 k-steps per wave, one row group per unit
    assert not gemm.gate_up_silu_wide_supported(64, 2816, 1024, dt)       # K = 1024: halves of 4 k-steps per wave (not built)
    g = torch.Generator().manual_seed(5)
    nw = torch.randn(hid, generator=g).to(dt).to(DEV)
    wgu = (torch.randn((2 * inter, hid), generator=g) * 0.03).to(dt).to(DEV)
    pg = gemm.pack_weight_xreg(wgu, interleave_halves=True)
    for M, n_it in ((64, 16), (33, 6), (48, 6), (52, 4), (36, 4)):      # (the engine pads decode batches to multiples of 4)
        xf, xf2 = (torch.zeros(gemm.fragment_major_elems(M, hid), dtype=dt, device=DEV) for _ in range(2))
        ga = torch.zeros(gemm.gate_up_xreg_workspace_floats(M, inter, hid), dtype=torch.float32, device=DEV)
        sync = torch.zeros((2 * n_it + 2, gemm.XREG_SYNC_WORDS), dtype=torch.int32, device=DEV)
        act = torch.full((gemm.fragment_major_elems(M, inter),), 7.0, dtype=dt, device=DEV)
        for it in range(n_it):
            slabs = torch.randn((4, M, hid), generator=g).to(DEV)
            h = torch.randn((M, hid), generator=g).to(dt).to(DEV)
            h1, h2 = h.clone(), h.clone()
            s = gemm.norm_gate_up_xreg(h1, slabs, 4, nw, 1e-5, xf, pg, inter, ga, sync[2 * it])
            want = silu_and_mul_slabs(ga, s, M, inter, dt, fragment_major=True)
            gemm.norm_gate_up_silu_wide_xreg(h2, slabs, 4, nw, 1e-5, xf2, pg, inter, act, sync[2 * it + 1])
            torch.cuda.synchronize()
            assert torch.equal(h1, h2), f"residual M={M} it={it}"
            assert torch.equal(gemm.from_fragment_major(xf, M, hid), gemm.from_fragment_major(xf2, M, hid))
            assert torch.equal(gemm.from_fragment_major(act, M, inter), gemm.from_fragment_major(want, M, inter)), f"act M={M} it={it}"
        pad = (M + 15) // 16 * 16
        if pad > M:      # rows M .. of the last block keep what the buffer held
            full = act.view(inter // 32, pad // 16, 4, 16, 8)       # [k / 32][row block][(k % 32) / 8][row % 16][k % 8]
            assert bool((full[:, -1, :, M % 16:, :] == 7.0).all())
        assert lib.hx_debug_set_option(b"xreg_no_producers", 1) == 0
        try:
            h3 = h.clone()
            act2 = torch.zeros_like(act)
            gemm.norm_gate_up_silu_wide_xreg(h3, slabs, 4, nw, 1e-5, xf2, pg, inter, act2, sync[2 * n_it])
            torch.cuda.synchronize()
        finally:
            lib.hx_debug_set_option(b"xreg_no_producers", 0)
        assert torch.equal(h3, h1) and torch.equal(gemm.from_fragment_major(act2, M, inter), gemm.from_fragment_major(want, M, inter))
        assert int(sync[:, 1].abs().sum()) == 0
